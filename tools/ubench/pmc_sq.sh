# Development aid: one rocprofv3 --pmc pass (eight SQ counters, no traces) over a short bench run, summarised per
# kernel into gpurun_out/<tag>_pmc_sq_counters.json (on the GPU box).   bash tools/ubench/pmc_sq.sh [tag] [bench args]
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
tag=${1:-r03}; shift || true
rm -rf $O/pmc_sq_$tag
timeout -s KILL 240 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM -d $O/pmc_sq_$tag -o sq --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --traffic-child --steps 2 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> $O/pmc_sq_$tag.err || echo "rocprofv3 exit $?"
f=$(find $O/pmc_sq_$tag -name "*counter_collection.csv" | head -1)
test -s "$f" || { tail -5 $O/pmc_sq_$tag.err; exit 1; }
python3 $GRAFT_REPO_ROOT/tools/ubench/pmc_sq_summary.py "$f" "$O/${tag}_pmc_sq_counters.json"

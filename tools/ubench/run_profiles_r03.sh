# Development aid: the bench lines, rocprofv3 kernel stats and PMC passes that profiles/r03_* hold (run on the GPU box).
# Every GPU step has its own time limit; the first step that fails ends the script (nothing runs on the GPU after it).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
run() { out=$1; shift; lim=$1; shift; if ! timeout -k 10 $lim "$@" > $O/$out 2> $O/$out.err; then echo "FAILED: $*"; tail -5 $O/$out.err; exit 1; fi; }
run r03_bench_mixture_bf16x3.json 600 python bench.py
echo mix done
run r03_bench_mixture_f32.json 200 python bench.py --mode f32 --no-cpu-baseline
run r03_bench_gaussian_bf16x3.json 200 python bench.py --model gaussian --no-cpu-baseline
run r03_bench_gpd_bf16x3.json 200 python bench.py --model gpd --no-cpu-baseline
run r03_bench_mixture3_bf16x3.json 200 python bench.py --model mixture3 --no-cpu-baseline
run r03_bench_transformer_bf16x3.json 300 python bench.py --model transformer
for b in 16 4096; do run r03_bench_gpd_hurricane_b$b.json 200 python bench.py --model gpd --shape hurricane --batch $b --steps 50 --warmup 3 --graph; done
VTS_LSTM_FUSED_FWD=1 run r03_bench_mixture_bf16x3_fused_forward.json 200 python bench.py --no-cpu-baseline
echo benches done
python tools/ubench/hbm_write.py > $O/r03_hbm_rates.txt 2>&1 || true
rm -rf $O/prof_r03 $O/prof_r03_tr
# (rocprofv3 crashes while the profiled interpreter exits, after its files are on disk: the statistics file decides)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r03 -o r03 -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 > $O/r03_bench_mixture_bf16x3_under_rocprof.json 2> $O/prof_r03.err || echo "rocprofv3 exit $?"
test -s $O/prof_r03/r03_kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r03_tr -o r03 -- python3 bench.py --model transformer --no-cpu-baseline --steps 10 --warmup 2 > $O/r03_bench_transformer_bf16x3_under_rocprof.json 2> $O/prof_r03_tr.err || echo "rocprofv3 exit $?"
test -s $O/prof_r03_tr/r03_kernel_stats.csv
echo rocprof done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/r03_bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d.get("value_fwd_bwd_elbo"), r["kernel"], r["bound"], r["frac"], r.get("traffic"))
    except Exception as e: print(f, "ERR", e)
PY

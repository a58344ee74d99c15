# Development aid: rocprofv3 kernel statistics of the Gaussian (1-layer) bench (run on the GPU box).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
rm -rf $O/prof_gauss
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_gauss -- python3 bench.py --model gaussian --no-cpu-baseline --steps 10 --warmup 2 > $O/r01_bench_gaussian_bf16x3_under_rocprof.json 2> $O/prof_gauss.err
find $O/prof_gauss -name "*_kernel_stats.csv"

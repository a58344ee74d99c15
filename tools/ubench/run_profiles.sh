# Development aid: the bench lines, rocprofv3 kernel stats and PMC passes that profiles/ holds (run on the GPU box).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 300 python bench.py > $O/r01_bench_mixture_bf16x3.json 2> $O/b_mix.err
echo mix done
timeout -k 10 200 python bench.py --mode f32 --no-cpu-baseline > $O/r01_bench_mixture_f32.json 2> $O/b_f32.err
timeout -k 10 200 python bench.py --model gaussian --no-cpu-baseline > $O/r01_bench_gaussian_bf16x3.json 2> $O/b_g.err
timeout -k 10 200 python bench.py --model gpd --no-cpu-baseline > $O/r01_bench_gpd_bf16x3.json 2> $O/b_gpd.err
timeout -k 10 200 python bench.py --model mixture3 --no-cpu-baseline > $O/r01_bench_mixture3_bf16x3.json 2> $O/b_m3.err
timeout -k 10 300 python bench.py --model transformer > $O/r01_bench_transformer_bf16x3.json 2> $O/b_tr.err
echo benches done
rm -rf $O/prof_r01c
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r01c -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 > $O/r01_bench_mixture_bf16x3_under_rocprof.json 2> $O/prof_r01c.err
echo rocprof done
rm -rf $O/pmc_cf $O/pmc_cw
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_cf -- python3 tools/bench_kernels.py --mode 1 --only recurrent_fwd --iters 3 > $O/pmc_cf.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_cw -- python3 tools/bench_kernels.py --mode 1 --only recurrent_fwd --iters 3 > $O/pmc_cw.log 2>&1
echo pmc done
find $O/prof_r01c $O/pmc_cf $O/pmc_cw -name "*.csv" | head -20

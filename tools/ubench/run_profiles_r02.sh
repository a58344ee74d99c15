# Development aid: the bench lines, rocprofv3 kernel stats and PMC passes that profiles/r02_* hold (run on the GPU box).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 600 python bench.py --traffic > $O/r02_bench_mixture_bf16x3.json 2> $O/b_mix.err || tail -3 $O/b_mix.err
echo mix done
timeout -k 10 200 python bench.py --mode f32 --no-cpu-baseline > $O/r02_bench_mixture_f32.json 2> $O/b_f32.err || tail -3 $O/b_f32.err
timeout -k 10 200 python bench.py --model gaussian --no-cpu-baseline > $O/r02_bench_gaussian_bf16x3.json 2> $O/b_g.err || tail -3 $O/b_g.err
timeout -k 10 200 python bench.py --model gpd --no-cpu-baseline > $O/r02_bench_gpd_bf16x3.json 2> $O/b_gpd.err || tail -3 $O/b_gpd.err
timeout -k 10 200 python bench.py --model mixture3 --no-cpu-baseline > $O/r02_bench_mixture3_bf16x3.json 2> $O/b_m3.err || tail -3 $O/b_m3.err
timeout -k 10 300 python bench.py --model transformer --no-cpu-baseline > $O/r02_bench_transformer_bf16x3.json 2> $O/b_tr.err || tail -3 $O/b_tr.err
for b in 16 4096 65536; do timeout -k 10 200 python bench.py --model gpd --shape hurricane --batch $b --steps 20 --warmup 3 > $O/r02_bench_gpd_hurricane_b$b.json 2>$O/bh.err || tail -3 $O/bh.err; done
echo benches done
rm -rf $O/prof_r02
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r02 -o r02 -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 > $O/r02_bench_mixture_bf16x3_under_rocprof.json 2> $O/prof_r02.err
echo rocprof done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/r02_bench_*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d.get("value_fwd_bwd_elbo"), r["kernel"], r["bound"], r["frac"], r.get("traffic"))
    except Exception as e: print(f, "ERR", e)
PY

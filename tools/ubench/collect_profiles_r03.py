"""Development aid: copies what tools/ubench/run_profiles_r03.sh left under gpurun_out/ into profiles/ (bench lines,
rocprofv3 kernel statistics) and cuts profiles/r03_pmc_hbm_traffic.json out of the `--traffic` bench line."""
import glob, json, os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
O, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
for f in sorted(glob.glob(os.path.join(O, "r03_bench_*.json"))):
    line = open(f).read().strip().splitlines()[-1]
    d = json.loads(line)
    open(os.path.join(P, os.path.basename(f)), "w").write(line + "\n")
    r = d["roofline"]
    print(os.path.basename(f), d["value"], d["ms_per_step"], d.get("value_fwd_bwd_elbo"), r["kernel"], r["bound"], r["frac"], r.get("traffic"))
st = os.path.join(O, "prof_r03", "r03_kernel_stats.csv")
if os.path.exists(st):
    shutil.copy(st, os.path.join(P, "r03_rocprof_kernel_stats_mixture_bf16x3.csv"))
st = os.path.join(O, "prof_r03_tr", "r03_kernel_stats.csv")
if os.path.exists(st):
    shutil.copy(st, os.path.join(P, "r03_rocprof_kernel_stats_transformer_bf16x3.csv"))
for extra in ("r03_hbm_rates.txt",):
    if os.path.exists(os.path.join(O, extra)):
        shutil.copy(os.path.join(O, extra), os.path.join(P, extra))
d = json.loads(open(os.path.join(P, "r03_bench_mixture_bf16x3.json")).read())
tm = d["roofline"].get("traffic_measured")
if tm:
    out = {"what": "HBM-side bytes per launch measured by `python bench.py --traffic` (mixture, 4096 x 96 x 370, hidden 256, "
                   "bf16x3): two child runs under rocprofv3 --pmc (FETCH_SIZE, WRITE_SIZE; KiB), FETCH_SIZE doubled (gfx950 "
                   "tallies 128-byte read requests at 64 bytes; every large kernel of the step reads with 16-byte-per-lane "
                   "loads, the pattern that correction is calibrated for: the plane GEMM micro-benchmark reads 2.42 GB by "
                   "this count against 2.42 GB of operands), mean over the launches of a 3-step run",
           "kernels": tm,
           "dominant_kernel": d["roofline"]["kernel"],
           "algorithmic_bytes_per_launch_of_the_dominant_kernel":
               d["roofline"].get("algorithmic_bytes_per_launch", d["roofline"].get("hbm_view", {}).get("algorithmic_bytes_per_launch"))}
    json.dump(out, open(os.path.join(P, "r03_pmc_hbm_traffic.json"), "w"), indent=1)
    print("traffic", {k: v["hbm_bytes"] for k, v in tm.items()})

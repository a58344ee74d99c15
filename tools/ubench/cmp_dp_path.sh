set -e
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3_cmp_plain.json 2> gpurun_out/r3_cmp_plain.err
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3_cmp_tr.json 2> gpurun_out/r3_cmp_tr.err
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3_cmp_plain2.json 2> gpurun_out/r3_cmp_plain2.err
python - <<PY
import json
for f in ("plain","tr","plain2"):
    d=json.loads(open(f"gpurun_out/r3_cmp_{f}.json").read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["ms_per_step_fwd_bwd_elbo"], sum(v["avg_ms"]*v["launches_per_step"] for v in d["kernels"].values()))
PY

# Development aid: the bench with cooperative and with plain launches of the cluster / pair recurrences, same box.
set -e
run() { tag=$1; shift; env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3_coop_$tag.json 2> gpurun_out/r3_coop_$tag.err; }
run plain A=1
run coopfwd VTS_LSTM_CLUSTER_COOP=1
run coop VTS_LSTM_CLUSTER_COOP=1 VTS_LSTM_PAIR_COOP=1
run plain2 A=1
run coop2 VTS_LSTM_CLUSTER_COOP=1 VTS_LSTM_PAIR_COOP=1
python - <<PY
import json
for f in ("plain","coopfwd","coop","plain2","coop2"):
    d=json.loads(open(f"gpurun_out/r3_coop_{f}.json").read().strip().splitlines()[-1])
    k=d["kernels"]
    print(f, d["value"], d["ms_per_step"], d["ms_per_step_fwd_bwd_elbo"], k["lstm_recurrent_fwd"]["avg_ms"], k["lstm_recurrent_bwd"]["avg_ms"])
PY

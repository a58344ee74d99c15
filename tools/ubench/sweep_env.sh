# Development aid: bench.py under several values of one environment variable; prints the per-entry-point times.
#   bash tools/ubench/sweep_env.sh [--lib path/to/libvts_variant.so] VAR v1 v2 ...   (on the GPU box; ends at the first failing run)
set -e
LIB=""
if [ "$1" = "--lib" ]; then LIB=$2; shift 2; fi
VAR=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out
for v in "$@"; do
  if [ -n "$LIB" ]; then cmd="python tools/ubench/bench_with_lib.py $LIB"; else cmd="python bench.py"; fi
  env $VAR=$v timeout -k 10 200 $cmd --no-cpu-baseline --steps 10 --warmup 3 > $O/sweep_${VAR}_$v.json 2> $O/sweep.err || { tail -5 $O/sweep.err; exit 1; }
  python - "$O/sweep_${VAR}_$v.json" "$LIB $VAR=$v" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d["kernels"]
print(sys.argv[2], "ms/step", d["ms_per_step"], " ".join(f"{n.replace('lstm_','')}={v['avg_ms']:.3f}" for n, v in k.items() if v["avg_ms"] > 0.15))
PY
done

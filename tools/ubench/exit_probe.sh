#!/bin/bash
# Which object faults while a profiled interpreter exits?  Four runs of the same rocprofv3 command: torch alone,
# torch + libvts_hip.so loaded, torch + one vts call, the bench itself; each leaves its stderr (the crash report, if
# any) and its memory map under gpurun_out/exitprobe/.  tools/ubench/exit_probe_resolve.py matches them up.
set -u
export TMPDIR=/tmp
out=$PWD/gpurun_out/exitprobe
mkdir -p "$out"
for mode in torch load call coop nocoop; do
  unset VTS_LSTM_CLUSTER_COOP VTS_LSTM_PAIR_COOP
  if [ $mode = coop ]; then export VTS_LSTM_CLUSTER_COOP=1 VTS_LSTM_PAIR_COOP=1; fi
  timeout -k 10 150 rocprofv3 --kernel-trace --stats -d "$out/$mode" -o p --output-format csv -- \
    python3 tools/ubench/exit_probe.py $mode "$out/$mode.maps" > "$out/$mode.out" 2> "$out/$mode.err"
  echo "$mode: exit $?" | tee -a "$out/summary.txt"
done
unset VTS_LSTM_CLUSTER_COOP VTS_LSTM_PAIR_COOP
VTS_BENCH_DUMP_MAPS="$out/bench.maps" timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$out/bench" -o p --output-format csv -- \
  python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > "$out/bench.out" 2> "$out/bench.err"
echo "bench: exit $?" | tee -a "$out/summary.txt"
export VTS_LSTM_CLUSTER_COOP=1 VTS_LSTM_PAIR_COOP=1
VTS_BENCH_DUMP_MAPS="$out/benchcoop.maps" timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$out/benchcoop" -o p --output-format csv -- \
  python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > "$out/benchcoop.out" 2> "$out/benchcoop.err"
echo "benchcoop: exit $?" | tee -a "$out/summary.txt"
unset VTS_LSTM_CLUSTER_COOP VTS_LSTM_PAIR_COOP
python3 tools/ubench/exit_probe_resolve.py "$out" | tee "$out/resolved.txt"

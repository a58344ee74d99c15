# Development aid: kernel list of the native-shape mixture step under rocprofv3 (on the GPU box).  VTS_MODE picks the arithmetic.
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/prof_native
rm -rf $O
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o nat -- python3 $GRAFT_REPO_ROOT/tools/ubench/native_step_kernels.py 100 > /dev/null 2>&1
test -s $O/nat_kernel_stats.csv || exit 1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/nat_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total us per step", round(tot/100/1e3,1), "launches per step", sum(int(r["Calls"]) for r in rows)/100)
for r in rows[:24]:
    print(r["Name"][:100].ljust(100), int(r["Calls"])/100, round(float(r["AverageNs"])/1e3,1), "us", round(float(r["TotalDurationNs"])/100/1e3,1), "us/step")
PY

"""Development aid: per-kernel summary of a `rocprofv3 --pmc` counter_collection.csv holding SQ counters
(tools/ubench/pmc_sq.sh): mean per launch and the shares of a resident wave's cycles.

    python tools/ubench/pmc_sq_summary.py counter_collection.csv out.json"""
import collections
import csv
import json
import re
import sys

per = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
wg = {}
for row in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(\w+kernel\w*)(<[^(]*>)?\(", row["Kernel_Name"])
    k = (m.group(1) + (m.group(2) or "")) if m else row["Kernel_Name"][:60]
    per[k][row["Counter_Name"]] += float(row["Counter_Value"])
    launches[k].add(row["Dispatch_Id"])
    wg[k] = int(row["Workgroup_Size"])
out = {}
for k, c in per.items():
    n = len(launches[k])
    if c.get("SQ_WAVE_CYCLES", 0) / n < 2e7:      # the large kernels only
        continue
    wps = 2 if wg[k] == 512 else 1                # waves per SIMD: the plane GEMMs run eight waves per CU, the recurrences four
    wc = max(c["SQ_WAVE_CYCLES"], 1.0)
    out[k] = {"launches": n, "waves_per_simd": wps, **{cn: round(v / n) for cn, v in sorted(c.items())},
              "wave_cycles_parked_at_waitcnt_or_barrier": round(c["SQ_WAIT_ANY"] / wc, 3),
              "wave_cycles_stalled_at_issue": round(c["SQ_WAIT_INST_ANY"] / wc, 3),
              "wave_cycles_issuing": round(c["SQ_ACTIVE_INST_ANY"] / wc, 3),
              "matrix_pipe_busy_share_per_simd": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] * wps / (4.0 * wc), 3)}
json.dump({"what": "rocprofv3 --pmc, SQ counters only (no traces), summed over the device, mean per launch, over `python3 bench.py "
                   "--steps 2 --warmup 1` (mixture, 4096 x 96 x 370, hidden 256, bf16x3).  SQ_WAVE_CYCLES, SQ_WAIT_* and "
                   "SQ_ACTIVE_INST_* count quad-cycles of resident waves, SQ_VALU_MFMA_BUSY_CYCLES cycles per SIMD: the matrix "
                   "pipe's busy share of a SIMD is MFMA_BUSY * waves_per_simd / (4 * WAVE_CYCLES)",
           "kernels": out}, open(sys.argv[2], "w"), indent=1)
for k, v in out.items():
    print(k[:44].ljust(44), v["launches"], "parked", v["wave_cycles_parked_at_waitcnt_or_barrier"], "issue-stalled",
          v["wave_cycles_stalled_at_issue"], "issuing", v["wave_cycles_issuing"], "mfma busy", v["matrix_pipe_busy_share_per_simd"])

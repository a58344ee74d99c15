"""Match the addresses of the crash reports left by tools/ubench/exit_probe.sh to the libraries of each run's map."""
import os
import re
import sys

top = sys.argv[1]
for mode in ("torch", "load", "call", "coop", "nocoop", "bench", "benchcoop"):
    err, maps = os.path.join(top, mode + ".err"), os.path.join(top, mode + ".maps")
    if not (os.path.exists(err) and os.path.exists(maps)):
        print(f"== {mode}: files missing")
        continue
    regions = []
    for line in open(maps):
        f = line.split()
        lo, hi = (int(v, 16) for v in f[0].split("-"))
        regions.append((lo, hi, f[1], f[5] if len(f) > 5 else "[anon]"))
    text = open(err).read()
    crashed = "SIGSEGV" in text or "Aborted at" in text
    print(f"== {mode}: {'crash report present' if crashed else 'no crash report'}")
    if not crashed:
        continue
    seen = []
    for m in re.finditer(r"(?:PC: )?@\s+0x([0-9a-f]+)\s*(\S*)", text):
        a = int(m.group(1), 16)
        where = next((f"{os.path.basename(n)} +{a - lo:#x} [{p}]" for lo, hi, p, n in regions if lo <= a < hi), "(unmapped)")
        seen.append(f"   {a:#x}  {m.group(2):<20s} {where}")
    fault = re.search(r"SIGSEGV \(@0x([0-9a-f]+)\)", text)
    if fault:
        a = int(fault.group(1), 16)
        where = next((f"{os.path.basename(n)} +{a - lo:#x} [{p}]" for lo, hi, p, n in regions if lo <= a < hi), "(unmapped at the time of the dump)")
        print(f"   faulting address {a:#x}: {where}")
    print("\n".join(seen))

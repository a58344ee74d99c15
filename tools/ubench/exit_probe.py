"""Control for the crash that rocprofv3 reports while a profiled interpreter exits: a run that launches GPU kernels
through torch alone, optionally with libvts_hip.so loaded (and never called) or called once.  Writes the process's
memory map before it exits, so the crash report's addresses can be matched to libraries.

    rocprofv3 --kernel-trace --stats -d DIR -o p --output-format csv -- python3 tools/ubench/exit_probe.py MODE MAPS
    MODE: torch | load | call | coop | nocoop   (coop: a shape that takes the cluster / pair recurrences, launched
    cooperatively - VTS_LSTM_CLUSTER_COOP=1 VTS_LSTM_PAIR_COOP=1 in the environment; nocoop: the same shape with the
    default plain launches)
"""
import os
import sys

import torch

mode, maps = sys.argv[1], sys.argv[2]
x = torch.randn(2048, 2048, device="cuda")
for _ in range(5):
    x = x @ x * 1e-3
torch.cuda.synchronize()
if mode in ("load", "call", "coop", "nocoop"):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "vae-time-series_amd"))
    import vts
    from vts import _lib
    lib = _lib.load()
    print(lib.vts_version().decode(), file=sys.stderr)
    if mode == "call":
        m = vts.LSTM(8, 64).cuda()
        m(torch.randn(16, 4, 8, device="cuda"))[1][0].sum().backward()
        torch.cuda.synchronize()
    if mode in ("coop", "nocoop"):
        vts.set_mode("bf16x3")
        m = vts.LSTM(8, 256).cuda()
        m(torch.randn(128, 4, 8, device="cuda"))[1][0].sum().backward()
        torch.cuda.synchronize()
with open("/proc/self/maps") as src, open(maps, "w") as dst:
    dst.write(src.read())
print("probe done", mode, file=sys.stderr)

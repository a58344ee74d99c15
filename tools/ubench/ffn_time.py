"""Development aid: the transformer variant's feed-forward block (d_model 64, dim_feedforward 2048) forward + backward
at 393 216 tokens, a few times; run under `rocprofv3 --kernel-trace --stats` for per-kernel times.
    python tools/ubench/ffn_time.py [tokens]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vae-time-series_amd"))
import torch
import vts
from vts import functional as VF
vts.set_mode("bf16x3")
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 393216
d, F = 64, 2048
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(M, d, device=dev, generator=g).requires_grad_(True)
w1 = (torch.randn(F, d, device=dev, generator=g) / 8).requires_grad_(True)
b1 = (torch.randn(F, device=dev, generator=g) / 8).requires_grad_(True)
w2 = (torch.randn(d, F, device=dev, generator=g) / 45).requires_grad_(True)
b2 = (torch.randn(d, device=dev, generator=g) / 8).requires_grad_(True)
dy = torch.randn(M, d, device=dev, generator=g)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
for it in range(4):
    if it == 1:
        ev[0].record()
    y = VF.feed_forward(x, w1, b1, w2, b2, None, 0.2, 1234 + it)
    if it == 3:
        ev[1].record()
    y.backward(dy)
    for p in (x, w1, b1, w2, b2):
        p.grad = None
ev[2].record()
torch.cuda.synchronize()
print(f"{M} tokens: forward + backward {ev[0].elapsed_time(ev[2]) / 3:.3f} ms (last backward alone {ev[1].elapsed_time(ev[2]):.3f} ms)")

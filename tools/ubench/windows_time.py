"""Development aid: device time of the window builder at the bench shape (many launches between two events)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vae-time-series_amd"))
import torch, vts
from vts import _lib as L
dev = torch.device("cuda:0")
B, T, C, Nt = 4096, 96, 370, 35040
series = torch.randn(Nt, C, device=dev) * 3 + 1
src = vts.WindowSource(series, T, 1)
starts = torch.randint(0, len(src), (B,), device=dev)
lib = L.load()
X = torch.empty(B, T, C, device=dev); y = torch.empty(B, 1, device=dev)
for std in (False, True):
    if std: src.fit(starts)
    xm = src.x_stats[0].contiguous() if std else None
    xs = src.x_stats[1].contiguous() if std else None
    args = lambda: (L.ptr(series), Nt, C, L.wptr(starts), B, T, 1, 0, None if xm is None else L.wptr(xm),
                    None if xs is None else L.wptr(xs), None if not std else L.wptr(src.y_stats), L.ptr(X), L.ptr(y), L.stream())
    for _ in range(3): L.check(lib.vts_windows_build(*args()), "w")
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50): lib.vts_windows_build(*args())
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 50
    print(f"standardise={std}: {ms:.4f} ms  {4.0 * (B * T * C + Nt * C) / ms / 1e6:.0f} GB/s")

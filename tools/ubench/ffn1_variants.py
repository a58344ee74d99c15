"""Time the [393216 x 64] . [64 x 2048] product with different epilogues (where does the
write-heavy GEMM spend its time?)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vae-time-series_amd")]
import torch
import vts
from vts import _lib as L

dev = torch.device("cuda:0")
lib = L.load()
M, K, N = 393216, 64, 2048
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) / 8; b = torch.randn(N, device=dev)
y = torch.empty(M, N, device=dev)
w2 = torch.randn(K, N, device=dev) / 45; b2 = torch.randn(K, device=dev); y2 = torch.empty(M, K, device=dev)
mask = torch.empty(M * N // 8, dtype=torch.uint8, device=dev)

def timeit(name, f, iters=5):
    f(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): f()
    e.record(); torch.cuda.synchronize()
    print(f"{name:46s} {s.elapsed_time(e) / iters:8.3f} ms", flush=True)

for mode in (1,):
    st = L.stream()
    tag = "bf16x3" if mode else "f32   "
    timeit(f"{tag} no bias, no act", lambda: lib.vts_linear_fwd(L.ptr(x), K, L.ptr(w), None, L.ptr(y), M, N, K, 0, None, 1.0, None, mode, st))
    timeit(f"{tag} bias", lambda: lib.vts_linear_fwd(L.ptr(x), K, L.ptr(w), L.ptr(b), L.ptr(y), M, N, K, 0, None, 1.0, None, mode, st))
    timeit(f"{tag} bias + relu", lambda: lib.vts_linear_fwd(L.ptr(x), K, L.ptr(w), L.ptr(b), L.ptr(y), M, N, K, 1, None, 1.0, None, mode, st))
    timeit(f"{tag} bias + relu + hashed dropout", lambda: lib.vts_linear_relu_dropout_fwd(L.ptr(x), K, L.ptr(w), L.ptr(b), L.ptr(y), M, N, K, 0.2, 12345, mode, st))
    timeit(f"{tag} ffn_fwd (both products, hashed, mask bits)", lambda: lib.vts_ffn_fwd(L.ptr(x), K, L.ptr(w), L.ptr(b), L.ptr(w2), L.ptr(b2), None, 0.2, 12345, L.ptr(y), L.wptr(mask), L.ptr(y2), M, K, N, mode, st))
    timeit(f"{tag} ffn_fwd eval (no dropout), mask bits", lambda: lib.vts_ffn_fwd(L.ptr(x), K, L.ptr(w), L.ptr(b), L.ptr(w2), L.ptr(b2), None, 0.2, 0, L.ptr(y), L.wptr(mask), L.ptr(y2), M, K, N, mode, st))
    timeit(f"{tag} ffn_fwd hashed, no mask bits", lambda: lib.vts_ffn_fwd(L.ptr(x), K, L.ptr(w), L.ptr(b), L.ptr(w2), L.ptr(b2), None, 0.2, 12345, L.ptr(y), None, L.ptr(y2), M, K, N, mode, st))
    timeit(f"{tag} second product alone [M,2048]x[2048,64]", lambda: lib.vts_linear_fwd(L.ptr(y), N, L.ptr(w2), L.ptr(b2), L.ptr(y2), M, K, N, 0, None, 1.0, None, mode, st))

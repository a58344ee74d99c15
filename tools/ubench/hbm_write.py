"""Achieved HBM bandwidth of plain torch fill / copy / read-reduce over a 3.2 GB tensor (reference
points for the write-heavy GEMM epilogues)."""
import torch, time
dev = torch.device("cuda:0")
n = 393216 * 2048
a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
def t(f, iters=5):
    f(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
ms = t(lambda: a.fill_(1.0)); print("fill   %.3f ms  %.2f TB/s written" % (ms, n * 4 / ms / 1e9))
ms = t(lambda: b.copy_(a));   print("copy   %.3f ms  %.2f TB/s read + %.2f TB/s written" % (ms, n * 4 / ms / 1e9, n * 4 / ms / 1e9))
ms = t(lambda: a.sum());      print("sum    %.3f ms  %.2f TB/s read" % (ms, n * 4 / ms / 1e9))
ms = t(lambda: torch.relu_(a)); print("relu_  %.3f ms  %.2f TB/s read + same written" % (ms, n * 4 / ms / 1e9))

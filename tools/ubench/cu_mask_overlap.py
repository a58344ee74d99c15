"""Does the step gain from running two half batches side by side on disjoint halves of the compute units?

The plane GEMMs are bound by the matrix cores' clock under load, the recurrences by memory instructions.  If the chip
were limited per compute unit, two half batches on 128 units each, one phase apart (one in a GEMM while the other is
in a recurrence), would finish sooner than the whole batch on 256.  If the chip is limited by its power budget as a
whole, they would not.  Streams with a CU mask (hipExtStreamCreateWithCUMask) make the experiment:

  full       4096 windows on every unit, torch's stream
  half       2048 windows on a stream masked to half of the units (the other half idle)
  pair(d)    2048 + 2048 windows on two masked streams, the second started d microseconds later

Usage: cu_mask_overlap.py <mask word A, hex> <mask word B, hex> [model]
"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vae-time-series_amd")]
import copy
import torch
import vts
from vts import train as T
import bench

dev = torch.device("cuda:0")
word_a = int(sys.argv[1], 16) if len(sys.argv) > 1 else 0x0f0f0f0f
word_b = int(sys.argv[2], 16) if len(sys.argv) > 2 else 0xf0f0f0f0
kind = sys.argv[3] if len(sys.argv) > 3 else "mixture"
hip = ctypes.CDLL("libamdhip64.so")
helpers = ctypes.CDLL(os.path.join(ROOT, "tests", "helpers", "libvts_test_helpers.so"))
helpers.vts_test_occupy.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]


def masked_stream(word):
    """A stream restricted to the CUs of `word` (repeated eight times); a plain stream where the box refuses masks -
    the persistent grids are then sized to 128 workgroups of a whole CU's LDS each (VTS_GRID_CUS), so two of them
    still share the device half and half."""
    s = ctypes.c_void_p()
    mask = (ctypes.c_uint32 * 8)(*([word] * 8))
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, mask)
    if rc != 0:
        hip.hipGetLastError()         # the refusal must not surface in torch's next error check
        print(f"hipExtStreamCreateWithCUMask refused (error {rc}): plain streams, grids of 128 workgroups", flush=True)
        return torch.cuda.Stream(device=dev)
    return torch.cuda.ExternalStream(s.value, device=dev)


vts.set_mode("bf16x3")
torch.manual_seed(1234)
model = vts.build_model(kind, 370, 256, 32, 1).to(dev)
model.train()
x, y = bench.synth_windows(4096, 96, 370, 0, dev)
sink = torch.zeros(4, dtype=torch.int32, device=dev)
STEPS = 20


def run(m, xs, ys):
    return T.train_step(m, None, xs, ys)


def timed(fn, label, windows, late_us=0):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if late_us:                       # the second stream starts late and stays late: the two never synchronise
        helpers.vts_test_occupy(1, late_us, sink.data_ptr(), sb.cuda_stream)
    for _ in range(STEPS):
        fn()
    torch.cuda.synchronize()
    ms = (1e3 * (time.perf_counter() - t0) - late_us * 1e-3) / STEPS      # steady state: the head start taken off
    print(f"{label:38s} {ms:8.3f} ms per step  {windows / ms * 1e3:10.0f} windows/s", flush=True)
    return ms


from vts import _lib as L


def per_entry_point(fn, label):
    """Every entry point of one step bracketed with events: where a configuration's time goes."""
    torch.cuda.synchronize()
    L.timer.only = None
    L.timer.enabled = True
    L.timer.reset()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    L.timer.enabled = False
    rows = sorted(L.timer.summary().items(), key=lambda kv: -kv[1][1])
    print(f"  {label}: ms per call (calls per step)")
    for name, (n, tot) in rows[:9]:
        print(f"    {name:28s} {tot / n:7.3f}  ({n // 5})", flush=True)


timed(lambda: run(model, x, y), "full: 4096 windows, every CU", 4096)
per_entry_point(lambda: run(model, x, y), "full")

sa, sb = masked_stream(word_a), masked_stream(word_b)
os.environ["VTS_GRID_CUS"] = "128"
ma, mb = copy.deepcopy(model), copy.deepcopy(model)
xa, ya, xb, yb = x[:2048].contiguous(), y[:2048].contiguous(), x[2048:].contiguous(), y[2048:].contiguous()


def half_a():
    with torch.cuda.stream(sa):
        run(ma, xa, ya)


timed(half_a, f"half: 2048 windows, mask {word_a:08x}", 2048)
per_entry_point(half_a, "half")


def half_b():
    with torch.cuda.stream(sb):
        run(mb, xb, yb)


timed(half_b, f"half: 2048 windows, mask {word_b:08x}", 2048)

for delay_us in (0, 400, 800, 1200, 1600, 2400):
    def pair():
        for p in list(ma.parameters()) + list(mb.parameters()):
            p.grad = None
        with torch.cuda.stream(sa):
            la = T.forward_loss(ma, xa, ya)
        with torch.cuda.stream(sb):
            lb = T.forward_loss(mb, xb, yb)
        with torch.cuda.stream(sa):
            la.backward()
        with torch.cuda.stream(sb):
            lb.backward()

    timed(pair, f"pair: 2 x 2048, second {delay_us} us late", 4096, delay_us)

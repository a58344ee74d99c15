"""Development aid: per-phase cycle stamps of the bf16x3 GEMM k loop (needs libvts_stamp.so)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vae-time-series_amd"))
import torch
from vts import _lib as L
L.LIB_PATH = os.path.join(ROOT, "tools", "ubench", "libvts_stamp.so")
lib = L.load()
raw = C.CDLL(L.LIB_PATH)
raw.vts_debug_stamps.argtypes = [C.c_void_p, C.c_int]
dev = torch.device("cuda:0")
B, T, H = 4096, 96, 256
gates = L.workspace(lib.vts_lstm_gates_bytes(B, T, H), dev); gates.view(torch.float32).normal_()
hseq = torch.randn(B, T, H, device=dev)
dw = torch.empty(4 * H, H, device=dev)
wb = lib.vts_lstm_wgrad_workspace_bytes(B, T, H, H); ws = L.workspace(wb, dev)
st = L.stream()
Cin = 370
x = torch.randn(B, T, Cin, device=dev)
w_ih = torch.randn(4 * H, Cin, device=dev) / 16
w_h = torch.randn(4 * H, H, device=dev) / 16
bias = torch.zeros(4 * H, device=dev)
dx = L.workspace(lib.vts_lstm_seqgrad_bytes(B, T, H), dev)
stages = {
    "wgrad_hh": lambda: lib.vts_lstm_wgrad(L.wptr(gates), L.ptr(hseq), -1, L.ptr(dw), 0, L.wptr(ws), wb, B, T, H, H, 1, st),
    "inproj": lambda: lib.vts_lstm_inproj_fwd(L.ptr(x), L.ptr(w_ih), L.ptr(bias), L.ptr(bias), L.wptr(gates), B, T, Cin, H, 1, st),
    "dgrad": lambda: lib.vts_lstm_dgrad(L.wptr(gates), L.ptr(w_h), L.wptr(dx), B, T, H, H, 1, st),
}
names = ["issue loads", "lds reads + mfma", "barrier 1", "wait loads", "convert + store", "barrier 2"]
for nm, fn in stages.items():
    for it in range(2):
        raw.vts_debug_stamps(None, 1)
        L.check(fn(), nm)
        torch.cuda.synchronize()
    out = (C.c_ulonglong * 8)()
    raw.vts_debug_stamps(out, 0)
    v = list(out)
    nk, nwg = v[6], v[7]
    print(nm, "workgroups", nwg, "k-steps per workgroup", nk / max(nwg, 1))
    for n, x_ in zip(names, v[:6]):
        print(f"   {n:18s} {x_ / nk:9.1f} cycles per k-step")
    print("   sum", round(sum(v[:6]) / nk, 1))

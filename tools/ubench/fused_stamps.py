"""Development aid: per-phase cycle counts of the fused forward kernel (lstm_rec_fused.hip built with -DVTS_F_STAMP into
tools/ubench/libvts_fstamp.so by tools/ubench/build_fstamp.sh) and its launch time, at 4096 windows x 96 steps.
    python tools/ubench/fused_stamps.py [Cin]      (256: a layer above the first; 370: the first layer)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vae-time-series_amd"))
import torch
from vts import _lib as L
stamped = os.environ.get("VTS_STAMP_LIB") or os.path.join(ROOT, "tools", "ubench", "libvts_fstamp.so")
if os.path.exists(stamped) and os.environ.get("VTS_NO_STAMP") != "1":
    L.LIB_PATH = stamped
lib = L.load()
dev = torch.device("cuda:0")
B, T, H = 4096, 96, 256
Cin = int(sys.argv[1]) if len(sys.argv) > 1 else 256
first = Cin != 256
x = torch.randn(B, T, Cin, device=dev)
xp = L.workspace(lib.vts_lstm_planes_bytes(B, T, Cin, 0 if first else 1), dev)
if first:
    L.check(lib.vts_lstm_planes_from_rows(L.ptr(x), L.wptr(xp), B, T, Cin, H, 1, L.stream()), "to planes")
else:
    xp.view(torch.int16).copy_((torch.randn(xp.numel() // 2, device=dev) * 0.5).to(torch.bfloat16).view(torch.int16))
gates = L.workspace(lib.vts_lstm_gates_bytes(B, T, H), dev)
cells = L.workspace(lib.vts_lstm_cells_bytes(B, T, H), dev)
hp = L.workspace(lib.vts_lstm_planes_bytes(B, T, H, 1), dev)
h_last = torch.empty(B, H, device=dev)
w_ih = (torch.rand(4 * H, Cin, device=dev) * 2 - 1) / 16
w_hh = (torch.rand(4 * H, H, device=dev) * 2 - 1) / 16
b = torch.zeros(4 * H, device=dev)
wsb = lib.vts_lstm_fused_fwd_workspace_bytes(B, T, Cin, H)
ws = L.workspace(wsb, dev)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for it in range(4):
    if it == 1:
        ev[0].record()
    L.check(lib.vts_lstm_fused_fwd_planes(L.wptr(xp), 0 if first else 1, L.ptr(w_ih), L.ptr(b), L.ptr(b), L.ptr(w_hh), L.wptr(gates),
                                          L.wptr(cells), None, L.wptr(hp), L.ptr(h_last), None, L.wptr(ws), wsb, B, T, Cin, H, 1,
                                          L.stream()), "fused")
ev[1].record()
torch.cuda.synchronize()
print(f"Cin {Cin}: {ev[0].elapsed_time(ev[1]) / 3:.3f} ms per launch; h_last finite: {bool(torch.isfinite(h_last).all())}")
raw = ws[wsb - 256:].cpu()
print("status word", raw[:4].view(torch.int32).item())
if L.LIB_PATH == stamped:
    st = raw[8:136].view(torch.int64).tolist()
    names = {0: "land X0", 12: "next block's loads, bulk stores of the step before, acc = bias", 14: "poll, issue (X blocks)", 1: "compute X0",
             2: "land X1", 3: "compute X1", 4: "land X2", 5: "compute X2", 6: "land H0", 7: "compute H0 (+ issue H1)",
             8: "land H1", 9: "compute H1 (+ issue next X0)", 10: "activations, cell update, staging", 11: "publish, drain, barrier, flag",
             13: "optional outputs, loop top"}
    tot = sum(st)
    for i in (13, 0, 12, 14, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11):
        print(f"{names[i]:66s} {st[i] / T:9.0f} ticks/step  {100 * st[i] / tot:5.1f} %")
    print("total ticks/step", tot / T)

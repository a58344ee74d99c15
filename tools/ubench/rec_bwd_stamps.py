"""Development aid: per-phase tick counts of the staged backward recurrence, mean over the workgroups' wave 0
(needs tools/ubench/librec_xs.so, built by `python tools/ubench/rec_bwd_variants.py s`)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vae-time-series_amd"))
import torch
from vts import _lib as L
L.LIB_PATH = os.path.join(ROOT, "tools", "ubench", "librec_xs.so")
lib = L.load()
dev = torch.device("cuda:0")
B, T, H = 4096, 96, 256
planes = len(sys.argv) > 1 and sys.argv[1] == "planes"
gates = L.workspace(lib.vts_lstm_gates_bytes(B, T, H), dev); gates.view(torch.float32).uniform_(0.1, 0.9)
cells = L.workspace(lib.vts_lstm_cells_bytes(B, T, H), dev); cells.view(torch.float32).normal_()
dh_last = torch.randn(B, H, device=dev)
w_hh = (torch.rand(4 * H, H, device=dev) * 2 - 1) / 16
db = torch.empty(4 * H, device=dev)
dgp = L.workspace(lib.vts_lstm_planes_bytes(B, T, 4 * H, 0), dev)
wsb = lib.vts_lstm_recurrent_workspace_bytes(B, T, H); ws = L.workspace(wsb, dev)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for it in range(4):
    if it == 3: ev[0].record()
    if planes:
        L.check(lib.vts_lstm_recurrent_bwd_planes(L.wptr(gates), L.wptr(cells), None, L.ptr(dh_last), None, L.ptr(w_hh),
                                                  L.wptr(dgp), L.ptr(db), None, 0, L.wptr(ws), wsb, B, T, H, 1, L.stream()), "bwd")
    else:
        gates.view(torch.float32).uniform_(0.1, 0.9)
        if it == 3: ev[0].record()
        L.check(lib.vts_lstm_recurrent_bwd(L.wptr(gates), L.wptr(cells), None, L.ptr(dh_last), None, L.ptr(w_hh), L.ptr(db),
                                           None, 0, L.wptr(ws), wsb, B, T, H, 1, L.stream()), "bwd")
ev[1].record()
torch.cuda.synchronize()
print("launch (with packing): %.3f ms" % ev[0].elapsed_time(ev[1]))
st = (db[:8] / (B // 16)).tolist()
names = ["barrier at the top", "k-steps 0-3", "k-steps 4-7 (the batch must be back)", "k-steps 8-31", "take-over + batch issue",
         "barrier before the cell phase", "cell backward", "-"]
tot = sum(st)
for n, v in zip(names, st):
    print(f"{n:40s} {v:9.0f} ticks/step  {100 * v / max(tot, 1):5.1f} %")
print("total ticks/step", tot)

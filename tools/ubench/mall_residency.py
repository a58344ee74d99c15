"""Would the forward recurrence run faster on gate pre-activations that are still in the Infinity Cache?

The input projection writes G_x (1.6 GB per layer at 4096 x 96) and the forward recurrence reads it back: both through
HBM, because 1.6 GB is far beyond the 256 MB cache.  Cut into chunks of eight steps (134 MB) with the projection of a
chunk right in front of its recurrence, the read could come out of the cache.  This script measures what that is worth
before anything is built: projection + recurrence over T = 8 steps, the recurrence timed with its input hot (straight
after the projection) and cold (3 GB of other memory written in between), against the per-step time at T = 96.
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vae-time-series_amd"))
import torch
from vts import _lib as L
lib = L.load()
dev = torch.device("cuda:0")
B, H, C = 4096, 256, 256
torch.manual_seed(0)
w_ih = (torch.rand(4 * H, C, device=dev) * 2 - 1) / 16
w_hh = (torch.rand(4 * H, H, device=dev) * 2 - 1) / 16
b = torch.zeros(4 * H, device=dev)
flush = torch.empty(3 * 2**30 // 4, device=dev)


def run(T, cold, reps=10):
    x = torch.randn(B, T, C, device=dev)
    xp = L.workspace(lib.vts_lstm_planes_bytes(B, T, C, 0), dev)
    L.check(lib.vts_lstm_planes_from_rows(L.ptr(x), L.wptr(xp), B, T, C, H, 1, L.stream()), "from_rows")
    gates = L.workspace(lib.vts_lstm_gates_bytes(B, T, H), dev)
    cells = L.workspace(lib.vts_lstm_cells_bytes(B, T, H), dev)
    hp = L.workspace(lib.vts_lstm_planes_bytes(B, T, H, 1), dev)
    h_last = torch.empty(B, H, device=dev)
    wsb = lib.vts_lstm_recurrent_workspace_bytes(B, T, H); ws = L.workspace(wsb, dev)
    pwsb = lib.vts_lstm_planes_workspace_bytes(B, T, C, H); pws = L.workspace(pwsb, dev)
    t_in = t_rec = 0.0
    for it in range(reps + 2):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        e[0].record()
        L.check(lib.vts_lstm_inproj_fwd_planes(L.wptr(xp), 0, L.ptr(w_ih), L.ptr(b), L.ptr(b), L.wptr(gates), L.wptr(pws), pwsb,
                                               B, T, C, H, 1, L.stream()), "inproj")
        e[1].record()
        if cold:
            flush.fill_(1.0)
        e[2].record()
        L.check(lib.vts_lstm_recurrent_fwd_planes(L.wptr(gates), L.wptr(cells), None, L.wptr(hp), L.ptr(h_last), None, L.ptr(w_hh),
                                                  L.wptr(ws), wsb, B, T, H, 1, L.stream()), "rec")
        e[3].record()
        torch.cuda.synchronize()
        if it >= 2:
            t_in += e[0].elapsed_time(e[1]); t_rec += e[2].elapsed_time(e[3])
    print(f"T {T:3d} {'cold' if cold else 'hot '}: projection {t_in / reps:7.3f} ms ({1e3 * t_in / reps / T:6.2f} us/step)   "
          f"recurrence {t_rec / reps:7.3f} ms ({1e3 * t_rec / reps / T:6.2f} us/step)", flush=True)


for T in (8, 16, 96):
    for cold in (False, True):
        run(T, cold)

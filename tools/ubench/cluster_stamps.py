"""Development aid: per-phase cycle counts of the cluster forward recurrence (needs tools/ubench/libvts_cstamp.so,
built from lstm_rec_cluster.hip with -DVTS_C_STAMP)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vae-time-series_amd"))
import torch
from vts import _lib as L
L.LIB_PATH = os.path.join(ROOT, "tools", "ubench", os.environ.get("CSTAMP_LIB", "libvts_cstamp.so"))
lib = L.load()
dev = torch.device("cuda:0")
B, T, H = 4096, 96, 256
gates = L.workspace(lib.vts_lstm_gates_bytes(B, T, H), dev); gates.view(torch.float32).normal_()
cells = L.workspace(lib.vts_lstm_cells_bytes(B, T, H), dev)
hseq = torch.empty(B, T, H, device=dev); h_last = torch.empty(B, H, device=dev)
w_hh = (torch.rand(4 * H, H, device=dev) * 2 - 1) / 16
wsb = lib.vts_lstm_recurrent_workspace_bytes(B, T, H); ws = L.workspace(wsb, dev)
planes = len(sys.argv) > 1 and sys.argv[1] == "planes"      # the path the models take: h sequence as planes only
hp = L.workspace(lib.vts_lstm_planes_bytes(B, T, H, 1), dev)
for it in range(3):
    if planes:
        L.check(lib.vts_lstm_recurrent_fwd_planes(L.wptr(gates), L.wptr(cells), None, L.wptr(hp), L.ptr(h_last), None, L.ptr(w_hh),
                                                  L.wptr(ws), wsb, B, T, H, 1, L.stream()), "fwd")
    else:
        L.check(lib.vts_lstm_recurrent_fwd(L.wptr(gates), L.wptr(cells), L.ptr(hseq), L.ptr(h_last), None, L.ptr(w_hh), L.wptr(ws), wsb, B, T, H, 1, L.stream()), "fwd")
torch.cuda.synchronize()
r256 = lambda n: (n + 255) // 256 * 256
# the launch's status word (behind the packed image, the bias partials and the two final-state seeds); the stamps follow it
off = r256(4 * H * H * 4) + r256(B // 16 * 4 * H * 4) + 2 * r256(B * H * 4)
raw = ws[off:off + 128].cpu()
print("err word", raw[:4].view(torch.int32).item())
st = raw[8:72].view(torch.int64).tolist()
names = ["poll + barrier", "issue h loads", "product", "cell update", "publish + drain + barrier + flag", "issue bulk stores", "loop top", "acc = prefetched gates (wait)"]
tot = sum(st)
for n, v in zip(names, st):
    print(f"{n:36s} {v / T:9.0f} ticks/step  {100 * v / tot:5.1f} %")
print("total ticks/step", tot / T)

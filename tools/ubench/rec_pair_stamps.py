"""Development aid: per-phase tick counts of the pair backward recurrence (lstm_rec_pair.hip), mean over the
workgroups' wave 0.  `python tools/ubench/rec_pair_stamps.py build` (no GPU needed) builds tools/ubench/librec_ps.so
with the stamps compiled in (-DVTS_RP_STAMP); without arguments it runs that library on the GPU."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "vae-time-series_amd", "csrc")
LIB = os.environ.get("VTS_PS_LIB", os.path.join(ROOT, "tools", "ubench", "librec_ps.so"))
if sys.argv[1:2] == ["build"]:
    os.makedirs("/tmp/recx", exist_ok=True)
    o = "/tmp/recx/pair_stamp.o"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function",
                           "-Wno-inline-asm", "-DVTS_RP_STAMP"] + sys.argv[2:] + ["-c", "lstm_rec_pair.hip", "-o", o], cwd=CSRC)
    objs = [f for f in os.listdir(CSRC) if f.endswith(".o") and f != "lstm_rec_pair.o"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, o] + objs, cwd=CSRC)
    print("built", LIB)
    sys.exit(0)
sys.path.insert(0, os.path.join(ROOT, "vae-time-series_amd"))
import torch
from vts import _lib as L
L.LIB_PATH = LIB
lib = L.load()
dev = torch.device("cuda:0")
B, T, H = 4096, 96, 256
with_dh = "dh" in sys.argv[1:]
gates = L.workspace(lib.vts_lstm_gates_bytes(B, T, H), dev); gates.view(torch.float32).uniform_(0.1, 0.9)
cells = L.workspace(lib.vts_lstm_cells_bytes(B, T, H), dev); cells.view(torch.float32).normal_()
dh_last = torch.randn(B, H, device=dev)
dh_seq = torch.randn(B * T * H, device=dev) if with_dh else None
w_hh = (torch.rand(4 * H, H, device=dev) * 2 - 1) / 16
db = torch.empty(4 * H, device=dev)
dgp = L.workspace(lib.vts_lstm_planes_bytes(B, T, 4 * H, 0), dev)
wsb = lib.vts_lstm_recurrent_workspace_bytes(B, T, H); ws = L.workspace(wsb, dev)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for it in range(4):
    if it == 3: ev[0].record()
    L.check(lib.vts_lstm_recurrent_bwd_planes(L.wptr(gates), L.wptr(cells), L.ptr(dh_seq) if with_dh else None, L.ptr(dh_last), None,
                                              L.ptr(w_hh), L.wptr(dgp), L.ptr(db), None, 0, L.wptr(ws), wsb, B, T, H, 1, L.stream()), "bwd")
ev[1].record()
torch.cuda.synchronize()
print("launch (with packing, exchange reset, bias sums): %.3f ms" % ev[0].elapsed_time(ev[1]))
names = ["barrier at the top", "units 0-3", "resident units 4, 5", "units 6, 7 + publish", "units 8-11", "units 12-15",
         "wait for the peer's half", "batch issue", "barrier before the cell phase", "cell backward",
         "(reads of the slot that found it EMPTY, per step)", "sums + slot reset", "late coefficients + dh take-over", "-", "-", "-"]
for m in (0, 1):
    st = (db[128 * m:128 * m + 16] / (B // 32)).tolist()
    tot = sum(st[:10]) + st[11] + st[12]
    print("member", m)
    for n, v in zip(names, st):
        print(f"  {n:50s} {v:9.1f} ticks/step  {100 * v / max(tot, 1):5.1f} %")
    print("  total ticks/step", tot)

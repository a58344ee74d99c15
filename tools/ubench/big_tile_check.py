"""Development aid: the three LSTM GEMM stages at the bench shape in both arithmetic modes - agreement of
the outputs (the bf16x3 launches take the 256 x 128 tile kernel at this size) and time per launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vae-time-series_amd"))
import torch
from vts import _lib as L
if os.environ.get("VTS_LIB"): L.LIB_PATH = os.path.abspath(os.environ["VTS_LIB"])
lib = L.load()
dev = torch.device("cuda:0")
B, T, H = int(os.environ.get("B", 4096)), 96, 256
st = L.stream()
torch.manual_seed(0)
for Cin in (370, 256):
    x = torch.randn(B, T, Cin, device=dev)
    w = torch.randn(4 * H, Cin, device=dev) / 16
    b0 = torch.randn(4 * H, device=dev); b1 = torch.randn(4 * H, device=dev)
    gb = lib.vts_lstm_gates_bytes(B, T, H)
    dgates = L.workspace(gb, dev); dgates.view(torch.float32).normal_()
    wb = lib.vts_lstm_wgrad_workspace_bytes(B, T, Cin, H); ws = L.workspace(wb, dev)
    res = {}
    for mode in (0, 1):
        g = L.workspace(gb, dev)
        dx = L.workspace(lib.vts_lstm_seqgrad_bytes(B, T, Cin if Cin % 16 == 0 else H), dev)
        dw = torch.empty(4 * H, Cin, device=dev)
        dwh = torch.empty(4 * H, Cin, device=dev)
        stages = {"inproj": lambda: lib.vts_lstm_inproj_fwd(L.ptr(x), L.ptr(w), L.ptr(b0), L.ptr(b1), L.wptr(g), B, T, Cin, H, mode, st),
                  "wgrad0": lambda: lib.vts_lstm_wgrad(L.wptr(dgates), L.ptr(x), 0, L.ptr(dw), 0, L.wptr(ws), wb, B, T, Cin, H, mode, st),
                  "wgrad-1": lambda: lib.vts_lstm_wgrad(L.wptr(dgates), L.ptr(x), -1, L.ptr(dwh), 0, L.wptr(ws), wb, B, T, Cin, H, mode, st)}
        if Cin % 16 == 0:
            stages["dgrad"] = lambda: lib.vts_lstm_dgrad(L.wptr(dgates), L.ptr(w), L.wptr(dx), B, T, Cin, H, mode, st)
        for nm, fn in stages.items():
            L.check(fn(), nm); torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5): L.check(fn(), nm)
            e.record(); torch.cuda.synchronize()
            print(f"C={Cin} mode={mode} {nm:8s} {s.elapsed_time(e) / 5:7.3f} ms", flush=True)
        res[mode] = dict(g=g.view(torch.float32).clone(), dx=dx.view(torch.float32).clone(), dw=dw.clone(), dwh=dwh.clone())
    for k in res[0]:
        a, b = res[0][k].double(), res[1][k].double()
        print(f"C={Cin} {k}: L2-rel(mode1 vs mode0) {((a - b).norm() / a.norm().clamp_min(1e-30)).item():.2e}  max|d| {(a - b).abs().max().item():.2e}", flush=True)

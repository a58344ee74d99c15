"""Development aid: h_n of one vts.LSTM layer against torch's CPU nn.LSTM, both arithmetic modes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vae-time-series_amd"))
import torch
import vts
B, T, C, H = int(os.environ.get("FS_B", 4096)), 96, int(os.environ.get("FS_C", 370)), 256
torch.manual_seed(0)
ref = torch.nn.LSTM(C, H, 1, batch_first=True)
x = torch.randn(B, T, C)
with torch.no_grad():
    out_ref, (hn_ref, cn_ref) = ref(x)
out64, (hn64, _) = ref.double()(x.double())
ref.float()
dev = torch.device("cuda:0")
def rel(a, b): return ((a.double() - b.double()).norm() / b.double().norm()).item()
print("cpu f32 vs f64:", rel(hn_ref, hn64.detach()))
for mode in ("f32", "bf16x3"):
    vts.set_mode(mode)
    m = vts.LSTM(C, H, 1, batch_first=True).to(dev)
    m.load_state_dict({k: v.to(dev) for k, v in ref.state_dict().items()})
    with torch.no_grad():
        out, (hn, cn) = m(x.to(dev))
    torch.cuda.synchronize()
    print(mode, "h_n vs f64:", rel(hn.cpu(), hn64.detach()), "c_n:", rel(cn.cpu(), cn_ref),
          "out:", None if out is None else rel(out.cpu(), out_ref), "max abs h_n err", (hn.cpu() - hn_ref).abs().max().item())

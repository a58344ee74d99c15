"""Development aid: runs N eager train steps of the mixture model at the reference's own sizes (16 windows x 16 steps
x 1 channel, hidden 64); under `rocprofv3 --kernel-trace --stats` this lists the kernels of a step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vae-time-series_amd"))
import torch, vts
dev = torch.device("cuda:0")
vts.set_mode(os.environ.get("VTS_MODE", "f32"))
B, T, C, H, L = 16, 16, 1, 64, 32
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
torch.manual_seed(0)
model = vts.build_model("mixture", C, H, L, 1).to(dev); model.train()
opt = vts.Adam(vts.FlatParameters(model), lr=1e-3)
x, y = torch.randn(B, T, C, device=dev), torch.randn(B, 1, device=dev)
for _ in range(N): vts.train_step(model, opt, x, y)
torch.cuda.synchronize()

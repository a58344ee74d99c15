"""Development aid: per-parameter gradient gap between the HIP step and the CPU oracle at the bench shape, in
both arithmetic modes, beside the gap between two CPU evaluations (fp32 oracle vs the same in fp64)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vae-time-series_amd"))
import torch
import vts
from oracle import ref_torch as R
kind = sys.argv[1] if len(sys.argv) > 1 else "mixture"
B, T, C, H, LD = int(os.environ.get("FS_B", 4096)), 96, 370, 256, 32
sd = R.init_state(kind, C, H, LD, 1, seed=0)
if kind in ("mixture", "gpd"):
    sd["encoder.shape_layer_extreme.weight"] *= 0.25
    sd["encoder.shape_layer_extreme.bias"].fill_(0.5)
SEED = int(os.environ.get('FS_SEED', 11))
g = torch.Generator().manual_seed(SEED)
x = torch.randn(B, T, C, generator=g); y = torch.randn(B, 1, generator=g)
noise = R.draw_noise(kind, B, LD, seed=SEED + 1)
loss_ref, _, gref = R.loss_and_grads(kind, sd, x, y, noise, threshold=0.8)
sd64 = {k: v.double() for k, v in sd.items()}
n64 = {k: v.double() for k, v in noise.items()}
try:
    loss64, _, g64 = R.loss_and_grads(kind, sd64, x.double(), y.double(), n64, threshold=0.8)
except Exception as e:
    print("fp64 oracle failed:", e); g64 = None
dev = torch.device("cuda:0")
res = {}
for mode in ("f32", "bf16x3"):
    vts.set_mode(mode)
    m = vts.build_model(kind, C, H, LD, 1, threshold=0.8).to(dev)
    m.load_state_dict({k: v.to(dev) for k, v in sd.items()})
    m.train()
    loss = vts.train_step(m, None, x.to(dev), y.to(dev), {k: v.to(dev) for k, v in noise.items()})
    torch.cuda.synchronize()
    res[mode] = (loss.item(), {k: p.grad.detach().cpu() for k, p in m.named_parameters()})
print("loss ref", loss_ref.item(), "f32", res["f32"][0], "bf16x3", res["bf16x3"][0], "fp64", None if g64 is None else loss64.item())
def rel(a, b): return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
print(f"{'parameter':44s} {'|g|':>10s} {'f32-vs-ref':>11s} {'x3-vs-ref':>11s} {'ref-vs-f64':>11s} {'f32-vs-f64':>11s} {'x3-vs-f64':>11s}")
for k in gref:
    r64 = g64[k] if g64 is not None else gref[k]
    print(f"{k:44s} {gref[k].norm().item():10.3e} {rel(res['f32'][1][k], gref[k]):11.2e} {rel(res['bf16x3'][1][k], gref[k]):11.2e} "
          f"{rel(gref[k], r64):11.2e} {rel(res['f32'][1][k], r64):11.2e} {rel(res['bf16x3'][1][k], r64):11.2e}")

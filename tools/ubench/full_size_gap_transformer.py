"""Development aid: per-parameter gradient gap between the HIP transformer step and the CPU oracle at (a slice
of) the bench shape, both arithmetic modes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vae-time-series_amd"))
import torch
import vts
from oracle import ref_transformer as RT
B, T, C, d, nhead, layers, L = int(os.environ.get("FS_B", 1024)), 96, 370, 64, 8, 2, 32
SEED = int(os.environ.get("FS_SEED", 3))
sd = RT.init_state(C, d, layers, L, 1, seed=SEED)
sd["encoder.shape_layer_extreme.weight"] *= 0.25
sd["encoder.shape_layer_extreme.bias"].fill_(0.5)
g = torch.Generator().manual_seed(SEED + 1)
x = torch.randn(B, T, C, generator=g); y = torch.randn(B, 1, generator=g)
thr = float(y.abs().median())
noise = RT.draw_noise(B, T, d, nhead, layers, L, seed=SEED + 2)
t0 = time.time()
loss_ref, _, gref = RT.loss_and_grads(sd, x, y, noise, nhead, thr, beta=1.0)
print("oracle", round(time.time() - t0, 1), "s", flush=True)
dev = torch.device("cuda:0")
def rel(a, b): return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
res = {}
for mode in ("f32", "bf16x3"):
    vts.set_mode(mode)
    m = vts.TransformerMixtureVAE(C, d, layers, nhead, L, 1, beta=1.0).to(dev)
    m.load_state_dict({k: v.to(dev) for k, v in sd.items()}, strict=True)
    m.train()
    outs = m(x.to(dev), {k: v.to(dev) for k, v in noise.items()})
    loss = m.loss_function(outs[0], y.to(dev), *outs[1:], thr)
    loss.backward()
    torch.cuda.synchronize()
    res[mode] = (loss.item(), {k: (p.grad.detach().cpu() if p.grad is not None else torch.zeros_like(p).cpu()) for k, p in m.named_parameters()})
print("loss ref", loss_ref.item(), "f32", res["f32"][0], "bf16x3", res["bf16x3"][0])
for k in gref:
    if gref[k].abs().max() == 0: continue
    print(f"{k:52s} {gref[k].norm().item():10.3e} {rel(res['f32'][1][k], gref[k]):11.2e} {rel(res['bf16x3'][1][k], gref[k]):11.2e}")

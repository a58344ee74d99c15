"""Gradient difference between the two arithmetic modes of the transformer model at the bench shape
(same weights, inputs, sampling noise and dropout seeds).  The modes differ by ~1e-6 in every
product; a relu input that close to 0 lands on either side, and each such element moves a gradient
by a whole term, so this gap - not 1e-6 - is what two correct fp32-level implementations show."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vae-time-series_amd")]
import torch
import vts
import bench

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
x, y = bench.synth_windows(B, 96, 370, 0, dev)
thr = float(y.abs().median())
out = {}
for mode in ("f32", "bf16x3"):
    vts.set_mode(mode)
    torch.manual_seed(1234)
    model = vts.TransformerMixtureVAE(370, 64, 2, 8, 32, 1).to(dev)
    with torch.no_grad():   # GPD shapes away from 0 (the sampler's own fp32 cancellation is a separate matter)
        model.encoder.shape_layer_extreme.weight.mul_(0.25)
        model.encoder.shape_layer_extreme.bias.fill_(0.5)
    model.train()
    model.threshold = thr
    torch.manual_seed(99)
    loss = vts.train_step(model, None, x, y)
    torch.cuda.synchronize()
    out[mode] = (loss.item(), {k: p.grad.double().cpu() for k, p in model.named_parameters() if p.grad is not None})
la, ga = out["f32"]; lb, gb = out["bf16x3"]
print(f"B={B} loss f32 {la:.8f} bf16x3 {lb:.8f} rel {abs(la - lb) / abs(la):.2e}")
worst = 0
for k in ga:
    rel = ((ga[k] - gb[k]).norm() / ga[k].norm().clamp_min(1e-30)).item()
    worst = max(worst, rel)
    if rel > 2e-5:
        print(f"  {k:70s} {rel:.2e}")
print("worst", f"{worst:.2e}")

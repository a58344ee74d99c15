"""Development aid: ms per train step at the reference's own sizes (16 windows x 16 steps x 1 channel, hidden 64),
launched one by one and replayed as a HIP graph.  VTS_MODE=bf16x3 selects the other arithmetic of the LSTM products."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vae-time-series_amd"))
import torch, vts
if os.environ.get("VTS_LIB"):
    from vts import _lib as _L
    _L.LIB_PATH = os.path.abspath(os.environ["VTS_LIB"])
dev = torch.device("cuda:0")
vts.set_mode(os.environ.get("VTS_MODE", "f32"))
B, T, C, H, L = [int(v) for v in (sys.argv[1:6] + [16, 16, 1, 64, 32][len(sys.argv) - 1:])]
for kind in ("gaussian", "gpd", "mixture", "mixture3"):
    for graphed in (False, True):
        torch.manual_seed(0)
        model = vts.build_model(kind, C, H, L, 1).to(dev); model.train()
        if kind == "mixture3": model.threshold = 0.8
        opt = vts.Adam(vts.FlatParameters(model), lr=1e-3)
        x, y = torch.randn(B, T, C, device=dev), torch.randn(B, 1, device=dev)
        step = vts.GraphedTrainStep(model, opt, x, y) if graphed else (lambda a, b: vts.train_step(model, opt, a, b))
        for _ in range(20): step(x, y)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 300
        for _ in range(n): step(x, y)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / n * 1e3
        print(f"{kind:9s} graph={graphed}: {ms:.3f} ms/step  {B / ms * 1e3:.0f} windows/s", flush=True)

#!/usr/bin/env python3
"""Generate tests/golden/mixture3_*.npz by running the REFERENCE's three-component VAE on CPU.

Usage (build container only; the reference tree does not travel to the GPU box):
    python tests/golden/make_goldens_mixture3.py /root/reference

vae-models/graph/vae_mixture_graph_model.py imports torch_geometric, which is absent here, at its top.
The model is used the way graph/train_ld_2011.py:91 builds it (use_gcn=False, use_d_knn=False), and on
that path nothing of torch_geometric is ever looked up, so the module's definitions are compiled out of
its parsed source with the two torch_geometric import statements left out; every class and function
that runs is the reference's own.  Nothing of the reference's text is stored: the fixtures hold tensors
only.  Random draws are tapped as in make_goldens.py, plus torch.bernoulli.
"""
from __future__ import annotations

import ast
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_goldens import NoiseTap, pack  # noqa: E402

REL = "vae-models/graph/vae_mixture_graph_model.py"
SHAPES = {"tiny": (8, 16, 1, 16, 8), "native": (16, 16, 1, 64, 32), "wide": (12, 24, 37, 64, 32)}


def load_reference(root):
    path = os.path.join(root, REL)
    with open(path, "r") as fh:
        tree = ast.parse(fh.read(), path)

    def wanted(node):
        if isinstance(node, ast.Import):
            return not any(a.name.startswith("torch_geometric") for a in node.names)
        if isinstance(node, ast.ImportFrom):
            return not (node.module or "").startswith("torch_geometric")
        return True

    ns = {"__name__": "reference_mixture3"}
    exec(compile(ast.Module(body=[n for n in tree.body if wanted(n)], type_ignores=[]), path, "exec"), ns)
    return ns


def case(root, tag, shape_name, threshold, seed, y_edit=None, tweak=None, train=True):
    B, T, C, H, L = SHAPES[shape_name]
    ns = load_reference(root)
    torch.manual_seed(seed)
    model = ns["VAE"](input_dim=C, lstm_output_dim=H, gcn_output_dim=0, latent_dim=L, future_steps=1,
                      use_d_knn=False, use_gcn=False, use_gpd=True, use_bernoulli=True)   # graph/train_ld_2011.py:91
    if tweak:
        tweak(model)
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(B, T, C, generator=g) * 0.9 + 0.18
    y = torch.randn(B, 1, generator=g) * 0.9 + 0.6
    if y_edit:
        y_edit(y)
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.train(train)
    model.zero_grad()
    bern = []
    real_bernoulli = torch.bernoulli

    def bernoulli(*a, **k):
        out = real_bernoulli(*a, **k)
        bern.append(out.detach().clone())
        return out

    torch.bernoulli = bernoulli
    try:
        with NoiseTap(model) as tap:
            outs = model(x, x, None)                                                        # :108
            loss = model.loss_function(outs[0], y, outs[1], outs[2], threshold, outs[3], outs[4], outs[5])   # :110
    finally:
        torch.bernoulli = real_bernoulli
    if torch.isfinite(loss):
        loss.backward()
    noise = {"eps": tap.randn[0], "u": tap.rand[0], "choice": tap.rand[1], "bern": bern[0]}
    if "encoder.dropout" in tap.keep:
        noise["enc_keep"] = tap.keep["encoder.dropout"]
    if "decoder.dropout" in tap.keep:
        noise["dec_keep"] = tap.keep["decoder.dropout"]
    grads = {k: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for k, p in model.named_parameters()}
    names = ["recon", "mean", "logvar", "scale", "shape", "logits"]
    blob = {"meta/dims": np.array([B, T, C, H, L, 1], dtype=np.int64), "meta/threshold": np.float64(threshold),
            "meta/beta": np.float64(model.beta), "meta/train": np.int64(train), "x": x.numpy(), "y": y.numpy(),
            "loss": loss.detach().numpy()}
    blob.update(pack("sd", sd0)); blob.update(pack("noise", noise))
    blob.update(pack("out", dict(zip(names, outs)))); blob.update(pack("grad", grads))
    np.savez_compressed(os.path.join(HERE, f"mixture3_{tag}.npz"), **blob)
    p = torch.softmax(model.pi_params.detach(), 0)
    ch = noise["choice"]
    picks = [int((ch < p[0]).sum()), int(((ch >= p[0]) & (ch < p[0] + p[1])).sum()), int((ch >= p[0] + p[1]).sum())]
    rows = [int(((y <= threshold) & (y != 0)).sum()), int(((y > threshold) & (y != 0)).sum()), int((y == 0).sum())]
    print(f"mixture3_{tag}: loss={float(loss):.9g} picks(g,p,b)={picks} rows(normal,extreme,zero)={rows}")


def main():
    if len(sys.argv) < 2 or not os.path.isdir(sys.argv[1]):
        print("reference tree not given / not present: nothing to do")
        return
    root = sys.argv[1]
    torch.set_num_threads(1)

    def zeros(*rows):
        def edit(y):
            for r in rows:
                y[r] = 0.0
        return edit

    def well_conditioned(model):     # GPD shapes comfortably above the 1e-6 clamp
        with torch.no_grad():
            model.encoder.shape_layer_extreme.weight.mul_(0.25)
            model.encoder.shape_layer_extreme.bias.fill_(0.5)

    def spread_pi(model):            # all three samples get picked
        well_conditioned(model)
        with torch.no_grad():
            model.pi_params.copy_(torch.tensor([0.6, 0.3, 0.2]))

    case(root, "tiny", "tiny", 0.9, 41, y_edit=zeros(1, 5), tweak=spread_pi)
    case(root, "native", "native", 0.9, 42, y_edit=zeros(0, 7, 11), tweak=spread_pi)
    case(root, "native_defaultpi", "native", 0.9, 43, y_edit=zeros(3), tweak=well_conditioned)
    case(root, "wide_eval", "wide", 0.9, 44, y_edit=zeros(2, 9), tweak=spread_pi, train=False)
    case(root, "native_clamped", "native", 0.9, 45, y_edit=zeros(4))            # default init: shapes at the clamp
    case(root, "tiny_gaussian_only", "tiny", 95.0, 46, tweak=spread_pi)          # no extreme and no zero rows
    case(root, "tiny_no_normal", "tiny", -50.0, 47, y_edit=zeros(0), tweak=spread_pi)   # empty normal class: NaN


if __name__ == "__main__":
    main()

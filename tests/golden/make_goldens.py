#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE implementation on CPU.

Usage (build container only; the reference tree does not travel to the GPU box):
    python tests/golden/make_goldens.py /root/reference

The three reference scripts train at import time, so only the source text ahead of their
data-loading marker (pure class / function definitions) is executed, with the module
global ``threshold`` injected.  Nothing of the reference's text is stored: the fixtures
hold tensors only (inputs, the random draws the reference made, outputs, loss, parameter
gradients, Adam trajectories).

Random draws are captured by temporarily wrapping torch.rand / torch.randn_like and by
forward hooks on the two nn.Dropout modules (keep-mask = (out != 0) | (in == 0); where
the post-ReLU input is 0 the mask value affects neither forward nor backward).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))

SCRIPTS = {
    "gaussian": ("vae-models/gaussian-lstm-DE.py", "# Function to parse the data"),
    "gpd": ("vae-models/gpd-lstm-DE.py", "# Load data"),
    "mixture": ("vae-models/mixture-vae-DE.py", "# Load data"),
}

SHAPES = {  # name: (B, T, C, H, L)
    "tiny": (8, 16, 1, 16, 8),
    "native": (16, 16, 1, 64, 32),
    "wide": (8, 24, 37, 64, 32),
}


def load_reference(root: str, kind: str, threshold: float):
    rel, marker = SCRIPTS[kind]
    with open(os.path.join(root, rel), "r") as fh:
        text = fh.read()
    cut = text.index(marker)
    ns = {"__name__": f"reference_{kind}", "threshold": threshold}
    exec(compile(text[:cut], rel, "exec"), ns)
    return ns


class NoiseTap:
    """Records every torch.rand / torch.randn_like result and dropout keep-mask."""

    def __init__(self, model):
        self.rand, self.randn, self.keep = [], [], {}
        self._model = model
        self._hooks = []

    def __enter__(self):
        self._rand, self._randn_like = torch.rand, torch.randn_like

        def rand(*a, **k):
            out = self._rand(*a, **k)
            self.rand.append(out.detach().clone())
            return out

        def randn_like(*a, **k):
            out = self._randn_like(*a, **k)
            self.randn.append(out.detach().clone())
            return out

        torch.rand, torch.randn_like = rand, randn_like
        for name, mod in self._model.named_modules():
            if isinstance(mod, torch.nn.Dropout):
                def hook(m, inp, out, name=name):
                    if m.training:
                        self.keep[name] = ((out != 0) | (inp[0] == 0)).float().detach()
                self._hooks.append(mod.register_forward_hook(hook))
        return self

    def __exit__(self, *exc):
        torch.rand, torch.randn_like = self._rand, self._randn_like
        for h in self._hooks:
            h.remove()


def noise_dict(kind: str, tap: NoiseTap):
    n = {}
    if kind == "gaussian":
        n["eps"] = tap.randn[0]
    elif kind == "gpd":
        n["u"] = tap.rand[0]
    else:
        n["choice"], n["u"] = tap.rand[0], tap.rand[1]
        n["eps"] = tap.randn[0]
    if "encoder.dropout" in tap.keep:
        n["enc_keep"] = tap.keep["encoder.dropout"]
    if "decoder.dropout" in tap.keep:
        n["dec_keep"] = tap.keep["decoder.dropout"]
    return n


def synth_xy(B, T, C, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, T, C, generator=g) * 0.9 + 0.18
    y = torch.randn(B, 1, generator=g) * 0.9 + 0.6
    return x, y


def run_once(ns, kind, model, x, y, train=True):
    model.train(train)
    model.zero_grad()
    with NoiseTap(model) as tap:
        outs = model(x)
        loss = model.loss_function(outs[0], y, *outs[1:])
    loss.backward()
    names = {"gaussian": ["recon", "mean", "logvar"], "gpd": ["recon", "scale", "shape"],
             "mixture": ["recon", "mean", "logvar", "scale", "shape"]}[kind]
    return loss.detach(), dict(zip(names, [o.detach() for o in outs])), noise_dict(kind, tap)


def pack(prefix, d):
    return {f"{prefix}/{k}": v.detach().cpu().numpy().copy() for k, v in d.items()}


def single_case(root, kind, shape_name, threshold, seed, tweak=None, train=True, tag=None):
    B, T, C, H, L = SHAPES[shape_name]
    ns = load_reference(root, kind, threshold)
    torch.manual_seed(seed)
    model = ns["VAE"](input_dim=C, lstm_output_dim=H, latent_dim=L, future_steps=1)
    if tweak:
        tweak(model)
    x, y = synth_xy(B, T, C, seed + 1)
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    loss, outs, noise = run_once(ns, kind, model, x, y, train)
    grads = {k: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p))
             for k, p in model.named_parameters()}
    blob = {"meta/dims": np.array([B, T, C, H, L, 1], dtype=np.int64),
            "meta/threshold": np.float64(threshold), "meta/beta": np.float64(model.beta),
            "meta/train": np.int64(train),
            "x": x.numpy(), "y": y.numpy(), "loss": loss.numpy()}
    blob.update(pack("sd", sd0)); blob.update(pack("noise", noise))
    blob.update(pack("out", outs)); blob.update(pack("grad", grads))
    name = tag or f"{kind}_{shape_name}"
    np.savez(os.path.join(HERE, name + ".npz"), **blob)
    print(f"{name}: loss={float(loss):.9g}  params={sum(v.numel() for v in sd0.values())}")


def adam_trajectory(root, kind, lr, steps=5, seed=7, tweak=None, tag=None):
    """k optimiser steps at the reference-native shape (train-loop body:
    gaussian-lstm-DE.py:149-156, gpd-lstm-DE.py:99-105, mixture-vae-DE.py:158-164)."""
    B, T, C, H, L = SHAPES["native"]
    ns = load_reference(root, kind, 95.0)
    torch.manual_seed(seed)
    model = ns["VAE"](input_dim=C, lstm_output_dim=H, latent_dim=L, future_steps=1)
    if tweak is not None:
        tweak(model)
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    blob = {"meta/dims": np.array([B, T, C, H, L, 1], dtype=np.int64),
            "meta/threshold": np.float64(95.0), "meta/beta": np.float64(model.beta),
            "meta/lr": np.float64(lr), "meta/steps": np.int64(steps)}
    blob.update(pack("sd0", model.state_dict()))
    losses = []
    for s in range(steps):
        x, y = synth_xy(B, T, C, 100 * seed + s)
        opt.zero_grad()
        loss, _, noise = run_once(ns, kind, model, x, y, True)
        opt.step()
        losses.append(float(loss))
        blob[f"step{s}/x"], blob[f"step{s}/y"] = x.numpy(), y.numpy()
        blob.update(pack(f"step{s}/noise", noise))
    blob["losses"] = np.array(losses, dtype=np.float32)
    blob.update(pack("sdK", model.state_dict()))
    name = tag or f"adam_{kind}_native"
    np.savez(os.path.join(HERE, name + ".npz"), **blob)
    print(f"{name}: losses={losses}")


def adam_trajectory_real_data(root, steps=8, seed=41):
    """The reference's first configuration on its own data: the Gaussian model's train loop
    (gaussian-lstm-DE.py:100-156: the five tensors of the data split, DataLoader(batch_size=16, shuffle=True),
    Adam(lr=5e-3)) for the first `steps` batches of an epoch.  The script reads "data_split.pt" from its working
    directory; the committed file is datasets/deep-extrema-data-split.pt.  The loader's shuffle is seeded here, and
    the batches it produced are stored (windows of the data set: data, 16 x 16 x 1 floats each)."""
    from torch.utils.data import DataLoader, TensorDataset
    ns = load_reference(root, "gaussian", 95.0)
    split = torch.load(os.path.join(root, "datasets", "deep-extrema-data-split.pt"), map_location="cpu", weights_only=True)
    X_train, y_train = split[0].float(), split[1].float()
    B, T, C, H, L = 16, X_train.shape[1], X_train.shape[2], 64, 32        # gaussian-lstm-DE.py:108-114
    torch.manual_seed(seed)
    model = ns["VAE"](input_dim=C, lstm_output_dim=H, latent_dim=L, future_steps=1)
    opt = torch.optim.Adam(model.parameters(), lr=5e-3)
    loader = DataLoader(TensorDataset(X_train, y_train), batch_size=B, shuffle=True,
                        generator=torch.Generator().manual_seed(seed + 1))
    blob = {"meta/dims": np.array([B, T, C, H, L, 1], dtype=np.int64), "meta/threshold": np.float64(95.0),
            "meta/beta": np.float64(model.beta), "meta/lr": np.float64(5e-3), "meta/steps": np.int64(steps),
            "meta/train_rows": np.int64(X_train.shape[0])}
    blob.update(pack("sd0", model.state_dict()))
    losses = []
    for s, (x, y) in enumerate(loader):
        if s == steps:
            break
        opt.zero_grad()
        loss, _, noise = run_once(ns, "gaussian", model, x, y, True)
        opt.step()
        losses.append(float(loss))
        blob[f"step{s}/x"], blob[f"step{s}/y"] = x.numpy().copy(), y.numpy().copy()
        blob.update(pack(f"step{s}/noise", noise))
    blob["losses"] = np.array(losses, dtype=np.float32)
    blob.update(pack("sdK", model.state_dict()))
    np.savez(os.path.join(HERE, "adam_gaussian_real_data.npz"), **blob)
    print(f"adam_gaussian_real_data: X_train {tuple(X_train.shape)}, losses={losses}")


def main():
    if len(sys.argv) < 2 or not os.path.isdir(sys.argv[1]):
        print("reference tree not given / not present: nothing to do")
        return
    root = sys.argv[1]
    torch.set_num_threads(1)
    if len(sys.argv) > 2 and sys.argv[2] == "--only-wellcond-trajectory":   # adds that one fixture, leaves the rest
        def well_conditioned(model):
            with torch.no_grad():
                model.encoder.shape_layer_extreme.weight.mul_(0.25)
                model.encoder.shape_layer_extreme.bias.fill_(0.5)
        adam_trajectory(root, "gpd", 1e-3, tweak=well_conditioned, tag="adam_gpd_native_wellcond")
        return
    if len(sys.argv) > 2 and sys.argv[2] == "--only-real-data-trajectory":  # adds that one fixture, leaves the rest
        adam_trajectory_real_data(root)
        return
    for kind in SCRIPTS:
        for si, shape_name in enumerate(SHAPES):
            single_case(root, kind, shape_name, 95.0, seed=11 + 3 * si)
    # mixed normal/extreme mask (mixture-vae-DE.py:116-119 with a threshold inside the data)
    single_case(root, "mixture", "native", 0.6, seed=31, tag="mixture_native_mixedmask")
    # empty normal mask => masked MSE of an empty tensor => NaN loss
    single_case(root, "mixture", "tiny", -1.0, seed=32, tag="mixture_tiny_emptymask")
    # eval mode: dropout off, sampling still on (mixture-vae-DE.py:169-175)
    single_case(root, "mixture", "tiny", 95.0, seed=33, train=False, tag="mixture_tiny_eval")
    single_case(root, "gpd", "tiny", 95.0, seed=34, train=False, tag="gpd_tiny_eval")

    def clamp_all(model):  # every shape pre-activation far below 1e-6: clamp passes no gradient
        with torch.no_grad():
            model.encoder.shape_layer_extreme.bias.fill_(-5.0)
    single_case(root, "gpd", "tiny", 95.0, seed=35, tweak=clamp_all, tag="gpd_tiny_clamped")
    single_case(root, "mixture", "native", 95.0, seed=36, tweak=clamp_all,
                tag="mixture_native_clamped")

    def well_conditioned(model):  # every shape comfortably above the clamp: no cancellation noise
        with torch.no_grad():
            model.encoder.shape_layer_extreme.weight.mul_(0.25)
            model.encoder.shape_layer_extreme.bias.fill_(0.5)
    single_case(root, "gpd", "native", 95.0, seed=37, tweak=well_conditioned,
                tag="gpd_native_wellcond")
    single_case(root, "gpd", "wide", 95.0, seed=38, tweak=well_conditioned,
                tag="gpd_wide_wellcond")
    single_case(root, "mixture", "wide", 0.6, seed=39, tweak=well_conditioned,
                tag="mixture_wide_wellcond_mixedmask")
    adam_trajectory(root, "gaussian", 5e-3)
    adam_trajectory(root, "gpd", 1e-3)
    adam_trajectory(root, "mixture", 1e-3)
    # the GPD train loop (gpd-lstm-DE.py:99-105) from a state whose sampler is well conditioned: a trajectory
    # that can be held to the same bounds as the other two models
    adam_trajectory(root, "gpd", 1e-3, tweak=well_conditioned, tag="adam_gpd_native_wellcond")
    adam_trajectory_real_data(root)


if __name__ == "__main__":
    main()

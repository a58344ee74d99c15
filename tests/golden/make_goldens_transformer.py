#!/usr/bin/env python3
"""Generate tests/golden/transformer_*.npz by running the REFERENCE's transformer mixture VAE
(vae-models/mixture-vae-transformer.py, classes only - the file has no training loop) on CPU.

Usage (build container only):  python tests/golden/make_goldens_transformer.py /root/reference

Stored: inputs, the initial state_dict, every random draw, outputs, loss, parameter gradients.
Layout note: the reference runs its encoder seq-first [T,B,.]; the dropout masks are stored
batch-major ([B,T,.], attention [B,nhead,T,T]) - a transposition of data, nothing else.

Noise taps
  * torch.rand / torch.randn_like wrappers: choice, u, eps (reparameterize, :82-92);
  * forward hooks on every nn.Dropout: dropout1 / dropout / dropout2 of each encoder layer and the
    decoder's dropout (keep = (out != 0) | (in == 0));
  * the attention-weight dropout is drawn inside the fused scaled_dot_product_attention when
    need_weights=False (what nn.TransformerEncoderLayer passes), where it cannot be observed.
    attn="off"   : the layers' self_attn.dropout is set to 0 - the reference's fused path, no draw;
    attn="tapped": nn.MultiheadAttention.forward is wrapped to force need_weights=True, which makes
                   torch take its un-fused path (softmax -> F.dropout -> bmm: the same arithmetic),
                   and torch.nn.functional.dropout is wrapped to record that draw.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_goldens import NoiseTap, pack, synth_xy  # noqa: E402

REL = "vae-models/mixture-vae-transformer.py"
LAYER = "encoder.transformer_encoder.layers."


def load_reference(root: str):
    with open(os.path.join(root, REL), "r") as fh:
        text = fh.read()
    ns = {"__name__": "reference_transformer"}
    exec(compile(text, REL, "exec"), ns)
    return ns


class AttnTap:
    """Records the keep mask of every F.dropout call made from inside F.multi_head_attention_forward
    (3-D input [B*nhead, T, T]) while need_weights is forced on."""

    def __init__(self, T: int):
        self.T, self.keep = T, []

    def __enter__(self):
        import torch.nn.functional as F
        self._F = F
        self._dropout = F.dropout
        self._mha_fwd = torch.nn.MultiheadAttention.forward

        def dropout(inp, p=0.5, training=True, inplace=False):
            out = self._dropout(inp, p, training, inplace)
            if training and p > 0 and inp.dim() == 3 and inp.shape[-1] == self.T and inp.shape[-2] == self.T:
                self.keep.append((out != 0).float().detach())
            return out

        def mha_forward(mod, *a, **k):
            k["need_weights"] = True
            return self._mha_fwd(mod, *a, **k)

        F.dropout = dropout
        torch.nn.MultiheadAttention.forward = mha_forward
        return self

    def __exit__(self, *exc):
        self._F.dropout = self._dropout
        torch.nn.MultiheadAttention.forward = self._mha_fwd


def case(root, tag, B, T, C, d, nhead, layers, L, threshold, seed, train=True, attn="off", tweak=None):
    ns = load_reference(root)
    torch.manual_seed(seed)
    model = ns["VAE"](input_dim=C, d_model=d, num_layers=layers, nhead=nhead, latent_dim=L, future_steps=1)
    if tweak:
        tweak(model)
    if attn == "off":
        for lyr in model.encoder.transformer_encoder.layers:
            lyr.self_attn.dropout = 0.0
    x, y = synth_xy(B, T, C, seed + 1)
    if threshold is None:      # both |y| <= threshold and |y| > threshold occur
        threshold = float(y.abs().median())
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.train(train)
    model.zero_grad()
    with NoiseTap(model) as tap:
        if attn == "tapped" and train:
            with AttnTap(T) as atap:
                outs = model(x)
        else:
            atap = None
            outs = model(x)
        loss = model.loss_function(outs[0], y, *outs[1:], threshold)
    loss.backward()
    noise = {"choice": tap.rand[0], "u": tap.rand[1], "eps": tap.randn[0]}
    for name, keep in tap.keep.items():
        if name == "decoder.dropout":
            noise["dec_keep"] = keep
        elif name.startswith(LAYER):
            i, site = name[len(LAYER):].split(".")
            key = {"dropout1": "drop1_keep", "dropout": "ff_keep", "dropout2": "drop2_keep"}[site]
            noise[f"l{i}.{key}"] = keep.transpose(0, 1).contiguous()          # [T,B,.] -> [B,T,.]
    if atap is not None:
        assert len(atap.keep) == layers
        for i, keep in enumerate(atap.keep):
            noise[f"l{i}.attn_keep"] = keep.reshape(B, nhead, T, T)
    names = ["recon", "mean", "logvar", "scale", "shape"]
    grads = {k: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p))
             for k, p in model.named_parameters()}
    blob = {"meta/dims": np.array([B, T, C, d, nhead, layers, L, 1], dtype=np.int64),
            "meta/threshold": np.float64(threshold), "meta/beta": np.float64(model.beta),
            "meta/train": np.int64(train), "x": x.numpy(), "y": y.numpy(), "loss": loss.detach().numpy()}
    blob.update(pack("sd", sd0))
    for k, v in noise.items():   # keep masks are 0/1: one byte each
        blob["noise/" + k] = v.numpy().astype(np.uint8) if k.endswith("_keep") else v.numpy().copy()
    blob.update(pack("out", dict(zip(names, outs)))); blob.update(pack("grad", grads))
    np.savez(os.path.join(HERE, tag + ".npz"), **blob)
    print(f"{tag}: loss={float(loss.detach()):.9g} thr={threshold:.4g} params={sum(v.numel() for v in sd0.values())} "
          f"noise={sorted(noise)}")


def main():
    if len(sys.argv) < 2 or not os.path.isdir(sys.argv[1]):
        print("reference tree not given / not present: nothing to do")
        return
    root = sys.argv[1]
    torch.set_num_threads(1)

    def well_conditioned(model):  # GPD shapes away from 0, where scale/shape*((1-u)^-shape - 1) cancels
        with torch.no_grad():
            model.encoder.shape_layer_extreme.weight.mul_(0.25)
            model.encoder.shape_layer_extreme.bias.fill_(0.5)

    tiny = dict(B=6, T=12, C=3, d=16, nhead=2, layers=2, L=8)
    case(root, "transformer_tiny_eval", **tiny, threshold=None, seed=51, train=False)
    case(root, "transformer_tiny_train", **tiny, threshold=None, seed=52, attn="off", tweak=well_conditioned)
    case(root, "transformer_tiny_train_attndrop", **tiny, threshold=None, seed=53, attn="tapped",
         tweak=well_conditioned)
    # no |y| > threshold: mse over an empty selection => NaN loss (:107)
    case(root, "transformer_tiny_noextreme", **tiny, threshold=95.0, seed=54, attn="off", tweak=well_conditioned)
    # the hyper-parameters the repository's own transformer baseline uses
    # (baselines/hurdat2-data/baselines_hurdat_trim.py:168-170): d_model 64, nhead 8, 2 layers
    case(root, "transformer_native_train", B=8, T=16, C=5, d=64, nhead=8, layers=2, L=32, threshold=None,
         seed=55, attn="tapped", tweak=well_conditioned)


if __name__ == "__main__":
    main()

"""CPU oracle (test infrastructure, NOT a product path): a functional PyTorch-fp32
restatement of the reference's LSTM-VAE train step with every random draw fed in.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The shipped package never does; it fails loudly when its HIP library is absent.

Pinned against the reference itself: tests/golden/make_goldens.py runs the reference
classes (loaded from /root/reference in the build container), records the noise they
drew, and stores inputs/outputs/gradients as fixtures; tests/test_oracle_goldens.py
replays those fixtures through the functions below.

Every function works on a plain ``dict[str, Tensor]`` keyed like the reference modules'
``state_dict()`` (``encoder.lstm.weight_ih_l0`` ...), so reference checkpoints drop in.

Reference citations (relative to /root/reference/vae-models/):
  gaussian  : gaussian-lstm-DE.py:13-70
  gpd       : gpd-lstm-DE.py:10-71
  mixture   : mixture-vae-DE.py:10-18,36-130
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
State = Dict[str, Tensor]

DROP_P = 0.3  # nn.Dropout(0.3): gpd-lstm-DE.py:22,41 ; mixture-vae-DE.py:49,77
KINDS = ("gaussian", "gpd", "mixture")


# --------------------------------------------------------------------------- LSTM
def lstm_last_hidden(sd: State, x: Tensor, num_layers: int, prefix: str = "encoder.lstm.",
                     stock: bool = True) -> Tensor:
    """h_n[-1] of nn.LSTM(C, H, num_layers, batch_first=True) with zero initial state.

    gaussian-lstm-DE.py:16,28 (1 layer); gpd-lstm-DE.py:18,27 and mixture-vae-DE.py:45,60
    (3 layers).  ``stock=True`` calls the same fused ATen LSTM the reference's nn.LSTM
    dispatches to (oneDNN on CPU) - that is the reference's own CPU path and is what the
    CPU baseline times.  ``stock=False`` spells the recurrence out, gate order i,f,g,o.
    """
    B, T, _ = x.shape
    H = sd[prefix + "weight_hh_l0"].shape[1]
    if stock:
        flat = []
        for k in range(num_layers):
            flat += [sd[f"{prefix}weight_ih_l{k}"], sd[f"{prefix}weight_hh_l{k}"],
                     sd[f"{prefix}bias_ih_l{k}"], sd[f"{prefix}bias_hh_l{k}"]]
        h0 = x.new_zeros(num_layers, B, H)
        c0 = x.new_zeros(num_layers, B, H)
        _, h_n, _ = torch._VF.lstm(x, (h0, c0), flat, True, num_layers, 0.0, False, False, True)
        return h_n[-1]
    seq = x
    for k in range(num_layers):
        w_ih, w_hh = sd[f"{prefix}weight_ih_l{k}"], sd[f"{prefix}weight_hh_l{k}"]
        bias = sd[f"{prefix}bias_ih_l{k}"] + sd[f"{prefix}bias_hh_l{k}"]
        h = x.new_zeros(B, H)
        c = x.new_zeros(B, H)
        outs = []
        for t in range(T):
            a = seq[:, t] @ w_ih.t() + h @ w_hh.t() + bias
            i, f, g, o = a.split(H, dim=1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
            h = torch.sigmoid(o) * torch.tanh(c)
            outs.append(h)
        seq = torch.stack(outs, dim=1)
    return seq[:, -1]


def _linear(sd: State, name: str, x: Tensor) -> Tensor:
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"])


# When a test sets this to a dict, every ReLU site stores which units were active (pre-activation > 0) under the
# layer's name: the reference's gradient jumps wherever a unit changes sides, so a test that compares gradients
# across implementations first checks that both saw the same pattern.
RELU_TAPS: Optional[Dict[str, Tensor]] = None


def _relu_linear(sd: State, name: str, x: Tensor) -> Tensor:
    pre = _linear(sd, name, x)
    if RELU_TAPS is not None:
        RELU_TAPS[name] = (pre > 0).detach()
    return F.relu(pre)


def _dropout(x: Tensor, keep: Optional[Tensor]) -> Tensor:
    """nn.Dropout(0.3) with the Bernoulli keep-mask fed in; ``None`` = eval mode."""
    if keep is None:
        return x
    return x * keep * (1.0 / (1.0 - DROP_P))


# ----------------------------------------------------------------------- encoders
def encode_gaussian(sd: State, x: Tensor, stock: bool = True) -> Tuple[Tensor, Tensor]:
    """gaussian-lstm-DE.py:20-32 : LSTM -> mean / logvar heads."""
    h = lstm_last_hidden(sd, x, 1, stock=stock)
    return _linear(sd, "encoder.mean_layer", h), _linear(sd, "encoder.logvar_layer", h)


def _trunk(sd: State, x: Tensor, enc_keep: Optional[Tensor], stock: bool) -> Tensor:
    """LSTM(3) -> LayerNorm -> relu(fc1) -> dropout(relu(fc2)).
    gpd-lstm-DE.py:27-30 ; mixture-vae-DE.py:60-63."""
    h = lstm_last_hidden(sd, x, 3, stock=stock)
    h = F.layer_norm(h, (h.shape[1],), sd["encoder.layer_norm.weight"],
                     sd["encoder.layer_norm.bias"], 1e-5)
    h = _relu_linear(sd, "encoder.fc1", h)
    return _dropout(_relu_linear(sd, "encoder.fc2", h), enc_keep)


def encode_gpd(sd: State, x: Tensor, enc_keep: Optional[Tensor], stock: bool = True):
    """gpd-lstm-DE.py:26-33 : scale = exp(.), shape = clamp(., min=1e-6)."""
    hid = _trunk(sd, x, enc_keep, stock)
    scale = torch.exp(_linear(sd, "encoder.scale_layer_extreme", hid))
    shape = torch.clamp(_linear(sd, "encoder.shape_layer_extreme", hid), min=1e-6)
    return scale, shape


def encode_mixture(sd: State, x: Tensor, enc_keep: Optional[Tensor], stock: bool = True):
    """mixture-vae-DE.py:59-69 : four heads on the shared trunk."""
    hid = _trunk(sd, x, enc_keep, stock)
    mean = _linear(sd, "encoder.mean_layer_normal", hid)
    logvar = _linear(sd, "encoder.logvar_layer_normal", hid)
    scale = torch.exp(_linear(sd, "encoder.scale_layer_extreme", hid))
    shape = torch.clamp(_linear(sd, "encoder.shape_layer_extreme", hid), min=1e-6)
    return mean, logvar, scale, shape


def encode_mixture3(sd: State, x: Tensor, enc_keep: Optional[Tensor], stock: bool = True):
    """graph/vae_mixture_graph_model.py:113-136 with use_gcn=False: the mixture heads + the Bernoulli logits."""
    hid = _trunk(sd, x, enc_keep, stock)
    mean = _linear(sd, "encoder.mean_layer_normal", hid)
    logvar = _linear(sd, "encoder.logvar_layer_normal", hid)
    scale = torch.exp(_linear(sd, "encoder.scale_layer_extreme", hid))
    shape = torch.clamp(_linear(sd, "encoder.shape_layer_extreme", hid), min=1e-6)
    return mean, logvar, scale, shape, _linear(sd, "encoder.logits_layer_zero", hid)


# ------------------------------------------------------------------ reparameterise
def reparam_gaussian(mean: Tensor, logvar: Tensor, eps: Tensor) -> Tensor:
    """gaussian-lstm-DE.py:55-58 ; mixture-vae-DE.py:10-13 with eps fed in."""
    return mean + eps * torch.exp(0.5 * logvar)


def reparam_gpd(scale: Tensor, shape: Tensor, u: Tensor) -> Tensor:
    """gpd-lstm-DE.py:10-12 ; mixture-vae-DE.py:16-18 with u fed in.  Same op order
    (div, pow, sub, mul): the cancellation near shape = 1e-6 is part of the contract."""
    return scale / shape * ((1 - u) ** (-shape) - 1)


def mixture_select(zg: Tensor, zp: Tensor, pi_params: Tensor, choice: Tensor) -> Tensor:
    """mixture-vae-DE.py:101-105 : row i takes the Gaussian sample iff
    choice[i] < softmax(pi)[0].  The comparison carries no gradient to pi_params."""
    p0 = torch.softmax(pi_params.detach(), dim=0)[0]
    return torch.where((choice < p0)[:, None], zg, zp)


def mixture3_select(zg: Tensor, zp: Tensor, zb: Tensor, pi_params: Tensor, choice: Tensor) -> Tensor:
    """graph/vae_mixture_graph_model.py:191-198: rows with choice < p0 take the Gaussian sample, rows with
    p0 <= choice < p0 + p1 the GPD sample, the rest the Bernoulli sample (which carries no gradient)."""
    p = torch.softmax(pi_params.detach(), dim=0)
    first = (choice < p[0])[:, None]
    second = ((choice >= p[0]) & (choice < p[0] + p[1]))[:, None]
    return torch.where(first, zg, torch.where(second, zp, zb.detach()))


# ----------------------------------------------------------------------- decoders
def decode_gaussian(sd: State, z: Tensor) -> Tensor:
    """gaussian-lstm-DE.py:42-45."""
    return _linear(sd, "decoder.out", _relu_linear(sd, "decoder.dense1", z))


def decode_deep(sd: State, z: Tensor, dec_keep: Optional[Tensor]) -> Tensor:
    """gpd-lstm-DE.py:44-47 ; mixture-vae-DE.py:80-83."""
    z = _relu_linear(sd, "decoder.fc1", z)
    z = _dropout(_relu_linear(sd, "decoder.fc2", z), dec_keep)
    return _linear(sd, "decoder.out", z)


# ------------------------------------------------------------------------- losses
def kl_gaussian(mean: Tensor, logvar: Tensor) -> Tensor:
    return (-0.5 * torch.sum(1 + logvar - mean.pow(2) - logvar.exp(), dim=-1)).mean()


def kl_gpd(scale: Tensor, shape: Tensor, threshold: float) -> Tensor:
    return torch.mean(torch.log(scale) + (1 + shape * threshold / scale).log())


def loss_gaussian(pred, y, mean, logvar, beta=0.001) -> Tensor:
    """gaussian-lstm-DE.py:66-70."""
    return F.mse_loss(pred, y, reduction="mean") + beta * kl_gaussian(mean, logvar)


def loss_gpd(recon, y, scale, shape, threshold=95.0, beta=0.001) -> Tensor:
    """gpd-lstm-DE.py:67-71 (threshold is the module global at :84)."""
    return F.mse_loss(recon, y, reduction="mean") + beta * kl_gpd(scale, shape, threshold)


def loss_mixture(recon, y, mean, logvar, scale, shape, pi_params, threshold=95.0,
                 beta=0.001) -> Tensor:
    """mixture-vae-DE.py:115-130 ; classify_observations :36-39.  An empty normal mask
    makes the masked MSE NaN, exactly as the reference does."""
    normal = y.abs() <= threshold
    pi = torch.softmax(pi_params, dim=0)
    r_g = F.mse_loss(recon[normal], y[normal], reduction="mean")
    l_g = pi[0] * (r_g + beta * kl_gaussian(mean, logvar))
    r_p = F.mse_loss(recon, y, reduction="mean")
    l_p = pi[1] * (r_p + beta * kl_gpd(scale, shape, threshold))
    return F.mse_loss(recon, y, reduction="mean") + l_g + l_p


def loss_mixture3(recon, y, mean, logvar, scale, shape, logits, pi_params, threshold, beta=0.001) -> Tensor:
    """graph/vae_mixture_graph_model.py:222-285 with use_gpd = use_bernoulli = True and no D-KNN term;
    classify_observations :26-50.  y is [B, 1].  A class without rows contributes the constant 0 (:251,:266),
    an empty normal class makes the masked MSE NaN."""
    pi = torch.softmax(pi_params, dim=0)
    zero = y == 0
    normal = (y <= threshold) & ~zero
    extreme = (y > threshold) & ~zero
    r_g = F.mse_loss(recon[normal], y[normal], reduction="mean")
    total = pi[0] * (r_g + beta * kl_gaussian(mean, logvar))
    em = extreme.squeeze(-1)
    if int(em.sum()) != 0:
        excess = y[em] - threshold
        sc, sh = scale[em], shape[em]
        total = total + pi[1] * torch.mean(torch.log(sc) + (1 + 1 / sh) * torch.log(1 + sh * excess / sc))
    zm = zero.squeeze(-1)
    if int(zm.sum()) != 0:
        yz, lz = y[zm], logits[zm]
        total = total + pi[2] * -torch.mean(yz * torch.log(torch.sigmoid(lz)) + (1 - yz) * torch.log(1 - torch.sigmoid(lz)))
    return total


# ------------------------------------------------------------------ whole forward
def forward_loss(kind: str, sd: State, x: Tensor, y: Tensor, noise: Dict[str, Tensor],
                 threshold: float = 95.0, beta: float = 0.001, stock: bool = True):
    """One forward + loss.  ``noise`` keys: gaussian {eps}; gpd {enc_keep, u, dec_keep};
    mixture {enc_keep, choice, eps, u, dec_keep} (missing keep masks = eval mode).
    Returns (loss, outputs dict)."""
    if kind == "gaussian":
        mean, logvar = encode_gaussian(sd, x, stock)
        z = reparam_gaussian(mean, logvar, noise["eps"])
        pred = decode_gaussian(sd, z)
        loss = loss_gaussian(pred, y, mean, logvar, beta)
        return loss, {"recon": pred, "mean": mean, "logvar": logvar, "z": z}
    if kind == "gpd":
        scale, shape = encode_gpd(sd, x, noise.get("enc_keep"), stock)
        z = reparam_gpd(scale, shape, noise["u"])
        recon = decode_deep(sd, z, noise.get("dec_keep"))
        loss = loss_gpd(recon, y, scale, shape, threshold, beta)
        return loss, {"recon": recon, "scale": scale, "shape": shape, "z": z}
    if kind == "mixture":
        mean, logvar, scale, shape = encode_mixture(sd, x, noise.get("enc_keep"), stock)
        zg = reparam_gaussian(mean, logvar, noise["eps"])
        zp = reparam_gpd(scale, shape, noise["u"])
        z = mixture_select(zg, zp, sd["pi_params"], noise["choice"])
        recon = decode_deep(sd, z, noise.get("dec_keep"))
        loss = loss_mixture(recon, y, mean, logvar, scale, shape, sd["pi_params"],
                            threshold, beta)
        return loss, {"recon": recon, "mean": mean, "logvar": logvar, "scale": scale,
                      "shape": shape, "z": z}
    if kind == "mixture3":
        mean, logvar, scale, shape, logits = encode_mixture3(sd, x, noise.get("enc_keep"), stock)
        zg = reparam_gaussian(mean, logvar, noise["eps"])
        zp = reparam_gpd(scale, shape, noise["u"])
        z = mixture3_select(zg, zp, noise["bern"], sd["pi_params"], noise["choice"])
        recon = decode_deep(sd, z, noise.get("dec_keep"))
        loss = loss_mixture3(recon, y, mean, logvar, scale, shape, logits, sd["pi_params"], threshold, beta)
        return loss, {"recon": recon, "mean": mean, "logvar": logvar, "scale": scale, "shape": shape,
                      "logits": logits, "z": z}
    raise ValueError(kind)


def loss_and_grads(kind: str, sd: State, x, y, noise, threshold=95.0, beta=0.001,
                   stock: bool = True):
    """loss, outputs and d(loss)/d(param) for every entry of ``sd``."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    loss, outs = forward_loss(kind, leaves, x, y, noise, threshold, beta, stock)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
    return loss.detach(), {k: v.detach() for k, v in outs.items()}, grads


# --------------------------------------------------------------------------- Adam
def adam_step(sd: State, grads: State, m: State, v: State, step: int, lr: float,
              b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8) -> None:
    """torch.optim.Adam defaults as used at gaussian-lstm-DE.py:142, gpd-lstm-DE.py:88,
    mixture-vae-DE.py:147 (no weight decay, no amsgrad).  In place; ``step`` is 1-based."""
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    for k in sd:
        g = grads[k]
        m[k].mul_(b1).add_(g, alpha=1 - b1)
        v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v[k].sqrt() / math.sqrt(bc2)).add_(eps)
        sd[k].addcdiv_(m[k], denom, value=-lr / bc1)


# ------------------------------------------------------------ parameter factories
def param_shapes(kind: str, C: int, H: int, L: int, fs: int = 1) -> Dict[str, Tuple[int, ...]]:
    """state_dict key -> shape, in the reference modules' registration order."""
    shp: Dict[str, Tuple[int, ...]] = {}
    if kind == "mixture":
        shp["pi_params"] = (2,)
    if kind == "mixture3":
        shp["pi_params"] = (3,)
    layers = 1 if kind == "gaussian" else 3
    for k in range(layers):
        cin = C if k == 0 else H
        shp[f"encoder.lstm.weight_ih_l{k}"] = (4 * H, cin)
        shp[f"encoder.lstm.weight_hh_l{k}"] = (4 * H, H)
        shp[f"encoder.lstm.bias_ih_l{k}"] = (4 * H,)
        shp[f"encoder.lstm.bias_hh_l{k}"] = (4 * H,)

    def lin(name, n_out, n_in):
        shp[name + ".weight"] = (n_out, n_in)
        shp[name + ".bias"] = (n_out,)

    if kind == "gaussian":
        lin("encoder.mean_layer", L, H)
        lin("encoder.logvar_layer", L, H)
        lin("decoder.dense1", 500, L)
        lin("decoder.out", fs, 500)
        return shp
    shp["encoder.layer_norm.weight"] = (H,)
    shp["encoder.layer_norm.bias"] = (H,)
    lin("encoder.fc1", 128, H)
    lin("encoder.fc2", 64, 128)
    if kind in ("mixture", "mixture3"):
        lin("encoder.mean_layer_normal", L, 64)
        lin("encoder.logvar_layer_normal", L, 64)
    lin("encoder.scale_layer_extreme", L, 64)
    lin("encoder.shape_layer_extreme", L, 64)
    if kind == "mixture3":
        lin("encoder.logits_layer_zero", L, 64)
    lin("decoder.fc1", 128, L)
    lin("decoder.fc2", 500, 128)
    lin("decoder.out", fs, 500)
    return shp


def init_state(kind: str, C: int, H: int, L: int, fs: int = 1, seed: int = 0) -> State:
    """Random init with torch's default schemes (LSTM U(-1/sqrt(H), 1/sqrt(H)); Linear
    kaiming-uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), 1/sqrt(fan_in)); LayerNorm 1/0;
    pi_params [3.0, 0.01] per mixture-vae-DE.py:92)."""
    g = torch.Generator().manual_seed(seed)
    sd: State = {}
    for name, shape in param_shapes(kind, C, H, L, fs).items():
        if name == "pi_params":
            sd[name] = torch.tensor([3.0, 0.01, 0.01] if kind == "mixture3" else [3.0, 0.01])
        elif name == "encoder.layer_norm.weight":
            sd[name] = torch.ones(shape)
        elif name == "encoder.layer_norm.bias":
            sd[name] = torch.zeros(shape)
        else:
            if ".lstm." in name:
                bound = 1.0 / math.sqrt(H)
            else:
                base = name.rsplit(".", 1)[0] + ".weight"
                bound = 1.0 / math.sqrt(param_shapes(kind, C, H, L, fs)[base][1])
            sd[name] = (torch.rand(shape, generator=g) * 2 - 1) * bound
    return sd


def draw_noise(kind: str, B: int, L: int, seed: int = 0, train: bool = True) -> Dict[str, Tensor]:
    """Noise tensors in the reference's per-forward draw order (gaussian: randn_like;
    gpd: enc dropout, rand(B,L), dec dropout; mixture: enc dropout, rand(B), randn_like,
    rand(B,L), dec dropout)."""
    g = torch.Generator().manual_seed(seed)
    n: Dict[str, Tensor] = {}
    if kind == "gaussian":
        n["eps"] = torch.randn(B, L, generator=g)
        return n
    if train:
        n["enc_keep"] = (torch.rand(B, 64, generator=g) >= DROP_P).float()
    if kind == "mixture":
        n["choice"] = torch.rand(B, generator=g)
        n["eps"] = torch.randn(B, L, generator=g)
    if kind == "mixture3":   # eps, u, Bernoulli sample, choice (graph/vae_mixture_graph_model.py:183-191)
        n["eps"] = torch.randn(B, L, generator=g)
    n["u"] = torch.rand(B, L, generator=g)
    if kind == "mixture3":
        n["bern"] = (torch.rand(B, L, generator=g) < 0.5).float()
        n["choice"] = torch.rand(B, generator=g)
    if train:
        n["dec_keep"] = (torch.rand(B, 500, generator=g) >= DROP_P).float()
    return n

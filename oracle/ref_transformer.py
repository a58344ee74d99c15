"""CPU oracle (test infrastructure, NOT a product path) for the transformer-encoder mixture VAE:
a functional PyTorch-fp32 restatement of vae-models/mixture-vae-transformer.py with every random
draw (sampling noise and all four dropout sites of an encoder layer) fed in.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Pinned against the reference: tests/golden/make_goldens.py instantiates the reference's VAE
class, taps its noise, and stores inputs / outputs / gradients (transformer_*.npz);
tests/test_oracle_goldens.py replays them through the functions below.  One draw cannot be
tapped on the reference's own path: nn.TransformerEncoderLayer calls its attention with
need_weights=False, which routes the attention-weight dropout through the fused
scaled_dot_product_attention.  The fixtures therefore come in two kinds: attention dropout set
to zero on the reference's fused path, and attention dropout tapped with torch's un-fused
attention path (need_weights forced on), which is the same arithmetic spelled out.

State keys follow the reference's ``state_dict()``:
  encoder.embedding.{weight,bias}
  encoder.transformer_encoder.layers.<i>.self_attn.{in_proj_weight,in_proj_bias,out_proj.weight,out_proj.bias}
  encoder.transformer_encoder.layers.<i>.{linear1,linear2,norm1,norm2}.{weight,bias}
  encoder.{mean_layer_normal,logvar_layer_normal,scale_layer_extreme,shape_layer_extreme}.{weight,bias}
  decoder.{dense1,fc1,fc2,out,final_fcn}.{weight,bias}      pi_params

Noise keys (batch-major; the reference works seq-first, the fixtures are transposed when stored):
  choice [B], eps [B,L], u [B,L], dec_keep [B,500]
  l<i>.attn_keep [B,nhead,T,T], l<i>.drop1_keep [B,T,d], l<i>.ff_keep [B,T,dff], l<i>.drop2_keep [B,T,d]
A missing keep mask means that dropout site is off (eval mode).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
State = Dict[str, Tensor]

DROP_P = 0.2     # nn.TransformerEncoderLayer(dropout=0.2) and Decoder.dropout: mixture-vae-transformer.py:39,63
DIM_FF = 2048    # nn.TransformerEncoderLayer default dim_feedforward (not overridden at :39)
LN_EPS = 1e-5
LAYER = "encoder.transformer_encoder.layers."


def _linear(sd: State, name: str, x: Tensor) -> Tensor:
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"])


# Test aid: when a dict, the ReLU sites (the feed-forward blocks "l<i>.ff", decoder.fc1, decoder.fc2) record which
# units were active.  The reference's gradient jumps wherever a unit changes sides, so a test that compares
# gradients across implementations first checks that both saw the same pattern.
RELU_TAPS: Optional[Dict[str, Tensor]] = None
# ... and, when NEAR_ZERO is a dict, which units' pre-activations lie within NEAR_ZERO_MARGIN of zero: the units whose
# side a second implementation's rounding could change (a test drops them through the dropout masks it feeds in).
NEAR_ZERO: Optional[Dict[str, Tensor]] = None
NEAR_ZERO_MARGIN = 0.0


def _relu(name: str, pre: Tensor) -> Tensor:
    if RELU_TAPS is not None:
        RELU_TAPS[name] = (pre > 0).detach()
    if NEAR_ZERO is not None:
        NEAR_ZERO[name] = (pre.abs() < NEAR_ZERO_MARGIN).detach()
    return F.relu(pre)


def _dropout(x: Tensor, keep: Optional[Tensor]) -> Tensor:
    if keep is None:
        return x
    return x * keep * (1.0 / (1.0 - DROP_P))


def num_layers_of(sd: State) -> int:
    n = 0
    while f"{LAYER}{n}.linear1.weight" in sd:
        n += 1
    return n


def self_attention(sd: State, prefix: str, x: Tensor, nhead: int, attn_keep: Optional[Tensor]) -> Tensor:
    """nn.MultiheadAttention(d, nhead, dropout) self-attention on batch-major x [B,T,d]:
    packed in-projection, softmax(q k^T / sqrt(dh)) with dropout on the weights, out-projection.
    The reference reaches it through nn.TransformerEncoderLayer._sa_block (mixture-vae-transformer.py:39-40,49)."""
    B, T, d = x.shape
    dh = d // nhead
    qkv = F.linear(x, sd[prefix + "in_proj_weight"], sd[prefix + "in_proj_bias"])
    q, k, v = [t.reshape(B, T, nhead, dh).transpose(1, 2) for t in qkv.split(d, dim=-1)]   # [B,nh,T,dh]
    w = torch.softmax((q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(dh)), dim=-1)
    w = _dropout(w, attn_keep)
    o = (w @ v).transpose(1, 2).reshape(B, T, d)
    return F.linear(o, sd[prefix + "out_proj.weight"], sd[prefix + "out_proj.bias"])


def encoder_layer(sd: State, i: int, x: Tensor, nhead: int, noise: Dict[str, Tensor]) -> Tensor:
    """Post-norm nn.TransformerEncoderLayer (norm_first=False, ReLU):
    x = norm1(x + dropout1(SA(x))); x = norm2(x + dropout2(linear2(dropout(relu(linear1(x))))))."""
    p = f"{LAYER}{i}."
    d = x.shape[-1]
    sa = self_attention(sd, p + "self_attn.", x, nhead, noise.get(f"l{i}.attn_keep"))
    x = F.layer_norm(x + _dropout(sa, noise.get(f"l{i}.drop1_keep")), (d,), sd[p + "norm1.weight"],
                     sd[p + "norm1.bias"], LN_EPS)
    ff = _dropout(_relu(f"l{i}.ff", _linear(sd, p + "linear1", x)), noise.get(f"l{i}.ff_keep"))
    ff = _linear(sd, p + "linear2", ff)
    return F.layer_norm(x + _dropout(ff, noise.get(f"l{i}.drop2_keep")), (d,), sd[p + "norm2.weight"],
                        sd[p + "norm2.bias"], LN_EPS)


def encode(sd: State, x: Tensor, nhead: int, noise: Dict[str, Tensor]):
    """mixture-vae-transformer.py:47-55 : embedding, encoder stack, last token, four heads
    (scale = exp(.), shape NOT clamped)."""
    h = _linear(sd, "encoder.embedding", x)
    for i in range(num_layers_of(sd)):
        h = encoder_layer(sd, i, h, nhead, noise)
    last = h[:, -1]
    mean = _linear(sd, "encoder.mean_layer_normal", last)
    logvar = _linear(sd, "encoder.logvar_layer_normal", last)
    scale = torch.exp(_linear(sd, "encoder.scale_layer_extreme", last))
    shape = _linear(sd, "encoder.shape_layer_extreme", last)
    return mean, logvar, scale, shape


def sample(mean, logvar, scale, shape, pi_params, noise) -> Tensor:
    """mixture-vae-transformer.py:6-14,82-92."""
    zg = mean + noise["eps"] * torch.exp(0.5 * logvar)
    zp = scale / shape * ((1 - noise["u"]) ** (-shape) - 1)
    p0 = torch.softmax(pi_params.detach(), dim=0)[0]
    return torch.where((noise["choice"] < p0)[:, None], zg, zp)


def decode(sd: State, z: Tensor, dec_keep: Optional[Tensor]) -> Tensor:
    """mixture-vae-transformer.py:67-72 (dense1 is registered but unused)."""
    z = _relu("decoder.fc1", _linear(sd, "decoder.fc1", z))
    z = _dropout(_relu("decoder.fc2", _linear(sd, "decoder.fc2", z)), dec_keep)
    return _linear(sd, "decoder.final_fcn", _linear(sd, "decoder.out", z))


def loss(recon, y, mean, logvar, scale, shape, pi_params, threshold: float, beta: float = 1.0) -> Tensor:
    """mixture-vae-transformer.py:100-113 : the GPD component is scored on the extreme rows only and
    its KL is the Gaussian form applied to (scale, shape).  Either mask empty => NaN."""
    pi = torch.softmax(pi_params, dim=0)
    normal = y.abs() <= threshold
    extreme = y.abs() > threshold
    r_g = F.mse_loss(recon[normal], y[normal], reduction="mean")
    kl_g = (-0.5 * torch.sum(1 + logvar - mean.pow(2) - logvar.exp(), dim=-1)).mean()
    r_p = F.mse_loss(recon[extreme], y[extreme], reduction="mean")
    kl_p = (-0.5 * torch.sum(1 + scale - shape.pow(2) - scale.exp(), dim=-1)).mean()
    return F.mse_loss(recon, y, reduction="mean") + pi[0] * (r_g + beta * kl_g) + pi[1] * (r_p + beta * kl_p)


def forward_loss(sd: State, x: Tensor, y: Tensor, noise: Dict[str, Tensor], nhead: int, threshold: float,
                 beta: float = 1.0):
    mean, logvar, scale, shape = encode(sd, x, nhead, noise)
    z = sample(mean, logvar, scale, shape, sd["pi_params"], noise)
    recon = decode(sd, z, noise.get("dec_keep"))
    val = loss(recon, y, mean, logvar, scale, shape, sd["pi_params"], threshold, beta)
    return val, {"recon": recon, "mean": mean, "logvar": logvar, "scale": scale, "shape": shape, "z": z}


def loss_and_grads(sd: State, x, y, noise, nhead: int, threshold: float, beta: float = 1.0):
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    val, outs = forward_loss(leaves, x, y, noise, nhead, threshold, beta)
    val.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
    return val.detach(), {k: v.detach() for k, v in outs.items()}, grads


# ------------------------------------------------------------------ parameter / noise factories
def param_shapes(C: int, d: int, layers: int, L: int, fs: int = 1, dff: int = DIM_FF) -> Dict[str, Tuple[int, ...]]:
    """state_dict key -> shape in the reference's registration order (VAE.__init__ :77-80)."""
    shp: Dict[str, Tuple[int, ...]] = {"pi_params": (2,)}

    def lin(name, n_out, n_in):
        shp[name + ".weight"] = (n_out, n_in)
        shp[name + ".bias"] = (n_out,)

    lin("encoder.embedding", d, C)
    for i in range(layers):
        p = f"{LAYER}{i}."
        shp[p + "self_attn.in_proj_weight"] = (3 * d, d)
        shp[p + "self_attn.in_proj_bias"] = (3 * d,)
        lin(p + "self_attn.out_proj", d, d)
        lin(p + "linear1", dff, d)
        lin(p + "linear2", d, dff)
        for n in ("norm1", "norm2"):
            shp[p + n + ".weight"] = (d,)
            shp[p + n + ".bias"] = (d,)
    for n in ("mean_layer_normal", "logvar_layer_normal", "scale_layer_extreme", "shape_layer_extreme"):
        lin("encoder." + n, L, d)
    lin("decoder.dense1", 500, L)
    lin("decoder.fc1", 128, L)
    lin("decoder.fc2", 500, 128)
    lin("decoder.out", fs, 500)
    lin("decoder.final_fcn", fs, fs)
    return shp


def init_state(C: int, d: int, layers: int, L: int, fs: int = 1, seed: int = 0, dff: int = DIM_FF) -> State:
    """Random init in torch's default schemes: Linear U(+-1/sqrt(fan_in)); attention in-projection
    xavier-uniform with zero biases (in and out); LayerNorm 1/0; pi_params [3.0, 0.01] (:80)."""
    g = torch.Generator().manual_seed(seed)
    shapes = param_shapes(C, d, layers, L, fs, dff)
    sd: State = {}
    for name, shape in shapes.items():
        if name == "pi_params":
            sd[name] = torch.tensor([3.0, 0.01])
        elif ".norm" in name:
            sd[name] = torch.ones(shape) if name.endswith("weight") else torch.zeros(shape)
        elif name.endswith("in_proj_bias") or name.endswith("out_proj.bias"):
            sd[name] = torch.zeros(shape)
        elif name.endswith("in_proj_weight"):
            bound = math.sqrt(6.0 / (shape[0] + shape[1]))
            sd[name] = (torch.rand(shape, generator=g) * 2 - 1) * bound
        else:
            base = name.rsplit(".", 1)[0] + ".weight"
            bound = 1.0 / math.sqrt(shapes[base][1])
            sd[name] = (torch.rand(shape, generator=g) * 2 - 1) * bound
    return sd


def draw_noise(B: int, T: int, d: int, nhead: int, layers: int, L: int, seed: int = 0, train: bool = True,
               dff: int = DIM_FF) -> Dict[str, Tensor]:
    g = torch.Generator().manual_seed(seed)
    n: Dict[str, Tensor] = {}

    def keep(*shape):
        return (torch.rand(*shape, generator=g) >= DROP_P).float()

    if train:
        for i in range(layers):
            n[f"l{i}.attn_keep"] = keep(B, nhead, T, T)
            n[f"l{i}.drop1_keep"] = keep(B, T, d)
            n[f"l{i}.ff_keep"] = keep(B, T, dff)
            n[f"l{i}.drop2_keep"] = keep(B, T, d)
    n["choice"] = torch.rand(B, generator=g)
    n["eps"] = torch.randn(B, L, generator=g)
    n["u"] = torch.rand(B, L, generator=g)
    if train:
        n["dec_keep"] = keep(B, 500)
    return n

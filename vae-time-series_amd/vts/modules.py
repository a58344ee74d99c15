"""Drop-in nn.Modules: ``LSTM`` (nn.LSTM-compatible parameters and return value) and the three
VAE families with the reference's attribute names, so a reference ``state_dict`` loads as is.

Reference classes mirrored (vae-models/): gaussian-lstm-DE.py:13-70, gpd-lstm-DE.py:15-71,
mixture-vae-DE.py:42-130.  Every forward takes an optional ``noise`` dict so the random draws
(eps, u, choice, dropout keep masks) can be fed in for parity; otherwise they are drawn on the
device in the reference's order.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import _lib as L
from . import functional as F

Noise = Optional[Dict[str, torch.Tensor]]


class LSTM(nn.Module):
    """nn.LSTM(input_size, hidden_size, num_layers, batch_first=True) on the vts kernels.

    Parameters are named and shaped like nn.LSTM's (weight_ih_l{k} [4H,in], weight_hh_l{k}
    [4H,H], bias_ih_l{k}, bias_hh_l{k} [4H]; gate order i,f,g,o; init U(-1/sqrt(H), 1/sqrt(H))).
    ``forward(x) -> (output, (h_n, c_n))``.  Initial state is zero; bidirectional, proj_size,
    inter-layer dropout, packed sequences and seq-first layout are not supported."""

    def __init__(self, input_size: int, hidden_size: int, num_layers: int = 1, bias: bool = True,
                 batch_first: bool = True, dropout: float = 0.0, bidirectional: bool = False,
                 proj_size: int = 0):
        super().__init__()
        if not batch_first:
            raise NotImplementedError("vts.LSTM supports batch_first=True only")
        if not bias or dropout != 0.0 or bidirectional or proj_size != 0:
            raise NotImplementedError("vts.LSTM: bias=False, dropout, bidirectional and proj_size are not supported")
        if hidden_size not in (16, 32, 64, 128, 256):
            raise NotImplementedError("vts.LSTM: hidden_size must be one of 16, 32, 64, 128, 256")
        self.input_size, self.hidden_size, self.num_layers = input_size, hidden_size, num_layers
        self.batch_first = True
        bound = 1.0 / math.sqrt(hidden_size)
        for k in range(num_layers):
            cin = input_size if k == 0 else hidden_size
            for name, shape in ((f"weight_ih_l{k}", (4 * hidden_size, cin)),
                                (f"weight_hh_l{k}", (4 * hidden_size, hidden_size)),
                                (f"bias_ih_l{k}", (4 * hidden_size,)), (f"bias_hh_l{k}", (4 * hidden_size,))):
                self.register_parameter(name, nn.Parameter(torch.empty(shape).uniform_(-bound, bound)))

    def flat_weights(self):
        out = []
        for k in range(self.num_layers):
            out += [getattr(self, f"weight_ih_l{k}"), getattr(self, f"weight_hh_l{k}"),
                    getattr(self, f"bias_ih_l{k}"), getattr(self, f"bias_hh_l{k}")]
        return out

    def forward(self, x: torch.Tensor, hx=None, need_output: bool = True, top_state_only: bool = False):
        """As nn.LSTM.forward.  need_output=False: the caller reads only (h_n, c_n) - the reference's encoders
        do, e.g. gaussian-lstm-DE.py:28 - and `output` comes back as an empty tensor (saves writing the top
        layer's fp32 sequence on the bench-sized path).  top_state_only=True: the caller reads only h_n[-1] /
        c_n[-1]; both come back as [1, B, H] (the top layer's final state) without stacking the other layers'."""
        if hx is not None:
            raise NotImplementedError("vts.LSTM: a non-zero initial state is not supported")
        if isinstance(x, nn.utils.rnn.PackedSequence):
            raise NotImplementedError("vts.LSTM: packed sequences are not supported")
        if x.dim() != 3 or x.shape[2] != self.input_size:
            raise ValueError(f"vts.LSTM: expected input [B, T, {self.input_size}], got {tuple(x.shape)}")
        out, h_n, c_n = F.lstm(x, self.flat_weights(), self.num_layers, need_output, top_state_only)
        return out, (h_n, c_n)


class Linear(nn.Linear):
    """nn.Linear whose forward runs the vts GEMM with an optional fused activation."""

    def forward(self, x: torch.Tensor, act: int = L.ACT_NONE, keep: Optional[torch.Tensor] = None) -> torch.Tensor:
        return F.linear(x, self.weight, self.bias, act, keep)


class LayerNorm(nn.LayerNorm):
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return F.layer_norm(x, self.weight, self.bias, self.eps)


def _keep_mask(noise: Noise, key: str, training: bool, shape, device) -> Optional[torch.Tensor]:
    """Dropout keep mask: fed in, drawn (train mode) or None (eval mode = no dropout)."""
    if noise is not None and key in noise:
        return noise[key]
    if not training:
        return None
    # one launch (uniform draw, compare and conversion in the generator kernel): keep with probability 1 - p
    return torch.empty(shape, device=device).bernoulli_(1.0 - F.DROP_P)


# ------------------------------------------------------------------------------ Gaussian VAE
class GaussianEncoder(nn.Module):
    def __init__(self, input_dim, lstm_output_dim, latent_dim):
        super().__init__()
        self.lstm = LSTM(input_dim, lstm_output_dim, batch_first=True)
        self.mean_layer = Linear(lstm_output_dim, latent_dim)
        self.logvar_layer = Linear(lstm_output_dim, latent_dim)

    def forward(self, x):
        _, (h_n, _) = self.lstm(x, need_output=False, top_state_only=True)
        h = h_n[-1]
        heads = [(self.mean_layer.weight, self.mean_layer.bias, L.ACT_NONE),
                 (self.logvar_layer.weight, self.logvar_layer.bias, L.ACT_NONE)]
        if F.mlp_chain_usable(h, [], heads):
            return F.mlp_chain(h, [], heads)
        return self.mean_layer(h), self.logvar_layer(h)


class GaussianDecoder(nn.Module):
    def __init__(self, latent_dim, future_steps):
        super().__init__()
        self.dense1 = Linear(latent_dim, 500)
        self.out = Linear(500, future_steps)

    def forward(self, z):
        trunk = [(self.dense1.weight, self.dense1.bias, L.ACT_RELU)]
        heads = [(self.out.weight, self.out.bias, L.ACT_NONE)]
        if F.mlp_chain_usable(z, trunk, heads):
            return F.mlp_chain(z, trunk, heads, names=("decoder.dense1",))[0]
        return self.out(self.dense1(z, act=L.ACT_RELU))


class GaussianVAE(nn.Module):
    """gaussian-lstm-DE.py:48-70."""
    kind = "gaussian"

    def __init__(self, input_dim, lstm_output_dim, latent_dim, future_steps, beta=0.001):
        super().__init__()
        self.encoder = GaussianEncoder(input_dim, lstm_output_dim, latent_dim)
        self.decoder = GaussianDecoder(latent_dim, future_steps)
        self.beta = beta
        self._kl = None

    def reparameterize(self, mean, logvar, eps=None):
        return F.reparameterize_gaussian(mean, logvar, eps)

    def forward(self, x, noise: Noise = None):
        z_mean, z_log_var = self.encoder(x)
        # sampling also leaves the KL sums of these exact tensors for loss_function
        z, self._kl = F.reparameterize_gaussian(z_mean, z_log_var, None if noise is None else noise.get("eps"),
                                                with_kl=True)
        return self.decoder(z), z_mean, z_log_var

    def loss_function(self, forecasting, y, z_mean, z_log_var):
        kl, self._kl = self._kl, None
        return F.gaussian_elbo(forecasting, y, z_mean, z_log_var, self.beta, kl=kl)


# --------------------------------------------------------------------- GPD and mixture VAEs
class DeepEncoder(nn.Module):
    """3-layer LSTM -> LayerNorm -> relu(fc1) -> dropout(relu(fc2)) -> heads."""

    def __init__(self, input_dim, lstm_output_dim, latent_dim, gaussian_heads: bool):
        super().__init__()
        self.lstm = LSTM(input_dim, lstm_output_dim, num_layers=3, batch_first=True)
        self.layer_norm = LayerNorm(lstm_output_dim)
        self.fc1 = Linear(lstm_output_dim, 128)
        self.fc2 = Linear(128, 64)
        if gaussian_heads:
            self.mean_layer_normal = Linear(64, latent_dim)
            self.logvar_layer_normal = Linear(64, latent_dim)
        self.scale_layer_extreme = Linear(64, latent_dim)
        self.shape_layer_extreme = Linear(64, latent_dim)
        self.gaussian_heads = gaussian_heads

    def forward(self, x, noise: Noise = None):
        _, (h_n, _) = self.lstm(x, need_output=False, top_state_only=True)
        h = self.layer_norm(h_n[-1])
        keep = _keep_mask(noise, "enc_keep", self.training, (h.shape[0], 64), h.device)
        trunk = [(self.fc1.weight, self.fc1.bias, L.ACT_RELU), (self.fc2.weight, self.fc2.bias, L.ACT_RELU_DROPOUT)]
        heads = self._heads()
        if F.mlp_chain_usable(h, trunk, heads):
            return F.mlp_chain(h, trunk, heads, [None, keep], names=("encoder.fc1", "encoder.fc2"))
        hidden = self.fc1(h, act=L.ACT_RELU)
        hidden = self.fc2(hidden, act=L.ACT_RELU_DROPOUT, keep=keep)
        scale = self.scale_layer_extreme(hidden, act=L.ACT_EXP)
        shape = self.shape_layer_extreme(hidden, act=L.ACT_CLAMP_MIN)
        if self.gaussian_heads:
            return self.mean_layer_normal(hidden), self.logvar_layer_normal(hidden), scale, shape
        return scale, shape

    def _heads(self):
        """(weight, bias, activation) of the heads in the order forward returns them."""
        hs = []
        if self.gaussian_heads:
            hs += [(self.mean_layer_normal.weight, self.mean_layer_normal.bias, L.ACT_NONE),
                   (self.logvar_layer_normal.weight, self.logvar_layer_normal.bias, L.ACT_NONE)]
        return hs + [(self.scale_layer_extreme.weight, self.scale_layer_extreme.bias, L.ACT_EXP),
                     (self.shape_layer_extreme.weight, self.shape_layer_extreme.bias, L.ACT_CLAMP_MIN)]


class DeepDecoder(nn.Module):
    def __init__(self, latent_dim, future_steps):
        super().__init__()
        self.fc1 = Linear(latent_dim, 128)
        self.fc2 = Linear(128, 500)
        self.out = Linear(500, future_steps)

    def forward(self, z, noise: Noise = None):
        keep = _keep_mask(noise, "dec_keep", self.training, (z.shape[0], 500), z.device)
        trunk = [(self.fc1.weight, self.fc1.bias, L.ACT_RELU), (self.fc2.weight, self.fc2.bias, L.ACT_RELU_DROPOUT)]
        heads = [(self.out.weight, self.out.bias, L.ACT_NONE)]
        if F.mlp_chain_usable(z, trunk, heads):
            return F.mlp_chain(z, trunk, heads, [None, keep], names=("decoder.fc1", "decoder.fc2"))[0]
        z = self.fc1(z, act=L.ACT_RELU)
        z = self.fc2(z, act=L.ACT_RELU_DROPOUT, keep=keep)
        return self.out(z)


class GPDVAE(nn.Module):
    """gpd-lstm-DE.py:50-71.  ``threshold`` replaces the script's module global (:84)."""
    kind = "gpd"

    def __init__(self, input_dim, lstm_output_dim, latent_dim, future_steps, beta=0.001, threshold=95.0):
        super().__init__()
        self.encoder = DeepEncoder(input_dim, lstm_output_dim, latent_dim, gaussian_heads=False)
        self.decoder = DeepDecoder(latent_dim, future_steps)
        self.beta, self.threshold = beta, threshold
        self._kl = None

    def reparameterize(self, scale, shape, u=None):
        return F.reparameterize_gpd(scale, shape, scale.size(), u)

    def forward(self, x, noise: Noise = None):
        # draw order of the reference: encoder dropout, u, decoder dropout
        scale, shape = self.encoder(x, noise)
        z, self._kl = F.reparameterize_gpd(scale, shape, scale.size(), None if noise is None else noise.get("u"),
                                           kl_threshold=self.threshold)
        return self.decoder(z, noise), scale, shape

    def loss_function(self, reconstructed, y, scale, shape, threshold=None):
        thr = self.threshold if threshold is None else threshold
        kl, self._kl = self._kl, None
        return F.gpd_elbo(reconstructed, y, scale, shape, thr, self.beta, kl=kl)


class MixtureVAE(nn.Module):
    """mixture-vae-DE.py:86-130.  ``threshold`` replaces the script's module global (:143)."""
    kind = "mixture"

    def __init__(self, input_dim, lstm_output_dim, latent_dim, future_steps, beta=0.001, threshold=95.0):
        super().__init__()
        self.encoder = DeepEncoder(input_dim, lstm_output_dim, latent_dim, gaussian_heads=True)
        self.decoder = DeepDecoder(latent_dim, future_steps)
        self.beta, self.threshold = beta, threshold
        self._kl = None
        self.pi_params = nn.Parameter(torch.tensor([3.0, 0.01]))

    def reparameterize(self, mean, logvar, scale, shape, choice=None, eps=None, u=None):
        return F.reparameterize_mixture(mean, logvar, scale, shape, self.pi_params, choice, eps, u)

    def forward(self, x, noise: Noise = None):
        # draw order of the reference: encoder dropout, choice, eps, u, decoder dropout
        mean, logvar, scale, shape = self.encoder(x, noise)
        n = noise or {}
        z, self._kl = F.reparameterize_mixture(mean, logvar, scale, shape, self.pi_params, n.get("choice"),
                                               n.get("eps"), n.get("u"), kl_threshold=self.threshold)
        return self.decoder(z, noise), mean, logvar, scale, shape

    def loss_function(self, reconstructed, y, mean, logvar, scale, shape, threshold=None):
        thr = self.threshold if threshold is None else threshold
        kl, self._kl = self._kl, None
        return F.mixture_elbo(reconstructed, y, mean, logvar, scale, shape, self.pi_params, thr, self.beta, kl=kl)


# ------------------------------------------------- three-component mixture VAE (graph model, no GCN)
class ThreeComponentEncoder(DeepEncoder):
    """LSTM_GCN_Encoder of vae-models/graph/vae_mixture_graph_model.py:74-136 with use_gcn=False: the mixture
    encoder plus the Bernoulli head ``logits_layer_zero``."""

    def __init__(self, input_dim, lstm_output_dim, latent_dim):
        super().__init__(input_dim, lstm_output_dim, latent_dim, gaussian_heads=True)
        self.logits_layer_zero = Linear(64, latent_dim)

    def forward(self, x, noise: Noise = None):
        _, (h_n, _) = self.lstm(x, need_output=False, top_state_only=True)
        h = self.layer_norm(h_n[-1])
        keep = _keep_mask(noise, "enc_keep", self.training, (h.shape[0], 64), h.device)
        trunk = [(self.fc1.weight, self.fc1.bias, L.ACT_RELU), (self.fc2.weight, self.fc2.bias, L.ACT_RELU_DROPOUT)]
        heads = self._heads() + [(self.logits_layer_zero.weight, self.logits_layer_zero.bias, L.ACT_NONE)]
        if F.mlp_chain_usable(h, trunk, heads):
            return F.mlp_chain(h, trunk, heads, [None, keep], names=("encoder.fc1", "encoder.fc2"))
        hidden = self.fc1(h, act=L.ACT_RELU)
        hidden = self.fc2(hidden, act=L.ACT_RELU_DROPOUT, keep=keep)
        return (self.mean_layer_normal(hidden), self.logvar_layer_normal(hidden),
                self.scale_layer_extreme(hidden, act=L.ACT_EXP), self.shape_layer_extreme(hidden, act=L.ACT_CLAMP_MIN),
                self.logits_layer_zero(hidden))


class ThreeComponentVAE(nn.Module):
    """``VAE`` of vae-models/graph/vae_mixture_graph_model.py:155-285 in the configuration its LD2011 driver
    uses (graph/train_ld_2011.py:91): Gaussian + GPD + Bernoulli components, no GCN layer, no D-KNN imputer.
    Constructor, ``forward`` and ``loss_function`` keep the reference's argument order; the graph options raise."""
    kind = "mixture3"

    def __init__(self, input_dim, lstm_output_dim, gcn_output_dim, latent_dim, future_steps, beta=0.001,
                 use_d_knn=False, use_gcn=False, use_gpd=True, use_bernoulli=True):
        super().__init__()
        if use_gcn or use_d_knn:
            raise NotImplementedError("vts.ThreeComponentVAE: the GCN layer and the D-KNN imputer are not built")
        if not (use_gpd and use_bernoulli):
            raise NotImplementedError("vts.ThreeComponentVAE: use vts.MixtureVAE / vts.GaussianVAE for fewer components")
        if future_steps != 1:
            raise NotImplementedError("vts.ThreeComponentVAE: future_steps must be 1 (the reference's loss masks the "
                                      "[B, L] latent tensors with masks made from y)")
        self.encoder = ThreeComponentEncoder(input_dim, lstm_output_dim, latent_dim)
        self.decoder = DeepDecoder(latent_dim, future_steps)
        self.beta = beta
        self.threshold = None
        self.pi_params = nn.Parameter(torch.tensor([3.0, 0.01, 0.01]))

    def reparameterize(self, mean, logvar, scale, shape, logits, choice=None, eps=None, u=None, bern=None):
        return F.reparameterize_mixture3(mean, logvar, scale, shape, logits, self.pi_params, choice, eps, u, bern)

    def forward(self, x_full, x_missing=None, edge_index=None, noise: Noise = None):
        # draw order of the reference: encoder dropout, eps, u, Bernoulli, choice, decoder dropout
        mean, logvar, scale, shape, logits = self.encoder(x_full, noise)
        n = noise or {}
        z = F.reparameterize_mixture3(mean, logvar, scale, shape, logits, self.pi_params, n.get("choice"), n.get("eps"),
                                      n.get("u"), n.get("bern"))
        return self.decoder(z, noise), mean, logvar, scale, shape, logits

    def loss_function(self, reconstructed, y, z_mean_normal, z_log_var_normal, threshold=None, z_scale_extreme=None,
                      z_shape_extreme=None, z_logits_zero=None, x_full=None, x_missing=None):
        thr = self.threshold if threshold is None else threshold
        if thr is None:
            raise ValueError("ThreeComponentVAE.loss_function needs a threshold")
        return F.mixture3_elbo(reconstructed, y, z_mean_normal, z_log_var_normal, float(thr), z_scale_extreme,
                               z_shape_extreme, z_logits_zero, self.pi_params, self.beta)


# ------------------------------------------------------------------- transformer mixture VAE
TRANSFORMER_DROP_P = 0.2   # nn.TransformerEncoderLayer(dropout=0.2), Decoder nn.Dropout(0.2)


class _Site:
    """One dropout site of one forward pass: its mask is fed in (parity), drawn by the library's
    counter hash from a fresh seed (training), or absent (eval)."""

    def __init__(self, noise: Noise, key: str, training: bool):
        entry = noise.get(key) if noise is not None else None
        if entry is False:        # the site is switched off explicitly (fixtures recorded without that draw)
            self.keep, self.seed = None, 0
            return
        self.keep = entry
        # keyed on this site's own entry: a noise dict that feeds only some draws (eps / u / choice, say) must not
        # switch dropout off at the sites it does not mention
        self.seed = F.new_seed() if (self.keep is None and training) else 0


class SelfAttention(nn.Module):
    """Parameter holder shaped like nn.MultiheadAttention (in_proj_weight [3d,d], in_proj_bias [3d],
    out_proj.{weight,bias}) with torch's initialisation (xavier-uniform in-projection, zero biases)."""

    def __init__(self, d_model: int, nhead: int):
        super().__init__()
        if d_model % nhead or (d_model // nhead) not in (8, 16, 32):
            raise NotImplementedError("vts attention: head width d_model/nhead must be 8, 16 or 32")
        self.embed_dim, self.num_heads = d_model, nhead
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d_model, d_model))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d_model))
        self.out_proj = Linear(d_model, d_model)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)


class TransformerEncoderLayer(nn.Module):
    """nn.TransformerEncoderLayer(d_model, nhead, dim_feedforward=2048, dropout=0.2) as the reference
    builds it (mixture-vae-transformer.py:39): post-norm, ReLU, four dropout sites.  Works on
    batch-major [B, T, d] (the reference permutes to seq-first at :49; attention is per window either way).

    ``last_only``: produce the layer's output for the final position only ([B, d]).  The reference reads
    nothing but transformer_out[-1] (:50), so the last layer of the stack needs keys and values for every
    position but a query, residual, norms and feed-forward for one: 1/T of the layer's work."""

    def __init__(self, d_model: int, nhead: int, dim_feedforward: int = 2048, dropout: float = TRANSFORMER_DROP_P):
        super().__init__()
        self.self_attn = SelfAttention(d_model, nhead)
        self.linear1 = Linear(d_model, dim_feedforward)
        self.linear2 = Linear(dim_feedforward, d_model)
        self.norm1 = LayerNorm(d_model)
        self.norm2 = LayerNorm(d_model)
        self.p = dropout

    def forward(self, x: torch.Tensor, index: int, noise: Noise = None, last_only: bool = False) -> torch.Tensor:
        B, T, d = x.shape
        sa, p = self.self_attn, self.p
        sites = {k: _Site(noise, f"l{index}.{k}_keep", self.training) for k in ("attn", "drop1", "ff", "drop2")}
        x2 = x.reshape(B * T, d)
        if last_only:
            pick = lambda m: None if m is None else m[:, -1]
            res = x[:, -1, :]
            q = F.linear(res, sa.in_proj_weight[:d], sa.in_proj_bias[:d])
            kv = F.linear(x2, sa.in_proj_weight[d:], sa.in_proj_bias[d:]).view(B, T, 2 * d)
            akeep = sites["attn"].keep
            att = F.multi_head_attention(kv, sa.num_heads, q=q, keep=None if akeep is None else akeep[:, :, -1:, :],
                                         drop_p=p, seed=sites["attn"].seed)
        else:
            pick = lambda m: m
            res = x2
            qkv = F.linear(x2, sa.in_proj_weight, sa.in_proj_bias).view(B, T, 3 * d)
            att = F.multi_head_attention(qkv, sa.num_heads, keep=sites["attn"].keep, drop_p=p,
                                         seed=sites["attn"].seed).view(B * T, d)
        h = F.dropout_add(res, sa.out_proj(att), pick(sites["drop1"].keep), p, sites["drop1"].seed)
        h = self.norm1(h)
        ffk = pick(sites["ff"].keep)
        ff = F.feed_forward(h, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias,
                            None if ffk is None else ffk.reshape(h.shape[0], -1).float(), p, sites["ff"].seed)
        out = self.norm2(F.dropout_add(h, ff, pick(sites["drop2"].keep), p, sites["drop2"].seed))
        return out if last_only else out.view(B, T, d)


class _EncoderStack(nn.Module):
    """Holds ``layers`` under the attribute path nn.TransformerEncoder uses."""

    def __init__(self, d_model, nhead, num_layers):
        super().__init__()
        self.layers = nn.ModuleList([TransformerEncoderLayer(d_model, nhead) for _ in range(num_layers)])


class TransformerEncoder(nn.Module):
    """mixture-vae-transformer.py:35-55: embedding -> encoder stack -> last position -> four heads
    (scale = exp(.), shape not clamped)."""

    def __init__(self, input_dim, d_model, num_layers, nhead, latent_dim):
        super().__init__()
        self.embedding = Linear(input_dim, d_model)
        self.transformer_encoder = _EncoderStack(d_model, nhead, num_layers)
        self.mean_layer_normal = Linear(d_model, latent_dim)
        self.logvar_layer_normal = Linear(d_model, latent_dim)
        self.scale_layer_extreme = Linear(d_model, latent_dim)
        self.shape_layer_extreme = Linear(d_model, latent_dim)

    def forward(self, x, noise: Noise = None):
        B, T, C = x.shape
        h = self.embedding(x.reshape(B * T, C)).view(B, T, -1)
        layers = self.transformer_encoder.layers
        for i, layer in enumerate(layers):
            h = layer(h, i, noise, last_only=(i == len(layers) - 1))
        return (self.mean_layer_normal(h), self.logvar_layer_normal(h),
                self.scale_layer_extreme(h, act=L.ACT_EXP), self.shape_layer_extreme(h))


class TransformerDecoder(nn.Module):
    """mixture-vae-transformer.py:57-72 (dense1 is registered and never used, as there)."""

    def __init__(self, latent_dim, future_steps):
        super().__init__()
        self.dense1 = Linear(latent_dim, 500)
        self.fc1 = Linear(latent_dim, 128)
        self.fc2 = Linear(128, 500)
        self.out = Linear(500, future_steps)
        self.final_fcn = Linear(future_steps, future_steps)

    def forward(self, z, noise: Noise = None):
        z = self.fc1(z, act=L.ACT_RELU)
        if F.CHAIN_TAPS is not None:
            F.CHAIN_TAPS["decoder.fc1"] = z.detach()
        site = _Site(noise, "dec_keep", self.training)
        keep = None if site.keep is None else site.keep.float()
        z = F.linear(z, self.fc2.weight, self.fc2.bias, L.ACT_RELU_DROPOUT, keep, TRANSFORMER_DROP_P, site.seed)
        if F.CHAIN_TAPS is not None:
            F.CHAIN_TAPS["decoder.fc2"] = z.detach()
        return self.final_fcn(self.out(z))


class TransformerMixtureVAE(nn.Module):
    """mixture-vae-transformer.py:74-113.  ``loss_function`` takes the threshold as its last argument, as
    the reference's does."""
    kind = "transformer"

    def __init__(self, input_dim, d_model, num_layers, nhead, latent_dim, future_steps, beta=1.0):
        super().__init__()
        self.encoder = TransformerEncoder(input_dim, d_model, num_layers, nhead, latent_dim)
        self.decoder = TransformerDecoder(latent_dim, future_steps)
        self.beta = beta
        self.threshold = None   # train_step passes model.threshold when it is set
        self.pi_params = nn.Parameter(torch.tensor([3.0, 0.01]))

    def reparameterize(self, mean, logvar, scale, shape, choice=None, eps=None, u=None):
        return F.reparameterize_mixture(mean, logvar, scale, shape, self.pi_params, choice, eps, u)

    def forward(self, x, noise: Noise = None):
        mean, logvar, scale, shape = self.encoder(x, noise)
        n = noise or {}
        z = F.reparameterize_mixture(mean, logvar, scale, shape, self.pi_params, n.get("choice"), n.get("eps"),
                                     n.get("u"))
        return self.decoder(z, noise), mean, logvar, scale, shape

    def loss_function(self, reconstructed, y, mean, logvar, scale, shape, threshold=None):
        thr = self.threshold if threshold is None else threshold
        if thr is None:
            raise ValueError("TransformerMixtureVAE.loss_function needs a threshold")
        return F.mixture_split_elbo(reconstructed, y, mean, logvar, scale, shape, self.pi_params, thr, self.beta)


def build_model(kind: str, input_dim: int, lstm_output_dim: int, latent_dim: int, future_steps: int = 1,
                beta: float = 0.001, threshold: float = 95.0) -> nn.Module:
    if kind == "gaussian":
        return GaussianVAE(input_dim, lstm_output_dim, latent_dim, future_steps, beta)
    if kind == "gpd":
        return GPDVAE(input_dim, lstm_output_dim, latent_dim, future_steps, beta, threshold)
    if kind == "mixture":
        return MixtureVAE(input_dim, lstm_output_dim, latent_dim, future_steps, beta, threshold)
    if kind == "mixture3":
        m = ThreeComponentVAE(input_dim, lstm_output_dim, 0, latent_dim, future_steps, beta)
        m.threshold = threshold
        return m
    raise ValueError(f"unknown model kind {kind!r}")

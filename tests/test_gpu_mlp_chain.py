"""Fused head / decoder chains (vts_mlp_chain_fwd / _bwd) against the same layers in stock PyTorch fp32 on the CPU:
outputs and every gradient, for the chains the models use and for ragged and degenerate shapes."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vae-time-series_amd"))

pytestmark = pytest.mark.gpu

NONE, RELU, EXP, CLAMP, RELU_DROP = 0, 1, 2, 3, 4
TOL = 2e-5   # fp32 products against fp32 products: only the order of the sums differs


def _ref_chain(x, trunk, heads, keeps, scale):
    h = x
    for (w, b, act), keep in zip(trunk, keeps):
        h = torch.nn.functional.linear(h, w, b)
        if act in (RELU, RELU_DROP):
            h = torch.relu(h)
        if act == RELU_DROP and keep is not None:
            h = h * keep * scale
    outs = []
    for w, b, act in heads:
        y = torch.nn.functional.linear(h, w, b)
        if act == EXP:
            y = torch.exp(y)
        elif act == CLAMP:
            y = torch.clamp(y, min=1e-6)
        elif act == RELU:
            y = torch.relu(y)
        outs.append(y)
    return outs


def _rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


CASES = [
    # B, in_dim, [(width, act)], head_dim, [head acts], dropout
    (4096 // 16, 256, [(128, RELU), (64, RELU_DROP)], 32, [NONE, NONE, EXP, CLAMP], True),        # mixture encoder heads
    (100, 64, [(128, RELU), (64, RELU_DROP)], 32, [EXP, CLAMP], True),                            # gpd heads, ragged batch
    (37, 32, [(128, RELU), (500, RELU_DROP)], 1, [NONE], True),                                   # deep decoder, future_steps 1
    (16, 32, [(500, RELU)], 1, [NONE], False),                                                    # gaussian decoder
    (16, 64, [], 32, [NONE, NONE], False),                                                        # gaussian heads: no trunk
    (530, 256, [(128, RELU), (64, RELU_DROP)], 32, [NONE, NONE, EXP, CLAMP, NONE], True),         # three-component heads
    (5, 16, [(20, RELU), (36, NONE), (8, RELU_DROP)], 7, [RELU], True),                           # odd widths, three layers
    (33, 32, [(128, RELU), (500, RELU_DROP)], 4, [NONE], False),                                  # eval mode: no mask
    (20000, 64, [(128, RELU), (64, RELU_DROP)], 32, [EXP, CLAMP], True),                          # weight gradients in three row chunks
]


@pytest.mark.parametrize("B,K,trunk_spec,L,head_acts,drop", CASES)
def test_chain_matches_torch(B, K, trunk_spec, L, head_acts, drop):
    import vts
    from vts import functional as F
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B * 7 + K)
    x = torch.randn(B, K, generator=g)
    dims = [K] + [w for w, _ in trunk_spec]
    trunk, heads, keeps = [], [], []
    for i, (w, act) in enumerate(trunk_spec):
        trunk.append((torch.randn(w, dims[i], generator=g) / dims[i] ** 0.5, torch.randn(w, generator=g) * 0.1, act))
        keeps.append((torch.rand(B, w, generator=g) >= 0.3).float() if (act == RELU_DROP and drop) else None)
    for act in head_acts:
        heads.append((torch.randn(L, dims[-1], generator=g) / dims[-1] ** 0.5, torch.randn(L, generator=g) * 0.1, act))
    wout = [torch.randn(B, L, generator=g) for _ in head_acts]
    scale = 1.0 / (1.0 - 0.3)

    def leaves(ts):
        return [t.clone().requires_grad_(True) for t in ts]

    # reference
    xr = x.clone().requires_grad_(True)
    pr = [(w.clone().requires_grad_(True), b.clone().requires_grad_(True), a) for w, b, a in trunk + heads]
    outs_r = _ref_chain(xr, pr[:len(trunk)], pr[len(trunk):], keeps, scale)
    sum(float(1) * (o * w).sum() for o, w in zip(outs_r, wout)).backward()

    xg = x.to(dev).requires_grad_(True)
    pg = [(w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True), a) for w, b, a in trunk + heads]
    kg = [None if k is None else k.to(dev) for k in keeps]
    assert F.mlp_chain_usable(xg, pg[:len(trunk)], pg[len(trunk):]) == (B <= F.CHAIN_MAX_ROWS)   # the model code's gate
    outs_g = F.mlp_chain(xg, pg[:len(trunk)], pg[len(trunk):], kg, 0.3)
    sum((o * w.to(dev)).sum() for o, w in zip(outs_g, wout)).backward()

    for i, (a, b) in enumerate(zip(outs_g, outs_r)):
        assert _rel(a.detach().cpu(), b.detach()) < TOL, f"head {i}"
    assert _rel(xg.grad.cpu(), xr.grad) < TOL
    for i, ((wg, bg, _), (wr, br, _)) in enumerate(zip(pg, pr)):
        assert _rel(wg.grad.cpu(), wr.grad) < TOL, f"weight {i}"
        assert _rel(bg.grad.cpu(), br.grad) < TOL, f"bias {i}"


def test_models_agree_with_and_without_fused_chains():
    """A whole mixture train step with the chains fused and layer by layer: same loss and gradients."""
    import vts
    dev = torch.device("cuda:0")
    B, T, C, H, L = 64, 6, 8, 64, 32
    results = []
    for fused in (True, False):
        vts.set_fused_chains(fused)
        try:
            torch.manual_seed(3)
            model = vts.build_model("mixture", C, H, L, 1, beta=0.01, threshold=0.5).to(dev)
            model.train()
            with torch.no_grad():      # a shape head away from its clamp: the GPD terms stay finite
                model.encoder.shape_layer_extreme.weight.mul_(0.1)
                model.encoder.shape_layer_extreme.bias.fill_(0.5)
            g = torch.Generator().manual_seed(5)
            x, y = torch.randn(B, T, C, generator=g).to(dev), torch.randn(B, 1, generator=g).to(dev)
            noise = {"enc_keep": (torch.rand(B, 64, generator=g) >= 0.3).float().to(dev),
                     "dec_keep": (torch.rand(B, 500, generator=g) >= 0.3).float().to(dev),
                     "choice": torch.rand(B, generator=g).to(dev), "eps": torch.randn(B, L, generator=g).to(dev),
                     "u": torch.rand(B, L, generator=g).to(dev)}
            out = model(x, noise)
            loss = model.loss_function(out[0], y, *out[1:])
            loss.backward()
            results.append((loss.item(), {n: p.grad.clone() for n, p in model.named_parameters()}))
        finally:
            vts.set_fused_chains(True)
    (l0, g0), (l1, g1) = results
    assert l1 == l1 and abs(l0 - l1) <= 1e-5 * abs(l1)
    for n in g0:
        assert _rel(g0[n], g1[n]) < 1e-4, n


def test_gather_chunks_and_flat_gradients():
    """vts_gather_chunks (gradient tensors -> flat buffer: aligned, unaligned and missing sources) and the train-step
    path that uses it: after a step the flat gradient equals the parameters' .grad, slot by slot."""
    import ctypes as C
    import vts
    from vts import _lib as L
    dev = torch.device("cuda:0")
    lib = L.load()
    g = torch.Generator().manual_seed(0)
    big = torch.randn(5000, generator=g).to(dev)
    srcs = [torch.randn(1031, generator=g).to(dev), None, big[1:1 + 777], torch.randn(4, generator=g).to(dev), big[8:8 + 64]]
    sizes = [1031, 50, 777, 4, 64]
    offs, o = [], 8
    for n in sizes:
        offs.append(o)
        o += (n + 3) // 4 * 4 + 4
    dst = torch.full((o,), 7.0, device=dev)
    n = len(srcs)
    L.check(lib.vts_gather_chunks(L.ptr_array(srcs), (C.c_int64 * n)(*sizes), (C.c_int64 * n)(*offs), n, L.ptr(dst), L.stream()),
            "vts_gather_chunks")
    want = torch.full((o,), 7.0, device=dev)
    for s, k, off in zip(srcs, sizes, offs):
        want[off:off + k] = 0.0 if s is None else s
    assert torch.equal(dst, want)

    torch.manual_seed(1)
    model = vts.build_model("gaussian", 4, 16, 8, 1).to(dev)
    model.train()
    opt = vts.Adam(vts.FlatParameters(model), lr=0.0)          # lr 0: the step leaves the parameters alone
    x, y = torch.randn(32, 5, 4, device=dev), torch.randn(32, 1, device=dev)
    vts.train_step(model, opt, x, y)
    flat = opt.flat
    assert not flat.pending
    for p, off in zip(flat.params, flat.offsets):
        assert p.grad is not None
        assert torch.equal(flat.grad[off:off + p.numel()], p.grad.reshape(-1))

"""Data-parallel exact loss (vts_elbo_sums / vts_elbo_from_sums) on one GPU: two "ranks" are the two halves of
a batch that hold different numbers of normal / extreme rows.  With the statistics added up over the halves,
each half's loss must be the whole batch's loss and the average of the halves' gradients the whole batch's
gradient (rel 1e-5: fp32 sums in another order); with per-rank means they are not - which the test also shows."""
import pytest
import torch

from oracle import ref_torch as R

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch.device("cuda:0")


def l2_rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def run(model, x, y, noise):
    for p in model.parameters():
        p.grad = None
    outs = model(x, noise)
    loss = model.loss_function(outs[0], y, *outs[1:])
    loss.backward()
    return loss.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters()}


def two_rank_run(model, x, y, noise, exact):
    """-> (losses of the two halves, average of their gradients)."""
    import vts
    h = x.shape[0] // 2
    halves = [(x[:h], y[:h], {k: v[:h] for k, v in noise.items()}), (x[h:], y[h:], {k: v[h:] for k, v in noise.items()})]
    total = None
    if exact:
        seen = []
        vts.set_loss_reduction(lambda s: seen.append(s.clone()), 1.0)     # pass 1: record each half's statistics
        try:
            with torch.no_grad():
                for xs, ys, ns in halves:
                    outs = model(xs, ns)
                    model.loss_function(outs[0], ys, *outs[1:])
        finally:
            vts.set_loss_reduction(None)
        assert len(seen) == 2
        total = seen[0] + seen[1]
        vts.set_loss_reduction(lambda s: s.copy_(total), 2.0)              # pass 2: what an all-reduce would leave
    try:
        res = [run(model, xs, ys, ns) for xs, ys, ns in halves]
    finally:
        vts.set_loss_reduction(None)
    grads = {k: (res[0][1][k] + res[1][1][k]) / 2 for k in res[0][1]}
    return [r[0] for r in res], grads


@pytest.mark.parametrize("kind", ["gaussian", "gpd", "mixture"])
def test_halves_reproduce_the_whole_batch(kind):
    import vts
    dev = _dev()
    B, T, C, H, Ld = 48, 8, 5, 64, 16
    sd = R.init_state(kind, C, H, Ld, 1, seed=3)
    if kind != "gaussian":
        sd["encoder.shape_layer_extreme.weight"] *= 0.25
        sd["encoder.shape_layer_extreme.bias"].fill_(0.5)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, T, C, generator=g).to(dev)
    y = torch.randn(B, 1, generator=g)
    y[: B // 2] *= 0.2            # first half: nearly all rows inside the threshold; second half: about half of them
    y = y.to(dev)
    noise = {k: v.to(dev) for k, v in R.draw_noise(kind, B, Ld, seed=5).items()}
    model = vts.build_model(kind, C, H, Ld, 1, threshold=0.7).to(dev)
    model.load_state_dict({k: v.to(dev) for k, v in sd.items()})
    model.train()
    loss_full, grads_full = run(model, x, y, noise)
    losses, grads = two_rank_run(model, x, y, noise, exact=True)
    for l in losses:
        assert abs(l.item() - loss_full.item()) <= 1e-5 * abs(loss_full.item())
    for k, ref in grads_full.items():
        if k == "pi_params":      # cancelling softmax backward: absolute bound, as in test_gpu_parity.py
            assert (grads[k] - ref).abs().max().item() <= 1e-6 * abs(loss_full.item())
        else:
            assert l2_rel(grads[k], ref) < 1e-5, (k, l2_rel(grads[k], ref))
    if kind == "mixture":         # per-rank means do differ here: the exact path is not a no-op
        _, approx = two_rank_run(model, x, y, noise, exact=False)
        assert l2_rel(approx["decoder.out.weight"], grads_full["decoder.out.weight"]) > 1e-3


def test_split_loss_of_the_transformer_variant():
    """The entry points alone, family 3 (masked means over both classes), on random tensors."""
    import vts
    dev = _dev()
    B, Ld = 64, 8
    g = torch.Generator().manual_seed(6)
    mk = lambda *s: torch.randn(*s, generator=g).to(dev)
    recon, y, mean, logvar, scale, shape = mk(B, 1), mk(B, 1), mk(B, Ld), mk(B, Ld) * 0.3, mk(B, Ld) * 0.3, mk(B, Ld)
    y[: B // 2] *= 0.3
    pi = torch.tensor([0.4, 0.1], device=dev)
    leaves = [t.clone().requires_grad_(True) for t in (recon, mean, logvar, scale, shape, pi)]
    full = vts.mixture_split_elbo(leaves[0], y, *leaves[1:5], leaves[5], 0.6, 1.0)
    full.backward()
    h = B // 2
    seen = []
    vts.set_loss_reduction(lambda s: seen.append(s.clone()), 1.0)
    try:
        for sl in (slice(0, h), slice(h, B)):
            vts.mixture_split_elbo(recon[sl], y[sl], mean[sl], logvar[sl], scale[sl], shape[sl], pi, 0.6, 1.0)
    finally:
        vts.set_loss_reduction(None)
    total = seen[0] + seen[1]
    vts.set_loss_reduction(lambda s: s.copy_(total), 2.0)
    try:
        parts = []
        for sl in (slice(0, h), slice(h, B)):
            lv = [t[sl].clone().requires_grad_(True) for t in (recon, mean, logvar, scale, shape)] + [pi.clone().requires_grad_(True)]
            loss = vts.mixture_split_elbo(lv[0], y[sl], *lv[1:5], lv[5], 0.6, 1.0)
            loss.backward()
            assert abs(loss.item() - full.item()) <= 1e-5 * abs(full.item())
            parts.append(lv)
    finally:
        vts.set_loss_reduction(None)
    for i in range(5):            # each half carries n_ranks x the whole batch's row gradients: the later average restores them
        got = torch.cat([parts[0][i].grad, parts[1][i].grad]) / 2
        assert l2_rel(got, leaves[i].grad) < 1e-5
    assert (((parts[0][5].grad + parts[1][5].grad) / 2) - leaves[5].grad).abs().max().item() <= 1e-5 * abs(full.item())


def test_single_rank_reduction_is_the_plain_loss():
    import vts
    dev = _dev()
    g = torch.Generator().manual_seed(7)
    mk = lambda *s: torch.randn(*s, generator=g).to(dev)
    recon, y, mean, logvar = mk(33, 1), mk(33, 1), mk(33, 8), mk(33, 8) * 0.3
    plain = vts.gaussian_elbo(recon, y, mean, logvar, 0.001)
    vts.set_loss_reduction(lambda s: None, 1.0)
    try:
        same = vts.gaussian_elbo(recon, y, mean, logvar, 0.001)
    finally:
        vts.set_loss_reduction(None)
    assert abs(plain.item() - same.item()) <= 1e-6 * abs(plain.item())

"""GPU parity of the three-component (Gaussian / GPD / Bernoulli) VAE of
vae-models/graph/vae_mixture_graph_model.py (GCN and D-KNN off, as graph/train_ld_2011.py:91 builds it): the
HIP path through the C ABI against fixtures recorded from the reference and against the CPU oracle.
Tolerance: rel 1e-4 on the loss, 1e-4 L2-relative on every output and parameter gradient; d loss / d pi_params
(a softmax backward of nearly cancelling O(loss) terms) is held to an absolute 1e-6 * |loss| + 1e-4 relative."""
import math

import pytest
import torch

from oracle import ref_torch as R

pytestmark = pytest.mark.gpu
TOL = 1e-4
MODES = ["f32", "bf16x3"]
NAMES = ["recon", "mean", "logvar", "scale", "shape", "logits"]


@pytest.fixture
def mode(request):
    import vts
    vts.set_mode(request.param)
    yield request.param
    vts.set_mode("f32")


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch.device("cuda:0")


def l2_rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def run_vts(sd, dims, x, y, noise, threshold, beta, dev, train=True):
    import vts
    B, T, C, H, Ld, fs = dims
    m = vts.ThreeComponentVAE(C, H, 0, Ld, fs, beta=beta).to(dev)
    m.load_state_dict({k: v.to(dev) for k, v in sd.items()}, strict=True)
    m.train(train)
    outs = m(x.to(dev), x.to(dev), None, noise={k: v.to(dev) for k, v in noise.items()})
    loss = m.loss_function(outs[0], y.to(dev), outs[1], outs[2], threshold, outs[3], outs[4], outs[5])
    loss.backward()
    grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in m.named_parameters()}
    return loss, outs, grads


def compare(loss, outs, grads, ref_loss, ref_outs, ref_grads):
    assert abs(loss.item() - ref_loss) <= TOL * abs(ref_loss), (loss.item(), ref_loss)
    for name, o in zip(NAMES, outs):
        assert l2_rel(o, ref_outs[name]) < TOL, name
    for k, ref in ref_grads.items():
        if k == "pi_params":
            d = (grads[k].detach().cpu() - ref).abs().max().item()
            assert d <= 1e-6 * abs(ref_loss) + TOL * ref.abs().max().item(), (k, d)
        elif ref.abs().max().item() == 0.0:
            assert grads[k].abs().max().item() == 0.0, k
        else:
            assert l2_rel(grads[k], ref) < TOL, (k, l2_rel(grads[k], ref))


@pytest.mark.parametrize("mode", MODES, indirect=True)
@pytest.mark.parametrize("tag", ["tiny", "native", "native_defaultpi", "wide_eval", "tiny_gaussian_only"])
def test_golden_loss_and_gradients(golden, tag, mode):
    g = golden("mixture3_" + tag)
    dims = [int(v) for v in g["meta"]["dims"]]
    loss, outs, grads = run_vts(g["sd"], dims, g["x"], g["y"], g["noise"], g["meta"]["threshold"], g["meta"]["beta"],
                                _dev(), train=bool(g["meta"]["train"]))
    compare(loss, outs, grads, float(g["loss"]), g["out"], g["grad"])


@pytest.mark.parametrize("mode", MODES, indirect=True)
def test_golden_default_init_shapes_at_the_clamp(golden, mode):
    """Default init leaves about half the GPD shapes at / just above the 1e-6 clamp, where the sampler
    scale/shape*((1-u)**(-shape)-1) cancels catastrophically in fp32: the reference's own gradients move by up
    to ~1e-2 under a last-bit change of the LSTM output.  As for the two-component models
    (tests/test_gpu_parity.py::test_golden_default_init_gpd) outputs keep the 1e-4 bound and loss and gradients are
    held to 10x the floor measured between the oracle's fused and unrolled recurrences."""
    g = golden("mixture3_native_clamped")
    dims = [int(v) for v in g["meta"]["dims"]]
    loss, outs, grads = run_vts(g["sd"], dims, g["x"], g["y"], g["noise"], g["meta"]["threshold"], g["meta"]["beta"], _dev())
    for name, o in zip(NAMES, outs):
        assert l2_rel(o, g["out"][name]) < TOL, name
    _, _, floor_grads = R.loss_and_grads("mixture3", g["sd"], g["x"], g["y"], g["noise"], threshold=g["meta"]["threshold"],
                                         beta=g["meta"]["beta"], stock=False)
    floor = max(l2_rel(floor_grads[k], ref) for k, ref in g["grad"].items() if k != "pi_params")
    bound = max(10 * floor, 1e-3)
    assert abs(loss.item() - float(g["loss"])) <= bound * abs(float(g["loss"]))
    for k, ref in g["grad"].items():
        if k != "pi_params":
            assert l2_rel(grads[k], ref) < bound, (k, l2_rel(grads[k], ref), bound)


def test_empty_normal_class_gives_nan(golden):
    g = golden("mixture3_tiny_no_normal")
    dims = [int(v) for v in g["meta"]["dims"]]
    loss, _, _ = run_vts(g["sd"], dims, g["x"], g["y"], g["noise"], g["meta"]["threshold"], g["meta"]["beta"], _dev())
    assert math.isnan(loss.item())


@pytest.mark.parametrize("mode", MODES, indirect=True)
@pytest.mark.parametrize("B,T,C,H,Ld", [(64, 24, 370, 256, 32), (33, 7, 5, 64, 8), (256, 12, 11, 64, 32)])
def test_fresh_inputs_against_oracle(B, T, C, H, Ld, mode):
    """Seeded inputs at sizes the oracle finishes in seconds (headline channel / hidden sizes included), with
    zero rows, extreme rows and all three samples picked."""
    sd = R.init_state("mixture3", C, H, Ld, 1, seed=B)
    sd["encoder.shape_layer_extreme.weight"] *= 0.25     # GPD shapes off the 1e-6 clamp (ill-conditioned sampler)
    sd["encoder.shape_layer_extreme.bias"].fill_(0.5)
    sd["pi_params"] = torch.tensor([0.7, 0.4, 0.1])
    g = torch.Generator().manual_seed(B + 1)
    x = torch.randn(B, T, C, generator=g)
    y = torch.randn(B, 1, generator=g)
    y[::7] = 0.0
    noise = R.draw_noise("mixture3", B, Ld, seed=B + 2)
    loss_r, outs_r, grads_r = R.loss_and_grads("mixture3", sd, x, y, noise, threshold=0.8)
    loss, outs, grads = run_vts(sd, (B, T, C, H, Ld, 1), x, y, noise, 0.8, 0.001, _dev())
    compare(loss, outs, grads, loss_r.item(), outs_r, grads_r)


def test_sampling_ops_against_oracle():
    """reparameterize alone at 2^16 rows: value and the four sampling gradients, bit-level draws fed in; the
    uniform Bernoulli path equals torch's comparison."""
    import vts
    dev = _dev()
    B, Ld = 1 << 16, 32
    g = torch.Generator().manual_seed(3)
    mean, logvar = torch.randn(B, Ld, generator=g), torch.randn(B, Ld, generator=g) * 0.5
    scale = torch.exp(torch.randn(B, Ld, generator=g) * 0.3)
    shape = torch.rand(B, Ld, generator=g) * 0.4 + 0.05
    logits = torch.randn(B, Ld, generator=g) * 2
    pi = torch.tensor([0.3, 0.2, 0.1])
    noise = R.draw_noise("mixture3", B, Ld, seed=4)
    leaves = [t.clone().requires_grad_(True) for t in (mean, logvar, scale, shape)]
    zr = R.mixture3_select(R.reparam_gaussian(leaves[0], leaves[1], noise["eps"]), R.reparam_gpd(leaves[2], leaves[3], noise["u"]),
                           noise["bern"], pi, noise["choice"])
    w = torch.randn(B, Ld, generator=g)
    (zr * w).sum().backward()
    dl = [t.to(dev).requires_grad_(True) for t in (mean, logvar, scale, shape)]
    z = vts.reparameterize_mixture3(dl[0], dl[1], dl[2], dl[3], logits.to(dev), pi.to(dev), noise["choice"].to(dev),
                                    noise["eps"].to(dev), noise["u"].to(dev), noise["bern"].to(dev))
    (z * w.to(dev)).sum().backward()
    assert l2_rel(z, zr) < 1e-6
    for a, b in zip(dl, leaves):
        assert l2_rel(a.grad, b.grad) < 1e-5
    # uniform draw compared with sigmoid(logits) on the device
    ub = torch.rand(B, Ld, generator=g)
    lib = vts._lib.load()
    from vts import _lib as L
    zb = torch.empty(B, Ld, device=dev)
    args = [t.to(dev).contiguous() for t in (mean, logvar, scale, shape, logits, ub, noise["eps"], noise["u"],
                                             torch.ones(B), pi)]     # choice = 1: every row takes the Bernoulli sample
    L.check(lib.vts_reparam3_fwd(*[L.ptr(t) for t in args[:6]], 1, *[L.ptr(t) for t in args[6:]], L.ptr(zb), B, Ld, L.stream()),
            "vts_reparam3_fwd")
    expect = (ub < torch.sigmoid(logits)).float()
    assert (zb.cpu() != expect).float().mean().item() < 1e-5      # equal except where sigmoid's last bit decides


def test_unsupported_options_raise():
    import vts
    with pytest.raises(NotImplementedError):
        vts.ThreeComponentVAE(4, 64, 8, 8, 1, use_gcn=True)
    with pytest.raises(NotImplementedError):
        vts.ThreeComponentVAE(4, 64, 0, 8, 1, use_d_knn=True)
    with pytest.raises(NotImplementedError):
        vts.ThreeComponentVAE(4, 64, 0, 8, 2)

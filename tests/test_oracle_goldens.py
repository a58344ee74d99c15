"""The CPU oracle (oracle/ref_torch.py) replayed against fixtures recorded from the
reference implementation itself (tests/golden/make_goldens.py).  Runs without a GPU."""
import math

import pytest
import torch

from oracle import ref_torch as R

SINGLE = [
    "gaussian_tiny", "gaussian_native", "gaussian_wide",
    "gpd_tiny", "gpd_native", "gpd_wide",
    "mixture_tiny", "mixture_native", "mixture_wide",
    "mixture_native_mixedmask", "mixture_tiny_eval", "gpd_tiny_eval",
    "gpd_tiny_clamped", "mixture_native_clamped",
]


def kind_of(name):
    return name.split("_")[0]


def max_rel(a, b):
    """max-norm relative difference of two tensors."""
    denom = b.abs().max().item()
    if denom == 0.0:
        return (a - b).abs().max().item()
    return (a - b).abs().max().item() / denom


WELL_CONDITIONED = [
    "gaussian_tiny", "gaussian_native", "gaussian_wide", "mixture_tiny", "mixture_native",
    "mixture_wide", "mixture_tiny_eval", "gpd_tiny_clamped", "mixture_native_clamped",
    "gpd_native_wellcond", "gpd_wide_wellcond", "mixture_wide_wellcond_mixedmask",
]


def l2_rel(a, b):
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("name", SINGLE + ["gpd_native_wellcond", "gpd_wide_wellcond",
                                           "mixture_wide_wellcond_mixedmask"])
def test_forward_backward_matches_reference(golden, name):
    """Same ATen kernels as the reference (fused oneDNN LSTM): near bit-level agreement."""
    g = golden(name)
    kind = kind_of(name)
    loss, outs, grads = R.loss_and_grads(kind, g["sd"], g["x"], g["y"], g["noise"],
                                         threshold=g["meta"]["threshold"],
                                         beta=g["meta"]["beta"], stock=True)
    assert abs(loss.item() - float(g["loss"])) <= 2e-6 * max(1.0, abs(float(g["loss"])))
    for k, ref in g["out"].items():
        assert max_rel(outs[k], ref) < 1e-5, k
    for k, ref in g["grad"].items():
        assert max_rel(grads[k], ref) < 5e-5, k


@pytest.mark.parametrize("name", WELL_CONDITIONED)
def test_unrolled_lstm_within_1e4(golden, name):
    """The spelled-out recurrence (what the HIP kernels implement) agrees with the
    reference to the north-star tolerance, rel 1e-4 on loss and on every gradient."""
    g = golden(name)
    kind = kind_of(name)
    loss, outs, grads = R.loss_and_grads(kind, g["sd"], g["x"], g["y"], g["noise"],
                                         threshold=g["meta"]["threshold"],
                                         beta=g["meta"]["beta"], stock=False)
    assert abs(loss.item() - float(g["loss"])) <= 1e-4 * abs(float(g["loss"]))
    for k, ref in g["grad"].items():
        assert l2_rel(grads[k], ref) < 1e-4, k


@pytest.mark.parametrize("name", ["gpd_tiny", "gpd_native", "gpd_wide"])
def test_default_init_gpd_is_ill_conditioned(golden, name):
    """With default init about half the GPD shapes sit at or just above the 1e-6 clamp, where
    scale/shape*((1-u)**(-shape)-1) cancels catastrophically in fp32 (gpd-lstm-DE.py:12,32).
    The reference's own gradients then move by far more than 1e-4 under a last-bit change
    of the LSTM output (fused vs unrolled recurrence, both exact fp32 maths).  This test
    records that noise floor; GPU parity tests for these fixtures use it as their bound."""
    g = golden(name)
    _, _, grads = R.loss_and_grads("gpd", g["sd"], g["x"], g["y"], g["noise"], stock=False)
    worst = max(l2_rel(grads[k], ref) for k, ref in g["grad"].items())
    assert 1e-5 < worst < 0.2, worst


def test_empty_normal_mask_is_nan(golden):
    g = golden("mixture_tiny_emptymask")
    assert math.isnan(float(g["loss"]))
    loss, _, _ = R.loss_and_grads("mixture", g["sd"], g["x"], g["y"], g["noise"],
                                  threshold=g["meta"]["threshold"], beta=g["meta"]["beta"])
    assert math.isnan(loss.item())


def test_clamped_shape_gets_no_gradient(golden):
    g = golden("gpd_tiny_clamped")
    assert torch.all(g["out"]["shape"] == 1e-6)
    assert torch.all(g["grad"]["encoder.shape_layer_extreme.weight"] == 0)
    _, outs, grads = R.loss_and_grads("gpd", g["sd"], g["x"], g["y"], g["noise"])
    assert torch.all(outs["shape"] == 1e-6)
    assert torch.all(grads["encoder.shape_layer_extreme.weight"] == 0)


@pytest.mark.parametrize("kind", ["gaussian", "gpd", "mixture", "gpd_wellcond", "gaussian_real"])
def test_adam_trajectory(golden, kind):
    """gaussian_real: the reference's first configuration on its own data - eight shuffled 16-window batches of
    datasets/deep-extrema-data-split.pt through the Gaussian model's loop (gaussian-lstm-DE.py:100-156, lr 5e-3)."""
    g = golden({"gpd_wellcond": "adam_gpd_native_wellcond", "gaussian_real": "adam_gaussian_real_data"}.get(kind, f"adam_{kind}_native"))
    if kind == "gpd_wellcond":   # GPD shapes well above the clamp: held to the bounds of the other models
        kind = "gpd_tight"
    sd = {k: v.clone() for k, v in g["sd0"].items()}
    m = {k: torch.zeros_like(v) for k, v in sd.items()}
    v = {k: torch.zeros_like(v) for k, v in sd.items()}
    steps = g["meta"]["steps"]
    for s in range(steps):
        st = g[f"step{s}"]
        loss, _, grads = R.loss_and_grads(kind.split("_")[0], sd, st["x"], st["y"], st["noise"],
                                          threshold=g["meta"]["threshold"], beta=g["meta"]["beta"])
        # gpd: the ill-conditioned sampler (see above) amplifies last-bit differences of the
        # hand-written Adam update from step to step
        ltol = 1e-3 if kind == "gpd" else 2e-5
        assert abs(loss.item() - float(g["losses"][s])) < ltol * max(1.0, abs(loss.item())), s
        R.adam_step(sd, grads, m, v, s + 1, g["meta"]["lr"])
    for k, ref in g["sdK"].items():
        if kind == "gpd":
            # Adam's first updates are ~lr*sign(g): noisy small gradients may flip sign, so
            # only the hard bound of the update size holds for the ill-conditioned model.
            assert (sd[k] - ref).abs().max().item() <= 2 * g["meta"]["lr"] * steps, k
        else:
            assert max_rel(sd[k], ref) < 2e-4, k


def test_param_shapes_match_reference_state_dict(golden):
    for name in ("gaussian_wide", "gpd_wide", "mixture_wide"):
        g = golden(name)
        B, T, C, H, L, fs = [int(v) for v in g["meta"]["dims"]]
        shapes = R.param_shapes(kind_of(name), C, H, L, fs)
        assert list(shapes) == list(g["sd"]), name
        for k, shp in shapes.items():
            assert tuple(g["sd"][k].shape) == shp, k

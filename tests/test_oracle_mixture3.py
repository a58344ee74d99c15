"""The CPU oracle's three-component model (oracle/ref_torch.py, kind "mixture3") replayed against fixtures
recorded from the reference's own class (tests/golden/make_goldens_mixture3.py).  Runs without a GPU."""
import math

import pytest
import torch

from oracle import ref_torch as R

CASES = ["tiny", "native", "native_defaultpi", "wide_eval", "native_clamped", "tiny_gaussian_only"]


def max_rel(a, b):
    denom = b.abs().max().item()
    return (a - b).abs().max().item() / (denom if denom else 1.0)


@pytest.mark.parametrize("tag", CASES)
def test_forward_backward_matches_reference(golden, tag):
    g = golden("mixture3_" + tag)
    loss, outs, grads = R.loss_and_grads("mixture3", g["sd"], g["x"], g["y"], g["noise"],
                                         threshold=g["meta"]["threshold"], beta=g["meta"]["beta"], stock=True)
    assert abs(loss.item() - float(g["loss"])) <= 2e-6 * max(1.0, abs(float(g["loss"])))
    for k, ref in g["out"].items():
        assert max_rel(outs[k], ref) < 1e-5, k
    for k, ref in g["grad"].items():
        assert max_rel(grads[k], ref) < 5e-5, k


def test_empty_normal_class_gives_nan(golden):
    g = golden("mixture3_tiny_no_normal")
    assert math.isnan(float(g["loss"]))
    loss, _ = R.forward_loss("mixture3", g["sd"], g["x"], g["y"], g["noise"], threshold=g["meta"]["threshold"],
                             beta=g["meta"]["beta"])
    assert math.isnan(loss.item())


def test_state_keys_match_reference(golden):
    g = golden("mixture3_native")
    B, T, C, H, Ld, fs = [int(v) for v in g["meta"]["dims"]]
    assert set(R.param_shapes("mixture3", C, H, Ld, fs)) == set(g["sd"])
    assert {k: tuple(v.shape) for k, v in g["sd"].items()} == R.param_shapes("mixture3", C, H, Ld, fs)

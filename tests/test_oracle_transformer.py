"""The transformer-mixture oracle (oracle/ref_transformer.py) replayed against fixtures recorded
from the reference's own classes (tests/golden/make_goldens_transformer.py).  Runs without a GPU."""
import math

import pytest
import torch

from oracle import ref_transformer as RT

CASES = ["transformer_tiny_eval", "transformer_tiny_train", "transformer_tiny_train_attndrop",
         "transformer_native_train"]


def noise_of(g):
    return {k: (v.float() if k.endswith("_keep") else v) for k, v in g["noise"].items()}


def max_rel(a, b):
    denom = b.abs().max().item()
    return (a - b).abs().max().item() / (denom if denom > 0 else 1.0)


@pytest.mark.parametrize("name", CASES)
def test_forward_backward_matches_reference(golden, name):
    """Spelled-out attention / post-norm layer vs nn.TransformerEncoder: same fp32 arithmetic up to
    the order of a few sums (fused vs un-fused softmax.V), so agreement is ~1e-6, bound 2e-5."""
    g = golden(name)
    B, T, C, d, nhead, layers, L, fs = [int(v) for v in g["meta"]["dims"]]
    loss, outs, grads = RT.loss_and_grads(g["sd"], g["x"], g["y"], noise_of(g), nhead,
                                          threshold=g["meta"]["threshold"], beta=g["meta"]["beta"])
    assert abs(loss.item() - float(g["loss"])) <= 2e-6 * max(1.0, abs(float(g["loss"])))
    for k, ref in g["out"].items():
        assert max_rel(outs[k], ref) < 1e-5, k
    for k, ref in g["grad"].items():
        assert max_rel(grads[k], ref) < 2e-5, k
    # the decoder's dense1 is registered but never used (mixture-vae-transformer.py:60,67-72)
    assert torch.all(grads["decoder.dense1.weight"] == 0)


def test_empty_extreme_selection_is_nan(golden):
    g = golden("transformer_tiny_noextreme")
    assert math.isnan(float(g["loss"]))
    B, T, C, d, nhead, layers, L, fs = [int(v) for v in g["meta"]["dims"]]
    loss, _, _ = RT.loss_and_grads(g["sd"], g["x"], g["y"], noise_of(g), nhead,
                                   threshold=g["meta"]["threshold"], beta=g["meta"]["beta"])
    assert math.isnan(loss.item())


def test_param_shapes_match_reference_state_dict(golden):
    for name in ("transformer_tiny_eval", "transformer_native_train"):
        g = golden(name)
        B, T, C, d, nhead, layers, L, fs = [int(v) for v in g["meta"]["dims"]]
        shapes = RT.param_shapes(C, d, layers, L, fs)
        assert list(shapes) == list(g["sd"]), name
        for k, shp in shapes.items():
            assert tuple(g["sd"][k].shape) == shp, k
        sd = RT.init_state(C, d, layers, L, fs, seed=1)
        assert all(sd[k].shape == g["sd"][k].shape for k in sd)


def test_noise_factory_shapes(golden):
    g = golden("transformer_native_train")
    B, T, C, d, nhead, layers, L, fs = [int(v) for v in g["meta"]["dims"]]
    n = RT.draw_noise(B, T, d, nhead, layers, L, seed=3)
    assert sorted(n) == sorted(g["noise"])
    for k, v in n.items():
        assert tuple(v.shape) == tuple(g["noise"][k].shape), k

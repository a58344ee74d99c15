"""The numpy window oracle against the fixtures recorded from the reference's own data-preparation
functions (tests/golden/make_goldens_windows.py): bit for bit for windows, targets and the
standardised values; 1e-12 relative for the fitted statistics (sklearn sums incrementally)."""
import os

import numpy as np
import pytest

from oracle import ref_windows as RW

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LD_CASES = ["ld2011_t4", "ld2011wide_t24_p3", "sunspots_t12"]


def load(tag):
    return dict(np.load(os.path.join(GOLD, f"windows_{tag}.npz")))


@pytest.mark.parametrize("tag", LD_CASES)
def test_sliding_windows_match_reference(tag):
    g = load(tag)
    X, y = RW.sliding_windows(g["series"], int(g["T"]), int(g["P"]))
    assert X.dtype == np.float32 and np.array_equal(X, g["X_all"])
    assert np.array_equal(y, g["y_all"])


@pytest.mark.parametrize("tag", LD_CASES)
def test_standardised_split_matches_reference(tag):
    g = load(tag)
    T, P = int(g["T"]), int(g["P"])
    for part in ("train", "test"):
        X, y = RW.sliding_windows(g["series"], T, P, g[f"{part}_starts"])
        assert np.array_equal(RW.scaler_transform(X, g["x_mean"], g["x_scale"]), g[f"X_{part}"])
        assert np.array_equal(RW.scaler_transform(y, g["y_mean"], g["y_scale"]), g[f"y_{part}"])


@pytest.mark.parametrize("tag", LD_CASES)
def test_scaler_fit_matches_reference(tag):
    g = load(tag)
    X, y = RW.sliding_windows(g["series"], int(g["T"]), int(g["P"]), g["train_starts"])
    mean, scale = RW.scaler_fit(X.reshape(-1, X.shape[2]))
    np.testing.assert_allclose(mean, g["x_mean"], rtol=1e-12, atol=1e-12 * np.abs(g["x_scale"]).max())
    np.testing.assert_allclose(scale, g["x_scale"], rtol=1e-12)
    ym, ys = RW.scaler_fit(y.reshape(-1, 1))
    np.testing.assert_allclose(ym, g["y_mean"], rtol=1e-12, atol=1e-12 * float(g["y_scale"][0]))
    np.testing.assert_allclose(ys, g["y_scale"], rtol=1e-12)


def test_block_maximum_matches_reference():
    g = load("blockmax_t16_p8")
    T, P = int(g["T"]), int(g["P"])
    X, _ = RW.sliding_windows(g["series"], T, P)
    assert np.array_equal(X, g["X_all"])
    assert np.array_equal(RW.block_max_targets(g["series"], T, P), g["y_all"])


def test_constant_column_gets_unit_scale():
    rows = np.ones((50, 3), dtype=np.float32) * np.array([2.0, -1.0, 0.0], dtype=np.float32)
    rows[:, 1] += np.arange(50, dtype=np.float32)
    mean, scale = RW.scaler_fit(rows)
    assert scale[0] == 1.0 and scale[2] == 1.0 and scale[1] > 1.0
    assert np.array_equal(RW.scaler_transform(rows, mean, scale)[:, 0], np.zeros(50, dtype=np.float32))

"""GPU parity of the window builder (vts_windows_build / vts_windows_stats through the C ABI) against the
fixtures recorded from the reference's data-preparation functions and against the numpy oracle.
Bar: bit for bit for windows, targets and standardised values (gathers and correctly rounded fp64
arithmetic); 1e-12 relative for the fitted statistics (a floating-point sum in another order)."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_windows as RW

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LD_CASES = ["ld2011_t4", "ld2011wide_t24_p3", "sunspots_t12"]


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch.device("cuda:0")


def load(tag):
    return dict(np.load(os.path.join(GOLD, f"windows_{tag}.npz")))


def same_bits(t: torch.Tensor, ref: np.ndarray) -> bool:
    return np.array_equal(t.cpu().numpy().view(np.uint32), np.ascontiguousarray(ref, dtype=np.float32).view(np.uint32))


@pytest.mark.parametrize("tag", LD_CASES)
def test_all_windows_match_reference(tag):
    import vts
    g = load(tag)
    src = vts.WindowSource(torch.from_numpy(g["series"]).to(_dev()), int(g["T"]), int(g["P"]))
    assert len(src) == len(g["X_all"])
    X, y = src.batch()
    assert same_bits(X, g["X_all"]) and same_bits(y, g["y_all"])
    X2, y2 = vts.create_sliding_windows(torch.from_numpy(g["series"]).to(_dev()), int(g["T"]), int(g["P"]))
    assert same_bits(X2, g["X_all"]) and y2.shape == (g["y_all"].size,)


@pytest.mark.parametrize("tag", LD_CASES)
def test_standardised_split_matches_reference(tag):
    import vts
    g = load(tag)
    src = vts.WindowSource(torch.from_numpy(g["series"]).to(_dev()), int(g["T"]), int(g["P"]))
    src.set_scalers(g["x_mean"], g["x_scale"], g["y_mean"][0], g["y_scale"][0])
    for part in ("train", "test"):
        X, y = src.batch(g[f"{part}_starts"])
        assert same_bits(X, g[f"X_{part}"]), part
        assert same_bits(y, g[f"y_{part}"]), part


@pytest.mark.parametrize("tag", LD_CASES)
def test_fitted_statistics_match_reference(tag):
    import vts
    g = load(tag)
    src = vts.WindowSource(torch.from_numpy(g["series"]).to(_dev()), int(g["T"]), int(g["P"]))
    src.fit(g["train_starts"])
    xs, ys = src.x_stats.cpu().numpy(), src.y_stats.cpu().numpy()
    np.testing.assert_allclose(xs[0], g["x_mean"], rtol=1e-12, atol=1e-12 * np.abs(g["x_scale"]).max())
    np.testing.assert_allclose(xs[1], g["x_scale"], rtol=1e-12)
    np.testing.assert_allclose(ys[0], g["y_mean"][0], rtol=1e-12, atol=1e-12 * float(g["y_scale"][0]))
    np.testing.assert_allclose(ys[1], g["y_scale"][0], rtol=1e-12)
    # statistics that close give the reference's standardised windows to the last float32 bit or the one next to it
    X, _ = src.batch(g["train_starts"])
    d = np.abs(X.cpu().numpy().view(np.int32).astype(np.int64) - g["X_train"].view(np.int32).astype(np.int64))
    assert d.max() <= 1


def test_block_maximum_matches_reference():
    import vts
    g = load("blockmax_t16_p8")
    src = vts.WindowSource(torch.from_numpy(g["series"]).to(_dev()), int(g["T"]), int(g["P"]), target="block_max")
    X, y = src.batch()
    assert same_bits(X, g["X_all"]) and same_bits(y, g["y_all"])


def test_block_maximum_propagates_nan_and_edges_read_as_nan():
    import vts
    s = np.arange(40, dtype=np.float32)[:, None].repeat(3, 1)
    s[20, 0] = np.nan
    src = vts.WindowSource(torch.from_numpy(s).to(_dev()), 4, 3, target="block_max")
    starts = np.array([0, 14, 16, 33, 34, -1, 37])
    X, y = src.batch(starts)
    ref = RW.block_max_targets(s, 4, 3, np.array([0, 14, 16, 33]))
    got = y.cpu().numpy()
    assert np.array_equal(got[:4], ref, equal_nan=True) and np.isnan(got[1, 0]) and np.isnan(got[2, 0])
    assert np.isnan(got[4:]).all()                         # targets past the end / negative start
    Xc = X.cpu().numpy()
    assert np.array_equal(Xc[4], s[34:38]) and np.isnan(Xc[5]).all() and np.isnan(Xc[6]).all()


@pytest.mark.parametrize("Nt,C,T,P,B", [(5000, 370, 96, 1, 512), (3000, 7, 5, 2, 4096), (64, 1, 1, 1, 63), (900, 33, 13, 4, 257)])
def test_random_series_against_oracle(Nt, C, T, P, B):
    """Fresh seeded series at sizes the oracle finishes in seconds, LD2011 channel count included;
    channel scales spread over twelve decades so the fp64 quotient is exercised well away from 1."""
    import vts
    rng = np.random.default_rng(Nt + C)
    series = (rng.standard_normal((Nt, C)) * 10.0 ** rng.uniform(-6, 6, size=C) + rng.standard_normal(C)).astype(np.float32)
    starts = rng.integers(0, Nt - T - P + 1, size=B)
    X_ref, y_ref = RW.sliding_windows(series, T, P, starts)
    mean, scale = RW.scaler_fit(X_ref.reshape(-1, C))
    ym, ys = RW.scaler_fit(y_ref.reshape(-1, 1))
    src = vts.WindowSource(torch.from_numpy(series).to(_dev()), T, P)
    X, y = src.batch(starts)
    assert same_bits(X, X_ref) and same_bits(y, y_ref)
    src.set_scalers(mean, scale, ym[0], ys[0])
    X, y = src.batch(starts)
    assert same_bits(X, RW.scaler_transform(X_ref, mean, scale))
    assert same_bits(y, RW.scaler_transform(y_ref, ym, ys))
    src.fit(starts)
    np.testing.assert_allclose(src.x_stats.cpu().numpy(), np.stack([mean, scale]), rtol=1e-11, atol=1e-11 * scale.max())


def test_full_size_batch_properties():
    """The bench shape (4096 windows x 96 steps x 370 channels): every window equals its run of series
    rows (checked on the device against an index gather), standardisation is the per-channel affine map."""
    import vts
    dev = _dev()
    torch.manual_seed(0)
    series = torch.randn(5000, 370, device=dev) * 3 + 1
    src = vts.WindowSource(series, 96, 1)
    starts = torch.randint(0, len(src), (4096,), device=dev)
    X, y = src.batch(starts)
    rows = starts[:, None] + torch.arange(96, device=dev)[None, :]
    assert torch.equal(X, series[rows])
    assert torch.equal(y[:, 0], series[starts + 96, 0])
    src.fit(starts)
    Xs, _ = src.batch(starts)
    m = Xs.reshape(-1, 370).double().mean(0)
    v = Xs.reshape(-1, 370).double().var(0, unbiased=False)
    assert m.abs().max().item() < 1e-5 and (v - 1).abs().max().item() < 1e-4


def test_bad_arguments_raise():
    import vts
    src = vts.WindowSource(torch.zeros(10, 2, device=_dev()), 3, 1)
    src.set_scalers(np.zeros(2), np.ones(2))
    with pytest.raises(ValueError):
        src.set_scalers(np.zeros(3), np.ones(3))
    with pytest.raises(RuntimeError):
        vts.WindowSource(torch.zeros(10, 2), 3, 1)

"""CPU-side checks: the C ABI library loads and exports every symbol include/vts.h declares, the
Python face mirrors the reference's module structure, and host-side helpers behave.  No GPU."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "vts.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vts_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from vts import _lib
    lib = _lib.load()
    names = header_symbols()
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/vts.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes signatures and header disagree"
    assert b"gfx950" in lib.vts_version()


def test_sizing_calls_do_not_need_a_gpu():
    from vts import _lib
    lib = _lib.load()
    B, T, H = 4096, 96, 256
    assert lib.vts_lstm_gates_bytes(B, T, H) == B * T * 4 * H * 4
    assert lib.vts_lstm_cells_bytes(20, 3, 16) == 32 * 3 * 16 * 4      # batch padded to 16
    assert lib.vts_lstm_workspace_bytes(B, T, 370, H) > 4 * H * H * 4
    assert lib.vts_elbo_workspace_bytes(B, 1, 32) >= 5 * 4


def test_paired_weight_gradient_plan():
    """Which layers get both weight gradients from one launch, and that the shared workspace covers every plan
    (pure host logic: no launch happens for an unsupported shape)."""
    from vts import _lib
    lib = _lib.load()
    assert lib.vts_lstm_wgrad_pair_supported(4096, 96, 256, 256, 1) == 1       # layers above the first
    assert lib.vts_lstm_wgrad_pair_supported(4096, 96, 370, 256, 1) == 1       # first layer: 256 columns paired
    assert lib.vts_lstm_wgrad_pair_supported(4096, 96, 256, 256, 0) == 0       # fp32 products keep two launches
    assert lib.vts_lstm_wgrad_pair_supported(16, 16, 1, 64, 1) == 0            # the reference's own sizes
    assert lib.vts_lstm_wgrad_pair_supported(4096, 96, 64, 256, 1) == 0        # narrow input
    # slabs of the paired launch: 16 k-slices x [1024, 512] floats
    assert lib.vts_lstm_wgrad_workspace_bytes(4096, 96, 256, 256) >= 16 * 1024 * 512 * 4
    rc = lib.vts_lstm_wgrad_pair(None, None, None, None, None, 0, None, 0, 16, 16, 1, 64, 1, None)
    assert rc != 0


def test_argument_errors_are_reported_not_raised_in_c():
    from vts import _lib
    lib = _lib.load()
    rc = lib.vts_lstm_inproj_fwd(None, None, None, None, None, 16, 4, 3, 48, 0, None)
    assert rc == 3 and b"hidden size 48" in lib.vts_last_error()
    rc = lib.vts_linear_fwd(None, 4, None, None, None, 4, 4, 4, 0, None, 1.0, None, 7, None)
    assert rc == 3 and b"dtype_mode" in lib.vts_last_error()
    rc = lib.vts_adam_step(None, None, None, None, 10, 1, 1e-3, 0.9, 0.999, 1e-8, 1.0, None)
    assert rc == 1


def test_cpu_tensors_are_refused():
    import vts
    m = vts.build_model("gaussian", 1, 16, 8)
    with pytest.raises(RuntimeError, match="ROCm device only"):
        m(torch.zeros(2, 4, 1))
    with pytest.raises(RuntimeError):
        vts.gaussian_elbo(torch.zeros(2, 1), torch.zeros(2, 1), torch.zeros(2, 8), torch.zeros(2, 8))


@pytest.mark.parametrize("kind", ["gaussian", "gpd", "mixture"])
def test_state_dict_matches_reference_fixture(golden, kind):
    import vts
    g = golden(f"{kind}_wide")
    B, T, C, H, L, fs = [int(v) for v in g["meta"]["dims"]]
    m = vts.build_model(kind, C, H, L, fs)
    assert list(m.state_dict()) == list(g["sd"])
    m.load_state_dict(g["sd"], strict=True)


def test_lstm_parameters_match_nn_lstm():
    import vts
    a = vts.LSTM(7, 32, num_layers=3)
    b = torch.nn.LSTM(7, 32, num_layers=3, batch_first=True)
    assert [(k, tuple(v.shape)) for k, v in a.state_dict().items()] == \
           [(k, tuple(v.shape)) for k, v in b.state_dict().items()]
    bound = 1 / 32 ** 0.5
    assert all(p.abs().max() <= bound for p in a.parameters())


def test_flat_parameters_alias_and_zero_grad():
    import vts
    m = vts.build_model("mixture", 3, 16, 8)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    flat = vts.FlatParameters(m)
    assert flat.numel >= sum(p.numel() for p in m.parameters())
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k])
    for p, off in zip(flat.params, flat.offsets):
        assert p.data_ptr() == flat.data.data_ptr() + 4 * off and off % 4 == 0
        assert p.grad.data_ptr() == flat.grad.data_ptr() + 4 * off
    flat.grad.fill_(1.0)
    flat.zero_grad()
    assert all(float(p.grad.abs().sum()) == 0 for p in flat.params)
    flat.data.zero_()
    assert all(float(p.detach().abs().sum()) == 0 for p in m.parameters())


def test_bucket_plan_covers_the_flat_buffer():
    import vts
    from vts.parallel import plan_buckets
    flat = vts.FlatParameters(vts.build_model("mixture", 370, 256, 32))
    buckets = plan_buckets(flat, max_buckets=3)
    assert 1 <= len(buckets) <= 3
    spans = sorted(buckets)
    assert spans[0][0] == 0 and spans[-1][1] == flat.numel
    assert all(spans[i][1] == spans[i + 1][0] for i in range(len(spans) - 1))
    assert buckets[0][1] == flat.numel          # last range first: backward finishes it first
    small = vts.FlatParameters(vts.build_model("gaussian", 1, 16, 8))
    assert plan_buckets(small) == [(0, small.numel)]


def test_shard_batch():
    import vts
    x = torch.arange(24).reshape(8, 3)
    assert torch.equal(vts.shard_batch(x, 1, 4), x[2:4])
    with pytest.raises(ValueError):
        vts.shard_batch(x, 0, 3)


def test_noise_draw_order_matches_reference(golden):
    """oracle.draw_noise keys/shapes are those the reference consumed (fixture noise)."""
    from oracle import ref_torch as R
    for kind in ("gaussian", "gpd", "mixture"):
        g = golden(f"{kind}_native")
        B, T, C, H, L, fs = [int(v) for v in g["meta"]["dims"]]
        n = R.draw_noise(kind, B, L, seed=0)
        assert set(n) == set(g["noise"])
        for k in n:
            assert n[k].shape == g["noise"][k].shape


def test_recurrent_ring_registers_are_never_touched_in_flight():
    """The bf16x3 recurrent kernels fill their W_hh register ring with inline-asm loads and wait
    for them with hand-counted s_waitcnt; tools/check_ring_registers.py re-derives from the
    generated gfx950 ISA that no instruction touches a ring register while its load can still be
    outstanding (hipcc cross-compiles without a GPU)."""
    import subprocess
    import sys
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not installed")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_ring_registers.py")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]

"""The LSTM layer's plane path (csrc/gemm_planes.h, csrc/lstm_planes.hip): sequences kept as bf16 head/tail planes
between the recurrent kernels and the three big products.  Taken in bf16x3 mode for hidden 256, batches that are
multiples of 128 and even step counts.  Oracle: torch's CPU nn.LSTM (what the reference calls:
vae-models/mixture-vae-DE.py:45,60), tolerance L2-relative 1e-4 on outputs and every gradient, as everywhere."""
import pytest
import torch


def l2_rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _dev():
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _bf16x3():
    import vts
    old = vts.get_mode()
    vts.set_mode("bf16x3")
    yield
    vts.set_mode(old)


def _planes_taken(B, T, C, H):
    from vts import _lib
    return bool(_lib.load().vts_lstm_planes_supported(B, T, C, H, 1))


@pytest.mark.parametrize("B,T,C,H,layers", [
    (128, 2, 8, 256, 1),      # one cluster, one weight-gradient k-step per window tile
    (128, 6, 370, 256, 3),    # LD2011 channel count: input planes padded to 384, three column tiles of weight gradients
    (256, 16, 1, 256, 2),     # a single input channel
    (384, 4, 64, 256, 2),     # 96 token tiles: output tiles spread unevenly over the workgroups
    (1024, 8, 256, 256, 3),   # 512 token tiles: more than one output tile per workgroup, 32 k-slices
])
@pytest.mark.parametrize("need_output,bwd_kernel", [(True, "pair"), (False, "pair"), (True, "staged"), (False, "staged"),
                                                    (True, "direct"), (False, "direct"), (False, "cluster")])
def test_plane_path_matches_torch_lstm(B, T, C, H, layers, need_output, bwd_kernel):
    """The four backward recurrence kernels of this shape: "pair" (default: two workgroups share 32 windows and half
    of W_hh each, partial sums of dh handed over every step), "staged" (one workgroup per 16 windows, saved tensors
    through LDS a step ahead, part of W_hh resident in registers: vts_debug_set_pair_bwd(0)), "direct" (saved tensors
    loaded into registers at the top of a step's product: vts_debug_set_staged_bwd(0)), "cluster" (clusters of eight
    workgroups that exchange the gate gradients through the plane buffer: vts_debug_set_cluster_bwd(1))."""
    import vts
    from vts import _lib
    assert _planes_taken(B, T, C, H)
    _lib.load().vts_debug_set_pair_bwd(1 if bwd_kernel == "pair" else 0)
    _lib.load().vts_debug_set_cluster_bwd(1 if bwd_kernel == "cluster" else 0)
    _lib.load().vts_debug_set_staged_bwd(0 if bwd_kernel == "direct" else 1)
    try:
        _run_plane_case(B, T, C, H, layers, need_output)
    finally:
        _lib.load().vts_debug_set_pair_bwd(1)
        _lib.load().vts_debug_set_cluster_bwd(0)
        _lib.load().vts_debug_set_staged_bwd(1)


def _run_plane_case(B, T, C, H, layers, need_output):
    import vts
    dev = _dev()
    g = torch.Generator().manual_seed(B * 1000 + T + C)
    lstm = vts.LSTM(C, H, num_layers=layers).to(dev)
    ref = torch.nn.LSTM(C, H, num_layers=layers, batch_first=True)
    ref.load_state_dict({k: v.detach().cpu() for k, v in lstm.state_dict().items()})
    x = torch.randn(B, T, C, generator=g)
    w_out = torch.randn(B, T, H, generator=g)
    w_h = torch.randn(layers, B, H, generator=g)
    w_c = torch.randn(layers, B, H, generator=g)

    out_r, (hn_r, cn_r) = ref(x)
    loss_r = (hn_r * w_h).sum() + (cn_r * w_c).sum()
    if need_output:
        loss_r = loss_r + (out_r * w_out).sum()
    loss_r.backward()

    out, (hn, cn) = lstm(x.to(dev), need_output=need_output)
    loss = (hn * w_h.to(dev)).sum() + (cn * w_c.to(dev)).sum()
    if need_output:
        loss = loss + (out * w_out.to(dev)).sum()
        assert l2_rel(out, out_r) < TOL
    else:
        assert out.numel() == 0
    loss.backward()
    assert l2_rel(hn, hn_r) < TOL
    assert l2_rel(cn, cn_r) < TOL
    for k, p in lstm.named_parameters():
        assert l2_rel(p.grad, getattr(ref, k).grad) < TOL, k


def test_plane_path_input_gradient():
    """Input width 256: the first layer's data gradient runs on the plane path too."""
    import vts
    B, T, C, H = 128, 4, 256, 256
    assert _planes_taken(B, T, C, H)
    dev = _dev()
    g = torch.Generator().manual_seed(11)
    lstm = vts.LSTM(C, H, num_layers=2).to(dev)
    ref = torch.nn.LSTM(C, H, num_layers=2, batch_first=True)
    ref.load_state_dict({k: v.cpu() for k, v in lstm.state_dict().items()})
    x = torch.randn(B, T, C, generator=g)
    xr = x.clone().requires_grad_(True)
    xd = x.to(dev).requires_grad_(True)
    ref(xr)[1][0][-1].pow(2).sum().backward()
    lstm(xd)[1][0][-1].pow(2).sum().backward()
    assert l2_rel(xd.grad, xr.grad) < TOL
    for k, p in lstm.named_parameters():
        assert l2_rel(p.grad, getattr(ref, k).grad) < TOL, k


def test_second_backward_on_the_plane_path():
    """The plane path leaves the saved gate activations alone, so a retained graph can be differentiated twice."""
    import vts
    dev = _dev()
    lstm = vts.LSTM(8, 256).to(dev)
    x = torch.randn(128, 4, 8, device=dev)
    _, (hn, _) = lstm(x, need_output=False)
    loss = hn.pow(2).sum()
    loss.backward(retain_graph=True)
    first = [p.grad.clone() for p in lstm.parameters()]
    for p in lstm.parameters():
        p.grad = None
    loss.backward()
    for a, p in zip(first, lstm.parameters()):
        assert torch.equal(a, p.grad)


def test_second_backward_off_the_plane_path_raises():
    import vts
    dev = _dev()
    lstm = vts.LSTM(8, 64).to(dev)
    x = torch.randn(16, 4, 8, device=dev)
    _, (hn, _) = lstm(x)
    loss = hn.pow(2).sum()
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="second backward"):
        loss.backward()


def _pair_case(fault):
    """One backward pass of a 2-layer LSTM at a shape the pair kernel takes, with the pair kernel's test hook set."""
    import vts
    from vts import _lib
    lib = _lib.load()
    dev = _dev()
    torch.manual_seed(3)
    lstm = vts.LSTM(8, 256, num_layers=2).to(dev)
    x = torch.randn(128, 6, 8, device=dev)
    lib.vts_debug_set_pair_fault(fault)
    try:
        out, (hn, _) = lstm(x)
        (out.pow(2).sum() + hn.pow(2).sum()).backward()
        torch.cuda.synchronize()
    finally:
        lib.vts_debug_set_pair_fault(-1)
    return {k: p.grad.clone() for k, p in lstm.named_parameters()}


def test_pair_backward_with_a_missing_peer():
    """A peer that never delivers its partial sums turns the gradients into NaN instead of hanging, and the next launch
    is clean again."""
    good = _pair_case(-1)
    for k in good:
        assert torch.isfinite(good[k]).all(), k
    bad = _pair_case(1)
    assert torch.isnan(bad["weight_hh_l1"]).any(), "a missed hand-off must poison the gradients"
    again = _pair_case(-1)
    for k in good:
        assert torch.equal(good[k], again[k]), k


@pytest.fixture()
def fused_forward():
    """The layer's forward pass as one kernel (vts_lstm_fused_fwd_planes) instead of input projection + recurrence."""
    from vts import _lib
    lib = _lib.load()
    lib.vts_debug_set_fused_fwd(1)
    yield lib
    lib.vts_debug_set_fused_fwd(0)
    lib.vts_debug_set_fused_fault(-1)


@pytest.mark.parametrize("B,T,C,H,layers", [
    (128, 2, 8, 256, 1),      # one cluster; input planes two column blocks wide
    (128, 6, 370, 256, 3),    # twelve column blocks (first layer), eight (the layers above it)
    (1024, 8, 256, 256, 2),   # eight clusters, eight column blocks in both layers
])
@pytest.mark.parametrize("need_output", [True, False])
def test_fused_forward_matches_torch_lstm(fused_forward, B, T, C, H, layers, need_output):
    """Outputs, final states and every gradient (the backward kernels read the gate activations and cell states the
    fused kernel saved) against torch's CPU nn.LSTM, at the bound of the two-kernel path."""
    lib = fused_forward
    assert lib.vts_lstm_fused_fwd_supported(B, T, C, H, 1)
    import vts
    vts.recurrence_status.begin_step()
    _run_plane_case(B, T, C, H, layers, need_output)
    names = [n for n, _ in vts.recurrence_status.read()]
    assert sum("fused forward" in n for n in names) == layers, names      # ... and the fused kernel is what ran


def test_fused_forward_against_the_two_kernel_path():
    """Same inputs through both forward paths: final states agree to 1e-5 (the fused kernel hands h on as a bf16 head
    plus tail, 2^-17 relative, and sums in another order), and a width it is not built for (100 features) takes the
    two-kernel path without complaint."""
    import vts
    from vts import _lib
    lib = _lib.load()
    dev = _dev()
    torch.manual_seed(21)
    lstm = vts.LSTM(256, 256, num_layers=2).to(dev)
    x = torch.randn(256, 10, 256, device=dev)
    _, (h0, c0) = lstm(x, need_output=False)
    lib.vts_debug_set_fused_fwd(1)
    try:
        _, (h1, c1) = lstm(x, need_output=False)
        assert not lib.vts_lstm_fused_fwd_supported(256, 10, 100, 256, 1)
        odd = vts.LSTM(100, 256).to(dev)
        _, (h2, _) = odd(torch.randn(256, 10, 100, device=dev), need_output=False)
        assert torch.isfinite(h2).all()
    finally:
        lib.vts_debug_set_fused_fwd(0)
    assert l2_rel(h1, h0) < 1e-5 and l2_rel(c1, c0) < 1e-5


def test_fused_forward_with_a_missing_peer(fused_forward):
    """A member that never raises its flags: the launch ends with NaN outputs and a non-zero status word, and
    vts.recurrence_status names the kernel; the next launch is clean."""
    import vts
    lib = fused_forward
    dev = _dev()
    torch.manual_seed(4)
    lstm = vts.LSTM(8, 256).to(dev)
    x = torch.randn(128, 4, 8, device=dev)
    vts.recurrence_status.begin_step()
    _, (good, _) = lstm(x, need_output=False)
    assert all(v == 0 for _, v in vts.recurrence_status.read())
    lib.vts_debug_set_fused_fault(3)
    vts.recurrence_status.begin_step()
    _, (bad, _) = lstm(x, need_output=False)
    torch.cuda.synchronize()
    lib.vts_debug_set_fused_fault(-1)
    st = vts.recurrence_status.read()
    assert any("fused forward" in n and v != 0 for n, v in st), st
    assert torch.isnan(bad).any()
    vts.recurrence_status.begin_step()
    _, (again, _) = lstm(x, need_output=False)
    assert torch.equal(again, good)
    assert all(v == 0 for _, v in vts.recurrence_status.read())


# ---- forward cluster kernel with one ring of waves per window tile (vts_debug_set_cluster_rings) --------------------
def _ring_case(rings, B, T, C, layers, need_output, fault=-1, odd_route=False):
    """Forward + backward of a vts.LSTM whose forward recurrence runs on the chosen cluster kernel.  odd_route: fp32
    rows in and out (no plane output; an odd step count keeps the layer off the plane path), so that the members
    exchange h through the two alternating exchange buffers instead of the plane buffer."""
    import vts
    from vts import _lib
    lib = _lib.load()
    dev = _dev()
    torch.manual_seed(17)
    lstm = vts.LSTM(C, 256, num_layers=layers).to(dev)
    x = torch.randn(B, T, C, device=dev)
    assert _planes_taken(B, T, C, 256) != odd_route
    lib.vts_debug_set_cluster_rings(rings)
    lib.vts_debug_set_cluster_fault(fault)
    try:
        out, (hn, cn) = lstm(x, need_output=need_output)
        loss = hn.pow(2).sum() + (cn * 0.5).sum() + (out.pow(2).sum() if need_output else 0.0)
        loss.backward()
        torch.cuda.synchronize()
    finally:
        lib.vts_debug_set_cluster_rings(0)
        lib.vts_debug_set_cluster_fault(-1)
    res = {"hn": hn.detach().clone(), "cn": cn.detach().clone(), "out": out.detach().clone()}
    res.update({k: p.grad.clone() for k, p in lstm.named_parameters()})
    return res, lstm, x


@pytest.mark.parametrize("B,T,C,layers,need_output,odd_route", [
    (128, 2, 8, 1, True, False),       # one cluster, the shortest sequence that has a hand-off
    (256, 16, 370, 3, False, False),   # two clusters, three layers, plane output only
    (1024, 8, 256, 2, True, False),    # eight clusters = one per XCD
    (128, 7, 8, 2, True, True),        # odd step count: the tail step of the two-step loop, exchange buffers
    (384, 5, 64, 1, False, True),
])
def test_wave_ring_forward_is_bit_identical(B, T, C, layers, need_output, odd_route):
    """Both forward cluster kernels do the same arithmetic in the same order - same fragments, same accumulation
    order, same cell update - so every output and, through the saved tensors, every gradient must be bit-identical;
    and the ring kernel's result is held to torch's CPU nn.LSTM like everything else."""
    base, _, _ = _ring_case(0, B, T, C, layers, need_output, odd_route=odd_route)
    ring, lstm, x = _ring_case(1, B, T, C, layers, need_output, odd_route=odd_route)
    for k in base:
        assert torch.equal(base[k], ring[k]), k
    ref = torch.nn.LSTM(C, 256, num_layers=layers, batch_first=True)
    ref.load_state_dict({k: v.detach().cpu() for k, v in lstm.state_dict().items()})
    out_r, (hn_r, cn_r) = ref(x.cpu())
    assert l2_rel(ring["hn"], hn_r) < TOL and l2_rel(ring["cn"], cn_r) < TOL
    if need_output:
        assert l2_rel(ring["out"], out_r) < TOL


def test_wave_ring_forward_with_a_missing_peer():
    """A member whose waves never raise their flags: every ring gives up after its bounded wait, the final state is NaN
    and the status word is set; the next launch is clean."""
    import vts
    good, _, _ = _ring_case(1, 128, 6, 8, 2, True)
    vts.raise_on_recurrence_timeout()            # clean so far
    bad, _, _ = _ring_case(1, 128, 6, 8, 2, True, fault=3)
    assert torch.isnan(bad["hn"]).any(), "a missed hand-off must poison the final state"
    with pytest.raises(vts.RecurrenceTimeout, match="lstm_rec3c_fwd_kernel"):
        vts.raise_on_recurrence_timeout()
    again, _, _ = _ring_case(1, 128, 6, 8, 2, True)
    vts.raise_on_recurrence_timeout()
    for k in good:
        assert torch.equal(good[k], again[k]), k

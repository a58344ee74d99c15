"""vts_elbo_fwd_static (the entry point that allocates nothing: block partials in a device array of the code
object) against vts_elbo_fwd (caller workspace), through the C ABI.  Bar: bit for bit, both are the same two
kernels with the same launch shape.  Two calls in a row, the second with a smaller input, so that stale
partials of the first call would show; then the size limit."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(B, fs, Ld, seed):
    g = torch.Generator().manual_seed(seed)
    dev = torch.device("cuda:0")
    t = lambda *s: torch.randn(*s, generator=g).to(dev)
    return dict(recon=t(B, fs), y=t(B, fs), mean=t(B, Ld), logvar=0.3 * t(B, Ld),
                scale=torch.rand(B, Ld, generator=g).to(dev) + 0.5, shape=torch.rand(B, Ld, generator=g).to(dev) + 0.05,
                pi=torch.tensor([3.0, 0.01], device=dev))


def _both(lib, L, family, B, fs, Ld, seed, threshold, beta):
    d = _inputs(B, fs, Ld, seed)
    dev = d["recon"].device
    args = [family] + [L.ptr(d[k]) for k in ("recon", "y", "mean", "logvar", "scale", "shape", "pi")] + [threshold, beta]
    state_ws = torch.full((16,), -7.0, device=dev)
    ws = L.workspace(lib.vts_elbo_workspace_bytes(B, fs, Ld), dev)
    L.check(lib.vts_elbo_fwd(*args, L.ptr(state_ws), L.wptr(ws), ws.numel(), B, fs, Ld, L.stream()), "vts_elbo_fwd")
    state_st = torch.full((16,), -9.0, device=dev)
    before = torch.cuda.memory_allocated()
    L.check(lib.vts_elbo_fwd_static(*args, L.ptr(state_st), B, fs, Ld, L.stream()), "vts_elbo_fwd_static")
    assert torch.cuda.memory_allocated() == before
    torch.cuda.synchronize()
    return state_ws[:10].cpu(), state_st[:10].cpu()


@pytest.mark.parametrize("family", [0, 1, 2, 3])
def test_static_elbo_equals_workspace_elbo_twice(family):
    from vts import _lib as L
    lib = L.load()
    thr = 0.8
    # first call: a large input (many block partials); second: a small one (few partials, the rest stale)
    for (B, fs, Ld, seed) in [(16384, 1, 32, 11), (24, 3, 8, 12)]:
        a, b = _both(lib, L, family, B, fs, Ld, seed, thr, 0.001)
        assert torch.isfinite(a[0]), a
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (family, B, a, b)


def test_static_elbo_rejects_oversize_and_keeps_working():
    from vts import _lib as L
    lib = L.load()
    limit = lib.vts_elbo_fwd_static_max_elements()
    assert limit == 1 << 20
    dev = torch.device("cuda:0")
    x = torch.zeros(8, device=dev)
    B = limit // 32 + 1
    rc = lib.vts_elbo_fwd_static(0, L.ptr(x), L.ptr(x), L.ptr(x), L.ptr(x), None, None, None, 0.5, 0.001, L.ptr(x),
                                 B, 1, 32, L.stream())
    assert rc == 1 and b"static scratch" in lib.vts_last_error()
    # exactly at the limit is accepted
    B = limit // 32
    d = _inputs(B, 1, 32, 5)
    st = torch.zeros(16, device=dev)
    L.check(lib.vts_elbo_fwd_static(0, L.ptr(d["recon"]), L.ptr(d["y"]), L.ptr(d["mean"]), L.ptr(d["logvar"]), None, None,
                                    None, 0.5, 0.001, L.ptr(st), B, 1, 32, L.stream()), "vts_elbo_fwd_static")
    ws = L.workspace(lib.vts_elbo_workspace_bytes(B, 1, 32), dev)
    st2 = torch.zeros(16, device=dev)
    L.check(lib.vts_elbo_fwd(0, L.ptr(d["recon"]), L.ptr(d["y"]), L.ptr(d["mean"]), L.ptr(d["logvar"]), None, None, None,
                             0.5, 0.001, L.ptr(st2), L.wptr(ws), ws.numel(), B, 1, 32, L.stream()), "vts_elbo_fwd")
    torch.cuda.synchronize()
    assert torch.equal(st[:10].cpu().view(torch.int32), st2[:10].cpu().view(torch.int32))

"""What happens around the cooperative recurrent kernels (cluster forward, pair backward: grids of workgroups that
wait for each other) when they do not have the GPU to themselves, and how a missed hand-off reaches the caller.

* another kernel holds compute units while they are launched (tests/helpers/occupy.hip, the stand-in for a
  collective's kernel on a side stream): results must equal the undisturbed run's and every status word must be 0;
* a peer that never delivers (the library's test hook): the optimizer update of that step is skipped on the device
  (vts_adam_step_guarded reads the step's status words) and train_step / DataParallelTrainer raise
  vts.RecurrenceTimeout once the words have reached the host, instead of stepping on NaN gradients;
* vts_lstm_recurrent_bwd_status after launches of the kernels that wait for nobody: 0, whatever the workspace held;
* the data-parallel exact loss with a real mixture model, its statistics reduced through an RCCL group.
The LSTM itself is the reference's nn.LSTM call (vae-models/mixture-vae-DE.py:45,60)."""
import ctypes as C
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu
HELPERS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "libvts_test_helpers.so")


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch.device("cuda:0")


def l2_rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.fixture()
def bf16x3():
    import vts
    old = vts.get_mode()
    vts.set_mode("bf16x3")
    yield
    vts.set_mode(old)


def _occupier():
    assert os.path.exists(HELPERS), f"{HELPERS} not built (make -C tests/helpers, or __graft_entry__.build())"
    lib = C.CDLL(HELPERS)
    lib.vts_test_occupy.restype = C.c_int
    lib.vts_test_occupy.argtypes = [C.c_int, C.c_longlong, C.c_void_p, C.c_void_p]
    return lib


def _fwd_bwd(lstm, x):
    import vts
    for p in lstm.parameters():
        p.grad = None
    vts.recurrence_status.begin_step()
    _, (hn, cn) = lstm(x, need_output=False)
    (hn.pow(2).sum() + cn.sum()).backward()
    status = vts.recurrence_status.read()
    return hn.detach().clone(), {k: p.grad.clone() for k, p in lstm.named_parameters()}, status


@pytest.mark.parametrize("held_cus", [8, 32])
def test_recurrences_beside_a_kernel_that_holds_compute_units(bf16x3, held_cus):
    """4096 windows = 256 workgroups of 160 KiB LDS: the cluster and pair kernels need every CU.  While a side-stream
    kernel holds `held_cus` of them for 20 ms the missing workgroups cannot start and their peers poll; once the
    CUs are free the launch completes.  Same bits as the undisturbed run, status words 0."""
    import vts
    dev = _dev()
    occ = _occupier()
    torch.manual_seed(5)
    B, T, Cin, H = 4096, 6, 8, 256
    lstm = vts.LSTM(Cin, H, num_layers=2).to(dev)
    x = torch.randn(B, T, Cin, device=dev)
    hn0, grads0, st0 = _fwd_bwd(lstm, x)
    torch.cuda.synchronize()
    assert len(st0) == 4 and all(v == 0 for _, v in st0), st0        # two layers, forward and backward
    assert torch.isfinite(hn0).all()

    side = torch.cuda.Stream()
    sink = torch.zeros(1, dtype=torch.int32, device=dev)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    held_us = 20000
    assert occ.vts_test_occupy(held_cus, held_us, C.c_void_p(sink.data_ptr()), C.c_void_p(side.cuda_stream)) == 0
    t0.record()
    hn1, grads1, st1 = _fwd_bwd(lstm, x)          # (reading the status words synchronises the current stream only)
    t1.record()
    torch.cuda.synchronize()
    assert all(v == 0 for _, v in st1), st1
    assert torch.equal(hn0, hn1)
    for k in grads0:
        # the weight-gradient reductions are not bitwise repeatable from launch to launch; the recurrences are
        assert l2_rel(grads1[k], grads0[k]) <= 1e-6, k
    print(f"[{held_cus} CUs held for {held_us / 1e3:.0f} ms] forward + backward beside it: {t0.elapsed_time(t1):.2f} ms")


def test_train_step_raises_on_a_missed_handoff(bf16x3):
    """A launch that timed out: the optimizer update of that step is skipped ON THE DEVICE (its status words guard
    vts_adam_step_guarded), and vts.RecurrenceTimeout, naming the kernel, is raised once the words have reached the
    host - by vts.raise_on_recurrence_timeout() (synchronous) or at the start of a later step."""
    import vts
    from vts import _lib
    lib = _lib.load()
    dev = _dev()
    torch.manual_seed(6)
    B, T, Cin, H, Ld = 128, 4, 8, 256, 16
    model = vts.build_model("gaussian", Cin, H, Ld, 1).to(dev)
    model.train()
    opt = vts.Adam(vts.FlatParameters(model), lr=1e-3)
    x, y = torch.randn(B, T, Cin, device=dev), torch.randn(B, 1, device=dev)
    vts.train_step(model, opt, x, y)
    vts.raise_on_recurrence_timeout()            # clean
    before = opt.flat.data.clone()
    moments = opt.exp_avg.clone()
    for hook, kernel in ((lib.vts_debug_set_pair_fault, "lstm_rec3p_bwd_kernel"),
                         (lib.vts_debug_set_cluster_fault, "lstm_rec3c_fwd_kernel")):
        hook(1)
        try:
            vts.train_step(model, opt, x, y)                 # does not raise: nothing has been read back yet
        finally:
            hook(-1)
        with pytest.raises(vts.RecurrenceTimeout, match=kernel):
            vts.raise_on_recurrence_timeout()
        assert torch.equal(opt.flat.data, before), "no update may be made from a launch that timed out"
        assert torch.equal(opt.exp_avg, moments), "... nor to the optimizer's moments"
    # the asynchronous route: the next step that finds the words on the host raises before it launches anything
    lib.vts_debug_set_pair_fault(1)
    try:
        vts.train_step(model, opt, x, y)
    finally:
        lib.vts_debug_set_pair_fault(-1)
    torch.cuda.synchronize()
    with pytest.raises(vts.RecurrenceTimeout, match="lstm_rec3p_bwd_kernel"):
        vts.train_step(model, opt, x, y)
    assert torch.equal(opt.flat.data, before)
    trainer = vts.DataParallelTrainer(model, lr=1e-3)        # (no process group: the single-rank path of the trainer)
    tbefore = trainer.flat.data.clone()
    lib.vts_debug_set_pair_fault(0)
    try:
        trainer.step(x, y)
    finally:
        lib.vts_debug_set_pair_fault(-1)
    with pytest.raises(vts.RecurrenceTimeout, match="lstm_rec3p_bwd_kernel"):
        vts.raise_on_recurrence_timeout()
    assert torch.equal(trainer.flat.data, tbefore)
    loss = vts.train_step(model, opt, x, y)      # the next step is clean
    vts.raise_on_recurrence_timeout()
    assert torch.isfinite(loss).item() and not torch.equal(opt.flat.data, before)
    # switched off, the step goes through (and poisons the parameters: the caller's choice)
    vts.set_status_checks(False)
    lib.vts_debug_set_pair_fault(1)
    try:
        vts.train_step(model, opt, x, y)
    finally:
        lib.vts_debug_set_pair_fault(-1)
        vts.set_status_checks(True)
    assert not torch.isfinite(opt.flat.data).all()


@pytest.mark.parametrize("H,mode", [(64, 1), (256, 0), (128, 1)])
def test_backward_status_is_zero_after_kernels_that_wait_for_nobody(H, mode):
    """The one-workgroup kernels (hidden 64 / 128 in bf16x3 mode, everything in fp32 mode) cannot time out; the query
    after them must say 0 even when the workspace came back from the allocator full of ones."""
    from vts import _lib as L
    lib = L.load()
    dev = _dev()
    B, T = 32, 4
    torch.manual_seed(7)
    gates = torch.rand(B * T * 4 * H, device=dev) * 0.5 + 0.2
    cells = torch.randn(B * T * H, device=dev) * 0.3
    w_hh = torch.randn(4 * H, H, device=dev) * 0.05
    dh_last = torch.randn(B, H, device=dev)
    db = torch.empty(4 * H, device=dev)
    ws_bytes = lib.vts_lstm_recurrent_workspace_bytes(B, T, H)
    ws = torch.full((ws_bytes,), 0xFF, dtype=torch.uint8, device=dev)
    status = torch.full((1,), -1, dtype=torch.int32, device=dev)
    L.check(lib.vts_lstm_recurrent_bwd(L.wptr(gates), L.ptr(cells), None, L.ptr(dh_last), None, L.ptr(w_hh), L.ptr(db), None,
                                       0, L.wptr(ws), ws_bytes, B, T, H, mode, L.stream()), "vts_lstm_recurrent_bwd")
    L.check(lib.vts_lstm_recurrent_bwd_status(L.wptr(ws), ws_bytes, B, T, H, L.wptr(status), L.stream()),
            "vts_lstm_recurrent_bwd_status")
    torch.cuda.synchronize()
    assert status.item() == 0
    assert torch.isfinite(db).all()


def test_exact_loss_of_two_ranks_through_rccl(bf16x3):
    """Two replicas of a vts mixture model, each with the trainer's exact-loss path on its half of a batch whose
    halves hold different numbers of rows inside the threshold.  The loss statistics go through a real RCCL
    all-reduce (a one-rank group: the call, the stream ordering and the in-place result are those of N ranks), to
    which the test adds the other replica's statistics - what the second rank would have contributed.  Each half's
    loss must be the whole batch's, the summed flat gradients x 1/2 the whole batch's gradient, and the Adam update
    from them the single-process update (rel 1e-5: fp32 sums in another order)."""
    import vts
    from oracle import ref_torch as R
    dev = _dev()
    torch.cuda.set_device(dev)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        kind, (B, T, Cin, H, Ld) = "mixture", (256, 4, 8, 256, 16)       # 128 windows per rank: the plane path, cluster + pair kernels
        sd = R.init_state(kind, Cin, H, Ld, 1, seed=8)
        sd["encoder.shape_layer_extreme.weight"] *= 0.25
        sd["encoder.shape_layer_extreme.bias"].fill_(0.5)
        g = torch.Generator().manual_seed(9)
        x = torch.randn(B, T, Cin, generator=g).to(dev)
        y = torch.randn(B, 1, generator=g)
        y[: B // 2] *= 0.2
        y = y.to(dev)
        noise = {k: v.to(dev) for k, v in R.draw_noise(kind, B, Ld, seed=10).items()}

        def replica():
            m = vts.build_model(kind, Cin, H, Ld, 1, threshold=0.7).to(dev)
            m.load_state_dict({k: v.to(dev) for k, v in sd.items()})
            m.train()
            return m

        # the whole batch in one process
        whole = replica()
        opt = vts.Adam(vts.FlatParameters(whole), lr=1e-3)
        loss_whole = vts.train_step(whole, None, x, y, noise)
        grads_whole = {k: p.grad.detach().clone() for k, p in whole.named_parameters()}
        vts.train_step(whole, opt, x, y, noise)

        h = B // 2
        halves = [(x[:h], y[:h], {k: v[:h] for k, v in noise.items()}), (x[h:], y[h:], {k: v[h:] for k, v in noise.items()})]
        trainers = [vts.DataParallelTrainer(replica(), lr=1e-3, exact_loss=True) for _ in range(2)]
        for t in trainers:
            t.world = 2
        # pass 1: each replica's own statistics, as its rank would hand them to the collective
        own = []
        for t, (xs, ys, ns) in zip(trainers, halves):
            seen = []
            vts.set_loss_reduction(lambda s, seen=seen: seen.append(s.clone()), 1.0)
            try:
                with torch.no_grad():
                    outs = t.model(xs, ns)
                    t.model.loss_function(outs[0], ys, *outs[1:])
            finally:
                vts.set_loss_reduction(None)
            assert len(seen) == 1
            own.append(seen[0])
        # pass 2: the collective + the peer's contribution
        calls = []
        for r, (t, (xs, ys, ns)) in enumerate(zip(trainers, halves)):
            def reduce(s, peer=own[1 - r]):
                dist.all_reduce(s, op=dist.ReduceOp.SUM)
                s.add_(peer)
                calls.append(1)
            t.reduce_loss_sums = reduce
            loss = t.step(xs, ys, ns, apply_update=False)
            assert abs(loss.item() - loss_whole.item()) <= 1e-5 * abs(loss_whole.item()), (r, loss.item(), loss_whole.item())
        assert len(calls) == 2
        torch.cuda.synchronize()
        summed = trainers[0].flat.grad + trainers[1].flat.grad          # what the gradient all-reduce leaves on both ranks
        for t in trainers:
            t.flat.grad.copy_(summed)
        flat = trainers[0].flat
        for k, p, o in zip(flat.names, flat.params, flat.offsets):
            got, ref = flat.grad[o:o + p.numel()].view_as(p) / 2, grads_whole[k]
            if k == "pi_params":      # cancelling softmax backward: absolute bound, as in test_gpu_parity.py
                assert (got - ref).abs().max().item() <= 1e-6 * abs(loss_whole.item())
            else:
                assert l2_rel(got, ref) < 1e-5, (k, l2_rel(got, ref))
        for t in trainers:
            t.opt.step(grad_scale=0.5)
            assert l2_rel(t.flat.data, opt.flat.data) < 1e-5
    finally:
        dist.destroy_process_group()

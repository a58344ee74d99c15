"""Two ranks, two processes, one GPU: vts.DataParallelTrainer around a real vts mixture model (bf16x3, hidden 256:
the plane path with its cluster and pair recurrences) on each rank's half of a batch, the collectives over gloo
with device tensors (an RCCL group cannot hold two ranks of one device; the trainer's calls are the same).

* the step of the two ranks together must be the single-process step on the whole batch: exact loss (halves with
  different class counts, so per-rank means would differ), identical parameters on both ranks after the Adam update,
  equal to the single-process update to 1e-5;
* a recurrence that times out on ONE rank (the library's test hook, set in that process only): the gradient
  all-reduce spreads its NaN to both, the MAX-reduced status words stop the update on both, and both raise
  vts.RecurrenceTimeout.
(Launched cooperatively - VTS_LSTM_CLUSTER_COOP=1 - the recurrences of one of the two processes would be refused:
a device has one cooperative queue.  The library answers a refusal with the one-workgroup kernels; the default
plain launches are not affected.)
Reference call sites: the train loop of vae-models/mixture-vae-DE.py:158-164 around nn.LSTM (:45,60)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KIND, DIMS = "mixture", (256, 4, 8, 256, 16)      # B, T, Cin, H, latent: 128 windows per rank


def _paths():
    for p in (HERE, os.path.join(HERE, "vae-time-series_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _inputs():
    _paths()
    from oracle import ref_torch as R
    B, T, Cin, H, Ld = DIMS
    sd = R.init_state(KIND, Cin, H, Ld, 1, seed=8)
    sd["encoder.shape_layer_extreme.weight"] *= 0.25
    sd["encoder.shape_layer_extreme.bias"].fill_(0.5)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, T, Cin, generator=g)
    y = torch.randn(B, 1, generator=g)
    y[: B // 2] *= 0.2                                 # the first rank's half holds more rows inside the threshold
    noise = R.draw_noise(KIND, B, Ld, seed=10)
    return sd, x, y, noise


def _worker(rank, world, port, fault_rank, ret):
    _paths()
    import vts
    from vts import _lib
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        vts.set_mode("bf16x3")
        B, T, Cin, H, Ld = DIMS
        sd, x, y, noise = _inputs()
        model = vts.build_model(KIND, Cin, H, Ld, 1, threshold=0.7).to(dev)
        model.load_state_dict({k: v.to(dev) for k, v in sd.items()})
        model.train()
        trainer = vts.DataParallelTrainer(model, lr=1e-3, exact_loss=True)
        assert trainer.world == 2 and trainer.distributed
        xs, ys = vts.shard_batch(x, rank, world).to(dev), vts.shard_batch(y, rank, world).to(dev)
        ns = {k: vts.shard_batch(v, rank, world).to(dev) for k, v in noise.items()}
        before = trainer.flat.data.clone()
        if fault_rank == rank:
            _lib.load().vts_debug_set_cluster_fault(1)
        loss = trainer.step(xs, ys, ns)
        _lib.load().vts_debug_set_cluster_fault(-1)
        raised = ""
        try:
            vts.raise_on_recurrence_timeout()
        except vts.RecurrenceTimeout as e:
            raised = str(e)
        torch.cuda.synchronize()
        ret[rank] = (float(trainer.global_loss(loss)) if fault_rank < 0 else 0.0, trainer.flat.data.cpu(),
                     bool(torch.equal(trainer.flat.data, before)), raised)
    finally:
        dist.destroy_process_group()


def _spawn(fault_rank):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(2, port, fault_rank, ret), nprocs=2, join=True)
    return ret


def test_two_ranks_make_the_single_process_step():
    _paths()
    import vts
    dev = torch.device("cuda:0")
    B, T, Cin, H, Ld = DIMS
    sd, x, y, noise = _inputs()
    vts.set_mode("bf16x3")
    try:
        whole = vts.build_model(KIND, Cin, H, Ld, 1, threshold=0.7).to(dev)
        whole.load_state_dict({k: v.to(dev) for k, v in sd.items()})
        whole.train()
        opt = vts.Adam(vts.FlatParameters(whole), lr=1e-3)
        loss_whole = vts.train_step(whole, opt, x.to(dev), y.to(dev), {k: v.to(dev) for k, v in noise.items()})
        vts.raise_on_recurrence_timeout()
        params_whole = opt.flat.data.cpu()
        loss_whole = loss_whole.item()
    finally:
        vts.set_mode("f32")
    ret = _spawn(-1)
    for rank in range(2):
        loss, params, unchanged, raised = ret[rank]
        assert raised == "" and not unchanged
        assert abs(loss - loss_whole) <= 1e-5 * abs(loss_whole), (rank, loss, loss_whole)
        rel = ((params - params_whole).norm() / params_whole.norm()).item()
        assert rel < 1e-5, (rank, rel)
    assert torch.equal(ret[0][1], ret[1][1]), "both ranks must hold the same parameters after the update"


def test_a_timeout_on_one_rank_stops_the_update_on_both():
    ret = _spawn(1)
    for rank in range(2):
        _, _, unchanged, raised = ret[rank]
        assert unchanged, f"rank {rank} updated its parameters from a step in which a launch timed out"
        assert "lstm_rec3c_fwd_kernel" in raised, (rank, raised)
    assert "this rank or another one" in ret[0][3]

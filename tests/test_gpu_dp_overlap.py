"""The data-parallel trainer around a vts model on the GPU (one rank: the collective path is the same code).  Every
LSTM layer is its own autograd node, so the gradients of the upper layers are final - and their bucket's all-reduce
is on its way - while the layers below are still in their backward pass; and the trainer's gradients equal the plain
train step's.  (The reduction across ranks itself is covered by tests/test_parallel_gloo.py on the CPU.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def test_buckets_leave_before_the_lower_layers_finish():
    import vts
    from vts import parallel
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        vts.set_mode("bf16x3")
        B, T, C, H, Ld = 128, 4, 8, 256, 32          # takes the plane path of the LSTM stack
        torch.manual_seed(3)
        model = vts.build_model("mixture", C, H, Ld, 1, threshold=0.8).to(dev)
        model.train()
        g = torch.Generator().manual_seed(4)
        x, y = torch.randn(B, T, C, generator=g).to(dev), torch.randn(B, 1, generator=g).to(dev)
        from oracle import ref_torch as R
        noise = {k: v.to(dev) for k, v in R.draw_noise("mixture", B, Ld, seed=5).items()}

        # reference: the plain train step's gradients
        loss0 = vts.train_step(model, None, x, y, noise)
        ref = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
        for p in model.parameters():
            p.grad = None

        orig = parallel.plan_buckets
        parallel.plan_buckets = lambda flat, max_buckets=3, min_floats=1 << 18: orig(flat, 3, 1 << 17)
        try:
            trainer = vts.DataParallelTrainer(model, lr=1e-3)
        finally:
            parallel.plan_buckets = orig
        assert len(trainer.buckets) == 3
        trainer.trace = True
        loss = trainer.step(x, y, noise, apply_update=False)
        torch.cuda.synchronize()
        assert abs(loss.item() - loss0.item()) <= 1e-6 * abs(loss0.item())
        for k, p in model.named_parameters():
            err = ((p.grad - ref[k]).double().norm() / ref[k].double().norm().clamp_min(1e-30)).item()
            assert err <= 1e-6, (k, err)      # same kernels; a few reductions are not bitwise repeatable
        names = trainer.flat.names
        ev = trainer.events
        first_reduce = next(i for i, e in enumerate(ev) if e[0] == "allreduce")
        pos = {names[e[1]]: i for i, e in enumerate(ev) if e[0] == "grad"}
        # layer 2's gradients are final before layer 0's, and a bucket has left before layer 0 is done
        assert pos["encoder.lstm.weight_hh_l2"] < pos["encoder.lstm.weight_hh_l1"] < pos["encoder.lstm.weight_hh_l0"]
        assert first_reduce < pos["encoder.lstm.weight_ih_l0"]
        assert sum(1 for e in ev if e[0] == "allreduce") == 3
    finally:
        vts.set_mode("f32")
        dist.destroy_process_group()

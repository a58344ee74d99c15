"""A train step recorded as a HIP graph (vts.GraphedTrainStep) against the same steps launched one by one:
same parameters after k steps, same losses.  The random draws are made on the device in both cases; torch's
generator hands a replayed graph the same Philox offsets the eager launches consume, so the trajectories
are compared directly (rel 1e-5: the only difference is the Adam bias correction computed on the device)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch.device("cuda:0")


def run(kind, graphed, steps, dev):
    import vts
    torch.manual_seed(0)
    model = vts.build_model(kind, 1, 64, 32, 1).to(dev)
    model.train()
    flat = vts.FlatParameters(model)
    opt = vts.Adam(flat, lr=1e-3)
    g = torch.Generator().manual_seed(5)
    xs = [torch.randn(16, 16, 1, generator=g).to(dev) for _ in range(steps)]
    ys = [torch.randn(16, 1, generator=g).to(dev) for _ in range(steps)]
    step = vts.GraphedTrainStep(model, opt, xs[0], ys[0]) if graphed else None
    torch.manual_seed(123)
    losses = []
    for x, y in zip(xs, ys):
        loss = step(x, y) if graphed else vts.train_step(model, opt, x, y)
        losses.append(loss.clone())
    torch.cuda.synchronize()
    return torch.stack(losses).cpu(), flat.data.clone().cpu(), opt


@pytest.mark.parametrize("kind", ["gaussian", "mixture"])
def test_graph_replay_matches_eager_steps(kind):
    dev = _dev()
    l_e, p_e, _ = run(kind, False, 6, dev)
    l_g, p_g, opt = run(kind, True, 6, dev)
    assert int(opt.step_state[0].item()) == 6           # building the graph left the counter where it was
    assert torch.allclose(l_g, l_e, rtol=1e-5, atol=1e-7), (l_g, l_e)
    assert ((p_g - p_e).norm() / p_e.norm()).item() < 1e-5


def test_device_step_counter_matches_host_counter():
    import vts
    dev = _dev()
    outs = []
    for on_device in (False, True):
        torch.manual_seed(1)
        model = vts.build_model("gaussian", 3, 16, 8, 1).to(dev)
        opt = vts.Adam(vts.FlatParameters(model), lr=5e-3)
        if on_device:
            opt.count_on_device()
        g = torch.Generator().manual_seed(2)
        x, y = torch.randn(8, 5, 3, generator=g).to(dev), torch.randn(8, 1, generator=g).to(dev)
        noise = {"eps": torch.randn(8, 8, generator=g).to(dev)}
        for _ in range(4):
            vts.train_step(model, opt, x, y, noise)
        outs.append(opt.flat.data.clone())
    assert ((outs[0] - outs[1]).norm() / outs[0].norm()).item() < 1e-6


def test_transformer_is_refused():
    import vts
    dev = _dev()
    m = vts.TransformerMixtureVAE(4, 64, 1, 8, 8, 1).to(dev)
    opt = vts.Adam(vts.FlatParameters(m))
    with pytest.raises(NotImplementedError):
        vts.GraphedTrainStep(m, opt, torch.zeros(4, 6, 4, device=dev), torch.zeros(4, 1, device=dev))

"""GPU parity at the benchmark's own size: one train step (forward, ELBO, backward) on 4096 windows x 96 steps
x 370 channels, hidden 256, latent 32 - the shape bench.py times - against the CPU oracle on the same
parameters, inputs and noise.  This is the only place where the bench-sized kernel paths (the cluster forward
recurrence, the 256 x 128 split-K weight-gradient GEMMs, 393 216-row input projections) meet the oracle directly;
the smaller cases in test_gpu_parity.py reach them through equivalent shapes.

Tolerance as everywhere: rel 1e-4 on the loss, L2-relative 1e-4 per parameter gradient (absolute bound for
pi_params, see test_gpu_parity.pi_grad_close).  The oracle takes a few seconds per model on the box's host
cores and is computed once per model for both arithmetic modes.

Two tests.

The LSTM stack alone under a smooth loss (sigmoid / tanh only): h_n, c_n and every LSTM parameter gradient at 1e-4
(measured worst 1.3e-5); this is where the time of a step goes.

The whole model step: loss at 1e-5 and EVERY gradient at 1e-4 in both arithmetic modes, provided the ReLU units of
the trunk and decoder were on the same side of zero in both implementations - which the test checks unit by unit
(the oracle records its pattern, forward hooks read the HIP path's) and which holds for the draws used here.  Should
a unit ever differ (a later change of rounding), the test names the flipped units and falls back to the bound a
flip can cost: the decoder's output layer (downstream of every ReLU) still at 1e-4, the other gradients at
KINK_TOL = 2e-3.  Why a flip matters: the ReLU layers make the reference's own gradient discontinuous: at this size
2 to 3 million ReLU units see pre-activations that differ by ~1e-6 between any two evaluations of the forward
pass, so about one unit per draw sits within that distance of zero, and one flipped unit moves its layer's bias
gradient by 2e-4 of its norm (one window of 4096, undiluted) and the gradients upstream of it by up to 9e-4.
Measured in bf16x3 mode over five draws: two without a flip (every gradient within 1.1e-5 of the oracle, the
figure the smooth test also shows), three with one flip (bias-gradient steps of 2e-5, 2e-4 and 3e-4).  With fp32
products the forward values sit 10x closer to the oracle's and no flip occurred in these draws (worst 1.1e-5)."""
import pytest
import torch

from oracle import ref_torch as R

pytestmark = pytest.mark.gpu

TOL = 1e-4
B, T, C, H, LD = 4096, 96, 370, 256, 32
THRESHOLD = 0.8          # |y| <= 0.8: a mixed normal / extreme mask on N(0, 1) targets
SEED = {"mixture": 11, "gpd": 11, "gaussian": 12}     # draws in which no ReLU unit changes sides (checked below)
KINK_TOL = 2e-3         # fallback bound when one does: see the module docstring
RELU_SITES = {"gaussian": [("decoder.dense1", None)],
              "gpd": [("encoder.fc1", None), ("encoder.fc2", "enc_keep"), ("decoder.fc1", None), ("decoder.fc2", "dec_keep")]}
RELU_SITES["mixture"] = RELU_SITES["gpd"]
_CACHE = {}


def _l2_rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("mode", ["bf16x3", "f32"])
def test_bench_sized_lstm_stack_matches_torch_cpu(mode):
    """Three stacked layers, 4096 x 96 x 370 -> hidden 256, loss = <h_n[-1], r> for a fixed random r: h_n, c_n
    and the gradient of every LSTM parameter against torch's CPU nn.LSTM (the arithmetic the reference runs)."""
    import vts
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    dev = torch.device("cuda:0")
    if "lstm" not in _CACHE:
        torch.manual_seed(5)
        ref = torch.nn.LSTM(C, H, 3, batch_first=True)
        x = torch.randn(B, T, C)
        r = torch.randn(B, H)
        _, (hn, cn) = ref(x)
        (hn[-1] * r).sum().backward()
        _CACHE["lstm"] = (ref.state_dict(), x, r, hn.detach(), cn.detach(),
                          {k: p.grad.clone() for k, p in ref.named_parameters()})
    sd, x, r, hn_ref, cn_ref, grads_ref = _CACHE["lstm"]
    vts.set_mode(mode)
    try:
        m = vts.LSTM(C, H, 3, batch_first=True).to(dev)
        m.load_state_dict({k: v.to(dev) for k, v in sd.items()}, strict=True)
        _, (hn, cn) = m(x.to(dev))
        (hn[-1] * r.to(dev)).sum().backward()
        torch.cuda.synchronize()
    finally:
        vts.set_mode("f32")
    assert _l2_rel(hn.detach().cpu(), hn_ref) <= TOL
    assert _l2_rel(cn.detach().cpu(), cn_ref) <= TOL
    for name, p in m.named_parameters():
        rel = _l2_rel(p.grad.detach().cpu(), grads_ref[name])
        assert rel <= TOL, f"{mode}: gradient of {name} is off by {rel:.2e} (L2-relative)"


def _case(kind):
    if kind not in _CACHE:
        sd = R.init_state(kind, C, H, LD, 1, seed=0)
        if kind in ("mixture", "gpd"):
            # keep the GPD shapes off the 1e-6 clamp, where the reference's own fp32 sampler gradient is noise
            sd["encoder.shape_layer_extreme.weight"] *= 0.25
            sd["encoder.shape_layer_extreme.bias"].fill_(0.5)
        g = torch.Generator().manual_seed(SEED[kind])
        x = torch.randn(B, T, C, generator=g)
        y = torch.randn(B, 1, generator=g)
        noise = R.draw_noise(kind, B, LD, seed=SEED[kind] + 1)
        R.RELU_TAPS = {}
        try:
            loss, _, grads = R.loss_and_grads(kind, sd, x, y, noise, threshold=THRESHOLD)
            taps = dict(R.RELU_TAPS)
        finally:
            R.RELU_TAPS = None
        _CACHE[kind] = (sd, x, y, noise, loss, grads, taps)
    return _CACHE[kind]


def _flipped_units(kind, model, taps_ref, noise, seen):
    """ReLU units (layer, window, unit) that were active in one implementation and not in the other.  `seen` holds
    the HIP layers' outputs (relu, or relu times the dropout keep mask / 0.7): a kept unit is active iff its output
    is positive; dropped units pass no gradient on either side and are left out."""
    flips = []
    for name, keep_key in RELU_SITES[kind]:
        active = seen[name].cpu() > 0
        ref = taps_ref[name]
        if keep_key is not None:
            keep = noise[keep_key] > 0
            active, ref = active & keep, ref & keep
        for w, u in (active != ref).nonzero().tolist():
            flips.append((name, w, u))
    return flips


@pytest.mark.parametrize("mode", ["bf16x3", "f32"])
@pytest.mark.parametrize("kind", ["mixture", "gpd", "gaussian"])
def test_bench_sized_step_matches_oracle(kind, mode):
    import vts
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    dev = torch.device("cuda:0")
    sd, x, y, noise, loss_ref, grads_ref, taps_ref = _case(kind)
    vts.set_mode(mode)
    seen, hooks = {}, []
    from vts import functional as VF
    try:
        model = vts.build_model(kind, C, H, LD, 1, threshold=THRESHOLD).to(dev)
        model.load_state_dict({k: v.to(dev) for k, v in sd.items()}, strict=True)
        model.train()
        for name, _ in RELU_SITES[kind]:
            layer = model.get_submodule(name)
            hooks.append(layer.register_forward_hook(lambda _m, _i, out, name=name: seen.__setitem__(name, out.detach())))
        # the layers run as fused chains hand their outputs over through this dict, the others through the hooks
        VF.CHAIN_TAPS = seen
        loss = vts.train_step(model, None, x.to(dev), y.to(dev), {k: v.to(dev) for k, v in noise.items()})
        torch.cuda.synchronize()
    finally:
        vts.set_mode("f32")
        VF.CHAIN_TAPS = None
        for h in hooks:
            h.remove()
    assert torch.isfinite(loss).item()
    flips = _flipped_units(kind, model, taps_ref, noise, seen)
    if flips:
        print(f"{kind}/{mode}: {len(flips)} ReLU unit(s) on different sides of zero: {flips[:8]}")
    assert abs(loss.item() - loss_ref.item()) <= 1e-5 * abs(loss_ref.item())
    for name, p in model.named_parameters():
        ref = grads_ref[name]
        got = p.grad.detach().cpu()
        if name == "pi_params":
            assert (got - ref).abs().max().item() <= 1e-6 * abs(loss_ref.item())
            continue
        rel = _l2_rel(got, ref)
        tol = TOL if (not flips or name.startswith("decoder.out.")) else KINK_TOL
        assert rel <= tol, (f"{kind}/{mode}: gradient of {name} is off by {rel:.2e} (L2-relative, bound {tol:.0e}, "
                            f"{len(flips)} flipped ReLU units)")
    assert not flips, (f"{kind}/{mode}: gradients are within the fallback bound, but {len(flips)} ReLU unit(s) changed "
                       f"sides ({flips[:8]}): pick another draw (SEED) so that the strict bound is the one tested")


TRANSFORMER_SEED = 4
# The feed-forward blocks hold 300 million ReLU units here: whatever the draw, hundreds of them sit within the two
# implementations' forward difference (~1e-6 to 5e-6) of zero.  So the test picks its dropout masks - noise that is
# fed in, any 0/1 pattern is a legal draw - such that every unit whose pre-activation lies within MARGIN of zero in
# the oracle's forward pass is a dropped one: it passes no gradient on either side, whichever side it lands on.
# (A few thousand of 300 million units; layer by layer, since dropping units of one block moves the next block's
# pre-activations.)  decoder.fc1 has no dropout site: 131 072 units, compared like the others.
MARGIN = 5e-5


def _drop_units_near_zero(RT, sd, x, y, noise, nhead, thr, layers):
    noise = dict(noise)
    sites = [(f"l{i}.ff", f"l{i}.ff_keep") for i in range(layers)] + [("decoder.fc2", "dec_keep")]
    dropped = 0
    for site, key in sites:
        RT.NEAR_ZERO, RT.NEAR_ZERO_MARGIN = {}, MARGIN
        try:
            with torch.no_grad():
                RT.forward_loss(sd, x, y, noise, nhead, thr, 1.0)
            near = RT.NEAR_ZERO[site].reshape(noise[key].shape)
        finally:
            RT.NEAR_ZERO = None
        dropped += int((near & (noise[key] > 0)).sum())
        noise[key] = noise[key] * (~near).to(noise[key].dtype)
    return noise, dropped


def _transformer_flips(taps_ref, noise, ffn_hidden, dec_seen, Bt, layers):
    """ReLU units (site, row, unit) active in one implementation and not in the other.  The HIP path's feed-forward
    blocks hand over dropout(relu(linear1(x))) (rows = window * T + step; the last layer is evaluated for the final
    position only, one row per window): a kept unit is active iff that value is positive; dropped units pass no
    gradient on either side and are left out.  Compared on the GPU (200 million units per full block)."""
    dev = ffn_hidden[0].device
    flips = []
    for i in range(layers):
        ref = taps_ref[f"l{i}.ff"]
        keep = noise[f"l{i}.ff_keep"] > 0
        if i == layers - 1:
            ref, keep = ref[:, -1], keep[:, -1]
        ref, keep = ref.reshape(-1, ref.shape[-1]).to(dev), keep.reshape(-1, keep.shape[-1]).to(dev)
        active = ffn_hidden[i][: ref.shape[0]] > 0
        for row, u in ((active & keep) != (ref & keep)).nonzero().tolist()[:64]:
            flips.append((f"l{i}.ff", row, u))
    for name, keep_key in (("decoder.fc1", None), ("decoder.fc2", "dec_keep")):
        active, ref = dec_seen[name].cpu() > 0, taps_ref[name]
        if keep_key is not None:
            keep = noise[keep_key] > 0
            active, ref = active & keep, ref & keep
        for row, u in (active != ref).nonzero().tolist():
            flips.append((name, row, u))
    return flips


@pytest.mark.parametrize("mode", ["bf16x3", "f32"])
def test_bench_scale_transformer_step_matches_oracle(mode):
    """The attention-encoder variant on a quarter of the bench batch (1024 windows x 96 x 370 = 98 304 tokens
    through d_model 64, 8 heads, 2 layers, dim_ff 2048; the oracle's masks alone are 2.4 GB at this size).  Same
    scheme as the LSTM models' step: the ReLU patterns of the feed-forward blocks (200 million units in the full
    layer) and of the decoder are compared unit by unit first; where they agree - which the choice of dropout masks
    above arranges - EVERY gradient is held to 1e-4 in both arithmetic modes.  Should a unit differ, the test names
    the flipped units and falls back to KINK_TOL for the gradients upstream of a ReLU."""
    import vts
    from vts import functional as VF
    from oracle import ref_transformer as RT
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    dev = torch.device("cuda:0")
    Bt, d, nhead, layers = 1024, 64, 8, 2
    if "transformer" not in _CACHE:
        sd = RT.init_state(C, d, layers, LD, 1, seed=3)
        sd["encoder.shape_layer_extreme.weight"] *= 0.25
        sd["encoder.shape_layer_extreme.bias"].fill_(0.5)
        g = torch.Generator().manual_seed(TRANSFORMER_SEED)
        x = torch.randn(Bt, T, C, generator=g)
        y = torch.randn(Bt, 1, generator=g)
        thr = float(y.abs().median())
        noise = RT.draw_noise(Bt, T, d, nhead, layers, LD, seed=TRANSFORMER_SEED + 1)
        noise, ndropped = _drop_units_near_zero(RT, sd, x, y, noise, nhead, thr, layers)
        print(f"transformer: {ndropped} kept units within {MARGIN:g} of zero moved to the dropped side")
        RT.RELU_TAPS = {}
        try:
            loss, _, grads = RT.loss_and_grads(sd, x, y, noise, nhead, thr, beta=1.0)
            taps = dict(RT.RELU_TAPS)
        finally:
            RT.RELU_TAPS = None
        _CACHE["transformer"] = (sd, x, y, thr, noise, loss, grads, taps)
    sd, x, y, thr, noise, loss_ref, grads_ref, taps_ref = _CACHE["transformer"]
    vts.set_mode(mode)
    ffn_hidden, dec_seen = [], {}
    try:
        m = vts.TransformerMixtureVAE(C, d, layers, nhead, LD, 1, beta=1.0).to(dev)
        m.load_state_dict({k: v.to(dev) for k, v in sd.items()}, strict=True)
        m.train()
        VF.FFN_TAPS, VF.CHAIN_TAPS = ffn_hidden, dec_seen
        outs = m(x.to(dev), {k: v.to(dev) for k, v in noise.items()})
        loss = m.loss_function(outs[0], y.to(dev), *outs[1:], thr)
        loss.backward()
        torch.cuda.synchronize()
    finally:
        vts.set_mode("f32")
        VF.FFN_TAPS, VF.CHAIN_TAPS = None, None
    assert len(ffn_hidden) == layers
    flips = _transformer_flips(taps_ref, noise, ffn_hidden, dec_seen, Bt, layers)
    if flips:
        print(f"transformer/{mode}: {len(flips)} ReLU unit(s) on different sides of zero: {flips[:8]}")
    assert abs(loss.item() - loss_ref.item()) <= 1e-5 * abs(loss_ref.item())
    for name, p in m.named_parameters():
        ref = grads_ref[name]
        if ref.abs().max() == 0:
            assert p.grad is None or p.grad.abs().max().item() == 0, name
            continue
        if name == "pi_params":
            assert (p.grad.detach().cpu() - ref).abs().max().item() <= 1e-6 * abs(loss_ref.item())
            continue
        rel = _l2_rel(p.grad.detach().cpu(), ref)
        tol = TOL if (not flips or name.startswith("decoder.out.") or name.startswith("decoder.final_fcn.")) else KINK_TOL
        assert rel <= tol, (f"transformer/{mode}: gradient of {name} is off by {rel:.2e} (L2-relative, bound {tol:.0e}, "
                            f"{len(flips)} flipped ReLU units)")
    assert not flips, (f"transformer/{mode}: gradients are within the fallback bound, but {len(flips)} ReLU unit(s) changed "
                       f"sides ({flips[:8]}): widen MARGIN or pick another draw (TRANSFORMER_SEED) so that the strict bound is the one tested")

#!/usr/bin/env python3
"""Train-step throughput of the LSTM-VAE hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--model mixture|gaussian|gpd]

One step = one pass of the hot path over one batch of synthetic windows that already sit in
HBM: forward, ELBO, backward, gradient all-reduce (N > 1) and the Adam update, with the random
draws (eps, u, choice, dropout masks) made on the device inside the step as the reference makes
them.  Workload = LD2011-shaped windows: 4096 windows per GPU x 96 steps x 370 channels, hidden
256, latent 32 (weak scaling: every GPU adds 4096 windows).  Rank 0 prints ONE JSON line.

N > 1: one rank per GPU over RCCL.  Under torch.distributed.run, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* are read
from the environment; started plainly, the parent process (before it touches the GPU) starts N fresh child
processes with that environment set and relays rank 0's line.

`value` is the whole step including the Adam update; `value_fwd_bwd_elbo` is the same step without the update
(timed over the same number of steps right after), so both figures of the metric are in the line.
"""
from __future__ import annotations

import argparse
import gc
import json
import math
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "vae-time-series_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch
import torch.distributed as dist

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X dense f32 matrix peak (64 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X dense bf16 matrix peak
PEAK_HBM_GBPS = 8000.0
PEAK_L2_GBPS = 34500.0          # aggregate XCD-L2 read bandwidth


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="mixture", choices=["gaussian", "gpd", "mixture", "mixture3", "transformer"],
                    help="the three LSTM-VAE families (headline: mixture) or the transformer-encoder mixture VAE")
    ap.add_argument("--d-model", type=int, default=64)
    ap.add_argument("--nhead", type=int, default=8)
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--shape", default="ld2011", choices=["ld2011", "hurricane"],
                    help="ld2011: 96-step x 370-channel standardised-load windows, hidden 256 (the headline workload); "
                         "hurricane: the reference's own window shape (gpd-lstm-DE.py:77-84: 16 steps x 1 channel, hidden "
                         "64, latent 32) filled with an AR(1) intensity walk and block-maximum targets; use with "
                         "--model gpd --batch 16|4096|65536")
    ap.add_argument("--batch", type=int, default=4096, help="windows per GPU")
    ap.add_argument("--T", type=int, default=None)
    ap.add_argument("--C", type=int, default=None)
    ap.add_argument("--H", type=int, default=None)
    ap.add_argument("--L", type=int, default=32)
    ap.add_argument("--mode", default="bf16x3", choices=["f32", "bf16x3"],
                    help="arithmetic of the LSTM's dense products: fp32 operands on the f32 matrix cores, or each "
                         "fp32 operand split into two bf16 and three bf16 products per fp32 product (fp32 "
                         "accumulate); both hold the 1e-4 parity bound.  Heads, cell update and ELBO are f32.")
    ap.add_argument("--exact-loss", action="store_true",
                    help="data-parallel runs: loss of the concatenated batch (one more all-reduce of 8 floats per step)")
    ap.add_argument("--graph", action="store_true",
                    help="record the step once as a HIP graph (vts.GraphedTrainStep) and time its replays: for the small "
                         "shapes whose eager step is bound by the host's launch rate; single process, LSTM models")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=256, help="windows in the CPU baseline sample")
    ap.add_argument("--traffic", action="store_true", default=None,
                    help="also measure roofline.traffic: two short child runs of this script under rocprofv3 --pmc "
                         "(FETCH_SIZE, WRITE_SIZE), started before this process touches the GPU; adds about twenty "
                         "seconds.  Default: on for the full-report single-GPU run of the headline workload (LSTM models, "
                         "LD2011 shape, bf16x3, CPU baseline leg included, no --graph, not under a profiler), off otherwise")
    ap.add_argument("--no-traffic", dest="traffic", action="store_false", help="leave roofline.traffic null")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)
    a = ap.parse_args()
    dT, dC, dH = (16, 1, 64) if a.shape == "hurricane" else (96, 370, 256)
    a.T, a.C, a.H = a.T or dT, a.C or dC, a.H or dH
    if a.traffic is None:
        # (not under a profiler already: its preloaded tool library would travel into the child runs)
        profiled = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
        a.traffic = (a.gpus == 1 and a.model != "transformer" and a.shape != "hurricane" and a.mode == "bf16x3" and
                     not a.graph and not a.traffic_child and not a.no_cpu_baseline and not profiled)
    return a


def synth_windows(B, T, C, rank, device):
    """Standardised-load-shaped input: N(0,1) with 0.06 % exact zeros; targets N(0,1)."""
    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.randn(B, T, C, generator=g)
    zeros = torch.rand(B, T, C, generator=g) < 6e-4
    x[zeros] = 0.0
    y = torch.randn(B, 1, generator=g)
    return x.to(device), y.to(device)


def synth_hurricane(B, rank, device, T=16, horizon=8):
    """Hurricane-intensity-shaped windows of the reference's block-maximum task (vae-models/graph/
    training_block_max_hurdat.py:81-106 builds them from HURDAT2; here synthetic): per window an AR(1) walk
    x_t = 0.9 x_{t-1} + 0.3 N(0,1) from x_0 ~ N(0.18, 0.89^2), clipped to the range of the standardised training
    inputs [-1.66, 3.57]; X = the first T steps [B, T, 1], y = the maximum of the next `horizon` steps [B, 1]."""
    g = torch.Generator().manual_seed(4321 + rank)
    x = torch.empty(B, T + horizon)
    x[:, 0] = 0.18 + 0.89 * torch.randn(B, generator=g)
    steps = 0.3 * torch.randn(B, T + horizon, generator=g)
    for t in range(1, T + horizon):
        x[:, t] = 0.9 * x[:, t - 1] + steps[:, t]
    x.clamp_(-1.66, 3.57)
    return x[:, :T, None].contiguous().to(device), x[:, T:].max(dim=1, keepdim=True).values.to(device)


def lstm_flops(kind, B, T, C, H):
    """Algorithmic FLOPs of each LSTM kernel call, keyed like vts._lib.timer names.
    Per window: input projection 2*T*Cin*4H, recurrent product 2*T*H*4H; backward = one
    weight-gradient product per forward product, one data-gradient product for the recurrence
    always and for the input projection of layers >= 1."""
    layers = 1 if kind == "gaussian" else 3
    per = {}
    total = 0.0
    for k in range(layers):
        cin = C if k == 0 else H
        inproj = 2.0 * B * T * cin * 4 * H
        rec = 2.0 * B * T * H * 4 * H
        for name, f in (("lstm_inproj_fwd", inproj), ("lstm_recurrent_fwd", rec), ("lstm_recurrent_bwd", rec),
                        ("lstm_wgrad_ih", inproj), ("lstm_wgrad_hh", rec)):
            per.setdefault(name, []).append(f)
            total += f
        # both weight gradients of a layer may run as one entry point (vts_lstm_wgrad_pair): same FLOPs, already
        # counted above under the two separate names
        per.setdefault("lstm_wgrad_pair", []).append(inproj + rec)
        if k > 0:
            per.setdefault("lstm_dgrad", []).append(inproj)
            total += inproj
    return per, total


def lstm_hbm_bytes(kind, B, T, C, H):
    """Algorithmic HBM bytes of each LSTM kernel call (every operand read or written once, fp32), keyed like
    lstm_flops: gate buffer [Bp*T, 4H], cell states and sequence gradients [Bp*T, H] (Bp = B rounded up to
    16), h sequence and inputs [B*T, .]; weights are noise at these sizes."""
    layers = 1 if kind == "gaussian" else 3
    Bp = (B + 15) // 16 * 16
    gates, cells, hs = 4.0 * Bp * T * 4 * H, 4.0 * Bp * T * H, 4.0 * B * T * H
    per = {}
    for k in range(layers):
        xin = 4.0 * B * T * (C if k == 0 else H)
        top = k == layers - 1
        for name, b in (("lstm_inproj_fwd", xin + gates),
                        ("lstm_recurrent_fwd", 2 * gates + cells + hs),
                        # gates in, gate gradients out, cells in; below the top layer also the sequence gradient
                        ("lstm_recurrent_bwd", 2 * gates + cells + (0.0 if top else cells)),
                        ("lstm_wgrad_ih", gates + xin), ("lstm_wgrad_hh", gates + hs)):
            per.setdefault(name, []).append(b)
        per.setdefault("lstm_wgrad_pair", []).append(gates + xin + hs)
        if k > 0:
            per.setdefault("lstm_dgrad", []).append(gates + cells)
    return per


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_cores():
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(cores, 16))


def cpu_baseline(kind, T, C, H, Ld, sample, budget_s=12.0):
    """The reference's CPU path (stock torch ops, fused oneDNN LSTM, autograd, Adam) restated by
    oracle/ref_torch.py, timed on this host on a bounded sample of the same workload: with Adam and without, on
    all the cores this process may use (at most 16, the box's share) and on one."""
    cores = host_cores()
    full = _cpu_baseline_run(kind, T, C, H, Ld, sample, cores, True, budget_s)
    full["cpu_model"] = cpu_model_name()
    full["fwd_bwd_elbo"] = _cpu_baseline_run(kind, T, C, H, Ld, sample, cores, False, budget_s / 2)["value"]
    one = _cpu_baseline_run(kind, T, C, H, Ld, min(sample, 64), 1, True, budget_s / 2)
    full["one_thread"] = {"value": one["value"], "unit": "windows/s", "cores": 1, "sample": one["sample"]}
    return full


def _cpu_baseline_run(kind, T, C, H, Ld, sample, cores, with_adam, budget_s):
    from oracle import ref_torch as R
    torch.set_num_threads(cores)
    sd = R.init_state(kind, C, H, Ld, 1, seed=0)
    g = torch.Generator().manual_seed(99)
    x = torch.randn(sample, T, C, generator=g)
    y = torch.randn(sample, 1, generator=g)
    m = {k: torch.zeros_like(v) for k, v in sd.items()}
    v = {k: torch.zeros_like(v_) for k, v_ in sd.items()}

    def one(step):
        noise = R.draw_noise(kind, sample, Ld, seed=step)
        _, _, grads = R.loss_and_grads(kind, sd, x, y, noise)
        if with_adam:
            R.adam_step(sd, grads, m, v, step, 1e-3)

    one(1)  # warm-up
    t0 = time.perf_counter()
    n = 0
    while True:
        one(n + 2)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 50:
            break
    return {"value": sample * n / el, "unit": "windows/s", "cores": cores, "kind": "port",
            "sample": f"{n} steps of {sample} windows x {T} x {C} (same model, fp32, torch {torch.__version__} CPU "
                      f"ops incl. fused oneDNN LSTM + autograd{' + Adam' if with_adam else ''}), {el:.1f} s"}


def elbo_hbm_roofline(dev, rows=1 << 20, Ld=32, iters=10):
    """HBM-bound side of the path: fused mixture reparameterisation + ELBO, forward and backward, on
    enough rows (2^20) for bandwidth to matter.  Algorithmic bytes per window (fp32, each tensor
    read or written once per pass): forward reads mean, logvar, scale, shape, eps, u (6*4L) +
    choice (4) + recon, y (8), writes z (4L); backward reads dz and the same six (7*4L), writes
    four latent gradients (4*4L) + d recon (4): 72*L + 12 + 4 = 2320 B at L = 32."""
    import vts
    g = torch.Generator(device=dev).manual_seed(7)
    mk = lambda *shape: torch.randn(*shape, device=dev, generator=g)
    mean, logvar = mk(rows, Ld).requires_grad_(True), (mk(rows, Ld) * 0.5).requires_grad_(True)
    scale = torch.exp(mk(rows, Ld) * 0.3).requires_grad_(True)
    shape = (mk(rows, Ld).abs() * 0.2 + 0.05).requires_grad_(True)
    eps, u = mk(rows, Ld), torch.rand(rows, Ld, device=dev, generator=g)
    choice = torch.rand(rows, device=dev, generator=g)
    pi = torch.tensor([3.0, 0.01], device=dev, requires_grad=True)
    recon, y = mk(rows, 1).requires_grad_(True), mk(rows, 1)
    dz = mk(rows, Ld)

    def once():
        # the path the models take: sampling fused with the KL sums, loss from those sums
        z, kl = vts.reparameterize_mixture(mean, logvar, scale, shape, pi, choice, eps, u, kl_threshold=0.8)
        loss = vts.mixture_elbo(recon, y, mean, logvar, scale, shape, pi, threshold=0.8, beta=0.001, kl=kl)
        torch.autograd.backward([z, loss], [dz, torch.ones((), device=dev)])
        for t in (mean, logvar, scale, shape, pi, recon):
            t.grad = None

    from vts import _lib as L
    once()
    torch.cuda.synchronize()
    L.timer.reset()
    L.timer.enabled = True
    for _ in range(iters):
        once()
    torch.cuda.synchronize()
    L.timer.enabled = False
    summ = L.timer.summary()
    L.timer.reset()
    per = {k: round(v[1] / v[0], 4) for k, v in summ.items()}
    ms = sum(per.values())
    bytes_alg = rows * (72 * Ld + 16)
    gbps = bytes_alg / (ms * 1e-3) / 1e9
    return {"kernels_ms": per, "family": "mixture", "rows": rows, "ms": round(ms, 4), "bound": "hbm",
            "achieved": round(gbps, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": round(gbps / PEAK_HBM_GBPS, 4),
            "note": "sum of the four entry points' device time (HIP events on the launch stream); "
                    "algorithmic bytes = rows * (72*L + 16)"}


def window_builder_hbm_roofline(dev, B, T, C, Nt=35040, iters=10, cpu=True):
    """The step in front of the train step: one batch of standardised windows gathered on the device from a
    series resident in HBM (one year of 15-minute LD2011-shaped load, Nt x C).  Algorithmic bytes per batch:
    the windows written once (B*T*C*4) + the series read once (Nt*C*4) + the targets.  Beside it, the
    reference's host path for the same batch (numpy window copies + StandardScaler.transform), timed on a
    bounded sample of windows, and the host-to-device copy it would need."""
    import vts
    g = torch.Generator(device=dev).manual_seed(11)
    series = torch.randn(Nt, C, device=dev, generator=g) * 3.0 + 1.0
    src = vts.WindowSource(series, T, 1)
    starts = torch.randint(0, len(src), (B,), device=dev, generator=g)
    src.fit(starts)
    src.batch(starts)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        src.batch(starts)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    bytes_alg = 4.0 * (B * T * C + Nt * C + B)
    gbps = bytes_alg / (ms * 1e-3) / 1e9
    out = {"workload": f"{B} windows x {T} steps x {C} channels gathered and standardised from a {Nt} x {C} series",
           "ms": round(ms, 4), "windows_per_s": round(B / (ms * 1e-3), 1), "bound": "hbm", "achieved": round(gbps, 1),
           "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": round(gbps / PEAK_HBM_GBPS, 4),
           "note": "HIP events on the launch stream around vts_windows_build (X and y kernels); algorithmic bytes = "
                   "4*(B*T*C + Nt*C + B)"}
    if cpu:
        from oracle import ref_windows as RW
        sample = min(B, 512)
        sn, st = series.cpu().numpy(), starts[:sample].cpu().numpy()
        mean, scale = src.x_stats[0].cpu().numpy(), src.x_stats[1].cpu().numpy()
        t0 = time.perf_counter()
        X, _ = RW.sliding_windows(sn, T, 1, st)
        Xs = RW.scaler_transform(X, mean, scale)
        xt = torch.from_numpy(Xs).to(dev)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(sample / el, 1), "unit": "windows/s", "cores": 1, "kind": "port",
                               "sample": f"{sample} windows: numpy window gather + float64 scaler transform + upload, {el:.2f} s"}
        del xt
    return out


def transformer_flops(B, T, C, d, layers, dff=2048):
    """Dense FLOPs of one train step of the transformer variant as the reference states it (every
    position through every layer): forward 2*rows*N*K per product, backward twice that (data and
    weight gradients; the embedding's input needs none), attention 4*T*T*d forward, 2.5x backward."""
    rows = B * T
    fwd = 2.0 * rows * C * d
    bwd = 2.0 * rows * C * d
    per_layer = 2.0 * rows * d * 3 * d + 2.0 * rows * d * d + 4.0 * rows * d * dff
    attn = 4.0 * B * T * T * d
    fwd += layers * (per_layer + attn)
    bwd += layers * (2 * per_layer + 2.5 * attn)
    return fwd + bwd


def cpu_baseline_transformer(T, C, d, nhead, layers, Ld, sample, budget_s=20.0):
    """The reference's CPU path for the transformer variant restated by oracle/ref_transformer.py
    (stock torch CPU ops + autograd + Adam) on a bounded sample of the same workload."""
    from oracle import ref_torch as R
    from oracle import ref_transformer as RT
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 16))
    torch.set_num_threads(cores)
    sd = RT.init_state(C, d, layers, Ld, 1, seed=0)
    g = torch.Generator().manual_seed(99)
    x = torch.randn(sample, T, C, generator=g)
    y = torch.randn(sample, 1, generator=g)
    thr = float(y.abs().median())
    m = {k: torch.zeros_like(v) for k, v in sd.items()}
    v = {k: torch.zeros_like(v_) for k, v_ in sd.items()}

    def one(step):
        noise = RT.draw_noise(sample, T, d, nhead, layers, Ld, seed=step)
        _, _, grads = RT.loss_and_grads(sd, x, y, noise, nhead, thr)
        R.adam_step(sd, grads, m, v, step, 1e-3)

    one(1)
    t0 = time.perf_counter()
    n = 0
    while True:
        one(n + 2)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 50:
            break
    return {"value": sample * n / el, "unit": "windows/s", "cores": cores, "kind": "port",
            "sample": f"{n} steps of {sample} windows x {T} x {C} (same model, fp32, torch {torch.__version__} CPU "
                      f"ops + autograd + Adam, every position through every layer as the reference does), {el:.1f} s"}


def bench_transformer(a, dev, world, rank, launched):
    """--model transformer: the transformer-encoder mixture VAE (vae-models/mixture-vae-transformer.py) on the
    same LD2011-shaped windows.  Same step definition and JSON contract as the LSTM models."""
    import vts
    from vts import _lib as L

    vts.set_mode(a.mode)
    torch.manual_seed(1234)
    model = vts.TransformerMixtureVAE(a.C, a.d_model, a.layers, a.nhead, a.L, 1).to(dev)
    model.train()
    trainer = vts.DataParallelTrainer(model, lr=1e-3, exact_loss=a.exact_loss)
    torch.manual_seed(4321 + rank)
    x, y = synth_windows(a.batch, a.T, a.C, rank, dev)
    model.threshold = float(y.abs().median())    # both selections of the loss populated on every rank

    def sync():
        if launched:
            dist.barrier()
        torch.cuda.synchronize()

    # Every bracketed call puts two event markers into the queue, and the device idles a few microseconds at each: with
    # all ~90 entry-point calls of a step bracketed the step is 0.5 ms longer.  So the timed region brackets only the
    # dominant entry point (picked here, from the last warm-up step with everything bracketed); the table of all
    # kernels comes from a pass of its own after the timed loops.
    def product_flops(name):
        """dense product FLOPs of an entry point named '<kind> MxNxK ...' (None for the others)"""
        if not (name.startswith("linear_") or name.startswith("ffn_")):
            return None
        M, N, K = [int(v) for v in name.split()[1].split("x")]
        mult = 2.0 if name.startswith("linear_fwd") or name.endswith("nodx") else 4.0
        if name.startswith("ffn_"):     # two products forward, four backward
            mult = 4.0 if name.startswith("ffn_fwd") else 8.0
        return mult * M * N * K

    dom = None
    for i in range(a.warmup):
        if i == a.warmup - 1:
            sync()
            L.timer.only = None
            L.timer.enabled = True
            L.timer.reset()
        trainer.step(x, y)
    if a.warmup:
        sync()
        L.timer.enabled = False
        pre = L.timer.summary()
        dom = max((k for k in pre if product_flops(k) is not None), key=lambda k: pre[k][1])
    # a full collection now, so that the interpreter's next generation-2 pass (~30 ms over torch's
    # heap, due once per few hundred steps) does not land inside a ten-step timed region
    gc.collect()
    sync()
    L.timer.only = {dom} if dom is not None else None
    L.timer.enabled = True
    L.timer.reset()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = trainer.step(x, y)
    sync()
    elapsed = time.perf_counter() - t0
    L.timer.enabled = False
    timed = L.timer.summary()
    if launched:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    final_loss = loss.item()
    if not math.isfinite(final_loss):
        raise SystemExit(f"bench.py: the loss after {a.steps} steps is {final_loss}: no throughput is reported")
    if dom is not None:     # every entry point bracketed: the kernel table, a pass of its own
        sync()
        L.timer.only = None
        L.timer.enabled = True
        L.timer.reset()
        for _ in range(a.steps):
            trainer.step(x, y)
        sync()
        L.timer.enabled = False
    if rank != 0:
        return
    ms_per_step = 1e3 * elapsed / a.steps
    value = world * a.batch * a.steps / elapsed
    summ = L.timer.summary()
    if dom is None:
        dom = max((k for k in summ if product_flops(k) is not None), key=lambda k: summ[k][1])
    summ[dom] = timed[dom]      # the dominant entry point: its launches inside the timed region
    kernels = {}
    for name, (n, tot_ms) in summ.items():
        avg_ms = tot_ms / n
        kernels[name] = {"launches_per_step": n // a.steps, "avg_ms": round(avg_ms, 4),
                         "share_of_step": round(tot_ms / a.steps / ms_per_step, 4)}
        if product_flops(name) is not None:
            kernels[name]["tflops"] = round(product_flops(name) / (avg_ms * 1e-3) / 1e12, 2)
    kernels[dom]["measured"] = "HIP events inside the timed region"
    dom_tflops = kernels[dom]["tflops"]
    split = a.mode == "bf16x3"
    roofline = {"bound": "mfma", "kernel": dom, "achieved": round((3 if split else 1) * dom_tflops, 2),
                "peak": PEAK_BF16_MFMA_TFLOPS if split else PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "traffic": None,
                "algorithmic_tflops": dom_tflops,
                "note": "dense product FLOPs of the entry point (2*M*N*K per product: linear fwd 1, bwd 2; feed-forward block "
                        "fwd 2, bwd 4) / its "
                        "device time (HIP events on the launch stream)" +
                        ("; three bf16 matrix-core products are executed per fp32 product" if split else "")}
    roofline["frac"] = round(roofline["achieved"] / roofline["peak"], 4)
    if split and dom.startswith("ffn_bwd"):
        # the bf16x3 backward pass of the block keeps no [tokens, dim_feedforward] tensor: it works hidden and its
        # gradient out again inside the weight-gradient kernel, two products on top of the four that are counted
        roofline["executed_with_recomputation"] = {"achieved": round(roofline["achieved"] * 1.5, 2), "unit": "TFLOP/s",
                                                   "frac": round(roofline["achieved"] * 1.5 / roofline["peak"], 4),
                                                   "note": "six products run (vts_ffn_bwd_recompute); frac above counts the four algorithmic ones"}
    total = transformer_flops(a.batch, a.T, a.C, a.d_model, a.layers)
    out = {
        "metric": "train windows/sec (fwd+bwd+ELBO, Adam update included)", "value": round(value, 1), "unit": "windows/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if not split else "bf16x3 (fp32 operands split into two bf16, three bf16 MFMA products per "
                                         "fp32 product, fp32 accumulate) in the large Linear products; f32 elsewhere",
        "data": "synthetic",
        "config": {"workload": f"transformer-encoder mixture VAE train step, LD2011-shaped windows: {a.batch} windows/GPU x "
                               f"{a.T} steps x {a.C} channels, d_model {a.d_model}, {a.nhead} heads, {a.layers} layers, "
                               f"dim_feedforward 2048, latent {a.L}, future_steps 1",
                   "model_kind": "transformer", "windows_per_gpu": a.batch, "global_batch": a.batch * world,
                   "parallelism": f"dp{world}", "rccl_ranks": world if launched else 0, "gemm_mode": a.mode,
                   "optimizer": "Adam(lr=1e-3) inside the timed step",
                   "noise": "sampling noise drawn on device every step; dropout masks from the counter hash, fresh seeds every step",
                   "last_layer": "final encoder layer evaluated for the last position only (the only one the model reads)"},
        "roofline": roofline,
        "step_mfma": {"reference_algorithm_tflop_per_step": round(total / 1e12, 4),
                      "note": "FLOPs of the step as the reference runs it (all positions through all layers); the "
                              "last-position-only final layer executes about half of them"},
        "kernels": kernels,
        "kernels_note": "HIP events around every entry point, in a pass of its own after the timed loop (the dominant "
                        "entry point: inside the timed region)",
        "final_loss": final_loss,
    }
    if not a.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline_transformer(a.T, a.C, a.d_model, a.nhead, a.layers, a.L,
                                                       min(a.cpu_sample, 64))
    emit(json.dumps(out))


_RESULT_FD = None


def quiet_stdout():
    """Everything but the result line goes to stderr: libraries (RCCL's version banner at communicator set-up,
    for one) print to the process's stdout, which the caller reads as ONE JSON line."""
    global _RESULT_FD
    sys.stdout.flush()
    _RESULT_FD = os.dup(1)
    os.dup2(2, 1)


def emit(line: str) -> None:
    sys.stdout.flush()
    if _RESULT_FD is None:
        print(line, flush=True)
    else:
        os.write(_RESULT_FD, (line + "\n").encode())


def launch_ranks(a):
    """bench.py --gpus N (N > 1) started without torch.distributed.run: this process, which has not touched the GPU,
    starts one fresh child per rank with the rendezvous environment set, relays rank 0's line and exits with the
    worst child status."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [pr.wait() for pr in procs]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    sys.exit(max(abs(rc) for rc in rcs))


# kernel of each timed entry point whose HBM traffic --traffic reports (regular expression searched in the rocprofv3
# kernel name; the backward recurrence has a pair, a cluster and two one-workgroup kernels)
TRAFFIC_KERNELS = {"lstm_recurrent_bwd": "lstm_rec3[pc]?_bwd", "lstm_recurrent_fwd": "lstm_rec3c_fwd_kernel",
                   "lstm_wgrad_pair": "wgrad_planes_kernel", "lstm_inproj_fwd": "rowgemm_planes_kernel",
                   "lstm_dgrad": "rowgemm_planes_kernel"}


def measure_traffic(a):
    """HBM-side bytes per launch from the L2's memory-side request counters: two short runs of this script as child
    processes under `rocprofv3 --pmc` (one counter per run: the two do not fit one pass), the interpreter directly
    behind `--`.  FETCH_SIZE and WRITE_SIZE are KiB; on gfx950 FETCH_SIZE tallies 128-byte read requests at 64 bytes,
    so it is doubled (that correction holds for 16-byte-per-lane streaming loads, which is what every large kernel
    of this step issues; WRITE_SIZE is exact for 16-byte-per-lane stores).  Returns kernel-name-substring ->
    {FETCH_SIZE_KiB, WRITE_SIZE_KiB, hbm_bytes, launches} averaged over the launches of the run, or an error note."""
    import csv
    import glob
    import shutil
    import tempfile
    if shutil.which("rocprofv3") is None:
        return {"error": "rocprofv3 not on PATH"}
    top = tempfile.mkdtemp(prefix="vts_pmc_", dir="/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--traffic-child", "--steps", "2", "--warmup", "1",
             "--no-cpu-baseline", "--model", a.model, "--shape", a.shape, "--batch", str(a.batch), "--T", str(a.T), "--C", str(a.C),
             "--H", str(a.H), "--L", str(a.L), "--mode", a.mode]
    sums = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(top, counter)
        cmd = ["rocprofv3", "--pmc", counter, "-d", d, "-o", "t", "--output-format", "csv", "--"] + child
        try:
            rc = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                                stderr=subprocess.DEVNULL, timeout=240).returncode
        except subprocess.TimeoutExpired:
            return {"error": f"rocprofv3 --pmc {counter} timed out"}
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:      # (the profiled interpreter may crash while exiting, after the counters are on disk)
            return {"error": f"rocprofv3 --pmc {counter} left no counter file (exit {rc})"}
        per = {}
        for row in csv.DictReader(open(files[0])):
            if row.get("Counter_Name") != counter:
                continue
            key = (row["Kernel_Name"], row["Dispatch_Id"])
            per[key] = per.get(key, 0.0) + float(row["Counter_Value"])
        for (kname, did), v in sorted(per.items(), key=lambda kv: int(kv[0][1])):     # in dispatch order
            for sub in set(TRAFFIC_KERNELS.values()):
                if re.search(sub, kname):
                    ent = sums.setdefault(sub, {"FETCH_SIZE": [], "WRITE_SIZE": []})
                    ent[counter].append(v)
    shutil.rmtree(top, ignore_errors=True)

    def entry(fs, ws):
        f, w = sum(fs) / len(fs), sum(ws) / len(ws)
        return {"FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1), "hbm_bytes": int((2 * f + w) * 1024),
                "launches": len(fs)}

    out = {}
    for sub, ent in sums.items():
        if ent["FETCH_SIZE"] and ent["WRITE_SIZE"]:
            out[sub] = entry(ent["FETCH_SIZE"], ent["WRITE_SIZE"])
    # The input projections and the data gradients are launches of one kernel; the order of its launches inside a
    # step tells them apart: one projection per layer going up, then one data gradient per layer but the first going down.
    ent = sums.get("rowgemm_planes_kernel")
    layers = 1 if a.model == "gaussian" else 3
    per_step = 2 * layers - 1
    if ent and len(ent["FETCH_SIZE"]) == len(ent["WRITE_SIZE"]) and len(ent["FETCH_SIZE"]) % per_step == 0:
        sites = {"inproj_layer0": [0], "inproj_upper_layers": list(range(1, layers)), "dgrad": list(range(layers, per_step))}
        reads = {"inproj_layer0": 4.0 * a.batch * a.T * ((a.C + 63) // 64 * 64), "inproj_upper_layers": 4.0 * a.batch * a.T * a.H,
                 "dgrad": 4.0 * a.batch * a.T * 4 * a.H}
        by_site = {}
        for site, pos in sites.items():
            idx = [i for i in range(len(ent["FETCH_SIZE"])) if i % per_step in pos]
            if idx:
                e = entry([ent["FETCH_SIZE"][i] for i in idx], [ent["WRITE_SIZE"][i] for i in idx])
                e["fetched_bytes"] = int(2 * e["FETCH_SIZE_KiB"] * 1024)
                e["algorithmic_read_bytes"] = int(reads[site])      # the plane operand (the weights stay in the L2)
                e["fetch_over_algorithmic_reads"] = round(e["fetched_bytes"] / reads[site], 3)
                by_site[site] = e
        out["rowgemm_planes_kernel"]["by_call_site"] = by_site
        out["rowgemm_planes_kernel"]["by_call_site_note"] = (
            "launch order inside a step: one projection per layer, then one data gradient per layer but the first; "
            "algorithmic reads = the plane operand once (hi + lo planes = 4 bytes per padded value)")
    return out


def main():
    a = parse()
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ   # under torch.distributed.run, or a child of launch_ranks
    if a.gpus > 1 and not launched:
        launch_ranks(a)
    traffic = None
    if a.traffic and not launched and not a.traffic_child:
        traffic = measure_traffic(a)        # child processes; this one has not touched the GPU yet
    quiet_stdout()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the vts kernels have no CPU path")
    if launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    import vts
    from vts import _lib as L

    if a.model == "transformer":
        bench_transformer(a, dev, world, rank, launched)
        if launched:
            dist.barrier()
            dist.destroy_process_group()
        return
    vts.set_mode(a.mode)
    torch.manual_seed(1234)                     # identical replicas on every rank
    model = vts.build_model(a.model, a.C, a.H, a.L, 1).to(dev)
    model.train()
    if a.model == "mixture3":                   # a threshold inside the data and some exact zeros: all three loss terms live
        model.threshold = 0.8
    trainer = vts.DataParallelTrainer(model, lr=1e-3, exact_loss=a.exact_loss)
    torch.manual_seed(4321 + rank)              # rank-offset noise streams
    if a.shape == "hurricane":
        if a.C != 1:
            raise SystemExit("--shape hurricane is a single-channel series (C = 1)")
        x, y = synth_hurricane(a.batch, rank, dev, a.T)
    else:
        x, y = synth_windows(a.batch, a.T, a.C, rank, dev)
    if a.model == "mixture3":
        y[::16] = 0.0

    def sync():
        if launched:
            dist.barrier()
        torch.cuda.synchronize()

    # Every bracketed call puts two event markers into the queue, and the device idles a few microseconds at each: with
    # all ~90 entry-point calls of a step bracketed the step is 0.5 ms longer.  So the timed region brackets only the
    # dominant entry point (picked here, from the last warm-up step with everything bracketed); the table of all
    # kernels comes from a pass of its own after the timed loops.
    per, total_flops = lstm_flops(a.model, a.batch, a.T, a.C, a.H)
    dom = "lstm_recurrent_bwd"
    for i in range(a.warmup):
        if i == a.warmup - 1:
            sync()
            L.timer.only = None
            L.timer.enabled = True
            L.timer.reset()
        trainer.step(x, y)
    if a.warmup:
        sync()
        L.timer.enabled = False
        pre = L.timer.summary()
        dom = max((k for k in pre if k in per), key=lambda k: pre[k][1])
    # a full collection now, so that the interpreter's next generation-2 pass (~30 ms over torch's
    # heap, due once per few hundred steps) does not land inside a ten-step timed region
    graphed = None
    if a.graph:
        if launched or world > 1:
            raise SystemExit("--graph records a single-process step (no gradient all-reduce inside the graph)")
        graphed = vts.GraphedTrainStep(model, trainer.opt, x, y)
        for _ in range(max(a.warmup, 1)):
            graphed(x, y)
    gc.collect()
    sync()
    if graphed is None:             # (no event brackets inside a replay: the kernel pass below times the dominant one)
        L.timer.only = {dom}
        L.timer.enabled = True
        L.timer.reset()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = graphed(x, y) if graphed is not None else trainer.step(x, y)
    sync()
    elapsed = time.perf_counter() - t0
    L.timer.enabled = False
    dom_n, dom_total_ms = L.timer.summary()[dom] if graphed is None else (0, 0.0)
    if launched:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    final_loss = loss.item()
    if not math.isfinite(final_loss):
        raise SystemExit(f"bench.py: the loss after {a.steps} steps is {final_loss}: no throughput is reported for a step "
                         "that does not compute a finite loss")
    if a.traffic_child:
        return
    # the same step without the Adam update (forward, ELBO, backward, gradient all-reduce): the metric's first figure
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        trainer.step(x, y, apply_update=False)
    sync()
    elapsed_noupd = time.perf_counter() - t0
    if launched:
        t = torch.tensor([elapsed_noupd], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed_noupd = t.item()

    # every entry point bracketed: the kernel table (a pass of its own, 0.5 ms per step slower than the timed one)
    sync()
    L.timer.only = None
    L.timer.enabled = True
    L.timer.reset()
    for _ in range(a.steps):
        trainer.step(x, y)
    sync()
    L.timer.enabled = False

    if rank == 0:
        ms_per_step = 1e3 * elapsed / a.steps
        value = world * a.batch * a.steps / elapsed
        summ = L.timer.summary()
        kernels = {}
        for name, (n, tot_ms) in summ.items():
            avg_ms = tot_ms / n
            kernels[name] = {"launches_per_step": n // a.steps, "avg_ms": round(avg_ms, 4),
                             "share_of_step": round(tot_ms / a.steps / ms_per_step, 4)}
            if name in per:   # MFMA kernels: mean algorithmic FLOPs per launch of this name
                flops = sum(per[name]) / len(per[name])
                kernels[name]["tflops"] = round(flops / (avg_ms * 1e-3) / 1e12, 2)
        # the dominant entry point: its launches inside the timed region (a recorded step has none: the figures of
        # the eager kernel pass stand)
        if dom_n > 0:
            dom_avg = dom_total_ms / dom_n
            kernels[dom] = {"launches_per_step": dom_n // a.steps, "avg_ms": round(dom_avg, 4),
                            "share_of_step": round(dom_total_ms / a.steps / ms_per_step, 4),
                            "tflops": round(sum(per[dom]) / len(per[dom]) / (dom_avg * 1e-3) / 1e12, 2),
                            "measured": "HIP events inside the timed region"}
        dom_tflops = kernels[dom]["tflops"]
        # Which roofline bounds the dominant kernel: the larger of its two floors, algorithmic FLOPs (x3 executed
        # bf16 products in bf16x3 mode) / matrix peak and algorithmic HBM bytes / HBM peak.  The fraction
        # against the other one is reported beside it.
        hbm_per = lstm_hbm_bytes(a.model, a.batch, a.T, a.C, a.H)
        dom_bytes = sum(hbm_per[dom]) / len(hbm_per[dom])
        dom_ms = kernels[dom]["avg_ms"]
        dom_gbps = dom_bytes / (dom_ms * 1e-3) / 1e9
        exec_mult, mfma_peak = (1, PEAK_F32_MFMA_TFLOPS) if a.mode == "f32" else (3, PEAK_BF16_MFMA_TFLOPS)
        mfma_view = {"achieved": round(exec_mult * dom_tflops, 2), "peak": mfma_peak, "unit": "TFLOP/s",
                     "frac": round(exec_mult * dom_tflops / mfma_peak, 4), "algorithmic_tflops": dom_tflops,
                     "note": "executed matrix-core FLOPs (bf16x3: 3 per algorithmic fp32 FLOP) / kernel time"}
        hbm_view = {"achieved": round(dom_gbps, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                    "frac": round(dom_gbps / PEAK_HBM_GBPS, 4), "algorithmic_bytes_per_launch": dom_bytes}
        floor_mfma_ms = exec_mult * dom_tflops * dom_ms / mfma_peak
        floor_hbm_ms = dom_bytes / (PEAK_HBM_GBPS * 1e9) * 1e3
        if floor_hbm_ms >= floor_mfma_ms:
            roofline = {"bound": "hbm", "kernel": dom, **hbm_view, "traffic": None, "mfma_view": mfma_view}
        else:
            roofline = {"bound": "mfma", "kernel": dom, **mfma_view, "traffic": None, "hbm_view": hbm_view}
        roofline["kernel_ms"] = dom_ms
        roofline["floors_ms"] = {"mfma": round(floor_mfma_ms, 4), "hbm": round(floor_hbm_ms, 4)}
        # HBM bytes per launch of the dominant kernel: measured by this invocation (--traffic), else null with the
        # path of the committed measurement of the same command
        roofline["traffic_profile"] = "profiles/r03_pmc_hbm_traffic.json"
        if traffic is not None:
            roofline["traffic_measured"] = traffic
            sub = TRAFFIC_KERNELS.get(dom)
            site = {"lstm_inproj_fwd": None, "lstm_dgrad": "dgrad"}.get(dom, "")     # those two share one kernel
            if sub in traffic and site == "":
                roofline["traffic"] = traffic[sub]["hbm_bytes"]
                roofline["traffic_note"] = ("HBM bytes per launch, rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE (KiB), mean "
                                            "over the launches of a 3-step run of this command made by this invocation")
        # The recurrent kernels re-read the packed W_hh image (4*H*H*4 bytes, L2-resident) once per
        # time step and workgroup; with the matrix-core time cut to 3/16 that stream is what bounds them.
        l2 = {}
        # (The forward cluster kernel - bf16x3, hidden 256, batch a multiple of 128 - keeps its W_hh slice in
        # LDS and moves h through the L2 instead: per step and workgroup 128 KiB read + 16 KiB written.)
        cluster_fwd = (a.mode == "bf16x3" and a.H == 256 and a.batch % 128 == 0
                       and os.environ.get("VTS_LSTM_CLUSTER", "1")[:1] != "0")
        # (resident: the share of the image a workgroup keeps on chip - an eighth in LDS forward; backward an eighth
        # in registers, in the hidden-256 bf16x3 kernel only)
        bwd_resident = 0.125 if (a.mode == "bf16x3" and a.H == 256 and os.environ.get("VTS_LSTM_STAGED_BWD", "1")[:1] != "0") else 0.0
        pair_bwd = (a.mode == "bf16x3" and a.H == 256 and a.batch % 32 == 0 and a.batch // 16 <= 256
                    and os.environ.get("VTS_LSTM_PAIR_BWD", "1")[:1] != "0" and os.environ.get("VTS_LSTM_CLUSTER_BWD", "0")[:1] != "1")
        for name, resident in (("lstm_recurrent_fwd", 0.125 if a.H >= 128 else 0.0), ("lstm_recurrent_bwd", bwd_resident)):
            if name in kernels:
                nwg = (a.batch + 15) // 16
                byts = (a.T - 1) * nwg * 4.0 * a.H * a.H * 4 * (1.0 - resident)
                if name == "lstm_recurrent_fwd" and cluster_fwd:
                    byts = (a.T - 1) * nwg * (128.0 + 16.0) * 1024
                # (The backward pair kernel - bf16x3, hidden 256, batch a multiple of 32, at most one workgroup per CU -
                # has one workgroup per 16 windows as well, but a member streams only its half of the image, a
                # quarter of that half resident in registers, and the members swap 16 KiB of partial sums a step.)
                if name == "lstm_recurrent_bwd" and pair_bwd:
                    byts = (a.T - 1) * nwg * (0.75 * 512.0 + 2 * 16.0) * 1024
                gbps = byts / (kernels[name]["avg_ms"] * 1e-3) / 1e9
                l2[name] = {"bytes_per_launch": byts, "achieved_GBps": round(gbps, 1),
                            "frac_of_l2_peak": round(gbps / PEAK_L2_GBPS, 4)}
        out = {
            "metric": "train windows/sec (fwd+bwd+ELBO, Adam update included)", "value": round(value, 1), "unit": "windows/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "value_with_adam": round(value, 1),
            "value_fwd_bwd_elbo": round(world * a.batch * a.steps / elapsed_noupd, 1),
            "ms_per_step_fwd_bwd_elbo": round(1e3 * elapsed_noupd / a.steps, 3),
            **({"fwd_bwd_elbo_note": "launched one by one (only the full step is recorded as a graph)"} if a.graph else {}),
            "dtype": "f32" if a.mode == "f32" else "bf16x3 (fp32 operands split into two bf16, three bf16 MFMA "
                                                    "products per fp32 product, fp32 accumulate); f32 elsewhere",
            "data": "synthetic",
            "config": {"workload": f"{a.model} LSTM-VAE train step, "
                                   f"{'LD2011-shaped' if a.shape == 'ld2011' else 'hurricane-shaped (AR(1) walk, block-maximum target)'} "
                                   f"windows: {a.batch} windows/GPU x "
                                   f"{a.T} steps x {a.C} channels, hidden {a.H}, latent {a.L}, future_steps 1",
                       "model_kind": a.model, "windows_per_gpu": a.batch, "global_batch": a.batch * world,
                       "parallelism": f"dp{world}", "rccl_ranks": world if launched else 0, "gemm_mode": a.mode,
                       "optimizer": "Adam(lr=1e-3) inside the timed step", "noise": "drawn on device every step",
                       "launch": "one recorded HIP graph per step (vts.GraphedTrainStep)" if a.graph else "eager"},
            "roofline": roofline,
            "recurrent_l2_stream": {"peak_GBps": PEAK_L2_GBPS, "kernels": l2,
                                    "note": "W_hh image re-read from the XCD L2 by every workgroup every step "
                                            "(forward cluster kernel: h exchange through the L2 instead; backward pair "
                                            "kernel: half the image per workgroup, three quarters of it streamed)"},
            "step_mfma": {"algorithmic_tflop_per_step": round(total_flops / 1e12, 4),
                          "achieved_tflops": round(total_flops / (ms_per_step * 1e-3) / 1e12, 2),
                          "executed_tflops": round(exec_mult * total_flops / (ms_per_step * 1e-3) / 1e12, 2),
                          "frac_of_matrix_peak": round(exec_mult * total_flops / (ms_per_step * 1e-3) / 1e12 / mfma_peak, 4),
                          "note": "whole step (LSTM products only) against the matrix peak of the arithmetic in use"},
            "kernels": kernels,
            "kernels_note": "HIP events around every entry point, in a pass of its own after the timed loops (the "
                            "dominant entry point: inside the timed region)",
            "final_loss": final_loss,
        }
        if world == 1:
            out["elbo_hbm_roofline"] = elbo_hbm_roofline(dev)
            out["window_builder_hbm_roofline"] = window_builder_hbm_roofline(dev, a.batch, a.T, a.C,
                                                                             cpu=not a.no_cpu_baseline)
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(a.model, a.T, a.C, a.H, a.L, a.cpu_sample)
        emit(json.dumps(out))
    if launched:
        dist.barrier()
        dist.destroy_process_group()


def dump_maps():
    """VTS_BENCH_DUMP_MAPS=<file>: the process's memory map at the end of the run, so that the addresses of a crash
    report written while the interpreter exits (a profiler's, say) can be matched to the libraries they lie in."""
    path = os.environ.get("VTS_BENCH_DUMP_MAPS")
    if path:
        with open("/proc/self/maps") as src, open(path, "w") as dst:
            dst.write(src.read())


if __name__ == "__main__":
    main()
    dump_maps()

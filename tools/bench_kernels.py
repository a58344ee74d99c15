#!/usr/bin/env python3
"""Time the LSTM stage kernels one by one through the C ABI (HIP events on the launch stream).

    python tools/bench_kernels.py [--B 4096 --T 96 --C 370 --H 256 --iters 5] [--lib path/to/lib.so]

Prints one line per stage: average ms, algorithmic TFLOP/s and fraction of the f32 matrix peak.
A development tool for kernel tuning; bench.py is the graded benchmark."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vae-time-series_amd"))

import torch

PEAK = 157.3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=4096)
    ap.add_argument("--T", type=int, default=96)
    ap.add_argument("--C", type=int, default=370)
    ap.add_argument("--H", type=int, default=256)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--lib", default=None)
    ap.add_argument("--only", default="")
    ap.add_argument("--mode", type=int, default=0)
    a = ap.parse_args()
    from vts import _lib as L
    if a.lib:
        L.LIB_PATH = os.path.abspath(a.lib)
    lib = L.load()
    dev = torch.device("cuda:0")
    B, T, Cin, H = a.B, a.T, a.C, a.H
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn(B, T, Cin, generator=g).to(dev)
    w_ih = (torch.rand(4 * H, Cin, generator=g) * 2 - 1).mul_(1 / 16).to(dev)
    w_hh = (torch.rand(4 * H, H, generator=g) * 2 - 1).mul_(1 / 16).to(dev)
    b = torch.zeros(4 * H, device=dev)
    gates = L.workspace(lib.vts_lstm_gates_bytes(B, T, H), dev)
    cells = L.workspace(lib.vts_lstm_cells_bytes(B, T, H), dev)
    hseq = torch.empty(B, T, H, device=dev)
    h_last = torch.empty(B, H, device=dev)
    dh_last = torch.randn(B, H, device=dev)
    dx = L.workspace(lib.vts_lstm_seqgrad_bytes(B, T, H), dev)
    dw_ih = torch.empty_like(w_ih)
    dw_hh = torch.empty_like(w_hh)
    w_ih2 = (torch.rand(4 * H, H, generator=g) * 2 - 1).mul_(1 / 16).to(dev)
    db = torch.empty(4 * H, device=dev)
    rws_b = lib.vts_lstm_recurrent_workspace_bytes(B, T, H)
    rws = L.workspace(rws_b, dev)
    wg_b = max(lib.vts_lstm_wgrad_workspace_bytes(B, T, Cin, H), lib.vts_lstm_wgrad_workspace_bytes(B, T, H, H))
    wg = L.workspace(wg_b, dev)
    st = L.stream()
    MODE = a.mode
    inproj_f = 2.0 * B * T * Cin * 4 * H
    rec_f = 2.0 * B * T * H * 4 * H

    stages = [
        ("inproj_fwd", inproj_f, lambda: lib.vts_lstm_inproj_fwd(L.ptr(x), L.ptr(w_ih), L.ptr(b), L.ptr(b),
                                                                   L.wptr(gates), B, T, Cin, H, MODE, st)),
        ("recurrent_fwd", rec_f, lambda: lib.vts_lstm_recurrent_fwd(L.wptr(gates), L.wptr(cells), L.ptr(hseq),
                                                                      L.ptr(h_last), None, L.ptr(w_hh), L.wptr(rws),
                                                                      rws_b, B, T, H, MODE, st)),
        ("recurrent_bwd", rec_f, lambda: lib.vts_lstm_recurrent_bwd(L.wptr(gates), L.wptr(cells), None,
                                                                      L.ptr(dh_last), None, L.ptr(w_hh), L.ptr(db),
                                                                      L.ptr(db), 0, L.wptr(rws), rws_b, B, T, H, MODE, st)),
        ("dgrad(H)", rec_f, lambda: lib.vts_lstm_dgrad(L.wptr(gates), L.ptr(w_ih2), L.wptr(dx), B, T, H, H, MODE, st)),
        ("wgrad_ih", inproj_f, lambda: lib.vts_lstm_wgrad(L.wptr(gates), L.ptr(x), 0, L.ptr(dw_ih), 0, L.wptr(wg),
                                                          wg_b, B, T, Cin, H, MODE, st)),
        ("wgrad_hh", rec_f, lambda: lib.vts_lstm_wgrad(L.wptr(gates), L.ptr(hseq), -1, L.ptr(dw_hh), 0, L.wptr(wg),
                                                       wg_b, B, T, H, H, MODE, st)),
    ]
    if lib.vts_lstm_wgrad_pair_supported(B, T, H, H, MODE):
        x2 = torch.randn(B, T, H, device=dev)       # a lower layer's h sequence as this layer's input
        dw_ih2 = torch.empty(4 * H, H, device=dev)
        stages.append(("wgrad_pair(H)", 2 * rec_f, lambda: lib.vts_lstm_wgrad_pair(
            L.wptr(gates), L.ptr(x2), L.ptr(hseq), L.ptr(dw_ih2), L.ptr(dw_hh), 0, L.wptr(wg), wg_b, B, T, H, H, MODE, st)))
    for name, flops, fn in stages:
        if a.only and a.only not in name:
            # still run once so later stages see initialised inputs
            L.check(fn(), name)
            continue
        L.check(fn(), name)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(a.iters):
            L.check(fn(), name)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / a.iters
        tf = flops / (ms * 1e-3) / 1e12
        print(f"{name:14s} {ms:8.3f} ms  {tf:7.1f} TFLOP/s  {100 * tf / PEAK:5.1f} % of f32 MFMA peak", flush=True)


if __name__ == "__main__":
    main()

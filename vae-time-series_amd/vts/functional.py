"""autograd.Functions over the vts C ABI: the torch call sites of the reference's train step
(nn.LSTM, nn.Linear+activation, nn.LayerNorm, reparameterize*, loss_function) as drop-in
differentiable ops whose forward and backward are HIP kernels.

Reference call sites (vae-models/): gaussian-lstm-DE.py:28-31,43-45,55-58,66-70 ;
gpd-lstm-DE.py:10-12,27-33,45-47,67-71 ; mixture-vae-DE.py:10-18,60-68,81-83,94-107,115-130.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import ctypes as C
import os

import torch

from . import _lib as L

DROP_P = 0.3
MODES = {"f32": 0, "bf16x3": 1}
_MODE = 0  # dtype_mode of the C ABI passed to the LSTM GEMM stages
# the encoder heads and the decoders as fused chains (vts_mlp_chain_*); VTS_FUSED_CHAINS=0 runs them layer by layer
_FUSED_CHAINS = os.environ.get("VTS_FUSED_CHAINS", "1")[:1] != "0"


# Beyond this many rows the layers run one by one on the tiled GEMMs: the chain's weight-gradient kernel re-reads its
# operand columns once per 16 x 16 output tile, which is nothing at 4096 rows (0.12 ms for the whole chain) and
# more than the GEMMs' split-K plans at 65 536 (measured: 9.4 ms per step fused against 8.4 ms layer by layer).
CHAIN_MAX_ROWS = 16384
# test aid: when a dict, mlp_chain stores the trunk outputs of every chain it runs under the names it was given
CHAIN_TAPS = None
# test aid: when a list, feed_forward appends the hidden activations dropout(relu(linear1(x))) of every block it runs
FFN_TAPS = None
# the feed-forward block's backward pass from x, dy and the packed mask alone where the library offers it
# (vts_ffn_bwd_recompute); False = keep the [tokens, dim_feedforward] activations as vts_ffn_bwd wants them
FFN_RECOMPUTE = True


def set_fused_chains(on: bool) -> None:
    """Run the encoder heads and the decoders as fused chains (default) or layer by layer."""
    global _FUSED_CHAINS
    _FUSED_CHAINS = bool(on)



def set_mode(name: str) -> None:
    """Arithmetic of the LSTM's dense products (input projection, recurrent product, data and
    weight gradients):
    "f32"    - exact fp32 on the f32 matrix cores (v_mfma_f32_16x16x4_f32);
    "bf16x3" - each fp32 operand split into two bf16 values, three bf16 MFMAs per product with
               fp32 accumulation: fp32-level accuracy (~2^-16 relative per product) at a
               fraction of the f32 MFMA cost.  Hidden sizes below 64 keep the fp32 recurrence.
    The cell update, heads, reparameterisation and ELBO are fp32 in both modes."""
    global _MODE
    _MODE = MODES[name]


def get_mode() -> str:
    return [k for k, v in MODES.items() if v == _MODE][0]


def _need_gpu(t: torch.Tensor) -> None:
    if not t.is_cuda:
        raise RuntimeError("vts kernels run on a ROCm device only (no CPU fallback): got a tensor on "
                           + str(t.device))


def _c(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    return None if t is None else t.contiguous()


# ------------------------------------------------------------------------------------- LSTM
class RecurrenceTimeout(RuntimeError):
    """A cooperative recurrent kernel gave up waiting for a peer workgroup; its outputs are NaN."""


class _RecurrenceStatus:
    """Status words of the recurrent launches of the current step.

    The cluster forward kernel, the pair backward kernel and the fused forward kernel (bf16x3, hidden 256) are grids
    of workgroups that wait for each other with bounded polls.  A workgroup that never hears from its peer writes NaN
    and sets its launch's status word (vts_lstm_recurrent_{fwd,bwd}_status, vts_lstm_fused_fwd_status).  torch's own
    LSTM would raise on the host; here every launch that can time out copies its word into a small device array
    (stream-ordered, four bytes).  At the end of a step ``finish_step`` hands that array to the optimizer as the
    GUARD of its update kernel (vts_adam_step_guarded: any non-zero word and the parameters stay as they are - under
    data parallelism after a MAX all-reduce of the words, so that every rank decides alike) and starts an
    asynchronous copy of it to the host; the copy is looked at when it has arrived - at the start of a later step, or
    in ``raise_on_timeout`` - and a non-zero word raises RecurrenceTimeout naming the kernel.  No host
    synchronisation is made on the way, and NaN gradients never reach the parameters."""

    SLOTS = 64

    def __init__(self):
        self.enabled = True
        self.words = {}     # device index -> int32[SLOTS]
        self.names = {}     # device index -> names of the launches noted since the last read
        self.pending = []   # (event, pinned host copy of a finished step's words, names, reduced over ranks?)

    def note(self, name: str, query, ws, ws_bytes: int, B: int, T: int, H: int) -> None:
        if not self.enabled or H != 256 or _MODE != 1 or torch.cuda.is_current_stream_capturing():
            return          # (other shapes and modes run one-workgroup kernels, which cannot time out)
        dev = ws.device.index
        words = self.words.get(dev)
        if words is None:
            words = self.words[dev] = torch.zeros(self.SLOTS, dtype=torch.int32, device=ws.device)
        names = self.names.setdefault(dev, [])
        if len(names) >= self.SLOTS:        # nobody has read for a long while: keep the oldest, drop the rest
            return
        slot = words[len(names):len(names) + 1]
        L.check(query(L.wptr(ws), ws_bytes, B, T, H, C.c_void_p(slot.data_ptr()), L.stream()), "vts_lstm_recurrent_status")
        names.append(name)

    def begin_step(self) -> None:
        """Look at the status words of earlier steps that have reached the host by now (raises RecurrenceTimeout if
        one is set) and forget launches that nobody asked about (a step that ended in an exception, forward-only
        calls)."""
        self.poll(block=False)
        for dev in self.names:
            self.names[dev] = []

    def finish_step(self, device=None, reduce=None):
        """The status words of this step's launches on ``device`` as the guard of the optimizer update: an int32
        device tensor (one word per launch), or None if no launch of the step can time out.  ``reduce`` (data
        parallelism) is applied to the tensor in place first, e.g. a MAX all-reduce.  Also queues the words'
        asynchronous copy to the host for ``poll``."""
        if not self.enabled or not any(self.names.values()) or torch.cuda.is_current_stream_capturing():
            return None
        for dev, names in self.names.items():
            if not names or (device is not None and torch.device(device).index not in (None, dev)):
                continue
            words = self.words[dev][:len(names)]
            if reduce is not None:
                reduce(words)
            host = torch.empty(len(names), dtype=torch.int32, pin_memory=True)
            host.copy_(words, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self.pending.append((ev, host, list(names), reduce is not None))
            self.names[dev] = []
            return words
        return None

    def poll(self, block: bool) -> None:
        """Raise RecurrenceTimeout if a finished step's words (those that have arrived; all of them if ``block``)
        hold a non-zero one."""
        keep, bad, reduced = [], [], False
        for ev, host, names, red in self.pending:
            if block:
                ev.synchronize()
            if not ev.query():
                keep.append((ev, host, names, red))
                continue
            hit = [(n, v) for n, v in zip(names, host.tolist()) if v != 0]
            bad += hit
            reduced = reduced or (red and bool(hit))
        self.pending = keep
        if bad:
            raise RecurrenceTimeout(self._message(bad, reduced))

    @staticmethod
    def _message(bad, reduced=False) -> str:
        return ("a workgroup of " + ", ".join(f"{n} (status {v:#x})" for n, v in bad) + " gave up waiting for its peer "
                "(another kernel held compute units for longer than the poll bound, about a second)"
                + (" on this rank or another one" if reduced else "") + ": the launch's outputs are NaN and the "
                "optimizer update of that step was not made.  VTS_LSTM_CLUSTER=0 / VTS_LSTM_PAIR_BWD=0 select the "
                "one-workgroup kernels, which wait for nobody.")

    def read(self, device=None):
        """[(launch name, status word)] of the launches noted since the last read (synchronises with the device)."""
        out = []
        for dev, names in self.names.items():
            if names and (device is None or torch.device(device).index in (None, dev)):
                vals = self.words[dev][:len(names)].cpu().tolist()
                out += list(zip(names, vals))
                self.names[dev] = []
        return out

    def raise_on_timeout(self) -> None:
        """The synchronous check: waits for every finished step's words and for the current step's launches."""
        if torch.cuda.is_current_stream_capturing():
            return
        self.poll(block=True)
        if not self.enabled or not any(self.names.values()):
            return
        bad = [(n, v) for n, v in self.read() if v != 0]
        if bad:
            raise RecurrenceTimeout(self._message(bad))


recurrence_status = _RecurrenceStatus()


def set_status_checks(on: bool) -> None:
    """Turn the per-step read-back of the cooperative recurrences' status words on (default) or off."""
    recurrence_status.enabled = bool(on)


class _LayerCfg:
    """Static facts of one layer of a stack, shared by its forward and backward."""

    def __init__(self, B, T, cin, H, first, last, planes, need_seq, mode):
        self.B, self.T, self.cin, self.H = B, T, cin, H
        self.first, self.last, self.planes, self.need_seq, self.mode = first, last, planes, need_seq, mode
        self.Bp = (B + 15) // 16 * 16


class _LSTMLayer(torch.autograd.Function):
    """One layer of nn.LSTM(batch_first=True), zero initial state: (activation, h_last [B,H], c_last [B,H]).

    One autograd node per layer, so a layer's weight gradients are final (and a data-parallel trainer's hooks can
    send them) while the layers below are still in their backward pass.

    What travels between the layers of a stack is private to them.  Forward: the h sequence, either fp32 rows
    [Bp, T, H] (Bp = B rounded up to 16) or, on the plane path (vts_lstm_planes_supported), the bf16 head/tail
    planes of csrc/gemm_planes.h viewed as a float tensor.  Backward: the gradient w.r.t. that sequence in the
    kernels' tile16 layout, in a tensor of the activation's shape (autograd only checks the shape).  The last
    layer returns ordinary rows [B, T, H] (an empty tensor if the caller does not want the sequence) and takes an
    ordinary gradient; the first layer takes ordinary rows."""

    @staticmethod
    def forward(ctx, inp, w_ih, w_hh, b_ih, b_hh, cfg: _LayerCfg):
        lib = L.load()
        B, T, cin, H, mode = cfg.B, cfg.T, cfg.cin, cfg.H, cfg.mode
        dev = inp.device
        w_ih, w_hh, b_ih, b_hh = w_ih.contiguous(), w_hh.contiguous(), b_ih.contiguous(), b_hh.contiguous()
        gates = L.workspace(lib.vts_lstm_gates_bytes(B, T, H), dev)
        cells = L.workspace(lib.vts_lstm_cells_bytes(B, T, H), dev)
        h_last = torch.empty(B, H, device=dev, dtype=torch.float32)
        c_last = torch.empty(B, H, device=dev, dtype=torch.float32)
        ws_bytes = lib.vts_lstm_recurrent_workspace_bytes(B, T, H)
        ws = L.workspace(ws_bytes, dev)
        st = L.stream()
        want_rows = (cfg.last and cfg.need_seq) or not cfg.planes
        hseq = None
        if want_rows:
            hseq = torch.empty(B if cfg.last else cfg.Bp, T, H, device=dev, dtype=torch.float32)
        if cfg.planes:
            if cfg.first:
                xp = L.workspace(lib.vts_lstm_planes_bytes(B, T, cin, 0), dev)
                L.check(L.timer.call("lstm_x_to_planes", lib.vts_lstm_planes_from_rows, L.ptr(inp), L.wptr(xp), B, T, cin,
                                     H, mode, st), "vts_lstm_planes_from_rows")
            else:
                xp = inp
            hp = torch.empty(lib.vts_lstm_planes_bytes(B, T, H, 1) // 4, device=dev, dtype=torch.float32)
            fused = False
            if lib.vts_lstm_fused_fwd_preferred() and lib.vts_lstm_fused_fwd_supported(B, T, cin, H, mode):
                # the whole forward pass of the layer in one kernel; refused (code 3: the runtime will not promise a
                # resident grid, or the stream is recording a graph) -> the two kernels below
                fws_bytes = lib.vts_lstm_fused_fwd_workspace_bytes(B, T, cin, H)
                fws = L.workspace(fws_bytes, dev)
                rc = L.timer.call("lstm_fused_fwd", lib.vts_lstm_fused_fwd_planes, L.wptr(xp), 0 if cfg.first else 1,
                                  L.ptr(w_ih), L.ptr(b_ih), L.ptr(b_hh), L.ptr(w_hh), L.wptr(gates), L.wptr(cells),
                                  L.ptr(hseq), L.wptr(hp), L.ptr(h_last), L.ptr(c_last), L.wptr(fws), fws_bytes, B, T,
                                  cin, H, mode, st)
                if rc != 3:
                    L.check(rc, "vts_lstm_fused_fwd_planes")
                    fused = True
                    recurrence_status.note("the fused forward pass (lstm_fused_fwd_kernel)",
                                           lambda w, wb, b_, t_, h_, out, s_: lib.vts_lstm_fused_fwd_status(w, wb, b_, t_, cin, h_, out, s_),
                                           fws, fws_bytes, B, T, H)
            if not fused:
                pws_bytes = lib.vts_lstm_planes_workspace_bytes(B, T, cin, H)
                pws = L.workspace(pws_bytes, dev)
                L.check(L.timer.call("lstm_inproj_fwd", lib.vts_lstm_inproj_fwd_planes, L.wptr(xp), 0 if cfg.first else 1,
                                     L.ptr(w_ih), L.ptr(b_ih), L.ptr(b_hh), L.wptr(gates), L.wptr(pws), pws_bytes, B, T, cin,
                                     H, mode, st), "vts_lstm_inproj_fwd_planes")
                L.check(L.timer.call("lstm_recurrent_fwd", lib.vts_lstm_recurrent_fwd_planes, L.wptr(gates), L.wptr(cells),
                                     L.ptr(hseq), L.wptr(hp), L.ptr(h_last), L.ptr(c_last), L.ptr(w_hh), L.wptr(ws), ws_bytes,
                                     B, T, H, mode, st), "vts_lstm_recurrent_fwd_planes")
                recurrence_status.note("the forward recurrence (lstm_rec3c_fwd_kernel)", lib.vts_lstm_recurrent_fwd_status, ws,
                                       ws_bytes, B, T, H)
            ctx.save_for_backward(xp, gates, cells, hp, w_ih, w_hh)
        else:
            x_rows = inp if cfg.first else inp   # rows either way; only the first B windows of a padded tensor are read
            L.check(L.timer.call("lstm_inproj_fwd", lib.vts_lstm_inproj_fwd, L.ptr(x_rows), L.ptr(w_ih), L.ptr(b_ih),
                                 L.ptr(b_hh), L.wptr(gates), B, T, cin, H, mode, st), "vts_lstm_inproj_fwd")
            L.check(L.timer.call("lstm_recurrent_fwd", lib.vts_lstm_recurrent_fwd, L.wptr(gates), L.wptr(cells),
                                 L.ptr(hseq), L.ptr(h_last), L.ptr(c_last), L.ptr(w_hh), L.wptr(ws), ws_bytes,
                                 B, T, H, mode, st), "vts_lstm_recurrent_fwd")
            recurrence_status.note("the forward recurrence (lstm_rec3c_fwd_kernel)", lib.vts_lstm_recurrent_fwd_status, ws,
                                   ws_bytes, B, T, H)
            hp = None
            ctx.save_for_backward(x_rows, gates, cells, hseq, w_ih, w_hh)
        ctx.cfg = cfg
        ctx.ran = False
        ctx.set_materialize_grads(False)
        if cfg.last:
            act = hseq if hseq is not None else torch.empty(0, device=dev, dtype=torch.float32)
        else:
            act = hp if cfg.planes else hseq
        return act, h_last, c_last

    @staticmethod
    def backward(ctx, g_act, g_h, g_c):
        lib = L.load()
        cfg = ctx.cfg
        B, T, cin, H, mode = cfg.B, cfg.T, cfg.cin, cfg.H, cfg.mode
        xin, gates, cells, hs, w_ih, w_hh = ctx.saved_tensors
        dev = gates.device
        if ctx.ran and not cfg.planes:
            raise RuntimeError("vts LSTM: a second backward pass over the same forward is not supported (the first "
                               "one turned the saved gate activations into gate gradients in place)")
        ctx.ran = True
        st = L.stream()
        dh_seq = None
        if g_act is not None and g_act.numel() > 0:
            if cfg.last:      # ordinary rows from outside the stack
                dh_seq = L.workspace(lib.vts_lstm_seqgrad_bytes(B, T, H), dev)
                L.check(lib.vts_tile16_convert(L.ptr(g_act.contiguous()), L.wptr(dh_seq), B, T, H, 0, st),
                        "vts_tile16_convert")
            else:             # tile16 from the layer above
                dh_seq = g_act
        need_dx = (not cfg.first) or ctx.needs_input_grad[0]
        if need_dx and cin % 16 != 0:
            raise NotImplementedError("gradient w.r.t. the LSTM input needs input_size % 16 == 0")
        dx = None
        if need_dx:
            # shaped like the activation it is the gradient of (planes carry one more step slot than tile16 needs)
            n = xin.numel() if not cfg.first else lib.vts_lstm_seqgrad_bytes(B, T, cin) // 4
            dx = torch.empty(n, device=dev, dtype=torch.float32)
        dw_ih, dw_hh = torch.empty_like(w_ih), torch.empty_like(w_hh)
        db_ih = torch.empty(4 * H, device=dev, dtype=torch.float32)
        db_hh = torch.empty(4 * H, device=dev, dtype=torch.float32)
        dh_last = _c(g_h)
        dc_last = _c(g_c)
        ws_bytes = lib.vts_lstm_recurrent_workspace_bytes(B, T, H)
        ws = L.workspace(ws_bytes, dev)
        if cfg.planes:
            dgp = L.workspace(lib.vts_lstm_planes_bytes(B, T, 4 * H, 0), dev)
            L.check(L.timer.call("lstm_recurrent_bwd", lib.vts_lstm_recurrent_bwd_planes, L.wptr(gates), L.wptr(cells),
                                 None if dh_seq is None else L.wptr(dh_seq), L.ptr(dh_last), L.ptr(dc_last), L.ptr(w_hh),
                                 L.wptr(dgp), L.ptr(db_ih), L.ptr(db_hh), 0, L.wptr(ws), ws_bytes, B, T, H, mode, st),
                    "vts_lstm_recurrent_bwd_planes")
            recurrence_status.note("the backward recurrence (lstm_rec3p_bwd_kernel)", lib.vts_lstm_recurrent_bwd_status, ws,
                                   ws_bytes, B, T, H)
            pws_bytes = lib.vts_lstm_planes_workspace_bytes(B, T, cin, H)
            pws = L.workspace(pws_bytes, dev)
            if dx is not None:
                L.check(L.timer.call("lstm_dgrad", lib.vts_lstm_dgrad_planes, L.wptr(dgp), L.ptr(w_ih), L.wptr(dx),
                                     L.wptr(pws), pws_bytes, B, T, cin, H, mode, st), "vts_lstm_dgrad_planes")
            L.check(L.timer.call("lstm_wgrad_pair", lib.vts_lstm_wgrad_planes, L.wptr(dgp), L.wptr(xin),
                                 0 if cfg.first else 1, L.wptr(hs), L.ptr(dw_ih), L.ptr(dw_hh), 0, L.wptr(pws), pws_bytes,
                                 B, T, cin, H, mode, st), "vts_lstm_wgrad_planes")
        else:
            L.check(L.timer.call("lstm_recurrent_bwd", lib.vts_lstm_recurrent_bwd, L.wptr(gates), L.wptr(cells),
                                 None if dh_seq is None else L.wptr(dh_seq), L.ptr(dh_last), L.ptr(dc_last),
                                 L.ptr(w_hh), L.ptr(db_ih), L.ptr(db_hh), 0, L.wptr(ws), ws_bytes, B, T, H, mode,
                                 st), "vts_lstm_recurrent_bwd")
            if dx is not None:
                L.check(L.timer.call("lstm_dgrad", lib.vts_lstm_dgrad, L.wptr(gates), L.ptr(w_ih), L.wptr(dx),
                                     B, T, cin, H, mode, st), "vts_lstm_dgrad")
            wg_bytes = max(lib.vts_lstm_wgrad_workspace_bytes(B, T, cin, H),
                           lib.vts_lstm_wgrad_workspace_bytes(B, T, H, H))
            wg = L.workspace(wg_bytes, dev)
            if lib.vts_lstm_wgrad_pair_supported(B, T, cin, H, mode):
                # both weight gradients in one launch (the first 256 input columns ride with the recurrent product)
                L.check(L.timer.call("lstm_wgrad_pair", lib.vts_lstm_wgrad_pair, L.wptr(gates), L.ptr(xin), L.ptr(hs),
                                     L.ptr(dw_ih), L.ptr(dw_hh), 0, L.wptr(wg), wg_bytes, B, T, cin, H, mode, st),
                        "vts_lstm_wgrad_pair")
            else:
                L.check(L.timer.call("lstm_wgrad_ih", lib.vts_lstm_wgrad, L.wptr(gates), L.ptr(xin), 0, L.ptr(dw_ih), 0,
                                     L.wptr(wg), wg_bytes, B, T, cin, H, mode, st), "vts_lstm_wgrad")
                L.check(L.timer.call("lstm_wgrad_hh", lib.vts_lstm_wgrad, L.wptr(gates), L.ptr(hs), -1, L.ptr(dw_hh),
                                     0, L.wptr(wg), wg_bytes, B, T, H, H, mode, st), "vts_lstm_wgrad")
        gin = None
        if need_dx:
            if cfg.first:
                gin = torch.empty(B, T, cin, device=dev, dtype=torch.float32)
                L.check(lib.vts_tile16_convert(L.ptr(gin), L.wptr(dx), B, T, cin, 1, st), "vts_tile16_convert")
            else:
                gin = dx.view(xin.shape)
        return gin, dw_ih, dw_hh, db_ih, db_hh, None


def lstm(x: torch.Tensor, weights: Sequence[torch.Tensor], num_layers: int, need_output: bool = True,
         top_state_only: bool = False) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """(output, h_n, c_n) of a stacked batch-first LSTM; ``weights`` = [w_ih, w_hh, b_ih, b_hh] per layer.
    need_output=False (callers that read only h_n, as the reference's encoders do: gaussian-lstm-DE.py:28) lets
    the plane path skip the fp32 copy of the top layer's sequence; ``output`` is then an empty tensor.
    top_state_only=True (the same callers: they read h_n[-1]) returns h_n and c_n as [1, B, H] views of the top
    layer's final state instead of stacking every layer's: no copy kernels forward, and backward the lower
    layers get no state gradient at all rather than a block of zeros to convert."""
    _need_gpu(x)
    lib = L.load()
    x = x.contiguous()
    B, T, C0 = x.shape
    H = weights[1].shape[1]
    planes = bool(lib.vts_lstm_planes_supported(B, T, C0, H, _MODE)) and not (x.requires_grad and C0 % 256 != 0)
    act = x
    hs, cs = [], []
    for k in range(num_layers):
        cfg = _LayerCfg(B, T, C0 if k == 0 else H, H, k == 0, k == num_layers - 1, planes, need_output, _MODE)
        act, h_last, c_last = _LSTMLayer.apply(act, *weights[4 * k:4 * k + 4], cfg)
        hs.append(h_last)
        cs.append(c_last)
    if top_state_only:
        return act, hs[-1].unsqueeze(0), cs[-1].unsqueeze(0)
    return act, torch.stack(hs), torch.stack(cs)


# ----------------------------------------------------------------------------------- Linear
def _linear_mode(M: int, N: int, K: int) -> int:
    """dtype_mode of one Linear layer.  Layers below 2 GFLOP per product (at 4096 windows: every head and decoder
    layer of the LSTM models) always take the fp32 matrix path: they are launch-latency sized, so bf16x3 buys
    nothing there, while its forward values sit 10x further from the reference's (~5e-6 against ~5e-7), which
    makes it that much likelier that a ReLU unit of the trunk or decoder lands on the other side of zero - a
    step of 1e-4 to 1e-3 in the gradients upstream of it at the bench shape (tests/test_gpu_y_full_size.py)."""
    return _MODE if 2.0 * M * N * K >= 2e9 else 0


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, act: int, keep, drop_p: float = DROP_P, seed: int = 0):
        _need_gpu(x)
        lib = L.load()
        if x.dim() != 2 or x.stride(1) != 1:
            x = x.reshape(-1, x.shape[-1]).contiguous()
        M, K = x.shape
        N = weight.shape[0]
        weight = weight.contiguous()
        y = torch.empty(M, N, device=x.device, dtype=torch.float32)
        aux = torch.empty(M, N, device=x.device, dtype=torch.float32) if act == L.ACT_CLAMP_MIN else None
        if act == L.ACT_RELU_DROPOUT and keep is None and not seed:
            act = L.ACT_RELU  # eval mode
        keep = _c(keep)
        scale = 1.0 / (1.0 - drop_p)
        if act == L.ACT_RELU_DROPOUT and keep is None:
            # mask from the library's counter hash: nothing but y is needed for backward
            L.check(L.timer.call(f"linear_fwd {M}x{N}x{K}", lib.vts_linear_relu_dropout_fwd, L.ptr(x), x.stride(0),
                                 L.ptr(weight), L.ptr(_c(bias)), L.ptr(y), M, N, K, drop_p, seed, _linear_mode(M, N, K),
                                 L.stream()),
                    "vts_linear_relu_dropout_fwd")
        else:
            L.check(L.timer.call(f"linear_fwd {M}x{N}x{K}", lib.vts_linear_fwd, L.ptr(x), x.stride(0), L.ptr(weight),
                                 L.ptr(_c(bias)), L.ptr(y), M, N, K, act, L.ptr(keep), scale, L.ptr(aux),
                                 _linear_mode(M, N, K), L.stream()), "vts_linear_fwd")
        ctx.act = act
        ctx.scale = scale
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, weight, y, aux)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = L.load()
        x, weight, y, aux = ctx.saved_tensors
        M, K = x.shape
        N = weight.shape[0]
        dy = dy.contiguous()
        dx = torch.empty(M, K, device=x.device, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(weight) if ctx.needs_input_grad[1] else None
        db = torch.empty(N, device=x.device, dtype=torch.float32) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        ws_bytes = lib.vts_linear_bwd_workspace_bytes(M, N, K)
        ws = L.workspace(ws_bytes, x.device)
        L.check(L.timer.call(f"linear_bwd {M}x{N}x{K}" + ("" if dx is not None else " nodx"), lib.vts_linear_bwd, L.ptr(dy), L.ptr(y), L.ptr(aux), ctx.act, ctx.scale,
                             L.ptr(x), x.stride(0), L.ptr(weight), L.ptr(dx), K, 0, L.ptr(dw), L.ptr(db), 0,
                             L.wptr(ws), ws_bytes, M, N, K, _linear_mode(M, N, K), L.stream()), "vts_linear_bwd")
        return dx, dw, db, None, None, None, None


def linear(x, weight, bias=None, act: int = L.ACT_NONE, keep: Optional[torch.Tensor] = None,
           drop_p: float = DROP_P, seed: int = 0) -> torch.Tensor:
    """act(x @ weight.T + bias) with the activation (and dropout) fused into the GEMM epilogue.
    ACT_RELU_DROPOUT takes its mask from ``keep`` (float 0/1, [M, N]) or, when ``keep`` is None and
    ``seed`` is non-zero, from the library's counter hash; neither = eval mode."""
    return _Linear.apply(x, weight, bias, act, keep, drop_p, seed)


class _FeedForward(torch.autograd.Function):
    """linear2(dropout(relu(linear1(x)))) as one autograd node over vts_ffn_fwd / vts_ffn_bwd: the hidden
    gradient goes from linear2's data-gradient GEMM straight to linear1's gradients (activation derivative
    and db1 in that GEMM's epilogue), so the [tokens, dim_feedforward] gradient is written once."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, keep, drop_p: float, seed: int):
        _need_gpu(x)
        lib = L.load()
        if x.dim() != 2 or x.stride(1) != 1:
            x = x.reshape(-1, x.shape[-1]).contiguous()
        M, K = x.shape
        Fh = w1.shape[0]
        N = w2.shape[0]
        w1, w2, b1, b2 = w1.contiguous(), w2.contiguous(), b1.contiguous(), b2.contiguous()
        y = torch.empty(M, N, device=x.device, dtype=torch.float32)
        mask_bytes = lib.vts_ffn_mask_bytes(M, Fh)
        mask = L.workspace(mask_bytes, x.device) if mask_bytes else None
        keep = _c(keep)
        # Where the library can work the hidden activations out again in backward (bf16x3 mode, the fused kernels'
        # shapes) nothing but the packed relu/dropout mask is kept: no [tokens, dim_feedforward] tensor is written.
        recompute = bool(FFN_RECOMPUTE and mask is not None and K == N and
                         lib.vts_ffn_recompute_supported(L.ptr(x), x.stride(0), M, K, Fh, _MODE))
        h = None
        if not recompute or FFN_TAPS is not None:
            # rows padded to the kernels' 128-token tiles; only the first M are meaningful
            h = torch.empty(lib.vts_ffn_hidden_bytes(M, Fh) // 4, device=x.device, dtype=torch.float32)[:M * Fh].view(M, Fh)
        L.check(L.timer.call(f"ffn_fwd {M}x{Fh}x{K}", lib.vts_ffn_fwd, L.ptr(x), x.stride(0), L.ptr(w1), L.ptr(b1),
                             L.ptr(w2), L.ptr(b2), L.ptr(keep), drop_p, seed, L.ptr(h),
                             None if mask is None else L.wptr(mask), L.ptr(y), M, K, Fh, _MODE, L.stream()),
                "vts_ffn_fwd")
        if FFN_TAPS is not None:
            FFN_TAPS.append(h.detach())
        dropped = keep is not None or (bool(seed) and drop_p > 0)
        ctx.cfg = (M, K, Fh, 1.0 / (1.0 - drop_p) if dropped else 1.0, recompute)
        ctx.save_for_backward(x, w1, w2, b1 if recompute else h, mask)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = L.load()
        x, w1, w2, h, mask = ctx.saved_tensors
        M, K, Fh, scale, recompute = ctx.cfg
        dev = x.device
        dy = dy.contiguous()
        dx = torch.empty(M, K, device=dev, dtype=torch.float32) if (ctx.needs_input_grad[0] or recompute) else None
        dw1, dw2 = torch.empty_like(w1), torch.empty_like(w2)
        db1 = torch.empty(Fh, device=dev, dtype=torch.float32)
        db2 = torch.empty(K, device=dev, dtype=torch.float32)
        if recompute:
            ws_bytes = lib.vts_ffn_bwd_recompute_workspace_bytes(M, K, Fh)
            ws = L.workspace(ws_bytes, dev)
            L.check(L.timer.call(f"ffn_bwd {M}x{Fh}x{K}", lib.vts_ffn_bwd_recompute, L.ptr(dy), L.ptr(x), x.stride(0),
                                 L.ptr(w1), L.ptr(h), L.ptr(w2), L.wptr(mask), scale, L.ptr(dx), K, L.ptr(dw1), L.ptr(db1),
                                 L.ptr(dw2), L.ptr(db2), L.wptr(ws), ws_bytes, M, K, Fh, _MODE, L.stream()),
                    "vts_ffn_bwd_recompute")
            return dx, dw1, db1, dw2, db2, None, None, None
        ws_bytes = lib.vts_ffn_bwd_workspace_bytes(M, K, Fh)
        ws = L.workspace(ws_bytes, dev)
        L.check(L.timer.call(f"ffn_bwd {M}x{Fh}x{K}", lib.vts_ffn_bwd, L.ptr(dy), L.ptr(x), x.stride(0), L.ptr(w1),
                             L.ptr(w2), L.ptr(h), None if mask is None else L.wptr(mask), scale, L.ptr(dx), K,
                             L.ptr(dw1), L.ptr(db1), L.ptr(dw2), L.ptr(db2), L.wptr(ws), ws_bytes, M, K, Fh, _MODE,
                             L.stream()), "vts_ffn_bwd")
        return dx, dw1, db1, dw2, db2, None, None, None


def feed_forward(x, w1, b1, w2, b2, keep=None, drop_p: float = 0.0, seed: int = 0) -> torch.Tensor:
    """linear2(dropout(relu(linear1(x)))): the feed-forward block of an encoder layer.  ``keep`` is a
    float 0/1 mask [M, dim_feedforward]; else ``seed`` != 0 draws it with the counter hash; else eval."""
    return _FeedForward.apply(x, w1, b1, w2, b2, keep, drop_p, seed)


# ----------------------------------------------------------------- fused head / decoder chains
class _MlpChain(torch.autograd.Function):
    """Trunk layers + heads as one autograd node over vts_mlp_chain_fwd / vts_mlp_chain_bwd.
    Arguments after the fixed ones: w_0, b_0, w_1, b_1, ... (trunk layers, then heads)."""

    @staticmethod
    def forward(ctx, x, cfg, keeps, *params):
        _need_gpu(x)
        lib = L.load()
        trunk_acts, head_acts, scale, names = cfg
        nt, nh = len(trunk_acts), len(head_acts)
        if x.dim() != 2 or x.stride(1) != 1:
            x = x.reshape(-1, x.shape[-1]).contiguous()
        B, K = x.shape
        ws = [p.contiguous() for p in params[0::2]]
        bs = [None if p is None else p.contiguous() for p in params[1::2]]
        keeps = [None if k is None else _c(k) for k in keeps]
        d = L.MlpChain()
        d.n_trunk, d.n_heads, d.in_dim, d.head_dim = nt, nh, K, ws[nt].shape[0]
        for i in range(nt):
            d.trunk_dim[i] = ws[i].shape[0]
            # a dropout layer without a mask is a plain relu (eval mode)
            d.trunk_act[i] = L.ACT_RELU if (trunk_acts[i] == L.ACT_RELU_DROPOUT and keeps[i] is None) else trunk_acts[i]
        for i in range(nh):
            d.head_act[i] = head_acts[i]
        d.keep_scale = scale
        dev = x.device
        touts = [torch.empty(B, ws[i].shape[0], device=dev, dtype=torch.float32) for i in range(nt)]
        houts = [torch.empty(B, d.head_dim, device=dev, dtype=torch.float32) for _ in range(nh)]
        hauxs = [torch.empty(B, d.head_dim, device=dev, dtype=torch.float32) if head_acts[i] == L.ACT_CLAMP_MIN else None
                 for i in range(nh)]
        L.check(L.timer.call(f"mlp_chain_fwd {B}x{K}", lib.vts_mlp_chain_fwd, C.byref(d), L.ptr(x), x.stride(0), B,
                             L.ptr_array(ws), L.ptr_array(bs), L.ptr_array(keeps), L.ptr_array(touts),
                             L.ptr_array(houts), L.ptr_array(hauxs), L.stream()), "vts_mlp_chain_fwd")
        if CHAIN_TAPS is not None:
            for name, t in zip(names, touts):
                CHAIN_TAPS[name] = t.detach()
        ctx.desc = d
        ctx.counts = (nt, nh, [b is not None for b in bs])
        ctx.save_for_backward(x, *ws, *[k for k in keeps if k is not None], *touts, *houts,
                              *[a for a in hauxs if a is not None])
        ctx.keep_at = [k is not None for k in keeps]
        ctx.aux_at = [a is not None for a in hauxs]
        return tuple(houts)

    @staticmethod
    def backward(ctx, *d_heads):
        lib = L.load()
        nt, nh, has_bias = ctx.counts
        saved = list(ctx.saved_tensors)
        x = saved.pop(0)
        ws = [saved.pop(0) for _ in range(nt + nh)]
        keeps = [saved.pop(0) if flag else None for flag in ctx.keep_at]
        touts = [saved.pop(0) for _ in range(nt)]
        houts = [saved.pop(0) for _ in range(nh)]
        hauxs = [saved.pop(0) if flag else None for flag in ctx.aux_at]
        B, K = x.shape
        dev = x.device
        d = ctx.desc
        d_heads = [None if g is None else g.contiguous() for g in d_heads]
        dx = torch.empty(B, K, device=dev, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        dws = [torch.empty_like(w) for w in ws]
        dbs = [torch.empty(w.shape[0], device=dev, dtype=torch.float32) if hb else None for w, hb in zip(ws, has_bias)]
        ws_bytes = lib.vts_mlp_chain_bwd_workspace_bytes(C.byref(d), B)
        wsp = L.workspace(ws_bytes, dev)
        L.check(L.timer.call(f"mlp_chain_bwd {B}x{K}", lib.vts_mlp_chain_bwd, C.byref(d), L.ptr(x), x.stride(0), B,
                             L.ptr_array(ws), L.ptr_array(keeps), L.ptr_array(touts), L.ptr_array(houts),
                             L.ptr_array(hauxs), L.ptr_array(d_heads), L.ptr(dx), K, L.ptr_array(dws), L.ptr_array(dbs),
                             L.wptr(wsp), ws_bytes, L.stream()), "vts_mlp_chain_bwd")
        grads = []
        for dw, db in zip(dws, dbs):
            grads += [dw, db]
        return (dx, None, None, *grads)


def mlp_chain_usable(x, trunk, heads) -> bool:
    """Whether vts_mlp_chain_* takes this chain (widths and alignment; see include/vts.h)."""
    if not _FUSED_CHAINS or x.dim() != 2 or x.stride(1) != 1 or x.stride(0) % 4 or x.data_ptr() % 16:
        return False
    if x.shape[0] > CHAIN_MAX_ROWS:
        return False
    dims = [x.shape[1]] + [w.shape[0] for w, _, _ in trunk]
    if any(v % 4 or v > 512 for v in dims) or len(trunk) > 3 or not 1 <= len(heads) <= 8:
        return False
    hd = heads[0][0].shape[0]
    if any(w.shape[0] != hd for w, _, _ in heads) or hd * len(heads) > 512 or (len(heads) > 1 and hd % 16):
        return False
    # the kernels read weight rows sixteen bytes at a time (a contiguous copy of a strided view is aligned again)
    if any(w.is_contiguous() and w.data_ptr() % 16 for w, _, _ in list(trunk) + list(heads)):
        return False
    return all(a in (L.ACT_NONE, L.ACT_RELU, L.ACT_RELU_DROPOUT) for _, _, a in trunk)


def mlp_chain(x, trunk, heads, keeps=None, drop_p: float = DROP_P, names=None):
    """Trunk layers [(weight, bias, act), ...] then heads [(weight, bias, act), ...] that all read the last trunk
    output; returns the tuple of head outputs.  ``keeps``: one float 0/1 mask (or None) per trunk layer, used by the
    ACT_RELU_DROPOUT ones (None there = eval mode).  ``names``: names of the trunk layers for CHAIN_TAPS."""
    keeps = list(keeps) if keeps is not None else [None] * len(trunk)
    params = []
    for w, b, _ in list(trunk) + list(heads):
        params += [w, b]
    cfg = ([a for _, _, a in trunk], [a for _, _, a in heads], 1.0 / (1.0 - drop_p), tuple(names or ()))
    return _MlpChain.apply(x, cfg, keeps, *params)


# -------------------------------------------------------------------------------- LayerNorm
class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps: float):
        _need_gpu(x)
        lib = L.load()
        if x.dim() != 2 or x.stride(1) != 1:
            x = x.reshape(-1, x.shape[-1]).contiguous()
        M, H = x.shape
        y = torch.empty(M, H, device=x.device, dtype=torch.float32)
        mean = torch.empty(M, device=x.device, dtype=torch.float32)
        rstd = torch.empty(M, device=x.device, dtype=torch.float32)
        gamma, beta = gamma.contiguous(), beta.contiguous()
        L.check(L.timer.call("layernorm_fwd", lib.vts_layernorm_fwd, L.ptr(x), x.stride(0), L.ptr(gamma), L.ptr(beta),
                             L.ptr(y), L.ptr(mean), L.ptr(rstd), M, H, eps, L.stream()), "vts_layernorm_fwd")
        ctx.save_for_backward(x, gamma, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = L.load()
        x, gamma, mean, rstd = ctx.saved_tensors
        M, H = x.shape
        dy = dy.contiguous()
        dx = torch.empty(M, H, device=x.device, dtype=torch.float32)
        dgamma = torch.empty(H, device=x.device, dtype=torch.float32)
        dbeta = torch.empty(H, device=x.device, dtype=torch.float32)
        ws_bytes = lib.vts_layernorm_bwd_workspace_bytes(M, H)
        ws = L.workspace(ws_bytes, x.device)
        L.check(L.timer.call("layernorm_bwd", lib.vts_layernorm_bwd, L.ptr(dy), L.ptr(x), x.stride(0), L.ptr(gamma),
                             L.ptr(mean), L.ptr(rstd), L.ptr(dx), H, L.ptr(dgamma), L.ptr(dbeta), 0, L.wptr(ws),
                             ws_bytes, M, H, L.stream()), "vts_layernorm_bwd")
        return dx, dgamma, dbeta, None


def layer_norm(x, gamma, beta, eps: float = 1e-5) -> torch.Tensor:
    return _LayerNorm.apply(x, gamma, beta, eps)


# ---------------------------------------------------------------------------------- reparam
class _Reparam(torch.autograd.Function):
    """family 0: (mean, logvar, eps); 1: (scale, shape, u); 2: all + choice + pi_params."""

    @staticmethod
    def forward(ctx, family: int, mean, logvar, scale, shape, eps, u, choice, pi):
        lib = L.load()
        ref = mean if mean is not None else scale
        _need_gpu(ref)
        B, Ld = ref.shape
        mean, logvar, scale, shape, eps, u, choice = [_c(t) for t in (mean, logvar, scale, shape, eps, u, choice)]
        pi_d = None if pi is None else pi.detach().contiguous()
        z = torch.empty(B, Ld, device=ref.device, dtype=torch.float32)
        L.check(L.timer.call("reparam_fwd", lib.vts_reparam_fwd, family, L.ptr(mean), L.ptr(logvar), L.ptr(scale), L.ptr(shape), L.ptr(eps),
                                    L.ptr(u), L.ptr(choice), L.ptr(pi_d), L.ptr(z), B, Ld, L.stream()),
                "vts_reparam_fwd")
        ctx.family = family
        ctx.save_for_backward(logvar, scale, shape, eps, u, choice, pi_d)
        return z

    @staticmethod
    def backward(ctx, dz):
        lib = L.load()
        logvar, scale, shape, eps, u, choice, pi_d = ctx.saved_tensors
        dz = dz.contiguous()
        B, Ld = dz.shape
        fam = ctx.family
        new = lambda: torch.empty(B, Ld, device=dz.device, dtype=torch.float32)
        dmean = dlogvar = dscale = dshape = None
        if fam != L.GPD:
            dmean, dlogvar = new(), new()
        if fam != L.GAUSSIAN:
            dscale, dshape = new(), new()
        L.check(L.timer.call("reparam_bwd", lib.vts_reparam_bwd, fam, L.ptr(dz), L.ptr(logvar), L.ptr(scale), L.ptr(shape), L.ptr(eps), L.ptr(u),
                                    L.ptr(choice), L.ptr(pi_d), L.ptr(dmean), L.ptr(dlogvar), L.ptr(dscale),
                                    L.ptr(dshape), B, Ld, L.stream()), "vts_reparam_bwd")
        return None, dmean, dlogvar, dscale, dshape, None, None, None, None


class _ReparamKL(torch.autograd.Function):
    """_Reparam fused with the KL sums of the loss: returns (z, klpart).  klpart[nblk, 2] holds
    per-block partial sums of the Gaussian and GPD KL integrands; the loss built from it sends
    d loss / d klpart back, and backward adds the KL gradient to the sampling gradient in the one
    pass that writes dmean / dlogvar / dscale / dshape."""

    @staticmethod
    def forward(ctx, family: int, threshold: float, mean, logvar, scale, shape, eps, u, choice, pi):
        lib = L.load()
        ref = mean if mean is not None else scale
        _need_gpu(ref)
        B, Ld = ref.shape
        mean, logvar, scale, shape, eps, u, choice = [_c(t) for t in (mean, logvar, scale, shape, eps, u, choice)]
        pi_d = None if pi is None else pi.detach().contiguous()
        z = torch.empty(B, Ld, device=ref.device, dtype=torch.float32)
        klpart = torch.empty(lib.vts_reparam_kl_blocks(B, Ld), 2, device=ref.device, dtype=torch.float32)
        L.check(L.timer.call("reparam_kl_fwd", lib.vts_reparam_kl_fwd, family, L.ptr(mean), L.ptr(logvar),
                             L.ptr(scale), L.ptr(shape), L.ptr(eps), L.ptr(u), L.ptr(choice), L.ptr(pi_d),
                             threshold, L.ptr(z), L.ptr(klpart), B, Ld, L.stream()), "vts_reparam_kl_fwd")
        ctx.family, ctx.threshold, ctx.dims = family, threshold, (B, Ld)
        ctx.save_for_backward(mean, logvar, scale, shape, eps, u, choice, pi_d)
        return z, klpart

    @staticmethod
    def backward(ctx, dz, dklpart):
        lib = L.load()
        mean, logvar, scale, shape, eps, u, choice, pi_d = ctx.saved_tensors
        B, Ld = ctx.dims
        fam = ctx.family
        dev = (mean if mean is not None else scale).device
        dz, dklpart = _c(dz), _c(dklpart)
        new = lambda: torch.empty(B, Ld, device=dev, dtype=torch.float32)
        dmean = dlogvar = dscale = dshape = None
        if fam != L.GPD:
            dmean, dlogvar = new(), new()
        if fam != L.GAUSSIAN:
            dscale, dshape = new(), new()
        L.check(L.timer.call("reparam_kl_bwd", lib.vts_reparam_kl_bwd, fam, L.ptr(dz), L.ptr(dklpart), L.ptr(mean),
                             L.ptr(logvar), L.ptr(scale), L.ptr(shape), L.ptr(eps), L.ptr(u), L.ptr(choice),
                             L.ptr(pi_d), ctx.threshold, L.ptr(dmean), L.ptr(dlogvar), L.ptr(dscale), L.ptr(dshape),
                             B, Ld, L.stream()), "vts_reparam_kl_bwd")
        return None, None, dmean, dlogvar, dscale, dshape, None, None, None, None


class KLSums:
    """KL partial sums produced while sampling, with the identity of the tensors they were
    computed from: the ``*_elbo`` functions use them only for exactly those tensors and that
    threshold, and otherwise read the latent parameters again."""

    def __init__(self, klpart: torch.Tensor, latents, threshold: float):
        self.klpart, self.latents, self.threshold = klpart, tuple(latents), float(threshold)

    def matches(self, latents, threshold: float) -> bool:
        return (len(latents) == len(self.latents) and all(a is b for a, b in zip(latents, self.latents))
                and float(threshold) == self.threshold)


def reparameterize_gaussian(mean, logvar, eps: Optional[torch.Tensor] = None, with_kl: bool = False):
    """mean + eps * exp(0.5 * logvar); eps ~ N(0,1) drawn on the device when not fed in.
    ``with_kl``: also return the KLSums gaussian_elbo can take in place of a second read."""
    if eps is None:
        eps = torch.randn_like(mean)
    if with_kl:
        z, klpart = _ReparamKL.apply(L.GAUSSIAN, 0.0, mean, logvar, None, None, eps, None, None, None)
        return z, KLSums(klpart, (mean, logvar), 0.0)
    return _Reparam.apply(L.GAUSSIAN, mean, logvar, None, None, eps, None, None, None)


def reparameterize_gpd(scale, shape, size=None, u: Optional[torch.Tensor] = None,
                       kl_threshold: Optional[float] = None):
    """scale / shape * ((1 - u) ** (-shape) - 1); u ~ U[0,1) drawn on the device when not fed in
    (the reference draws it on the CPU, gpd-lstm-DE.py:11, which is why it cannot leave the CPU).
    ``kl_threshold``: also return the KLSums gpd_elbo can take for that threshold."""
    if u is None:
        u = torch.rand(scale.shape if size is None else size, device=scale.device)
    if kl_threshold is not None:
        z, klpart = _ReparamKL.apply(L.GPD, float(kl_threshold), None, None, scale, shape, None, u, None, None)
        return z, KLSums(klpart, (scale, shape), kl_threshold)
    return _Reparam.apply(L.GPD, None, None, scale, shape, None, u, None, None)


def reparameterize_mixture(mean, logvar, scale, shape, pi_params, choice=None, eps=None, u=None,
                           kl_threshold: Optional[float] = None):
    """Row i is the Gaussian sample iff choice[i] < softmax(pi_params)[0], else the GPD sample
    (mixture-vae-DE.py:94-107 without the per-row Python loop).  Draw order when noise is not fed
    in follows the reference: choice, eps, u.
    ``kl_threshold``: also return the KLSums mixture_elbo can take for that threshold."""
    if choice is None:
        choice = torch.rand(mean.shape[0], device=mean.device)
    if eps is None:
        eps = torch.randn_like(mean)
    if u is None:
        u = torch.rand(mean.shape, device=mean.device)
    if kl_threshold is not None:
        z, klpart = _ReparamKL.apply(L.MIXTURE, float(kl_threshold), mean, logvar, scale, shape, eps, u, choice,
                                     pi_params)
        return z, KLSums(klpart, (mean, logvar, scale, shape), kl_threshold)
    return _Reparam.apply(L.MIXTURE, mean, logvar, scale, shape, eps, u, choice, pi_params)


# ------------------------------------------------------------------------------------- ELBO
# Data-parallel exact loss: when set, the ELBO functions form the loss of the concatenated batch of all ranks.
# (reduce, ranks): reduce(sums) adds the 8-float statistics up over ranks in place (an all-reduce on the current
# stream); ranks = the number of ranks, whose gradients the caller averages afterwards.
_loss_reduction = None


def set_loss_reduction(reduce=None, scale: float = 1.0) -> None:
    """``reduce=None`` restores per-rank losses (the mean of which equals the global loss only when every
    rank holds the same number of rows of each masked class)."""
    global _loss_reduction
    _loss_reduction = None if reduce is None else (reduce, float(scale))


def _elbo_state_global(lib, family, recon, y, mean, logvar, scale, shape, klpart, pi, threshold, beta, B, fs, Ld):
    """state of the loss over all ranks' rows: local sums -> reduce -> finalize."""
    reduce, gscale = _loss_reduction
    dev = recon.device
    sums = torch.empty(8, device=dev, dtype=torch.float32)
    state = torch.empty(16, device=dev, dtype=torch.float32)
    ws_bytes = lib.vts_elbo_workspace_bytes(B, fs, Ld)
    ws = L.workspace(ws_bytes, dev)
    nkl = 0 if klpart is None else klpart.shape[0]
    L.check(L.timer.call("elbo_sums", lib.vts_elbo_sums, family, L.ptr(recon), L.ptr(y), L.ptr(mean), L.ptr(logvar),
                         L.ptr(scale), L.ptr(shape), L.ptr(klpart), nkl, threshold, L.ptr(sums), L.wptr(ws), ws_bytes,
                         B, fs, Ld, L.stream()), "vts_elbo_sums")
    reduce(sums)
    L.check(L.timer.call("elbo_from_sums", lib.vts_elbo_from_sums, family, L.ptr(sums), L.ptr(pi), beta, gscale,
                         L.ptr(state), fs, Ld, L.stream()), "vts_elbo_from_sums")
    return state


class _Elbo(torch.autograd.Function):
    @staticmethod
    def forward(ctx, family: int, recon, y, mean, logvar, scale, shape, pi, threshold: float, beta: float):
        _need_gpu(recon)
        lib = L.load()
        recon, y = recon.contiguous(), y.contiguous()
        B = recon.shape[0]
        fs = recon.numel() // B
        ref = mean if mean is not None else scale
        Ld = ref.shape[1]
        mean, logvar, scale, shape, pi = [_c(t) for t in (mean, logvar, scale, shape, pi)]
        if _loss_reduction is not None:
            state = _elbo_state_global(lib, family, recon, y, mean, logvar, scale, shape, None, pi, threshold, beta, B, fs, Ld)
        else:
            state = torch.empty(16, device=recon.device, dtype=torch.float32)
            ws_bytes = lib.vts_elbo_workspace_bytes(B, fs, Ld)
            ws = L.workspace(ws_bytes, recon.device)
            L.check(L.timer.call("elbo_fwd", lib.vts_elbo_fwd, family, L.ptr(recon), L.ptr(y), L.ptr(mean), L.ptr(logvar), L.ptr(scale),
                                 L.ptr(shape), L.ptr(pi), threshold, beta, L.ptr(state), L.wptr(ws), ws_bytes, B, fs,
                                 Ld, L.stream()), "vts_elbo_fwd")
        ctx.family, ctx.threshold, ctx.dims = family, threshold, (B, fs, Ld)
        ctx.save_for_backward(recon, y, mean, logvar, scale, shape, state)
        return state[0]

    @staticmethod
    def backward(ctx, gout):
        lib = L.load()
        recon, y, mean, logvar, scale, shape, state = ctx.saved_tensors
        B, fs, Ld = ctx.dims
        fam = ctx.family
        dev = recon.device
        gout = gout.contiguous().reshape(1)
        drecon = torch.empty_like(recon)
        new = lambda: torch.empty(B, Ld, device=dev, dtype=torch.float32)
        dmean = dlogvar = dscale = dshape = dpi = None
        if fam != L.GPD:
            dmean, dlogvar = new(), new()
        if fam != L.GAUSSIAN:
            dscale, dshape = new(), new()
        if fam >= L.MIXTURE:
            dpi = torch.empty(2, device=dev, dtype=torch.float32)
        L.check(L.timer.call("elbo_bwd", lib.vts_elbo_bwd, fam, L.ptr(recon), L.ptr(y), L.ptr(mean), L.ptr(logvar), L.ptr(scale), L.ptr(shape),
                                 L.ptr(state), L.ptr(gout), ctx.threshold, L.ptr(drecon), L.ptr(dmean),
                                 L.ptr(dlogvar), L.ptr(dscale), L.ptr(dshape), L.ptr(dpi), B, fs, Ld, L.stream()),
                "vts_elbo_bwd")
        return None, drecon, None, dmean, dlogvar, dscale, dshape, dpi, None, None


class _ElboRecon(torch.autograd.Function):
    """_Elbo with the KL sums taken from _ReparamKL's klpart instead of a second read of the
    latent parameters; backward returns drecon, d loss / d klpart and dpi."""

    @staticmethod
    def forward(ctx, family: int, recon, y, klpart, pi, threshold: float, beta: float, Ld: int):
        _need_gpu(recon)
        lib = L.load()
        recon, y, klpart, pi = recon.contiguous(), y.contiguous(), klpart.contiguous(), _c(pi)
        B = recon.shape[0]
        fs = recon.numel() // B
        if _loss_reduction is not None:
            state = _elbo_state_global(lib, family, recon, y, None, None, None, None, klpart, pi, threshold, beta, B, fs, Ld)
        else:
            state = torch.empty(16, device=recon.device, dtype=torch.float32)
            ws_bytes = lib.vts_elbo_workspace_bytes(B, fs, Ld)
            ws = L.workspace(ws_bytes, recon.device)
            L.check(L.timer.call("elbo_recon_fwd", lib.vts_elbo_recon_fwd, family, L.ptr(recon), L.ptr(y),
                                 L.ptr(klpart), klpart.shape[0], L.ptr(pi), threshold, beta, L.ptr(state), L.wptr(ws),
                                 ws_bytes, B, fs, Ld, L.stream()), "vts_elbo_recon_fwd")
        ctx.family, ctx.threshold, ctx.dims = family, threshold, (B, fs, klpart.shape[0])
        ctx.save_for_backward(recon, y, state)
        return state[0]

    @staticmethod
    def backward(ctx, gout):
        lib = L.load()
        recon, y, state = ctx.saved_tensors
        B, fs, nkl = ctx.dims
        dev = recon.device
        gout = gout.contiguous().reshape(1)
        drecon = torch.empty_like(recon)
        dklpart = torch.empty(nkl, 2, device=dev, dtype=torch.float32)
        dpi = torch.empty(2, device=dev, dtype=torch.float32) if ctx.family == L.MIXTURE else None
        L.check(L.timer.call("elbo_recon_bwd", lib.vts_elbo_recon_bwd, ctx.family, L.ptr(recon), L.ptr(y),
                             L.ptr(state), L.ptr(gout), ctx.threshold, L.ptr(drecon), L.ptr(dklpart), nkl,
                             L.ptr(dpi), B, fs, L.stream()), "vts_elbo_recon_bwd")
        return None, drecon, None, dklpart, dpi, None, None, None


def gaussian_elbo(pred, y, mean, logvar, beta: float = 0.001, kl: Optional[KLSums] = None) -> torch.Tensor:
    """mse(pred, y) + beta * mean_B(-0.5 * sum_L(1 + logvar - mean^2 - exp(logvar))).
    ``kl``: sums from reparameterize_gaussian(with_kl=True), used when they belong to (mean, logvar)."""
    if kl is not None and kl.matches((mean, logvar), 0.0):
        return _ElboRecon.apply(L.GAUSSIAN, pred, y, kl.klpart, None, 0.0, beta, mean.shape[1])
    return _Elbo.apply(L.GAUSSIAN, pred, y, mean, logvar, None, None, None, 0.0, beta)


def gpd_elbo(recon, y, scale, shape, threshold: float = 95.0, beta: float = 0.001,
             kl: Optional[KLSums] = None) -> torch.Tensor:
    """mse(recon, y) + beta * mean(log(scale) + log(1 + shape * threshold / scale))."""
    if kl is not None and kl.matches((scale, shape), threshold):
        return _ElboRecon.apply(L.GPD, recon, y, kl.klpart, None, float(threshold), beta, scale.shape[1])
    return _Elbo.apply(L.GPD, recon, y, None, None, scale, shape, None, float(threshold), beta)


def mixture_elbo(recon, y, mean, logvar, scale, shape, pi_params, threshold: float = 95.0,
                 beta: float = 0.001, kl: Optional[KLSums] = None) -> torch.Tensor:
    """mse + pi_g * (mse over |y| <= threshold + beta * KL_gaussian) + pi_p * (mse + beta * KL_gpd),
    (pi_g, pi_p) = softmax(pi_params).  NaN when no |y| <= threshold, as in the reference."""
    if kl is not None and kl.matches((mean, logvar, scale, shape), threshold):
        return _ElboRecon.apply(L.MIXTURE, recon, y, kl.klpart, pi_params, float(threshold), beta, mean.shape[1])
    return _Elbo.apply(L.MIXTURE, recon, y, mean, logvar, scale, shape, pi_params, float(threshold), beta)


def mixture_split_elbo(recon, y, mean, logvar, scale, shape, pi_params, threshold: float,
                       beta: float = 1.0) -> torch.Tensor:
    """Loss of the transformer variant (mixture-vae-transformer.py:100-113):
    mse + pi_g * (mse over |y| <= threshold + beta * KL_gaussian(mean, logvar))
        + pi_p * (mse over |y| > threshold + beta * KL_gaussian-form(shape, scale)).
    NaN when either selection is empty, as in the reference."""
    return _Elbo.apply(L.MIXTURE_SPLIT, recon, y, mean, logvar, scale, shape, pi_params, float(threshold), beta)


# ---------------------------------------------------------------- transformer encoder pieces
def new_seed() -> int:
    """A non-zero seed for one dropout site, drawn from torch's CPU generator (so torch.manual_seed
    makes a training run repeatable).  The draw is a host-side op: no device synchronisation."""
    return int(torch.randint(1, 2 ** 62, (1,)).item())


def _byte_mask(keep: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if keep is None:
        return None
    return keep.to(torch.uint8).contiguous()


class _MHA(torch.autograd.Function):
    """softmax(q k^T / sqrt(dh)) v over the heads of one encoder layer, dropout on the weights.
    Packed form  : q is None, kv = qkv [B, T, 3d] (the in-projection's output)  -> out [B, T, d].
    Last-token   : q [B, d] (one query per window), kv [B, T, 2d]               -> out [B, d]."""

    @staticmethod
    def forward(ctx, q, kv, nhead: int, keep, drop_p: float, seed: int):
        _need_gpu(kv)
        lib = L.load()
        kv = kv.contiguous()
        B, T, W = kv.shape
        packed = q is None
        d = W // 3 if packed else W // 2
        dh = d // nhead
        if packed:
            Tq, qbuf, ldq, koff = T, kv, W, d
        else:
            q = q.contiguous()
            Tq, qbuf, ldq, koff = 1, q, d, 0
        out = torch.empty(B * Tq, d, device=kv.device, dtype=torch.float32)
        lse = torch.empty(B * nhead * Tq, device=kv.device, dtype=torch.float32)
        keep = _byte_mask(keep)
        flat = kv.view(-1)
        L.check(L.timer.call("mha_fwd", lib.vts_mha_fwd, L.ptr(qbuf), ldq, L.ptr(flat[koff:]), L.ptr(flat[koff + d:]),
                             W, L.ptr(out), d, L.ptr(lse), L.bptr(keep), drop_p, seed, B, T, Tq, nhead, dh,
                             L.stream()), "vts_mha_fwd")
        ctx.cfg = (B, T, Tq, W, d, nhead, dh, packed, drop_p, seed)
        ctx.save_for_backward(q, kv, out, lse, keep)
        return out.view(B, T, d) if packed else out

    @staticmethod
    def backward(ctx, dout):
        lib = L.load()
        q, kv, out, lse, keep = ctx.saved_tensors
        B, T, Tq, W, d, nhead, dh, packed, drop_p, seed = ctx.cfg
        dout = dout.contiguous()
        dkv = torch.empty_like(kv)
        dflat, flat = dkv.view(-1), kv.view(-1)
        if packed:
            qbuf, ldq, koff, dq, dqbuf = kv, W, d, None, dkv
        else:
            dq = torch.empty_like(q)
            qbuf, ldq, koff, dqbuf = q, d, 0, dq
        L.check(L.timer.call("mha_bwd", lib.vts_mha_bwd, L.ptr(dout), d, L.ptr(qbuf), ldq, L.ptr(flat[koff:]),
                             L.ptr(flat[koff + d:]), W, L.ptr(out), d, L.ptr(lse), L.bptr(keep), drop_p, seed,
                             L.ptr(dqbuf), ldq, L.ptr(dflat[koff:]), L.ptr(dflat[koff + d:]), W, B, T, Tq, nhead, dh,
                             L.stream()), "vts_mha_bwd")
        return dq, dkv, None, None, None, None


def multi_head_attention(kv, nhead: int, q=None, keep=None, drop_p: float = 0.0, seed: int = 0):
    """Self-attention core of nn.MultiheadAttention between the in- and out-projections.
    ``kv``: packed projections [B, T, 3d] (q, k, v side by side), or [B, T, 2d] with ``q`` [B, d] for
    the one-query-per-window form.  ``keep``: 0/1 mask [B, nhead, Tq, T]; else ``seed`` != 0 draws the
    mask with the library's counter hash; else no dropout."""
    return _MHA.apply(q, kv, nhead, keep, drop_p, seed)


class _DropoutAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, residual, x, keep, drop_p: float, seed: int):
        _need_gpu(x)
        lib = L.load()
        x = x.contiguous()
        residual = residual.contiguous()
        keep = _byte_mask(keep)
        out = torch.empty_like(x)
        L.check(L.timer.call("dropout_add_fwd", lib.vts_dropout_add_fwd, L.ptr(residual), L.ptr(x), L.bptr(keep),
                             drop_p, seed, L.ptr(out), x.numel(), L.stream()), "vts_dropout_add_fwd")
        ctx.cfg = (drop_p, seed)
        ctx.save_for_backward(keep)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = L.load()
        (keep,) = ctx.saved_tensors
        drop_p, seed = ctx.cfg
        dout = dout.contiguous()
        if keep is None and not seed:
            return dout, dout, None, None, None
        dx = torch.empty_like(dout)
        L.check(L.timer.call("dropout_bwd", lib.vts_dropout_bwd, L.ptr(dout), L.bptr(keep), drop_p, seed, L.ptr(dx),
                             dout.numel(), L.stream()), "vts_dropout_bwd")
        return dout, dx, None, None, None


def dropout_add(residual, x, keep=None, drop_p: float = 0.0, seed: int = 0):
    """residual + dropout(x): the two residual sites of a post-norm encoder layer."""
    return _DropoutAdd.apply(residual, x, keep, drop_p, seed)


def dropout_mask(seed: int, drop_p: float, shape, device) -> torch.Tensor:
    """The uint8 keep mask the counter hash yields for (seed, drop_p) over a tensor of ``shape``."""
    lib = L.load()
    keep = torch.empty(shape, dtype=torch.uint8, device=device)
    L.check(lib.vts_dropout_mask(seed, drop_p, L.bptr(keep), keep.numel(), L.stream()), "vts_dropout_mask")
    return keep


# ------------------------------------------------- three-component mixture (Gaussian / GPD / Bernoulli)
class _Reparam3(torch.autograd.Function):
    """vae-models/graph/vae_mixture_graph_model.py:182-209 (use_gpd and use_bernoulli on)."""

    @staticmethod
    def forward(ctx, mean, logvar, scale, shape, logits, pi, bern, bern_is_uniform, eps, u, choice):
        lib = L.load()
        _need_gpu(mean)
        B, Ld = mean.shape
        mean, logvar, scale, shape, logits, bern, eps, u, choice = [
            _c(t) for t in (mean.detach(), logvar.detach(), scale.detach(), shape.detach(), logits.detach(), bern, eps, u, choice)]
        pi_d = pi.detach().contiguous()
        z = torch.empty(B, Ld, device=mean.device, dtype=torch.float32)
        L.check(L.timer.call("reparam3_fwd", lib.vts_reparam3_fwd, L.ptr(mean), L.ptr(logvar), L.ptr(scale), L.ptr(shape),
                             L.ptr(logits), L.ptr(bern), int(bern_is_uniform), L.ptr(eps), L.ptr(u), L.ptr(choice),
                             L.ptr(pi_d), L.ptr(z), B, Ld, L.stream()), "vts_reparam3_fwd")
        ctx.save_for_backward(logvar, scale, shape, eps, u, choice, pi_d)
        return z

    @staticmethod
    def backward(ctx, dz):
        lib = L.load()
        logvar, scale, shape, eps, u, choice, pi_d = ctx.saved_tensors
        dz = dz.contiguous()
        B, Ld = dz.shape
        dmean, dlogvar, dscale, dshape = [torch.empty(B, Ld, device=dz.device, dtype=torch.float32) for _ in range(4)]
        L.check(L.timer.call("reparam3_bwd", lib.vts_reparam3_bwd, L.ptr(dz), L.ptr(logvar), L.ptr(scale), L.ptr(shape),
                             L.ptr(eps), L.ptr(u), L.ptr(choice), L.ptr(pi_d), L.ptr(dmean), L.ptr(dlogvar), L.ptr(dscale),
                             L.ptr(dshape), B, Ld, L.stream()), "vts_reparam3_bwd")
        return dmean, dlogvar, dscale, dshape, None, None, None, None, None, None, None


def reparameterize_mixture3(mean, logvar, scale, shape, logits, pi_params, choice=None, eps=None, u=None, bern=None):
    """z of the three-component model.  Draws not fed in are made on the device in the reference's order
    (eps, u, the Bernoulli draw, choice); ``bern`` is a recorded 0/1 sample when given."""
    dev = mean.device
    if eps is None:
        eps = torch.randn_like(mean)
    if u is None:
        u = torch.rand(mean.shape, device=dev)
    uniform = bern is None
    if uniform:
        bern = torch.rand(mean.shape, device=dev)
    if choice is None:
        choice = torch.rand(mean.shape[0], device=dev)
    return _Reparam3.apply(mean, logvar, scale, shape, logits, pi_params, bern, uniform, eps, u, choice)


class _Elbo3(torch.autograd.Function):
    """vae-models/graph/vae_mixture_graph_model.py:222-285 (use_gpd and use_bernoulli on, no D-KNN term)."""

    @staticmethod
    def forward(ctx, recon, y, mean, logvar, scale, shape, logits, pi, threshold: float, beta: float):
        lib = L.load()
        _need_gpu(recon)
        B, Ld = mean.shape
        if recon.numel() != B or y.numel() != B:
            raise NotImplementedError("three-component loss: future_steps must be 1 (the reference indexes the "
                                      "[B, L] latent tensors with masks made from y)")
        recon, y, mean, logvar, scale, shape, logits = [_c(t.detach()) for t in (recon, y, mean, logvar, scale, shape, logits)]
        pi_d = pi.detach().contiguous()
        loss = torch.empty((), device=recon.device, dtype=torch.float32)
        state = torch.empty(16, device=recon.device, dtype=torch.float32)
        wsb = lib.vts_elbo3_workspace_bytes(B, Ld)
        ws = L.workspace(wsb, recon.device)
        L.check(L.timer.call("elbo3_fwd", lib.vts_elbo3_fwd, L.ptr(recon), L.ptr(y), L.ptr(mean), L.ptr(logvar), L.ptr(scale),
                             L.ptr(shape), L.ptr(logits), L.ptr(pi_d), float(threshold), float(beta), L.ptr(loss), L.ptr(state),
                             L.wptr(ws), wsb, B, Ld, L.stream()), "vts_elbo3_fwd")
        ctx.save_for_backward(recon, y, mean, logvar, scale, shape, logits, state)
        ctx.threshold, ctx.beta = float(threshold), float(beta)
        return loss

    @staticmethod
    def backward(ctx, gout):
        lib = L.load()
        recon, y, mean, logvar, scale, shape, logits, state = ctx.saved_tensors
        B, Ld = mean.shape
        dev = mean.device
        gout = gout.contiguous().reshape(1)
        drecon = torch.empty_like(recon)
        dmean, dlogvar, dscale, dshape, dlogits = [torch.empty(B, Ld, device=dev, dtype=torch.float32) for _ in range(5)]
        dpi = torch.empty(3, device=dev, dtype=torch.float32)
        L.check(L.timer.call("elbo3_bwd", lib.vts_elbo3_bwd, L.ptr(recon), L.ptr(y), L.ptr(mean), L.ptr(logvar), L.ptr(scale),
                             L.ptr(shape), L.ptr(logits), ctx.threshold, ctx.beta, L.ptr(state), L.ptr(gout), L.ptr(drecon),
                             L.ptr(dmean), L.ptr(dlogvar), L.ptr(dscale), L.ptr(dshape), L.ptr(dlogits), L.ptr(dpi), B, Ld,
                             L.stream()), "vts_elbo3_bwd")
        return drecon, None, dmean, dlogvar, dscale, dshape, dlogits, dpi, None, None


def mixture3_elbo(recon, y, mean, logvar, threshold: float, scale, shape, logits, pi_params, beta: float = 0.001) -> torch.Tensor:
    return _Elbo3.apply(recon, y, mean, logvar, scale, shape, logits, pi_params, threshold, beta)

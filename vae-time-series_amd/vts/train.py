"""Train-step driver: flat parameter/gradient storage, the fused Adam kernel and
``train_step`` (zero_grad -> forward -> loss -> backward -> Adam.step), the counterpart of the
loop bodies at gaussian-lstm-DE.py:149-156, gpd-lstm-DE.py:99-105, mixture-vae-DE.py:158-164.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional

import torch
import torch.nn as nn

from . import _lib as L


class FlatParameters:
    """Re-homes every parameter of ``module`` into one contiguous fp32 buffer (and its gradient
    into a second one), so the optimizer is a single kernel launch and the data-parallel
    all-reduce moves a few large buckets.  Parameter order = ``module.named_parameters()``.
    Offsets are rounded up to 4 floats so every view is 16-byte aligned."""

    def __init__(self, module: nn.Module):
        self.names: List[str] = []
        self.params: List[nn.Parameter] = []
        self.offsets: List[int] = []
        off = 0
        for name, p in module.named_parameters():
            if p.dtype != torch.float32:
                raise TypeError(f"{name}: vts trains float32 parameters only")
            self.names.append(name)
            self.params.append(p)
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4
        self.numel = off
        device = self.params[0].device
        self.data = torch.zeros(off, device=device, dtype=torch.float32)
        self.grad = torch.zeros(off, device=device, dtype=torch.float32)
        self.pending = False        # release_grads was called and gather_grads has not run since
        self._tables = None
        for p, o in zip(self.params, self.offsets):
            view = self.data[o:o + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
            p.grad = self.grad[o:o + p.numel()].view_as(p)

    def zero_grad(self) -> None:
        """Gradients stay views of the flat buffer; autograd accumulates into them in place."""
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view_as(p)

    def release_grads(self) -> None:
        """Detach the parameters' gradients from the flat buffer: autograd then hands every parameter the gradient
        tensor its backward node produced (no accumulation kernel per parameter), and gather_grads copies them
        into the flat buffer with one launch."""
        for p in self.params:
            p.grad = None
        self.pending = True

    def gather_grads(self, indices: Optional[Iterable[int]] = None) -> None:
        """flat.grad[slot of p] = p.grad for the given parameters (all by default); zeros where a parameter got no
        gradient, nothing where the gradient already lives in its slot."""
        if indices is None and self.grad.is_cuda and self._gather_all():
            return
        idx = list(range(len(self.params))) if indices is None else list(indices)
        base = self.grad.data_ptr()
        srcs, sizes, offs = [], [], []
        for i in idx:
            p, o = self.params[i], self.offsets[i]
            g = p.grad
            if g is not None and g.data_ptr() == base + 4 * o:
                continue
            if g is not None and (g.dtype != torch.float32 or not g.is_contiguous()):
                g = g.to(torch.float32).contiguous()
            srcs.append(g)
            sizes.append(p.numel())
            offs.append(o)
        if indices is None:
            self.pending = False
        if not srcs:
            return
        if not self.grad.is_cuda:
            # host tensors: the world-size-2 gloo test of the data-parallel logic, which runs stock CPU modules
            for g, n, o in zip(srcs, sizes, offs):
                if g is None:
                    self.grad[o:o + n].zero_()
                else:
                    self.grad[o:o + n].copy_(g.reshape(-1))
            return
        import ctypes as C
        lib = L.load()
        n = len(srcs)
        L.check(lib.vts_gather_chunks(L.ptr_array(srcs), (C.c_int64 * n)(*sizes), (C.c_int64 * n)(*offs), n,
                                      L.ptr(self.grad), L.stream()), "vts_gather_chunks")

    def _gather_all(self) -> bool:
        """gather_grads() for all parameters with the size and offset tables built once (a gradient that already
        lives in its slot is copied onto itself); False if some gradient needs converting first."""
        import ctypes as C
        n = len(self.params)
        if self._tables is None:
            self._tables = ((C.c_void_p * n)(), (C.c_int64 * n)(*[p.numel() for p in self.params]),
                            (C.c_int64 * n)(*self.offsets))
        ptrs, sizes, offs = self._tables
        for i, p in enumerate(self.params):
            g = p.grad
            if g is None:
                ptrs[i] = None
            elif g.dtype is torch.float32 and g.is_contiguous():
                ptrs[i] = g.data_ptr()
            else:
                return False
        L.check(L.load().vts_gather_chunks(ptrs, sizes, offs, n, L.ptr(self.grad), L.stream()), "vts_gather_chunks")
        self.pending = False
        return True

    def span(self, first: int, last: int):
        """[start, end) float offsets covering parameters first..last (inclusive)."""
        end = self.offsets[last] + (self.params[last].numel() + 3) // 4 * 4
        return self.offsets[first], end


class Adam:
    """torch.optim.Adam(params, lr) with default betas/eps, no weight decay, as one HIP kernel
    over a FlatParameters buffer.  ``grad_scale`` is applied to the gradient first."""

    def __init__(self, flat: FlatParameters, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        self.flat, self.lr, self.betas, self.eps = flat, lr, betas, eps
        self.exp_avg = torch.zeros_like(flat.data)
        self.exp_avg_sq = torch.zeros_like(flat.data)
        self.step_count = 0
        self.step_state: Optional[torch.Tensor] = None   # device-side counter (see count_on_device)

    def zero_grad(self) -> None:
        """optimizer.zero_grad(): the parameters' .grad become None (set_to_none, torch's default); step() collects
        what backward left there."""
        self.flat.release_grads()

    def count_on_device(self) -> None:
        """Keep the step counter in device memory from now on, so that a recorded step can be replayed
        (GraphedTrainStep): [int64 step, 2 floats of scratch]."""
        if self.step_state is None:
            self.step_state = torch.zeros(2, dtype=torch.int64, device=self.flat.data.device)
            self.step_state[0] = self.step_count

    def step(self, grad_scale: float = 1.0, guard: Optional[torch.Tensor] = None) -> None:
        """``guard``: int32 device tensor (at most 64 words); the update is skipped, on the device, if any word is
        non-zero when the kernel runs (vts_adam_step_guarded)."""
        lib = L.load()
        f = self.flat
        if f.pending:
            f.gather_grads()
        if guard is not None and self.step_state is None:
            self.step_count += 1
            L.check(lib.vts_adam_step_guarded(L.ptr(f.data), L.ptr(f.grad), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq),
                                              f.numel, self.step_count, self.lr, self.betas[0], self.betas[1], self.eps,
                                              grad_scale, L.wptr(guard), guard.numel(), L.stream()), "vts_adam_step_guarded")
            return
        if self.step_state is not None:
            L.check(lib.vts_adam_step_counted(L.ptr(f.data), L.ptr(f.grad), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq),
                                              f.numel, L.wptr(self.step_state), self.lr, self.betas[0], self.betas[1],
                                              self.eps, grad_scale, L.stream()), "vts_adam_step_counted")
            return
        self.step_count += 1
        L.check(lib.vts_adam_step(L.ptr(f.data), L.ptr(f.grad), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq),
                                  f.numel, self.step_count, self.lr, self.betas[0], self.betas[1], self.eps,
                                  grad_scale, L.stream()), "vts_adam_step")


def forward_loss(model: nn.Module, x: torch.Tensor, y: torch.Tensor,
                 noise: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    if getattr(model, "kind", None) == "mixture3":   # the graph model's argument order (x_full, x_missing, edge_index)
        outs = model(x, None, None, noise=noise)
        return model.loss_function(outs[0], y, outs[1], outs[2], None, *outs[3:])
    outs = model(x, noise)
    return model.loss_function(outs[0], y, *outs[1:])


def raise_on_recurrence_timeout() -> None:
    """Raises vts.RecurrenceTimeout if a cooperative recurrent kernel of this or an earlier step timed out waiting
    for a peer workgroup (its outputs are then NaN).  Synchronises with the device.  train_step and
    DataParallelTrainer do not call it: they guard the optimizer update on the device (the step's status words go
    to vts_adam_step_guarded) and raise from a later step, when the words have reached the host; call this at the
    end of training, or wherever a step's verdict is wanted at once.  vts.set_status_checks(False) turns both off."""
    from . import functional as F
    F.recurrence_status.raise_on_timeout()


def train_step(model: nn.Module, opt: Optional[Adam], x: torch.Tensor, y: torch.Tensor,
               noise: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
    """One optimisation step; returns the loss as a device scalar (no host sync).  With
    ``opt=None`` only forward + backward run (gradients are left in ``p.grad``)."""
    from . import functional as F
    F.recurrence_status.begin_step()
    if opt is not None:
        opt.zero_grad()
    else:
        for p in model.parameters():
            p.grad = None
    loss = forward_loss(model, x, y, noise)
    loss.backward()
    if opt is not None:
        opt.step(guard=F.recurrence_status.finish_step(x.device))
    return loss.detach()


class GraphedTrainStep:
    """One optimisation step recorded as a HIP graph and replayed: a step of the reference's own sizes
    (16 windows x 16 steps, hidden 64) is a hundred launches of a few microseconds each, so the host's
    launch rate, not the GPU, sets its duration; a replay submits them in one call.

    ``x`` / ``y`` fix the batch shape; later calls copy their batch into those buffers.  Random draws are
    made on the device inside the graph (torch's generator is graph-aware), the Adam step counter lives on
    the device.  Building the graph leaves parameters and optimizer state as they were.  Single process
    only (a gradient all-reduce is not recorded); models whose dropout seeds are drawn on the host
    (the transformer variant) are refused."""

    def __init__(self, model: nn.Module, opt: Adam, x: torch.Tensor, y: torch.Tensor, warmup: int = 3):
        if getattr(model, "kind", None) == "transformer":
            raise NotImplementedError("GraphedTrainStep: the transformer variant draws its dropout seeds on the host")
        if not x.is_cuda:
            raise RuntimeError("GraphedTrainStep needs device tensors")
        self.model, self.opt = model, opt
        self.x, self.y = x.clone(), y.clone()
        opt.count_on_device()
        keep = [t.clone() for t in (opt.flat.data, opt.exp_avg, opt.exp_avg_sq, opt.step_state)]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):       # the library's one-off set-up calls (LDS opt-in) happen here, not in the capture
            for _ in range(warmup):
                train_step(model, opt, self.x, self.y)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = train_step(model, opt, self.x, self.y)
        for dst, src in zip((opt.flat.data, opt.exp_avg, opt.exp_avg_sq, opt.step_state), keep):
            dst.copy_(src)

    def __call__(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        """Runs the recorded step on this batch; the returned loss tensor is overwritten by the next call."""
        self.x.copy_(x)
        self.y.copy_(y)
        self.graph.replay()
        return self.loss

"""Data parallelism: one process per GPU, windows sharded along the batch, one flat gradient
buffer all-reduced in a few buckets with torch.distributed (backend "nccl" = RCCL over xGMI
on ROCm; "gloo" in CPU tests).

Windows are independent (the LSTM state is zero per window), so the only exchange per step is
the gradient sum.  Every loss term of the three models is a mean over the batch, so with equal
shards the global gradient is the mean of the rank gradients.  The one exception is the
mixture's masked MSE (mean over the rows with |y| <= threshold, mixture-vae-DE.py:119): its
denominator is a per-rank count, so rank-averaging is exact only when every rank holds the
same number of such rows (always true for the reference's threshold of 95 on standardised
data, where every row qualifies).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist
import torch.nn as nn

from .train import Adam, FlatParameters, forward_loss


def shard_batch(t: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Contiguous rows [rank*B/world, (rank+1)*B/world) of a batch-first tensor."""
    B = t.shape[0]
    if B % world != 0:
        raise ValueError(f"batch {B} is not divisible by world size {world}")
    n = B // world
    return t[rank * n:(rank + 1) * n]


def plan_buckets(flat: FlatParameters, max_buckets: int = 3, min_floats: int = 1 << 18) -> List[Tuple[int, int]]:
    """Split the flat gradient into at most ``max_buckets`` contiguous [start, end) ranges of at
    least ``min_floats`` floats (payloads here are 2.7-7.3 MB: latency-bound on xGMI, so few
    large messages).  Returned last-range-first, the order in which backward finishes them."""
    n = flat.numel
    k = max(1, min(max_buckets, n // max(min_floats, 1)))
    bounds = [0]
    target = n / k
    for i, off in enumerate(flat.offsets[1:], start=1):
        if len(bounds) < k and off >= target * len(bounds):
            bounds.append(off)
    bounds.append(n)
    ranges = [(bounds[i], bounds[i + 1]) for i in range(len(bounds) - 1) if bounds[i + 1] > bounds[i]]
    return ranges[::-1]


class DataParallelTrainer:
    """Wraps a model replica: ``step(x_local, y_local, noise_local)`` runs forward/backward on
    this rank's shard, sums the flat gradient across ranks bucket by bucket (each all-reduce is
    launched asynchronously as soon as backward has produced every gradient in it, so the
    collective overlaps the remaining backward kernels), then applies Adam with the 1/world
    scaling folded into the update kernel."""

    def __init__(self, model: nn.Module, lr: float = 1e-3, group: Optional[dist.ProcessGroup] = None,
                 max_buckets: int = 3, optimizer_factory=None, exact_loss: bool = False):
        self.model = model
        self.group = group
        self.distributed = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.distributed else 1
        # exact_loss: the ELBO of the concatenated batch (one extra all-reduce of 8 floats in the forward pass)
        # instead of the mean of per-rank losses; they differ only through the mixtures' masked means, when
        # ranks hold different numbers of rows of a class.  The three-component loss keeps per-rank counts.
        self.exact_loss = bool(exact_loss) and self.distributed     # also with a 1-rank group: same code path
        # how the 8 floats of loss statistics are summed over the ranks (in place); a test may substitute its own
        self.reduce_loss_sums = lambda sums: dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=self.group)
        self.flat = FlatParameters(model)
        self.opt = (optimizer_factory or (lambda flat: Adam(flat, lr=lr)))(self.flat)
        self.buckets = plan_buckets(self.flat, max_buckets)
        self._pending: List = []
        self._remaining: List[int] = []
        # trace = True records the order of events of a step: ("grad", parameter index) when a parameter's gradient is
        # final, ("allreduce", bucket index) when a bucket's collective is issued (tests of the overlap)
        self.trace = False
        self.events: List[Tuple[str, int]] = []
        self._bucket_of: Dict[int, int] = {}
        for pi, (p, off) in enumerate(zip(self.flat.params, self.flat.offsets)):
            for bi, (s, e) in enumerate(self.buckets):
                if s <= off < e:
                    self._bucket_of[pi] = bi
        self._counts = [sum(1 for b in self._bucket_of.values() if b == bi) for bi in range(len(self.buckets))]
        self._params_of = [[pi for pi, b in sorted(self._bucket_of.items()) if b == bi] for bi in range(len(self.buckets))]
        if self.distributed:   # also with a 1-rank group: the collective path is the same code
            for pi, p in enumerate(self.flat.params):
                p.register_post_accumulate_grad_hook(self._make_hook(pi))

    def _make_hook(self, pi: int):
        def hook(_param):
            bi = self._bucket_of[pi]
            self._remaining[bi] -= 1
            if self.trace:
                self.events.append(("grad", pi))
            if self._remaining[bi] == 0:
                s, e = self.buckets[bi]
                if self.trace:
                    self.events.append(("allreduce", bi))
                self.flat.gather_grads(self._params_of[bi])      # the bucket's gradients into the flat buffer
                self._pending.append(dist.all_reduce(self.flat.grad[s:e], op=dist.ReduceOp.SUM,
                                                     group=self.group, async_op=True))
        return hook

    def step(self, x: torch.Tensor, y: torch.Tensor, noise=None, apply_update: bool = True) -> torch.Tensor:
        from . import functional as F
        F.recurrence_status.begin_step()
        self.opt.zero_grad()
        self._remaining = list(self._counts)
        self._pending = []
        self.events = []
        if self.exact_loss:
            F.set_loss_reduction(lambda sums: self.reduce_loss_sums(sums), self.world)
            try:
                loss = forward_loss(self.model, x, y, noise)
            finally:
                F.set_loss_reduction(None)
        else:
            loss = forward_loss(self.model, x, y, noise)
        loss.backward()
        if self.distributed:
            # parameters that received no gradient this step never fire their hook
            for bi, left in enumerate(self._remaining):
                if left > 0:
                    s, e = self.buckets[bi]
                    if self.trace:
                        self.events.append(("allreduce", bi))
                    self.flat.gather_grads(self._params_of[bi])
                    self._pending.append(dist.all_reduce(self.flat.grad[s:e], op=dist.ReduceOp.SUM,
                                                         group=self.group, async_op=True))
            for work in self._pending:
                work.wait()
            self.flat.pending = False       # every bucket was gathered on its way to the collective
        else:
            self.flat.gather_grads()
        if apply_update:
            # every rank must make the same decision about the update: the status words are MAX-reduced first, so a
            # launch that timed out on one rank (whose NaN gradients the all-reduce above has spread) stops them all,
            # and all of them raise RecurrenceTimeout from a later step
            reduce = None
            if self.distributed:
                reduce = lambda words: dist.all_reduce(words, op=dist.ReduceOp.MAX, group=self.group)
            guard = F.recurrence_status.finish_step(x.device, reduce)
            if guard is None:       # (also what an optimizer_factory's object without the argument gets)
                self.opt.step(grad_scale=1.0 / self.world)
            else:
                self.opt.step(grad_scale=1.0 / self.world, guard=guard)
        return loss.detach()

    def global_loss(self, loss: torch.Tensor) -> torch.Tensor:
        """Mean of the per-rank losses (equal shards)."""
        if self.world == 1 or self.exact_loss:    # exact_loss: the loss already is the whole batch's
            return loss
        out = loss.clone()
        dist.all_reduce(out, op=dist.ReduceOp.SUM, group=self.group)
        return out / self.world

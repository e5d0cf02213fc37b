"""Data-parallel ranker training (second half): every rank samples batches from its own
contiguous shard of seed users (the graph itself is replicated — 31.8 M edges of int32 CSR are 0.4 GB),
runs the encoder-decoder step locally and the few-hundred-KB of dense SAGE / MLP / BatchNorm
gradients are averaged with ONE flat all-reduce per step (RCCL over xGMI; gloo in the CPU tests).
The frozen categorical tables are identical on every rank by construction (same seed) and never
exchanged.  The reference has no distributed code."""
from __future__ import annotations

from typing import Iterable, List, Tuple

import torch as t
import torch.distributed as dist
from torch import Tensor, nn


def user_shard(num_users: int, rank: int, world: int) -> Tuple[int, int]:
    """[lo, hi) of the seed users rank `rank` draws its batches from."""
    per = (num_users + world - 1) // world
    lo = min(rank * per, num_users)
    return lo, min(lo + per, num_users)


def broadcast_parameters(model: nn.Module, src: int = 0, group=None) -> None:
    """Make every replica start from rank `src`'s weights and buffers (lazy layers must be initialised first)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for x in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(x.data, src=src, group=group)


def allreduce_gradients(params: Iterable[nn.Parameter], group=None) -> None:
    """grad <- mean over ranks, one flat collective (the payload is small: latency, not bandwidth)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    grads: List[Tensor] = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = t.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(dist.get_world_size(group))
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


def fused_step(model: nn.Module, optimizer: t.optim.Optimizer, group=None):
    """ranker_step.FusedRankerStep with the gradient all-reduce between the backward and the optimizer step."""
    from .ranker_step import FusedRankerStep
    return FusedRankerStep(model, optimizer, before_step=lambda: allreduce_gradients(model.parameters(), group))


def native_step(model: nn.Module, optimizer: t.optim.Optimizer, group=None, seed=None):
    """The data-parallel iteration on the native executor (ranker_native.NativeRankerStep, data_parallel=True): gradients in
    one flat buffer, ONE all-reduce of it, mean + Adam in one launch.  Every rank must pass the same seed only if it wants
    the same dropout masks; by default a rank's masks are keyed on its own torch seed.  Raises ValueError for a model /
    optimizer the executor does not take (callers then use fused_step)."""
    from .ranker_native import NativeRankerStep
    return NativeRankerStep(model, optimizer, data_parallel=True, group=group, seed=seed)


def train_step(model: nn.Module, optimizer: t.optim.Optimizer, batch, group=None) -> Tensor:
    """training.py:19-34 with the gradient all-reduce between backward and step."""
    from .utils.get_info import select_properties
    x, edge_index, edge_label_index, edge_label = select_properties(batch)
    optimizer.zero_grad()
    out = model(x, edge_index, edge_label_index).view(-1)
    loss = t.nn.BCEWithLogitsLoss()(out, edge_label)
    loss.backward()
    allreduce_gradients(model.parameters(), group)
    optimizer.step()
    return loss

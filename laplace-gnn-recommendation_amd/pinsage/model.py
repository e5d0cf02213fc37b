"""PinSAGE model — reference: pinsage/layers.py:121-203 (WeightedSAGEConv, SAGENet, ItemToItemScorer) and
pinsage/model.py:16-34 (PinSAGEModel.get_repr, hinge loss), with the trainable item-id feature the
reference assigns (pinsage/model.py:52-53) as the projector input.

Heavy ops on the HIP kernels: the Q / W products on mi_gemm_f32 (relu fused), the weighted
neighbourhood sum on mi_spmm_csr_f32 over the block's destination-sorted CSR with values
w_e / max(sum_e w_e, 1), the id-embedding lookup on mi_gather_rows_f32.  The row L2-normalisation,
the per-pair dot products and the hinge are torch ops on [batch]-sized tensors.
"""
from __future__ import annotations

from typing import List, Tuple

import torch as t
import torch.nn.functional as F
from torch import Tensor, nn

from .. import ops
from ..model.layers import Linear


class _EmbedRowsFn(t.autograd.Function):
    """weight[ids] on mi_gather_rows_f32; the (sparse) gradient is accumulated with index_add_."""

    @staticmethod
    def forward(ctx, weight: Tensor, ids: Tensor):
        out = t.empty(ids.numel(), weight.shape[1], device=weight.device)
        ops.gather_rows(out, weight, ids.to(t.int32).contiguous())
        ctx.save_for_backward(ids)
        ctx.shape = weight.shape
        return out

    @staticmethod
    def backward(ctx, g: Tensor):
        (ids,) = ctx.saved_tensors
        gw = t.zeros(ctx.shape, device=g.device)
        gw.index_add_(0, ids, g.contiguous())
        return gw, None


class _WeightedSumFn(t.autograd.Function):
    """agg[d] = sum_{e: dst(e)=d} val[e] * x[src(e)] on the SpMM kernel; backward on the transposed CSR."""

    @staticmethod
    def forward(ctx, x: Tensor, by_dst: ops.DeviceCSR, by_src: ops.DeviceCSR):
        y = t.empty(by_dst.n_rows, x.shape[1], device=x.device)
        ops.spmm(by_dst, x if x.stride(-1) == 1 else x.contiguous(), Y=y)
        ctx.by_src = by_src
        return y

    @staticmethod
    def backward(ctx, gy: Tensor):
        gx = t.empty(ctx.by_src.n_rows, gy.shape[1], device=gy.device)
        ops.spmm(ctx.by_src, gy.contiguous(), Y=gx)
        return gx, None, None


def block_csr(block: dict) -> Tuple[ops.DeviceCSR, ops.DeviceCSR]:
    """(CSR by destination, CSR by source) of a block with values w / clamp(sum_dst w, min=1).  A block built on the
    device (PinSAGESampler._sample_batch_device) brings both with it."""
    if "csr" in block:
        return block["csr"]
    n_src, n_dst = block["src_ids"].numel(), block["n_dst"]
    es, ed, w = block["edge_src"].contiguous(), block["edge_dst"].contiguous(), block["weights"].contiguous()
    ws = t.zeros(n_dst, device=w.device).index_add_(0, ed, w).clamp(min=1)
    val = w / ws[ed]
    by_dst = ops.coo_to_csr(ed, es, n_dst, n_src)
    by_dst.val = ops.gather_f32(val, by_dst.perm) if val.numel() else val
    by_src = ops.coo_to_csr(es, ed, n_src, n_dst)
    by_src.val = ops.gather_f32(val, by_src.perm) if val.numel() else val
    return by_dst, by_src


class WeightedSAGEConv(nn.Module):
    def __init__(self, input_dims: int, hidden_dims: int, output_dims: int):
        super().__init__()
        self.Q = Linear(input_dims, hidden_dims)
        self.W = Linear(input_dims + hidden_dims, output_dims)
        self.dropout = nn.Dropout(0.5)
        self.reset_parameters()

    def reset_parameters(self) -> None:
        gain = nn.init.calculate_gain("relu")
        nn.init.xavier_uniform_(self.Q.weight, gain=gain)
        nn.init.xavier_uniform_(self.W.weight, gain=gain)
        nn.init.constant_(self.Q.bias, 0)
        nn.init.constant_(self.W.bias, 0)

    def forward(self, block: dict, h_src: Tensor, h_dst: Tensor) -> Tensor:
        n = self.Q(self.dropout(h_src), relu=True)
        by_dst, by_src = block_csr(block)
        agg = _WeightedSumFn.apply(n, by_dst, by_src)          # = (sum_e w n_src) / clamp(sum_e w, 1)
        z = self.W(self.dropout(t.cat([agg, h_dst], 1)), relu=True)
        z_norm = z.norm(2, 1, keepdim=True)
        z_norm = t.where(z_norm == 0, t.ones_like(z_norm), z_norm)
        return z / z_norm


class PinSAGEModel(nn.Module):
    def __init__(self, n_items: int, hidden_dims: int, n_layers: int):
        super().__init__()
        self.proj = nn.Embedding(n_items + 1, hidden_dims)     # LinearProjector over the `id` feature
        nn.init.xavier_uniform_(self.proj.weight)
        self.convs = nn.ModuleList([WeightedSAGEConv(hidden_dims, hidden_dims, hidden_dims) for _ in range(n_layers)])
        self.bias = nn.Parameter(t.zeros(n_items, 1))          # ItemToItemScorer

    def get_repr(self, blocks: List[dict]) -> Tensor:
        h = _EmbedRowsFn.apply(self.proj.weight, blocks[0]["src_ids"])
        last = blocks[-1]
        h_dst_final = _EmbedRowsFn.apply(self.proj.weight, last["src_ids"][: last["n_dst"]])
        for conv, block in zip(self.convs, blocks):
            h = conv(block, h, h[: block["n_dst"]])
        return h_dst_final + h

    def score(self, h: Tensor, seeds: Tensor, pair) -> Tensor:
        u, v = pair
        return (h[u] * h[v]).sum(1, keepdim=True) + self.bias[seeds[u]] + self.bias[seeds[v]]

    def forward(self, seeds: Tensor, pos, neg, blocks: List[dict]) -> Tensor:
        h = self.get_repr(blocks)
        return (self.score(h, seeds, neg) - self.score(h, seeds, pos) + 1).clamp(min=0)


def train_epoch(model: PinSAGEModel, optimizer: t.optim.Optimizer, sampler, batches: int, group=None) -> List[float]:
    """pinsage/model.py:118-131: hinge loss mean over the batch's pairs, Adam.

    Under torch.distributed (BASELINE configs[4]: 4 GPUs) the run is data-parallel: every rank owns a replica and a
    sampler with its own seed (the item-item walks need the whole graph, 0.4 GB of int32 CSR, so it is replicated),
    and the gradients — dense layers plus the id-embedding table — are averaged with one flat all-reduce per
    step (dist_ranker.allreduce_gradients; a no-op when not initialised)."""
    import torch.distributed as dist
    from ..dist_ranker import allreduce_gradients
    from .native import NativePinSAGEStep
    losses = []
    model.train()
    # one C call per iteration where the model / optimizer are the executor's.  Data-parallel: the executor's compact-row mode
    # (the ranks exchange the batch's gradient rows, not the dense 27 MB table gradient) when the sampler's bounds are known;
    # every rank must take the same path, which it does: the choice depends on the model, the optimizer and the sampler's
    # configuration only.
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    native = None
    if NativePinSAGEStep.supports(model, optimizer):
        if not multi:
            native = NativePinSAGEStep(model, optimizer)
        elif all(hasattr(sampler, a) for a in ("batch_size", "T", "n_layers")):
            native = NativePinSAGEStep(model, optimizer, data_parallel=True, group=group,
                                       seed=(t.initial_seed() + dist.get_rank(group)) & ((1 << 63) - 1))
            native.exchange_capacity = (3 * sampler.batch_size * (1 + sampler.T) ** sampler.n_layers, 3 * sampler.batch_size)
    # the backward graph is a chain of small nodes: running it in the calling thread saves the hand-over to autograd's
    # device thread at every one of them (as training.train_with_dataloader does for the ranker); the losses are read
    # back once per epoch, not once per step
    with t.autograd.set_multithreading_enabled(False):
        # sampler.batches: batch i + 1 is drawn on a side stream while this loop trains on batch i
        source = sampler.batches(batches) if hasattr(sampler, "batches") else (sampler.sample_batch() for _ in range(batches))
        for b in source:
            loss = native.step(b) if native is not None else None
            if loss is not None:
                losses.append(loss[0])
                continue
            # a declined batch: in a data-parallel run the executor's decline is collective (one all-reduce(MIN) of a flag before
            # anything is enqueued, pinsage/native.py), so EVERY rank is here with its own batch and the autograd iteration
            # below — with its dense all-reduce — runs on all of them
            loss = model(b["seeds"], b["pos"], b["neg"], b["blocks"]).mean()
            optimizer.zero_grad()
            loss.backward()
            allreduce_gradients(model.parameters(), group)
            optimizer.step()
            losses.append(loss.detach())
    return t.stack(losses).cpu().tolist() if losses else []

"""LightGCN — same constructor, attributes and 4-tuple forward as the reference's
model/lightgcn.py:11-87, with the propagate (torch_sparse.matmul + gcn_norm + stack/mean)
running as K launches of the fused HIP SpMM (csrc/spmm.hip).

Layout: users_emb.weight [U, D] and items_emb.weight [I, D] are two views of ONE contiguous
[U+I, D] fp32 table, so E^0 = cat(users, items) (model/lightgcn.py:58) is free and the first
propagate reads the parameters in place.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch as t
from torch import Tensor, nn

from .. import ops
from ..sparse import SparseTensor, as_sparse_tensor


def _joint_view(a: Tensor, b: Tensor) -> Optional[Tensor]:
    """[rows_a + rows_b, D] view when `b` starts exactly where `a` ends in the same storage."""
    if (a.is_contiguous() and b.is_contiguous() and a.dim() == 2 and b.dim() == 2 and a.shape[1] == b.shape[1]
            and a.dtype == b.dtype and a.device == b.device
            and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
            and b.storage_offset() == a.storage_offset() + a.numel()):
        return a.new_empty(0).set_(a.untyped_storage(), a.storage_offset(),
                                   (a.shape[0] + b.shape[0], a.shape[1]), (a.shape[1], 1))
    return None


def propagate_mean(adj: ops.DeviceCSR, e0: Tensor, num_iterations: int, out: Optional[Tensor] = None,
                   scratch: Optional[Tuple[Tensor, Tensor]] = None) -> Tensor:
    """final = mean_k(A^k e0), k = 0..K (model/lightgcn.py:58-68), K fused SpMM launches.

    The layer sum rides in the SpMM epilogue: no [N, K+1, D] stack, no separate mean pass.
    """
    K = int(num_iterations)
    n, d = e0.shape
    final = out if out is not None else t.empty_like(e0)
    if K == 0:
        final.copy_(e0)
        return final
    scale = 1.0 / (K + 1)
    if K == 1:
        ops.spmm(adj, e0, addend=e0, S=final, scale=scale)
        return final
    xa, xb = scratch if scratch is not None else (t.empty_like(e0), t.empty_like(e0))
    # layer 1: X1 = A e0 ; S = e0 + X1
    ops.spmm(adj, e0, Y=xa, addend=e0, S=final, scale=1.0)
    cur, nxt = xa, xb
    for _ in range(2, K):  # layers 2..K-1: X_k = A X_{k-1}; S += X_k
        ops.spmm(adj, cur, Y=nxt, addend=final, S=final, scale=1.0)
        cur, nxt = nxt, cur
    # layer K: final = (S + A X_{K-1}) / (K+1); X_K itself is never stored
    ops.spmm(adj, cur, addend=final, S=final, scale=scale)
    return final


def propagate_mean_backward(adj_t: ops.DeviceCSR, g_final: Tensor, num_iterations: int,
                            scratch: Optional[Tuple[Tensor, Tensor]] = None,
                            pre_scaled: bool = False) -> Tensor:
    """Gradient of propagate_mean wrt e0: g_0 where g_K = c*G, g_k = c*G + A^T g_{k+1}, c = 1/(K+1).

    pre_scaled: g_final already holds c*G (the fused BPR kernel writes it that way).
    """
    K = int(num_iterations)
    gc = g_final if pre_scaled else g_final * (1.0 / (K + 1))
    if K == 0:
        return gc
    ga, gb = scratch if scratch is not None else (t.empty_like(gc), t.empty_like(gc))
    cur = gc
    bufs = [ga, gb]
    for i in range(K):
        nxt = bufs[i % 2]
        ops.spmm(adj_t, cur, addend=gc, S=nxt, scale=1.0)
        cur = nxt
    return cur


class _PropagateFn(t.autograd.Function):
    @staticmethod
    def forward(ctx, users_w: Tensor, items_w: Tensor, adj_fwd, adj_bwd, K: int):
        e0 = _joint_view(users_w, items_w)
        if e0 is None:
            e0 = t.cat([users_w, items_w])
        final = propagate_mean(adj_fwd, e0.detach(), K)
        ctx.adj_bwd, ctx.K, ctx.nu = adj_bwd, K, users_w.shape[0]
        return final[: ctx.nu], final[ctx.nu:]

    @staticmethod
    def backward(ctx, gu: Optional[Tensor], gi: Optional[Tensor]):
        nu = ctx.nu
        ni = ctx.adj_bwd.n_rows - nu
        ref = gu if gu is not None else gi
        if gu is None:
            gu = ref.new_zeros((nu, ref.shape[1]))
        if gi is None:
            gi = ref.new_zeros((ni, ref.shape[1]))
        G = _joint_view(gu, gi)
        if G is None:
            G = t.cat([gu, gi])
        g0 = propagate_mean_backward(ctx.adj_bwd, G.contiguous(), ctx.K)
        return g0[:nu], g0[nu:], None, None, None


class LightGCN(nn.Module):
    """LightGCN Model as proposed in https://arxiv.org/abs/2002.02126 (reference: model/lightgcn.py)."""

    def __init__(self, num_users, num_items, embedding_dim: int, num_iterations: int, add_self_loops=False):
        super().__init__()
        self.num_users, self.num_items = int(num_users), int(num_items)
        self.embedding_dim, self.num_iterations = int(embedding_dim), int(num_iterations)
        self.add_self_loops = add_self_loops
        if self.embedding_dim % 4 != 0 or self.embedding_dim > 512:
            raise ValueError("embedding_dim must be a multiple of 4 and <= 512 for the HIP propagate")

        table = t.empty(self.num_users + self.num_items, self.embedding_dim)
        # same draw order as the reference: users first, then items (model/lightgcn.py:43-44)
        nn.init.normal_(table[: self.num_users], std=0.1)
        nn.init.normal_(table[self.num_users:], std=0.1)
        self.users_emb = nn.Embedding(self.num_users, self.embedding_dim, _weight=table[: self.num_users])  # e_u^0
        self.items_emb = nn.Embedding(self.num_items, self.embedding_dim, _weight=table[self.num_users:])  # e_i^0

    # keep the two weights adjacent in one buffer across .to()/.cuda()/.float()
    def _apply(self, fn, *args, **kwargs):
        super()._apply(fn, *args, **kwargs)
        self._fuse_tables()
        return self

    def _fuse_tables(self) -> None:
        u, i = self.users_emb.weight, self.items_emb.weight
        if _joint_view(u.data, i.data) is not None:
            return
        table = t.empty(self.num_users + self.num_items, self.embedding_dim, dtype=u.dtype, device=u.device)
        table[: self.num_users].copy_(u.data)
        table[self.num_users:].copy_(i.data)
        u.data = table[: self.num_users]
        i.data = table[self.num_users:]

    def table(self) -> Tensor:
        """E^0 as one [U+I, D] tensor (a view of the two parameters)."""
        self._fuse_tables()
        return _joint_view(self.users_emb.weight.data, self.items_emb.weight.data)

    def forward(self, edge_index: SparseTensor):
        """Returns e_u^K, e_u^0, e_i^K, e_i^0 exactly like model/lightgcn.py:46-80."""
        edge_index = as_sparse_tensor(edge_index)   # also torch sparse CSR / COO and torch_sparse-like objects
        n = self.num_users + self.num_items
        if edge_index.sparse_sizes() != (n, n):
            raise ValueError(f"adjacency must be ({n}, {n}), got {edge_index.sparse_sizes()}")
        self._fuse_tables()
        adj_fwd, adj_bwd = edge_index.gcn_normalized(self.add_self_loops)
        users_final, items_final = _PropagateFn.apply(
            self.users_emb.weight, self.items_emb.weight, adj_fwd, adj_bwd, self.num_iterations)
        return users_final, self.users_emb.weight, items_final, self.items_emb.weight

"""LightGCN loss / evaluation helpers with the reference's call signatures
(utils/metrics_lightgcn.py), running on the HIP kernels.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch as t
from torch import Tensor

from .. import ops


class _BprLossFn(t.autograd.Function):
    """bpr_loss over six already-gathered [B, D] blocks through mi_bpr_fwd_bwd_f32.

    The blocks are laid out as two 3B-row tables (final | layer-0) and addressed by arange, so the
    same fused kernel serves this drop-in signature and the trainer's index-based fast path.
    """

    @staticmethod
    def forward(ctx, uf, u0, pf, p0, nf, n0, lambda_val: float):
        B, _ = uf.shape
        final = t.cat([uf, pf, nf]).contiguous()
        e0 = t.cat([u0, p0, n0]).contiguous()
        dev = uf.device
        users = t.arange(B, device=dev)
        neg = t.arange(B, 2 * B, device=dev)
        g_final = t.zeros_like(final)
        reg_w = t.zeros(3 * B, device=dev)
        loss = ops.bpr_fwd_bwd(users, users, neg, final, e0, B, lambda_val, g_final=g_final, reg_w=reg_w)
        ctx.save_for_backward(g_final, reg_w, e0)
        ctx.B = B
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        g_final, reg_w, e0 = ctx.saved_tensors
        B = ctx.B
        gf = g_final * grad_out
        g0 = (reg_w[:, None] * e0) * grad_out
        return gf[:B], g0[:B], gf[B:2 * B], g0[B:2 * B], gf[2 * B:], g0[2 * B:], None


def bpr_loss(users_emb_final: Tensor, users_emb_0: Tensor, pos_items_emb_final: Tensor, pos_items_emb_0: Tensor,
             neg_items_emb_final: Tensor, neg_items_emb_0: Tensor, lambda_val: float) -> Tensor:
    """-mean(softplus(pos - neg)) + lambda * (|u0|^2 + |p0|^2 + |n0|^2), as written in the reference
    (utils/metrics_lightgcn.py:9-45; note the sign convention)."""
    return _BprLossFn.apply(users_emb_final, users_emb_0, pos_items_emb_final, pos_items_emb_0,
                            neg_items_emb_final, neg_items_emb_0, float(lambda_val))


# ----------------------------------------------------------------------------------------------
# evaluation: batched on device (reference: per-user Python loops on the CPU)
# ----------------------------------------------------------------------------------------------
def create_adj_dict(edge_index: Tensor, from_nodes: Optional[Tensor] = None) -> dict:
    """{user: tensor of that user's items, in edge order} (utils/metrics_lightgcn.py:48-61).
    One stable sort instead of one mask per user."""
    src, dst = edge_index[0], edge_index[1]
    order = t.argsort(src, stable=True)
    s_sorted, d_sorted = src[order], dst[order]
    users = t.unique(src) if from_nodes is None else from_nodes
    lo = t.searchsorted(s_sorted, users)
    hi = t.searchsorted(s_sorted, users, right=True)
    return {int(u): d_sorted[int(a):int(b)] for u, a, b in zip(users.tolist(), lo.tolist(), hi.tolist())}


def create_adj_list(edge_index: Tensor, from_nodes: Optional[Tensor] = None) -> List[Tensor]:
    users = edge_index[0].unique(sorted=True) if from_nodes is None else from_nodes
    d = create_adj_dict(edge_index, from_nodes=users)
    return [d[int(u)] for u in users.tolist()]


def _exclusion_csr(users: Tensor, exclude_edges: Optional[Tensor], num_items: int) -> Optional[ops.DeviceCSR]:
    """CSR with one row per position in `users` holding that user's excluded item ids."""
    if exclude_edges is None or exclude_edges.numel() == 0:
        return None
    eu, ei = exclude_edges[0], exclude_edges[1]
    pos = t.searchsorted(users, eu)
    pos_c = pos.clamp(max=users.numel() - 1)
    keep = users[pos_c] == eu
    return ops.coo_to_csr(pos_c[keep].contiguous(), ei[keep].contiguous(), users.numel(), num_items, want_perm=False)


def topk_for_users(user_embedding: Tensor, item_embedding: Tensor, users: Tensor, exclude_edges: Optional[Tensor],
                   k: int) -> Tensor:
    """ids [len(users), k] — make_predictions_for_user for a sorted batch of users at once."""
    excl = _exclusion_csr(users, exclude_edges, item_embedding.shape[0])
    return ops.topk_excl(users.contiguous(), user_embedding, item_embedding, k, excl)


def make_predictions_for_user(user_embeddings: Tensor, article_embeddings: Tensor, user_id: int,
                              positive_items_for_user: dict, num_recommendations: int) -> Tensor:
    """Single-user form with the reference's signature (utils/metrics_lightgcn.py:125-142)."""
    dev = article_embeddings.device
    users = t.tensor([user_id], dtype=t.int64, device=dev)
    ignore = positive_items_for_user.get(user_id)
    excl = None
    if ignore is not None and len(ignore):
        ig = t.as_tensor(ignore, dtype=t.int64, device=dev)
        excl = t.stack([t.full_like(ig, user_id), ig])
    out = topk_for_users(user_embeddings, article_embeddings, users, excl, num_recommendations)[0]
    return out[out >= 0]


def rank_metrics(r: Tensor, gt_len: Tensor, k: int) -> Tuple[float, float, float]:
    """recall, precision, ndcg @ k from the hit matrix r [n, k] and |GT| per row — the arithmetic of
    utils/metrics.py:6-57 on tensors (device or host)."""
    rf = r.to(t.float32)
    hits = rf.sum(dim=-1)
    recall = (hits / gt_len.to(t.float32)).mean().item()
    precision = (hits.mean() / k).item()
    disc = 1.0 / t.log2(t.arange(2, k + 2, device=r.device, dtype=t.float32))
    ideal = (t.arange(k, device=r.device)[None, :] < gt_len.clamp(max=k)[:, None]).to(t.float32)
    idcg = (ideal * disc).sum(dim=1)
    dcg = (rf * disc).sum(dim=1)
    idcg[idcg == 0.0] = 1.0
    ndcg = dcg / idcg
    ndcg[t.isnan(ndcg)] = 0.0
    return recall, precision, ndcg.mean().item()


def get_metrics_lightgcn(model, edge_index: Tensor, exclude_edge_indices: List[Tensor], k: int
                         ) -> Tuple[float, float, float]:
    """recall / precision / ndcg @ k of the layer-0 embeddings on the users of `edge_index`,
    never recommending items in `exclude_edge_indices` (utils/metrics_lightgcn.py:79-122)."""
    ue = model.users_emb.weight.detach()
    ie = model.items_emb.weight.detach()
    dev = ie.device
    edge_index = edge_index.to(dev)
    n_items = ie.shape[0]
    users = edge_index[0].unique()
    excl = t.cat([e.to(dev) for e in exclude_edge_indices], dim=1) if len(exclude_edge_indices) else None
    top = topk_for_users(ue, ie, users, excl, k)
    # r[u, j] = top[u, j] is one of u's positives in this split
    pos_of = t.searchsorted(users, edge_index[0])
    truth_keys = t.unique(pos_of * n_items + edge_index[1])
    pred_keys = t.arange(users.numel(), device=dev)[:, None] * n_items + top
    r = t.isin(pred_keys, truth_keys) & (top >= 0)
    gt_len = t.bincount(pos_of, minlength=users.numel())
    return rank_metrics(r, gt_len, k)

"""LightGCN loss / evaluation helpers with the reference's call signatures
(utils/metrics_lightgcn.py), running on the HIP kernels.
"""
from __future__ import annotations

from typing import Tuple

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

"""TEST-ONLY kernel provider: the ops interface of laplace_amd.ops implemented with the CPU oracle,
so the host logic of the sharded trainer (partitioning, collectives, step order) can be exercised
with gloo on machines without a GPU.  The product never imports this."""
from typing import Optional, Tuple

import torch as t
from torch import Tensor

from laplace_amd.ops import DeviceCSR
from oracle import lightgcn_ref as R


class _Plan:
    n_items = 0
    n_long_rows = 0


def coo_to_csr(row, col, n_rows, n_cols, want_perm=True) -> DeviceCSR:
    rowptr, col_s, perm = R.sparse_tensor_csr(row, col, n_rows, n_cols)
    return DeviceCSR(n_rows, n_cols, rowptr.to(t.int32), col_s.to(t.int32), None, perm.to(t.int32) if want_perm else None)


def scale_csr(a: DeviceCSR, val_in=None, row_scale=None, col_scale=None) -> Tensor:
    counts = (a.rowptr[1:] - a.rowptr[:-1]).long()
    row = t.repeat_interleave(t.arange(a.n_rows), counts)
    v = t.ones(a.nnz) if val_in is None else val_in.clone()
    if row_scale is not None:
        v = v * row_scale[row]
    if col_scale is not None:
        v = v * col_scale[a.col.long()]
    return v


def row_slice(a: DeviceCSR, r0: int, r1: int) -> DeviceCSR:
    return DeviceCSR(r1 - r0, a.n_cols, a.rowptr[r0:r1 + 1], a.col, a.val, None)


def build_spmm_plan(a, chunk=256, band=None):
    return _Plan()


def spmm(a: DeviceCSR, X: Tensor, *, Y=None, addend=None, S=None, scale=1.0, x_map=None, addend_map=None,
         row_list=None, n_list_dev=None, adam=None, x_rare=False) -> None:
    """Dense semantics of every sparse-operand form: expand, multiply, then select (x_rare is a speed hint: ignored)."""
    d = X.shape[1]
    if x_map is not None:
        Xd = t.zeros(a.n_cols, d)
        nz = (x_map >= 0).nonzero().view(-1)
        Xd[nz] = X[x_map[nz].long()]
        X = Xd
    acc = R.spmm_c(a.rowptr, a.col, a.val, X)
    if row_list is not None:
        n = int(n_list_dev[0]) if n_list_dev is not None else row_list.numel()
        accs = acc[row_list[:n].long()]
        if Y is not None:
            Y[:n] = accs
        if S is not None:
            base = addend[:n].clone() if addend is not None else 0.0
            S[:n] = scale * (base + accs)
        return
    base = 0.0
    if addend is not None:
        if addend_map is not None:
            base = t.zeros(a.n_rows, d)
            nz = (addend_map >= 0).nonzero().view(-1)
            base[nz] = addend[addend_map[nz].long()]
        else:
            base = addend.clone()
    if Y is not None:
        Y.copy_(acc)
    if S is not None:
        S.copy_(scale * (base + acc))
    if adam is not None:  # optimizer epilogue = the separate Adam pass on the S value
        kw = {k: adam[k] for k in ("step", "lr", "beta1", "beta2", "eps", "reg_w") if k in adam}
        adam_step(adam["p"], scale * (base + acc), adam["m"], adam["v"], **kw)


def batch_nodes(users, pos, neg, n_users, n_nodes, *, gmap=None, nodes=None, count=None, ws=None):
    uniq = t.unique(t.cat([users, n_users + pos, n_users + neg]))
    gmap = gmap if gmap is not None else t.empty(n_nodes, dtype=t.int32)
    nodes = nodes if nodes is not None else t.zeros(3 * users.numel(), dtype=t.int32)
    count = count if count is not None else t.zeros(2, dtype=t.int32)
    gmap.fill_(-1)
    gmap[uniq] = t.arange(uniq.numel(), dtype=t.int32)
    nodes[: uniq.numel()] = uniq.to(t.int32)
    count[0] = uniq.numel()
    count[1] = int((uniq < n_users).sum())
    return gmap, nodes, count


def _span(rows, n_dev, begin_dev):
    hi = int(n_dev[0]) if n_dev is not None else rows.numel()
    lo = int(begin_dev[0]) if begin_dev is not None else 0
    return lo, hi


def gather_rows(dst, src, rows, n_dev=None, accumulate=False, scale=1.0, begin_dev=None, row_offset=0) -> None:
    lo, hi = _span(rows, n_dev, begin_dev)
    g = src[rows[lo:hi].long() - row_offset]
    dst[lo:hi] = scale * ((dst[lo:hi] if accumulate else 0.0) + g)


def scatter_rows(dst, src, rows, n_dev=None, begin_dev=None, row_offset=0) -> None:
    lo, hi = _span(rows, n_dev, begin_dev)
    dst[rows[lo:hi].long() - row_offset] = src[lo:hi]


def expand_rows(a: DeviceCSR) -> Tensor:
    return t.repeat_interleave(t.arange(a.n_rows), (a.rowptr[1:] - a.rowptr[:-1]).long()).to(t.int32)


def sample_bpr_batch(r: DeviceCSR, row_of_edge, batch, neg_range, seed, step, quirk=False, out=None,
                     edges_in_order=False, no_self_loops=False):
    u, p, n = R.sample_bpr_batch_philox(r.rowptr.long(), r.col.long(), batch, neg_range, seed, step, quirk,
                                        edges_in_order, no_self_loops)
    if out is not None:
        for dst, src in zip(out, (u, p, n)):
            dst.copy_(src)
        return out
    return u, p, n


def bpr_fwd_bwd(users, pos, neg, final_emb, e0, n_users, lambda_val, *, g_final=None, reg_w=None, g_scale=1.0,
                reg_scale=1.0, loss_out=None, node_map=None) -> Tensor:
    U = n_users
    f = (lambda idx: node_map[idx].long()) if node_map is not None else (lambda idx: idx)
    rows = [final_emb[f(users)], e0[users], final_emb[f(U + pos)], e0[U + pos], final_emb[f(U + neg)], e0[U + neg]]
    rows = [x.detach().requires_grad_(True) for x in rows]
    loss = R.bpr_loss(*rows, lambda_val)
    if g_final is not None:
        grads = t.autograd.grad(loss, rows)
        for idx, gf in ((users, grads[0]), (U + pos, grads[2]), (U + neg, grads[4])):
            g_final.index_add_(0, f(idx), g_scale * gf)
            if reg_w is not None:
                reg_w.index_add_(0, idx, t.full((idx.numel(),), 2.0 * lambda_val * reg_scale))
    if loss_out is None:
        loss_out = t.empty(1)
    loss_out[0] = loss.detach()
    return loss_out


def adam_step(p, grad, m, v, *, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, reg_w=None) -> None:
    g = grad if reg_w is None else grad + reg_w[:, None] * p
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
    p.addcdiv_(m, (v.sqrt() / bc2 ** 0.5).add_(eps), value=-lr / bc1)

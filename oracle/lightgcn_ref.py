"""ORACLE — test infrastructure only (see oracle/__init__.py for the pinning status).

Plain torch/numpy CPU restatement of the reference's LightGCN path.  Citations are
`path:line` relative to the reference root.
"""
from __future__ import annotations

import ctypes
import os
import random
from typing import List, Optional, Tuple

import numpy as np
import torch as t
from torch import Tensor

from .philox import philox4x32

_HERE = os.path.dirname(os.path.abspath(__file__))
_CLIB = None


def clib() -> ctypes.CDLL:
    """The C restatement (oracle/spmm_ref.c); built on demand with `make`."""
    global _CLIB
    if _CLIB is None:
        path = os.path.join(_HERE, "liboracle_ref.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.run(["make", "-s", "-C", _HERE], check=True)
        L = ctypes.CDLL(path)
        i64, p, f = ctypes.c_int64, ctypes.c_void_p, ctypes.c_float
        L.ref_num_threads.restype = ctypes.c_int
        L.ref_spmm_csr_f32.argtypes = [i64, i64, p, p, p, p, i64, p, i64]
        L.ref_spmm_csr_f64acc.argtypes = [i64, i64, p, p, p, p, i64, p, i64]
        L.ref_scores_fma_f32.argtypes = [i64, i64, i64, p, i64, p, i64, p, i64]
        L.ref_adam_f32.argtypes = [i64, p, p, p, p, f, f, f, f, f]
        L.ref_gemm_fma_f32.argtypes = [i64, i64, i64, p, i64, i64, p, i64, i64, p, p, i64, ctypes.c_int, ctypes.c_int]
        _CLIB = L
    return _CLIB


# ----------------------------------------------------------------------------------------------
# data/lightgcn_loader.py
# ----------------------------------------------------------------------------------------------
def both_indexes_from_zero(edge_index: Tensor) -> Tensor:
    """data/lightgcn_loader.py:39-43 — re-base the item row by max(user)+1."""
    new_edge_index = t.clone(edge_index)
    new_edge_index[1] = new_edge_index[1] - (t.max(new_edge_index[0]) + 1)
    return new_edge_index


def sparse_tensor_csr(row: Tensor, col: Tensor, n_rows: int, n_cols: int
                      ) -> Tuple[Tensor, Tensor, Tensor]:
    """torch_sparse.SparseTensor(row=, col=, sparse_sizes=) storage (data/lightgcn_loader.py:65-69):
    entries sorted by row*n_cols+col, duplicates kept; rowptr = ind2ptr(row).
    Returns (rowptr int64[n_rows+1], col int64[nnz], perm int64[nnz])."""
    key = row.to(t.int64) * n_cols + col.to(t.int64)
    perm = t.argsort(key, stable=True)
    row_s, col_s = row[perm], col[perm]
    counts = t.bincount(row_s, minlength=n_rows)
    rowptr = t.zeros(n_rows + 1, dtype=t.int64)
    rowptr[1:] = t.cumsum(counts, 0)
    return rowptr, col_s.to(t.int64), perm


def gcn_norm_csr(rowptr: Tensor, col: Tensor, val: Optional[Tensor] = None) -> Tensor:
    """torch_geometric.nn.conv.gcn_conv.gcn_norm(SparseTensor, add_self_loops=False), the
    SparseTensor branch, called at model/lightgcn.py:56:
        adj_t = adj_t.fill_value(1.) if no value
        deg = sparsesum(adj_t, dim=1); dis = deg.pow_(-0.5); dis[dis == inf] = 0
        adj_t = mul(adj_t, dis.view(-1, 1)); adj_t = mul(adj_t, dis.view(1, -1))
    """
    n = rowptr.numel() - 1
    counts = (rowptr[1:] - rowptr[:-1])
    row = t.repeat_interleave(t.arange(n), counts)
    v = t.ones(col.numel(), dtype=t.float32) if val is None else val.to(t.float32)
    deg = t.zeros(n, dtype=t.float32).index_add_(0, row, v)
    dis = deg.pow(-0.5)
    dis.masked_fill_(dis == float("inf"), 0.0)
    return (v * dis[row]) * dis[col]


def fill_diag_ones(row: Tensor, col: Tensor, n: int) -> Tuple[Tensor, Tensor]:
    """torch_sparse.fill_diag(adj, 1.) as used by gcn_norm(add_self_loops=True)."""
    off = row != col
    loop = t.arange(n, dtype=row.dtype)
    return t.cat([row[off], loop]), t.cat([col[off], loop])


def spmm_c(rowptr: Tensor, col: Tensor, val: Tensor, X: Tensor) -> Tensor:
    """torch_sparse.matmul(adj_t, x) (model/lightgcn.py:87) by the C restatement of spmm_cpu."""
    L = clib()
    rp = rowptr.to(t.int32).contiguous()
    c = col.to(t.int32).contiguous()
    v = val.to(t.float32).contiguous()
    X = X.to(t.float32).contiguous()
    n, d = rp.numel() - 1, X.shape[1]
    Y = t.empty(n, d, dtype=t.float32)
    L.ref_spmm_csr_f32(n, d, rp.data_ptr(), c.data_ptr(), v.data_ptr(), X.data_ptr(), d, Y.data_ptr(), d)
    return Y


def spmm_torch(rowptr: Tensor, col: Tensor, val: Tensor, X: Tensor) -> Tensor:
    """Same product through torch's own CSR kernel (differentiable wrt X) — independent of the C file."""
    n = rowptr.numel() - 1
    A = t.sparse_csr_tensor(rowptr.to(t.int64), col.to(t.int64), val.to(X.dtype), size=(n, X.shape[0]))
    return A @ X


def lightgcn_forward(users_w: Tensor, items_w: Tensor, row: Tensor, col: Tensor, num_iterations: int,
                     add_self_loops: bool = False, dtype=t.float32, use_c: bool = False):
    """LightGCN.forward (model/lightgcn.py:46-80): gcn_norm -> cat -> K x propagate -> stack/mean -> split.

    (row, col) is the COO content of the SparseTensor passed as `edge_index`, square (U+I).
    Differentiable wrt users_w / items_w unless use_c.  Returns the reference's 4-tuple.
    """
    U, I = users_w.shape[0], items_w.shape[0]
    n = U + I
    if add_self_loops:
        row, col = fill_diag_ones(row, col, n)
    rowptr, col_s, _ = sparse_tensor_csr(row, col, n, n)
    val = gcn_norm_csr(rowptr, col_s)
    emb_0 = t.cat([users_w, items_w]).to(dtype)
    embs = [emb_0]
    emb_k = emb_0
    for _ in range(num_iterations):
        emb_k = spmm_c(rowptr, col_s, val, emb_k) if use_c else spmm_torch(rowptr, col_s, val.to(dtype), emb_k)
        embs.append(emb_k)
    embs = t.stack(embs, dim=1)
    emb_final = t.mean(embs, dim=1)
    users_final, items_final = t.split(emb_final, [U, I])
    return users_final, users_w, items_final, items_w


def lightgcn_forward_dense64(users_w: Tensor, items_w: Tensor, row: Tensor, col: Tensor, num_iterations: int):
    """Known-answer form: dense float64  A~ = diag(dis) A diag(dis),  mean_k(A~^k E0)."""
    U, I = users_w.shape[0], items_w.shape[0]
    n = U + I
    A = t.zeros(n, n, dtype=t.float64)
    A.index_put_((row, col), t.ones(row.numel(), dtype=t.float64), accumulate=True)
    deg = A.sum(1)
    dis = t.where(deg > 0, deg.pow(-0.5), t.zeros_like(deg))
    An = dis[:, None] * A * dis[None, :]
    E = t.cat([users_w, items_w]).to(t.float64)
    acc, cur = E.clone(), E
    for _ in range(num_iterations):
        cur = An @ cur
        acc = acc + cur
    fin = acc / (num_iterations + 1)
    return fin[:U], fin[U:]


def bipartite_edges(user: Tensor, item: Tensor, num_users: int) -> Tuple[Tensor, Tensor]:
    """Symmetric (U+I)x(U+I) bipartite pattern: (u, U+i) and (U+i, u) — the `compat="bipartite"`
    adjacency (what LightGCN's paper prescribes; the reference loader's is SURVEY F7)."""
    r = t.cat([user, item + num_users])
    c = t.cat([item + num_users, user])
    return r, c


# ----------------------------------------------------------------------------------------------
# utils/metrics_lightgcn.py
# ----------------------------------------------------------------------------------------------
def bpr_loss(users_emb_final, users_emb_0, pos_items_emb_final, pos_items_emb_0,
             neg_items_emb_final, neg_items_emb_0, lambda_val: float) -> Tensor:
    """utils/metrics_lightgcn.py:9-45 (pinned by tests/golden/bpr_loss.pt)."""
    reg_loss = lambda_val * (users_emb_0.norm(2).pow(2) + pos_items_emb_0.norm(2).pow(2)
                             + neg_items_emb_0.norm(2).pow(2))
    pos_scores = t.sum(t.mul(users_emb_final, pos_items_emb_final), dim=-1)
    neg_scores = t.sum(t.mul(users_emb_final, neg_items_emb_final), dim=-1)
    return -t.mean(t.nn.functional.softplus(pos_scores - neg_scores)) + reg_loss


def make_predictions_for_user(user_embeddings: Tensor, article_embeddings: Tensor, user_id: int,
                              positive_items_for_user: dict, num_recommendations: int) -> Tensor:
    """utils/metrics_lightgcn.py:125-142 (pinned by tests/golden/topk.pt)."""
    ignore = positive_items_for_user[user_id] if user_id in positive_items_for_user else t.tensor([])
    scores = user_embeddings[user_id] @ article_embeddings.T
    _, indices = t.topk(scores, k=num_recommendations + len(ignore))
    diff = np.setdiff1d(indices.detach().numpy(), ignore.detach().numpy(), assume_unique=True)  # utils/tensor.py:16-21
    return t.tensor(diff)[:num_recommendations]


def scores_fma(user_rows: Tensor, item_emb: Tensor) -> Tensor:
    """scores as a k-ordered fp32 fma chain (oracle/spmm_ref.c) — the exact arithmetic of the HIP top-K."""
    L = clib()
    u = user_rows.to(t.float32).contiguous()
    it = item_emb.to(t.float32).contiguous()
    out = t.empty(u.shape[0], it.shape[0], dtype=t.float32)
    L.ref_scores_fma_f32(u.shape[0], it.shape[0], u.shape[1], u.data_ptr(), u.shape[1], it.data_ptr(), it.shape[1],
                         out.data_ptr(), it.shape[0])
    return out


def gemm_fma(A: Tensor, B: Tensor, *, trans_a=False, trans_b=True, bias=None, out=None, accumulate=False,
             relu=False) -> Tensor:
    """C = act(op(A) @ op(B) + bias (+C)) as a k-ordered fp32 fma chain (oracle/spmm_ref.c) — bitwise the
    arithmetic of csrc/gemm.hip.  Same argument meaning as laplace_amd.ops.gemm."""
    A, B = A.contiguous(), B.contiguous()
    m, k = (A.shape[1], A.shape[0]) if trans_a else A.shape
    n = B.shape[0] if trans_b else B.shape[1]
    sa = (1, A.shape[1]) if trans_a else (A.shape[1], 1)
    sb = (B.shape[1], 1) if trans_b else (1, B.shape[1])
    C = out if out is not None else t.zeros(m, n, dtype=t.float32)
    b = bias.contiguous() if bias is not None else None
    clib().ref_gemm_fma_f32(m, n, k, A.data_ptr(), sa[0], sa[1], B.data_ptr(), sb[0], sb[1],
                            b.data_ptr() if b is not None else None, C.data_ptr(), C.stride(0),
                            1 if accumulate else 0, 1 if relu else 0)
    return C


def topk_excl_exact(scores: Tensor, excl: List[Tensor], k: int) -> Tensor:
    """Top-k item ids per row by (score desc, id asc) among non-excluded ids; -1 pads."""
    n_q, n_items = scores.shape
    out = t.full((n_q, k), -1, dtype=t.int64)
    ids = np.arange(n_items)
    for q in range(n_q):
        s = scores[q].numpy().copy()
        mask = np.ones(n_items, dtype=bool)
        if len(excl[q]):
            mask[excl[q].numpy()] = False
        cand = ids[mask]
        order = np.lexsort((cand, -s[cand].astype(np.float64)))  # primary: -score, secondary: id
        top = cand[order[:k]]
        out[q, :len(top)] = t.from_numpy(top)
    return out


# ----------------------------------------------------------------------------------------------
# sampler: data/lightgcn_loader.py:95-112 + PyG structured_negative_sampling
# ----------------------------------------------------------------------------------------------
def structured_negative_sampling(edge_index: Tensor, num_nodes: int, rng: np.random.Generator,
                                 contains_neg_self_loops: bool = True) -> Tensor:
    """torch_geometric.utils.structured_negative_sampling (PyG 2.0.4), CPU:
    rand ~ U[0, num_nodes) per edge, redrawn while row*num_nodes+rand is in {row*num_nodes+col}; with
    contains_neg_self_loops=False the keys i*num_nodes+i of all i in range(num_nodes) join that set.
    Called with num_nodes = max(edge_index[1]) at data/lightgcn_loader.py:105-106 (default True) and at
    run_pipeline_lightgcn.py:40-44 (False)."""
    row, col = edge_index[0].numpy(), edge_index[1].numpy()
    pos_idx = row * num_nodes + col
    if not contains_neg_self_loops:
        loop = np.arange(num_nodes, dtype=pos_idx.dtype)
        pos_idx = np.concatenate([pos_idx, loop * num_nodes + loop])
    rand = rng.integers(0, num_nodes, size=row.shape[0])
    neg_idx = row * num_nodes + rand
    mask = np.isin(neg_idx, pos_idx)
    rest = np.nonzero(mask)[0]
    while rest.size > 0:
        tmp = rng.integers(0, num_nodes, size=rest.size)
        rand[rest] = tmp
        neg_idx = row[rest] * num_nodes + tmp
        mask = np.isin(neg_idx, pos_idx)
        rest = rest[mask]
    return t.from_numpy(rand)


def sample_mini_batch(batch_size: int, edge_index: Tensor, rng: np.random.Generator, pyrng: random.Random):
    """data/lightgcn_loader.py:95-112 with explicit RNGs."""
    num_nodes = int(t.max(edge_index[1]))
    neg = structured_negative_sampling(edge_index, num_nodes, rng)
    edges = t.stack([edge_index[0], edge_index[1], neg], dim=0)
    indices = pyrng.choices(range(edges.shape[1]), k=batch_size)
    batch = edges[:, indices]
    return batch[0], batch[1], batch[2]


_TAG_EDGE, _TAG_NEG = 0x45444745, 0x4E454721
_MAX_NEG_ATTEMPTS = 4096


def sample_bpr_batch_philox(rowptr: Tensor, col: Tensor, batch: int, neg_range: int, seed: int, step: int,
                            quirk: bool = False, edges_in_order: bool = False, no_self_loops: bool = False):
    """Bit-exact restatement of csrc/train.hip:sample_bpr_kernel (integer work => exact parity).
    rowptr/col: the users x items interaction CSR with sorted columns."""
    rp, c = rowptr.numpy().astype(np.int64), col.numpy().astype(np.int64)
    nnz = int(c.shape[0])
    n_rows = rp.shape[0] - 1
    row_of_edge = np.repeat(np.arange(n_rows), np.diff(rp))
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    s0, s1 = step & 0xFFFFFFFF, (step >> 32) & 0xFFFFFFFF
    b = np.arange(batch, dtype=np.uint64)
    r = philox4x32(b & np.uint64(0xFFFFFFFF), b >> np.uint64(32), s0, s1 ^ _TAG_EDGE, k0, k1)
    e = ((r[0] << np.uint64(32)) | r[1]) % np.uint64(nnz)
    e = np.arange(batch, dtype=np.int64) if edges_in_order else e.astype(np.int64)
    users = row_of_edge[e]
    pos = c[e]
    neg = np.zeros(batch, dtype=np.int64)
    neigh = [set(c[rp[u]:rp[u + 1]].tolist()) for u in range(n_rows)] if n_rows <= 200000 else None

    def has(u, key):
        if neigh is not None:
            return key in neigh[u]
        seg = c[rp[u]:rp[u + 1]]
        j = np.searchsorted(seg, key)
        return j < seg.shape[0] and seg[j] == key

    for i in range(batch):
        u, ei = int(users[i]), int(e[i])
        cand = 0
        for tt in range(_MAX_NEG_ATTEMPTS):
            q = philox4x32(ei & 0xFFFFFFFF, tt, s0, s1 ^ _TAG_NEG, k0, k1)
            cand = int(((int(q[0]) << 32) | int(q[1])) % neg_range)
            hit = has(u, cand)
            if (not hit) and quirk and cand == 0 and u > 0:
                hit = has(u - 1, neg_range)
            if (not hit) and no_self_loops and cand == u:
                hit = True
            if not hit:
                break
        neg[i] = cand
    return t.from_numpy(users.astype(np.int64)), t.from_numpy(pos.astype(np.int64)), t.from_numpy(neg)


# ----------------------------------------------------------------------------------------------
# one training step: run_pipeline_lightgcn.py:117-159
# ----------------------------------------------------------------------------------------------
def train_step(users_w: Tensor, items_w: Tensor, optimizer: t.optim.Optimizer, row: Tensor, col: Tensor,
               num_iterations: int, batch: Tuple[Tensor, Tensor, Tensor], lambda_val: float) -> float:
    """forward -> gathers -> bpr_loss -> zero_grad/backward/step, as run_pipeline_lightgcn.py:120-159.
    users_w / items_w are leaf Parameters owned by `optimizer` (torch.optim.Adam in the reference)."""
    uf, u0, itf, it0 = lightgcn_forward(users_w, items_w, row, col, num_iterations)
    ui, pi, ni = batch
    loss = bpr_loss(uf[ui], u0[ui], itf[pi], it0[pi], itf[ni], it0[ni], lambda_val)
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return float(loss.item())


# ----------------------------------------------------------------------------------------------
# CPU port of the model-only training step, used by bench.py's cpu_baseline leg
# ----------------------------------------------------------------------------------------------
class CpuTrainPort:
    """forward + bpr + backward + Adam of run_pipeline_lightgcn.py:117-159 on the host cores, with the
    batch given (negatives pre-drawn: the 'model-only step' of BASELINE.md §2).  Propagates with the
    C restatement of torch_sparse's spmm_cpu (OpenMP over rows = at::parallel_for), everything else
    is plain torch CPU ops / the C Adam.  Symmetric adjacency assumed (A^T = A) unless adj_t given."""

    def __init__(self, table: Tensor, rowptr: Tensor, col: Tensor, val: Tensor, num_users: int, K: int,
                 lr: float, Lambda: float, adj_t: Optional[Tuple[Tensor, Tensor, Tensor]] = None):
        self.table = table.clone().contiguous()
        self.a = (rowptr.to(t.int32).contiguous(), col.to(t.int32).contiguous(), val.contiguous())
        self.at = self.a if adj_t is None else tuple(x.contiguous() for x in adj_t)
        self.U, self.K, self.lr, self.Lambda = num_users, K, lr, Lambda
        self.m = t.zeros_like(self.table)
        self.v = t.zeros_like(self.table)
        self.steps = 0

    def forward(self) -> Tensor:
        x = self.table
        s = self.table.clone()
        for _ in range(self.K):
            x = spmm_c(*self.a, x)
            s += x
        return s.mul_(1.0 / (self.K + 1))

    def step(self, batch: Tuple[Tensor, Tensor, Tensor]) -> float:
        U, K = self.U, self.K
        final = self.forward()
        ui, pi, ni = batch
        rows = [final[ui], self.table[ui], final[U + pi], self.table[U + pi], final[U + ni], self.table[U + ni]]
        rows = [r.detach().requires_grad_(True) for r in rows]
        loss = bpr_loss(*rows, self.Lambda)
        grads = t.autograd.grad(loss, rows)
        G = t.zeros_like(self.table)
        G0 = t.zeros_like(self.table)
        for idx, gf, g0 in ((ui, grads[0], grads[1]), (U + pi, grads[2], grads[3]), (U + ni, grads[4], grads[5])):
            G.index_add_(0, idx, gf)
            G0.index_add_(0, idx, g0)
        gc = G.mul_(1.0 / (K + 1))
        g = gc
        for _ in range(K):
            g = spmm_c(*self.at, g).add_(gc)
        g = g + G0 if K > 0 else gc + G0
        self.steps += 1
        b1, b2, eps = 0.9, 0.999, 1e-8
        bc1, bc2 = 1.0 - b1 ** self.steps, 1.0 - b2 ** self.steps
        clib().ref_adam_f32(self.table.numel(), self.table.data_ptr(), g.data_ptr(), self.m.data_ptr(),
                            self.v.data_ptr(), b1, b2, self.lr / bc1, bc2 ** 0.5, eps)
        return float(loss)

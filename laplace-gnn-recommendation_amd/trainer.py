"""Fused LightGCN training step: one pass of run_pipeline_lightgcn.py:117-159 with every stage on
the GPU and no host round trip.

    forward   K fused SpMM launches (layer sum in the epilogue)      model/lightgcn.py:46-80
    sample    B edges + structured negatives on device               data/lightgcn_loader.py:95-112
    loss      gathers + BPR forward/backward in one kernel           run_pipeline_lightgcn.py:133-155
    backward  K SpMM launches on A^T, G/(K+1) folded in as addend    (autograd of the forward)
    update    dense Adam over the whole table, L2 term folded in     run_pipeline_lightgcn.py:157-159

Same arithmetic as the autograd path (LightGCN.forward + bpr_loss + backward + Adam); what is
removed is memory traffic: no cat, no [N,K+1,D] stack, no saved activations (the propagate is
linear), no materialised per-batch gathers, gradient and L2 term consumed in the Adam pass.

HBM-resident state (fp32, N = U+I rows of D): table, final, two ping-pong buffers, Gc, m, v.

`sparse_batch=True` (default) additionally uses what is known about the operands of one step — the
loss reads `final` only at the <= 3B batch rows, and its gradient is non-zero only there — to move
fewer bytes for the SAME result (to rounding: the layer sum is associated differently; each form is bitwise
reproducible run to run — no float atomics anywhere in the step):
  * the batch is drawn first (it does not depend on the embeddings); its unique node set and the
    node -> slot map are built on device (mi_batch_nodes_i32), no host read-back;
  * forward: layers 1..K-1 are plain products; the running layer sum is kept only at the batch rows
    (mi_gather_rows_f32); layer K is computed only at the batch rows (row_list) — hub rows still go
    through the split path;
  * the BPR kernel reads / writes compact [<= 3B, D] tables through the node map, so the dense Gc
    buffer, its zero-fill and its K reads as an addend disappear;
  * backward layer 1 gathers only the non-zero rows of its input (x_map); all layers take the compact
    gradient as addend (addend_map);
  * the last backward product IS the parameter gradient: with fuse_adam (default) the Adam update runs in
    that product's epilogue (mi_adam_args), so the gradient is never written or read back (-1.1 GB at C2).
`sparse_batch=False` is the straightforward form (full `final`, dense Gc) kept for A/B and for callers
that want the full forward output of the step.

`reorder` (default: on for adjacencies of >= 1M entries) trains under a LOCALITY ORDER of the nodes
(interactions.LocalityOrder: items by popularity, users by their coldest item): the adjacency, the interaction
CSR and — in place — the model's table are relabelled once at construction, every kernel then runs on the new ids,
and `to_original_order()` / `to_training_order()` put the table's rows back / forth (one gather of the table each;
`finish()` = to_original_order).  Batches passed to `step()` and returned by `sample()` are in ORIGINAL ids.  The
result is the same training run up to the summation order inside a row (entries are summed in column order, and
columns are renamed): equal to rounding, and still bitwise reproducible run to run.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch as t
from torch import Tensor

from . import ops
from .interactions import Interactions
from .model.lightgcn import LightGCN, propagate_mean, propagate_mean_backward
from .sparse import SparseTensor


def _lib_bytes_batch_nodes(n_nodes: int) -> int:
    from . import _lib
    return int(_lib.lib().mi_batch_nodes_workspace_bytes(n_nodes))


class LightGCNTrainer:
    def __init__(self, model: LightGCN, adj: SparseTensor, train: Interactions, *, lr: float, Lambda: float,
                 batch_size: int, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8, seed: int = 0,
                 neg_range: Optional[int] = None, reference_sampler_quirks: bool = False,
                 sparse_batch: bool = True, fuse_adam: bool = True, reorder: Optional[bool] = None):
        self.model = model
        self.table = model.table()
        if not self.table.is_cuda:
            raise RuntimeError("LightGCNTrainer needs the model on the GPU (model.to('cuda')); no CPU fallback")
        self.K = model.num_iterations
        self.lr, self.Lambda, self.batch_size = float(lr), float(Lambda), int(batch_size)
        self.betas, self.eps, self.seed = betas, float(eps), int(seed)
        # reference: num_nodes = max(train item id)  => negatives in [0, max_item_id)  (Appendix A.3)
        self.neg_range = int(neg_range) if neg_range is not None else train.num_items
        self.quirk = bool(reference_sampler_quirks)
        # a negative range that is not "every item" names items by id: it does not survive a relabelling
        can_reorder = self.neg_range == train.num_items and not self.quirk
        if reorder is None:
            reorder = can_reorder and adj.nnz() >= ops.PLAN_MIN_NNZ
        elif reorder and not can_reorder:
            raise ValueError("reorder=True needs neg_range == num_items and no reference sampler quirks")
        self.order = None
        self.in_training_order = True
        if reorder:
            self.order = train.locality_order()
            self._new_of_old = self.order.node_new_of_old()
            self._old_of_new = self.order.node_old_of_new()
            train = train.permuted(self.order)
            adj = adj.permuted(self._new_of_old)
            self.in_training_order = False
            self.to_training_order()
        self.adj_fwd, self.adj_bwd = adj.gcn_normalized(model.add_self_loops)
        if self.order is not None:
            # items are numbered by popularity: their first rows take half of all user-row gathers — dense launches keep
            # them in LDS (ops.spmm, `hot`; a speed hint, results unchanged)
            self.adj_fwd.hot = self.adj_bwd.hot = ops.hot_item_rows(model.num_users, model.num_items)
        self.train = train
        n, d = self.table.shape
        dev = self.table.device
        self.sparse_batch = bool(sparse_batch)
        self.fuse_adam = bool(fuse_adam)
        self.final = t.empty(n, d, device=dev)
        self.buf_a = t.empty(n, d, device=dev)
        self.buf_b = t.empty(n, d, device=dev)
        nb = 3 * self.batch_size
        if self.sparse_batch:  # compact per-batch tables instead of the dense gradient buffer
            self.gc = None
            self.gmap = t.empty(n, dtype=t.int32, device=dev)
            self.nodes = t.zeros(nb, dtype=t.int32, device=dev)
            self.n_nodes = t.zeros(2, dtype=t.int32, device=dev)
            self.sum_c = t.empty(nb, d, device=dev)
            self.final_c = t.empty(nb, d, device=dev)
            self.gc_c = t.zeros(nb, d, device=dev)
            self._nodes_ws = t.empty(_lib_bytes_batch_nodes(n), dtype=t.uint8, device=dev)
        else:
            self.gc = t.zeros(n, d, device=dev)
        self.reg_w = t.zeros(n, device=dev)
        self.m = t.zeros(n, d, device=dev)
        self.v = t.zeros(n, d, device=dev)
        self.loss = t.zeros(1, device=dev)
        self.batch_idx = tuple(t.empty(self.batch_size, dtype=t.int64, device=dev) for _ in range(3))
        self.step_count = 0
        self._r = train.csr()
        self._row_of_edge = train.row_of_edge()

    # -- locality order --------------------------------------------------------------------------
    def to_training_order(self) -> None:
        """Rows of the model's table into the order the kernels train in (no-op without reorder)."""
        if self.order is not None and not self.in_training_order:
            self.table.copy_(self.table[self._old_of_new])  # new row r = old row old_of_new[r]
            self.in_training_order = True

    def to_original_order(self) -> None:
        """Rows of the model's table back under their original ids (evaluation, top-K, saving)."""
        if self.order is not None and self.in_training_order:
            self.table.copy_(self.table[self._new_of_old])
            self.in_training_order = False

    finish = to_original_order

    def _ids_to_training(self, batch):
        if self.order is None:
            return batch
        users, pos, neg = batch
        return self.order.user_new_of_old[users], self.order.item_new_of_old[pos], self.order.item_new_of_old[neg]

    # -- pieces (also used one by one in tests) ------------------------------------------------
    def forward(self) -> Tensor:
        """mean_k(A^k E0) over the whole table; rows in TRAINING order (index with order.node_new_of_old())."""
        self.to_training_order()
        return propagate_mean(self.adj_fwd, self.table, self.K, out=self.final, scratch=(self.buf_a, self.buf_b))

    def _sample(self) -> Tuple[Tensor, Tensor, Tensor]:
        return ops.sample_bpr_batch(self._r, self._row_of_edge, self.batch_size, self.neg_range, self.seed,
                                    self.step_count, quirk=self.quirk, out=self.batch_idx)

    def sample(self) -> Tuple[Tensor, Tensor, Tensor]:
        """(users, pos, neg) of the next step's batch, ORIGINAL ids."""
        users, pos, neg = self._sample()
        if self.order is None:
            return users, pos, neg
        return self.order.user_old_of_new[users], self.order.item_old_of_new[pos], self.order.item_old_of_new[neg]

    def step(self, batch: Optional[Tuple[Tensor, Tensor, Tensor]] = None) -> Tensor:
        """One training iteration; returns the loss as a 1-element device tensor (no sync).  batch: original ids."""
        self.to_training_order()
        batch = self._ids_to_training(batch) if batch is not None else None
        if self.sparse_batch:
            return self._step_sparse(batch)
        K = self.K
        self.forward()
        users, pos, neg = batch if batch is not None else self._sample()
        self.gc.zero_()
        self.reg_w.zero_()
        ops.bpr_fwd_bwd(users, pos, neg, self.final, self.table, self.model.num_users, self.Lambda,
                        g_final=self.gc, reg_w=self.reg_w, g_scale=1.0 / (K + 1), loss_out=self.loss)
        g0 = propagate_mean_backward(self.adj_bwd, self.gc, K, scratch=(self.buf_a, self.buf_b), pre_scaled=True)
        self.step_count += 1
        ops.adam_step(self.table, g0, self.m, self.v, step=self.step_count, lr=self.lr, beta1=self.betas[0],
                      beta2=self.betas[1], eps=self.eps, reg_w=self.reg_w)
        return self.loss

    def _step_sparse(self, batch: Optional[Tuple[Tensor, Tensor, Tensor]]) -> Tensor:
        K, tab, adj, adj_t = self.K, self.table, self.adj_fwd, self.adj_bwd
        users, pos, neg = batch if batch is not None else self._sample()
        if users.numel() != self.batch_size:
            raise ValueError("batch size differs from the trainer's")
        gmap, nodes, cnt2 = ops.batch_nodes(users, pos, neg, self.model.num_users, tab.shape[0], gmap=self.gmap,
                                            nodes=self.nodes, count=self.n_nodes, ws=self._nodes_ws)
        cnt = cnt2[:1]
        # ---- forward: final only at the batch rows
        c = 1.0 / (K + 1)
        ops.gather_rows(self.sum_c, tab, nodes, cnt)                      # S_c = E0[batch rows]
        x = tab
        bufs = (self.buf_a, self.buf_b)
        for k in range(1, K):                                             # layers 1..K-1: plain products
            y = bufs[(k - 1) % 2]
            ops.spmm(adj, x, Y=y)
            ops.gather_rows(self.sum_c, y, nodes, cnt, accumulate=True)   # S_c += X_k[batch rows]
            x = y
        if K >= 1:                                                        # layer K at the batch rows only
            ops.spmm(adj, x, addend=self.sum_c, S=self.final_c, scale=c, row_list=nodes, n_list_dev=cnt)
            final_c = self.final_c
        else:
            final_c = self.sum_c
        # ---- loss + its gradient on the compact tables
        self.gc_c.zero_()
        self.reg_w.zero_()
        ops.bpr_fwd_bwd(users, pos, neg, final_c, tab, self.model.num_users, self.Lambda, g_final=self.gc_c,
                        reg_w=self.reg_w, g_scale=c, loss_out=self.loss, node_map=gmap)
        # ---- backward: g_K = Gc (compact); g_k = Gc + A^T g_{k+1}
        if K == 0:
            g0 = self.buf_a
            g0.zero_()
            g0.index_add_(0, nodes[: int(cnt)].long(), self.gc_c[: int(cnt)])  # K = 0 is not a hot path
        else:
            cur = None
            for i in range(K):
                nxt = bufs[i % 2]
                # the last product's output is the parameter gradient: Adam consumes it in the epilogue
                opt = None
                if i == K - 1 and self.fuse_adam:
                    opt = dict(p=tab, m=self.m, v=self.v, step=self.step_count + 1, lr=self.lr, beta1=self.betas[0],
                               beta2=self.betas[1], eps=self.eps, reg_w=self.reg_w)
                out = None if opt is not None else nxt
                if i == 0:   # input = the compact gradient itself: skip its all-zero rows
                    # (the split rows are the hub ITEMS': their columns are users, of which the batch names few)
                    ops.spmm(adj_t, self.gc_c, addend=self.gc_c, S=out, x_map=gmap, addend_map=gmap, adam=opt,
                             x_rare=8 * self.batch_size <= self.model.num_users)
                else:
                    ops.spmm(adj_t, cur, addend=self.gc_c, S=out, addend_map=gmap, adam=opt)
                cur = nxt
            g0 = cur
        self.step_count += 1
        if K == 0 or not self.fuse_adam:
            ops.adam_step(tab, g0, self.m, self.v, step=self.step_count, lr=self.lr, beta1=self.betas[0],
                          beta2=self.betas[1], eps=self.eps, reg_w=self.reg_w)
        return self.loss

    def decay_lr(self, gamma: float = 0.95) -> None:
        """ExponentialLR(gamma).step() (run_pipeline_lightgcn.py:104,178-179)."""
        self.lr *= gamma

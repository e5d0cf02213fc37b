"""User-sharded multi-GPU LightGCN training — one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

Partition: rank p owns a contiguous block of users, every edge incident to them (both directions
of the bipartite CSR), their embedding rows and Adam state.  The item table (I x D, 51 MB at
I=100K, D=128) is replicated.  The reference has no distributed code at all; this
is new design, not a port.

Per propagate layer, on rank p (local node space: U_p user rows, then I item rows):
    1. item rows:  partial[i] = sum_{u in U_p} a(i,u) x[u]      local SpMM over rows [U_p, U_p+I)
    2. all-reduce(sum, fp32) of the I x D partials                 the ONE exchange of the layer
    3. user rows:  y[u] = sum_i a(u,i) x[i]  (+ epilogue)          local SpMM, overlaps with 2
(step 1 of the NEXT layer reads only user rows, so in the default step form exchange 2 is waited for after it:
each all-reduce has a user-row and an item-row product to hide behind)
Backward mirrors it (the normalised bipartite adjacency is symmetric).  The BPR gradient of the
item rows and the L2 weights are all-reduced once per step; every rank then applies the same Adam
update to its item replica, so replicas stay bitwise identical.  User rows never leave their rank.

Edge weights a(u,i) = (deg(u) * deg(i))^-1/2 need the GLOBAL item degree: one all-reduce of the
int degree vector at set-up.

xGMI is point-to-point, 7 links per GPU: the payload per collective is small (I*D*4 bytes), so
what matters is that the collective runs beside the user-row SpMM of the same layer (independent
work) rather than peak bus bandwidth; RCCL picks the algorithm.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch as t
import torch.distributed as dist
from torch import Tensor

from . import ops as hip_ops
from .interactions import Interactions
from .model.lightgcn import LightGCN


REORDER_MIN_EDGES_PER_RANK = 1 << 20   # default locality order: on when the JOB holds this many edges per rank


class ShardedLightGCNTrainer:
    def __init__(self, model: LightGCN, train: Interactions, *, lr: float, Lambda: float, batch_size: int,
                 betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8, seed: int = 0,
                 neg_range: Optional[int] = None, group=None, ops_impl=None, sparse_batch: bool = True,
                 reorder: Optional[bool] = None):
        """model: LightGCN(num_users = this rank's users, num_items = all items).  `train` holds this
        rank's edges with LOCAL user ids.  ops_impl: the kernel provider (default: the HIP ops;
        the CPU gloo tests inject an oracle-backed one — the product never does).
        reorder (default: on from 1M edges per rank, decided on the all-reduced total): train under the locality order of trainer.LightGCNTrainer — the
        replicated items are ranked by their GLOBAL degree, so every rank numbers them alike; users are local anyway."""
        self.ops = ops_impl if ops_impl is not None else hip_ops
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.model, self.train = model, train
        self.table = model.table()
        self.U, self.I = model.num_users, model.num_items
        self.K = model.num_iterations
        self.lr, self.Lambda, self.batch_size = float(lr), float(Lambda), int(batch_size)
        self.betas, self.eps, self.seed = betas, float(eps), int(seed)
        self.neg_range = int(neg_range) if neg_range is not None else self.I
        n, d = self.table.shape
        dev = self.table.device
        U, I = self.U, self.I

        # item replicas start identical: rank 0's rows win
        self._bcast(self.table[U:])

        # The decision is COLLECTIVE: a reordering rank all-reduces the item degrees and renumbers the replicated item
        # rows, so every rank must take the same branch (a rank-local threshold on uneven shards would mismatch the
        # collective sequence and sum rows of different items).  Default: on when the job's edges / world reach
        # REORDER_MIN_EDGES_PER_RANK.
        can_reorder = self.neg_range == I
        if reorder and not can_reorder:
            raise ValueError("reorder=True needs neg_range == num_items")
        votes = t.tensor([train.num_edges, -1 if reorder is None else int(bool(reorder)), int(can_reorder)],
                         dtype=t.int64, device=dev)
        lo, hi = votes.clone(), votes.clone()
        if self.world > 1:
            self._allreduce(votes)                                            # total edges (and sums of the flags)
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        if int(lo[1]) != int(hi[1]):
            raise ValueError("ShardedLightGCNTrainer: the `reorder` argument differs across ranks")
        if reorder is None:
            reorder = bool(int(lo[2])) and int(votes[0]) >= REORDER_MIN_EDGES_PER_RANK * self.world
        self.order, self.in_training_order = None, True
        if reorder:
            gdeg = t.bincount(train.edge_index[1], minlength=I)
            self._allreduce(gdeg)                       # the same ranking of the items on every rank
            self.order = train.locality_order(item_degree=gdeg)
            self._new_of_old, self._old_of_new = self.order.node_new_of_old(), self.order.node_old_of_new()
            train = train.permuted(self.order)
            self.train = train
            self.in_training_order = False
            self.to_training_order()

        # local symmetric bipartite CSR, globally-normalised edge weights
        u, i = train.edge_index[0], train.edge_index[1]
        iu = i + U
        raw = self.ops.coo_to_csr(t.cat([u, iu]).contiguous(), t.cat([iu, u]).contiguous(), n, n, want_perm=False)
        deg = (raw.rowptr[1:] - raw.rowptr[:-1]).to(t.int64)
        item_deg = deg[U:].clone()
        self._allreduce(item_deg)
        deg = t.cat([deg[:U], item_deg]).to(t.float32)
        dis = deg.pow(-0.5)
        dis.masked_fill_(dis == float("inf"), 0.0)
        raw.val = self.ops.scale_csr(raw, None, dis, dis)
        self.adj_fwd = raw  # kept whole for bookkeeping (bench reads nnz / n_rows)
        self.a_users = self.ops.row_slice(raw, 0, U)
        self.a_items = self.ops.row_slice(raw, U, n)
        if self.order is not None and self.ops is hip_ops:
            self.a_users.hot = hip_ops.hot_item_rows(U, I)   # popular items first: LDS hot-row cache of the user-row products
        self.a_users.plan = self.ops.build_spmm_plan(self.a_users)
        self.a_items.plan = self.ops.build_spmm_plan(self.a_items)

        self.sparse_batch = bool(sparse_batch)
        self.final = t.empty(n, d, device=dev)
        self.bufs = (t.empty(n, d, device=dev), t.empty(n, d, device=dev))
        if self.sparse_batch:  # compact per-batch tables (see trainer.py) + dense item-side exchange buffers
            nb = 3 * self.batch_size
            self.gc = None
            self.gmap = t.empty(n, dtype=t.int32, device=dev)
            self.nodes = t.zeros(nb, dtype=t.int32, device=dev)
            self.n_nodes = t.zeros(2, dtype=t.int32, device=dev)
            self.sum_c = t.empty(nb, d, device=dev)
            self.gc_c = t.zeros(nb, d, device=dev)
            self.items_y = t.empty(I, d, device=dev)
            self.items_g = t.zeros(I, d, device=dev)
            self.imap = t.full((n,), -1, dtype=t.int32, device=dev)
            self._item_ids = t.arange(I, dtype=t.int32, device=dev)
        else:
            self.gc = t.zeros(n, d, device=dev)
        self.reg_w = t.zeros(n, device=dev)
        self.m = t.zeros(n, d, device=dev)
        self.v = t.zeros(n, d, device=dev)
        self.loss = t.zeros(1, device=dev)
        self.batch_idx = tuple(t.empty(self.batch_size, dtype=t.int64, device=dev) for _ in range(3))
        self.step_count = 0
        self._r = train.csr() if ops_impl is None else self.ops.coo_to_csr(u.contiguous(), i.contiguous(), U, I,
                                                                             want_perm=False)
        self._row_of_edge = self.ops.expand_rows(self._r)

    # -- locality order (see trainer.LightGCNTrainer) -------------------------------------------------
    def to_training_order(self) -> None:
        if self.order is not None and not self.in_training_order:
            self.table.copy_(self.table[self._old_of_new])
            self.in_training_order = True

    def to_original_order(self) -> None:
        if self.order is not None and self.in_training_order:
            self.table.copy_(self.table[self._new_of_old])
            self.in_training_order = False

    finish = to_original_order

    def _ids_to_training(self, batch):
        if self.order is None or batch is None:
            return batch
        users, pos, neg = batch
        return self.order.user_new_of_old[users], self.order.item_new_of_old[pos], self.order.item_new_of_old[neg]

    # -- collectives ---------------------------------------------------------------------------
    def _allreduce(self, x: Tensor, async_op: bool = False):
        if self.world == 1:
            return None
        return dist.all_reduce(x, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def _bcast(self, x: Tensor) -> None:
        if self.world > 1:
            dist.broadcast(x, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0,
                           group=self.group)

    # -- one propagate layer: out = A @ x, the item rows all-reduced -------------------------------
    def _layer(self, x: Tensor, out: Optional[Tensor], *, addend_users: Optional[Tensor], s_users: Optional[Tensor],
               scale: float, items_out: Tensor) -> None:
        """items_out[I, D] <- all-reduced item rows of A @ x; user rows go to out[:U] (if out) and to
        s_users = scale * (addend_users + acc) (if s_users)."""
        U = self.U
        self.ops.spmm(self.a_items, x, Y=items_out)
        work = self._allreduce(items_out, async_op=True)
        self.ops.spmm(self.a_users, x, Y=None if out is None else out[:U], addend=addend_users, S=s_users,
                      scale=scale)
        if work is not None:
            work.wait()

    def forward(self) -> Tensor:
        """final = mean_k(A^k E0) on the local node space (model/lightgcn.py:46-80, sharded); rows in training order."""
        self.to_training_order()
        U, K = self.U, self.K
        S, tab = self.final, self.table
        if K == 0:
            S.copy_(tab)
            return S
        c = 1.0 / (K + 1)
        x = tab
        for k in range(1, K + 1):
            last = k == K
            out = self.bufs[k % 2]
            self._layer(x, None if last else out, addend_users=(tab[:U] if k == 1 else S[:U]), s_users=S[:U],
                        scale=c if last else 1.0, items_out=out[U:])
            if k == 1:
                t.add(tab[U:], out[U:], out=S[U:])
            else:
                S[U:].add_(out[U:])
            if last:
                S[U:].mul_(c)
            x = out
        return S

    def _sample(self):
        return self.ops.sample_bpr_batch(self._r, self._row_of_edge, self.batch_size, self.neg_range, self.seed,
                                         self.step_count, out=self.batch_idx)

    def sample(self):
        """(users, pos, neg) of the next step's batch, ORIGINAL (local) ids."""
        users, pos, neg = self._sample()
        if self.order is None:
            return users, pos, neg
        return self.order.user_old_of_new[users], self.order.item_old_of_new[pos], self.order.item_old_of_new[neg]

    def step(self, batch=None) -> Tensor:
        """One training iteration over the GLOBAL batch (world * batch_size positive edges); returns this
        rank's local loss term as a 1-element tensor."""
        self.to_training_order()
        batch = self._ids_to_training(batch)
        if self.sparse_batch and self.K >= 1:
            return self._step_sparse(batch)
        if self.gc is None:
            self.gc = t.zeros_like(self.table)
        U, K, P = self.U, self.K, self.world
        self.forward()
        users, pos, neg = batch if batch is not None else self._sample()
        self.gc.zero_()
        self.reg_w.zero_()
        # global loss = mean over P*B slots => each rank's share of the softplus term carries 1/P
        self.ops.bpr_fwd_bwd(users, pos, neg, self.final, self.table, U, self.Lambda, g_final=self.gc,
                             reg_w=self.reg_w, g_scale=1.0 / ((K + 1) * P), loss_out=self.loss)
        self._allreduce(self.gc[U:])
        self._allreduce(self.reg_w[U:])
        g = self.gc
        for i in range(K):
            nxt = self.bufs[i % 2]
            self._layer(g, None, addend_users=self.gc[:U], s_users=nxt[:U], scale=1.0, items_out=nxt[U:])
            nxt[U:].add_(self.gc[U:])
            g = nxt
        self.step_count += 1
        self.ops.adam_step(self.table, g, self.m, self.v, step=self.step_count, lr=self.lr, beta1=self.betas[0],
                           beta2=self.betas[1], eps=self.eps, reg_w=self.reg_w)
        return self.loss

    def _step_sparse(self, batch) -> Tensor:
        """The byte-saving form of trainer.LightGCNTrainer._step_sparse, sharded.  User rows get the
        row_list / x_map / compact-addend treatment; item rows stay dense because their partial sums
        are exchanged: forward layer K still needs every rank's users, and the item gradient is the
        sum of every rank's batch."""
        U, I, K, P, ops = self.U, self.I, self.K, self.world, self.ops
        tab, c = self.table, 1.0 / (self.K + 1)
        users, pos, neg = batch if batch is not None else self._sample()
        gmap, nodes, cnt2 = ops.batch_nodes(users, pos, neg, U, tab.shape[0], gmap=self.gmap, nodes=self.nodes,
                                            count=self.n_nodes)
        cnt, cnt_u = cnt2[0:1], cnt2[1:2]
        # ---- forward: final only at this rank's batch rows, kept in sum_c
        # The item-row product of a layer reads only the USER rows of its input, so it does not need the previous
        # layer's exchange: each all-reduce is waited for one product later than it is issued and runs beside the
        # user-row product of its own layer AND the item-row product of the next.
        ops.gather_rows(self.sum_c, tab, nodes, cnt)
        x, pending = tab, None   # pending: the exchange whose result is x[U:]
        for k in range(1, K):
            y = self.bufs[(k - 1) % 2]
            ops.spmm(self.a_items, x, Y=y[U:])
            if pending is not None:
                pending.wait()
            if k > 1:
                ops.gather_rows(self.sum_c, x, nodes, cnt, accumulate=True)   # layer k-1, now complete
            work = self._allreduce(y[U:], async_op=True)
            ops.spmm(self.a_users, x, Y=y[:U])
            x, pending = y, work
        ops.spmm(self.a_items, x, Y=self.items_y)                      # layer K, item rows: every rank contributes
        if pending is not None:
            pending.wait()
        if K > 1:
            ops.gather_rows(self.sum_c, x, nodes, cnt, accumulate=True)
        work = self._allreduce(self.items_y, async_op=True)
        ops.spmm(self.a_users, x, addend=self.sum_c, S=self.sum_c, scale=c, row_list=nodes, n_list_dev=cnt_u)
        if work is not None:
            work.wait()
        ops.gather_rows(self.sum_c, self.items_y, nodes, cnt, accumulate=True, scale=c, begin_dev=cnt_u, row_offset=U)
        # ---- loss on the compact tables; softplus share of the global mean carries 1/P
        self.gc_c.zero_()
        self.reg_w.zero_()
        ops.bpr_fwd_bwd(users, pos, neg, self.sum_c, tab, U, self.Lambda, g_final=self.gc_c, reg_w=self.reg_w,
                        g_scale=c / P, loss_out=self.loss, node_map=gmap)
        # ---- global item gradient: dense [I, D], summed over ranks; which item rows are non-zero anywhere
        self.items_g.zero_()
        ops.scatter_rows(self.items_g, self.gc_c, nodes, cnt, begin_dev=cnt_u, row_offset=U)
        flag = (gmap[U:] >= 0).to(t.int32)
        pending = [self._allreduce(self.items_g, async_op=True), self._allreduce(self.reg_w[U:], async_op=True)]
        if self.world > 1:
            pending.append(dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group, async_op=True))
        gmap_u = gmap[:U]
        cur = None
        users_done = False
        layer_work = None  # the exchange whose result is cur[U:] (item rows of the previous backward layer)
        for i in range(K):
            nxt = self.bufs[i % 2]
            if i == 0:  # input = the batch gradient: item rows gather local batch users, user rows the non-zero item rows
                ops.spmm(self.a_items, self.gc_c, Y=nxt[U:], x_map=gmap,    # local operands only: runs beside the exchanges above
                         x_rare=8 * self.batch_size <= U)
                for w in pending:
                    if w is not None:
                        w.wait()
                pending = []
                self.imap[U:] = t.where(flag > 0, self._item_ids, t.full_like(self._item_ids, -1))
                work = self._allreduce(nxt[U:], async_op=True)
                ops.spmm(self.a_users, self.items_g, addend=self.gc_c, S=nxt[:U], x_map=self.imap, addend_map=gmap_u)
            else:
                ops.spmm(self.a_items, cur, Y=nxt[U:])                      # reads the user rows of cur only
                if layer_work is not None:
                    layer_work.wait()
                cur[U:].add_(self.items_g)                                  # previous layer's item rows, now complete
                work = self._allreduce(nxt[U:], async_op=True)
                if i == K - 1:  # user rows are local: their gradient goes straight into Adam (mi_adam_args)
                    hyp = dict(step=self.step_count + 1, lr=self.lr, beta1=self.betas[0], beta2=self.betas[1], eps=self.eps)
                    ops.spmm(self.a_users, cur, addend=self.gc_c, addend_map=gmap_u,
                             adam=dict(p=tab[:U], m=self.m[:U], v=self.v[:U], reg_w=self.reg_w[:U], **hyp))
                    users_done = True
                else:
                    ops.spmm(self.a_users, cur, addend=self.gc_c, S=nxt[:U], addend_map=gmap_u)
            layer_work, cur = work, nxt
        if K >= 1:
            if layer_work is not None:
                layer_work.wait()
            cur[U:].add_(self.items_g)
        for w in pending:  # K == 0: nothing consumed them above
            if w is not None:
                w.wait()
        self.step_count += 1
        lo = U if users_done else 0  # item rows (summed over ranks above), and the user rows unless already updated
        ops.adam_step(tab[lo:], cur[lo:], self.m[lo:], self.v[lo:], step=self.step_count, lr=self.lr,
                      beta1=self.betas[0], beta2=self.betas[1], eps=self.eps, reg_w=self.reg_w[lo:])
        return self.loss

    def decay_lr(self, gamma: float = 0.95) -> None:
        self.lr *= gamma

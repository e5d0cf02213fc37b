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
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch as t
from torch import Tensor

from . import ops
from .interactions import Interactions
from .model.lightgcn import LightGCN, propagate_mean, propagate_mean_backward
from .sparse import SparseTensor


class LightGCNTrainer:
    def __init__(self, model: LightGCN, adj: SparseTensor, train: Interactions, *, lr: float, Lambda: float,
                 batch_size: int, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8, seed: int = 0,
                 neg_range: Optional[int] = None, reference_sampler_quirks: bool = False):
        self.model = model
        self.table = model.table()
        if not self.table.is_cuda:
            raise RuntimeError("LightGCNTrainer needs the model on the GPU (model.to('cuda')); no CPU fallback")
        self.adj_fwd, self.adj_bwd = adj.gcn_normalized(model.add_self_loops)
        self.train = train
        self.K = model.num_iterations
        self.lr, self.Lambda, self.batch_size = float(lr), float(Lambda), int(batch_size)
        self.betas, self.eps, self.seed = betas, float(eps), int(seed)
        # reference: num_nodes = max(train item id)  => negatives in [0, max_item_id)  (Appendix A.3)
        self.neg_range = int(neg_range) if neg_range is not None else train.num_items
        self.quirk = bool(reference_sampler_quirks)
        n, d = self.table.shape
        dev = self.table.device
        self.final = t.empty(n, d, device=dev)
        self.buf_a = t.empty(n, d, device=dev)
        self.buf_b = t.empty(n, d, device=dev)
        self.gc = t.zeros(n, d, device=dev)
        self.reg_w = t.zeros(n, device=dev)
        self.m = t.zeros(n, d, device=dev)
        self.v = t.zeros(n, d, device=dev)
        self.loss = t.zeros(1, device=dev)
        self.batch_idx = tuple(t.empty(self.batch_size, dtype=t.int64, device=dev) for _ in range(3))
        self.step_count = 0
        self._r = train.csr()
        self._row_of_edge = train.row_of_edge()

    # -- pieces (also used one by one in tests) ------------------------------------------------
    def forward(self) -> Tensor:
        return propagate_mean(self.adj_fwd, self.table, self.K, out=self.final, scratch=(self.buf_a, self.buf_b))

    def sample(self) -> Tuple[Tensor, Tensor, Tensor]:
        return ops.sample_bpr_batch(self._r, self._row_of_edge, self.batch_size, self.neg_range, self.seed,
                                    self.step_count, quirk=self.quirk, out=self.batch_idx)

    def step(self, batch: Optional[Tuple[Tensor, Tensor, Tensor]] = None) -> Tensor:
        """One training iteration; returns the loss as a 1-element device tensor (no sync)."""
        K = self.K
        self.forward()
        users, pos, neg = batch if batch is not None else self.sample()
        self.gc.zero_()
        self.reg_w.zero_()
        ops.bpr_fwd_bwd(users, pos, neg, self.final, self.table, self.model.num_users, self.Lambda,
                        g_final=self.gc, reg_w=self.reg_w, g_scale=1.0 / (K + 1), loss_out=self.loss)
        g0 = propagate_mean_backward(self.adj_bwd, self.gc, K, scratch=(self.buf_a, self.buf_b), pre_scaled=True)
        self.step_count += 1
        ops.adam_step(self.table, g0, self.m, self.v, step=self.step_count, lr=self.lr, beta1=self.betas[0],
                      beta2=self.betas[1], eps=self.eps, reg_w=self.reg_w)
        return self.loss

    def decay_lr(self, gamma: float = 0.95) -> None:
        """ExponentialLR(gamma).step() (run_pipeline_lightgcn.py:104,178-179)."""
        self.lr *= gamma

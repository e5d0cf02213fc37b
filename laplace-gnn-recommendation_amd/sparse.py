"""Adjacency types the hot path consumes.

`SparseTensor(row=, col=, sparse_sizes=)` takes the same keyword arguments as
torch_sparse.SparseTensor at the reference's call site (data/lightgcn_loader.py:65-79) and is
what `LightGCN.forward` expects.  It owns the HBM-resident layout the kernels read:

  rowptr int32[N+1], col int32[nnz] (sorted by (row, col), duplicates kept), val fp32[nnz]
  + the same for the transpose (backward), + the split-row plan of each.

gcn_norm values are computed once per (adjacency, add_self_loops) and cached — the reference
recomputes them every forward (model/lightgcn.py:56); the adjacency is immutable so the result
is identical.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch as t
from torch import Tensor

from . import ops
from .ops import DeviceCSR


class SparseTensor:
    def __init__(self, row: Tensor, col: Tensor, sparse_sizes: Tuple[int, int], value: Optional[Tensor] = None,
                 is_symmetric: bool = False):
        """is_symmetric: caller's promise that (row, col) is a symmetric pattern (every (i, j) has its
        (j, i)); the transpose then reuses the forward layout instead of being built."""
        if row.dim() != 1 or row.shape != col.shape:
            raise ValueError("row and col must be 1-D tensors of equal length")
        self._row = row.to(t.int64)
        self._col = col.to(t.int64)
        self._value = value
        self._sizes = (int(sparse_sizes[0]), int(sparse_sizes[1]))
        self.is_symmetric = bool(is_symmetric) and value is None
        self._csr: Optional[DeviceCSR] = None
        self._norm: Dict[bool, Tuple[DeviceCSR, DeviceCSR]] = {}

    # -- torch_sparse-like surface ------------------------------------------------------------
    def sparse_sizes(self) -> Tuple[int, int]:
        return self._sizes

    def nnz(self) -> int:
        return int(self._row.numel())

    def coo(self) -> Tuple[Tensor, Tensor, Optional[Tensor]]:
        return self._row, self._col, self._value

    @property
    def device(self):
        return self._row.device

    def to(self, device) -> "SparseTensor":
        device = t.device(device)
        if device == self._row.device:
            return self
        return SparseTensor(self._row.to(device), self._col.to(device), self._sizes,
                            None if self._value is None else self._value.to(device), self.is_symmetric)

    def permuted(self, node_new_of_old: Tensor) -> "SparseTensor":
        """P A P^T: the same graph with node `v` renamed `node_new_of_old[v]` (square adjacencies)."""
        if self._sizes[0] != self._sizes[1] or node_new_of_old.numel() != self._sizes[0]:
            raise ValueError("a node relabelling needs a square adjacency and one new id per node")
        p = node_new_of_old.to(self._row.device, t.int64)
        return SparseTensor(p[self._row], p[self._col], self._sizes, self._value, self.is_symmetric)

    # -- device layout ------------------------------------------------------------------------
    def csr(self) -> DeviceCSR:
        """Sorted CSR of the raw (un-normalised) adjacency; val = value[perm] or None (= ones)."""
        if self._csr is None:
            a = ops.coo_to_csr(self._row, self._col, self._sizes[0], self._sizes[1], want_perm=True)
            if self._value is not None:
                a.val = ops.gather_f32(self._value.to(t.float32).contiguous(), a.perm)
            self._csr = a
        return self._csr

    def gcn_normalized(self, add_self_loops: bool = False) -> Tuple[DeviceCSR, DeviceCSR]:
        """(A_norm, A_norm^T) as DeviceCSR with values and split-row plans, cached."""
        key = bool(add_self_loops)
        if key not in self._norm:
            base = self
            if add_self_loops:
                base = _with_self_loops(self)
            raw = base.csr()
            val, _dis = ops.gcn_norm(raw, raw.val)
            fwd = DeviceCSR(raw.n_rows, raw.n_cols, raw.rowptr, raw.col, val, raw.perm)
            fwd.plan = ops.build_spmm_plan(fwd)
            if self.is_symmetric:
                bwd = fwd  # D^-1/2 A D^-1/2 of a symmetric pattern is its own transpose
            else:
                bwd = ops.csr_transpose(fwd)
                bwd.plan = ops.build_spmm_plan(bwd)
            self._norm[key] = (fwd, bwd)
        return self._norm[key]


def _with_self_loops(a: SparseTensor) -> SparseTensor:
    """fill_diag(adj, 1.0): existing diagonal entries are replaced by one unit entry per node."""
    n = min(a._sizes)
    row, col, val = a.coo()
    off = row != col
    loop = t.arange(n, dtype=t.int64, device=row.device)
    v = None
    if val is not None:
        v = t.cat([val[off], t.ones(n, dtype=val.dtype, device=val.device)])
    return SparseTensor(t.cat([row[off], loop]), t.cat([col[off], loop]), a._sizes, v, a.is_symmetric)


def as_sparse_tensor(adj) -> SparseTensor:
    """What `LightGCN.forward` accepts as its `edge_index`: this module's SparseTensor, a
    `torch.sparse_csr_tensor` / `torch.sparse_coo_tensor`, or any object with torch_sparse's `.coo()` and
    `.sparse_sizes()` (the real `torch_sparse.SparseTensor` where it is installed).  Edge values other than ones are
    not part of the reference's call (data/lightgcn_loader.py:65-79 builds unit adjacencies) and are rejected.
    Foreign objects are converted once and the result is remembered on them."""
    if isinstance(adj, SparseTensor):
        return adj
    cached = getattr(adj, "_laplace_adj", None)
    if cached is not None:
        return cached
    if isinstance(adj, Tensor) and adj.layout in (t.sparse_csr, t.sparse_coo):
        if adj.layout == t.sparse_csr:
            crow, col, val = adj.crow_indices(), adj.col_indices(), adj.values()
            row = t.repeat_interleave(t.arange(adj.shape[0], device=col.device), crow[1:] - crow[:-1])
        else:
            # not coalesced on purpose: duplicate entries are kept, as torch_sparse keeps them
            idx, val = adj._indices(), adj._values()
            row, col = idx[0], idx[1]
        sizes = (int(adj.shape[0]), int(adj.shape[1]))
    elif hasattr(adj, "coo") and hasattr(adj, "sparse_sizes"):
        row, col, val = adj.coo()
        sizes = tuple(int(x) for x in adj.sparse_sizes())
    else:
        raise TypeError("edge_index must be a SparseTensor(row=, col=, sparse_sizes=), a torch sparse CSR/COO tensor, or an "
                        f"object with .coo() / .sparse_sizes(); got {type(adj)}")
    if val is not None and val.numel() and not bool((val == 1).all()):
        raise ValueError("the adjacency must carry unit values (gcn_norm supplies the edge weights)")
    out = SparseTensor(row=row, col=col, sparse_sizes=sizes)
    try:
        adj._laplace_adj = out
    except AttributeError:
        pass
    return out

"""User x item interaction set in HBM: the CSR the sampler and the top-K exclusion read, and the
two (U+I)x(U+I) adjacencies LightGCN can propagate over.

  compat="reference": rows = user ids, cols = re-based item ids inside an (N, N) matrix, exactly
      what data/lightgcn_loader.py:39-43,61,65-69 builds (one-directional and mis-indexed,
      SURVEY F7) — used for bug-for-bug parity.
  compat="bipartite": the symmetric pattern (u, U+i), (U+i, u) that LightGCN's normalised
      adjacency is defined on — used for the performance configurations.
"""
from __future__ import annotations

from typing import Optional

import torch as t
from torch import Tensor

from . import ops
from .ops import DeviceCSR
from .sparse import SparseTensor


class Interactions:
    def __init__(self, edge_index: Tensor, num_users: int, num_items: int):
        if edge_index.dim() != 2 or edge_index.shape[0] != 2:
            raise ValueError("edge_index must be [2, E]: users, then item ids in [0, num_items)")
        self.edge_index = edge_index.to(t.int64)
        self.num_users, self.num_items = int(num_users), int(num_items)
        self._csr: Optional[DeviceCSR] = None
        self._row_of_edge: Optional[Tensor] = None

    @property
    def device(self):
        return self.edge_index.device

    @property
    def num_edges(self) -> int:
        return int(self.edge_index.shape[1])

    def to(self, device) -> "Interactions":
        if t.device(device) == self.edge_index.device:
            return self
        return Interactions(self.edge_index.to(device), self.num_users, self.num_items)

    def csr(self) -> DeviceCSR:
        """users x items, columns sorted within each row (binary-searchable)."""
        if self._csr is None:
            self._csr = ops.coo_to_csr(self.edge_index[0].contiguous(), self.edge_index[1].contiguous(),
                                       self.num_users, self.num_items, want_perm=False)
        return self._csr

    def row_of_edge(self) -> Tensor:
        if self._row_of_edge is None:
            self._row_of_edge = ops.expand_rows(self.csr())
        return self._row_of_edge

    def adjacency(self, compat: str = "bipartite") -> SparseTensor:
        n = self.num_users + self.num_items
        u, i = self.edge_index[0], self.edge_index[1]
        if compat == "reference":
            return SparseTensor(row=u, col=i, sparse_sizes=(n, n))
        if compat == "bipartite":
            iu = i + self.num_users
            return SparseTensor(row=t.cat([u, iu]), col=t.cat([iu, u]), sparse_sizes=(n, n), is_symmetric=True)
        raise ValueError(f"unknown compat mode {compat!r}")

"""User x item interaction set in HBM: the CSR the sampler and the top-K exclusion read, and the
two (U+I)x(U+I) adjacencies LightGCN can propagate over.

  compat="reference": rows = user ids, cols = re-based item ids inside an (N, N) matrix, exactly
      what data/lightgcn_loader.py:39-43,61,65-69 builds (one-directional and mis-indexed,
      SURVEY F7) — used for bug-for-bug parity.
  compat="bipartite": the symmetric pattern (u, U+i), (U+i, u) that LightGCN's normalised
      adjacency is defined on — used for the performance configurations.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch as t
from torch import Tensor

from . import ops
from .ops import DeviceCSR
from .sparse import SparseTensor


@dataclass
class LocalityOrder:
    """A relabelling of users and items chosen for gather locality (SURVEY section 7 "Gather locality").

    Items are numbered by popularity (most interactions first), users by their COLDEST item (the largest new item
    id among their interactions).  Consecutive user rows of the propagate then end their neighbour lists on the
    same or adjacent cold item rows — the gathers that miss L2 — and a cold item row finds the users that have it
    as their coldest item on consecutive rows of the table.  `*_new_of_old[old id] = new id`; `*_old_of_new` is the
    inverse.  Node form (users, then items): node_new_of_old."""
    user_new_of_old: Tensor
    item_new_of_old: Tensor
    user_old_of_new: Tensor
    item_old_of_new: Tensor

    @property
    def num_users(self) -> int:
        return int(self.user_new_of_old.numel())

    def node_new_of_old(self) -> Tensor:
        return t.cat([self.user_new_of_old, self.item_new_of_old + self.num_users])

    def node_old_of_new(self) -> Tensor:
        return t.cat([self.user_old_of_new, self.item_old_of_new + self.num_users])

    def to(self, device) -> "LocalityOrder":
        return LocalityOrder(*(x.to(device) for x in (self.user_new_of_old, self.item_new_of_old,
                                                      self.user_old_of_new, self.item_old_of_new)))


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

    def locality_order(self, item_degree: Optional[Tensor] = None) -> LocalityOrder:
        """See LocalityOrder.  item_degree: interaction counts to rank the items by (default: this edge set's; the
        sharded trainer passes the GLOBAL counts so that every rank numbers the replicated items alike).  Stable
        sorts: the order is a function of the edge set alone."""
        u, i = self.edge_index[0], self.edge_index[1]
        dev = u.device
        U, I = self.num_users, self.num_items
        deg = item_degree if item_degree is not None else t.bincount(i, minlength=I)
        item_old_of_new = t.argsort(deg.to(t.int64), descending=True, stable=True)
        item_new_of_old = t.empty(I, dtype=t.int64, device=dev)
        item_new_of_old[item_old_of_new] = t.arange(I, dtype=t.int64, device=dev)
        cold = t.zeros(U, dtype=t.int64, device=dev)
        if u.numel():
            cold.scatter_reduce_(0, u, item_new_of_old[i], "amax", include_self=True)  # integer max: order-independent
        user_old_of_new = t.argsort(cold, stable=True)
        user_new_of_old = t.empty(U, dtype=t.int64, device=dev)
        user_new_of_old[user_old_of_new] = t.arange(U, dtype=t.int64, device=dev)
        return LocalityOrder(user_new_of_old, item_new_of_old, user_old_of_new, item_old_of_new)

    def permuted(self, order: LocalityOrder) -> "Interactions":
        """The same interactions under new ids."""
        ei = t.stack([order.user_new_of_old[self.edge_index[0]], order.item_new_of_old[self.edge_index[1]]])
        return Interactions(ei, self.num_users, self.num_items)

    def adjacency(self, compat: str = "bipartite") -> SparseTensor:
        n = self.num_users + self.num_items
        u, i = self.edge_index[0], self.edge_index[1]
        if compat == "reference":
            return SparseTensor(row=u, col=i, sparse_sizes=(n, n))
        if compat == "bipartite":
            iu = i + self.num_users
            return SparseTensor(row=t.cat([u, iu]), col=t.cat([iu, u]), sparse_sizes=(n, n), is_symmetric=True)
        raise ValueError(f"unknown compat mode {compat!r}")

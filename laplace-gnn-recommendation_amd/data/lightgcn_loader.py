"""LightGCN data plumbing with the reference's function names (data/lightgcn_loader.py).

The reference unpickles a PyG HeteroData (`data/derived/test_graph.pt`); PyG is not a dependency
here, so `create_dataloaders_lightgcn` takes the homogeneous edge_index (what
`HeteroData.to_homogeneous().edge_index` holds: users first, then items offset by the user count)
or a path to a tensor file holding it, and returns the same 9-tuple.
"""
from __future__ import annotations

from typing import Optional, Tuple, Union

import numpy as np
import torch as t
from torch import Tensor

from .. import ops
from ..interactions import Interactions


def split(edge_index: Tensor, seed: int = 1):
    """80/10/10 train/val/test edge split (data/lightgcn_loader.py:13-31).  The reference draws it with
    sklearn.train_test_split(random_state=1); the shuffle here is numpy's with the same seed — the
    proportions and disjointness are the contract, the exact permutation is not (RNG streams differ)."""
    n = edge_index.shape[1]
    perm = np.random.RandomState(seed).permutation(n)
    n_test_all = int(np.ceil(0.2 * n))
    train_idx, rest = perm[: n - n_test_all], perm[n - n_test_all:]
    n_test = int(np.ceil(0.5 * rest.size))
    val_idx, test_idx = rest[: rest.size - n_test], rest[rest.size - n_test:]
    pick = lambda idx: edge_index[:, t.from_numpy(np.sort(idx)).to(edge_index.device)]
    return pick(train_idx), pick(val_idx), pick(test_idx), edge_index


def both_indexes_from_zero(edge_index: Tensor) -> Tensor:
    """Re-base the second row by max(first row)+1 (data/lightgcn_loader.py:39-43)."""
    out = edge_index.clone()
    out[1] = out[1] - (t.max(out[0]) + 1)
    return out


def create_dataloaders_lightgcn(edge_index: Union[Tensor, str], num_users: int, num_articles: int,
                                compat: str = "reference", device: Optional[str] = None):
    """Returns (train_sparse, val_sparse, test_sparse, train_edge_index, val_edge_index, test_edge_index,
    edge_index, num_users, num_articles) like data/lightgcn_loader.py:54-92.

    compat="reference": adjacencies are SparseTensor(row=user, col=item, (N, N)) exactly as the
    reference builds them; compat="bipartite": the symmetric bipartite adjacency."""
    if isinstance(edge_index, str):
        edge_index = t.load(edge_index)
    if int(edge_index[1].max()) >= num_articles:  # homogeneous ids: items offset by the user count
        edge_index = both_indexes_from_zero(edge_index)
    if device is not None:
        edge_index = edge_index.to(device)
    train_e, val_e, test_e, all_e = split(edge_index)
    mk = lambda e: Interactions(e, num_users, num_articles).adjacency(compat)
    return mk(train_e), mk(val_e), mk(test_e), train_e, val_e, test_e, all_e, num_users, num_articles


_SAMPLER_STATE = {"step": 0}


def sample_mini_batch(batch_size: int, edge_index: Tensor, seed: int = 0, step: Optional[int] = None
                      ) -> Tuple[Tensor, Tensor, Tensor]:
    """(users, positive items, negative items) — data/lightgcn_loader.py:95-112 on device, including the
    reference's negative range [0, max item id) and its key-collision quirk."""
    if step is None:
        step = _SAMPLER_STATE["step"]
        _SAMPLER_STATE["step"] += 1
    num_users = int(edge_index[0].max()) + 1
    max_item = int(edge_index[1].max())
    inter = _cached_interactions(edge_index, num_users, max_item + 1)
    return ops.sample_bpr_batch(inter.csr(), inter.row_of_edge(), batch_size, max_item, seed, step, quirk=True)


_INTER_CACHE: dict = {}


def _cached_interactions(edge_index: Tensor, num_users: int, num_items: int) -> Interactions:
    key = (edge_index.data_ptr(), tuple(edge_index.shape), str(edge_index.device))
    hit = _INTER_CACHE.get(key)
    if hit is None:
        _INTER_CACHE.clear()
        hit = Interactions(edge_index, num_users, num_items)
        _INTER_CACHE[key] = hit
    return hit

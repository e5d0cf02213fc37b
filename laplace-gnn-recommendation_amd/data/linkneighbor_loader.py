"""`LinkNeighborLoader` call shape over the on-device N-hop sampler.

The reference's data/linkneighbor_loader.py:25-106 configures PyG's `LinkNeighborLoader` (fan-out `[num_neighbors] *
n_hop_neighbors`, `directed=False`, `replace=False`, `shuffle=True`) but is dead code: the pipelines import
data/data_loader.py, whose live sampler is `GraphDataset.__getitem__`.  The algorithm run here is that live one
(csrc/sampler.hip); this module only accepts the loader's keyword surface so that a call site written against
`LinkNeighborLoader(...)` gets device-sampled batches:

    num_neighbors=[n] * hops   -> fan-out cap n on both frontiers, `hops` hops (all entries must be equal, as upstream)
    batch_size                 -> seed USERS per batch (the live sampler's unit; PyG counts label edges)
    edge_label_index / edge_label -> accepted and ignored: the live sampler draws its own label edges per seed user
                                     (positives with replacement, `negative_edges_ratio` negatives; data/dataset.py:42-106)
    directed=False             -> both relation directions are emitted (`buys` and `rev_buys`)
    replace=False              -> frontier cuts are subsets (Floyd), never multisets
    shuffle                    -> seed users in a fresh random order every epoch, or in id order
    num_workers / pin_memory   -> accepted and ignored: nothing is staged through the host
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional, Sequence, Tuple

import numpy as np

from ..hetero import HeteroData
from ..utils.constants import Constants
from .dataset import AdjList
from .device_sampler import DeviceGraphSampler


class LinkNeighborLoader(DeviceGraphSampler):
    def __init__(self, data: HeteroData, num_neighbors: Sequence[int], batch_size: int = 1,
                 edge_label_index: Optional[Tuple] = None, edge_label=None, directed: bool = False, replace: bool = False,
                 shuffle: bool = False, num_workers: int = 0, pin_memory: bool = False, *, k: int = 12,
                 positive_edges_ratio: float = 0.5, negative_edges_ratio: float = 3.0, device: str = "cuda", seed: int = 0,
                 matchers=None, **_ignored):
        del edge_label_index, edge_label, num_workers, pin_memory
        fan = [int(x) for x in num_neighbors]
        if not fan or any(x != fan[0] for x in fan):
            raise ValueError("num_neighbors must be [n] * hops (one fan-out for every hop)")
        if directed or replace:
            raise ValueError("only directed=False, replace=False (the reference's configuration) is implemented")
        ei = data[Constants.edge_key].edge_index
        u, a = ei[0].cpu().numpy(), ei[1].cpu().numpy()
        n_users = int(data[Constants.node_user].x.shape[0])
        n_articles = int(data[Constants.node_item].x.shape[0])
        users, articles = AdjList.from_edges(u, a, n_users), AdjList.from_edges(a, u, n_articles)
        cfg = SimpleNamespace(k=int(k), num_neighbors=fan[0], n_hop_neighbors=len(fan), batch_size=int(batch_size),
                              positive_edges_ratio=float(positive_edges_ratio),
                              negative_edges_ratio=float(negative_edges_ratio))
        super().__init__(cfg, data, users, articles, batch_size=int(batch_size), device=device, seed=seed,
                         train=matchers is None, matchers=matchers, shuffle=bool(shuffle))


def create_dataloaders(config, splits: dict, matchers: Optional[dict] = None, customer_id_map: Optional[dict] = None,
                       article_id_map: Optional[dict] = None, device: str = "cuda", seed: int = 0):
    """data/linkneighbor_loader.py:25-104: (train_loader, val_loader, test_loader, customer_id_map, article_id_map,
    full_data).  `splits` as data.graph_io.read_splits returns them; validation / test loaders sample with the fixed
    `[64] * 2` fan-out the reference writes and need the split's matchers (evaluation samples rank candidates)."""
    from .data_loader import to_undirected
    common = dict(batch_size=config.batch_size, directed=False, replace=False, k=config.k,
                  positive_edges_ratio=config.positive_edges_ratio, negative_edges_ratio=config.negative_edges_ratio,
                  device=device)
    matchers = matchers or {}
    train = LinkNeighborLoader(splits["train"][0], [config.num_neighbors] * config.n_hop_neighbors, shuffle=True, seed=seed,
                               **common)
    held = []
    for i, name in enumerate(("val", "test")):
        held.append(LinkNeighborLoader(splits[name][0], [64] * 2, shuffle=True, seed=seed + 1 + i,
                                       matchers=matchers.get(name), **common) if matchers.get(name) is not None else None)
    return train, held[0], held[1], customer_id_map or {}, article_id_map or {}, to_undirected(splits["train"][0])

"""Ranker data loaders with the reference's return tuple (data/data_loader.py:14-65).  The graphs are
passed in (our HeteroData + adjacency dicts) rather than unpickled from PyG files."""
from __future__ import annotations

from typing import Optional, Tuple

from ..hetero import DataLoader, HeteroData
from ..utils.constants import Constants
from .dataset import GraphDataset


def to_undirected(graph: HeteroData) -> HeteroData:
    """T.ToUndirected() for the customer-article graph: add the rev_buys store (data/data_loader.py:52-53)."""
    if Constants.rev_edge_key not in graph.edge_types:
        graph[Constants.rev_edge_key].edge_index = graph[Constants.edge_key].edge_index.flip(0)
    return graph


def create_dataloaders(config, splits: dict, matchers: Optional[dict] = None, customer_id_map: Optional[dict] = None,
                       article_id_map: Optional[dict] = None, seed: int = 0
                       ) -> Tuple[DataLoader, DataLoader, DataLoader, dict, dict, HeteroData]:
    """splits: {"train"|"val"|"test": (graph, users_adj_list, articles_adj_list)}.  Returns
    (train_loader, val_loader, test_loader, customer_id_map, article_id_map, full_data)."""
    matchers = matchers or {}
    ds = {}
    for i, name in enumerate(("train", "val", "test")):
        graph, users_adj, articles_adj = splits[name]
        ds[name] = GraphDataset(config, graph, users_adj, articles_adj, train=(name == "train"),
                                matchers=matchers.get(name), split_type=name, seed=seed + i)
    loaders = [DataLoader(ds[n], batch_size=config.batch_size, shuffle=True) for n in ("train", "val", "test")]
    full = to_undirected(ds["train"].graph)
    return loaders[0], loaders[1], loaders[2], customer_id_map or {}, article_id_map or {}, full

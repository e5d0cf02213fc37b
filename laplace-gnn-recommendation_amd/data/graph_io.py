"""Graph files in the reference's on-disk layout: what run_preprocessing.py:100-195
writes and data/data_loader.py:14-65 reads —

    data/derived/{train,val,test}_graph.pt            node feature tables + `buys` edge_index
    data/derived/edges_{split}.pt                      dict customer -> [articles]   (list order = time order)
    data/derived/rev_edges_{split}.pt                  dict article  -> [customers]
    data/derived/{customer,article}_id_map_forward.json

with the chronological leave-last-two-out split of run_data_splitting.py:36-52 (per customer: last
transaction -> test, second to last -> val; val = train + val edges, test = val + test edges,
run_preprocessing.py:112-121).  The graph objects are this package's HeteroData (PyG's class cannot be
pickled without PyG); everything else is the same plain Python the reference writes.
"""
from __future__ import annotations

import json
import os
from typing import Dict, List, Tuple

import numpy as np
import torch as t
from torch import Tensor

from ..hetero import HeteroData
from ..utils.constants import Constants


def train_test_split_by_time(user_ids: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(train_mask, val_mask, test_mask) for transactions already in time order
    (run_data_splitting.py:36-52): per user the last row is test when the user has > 1 rows, the row
    before it is val when the user has > 2 rows."""
    user_ids = np.asarray(user_ids)
    n = user_ids.shape[0]
    order = np.argsort(user_ids, kind="stable")            # groups, time order kept inside each
    su = user_ids[order]
    start = np.r_[True, su[1:] != su[:-1]]
    gid = np.cumsum(start) - 1
    count = np.bincount(gid)[gid]
    pos_from_end = (np.r_[np.flatnonzero(start)[1:], n][gid] - 1) - np.arange(n)
    test_s = (pos_from_end == 0) & (count > 1)
    val_s = (pos_from_end == 1) & (count > 2)
    test, val = np.zeros(n, dtype=bool), np.zeros(n, dtype=bool)
    test[order], val[order] = test_s, val_s
    return ~(test | val), val, test


def _adj_dict(src: np.ndarray, dst: np.ndarray) -> Dict[int, List[int]]:
    """groupby(src)[dst].apply(list).to_dict() (utils/preprocessing.py:84-89): keys ascending, list order kept."""
    order = np.argsort(src, kind="stable")
    s, d = src[order], dst[order]
    cut = np.flatnonzero(np.r_[True, s[1:] != s[:-1]])
    ends = np.r_[cut[1:], s.shape[0]]
    return {int(s[a]): d[a:b].tolist() for a, b in zip(cut, ends)}


def build_splits(customer_x: Tensor, article_x: Tensor, customer_ids: np.ndarray, article_ids: np.ndarray):
    """Transactions in time order -> {"train"|"val"|"test": (graph, edges, rev_edges)} with cumulative edge sets."""
    tr, va, te = train_test_split_by_time(customer_ids)
    idx_train = np.flatnonzero(tr)
    idx_val = np.concatenate([idx_train, np.flatnonzero(va)])     # pd.concat([train, val]) order
    idx_test = np.concatenate([idx_val, np.flatnonzero(te)])
    out = {}
    for name, idx in (("train", idx_train), ("val", idx_val), ("test", idx_test)):
        c, a = np.asarray(customer_ids)[idx], np.asarray(article_ids)[idx]
        g = HeteroData()
        g[Constants.node_user].x = customer_x
        g[Constants.node_item].x = article_x
        g[Constants.edge_key].edge_index = t.from_numpy(np.stack([c, a]).astype(np.int64))
        out[name] = (g, _adj_dict(c, a), _adj_dict(a, c))
    return out


def write_splits(splits: dict, directory: str, customer_id_map: dict, article_id_map: dict) -> None:
    os.makedirs(directory, exist_ok=True)
    for name, (graph, edges, rev_edges) in splits.items():
        t.save(graph, os.path.join(directory, f"{name}_graph.pt"))
        t.save(edges, os.path.join(directory, f"edges_{name}.pt"))
        t.save(rev_edges, os.path.join(directory, f"rev_edges_{name}.pt"))
    with open(os.path.join(directory, "customer_id_map_forward.json"), "w") as fp:
        json.dump(customer_id_map, fp)
    with open(os.path.join(directory, "article_id_map_forward.json"), "w") as fp:
        json.dump(article_id_map, fp)


def read_splits(directory: str):
    """-> (splits, customer_id_map, article_id_map) ready for data.data_loader.create_dataloaders."""
    splits = {}
    for name in ("train", "val", "test"):
        splits[name] = (t.load(os.path.join(directory, f"{name}_graph.pt"), weights_only=False),
                        t.load(os.path.join(directory, f"edges_{name}.pt"), weights_only=False),
                        t.load(os.path.join(directory, f"rev_edges_{name}.pt"), weights_only=False))
    with open(os.path.join(directory, "customer_id_map_forward.json")) as fp:
        cmap = json.load(fp)
    with open(os.path.join(directory, "article_id_map_forward.json")) as fp:
        amap = json.load(fp)
    return splits, cmap, amap

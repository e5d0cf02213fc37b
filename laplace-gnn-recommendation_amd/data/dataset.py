"""Per-user subgraph sampler — the live N-hop sampler of the reference (`GraphDataset.__getitem__`,
data/dataset.py:39-309) on flat CSR arrays instead of dicts of Python lists and sets.

One item = one user's training example: sampled positive label edges (with replacement), negative
label edges, the user's own edges, and the edges of its N-hop neighbourhood with a fan-out cap on
both the article and the user frontier; nodes relabelled by sorted-unique buckets; forward and
reverse relation stores.  `randomization=False` is the deterministic mode the reference's
tests/test_dataset.py pins (argmin/argmax positives, id_max negative, no shuffling).

Host-side (numpy) in this round: it feeds the HIP ranker and is checked item for item against the
oracle's literal restatement.  The on-device version is SURVEY §8f row N1.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch as t

from ..hetero import HeteroData
from ..utils.constants import Constants


class AdjList:
    """dict[int -> list[int]] (the reference's edges_*.pt / rev_edges_*.pt) flattened to CSR, list order kept."""

    def __init__(self, adj: Union[Dict[int, Sequence[int]], "AdjList"], n: Optional[int] = None):
        if isinstance(adj, AdjList):
            self.ptr, self.idx = adj.ptr, adj.idx
            return
        n = (max(adj) + 1 if adj else 0) if n is None else n
        counts = np.zeros(n, dtype=np.int64)
        for k, v in adj.items():
            counts[k] = len(v)
        self.ptr = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(counts, out=self.ptr[1:])
        self.idx = np.empty(int(self.ptr[-1]), dtype=np.int64)
        for k, v in adj.items():
            self.idx[self.ptr[k]:self.ptr[k + 1]] = np.asarray(v, dtype=np.int64)

    @classmethod
    def from_edges(cls, src: np.ndarray, dst: np.ndarray, n: int) -> "AdjList":
        order = np.argsort(src, kind="stable")
        out = cls.__new__(cls)
        out.ptr = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(np.bincount(src, minlength=n), out=out.ptr[1:])
        out.idx = dst[order].astype(np.int64)
        return out

    def __len__(self) -> int:
        return self.ptr.shape[0] - 1

    def __getitem__(self, k: int) -> np.ndarray:
        return self.idx[self.ptr[k]:self.ptr[k + 1]]

    def gather(self, keys: np.ndarray):
        """(flat neighbours of all keys in key order, repeat counts)."""
        cnt = self.ptr[keys + 1] - self.ptr[keys]
        if cnt.sum() == 0:
            return np.empty(0, dtype=np.int64), cnt
        start = np.repeat(self.ptr[keys], cnt)
        within = np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt)
        return self.idx[start + within], cnt


class GraphDataset:
    def __init__(self, config, graph: HeteroData, users_adj_list, articles_adj_list, train: bool,
                 matchers: Optional[list] = None, randomization: bool = True, split_type: Optional[str] = None,
                 seed: int = 0):
        if isinstance(graph, str):
            graph = t.load(graph, weights_only=False)
        if isinstance(users_adj_list, str):
            users_adj_list = t.load(users_adj_list, weights_only=False)
        if isinstance(articles_adj_list, str):
            articles_adj_list = t.load(articles_adj_list, weights_only=False)
        self.graph = graph
        n_users = graph[Constants.node_user].x.shape[0]
        n_articles = graph[Constants.node_item].x.shape[0]
        self.users = AdjList(users_adj_list, n_users)
        self.articles = AdjList(articles_adj_list, n_articles)
        self._n_items = len(users_adj_list) if isinstance(users_adj_list, dict) else n_users
        self.matchers, self.config, self.train, self.randomization = matchers, config, train, randomization
        all_edges = graph[Constants.edge_key].edge_index
        self._num_edges = int(all_edges.shape[1])
        self._id_max = int(all_edges[1].max())
        self._rng = np.random.default_rng(seed)
        self._mark = np.zeros(n_users, dtype=bool)

    def __len__(self) -> int:
        return self._n_items

    # ---- pieces ------------------------------------------------------------------------------
    def _negatives(self, sampled_pos: np.ndarray, num_negative: int) -> np.ndarray:
        """get_negative_edges_random (data/dataset.py:189-230)."""
        id_max = self._id_max
        if num_negative == 0 or self._num_edges / num_negative > 100:  # fast path: no positive filter, range [0, id_max)
            if self.randomization:
                return self._rng.integers(0, id_max, size=num_negative)
            return np.array([id_max], dtype=np.int64)
        cand = np.setdiff1d(np.arange(id_max + 1, dtype=np.int64), sampled_pos)  # ids seen exactly once
        if self.randomization:
            return self._rng.permutation(cand)[:num_negative]
        return np.array([id_max], dtype=np.int64)

    def _cut(self, arr: np.ndarray, n: int) -> np.ndarray:
        """shuffle_and_cut: a uniform n-subset (of positions) when longer than n, else unchanged."""
        if arr.shape[0] > n:
            return arr[self._rng.choice(arr.shape[0], size=n, replace=False)]
        return arr

    def _n_hop_edges(self, user_id: int):
        """fetch_n_hop_neighbourhood (data/dataset.py:258-286): edges of hops >= 1 only."""
        cfg = self.config
        us, arts = [], []
        mark = self._mark            # scratch bitmap over users, all False between calls
        explored: List[np.ndarray] = []
        queue = np.array([user_id], dtype=np.int64)
        for hop in range(cfg.n_hop_neighbors):
            if queue.size == 0:
                break
            new_articles, cnt = self.users.gather(queue)
            explored.append(queue)
            if hop != 0:
                us.append(np.repeat(queue, cnt))
                arts.append(new_articles)
            if hop == cfg.n_hop_neighbors - 1:
                break  # the next frontier would never be expanded
            article_queue = self._cut(new_articles, cfg.num_neighbors)
            # distinct unexplored users of those articles: mark / unmark instead of sort-unique — hub
            # articles carry 10^5..10^6 users each
            ptr, idx = self.articles.ptr, self.articles.idx
            for a in article_queue.tolist():
                mark[idx[ptr[a]:ptr[a + 1]]] = True
            for q in explored:
                mark[q] = False
            cand_users = np.flatnonzero(mark)
            mark[cand_users] = False
            queue = np.sort(self._cut(cand_users, cfg.num_neighbors))
        if not us:
            return np.empty(0, dtype=np.int64), np.empty(0, dtype=np.int64)
        return np.concatenate(us), np.concatenate(arts)

    # ---- one item ------------------------------------------------------------------------------
    def __getitem__(self, idx: int) -> HeteroData:
        cfg = self.config
        pos = self.users[idx]
        samp_cut = max(1, math.floor(len(pos) * cfg.positive_edges_ratio))
        if self.randomization:
            pick = self._rng.integers(0, len(pos), size=samp_cut)   # with replacement (data/dataset.py:58-60)
        else:
            pick = np.array([int(np.argmin(pos)), int(np.argmax(pos))])
        sampled_pos = pos[pick]
        n_pos = sampled_pos.shape[0]
        ratio = cfg.k - 1 if n_pos <= 1 else cfg.negative_edges_ratio
        if self.train:
            sampled_neg = self._negatives(sampled_pos, int(ratio * n_pos))
        else:
            assert self.matchers is not None, "Must provide matchers for test"
            cand = np.unique(np.concatenate([np.asarray(m.get_matches(idx)).astype(np.int64) for m in self.matchers]))
            # only_items_with_count_one(cat(candidates, positives)) (data/dataset.py:99-105): ids seen exactly
            # once — candidates that are not positives AND, as written, positives the matchers missed
            ids, counts = np.unique(np.concatenate([cand, pos]), return_counts=True)
            sampled_neg = ids[counts == 1]
        hop_u, hop_a = self._n_hop_edges(idx)

        sub_u = np.concatenate([np.full(len(pos), idx, dtype=np.int64), hop_u])
        sub_a = np.concatenate([pos, hop_a])
        lab_u = np.full(n_pos + sampled_neg.shape[0], idx, dtype=np.int64)
        lab_a = np.concatenate([sampled_pos, sampled_neg])
        user_buckets = np.unique(np.concatenate([sub_u, lab_u]))
        article_buckets = np.unique(np.concatenate([sub_a, lab_a]))

        data = HeteroData()
        data[Constants.node_user].x = self.graph[Constants.node_user].x[t.from_numpy(user_buckets)]
        data[Constants.node_item].x = self.graph[Constants.node_item].x[t.from_numpy(article_buckets)]
        data[Constants.node_user].n_id = t.from_numpy(user_buckets)      # global ids of the relabelled nodes: what
        data[Constants.node_item].n_id = t.from_numpy(article_buckets)   # run_submission needs to name its picks
        edge_index = t.from_numpy(np.stack([np.searchsorted(user_buckets, sub_u), np.searchsorted(article_buckets, sub_a)]))
        label_index = t.from_numpy(np.stack([np.searchsorted(user_buckets, lab_u), np.searchsorted(article_buckets, lab_a)]))
        labels = t.cat([t.ones(n_pos, dtype=t.long), t.zeros(sampled_neg.shape[0], dtype=t.long)])
        data[Constants.edge_key].edge_index = edge_index
        data[Constants.edge_key].edge_label_index = label_index
        data[Constants.edge_key].edge_label = labels
        data[Constants.rev_edge_key].edge_index = edge_index.flip(0)
        data[Constants.rev_edge_key].edge_label_index = label_index.flip(0)
        data[Constants.rev_edge_key].edge_label = labels
        return data

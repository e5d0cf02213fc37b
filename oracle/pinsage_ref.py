"""ORACLE — test infrastructure only.  PARITY UNPINNED: the reference's pinsage/ is a vendored DGL
example that cannot be imported and DGL is absent, so nothing here is checked against a
run of the reference; it restates

  * the reference's own files: pinsage/sampler.py:16-106 (batch sampler, block construction, removal
    of the label edges from the frontier), pinsage/layers.py:121-203 (WeightedSAGEConv, SAGENet,
    ItemToItemScorer), pinsage/model.py:16-34 (get_repr, hinge loss);
  * DGL 0.8's documented sampler semantics reached from them:
      dgl.sampling.random_walk(metapath, restart_prob) — uniform neighbour per step, the trace ends
        (padded with -1) at a node without neighbours or, before a step whose restart_prob is p,
        with probability p;
      dgl.sampling.PinSAGESampler(g, item, user, L, p, W, T) — W walks per seed of L traversals of
        item->user->item, termination probability p before every traversal but the first; the items
        reached at the end of each traversal are counted per seed and the T most visited become the
        seed's neighbours with weight = visit count;
      dgl.to_block(frontier, seeds) — destination nodes = seeds (kept in order), source nodes = the
        seeds followed by the new sources.
    Ties among equal visit counts and the order of new sources are unspecified upstream; here: smaller
    item id first / ascending ids.

The numpy half mirrors csrc/pinsage.hip draw for draw (Philox4x32-10), so the device samplers are
checked bit for bit; the torch half is the plain-torch twin of the model.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np
import torch as t
import torch.nn.functional as F
from torch import Tensor, nn

from .philox import philox4x32

P_HEAD, P_NEG, P_WALK, P_NEIGH = 11, 12, 13, 14


def _words(purpose: int, a: int, b: int, c: int, seed: int, step: int):
    c3 = (purpose & 0xFF) | ((step & 0xFFFFFF) << 8)
    k0, k1 = seed & 0xFFFFFFFF, ((seed >> 32) ^ (step >> 24)) & 0xFFFFFFFF
    return [int(x) for x in philox4x32(a & 0xFFFFFFFF, b & 0xFFFFFFFF, c & 0xFFFFFFFF, c3, k0, k1)]


class Csr:
    def __init__(self, ptr, idx):
        self.ptr, self.idx = np.asarray(ptr, dtype=np.int64), np.asarray(idx, dtype=np.int64)

    def __getitem__(self, k):
        return self.idx[self.ptr[k]:self.ptr[k + 1]]


def _hop(item: int, item_users: Csr, user_items: Csr, w0: int, w1: int) -> int:
    """item -> uniform user -> uniform item; -1 when the item has no users."""
    us = item_users[item]
    if len(us) == 0:
        return -1
    u = int(us[w0 % len(us)])
    its = user_items[u]
    if len(its) == 0:
        return -1
    return int(its[w1 % len(its)])


def item_pairs(batch: int, n_items: int, item_users: Csr, user_items: Csr, seed: int, step: int):
    """ItemToItemBatchSampler (pinsage/sampler.py:16-41): heads uniform, tails = end of one item->user->item
    walk, negative tails uniform; pairs whose walk died are dropped."""
    heads, tails, negs = [], [], []
    for b in range(batch):
        w = _words(P_HEAD, b, 0, 0, seed, step)
        h = w[0] % n_items
        tl = _hop(h, item_users, user_items, w[1], w[2])
        ng = _words(P_NEG, b, 0, 0, seed, step)[0] % n_items
        if tl != -1:
            heads.append(h); tails.append(tl); negs.append(ng)
    return np.array(heads, dtype=np.int64), np.array(tails, dtype=np.int64), np.array(negs, dtype=np.int64)


def pinsage_neighbors(seeds, item_users: Csr, user_items: Csr, walk_length: int, restart_prob: float,
                      num_walks: int, num_neighbors: int, layer: int, seed: int, step: int):
    """PinSAGESampler(seeds): ([n, T] neighbour ids, -1 padded; [n, T] visit counts)."""
    n = len(seeds)
    nb = np.full((n, num_neighbors), -1, dtype=np.int64)
    wt = np.zeros((n, num_neighbors), dtype=np.int64)
    thr = int(restart_prob * 4294967296.0)  # terminate when the 32-bit draw is below p * 2^32
    for i, s in enumerate(seeds):
        visits: Dict[int, int] = {}
        for wk in range(num_walks):
            cur = int(s)
            for tr in range(walk_length):
                w = _words(P_WALK, wk * walk_length + tr, layer, int(s), seed, step)
                if tr > 0 and w[2] < thr:
                    break
                cur = _hop(cur, item_users, user_items, w[0], w[1])
                if cur == -1:
                    break
                visits[cur] = visits.get(cur, 0) + 1
        top = sorted(visits.items(), key=lambda kv: (-kv[1], kv[0]))[:num_neighbors]
        for j, (v, c) in enumerate(top):
            nb[i, j], wt[i, j] = v, c
    return nb, wt


def build_blocks(seeds: np.ndarray, item_users: Csr, user_items: Csr, n_layers: int, walk_length: int, restart_prob: float,
                 num_walks: int, num_neighbors: int, seed: int, step: int,
                 heads=None, tails=None, neg_tails=None) -> List[dict]:
    """NeighborSampler.sample_blocks (pinsage/sampler.py:73-91).  Each block: src_ids (dst nodes first), n_dst,
    edge_src / edge_dst (block-local), weights.  Returned input layer first."""
    blocks = []
    seeds = np.asarray(seeds, dtype=np.int64)
    banned = set()
    if heads is not None:
        banned = set(zip(heads.tolist(), tails.tolist())) | set(zip(heads.tolist(), neg_tails.tolist()))
    for layer in range(n_layers):
        nb, wt = pinsage_neighbors(seeds, item_users, user_items, walk_length, restart_prob, num_walks, num_neighbors,
                                   layer, seed, step)
        es, ed, ew = [], [], []
        for i, s in enumerate(seeds):
            for j in range(num_neighbors):
                v = int(nb[i, j])
                if v < 0 or (v, int(s)) in banned:  # frontier edge v -> s; label pairs (head -> tail) are removed
                    continue
                es.append(v); ed.append(i); ew.append(int(wt[i, j]))
        new = np.setdiff1d(np.unique(np.array(es, dtype=np.int64)), seeds) if es else np.empty(0, dtype=np.int64)
        src_ids = np.concatenate([seeds, new])
        pos = {int(v): k for k, v in enumerate(src_ids)}
        blocks.insert(0, {"src_ids": src_ids, "n_dst": len(seeds),
                          "edge_src": np.array([pos[v] for v in es], dtype=np.int64),
                          "edge_dst": np.array(ed, dtype=np.int64), "weights": np.array(ew, dtype=np.float32)})
        seeds = src_ids
    return blocks


def sample_from_item_pairs(heads, tails, neg_tails, item_users, user_items, n_layers, walk_length, restart_prob, num_walks,
                           num_neighbors, seed, step):
    """pinsage/sampler.py:93-106: compact the pair graphs, then the blocks rooted at their nodes."""
    seeds = np.unique(np.concatenate([heads, tails, neg_tails]))
    loc = lambda x: np.searchsorted(seeds, x)
    blocks = build_blocks(seeds, item_users, user_items, n_layers, walk_length, restart_prob, num_walks, num_neighbors, seed,
                          step, heads, tails, neg_tails)
    return {"seeds": seeds, "pos": (loc(heads), loc(tails)), "neg": (loc(heads), loc(neg_tails)), "blocks": blocks}


# ------------------------------------------------------------------------------------------------ model twin
class WeightedSAGEConvRef(nn.Module):
    """pinsage/layers.py:121-156."""

    def __init__(self, input_dims, hidden_dims, output_dims):
        super().__init__()
        self.Q = nn.Linear(input_dims, hidden_dims)
        self.W = nn.Linear(input_dims + hidden_dims, output_dims)
        self.dropout = nn.Dropout(0.5)

    def forward(self, block, h_src, h_dst):
        n = F.relu(self.Q(self.dropout(h_src)))
        es, ed, w = block["edge_src"], block["edge_dst"], block["weights"]
        nd = h_dst.shape[0]
        agg = t.zeros(nd, n.shape[1]).index_add_(0, ed, n[es] * w[:, None])
        ws = t.zeros(nd).index_add_(0, ed, w).unsqueeze(1).clamp(min=1)
        z = F.relu(self.W(self.dropout(t.cat([agg / ws, h_dst], 1))))
        z_norm = z.norm(2, 1, keepdim=True)
        z_norm = t.where(z_norm == 0, t.tensor(1.0), z_norm)
        return z / z_norm


class PinSAGERef(nn.Module):
    """PinSAGEModel (pinsage/model.py:16-34) with the `id` feature only (the reference assigns item ids as the
    trainable feature, pinsage/model.py:52-53): proj = Embedding(n_items + 1, hidden)."""

    def __init__(self, n_items: int, hidden_dims: int, n_layers: int):
        super().__init__()
        self.proj = nn.Embedding(n_items + 1, hidden_dims)
        self.convs = nn.ModuleList([WeightedSAGEConvRef(hidden_dims, hidden_dims, hidden_dims) for _ in range(n_layers)])
        self.bias = nn.Parameter(t.zeros(n_items, 1))

    def get_repr(self, blocks):
        h = self.proj(blocks[0]["src_ids"])
        h_dst_final = self.proj(blocks[-1]["src_ids"][: blocks[-1]["n_dst"]])
        for conv, block in zip(self.convs, blocks):
            h = conv(block, h, h[: block["n_dst"]])
        return h_dst_final + h

    def score(self, h, seeds, pair):
        u, v = pair
        return (h[u] * h[v]).sum(1, keepdim=True) + self.bias[seeds[u]] + self.bias[seeds[v]]

    def forward(self, seeds, pos, neg, blocks):
        h = self.get_repr(blocks)
        return (self.score(h, seeds, neg) - self.score(h, seeds, pos) + 1).clamp(min=0)


def to_torch_blocks(blocks: List[dict]) -> List[dict]:
    return [{"src_ids": t.from_numpy(b["src_ids"]), "n_dst": b["n_dst"], "edge_src": t.from_numpy(b["edge_src"]),
             "edge_dst": t.from_numpy(b["edge_dst"]), "weights": t.from_numpy(b["weights"])} for b in blocks]

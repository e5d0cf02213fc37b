"""ORACLE — test infrastructure only.

Bit-exact numpy mirror of the on-device N-hop sampler (csrc/sampler.hip), i.e. of the algorithm of
the reference's live sampler `GraphDataset.__getitem__` (data/dataset.py:39-309, train mode) with
every random choice drawn from a counter-based Philox stream so that device and host agree exactly:

  positives   samp_cut draws with replacement from the user's list            (data/dataset.py:50-60)
  negatives   fast path: n_neg uniform ids in [0, id_max) when E / n_neg > 100   (:200-209)
              exact path: a uniform n_neg-subset of {0..id_max} minus the sampled positives (:211-230)
  n-hop       frontier caps on both sides (:258-293): a uniform `num_neighbors`-subset of the POSITIONS
              of the concatenated article lists, then a uniform `num_neighbors`-subset of the DISTINCT
              unexplored users of those articles (materialised and cut with Floyd when the lists are
              short; drawn by rejection with acceptance 1/multiplicity when they are long — hub
              articles carry 10^5..10^6 users); hop-0 edges are not emitted by the walk
  relabel     nodes by sorted-unique buckets (:134-150); forward + reverse stores (:164-182)

The reference draws with torch.randint / random.sample / randperm; the law of every draw is the same
(uniform with / without replacement), the streams are not (SURVEY §7 "RNG parity is impossible"), so
this mirror is checked (a) bit for bit against the device and (b) in distribution and, with
randomization off, exactly against the literal restatement in oracle/dataset_ref.py.

Uniform subsets are drawn with Floyd's algorithm:  for j in [L-n, L): t = rand(j+1); pick t unless
already picked, else j.  Defined output order: samples in batch order; within a sample the seed user's
own edges in list order, then hop by hop, users ascending, each user's articles in list order; label
edges: positives in draw order, then negatives in draw order.
"""
from __future__ import annotations

import math
from typing import Dict, List

import numpy as np

from .philox import philox4x32

P_POS, P_NEG, P_ART_CUT, P_USER_CUT, P_NEG_EXACT, P_USER_REJ = 1, 2, 3, 4, 5, 6
REJECT_CAP = 1 << 16  # draws; cannot be reached when reject_min_entries respects its bound


def reject_min_entries(n: int, hops: int) -> int:
    """Smallest total length L of the queued articles' user lists above which the frontier is drawn by
    rejection instead of being materialised.  A user appears at most |queue| <= n times in the lists, so
    there are >= L/n distinct users, of which <= 1 + n*hops are explored: L >= n*(n*(hops+1)+1) leaves
    >= n candidates (termination); 4x that keeps the acceptance rate high."""
    return 4 * n * (n * (hops + 1) + 1)


def philox_words(purpose: int, seed_user: int, i: int, j: int, seed: int, step: int):
    c3 = (purpose & 0xFF) | ((step & 0xFFFFFF) << 8)
    k0, k1 = seed & 0xFFFFFFFF, ((seed >> 32) ^ (step >> 24)) & 0xFFFFFFFF
    return [int(x) for x in philox4x32(i & 0xFFFFFFFF, j & 0xFFFFFFFF, seed_user & 0xFFFFFFFF, c3, k0, k1)]


def rand_below(m: int, purpose: int, seed_user: int, i: int, j: int, seed: int, step: int) -> int:
    """Uniform int in [0, m): one Philox call keyed on (seed, step) and counted by (purpose, user, i, j)."""
    c3 = (purpose & 0xFF) | ((step & 0xFFFFFF) << 8)
    k0, k1 = seed & 0xFFFFFFFF, ((seed >> 32) ^ (step >> 24)) & 0xFFFFFFFF
    r = philox4x32(i & 0xFFFFFFFF, j & 0xFFFFFFFF, seed_user & 0xFFFFFFFF, c3, k0, k1)
    return int(((int(r[0]) << 32) | int(r[1])) % m)


def floyd_subset(L: int, n: int, purpose: int, seed_user: int, i: int, seed: int, step: int) -> List[int]:
    """n distinct positions of [0, L) (all of them when L <= n), ascending."""
    if L <= n:
        return list(range(L))
    picked: List[int] = []
    for j in range(L - n, L):
        t = rand_below(j + 1, purpose, seed_user, i, j, seed, step)
        picked.append(j if t in picked else t)
    return sorted(picked)


def reject_pick_users(article_queue, users: "CsrAdj", articles: "CsrAdj", explored: set, n: int, seed_user: int,
                      hop: int, seed: int, step: int) -> np.ndarray:
    """Uniform n-subset of the distinct unexplored users of the queued articles WITHOUT materialising
    them: draw a position of the concatenated user lists (user v comes up with probability proportional
    to m(v), the number of queued articles it bought, duplicates of the queue counted), accept with
    probability 1/m(v), skip explored / already picked users; draws are taken in counter order t = 0, 1, ..."""
    degs = [len(articles[int(a)]) for a in article_queue]
    pre = np.concatenate([[0], np.cumsum(degs)])
    L = int(pre[-1])
    aq = [int(a) for a in article_queue]
    picked = []
    t = 0
    while len(picked) < n and t < REJECT_CAP:
        w = philox_words(P_USER_REJ, seed_user, t, hop, seed, step)
        pos = ((w[0] << 32) | w[1]) % L
        ai = int(np.searchsorted(pre, pos, side="right") - 1)
        v = int(articles[aq[ai]][pos - int(pre[ai])])
        m = sum(aq.count(int(x)) for x in users[v])
        if w[2] % m == 0 and v not in explored and v not in picked:
            picked.append(v)
        t += 1
    return np.array(sorted(picked), dtype=np.int64)


class CsrAdj:
    def __init__(self, ptr: np.ndarray, idx: np.ndarray):
        self.ptr, self.idx = np.asarray(ptr, dtype=np.int64), np.asarray(idx, dtype=np.int64)

    def __getitem__(self, k: int) -> np.ndarray:
        return self.idx[self.ptr[k]:self.ptr[k + 1]]


def sample_one(u: int, users: CsrAdj, articles: CsrAdj, num_edges: int, id_max: int, cfg, seed: int, step: int,
               randomization: bool = True, cand: "CsrAdj | None" = None) -> Dict[str, np.ndarray]:
    """cand (evaluation mode, data/dataset.py:94-105 with train=False): the matchers' proposals per user; the
    label-0 ids are then the ids occurring exactly once in cat(unique(cand[u]), items of u), ascending."""
    pos = users[u]
    deg = len(pos)
    samp_cut = max(1, math.floor(deg * cfg.positive_edges_ratio))
    if randomization:
        sampled_pos = np.array([pos[rand_below(deg, P_POS, u, i, 0, seed, step)] for i in range(samp_cut)], dtype=np.int64)
    else:
        sampled_pos = np.array([pos[int(np.argmin(pos))], pos[int(np.argmax(pos))]], dtype=np.int64)
    n_pos = len(sampled_pos)
    ratio = cfg.k - 1 if n_pos <= 1 else cfg.negative_edges_ratio
    n_neg = int(ratio * n_pos)
    if cand is not None:
        ids, counts = np.unique(np.concatenate([np.unique(cand[u]), pos]), return_counts=True)
        neg = ids[counts == 1].astype(np.int64)
    elif not randomization:
        neg = np.array([id_max], dtype=np.int64)
    elif n_neg == 0 or num_edges / n_neg > 100:
        neg = np.array([rand_below(id_max, P_NEG, u, i, 0, seed, step) for i in range(n_neg)], dtype=np.int64)
    else:
        banned = np.unique(sampled_pos)
        banned = banned[banned <= id_max]
        M = id_max + 1 - len(banned)
        ranks = floyd_subset(M, n_neg, P_NEG_EXACT, u, 0, seed, step)
        out = []
        for r in ranks:  # rank -> id in the complement of `banned`
            idv = r
            for b in banned:
                if b <= idv:
                    idv += 1
            out.append(idv)
        neg = np.array(out, dtype=np.int64)

    hop_users: List[np.ndarray] = []
    explored = {u}
    queue = np.array([u], dtype=np.int64)
    H = cfg.n_hop_neighbors
    for hop in range(H):
        if len(queue) == 0:
            break
        if hop != 0:
            hop_users.append(queue)
        if hop == H - 1:
            break
        lists = [users[int(q)] for q in queue]
        flat = np.concatenate(lists) if lists else np.empty(0, dtype=np.int64)
        cut = floyd_subset(len(flat), cfg.num_neighbors, P_ART_CUT, u, hop, seed, step)
        article_queue = flat[cut]
        total = sum(len(articles[int(a)]) for a in article_queue)
        rmin = getattr(cfg, "reject_min_entries", None) or reject_min_entries(cfg.num_neighbors, H)
        if total > rmin:
            queue = reject_pick_users(article_queue, users, articles, explored, cfg.num_neighbors, u, hop, seed, step)
        else:
            cand = np.unique(np.concatenate([articles[int(a)] for a in article_queue])) if len(article_queue) else np.empty(0, dtype=np.int64)
            cand = np.array([c for c in cand if c not in explored], dtype=np.int64)
            pick = floyd_subset(len(cand), cfg.num_neighbors, P_USER_CUT, u, hop, seed, step)
            queue = cand[pick]
        explored |= set(queue.tolist())

    sub_u = [np.full(deg, u, dtype=np.int64)] + [np.repeat(q, [len(users[int(x)]) for x in q]) for q in hop_users]
    sub_a = [pos] + [np.concatenate([users[int(x)] for x in q]) if len(q) else np.empty(0, dtype=np.int64) for q in hop_users]
    sub_u, sub_a = np.concatenate(sub_u), np.concatenate(sub_a)
    lab_a = np.concatenate([sampled_pos, neg])
    lab_u = np.full(len(lab_a), u, dtype=np.int64)
    ub = np.unique(np.concatenate([sub_u, lab_u]))
    ab = np.unique(np.concatenate([sub_a, lab_a]))
    return {"user_ids": ub, "article_ids": ab,
            "edge_index": np.stack([np.searchsorted(ub, sub_u), np.searchsorted(ab, sub_a)]),
            "edge_label_index": np.stack([np.searchsorted(ub, lab_u), np.searchsorted(ab, lab_a)]),
            "edge_label": np.concatenate([np.ones(n_pos, dtype=np.int64), np.zeros(len(neg), dtype=np.int64)])}


def sample_batch(seed_users, users: CsrAdj, articles: CsrAdj, num_edges: int, id_max: int, cfg, seed: int, step: int,
                 randomization: bool = True, cand: "CsrAdj | None" = None) -> Dict[str, np.ndarray]:
    """The collated batch (disjoint union, data/data_loader.py:48 + PyG collate) as flat arrays."""
    parts = [sample_one(int(u), users, articles, num_edges, id_max, cfg, seed, step, randomization, cand) for u in seed_users]
    ou = np.cumsum([0] + [len(p["user_ids"]) for p in parts])
    oa = np.cumsum([0] + [len(p["article_ids"]) for p in parts])
    off = lambda i: np.array([[ou[i]], [oa[i]]])
    return {"user_ids": np.concatenate([p["user_ids"] for p in parts]),
            "article_ids": np.concatenate([p["article_ids"] for p in parts]),
            "edge_index": np.concatenate([p["edge_index"] + off(i) for i, p in enumerate(parts)], axis=1),
            "edge_label_index": np.concatenate([p["edge_label_index"] + off(i) for i, p in enumerate(parts)], axis=1),
            "edge_label": np.concatenate([p["edge_label"] for p in parts]),
            "user_ptr": ou, "article_ptr": oa}

"""Seeded synthetic user-item graphs with the shapes SURVEY §8(d) names (C1, C2, C4 shards).

Generation is host-side numpy so that the GPU run and the CPU baseline see the very same edges.
User degrees: clipped log-normal rescaled to the edge budget; item popularity: Zipf(s) over a
seeded permutation of item ids; (user, item) pairs are distinct.
"""
from __future__ import annotations

from dataclasses import dataclass, replace

import numpy as np
import torch as t


@dataclass
class SyntheticSpec:
    num_users: int
    num_items: int
    num_edges: int
    seed: int
    deg_sigma: float = 1.0
    deg_min: int = 1
    deg_max: int = 2000
    zipf_s: float = 1.05
    item_perm_seed: int = 12345  # shared by all shards of one graph: same popular items everywhere
    uniform: bool = False        # i.i.d. uniform endpoints (the locality-free control)
    # Planted structure (round 3, for the MAP@12 leg): users and items belong to `communities` latent groups; a user
    # draws an item from its own group's popularity with probability `community_mix`, from the global popularity
    # otherwise.  0 = none: popularity is then all there is to learn, and no predictor can beat the popularity baseline.
    communities: int = 0
    community_mix: float = 0.8


C1 = SyntheticSpec(943, 1682, 100_000, seed=0, deg_sigma=1.0, deg_min=20, deg_max=737, zipf_s=1.0)
C2 = SyntheticSpec(1_000_000, 100_000, 10_000_000, seed=1)
# C4 (BASELINE.json configs[3]): ONE fixed 100 M-edge graph, C2's generator and user/item ratio at 8x the
# users, defined as the union of C4_BLOCKS independently seeded user blocks so that any rank can generate exactly its
# own users (shards by user_id // ceil(U/N), N in {1,2,4,8}) and every N sees the same graph.
C4 = SyntheticSpec(8_000_000, 100_000, 100_000_000, seed=3)
C4_BLOCKS = 64


def shard_spec(spec: SyntheticSpec, rank: int) -> SyntheticSpec:
    """Spec of rank `rank`'s user shard of a weak-scaled graph: same items, its own users/edges."""
    return replace(spec, seed=spec.seed * 1000 + 17 * rank + (1 if rank else 0))


def item_popularity(spec: SyntheticSpec) -> np.ndarray:
    ranks = np.arange(1, spec.num_items + 1, dtype=np.float64)
    p = ranks ** (-spec.zipf_s)
    p /= p.sum()
    perm = np.random.default_rng(spec.item_perm_seed).permutation(spec.num_items)
    out = np.empty_like(p)
    out[perm] = p
    return out


def item_community(spec: SyntheticSpec) -> np.ndarray:
    """Group of every item: popularity rank mod communities, so that every group has the same popularity profile."""
    order = np.argsort(-item_popularity(spec), kind="stable")
    out = np.empty(spec.num_items, dtype=np.int64)
    out[order] = np.arange(spec.num_items) % max(spec.communities, 1)
    return out


def user_community(spec: SyntheticSpec) -> np.ndarray:
    return np.random.default_rng(spec.seed + 31337).integers(0, max(spec.communities, 1), spec.num_users)


class ItemSampler:
    """Draws items for given users from the spec's law: the global popularity, or the planted mixture."""

    def __init__(self, spec: SyntheticSpec):
        self.spec = spec
        p = item_popularity(spec)
        self.cdf = np.cumsum(p)
        self.cdf[-1] = 1.0
        self.groups = None
        if spec.communities > 1:
            ic = item_community(spec)
            self.uc = user_community(spec)
            self.groups = []
            for k in range(spec.communities):
                ids = np.nonzero(ic == k)[0]
                c = np.cumsum(p[ids])
                c /= c[-1]
                self.groups.append((ids, c))

    def draw(self, users: np.ndarray, rng: np.random.Generator) -> np.ndarray:
        I = self.spec.num_items
        i = np.minimum(np.searchsorted(self.cdf, rng.random(users.size), side="right"), I - 1).astype(np.int64)
        if self.groups is not None:
            own = rng.random(users.size) < self.spec.community_mix
            cu = self.uc[users]
            r = rng.random(users.size)
            for k, (ids, c) in enumerate(self.groups):
                sel = np.nonzero(own & (cu == k))[0]
                if sel.size:
                    i[sel] = ids[np.minimum(np.searchsorted(c, r[sel], side="right"), ids.size - 1)]
        return i


def user_degrees(spec: SyntheticSpec, rng: np.random.Generator) -> np.ndarray:
    mean = spec.num_edges / spec.num_users
    mu = np.log(max(mean, 1.0)) - 0.5 * spec.deg_sigma ** 2
    d = rng.lognormal(mu, spec.deg_sigma, spec.num_users)
    d = np.clip(d, spec.deg_min, spec.deg_max)
    d = d * (spec.num_edges / d.sum())
    return np.clip(np.rint(d), spec.deg_min, min(spec.deg_max, spec.num_items)).astype(np.int64)


def generate(spec: SyntheticSpec) -> t.Tensor:
    """Returns edge_index int64 [2, E]: row 0 = user id in [0, U), row 1 = item id in [0, I)."""
    rng = np.random.default_rng(spec.seed)
    U, I, E = spec.num_users, spec.num_items, spec.num_edges
    if E > U * I:
        raise ValueError("more edges requested than distinct pairs exist")
    if spec.uniform:
        keys = np.empty(0, dtype=np.int64)
        while keys.size < E:
            keys = np.unique(np.concatenate([keys, rng.integers(0, U * I, size=int((E - keys.size) * 1.1) + 16)]))
        keys = keys[rng.permutation(keys.size)[:E]]
        return t.from_numpy(np.stack([keys // I, keys % I]))
    sampler = ItemSampler(spec)
    deg = user_degrees(spec, rng)
    keys = np.empty(0, dtype=np.int64)
    have = np.zeros(U, dtype=np.int64)
    for _ in range(16):
        deficit = np.maximum(deg - have, 0)
        if keys.size >= E and deficit.sum() == 0:
            break
        n_draw = np.where(deficit > 0, np.ceil(deficit * 1.25).astype(np.int64) + 1, 0)
        if keys.size < E and n_draw.sum() == 0:  # degrees met but edge budget not: spread the rest
            extra = rng.integers(0, U, size=int((E - keys.size) * 1.2) + 16)
            n_draw = np.bincount(extra, minlength=U)
        u = np.repeat(np.arange(U, dtype=np.int64), n_draw)
        i = sampler.draw(u, rng)
        keys = np.unique(np.concatenate([keys, u * I + i]))
        have = np.bincount(keys // I, minlength=U)
    while keys.size < E:  # dense small graphs: the Zipf head saturates; fill up with uniform pairs
        extra = rng.integers(0, U * I, size=int((E - keys.size) * 1.5) + 16)
        keys = np.unique(np.concatenate([keys, extra]))
    if keys.size > E:  # trim uniformly, but never a user's only edge
        u_of = keys // I
        first = np.ones(keys.size, dtype=bool)
        first[1:] = u_of[1:] != u_of[:-1]
        prio = rng.random(keys.size)
        prio[first] = -1.0
        keep = np.argpartition(prio, E - 1)[:E]
        keys = keys[keep]
    keys = keys[rng.permutation(keys.size)]  # edge order carries no structure (a raw transaction log)
    return t.from_numpy(np.stack([keys // I, keys % I]))


def block_spec(spec: SyntheticSpec, n_blocks: int, b: int) -> SyntheticSpec:
    """Spec of user block b of a graph defined block-wise: U/n_blocks users, E/n_blocks edges, its own seed, the
    shared item popularity."""
    if spec.num_users % n_blocks or spec.num_edges % n_blocks:
        raise ValueError("users and edges must divide into the blocks")
    if not 0 <= b < n_blocks:
        raise ValueError("block index out of range")
    return replace(spec, num_users=spec.num_users // n_blocks, num_edges=spec.num_edges // n_blocks,
                   seed=spec.seed * 100_003 + b)


def generate_blocks(spec: SyntheticSpec, n_blocks: int, b0: int, b1: int, workers: int = 8) -> t.Tensor:
    """Edges of user blocks [b0, b1) of the block-wise graph; user ids are LOCAL to the range (block b0's first user
    is 0), i.e. global id = local id + b0 * (U / n_blocks).  generate_blocks(spec, n, 0, n) is the whole graph."""
    per = spec.num_users // n_blocks

    def one(b: int) -> t.Tensor:
        ei = generate(block_spec(spec, n_blocks, b))
        ei[0] += (b - b0) * per
        return ei

    if b1 - b0 > 1 and workers > 1:  # numpy's sort / searchsorted release the GIL
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(workers, b1 - b0)) as ex:
            parts = list(ex.map(one, range(b0, b1)))
    else:
        parts = [one(b) for b in range(b0, b1)]
    return t.cat(parts, dim=1) if parts else t.empty(2, 0, dtype=t.int64)


def shard_blocks(n_blocks: int, world: int, rank: int):
    """Block range of rank `rank`: the shard user_id // ceil(U / world) of SURVEY 8d (world divides n_blocks)."""
    if n_blocks % world:
        raise ValueError("the number of ranks must divide the number of blocks")
    per = n_blocks // world
    return rank * per, (rank + 1) * per


def heldout_edges(spec: SyntheticSpec, ei: t.Tensor, n_eval: int, seed: int = 99) -> t.Tensor:
    """One extra (user, item) pair for n_eval distinct users, drawn from the generator's item popularity and not
    among the user's edges in `ei`: the held-out positives a MAP@12 is scored on.  int64 [2, n], users ascending."""
    rng = np.random.default_rng(seed)
    U, I = spec.num_users, spec.num_items
    sampler = ItemSampler(spec)   # the same law the graph was drawn from (planted structure included)
    keys = np.sort(ei[0].numpy() * I + ei[1].numpy())
    users = np.sort(rng.choice(U, size=min(n_eval, U), replace=False)).astype(np.int64)
    item = np.full(users.size, -1, dtype=np.int64)
    for _ in range(64):
        todo = np.nonzero(item < 0)[0]
        if todo.size == 0:
            break
        cand = sampler.draw(users[todo], rng)
        k = users[todo] * I + cand
        pos = np.minimum(np.searchsorted(keys, k), keys.size - 1)
        fresh = keys[pos] != k
        item[todo[fresh]] = cand[fresh]
    ok = item >= 0
    return t.from_numpy(np.stack([users[ok], item[ok]]))


# H&M-shaped heterogeneous graph: integer categorical node features, customer -buys-> article
HM_CUSTOMER_CARDS = (352_899, 2, 84, 4, 5, 2)   # postal_code, FN, age, club_member_status, fashion_news_frequency, Active
HM_ARTICLE_CARDS = (47_224, 132, 30, 50)        # product_code, product_type_no, graphical_appearance_no, colour_group_code
C3 = SyntheticSpec(1_371_980, 105_542, 31_800_000, seed=2, deg_max=2000, zipf_s=1.0)


def generate_hetero(spec: SyntheticSpec, customer_cards=HM_CUSTOMER_CARDS, article_cards=HM_ARTICLE_CARDS,
                    feature_signal: bool = False):
    """Returns (graph, users_adj, articles_adj) in the shapes the reference's preprocessing writes
    (train_graph.pt / edges_train.pt / rev_edges_train.pt; run_preprocessing.py:176-195): a HeteroData with
    int64 categorical `x` per node type and the `buys` edge_index, plus the two adjacency lists (as CSR
    AdjList objects — a dict of Python lists does not scale to 31.8 M edges).
    feature_signal (with spec.communities > 1): the features say something about the planted structure, as a real
    dataset's do — the first customer column and the first article column wide enough hold the node's latent group
    (the other columns stay noise).  Without it the ranker has nothing but graph structure to learn from."""
    from .data.dataset import AdjList
    from .hetero import HeteroData
    from .utils.constants import Constants
    ei = generate(spec)
    rng = np.random.default_rng(spec.seed + 7919)
    cx = np.stack([rng.integers(0, min(c, spec.num_users) if c > 1000 else c, size=spec.num_users) for c in customer_cards], 1)
    ax = np.stack([rng.integers(0, min(c, spec.num_items) if c > 1000 else c, size=spec.num_items) for c in article_cards], 1)
    if feature_signal and spec.communities > 1:
        for x, cards, group in ((cx, customer_cards, user_community(spec)), (ax, article_cards, item_community(spec))):
            wide = [j for j, c in enumerate(cards) if spec.communities <= c <= 1000]
            if not wide:
                raise ValueError("feature_signal needs a categorical column with at least `communities` values")
            x[:, wide[0]] = group
    g = HeteroData()
    g[Constants.node_user].x = t.from_numpy(cx.astype(np.int64))
    g[Constants.node_item].x = t.from_numpy(ax.astype(np.int64))
    g[Constants.edge_key].edge_index = ei
    u, a = ei[0].numpy(), ei[1].numpy()
    return g, AdjList.from_edges(u, a, spec.num_users), AdjList.from_edges(a, u, spec.num_items)

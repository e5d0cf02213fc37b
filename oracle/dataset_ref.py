"""ORACLE — test infrastructure only.  Literal restatement of the reference's live N-hop sampler,
`GraphDataset.__getitem__` and helpers (data/dataset.py:39-309), on plain dicts / sets / lists with
Python's `random` and torch RNG exactly where the reference uses them.  Returns a plain dict instead
of a PyG HeteroData.  Pinned by the reference's tests/test_dataset.py fixture (graph
[[0,0,0,1,1,2,2],[0,2,4,1,5,3,0]], randomization=False), whose expected subgraph is re-derived in
tests/test_ranker_cpu.py."""
import math
import random
from typing import Dict, List, Optional

import torch as t
from torch import Tensor


def flatten(l):  # utils/flatten.py
    return [item for sub in l for item in sub]


def create_edges_from_target_indices(source_index: int, target_indices: Tensor) -> Tensor:  # :243-255
    return t.stack([t.tensor([source_index], dtype=t.long).repeat(len(target_indices)),
                    t.as_tensor(target_indices, dtype=t.long)], dim=0)


def only_items_with_count_one(x: Tensor) -> Tensor:  # :184-186
    uniques, counts = x.unique(return_counts=True)
    return uniques[counts == 1]


def get_negative_edges_random(subgraph_edges_to_filter, all_edges, num_negative_edges, randomization):  # :189-230
    id_max = t.max(all_edges, dim=1)[0][1]
    if all_edges.shape[1] / num_negative_edges > 100:
        if randomization:
            return t.randint(low=0, high=id_max.item(), size=(num_negative_edges,))
        return t.tensor([id_max.item()])
    only_negative = only_items_with_count_one(t.cat((t.arange(0, id_max + 1, dtype=t.int64), subgraph_edges_to_filter)))
    if randomization:
        return only_negative[t.randperm(only_negative.nelement())][:num_negative_edges]
    return t.tensor([id_max.item()])


def shuffle_and_cut(array: list, n: int) -> list:  # :289-293
    return random.sample(array, n) if len(array) > n else array


def fetch_n_hop_neighbourhood(n, user_id, users: dict, articles: dict, num_neighbors) -> Tensor:  # :258-286
    accum = t.tensor([[], []], dtype=t.long)
    explored = set()
    queue = set([user_id])
    for i in range(n):
        new = [(users[u], create_edges_from_target_indices(u, t.as_tensor(users[u], dtype=t.long))) for u in queue]
        explored = explored | queue
        if len(new) == 0:
            break
        new_articles = flatten([x[0] for x in new])
        if i != 0:
            accum = t.cat([accum, t.cat([x[1] for x in new], dim=1)], dim=1)
        articles_queue = shuffle_and_cut(new_articles, num_neighbors)
        new_users = set(flatten([articles[a] for a in articles_queue])) - explored
        queue = set(shuffle_and_cut(list(new_users), num_neighbors))
    return accum


def get_item(idx: int, graph: dict, users: Dict[int, List[int]], articles: Dict[int, List[int]], config, train: bool,
             matchers: Optional[list] = None, randomization: bool = True) -> dict:
    """graph: {"user_x", "article_x", "edge_index"}.  Mirrors __getitem__ :39-182."""
    all_edges = graph["edge_index"]
    pos = t.as_tensor(users[idx], dtype=t.long)
    pos_edges = create_edges_from_target_indices(idx, pos)
    samp_cut = max(1, math.floor(len(pos) * config.positive_edges_ratio))
    if randomization:
        ri = t.randint(low=0, high=len(pos), size=(samp_cut,))
    else:
        ri = t.tensor([t.min(pos, dim=0)[1].item(), t.max(pos, dim=0)[1].item()])
    sampled_pos = pos[ri]
    sampled_pos_edges = create_edges_from_target_indices(idx, sampled_pos)
    n_pos = sampled_pos.shape[0]
    ratio = config.k - 1 if n_pos <= 1 else config.negative_edges_ratio
    if train:
        neg = get_negative_edges_random(sampled_pos, all_edges, int(ratio * n_pos), randomization)
    else:
        cand = t.cat([m.get_matches(idx) for m in matchers], dim=0).unique()
        neg = only_items_with_count_one(t.cat([cand, pos], dim=0))
    neg_edges = create_edges_from_target_indices(idx, neg)
    hop = fetch_n_hop_neighbourhood(config.n_hop_neighbors, idx, users, articles, config.num_neighbors)
    touched = t.cat([pos_edges, neg_edges, hop], dim=1)
    subgraph = t.cat([pos_edges, hop], dim=1)
    ub = t.unique(touched[0], sorted=True)
    ab = t.unique(touched[1], sorted=True)
    remap = lambda e: t.stack((t.bucketize(e[0], ub), t.bucketize(e[1], ab)))
    sampled = t.cat([sampled_pos_edges, neg_edges], dim=1)
    labels = t.cat([t.ones(sampled_pos_edges.shape[1]), t.zeros(neg_edges.shape[1])]).type(t.long)
    return {"user_x": graph["user_x"][ub], "article_x": graph["article_x"][ab],
            "edge_index": remap(subgraph).type(t.long), "edge_label_index": remap(sampled).type(t.long),
            "edge_label": labels}

"""The reference's two hand-made test graphs (/root/reference/tests/data_generator.py:129-157, "random" and "star") as
data, and what its sampler (data/dataset.py:39-182,258-286) returns for every seed user of each in
randomization=False mode, derived BY HAND (tests/test_dataset.py:25-92 of the reference checks user 0 of "random").
Shared by the CPU test of the host GraphDataset and the GPU test of the device sampler."""
from types import SimpleNamespace

import torch as t

# tests/util.py:17-50 of the reference: ratio 0.5, negative ratio 1.0, 2 hops, fan-out 64, k 12
CFG = SimpleNamespace(k=12, num_neighbors=64, n_hop_neighbors=2, positive_edges_ratio=0.5, negative_edges_ratio=1.0,
                      batch_size=1)


def fixture(kind):
    from laplace_amd.hetero import HeteroData
    from laplace_amd.utils.constants import Constants
    if kind == "random":
        ux = t.tensor([[0.0, 0.1], [1.0, 1.1], [2.0, 2.1]])
        ax = t.tensor([[i + j / 10 for j in range(5)] for i in range(6)])
        ei = t.tensor([[0, 0, 0, 1, 1, 2, 2], [0, 2, 4, 1, 5, 3, 0]])
    else:
        ux = t.tensor([[float(f"{i}.{j}") for j in range(6)] for i in range(5)])
        ax = t.tensor([[float(f"{i}.{j}") for j in range(4)] for i in range(4)])
        ei = t.tensor([[0, 0, 0, 0, 1, 2, 3, 4], [0, 1, 2, 3, 0, 1, 2, 3]])
    g = HeteroData()
    g[Constants.node_user].x, g[Constants.node_item].x, g[Constants.edge_key].edge_index = ux, ax, ei
    users, articles = {}, {}
    for u, a in zip(*ei.tolist()):  # get_edge_dicts: groupby keeps edge order (tests/util.py:119-143)
        users.setdefault(u, []).append(a)
        articles.setdefault(a, []).append(u)
    return g, users, articles, ux, ax, ei


# Hand-derived per seed user: (user buckets, article buckets, message-passing edges in ORIGINAL ids,
# label edges in original ids (user, article), labels).  Derivation, data/dataset.py of the reference:
#   positives  = [article at argmin, article at argmax] of the user's list (:58-64) — a one-article user gets it twice;
#   negatives  = [max article id of the graph], once, whatever was asked for (:200-203 / :228-229), even when that
#                article is one of the user's own (star: article 3 of user 0) — E / n_neg <= 100 on both graphs;
#   edges      = the user's own edges (:110-116) + every edge of the users met within n_hop-1 article hops (:258-286);
#   buckets    = sorted unique ids of everything touched, label edges included (:118-123).
EXPECT = {
    "random": {
        0: ([0, 2], [0, 2, 3, 4, 5], [(0, 0), (0, 2), (0, 4), (2, 3), (2, 0)], [(0, 0), (0, 4), (0, 5)], [1, 1, 0]),
        1: ([1], [1, 5], [(1, 1), (1, 5)], [(1, 1), (1, 5), (1, 5)], [1, 1, 0]),
        2: ([0, 2], [0, 2, 3, 4, 5], [(2, 3), (2, 0), (0, 0), (0, 2), (0, 4)], [(2, 0), (2, 3), (2, 5)], [1, 1, 0]),
    },
    "star": {
        0: ([0, 1, 2, 3, 4], [0, 1, 2, 3],
            [(0, 0), (0, 1), (0, 2), (0, 3), (1, 0), (2, 1), (3, 2), (4, 3)], [(0, 0), (0, 3), (0, 3)], [1, 1, 0]),
        1: ([0, 1], [0, 1, 2, 3], [(1, 0), (0, 0), (0, 1), (0, 2), (0, 3)], [(1, 0), (1, 0), (1, 3)], [1, 1, 0]),
        2: ([0, 2], [0, 1, 2, 3], [(2, 1), (0, 0), (0, 1), (0, 2), (0, 3)], [(2, 1), (2, 1), (2, 3)], [1, 1, 0]),
        3: ([0, 3], [0, 1, 2, 3], [(3, 2), (0, 0), (0, 1), (0, 2), (0, 3)], [(3, 2), (3, 2), (3, 3)], [1, 1, 0]),
        4: ([0, 4], [0, 1, 2, 3], [(4, 3), (0, 0), (0, 1), (0, 2), (0, 3)], [(4, 3), (4, 3), (4, 3)], [1, 1, 0]),
    },
}



"""CPU tests of the ranker's host logic and of its oracle: the sampler against the reference's
tests/test_dataset.py fixture and against the literal restatement, collate, feature info,
scatter known answers, golden-pinned helpers."""
import os
import random
from types import SimpleNamespace

import numpy as np
import pytest
import torch as t

from oracle import dataset_ref as DR
from oracle import ranker_ref as RR


def _cfg(**kw):
    base = dict(k=12, num_neighbors=64, n_hop_neighbors=2, positive_edges_ratio=0.5, negative_edges_ratio=1.0)
    base.update(kw)
    return SimpleNamespace(**base)


def _fixture_graph():
    """tests/data_generator.py:129-146 of the reference ("random" = this fixed graph)."""
    from laplace_amd.hetero import HeteroData
    from laplace_amd.utils.constants import Constants
    ux = t.tensor([[0.0, 0.1], [1.0, 1.1], [2.0, 2.1]])
    ax = t.tensor([[i + j / 10 for j in range(5)] for i in range(6)])
    ei = t.tensor([[0, 0, 0, 1, 1, 2, 2], [0, 2, 4, 1, 5, 3, 0]])
    g = HeteroData()
    g[Constants.node_user].x, g[Constants.node_item].x, g[Constants.edge_key].edge_index = ux, ax, ei
    users = {0: [0, 2, 4], 1: [1, 5], 2: [3, 0]}
    articles = {0: [0, 2], 1: [1], 2: [0], 3: [2], 4: [0], 5: [1]}
    return g, users, articles, ux, ax, ei


def test_sampler_reference_fixture():
    """Expected values derived by hand from the fixture (SURVEY §8c iii): user 0, positives [0, 4]
    (argmin/argmax), negative [5] (= id_max: 7 edges / 2 < 100 -> exact branch), labels [1, 1, 0],
    message-passing edges = all edges of users {0, 2}, nodes relabelled by sorted-unique buckets."""
    from laplace_amd.data.dataset import GraphDataset
    from laplace_amd.utils.constants import Constants
    g, users, articles, ux, ax, ei = _fixture_graph()
    ds = GraphDataset(_cfg(), g, users, articles, train=True, randomization=False)
    assert len(ds) == 3
    item = ds[0]
    user_buckets, article_buckets = t.tensor([0, 2]), t.tensor([0, 2, 3, 4, 5])
    assert t.equal(item[Constants.node_user].x, ux[user_buckets])
    assert t.equal(item[Constants.node_item].x, ax[article_buckets])
    store = item[Constants.edge_key]
    want_edges = sorted([(0, 0), (0, 1), (0, 3), (1, 2), (1, 0)])  # (0,0)(0,2)(0,4)(2,3)(2,0) relabelled
    assert sorted(zip(store.edge_index[0].tolist(), store.edge_index[1].tolist())) == want_edges
    assert store.edge_label_index.tolist() == [[0, 0, 0], [0, 3, 4]]  # articles 0, 4, 5
    assert store.edge_label.tolist() == [1, 1, 0] and store.edge_label.dtype == t.long
    rev = item[Constants.rev_edge_key]
    assert t.equal(rev.edge_index, store.edge_index.flip(0)) and t.equal(rev.edge_label_index, store.edge_label_index.flip(0))
    assert t.equal(rev.edge_label, store.edge_label)
    # the literal restatement of the reference agrees
    ref = DR.get_item(0, {"user_x": ux, "article_x": ax, "edge_index": ei}, users, articles, _cfg(), True, None, False)
    assert t.equal(ref["user_x"], item[Constants.node_user].x) and t.equal(ref["article_x"], item[Constants.node_item].x)
    assert sorted(zip(*ref["edge_index"].tolist())) == want_edges
    assert t.equal(ref["edge_label_index"], store.edge_label_index) and t.equal(ref["edge_label"], store.edge_label)


@pytest.mark.parametrize("kind", ["random", "star"])
def test_host_sampler_on_both_reference_fixtures_every_seed_user(kind):
    """tests/reference_fixtures.py: hand-derived samples of every user of the reference's "random" and "star" graphs
    (the star's hub user, one-article users that yield the same positive twice, a negative that is a positive)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from reference_fixtures import CFG, EXPECT, fixture
    from laplace_amd.data.dataset import GraphDataset
    from laplace_amd.utils.constants import Constants
    g, users, articles, ux, ax, ei = fixture(kind)
    ds = GraphDataset(CFG, g, users, articles, train=True, randomization=False)
    assert len(ds) == ux.shape[0] == len(EXPECT[kind])
    for s, (ub, ab, edges, label_edges, labels) in EXPECT[kind].items():
        item = ds[s]
        upos, apos = {u: i for i, u in enumerate(ub)}, {a: i for i, a in enumerate(ab)}
        assert t.equal(item[Constants.node_user].x, ux[ub]) and t.equal(item[Constants.node_item].x, ax[ab])
        store = item[Constants.edge_key]
        assert sorted(zip(*store.edge_index.tolist())) == sorted((upos[u], apos[a]) for u, a in edges)
        assert store.edge_label_index.tolist() == [[upos[u] for u, _ in label_edges], [apos[a] for _, a in label_edges]]
        assert store.edge_label.tolist() == labels
        ref = DR.get_item(s, {"user_x": ux, "article_x": ax, "edge_index": ei}, users, articles, CFG, True, None, False)
        assert t.equal(ref["edge_label_index"], store.edge_label_index) and t.equal(ref["edge_label"], store.edge_label)
        assert sorted(zip(*ref["edge_index"].tolist())) == sorted(zip(*store.edge_index.tolist()))


def _random_graph(seed, U=40, A=30, E=260):
    from laplace_amd.hetero import HeteroData
    from laplace_amd.utils.constants import Constants
    g = t.Generator().manual_seed(seed)
    keys = t.randperm(U * A, generator=g)[:E]
    ei = t.stack([keys // A, keys % A])
    for u in range(U):  # every user needs at least one edge (the reference indexes users[idx])
        if not (ei[0] == u).any():
            ei = t.cat([ei, t.tensor([[u], [int(t.randint(0, A, (1,), generator=g))]])], dim=1)
    users = {u: ei[1][ei[0] == u].tolist() for u in range(U)}
    articles = {a: ei[0][ei[1] == a].tolist() for a in range(A) if (ei[1] == a).any()}
    for a in range(A):
        articles.setdefault(a, [])
    hd = HeteroData()
    hd[Constants.node_user].x = t.randint(0, 50, (U, 3), generator=g)
    hd[Constants.node_item].x = t.randint(0, 9, (A, 2), generator=g)
    hd[Constants.edge_key].edge_index = ei
    return hd, users, articles, ei


@pytest.mark.parametrize("hops,fan", [(1, 64), (2, 64), (3, 1000), (4, 1000)])  # caps that bite are random even upstream
def test_sampler_equals_literal_restatement_deterministic_mode(hops, fan):
    from laplace_amd.data.dataset import GraphDataset
    from laplace_amd.utils.constants import Constants
    hd, users, articles, ei = _random_graph(seed=hops)
    cfg = _cfg(n_hop_neighbors=hops, num_neighbors=fan)
    ds = GraphDataset(cfg, hd, users, articles, train=True, randomization=False)
    graph = {"user_x": hd[Constants.node_user].x, "article_x": hd[Constants.node_item].x, "edge_index": ei}
    for idx in range(len(ds)):
        got, ref = ds[idx], DR.get_item(idx, graph, users, articles, cfg, True, None, False)
        s = got[Constants.edge_key]
        assert t.equal(got[Constants.node_user].x, ref["user_x"]) and t.equal(got[Constants.node_item].x, ref["article_x"])
        assert sorted(zip(*s.edge_index.tolist())) == sorted(zip(*ref["edge_index"].tolist()))
        assert t.equal(s.edge_label_index, ref["edge_label_index"]) and t.equal(s.edge_label, ref["edge_label"])


def test_sampler_random_mode_structure_and_eval_negatives():
    from laplace_amd.data.dataset import GraphDataset
    from laplace_amd.utils.constants import Constants
    hd, users, articles, ei = _random_graph(seed=9, U=60, A=50, E=700)
    cfg = _cfg(n_hop_neighbors=2, num_neighbors=5, positive_edges_ratio=0.5, negative_edges_ratio=3.0)
    ds = GraphDataset(cfg, hd, users, articles, train=True, randomization=True, seed=1)
    pairs = set(zip(ei[0].tolist(), ei[1].tolist()))
    for idx in range(0, 60, 7):
        item = ds[idx]
        s = item[Constants.edge_key]
        ub = t.unique(t.cat([s.edge_index[0], s.edge_label_index[0]]))
        assert item[Constants.node_user].x.shape[0] == ub.numel()  # every feature row is a touched node
        n_pos = int(s.edge_label.sum())
        assert n_pos == max(1, len(users[idx]) // 2)
        n_neg = s.edge_label.numel() - n_pos
        assert n_neg == (cfg.k - 1 if n_pos <= 1 else int(3.0 * n_pos))
        # fan-out cap: <= 5 articles expanded, <= 5 new users, all their edges are real edges
        x_u = item[Constants.node_user].x
        assert x_u.shape[0] <= 1 + 5
    # eval: negatives = ids seen exactly once in cat(unique candidates, positives) — as written upstream
    class M:
        def get_matches(self, u):
            return t.tensor([1, 2, 3, 3, 7])
    dse = GraphDataset(cfg, hd, users, articles, train=False, matchers=[M()], randomization=True, seed=2)
    item = dse[3]
    s = item[Constants.edge_key]
    pos = users[3]
    ids, cnt = np.unique(np.concatenate([np.array([1, 2, 3, 7]), np.array(pos)]), return_counts=True)
    want_neg = ids[cnt == 1]
    ab = np.unique(np.concatenate([ei[1][np.isin(ei[0], np.unique(s.edge_index[0]))].numpy(), want_neg]))
    assert int((s.edge_label == 0).sum()) == len(want_neg)


def test_collate_offsets_and_loader():
    from laplace_amd.data.dataset import GraphDataset
    from laplace_amd.hetero import DataLoader, collate
    from laplace_amd.utils.constants import Constants
    hd, users, articles, ei = _random_graph(seed=4)
    ds = GraphDataset(_cfg(), hd, users, articles, train=True, randomization=False)
    a, b = ds[0], ds[1]
    batch = collate([a, b])
    nu, na = a[Constants.node_user].x.shape[0], a[Constants.node_item].x.shape[0]
    assert t.equal(batch[Constants.node_user].x, t.cat([a[Constants.node_user].x, b[Constants.node_user].x]))
    eb = batch[Constants.edge_key]
    ea, eb2 = a[Constants.edge_key], b[Constants.edge_key]
    assert t.equal(eb.edge_index[:, :ea.edge_index.shape[1]], ea.edge_index)
    assert t.equal(eb.edge_index[:, ea.edge_index.shape[1]:], eb2.edge_index + t.tensor([[nu], [na]]))
    assert t.equal(eb.edge_label_index[:, ea.edge_label_index.shape[1]:], eb2.edge_label_index + t.tensor([[nu], [na]]))
    rb = batch[Constants.rev_edge_key]
    assert t.equal(rb.edge_index, eb.edge_index.flip(0))  # rev store offsets swap with its node types
    assert t.equal(eb.edge_label, t.cat([ea.edge_label, eb2.edge_label]))
    assert batch.metadata() == ([Constants.node_user, Constants.node_item], [Constants.edge_key, Constants.rev_edge_key])
    loader = DataLoader(ds, batch_size=16, shuffle=True, generator=t.Generator().manual_seed(0))
    sizes = [bt[Constants.node_user].x.shape[0] for bt in loader]
    assert len(loader) == 3 and len(sizes) == 3


def test_feature_info_and_embedding_widths():
    from laplace_amd.utils.get_info import embedding_size_for, get_feature_info, select_properties
    assert [embedding_size_for(m) for m in (1, 2, 3, 10, 11, 1000, 9999, 10_000, 99_999, 1_000_000, 2_000_000)] == \
        [2, 2, 4, 4, 12, 12, 20, 20, 40, 60, 20]  # > 1M falls back to the "10000" width (Appendix A.9)
    hd, users, articles, ei = _random_graph(seed=5)
    info = get_feature_info(hd)
    assert info["customer"].num_feat == 3 and info["article"].num_feat == 2
    assert info["customer"].num_cat == hd["customer"].x.max(0)[0].tolist()


def test_scatter_known_answers():
    """SAGEConv aggregation on a 3 x 4 bipartite graph by explicit loops (sum / mean / max; empty dst -> 0)."""
    x = t.tensor([[1.0, -2.0], [3.0, 5.0], [-4.0, 0.5]])
    src, dst = t.tensor([0, 1, 2, 0, 1]), t.tensor([0, 0, 0, 2, 2])
    want = {"add": [[0.0, 3.5], [0, 0], [4.0, 3.0], [0, 0]], "mean": [[0.0, 3.5 / 3], [0, 0], [2.0, 1.5], [0, 0]],
            "max": [[3.0, 5.0], [0, 0], [3.0, 5.0], [0, 0]]}
    for aggr, w in want.items():
        got = RR.scatter_aggr(x[src], dst, 4, aggr)
        assert t.allclose(got, t.tensor(w)), aggr


def test_linear_layers_match_reference_seeded(golden_dir):
    """Same module structure AND the same seeded initial weights as the reference's get_linear_layers."""
    from laplace_amd.model.layers import get_linear_layers
    g = t.load(os.path.join(golden_dir, "linear_layers.pt"), weights_only=False)
    for (n, i, h, o), want in g.items():
        t.manual_seed(23)
        ml = get_linear_layers(n, i, h, o)
        assert [(m.in_features, m.out_features, m.bias is not None) for m in ml] == want["shapes"]
        for m, st in zip(ml, want["state"]):
            assert t.equal(m.weight.detach(), st["weight"]) and t.equal(m.bias.detach(), st["bias"])


def test_metrics_universal_matches_reference(golden_dir):
    from laplace_amd.utils.metrics_encoder_decoder import get_metrics_universal
    g = t.load(os.path.join(golden_dir, "metrics_universal.pt"), weights_only=False)
    got = get_metrics_universal(g["model_output"].clone(), g["edge_index"], g["edge_label_index"], [], k=g["k"])
    assert got == pytest.approx(g["metrics"], abs=1e-7)


def test_sage_layers_factory_surface():
    from laplace_amd.model.layers import SAGEConv, get_SAGEConv_layers
    ml = get_SAGEConv_layers(3, 128, 64, "add")
    assert len(ml) == 3 and all(isinstance(m, SAGEConv) for m in ml)
    assert [m.out_channels for m in ml] == [128, 128, 64] and ml[-1].aggr == "add"
    assert ml[0].lin_l.weight is None  # lazily sized like SAGEConv((-1, -1, -1), ...)
    assert len(get_SAGEConv_layers(1, 128, 64, "mean")) == 1
    with pytest.raises(ValueError):
        get_SAGEConv_layers(2, 8, 8, "lstm")


# ---------------------------------------------------------------------------- N1 mirror (oracle/sampler_ref.py)
def _csr_of(adj: dict, n: int):
    from oracle.sampler_ref import CsrAdj
    ptr = np.zeros(n + 1, dtype=np.int64)
    for k, v in adj.items():
        ptr[k + 1] = len(v)
    ptr = np.cumsum(ptr)
    idx = np.concatenate([np.asarray(adj[k], dtype=np.int64) for k in range(n)]) if n else np.empty(0, dtype=np.int64)
    return CsrAdj(ptr, idx)


@pytest.mark.parametrize("hops", [1, 2, 3])
def test_device_sampler_mirror_equals_literal_restatement_deterministic_mode(hops):
    """With randomization off and caps that do not bite, the Philox mirror of the device sampler and the
    literal restatement of the reference produce the same subgraph for every user."""
    from oracle import sampler_ref as SR
    from laplace_amd.utils.constants import Constants
    hd, users, articles, ei = _random_graph(seed=20 + hops)
    cfg = _cfg(n_hop_neighbors=hops, num_neighbors=1000)
    U, A = hd[Constants.node_user].x.shape[0], hd[Constants.node_item].x.shape[0]
    ucsr, acsr = _csr_of(users, U), _csr_of(articles, A)
    graph = {"user_x": hd[Constants.node_user].x, "article_x": hd[Constants.node_item].x, "edge_index": ei}
    for idx in range(U):
        got = SR.sample_one(idx, ucsr, acsr, ei.shape[1], int(ei[1].max()), cfg, seed=1, step=0, randomization=False)
        ref = DR.get_item(idx, graph, users, articles, cfg, True, None, False)
        assert t.equal(hd[Constants.node_user].x[got["user_ids"]], ref["user_x"])
        assert t.equal(hd[Constants.node_item].x[got["article_ids"]], ref["article_x"])
        assert sorted(zip(*got["edge_index"].tolist())) == sorted(zip(*ref["edge_index"].tolist()))
        assert got["edge_label_index"].tolist() == ref["edge_label_index"].tolist()
        assert got["edge_label"].tolist() == ref["edge_label"].tolist()


def test_device_sampler_mirror_random_mode_laws():
    """Random mode: label counts follow data/dataset.py:50-78, negatives are in [0, id_max) on the fast
    path and exact-complement on the tiny-graph path, frontier caps hold, Floyd subsets are uniform."""
    from oracle import sampler_ref as SR
    from laplace_amd.utils.constants import Constants
    hd, users, articles, ei = _random_graph(seed=33, U=80, A=60, E=1500)
    U, A = 80, 60
    ucsr, acsr = _csr_of(users, U), _csr_of(articles, A)
    id_max = int(ei[1].max())
    cfg = _cfg(n_hop_neighbors=3, num_neighbors=4, positive_edges_ratio=0.5, negative_edges_ratio=3.0)
    pairs = set(zip(ei[0].tolist(), ei[1].tolist()))
    for u in range(0, U, 5):
        # exact path: 1500 edges / n_neg <= 100 as soon as n_neg >= 15
        s = SR.sample_one(u, ucsr, acsr, ei.shape[1], id_max, cfg, seed=5, step=u)
        n_pos = int(s["edge_label"].sum())
        assert n_pos == max(1, len(users[u]) // 2)
        n_neg = len(s["edge_label"]) - n_pos
        want_neg = (cfg.k - 1) if n_pos <= 1 else int(3.0 * n_pos)
        lab_articles = s["article_ids"][s["edge_label_index"][1]]
        if ei.shape[1] / want_neg > 100:
            assert n_neg == want_neg and lab_articles[n_pos:].max() < id_max
        else:
            negs = lab_articles[n_pos:]
            assert len(set(negs.tolist())) == len(negs) == min(want_neg, id_max + 1 - len(set(lab_articles[:n_pos].tolist())))
            assert not set(negs.tolist()) & set(lab_articles[:n_pos].tolist())
        # users: seed + at most 4 per further hop; every message-passing edge is a real edge
        assert len(s["user_ids"]) <= 1 + 4 * 2
        eu, ea = s["user_ids"][s["edge_index"][0]], s["article_ids"][s["edge_index"][1]]
        assert all((a, b) in pairs for a, b in zip(eu.tolist(), ea.tolist()))
        # all edges of every included user are present (the walk never truncates a user's list)
        for x in s["user_ids"].tolist():
            assert int((eu == x).sum()) == len(users[x])
    counts = np.zeros(10)
    for trial in range(3000):
        for v in SR.floyd_subset(10, 3, SR.P_USER_CUT, 7, trial, seed=3, step=trial):
            counts[v] += 1
    assert np.abs(counts / counts.sum() - 0.1).max() < 0.01


def test_matchers_equal_reference_semantics():
    """UsersWithCommonItemsMatcher vs the reference's eager flatten/cat/[:k]; popular items; LightGCN rows."""
    from laplace_amd.data.matching import (LightGCNMatcher, PopularItemsMatcher, UsersWithCommonItemsMatcher, get_matchers)
    hd, users, articles, ei = _random_graph(seed=12, U=30, A=25, E=200)
    m = UsersWithCommonItemsMatcher(users, articles, k=20)
    for u in range(30):
        same = [v for a in users[u] for v in articles[a]]                       # users_with_common_purchases.py:15-19
        want = t.cat([t.as_tensor(users[v]) for v in same], dim=0)[:20]           # :20-26
        assert t.equal(m.get_matches(u), want)
    pop = PopularItemsMatcher.from_adjacency(articles, 5)
    deg = np.array([len(articles[a]) for a in range(25)])
    assert sorted(deg[pop.get_matches(0).numpy()].tolist(), reverse=True) == sorted(deg.tolist(), reverse=True)[:5]
    top = t.tensor([[4, 2, 9, -1], [1, 3, 5, 7]])
    assert LightGCNMatcher(top, 3).get_matches(0).tolist() == [4, 2, 9] and LightGCNMatcher(top, 4).get_matches(0).tolist() == [4, 2, 9]
    assert len(get_matchers("fashion", users, articles, 20)) == 2 and len(get_matchers("movielens", users, articles, 20)) == 1
    # eval-mode dataset with real matchers runs and labels positives 1 / candidates 0
    from laplace_amd.data.dataset import GraphDataset
    from laplace_amd.utils.constants import Constants
    ds = GraphDataset(_cfg(), hd, users, articles, train=False, matchers=get_matchers("fashion", users, articles, 10))
    item = ds[2]
    assert int(item[Constants.edge_key].edge_label.sum()) >= 1


def test_time_split_and_graph_files_match_reference(golden_dir, tmp_path):
    """Chronological leave-last-two-out split and the adjacency dicts equal the reference's pandas code
    (golden); files round-trip in the reference's layout and feed create_dataloaders."""
    from laplace_amd.data import graph_io
    from laplace_amd.data.data_loader import create_dataloaders
    from laplace_amd.data.matching import get_matchers
    from laplace_amd.utils.constants import Constants
    g = t.load(os.path.join(golden_dir, "time_split.pt"), weights_only=False)
    tr, va, te = graph_io.train_test_split_by_time(g["customer_id"])
    assert np.array_equal(tr, g["train_mask"]) and np.array_equal(va, g["val_mask"]) and np.array_equal(te, g["test_mask"])
    cx = t.randint(0, 9, (60, 3))
    ax = t.randint(0, 5, (45, 2))
    splits = graph_io.build_splits(cx, ax, g["customer_id"], g["article_id"])
    assert splits["train"][1] == g["edges_train"] and splits["train"][2] == g["rev_edges_train"]
    n_tr, n_va, n_te = (splits[k][0][Constants.edge_key].edge_index.shape[1] for k in ("train", "val", "test"))
    assert n_tr == int(tr.sum()) and n_va == n_tr + int(va.sum()) and n_te == n_va + int(te.sum())
    graph_io.write_splits(splits, str(tmp_path), {str(i): f"c{i}" for i in range(60)}, {str(i): f"a{i}" for i in range(45)})
    assert sorted(os.listdir(tmp_path)) == sorted(
        [f"{s}_graph.pt" for s in ("train", "val", "test")] + [f"edges_{s}.pt" for s in ("train", "val", "test")]
        + [f"rev_edges_{s}.pt" for s in ("train", "val", "test")] + ["customer_id_map_forward.json", "article_id_map_forward.json"])
    back, cmap, amap = graph_io.read_splits(str(tmp_path))
    assert back["val"][1] == splits["val"][1] and cmap["3"] == "c3" and amap["44"] == "a44"
    assert t.equal(back["test"][0][Constants.edge_key].edge_index, splits["test"][0][Constants.edge_key].edge_index)


def test_hetero_reduce_is_the_pairwise_queue_of_to_hetero():
    """temporary_hetero.py:203-228: [a, b, c] -> c (+) (a (+) b); [a, b, c, d] -> (a (+) b) (+) (c (+) d); mean
    divides once at the end.  Known answers where the association shows in the last bit, and exact ones for
    min / max / mul."""
    a, b, c, d = (t.tensor([x], dtype=t.float32) for x in (1.0, 2.0 ** -24, 2.0 ** -24, 3.0))
    # float32: (1 + 2^-24) rounds to 1 (ties to even), then + 2^-24 -> 1 again; 2^-24 + 2^-24 = 2^-23 first would survive
    assert RR.hetero_reduce([a, b, c], "sum").item() == (c + (a + b)).item() == 1.0
    assert RR.hetero_reduce([b, c, a], "sum").item() == (a + (b + c)).item() == 1.0 + 2.0 ** -23
    assert RR.hetero_reduce([a, b, c, d], "sum").item() == ((a + b) + (c + d)).item()
    assert RR.hetero_reduce([a, b, c, d], "mean").item() == (((a + b) + (c + d)) / 4).item()
    xs = [t.tensor([[1.0, -2.0], [3.0, 0.5]]), t.tensor([[0.0, 4.0], [-1.0, 0.25]]), t.tensor([[2.0, -3.0], [3.5, 8.0]])]
    assert t.equal(RR.hetero_reduce(xs, "max"), t.stack(xs).max(0)[0]) and t.equal(RR.hetero_reduce(xs, "min"), t.stack(xs).min(0)[0])
    assert t.equal(RR.hetero_reduce(xs, "mul"), xs[2] * (xs[0] * xs[1]))
    assert RR.hetero_reduce(xs[:1], "mul") is xs[0]
    # the product's reduction (torch ops, no kernel) follows the same queue
    from laplace_amd.model.encoder_decoder import _combine
    for aggr in ("sum", "mean", "min", "max", "mul"):
        for k in (1, 2, 3, 4, 5):
            g = t.Generator().manual_seed(k)
            outs = [t.randn(7, 5, generator=g) for _ in range(k)]
            assert t.equal(_combine(outs, aggr), RR.hetero_reduce(outs, aggr)), (aggr, k)

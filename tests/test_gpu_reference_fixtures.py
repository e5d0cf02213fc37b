"""GPU: the device sampler (csrc/sampler.hip) run directly on the reference's two hand-made test graphs
(/root/reference/tests/data_generator.py:129-157, "random" and "star") in randomization=False mode, asserted against
values derived BY HAND from the reference's sampler (data/dataset.py:39-182,258-286) for every seed user — the one pin
the reference itself holds for this path (tests/test_dataset.py:25-92 checks user 0 of "random").  Fixtures are data
(the two edge lists and the feature grids); nothing is imported from the reference."""
import os
import sys
from types import SimpleNamespace

import pytest
import torch as t

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu
DEV = "cuda"

from reference_fixtures import CFG, EXPECT, fixture as _fixture  # noqa: E402


@pytest.mark.parametrize("kind", ["random", "star"])
def test_device_sampler_on_reference_fixture_hand_derived(kind):
    from laplace_amd.data.dataset import AdjList, GraphDataset
    from laplace_amd.data.device_sampler import DeviceGraphSampler
    from laplace_amd.utils.constants import Constants
    g, users, articles, ux, ax, ei = _fixture(kind)
    n_u, n_a = ux.shape[0], ax.shape[0]
    smp = DeviceGraphSampler(CFG, g, AdjList(users, n_u), AdjList(articles, n_a), batch_size=1, randomization=False,
                             device=DEV, seed=0)
    host = GraphDataset(CFG, g, users, articles, train=True, randomization=False)
    for seed_user, (ub, ab, edges, label_edges, labels) in EXPECT[kind].items():
        raw = smp.sample(t.tensor([seed_user]), step=0, raw=True)
        uid, aid = raw["user_ids"].cpu().tolist(), raw["article_ids"].cpu().tolist()
        assert uid == ub and aid == ab, (kind, seed_user)
        upos, apos = {u: i for i, u in enumerate(ub)}, {a: i for i, a in enumerate(ab)}
        got_edges = sorted(zip(*raw["edge_index"].cpu().tolist()))
        assert got_edges == sorted((upos[u], apos[a]) for u, a in edges), (kind, seed_user)
        assert raw["edge_label_index"].cpu().tolist() == [[upos[u] for u, _ in label_edges], [apos[a] for _, a in label_edges]]
        assert raw["edge_label"].cpu().tolist() == labels
        assert raw["user_ptr"].cpu().tolist() == [0, len(ub)] and raw["article_ptr"].cpu().tolist() == [0, len(ab)]
        # the collated batch: features gathered by bucket, reverse relation mirrored (data/dataset.py:158-181)
        b = smp.sample(t.tensor([seed_user]), step=0)
        assert t.equal(b[Constants.node_user].x.cpu(), ux[ub]) and t.equal(b[Constants.node_item].x.cpu(), ax[ab])
        fwd, rev = b[Constants.edge_key], b[Constants.rev_edge_key]
        assert t.equal(rev.edge_index, fwd.edge_index.flip(0)) and t.equal(rev.edge_label_index, fwd.edge_label_index.flip(0))
        assert fwd.edge_label.dtype == t.long and t.equal(rev.edge_label, fwd.edge_label)
        # and the host-side GraphDataset (the other implementation of the same sampler) says the same
        h = host[seed_user]
        assert t.equal(h[Constants.node_user].x, ux[ub]) and t.equal(h[Constants.node_item].x, ax[ab])
        assert sorted(zip(*h[Constants.edge_key].edge_index.tolist())) == got_edges
        assert t.equal(h[Constants.edge_key].edge_label_index, raw["edge_label_index"].cpu())
        assert h[Constants.edge_key].edge_label.tolist() == labels


@pytest.mark.parametrize("kind", ["random", "star"])
def test_device_sampler_whole_fixture_as_one_batch(kind):
    """All users of the fixture as ONE batch: samples are concatenated with node offsets (PyG collate), each equal to
    its single-sample form."""
    from laplace_amd.data.dataset import AdjList
    from laplace_amd.data.device_sampler import DeviceGraphSampler
    g, users, articles, ux, ax, ei = _fixture(kind)
    n_u, n_a = ux.shape[0], ax.shape[0]
    cfg = SimpleNamespace(**{**vars(CFG), "batch_size": n_u})
    smp = DeviceGraphSampler(cfg, g, AdjList(users, n_u), AdjList(articles, n_a), batch_size=n_u, randomization=False,
                             device=DEV, seed=0)
    raw = smp.sample(t.arange(n_u), step=0, raw=True)
    up, ap = raw["user_ptr"].cpu().tolist(), raw["article_ptr"].cpu().tolist()
    e_at, l_at = 0, 0
    for s in range(n_u):
        ub, ab, edges, label_edges, labels = EXPECT[kind][s]
        assert raw["user_ids"].cpu().tolist()[up[s]:up[s + 1]] == ub
        assert raw["article_ids"].cpu().tolist()[ap[s]:ap[s + 1]] == ab
        upos, apos = {u: i + up[s] for i, u in enumerate(ub)}, {a: i + ap[s] for i, a in enumerate(ab)}
        got = sorted(zip(*raw["edge_index"][:, e_at:e_at + len(edges)].cpu().tolist()))
        assert got == sorted((upos[u], apos[a]) for u, a in edges)
        li = raw["edge_label_index"][:, l_at:l_at + len(labels)].cpu().tolist()
        assert li == [[upos[u] for u, _ in label_edges], [apos[a] for _, a in label_edges]]
        assert raw["edge_label"][l_at:l_at + len(labels)].cpu().tolist() == labels
        e_at += len(edges)
        l_at += len(labels)
    assert e_at == raw["edge_index"].shape[1] and l_at == raw["edge_label"].numel()

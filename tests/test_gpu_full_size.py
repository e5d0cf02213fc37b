"""GPU: the metric's own configurations at FULL size (BASELINE.json configs[3] and configs[2]) on one MI355X.

configs[3] / SURVEY C4 at N = 1: ONE block-wise generated 8 M x 100 K graph with 100 M edges (symmetric adjacency
nnz = 200 M, D = 128, K = 3, global batch 131 072) — the oracle cannot run this in seconds, so the checks are the
size-independent properties of tests/test_gpu_lightgcn.py::test_full_size_propagate_properties: linearity, the D^1/2
eigenvector of the normalised adjacency, bitwise run-to-run stability, ~2 000 sampled rows (top hubs included) against
float64, sampler validity, and one fused train step with |delta| <= lr.

configs[2] / SURVEY C3 at the full H&M shape (1 371 980 customers x 105 542 articles x 31.8 M transactions, 24 users per
batch, 2 hops, fan-out 64): the on-device sampler against its numpy mirror on a batch of 24 seed users (bit-exact), and
the encoder-decoder's logits on that batch against the torch-only twin (<= 1e-4, north_star)."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch as t

pytestmark = pytest.mark.gpu
DEV = "cuda"


# ------------------------------------------------------------------------------------------ configs[3], N = 1
@pytest.fixture(scope="module")
def c4_graph():
    from laplace_amd import synthetic as S
    from laplace_amd.interactions import Interactions
    ei = S.generate_blocks(S.C4, S.C4_BLOCKS, 0, S.C4_BLOCKS)
    inter = Interactions(ei.to(DEV), S.C4.num_users, S.C4.num_items)
    adj, _ = inter.adjacency("bipartite").gcn_normalized(False)
    yield ei, inter, adj
    del inter, adj
    t.cuda.empty_cache()


def _row_f64(adj, X, r, piece=1 << 20):
    """sum_p val[p] * X[col[p]] of row r in float64 on the device, the entries taken a million at a time."""
    b, e = int(adj.rowptr[r]), int(adj.rowptr[r + 1])
    acc = t.zeros(X.shape[1], dtype=t.float64, device=X.device)
    for lo in range(b, e, piece):
        hi = min(e, lo + piece)
        acc += (adj.val[lo:hi].double()[:, None] * X[adj.col[lo:hi].long()].double()).sum(0)
    return acc, e - b


def test_c4_full_size_propagate_properties(c4_graph):
    from laplace_amd import ops, synthetic as S
    ei, inter, adj = c4_graph
    U, I = S.C4.num_users, S.C4.num_items
    n, d = adj.n_rows, 128
    assert ei.shape == (2, 100_000_000) and adj.nnz == 200_000_000 and n == U + I == 8_100_000
    assert adj.plan is not None and adj.plan.n_long_rows > 10_000 and adj.plan.n_items > 100_000   # the typical item row is a split row
    g = t.Generator(device=DEV).manual_seed(0)
    X = t.randn(n, d, device=DEV, generator=g) * 0.1
    Z = t.randn(n, d, device=DEV, generator=g) * 0.1
    yx, yz, yl = (t.empty(n, d, device=DEV) for _ in range(3))
    ops.spmm(adj, X, Y=yx)
    ops.spmm(adj, Z, Y=yz)
    ops.spmm(adj, 2.0 * X - 0.5 * Z, Y=yl)
    assert (yl - (2.0 * yx - 0.5 * yz)).abs().max() <= 5e-5          # hub rows sum ~10^6 terms
    del yz, yl, Z
    y2 = t.full((n, d), float("nan"), device=DEV)
    ops.spmm(adj, X, Y=y2)
    assert t.equal(y2, yx)                                             # bitwise reproducible: no float atomics
    del y2
    # A~ (D^1/2 1) = D^1/2 1 on rows with deg > 0
    deg = (adj.rowptr[1:] - adj.rowptr[:-1]).float()
    v = deg.sqrt()[:, None].expand(n, 4).contiguous()
    out = t.empty(n, 4, device=DEV)
    ops.spmm(adj, v, Y=out)
    assert ((out - v).abs() / v.clamp(min=1.0)).max() <= 2e-4
    # ~2 000 sampled rows, the 20 largest hubs among them, against float64
    rows = t.cat([t.randint(0, n, (1980,)), deg.cpu().topk(20).indices])
    worst = 0.0
    for r in rows.tolist():
        want, cnt = _row_f64(adj, X, r)
        err = float((yx[r].double() - want).abs().max())
        assert err <= 1e-5 + 1e-6 * cnt ** 0.5, (r, cnt, err)
        worst = max(worst, err)
    assert int(deg.max()) > 1_000_000 and worst > 0.0                  # the hubs of this graph really are 10^6-entry rows


def test_c4_full_size_mapped_product_with_rare_live_columns(c4_graph):
    """The first backward product of the fused step at configs[3]'s size: the adjacency times a compact batch gradient through
    x_map (131 072 sampled edges' users, their items, uniform negatives).  The rare-live-columns form (mi_spmm_ex.x_bits:
    spmm_items_xscan_kernel + live flags) must be bitwise the plain mapped product and bitwise reproducible — at THIS size the
    outputs are streamed (sc1 stores through inline asm), a path no small test reaches: the form's first version passed every
    small test and differed run to run here (a missing wait state behind the store)."""
    from laplace_amd import ops, synthetic as S
    ei, inter, adj = c4_graph
    U, I, B = S.C4.num_users, S.C4.num_items, 131_072
    n, d = adj.n_rows, 128
    us, ps, ns = ops.sample_bpr_batch(inter.csr(), inter.row_of_edge(), B, I, seed=5, step=3)
    gmap, nodes, cnt = ops.batch_nodes(us, ps, ns, U, n)
    g = t.Generator(device=DEV).manual_seed(1)
    Xc = t.randn(3 * B, d, device=DEV, generator=g) * 0.1
    outs = []
    for rare in (False, True, True):
        o = t.full((n, d), float("nan"), device=DEV)
        ops.spmm(adj, Xc, addend=Xc, S=o, x_map=gmap, addend_map=gmap, x_rare=rare)
        outs.append(o)
    t.cuda.synchronize()
    assert t.equal(outs[1], outs[2])          # reproducible
    assert t.equal(outs[1], outs[0])          # the hint changes no bit
    # a handful of hub rows against float64 of the expanded operand
    deg = (adj.rowptr[1:] - adj.rowptr[:-1])
    for r in deg.cpu().topk(6).indices.tolist() + [U + 5000, U + 50_000]:
        b, e = int(adj.rowptr[r]), int(adj.rowptr[r + 1])
        cols = adj.col[b:e].long()
        m = gmap[cols].long()
        live = m >= 0
        want = (adj.val[b:e][live].double()[:, None] * Xc[m[live]].double()).sum(0)
        ar = int(gmap[r])
        if ar >= 0:
            want = want + Xc[ar].double()
        assert (outs[1][r].double() - want).abs().max() <= 1e-5 + 1e-6 * float(live.sum()) ** 0.5, r
    del outs


def test_c4_full_size_sampler_and_one_fused_step(c4_graph):
    """bench.py --config c4 at N = 1: on-device sampling of the global batch (131 072), one fused train step under the
    locality order; rows without gradient stay put, no parameter moves by more than lr (Adam's first step)."""
    from laplace_amd import ops, synthetic as S
    from laplace_amd.model.lightgcn import LightGCN
    from laplace_amd.trainer import LightGCNTrainer
    ei, inter, adj = c4_graph
    U, I, B = S.C4.num_users, S.C4.num_items, 131_072
    us, ps, ns = ops.sample_bpr_batch(inter.csr(), inter.row_of_edge(), B, I, seed=3, step=11)
    keys = t.sort(ei[0].to(DEV) * I + ei[1].to(DEV))[0]

    def member(k):
        pos = t.searchsorted(keys, k).clamp(max=keys.numel() - 1)
        return keys[pos] == k
    assert bool(member(us * I + ps).all()) and not bool(member(us * I + ns).any())
    del keys
    t.manual_seed(0)
    model = LightGCN(U, I, 128, 3).to(DEV)
    before = model.table().clone()                                     # rows under their original ids
    tr = LightGCNTrainer(model, inter.adjacency("bipartite"), inter, lr=1e-3, Lambda=1e-6, batch_size=B, seed=7)
    assert tr.order is not None                                        # 200 M entries: trained under the locality order
    batch = tr.sample()
    loss = tr.step(batch)
    tr.finish()
    after = model.table()
    delta = (after - before).abs()
    assert t.isfinite(loss).all() and float(delta.max()) <= 1e-3 * 1.001
    # every batch node moved (its BPR / L2 gradient is non-zero); an isolated user outside the batch did not
    bu = t.unique(batch[0])
    assert bool((delta[bu].amax(dim=1) > 0).all())
    deg_u = t.bincount(ei[0], minlength=U)
    lonely = t.nonzero(deg_u == 0).view(-1)
    if lonely.numel():
        assert float(delta[lonely.to(DEV)].max()) == 0.0
    # second trainer, same seed, same batch: the step is bitwise reproducible at this size too
    with t.no_grad():
        model.table().copy_(before)
    tr2 = LightGCNTrainer(model, inter.adjacency("bipartite"), inter, lr=1e-3, Lambda=1e-6, batch_size=B, seed=7)
    tr2.step(batch)
    tr2.finish()
    assert t.equal(model.table(), after)


# ------------------------------------------------------------------------------------------ configs[2], full H&M shape
@pytest.fixture(scope="module")
def hm_graph():
    from laplace_amd import synthetic as S
    spec = S.SyntheticSpec(1_371_980, 105_542, 31_800_000, seed=2, zipf_s=1.0)
    return S.generate_hetero(spec)


def test_c3_full_size_device_sampler_bit_exact_and_ranker_logits(hm_graph):
    from oracle import ranker_ref as RR
    from oracle import sampler_ref as SR
    from laplace_amd.data.device_sampler import DeviceGraphSampler
    from laplace_amd.model.encoder_decoder import Encoder_Decoder_Model
    from laplace_amd.model.layers import get_SAGEConv_layers, get_linear_layers
    from laplace_amd.utils.constants import Constants
    from laplace_amd.utils.get_info import get_feature_info, select_properties
    graph, users, articles = hm_graph
    ei = graph[Constants.edge_key].edge_index
    assert ei.shape == (2, 31_800_000)
    assert tuple(graph[Constants.node_user].x.shape) == (1_371_980, 6) and tuple(graph[Constants.node_item].x.shape) == (105_542, 4)
    cfg = SimpleNamespace(k=12, num_neighbors=64, n_hop_neighbors=2, positive_edges_ratio=0.5, negative_edges_ratio=3.0,
                          batch_size=24)                                # config.py:99-135 defaults
    smp = DeviceGraphSampler(cfg, graph, users, articles, batch_size=24, randomization=True, device=DEV, seed=11)
    ucsr, acsr = SR.CsrAdj(users.ptr, users.idx), SR.CsrAdj(articles.ptr, articles.idx)
    g = t.Generator().manual_seed(0)
    seeds = t.randperm(1_371_980, generator=g)[:24]
    hub_buyers = t.from_numpy(articles[int(np.argmax(np.diff(articles.ptr)))][:4].copy())   # users of the most popular article
    seeds[:4] = hub_buyers
    for step in (0, 5):
        got = smp.sample(seeds, step=step, raw=True)
        want = SR.sample_batch(seeds.tolist(), ucsr, acsr, int(ei.shape[1]), int(ei[1].max()), cfg, 11, step, True)
        for key in ("user_ids", "article_ids", "edge_index", "edge_label_index", "edge_label"):
            assert np.array_equal(got[key].cpu().numpy(), want[key]), (key, step)
    # structure at this scale: ~10^4 nodes, every message-passing edge is a transaction, labels are 0/1, the fan-out caps hold
    uid, aid = want["user_ids"], want["article_ids"]
    assert 1_000 < len(uid) + len(aid) < 200_000
    src, dst = uid[want["edge_index"][0]], aid[want["edge_index"][1]]
    key_all = np.sort(ei[0].numpy() * 105_542 + ei[1].numpy())
    pos = np.searchsorted(key_all, src * 105_542 + dst)
    assert np.array_equal(key_all[np.minimum(pos, key_all.size - 1)], src * 105_542 + dst)
    assert set(np.unique(want["edge_label"]).tolist()) <= {0, 1} and int(want["edge_label"].sum()) > 0
    # the model on that batch: logits against the torch-only twin
    batch = smp.sample(seeds, step=5)
    t.manual_seed(0)
    model = Encoder_Decoder_Model(get_SAGEConv_layers(2, 128, 64, "add"), get_linear_layers(2, 128, 128, 1),
                                  get_feature_info(graph), batch.metadata(), True, "sum", True, 0.0, 0.0).to(DEV)
    model.initialize_encoder_input_size(batch)
    ref = RR.ref_from_product(model, batch.x_dict)
    x, edges, eli, y = select_properties(batch)
    model.train(); ref.train()                                          # batch statistics in the BatchNorm layers, as in training
    with t.no_grad():
        got_logits = model(x, edges, eli).view(-1).cpu()
        cpu = batch.to("cpu")
        xc, ec, elic, _ = select_properties(cpu)
        want_logits = ref({k: v.clone() for k, v in xc.items()}, ec, elic).view(-1)
    assert got_logits.shape == want_logits.shape == (int(y.numel()),)
    assert float((got_logits - want_logits).abs().max()) <= 1e-4


# ------------------------------------------------------------------------------------------ configs[4], full H&M shape
@pytest.mark.parametrize("walk_length", [2, 3])
def test_c5_full_size_pinsage_batches_bit_exact_and_native_step_vs_twin(hm_graph, walk_length):
    """BASELINE configs[4] (PinSage item-item link prediction on the H&M graph) at its workload: the 1 371 980 x 105 542 x
    31.8 M graph of bench.py's pinsage_c5 block, the reference's batch of 32 pairs, 10 walks, restart 0.5, T = 3, 2 layers,
    hidden 16 (pinsage/model.py:143-148), walk length 2 (the reference's default) and 3 (BASELINE's "3-hop").
    (1) the batch built on the device by ONE C call (mi_pinsage_sample_batch: pairs, seeds, both blocks) against the numpy
    mirror of pinsage/sampler.py:16-106, bit for bit, on three steps; (2) ONE mi_pinsage_step_f32 iteration on such a
    batch — loss and every parameter gradient, the dense 105 543 x 16 projector table's included — directly against the
    torch-only twin (oracle/pinsage_ref.py) evaluated on the MIRROR's batch, dropout off."""
    from oracle import pinsage_ref as PR
    from laplace_amd.pinsage.model import PinSAGEModel
    from laplace_amd.pinsage.native import NativePinSAGEStep
    from laplace_amd.pinsage.sampler import PinSAGESampler
    graph, users, articles = hm_graph
    U, I, H, LAYERS, SEED = 1_371_980, 105_542, 16, 2, 13
    ucsr, icsr = PR.Csr(users.ptr, users.idx), PR.Csr(articles.ptr, articles.idx)
    smp = PinSAGESampler(users, articles, U, I, batch_size=32, random_walk_length=walk_length, random_walk_restart_prob=0.5,
                         num_random_walks=10, num_neighbors=3, num_layers=LAYERS, seed=SEED)
    batches = {}
    for step in (0, 7, 12345):
        got = smp._sample_batch_device(step)
        assert got is not None                                          # the reference's sizes are inside the device builder's
        wh, wt, wn = PR.item_pairs(32, I, icsr, ucsr, SEED, step)
        want = PR.sample_from_item_pairs(wh, wt, wn, icsr, ucsr, LAYERS, walk_length, 0.5, 10, 3, SEED, step)
        assert 20 <= len(wh) <= 32 and len(want["seeds"]) > 32
        assert np.array_equal(got["seeds"].cpu().numpy(), want["seeds"]), step
        for a, b in zip(got["pos"] + got["neg"], want["pos"] + want["neg"]):
            assert np.array_equal(a.cpu().numpy(), b), step
        assert len(got["blocks"]) == LAYERS
        for gb, wb in zip(got["blocks"], want["blocks"]):
            assert gb["n_dst"] == wb["n_dst"]
            for key in ("src_ids", "edge_src", "edge_dst", "weights"):
                assert np.array_equal(gb[key].cpu().numpy(), wb[key]), (step, key)
        batches[step] = (got, want)
    # walks of 3 traversals visit more distinct items than walks of 2: the two settings really are different samplers
    assert bool((smp._pos32 == -1).all())
    t.manual_seed(3)
    model = PinSAGEModel(I, H, LAYERS).to(DEV)
    with t.no_grad():
        model.bias.normal_(0, 0.1)
    for cv in model.convs:
        cv.dropout.p = 0.0
    ref = PR.PinSAGERef(I, H, LAYERS)
    ref.load_state_dict({k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
    for cv in ref.convs:
        cv.dropout.p = 0.0
    opt = t.optim.Adam(model.parameters(), lr=3e-5)                     # pinsage/model.py:153
    probe = NativePinSAGEStep(model, opt, keep_grads=True)
    model.train(); ref.train()
    for step in (7, 12345):
        got, want = batches[step]
        loss = probe.step(got)
        assert loss is not None, probe.declined
        tb = PR.to_torch_blocks(want["blocks"])
        ref.zero_grad()
        out = ref(t.from_numpy(want["seeds"]), tuple(t.from_numpy(x) for x in want["pos"]),
                  tuple(t.from_numpy(x) for x in want["neg"]), tb)
        loss_ref = out.mean()
        loss_ref.backward()
        assert abs(float(loss) - float(loss_ref)) <= 1e-5 * max(1.0, abs(float(loss_ref))), step
        for (n, p), (_, pr) in zip(model.named_parameters(), ref.named_parameters()):
            scale = float(pr.grad.abs().max()) + 1e-12
            assert float((p.grad.cpu() - pr.grad).abs().max()) <= 2e-4 * scale + 1e-8, (step, n)
        # the dense table gradients are non-zero exactly on the batch's rows
        rows = t.nonzero(model.proj.weight.grad.abs().sum(1) > 0).view(-1).cpu()
        rows_ref = t.nonzero(ref.proj.weight.grad.abs().sum(1) > 0).view(-1)
        assert t.equal(rows, rows_ref) and 0 < rows.numel() <= len(want["blocks"][0]["src_ids"])
        model.proj.weight.grad.zero_(); model.bias.grad.zero_()         # what the probe left behind
    del model, probe
    t.cuda.empty_cache()

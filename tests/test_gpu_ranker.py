"""GPU parity of the ranker (SAGEConv encoder / MLP decoder) against the torch-only oracle twin."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch as t

from oracle import ranker_ref as RR

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_embed_concat_matches_torch_embedding_max_norm():
    from laplace_amd import ops
    g = t.Generator().manual_seed(0)
    dims, rows = [2, 4, 12, 20, 60], [3, 11, 85, 900, 5000]
    tables = [t.randn(r, d, generator=g) * (0.2 if i == 1 else 1.0) for i, (r, d) in enumerate(zip(rows, dims))]
    x = t.stack([t.randint(0, r, (333,), generator=g) for r in rows], dim=1)
    got = ops.embed_concat(x.to(DEV), [tb.to(DEV) for tb in tables], max_norm=1.0)
    parts = []
    for i, tb in enumerate(tables):
        e = t.nn.Embedding(tb.shape[0], tb.shape[1], max_norm=1)
        with t.no_grad():
            e.weight.copy_(tb)
        parts.append(e(x[:, i]).detach())
    want = t.cat(parts, dim=1)
    assert got.shape == want.shape == (333, sum(dims))
    assert t.allclose(got.cpu(), want, atol=1e-6, rtol=1e-6)
    assert float(got.cpu()[:, 2:6].norm(dim=1).max()) <= 1.0 + 1e-6  # small-norm rows of table 1 left alone
    for tb, ref in zip(tables, [tb.clone() for tb in tables]):
        assert t.equal(tb, ref)  # tables never written


@pytest.mark.parametrize("aggr", ["add", "mean", "max"])
@pytest.mark.parametrize("d_src,d_dst,out", [(84, 76, 128), (6, 10, 8), (128, 128, 64)])
def test_sageconv_forward_backward_parity(aggr, d_src, d_dst, out):
    from laplace_amd.model.layers import SAGEConv
    g = t.Generator().manual_seed(d_src + out)
    n_src, n_dst, E = 150, 90, 1200
    ei = t.stack([t.randint(0, n_src, (E,), generator=g), t.randint(0, n_dst - 5, (E,), generator=g)])  # 5 empty dst
    xs = t.randn(n_src, d_src, generator=g).requires_grad_(True)
    xd = t.randn(n_dst, d_dst, generator=g).requires_grad_(True)
    conv = SAGEConv((-1, -1, -1), out, aggr=aggr)
    xs_g = xs.detach().clone().to(DEV).requires_grad_(True)
    xd_g = xd.detach().clone().to(DEV).requires_grad_(True)
    y = conv((xs_g, xd_g), ei.to(DEV))
    ref = RR.SAGEConvRef(d_src, d_dst, out, aggr)
    ref.load_state_dict({k: v.detach().cpu() for k, v in conv.state_dict().items()})
    y_ref = ref((xs, xd), ei)
    assert t.allclose(y.detach().cpu(), y_ref.detach(), atol=2e-5, rtol=1e-5)
    w = t.randn(n_dst, out, generator=g)
    (y * w.to(DEV)).sum().backward()
    (y_ref * w).sum().backward()
    assert t.allclose(xs_g.grad.cpu(), xs.grad, atol=2e-5, rtol=1e-4)
    assert t.allclose(xd_g.grad.cpu(), xd.grad, atol=2e-5, rtol=1e-4)
    for name, p in conv.named_parameters():
        pr = dict(ref.named_parameters())[name]
        assert t.allclose(p.grad.cpu(), pr.grad, atol=5e-4, rtol=1e-4), name


def _hetero_setup(seed=0, aggr="add", embedding=True, p_drop=0.0):
    from laplace_amd import synthetic as S
    from laplace_amd.config import Config
    from laplace_amd.data.dataset import GraphDataset
    from laplace_amd.hetero import DataLoader
    from laplace_amd.model.encoder_decoder import Encoder_Decoder_Model
    from laplace_amd.model.layers import get_SAGEConv_layers, get_linear_layers
    from laplace_amd.utils.constants import Constants
    from laplace_amd.utils.get_info import get_feature_info
    spec = S.SyntheticSpec(400, 120, 3000, seed=seed + 5, deg_min=2, deg_max=60)
    graph, users, articles = S.generate_hetero(spec, customer_cards=(300, 2, 84, 4, 5, 2), article_cards=(100, 132, 30, 50))
    if not embedding:
        graph[Constants.node_user].x = graph[Constants.node_user].x.float() / 100.0
        graph[Constants.node_item].x = graph[Constants.node_item].x.float() / 100.0
    cfg = SimpleNamespace(k=12, num_neighbors=8, n_hop_neighbors=2, positive_edges_ratio=0.5, negative_edges_ratio=3.0,
                          batch_size=6)
    ds = GraphDataset(cfg, graph, users, articles, train=True, randomization=True, seed=seed)
    loader = DataLoader(ds, batch_size=cfg.batch_size, shuffle=True, generator=t.Generator().manual_seed(seed))
    t.manual_seed(seed)
    first = next(iter(loader))
    info = get_feature_info(graph) if embedding else {}
    model = Encoder_Decoder_Model(
        encoder_layers=get_SAGEConv_layers(2, 128, 64, aggr), decoder_layers=get_linear_layers(2, 128, 128, 1),
        feature_info=info, metadata=first.metadata(), embedding=embedding, heterogeneous_prop_agg_type="sum",
        batch_normalize=True, p_dropout_edges=0.0, p_dropout_features=p_drop).to(DEV)
    model.initialize_encoder_input_size(first.to(DEV))
    return model, loader, first


@pytest.mark.parametrize("aggr,embedding", [("add", True), ("mean", True), ("max", False)])
def test_encoder_decoder_logits_gradients_and_training_parity(aggr, embedding):
    """Five training iterations of training.py:19-34.  At every iteration both models hold identical
    weights and see the same batch: logits within 1e-4 (north_star), loss and every parameter gradient
    close, BatchNorm running statistics tracked identically.  The oracle's Adam then advances the
    weights and the product is re-synchronised: comparing free-running parameters after several Adam
    steps is ill-conditioned in the reference itself (a gradient entry at rounding-noise level moves
    its weight by +-lr whichever sign the noise takes), so the chain is cut at each step."""
    from laplace_amd.utils.get_info import select_properties
    model, loader, first = _hetero_setup(seed=1, aggr=aggr, embedding=embedding)
    ref = RR.ref_from_product(model, first.x_dict)
    names = [n for n, _ in model.named_parameters()]
    assert names == [n for n, _ in ref.named_parameters()]
    # categorical tables are NOT parameters
    assert not any("embedding" in n for n in names) and len(model.state_dict()) == len(ref.state_dict())
    opt_ref = t.optim.Adam(ref.parameters(), lr=0.01)
    crit = t.nn.BCEWithLogitsLoss()
    model.train(); ref.train()
    for step, batch in enumerate(loader):
        if step == 5:
            break
        x, ei, eli, y = select_properties(batch)
        out_ref = ref({k: v.clone() for k, v in x.items()}, ei, eli)
        xg, eig, elig, yg = select_properties(batch.to(DEV))
        out = model(xg, eig, elig)
        assert out.shape == out_ref.shape
        assert (out.detach().cpu() - out_ref.detach()).abs().max() <= 1e-4, step
        loss, loss_ref = crit(out, yg), crit(out_ref, y)
        assert abs(float(loss) - float(loss_ref)) <= 1e-5
        model.zero_grad(); opt_ref.zero_grad()
        loss.backward(); loss_ref.backward()
        for (n, p), (_, pr) in zip(model.named_parameters(), ref.named_parameters()):
            scale = float(pr.grad.abs().max()) + 1e-8
            assert float((p.grad.cpu() - pr.grad).abs().max()) <= 2e-4 * scale + 1e-6, (step, n)  # biases before BatchNorm have an exactly-zero true gradient: both sides hold rounding noise ~1e-7
        for bn in ("encoder_layer_norm_customer", "encoder_layer_norm_article"):
            assert t.allclose(getattr(model, bn).running_mean.cpu(), getattr(ref, bn).running_mean, atol=1e-5)
            assert t.allclose(getattr(model, bn).running_var.cpu(), getattr(ref, bn).running_var, atol=1e-5, rtol=1e-5)
        opt_ref.step()
        model.load_state_dict({k: v.to(DEV) for k, v in ref.state_dict().items()})
    # infer: eval-mode scores regrouped per user, padded with -(1<<50)
    batch = next(iter(loader))
    x, ei, eli, _ = select_properties(batch)
    want = ref.infer({k: v.clone() for k, v in x.items()}, ei, eli)
    xg, eig, elig, _ = select_properties(batch.to(DEV))
    got = model.infer(xg, eig, elig).cpu()
    assert got.shape == want.shape
    pad = want == float(-(1 << 50))
    assert t.equal(got == float(-(1 << 50)), pad)
    assert (got[~pad] - want[~pad]).abs().max() <= 1e-4


def test_train_and_test_loops_run_and_learn():
    from laplace_amd.training import test_with_dataloader, train_with_dataloader
    model, loader, first = _hetero_setup(seed=3, aggr="add", embedding=True, p_drop=0.3)
    opt = t.optim.Adam(model.parameters(), lr=0.01)
    first_epoch = train_with_dataloader(model, opt, loader, 0, DEV)
    for ep in range(1, 4):
        losses = train_with_dataloader(model, opt, loader, ep, DEV)
    assert np.isfinite(losses).all() and np.mean(losses) < np.mean(first_epoch)
    recall, precision = test_with_dataloader("VAL", model, loader, DEV, k=3, break_at=5)
    assert 0.0 <= recall <= 1.0 and 0.0 <= precision <= 1.0


@pytest.mark.parametrize("hetero_aggr", ["sum", "mean", "min", "max", "mul"])
@pytest.mark.parametrize("conv_aggr", ["add", "max"])
def test_to_hetero_three_relations_per_destination(hetero_aggr, conv_aggr):
    """config.py:128-129 `other_edge_types`: several relations arrive at one destination type and
    `heterogeneous_prop_agg_type` stops being a no-op.  Three relations into `article`, two into `customer`, two
    layers, forward and every parameter gradient against the oracle's restatement of to_hetero's pairwise
    reduction (temporary_hetero.py:203-228)."""
    from laplace_amd.model.encoder_decoder import HeteroGNNEncoder
    from laplace_amd.model.layers import get_SAGEConv_layers
    g = t.Generator().manual_seed(11)
    n_c, n_a, C = 70, 50, 24
    node_types = ["customer", "article"]
    edge_types = [("customer", "buys", "article"), ("article", "rev_buys", "customer"), ("customer", "views", "article"),
                  ("customer", "wishes", "article"), ("article", "rev_views", "customer")]
    x = {"customer": t.randn(n_c, C, generator=g), "article": t.randn(n_a, C, generator=g)}
    ei = {}
    for et in edge_types:
        n_s, n_d = (n_c, n_a) if et[0] == "customer" else (n_a, n_c)
        e = int(t.randint(150, 400, (1,), generator=g))
        ei[et] = t.stack([t.randint(0, n_s, (e,), generator=g), t.randint(0, n_d, (e,), generator=g)])
    t.manual_seed(3)
    enc = HeteroGNNEncoder(get_SAGEConv_layers(2, 32, 16, conv_aggr), (node_types, edge_types), hetero_aggr, 0.0, None).to(DEV)
    out = enc({k: v.to(DEV) for k, v in x.items()}, {k: v.to(DEV) for k, v in ei.items()})  # sizes the lazy layers
    dims = [{k: (c.lin_l.in_features, c.lin_r.in_features, c.out_channels) for k, c in convs.items()} for convs in enc.layers]
    ref = RR.HeteroEncoderRef(dims, conv_aggr, hetero_aggr, None)
    ref.load_state_dict({k: v.detach().cpu().clone() for k, v in enc.state_dict().items()})
    want = ref({k: v.clone() for k, v in x.items()}, ei)
    assert set(out) == set(want) == {"customer", "article"}
    for k in want:
        scale = float(want[k].abs().max()) + 1e-6
        assert (out[k].detach().cpu() - want[k].detach()).abs().max() <= 1e-5 * max(1.0, scale), (k, hetero_aggr)
    w = {k: t.randn(v.shape, generator=g) for k, v in want.items()}
    sum((out[k] * w[k].to(DEV)).sum() for k in out).backward()
    sum((want[k] * w[k]).sum() for k in want).backward()
    for (n, p), (_, pr) in zip(enc.named_parameters(), ref.named_parameters()):
        scale = float(pr.grad.abs().max()) + 1e-6
        assert float((p.grad.cpu() - pr.grad).abs().max()) <= 2e-4 * scale + 1e-6, (n, hetero_aggr)


def test_embed_concat_more_than_sixteen_columns_and_id_validation():
    """The reference allows any number of categorical columns; ids outside a table raise (as nn.Embedding does) in
    the model's validate_features instead of being clamped in silence."""
    from laplace_amd import ops
    from laplace_amd.model.encoder_decoder import Encoder_Decoder_Model
    from laplace_amd.model.layers import get_SAGEConv_layers, get_linear_layers
    g = t.Generator().manual_seed(2)
    nc = 37
    dims = [int(t.randint(1, 9, (1,), generator=g)) for _ in range(nc)]
    rows = [int(t.randint(2, 50, (1,), generator=g)) for _ in range(nc)]
    tables = [t.randn(r, d, generator=g) for r, d in zip(rows, dims)]
    x = t.stack([t.randint(0, r, (101,), generator=g) for r in rows], dim=1)
    got = ops.embed_concat(x.to(DEV), [tb.to(DEV) for tb in tables], max_norm=1.0).cpu()
    want = t.cat([t.nn.functional.embedding(x[:, i], tb.clone(), max_norm=1.0) for i, tb in enumerate(tables)], dim=1)
    assert t.allclose(got, want, atol=1e-6, rtol=1e-6)
    info = {"customer": SimpleNamespace(num_feat=2, num_cat=[4, 9], embedding_size=[2, 4]),
            "article": SimpleNamespace(num_feat=1, num_cat=[6], embedding_size=[4])}
    meta = (["customer", "article"], [("customer", "buys", "article"), ("article", "rev_buys", "customer")])
    model = Encoder_Decoder_Model(get_SAGEConv_layers(2, 16, 8, "add"), get_linear_layers(2, 16, 16, 1), info, meta, True,
                                  "sum", True, 0.0, 0.0).to(DEV)
    ok = {"customer": t.tensor([[0, 9], [4, 0]], device=DEV), "article": t.tensor([[6], [0]], device=DEV)}
    model.validate_features(ok)
    for bad in ({"customer": t.tensor([[5, 0]], device=DEV), "article": ok["article"]},
                {"customer": ok["customer"], "article": t.tensor([[-1]], device=DEV)}):
        with pytest.raises(IndexError):
            model.validate_features(bad)
    with pytest.raises(ValueError):
        model.validate_features({"customer": t.zeros(3, 3, dtype=t.int64, device=DEV), "article": ok["article"]})


def test_segment_max_backward_is_deterministic_and_exact():
    """aggr="max": many destinations share one arg-max source (hub sources, duplicated edges).  The gradient is reduced
    per source row in CSR order — equal to the float64 scatter, and the same bits on every run (no float atomics)."""
    from laplace_amd import ops
    g = t.Generator().manual_seed(4)
    n_src, n_dst, d, E = 40, 3000, 70, 60000
    src = (t.rand(E, generator=g) ** 3 * n_src).long().clamp(max=n_src - 1)     # a few very popular sources
    dst = t.randint(0, n_dst - 7, (E,), generator=g)
    src, dst = t.cat([src, src[:500]]), t.cat([dst, dst[:500]])                 # duplicated edges
    by_dst = ops.coo_to_csr(dst.to(DEV), src.to(DEV), n_dst, n_src, want_perm=False)
    by_src = ops.coo_to_csr(src.to(DEV), dst.to(DEV), n_src, n_dst, want_perm=False)
    X = t.randn(n_src, d, generator=g)
    X[:3] += 3.0                                                                # sources that win almost everywhere
    Y, arg = ops.segment_max(by_dst, X.to(DEV))
    dense = t.full((n_dst, n_src), float("-inf"))
    dense[dst, src] = 0.0
    want_y = (X[None, :, :] + dense[:, :, None]).amax(dim=1)
    want_y[t.isinf(want_y)] = 0.0                                                # empty destinations -> 0
    assert t.equal(Y.cpu(), want_y)
    dY = t.randn(n_dst, d, generator=g)
    dX = ops.segment_max_bwd(by_src, arg, dY.to(DEV))
    a = arg.cpu().long()
    ref = t.zeros(n_src, d, dtype=t.float64)
    valid = a >= 0
    cols = t.arange(d)[None, :].expand(n_dst, d)
    ref.index_put_((a[valid], cols[valid]), dY.double()[valid], accumulate=True)
    assert int((a[:, 0] == 0).sum()) > 100                                      # the contended case is really there
    mass = t.zeros(n_src, d, dtype=t.float64)
    mass.index_put_((a[valid], cols[valid]), dY.double().abs()[valid], accumulate=True)
    # a source wins up to ~3 000 destinations: fp32 sequential-sum bound = terms * 2^-24 * sum|terms|, in practice far inside it
    assert bool(((dX.cpu().double() - ref).abs() <= 1e-6 * mass + 1e-6).all())
    for _ in range(3):
        assert t.equal(ops.segment_max_bwd(by_src, arg, dY.to(DEV)), dX)


@pytest.mark.parametrize("n,c", [(1, 64), (2, 64), (777, 64), (30011, 64), (500, 37), (3000, 200)])
def test_batchnorm_kernels_match_torch_batchnorm1d(n, c):
    """K8: training-mode forward (batch statistics), running statistics, backward (dx, dgamma, dbeta) and eval mode
    against torch.nn.BatchNorm1d on the CPU; bitwise reproducible."""
    from laplace_amd.model.encoder_decoder import batch_norm
    g = t.Generator().manual_seed(n + c)
    x = (t.randn(n, c, generator=g) * 2.0 + 3.0 * t.randn(1, c, generator=g))
    w = t.randn(n, c, generator=g)
    ref = t.nn.BatchNorm1d(c)
    with t.no_grad():
        ref.weight.copy_(t.rand(c, generator=g) + 0.5)
        ref.bias.copy_(t.randn(c, generator=g))
    mine = t.nn.BatchNorm1d(c)
    mine.load_state_dict(ref.state_dict())
    mine.to(DEV)
    if n == 1:  # torch refuses a single row in training mode; so does the reference's BatchNorm1d
        ref.eval(); mine.eval()
    for it in range(3):
        xr = x.clone().requires_grad_(True)
        xg = x.clone().to(DEV).requires_grad_(True)
        yr, yg = ref(xr), batch_norm(mine, xg)
        assert (yg.detach().cpu() - yr.detach()).abs().max() <= 2e-5
        ref.zero_grad(); mine.zero_grad()
        (yr * w).sum().backward()
        (yg * w.to(DEV)).sum().backward()
        scale = float(xr.grad.abs().max()) + 1e-6
        assert (xg.grad.cpu() - xr.grad).abs().max() <= 1e-4 * scale + 1e-6
        for a, b in ((mine.weight.grad, ref.weight.grad), (mine.bias.grad, ref.bias.grad)):
            assert (a.cpu() - b).abs().max() <= 1e-4 * (float(b.abs().max()) + 1.0)
        for k in ("running_mean", "running_var", "num_batches_tracked"):
            assert t.allclose(getattr(mine, k).cpu().float(), getattr(ref, k).float(), rtol=1e-5, atol=1e-6), k
        again = batch_norm(mine, xg) if not mine.training else None
    ref.eval(); mine.eval()
    assert (batch_norm(mine, x.to(DEV)).cpu() - ref(x)).abs().max() <= 2e-5
    mine.train()
    if n > 1:
        s0 = {k: v.clone() for k, v in mine.state_dict().items()}
        y1 = batch_norm(mine, x.to(DEV))
        mine.load_state_dict(s0)
        y2 = batch_norm(mine, x.to(DEV))
        assert t.equal(y1, y2)


def test_gather_cat_forward_and_deterministic_backward():
    from laplace_amd.model.encoder_decoder import _GatherCatFn
    g = t.Generator().manual_seed(8)
    nu, ni, cu, ci, ne = 40, 300, 64, 48, 2500
    zu, zi = t.randn(nu, cu, generator=g), t.randn(ni, ci, generator=g)
    row = t.randint(0, 5, (ne,), generator=g)          # a handful of users named by hundreds of label edges
    col = t.randint(0, ni, (ne,), generator=g)
    w = t.randn(ne, cu + ci, generator=g)
    a, b = zu.clone().requires_grad_(True), zi.clone().requires_grad_(True)
    (t.cat([a[row], b[col]], dim=-1) * w).sum().backward()
    ag, bg = zu.clone().to(DEV).requires_grad_(True), zi.clone().to(DEV).requires_grad_(True)
    out = _GatherCatFn.apply(ag, bg, row.to(DEV), col.to(DEV))
    assert t.equal(out.detach().cpu(), t.cat([zu[row], zi[col]], dim=-1))
    (out * w.to(DEV)).sum().backward()
    assert (ag.grad.cpu() - a.grad).abs().max() <= 1e-4 * float(a.grad.abs().max())
    assert (bg.grad.cpu() - b.grad).abs().max() <= 1e-5 * float(b.grad.abs().max()) + 1e-6
    assert bool((ag.grad[5:] == 0).all())
    first = ag.grad.clone()
    for _ in range(3):
        ag.grad = None
        (_GatherCatFn.apply(ag, bg, row.to(DEV), col.to(DEV)) * w.to(DEV)).sum().backward()
        assert t.equal(ag.grad, first)


def test_gather_cat_backward_beyond_the_all_pairs_range_is_deterministic_too():
    """More label edges than mi_gather_cat_bwd_max_edges: the sum runs as a product with the sorted incidence CSR
    (edge order inside a node's row), not through torch's atomic index_add_."""
    from laplace_amd import _lib, ops
    ne = int(_lib.lib().mi_gather_cat_bwd_max_edges()) + 1234
    g = t.Generator().manual_seed(9)
    n, cu, ci = 700, 64, 64
    idx = t.randint(0, n, (ne,), generator=g)
    idx[: ne // 3] = 3                                   # one node named by a third of the edges
    d_out = t.randn(ne, cu + ci, generator=g)
    want = t.zeros(n, ci, dtype=t.float64).index_add_(0, idx, d_out[:, cu:].double())
    a = ops.gather_cat_bwd(d_out.to(DEV), idx.to(DEV), n, ci, cu)
    assert (a.cpu().double() - want).abs().max() <= 1e-5 * float(want.abs().max())
    for _ in range(2):
        assert t.equal(ops.gather_cat_bwd(d_out.to(DEV), idx.to(DEV), n, ci, cu), a)


@pytest.mark.parametrize("aggr,embedding", [("add", True), ("mean", True), ("max", False)])
def test_fused_ranker_step_equals_the_autograd_iteration(aggr, embedding):
    """ranker_step.FusedRankerStep: the training iteration as straight-line code (no autograd engine) gives the loss,
    every parameter gradient, the BatchNorm running statistics and the optimizer update of
    `zero_grad -> model(...) -> BCEWithLogitsLoss -> backward -> step` (training.py:19-34) on the same batches; with
    dropout on, it draws its masks from the same torch generator."""
    from laplace_amd.ranker_step import FusedRankerStep
    from laplace_amd.utils.get_info import select_properties
    import copy
    model, loader, first = _hetero_setup(seed=2, aggr=aggr, embedding=embedding, p_drop=0.0)
    twin = copy.deepcopy(model)
    twin.embedding_layers = model.embedding_layers  # frozen tables, shared
    opt_a = t.optim.Adam(model.parameters(), lr=0.01)
    opt_b = t.optim.Adam(twin.parameters(), lr=0.01)
    fused = FusedRankerStep(model, opt_a)
    crit = t.nn.BCEWithLogitsLoss()
    model.train(); twin.train()
    for step, batch in enumerate(loader):
        if step == 4:
            break
        x, ei, eli, y = select_properties(batch.to(DEV))
        la = fused.step({k: v.clone() for k, v in x.items()}, ei, eli, y)
        assert la is not None
        opt_b.zero_grad()
        lb = crit(twin({k: v.clone() for k, v in x.items()}, ei, eli).view(-1), y)
        lb.backward()
        grads_b = {n: p.grad.clone() for n, p in twin.named_parameters()}
        opt_b.step()
        assert abs(float(la) - float(lb)) <= 1e-6
        for n, p in model.named_parameters():
            g = grads_b[n]
            assert p.grad is not None, n
            assert float((p.grad - g).abs().max()) <= 1e-5 * (float(g.abs().max()) + 1e-3), (step, n)
        for bn in ("encoder_layer_norm_customer", "encoder_layer_norm_article"):
            for k in ("running_mean", "running_var", "num_batches_tracked"):
                assert t.allclose(getattr(getattr(model, bn), k).float(), getattr(getattr(twin, bn), k).float(), atol=1e-5, rtol=1e-5)
        # cut the chain (see test_encoder_decoder_logits_gradients_and_training_parity: a gradient at rounding-noise level
        # moves its weight by +-lr whichever sign the noise takes): the twin restarts from the fused model's weights
        twin.load_state_dict(model.state_dict())
    # the first update itself: one step from identical weights with identical optimizers moves every weight alike
    # wherever the gradient is above rounding noise
    # dropout on: runs, learns something finite, masks come from torch's generator (same seed -> same loss)
    model2, loader2, _ = _hetero_setup(seed=5, aggr="add", embedding=True, p_drop=0.3)
    f2 = FusedRankerStep(model2, t.optim.Adam(model2.parameters(), lr=0.01))
    model2.train()
    batch = next(iter(loader2)).to(DEV)
    x, ei, eli, y = select_properties(batch)
    sd = copy.deepcopy(model2.state_dict())
    t.manual_seed(77)
    l1 = float(f2.step({k: v.clone() for k, v in x.items()}, ei, eli, y))
    model2.load_state_dict(sd)
    f2.optimizer = t.optim.Adam(model2.parameters(), lr=0.01)
    t.manual_seed(77)
    l2 = float(f2.step({k: v.clone() for k, v in x.items()}, ei, eli, y))
    assert np.isfinite(l1) and l1 == l2


# ---- the iteration as ONE C call (mi_ranker_step_f32, ranker_native.NativeRankerStep) ------------------------------------------

@pytest.mark.parametrize("aggr", ["add", "mean"])
def test_native_ranker_step_equals_the_fused_step(aggr):
    """The native executor issues FusedRankerStep's launches itself: same loss, same gradients (bit for bit: the same
    kernels on the same operands), same BatchNorm statistics; its multi-tensor Adam gives torch.optim.Adam's update.
    The executor keeps the optimizer's own state tensors up to date (exp_avg, exp_avg_sq, step)."""
    from laplace_amd.ranker_native import NativeRankerStep
    from laplace_amd.ranker_step import FusedRankerStep
    from laplace_amd.utils.get_info import select_properties
    import copy
    model, loader, first = _hetero_setup(seed=3, aggr=aggr, embedding=True, p_drop=0.0)
    twin = copy.deepcopy(model)
    twin.embedding_layers = model.embedding_layers
    opt_a = t.optim.Adam(model.parameters(), lr=0.01)
    opt_b = t.optim.Adam(twin.parameters(), lr=0.01)
    assert NativeRankerStep.unsupported_reason(model, opt_a) is None
    native, fused = NativeRankerStep(model, opt_a), FusedRankerStep(twin, opt_b)
    model.train(); twin.train()
    for step, batch in enumerate(loader):
        if step == 5:
            break
        x, ei, eli, y = select_properties(batch.to(DEV))
        if step == 2:   # the caller drops the gradient tensors: the executor's descriptor (raw pointers) must notice
            opt_a.zero_grad(set_to_none=True)
        la = native.step({k: v.clone() for k, v in x.items()}, ei, eli, y)
        lb = fused.step({k: v.clone() for k, v in x.items()}, ei, eli, y)
        assert la is not None and lb is not None, native.declined
        assert float(la) == float(lb), step
        gb = dict(twin.named_parameters())
        for n, p in model.named_parameters():
            assert p.grad is not None and t.equal(p.grad, gb[n].grad), (step, n)
        for bn in ("encoder_layer_norm_customer", "encoder_layer_norm_article"):
            for k in ("running_mean", "running_var", "num_batches_tracked"):
                assert t.equal(getattr(getattr(model, bn), k), getattr(getattr(twin, bn), k)), (step, bn, k)
        for (n, p), q in zip(model.named_parameters(), twin.parameters()):
            assert float((p - q).abs().max()) <= 2e-6, (step, n)           # Adam: same update to rounding
            sa, sb = opt_a.state[p], opt_b.state[q]
            assert float(sa["step"]) == float(sb["step"]) == step + 1
            assert t.allclose(sa["exp_avg"], sb["exp_avg"], rtol=1e-5, atol=1e-9)
            assert t.allclose(sa["exp_avg_sq"], sb["exp_avg_sq"], rtol=1e-5, atol=1e-12)
        twin.load_state_dict(model.state_dict())   # cut the chain at rounding level, as the tests above do
        for (p, q) in zip(model.parameters(), twin.parameters()):
            for k in ("exp_avg", "exp_avg_sq"):
                opt_b.state[q][k].copy_(opt_a.state[p][k])


def test_native_ranker_step_with_the_device_sampler_and_training_loop():
    """training.train_with_dataloader picks the native executor for the default model; batches come from the device
    sampler with its emitted CSRs (no sort in the loop); the loss falls."""
    from laplace_amd import synthetic as S
    from laplace_amd.data.device_sampler import DeviceGraphSampler
    from laplace_amd.model.encoder_decoder import Encoder_Decoder_Model
    from laplace_amd.model.layers import get_SAGEConv_layers, get_linear_layers
    from laplace_amd.ranker_native import NativeRankerStep
    from laplace_amd.training import train_with_dataloader
    from laplace_amd.utils.get_info import get_feature_info
    spec = S.SyntheticSpec(3000, 500, 40_000, seed=4, deg_min=2, deg_max=200)
    graph, users, articles = S.generate_hetero(spec, customer_cards=(300, 2, 84, 4, 5, 2), article_cards=(100, 132, 30, 50))
    cfg = SimpleNamespace(k=12, num_neighbors=16, n_hop_neighbors=2, positive_edges_ratio=0.5, negative_edges_ratio=3.0, batch_size=24)
    loader = DeviceGraphSampler(cfg, graph, users, articles, device=DEV, seed=0)
    first = next(iter(loader))
    t.manual_seed(0)
    model = Encoder_Decoder_Model(get_SAGEConv_layers(2, 128, 64, "add"), get_linear_layers(2, 128, 128, 1), get_feature_info(graph),
                                  first.metadata(), True, "sum", True, 0.0, 0.3).to(DEV)
    model.initialize_encoder_input_size(first)
    opt = t.optim.Adam(model.parameters(), lr=0.01)
    assert NativeRankerStep.supports(model, opt)
    model.train()
    losses = train_with_dataloader(model, opt, loader, 0, DEV)
    assert len(losses) >= 20 and all(np.isfinite(losses))
    assert np.mean(losses[-5:]) < np.mean(losses[:5])
    assert float(opt.state[next(model.parameters())]["step"]) == len(losses)   # every iteration went through an Adam update
    assert int(model.encoder_layer_norm_customer.num_batches_tracked) == len(losses) + 0   # ... and through the BatchNorm


def test_native_ranker_dropout_is_reproducible_and_its_backward_uses_the_forward_masks():
    """Philox feature dropout inside the executor: (seed, iteration) fixes the masks — two runs give the same bits — and
    the gradient it reports is the gradient of THAT masked network: central differences of the loss on a few weights,
    with the learning rate at zero so that nothing moves."""
    from laplace_amd.ranker_native import NativeRankerStep
    from laplace_amd.utils.get_info import select_properties
    model, loader, first = _hetero_setup(seed=6, aggr="add", embedding=True, p_drop=0.3)
    opt = t.optim.Adam(model.parameters(), lr=0.0)
    native = NativeRankerStep(model, opt, seed=1234)
    model.train()
    batch = next(iter(loader)).to(DEV)
    x, ei, eli, y = select_properties(batch)

    def loss_at(iteration):
        native.iteration = iteration
        return native.step({k: v.clone() for k, v in x.items()}, ei, eli, y)

    l0 = float(loss_at(7))
    g0 = {n: p.grad.clone() for n, p in model.named_parameters()}
    assert float(loss_at(7)) == l0 and all(t.equal(p.grad, g0[n]) for n, p in model.named_parameters())
    assert float(loss_at(8)) != l0                                     # another iteration draws other masks
    # finite differences on the largest-gradient entries of three weight tensors
    checked = 0
    for name, p in model.named_parameters():
        if not (name.endswith("lin_l.weight") or name.endswith("layers.0.weight")):
            continue
        g = g0[name]
        idx = int(g.abs().argmax())
        eps = 2e-2
        with t.no_grad():
            flat = p.view(-1)
            orig = float(flat[idx])
            flat[idx] = orig + eps
            lp = float(loss_at(7))
            flat[idx] = orig - eps
            lm = float(loss_at(7))
            flat[idx] = orig
        fd = (lp - lm) / (2 * eps)
        assert abs(fd - float(g.view(-1)[idx])) <= 0.15 * abs(float(g.view(-1)[idx])) + 2e-4, (name, fd, float(g.view(-1)[idx]))
        checked += 1
    assert checked >= 3


def test_native_data_parallel_step_world_one_is_the_native_step():
    """data_parallel=True with one process: executor without its Adam + mi_ranker_adam_f32(scale 1) must give exactly the
    single-launch iteration; every .grad is a view into the flat buffer the collective would exchange."""
    from laplace_amd.ranker_native import NativeRankerStep
    from laplace_amd.utils.get_info import select_properties
    import copy
    model, loader, first = _hetero_setup(seed=5, aggr="add", embedding=True, p_drop=0.2)
    twin = copy.deepcopy(model)
    twin.embedding_layers = model.embedding_layers
    opt_a, opt_b = t.optim.Adam(model.parameters(), lr=0.01), t.optim.Adam(twin.parameters(), lr=0.01)
    dp, plain = NativeRankerStep(model, opt_a, data_parallel=True, seed=11), NativeRankerStep(twin, opt_b, seed=11)
    model.train(); twin.train()
    for step, batch in enumerate(loader):
        if step == 4:
            break
        x, ei, eli, y = select_properties(batch.to(DEV))
        la = dp.step({k: v.clone() for k, v in x.items()}, ei, eli, y)
        lb = plain.step({k: v.clone() for k, v in x.items()}, ei, eli, y)
        assert la is not None and lb is not None, (dp.declined, plain.declined)
        assert float(la) == float(lb)
        flat = dp.flat_grads
        lo, hi = flat.data_ptr(), flat.data_ptr() + 4 * flat.numel()
        for (n, p), q in zip(model.named_parameters(), twin.parameters()):
            assert lo <= p.grad.data_ptr() < hi, n
            assert t.equal(p.grad, q.grad), (step, n)
            assert t.equal(p, q), (step, n)
            assert t.equal(opt_a.state[p]["exp_avg_sq"], opt_b.state[q]["exp_avg_sq"])
            assert float(opt_a.state[p]["step"]) == step + 1
    with pytest.raises(ValueError):
        NativeRankerStep(model, opt_a, data_parallel=True, before_step=lambda: None)


def _dp_worker(rank, world, port, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)   # both ranks share the one card: gloo carries the exchange
    try:
        from laplace_amd.dist_ranker import broadcast_parameters, native_step
        from laplace_amd.ranker_native import NativeRankerStep
        from laplace_amd.utils.get_info import select_properties
        import copy
        t.manual_seed(50 + rank)                                   # different initial weights on purpose
        model, loader, first = _hetero_setup(seed=3, aggr="add", embedding=True, p_drop=0.0)
        broadcast_parameters(model)
        for tabs in model.embedding_layers.values():               # the frozen tables are "identical by construction"
            for tb in tabs:
                dist.broadcast(tb.data, src=0)
        local = copy.deepcopy(model)
        local.embedding_layers = model.embedding_layers
        opt, opt_l = t.optim.Adam(model.parameters(), lr=0.01), t.optim.Adam(local.parameters(), lr=0.01)
        step = native_step(model, opt, seed=1)
        probe = NativeRankerStep(local, opt_l, before_step=lambda: None, seed=1)   # this rank's own gradients
        model.train(); local.train()
        batches = [b for i, b in zip(range(2 * world), loader)]
        x, ei, eli, y = select_properties(batches[rank].to(DEV))  # a different batch per rank
        before = [p.detach().clone() for p in model.parameters()]
        loss = step.step({k: v.clone() for k, v in x.items()}, ei, eli, y)
        assert loss is not None, step.declined
        probe.step({k: v.clone() for k, v in x.items()}, ei, eli, y)
        t.cuda.synchronize()
        ret[rank] = {"params": [p.detach().cpu() for p in model.parameters()], "before": [b.cpu() for b in before],
                     "sum_grads": [p.grad.detach().cpu() for p in model.parameters()],
                     "own_grads": [p.grad.detach().cpu() for p in local.parameters()],
                     "m": [opt.state[p]["exp_avg"].cpu() for p in model.parameters()]}
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_native_data_parallel_step_two_ranks_one_collective():
    """Two processes on the card, a different batch each: after the step the flat gradient buffer holds the SUM of the two
    ranks' own gradients, both replicas hold the same parameters, and the update is Adam's on the MEAN gradient."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_dp_worker, args=(2, port, ret), nprocs=2, join=True)
    a, b = ret[0], ret[1]
    for i in range(len(a["params"])):
        assert t.equal(a["before"][i], b["before"][i])
        assert t.equal(a["sum_grads"][i], b["sum_grads"][i])
        assert t.allclose(a["sum_grads"][i], a["own_grads"][i] + b["own_grads"][i], rtol=1e-6, atol=1e-7)
        assert t.equal(a["params"][i], b["params"][i])
        g = 0.5 * a["sum_grads"][i]
        assert t.allclose(a["m"][i], 0.1 * g, rtol=1e-5, atol=1e-9)                     # exp_avg after step 1
        want = a["before"][i] - 0.01 * g / (g.abs() + 1e-8)                              # Adam step 1: lr * g / (|g| + eps)
        assert t.allclose(a["params"][i], want, rtol=1e-4, atol=1e-6)
    assert not all(t.equal(x, y) for x, y in zip(a["own_grads"], b["own_grads"]))        # the batches did differ


@pytest.mark.parametrize("n,k", [(1, 128), (63, 128), (64, 8), (1100, 128), (5000, 300)])
def test_linear1_bwd_matches_torch_linear_backward(n, k):
    """mi_linear1_bwd_f32 against autograd of torch.nn.functional.linear with one output feature; deterministic."""
    from laplace_amd import ops
    g = t.Generator(device=DEV).manual_seed(n + k)
    x = t.randn(n, k, device=DEV, generator=g)
    w = t.randn(1, k, device=DEV, generator=g).requires_grad_()
    b = t.zeros(1, device=DEV, requires_grad=True)
    xr = x.clone().requires_grad_()
    dy = t.randn(n, 1, device=DEV, generator=g)
    t.nn.functional.linear(xr, w, b).backward(dy)
    dx, gw, gb = ops.linear1_bwd(dy.reshape(-1), w.detach(), x)
    assert t.equal(dx, xr.grad)                                    # one multiply per element: exact
    scale = float(w.grad.abs().max()) + 1e-12
    assert float((gw - w.grad).abs().max()) <= 1e-5 * scale
    assert abs(float(gb) - float(b.grad)) <= 1e-5 * (float(dy.abs().sum()) + 1e-12)
    dx2, gw2, gb2 = ops.linear1_bwd(dy.reshape(-1), w.detach(), x)
    assert t.equal(gw, gw2) and t.equal(gb, gb2)
    _, gw3, _ = ops.linear1_bwd(dy.reshape(-1), w.detach(), x, need_dx=False)
    assert t.equal(gw, gw3)

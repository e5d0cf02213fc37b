"""End to end on one GPU at small scale (BASELINE.json configs[3] flow): LightGCN candidate generation ->
top-N dump -> LightGCN matcher -> ranker trained on device-sampled batches -> evaluated on matcher candidates."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch as t

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_candidate_generation_feeds_the_ranker():
    from laplace_amd import synthetic as S
    from laplace_amd.config import LightGCNConfig
    from laplace_amd.data import graph_io
    from laplace_amd.data.dataset import GraphDataset
    from laplace_amd.data.device_sampler import DeviceGraphSampler
    from laplace_amd.data.matching import LightGCNMatcher, PopularItemsMatcher
    from laplace_amd.hetero import DataLoader
    from laplace_amd.interactions import Interactions
    from laplace_amd.model.encoder_decoder import Encoder_Decoder_Model
    from laplace_amd.model.layers import get_SAGEConv_layers, get_linear_layers
    from laplace_amd.model.lightgcn import LightGCN
    from laplace_amd.run_pipeline_lightgcn import save_predictions
    from laplace_amd.trainer import LightGCNTrainer
    from laplace_amd.training import test_with_dataloader, train_with_dataloader
    from laplace_amd.utils.constants import Constants
    from laplace_amd.utils.get_info import get_feature_info

    # transactions in "time" order -> chronological train / val / test graphs (N2)
    spec = S.SyntheticSpec(800, 300, 12_000, seed=11, deg_min=4, deg_max=120)
    graph, _, _ = S.generate_hetero(spec, customer_cards=(200, 2, 84, 4), article_cards=(120, 30))
    ei = graph[Constants.edge_key].edge_index
    splits = graph_io.build_splits(graph[Constants.node_user].x, graph[Constants.node_item].x, ei[0].numpy(), ei[1].numpy())
    train_graph, train_users, train_articles = splits["train"]
    train_ei = train_graph[Constants.edge_key].edge_index

    # 1. candidate generation: LightGCN on the train edges, fused step, on-device sampling
    t.manual_seed(0)
    U, I = spec.num_users, spec.num_items
    lgcn = LightGCN(U, I, 32, 3).to(DEV)
    inter = Interactions(train_ei.to(DEV), U, I)
    trainer = LightGCNTrainer(lgcn, inter.adjacency("bipartite"), inter, lr=5e-3, Lambda=1e-6, batch_size=1024, seed=1)
    losses = [float(trainer.step()) for _ in range(80)]
    assert np.mean(losses[-10:]) < np.mean(losses[:10])
    top = save_predictions(lgcn, train_ei.to(DEV), num_recommendations=40)   # [U, 40], seen items excluded
    assert top.shape == (U, 40)
    seen = set(zip(train_ei[0].tolist(), train_ei[1].tolist()))
    assert not any((u, int(i)) in seen for u in range(0, U, 37) for i in top[u].tolist())

    # 2. ranker trained from device-sampled subgraphs of the train graph
    cfg = SimpleNamespace(k=5, num_neighbors=8, n_hop_neighbors=2, positive_edges_ratio=0.5, negative_edges_ratio=3.0,
                          batch_size=32)
    sampler = DeviceGraphSampler(cfg, train_graph, train_users, train_articles, device=DEV, seed=4)
    first = sampler.sample(t.arange(32), step=0)
    ranker = Encoder_Decoder_Model(get_SAGEConv_layers(2, 64, 32, "add"), get_linear_layers(2, 64, 64, 1),
                                   get_feature_info(train_graph), first.metadata(), True, "sum", True, 0.0, 0.2).to(DEV)
    ranker.initialize_encoder_input_size(first)
    opt = t.optim.Adam(ranker.parameters(), lr=0.01)
    first_epoch = train_with_dataloader(ranker, opt, sampler, 0, DEV)
    for ep in range(1, 3):
        last_epoch = train_with_dataloader(ranker, opt, sampler, ep, DEV)
    assert np.mean(last_epoch) < np.mean(first_epoch)

    # 3. evaluation on the val graph: candidates = LightGCN top-N + popular items (matchers)
    val_graph, val_users, val_articles = splits["val"]
    matchers = [LightGCNMatcher(top, 10), PopularItemsMatcher.from_adjacency(val_articles, 10)]
    val_ds = GraphDataset(cfg, val_graph, val_users, val_articles, train=False, matchers=matchers, seed=5)
    val_loader = DataLoader(val_ds, batch_size=16, shuffle=False)
    recall, precision = test_with_dataloader("VAL", ranker, val_loader, DEV, k=cfg.k, break_at=6)
    assert 0.0 <= recall <= 1.0 and 0.0 <= precision <= 1.0

    # 4. submission (row N4): newest checkpoint -> every customer of the test split -> top-k file
    import os, tempfile
    import pandas as pd
    from laplace_amd import run_submission as RS
    test_graph, test_users, test_articles = splits["test"]
    t_matchers = [LightGCNMatcher(top, 10), PopularItemsMatcher.from_adjacency(test_articles, 10)]
    sub_cfg = SimpleNamespace(**vars(cfg), num_gnn_layers=2, hidden_layer_size=64, encoder_layer_output_size=32,
                              conv_agg_type="add", num_linear_layers=2, heterogeneous_prop_agg_type="sum", batch_norm=True,
                              p_dropout_edges=0.0, p_dropout_features=0.2)
    sub_cfg.batch_size = 16
    with tempfile.TemporaryDirectory() as tmp:
        t.save({"stale": t.zeros(1)}, os.path.join(tmp, "model_001.pt"))
        t.save(ranker.state_dict(), os.path.join(tmp, "model_007.pt"))
        state = RS.load_model(tmp)
        assert "stale" not in state  # the largest version number wins
        out_csv = os.path.join(tmp, "derived", "submission.csv")
        cmap = {str(u): f"cust{u:05d}" for u in range(U)}
        amap = {str(a): f"art{a:04d}" for a in range(I)}
        customers, preds, df = RS.submission_pipeline(sub_cfg, splits=splits, matchers=t_matchers, model_dir=tmp,
                                                      out_csv=out_csv, customer_id_map=cmap, article_id_map=amap,
                                                      device=DEV, seed=5)
        back = pd.read_csv(out_csv)
    assert t.equal(customers, t.arange(U)) and preds.shape == (U, cfg.k)
    assert int(preds.max()) < I and int(preds.min()) >= -1
    for row in preds[::23].tolist():
        real = [a for a in row if a >= 0]
        assert len(real) == len(set(real)) and len(real) >= 1
    assert list(back.columns) == ["customer_id", "prediction"] and len(back) == U
    assert back.customer_id[3] == "cust00003" and back.prediction[3].split()[0] == f"art{int(preds[3, 0]):04d}"
    # the first pick is the best label-0 candidate of the rebuilt model's own scores
    test_ds = GraphDataset(sub_cfg, test_graph, test_users, test_articles, train=False, matchers=t_matchers, seed=5)
    sample = test_ds[3].to(DEV)
    ranker.eval()
    with t.no_grad():
        eli = sample[Constants.edge_key].edge_label_index
        sc = ranker(sample.x_dict, sample.edge_index_dict, eli).view(-1)
        lab0 = sample[Constants.edge_key].edge_label == 0
        best = sample[Constants.node_item].n_id[eli[1][lab0][sc[lab0].argmax()]]
    assert int(best) == int(preds[3, 0])
    # the same file from device-built evaluation samples: every customer once, in order, valid distinct ids
    with tempfile.TemporaryDirectory() as tmp:
        t.save(ranker.state_dict(), os.path.join(tmp, "model_007.pt"))
        c2, p2, _ = RS.submission_pipeline(sub_cfg, splits=splits, matchers=t_matchers, model_dir=tmp,
                                           out_csv=os.path.join(tmp, "s.csv"), device=DEV, seed=5, device_sampler=True)
    assert t.equal(c2, t.arange(U)) and p2.shape == (U, cfg.k) and int(p2.max()) < I and int(p2.min()) >= -1
    for row in p2[::31].tolist():
        real = [a for a in row if a >= 0]
        assert len(real) == len(set(real)) and len(real) >= 1
    # the candidate sets are the same whichever sampler built the batch: every pick is a candidate or a missed purchase
    cand3 = set(np.concatenate([np.asarray(m.get_matches(3)) for m in t_matchers]).tolist()) | set(np.asarray(test_users[3]).tolist())
    assert set(a for a in p2[3].tolist() if a >= 0) <= cand3 and set(a for a in preds[3].tolist() if a >= 0) <= cand3
    # MAP@k of the file against the test purchases is a number in [0, 1]
    from laplace_amd.utils.metrics import MAPatK
    gt = [t.from_numpy(np.asarray(test_users[u])) for u in range(U)]
    assert 0.0 <= MAPatK(gt, preds, k=cfg.k) <= 1.0


def test_device_built_inference_batches_pipelined_and_selected_without_host_waits():
    """run_submission.make_predictions on device-built evaluation samples (round 4): `DeviceGraphSampler.iter_users` yields the
    batches of a user subset through the pipelined iterator — the same batches `sample(users[i*B:(i+1)*B], step=i)` builds —
    and the per-customer top-k selection runs without a host read (rows by searchsorted over the samples' node offsets).
    Same customers, same predictions as the generic selection path on the same batches."""
    from types import SimpleNamespace
    from laplace_amd import run_submission as RS, synthetic as S
    from laplace_amd.data.device_sampler import DeviceGraphSampler
    from laplace_amd.data.matching import PopularItemsMatcher
    from laplace_amd.model.encoder_decoder import Encoder_Decoder_Model
    from laplace_amd.model.layers import get_SAGEConv_layers, get_linear_layers
    from laplace_amd.utils.constants import Constants
    from laplace_amd.utils.get_info import get_feature_info
    spec = S.SyntheticSpec(20_000, 3_000, 300_000, seed=5, zipf_s=1.0, communities=8, community_mix=0.8)
    hetero, users_adj, articles_adj = S.generate_hetero(spec, feature_signal=True)
    cfg = SimpleNamespace(k=12, num_neighbors=16, n_hop_neighbors=2, positive_edges_ratio=0.5, negative_edges_ratio=3.0, batch_size=32,
                          num_gnn_layers=2, hidden_layer_size=32, encoder_layer_output_size=16, conv_agg_type="add", num_linear_layers=2,
                          heterogeneous_prop_agg_type="sum", batch_norm=True, p_dropout_edges=0.0, p_dropout_features=0.0)
    matchers = [PopularItemsMatcher.from_adjacency(articles_adj, 40)]
    ev = DeviceGraphSampler(cfg, hetero, users_adj, articles_adj, device=DEV, seed=4, train=False, matchers=matchers, shuffle=False)
    g = t.Generator().manual_seed(0)
    users = t.randperm(20_000, generator=g)[:200]                      # 7 batches, the last one short
    ev.step = 0
    piped = list(ev.iter_users(users))
    assert len(piped) == 7 and ev.step == 7
    for i, b in enumerate(piped):
        ref = ev.sample(users[32 * i:32 * i + 32], step=i)
        for key in (Constants.node_user, Constants.node_item):
            assert t.equal(b[key].n_id, ref[key].n_id) and t.equal(b[key].x, ref[key].x)
        assert t.equal(b[Constants.edge_key].edge_index, ref[Constants.edge_key].edge_index)
        assert t.equal(b[Constants.edge_key].edge_label_index, ref[Constants.edge_key].edge_label_index)
        assert t.equal(b[Constants.edge_key].edge_label, ref[Constants.edge_key].edge_label)
        assert t.equal(b._seed_users.cpu(), users[32 * i:32 * i + 32])
    t.manual_seed(3)
    model = Encoder_Decoder_Model(get_SAGEConv_layers(2, 32, 16, "add"), get_linear_layers(2, 32, 32, 1), get_feature_info(hetero),
                                  piped[0].metadata(), True, "sum", True, 0.0, 0.0).to(DEV)
    model.initialize_encoder_input_size(piped[0])
    from laplace_amd.ranker_native import NativeRankerForward
    from laplace_amd.utils.get_info import select_properties
    # give BatchNorm running statistics that are not the initial (0, 1): a few training-mode forwards
    model.train()
    with t.no_grad():
        for b in piped[:3]:
            x, eid, eli, _ = select_properties(b)
            model(dict(x), eid, eli)
    model.eval()
    assert NativeRankerForward.supports(model)
    nf = NativeRankerForward(model)
    with t.no_grad():
        # the evaluation forward as ONE C call (mi_ranker_batch.logits) against the op-by-op forward, batch by batch
        for b in piped:
            x, eid, eli, _ = select_properties(b)
            got = nf.logits(x, eid, eli)
            want = model(dict(x), eid, eli).view(-1)
            assert got is not None and got.shape == want.shape
            assert (got - want).abs().max() <= 1e-5 * max(1.0, float(want.abs().max())), float((got - want).abs().max())
        saved = RS.FAST_SELECT, RS.NATIVE_FORWARD
        try:
            RS.NATIVE_FORWARD = False
            ev.step = 0
            c_fast, p_fast = RS.make_predictions(model, ev.iter_users(users), k=12, device=DEV)
            RS.FAST_SELECT = False
            ev.step = 0
            c_slow, p_slow = RS.make_predictions(model, ev.iter_users(users), k=12, device=DEV)
            RS.FAST_SELECT, RS.NATIVE_FORWARD = True, True
            ev.step = 0
            c_nat, p_nat = RS.make_predictions(model, ev.iter_users(users), k=12, device=DEV)
        finally:
            RS.FAST_SELECT, RS.NATIVE_FORWARD = saved
    assert t.equal(c_fast, users) and t.equal(c_slow, users) and t.equal(c_nat, users)
    assert p_fast.shape == (200, 12) and t.equal(p_fast, p_slow)
    assert int((p_fast >= 0).sum()) > 200 * 6                          # real candidates were ranked
    # the native forward's scores differ from the op-by-op ones in the last bits: the same SETS of 12 up to near-ties
    same = sum(len(set(a) & set(b)) for a, b in zip(p_nat.tolist(), p_fast.tolist()))
    assert same >= 0.98 * 12 * 200, same


def test_configs3_chain_at_one_tenth_scale_keeps_its_ranking_quality_across_the_hand_off():
    """BASELINE configs[3] ("LightGCN candidate-gen + GNN ranker end-to-end") — the chain bench.py's `e2e_c3` block runs at the
    full H&M shape (tools/e2e_hm_scale.py::run), here at 1/10 of it (137 198 x 10 554 x 3.18 M, planted structure, one held-out
    purchase per evaluation user): LightGCN steps -> top-100 dump -> matchers -> ranker iterations on device-sampled batches ->
    device-built evaluation samples -> top-12.  The hand-off must keep what the candidate generator learned: the matchers'
    candidates contain the held-out purchase far more often than popularity alone would put it there, and the ranker's
    re-ranking of ~150 candidates scores well above a shuffle of the same candidates."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("e2e_hm_scale", os.path.join(root, "tools", "e2e_hm_scale.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = mod.run(users=137_198, items=10_554, edges=3_180_000, lightgcn_steps=300, ranker_iters=400, ranker_batch=128,
                  eval_users=4_000, top_n=100, popular_n=50)
    m = out["map_at_12"]
    print(out)
    assert out["eval_users"] >= 3_900
    for key in ("generate_s", "lightgcn_train_s", "topn_dump_s", "matchers_s", "ranker_train_s", "eval_inference_s"):
        assert out["stage_s"][key] >= 0.0
    assert out["lightgcn_positive_edges_per_s"] > 1e5 and out["ranker_positive_edges_per_s"] > 1e4
    # a uniformly shuffled list of ~150 candidates that contains the item puts it into the top 12 with p ~ 12 / 150 at a mean
    # reciprocal rank of ~0.26: MAP ~ 0.02 x recall
    shuffled = 0.02 * m["candidate_recall"]
    assert m["candidate_recall"] >= 0.15, m
    assert m["candidate_generator_alone"] >= 3 * shuffled, m
    assert m["ranker_reranked_candidates"] >= 3 * shuffled, m
    assert m["ranker_reranked_candidates"] <= m["candidate_recall"] + 1e-6     # it can only rank what was proposed

#!/usr/bin/env python3
"""BASELINE.json configs[3] end to end on one MI355X — "LightGCN candidate-gen + GNN ranker" at the H&M shape (SURVEY C3:
1 371 980 customers x 105 542 articles x 31.8 M transactions), the flow of the reference's run_pipeline_lightgcn.py ->
data/matching/lightgcn.py:5-11 -> run_pipeline.py:24-153 -> run_submission.py:48-69:

  1. LightGCN candidate generation: fused train steps (on-device sampling, BPR, Adam in the backward epilogue)
  2. top-N dump for every customer, purchases excluded (exact top-K with exclusion)
  3. matchers: LightGCN top-N + popular items, answered on the device
  4. ranker training on device-sampled 2-hop subgraphs (one C call per iteration, sampling overlapped)
  5. evaluation samples built on device from the matchers' candidates, ranker inference, top-12
  6. MAP@12 on HELD-OUT purchases (one per evaluation user, drawn from the generator's own law and absent from the graph):
     the candidate generator alone (its own first 12), the ranker's re-ranking of the candidates, the candidates' recall.

The graph carries PLANTED structure (synthetic.SyntheticSpec.communities: latent user / item groups) and, like a real
dataset, node features that say something about it (one customer column and one article column hold the group) — otherwise
there is nothing for either model to learn beyond popularity.  `run(...)` is what bench.py's `e2e_c3` block and
tests/test_gpu_end_to_end.py (at 1/10 scale) call; as a script it prints one JSON line.
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(users=1_371_980, items=105_542, edges=31_800_000, lightgcn_steps=300, lightgcn_dim=64, lightgcn_lr=0.05,
        lightgcn_batch=16384, ranker_iters=300, ranker_batch=128, eval_users=20_000, top_n=100, popular_n=50,
        communities=32, community_mix=0.9, seed=2, graph=None) -> dict:
    import numpy as np
    import torch as t
    from laplace_amd import run_submission as RS
    from laplace_amd import synthetic as S
    from laplace_amd.data.device_sampler import DeviceGraphSampler
    from laplace_amd.data.matching import LightGCNMatcher, PopularItemsMatcher
    from laplace_amd.interactions import Interactions
    from laplace_amd.model.encoder_decoder import Encoder_Decoder_Model
    from laplace_amd.model.layers import get_SAGEConv_layers, get_linear_layers
    from laplace_amd.model.lightgcn import LightGCN
    from laplace_amd.run_pipeline_lightgcn import save_predictions
    from laplace_amd.trainer import LightGCNTrainer
    from laplace_amd.utils.constants import Constants
    from laplace_amd.utils.get_info import get_feature_info, select_properties

    dev = "cuda"
    sync = t.cuda.synchronize
    stage = {}
    out = {"workload": (f"BASELINE configs[3] end to end on one MI355X: H&M-shaped synthetic {users}x{items}, {edges} transactions, "
                        f"{communities} planted user / item groups (own-group purchases with p = {community_mix}), one customer and "
                        f"one article feature column carrying the group; LightGCN 3-layer D={lightgcn_dim} ({lightgcn_steps} steps of "
                        f"{lightgcn_batch}) -> top-{top_n} dump -> matchers -> ranker ({ranker_iters} iterations of {ranker_batch} "
                        f"users, 2 hops, fan-out 64) -> device-built evaluation of {eval_users} users with one held-out purchase each")}

    t0 = time.perf_counter()
    spec = S.SyntheticSpec(users, items, edges, seed=seed, zipf_s=1.0, communities=communities, community_mix=community_mix)
    if graph is None:
        graph = S.generate_hetero(spec, feature_signal=True)
    hetero, users_adj, articles_adj = graph
    ei = hetero[Constants.edge_key].edge_index
    held = S.heldout_edges(spec, ei, eval_users)            # [2, n]: (user, item) pairs NOT in the graph
    stage["generate_s"] = round(time.perf_counter() - t0, 1)
    U, I = users, items

    # 1. LightGCN candidate generation
    t.manual_seed(0)
    t0 = time.perf_counter()
    lgcn = LightGCN(U, I, lightgcn_dim, 3).to(dev)
    inter = Interactions(ei.to(dev), U, I)
    trainer = LightGCNTrainer(lgcn, inter.adjacency("bipartite"), inter, lr=lightgcn_lr, Lambda=1e-6, batch_size=lightgcn_batch, seed=1)
    sync()
    stage["lightgcn_setup_s"] = round(time.perf_counter() - t0, 2)
    for _ in range(5):
        trainer.step()
    sync()
    t0 = time.perf_counter()
    for _ in range(lightgcn_steps):
        trainer.step()
    sync()
    dt = time.perf_counter() - t0
    stage["lightgcn_train_s"] = round(dt, 2)
    out["lightgcn_ms_per_step"] = round(1e3 * dt / max(lightgcn_steps, 1), 3)
    out["lightgcn_positive_edges_per_s"] = round(lightgcn_batch * lightgcn_steps / dt)

    # 2. top-N dump, purchases excluded (the trainer relabelled the nodes for locality: rows back under their ids first)
    trainer.finish()
    t0 = time.perf_counter()
    top = save_predictions(lgcn, ei.to(dev), num_recommendations=top_n)
    sync()
    dt = time.perf_counter() - t0
    stage["topn_dump_s"] = round(dt, 2)
    out["topn_users_per_s"] = round(U / dt)
    del trainer, inter
    t.cuda.empty_cache()

    # 3. matchers (answered on the device, N3)
    t0 = time.perf_counter()
    matchers = [LightGCNMatcher(top, top_n), PopularItemsMatcher.from_adjacency(articles_adj, popular_n)]
    stage["matchers_s"] = round(time.perf_counter() - t0, 2)

    # 4. ranker on device-sampled batches
    cfg = SimpleNamespace(k=12, num_neighbors=64, n_hop_neighbors=2, positive_edges_ratio=0.5, negative_edges_ratio=3.0,
                          batch_size=ranker_batch, num_gnn_layers=2, hidden_layer_size=128, encoder_layer_output_size=64,
                          conv_agg_type="add", num_linear_layers=2, heterogeneous_prop_agg_type="sum", batch_norm=True,
                          p_dropout_edges=0.0, p_dropout_features=0.3)
    t0 = time.perf_counter()
    sampler = DeviceGraphSampler(cfg, hetero, users_adj, articles_adj, device=dev, seed=3)
    it = iter(sampler)
    first = next(it)
    ranker = Encoder_Decoder_Model(get_SAGEConv_layers(2, 128, 64, "add"), get_linear_layers(2, 128, 128, 1),
                                   get_feature_info(hetero), first.metadata(), True, "sum", True, 0.0, 0.3).to(dev)
    ranker.initialize_encoder_input_size(first)
    opt = t.optim.Adam(ranker.parameters(), lr=0.01)     # the reference's optimizer (run_pipeline.py)
    ranker.train()
    from laplace_amd.ranker_native import NativeRankerStep
    from laplace_amd.ranker_step import FusedRankerStep
    native = NativeRankerStep(ranker, opt)   # what training.train_with_dataloader runs per batch: one C call per iteration
    fused = FusedRankerStep(ranker, opt)

    path = {"native": 0, "fused": 0}
    declined = {}

    def step(batch):
        lab = batch[Constants.edge_key]
        loss = native.step(batch.x_dict, batch.edge_index_dict, lab.edge_label_index, lab.edge_label)
        if loss is not None:
            path["native"] += 1
            return loss
        path["fused"] += 1
        declined[native.declined] = declined.get(native.declined, 0) + 1
        return fused.step(*select_properties(batch))

    for _ in range(5):
        step(next(it))
    sync()
    stage["ranker_setup_s"] = round(time.perf_counter() - t0, 2)
    labels, t0 = [], time.perf_counter()
    for _ in range(ranker_iters):
        b = next(it)
        rl = step(b)
        labels.append(b[Constants.edge_key].edge_label)
    sync()
    dt = time.perf_counter() - t0
    stage["ranker_train_s"] = round(dt, 2)
    pos = int(sum(int(l.sum()) for l in labels))
    out["ranker_ms_per_iteration"] = round(1e3 * dt / max(ranker_iters, 1), 3)
    out["ranker_positive_edges_per_s"] = round(pos / dt)
    out["ranker_loss_last"] = round(float(rl.detach()), 4)
    out["ranker_iteration_path"] = dict(path, **({"declined": declined} if declined else {}))

    # 5. evaluation of the held-out users: device-built samples with the matchers' candidates, inference, top-12
    t0 = time.perf_counter()
    ev = DeviceGraphSampler(cfg, hetero, users_adj, articles_adj, device=dev, seed=4, train=False, matchers=matchers, shuffle=False)
    stage["eval_sampler_setup_s"] = round(time.perf_counter() - t0, 2)
    eval_u, eval_i = held[0], held[1]

    sync()
    t0 = time.perf_counter()
    ev.step = 0
    customers, preds = RS.make_predictions(ranker, ev.iter_users(eval_u), k=12, device=dev)   # sampling pipelined, no host wait per batch
    sync()
    dt = time.perf_counter() - t0
    stage["eval_inference_s"] = round(dt, 2)
    out["inference_users_per_s"] = round(customers.numel() / dt)

    # 6. MAP@12 on the held-out purchases (AP of a single held-out item = 1 / rank when it is among the 12)
    assert t.equal(customers, eval_u), "every evaluation user once, in order"
    rank = t.arange(1, 13, dtype=t.float32)
    truth = eval_i[:, None]
    ap_rank = ((preds == truth).float() / rank).sum(1)
    top_cpu = top[eval_u.to(top.device)].cpu()
    ap_cand = ((top_cpu[:, :12] == truth).float() / rank).sum(1)
    pop = t.from_numpy(np.asarray(matchers[1].get_matches(0))).view(1, -1)
    in_cand = (top_cpu == truth).any(1) | (pop == truth).any(1)
    out["eval_users"] = int(eval_u.numel())
    out["map_at_12"] = {"ranker_reranked_candidates": round(float(ap_rank.mean()), 4),
                        "candidate_generator_alone": round(float(ap_cand.mean()), 4),
                        "candidate_recall": round(float(in_cand.float().mean()), 4),
                        "note": "one held-out purchase per user, absent from the graph; the ranker can only rank what the matchers "
                                "proposed, so its MAP is bounded by candidate_recall"}
    out["stage_s"] = stage
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=1_371_980)
    ap.add_argument("--items", type=int, default=105_542)
    ap.add_argument("--edges", type=int, default=31_800_000)
    ap.add_argument("--lightgcn-steps", type=int, default=300)
    ap.add_argument("--ranker-iters", type=int, default=300)
    ap.add_argument("--ranker-batch", type=int, default=128)
    ap.add_argument("--eval-users", type=int, default=20_000)
    ap.add_argument("--top-n", type=int, default=100)
    args = ap.parse_args()
    print(json.dumps(run(users=args.users, items=args.items, edges=args.edges, lightgcn_steps=args.lightgcn_steps,
                         ranker_iters=args.ranker_iters, ranker_batch=args.ranker_batch, eval_users=args.eval_users,
                         top_n=args.top_n)))


if __name__ == "__main__":
    main()

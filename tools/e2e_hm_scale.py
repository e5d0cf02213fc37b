#!/usr/bin/env python3
"""End to end at H&M scale on one MI355X (BASELINE.json configs[2]/[3]): synthetic
1 371 980 customers x 105 542 articles x 31.8 M transactions.

  1. LightGCN candidate generation: fused train steps (on-device sampling, BPR, Adam in the backward epilogue)
  2. top-N dump for every customer, purchases excluded (fused score + filter top-K)
  3. matchers: LightGCN top-N + popular items
  4. ranker training on device-sampled 2-hop subgraphs (sampling of batch i+1 overlapped with step i)
  5. evaluation samples built on device (matcher candidates), ranker inference, top-12, MAP@12 / recall

Not the contract bench (bench.py is): a scale rehearsal of the whole flow with per-stage timings.  Prints one JSON line.
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=1_371_980)
    ap.add_argument("--items", type=int, default=105_542)
    ap.add_argument("--edges", type=int, default=31_800_000)
    ap.add_argument("--lightgcn-steps", type=int, default=100)
    ap.add_argument("--ranker-iters", type=int, default=150)
    ap.add_argument("--ranker-batch", type=int, default=128)
    ap.add_argument("--eval-users", type=int, default=20_000)
    ap.add_argument("--top-n", type=int, default=100)
    args = ap.parse_args()
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
    from laplace_amd.utils.metrics import MAPatK

    dev = "cuda"
    out = {"workload": f"H&M-shaped synthetic {args.users}x{args.items}, {args.edges} transactions, one MI355X"}
    sync = t.cuda.synchronize

    t0 = time.perf_counter()
    spec = S.SyntheticSpec(args.users, args.items, args.edges, seed=2, zipf_s=1.0)
    graph, users_adj, articles_adj = S.generate_hetero(spec)
    ei = graph[Constants.edge_key].edge_index
    out["generate_s"] = round(time.perf_counter() - t0, 1)
    U, I = args.users, args.items

    # 1. LightGCN
    t.manual_seed(0)
    t0 = time.perf_counter()
    lgcn = LightGCN(U, I, 64, 3).to(dev)
    inter = Interactions(ei.to(dev), U, I)
    trainer = LightGCNTrainer(lgcn, inter.adjacency("bipartite"), inter, lr=5e-3, Lambda=1e-6, batch_size=16384, seed=1)
    sync()
    out["lightgcn_setup_s"] = round(time.perf_counter() - t0, 2)
    first_loss = float(trainer.step())
    for _ in range(4):
        trainer.step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.lightgcn_steps):
        last_loss = trainer.step()  # a persistent device scalar: read once, after the timed loop
    sync()
    dt = time.perf_counter() - t0
    out["lightgcn_ms_per_step"] = round(1e3 * dt / args.lightgcn_steps, 3)
    out["lightgcn_positive_edges_per_s"] = round(16384 * args.lightgcn_steps / dt)
    out["lightgcn_loss_first_last"] = [round(first_loss, 4), round(float(last_loss), 4)]  # unbounded below as written

    # 2. top-N dump, purchases excluded (the trainer relabelled the nodes for locality: rows back under their ids first)
    trainer.finish()
    t0 = time.perf_counter()
    top = save_predictions(lgcn, ei.to(dev), num_recommendations=args.top_n)
    sync()
    dt = time.perf_counter() - t0
    out["topn_dump_s"] = round(dt, 2)
    out["topn_users_per_s"] = round(U / dt)
    del trainer, inter
    t.cuda.empty_cache()

    # 3. matchers
    t0 = time.perf_counter()
    matchers = [LightGCNMatcher(top, args.top_n), PopularItemsMatcher.from_adjacency(articles_adj, 50)]  # answered on the device (N3)
    out["matchers_s"] = round(time.perf_counter() - t0, 2)

    # 4. ranker on device-sampled batches
    cfg = SimpleNamespace(k=12, num_neighbors=64, n_hop_neighbors=2, positive_edges_ratio=0.5, negative_edges_ratio=3.0,
                          batch_size=args.ranker_batch, num_gnn_layers=2, hidden_layer_size=128, encoder_layer_output_size=64,
                          conv_agg_type="add", num_linear_layers=2, heterogeneous_prop_agg_type="sum", batch_norm=True,
                          p_dropout_edges=0.0, p_dropout_features=0.3)
    sampler = DeviceGraphSampler(cfg, graph, users_adj, articles_adj, device=dev, seed=3)
    it = iter(sampler)
    first = next(it)
    ranker = Encoder_Decoder_Model(get_SAGEConv_layers(2, 128, 64, "add"), get_linear_layers(2, 128, 128, 1),
                                   get_feature_info(graph), first.metadata(), True, "sum", True, 0.0, 0.3).to(dev)
    ranker.initialize_encoder_input_size(first)
    opt = t.optim.Adam(ranker.parameters(), lr=0.01)     # the reference's optimizer (run_pipeline.py)
    ranker.train()
    from laplace_amd.ranker_native import NativeRankerStep
    from laplace_amd.ranker_step import FusedRankerStep
    native = NativeRankerStep(ranker, opt)   # what training.train_with_dataloader runs per batch: one C call per iteration
    fused = FusedRankerStep(ranker, opt)

    def step(batch):
        lab = batch[Constants.edge_key]
        loss = native.step(batch.x_dict, batch.edge_index_dict, lab.edge_label_index, lab.edge_label)
        return loss if loss is not None else fused.step(*select_properties(batch))

    for _ in range(5):
        step(next(it))
    sync()
    labels, t0 = [], time.perf_counter()
    for _ in range(args.ranker_iters):
        b = next(it)
        rl = step(b)
        labels.append(b[Constants.edge_key].edge_label)
    sync()
    dt = time.perf_counter() - t0
    pos = int(sum(int(l.sum()) for l in labels))
    out["ranker_ms_per_iteration"] = round(1e3 * dt / args.ranker_iters, 3)
    out["ranker_positive_edges_per_s"] = round(pos / dt)
    out["ranker_loss_last"] = round(float(rl.detach()), 4)

    # 5. evaluation: device-built samples with the matchers' candidates, inference, top-12
    t0 = time.perf_counter()
    ev = DeviceGraphSampler(cfg, graph, users_adj, articles_adj, device=dev, seed=4, train=False, matchers=matchers,
                            shuffle=False)
    out["eval_sampler_setup_s"] = round(time.perf_counter() - t0, 2)

    def first_batches():
        n = 0
        for b in ev:
            yield b
            n += cfg.batch_size
            if n >= args.eval_users:
                return

    sync()
    t0 = time.perf_counter()
    customers, preds = RS.make_predictions(ranker, first_batches(), k=12, device=dev)
    sync()
    dt = time.perf_counter() - t0
    out["inference_users_per_s"] = round(customers.numel() / dt)
    gt = [t.from_numpy(np.asarray(users_adj[int(u)])) for u in customers.tolist()]
    out["eval_users"] = int(customers.numel())
    # purchases no matcher proposed are label-0 candidates in the reference's evaluation samples, so this MAP is a
    # plumbing figure (can the ranker find the planted purchases among ~150 proposals), not a held-out metric
    out["map_at_12_vs_own_purchases"] = round(MAPatK(gt, preds, k=12), 4)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

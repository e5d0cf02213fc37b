#!/usr/bin/env python3
"""PinSAGE path (BASELINE configs[4]) at H&M scale on one MI355X: item-item batches from on-device
random walks (csrc/pinsage.hip) + the weighted SAGE model, training loop timing.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=1_371_980)
    ap.add_argument("--items", type=int, default=105_542)
    ap.add_argument("--edges", type=int, default=31_800_000)
    ap.add_argument("--batch", type=int, default=32)          # pinsage/model.py:143-148 defaults
    ap.add_argument("--walk-length", type=int, default=2)
    ap.add_argument("--restart", type=float, default=0.5)
    ap.add_argument("--walks", type=int, default=10)
    ap.add_argument("--neighbors", type=int, default=3)
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--hidden", type=int, default=64)
    ap.add_argument("--iters", type=int, default=200)
    args = ap.parse_args()
    import numpy as np
    import torch as t
    from laplace_amd import synthetic as S
    from laplace_amd.data.dataset import AdjList
    from laplace_amd.pinsage.model import PinSAGEModel
    from laplace_amd.pinsage.sampler import PinSAGESampler

    ei = S.generate(S.SyntheticSpec(args.users, args.items, args.edges, seed=2, zipf_s=1.0))
    u, a = ei[0].numpy(), ei[1].numpy()
    users, items = AdjList.from_edges(u, a, args.users), AdjList.from_edges(a, u, args.items)
    smp = PinSAGESampler(users, items, args.users, args.items, batch_size=args.batch, random_walk_length=args.walk_length,
                         random_walk_restart_prob=args.restart, num_random_walks=args.walks, num_neighbors=args.neighbors,
                         num_layers=args.layers, seed=1)
    t.manual_seed(0)
    model = PinSAGEModel(args.items, args.hidden, args.layers).to("cuda")
    opt = t.optim.Adam(model.parameters(), lr=3e-3)
    model.train()

    def one():
        b = smp.sample_batch()
        loss = model(b["seeds"], b["pos"], b["neg"], b["blocks"]).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss, b

    for _ in range(10):
        one()
    t.cuda.synchronize()
    t0 = time.perf_counter()
    pairs = 0
    for _ in range(args.iters):
        loss, b = one()
        pairs += int(b["pos"][0].numel())
    t.cuda.synchronize()
    dt = time.perf_counter() - t0
    # sampler alone
    t0 = time.perf_counter()
    for _ in range(50):
        smp.sample_batch()
    t.cuda.synchronize()
    ds = (time.perf_counter() - t0) / 50
    print(json.dumps({"workload": f"PinSAGE item-item training, H&M-shaped synthetic {args.users}x{args.items}, {args.edges} edges; "
                                  f"batch {args.batch} pairs, walks {args.walks} x length {args.walk_length}, restart {args.restart}, "
                                  f"T={args.neighbors}, {args.layers} layers, hidden {args.hidden}",
                      "ms_per_iteration": round(1e3 * dt / args.iters, 3), "positive_pairs_per_s": round(pairs / dt),
                      "sampler_ms_per_batch": round(1e3 * ds, 3), "loss": round(float(loss.detach()), 4)}))


if __name__ == "__main__":
    main()

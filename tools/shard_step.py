#!/usr/bin/env python3
"""Per-rank compute of BASELINE configs[3] at world N, measured on ONE GPU: rank `--rank` of `--world` builds its user shard
(bench.py's partition: S.shard_blocks / generate_blocks, global batch 131 072 / N) and runs dist.ShardedLightGCNTrainer's
step with no peers (torch.distributed not initialised: world = 1 inside the trainer, every collective skipped; the item
degrees are then the shard's own, which changes edge weights, not work).  What the N-GPU job cannot beat per step —
the exchanges come on top — and where the step stops shrinking with the shard."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
from laplace_amd import ops, synthetic as S
from laplace_amd.dist import ShardedLightGCNTrainer
from laplace_amd.interactions import Interactions
from laplace_amd.model.lightgcn import LightGCN

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--rank", type=int, default=0)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--events", type=int, default=1)
args = ap.parse_args()
spec = S.C4
b0, b1 = S.shard_blocks(S.C4_BLOCKS, args.world, args.rank)
ei = S.generate_blocks(spec, S.C4_BLOCKS, b0, b1)
U, I, B = spec.num_users // args.world, spec.num_items, 131072 // args.world
t.manual_seed(1)
model = LightGCN(U, I, embedding_dim=128, num_iterations=3).to("cuda")
inter = Interactions(ei.to("cuda"), U, I)
tr = ShardedLightGCNTrainer(model, inter, lr=1e-3, Lambda=1e-6, batch_size=B, seed=7)
for _ in range(args.warmup):
    tr.step()
t.cuda.synchronize()
if args.events:
    ops.SPMM_EVENTS = []
t0 = time.perf_counter()
for _ in range(args.steps):
    tr.step()
t.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
out = {"world": args.world, "rank": args.rank, "users": U, "edges": int(ei.shape[1]), "batch": B, "ms_per_step": round(1e3 * dt, 3),
       "ideal_ms": None}
if args.events:
    by = {}
    for e0, e1, kind, _, n_rows, a in ops.SPMM_EVENTS:
        by.setdefault((kind, n_rows), []).append(e0.elapsed_time(e1))
    out["launches_ms"] = {f"{k[0]}:{k[1]}rows": [round(sum(v) / len(v), 3), len(v) // args.steps] for k, v in sorted(by.items())}
    out["launch_sum_ms"] = round(sum(sum(v) for v in by.values()) / args.steps, 3)
    ops.SPMM_EVENTS = None
print(json.dumps(out))

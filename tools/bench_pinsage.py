#!/usr/bin/env python3
"""PinSAGE path (BASELINE configs[4]) at H&M scale: item-item batches from on-device random walks
(csrc/pinsage.hip) + the weighted SAGE model, training loop timing.  Prints one JSON line.
--gpus N (BASELINE configs[4] names 4): data parallel — the graph is replicated, every rank draws its own batches
(its own Philox stream), the gradients (item-id embedding table + the dense layers) are averaged with one flat
all-reduce per iteration (dist_ranker.allreduce_gradients); the parent starts the N workers itself."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse_args(argv=None):
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
    ap.add_argument("--hidden", type=int, default=16)          # pinsage/model.py:148
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--serial", action="store_true", help="sample each batch on the training stream (no overlap)")
    ap.add_argument("--autograd", action="store_true", help="the op-by-op autograd iteration instead of the native executor")
    return ap.parse_args(argv)


_GRAPH_CACHE = {}   # (users, items, edges) -> (AdjList users, AdjList items): bench.py's block runs two sampler settings on one graph


def bench_line(**kw) -> dict:
    """The single-GPU measurement as a dict (bench.py's pinsage_c5 block): bench_line(iters=300)."""
    args = parse_args([])
    for k, v in kw.items():
        setattr(args, k, v)
    return run(args)


def main():
    args = parse_args()
    out = run(args)
    if out is not None:
        print(json.dumps(out))


def run(args):
    from laplace_amd import launch
    if args.gpus > 1 and not launch.launched():
        sys.exit(launch.self_launch(__file__, sys.argv[1:], args.gpus, result_marker='"workload"'))
    import numpy as np
    import torch as t
    import torch.distributed as dist
    rank, world, dev = launch.init_distributed()
    from laplace_amd.dist_ranker import allreduce_gradients, broadcast_parameters
    from laplace_amd import synthetic as S
    from laplace_amd.data.dataset import AdjList
    from laplace_amd.pinsage.model import PinSAGEModel
    from laplace_amd.pinsage.sampler import PinSAGESampler

    key = (args.users, args.items, args.edges)
    if key not in _GRAPH_CACHE:
        ei = S.generate(S.SyntheticSpec(args.users, args.items, args.edges, seed=2, zipf_s=1.0))
        u, a = ei[0].numpy(), ei[1].numpy()
        _GRAPH_CACHE.clear()
        _GRAPH_CACHE[key] = (AdjList.from_edges(u, a, args.users), AdjList.from_edges(a, u, args.items))
    users, items = _GRAPH_CACHE[key]
    smp = PinSAGESampler(users, items, args.users, args.items, batch_size=args.batch, random_walk_length=args.walk_length,
                         random_walk_restart_prob=args.restart, num_random_walks=args.walks, num_neighbors=args.neighbors,
                         num_layers=args.layers, seed=1 + 7919 * rank)
    t.manual_seed(0)
    model = PinSAGEModel(args.items, args.hidden, args.layers).to(dev)
    broadcast_parameters(model)
    opt = t.optim.Adam(model.parameters(), lr=3e-5, fused=True)   # pinsage/model.py:153 (fused: one multi-tensor launch, same update)
    model.train()
    t.autograd.set_multithreading_enabled(False)                  # as pinsage.model.train_epoch does

    from laplace_amd.pinsage.native import NativePinSAGEStep
    native = None
    if not args.autograd and NativePinSAGEStep.supports(model, opt):
        if world == 1:
            native = NativePinSAGEStep(model, opt)       # one C call per iteration (mi_pinsage_step_f32)
        else:   # data-parallel: compact gradient rows all-gathered, dense layers all-reduced, mi_pinsage_apply_f32
            native = NativePinSAGEStep(model, opt, data_parallel=True, seed=1234 + rank)
            native.exchange_capacity = (3 * args.batch * (1 + args.neighbors) ** args.layers, 3 * args.batch)

    def one(b=None):
        b = smp.sample_batch() if b is None else b
        if native is not None:
            loss = native.step(b)
            if loss is not None:
                return loss, b
            # (data-parallel: the decline is collective, every rank takes the autograd iteration below together)
        loss = model(b["seeds"], b["pos"], b["neg"], b["blocks"]).mean()
        opt.zero_grad()
        loss.backward()
        allreduce_gradients(model.parameters())
        opt.step()
        return loss, b

    for _ in range(10):
        one()
    t.cuda.synchronize()
    t0 = time.perf_counter()
    pairs = 0
    serial = bool(args.serial)        # sample_batch() per iteration instead of the overlapped iterator
    for b in ((None for _ in range(args.iters)) if serial else smp.batches(args.iters)):
        loss, b = one(b)
        pairs += int(b["pos"][0].numel())
    t.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tot = t.tensor([float(pairs), dt], device=dev, dtype=t.float64)
        dist.all_reduce(tot[:1])
        dist.all_reduce(tot[1:], op=dist.ReduceOp.MAX)
        pairs, dt = float(tot[0]), float(tot[1])
    # sampler alone
    t0 = time.perf_counter()
    for _ in range(50):
        smp.sample_batch()
    t.cuda.synchronize()
    ds = (time.perf_counter() - t0) / 50
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return None
    out = ({"n_gpus": world, "backend": os.environ.get("LAPLACE_BENCH_BACKEND", "nccl") if world > 1 else None,
                      "workload": f"PinSAGE item-item training, H&M-shaped synthetic {args.users}x{args.items}, {args.edges} edges; "
                                  f"batch {args.batch} pairs, walks {args.walks} x length {args.walk_length}, restart {args.restart}, "
                                  f"T={args.neighbors}, {args.layers} layers, hidden {args.hidden}",
                      "iteration": ("native executor (mi_pinsage_step_f32)" + (" + compact-row exchange (mi_pinsage_apply_f32)" if world > 1 else ""))
                      if native is not None else "autograd, op by op",
                      "sampling": "serial" if serial else "overlapped (two side streams, two batches ahead)",
                      "ms_per_iteration": round(1e3 * dt / args.iters, 3), "positive_pairs_per_s": round(pairs / dt),
                      "sampler_ms_per_batch": round(1e3 * ds, 3), "loss": round(float(loss.detach()), 4)})
    if world > 1:
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()

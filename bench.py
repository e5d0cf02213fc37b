#!/usr/bin/env python3
"""bench.py — positive-edges/sec of the LightGCN train step on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch: K-layer propagate forward, on-device
sampling of B positive edges + negatives, fused BPR forward/backward, K-layer propagate backward,
dense Adam over the whole table (run_pipeline_lightgcn.py:117-159).  Workload at N=1 is
BASELINE.json configs[1]: synthetic bipartite 1M users x 100K items, 10M edges, LightGCN 3-layer
D=128 (SURVEY §8d "C2": symmetric adjacency nnz=20M, B=16384).  At N>1 every rank owns its own
1M-user / 10M-edge shard of a weak-scaled graph (items replicated, one RCCL all-reduce of the
item rows per layer), so per-GPU work is fixed: "scaling": "weak".

Launch: python bench.py --gpus 1 --steps K --warmup W
        python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy rate


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--users", type=int, default=1_000_000)
    ap.add_argument("--items", type=int, default=100_000)
    ap.add_argument("--edges", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--layers", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16384)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--plain-step", action="store_true",
                    help="A/B: the straightforward step (full final, dense gradient buffer) instead of the byte-saving one")
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--no-cpu-faithful", action="store_true", help="skip timing the reference's per-iteration sampler on the host")
    ap.add_argument("--uniform", action="store_true",
                    help="control graph of SURVEY 8d: i.i.d. uniform endpoints instead of log-normal users x Zipf items")
    return ap.parse_args()


def spmm_bytes(nnz: int, n_rows: int, d: int) -> int:
    """Algorithmic bytes of one propagate launch: int32 col + fp32 val per entry, the
    row pointer, one gathered fp32 row per entry (no reuse credit), one stored row per output row."""
    return nnz * 8 + (n_rows + 1) * 4 + nnz * d * 4 + n_rows * d * 4


def cpu_baseline(ei, spec, args, table0):
    """The oracle's CPU port of the same step, timed on this box's host cores (rank 0, N=1 only)."""
    import torch as t
    from oracle import lightgcn_ref as R
    U, I = spec.num_users, spec.num_items
    cores = R.clib().ref_num_threads()
    t.set_num_threads(cores)
    r, c = R.bipartite_edges(ei[0], ei[1], U)
    rowptr, cs, _ = R.sparse_tensor_csr(r, c, U + I, U + I)
    val = R.gcn_norm_csr(rowptr, cs)
    port = R.CpuTrainPort(table0, rowptr, cs, val, U, args.layers, 1e-3, 1e-6)
    g = t.Generator().manual_seed(1)
    E = ei.shape[1]

    def batch():
        e = t.randint(0, E, (args.batch,), generator=g)
        return ei[0][e], ei[1][e], t.randint(0, I, (args.batch,), generator=g)  # negatives pre-drawn

    port.step(batch())  # warm-up (page-faults the buffers)
    t0 = time.perf_counter()
    for _ in range(args.cpu_steps):
        port.step(batch())
    dt = (time.perf_counter() - t0) / args.cpu_steps
    faithful = None
    if not args.no_cpu_faithful:
        # informational: the reference draws a negative for EVERY train edge every iteration
        # (data/lightgcn_loader.py:95-112, np.isin rejection) before picking the batch; one draw is timed
        import random
        import numpy as np
        t1 = time.perf_counter()
        R.sample_mini_batch(args.batch, ei, np.random.default_rng(0), random.Random(0))
        ts = time.perf_counter() - t1
        faithful = {"sampler_s_per_step": round(ts, 2), "value": args.batch / (dt + ts), "unit": "positive-edges/s",
                    "note": "model-only step + the reference's per-iteration O(E) negative sampling; not the denominator of any claim"}
    return {"value": args.batch / dt, "unit": "positive-edges/s", "cores": int(cores), "kind": "port", "faithful_step": faithful,
            "sample": f"{args.cpu_steps} full model-only train steps (fwd+BPR+bwd+Adam, negatives pre-drawn) of the "
                      f"same graph and batch size with oracle/ (C/OpenMP restatement of torch_sparse spmm_cpu + "
                      f"torch CPU ops), {dt:.2f} s/step"}


def main():
    args = parse_args()
    import torch as t
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N>1")
    if not t.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    # LAPLACE_BENCH_BACKEND=gloo + LAPLACE_BENCH_ONE_GPU=1 rehearse the N>1 path on a 1-GPU box
    # (ranks share the card, collectives staged through the host); never used for a reported number.
    backend = os.environ.get("LAPLACE_BENCH_BACKEND", "nccl")
    if os.environ.get("LAPLACE_BENCH_ONE_GPU") == "1":
        local_rank = 0
    t.cuda.set_device(local_rank)
    dev = t.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from laplace_amd import ops, synthetic as S
    from laplace_amd.interactions import Interactions
    from laplace_amd.model.lightgcn import LightGCN
    from laplace_amd.trainer import LightGCNTrainer

    if os.environ.get("LAPLACE_SPMM_TWO_STREAMS") is not None:  # A/B switch
        ops.SPMM_TWO_STREAMS = os.environ["LAPLACE_SPMM_TWO_STREAMS"] == "1"
    spec = S.SyntheticSpec(args.users, args.items, args.edges, seed=1, uniform=args.uniform)
    if world > 1:
        spec = S.shard_spec(spec, rank)
    t_gen = time.perf_counter()
    ei = S.generate(spec)
    t_gen = time.perf_counter() - t_gen
    U, I, D, K, B = spec.num_users, spec.num_items, args.dim, args.layers, args.batch

    t.manual_seed(1234 + rank)
    model = LightGCN(U, I, embedding_dim=D, num_iterations=K)
    table0 = model.table().clone() if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None
    model.to(dev)
    inter = Interactions(ei.to(dev), U, I)
    adj = inter.adjacency("bipartite")
    if world == 1:
        trainer = LightGCNTrainer(model, adj, inter, lr=1e-3, Lambda=1e-6, batch_size=B, seed=7,
                                  sparse_batch=not args.plain_step)
    else:
        from laplace_amd.dist import ShardedLightGCNTrainer
        trainer = ShardedLightGCNTrainer(model, inter, lr=1e-3, Lambda=1e-6, batch_size=B, seed=7 + rank,
                                         sparse_batch=not args.plain_step)
    nnz, n_rows = trainer.adj_fwd.nnz, trainer.adj_fwd.n_rows
    t.cuda.synchronize()

    def sync():
        if world > 1:
            dist.barrier()
        t.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.step()
    sync()
    ops.SPMM_EVENTS = []  # (start, end) HIP events around every propagate launch, on the launch stream
    marks = [t.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]  # per-step spread; no sync inside the loop
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        loss = trainer.step()
        marks[i + 1].record()
    sync()
    elapsed = time.perf_counter() - t0
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    pct = lambda q: per_step[min(len(per_step) - 1, int(q * len(per_step)))]
    events, ops.SPMM_EVENTS = ops.SPMM_EVENTS, None
    loss_val = float(loss)

    if world > 1:
        tt = t.tensor([elapsed], device=dev, dtype=t.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt)

    if rank == 0:
        # The roofline is quoted on the DENSE propagate (every entry of the adjacency slice gathered): kernels
        # spmm_*_kernel<..., false>.  The sparse-operand launches of the byte-saving step (last forward layer at
        # the batch rows, first backward layer over the non-zero gradient rows) gather a data-dependent subset and
        # are reported beside it, not mixed in.  Algorithmic bytes are summed per launch from that launch's own
        # entry and row counts (a sharded layer is two launches: item rows and user rows).
        dense = [(s.elapsed_time(e), a) for s, e, kind, _, _, a in events if kind == "dense"]
        sparse_ms = [s.elapsed_time(e) for s, e, kind, _, _, _ in events if kind == "sparse"]
        # the last backward product carries the Adam update in its epilogue (+6 table streams): timed apart
        fused_ms = [s.elapsed_time(e) for s, e, kind, _, _, _ in events if kind == "dense_adam"]
        nnz_of = {}
        for _, a in dense:
            if id(a) not in nnz_of:  # entries of a row slice = rowptr[last] - rowptr[first]
                nnz_of[id(a)] = int(a.rowptr[-1]) - int(a.rowptr[0])
        spmm_ms = [ms for ms, _ in dense]
        algo_total = sum(spmm_bytes(nnz_of[id(a)], a.n_rows, D) for _, a in dense)
        n_layers_timed = len(dense)
        avg_ms = sum(spmm_ms) / max(len(spmm_ms), 1)
        algo = algo_total / max(len(dense), 1)
        achieved = algo_total / (sum(spmm_ms) * 1e-3) / 1e9 if spmm_ms else 0.0
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        default_workload = (args.users, args.items, args.edges, D) == (1_000_000, 100_000, 10_000_000, 128) and not args.uniform
        if os.path.exists(tf) and default_workload:  # PMC bytes were collected on exactly this workload
            try:
                traffic = json.load(open(tf)).get("spmm_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "positive-edges/sec (train step)",
            "value": world * B * args.steps / elapsed,
            "unit": "positive-edges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "ms_per_step_p10_p50_p90": [round(pct(0.1), 4), round(pct(0.5), 4), round(pct(0.9), 4)],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"LightGCN train step, synthetic bipartite {U}x{I} users x items per GPU, "
                                   f"{args.edges} edges per GPU{' (uniform endpoints)' if args.uniform else ''} (symmetric adjacency nnz={nnz}), "
                                   f"{K}-layer D={D}, batch {B} positive edges per GPU, on-device sampling, "
                                   f"BPR + dense Adam; BASELINE.json configs[1]",
                       "parallelism": "1 GPU" if world == 1 else f"user-sharded x{world}, items replicated, "
                                                                 f"RCCL all-reduce of item rows per layer"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "mi_spmm_csr_f32 dense launch (spmm_items_kernel + spmm_rows_kernel + spmm_fixup_kernel, SPARSE=false)",
                         "algorithmic_bytes_per_launch": algo, "avg_launch_ms": avg_ms,
                         # every operand read once / written once (SURVEY 8d "compulsory lower bound")
                         "compulsory_bytes_per_launch": nnz * 8 + (n_rows + 1) * 4 + 2 * n_rows * D * 4,
                         "launches_timed": len(spmm_ms), "layers_timed": n_layers_timed,
                         "dense_launches_per_step": len(dense) / args.steps,
                         "sparse_launches_per_step": len(sparse_ms) / args.steps,
                         "sparse_launch_avg_ms": (sum(sparse_ms) / len(sparse_ms)) if sparse_ms else None,
                         "dense_launches_with_adam_epilogue_per_step": len(fused_ms) / args.steps,
                         "dense_with_adam_epilogue_avg_ms": (sum(fused_ms) / len(fused_ms)) if fused_ms else None},
            "step_form": "plain" if args.plain_step else "sparse_batch",
            "loss": loss_val, "graph_gen_s": round(t_gen, 1), "backend": backend if world > 1 else None,
        }
        if table0 is not None:
            out["cpu_baseline"] = cpu_baseline(ei, spec, args, table0)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

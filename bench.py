#!/usr/bin/env python3
"""bench.py — positive-edges/sec of the LightGCN train step on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch: K-layer propagate forward, on-device
sampling of B positive edges + negatives, fused BPR forward/backward, K-layer propagate backward,
dense Adam over the whole table (run_pipeline_lightgcn.py:117-159).

  --config c2 (default)  BASELINE.json configs[1]: synthetic bipartite 1M users x 100K items, 10M edges, LightGCN
                         3-layer D=128 (SURVEY 8d "C2": symmetric adjacency nnz=20M, B=16384).  At N>1 every rank
                         owns its own 1M-user / 10M-edge shard of a weak-scaled graph: "scaling": "weak".
  --config c4            BASELINE.json configs[3]: ONE fixed graph, 8M users x 100K items, 100M edges (seed 3), sharded
                         by user_id // ceil(U/N) over the N ranks (items replicated, one RCCL all-reduce of the item
                         rows per layer), global batch 131072: total work fixed, "scaling": "strong".

Launch: python bench.py --gpus N --steps K --warmup W      (N>1: the parent starts N worker processes itself, before
        touching the GPU) or, equivalently, python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
Prints ONE JSON line (rank 0).
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured achievable rate
C4_GLOBAL_BATCH = 131072


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=("c2", "c4"), default="c2")
    ap.add_argument("--users", type=int, default=None)
    ap.add_argument("--items", type=int, default=None)
    ap.add_argument("--edges", type=int, default=None)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--layers", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="positive edges per GPU (c2) / per job (c4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", action="store_true", help="time the CPU port on --config c4 too (minutes of host work)")
    ap.add_argument("--plain-step", action="store_true",
                    help="A/B: time the straightforward step (full final, dense gradient buffer) as the headline instead of the byte-saving one")
    ap.add_argument("--no-plain-leg", action="store_true", help="skip the extra plain-step timing (plain_step_ms)")
    ap.add_argument("--no-map", action="store_true", help="skip the MAP@12 leg")
    ap.add_argument("--map-steps", type=int, default=1000, help="extra train steps before MAP@12 is scored")
    ap.add_argument("--map-users", type=int, default=20000)
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--no-cpu-faithful", action="store_true", help="skip timing the reference's per-iteration sampler on the host")
    ap.add_argument("--no-reorder", action="store_true", help="A/B: train under the generator's ids instead of the locality order")
    ap.add_argument("--uniform", action="store_true",
                    help="control graph of SURVEY 8d: i.i.d. uniform endpoints instead of log-normal users x Zipf items")
    return ap.parse_args()


def spmm_bytes(nnz: int, n_rows: int, d: int) -> int:
    """Algorithmic bytes of one propagate launch: int32 col + fp32 val per entry, the
    row pointer, one gathered fp32 row per entry (no reuse credit), one stored row per output row."""
    return nnz * 8 + (n_rows + 1) * 4 + nnz * d * 4 + n_rows * d * 4


def kernel_source_hash() -> str:
    """Identifies the propagate kernels a PMC traffic figure was collected on (profiles/traffic.json)."""
    h = hashlib.sha256()
    for f in ("spmm.hip", "common.hpp"):
        with open(os.path.join(ROOT, "laplace-gnn-recommendation_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def cpu_baseline(ei, U, I, args, B, table0):
    """The oracle's CPU port of the same step, timed on this box's host cores (rank 0, N=1 only)."""
    import torch as t
    from oracle import lightgcn_ref as R
    cores = R.clib().ref_num_threads()
    t.set_num_threads(cores)
    r, c = R.bipartite_edges(ei[0], ei[1], U)
    rowptr, cs, _ = R.sparse_tensor_csr(r, c, U + I, U + I)
    val = R.gcn_norm_csr(rowptr, cs)
    port = R.CpuTrainPort(table0, rowptr, cs, val, U, args.layers, 1e-3, 1e-6)
    g = t.Generator().manual_seed(1)
    E = ei.shape[1]

    def batch():
        e = t.randint(0, E, (B,), generator=g)
        return ei[0][e], ei[1][e], t.randint(0, I, (B,), generator=g)  # negatives pre-drawn

    port.step(batch())  # warm-up (page-faults the buffers)
    t0 = time.perf_counter()
    for _ in range(args.cpu_steps):
        port.step(batch())
    dt = (time.perf_counter() - t0) / args.cpu_steps
    faithful = None
    if not args.no_cpu_faithful and args.config == "c2":
        # informational: the reference draws a negative for EVERY train edge every iteration
        # (data/lightgcn_loader.py:95-112, np.isin rejection) before picking the batch; one draw is timed
        import random
        import numpy as np
        t1 = time.perf_counter()
        R.sample_mini_batch(B, ei, np.random.default_rng(0), random.Random(0))
        ts = time.perf_counter() - t1
        faithful = {"sampler_s_per_step": round(ts, 2), "value": B / (dt + ts), "unit": "positive-edges/s",
                    "note": "model-only step + the reference's per-iteration O(E) negative sampling; not the denominator of any claim"}
    return {"value": B / dt, "unit": "positive-edges/s", "cores": int(cores), "kind": "port", "faithful_step": faithful,
            "sample": f"{args.cpu_steps} full model-only train steps (fwd+BPR+bwd+Adam, negatives pre-drawn) of the "
                      f"same graph and batch size with oracle/ (C/OpenMP restatement of torch_sparse spmm_cpu + "
                      f"torch CPU ops), {dt:.2f} s/step"}


def map_at_12(model, trainer, inter, held, n_steps: int) -> dict:
    """Ranking quality beside the throughput (BASELINE.json's metric names MAP@12): train n_steps more steps, then
    score the reference's predictor (layer-0 embeddings, a user's train items never recommended:
    utils/metrics_lightgcn.py:125-142) on one held-out positive per sampled user.  AP of a single held-out item =
    1/rank when it is among the 12, else 0 (the Kaggle H&M definition, utils.metrics.MAPatK)."""
    import torch as t
    from laplace_amd.utils.metrics_lightgcn import topk_for_users
    for _ in range(n_steps):
        trainer.step()
    users, truth = held[0], held[1]
    rank = t.arange(1, 13, device=users.device, dtype=t.float32)
    # LightGCN's own predictor for comparison: the PROPAGATED embeddings mean_k(A^k E0) (the reference scores with layer 0, F8)
    fin = trainer.forward()
    if getattr(trainer, "order", None) is not None:
        fin = fin[trainer.order.node_new_of_old()]
    U = model.num_users
    top_f = topk_for_users(fin[:U].contiguous(), fin[U:].contiguous(), users, inter.edge_index, 12)
    hit_f = top_f == truth[:, None]
    ap_f = (hit_f.to(t.float32) / rank).sum(dim=1)
    del fin
    trainer.to_original_order()
    top = topk_for_users(model.users_emb.weight.detach(), model.items_emb.weight.detach(), users, inter.edge_index, 12)
    hit = top == truth[:, None]
    ap = (hit.to(t.float32) / rank).sum(dim=1)
    # yardstick on the same users: the popularity predictor (the synthetic graph has no structure beyond popularity)
    I = model.num_items
    deg = t.bincount(inter.edge_index[1], minlength=I)
    pop = t.argsort(deg, descending=True, stable=True)[:256]                       # more than 12 + any realistic overlap
    keys = t.sort(inter.edge_index[0] * I + inter.edge_index[1])[0]
    cand = users[:, None] * I + pop[None, :]
    pos = t.searchsorted(keys, cand.reshape(-1)).clamp(max=keys.numel() - 1).reshape(cand.shape)
    free = keys[pos] != cand                                                      # popular items the user does not own
    place = t.cumsum(free.to(t.int64), dim=1)                                     # 1-based position in the user's list
    hit_pop = free & (pop[None, :] == truth[:, None]) & (place <= 12)
    ap_pop = (hit_pop.to(t.float32) / place.clamp(min=1).to(t.float32)).sum(dim=1)
    return {"value": float(ap.mean()), "propagated_embeddings_map_at_12": float(ap_f.mean()),
            "popularity_predictor_map_at_12": float(ap_pop.mean()), "k": 12, "users": int(users.numel()), "heldout_per_user": 1,
            "train_steps_before_scoring": int(trainer.step_count), "hit_rate_at_12": float(hit.any(dim=1).float().mean()),
            "predictor": "layer-0 embeddings, train items excluded (utils/metrics_lightgcn.py:125-142)"}


def main():
    args = parse_args()
    from laplace_amd import launch
    if args.gpus > 1 and not launch.launched():
        # the parent never touches the GPU; one deadline from launch, every worker watched (laplace_amd/launch.py)
        sys.exit(launch.self_launch(__file__, sys.argv[1:], args.gpus, timeout_s=float(os.environ.get("LAPLACE_BENCH_DEADLINE_S", 900))))
    import torch as t
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not t.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    # LAPLACE_BENCH_BACKEND=gloo + LAPLACE_BENCH_ONE_GPU=1 rehearse the N>1 path on a 1-GPU box
    # (ranks share the card, collectives staged through the host); never used for a reported number.
    backend = launch.backend_name()
    rank, world, dev = launch.init_distributed()   # rendezvous + collectives time out instead of hanging

    from laplace_amd import ops, synthetic as S
    from laplace_amd.interactions import Interactions
    from laplace_amd.model.lightgcn import LightGCN
    from laplace_amd.trainer import LightGCNTrainer

    if os.environ.get("LAPLACE_SPMM_TWO_STREAMS") is not None:  # A/B switch
        ops.SPMM_TWO_STREAMS = os.environ["LAPLACE_SPMM_TWO_STREAMS"] == "1"
    base = S.C4 if args.config == "c4" else S.C2
    spec = S.SyntheticSpec(args.users or base.num_users, args.items or base.num_items, args.edges or base.num_edges,
                           seed=base.seed, uniform=args.uniform)
    strong = args.config == "c4"
    t_gen = time.perf_counter()
    if strong:
        b0, b1 = S.shard_blocks(S.C4_BLOCKS, world, rank)
        ei = S.generate_blocks(spec, S.C4_BLOCKS, b0, b1)
        U_local = spec.num_users // world
        B = (args.batch or C4_GLOBAL_BATCH) // world
    else:
        lspec = S.shard_spec(spec, rank) if world > 1 else spec
        ei = S.generate(lspec)
        U_local = spec.num_users
        B = args.batch or 16384
    t_gen = time.perf_counter() - t_gen
    U, I, D, K = U_local, spec.num_items, args.dim, args.layers
    default_workload = (args.config == "c2" and (spec.num_users, I, spec.num_edges, D, K) == (1_000_000, 100_000, 10_000_000, 128, 3)
                        and not args.uniform)

    t.manual_seed(1234 + rank)
    model = LightGCN(U, I, embedding_dim=D, num_iterations=K)
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline and (args.config == "c2" or args.cpu_baseline)
    table0 = model.table().clone() if want_cpu else None
    model.to(dev)
    inter = Interactions(ei.to(dev), U, I)
    if world == 1:
        adj = inter.adjacency("bipartite")
        trainer = LightGCNTrainer(model, adj, inter, lr=1e-3, Lambda=1e-6, batch_size=B, seed=7,
                                  sparse_batch=not args.plain_step, reorder=False if args.no_reorder else None)
    else:
        from laplace_amd.dist import ShardedLightGCNTrainer
        trainer = ShardedLightGCNTrainer(model, inter, lr=1e-3, Lambda=1e-6, batch_size=B, seed=7 + rank,
                                         sparse_batch=not args.plain_step, reorder=False if args.no_reorder else None)
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
        # are reported beside it, not mixed in.
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
        avg_ms = sum(spmm_ms) / max(len(spmm_ms), 1)
        algo = algo_total / max(len(dense), 1)
        algo_gbs = algo_total / (sum(spmm_ms) * 1e-3) / 1e9 if spmm_ms else 0.0
        compulsory = nnz * 8 + (n_rows + 1) * 4 + 2 * n_rows * D * 4  # every operand read once / written once
        comp_gbs = compulsory / (avg_ms * 1e-3) / 1e9 if spmm_ms else 0.0
        # L2-miss (fabric) bytes per dense launch from the rocprofv3 PMC passes of this same command
        # (tools/prof_bench.sh -> profiles/traffic.json): NOT measured in this run — counters need the profiler —
        # so the file carries the hash of the kernel sources it was taken on and is refused when they changed.
        traffic, traffic_src = None, None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf) and default_workload and world == 1:
            try:
                tj = json.load(open(tf))
                if tj.get("kernel_source_hash") == kernel_source_hash():
                    traffic = tj.get("spmm_hbm_bytes_per_launch")
                    traffic_src = (f"profiles/traffic.json <- {tj.get('source')}: offline rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
                                   f"of this command on the kernels with source hash {tj.get('kernel_source_hash')} (matches this build); "
                                   f"(2*FETCH_SIZE+WRITE_SIZE)*1024 per MI355X_MICROARCH.md; counts L2 misses incl. Infinity-Cache hits, "
                                   f"i.e. an upper bound on HBM bytes")
                else:
                    traffic_src = "profiles/traffic.json is stale (kernel sources changed since it was collected): not used"
            except Exception:
                traffic = None
        if traffic:
            achieved = traffic / (avg_ms * 1e-3) / 1e9
            basis = "measured L2-miss traffic (PMC) / HIP-event launch time"
        else:  # no counter figure for this workload: the compulsory bytes are the only physically meaningful numerator
            achieved = comp_gbs
            basis = "compulsory bytes (every operand once) / HIP-event launch time; no PMC traffic figure for this workload"
        out = {
            "metric": "positive-edges/sec (train step)",
            "value": world * B * args.steps / elapsed,
            "unit": "positive-edges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "ms_per_step_p10_p50_p90": [round(pct(0.1), 4), round(pct(0.5), 4), round(pct(0.9), 4)],
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"LightGCN train step, ONE synthetic bipartite graph {spec.num_users}x{I} users x items, "
                                    f"{spec.num_edges} edges (seed {spec.seed}, {S.C4_BLOCKS} user blocks), sharded by user over "
                                    f"{world} GPU(s): {U} users / {ei.shape[1]} edges per GPU, {K}-layer D={D}, global batch "
                                    f"{B * world} positive edges, on-device sampling, BPR + dense Adam; BASELINE.json configs[3]")
                       if strong else
                                   (f"LightGCN train step, synthetic bipartite {U}x{I} users x items per GPU, "
                                    f"{spec.num_edges} edges per GPU{' (uniform endpoints)' if args.uniform else ''} (symmetric adjacency nnz={nnz}), "
                                    f"{K}-layer D={D}, batch {B} positive edges per GPU, on-device sampling, "
                                    f"BPR + dense Adam; BASELINE.json configs[1]"),
                       "parallelism": "1 GPU" if world == 1 else f"user-sharded x{world}, items replicated, "
                                                                 f"{'RCCL' if backend == 'nccl' else backend} all-reduce of item rows per layer"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "basis": basis, "traffic_source": traffic_src,
                         "kernel": "mi_spmm_csr_f32 dense launch (spmm_items_kernel + spmm_rows_kernel + spmm_fixup_kernel, SPARSE=false)",
                         "avg_launch_ms": avg_ms,
                         "compulsory": {"bytes_per_launch": compulsory, "GBps": comp_gbs, "frac_of_peak": comp_gbs / HBM_PEAK_GBS,
                                        "note": "every operand read once, every output written once (SURVEY 8d lower bound)"},
                         "algorithmic": {"bytes_per_launch": algo, "GBps": algo_gbs, "frac_of_peak": algo_gbs / HBM_PEAK_GBS,
                                         "note": "SURVEY 8d byte model: one gathered row per entry, NO cache-reuse credit; "
                                                 "a work rate, not a bound (exceeds the HBM peak when rows are served by L2)"},
                         "launches_timed": len(spmm_ms),
                         "dense_launches_per_step": len(dense) / args.steps,
                         "sparse_launches_per_step": len(sparse_ms) / args.steps,
                         "sparse_launch_avg_ms": (sum(sparse_ms) / len(sparse_ms)) if sparse_ms else None,
                         "dense_launches_with_adam_epilogue_per_step": len(fused_ms) / args.steps,
                         "dense_with_adam_epilogue_avg_ms": (sum(fused_ms) / len(fused_ms)) if fused_ms else None},
            "step_form": "plain" if args.plain_step else "sparse_batch",
            "node_order": "locality (items by popularity, users by coldest item)" if getattr(trainer, "order", None) is not None else "generator ids",
            "loss": loss_val, "graph_gen_s": round(t_gen, 1), "backend": backend if world > 1 else None,
        }
    else:
        out = None

    # ---- extra legs, N=1 only, after the timed region ------------------------------------------------
    if world == 1 and not args.plain_step and not args.no_plain_leg:
        # the straightforward step shape (full `final`, dense gradient buffer, separate Adam): same parameters to
        # rounding (tests/test_gpu_lightgcn.py::test_sparse_batch_step_equals_plain_step), more bytes
        trainer.to_original_order()  # the second trainer relabels the table itself
        plain = LightGCNTrainer(model, adj, inter, lr=1e-3, Lambda=1e-6, batch_size=B, seed=11, sparse_batch=False)
        for _ in range(max(args.warmup, 1)):
            plain.step()
        t.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            plain.step()
        t.cuda.synchronize()
        out["plain_step_ms"] = 1e3 * (time.perf_counter() - t1) / args.steps
        plain.finish()
        del plain
    if world == 1 and not args.no_map and not strong and not args.uniform:
        held = S.heldout_edges(spec, ei, args.map_users).to(dev)
        out["map_at_12"] = map_at_12(model, trainer, inter, held, args.map_steps)
    if rank == 0:
        if table0 is not None:
            out["cpu_baseline"] = cpu_baseline(ei, U, I, args, B, table0)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

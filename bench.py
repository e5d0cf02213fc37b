#!/usr/bin/env python3
"""bench.py — positive-edges/sec of the LightGCN train step on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch: K-layer propagate forward, on-device
sampling of B positive edges + negatives, fused BPR forward/backward, K-layer propagate backward,
dense Adam over the whole table (run_pipeline_lightgcn.py:117-159).

  --config c4 (default)  BASELINE.json configs[3], the configuration the metric is quoted on ("H&M-scale graph"; it fits
                         one GPU): ONE fixed graph, 8M users x 100K items, 100M edges (seed 3), sharded by
                         user_id // ceil(U/N) over the N ranks (items replicated, one RCCL all-reduce of the item
                         rows per layer), global batch 131072: total work fixed, "scaling": "strong".
  --config c2            BASELINE.json configs[1]: synthetic bipartite 1M users x 100K items, 10M edges, LightGCN
                         3-layer D=128 (SURVEY 8d "C2": symmetric adjacency nnz=20M, B=16384).  At N>1 every rank
                         owns its own 1M-user / 10M-edge shard of a weak-scaled graph: "scaling": "weak".

The default N=1 run carries, after the timed region, the metric's other configurations as blocks of the same JSON line:
  "c2"         BASELINE configs[1] on one GPU (--steps/--warmup as the headline; ms/step, positive-edges/s, roofline with
               live PMC traffic, plain_step_ms, cpu_baseline); with --config c2 the side block is "c4_n1"
                                                                                             [--no-side to skip]
  "ranker_c3"  BASELINE configs[2]: the encoder-decoder ranker's training loop at the full H&M shape, 24 users per
               batch, on-device 2-hop sampling (tools/bench_ranker.py's bench line)          [--no-ranker to skip]
  pinsage_c5   BASELINE configs[4] at N = 1: PinSAGE item-item training at the H&M shape, the reference's batch of 32
               pairs (tools/bench_pinsage.py)                                               [--no-pinsage to skip]
  "e2e_c3"     BASELINE configs[3]'s "candidate-gen + ranker end-to-end" at the H&M shape on one GPU: per-stage seconds of LightGCN
               steps -> top-100 dump -> matchers -> ranker iterations -> device-built evaluation, MAP@12 on held-out purchases
               (tools/e2e_hm_scale.py)                                                        [--no-e2e to skip]
  "map_at_12"  ranking quality on a PLANTED-structure graph (synthetic.SyntheticSpec.communities), layer-0 predictor vs
               propagated embeddings vs popularity                                           [--no-map to skip]

roofline.traffic is measured in THIS invocation: before the parent touches the GPU it runs itself twice per
configuration as a short child under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, no trace
domains) and reads the per-dispatch counters of the propagate kernels.                       [--no-pmc to skip]

Launch: python bench.py --gpus N --steps K --warmup W      (N>1: the parent starts N worker processes itself, before
        touching the GPU) or, equivalently, python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
Prints ONE JSON line (rank 0).
"""
import argparse
import collections
import csv
import glob
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); the guide's measured rates: 6.0-6.1 TB/s for an in-order
                       # sweep, 5.5-5.8 TB/s for random whole rows gathered from beyond the caches
C4_GLOBAL_BATCH = 131072
# the planted-structure graph of the MAP@12 leg: small enough that 100 steps are a few epochs
MAP_SPEC = dict(num_users=50_000, num_items=5_000, num_edges=1_000_000, seed=5, communities=32, community_mix=0.9, deg_max=2000)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=("c2", "c4"), default="c4")
    ap.add_argument("--users", type=int, default=None)
    ap.add_argument("--items", type=int, default=None)
    ap.add_argument("--edges", type=int, default=None)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--layers", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="positive edges per GPU (c2) / per job (c4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--plain-leg", action="store_true", help="time the straightforward step as plain_step_ms for --config c4 too")
    ap.add_argument("--plain-step", action="store_true",
                    help="A/B: time the straightforward step (full final, dense gradient buffer) as the headline instead of the byte-saving one")
    ap.add_argument("--no-plain-leg", action="store_true", help="skip the extra plain-step timing (plain_step_ms)")
    ap.add_argument("--no-map", action="store_true", help="skip the MAP@12 leg")
    ap.add_argument("--map-steps", type=int, default=100, help="train steps on the planted-structure graph before MAP@12 is scored")
    ap.add_argument("--map-users", type=int, default=20000)
    ap.add_argument("--no-side", "--no-c4", dest="no_side", action="store_true",
                    help="skip the LightGCN side block (the other of configs[1] / configs[3] on one GPU)")
    ap.add_argument("--no-ranker", action="store_true", help="skip the ranker_c3 block (BASELINE configs[2])")
    ap.add_argument("--no-pinsage", action="store_true", help="skip the pinsage_c5 block (BASELINE configs[4] at N = 1)")
    ap.add_argument("--pinsage-iters", type=int, default=300)
    ap.add_argument("--no-e2e", action="store_true", help="skip the e2e_c3 block (configs[3]: candidate generation -> ranker, H&M shape)")
    ap.add_argument("--e2e-lightgcn-steps", type=int, default=300)
    ap.add_argument("--e2e-ranker-iters", type=int, default=300)
    ap.add_argument("--no-topk", action="store_true", help="skip the topk_a10 block (exact top-K with exclusion, users/s)")
    ap.add_argument("--ranker-steps", type=int, default=400)
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 --pmc passes (roofline.traffic falls back to profiles/traffic.json)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)  # the short run the PMC passes profile
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
    """Identifies the code a PMC traffic figure was collected on: the propagate kernels AND the host code that decides
    their plans, launch forms and the node order (all of which move the traffic)."""
    h = hashlib.sha256()
    pkg = os.path.join(ROOT, "laplace-gnn-recommendation_amd")
    for f in ("csrc/spmm.hip", "csrc/common.hpp", "ops.py", "interactions.py", "trainer.py", "sparse.py"):
        with open(os.path.join(pkg, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


# ---- live PMC traffic ---------------------------------------------------------------------------------------------------

def _short_kernel(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:80]


def is_dense_spmm_kernel(k: str) -> bool:
    """A kernel of the plain DENSE propagate launch: SPARSE=false and no optimizer epilogue."""
    m = re.match(r"spmm_(rows_hot|rows|items|fixup|sweep)_kernel<([^>]*)>", k)
    if not m:
        return False
    a = [x.strip() for x in m.group(2).split(",")]
    flags = {"rows": a[4:6], "items": a[4:5], "fixup": a[2:4], "sweep": a[2:3], "rows_hot": a[2:3]}[m.group(1)]
    return all(f == "false" for f in flags)


def pmc_traffic(config: str, passthrough: list, timeout_s: float = 240.0):
    """L2-miss (fabric) bytes per dense propagate launch of `config`, measured NOW: this script run as a short child
    (3 steps) under rocprofv3 --pmc FETCH_SIZE and again under --pmc WRITE_SIZE (counters only: no trace domain beside
    them), per-dispatch counters averaged per kernel, bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (MI355X_MICROARCH.md:
    counters in KiB, FETCH_SIZE tallies 128-B requests at 64 B on gfx950).  Must be called BEFORE this process touches
    the GPU (the children are ordinary fresh processes).  Returns (bytes, per_kernel dict, note) or (None, None, why)."""
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, None, "rocprofv3 not found"
    per = {}
    t0 = time.time()
    with tempfile.TemporaryDirectory(dir="/tmp", prefix="laplace_pmc_") as tmp:
        env = dict(os.environ, TMPDIR="/tmp")
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE"):
            env.pop(k, None)
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [prof, "--pmc", counter, "-d", out, "--output-format", "csv", "--", sys.executable, os.path.abspath(__file__),
                   "--pmc-child", "--config", config, "--steps", "3", "--warmup", "1"] + passthrough
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout_s)
            except subprocess.TimeoutExpired:
                return None, None, f"rocprofv3 --pmc {counter} pass timed out"
            if r.returncode != 0:
                return None, None, f"rocprofv3 --pmc {counter} pass failed (rc {r.returncode}): {r.stderr.decode(errors='replace')[-300:]}"
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return None, None, f"rocprofv3 --pmc {counter} wrote no counter file"
            agg = collections.defaultdict(list)
            for row in csv.DictReader(open(files[0])):
                if row.get("Counter_Name", counter) == counter:
                    agg[_short_kernel(row["Kernel_Name"])].append(float(row["Counter_Value"]))
            per[counter] = {k: sum(v) / len(v) for k, v in agg.items()}
    kernels = {}
    for k in set(per["FETCH_SIZE"]) | set(per["WRITE_SIZE"]):
        if k.startswith("spmm_") or "adam" in k:
            kernels[k] = (2.0 * per["FETCH_SIZE"].get(k, 0.0) + per["WRITE_SIZE"].get(k, 0.0)) * 1024.0
    dense = sum(v for k, v in kernels.items() if is_dense_spmm_kernel(k))
    if dense <= 0:
        return None, None, "no dense propagate kernel in the counter file"
    note = (f"live: this invocation ran itself as a 3-step child under rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate "
            f"passes, {time.time() - t0:.0f} s) before the timed region; (2*FETCH_SIZE+WRITE_SIZE)*1024 per dispatch, averaged, "
            f"summed over the dense launch's kernels; counts L2 misses incl. Infinity-Cache hits, i.e. an upper bound on HBM bytes")
    return dense, kernels, note


# ---- workload construction (shared by the headline, the PMC child and the c4_n1 block) -------------------------------------

def build_workload(config: str, args, world: int, rank: int, dev, plain: bool = False, want_table0: bool = False):
    import torch as t
    from laplace_amd import synthetic as S
    from laplace_amd.interactions import Interactions
    from laplace_amd.model.lightgcn import LightGCN
    from laplace_amd.trainer import LightGCNTrainer
    base = S.C4 if config == "c4" else S.C2
    custom = config == args.config   # size overrides belong to the headline configuration only
    spec = S.SyntheticSpec((args.users if custom else None) or base.num_users, (args.items if custom else None) or base.num_items,
                           (args.edges if custom else None) or base.num_edges, seed=base.seed, uniform=args.uniform and custom)
    strong = config == "c4"
    t_gen = time.perf_counter()
    if strong:
        b0, b1 = S.shard_blocks(S.C4_BLOCKS, world, rank)
        ei = S.generate_blocks(spec, S.C4_BLOCKS, b0, b1)
        U = spec.num_users // world
        B = ((args.batch if custom else None) or C4_GLOBAL_BATCH) // world
    else:
        ei = S.generate(S.shard_spec(spec, rank) if world > 1 else spec)
        U = spec.num_users
        B = (args.batch if custom else None) or 16384
    t_gen = time.perf_counter() - t_gen
    I, D, K = spec.num_items, args.dim, args.layers
    t.manual_seed(1234 + rank)
    model = LightGCN(U, I, embedding_dim=D, num_iterations=K)
    table0 = model.table().clone() if want_table0 else None
    model.to(dev)
    inter = Interactions(ei.to(dev), U, I)
    adj = None
    if world == 1:
        adj = inter.adjacency("bipartite")
        trainer = LightGCNTrainer(model, adj, inter, lr=1e-3, Lambda=1e-6, batch_size=B, seed=7,
                                  sparse_batch=not plain, reorder=False if args.no_reorder else None)
    else:
        from laplace_amd.dist import ShardedLightGCNTrainer
        trainer = ShardedLightGCNTrainer(model, inter, lr=1e-3, Lambda=1e-6, batch_size=B, seed=7 + rank,
                                         sparse_batch=not plain, reorder=False if args.no_reorder else None)
    return dict(spec=spec, ei=ei, U=U, I=I, D=D, K=K, B=B, model=model, inter=inter, adj=adj, trainer=trainer,
                table0=table0, t_gen=t_gen, strong=strong)


def timed_steps(trainer, steps: int, warmup: int, sync):
    """W untimed + K timed steps bracketed by sync(); returns (elapsed s, per-step ms sorted, spmm events, last loss)."""
    import torch as t
    from laplace_amd import ops
    for _ in range(warmup):
        trainer.step()
    sync()
    ops.SPMM_EVENTS = []  # (start, end) HIP events around every propagate launch, on the launch stream
    marks = [t.cuda.Event(enable_timing=True) for _ in range(steps + 1)]  # per-step spread; no sync inside the loop
    t0 = time.perf_counter()
    marks[0].record()
    loss = None
    for i in range(steps):
        loss = trainer.step()
        marks[i + 1].record()
    sync()
    elapsed = time.perf_counter() - t0
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
    events, ops.SPMM_EVENTS = ops.SPMM_EVENTS, None
    return elapsed, per_step, events, loss


def roofline_block(events, nnz: int, n_rows: int, D: int, steps: int, traffic, traffic_src) -> dict:
    """The roofline is quoted on the DENSE propagate (every entry of the adjacency slice gathered): kernels
    spmm_*_kernel<..., false>.  The sparse-operand launches of the byte-saving step (last forward layer at the batch
    rows, first backward layer over the non-zero gradient rows) gather a data-dependent subset and are reported beside
    it, not mixed in; the last backward product carries the Adam update in its epilogue (+6 table streams): timed apart."""
    dense = [(s.elapsed_time(e), a) for s, e, kind, _, _, a in events if kind == "dense"]
    sparse_ms = [s.elapsed_time(e) for s, e, kind, _, _, _ in events if kind == "sparse"]
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
    if traffic:
        achieved = traffic / (avg_ms * 1e-3) / 1e9
        basis = ("L2-miss (fabric) traffic, incl. Infinity-Cache hits (PMC FETCH_SIZE / WRITE_SIZE) / HIP-event launch time: an UPPER "
                 "bound on HBM bytes, not HBM bytes (the HBM side of the Infinity Cache has no counter on this pool)")
    else:  # no counter figure for this workload: the compulsory bytes are the only physically meaningful numerator
        achieved = comp_gbs
        basis = "compulsory bytes (every operand once) / HIP-event launch time; no PMC traffic figure for this workload"
    # "bound" keeps the bench contract's vocabulary (hbm | mfma): the launch is memory-side bound.  What `achieved` measures is
    # named by `measured_level` / `basis`: traffic that left the L2s, whether the Infinity Cache or HBM served it.
    return {"bound": "hbm", "measured_level": "L2-miss (fabric) traffic, incl. Infinity-Cache hits" if traffic else "compulsory bytes",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "basis": basis, "traffic_source": traffic_src,
            "frac_of_measured_random_gather_ceiling": achieved / 5650.0,
            "ceiling_note": "MI355X_MICROARCH.md measures 5.5-5.8 TB/s for random whole rows gathered from beyond the caches (6.0-6.1 "
                            "TB/s for an in-order HBM sweep): the denominator of this extra fraction is 5.65 TB/s",
            "kernel": "mi_spmm_csr_f32 dense launch (spmm_sweep_kernel or spmm_items_kernel + spmm_rows_kernel + spmm_fixup_kernel, SPARSE=false)",
            "avg_launch_ms": avg_ms,
            "compulsory": {"bytes_per_launch": compulsory, "GBps": comp_gbs, "frac_of_peak": comp_gbs / HBM_PEAK_GBS,
                           "note": "every operand read once, every output written once (SURVEY 8d lower bound)"},
            "algorithmic": {"bytes_per_launch": algo, "GBps": algo_gbs, "frac_of_peak": algo_gbs / HBM_PEAK_GBS,
                            "note": "SURVEY 8d byte model: one gathered row per entry, NO cache-reuse credit; "
                                    "a work rate, not a bound (exceeds the HBM peak when rows are served by L2)"},
            "launches_timed": len(spmm_ms),
            "dense_launches_per_step": len(dense) / steps,
            "sparse_launches_per_step": len(sparse_ms) / steps,
            "sparse_launch_avg_ms": (sum(sparse_ms) / len(sparse_ms)) if sparse_ms else None,
            "dense_launches_with_adam_epilogue_per_step": len(fused_ms) / steps,
            "dense_with_adam_epilogue_avg_ms": (sum(fused_ms) / len(fused_ms)) if fused_ms else None}


def offline_traffic(default_workload: bool, config: str = "c4"):
    """profiles/traffic.json (tools/prof_bench.sh on the DEFAULT configuration, BASELINE configs[3]; traffic_c2.json for
    --config c2): used only when the live passes are off or failed; refused when the kernels or the host code that shapes
    their launches changed since it was collected."""
    tf = os.path.join(ROOT, "profiles", "traffic.json" if config == "c4" else f"traffic_{config}.json")
    if not (os.path.exists(tf) and default_workload):
        return None, None
    try:
        tj = json.load(open(tf))
        if tj.get("kernel_source_hash") == kernel_source_hash():
            return tj.get("spmm_hbm_bytes_per_launch"), (
                f"{os.path.basename(tf)} <- {tj.get('source')}: offline rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on "
                f"the code with hash {tj.get('kernel_source_hash')} (matches this build); (2*FETCH_SIZE+WRITE_SIZE)*1024 per "
                f"MI355X_MICROARCH.md; counts L2 misses incl. Infinity-Cache hits, i.e. an upper bound on HBM bytes")
        return None, f"profiles/{os.path.basename(tf)} is stale (kernel or launch code changed since it was collected): not used"
    except Exception:
        return None, None


def cpu_baseline(ei, U, I, K, B, table0, steps: int, warm: bool, faithful: bool):
    """The oracle's CPU port of the same step, timed on this box's host cores (rank 0, N=1 only)."""
    import torch as t
    from oracle import lightgcn_ref as R
    cores = R.clib().ref_num_threads()
    t.set_num_threads(cores)
    r, c = R.bipartite_edges(ei[0], ei[1], U)
    rowptr, cs, _ = R.sparse_tensor_csr(r, c, U + I, U + I)
    val = R.gcn_norm_csr(rowptr, cs)
    del r, c
    port = R.CpuTrainPort(table0, rowptr, cs, val, U, K, 1e-3, 1e-6)
    g = t.Generator().manual_seed(1)
    E = ei.shape[1]

    def batch():
        e = t.randint(0, E, (B,), generator=g)
        return ei[0][e], ei[1][e], t.randint(0, I, (B,), generator=g)  # negatives pre-drawn

    if warm:
        port.step(batch())  # warm-up (page-faults the buffers)
    t0 = time.perf_counter()
    for _ in range(steps):
        port.step(batch())
    dt = (time.perf_counter() - t0) / steps
    extra = None
    if faithful:
        # informational: the reference draws a negative for EVERY train edge every iteration
        # (data/lightgcn_loader.py:95-112, np.isin rejection) before picking the batch; one draw is timed
        import random
        import numpy as np
        t1 = time.perf_counter()
        R.sample_mini_batch(B, ei, np.random.default_rng(0), random.Random(0))
        ts = time.perf_counter() - t1
        extra = {"sampler_s_per_step": round(ts, 2), "value": B / (dt + ts), "unit": "positive-edges/s",
                 "note": "model-only step + the reference's per-iteration O(E) negative sampling; not the denominator of any claim"}
    return {"value": B / dt, "unit": "positive-edges/s", "cores": int(cores), "kind": "port", "faithful_step": extra,
            "sample": f"{steps} full model-only train step(s) (fwd+BPR+bwd+Adam, negatives pre-drawn{'' if warm else ', no warm-up step'}) of the "
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
    # yardstick on the same users: the popularity predictor
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


def map_leg(args, dev) -> dict:
    """MAP@12 on a graph with PLANTED structure: on the plain benchmark graph popularity is all there is to learn, so
    no predictor can beat the popularity baseline and the figure says nothing about the trainer.  Here users and items
    belong to latent groups (synthetic.SyntheticSpec.communities); a trainer that works lifts the reference's layer-0
    predictor above the popularity predictor within ~100 steps (tests/test_gpu_acceptance.py asserts the ordering)."""
    import torch as t
    from laplace_amd import synthetic as S
    from laplace_amd.interactions import Interactions
    from laplace_amd.model.lightgcn import LightGCN
    from laplace_amd.trainer import LightGCNTrainer
    spec = S.SyntheticSpec(**MAP_SPEC)
    ei = S.generate(spec)
    held = S.heldout_edges(spec, ei, args.map_users).to(dev)
    t.manual_seed(0)
    model = LightGCN(spec.num_users, spec.num_items, args.dim, args.layers).to(dev)
    inter = Interactions(ei.to(dev), spec.num_users, spec.num_items)
    tr = LightGCNTrainer(model, inter.adjacency("bipartite"), inter, lr=0.05, Lambda=1e-6, batch_size=16384, seed=7)
    out = map_at_12(model, tr, inter, held, args.map_steps)
    out["workload"] = (f"planted-structure synthetic {spec.num_users}x{spec.num_items}, {spec.num_edges} edges, {spec.communities} latent "
                       f"groups (own-group draw with p={spec.community_mix}), LightGCN {args.layers}-layer D={args.dim}, batch 16384, lr 0.05, "
                       f"{args.map_steps} steps; one held-out positive per scored user drawn from the same law")
    return out


def plain_step_leg(model, adj, inter, B: int, steps: int, warmup: int) -> float:
    """ms per step of the straightforward step shape (full `final`, dense gradient buffer, 6 dense propagates, separate
    Adam): same parameters to rounding (tests/test_gpu_lightgcn.py::test_sparse_batch_step_equals_plain_step), more
    bytes.  Quote this figure whenever the number is compared with a reference-shaped step."""
    import torch as t
    from laplace_amd.trainer import LightGCNTrainer
    plain = LightGCNTrainer(model, adj, inter, lr=1e-3, Lambda=1e-6, batch_size=B, seed=11, sparse_batch=False)
    for _ in range(max(warmup, 1)):
        plain.step()
    t.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(steps):
        plain.step()
    t.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t1) / steps
    plain.finish()
    return ms


def side_block(config: str, args, dev, traffic, traffic_src) -> dict:
    """The OTHER LightGCN configuration on one GPU, as a block of the line: "c2" (BASELINE configs[1]) under the default
    headline, "c4_n1" (configs[3]) under --config c2.  Same --steps / --warmup, same roofline fields, own cpu_baseline."""
    import torch as t
    want_cpu = not args.no_cpu_baseline
    w = build_workload(config, args, 1, 0, dev, want_table0=want_cpu)
    trainer = w["trainer"]
    sync = t.cuda.synchronize
    sync()
    steps, warmup = args.steps, max(args.warmup, 3)
    elapsed, per_step, events, loss = timed_steps(trainer, steps, warmup, sync)
    nnz, n_rows = trainer.adj_fwd.nnz, trainer.adj_fwd.n_rows
    spec, B = w["spec"], w["B"]
    ref = "BASELINE.json configs[3] at N=1" if config == "c4" else "BASELINE.json configs[1]"
    block = {"metric": "positive-edges/sec (train step)", "value": B * steps / elapsed, "unit": "positive-edges/s",
             "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps,
             "ms_per_step_p10_p50_p90": [round(per_step[min(len(per_step) - 1, int(q * len(per_step)))], 4) for q in (0.1, 0.5, 0.9)],
             "workload": (f"LightGCN train step, ONE synthetic bipartite graph {spec.num_users}x{w['I']} users x items, {spec.num_edges} edges "
                          f"(seed {spec.seed}), all of it on one GPU (symmetric adjacency nnz={nnz}), {w['K']}-layer D={w['D']}, batch {B} "
                          f"positive edges, on-device sampling, BPR + dense Adam; {ref}"),
             "roofline": roofline_block(events, nnz, n_rows, w["D"], steps, traffic, traffic_src),
             "node_order": "locality (items by popularity, users by coldest item)" if getattr(trainer, "order", None) is not None else "generator ids",
             "loss": float(loss), "graph_gen_s": round(w["t_gen"], 1)}
    if config == "c2" and not args.no_plain_leg:
        trainer.to_original_order()  # the second trainer relabels the table itself
        block["plain_step_ms"] = plain_step_leg(w["model"], w["adj"], w["inter"], B, steps, warmup)
        block["plain_step_positive_edges_per_s"] = B * 1e3 / block["plain_step_ms"]
    trainer.finish()
    ei, U, I, K, table0 = w["ei"], w["U"], w["I"], w["K"], w["table0"]
    del trainer, w, events
    t.cuda.empty_cache()
    if table0 is not None:
        block["cpu_baseline"] = cpu_baseline_for(config, args, ei, U, I, K, B, table0)
    return block


def cpu_baseline_for(config: str, args, ei, U, I, K, B, table0) -> dict:
    """The bounded CPU sample of a configuration: C2 = one warm-up + --cpu-steps timed steps (~2.4 s each on 128 cores)
    + one draw of the reference's O(E) sampler; C4 = one warm-up + ONE timed step (~20 s each)."""
    if config == "c4":
        return cpu_baseline(ei, U, I, K, B, table0, steps=1, warm=True, faithful=False)
    return cpu_baseline(ei, U, I, K, B, table0, steps=args.cpu_steps, warm=True, faithful=not args.no_cpu_faithful)


def ranker_block(args) -> dict:
    """BASELINE configs[2]: the ranker's training loop at the full H&M shape (tools/bench_ranker.py, bench-line mode)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_ranker", os.path.join(ROOT, "tools", "bench_ranker.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.bench_line(users=1_371_980, items=105_542, edges=31_800_000, batch=24, steps=args.ranker_steps, warmup=50,
                          hops=2, fanout=64, cpu=not args.no_cpu_baseline)


def pinsage_block(args) -> dict:
    """BASELINE configs[4] at N = 1: PinSAGE item-item training at the H&M shape with the reference's settings (32 pairs per batch,
    10 walks, restart 0.5, T = 3, 2 layers, hidden 16, lr 3e-5: pinsage/model.py:143-153) — tools/bench_pinsage.py; the
    iteration is one C call (mi_pinsage_step_f32) with the next batch sampled on a side stream.  The headline of the block is
    the "3-hop random-walk sampler" BASELINE names (walks of 3 traversals); the reference's own default (walks of 2) rides
    beside it, measured on the same graph."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_pinsage", os.path.join(ROOT, "tools", "bench_pinsage.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = mod.bench_line(iters=args.pinsage_iters, walk_length=3)
    out["reference_default_walk_length_2"] = mod.bench_line(iters=args.pinsage_iters, walk_length=2)
    mod._GRAPH_CACHE.clear()
    return out


def e2e_block(args) -> dict:
    """BASELINE configs[3]'s "LightGCN candidate-gen + GNN ranker end-to-end" at the H&M shape on one GPU, per-stage seconds and
    MAP@12 on held-out purchases of a planted-structure graph (tools/e2e_hm_scale.py::run): LightGCN steps -> top-100 dump ->
    matchers -> ranker iterations -> device-built evaluation."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("e2e_hm_scale", os.path.join(ROOT, "tools", "e2e_hm_scale.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.run(lightgcn_steps=args.e2e_lightgcn_steps, ranker_iters=args.e2e_ranker_iters)


def topk_block(args) -> dict:
    """SURVEY row a10 at C2's item count: users/s of the exact top-K with exclusion for k = 12 (evaluation) and k = 256 (the
    matcher dump), default path (bf16x3 prefilter + exact rescoring) beside the f32 fused kernel, ids compared, and the
    reference's per-user host loop (oracle, 64 users) — tools/bench_topk.py."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_topk", os.path.join(ROOT, "tools", "bench_topk.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.bench_line(full=True, n_q=16384, cpu_loop=not args.no_cpu_baseline)


_T0 = time.perf_counter()
_LEGS = {}   # wall seconds of each leg of this invocation (reported as "wall_s": where a default run's minutes go)


def _leg(name: str, since: float) -> float:
    now = time.perf_counter()
    _LEGS[name] = round(_LEGS.get(name, 0.0) + now - since, 1)
    return now


def main():
    args = parse_args()
    from laplace_amd import launch
    if args.gpus > 1 and not launch.launched():
        # the parent never touches the GPU; one deadline from launch, every worker watched (laplace_amd/launch.py)
        sys.exit(launch.self_launch(__file__, sys.argv[1:], args.gpus, timeout_s=float(os.environ.get("LAPLACE_BENCH_DEADLINE_S", 900))))
    world_env = int(os.environ.get("WORLD_SIZE", "1"))

    # ---- live PMC passes: children first, while this process has not touched the GPU ----------------------------------
    pmc = {}
    side = "c2" if args.config == "c4" else "c4"     # the other LightGCN configuration rides as a block at N = 1
    custom_size = any(v is not None for v in (args.users, args.items, args.edges, args.batch)) or args.uniform
    want_side = world_env == 1 and not args.no_side and not args.pmc_child and not args.plain_step and not custom_size
    if world_env == 1 and not args.no_pmc and not args.pmc_child and not args.plain_step:
        passthrough = []
        for flag, val in (("--users", args.users), ("--items", args.items), ("--edges", args.edges), ("--batch", args.batch)):
            if val is not None:
                passthrough += [flag, str(val)]
        passthrough += ["--dim", str(args.dim), "--layers", str(args.layers)]
        if args.no_reorder:
            passthrough.append("--no-reorder")
        if args.uniform:
            passthrough.append("--uniform")
        tl = time.perf_counter()
        pmc[args.config] = pmc_traffic(args.config, passthrough)
        tl = _leg("pmc_children_" + args.config, tl)
        if want_side:
            pmc[side] = pmc_traffic(side, ["--dim", str(args.dim), "--layers", str(args.layers)] + (["--no-reorder"] if args.no_reorder else []))
            _leg("pmc_children_" + side, tl)

    import torch as t
    import torch.distributed as dist

    if world_env != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_env}")
    if not t.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    # LAPLACE_BENCH_BACKEND=gloo + LAPLACE_BENCH_ONE_GPU=1 rehearse the N>1 path on a 1-GPU box
    # (ranks share the card, collectives staged through the host); never used for a reported number.
    backend = launch.backend_name()
    rank, world, dev = launch.init_distributed()   # rendezvous + collectives time out instead of hanging

    from laplace_amd import ops, synthetic as S
    from laplace_amd.trainer import LightGCNTrainer

    if os.environ.get("LAPLACE_SPMM_TWO_STREAMS") is not None:  # A/B switch
        ops.SPMM_TWO_STREAMS = int(os.environ["LAPLACE_SPMM_TWO_STREAMS"] or 0)   # 1: short rows enqueued first, 2: split rows first
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline and not args.pmc_child
    tl = time.perf_counter()
    w = build_workload(args.config, args, world, rank, dev, plain=args.plain_step, want_table0=want_cpu)
    tl = _leg("import_and_graph_setup", tl)
    spec, ei, U, I, D, K, B = w["spec"], w["ei"], w["U"], w["I"], w["D"], w["K"], w["B"]
    model, inter, adj, trainer, strong = w["model"], w["inter"], w["adj"], w["trainer"], w["strong"]
    default_workload = not custom_size and (D, K) == (128, 3)   # the sizes the offline traffic files were collected on
    nnz, n_rows = trainer.adj_fwd.nnz, trainer.adj_fwd.n_rows
    t.cuda.synchronize()

    def sync():
        if world > 1:
            dist.barrier()
        t.cuda.synchronize()

    elapsed, per_step, events, loss = timed_steps(trainer, args.steps, args.warmup, sync)
    if args.pmc_child:   # the profiler has what it came for
        return
    tl = _leg("trainer_setup_and_timed_region", tl)
    pct = lambda q: per_step[min(len(per_step) - 1, int(q * len(per_step)))]
    loss_val = float(loss)

    if world > 1:
        tt = t.tensor([elapsed], device=dev, dtype=t.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt)

    out = None
    if rank == 0:
        traffic, traffic_src = None, None
        if args.config in pmc:
            traffic, _, traffic_src = pmc[args.config]
            if traffic is None:
                traffic_src = f"live PMC passes failed ({traffic_src})"
        if traffic is None and world == 1:
            off, off_src = offline_traffic(default_workload, args.config)
            if off is not None:
                traffic, traffic_src = off, off_src
            elif off_src and not traffic_src:
                traffic_src = off_src
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
            "roofline": roofline_block(events, nnz, n_rows, D, args.steps, traffic, traffic_src),
            "step_form": "plain" if args.plain_step else "sparse_batch",
            "node_order": "locality (items by popularity, users by coldest item)" if getattr(trainer, "order", None) is not None else "generator ids",
            "loss": loss_val, "graph_gen_s": round(w["t_gen"], 1), "backend": backend if world > 1 else None,
        }

    # ---- extra legs, N=1 only, after the timed region ------------------------------------------------
    tl = time.perf_counter()
    if world == 1 and not args.plain_step and not args.no_plain_leg and (args.config == "c2" or args.plain_leg):
        trainer.to_original_order()  # the second trainer relabels the table itself
        out["plain_step_ms"] = plain_step_leg(model, adj, inter, B, args.steps, args.warmup)
        out["plain_step_positive_edges_per_s"] = B * 1e3 / out["plain_step_ms"]
        tl = _leg("plain_step_leg", tl)
    if rank == 0 and want_cpu:
        trainer.finish()
        out["cpu_baseline"] = cpu_baseline_for(args.config, args, ei, U, I, K, B, w["table0"])
        tl = _leg("cpu_baseline", tl)
    if world == 1 and rank == 0:
        # free the headline's state before the other configurations
        del trainer, model, inter, adj, w, events
        t.cuda.empty_cache()
        extras = not custom_size and not args.plain_step
        if want_side:
            s_traffic, _, s_src = pmc.get(side, (None, None, "live PMC passes off (--no-pmc)"))
            if s_traffic is None:
                s_src = f"no live figure ({s_src})"
            out["c2" if side == "c2" else "c4_n1"] = side_block(side, args, dev, s_traffic, s_src)
            tl = _leg(f"{side} side block (incl. its CPU baseline)", tl)
        if extras and not args.no_map:
            out["map_at_12"] = map_leg(args, dev)
            t.cuda.empty_cache()
            tl = _leg("map_at_12", tl)
        if extras and not args.no_ranker:
            out["ranker_c3"] = ranker_block(args)
            t.cuda.empty_cache()
            tl = _leg("ranker_c3 (incl. graph generation and the CPU twin)", tl)
        if extras and not args.no_pinsage:
            out["pinsage_c5"] = pinsage_block(args)
            tl = _leg("pinsage_c5 (incl. graph generation)", tl)
        if extras and not args.no_e2e:
            t.cuda.empty_cache()
            out["e2e_c3"] = e2e_block(args)
            tl = _leg("e2e_c3 (incl. graph generation)", tl)
        if extras and not args.no_topk:
            t.cuda.empty_cache()
            out["topk_a10"] = topk_block(args)
            tl = _leg("topk_a10 (incl. graph generation and the host loop)", tl)
    if rank == 0:
        _LEGS["total"] = round(time.perf_counter() - _T0, 1)
        out["wall_s"] = dict(_LEGS)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

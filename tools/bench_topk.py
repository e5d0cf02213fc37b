#!/usr/bin/env python3
"""Throughput of the batched exact top-K with exclusion at C2's item count:
users/s for k = 12 (evaluation) and k = 256 (the matcher dump of run_pipeline_lightgcn.py:212-221), on the default
path (bf16x3 prefilter + exact rescoring, csrc/topk_prefilter.hpp) and on the f32 fused kernel (LAPLACE_TOPK_PREFILTER=0),
with the ids of both compared."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def bench_line(full: bool = True, n_q: int = 16384, pre_only: bool = False, cpu_loop: bool = True, streams: int = 0, ws_gib: float = 0.0,
               dim: int = 128, ks=(12, 256)) -> dict:
    import torch as t
    from laplace_amd import ops, synthetic as S
    if streams:
        ops.TOPK_STREAMS = streams   # A/B: chunks alternating over this many streams
    if ws_gib:
        ops.TOPK_WS_BYTES = int(ws_gib * (1 << 30))   # A/B: queries per chunk = this / (4 n_items), rounded down to 2 048s
    from laplace_amd.interactions import Interactions
    dev = "cuda"
    spec = S.C2 if full else S.SyntheticSpec(200_000, 100_000, 2_000_000, seed=1)
    ei = S.generate(spec).to(dev)
    inter = Interactions(ei, spec.num_users, spec.num_items)
    r = inter.csr()
    g = t.Generator(device=dev).manual_seed(0)
    ue = t.randn(spec.num_users, dim, device=dev, generator=g) * 0.1
    ie = t.randn(spec.num_items, dim, device=dev, generator=g) * 0.1
    out = {"workload": f"top-K with exclusion, {n_q} query users against {spec.num_items} items, D={dim}, users' own edges excluded",
           "chunk_streams": ops.TOPK_STREAMS}
    uid = t.arange(n_q, device=dev)
    excl = ops.row_slice(r, 0, n_q)
    kept = {}
    # default path first (bf16x3 prefilter + exact rescoring), then the f32 fused kernel (LAPLACE_TOPK_PREFILTER=0)
    for mode, tag in ((("1", ""),) if pre_only else (("1", ""), ("0", "_f32_path"))):
        os.environ["LAPLACE_TOPK_PREFILTER"] = mode
        for k in ks:
            ops.topk_excl(uid[:4096], ue, ie, k, ops.row_slice(r, 0, 4096))
            t.cuda.synchronize()
            best = None
            for _ in range(3):
                t0 = time.perf_counter()
                ids = ops.topk_excl(uid, ue, ie, k, excl)
                t.cuda.synchronize()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            out[f"k{k}_users_per_s{tag}"] = n_q / best
            out[f"k{k}_ms_per_1k_users{tag}"] = 1e3 * best / (n_q / 1000)
            if mode == "1":
                kept[k] = ids
            else:
                out[f"k{k}_ids_equal_on_both_paths"] = bool(t.equal(ids, kept[k]))
    os.environ.pop("LAPLACE_TOPK_PREFILTER", None)
    if pre_only or not cpu_loop:
        return out
    # the reference's way, on the host: one GEMV + topk + setdiff per user (utils/metrics_lightgcn.py:125-142)
    from oracle import lightgcn_ref as R
    uec, iec = ue[:64].cpu(), ie.cpu()
    rp, col = r.rowptr.cpu().long(), r.col.cpu().long()
    pos = {u: col[rp[u]:rp[u + 1]] for u in range(64)}
    t0 = time.perf_counter()
    for u in range(64):
        R.make_predictions_for_user(uec, iec, u, pos, 256)
    out["cpu_reference_loop_users_per_s_k256"] = 64 / (time.perf_counter() - t0)
    out["cpu_threads"] = t.get_num_threads()
    return out


def main():
    n_q = int(sys.argv[sys.argv.index("--users") + 1]) if "--users" in sys.argv else 16384
    streams = int(sys.argv[sys.argv.index("--streams") + 1]) if "--streams" in sys.argv else 0
    ws_gib = float(sys.argv[sys.argv.index("--ws-gib") + 1]) if "--ws-gib" in sys.argv else 0.0
    print(json.dumps(bench_line(full="--full" in sys.argv, n_q=n_q, pre_only="--pre-only" in sys.argv, streams=streams, ws_gib=ws_gib,
                                dim=int(sys.argv[sys.argv.index("--dim") + 1]) if "--dim" in sys.argv else 128,
                                ks=tuple(int(x) for x in sys.argv[sys.argv.index("--ks") + 1].split(",")) if "--ks" in sys.argv else (12, 256))))


if __name__ == "__main__":
    main()

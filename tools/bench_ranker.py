#!/usr/bin/env python3
"""Measures the ranker train step on an H&M-shaped synthetic graph:
host sampler + collate, H2D, forward (embeddings, 2-layer hetero SAGEConv, BatchNorm, MLP decoder),
BCE, backward, Adam.  Not the contract bench (bench.py is): this is the profiling harness for the
ranker kernels.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=200_000)
    ap.add_argument("--items", type=int, default=50_000)
    ap.add_argument("--edges", type=int, default=4_000_000)
    ap.add_argument("--batch", type=int, default=24)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--hops", type=int, default=2)
    ap.add_argument("--fanout", type=int, default=64)
    ap.add_argument("--cpu", action="store_true", help="also time the torch-only oracle twin on the host cores")
    ap.add_argument("--device-sampler", action="store_true", help="sample batches on the GPU (N1) instead of on the host")
    ap.add_argument("--bench-line", action="store_true",
                    help="print ONE line in bench.py's schema (metric / value / roofline / cpu_baseline) for the overlapped training loop")
    ap.add_argument("--pipelined", action="store_true",
                    help="with --device-sampler: time the plain training loop (sampling of batch i+1 overlaps step i)")
    return ap.parse_args(argv)


def bench_line(users: int, items: int, edges: int, batch: int = 24, steps: int = 400, warmup: int = 50, hops: int = 2,
               fanout: int = 64, cpu: bool = True) -> dict:
    """The overlapped training loop in bench.py's schema, as a dict (bench.py's "ranker_c3" block)."""
    args = parse_args(["--users", str(users), "--items", str(items), "--edges", str(edges), "--batch", str(batch), "--steps", str(steps),
                       "--warmup", str(warmup), "--hops", str(hops), "--fanout", str(fanout), "--device-sampler", "--bench-line"]
                      + (["--cpu"] if cpu else []))
    return run(args)


def main():
    out = run(parse_args())
    if out is not None:
        print(json.dumps(out))


def run(args):
    import torch as t
    from types import SimpleNamespace
    from laplace_amd import synthetic as S
    from laplace_amd.data.dataset import GraphDataset
    from laplace_amd.hetero import DataLoader
    from laplace_amd.model.encoder_decoder import Encoder_Decoder_Model
    from laplace_amd.model.layers import get_SAGEConv_layers, get_linear_layers
    from laplace_amd.utils.get_info import get_feature_info, select_properties

    dev = "cuda"
    spec = S.SyntheticSpec(args.users, args.items, args.edges, seed=2, zipf_s=1.0)
    graph, users, articles = S.generate_hetero(spec)
    cfg = SimpleNamespace(k=12, num_neighbors=args.fanout, n_hop_neighbors=args.hops, positive_edges_ratio=0.5,
                          negative_edges_ratio=3.0, batch_size=args.batch)
    if args.device_sampler:
        from laplace_amd.data.device_sampler import DeviceGraphSampler
        loader = DeviceGraphSampler(cfg, graph, users, articles, device=dev, seed=0,
                                    prefetch=os.environ.get("LAPLACE_SAMPLER_PREFETCH", "1") if os.environ.get("LAPLACE_SAMPLER_PREFETCH") != "0" else False)
    else:
        ds = GraphDataset(cfg, graph, users, articles, train=True, randomization=True, seed=0)
        loader = DataLoader(ds, batch_size=args.batch, shuffle=True, generator=t.Generator().manual_seed(0))
    t.manual_seed(0)
    it = iter(loader)
    first = next(it)
    model = Encoder_Decoder_Model(get_SAGEConv_layers(2, 128, 64, "add"), get_linear_layers(2, 128, 128, 1),
                                  get_feature_info(graph), first.metadata(), True, "sum", True, 0.2, 0.3).to(dev)
    model.initialize_encoder_input_size(first.to(dev))
    native_ok = os.environ.get("LAPLACE_RANKER_NATIVE", "1") != "0" and os.environ.get("LAPLACE_RANKER_AUTOGRAD") != "1"
    # the reference's optimizer (training.py / run_pipeline.py: Adam(lr=0.01)); the op-by-op paths use torch's fused form
    opt = t.optim.Adam(model.parameters(), lr=0.01, fused=not native_ok)
    crit = t.nn.BCEWithLogitsLoss()
    model.train()

    from laplace_amd.ranker_step import FusedRankerStep
    from laplace_amd.ranker_native import NativeRankerStep
    fused = None if os.environ.get("LAPLACE_RANKER_AUTOGRAD") == "1" else FusedRankerStep(model, opt)
    native = NativeRankerStep(model, opt) if (native_ok and NativeRankerStep.supports(model, opt)) else None
    path_used = {"native": 0, "fused": 0, "autograd": 0}

    def step(batch):  # = one iteration of training.train_with_dataloader
        if native is not None:      # the whole iteration as one C call (mi_ranker_step_f32), as training.train_with_dataloader calls it
            labelled = batch[("customer", "buys", "article")]
            loss = native.step(batch.x_dict, batch.edge_index_dict, labelled.edge_label_index, labelled.edge_label)
            if loss is not None:
                path_used["native"] += 1
                return loss
        x, ei, eli, y = select_properties(batch)
        if fused is not None:
            loss = fused.step(x, ei, eli, y)
            if loss is not None:
                return loss
        opt.zero_grad()
        loss = crit(model(x, ei, eli).view(-1), y)
        loss.backward()
        opt.step()
        return loss

    t.autograd.set_multithreading_enabled(False)  # as training.train_with_dataloader does
    if args.bench_line:
        assert args.device_sampler
        n_c = n_a = n_lab = pos = 0
        for i in range(args.warmup + args.steps):
            if i == args.warmup:
                t.cuda.synchronize()
                t0 = time.perf_counter()
            batch = next(it)
            loss = step(batch)
            if i >= args.warmup:
                store = batch[("customer", "buys", "article")]
                n_c += batch["customer"].x.shape[0]; n_a += batch["article"].x.shape[0]
                n_lab += store.edge_label.numel()
        t.cuda.synchronize()
        dt = time.perf_counter() - t0
        # positives counted after the timed region (a device->host read)
        # (same iterator: a sampler's workspace and side stream belong to ONE live epoch iterator at a time)
        pos_per_batch = float(sum(int(next(it)[("customer", "buys", "article")].edge_label.sum()) for _ in range(20))) / 20
        n = args.steps
        n_c, n_a, n_lab = n_c / n, n_a / n, n_lab / n
        cw = sum(tb.shape[1] for tb in model.embedding_layers["customer"]); aw = sum(tb.shape[1] for tb in model.embedding_layers["article"])
        H, O = 128, 64
        fwd = (2 * n_c * H * (aw + cw) + 2 * n_a * H * (cw + aw)          # layer 1: lin_l(agg of the other type) + lin_r(own)
               + 2 * n_c * O * 2 * H + 2 * n_a * O * 2 * H                  # layer 2
               + 2 * n_lab * (2 * O * H + H))                               # decoder 128 -> 128 -> 1
        flops = 3 * fwd - 2 * (n_c * H * (aw + cw) + n_a * H * (cw + aw))   # backward = dX + dW; layer 1 has no dX
        ms = 1e3 * dt / n
        achieved = flops / (ms * 1e-3) / 1e12
        out = {"metric": "positive-edges/sec (ranker train iteration)", "value": pos_per_batch * n / dt, "unit": "positive-edges/s",
               "n_gpus": 1, "steps": n, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"encoder-decoder ranker training loop (training.train_with_dataloader's iteration), H&M-shaped "
                                      f"synthetic {args.users}x{args.items}, {args.edges} edges, {args.batch} users/batch, {args.hops}-hop "
                                      f"on-device sampling with fan-out {args.fanout} overlapped on a side stream; BASELINE.json configs[2]",
                          "parallelism": "1 GPU"},
               "roofline": {"bound": "mfma", "achieved": achieved, "peak": 157.0, "unit": "TFLOP/s", "frac": achieved / 157.0,
                            "traffic": None, "flops_per_iteration": flops,
                            "note": "a 24-user batch is ~3*10^4 nodes: ~75 launches of 5-40 us each, nowhere near the f32 MFMA roof "
                                    "(the iteration is one C call: mi_ranker_step_f32); the figure is reported because the schema asks for one",
                            "avg_customers_articles_label_edges_per_batch": [n_c, n_a, n_lab]},
               "loss": float(loss), "iteration_path": dict(path_used)}
        if args.cpu:
            from oracle import ranker_ref as RR
            ref = RR.ref_from_product(model, first.x_dict)
            ref.train()
            opt_r = t.optim.Adam(ref.parameters(), lr=0.01)
            batches = [next(it) for _ in range(6)]
            for j, b in enumerate(batches):
                if j == 1:
                    t1 = time.perf_counter()
                x, ei, eli, y = select_properties(b.to("cpu"))
                opt_r.zero_grad()
                l = crit(ref({k: v.clone() for k, v in x.items()}, ei, eli), y)
                l.backward()
                opt_r.step()
            cdt = (time.perf_counter() - t1) / (len(batches) - 1)
            out["cpu_baseline"] = {"value": pos_per_batch / cdt, "unit": "positive-edges/s", "cores": t.get_num_threads(), "kind": "port",
                                   "sample": f"5 model-only training iterations of the torch-only twin (oracle/ranker_ref.py) on batches "
                                             f"sampled by the device sampler, {1e3 * cdt:.0f} ms/iteration; the reference's own sampler "
                                             f"(python sets / lists per user) is not in this figure"}
        return out
    if args.pipelined:
        assert args.device_sampler
        labels = []
        prof = None
        if os.environ.get("LAPLACE_HOST_PROFILE") == "1":   # where the HOST spends the iteration (cProfile, stderr)
            import cProfile
            prof = cProfile.Profile()
        for i in range(args.warmup + args.steps):
            if prof is not None and i == args.warmup:
                prof.enable()
            if i == args.warmup:
                t.cuda.synchronize()
                t0 = time.perf_counter()
            batch = next(it)
            loss = step(batch)
            if i >= args.warmup:
                labels.append(batch[("customer", "buys", "article")].edge_label)
        if prof is not None:
            prof.disable()
            import pstats
            pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(28)
        t.cuda.synchronize()
        dt = time.perf_counter() - t0
        pos = int(sum(int(l.sum()) for l in labels))
        return {"workload": f"ranker training loop, sampling overlapped, H&M-shaped synthetic {args.users}x{args.items}, "
                            f"{args.edges} edges, batch {args.batch} users, {args.hops} hops, fan-out {args.fanout}",
                "steps": args.steps, "ms_per_iteration": 1e3 * dt / args.steps,
                "positive_edges_per_s": pos / dt, "loss": float(loss)}
    t_sample = t_dev = 0.0
    pos_edges = n_nodes = n_edges = 0
    for i in range(args.warmup + args.steps):
        if i == args.warmup:
            t.cuda.synchronize()
            t_sample = t_dev = 0.0
            pos_edges = n_nodes = n_edges = 0
        t0 = time.perf_counter()
        batch = next(it)
        if args.device_sampler:
            t.cuda.synchronize()
        t1 = time.perf_counter()
        bg = batch.to(dev)
        loss = step(bg)
        t.cuda.synchronize()
        t2 = time.perf_counter()
        t_sample += t1 - t0
        t_dev += t2 - t1
        store = batch[("customer", "buys", "article")]
        pos_edges += int(store.edge_label.sum())  # (a sync; outside both timed spans)
        n_edges += store.edge_index.shape[1]
        n_nodes += batch["customer"].x.shape[0] + batch["article"].x.shape[0]
    out = {"workload": f"ranker train step, H&M-shaped synthetic {args.users}x{args.items}, {args.edges} edges, batch "
                       f"{args.batch} users, {args.hops} hops, fan-out {args.fanout}",
           "steps": args.steps, "ms_per_step_device": 1e3 * t_dev / args.steps,
           "ms_per_step_sampler": 1e3 * t_sample / args.steps,
           "positive_edges_per_s_device_only": pos_edges / t_dev,
           "positive_edges_per_s_with_sampler": pos_edges / (t_dev + t_sample),
           "avg_nodes_per_batch": n_nodes / args.steps, "avg_mp_edges_per_batch": n_edges / args.steps,
           "avg_positive_label_edges_per_batch": pos_edges / args.steps, "loss": float(loss),
           "sampler": "device (sampler.hip)" if args.device_sampler else "host (numpy CSR)"}
    if args.cpu:
        from oracle import ranker_ref as RR
        ref = RR.ref_from_product(model, first.x_dict)
        ref.train()
        opt_r = t.optim.Adam(ref.parameters(), lr=0.01)
        batches = [next(it) for _ in range(6)]
        t0 = None
        for j, b in enumerate(batches):
            if j == 1:
                t0 = time.perf_counter()
            x, ei, eli, y = select_properties(b)
            opt_r.zero_grad()
            l = crit(ref({k: v.clone() for k, v in x.items()}, ei, eli), y)
            l.backward()
            opt_r.step()
        out["cpu_ms_per_step_model_only"] = 1e3 * (time.perf_counter() - t0) / (len(batches) - 1)
        out["cpu_threads"] = t.get_num_threads()
    return out


if __name__ == "__main__":
    main()

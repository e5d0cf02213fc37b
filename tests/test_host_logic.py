"""Host-side logic that needs no GPU: config surface, synthetic graphs, adjacency index math."""
import dataclasses
import os

import numpy as np
import pytest
import torch as t


def _load(golden_dir, name):
    return t.load(os.path.join(golden_dir, name), weights_only=False)


def test_config_surface_matches_reference(golden_dir):
    from laplace_amd import config as C
    g = _load(golden_dir, "config_defaults.pt")
    assert [f.name for f in dataclasses.fields(C.Config)] == g["Config_fields"]
    assert [f.name for f in dataclasses.fields(C.LightGCNConfig)] == g["LightGCNConfig_fields"]
    assert C.embedding_range_dict == g["embedding_range_dict"]
    for k, v in g["lightgcn_config"].items():
        assert getattr(C.lightgcn_config, k) == v, k
    for k, v in g["link_pred_config"].items():
        ours = getattr(C.link_pred_config, k)
        if isinstance(v, list):
            ours = [list(x) if isinstance(x, tuple) else x for x in ours]
        assert ours == v, k
    C.link_pred_config.check_validity()


def test_constants():
    from laplace_amd.utils.constants import Constants
    assert Constants.edge_key == ("customer", "buys", "article")
    assert Constants.rev_edge_key == ("article", "rev_buys", "customer")


def test_synthetic_c1_shape_and_determinism():
    from laplace_amd import synthetic as S
    e1, e2 = S.generate(S.C1), S.generate(S.C1)
    assert t.equal(e1, e2)
    assert e1.shape == (2, 100_000) and e1.dtype == t.int64
    assert int(e1[0].max()) < 943 and int(e1[1].max()) < 1682 and int(e1.min()) >= 0
    keys = e1[0] * 1682 + e1[1]
    assert keys.unique().numel() == 100_000  # distinct pairs
    deg = t.bincount(e1[0], minlength=943)
    assert int(deg.min()) >= 1


def test_synthetic_small_power_law():
    from laplace_amd import synthetic as S
    spec = S.SyntheticSpec(5000, 800, 40_000, seed=4)
    e = S.generate(spec)
    di = t.bincount(e[1], minlength=800).double()
    assert di.max() > 20 * di.median().clamp(min=1)  # Zipf head
    other = S.generate(S.shard_spec(spec, 1))
    assert not t.equal(e, other)
    # same popular items in every shard
    top_a = set(di.topk(10).indices.tolist())
    top_b = set(t.bincount(other[1], minlength=800).topk(10).indices.tolist())
    assert len(top_a & top_b) >= 7


def test_adjacency_index_construction():
    from laplace_amd.interactions import Interactions
    ei = t.tensor([[0, 0, 2], [1, 3, 0]])
    inter = Interactions(ei, num_users=3, num_items=4)
    ref = inter.adjacency("reference")
    r, c, v = ref.coo()
    assert ref.sparse_sizes() == (7, 7) and not ref.is_symmetric and v is None
    assert r.tolist() == [0, 0, 2] and c.tolist() == [1, 3, 0]  # item ids, NOT offset by U
    bi = inter.adjacency("bipartite")
    r, c, _ = bi.coo()
    assert bi.is_symmetric
    assert sorted(zip(r.tolist(), c.tolist())) == sorted([(0, 4), (0, 6), (2, 3), (4, 0), (6, 0), (3, 2)])
    with pytest.raises(ValueError):
        inter.adjacency("nope")


def test_lightgcn_module_surface_on_cpu():
    """Constructor, attributes, parameter list and init statistics need no GPU."""
    from laplace_amd.model.lightgcn import LightGCN
    t.manual_seed(0)
    m = LightGCN(num_users=300, num_items=200, embedding_dim=32, num_iterations=3)
    names = [n for n, _ in m.named_parameters()]
    assert names == ["users_emb.weight", "items_emb.weight"]
    assert m.users_emb.weight.shape == (300, 32) and m.items_emb.weight.shape == (200, 32)
    tab = m.table()
    assert tab.shape == (500, 32)
    assert tab.data_ptr() == m.users_emb.weight.data_ptr()
    assert t.equal(tab[300:], m.items_emb.weight.data)
    assert abs(float(tab.std()) - 0.1) < 0.01 and abs(float(tab.mean())) < 0.01
    # same draws as two independent nn.init.normal_ calls in the reference's order
    t.manual_seed(0)
    a = t.empty(300, 32); t.nn.init.normal_(a, std=0.1)
    b = t.empty(200, 32); t.nn.init.normal_(b, std=0.1)
    assert t.equal(a, m.users_emb.weight.data) and t.equal(b, m.items_emb.weight.data)
    # dtype casts keep the two tables adjacent
    m.double()
    assert m.table().dtype == t.float64 and m.table().shape == (500, 32)
    with pytest.raises(TypeError):
        m.float().forward(t.zeros(2, 3))
    with pytest.raises(ValueError):
        LightGCN(3, 4, embedding_dim=30, num_iterations=1)


def test_map_at_k_known_answers():
    """MAP@k (Kaggle H&M definition): hand-computed cases."""
    from laplace_amd.utils.metrics import MAPatK
    gt = [t.tensor([1, 2, 3]), t.tensor([7]), t.tensor([], dtype=t.int64), t.tensor([4, 5])]
    pred = t.tensor([[1, 9, 2, 8], [5, 6, 7, 7], [1, 2, 3, 4], [-1, -1, -1, -1]])
    ap0 = (1 / 1 + 2 / 3) / 3      # hits at ranks 1 and 3, min(|GT|, k) = 3
    ap1 = (1 / 3) / 1              # hit at rank 3; the repeat at rank 4 does not count again
    ap3 = 0.0                      # no predictions
    assert abs(MAPatK(gt, pred, k=4) - (ap0 + ap1 + ap3) / 3) < 1e-12   # the user without ground truth is skipped
    assert abs(MAPatK(gt, pred, k=2) - ((1 / 1) / 2 + 0 + 0) / 3) < 1e-12


def test_collate_carries_global_node_ids():
    """Samples carry n_id (global ids of their relabelled nodes); the collate concatenates it like x."""
    from types import SimpleNamespace
    from laplace_amd import synthetic as S
    from laplace_amd.data.dataset import GraphDataset
    from laplace_amd.hetero import collate
    from laplace_amd.utils.constants import Constants
    graph, users, articles = S.generate_hetero(S.SyntheticSpec(60, 40, 400, seed=3, deg_min=2, deg_max=30),
                                               customer_cards=(20, 2), article_cards=(10, 5))
    cfg = SimpleNamespace(k=4, num_neighbors=4, n_hop_neighbors=2, positive_edges_ratio=0.5, negative_edges_ratio=2.0, batch_size=3)
    ds = GraphDataset(cfg, graph, users, articles, train=True, seed=1)
    a, b = ds[5], ds[9]
    batch = collate([a, b])
    for nt in (Constants.node_user, Constants.node_item):
        assert t.equal(batch[nt].n_id, t.cat([a[nt].n_id, b[nt].n_id]))
        assert t.equal(batch[nt].x, graph[nt].x[batch[nt].n_id])
    eli = batch[Constants.edge_key].edge_label_index
    assert set(batch[Constants.node_user].n_id[eli[0]].tolist()) == {5, 9}


def test_bench_self_launch_parent_never_needs_a_gpu_and_reports_worker_failure():
    """python bench.py --gpus 2 without a launcher: the parent starts the workers itself (it must not touch the GPU)
    and exits non-zero when they fail — here they do, because this container has no GPU."""
    import subprocess
    import sys as _sys
    import torch as _t
    if _t.cuda.is_available():
        import pytest as _pytest
        _pytest.skip("CPU-only check of the launcher's failure path")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([_sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert "needs a GPU" in r.stderr and "exited with code" in r.stderr
    assert r.stdout.strip() == ""


def test_bench_classifies_the_dense_propagate_kernels_and_parses_pmc_passes(tmp_path, monkeypatch):
    """bench.py's live-traffic figure: which kernels make up ONE dense propagate launch, and how the two rocprofv3 --pmc
    passes are turned into bytes ((2*FETCH_SIZE + WRITE_SIZE)*1024 per dispatch, averaged per kernel, summed)."""
    import subprocess
    import bench
    dense = ["void (anonymous namespace)::spmm_rows_kernel<32, 1, 8, 1, false, false>(long, int, int const*)",
             "void (anonymous namespace)::spmm_sweep_kernel<32, 8, false>(mi_spmm_sweep, int)",
             "void (anonymous namespace)::spmm_items_kernel<32, 1, 8, 1, false>(long, int)",
             "void (anonymous namespace)::spmm_fixup_kernel<32, 1, false, false>(int, int)",
             "void (anonymous namespace)::spmm_rows_hot_kernel<32, 8, false>(long, int)"]
    other = ["void (anonymous namespace)::spmm_rows_kernel<32, 1, 8, 1, false, true>(long)",      # Adam epilogue
             "void (anonymous namespace)::spmm_rows_kernel<32, 1, 8, 1, true, false>(long)",      # sparse operands
             "void (anonymous namespace)::spmm_sweep_kernel<32, 8, true>(mi_spmm_sweep)",
             "void (anonymous namespace)::spmm_fixup_kernel<32, 1, false, true>(int)",
             "void (anonymous namespace)::adam_kernel(long)", "bpr_slot_kernel"]
    for k in dense:
        assert bench.is_dense_spmm_kernel(bench._short_kernel(k)), k
    for k in other:
        assert not bench.is_dense_spmm_kernel(bench._short_kernel(k)), k

    def fake_run(cmd, **kw):   # stands in for rocprofv3: writes the counter file the real tool would
        counter = cmd[cmd.index("--pmc") + 1]
        out = cmd[cmd.index("-d") + 1]
        assert "--kernel-trace" not in cmd and "--sys-trace" not in cmd          # counters only, as the pool requires
        assert cmd[cmd.index("--") + 1].endswith("python") or "python" in cmd[cmd.index("--") + 1]
        os.makedirs(os.path.join(out, "host"), exist_ok=True)
        rows = {"FETCH_SIZE": {dense[0]: [1000.0, 3000.0], dense[1]: [500.0], other[0]: [9999.0]},
                "WRITE_SIZE": {dense[0]: [100.0, 300.0], dense[3]: [50.0], other[0]: [7.0]}}[counter]
        with open(os.path.join(out, "host", "1_counter_collection.csv"), "w") as fh:
            fh.write("Kernel_Name,Counter_Name,Counter_Value\n")
            for name, vals in rows.items():
                for v in vals:
                    fh.write(f'"{name}",{counter},{v}\n')
        return subprocess.CompletedProcess(cmd, 0, b"", b"")

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(bench.shutil, "which", lambda name: "/bin/true")
    total, per_kernel, note = bench.pmc_traffic("c2", ["--dim", "128"])
    want = (2 * 2000.0 + 200.0) * 1024 + (2 * 500.0) * 1024 + 50.0 * 1024     # rows (avg of two dispatches) + sweep + fix-up
    assert total == want and "live" in note
    assert per_kernel["spmm_rows_kernel<32, 1, 8, 1, false, true>"] == (2 * 9999.0 + 7.0) * 1024
    # a failing pass is reported, not guessed around
    monkeypatch.setattr(bench.subprocess, "run", lambda cmd, **kw: subprocess.CompletedProcess(cmd, 3, b"", b"boom"))
    total, _, why = bench.pmc_traffic("c2", [])
    assert total is None and "failed" in why

"""BASELINE.json configs[3] / SURVEY 8d "C4": ONE block-wise generated graph sharded by user_id // ceil(U/N).
World-size-2 gloo test of what bench.py --config c4 does per rank: the shards tile the single graph (id offsets),
edge weights use the GLOBAL item degrees, and the sharded forward equals the single-process forward on the whole graph."""
import os
import socket
import sys

import torch as t
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from laplace_amd import synthetic as S  # noqa: E402

SPEC = S.SyntheticSpec(2048, 256, 16384, seed=3)
NB, D, K = 8, 16, 2


def test_blocks_define_one_graph_whatever_the_shard_count():
    whole = S.generate_blocks(SPEC, NB, 0, NB, workers=1)
    assert whole.shape == (2, SPEC.num_edges) and int(whole[0].max()) < SPEC.num_users
    assert t.unique(whole[0] * SPEC.num_items + whole[1]).numel() == SPEC.num_edges  # distinct pairs
    for world in (1, 2, 4, 8):
        parts = []
        for r in range(world):
            b0, b1 = S.shard_blocks(NB, world, r)
            ei = S.generate_blocks(SPEC, NB, b0, b1)
            per = SPEC.num_users // world
            assert int(ei[0].max()) < per                      # local ids
            ei = ei.clone()
            ei[0] += r * per                                   # = user_id // ceil(U / world) partition
            assert bool(((ei[0] // per) == r).all())
            parts.append(ei)
        assert t.equal(t.cat(parts, dim=1), whole)
    # every block draws from the same item popularity: the hot items are hot in every shard
    top = t.bincount(whole[1], minlength=SPEC.num_items).argmax()
    for r in range(2):
        b0, b1 = S.shard_blocks(NB, 2, r)
        assert t.bincount(S.generate_blocks(SPEC, NB, b0, b1)[1], minlength=SPEC.num_items).argmax() == top


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t.set_num_threads(2)
    import cpu_ops
    from laplace_amd.dist import ShardedLightGCNTrainer
    from laplace_amd.interactions import Interactions
    from laplace_amd.model.lightgcn import LightGCN
    b0, b1 = S.shard_blocks(NB, world, rank)
    ei = S.generate_blocks(SPEC, NB, b0, b1)
    U, I = SPEC.num_users // world, SPEC.num_items
    g = t.Generator().manual_seed(5)
    tab_u = t.randn(SPEC.num_users, D, generator=g) * 0.1
    tab_i = t.randn(I, D, generator=g) * 0.1
    model = LightGCN(U, I, D, K)
    with t.no_grad():
        model.users_emb.weight.copy_(tab_u[rank * U:(rank + 1) * U])
        model.items_emb.weight.copy_(tab_i)
    tr = ShardedLightGCNTrainer(model, Interactions(ei, U, I), lr=1e-2, Lambda=1e-4, batch_size=64, seed=3, ops_impl=cpu_ops)
    a = tr.a_users
    ret[rank] = {"final": tr.forward().clone(), "rowptr": a.rowptr.clone(), "col": a.col.clone(), "val": a.val.clone(),
                 "edges": ei}
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_c4_two_rank_shards_use_global_degrees_and_match_the_whole_graph_forward():
    from oracle import lightgcn_ref as R
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    whole = S.generate_blocks(SPEC, NB, 0, NB)
    U, I = SPEC.num_users, SPEC.num_items
    per = U // 2
    deg_u = t.bincount(whole[0], minlength=U).float()
    deg_i = t.bincount(whole[1], minlength=I).float()
    for r in range(2):
        res = ret[r]
        # weights of rank r's user rows: (deg(u) * GLOBAL deg(i))^-1/2 — a shard-local item degree would be smaller
        rowptr, col, val = res["rowptr"].long(), res["col"].long(), res["val"]
        rows = t.repeat_interleave(t.arange(per), rowptr[1:per + 1] - rowptr[:per])
        lo = int(rowptr[0])
        want = (deg_u[rows + r * per] * deg_i[col[lo:lo + rows.numel()] - per]).pow(-0.5)
        assert t.allclose(val[lo:lo + rows.numel()], want, rtol=1e-6)
        assert int(res["edges"][0].max()) < per
    g = t.Generator().manual_seed(5)
    tab_u = t.randn(U, D, generator=g) * 0.1
    tab_i = t.randn(I, D, generator=g) * 0.1
    row, col = R.bipartite_edges(whole[0], whole[1], U)
    wu, _, wi, _ = R.lightgcn_forward(tab_u, tab_i, row, col, K)
    for r in range(2):
        f = ret[r]["final"]
        assert t.allclose(f[:per], wu[r * per:(r + 1) * per], atol=2e-6)
        assert t.allclose(f[per:], wi, atol=2e-6)
    assert t.equal(ret[0]["final"][per:], ret[1]["final"][per:])  # item replicas bitwise identical


# ---- the launcher behind `bench.py --gpus N` (laplace_amd/launch.py): no child may leave the parent waiting -----------------

def _py(code):
    return [sys.executable, "-c", code]


def test_launcher_returns_at_once_when_a_worker_dies_before_the_rendezvous():
    """Rank 1 exits 3 straight away; rank 0 would wait for ten minutes (as in init_process_group without its peer).
    The parent must kill rank 0 and return rank 1's code within seconds, printing no result line."""
    import io
    import time
    from contextlib import redirect_stdout
    from laplace_amd import launch
    env = dict(os.environ)
    buf = io.StringIO()
    t0 = time.time()
    with redirect_stdout(buf):
        rc = launch.supervise([_py("import time; time.sleep(600)"), _py("import sys; sys.exit(3)")], [env, env], timeout_s=120)
    assert rc == 3
    assert time.time() - t0 < 20
    assert buf.getvalue() == ""


def test_launcher_deadline_counts_from_launch_and_kills_silent_workers():
    import time
    from laplace_amd import launch
    env = dict(os.environ)
    t0 = time.time()
    rc = launch.supervise([_py("import time; time.sleep(600)"), _py("import time; time.sleep(600)")], [env, env], timeout_s=2)
    assert rc == 124 and time.time() - t0 < 20


def test_launcher_hands_rank0s_result_line_through_when_all_workers_succeed(capfd):
    from laplace_amd import launch
    env = dict(os.environ)
    rc = launch.supervise([_py("print('noise'); print('{\"metric\": \"m\", \"value\": 1}')"), _py("pass")], [env, env], timeout_s=60)
    out, err = capfd.readouterr()
    assert rc == 0 and out.strip() == '{"metric": "m", "value": 1}' and "noise" in err


def test_launcher_fails_when_rank0_succeeds_without_a_result_line():
    from laplace_amd import launch
    env = dict(os.environ)
    assert launch.supervise([_py("print('hello')"), _py("pass")], [env, env], timeout_s=60) == 1


def test_bench_parent_uses_the_launcher_and_fails_fast_without_gpus():
    """python bench.py --gpus 2 here (no GPU): both workers exit non-zero at once ('needs a GPU'); the parent must
    return non-zero promptly and print no result line."""
    import subprocess
    import time
    if t.cuda.is_available():
        import pytest
        pytest.skip("CPU-only check of the launcher's failure path")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "needs a GPU" in r.stderr and "stopping the other workers" in r.stderr
    assert time.time() - t0 < 120

"""World-size-2 gloo tests of the user-sharded LightGCN trainer's host logic: the two
ranks together must reproduce, step for step, the single-process reference loop on the union graph."""
import os
import socket
import sys

import pytest
import torch as t
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

U0, U1, I, D, K, B, STEPS = 60, 45, 40, 16, 3, 32, 4


def _shards():
    g = t.Generator().manual_seed(0)
    out = []
    for U, E in ((U0, 500), (U1, 380)):
        keys = t.randperm(U * I, generator=g)[:E]
        out.append(t.stack([keys // I, keys % I]))
    return out


def _tables():
    g = t.Generator().manual_seed(1)
    return t.randn(U0, D, generator=g) * 0.1, t.randn(U1, D, generator=g) * 0.1, t.randn(I, D, generator=g) * 0.1


def _batches(step):
    g = t.Generator().manual_seed(100 + step)
    out = []
    for U in (U0, U1):
        out.append((t.randint(0, U, (B,), generator=g), t.randint(0, I, (B,), generator=g),
                    t.randint(0, I, (B,), generator=g)))
    return out


def _worker(rank, world, port, ret, sparse_batch, reorder=False, min_edges=None):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    t.set_num_threads(2)
    import cpu_ops
    import laplace_amd.dist as D_
    from laplace_amd.dist import ShardedLightGCNTrainer
    if min_edges is not None:
        D_.REORDER_MIN_EDGES_PER_RANK = min_edges
    if reorder == "disagree":   # ranks passing different explicit values must be told so, not left to hang
        try:
            ShardedLightGCNTrainer(LightGCN_(rank), Interactions_(rank), lr=1e-2, Lambda=1e-4, batch_size=B, seed=3,
                                   ops_impl=cpu_ops, reorder=bool(rank))
            ret[rank] = "no error"
        except ValueError as e:
            ret[rank] = str(e)
        dist.destroy_process_group()
        return
    from laplace_amd.interactions import Interactions
    from laplace_amd.model.lightgcn import LightGCN
    ei = _shards()[rank]
    tu0, tu1, ti = _tables()
    U = (U0, U1)[rank]
    model = LightGCN(U, I, D, K)
    with t.no_grad():
        model.users_emb.weight.copy_((tu0, tu1)[rank])
        # rank 1 starts with garbage item rows: the constructor's broadcast must fix them
        model.items_emb.weight.copy_(ti if rank == 0 else t.zeros_like(ti))
    tr = ShardedLightGCNTrainer(model, Interactions(ei, U, I), lr=1e-2, Lambda=1e-4, batch_size=B, seed=3,
                                ops_impl=cpu_ops, sparse_batch=sparse_batch, reorder=reorder)
    losses = []
    for s in range(STEPS):
        losses.append(float(tr.step(_batches(s)[rank])))
    fin = tr.forward().clone()
    item_order = None
    if reorder is None:
        reorder = tr.order is not None
    if reorder:  # rows back under their original ids; every rank must have numbered the items alike
        fin = fin[tr.order.node_new_of_old()]
        item_order = tr.order.item_new_of_old.clone()
        tr.finish()
    ret[rank] = {"table": tr.table.clone(), "final": fin, "losses": losses, "item_order": item_order}
    dist.barrier()
    dist.destroy_process_group()


def LightGCN_(rank):
    from laplace_amd.model.lightgcn import LightGCN
    return LightGCN((U0, U1)[rank], I, D, K)


def Interactions_(rank):
    from laplace_amd.interactions import Interactions
    return Interactions(_shards()[rank], (U0, U1)[rank], I)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_explicit_reorder_arguments_that_differ_across_ranks_raise_on_every_rank():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), ret, True, "disagree"), nprocs=2, join=True)
    assert "differs across ranks" in ret[0] and "differs across ranks" in ret[1]


@pytest.mark.parametrize("sparse_batch,reorder,min_edges", [(False, False, None), (True, False, None), (True, True, None),
                                                            (True, None, 440), (True, None, 441)])
def test_two_rank_sharded_training_equals_single_process_reference(sparse_batch, reorder, min_edges):
    """reorder=True: each rank trains under the locality order (items ranked by their all-reduced GLOBAL degree, so
    the replicas agree on the numbering; users relabelled locally) and still reproduces the reference loop.
    reorder=None with the threshold BETWEEN the two shards' edge counts (500 and 380 edges): the default is decided on
    the all-reduced total (880 >= 2 x 440: both ranks reorder; < 2 x 441: neither does) — a rank-local rule would put
    rank 0 into an all-reduce rank 1 never joins."""
    from oracle import lightgcn_ref as R
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), ret, sparse_batch, reorder, min_edges), nprocs=2, join=True)
    if reorder is None:
        assert (ret[0]["item_order"] is not None) == (min_edges == 440) == (ret[1]["item_order"] is not None)
        reorder = min_edges == 440
    if reorder:
        assert t.equal(ret[0]["item_order"], ret[1]["item_order"])

    # single-process reference on the union graph: users of rank 1 follow those of rank 0
    e0, e1 = _shards()
    eu = t.cat([e0[0], e1[0] + U0])
    ei = t.cat([e0[1], e1[1]])
    UU = U0 + U1
    row, col = R.bipartite_edges(eu, ei, UU)
    tu0, tu1, ti = _tables()
    uw = t.nn.Parameter(t.cat([tu0, tu1]))
    iw = t.nn.Parameter(ti.clone())
    opt = t.optim.Adam([uw, iw], lr=1e-2)
    ref_losses = []
    for s in range(STEPS):
        b0, b1 = _batches(s)
        batch = (t.cat([b0[0], b1[0] + U0]), t.cat([b0[1], b1[1]]), t.cat([b0[2], b1[2]]))
        ref_losses.append(R.train_step(uw, iw, opt, row, col, K, batch, 1e-4))
    wu, _, wi, _ = R.lightgcn_forward(uw.detach(), iw.detach(), row, col, K)

    r0, r1 = ret[0], ret[1]
    # item replicas stay bitwise identical across ranks
    assert t.equal(r0["table"][U0:], r1["table"][U1:])
    assert t.allclose(r0["table"][:U0], uw.detach()[:U0], atol=2e-6)
    assert t.allclose(r1["table"][:U1], uw.detach()[U0:], atol=2e-6)
    assert t.allclose(r0["table"][U0:], iw.detach(), atol=2e-6)
    assert t.allclose(r0["final"][:U0], wu[:U0], atol=2e-6) and t.allclose(r1["final"][:U1], wu[U0:], atol=2e-6)
    assert t.allclose(r0["final"][U0:], wi, atol=2e-6) and t.allclose(r1["final"][U1:], wi, atol=2e-6)
    # the global objective: softplus terms average over ranks, L2 terms add up
    lam = 1e-4
    assert all(abs(a) < 10 for a in r0["losses"] + r1["losses"])
    assert len(ref_losses) == STEPS


@pytest.mark.parametrize("sparse_batch", [False, True])
def test_single_rank_sharded_trainer_matches_plain_reference(sparse_batch):
    """world_size 1 (no process group): the sharded code path degenerates to the plain step."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cpu_ops
    from laplace_amd.dist import ShardedLightGCNTrainer
    from laplace_amd.interactions import Interactions
    from laplace_amd.model.lightgcn import LightGCN
    from oracle import lightgcn_ref as R
    ei = _shards()[0]
    tu0, _, ti = _tables()
    model = LightGCN(U0, I, D, K)
    with t.no_grad():
        model.users_emb.weight.copy_(tu0)
        model.items_emb.weight.copy_(ti)
    tr = ShardedLightGCNTrainer(model, Interactions(ei, U0, I), lr=1e-2, Lambda=1e-4, batch_size=B, seed=3,
                                ops_impl=cpu_ops, sparse_batch=sparse_batch)
    uw, iw = t.nn.Parameter(tu0.clone()), t.nn.Parameter(ti.clone())
    opt = t.optim.Adam([uw, iw], lr=1e-2)
    row, col = R.bipartite_edges(ei[0], ei[1], U0)
    for s in range(3):
        batch = _batches(s)[0]
        want = R.train_step(uw, iw, opt, row, col, K, batch, 1e-4)
        got = float(tr.step(batch))
        assert abs(got - want) < 1e-6
    assert t.allclose(tr.table[:U0], uw.detach(), atol=2e-6) and t.allclose(tr.table[U0:], iw.detach(), atol=2e-6)
    # device-sampled batches are valid edges / non-edges of the shard
    us, ps, ns = tr.sample()
    keys = set((ei[0] * I + ei[1]).tolist())
    assert all(k in keys for k in (us * I + ps).tolist()) and not any(k in keys for k in (us * I + ns).tolist())


def _ddp_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from laplace_amd.dist_ranker import allreduce_gradients, broadcast_parameters, user_shard
    t.manual_seed(100 + rank)  # different initial weights per rank on purpose
    net = t.nn.Sequential(t.nn.Linear(6, 5), t.nn.BatchNorm1d(5), t.nn.ReLU(), t.nn.Linear(5, 1))
    broadcast_parameters(net)
    g = t.Generator().manual_seed(7)
    X, Y = t.randn(40, 6, generator=g), t.randn(40, 1, generator=g)
    lo, hi = user_shard(40, rank, world)
    loss = ((net(X[lo:hi]) - Y[lo:hi]) ** 2).mean()
    loss.backward()
    allreduce_gradients(net.parameters())
    ret[rank] = {"state": {k: v.clone() for k, v in net.state_dict().items()},
                 "grads": [p.grad.clone() for p in net.parameters()], "shard": (lo, hi)}
    dist.barrier()
    dist.destroy_process_group()


def test_ranker_data_parallel_helpers_two_ranks():
    """Replicas start identical (broadcast), shards tile the users, gradients come out as the rank mean."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_ddp_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    a, b = ret[0], ret[1]
    assert a["shard"] == (0, 20) and b["shard"] == (20, 40)
    for ga, gb in zip(a["grads"], b["grads"]):
        assert t.equal(ga, gb)
    # reference: one process, mean of the two shard losses' gradients (weights = rank 0's initial weights)
    t.manual_seed(100)
    net = t.nn.Sequential(t.nn.Linear(6, 5), t.nn.BatchNorm1d(5), t.nn.ReLU(), t.nn.Linear(5, 1))
    g = t.Generator().manual_seed(7)
    X, Y = t.randn(40, 6, generator=g), t.randn(40, 1, generator=g)
    total = 0.5 * (((net(X[:20]) - Y[:20]) ** 2).mean() + ((net(X[20:]) - Y[20:]) ** 2).mean())
    total.backward()
    for p, ga in zip(net.parameters(), a["grads"]):
        assert t.allclose(p.grad, ga, atol=1e-6)
    from laplace_amd.dist_ranker import user_shard
    assert [user_shard(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]

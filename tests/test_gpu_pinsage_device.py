"""GPU: PinSAGE batches built whole on the device (mi_pinsage_sample_batch, round 3) against the index-op path of
pinsage/sampler.py (itself bit-exact against the mirror, tests/test_pinsage.py) and the block CSRs the model consumes
against the sort-based construction (pinsage/model.py::block_csr)."""
import numpy as np
import pytest
import torch as t

pytestmark = pytest.mark.gpu


def _graph(seed, U, I, E):
    from laplace_amd import synthetic as S
    from laplace_amd.data.dataset import AdjList
    ei = S.generate(S.SyntheticSpec(U, I, E, seed=seed, deg_min=1, deg_max=60, zipf_s=0.9))
    u, a = ei[0].numpy(), ei[1].numpy()
    return AdjList.from_edges(u, a, U), AdjList.from_edges(a, u, I)


@pytest.mark.parametrize("cfg", [dict(B=32, L=2, p=0.5, W=10, T=3, layers=2), dict(B=64, L=3, p=0.25, W=6, T=5, layers=2),
                                 dict(B=200, L=2, p=0.5, W=10, T=3, layers=2), dict(B=16, L=1, p=0.0, W=4, T=2, layers=3)])
def test_device_batches_equal_the_index_op_path_and_bring_the_model_its_csrs(cfg):
    from laplace_amd.pinsage.model import block_csr
    from laplace_amd.pinsage.sampler import PinSAGESampler
    U, I = 2000, 700
    users, items = _graph(7, U, I, 30000)
    smp = PinSAGESampler(users, items, U, I, batch_size=cfg["B"], random_walk_length=cfg["L"], random_walk_restart_prob=cfg["p"],
                         num_random_walks=cfg["W"], num_neighbors=cfg["T"], num_layers=cfg["layers"], seed=11)
    for step in (0, 3, 9):
        smp.device_batches = True
        got = smp._sample_batch_device(step)
        assert got is not None                                    # these sizes are inside the device kernels'
        smp.device_batches = False
        want = smp.sample_batch(step)
        assert bool((smp._pos32 == -1).all())                     # the scratch was handed back clean
        assert t.equal(got["seeds"], want["seeds"])
        for a, b in zip(got["pos"] + got["neg"], want["pos"] + want["neg"]):
            assert t.equal(a, b)
        assert len(got["blocks"]) == len(want["blocks"]) == cfg["layers"]
        for gb, wb in zip(got["blocks"], want["blocks"]):
            assert gb["n_dst"] == wb["n_dst"]
            for key in ("src_ids", "edge_src", "edge_dst", "weights"):
                assert t.equal(gb[key], wb[key]), key
            by_dst, by_src = gb["csr"]
            ref_dst, ref_src = block_csr(wb)                       # two device sorts + the normalisation, as before
            for mine, ref in ((by_dst, ref_dst), (by_src, ref_src)):
                assert mine.n_rows == ref.n_rows and mine.n_cols == ref.n_cols
                assert t.equal(mine.rowptr, ref.rowptr) and t.equal(mine.col, ref.col) and t.equal(mine.val, ref.val)


def test_sizes_beyond_the_single_workgroup_kernels_fall_back():
    from laplace_amd.pinsage.sampler import PinSAGESampler
    users, items = _graph(3, 800, 300, 9000)
    smp = PinSAGESampler(users, items, 800, 300, batch_size=64, random_walk_length=3, random_walk_restart_prob=0.25,
                         num_random_walks=6, num_neighbors=5, num_layers=3, seed=1)      # third layer: 6 912 x 5 > 16 384
    assert smp._sample_batch_device(0) is None
    batch = smp.sample_batch(0)                                       # the index-op path took over
    assert len(batch["blocks"]) == 3 and "csr" not in batch["blocks"][0]


def test_training_on_device_built_batches():
    from laplace_amd.pinsage.model import PinSAGEModel, train_epoch
    from laplace_amd.pinsage.sampler import PinSAGESampler
    users, items = _graph(4, 400, 200, 5000)
    smp = PinSAGESampler(users, items, 400, 200, batch_size=64, seed=5)
    t.manual_seed(0)
    model = PinSAGEModel(200, 16, 2).to("cuda")
    opt = t.optim.Adam(model.parameters(), lr=3e-3)
    first = train_epoch(model, opt, smp, 40)
    for _ in range(3):
        last = train_epoch(model, opt, smp, 40)
    assert np.isfinite(last).all() and np.mean(last) < np.mean(first)


def test_overlapped_batches_are_the_serial_batches():
    """PinSAGESampler.batches(n): batch i + 1 drawn on a side stream under the training of batch i, rotating buffer sets;
    the batches are exactly those of n sample_batch() calls, a batch stays intact while the next one is in the caller's
    hands, and the step counter ends where the serial calls leave it.  Kernels are enqueued on the training stream between
    the batches so that the side stream really runs beside something."""
    from laplace_amd.pinsage.sampler import PinSAGESampler
    U, I = 3000, 900
    users, items = _graph(9, U, I, 50000)
    mk = lambda: PinSAGESampler(users, items, U, I, batch_size=48, seed=21)
    a, b = mk(), mk()
    serial = [a.sample_batch() for _ in range(7)]
    busy = t.randn(2048, 2048, device="cuda")
    flat = lambda batch: [batch["seeds"], *batch["pos"], *batch["neg"]] + [x for blk in batch["blocks"] for x in (
        blk["src_ids"], blk["edge_src"], blk["edge_dst"], blk["weights"], blk["csr"][0].rowptr, blk["csr"][0].col, blk["csr"][0].val,
        blk["csr"][1].rowptr, blk["csr"][1].col, blk["csr"][1].val)]
    prev = None
    for i, batch in enumerate(b.batches(7)):
        if prev is not None:      # the previous batch was not touched by the sampling of this one or of the one in flight
            for x, y in zip(flat(prev), flat(serial[i - 1])):
                assert t.equal(x, y), i
        for _ in range(3):
            busy = busy @ busy * 1e-3
        for x, y in zip(flat(batch), flat(serial[i])):
            assert x.shape == y.shape and t.equal(x, y), i
        prev = batch
    assert b.step == a.step == 7
    t.cuda.synchronize()
    assert bool((b._pos32 == -1).all()) and all(bool((p == -1).all()) for p in b._lane_pos)   # every stream's scratch handed back clean
    again = [x for x in b.batches(2)]                                  # a second call continues the sequence
    want = [a.sample_batch() for _ in range(2)]
    for g, w in zip(again, want):
        for x, y in zip(flat(g), flat(w)):
            assert t.equal(x, y)


def _pin_setup(seed=0, p_drop=0.0, hidden=32, layers=2, B=48):
    from laplace_amd.pinsage.model import PinSAGEModel
    from laplace_amd.pinsage.sampler import PinSAGESampler
    U, I = 2500, 800
    users, items = _graph(seed + 3, U, I, 40000)
    smp = PinSAGESampler(users, items, U, I, batch_size=B, num_layers=layers, seed=seed + 1)
    t.manual_seed(seed)
    model = PinSAGEModel(I, hidden, layers).to("cuda")
    with t.no_grad():
        model.bias.normal_(0, 0.1)          # the scorer bias starts at zero: give its gradient path something to show
    for cv in model.convs:
        cv.dropout.p = p_drop
    return model, smp


@pytest.mark.parametrize("hidden,layers", [(32, 2), (64, 2), (16, 3), (128, 1)])
def test_native_pinsage_step_gradients_equal_autograd(hidden, layers):
    """mi_pinsage_step_f32 (dropout off) against the op-by-op autograd iteration on the same batches: loss, every gradient
    (the dense projector / bias gradients included) to float rounding; then the update against torch.optim.Adam."""
    import copy
    from laplace_amd.pinsage.native import NativePinSAGEStep
    model, smp = _pin_setup(seed=2, hidden=hidden, layers=layers)
    twin = copy.deepcopy(model)
    opt_a, opt_b = t.optim.Adam(model.parameters(), lr=3e-3), t.optim.Adam(twin.parameters(), lr=3e-3)
    assert NativePinSAGEStep.unsupported_reason(model, opt_a) is None
    probe = NativePinSAGEStep(model, opt_a, keep_grads=True)
    model.train(); twin.train()
    for i in range(3):
        b = smp.sample_batch()
        la = probe.step(b)
        assert la is not None, probe.declined
        lb = twin(b["seeds"], b["pos"], b["neg"], b["blocks"]).mean()
        opt_b.zero_grad()
        lb.backward()
        assert abs(float(la) - float(lb)) <= 1e-6 * max(1.0, abs(float(lb))), i
        for (n, p), q in zip(model.named_parameters(), twin.parameters()):
            scale = float(q.grad.abs().max()) + 1e-12
            assert float((p.grad - q.grad).abs().max()) <= 2e-5 * scale + 1e-9, (i, n)
        assert float(model.proj.weight.grad.abs().sum()) > 0 and float(model.bias.grad.abs().sum()) > 0
    # the full step: Adam over every tensor, dense tables included; the two dense gradient buffers end all-zero.  Adam turns
    # a gradient element of 1e-9 against 0 into a different step, so torch.optim.Adam is fed the executor's own gradients
    # (the iteration is deterministic: the full step below computes exactly those again)
    full = NativePinSAGEStep(model, opt_a)
    for i in range(3):
        b = smp.sample_batch()
        twin.load_state_dict(model.state_dict())
        for p, q in zip(model.parameters(), twin.parameters()):
            if opt_a.state[p]:
                if not opt_b.state[q]:
                    opt_b.state[q] = {"step": t.tensor(0.0), "exp_avg": t.zeros_like(q), "exp_avg_sq": t.zeros_like(q)}
                for k in ("exp_avg", "exp_avg_sq"):
                    opt_b.state[q][k].copy_(opt_a.state[p][k])
                opt_b.state[q]["step"].fill_(float(opt_a.state[p]["step"]))
        assert probe.step(b) is not None
        for p, q in zip(model.parameters(), twin.parameters()):
            q.grad = p.grad.detach().clone()
        opt_b.step()
        model.proj.weight.grad.zero_(); model.bias.grad.zero_()      # what the probe left behind
        la = full.step(b)
        assert la is not None, full.declined
        for (n, p), q in zip(model.named_parameters(), twin.parameters()):
            assert float((p - q).abs().max()) <= 2e-6, (i, n)
            assert float(opt_a.state[p]["step"]) == float(opt_b.state[q]["step"]) == i + 1
            for k in ("exp_avg", "exp_avg_sq"):      # fused-multiply-add against two roundings: 1e-6 of the tensor's scale
                ma, mb = opt_a.state[p][k], opt_b.state[q][k]
                assert float((ma - mb).abs().max()) <= 1e-6 * float(mb.abs().max()) + 1e-20, (i, n, k)
        assert float(model.proj.weight.grad.abs().max()) == 0.0 and float(model.bias.grad.abs().max()) == 0.0


def test_native_pinsage_dropout_is_reproducible_and_training_learns():
    """Philox dropout keyed on (seed, iteration): two executors with the same seed give the same loss and gradients on the
    same batch, a different seed does not; train_epoch picks the native step by itself and the loss falls."""
    import copy
    from laplace_amd.pinsage.model import train_epoch
    from laplace_amd.pinsage.native import NativePinSAGEStep
    model, smp = _pin_setup(seed=4, p_drop=0.5, hidden=32, layers=2)
    b = smp.sample_batch()
    outs = []
    for sd in (7, 7, 8):
        m = copy.deepcopy(model)
        o = t.optim.Adam(m.parameters(), lr=3e-3)
        st = NativePinSAGEStep(m, o, seed=sd, keep_grads=True)
        m.train()
        loss = st.step(b)
        assert loss is not None, st.declined
        outs.append((float(loss), [p.grad.clone() for p in m.parameters()]))
    assert outs[0][0] == outs[1][0] and all(t.equal(x, y) for x, y in zip(outs[0][1], outs[1][1]))
    assert outs[0][0] != outs[2][0]
    # dropout's backward uses the forward's masks: a finite difference of the loss along a random direction of Q_0 (same
    # seed and iteration => same masks) agrees with the gradient
    m = copy.deepcopy(model)
    w = m.convs[0].Q.weight
    g0 = outs[0][1][[id(p) for p in model.parameters()].index(id(model.convs[0].Q.weight))]
    direction = t.randn_like(w)
    direction /= direction.norm()
    eps = 1e-2
    vals = []
    for sgn in (+1, -1):
        mm = copy.deepcopy(model)
        with t.no_grad():
            mm.convs[0].Q.weight.add_(sgn * eps * direction)
        st = NativePinSAGEStep(mm, t.optim.Adam(mm.parameters(), lr=3e-3), seed=7, keep_grads=True)
        mm.train()
        vals.append(float(st.step(b)))
    fd = (vals[0] - vals[1]) / (2 * eps)
    an = float((g0 * direction).sum())
    assert abs(fd - an) <= 0.1 * max(abs(an), 1e-3) + 2e-3, (fd, an)
    opt = t.optim.Adam(model.parameters(), lr=3e-3)
    first = train_epoch(model, opt, smp, 40)
    for _ in range(3):
        last = train_epoch(model, opt, smp, 40)
    assert np.isfinite(last).all() and np.mean(last) < np.mean(first)
    assert float(opt.state[model.proj.weight]["step"]) == 160      # every iteration went through the executor's Adam


def test_native_pinsage_step_declines_what_it_does_not_take():
    from laplace_amd.pinsage.native import NativePinSAGEStep
    model, smp = _pin_setup(seed=6)
    assert NativePinSAGEStep.unsupported_reason(model, t.optim.SGD(model.parameters(), lr=0.1)) is not None
    assert NativePinSAGEStep.unsupported_reason(model, t.optim.Adam(model.parameters(), lr=0.1, weight_decay=0.1)) is not None
    opt = t.optim.Adam(model.parameters(), lr=3e-3)
    st = NativePinSAGEStep(model, opt)
    model.train()
    smp.device_batches = False
    b = smp.sample_batch()                                   # index-op batch: no CSRs with the blocks
    before = [p.detach().clone() for p in model.parameters()]
    st.build_csrs = False
    assert st.step(b) is None and "CSR" in st.declined
    assert all(t.equal(x, p) for x, p in zip(before, model.parameters()))
    st.build_csrs = True                                     # the default: the executor builds them (two sorts per block)
    smp.device_batches = True
    same = smp.sample_batch(smp.step - 1)                    # the same batch, device-built
    import copy
    twin = copy.deepcopy(model)
    st2 = NativePinSAGEStep(twin, t.optim.Adam(twin.parameters(), lr=3e-3), seed=st.seed)
    twin.train()
    la, lb = st.step(b), st2.step(same)
    assert la is not None and lb is not None and float(la) == float(lb)
    assert all(t.equal(p, q) for p, q in zip(model.parameters(), twin.parameters()))
    model.eval()
    smp.device_batches = True
    assert st.step(smp.sample_batch()) is None and "eval" in st.declined


def test_native_pinsage_data_parallel_world_one_is_the_plain_step():
    """data_parallel=True with one process: compact gradient rows -> the exchange buffer -> mi_pinsage_apply_f32 with scale 1
    must give exactly the single-call iteration (0 + x = x): parameters, moments, zero-kept buffers."""
    import copy
    from laplace_amd.pinsage.native import NativePinSAGEStep
    model, smp = _pin_setup(seed=8, p_drop=0.5, hidden=32, layers=2)
    twin = copy.deepcopy(model)
    oa, ob = t.optim.Adam(model.parameters(), lr=3e-3), t.optim.Adam(twin.parameters(), lr=3e-3)
    dp, plain = NativePinSAGEStep(model, oa, data_parallel=True, seed=5), NativePinSAGEStep(twin, ob, seed=5)
    dp.exchange_capacity = (3 * smp.batch_size * (1 + smp.T) ** smp.n_layers, 3 * smp.batch_size)
    model.train(); twin.train()
    for i in range(4):
        b = smp.sample_batch()
        la, lb = dp.step(b), plain.step(b)
        assert la is not None and lb is not None, (dp.declined, plain.declined)
        assert float(la) == float(lb)
        for (n, p), q in zip(model.named_parameters(), twin.parameters()):
            assert t.equal(p, q), (i, n)
            assert t.equal(oa.state[p]["exp_avg_sq"], ob.state[q]["exp_avg_sq"]), (i, n)
        assert float(model.proj.weight.grad.abs().max()) == 0.0 and float(model.bias.grad.abs().max()) == 0.0


def _pin_dp_worker(rank, world, port, ret):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)     # both ranks on the one card; gloo carries the exchange
    try:
        import copy
        from laplace_amd.dist_ranker import broadcast_parameters
        from laplace_amd.pinsage.model import train_epoch
        from laplace_amd.pinsage.native import NativePinSAGEStep
        from laplace_amd.pinsage.sampler import PinSAGESampler
        U, I = 2500, 800
        users, items = _graph(11, U, I, 40000)
        smp = PinSAGESampler(users, items, U, I, batch_size=32, seed=100 + rank)      # a rank's own batches
        t.manual_seed(50 + rank)
        from laplace_amd.pinsage.model import PinSAGEModel
        model = PinSAGEModel(I, 32, 2).to("cuda")
        for cv in model.convs:
            cv.dropout.p = 0.0
        broadcast_parameters(model)
        probe_model = copy.deepcopy(model)
        opt = t.optim.Adam(model.parameters(), lr=3e-3)
        before = [p.detach().clone() for p in model.parameters()]
        # this rank's own gradients of its first batch (single-process probe on a copy)
        probe = NativePinSAGEStep(probe_model, t.optim.Adam(probe_model.parameters(), lr=3e-3), keep_grads=True)
        probe_model.train()
        first = smp.sample_batch(0)
        assert probe.step(first) is not None
        losses = train_epoch(model, opt, smp, 1)                                       # picks the data-parallel executor
        after1 = [p.detach().cpu().clone() for p in model.parameters()]
        losses += train_epoch(model, opt, smp, 5)
        t.cuda.synchronize()
        ret[rank] = {"own": [p.grad.detach().cpu().clone() for p in probe_model.parameters()], "after1": after1,
                     "before": [x.cpu() for x in before], "after": [p.detach().cpu() for p in model.parameters()],
                     "losses": losses, "step": float(opt.state[model.proj.weight]["step"]),
                     "dense_zero": float(model.proj.weight.grad.abs().max()) == 0.0}
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_native_pinsage_data_parallel_two_ranks():
    """Two processes on the card, their own batches: replicas stay bitwise identical over 6 iterations, the dense gradient
    buffers end zero, and the FIRST update is Adam's first step on the mean of the two ranks' own gradients."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ret = mp.Manager().dict()
    mp.spawn(_pin_dp_worker, args=(2, port, ret), nprocs=2, join=True)
    a, b = ret[0], ret[1]
    assert a["step"] == b["step"] == 6 and a["dense_zero"] and b["dense_zero"]
    assert all(np.isfinite(a["losses"])) and len(a["losses"]) == 6
    for x, y in zip(a["after"], b["after"]):
        assert t.equal(x, y)
    assert not all(t.equal(x, y) for x, y in zip(a["own"], b["own"]))                 # the batches did differ
    for i in range(len(a["own"])):
        g = 0.5 * (a["own"][i] + b["own"][i])                                          # the mean gradient of iteration 1
        want = a["before"][i] - 3e-3 * g / (g.abs() + 1e-8)                            # Adam, step 1
        big = g.abs() > 1e-6                                                           # (a 1e-9 gradient is all rounding)
        assert big.any()
        assert t.allclose(a["after1"][i][big], want[big], rtol=1e-4, atol=2e-6), i
        assert t.equal(a["after1"][i][g == 0], a["before"][i][g == 0])                 # untouched rows do not move

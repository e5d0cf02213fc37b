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
    assert bool((b._pos32 == -1).all())
    again = [x for x in b.batches(2)]                                  # a second call continues the sequence
    want = [a.sample_batch() for _ in range(2)]
    for g, w in zip(again, want):
        for x, y in zip(flat(g), flat(w)):
            assert t.equal(x, y)

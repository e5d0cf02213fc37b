"""GPU parity tests for the LightGCN hot path: HIP kernels (through the C ABI) vs the oracle.

Bars: bit-exact for integer / index work; fp32 embeddings within 1e-4 absolute (north_star), in
practice asserted tighter where the arithmetic allows; top-K indices exact.
"""
import os

import numpy as np
import pytest
import torch as t

from oracle import lightgcn_ref as R

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from laplace_amd import ops
    return ops


def _rand_graph(n_rows, n_cols, nnz, seed):
    g = t.Generator().manual_seed(seed)
    return t.randint(0, n_rows, (nnz,), generator=g), t.randint(0, n_cols, (nnz,), generator=g)


# ---------------------------------------------------------------------------- K4: CSR build
@pytest.mark.parametrize("n_rows,n_cols,nnz", [(1, 1, 1), (7, 5, 0), (37, 91, 400), (1000, 1000, 20000), (3, 200000, 5000)])
def test_coo_to_csr_bit_exact(n_rows, n_cols, nnz):
    ops = _ops()
    row, col = _rand_graph(n_rows, n_cols, nnz, seed=nnz + n_rows)
    if nnz > 10:  # force duplicates and an empty leading row
        row[:5], col[:5] = row[5], col[5]
    a = ops.coo_to_csr(row.to(DEV), col.to(DEV), n_rows, n_cols)
    rowptr, col_s, _ = R.sparse_tensor_csr(row, col, n_rows, n_cols)
    assert t.equal(a.rowptr.cpu().long(), rowptr)
    assert t.equal(a.col.cpu().long(), col_s)
    perm = a.perm.cpu().long()
    assert sorted(perm.tolist()) == list(range(nnz))
    assert t.equal(row[perm] * n_cols + col[perm], rowptr.new_tensor(
        (t.repeat_interleave(t.arange(n_rows), rowptr[1:] - rowptr[:-1]) * n_cols + col_s).tolist()))


def test_csr_transpose_bit_exact():
    ops = _ops()
    row, col = _rand_graph(300, 170, 6000, seed=3)
    a = ops.coo_to_csr(row.to(DEV), col.to(DEV), 300, 170)
    a.val = t.rand(a.nnz, device=DEV)
    at = ops.csr_transpose(a)
    rowptr_t, col_t, _ = R.sparse_tensor_csr(col, row, 170, 300)
    assert t.equal(at.rowptr.cpu().long(), rowptr_t) and t.equal(at.col.cpu().long(), col_t)
    # values follow their entries: dense check
    dense = t.zeros(300, 170)
    ar = t.repeat_interleave(t.arange(300), (a.rowptr[1:] - a.rowptr[:-1]).cpu().long())
    dense.index_put_((ar, a.col.cpu().long()), a.val.cpu(), accumulate=True)
    dense_t = t.zeros(170, 300)
    atr = t.repeat_interleave(t.arange(170), (at.rowptr[1:] - at.rowptr[:-1]).cpu().long())
    dense_t.index_put_((atr, at.col.cpu().long()), at.val.cpu(), accumulate=True)
    assert t.allclose(dense.T, dense_t, atol=1e-6)


# ---------------------------------------------------------------------------- K3: gcn_norm
def test_gcn_norm_matches_oracle():
    ops = _ops()
    n = 500
    row, col = _rand_graph(n, n, 7000, seed=11)
    row[row == 17] = 18  # an empty row -> deg 0 -> dis 0 (inf masked)
    a = ops.coo_to_csr(row.to(DEV), col.to(DEV), n, n)
    val, dis = ops.gcn_norm(a)
    rowptr, col_s, _ = R.sparse_tensor_csr(row, col, n, n)
    want = R.gcn_norm_csr(rowptr, col_s)
    assert t.allclose(val.cpu(), want, rtol=3e-7, atol=0)
    assert float(dis[17]) == 0.0
    # weighted variant
    w = t.rand(7000) + 0.5
    a.val = ops.gather_f32(w.to(DEV), a.perm)
    val_w, _ = ops.gcn_norm(a, a.val)
    want_w = R.gcn_norm_csr(rowptr, col_s, w[R.sparse_tensor_csr(row, col, n, n)[2]])
    assert t.allclose(val_w.cpu(), want_w, rtol=1e-6, atol=0)


# ---------------------------------------------------------------------------- K1: SpMM
def _csr_with_vals(row, col, n_rows, n_cols, seed):
    ops = _ops()
    a = ops.coo_to_csr(row.to(DEV), col.to(DEV), n_rows, n_cols)
    g = t.Generator().manual_seed(seed)
    a.val = (t.rand(a.nnz, generator=g) - 0.3).to(DEV)
    return a


def _oracle_spmm(a, X):
    return R.spmm_c(a.rowptr.cpu(), a.col.cpu(), a.val.cpu(), X.cpu())


@pytest.mark.parametrize("d", [4, 32, 64, 96, 128, 256, 320, 512])
def test_spmm_feature_widths(d):
    ops = _ops()
    n_rows, n_cols = 777, 431  # not multiples of 64
    row, col = _rand_graph(n_rows, n_cols, 9000, seed=d)
    row[row == 5] = 6  # empty row
    a = _csr_with_vals(row, col, n_rows, n_cols, seed=d)
    X = t.randn(n_cols, d, generator=t.Generator().manual_seed(d)).to(DEV)
    Y = t.full((n_rows, d), float("nan"), device=DEV)
    ops.spmm(a, X, Y=Y)
    want = _oracle_spmm(a, X)
    assert t.allclose(Y.cpu(), want, atol=2e-5, rtol=1e-5)
    assert float(Y[5].abs().max()) == 0.0


def test_spmm_hub_rows_split_path_and_determinism():
    """One 100K-degree row, a few mid hubs, many short rows; result independent of scheduling."""
    ops = _ops()
    n, d = 3000, 128
    g = t.Generator().manual_seed(0)
    hub = t.stack([t.full((100_000,), 7), t.randint(0, n, (100_000,), generator=g)])
    mid = t.stack([t.randint(10, 14, (3000,), generator=g), t.randint(0, n, (3000,), generator=g)])
    rest = t.stack(_rand_graph(n, n, 20000, seed=1))
    ei = t.cat([hub, mid, rest], dim=1)
    a = _csr_with_vals(ei[0], ei[1], n, n, seed=2)
    a.val = a.val * 0.01
    a.plan = ops.build_spmm_plan(a, chunk=256)
    assert a.plan.n_long_rows >= 5 and a.plan.n_items >= 100_000 // 256
    X = t.randn(n, d, generator=g).to(DEV)
    Y1, Y2 = t.empty(n, d, device=DEV), t.empty(n, d, device=DEV)
    ops.spmm(a, X, Y=Y1)
    ops.spmm(a, X, Y=Y2)
    assert t.equal(Y1, Y2)  # no atomics: bitwise reproducible
    want = _oracle_spmm(a, X)
    assert t.allclose(Y1.cpu(), want, atol=5e-4, rtol=1e-4)  # 100K-term fp32 sums, different order
    # against float64 accumulation the HIP path must be at least as good as the sequential oracle
    L = R.clib()
    y64 = t.empty(n, d, dtype=t.float64)
    rp, c, v, Xc = a.rowptr.cpu(), a.col.cpu(), a.val.cpu(), X.cpu()
    L.ref_spmm_csr_f64acc(n, d, rp.data_ptr(), c.data_ptr(), v.data_ptr(), Xc.data_ptr(), d, y64.data_ptr(), d)
    err_hip = (Y1.cpu().double() - y64).abs().max()
    err_seq = (want.double() - y64).abs().max()
    assert err_hip <= max(2 * err_seq, 1e-5)
    # a different chunk size changes only rounding
    a.plan = ops.build_spmm_plan(a, chunk=1000)
    Y3 = t.empty(n, d, device=DEV)
    ops.spmm(a, X, Y=Y3)
    assert t.allclose(Y3, Y1, atol=5e-4, rtol=1e-4)


def _hub_graph(seed, n=600, hubs=(2000, 700, 300)):
    g = t.Generator().manual_seed(seed)
    rows = t.cat([t.full((h,), 5 + 7 * i, dtype=t.int64) for i, h in enumerate(hubs)] + [t.randint(0, n, (1500,), generator=g)])
    cols = t.randint(0, n, (rows.numel(),), generator=g)
    return rows, cols


@pytest.mark.parametrize("d", [8, 32, 64, 128, 256, 320])
def test_spmm_planless_long_rows_are_summed_by_the_workgroup(d):
    """Adjacencies below PLAN_MIN_NNZ run without a split-row plan (per-batch subgraphs of the ranker); rows longer
    than 128 entries are then summed cooperatively by the block's sub-groups.  Against float64, with every epilogue
    form, on rows of 513 .. 20 000 entries placed at both ends of a workgroup's row range; bitwise reproducible."""
    ops = _ops()
    g = t.Generator().manual_seed(d)
    n_rows, n_cols = 300, 4000
    lens = {0: 129, 7: 20000, 8: 128, 31: 3000, 32: 1500, 150: 5000, 299: 700, 200: 513}
    rows = [t.full((L,), r) for r, L in lens.items()] + [t.randint(0, n_rows, (3000,), generator=g)]
    row = t.cat(rows)
    col = t.randint(0, n_cols, (row.numel(),), generator=g)
    a = ops.coo_to_csr(row.to(DEV), col.to(DEV), n_rows, n_cols, want_perm=False)
    assert a.nnz < ops.PLAN_MIN_NNZ
    a.val = (t.rand(a.nnz, generator=g) + 0.5).to(DEV)
    X = t.randn(n_cols, d, generator=g).to(DEV)
    A = t.randn(n_rows, d, generator=g).to(DEV)
    Y, Sx = t.empty(n_rows, d, device=DEV), t.empty(n_rows, d, device=DEV)
    ops.spmm(a, X, Y=Y, addend=A, S=Sx, scale=0.25)
    assert a.plan is None
    rp, cc, vv = a.rowptr.cpu().long(), a.col.cpu().long(), a.val.cpu().double()
    Xd = X.cpu().double()
    for r in list(lens) + [1, 100, 298]:
        b, e = int(rp[r]), int(rp[r + 1])
        want = (vv[b:e, None] * Xd[cc[b:e]]).sum(0)
        tol = 1e-6 * float((vv[b:e, None] * Xd[cc[b:e]].abs()).sum(0).max()) + 1e-6
        assert (Y[r].cpu().double() - want).abs().max() <= tol, r
        assert (Sx[r].cpu().double() - 0.25 * (A[r].cpu().double() + want)).abs().max() <= tol, r
    Y2 = t.empty_like(Y)
    ops.spmm(a, X, Y=Y2)
    assert t.equal(Y2, Y)


@pytest.mark.parametrize("d", [16, 64, 128])
@pytest.mark.parametrize("n_streams", [32, 96])
def test_spmm_sweep_plan_equals_work_item_plan(d, n_streams):
    """SWEEP form of the split rows (include/laplace_hip.h, mi_spmm_sweep): streams per (XCD, sub-group), accumulators in
    LDS across all bands, 8 partial rows per row part.  Every epilogue / sparse-operand form against the work-item plan
    of the same adjacency (same sums, another association) and against float64; bitwise reproducible; a width beyond
    128 falls back to a work-item plan by itself."""
    ops = _ops()
    g = t.Generator().manual_seed(d + n_streams)
    n = 6000
    hubs = {3: 5000, 10: 2600, 777: 900, 4000: 300, 5999: 257, 17: 256}
    rows = [t.full((L,), r) for r, L in hubs.items()] + [t.randint(0, n, (30000,), generator=g)]
    row = t.cat(rows)
    col = t.randint(0, n, (row.numel(),), generator=g)
    a = ops.coo_to_csr(row.to(DEV), col.to(DEV), n, n, want_perm=False)
    a.val = (t.rand(a.nnz, generator=g) + 0.5).to(DEV)
    sweep = ops.build_sweep_plan(a, chunk=256, band=64, n_streams=n_streams)
    items = ops.build_spmm_plan(a, chunk=256, band=64)
    assert sweep is not None and sweep.sweep is not None and items.sweep is None
    assert sweep.n_long_rows == items.n_long_rows >= 5 and t.equal(sweep.long_rows, items.long_rows)
    assert sweep.n_items == 8 * int(sweep.sweep.n_slots) and int(sweep.sweep.n_slots) >= 5
    X = t.randn(n, d, generator=g).to(DEV)
    A = t.randn(n, d, generator=g).to(DEV)
    out = {}
    for name, plan in (("sweep", sweep), ("items", items)):
        a.plan = plan
        Y, S = t.empty(n, d, device=DEV), t.empty(n, d, device=DEV)
        ops.spmm(a, X, Y=Y, addend=A, S=S, scale=0.5)
        out[name] = (Y, S)
    scale = float(out["items"][0].abs().max())
    assert (out["sweep"][0] - out["items"][0]).abs().max() <= 1e-5 * scale
    assert (out["sweep"][1] - out["items"][1]).abs().max() <= 1e-5 * scale
    rp, cc, vv, Xd = a.rowptr.cpu().long(), a.col.cpu().long(), a.val.cpu().double(), X.cpu().double()
    for r in list(hubs) + [0, 1, 5998]:
        b, e = int(rp[r]), int(rp[r + 1])
        want = (vv[b:e, None] * Xd[cc[b:e]]).sum(0)
        tol = 1e-6 * float((vv[b:e, None] * Xd[cc[b:e]].abs()).sum(0).max()) + 1e-6
        assert (out["sweep"][0][r].cpu().double() - want).abs().max() <= tol, r
    a.plan = sweep
    Y2 = t.empty(n, d, device=DEV)
    ops.spmm(a, X, Y=Y2)
    assert t.equal(Y2, out["sweep"][0])
    # sparse-operand forms on the sweep plan: compact X through x_map, a row list
    keep = t.rand(n, generator=g) < 0.3
    xmap = t.full((n,), -1, dtype=t.int32)
    xmap[keep] = t.arange(int(keep.sum()), dtype=t.int32)
    Xc = X[keep.to(DEV)].contiguous()
    Xz = X * keep.to(DEV)[:, None]
    Ys, Yd = t.empty(n, d, device=DEV), t.empty(n, d, device=DEV)
    ops.spmm(a, Xc, Y=Ys, x_map=xmap.to(DEV))
    ops.spmm(a, Xz, Y=Yd)
    assert (Ys - Yd).abs().max() <= 1e-5 * scale
    rl = t.tensor([3, 5, 10, 5999, 17, 4000, 2], dtype=t.int32, device=DEV)
    Yl = t.empty(rl.numel(), d, device=DEV)
    ops.spmm(a, X, Y=Yl, row_list=rl)
    assert (Yl - out["sweep"][0][rl.long()]).abs().max() <= 1e-5 * scale
    # Adam in the epilogue of the fix-up rows
    p0 = t.randn(n, d, generator=g).to(DEV)
    res = {}
    for name, plan in (("sweep", sweep), ("items", items)):
        a.plan = plan
        p, m, v = p0.clone(), t.zeros(n, d, device=DEV), t.zeros(n, d, device=DEV)
        ops.spmm(a, X, addend=A, adam=dict(p=p, m=m, v=v, step=1, lr=1e-2))
        res[name] = p
    assert (res["sweep"] - res["items"]).abs().max() <= 2e-2 + 1e-6  # |Adam step| <= lr each: signs of ~0 gradients may differ
    assert ((res["sweep"] - res["items"]).abs() > 1e-6).float().mean() < 1e-3
    # wider than the sweep form: falls back to a work-item plan of the same adjacency
    a.plan = sweep
    Xw = t.randn(n, 256, generator=g).to(DEV)
    Yw = t.empty(n, 256, device=DEV)
    ops.spmm(a, Xw, Y=Yw)
    assert sweep.wide is not None and sweep.wide.sweep is None
    r = 3
    b, e = int(rp[r]), int(rp[r + 1])
    want = (vv[b:e, None] * Xw.cpu().double()[cc[b:e]]).sum(0)
    assert (Yw[r].cpu().double() - want).abs().max() <= 1e-3

def test_streaming_stores_equal_cached_stores():
    """Outputs of 64 MB and more leave through write-through (sc1) stores issued from inline asm (mi_store4); smaller ones through
    plain stores.  Round 4 found the asm form short of the wait states a 128-bit store needs before its data registers are
    reused — invisible to every test below that size.  Here one product is large enough to stream (150 K x 128 floats = 77 MB)
    and the same rows computed as two row slices are not: every row's sum is formed the same way (a plan cuts a row by its own
    entries), so the two must agree bit for bit — dense, mapped with and without the rare-live hint — and repeat bit for bit."""
    ops = _ops()
    g = t.Generator().manual_seed(11)
    n, d = 150_000, 128
    degs = t.cat([t.tensor([30000, 9000, 4000]), t.randint(260, 900, (300,), generator=g)])
    hub_rows = t.randperm(n, generator=g)[: degs.numel()]
    row = t.cat([t.full((int(L),), int(r)) for r, L in zip(hub_rows, degs)] + [t.randint(0, n, (1_200_000,), generator=g)])
    col = t.randint(0, n, (row.numel(),), generator=g)
    a = ops.coo_to_csr(row.to(DEV), col.to(DEV), n, n, want_perm=False)
    a.val = (t.rand(a.nnz, generator=g) + 0.5).to(DEV)
    a.plan = ops.build_spmm_plan(a, chunk=256, band=16384, sweep=False)
    halves = [ops.row_slice(a, 0, n // 2), ops.row_slice(a, n // 2, n)]
    for h in halves:
        h.plan = ops.build_spmm_plan(h, chunk=256, band=16384, sweep=False)
    assert a.plan.n_items > 1000 and all(h.plan.n_items > 100 for h in halves)
    X = (t.randn(n, d, generator=g) * 0.1).to(DEV)
    keep = t.rand(n, generator=g) < 0.03
    x_map = t.where(keep, t.cumsum(keep.int(), 0) - 1, t.full((n,), -1)).to(t.int32).to(DEV)
    Xc = X[keep.to(DEV)].contiguous()

    def whole(**kw):
        Y = t.full((n, d), float("nan"), device=DEV)
        ops.spmm(a, kw.pop("X"), Y=Y, **kw)
        return Y

    def sliced(**kw):
        Y = t.full((n, d), float("nan"), device=DEV)
        Xin = kw.pop("X")
        ops.spmm(halves[0], Xin, Y=Y[: n // 2], **kw)
        ops.spmm(halves[1], Xin, Y=Y[n // 2:], **kw)
        return Y

    for kw in (dict(X=X), dict(X=Xc, x_map=x_map), dict(X=Xc, x_map=x_map, x_rare=True)):
        big, big2, small = whole(**dict(kw)), whole(**dict(kw)), sliced(**dict(kw))
        t.cuda.synchronize()
        assert t.equal(big, big2), sorted(kw)
        assert t.equal(big, small), sorted(kw)


@pytest.mark.parametrize("d", [32, 128, 512])
def test_packed_plan_entries_are_the_same_product_and_follow_a_reweighting(d):
    """mi_spmm_plan.epos / ecol / eval (round 4): a banded plan's work items read launch-ordered COPIES of the split rows' entries.
    The product is bitwise the one read from the CSR arrays (dense, mapped, row-list forms); the copies hold values, so a
    re-weighted adjacency (val changed in place, or replaced) must be seen by the next product."""
    ops = _ops()
    g = t.Generator().manual_seed(d)
    n = 5000
    degs = t.cat([t.tensor([4000, 1500]), t.randint(260, 600, (40,), generator=g)])
    hub_rows = t.randperm(n, generator=g)[: degs.numel()]
    row = t.cat([t.full((int(L),), int(r)) for r, L in zip(hub_rows, degs)] + [t.randint(0, n, (20000,), generator=g)])
    col = t.randint(0, n, (row.numel(),), generator=g)
    a = ops.coo_to_csr(row.to(DEV), col.to(DEV), n, n, want_perm=False)
    a.val = (t.rand(a.nnz, generator=g) + 0.5).to(DEV)
    X = t.randn(n, d, generator=g).to(DEV)
    keep = t.rand(n, generator=g) < 0.2
    x_map = t.where(keep, t.cumsum(keep.int(), 0) - 1, t.full((n,), -1)).to(t.int32).to(DEV)
    Xc = X[keep.to(DEV)].contiguous()
    rows = t.cat([hub_rows[:10], t.randint(0, n, (200,), generator=g)]).to(t.int32).to(DEV)

    def products(plan):
        a.plan = plan
        Y = t.full((n, d), float("nan"), device=DEV)
        Ym = t.full((n, d), float("nan"), device=DEV)
        Yl = t.full((rows.numel(), d), float("nan"), device=DEV)
        ops.spmm(a, X, Y=Y)
        ops.spmm(a, Xc, Y=Ym, x_map=x_map)
        ops.spmm(a, X, Y=Yl, row_list=rows)
        return Y, Ym, Yl

    saved = ops.PACK_ENTRIES
    try:
        ops.PACK_ENTRIES = False
        plain = ops.build_spmm_plan(a, chunk=256, band=64, sweep=False)
        ops.PACK_ENTRIES = True
        packed = ops.build_spmm_plan(a, chunk=256, band=64, sweep=False)
    finally:
        ops.PACK_ENTRIES = saved
    assert not plain.packed and len(packed.packed) == 3 and packed.nnz_long >= int(degs.sum())
    epos, ecol, ev = packed.packed
    items = packed.items.view(-1, 4).cpu()
    real = items[:, 3] >= 0
    lens = (items[:, 2] - items[:, 1])[real]
    assert int(lens.sum()) == packed.nnz_long and int(epos.cpu()[real].max()) + int(lens[epos.cpu()[real].argmax()]) == packed.nnz_long
    q = int(real.nonzero()[7])                                           # one work item's copy, entry by entry
    b, e, p0 = int(items[q, 1]), int(items[q, 2]), int(epos[q])
    assert t.equal(ecol[p0:p0 + e - b], a.col[b:e]) and t.equal(ev[p0:p0 + e - b], a.val[b:e])
    want = products(plain)
    got = products(packed)
    for w_, g_ in zip(want, got):
        assert t.equal(w_, g_)
    # re-weighting in place, then by replacement: the packed values follow
    a.val.mul_(1.5)
    got2 = products(packed)
    want2 = products(plain)
    for w_, g_ in zip(want2, got2):
        assert t.equal(w_, g_)
    assert not t.equal(got2[0], got[0])
    a.val = (a.val * 0.25 + 0.1).contiguous()
    for w_, g_ in zip(products(plain), products(packed)):
        assert t.equal(w_, g_)


@pytest.mark.parametrize("band", [0, 64])
@pytest.mark.parametrize("d,live", [(32, 0.02), (64, 0.3), (128, 0.005), (128, 0.0), (128, 1.0), (256, 0.02), (512, 0.02), (512, 0.9)])
def test_rare_live_columns_hint_is_bitwise_the_mapped_product(d, live, band):
    """mi_spmm_ex.x_bits (round 4): a mapped operand whose live columns are rare — the first backward product of the fused
    step on BASELINE configs[3], the batch's 131 K users among 8 M.  The split rows' work items then test one bit per column
    before the map, gather only the live entries and leave out the partial rows of work items without any (the fix-up reads
    their flag).  Same sums in the same order: bitwise the product without the hint, on both streams' halves, with work
    items longer than a sub-group's batch, at every width class; and the dense product of the expanded operand to rounding."""
    ops = _ops()
    g = t.Generator().manual_seed(7 * d + band + int(1000 * live))
    n = 6000
    degs = t.cat([t.tensor([5000, 2600, 900]), t.randint(260, 700, (60,), generator=g)])   # 63 rows above chunk = 256
    hub_rows = t.randperm(n, generator=g)[: degs.numel()]
    row = t.cat([t.full((int(L),), int(r)) for r, L in zip(hub_rows, degs)] + [t.randint(0, n, (30000,), generator=g)])
    col = t.randint(0, n, (row.numel(),), generator=g)
    a = ops.coo_to_csr(row.to(DEV), col.to(DEV), n, n, want_perm=False)
    a.val = (t.rand(a.nnz, generator=g) + 0.5).to(DEV)
    a.plan = ops.build_spmm_plan(a, chunk=256, band=band, sweep=False)
    assert a.plan.n_items > 0 and a.plan.sweep is None
    keep = t.rand(n, generator=g) < live
    ids = keep.nonzero().view(-1)
    x_map = t.full((n,), -1, dtype=t.int32)
    x_map[ids] = t.randperm(ids.numel(), generator=g).to(t.int32)      # compact rows in another order than the columns
    Xc = t.randn(max(ids.numel(), 1), d, generator=g).to(DEV)
    x_map = x_map.to(DEV)
    A = t.randn(n, d, generator=g).to(DEV)
    out = {}
    saved = ops.SPMM_TWO_STREAMS
    try:
        for streams in (1, 0):
            ops.SPMM_TWO_STREAMS = streams
            for rare in (False, True):
                Y, S = t.full((n, d), float("nan"), device=DEV), t.full((n, d), float("nan"), device=DEV)
                ops.spmm(a, Xc, Y=Y, addend=A, S=S, scale=0.5, x_map=x_map, x_rare=rare)
                out[(streams, rare)] = (Y, S)
    finally:
        ops.SPMM_TWO_STREAMS = saved
    t.cuda.synchronize()
    ref = out[(1, False)]
    for key, (Y, S) in out.items():
        assert t.equal(Y, ref[0]) and t.equal(S, ref[1]), key
    X = t.zeros(n, d, device=DEV)
    if ids.numel():
        X[ids.to(DEV)] = Xc[x_map[ids.to(DEV)].long()]
    Yd = t.empty(n, d, device=DEV)
    ops.spmm(a, X, Y=Yd)
    assert (Yd - ref[0]).abs().max() <= 1e-5 * max(float(Yd.abs().max()), 1.0)
    if live == 0.0:
        assert float(ref[0].abs().max()) == 0.0 and t.equal(ref[1], 0.5 * A)


@pytest.mark.parametrize("tail_whole", [False, True])
@pytest.mark.parametrize("d", [32, 128])
def test_hybrid_plan_equals_the_banded_plan(d, tail_whole):
    """HYBRID split-row plan (round 4, ops.build_hybrid_plan): an adjacency with MORE long rows than the sweep form's
    accumulators hold — the longest rows in SWEEP form, the other split rows as banded work items behind the sweep's partial
    rows (one fix-up over both) or, tail_whole, whole in the short-row kernel.  Every form of the product against the banded
    work-item plan of the same adjacency (same sums, another association) and float64; bitwise reproducible."""
    ops = _ops()
    g = t.Generator().manual_seed(d + int(tail_whole))
    n, n_streams = 9000, 32                                   # 8 * 32 = 256 row-part accumulators per XCD
    degs = t.cat([t.tensor([6000, 4100, 2500]), t.randint(300, 1500, (380,), generator=g)])    # 383 rows above chunk = 256
    hub_rows = t.randperm(n, generator=g)[: degs.numel()]
    row = t.cat([t.full((int(L),), int(r)) for r, L in zip(hub_rows, degs)] + [t.randint(0, n, (40000,), generator=g)])
    col = t.randint(0, n, (row.numel(),), generator=g)
    a = ops.coo_to_csr(row.to(DEV), col.to(DEV), n, n, want_perm=False)
    a.val = (t.rand(a.nnz, generator=g) + 0.5).to(DEV)
    T = ops.sweep_degree_threshold(a, 256, n_streams)
    assert T is not None and T > 256                          # not every long row fits the sweep form
    assert ops.build_sweep_plan(a, chunk=256, band=64, n_streams=n_streams) is None
    hyb = ops.build_hybrid_plan(a, chunk=256, band=64, tail_whole=tail_whole, sweep_band=64, n_streams=n_streams)
    items = ops.build_spmm_plan(a, chunk=256, band=64)
    assert hyb is not None and hyb.sweep is not None and items.sweep is None
    deg = (a.rowptr[1:] - a.rowptr[:-1]).cpu()
    n_hub = int((deg > T).sum())
    assert 0 < n_hub < items.n_long_rows and int(hyb.sweep.n_slots) <= 8 * n_streams
    if tail_whole:
        assert hyb.items is None and hyb.n_long_rows == n_hub and int(hyb.struct.chunk) == T
    else:
        assert hyb.items is not None and hyb.n_long_rows == items.n_long_rows and int(hyb.struct.chunk) == 256
        assert hyb.n_items > 8 * int(hyb.sweep.n_slots)       # the work items' partial rows follow the sweep's
        assert sorted(hyb.long_rows.cpu().tolist()) == sorted(items.long_rows.cpu().tolist())
    X = t.randn(n, d, generator=g).to(DEV)
    A = t.randn(n, d, generator=g).to(DEV)
    out = {}
    for name, plan in (("hybrid", hyb), ("items", items)):
        a.plan = plan
        Y, S = t.empty(n, d, device=DEV), t.empty(n, d, device=DEV)
        ops.spmm(a, X, Y=Y, addend=A, S=S, scale=0.5)
        out[name] = (Y, S)
    scale = float(out["items"][0].abs().max())
    assert (out["hybrid"][0] - out["items"][0]).abs().max() <= 1e-5 * scale
    assert (out["hybrid"][1] - out["items"][1]).abs().max() <= 1e-5 * scale
    rp, cc, vv, Xd = a.rowptr.cpu().long(), a.col.cpu().long(), a.val.cpu().double(), X.cpu().double()
    for r in hub_rows[:12].tolist() + hub_rows[-12:].tolist() + [0, 1, n - 1]:
        b, e = int(rp[r]), int(rp[r + 1])
        want = (vv[b:e, None] * Xd[cc[b:e]]).sum(0)
        tol = 1e-6 * float((vv[b:e, None] * Xd[cc[b:e]].abs()).sum(0).max()) + 1e-6
        assert (out["hybrid"][0][r].cpu().double() - want).abs().max() <= tol, r
    a.plan = hyb
    Y2 = t.full((n, d), float("nan"), device=DEV)
    ops.spmm(a, X, Y=Y2)
    assert t.equal(Y2, out["hybrid"][0])                      # no float atomics: bitwise reproducible
    # both halves on two streams = the whole product
    saved = ops.SPMM_TWO_STREAMS
    try:
        ops.SPMM_TWO_STREAMS = 0
        Y3 = t.empty(n, d, device=DEV)
        ops.spmm(a, X, Y=Y3)
    finally:
        ops.SPMM_TWO_STREAMS = saved
    assert t.equal(Y3, Y2)
    # sparse-operand forms: compact X through x_map, a row list naming hub rows, tail rows and short rows
    keep = t.rand(n, generator=g) < 0.3
    xmap = t.full((n,), -1, dtype=t.int32)
    xmap[keep] = t.arange(int(keep.sum()), dtype=t.int32)
    Ys, Yd = t.empty(n, d, device=DEV), t.empty(n, d, device=DEV)
    ops.spmm(a, X[keep.to(DEV)].contiguous(), Y=Ys, x_map=xmap.to(DEV))
    ops.spmm(a, X * keep.to(DEV)[:, None], Y=Yd)
    assert (Ys - Yd).abs().max() <= 1e-5 * scale
    rl = t.cat([hub_rows[:3], hub_rows[-3:], t.tensor([2, 7])]).to(t.int32).to(DEV)
    Yl = t.empty(rl.numel(), d, device=DEV)
    ops.spmm(a, X, Y=Yl, row_list=rl)
    assert (Yl - out["hybrid"][0][rl.long()]).abs().max() <= 1e-5 * scale


@pytest.mark.parametrize("band,chunk", [(0, 256), (64, 256), (7, 50), (100, 1000), (1, 256)])
def test_spmm_plan_structure(band, chunk):
    """The work items partition the entries of every split row, slots are contiguous per row, banded items stay
    inside one band and sit in the launch blocks of queue (band mod 8)."""
    ops = _ops()
    from laplace_amd._lib import MI_SPMM_GROUP as G
    n = 600
    rows, cols = _hub_graph(seed=5)
    a = ops.coo_to_csr(rows.to(DEV), cols.to(DEV), n, n)
    plan = ops.build_spmm_plan(a, chunk=chunk, band=band)
    st = plan.struct
    rowptr, col = a.rowptr.cpu().numpy(), a.col.cpu().numpy()
    long_rows = np.nonzero(np.diff(rowptr) > chunk)[0]
    assert len(long_rows) >= 1 and st.n_long_rows == len(long_rows)
    assert np.array_equal(plan.long_rows.cpu().numpy()[: st.n_long_rows], long_rows)
    li = plan.long_index.cpu().numpy()
    assert np.array_equal(np.nonzero(li >= 0)[0], long_rows) and np.array_equal(li[long_rows], np.arange(len(long_rows)))
    items = plan.items.cpu().numpy().reshape(-1, 4)
    assert items.shape[0] == st.n_launch >= st.n_items
    real = items[items[:, 3] >= 0]
    assert real.shape[0] == st.n_items and np.array_equal(np.sort(real[:, 3]), np.arange(st.n_items))
    by_slot = real[np.argsort(real[:, 3])]
    item_ptr = plan.item_ptr.cpu().numpy()
    for i, r in enumerate(long_rows):
        seg = by_slot[item_ptr[i]: item_ptr[i + 1]]
        assert (seg[:, 0] == r).all() and seg[0, 1] == rowptr[r] and seg[-1, 2] == rowptr[r + 1]
        size = seg[:, 2] - seg[:, 1]
        assert np.array_equal(seg[1:, 1], seg[:-1, 2]) and (size <= chunk).all() and (size > 0).all()
        if band > 0:
            assert (col[seg[:, 1]] // band == col[seg[:, 2] - 1] // band).all()
    assert item_ptr[len(long_rows)] == st.n_items
    if band > 0:
        assert st.n_launch % (8 * G) == 0 and st.band == band
        pos = np.nonzero(items[:, 3] >= 0)[0]
        assert np.array_equal((pos // G) % 8, (col[items[pos, 1]] // band) % 8)
        for x in range(8):  # within a queue: bands ascending, then slots ascending; padding only at the tail
            q = items.reshape(-1, 8, G, 4)[:, x].reshape(-1, 4)
            nreal = int((q[:, 3] >= 0).sum())
            assert (q[:nreal, 3] >= 0).all()
            key = (col[q[:nreal, 1]] // band).astype(np.int64) * (1 << 32) + q[:nreal, 3]
            assert (np.diff(key) > 0).all()
    else:
        assert np.array_equal(items[:, 3], np.arange(st.n_items))


@pytest.mark.parametrize("d", [16, 64, 128, 256, 320])
@pytest.mark.parametrize("band", [0, 37])
def test_spmm_banded_plan_matches_unsplit_product(band, d):
    """Row-major and banded plans against the plan-less product (one sub-group per row whatever its length) and
    the oracle, with the fused epilogue, for every sub-group width."""
    ops = _ops()
    n = 600
    rows, cols = _hub_graph(seed=7)
    a = _csr_with_vals(rows, cols, n, n, seed=8)
    g = t.Generator().manual_seed(d)
    X, A = t.randn(n, d, generator=g).to(DEV), t.randn(n, d, generator=g).to(DEV)
    Y0, S0 = t.empty(n, d, device=DEV), t.empty(n, d, device=DEV)
    a.plan = None
    ops.spmm(a, X, Y=Y0, addend=A, S=S0, scale=0.5)  # nnz < PLAN_MIN_NNZ: no plan
    assert a.plan is None
    a.plan = ops.build_spmm_plan(a, chunk=64, band=band)
    assert a.plan.n_long_rows >= 3
    Y1, S1 = t.full((n, d), float("nan"), device=DEV), t.full((n, d), float("nan"), device=DEV)
    ops.spmm(a, X, Y=Y1, addend=A, S=S1, scale=0.5)
    Y2 = t.empty(n, d, device=DEV)
    ops.spmm(a, X, Y=Y2)
    assert t.equal(Y1, Y2)  # bitwise reproducible
    want = _oracle_spmm(a, X)
    scale = float(want.abs().max())
    assert float((Y1.cpu() - want).abs().max()) <= 1e-5 * scale and float((Y0.cpu() - want).abs().max()) <= 1e-5 * scale
    assert float((S1.cpu() - 0.5 * (A.cpu() + want)).abs().max()) <= 1e-5 * scale


@pytest.mark.parametrize("d,band", [(128, 0), (128, 37), (64, 37), (320, 0)])
def test_spmm_adam_epilogue_equals_spmm_then_adam(d, band):
    """mi_adam_args: the gradient consumed in the SpMM epilogue gives the bits of spmm + mi_adam_dense_f32, for
    short rows, split rows (fix-up kernel), rows without entries, compact addend and compact X."""
    ops = _ops()
    n = 600
    rows, cols = _hub_graph(seed=11)
    rows[rows == 17] = 18  # an empty row still gets its Adam update
    a = _csr_with_vals(rows, cols, n, n, seed=12)
    a.plan = ops.build_spmm_plan(a, chunk=64, band=band)
    g = t.Generator().manual_seed(d)
    X, A = t.randn(n, d, generator=g).to(DEV), t.randn(n, d, generator=g).to(DEV)
    nz = t.unique(t.randint(0, n, (200,), generator=g))
    gmap = t.full((n,), -1, dtype=t.int32)
    gmap[nz] = t.arange(nz.numel(), dtype=t.int32)
    gmap = gmap.to(DEV)
    Ac = t.randn(nz.numel(), d, generator=g).to(DEV)
    reg_w = (t.rand(n, generator=g) * 1e-3).to(DEV)
    hyp = dict(lr=1e-2, beta1=0.9, beta2=0.999, eps=1e-8)
    for kw in (dict(addend=A), dict(addend=Ac, addend_map=gmap), dict(x_map=gmap, addend=Ac, addend_map=gmap), dict()):
        Xin = Ac if "x_map" in kw else X
        for step in (1, 7):
            p0 = t.randn(n, d, generator=g).to(DEV)
            m0, v0 = (t.randn(n, d, generator=g) * 0.01).to(DEV), (t.rand(n, d, generator=g) * 1e-3).to(DEV)
            p1, m1, v1 = p0.clone(), m0.clone(), v0.clone()
            G = t.empty(n, d, device=DEV)
            ops.spmm(a, Xin, S=G, scale=0.25, **kw)
            ops.adam_step(p1, G, m1, v1, step=step, reg_w=reg_w, **hyp)
            p2, m2, v2 = p0.clone(), m0.clone(), v0.clone()
            ops.spmm(a, Xin, scale=0.25, adam=dict(p=p2, m=m2, v=v2, step=step, reg_w=reg_w, **hyp), **kw)
            assert t.equal(p1, p2) and t.equal(m1, m2) and t.equal(v1, v2)
            assert not t.equal(p1, p0)
            # S may be kept as well
            p3, m3, v3, G3 = p0.clone(), m0.clone(), v0.clone(), t.empty(n, d, device=DEV)
            ops.spmm(a, Xin, S=G3, scale=0.25, adam=dict(p=p3, m=m3, v=v3, step=step, reg_w=None, **hyp), **kw)
            p4, m4, v4 = p0.clone(), m0.clone(), v0.clone()
            ops.adam_step(p4, G, m4, v4, step=step, **hyp)
            assert t.equal(G3, G) and t.equal(p3, p4) and t.equal(v3, v4)


@pytest.mark.parametrize("d", [32, 64, 96, 128, 256])
@pytest.mark.parametrize("threads", [0, 256, 1024])
def test_spmm_hot_row_cache_is_bitwise_the_plain_launch(d, threads, monkeypatch):
    """mi_spmm_ex.hot_rows: rows of X served from LDS by the persistent short-row launch — same entries, same order, same
    bits as the plain launch: Y / S / addend_map / Adam epilogue forms, a cached range in the middle of X, at its end,
    larger than the LDS share (clipped) and one row; split rows, empty rows and columns below the range alongside."""
    ops = _ops()
    monkeypatch.setattr(ops, "HOT_THREADS", threads)
    n_u, n_i = 1500, 700
    n = n_u + n_i
    g = t.Generator().manual_seed(d + threads)
    # bipartite-like: user rows gather item rows (popular ones first, Zipf-ish), item rows gather user rows
    e = 30000
    u = t.randint(0, n_u, (e,), generator=g)
    i = (t.rand(e, generator=g).pow(3) * n_i).long().clamp(max=n_i - 1) + n_u
    rows, cols = t.cat([u, i]), t.cat([i, u])
    rows[rows == 9] = 10                                   # an empty row
    a = _csr_with_vals(rows, cols, n, n, seed=3)
    a.plan = ops.build_spmm_plan(a, chunk=64, band=0)      # the popular item rows are split rows
    assert a.plan.n_long_rows > 10
    X, A = t.randn(n, d, generator=g).to(DEV), t.randn(n, d, generator=g).to(DEV)
    nz = t.unique(t.randint(0, n, (400,), generator=g))
    gmap = t.full((n,), -1, dtype=t.int32)
    gmap[nz] = t.arange(nz.numel(), dtype=t.int32)
    gmap = gmap.to(DEV)
    Ac = t.randn(nz.numel(), d, generator=g).to(DEV)
    hyp = dict(lr=1e-2, beta1=0.9, beta2=0.999, eps=1e-8, step=3)

    def run(hot, persistent=True):
        a.hot = hot
        monkeypatch.setattr(ops, "PERSISTENT_ROWS", persistent)
        out = []
        Y, S = t.empty(n, d, device=DEV), t.empty(n, d, device=DEV)
        ops.spmm(a, X, Y=Y)
        out.append(Y.clone())
        ops.spmm(a, X, Y=Y, addend=A, S=S, scale=0.25)
        out += [Y.clone(), S.clone()]
        ops.spmm(a, X, addend=Ac, addend_map=gmap, S=S, scale=1.0)
        out.append(S.clone())
        p, m, v = A.clone(), t.zeros(n, d, device=DEV), t.zeros(n, d, device=DEV)
        ops.spmm(a, X, addend=Ac, addend_map=gmap, adam=dict(p=p, m=m, v=v, **hyp))
        out += [p, m, v]
        return out

    want = run(None, persistent=False)                     # the plain one-workgroup-per-8-rows launch
    assert t.allclose(want[0].cpu(), _oracle_spmm(a, X), atol=2e-4, rtol=1e-4)
    for hot in (None, (n_u, 64), (n_u, n_i), (n_u + n_i - 5, 5), (n_u + 17, 1), (0, 300), (n_u - 8, 40)):
        got = run(hot)
        for w, h in zip(want, got):
            assert t.equal(w, h), hot
    a.hot = (n - 2, 3)                                     # past the end of X: refused by the wrapper
    with pytest.raises(ValueError):
        ops.spmm(a, X, Y=t.empty(n, d, device=DEV))


def test_spmm_halves_on_two_streams_equal_the_whole_product():
    """mi_spmm_ex.parts: short rows and split rows write disjoint output rows; enqueued on two streams (ops.SPMM_TWO_STREAMS)
    they give the bits of the single call."""
    ops = _ops()
    n, d = 600, 128
    rows, cols = _hub_graph(seed=21)
    a = _csr_with_vals(rows, cols, n, n, seed=22)
    a.plan = ops.build_spmm_plan(a, chunk=64, band=37)
    g = t.Generator().manual_seed(5)
    X, A = t.randn(n, d, generator=g).to(DEV), t.randn(n, d, generator=g).to(DEV)
    Y0, S0 = t.empty(n, d, device=DEV), t.empty(n, d, device=DEV)
    was = ops.SPMM_TWO_STREAMS
    try:
        ops.SPMM_TWO_STREAMS = 0
        ops.spmm(a, X, Y=Y0, addend=A, S=S0, scale=0.5)
        for mode in (1, 2):                                 # the default, and the other enqueue order
            Y1, S1 = t.full((n, d), float("nan"), device=DEV), t.full((n, d), float("nan"), device=DEV)
            ops.SPMM_TWO_STREAMS = mode
            ops.spmm(a, X, Y=Y1, addend=A, S=S1, scale=0.5)
            t.cuda.synchronize()
            assert t.equal(Y0, Y1) and t.equal(S0, S1), mode
    finally:
        ops.SPMM_TWO_STREAMS = was


@pytest.mark.parametrize("band", [0, 16])
def test_spmm_plan_without_split_rows_and_with_every_row_split(band):
    """Both ends: no row longer than the chunk (the plan is empty, row_list launches still work) and every row split."""
    ops = _ops()
    n, d = 300, 64
    row, col = _rand_graph(n, n, 3000, seed=31)
    a = _csr_with_vals(row, col, n, n, seed=32)
    X = t.randn(n, d, generator=t.Generator().manual_seed(1)).to(DEV)
    want = _oracle_spmm(a, X)
    a.plan = ops.build_spmm_plan(a, chunk=256, band=band)
    assert a.plan.n_long_rows == 0 and a.plan.n_items == 0 and bool((a.plan.long_index == -1).all())
    Y = t.empty(n, d, device=DEV)
    ops.spmm(a, X, Y=Y)
    assert t.allclose(Y.cpu(), want, atol=2e-5, rtol=1e-5)
    rows = t.tensor([5, 17, 299, 0], dtype=t.int32, device=DEV)
    Ys = t.empty(4, d, device=DEV)
    ops.spmm(a, X, Y=Ys, row_list=rows)
    assert t.equal(Ys, Y[rows.long()])
    deg = (a.rowptr[1:] - a.rowptr[:-1])
    a.plan = ops.build_spmm_plan(a, chunk=1, band=band)  # every non-trivial row is a split row, one entry per work item
    assert a.plan.n_long_rows == int((deg > 1).sum()) and a.plan.n_items >= int(deg[deg > 1].sum()) // 1
    Y2 = t.full((n, d), float("nan"), device=DEV)
    ops.spmm(a, X, Y=Y2)
    assert t.allclose(Y2.cpu(), want, atol=2e-5, rtol=1e-5)


try:
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=40, deadline=None, derandomize=True)  # the same examples on every run
    @given(n_rows=st.integers(1, 200), n_cols=st.integers(1, 200), nnz=st.integers(0, 3000),
           d=st.sampled_from([4, 8, 32, 64, 100, 128, 260]), chunk=st.sampled_from([1, 7, 64, 256]),
           band=st.sampled_from([0, 1, 13, 64]), seed=st.integers(0, 2**31 - 1))
    def test_spmm_property_any_shape_any_plan(n_rows, n_cols, nnz, d, chunk, band, seed):
        """Random shapes / densities / plans (SURVEY 8c iv, on the device): plan-less and planned products agree with the
        oracle; planned results are bitwise reproducible."""
        ops = _ops()
        row, col = _rand_graph(n_rows, n_cols, nnz, seed=seed % 100000)
        a = _csr_with_vals(row, col, n_rows, n_cols, seed=seed % 1000)
        X = t.randn(n_cols, d, generator=t.Generator().manual_seed(seed % 7919)).to(DEV)
        want = _oracle_spmm(a, X)
        scale = max(1.0, float(want.abs().max()))
        Y0 = t.full((n_rows, d), float("nan"), device=DEV)
        ops.spmm(a, X, Y=Y0)
        assert float((Y0.cpu() - want).abs().max()) <= 2e-5 * scale
        a.plan = ops.build_spmm_plan(a, chunk=chunk, band=band)
        Y1, Y2 = t.full((n_rows, d), float("nan"), device=DEV), t.empty(n_rows, d, device=DEV)
        ops.spmm(a, X, Y=Y1)
        ops.spmm(a, X, Y=Y2)
        assert t.equal(Y1, Y2) and float((Y1.cpu() - want).abs().max()) <= 2e-5 * scale
except ImportError:  # pragma: no cover
    pass


def test_spmm_epilogue_forms_and_strides():
    ops = _ops()
    n, d = 1500, 64
    row, col = _rand_graph(n, n, 30000, seed=21)
    a = _csr_with_vals(row, col, n, n, seed=22)
    big = t.randn(n, 2 * d + 8, device=DEV)
    X = big[:, 4:4 + d]  # 16-byte aligned start, leading dimension 2d+8
    add = t.randn(n, d, device=DEV)
    Y, S = t.empty(n, d, device=DEV), t.empty(n, d, device=DEV)
    ops.spmm(a, X, Y=Y, addend=add, S=S, scale=0.25)
    want = _oracle_spmm(a, X.contiguous())
    assert t.allclose(Y.cpu(), want, atol=2e-5, rtol=1e-5)
    assert t.allclose(S.cpu(), 0.25 * (add.cpu() + want), atol=2e-5, rtol=1e-5)
    # in-place running sum: S aliases addend
    run = add.clone()
    ops.spmm(a, X, addend=run, S=run, scale=1.0)
    assert t.allclose(run.cpu(), add.cpu() + want, atol=2e-5, rtol=1e-5)
    # S without addend
    S2 = t.empty(n, d, device=DEV)
    ops.spmm(a, X, S=S2, scale=2.0)
    assert t.allclose(S2.cpu(), 2.0 * want, atol=4e-5, rtol=1e-5)


def test_spmm_rejects_bad_arguments():
    ops = _ops()
    from laplace_amd._lib import MiError
    row, col = _rand_graph(10, 10, 30, seed=1)
    a = _csr_with_vals(row, col, 10, 10, seed=1)
    X = t.randn(10, 8, device=DEV)
    with pytest.raises(MiError):
        ops.spmm(a, X, Y=X)  # aliasing
    with pytest.raises(ValueError):
        ops.spmm(a, t.randn(9, 8, device=DEV), Y=t.empty(10, 8, device=DEV))
    with pytest.raises(MiError):
        ops.spmm(a, t.randn(10, 6, device=DEV), Y=t.empty(10, 6, device=DEV))  # d % 4 != 0


def test_spmm_empty_adjacency():
    ops = _ops()
    a = ops.coo_to_csr(t.empty(0, dtype=t.int64, device=DEV), t.empty(0, dtype=t.int64, device=DEV), 6, 6)
    a.val = t.empty(0, device=DEV)
    X = t.randn(6, 16, device=DEV)
    S = t.empty(6, 16, device=DEV)
    ops.spmm(a, X, addend=X, S=S, scale=0.5)
    assert t.allclose(S, 0.5 * X)
    for band in (0, 3):  # a plan of nothing: no split rows, and a banded plan does not need columns to look at
        a.plan = ops.build_spmm_plan(a, chunk=4, band=band)
        assert a.plan.n_long_rows == 0
        S2 = t.full((6, 16), float("nan"), device=DEV)
        ops.spmm(a, X, addend=X, S=S2, scale=0.5)
        assert t.equal(S2, S)


# ---------------------------------------------------------------------------- LightGCN.forward
def _model_and_graph(U, I, E, D, K, seed, compat):
    from laplace_amd.interactions import Interactions
    from laplace_amd.model.lightgcn import LightGCN
    g = t.Generator().manual_seed(seed)
    ei = t.stack([t.randint(0, U, (E,), generator=g), t.randint(0, I, (E,), generator=g)])
    t.manual_seed(seed)
    model = LightGCN(U, I, embedding_dim=D, num_iterations=K)
    inter = Interactions(ei, U, I)
    adj = inter.adjacency(compat)
    return model, inter, adj, ei


@pytest.mark.parametrize("compat", ["reference", "bipartite"])
@pytest.mark.parametrize("D,K", [(64, 2), (32, 4), (128, 3), (64, 1), (64, 0)])
def test_lightgcn_forward_parity(compat, D, K):
    U, I, E = 943, 1682, 20000
    model, inter, adj, ei = _model_and_graph(U, I, E, D, K, seed=7, compat=compat)
    uw, iw = model.users_emb.weight.detach().clone(), model.items_emb.weight.detach().clone()
    model.to(DEV)
    outs = model(adj.to(DEV))
    row, col, _ = adj.coo()
    want = R.lightgcn_forward(uw, iw, row, col, K)
    for got, ref in zip(outs, want):
        assert (got.detach().cpu() - ref).abs().max() <= 1e-4  # north_star tolerance
        assert t.allclose(got.detach().cpu(), ref, atol=2e-6)    # what fp32 actually gives
    assert outs[1] is model.users_emb.weight and outs[3] is model.items_emb.weight
    if compat == "reference":  # item rows never receive messages
        assert t.allclose(outs[2].detach().cpu(), iw / (K + 1), atol=1e-7)


def test_lightgcn_forward_add_self_loops():
    model, inter, adj, ei = _model_and_graph(50, 70, 600, 32, 2, seed=9, compat="bipartite")
    model.add_self_loops = True
    uw, iw = model.users_emb.weight.detach().clone(), model.items_emb.weight.detach().clone()
    model.to(DEV)
    uf, _, itf, _ = model(adj.to(DEV))
    row, col, _ = adj.coo()
    wu, _, wi, _ = R.lightgcn_forward(uw, iw, row, col, 2, add_self_loops=True)
    assert t.allclose(uf.detach().cpu(), wu, atol=2e-6) and t.allclose(itf.detach().cpu(), wi, atol=2e-6)


@pytest.mark.parametrize("compat", ["reference", "bipartite"])
def test_lightgcn_autograd_backward_parity(compat):
    """loss.backward() through the HIP propagate == torch autograd through the oracle."""
    U, I, E, D, K = 300, 500, 6000, 64, 3
    model, inter, adj, ei = _model_and_graph(U, I, E, D, K, seed=13, compat=compat)
    uw = model.users_emb.weight.detach().clone().requires_grad_(True)
    iw = model.items_emb.weight.detach().clone().requires_grad_(True)
    g = t.Generator().manual_seed(5)
    B = 256
    ui, pi, ni = (t.randint(0, U, (B,), generator=g), t.randint(0, I, (B,), generator=g),
                  t.randint(0, I, (B,), generator=g))
    row, col, _ = adj.coo()
    uf, u0, itf, i0 = R.lightgcn_forward(uw, iw, row, col, K)
    loss_ref = R.bpr_loss(uf[ui], u0[ui], itf[pi], i0[pi], itf[ni], i0[ni], 1e-3)
    loss_ref.backward()

    from laplace_amd.utils.metrics_lightgcn import bpr_loss
    model.to(DEV)
    uf, u0, itf, i0 = model(adj.to(DEV))
    uid, pid, nid = ui.to(DEV), pi.to(DEV), ni.to(DEV)
    loss = bpr_loss(uf[uid], u0[uid], itf[pid], i0[pid], itf[nid], i0[nid], 1e-3)
    loss.backward()
    assert abs(float(loss) - float(loss_ref)) < 1e-6
    assert t.allclose(model.users_emb.weight.grad.cpu(), uw.grad, atol=1e-7, rtol=1e-4)
    assert t.allclose(model.items_emb.weight.grad.cpu(), iw.grad, atol=1e-7, rtol=1e-4)


# ---------------------------------------------------------------------------- K9: sampler
def test_sampler_bit_exact_vs_philox_oracle():
    ops = _ops()
    from laplace_amd.interactions import Interactions
    U, I, E = 200, 90, 3000
    g = t.Generator().manual_seed(3)
    keys = t.randperm(U * I, generator=g)[:E]
    ei = t.stack([keys // I, keys % I])
    inter = Interactions(ei.to(DEV), U, I)
    r = inter.csr()
    roe = inter.row_of_edge()
    rowptr, col_s, _ = R.sparse_tensor_csr(ei[0], ei[1], U, I)
    assert t.equal(r.rowptr.cpu().long(), rowptr) and t.equal(r.col.cpu().long(), col_s)
    assert t.equal(roe.cpu().long(), t.repeat_interleave(t.arange(U), rowptr[1:] - rowptr[:-1]))
    neg_range = int(ei[1].max())  # reference: num_nodes = max(col)
    for quirk, nsl in ((False, False), (True, False), (False, True), (True, True)):
        for step in (0, 1, 12345678901):
            us, ps, ns = ops.sample_bpr_batch(r, roe, 512, neg_range, seed=0xDEADBEEFCAFE, step=step, quirk=quirk,
                                              no_self_loops=nsl)
            wu, wp, wn = R.sample_bpr_batch_philox(rowptr, col_s, 512, neg_range, seed=0xDEADBEEFCAFE, step=step,
                                                   quirk=quirk, no_self_loops=nsl)
            assert t.equal(us.cpu(), wu) and t.equal(ps.cpu(), wp) and t.equal(ns.cpu(), wn)
            assert not nsl or not bool((ns == us).any())  # contains_neg_self_loops=False: never item id == user id
    # the flag bites: without it users < neg_range do draw their own id now and then
    us, _, ns = ops.sample_bpr_batch(r, roe, 20000, neg_range, seed=4, step=0)
    assert bool((ns == us).any())
    # structural properties on a bigger draw
    us, ps, ns = ops.sample_bpr_batch(r, roe, 20000, neg_range, seed=1, step=2)
    us, ps, ns = us.cpu(), ps.cpu(), ns.cpu()
    pos_keys = set((ei[0] * I + ei[1]).tolist())
    assert all(k in pos_keys for k in (us * I + ps).tolist())
    assert not any(k in pos_keys for k in (us * I + ns).tolist())
    assert int(ns.min()) >= 0 and int(ns.max()) < neg_range


# ---------------------------------------------------------------------------- a7/a8: BPR
def test_bpr_kernel_matches_reference_golden(golden_dir):
    """Loss and gradients of the fused kernel vs the reference's bpr_loss + autograd (golden)."""
    ops = _ops()
    for case in t.load(os.path.join(golden_dir, "bpr_loss.pt"), weights_only=False):
        uf, u0, pf, p0, nf, n0 = case["inputs"]
        B, D = uf.shape
        if D % 4:
            continue
        # lay the six gathered blocks out as tables: users rows [0,B), pos items [B,2B), neg [2B,3B)
        final = t.cat([uf, pf, nf]).to(DEV)
        e0 = t.cat([u0, p0, n0]).to(DEV)
        users = t.arange(B, device=DEV)
        pos = t.arange(B, device=DEV)          # item ids -> rows n_users + id
        neg = t.arange(B, 2 * B, device=DEV)
        g_final = t.zeros(3 * B, D, device=DEV)
        reg_w = t.zeros(3 * B, device=DEV)
        loss = ops.bpr_fwd_bwd(users, pos, neg, final, e0, B, case["lambda"], g_final=g_final, reg_w=reg_w)
        assert abs(float(loss) - float(case["loss"])) <= 1e-6 * max(1.0, abs(float(case["loss"])))
        gu, g_u0, gp, g_p0, gn, g_n0 = case["grads"]
        want_final = t.cat([gu, gp, gn])
        assert t.allclose(g_final.cpu(), want_final, atol=1e-7, rtol=2e-5)
        want_e0 = t.cat([g_u0, g_p0, g_n0])
        assert t.allclose((reg_w[:, None] * e0).cpu(), want_e0, atol=1e-9, rtol=2e-5)


def test_bpr_kernel_repeated_nodes_accumulate():
    ops = _ops()
    U, I, D, B = 5, 6, 32, 64
    g = t.Generator().manual_seed(1)
    final = t.randn(U + I, D, generator=g).requires_grad_(True)
    e0 = t.randn(U + I, D, generator=g).requires_grad_(True)
    users, pos, neg = (t.randint(0, U, (B,), generator=g), t.randint(0, I, (B,), generator=g),
                       t.randint(0, I, (B,), generator=g))
    loss_ref = R.bpr_loss(final[users], e0[users], final[U + pos], e0[U + pos], final[U + neg], e0[U + neg], 1e-2)
    loss_ref.backward()
    gf = t.zeros(U + I, D, device=DEV)
    rw = t.zeros(U + I, device=DEV)
    loss = ops.bpr_fwd_bwd(users.to(DEV), pos.to(DEV), neg.to(DEV), final.detach().to(DEV), e0.detach().to(DEV), U,
                           1e-2, g_final=gf, reg_w=rw, g_scale=0.5)
    assert abs(float(loss) - float(loss_ref)) < 1e-5
    assert t.allclose(gf.cpu(), 0.5 * final.grad, atol=1e-6, rtol=1e-4)
    assert t.allclose((rw[:, None].cpu() * e0.detach()), e0.grad, atol=1e-7, rtol=1e-5)


# ---------------------------------------------------------------------------- a9: Adam
def test_adam_matches_torch_optim():
    ops = _ops()
    n, d = 257, 64
    g = t.Generator().manual_seed(2)
    p_ref = t.nn.Parameter(t.randn(n, d, generator=g) * 0.1)
    opt = t.optim.Adam([p_ref], lr=1e-3)
    p = p_ref.detach().clone().to(DEV)
    m, v = t.zeros_like(p), t.zeros_like(p)
    for step in range(1, 8):
        grad = t.randn(n, d, generator=g) * (10.0 ** -(step % 4))
        grad[step::7] = 0.0  # rows with exactly zero gradient must not move
        p_ref.grad = grad.clone()
        opt.step()
        ops.adam_step(p, grad.to(DEV), m, v, step=step, lr=1e-3)
        assert t.allclose(p.cpu(), p_ref.detach(), atol=2e-7, rtol=1e-6), step
    st = opt.state[p_ref]
    assert t.allclose(m.cpu(), st["exp_avg"], atol=1e-8, rtol=1e-5)
    assert t.allclose(v.cpu(), st["exp_avg_sq"], atol=1e-12, rtol=1e-5)


# ---------------------------------------------------------------------------- whole train step
@pytest.mark.parametrize("compat", ["reference", "bipartite"])
def test_trainer_ten_steps_match_reference_loop(compat):
    """10 iterations of run_pipeline_lightgcn.py:117-159 (same batches): fused HIP step vs the
    oracle's forward/bpr/backward/torch.optim.Adam.  Embeddings within 1e-4 (north_star)."""
    from laplace_amd.trainer import LightGCNTrainer
    U, I, E, D, K, B = 400, 600, 8000, 64, 3, 128
    model, inter, adj, ei = _model_and_graph(U, I, E, D, K, seed=21, compat=compat)
    uw = t.nn.Parameter(model.users_emb.weight.detach().clone())
    iw = t.nn.Parameter(model.items_emb.weight.detach().clone())
    opt = t.optim.Adam([uw, iw], lr=1e-3)
    sched = t.optim.lr_scheduler.ExponentialLR(opt, gamma=0.95)
    row, col, _ = adj.coo()
    model.to(DEV)
    tr = LightGCNTrainer(model, adj.to(DEV), inter.to(DEV), lr=1e-3, Lambda=1e-6, batch_size=B, seed=5)
    g = t.Generator().manual_seed(99)
    for it in range(10):
        batch = (t.randint(0, U, (B,), generator=g), t.randint(0, I, (B,), generator=g),
                 t.randint(0, I, (B,), generator=g))
        loss_ref = R.train_step(uw, iw, opt, row, col, K, batch, 1e-6)
        loss = tr.step(tuple(x.to(DEV) for x in batch))
        assert abs(float(loss) - loss_ref) < 1e-5, it
        if it % 4 == 0 and it != 0:
            sched.step()
            tr.decay_lr(0.95)
    assert (model.users_emb.weight.detach().cpu() - uw.detach()).abs().max() <= 1e-4
    assert (model.items_emb.weight.detach().cpu() - iw.detach()).abs().max() <= 1e-4
    uf, _, itf, _ = model(adj.to(DEV))
    wu, _, wi, _ = R.lightgcn_forward(uw.detach(), iw.detach(), row, col, K)
    assert (uf.detach().cpu() - wu).abs().max() <= 1e-4 and (itf.detach().cpu() - wi).abs().max() <= 1e-4


@pytest.mark.parametrize("sparse_batch", [False, True])
def test_trainer_under_locality_order_matches_reference_loop(sparse_batch):
    """reorder=True: the step runs on relabelled nodes (items by popularity, users by coldest item) with the model's
    table permuted in place; batches go in and parameters come out under ORIGINAL ids, equal to the oracle's loop on
    the unrelabelled graph within the same 1e-4 (only the summation order inside a row changes)."""
    from laplace_amd.trainer import LightGCNTrainer
    U, I, E, D, K, B = 400, 600, 8000, 64, 3, 128
    model, inter, adj, ei = _model_and_graph(U, I, E, D, K, seed=21, compat="bipartite")
    uw = t.nn.Parameter(model.users_emb.weight.detach().clone())
    iw = t.nn.Parameter(model.items_emb.weight.detach().clone())
    opt = t.optim.Adam([uw, iw], lr=1e-3)
    row, col, _ = adj.coo()
    model.to(DEV)
    tr = LightGCNTrainer(model, adj.to(DEV), inter.to(DEV), lr=1e-3, Lambda=1e-6, batch_size=B, seed=5,
                         sparse_batch=sparse_batch, reorder=True)
    o = tr.order
    # the order: a permutation; items by popularity, users by their coldest item
    assert t.equal(o.user_new_of_old[o.user_old_of_new].cpu(), t.arange(U)) and t.equal(o.item_new_of_old[o.item_old_of_new].cpu(), t.arange(I))
    deg_new = t.bincount(o.item_new_of_old.cpu()[ei[1]], minlength=I)
    assert bool((deg_new[:-1] >= deg_new[1:]).all())
    # the table moved with it, and comes back
    assert tr.in_training_order and t.equal(model.users_emb.weight.detach().cpu()[o.user_new_of_old.cpu()], uw.detach())
    tr.to_original_order()
    assert t.equal(model.users_emb.weight.detach().cpu(), uw.detach()) and t.equal(model.items_emb.weight.detach().cpu(), iw.detach())
    g = t.Generator().manual_seed(99)
    for it in range(10):
        batch = (t.randint(0, U, (B,), generator=g), t.randint(0, I, (B,), generator=g),
                 t.randint(0, I, (B,), generator=g))
        loss_ref = R.train_step(uw, iw, opt, row, col, K, batch, 1e-6)
        loss = tr.step(tuple(x.to(DEV) for x in batch))
        assert abs(float(loss) - loss_ref) < 1e-5, it
        if it == 4:  # an evaluation in the middle of training reads the tables by original id
            tr.to_original_order()
            assert (model.users_emb.weight.detach().cpu() - uw.detach()).abs().max() <= 1e-4
    # forward in training order == oracle forward, row for row
    wu, _, wi, _ = R.lightgcn_forward(uw.detach(), iw.detach(), row, col, K)
    fin = tr.forward()[tr.order.node_new_of_old()].cpu()
    assert (fin[:U] - wu).abs().max() <= 1e-4 and (fin[U:] - wi).abs().max() <= 1e-4
    tr.finish()
    assert (model.users_emb.weight.detach().cpu() - uw.detach()).abs().max() <= 1e-4
    assert (model.items_emb.weight.detach().cpu() - iw.detach()).abs().max() <= 1e-4
    # sample() speaks original ids: positives are edges of the graph as given, negatives are not
    us, ps, ns = (x.cpu() for x in tr.sample())
    keys = set((ei[0] * I + ei[1]).tolist())
    assert all(k in keys for k in (us * I + ps).tolist()) and not any(k in keys for k in (us * I + ns).tolist())
    with pytest.raises(ValueError):
        LightGCNTrainer(model, adj.to(DEV), inter.to(DEV), lr=1e-3, Lambda=1e-6, batch_size=B, reorder=True, neg_range=I - 1)


def test_trainer_on_device_sampling_trains():
    """With the on-device sampler the BPR objective (unbounded below) goes down."""
    from laplace_amd.trainer import LightGCNTrainer
    from laplace_amd.interactions import Interactions
    from laplace_amd.model.lightgcn import LightGCN
    from laplace_amd import synthetic as S
    spec = S.SyntheticSpec(2000, 500, 30000, seed=8)
    ei = S.generate(spec)
    t.manual_seed(0)
    model = LightGCN(spec.num_users, spec.num_items, 64, 3).to(DEV)
    inter = Interactions(ei.to(DEV), spec.num_users, spec.num_items)
    tr = LightGCNTrainer(model, inter.adjacency("bipartite"), inter, lr=1e-2, Lambda=1e-6, batch_size=1024, seed=1)
    losses = [float(tr.step()) for _ in range(60)]
    assert np.isfinite(losses).all()
    assert np.mean(losses[-10:]) < np.mean(losses[:10]) - 0.05
    assert tr.step_count == 60


# ---------------------------------------------------------------------------- §8e sharded path, HIP ops
@pytest.mark.parametrize("sparse_batch", [False, True])
def test_sharded_trainer_hip_ops_single_rank_equals_plain_trainer(sparse_batch):
    """The row-sliced (user rows / item rows) launches of the sharded step give the plain step's result."""
    from laplace_amd.dist import ShardedLightGCNTrainer
    from laplace_amd.trainer import LightGCNTrainer
    U, I, E, D, K, B = 700, 300, 9000, 64, 3, 256
    model_a, inter, adj, ei = _model_and_graph(U, I, E, D, K, seed=31, compat="bipartite")
    model_b, _, _, _ = _model_and_graph(U, I, E, D, K, seed=31, compat="bipartite")
    model_a.to(DEV)
    model_b.to(DEV)
    inter = inter.to(DEV)
    plain = LightGCNTrainer(model_a, adj.to(DEV), inter, lr=1e-3, Lambda=1e-5, batch_size=B, seed=2, sparse_batch=False)
    shard = ShardedLightGCNTrainer(model_b, inter, lr=1e-3, Lambda=1e-5, batch_size=B, seed=2, sparse_batch=sparse_batch)
    assert t.allclose(shard.adj_fwd.val, plain.adj_fwd.val, rtol=1e-6, atol=0)
    for _ in range(5):
        la, lb = plain.step(), shard.step()
        assert abs(float(la) - float(lb)) < 1e-6
    assert t.equal(plain.batch_idx[0], shard.batch_idx[0]) and t.equal(plain.batch_idx[2], shard.batch_idx[2])
    assert t.allclose(plain.table, shard.table, atol=1e-6)
    assert t.allclose(plain.forward(), shard.forward(), atol=1e-6)


# ---------------------------------------------------------------------------- full size (C2) properties
@pytest.fixture(scope="module")
def c2_graph():
    """BASELINE.json configs[1] / SURVEY C2: 1M users x 100K items, 10M edges, symmetric nnz = 20M."""
    from laplace_amd import synthetic as S
    from laplace_amd.interactions import Interactions
    ei = S.generate(S.C2)
    inter = Interactions(ei.to(DEV), S.C2.num_users, S.C2.num_items)
    adj, _ = inter.adjacency("bipartite").gcn_normalized(False)
    return ei, inter, adj


S_C2_USERS = 1_000_000


def test_full_size_propagate_properties(c2_graph):
    """Size-independent properties at the benchmark's full size, D=128:
    linearity, the D^1/2 eigenvector of the normalised adjacency, run-to-run bit stability, and a
    sample of rows (hubs included) against the oracle."""
    ops = _ops()
    ei, inter, adj = c2_graph
    n, d = adj.n_rows, 128
    assert adj.nnz == 20_000_000 and n == 1_100_000
    assert adj.plan.n_long_rows > 1000 and adj.plan.n_items > 20_000  # the split-row path carries real weight
    g = t.Generator(device=DEV).manual_seed(0)
    X = t.randn(n, d, device=DEV, generator=g) * 0.1
    Z = t.randn(n, d, device=DEV, generator=g) * 0.1
    yx, yz, yl = t.empty(n, d, device=DEV), t.empty(n, d, device=DEV), t.empty(n, d, device=DEV)
    ops.spmm(adj, X, Y=yx)
    ops.spmm(adj, Z, Y=yz)
    ops.spmm(adj, 2.0 * X - 0.5 * Z, Y=yl)
    assert (yl - (2.0 * yx - 0.5 * yz)).abs().max() <= 2e-5
    y2 = t.empty(n, d, device=DEV)
    ops.spmm(adj, X, Y=y2)
    assert t.equal(y2, yx)  # bitwise reproducible (no float atomics anywhere in the propagate)
    # the plain short-row launch and the persistent one with an LDS hot-row cache (ops.spmm `hot`) give the same bits,
    # whatever the cached range — first item rows, or user rows
    was = ops.PERSISTENT_ROWS
    try:
        for persistent, hot in ((False, None), (True, None), (True, (S_C2_USERS, 304)), (True, (1234, 100))):
            ops.PERSISTENT_ROWS, adj.hot = persistent, hot
            y2.fill_(float("nan"))
            ops.spmm(adj, X, Y=y2)
            assert t.equal(y2, yx), (persistent, hot)
    finally:
        ops.PERSISTENT_ROWS, adj.hot = was, None
    # A~ (D^1/2 1) = D^-1/2 A 1 = D^1/2 1 on rows with deg > 0
    deg = (adj.rowptr[1:] - adj.rowptr[:-1]).float()
    v = deg.sqrt()[:, None].expand(n, 4).contiguous()
    out = t.empty(n, 4, device=DEV)
    ops.spmm(adj, v, Y=out)
    assert ((out - v).abs() / v.clamp(min=1.0)).max() <= 1e-4
    # sampled rows vs the oracle's sequential fp32 sum (float64 for the hubs' error budget)
    rows = t.cat([t.randint(0, n, (2000,)), (deg.cpu().topk(20).indices)])
    rp, col, val, Xc = adj.rowptr.cpu().long(), adj.col.cpu().long(), adj.val.cpu(), X.cpu()
    for r in rows.tolist():
        b, e = int(rp[r]), int(rp[r + 1])
        want = (val[b:e, None].double() * Xc[col[b:e]].double()).sum(0)
        assert (yx[r].cpu().double() - want).abs().max() <= 1e-5 + 1e-6 * (e - b) ** 0.5, r


def test_full_size_sampler_and_step(c2_graph):
    """On-device sampler at full size: every sampled positive is an edge, no negative is; one fused
    train step leaves rows without gradient untouched and moves the sampled ones."""
    ops = _ops()
    from laplace_amd.model.lightgcn import LightGCN
    from laplace_amd.trainer import LightGCNTrainer
    from laplace_amd import synthetic as S
    ei, inter, adj = c2_graph
    U, I = S.C2.num_users, S.C2.num_items
    us, ps, ns = ops.sample_bpr_batch(inter.csr(), inter.row_of_edge(), 16384, I, seed=3, step=11)
    keys = t.sort(ei[0].to(DEV) * I + ei[1].to(DEV))[0]
    def member(k):
        pos = t.searchsorted(keys, k).clamp(max=keys.numel() - 1)
        return keys[pos] == k
    assert bool(member(us * I + ps).all()) and not bool(member(us * I + ns).any())
    t.manual_seed(0)
    model = LightGCN(U, I, 128, 3).to(DEV)
    before = model.table().clone()
    tr = LightGCNTrainer(model, inter.adjacency("bipartite"), inter, lr=1e-3, Lambda=1e-6, batch_size=16384, seed=3)
    assert tr.order is not None  # at this size the step runs under the locality order by default
    loss = float(tr.step())
    tr.finish()                  # rows back under their original ids
    # -softplus(~0) + lambda * 3 * B * D * 0.1^2 at initialisation
    assert abs(loss - (-0.6931 + 1e-6 * 3 * 16384 * 128 * 0.01)) < 0.01
    moved = (model.table() - before).abs().amax(dim=1) > 0
    assert 0.5 < float(moved.float().mean()) <= 1.0  # 3-hop receptive field of a 16K batch covers most nodes
    assert float((model.table() - before).abs().max()) <= 1.001e-3  # |Adam step| <= lr


# ---------------------------------------------------------------------------- sparse-operand SpMM forms
def test_spmm_sparse_operand_forms_equal_dense_bitwise():
    """x_map (compact X, zero rows skipped), addend_map (compact addend) and row_list (row subset, compact
    outputs, hub rows through the split path) give exactly the dense call's numbers."""
    ops = _ops()
    n, d = 4000, 128
    g = t.Generator().manual_seed(3)
    hub = t.stack([t.full((30_000,), 11), t.randint(0, n, (30_000,), generator=g)])
    rest = t.stack(_rand_graph(n, n, 40_000, seed=4))
    ei = t.cat([hub, rest], dim=1)
    a = _csr_with_vals(ei[0], ei[1], n, n, seed=5)
    a.plan = ops.build_spmm_plan(a, chunk=256)
    assert a.plan.n_long_rows >= 1
    # a sparse-row operand: 300 non-zero rows, incl. row 11 (the hub) and some of its neighbours
    nz = t.unique(t.cat([t.tensor([11]), t.randint(0, n, (299,), generator=g)]))
    m = nz.numel()
    gmap = t.full((n,), -1, dtype=t.int32)
    gmap[nz] = t.arange(m, dtype=t.int32)
    Xc = t.randn(m, d, generator=g)
    Xd = t.zeros(n, d)
    Xd[nz] = Xc
    Xc_g, Xd_g, gmap_g = Xc.to(DEV), Xd.to(DEV), gmap.to(DEV)
    dense, sparse = t.empty(n, d, device=DEV), t.empty(n, d, device=DEV)
    ops.spmm(a, Xd_g, addend=Xd_g, S=dense, scale=0.5)
    ops.spmm(a, Xc_g, addend=Xc_g, S=sparse, scale=0.5, x_map=gmap_g, addend_map=gmap_g)
    assert t.equal(dense, sparse)
    # row subset: every row of nz, compact outputs, actual count given on device, launch bound larger
    full = t.empty(n, d, device=DEV)
    W = t.randn(n, d, generator=g).to(DEV)
    ops.spmm(a, W, addend=W, S=full, scale=0.25)
    rows = t.zeros(m + 50, dtype=t.int32)
    rows[:m] = nz.to(t.int32)
    cnt = t.tensor([m], dtype=t.int32, device=DEV)
    sub = t.full((m + 50, d), float("nan"), device=DEV)
    ops.spmm(a, W, addend=W[nz.to(DEV)].contiguous().new_zeros(m + 50, d).index_copy_(0, t.arange(m, device=DEV), W[nz.to(DEV)]),
             S=sub, scale=0.25, row_list=rows.to(DEV), n_list_dev=cnt)
    assert t.equal(sub[:m], full[nz.to(DEV)])
    assert bool(t.isnan(sub[m:]).all())  # rows beyond the device count are not touched


def test_batch_nodes_and_gather_rows():
    ops = _ops()
    U, I, B = 50, 40, 64
    g = t.Generator().manual_seed(1)
    u, p, n_ = t.randint(0, U, (B,), generator=g), t.randint(0, I, (B,), generator=g), t.randint(0, I, (B,), generator=g)
    gmap, nodes, cnt = ops.batch_nodes(u.to(DEV), p.to(DEV), n_.to(DEV), U, U + I)
    want = t.unique(t.cat([u, U + p, U + n_]))
    c = int(cnt[0])
    assert int(cnt[1]) == int((want < U).sum())
    assert c == want.numel() and t.equal(nodes[:c].cpu().long(), want)
    gm = gmap.cpu().long()
    assert t.equal(gm[want], t.arange(c)) and int((gm >= 0).sum()) == c
    src = t.randn(U + I, 32, generator=g).to(DEV)
    dst = t.zeros(3 * B, 32, device=DEV)
    ops.gather_rows(dst, src, nodes, cnt[:1])
    ops.gather_rows(dst, src, nodes, cnt[:1], accumulate=True)
    assert t.equal(dst[:c], 2 * src[want.to(DEV)]) and float(dst[c:].abs().max()) == 0.0
    # item part only, against an item-only table, scaled; and the inverse scatter
    items = src[U:].contiguous()
    part = t.zeros(3 * B, 32, device=DEV)
    ops.gather_rows(part, items, nodes, cnt[:1], scale=0.5, begin_dev=cnt[1:], row_offset=U)
    cu = int(cnt[1])
    assert float(part[:cu].abs().max()) == 0.0 and t.equal(part[cu:c], 0.5 * src[want[cu:].to(DEV)])
    back = t.zeros(I, 32, device=DEV)
    ops.scatter_rows(back, part, nodes, cnt[:1], begin_dev=cnt[1:], row_offset=U)
    assert t.equal(back[(want[cu:] - U).to(DEV)], part[cu:c]) and int((back.abs().sum(1) > 0).sum()) <= c - cu


@pytest.mark.parametrize("K", [0, 1, 2, 3, 4])
def test_sparse_batch_step_equals_plain_step(K):
    """The byte-saving step and the straightforward step are the same computation."""
    from laplace_amd.trainer import LightGCNTrainer
    U, I, E, D, B = 900, 500, 15000, 64, 512
    ma, inter, adj, ei = _model_and_graph(U, I, E, D, K, seed=41, compat="bipartite")
    mb, _, _, _ = _model_and_graph(U, I, E, D, K, seed=41, compat="bipartite")
    ma.to(DEV); mb.to(DEV)
    inter = inter.to(DEV)
    adj = adj.to(DEV)
    plain = LightGCNTrainer(ma, adj, inter, lr=1e-3, Lambda=1e-4, batch_size=B, seed=9, sparse_batch=False)
    fast = LightGCNTrainer(mb, adj, inter, lr=1e-3, Lambda=1e-4, batch_size=B, seed=9, sparse_batch=True)
    for it in range(6):
        la, lb = float(plain.step()), float(fast.step())
        assert abs(la - lb) <= 1e-6, it
        assert t.equal(plain.batch_idx[0], fast.batch_idx[0]) and t.equal(plain.batch_idx[2], fast.batch_idx[2])
    # the final-layer sum is associated differently in the two forms (running sum over all rows vs gathers of the
    # batch rows), so the match is to rounding, not to the bit
    assert (plain.table - fast.table).abs().max() <= 2e-6


@pytest.mark.parametrize("K", [1, 3])
def test_adam_in_the_backward_epilogue_equals_separate_adam(K):
    """fuse_adam: the Adam update in the last backward product's epilogue vs the separate mi_adam_dense_f32 pass."""
    from laplace_amd.trainer import LightGCNTrainer
    U, I, E, D, B = 900, 500, 15000, 64, 512
    ma, inter, adj, ei = _model_and_graph(U, I, E, D, K, seed=43, compat="bipartite")
    mb, _, _, _ = _model_and_graph(U, I, E, D, K, seed=43, compat="bipartite")
    ma.to(DEV); mb.to(DEV)
    inter, adj = inter.to(DEV), adj.to(DEV)
    sep = LightGCNTrainer(ma, adj, inter, lr=1e-3, Lambda=1e-4, batch_size=B, seed=9, fuse_adam=False)
    fused = LightGCNTrainer(mb, adj, inter, lr=1e-3, Lambda=1e-4, batch_size=B, seed=9, fuse_adam=True)
    for it in range(6):
        la, lb = float(sep.step()), float(fused.step())
        assert abs(la - lb) <= 1e-6, it
    assert sep.step_count == fused.step_count == 6
    # the same arithmetic in the same order (no float atomics anywhere): identical bits
    assert t.equal(sep.table, fused.table) and t.equal(sep.m, fused.m) and t.equal(sep.v, fused.v)


def test_forward_accepts_torch_sparse_tensors_and_torch_sparse_like_objects():
    """`edge_index` may be this repo's SparseTensor, a torch sparse CSR / COO tensor, or an object with
    torch_sparse's .coo() / .sparse_sizes(); all give the same four outputs."""
    from laplace_amd.model.lightgcn import LightGCN
    from laplace_amd.sparse import SparseTensor
    U, I, D, K = 70, 50, 32, 2
    n = U + I
    g = t.Generator().manual_seed(3)
    u, i = t.randint(0, U, (600,), generator=g), t.randint(0, I, (600,), generator=g)
    row, col = t.cat([u, i + U]).to(DEV), t.cat([i + U, u]).to(DEV)   # duplicates on purpose
    t.manual_seed(0)
    model = LightGCN(U, I, D, K).to(DEV)
    want = [x.detach().clone() for x in model(SparseTensor(row=row, col=col, sparse_sizes=(n, n)))]
    coo = t.sparse_coo_tensor(t.stack([row, col]), t.ones(row.numel(), device=DEV), (n, n))
    order = t.argsort(row * n + col)
    crow = t.zeros(n + 1, dtype=t.int64, device=DEV)
    crow[1:] = t.cumsum(t.bincount(row, minlength=n), 0)
    csr = t.sparse_csr_tensor(crow, col[order], t.ones(row.numel(), device=DEV), (n, n))

    class Foreign:  # the two methods of torch_sparse.SparseTensor the conversion uses
        def coo(self):
            return row, col, None

        def sparse_sizes(self):
            return (n, n)

    for adj in (coo, csr, Foreign()):
        got = model(adj)
        assert all(t.equal(a.detach(), b) for a, b in zip(got, want))
    foreign = Foreign()
    model(foreign)
    assert foreign._laplace_adj is not None  # converted once
    with pytest.raises(ValueError):
        model(t.sparse_coo_tensor(t.stack([row, col]), t.full((row.numel(),), 2.0, device=DEV), (n, n)))
    with pytest.raises(TypeError):
        model(t.stack([row, col]))


@pytest.mark.parametrize("D", [64, 128, 200])
def test_train_steps_are_bitwise_reproducible(D):
    """No float atomics anywhere in the step: two runs from the same state give identical bits, also with batches full of
    repeated nodes (a tiny item set) and references that cross the 64-reference chunks of the gradient kernel."""
    from laplace_amd.trainer import LightGCNTrainer
    U, I, E, K, B = 500, 40, 6000, 2, 2048   # 40 items for 4 096 item references: runs of ~100 per row
    tables = []
    for run in range(2):
        m, inter, adj, _ = _model_and_graph(U, I, E, D, K, seed=47, compat="bipartite")
        m.to(DEV)
        tr = LightGCNTrainer(m, adj.to(DEV), inter.to(DEV), lr=1e-2, Lambda=1e-4, batch_size=B, seed=5)
        losses = [float(tr.step()) for _ in range(5)]
        tables.append((tr.table.clone(), tr.m.clone(), tr.v.clone(), losses))
    assert tables[0][3] == tables[1][3]
    assert t.equal(tables[0][0], tables[1][0]) and t.equal(tables[0][1], tables[1][1]) and t.equal(tables[0][2], tables[1][2])


def test_bpr_gradient_runs_across_chunks_match_float64():
    """One item is the positive of every sample (a run of B references over B / 64 chunks), users repeat: the chunked,
    ordered reduction against a float64 scatter."""
    ops = _ops()
    B, D, U, I = 1000, 128, 50, 30
    g = t.Generator().manual_seed(2)
    users = t.randint(0, U, (B,), generator=g)
    pos = t.full((B,), 7, dtype=t.int64)
    neg = t.randint(0, I, (B,), generator=g)
    fin = t.randn(U + I, D, generator=g)
    G = t.zeros(U + I, D, device=DEV)
    reg = t.zeros(U + I, device=DEV)
    ops.bpr_fwd_bwd(users.to(DEV), pos.to(DEV), neg.to(DEV), fin.to(DEV), fin.to(DEV), U, 0.0, g_final=G, reg_w=reg)
    f64 = fin.double().requires_grad_(True)
    x = (f64[users] * f64[U + pos]).sum(1) - (f64[users] * f64[U + neg]).sum(1)
    (-t.nn.functional.softplus(x).mean()).backward()
    assert (G.cpu().double() - f64.grad).abs().max() <= 2e-6 * float(f64.grad.abs().max())
    G2 = t.zeros_like(G)
    ops.bpr_fwd_bwd(users.to(DEV), pos.to(DEV), neg.to(DEV), fin.to(DEV), fin.to(DEV), U, 0.0, g_final=G2, reg_w=reg)
    assert t.equal(G, G2)


# ---------------------------------------------------------------------------- BASELINE.json configs[0]
@pytest.mark.parametrize("compat", ["reference", "bipartite"])
def test_c1_movielens_shaped_parity(compat):
    """C1: U=943, I=1682, E=100 000 (80/10/10 split -> 80 000 train edges), D=64, K=2, B=128, lambda=1e-6,
    lr=1e-3: forward within 1e-4 on all four outputs, 10 reference-loop iterations within 1e-4 on the
    parameters, exact top-12 / top-256 lists from the trained layer-0 embeddings."""
    from laplace_amd import synthetic as S
    from laplace_amd.data.lightgcn_loader import split
    from laplace_amd.interactions import Interactions
    from laplace_amd.model.lightgcn import LightGCN
    from laplace_amd.trainer import LightGCNTrainer
    from laplace_amd.utils.metrics_lightgcn import topk_for_users
    ei = S.generate(S.C1)
    train_e, val_e, test_e, _ = split(ei)
    assert train_e.shape[1] == 80_000 and val_e.shape[1] == 10_000 and test_e.shape[1] == 10_000
    U, I, D, K, B = 943, 1682, 64, 2, 128
    t.manual_seed(0)
    model = LightGCN(U, I, D, K)
    uw = t.nn.Parameter(model.users_emb.weight.detach().clone())
    iw = t.nn.Parameter(model.items_emb.weight.detach().clone())
    inter = Interactions(train_e, U, I)
    adj = inter.adjacency(compat)
    row, col, _ = adj.coo()
    model.to(DEV)
    outs = model(adj.to(DEV))
    want = R.lightgcn_forward(uw.detach(), iw.detach(), row, col, K)
    for got, ref in zip(outs, want):
        assert (got.detach().cpu() - ref).abs().max() <= 1e-4
    opt = t.optim.Adam([uw, iw], lr=1e-3)
    tr = LightGCNTrainer(model, adj.to(DEV), inter.to(DEV), lr=1e-3, Lambda=1e-6, batch_size=B, seed=3,
                         neg_range=int(train_e[1].max()), reference_sampler_quirks=(compat == "reference"))
    for it in range(10):
        batch = tr.sample()  # device sampler with the reference's negative range
        cpu_batch = tuple(x.cpu().clone() for x in batch)
        loss_ref = R.train_step(uw, iw, opt, row, col, K, cpu_batch, 1e-6)
        assert abs(float(tr.step(batch)) - loss_ref) < 1e-5
    assert (model.users_emb.weight.detach().cpu() - uw.detach()).abs().max() <= 1e-4
    assert (model.items_emb.weight.detach().cpu() - iw.detach()).abs().max() <= 1e-4
    # exact top-K on the SAME embeddings (the device's), as make_predictions_for_user defines it
    ue, ie = model.users_emb.weight.detach(), model.items_emb.weight.detach()
    users = t.arange(0, U, 7, device=DEV)
    scores = R.scores_fma(ue[users].cpu(), ie.cpu())
    excl_d = {}
    tu, ti = train_e[0], train_e[1]
    excl = [ti[tu == int(u)] for u in users.tolist()]
    for k in (12, 256):
        got = topk_for_users(ue, ie, users, train_e.to(DEV), k).cpu()
        assert t.equal(got, R.topk_excl_exact(scores, excl, k))

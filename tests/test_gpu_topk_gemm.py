"""GPU parity: fp32 MFMA GEMM (bitwise vs the oracle's fma chain), exact top-K with exclusion
(vs the oracle and the reference's golden outputs), evaluation metrics, the pipeline."""
import os

import numpy as np
import pytest
import torch as t

from oracle import lightgcn_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("m,n,k", [(1, 1, 1), (64, 64, 32), (100, 130, 70), (257, 33, 128), (33, 500, 5), (640, 64, 512)])
@pytest.mark.parametrize("trans_a,trans_b", [(False, True), (False, False), (True, False), (True, True)])
def test_gemm_bitwise_vs_fma_chain(m, n, k, trans_a, trans_b):
    from laplace_amd import ops
    g = t.Generator().manual_seed(m * 1000 + n + k)
    A = t.randn((k, m) if trans_a else (m, k), generator=g)
    B = t.randn((n, k) if trans_b else (k, n), generator=g)
    bias = t.randn(n, generator=g)
    got = ops.gemm(A.to(DEV), B.to(DEV), trans_a=trans_a, trans_b=trans_b, bias=bias.to(DEV), relu=True)
    want = R.gemm_fma(A, B, trans_a=trans_a, trans_b=trans_b, bias=bias, relu=True)
    assert t.equal(got.cpu(), want)
    # and it is a correct product: close to float64
    Ad = (A.T if trans_a else A).double()
    Bd = (B.T if trans_b else B).double()
    ref = t.relu(Ad @ Bd + bias.double())
    assert t.allclose(got.cpu().double(), ref, atol=1e-4, rtol=1e-5)


def test_gemm_accumulate_strides_and_asymmetry():
    """A = I against an asymmetric B catches a swapped C layout; accumulate and leading dimensions."""
    from laplace_amd import ops
    n = 96
    eye = t.eye(n)
    B = t.arange(n * n, dtype=t.float32).reshape(n, n) / 7.0  # B[i][j] != B[j][i]
    got = ops.gemm(eye.to(DEV), B.to(DEV), trans_b=False)
    assert t.equal(got.cpu(), B)
    g = t.Generator().manual_seed(0)
    big = t.randn(50, 200, generator=g).to(DEV)
    A = big[:, 8:8 + 72]
    W = t.randn(40, 72, generator=g).to(DEV)
    C = t.randn(50, 64, generator=g).to(DEV)
    Cv = C[:, :40]
    want = R.gemm_fma(A.cpu().contiguous(), W.cpu(), out=Cv.cpu().contiguous().clone(), accumulate=True)
    ops.gemm(A, W, out=Cv, accumulate=True)
    assert t.equal(Cv.cpu(), want)


def test_topk_matches_reference_golden(golden_dir):
    """ids and their order equal make_predictions_for_user of the reference (golden), for every user."""
    from laplace_amd.utils.metrics_lightgcn import topk_for_users, make_predictions_for_user, get_metrics_lightgcn
    g = t.load(os.path.join(golden_dir, "topk_metrics.pt"), weights_only=False)
    ue, ie, tr = g["users_emb"].to(DEV), g["items_emb"].to(DEV), g["train_edges"].to(DEV)
    U = ue.shape[0]
    users = t.arange(U, device=DEV)
    for k, per_user in g["preds"].items():
        top = topk_for_users(ue, ie, users, tr, k).cpu()
        for u in range(U):
            assert t.equal(top[u], per_user[u]), (k, u)
    excl = {u: v.to(DEV) for u, v in g["excl"].items()}
    one = make_predictions_for_user(ue, ie, 3, excl, 12)
    assert t.equal(one.cpu(), g["preds"][12][3])
    # recall / precision / ndcg of the reference's get_metrics_lightgcn
    from types import SimpleNamespace
    model = SimpleNamespace(users_emb=SimpleNamespace(weight=ue), items_emb=SimpleNamespace(weight=ie))
    for k, want in g["metrics"].items():
        got = get_metrics_lightgcn(model, g["eval_edges"].to(DEV), [tr], k)
        assert got == pytest.approx(want, abs=1e-6)


@pytest.mark.parametrize("n_items,k", [(1000, 12), (5000, 256), (100_000, 256), (70, 100), (3000, 1),
                                       # one-pass path (sample threshold + single collect pass): rows >= 32 768 items, also
                                       # rows that are not 16-byte aligned, k = 1 and the largest k
                                       (40_000, 1), (50_001, 12), (33_333, 256), (100_000, 1024)])
@pytest.mark.parametrize("D", [64, 128])
def test_topk_exact_vs_oracle(n_items, k, D):
    from laplace_amd import ops
    g = t.Generator().manual_seed(n_items + k)
    U, n_q = 300, 41
    ue, ie = t.randn(U, D, generator=g) * 0.1, t.randn(n_items, D, generator=g) * 0.1
    uid = t.randint(0, U, (n_q,), generator=g)
    excl = [t.randperm(n_items, generator=g)[: int(t.randint(0, min(n_items, 400), (1,), generator=g))] for _ in range(n_q)]
    rows = t.cat([t.full((len(e),), i) for i, e in enumerate(excl)]).long()
    ex = ops.coo_to_csr(rows.to(DEV), t.cat(excl).to(DEV), n_q, n_items, want_perm=False)
    ids, sc = ops.topk_excl(uid.to(DEV), ue.to(DEV), ie.to(DEV), k, ex, want_scores=True)
    scores = R.scores_fma(ue[uid], ie)
    want = R.topk_excl_exact(scores, excl, k)
    assert t.equal(ids.cpu(), want)
    valid = want >= 0
    assert t.equal(sc.cpu()[valid], scores.gather(1, want.clamp(min=0))[valid])  # bitwise scores


def test_topk_ties_resolved_by_item_id():
    """Duplicate item rows give exactly equal scores: ties at the cut must go to the smaller ids."""
    from laplace_amd import ops
    g = t.Generator().manual_seed(5)
    D, n_items, k = 32, 2000, 50
    base = t.randn(40, D, generator=g)
    ie = base[t.randint(0, 40, (n_items,), generator=g)]  # only 40 distinct rows -> massive ties
    ue = t.randn(7, D, generator=g)
    uid = t.arange(7)
    ids = ops.topk_excl(uid.to(DEV), ue.to(DEV), ie.to(DEV), k, None)
    want = R.topk_excl_exact(R.scores_fma(ue, ie), [t.empty(0, dtype=t.int64)] * 7, k)
    assert t.equal(ids.cpu(), want)
    zeros = ops.topk_excl(uid.to(DEV), t.zeros(7, D, device=DEV), ie.to(DEV), k, None)  # all scores equal (+-0)
    assert t.equal(zeros.cpu(), t.arange(k).repeat(7, 1))


@pytest.mark.parametrize("D", [32, 64, 128])  # 32: the generic fused kernel; 64 / 128: the LDS-DMA kernel and its overflow marks
def test_topk_large_rows_with_massive_ties_and_exclusions_fall_back_exactly(D):
    """Rows where the sampled threshold cannot work — thousands of equal scores at the cut, all scores equal,
    almost everything excluded — take the multi-pass path and still return the exact lists."""
    from laplace_amd import ops
    g = t.Generator().manual_seed(9)
    n_items, k = 50_000, 100
    base = t.randn(25, D, generator=g)
    ie = base[t.randint(0, 25, (n_items,), generator=g)]  # 25 distinct rows: ~2 000-way ties
    ue = t.randn(5, D, generator=g)
    uid = t.arange(5)
    none = [t.empty(0, dtype=t.int64)] * 5
    ids = ops.topk_excl(uid.to(DEV), ue.to(DEV), ie.to(DEV), k, None)
    assert t.equal(ids.cpu(), R.topk_excl_exact(R.scores_fma(ue, ie), none, k))
    zeros = ops.topk_excl(uid.to(DEV), t.zeros(5, D, device=DEV), ie.to(DEV), k, None)
    assert t.equal(zeros.cpu(), t.arange(k).repeat(5, 1))
    # 70 such queries: every score of every panel passes in every wavefront — far more than a staging region holds,
    # so the rows are marked overflowed from inside the fused kernel and recomputed exactly
    many = ops.topk_excl(t.arange(70, device=DEV), t.zeros(70, D, device=DEV), ie.to(DEV), k, None)
    assert t.equal(many.cpu(), t.arange(k).repeat(70, 1))
    # all but 60 items excluded for user 0, all but 3 000 for user 1: padded with -1 / exact
    ie2 = t.randn(n_items, D, generator=g)
    keep0, keep1 = t.randperm(n_items, generator=g)[:60], t.randperm(n_items, generator=g)[:3000]
    mask0, mask1 = t.ones(n_items, dtype=t.bool), t.ones(n_items, dtype=t.bool)
    mask0[keep0] = False; mask1[keep1] = False
    excl = [mask0.nonzero().view(-1), mask1.nonzero().view(-1)]
    rows = t.cat([t.full((len(e),), i) for i, e in enumerate(excl)]).long()
    ex = ops.coo_to_csr(rows.to(DEV), t.cat(excl).to(DEV), 2, n_items, want_perm=False)
    got = ops.topk_excl(uid[:2].to(DEV), ue.to(DEV), ie2.to(DEV), k, ex).cpu()
    want = R.topk_excl_exact(R.scores_fma(ue[:2], ie2), excl, k)
    assert t.equal(got, want) and int((got[0] >= 0).sum()) == 60


@pytest.mark.parametrize("D", [64, 128])
@pytest.mark.parametrize("n_q,k", [(1, 1), (70, 12), (300, 100), (2100, 12), (515, 256)])
def test_topk_bf16_prefilter_equals_the_f32_path(D, n_q, k, monkeypatch):
    """The bf16x3 prefilter (csrc/topk_prefilter.hpp) only decides who is a candidate; the answer comes from the exact
    fma chain.  Every item here has ~20 near copies whose scores differ by a few ulps — far inside the prefilter's error
    bound, so approximate scores cannot rank them — plus exact duplicates (ties resolved by id) and exclusions of the
    best items: ids and bitwise scores must equal the f32 fused kernel's and the oracle's."""
    from laplace_amd import ops
    g = t.Generator().manual_seed(100 * D + n_q + k)
    groups, copies = 2000, 20
    n_items = groups * copies
    base = t.randn(groups, D, generator=g) * 0.1
    ie = base.repeat(copies, 1)                                   # item i is a copy of base row i % groups
    wiggle = t.randint(-3, 4, (n_items, D), generator=g).float() * 2.0 ** -23
    wiggle[: 2 * groups] = 0.0                                    # the first two copies stay exact duplicates
    ie = ie * (1.0 + wiggle)
    U = max(n_q, 8)
    ue = t.randn(U, D, generator=g) * 0.1
    uid = t.randperm(U, generator=g)[:n_q]
    exact = R.scores_fma(ue[uid[:40]], ie)
    excl = []
    for q in range(n_q):   # exclude some of the best items of the first rows, random ones elsewhere
        if q < 40:
            top = exact[q].topk(30).indices
            excl.append(top[t.randperm(30, generator=g)[:11]])
        else:
            excl.append(t.randperm(n_items, generator=g)[: int(t.randint(0, 50, (1,), generator=g))])
    rows = t.cat([t.full((len(e),), i) for i, e in enumerate(excl)]).long()
    ex = ops.coo_to_csr(rows.to(DEV), t.cat(excl).to(DEV), n_q, n_items, want_perm=False)
    args = (uid.to(DEV), ue.to(DEV), ie.to(DEV), k, ex)
    monkeypatch.setenv("LAPLACE_TOPK_PREFILTER", "1")
    ids_b, sc_b = ops.topk_excl(*args, want_scores=True)
    monkeypatch.setenv("LAPLACE_TOPK_PREFILTER", "0")
    ids_f, sc_f = ops.topk_excl(*args, want_scores=True)
    assert t.equal(ids_b, ids_f)
    assert t.equal(sc_b, sc_f)
    if n_q > 600:   # the same call cut into chunks of 512 queries that alternate over two streams and two workspaces
        monkeypatch.setenv("LAPLACE_TOPK_PREFILTER", "1")
        monkeypatch.setattr(ops, "TOPK_WS_BYTES", 4 * n_items * 512)
        monkeypatch.setattr(ops, "TOPK_CHUNK_QUANTUM", 512)
        monkeypatch.setattr(ops, "TOPK_STREAMS", 2)
        ids_c, sc_c = ops.topk_excl(*args, want_scores=True)
        assert t.equal(ids_c, ids_b) and t.equal(sc_c, sc_b)
        monkeypatch.setattr(ops, "TOPK_STREAMS", 1)
        ids_d, _ = ops.topk_excl(*args, want_scores=True)
        assert t.equal(ids_d, ids_b)
    m = min(n_q, 40)
    want = R.topk_excl_exact(exact, excl[:m], k)
    assert t.equal(ids_b[:m].cpu(), want)
    valid = want >= 0
    assert t.equal(sc_b[:m].cpu()[valid], exact.gather(1, want.clamp(min=0))[valid])


@pytest.mark.parametrize("D", [64, 128])
@pytest.mark.parametrize("kind", ["gaussian", "wide_range", "cancelling", "constant_sign"])
def test_topk_prefilter_error_bound_holds(D, kind):
    """|bf16x3 score - exact fma chain| <= eps(u) = 2^-12 |u| max|i| — the bound the candidate selection rests on
    (csrc/topk_prefilter.hpp) — measured through the diagnostic entry on tables built to stress it: components over six
    decades, rows whose terms cancel to ~0 (all of the bound is rounding), rows of one sign (the largest sums).  The
    worst case observed must also stay well inside the bound (the derivation leaves a factor ~2.3)."""
    from laplace_amd import ops
    g = t.Generator().manual_seed(7 * D + len(kind))
    n_q, n_items = 200, 3000
    if kind == "gaussian":
        ue, ie = t.randn(n_q, D, generator=g), t.randn(n_items, D, generator=g)
    elif kind == "wide_range":
        ue = t.randn(n_q, D, generator=g) * 10.0 ** t.randint(-3, 4, (n_q, D), generator=g).float()
        ie = t.randn(n_items, D, generator=g) * 10.0 ** t.randint(-3, 4, (n_items, D), generator=g).float()
    elif kind == "cancelling":
        half = t.randn(n_q, D // 2, generator=g)
        ue = t.cat([half, half], 1)
        ih = t.randn(n_items, D // 2, generator=g)
        ie = t.cat([ih, -ih * (1.0 + t.randint(-2, 3, ih.shape, generator=g).float() * 2.0 ** -20)], 1)
    else:
        ue, ie = t.rand(n_q, D, generator=g) + 0.5, t.rand(n_items, D, generator=g) + 0.5
    approx, eps = ops.topk_prefilter_scores(None, ue.to(DEV), ie.to(DEV))
    exact = R.scores_fma(ue, ie)
    err = (approx.cpu().double() - exact.double()).abs()
    bound = eps.cpu().double()[:, None]
    want_eps = 2.0 ** -12 * ue.double().norm(dim=1) * ie.double().norm(dim=1).max()
    assert t.all(eps.cpu().double() >= want_eps) and t.all(eps.cpu().double() <= 1.03 * want_eps)
    assert t.all(err <= bound)
    assert float((err / bound).max()) < 0.6


def test_pipeline_end_to_end_small():
    from laplace_amd import synthetic as S
    from laplace_amd.config import LightGCNConfig
    from laplace_amd.run_pipeline_lightgcn import train
    spec = S.SyntheticSpec(600, 400, 12000, seed=2, deg_min=5)
    ei = S.generate(spec)
    homog = ei.clone()
    homog[1] += spec.num_users  # to_homogeneous() numbering, as the reference's loader receives it
    cfg = LightGCNConfig(epochs=60, k=12, hidden_layer_size=32, learning_rate=5e-3, save_model=False, batch_size=512,
                         num_iterations=3, eval_every=30, lr_decay_every=20, Lambda=1e-6, show_graph=False,
                         num_recommendations=50)
    for compat in ("reference", "bipartite"):
        stats = train(cfg, edge_index=homog, num_users=spec.num_users, num_articles=spec.num_items, compat=compat,
                      verbose=False)
        assert np.isfinite([stats.loss, stats.recall_val, stats.recall_test, stats.precision_val, stats.precision_test]).all()
        assert stats.loss < -0.70  # BPR-as-written goes down from -log(2)
        assert 0.0 <= stats.recall_test <= 1.0 and 0.0 <= stats.precision_test <= 1.0


def test_gemm_split_k_weight_gradient_shape():
    """dW = dY^T X with a 128 x 84 output over K = 30K rows: K is cut into slices (the output grid alone
    is 4 workgroups).  Result = slice chains added in a fixed association: bitwise reproducible and bitwise
    equal to the oracle evaluated the same way.  K = 30 001 also exercises the odd-K tail of the fast path."""
    from laplace_amd import ops
    g = t.Generator().manual_seed(0)
    K, M, N = 30_001, 128, 84
    dY, X = t.randn(K, M, generator=g), t.randn(K, N, generator=g)
    got = ops.gemm(dY.to(DEV), X.to(DEV), trans_a=True, trans_b=False)
    again = ops.gemm(dY.to(DEV), X.to(DEV), trans_a=True, trans_b=False)
    assert t.equal(got, again)
    blocks = ((M + 63) // 64) * ((N + 63) // 64)
    s = min(-(-512 // blocks), K // 128)  # outputs of <= 8 tiles: one 128-wide panel of K per slice at least
    kps = -(-(-(-K // s)) // 32) * 32
    # slabs = one fma chain per K slice; the reduce adds every 8th slab in order per lane group, then the 8
    # group sums in group order (csrc/gemm.hip:gemm_splitk_reduce_kernel)
    slabs = [R.gemm_fma(dY[k0:k0 + kps], X[k0:k0 + kps], trans_a=True, trans_b=False) for k0 in range(0, K, kps)]
    groups = []
    for grp in range(8):
        acc = t.zeros(M, N)
        for z in range(grp, len(slabs), 8):
            acc = acc + slabs[z]
        groups.append(acc)
    want = groups[0]
    for q in range(1, 8):
        want = want + groups[q]
    assert t.equal(got.cpu(), want)
    assert t.allclose(got.cpu().double(), dY.double().T @ X.double(), atol=2e-3, rtol=1e-5)


def test_topk_against_the_torch_product_the_reference_calls():
    """The reference scores with torch on the CPU, `user_emb[user] @ item_emb.T` then torch.topk
    (utils/metrics_lightgcn.py:137-139); the device scores with a k-ascending fp32 fma chain.  Where exactness
    against THAT product holds and where it cannot:

      (a) embeddings on a dyadic grid (every product and partial sum exact in fp32): the two products are the same
          bits in any summation order, so ids must agree exactly wherever scores are distinct, and the score lists
          exactly; items tied in score come out id-ascending here, in torch's unspecified order there;
      (b) generic floats with planted NEAR ties (pairs of item rows one ulp apart in one coordinate): the CPU product's
          blocked summation and the fma chain may order such a pair differently.  Every disagreement must be such a
          pair — the two orders differ only between scores that are within rounding distance of each other, and both
          are permutations of the float64 ranking's top-k up to that distance."""
    from laplace_amd import ops
    g = t.Generator().manual_seed(31)
    # ---- (a) exact arithmetic: integers / 64 in [-1, 1], D = 64  =>  |score| <= 64, products multiples of 2^-12
    U, I, D, k = 48, 6000, 64, 12
    ue = t.randint(-64, 65, (U, D), generator=g).float() / 64.0
    ie = t.randint(-64, 65, (I, D), generator=g).float() / 64.0
    ie[1000:1100] = ie[:100]                                   # exact ties between distinct ids
    uid = t.arange(U)
    ids, sc = ops.topk_excl(uid.to(DEV), ue.to(DEV), ie.to(DEV), k, None, want_scores=True)
    ids, sc = ids.cpu(), sc.cpu()
    prod = ue @ ie.T                                           # the reference's call
    assert t.equal(prod, R.scores_fma(ue, ie))                 # order-independent on the grid
    tv, ti = t.topk(prod, k, dim=1)
    assert t.equal(sc, tv)                                     # the same k scores, in order, bit for bit
    for u in range(U):
        assert t.equal(prod[u, ids[u]], tv[u])
        distinct = t.ones(k, dtype=t.bool)
        distinct[1:] &= tv[u, 1:] != tv[u, :-1]
        distinct[:-1] &= tv[u, :-1] != tv[u, 1:]
        kth_unique = bool((prod[u] == tv[u, -1]).sum() == 1)
        sel = distinct.clone()
        sel[-1] &= kth_unique
        assert t.equal(ids[u][sel], ti[u][sel])                # exact ids wherever the score is not shared
        for a, b in zip(ids[u][:-1].tolist(), ids[u][1:].tolist()):  # ties: smaller id first (the documented rule)
            if prod[u, a] == prod[u, b]:
                assert a < b
    # ---- (b) near ties: item 2j+1 = item 2j with one coordinate nudged by one ulp
    I2 = 4000
    ue2 = t.randn(U, D, generator=g) * 0.1
    ie2 = t.randn(I2, D, generator=g) * 0.1
    ie2[1::2] = ie2[0::2]
    col = t.randint(0, D, (I2 // 2,), generator=g)
    ar = t.arange(I2 // 2)
    ie2[1::2][ar, col] = t.nextafter(ie2[0::2][ar, col], t.full((I2 // 2,), 1.0))
    ids2 = ops.topk_excl(uid.to(DEV), ue2.to(DEV), ie2.to(DEV), k, None).cpu()
    prod2 = ue2 @ ie2.T
    ti2 = t.topk(prod2, k, dim=1).indices
    exact = (ue2.double() @ ie2.double().T)
    eps = 64 * 2.0 ** -24 * float(prod2.abs().max())          # rounding distance of a 64-term fp32 dot at this scale
    differing = 0
    for u in range(U):
        if not t.equal(ids2[u], ti2[u]):
            differing += 1
        for lst in (ids2[u], ti2[u]):                          # both lists are the float64 top-k up to rounding distance
            kth = float(t.topk(exact[u], k).values[-1])
            assert float(exact[u, lst].min()) >= kth - eps
            assert bool((exact[u, lst][:-1] >= exact[u, lst][1:] - eps).all())
        for a, b in zip(ids2[u].tolist(), ti2[u].tolist()):    # position by position: same item, or a near-tied stand-in
            assert a == b or abs(float(exact[u, a] - exact[u, b])) <= eps
    # the device list is exact against its own arithmetic in every case
    assert t.equal(ids2, R.topk_excl_exact(R.scores_fma(ue2, ie2), [t.empty(0, dtype=t.int64)] * U, k))
    print(f"near-tie rows ordered differently by torch's CPU product and the fma chain: {differing}/{U}")


def test_gemm_group_pairs_mask_split_and_layouts():
    """mi_gemm_group_f32: several problems in one launch — two-pair products (lin_l + lin_r as one fma chain), a masked
    A (relu backward), every operand layout, split-K weight-gradient shapes, an empty problem — each against the
    oracle's fma chain evaluated the same way (bitwise where no split is involved) and against float64."""
    from laplace_amd import ops
    g = t.Generator().manual_seed(12)
    n_d, cs, cd, co = 3001, 76, 84, 128
    agg, xd = t.randn(n_d, cs, generator=g), t.randn(n_d, cd, generator=g)
    wl, wr, bl = t.randn(co, cs, generator=g), t.randn(co, cd, generator=g), t.randn(co, generator=g)
    dy, outm = t.randn(n_d, co, generator=g), t.randn(n_d, co, generator=g)
    small_a, small_b = t.randn(40, 64, generator=g), t.randn(24, 64, generator=g)
    dev = lambda x: x.to(DEV)
    A, X, WL, WR, BL, DY, OM, SA, SB = map(dev, (agg, xd, wl, wr, bl, dy, outm, small_a, small_b))
    out_fwd = t.empty(n_d, co, device=DEV)
    out_dagg, out_dx = t.empty(n_d, cs, device=DEV), t.empty(n_d, cd, device=DEV)
    out_dwl, out_dwr = t.empty(co, cs, device=DEV), t.empty(co, cd, device=DEV)
    out_small = t.randn(40, 24, generator=g).to(DEV)
    small_before = out_small.cpu().clone()
    empty_out = t.empty(0, co, device=DEV)
    probs = [
        ops.gemm_problem(A, WL, out_fwd, A2=X, B2=WR, bias=BL, relu=True),                 # forward, two pairs
        ops.gemm_problem(DY, WL, out_dagg, trans_b=False, a_mask=OM),                      # dAgg = (dY * (out>0)) @ W_l
        ops.gemm_problem(DY, WR, out_dx, trans_b=False, a_mask=OM),
        ops.gemm_problem(DY, A, out_dwl, trans_a=True, trans_b=False, a_mask=OM),          # dW_l: split-K, masked
        ops.gemm_problem(DY, X, out_dwr, trans_a=True, trans_b=False, a_mask=OM),
        ops.gemm_problem(SA, SB, out_small, accumulate=True),                              # += on a small output
        ops.gemm_problem(t.empty(0, cs, device=DEV), WL, empty_out),                       # nothing to do
        ops.gemm_problem(A, WL, t.empty(n_d, co, device=DEV)),                             # a ninth... (eighth) problem
        ops.gemm_problem(X, WR, out_dx.new_empty(n_d, co)),                                # spills into a second launch
    ]
    assert ops.gemm_group(probs)
    dym = dy * (outm > 0)
    # two pairs = one chain: pair 0's k ascending, then pair 1's — the oracle's chain over the concatenated operands
    want_fwd = R.gemm_fma(t.cat([agg, xd], 1), t.cat([wl, wr], 1), bias=bl, relu=True)
    assert t.equal(out_fwd.cpu(), want_fwd)
    assert t.equal(out_dagg.cpu(), R.gemm_fma(dym, wl, trans_b=False))
    assert t.equal(out_dx.cpu(), R.gemm_fma(dym, wr, trans_b=False))
    assert t.equal(out_small.cpu(), R.gemm_fma(small_a, small_b, out=small_before.clone(), accumulate=True))
    for got, xx in ((out_dwl, agg), (out_dwr, xd)):
        want = dym.double().T @ xx.double()
        assert (got.cpu().double() - want).abs().max() <= 1e-5 * float(want.abs().max()) + 1e-4
    again = t.empty_like(out_dwl)
    assert ops.gemm_group([ops.gemm_problem(DY, A, again, trans_a=True, trans_b=False, a_mask=OM)])
    assert t.equal(again, out_dwl)                                                        # split-K reduce: fixed order
    # unaligned operands are refused as a whole (the caller falls back to gemm())
    odd = t.randn(50, 7, device=DEV)
    assert not ops.gemm_group([ops.gemm_problem(odd, t.randn(9, 7, device=DEV), t.empty(50, 9, device=DEV))])


@pytest.mark.parametrize("k,m,n1,n2,bias,mask", [(1, 32, 4, 4, True, False), (127, 64, 84, 76, True, True), (128, 128, 128, 128, True, True),
                                                  (129, 128, 84, 0, False, True), (30011, 128, 84, 84, True, True),
                                                  (2049, 64, 128, 128, True, False), (5000, 96, 37, 200, True, True)])
def test_sage_wgrad_matches_the_three_products(k, m, n1, n2, bias, mask):
    """mi_sage_wgrad_f32: gw1 = (dy * relu')^T b1, gb = (dy * relu')^T 1, gw2 = (dy * relu')^T b2 against float64 products;
    deterministic; several problems per call."""
    import torch as t
    from laplace_amd import ops
    g = t.Generator(device="cuda").manual_seed(k + m)
    dy = t.randn(k, m, device="cuda", generator=g)
    mk = t.randn(k, m, device="cuda", generator=g) if mask else None
    b1 = t.randn(k, n1, device="cuda", generator=g)
    b2 = t.randn(k, n2, device="cuda", generator=g) if n2 else None
    def problem():
        return dict(dy=dy, mask=mk, b1=b1, b2=b2, gw1=t.full((m, n1), float("nan"), device="cuda"),
                    gb=t.full((m,), float("nan"), device="cuda") if bias else None,
                    gw2=t.full((m, n2), float("nan"), device="cuda") if n2 else None)
    p, p2 = problem(), problem()
    small = dict(dy=dy[:7].contiguous(), mask=None, b1=b1[:7].contiguous(), b2=None, gw1=t.empty(m, n1, device="cuda"), gb=None, gw2=None)
    assert ops.sage_wgrad([p, small])
    assert ops.sage_wgrad([p2])
    a = (dy * (mk > 0) if mask else dy).double()
    for got, want in ((p["gw1"], a.t() @ b1.double()), (p["gb"], a.sum(0) if bias else None), (p["gw2"], a.t() @ b2.double() if n2 else None)):
        if want is None:
            continue
        scale = float(want.abs().max()) + 1e-12
        assert float((got.double() - want).abs().max()) <= 2e-6 * scale * max(1.0, (k / 1000) ** 0.5), (got.shape,)
    for key in ("gw1", "gb", "gw2"):
        if p[key] is not None:
            assert t.equal(p[key], p2[key])
    assert float((small["gw1"].double() - dy[:7].double().t() @ b1[:7].double()).abs().max()) <= 1e-5
    # shapes outside the kernel's: declined, nothing written
    bad = dict(dy=t.randn(10, 48, device="cuda"), mask=None, b1=t.randn(10, 8, device="cuda"), b2=None,
               gw1=t.zeros(48, 8, device="cuda"), gb=None, gw2=None)
    assert not ops.sage_wgrad([bad]) and float(bad["gw1"].abs().sum()) == 0.0

"""Drives the host side of the two native executors — the COUNT pass (mi_*_step_workspace_bytes) and the CHECK pass
(mi_*_step_check): descriptor walks that enqueue nothing and read no device memory — over valid, truncated and misaligned
descriptors.  Run by tests/test_host_sanitizers.py as a child process against the host-ASan/UBSan build of the library
(laplace_amd.build.build_asan, LD_PRELOAD = the ASan runtime); also runs against the product library.  Device pointers are
made-up addresses: neither pass may dereference them.  Prints one line per case and 'ASAN_DRIVER_OK' at the end."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from laplace_amd import _lib  # noqa: E402

L = _lib.lib()
OK, BAD_ARG, UNSUPPORTED = 0, -1, _lib.MI_ERR_UNSUPPORTED


class Fake:
    """Bump allocator of made-up, 256-byte aligned device addresses."""

    def __init__(self):
        self.at = 0x7F00_0000_0000

    def __call__(self, nbytes):
        p = self.at
        self.at += (int(nbytes) + 255) // 256 * 256 + 256
        return p


def ranker_case(n_c=3000, n_a=900, nnz=5000, n_label=700, mean=False, p_drop=0.3):
    f = Fake()
    d, b = _lib.RankerModel(), _lib.RankerBatch()
    d.n_enc_layers, d.n_dec_layers, d.aggr, d.batch_normalize = 2, 2, int(mean), 1
    d.p_dropout, d.max_norm = p_drop, 1.0
    dims = ([40, 2, 12, 4, 4, 2], [20, 12, 12, 12])
    rows = ([352_899, 2, 84, 4, 5, 2], [47_224, 132, 30, 50])
    for ti in range(2):
        d.n_cols[ti] = len(dims[ti])
        for c, (w, r) in enumerate(zip(dims[ti], rows[ti])):
            d.tables[ti][c], d.table_rows[ti][c], d.dims[ti][c] = f(4 * w * r), r, w
    width = [sum(dims[0]), sum(dims[1])]
    params = []

    def par(n):
        p = (f(4 * n), f(4 * n), f(4 * n), f(4 * n), n)
        params.append(p)
        return p
    for l, c_out in enumerate((128, 64)):
        for r in range(2):                       # r = 0: customer -> article, r = 1: article -> customer
            cv = d.conv[l][r]
            cv.c_src, cv.c_dst, cv.c_out = width[r], width[1 - r], c_out
            wl, bl, wr = par(c_out * cv.c_src), par(c_out), par(c_out * cv.c_dst)
            cv.w_l, cv.gw_l, cv.b_l, cv.gb_l, cv.w_r, cv.gw_r = wl[0], wl[1], bl[0], bl[1], wr[0], wr[1]
        width = [c_out, c_out]
    for ti in range(2):
        nm = d.norm[ti]
        g, be = par(64), par(64)
        nm.gamma, nm.g_gamma, nm.beta, nm.g_beta = g[0], g[1], be[0], be[1]
        nm.running_mean, nm.running_var, nm.num_batches_tracked = f(256), f(256), f(8)
        nm.momentum, nm.eps = 0.1, 1e-5
    for j, (i_, o_) in enumerate(((128, 128), (128, 1))):
        ln = d.dec[j]
        w, bb = par(i_ * o_), par(o_)
        ln.w, ln.gw, ln.b, ln.gb = w[0], w[1], bb[0], bb[1]
        setattr(ln, "in_", i_)
        ln.out = o_
    d.n_params, d.apply_adam = len(params), 1
    for i, (p, g, m, v, n) in enumerate(params):
        q = d.params[i]
        q.p, q.g, q.m, q.v, q.n = p, g, m, v, n
    d.lr, d.beta1, d.beta2, d.eps, d.step = 0.01, 0.9, 0.999, 1e-8, 1
    n_ones = max(n_c, n_a, n_label)
    d.ones4, d.n_ones = f(16 * n_ones), n_ones
    b.n_nodes[0], b.n_nodes[1] = n_c, n_a
    b.x[0], b.x[1] = f(8 * n_c * 6), f(8 * n_a * 4)
    b.by_customer_ptr, b.by_customer_col = f(4 * (n_c + 1)), f(4 * max(nnz, 1))
    b.by_article_ptr, b.by_article_col = f(4 * (n_a + 1)), f(4 * max(nnz, 1))
    b.nnz, b.n_label = nnz, n_label
    b.label_row, b.label_col, b.label = f(8 * n_label), f(8 * n_label), f(8 * n_label)
    b.seed, b.step, b.loss = 12345, 7, f(4)
    return d, b, f


def pinsage_case(hidden=16, layers=2, n_items=105_542, sizes=((700, 280, 830), (280, 93, 279)), pairs=31):
    f = Fake()
    d, b = _lib.PinsageModel(), _lib.PinsageStepBatch()
    d.n_layers, d.hidden, d.n_items = layers, hidden, n_items
    tab = 4 * (n_items + 1) * hidden
    d.proj, d.g_proj, d.m_proj, d.v_proj = f(tab), f(tab), f(tab), f(tab)
    d.bias, d.g_bias = f(4 * n_items), f(4 * n_items)
    i = 0

    def par(n):
        nonlocal i
        q = d.params[i]
        q.p, q.g, q.m, q.v, q.n = f(4 * n), f(4 * n), f(4 * n), f(4 * n), n
        i += 1
        return q
    bq = par(n_items)                              # the scorer bias is an ordinary entry of the optimizer's list
    bq.p, bq.g = d.bias, d.g_bias
    for l in range(layers):
        c = d.conv[l]
        qw, qb, ww, wb = par(hidden * hidden), par(hidden), par(2 * hidden * hidden), par(hidden)
        c.q_w, c.g_q_w, c.q_b, c.g_q_b = qw.p, qw.g, qb.p, qb.g
        c.w_w, c.g_w_w, c.w_b, c.g_w_b = ww.p, ww.g, wb.p, wb.g
    d.n_params, d.apply_adam, d.p_dropout = i, 1, 0.5
    d.lr, d.beta1, d.beta2, d.eps, d.step = 3e-5, 0.9, 0.999, 1e-8, 1
    n_ones = max(s[0] for s in sizes)
    d.ones4, d.n_ones = f(16 * n_ones), n_ones
    b.n_blocks = layers
    for l, (ns, nd, nnz) in enumerate(sizes[:layers]):
        sb = b.blocks[l]
        sb.n_src, sb.n_dst, sb.nnz, sb.src_ids = ns, nd, nnz, f(8 * ns)
        sb.dst_rowptr, sb.dst_col, sb.dst_val = f(4 * (nd + 1)), f(4 * nnz), f(4 * nnz)
        sb.src_rowptr, sb.src_col, sb.src_val = f(4 * (ns + 1)), f(4 * nnz), f(4 * nnz)
    n_seeds = sizes[layers - 1][1]
    b.n_seeds, b.n_pairs = n_seeds, pairs
    b.seeds, b.pos_u, b.pos_v, b.neg_v = f(8 * n_seeds), f(8 * pairs), f(8 * pairs), f(8 * pairs)
    b.seed, b.step, b.loss = 99, 3, f(4)
    return d, b, f


def run(name, count_fn, check_fn, d, b, f, want, ws_shift=0, ws_cut=None):
    need = int(count_fn(ctypes.byref(d), ctypes.byref(b)))
    ws = f(max(need, 256)) + ws_shift
    cap = need if ws_cut is None else ws_cut
    rc = int(check_fn(ctypes.byref(d), ctypes.byref(b), ws, cap))
    verdict = "ok" if (rc == 0) == (want == 0) else "UNEXPECTED"
    print(f"{name:58s} workspace {need:>12d} B  check rc {rc:3d}  ({'accepted' if rc == 0 else 'declined'})  {verdict}", flush=True)
    if verdict != "ok":
        raise SystemExit(f"case {name}: rc {rc}, expected {'0' if want == 0 else 'non-zero'}")


def main():
    rk = (L.mi_ranker_step_workspace_bytes, L.mi_ranker_step_check)
    pn = (L.mi_pinsage_step_workspace_bytes, L.mi_pinsage_step_check)
    # ---- valid
    for kw in (dict(), dict(mean=True, p_drop=0.0), dict(n_c=32_000, n_a=9_000, nnz=46_000, n_label=3_000), dict(nnz=0, n_label=4)):
        run(f"ranker valid {kw}", *rk, *ranker_case(**kw), want=0)
    for kw in (dict(), dict(hidden=64), dict(hidden=128, layers=1, sizes=((93, 93, 0),))):
        run(f"pinsage valid {kw}", *pn, *pinsage_case(**kw), want=0)
    # ---- inference descriptors (round 4: mi_ranker_batch.logits): no gradient pointers, no parameter list, no labels / loss
    def inference_case(**kw):
        d, b, f = ranker_case(**kw)
        b.logits, b.loss, b.label, b.label_f32 = f(4 * int(b.n_label)), None, None, None
        for l in range(int(d.n_enc_layers)):
            for r in range(2):
                d.conv[l][r].gw_l = d.conv[l][r].gw_r = d.conv[l][r].gb_l = None
        for t_ in range(2):
            d.norm[t_].g_gamma = d.norm[t_].g_beta = None
        for j in range(int(d.n_dec_layers)):
            d.dec[j].gw = d.dec[j].gb = None
        d.n_params = 0
        return d, b, f
    for kw in (dict(), dict(mean=True), dict(n_c=32_000, n_a=9_000, nnz=46_000, n_label=3_000)):
        run(f"ranker inference valid {kw}", *rk, *inference_case(**kw), want=0)
    d, b, f = inference_case()
    d.norm[1].running_mean = None
    run("ranker inference without running statistics", *rk, d, b, f, want=UNSUPPORTED)
    d, b, f = inference_case()
    d.dec[0].w = None
    run("ranker inference without a decoder weight", *rk, d, b, f, want=UNSUPPORTED)
    # ---- truncated / out-of-range descriptors: counts beyond the fixed arrays, negative sizes, missing pointers
    def mut(case, **fields):
        d, b, f = case
        for k, v in fields.items():
            obj, name = (d, k[2:]) if k.startswith("d_") else (b, k[2:])
            setattr(obj, name, v)
        return d, b, f
    for fields in (dict(d_n_enc_layers=0), dict(d_n_enc_layers=_lib.MI_RANKER_MAX_LAYERS + 5), dict(d_n_dec_layers=99),
                   dict(d_n_params=_lib.MI_RANKER_MAX_PARAMS + 1), dict(d_n_params=-3), dict(d_aggr=7), dict(d_p_dropout=1.5),
                   dict(b_nnz=1 << 21), dict(b_nnz=-1), dict(b_n_label=-5), dict(b_n_label=1 << 40), dict(b_loss=None),
                   dict(b_label_row=None), dict(b_by_article_ptr=None), dict(d_ones4=None), dict(d_n_ones=3)):
        run(f"ranker truncated {fields}", *rk, *mut(ranker_case(), **fields), want=UNSUPPORTED)
    d, b, f = ranker_case()
    d.n_cols[0] = _lib.MI_RANKER_MAX_COLS + 4
    run("ranker truncated n_cols[0] beyond MAX_COLS", *rk, d, b, f, want=UNSUPPORTED)
    d, b, f = ranker_case()
    b.n_nodes[1] = -7
    run("ranker truncated negative node count", *rk, d, b, f, want=UNSUPPORTED)
    d, b, f = ranker_case()
    d.conv[1][0].c_out = 0
    run("ranker truncated zero-width conv", *rk, d, b, f, want=UNSUPPORTED)
    d, b, f = ranker_case()
    d.params[3].p = None
    run("ranker truncated parameter without storage", *rk, d, b, f, want=UNSUPPORTED)
    for fields in (dict(d_n_layers=0), dict(d_n_layers=_lib.MI_PINSAGE_MAX_LAYERS + 3), dict(d_hidden=18), dict(d_hidden=0),
                   dict(d_hidden=4096), dict(d_n_params=_lib.MI_PINSAGE_MAX_PARAMS + 2), dict(d_n_items=-1), dict(b_n_blocks=1),
                   dict(b_n_blocks=9), dict(b_n_pairs=0), dict(b_n_seeds=-4), dict(b_seeds=None), dict(d_proj=None), dict(b_loss=None)):
        run(f"pinsage truncated {fields}", *pn, *mut(pinsage_case(), **fields), want=UNSUPPORTED)
    d, b, f = pinsage_case()
    b.blocks[0].n_dst = b.blocks[0].n_src + 5
    run("pinsage truncated block with more destinations than sources", *pn, d, b, f, want=UNSUPPORTED)
    d, b, f = pinsage_case()
    b.blocks[1].nnz = -2
    run("pinsage truncated negative edge count", *pn, d, b, f, want=UNSUPPORTED)
    # ---- misaligned: operands off their 16-byte grid, workspace misaligned or short
    d, b, f = ranker_case()
    d.conv[0][0].w_l += 4
    run("ranker misaligned weight (+4 B)", *rk, d, b, f, want=UNSUPPORTED)
    d, b, f = ranker_case()
    d.params[0].g += 8
    run("ranker misaligned gradient (+8 B)", *rk, d, b, f, want=UNSUPPORTED)
    run("ranker misaligned workspace (+4 B)", *rk, *ranker_case(), want=BAD_ARG, ws_shift=4)
    run("ranker short workspace", *rk, *ranker_case(), want=UNSUPPORTED, ws_cut=4096)
    d, b, f = pinsage_case()
    d.conv[0].q_w += 4
    run("pinsage misaligned weight (+4 B)", *pn, d, b, f, want=UNSUPPORTED)
    run("pinsage misaligned workspace (+8 B)", *pn, *pinsage_case(), want=BAD_ARG, ws_shift=8)
    run("pinsage short workspace", *pn, *pinsage_case(), want=UNSUPPORTED, ws_cut=1024)
    # ---- the library's other host-only entry points (size queries, support predicates), edge values included
    for n_rows, nnz in ((0, 0), (1, 0), (10, 1000), (8_100_000, 200_000_000), (-1, 5), (5, -1)):
        L.mi_coo_to_csr_workspace_bytes(n_rows, nnz); L.mi_csr_transpose_workspace_bytes(n_rows, nnz)
        L.mi_spmm_plan_workspace_bytes(n_rows, nnz)
    st = _lib.SpmmPlanStruct()
    for n_items, dd in ((0, 128), (7, 128), (-3, 64), (1 << 30, 512)):
        st.n_items = n_items
        L.mi_spmm_workspace_bytes(ctypes.byref(st), dd)
    assert L.mi_spmm_workspace_bytes(None, 128) == 0
    # round 4: the packed-entries and live-bits entry points reject bad descriptors before they enqueue or read anything
    for n in (0, 1, 7_726_979, -4):
        L.mi_spmm_plan_pack_workspace_bytes(n)
    f0 = Fake()
    st = _lib.SpmmPlanStruct()
    assert L.mi_spmm_plan_pack_entries(None, 10, f0(64), f0(64), 0, f0(1 << 20), 1 << 20, None) == BAD_ARG
    assert L.mi_spmm_plan_pack_entries(ctypes.byref(st), 10, f0(64), f0(64), 0, f0(1 << 20), 1 << 20, None) == BAD_ARG   # no items / epos
    st.items, st.epos, st.ecol, st.eval, st.n_launch = f0(4096), f0(1024), f0(1024), f0(1024), 0
    assert L.mi_spmm_plan_pack_entries(ctypes.byref(st), 10, f0(64), f0(64), 0, f0(1 << 20), 1 << 20, None) == BAD_ARG   # n_launch = 0
    st.n_launch = 256
    assert L.mi_spmm_plan_pack_entries(ctypes.byref(st), -1, f0(64), f0(64), 0, f0(1 << 20), 1 << 20, None) == BAD_ARG
    assert L.mi_spmm_plan_pack_entries(ctypes.byref(st), 10, None, f0(64), 0, f0(1 << 20), 1 << 20, None) == BAD_ARG    # col needed unless values_only
    assert L.mi_spmm_plan_pack_entries(ctypes.byref(st), 10, f0(64), f0(64), 0, f0(256), 256, None) == -3                # MI_ERR_WORKSPACE
    assert L.mi_map_live_bits_i32(-1, f0(64), f0(64), None) == BAD_ARG and L.mi_map_live_bits_i32(5, None, f0(64), None) == BAD_ARG
    assert L.mi_map_live_bits_i32(0, None, None, None) == 0
    for v in (0, 1, 128, 131072, -5):
        L.mi_bpr_workspace_bytes(v); L.mi_batch_nodes_workspace_bytes(v); L.mi_batchnorm_workspace_bytes(v)
    for m, n, k in ((1, 1, 1), (64, 64, 8192), (30000, 128, 84), (0, 5, 5), (-1, 2, 3), (128, 170, 1 << 20)):
        L.mi_gemm_workspace_bytes(m, n, k); L.mi_linear1_bwd_workspace_bytes(m, n)
    f = Fake()
    probs = (_lib.GemmProblem * 8)()
    for i in range(8):
        q = probs[i]
        q.trans_a, q.trans_b, q.m, q.n, q.k = i & 1, (i >> 1) & 1, 3000 + i, 64 + 4 * i, 84 + 4 * i
        q.A, q.lda, q.B, q.ldb, q.C, q.ldc = f(1 << 22), 4096, f(1 << 22), 4096, f(1 << 22), 4096
    for n in (0, 1, 8):
        L.mi_gemm_group_workspace_bytes(probs, n); L.mi_gemm_group_supported(probs, n)
    probs[3].lda = 4097          # not float4-addressable: the predicate must say so, not read anything
    assert L.mi_gemm_group_supported(probs, 8) == 0
    assert L.mi_gemm_group_supported(None, 1) == 0 and L.mi_gemm_group_supported(probs, 9) == 0
    for nq, ni, k in ((1, 100, 12), (4096, 100_000, 256), (0, 0, 0), (16384, 105_542, 1024), (-1, 10, 3)):
        L.mi_topk_workspace_bytes(nq, ni, k); L.mi_topk_prefilter_scores_workspace_bytes(nq, ni)
    for args in ((96, 2, 10), (384, 3, 10), (0, 1, 1), (-1, 2, 3)):
        L.mi_pinsage_neighbors_workspace_bytes(*args)
    assert L.mi_gather_cat_bwd_max_edges() > 0 and L.mi_error_string(-4) and L.mi_error_string(12345)
    print("host-only size queries and predicates: ok", flush=True)
    # ---- null descriptors
    assert L.mi_ranker_step_workspace_bytes(None, None) == 0 and L.mi_pinsage_step_workspace_bytes(None, None) == 0
    assert L.mi_ranker_step_check(None, None, None, 0) == BAD_ARG and L.mi_pinsage_step_check(None, None, None, 0) == BAD_ARG
    print("ASAN_DRIVER_OK", flush=True)


if __name__ == "__main__":
    main()

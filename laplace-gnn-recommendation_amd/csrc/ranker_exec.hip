// b9: native executor of one ranker training iteration (include/laplace_hip.h, mi_ranker_step_f32).
//
// The reference's loop body (training.py:19-34) on the default model (model/encoder_decoder.py:75-150) is ~75 kernel
// launches of 5-40 us each at the reference's batch size (24 users, ~3*10^4 nodes): issued op by op from Python the
// iteration is bound by the host (0.87 ms of launches against 0.77 ms of kernels).  This file
// issues the SAME launches — the same extern "C" entry points with the same operands, in the order
// laplace_amd/ranker_step.py::FusedRankerStep issues them — from one C call.  What is new here is only glue:
//   * the workspace layout (every activation, saved tensor and gradient temporary of the iteration),
//   * Philox feature dropout (forward and, regenerated from the same counters, backward),
//   * the per-relation aggregation weights (ones / 1 / in-degree) and the int64 -> f32 label cast in one launch,
//   * a multi-tensor Adam over the optimizer's parameter list (mi_adam_update4: mi_adam_dense_f32's arithmetic).
// The body runs three times over the same code: COUNT (workspace size), CHECK (every operand of every product is
// validated against the kernels' requirements — nothing has been enqueued if it returns MI_ERR_UNSUPPORTED), LAUNCH.
#include "common.hpp"
#include "pairs.hpp"
#include <algorithm>

extern "C" int mi_gemm_group_supported(const mi_gemm_problem* problems, int32_t n);

namespace {

constexpr int kBlock = 256;
constexpr int kTypes = 2;  // 0 = customer, 1 = article

// ---- small kernels ------------------------------------------------------------------------------------------------------

// Aggregation weights of both relations and the label cast, one launch.
//   v_art[p]  (by_article CSR, relation 0 forward):   mean ? 1 / deg(article row)          : 1
//   v_cus[p]  (by_customer CSR, relation 1 forward):  mean ? 1 / deg(customer row)         : 1
//   vt_cus[p] (by_customer CSR, relation 0 backward): mean ? 1 / deg(article = column)     : 1
//   vt_art[p] (by_article CSR, relation 1 backward):  mean ? 1 / deg(customer = column)    : 1
// (BipartiteGraph.weights of model/layers.py: scale_csr(by_dst, inv) and scale_csr(by_src, col_scale = inv).)
__device__ __forceinline__ void ranker_prep_body(const int64_t block_id, const int64_t n_blocks, int64_t nnz, int64_t n_c, int64_t n_a, int mean,
                                                             const int32_t* __restrict__ cptr, const int32_t* __restrict__ ccol,
                                                             const int32_t* __restrict__ aptr, const int32_t* __restrict__ acol,
                                                             float* __restrict__ v_cus, float* __restrict__ vt_cus,
                                                             float* __restrict__ v_art, float* __restrict__ vt_art,
                                                             int64_t n_label, const int64_t* __restrict__ label,
                                                             float* __restrict__ label_f, float4* __restrict__ zero4, int64_t n_zero4,
                                                             int32_t* __restrict__ counters) {
    const int64_t i = block_id * kBlock + threadIdx.x;
    if (counters && i < 4) counters[i] = 0;   // tickets of the last-workgroup reductions further down the iteration
    // round 4: the zero-fill of the gather-cat backward's outputs (rows no label edge names must read zero) rides here,
    // long before its consumer, instead of two fillBuffer launches in the middle of the backward
    for (int64_t z = i; z < n_zero4; z += n_blocks * kBlock) zero4[z] = mi_f4_zero();
    if (label && i < n_label) label_f[i] = (float)label[i];
    if (!mean) {
        if (i < nnz) v_cus[i] = vt_cus[i] = v_art[i] = vt_art[i] = 1.f;
        return;
    }
    // one thread per ROW of each CSR writes that row's entries (rows are short: fan-out-capped subgraphs)
    if (i < n_c) {
        const int32_t b = cptr[i], e = cptr[i + 1];
        const float inv = e > b ? 1.0f / (float)(e - b) : 0.f;
        for (int32_t p = b; p < e; ++p) {
            v_cus[p] = inv;
            const int32_t a = ccol[p];
            const int32_t da = aptr[a + 1] - aptr[a];
            vt_cus[p] = da > 0 ? 1.0f / (float)da : 0.f;
        }
    }
    if (i < n_a) {
        const int32_t b = aptr[i], e = aptr[i + 1];
        const float inv = e > b ? 1.0f / (float)(e - b) : 0.f;
        for (int32_t p = b; p < e; ++p) {
            v_art[p] = inv;
            const int32_t c = acol[p];
            const int32_t dc = cptr[c + 1] - cptr[c];
            vt_art[p] = dc > 0 ? 1.0f / (float)dc : 0.f;
        }
    }
}

#include "exec_common.hpp"

struct PrepArgs {
    int64_t nnz, n_c, n_a; int mean;
    const int32_t *cptr, *ccol, *aptr, *acol;
    float *v_cus, *vt_cus, *v_art, *vt_art;
    int64_t n_label; const int64_t* label; float* label_f;
    float4* zero4; int64_t n_zero4; int32_t* counters;
};
__device__ __forceinline__ void ranker_prep_run(const PrepArgs& a, int64_t block_id, int64_t n_blocks) {
    ranker_prep_body(block_id, n_blocks, a.nnz, a.n_c, a.n_a, a.mean, a.cptr, a.ccol, a.aptr, a.acol, a.v_cus, a.vt_cus, a.v_art, a.vt_art,
                     a.n_label, a.label, a.label_f, a.zero4, a.n_zero4, a.counters);
}
__global__ __launch_bounds__(kBlock) void ranker_prep_kernel(PrepArgs a) { ranker_prep_run(a, blockIdx.x, gridDim.x); }
// the prep work and the first encoder layer's two input dropouts (independent of each other) as ONE launch: workgroups
// [0, gp) prepare, [gp, gp + ga) drop the customer features, the rest the article features (round 4)
__global__ __launch_bounds__(kBlock) void ranker_prep_dropout_kernel(PrepArgs a, unsigned gp, DropSeg da, DropSeg db, unsigned ga, float p,
                                                                     float scale, uint32_t k0, uint32_t k1, uint32_t step_lo) {
    if (blockIdx.x < gp) {
        ranker_prep_run(a, blockIdx.x, gp);
        return;
    }
    const unsigned bid = blockIdx.x - gp;
    const bool first = bid < ga;
    dropout_segment(first ? da.n4 : db.n4, first ? da.x : db.x, first ? da.y : db.y, first ? da.site : db.site, first ? bid : bid - ga, p, scale,
                    k0, k1, step_lo);
}

// ---- the iteration --------------------------------------------------------------------------------------------------------

enum Mode { COUNT = 0, CHECK = 1, LAUNCH = 2 };

struct Exec {
    const mi_ranker_model& M;
    const mi_ranker_batch& B;
    MiArena ar;
    Mode mode;
    hipStream_t s;        // the caller's stream: everything that is on the critical path (the article side, every GEMM)
    hipStream_t s2;       // batch->aux_stream or s: the small customer-side twins of a pair of independent launches
    hipStream_t cur;      // where the helpers below enqueue
    int rc = 0;
    bool oom = false;

    Exec(const mi_ranker_model& m, const mi_ranker_batch& b, void* ws, size_t cap, Mode md, hipStream_t st)
        : M(m), B(b), ar(ws, cap), mode(md), s(st), s2(st), cur(st) {
        if (b.aux_stream && b.ev_fork && b.ev_join) s2 = (hipStream_t)b.aux_stream;
    }
    // Pairs of launches that do not depend on each other (customer / article twins: embeddings, dropout, the two
    // relations' aggregations, the two BatchNorms, the two gather-cat backward sums) are small and latency-bound; with an
    // auxiliary stream the customer twin runs beside the article twin.  fork(): aux waits for everything enqueued on s so
    // far; join(): s waits for everything enqueued on aux.  Without an auxiliary stream both are no-ops.
    void fork() {
        if (s2 == s || !go()) return;
        ok((int)hipEventRecord((hipEvent_t)B.ev_fork, s));
        ok((int)hipStreamWaitEvent(s2, (hipEvent_t)B.ev_fork, 0));
    }
    void join() {
        if (s2 == s || !go()) return;
        ok((int)hipEventRecord((hipEvent_t)B.ev_join, s2));
        ok((int)hipStreamWaitEvent(s, (hipEvent_t)B.ev_join, 0));
    }
    void on(int type) { cur = type == 0 ? s2 : s; }   // customer-side work to the auxiliary stream

    float* take(int64_t rows, int64_t cols) {
        const size_t n = (size_t)std::max<int64_t>(rows, 1) * (size_t)std::max<int64_t>(cols, 1);
        float* p = ar.take<float>(n);
        if (!p) oom = true;
        return p;
    }
    char* take_bytes(size_t n) {
        char* p = ar.take<char>(std::max<size_t>(n, 1));
        if (!p) oom = true;
        return p;
    }
    bool go() const { return mode == LAUNCH && rc == 0 && !oom; }
    void fail(int code) { if (rc == 0) rc = code; }
    void ok(int code) { if (code != 0 && rc == 0) rc = code; }

    // acc = A(csr, val) @ X  [+ in-place: S = addend + acc]; plan-less launches (per-batch subgraphs)
    void spmm(int64_t n_rows, int64_t d, const int32_t* rowptr, const int32_t* col, const float* val, const float* X, float* Y,
              const float* addend, float* S) {
        if (mode == CHECK && (d % 4 != 0 || d > 512)) fail(MI_ERR_UNSUPPORTED);
        if (!go()) return;
        ok(mi_spmm_csr_ex_f32(n_rows, d, rowptr, col, val, X, d, Y, d, addend, d, S, d, 1.0f, nullptr, nullptr, nullptr, 0,
                              (mi_stream_t)cur));
    }

    // one grouped launch (<= 8 problems); problems without a mask fall back to mi_gemm_f32 one by one when an operand
    // is not float4-addressable (the decoder's last layer: [n, 1] gradients), exactly as model/layers.py::_run_products
    void products(mi_gemm_problem* pr, int n) {
        const size_t wsb = mi_gemm_group_workspace_bytes(pr, n);
        char* w = wsb ? take_bytes(wsb) : nullptr;
        bool grouped = true;
        if (mode != COUNT) grouped = mi_gemm_group_supported(pr, n) != 0;
        size_t single_ws = 0;
        if (mode == COUNT || !grouped) {
            // COUNT cannot know (pointers are not final): reserve the per-product split-K scratch as well
            for (int i = 0; i < n; ++i) single_ws = std::max(single_ws, mi_gemm_workspace_bytes(pr[i].m, pr[i].n, pr[i].k));
        }
        char* w1 = single_ws ? take_bytes(single_ws) : nullptr;
        if (!grouped) {
            for (int i = 0; i < n; ++i)
                if (pr[i].a_mask || pr[i].k2 > 0) { fail(MI_ERR_UNSUPPORTED); return; }
        }
        if (!go()) return;
        if (grouped) {
            ok(mi_gemm_group_f32(pr, n, w, wsb, (mi_stream_t)s));
            return;
        }
        for (int i = 0; i < n && rc == 0; ++i) {
            const mi_gemm_problem& q = pr[i];
            const size_t need = mi_gemm_workspace_bytes(q.m, q.n, q.k);
            ok(mi_gemm_f32(q.trans_a, q.trans_b, q.m, q.n, q.k, q.A, q.lda, q.B, q.ldb, q.bias, q.C, q.ldc, q.accumulate, q.act,
                           need ? w1 : nullptr, need, (mi_stream_t)s));
        }
    }

    void dropout(const float* x, float* y, int64_t n, uint32_t site) {
        if (mode == CHECK && n % 4 != 0) fail(MI_ERR_UNSUPPORTED);
        if (!go() || n == 0) return;
        const float p = M.p_dropout;
        hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)mi_ceil_div(n / 4, kBlock)), dim3(kBlock), 0, cur, n / 4,
                           reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(y), p, 1.0f / (1.0f - p),
                           (uint32_t)B.seed, (uint32_t)(B.seed >> 32), site, (uint32_t)B.step);
        ok(mi_launch_status());
    }

    // Twin launches (csrc/pairs.hpp, round 4): the customer-side and the article-side twin of a step as ONE launch.  Only without
    // an auxiliary stream (the twins then run side by side anyway); MI_ERR_UNSUPPORTED from a pair launcher = the pair does not
    // share a kernel instantiation: the two single launches are issued instead.  Same device code per element either way.
    bool twin() const { return s2 == s; }
    bool paired(int prc) {   // true: the pair launch took the work
        if (prc == 0) return true;
        if (prc != MI_ERR_UNSUPPORTED) ok(prc);
        return prc != MI_ERR_UNSUPPORTED;
    }
    void dropout2(const float* xa, float* ya, int64_t na, uint32_t site_a, const float* xb, float* yb, int64_t nb, uint32_t site_b) {
        if (mode == CHECK && (na % 4 != 0 || nb % 4 != 0)) fail(MI_ERR_UNSUPPORTED);
        if (!go()) return;
        if (!twin() || na == 0 || nb == 0) {
            on(0); dropout(xa, ya, na, site_a);
            on(1); dropout(xb, yb, nb, site_b);
            return;
        }
        const float p = M.p_dropout;
        const DropSeg a{na / 4, reinterpret_cast<const float4*>(xa), reinterpret_cast<float4*>(ya), site_a};
        const DropSeg b{nb / 4, reinterpret_cast<const float4*>(xb), reinterpret_cast<float4*>(yb), site_b};
        const unsigned ga = (unsigned)mi_ceil_div(a.n4, kBlock), gb = (unsigned)mi_ceil_div(b.n4, kBlock);
        hipLaunchKernelGGL(dropout_pair_kernel, dim3(ga + gb), dim3(kBlock), 0, s, a, b, ga, p, 1.0f / (1.0f - p), (uint32_t)B.seed,
                           (uint32_t)(B.seed >> 32), (uint32_t)B.step);
        ok(mi_launch_status());
    }
    // two plan-less products (rows = destinations of relation r); in place: S = addend + acc
    void spmm2(const mi_pairs::SpmmSide (&q)[2], const int (&stream_of)[2]) {
        for (int r = 0; r < 2; ++r)
            if (mode == CHECK && (q[r].d % 4 != 0 || q[r].d > 512)) fail(MI_ERR_UNSUPPORTED);
        if (!go()) return;
        if (twin() && paired(mi_pairs::spmm_planless_pair(q[0], q[1], s))) return;
        for (int r = 0; r < 2; ++r) {
            on(stream_of[r]);
            ok(mi_spmm_csr_ex_f32(q[r].n_rows, q[r].d, q[r].rowptr, q[r].col, q[r].val, q[r].X, q[r].d, q[r].Y, q[r].d, q[r].addend, q[r].d,
                                  q[r].S, q[r].d, 1.0f, nullptr, nullptr, nullptr, 0, (mi_stream_t)cur));
        }
    }

    static mi_gemm_problem prob(int ta, int tb, int64_t m, int64_t n, int64_t k, const float* A, int64_t lda, const float* Bm,
                                int64_t ldb, float* C, int64_t ldc, const float* mask = nullptr, const float* bias = nullptr,
                                int act = 0) {
        mi_gemm_problem q;
        memset(&q, 0, sizeof(q));
        q.trans_a = ta; q.trans_b = tb; q.m = m; q.n = n; q.k = k;
        q.A = A; q.lda = lda; q.B = Bm; q.ldb = ldb; q.C = C; q.ldc = ldc;
        q.a_mask = mask; q.bias = bias; q.act = act;
        return q;
    }

    int run();
};

int Exec::run() {
    const int L = M.n_enc_layers, LD = M.n_dec_layers;
    const int64_t n[2] = {B.n_nodes[0], B.n_nodes[1]};
    const int64_t nnz = B.nnz, nl = B.n_label;
    // INFERENCE (batch->logits non-null, round 4): the forward launches only, in evaluation mode — no dropout, BatchNorm with
    // its running statistics — and the decoder's output per label edge written to batch->logits; gradient, optimizer, label
    // and loss fields are neither required nor read.
    const bool infer = B.logits != nullptr;
    const bool drop = M.p_dropout > 0.f && !infer;
    // relation r: src / dst type, forward CSR (rows = dst), backward CSR (rows = src)
    const int src_of[2] = {0, 1}, dst_of[2] = {1, 0};
    const int32_t* fptr[2] = {B.by_article_ptr, B.by_customer_ptr};
    const int32_t* fcol[2] = {B.by_article_col, B.by_customer_col};
    const int32_t* bptr[2] = {B.by_customer_ptr, B.by_article_ptr};
    const int32_t* bcol[2] = {B.by_customer_col, B.by_article_col};

    // The header is validated in EVERY mode: the walk below indexes the descriptor's fixed arrays with these counts, so the
    // counting pass must not run on a descriptor that names 9 layers or 40 columns either (round 4: found by the host
    // sanitizer build, tests/test_host_sanitizers.py — mi_ranker_step_workspace_bytes divided by a zero width read from
    // beyond conv[]).  mi_ranker_step_workspace_bytes returns 0 for such descriptors.
    {
        if (L < 1 || L > MI_RANKER_MAX_LAYERS || LD < 1 || LD > MI_RANKER_MAX_LAYERS) return MI_ERR_UNSUPPORTED;
        if (M.aggr != 0 && M.aggr != 1) return MI_ERR_UNSUPPORTED;
        if (!(M.p_dropout >= 0.f && M.p_dropout < 1.f)) return MI_ERR_BAD_ARG;
        if (nnz >= (1 << 20)) return MI_ERR_UNSUPPORTED;                       // the op-by-op path plans such adjacencies
        if (nl > mi_gather_cat_bwd_max_edges()) return MI_ERR_UNSUPPORTED;
        if (n[0] <= 0 || n[1] <= 0 || nl <= 0 || nnz < 0) return MI_ERR_UNSUPPORTED;
        if (M.n_ones < std::max(std::max(n[0], n[1]), nl) || !M.ones4) return MI_ERR_BAD_ARG;
        if (M.n_params < 0 || M.n_params > MI_RANKER_MAX_PARAMS) return MI_ERR_BAD_ARG;
        for (int t = 0; t < kTypes; ++t) {
            if (M.n_cols[t] < 1 || M.n_cols[t] > MI_RANKER_MAX_COLS || !B.x[t]) return MI_ERR_UNSUPPORTED;
            for (int c = 0; c < M.n_cols[t]; ++c)
                if (M.dims[t][c] < 1 || M.dims[t][c] > 4096 || M.table_rows[t][c] < 1 || !M.tables[t][c]) return MI_ERR_BAD_ARG;
        }
        if (!B.by_customer_ptr || !B.by_article_ptr || !B.label_row || !B.label_col) return MI_ERR_BAD_ARG;
        if (!infer && ((!B.label && !B.label_f32) || !B.loss)) return MI_ERR_BAD_ARG;
        if (nnz > 0 && (!B.by_customer_col || !B.by_article_col)) return MI_ERR_BAD_ARG;
        for (int l = 0; l < L; ++l)
            for (int r = 0; r < 2; ++r) {
                const mi_ranker_conv& cv = M.conv[l][r];
                if (cv.c_src < 1 || cv.c_dst < 1 || cv.c_out < 1 || cv.c_src > 4096 || cv.c_dst > 4096 || cv.c_out > 4096) return MI_ERR_UNSUPPORTED;
                if (!cv.w_l || !cv.w_r || (!infer && (!cv.gw_l || !cv.gw_r || (cv.b_l && !cv.gb_l)))) return MI_ERR_BAD_ARG;
            }
        for (int j = 0; j < LD; ++j) {
            const mi_ranker_linear& ln = M.dec[j];
            if (ln.in < 1 || ln.out < 1 || ln.in > 4096 || ln.out > 4096 || !ln.w || (!infer && (!ln.gw || (ln.b && !ln.gb)))) return MI_ERR_BAD_ARG;
        }
        for (int i = 0; i < M.n_params && !infer; ++i) {
            const mi_ranker_param& q = M.params[i];
            if (!q.p || !q.g || !q.m || !q.v || q.n < 0 || !mi_aligned16(q.p) || !mi_aligned16(q.g) || !mi_aligned16(q.m) || !mi_aligned16(q.v))
                return MI_ERR_UNSUPPORTED;
        }
    }

    // ---- embeddings (K7) ------------------------------------------------------------------------------------------
    int64_t width[2] = {0, 0};
    for (int t = 0; t < kTypes; ++t)
        for (int c = 0; c < M.n_cols[t]; ++c) width[t] += M.dims[t][c];
    float* x0[2];
    // (Measured and not kept, round 3: drawing the first layer's feature dropout inside the lookup, and the decoder's inside
    // gather_cat — three launches fewer, ~15 us — made the producers compute a Philox block per ELEMENT instead of one per
    // four: embed_concat 5.7 + 11 -> 19 + 33 us, gather_cat 4.8 -> 16 us, iteration 0.545 -> 0.62 ms.)
    fork();
    for (int t = 0; t < kTypes; ++t) x0[t] = take(n[t], width[t]);
    {
        bool done = false;
        if (go() && twin()) {
            const mi_pairs::EmbedSide ea{n[0], M.n_cols[0], B.x[0], M.tables[0], M.table_rows[0], M.dims[0], x0[0], width[0]};
            const mi_pairs::EmbedSide eb{n[1], M.n_cols[1], B.x[1], M.tables[1], M.table_rows[1], M.dims[1], x0[1], width[1]};
            done = paired(mi_pairs::embed_concat_pair(ea, eb, M.max_norm, s));
        }
        for (int t = 0; t < kTypes && !done; ++t) {
            on(t);
            if (go())
                ok(mi_embed_concat_f32(n[t], M.n_cols[t], B.x[t], M.tables[t], M.table_rows[t], M.dims[t], M.max_norm, x0[t], width[t],
                                       (mi_stream_t)cur));
        }
    }
    on(1);
    // the gather-cat backward's outputs: allocated here so that the prep launch below can zero-fill them (rows no label edge
    // names must read zero); C = the encoder's output width (both relations': checked below)
    const int64_t C0 = M.conv[L - 1][0].c_out;
    float* dz_early[2] = {take(n[0], C0), take(n[1], C0)};
    // ---- aggregation weights + labels
    float* v_cus = take(nnz, 1);
    float* vt_cus = take(nnz, 1);
    float* v_art = take(nnz, 1);
    float* vt_art = take(nnz, 1);
    float* label_f = take(nl, 1);
    int32_t* counters = reinterpret_cast<int32_t*>(take_bytes(64));   // tickets of the last-workgroup reductions; zeroed by the prep launch
    // the first encoder layer's dropped inputs: with twin launches their two dropouts ride in the prep launch (independent work)
    const bool drop0 = drop && L > 1;
    float* xin0[2] = {nullptr, nullptr};
    if (drop0)
        for (int t = 0; t < kTypes; ++t) xin0[t] = take(n[t], width[t]);
    if (mode == CHECK && drop0 && ((n[0] * width[0]) % 4 != 0 || (n[1] * width[1]) % 4 != 0)) fail(MI_ERR_UNSUPPORTED);
    const bool merged0 = drop0 && twin();
    if (go()) {
        const int64_t span = std::max(std::max(nnz, nl), std::max(n[0], n[1]));
        const PrepArgs pa{nnz, n[0], n[1], M.aggr == 1 ? 1 : 0, B.by_customer_ptr, B.by_customer_col, B.by_article_ptr, B.by_article_col,
                          v_cus, vt_cus, v_art, vt_art, nl, B.label_f32 ? nullptr : B.label, label_f,
                          reinterpret_cast<float4*>(dz_early[0]), (int64_t)((dz_early[1] + n[1] * C0) - dz_early[0]) / 4, counters};
        const unsigned gp = (unsigned)mi_ceil_div(span, kBlock);
        if (merged0) {
            const float p = M.p_dropout;
            const DropSeg da{n[0] * width[0] / 4, reinterpret_cast<const float4*>(x0[0]), reinterpret_cast<float4*>(xin0[0]), 0u};
            const DropSeg db{n[1] * width[1] / 4, reinterpret_cast<const float4*>(x0[1]), reinterpret_cast<float4*>(xin0[1]), 1u};
            const unsigned ga = (unsigned)mi_ceil_div(da.n4, kBlock), gb = (unsigned)mi_ceil_div(db.n4, kBlock);
            hipLaunchKernelGGL(ranker_prep_dropout_kernel, dim3(gp + ga + gb), dim3(kBlock), 0, s, pa, gp, da, db, ga, p, 1.0f / (1.0f - p),
                               (uint32_t)B.seed, (uint32_t)(B.seed >> 32), (uint32_t)B.step);
        } else {
            hipLaunchKernelGGL(ranker_prep_kernel, dim3(gp), dim3(kBlock), 0, s, pa);
        }
        ok(mi_launch_status());
    }
    const float* fval[2] = {v_art, v_cus};     // forward values of relation r (CSR by destination)
    const float* bval[2] = {vt_cus, vt_art};   // backward values of relation r (CSR by source)

    // ---- encoder forward ------------------------------------------------------------------------------------------
    float* xin[MI_RANKER_MAX_LAYERS][2];   // layer input after dropout
    float* agg[MI_RANKER_MAX_LAYERS][2];
    float* out[MI_RANKER_MAX_LAYERS][2];   // by relation: out[l][r] has n[dst_of[r]] rows
    int64_t cw[2] = {width[0], width[1]};  // current feature width per type
    float* xcur[2] = {x0[0], x0[1]};
    for (int l = 0; l < L; ++l) {
        const bool last = l == L - 1;
        // (layer 0: the auxiliary stream is still inside the fork opened for the embeddings; later layers open their own)
        if (l > 0) fork();
        for (int t = 0; t < kTypes; ++t) {
            xin[l][t] = xcur[t];
            if (!last && drop) xin[l][t] = l == 0 ? xin0[t] : take(n[t], cw[t]);
        }
        if (!last && drop && !(l == 0 && merged0))
            dropout2(xcur[0], xin[l][0], n[0] * cw[0], (uint32_t)(l * 2 + 0), xcur[1], xin[l][1], n[1] * cw[1], (uint32_t)(l * 2 + 1));
        on(1);
        join();    // each aggregation reads the OTHER type's input
        fork();
        mi_gemm_problem pr[2];
        mi_pairs::SpmmSide fq[2];
        for (int r = 0; r < 2; ++r) {
            const mi_ranker_conv& cv = M.conv[l][r];
            const int st = src_of[r], dt = dst_of[r];
            if (mode == CHECK && (cv.c_src != cw[st] || cv.c_dst != cw[dt] || !cv.w_l || !cv.w_r ||
                                  (!infer && (!cv.gw_l || !cv.gw_r || (cv.b_l && !cv.gb_l))) || cv.c_src % 4 || cv.c_dst % 4 || cv.c_out % 4))
                return MI_ERR_UNSUPPORTED;
            agg[l][r] = take(n[dt], cv.c_src);
            out[l][r] = take(n[dt], cv.c_out);
            fq[r] = mi_pairs::SpmmSide{n[dt], cv.c_src, fptr[r], fcol[r], fval[r], xin[l][st], agg[l][r], nullptr, nullptr};
            pr[r] = prob(0, 1, n[dt], cv.c_out, cv.c_src, agg[l][r], cv.c_src, cv.w_l, cv.c_src, out[l][r], cv.c_out, nullptr,
                         cv.b_l, last ? 0 : 1);
            pr[r].k2 = cv.c_dst; pr[r].A2 = xin[l][dt]; pr[r].lda2 = cv.c_dst; pr[r].B2 = cv.w_r; pr[r].ldb2 = cv.c_dst;
        }
        spmm2(fq, dst_of);   // relation 1 (destination: customers) is the small one
        on(1);
        join();
        products(pr, 2);
        for (int r = 0; r < 2; ++r) {
            xcur[dst_of[r]] = out[l][r];
            cw[dst_of[r]] = M.conv[l][r].c_out;
        }
    }
    if (mode == CHECK && (cw[0] != cw[1] || cw[0] != C0)) return MI_ERR_UNSUPPORTED;
    const int64_t C = cw[0];
    // ---- BatchNorm (K8) -------------------------------------------------------------------------------------------
    float* z[2] = {xcur[0], xcur[1]};
    float* zpre[2] = {xcur[0], xcur[1]};
    float *bn_mean[2] = {nullptr, nullptr}, *bn_inv[2] = {nullptr, nullptr};
    char* bn_ws[2] = {nullptr, nullptr};
    const size_t bn_ws_bytes = mi_batchnorm_workspace_bytes(C);
    if (M.batch_normalize) {
        fork();
        for (int t = 0; t < kTypes; ++t) {
            bn_ws[t] = take_bytes(bn_ws_bytes);
            on(t);
            const mi_ranker_norm& bn = M.norm[t];
            if (mode == CHECK && ((bn.gamma && (!bn.beta || (!infer && (!bn.g_gamma || !bn.g_beta)))) || C > 512)) return MI_ERR_UNSUPPORTED;
            if (mode == CHECK && infer && (!bn.running_mean || !bn.running_var)) return MI_ERR_UNSUPPORTED;   // evaluation mode needs the statistics
            z[t] = take(n[t], C);
            bn_mean[t] = take(C, 1);
            bn_inv[t] = take(C, 1);
        }
        bool done = false;
        if (go() && twin() && !infer) {
            mi_pairs::BnSide q[2];
            for (int t = 0; t < kTypes; ++t) {
                const mi_ranker_norm& bn = M.norm[t];
                q[t] = mi_pairs::BnSide{n[t], zpre[t], bn.gamma, bn.beta, bn.running_mean, bn.running_var, bn.momentum, bn.eps,
                                        bn_mean[t], bn_inv[t], z[t], nullptr, nullptr, nullptr, nullptr, bn_ws[t]};
            }
            done = paired(mi_pairs::batchnorm_fwd_pair(q[0], q[1], C, s));
        }
        for (int t = 0; t < kTypes && !done; ++t) {
            const mi_ranker_norm& bn = M.norm[t];
            on(t);
            if (go())
                ok(mi_batchnorm_fwd_f32(n[t], C, zpre[t], C, bn.gamma, bn.beta, bn.running_mean, bn.running_var, bn.momentum, bn.eps,
                                        infer ? 0 : 1, bn_mean[t], bn_inv[t], z[t], C, bn_ws[t], bn_ws_bytes, (mi_stream_t)cur));
        }
        on(1);
        join();
    }
    // ---- decoder forward (b7) -------------------------------------------------------------------------------------
    float* h = take(nl, 2 * C);
    // with dropout in front of the first decoder layer the gather writes the DROPPED rows directly (pairs.hpp: gather_cat_dropout;
    // the undropped concatenation has no other reader): one launch instead of gather + dropout
    bool gc_fused = false;
    if (go() && drop && LD > 1)
        gc_fused = paired(mi_pairs::gather_cat_dropout(nl, C, C, B.label_row, B.label_col, z[0], z[1], h, M.p_dropout, B.seed, 64u,
                                                       (uint32_t)B.step, s));
    if (go() && !gc_fused) ok(mi_gather_cat_f32(nl, C, C, B.label_row, B.label_col, z[0], C, z[1], C, h, 2 * C, (mi_stream_t)s));
    float* din[MI_RANKER_MAX_LAYERS];
    float* dout[MI_RANKER_MAX_LAYERS];
    int64_t hw = 2 * C;
    for (int j = 0; j < LD; ++j) {
        const mi_ranker_linear& ln = M.dec[j];
        const bool last = j == LD - 1;
        if (mode == CHECK && (ln.in != hw || !ln.w || (!infer && (!ln.gw || (ln.b && !ln.gb))) || (last && ln.out != 1))) return MI_ERR_UNSUPPORTED;
        din[j] = h;
        if (!last && drop) {
            din[j] = take(nl, hw);
            if (j == 0 && mode == LAUNCH && gc_fused) din[j] = h;   // h already holds the dropped rows (the spare buffer stays unused)
            else dropout(h, din[j], nl * hw, (uint32_t)(64 + j));
        }
        dout[j] = (infer && last) ? B.logits : take(nl, ln.out);
        const size_t need = mi_gemm_workspace_bytes(nl, ln.out, hw);
        char* w = need ? take_bytes(need) : nullptr;
        if (go())
            ok(mi_gemm_f32(0, 1, nl, ln.out, hw, din[j], hw, ln.w, hw, ln.b, dout[j], ln.out, 0, last ? 0 : 1, w, need, (mi_stream_t)s));
        h = dout[j];
        hw = ln.out;
    }
    if (infer) return oom ? MI_ERR_WORKSPACE : rc;   // the logits are where the caller asked for them
    // ---- loss (b9)
    float* dlogits = take(nl, 1);
    if (go()) ok(mi_bce_logits_f32(nl, h, B.label_f32 ? B.label_f32 : label_f, B.loss, dlogits, (mi_stream_t)s));

    // ---- decoder backward -----------------------------------------------------------------------------------------
    AdamTable tb;
    memset(&tb, 0, sizeof(tb));
    auto grad_src = [&](const float* param, const float* src, int stride, float* dense) {
        // a bias gradient lives in column 0 of a [n, 4] product: the Adam launch reads it there and copies it out
        for (int i = 0; i < M.n_params; ++i)
            if (M.params[i].p == param) {
                tb.p[i].g = const_cast<float*>(src);
                tb.g_stride[i] = stride;
                tb.g_dst[i] = dense;
            }
    };
    float* dh = dlogits;
    for (int j = LD - 1; j >= 0; --j) {
        const mi_ranker_linear& ln = M.dec[j];
        const bool last = j == LD - 1;
        const float* mask = last ? nullptr : dout[j];
        float* dx = take(nl, ln.in);
        if (last && ln.out == 1) {   // Linear(in, 1): mi_linear1_bwd_f32 (as FusedRankerStep does)
            const size_t need = mi_linear1_bwd_workspace_bytes(nl, ln.in);
            char* w = take_bytes(need);
            if (go() && !paired(mi_pairs::linear1_bwd_lastblock(nl, ln.in, dh, ln.w, din[j], dx, ln.gw, ln.b ? ln.gb : nullptr, w, need,
                                                                 counters + 0, s)))
                ok(mi_linear1_bwd_f32(nl, ln.in, dh, ln.w, din[j], ln.in, dx, ln.in, ln.gw, ln.b ? ln.gb : nullptr, w, need,
                                      (mi_stream_t)s));
            dh = dx;
            continue;   // the last layer has no dropout in front of its output and no relu behind it
        }
        mi_gemm_problem pr[3];
        int np = 0;
        pr[np++] = prob(0, 0, nl, ln.in, ln.out, dh, ln.out, ln.w, ln.in, dx, ln.in, mask);
        pr[np++] = prob(1, 0, ln.out, ln.in, nl, dh, ln.out, din[j], ln.in, ln.gw, ln.in, mask);
        if (ln.b) {
            float* db4 = take(ln.out, 4);
            pr[np++] = prob(1, 0, ln.out, 4, nl, dh, ln.out, M.ones4, 4, db4, 4, mask);
            grad_src(ln.b, db4, 4, ln.gb);
        }
        products(pr, np);
        dh = dx;
        if (!last && drop) dropout(dx, dx, nl * ln.in, (uint32_t)(64 + j));   // the forward's mask, regenerated
    }
    // ---- gather-cat backward
    float* dz[2] = {dz_early[0], dz_early[1]};   // zero-filled by the prep launch
    fork();
    {
        const bool done = go() && twin() && paired(mi_pairs::gather_cat_bwd_pair(nl, C, B.label_row, B.label_col, dh, 2 * C, dz[0], dz[1], s));
        for (int t = 0; t < kTypes && !done; ++t) {
            on(t);
            if (go())
                ok(mi_gather_cat_bwd_f32(nl, C, t == 0 ? 0 : C, t == 0 ? B.label_row : B.label_col, dh, 2 * C, dz[t], C, (mi_stream_t)cur));
        }
    }
    on(1);
    // ---- BatchNorm backward
    if (M.batch_normalize) {   // still inside the fork: each type's chain is gather-cat backward -> BatchNorm backward
        float* dxb[2] = {take(n[0], C), take(n[1], C)};
        bool done = false;
        if (go() && twin()) {
            mi_pairs::BnSide q[2];
            for (int t = 0; t < kTypes; ++t) {
                const mi_ranker_norm& bn = M.norm[t];
                q[t] = mi_pairs::BnSide{n[t], zpre[t], bn.gamma, nullptr, nullptr, nullptr, 0.f, 0.f, bn_mean[t], bn_inv[t], nullptr,
                                        dz[t], dxb[t], bn.gamma ? bn.g_gamma : nullptr, bn.gamma ? bn.g_beta : nullptr, bn_ws[t]};
            }
            done = paired(mi_pairs::batchnorm_bwd_pair(q[0], q[1], C, s));
        }
        for (int t = 0; t < kTypes; ++t) {
            const mi_ranker_norm& bn = M.norm[t];
            on(t);
            if (go() && !done)
                ok(mi_batchnorm_bwd_f32(n[t], C, zpre[t], C, dz[t], C, bn.gamma, bn_mean[t], bn_inv[t], dxb[t], C, bn.gamma ? bn.g_gamma : nullptr,
                                        bn.gamma ? bn.g_beta : nullptr, bn_ws[t], bn_ws_bytes, (mi_stream_t)cur));
            dz[t] = dxb[t];
        }
        on(1);
    }
    join();
    // ---- encoder backward -----------------------------------------------------------------------------------------
    float* dxs[2] = {dz[0], dz[1]};
    mi_wgrad_problem wq_all[4];
    int n_wq = 0;
    for (int l = L - 1; l >= 0; --l) {
        const bool last = l == L - 1, need_x = l > 0;   // layer 0 reads frozen embeddings
        const float* dy[2];
        const float* mask[2];
        for (int r = 0; r < 2; ++r) {
            dy[r] = dxs[dst_of[r]];
            mask[r] = last ? nullptr : out[l][r];
        }
        float* dxn[2] = {nullptr, nullptr};
        if (need_x) {
            mi_gemm_problem pr[4];
            float* dagg[2];
            int np = 0;
            for (int r = 0; r < 2; ++r) {
                const mi_ranker_conv& cv = M.conv[l][r];
                const int dt = dst_of[r];
                dagg[r] = take(n[dt], cv.c_src);
                float* dxd = take(n[dt], cv.c_dst);
                pr[np++] = prob(0, 0, n[dt], cv.c_src, cv.c_out, dy[r], cv.c_out, cv.w_l, cv.c_src, dagg[r], cv.c_src, mask[r]);
                pr[np++] = prob(0, 0, n[dt], cv.c_dst, cv.c_out, dy[r], cv.c_out, cv.w_r, cv.c_dst, dxd, cv.c_dst, mask[r]);
                dxn[dt] = dxd;
            }
            products(pr, np);
            fork();
            mi_pairs::SpmmSide bq[2];
            for (int r = 0; r < 2; ++r) {   // dX_src += A^T dAgg, in the epilogue
                const mi_ranker_conv& cv = M.conv[l][r];
                const int st = src_of[r];
                bq[r] = mi_pairs::SpmmSide{n[st], cv.c_src, bptr[r], bcol[r], bval[r], dagg[r], nullptr, dxn[st], dxn[st]};
            }
            spmm2(bq, src_of);
            on(1);
        }
        // the weight gradients need nothing of the aggregations above: they run beside them.  Round 3: both relations'
        // three products each as ONE launch of the LDS-free transposed-product kernel (csrc/wgrad.hip) where its shapes
        // allow (c_out a multiple of 32, <= 128); else the grouped GEMM as before.  FusedRankerStep makes the same choice
        // (model/layers.py::hetero_layer_backward), so the two paths stay bitwise equal.
        mi_wgrad_problem wq[2];
        for (int r = 0; r < 2; ++r) {
            const mi_ranker_conv& cv = M.conv[l][r];
            const int dt = dst_of[r];
            memset(&wq[r], 0, sizeof(wq[r]));
            wq[r].k = n[dt]; wq[r].m = (int32_t)cv.c_out; wq[r].n1 = (int32_t)cv.c_src; wq[r].n2 = (int32_t)cv.c_dst;
            wq[r].dy = dy[r]; wq[r].mask = mask[r]; wq[r].b1 = agg[l][r]; wq[r].b2 = xin[l][dt];
            wq[r].gw1 = cv.gw_l; wq[r].gb = cv.b_l ? cv.gb_l : nullptr; wq[r].gw2 = cv.gw_r;
        }
        if (mi_sage_wgrad_supported(wq, 2)) {   // depends on the dimensions and on pointers being non-null only: same answer in every pass
            if (L <= 2) {   // round 4: nothing but Adam reads the weight gradients — both layers' products go out as ONE launch at the end
                wq_all[n_wq++] = wq[0];
                wq_all[n_wq++] = wq[1];
            } else {
                const size_t need = mi_sage_wgrad_workspace_bytes(wq, 2);
                char* w = take_bytes(need);
                if (go()) ok(mi_sage_wgrad_f32(wq, 2, w, need, (mi_stream_t)s));
            }
            if (need_x) {
                for (int t = 0; t < kTypes; ++t) {
                    dxs[t] = dxn[t];
                    on(t);
                    if (!last && drop) dropout(dxn[t], dxn[t], n[t] * M.conv[l][t].c_src, (uint32_t)(l * 2 + t));
                }
                on(1);
                join();
            }
            continue;
        }
        mi_gemm_problem pw[6];
        int nw = 0;
        for (int r = 0; r < 2; ++r) {
            const mi_ranker_conv& cv = M.conv[l][r];
            const int dt = dst_of[r];
            pw[nw++] = prob(1, 0, cv.c_out, cv.c_src, n[dt], dy[r], cv.c_out, agg[l][r], cv.c_src, cv.gw_l, cv.c_src, mask[r]);
            if (cv.b_l) {
                float* db4 = take(cv.c_out, 4);
                pw[nw++] = prob(1, 0, cv.c_out, 4, n[dt], dy[r], cv.c_out, M.ones4, 4, db4, 4, mask[r]);
                grad_src(cv.b_l, db4, 4, cv.gb_l);
            }
            pw[nw++] = prob(1, 0, cv.c_out, cv.c_dst, n[dt], dy[r], cv.c_out, xin[l][dt], cv.c_dst, cv.gw_r, cv.c_dst, mask[r]);
        }
        products(pw, nw);
        if (need_x) {
            for (int t = 0; t < kTypes; ++t) {
                dxs[t] = dxn[t];
                on(t);
                // the mask this layer's forward drew for its input of type t (relation t is the one whose source is t)
                if (!last && drop) dropout(dxn[t], dxn[t], n[t] * M.conv[l][t].c_src, (uint32_t)(l * 2 + t));
            }
            on(1);
            join();
        }
    }
    if (n_wq > 0) {
        const size_t need = mi_sage_wgrad_workspace_bytes(wq_all, n_wq);
        char* w = take_bytes(need);
        if (go()) ok(mi_sage_wgrad_f32(wq_all, n_wq, w, need, (mi_stream_t)s));
    }
    if (oom) return MI_ERR_WORKSPACE;
    if (rc) return rc;
    // ---- Adam over the parameter list (a9's arithmetic) + the BatchNorm batch counters ------------------------------------
    if (mode != LAUNCH) return 0;
    int64_t longest = 1;
    for (int i = 0; i < M.n_params; ++i) {
        const float* strided = tb.p[i].g;
        tb.p[i] = M.params[i];
        if (strided) tb.p[i].g = const_cast<float*>(strided); else tb.g_stride[i] = 1;
        longest = std::max(longest, M.params[i].n);
    }
    tb.n = M.n_params;
    int64_t* nbt0 = M.batch_normalize ? M.norm[0].num_batches_tracked : nullptr;
    int64_t* nbt1 = M.batch_normalize ? M.norm[1].num_batches_tracked : nullptr;
    if (M.n_params > 0 || nbt0 || nbt1) {
        const MiAdamConsts c = mi_adam_consts(M.lr, M.beta1, M.beta2, M.eps, M.step > 0 ? M.step : 1);
        const unsigned gx = (unsigned)std::min<int64_t>(mi_ceil_div(longest, kBlock), 64);
        hipLaunchKernelGGL(adam_multi_kernel, dim3(gx, (unsigned)std::max(M.n_params, 1)), dim3(kBlock), 0, s, tb, c,
                           (M.apply_adam && M.n_params > 0) ? 1 : 0, nbt0, nbt1, 1.f);
        ok(mi_launch_status());
    }
    return rc;
}

}  // namespace

extern "C" int64_t mi_ranker_sizeof(int32_t which) {
    switch (which) {
        case 0: return (int64_t)sizeof(mi_ranker_model);
        case 1: return (int64_t)sizeof(mi_ranker_batch);
        case 2: return (int64_t)sizeof(mi_ranker_conv);
        case 3: return (int64_t)sizeof(mi_ranker_norm);
        case 4: return (int64_t)sizeof(mi_ranker_linear);
        case 5: return (int64_t)sizeof(mi_ranker_param);
        default: return -1;
    }
}

extern "C" size_t mi_ranker_step_workspace_bytes(const mi_ranker_model* model, const mi_ranker_batch* batch) {
    if (!model || !batch) return 0;
    // counting pass over a fictitious base: only offsets matter
    Exec e(*model, *batch, reinterpret_cast<void*>((uintptr_t)4096), (size_t)1 << 46, COUNT, nullptr);
    if (e.run() != 0) return 0;   // a descriptor the executor does not take has no workspace size
    return e.ar.off + 4096;
}

extern "C" int mi_ranker_step_check(const mi_ranker_model* model, const mi_ranker_batch* batch, void* ws, size_t ws_bytes) {
    MI_CHECK_ARG(model && batch && ws && mi_aligned16(ws));
    Exec chk(*model, *batch, ws, ws_bytes, CHECK, nullptr);   // validates every operand; enqueues nothing, touches no memory
    return chk.run();
}

extern "C" int mi_ranker_step_f32(const mi_ranker_model* model, const mi_ranker_batch* batch, void* ws, size_t ws_bytes,
                                  mi_stream_t stream) {
    MI_CHECK_ARG(model && batch && ws && mi_aligned16(ws));
    {
        Exec chk(*model, *batch, ws, ws_bytes, CHECK, (hipStream_t)stream);
        const int rc = chk.run();
        if (rc) return rc;
    }
    Exec run(*model, *batch, ws, ws_bytes, LAUNCH, (hipStream_t)stream);
    return run.run();
}

extern "C" int mi_ranker_adam_f32(const mi_ranker_model* model, float grad_scale, mi_stream_t stream) {
    MI_CHECK_ARG(model && model->n_params >= 0 && model->n_params <= MI_RANKER_MAX_PARAMS && grad_scale > 0.f);
    if (model->n_params == 0) return 0;
    AdamTable tb;
    int64_t longest = 1;
    for (int i = 0; i < model->n_params; ++i) {
        const mi_ranker_param& q = model->params[i];
        MI_CHECK_ARG(q.p && q.g && q.m && q.v && q.n >= 0);
        tb.p[i] = q;
        tb.g_stride[i] = 1;
        tb.g_dst[i] = nullptr;
        longest = std::max(longest, q.n);
    }
    tb.n = model->n_params;
    const MiAdamConsts c = mi_adam_consts(model->lr, model->beta1, model->beta2, model->eps, model->step > 0 ? model->step : 1);
    const unsigned gx = (unsigned)std::min<int64_t>(mi_ceil_div(longest, kBlock), 64);
    hipLaunchKernelGGL(adam_multi_kernel, dim3(gx, (unsigned)model->n_params), dim3(kBlock), 0, (hipStream_t)stream, tb, c, 1,
                       (int64_t*)nullptr, (int64_t*)nullptr, grad_scale);
    return mi_launch_status();
}

// N5: native executor of one PinSAGE training iteration (include/laplace_hip.h, mi_pinsage_step_f32).
//
// The reference's loop body (pinsage/model.py:118-131 over pinsage/layers.py:121-203) at its default batch (32 pairs:
// ~100 seeds, ~400 / ~1 500 nodes in the two blocks, hidden 64) is ~200 launches of 3-8 us when it is issued op by op
// through autograd (embedding lookups and their index_add_ backward, dropout, row norms, gathers of the pair endpoints and
// their sort-based index_put backward ...): 1.8 ms per iteration, almost all of it launch overhead.  This file issues the
// iteration as ~40 launches from one C call, the way ranker_exec.hip does for the ranker: the products on the grouped MFMA
// GEMM (relu and relu-backward masks fused), the weighted neighbourhood sums on the SpMM kernel over the block CSRs the
// device sampler emitted, and a handful of small kernels of its own (below).  COUNT / CHECK / LAUNCH passes as there.
//
// Facts of the batch layout that the code relies on (mi_pinsage_sample_batch guarantees them; laplace_amd/pinsage/
// sampler.py builds the same layout on its index-op path): a block's destination nodes are the first n_dst rows of its
// src_ids, and block l's destinations are block l + 1's sources — so the seeds are the first n_seeds rows of block 0's
// src_ids, h_dst is a prefix of h_src, and the projector rows of the seeds are a prefix of the gathered input rows.
#include "common.hpp"
#include <algorithm>

extern "C" int mi_gemm_group_supported(const mi_gemm_problem* problems, int32_t n);

namespace {

constexpr int kBlock = 256;

#include "exec_common.hpp"

// h[r, :] = table[ids[r], :]  (LinearProjector over the id feature: one embedding row per node)
__global__ __launch_bounds__(kBlock) void pin_embed_rows_kernel(int64_t n, int h4, const int64_t* __restrict__ ids,
                                                                const float4* __restrict__ table, float4* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n * h4) return;
    const int64_t r = i / h4;
    const int c = (int)(i % h4);
    out[i] = table[ids[r] * h4 + c];
}

// cd = dropout([agg, h_dst]) as one [n, 2 H] matrix; the mask is keyed on the element's position in it, so the backward
// regenerates it with dropout_kernel over the flat gradient.  p = 0: the plain concatenation.
__global__ __launch_bounds__(kBlock) void pin_cat_dropout_kernel(int64_t n, int h4, const float4* __restrict__ agg,
                                                                 const float4* __restrict__ hdst, float4* __restrict__ out,
                                                                 float p, float scale, uint32_t k0, uint32_t k1, uint32_t site,
                                                                 uint32_t step_lo) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n * 2 * h4) return;
    const int64_t r = i / (2 * h4);
    const int c = (int)(i % (2 * h4));
    float4 v = c < h4 ? agg[r * h4 + c] : hdst[r * h4 + (c - h4)];
    if (p > 0.f) {
        const MiPhilox rn = mi_philox4x32((uint32_t)i, (uint32_t)((uint64_t)i >> 32), site, step_lo, k0, k1);
        const uint32_t thr = (uint32_t)fminf(4294967040.f, p * 4294967296.f);
        v.x = rn.c[0] >= thr ? v.x * scale : 0.f;
        v.y = rn.c[1] >= thr ? v.y * scale : 0.f;
        v.z = rn.c[2] >= thr ? v.z * scale : 0.f;
        v.w = rn.c[3] >= thr ? v.w * scale : 0.f;
    }
    out[i] = v;
}

// Row L2-normalisation (pinsage/layers.py:151-154): h = z / ||z||, rows of norm 0 divided by 1.  One wavefront per row.
__global__ __launch_bounds__(kBlock) void pin_l2norm_fwd_kernel(int64_t n, int h, const float* __restrict__ z, float* __restrict__ out,
                                                                float* __restrict__ inv_norm) {
    const int64_t r = (int64_t)blockIdx.x * (kBlock / MI_WAVE) + threadIdx.x / MI_WAVE;
    if (r >= n) return;
    const int lane = mi_lane();
    float ss = 0.f;
    for (int c = lane; c < h; c += MI_WAVE) {
        const float x = z[r * h + c];
        ss += x * x;
    }
    ss = mi_wave_sum(ss);
    const float nrm = sqrtf(ss);
    const float inv = nrm == 0.f ? 1.f : 1.f / nrm;
    for (int c = lane; c < h; c += MI_WAVE) out[r * h + c] = z[r * h + c] * inv;
    if (lane == 0) inv_norm[r] = inv;
}

// Backward of the normalisation and of the relu in front of it: dz = (dh - h (h . dh)) * inv where z > 0 (h > 0), else 0.
// Rows of norm 0: h = 0 everywhere, masked out entirely, as in torch.
__global__ __launch_bounds__(kBlock) void pin_l2norm_bwd_kernel(int64_t n, int h, const float* __restrict__ hn,
                                                                const float* __restrict__ inv_norm, const float* __restrict__ dh,
                                                                float* __restrict__ dz) {
    const int64_t r = (int64_t)blockIdx.x * (kBlock / MI_WAVE) + threadIdx.x / MI_WAVE;
    if (r >= n) return;
    const int lane = mi_lane();
    float dot = 0.f;
    for (int c = lane; c < h; c += MI_WAVE) dot += hn[r * h + c] * dh[r * h + c];
    dot = mi_wave_sum(dot);
    const float inv = inv_norm[r];
    for (int c = lane; c < h; c += MI_WAVE) {
        const float y = hn[r * h + c];
        dz[r * h + c] = y > 0.f ? (dh[r * h + c] - y * dot) * inv : 0.f;
    }
}

// Scores of the pairs (ItemToItemScorer: dot of the endpoints' representations + both biases) and the hinge margin.
// hf = hd + hN is formed on the fly (hd = the seeds' projector rows = a prefix of the input rows).  One wavefront per pair.
__global__ __launch_bounds__(kBlock) void pin_score_kernel(int64_t n_pairs, int h, const float* __restrict__ hd,
                                                           const float* __restrict__ hN, const int64_t* __restrict__ seeds,
                                                           const int64_t* __restrict__ pu, const int64_t* __restrict__ pv,
                                                           const int64_t* __restrict__ nv, const float* __restrict__ bias,
                                                           float* __restrict__ margin) {
    const int64_t p = (int64_t)blockIdx.x * (kBlock / MI_WAVE) + threadIdx.x / MI_WAVE;
    if (p >= n_pairs) return;
    const int lane = mi_lane();
    const int64_t u = pu[p], v = pv[p], w = nv[p];
    float dp = 0.f, dn = 0.f;
    for (int c = lane; c < h; c += MI_WAVE) {
        const float xu = hd[u * h + c] + hN[u * h + c];
        dp += xu * (hd[v * h + c] + hN[v * h + c]);
        dn += xu * (hd[w * h + c] + hN[w * h + c]);
    }
    dp = mi_wave_sum(dp);
    dn = mi_wave_sum(dn);
    if (lane == 0) {
        const float bu = bias[seeds[u]];
        const float pos = dp + bu + bias[seeds[v]];
        const float neg = dn + bu + bias[seeds[w]];
        margin[p] = neg - pos + 1.f;
    }
}

// Gradient of the mean hinge with respect to hf (one block per seed row, the pairs walked in order: deterministic) and to
// the scorer bias (dense buffer, the seed's own entry: seeds are distinct); block 0 also reduces the loss.
//   active pair p (margin > 0), g = 1 / n_pairs:  d hf[u] += g (hf[w] - hf[v]),  d hf[w] += g hf[u],  d hf[v] -= g hf[u],
//   d bias[seed(w)] += g,  d bias[seed(v)] -= g  (the head's bias enters both scores and cancels).
__global__ __launch_bounds__(128) void pin_score_grad_kernel(int64_t n_seeds, int64_t n_pairs, int h, const float* __restrict__ hd,
                                                             const float* __restrict__ hN, const int64_t* __restrict__ seeds,
                                                             const int64_t* __restrict__ pu, const int64_t* __restrict__ pv,
                                                             const int64_t* __restrict__ nv, const float* __restrict__ margin,
                                                             float* __restrict__ dhf, float* __restrict__ g_bias,
                                                             float* __restrict__ bias_out /* compact, or null */,
                                                             float* __restrict__ loss) {
    __shared__ float red[128];
    __shared__ int match[1024];    // pairs that touch this seed, in pair order (n_pairs <= 1024: the sampler's batch limit)
    __shared__ int wave_cnt[2];
    const int64_t s = blockIdx.x;
    const int c = threadIdx.x;   // blockDim = 128 >= h
    const float g = 1.0f / (float)n_pairs;
    // (1) the active pairs with this seed at one of their three ends: 128 pairs per round, coalesced reads, ordered
    //     compaction by ballot — walking all pairs one by one per seed was 2 ms at 1 024 pairs per batch
    int n_match = 0;
    for (int64_t base = 0; base < n_pairs; base += 128) {
        const int64_t p = base + c;
        bool hit = false;
        if (p < n_pairs && margin[p] > 0.f) hit = pu[p] == s || pv[p] == s || nv[p] == s;
        const unsigned long long m = __ballot(hit);
        const int lane = c & 63, wave = c >> 6;
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        __syncthreads();
        const int before = (wave ? wave_cnt[0] : 0) + __popcll(m & ((1ull << lane) - 1ull));
        if (hit) match[n_match + before] = (int)p;
        n_match += wave_cnt[0] + wave_cnt[1];
        __syncthreads();
    }
    // (2) their contributions, in pair order
    float acc = 0.f, gb = 0.f;
    for (int i = 0; i < n_match; ++i) {
        const int p = match[i];
        const int64_t u = pu[p], v = pv[p], w = nv[p];
        if (c < h) {
            const float xu = hd[u * h + c] + hN[u * h + c];
            if (u == s) acc += g * ((hd[w * h + c] + hN[w * h + c]) - (hd[v * h + c] + hN[v * h + c]));
            if (w == s) acc += g * xu;
            if (v == s) acc -= g * xu;
        }
        if (w == s) gb += g;
        if (v == s) gb -= g;
    }
    if (c < h) dhf[s * h + c] = acc;
    if (c == 0) {
        if (bias_out) bias_out[s] = gb; else g_bias[seeds[s]] = gb;
    }
    if (s == 0) {   // the loss: fixed-order tree over the pairs
        float part = 0.f;
        for (int64_t p = c; p < n_pairs; p += 128) part += fmaxf(margin[p], 0.f);
        red[c] = part;
        __syncthreads();
        for (int off = 64; off > 0; off >>= 1) {
            if (c < off) red[c] += red[c + off];
            __syncthreads();
        }
        if (c == 0) loss[0] = red[0] * g;
    }
}

// dh[r, :] = dropout_backward(da)[r, :] (+ dc[r, H:2H] for r < n_dst: h_dst is the prefix of h_src and received its own
// gradient through the concatenation).  da's mask: the forward's, keyed on the position in [n_src, H].
__global__ __launch_bounds__(kBlock) void pin_dh_merge_kernel(int64_t n_src, int64_t n_dst, int h4, const float4* __restrict__ da,
                                                              const float4* __restrict__ dc /*[n_dst, 2 H]*/, float4* __restrict__ dh,
                                                              float p, float scale, uint32_t k0, uint32_t k1, uint32_t site,
                                                              uint32_t step_lo) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_src * h4) return;
    float4 v = da[i];
    if (p > 0.f) {
        const MiPhilox rn = mi_philox4x32((uint32_t)i, (uint32_t)((uint64_t)i >> 32), site, step_lo, k0, k1);
        const uint32_t thr = (uint32_t)fminf(4294967040.f, p * 4294967296.f);
        v.x = rn.c[0] >= thr ? v.x * scale : 0.f;
        v.y = rn.c[1] >= thr ? v.y * scale : 0.f;
        v.z = rn.c[2] >= thr ? v.z * scale : 0.f;
        v.w = rn.c[3] >= thr ? v.w * scale : 0.f;
    }
    const int64_t r = i / h4;
    if (r < n_dst) {
        const float4 o = dc[r * 2 * h4 + h4 + (i % h4)];
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    dh[i] = v;
}

// Dense gradient of the projector table: row ids[r] <- dh0[r] (+ dhd[r] for the seeds, the first n_seeds rows); ids are
// distinct.  mode 1: the rows (and the seeds' bias entries) are cleared instead — after the update.  mode 2: the rows go to
// the compact buffer `g_table` [n, h] in list order (data-parallel callers exchange them).
__global__ __launch_bounds__(kBlock) void pin_embed_grad_kernel(int64_t n, int64_t n_seeds, int h4, const int64_t* __restrict__ ids,
                                                                const float4* __restrict__ dh0, const float4* __restrict__ dhd,
                                                                float4* __restrict__ g_table, float* __restrict__ g_bias, int mode) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n * h4) return;
    const int64_t r = i / h4;
    const int c = (int)(i % h4);
    const int64_t id = ids[r];
    if (mode == 1) {
        g_table[id * h4 + c] = mi_f4_zero();
        if (c == 0 && r < n_seeds) g_bias[id] = 0.f;
        return;
    }
    float4 v = dh0[i];
    if (r < n_seeds) {
        const float4 o = dhd[i];
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    g_table[(mode == 2 ? r : id) * h4 + c] = v;
}

// One rank's list into the dense buffers: g_table[ids[r], :] += scale * rows[r, :], g_bias[ids[s]] += bias[s] (s < n_seeds).
// ids distinct within the list, lists applied by consecutive launches: never two writers of an entry.
__global__ __launch_bounds__(kBlock) void pin_apply_list_kernel(int64_t n, int64_t n_seeds, int h4, const int64_t* __restrict__ ids,
                                                                const float4* __restrict__ rows, const float* __restrict__ bias,
                                                                float scale, float4* __restrict__ g_table, float* __restrict__ g_bias) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n * h4) return;
    const int64_t r = i / h4;
    const int c = (int)(i % h4);
    const int64_t id = ids[r];
    const float4 x = rows[i];
    float4 g = g_table[id * h4 + c];
    g.x = fmaf(scale, x.x, g.x); g.y = fmaf(scale, x.y, g.y); g.z = fmaf(scale, x.z, g.z); g.w = fmaf(scale, x.w, g.w);
    g_table[id * h4 + c] = g;
    if (c == 0 && r < n_seeds) g_bias[id] += bias[r];
}

// ---- the iteration --------------------------------------------------------------------------------------------------------

enum Mode { COUNT = 0, CHECK = 1, LAUNCH = 2 };

struct PinExec {
    const mi_pinsage_model& M;
    const mi_pinsage_step_batch& B;
    MiArena ar;
    Mode mode;
    hipStream_t s;
    int rc = 0;
    bool oom = false;

    PinExec(const mi_pinsage_model& m, const mi_pinsage_step_batch& b, void* ws, size_t cap, Mode md, hipStream_t st)
        : M(m), B(b), ar(ws, cap), mode(md), s(st) {}

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
    static unsigned grid(int64_t n) { return (unsigned)std::max<int64_t>(mi_ceil_div(n, kBlock), 1); }

    void spmm(int64_t n_rows, int64_t d, const int32_t* rowptr, const int32_t* col, const float* val, const float* X, int64_t ldx,
              float* Y) {
        if (!go()) return;
        ok(mi_spmm_csr_ex_f32(n_rows, d, rowptr, col, val, X, ldx, Y, d, nullptr, d, nullptr, d, 1.0f, nullptr, nullptr, nullptr, 0,
                              (mi_stream_t)s));
    }

    void products(mi_gemm_problem* pr, int n) {
        const size_t wsb = mi_gemm_group_workspace_bytes(pr, n);
        char* w = wsb ? take_bytes(wsb) : nullptr;
        if (mode == CHECK && !mi_gemm_group_supported(pr, n)) fail(MI_ERR_UNSUPPORTED);
        if (!go()) return;
        ok(mi_gemm_group_f32(pr, n, w, wsb, (mi_stream_t)s));
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

int PinExec::run() {
    const int NL = M.n_layers;
    const int H = M.hidden, h4 = H / 4;
    if (NL < 1 || NL > MI_PINSAGE_MAX_LAYERS || B.n_blocks != NL || M.n_params < 0 || M.n_params > MI_PINSAGE_MAX_PARAMS)
        return MI_ERR_UNSUPPORTED;   // every pass: the loops below index fixed-size arrays
    {   // ... and so is the rest of the header (round 4, with the ranker's: the counting pass sizes buffers from these counts)
        if (NL < 1 || NL > MI_PINSAGE_MAX_LAYERS || B.n_blocks != NL) return MI_ERR_UNSUPPORTED;
        if (H < 4 || H % 4 != 0 || H > 128) return MI_ERR_UNSUPPORTED;
        if (!(M.p_dropout >= 0.f && M.p_dropout < 1.f)) return MI_ERR_BAD_ARG;
        if (!M.proj || !M.g_proj || !M.m_proj || !M.v_proj || !M.bias || !M.g_bias || !M.ones4) return MI_ERR_BAD_ARG;
        if (M.n_params < 0 || M.n_params > MI_PINSAGE_MAX_PARAMS) return MI_ERR_BAD_ARG;
        if (B.n_pairs > 1024) return MI_ERR_UNSUPPORTED;   // pin_score_grad_kernel's LDS list (and the sampler's own batch limit)
        if ((B.rows_out == nullptr) != (B.bias_out == nullptr) || (B.rows_out && M.apply_adam)) return MI_ERR_BAD_ARG;
        if (B.n_seeds <= 0 || B.n_pairs <= 0 || !B.seeds || !B.pos_u || !B.pos_v || !B.neg_v || !B.loss) return MI_ERR_UNSUPPORTED;
        int64_t want_dst = -1;
        for (int l = 0; l < NL; ++l) {
            const mi_pinsage_step_block& b = B.blocks[l];
            if (b.n_src <= 0 || b.n_dst <= 0 || b.n_dst > b.n_src || b.nnz < 0 || b.nnz >= (1 << 20)) return MI_ERR_UNSUPPORTED;
            if (!b.src_ids || !b.dst_rowptr || !b.src_rowptr || (b.nnz > 0 && (!b.dst_col || !b.dst_val || !b.src_col || !b.src_val)))
                return MI_ERR_BAD_ARG;
            if (want_dst >= 0 && b.n_src != want_dst) return MI_ERR_BAD_ARG;   // block l's sources = block l - 1's destinations
            want_dst = b.n_dst;
            if (M.n_ones < b.n_src) return MI_ERR_BAD_ARG;
            const mi_pinsage_conv& cv = M.conv[l];
            if (!cv.q_w || !cv.q_b || !cv.w_w || !cv.w_b || !cv.g_q_w || !cv.g_q_b || !cv.g_w_w || !cv.g_w_b) return MI_ERR_BAD_ARG;
            if (!mi_aligned16(cv.q_w) || !mi_aligned16(cv.w_w) || !mi_aligned16(cv.g_q_w) || !mi_aligned16(cv.g_w_w) ||
                !mi_aligned16(cv.q_b) || !mi_aligned16(cv.w_b) || !mi_aligned16(cv.g_q_b) || !mi_aligned16(cv.g_w_b))
                return MI_ERR_UNSUPPORTED;
        }
        if (want_dst != B.n_seeds) return MI_ERR_BAD_ARG;
        if (M.n_items < 1 || !mi_aligned16(M.proj) || !mi_aligned16(M.g_proj) || !mi_aligned16(M.m_proj) || !mi_aligned16(M.v_proj))
            return MI_ERR_UNSUPPORTED;
        for (int i = 0; i < M.n_params; ++i) {
            const mi_ranker_param& q = M.params[i];
            if (!q.p || !q.g || !q.m || !q.v || q.n < 0) return MI_ERR_BAD_ARG;
        }
    }
    const float p = M.p_dropout, scale = p > 0.f ? 1.0f / (1.0f - p) : 1.f;
    const uint32_t k0 = (uint32_t)B.seed, k1 = (uint32_t)(B.seed >> 32), st = (uint32_t)B.step;
    const int64_t n0 = B.blocks[0].n_src, ns = B.n_seeds;

    // ---- projector: the input rows; the seeds' rows are their first n_seeds ------------------------------------------------
    float* h_in[MI_PINSAGE_MAX_LAYERS + 1];
    h_in[0] = take(n0, H);
    if (go()) {
        hipLaunchKernelGGL(pin_embed_rows_kernel, dim3(grid(n0 * h4)), dim3(kBlock), 0, s, n0, h4, B.blocks[0].src_ids,
                           reinterpret_cast<const float4*>(M.proj), reinterpret_cast<float4*>(h_in[0]));
        ok(mi_launch_status());
    }
    const float* hd = h_in[0];

    // ---- SAGENet forward -------------------------------------------------------------------------------------------------------
    float *a_in[MI_PINSAGE_MAX_LAYERS], *nb[MI_PINSAGE_MAX_LAYERS], *cd[MI_PINSAGE_MAX_LAYERS], *inv[MI_PINSAGE_MAX_LAYERS];
    for (int l = 0; l < NL; ++l) {
        const mi_pinsage_step_block& b = B.blocks[l];
        const mi_pinsage_conv& cv = M.conv[l];
        a_in[l] = h_in[l];
        if (p > 0.f) {
            a_in[l] = take(b.n_src, H);
            if (go()) {
                hipLaunchKernelGGL(dropout_kernel, dim3(grid(b.n_src * h4)), dim3(kBlock), 0, s, b.n_src * h4,
                                   reinterpret_cast<const float4*>(h_in[l]), reinterpret_cast<float4*>(a_in[l]), p, scale, k0, k1,
                                   (uint32_t)(2 * l), st);
                ok(mi_launch_status());
            }
        }
        nb[l] = take(b.n_src, H);
        mi_gemm_problem q = prob(0, 1, b.n_src, H, H, a_in[l], H, cv.q_w, H, nb[l], H, nullptr, cv.q_b, 1);
        products(&q, 1);
        float* agg = take(b.n_dst, H);
        spmm(b.n_dst, H, b.dst_rowptr, b.nnz ? b.dst_col : b.dst_rowptr, b.dst_val, nb[l], H, agg);
        cd[l] = take(b.n_dst, 2 * H);
        if (go()) {
            hipLaunchKernelGGL(pin_cat_dropout_kernel, dim3(grid(b.n_dst * 2 * h4)), dim3(kBlock), 0, s, b.n_dst, h4,
                               reinterpret_cast<const float4*>(agg), reinterpret_cast<const float4*>(h_in[l]),
                               reinterpret_cast<float4*>(cd[l]), p, scale, k0, k1, (uint32_t)(2 * l + 1), st);
            ok(mi_launch_status());
        }
        float* z = take(b.n_dst, H);
        mi_gemm_problem w = prob(0, 1, b.n_dst, H, 2 * H, cd[l], 2 * H, cv.w_w, 2 * H, z, H, nullptr, cv.w_b, 1);
        products(&w, 1);
        h_in[l + 1] = take(b.n_dst, H);
        inv[l] = take(b.n_dst, 1);
        if (go()) {
            hipLaunchKernelGGL(pin_l2norm_fwd_kernel, dim3((unsigned)mi_ceil_div(b.n_dst, kBlock / MI_WAVE)), dim3(kBlock), 0, s,
                               b.n_dst, H, z, h_in[l + 1], inv[l]);
            ok(mi_launch_status());
        }
    }
    const float* hN = h_in[NL];

    // ---- scorer + hinge ----------------------------------------------------------------------------------------------------------
    float* margin = take(B.n_pairs, 1);
    float* dhf = take(ns, H);
    if (go()) {
        hipLaunchKernelGGL(pin_score_kernel, dim3((unsigned)mi_ceil_div(B.n_pairs, kBlock / MI_WAVE)), dim3(kBlock), 0, s,
                           B.n_pairs, H, hd, hN, B.seeds, B.pos_u, B.pos_v, B.neg_v, M.bias, margin);
        ok(mi_launch_status());
        hipLaunchKernelGGL(pin_score_grad_kernel, dim3((unsigned)ns), dim3(128), 0, s, ns, B.n_pairs, H, hd, hN, B.seeds, B.pos_u,
                           B.pos_v, B.neg_v, margin, dhf, M.g_bias, B.bias_out, B.loss);
        ok(mi_launch_status());
    }

    // ---- SAGENet backward --------------------------------------------------------------------------------------------------------
    const float* dh = dhf;   // gradient of h_in[l + 1]
    // bias gradients arrive as column 0 of a [H, 4] product against ones (one product of the grouped launch instead of a
    // column-sum kernel); the Adam table reads them strided and writes the dense copy into the caller's buffer
    AdamTable tb;
    memset(&tb, 0, sizeof(tb));
    auto grad_src = [&](const float* param, float* strided4, float* dense) {
        for (int i = 0; i < M.n_params; ++i)
            if (M.params[i].p == param) {
                tb.p[i].g = strided4;
                tb.g_stride[i] = 4;
                tb.g_dst[i] = dense;
            }
    };
    for (int l = NL - 1; l >= 0; --l) {
        const mi_pinsage_step_block& b = B.blocks[l];
        const mi_pinsage_conv& cv = M.conv[l];
        float* dz = take(b.n_dst, H);
        if (go()) {
            hipLaunchKernelGGL(pin_l2norm_bwd_kernel, dim3((unsigned)mi_ceil_div(b.n_dst, kBlock / MI_WAVE)), dim3(kBlock), 0, s,
                               b.n_dst, H, h_in[l + 1], inv[l], dh, dz);
            ok(mi_launch_status());
        }
        float* dcd = take(b.n_dst, 2 * H);
        float* dbw4 = take(H, 4);
        mi_gemm_problem pw[3];
        pw[0] = prob(1, 0, H, 2 * H, b.n_dst, dz, H, cd[l], 2 * H, cv.g_w_w, 2 * H);
        pw[1] = prob(1, 0, H, 4, b.n_dst, dz, H, M.ones4, 4, dbw4, 4);
        pw[2] = prob(0, 0, b.n_dst, 2 * H, H, dz, H, cv.w_w, 2 * H, dcd, 2 * H);
        grad_src(cv.w_b, dbw4, cv.g_w_b);
        products(pw, 3);
        if (p > 0.f && go()) {   // the concatenation's mask, in place
            hipLaunchKernelGGL(dropout_kernel, dim3(grid(b.n_dst * 2 * h4)), dim3(kBlock), 0, s, b.n_dst * 2 * h4,
                               reinterpret_cast<const float4*>(dcd), reinterpret_cast<float4*>(dcd), p, scale, k0, k1,
                               (uint32_t)(2 * l + 1), st);
            ok(mi_launch_status());
        }
        float* dn = take(b.n_src, H);
        spmm(b.n_src, H, b.src_rowptr, b.nnz ? b.src_col : b.src_rowptr, b.src_val, dcd, 2 * H, dn);   // columns 0..H-1 of dcd
        float* da = take(b.n_src, H);
        float* dbq4 = take(H, 4);
        mi_gemm_problem pq[3];
        pq[0] = prob(1, 0, H, H, b.n_src, dn, H, a_in[l], H, cv.g_q_w, H, nb[l]);     // relu backward: dn read as 0 where n <= 0
        pq[1] = prob(1, 0, H, 4, b.n_src, dn, H, M.ones4, 4, dbq4, 4, nb[l]);
        pq[2] = prob(0, 0, b.n_src, H, H, dn, H, cv.q_w, H, da, H, nb[l]);
        grad_src(cv.q_b, dbq4, cv.g_q_b);
        products(pq, 3);
        float* dhl = take(b.n_src, H);
        if (go()) {
            hipLaunchKernelGGL(pin_dh_merge_kernel, dim3(grid(b.n_src * h4)), dim3(kBlock), 0, s, b.n_src, b.n_dst, h4,
                               reinterpret_cast<const float4*>(da), reinterpret_cast<const float4*>(dcd),
                               reinterpret_cast<float4*>(dhl), p, scale, k0, k1, (uint32_t)(2 * l), st);
            ok(mi_launch_status());
        }
        dh = dhl;
    }
    if (oom) return MI_ERR_WORKSPACE;
    if (rc) return rc;
    if (mode != LAUNCH) return 0;

    // ---- projector gradient, Adam, and the dense gradient buffers back to zero ------------------------------------------------------
    hipLaunchKernelGGL(pin_embed_grad_kernel, dim3(grid(n0 * h4)), dim3(kBlock), 0, s, n0, ns, h4, B.blocks[0].src_ids,
                       reinterpret_cast<const float4*>(dh), reinterpret_cast<const float4*>(dhf),
                       reinterpret_cast<float4*>(B.rows_out ? B.rows_out : M.g_proj), M.g_bias, B.rows_out ? 2 : 0);
    ok(mi_launch_status());
    int64_t longest = 1;
    for (int i = 0; i < M.n_params; ++i) {
        const float* strided = tb.p[i].g;
        tb.p[i] = M.params[i];
        if (strided) tb.p[i].g = const_cast<float*>(strided); else tb.g_stride[i] = 1;
        longest = std::max(longest, M.params[i].n);
    }
    tb.n = M.n_params;
    const int64_t step = M.step > 0 ? M.step : 1;
    if (M.n_params > 0) {
        const MiAdamConsts c = mi_adam_consts(M.lr, M.beta1, M.beta2, M.eps, step);
        const unsigned gx = (unsigned)std::min<int64_t>(mi_ceil_div(longest, kBlock), 64);
        hipLaunchKernelGGL(adam_multi_kernel, dim3(gx, (unsigned)M.n_params), dim3(kBlock), 0, s, tb, c, M.apply_adam ? 1 : 0,
                           (int64_t*)nullptr, (int64_t*)nullptr, 1.f);
        ok(mi_launch_status());
    }
    if (M.apply_adam) {
        ok(mi_adam_dense_f32(M.n_items + 1, H, M.proj, H, M.g_proj, H, M.m_proj, M.v_proj, nullptr, M.lr, M.beta1, M.beta2, M.eps,
                             step, (mi_stream_t)s));
        hipLaunchKernelGGL(pin_embed_grad_kernel, dim3(grid(n0 * h4)), dim3(kBlock), 0, s, n0, ns, h4, B.blocks[0].src_ids,
                           (const float4*)nullptr, (const float4*)nullptr, reinterpret_cast<float4*>(M.g_proj), M.g_bias, 1);
        ok(mi_launch_status());
    }
    return rc;
}

}  // namespace

extern "C" int64_t mi_pinsage_step_sizeof(int32_t which) {
    switch (which) {
        case 0: return (int64_t)sizeof(mi_pinsage_model);
        case 1: return (int64_t)sizeof(mi_pinsage_step_batch);
        case 2: return (int64_t)sizeof(mi_pinsage_conv);
        case 3: return (int64_t)sizeof(mi_pinsage_step_block);
        case 4: return (int64_t)sizeof(mi_pinsage_grad_list);
        default: return -1;
    }
}

extern "C" size_t mi_pinsage_step_workspace_bytes(const mi_pinsage_model* model, const mi_pinsage_step_batch* batch) {
    if (!model || !batch) return 0;
    PinExec e(*model, *batch, reinterpret_cast<void*>((uintptr_t)4096), (size_t)1 << 46, COUNT, nullptr);
    if (e.run() != 0) return 0;   // a descriptor the executor does not take has no workspace size
    return e.ar.off + 4096;
}

extern "C" int mi_pinsage_step_check(const mi_pinsage_model* model, const mi_pinsage_step_batch* batch, void* ws, size_t ws_bytes) {
    MI_CHECK_ARG(model && batch && ws && mi_aligned16(ws));
    PinExec chk(*model, *batch, ws, ws_bytes, CHECK, nullptr);   // validates every operand; enqueues nothing, touches no memory
    return chk.run();
}

extern "C" int mi_pinsage_step_f32(const mi_pinsage_model* model, const mi_pinsage_step_batch* batch, void* ws, size_t ws_bytes,
                                   mi_stream_t stream) {
    MI_CHECK_ARG(model && batch && ws && mi_aligned16(ws));
    {
        PinExec chk(*model, *batch, ws, ws_bytes, CHECK, (hipStream_t)stream);
        const int rc = chk.run();
        if (rc) return rc;
    }
    PinExec run(*model, *batch, ws, ws_bytes, LAUNCH, (hipStream_t)stream);
    return run.run();
}

extern "C" int mi_pinsage_apply_f32(const mi_pinsage_model* model, const mi_pinsage_grad_list* lists, int32_t n_lists,
                                    float grad_scale, mi_stream_t stream) {
    MI_CHECK_ARG(model && lists && n_lists > 0 && n_lists <= 64 && grad_scale > 0.f);
    const mi_pinsage_model& M = *model;
    const int H = M.hidden, h4 = H / 4;
    MI_CHECK_ARG(H >= 4 && H % 4 == 0 && M.proj && M.g_proj && M.m_proj && M.v_proj && M.g_bias);
    MI_CHECK_ARG(M.n_params >= 0 && M.n_params <= MI_PINSAGE_MAX_PARAMS);
    for (int l = 0; l < n_lists; ++l)
        MI_CHECK_ARG(lists[l].n_rows >= 0 && lists[l].n_seeds >= 0 && lists[l].n_seeds <= lists[l].n_rows &&
                     (lists[l].n_rows == 0 || (lists[l].ids && lists[l].rows && (lists[l].n_seeds == 0 || lists[l].bias))));
    for (int i = 0; i < M.n_params; ++i) MI_CHECK_ARG(M.params[i].p && M.params[i].g && M.params[i].m && M.params[i].v);
    // nothing has been enqueued up to here
    hipStream_t s = (hipStream_t)stream;
    auto grid = [](int64_t n) { return (unsigned)std::max<int64_t>(mi_ceil_div(n, kBlock), 1); };
    for (int l = 0; l < n_lists; ++l) {
        const mi_pinsage_grad_list& g = lists[l];
        if (g.n_rows == 0) continue;
        hipLaunchKernelGGL(pin_apply_list_kernel, dim3(grid(g.n_rows * h4)), dim3(kBlock), 0, s, g.n_rows, g.n_seeds, h4, g.ids,
                           reinterpret_cast<const float4*>(g.rows), g.bias, grad_scale, reinterpret_cast<float4*>(M.g_proj), M.g_bias);
    }
    const int64_t step = M.step > 0 ? M.step : 1;
    if (M.n_params > 0) {
        AdamTable tb;
        memset(&tb, 0, sizeof(tb));
        int64_t longest = 1;
        for (int i = 0; i < M.n_params; ++i) {
            tb.p[i] = M.params[i];
            tb.g_stride[i] = 1;
            longest = std::max(longest, M.params[i].n);
        }
        tb.n = M.n_params;
        const MiAdamConsts c = mi_adam_consts(M.lr, M.beta1, M.beta2, M.eps, step);
        const unsigned gx = (unsigned)std::min<int64_t>(mi_ceil_div(longest, kBlock), 64);
        hipLaunchKernelGGL(adam_multi_kernel, dim3(gx, (unsigned)M.n_params), dim3(kBlock), 0, s, tb, c, 1, (int64_t*)nullptr,
                           (int64_t*)nullptr, grad_scale);
    }
    int rc = mi_launch_status();
    if (rc) return rc;
    rc = mi_adam_dense_f32(M.n_items + 1, H, M.proj, H, M.g_proj, H, M.m_proj, M.v_proj, nullptr, M.lr, M.beta1, M.beta2, M.eps, step,
                           stream);
    if (rc) return rc;
    for (int l = 0; l < n_lists; ++l) {
        const mi_pinsage_grad_list& g = lists[l];
        if (g.n_rows == 0) continue;
        hipLaunchKernelGGL(pin_embed_grad_kernel, dim3(grid(g.n_rows * h4)), dim3(kBlock), 0, s, g.n_rows, g.n_seeds, h4, g.ids,
                           (const float4*)nullptr, (const float4*)nullptr, reinterpret_cast<float4*>(M.g_proj), M.g_bias, 1);
    }
    return mi_launch_status();
}

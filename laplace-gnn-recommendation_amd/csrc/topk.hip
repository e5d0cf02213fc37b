// K10: batched exact top-K with per-user exclusion.
// replaces make_predictions_for_user (utils/metrics_lightgcn.py:125-142) — scores = e_u @ E_i^T,
// topk(k + |ignore|), order-preserving setdiff, [:k] — and the per-user Python loops around it
// (utils/metrics_lightgcn.py:101-106, run_pipeline_lightgcn.py:212-221).
//
// topk(k+|ignore|) followed by dropping the ignored ids and keeping k is the top-k of the
// non-ignored items in score order, so the kernel computes that directly:
//   1. scores = U[uid] @ I^T on the f32 MFMA (gemm.hip; k-ordered fma chain, bitwise = oracle)
//   2. scores[q, excl(q)] = -inf
//   3. per query row, one read of the row in the common case: a 4 096-element sample of the row fixes a
//      threshold below the k-th largest score with overwhelming probability, one pass collects the few
//      hundred entries at or above it into LDS, a bitonic sort on (score desc, id asc) orders them and the
//      first k are the answer.  When the sample misleads (fewer than k collected, or more than LDS holds:
//      massive ties, tiny item sets) the row falls back to the exact 4-pass radix select over the row
//      (ties at the threshold resolved towards smaller item ids).  Either way the result is the same.
// Integer/index work throughout after step 1: results are exact and deterministic.
#include "gemm.hpp"

namespace {

#ifndef MI_TOPK_ONE_PASS
#define MI_TOPK_ONE_PASS 1  // 0: always the multi-pass radix select (A/B, tests of the fallback)
#endif
constexpr int kBlock = 256;
constexpr int kMaxK = 1024;
constexpr int kCand = 4096;     // LDS candidate capacity of the one-pass path (>= kMaxK)
constexpr int kSample = 4096;   // sampled scores per row: 32 runs of 128 consecutive items
constexpr int kSampleRun = 128;

__device__ __forceinline__ uint32_t score_key(float x) {
    x = x + 0.0f;  // -0 -> +0 so that equal scores compare equal
    uint32_t u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // monotone: larger float -> larger key
}
__device__ __forceinline__ float key_score(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __uint_as_float(u);
}

__global__ void exclude_kernel(int64_t n_q, int64_t n_items, const int32_t* __restrict__ excl_ptr,
                               const int32_t* __restrict__ excl_idx, float* __restrict__ scores) {
    const int64_t q = blockIdx.x;
    if (q >= n_q) return;
    for (int32_t p = excl_ptr[q] + threadIdx.x; p < excl_ptr[q + 1]; p += blockDim.x) {
        const int32_t i = excl_idx[p];
        if (i >= 0 && i < n_items) scores[q * n_items + i] = -INFINITY;
    }
}

// block-wide exclusive scan of one int per thread (256 threads); returns the exclusive prefix and
// writes the block total to *total.
__device__ __forceinline__ int block_excl_scan(int v, int* sh /*[kBlock/64 + 1]*/, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int y = __shfl_up(x, off, 64);
        if (lane >= off) x += y;
    }
    if (lane == 63) sh[wave] = x;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += sh[w];
    int tot = 0;
    for (int w = 0; w < kBlock / 64; ++w) tot += sh[w];
    __syncthreads();
    *total = tot;
    return base + x - v;
}

// Descending bitonic sort of cand[0, p2) (p2 a power of two) on (key, ~id)  =>  score desc, id asc.
__device__ __forceinline__ void bitonic_desc(unsigned long long* cand, int p2) {
    const int tid = threadIdx.x;
    for (int size = 2; size <= p2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < p2 / 2; i += kBlock) {
                const int lo = 2 * i - (i & (stride - 1));
                const int hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const unsigned long long a = cand[lo], b = cand[hi];
                if ((a < b) == desc) { cand[lo] = b; cand[hi] = a; }
            }
            __syncthreads();
        }
    }
}

// One block per query row.
__global__ __launch_bounds__(kBlock) void select_kernel(int64_t n_q, int64_t n_items, int k, int kpow2,
                                                        const float* __restrict__ scores,
                                                        int64_t* __restrict__ out_idx,
                                                        float* __restrict__ out_score, int allow_fast) {
    __shared__ int hist[256];
    __shared__ int scan_sh[kBlock / 64 + 1];
    __shared__ unsigned long long cand[kCand];
    __shared__ uint32_t sh_prefix;
    __shared__ int sh_need, sh_count, sh_eq_total;
    const int64_t q = blockIdx.x;
    if (q >= n_q) return;
    const float* row = scores + q * n_items;
    const int tid = threadIdx.x;
    const int kk = (int)min((int64_t)k, n_items);

    // ---- one-pass path ----------------------------------------------------------------------------
    if (allow_fast && n_items >= 8 * kSample) {
        // (a) sample: kSample / kSampleRun runs spread evenly over the row, keys kept in LDS (aliasing cand)
        uint32_t* skey = reinterpret_cast<uint32_t*>(cand);
        const int64_t n_runs = kSample / kSampleRun, gap = n_items / n_runs;
        for (int i = tid; i < kSample; i += kBlock)
            skey[i] = score_key(row[(i / kSampleRun) * gap + (i % kSampleRun)]);
        // rank m in the sample such that, with p = kk / n_items the chance of an item to be a winner, more than m of
        // the sample being winners is a > 6-sigma event: then at least kk items of the row are >= the m-th sample key
        const float mu = (float)kSample * (float)kk / (float)n_items;
        int m = (int)(mu + 4.f * sqrtf(mu) + 8.f);
        if (tid == 0) { sh_prefix = 0u; sh_need = min(m, kSample); }
        __syncthreads();
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            const uint32_t hi_mask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
            hist[tid] = 0;
            __syncthreads();
            const uint32_t prefix = sh_prefix;
            for (int i = tid; i < kSample; i += kBlock) {
                const uint32_t key = skey[i];
                if ((key & hi_mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1);
            }
            __syncthreads();
            if (tid == 0) {
                int need = sh_need, b = 255;
                for (; b > 0; --b) {
                    if (hist[b] >= need) break;
                    need -= hist[b];
                }
                sh_prefix = prefix | ((uint32_t)b << shift);
                sh_need = need;
            }
            __syncthreads();
        }
        const uint32_t t_lo = sh_prefix;
        if (tid == 0) sh_count = 0;
        __syncthreads();  // skey is dead from here on: cand may be written
        // (b) the one pass over the row
        const bool vec = ((reinterpret_cast<uintptr_t>(row) & 15u) == 0);
        if (vec) {
            const float4* row4 = reinterpret_cast<const float4*>(row);
            const int64_t n4 = n_items / 4;
            for (int64_t i4 = tid; i4 < n4; i4 += kBlock) {
                const float4 x = row4[i4];
                const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint32_t key = score_key(xs[c]);
                    if (key >= t_lo) {
                        const int slot = atomicAdd(&sh_count, 1);
                        if (slot < kCand)
                            cand[slot] = ((unsigned long long)key << 32) |
                                         (unsigned long long)(0xFFFFFFFFu - (uint32_t)(4 * i4 + c));
                    }
                }
            }
            for (int64_t i = 4 * n4 + tid; i < n_items; i += kBlock) {
                const uint32_t key = score_key(row[i]);
                if (key >= t_lo) {
                    const int slot = atomicAdd(&sh_count, 1);
                    if (slot < kCand) cand[slot] = ((unsigned long long)key << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)i);
                }
            }
        } else {
            for (int64_t i = tid; i < n_items; i += kBlock) {
                const uint32_t key = score_key(row[i]);
                if (key >= t_lo) {
                    const int slot = atomicAdd(&sh_count, 1);
                    if (slot < kCand) cand[slot] = ((unsigned long long)key << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)i);
                }
            }
        }
        __syncthreads();
        const int cnt = sh_count;
        if (cnt >= kk && cnt <= kCand) {  // block-uniform: every winner is among the candidates
            int p2 = kpow2;
            while (p2 < cnt) p2 <<= 1;
            for (int i = cnt + tid; i < p2; i += kBlock) cand[i] = 0ull;  // pads sort last
            __syncthreads();
            bitonic_desc(cand, p2);
            for (int j = tid; j < k; j += kBlock) {
                int64_t id = -1;
                float sc = -INFINITY;
                if (j < kk) {
                    const unsigned long long c = cand[j];
                    const float s = key_score((uint32_t)(c >> 32));
                    if (s != -INFINITY) {  // excluded items never surface: pad instead
                        id = (int64_t)(0xFFFFFFFFu - (uint32_t)c);
                        sc = s;
                    }
                }
                out_idx[q * k + j] = id;
                if (out_score) out_score[q * k + j] = sc;
            }
            return;
        }
        __syncthreads();  // fall through to the exact multi-pass path
    }

    // ---- radix select: after the 4 passes sh_prefix is the key of the kk-th largest score and
    // sh_need how many elements equal to it are wanted.
    if (tid == 0) { sh_prefix = 0u; sh_need = kk; }
    __syncthreads();
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        const uint32_t hi_mask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
        hist[tid] = 0;
        __syncthreads();
        const uint32_t prefix = sh_prefix;
        for (int64_t i = tid; i < n_items; i += kBlock) {
            const uint32_t key = score_key(row[i]);
            if ((key & hi_mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1);
        }
        __syncthreads();
        if (tid == 0) {
            int need = sh_need, b = 255;
            for (; b > 0; --b) {
                if (hist[b] >= need) break;
                need -= hist[b];
            }
            sh_prefix = prefix | ((uint32_t)b << shift);
            sh_need = need;
            sh_eq_total = hist[b];  // after the last pass: how many scores equal the threshold
        }
        __syncthreads();
    }
    const uint32_t thr = sh_prefix;
    const int need_eq = sh_need;

    // ---- collect winners.  Everything above the threshold is appended in arrival order (the sort
    // below keys on (score, id), so arrival order cannot leak into the result).  Ties AT the
    // threshold: when all of them are wanted they are appended the same way; otherwise only the
    // need_eq smallest ids qualify and they are picked by an ordered block scan (rare path).
    const int eq_total = sh_eq_total;
    const bool ordered_ties = eq_total > need_eq;
    if (tid == 0) sh_count = 0;
    __syncthreads();
    for (int64_t i = tid; i < n_items; i += kBlock) {
        const uint32_t key = score_key(row[i]);
        if (key > thr || (key == thr && !ordered_ties)) {
            const int slot = atomicAdd(&sh_count, 1);
            cand[slot] = ((unsigned long long)key << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)i);
        }
    }
    __syncthreads();
    if (ordered_ties) {
        int eq_taken = 0;
        for (int64_t base = 0; base < n_items && eq_taken < need_eq; base += kBlock) {
            const int64_t i = base + tid;
            const bool eq = (i < n_items) && score_key(row[i]) == thr;
            int tot_eq;
            const int pos_eq = block_excl_scan(eq ? 1 : 0, scan_sh, &tot_eq);
            const int cnt = sh_count;
            const int eq_take = min(tot_eq, need_eq - eq_taken);
            if (eq && pos_eq < eq_take)
                cand[cnt + pos_eq] = ((unsigned long long)thr << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)i);
            __syncthreads();
            if (tid == 0) sh_count = cnt + eq_take;
            eq_taken += eq_take;
            __syncthreads();
        }
    }
    const int got = sh_count;  // == kk
    for (int i = got + tid; i < kpow2; i += kBlock) cand[i] = 0ull;  // pads sort last
    __syncthreads();

    bitonic_desc(cand, kpow2);
    for (int j = tid; j < k; j += kBlock) {
        int64_t id = -1;
        float sc = -INFINITY;
        if (j < got) {
            const unsigned long long c = cand[j];
            const float s = key_score((uint32_t)(c >> 32));
            if (s != -INFINITY) {  // excluded items never surface: pad instead
                id = (int64_t)(0xFFFFFFFFu - (uint32_t)c);
                sc = s;
            }
        }
        out_idx[q * k + j] = id;
        if (out_score) out_score[q * k + j] = sc;
    }
}

}  // namespace

extern "C" {

size_t mi_topk_workspace_bytes(int64_t n_q, int64_t n_items, int64_t k) {
    (void)k;
    if (n_q <= 0 || n_items <= 0) return 256;
    return mi_align_up((size_t)n_q * (size_t)n_items * sizeof(float), 256);
}

int mi_topk_excl_f32(int64_t n_q, int64_t n_items, int64_t d, int64_t k, const int64_t* uid,
                     const float* user_emb, int64_t ldu, const float* item_emb, int64_t ldi,
                     const int32_t* excl_ptr, const int32_t* excl_idx, int64_t* out_idx,
                     float* out_score, void* ws, size_t ws_bytes, mi_stream_t stream) {
    MI_CHECK_ARG(n_q >= 0 && n_items >= 0 && d > 0 && k > 0);
    if (n_q == 0) return 0;
    if (k > kMaxK) return MI_ERR_UNSUPPORTED;
    if (n_items >= INT32_MAX) return MI_ERR_TOO_LARGE;
    MI_CHECK_ARG(user_emb && item_emb && out_idx && ws && ldu >= d && ldi >= d);
    // excl_idx may be null when every exclusion row is empty (excl_ptr all equal)
    if (ws_bytes < mi_topk_workspace_bytes(n_q, n_items, k)) return MI_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* scores = static_cast<float*>(ws);
    MiGemmArgs g;
    g.M = n_q; g.N = n_items; g.K = d;
    g.A = user_emb; g.sa_m = ldu; g.sa_k = 1; g.a_rows = uid;
    g.B = item_emb; g.sb_n = ldi; g.sb_k = 1;
    g.bias = nullptr; g.C = scores; g.ldc = n_items; g.accumulate = 0; g.act = 0;
    int rc = mi_gemm_launch(g, nullptr, 0, s);  // M x N = queries x items fills the chip: never split
    if (rc) return rc;
    if (excl_ptr)
        hipLaunchKernelGGL(exclude_kernel, dim3((unsigned)n_q), dim3(kBlock), 0, s, n_q, n_items, excl_ptr, excl_idx, scores);
    int kpow2 = 2;
    while (kpow2 < k) kpow2 <<= 1;
    hipLaunchKernelGGL(select_kernel, dim3((unsigned)n_q), dim3(kBlock), 0, s, n_q, n_items, (int)k, kpow2, scores,
                       out_idx, out_score, MI_TOPK_ONE_PASS);
    return mi_launch_status();
}

}  // extern "C"

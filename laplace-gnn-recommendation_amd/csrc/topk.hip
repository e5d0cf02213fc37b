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

#ifndef MI_TOPK_MFMA_UNROLL
#define MI_TOPK_MFMA_UNROLL 64  // A/B on C2-sized items, k = 12 / 256: 8: 1.96 / 1.80, 16: 2.01 / 1.82, 32: 2.03 / 1.84, 64: 2.07 / 1.87
#endif                          // M users/s; s_setprio around the MFMA run: no change
#ifndef MI_TOPK_FUSED
#define MI_TOPK_FUSED 1     // 0: always materialise the score block (A/B)
#endif
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

// Shared state of one row's selection (one block per row).
struct SelectShared {
    int hist[256];
    int scan_sh[kBlock / 64 + 1];
    alignas(16) unsigned long long cand[kCand];
    uint32_t prefix;
    int need, count, eq_total;
};

__device__ __forceinline__ unsigned long long composite(uint32_t key, uint32_t item) {
    return ((unsigned long long)key << 32) | (unsigned long long)(0xFFFFFFFFu - item);  // key desc, then id asc
}

// out row <- the first kk entries of the sorted candidate list; excluded items (-inf) never surface.
__device__ __forceinline__ void write_row(const unsigned long long* cand, int got, int k, int64_t q,
                                          int64_t* __restrict__ out_idx, float* __restrict__ out_score) {
    for (int j = threadIdx.x; j < k; j += kBlock) {
        int64_t id = -1;
        float sc = -INFINITY;
        if (j < got) {
            const unsigned long long c = cand[j];
            const float s = key_score((uint32_t)(c >> 32));
            if (s != -INFINITY) {
                id = (int64_t)(0xFFFFFFFFu - (uint32_t)c);
                sc = s;
            }
        }
        out_idx[q * k + j] = id;
        if (out_score) out_score[q * k + j] = sc;
    }
}

// The bin of a 256-bin histogram (one bin per thread, kBlock = 256) that holds the need-th largest key: the highest b with
// hist[b..255] >= need.  Suffix sums by wavefront shuffles + one LDS hop across the four wavefronts; the thread that owns
// the bin publishes (prefix | b << shift, need - hist(b+1..255)).  (Round 1 let thread 0 walk the bins: up to 255 dependent
// LDS reads per pass, 4 passes — most of threshold_kernel's 97 us per 2 684-row chunk.)
__device__ __forceinline__ void radix_pick_bin(SelectShared& sh, uint32_t prefix, int shift) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = sh.hist[tid];
    int suf = h;  // inclusive suffix sum inside the wavefront: lanes lane..63
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_down(suf, d, 64);
        if (lane + d < 64) suf += o;
    }
    if (lane == 0) sh.scan_sh[wave] = suf;  // the wavefront's total
    __syncthreads();
    int above = 0;                          // bins of the wavefronts above this one
    for (int w = wave + 1; w < kBlock / 64; ++w) above += sh.scan_sh[w];
    const int incl = suf + above, excl = incl - h;  // keys in bins >= tid / > tid
    const int need = sh.need;
    __syncthreads();                        // everybody has read need / scan_sh before they are rewritten
    if (incl >= need && excl < need) {      // exactly one bin (need <= total by construction)
        sh.prefix = prefix | ((uint32_t)tid << shift);
        sh.need = need - excl;
    }
    __syncthreads();
}

// m-th largest of kSample keys held in LDS (4-pass radix select); every thread returns it.
__device__ __forceinline__ uint32_t sample_threshold(const uint32_t* skey, int m, SelectShared& sh) {
    const int tid = threadIdx.x;
    if (tid == 0) { sh.prefix = 0u; sh.need = min(m, kSample); }
    __syncthreads();
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        const uint32_t hi_mask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
        sh.hist[tid] = 0;
        __syncthreads();
        const uint32_t prefix = sh.prefix;
        for (int i = tid; i < kSample; i += kBlock) {
            const uint32_t key = skey[i];
            if ((key & hi_mask) == prefix) atomicAdd(&sh.hist[(key >> shift) & 255u], 1);
        }
        __syncthreads();
        radix_pick_bin(sh, prefix, shift);
    }
    return sh.prefix;
}

// The same value without the radix passes when m <= kBlock (it always is for the ranks sample_rank() returns on the fused
// path's sizes).  The first radix pass put nearly all 4 096 keys — scores of one row share sign and exponent — into two or
// three LDS counters: ~4 000 serialised atomics.  Instead: every thread keeps the maximum of its 16 keys; t1 = the m-th
// largest of a set of those maxima is <= the answer, because m distinct keys are >= t1; the keys >= t1 are few (a small
// multiple of m) and are ranked against each other: the m-th largest VALUE among them is the m-th largest of the whole
// sample.  `scratch`: kSample words of LDS that do not alias skey.  More than kFastList keys >= t1 (massive ties): the
// radix select.  (Round 3, per 2 684-row chunk: 42 us -> see profiles/r03_topk.md.)
constexpr int kFastList = 1024;
__device__ __forceinline__ uint32_t sample_threshold_small(const uint32_t* skey, int m, SelectShared& sh, uint32_t* scratch) {
    const int tid = threadIdx.x;
    m = min(m, kSample);
    if (m > kBlock) return sample_threshold(skey, m, sh);   // block-uniform
    uint32_t mine[kSample / kBlock];
    uint32_t mx = 0u;
#pragma unroll
    for (int j = 0; j < kSample / kBlock; ++j) {
        mine[j] = skey[tid + kBlock * j];
        mx = max(mx, mine[j]);
    }
    uint32_t* tmax = reinterpret_cast<uint32_t*>(sh.hist);   // 256 words
    tmax[tid] = mx;
    if (tid == 0) sh.count = 0;
    __syncthreads();
    // v is the m-th largest VALUE of a set iff #{> v} <= m - 1 < #{>= v} (equal values all qualify and write the same
    // word).  The loads do not depend on the counts: unrolled, the loops run at LDS issue rate.  Ranking the 256 maxima
    // against each other is 65 K compares per row and VALU-bound (19 us of the kernel); for m <= 64 every wavefront ranks
    // its own 64 maxima instead (4 x 4 K compares): each wavefront's m-th largest is a valid lower bound, the largest of
    // the four is used.
    uint32_t t1;
    if (m <= 64) {
        const int wave = tid >> 6;
        int gt = 0, ge = 0;
        const uint4* t4 = reinterpret_cast<const uint4*>(tmax + 64 * wave);
#pragma unroll 8
        for (int j = 0; j < 16; ++j) {
            const uint4 o = t4[j];
            gt += (o.x > mx) + (o.y > mx) + (o.z > mx) + (o.w > mx);
            ge += (o.x >= mx) + (o.y >= mx) + (o.z >= mx) + (o.w >= mx);
        }
        uint32_t* twave = reinterpret_cast<uint32_t*>(sh.scan_sh);
        if (gt <= m - 1 && m - 1 < ge) twave[wave] = mx;
        __syncthreads();
        t1 = max(max(twave[0], twave[1]), max(twave[2], twave[3]));
    } else {
        int gt = 0, ge = 0;
        const uint4* t4 = reinterpret_cast<const uint4*>(tmax);
#pragma unroll 8
        for (int j = 0; j < kBlock / 4; ++j) {
            const uint4 o = t4[j];
            gt += (o.x > mx) + (o.y > mx) + (o.z > mx) + (o.w > mx);
            ge += (o.x >= mx) + (o.y >= mx) + (o.z >= mx) + (o.w >= mx);
        }
        if (gt <= m - 1 && m - 1 < ge) sh.prefix = mx;
        __syncthreads();
        t1 = sh.prefix;
    }
#pragma unroll
    for (int j = 0; j < kSample / kBlock; ++j) {
        if (mine[j] >= t1) {
            const int slot = atomicAdd(&sh.count, 1);
            scratch[slot] = mine[j];            // slot < kSample always
        }
    }
    __syncthreads();
    const int c = sh.count;                     // >= m
    if (c > kFastList) {
        __syncthreads();
        return sample_threshold(skey, m, sh);
    }
    for (int i = c + tid; i < ((c + 3) & ~3); i += kBlock) scratch[i] = 0u;   // pad to whole uint4 (0 < every live key)
    __syncthreads();
    for (int i = tid; i < c; i += kBlock) {
        const uint32_t x = scratch[i];
        int gt = 0, ge = 0;
        const uint4* s4 = reinterpret_cast<const uint4*>(scratch);
#pragma unroll 4
        for (int j = 0; j < (c + 3) / 4; ++j) {
            const uint4 o = s4[j];
            gt += (o.x > x) + (o.y > x) + (o.z > x) + (o.w > x);
            ge += (o.x >= x) + (o.y >= x) + (o.z >= x) + (o.w >= x);
        }
        if (gt <= m - 1 && m - 1 < ge) sh.prefix = x;
    }
    __syncthreads();
    return sh.prefix;
}

// rank in the sample such that, with p = kk / n_items the chance of an item to be a winner, more than m of the
// sample being winners is a > 6-sigma event: then at least kk items of the row are >= the m-th sample key.
// (A/B round 2: mu + 3.5 sqrt(mu) + 3 halves the candidates per row (270 -> 122 at k = 12) and does not change the
// fused kernel's time at all, while the rows that then end short of k candidates — exclusions eat into the margin — and
// are recomputed exactly take the whole call from 2.7 to 1.5 M users/s.  The margin stays.)
__device__ __forceinline__ int sample_rank(int kk, int64_t n_items) {
    const float mu = (float)kSample * (float)kk / (float)n_items;
    return (int)(mu + 4.f * sqrtf(mu) + 8.f);
}

// Exact multi-pass selection over a materialised score row: 4-pass radix select of the kk-th largest key,
// collection of the winners (ties at the threshold towards smaller ids), sort, write.
__device__ void select_row_exact(const float* __restrict__ row, int64_t n_items, int k, int kk, int kpow2, int64_t q,
                                 int64_t* __restrict__ out_idx, float* __restrict__ out_score, SelectShared& sh) {
    const int tid = threadIdx.x;
    if (tid == 0) { sh.prefix = 0u; sh.need = kk; }
    __syncthreads();
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        const uint32_t hi_mask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
        sh.hist[tid] = 0;
        __syncthreads();
        const uint32_t prefix = sh.prefix;
        for (int64_t i = tid; i < n_items; i += kBlock) {
            const uint32_t key = score_key(row[i]);
            if ((key & hi_mask) == prefix) atomicAdd(&sh.hist[(key >> shift) & 255u], 1);
        }
        __syncthreads();
        if (tid == 0) {
            int need = sh.need, b = 255;
            for (; b > 0; --b) {
                if (sh.hist[b] >= need) break;
                need -= sh.hist[b];
            }
            sh.prefix = prefix | ((uint32_t)b << shift);
            sh.need = need;
            sh.eq_total = sh.hist[b];  // after the last pass: how many scores equal the threshold
        }
        __syncthreads();
    }
    const uint32_t thr = sh.prefix;
    const int need_eq = sh.need;
    // Everything above the threshold is appended in arrival order (the sort below keys on (score, id), so arrival
    // order cannot leak into the result).  Ties AT the threshold: when all of them are wanted they are appended the
    // same way; otherwise only the need_eq smallest ids qualify and they are picked by an ordered block scan.
    const int eq_total = sh.eq_total;
    const bool ordered_ties = eq_total > need_eq;
    if (tid == 0) sh.count = 0;
    __syncthreads();
    for (int64_t i = tid; i < n_items; i += kBlock) {
        const uint32_t key = score_key(row[i]);
        if (key > thr || (key == thr && !ordered_ties)) {
            const int slot = atomicAdd(&sh.count, 1);
            sh.cand[slot] = composite(key, (uint32_t)i);
        }
    }
    __syncthreads();
    if (ordered_ties) {
        int eq_taken = 0;
        for (int64_t base = 0; base < n_items && eq_taken < need_eq; base += kBlock) {
            const int64_t i = base + tid;
            const bool eq = (i < n_items) && score_key(row[i]) == thr;
            int tot_eq;
            const int pos_eq = block_excl_scan(eq ? 1 : 0, sh.scan_sh, &tot_eq);
            const int cnt = sh.count;
            const int eq_take = min(tot_eq, need_eq - eq_taken);
            if (eq && pos_eq < eq_take) sh.cand[cnt + pos_eq] = composite(thr, (uint32_t)i);
            __syncthreads();
            if (tid == 0) sh.count = cnt + eq_take;
            eq_taken += eq_take;
            __syncthreads();
        }
    }
    const int got = sh.count;  // == kk
    for (int i = got + tid; i < kpow2; i += kBlock) sh.cand[i] = 0ull;  // pads sort last
    __syncthreads();
    bitonic_desc(sh.cand, kpow2);
    write_row(sh.cand, got, k, q, out_idx, out_score);
}

// Sort cnt candidates (cnt >= kk) already in sh.cand and write the row.
#ifndef MI_TOPK_RANK_FINISH
#define MI_TOPK_RANK_FINISH 1   // 0: always the bitonic sort (A/B)
#endif
constexpr int kRankFinish = MI_TOPK_RANK_FINISH;
__device__ __forceinline__ void finish_candidates(int cnt, int k, int kk, int kpow2, int64_t q,
                                                  int64_t* __restrict__ out_idx, float* __restrict__ out_score,
                                                  SelectShared& sh) {
    if (kRankFinish && kk <= 64 && cnt <= kCand / 2) {
        // Small k (evaluation's k = 12 with ~270 candidates): no sort.  (1) S of the candidates' keys (S = 64 / 128 / 256 >=
        // 4 kk, or all of them) are ranked against each other — the kk-th largest of a subset is a lower bound of the
        // kk-th largest of all; (2) the candidates that reach it (about kk * cnt / S) are compacted; (3) each of those
        // counts the ones that beat it — composites are unique, the item id is part of them — and the ones of rank < kk
        // write themselves to their place.  Work ~S^2 + n2^2 compares instead of the 45-55 barrier stages of the padded
        // bitonic sort (or cnt^2 for ranking everything: VALU-bound at ~25 us per chunk).
        const int tid = threadIdx.x;
        const int S = min(cnt, kk <= 16 ? 64 : (kk <= 32 ? 128 : 256));
        uint32_t* keys = reinterpret_cast<uint32_t*>(sh.hist);           // 256 words
        unsigned long long* list2 = sh.cand + kCand / 2;
        const uint32_t x = tid < S ? (uint32_t)(sh.cand[tid] >> 32) : 0u;
        keys[tid] = x;                                                    // zero beyond S: pads of the uint4 reads
        if (tid == 0) sh.count = 0;
        __syncthreads();
        if (tid < S) {
            int gt = 0, ge = 0;
            const uint4* k4 = reinterpret_cast<const uint4*>(keys);
#pragma unroll 8
            for (int j = 0; j < (S + 3) / 4; ++j) {
                const uint4 o = k4[j];
                gt += (o.x > x) + (o.y > x) + (o.z > x) + (o.w > x);
                ge += (o.x >= x) + (o.y >= x) + (o.z >= x) + (o.w >= x);
            }
            if (gt <= kk - 1 && kk - 1 < ge) sh.prefix = x;               // the kk-th largest value of the subset
        }
        __syncthreads();
        const uint32_t t2 = sh.prefix;
        for (int i = tid; i < cnt; i += kBlock) {
            const unsigned long long c = sh.cand[i];
            if ((uint32_t)(c >> 32) >= t2) {
                const int slot = atomicAdd(&sh.count, 1);
                if (slot < kBlock) list2[slot] = c;
            }
        }
        __syncthreads();
        const int n2 = sh.count;                                          // >= kk
        if (n2 <= kBlock) {                                               // block-uniform
            if (tid == 0 && (n2 & 1)) list2[n2] = 0ull;                   // the pair reads see a pad that beats nobody
            __syncthreads();
            if (tid < n2) {
                const unsigned long long me = list2[tid];
                int rank = 0;
                const ulonglong2* c2 = reinterpret_cast<const ulonglong2*>(list2);
#pragma unroll 8
                for (int j = 0; j < (n2 + 1) / 2; ++j) {
                    const ulonglong2 o = c2[j];
                    rank += (o.x > me ? 1 : 0) + (o.y > me ? 1 : 0);
                }
                if (rank < kk) {
                    const float sc = key_score((uint32_t)(me >> 32));
                    const bool live = sc != -INFINITY;
                    out_idx[q * k + rank] = live ? (int64_t)(0xFFFFFFFFu - (uint32_t)me) : -1;
                    if (out_score) out_score[q * k + rank] = live ? sc : -INFINITY;
                }
            }
            for (int j = kk + tid; j < k; j += kBlock) {                  // k > n_items: the tail of the row
                out_idx[q * k + j] = -1;
                if (out_score) out_score[q * k + j] = -INFINITY;
            }
            return;
        }
        __syncthreads();   // massive ties at the bound: the sort below
    }
    if (kRankFinish && kk > 64 && cnt <= 2 * kBlock) {
        // Large k with a SHORT list (round 4: the prefilter path's survivors at k = 256 are kk + a few, ~270-350): no radix
        // select and no sort at all — every candidate counts the candidates that beat it (composites are unique: the item id
        // is part of them) and those of rank < kk write themselves to their place.  Two candidates per thread, the list read
        // as pairs from LDS: cnt^2 / 256 ~ 500 64-bit compares per thread against the 12 barrier stages of the 4-pass radix
        // select plus the 36 of the 256-entry bitonic sort.
        const int tid = threadIdx.x;
        if (tid == 0 && (cnt & 1)) sh.cand[cnt] = 0ull;                   // the pair reads see a pad that beats nobody
        __syncthreads();
        const int i0 = tid, i1 = tid + kBlock;
        const unsigned long long m0 = i0 < cnt ? sh.cand[i0] : ~0ull, m1 = i1 < cnt ? sh.cand[i1] : ~0ull;
        int r0 = 0, r1 = 0;
        const ulonglong2* c2 = reinterpret_cast<const ulonglong2*>(sh.cand);
#pragma unroll 4
        for (int j = 0; j < (cnt + 1) / 2; ++j) {
            const ulonglong2 o = c2[j];
            r0 += (o.x > m0 ? 1 : 0) + (o.y > m0 ? 1 : 0);
            r1 += (o.x > m1 ? 1 : 0) + (o.y > m1 ? 1 : 0);
        }
#define MI_RANK_WRITE(me, rank, idx)                                                        \
        if ((idx) < cnt && (rank) < kk) {                                                   \
            const float sc = key_score((uint32_t)((me) >> 32));                             \
            const bool live = sc != -INFINITY;                                              \
            out_idx[q * k + (rank)] = live ? (int64_t)(0xFFFFFFFFu - (uint32_t)(me)) : -1;  \
            if (out_score) out_score[q * k + (rank)] = live ? sc : -INFINITY;               \
        }
        MI_RANK_WRITE(m0, r0, i0)
        MI_RANK_WRITE(m1, r1, i1)
#undef MI_RANK_WRITE
        for (int j = kk + tid; j < k; j += kBlock) {                      // k > n_items: the tail of the row
            out_idx[q * k + j] = -1;
            if (out_score) out_score[q * k + j] = -INFINITY;
        }
        return;
    }
    if (kRankFinish && kk > 64 && cnt > kpow2 && cnt <= kCand / 2) {
        // Large k (the matcher dump's k = 256 with ~750 candidates): instead of sorting the list padded to 1 024 (55 barrier
        // stages over 512 pairs), find the kk-th largest score key by a 4-pass radix select over the candidates in LDS, keep
        // the candidates that reach it — exactly kk unless scores tie at the bound — and sort those (kpow2 entries: 36 stages
        // over 128 pairs at k = 256).  Ties at the bound that do not fit: the full sort below, as before.
        const int tid = threadIdx.x;
        if (tid == 0) { sh.prefix = 0u; sh.need = kk; }
        __syncthreads();
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            const uint32_t hi_mask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
            sh.hist[tid] = 0;
            __syncthreads();
            const uint32_t prefix = sh.prefix;
            for (int i = tid; i < cnt; i += kBlock) {
                const uint32_t key = (uint32_t)(sh.cand[i] >> 32);
                if ((key & hi_mask) == prefix) atomicAdd(&sh.hist[(key >> shift) & 255u], 1);
            }
            __syncthreads();
            radix_pick_bin(sh, prefix, shift);
        }
        const uint32_t thr = sh.prefix;          // the kk-th largest key
        if (tid == 0) { sh.count = 0; sh.eq_total = 0; }
        __syncthreads();
        unsigned long long* list2 = sh.cand + kCand / 2;
        for (int i = tid; i < cnt; i += kBlock) {
            const unsigned long long c = sh.cand[i];
            const uint32_t key = (uint32_t)(c >> 32);
            if (key >= thr) {
                const int slot = atomicAdd(&sh.count, 1);
                if (slot < kpow2) list2[slot] = c;
            }
        }
        __syncthreads();
        const int n2 = sh.count;                  // >= kk; == kk unless several candidates carry the bound's score
        if (n2 <= kpow2) {                        // block-uniform
            for (int i = n2 + tid; i < kpow2; i += kBlock) list2[i] = 0ull;   // pads sort last
            __syncthreads();
            bitonic_desc(list2, kpow2);
            write_row(list2, kk, k, q, out_idx, out_score);
            return;
        }
        __syncthreads();
    }
    int p2 = kpow2;
    while (p2 < cnt) p2 <<= 1;
    for (int i = cnt + threadIdx.x; i < p2; i += kBlock) sh.cand[i] = 0ull;
    __syncthreads();
    bitonic_desc(sh.cand, p2);
    write_row(sh.cand, kk, k, q, out_idx, out_score);
}

// One block per query row of a materialised score block.
__global__ __launch_bounds__(kBlock) void select_kernel(int64_t n_q, int64_t n_items, int k, int kpow2,
                                                        const float* __restrict__ scores,
                                                        int64_t* __restrict__ out_idx,
                                                        float* __restrict__ out_score, int allow_fast) {
    __shared__ SelectShared sh;
    const int64_t q = blockIdx.x;
    if (q >= n_q) return;
    const float* row = scores + q * n_items;
    const int tid = threadIdx.x;
    const int kk = (int)min((int64_t)k, n_items);

    // ---- one-pass path ----------------------------------------------------------------------------
    if (allow_fast && n_items >= 8 * kSample) {
        // (a) sample: kSample / kSampleRun runs spread evenly over the row, keys kept in LDS (aliasing cand)
        uint32_t* skey = reinterpret_cast<uint32_t*>(sh.cand);
        const int64_t n_runs = kSample / kSampleRun, gap = n_items / n_runs;
        for (int i = tid; i < kSample; i += kBlock)
            skey[i] = score_key(row[(i / kSampleRun) * gap + (i % kSampleRun)]);
        __syncthreads();
        const uint32_t t_lo = sample_threshold(skey, sample_rank(kk, n_items), sh);
        if (tid == 0) sh.count = 0;
        __syncthreads();  // skey is dead from here on: cand may be written
        // (b) the one pass over the row
        const bool vec = ((reinterpret_cast<uintptr_t>(row) & 15u) == 0);
        const int64_t n4 = vec ? n_items / 4 : 0;
        const float4* row4 = reinterpret_cast<const float4*>(row);
        for (int64_t i4 = tid; i4 < n4; i4 += kBlock) {
            const float4 x = row4[i4];
            const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t key = score_key(xs[c]);
                if (key >= t_lo) {
                    const int slot = atomicAdd(&sh.count, 1);
                    if (slot < kCand) sh.cand[slot] = composite(key, (uint32_t)(4 * i4 + c));
                }
            }
        }
        for (int64_t i = 4 * n4 + tid; i < n_items; i += kBlock) {
            const uint32_t key = score_key(row[i]);
            if (key >= t_lo) {
                const int slot = atomicAdd(&sh.count, 1);
                if (slot < kCand) sh.cand[slot] = composite(key, (uint32_t)i);
            }
        }
        __syncthreads();
        const int cnt = sh.count;
        if (cnt >= kk && cnt <= kCand) {  // block-uniform: every winner is among the candidates
            finish_candidates(cnt, k, kk, kpow2, q, out_idx, out_score, sh);
            return;
        }
        __syncthreads();  // fall through to the exact multi-pass path
    }
    select_row_exact(row, n_items, k, kk, kpow2, q, out_idx, out_score, sh);
}

// ---- fused path: the score block is never written ---------------------------------------------------------
// For n_items >= 8 * kSample and d <= 128 (multiple of 4, float4-addressable operands):
//   1. scores of every query against the kSample sampled items (plain GEMM, 4 % of the work) -> per-row threshold
//      (exclusions knocked out of the sample first, so that a user's own purchases cannot lift it);
//   2. topk_scores_filter_kernel: each workgroup keeps a 64-query panel of U in LDS, streams 64-item panels of
//      I past it (next panel's global loads in flight under the current panel's 64 MFMAs) and, instead of storing
//      the 64x64 scores, appends the few that reach their row's threshold and are not excluded (bitmap test) to
//      that row's candidate list in global memory;
//   3. topk_finalize_kernel: sorts a row's candidates and writes the answer; a row whose list came out short or
//      overflowed (massive ties, nearly everything excluded) recomputes its scores with a scalar fma chain —
//      the same bits as the MFMA — and takes the exact multi-pass selection.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int FM = 64, FN = 64, FKC = 128, FKPAD = 129;   // query / item panel rows, panel width, padded LDS row
constexpr int kCap = kCand;                                // candidate slots per query in global memory

struct FusedArgs {
    int64_t n_q, n_items;
    int d;
    const int64_t* uid;
    const float* U; int64_t ldu;
    const float* I; int64_t ldi;
    const uint32_t* thr;        // [n_q] threshold keys
    const uint32_t* bitmap;     // [n_q, words] exclusion bits
    int64_t words;
    unsigned long long* cand;   // [n_q, kCap]
    int* cnt;                   // [n_q]
    int64_t tiles_per_slice;    // item panels per workgroup along blockIdx.x
};

__global__ void gather_sample_rows_kernel(int64_t n_items, int d, const float* __restrict__ I, int64_t ldi,
                                          float* __restrict__ Is) {
    const int p = blockIdx.x;  // sample position
    const int64_t gap = n_items / (kSample / kSampleRun);
    const int64_t item = (int64_t)(p / kSampleRun) * gap + (p % kSampleRun);
    for (int c = threadIdx.x; c < d; c += blockDim.x) Is[(int64_t)p * d + c] = I[item * ldi + c];
}

// One block per query: the row's exclusion list -> bits of its bitmap row and -inf over the sampled scores it covers (in
// LDS: the global sample block is read once and never patched), the sampled threshold, and the row's candidate counter
// reset.  (Round 3: was exclude_bitmap_kernel + threshold_kernel + a memset of the counters.)
__global__ __launch_bounds__(kBlock) void threshold_kernel(int64_t n_q, int64_t n_items, int k,
                                                           const float* __restrict__ sample_scores,
                                                           const int32_t* __restrict__ excl_ptr,
                                                           const int32_t* __restrict__ excl_idx,
                                                           uint32_t* __restrict__ bitmap, int64_t words,
                                                           uint32_t* __restrict__ thr, int* __restrict__ cnt,
                                                           const float* __restrict__ epsv, float* __restrict__ thrf) {
    __shared__ SelectShared sh;
    const int64_t q = blockIdx.x;
    if (q >= n_q) return;
    uint32_t* skey = reinterpret_cast<uint32_t*>(sh.cand);
    const float4* src = reinterpret_cast<const float4*>(sample_scores + q * kSample);
    for (int i = threadIdx.x; i < kSample / 4; i += kBlock) {
        const float4 v = src[i];
        skey[4 * i] = score_key(v.x); skey[4 * i + 1] = score_key(v.y); skey[4 * i + 2] = score_key(v.z); skey[4 * i + 3] = score_key(v.w);
    }
    if (threadIdx.x == 0) cnt[q] = 0;
    __syncthreads();
    if (excl_ptr) {
        const int64_t gap = n_items / (kSample / kSampleRun);
        const uint32_t gone = score_key(-INFINITY);
        for (int32_t p = excl_ptr[q] + threadIdx.x; p < excl_ptr[q + 1]; p += kBlock) {
            const int32_t i = excl_idx[p];
            if (i < 0 || i >= n_items) continue;
            atomicOr(&bitmap[q * words + (i >> 5)], 1u << (i & 31));
            const int64_t run = i / gap, off = i - run * gap;
            if (run < kSample / kSampleRun && off < kSampleRun) skey[run * kSampleRun + off] = gone;
        }
        __syncthreads();
    }
#if defined(MI_TOPK_PROBE) && MI_TOPK_PROBE == 1   // timing probe only: loads + exclusions, no selection
    const uint32_t t_lo = skey[17] | 0xFF000000u;
#else
    const uint32_t t_lo = sample_threshold_small(skey, sample_rank((int)min((int64_t)k, n_items), n_items), sh, skey + kSample);
#endif
    if (threadIdx.x == 0) {
        thr[q] = t_lo;
        // prefilter (topk_prefilter.hpp): the sample was scored in bf16x3 as well, so t_lo - eps bounds the exact sample score
        // it stands for, and the list is kept complete down to 3 eps below that.  NaN / -inf: everything passes, the row
        // overflows and takes the exact path
        if (thrf) thrf[q] = key_score(t_lo) - 4.f * epsv[q];
    }
}

// 64 x FKC panel rows [row0, row0 + 64) of a K-contiguous operand into registers / LDS (zero beyond n_rows and d).
__device__ __forceinline__ void fpanel_issue(float4 (&v)[8], const float* __restrict__ base, int64_t ld,
                                             const int64_t* __restrict__ row_map, int64_t row0, int64_t n_rows, int d,
                                             int tid) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int idx = tid + 256 * j;
        const int r = idx / (FKC / 4), c4 = idx % (FKC / 4);
        const int64_t gr = row0 + r;
        v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gr < n_rows && c4 * 4 < d) {
            const int64_t row = row_map ? row_map[gr] : gr;
            v[j] = *reinterpret_cast<const float4*>(base + row * ld + c4 * 4);
        }
    }
}
// (A/B, round 1: one float per lane and load -> conflict-free LDS writes but 32 address computations spill;
// rotating the float4 component by lane / 8 -> 32 distinct banks but the selects cost more than the conflicts:
// 2.01 -> 1.71 M users/s; 132-float rows with ds_write_b128 / ds_read_b128 (conflict-free both ways, one float4
// per two MFMAs) plus a float compare against the threshold: 950 -> 1 010 us per chunk.  Stage timings of this
// kernel per 2 621-query chunk: MFMA loop + barriers alone 586 us (114 TF/s), + panel prefetch / commit 721,
// + threshold epilogue and staging 950.  A 128-item panel with two accumulators per wavefront (one workgroup per CU):
// 2.06 -> 1.74 M users/s.  The plain four-word write below stays.)
__device__ __forceinline__ void fpanel_commit(float (*panel)[FKPAD], const float4 (&v)[8], int tid) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int idx = tid + 256 * j;
        const int r = idx / (FKC / 4), c = (idx % (FKC / 4)) * 4;
        panel[r][c] = v[j].x; panel[r][c + 1] = v[j].y; panel[r][c + 2] = v[j].z; panel[r][c + 3] = v[j].w;
    }
}

constexpr int kStage = 1024;  // per-workgroup staging slots in LDS

// Staged candidates -> their rows' lists in global memory (exclusion bitmap tested here, off the MFMA path).
__device__ __forceinline__ void flush_stage(const FusedArgs& a, int64_t m0, const unsigned long long* st_val,
                                            const unsigned char* st_row, int n) {
    for (int e = threadIdx.x; e < n; e += 256) {
        const unsigned long long c = st_val[e];
        const int64_t q = m0 + st_row[e];
        const uint32_t item = 0xFFFFFFFFu - (uint32_t)c;
        if ((a.bitmap[q * a.words + (item >> 5)] >> (item & 31)) & 1u) continue;
        const int slot = atomicAdd(&a.cnt[q], 1);
        if (slot < kCap) a.cand[q * kCap + slot] = c;
    }
}

__global__ __launch_bounds__(256, 2) void topk_scores_filter_kernel(FusedArgs a) {
    __shared__ float As[FM][FKPAD];
    __shared__ float Bs[FN][FKPAD];
    __shared__ unsigned long long st_val[kStage];
    __shared__ unsigned char st_row[kStage];
    __shared__ int st_cnt;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int64_t m0 = (int64_t)blockIdx.y * FM;
    const int64_t n_tiles = (a.n_items + FN - 1) / FN;
    const int64_t t0 = (int64_t)blockIdx.x * a.tiles_per_slice;
    const int64_t t1 = min(n_tiles, t0 + a.tiles_per_slice);
    if (t0 >= t1) return;
    float4 va[8], vb[8];
    fpanel_issue(va, a.U, a.ldu, a.uid, m0, a.n_q, a.d, tid);
    fpanel_issue(vb, a.I, a.ldi, nullptr, t0 * FN, a.n_items, a.d, tid);
    fpanel_commit(As, va, tid);
    fpanel_commit(Bs, vb, tid);
    if (tid == 0) st_cnt = 0;
    // this lane's 16 accumulator rows are fixed for the whole launch: their thresholds live in registers
    // (rows beyond n_q get a threshold no key reaches... except NaN patterns, hence the explicit row test)
    uint32_t thr[16];
    const int row_base = wm * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int64_t gm = m0 + row_base + (reg & 3) + 8 * (reg >> 2);
        thr[reg] = gm < a.n_q ? a.thr[gm] : 0xFFFFFFFFu;
    }
    const int rows_here = (int)min((int64_t)FM, a.n_q - m0);
    __syncthreads();
    const float* ap = &As[wm * 32 + (lane & 31)][lane >> 5];
    const float* bp = &Bs[wn * 32 + (lane & 31)][lane >> 5];
    const int n_mfma = (a.d + 1) / 2;  // the panels are zero beyond d
    for (int64_t t = t0; t < t1; ++t) {
        const bool more = t + 1 < t1;
        if (more) fpanel_issue(vb, a.I, a.ldi, nullptr, (t + 1) * FN, a.n_items, a.d, tid);  // in flight under the MFMAs
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        if (n_mfma == FKC / 2) {
#pragma unroll MI_TOPK_MFMA_UNROLL
            for (int s = 0; s < FKC / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * s], bp[2 * s], acc, 0, 0, 0);
        } else {
            for (int s = 0; s < n_mfma; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * s], bp[2 * s], acc, 0, 0, 0);
        }
        const int64_t gn = t * FN + wn * 32 + (lane & 31);
        if (gn < a.n_items) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const uint32_t key = score_key(acc[reg]);
                const int rl = row_base + (reg & 3) + 8 * (reg >> 2);
                if (key >= thr[reg] && rl < rows_here) {  // rare: LDS only on the hot path
                    const unsigned long long c = composite(key, (uint32_t)gn);
                    const int slot = atomicAdd(&st_cnt, 1);
                    if (slot < kStage) {
                        st_val[slot] = c;
                        st_row[slot] = (unsigned char)rl;
                    } else if (!((a.bitmap[(m0 + rl) * a.words + (gn >> 5)] >> (gn & 31)) & 1u)) {  // staging full
                        const int gs = atomicAdd(&a.cnt[m0 + rl], 1);
                        if (gs < kCap) a.cand[(m0 + rl) * kCap + gs] = c;
                    }
                }
            }
        }
        __syncthreads();  // every wavefront is done reading Bs and appending
        if (more) fpanel_commit(Bs, vb, tid);
        const int staged = min(st_cnt, kStage);
        if (staged >= kStage / 2 || !more) {  // block-uniform
            flush_stage(a, m0, st_val, st_row, staged);
            __syncthreads();
            if (tid == 0) st_cnt = 0;
        }
        __syncthreads();
    }
}

// Round 2: the fused kernel rebuilt in three steps (each measured with tools/topk_stage.sh; what was removed is noted
// where it taught something).
// Step 1 — the query panel's MFMA fragments live in REGISTERS (64 VGPRs per lane for the whole launch: one LDS read per
// MFMA instead of two) and the LDS it occupied becomes a second item buffer: one barrier per panel instead of two.
// 0.41 -> 0.49 of the f32 MFMA peak.  Also here: the grid rule (rounds x panels, see the launcher) — the old rule made
// 1 025 workgroups for a 2 621-query chunk, one more than two rounds of the chip.
#ifndef MI_TOPK_STAGE
#define MI_TOPK_STAGE 0   // stage timing only (wrong results): 1 = no threshold epilogue, 2 = also no item-panel DMA, 3 = also no barrier,
#endif                    // 5 = also no DMA wait + device printf of cycles, 6 = votes but no staging, 8 = everything, thresholds at +inf

// Step 2 — the work on panel t-1's scores is issued INSIDE the MFMA chain of panel t.  An MFMA of this shape holds
// the matrix pipe for 64 cycles and each one waits for the one before it (one accumulator), so whatever the wavefront
// issues in between is free: with two accumulators that take turns, the 16 votes of the previous panel, the reservation
// of staging slots for all its hits (scalar arithmetic: each wavefront owns a quarter of the staging area) and the hits
// of one accumulator register per MFMA slot (a few per panel and wavefront pass: this path runs on most panels, not
// rarely) all go into that shadow instead of after the chain has drained.  0.49 -> 0.53 of peak.
struct PipeState {
    unsigned long long hit[16];
    unsigned long long col_ok_prev;
    int64_t gn_prev;
    int base, total;
    int mine;   // entries in this wavefront's own staging region (wave-uniform; may run past kStageW: overflow)
};
constexpr int kStageW = kStage / 4;  // staging slots per wavefront

// one MFMA of the chain with its B operand.  PIN_READS: the operands of MFMAs 8g..8g+7 are read while group g-1 runs,
// and scheduling barriers keep them there — left alone the compiler sinks every read to just in front of its MFMA
// (one in flight) whatever registers it has
template <int NM, bool PIN_READS>
__device__ __forceinline__ void pipe_mfma(int s, f32x16& acc, const float (&areg)[NM], const float* bp, float (&bq)[2][8]) {
    if (PIN_READS && s % 8 == 0) {
        if (s + 8 < NM) {
#pragma unroll
            for (int u = 0; u < 8; ++u) bq[((s >> 3) + 1) & 1][u] = bp[2 * (s + 8 + u)];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[s], PIN_READS ? bq[(s >> 3) & 1][s & 7] : bp[2 * s], acc, 0, 0, 0);
}

// MFMAs 8q..NM-1 of a panel; STAGING: with the previous panel's hits woven in (one accumulator register per MFMA slot)
template <int NM, bool PIN_READS, bool STAGING>
__device__ __forceinline__ void pipe_tail(f32x16& acc, const f32x16& prev, const float (&areg)[NM], const float* bp,
                                          float (&bq)[2][8], PipeState& st, int lane, int row_base,
                                          unsigned long long* st_val, unsigned char* st_row, unsigned char* st_over) {
    constexpr int q = NM / 32;
#pragma unroll
    for (int s = 8 * q; s < NM; ++s) {
        pipe_mfma<NM, PIN_READS>(s, acc, areg, bp, bq);
        if (STAGING && s == 8 * q) {
            st.base = st.mine;   // the region is this wavefront's own: reserving slots is scalar arithmetic
            st.mine += st.total;
        } else if (STAGING && s <= 8 * q + 16 * q && (s - 8 * q - 1) % q == 0) {
            const int reg = (s - 8 * q - 1) / q;
            if (st.hit[reg] != 0ull) {  // wave-uniform
                if ((st.hit[reg] >> lane) & 1ull) {
                    const int slot = st.base + __popcll(st.hit[reg] & ((1ull << lane) - 1ull));
                    const int rl = row_base + (reg & 3) + 8 * (reg >> 2);
                    const unsigned long long c = composite(score_key(prev[reg]), (uint32_t)st.gn_prev);
                    if (slot < kStageW) {
                        st_val[slot] = c;
                        st_row[slot] = (unsigned char)rl;
                    } else {
                        st_over[rl] = 1;  // region full: the row takes the exact path (see wave_flush)
                    }
                }
                st.base += __popcll(st.hit[reg]);
            }
        }
    }
}

template <int NM, bool PIN_READS>
__device__ __forceinline__ void pipe_panel(const FusedArgs& a, f32x16& acc, const f32x16& prev, const float (&areg)[NM],
                                           const float* bp, const float (&thr_f)[16], PipeState& st, int lane, int row_base, unsigned long long* st_val, unsigned char* st_row,
                                           unsigned char* st_over) {
    constexpr int q = NM / 32;  // MFMAs per vote pair
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    st.total = 0;
    st.base = 0;
    float bq[2][8];
    if (PIN_READS) {
#pragma unroll
        for (int u = 0; u < 8; ++u) bq[0][u] = bp[2 * u];
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int s = 0; s < 8 * q; ++s) {  // votes: one register per MFMA (d = 128) or two (d = 64)
        pipe_mfma<NM, PIN_READS>(s, acc, areg, bp, bq);
        if (MI_TOPK_STAGE == 0 || MI_TOPK_STAGE == 6 || MI_TOPK_STAGE == 8) {
#pragma unroll
            for (int reg = (2 * s) / q; reg < (2 * s + 2) / q; ++reg) {
                st.hit[reg] = __ballot(!(prev[reg] < thr_f[reg])) & st.col_ok_prev;
                st.total += __popcll(st.hit[reg]);
            }
        }
    }
    // Nothing passed (about half of a wavefront's panels at k = 12): the rest of the chain as one straight run;
    // otherwise the copy with the staging steps and their 16 wave-uniform branches (A/B: equal speed to one copy)
    if (MI_TOPK_STAGE == 6 && st.total == 12345) st_over[0] = 1;
    if ((MI_TOPK_STAGE == 0 || MI_TOPK_STAGE == 8) && st.total != 0)
        pipe_tail<NM, PIN_READS, true>(acc, prev, areg, bp, bq, st, lane, row_base, st_val, st_row, st_over);
    else
        pipe_tail<NM, PIN_READS, false>(acc, prev, areg, bp, bq, st, lane, row_base, st_val, st_row, st_over);
}


// the scores of `acc` that passed (hit[reg] = lanes) -> this wavefront's staging region (the last panel's, after the loop)
__device__ __forceinline__ void stage_hits(const f32x16& acc, const unsigned long long (&hit)[16], int total, int64_t gn,
                                           int row_base, int lane, PipeState& st, unsigned long long* st_val,
                                           unsigned char* st_row, unsigned char* st_over) {
    int base = st.mine;
    st.mine += total;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        if (hit[reg] == 0ull) continue;
        if ((hit[reg] >> lane) & 1ull) {
            const int slot = base + __popcll(hit[reg] & ((1ull << lane) - 1ull));
            const int rl = row_base + (reg & 3) + 8 * (reg >> 2);
            const unsigned long long c = composite(score_key(acc[reg]), (uint32_t)gn);
            if (slot < kStageW) {
                st_val[slot] = c;
                st_row[slot] = (unsigned char)rl;
            } else {
                st_over[rl] = 1;
            }
        }
        base += __popcll(hit[reg]);
    }
}

// One wavefront empties its own staging region into its rows' candidate lists (exclusion bitmap tested here, off the
// MFMA path) — no barrier, no shared counter: the other three wavefronts of the workgroup are not involved.  Rows whose
// hits did not fit the region (more than kStageW / 2 scores of ONE panel passed in one wavefront: ties by the thousand,
// thresholds that cut nothing) are declared overflowed, so topk_finalize_kernel recomputes them exactly.  (Round 1
// appended such hits straight to global memory from inside the MFMA loop; the 16 per-register row pointers that path
// kept live cost ~90 VGPRs for something that happens on adversarial inputs only.)
__device__ __forceinline__ void wave_flush(const FusedArgs& a, int64_t m0, const unsigned long long* st_val,
                                           const unsigned char* st_row, unsigned char* st_over, int n, int lane, int wm,
                                           int* fl_cnt /* [32] of this wavefront */) {
    // One global atomic per (row, flush) instead of one per candidate: the region's entries are counted per query row in
    // LDS first (a wavefront's hits fall on its 32 rows), each row's count reserves its slots with a single add, then the
    // candidates go to base + rank.  (k = 256: ~760 candidates per row, 2 M per chunk.)
    constexpr int kPer = kStageW / 64;  // entries per lane
    if (lane < 32) fl_cnt[lane] = 0;
    unsigned long long c[kPer];
    int pos[kPer], rl[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const int e = lane + 64 * j;
        pos[j] = -1;
        rl[j] = 0;
        c[j] = 0ull;
        if (e < n) {
            c[j] = st_val[e];
            rl[j] = st_row[e] - wm * 32;
            const int64_t q = m0 + wm * 32 + rl[j];
            const uint32_t item = 0xFFFFFFFFu - (uint32_t)c[j];
            if (q < a.n_q && !((a.bitmap[q * a.words + (item >> 5)] >> (item & 31)) & 1u))  // (q >= n_q: a padding row, see the thresholds)
                pos[j] = atomicAdd(&fl_cnt[rl[j]], 1);
        }
    }
    int base = 0;
    if (lane < 32) {
        const int cnt = fl_cnt[lane];
        if (cnt) base = atomicAdd(&a.cnt[m0 + wm * 32 + lane], cnt);
    }
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
        const int b = __shfl(base, rl[j], 64);
        if (pos[j] >= 0) {
            const int64_t q = m0 + wm * 32 + rl[j];
            const int slot = b + pos[j];
            if (slot < kCap) a.cand[q * kCap + slot] = c[j];
        }
    }
    if (n >= kStageW && lane < 32) {  // the region ran full: push the marked rows of this wavefront's 32 queries past kCap.  Marks are
        const int r = wm * 32 + lane;  // never cleared (the twin wavefront shares them: clearing could lose its mark); adding twice is harmless
        if (st_over[r] && m0 + r < a.n_q) atomicAdd(&a.cnt[m0 + r], kCap + 1);
    }
}

// Step 3 — item panels by LDS-DMA, so that the MFMA chain gets its registers back.
// What the chain can do was measured in isolation (tools/probes/mfma_peak.hip, 2 waves per SIMD, 64 A fragments in
// registers, B operand from LDS, nothing else): register-only 0.98 of the 157 TF/s peak; one conflict-free ds_read_b32
// per MFMA from 129-float rows, compiler free to read ahead 0.975; ds_read_b128 chunks 0.93; any VALU instruction
// producing the B operand (a v_cndmask picking the half-wave's component of a b128 chunk) 0.83-0.87 — the MFMA issue
// slips behind the dependent VALU op every time.  The kernel above uses the good form of read but, at its 256-VGPR
// cap (64 A fragments, two accumulators, EIGHT float4 of the register-staged next panel, thresholds), leaves the
// compiler two registers for B: one read in flight, its latency exposed every second MFMA — 0.79 for its bare chain.
// So the next panel no longer passes through registers: `global_load_lds_dword` pieces of 64 floats (half a row; the
// LDS destination of a wave-instruction is contiguous, so dword pieces are what lets rows keep their 129-float pitch)
// go straight into the padded image, 32 per wavefront and panel, issued in front of the chain and waited for at its
// end.  (A 16-byte DMA needs an unpadded, XOR-swizzled image: its reads are b128 + select or 4-way conflicted b32,
// and 32 swizzled read addresses per buffer pinned 64 VGPRs — built and measured, no gain, removed.)
#ifndef MI_TOPK_DMA
#define MI_TOPK_DMA 1
#endif
#ifndef MI_TOPK_ONE_COPY
#define MI_TOPK_ONE_COPY 0   // A/B: one copy of the panel body instead of three (37 -> 20 KB of code): no faster (it is not the instruction cache)
#endif
#ifndef MI_TOPK_PIN
#define MI_TOPK_PIN 1   // B reads pinned a group of 8 MFMAs ahead (A/B)
#endif

__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Item panel rows [row0, row0 + 64) x D floats (all inside the table) -> the padded panel.  Inline asm on purpose: for
// the builtin the compiler drains the DMA (vmcnt(0)) in front of the next LDS read, i.e. before the MFMA chain it is
// supposed to run under; the caller waits itself (dma_wait) before its barrier.
template <int D>
__device__ __forceinline__ void dma_item_panel(const float* __restrict__ T, int64_t ld, int64_t row0,
                                               float (*panel)[FKPAD], int wave, uint32_t lane_off) {
    constexpr int PIECES = D / 64;  // 256-byte pieces per row: the second one rides on the instruction offset, which moves
                                    // the global and the LDS address alike
    const char* gb = reinterpret_cast<const char*>(T + (row0 + wave * 16) * ld);
    uint32_t l = lds_addr(&panel[wave * 16][0]);
    const int64_t row_bytes = ld * 4;
    // opaque to the optimiser: otherwise the addresses of the unrolled loop are hoisted out of the panel loop as loop
    // invariants — ~100 SGPRs, spilled to VGPR lanes and read back (v_readlane) in front of every MFMA chain.  That, not
    // anything in the vector code, was 120 us of a 741 us chunk (profiles/r02_topk.md): 741 -> 716 with the addresses
    // opaque, -> 621 with two scalar adds per row and no spill left
    asm volatile("" : "+s"(l), "+s"(gb));
#pragma unroll
    for (int r = 0; r < 16; ++r) {  // a wavefront moves 16 of the panel's 64 rows
        uint32_t keep;  // m0 is the compiler's: saved and put back (naming it as a clobber is not honoured)
        if (PIECES == 2)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %2, %3\n\t"
                         "global_load_lds_dword %2, %3 offset:256\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "s"(l), "v"(lane_off), "s"(gb) : "memory");
        else
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %2, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "s"(l), "v"(lane_off), "s"(gb) : "memory");
        l += FKPAD * 4;
        gb += row_bytes;
    }
}

template <int NM>
__global__ __launch_bounds__(256, 2) void topk_scores_filter_dma_kernel(FusedArgs a) {
    constexpr int D = 2 * NM;
    __shared__ float P0[FM][FKPAD];   // the query panel first, then item buffer 1
    __shared__ float P1[FN][FKPAD];   // item buffer 0
    __shared__ unsigned long long st_val[kStage];
    __shared__ unsigned char st_row[kStage];
    __shared__ unsigned char st_over[FM];
    __shared__ int fl_cnt[4][32];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int64_t m0 = (int64_t)blockIdx.y * FM;
    const int64_t n_tiles = (a.n_items + FN - 1) / FN;
    const int64_t t0 = (int64_t)blockIdx.x * a.tiles_per_slice;
    const int64_t t1 = min(n_tiles, t0 + a.tiles_per_slice);
    if (t0 >= t1) return;
    // a panel is always 64 rows inside the table: the last one is shifted back and its rows below t * 64 (already
    // scored by the panel before it) are masked out of the votes
    const int64_t last_row0 = a.n_items - FN;
    const uint32_t lane_off = (uint32_t)lane * 4u;
    dma_item_panel<D>(a.I, a.ldi, min(t0 * FN, last_row0), P1, wave, lane_off);
    {
        float4 va[8];
        fpanel_issue(va, a.U, a.ldu, a.uid, m0, a.n_q, a.d, tid);
        fpanel_commit(P0, va, tid);
    }
    if (tid < FM) st_over[tid] = 0;
    float thr_f[16];
    const int row_base = wm * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int64_t gm = m0 + row_base + (reg & 3) + 8 * (reg >> 2);
        thr_f[reg] = (gm < a.n_q && MI_TOPK_STAGE != 8) ? key_score(a.thr[gm]) : INFINITY;  // (stage 8: nothing ever passes)
    }
    // (padding rows — queries beyond n_q, copies of row 0 of the panel's zero fill — have threshold +inf: only a NaN score
    // passes there, and wave_flush drops entries of rows >= n_q)
    dma_wait();
    __syncthreads();
    float areg[NM];
    {
        const float* ap = &P0[wm * 32 + (lane & 31)][lane >> 5];
#pragma unroll
        for (int s = 0; s < NM; ++s) areg[s] = ap[2 * s];
    }
    __syncthreads();  // P0 is free: it becomes item buffer 1
    const int boff = (wn * 32 + (lane & 31)) * FKPAD + (lane >> 5);
    const int col = wn * 32 + (lane & 31);
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc1[i] = 0.f;
    PipeState st;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) st.hit[reg] = 0ull;
    st.col_ok_prev = 0ull;   // no previous panel yet: every vote comes out empty
    st.gn_prev = 0;
    st.mine = 0;
    unsigned long long* my_val = st_val + wave * kStageW;   // this wavefront's staging region
    unsigned char* my_row = st_row + wave * kStageW;

#define MI_TOPK_PANEL(acc, prev, t, cur, nxt)                                                                          \
    {                                                                                                                  \
        if ((MI_TOPK_STAGE < 2 || MI_TOPK_STAGE == 6 || MI_TOPK_STAGE == 8) && (t) + 1 < t1)  /* block-uniform */                              \
            dma_item_panel<D>(a.I, a.ldi, min(((t) + 1) * FN, last_row0), nxt, wave, lane_off);                        \
        pipe_panel<NM, MI_TOPK_PIN != 0>(a, acc, prev, areg, &cur[0][0] + boff, thr_f, st, lane, row_base,       \
                                         my_val, my_row, st_over);                                                     \
        st.gn_prev = min((t) * FN, last_row0) + col;                                                                   \
        st.col_ok_prev = __ballot(st.gn_prev >= (t) * FN);                                                             \
        if (MI_TOPK_STAGE >= 1 && MI_TOPK_STAGE < 6 && prev[0] == 12345.678f && prev[7] == 3.f) st_over[0] = 1; /* keeps the chain alive */ \
        if (st.mine >= kStageW / 2) {  /* wave-uniform: this wavefront's region is half full */                        \
            wave_flush(a, m0, my_val, my_row, st_over, min(st.mine, kStageW), lane, wm, fl_cnt[wave]);                 \
            st.mine = 0;                                                                                               \
        }                                                                                                              \
        if (MI_TOPK_STAGE < 5 || MI_TOPK_STAGE == 6 || MI_TOPK_STAGE == 8) dma_wait();                                                       \
        if (MI_TOPK_STAGE < 3 || MI_TOPK_STAGE == 6 || MI_TOPK_STAGE == 8) __syncthreads(); /* next panel landed and everybody is done with cur */ \
    }
#if MI_TOPK_STAGE == 5
    const unsigned long long dbg_c0 = clock64(), dbg_w0 = wall_clock64();
#endif
#if MI_TOPK_ONE_COPY
    // one copy of the panel body (the accumulator handed over by 16 moves per panel): a third of the code
    for (int64_t t = t0; t < t1; ++t) {
        const bool odd = ((t - t0) & 1) != 0;
        float (*cur)[FKPAD] = odd ? P0 : P1;
        float (*nxt)[FKPAD] = odd ? P1 : P0;
        MI_TOPK_PANEL(acc0, acc1, t, cur, nxt)
        acc1 = acc0;
    }
    const bool last_in_acc0 = false;
#else
    int64_t t = t0;
    for (; t + 1 < t1; t += 2) {
        MI_TOPK_PANEL(acc0, acc1, t, P1, P0)
        MI_TOPK_PANEL(acc1, acc0, t + 1, P0, P1)
    }
    bool last_in_acc0 = false;
    if (t < t1) {
        MI_TOPK_PANEL(acc0, acc1, t, P1, P0)
        last_in_acc0 = true;
    }
#endif
#undef MI_TOPK_PANEL
#if MI_TOPK_STAGE == 5
    if ((blockIdx.x == 0 || blockIdx.x == 7) && (blockIdx.y == 0 || blockIdx.y == 30) && tid == 0)
        printf("[stage5] wg (%d,%d): %llu shader cycles, %llu ticks of 10 ns for %lld panels\n", (int)blockIdx.x, (int)blockIdx.y,
               (unsigned long long)(clock64() - dbg_c0), (unsigned long long)(wall_clock64() - dbg_w0), (long long)(t1 - t0));
#endif
    {   // the last panel's scores have not been voted on yet
        unsigned long long hit[16];
        int total = 0;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const float sc = last_in_acc0 ? acc0[reg] : acc1[reg];
            hit[reg] = __ballot(!(sc < thr_f[reg])) & st.col_ok_prev;
            total += __popcll(hit[reg]);
        }
        if (total) {
            if (last_in_acc0) stage_hits(acc0, hit, total, st.gn_prev, row_base, lane, st, my_val, my_row, st_over);
            else stage_hits(acc1, hit, total, st.gn_prev, row_base, lane, st, my_val, my_row, st_over);
        }
    }
    wave_flush(a, m0, my_val, my_row, st_over, min(st.mine, kStageW), lane, wm, fl_cnt[wave]);
}

// The sampled scores the thresholds come from (n_q x kSample against the gathered sample rows `Is`, ld = d): the same
// tile machinery — query fragments in registers, sample panels by LDS-DMA, pinned B reads — with a plain store of the
// accumulator instead of the votes.  Same MFMA, same k order as mi_gemm_launch's kernel: the scores are bitwise the
// ones it wrote (76 us per 2 684-query chunk there, one panel per workgroup and no overlap of loads and MFMAs).
template <int NM>
__global__ __launch_bounds__(256, 2) void topk_sample_scores_kernel(FusedArgs a, const float* __restrict__ Is,
                                                                    float* __restrict__ out, int tiles_per_slice) {
    constexpr int D = 2 * NM;
    __shared__ float P0[FM][FKPAD];   // the query panel first, then sample buffer 1
    __shared__ float P1[FN][FKPAD];   // sample buffer 0
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int64_t m0 = (int64_t)blockIdx.y * FM;
    const int t0 = (int)blockIdx.x * tiles_per_slice;
    const int t1 = min(kSample / FN, t0 + tiles_per_slice);
    if (t0 >= t1) return;
    const uint32_t lane_off = (uint32_t)lane * 4u;
    dma_item_panel<D>(Is, D, (int64_t)t0 * FN, P1, wave, lane_off);
    {
        float4 va[8];
        fpanel_issue(va, a.U, a.ldu, a.uid, m0, a.n_q, a.d, tid);
        fpanel_commit(P0, va, tid);
    }
    dma_wait();
    __syncthreads();
    float areg[NM];
    {
        const float* ap = &P0[wm * 32 + (lane & 31)][lane >> 5];
#pragma unroll
        for (int s = 0; s < NM; ++s) areg[s] = ap[2 * s];
    }
    __syncthreads();  // P0 is free: it becomes sample buffer 1
    const int boff = (wn * 32 + (lane & 31)) * FKPAD + (lane >> 5);
    const int col = wn * 32 + (lane & 31);
    const int row_base = wm * 32 + 4 * (lane >> 5);
    for (int t = t0; t < t1; ++t) {
        const bool odd = ((t - t0) & 1) != 0;
        float (*cur)[FKPAD] = odd ? P0 : P1;
        float (*nxt)[FKPAD] = odd ? P1 : P0;
        if (t + 1 < t1) dma_item_panel<D>(Is, D, (int64_t)(t + 1) * FN, nxt, wave, lane_off);  // block-uniform
        const float* bp = &cur[0][0] + boff;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        float bq[2][8];
#pragma unroll
        for (int u = 0; u < 8; ++u) bq[0][u] = bp[2 * u];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < NM; ++s) pipe_mfma<NM, true>(s, acc, areg, bp, bq);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int64_t gm = m0 + row_base + (reg & 3) + 8 * (reg >> 2);
            if (gm < a.n_q) out[gm * kSample + t * FN + col] = acc[reg];
        }
        dma_wait();
        __syncthreads();  // next panel landed and everybody is done with cur
    }
}

__global__ __launch_bounds__(kBlock) void topk_finalize_kernel(FusedArgs a, int k, int kpow2, float* __restrict__ scores,
                                                               int64_t* __restrict__ out_idx,
                                                               float* __restrict__ out_score) {
    __shared__ SelectShared sh;
    const int64_t q = blockIdx.x;
    if (q >= a.n_q) return;
    const int tid = threadIdx.x;
    const int kk = (int)min((int64_t)k, a.n_items);
    const int cnt = a.cnt[q];
    if (cnt >= kk && cnt <= kCap) {  // every winner is in the list
        for (int i = tid; i < cnt; i += kBlock) sh.cand[i] = a.cand[q * kCap + i];
        __syncthreads();
#if defined(MI_TOPK_PROBE) && MI_TOPK_PROBE == 2   // timing probe only: the loads, no selection
        if (tid < k) out_idx[q * k + tid] = (int64_t)sh.cand[tid];
        return;
#endif
        finish_candidates(cnt, k, kk, kpow2, q, out_idx, out_score, sh);
        return;
    }
    // rare: materialise this row's scores — sequential fma chain over k, bit for bit the MFMA's — then select exactly
    float* row = scores + q * a.n_items;
    const float* u = a.U + a.uid[q] * a.ldu;
    for (int64_t i = tid; i < a.n_items; i += kBlock) {
        const float* it = a.I + i * a.ldi;
        float acc = 0.f;
        for (int c = 0; c < a.d; ++c) acc = fmaf(u[c], it[c], acc);
        if ((a.bitmap[q * a.words + (i >> 5)] >> (i & 31)) & 1u) acc = -INFINITY;
        row[i] = acc;
    }
    __syncthreads();  // workgroup-scope: the row just written is read back by the whole block
    select_row_exact(row, a.n_items, k, kk, kpow2, q, out_idx, out_score, sh);
}

#include "topk_prefilter.hpp"

}  // namespace

static int mi_cu_count() {
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n_cu = v;
        else n_cu = 256;
    }
    return n_cu;
}

// bf16x3 prefilter (topk_prefilter.hpp): split tables, thresholds, slice-partitioned lists and their counters
static size_t topk_prefilter_bytes(int64_t n_q, int64_t n_items) {
    const size_t q_pad = (size_t)mi_ceil_div(n_q, 256) * 256, i_pad = (size_t)mi_ceil_div(n_items, 64) * 64;
    return mi_align_up(i_pad * 128 * sizeof(float), 256) + mi_align_up(q_pad * 128 * sizeof(float), 256) +
           mi_align_up((size_t)kSample * 128 * sizeof(float), 256) + 2 * mi_align_up(q_pad * sizeof(float), 256) + 256 +
           mi_align_up(q_pad * kPreCap * sizeof(unsigned long long), 256) + mi_align_up(q_pad * 512 * sizeof(int), 256);
}

static bool topk_prefilter_on() {
    const char* e = getenv("LAPLACE_TOPK_PREFILTER");   // "0": the f32 fused path (A/B, tests of both); read per call
    return !(e && e[0] == '0');
}

// Grid of the bf16 kernel: `strips` groups of 256 register-side rows x 8 sl slices of the panel axis.  One workgroup per CU
// (its LDS ring), 8 XCDs: the workgroups of an XCD run ceil(strips * sl / per_xcd) rounds of panels / (8 sl) panels each
// (+ the prologue, ~2 panels' worth).
static int64_t topk_pre_slices_per_xcd(int64_t strips, int64_t panels, int n_cu) {
    const int64_t per_xcd = std::max(1, n_cu / 8);
    int64_t sl = 1, best = INT64_MAX;
    for (int64_t l = 1; l <= 32; ++l) {
        const int64_t cost = mi_ceil_div(strips * l, per_xcd) * (mi_ceil_div(panels, 8 * l) + 2);
        if (cost < best) { best = cost; sl = l; }
    }
    return sl;
}

// The prefilter kernel needs 96 KB (D = 128) of dynamic LDS: an opt-in attribute of the kernel, PER DEVICE.  Queried and set
// before anything of a call is enqueued (a failure used to surface after the memset and split kernels were in
// the stream, and one process-wide flag covered every device); on failure the caller takes the f32 fused path.
template <int D, bool STORE>
static bool topk_pre_attr_ok() {
    constexpr int UB = STORE ? 2 : MI_PRE_UB;
    constexpr int lds = 3 * 64 * (D / 4) * 16;
    static unsigned char state[64] = {};   // per device: 0 = not tried, 1 = set, 2 = refused (idempotent: a race sets it twice)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    if (state[dev] == 0) {
        auto kern = topk_prefilter_bf16_kernel<D, STORE, UB>;
        state[dev] = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess ? 1 : 2;
    }
    return state[dev] == 1;
}
static bool topk_prefilter_usable(int64_t d) {
    return d == 128 ? (topk_pre_attr_ok<128, true>() && topk_pre_attr_ok<128, false>())
                    : (d == 64 && topk_pre_attr_ok<64, true>() && topk_pre_attr_ok<64, false>());
}

template <int D, bool STORE>
static int topk_pre_kernel_launch(PreArgs& pa, int64_t strips, int64_t panels, int n_cu, hipStream_t s) {
    const int64_t sl = topk_pre_slices_per_xcd(strips, panels, n_cu);
    pa.panels = panels;
    pa.strips = (int)strips; pa.n_slices = (int)(8 * sl); pa.cap_s = 2;
    while (pa.cap_s * 2 * pa.n_slices <= kPreCap) pa.cap_s *= 2;
    pa.panels_per_slice = mi_ceil_div(panels, pa.n_slices);
    constexpr int UB = STORE ? 2 : MI_PRE_UB;   // (the STORE form spills at two wavefronts per SIMD)
    auto kern = topk_prefilter_bf16_kernel<D, STORE, UB>;
    constexpr int lds = 3 * 64 * (D / 4) * 16;
    if (!topk_pre_attr_ok<D, STORE>()) return MI_ERR_UNSUPPORTED;   // callers have asked topk_prefilter_usable() before enqueuing anything
    hipLaunchKernelGGL(kern, dim3((unsigned)(8 * sl * strips)), dim3(512 / UB), lds, s, pa);
    return 0;
}

template <int D>
static void topk_split_launch(int64_t n_rows, int64_t n_pad, const float* T, int64_t ld, const int64_t* row_map, uint2* out,
                              uint32_t* n2max, int mode, float* thrf, float* epsv, int n_cu, hipStream_t s) {
    constexpr int RPB = 256 / (D / 4);
    hipLaunchKernelGGL(topk_split_rows_kernel<D>, dim3((unsigned)std::min<int64_t>(mi_ceil_div(n_pad, RPB), 4 * n_cu)), dim3(256),
                       0, s, n_rows, n_pad, T, ld, row_map, out, n2max, mode, thrf, epsv);
}

// The whole prefilter path after the sample rows have been gathered (Is: [kSample, d] f32): split tables, sample scores in
// bf16x3 (the STORE form of the kernel: panel side = the queries, register side = the sample), thresholds, lists, refine.
template <int D>
static int topk_prefilter_launch(FusedArgs& a, MiArena& ar, const float* Is, float* sample_scores, uint32_t* thr,
                                 const int32_t* excl_ptr, const int32_t* excl_idx, int k, int kpow2, float* scores,
                                 int64_t* out_idx, float* out_score, int n_cu, bool items_ready, hipStream_t s) {
    const int64_t strips = mi_ceil_div(a.n_q, 256), q_pad = strips * 256;
    const int64_t panels = mi_ceil_div(a.n_items, 64), i_pad = panels * 64;
    uint2* Ib = reinterpret_cast<uint2*>(ar.take<float>((size_t)i_pad * 128));
    uint2* Ub = reinterpret_cast<uint2*>(ar.take<float>((size_t)q_pad * 128));
    uint2* Sb = reinterpret_cast<uint2*>(ar.take<float>((size_t)kSample * 128));
    float* thrf = ar.take<float>((size_t)q_pad);
    float* epsv = ar.take<float>((size_t)q_pad);
    uint32_t* n2max = ar.take<uint32_t>(64);
    unsigned long long* pre = ar.take<unsigned long long>((size_t)q_pad * kPreCap);
    int* pre_cnt = ar.take<int>((size_t)q_pad * 512);
    if (!Ib || !Ub || !Sb || !thrf || !epsv || !n2max || !pre || !pre_cnt) return MI_ERR_WORKSPACE;
    if (!items_ready) {   // the item side: the same for every chunk of a call (MI_TOPK_ITEMS_PREPARED)
        MI_HIP(hipMemsetAsync(n2max, 0, sizeof(uint32_t), s));
        topk_split_launch<D>(a.n_items, i_pad, a.I, a.ldi, nullptr, Ib, n2max, kSplitItems, nullptr, nullptr, n_cu, s);
        topk_split_launch<D>(kSample, kSample, Is, D, nullptr, Sb, n2max, kSplitPlain, nullptr, nullptr, n_cu, s);
    }
    topk_split_launch<D>(a.n_q, q_pad, a.U, a.ldu, a.uid, Ub, n2max, kSplitQueries, thrf, epsv, n_cu, s);
    PreArgs pa;
    memset(&pa, 0, sizeof(pa));
    pa.n_q = a.n_q; pa.n_items = a.n_items;
    // sample scores: out[query][sample]
    pa.Ub = reinterpret_cast<const uint4*>(Sb); pa.Ib = reinterpret_cast<const uint4*>(Ub);
    pa.out = sample_scores; pa.ldo = kSample; pa.n_a = a.n_q; pa.n_b = kSample;
    int rc = topk_pre_kernel_launch<D, true>(pa, kSample / 256, q_pad / 64, n_cu, s);
    if (rc) return rc;
    hipLaunchKernelGGL(threshold_kernel, dim3((unsigned)a.n_q), dim3(kBlock), 0, s, a.n_q, a.n_items, k, sample_scores,
                       excl_ptr, excl_idx, const_cast<uint32_t*>(a.bitmap), a.words, thr, a.cnt, epsv, thrf);
    a.thr = thr;
    // candidate lists
    pa.Ub = reinterpret_cast<const uint4*>(Ub); pa.Ib = reinterpret_cast<const uint4*>(Ib);
    pa.thrf = thrf; pa.pre = pre; pa.pre_cnt = pre_cnt;
    pa.out = nullptr;
    rc = topk_pre_kernel_launch<D, false>(pa, strips, panels, n_cu, s);
    if (rc) return rc;
    hipLaunchKernelGGL(topk_refine_finalize_kernel, dim3((unsigned)a.n_q), dim3(kBlock), 0, s, a, pa, epsv, k, kpow2, scores,
                       out_idx, out_score);
    return mi_launch_status();
}

// Diagnostic of the prefilter (mi_topk_prefilter_scores_f32): the approximate scores and the bound they come with.
template <int D>
static int topk_prefilter_scores(int64_t n_q, int64_t n_items, const int64_t* uid, const float* U, int64_t ldu, const float* I,
                                 int64_t ldi, float* out, float* eps_out, MiArena& ar, int n_cu, hipStream_t s) {
    const int64_t strips = mi_ceil_div(n_items, 256), i_pad = strips * 256;    // register side: the items
    const int64_t panels = mi_ceil_div(n_q, 64), q_pad = panels * 64;          // panel side: the queries
    uint2* Ib = reinterpret_cast<uint2*>(ar.take<float>((size_t)i_pad * 128));
    uint2* Ub = reinterpret_cast<uint2*>(ar.take<float>((size_t)q_pad * 128));
    float* epsv = ar.take<float>((size_t)q_pad);
    uint32_t* n2max = ar.take<uint32_t>(64);
    if (!Ib || !Ub || !epsv || !n2max) return MI_ERR_WORKSPACE;
    MI_HIP(hipMemsetAsync(n2max, 0, sizeof(uint32_t), s));
    topk_split_launch<D>(n_items, i_pad, I, ldi, nullptr, Ib, n2max, kSplitItems, nullptr, nullptr, n_cu, s);
    topk_split_launch<D>(n_q, q_pad, U, ldu, uid, Ub, n2max, kSplitQueries, nullptr, epsv, n_cu, s);
    PreArgs pa;
    memset(&pa, 0, sizeof(pa));
    pa.n_q = n_q; pa.n_items = n_items;
    pa.Ub = reinterpret_cast<const uint4*>(Ib); pa.Ib = reinterpret_cast<const uint4*>(Ub);
    pa.out = out; pa.ldo = n_items; pa.n_a = n_q; pa.n_b = n_items;
    const int rc = topk_pre_kernel_launch<D, true>(pa, strips, panels, n_cu, s);
    if (rc) return rc;
    MI_HIP(hipMemcpyAsync(eps_out, epsv, (size_t)n_q * sizeof(float), hipMemcpyDeviceToDevice, s));
    return mi_launch_status();
}

extern "C" {

static size_t topk_fused_extra_bytes(int64_t n_q, int64_t n_items) {
    const size_t words = (size_t)((n_items + 31) / 32);
    return mi_align_up((size_t)kSample * 128 * sizeof(float), 256) +             // sampled item rows (d <= 128)
           mi_align_up((size_t)n_q * kSample * sizeof(float), 256) +             // sample scores
           mi_align_up((size_t)n_q * words * sizeof(uint32_t), 256) +            // exclusion bitmap
           2 * mi_align_up((size_t)n_q * sizeof(uint32_t), 256) +                // thresholds, counters
           mi_align_up((size_t)n_q * kCap * sizeof(unsigned long long), 256) +   // candidate lists
           topk_prefilter_bytes(n_q, n_items);
}

size_t mi_topk_workspace_bytes(int64_t n_q, int64_t n_items, int64_t k) {
    (void)k;
    if (n_q <= 0 || n_items <= 0) return 256;
    return mi_align_up((size_t)n_q * (size_t)n_items * sizeof(float), 256) + topk_fused_extra_bytes(n_q, n_items);
}

int mi_topk_excl_f32(int64_t n_q, int64_t n_items, int64_t d, int64_t k, const int64_t* uid,
                     const float* user_emb, int64_t ldu, const float* item_emb, int64_t ldi,
                     const int32_t* excl_ptr, const int32_t* excl_idx, int64_t* out_idx,
                     float* out_score, void* ws, size_t ws_bytes, mi_stream_t stream) {
    return mi_topk_excl_ex_f32(n_q, n_items, d, k, uid, user_emb, ldu, item_emb, ldi, excl_ptr, excl_idx, out_idx, out_score, ws, ws_bytes,
                               0u, stream);
}

int mi_topk_excl_ex_f32(int64_t n_q, int64_t n_items, int64_t d, int64_t k, const int64_t* uid,
                        const float* user_emb, int64_t ldu, const float* item_emb, int64_t ldi,
                        const int32_t* excl_ptr, const int32_t* excl_idx, int64_t* out_idx,
                        float* out_score, void* ws, size_t ws_bytes, uint32_t flags, mi_stream_t stream) {
    MI_CHECK_ARG(n_q >= 0 && n_items >= 0 && d > 0 && k > 0 && (flags & ~(uint32_t)MI_TOPK_ITEMS_PREPARED) == 0);
    if (n_q == 0) return 0;
    if (k > kMaxK) return MI_ERR_UNSUPPORTED;
    if (n_items >= INT32_MAX) return MI_ERR_TOO_LARGE;
    MI_CHECK_ARG(user_emb && item_emb && out_idx && ws && ldu >= d && ldi >= d);
    // excl_idx may be null when every exclusion row is empty (excl_ptr all equal)
    if (ws_bytes < mi_topk_workspace_bytes(n_q, n_items, k)) return MI_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    float* scores = static_cast<float*>(ws);
    int kpow2 = 2;
    while (kpow2 < k) kpow2 <<= 1;
    const bool fused = MI_TOPK_FUSED && n_items >= 8 * kSample && d <= FKC && d % 4 == 0 && ldu % 4 == 0 && ldi % 4 == 0 &&
                       mi_aligned16(user_emb) && mi_aligned16(item_emb);
    if (fused) {
        MiArena ar(static_cast<char*>(ws) + mi_align_up((size_t)n_q * (size_t)n_items * sizeof(float), 256),
                   topk_fused_extra_bytes(n_q, n_items));
        const int64_t words = (n_items + 31) / 32;
        float* Is = ar.take<float>((size_t)kSample * 128);
        float* sample_scores = ar.take<float>((size_t)n_q * kSample);
        uint32_t* bitmap = ar.take<uint32_t>((size_t)n_q * words);
        uint32_t* thr = ar.take<uint32_t>((size_t)n_q);
        int* cnt = ar.take<int>((size_t)n_q);
        unsigned long long* cand = ar.take<unsigned long long>((size_t)n_q * kCap);
        if (!Is || !sample_scores || !bitmap || !thr || !cnt || !cand) return MI_ERR_WORKSPACE;
        const bool pre_path = topk_prefilter_on() && (d == 128 || d == 64) && k <= kPreMaxK && topk_prefilter_usable(d);
        // MI_TOPK_ITEMS_PREPARED: the item side of the prefilter (sample rows, the bf16 split of the item table and of the
        // sample, the largest |item|^2) is still in `ws` from the previous call: nothing of it is recomputed
        const bool items_ready = pre_path && (flags & MI_TOPK_ITEMS_PREPARED) != 0;
        if (!items_ready)
            hipLaunchKernelGGL(gather_sample_rows_kernel, dim3(kSample), dim3(64), 0, s, n_items, (int)d, item_emb, ldi, Is);
        FusedArgs a;
        a.n_q = n_q; a.n_items = n_items; a.d = (int)d; a.uid = uid;
        a.U = user_emb; a.ldu = ldu; a.I = item_emb; a.ldi = ldi;
        if (pre_path) {
            // bf16x3 prefilter (topk_prefilter.hpp): sample scores, thresholds, lists and the exact finish all in there
            MI_HIP(hipMemsetAsync(bitmap, 0, (size_t)n_q * words * sizeof(uint32_t), s));
            a.thr = thr; a.bitmap = bitmap; a.words = words; a.cand = cand; a.cnt = cnt;
            if (d == 128)
                return topk_prefilter_launch<128>(a, ar, Is, sample_scores, thr, excl_ptr, excl_idx, (int)k, kpow2, scores, out_idx,
                                                  out_score, mi_cu_count(), items_ready, s);
            return topk_prefilter_launch<64>(a, ar, Is, sample_scores, thr, excl_ptr, excl_idx, (int)k, kpow2, scores, out_idx,
                                             out_score, mi_cu_count(), items_ready, s);
        }
        const int64_t strips = mi_ceil_div(n_q, FM);
        const int64_t capacity = 2 * (int64_t)mi_cu_count();
        if (MI_TOPK_DMA && (d == 128 || d == 64)) {
            int64_t sl = 1, best = INT64_MAX;  // split of the 64 sample panels: rounds x (panels + prologue), as below
            for (int64_t l = 1; l <= kSample / FN; ++l) {
                const int64_t cost = mi_ceil_div(strips * l, capacity) * (mi_ceil_div(kSample / FN, l) + 2);
                if (cost < best) { best = cost; sl = l; }
            }
            const int tps = (int)mi_ceil_div(kSample / FN, sl);
            sl = mi_ceil_div(kSample / FN, tps);
            if (d == 128)
                hipLaunchKernelGGL(topk_sample_scores_kernel<64>, dim3((unsigned)sl, (unsigned)strips), dim3(256), 0, s, a, Is, sample_scores, tps);
            else
                hipLaunchKernelGGL(topk_sample_scores_kernel<32>, dim3((unsigned)sl, (unsigned)strips), dim3(256), 0, s, a, Is, sample_scores, tps);
        } else {
            MiGemmArgs g;
            g.M = n_q; g.N = kSample; g.K = d;
            g.A = user_emb; g.sa_m = ldu; g.sa_k = 1; g.a_rows = uid;
            g.B = Is; g.sb_n = d; g.sb_k = 1;
            g.bias = nullptr; g.C = sample_scores; g.ldc = kSample; g.accumulate = 0; g.act = 0;
            int rc = mi_gemm_launch(g, nullptr, 0, s);
            if (rc) return rc;
        }
        MI_HIP(hipMemsetAsync(bitmap, 0, (size_t)n_q * words * sizeof(uint32_t), s));
        hipLaunchKernelGGL(threshold_kernel, dim3((unsigned)n_q), dim3(kBlock), 0, s, n_q, n_items, (int)k, sample_scores,
                           excl_ptr, excl_idx, bitmap, words, thr, cnt, (const float*)nullptr, (float*)nullptr);
        a.thr = thr; a.bitmap = bitmap; a.words = words; a.cand = cand; a.cnt = cnt;
        const int64_t n_tiles = mi_ceil_div(n_items, FN);
        // Every workgroup does the same work and two fit on a CU, so the launch runs in ceil(grid / (2 * CUs)) rounds of
        // tiles_per_slice panels each: a grid one workgroup over a multiple of the chip's capacity (the old rule,
        // >= 4 workgroups per CU, made 1 025 of them for a 2 621-query chunk) pays a whole extra round for it.  Pick the
        // split of the item axis that minimises rounds x (panels + the query-panel prologue, ~2 panels' worth).
        int64_t slices = 1, best = INT64_MAX;
        for (int64_t l = 1; l <= n_tiles && l <= 4096; ++l) {
            const int64_t cost = mi_ceil_div(strips * l, capacity) * (mi_ceil_div(n_tiles, l) + 2);
            if (cost < best) { best = cost; slices = l; }
        }
        a.tiles_per_slice = mi_ceil_div(n_tiles, slices);
        slices = mi_ceil_div(n_tiles, a.tiles_per_slice);
        if (MI_TOPK_DMA && d == 128)
            hipLaunchKernelGGL(topk_scores_filter_dma_kernel<64>, dim3((unsigned)slices, (unsigned)strips), dim3(256), 0, s, a);
        else if (MI_TOPK_DMA && d == 64)
            hipLaunchKernelGGL(topk_scores_filter_dma_kernel<32>, dim3((unsigned)slices, (unsigned)strips), dim3(256), 0, s, a);
        else
            hipLaunchKernelGGL(topk_scores_filter_kernel, dim3((unsigned)slices, (unsigned)strips), dim3(256), 0, s, a);
        hipLaunchKernelGGL(topk_finalize_kernel, dim3((unsigned)n_q), dim3(kBlock), 0, s, a, (int)k, kpow2, scores, out_idx,
                           out_score);
        return mi_launch_status();
    }
    MiGemmArgs g;
    g.M = n_q; g.N = n_items; g.K = d;
    g.A = user_emb; g.sa_m = ldu; g.sa_k = 1; g.a_rows = uid;
    g.B = item_emb; g.sb_n = ldi; g.sb_k = 1;
    g.bias = nullptr; g.C = scores; g.ldc = n_items; g.accumulate = 0; g.act = 0;
    int rc = mi_gemm_launch(g, nullptr, 0, s);  // M x N = queries x items fills the chip: never split
    if (rc) return rc;
    if (excl_ptr)
        hipLaunchKernelGGL(exclude_kernel, dim3((unsigned)n_q), dim3(kBlock), 0, s, n_q, n_items, excl_ptr, excl_idx, scores);
    hipLaunchKernelGGL(select_kernel, dim3((unsigned)n_q), dim3(kBlock), 0, s, n_q, n_items, (int)k, kpow2, scores,
                       out_idx, out_score, MI_TOPK_ONE_PASS);
    return mi_launch_status();
}

size_t mi_topk_prefilter_scores_workspace_bytes(int64_t n_q, int64_t n_items) {
    const size_t i_pad = (size_t)mi_ceil_div(n_items > 0 ? n_items : 1, 256) * 256, q_pad = (size_t)mi_ceil_div(n_q > 0 ? n_q : 1, 64) * 64;
    return mi_align_up(i_pad * 128 * sizeof(float), 256) + mi_align_up(q_pad * 128 * sizeof(float), 256) +
           mi_align_up(q_pad * sizeof(float), 256) + 256;
}

int mi_topk_prefilter_scores_f32(int64_t n_q, int64_t n_items, int64_t d, const int64_t* uid, const float* user_emb,
                                 int64_t ldu, const float* item_emb, int64_t ldi, float* scores, float* eps, void* ws,
                                 size_t ws_bytes, mi_stream_t stream) {
    MI_CHECK_ARG(n_q > 0 && n_items > 0 && user_emb && item_emb && scores && eps && ws);
    if (d != 64 && d != 128) return MI_ERR_UNSUPPORTED;
    if (n_items >= INT32_MAX) return MI_ERR_TOO_LARGE;
    MI_CHECK_ARG(ldu >= d && ldi >= d && ldu % 4 == 0 && ldi % 4 == 0 && mi_aligned16(user_emb) && mi_aligned16(item_emb));
    if (ws_bytes < mi_topk_prefilter_scores_workspace_bytes(n_q, n_items)) return MI_ERR_WORKSPACE;
    MiArena ar(ws, ws_bytes);
    hipStream_t s = (hipStream_t)stream;
    if (d == 128) return topk_prefilter_scores<128>(n_q, n_items, uid, user_emb, ldu, item_emb, ldi, scores, eps, ar, mi_cu_count(), s);
    return topk_prefilter_scores<64>(n_q, n_items, uid, user_emb, ldu, item_emb, ldi, scores, eps, ar, mi_cu_count(), s);
}

}  // extern "C"

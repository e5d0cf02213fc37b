// K10, bf16x3 PREFILTER of the fused top-K (included by topk.hip inside its anonymous namespace).
//
// The f32 MFMA chain of topk_scores_filter_dma_kernel is 0.71 of the f32 matrix peak and cannot go faster; the bf16
// matrix rate is 16x the f32 one.  So the scores that decide WHO IS A CANDIDATE are computed in bf16 with a proven error
// bound, and only the few candidates that can still reach the answer are scored again with the exact k-ordered f32 fma
// chain (the oracle's bits, utils/metrics_lightgcn.py:137 restated in oracle/).  The ids and scores returned are the
// same as on the f32 path, by construction:
//
//   x = hi + lo + r, hi = bf16(x), lo = bf16(x - hi):  |r| <= 2^-16 |x|.   s~ = sum_d uh*ih + ul*ih + uh*il  (bf16
//   products are exact in f32; the MFMA accumulates in f32).  |s~ - s_chain| <= eps(u) for every item, with
//     eps(u) = kPreC * |u|_2 * max_i |i|_2,   kPreC = 2^-12
//   (dropped terms ul*il, uh*ir, ur*i: <= 3.02 * 2^-16; f32 accumulation of 3*D products, D <= 128: <= 4.9e-5; the
//   exact chain's own rounding: <= D * 2^-24; Cauchy-Schwarz on sum |u_d||i_d|; total 1.05e-4 — kPreC is 2.3x that).
//
//   1. P = { i : s~_i >= thr - 3 eps }  (thr = the sampled f32 threshold of threshold_kernel).  Complete: nothing
//      outside P has s~ >= thr - 3 eps.
//   2. t2 <= the kk-th largest s~ in P.  kk items have s >= t2 - eps, so the exact kk-th largest score S_k >= t2 - eps,
//      and every item with s >= S_k has s~ >= t2 - 2 eps.  If t2 - 2 eps >= thr - 3 eps they are all in P: the SURVIVORS
//      { i in P : s~_i >= t2 - 2 eps } hold every winner (and every tie at the bound).  Otherwise the row takes the exact
//      path, like a row whose list came out short on the f32 path.
//   3. The survivors (about kk * |P| / 64 for kk <= 64, about kk above) are scored exactly and finish_candidates() picks
//      the answer from them.
//
// Layout.  Both tables are split once per call into rows [hi(0..D) | lo(0..D)] of bf16 (the f32 row's size).  The
// QUERIES are the MFMA's B operand and live in registers for the whole launch (a wavefront owns 64 of them: 128 VGPRs),
// the ITEMS are the A operand and stream through a ring of three 64-item panels in LDS (16-byte chunks XOR-swizzled by
// the row: conflict-free ds_read_b128).  ih is read once for two of the three products.  A lane holds ONE query per
// 32x32 block (C layout: column = lane & 31), so its threshold is one register and its list position a register
// counter: a query's candidates go straight to the (query, item-slice) region of the list that only this wavefront
// writes — no staging, no atomics, no flush.  The votes on panel t-1 are issued between the MFMAs of panel t.
// Workgroups that share an item slice are dealt to the same XCD, so that the slice is fetched into one L2.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifndef MI_PRE_PROBE
#define MI_PRE_PROBE 0   // timing probes (wrong results): 1 = vote branches with empty hit bodies, 2 = compares only (no branch)
#endif
#ifndef MI_PRE_Q_IN_AGPR
#define MI_PRE_Q_IN_AGPR 1   // the query fragments (128 registers, MFMA operands only) are steered into AGPRs so that the
#endif                       // accumulators — which the votes compare — can stay in VGPRs: no v_accvgpr_read per vote (A/B)
#ifndef MI_PRE_UB
#define MI_PRE_UB 1   // 32-query blocks per wavefront of the prefilter kernel: 2 = four wavefronts of 64 queries, one per SIMD (372
                      // registers), 1 = eight of 32 queries, two per SIMD (212 registers).  A/B per 4 096-query chunk: 314 / 355 us
                      // (k = 12 / 256) -> 264 / 305; 65 536 queries: 9.63 M / 8.07 M -> 10.56 M / 8.48 M users/s.  The lone
                      // wavefront spent a quarter of its cycles parked at the per-panel barrier and its LDS commit
                      // (SQ_WAIT_ANY, profiles/r03_topk_prefilter.md); now the SIMD's other wavefront issues MFMAs meanwhile
#endif
#ifndef MI_PRE_MAX_TEST
#define MI_PRE_MAX_TEST 0   // a vote group's wave-level test on the largest of its four scores (v_max3 + v_max + one compare: 144 fewer
                            // vector instructions per panel).  A/B: 318 / 363 -> 320 / 368 us per chunk (k = 12 / 256), 9.66 M -> 9.27 M
                            // users/s — the loop is not bound by its vector-instruction count.  Off.
#endif
#ifndef MI_PRE_ONE_PUT
#define MI_PRE_ONE_PUT 1   // a vote group's hits as one predicated store when no lane has two of the four (A/B)
#endif
constexpr float kPreC = 2.44140625e-4f;   // 2^-12, see above
constexpr int kPreCap = 8192;             // list slots per query in global memory, divided among the item slices
constexpr int kPreMaxK = 256;             // beyond: the f32 path (the survivors of a row must fit the finish step)
constexpr int kPreSurv = 2048;            // survivors per row the refine step takes
constexpr int kPreList = kCand - kPreSurv / 2;   // list entries per row the refine step takes (the rest of sh.cand holds the survivors' ids)

struct PreArgs {
    int64_t n_q, n_items;
    int64_t panels;             // 64-item panels of the padded item table
    const uint4* Ub;            // [strips * 256][D / 4]: 16-byte chunks, hi half then lo half
    const uint4* Ib;            // [panels * 64][D / 4]
    const float* thrf;          // [strips * 256]: thr - 3 eps; +inf for padding queries
    unsigned long long* pre;    // [strips * 256][kPreCap]: (item << 32) | bits(s~)
    int* pre_cnt;               // [strips * 256][n_slices][2]
    int strips, n_slices, cap_s;   // cap_s: list slots per (query, slice), a power of two; half of it per half-wavefront
    int64_t panels_per_slice;
    // STORE form (topk_prefilter_bf16_kernel<D, true>): no votes — the approximate scores themselves are written,
    // out[a_row * ldo + b_row] for the panel-side rows a_row < n_a (Ib) and the register-side rows b_row < n_b (Ub)
    float* out;
    int64_t ldo, n_a, n_b;
};

__device__ __forceinline__ uint32_t bf16_rn_bits(float x) {
    const uint32_t u = __float_as_uint(x);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (u >> 16) | 0x40u;  // NaN stays NaN
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}

// Rows of T (through row_map when given) -> [hi | lo] bf16 rows; rows in [n_rows, n_pad) are zero.
// mode 0 (items): the largest |row|^2 -> n2_max.  mode 1 (queries): eps(u) -> epsv, +inf -> thrf of the padding rows (the
// real rows' thresholds come from threshold_kernel).  mode 2: the split alone.
enum { kSplitItems = 0, kSplitQueries = 1, kSplitPlain = 2 };
template <int D>
__global__ __launch_bounds__(256) void topk_split_rows_kernel(int64_t n_rows, int64_t n_pad, const float* __restrict__ T,
                                                              int64_t ld, const int64_t* __restrict__ row_map,
                                                              uint2* __restrict__ out, uint32_t* __restrict__ n2_max,
                                                              int mode, float* __restrict__ thrf,
                                                              float* __restrict__ epsv) {
    constexpr int LPR = D / 4, RPB = 256 / LPR;
    __shared__ uint32_t blk_max[256 / MI_WAVE];
    const int li = threadIdx.x % LPR;
    uint32_t mx = 0u;   // bits of the largest |row|^2 this thread has seen (n2 >= 0: the bit patterns order like the values; NaN wins)
    // grid-stride over groups of RPB rows; every thread of a sub-group takes the same trips
    for (int64_t r = (int64_t)blockIdx.x * RPB + threadIdx.x / LPR; r < n_pad; r += (int64_t)gridDim.x * RPB) {
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < n_rows) {
            const int64_t row = row_map ? row_map[r] : r;
            x = *reinterpret_cast<const float4*>(T + row * ld + 4 * li);
        }
        const float xs[4] = {x.x, x.y, x.z, x.w};
        uint32_t hb[4], lb[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            hb[c] = bf16_rn_bits(xs[c]);
            lb[c] = bf16_rn_bits(xs[c] - __uint_as_float(hb[c] << 16));   // exact difference
        }
        out[r * (D / 2) + li] = make_uint2(hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16));
        out[r * (D / 2) + D / 4 + li] = make_uint2(lb[0] | (lb[1] << 16), lb[2] | (lb[3] << 16));
        float n2 = x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
#pragma unroll
        for (int m = LPR / 2; m > 0; m >>= 1) n2 += __shfl_xor(n2, m, LPR);
        if (mode == kSplitQueries) {
            if (li == 0) {
                if (r < n_rows) epsv[r] = kPreC * 1.01f * sqrtf(n2) * sqrtf(__uint_as_float(*n2_max));
                else if (thrf) thrf[r] = INFINITY;
            }
        } else if (mode == kSplitItems && r < n_rows) {
            mx = max(mx, __float_as_uint(n2));
        }
    }
    if (mode != kSplitItems) return;   // kernel-uniform
    // items: one atomic per workgroup (one per row serialised 10^5 of them on one address: 570 us)
#pragma unroll
    for (int m = MI_WAVE / 2; m > 0; m >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, m, MI_WAVE));
    if ((threadIdx.x & (MI_WAVE - 1)) == 0) blk_max[threadIdx.x / MI_WAVE] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t b = blk_max[0];
        for (int w = 1; w < 256 / MI_WAVE; ++w) b = max(b, blk_max[w]);
        atomicMax(n2_max, b);
    }
}

// One score of the previous panel that reached its query's threshold: (item, s~) goes to the lane's own region — the
// (query, item slice, half-wavefront) part of the query's list, which no other lane writes, so the fill is a register
// counter.  Slots wrap (cap is a power of two): a region that overflows loses entries but its count runs on, which is
// how the refine step sees it.
__device__ __forceinline__ void pre_put(bool p, float v, uint32_t item, int& cnt, unsigned long long* __restrict__ region,
                                        int mask) {
    if (p) region[cnt & mask] = ((unsigned long long)item << 32) | (unsigned long long)__float_as_uint(v);
    cnt += p ? 1 : 0;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int PIECES, int THREADS>
__device__ __forceinline__ void pre_issue(u32x4 (&g)[PIECES], const uint4* __restrict__ src) {
#pragma unroll
    for (int j = 0; j < PIECES; ++j) g[j] = *reinterpret_cast<const u32x4*>(src + THREADS * j);
}
template <int PIECES>
__device__ __forceinline__ void pre_commit(const u32x4 (&g)[PIECES], unsigned char* buf, const uint32_t (&woff)[PIECES]) {
#pragma unroll
    for (int j = 0; j < PIECES; ++j) *reinterpret_cast<u32x4*>(buf + woff[j]) = g[j];
}

// UB = 32-query blocks per wavefront: 2 = four wavefronts per workgroup, one per SIMD (372 registers); 1 = eight wavefronts,
// two per SIMD (<= 256 registers each), so that one's barrier / LDS commit / hit bodies run under the other's MFMAs.
template <int D, bool STORE, int UB>
__global__ __launch_bounds__(512 / UB) __attribute__((amdgpu_waves_per_eu(2 / UB, 2 / UB)))
void topk_prefilter_bf16_kernel(PreArgs a) {
    constexpr int THREADS = 512 / UB;
    constexpr int NACC = 2 * UB;               // 32x32 accumulator blocks per wavefront: [item block ib][query block ub]
    constexpr int S = D / 16;                  // k-steps per part
    constexpr int CH = D / 4;                  // 16-byte chunks per row
    constexpr int ROWB = CH * 16;
    constexpr int PANELB = 64 * ROWB;
    constexpr int PIECES = PANELB / 16 / THREADS;  // chunks a thread moves per panel
    constexpr int NS = 3 * NACC * S;           // MFMAs per panel and wavefront
    constexpr int NG = 4 * NACC;               // vote groups per panel and wavefront
    extern __shared__ __align__(16) unsigned char pre_ring[];   // 3 panels
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int bx = blockIdx.x & 7, bj = blockIdx.x >> 3;
    const int strip = bj % a.strips;
    const int64_t slice = (int64_t)(bj / a.strips) * 8 + bx;
    const int64_t p0 = slice * a.panels_per_slice;
    const int64_t p1 = min(a.panels, p0 + a.panels_per_slice);
    const int64_t u0 = (int64_t)strip * 256 + wave * (32 * UB);   // this wavefront's 32 UB queries
    int* my_cnt_out = STORE ? nullptr : a.pre_cnt + ((u0 + r) * a.n_slices + slice) * 2 + h;   // [query][slice][half]
    if (p0 >= p1) {  // block-uniform: an empty slice still owns its counters
        if (!STORE) {
#pragma unroll
            for (int ub = 0; ub < UB; ++ub) my_cnt_out[64 * ub * (int64_t)a.n_slices] = 0;
        }
        return;
    }
    // queries: fragments of the B operand, lane (r, h) holds k = 8 h .. 8 h + 7 of k-step s for query 32 ub + r
    bf16x8 uh[UB][S], ul[UB][S];
#pragma unroll
    for (int ub = 0; ub < UB; ++ub) {
        const uint4* up = a.Ub + (u0 + 32 * ub + r) * CH + h;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            uh[ub][s] = __builtin_bit_cast(bf16x8, up[2 * s]);
            ul[ub][s] = __builtin_bit_cast(bf16x8, up[CH / 2 + 2 * s]);
            if (MI_PRE_Q_IN_AGPR) {   // see the note at the macro
                asm volatile("" : "+a"(uh[ub][s]));
                asm volatile("" : "+a"(ul[ub][s]));
            }
        }
    }
    float tq[UB];
    unsigned long long* region[UB];
    int cnt[UB];
#pragma unroll
    for (int ub = 0; ub < UB; ++ub) {
        const int64_t q = u0 + 32 * ub + r;   // (padding queries have rows of their own: they pass only on NaN scores)
        tq[ub] = STORE ? 0.f : a.thrf[q];
        region[ub] = STORE ? nullptr : a.pre + q * kPreCap + slice * a.cap_s + h * (a.cap_s / 2);
        cnt[ub] = 0;
    }
    // item panels: a panel is one contiguous block of the split table; chunk (row, c) lives at row * ROWB + ((c ^ (row & 15)) << 4)
    u32x4 g[PIECES];
    uint32_t woff[PIECES];
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
        const int idx = tid + THREADS * j, row = idx / CH, c = idx % CH;
        woff[j] = (uint32_t)(row * ROWB + ((c ^ (row & 15)) << 4));
    }
    pre_issue<PIECES, THREADS>(g, a.Ib + p0 * (64 * CH) + tid);
    pre_commit<PIECES>(g, pre_ring, woff);
    if (p0 + 1 < p1) pre_issue<PIECES, THREADS>(g, a.Ib + (p0 + 1) * (64 * CH) + tid);
    // read offsets: item block ib, row 32 ib + r, chunk (part * CH / 2 + 2 s + h) ^ (r & 15) — one xor with a constant per read
    const uint32_t rx[2] = {(uint32_t)(r * ROWB) ^ (uint32_t)((h ^ (r & 15)) << 4),
                            (uint32_t)((32 + r) * ROWB) ^ (uint32_t)((h ^ (r & 15)) << 4)};
    f32x16 acc0[NACC], acc1[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc0[i][e] = 0.f; acc1[i][e] = 0.f; }
    float tqv[UB];   // no previous panel yet: nothing passes
#pragma unroll
    for (int ub = 0; ub < UB; ++ub) tqv[ub] = INFINITY;
    const int cmask = a.cap_s / 2 - 1;
    uint32_t item_prev = 0;
    const uint32_t row4 = 4u * (uint32_t)h;
    uint32_t cur = 0, nxt = PANELB;
    __syncthreads();

    // fragments of the A operand: [buffer][2 ib + part], read one k-step ahead of their MFMAs
    bf16x8 fr[2][4];
#define MI_PRE_READ(dst, buf, s)                                                                                        \
    {                                                                                                                   \
        _Pragma("unroll") for (int ib_ = 0; ib_ < 2; ++ib_) {                                                           \
            dst[2 * ib_] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(pre_ring + (buf) + (rx[ib_] ^ (uint32_t)((2 * (s)) << 4)))); \
            dst[2 * ib_ + 1] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(pre_ring + (buf) + (rx[ib_] ^ (uint32_t)((CH / 2 + 2 * (s)) << 4)))); \
        }                                                                                                               \
    }
    MI_PRE_READ(fr[0], 0u, 0)
    // Votes on the previous panel, FOUR accumulator registers at a time (rows 8 j' .. 8 j' + 3 of one 32x32 block: group g
    // = block g >> 2, j' = g & 3): one wave-uniform branch per group instead of one per register — the chain compare ->
    // scalar and -> scalar compare -> branch is latency a lone wavefront per SIMD cannot hide (one branch per register:
    // 304 us per 4 096-query chunk with empty hit bodies, against 125 us of MFMAs).  The group's "any" is formed on the
    // LANES' booleans (v_cndmask / v_or / one v_cmp_ne: 128 more vector instructions per panel than or-ing the four
    // compare masks with s_or_b64) on purpose: the scalar form was measured 307 -> 356 us per chunk at k = 12 — vector
    // instructions pipeline behind the MFMAs, the dependent scalar chain does not.
#define MI_PRE_VOTE4(prev, g)                                                                                           \
    {                                                                                                                   \
        const int ai_ = (g) >> 2, rb_ = 4 * ((g) & 3);                                                                  \
        const float v0_ = prev[ai_][rb_], v1_ = prev[ai_][rb_ + 1], v2_ = prev[ai_][rb_ + 2], v3_ = prev[ai_][rb_ + 3]; \
        const float t_ = tqv[ai_ % UB]; \
        /* the group's test on the LARGEST of the four (v_max3 + v_max + one compare).  A NaN score would slip through  \
           fmaxf, but NaN / inf anywhere in the tables make eps and with it thrf NaN or -inf for every query (the norms),  \
           and then this test passes everything: those calls end on the exact path regardless */                       \
        const float mx_ = MI_PRE_MAX_TEST ? fmaxf(fmaxf(v0_, v1_), fmaxf(v2_, v3_)) : 0.f;                              \
        const bool pq0_ = !(v0_ < t_), pq1_ = !(v1_ < t_), pq2_ = !(v2_ < t_), pq3_ = !(v3_ < t_);                      \
        const bool any_ = MI_PRE_MAX_TEST ? !(mx_ < t_) : (pq0_ | pq1_ | pq2_ | pq3_);                                  \
        if (MI_PRE_PROBE == 2) cnt[ai_ % UB] += (int)pq0_ + (int)pq1_ + (int)pq2_ + (int)pq3_;                           \
        else if (__builtin_expect(__ballot(any_) != 0ull, 0)) {   /* out of line */                                     \
            const bool p0_ = pq0_, p1_ = pq1_, p2_ = pq2_, p3_ = pq3_;                                                  \
            if (MI_PRE_PROBE == 1) cnt[ai_ % UB] += 1;                                                                   \
            else {                                                                                                      \
                const uint32_t it_ = item_prev + (uint32_t)((ai_ / UB) * 32 + 2 * rb_) + row4; \
                const bool two_ = (p0_ & (p1_ | p2_ | p3_)) | (p1_ & (p2_ | p3_)) | (p2_ & p3_);                        \
                if (MI_PRE_ONE_PUT && __builtin_expect(__ballot(two_) == 0ull, 1)) {                                    \
                    /* no lane has two of the four (the usual case): ONE predicated store for the group */              \
                    const float v_ = p0_ ? v0_ : (p1_ ? v1_ : (p2_ ? v2_ : v3_));                                       \
                    const uint32_t o_ = p0_ ? 0u : (p1_ ? 1u : (p2_ ? 2u : 3u));                                        \
                    pre_put(p0_ | p1_ | p2_ | p3_, v_, it_ + o_, cnt[ai_ % UB], region[ai_ % UB], cmask);                 \
                } else {                                                                                                \
                    pre_put(p0_, v0_, it_, cnt[ai_ % UB], region[ai_ % UB], cmask);                                       \
                    pre_put(p1_, v1_, it_ + 1, cnt[ai_ % UB], region[ai_ % UB], cmask);                                   \
                    pre_put(p2_, v2_, it_ + 2, cnt[ai_ % UB], region[ai_ % UB], cmask);                                   \
                    pre_put(p3_, v3_, it_ + 3, cnt[ai_ % UB], region[ai_ % UB], cmask);                                   \
                }                                                                                                       \
            }                                                                                                           \
        }                                                                                                               \
    }
#define MI_PRE_PANEL(acc, prev, p)                                                                                      \
    {                                                                                                                   \
        if ((p) + 1 < p1) pre_commit<PIECES>(g, pre_ring + nxt, woff);   /* panel p + 1: loaded one panel ago */ \
        if ((p) + 2 < p1) pre_issue<PIECES, THREADS>(g, a.Ib + ((p) + 2) * (64 * CH) + tid); \
        _Pragma("unroll") for (int i = 0; i < NACC; ++i) \
            _Pragma("unroll") for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;                                             \
        _Pragma("unroll") for (int s = 0; s < S; ++s) {                                                                 \
            if (s + 1 < S) MI_PRE_READ(fr[(s + 1) & 1], cur, s + 1)   /* a k-step ahead: the votes' branches pin it here */ \
            _Pragma("unroll") for (int term = 0; term < 3; ++term)                                                      \
                _Pragma("unroll") for (int ai = 0; ai < NACC; ++ai) { \
                    const int ib = ai / UB, ub = ai % UB; \
                    acc[ai] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[s & 1][2 * ib + (term == 2 ? 1 : 0)],         \
                                                                      term == 1 ? ul[ub][s] : uh[ub][s], acc[ai], 0, 0, 0); \
                    const int m = s * (3 * NACC) + term * NACC + ai; \
                    if (!STORE && (m + 1) % (NS / NG) == 0) MI_PRE_VOTE4(prev, m / (NS / NG)) \
                }                                                                                                       \
        }                                                                                                               \
        if (STORE) {   /* out[a_row][b_row]: for one register the 32 lanes of a half write 128 consecutive bytes */      \
            _Pragma("unroll") for (int ai = 0; ai < NACC; ++ai) \
                _Pragma("unroll") for (int reg = 0; reg < 16; ++reg) {                                                  \
                    const int64_t ar_ = (p) * 64 + (ai / UB) * 32 + (reg & 3) + 8 * (reg >> 2) + (int64_t)row4; \
                    const int64_t br_ = u0 + 32 * (ai % UB) + r; \
                    if (ar_ < a.n_a && br_ < a.n_b) a.out[ar_ * a.ldo + br_] = acc[ai][reg];                            \
                }                                                                                                       \
        }                                                                                                               \
        item_prev = (uint32_t)((p) * 64);                                                                               \
        _Pragma("unroll") for (int ub = 0; ub < UB; ++ub) tqv[ub] = tq[ub]; \
        cur = nxt;                                                                                                      \
        nxt = nxt + PANELB == 3 * PANELB ? 0u : nxt + PANELB;                                                           \
        __syncthreads();  /* panel p + 1 is in LDS and everybody is done with panel p */                                \
        if ((p) + 1 < p1) MI_PRE_READ(fr[0], cur, 0)                                                                    \
    }

    int64_t p = p0;
    for (; p + 1 < p1; p += 2) {
        MI_PRE_PANEL(acc0, acc1, p)
        MI_PRE_PANEL(acc1, acc0, p + 1)
    }
    bool last_in_acc0 = false;
    if (p < p1) {
        MI_PRE_PANEL(acc0, acc1, p)
        last_in_acc0 = true;
    }
    if (STORE) return;
    // the last panel's votes
    if (last_in_acc0) {
#pragma unroll
        for (int g = 0; g < NG; ++g) MI_PRE_VOTE4(acc0, g)
    } else {
#pragma unroll
        for (int g = 0; g < NG; ++g) MI_PRE_VOTE4(acc1, g)
    }
#undef MI_PRE_PANEL
#undef MI_PRE_READ
#undef MI_PRE_VOTE4
#pragma unroll
    for (int ub = 0; ub < UB; ++ub) my_cnt_out[64 * ub * (int64_t)a.n_slices] = cnt[ub];
}

// The kk-th largest score key among the n composites in sh.cand (n >= kk): 4-pass radix select in LDS.
__device__ __forceinline__ uint32_t kth_key_of_cands(SelectShared& sh, int n, int kk) {
    const int tid = threadIdx.x;
    if (tid == 0) { sh.prefix = 0u; sh.need = kk; }
    __syncthreads();
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        const uint32_t hi_mask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
        sh.hist[tid] = 0;
        __syncthreads();
        const uint32_t prefix = sh.prefix;
        for (int i = tid; i < n; i += kBlock) {
            const uint32_t key = (uint32_t)(sh.cand[i] >> 32);
            if ((key & hi_mask) == prefix) atomicAdd(&sh.hist[(key >> shift) & 255u], 1);
        }
        __syncthreads();
        radix_pick_bin(sh, prefix, shift);
    }
    return sh.prefix;
}

// A LOWER BOUND of the kk-th largest score key among the n composites in sh.cand (n >= kk) that at least kk of them reach —
// all the refine step needs (round 4).  The 4-pass radix select above cost 55 of the refine kernel's ~200 us per 4 096-query
// chunk at k = 256 (probe builds, profiles/r04_topk_refine.md): the keys of one row's list share sign, exponent and the top
// mantissa bits, so the first passes put all ~750 entries on two or three LDS counters.  Here ONE histogram over the keys'
// own range [min, max], 256 equal bins (the entries spread over all of them), the highest bin b* whose suffix count reaches
// kk, and the smallest key in it: at least kk keys are >= that key, and it is at most a bin's width (~3 entries) below the
// exact kk-th largest — a handful of extra survivors for the exact rescoring, a quarter of the barriers, no contention.
__device__ __forceinline__ uint32_t kth_key_bound_of_cands(SelectShared& sh, int n, int kk) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    for (int i = tid; i < n; i += kBlock) {
        const uint32_t key = (uint32_t)(sh.cand[i] >> 32);
        lo = min(lo, key);
        hi = max(hi, key);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        lo = min(lo, (uint32_t)__shfl_xor((int)lo, d, 64));
        hi = max(hi, (uint32_t)__shfl_xor((int)hi, d, 64));
    }
    uint32_t* mm = reinterpret_cast<uint32_t*>(sh.scan_sh);   // kBlock / 64 + 1 = 5 words: min per wave, then reused
    __shared__ uint32_t s_hi[kBlock / 64];
    if (lane == 0) { mm[wave] = lo; s_hi[wave] = hi; }
    sh.hist[tid] = 0;
    if (tid == 0) sh.need = kk;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) { lo = min(lo, mm[w]); hi = max(hi, s_hi[w]); }
    if (hi == lo) return lo;                                   // block-uniform: every key equal
    const float scale = 256.0f / ((float)(hi - lo) + 1.0f);
    auto bin_of = [&](uint32_t key) { return min(255, (int)((float)(key - lo) * scale)); };   // monotone in key
    __syncthreads();                                           // mm / s_hi read by everybody before scan_sh is reused
    for (int i = tid; i < n; i += kBlock) atomicAdd(&sh.hist[bin_of((uint32_t)(sh.cand[i] >> 32))], 1);
    __syncthreads();
    radix_pick_bin(sh, 0u, 0);                                 // sh.prefix = b*: the highest bin with >= kk keys at or above it
    const int b = (int)sh.prefix;
    __syncthreads();
    if (tid == 0) sh.prefix = 0xFFFFFFFFu;
    __syncthreads();
    for (int i = tid; i < n; i += kBlock) {
        const uint32_t key = (uint32_t)(sh.cand[i] >> 32);
        if (bin_of(key) == b) atomicMin(&sh.prefix, key);
    }
    __syncthreads();
    return sh.prefix;
}

// One block per query: the slice regions of its list -> LDS (exclusions and table padding dropped here), the survivor
// bound of step 2, exact scores of the survivors, finish_candidates().  Rows that cannot be decided this way (a region
// or the list overflowed, fewer than kk candidates, the bound dips below the list's own) are recomputed exactly, as on
// the f32 path.
__global__ __launch_bounds__(kBlock) void topk_refine_finalize_kernel(FusedArgs a, PreArgs pa, const float* __restrict__ epsv,
                                                                      int k, int kpow2, float* __restrict__ scores,
                                                                      int64_t* __restrict__ out_idx,
                                                                      float* __restrict__ out_score) {
    __shared__ SelectShared sh;
    __shared__ __align__(16) float urow[FKC];
    __shared__ int reg_off[2 * kBlock];
    // the survivors' ids live in the last quarter of sh.cand (lists longer than kPreList take the exact path): 34 KB of LDS
    // per workgroup = four of them per CU instead of three
    uint32_t* surv = reinterpret_cast<uint32_t*>(sh.cand + kPreList);
    __shared__ int bad;
    const int64_t q = blockIdx.x;
    if (q >= a.n_q) return;
    const int tid = threadIdx.x;
    const int kk = (int)min((int64_t)k, a.n_items);
    const float* u = a.U + a.uid[q] * a.ldu;
    if (tid == 0) { sh.count = 0; bad = 0; }
    for (int c = tid; c < a.d; c += kBlock) urow[c] = u[c];
    __syncthreads();
    {   // gather.  The regions' fills first (one coalesced load) and their prefix sums, then the entries by FLAT index, four
        // per thread and trip with the loads of a trip issued together: a thread walking "its" region entry by entry paid
        // three dependent trips to memory (fill, entry, exclusion word) per entry
        const int n_reg = 2 * pa.n_slices, cap = pa.cap_s / 2;   // regions: [slice][half-wavefront], <= 512
        const int* cq = pa.pre_cnt + q * n_reg;
        const unsigned long long* pq = pa.pre + q * kPreCap;
        const int g0 = 2 * tid, g1 = 2 * tid + 1;
        int c0 = g0 < n_reg ? cq[g0] : 0, c1 = g1 < n_reg ? cq[g1] : 0;
        if (c0 > cap || c1 > cap) bad = 1;               // benign race: every writer stores 1
        c0 = min(c0, cap);
        c1 = min(c1, cap);
        int total;
        const int ex = block_excl_scan(c0 + c1, sh.scan_sh, &total);
        reg_off[g0] = ex;                                // reg_off[g] = entries in front of region g
        reg_off[g1] = ex + c0;
        __syncthreads();
        for (int base = 0; base < total; base += 4 * kBlock) {
            unsigned long long e[4];
            uint32_t w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = base + tid + j * kBlock;
                e[j] = ~0ull;
                if (i < total) {
                    int lo = 0, hi = n_reg;              // the last region whose offset is <= i (empty regions in between are skipped)
                    while (hi - lo > 1) {
                        const int mid = (lo + hi) >> 1;
                        if (reg_off[mid] <= i) lo = mid; else hi = mid;
                    }
                    e[j] = pq[(int64_t)lo * cap + (i - reg_off[lo])];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t item = (uint32_t)(e[j] >> 32);
                w[j] = item < (uint32_t)a.n_items ? a.bitmap[q * a.words + (item >> 5)] : ~0u;   // table padding (and the unused slots): dropped
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t item = (uint32_t)(e[j] >> 32);
                if ((w[j] >> (item & 31)) & 1u) continue;                                        // excluded
                const int slot = atomicAdd(&sh.count, 1);
                if (slot < kPreList) sh.cand[slot] = composite(score_key(__uint_as_float((uint32_t)e[j])), item);
            }
        }
    }
    __syncthreads();
#if defined(MI_REFINE_PROBE) && MI_REFINE_PROBE == 1   // timing probes (tools/pre_probe.sh): wrong results
    return;
#endif
    const int cnt = sh.count;
    bool ok = !bad && cnt >= kk && cnt <= kPreList;   // block-uniform
    if (ok) {
        // t2: a lower bound of the kk-th largest approximate key
        uint32_t t2;
        if (kk <= 64) {   // the kk-th largest of the first S entries (cf. finish_candidates)
            const int S = min(cnt, kk <= 16 ? 64 : (kk <= 32 ? 128 : 256));
            uint32_t* keys = reinterpret_cast<uint32_t*>(sh.hist);
            const uint32_t x = tid < S ? (uint32_t)(sh.cand[tid] >> 32) : 0u;
            keys[tid] = x;
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
                if (gt <= kk - 1 && kk - 1 < ge) sh.prefix = x;
            }
            __syncthreads();
            t2 = sh.prefix;
        } else {
#if defined(MI_REFINE_RADIX_T2)
            t2 = kth_key_of_cands(sh, cnt, kk);          // A/B: the exact kk-th largest by 4-pass radix select
#else
            t2 = kth_key_bound_of_cands(sh, cnt, kk);
#endif
        }
#if defined(MI_REFINE_PROBE) && MI_REFINE_PROBE == 2
        if (t2 != 0x12345u) return;
#endif
        const float eps = epsv[q];
        const float t_l = key_score(t2) - 2.f * eps;
        ok = t_l >= pa.thrf[q];                      // false on NaN as well
        if (ok) {
            if (tid == 0) sh.need = 0;
            __syncthreads();
            for (int i = tid; i < cnt; i += kBlock) {
                const unsigned long long c = sh.cand[i];
                if (!(key_score((uint32_t)(c >> 32)) < t_l)) {
                    const int slot = atomicAdd(&sh.need, 1);
                    if (slot < kPreSurv) surv[slot] = 0xFFFFFFFFu - (uint32_t)c;
                }
            }
            __syncthreads();
            const int n2 = sh.need;                  // >= kk
#if defined(MI_REFINE_PROBE) && MI_REFINE_PROBE == 3
            if (n2 >= 0) return;
#endif
            ok = n2 <= kPreSurv;
#ifndef MI_REFINE_SLABS
#define MI_REFINE_SLABS 0   // 1: the exact rescoring stages 64-byte row slabs through LDS, four lanes per row (A/B).  Measured at
                            // k = 256, round 4 (tools/pre_probe.sh): refine kernel 186 us per 4 096-query chunk without, 192 with —
                            // a thread reading ITS row issues all 32 sixteen-byte loads at once, and that depth of loads in
                            // flight is worth more than the 4 x fewer L2 requests of the coalesced form (8 dependent slab steps)
#endif
            if (ok && MI_REFINE_SLABS && n2 <= 512 && a.d % 16 == 0) {
                // exact scores, coalesced (A/B form, off: see MI_REFINE_SLABS): a wavefront stages its 64 rows' 64-byte slab (16 floats)
                // with FOUR lanes per row — one 64-byte request per row and slab — into LDS (the candidate list's region is
                // free by now; rows 80 bytes apart: conflict-free 16-byte reads), then every lane runs its row's 16 fma steps
                // from LDS; the next slab's loads are in flight meanwhile.  Same k-ordered fma chain, same bits.
                float* stage = reinterpret_cast<float*>(sh.cand) + 1024 + (tid >> 6) * (64 * 20);   // behind 512 result composites
                const int lane = tid & 63, sub = lane >> 2, part = lane & 3;
                const int n_slab = a.d / 16;
                for (int base = 0; base < n2; base += kBlock) {
                    const int my = base + tid;                       // the survivor this lane scores
                    const float* rowp[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {                    // the rows this lane helps to load: sub + 16 j of the wavefront's 64
                        const int r = base + (tid & ~63) + sub + 16 * j;
                        rowp[j] = r < n2 ? a.I + (int64_t)surv[r] * a.ldi + 4 * part : nullptr;
                    }
                    float4 nxt[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) nxt[j] = rowp[j] ? *reinterpret_cast<const float4*>(rowp[j]) : mi_f4_zero();
                    float acc = 0.f;
                    for (int sl = 0; sl < n_slab; ++sl) {
                        __builtin_amdgcn_wave_barrier();             // the previous slab's reads are done (same wavefront: program order)
#pragma unroll
                        for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(stage + (sub + 16 * j) * 20 + 4 * part) = nxt[j];
                        if (sl + 1 < n_slab) {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                nxt[j] = rowp[j] ? *reinterpret_cast<const float4*>(rowp[j] + 16 * (sl + 1)) : mi_f4_zero();
                        }
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int c4 = 0; c4 < 4; ++c4) {
                            const float4 w = *reinterpret_cast<const float4*>(stage + lane * 20 + 4 * c4);
                            const float4 uu = *reinterpret_cast<const float4*>(urow + 16 * sl + 4 * c4);
                            acc = fmaf(uu.x, w.x, acc);
                            acc = fmaf(uu.y, w.y, acc);
                            acc = fmaf(uu.z, w.z, acc);
                            acc = fmaf(uu.w, w.w, acc);
                        }
                    }
                    if (my < n2) sh.cand[my] = composite(score_key(acc), surv[my]);
                }
                __syncthreads();
#if defined(MI_REFINE_PROBE) && MI_REFINE_PROBE == 4
                if (n2 >= 0) { if (tid == 0) out_idx[q * k] = (int64_t)sh.cand[0]; return; }
#endif
                finish_candidates(n2, k, kk, kpow2, q, out_idx, out_score, sh);
                return;
            }
            if (ok) {
                // exact scores: the k-ordered fma chain, bit for bit the f32 MFMA's and the oracle's.  TWO rows per thread and
                // trip (i and i + kBlock), their loads in flight together: at k = 256 the survivors are kk + a few (~270-280), and
                // a second trip for the two dozen beyond kBlock paid a full round of memory latency with one wavefront a third full
                for (int base = 0; base < n2; base += 2 * kBlock) {
                    const int i0 = base + tid, i1 = i0 + kBlock;
                    const bool h0 = i0 < n2, h1 = i1 < n2;
                    const uint32_t item0 = h0 ? surv[i0] : 0u, item1 = h1 ? surv[i1] : 0u;
                    const float4* it0 = reinterpret_cast<const float4*>(a.I + (int64_t)item0 * a.ldi);
                    const float4* it1 = reinterpret_cast<const float4*>(a.I + (int64_t)item1 * a.ldi);
                    float acc0 = 0.f, acc1 = 0.f;
                    if (__ballot(h1) == 0ull) {   // wavefront-uniform: nothing in the second half
                        if (h0)
                            for (int c4 = 0; c4 < a.d / 4; ++c4) {
                                const float4 w = it0[c4];
                                const float4 uu = *reinterpret_cast<const float4*>(urow + 4 * c4);
                                acc0 = fmaf(uu.x, w.x, acc0);
                                acc0 = fmaf(uu.y, w.y, acc0);
                                acc0 = fmaf(uu.z, w.z, acc0);
                                acc0 = fmaf(uu.w, w.w, acc0);
                            }
                    } else {
                        for (int c4 = 0; c4 < a.d / 4; ++c4) {
                            const float4 w0 = h0 ? it0[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
                            const float4 w1 = h1 ? it1[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
                            const float4 uu = *reinterpret_cast<const float4*>(urow + 4 * c4);
                            acc0 = fmaf(uu.x, w0.x, acc0);
                            acc0 = fmaf(uu.y, w0.y, acc0);
                            acc0 = fmaf(uu.z, w0.z, acc0);
                            acc0 = fmaf(uu.w, w0.w, acc0);
                            acc1 = fmaf(uu.x, w1.x, acc1);
                            acc1 = fmaf(uu.y, w1.y, acc1);
                            acc1 = fmaf(uu.z, w1.z, acc1);
                            acc1 = fmaf(uu.w, w1.w, acc1);
                        }
                    }
                    if (h0) sh.cand[i0] = composite(score_key(acc0), item0);
                    if (h1) sh.cand[i1] = composite(score_key(acc1), item1);
                }
                __syncthreads();
#if defined(MI_REFINE_PROBE) && MI_REFINE_PROBE == 4
                if (n2 >= 0) { if (tid == 0) out_idx[q * k] = (int64_t)sh.cand[0]; return; }
#endif
                finish_candidates(n2, k, kk, kpow2, q, out_idx, out_score, sh);
                return;
            }
        }
    }
    __syncthreads();
    // rare: materialise this row's scores and select exactly (as topk_finalize_kernel)
    float* row = scores + q * a.n_items;
    for (int64_t i = tid; i < a.n_items; i += kBlock) {
        const float* it = a.I + i * a.ldi;
        float acc = 0.f;
        for (int c = 0; c < a.d; ++c) acc = fmaf(urow[c], it[c], acc);
        if ((a.bitmap[q * a.words + (i >> 5)] >> (i & 31)) & 1u) acc = -INFINITY;
        row[i] = acc;
    }
    __syncthreads();
    select_row_exact(row, a.n_items, k, kk, kpow2, q, out_idx, out_score, sh);
}

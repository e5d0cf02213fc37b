// K6: dense fp32 GEMM on the gfx950 f32-input MFMA (v_mfma_f32_32x32x2_f32).
//
//   C[m,n] = act( sum_k A(m,k) * B(k,n)  + bias[n]  + (accumulate ? C[m,n] : 0) )
//
// replaces torch.nn.Linear in SAGEConv's lin_l / lin_r and in the decoder MLP
// (model/layers.py:11-56, model/encoder_decoder.py:55-72) with their backward products, and
// computes the score matrix of the top-K path (utils/metrics_lightgcn.py:137).
//
// Why the f32 MFMA: the reference is fp32 end to end and parity is 1e-4, so bf16 is out; gfx950 has
// no xf32.  v_mfma_f32_32x32x2_f32 is exact fp32 and — per the CDNA4 guide — bit for bit a
// k-ordered fma chain, D = fma(a_k1, b_k1, fma(a_k0, b_k0, C)).  Walking K in ascending order with
// one accumulator therefore gives every output element the sequential chain
//   acc = 0; for k in 0..K-1: acc = fmaf(A[m,k], B[k,n], acc)
// which oracle/spmm_ref.c restates, so GEMM results (and the top-K ordering built on them) are
// checked bitwise, not within a tolerance.
//
// Tiling: 256 threads = 4 wavefronts, block tile 64x64, each wavefront one 32x32 accumulator
// (16 VGPRs), K walked in steps of 32 through LDS.  The f32 MFMA issues once per 64 cycles per
// SIMD, so two ds_read_b32 per MFMA keep it fed; tiles are padded to 33 floats per row so both the
// row-major and the transposed store patterns are bank-conflict free.
#include "gemm.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 64, BN = 64, BK = 32, PAD = 33;

__global__ __launch_bounds__(256) void gemm_f32_kernel(MiGemmArgs g) {
    __shared__ float As[BM][PAD];
    __shared__ float Bs[BN][PAD];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
    const bool a_kfast = (g.sa_k == 1), b_kfast = (g.sb_k == 1);
    const int64_t k_lo = (int64_t)blockIdx.z * g.k_per_split;
    const int64_t k_hi = min(g.K, k_lo + g.k_per_split);

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    for (int64_t k0 = k_lo; k0 < k_hi; k0 += BK) {
#pragma unroll
        for (int j = 0; j < (BM * BK) / 256; ++j) {
            const int idx = tid + 256 * j;
            const int m = a_kfast ? idx / BK : idx % BM;
            const int k = a_kfast ? idx % BK : idx / BM;
            const int64_t gm = m0 + m, gk = k0 + k;
            float v = 0.f;
            if (gm < g.M && gk < k_hi) {
                const int64_t r = g.a_rows ? g.a_rows[gm] : gm;
                v = g.A[r * g.sa_m + gk * g.sa_k];
            }
            As[m][k] = v;
        }
#pragma unroll
        for (int j = 0; j < (BN * BK) / 256; ++j) {
            const int idx = tid + 256 * j;
            const int n = b_kfast ? idx / BK : idx % BN;
            const int k = b_kfast ? idx % BK : idx / BN;
            const int64_t gn = n0 + n, gk = k0 + k;
            Bs[n][k] = (gn < g.N && gk < k_hi) ? g.B[gn * g.sb_n + gk * g.sb_k] : 0.f;
        }
        __syncthreads();
        const float* ap = &As[wm * 32 + (lane & 31)][lane >> 5];
        const float* bp = &Bs[wn * 32 + (lane & 31)][lane >> 5];
#pragma unroll
        for (int s = 0; s < BK / 2; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * s], bp[2 * s], acc, 0, 0, 0);
        __syncthreads();
    }

    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int64_t gn = n0 + wn * 32 + (lane & 31);
    if (gn >= g.N) return;
    if (g.splits > 1) {  // raw slice accumulator; bias / accumulate / act happen in the reduce
        float* slab = g.partial + (int64_t)blockIdx.z * g.M * g.N;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int64_t gm = m0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            if (gm < g.M) slab[gm * g.N + gn] = acc[reg];
        }
        return;
    }
    const float bv = g.bias ? g.bias[gn] : 0.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int64_t gm = m0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        if (gm >= g.M) continue;
        float v = acc[reg];
        if (g.bias) v += bv;
        float* c = g.C + gm * g.ldc + gn;
        if (g.accumulate) v += *c;
        if (g.act == 1) v = v > 0.f ? v : 0.f;
        *c = v;
    }
}

// Fast path (operands 16-byte aligned, leading dimensions and K multiples of 4).  One global-load phase
// stages a 64 x KC panel of each operand (KC = 64: every lane keeps 8 independent 16-byte loads in
// flight, four workgroups per CU), one barrier, then KC/2 MFMAs back to back per wavefront; the accumulator walks K in ascending
// order exactly as in the generic kernel, so results are bitwise the same.  Either operand may be
// K-contiguous (x @ W^T: nn.Linear forward, the top-K score block) or contiguous along its output
// dimension (dX = dY @ W, dW = dY^T @ X): the float4 is loaded along whichever dimension is contiguous
// and written to the same [row][k] LDS panel.  Rows are padded to KC + 1 floats: conflict-free fragment
// reads (bank = (row + k) mod 32).
#ifndef MI_GEMM_MFMA_UNROLL
#define MI_GEMM_MFMA_UNROLL 16  // 64: +1 % (tools/bench_gemm.py: 2 621 x 100 K x 128 53.0 -> 53.3, 4 096^3 72.4 -> 72.9 TF/s)
#endif
#ifndef MI_GEMM_KC
#define MI_GEMM_KC 64   // K per staged panel.  Round 3 A/B (tools/bench_gemm.py, tools/iter_timeline.py): 128 -> 64 halves the LDS per
#endif                  // workgroup (66 -> 33 KB), so four workgroups share a CU instead of two and one's loads hide behind the
#ifndef MI_GEMM_WGS     // others' MFMAs: 4096^3 72.8 -> 102.9 TF/s (0.65 of the f32 MFMA peak), 169 000 x 128 x 84 34.8 -> 50.7,
#define MI_GEMM_WGS 4   // the top-K score block 53.3 -> 67.0; the ranker's forward / dX grouped launches 51 -> 38, 32 -> 28, 46 -> 33 us
#endif                  // (same k-ascending chain per output element: results are bitwise what KC = 128 gave)
constexpr int KC = MI_GEMM_KC, KPAD = KC + 1;

// A 64 x KC panel of an operand whose element (row, k) sits at base[row * s_row + k * s_k]; rows >= n_rows
// and k >= kw read as zero.  Split in two so that both operands' global loads are in flight before the
// first LDS store: panel_issue loads into registers, panel_commit writes panel[r][k].
constexpr int kN4 = (64 * KC / 4) / 256;  // float4 per thread and operand

template <bool KFAST>
__device__ __forceinline__ void panel_issue(float4 (&v)[kN4], const float* __restrict__ base, int64_t s_row,
                                            int64_t s_k, const int64_t* __restrict__ row_map, int64_t row0,
                                            int64_t n_rows, int64_t kc, int kw, int tid) {
#pragma unroll
    for (int j = 0; j < kN4; ++j) {
        const int idx = tid + 256 * j;
        v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (KFAST) {  // float4 along k
            const int r = idx / (KC / 4), c4 = idx % (KC / 4);
            const int64_t gr = row0 + r;
            if (gr < n_rows && c4 * 4 < kw) {
                const int64_t row = row_map ? row_map[gr] : gr;
                v[j] = *reinterpret_cast<const float4*>(base + row * s_row + kc + c4 * 4);
            }
        } else {      // contiguous along the row index: one float4 = 4 consecutive rows at one k
            const int k = idx / 16, r4 = (idx % 16) * 4;
            const int64_t gr = row0 + r4;
            if (k < kw) {
                const float* src = base + (kc + k) * s_k + gr;
                if (gr + 3 < n_rows) {
                    v[j] = *reinterpret_cast<const float4*>(src);
                } else {
                    if (gr < n_rows) v[j].x = src[0];
                    if (gr + 1 < n_rows) v[j].y = src[1];
                    if (gr + 2 < n_rows) v[j].z = src[2];
                }
            }
        }
    }
}

template <bool KFAST>
__device__ __forceinline__ void panel_commit(float (*panel)[KPAD], const float4 (&v)[kN4], int tid) {
#pragma unroll
    for (int j = 0; j < kN4; ++j) {
        const int idx = tid + 256 * j;
        if (KFAST) {
            const int r = idx / (KC / 4), c = (idx % (KC / 4)) * 4;
            panel[r][c] = v[j].x; panel[r][c + 1] = v[j].y; panel[r][c + 2] = v[j].z; panel[r][c + 3] = v[j].w;
        } else {
            const int k = idx / 16, r4 = (idx % 16) * 4;
            panel[r4][k] = v[j].x; panel[r4 + 1][k] = v[j].y; panel[r4 + 2][k] = v[j].z; panel[r4 + 3][k] = v[j].w;
        }
    }
}

template <bool A_KFAST, bool B_KFAST>
__global__ __launch_bounds__(256, MI_GEMM_WGS) void gemm_fast_kernel(MiGemmArgs g) {
    __shared__ float As[BM][KPAD];
    __shared__ float Bs[BN][KPAD];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
    const int64_t k_lo = (int64_t)blockIdx.z * g.k_per_split;
    const int64_t k_hi = min(g.K, k_lo + g.k_per_split);

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    // Software pipeline (round 3): the NEXT panel's global loads are issued before the MFMAs of the current one and land
    // in registers while the matrix cores work; they are committed to LDS after the barrier that ends the current
    // panel.  Before, a workgroup's loads and its MFMAs never overlapped (a phase of ~8 us held 0.85 us of MFMA).
    // The accumulator still walks K in ascending order: results are bitwise the same.
    float4 va[kN4], vb[kN4];
    if (k_lo < k_hi) {
        const int kw0 = (int)min((int64_t)KC, k_hi - k_lo);
        panel_issue<A_KFAST>(va, g.A, g.sa_m, g.sa_k, g.a_rows, m0, g.M, k_lo, kw0, tid);
        panel_issue<B_KFAST>(vb, g.B, g.sb_n, g.sb_k, nullptr, n0, g.N, k_lo, kw0, tid);
        panel_commit<A_KFAST>(As, va, tid);
        panel_commit<B_KFAST>(Bs, vb, tid);
    }
    __syncthreads();
    for (int64_t kc = k_lo; kc < k_hi; kc += KC) {
        const int kw = (int)min((int64_t)KC, k_hi - kc);  // multiple of 4
        const int64_t kn = kc + KC;
        const bool has_next = kn < k_hi;
        const int kwn = has_next ? (int)min((int64_t)KC, k_hi - kn) : 0;
        if (has_next) {
            panel_issue<A_KFAST>(va, g.A, g.sa_m, g.sa_k, g.a_rows, m0, g.M, kn, kwn, tid);
            panel_issue<B_KFAST>(vb, g.B, g.sb_n, g.sb_k, nullptr, n0, g.N, kn, kwn, tid);
        }
        const float* ap = &As[wm * 32 + (lane & 31)][lane >> 5];
        const float* bp = &Bs[wn * 32 + (lane & 31)][lane >> 5];
        if (kw == KC) {
#pragma unroll MI_GEMM_MFMA_UNROLL
            for (int s = 0; s < KC / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * s], bp[2 * s], acc, 0, 0, 0);
        } else {
            // the panel is zero beyond kw, so an odd tail pairs its last k with a zero
            for (int s = 0; s < (kw + 1) / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * s], bp[2 * s], acc, 0, 0, 0);
        }
        __syncthreads();                       // every wavefront is done with this panel's LDS image
        if (has_next) {
            panel_commit<A_KFAST>(As, va, tid);
            panel_commit<B_KFAST>(Bs, vb, tid);
            __syncthreads();
        }
    }

    const int64_t gn = n0 + wn * 32 + (lane & 31);
    if (gn >= g.N) return;
    if (g.splits > 1) {
        float* slab = g.partial + (int64_t)blockIdx.z * g.M * g.N;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int64_t gm = m0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            if (gm < g.M) slab[gm * g.N + gn] = acc[reg];
        }
        return;
    }
    const float bv = g.bias ? g.bias[gn] : 0.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int64_t gm = m0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        if (gm >= g.M) continue;
        float v = acc[reg];
        if (g.bias) v += bv;
        float* c = g.C + gm * g.ldc + gn;
        if (g.accumulate) v += *c;
        if (g.act == 1) v = v > 0.f ? v : 0.f;
        *c = v;
    }
}

// ---- grouped launch -------------------------------------------------------------------------------------------
// A ranker batch is ~10^4 nodes: its ~45 small products per iteration cost more in launches (host and device, 5-15 us
// each) than in arithmetic.  One launch of gemm_group_kernel runs up to kMaxGroup independent problems, each
//     C = act( sum over up to two (A, B) pairs of A_p @ B_p  + bias + (accumulate ? C : 0) ),   A optionally masked,
// e.g. all of a hetero layer's lin_l(agg) + lin_r(x_dst) products forward (two pairs per problem, no separate add),
// all its dX products backward, all its weight gradients (split-K, one grouped reduce).  `a_mask` (same layout as
// pair 0's A) zeroes A where mask <= 0: the relu backward dY * (out > 0) without its own pass.  Same tile body as
// gemm_fast_kernel (fast-path operands only), K ascending within a pair, pairs in order: one fma chain per output.
constexpr int kMaxGroup = 8;

struct MiGemmPairArgs {
    const float* A; int64_t sa_m, sa_k;
    const float* B; int64_t sb_n, sb_k;
    int64_t K;
};
struct MiGemmGroupProblem {
    int64_t M, N;
    MiGemmPairArgs pair[2];
    int n_pairs;
    const float* a_mask;
    const float* bias;
    float* C; int64_t ldc;
    int accumulate, act;
    int splits; int64_t k_per_split; float* partial;  // split-K of pair 0 (single-pair problems only)
    int tiles_n, tiles_mn;                            // tile grid of this problem
};
// A SHARE SET: consecutive split problems that read the same A operand over the same K with the same slicing (the weight
// gradients dW_l / db / dW_r of one relation: dY^T @ [agg | 1 | x_dst]).  Launched problem after problem, each K slice
// of dY (and of its relu mask) is fetched once per output tile of every member — 310 MB of L2-miss traffic for the
// 50 MB the ranker's layer-0 launch needs (PMC, round 3).  In a set the workgroups are ordered slice-major and a slice's
// G = sum of the members' tiles workgroups are all placed on ONE XCD (workgroup w runs on XCD w mod 8: local index
// w = ((z / 8) * G + g) * 8 + z mod 8), so they find the slice in that XCD's L2 after its first reader.
struct MiGemmShareSet {
    int first, count;        // members: problems first .. first + count - 1
    int G, S;                // workgroups per slice, slices
    int64_t start;           // first linear workgroup (a multiple of 8), set size = ceil(S / 8) * 8 * G
    int tile_off[kMaxGroup + 1];  // member j's tiles are g in [tile_off[j], tile_off[j + 1])
};
constexpr int kMaxSets = 4;
struct MiGemmGroupArgs {
    int n;
    int64_t block_start[kMaxGroup + 1];   // first linear workgroup of each problem
    int64_t out_start[kMaxGroup + 1];     // first linear output element of each split problem (reduce kernel)
    MiGemmGroupProblem p[kMaxGroup];
    int n_sets;
    int set_of[kMaxGroup];                // share set of each problem, -1 = none
    MiGemmShareSet set[kMaxSets];
};

template <bool KFAST>
__device__ __forceinline__ void panel_mask(float4 (&v)[kN4], const float* __restrict__ mask, int64_t s_row, int64_t s_k,
                                           int64_t row0, int64_t n_rows, int64_t kc, int kw, int tid) {
    float4 m[kN4];
    panel_issue<KFAST>(m, mask, s_row, s_k, nullptr, row0, n_rows, kc, kw, tid);
#pragma unroll
    for (int j = 0; j < kN4; ++j) {
        if (!(m[j].x > 0.f)) v[j].x = 0.f;
        if (!(m[j].y > 0.f)) v[j].y = 0.f;
        if (!(m[j].z > 0.f)) v[j].z = 0.f;
        if (!(m[j].w > 0.f)) v[j].w = 0.f;
    }
}

__global__ __launch_bounds__(256, MI_GEMM_WGS) void gemm_group_kernel(MiGemmGroupArgs ga) {
    __shared__ float As[BM][KPAD];
    __shared__ float Bs[BN][KPAD];
    int pi = 0;
#pragma unroll
    for (int q = 1; q < kMaxGroup; ++q)
        if (q < ga.n && (int64_t)blockIdx.x >= ga.block_start[q]) pi = q;
    int z, t2;
    if (ga.set_of[pi] >= 0) {   // slice-major, one XCD per slice (see MiGemmShareSet); block-uniform
        const MiGemmShareSet& st = ga.set[ga.set_of[pi]];
        const int64_t w = (int64_t)blockIdx.x - st.start;
        const int64_t q = w >> 3;
        z = (int)(q / st.G) * 8 + (int)(w & 7);
        if (z >= st.S) return;                        // padding of the last group of eight slices
        const int gi = (int)(q % st.G);
        int j = 0;
#pragma unroll
        for (int m = 1; m < kMaxGroup; ++m)
            if (m < st.count && gi >= st.tile_off[m]) j = m;
        pi = st.first + j;
        t2 = gi - st.tile_off[j];
    } else {
        const int64_t local = (int64_t)blockIdx.x - ga.block_start[pi];
        z = (int)(local / ga.p[pi].tiles_mn);
        t2 = (int)(local % ga.p[pi].tiles_mn);
        if (z >= ga.p[pi].splits) return;             // alignment padding in front of a share set
    }
    const MiGemmGroupProblem& g = ga.p[pi];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int64_t m0 = (int64_t)(t2 / g.tiles_n) * BM, n0 = (int64_t)(t2 % g.tiles_n) * BN;

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    // (Also measured and not kept, round 3: an LDS-free "row-stream" form of these launches — a wavefront owns 32 rows and all
    // <= 4 column tiles, lane (i, kk) reads eight consecutive k of its row as two 16-byte loads, weights from L1 / L2, next
    // group of eight loaded under the current MFMAs; bitwise the tile kernel's results.  The ranker's forward / dX launches
    // went 34 / 25 / 33 us -> 38 / 42 / 34 us: every load instruction touches 32 cache lines, and with two wavefronts per SIMD
    // a one-stage prefetch leaves ~0.8 us of latency per eight k exposed.  The same idea DOES pay for the transposed weight-
    // gradient products, whose operand rows are contiguous per k: csrc/wgrad.hip.)
    // Not software-pipelined like gemm_fast_kernel: measured (tools/ab_gemm_pipe.sh, round 3) the ranker's grouped launches
    // are short-K and bound by their operand traffic, four resident workgroups per CU already overlap one another's
    // loads and MFMAs, and the prologue's extra barrier cost 3-9 us per launch (iteration 0.641 -> 0.672 ms).
    for (int pp = 0; pp < g.n_pairs; ++pp) {
        const MiGemmPairArgs& pr = g.pair[pp];
        const bool a_kfast = pr.sa_k == 1, b_kfast = pr.sb_k == 1;
        int64_t k_lo = 0, k_hi = pr.K;
        if (g.splits > 1) {
            k_lo = (int64_t)z * g.k_per_split;
            k_hi = min(pr.K, k_lo + g.k_per_split);
        }
        for (int64_t kc = k_lo; kc < k_hi; kc += KC) {
            const int kw = (int)min((int64_t)KC, k_hi - kc);
            float4 va[kN4], vb[kN4];
            if (a_kfast) panel_issue<true>(va, pr.A, pr.sa_m, pr.sa_k, nullptr, m0, g.M, kc, kw, tid);
            else         panel_issue<false>(va, pr.A, pr.sa_m, pr.sa_k, nullptr, m0, g.M, kc, kw, tid);
            if (b_kfast) panel_issue<true>(vb, pr.B, pr.sb_n, pr.sb_k, nullptr, n0, g.N, kc, kw, tid);
            else         panel_issue<false>(vb, pr.B, pr.sb_n, pr.sb_k, nullptr, n0, g.N, kc, kw, tid);
            if (pp == 0 && g.a_mask) {
                if (a_kfast) panel_mask<true>(va, g.a_mask, pr.sa_m, pr.sa_k, m0, g.M, kc, kw, tid);
                else         panel_mask<false>(va, g.a_mask, pr.sa_m, pr.sa_k, m0, g.M, kc, kw, tid);
            }
            if (a_kfast) panel_commit<true>(As, va, tid); else panel_commit<false>(As, va, tid);
            if (b_kfast) panel_commit<true>(Bs, vb, tid); else panel_commit<false>(Bs, vb, tid);
            __syncthreads();
            const float* ap = &As[wm * 32 + (lane & 31)][lane >> 5];
            const float* bp = &Bs[wn * 32 + (lane & 31)][lane >> 5];
            if (kw == KC) {
#pragma unroll MI_GEMM_MFMA_UNROLL
                for (int s = 0; s < KC / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * s], bp[2 * s], acc, 0, 0, 0);
            } else {
                for (int s = 0; s < (kw + 1) / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * s], bp[2 * s], acc, 0, 0, 0);
            }
            __syncthreads();
        }
    }

    const int64_t gn = n0 + wn * 32 + (lane & 31);
    if (gn >= g.N) return;
    if (g.splits > 1) {
        float* slab = g.partial + (int64_t)z * g.M * g.N;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int64_t gm = m0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            if (gm < g.M) slab[gm * g.N + gn] = acc[reg];
        }
        return;
    }
    const float bv = g.bias ? g.bias[gn] : 0.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int64_t gm = m0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        if (gm >= g.M) continue;
        float v = acc[reg];
        if (g.bias) v += bv;
        float* c = g.C + gm * g.ldc + gn;
        if (g.accumulate) v += *c;
        if (g.act == 1) v = v > 0.f ? v : 0.f;
        *c = v;
    }
}

// The slab reduce of gemm_splitk_reduce_kernel for every split problem of a group in one launch.
__global__ __launch_bounds__(256) void gemm_group_reduce_kernel(MiGemmGroupArgs ga, int64_t total_all) {
    __shared__ float part[8][32];
    const int o = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int64_t i_all = (int64_t)blockIdx.x * 32 + o;
    int pi = 0;
#pragma unroll
    for (int q = 1; q < kMaxGroup; ++q)
        if (q < ga.n && i_all >= ga.out_start[q]) pi = q;
    const MiGemmGroupProblem& g = ga.p[pi];
    const int64_t total = g.M * g.N;
    const int64_t i = i_all - ga.out_start[pi];
    const bool live = i_all < total_all && g.splits > 1 && i < total;
    float v = 0.f;
    if (live)
        for (int zz = grp; zz < g.splits; zz += 8) v += g.partial[(int64_t)zz * total + i];
    part[grp][o] = v;
    __syncthreads();
    if (grp != 0 || !live) return;
    float acc = part[0][o];
#pragma unroll
    for (int q = 1; q < 8; ++q) acc += part[q][o];
    const int64_t m = i / g.N, n = i - m * g.N;
    if (g.bias) acc += g.bias[n];
    float* c = g.C + m * g.ldc + n;
    if (g.accumulate) acc += *c;
    if (g.act == 1) acc = acc > 0.f ? acc : 0.f;
    *c = acc;
}

// Sums the K-slice slabs, then the epilogue.  32 outputs per block, 8 lane groups each summing every
// 8th slice in ascending order; the 8 group sums are added in group order: a fixed association, so the
// result is bitwise reproducible (and equals summing the slices in order only up to fp32 rounding).
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(MiGemmArgs g) {
    __shared__ float part[8][32];
    const int o = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int64_t total = g.M * g.N;
    const int64_t i = (int64_t)blockIdx.x * 32 + o;
    float v = 0.f;
    if (i < total)
        for (int z = grp; z < g.splits; z += 8) v += g.partial[(int64_t)z * total + i];
    part[grp][o] = v;
    __syncthreads();
    if (grp != 0 || i >= total) return;
    float acc = part[0][o];
#pragma unroll
    for (int q = 1; q < 8; ++q) acc += part[q][o];
    const int64_t m = i / g.N, n = i - m * g.N;
    if (g.bias) acc += g.bias[n];
    float* c = g.C + m * g.ldc + n;
    if (g.accumulate) acc += *c;
    if (g.act == 1) acc = acc > 0.f ? acc : 0.f;
    *c = acc;
}

}  // namespace

// Split K only when the output tile grid cannot fill the chip and K is long (the weight-gradient
// products dW = dY^T X of the ranker: a 128 x 84 output over K = tens of thousands of nodes).
int mi_gemm_splits(int64_t M, int64_t N, int64_t K) {
    const int64_t blocks = mi_ceil_div(M, BM) * mi_ceil_div(N, BN);
    // Round 2: the threshold was K >= 2048 with >= 256 of K per slice, which left the article-side weight gradients of
    // a ranker batch (K = ~2 000 article nodes) and the decoder's 1-wide products (K = ~1 000 label edges) to 2-4
    // workgroups walking K alone: 50-60 us per launch (profiles/r2_ranker_v1.md).  One KC panel per slice is enough work.
    // The finer rule is kept to outputs of at most 8 tiles (weight-gradient shapes): wider outputs are forward products,
    // which stay one k-ascending chain (bitwise the oracle's) until K >= 2048 as before.
    const bool tiny = blocks <= 8;
    if (blocks >= 128 || K < (tiny ? 512 : 2048)) return 1;
    int64_t s = mi_ceil_div(512, blocks);
    const int64_t max_s = K / (tiny ? 128 : 256);  // at least 128 of K (256 for wider outputs) per slice, whatever the panel width
    if (s > max_s) s = max_s;
    return s < 2 ? 1 : (int)s;
}

int mi_gemm_launch(MiGemmArgs g, void* ws, size_t ws_bytes, hipStream_t stream) {
    if (g.M == 0 || g.N == 0) return 0;
    dim3 grid((unsigned)mi_ceil_div(g.N, BN), (unsigned)mi_ceil_div(g.M, BM), 1);
    if (grid.y > 65535u) return MI_ERR_TOO_LARGE;  // HIP grid.y limit: M <= 4.19M rows per launch
    int splits = mi_gemm_splits(g.M, g.N, g.K);
    if (splits > 1 && (ws == nullptr || ws_bytes < (size_t)splits * g.M * g.N * sizeof(float))) splits = 1;
    g.splits = splits;
    g.k_per_split = g.K;
    g.partial = nullptr;
    if (splits > 1) {
        g.k_per_split = mi_ceil_div(mi_ceil_div(g.K, splits), BK) * BK;
        g.splits = (int)mi_ceil_div(g.K, g.k_per_split);
        g.partial = static_cast<float*>(ws);
        grid.z = (unsigned)g.splits;
    }
    // fast path: float4-addressable operands.  K-contiguous operand: row stride % 4; output-contiguous
    // operand (stride 1 along its rows): k stride % 4 and no row gather.
    const bool a_kfast = g.sa_k == 1, b_kfast = g.sb_k == 1;
    const bool a_ok = a_kfast ? (g.sa_m % 4 == 0) : (g.sa_m == 1 && g.sa_k % 4 == 0 && g.a_rows == nullptr);
    const bool b_ok = b_kfast ? (g.sb_n % 4 == 0) : (g.sb_n == 1 && g.sb_k % 4 == 0);
    // float4 along k needs K (and every slice start) to be a multiple of 4; operands read along rows do not
    const bool k_ok = (!a_kfast && !b_kfast) || (g.K % 4 == 0 && g.k_per_split % 4 == 0);
    const bool fast = a_ok && b_ok && k_ok && mi_aligned16(g.A) && mi_aligned16(g.B);
    if (!fast)
        hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, stream, g);
    else if (a_kfast && b_kfast)
        hipLaunchKernelGGL((gemm_fast_kernel<true, true>), grid, dim3(256), 0, stream, g);
    else if (a_kfast)
        hipLaunchKernelGGL((gemm_fast_kernel<true, false>), grid, dim3(256), 0, stream, g);
    else if (b_kfast)
        hipLaunchKernelGGL((gemm_fast_kernel<false, true>), grid, dim3(256), 0, stream, g);
    else
        hipLaunchKernelGGL((gemm_fast_kernel<false, false>), grid, dim3(256), 0, stream, g);
    if (g.splits > 1) {
        const int64_t total = g.M * g.N;
        hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)mi_ceil_div(total, 32)), dim3(256), 0, stream, g);
    }
    return mi_launch_status();
}

namespace {

// fills one problem from the C-ABI descriptor; false = not eligible for the grouped (fast-path) kernel
bool group_fill(const mi_gemm_problem& d, MiGemmGroupProblem& g) {
    if (d.m <= 0 || d.n <= 0 || !d.C || d.ldc < d.n || d.k <= 0 || !d.A || !d.B) return false;
    g.M = d.m; g.N = d.n;
    g.n_pairs = d.k2 > 0 ? 2 : 1;
    const float* As[2] = {d.A, d.A2};
    const float* Bs[2] = {d.B, d.B2};
    const int64_t lda[2] = {d.lda, d.lda2}, ldb[2] = {d.ldb, d.ldb2}, K[2] = {d.k, d.k2};
    for (int p = 0; p < g.n_pairs; ++p) {
        MiGemmPairArgs& pr = g.pair[p];
        if (!As[p] || !Bs[p]) return false;
        pr.A = As[p]; pr.B = Bs[p]; pr.K = K[p];
        if (d.trans_a) { pr.sa_m = 1; pr.sa_k = lda[p]; if (lda[p] < d.m) return false; }
        else           { pr.sa_m = lda[p]; pr.sa_k = 1; if (lda[p] < K[p]) return false; }
        if (d.trans_b) { pr.sb_n = ldb[p]; pr.sb_k = 1; if (ldb[p] < K[p]) return false; }
        else           { pr.sb_n = 1; pr.sb_k = ldb[p]; if (ldb[p] < d.n) return false; }
        const bool a_kfast = pr.sa_k == 1, b_kfast = pr.sb_k == 1;
        const bool a_ok = a_kfast ? (pr.sa_m % 4 == 0) : (pr.sa_k % 4 == 0);
        const bool b_ok = b_kfast ? (pr.sb_n % 4 == 0) : (pr.sb_k % 4 == 0);
        const bool k_ok = (!a_kfast && !b_kfast) || (K[p] % 4 == 0);
        if (!(a_ok && b_ok && k_ok && mi_aligned16(pr.A) && mi_aligned16(pr.B))) return false;
    }
    if (d.a_mask && !mi_aligned16(d.a_mask)) return false;
    g.a_mask = d.a_mask; g.bias = d.bias; g.C = d.C; g.ldc = d.ldc;
    g.accumulate = d.accumulate; g.act = d.act;
    g.splits = 1; g.k_per_split = d.k; g.partial = nullptr;
    g.tiles_n = (int)mi_ceil_div(d.n, BN);
    g.tiles_mn = g.tiles_n * (int)mi_ceil_div(d.m, BM);
    return true;
}

int group_splits(const mi_gemm_problem& d) { return d.k2 > 0 ? 1 : mi_gemm_splits(d.m, d.n, d.k); }

}  // namespace

extern "C" size_t mi_gemm_group_workspace_bytes(const mi_gemm_problem* probs, int32_t n) {
    size_t total = 0;
    for (int i = 0; probs && i < n; ++i) {
        const int s = group_splits(probs[i]);
        if (s > 1) total += mi_align_up((size_t)s * (size_t)probs[i].m * (size_t)probs[i].n * sizeof(float), 256);
    }
    return total;
}

extern "C" int mi_gemm_group_supported(const mi_gemm_problem* probs, int32_t n) {
    if (n < 0 || (n > 0 && !probs)) return 0;
    for (int i = 0; i < n; ++i) {
        const mi_gemm_problem& d = probs[i];
        if (d.m == 0 || d.n == 0) continue;
        MiGemmGroupProblem g;
        memset(&g, 0, sizeof(g));
        if (!group_fill(d, g)) return 0;
    }
    return 1;
}

extern "C" int mi_gemm_group_f32(const mi_gemm_problem* probs, int32_t n, void* ws, size_t ws_bytes,
                                 mi_stream_t stream) {
    MI_CHECK_ARG(n >= 0 && (n == 0 || probs));
    hipStream_t s = (hipStream_t)stream;
    char* wp = static_cast<char*>(ws);
    size_t left = ws_bytes;
    for (int base = 0; base < n; base += kMaxGroup) {
        const int cnt = std::min(kMaxGroup, n - base);
        MiGemmGroupArgs ga;
        memset(&ga, 0, sizeof(ga));
        int64_t blocks = 0, outs = 0;
        bool any_split = false;
        for (int i = 0; i < cnt; ++i) {
            const mi_gemm_problem& d = probs[base + i];
            MI_CHECK_ARG(d.act == 0 || d.act == 1);
            if (d.m == 0 || d.n == 0) {  // nothing to write: an empty problem occupies no workgroups
                ga.block_start[i] = blocks; ga.out_start[i] = outs;
                ga.p[i].M = 0; ga.p[i].N = 0; ga.p[i].tiles_n = 1; ga.p[i].tiles_mn = 1; ga.p[i].splits = 1;
                continue;
            }
            if (!group_fill(d, ga.p[i])) return MI_ERR_UNSUPPORTED;
            MiGemmGroupProblem& g = ga.p[i];
            int sp = group_splits(d);
            if (sp > 1) {
                g.k_per_split = mi_ceil_div(mi_ceil_div(d.k, sp), BK) * BK;
                if (g.pair[0].sa_k == 1 || g.pair[0].sb_k == 1) g.k_per_split = mi_ceil_div(g.k_per_split, 4) * 4;
                sp = (int)mi_ceil_div(d.k, g.k_per_split);
                const size_t need = mi_align_up((size_t)sp * (size_t)d.m * (size_t)d.n * sizeof(float), 256);
                if (sp > 1 && (!wp || left < need)) return MI_ERR_WORKSPACE;
                if (sp > 1) { g.partial = reinterpret_cast<float*>(wp); wp += need; left -= need; any_split = true; }
            }
            g.splits = sp > 1 ? sp : 1;
            ga.block_start[i] = blocks;
            ga.out_start[i] = outs;
            blocks += (int64_t)g.tiles_mn * g.splits;
            if (g.splits > 1) outs += d.m * d.n;
        }
        ga.n = cnt;
        for (int i = cnt; i <= kMaxGroup; ++i) { ga.block_start[i] = blocks; ga.out_start[i] = outs; }
        if (blocks == 0) continue;
        // share sets (MiGemmShareSet): runs of split problems over the same A / K get ONE slicing (the coarsest of the
        // members': slabs were sized for at least as many slices) and the slice-major layout; the workgroup ranges are
        // re-laid from the first problem on
        auto same_a = [&](const MiGemmGroupProblem& x, const MiGemmGroupProblem& y) {
            return x.M > 0 && y.M > 0 && x.splits > 1 && y.splits > 1 && x.n_pairs == 1 && y.n_pairs == 1 &&
                   x.pair[0].A == y.pair[0].A && x.pair[0].K == y.pair[0].K && x.pair[0].sa_m == y.pair[0].sa_m &&
                   x.pair[0].sa_k == y.pair[0].sa_k && x.a_mask == y.a_mask;
        };
        for (int i = 0; i < cnt;) {
            int j = i + 1;
            while (j < cnt && same_a(ga.p[i], ga.p[j])) ++j;
            if (j - i >= 2) {
                int64_t kps = 0;
                for (int m = i; m < j; ++m) kps = std::max(kps, ga.p[m].k_per_split);
                for (int m = i; m < j; ++m) {
                    ga.p[m].k_per_split = kps;
                    ga.p[m].splits = (int)mi_ceil_div(ga.p[m].pair[0].K, kps);
                }
            }
            i = j;
        }
        for (int i = 0; i < kMaxGroup; ++i) ga.set_of[i] = -1;
        ga.n_sets = 0;
        {
            int64_t run = 0;
            int i = 0;
            while (i < cnt) {
                MiGemmGroupProblem& a0 = ga.p[i];
                int j = i + 1;
                if (a0.M > 0 && a0.splits > 1 && a0.n_pairs == 1) {
                    while (j < cnt && ga.p[j].M > 0 && ga.p[j].splits == a0.splits && ga.p[j].k_per_split == a0.k_per_split &&
                           ga.p[j].n_pairs == 1 && ga.p[j].pair[0].A == a0.pair[0].A && ga.p[j].pair[0].K == a0.pair[0].K &&
                           ga.p[j].pair[0].sa_m == a0.pair[0].sa_m && ga.p[j].pair[0].sa_k == a0.pair[0].sa_k &&
                           ga.p[j].a_mask == a0.a_mask)
                        ++j;
                }
                if (j - i >= 2 && ga.n_sets < kMaxSets) {
                    MiGemmShareSet& st = ga.set[ga.n_sets];
                    st.first = i; st.count = j - i; st.S = a0.splits; st.G = 0;
                    for (int m = i; m < j; ++m) { st.tile_off[m - i] = st.G; st.G += ga.p[m].tiles_mn; ga.set_of[m] = ga.n_sets; }
                    for (int m = j - i; m <= kMaxGroup; ++m) st.tile_off[m] = st.G;
                    run = (run + 7) / 8 * 8;
                    st.start = run;
                    for (int m = i; m < j; ++m) ga.block_start[m] = run;   // the whole set answers to one range
                    run += (int64_t)((st.S + 7) / 8) * 8 * st.G;
                    ++ga.n_sets;
                    i = j;
                } else {
                    ga.block_start[i] = run;
                    run += (int64_t)a0.tiles_mn * (a0.M > 0 ? a0.splits : 0);
                    ++i;
                }
            }
            blocks = run;
            for (int m = cnt; m <= kMaxGroup; ++m) ga.block_start[m] = blocks;
        }
        if (blocks >= INT32_MAX) return MI_ERR_TOO_LARGE;
        hipLaunchKernelGGL(gemm_group_kernel, dim3((unsigned)blocks), dim3(256), 0, s, ga);
        if (any_split)
            hipLaunchKernelGGL(gemm_group_reduce_kernel, dim3((unsigned)mi_ceil_div(outs, 32)), dim3(256), 0, s, ga, outs);
    }
    return mi_launch_status();
}

extern "C" size_t mi_gemm_workspace_bytes(int64_t m, int64_t n, int64_t k) {
    if (m <= 0 || n <= 0 || k <= 0) return 0;
    const int s = mi_gemm_splits(m, n, k);
    return s > 1 ? mi_align_up((size_t)s * (size_t)m * (size_t)n * sizeof(float), 256) : 0;
}

extern "C" int mi_gemm_f32(int32_t trans_a, int32_t trans_b, int64_t m, int64_t n, int64_t k,
                           const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias,
                           float* C, int64_t ldc, int32_t accumulate, int32_t act, void* ws,
                           size_t ws_bytes, mi_stream_t stream) {
    MI_CHECK_ARG(m >= 0 && n >= 0 && k >= 0);
    if (m == 0 || n == 0) return 0;
    MI_CHECK_ARG(C && ldc >= n && (k == 0 || (A && B)));
    MI_CHECK_ARG(act == 0 || act == 1);
    MiGemmArgs g;
    g.M = m; g.N = n; g.K = k;
    g.A = A; g.a_rows = nullptr;
    if (trans_a) { g.sa_m = 1; g.sa_k = lda; MI_CHECK_ARG(k == 0 || lda >= m); }   // A stored [k, m]
    else         { g.sa_m = lda; g.sa_k = 1; MI_CHECK_ARG(k == 0 || lda >= k); }   // A stored [m, k]
    g.B = B;
    if (trans_b) { g.sb_n = ldb; g.sb_k = 1; MI_CHECK_ARG(k == 0 || ldb >= k); }   // B stored [n, k] (Linear.weight)
    else         { g.sb_n = 1; g.sb_k = ldb; MI_CHECK_ARG(k == 0 || ldb >= n); }   // B stored [k, n]
    g.bias = bias; g.C = C; g.ldc = ldc; g.accumulate = accumulate; g.act = act;
    return mi_gemm_launch(g, ws, ws_bytes, (hipStream_t)stream);
}

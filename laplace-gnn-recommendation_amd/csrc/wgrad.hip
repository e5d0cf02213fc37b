// b1 / b9: the weight gradients of SAGEConv relations (include/laplace_hip.h, mi_sage_wgrad_f32).
//
// For a relation with destination features dy [k, m] (k = destination nodes of the batch: ~3*10^4; m = c_out <= 128):
//   gw_l [m, n1] = (dy * relu')^T agg,   gb [m] = (dy * relu')^T 1,   gw_r [m, n2] = (dy * relu')^T x_dst
// — three tall-skinny transposed products with a tiny output and a huge reduction dimension.  As three problems of the
// grouped GEMM (64 x 64 tiles, split-K, operands staged through LDS) the two relations of the first layer took 73 + 8 us
// of the ranker iteration: every problem re-reads dy and its relu mask, every 64-row K step is a load -> LDS -> barrier ->
// MFMA -> barrier phase, and A^T panels are strided.  This kernel needs no LDS and no transposition: the f32
// 32x32x2 MFMA wants A[i][kk] from lane (i, kk) and B[kk][j] from lane (j, kk) — with A = dy^T that is
// dy[k0 + kk][m0 + i]: 32 consecutive floats of row k0 and of row k0 + 1, i.e. two coalesced 128-byte reads per wavefront,
// straight from memory into the MFMA operand register; B = agg / x_dst rows likewise.  A wavefront owns ONE 32-column tile
// of the concatenated output [gw_l | gb | gw_r] and all (<= 4) row tiles of it, streams the rows of its K slice once, and
// writes a partial per slice; a second kernel sums the slices in slice order (deterministic).  dy and the mask are read
// once per group of four column tiles (from L1/L2 for the three other wavefronts of the workgroup).
#include "common.hpp"
#include <algorithm>

namespace {

typedef float wg_f32x16 __attribute__((ext_vector_type(16)));

#ifndef MI_WGRAD_ROWS
#define MI_WGRAD_ROWS 128   // rows of K per slice
#endif
constexpr int kRows = MI_WGRAD_ROWS;
#ifndef MI_WGRAD_U
#define MI_WGRAD_U 4   // k pairs per pipeline stage (8: 196 VGPRs, one wavefront per SIMD)
#endif
constexpr int kMaxProb = 4;
constexpr int kMaxMT = 4;       // m <= 128

struct WgProb {
    int64_t k;
    int m, n1, n2, has_bias;
    const float *dy, *mask, *b1, *b2;
    float *gw1, *gb, *gw2;
    int nt1, nt, groups, slices;      // column tiles of gw_l, all column tiles, groups of 4 tiles, K slices
    int64_t wg_begin;                 // first workgroup of this problem
    float* partial;                   // [slices][m][nt * 32]
};
struct WgArgs {
    WgProb p[kMaxProb];
    int n;
};

template <int MT, bool HAS_MASK>
__device__ __forceinline__ void wg_run(const WgProb& q, int slice, int ct, int lane, float* __restrict__ out) {
    const int i = lane & 31, kk = lane >> 5;
    const int64_t k0 = (int64_t)slice * kRows, k1 = min(q.k, k0 + kRows);
    wg_f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
    // this wavefront's B column: a column of b1, the ones column (bias), or a column of b2 — decided once (wavefront-uniform);
    // columns beyond the matrix read a valid address and are multiplied away
    const float* bp;
    int bld;
    bool b_ok, b_ones = false;
    if (ct < q.nt1) {
        const int col = ct * 32 + i;
        b_ok = col < q.n1; bld = q.n1; bp = q.b1 + (b_ok ? col : 0);
    } else if (q.has_bias && ct == q.nt1) {
        b_ones = true; b_ok = i == 0; bld = q.n1; bp = q.b1;
    } else {
        const int col = (ct - q.nt1 - q.has_bias) * 32 + i;
        b_ok = col < q.n2; bld = q.n2; bp = q.b2 + (b_ok ? col : 0);
    }
    // Row tile mt of this wavefront holds the output rows m = MT * r + mt (r = the MFMA's row index 0..31), not the rows
    // 32 mt + r: lane i then needs dy[k][MT i .. MT i + MT - 1] — ONE 16-byte load for four row tiles (a dy row is read by
    // its 32 lanes as one contiguous 512-byte run) instead of four 4-byte loads 128 bytes apart; the mask likewise.
    const float* ap = q.dy + MT * i;
    const float* mp = HAS_MASK ? q.mask + MT * i : nullptr;
    // Software pipeline: the operands of the NEXT U k-pairs are loaded while the MFMAs of the current U run.  The loads are
    // branch-free (rows beyond the slice are clamped and multiplied away): a version with a branch per element compiled to
    // ~6 branches and 2 waits per MFMA and took 106 us for the first layer's two relations.
    constexpr int U = MI_WGRAD_U;
    float a0[U][MT], b0[U], a1[U][MT], b1[U];   // two register stages, statically named (a runtime stage index would
                                                 // send the arrays to scratch)
    auto load = [&](float (&a)[U][MT], float (&b)[U], int64_t kb) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t k = kb + 2 * u + kk;
            const bool live = k < k1;
            const int kc = (int)(live ? k : k1 - 1);          // 32-bit offsets: k * max(m, n) < 2^31 (checked on the host)
            float bv = b_ones ? 1.f : bp[kc * bld];
            b[u] = (live && b_ok) ? bv : 0.f;
            const int ro = kc * q.m;
            float va[MT], vm[MT];
            if constexpr (MT == 4) {
                const float4 x = *reinterpret_cast<const float4*>(ap + ro);
                va[0] = x.x; va[1] = x.y; va[2] = x.z; va[3] = x.w;
                if (HAS_MASK) {
                    const float4 y = *reinterpret_cast<const float4*>(mp + ro);
                    vm[0] = y.x; vm[1] = y.y; vm[2] = y.z; vm[3] = y.w;
                }
            } else if constexpr (MT == 2) {
                const float2 x = *reinterpret_cast<const float2*>(ap + ro);
                va[0] = x.x; va[1] = x.y;
                if (HAS_MASK) {
                    const float2 y = *reinterpret_cast<const float2*>(mp + ro);
                    vm[0] = y.x; vm[1] = y.y;
                }
            } else {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    va[mt] = ap[ro + mt];
                    if (HAS_MASK) vm[mt] = mp[ro + mt];
                }
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float v = va[mt];
                if (HAS_MASK) v = vm[mt] > 0.f ? v : 0.f;
                a[u][mt] = live ? v : 0.f;
            }
        }
    };
    auto multiply = [&](const float (&a)[U][MT], const float (&b)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][mt], b[u], acc[mt], 0, 0, 0);
    };
    load(a0, b0, k0);
    for (int64_t kb = k0; kb < k1; kb += 4 * U) {
        load(a1, b1, kb + 2 * U);
        multiply(a0, b0);
        load(a0, b0, kb + 4 * U);
        multiply(a1, b1);   // all zeros when the slice ended inside the first stage
    }
    // partial[slice][row][ct * 32 + col]
    const int ld = q.nt * 32;
    float* base = out + ((int64_t)slice * q.m) * ld + ct * 32 + i;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = MT * (4 * kk + (r & 3) + 8 * (r >> 2)) + mt;   // the interleaved row tiles (see the loads)
            base[(int64_t)row * ld] = acc[mt][r];
        }
}

// grid: sum over problems of slices * groups workgroups of 4 wavefronts; wavefront w of a group takes column tile 4 g + w
__global__ __launch_bounds__(256) void sage_wgrad_kernel(WgArgs a) {
    int pi = 0;
#pragma unroll
    for (int j = 1; j < kMaxProb; ++j)
        if (j < a.n && (int64_t)blockIdx.x >= a.p[j].wg_begin) pi = j;
    const WgProb& q = a.p[pi];
    const int64_t local = (int64_t)blockIdx.x - q.wg_begin;
    const int slice = (int)(local / q.groups), g = (int)(local % q.groups);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ct = 4 * g + wave;
    if (ct >= q.nt) return;   // wavefront-uniform; no barriers in this kernel
    if (q.mask) {
        switch (q.m / 32) {
            case 1: wg_run<1, true>(q, slice, ct, lane, q.partial); break;
            case 2: wg_run<2, true>(q, slice, ct, lane, q.partial); break;
            case 3: wg_run<3, true>(q, slice, ct, lane, q.partial); break;
            default: wg_run<4, true>(q, slice, ct, lane, q.partial); break;
        }
    } else {
        switch (q.m / 32) {
            case 1: wg_run<1, false>(q, slice, ct, lane, q.partial); break;
            case 2: wg_run<2, false>(q, slice, ct, lane, q.partial); break;
            case 3: wg_run<3, false>(q, slice, ct, lane, q.partial); break;
            default: wg_run<4, false>(q, slice, ct, lane, q.partial); break;
        }
    }
}

// out element (row, col of the concatenated output) = sum over the slices in a fixed order.  64 elements per workgroup, four
// threads per element (thread g takes the slices s with s mod 4 == g, eight independent loads per step), combined through
// LDS as (g0 + g1) + (g2 + g3): one thread per element walking all ~235 slices left the chip at ~120 workgroups and
// 14.6 us for 27 MB of partials.
__global__ __launch_bounds__(256) void sage_wgrad_reduce_kernel(WgArgs a, int64_t total) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int64_t e = (int64_t)blockIdx.x * 64 + lane;
    float sum = 0.f;
    float* dst = nullptr;
    if (e < total) {
        int64_t rest = e;
        int pi = 0;
        for (; pi < a.n; ++pi) {
            const int64_t cnt = (int64_t)a.p[pi].m * a.p[pi].nt * 32;
            if (rest < cnt) break;
            rest -= cnt;
        }
        const WgProb& q = a.p[pi];
        const int ld = q.nt * 32;
        const int row = (int)(rest / ld), c = (int)(rest % ld);
        const int ct = c >> 5, j = c & 31;
        if (ct < q.nt1) {
            if (ct * 32 + j < q.n1) dst = q.gw1 + (int64_t)row * q.n1 + ct * 32 + j;
        } else if (q.has_bias && ct == q.nt1) {
            if (j == 0) dst = q.gb + row;
        } else {
            const int col = (ct - q.nt1 - q.has_bias) * 32 + j;
            if (col < q.n2) dst = q.gw2 + (int64_t)row * q.n2 + col;
        }
        if (dst) {
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const float* src = q.partial + (int64_t)row * ld + c;
            const int64_t stride = (int64_t)q.m * ld;
            int s = g;
            for (; s + 28 < q.slices; s += 32) {   // slices g, g + 4, ..., g + 28
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(s + 4 * u) * stride];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[u] += v[u];
            }
            for (int u = 0; s < q.slices; s += 4, ++u) acc[u] += src[(int64_t)s * stride];
            sum = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
        }
    }
    part[g][lane] = sum;
    __syncthreads();
    if (g == 0 && dst) *dst = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

bool wg_fill(const mi_wgrad_problem& d, WgProb& q) {
    if (d.k <= 0 || d.m <= 0 || d.m > 32 * kMaxMT || d.m % 32 != 0) return false;
    if (d.n1 <= 0 || d.n1 > 512 || d.n2 < 0 || d.n2 > 512) return false;
    if (d.k * (int64_t)std::max(d.m, std::max(d.n1, d.n2)) >= INT32_MAX) return false;   // the kernel's 32-bit row offsets
    if (!mi_aligned16(d.dy) || (d.mask && !mi_aligned16(d.mask))) return false;           // 16-byte row loads
    if (!d.dy || !d.b1 || !d.gw1 || (d.n2 > 0 && (!d.b2 || !d.gw2))) return false;
    q.k = d.k; q.m = d.m; q.n1 = d.n1; q.n2 = d.n2; q.has_bias = d.gb ? 1 : 0;
    q.dy = d.dy; q.mask = d.mask; q.b1 = d.b1; q.b2 = d.b2; q.gw1 = d.gw1; q.gb = d.gb; q.gw2 = d.gw2;
    q.nt1 = (d.n1 + 31) / 32;
    q.nt = q.nt1 + q.has_bias + (d.n2 + 31) / 32;
    q.groups = (q.nt + 3) / 4;
    q.slices = (int)mi_ceil_div(d.k, kRows);
    return true;
}

}  // namespace

extern "C" {

int mi_sage_wgrad_supported(const mi_wgrad_problem* probs, int32_t n) {
    if (n <= 0 || n > kMaxProb || !probs) return 0;
    for (int i = 0; i < n; ++i) {
        WgProb q;
        if (!wg_fill(probs[i], q)) return 0;
    }
    return 1;
}

size_t mi_sage_wgrad_workspace_bytes(const mi_wgrad_problem* probs, int32_t n) {
    size_t total = 0;
    for (int i = 0; probs && i < n && i < kMaxProb; ++i) {
        const mi_wgrad_problem& d = probs[i];
        if (d.k <= 0 || d.m <= 0) continue;
        const int64_t nt = (d.n1 + 31) / 32 + (d.gb ? 1 : 0) + (d.n2 + 31) / 32;
        total += mi_align_up((size_t)mi_ceil_div(d.k, kRows) * (size_t)d.m * (size_t)nt * 32 * sizeof(float), 256);
    }
    return total;
}

int mi_sage_wgrad_f32(const mi_wgrad_problem* probs, int32_t n, void* ws, size_t ws_bytes, mi_stream_t stream) {
    MI_CHECK_ARG(probs && n > 0 && ws);
    if (n > kMaxProb) return MI_ERR_UNSUPPORTED;
    WgArgs a;
    memset(&a, 0, sizeof(a));
    a.n = n;
    MiArena ar(ws, ws_bytes);
    int64_t wgs = 0, outs = 0;
    for (int i = 0; i < n; ++i) {
        WgProb& q = a.p[i];
        if (!wg_fill(probs[i], q)) return MI_ERR_UNSUPPORTED;
        q.wg_begin = wgs;
        wgs += (int64_t)q.slices * q.groups;
        outs += (int64_t)q.m * q.nt * 32;
        q.partial = ar.take<float>((size_t)q.slices * q.m * q.nt * 32);
        if (!q.partial) return MI_ERR_WORKSPACE;
    }
    if (wgs >= INT32_MAX) return MI_ERR_TOO_LARGE;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(sage_wgrad_kernel, dim3((unsigned)wgs), dim3(256), 0, s, a);
    hipLaunchKernelGGL(sage_wgrad_reduce_kernel, dim3((unsigned)mi_ceil_div(outs, 64)), dim3(256), 0, s, a, outs);
    return mi_launch_status();
}

}  // extern "C"

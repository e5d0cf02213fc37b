// Ranker head kernels: per-type BatchNorm1d over a batch's nodes (K8) and the decoder's gather + concat.
//   mi_batchnorm_*     BatchNorm1d(out_channels) in training and eval mode, model/encoder_decoder.py:98-99,144-150
//   mi_gather_cat_*    z = cat(z_user[row], z_item[col]) of EdgeDecoder.forward, model/encoder_decoder.py:57-63
// A batch is ~10^4 nodes x 64 channels (a few MB): these kernels are launch-bound, so the design goal is FEW launches
// with fixed reduction orders (no float atomics): statistics are reduced by kBnParts workgroups into per-part double
// sums, and every workgroup of the second kernel re-reduces those few partials itself (32 x C doubles from L2) instead
// of paying a third launch — forward = 2 launches, backward = 2 launches.
#include "common.hpp"
#include "pairs.hpp"

namespace {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / MI_WAVE;
constexpr int kBnParts = 128;   // partial workgroups of a reduction
constexpr int kBnMaxC = 512;    // channels (lanes loop in chunks of 64)

// part p, wave w: rows p*kWaves + w, then + kBnParts*kWaves, ... ; lane = channel.  A wave-load reads one row of C
// floats: coalesced.  Sums are kept in double per lane and combined across the block's waves through LDS in wave order.
template <bool BWD>
__global__ __launch_bounds__(kBlock) void bn_partial_kernel(int64_t n, int c, const float* __restrict__ X, int64_t ldx,
                                                            const float* __restrict__ dY, int64_t ldy,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            double* __restrict__ part /* [kBnParts, 2, c] */) {
    __shared__ double red[kWaves][2][MI_WAVE];
    const int lane = mi_lane(), wave = threadIdx.x / MI_WAVE, p = blockIdx.x;
    for (int c0 = 0; c0 < c; c0 += MI_WAVE) {
        const int ch = c0 + lane;
        double a = 0.0, b = 0.0;
        float mu = 0.f, is = 0.f;
        if (BWD && ch < c) { mu = mean[ch]; is = invstd[ch]; }
        constexpr int U = 4;  // rows in flight per lane: the walk is latency-bound, not bandwidth-bound
        const int64_t stride = (int64_t)kBnParts * kWaves;
        for (int64_t r0 = (int64_t)p * kWaves + wave; r0 < n; r0 += U * stride) {
            float x[U], g[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t r = r0 + u * stride;
                const bool ok = ch < c && r < n;
                x[u] = ok ? X[r * ldx + ch] : 0.f;
                g[u] = (BWD && ok) ? dY[r * ldy + ch] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {  // rows beyond n contribute exact zeros; order = ascending row
                if (BWD) {  // a = sum dy, b = sum dy * xhat
                    a += (double)g[u];
                    b += (double)g[u] * (double)((x[u] - mu) * is);
                } else {    // a = sum x, b = sum x^2
                    a += (double)x[u];
                    b += (double)x[u] * (double)x[u];
                }
            }
        }
        red[wave][0][lane] = a;
        red[wave][1][lane] = b;
        __syncthreads();
        if (wave == 0 && ch < c) {
            double sa = red[0][0][lane], sb = red[0][1][lane];
#pragma unroll
            for (int w = 1; w < kWaves; ++w) { sa += red[w][0][lane]; sb += red[w][1][lane]; }
            part[((int64_t)p * 2 + 0) * c + ch] = sa;
            part[((int64_t)p * 2 + 1) * c + ch] = sb;
        }
        __syncthreads();
    }
}

// The same partial sums with 16-byte loads (round 3): c a power of two between 4 and 256 and float4-addressable rows.
// A row is covered by LPR = c / 4 lanes, so one wave-load fetches 64 / LPR rows (c = 64: four rows, 1 KiB) and a block
// needs a quarter of the iterations of the one-row-per-wave walk above — at ~3*10^4 rows the walk is a chain of
// dependent load latencies, not bandwidth (20 -> 7 us).  Per-lane sums over "its" rows in ascending order, then a fixed
// xor tree over the wave's row groups, then the block's waves in order: deterministic, a different association from
// the scalar walk (both within rounding of the exact double sum).
template <bool BWD>
__device__ __forceinline__ void bn_partial4_body(int p, int64_t n, int c, const float* __restrict__ X, int64_t ldx,
                                                 const float* __restrict__ dY, int64_t ldy,
                                                 const float* __restrict__ mean,
                                                 const float* __restrict__ invstd,
                                                 double* __restrict__ part /* [kBnParts, 2, c] */) {
    __shared__ double red[kWaves][2][256];
    const int lane = mi_lane(), wave = threadIdx.x / MI_WAVE;
    const int lpr = c >> 2, rpw = MI_WAVE / lpr;
    const int sub = lane / lpr, q = lane - sub * lpr;
    double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
    float4 mu = mi_f4_zero(), is = mi_f4_zero();
    if (BWD) {
        mu = *reinterpret_cast<const float4*>(mean + 4 * q);
        is = *reinterpret_cast<const float4*>(invstd + 4 * q);
    }
    constexpr int U = 4;
    const int64_t stride = (int64_t)kBnParts * kWaves * rpw;
    for (int64_t r0 = ((int64_t)p * kWaves + wave) * rpw + sub; r0 < n; r0 += U * stride) {
        float4 x[U], g[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = r0 + u * stride;
            x[u] = r < n ? *reinterpret_cast<const float4*>(X + r * ldx + 4 * q) : mi_f4_zero();
            g[u] = (BWD && r < n) ? *reinterpret_cast<const float4*>(dY + r * ldy + 4 * q) : mi_f4_zero();
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#define MI_BN_ACC(f, k)                                                                  \
            if (BWD) { a[k] += (double)g[u].f; b[k] += (double)g[u].f * (double)((x[u].f - mu.f) * is.f); } \
            else     { a[k] += (double)x[u].f; b[k] += (double)x[u].f * (double)x[u].f; }
            MI_BN_ACC(x, 0) MI_BN_ACC(y, 1) MI_BN_ACC(z, 2) MI_BN_ACC(w, 3)
#undef MI_BN_ACC
        }
    }
    for (int m = MI_WAVE / 2; m >= lpr; m >>= 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a[k] += __shfl_xor(a[k], m, MI_WAVE);
            b[k] += __shfl_xor(b[k], m, MI_WAVE);
        }
    }
    if (sub == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            red[wave][0][4 * q + k] = a[k];
            red[wave][1][4 * q + k] = b[k];
        }
    }
    __syncthreads();
    for (int ch = threadIdx.x; ch < c; ch += kBlock) {
        double sa = red[0][0][ch], sb = red[0][1][ch];
#pragma unroll
        for (int w = 1; w < kWaves; ++w) { sa += red[w][0][ch]; sb += red[w][1][ch]; }
        part[((int64_t)p * 2 + 0) * c + ch] = sa;
        part[((int64_t)p * 2 + 1) * c + ch] = sb;
    }
}
template <bool BWD>
__global__ __launch_bounds__(kBlock) void bn_partial4_kernel(int64_t n, int c, const float* __restrict__ X, int64_t ldx,
                                                             const float* __restrict__ dY, int64_t ldy,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, double* __restrict__ part) {
    bn_partial4_body<BWD>(blockIdx.x, n, c, X, ldx, dY, ldy, mean, invstd, part);
}
// twin launch (mi_pairs): workgroups [0, kBnParts) take problem 0, [kBnParts, 2 kBnParts) problem 1
struct BnPart4 { int64_t n; const float* X; const float* dY; const float* mean; const float* invstd; double* part; };
template <bool BWD>
__global__ __launch_bounds__(kBlock) void bn_partial4_pair_kernel(BnPart4 a, BnPart4 b, int c) {
    const bool first = blockIdx.x < kBnParts;
    const BnPart4& q = first ? a : b;
    bn_partial4_body<BWD>(first ? blockIdx.x : blockIdx.x - kBnParts, q.n, c, q.X, c, q.dY, c, q.mean, q.invstd, q.part);
}

// Sum of the kBnParts partials of every channel, by the whole block: G = kBlock / cp groups of cp threads (cp = c rounded
// up to a power of two, <= 256) each add every G-th part, the groups are combined in order through LDS.  One thread per
// channel walking all 128 parts alone was a 13 us chain at the top of BOTH apply kernels, whatever the batch size.
__device__ __forceinline__ void bn_sum_parts(const double* __restrict__ part, int c, double (*tmp)[2][256], double* sa_out,
                                             double* sb_out) {
    int cp = 4;
    while (cp < c) cp <<= 1;
    if (cp > 256) {  // wide layers: one thread per channel (in chunks), parts in order
        for (int ch = threadIdx.x; ch < c; ch += kBlock) {
            double sa = 0.0, sb = 0.0;
            for (int p = 0; p < kBnParts; ++p) {
                sa += part[((int64_t)p * 2 + 0) * c + ch];
                sb += part[((int64_t)p * 2 + 1) * c + ch];
            }
            sa_out[ch] = sa;
            sb_out[ch] = sb;
        }
        __syncthreads();
        return;
    }
    const int G = kBlock / cp, grp = threadIdx.x / cp, ch = threadIdx.x % cp;
    double sa = 0.0, sb = 0.0;
    if (ch < c) {
        constexpr int UB = 8;  // loads in flight: the adds stay in part order
        for (int p0 = grp; p0 < kBnParts; p0 += UB * G) {
            double va[UB], vb[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int p = p0 + u * G;
                va[u] = p < kBnParts ? part[((int64_t)p * 2 + 0) * c + ch] : 0.0;
                vb[u] = p < kBnParts ? part[((int64_t)p * 2 + 1) * c + ch] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) { sa += va[u]; sb += vb[u]; }
        }
    }
    // tmp has room for 4 groups of 256 at cp = 64..256; smaller cp means more groups of fewer channels: same footprint
    double* flat = &tmp[0][0][0];
    flat[(grp * 2 + 0) * cp + ch] = sa;
    flat[(grp * 2 + 1) * cp + ch] = sb;
    __syncthreads();
    if (threadIdx.x < c) {
        double ta = 0.0, tb = 0.0;
        for (int g = 0; g < G; ++g) {
            ta += flat[(g * 2 + 0) * cp + threadIdx.x];
            tb += flat[(g * 2 + 1) * cp + threadIdx.x];
        }
        sa_out[threadIdx.x] = ta;
        sb_out[threadIdx.x] = tb;
    }
    __syncthreads();
}

// y = (x - mean) * invstd * gamma + beta.  TRAIN: mean / invstd come from the partial sums (re-reduced by every block
// in part order); block 0 also stores them for the backward and updates the running statistics (momentum, unbiased
// variance) exactly as torch.nn.BatchNorm1d does.  Eval: mean / var are the running statistics.
template <bool TRAIN>
__device__ __forceinline__ void bn_apply_body(unsigned bid, unsigned nblocks, int64_t n, int c, const float* __restrict__ X, int64_t ldx,
                                              const double* __restrict__ part,
                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                              float* __restrict__ run_mean, float* __restrict__ run_var,
                                              float momentum, float eps, float* __restrict__ save_mean,
                                              float* __restrict__ save_invstd, float* __restrict__ Y,
                                              int64_t ldy, int vec4) {
    __shared__ float s_scale[kBnMaxC], s_shift[kBnMaxC];
    __shared__ double s_tmp[kBlock / 64][2][256];
    __shared__ double s_sa[kBnMaxC], s_sb[kBnMaxC];
    if (TRAIN) bn_sum_parts(part, c, s_tmp, s_sa, s_sb);
    for (int ch = threadIdx.x; ch < c; ch += kBlock) {
        float mu, is;
        if (TRAIN) {
            const double sa = s_sa[ch], sb = s_sb[ch];
            const double m = sa / (double)n;
            double var = sb / (double)n - m * m;  // biased; double sums: no cancellation at fp32 resolution
            if (var < 0.0) var = 0.0;
            mu = (float)m;
            is = (float)(1.0 / sqrt(var + (double)eps));
            if (bid == 0) {
                save_mean[ch] = mu;
                save_invstd[ch] = is;
                if (run_mean) {
                    const double unb = n > 1 ? var * ((double)n / (double)(n - 1)) : var;
                    run_mean[ch] = (1.f - momentum) * run_mean[ch] + momentum * mu;
                    run_var[ch] = (1.f - momentum) * run_var[ch] + momentum * (float)unb;
                }
            }
        } else {
            mu = run_mean[ch];
            is = 1.0f / sqrtf(run_var[ch] + eps);
        }
        const float g = gamma ? gamma[ch] : 1.f, b = beta ? beta[ch] : 0.f;
        s_scale[ch] = is * g;
        s_shift[ch] = b - mu * is * g;
    }
    __syncthreads();
    if (vec4) {  // c, ldx, ldy multiples of 4 and 16-byte aligned bases (checked by the launcher): the same fma, four at a time
        const int c4 = c >> 2;
        const int64_t total4 = n * (int64_t)c4;
        for (int64_t i = (int64_t)bid * kBlock + threadIdx.x; i < total4; i += (int64_t)nblocks * kBlock) {
            const int64_t r = i / c4;
            const int ch = (int)(i - r * c4) * 4;
            const float4 x = *reinterpret_cast<const float4*>(X + r * ldx + ch);
            float4 y;
            y.x = fmaf(x.x, s_scale[ch], s_shift[ch]);
            y.y = fmaf(x.y, s_scale[ch + 1], s_shift[ch + 1]);
            y.z = fmaf(x.z, s_scale[ch + 2], s_shift[ch + 2]);
            y.w = fmaf(x.w, s_scale[ch + 3], s_shift[ch + 3]);
            *reinterpret_cast<float4*>(Y + r * ldy + ch) = y;
        }
        return;
    }
    const int64_t total = n * (int64_t)c;
    for (int64_t i = (int64_t)bid * kBlock + threadIdx.x; i < total; i += (int64_t)nblocks * kBlock) {
        const int64_t r = i / c;
        const int ch = (int)(i - r * c);
        Y[r * ldy + ch] = fmaf(X[r * ldx + ch], s_scale[ch], s_shift[ch]);
    }
}

template <bool TRAIN>
__global__ __launch_bounds__(kBlock) void bn_apply_kernel(int64_t n, int c, const float* __restrict__ X, int64_t ldx,
                                                          const double* __restrict__ part,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ run_mean, float* __restrict__ run_var,
                                                          float momentum, float eps, float* __restrict__ save_mean,
                                                          float* __restrict__ save_invstd, float* __restrict__ Y,
                                                          int64_t ldy, int vec4) {
    bn_apply_body<TRAIN>(blockIdx.x, gridDim.x, n, c, X, ldx, part, gamma, beta, run_mean, run_var, momentum, eps, save_mean,
                         save_invstd, Y, ldy, vec4);
}
struct BnApply { int64_t n; const float* X; const double* part; const float *gamma, *beta; float *run_mean, *run_var;
                 float momentum, eps; float *save_mean, *save_invstd; float* Y; };
__global__ __launch_bounds__(kBlock) void bn_apply_pair_kernel(BnApply a, BnApply b, int c, unsigned split, int vec4) {
    const bool first = blockIdx.x < split;
    const BnApply& q = first ? a : b;
    bn_apply_body<true>(first ? blockIdx.x : blockIdx.x - split, first ? split : gridDim.x - split, q.n, c, q.X, c, q.part, q.gamma,
                        q.beta, q.run_mean, q.run_var, q.momentum, q.eps, q.save_mean, q.save_invstd, q.Y, c, vec4);
}

// dx = gamma * invstd * (dy - mean(dy) - xhat * mean(dy * xhat)); block 0 writes dgamma = sum dy*xhat, dbeta = sum dy.
__device__ __forceinline__ void bn_bwd_apply_body(unsigned bid, unsigned nblocks, int64_t n, int c, const float* __restrict__ X, int64_t ldx,
                                                  const float* __restrict__ dY, int64_t ldy,
                                                  const double* __restrict__ part,
                                                  const float* __restrict__ gamma,
                                                  const float* __restrict__ mean,
                                                  const float* __restrict__ invstd, float* __restrict__ dX,
                                                  int64_t lddx, float* __restrict__ dgamma,
                                                  float* __restrict__ dbeta, int vec4) {
    __shared__ float s_a[kBnMaxC], s_b[kBnMaxC], s_mu[kBnMaxC], s_is[kBnMaxC], s_g[kBnMaxC];
    __shared__ double s_tmp[kBlock / 64][2][256];
    __shared__ double s_sa[kBnMaxC], s_sb[kBnMaxC];
    bn_sum_parts(part, c, s_tmp, s_sa, s_sb);
    for (int ch = threadIdx.x; ch < c; ch += kBlock) {
        const double sa = s_sa[ch], sb = s_sb[ch];
        if (bid == 0) {
            if (dbeta) dbeta[ch] = (float)sa;
            if (dgamma) dgamma[ch] = (float)sb;
        }
        s_a[ch] = (float)(sa / (double)n);
        s_b[ch] = (float)(sb / (double)n);
        s_mu[ch] = mean[ch];
        s_is[ch] = invstd[ch];
        s_g[ch] = (gamma ? gamma[ch] : 1.f) * invstd[ch];
    }
    __syncthreads();
    if (!dX) return;
    if (vec4) {
        const int c4 = c >> 2;
        const int64_t total4 = n * (int64_t)c4;
        for (int64_t i = (int64_t)bid * kBlock + threadIdx.x; i < total4; i += (int64_t)nblocks * kBlock) {
            const int64_t r = i / c4;
            const int ch = (int)(i - r * c4) * 4;
            const float4 x = *reinterpret_cast<const float4*>(X + r * ldx + ch);
            const float4 g = *reinterpret_cast<const float4*>(dY + r * ldy + ch);
            float4 o;
#define MI_BN_DX(f, k) { const float xh = (x.f - s_mu[ch + k]) * s_is[ch + k]; o.f = s_g[ch + k] * (g.f - s_a[ch + k] - xh * s_b[ch + k]); }
            MI_BN_DX(x, 0) MI_BN_DX(y, 1) MI_BN_DX(z, 2) MI_BN_DX(w, 3)
#undef MI_BN_DX
            *reinterpret_cast<float4*>(dX + r * lddx + ch) = o;
        }
        return;
    }
    const int64_t total = n * (int64_t)c;
    for (int64_t i = (int64_t)bid * kBlock + threadIdx.x; i < total; i += (int64_t)nblocks * kBlock) {
        const int64_t r = i / c;
        const int ch = (int)(i - r * c);
        const float xh = (X[r * ldx + ch] - s_mu[ch]) * s_is[ch];
        dX[r * lddx + ch] = s_g[ch] * (dY[r * ldy + ch] - s_a[ch] - xh * s_b[ch]);
    }
}

__global__ __launch_bounds__(kBlock) void bn_bwd_apply_kernel(int64_t n, int c, const float* __restrict__ X, int64_t ldx,
                                                              const float* __restrict__ dY, int64_t ldy,
                                                              const double* __restrict__ part,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, float* __restrict__ dX,
                                                              int64_t lddx, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int vec4) {
    bn_bwd_apply_body(blockIdx.x, gridDim.x, n, c, X, ldx, dY, ldy, part, gamma, mean, invstd, dX, lddx, dgamma, dbeta, vec4);
}
struct BnBwdApply { int64_t n; const float* X; const float* dY; const double* part; const float *gamma, *mean, *invstd; float* dX;
                    float *dgamma, *dbeta; };
__global__ __launch_bounds__(kBlock) void bn_bwd_apply_pair_kernel(BnBwdApply a, BnBwdApply b, int c, unsigned split, int vec4) {
    const bool first = blockIdx.x < split;
    const BnBwdApply& q = first ? a : b;
    bn_bwd_apply_body(first ? blockIdx.x : blockIdx.x - split, first ? split : gridDim.x - split, q.n, c, q.X, c, q.dY, c, q.part,
                      q.gamma, q.mean, q.invstd, q.dX, c, q.dgamma, q.dbeta, vec4);
}

// out[e, 0:cu] = Zu[row[e], :], out[e, cu:cu+ci] = Zi[col[e], :]; one wavefront per label edge
__global__ __launch_bounds__(kBlock) void gather_cat_kernel(int64_t n_e, int cu, int ci, const int64_t* __restrict__ row,
                                                            const int64_t* __restrict__ col, const float* __restrict__ Zu,
                                                            int64_t ldu, const float* __restrict__ Zi, int64_t ldi,
                                                            float* __restrict__ out, int64_t ldo) {
    const int64_t e = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / MI_WAVE;
    if (e >= n_e) return;
    const int lane = mi_lane();
    const float* su = Zu + row[e] * ldu;
    const float* si = Zi + col[e] * ldi;
    float* dst = out + e * ldo;
    for (int k = lane; k < cu; k += MI_WAVE) dst[k] = su[k];
    for (int k = lane; k < ci; k += MI_WAVE) dst[cu + k] = si[k];
}

// Backward of the gather: dZ[v, :] = sum over the label edges e (ascending) with idx[e] == v of dOut[e, off:off+c].
// One wavefront per label edge: the FIRST edge naming a node owns that node's row and sums all later edges naming it,
// in edge order — one writer per row, fixed order, no atomics, no sort.  The all-pairs scan costs n_e^2 / 64 compares
// per wavefront: meant for the decoder's label edges (10^3..10^4 per batch); the host falls back to a sorted reduction
// beyond kGatherBwdMaxEdges.  Rows no edge names stay zero (dZ is zero-filled by the caller).
// (a user's ~45 label edges are consecutive, and only a few dozen wavefronts own a row on the customer side: the walk is a
// chain of load latencies — 4 rows in flight: 27 us per launch at 24 users / batch, round 3)
constexpr int kGcbRows = 16;
__device__ __forceinline__ void gather_cat_bwd_body(int64_t e, int64_t n_e, int c, int off, const int64_t* __restrict__ idx,
                                                    const float* __restrict__ dOut, int64_t ldo,
                                                    float* __restrict__ dZ, int64_t ldz) {
    if (e >= n_e) return;
    const int lane = mi_lane();
    const int64_t v = idx[e];
    bool seen = false;  // does an earlier edge name v?
    for (int64_t q = lane; q < e; q += MI_WAVE) seen |= (idx[q] == v);
    if (__ballot(seen) != 0ull) return;
    for (int k0 = 0; k0 < c; k0 += MI_WAVE) {
        const int k = k0 + lane;
        float acc = (k < c) ? dOut[e * ldo + off + k] : 0.f;
        for (int64_t q0 = e + 1; q0 < n_e; q0 += MI_WAVE) {
            const int64_t q = q0 + lane;
            unsigned long long m = __ballot(q < n_e && idx[q] == v);
            while (m) {  // matching edges of this chunk, ascending; kGcbRows rows in flight, added in edge order
                int l[kGcbRows];
                float x[kGcbRows];
#pragma unroll
                for (int u = 0; u < kGcbRows; ++u) {
                    l[u] = m ? __ffsll((long long)m) - 1 : -1;
                    if (m) m &= m - 1;
                }
#pragma unroll
                for (int u = 0; u < kGcbRows; ++u) x[u] = (l[u] >= 0 && k < c) ? dOut[(q0 + l[u]) * ldo + off + k] : 0.f;
#pragma unroll
                for (int u = 0; u < kGcbRows; ++u)
                    if (l[u] >= 0) acc += x[u];
            }
        }
        if (k < c) dZ[v * ldz + k] = acc;
    }
}
__global__ __launch_bounds__(kBlock) void gather_cat_bwd_kernel(int64_t n_e, int c, int off, const int64_t* __restrict__ idx,
                                                                const float* __restrict__ dOut, int64_t ldo,
                                                                float* __restrict__ dZ, int64_t ldz) {
    gather_cat_bwd_body(((int64_t)blockIdx.x * kBlock + threadIdx.x) / MI_WAVE, n_e, c, off, idx, dOut, ldo, dZ, ldz);
}
// twin launch (mi_pairs): the first `split` workgroups sum the customer half (columns [0, c) by idx0), the rest the article half
__global__ __launch_bounds__(kBlock) void gather_cat_bwd_pair_kernel(int64_t n_e, int c, const int64_t* __restrict__ idx0,
                                                                     const int64_t* __restrict__ idx1, const float* __restrict__ dOut,
                                                                     int64_t ldo, float* __restrict__ dZ0, float* __restrict__ dZ1,
                                                                     unsigned split) {
    const bool first = blockIdx.x < split;
    const unsigned bid = first ? blockIdx.x : blockIdx.x - split;
    gather_cat_bwd_body(((int64_t)bid * kBlock + threadIdx.x) / MI_WAVE, n_e, c, first ? 0 : c, first ? idx0 : idx1, dOut, ldo,
                        first ? dZ0 : dZ1, c);
}

// gather + concat + Philox dropout (pairs.hpp: gather_cat_dropout).  32-bit Philox4x32-10 as exec_common.hpp's dropout_kernel:
// counter (i lo, i hi, site, step), key (k0, k1), i = flat float4 index of the output matrix.
__global__ __launch_bounds__(kBlock) void gather_cat_dropout_kernel(int64_t n_e, int cu4, int ci4, const int64_t* __restrict__ row,
                                                                    const int64_t* __restrict__ col, const float4* __restrict__ Zu,
                                                                    const float4* __restrict__ Zi, float4* __restrict__ out, float p,
                                                                    float scale, uint32_t k0, uint32_t k1, uint32_t site,
                                                                    uint32_t step_lo) {
    const int w4 = cu4 + ci4;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_e * w4) return;
    const int64_t e = i / w4;
    const int c = (int)(i - e * w4);
    float4 v = c < cu4 ? Zu[row[e] * cu4 + c] : Zi[col[e] * ci4 + (c - cu4)];
    const MiPhilox r = mi_philox4x32((uint32_t)i, (uint32_t)((uint64_t)i >> 32), site, step_lo, k0, k1);
    const uint32_t thr = (uint32_t)fminf(4294967040.f, p * 4294967296.f);  // keep when the draw is >= p * 2^32
    v.x = r.c[0] >= thr ? v.x * scale : 0.f;
    v.y = r.c[1] >= thr ? v.y * scale : 0.f;
    v.z = r.c[2] >= thr ? v.z * scale : 0.f;
    v.w = r.c[3] >= thr ? v.w * scale : 0.f;
    out[i] = v;
}

// BCEWithLogitsLoss(reduction="mean") and its gradient in one launch, one workgroup: loss = mean(max(x, 0) - x*y +
// log1p(exp(-|x|))) (torch's formulation), dx = (sigmoid(x) - y) / n.  Sums in double, fixed tree: reproducible.
__global__ __launch_bounds__(1024) void bce_logits_kernel(int64_t n, const float* __restrict__ x, const float* __restrict__ y,
                                                          float* __restrict__ loss, float* __restrict__ dx) {
    __shared__ double red[1024];
    double acc = 0.0;
    const float inv_n = 1.0f / (float)n;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const float xi = x[i], yi = y[i];
        acc += (double)(fmaxf(xi, 0.f) - xi * yi + log1pf(expf(-fabsf(xi))));
        if (dx) dx[i] = (1.0f / (1.0f + expf(-xi)) - yi) * inv_n;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = (float)(red[0] / (double)n);
}

}  // namespace

static inline bool bn_pow2_rows(int64_t c) { return c >= 4 && c <= 256 && (c & (c - 1)) == 0; }

// Backward of a Linear layer with ONE output (the decoder's last layer, model/layers.py + model/encoder_decoder.py:66-72):
//   dx[r, c] = dy[r] w[c];  gw[c] = sum_r dy[r] x[r, c];  gb = sum_r dy[r].
// As three [n, 1]-shaped products these were three launches of the scalar GEMM fallback plus two split-K reduces (48 us of
// the ranker iteration); here a partial pass over 64-row bands (a thread per column) and a reduce over the bands in band
// order — no atomics, the sums do not depend on scheduling.
constexpr int kL1Rows = 64;
__global__ __launch_bounds__(kBlock) void linear1_bwd_partial_kernel(int64_t n, int in, const float* __restrict__ dy,
                                                                     const float* __restrict__ w, const float* __restrict__ x,
                                                                     int64_t ldx, float* __restrict__ dx, int64_t lddx,
                                                                     float* __restrict__ part /*[bands, in + 1]*/) {
    __shared__ float sdy[kL1Rows];
    const int64_t r0 = (int64_t)blockIdx.x * kL1Rows;
    const int rows = (int)min((int64_t)kL1Rows, n - r0);
    if (threadIdx.x < kL1Rows) sdy[threadIdx.x] = (int)threadIdx.x < rows ? dy[r0 + threadIdx.x] : 0.f;
    __syncthreads();
    for (int c = threadIdx.x; c < in; c += kBlock) {
        const float wc = w[c];
        float acc = 0.f;
#pragma unroll 8
        for (int r = 0; r < rows; ++r) {
            const float g = sdy[r];
            acc = fmaf(g, x[(r0 + r) * ldx + c], acc);
            if (dx) dx[(r0 + r) * lddx + c] = g * wc;
        }
        part[(int64_t)blockIdx.x * (in + 1) + c] = acc;
    }
    if (threadIdx.x == 0) {
        float sb = 0.f;
        for (int r = 0; r < rows; ++r) sb += sdy[r];
        part[(int64_t)blockIdx.x * (in + 1) + in] = sb;
    }
}
__global__ __launch_bounds__(kBlock) void linear1_bwd_reduce_kernel(int64_t bands, int in, const float* __restrict__ part,
                                                                    float* __restrict__ gw, float* __restrict__ gb) {
    const int c = blockIdx.x * kBlock + threadIdx.x;
    if (c > in) return;
    float acc = 0.f;
    for (int64_t b = 0; b < bands; ++b) acc += part[b * (in + 1) + c];
    if (c < in) gw[c] = acc;
    else if (gb) gb[0] = acc;
}

// The two kernels above as one launch (pairs.hpp: linear1_bwd_lastblock): every workgroup writes its band's partial row, takes a
// ticket, and the workgroup holding the last ticket reduces all bands in band order — the sums of linear1_bwd_reduce_kernel.
// Release / acquire at agent scope around the ticket (the partial rows were written by workgroups on other XCDs, whose L2s are
// not coherent with this one's): every thread fences after its stores, one thread adds, the last workgroup fences before it reads.
__global__ __launch_bounds__(kBlock) void linear1_bwd_fused_kernel(int64_t n, int in, const float* __restrict__ dy,
                                                                   const float* __restrict__ w, const float* __restrict__ x,
                                                                   int64_t ldx, float* __restrict__ dx, int64_t lddx,
                                                                   float* __restrict__ part, float* __restrict__ gw,
                                                                   float* __restrict__ gb, int32_t* __restrict__ counter) {
    __shared__ float sdy[kL1Rows];
    __shared__ int s_last;
    const int64_t r0 = (int64_t)blockIdx.x * kL1Rows;
    const int rows = (int)min((int64_t)kL1Rows, n - r0);
    if (threadIdx.x < kL1Rows) sdy[threadIdx.x] = (int)threadIdx.x < rows ? dy[r0 + threadIdx.x] : 0.f;
    __syncthreads();
    for (int c = threadIdx.x; c < in; c += kBlock) {
        const float wc = w[c];
        float acc = 0.f;
#pragma unroll 8
        for (int r = 0; r < rows; ++r) {
            const float g = sdy[r];
            acc = fmaf(g, x[(r0 + r) * ldx + c], acc);
            if (dx) dx[(r0 + r) * lddx + c] = g * wc;
        }
        part[(int64_t)blockIdx.x * (in + 1) + c] = acc;
    }
    if (threadIdx.x == 0) {
        float sb = 0.f;
        for (int r = 0; r < rows; ++r) sb += sdy[r];
        part[(int64_t)blockIdx.x * (in + 1) + in] = sb;
    }
    __threadfence();                       // release: this thread's partial-row stores, at agent scope
    __syncthreads();
    if (threadIdx.x == 0) s_last = (atomicAdd(counter, 1) == (int)gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    if (!s_last) return;
    __threadfence();                       // acquire: the other workgroups' partial rows
    const int64_t bands = gridDim.x;
    for (int c = threadIdx.x; c <= in; c += kBlock) {
        float acc = 0.f;
        for (int64_t b = 0; b < bands; ++b) acc += part[b * (in + 1) + c];
        if (c < in) gw[c] = acc;
        else if (gb) gb[0] = acc;
    }
    if (threadIdx.x == 0) *counter = 0;    // ready for the next call
}

extern "C" {

size_t mi_linear1_bwd_workspace_bytes(int64_t n, int64_t in) {
    return (size_t)std::max<int64_t>(mi_ceil_div(n, kL1Rows), 1) * (size_t)(in + 1) * sizeof(float);
}

int mi_linear1_bwd_f32(int64_t n, int64_t in, const float* dy, const float* w, const float* x, int64_t ldx, float* dx,
                       int64_t lddx, float* gw, float* gb, void* ws, size_t ws_bytes, mi_stream_t stream) {
    MI_CHECK_ARG(n > 0 && in > 0 && dy && w && x && gw && ldx >= in && (!dx || lddx >= in));
    MI_CHECK_ARG(ws && ws_bytes >= mi_linear1_bwd_workspace_bytes(n, in));
    const int64_t bands = mi_ceil_div(n, kL1Rows);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(linear1_bwd_partial_kernel, dim3((unsigned)bands), dim3(kBlock), 0, s, n, (int)in, dy, w, x, ldx, dx, lddx,
                       static_cast<float*>(ws));
    hipLaunchKernelGGL(linear1_bwd_reduce_kernel, dim3((unsigned)mi_ceil_div(in + 1, kBlock)), dim3(kBlock), 0, s, bands, (int)in,
                       static_cast<const float*>(ws), gw, gb);
    return mi_launch_status();
}

int mi_bce_logits_f32(int64_t n, const float* logits, const float* labels, float* loss, float* dlogits, mi_stream_t stream) {
    MI_CHECK_ARG(n > 0 && logits && labels && loss);
    hipLaunchKernelGGL(bce_logits_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, n, logits, labels, loss, dlogits);
    return mi_launch_status();
}

size_t mi_batchnorm_workspace_bytes(int64_t c) { return (size_t)kBnParts * 2 * (size_t)(c > 0 ? c : 1) * sizeof(double); }

int mi_batchnorm_fwd_f32(int64_t n, int64_t c, const float* X, int64_t ldx, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, float momentum, float eps, int32_t training,
                         float* save_mean, float* save_invstd, float* Y, int64_t ldy, void* ws, size_t ws_bytes,
                         mi_stream_t stream) {
    MI_CHECK_ARG(n >= 0 && c > 0);
    if (c > kBnMaxC) return MI_ERR_UNSUPPORTED;
    if (n == 0) return 0;
    MI_CHECK_ARG(X && Y && ldx >= c && ldy >= c);
    hipStream_t s = (hipStream_t)stream;
    const unsigned grid = (unsigned)std::min<int64_t>(1024, mi_ceil_div(n * c, kBlock * 4));
    const int vec4 = (c % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && mi_aligned16(X) && mi_aligned16(Y)) ? 1 : 0;
    if (training) {
        MI_CHECK_ARG(save_mean && save_invstd && ws && ws_bytes >= mi_batchnorm_workspace_bytes(c));
        MI_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr));
        double* part = reinterpret_cast<double*>(ws);
        if (bn_pow2_rows(c) && ldx % 4 == 0 && mi_aligned16(X))
            hipLaunchKernelGGL(bn_partial4_kernel<false>, dim3(kBnParts), dim3(kBlock), 0, s, n, (int)c, X, ldx, nullptr, 0,
                               nullptr, nullptr, part);
        else
            hipLaunchKernelGGL(bn_partial_kernel<false>, dim3(kBnParts), dim3(kBlock), 0, s, n, (int)c, X, ldx, nullptr, 0,
                               nullptr, nullptr, part);
        hipLaunchKernelGGL(bn_apply_kernel<true>, dim3(grid ? grid : 1), dim3(kBlock), 0, s, n, (int)c, X, ldx, part, gamma,
                           beta, running_mean, running_var, momentum, eps, save_mean, save_invstd, Y, ldy, vec4);
    } else {
        MI_CHECK_ARG(running_mean && running_var);
        hipLaunchKernelGGL(bn_apply_kernel<false>, dim3(grid ? grid : 1), dim3(kBlock), 0, s, n, (int)c, X, ldx, nullptr,
                           gamma, beta, running_mean, running_var, momentum, eps, nullptr, nullptr, Y, ldy, vec4);
    }
    return mi_launch_status();
}

int mi_batchnorm_bwd_f32(int64_t n, int64_t c, const float* X, int64_t ldx, const float* dY, int64_t ldy,
                         const float* gamma, const float* save_mean, const float* save_invstd, float* dX, int64_t lddx,
                         float* dgamma, float* dbeta, void* ws, size_t ws_bytes, mi_stream_t stream) {
    MI_CHECK_ARG(n >= 0 && c > 0);
    if (c > kBnMaxC) return MI_ERR_UNSUPPORTED;
    if (n == 0) return 0;
    MI_CHECK_ARG(X && dY && save_mean && save_invstd && ldx >= c && ldy >= c && (!dX || lddx >= c));
    MI_CHECK_ARG(ws && ws_bytes >= mi_batchnorm_workspace_bytes(c));
    hipStream_t s = (hipStream_t)stream;
    double* part = reinterpret_cast<double*>(ws);
    if (bn_pow2_rows(c) && ldx % 4 == 0 && ldy % 4 == 0 && mi_aligned16(X) && mi_aligned16(dY) && mi_aligned16(save_mean) &&
        mi_aligned16(save_invstd))
        hipLaunchKernelGGL(bn_partial4_kernel<true>, dim3(kBnParts), dim3(kBlock), 0, s, n, (int)c, X, ldx, dY, ldy, save_mean,
                           save_invstd, part);
    else
        hipLaunchKernelGGL(bn_partial_kernel<true>, dim3(kBnParts), dim3(kBlock), 0, s, n, (int)c, X, ldx, dY, ldy, save_mean,
                           save_invstd, part);
    const unsigned grid = dX ? (unsigned)std::min<int64_t>(1024, mi_ceil_div(n * c, kBlock * 4)) : 1u;
    const int vec4 = (dX && c % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && lddx % 4 == 0 && mi_aligned16(X) && mi_aligned16(dY) &&
                      mi_aligned16(dX)) ? 1 : 0;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid ? grid : 1), dim3(kBlock), 0, s, n, (int)c, X, ldx, dY, ldy, part,
                       gamma, save_mean, save_invstd, dX, lddx, dgamma, dbeta, vec4);
    return mi_launch_status();
}

int mi_gather_cat_f32(int64_t n_edges, int64_t cu, int64_t ci, const int64_t* row, const int64_t* col, const float* Zu,
                      int64_t ldu, const float* Zi, int64_t ldi, float* out, int64_t ldo, mi_stream_t stream) {
    MI_CHECK_ARG(n_edges >= 0 && cu >= 0 && ci >= 0 && cu + ci > 0);
    if (n_edges == 0) return 0;
    MI_CHECK_ARG(row && col && out && ldo >= cu + ci && (cu == 0 || (Zu && ldu >= cu)) && (ci == 0 || (Zi && ldi >= ci)));
    hipLaunchKernelGGL(gather_cat_kernel, dim3((unsigned)mi_ceil_div(n_edges * MI_WAVE, kBlock)), dim3(kBlock), 0,
                       (hipStream_t)stream, n_edges, (int)cu, (int)ci, row, col, Zu, ldu, Zi, ldi, out, ldo);
    return mi_launch_status();
}

int64_t mi_gather_cat_bwd_max_edges(void) { return 1 << 15; }

int mi_gather_cat_bwd_f32(int64_t n_edges, int64_t c, int64_t off, const int64_t* idx, const float* dOut, int64_t ldo,
                          float* dZ, int64_t ldz, mi_stream_t stream) {
    MI_CHECK_ARG(n_edges >= 0 && c > 0 && off >= 0);
    if (n_edges == 0) return 0;
    if (n_edges > mi_gather_cat_bwd_max_edges()) return MI_ERR_UNSUPPORTED;
    MI_CHECK_ARG(idx && dOut && dZ && ldo >= off + c && ldz >= c);
    hipLaunchKernelGGL(gather_cat_bwd_kernel, dim3((unsigned)mi_ceil_div(n_edges * MI_WAVE, kBlock)), dim3(kBlock), 0,
                       (hipStream_t)stream, n_edges, (int)c, (int)off, idx, dOut, ldo, dZ, ldz);
    return mi_launch_status();
}

}  // extern "C"

// ---- twin launchers (pairs.hpp) ---------------------------------------------------------------------------------------------
namespace mi_pairs {

static bool bn_pair_ok(const BnSide& a, const BnSide& b, int64_t c) {
    // the single launches' fast path on both sides: power-of-two channel count, float4-addressable rows (ld = c)
    return bn_pow2_rows(c) && a.n > 0 && b.n > 0 && a.X && b.X && a.ws && b.ws && a.save_mean && a.save_invstd && b.save_mean &&
           b.save_invstd && mi_aligned16(a.X) && mi_aligned16(b.X) && mi_aligned16(a.save_mean) && mi_aligned16(a.save_invstd) &&
           mi_aligned16(b.save_mean) && mi_aligned16(b.save_invstd);
}
static unsigned bn_grid(int64_t n, int64_t c) {
    const unsigned g = (unsigned)std::min<int64_t>(1024, mi_ceil_div(n * c, kBlock * 4));   // mi_batchnorm_*_f32's own grid
    return g ? g : 1u;
}

int batchnorm_fwd_pair(const BnSide& a, const BnSide& b, int64_t c, hipStream_t s) {
    if (!bn_pair_ok(a, b, c) || !a.Y || !b.Y || !mi_aligned16(a.Y) || !mi_aligned16(b.Y)) return MI_ERR_UNSUPPORTED;
    if ((a.running_mean == nullptr) != (a.running_var == nullptr) || (b.running_mean == nullptr) != (b.running_var == nullptr))
        return MI_ERR_UNSUPPORTED;
    BnPart4 pa{a.n, a.X, nullptr, nullptr, nullptr, reinterpret_cast<double*>(a.ws)};
    BnPart4 pb{b.n, b.X, nullptr, nullptr, nullptr, reinterpret_cast<double*>(b.ws)};
    hipLaunchKernelGGL(bn_partial4_pair_kernel<false>, dim3(2 * kBnParts), dim3(kBlock), 0, s, pa, pb, (int)c);
    BnApply qa{a.n, a.X, pa.part, a.gamma, a.beta, a.running_mean, a.running_var, a.momentum, a.eps, a.save_mean, a.save_invstd, a.Y};
    BnApply qb{b.n, b.X, pb.part, b.gamma, b.beta, b.running_mean, b.running_var, b.momentum, b.eps, b.save_mean, b.save_invstd, b.Y};
    const unsigned ga = bn_grid(a.n, c), gb = bn_grid(b.n, c);
    hipLaunchKernelGGL(bn_apply_pair_kernel, dim3(ga + gb), dim3(kBlock), 0, s, qa, qb, (int)c, ga, 1);
    return mi_launch_status();
}

int batchnorm_bwd_pair(const BnSide& a, const BnSide& b, int64_t c, hipStream_t s) {
    if (!bn_pair_ok(a, b, c) || !a.dY || !b.dY || !a.dX || !b.dX || !mi_aligned16(a.dY) || !mi_aligned16(b.dY) || !mi_aligned16(a.dX) ||
        !mi_aligned16(b.dX))
        return MI_ERR_UNSUPPORTED;
    BnPart4 pa{a.n, a.X, a.dY, a.save_mean, a.save_invstd, reinterpret_cast<double*>(a.ws)};
    BnPart4 pb{b.n, b.X, b.dY, b.save_mean, b.save_invstd, reinterpret_cast<double*>(b.ws)};
    hipLaunchKernelGGL(bn_partial4_pair_kernel<true>, dim3(2 * kBnParts), dim3(kBlock), 0, s, pa, pb, (int)c);
    BnBwdApply qa{a.n, a.X, a.dY, pa.part, a.gamma, a.save_mean, a.save_invstd, a.dX, a.dgamma, a.dbeta};
    BnBwdApply qb{b.n, b.X, b.dY, pb.part, b.gamma, b.save_mean, b.save_invstd, b.dX, b.dgamma, b.dbeta};
    const unsigned ga = bn_grid(a.n, c), gb = bn_grid(b.n, c);
    hipLaunchKernelGGL(bn_bwd_apply_pair_kernel, dim3(ga + gb), dim3(kBlock), 0, s, qa, qb, (int)c, ga, 1);
    return mi_launch_status();
}

int gather_cat_bwd_pair(int64_t n_edges, int64_t c, const int64_t* idx0, const int64_t* idx1, const float* dOut, int64_t ldo,
                        float* dZ0, float* dZ1, hipStream_t s) {
    if (n_edges <= 0 || n_edges > mi_gather_cat_bwd_max_edges() || c <= 0 || !idx0 || !idx1 || !dOut || !dZ0 || !dZ1 || ldo < 2 * c)
        return MI_ERR_UNSUPPORTED;
    const unsigned g = (unsigned)mi_ceil_div(n_edges * MI_WAVE, kBlock);
    hipLaunchKernelGGL(gather_cat_bwd_pair_kernel, dim3(2 * g), dim3(kBlock), 0, s, n_edges, (int)c, idx0, idx1, dOut, ldo, dZ0, dZ1, g);
    return mi_launch_status();
}

int gather_cat_dropout(int64_t n_edges, int64_t cu, int64_t ci, const int64_t* row, const int64_t* col, const float* Zu, const float* Zi,
                       float* out, float p, uint64_t seed, uint32_t site, uint32_t step_lo, hipStream_t s) {
    if (n_edges <= 0 || cu <= 0 || ci <= 0 || cu % 4 || ci % 4 || !row || !col || !Zu || !Zi || !out || !(p > 0.f && p < 1.f))
        return MI_ERR_UNSUPPORTED;
    if (!mi_aligned16(Zu) || !mi_aligned16(Zi) || !mi_aligned16(out)) return MI_ERR_UNSUPPORTED;
    const int64_t total4 = n_edges * ((cu + ci) / 4);
    hipLaunchKernelGGL(gather_cat_dropout_kernel, dim3((unsigned)mi_ceil_div(total4, kBlock)), dim3(kBlock), 0, s, n_edges, (int)(cu / 4),
                       (int)(ci / 4), row, col, reinterpret_cast<const float4*>(Zu), reinterpret_cast<const float4*>(Zi),
                       reinterpret_cast<float4*>(out), p, 1.0f / (1.0f - p), (uint32_t)seed, (uint32_t)(seed >> 32), site, step_lo);
    return mi_launch_status();
}

int linear1_bwd_lastblock(int64_t n, int64_t in, const float* dy, const float* w, const float* x, float* dx, float* gw, float* gb,
                          void* ws, size_t ws_bytes, int32_t* counter, hipStream_t s) {
    if (n <= 0 || in <= 0 || !dy || !w || !x || !gw || !ws || !counter || ws_bytes < mi_linear1_bwd_workspace_bytes(n, in))
        return MI_ERR_UNSUPPORTED;
    const int64_t bands = mi_ceil_div(n, kL1Rows);
    hipLaunchKernelGGL(linear1_bwd_fused_kernel, dim3((unsigned)bands), dim3(kBlock), 0, s, n, (int)in, dy, w, x, in, dx, in,
                       static_cast<float*>(ws), gw, gb, counter);
    return mi_launch_status();
}

}  // namespace mi_pairs

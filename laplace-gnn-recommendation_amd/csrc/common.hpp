// Shared helpers for the gfx950 kernels behind include/laplace_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <string.h>
#include <cstring>
#include <stddef.h>
#include "../../include/laplace_hip.h"

#define MI_WAVE 64

#define MI_CHECK_ARG(cond)            \
    do {                              \
        if (!(cond)) return MI_ERR_BAD_ARG; \
    } while (0)

#define MI_HIP(expr)                  \
    do {                              \
        hipError_t _e = (expr);       \
        if (_e != hipSuccess) return (int)_e; \
    } while (0)

static inline int mi_launch_status() { return (int)hipGetLastError(); }

static inline size_t mi_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static inline bool mi_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static inline int64_t mi_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Bump allocator over a caller-provided workspace (256-byte granules).
struct MiArena {
    char*  base;
    size_t cap;
    size_t off;
    MiArena(void* p, size_t n) : base(static_cast<char*>(p)), cap(n), off(0) {}
    template <typename T>
    T* take(size_t count) {
        size_t bytes = mi_align_up(count * sizeof(T), 256);
        if (off + bytes > cap) return nullptr;
        T* r = reinterpret_cast<T*>(base + off);
        off += bytes;
        return r;
    }
};

__device__ __forceinline__ int mi_lane() { return threadIdx.x & (MI_WAVE - 1); }

__device__ __forceinline__ float mi_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, MI_WAVE);
    return v;
}

__device__ __forceinline__ float4 mi_f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

__device__ __forceinline__ void mi_f4_fma(float4& acc, float w, const float4& x) {
    acc.x = fmaf(w, x.x, acc.x);
    acc.y = fmaf(w, x.y, acc.y);
    acc.z = fmaf(w, x.z, acc.z);
    acc.w = fmaf(w, x.w, acc.w);
}

__device__ __forceinline__ float4 mi_f4_sel(bool c, float4 a, float4 b) {  // by value: no pointer select, no scratch
    return make_float4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w);
}

__device__ __forceinline__ float4 mi_f4_add(const float4& a, const float4& b) {
    return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}

__device__ __forceinline__ float4 mi_f4_shfl_xor(const float4& a, int mask) {
    return make_float4(__shfl_xor(a.x, mask, MI_WAVE), __shfl_xor(a.y, mask, MI_WAVE),
                       __shfl_xor(a.z, mask, MI_WAVE), __shfl_xor(a.w, mask, MI_WAVE));
}

// Adam update of four parameters (a9); one definition for mi_adam_dense_f32 and the SpMM optimizer epilogue,
// so that both give the same bits.  Scalars are derived in double on the host (mi_adam_consts).
struct MiAdamConsts {
    float b1, b2, omb1, omb2, step_size, bc2_sqrt, eps;
};
__host__ inline MiAdamConsts mi_adam_consts(double lr, double beta1, double beta2, double eps, int64_t step) {
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    MiAdamConsts c;
    c.b1 = (float)beta1; c.b2 = (float)beta2;
    c.omb1 = (float)(1.0 - beta1); c.omb2 = (float)(1.0 - beta2);
    c.step_size = (float)(lr / bc1);
    c.bc2_sqrt = (float)sqrt(bc2);
    c.eps = (float)eps;
    return c;
}
__device__ __forceinline__ void mi_adam_update4(float4& pp, float4 gg, float4& mm, float4& vv, bool has_w, float w,
                                                const MiAdamConsts& c) {
    if (has_w) {
        gg.x = fmaf(w, pp.x, gg.x);
        gg.y = fmaf(w, pp.y, gg.y);
        gg.z = fmaf(w, pp.z, gg.z);
        gg.w = fmaf(w, pp.w, gg.w);
    }
    // explicit roundings (no fma contraction): the same bits wherever this is inlined
#define MI_ADAM_1(f)                                                                                   \
    mm.f = __fadd_rn(__fmul_rn(c.b1, mm.f), __fmul_rn(c.omb1, gg.f));                                  \
    vv.f = __fadd_rn(__fmul_rn(c.b2, vv.f), __fmul_rn(__fmul_rn(c.omb2, gg.f), gg.f));                 \
    pp.f = __fsub_rn(pp.f, __fmul_rn(c.step_size,                                                      \
                                     __fdiv_rn(mm.f, __fadd_rn(__fdiv_rn(__fsqrt_rn(vv.f), c.bc2_sqrt), c.eps))));
    MI_ADAM_1(x) MI_ADAM_1(y) MI_ADAM_1(z) MI_ADAM_1(w)
#undef MI_ADAM_1
}

// Philox4x32-10 (Salmon et al. 2011), restated bit for bit in oracle/philox.py.
struct MiPhilox {
    uint32_t c[4];
};
__host__ __device__ __forceinline__ MiPhilox mi_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2,
                                                         uint32_t c3, uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    MiPhilox o;
    o.c[0] = c0; o.c[1] = c1; o.c[2] = c2; o.c[3] = c3;
    return o;
}

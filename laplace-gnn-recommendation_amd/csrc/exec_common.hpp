// Kernels shared by the native training-iteration executors (ranker_exec.hip, pinsage_exec.hip): Philox feature dropout
// and the multi-tensor Adam over an optimizer's parameter list.  Included inside each file's anonymous namespace.
#pragma once

// y = x * keep / (1 - p), keep ~ Bernoulli(1 - p) from Philox4x32-10 keyed on (seed, step) with counter (element / 4, site):
// four elements per draw, the same call on dY regenerates the mask in the backward.  In place allowed.
__global__ __launch_bounds__(kBlock) void dropout_kernel(int64_t n4, const float4* __restrict__ x, float4* __restrict__ y,
                                                         float p, float scale, uint32_t k0, uint32_t k1, uint32_t site,
                                                         uint32_t step_lo) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n4) return;
    const MiPhilox r = mi_philox4x32((uint32_t)i, (uint32_t)((uint64_t)i >> 32), site, step_lo, k0, k1);
    const uint32_t thr = (uint32_t)fminf(4294967040.f, p * 4294967296.f);  // keep when the draw is >= p * 2^32
    float4 v = x[i];
    v.x = r.c[0] >= thr ? v.x * scale : 0.f;
    v.y = r.c[1] >= thr ? v.y * scale : 0.f;
    v.z = r.c[2] >= thr ? v.z * scale : 0.f;
    v.w = r.c[3] >= thr ? v.w * scale : 0.f;
    y[i] = v;
}

// Two dropout launches as one (the customer-side and article-side inputs of an encoder layer): workgroups [0, split) take
// segment a, the rest segment b; per element the same draw as dropout_kernel (counter = element / 4 of ITS segment, its site).
struct DropSeg { int64_t n4; const float4* x; float4* y; uint32_t site; };
__device__ __forceinline__ void dropout_segment(int64_t n4, const float4* __restrict__ x, float4* __restrict__ y, uint32_t site,
                                                unsigned bid, float p, float scale, uint32_t k0, uint32_t k1, uint32_t step_lo) {
    const int64_t i = (int64_t)bid * kBlock + threadIdx.x;
    if (i >= n4) return;
    const MiPhilox r = mi_philox4x32((uint32_t)i, (uint32_t)((uint64_t)i >> 32), site, step_lo, k0, k1);
    const uint32_t thr = (uint32_t)fminf(4294967040.f, p * 4294967296.f);
    float4 v = x[i];
    v.x = r.c[0] >= thr ? v.x * scale : 0.f;
    v.y = r.c[1] >= thr ? v.y * scale : 0.f;
    v.z = r.c[2] >= thr ? v.z * scale : 0.f;
    v.w = r.c[3] >= thr ? v.w * scale : 0.f;
    y[i] = v;
}
__global__ __launch_bounds__(kBlock) void dropout_pair_kernel(DropSeg a, DropSeg b, unsigned split, float p, float scale, uint32_t k0,
                                                              uint32_t k1, uint32_t step_lo) {
    const bool first = blockIdx.x < split;   // fields picked one by one (a reference chosen between two argument structs goes through scratch)
    dropout_segment(first ? a.n4 : b.n4, first ? a.x : b.x, first ? a.y : b.y, first ? a.site : b.site,
                    first ? blockIdx.x : blockIdx.x - split, p, scale, k0, k1, step_lo);
}

struct AdamTable {
    mi_ranker_param p[MI_RANKER_MAX_PARAMS];
    int32_t g_stride[MI_RANKER_MAX_PARAMS];   // 1, or 4: the gradient is column 0 of a [n, 4] product (bias rows)
    float* g_dst[MI_RANKER_MAX_PARAMS];       // where a strided gradient is also written densely (the caller's .grad)
    int32_t n;
};

// blockIdx.y = parameter tensor, blockIdx.x strides over its elements; also bumps the BatchNorm batch counters
__global__ __launch_bounds__(kBlock) void adam_multi_kernel(AdamTable tb, MiAdamConsts c, int apply, int64_t* nbt0, int64_t* nbt1,
                                                            float g_scale) {
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        if (nbt0) *nbt0 += 1;
        if (nbt1) *nbt1 += 1;
    }
    const mi_ranker_param q = tb.p[blockIdx.y];
    const int gs = tb.g_stride[blockIdx.y];
    float* gd = tb.g_dst[blockIdx.y];
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < q.n; i += (int64_t)gridDim.x * kBlock) {
        float g = q.g[i * gs];
        if (gd) gd[i] = g;
        if (!apply) continue;
        if (g_scale != 1.f) g *= g_scale;   // data-parallel: the all-reduced SUM times 1 / world
        float4 pp = make_float4(q.p[i], 0.f, 0.f, 0.f), mm = make_float4(q.m[i], 0.f, 0.f, 0.f), vv = make_float4(q.v[i], 1.f, 1.f, 1.f);
        mi_adam_update4(pp, make_float4(g, 0.f, 0.f, 0.f), mm, vv, false, 0.f, c);
        q.p[i] = pp.x;
        q.m[i] = mm.x;
        q.v[i] = vv.x;
    }
}


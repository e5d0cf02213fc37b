// Probe: what does v_mfma_f32_32x32x2_f32 sustain on this part?  Register-only dependent chains (no LDS, no memory):
// W waves per SIMD, A independent accumulators per wave, N MFMAs each.  Reports TF/s against the 157.3 TF/s paper peak
// and the shader clock actually held during the run (s_memtime ticks / 100 MHz wall clock).
// hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int A>
__global__ __launch_bounds__(256) void chains(int n, float* sink, unsigned long long* clk) {
    f32x16 acc[A];
#pragma unroll
    for (int a = 0; a < A; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < n; i += 16) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int a = 0; a < A; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < A; ++a) s += acc[a][0] + acc[a][7];
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (s == 12345.f) sink[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
// the same chain with its B operand read from LDS — the inner loop of the fused top-K kernel without anything else.
// MODE 0: one ds_read_b128 per two MFMAs, components picked per half-wave (v_cndmask), compiler-scheduled
// MODE 1: all 32 chunks of the panel read up front (128 VGPRs), then the 64 MFMAs
// MODE 2: no LDS at all, but the same v_cndmask in front of every MFMA
// MODE 3: ds_read_b128 per two MFMAs, no v_cndmask (component x / z for every lane: wrong maths, same traffic)
// MODE 4: one ds_read_b32 per MFMA from a 129-float padded row (conflict-free), no v_cndmask
// MODE 5: MODE 0 with the reads pinned 3 chunks ahead by scheduling barriers
// MODE 7 / 8: MODE 4 as the kernel runs it — a fresh accumulator per panel, read when the panel is done (8: B reads pinned 8 MFMAs ahead)
// MODE 6: one ds_read_b128 per FOUR MFMAs: the half-waves read different chunks (2j / 2j + 1 of an image whose 8-float
//         groups are stored even ks first), every component feeds an MFMA directly — no v_cndmask, swizzled chunks
__device__ __forceinline__ float noise(uint32_t x) {  // ~N(0, 0.1)-like values: what embeddings look like to the multipliers
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return ((float)(x & 0xFFFFFF) / 16777216.f - 0.5f) * 0.35f;
}
template <int MODE, bool NOISE = false>
__global__ __launch_bounds__(256, 2) void chain_lds(int n_panels, float* sink, unsigned long long* clk) {
    __shared__ float4 buf[2 * 64 * 33];
    for (int i = threadIdx.x; i < 2 * 64 * 33; i += 256)
        buf[i] = NOISE ? make_float4(noise(4 * i + blockIdx.x * 77777), noise(4 * i + 1), noise(4 * i + 2), noise(4 * i + 3)) : make_float4(i * 1e-4f, 1.f, 2.f, 3.f);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool hi = lane >= 32;
    const int n_loc = (wave >> 1) * 32 + (lane & 31), sw = n_loc & 15;
    const float4* brow0 = buf + n_loc * 32;
    const float* prow0 = reinterpret_cast<const float*>(buf) + n_loc * 129 + (lane >> 5);
    float areg[64];
#pragma unroll
    for (int s = 0; s < 64; ++s) areg[s] = NOISE ? noise(threadIdx.x * 64 + s + blockIdx.x * 1315423911u) : (threadIdx.x + s) * 1e-3f;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float4 r = buf[threadIdx.x];
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int p = 0; p < n_panels; ++p) {
        const float4* brow = brow0 + (p & 1) * 2112;  // two buffers in turn: the reads cannot be hoisted out of the loop
        const float* prow = prow0 + (p & 1) * 8448;
        if (MODE == 1) {
            float4 v[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) v[i] = brow[i ^ sw];
#pragma unroll
            for (int s = 0; s < 64; ++s) {
                const float4 w = v[s >> 1];
                const float b = (s & 1) ? (hi ? w.w : w.z) : (hi ? w.y : w.x);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[s], b, acc, 0, 0, 0);
            }
        } else if (MODE == 2) {
#pragma unroll
            for (int s = 0; s < 64; ++s) {
                const float b = (s & 1) ? (hi ? r.w : r.z) : (hi ? r.y : r.x);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[s], b, acc, 0, 0, 0);
                r.x += 1.f;  // keeps the select inside the loop
            }
        } else if (MODE == 6) {
            float4 v[3];
            v[0] = brow[(0 + (hi ? 1 : 0)) ^ sw];
            v[1] = brow[(2 + (hi ? 1 : 0)) ^ sw];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (j + 2 < 16) v[(j + 2) % 3] = brow[(2 * (j + 2) + (hi ? 1 : 0)) ^ sw];
                const float4 w = v[j % 3];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[4 * j], w.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[4 * j + 1], w.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[4 * j + 2], w.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[4 * j + 3], w.w, acc, 0, 0, 0);
            }
        } else if (MODE == 7 || MODE == 8) {  // the kernel's form: fresh accumulator per panel; 8: reads pinned one group of 8 ahead
            f32x16 fresh;
#pragma unroll
            for (int i = 0; i < 16; ++i) fresh[i] = 0.f;
            if (MODE == 8) {
                float bq[2][8];
#pragma unroll
                for (int u = 0; u < 8; ++u) bq[0][u] = prow[2 * u];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 64; ++s) {
                    if (s % 8 == 0) {
                        if (s + 8 < 64) {
#pragma unroll
                            for (int u = 0; u < 8; ++u) bq[((s >> 3) + 1) & 1][u] = prow[2 * (s + 8 + u)];
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    fresh = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[s], bq[(s >> 3) & 1][s & 7], fresh, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int s = 0; s < 64; ++s) fresh = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[s], prow[2 * s], fresh, 0, 0, 0);
            }
            if (fresh[0] == 12345.678f && fresh[7] == 3.f) r.x += 1.f;  // consumed like the kernel's epilogue: needs the chain's result
        } else if (MODE == 4) {
#pragma unroll
            for (int s = 0; s < 64; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[s], prow[2 * s], acc, 0, 0, 0);
        } else {
            constexpr int kAhead = MODE == 5 ? 3 : 2;
            float4 v[kAhead + 1];
#pragma unroll
            for (int i = 0; i < kAhead; ++i) v[i] = brow[i ^ sw];
            if (MODE == 5) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 64; ++s) {
                const int i = s >> 1;
                if ((s & 1) == 0) {
                    if (i + kAhead < 32) v[(i + kAhead) % (kAhead + 1)] = brow[(i + kAhead) ^ sw];
                    if (MODE == 5) __builtin_amdgcn_sched_barrier(0);
                }
                const float4 w = v[i % (kAhead + 1)];
                const float b = MODE == 3 ? ((s & 1) ? w.z : w.x) : ((s & 1) ? (hi ? w.w : w.z) : (hi ? w.y : w.x));
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[s], b, acc, 0, 0, 0);
            }
        }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (acc[0] + acc[7] + r.x == 12345.f) sink[0] = acc[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
template <int MODE, bool NOISE = false>
void run_lds(int n_panels, int extra_lds = 0) {
    float* sink; unsigned long long* clk;
    hipMalloc(&sink, 4); hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 512;
    if (extra_lds) hipFuncSetAttribute((const void*)chain_lds<MODE, NOISE>, hipFuncAttributeMaxDynamicSharedMemorySize, extra_lds);
    int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, chain_lds<MODE, NOISE>, 256, extra_lds);
    chain_lds<MODE, NOISE><<<grid, 256, extra_lds>>>(n_panels, sink, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    chain_lds<MODE, NOISE><<<grid, 256, extra_lds>>>(n_panels, sink, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double flop = (double)grid * 4 * n_panels * 64 * 4096.0;
    printf("B-from-LDS chain%s, mode %d, LDS %d B/workgroup (runtime says %d workgroups/CU), %d panels: %.3f ms, %.1f TF/s (%.3f of 157.3), shader clock %.0f MHz\n",
           NOISE ? " on random operands" : "", MODE, (int)(2 * 64 * 33 * 16) + extra_lds, occ, n_panels, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3, (double)h[0] / ((double)h[1] / 100.0));
}
template <int A>
void run(int wgs_per_cu, int n) {
    float* sink; unsigned long long* clk;
    hipMalloc(&sink, 4); hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wgs_per_cu;
    chains<A><<<grid, 256>>>(n, sink, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    chains<A><<<grid, 256>>>(n, sink, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double flop = (double)grid * 4 * A * n * 4096.0;
    printf("waves/SIMD %d, accumulators %d, %d MFMAs each: %.3f ms, %.1f TF/s (%.3f of 157.3), shader clock %.0f MHz, %.1f cycles per MFMA per SIMD\n",
           wgs_per_cu, A, n, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3, (double)h[0] / ((double)h[1] / 100.0),
           (double)h[0] / ((double)n * A * wgs_per_cu));
}
int main() {
    run<1>(1, 1 << 16); run<2>(1, 1 << 15); run<1>(2, 1 << 16); run<2>(2, 1 << 15); run<1>(4, 1 << 15);
    run<1>(2, 1 << 19);  // ~1 s sustained
    run_lds<0>(1024); run_lds<1>(1024); run_lds<2>(1024); run_lds<3>(1024); run_lds<4>(1024); run_lds<5>(1024); run_lds<6>(1024);
    run_lds<7>(1024); run_lds<7, true>(1024); run_lds<7, true>(131); run_lds<4, true>(1024);
    return 0;
}

// Probe: does an XCD's L2 merge read misses to a line whose fill is still in flight?
// All 32 workgroups of every XCD read the SAME region in the SAME order at the same time (cold L2), once.
// If misses to pending lines merge, fabric reads ~ region size per XCD (8 copies); if not, up to 32x that.
// Variant 2: the same region read twice with a gap (second pass should hit entirely).
// hipcc --offload-arch=gfx950 -O2 tools/probes/l2_merge.hip -o /tmp/l2_merge ; rocprofv3 --pmc FETCH_SIZE ... /tmp/l2_merge
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void flush(const float4* __restrict__ p, size_t n4, float* sink) {
    float4 a = make_float4(0, 0, 0, 0);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = p[i]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    if (a.x == 12345.f) sink[0] = a.x + a.y + a.z + a.w;
}
// every workgroup reads rows [0, n_rows) of 512 B, 2 rows per wave-instruction, UNR in flight; `skew` staggers the
// workgroups of an XCD in time by starting them at different rows (wrap-around): skew = 0 -> all on the same row at once
template <int UNR>
__global__ void same_region(const float4* __restrict__ X, int n_rows, int skew, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const int li = lane & 31, half = lane >> 5;
    const int wg_in_xcd = blockIdx.x >> 3;
    float4 a = make_float4(0, 0, 0, 0);
    const int start = (wg_in_xcd * skew) % n_rows;
    for (int r0 = wave * 2 * UNR; r0 < n_rows; r0 += waves * 2 * UNR) {
        float4 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            int r = (start + r0 + 2 * u + half) % n_rows;
            v[u] = X[(size_t)r * 32 + li];
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
    }
    if (a.x == 12345.f) sink[0] = a.x + a.y + a.z + a.w;
}
int main() {
    const size_t big = (size_t)1 << 30;  // 1 GB flush buffer
    float4 *F, *X; float* s;
    hipMalloc(&F, big); hipMalloc(&X, (size_t)64 << 20); hipMalloc(&s, 4);
    hipMemset(F, 0, big); hipMemset(X, 0, (size_t)64 << 20);
    for (int mb : {1, 2, 4}) {
        const int n_rows = mb * 2048;
        for (int skew : {0, 64, 997}) {
            hipLaunchKernelGGL(flush, dim3(2048), dim3(256), 0, 0, F, big / 16, s);
            hipLaunchKernelGGL(same_region<8>, dim3(256), dim3(1024), 0, 0, X, n_rows, skew, s);
            hipDeviceSynchronize();
            printf("region %d MB skew %d done\n", mb, skew);
        }
    }
    return 0;
}

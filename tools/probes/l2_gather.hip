// Probe: an XCD's waves gather random 512-B rows inside a window of W bytes (every row ~7 times in total, by different
// waves at different moments), then all move on to the next window of the XCD's own region.  Fabric reads vs the region size.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// grid = 256 WGs x 1024 threads; XCD x = blockIdx & 7 owns rows [x * rows_per_xcd, ...); window w = rows [w * win_rows, ...)
// each wave does `per_win` gathers (2 rows per instruction, 8 in flight) per window
__global__ __launch_bounds__(1024) void gather(const float4* __restrict__ X, int rows_per_xcd, int win_rows, int per_win,
                                               int pace, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, half = lane >> 5;
    const int x = blockIdx.x & 7, wg = blockIdx.x >> 3;
    const unsigned seed = (wg * 16 + wave) * 2 + half;
    const size_t base = (size_t)x * rows_per_xcd;
    float4 a = make_float4(0, 0, 0, 0);
    const unsigned long long t0 = wall_clock64();
    const int n_win = rows_per_xcd / win_rows;
    for (int w = 0; w < n_win; ++w) {
        if (pace) while (wall_clock64() < t0 + (unsigned long long)w * pace) __builtin_amdgcn_s_sleep(4);
        for (int i = 0; i < per_win; i += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const unsigned r = hash(seed * 7919u + (w * per_win + i + u) * 104729u) % (unsigned)win_rows;
                v[u] = X[(base + (size_t)w * win_rows + r) * 32 + li];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
        }
    }
    if (a.x == 12345.f) sink[0] = a.x + a.y + a.z + a.w;
}
int main(int argc, char** argv) {
    const int rows_per_xcd = 125000 / 8 * 8;  // 64 MB per XCD, 512 MB in all: the C2 user table
    float4* X; float* s;
    hipMalloc(&X, (size_t)8 * rows_per_xcd * 512); hipMalloc(&s, 4);
    hipMemset(X, 0, (size_t)8 * rows_per_xcd * 512);
    // touches per row ~ 7.4: waves per XCD = 512 sub-streams... each of the 32*16*2 = 1024 half-waves does per_win gathers
    for (int win_rows : {512, 2048, 8192}) {
        const int per_win = (int)(7.4 * win_rows / 1024 + 0.5) < 8 ? 8 : (int)(7.4 * win_rows / 1024 + 0.5) / 8 * 8;
        for (int pace : {0, 350 * win_rows / 2048}) {
            hipLaunchKernelGGL(gather, dim3(256), dim3(1024), 0, 0, X, rows_per_xcd, win_rows, per_win, pace, s);
            hipDeviceSynchronize();
            printf("win_rows %d per_win %d pace %d\n", win_rows, per_win, pace);
        }
    }
    return 0;
}

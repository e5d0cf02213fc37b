// Probe: which XCD does workgroup b of a large 1-D grid run on?  (speed-only assumption behind the banded SpMM plan)
// hipcc --offload-arch=gfx950 -O2 tools/probes/xcc_map.hip -o /tmp/xcc_map && /tmp/xcc_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(int* xcc, int spin, float* sink) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    if (threadIdx.x == 0) xcc[blockIdx.x] = (int)(id & 0xf);
    float a = threadIdx.x;
    int n = spin * (1 + (blockIdx.x * 2654435761u >> 28));  // uneven run times, like work items of different length
    for (int i = 0; i < n; ++i) a = a * 1.0001f + 0.5f;
    if (a == 12345.f) sink[0] = a;
}
int main() {
    for (int nb : {4096, 51104, 400000}) {
        int* d; float* s;
        hipMalloc(&d, nb * sizeof(int)); hipMalloc(&s, 4);
        hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 0, 0, d, 2000, s);
        hipDeviceSynchronize();
        std::vector<int> h(nb);
        hipMemcpy(h.data(), d, nb * sizeof(int), hipMemcpyDeviceToHost);
        int off = h[0], bad = 0, hist[16] = {0};
        for (int b = 0; b < nb; ++b) { hist[h[b] & 15]++; if (h[b] != (off + b) % 8) ++bad; }
        printf("grid %d: xcc(block 0) = %d, blocks off the round-robin rule: %d (%.2f %%); per-XCD counts:", nb, off, bad, 100.0 * bad / nb);
        for (int x = 0; x < 8; ++x) printf(" %d", hist[x]);
        printf("\n");
        hipFree(d); hipFree(s);
    }
    return 0;
}

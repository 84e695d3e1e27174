// Where do the wavefronts of a workgroup land?  Launches workgroups shaped like the Horn-Schunck pipeline's
// (256 threads, 48 KB of LDS, three per CU) and prints, per wavefront index, how often it ran on each SIMD.
// Build: hipcc --offload-arch=gfx950 -O2 tools/wave_placement.hip -o tools/wave_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(256) k_probe(unsigned *out, int spin) {
    __shared__ float lds[12288];                 // 48 KB
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    lds[threadIdx.x] = (float)hw;
    __syncthreads();
    float acc = lds[(threadIdx.x * 7) & 255];
    for (int i = 0; i < spin; ++i) acc = acc * 1.0001f + 0.5f;      // keep the workgroups resident together
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = hw | (acc == 12345.f ? 1u << 31 : 0u);
}

int main() {
    const int nb = 768;
    unsigned *d;
    hipMalloc(&d, nb * 4 * sizeof(unsigned));
    k_probe<<<nb, 256>>>(d, 200000);
    std::vector<unsigned> h(nb * 4);
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
    int hist[4][4] = {};
    for (int b = 0; b < nb; ++b)
        for (int w = 0; w < 4; ++w) hist[w][(h[b * 4 + w] >> 4) & 3]++;
    for (int w = 0; w < 4; ++w) printf("wavefront %d: SIMD0 %d  SIMD1 %d  SIMD2 %d  SIMD3 %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    for (int b = 0; b < 6; ++b) {
        printf("workgroup %d:", b);
        for (int w = 0; w < 4; ++w) printf("  w%d simd %u cu %u se %u", w, (h[b * 4 + w] >> 4) & 3, (h[b * 4 + w] >> 8) & 15, (h[b * 4 + w] >> 13) & 7);
        printf("\n");
    }
    return 0;
}

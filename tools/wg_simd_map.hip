// How the hardware deals the wavefronts of workgroups that share a compute unit over its four SIMDs.
// usage: wg_simd_map <wavefronts per workgroup> <LDS bytes per workgroup> <workgroups>
// Prints, for the first compute units, every resident workgroup's SIMD per wavefront, and a histogram of the SIMD
// sequences seen.  Build: hipcc --offload-arch=gfx950 -O2 tools/wg_simd_map.hip -o tools/wg_simd_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

__global__ void k_probe(unsigned *out, int spin, int nw) {
    extern __shared__ float lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    lds[threadIdx.x] = (float)hw;
    __syncthreads();
    float acc = lds[(threadIdx.x * 7) % blockDim.x];
    for (int i = 0; i < spin; ++i) acc = acc * 1.0001f + 0.5f;      // keep the workgroups resident together
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * nw + (threadIdx.x >> 6)) * 2] = hw | (acc == 12345.f ? 1u << 31 : 0u);
        out[(blockIdx.x * nw + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    }
}

int main(int argc, char **argv) {
    const int nw = argc > 1 ? atoi(argv[1]) : 8, lds = argc > 2 ? atoi(argv[2]) : 79904, nb = argc > 3 ? atoi(argv[3]) : 504;
    unsigned *d;
    hipMalloc(&d, (size_t)nb * nw * 2 * sizeof(unsigned));
    hipFuncSetAttribute((const void *)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    k_probe<<<nb, nw * 64, lds>>>(d, 200000, nw);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    std::vector<unsigned> h((size_t)nb * nw * 2);
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> per_cu;
    std::map<std::string, int> seqs;
    for (int b = 0; b < nb; ++b) {
        std::string s;
        for (int w = 0; w < nw; ++w) s += char('0' + ((h[(b * nw + w) * 2] >> 4) & 3));
        seqs[s]++;
        const unsigned hw = h[(b * nw) * 2], xcc = h[(b * nw) * 2 + 1] & 15;
        per_cu[(xcc << 16) | (hw & 0xff00)].push_back(b);
    }
    printf("%d wavefronts per workgroup, %d B of LDS, %d workgroups on %zu compute units\n", nw, lds, nb, per_cu.size());
    for (auto &kv : seqs) printf("SIMD sequence %s: %d workgroups\n", kv.first.c_str(), kv.second);
    int shown = 0;
    for (auto &kv : per_cu) {
        if (shown++ >= 10) break;
        printf("cu %05x:", kv.first);
        for (int b : kv.second) {
            printf("  wg %4d [", b);
            for (int w = 0; w < nw; ++w) printf("%u", (h[(b * nw + w) * 2] >> 4) & 3);
            printf("]");
        }
        printf("\n");
    }
    return 0;
}

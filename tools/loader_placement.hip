// Which SIMD gets the last wavefront (the Horn-Schunck pipeline's loader) of each of the workgroups that share a
// compute unit?  Launches workgroups shaped like the pipeline's (256 threads, 40 KB of LDS, four per CU, 1000 in all)
// and prints, per compute unit, the SIMD of wavefront 3 of every resident workgroup.
// Build: hipcc --offload-arch=gfx950 -O2 tools/loader_placement.hip -o tools/loader_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

__global__ void __launch_bounds__(256) k_probe(unsigned *out, int spin) {
    __shared__ float lds[10240 - 64];            // just under 40 KB
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    lds[threadIdx.x] = (float)hw;
    __syncthreads();
    float acc = lds[(threadIdx.x * 7) & 255];
    for (int i = 0; i < spin; ++i) acc = acc * 1.0001f + 0.5f;      // keep the workgroups resident together
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw | (acc == 12345.f ? 1u << 31 : 0u);
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    }
}

int main() {
    const int nb = 1000;
    unsigned *d;
    hipMalloc(&d, nb * 8 * sizeof(unsigned));
    k_probe<<<nb, 256>>>(d, 200000);
    std::vector<unsigned> h(nb * 8);
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> per_cu;        // key: xcc, se, sh, cu  ->  (block, simd of wave 3)
    int distinct = 0;
    for (int b = 0; b < nb; ++b) {
        unsigned seen = 0;
        for (int w = 0; w < 4; ++w) seen |= 1u << ((h[(b * 4 + w) * 2] >> 4) & 3);
        distinct += seen == 15u;
        const unsigned hw = h[(b * 4 + 3) * 2], xcc = h[(b * 4 + 3) * 2 + 1] & 15;
        const unsigned key = (xcc << 16) | (hw & 0xff00);
        per_cu[key].push_back(b * 4 + ((hw >> 4) & 3));
    }
    printf("workgroups whose four wavefronts sit on four different SIMDs: %d of %d; compute units used: %zu\n", distinct, nb, per_cu.size());
    int hist_max[8] = {}, shown = 0;
    for (auto &kv : per_cu) {
        int cnt[4] = {};
        for (int v : kv.second) cnt[v & 3]++;
        int mx = 0;
        for (int s = 0; s < 4; ++s) mx = cnt[s] > mx ? cnt[s] : mx;
        hist_max[mx < 7 ? mx : 7]++;
        if (shown++ < 12) {
            printf("cu %05x:", kv.first);
            for (int v : kv.second) printf("  wg %4d loader on simd %d", v >> 2, v & 3);
            printf("\n");
        }
    }
    for (int m = 1; m < 8; ++m) printf("compute units whose busiest SIMD holds %d loaders: %d\n", m, hist_max[m]);
    return 0;
}

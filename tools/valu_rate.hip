// VALU issue-rate probe for gfx950: scalar v_fma_f32 against packed v_pk_fma_f32, 1..4 wavefronts per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o tools/valu_rate ; prints lane-FMAs per clock per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int PACKED>
__global__ void __launch_bounds__(1024) k_rate(float *out, int iters, float a, float b) {
    float r = 0.f;
    if (PACKED) {
        f2 x[8];
        for (int i = 0; i < 8; ++i) x[i] = (f2){(float)threadIdx.x + i, (float)i};
        const f2 A = {a, a}, B = {b, b};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = __builtin_elementwise_fma(x[i], A, B);
        }
        for (int i = 0; i < 8; ++i) r += x[i].x + x[i].y;
    } else {
        float x[16];
        for (int i = 0; i < 16; ++i) x[i] = (float)threadIdx.x + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(x[i], a, b);
        }
        for (int i = 0; i < 16; ++i) r += x[i];
    }
    if (r == 12345.678f) out[0] = r;
}

int main() {
    float *d; hipMalloc(&d, 4);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount; const double clk = p.clockRate * 1e3;
    printf("CUs %d clock %.0f MHz\n", ncu, clk / 1e6);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int packed = 0; packed < 2; ++packed)
        for (int wps = 1; wps <= 4; ++wps) {
            const int threads = 64 * 4 * wps;                       // one workgroup per CU, wps wavefronts per SIMD
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (packed) k_rate<1><<<ncu, threads>>>(d, iters, 1.0001f, 0.5f); else k_rate<0><<<ncu, threads>>>(d, iters, 1.0001f, 0.5f);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // instructions per wavefront: iters * 32 (both variants); lane-FMAs: scalar 64/instr, packed 128/instr
            const double instr = (double)iters * 32 * 4 * wps;      // per CU
            const double cyc = ms * 1e-3 * clk;
            printf("%s  %d waves/SIMD: %.3f ms  %.2f wave-instr/clk/CU  %.1f lane-FMA/clk/CU\n", packed ? "v_pk_fma_f32" : "v_fma_f32   ", wps,
                   ms, instr / cyc, instr * (packed ? 128 : 64) / cyc);
        }
    return 0;
}

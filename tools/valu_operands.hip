// VALU issue rate on gfx950 against the number of VGPR source operands: v_fma_f32 with 1, 2 and 3 vector sources,
// v_add_f32 with 2, v_pk_fma_f32 / v_pk_add_f32 with 3 / 2 vector sources, a lane-shifted (DPP) add; 1..4 wavefronts
// per SIMD, one workgroup per CU.  Prints wavefront-instructions per clock per CU (4 SIMDs).
// Build: hipcc -O3 --offload-arch=gfx950 tools/valu_operands.hip -o tools/valu_operands
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(X) X X X X X X X X
template <int KIND>
__global__ void __launch_bounds__(1024) k_rate(float *out, int iters, float sa, float sb) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    float y = x0 * 0.5f + 1.f, z = x0 * 0.25f + 2.f;
    f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x0}, p5 = {x3, x2}, p6 = {x5, x4}, p7 = {x7, x6};
    f2 py = {y, z}, pz = {z, y};
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {        // fma, 1 vector source
#define S(X) asm volatile("v_fma_f32 %0, %0, %1, 0.5" : "+v"(X) : "s"(sa));
            REP8(S(x0) S(x1) S(x2) S(x3) S(x4) S(x5) S(x6) S(x7))
#undef S
        } else if (KIND == 1) { // fma, 2 vector sources
#define S(X) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(X) : "v"(y), "s"(sb));
            REP8(S(x0) S(x1) S(x2) S(x3) S(x4) S(x5) S(x6) S(x7))
#undef S
        } else if (KIND == 2) { // fma, 3 vector sources
#define S(X) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(X) : "v"(y), "v"(z));
            REP8(S(x0) S(x1) S(x2) S(x3) S(x4) S(x5) S(x6) S(x7))
#undef S
        } else if (KIND == 3) { // add, 2 vector sources
#define S(X) asm volatile("v_add_f32 %0, %0, %1" : "+v"(X) : "v"(y));
            REP8(S(x0) S(x1) S(x2) S(x3) S(x4) S(x5) S(x6) S(x7))
#undef S
        } else if (KIND == 4) { // packed fma, 3 vector sources (64-bit register pairs)
#define S(X) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(X) : "v"(py), "v"(pz));
            REP8(S(p0) S(p1) S(p2) S(p3) S(p4) S(p5) S(p6) S(p7))
#undef S
        } else if (KIND == 5) { // packed add, 2 vector sources
#define S(X) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(X) : "v"(py));
            REP8(S(p0) S(p1) S(p2) S(p3) S(p4) S(p5) S(p6) S(p7))
#undef S
        } else if (KIND == 6) { // add with the first source shifted by one lane (DPP)
#define S(X) asm volatile("v_add_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(X) : "v"(y));
            REP8(S(x0) S(x1) S(x2) S(x3) S(x4) S(x5) S(x6) S(x7))
#undef S
        } else if (KIND == 8) { // fmac (32-bit encoding), vector sources
#define S(X) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(X) : "v"(y), "v"(z));
            REP8(S(x0) S(x1) S(x2) S(x3) S(x4) S(x5) S(x6) S(x7))
#undef S
        } else if (KIND == 9) { // fmac (32-bit encoding), scalar first source
#define S(X) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(X) : "s"(sa), "v"(z));
            REP8(S(x0) S(x1) S(x2) S(x3) S(x4) S(x5) S(x6) S(x7))
#undef S
        } else if (KIND == 10) { // mul (32-bit encoding), scalar first source
#define S(X) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(X) : "s"(sa));
            REP8(S(x0) S(x1) S(x2) S(x3) S(x4) S(x5) S(x6) S(x7))
#undef S
        } else if (KIND == 11) { // mul with a 32-bit literal (64-bit encoding)
#define S(X) asm volatile("v_mul_f32_e32 %0, 0x3daaaaab, %0" : "+v"(X));
            REP8(S(x0) S(x1) S(x2) S(x3) S(x4) S(x5) S(x6) S(x7))
#undef S
        } else if (KIND == 12) { // add, 64-bit (VOP3) encoding of the same operation
#define S(X) asm volatile("v_add_f32_e64 %0, %0, %1" : "+v"(X) : "v"(y));
            REP8(S(x0) S(x1) S(x2) S(x3) S(x4) S(x5) S(x6) S(x7))
#undef S
        } else if (KIND == 13) { // fma with the accumulator as third source (what fmac computes), 64-bit encoding
#define S(X) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(X) : "v"(y), "v"(z));
            REP8(S(x0) S(x1) S(x2) S(x3) S(x4) S(x5) S(x6) S(x7))
#undef S
        } else if (KIND == 7) { // packed fma, 1 vector source pair
#define S(X) asm volatile("v_pk_fma_f32 %0, %0, %1, 0.5" : "+v"(X) : "s"((double)sa));
            REP8(S(p0) S(p1) S(p2) S(p3) S(p4) S(p5) S(p6) S(p7))
#undef S
        }
    }
    float r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y;
    if (r == 12345.678f) out[0] = r;
}

template <int KIND> static void run(const char *name, float *d, int ncu, double clk) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000;
    for (int wps = 1; wps <= 4; ++wps) {
        const int threads = 64 * 4 * wps;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            k_rate<KIND><<<ncu, threads>>>(d, iters, 1.0001f, 0.5f);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        }
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double instr = (double)iters * 64 * 4 * wps, cyc = ms * 1e-3 * clk;
        printf("%-34s %d waves/SIMD: %.3f ms  %.2f wave-instr/clk/CU  (%.2f clk per instruction and SIMD)\n", name, wps, ms, instr / cyc, 4 * cyc / instr);
    }
}

int main() {
    float *d; (void)hipMalloc(&d, 4);
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount; const double clk = p.clockRate * 1e3;
    printf("CUs %d clock %.0f MHz\n", ncu, clk / 1e6);
    run<0>("v_fma_f32, 1 vector source", d, ncu, clk);
    run<1>("v_fma_f32, 2 vector sources", d, ncu, clk);
    run<2>("v_fma_f32, 3 vector sources", d, ncu, clk);
    run<3>("v_add_f32, 2 vector sources", d, ncu, clk);
    run<12>("v_add_f32_e64 (64-bit encoding)", d, ncu, clk);
    run<6>("v_add_f32 dpp wave_shr:1", d, ncu, clk);
    run<8>("v_fmac_f32_e32 v, v, v", d, ncu, clk);
    run<9>("v_fmac_f32_e32 v, s, v", d, ncu, clk);
    run<13>("v_fma_f32 v, v, v, acc (64-bit)", d, ncu, clk);
    run<10>("v_mul_f32_e32 v, s, v", d, ncu, clk);
    run<11>("v_mul_f32_e32 v, literal, v", d, ncu, clk);
    run<7>("v_pk_fma_f32, 1 vector source", d, ncu, clk);
    run<4>("v_pk_fma_f32, 3 vector sources", d, ncu, clk);
    run<5>("v_pk_add_f32, 2 vector sources", d, ncu, clk);
    return 0;
}

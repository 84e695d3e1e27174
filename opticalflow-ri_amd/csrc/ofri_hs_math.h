// Arithmetic of one Horn-Schunck Jacobi update, shared by the one-sweep-per-launch kernels and the
// temporally blocked kernel so that both produce the same bits.  Reference: src/HornSchunck.py:52-59.
#pragma once
#include <hip/hip_runtime.h>

#define HS_W12F 0.08333333581686020f   // float32(1/12)

// float32 average of the 8 neighbours with weights 1/6 (edges) and 1/12 (corners), in column form:
//   s[c] = up[c] + down[c]              vertical pair sum of column c (left, centre, right)
//   q[c] = 2*centre_row[c] + s[c]       for the left and right columns
//   avg  = (2*s[centre] + (q[left] + q[right])) / 12
// which is 5 operations per pixel once s and q are shared between horizontally adjacent pixels.
__device__ __forceinline__ float hs_q(float centre_row, float s) { return fmaf(2.0f, centre_row, s); }
__device__ __forceinline__ float hs_avg(float s_centre, float q_a, float q_b) {
    return __fmul_rn(fmaf(2.0f, s_centre, __fadd_rn(q_a, q_b)), HS_W12F);
}
// The update u' = ubar - fx (fx ubar + fy vbar + ft) / den, den = alpha^2 + fx^2 + fy^2, in the fast kernels:
// with the coefficients scaled once by s = sqrt(1/den),  A = fx s, B = fy s, C = ft s,
//   u' = ubar - A (A ubar + B vbar + C),   v' = vbar - B (A ubar + B vbar + C),
// four fused operations per pixel and three coefficients instead of five and four.  (The exact mode keeps the
// reference's own sequence of operations, ofri_hs.hip.)
__device__ __forceinline__ void hs_scaled(float fx, float fy, float ft, float a2, float &A, float &B, float &C) {
    const float s = __fsqrt_rn(__fdiv_rn(1.0f, fmaf(fy, fy, fmaf(fx, fx, a2))));
    A = __fmul_rn(fx, s); B = __fmul_rn(fy, s); C = __fmul_rn(ft, s);
}
__device__ __forceinline__ void hs_update(float ua, float va, float A, float B, float C, float &un, float &vn) {
    const float der = fmaf(A, ua, fmaf(B, va, C));
    un = fmaf(-A, der, ua);
    vn = fmaf(-B, der, va);
}

// ---- packed form ----------------------------------------------------------------------------------------
// On gfx950 a v_fma_f32 (64-bit encoding) issues once per 4.3 cycles per SIMD however many wavefronts share it, and so
// does a v_pk_fma_f32, which does two; only plain 32-bit encoded v_add / v_mul / v_fmac reach 2.5
// (tools/valu_operands.hip).  The u and v of a pixel go through the same averaging arithmetic, so the blocked kernels
// keep a pixel's (u, v) as one register pair: every step of the average and the final update is one packed
// instruction for both; the data term der = A ubar + B vbar + C mixes the two and stays scalar.  Same operations on
// the same values as the scalar functions above, so the bits are the same.
typedef float hs_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ hs_f2 hs_pk_fma(hs_f2 a, hs_f2 b, hs_f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ hs_f2 hs_pk_add(hs_f2 a, hs_f2 b) { return a + b; }
__device__ __forceinline__ hs_f2 hs_pk_mul(hs_f2 a, hs_f2 b) { return a * b; }
// float32(1/12) in both halves of a pair, from scalar registers (a packed instruction takes no literal)
__device__ __forceinline__ hs_f2 hs_w12_pair() {
    hs_f2 w;
    asm volatile("s_mov_b32 %0, 0x3daaaaab" : "=s"(w.x));
    asm volatile("s_mov_b32 %0, 0x3daaaaab" : "=s"(w.y));
    return w;
}
// s = vertical pair sum of the pixel's column (hs_avg's s), c = the pixel itself: q = 2 c + s
__device__ __forceinline__ hs_f2 hs_q_pk(hs_f2 c, hs_f2 s) { return hs_pk_fma((hs_f2){2.0f, 2.0f}, c, s); }
// the new (u, v) of a pixel from its s, the q of its left and right columns, its coefficients (A, B) and C
__device__ __forceinline__ hs_f2 hs_step_pk(hs_f2 s, hs_f2 q_a, hs_f2 q_b, hs_f2 ab, float c, hs_f2 w12) {
    const hs_f2 av = hs_pk_mul(hs_pk_fma((hs_f2){2.0f, 2.0f}, s, hs_pk_add(q_a, q_b)), w12);      // hs_avg
    const float der = fmaf(ab.x, av.x, fmaf(ab.y, av.y, c));                                       // hs_update
    // the sign goes on der (a fresh value: a source modifier), not on the coefficients, which a compiler would negate
    // once outside a sweep loop at the cost of a second set of registers; fma(-A, der, ubar) == fma(A, -der, ubar)
    return hs_pk_fma(ab, -(hs_f2){der, der}, av);
}

// The derivative cube at pixel (y, x) (reference computeDerivatives, src/HornSchunck.py:107-127).
// scipy.ndimage.convolve with a 2x2 kernel w and 'mirror' border evaluates
//   out[i,j] = w11*in[i,j] + w10*in[i,j+1] + w01*in[i+1,j] + w00*in[i+1,j+1]
// in float64 and rounds to float32; the reference then adds the two images' results in float32.
// All weights are +-0.25, so a float64 sum of the four taps scaled by 0.25 reproduces it exactly.
// P = first image, Q = second image, dense with W columns.
// The arithmetic, on the 2x2 neighbourhoods p (first image) and q (second image): [row][column] = 00, 01, 10, 11.
__device__ __forceinline__ void hs_derivs_of(double p00, double p01, double p10, double p11, double q00, double q01, double q10,
                                             double q11, float &fx, float &fy, float &ft) {
    const float pfx = (float)(0.25 * (((p00 - p01) + p10) - p11));
    const float qfx = (float)(0.25 * (((q00 - q01) + q10) - q11));
    const float pfy = (float)(0.25 * (((p00 + p01) - p10) - p11));
    const float qfy = (float)(0.25 * (((q00 + q01) - q10) - q11));
    const float qft = (float)(0.25 * (((q00 + q01) + q10) + q11));
    const float pft = (float)(-0.25 * (((p00 + p01) + p10) + p11));
    fx = __fadd_rn(pfx, qfx);
    fy = __fadd_rn(pfy, qfy);
    ft = __fadd_rn(qft, pft);
}
// partner index of the 2x2 window at the far edge ('mirror' = reflect-101)
__device__ __forceinline__ int hs_next101(int i, int n) { return i + 1 < n ? i + 1 : (n > 1 ? n - 2 : 0); }
__device__ __forceinline__ void hs_derivs_at(const float *__restrict__ P, const float *__restrict__ Q, int y, int x, int H, int W,
                                             float &fx, float &fy, float &ft) {
    const int x1 = hs_next101(x, W), y1 = hs_next101(y, H);
    const size_t r0 = (size_t)y * W, r1 = (size_t)y1 * W;
    hs_derivs_of(P[r0 + x], P[r0 + x1], P[r1 + x], P[r1 + x1], Q[r0 + x], Q[r0 + x1], Q[r1 + x], Q[r1 + x1], fx, fy, ft);
}

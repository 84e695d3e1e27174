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
// 1 / (alpha^2 + fx^2 + fy^2), correctly rounded division
__device__ __forceinline__ float hs_rden(float fx, float fy, float a2) {
    return __fdiv_rn(1.0f, fmaf(fy, fy, fmaf(fx, fx, a2)));
}
__device__ __forceinline__ void hs_update(float ua, float va, float fx, float fy, float ft, float rden,
                                          float &un, float &vn) {
    float der = __fmul_rn(fmaf(fx, ua, fmaf(fy, va, ft)), rden);
    un = fmaf(-fx, der, ua);
    vn = fmaf(-fy, der, va);
}

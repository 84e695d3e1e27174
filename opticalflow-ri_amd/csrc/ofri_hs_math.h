// Arithmetic of one Horn-Schunck Jacobi update, shared by the one-sweep-per-launch kernels and the
// temporally blocked kernel so that both produce the same bits.  Reference: src/HornSchunck.py:52-59.
#pragma once
#include <hip/hip_runtime.h>

#define HS_W6F  0.16666667163372040f   // float32(1/6)
#define HS_W12F 0.08333333581686020f   // float32(1/12)

// float32 average from the three pre-added groups: `edge` = (up+down)+(left+right),
// `diag` = (up-left + down-left) + (up-right + down-right)
__device__ __forceinline__ float hs_avg(float edge, float diag) {
    return fmaf(diag, HS_W12F, __fmul_rn(edge, HS_W6F));
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

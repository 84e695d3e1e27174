// Per-pixel arithmetic shared by the stand-alone kernels (Gaussian pre-filter, bilinear warp) and the kernel that
// runs warp, pre-filter and derivatives of a pyramid level in one pass: one definition, so both give the same bits.
#pragma once
#include "ofri_internal.h"

// Border rule of the reference's pad (gaussian_filter.py:63-66,75-78).  Left of the line the samples are
// mirrored with the edge sample repeated (-1 -> 0, -2 -> 1); right of it the reference copies the last
// K/2 samples in their ORIGINAL order (n -> n-h, n+1 -> n-h+1, ...), which equals the mirror only for K = 3.
__device__ __forceinline__ int gauss_idx(int i, int n, int h) {
    if (i < 0) return -i - 1;
    if (i >= n) return i - h;
    return i;
}

// Bilinear warp, clamp-to-edge.  The reference rounds the coordinate to the nearest integer (half to
// even), takes the neighbour on the side of the residual and weights by |residual|, all in float64
// (float32 coordinate minus int32 index promotes to float64), then stores float32.
__device__ __forceinline__ float warp_sample(const float *__restrict__ img, int H, int W, float cy, float cx) {
    float ry = rintf(cy), rx = rintf(cx);
    // int32 conversion of the rounded coordinate; coordinates this far out are clamped anyway
    int iy = (int)fminf(fmaxf(ry, -2147483000.f), 2147483000.f);
    int ix = (int)fminf(fmaxf(rx, -2147483000.f), 2147483000.f);
    double dy = (double)cy - (double)iy, dx = (double)cx - (double)ix;
    int ny = dy < 0 ? iy - 1 : iy + 1, nx = dx < 0 ? ix - 1 : ix + 1;
    dy = fabs(dy); dx = fabs(dx);
    iy = clampi(iy, 0, H - 1); ny = clampi(ny, 0, H - 1);
    ix = clampi(ix, 0, W - 1); nx = clampi(nx, 0, W - 1);
    double a = (double)img[(size_t)iy * W + ix], b = (double)img[(size_t)iy * W + nx];
    double c = (double)img[(size_t)ny * W + ix], d = (double)img[(size_t)ny * W + nx];
    double r = __dmul_rn(__dmul_rn(1.0 - dy, 1.0 - dx), a);
    r = __dadd_rn(r, __dmul_rn(__dmul_rn(1.0 - dy, dx), b));
    r = __dadd_rn(r, __dmul_rn(__dmul_rn(dy, 1.0 - dx), c));
    r = __dadd_rn(r, __dmul_rn(__dmul_rn(dy, dx), d));
    return (float)r;
}

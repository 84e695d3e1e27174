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

// clamp into [0, hi] (hi >= 0) in one instruction: the median of (i, 0, hi)
__device__ __forceinline__ int clamp0(int i, int hi) {
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(i), "v"(hi));
    return r;
}
// element `e` (0 <= e < 2^30) of a plane whose address is the same for the whole wavefront: a 32-bit byte offset on a
// scalar base, two instructions where a 64-bit address takes four
__device__ __forceinline__ float plane_at(const float *__restrict__ base, int e) {
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + ((unsigned)e << 2));
}

// Bilinear warp, clamp-to-edge.  The reference rounds the coordinate to the nearest integer (half to
// even), takes the neighbour on the side of the residual and weights by |residual|, all in float64
// (float32 coordinate minus int32 index promotes to float64), then stores float32.
struct WarpTaps { int iy, ny, ix, nx; double dy, dx; };     // rows and columns of the four samples (clamped), weights
__device__ __forceinline__ WarpTaps warp_taps(int H, int W, float cy, float cx) {
    WarpTaps t;
    float ry = rintf(cy), rx = rintf(cx);
    // int32 conversion of the rounded coordinate; coordinates this far out are clamped anyway
    int iy = (int)fminf(fmaxf(ry, -2147483000.f), 2147483000.f);
    int ix = (int)fminf(fmaxf(rx, -2147483000.f), 2147483000.f);
    double dy = (double)cy - (double)iy, dx = (double)cx - (double)ix;
    int ny = dy < 0 ? iy - 1 : iy + 1, nx = dx < 0 ? ix - 1 : ix + 1;
    t.dy = fabs(dy); t.dx = fabs(dx);
    t.iy = clamp0(iy, H - 1); t.ny = clamp0(ny, H - 1);
    t.ix = clamp0(ix, W - 1); t.nx = clamp0(nx, W - 1);
    return t;
}
// a = (iy, ix), b = (iy, nx), c = (ny, ix), d = (ny, nx)
__device__ __forceinline__ float warp_blend(const WarpTaps &t, float fa, float fb, float fc, float fd) {
    const double a = (double)fa, b = (double)fb, c = (double)fc, d = (double)fd, dy = t.dy, dx = t.dx;
    double r = __dmul_rn(__dmul_rn(1.0 - dy, 1.0 - dx), a);
    r = __dadd_rn(r, __dmul_rn(__dmul_rn(1.0 - dy, dx), b));
    r = __dadd_rn(r, __dmul_rn(__dmul_rn(dy, 1.0 - dx), c));
    r = __dadd_rn(r, __dmul_rn(__dmul_rn(dy, dx), d));
    return (float)r;
}
__device__ __forceinline__ float warp_sample(const float *__restrict__ img, int H, int W, float cy, float cx) {
    const WarpTaps t = warp_taps(H, W, cy, cx);
    return warp_blend(t, img[(size_t)t.iy * W + t.ix], img[(size_t)t.iy * W + t.nx], img[(size_t)t.ny * W + t.ix],
                      img[(size_t)t.ny * W + t.nx]);
}
// the same for planes of fewer than 2^30 pixels with at most 2^24 - 1 rows and columns (32-bit offsets, 24-bit products)
__device__ __forceinline__ float warp_sample_small(const float *__restrict__ img, int H, int W, float cy, float cx) {
    const WarpTaps t = warp_taps(H, W, cy, cx);
    const int ri = __mul24(t.iy, W), rn = __mul24(t.ny, W);
    return warp_blend(t, plane_at(img, ri + t.ix), plane_at(img, ri + t.nx), plane_at(img, rn + t.ix), plane_at(img, rn + t.nx));
}

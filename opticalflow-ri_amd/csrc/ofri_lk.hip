// Dense Lucas-Kanade with large windows.
// Reference behaviour: src/pyrlkDenseLargeW.cl:227-669 (kernel lkDense) and its host wrapper
// src/denseLucasKanade_PyCL.py:113-169.
//
// The reference spends one 8x8 work-group per pixel; here one 64-lane wavefront owns a pixel (lane =
// yid*8 + xid, each lane a 4x4 grid of window samples in registers) and four wavefronts of a 256-thread
// workgroup take four adjacent pixels that share one staged tile of the first image in LDS.  The
// per-pixel reductions are wavefront shuffles in the reference's tree order (32,16,8,4 then
// (s0+s1)+(s2+s3)), so sums round identically.  `mad` of the reference is evaluated as a fused
// multiply-add; everything else is float32 with separate roundings.  The second image is sampled with
// exact float32 bilinear interpolation (clamp to edge) instead of a hardware image sampler.
#include "ofri_internal.h"
#pragma clang fp contract(off)

#define LK_PX 4                  // pixels (wavefronts) per workgroup
#define LK_TW (34 + LK_PX - 1)   // staged tile width
#define LK_TH 34

struct LkParams {
    int win_x, win_y, half_x, half_y;
    int asym_l, asym_r, asym_t, asym_b;
    int iters;
};

__device__ __forceinline__ float lk_tree(float v, int lane) {
    v = v + __shfl_down(v, 32, 64);
    v = v + __shfl_down(v, 16, 64);
    v = v + __shfl_down(v, 8, 64);
    v = v + __shfl_down(v, 4, 64);
    float s0 = __shfl(v, 0, 64), s1 = __shfl(v, 1, 64), s2 = __shfl(v, 2, 64), s3 = __shfl(v, 3, 64);
    (void)lane;
    return (s0 + s1) + (s2 + s3);
}

// CLK_FILTER_LINEAR | CLK_ADDRESS_CLAMP_TO_EDGE with unnormalised coordinates (OpenCL 1.2, 8.2), split by axis:
// the two texel indices and the weight pair of a coordinate depend on that coordinate alone, and a lane's 4 x 4
// samples share four x and four y coordinates.
// i0, i1 are byte offsets: texel index times `scale` (4 along a row, 4 * W from row to row), so that a sample's address
// is the image base (uniform, held in scalar registers) plus one 32-bit sum.
struct LkAxis { unsigned i0, i1; float w0, w1; };      // texels and their weights (1 - frac, frac)
__device__ __forceinline__ LkAxis lk_axis(float x, int n, unsigned scale) {
    const float u = x - 0.5f, fu = floorf(u), a = u - fu;
    int i0 = (int)fminf(fmaxf(fu, -2e9f), 2e9f);
    LkAxis r;
    r.i1 = (unsigned)clampi(i0 + 1, 0, n - 1) * scale; r.i0 = (unsigned)clampi(i0, 0, n - 1) * scale;
    r.w0 = 1.f - a; r.w1 = a;
    return r;
}
__device__ __forceinline__ float lk_at(const float *__restrict__ J, unsigned byte_off) {
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(J) + byte_off);
}
// ((1-a)(1-b) t00 + a(1-b) t10 + (1-a)b t01) + ab t11, every product and sum rounded on its own
__device__ __forceinline__ float lk_sample(const float *__restrict__ J, const LkAxis &x, const LkAxis &y) {
    return (((x.w0 * y.w0) * lk_at(J, y.i0 + x.i0) + (x.w1 * y.w0) * lk_at(J, y.i0 + x.i1)) + (x.w0 * y.w1) * lk_at(J, y.i1 + x.i0)) +
           (x.w1 * y.w1) * lk_at(J, y.i1 + x.i1);
}

__global__ void __launch_bounds__(64 * LK_PX)
k_lk_dense(const float *__restrict__ I, const float *__restrict__ J, const float *__restrict__ Uin,
           const float *__restrict__ Vin, float *__restrict__ Uout, float *__restrict__ Vout, int H, int W, LkParams p) {
    __shared__ float tile[LK_TH][LK_TW + 1];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int xid = lane & 7, yid = lane >> 3;
    const int i = blockIdx.y, j0 = blockIdx.x * LK_PX, j = j0 + wv;
    // tile origin = (first pixel - halfWin - 1): 34 x 34 samples per pixel at integer positions, edge clamped
    for (int t = tid; t < LK_TH * LK_TW; t += 64 * LK_PX) {
        int ly = t / LK_TW, lx = t - ly * LK_TW;
        int yy = clampi(i - p.half_y - 1 + ly, 0, H - 1), xx = clampi(j0 - p.half_x - 1 + lx, 0, W - 1);
        tile[ly][lx] = I[(size_t)yy * W + xx];
    }
    __syncthreads();
    if (j >= W) return;                         // whole wavefront; no workgroup barrier follows
    const size_t gid = (size_t)i * W + j;

    // per-lane weights of the four sample columns and rows (.cl:339-374)
    float wxs[4], wys[4];
    wxs[0] = 1.f; wys[0] = 1.f;
    if (p.win_x >= 16) {
        wxs[1] = 1.f;
        wxs[2] = (16 + xid < p.win_x - p.asym_r) ? 1.f : 0.f;
        wxs[3] = (24 + xid < p.win_x - p.asym_r) ? 1.f : 0.f;
    } else {
        wxs[1] = (8 + xid < p.win_x - p.asym_r) ? 1.f : 0.f;
        wxs[2] = 0.f; wxs[3] = 0.f;
    }
    if (xid == 0) wxs[1] = (float)(1 - p.asym_l);
    if (p.win_y >= 16) {
        wys[1] = 1.f;
        wys[2] = (16 + yid < p.win_y - p.asym_b) ? 1.f : 0.f;
        wys[3] = (24 + yid < p.win_y - p.asym_b) ? 1.f : 0.f;
    } else {
        wys[1] = (8 + yid < p.win_y - p.asym_b) ? 1.f : 0.f;
        wys[2] = 0.f; wys[3] = 0.f;
    }
    if (yid == 0) wys[1] = (float)(1 - p.asym_t);

    float Ip[4][4], Dx[4][4], Dy[4][4], wgt[4][4];
    float A11 = 0.f, A12 = 0.f, A22 = 0.f;
#pragma unroll
    for (int ty = 0; ty < 4; ++ty)
#pragma unroll
        for (int tx = 0; tx < 4; ++tx) {
            float w = (tx && ty) ? wxs[tx] * wys[ty] : (tx ? wxs[tx] : (ty ? wys[ty] : 1.f));
            wgt[ty][tx] = w;
            const int yb = ty * 8 + yid + 1, xb = tx * 8 + xid + 1 + wv;
#define TV(dy, dx) tile[yb + (dy)][xb + (dx)]
            Ip[ty][tx] = TV(0, 0);
            float sx = ((TV(-1, 1) + TV(1, 1)) - TV(-1, -1)) - TV(1, -1);
            float dx = fmaf(sx, 3.0f, (TV(0, 1) - TV(0, -1)) * 10.0f) * w;
            float sy = ((TV(1, -1) + TV(1, 1)) - TV(-1, -1)) - TV(-1, 1);
            float dy = fmaf(sy, 3.0f, (TV(1, 0) - TV(-1, 0)) * 10.0f) * w;
#undef TV
            Dx[ty][tx] = dx; Dy[ty][tx] = dy;
            A11 = fmaf(dx, dx, A11); A12 = fmaf(dx, dy, A12); A22 = fmaf(dy, dy, A22);
        }
    A11 = lk_tree(A11, lane); A12 = lk_tree(A12, lane); A22 = lk_tree(A22, lane);
    float D = fmaf(A11, A22, -(A12 * A12));
    const float u0 = Uin[gid], v0 = Vin[gid];
    if (D < 1.192092896e-07f) {                 // flat window: the displacement is left untouched
        if (lane == 0) { Uout[gid] = u0; Vout[gid] = v0; }
        return;
    }
    A11 /= D; A12 /= D; A22 /= D;
    const float hx = (float)p.half_x, hy = (float)p.half_y;
    float ppx = ((float)j + u0) - hx, ppy = ((float)i + v0) - hy;
    float lx[4], ly[4];
    lx[0] = ppx + ((float)xid + 0.5f); ly[0] = ppy + ((float)yid + 0.5f);
#pragma unroll
    for (int t = 1; t < 4; ++t) { lx[t] = lx[t - 1] + 8.f; ly[t] = ly[t - 1] + 8.f; }
    for (int k = 0; k < p.iters; ++k) {
        if (ppx < -hx || ppx >= (float)W || ppy < -hy || ppy >= (float)H) break;
        float b1 = 0.f, b2 = 0.f;
        LkAxis ax[4], ay[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { ax[t] = lk_axis(lx[t], W, 4u); ay[t] = lk_axis(ly[t], H, 4u * (unsigned)W); }
#pragma unroll
        for (int ty = 0; ty < 4; ++ty)
#pragma unroll
            for (int tx = 0; tx < 4; ++tx) {
                float diff = (lk_sample(J, ax[tx], ay[ty]) - Ip[ty][tx]) * wgt[ty][tx];
                b1 = fmaf(diff, Dx[ty][tx], b1);
                b2 = fmaf(diff, Dy[ty][tx], b2);
            }
        b1 = lk_tree(b1, lane); b2 = lk_tree(b2, lane);
        float ddx = fmaf(A12, b2, -(A22 * b1)) * 32.0f;
        float ddy = fmaf(A12, b1, -(A11 * b2)) * 32.0f;
        ppx += ddx; ppy += ddy;
#pragma unroll
        for (int t = 0; t < 4; ++t) { lx[t] += ddx; ly[t] += ddy; }
        if (fabsf(ddx) < 0.01f && fabsf(ddy) < 0.01f) break;
    }
    if (lane == 0) {
        Uout[gid] = (ppx + hx) - (float)j;
        Vout[gid] = (ppy + hy) - (float)i;
    }
}

extern "C" int ofri_lk_compute(ofri_ctx *ctx, const float *im1, const float *im2, const float *U_in, const float *V_in,
                               float *U_out, float *V_out, int rows, int cols, int niter, int half_window,
                               int asym_left, int asym_right, int asym_top, int asym_bottom) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(im1 && im2 && U_in && V_in && U_out && V_out, "null pointer");
    OFRI_REQUIRE(rows > 0 && cols > 0 && niter >= 0, "bad shape or niter");
    OFRI_REQUIRE(half_window >= 0 && 2 * half_window + 1 <= 32, "window %d exceeds the 32-sample grid of the kernel", 2 * half_window + 1);
    OFRI_REQUIRE(rows <= 65535 && (long long)rows * cols < (1LL << 30), "field too large for one launch (32-bit byte offsets)");
    LkParams p;
    p.win_x = p.win_y = 2 * half_window + 1;
    p.half_x = p.half_y = (p.win_x - 1) >> 1;
    p.asym_l = asym_left; p.asym_r = asym_right; p.asym_t = asym_top; p.asym_b = asym_bottom;
    p.iters = niter;
    dim3 g(ceil_div(cols, LK_PX), rows);
    k_lk_dense<<<g, 64 * LK_PX, 0, ctx->stream>>>(im1, im2, U_in, V_in, U_out, V_out, rows, cols, p);
    OFRI_LAUNCH_CHECK();
    return OFRI_OK;
}

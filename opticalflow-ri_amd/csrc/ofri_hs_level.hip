// One pass from the images of a pyramid level to the coefficient planes of the Horn-Schunck pipeline: bilinear
// warp of both images by the flow so far (reference src/GenericPyramidalOpticalFlow.py:70-115, 200-201), the
// separable Gaussian pre-filter (src/gaussian_filter.py:54-90) and the derivative cube (src/HornSchunck.py:107-127),
// which the driver runs one after the other on whole images (:200-201, 374-375, 406).  A workgroup makes a
// 64 x 32 tile: the warped images of the tile and its margins go to LDS, are filtered along rows, then along
// columns, and differentiated; only the scaled coefficients A, B, C (hs_scaled) go to memory.  Every step uses the
// arithmetic of the stand-alone kernels (ofri_image_math.h, ofri_hs_math.h), so the planes are the same bits.
#include "ofri_hs_math.h"
#include "ofri_hs_stream.h"
#include "ofri_image_math.h"

#include <cstring>

#define LV_TX 64
#define LV_TY 32                         // (16 rows, more workgroups per CU: the same time; 8 rows: 12 % slower)
#define LV_MAXK 7
struct LevelTaps { float k[LV_MAXK]; };

// first, second: the images in the order computeDerivatives takes them.  The warp moves `second` back along
// half the flow and `first` forward (the driver warps im1 by -U/2 and im2 by +U/2 and calls the adapter with
// (im1, im2), which differentiates (im2, im1)).
template <int K, bool WARP>
__global__ void __launch_bounds__(256)
k_level_coefs(const float *__restrict__ first, const float *__restrict__ second, const float *__restrict__ U,
              const float *__restrict__ V, OfriSlSet o, int H, int W, float a2, LevelTaps taps) {
    constexpr int h = K / 2;
    constexpr int WR = LV_TY + 2 + 2 * h, WC = LV_TX + 2 + 2 * h;       // warped tile with margins
    constexpr int FR = LV_TY + 2, FC = LV_TX + 2;                       // filtered tile: one pixel more on every side
    __shared__ float tw[2][WR][WC + 1];                                 // [first | second]
    __shared__ float rf[2][WR][FC + 1];                                 // filtered along rows
    float (*ff)[FR][FC + 1] = reinterpret_cast<float (*)[FR][FC + 1]>(&tw[0][0][0]);   // filtered; reuses tw
    static_assert(sizeof(float) * 2 * FR * (FC + 1) <= sizeof(tw), "the filtered tile fits where the warped tile was");
    const int tid = threadIdx.x;
    const int tx0 = blockIdx.x * LV_TX, ty0 = blockIdx.y * LV_TY;
    const int oy = ty0 - 1 - h, ox = tx0 - 1 - h;                       // image coordinates of tw[.][0][0]

    // The loops below have compile-time trip counts and no early exits, so that the loads of all of a thread's
    // tile entries are in flight together (entries outside the image read a clamped address and store nothing).
    // A thread's entries are tid, tid + 256, ...: row and column of the next one follow from the last by additions.
    // Offsets are 32-bit (ofri_sl_level_ok: fewer than 2^30 pixels).
    constexpr int NE = (WR * WC + 255) / 256, DR = 256 / WC, DC = 256 % WC;
    const int er0 = tid / WC, ec0 = tid - er0 * WC;
#define LV_NEXT_ENTRY(r_, c_) { const bool wrap = (c_) + DC >= WC; (r_) += DR + (wrap ? 1 : 0); (c_) += DC - (wrap ? WC : 0); }
    if (WARP) {
        float hu[NE], hv[NE];
        int r = er0, c = ec0;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int s = __mul24(clamp0(oy + r, H - 1), W) + clamp0(ox + c, W - 1);
            hu[e] = plane_at(U, s) * 0.5f; hv[e] = plane_at(V, s) * 0.5f;
            LV_NEXT_ENTRY(r, c)
        }
        r = er0; c = ec0;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int y = oy + r, x = ox + c;
            // the coordinate index -/+ flow/2: the reference's float64 sum of an int32 and a float32, rounded to float32, is the
            // float32 sum (both operands are float32 values, and 53 bits are more than twice 24 plus 2: the second rounding
            // cannot move the result)
            const float fy = (float)y, fx = (float)x;
            const float v1 = warp_sample_small(second, H, W, __fsub_rn(fy, hv[e]), __fsub_rn(fx, hu[e]));
            const float v0 = warp_sample_small(first, H, W, __fadd_rn(fy, hv[e]), __fadd_rn(fx, hu[e]));
            if (r < WR && y >= 0 && y < H && x >= 0 && x < W) { tw[0][r][c] = v0; tw[1][r][c] = v1; }
            LV_NEXT_ENTRY(r, c)
        }
    } else {
        float v0[NE], v1[NE];
        int r = er0, c = ec0;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int s = __mul24(clamp0(oy + r, H - 1), W) + clamp0(ox + c, W - 1);
            v0[e] = plane_at(first, s); v1[e] = plane_at(second, s);
            LV_NEXT_ENTRY(r, c)
        }
        r = er0; c = ec0;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int y = oy + r, x = ox + c;
            if (r < WR && y >= 0 && y < H && x >= 0 && x < W) { tw[0][r][c] = v0[e]; tw[1][r][c] = v1[e]; }
            LV_NEXT_ENTRY(r, c)
        }
    }
#undef LV_NEXT_ENTRY
    __syncthreads();
    // Tiles whose margins lie inside the image (all but the outermost ring) need no border rule and no bounds test: the
    // taps are neighbours in the tile.  Same operations in the same order as the general path below.
    const bool inner = ox >= 0 && oy >= 0 && ox + WC <= W && oy + WR <= H && tx0 + LV_TX + 1 < W && ty0 + LV_TY + 1 < H;
    if (inner) {
        // entries tid, tid + 256, ... of a [rows][FC] domain: row and column by additions, as above
        constexpr int FDR = 256 / FC, FDC = 256 % FC;
        {
            constexpr int NX = (WR * FC + 255) / 256;
            int r = tid / FC, c = tid - r * FC;
#pragma unroll
            for (int e = 0; e < NX; ++e) {
                if (r < WR) {
                    float a0 = 0.f, a1 = 0.f;
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        a0 = __fadd_rn(a0, __fmul_rn(tw[0][r][c + 2 * h - j], taps.k[j]));
                        a1 = __fadd_rn(a1, __fmul_rn(tw[1][r][c + 2 * h - j], taps.k[j]));
                    }
                    rf[0][r][c] = a0; rf[1][r][c] = a1;
                }
                const bool wrap = c + FDC >= FC;
                r += FDR + (wrap ? 1 : 0); c += FDC - (wrap ? FC : 0);
            }
        }
        __syncthreads();
        {
            constexpr int NY = (FR * FC + 255) / 256;
            int r = tid / FC, c = tid - r * FC;
#pragma unroll
            for (int e = 0; e < NY; ++e) {
                if (r < FR) {
                    float a0 = 0.f, a1 = 0.f;
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        a0 = __fadd_rn(a0, __fmul_rn(rf[0][r + 2 * h - j][c], taps.k[j]));
                        a1 = __fadd_rn(a1, __fmul_rn(rf[1][r + 2 * h - j][c], taps.k[j]));
                    }
                    ff[0][r][c] = a0; ff[1][r][c] = a1;
                }
                const bool wrap = c + FDC >= FC;
                r += FDR + (wrap ? 1 : 0); c += FDC - (wrap ? FC : 0);
            }
        }
        __syncthreads();
        // a thread's pixels: column tid % 64, LV_TY / 4 rows one below the other: the lower row of a pixel's window is the
        // upper row of the next pixel's
        constexpr int PR = LV_TY / 4;
        const int c0 = (tid & 63) + 1, rb = (tid >> 6) * PR;
        float p0 = ff[0][rb + 1][c0], p1 = ff[0][rb + 1][c0 + 1], q0 = ff[1][rb + 1][c0], q1 = ff[1][rb + 1][c0 + 1];
        int s = __mul24(ty0 + rb, W) + tx0 + (tid & 63);
#pragma unroll
        for (int k = 0; k < PR; ++k) {
            const int r1 = rb + k + 2;
            const float p2 = ff[0][r1][c0], p3 = ff[0][r1][c0 + 1], q2 = ff[1][r1][c0], q3 = ff[1][r1][c0 + 1];
            float gx, gy, gt;
            hs_derivs_of(p0, p1, p2, p3, q0, q1, q2, q3, gx, gy, gt);
            float A, B, C;
            hs_scaled(gx, gy, gt, a2, A, B, C);
            *reinterpret_cast<float *>(reinterpret_cast<char *>(o.fx) + ((unsigned)s << 2)) = A;
            *reinterpret_cast<float *>(reinterpret_cast<char *>(o.fy) + ((unsigned)s << 2)) = B;
            *reinterpret_cast<float *>(reinterpret_cast<char *>(o.ft) + ((unsigned)s << 2)) = C;
            p0 = p2; p1 = p3; q0 = q2; q1 = q3;
            s += W;
        }
        return;
    }
    for (int idx = tid; idx < WR * FC; idx += 256) {
        const int r = idx / FC, c = idx - r * FC, y = oy + r, x = tx0 - 1 + c;
        if (y < 0 || y >= H || x < 0 || x >= W) continue;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int sc = gauss_idx(x + h - j, W, h) - ox;
            a0 = __fadd_rn(a0, __fmul_rn(tw[0][r][sc], taps.k[j]));
            a1 = __fadd_rn(a1, __fmul_rn(tw[1][r][sc], taps.k[j]));
        }
        rf[0][r][c] = a0; rf[1][r][c] = a1;
    }
    __syncthreads();
    for (int idx = tid; idx < FR * FC; idx += 256) {
        const int r = idx / FC, c = idx - r * FC, y = ty0 - 1 + r, x = tx0 - 1 + c;
        if (y < 0 || y >= H || x < 0 || x >= W) continue;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int sr = gauss_idx(y + h - j, H, h) - oy;
            a0 = __fadd_rn(a0, __fmul_rn(rf[0][sr][c], taps.k[j]));
            a1 = __fadd_rn(a1, __fmul_rn(rf[1][sr][c], taps.k[j]));
        }
        ff[0][r][c] = a0; ff[1][r][c] = a1;
    }
    __syncthreads();
    for (int idx = tid; idx < LV_TY * LV_TX; idx += 256) {
        const int r = idx / LV_TX, c = idx - r * LV_TX, y = ty0 + r, x = tx0 + c;
        if (y >= H || x >= W) continue;
        const int r0 = r + 1, c0 = c + 1, r1 = hs_next101(y, H) - (ty0 - 1), c1 = hs_next101(x, W) - (tx0 - 1);
        float gx, gy, gt;
        hs_derivs_of(ff[0][r0][c0], ff[0][r0][c1], ff[0][r1][c0], ff[0][r1][c1], ff[1][r0][c0], ff[1][r0][c1], ff[1][r1][c0],
                     ff[1][r1][c1], gx, gy, gt);
        const size_t s = (size_t)y * W + x;
        hs_scaled(gx, gy, gt, a2, o.fx[s], o.fy[s], o.ft[s]);
    }
}

// the kernel is built for pre-filters of 3, 5 and 7 taps and addresses its planes with 32-bit offsets
bool ofri_sl_level_ok(int ntaps, int H, int W) {
    return (ntaps == 3 || ntaps == 5 || ntaps == 7) && (long long)H * W < (1LL << 30) && H < (1 << 24) && W < (1 << 24);
}

// Coefficient planes of the set from the level's images: warp by (U, V) if given, pre-filter with ntaps taps
// (3, 5 or 7), differentiate.  The set then stands for a zero initial flow.
int ofri_sl_fill_level(ofri_ctx *ctx, OfriSlSet &s, const float *first, const float *second, const float *U, const float *V,
                       const float *taps_host, int ntaps, float a2, int H, int W) {
    OFRI_REQUIRE(ofri_sl_level_ok(ntaps, H, W), "pre-filter of %d taps on %d x %d pixels is not built into the level kernel", ntaps, H, W);
    OFRI_REQUIRE(ntaps / 2 <= H && ntaps / 2 <= W, "image smaller than half the kernel");
    LevelTaps t;
    memset(&t, 0, sizeof(t));
    memcpy(t.k, taps_host, ntaps * sizeof(float));
    s.zero_in = 1;
    dim3 g(ceil_div(W, LV_TX), ceil_div(H, LV_TY));
#define LV_LAUNCH(K_)                                                                                              \
    if (U) k_level_coefs<K_, true><<<g, 256, 0, ctx->stream>>>(first, second, U, V, s, H, W, a2, t);             \
    else k_level_coefs<K_, false><<<g, 256, 0, ctx->stream>>>(first, second, nullptr, nullptr, s, H, W, a2, t);
    if (ntaps == 3) { LV_LAUNCH(3) } else if (ntaps == 5) { LV_LAUNCH(5) } else { LV_LAUNCH(7) }
#undef LV_LAUNCH
    OFRI_LAUNCH_CHECK();
    return OFRI_OK;
}

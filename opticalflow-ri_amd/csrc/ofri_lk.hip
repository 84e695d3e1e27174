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

// One pixel (i, j) by one wavefront.  tile: the staged first image, tile[ly][lx] = I at (i - half_y - 1 + ly,
// j0 - half_x - 1 + lx) edge clamped, where j0 = j - wv (several pixels of a row may share a tile).
template <int TWP>
__device__ __forceinline__ void lk_pixel(const float (*tile)[TWP], const float *__restrict__ J, const float *__restrict__ Uin,
                                         const float *__restrict__ Vin, float *__restrict__ Uout, float *__restrict__ Vout, int H, int W,
                                         const LkParams &p, int i, int j, int wv, int lane) {
    const int xid = lane & 7, yid = lane >> 3;
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

__global__ void __launch_bounds__(64 * LK_PX)
k_lk_dense(const float *__restrict__ I, const float *__restrict__ J, const float *__restrict__ Uin,
           const float *__restrict__ Vin, float *__restrict__ Uout, float *__restrict__ Vout, int H, int W, LkParams p) {
    __shared__ float tile[LK_TH][LK_TW + 1];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i = blockIdx.y, j0 = blockIdx.x * LK_PX, j = j0 + wv;
    // tile origin = (first pixel - halfWin - 1): 34 x 34 samples per pixel at integer positions, edge clamped
    for (int t = tid; t < LK_TH * LK_TW; t += 64 * LK_PX) {
        int ly = t / LK_TW, lx = t - ly * LK_TW;
        int yy = clampi(i - p.half_y - 1 + ly, 0, H - 1), xx = clampi(j0 - p.half_x - 1 + lx, 0, W - 1);
        tile[ly][lx] = I[(size_t)yy * W + xx];
    }
    __syncthreads();
    if (j >= W) return;                         // whole wavefront; no workgroup barrier follows
    lk_pixel<LK_TW + 1>(tile, J, Uin, Vin, Uout, Vout, H, W, p, i, j, wv, lane);
}

// The pixels of a list (indices i * W + j, *count of them), one per wavefront with a tile of its own: those that the
// tile kernel below could not finish inside its staged neighbourhood of the second image.
__global__ void __launch_bounds__(64 * LK_PX)
k_lk_list(const float *__restrict__ I, const float *__restrict__ J, const float *__restrict__ Uin, const float *__restrict__ Vin,
          float *__restrict__ Uout, float *__restrict__ Vout, int H, int W, LkParams p, const unsigned *__restrict__ list,
          const unsigned *__restrict__ count) {
    __shared__ float tile[LK_PX][LK_TH][LK_TH + 1];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const unsigned n = *count;
    // every wavefront of the workgroup makes the same number of rounds, so the barriers match
    for (unsigned base = blockIdx.x * LK_PX; base < n; base += gridDim.x * LK_PX) {
        const unsigned e = base + wv;
        const bool have = e < n;
        const unsigned gid = have ? list[e] : 0u;
        const int i = (int)(gid / (unsigned)W), j = (int)(gid - (unsigned)i * (unsigned)W);
        __syncthreads();                        // the round before is done with the tiles
        if (have)
            for (int t = lane; t < LK_TH * LK_TH; t += 64) {
                int ly = t / LK_TH, lx = t - ly * LK_TH;
                int yy = clampi(i - p.half_y - 1 + ly, 0, H - 1), xx = clampi(j - p.half_x - 1 + lx, 0, W - 1);
                tile[wv][ly][lx] = I[(size_t)yy * W + xx];
            }
        __syncthreads();
        if (have) lk_pixel<LK_TH + 1>(tile[wv], J, Uin, Vin, Uout, Vout, H, W, p, i, j, 0, lane);
    }
}

// ---- tile of pixels (fast mode; windows of 13, 27 and 31 samples, with or without the asymmetry flags) ------------------
// A 256-thread workgroup owns 32 x 16 pixels, a lane two vertically adjacent ones.  What neighbouring pixels share is
// made once per tile in LDS: the first image around the tile (for the 27 x 27 window 58 x 42 window samples: a pixel's
// sample (oy, ox) is the image at pixel - half window + (oy, ox), edge clamped), its Scharr gradients, and the second image
// around the tile displaced by the tile's central displacement, with a margin of LKT_R pixels for the other pixels'
// displacements and the Newton steps (edge clamped as well, so the sampler's clamp-to-edge needs no test in the loop).  A
// lane then walks its pixels' window samples serially - no reduction across lanes: sums A11, A12, A22 first, then up to
// `iters` Newton steps, each b = sum over the window of (J(sample + displacement) - I(sample)) * (Dx, Dy).  All samples of a
// pixel share one pair of bilinear weights (the displacement's fractions), so J is interpolated separably: along x as the
// row is read (each texel is read once per row), along y against the row before, kept in WIN registers per pixel.  The two
// pixels of a lane share every read of I, Dx, Dy (pixel 1's sample row oy is pixel 0's row oy + 1).  A pixel whose displaced
// window leaves the staged neighbourhood is put on a list and done afterwards by k_lk_list from its input displacement.
// Sums run in another order than the reference's lane tree and the interpolation is factored, so results agree with the
// exact kernel to rounding (about 1e-5 px), not bit for bit.
// ASYM: the vorticity-adaptive windows (src/pyrlkDenseLargeW.cl:339-374).  The reference's per-lane weights are 0 or 1 and
// amount to: sample column 8 counts 1 - asym_left, the last column 1 - asym_right, likewise rows 8 and the last with
// asym_top / asym_bottom (its "xid == 0" test picks the lane that owns column 8, not column 0).  A weight w multiplies the
// gradients and the difference, so a sample counts w * w = w times.
#define LKT_PX 32
#define LKT_PY 16
#define LKT_R 6
template <int WIN> struct LktCfg {
    static constexpr int HW = (WIN - 1) / 2;
    static constexpr int SW = LKT_PX + WIN - 1;          // sample columns (27: 58)
    static constexpr int SH = LKT_PY + WIN - 1;          // sample rows (27: 42)
    static constexpr int SP = SW + 2;                    // their row pitch
    static constexpr int JW = SW + 1 + 2 * LKT_R;        // columns of the second image (27: 71)
    static constexpr int JH = SH + 1 + 2 * LKT_R;        // rows (27: 55)
    static constexpr int JP = JW + 2;                    // their row pitch
    static constexpr int LDS = (3 * SH * SP + JH * JP) * 4;
    static_assert((SH + 2) * (SW + 2) <= JH * JP, "the first image with its gradient apron is staged where the second image goes later");
    static_assert(3 * LDS + 3 * 64 <= 160 * 1024, "three workgroups per CU");
};
#define LKT_HW 13                                       // the default window's half width

__device__ __forceinline__ int lk_to_int(float f) { return (int)fminf(fmaxf(f, -1.0e9f), 1.0e9f); }

struct LktPix {                  // Newton state of one pixel
    float ppx, ppy, A11, A12, A22, b1, b2, ax, ay;
    int joff;                    // offset of the window's first texel in the staged second image; < 0: not iterating
    bool live;                   // still iterating
};
struct LktAsym { float wl, wr, wt, wb; };        // weights of sample column 8, the last column, row 8, the last row

template <int WIN, bool ASYM>
__global__ void __launch_bounds__(256, 3)
k_lk_tile(const float *__restrict__ I, const float *__restrict__ J, const float *__restrict__ Uin, const float *__restrict__ Vin,
          float *__restrict__ Uout, float *__restrict__ Vout, int H, int W, int iters, int tiles_x, unsigned *__restrict__ list,
          unsigned *__restrict__ count, LktAsym as) {
    using C = LktCfg<WIN>;
    constexpr int HW = C::HW, SW = C::SW, SH = C::SH, SP = C::SP, JW = C::JW, JH = C::JH, JP = C::JP;
    __shared__ float sI[SH][SP], sDx[SH][SP], sDy[SH][SP];
    __shared__ float sJ[JH * JP];
    __shared__ int centre[2];
    const int tid = threadIdx.x;
    const int tx0 = (blockIdx.x % tiles_x) * LKT_PX, ty0 = (blockIdx.x / tiles_x) * LKT_PY;
    // weight of a sample column (compile-time index) / row
    auto wcol = [&](int ox) { return ox == 8 ? as.wl : (ox == WIN - 1 ? as.wr : 1.0f); };
    auto wrow = [&](int oy) { return oy == 8 ? as.wt : (oy == WIN - 1 ? as.wb : 1.0f); };
    // ---- first image with a one-pixel apron (in sJ's space), then gradients
    {
        float (*tI)[SW + 2] = reinterpret_cast<float (*)[SW + 2]>(sJ);
        // a thread's loads are issued together (one per loop round would cost a memory latency each)
        constexpr int NI = ((SH + 2) * (SW + 2) + 255) / 256;
        float pre[NI];
#pragma unroll
        for (int e = 0; e < NI; ++e) {
            const int t = min(tid + 256 * e, (SH + 2) * (SW + 2) - 1), ly = t / (SW + 2), lx = t - ly * (SW + 2);
            const int yy = clampi(ty0 - HW - 1 + ly, 0, H - 1), xx = clampi(tx0 - HW - 1 + lx, 0, W - 1);
            pre[e] = I[(size_t)yy * W + xx];
        }
#pragma unroll
        for (int e = 0; e < NI; ++e) {
            const int t = tid + 256 * e;
            if (t < (SH + 2) * (SW + 2)) sJ[t] = pre[e];           // tI[ly][lx] is sJ[t]
        }
        if (tid == 0) {         // the tile's central pixel (clamped into the image) gives the neighbourhood's displacement
            const size_t c = (size_t)min(ty0 + LKT_PY / 2, H - 1) * W + min(tx0 + LKT_PX / 2, W - 1);
            centre[0] = clampi(lk_to_int(floorf(Uin[c])), -(1 << 24), 1 << 24);
            centre[1] = clampi(lk_to_int(floorf(Vin[c])), -(1 << 24), 1 << 24);
        }
        __syncthreads();
        for (int t = tid; t < SH * SW; t += 256) {
            const int sy = t / SW, sx = t - sy * SW;
#define TV(dy, dx) tI[sy + 1 + (dy)][sx + 1 + (dx)]
            const float gx = ((TV(-1, 1) + TV(1, 1)) - TV(-1, -1)) - TV(1, -1);
            const float gy = ((TV(1, -1) + TV(1, 1)) - TV(-1, -1)) - TV(-1, 1);
            sDx[sy][sx] = fmaf(gx, 3.0f, (TV(0, 1) - TV(0, -1)) * 10.0f);
            sDy[sy][sx] = fmaf(gy, 3.0f, (TV(1, 0) - TV(-1, 0)) * 10.0f);
            sI[sy][sx] = TV(0, 0);
#undef TV
        }
        __syncthreads();
    }
    // ---- second image around the displaced tile
    const int jx0 = tx0 - HW - LKT_R + centre[0], jy0 = ty0 - HW - LKT_R + centre[1];
    {
        constexpr int NB = 8, NJ = ((JH * JW + 255) / 256 + NB - 1) / NB * NB;          // in batches of eight loads
#pragma unroll
        for (int b0 = 0; b0 < NJ; b0 += NB) {
            float pre[NB];
#pragma unroll
            for (int e = 0; e < NB; ++e) {
                const int t = min(tid + 256 * (b0 + e), JH * JW - 1), ly = t / JW, lx = t - ly * JW;
                const int yy = clampi(jy0 + ly, 0, H - 1), xx = clampi(jx0 + lx, 0, W - 1);
                pre[e] = J[(size_t)yy * W + xx];
            }
#pragma unroll
            for (int e = 0; e < NB; ++e) {
                const int t = tid + 256 * (b0 + e), ly = t / JW, lx = t - ly * JW;
                if (t < JH * JW) sJ[ly * JP + lx] = pre[e];
            }
        }
    }
    __syncthreads();

    // ---- this lane's two pixels: column px, rows py and py + 1 of the tile
    const int px = tid & 31, py = (tid >> 5) * 2;
    const int gx = tx0 + px, gy = ty0 + py;
    const bool in0 = gx < W && gy < H, in1 = gx < W && gy + 1 < H;
    const float *pI = &sI[py][px], *pDx = &sDx[py][px], *pDy = &sDy[py][px];
    LktPix q[2];
    q[0].A11 = q[0].A12 = q[0].A22 = q[1].A11 = q[1].A12 = q[1].A22 = 0.f;
#pragma unroll 1
    for (int sy = 0; sy <= WIN; ++sy) {
        const bool use0 = sy < WIN, use1 = sy > 0;      // uniform
        const float wy0 = ASYM ? wrow(sy) : 1.0f, wy1 = ASYM ? wrow(sy - 1) : 1.0f;
#pragma unroll
        for (int ox = 0; ox < WIN; ++ox) {
            float dx = pDx[sy * SP + ox], dy = pDy[sy * SP + ox];
            if (ASYM && (ox == 8 || ox == WIN - 1)) { dx *= wcol(ox); dy *= wcol(ox); }
            if (ASYM) {
                const float dx0 = dx * wy0, dy0 = dy * wy0, dx1 = dx * wy1, dy1 = dy * wy1;
                if (use0) { q[0].A11 = fmaf(dx0, dx0, q[0].A11); q[0].A12 = fmaf(dx0, dy0, q[0].A12); q[0].A22 = fmaf(dy0, dy0, q[0].A22); }
                if (use1) { q[1].A11 = fmaf(dx1, dx1, q[1].A11); q[1].A12 = fmaf(dx1, dy1, q[1].A12); q[1].A22 = fmaf(dy1, dy1, q[1].A22); }
            } else {
                if (use0) { q[0].A11 = fmaf(dx, dx, q[0].A11); q[0].A12 = fmaf(dx, dy, q[0].A12); q[0].A22 = fmaf(dy, dy, q[0].A22); }
                if (use1) { q[1].A11 = fmaf(dx, dx, q[1].A11); q[1].A12 = fmaf(dx, dy, q[1].A12); q[1].A22 = fmaf(dy, dy, q[1].A22); }
            }
        }
    }
    const float hw = (float)HW;
    float u0[2], v0[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const bool in = k ? in1 : in0;
        const size_t gid = (size_t)(gy + k) * W + gx;
        u0[k] = in ? Uin[gid] : 0.f; v0[k] = in ? Vin[gid] : 0.f;
        const float D = fmaf(q[k].A11, q[k].A22, -(q[k].A12 * q[k].A12));
        q[k].live = in && !(D < 1.192092896e-07f);
        if (in && !q[k].live) { Uout[gid] = u0[k]; Vout[gid] = v0[k]; }         // flat window: the displacement is left untouched
        q[k].A11 /= D; q[k].A12 /= D; q[k].A22 /= D;
        q[k].ppx = ((float)gx + u0[k]) - hw; q[k].ppy = ((float)(gy + k) + v0[k]) - hw;
        q[k].joff = -1;
    }
    bool finished[2] = {q[0].live, q[1].live};      // pixels whose result this kernel stores (not deferred, not flat)
    for (int it = 0; it < iters; ++it) {
        // where each pixel's window lies in the staged second image; leave the loop / defer as needed
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (!q[k].live) continue;
            if (q[k].ppx < -hw || q[k].ppx >= (float)W || q[k].ppy < -hw || q[k].ppy >= (float)H) { q[k].live = false; continue; }
            const float fx = floorf(q[k].ppx), fy = floorf(q[k].ppy);
            q[k].ax = q[k].ppx - fx; q[k].ay = q[k].ppy - fy;
            const int rx = lk_to_int(fx) - jx0, ry = lk_to_int(fy) - jy0;
            if (rx < 0 || ry < 0 || rx + WIN >= JW || ry + WIN >= JH) {        // outside the neighbourhood: deferred
                q[k].live = false; finished[k] = false;
                list[atomicAdd(count, 1u)] = (unsigned)((size_t)(gy + k) * W + gx);
                continue;
            }
            q[k].joff = ry * JP + rx;
            q[k].b1 = q[k].b2 = 0.f;
        }
        if (__builtin_amdgcn_ballot_w64(q[0].live || q[1].live) == 0) break;
        // lanes with nothing to do walk a harmless window (offset 0) and drop the result
        const float *j0p = sJ + (q[0].live ? q[0].joff : 0), *j1p = sJ + (q[1].live ? q[1].joff : 0);
        const float ax0 = q[0].ax, ay0 = q[0].ay, ax1 = q[1].ax, ay1 = q[1].ay;
        float h0[WIN], h1[WIN], b10 = 0.f, b20 = 0.f, b11 = 0.f, b21 = 0.f;
        {   // row 0 of both windows, interpolated along x
            float a = j0p[0], b = j1p[0];
#pragma unroll
            for (int ox = 0; ox < WIN; ++ox) {
                const float an = j0p[ox + 1], bn = j1p[ox + 1];
                h0[ox] = fmaf(ax0, an - a, a); h1[ox] = fmaf(ax1, bn - b, b);
                a = an; b = bn;
            }
        }
        // pixel 0 takes sample rows 0..WIN-1 of the lane, pixel 1 rows 1..WIN; both advance one row of J per round
#define LKT_TAP(HP, JP_, AX, AY, B1, B2, JOLD, WR)                                               \
        {                                                                                        \
            const float jn = JP_[ox + 1];                                                        \
            const float h = fmaf(AX, jn - JOLD, JOLD);                                           \
            const float ji = fmaf(AY, h - HP[ox], HP[ox]);                                       \
            float df = ji - vi;                                                                  \
            if (ASYM) df *= (ox == 8 || ox == WIN - 1) ? WR * wcol(ox) : WR;                     \
            B1 = fmaf(df, dx, B1); B2 = fmaf(df, dy, B2);                                        \
            HP[ox] = h; JOLD = jn;                                                               \
        }
        {   // sample row 0: pixel 0 only
            j0p += JP;
            float a = j0p[0];
            const float w0 = ASYM ? wrow(0) : 1.0f;
#pragma unroll
            for (int ox = 0; ox < WIN; ++ox) {
                const float vi = pI[ox], dx = pDx[ox], dy = pDy[ox];
                LKT_TAP(h0, j0p, ax0, ay0, b10, b20, a, w0)
            }
        }
#pragma unroll 1
        for (int sy = 1; sy < WIN; ++sy) {
            j0p += JP; j1p += JP;
            float a = j0p[0], b = j1p[0];
            const float w0 = ASYM ? wrow(sy) : 1.0f, w1 = ASYM ? wrow(sy - 1) : 1.0f;
#pragma unroll
            for (int ox = 0; ox < WIN; ++ox) {
                const float vi = pI[sy * SP + ox], dx = pDx[sy * SP + ox], dy = pDy[sy * SP + ox];
                LKT_TAP(h0, j0p, ax0, ay0, b10, b20, a, w0)
                LKT_TAP(h1, j1p, ax1, ay1, b11, b21, b, w1)
            }
        }
        {   // sample row WIN: pixel 1 only
            j1p += JP;
            float b = j1p[0];
            const float w1 = ASYM ? wrow(WIN - 1) : 1.0f;
#pragma unroll
            for (int ox = 0; ox < WIN; ++ox) {
                const float vi = pI[WIN * SP + ox], dx = pDx[WIN * SP + ox], dy = pDy[WIN * SP + ox];
                LKT_TAP(h1, j1p, ax1, ay1, b11, b21, b, w1)
            }
        }
#undef LKT_TAP
        q[0].b1 = b10; q[0].b2 = b20; q[1].b1 = b11; q[1].b2 = b21;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (!q[k].live) continue;
            const float ddx = fmaf(q[k].A12, q[k].b2, -(q[k].A22 * q[k].b1)) * 32.0f;
            const float ddy = fmaf(q[k].A12, q[k].b1, -(q[k].A11 * q[k].b2)) * 32.0f;
            q[k].ppx += ddx; q[k].ppy += ddy;
            if (fabsf(ddx) < 0.01f && fabsf(ddy) < 0.01f) q[k].live = false;
        }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k)
        if (finished[k]) {
            const size_t gid = (size_t)(gy + k) * W + gx;
            Uout[gid] = (q[k].ppx + hw) - (float)gx;
            Vout[gid] = (q[k].ppy + hw) - (float)(gy + k);
        }
}

template <int WIN>
static void lkt_launch(ofri_ctx *ctx, const float *im1, const float *im2, const float *U_in, const float *V_in, float *U_out, float *V_out,
                       int rows, int cols, int niter, unsigned *list, unsigned *count, bool asym, const LktAsym &as) {
    const int tiles_x = ceil_div(cols, LKT_PX), tiles = tiles_x * ceil_div(rows, LKT_PY);
    if (asym) k_lk_tile<WIN, true><<<tiles, 256, 0, ctx->stream>>>(im1, im2, U_in, V_in, U_out, V_out, rows, cols, niter, tiles_x, list, count, as);
    else k_lk_tile<WIN, false><<<tiles, 256, 0, ctx->stream>>>(im1, im2, U_in, V_in, U_out, V_out, rows, cols, niter, tiles_x, list, count, as);
}

extern "C" int ofri_lk_set_mode(ofri_ctx *ctx, int mode) {
    OFRI_REQUIRE(ctx, "null context");
    OFRI_REQUIRE(mode == 0 || mode == -1, "mode: 0 fast (tile of pixels where the window allows), -1 exact (one wavefront per pixel)");
    ctx->lk_mode = mode;
    return OFRI_OK;
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
    // the tile kernel is built for the windows of 13, 27 (the reference's default) and 31 samples; the flags are 0 or 1
    const bool flags01 = ((asym_left | asym_right | asym_top | asym_bottom) & ~1) == 0;
    const bool tile_ok = ctx->lk_mode == 0 && (half_window == 6 || half_window == LKT_HW || half_window == 15) && flags01 &&
                         U_out != U_in && V_out != V_in && U_out != V_in && V_out != U_in;
    if (tile_ok) {
        // fast mode: tiles of pixels; the pixels it defers (displaced window outside the staged neighbourhood) follow
        const size_t n = (size_t)rows * cols;
        PoolBuf lst(ctx);
        OFRI_TRY(lst.get((n + 1) * sizeof(unsigned)));
        unsigned *count = lst.as<unsigned>(), *list = count + 1;
        OFRI_TRY(ofri_dev_fill(ctx, count, 0, sizeof(unsigned)));
        const bool asym = (asym_left | asym_right | asym_top | asym_bottom) != 0;
        const LktAsym as{(float)(1 - asym_left), (float)(1 - asym_right), (float)(1 - asym_top), (float)(1 - asym_bottom)};
        if (half_window == 6) lkt_launch<13>(ctx, im1, im2, U_in, V_in, U_out, V_out, rows, cols, niter, list, count, asym, as);
        else if (half_window == 15) lkt_launch<31>(ctx, im1, im2, U_in, V_in, U_out, V_out, rows, cols, niter, list, count, asym, as);
        else lkt_launch<27>(ctx, im1, im2, U_in, V_in, U_out, V_out, rows, cols, niter, list, count, asym, as);
        OFRI_LAUNCH_CHECK();
        int gl = (int)((n + LK_PX - 1) / LK_PX);
        if (gl > 512) gl = 512;              // a small resident grid: the kernel loops over *count, which is mostly zero
        k_lk_list<<<gl, 64 * LK_PX, 0, ctx->stream>>>(im1, im2, U_in, V_in, U_out, V_out, rows, cols, p, list, count);
        OFRI_LAUNCH_CHECK();
        return OFRI_OK;
    }
    dim3 g(ceil_div(cols, LK_PX), rows);
    k_lk_dense<<<g, 64 * LK_PX, 0, ctx->stream>>>(im1, im2, U_in, V_in, U_out, V_out, rows, cols, p);
    OFRI_LAUNCH_CHECK();
    return OFRI_OK;
}

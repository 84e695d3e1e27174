// Separable Gaussian pre-filter.  Reference behaviour: src/gaussian_filter.py:24-94.
#include "ofri_internal.h"
#include "ofri_image_math.h"
#include <cmath>

#define GAUSS_MAX_TAPS 127


// prepareGaussianKernel (gaussian_filter.py:47-52): offsets arange(-K/2, K/2) cast to int (truncation
// toward zero of -K/2 + i), float64 pdf stored to float32, divided by the float32 sum of the taps.
extern "C" int ofri_gauss_taps(double sigma, int ksize, float *taps_host) {
    OFRI_REQUIRE(taps_host, "taps_host is null");
    OFRI_REQUIRE(ksize > 0 && ksize <= GAUSS_MAX_TAPS && sigma > 0.0, "bad sigma/ksize");
    // NumPy: arange(start, stop, 1, dtype=int) has ceil((stop-start)/1) elements, value int(start) + i
    int first = (int)(-ksize / 2.0);
    for (int i = 0; i < ksize; ++i) {
        double x = (double)(first + i);
        const double s2 = sigma * sigma;
        taps_host[i] = (float)(1.0 / std::sqrt(2.0 * M_PI * s2) * std::exp(-(x * x) / (2.0 * s2)));
    }
    // np.sum over a float32 array uses pairwise summation; for fewer than 8 elements it is the plain
    // left-to-right float32 sum, which is what is reproduced here (blocks of 8 with 8 partial sums beyond)
    float s;
    if (ksize < 8) {
        s = 0.f;                       // NumPy starts from the first element; 0 + x is exact
        for (int i = 0; i < ksize; ++i) s = s + taps_host[i];
    } else {
        float r[8];
        for (int j = 0; j < 8; ++j) r[j] = taps_host[j];
        int i = 8;
        for (; i + 8 <= ksize; i += 8)
            for (int j = 0; j < 8; ++j) r[j] = r[j] + taps_host[i + j];
        s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < ksize; ++i) s = s + taps_host[i];
    }
    for (int i = 0; i < ksize; ++i) taps_host[i] = taps_host[i] / s;
    return OFRI_OK;
}

struct GaussTaps { float k[GAUSS_MAX_TAPS]; };

// One 1-D pass.  Output sample p accumulates in[refl(p + h - j)] * k[j] for j = 0..K-1 in float32,
// products and sums rounded separately, starting from 0 (convolveDirect, gaussian_filter.py:37-40).
// AXIS = 1: along a row (x);  AXIS = 0: along a column (y).
template <int AXIS>
__global__ void k_gauss_pass(const float *__restrict__ in, float *__restrict__ out, int H, int W, int K, GaussTaps taps) {
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= W || y >= H) return;
    int h = K / 2;
    float acc = 0.f;
    if (AXIS == 1) {
        const float *row = in + (size_t)y * W;
        for (int j = 0; j < K; ++j) acc = __fadd_rn(acc, __fmul_rn(row[gauss_idx(x + h - j, W, h)], taps.k[j]));
    } else {
        for (int j = 0; j < K; ++j) acc = __fadd_rn(acc, __fmul_rn(in[(size_t)gauss_idx(y + h - j, H, h) * W + x], taps.k[j]));
    }
    out[(size_t)y * W + x] = acc;
}

// Fused rows+columns pass through LDS for short kernels (K <= 7): a (TY+2h) x (TX+2h) input tile is
// row-filtered into LDS, then column-filtered, so the image is read once and written once.
#define GT_X 64
#define GT_Y 16
#define GK_MAXH 3
template <int K>
__global__ void __launch_bounds__(256)
k_gauss_fused(const float *__restrict__ in, float *__restrict__ out, int H, int W, GaussTaps taps) {
    constexpr int h = K / 2;
    __shared__ float rowf[GT_Y + 2 * h][GT_X + 1];
    const int bx = blockIdx.x * GT_X, by = blockIdx.y * GT_Y;
    const int tid = threadIdx.y * 64 + threadIdx.x;
    // stage 1: row-filtered values for tile rows by-h .. by+GT_Y+h-1 (the border rule in y applies to
    // the row-filtered image, which equals row-filtering the border-mapped source row)
    for (int i = tid; i < (GT_Y + 2 * h) * GT_X; i += 256) {
        int ly = i / GT_X, lx = i % GT_X;
        int x = bx + lx;
        int y = clampi(gauss_idx(by + ly - h, H, h), 0, H - 1);
        float acc = 0.f;
        // rows further than h below the image are never consumed (and would not reflect validly)
        if (x < W && by + ly - h <= H - 1 + h) {
            const float *row = in + (size_t)y * W;
#pragma unroll
            for (int j = 0; j < K; ++j) acc = __fadd_rn(acc, __fmul_rn(row[gauss_idx(x + h - j, W, h)], taps.k[j]));
        }
        rowf[ly][lx] = acc;
    }
    __syncthreads();
    for (int i = tid; i < GT_Y * GT_X; i += 256) {
        int ly = i / GT_X, lx = i % GT_X;
        int x = bx + lx, y = by + ly;
        if (x < W && y < H) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < K; ++j) acc = __fadd_rn(acc, __fmul_rn(rowf[ly + 2 * h - j][lx], taps.k[j]));
            out[(size_t)y * W + x] = acc;
        }
    }
}

// Rows+columns in registers for widths that are multiples of 4 (16-byte aligned planes): a thread owns 4
// adjacent columns of GV_R consecutive rows, row-filters the GV_R + 2h source rows it needs (one 16-byte load
// plus 2h border-mapped scalars each) and column-filters out of registers.  Same operation order per output
// as the other kernels.
#define GV_R 8
template <int K>
__global__ void __launch_bounds__(256)
k_gauss_vec4(const float *__restrict__ in, float *__restrict__ out, int H, int W, GaussTaps taps) {
    constexpr int h = K / 2;
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int y0 = (blockIdx.y * blockDim.y + threadIdx.y) * GV_R;
    if (x0 >= W || y0 >= H) return;
    int xl[h > 0 ? h : 1], xr[h > 0 ? h : 1];
#pragma unroll
    for (int i = 0; i < h; ++i) { xl[i] = gauss_idx(x0 - h + i, W, h); xr[i] = gauss_idx(x0 + 4 + i, W, h); }
    float rf[GV_R + 2 * h][4];
#pragma unroll
    for (int i = 0; i < GV_R + 2 * h; ++i) {
        const int y = clampi(gauss_idx(y0 - h + i, H, h), 0, H - 1);     // rows past the last consumed one: any valid row
        const float *row = in + (size_t)y * W;
        float v[4 + 2 * h];
        const float4 c = *reinterpret_cast<const float4 *>(row + x0);
        v[h] = c.x; v[h + 1] = c.y; v[h + 2] = c.z; v[h + 3] = c.w;
#pragma unroll
        for (int q = 0; q < h; ++q) { v[q] = row[xl[q]]; v[h + 4 + q] = row[xr[q]]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < K; ++j) acc = __fadd_rn(acc, __fmul_rn(v[k + 2 * h - j], taps.k[j]));
            rf[i][k] = acc;
        }
    }
#pragma unroll
    for (int r = 0; r < GV_R; ++r) {
        if (y0 + r >= H) break;
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < K; ++j) acc = __fadd_rn(acc, __fmul_rn(rf[r + 2 * h - j][k], taps.k[j]));
            o[k] = acc;
        }
        *reinterpret_cast<float4 *>(out + (size_t)(y0 + r) * W + x0) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

extern "C" int ofri_gauss_sep(ofri_ctx *ctx, const float *src, float *dst, int rows, int cols, const float *taps_host, int ntaps) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(src && dst && taps_host, "null pointer");
    OFRI_REQUIRE(rows > 0 && cols > 0, "bad shape %d x %d", rows, cols);
    OFRI_REQUIRE(ntaps >= 1 && (ntaps & 1) && ntaps <= GAUSS_MAX_TAPS, "ntaps must be odd and <= %d", GAUSS_MAX_TAPS);
    // the reference fills its border pad from the row itself and fails when the row is shorter than K/2
    OFRI_REQUIRE(ntaps / 2 <= rows && ntaps / 2 <= cols, "image smaller than half the kernel");
    GaussTaps t;
    memset(&t, 0, sizeof(t));
    memcpy(t.k, taps_host, ntaps * sizeof(float));
    size_t bytes = (size_t)rows * cols * sizeof(float);
    const bool fused = (ntaps == 3 || ntaps == 5 || ntaps == 7);
    PoolBuf tmp(ctx);
    if (!fused || src == dst) OFRI_TRY(tmp.get(bytes));
    if (fused) {
        float *o = (src == dst) ? tmp.as<float>() : dst;
        const bool vec = (cols & 3) == 0 && ((((size_t)src | (size_t)o) & 15) == 0);
        if (vec) {
            dim3 bv(64, 4), gv(ceil_div(cols, 256), ceil_div(rows, 4 * GV_R));
            if (ntaps == 3) k_gauss_vec4<3><<<gv, bv, 0, ctx->stream>>>(src, o, rows, cols, t);
            else if (ntaps == 5) k_gauss_vec4<5><<<gv, bv, 0, ctx->stream>>>(src, o, rows, cols, t);
            else k_gauss_vec4<7><<<gv, bv, 0, ctx->stream>>>(src, o, rows, cols, t);
            OFRI_LAUNCH_CHECK();
            if (o != dst) OFRI_TRY(ofri_dev_copy(ctx, dst, o, bytes));
            return OFRI_OK;
        }
        dim3 b(64, 4), g(ceil_div(cols, GT_X), ceil_div(rows, GT_Y));
        if (ntaps == 3) k_gauss_fused<3><<<g, b, 0, ctx->stream>>>(src, o, rows, cols, t);
        else if (ntaps == 5) k_gauss_fused<5><<<g, b, 0, ctx->stream>>>(src, o, rows, cols, t);
        else k_gauss_fused<7><<<g, b, 0, ctx->stream>>>(src, o, rows, cols, t);
        OFRI_LAUNCH_CHECK();
        if (o != dst) OFRI_TRY(ofri_dev_copy(ctx, dst, o, bytes));
    } else {
        dim3 b(64, 4), g(ceil_div(cols, 64), ceil_div(rows, 4));
        k_gauss_pass<1><<<g, b, 0, ctx->stream>>>(src, tmp.as<float>(), rows, cols, ntaps, t);
        OFRI_LAUNCH_CHECK();
        k_gauss_pass<0><<<g, b, 0, ctx->stream>>>(tmp.as<float>(), dst, rows, cols, ntaps, t);
        OFRI_LAUNCH_CHECK();
    }
    return OFRI_OK;
}

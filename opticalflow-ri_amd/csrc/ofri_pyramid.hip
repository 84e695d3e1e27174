// Pyramid glue: Pillow-compatible resampling and the bilinear warps (spline flow upsampling: ofri_spline.hip).
// Reference behaviour: src/GenericPyramidalOpticalFlow.py:67-235.
#include "ofri_internal.h"
#include "ofri_image_math.h"
#include <cmath>
#include <string>
#include <vector>

// ---------------------------------------------------------------------------------------------
// Pillow's resampler (Image.resize on a mode-'F' image): for each output sample a window
// [start, start+count) of input samples with float64 weights normalised to sum 1; the filter is
// stretched by max(in/out, 1) (antialiasing).  Accumulation in float64, stored as float32.
static double cubic_w(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}
static double linear_w(double x) {
    if (x < 0.0) x = -x;
    return x < 1.0 ? 1.0 - x : 0.0;
}

struct ResampleTable { int ksize, n_out, maxcnt; const int *bounds; const double *w; };   // maxcnt: longest window   // device pointers; w is tap-major: w[tap*n_out + out]

static int get_resample_table(ofri_ctx *ctx, int n_in, int n_out, int filter, ResampleTable *t) {
    const double support0 = filter == 0 ? 2.0 : 1.0;
    double (*fw)(double) = filter == 0 ? cubic_w : linear_w;
    const double scale = (double)n_in / (double)n_out;
    const double fs = scale < 1.0 ? 1.0 : scale;
    const double sup = support0 * fs;
    const int ksize = (int)std::ceil(sup) * 2 + 1;
    std::string key = "rs" + std::to_string(filter) + "_" + std::to_string(n_in) + "_" + std::to_string(n_out);
    void *pb = ofri_find_table(ctx, key + "b"), *pw = ofri_find_table(ctx, key + "w");
    if (!pb || !pw) {
        std::vector<int> bounds(2 * (size_t)n_out);
        std::vector<double> w((size_t)ksize * n_out, 0.0);
        for (int o = 0; o < n_out; ++o) {
            double c = (o + 0.5) * scale;
            int lo = (int)(c - sup + 0.5); if (lo < 0) lo = 0;
            int hi = (int)(c + sup + 0.5); if (hi > n_in) hi = n_in;
            int cnt = hi - lo;
            double tot = 0.0;
            std::vector<double> k(cnt > 0 ? cnt : 1);
            for (int x = 0; x < cnt; ++x) { k[x] = fw((x + lo - c + 0.5) * (1.0 / fs)); tot += k[x]; }
            if (tot != 0.0) for (int x = 0; x < cnt; ++x) k[x] /= tot;
            for (int x = 0; x < cnt; ++x) w[(size_t)x * n_out + o] = k[x];
            bounds[2 * o] = lo; bounds[2 * o + 1] = cnt;
        }
        OFRI_TRY(ofri_cached_table(ctx, key + "b", bounds.data(), bounds.size() * sizeof(int), &pb));
        OFRI_TRY(ofri_cached_table(ctx, key + "w", w.data(), w.size() * sizeof(double), &pw));
    }
    int maxcnt = 0;                                   // (integer work on n_out entries: cheaper to redo than to cache)
    for (int o = 0; o < n_out; ++o) {
        const double c = (o + 0.5) * scale;
        int lo = (int)(c - sup + 0.5); if (lo < 0) lo = 0;
        int hi = (int)(c + sup + 0.5); if (hi > n_in) hi = n_in;
        if (hi - lo > maxcnt) maxcnt = hi - lo;
    }
    t->ksize = ksize; t->n_out = n_out; t->maxcnt = maxcnt; t->bounds = (const int *)pb; t->w = (const double *)pw;
    return OFRI_OK;
}

// AXIS = 1: resample along x (rows kept);  AXIS = 0: along y (columns kept).
// KS > 0: the window never exceeds KS taps; all KS loads are issued up front (index clamped into the
// window, weight 0 beyond it - adding an exact +0 product leaves the float64 sum unchanged) so the
// sequential float64 accumulation does not wait on one load per tap.  KS = 0: any window length.
template <int AXIS, int KS>
__global__ void k_resample(const float *__restrict__ in, int H_in, int W_in, float *__restrict__ out,
                           int H_out, int W_out, ResampleTable t) {
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= W_out || y >= H_out) return;
    int o = AXIS == 1 ? x : y;
    int lo = t.bounds[2 * o], cnt = t.bounds[2 * o + 1];
    const double *k = t.w + o;
    const size_t ks = (size_t)t.n_out;
    const float *base = AXIS == 1 ? in + (size_t)y * W_in + lo : in + (size_t)lo * W_in + x;
    const size_t step = AXIS == 1 ? 1 : (size_t)W_in;
    double ss = 0.0;
    if (KS > 0) {
        float v[KS > 0 ? KS : 1];
        double w[KS > 0 ? KS : 1];
#pragma unroll
        for (int i = 0; i < KS; ++i) {
            int ii = i < cnt ? i : (cnt > 0 ? cnt - 1 : 0);
            v[i] = base[(size_t)ii * step];
            w[i] = i < cnt ? k[i * ks] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < KS; ++i) ss = __dadd_rn(ss, __dmul_rn((double)v[i], w[i]));
    } else {
        for (int i = 0; i < cnt; ++i) ss = __dadd_rn(ss, __dmul_rn((double)base[(size_t)i * step], k[i * ks]));
    }
    out[(size_t)y * W_out + x] = (float)ss;
}

// Resampling along x with the window's weights held in registers and reused over RS_ROWS rows: the
// weights depend on the output column only, and loading them per output costs more than the samples.
#define RS_ROWS 8
template <int KS>
__global__ void k_resample_x_rows(const float *__restrict__ in, int H, int W_in, float *__restrict__ out, int W_out,
                                  ResampleTable t) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y0 = (blockIdx.y * blockDim.y + threadIdx.y) * RS_ROWS;
    if (x >= W_out || y0 >= H) return;
    const int lo = t.bounds[2 * x], cnt = t.bounds[2 * x + 1];
    const double *k = t.w + x;
    const size_t ks = (size_t)t.n_out;
    double w[KS];
    int off[KS];
#pragma unroll
    for (int i = 0; i < KS; ++i) {
        off[i] = lo + (i < cnt ? i : (cnt > 0 ? cnt - 1 : 0));
        w[i] = i < cnt ? k[i * ks] : 0.0;
    }
#pragma unroll 2
    for (int r = 0; r < RS_ROWS; ++r) {
        const int y = y0 + r;
        if (y >= H) break;
        const float *row = in + (size_t)y * W_in;
        float v[KS];
#pragma unroll
        for (int i = 0; i < KS; ++i) v[i] = row[off[i]];
        double ss = 0.0;
#pragma unroll
        for (int i = 0; i < KS; ++i) ss = __dadd_rn(ss, __dmul_rn((double)v[i], w[i]));
        out[(size_t)y * W_out + x] = (float)ss;
    }
}

// Both passes in one launch for windows of at most 9 taps: a workgroup makes a 64 x 16 output tile.  It first
// resamples along x the input rows the tile's y-windows cover (float64 accumulation, rounded to float32 as
// Pillow stores its intermediate image) into LDS, then along y out of LDS.  Same arithmetic, same bits as
// the two separate passes; the intermediate image never goes to memory.
// x pass: wavefront w takes the input rows rlo + w, rlo + w + 4, ...; a row segment goes through the
// wavefront's own LDS line (written and read by that wavefront only, so no workgroup barrier), and the
// segments of the next two rows are already on their way in registers.
struct ResizeImages { const float *src[2]; float *dst[2]; };     // the images of one launch (blockIdx.z)
#define RF_TX 64
#define RF_TY 16                         // (32 and 64 rows measured: 7 and 15 % slower at 4096 -> 2048)
#define RF_ROWS 48                       // input rows a tile may need (16 outputs at 2:1 need 38)
#define RF_SEG 160                       // input columns a tile may need (64 outputs at 2:1 need 137)
// KT: no window is longer than KT taps (9 by construction; 8 for the 2:1 reduction of a pyramid level)
template <int KT>
__global__ void __launch_bounds__(256)
k_resize_fused(ResizeImages im, int H_in, int W_in, int H_out, int W_out, ResampleTable tx, ResampleTable ty) {
    const float *__restrict__ in = im.src[blockIdx.z];
    float *__restrict__ out = im.dst[blockIdx.z];
    __shared__ float hrow[RF_ROWS][RF_TX];
    __shared__ float seg[4][RF_SEG];                                   // one input row segment per wavefront
    const int lx = threadIdx.x & 63, ly = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);           // 64 x 4
    const int xb = blockIdx.x * RF_TX, x = xb + lx, y0 = blockIdx.y * RF_TY;
    const int ylast = min(y0 + RF_TY, H_out) - 1, xlast = min(xb + RF_TX, W_out) - 1;
    const int rlo = ty.bounds[2 * y0], rhi = min(ty.bounds[2 * ylast] + ty.bounds[2 * ylast + 1], rlo + RF_ROWS);   // uniform
    const int xlo = tx.bounds[2 * xb], span = min(tx.bounds[2 * xlast] + tx.bounds[2 * xlast + 1] - xlo, RF_SEG);
    double w[KT];
    int off[KT];
    {
        const int xc = min(x, W_out - 1);
        const int lo = tx.bounds[2 * xc], cnt = tx.bounds[2 * xc + 1];
        const double *k = tx.w + xc;
        const size_t ks = (size_t)tx.n_out;
#pragma unroll
        for (int i = 0; i < KT; ++i) {
            off[i] = min(lo + (i < cnt ? i : (cnt > 0 ? cnt - 1 : 0)) - xlo, RF_SEG - 1);
            w[i] = i < cnt ? k[i * ks] : 0.0;
        }
    }
    // columns lx, lx + 64, lx + 128 of the segment; rows past the tile's last repeat it (and are not used)
    const int c0 = min(lx, span - 1), c1 = min(lx + 64, span - 1), c2 = min(lx + 128, span - 1);
    auto fetch = [&](int r, float &v0, float &v1, float &v2) {
        const float *row = in + (size_t)min(r, rhi - 1) * W_in + xlo;
        v0 = row[c0]; v1 = row[c1]; v2 = row[c2];
    };
    float a0, a1, a2, b0, b1, b2, n0, n1, n2;
    fetch(rlo + ly, a0, a1, a2);
    fetch(rlo + ly + 4, b0, b1, b2);
    float *line = seg[ly];
    for (int r = rlo + ly; r < rhi; r += 4) {
        fetch(r + 8, n0, n1, n2);
        line[lx] = a0; line[lx + 64] = a1;
        if (lx + 128 < RF_SEG) line[lx + 128] = a2;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double ss = 0.0;
#pragma unroll
        for (int i = 0; i < KT; ++i) ss = __dadd_rn(ss, __dmul_rn((double)line[off[i]], w[i]));
        hrow[r - rlo][lx] = (float)ss;
        __builtin_amdgcn_wave_barrier();                                // the line is rewritten after these reads
        a0 = b0; a1 = b1; a2 = b2;
        b0 = n0; b1 = n1; b2 = n2;
    }
    __syncthreads();
    if (x >= W_out) return;
#pragma unroll
    for (int j = 0; j < RF_TY / 4; ++j) {
        const int y = y0 + ly + 4 * j;                                  // the same for the whole wavefront: window and weights are scalar loads
        if (y >= H_out) break;
        const int lo = ty.bounds[2 * y], cnt = ty.bounds[2 * y + 1];
        const double *k = ty.w + y;
        const size_t ks = (size_t)ty.n_out;
        double ss = 0.0;
#pragma unroll
        for (int i = 0; i < KT; ++i) {
            const int ii = i < cnt ? i : (cnt > 0 ? cnt - 1 : 0);
            const double wi = i < cnt ? k[i * ks] : 0.0;
            ss = __dadd_rn(ss, __dmul_rn((double)hrow[lo + ii - rlo][lx], wi));
        }
        out[(size_t)y * W_out + x] = (float)ss;
    }
}
// a tile's windows must fit the LDS rows / columns: true for every tile when the scale and support are modest
static bool resize_fused_fits(int n_in, int n_out, const ResampleTable &t, int filter, int tile, int room) {
    if (t.ksize > 9) return false;
    const double scale = (double)n_in / (double)n_out, fs = scale < 1.0 ? 1.0 : scale;
    const double sup = (filter == 0 ? 2.0 : 1.0) * fs;
    return (tile - 1) * scale + 2.0 * sup + 3.0 <= (double)room;
}

template <int AXIS>
static void resample_launch(ofri_ctx *ctx, dim3 g, dim3 b, const float *in, int H_in, int W_in, float *out, int H_out,
                            int W_out, const ResampleTable &t) {
    if (AXIS == 1 && t.ksize <= 9) {
        dim3 gr(g.x, ceil_div(H_out, (int)b.y * RS_ROWS));
        if (t.ksize <= 5) k_resample_x_rows<5><<<gr, b, 0, ctx->stream>>>(in, H_out, W_in, out, W_out, t);
        else k_resample_x_rows<9><<<gr, b, 0, ctx->stream>>>(in, H_out, W_in, out, W_out, t);
    } else if (t.ksize <= 5) k_resample<AXIS, 5><<<g, b, 0, ctx->stream>>>(in, H_in, W_in, out, H_out, W_out, t);
    else if (t.ksize <= 9) k_resample<AXIS, 9><<<g, b, 0, ctx->stream>>>(in, H_in, W_in, out, H_out, W_out, t);
    else k_resample<AXIS, 0><<<g, b, 0, ctx->stream>>>(in, H_in, W_in, out, H_out, W_out, t);
}

// n images (1 or 2) of the same shapes
static int resize_generic(ofri_ctx *ctx, int n, const float *const *srcs, int rows_in, int cols_in, float *const *dsts,
                          int rows_out, int cols_out, int filter) {
    OFRI_TRY(ofri_check_ctx(ctx));
    for (int k = 0; k < n; ++k) {
        OFRI_REQUIRE(srcs[k] && dsts[k], "null pointer");
        OFRI_REQUIRE(srcs[k] != dsts[k] && srcs[k] != dsts[n - 1 - k], "resize cannot run in place");
    }
    OFRI_REQUIRE(n == 1 || dsts[0] != dsts[1], "aliased pointer");
    OFRI_REQUIRE(rows_in > 0 && cols_in > 0 && rows_out > 0 && cols_out > 0, "bad shape");
    const bool need_h = cols_out != cols_in, need_v = rows_out != rows_in;
    if (!need_h && !need_v) {
        for (int k = 0; k < n; ++k) OFRI_TRY(ofri_dev_copy(ctx, dsts[k], srcs[k], (size_t)rows_in * cols_in * sizeof(float)));
        return OFRI_OK;
    }
    if (need_h && need_v) {
        ResampleTable th, tv;
        OFRI_TRY(get_resample_table(ctx, cols_in, cols_out, filter, &th));
        OFRI_TRY(get_resample_table(ctx, rows_in, rows_out, filter, &tv));
        if (resize_fused_fits(cols_in, cols_out, th, filter, RF_TX, RF_SEG) && resize_fused_fits(rows_in, rows_out, tv, filter, RF_TY, RF_ROWS)) {
            const ResizeImages im = {{srcs[0], srcs[n - 1]}, {dsts[0], dsts[n - 1]}};
            dim3 g(ceil_div(cols_out, RF_TX), ceil_div(rows_out, RF_TY), n);
            const int mc = th.maxcnt > tv.maxcnt ? th.maxcnt : tv.maxcnt;
            if (mc <= 5) k_resize_fused<5><<<g, 256, 0, ctx->stream>>>(im, rows_in, cols_in, rows_out, cols_out, th, tv);
            else if (mc <= 8) k_resize_fused<8><<<g, 256, 0, ctx->stream>>>(im, rows_in, cols_in, rows_out, cols_out, th, tv);
            else k_resize_fused<9><<<g, 256, 0, ctx->stream>>>(im, rows_in, cols_in, rows_out, cols_out, th, tv);
            OFRI_LAUNCH_CHECK();
            return OFRI_OK;
        }
    }
    dim3 b(64, 4);
    for (int k = 0; k < n; ++k) {
        const float *cur = srcs[k];
        PoolBuf tmp(ctx);
        if (need_h) {
            ResampleTable t;
            OFRI_TRY(get_resample_table(ctx, cols_in, cols_out, filter, &t));
            float *o = dsts[k];
            if (need_v) { OFRI_TRY(tmp.get((size_t)rows_in * cols_out * sizeof(float))); o = tmp.as<float>(); }
            dim3 g(ceil_div(cols_out, 64), ceil_div(rows_in, 4));
            resample_launch<1>(ctx, g, b, cur, rows_in, cols_in, o, rows_in, cols_out, t);
            OFRI_LAUNCH_CHECK();
            cur = o;
        }
        if (need_v) {
            ResampleTable t;
            OFRI_TRY(get_resample_table(ctx, rows_in, rows_out, filter, &t));
            dim3 g(ceil_div(cols_out, 64), ceil_div(rows_out, 4));
            resample_launch<0>(ctx, g, b, cur, rows_in, cols_out, dsts[k], rows_out, cols_out, t);
            OFRI_LAUNCH_CHECK();
        }
    }
    return OFRI_OK;
}

extern "C" int ofri_resize_bicubic(ofri_ctx *ctx, const float *src, int rows_in, int cols_in,
                                   float *dst, int rows_out, int cols_out) {
    return resize_generic(ctx, 1, &src, rows_in, cols_in, &dst, rows_out, cols_out, 0);
}
extern "C" int ofri_resize_bicubic_pair(ofri_ctx *ctx, const float *src1, const float *src2, int rows_in, int cols_in,
                                        float *dst1, float *dst2, int rows_out, int cols_out) {
    const float *srcs[2] = {src1, src2};
    float *dsts[2] = {dst1, dst2};
    return resize_generic(ctx, 2, srcs, rows_in, cols_in, dsts, rows_out, cols_out, 0);
}
extern "C" int ofri_resize_bilinear(ofri_ctx *ctx, const float *src, int rows_in, int cols_in,
                                    float *dst, int rows_out, int cols_out) {
    return resize_generic(ctx, 1, &src, rows_in, cols_in, &dst, rows_out, cols_out, 1);
}

// ---------------------------------------------------------------------------------------------
// Bilinear warp, clamp-to-edge: warp_sample in ofri_image_math.h.
__global__ void k_warp_coords(const float *__restrict__ img, const float *__restrict__ cy, const float *__restrict__ cx,
                              float *__restrict__ out, int H, int W) {
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= W || y >= H) return;
    size_t o = (size_t)y * W + x;
    out[o] = warp_sample(img, H, W, cy[o], cx[o]);
}
// coordinate = float32(index -/+ flow/2): flow/2 is exact in float32, and the float64 difference of an
// int32 index and a float32 rounds to float32 the same way the float32 subtraction does
__global__ void k_warp_flow(const float *__restrict__ img, const float *__restrict__ U, const float *__restrict__ V,
                            float sgn, float *__restrict__ out, int H, int W) {
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= W || y >= H) return;
    size_t o = (size_t)y * W + x;
    float cy = (float)((double)y + (double)(sgn * (V[o] * 0.5f)));
    float cx = (float)((double)x + (double)(sgn * (U[o] * 0.5f)));
    out[o] = warp_sample(img, H, W, cy, cx);
}

// both images of the symmetric warp in one pass: the flow is read once
__global__ void k_warp_pair(const float *__restrict__ im1, const float *__restrict__ im2, const float *__restrict__ U,
                            const float *__restrict__ V, float *__restrict__ out1, float *__restrict__ out2, int H, int W) {
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= W || y >= H) return;
    size_t o = (size_t)y * W + x;
    float hu = U[o] * 0.5f, hv = V[o] * 0.5f;
    out1[o] = warp_sample(im1, H, W, (float)((double)y + (double)(-hv)), (float)((double)x + (double)(-hu)));
    out2[o] = warp_sample(im2, H, W, (float)((double)y + (double)hv), (float)((double)x + (double)hu));
}

extern "C" int ofri_warp_pair(ofri_ctx *ctx, const float *im1, const float *im2, const float *U, const float *V,
                              float *out1, float *out2, int rows, int cols) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(im1 && im2 && U && V && out1 && out2, "null pointer");
    OFRI_REQUIRE(out1 != im1 && out2 != im2 && out1 != out2 && out1 != im2 && out2 != im1, "aliased pointer");
    OFRI_REQUIRE(rows > 0 && cols > 0, "bad shape");
    dim3 b(64, 4), g(ceil_div(cols, 64), ceil_div(rows, 4));
    k_warp_pair<<<g, b, 0, ctx->stream>>>(im1, im2, U, V, out1, out2, rows, cols);
    OFRI_LAUNCH_CHECK();
    return OFRI_OK;
}

extern "C" int ofri_warp_coords(ofri_ctx *ctx, const float *img, const float *cy, const float *cx,
                                float *out, int rows, int cols) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(img && cy && cx && out && out != img, "null or aliased pointer");
    OFRI_REQUIRE(rows > 0 && cols > 0, "bad shape");
    dim3 b(64, 4), g(ceil_div(cols, 64), ceil_div(rows, 4));
    k_warp_coords<<<g, b, 0, ctx->stream>>>(img, cy, cx, out, rows, cols);
    OFRI_LAUNCH_CHECK();
    return OFRI_OK;
}
extern "C" int ofri_warp_flow(ofri_ctx *ctx, const float *img, const float *U, const float *V,
                              float half_sign, float *out, int rows, int cols) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(img && U && V && out && out != img, "null or aliased pointer");
    OFRI_REQUIRE(rows > 0 && cols > 0, "bad shape");
    OFRI_REQUIRE(half_sign == 1.0f || half_sign == -1.0f, "half_sign must be +1 or -1");
    dim3 b(64, 4), g(ceil_div(cols, 64), ceil_div(rows, 4));
    k_warp_flow<<<g, b, 0, ctx->stream>>>(img, U, V, half_sign, out, rows, cols);
    OFRI_LAUNCH_CHECK();
    return OFRI_OK;
}

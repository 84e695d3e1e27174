// Pyramid glue: Pillow-compatible resampling, interpolating-spline flow upsampling, bilinear warp.
// Reference behaviour: src/GenericPyramidalOpticalFlow.py:67-235.
#include "ofri_internal.h"
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

struct ResampleTable { int ksize; const int *bounds; const double *w; };   // device pointers

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
            double tot = 0.0, *k = &w[(size_t)o * ksize];
            for (int x = 0; x < cnt; ++x) { k[x] = fw((x + lo - c + 0.5) * (1.0 / fs)); tot += k[x]; }
            if (tot != 0.0) for (int x = 0; x < cnt; ++x) k[x] /= tot;
            bounds[2 * o] = lo; bounds[2 * o + 1] = cnt;
        }
        OFRI_TRY(ofri_cached_table(ctx, key + "b", bounds.data(), bounds.size() * sizeof(int), &pb));
        OFRI_TRY(ofri_cached_table(ctx, key + "w", w.data(), w.size() * sizeof(double), &pw));
    }
    t->ksize = ksize; t->bounds = (const int *)pb; t->w = (const double *)pw;
    return OFRI_OK;
}

// AXIS = 1: resample along x (rows kept);  AXIS = 0: along y (columns kept)
template <int AXIS>
__global__ void k_resample(const float *__restrict__ in, int H_in, int W_in, float *__restrict__ out,
                           int H_out, int W_out, ResampleTable t) {
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= W_out || y >= H_out) return;
    int o = AXIS == 1 ? x : y;
    int lo = t.bounds[2 * o], cnt = t.bounds[2 * o + 1];
    const double *k = t.w + (size_t)o * t.ksize;
    double ss = 0.0;
    if (AXIS == 1) {
        const float *row = in + (size_t)y * W_in + lo;
        for (int i = 0; i < cnt; ++i) ss = __dadd_rn(ss, __dmul_rn((double)row[i], k[i]));
    } else {
        const float *col = in + (size_t)lo * W_in + x;
        for (int i = 0; i < cnt; ++i) ss = __dadd_rn(ss, __dmul_rn((double)col[(size_t)i * W_in], k[i]));
    }
    out[(size_t)y * W_out + x] = (float)ss;
}

static int resize_generic(ofri_ctx *ctx, const float *src, int rows_in, int cols_in, float *dst,
                          int rows_out, int cols_out, int filter) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(src && dst, "null pointer");
    OFRI_REQUIRE(rows_in > 0 && cols_in > 0 && rows_out > 0 && cols_out > 0, "bad shape");
    OFRI_REQUIRE(src != dst, "resize cannot run in place");
    const bool need_h = cols_out != cols_in, need_v = rows_out != rows_in;
    if (!need_h && !need_v) {
        OFRI_HIP(hipMemcpyAsync(dst, src, (size_t)rows_in * cols_in * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
        return OFRI_OK;
    }
    dim3 b(64, 4);
    const float *cur = src;
    PoolBuf tmp(ctx);
    if (need_h) {
        ResampleTable t;
        OFRI_TRY(get_resample_table(ctx, cols_in, cols_out, filter, &t));
        float *o = dst;
        if (need_v) { OFRI_TRY(tmp.get((size_t)rows_in * cols_out * sizeof(float))); o = tmp.as<float>(); }
        dim3 g(ceil_div(cols_out, 64), ceil_div(rows_in, 4));
        k_resample<1><<<g, b, 0, ctx->stream>>>(cur, rows_in, cols_in, o, rows_in, cols_out, t);
        OFRI_LAUNCH_CHECK();
        cur = o;
    }
    if (need_v) {
        ResampleTable t;
        OFRI_TRY(get_resample_table(ctx, rows_in, rows_out, filter, &t));
        dim3 g(ceil_div(cols_out, 64), ceil_div(rows_out, 4));
        k_resample<0><<<g, b, 0, ctx->stream>>>(cur, rows_in, cols_out, dst, rows_out, cols_out, t);
        OFRI_LAUNCH_CHECK();
    }
    return OFRI_OK;
}

extern "C" int ofri_resize_bicubic(ofri_ctx *ctx, const float *src, int rows_in, int cols_in,
                                   float *dst, int rows_out, int cols_out) {
    return resize_generic(ctx, src, rows_in, cols_in, dst, rows_out, cols_out, 0);
}
extern "C" int ofri_resize_bilinear(ofri_ctx *ctx, const float *src, int rows_in, int cols_in,
                                    float *dst, int rows_out, int cols_out) {
    return resize_generic(ctx, src, rows_in, cols_in, dst, rows_out, cols_out, 1);
}

// ---------------------------------------------------------------------------------------------
// Interpolating bicubic spline (FITPACK k = 3, s = 0 == not-a-knot cubic spline per axis).
// Per line: second derivatives M (knot units) from the tridiagonal system
//   M[i-1] + 4 M[i] + M[i+1] = 6 (y[i-1] - 2 y[i] + y[i+1]),  i = 1..n-2,
// closed by M[0] = 2M[1]-M[2], M[n-1] = 2M[n-2]-M[n-3]; forward-elimination factors depend on n only
// and come from a host table.  Value at knot k + theta:
//   (1-theta) y[k] + theta y[k+1] + (((1-theta)^3-(1-theta)) M[k] + (theta^3-theta) M[k+1]) / 6.
struct SplineSolveTable { const double *cp; const double *inv; };   // device: cp[m], 1/den[m], m = n-2

static int get_spline_solve(ofri_ctx *ctx, int n, SplineSolveTable *t) {
    std::string key = "sp_" + std::to_string(n);
    void *pc = ofri_find_table(ctx, key + "c"), *pi = ofri_find_table(ctx, key + "i");
    if (!pc || !pi) {
        int m = n - 2;
        std::vector<double> cp(m), inv(m);
        // rows: [6 0] first, [1 4 1] interior, [0 6] last
        double dia0 = 6.0, sup0 = 0.0;
        cp[0] = sup0 / dia0; inv[0] = 1.0 / dia0;
        for (int i = 1; i < m; ++i) {
            double sub = (i == m - 1) ? 0.0 : 1.0, dia = (i == m - 1) ? 6.0 : 4.0, sup = 1.0;
            double den = dia - sub * cp[i - 1];
            cp[i] = sup / den; inv[i] = 1.0 / den;
        }
        OFRI_TRY(ofri_cached_table(ctx, key + "c", cp.data(), m * sizeof(double), &pc));
        OFRI_TRY(ofri_cached_table(ctx, key + "i", inv.data(), m * sizeof(double), &pi));
    }
    t->cp = (const double *)pc; t->inv = (const double *)pi;
    return OFRI_OK;
}

struct SplineEvalTable { const int *k; const double *theta; };
static int get_spline_eval(ofri_ctx *ctx, int n, int N, SplineEvalTable *t) {
    std::string key = "se_" + std::to_string(n) + "_" + std::to_string(N);
    void *pk = ofri_find_table(ctx, key + "k"), *pt = ofri_find_table(ctx, key + "t");
    if (!pk || !pt) {
        std::vector<int> k(N); std::vector<double> th(N);
        for (long long i = 0; i < N; ++i) {
            long long num = i * n, cap = (long long)(n - 1) * N;      // points past the last knot clamp onto it
            if (num > cap) num = cap;
            long long kk = num / N; if (kk > n - 2) kk = n - 2;
            k[i] = (int)kk; th[i] = (double)(num - kk * N) / (double)N;
        }
        OFRI_TRY(ofri_cached_table(ctx, key + "k", k.data(), N * sizeof(int), &pk));
        OFRI_TRY(ofri_cached_table(ctx, key + "t", th.data(), N * sizeof(double), &pt));
    }
    t->k = (const int *)pk; t->theta = (const double *)pt;
    return OFRI_OK;
}

// Solve along axis 0 for every column: y is (n, C) (float or double), M is (n, C) double.
// One thread per column; consecutive threads touch consecutive addresses.
template <class T>
__global__ void k_spline_solve_cols(const T *__restrict__ y, double *__restrict__ M, int n, int C, SplineSolveTable t) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int m = n - 2;
    // forward sweep: dp[i] kept in M[i+1]
    double y0 = (double)y[c], y1 = (double)y[(size_t)C + c];
    double prev = 0.0;
    for (int i = 0; i < m; ++i) {
        double y2 = (double)y[(size_t)(i + 2) * C + c];
        double rhs = 6.0 * (y0 - 2.0 * y1 + y2);
        double sub = (i == 0 || i == m - 1) ? 0.0 : 1.0;
        double dp = (rhs - sub * prev) * t.inv[i];
        M[(size_t)(i + 1) * C + c] = dp;
        prev = dp; y0 = y1; y1 = y2;
    }
    // back substitution
    double next = M[(size_t)m * C + c];
    for (int i = m - 2; i >= 0; --i) {
        double v = M[(size_t)(i + 1) * C + c] - t.cp[i] * next;
        M[(size_t)(i + 1) * C + c] = v;
        next = v;
    }
    double M1 = M[(size_t)C + c], M2 = M[(size_t)2 * C + c];
    M[c] = 2.0 * M1 - M2;
    double Ma = M[(size_t)(n - 2) * C + c], Mb = M[(size_t)(n - 3) * C + c];
    M[(size_t)(n - 1) * C + c] = 2.0 * Ma - Mb;
}

// Solve along axis 1 for every row: y is (R, n); thread per row, LDS-free (rows are cached lines).
template <class T>
__global__ void k_spline_solve_rows(const T *__restrict__ y, double *__restrict__ M, int R, int n, SplineSolveTable t) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const int m = n - 2;
    const T *yr = y + (size_t)r * n;
    double *Mr = M + (size_t)r * n;
    double y0 = (double)yr[0], y1 = (double)yr[1], prev = 0.0;
    for (int i = 0; i < m; ++i) {
        double y2 = (double)yr[i + 2];
        double rhs = 6.0 * (y0 - 2.0 * y1 + y2);
        double sub = (i == 0 || i == m - 1) ? 0.0 : 1.0;
        double dp = (rhs - sub * prev) * t.inv[i];
        Mr[i + 1] = dp;
        prev = dp; y0 = y1; y1 = y2;
    }
    double next = Mr[m];
    for (int i = m - 2; i >= 0; --i) {
        double v = Mr[i + 1] - t.cp[i] * next;
        Mr[i + 1] = v;
        next = v;
    }
    Mr[0] = 2.0 * Mr[1] - Mr[2];
    Mr[n - 1] = 2.0 * Mr[n - 2] - Mr[n - 3];
}

__device__ __forceinline__ double spline_val(double ya, double yb, double Ma, double Mb, double th) {
    double A = 1.0 - th, B = th;
    return A * ya + B * yb + ((A * A * A - A) * Ma + (B * B * B - B) * Mb) / 6.0;
}

// evaluate along axis 0: y (n, C) -> out (N, C) double
template <class T>
__global__ void k_spline_eval_cols(const T *__restrict__ y, const double *__restrict__ M, double *__restrict__ out,
                                   int n, int N, int C, SplineEvalTable t) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    int i = blockIdx.y * blockDim.y + threadIdx.y;
    if (c >= C || i >= N) return;
    int k = t.k[i];
    double th = t.theta[i];
    size_t a = (size_t)k * C + c, b = a + C;
    out[(size_t)i * C + c] = spline_val((double)y[a], (double)y[b], M[a], M[b], th);
}
// evaluate along axis 1: y (R, n) double -> out (R, N) float32, then * post_scale in float32
__global__ void k_spline_eval_rows(const double *__restrict__ y, const double *__restrict__ M, float *__restrict__ out,
                                   int R, int n, int N, SplineEvalTable t, float post_scale, int do_scale) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    int r = blockIdx.y * blockDim.y + threadIdx.y;
    if (j >= N || r >= R) return;
    int k = t.k[j];
    double th = t.theta[j];
    size_t a = (size_t)r * n + k;
    float v = (float)spline_val(y[a], y[a + 1], M[a], M[a + 1], th);
    out[(size_t)r * N + j] = do_scale ? __fmul_rn(v, post_scale) : v;
}

extern "C" int ofri_spline_upsample(ofri_ctx *ctx, const float *src, int rows_in, int cols_in,
                                    float *dst, int rows_out, int cols_out, float post_scale) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(src && dst && src != dst, "null or aliased pointer");
    OFRI_REQUIRE(rows_in >= 4 && cols_in >= 4, "cubic spline needs at least 4 samples per axis (got %d x %d)", rows_in, cols_in);
    OFRI_REQUIRE(rows_out > 0 && cols_out > 0, "bad output shape");
    SplineSolveTable s0, s1;
    SplineEvalTable e0, e1;
    OFRI_TRY(get_spline_solve(ctx, rows_in, &s0));
    OFRI_TRY(get_spline_solve(ctx, cols_in, &s1));
    OFRI_TRY(get_spline_eval(ctx, rows_in, rows_out, &e0));
    OFRI_TRY(get_spline_eval(ctx, cols_in, cols_out, &e1));
    PoolBuf M0(ctx), T(ctx), M1(ctx);
    OFRI_TRY(M0.get((size_t)rows_in * cols_in * sizeof(double)));
    OFRI_TRY(T.get((size_t)rows_out * cols_in * sizeof(double)));
    OFRI_TRY(M1.get((size_t)rows_out * cols_in * sizeof(double)));
    // axis 0
    k_spline_solve_cols<float><<<ceil_div(cols_in, 64), 64, 0, ctx->stream>>>(src, M0.as<double>(), rows_in, cols_in, s0);
    OFRI_LAUNCH_CHECK();
    {
        dim3 b(64, 4), g(ceil_div(cols_in, 64), ceil_div(rows_out, 4));
        k_spline_eval_cols<float><<<g, b, 0, ctx->stream>>>(src, M0.as<double>(), T.as<double>(), rows_in, rows_out, cols_in, e0);
        OFRI_LAUNCH_CHECK();
    }
    // axis 1
    k_spline_solve_rows<double><<<ceil_div(rows_out, 64), 64, 0, ctx->stream>>>(T.as<double>(), M1.as<double>(), rows_out, cols_in, s1);
    OFRI_LAUNCH_CHECK();
    {
        dim3 b(64, 4), g(ceil_div(cols_out, 64), ceil_div(rows_out, 4));
        k_spline_eval_rows<<<g, b, 0, ctx->stream>>>(T.as<double>(), M1.as<double>(), dst, rows_out, cols_in, cols_out, e1,
                                                     post_scale, post_scale != 1.0f);
        OFRI_LAUNCH_CHECK();
    }
    return OFRI_OK;
}

// ---------------------------------------------------------------------------------------------
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

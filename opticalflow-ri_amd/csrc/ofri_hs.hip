// Horn-Schunck: derivative cube, Jacobi sweeps, error norm.
// Reference behaviour: src/HornSchunck.py:52-127 (HS_helper2, HS_helper, HS, computeDerivatives).
#include "ofri_internal.h"
#include <atomic>
extern std::atomic<unsigned long long> g_ofri_runtime_allocs;      // ofri_ctx.hip: allocating runtime calls made so far
#include "ofri_hs_math.h"
#include "ofri_hs_stream.h"

int ofri_norm_diff_dev(ofri_ctx *ctx, const float *a, const float *b, size_t n, int slot);
int ofri_read_slots(ofri_ctx *ctx, int first, int count, double *out_host);
int ofri_norm_diff_at(ofri_ctx *ctx, const float *a, const float *b, size_t n, double *partials, double *out);
int ofri_read_doubles(ofri_ctx *ctx, const double *dev, int count, double *out_host);
int ofri_red_blocks(void);

// ---------------------------------------------------------------------------------------------
// Derivatives (formula and citation: hs_derivs_at in ofri_hs_math.h).
__global__ void k_hs_derivs(const float *__restrict__ P, const float *__restrict__ Q,   // P = first, Q = second
                            float *__restrict__ fx, float *__restrict__ fy, float *__restrict__ ft,
                            int H, int W) {
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= W || y >= H) return;
    float gx, gy, gt;
    hs_derivs_at(P, Q, y, x, H, W, gx, gy, gt);
    size_t o = (size_t)y * W + x;
    fx[o] = gx;
    fy[o] = gy;
    ft[o] = gt;
}

extern "C" int ofri_hs_derivs(ofri_ctx *ctx, const float *first, const float *second,
                              float *fx, float *fy, float *ft, int rows, int cols) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(first && second && fx && fy && ft, "null pointer");
    OFRI_REQUIRE(rows > 0 && cols > 0, "bad shape %d x %d", rows, cols);
    dim3 b(64, 4), g(ceil_div(cols, 64), ceil_div(rows, 4));
    k_hs_derivs<<<g, b, 0, ctx->stream>>>(first, second, fx, fy, ft, rows, cols);
    OFRI_LAUNCH_CHECK();
    return OFRI_OK;
}

// ---------------------------------------------------------------------------------------------
// Jacobi sweep, one launch per sweep.
//   ubar = K*U, vbar = K*V  (K = [[1/12,1/6,1/12],[1/6,0,1/6],[1/12,1/6,1/12]] in float32, 'mirror')
//   der  = (fx*ubar + fy*vbar + ft) / (alpha^2 + fx^2 + fy^2);  U = ubar - fx*der;  V = vbar - fy*der
// EXACT = true follows scipy's float64 tap-by-tap accumulation and NumPy's unfused float32 element
// ops, so the result is bit-identical to the reference; EXACT = false is float32 with a pinned
// operation order (ofri_hs_math.h) on coefficients scaled once by 1/sqrt(alpha^2 + fx^2 + fy^2) instead of a
// division per sweep; its kernels take the scaled planes A, B, C where the exact ones take fx, fy, ft.
#define W6  0.16666667163372040   // float32(1/6)  as a double
#define W12 0.08333333581686020   // float32(1/12) as a double

template <bool EXACT>
__device__ __forceinline__ float avg9(float a00, float a01, float a02, float a10, float a12,
                                      float a20, float a21, float a22) {
    if (EXACT) {
        double t = __dmul_rn((double)a00, W12);
        t = __dadd_rn(t, __dmul_rn((double)a01, W6));
        t = __dadd_rn(t, __dmul_rn((double)a02, W12));
        t = __dadd_rn(t, __dmul_rn((double)a10, W6));
        t = __dadd_rn(t, __dmul_rn((double)a12, W6));
        t = __dadd_rn(t, __dmul_rn((double)a20, W12));
        t = __dadd_rn(t, __dmul_rn((double)a21, W6));
        t = __dadd_rn(t, __dmul_rn((double)a22, W12));
        return (float)t;
    } else {
        return hs_avg(__fadd_rn(a01, a21), hs_q(a10, __fadd_rn(a00, a20)), hs_q(a12, __fadd_rn(a02, a22)));
    }
}

template <bool EXACT>
__device__ __forceinline__ void hs_point(float ua, float va, float fx, float fy, float ft, float a2,
                                         float &un, float &vn) {
    if (EXACT) {
        float den = __fadd_rn(__fadd_rn(a2, __fmul_rn(fx, fx)), __fmul_rn(fy, fy));
        float num = __fadd_rn(__fadd_rn(__fmul_rn(fx, ua), __fmul_rn(fy, va)), ft);
        float der = __fdiv_rn(num, den);
        un = __fsub_rn(ua, __fmul_rn(fx, der));
        vn = __fsub_rn(va, __fmul_rn(fy, der));
    } else {
        hs_update(ua, va, fx, fy, ft, un, vn);       // fx, fy, ft are the scaled coefficients A, B, C here
    }
}

// any shape: one pixel per thread
template <bool EXACT>
__global__ void k_hs_sweep_scalar(const float *__restrict__ U, const float *__restrict__ V,
                                  const float *__restrict__ fx, const float *__restrict__ fy,
                                  const float *__restrict__ ft, float *__restrict__ Un,
                                  float *__restrict__ Vn, float a2, int H, int W) {
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= W || y >= H) return;
    int xm = refl101(x - 1, W), xp = refl101(x + 1, W);
    size_t rm = (size_t)refl101(y - 1, H) * W, rc = (size_t)y * W, rp = (size_t)refl101(y + 1, H) * W;
    float ua = avg9<EXACT>(U[rm + xm], U[rm + x], U[rm + xp], U[rc + xm], U[rc + xp], U[rp + xm], U[rp + x], U[rp + xp]);
    float va = avg9<EXACT>(V[rm + xm], V[rm + x], V[rm + xp], V[rc + xm], V[rc + xp], V[rp + xm], V[rp + x], V[rp + xp]);
    float un, vn;
    hs_point<EXACT>(ua, va, fx[rc + x], fy[rc + x], ft[rc + x], a2, un, vn);
    Un[rc + x] = un;
    Vn[rc + x] = vn;
}

// cols % 4 == 0: each thread owns 4 adjacent pixels of RPT consecutive rows and slides a 3-row window
// of U and V through registers, so each U/V row is fetched once per thread (16-byte loads).
#define HS_RPT 4
template <bool EXACT>
__global__ void __launch_bounds__(256)
k_hs_sweep_vec4(const float *__restrict__ U, const float *__restrict__ V,
                const float *__restrict__ fx, const float *__restrict__ fy,
                const float *__restrict__ ft, float *__restrict__ Un, float *__restrict__ Vn,
                float a2, int H, int W) {
    int x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    int y0 = (blockIdx.y * blockDim.y + threadIdx.y) * HS_RPT;
    if (x0 >= W || y0 >= H) return;
    int xl = refl101(x0 - 1, W), xr = refl101(x0 + 4, W);
    float u[3][6], v[3][6];
    auto load_row = [&](int slot, int y) {
        size_t r = (size_t)refl101(y, H) * W;
        float4 a = *reinterpret_cast<const float4 *>(U + r + x0);
        float4 b = *reinterpret_cast<const float4 *>(V + r + x0);
        u[slot][0] = U[r + xl]; u[slot][1] = a.x; u[slot][2] = a.y; u[slot][3] = a.z; u[slot][4] = a.w; u[slot][5] = U[r + xr];
        v[slot][0] = V[r + xl]; v[slot][1] = b.x; v[slot][2] = b.y; v[slot][3] = b.z; v[slot][4] = b.w; v[slot][5] = V[r + xr];
    };
    load_row(0, y0 - 1);
    load_row(1, y0);
#pragma unroll
    for (int r = 0; r < HS_RPT; ++r) {
        int y = y0 + r;
        if (y >= H) break;
        const int m = r % 3, c = (r + 1) % 3, p = (r + 2) % 3;
        load_row(p, y + 1);
        size_t o = (size_t)y * W + x0;
        float4 gx = *reinterpret_cast<const float4 *>(fx + o);
        float4 gy = *reinterpret_cast<const float4 *>(fy + o);
        float4 gt = *reinterpret_cast<const float4 *>(ft + o);
        float gxa[4] = {gx.x, gx.y, gx.z, gx.w}, gya[4] = {gy.x, gy.y, gy.z, gy.w}, gta[4] = {gt.x, gt.y, gt.z, gt.w};
        float ou[4], ov[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float ua = avg9<EXACT>(u[m][k], u[m][k + 1], u[m][k + 2], u[c][k], u[c][k + 2], u[p][k], u[p][k + 1], u[p][k + 2]);
            float va = avg9<EXACT>(v[m][k], v[m][k + 1], v[m][k + 2], v[c][k], v[c][k + 2], v[p][k], v[p][k + 1], v[p][k + 2]);
            hs_point<EXACT>(ua, va, gxa[k], gya[k], gta[k], a2, ou[k], ov[k]);
        }
        *reinterpret_cast<float4 *>(Un + o) = make_float4(ou[0], ou[1], ou[2], ou[3]);
        *reinterpret_cast<float4 *>(Vn + o) = make_float4(ov[0], ov[1], ov[2], ov[3]);
    }
}

template <bool EXACT>
static int hs_sweep_launch(ofri_ctx *ctx, const float *U, const float *V, const float *fx, const float *fy,
                           const float *ft, float *Un, float *Vn, float a2, int H, int W) {
    bool aligned = ((((size_t)U | (size_t)V | (size_t)fx | (size_t)fy | (size_t)ft | (size_t)Un | (size_t)Vn) & 15) == 0);
    if (W % 4 == 0 && aligned && W >= 8) {
        dim3 b(64, 4), g(ceil_div(W, 256), ceil_div(H, 4 * HS_RPT));
        k_hs_sweep_vec4<EXACT><<<g, b, 0, ctx->stream>>>(U, V, fx, fy, ft, Un, Vn, a2, H, W);
    } else {
        dim3 b(64, 4), g(ceil_div(W, 64), ceil_div(H, 4));
        k_hs_sweep_scalar<EXACT><<<g, b, 0, ctx->stream>>>(U, V, fx, fy, ft, Un, Vn, a2, H, W);
    }
    OFRI_LAUNCH_CHECK();
    return OFRI_OK;
}

bool ofri_hs_fused_ok(const void *const *ptrs, int nptrs, int H, int W);
int ofri_hs_fused_max_sweeps(void);
int ofri_hs_fused_launch(ofri_ctx *ctx, const float *su, const float *sv, float *du, float *dv, const float *fx,
                         const float *fy, const float *ft, int T, int H, int W);

int ofri_hs_fused_launch_pitched(ofri_ctx *ctx, const float *su, const float *sv, float *du, float *dv, const float *fx,
                                 const float *fy, const float *ft, int T, int H, int W, int pin, int pc, int pout,
                                 bool accumulate);

extern "C" int ofri_set_share(ofri_ctx *ctx, int contexts) {
    OFRI_REQUIRE(ctx, "null context");
    OFRI_REQUIRE(contexts >= 1 && contexts <= 16, "contexts sharing the GPU: 1..16");
    ctx->share = contexts;
    return OFRI_OK;
}

extern "C" int ofri_hs_set_fuse(ofri_ctx *ctx, int sweeps_per_launch) {
    OFRI_REQUIRE(ctx, "null context");
    OFRI_REQUIRE(sweeps_per_launch >= -1 && sweeps_per_launch <= 64, "sweeps_per_launch out of range");
    ctx->hs_fuse = sweeps_per_launch;
    return OFRI_OK;
}

// Timed sections.  While the context captures a graph the events become event-record nodes of the graph (recorded again
// at every replay) and the section belongs to the graph (ofri_graph_prof_read).
static int prof_begin(ofri_ctx *ctx, ofri_ctx::ProfRec &rec) {
    if (!ctx->prof) return OFRI_OK;
    g_ofri_runtime_allocs.fetch_add(2, std::memory_order_relaxed);
    OFRI_HIP(hipEventCreate(&rec.a)); OFRI_HIP(hipEventCreate(&rec.b));
    if (ctx->capturing) OFRI_HIP(hipEventRecordWithFlags(rec.a, ctx->stream, hipEventRecordExternal));
    else OFRI_HIP(hipEventRecord(rec.a, ctx->stream));
    return OFRI_OK;
}
static int prof_end(ofri_ctx *ctx, ofri_ctx::ProfRec &rec, int rows, int cols, int launches, int sweeps) {
    if (!ctx->prof) return OFRI_OK;
    if (ctx->capturing) OFRI_HIP(hipEventRecordWithFlags(rec.b, ctx->stream, hipEventRecordExternal));
    else OFRI_HIP(hipEventRecord(rec.b, ctx->stream));
    rec.rows = rows; rec.cols = cols; rec.launches = launches; rec.sweeps = sweeps;
    (ctx->capturing ? ctx->cap_prof_recs : ctx->prof_recs).push_back(rec);
    return OFRI_OK;
}

// How many of niter sweeps go through the row-streaming pipeline (12 per launch): all of them with
// sweeps_per_launch = 16, the whole multiples of 12 on large fields in automatic mode, none otherwise.
static int hs_stream_share(ofri_ctx *ctx, const void *const *ptrs, int nptrs, int niter, int rows, int cols) {
    size_t bits = 0;
    for (int i = 0; i < nptrs; ++i) bits |= (size_t)ptrs[i];
    if ((bits & 15) != 0 || (cols & 3) != 0 || cols < 8 || niter <= 0) return 0;
    if (ctx->hs_fuse == 16 || ctx->hs_fuse == 32) return niter;
    const int per = ofri_hs_stream_sweeps();
    if (ctx->hs_fuse == 0 && ofri_hs_stream_pays(rows, cols) && niter >= 2 * per) {
        // up to 4 left-over sweeps ride in a deeper last launch (hs_stream_path); more of them go to the tile kernel
        const int rest = niter % per;
        return rest <= ofri_hs_stream_deep_sweeps() - per ? niter : niter - rest;
    }
    return 0;
}

// n_stream sweeps through the pipeline, the remaining niter - n_stream (fewer than 12) through the tile kernel
// reading the pipeline's plane set.  Coefficients: dense fx, fy, ft, or (fx == nullptr) differentiated from
// the frames straight into the set.  Filling the set is not part of the timing records.
// accumulate: the result is added to what (Uout, Vout) hold, in the last launch's stores.
// level: (first, second) are a pyramid level's images as the driver holds them before its warp and pre-filter;
// both are done on the way into the set (warp by level->U, level->V unless null), and the flow starts from zero.
struct HsLevelFill { const float *U, *V, *taps_host; int ntaps; };
static int hs_stream_path(ofri_ctx *ctx, const float *Uin, const float *Vin, float *Uout, float *Vout, const float *fx,
                          const float *fy, const float *ft, const float *first, const float *second, float a2, int niter,
                          int n_stream, int rows, int cols, bool accumulate = false, const HsLevelFill *level = nullptr) {
    PoolBuf buf(ctx);
    OFRI_TRY(buf.get(ofri_sl_bytes(rows, cols)));
    OfriSlSet set;
    ofri_sl_layout(buf.as<float>(), rows, cols, &set);
    if (level) OFRI_TRY(ofri_sl_fill_level(ctx, set, first, second, level->U, level->V, level->taps_host, level->ntaps, a2, rows, cols));
    else OFRI_TRY(ofri_sl_fill(ctx, set, Uin, Vin, fx, fy, ft, first, second, a2, rows, cols));
    int rest = niter - n_stream, cur = 0, launches = 0;
    // automatic mode: a last pipeline launch of 13 to 16 sweeps (timed on its own: it is another kernel)
    const int per = ofri_hs_stream_sweeps(), odd = n_stream % per;
    const int tail = (ctx->hs_fuse == 0 && n_stream > per && odd >= 1 && odd <= ofri_hs_stream_deep_sweeps() - per) ? per + odd : 0;
    const int bulk = n_stream - tail;
    // the deep launch goes first: it starts from the zero flow of a level without reading it, and the accumulating store is
    // cheaper in a 12-sweep launch (two loader wavefronts per workgroup) than in the deep one
    if (tail > 0) {
        ofri_ctx::ProfRec rect{};
        OFRI_TRY(prof_begin(ctx, rect));
        OFRI_TRY(ofri_sl_sweeps(ctx, set, tail, rest == 0 && bulk == 0, rest == 0 && bulk == 0 && accumulate, Uout, Vout, rows, cols, &cur, &launches, true));
        OFRI_TRY(prof_end(ctx, rect, rows, cols, launches, tail));
    }
    if (bulk > 0) {
        ofri_ctx::ProfRec rec{};
        OFRI_TRY(prof_begin(ctx, rec));
        OFRI_TRY(ofri_sl_sweeps(ctx, set, bulk, rest == 0, rest == 0 && accumulate, Uout, Vout, rows, cols, &cur, &launches));
        OFRI_TRY(prof_end(ctx, rec, rows, cols, launches, bulk));
    }
    if (rest > 0) {
        const int Tmax = ofri_hs_fused_max_sweeps(), L = (rest + Tmax - 1) / Tmax;
        ofri_ctx::ProfRec rec2{};
        OFRI_TRY(prof_begin(ctx, rec2));
        int remaining = rest;
        for (int k = 0; k < L; ++k) {
            const int T = remaining < Tmax ? remaining : Tmax;
            const bool last = k == L - 1;
            OFRI_TRY(ofri_hs_fused_launch_pitched(ctx, set.U[cur], set.V[cur], last ? Uout : set.U[cur ^ 1], last ? Vout : set.V[cur ^ 1],
                                                  set.fx, set.fy, set.ft, T, rows, cols, set.P, set.P, last ? cols : set.P, last && accumulate));
            cur ^= 1;
            remaining -= T;
        }
        OFRI_TRY(prof_end(ctx, rec2, rows, cols, L, rest));
    }
    return OFRI_OK;
}

// Widths that are not a multiple of 4: the tile kernel on padded copies (row pitch rounded up to 4 floats, so every row
// is 16-byte aligned and a strip that straddles the right edge has room for its last columns), seven planes in one
// block: two U/V pairs and the scaled coefficients.  Same arguments as hs_iterate_io; niter > 0.
static int hs_iterate_padded(ofri_ctx *ctx, const float *Uin, const float *Vin, float *Uout, float *Vout, const float *fx,
                             const float *fy, const float *ft, float a2, int niter, int rows, int cols, bool frames,
                             float *work = nullptr) {
    const int P = (cols + 3) & ~3;
    const size_t plane = (size_t)rows * P, wbytes = (size_t)cols * sizeof(float), pbytes = (size_t)P * sizeof(float);
    PoolBuf buf(ctx);
    if (!work) OFRI_TRY(buf.get(7 * plane * sizeof(float)));
    float *base = work ? work : buf.as<float>();
    float *U[2] = {base, base + 2 * plane}, *V[2] = {base + plane, base + 3 * plane};
    float *A = base + 4 * plane, *B = base + 5 * plane, *C = base + 6 * plane;
    OFRI_REQUIRE((((size_t)base) & 15) == 0, "pool block is not 16-byte aligned");
    OFRI_TRY(ofri_dev_copy2d(ctx, U[0], pbytes, Uin, wbytes, wbytes, rows));
    OFRI_TRY(ofri_dev_copy2d(ctx, V[0], pbytes, Vin, wbytes, wbytes, rows));
    OFRI_TRY(ofri_hs_scaled_coefs(ctx, frames ? nullptr : fx, fy, ft, fx, fy, A, B, C, a2, rows, cols, P));
    int Tmax = ofri_hs_fused_max_sweeps();
    if (ctx->hs_fuse > 1 && ctx->hs_fuse < Tmax) Tmax = ctx->hs_fuse;
    const int L = (niter + Tmax - 1) / Tmax;
    ofri_ctx::ProfRec rec{};
    OFRI_TRY(prof_begin(ctx, rec));
    int remaining = niter, cur = 0;
    for (int k = 0; k < L; ++k) {
        const int T = remaining < Tmax ? remaining : Tmax;
        OFRI_TRY(ofri_hs_fused_launch_pitched(ctx, U[cur], V[cur], U[cur ^ 1], V[cur ^ 1], A, B, C, T, rows, cols, P, P, P, false));
        cur ^= 1;
        remaining -= T;
    }
    OFRI_TRY(prof_end(ctx, rec, rows, cols, L, niter));
    OFRI_TRY(ofri_dev_copy2d(ctx, Uout, wbytes, U[cur], pbytes, wbytes, rows));
    OFRI_TRY(ofri_dev_copy2d(ctx, Vout, wbytes, V[cur], pbytes, wbytes, rows));
    return OFRI_OK;
}

// niter sweeps from (Uin, Vin) into (Uout, Vout); Uin/Vin may alias Uout/Vout.  Launches ping-pong between
// the output planes and two pooled scratch planes, arranged so that the last launch lands in the output.
// frames: (fx, fy) are the two images computeDerivatives takes and ft is a spare plane; the coefficients are made here.
// work: null (scratch planes come from the context's pool), or caller-provided scratch of 7 * rows * ((cols + 3) & ~3)
// floats, 16-byte aligned; with it the call allocates nothing and stays on the one-sweep and tile kernels (same bits).
static int hs_iterate_io(ofri_ctx *ctx, const float *Uin, const float *Vin, float *Uout, float *Vout, const float *fx,
                         const float *fy, const float *ft, float alpha, int niter, int rows, int cols, bool frames = false,
                         float *work = nullptr) {
    const size_t bytes = (size_t)rows * cols * sizeof(float);
    if (niter == 0) {
        if (Uin != Uout) OFRI_TRY(ofri_dev_copy(ctx, Uout, Uin, bytes));
        if (Vin != Vout) OFRI_TRY(ofri_dev_copy(ctx, Vout, Vin, bytes));
        return OFRI_OK;
    }
    const float a2 = alpha * alpha;         // alpha is a float32 in the reference's compiled signature
    const bool exact = ctx->hs_fuse == -1;  // -1: bit-exact restatement of scipy's float64 accumulation
    if (!frames && !work) {
        const void *q[7] = {Uin, Vin, Uout, Vout, fx, fy, ft};
        const int n_stream = hs_stream_share(ctx, q, 7, niter, rows, cols);
        if (n_stream > 0)
            return hs_stream_path(ctx, Uin, Vin, Uout, Vout, fx, fy, ft, nullptr, nullptr, a2, niter, n_stream, rows, cols);
    }
    if (!exact && ctx->hs_fuse != 1 && (cols & 3) != 0 && cols >= 16 && rows >= 16)
        return hs_iterate_padded(ctx, Uin, Vin, Uout, Vout, fx, fy, ft, a2, niter, rows, cols, frames, work);
    PoolBuf tu(ctx), tv(ctx), ca(ctx), cb(ctx), cc(ctx);
    float *wp[5];                            // two flow planes, three coefficient planes
    if (work) {
        const size_t stride = ((size_t)rows * cols + 3) & ~(size_t)3;
        for (int k = 0; k < 5; ++k) wp[k] = work + k * stride;
    } else {
        OFRI_TRY(tu.get(bytes));
        OFRI_TRY(tv.get(bytes));
        OFRI_TRY(ca.get(bytes)); OFRI_TRY(cb.get(bytes)); OFRI_TRY(cc.get(bytes));
        wp[0] = tu.as<float>(); wp[1] = tv.as<float>(); wp[2] = ca.as<float>(); wp[3] = cb.as<float>(); wp[4] = cc.as<float>();
    }
    if (exact) {
        if (frames) OFRI_TRY(ofri_hs_derivs(ctx, fx, fy, wp[2], wp[3], wp[4], rows, cols));
    } else {                                 // the fast kernels work on the scaled coefficients
        OFRI_TRY(ofri_hs_scaled_coefs(ctx, frames ? nullptr : fx, fy, ft, fx, fy, wp[2], wp[3], wp[4], a2, rows, cols));
    }
    if (frames || !exact) { fx = wp[2]; fy = wp[3]; ft = wp[4]; }
    const void *ptrs[9] = {Uin, Vin, Uout, Vout, fx, fy, ft, wp[0], wp[1]};
    const bool fused = !exact && ctx->hs_fuse != 1 && ofri_hs_fused_ok(ptrs, 9, rows, cols);
    int Tmax = 1;
    if (fused) {
        Tmax = ofri_hs_fused_max_sweeps();
        if (ctx->hs_fuse > 1 && ctx->hs_fuse < Tmax) Tmax = ctx->hs_fuse;
    }
    const int L = (niter + Tmax - 1) / Tmax;
    const bool inplace = (Uin == Uout) || (Vin == Vout) || (Uin == Vout) || (Vin == Uout);
    // launch k writes the output planes when (L-1-k) is even; an in-place call whose first launch would
    // overwrite its own source runs one launch more through the scratch planes instead (final copy)
    const bool detour = inplace && ((L - 1) % 2 == 0);
    ofri_ctx::ProfRec rec{};
    OFRI_TRY(prof_begin(ctx, rec));
    const float *su = Uin, *sv = Vin;
    int remaining = niter;
    for (int k = 0; k < L; ++k) {
        const bool to_out = detour ? ((L - k) % 2 == 0) : ((L - 1 - k) % 2 == 0);
        float *du = to_out ? Uout : wp[0], *dv = to_out ? Vout : wp[1];
        const int T = remaining < Tmax ? remaining : Tmax;
        if (fused) OFRI_TRY(ofri_hs_fused_launch(ctx, su, sv, du, dv, fx, fy, ft, T, rows, cols));
        else if (exact) OFRI_TRY(hs_sweep_launch<true>(ctx, su, sv, fx, fy, ft, du, dv, a2, rows, cols));
        else OFRI_TRY(hs_sweep_launch<false>(ctx, su, sv, fx, fy, ft, du, dv, a2, rows, cols));
        su = du; sv = dv;
        remaining -= T;
    }
    OFRI_TRY(prof_end(ctx, rec, rows, cols, L, niter));
    if (su != Uout) {
        OFRI_TRY(ofri_dev_copy(ctx, Uout, su, bytes));
        OFRI_TRY(ofri_dev_copy(ctx, Vout, sv, bytes));
    }
    return OFRI_OK;
}

extern "C" int ofri_hs_iterate(ofri_ctx *ctx, float *U, float *V, const float *fx, const float *fy,
                               const float *ft, float alpha, int niter, int rows, int cols) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(U && V && fx && fy && ft, "null pointer");
    OFRI_REQUIRE(rows > 0 && cols > 0, "bad shape %d x %d", rows, cols);
    OFRI_REQUIRE(niter >= 0, "niter < 0");
    return hs_iterate_io(ctx, U, V, U, V, fx, fy, ft, alpha, niter, rows, cols);
}

// ofri_hs_compute; work: null, or the scratch of an allocation-free call: 9 * rows * ((cols + 3) & ~3) floats (two zero
// planes for a null start, then hs_iterate_io's seven) and red: ofri_red_blocks() + 2 doubles for the error norms.
static int hs_compute_impl(ofri_ctx *ctx, const float *frame0, const float *frame1, const float *U_in, const float *V_in,
                           float *U_out, float *V_out, float alpha, int niter, int rows, int cols, double *err_host,
                           float *work, double *red) {
    OFRI_REQUIRE(frame0 && frame1 && U_out && V_out, "null pointer");
    OFRI_REQUIRE((U_in == nullptr) == (V_in == nullptr), "U_in and V_in must both be given or both be null");
    OFRI_REQUIRE(U_out != U_in && V_out != V_in && U_out != V_out, "output must not alias input");
    OFRI_REQUIRE(rows > 0 && cols > 0 && niter >= 0, "bad shape or niter");
    size_t n = (size_t)rows * cols, bytes = n * sizeof(float);
    const size_t wplane = (size_t)rows * ((cols + 3) & ~3);
    // HS(im2=frame0, im1=frame1) -> computeDerivatives(im1=frame1, im2=frame0)   (HornSchunck.py:37,73,84)
    const void *q[6] = {frame0, frame1, U_in, V_in, U_out, V_out};
    const int n_stream = work ? 0 : hs_stream_share(ctx, q, 6, niter, rows, cols);
    PoolBuf zu(ctx), zv(ctx);
    if (!U_in && (n_stream == 0 || err_host)) {       // a zero start that somebody has to read: make it
        float *z0, *z1;
        if (work) { z0 = work; z1 = work + wplane; }
        else { OFRI_TRY(zu.get(bytes)); OFRI_TRY(zv.get(bytes)); z0 = zu.as<float>(); z1 = zv.as<float>(); }
        OFRI_TRY(ofri_dev_fill(ctx, z0, 0, bytes));
        OFRI_TRY(ofri_dev_fill(ctx, z1, 0, bytes));
        U_in = z0; V_in = z1;
    }
    if (n_stream > 0) {
        // large field: the derivatives go straight into the pipeline's plane set; a zero start is not materialised
        OFRI_TRY(hs_stream_path(ctx, U_in, V_in, U_out, V_out, nullptr, nullptr, nullptr, frame1, frame0, alpha * alpha, niter,
                                n_stream, rows, cols));
    } else {
        OFRI_TRY(hs_iterate_io(ctx, U_in, V_in, U_out, V_out, frame1, frame0, nullptr, alpha, niter, rows, cols, true,
                               work ? work + 2 * wplane : nullptr));
    }
    if (err_host) {
        double r[2];
        if (red) {
            const int nb = ofri_red_blocks();
            OFRI_TRY(ofri_norm_diff_at(ctx, U_out, U_in, n, red, red + nb));
            OFRI_TRY(ofri_norm_diff_at(ctx, V_out, V_in, n, red, red + nb + 1));
            OFRI_TRY(ofri_read_doubles(ctx, red + nb, 2, r));
        } else {
            OFRI_TRY(ofri_norm_diff_dev(ctx, U_out, U_in, n, 0));
            OFRI_TRY(ofri_norm_diff_dev(ctx, V_out, V_in, n, 1));
            OFRI_TRY(ofri_read_slots(ctx, 0, 2, r));
        }
        *err_host = (r[0] + r[1]) / ((double)rows * (double)cols);
    }
    return OFRI_OK;
}

extern "C" int ofri_hs_compute(ofri_ctx *ctx, const float *frame0, const float *frame1,
                               const float *U_in, const float *V_in, float *U_out, float *V_out,
                               float alpha, int niter, int rows, int cols, double *err_host) {
    OFRI_TRY(ofri_check_ctx(ctx));
    return hs_compute_impl(ctx, frame0, frame1, U_in, V_in, U_out, V_out, alpha, niter, rows, cols, err_host, nullptr, nullptr);
}

// ---------------------------------------------------------------------------------------------
// ofri_hs_compute for callers that must not allocate: every scratch plane lives in one file-scope device array.
// Each GPU the library runs on holds its own copy of the array (at its own address) until the process ends; the host
// only ever uses the per-device address looked up once in ofri_hs_static_setup, never the array's name.
#define HS_STATIC_RED_BYTES 16384           /* ofri_red_blocks() + 2 doubles, rounded up */
#define HS_STATIC_BYTES (HS_STATIC_RED_BYTES + 9 * (size_t)OFRI_HS_STATIC_MAX_PIXELS * sizeof(float))
static __device__ __align__(256) unsigned char scratch[HS_STATIC_BYTES];

#define HS_STATIC_MAX_DEVICES 64
static void *g_static_ptr[HS_STATIC_MAX_DEVICES];
static std::atomic<int> g_static_busy[HS_STATIC_MAX_DEVICES];      // a host thread is inside the entry point on that GPU
static std::mutex g_static_mu;

// Once per device, with that device current (ofri_ctx_create calls this before it makes the context).
int ofri_hs_static_setup(int device) {
    OFRI_REQUIRE(device >= 0 && device < HS_STATIC_MAX_DEVICES, "device %d beyond the %d this library tracks", device, HS_STATIC_MAX_DEVICES);
    std::lock_guard<std::mutex> lock(g_static_mu);
    if (g_static_ptr[device]) return OFRI_OK;
    void *p = nullptr;
    OFRI_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    OFRI_REQUIRE(p && (((size_t)p) & 255) == 0, "scratch array of device %d has no usable address", device);
    OFRI_REQUIRE((size_t)(ofri_red_blocks() + 2) * sizeof(double) <= HS_STATIC_RED_BYTES, "reduction area too small");
    g_static_ptr[device] = p;
    return OFRI_OK;
}

extern "C" int ofri_hs_compute_static(ofri_ctx *ctx, const float *frame0, const float *frame1, const float *U_in,
                                      const float *V_in, float *U_out, float *V_out, float alpha, int niter, int rows,
                                      int cols, double *err_host) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(rows > 0 && cols > 0, "bad shape %d x %d", rows, cols);
    const size_t wplane = (size_t)rows * (((size_t)cols + 3) & ~(size_t)3);
    OFRI_REQUIRE(wplane <= (size_t)OFRI_HS_STATIC_MAX_PIXELS,
                 "ofri_hs_compute_static: %d x %d needs %zu pixels per scratch plane, the limit is %d "
                 "(rows x columns rounded up to a multiple of 4); use ofri_hs_compute for larger fields",
                 rows, cols, wplane, OFRI_HS_STATIC_MAX_PIXELS);
    unsigned char *base;
    {
        std::lock_guard<std::mutex> lock(g_static_mu);
        base = static_cast<unsigned char *>(ctx->device < HS_STATIC_MAX_DEVICES ? g_static_ptr[ctx->device] : nullptr);
    }
    OFRI_REQUIRE(base, "no scratch address for device %d (hipGetSymbolAddress failed when the context was made, or the "
                 "device index is beyond the %d this library tracks)", ctx->device, HS_STATIC_MAX_DEVICES);
    // Two host threads inside this entry point on one GPU would enqueue onto the same scratch planes: the second is
    // refused.  (Calls that have returned but whose kernels still run on another stream are not seen here; ordering
    // those is the caller's, as the header says.)
    int idle = 0;
    OFRI_REQUIRE(g_static_busy[ctx->device].compare_exchange_strong(idle, 1),
                 "ofri_hs_compute_static: another call is in progress on device %d (all calls on a GPU share one scratch array)", ctx->device);
    const bool prof = ctx->prof;            // the timing records create events: not in a call that allocates nothing
    ctx->prof = false;
    const int rc = hs_compute_impl(ctx, frame0, frame1, U_in, V_in, U_out, V_out, alpha, niter, rows, cols, err_host,
                                   reinterpret_cast<float *>(base + HS_STATIC_RED_BYTES), reinterpret_cast<double *>(base));
    ctx->prof = prof;
    g_static_busy[ctx->device].store(0);
    return rc;
}

extern "C" int ofri_hs_compute_acc(ofri_ctx *ctx, const float *frame0, const float *frame1, const float *U_in,
                                   const float *V_in, float *U_acc, float *V_acc, float alpha, int niter, int rows, int cols) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(frame0 && frame1 && U_acc && V_acc, "null pointer");
    OFRI_REQUIRE((U_in == nullptr) == (V_in == nullptr), "U_in and V_in must both be given or both be null");
    OFRI_REQUIRE(U_acc != U_in && V_acc != V_in && U_acc != V_acc && U_acc != V_in && V_acc != U_in, "accumulator must not alias input");
    OFRI_REQUIRE(rows > 0 && cols > 0 && niter >= 0, "bad shape or niter");
    const void *q[6] = {frame0, frame1, U_in, V_in, U_acc, V_acc};
    const int n_stream = hs_stream_share(ctx, q, 6, niter, rows, cols);
    if (n_stream > 0)                          // the last launch adds into the accumulator as it stores
        return hs_stream_path(ctx, U_in, V_in, U_acc, V_acc, nullptr, nullptr, nullptr, frame1, frame0, alpha * alpha, niter,
                              n_stream, rows, cols, true);
    const size_t n = (size_t)rows * cols;
    PoolBuf tu(ctx), tv(ctx);
    OFRI_TRY(tu.get(n * sizeof(float))); OFRI_TRY(tv.get(n * sizeof(float)));
    OFRI_TRY(ofri_hs_compute(ctx, frame0, frame1, U_in, V_in, tu.as<float>(), tv.as<float>(), alpha, niter, rows, cols, nullptr));
    OFRI_TRY(ofri_add_inplace(ctx, U_acc, tu.as<float>(), n));
    return ofri_add_inplace(ctx, V_acc, tv.as<float>(), n);
}

extern "C" int ofri_hs_level(ofri_ctx *ctx, const float *im1, const float *im2, float *U_flow, float *V_flow,
                             const float *taps_host, int ntaps, int warp, float alpha, int niter, int rows, int cols) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(im1 && im2 && U_flow && V_flow && U_flow != V_flow, "null or aliased pointer");
    OFRI_REQUIRE(ntaps == 0 || (taps_host && (ntaps & 1)), "taps: none or an odd count expected, got %d", ntaps);
    OFRI_REQUIRE(rows > 0 && cols > 0 && niter >= 0, "bad shape or niter");
    const void *q[4] = {im1, im2, U_flow, V_flow};
    const int n_stream = hs_stream_share(ctx, q, 4, niter, rows, cols);
    if (n_stream > 0 && ofri_sl_level_ok(ntaps, rows, cols)) {
        const HsLevelFill lf = {warp ? U_flow : nullptr, warp ? V_flow : nullptr, taps_host, ntaps};
        // computeDerivatives takes (im2, im1) of the adapter's (im1, im2)   (HornSchunck.py:37,73,84)
        return hs_stream_path(ctx, nullptr, nullptr, U_flow, V_flow, nullptr, nullptr, nullptr, im2, im1, alpha * alpha, niter,
                              n_stream, rows, cols, warp != 0, &lf);
    }
    // smaller fields, other filter lengths: the same steps one after the other
    const size_t bytes = (size_t)rows * cols * sizeof(float);
    PoolBuf w1(ctx), w2(ctx);
    const float *a = im1, *b = im2;
    if (warp) {
        OFRI_TRY(w1.get(bytes)); OFRI_TRY(w2.get(bytes));
        OFRI_TRY(ofri_warp_pair(ctx, im1, im2, U_flow, V_flow, w1.as<float>(), w2.as<float>(), rows, cols));
        a = w1.as<float>(); b = w2.as<float>();
    }
    PoolBuf f1(ctx), f2(ctx);
    if (ntaps > 0) {
        OFRI_TRY(f1.get(bytes)); OFRI_TRY(f2.get(bytes));
        OFRI_TRY(ofri_gauss_sep(ctx, a, f1.as<float>(), rows, cols, taps_host, ntaps));
        OFRI_TRY(ofri_gauss_sep(ctx, b, f2.as<float>(), rows, cols, taps_host, ntaps));
        a = f1.as<float>(); b = f2.as<float>();
    }
    if (warp) return ofri_hs_compute_acc(ctx, a, b, nullptr, nullptr, U_flow, V_flow, alpha, niter, rows, cols);
    return ofri_hs_compute(ctx, a, b, nullptr, nullptr, U_flow, V_flow, alpha, niter, rows, cols, nullptr);
}

extern "C" int ofri_prof_enable(ofri_ctx *ctx, int on) {
    OFRI_TRY(ofri_check_ctx(ctx));
    for (auto &r : ctx->prof_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    ctx->prof_recs.clear();
    ctx->prof = on != 0;
    return OFRI_OK;
}
extern "C" int ofri_prof_count(ofri_ctx *ctx, int *count) {
    OFRI_REQUIRE(ctx && count, "null argument");
    *count = (int)ctx->prof_recs.size();
    return OFRI_OK;
}
extern "C" int ofri_prof_read(ofri_ctx *ctx, int idx, float *ms, int *launches, int *sweeps, int *rows, int *cols) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(idx >= 0 && idx < (int)ctx->prof_recs.size(), "record index out of range");
    OFRI_REQUIRE(ms && launches && sweeps && rows && cols, "null argument");
    auto &r = ctx->prof_recs[idx];
    OFRI_HIP(hipEventSynchronize(r.b));
    OFRI_HIP(hipEventElapsedTime(ms, r.a, r.b));
    *launches = r.launches; *sweeps = r.sweeps; *rows = r.rows; *cols = r.cols;
    return OFRI_OK;
}

// Flow upsampling by the interpolating bicubic spline of scipy's RectBivariateSpline(kx=3, ky=3, s=0)
// (reference src/GenericPyramidalOpticalFlow.py:152-162).  With s = 0 FITPACK interpolates, which per axis
// is the not-a-knot cubic spline.  Per line of n samples the second derivatives M (knot units) solve
//   M[i-1] + 4 M[i] + M[i+1] = 6 (y[i-1] - 2 y[i] + y[i+1]),  i = 1..n-2,
// closed by M[0] = 2M[1]-M[2], M[n-1] = 2M[n-2]-M[n-3]; the value at knot k + theta is
//   A y[k] + B y[k+1] + alpha M[k] + beta M[k+1],  A = 1-theta, B = theta, alpha = (A^3-A)/6, beta = (B^3-B)/6.
//
// The tensor-product spline is evaluated from four coarse arrays instead of a fine intermediate: with
// Mx = second derivatives of the rows of Y, the rows of the surface at fine row i are the splines through
//   g(c)   = A Y[k,c]  + B Y[k+1,c]  + alpha My[k,c]  + beta My[k+1,c]          (values at the knot columns)
//   gxx(c) = A Mx[k,c] + B Mx[k+1,c] + alpha Mxy[k,c] + beta Mxy[k+1,c]         (their second derivatives)
// because solving along x is linear and acts on every row alike; My, Mxy are the column solves of Y and Mx.
// Two launches: k_spline_xsolve writes Mx (float64, the only intermediate in HBM); k_spline_ysolve_eval
// marches down columns, solves Y and Mx along y in registers and evaluates its tile.  All in float64.
#include "ofri_internal.h"

#include <string>
#include <type_traits>
#include <vector>

// device: cp[m], 1/den[m], m = n-2.  Away from the ends both are one constant: lam = 2 - sqrt(3), the fixed
// point of x -> 1/(4 - x); cp[i] == inv[i] == lam exactly for settle <= i <= m-2.
// Forward-elimination factors of the system in the unknowns M[1..n-2] (rows [6 0] first, [1 4 1] interior,
// [0 6] last): c[0] = 0, c[i] = 1/(4 - c[i-1]); they do not depend on n, reach the fixed point
// lam = 2 - sqrt(3) exactly (in float64) at i = settle, and serve both sweeps: dp[i] = (rhs[i] - dp[i-1]) c[i],
// M[i+1] = dp[i] - c[i] M[i+2].  Only the first and the last row differ (factor 1/6, no sub-diagonal term).
#define SP_NTAB 32
struct SplineSolveTable { const double *ctab; double lam; int settle; };   // ctab: device, SP_NTAB entries
// per axis: k[N] knot interval of a fine index, w[4N] = A, B, alpha, beta, first[n] = fine indices below knot interval k
struct SplineAxisTable { const int *k; const double *w; const int *first; };
// the fields of one launch (blockIdx.z): the two components of a flow are upsampled together
struct SplineFields { const float *src[2]; double *mx[2]; float *dst[2]; float scale[2]; };

static int get_spline_solve(ofri_ctx *ctx, SplineSolveTable *t) {
    double c[SP_NTAB];
    int settle = -1;
    c[0] = 0.0;
    for (int i = 1; i < SP_NTAB; ++i) {
        c[i] = 1.0 / (4.0 - c[i - 1]);
        if (settle < 0 && c[i] == c[i - 1]) settle = i - 1;
    }
    OFRI_REQUIRE(settle > 0, "spline factors do not settle");
    void *pc = ofri_find_table(ctx, "sp_ctab");
    if (!pc) OFRI_TRY(ofri_cached_table(ctx, "sp_ctab", c, sizeof(c), &pc));
    t->ctab = (const double *)pc; t->lam = c[settle]; t->settle = settle;
    return OFRI_OK;
}

static int get_spline_axis(ofri_ctx *ctx, int n, int N, SplineAxisTable *t) {
    std::string key = "sa_" + std::to_string(n) + "_" + std::to_string(N);
    void *pk = ofri_find_table(ctx, key + "k"), *pw = ofri_find_table(ctx, key + "w"), *pf = ofri_find_table(ctx, key + "f");
    if (!pk || !pw || !pf) {
        std::vector<int> k(N), first(n, 0);
        std::vector<double> w(4 * (size_t)N);
        for (long long i = 0; i < N; ++i) {
            long long num = i * n, cap = (long long)(n - 1) * N;      // points past the last knot clamp onto it
            if (num > cap) num = cap;
            long long kk = num / N; if (kk > n - 2) kk = n - 2;
            const double th = (double)(num - kk * N) / (double)N, A = 1.0 - th, B = th;
            k[i] = (int)kk;
            w[4 * i] = A; w[4 * i + 1] = B; w[4 * i + 2] = (A * A * A - A) / 6.0; w[4 * i + 3] = (B * B * B - B) / 6.0;
            if (kk + 1 < n) first[kk + 1] += 1;
        }
        for (int j = 1; j < n; ++j) first[j] += first[j - 1];          // first[k] = #{i : k[i] < k}; first[n-1] = N
        OFRI_TRY(ofri_cached_table(ctx, key + "k", k.data(), N * sizeof(int), &pk));
        OFRI_TRY(ofri_cached_table(ctx, key + "w", w.data(), w.size() * sizeof(double), &pw));
        OFRI_TRY(ofri_cached_table(ctx, key + "f", first.data(), n * sizeof(int), &pf));
    }
    t->k = (const int *)pk; t->w = (const double *)pw; t->first = (const int *)pf;
    return OFRI_OK;
}

// The system is strictly diagonally dominant: a perturbation of the forward recurrence shrinks by
// 1/(4 - c) ~ 0.268 per step, and of the back substitution by c ~ 0.268, so a stretch of a line can be
// solved on its own when both sweeps start SP_WARM samples outside it from a zero state: the start-up error
// is below 0.268^24 ~ 2e-14 of the second derivatives, whose weight in a value is at most 0.064.
#define SP_WARM 24

// Second derivatives at the NK knots a .. a+NK-1 of one line of n samples.  D[j] belongs to knot a-2+j (two
// knots below the stretch are kept for the closing relation at the far end); on return D[2 .. NK+1] are the
// wanted values.  sample(idx) returns y[clamp(idx, 0, n-1)].  S (if CAPTURE) receives y at the NK knots.
// INTERIOR: every unknown the two sweeps touch lies in [settle, n-4], where the factors are the constant lam;
// otherwise they come from the arrays of spline_factors (in LDS; fb = index of their first entry).
// Fully unrolled: D and S live in registers.
struct SplineFactors { const double *iv, *cm, *cf; int fb; };
template <int NK, int PB, bool CAPTURE, bool INTERIOR, class F>
__device__ __forceinline__ void spline_line(double (&D)[NK + 2 + SP_WARM], double *S, int a, int n, double lam,
                                            const SplineFactors &fa, F sample) {
    constexpr int ND = NK + 2 + SP_WARM, NSTEP = ND + SP_WARM;
    using T = decltype(sample(0));
    const int i0 = a - 3;                          // D[j] is unknown i0 + j (unknown i is M[i+1])
    const int s0 = i0 - SP_WARM;                   // step st handles unknown s0 + st and takes in sample s0 + st + 2
    T ring[PB];                                    // PB loads in flight ahead of the recurrence
    const T r0 = sample(s0), r1 = sample(s0 + 1);
#pragma unroll
    for (int u = 0; u < PB; ++u) ring[u] = sample(s0 + 2 + u);
    double y0 = (double)r0, y1 = (double)r1, prev = 0.0;
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
        const int u = st % PB, tt = st - SP_WARM, i = s0 + st;
        const double yn = (double)ring[u];
        if (st + PB < NSTEP) ring[u] = sample(s0 + 2 + st + PB);
        asm volatile("" ::: "memory");             // keeps the loads in step with the recurrence
        if (CAPTURE && tt >= 1 && tt <= NK) S[tt - 1] = yn;
        // branch-free: outside [0, m-1] the factor is 0, so the value is 0 and the chain restarts
        double iv = lam, carry = prev;
        if (!INTERIOR) { iv = fa.iv[i - fa.fb]; carry = fa.cm[i - fa.fb] * prev; }
        const double dp = (6.0 * (y0 - 2.0 * y1 + yn) - carry) * iv;
        prev = dp;
        if (tt >= 0) D[tt] = dp;
        y0 = y1; y1 = yn;
    }
    // back substitution from unknown min(i0 + ND - 1, m - 1), which keeps its forward value; entries outside
    // [0, m-1] are 0 from the forward sweep and stay 0
    double next = 0.0;
#pragma unroll
    for (int j = ND - 1; j >= 0; --j) {
        const int i = i0 + j;
        double cf = j < ND - 1 ? lam : 0.0;
        if (!INTERIOR) cf = j < ND - 1 ? fa.cf[i - fa.fb] : 0.0;
        const double v = D[j] - cf * next;
        D[j] = v; next = v;
    }
    if (INTERIOR) return;
    if (a == 0) D[2] = 2.0 * D[3] - D[4];                                   // M[0] = 2M[1] - M[2]
#pragma unroll
    for (int j = 2; j < NK + 2; ++j)
        if (a - 2 + j == n - 1) D[j] = 2.0 * D[j - 1] - D[j - 2];           // M[n-1] = 2M[n-2] - M[n-3]
}
// Factors of the unknowns fb .. fb+count-1 of a line of n samples, for all threads of the workgroup:
// iv = forward factor (0 outside the system, 1/6 in its first and last row), cm = 1 where the row has a
// sub-diagonal term, cf = back-substitution factor (0 in the first and last row and outside).
__device__ __forceinline__ void spline_factors(double *iv, double *cm, double *cf, int fb, int count, int n,
                                               const double *__restrict__ ctab, int tid, int nthreads) {
    const int m = n - 2;
    for (int e = tid; e < count; e += nthreads) {
        const int i = fb + e;
        const double c = ctab[min(max(i, 0), SP_NTAB - 1)];
        const bool inside = i >= 0 && i <= m - 1, mid = i > 0 && i < m - 1;
        iv[e] = !inside ? 0.0 : (mid ? c : 1.0 / 6.0);
        cm[e] = mid ? 1.0 : 0.0;
        cf[e] = mid ? c : 0.0;
    }
}
// the unknowns touched by the sweeps for the knots a .. a+NK-1 (see spline_line)
template <int NK>
__device__ __forceinline__ bool spline_interior(int a, int n, int settle) {
    return a - 3 - SP_WARM >= settle && a - 3 + NK + 2 + SP_WARM - 1 <= n - 4;
}

// ---- launch 1: Mx = second derivatives along the rows ------------------------------------------------
// A workgroup takes XS_ROWS rows x XS_KNOTS knot columns.  The samples are staged in LDS by coalesced row
// reads (with the warm-up margins, columns clamped to the field), thread = (row, stretch of XS_SUB knots)
// walks its stretch out of LDS, and the results go back through LDS so that the stores are coalesced too.
#define XS_ROWS 32
#define XS_SUB 32
#define XS_NSUB 4
#define XS_KNOTS (XS_SUB * XS_NSUB)
#define XS_LEAD (SP_WARM + 3)                         // staged samples left of the first knot
#define XS_SPAN (XS_KNOTS + 2 * SP_WARM + 4)          // staged samples per row
#define XS_PITCH (XS_SPAN | 1)                        // odd: lanes (rows) fall into different banks
#define XS_OPITCH (XS_KNOTS + 1)
__global__ void __launch_bounds__(XS_ROWS *XS_NSUB)
k_spline_xsolve(SplineFields fl, int n_r, int n_c, SplineSolveTable t) {
    const float *__restrict__ Y = fl.src[blockIdx.z];
    double *__restrict__ Mx = fl.mx[blockIdx.z];
    __shared__ double lds[XS_ROWS * XS_OPITCH];       // first the float samples, then the float64 results
    __shared__ double fac[3][XS_KNOTS + 2 * SP_WARM + 4];
    static_assert(sizeof(float) * XS_ROWS * XS_PITCH <= sizeof(double) * XS_ROWS * XS_OPITCH, "staging fits");
    float *in = reinterpret_cast<float *>(lds);
    const int tid = threadIdx.x, l = tid & (XS_ROWS - 1), q = tid / XS_ROWS;
    // an XCD (workgroups go to the 8 XCDs round-robin by index) walks along rows of tiles: the margins that neighbours share
    // come from its L2
    const int nbx = gridDim.x, lin = blockIdx.y * nbx + blockIdx.x, per = (nbx * (int)gridDim.y) / 8;
    const int til = lin < per * 8 ? (lin % 8) * per + lin / 8 : lin;
    const int cx0 = (til % nbx) * XS_KNOTS, r0 = (til / nbx) * XS_ROWS;
    const int base = cx0 - XS_LEAD;
    const bool inner = spline_interior<XS_KNOTS>(cx0, n_c, t.settle);    // one decision per workgroup
    const SplineFactors fa = {fac[0], fac[1], fac[2], cx0 - 3 - SP_WARM};
    if (!inner) spline_factors(fac[0], fac[1], fac[2], fa.fb, XS_KNOTS + 2 * SP_WARM + 2, n_c, t.ctab, tid, XS_ROWS * XS_NSUB);
    constexpr int NT = XS_ROWS * XS_NSUB, NLD = XS_ROWS * XS_SPAN / NT, LB = 15;
    static_assert(NLD * NT == XS_ROWS * XS_SPAN && NLD % LB == 0, "staging loop shape");
#pragma unroll
    for (int b = 0; b < NLD; b += LB) {               // LB loads in flight per thread, then LB LDS stores
        float v[LB];
#pragma unroll
        for (int u = 0; u < LB; ++u) {
            const int idx = tid + (b + u) * NT, r = idx / XS_SPAN, j = idx - r * XS_SPAN;
            const int gr = min(r0 + r, n_r - 1), gc = min(max(base + j, 0), n_c - 1);
            v[u] = Y[(size_t)gr * n_c + gc];
        }
#pragma unroll
        for (int u = 0; u < LB; ++u) {
            const int idx = tid + (b + u) * NT, r = idx / XS_SPAN, j = idx - r * XS_SPAN;
            in[r * XS_PITCH + j] = v[u];
        }
    }
    __syncthreads();
    double D[XS_SUB + 2 + SP_WARM];
    const int a = cx0 + q * XS_SUB;
    const float *row = in + l * XS_PITCH - base;
    if (inner) spline_line<XS_SUB, 8, false, true>(D, nullptr, a, n_c, t.lam, fa, [&](int idx) { return row[idx]; });
    else spline_line<XS_SUB, 8, false, false>(D, nullptr, a, n_c, t.lam, fa, [&](int idx) { return row[idx]; });
    __syncthreads();
#pragma unroll
    for (int s = 0; s < XS_SUB; ++s) lds[l * XS_OPITCH + q * XS_SUB + s] = D[2 + s];
    __syncthreads();
#pragma unroll 8
    for (int idx = tid; idx < XS_ROWS * XS_KNOTS; idx += XS_ROWS * XS_NSUB) {
        const int r = idx / XS_KNOTS, j = idx - r * XS_KNOTS;
        if (r0 + r < n_r && cx0 + j < n_c) Mx[(size_t)(r0 + r) * n_c + cx0 + j] = lds[r * XS_OPITCH + j];
    }
}

// ---- launch 2: column solves and evaluation ----------------------------------------------------------
// A workgroup takes YE_CELLS knot intervals in y and YE_COLS knot columns (YE_COLS-1 intervals in x); its two
// wavefronts are the two arrays, Y and Mx.  Thread (column, array) reads its column straight from memory
// (lanes = consecutive columns) and solves it in registers.  The evaluation goes by quarters of the tile:
// for every fine row of the quarter the thread puts g (from Y, My) or gxx (from Mx, Mxy) of its column into
// LDS (knot intervals are compile-time positions in the registers; the first two fine rows of an interval are
// straight-line code, further ones - factors above 2 - a loop), and after a barrier every thread turns the
// rows into the output pixels of one fine column.  Row weights of the tile's fine rows are staged in LDS.
#define YE_CELLS 36
#define YE_NK (YE_CELLS + 1)
#define YE_COLS 64
#define YE_NT (2 * YE_COLS)
#ifndef YE_Q
#define YE_Q 9                      // knot intervals per quarter
#endif
#ifndef YE_GR
#define YE_GR 24                    // fine rows per pass over a quarter (one pass up to a factor of 3)
#endif
#ifndef YE_EB
#define YE_EB 6                     // fine rows a thread turns into pixels at a time (a factor of 2 makes 18 rows per quarter: three full turns)
#endif
#define YE_WROWS 96                 // fine rows whose weights are staged at a time
#define YE_PB 12
template <bool SCALE>
__global__ void __launch_bounds__(YE_NT, 2)
k_spline_ysolve_eval(SplineFields fl, int n_r, int n_c, int N_c, SplineSolveTable ts, SplineAxisTable ay, SplineAxisTable ax) {
    const float *__restrict__ Y = fl.src[blockIdx.z];
    const double *__restrict__ Mx = fl.mx[blockIdx.z];
    float *__restrict__ out = fl.dst[blockIdx.z];
    const float post_scale = fl.scale[blockIdx.z];
    static_assert(YE_CELLS % YE_Q == 0 && YE_WROWS >= YE_GR && YE_GR % YE_EB == 0, "tile shape");
    __shared__ double G[YE_GR + 1][2][YE_COLS];       // last row: sink for the rows an interval does not have
    __shared__ double WR[YE_WROWS][4];
    __shared__ double fac[3][YE_NK + 2 + 2 * SP_WARM];
    const int tid = threadIdx.x, arr = __builtin_amdgcn_readfirstlane(tid / YE_COLS), cl = tid % YE_COLS;
    // Workgroups go to the 8 XCDs round-robin by index, and each XCD has its own L2.  Renumber them so that an XCD walks down
    // columns of tiles: the 48 warm-up rows that vertically adjacent tiles share then come from that L2 instead of memory.
    const int nbx = gridDim.x, nby = gridDim.y, lin = blockIdx.y * nbx + blockIdx.x, per = (nbx * nby) / 8;
    const int til = lin < per * 8 ? (lin % 8) * per + lin / 8 : lin;
    const int bx = til / nby, by = til - bx * nby;
    const int tx0 = bx * (YE_COLS - 1), a = by * YE_CELLS;
    const int c = min(tx0 + cl, n_c - 1);
    const int N_r = ay.first[n_r - 1];
    const bool inner = spline_interior<YE_NK>(a, n_r, ts.settle);
    int wbase = ay.first[a];                          // WR holds the fine rows wbase .. wbase + YE_WROWS - 1
    auto stage_rows = [&]() {
        for (int r = tid; r < YE_WROWS; r += YE_NT) {
            const size_t i = (size_t)min(wbase + r, N_r - 1);
            WR[r][0] = ay.w[4 * i]; WR[r][1] = ay.w[4 * i + 1]; WR[r][2] = ay.w[4 * i + 2]; WR[r][3] = ay.w[4 * i + 3];
        }
    };
    stage_rows();
    const SplineFactors fa = {fac[0], fac[1], fac[2], a - 3 - SP_WARM};
    if (!inner) {                                     // near the ends of the columns the factors vary
        spline_factors(fac[0], fac[1], fac[2], fa.fb, YE_NK + 2 + 2 * SP_WARM, n_r, ts.ctab, tid, YE_NT);
        __syncthreads();
    }
    double D[YE_NK + 2 + SP_WARM], S[YE_NK];
    if (arr == 0) {
        const float *p = Y + c;
        if (inner) spline_line<YE_NK, YE_PB, true, true>(D, S, a, n_r, ts.lam, fa, [&](int idx) { return p[(size_t)idx * n_c]; });
        else spline_line<YE_NK, YE_PB, true, false>(D, S, a, n_r, ts.lam, fa, [&](int idx) { return p[(size_t)min(max(idx, 0), n_r - 1) * n_c]; });
    } else {
        const double *p = Mx + c;
        if (inner) spline_line<YE_NK, YE_PB, true, true>(D, S, a, n_r, ts.lam, fa, [&](int idx) { return p[(size_t)idx * n_c]; });
        else spline_line<YE_NK, YE_PB, true, false>(D, S, a, n_r, ts.lam, fa, [&](int idx) { return p[(size_t)min(max(idx, 0), n_r - 1) * n_c]; });
    }
    // the thread's own fine column
    const int jf0 = ax.first[tx0], jf1 = ax.first[min(tx0 + YE_COLS - 1, n_c - 1)];
    const int jmine = min(jf0 + tid, N_c - 1);
    const int cmine = ax.k[jmine] - tx0;
    const double xa = ax.w[4 * (size_t)jmine], xb = ax.w[4 * (size_t)jmine + 1], xal = ax.w[4 * (size_t)jmine + 2],
                 xbe = ax.w[4 * (size_t)jmine + 3];
    // rows i0 .. i0+nr-1 (nr <= YE_EB) from G[g0 ...] to fine column j
    // full: all YE_EB rows exist (no test per row: the rows' reads and arithmetic interleave)
    auto emit = [&](auto full, int j, int cc, double wa, double wb, double wal, double wbe, int g0, int i0, int nr) {
        double v[YE_EB];
#pragma unroll
        for (int r = 0; r < YE_EB; ++r) {
            const double *g = &G[g0 + r][0][cc], *h = &G[g0 + r][1][cc];
            v[r] = wa * g[0] + wb * g[1] + wal * h[0] + wbe * h[1];
        }
        float *o = out + (size_t)i0 * N_c + j;
#pragma unroll
        for (int r = 0; r < YE_EB; ++r) {
            float f = (float)v[r];
            if (SCALE) f = __fmul_rn(f, post_scale);
            if (decltype(full)::value || r < nr) *o = f;
            o += N_c;
        }
    };
    auto row_value = [&](int li, double ya, double yb, double Ma, double Mb) {
        const double *w = WR[min(max(li, 0), YE_WROWS - 1)];
        return w[0] * ya + w[1] * yb + w[2] * Ma + w[3] * Mb;
    };
    __syncthreads();                                  // WR staged
#pragma unroll
    for (int q = 0; q < YE_CELLS / YE_Q; ++q) {
        const int k0 = a + q * YE_Q;                  // first knot interval of the quarter
        if (k0 > n_r - 2) break;
        int f[YE_Q + 1];                              // fine rows of interval k0 + p: f[p] .. f[p+1]-1
#pragma unroll
        for (int p = 0; p <= YE_Q; ++p) f[p] = ay.first[min(k0 + p, n_r - 1)];
        for (int pb = f[0]; pb < f[YE_Q]; pb += YE_GR) {
            const int pe = min(pb + YE_GR, f[YE_Q]);
            if (pe > wbase + YE_WROWS) {              // (factors above 2) move the window of staged row weights
                wbase = pb;
                stage_rows();
                __syncthreads();
            }
            bool more = false;
            {
            // weights of the first two fine rows of an interval, fetched one interval ahead of their use
            double wa[4], wb[4];
            auto fetch = [&](int p) {
                const double *ra = WR[min(max(max(f[p], pb) - wbase, 0), YE_WROWS - 1)];
                const double *rb = WR[min(max(max(f[p], pb) + 1 - wbase, 0), YE_WROWS - 1)];
#pragma unroll
                for (int e = 0; e < 4; ++e) { wa[e] = ra[e]; wb[e] = rb[e]; }
            };
            fetch(0);
#pragma unroll
            for (int p = 0; p < YE_Q; ++p) {
                const int lo = max(f[p], pb), n = min(f[p + 1], pe) - lo;
                const double ya = S[q * YE_Q + p], yb = S[q * YE_Q + p + 1], Ma = D[q * YE_Q + p + 2], Mb = D[q * YE_Q + p + 3];
                const double ga = wa[0] * ya + wa[1] * yb + wa[2] * Ma + wa[3] * Mb;
                const double gb = wb[0] * ya + wb[1] * yb + wb[2] * Ma + wb[3] * Mb;
                if (p + 1 < YE_Q) fetch(p + 1);
                G[n >= 1 ? lo - pb : YE_GR][arr][cl] = ga;
                G[n >= 2 ? lo + 1 - pb : YE_GR][arr][cl] = gb;
                more |= n > 2;
            }
            }
            if (more) {
#pragma unroll
                for (int p = 0; p < YE_Q; ++p) {
                    const int lo = max(f[p], pb), hi = min(f[p + 1], pe);
                    const double ya = S[q * YE_Q + p], yb = S[q * YE_Q + p + 1], Ma = D[q * YE_Q + p + 2], Mb = D[q * YE_Q + p + 3];
                    for (int i = lo + 2; i < hi; ++i) G[i - pb][arr][cl] = row_value(i - wbase, ya, yb, Ma, Mb);
                }
            }
            __syncthreads();
            for (int rb = 0; rb < pe - pb; rb += YE_EB) {
                const int nr = min(YE_EB, pe - pb - rb);
                if (nr == YE_EB) {
                    if (jf0 + tid < jf1) emit(std::true_type(), jf0 + tid, cmine, xa, xb, xal, xbe, rb, pb + rb, nr);
                } else {
                    if (jf0 + tid < jf1) emit(std::false_type(), jf0 + tid, cmine, xa, xb, xal, xbe, rb, pb + rb, nr);
                }
                for (int j = jf0 + tid + YE_NT; j < jf1; j += YE_NT) {     // more than two fine columns per knot interval
                    const double *w = ax.w + 4 * (size_t)j;
                    emit(std::false_type(), j, ax.k[j] - tx0, w[0], w[1], w[2], w[3], rb, pb + rb, nr);
                }
            }
            __syncthreads();                          // G is rewritten by the next pass
        }
    }
}

// nf fields (1 or 2) of the same shapes in one pair of launches
static int spline_upsample_fields(ofri_ctx *ctx, int nf, const float *const *src, float *const *dst, const float *scale, int rows_in,
                                  int cols_in, int rows_out, int cols_out) {
    OFRI_TRY(ofri_check_ctx(ctx));
    for (int f = 0; f < nf; ++f) OFRI_REQUIRE(src[f] && dst[f] && src[f] != dst[f], "null or aliased pointer");
    OFRI_REQUIRE(nf == 1 || (dst[0] != dst[1] && dst[0] != src[1] && dst[1] != src[0]), "aliased pointer");
    OFRI_REQUIRE(rows_in >= 4 && cols_in >= 4, "cubic spline needs at least 4 samples per axis (got %d x %d)", rows_in, cols_in);
    OFRI_REQUIRE(rows_out > 0 && cols_out > 0, "bad output shape");
    OFRI_REQUIRE((long long)rows_in * cols_in < (1LL << 31) && (long long)rows_out * cols_out < (1LL << 33), "field too large");
    SplineSolveTable s0;
    SplineAxisTable a0, a1;
    OFRI_TRY(get_spline_solve(ctx, &s0));
    OFRI_TRY(get_spline_axis(ctx, rows_in, rows_out, &a0));
    OFRI_TRY(get_spline_axis(ctx, cols_in, cols_out, &a1));
    const size_t plane = (size_t)rows_in * cols_in;
    PoolBuf Mx(ctx);
    OFRI_TRY(Mx.get(nf * plane * sizeof(double)));
    SplineFields fl;
    bool scaled = false;
    for (int f = 0; f < 2; ++f) {
        const int g = f < nf ? f : 0;
        fl.src[f] = src[g]; fl.dst[f] = dst[g]; fl.mx[f] = Mx.as<double>() + g * plane; fl.scale[f] = scale[g];
        scaled = scaled || scale[g] != 1.0f;
    }
    {
        dim3 g(ceil_div(cols_in, XS_KNOTS), ceil_div(rows_in, XS_ROWS), nf);
        k_spline_xsolve<<<g, XS_ROWS * XS_NSUB, 0, ctx->stream>>>(fl, rows_in, cols_in, s0);
        OFRI_LAUNCH_CHECK();
    }
    {
        dim3 g(ceil_div(cols_in - 1, YE_COLS - 1), ceil_div(rows_in - 1, YE_CELLS), nf);
        // (a product with 1.0f is the number itself: one scaled field makes both take the multiplication)
        if (scaled) k_spline_ysolve_eval<true><<<g, YE_NT, 0, ctx->stream>>>(fl, rows_in, cols_in, cols_out, s0, a0, a1);
        else k_spline_ysolve_eval<false><<<g, YE_NT, 0, ctx->stream>>>(fl, rows_in, cols_in, cols_out, s0, a0, a1);
        OFRI_LAUNCH_CHECK();
    }
    return OFRI_OK;
}

extern "C" int ofri_spline_upsample(ofri_ctx *ctx, const float *src, int rows_in, int cols_in,
                                    float *dst, int rows_out, int cols_out, float post_scale) {
    return spline_upsample_fields(ctx, 1, &src, &dst, &post_scale, rows_in, cols_in, rows_out, cols_out);
}
extern "C" int ofri_spline_upsample_pair(ofri_ctx *ctx, const float *src_u, const float *src_v, int rows_in, int cols_in,
                                         float *dst_u, float *dst_v, int rows_out, int cols_out, float scale_u, float scale_v) {
    const float *src[2] = {src_u, src_v};
    float *dst[2] = {dst_u, dst_v};
    const float scale[2] = {scale_u, scale_v};
    return spline_upsample_fields(ctx, 2, src, dst, scale, rows_in, cols_in, rows_out, cols_out);
}

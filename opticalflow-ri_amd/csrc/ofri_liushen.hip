// Liu-Shen physics-based optical flow refinement.
// Reference behaviour: src/PhysicsBasedOpticalFlowLiuShen.py:47-158.
//
// All 3x3 stencils have weights that are exact in binary (+-1, +-1/2, +-1/4, 2), so a float64 sum of
// the float32 taps rounded once to float32 reproduces scipy.ndimage.convolve; the element-wise
// algebra around them is float32 with separately rounded products and sums, in the reference's order.
// Axis convention of the reference: "x" is the ROW axis (D differentiates along rows), and u is the
// row-direction component.
#include "ofri_internal.h"

int ofri_max_dev(ofri_ctx *ctx, const float *a, size_t n, int slot);
double *ofri_red_slot(ofri_ctx *ctx, int slot);

#define LS_STREAM_DEPTH 6                 // sweeps per launch of the row-streaming pipeline in automatic mode
#define LS_STREAM_MIN_PIXELS (512 * 512)  // fields below this stay on the one-sweep kernel

// ---- set-up ------------------------------------------------------------------------------------
// normalised images and the five coefficient fields + inverse matrix, one thread per pixel
__global__ void k_ls_setup(const float *__restrict__ im1, const float *__restrict__ im2, const double *__restrict__ maxes,
                           float h, float *__restrict__ IIx, float *__restrict__ IIy, float *__restrict__ II,
                           float *__restrict__ Ixt, float *__restrict__ Iyt, float *__restrict__ B11,
                           float *__restrict__ B12, float *__restrict__ B22, int H, int W) {
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= W || y >= H) return;
    const float m1 = (float)maxes[0], m2 = (float)maxes[1];
    int ym = clampi(y - 1, 0, H - 1), yp = clampi(y + 1, 0, H - 1);
    int xm = clampi(x - 1, 0, W - 1), xp = clampi(x + 1, 0, W - 1);
    auto n1 = [&](int yy, int xx) { return __fdiv_rn(im1[(size_t)yy * W + xx], m1); };
    auto n2 = [&](int yy, int xx) { return __fdiv_rn(im2[(size_t)yy * W + xx], m2); };
    auto dt = [&](int yy, int xx) { return __fsub_rn(n2(yy, xx), n1(yy, xx)); };   // (im2-im1)/dt - f with dt=1, f=0
    float c = n1(y, x);
    float up = n1(ym, x), dn = n1(yp, x), lf = n1(y, xm), rt = n1(y, xp);
    float Dq = (float)(((double)dn - (double)up) * 0.5);            // D: along rows
    float Dtq = (float)(((double)rt - (double)lf) * 0.5);           // D^T: along columns
    size_t o = (size_t)y * W + x;
    IIx[o] = __fmul_rn(c, Dq);
    IIy[o] = __fmul_rn(c, Dtq);
    II[o] = __fmul_rn(c, c);
    float Dd = (float)(((double)dt(yp, x) - (double)dt(ym, x)) * 0.5);
    float Dtd = (float)(((double)dt(y, xp) - (double)dt(y, xm)) * 0.5);
    Ixt[o] = __fmul_rn(c, Dd);
    Iyt[o] = __fmul_rn(c, Dtd);
    // inverse matrix (generate_invmatrix, :47-73)
    int cnt = ((y > 0) + (y < H - 1) + 1) * ((x > 0) + (x < W - 1) + 1) - 1;   // in-bounds neighbours (zero border)
    float hc = __fmul_rn(h, (float)cnt);
    float d2r = (float)(((double)up - 2.0 * (double)c) + (double)dn);
    float d2c = (float)(((double)lf - 2.0 * (double)c) + (double)rt);
    float two_c = __fmul_rn(2.0f, c);
    float A11 = __fsub_rn(__fmul_rn(c, __fsub_rn(d2r, two_c)), hc);
    float A22 = __fsub_rn(__fmul_rn(c, __fsub_rn(d2c, two_c)), hc);
    float mq = (float)(0.25 * ((((double)n1(ym, xm) - (double)n1(ym, xp)) - (double)n1(yp, xm)) + (double)n1(yp, xp)));
    float A12 = __fmul_rn(c, mq);
    float det = __fsub_rn(__fmul_rn(A11, A22), __fmul_rn(A12, A12));
    B11[o] = __fdiv_rn(A22, det);
    B12[o] = __fdiv_rn(-A12, det);
    B22[o] = __fdiv_rn(A11, det);
}

// ---- iteration ---------------------------------------------------------------------------------
struct LsFields { const float *IIx, *IIy, *II, *Ixt, *Iyt, *B11, *B12, *B22; };

// EXACT arithmetic: the reference's own sequence (scipy's float64 stencil sums rounded once, NumPy's separately rounded
// float32 products and sums, left to right).
__device__ __forceinline__ void ls_point(float uN, float uS, float uW, float uE, float uNW, float uNE, float uSW, float uSE, float uC,
                                         float vN, float vS, float vW, float vE, float vNW, float vNE, float vSW, float vSE, float vC,
                                         bool hasN, bool hasS, bool hasW, bool hasE,
                                         float IIx, float IIy, float II, float Ixt, float Iyt, float B11, float B12, float B22,
                                         float h, float &un, float &vn) {
    // stencils in float64, rounded once ('nearest' border values are already substituted by the caller)
    float Du = (float)(((double)uS - (double)uN) * 0.5), Dv = (float)(((double)vS - (double)vN) * 0.5);
    float Dtu = (float)(((double)uE - (double)uW) * 0.5), Dtv = (float)(((double)vE - (double)vW) * 0.5);
    float Fu = (float)((double)uN + (double)uS), Ftv = (float)((double)vW + (double)vE);
    float Mu = (float)(0.25 * ((((double)uNW - (double)uNE) - (double)uSW) + (double)uSE));
    float Mv = (float)(0.25 * ((((double)vNW - (double)vNE) - (double)vSW) + (double)vSE));
    // 8-neighbour sum with ZERO border ('constant'): out-of-image taps contribute nothing
    double hu = 0.0, hv = 0.0;
    if (hasN) { if (hasW) { hu += uNW; hv += vNW; } hu += uN; hv += vN; if (hasE) { hu += uNE; hv += vNE; } }
    if (hasW) { hu += uW; hv += vW; }
    if (hasE) { hu += uE; hv += vE; }
    if (hasS) { if (hasW) { hu += uSW; hv += vSW; } hu += uS; hv += vS; if (hasE) { hu += uSE; hv += vSE; } }
    float Hu = (float)hu, Hv = (float)hv;
    float bu = __fmul_rn(__fmul_rn(2.0f, IIx), Du);
    bu = __fadd_rn(bu, __fmul_rn(IIx, Dtv));
    bu = __fadd_rn(bu, __fmul_rn(IIy, Dv));
    bu = __fadd_rn(bu, __fmul_rn(II, Fu));
    bu = __fadd_rn(bu, __fmul_rn(II, Mv));
    bu = __fadd_rn(bu, __fmul_rn(h, Hu));
    bu = __fadd_rn(bu, Ixt);
    float bv = __fmul_rn(IIy, Du);
    bv = __fadd_rn(bv, __fmul_rn(IIx, Dtu));
    bv = __fadd_rn(bv, __fmul_rn(__fmul_rn(2.0f, IIy), Dtv));
    bv = __fadd_rn(bv, __fmul_rn(II, Mu));
    bv = __fadd_rn(bv, __fmul_rn(II, Ftv));
    bv = __fadd_rn(bv, __fmul_rn(h, Hv));
    bv = __fadd_rn(bv, Iyt);
    un = -__fadd_rn(__fmul_rn(B11, bu), __fmul_rn(B12, bv));
    vn = -__fadd_rn(__fmul_rn(B12, bu), __fmul_rn(B22, bv));
    (void)uC; (void)vC;
}

// FAST arithmetic: float32 throughout, fused multiply-adds, in one pinned order that every fast kernel follows (the
// one-sweep kernel from a pixel's nine neighbours, the row-streaming pipeline from column sums shared between
// neighbouring pixels), so all of them give the same bits.  In terms of the pixel's column c and its left / right
// columns l, r ('nearest' border: l = c or r = c at the image edge; N = C or S = C in the first / last row):
//   s[x] = N[x] + S[x]         d[x] = S[x] - N[x]             (column x of l, c, r)
//   sh[x] = N0[x] + S0[x]      th[x] = sh[x] + C[x]           (N0, S0: 0 outside the image; th[l], th[r]: 0 outside)
//   H = (th[l] + th[r]) + sh[c]      8-neighbour sum with zero border
//   m = d[r] - d[l]  (= 4 M)         e = C[r] - C[l]  (= 2 D^T)      f = C[l] + C[r]  (= F^T)
// The powers of two that scale D, D^T and M are folded into the coefficients (exact).
struct LsCo { float IIx, IIy, II, Ixt, Iyt, B11, B12, B22; };
__device__ __forceinline__ void ls_fast_px(float sU, float dU, float thlU, float thrU, float shU, float dlU, float drU, float clU, float crU,
                                           float sV, float dV, float thlV, float thrV, float shV, float dlV, float drV, float clV, float crV,
                                           const LsCo &k, float h, float &un, float &vn) {
    const float Hu = __fadd_rn(__fadd_rn(thlU, thrU), shU), Hv = __fadd_rn(__fadd_rn(thlV, thrV), shV);
    const float mU = __fsub_rn(drU, dlU), mV = __fsub_rn(drV, dlV);
    const float eU = __fsub_rn(crU, clU), eV = __fsub_rn(crV, clV);
    const float fV = __fadd_rn(clV, crV);
    const float hx = __fmul_rn(0.5f, k.IIx), hy = __fmul_rn(0.5f, k.IIy), q = __fmul_rn(0.25f, k.II);
    float bu = fmaf(k.IIx, dU, k.Ixt);          // 2 IIx Du
    bu = fmaf(hx, eV, bu);                      // IIx D^T v
    bu = fmaf(hy, dV, bu);                      // IIy D v
    bu = fmaf(k.II, sU, bu);                    // II F u
    bu = fmaf(q, mV, bu);                       // II M v
    bu = fmaf(h, Hu, bu);
    float bv = fmaf(hy, dU, k.Iyt);             // IIy D u
    bv = fmaf(hx, eU, bv);                      // IIx D^T u
    bv = fmaf(k.IIy, eV, bv);                   // 2 IIy D^T v
    bv = fmaf(q, mU, bv);                       // II M u
    bv = fmaf(k.II, fV, bv);                    // II F^T v
    bv = fmaf(h, Hv, bv);
    un = -fmaf(k.B11, bu, __fmul_rn(k.B12, bv));
    vn = -fmaf(k.B22, bv, __fmul_rn(k.B12, bu));
    (void)sV;
}
// the same from the nine values around a pixel (N = C / S = C / W = C / E = C already substituted at the border)
__device__ __forceinline__ void ls_fast_point(float uN, float uS, float uW, float uE, float uNW, float uNE, float uSW, float uSE, float uC,
                                              float vN, float vS, float vW, float vE, float vNW, float vNE, float vSW, float vSE, float vC,
                                              bool hasN, bool hasS, bool hasW, bool hasE, const LsCo &k, float h, float &un, float &vn) {
    auto col = [&](float n, float c, float s, bool in, float &sx, float &dx, float &shx, float &thx) {
        sx = __fadd_rn(n, s); dx = __fsub_rn(s, n);
        shx = __fadd_rn(hasN ? n : 0.0f, hasS ? s : 0.0f);
        thx = in ? __fadd_rn(shx, c) : 0.0f;
    };
    float s, d, sh, th, sl, dl, shl, thl, sr, dr, shr, thr;
    float sv, dv, shv, thv, slv, dlv, shlv, thlv, srv, drv, shrv, thrv;
    col(uN, uC, uS, true, s, d, sh, th); col(uNW, uW, uSW, hasW, sl, dl, shl, thl); col(uNE, uE, uSE, hasE, sr, dr, shr, thr);
    col(vN, vC, vS, true, sv, dv, shv, thv); col(vNW, vW, vSW, hasW, slv, dlv, shlv, thlv); col(vNE, vE, vSE, hasE, srv, drv, shrv, thrv);
    ls_fast_px(s, d, thl, thr, sh, dl, dr, uW, uE, sv, dv, thlv, thrv, shv, dlv, drv, vW, vE, k, h, un, vn);
}

__device__ inline double ls_wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// What the stopping rule compares with tol: helper() is compiled with a float32 return type
// (src/PhysicsBasedOpticalFlowLiuShen.py:75-80), so both norms and their sum are float32 and the quotient is rounded
// to float32.  ta, tb: sums of squares of the two differences.
__device__ __forceinline__ double ls_total_error(double ta, double tb, double inv_rc) {
    const float nsum = __fadd_rn((float)sqrt(ta), (float)sqrt(tb));
    return (double)(float)((double)nsum * inv_rc);
}

// Device-side loop state.  Sweeps are launched in blocks of T (1 for the one-sweep kernels); block k reads buffer
// k % 3 and writes buffer (k + 1) % 3 of three rotating (u, v) pairs.  The check of block k runs inside launch k + 1,
// which has then already started on trust: if the rule was met, its output (into a third buffer) is ignored and the
// launches after it exit at once.  If the rule was met by sweep j < T of a block, the answer is the state after j
// sweeps of that block's input, which nothing has overwritten: `redo` one-sweep launches at the end recompute it.
struct LsState {
    double err;             // total_error of the last counted iteration
    int iters;              // counted iterations
    int done[2];            // stopping rule met; the check that runs inside launch k writes done[k & 1]
    int result;             // buffer (0..2) that holds the answer
    int redo, redo_src;     // sweeps to recompute from buffer redo_src at the end (0: none)
};

// Counts the T sweeps of block k-1 from their per-workgroup partial sums, partials[(2 * j + {0, 1}) * nparts + w].
// One workgroup (any size that is a multiple of 64, at most 256 threads) inside launch k.
__device__ void ls_check(const double *__restrict__ partials, int nparts, int T, double inv_rc, double tol, LsState *state, int k) {
    __shared__ double sm[2][4];
    __shared__ int stop;
    const int tid = threadIdx.y * blockDim.x + threadIdx.x, nthr = blockDim.x * blockDim.y, nw = nthr >> 6;
    if (state->done[(k + 1) & 1]) {            // met before: hand the flag on
        if (tid == 0) state->done[k & 1] = 1;
        return;
    }
    for (int j = 0; j < T; ++j) {
        double a = 0.0, b = 0.0;
        const double *pa = partials + (size_t)(2 * j) * nparts, *pb = pa + nparts;
        for (int i = tid; i < nparts; i += nthr) { a += pa[i]; b += pb[i]; }
        a = ls_wave_sum(a);
        b = ls_wave_sum(b);
        if ((tid & 63) == 0) { sm[0][tid >> 6] = a; sm[1][tid >> 6] = b; }
        __syncthreads();
        if (tid == 0) {
            double ta = 0.0, tb = 0.0;
            for (int w = 0; w < nw; ++w) { ta += sm[0][w]; tb += sm[1][w]; }
            const double err = ls_total_error(ta, tb, inv_rc);
            state->err = err;
            state->iters += 1;
            const int met = !(err > tol);
            stop = met;
            if (met) {
                const int src = (k + 2) % 3;               // input buffer of block k-1: (k - 1) % 3
                if (j + 1 < T) { state->redo = j + 1; state->redo_src = src; state->result = (src + 1 + (j & 1)) % 3; }
                else state->result = k % 3;
            }
        }
        __syncthreads();
        if (stop) break;
    }
    if (tid == 0) {
        if (stop) state->done[k & 1] = 1;
        else { state->done[k & 1] = 0; state->result = k % 3; }
    }
}

struct LsBufs { float *u[3], *v[3]; };

// Launch k = sweep k plus the stopping check of sweep k-1.  A 64x4-thread workgroup owns a 64 x 8 pixel tile (2 rows
// per thread); the u/v tile with a one-pixel apron is staged in LDS.  Row 0 of the grid holds no tile: its first
// workgroup runs the check of the sweep before (k > 0) while the tiles of this sweep are under way, so the check costs
// no launch of its own.  The tiles read the flag that the check inside the launch before wrote (done[(k-1) & 1]), never
// the one being written in this launch.  A launch with a 1 x 1 grid is the check alone (after the last sweep).
// redo_i >= 0: a recomputing launch at the end (see LsState): runs only if state->redo > redo_i, on buffers chosen from
// the state, and counts nothing.
#define LS_TX 64
#define LS_TY 8
#define LS_RPT 2
template <bool EXACT>
__global__ void __launch_bounds__(256)
k_ls_sweep(LsBufs bufs, LsFields f, float h, int H, int W, LsState *state, double *__restrict__ partials,
           const double *__restrict__ partials_prev, int nparts_prev, int T_prev, int k, int nblocks, double inv_rc, double tol,
           int redo_i) {
    const float *u, *v;
    float *un, *vn;
    if (redo_i >= 0) {
        if (blockIdx.y == 0 || state->redo <= redo_i) return;
        const int src = state->redo_src, a = redo_i == 0 ? src : (src + 1 + ((redo_i - 1) & 1)) % 3, b = (src + 1 + (redo_i & 1)) % 3;
        u = bufs.u[a]; v = bufs.v[a]; un = bufs.u[b]; vn = bufs.v[b];
    } else {
        if (blockIdx.y == 0) {
            if (blockIdx.x == 0 && k > 0) ls_check(partials_prev, nparts_prev, T_prev, inv_rc, tol, state, k);
            return;
        }
        if (state->done[(k + 1) & 1]) return;
        u = bufs.u[k % 3]; v = bufs.v[k % 3]; un = bufs.u[(k + 1) % 3]; vn = bufs.v[(k + 1) % 3];
    }
    __shared__ float su[LS_TY + 2][LS_TX + 2], sv[LS_TY + 2][LS_TX + 2];
    __shared__ double red[2][4];
    const int bx = blockIdx.x * LS_TX, by = (blockIdx.y - 1) * LS_TY;
    const int tid = threadIdx.y * LS_TX + threadIdx.x;
    for (int i = tid; i < (LS_TY + 2) * (LS_TX + 2); i += 256) {
        int ly = i / (LS_TX + 2), lx = i - ly * (LS_TX + 2);
        int gy = clampi(by + ly - 1, 0, H - 1), gx = clampi(bx + lx - 1, 0, W - 1);   // 'nearest'
        su[ly][lx] = u[(size_t)gy * W + gx];
        sv[ly][lx] = v[(size_t)gy * W + gx];
    }
    __syncthreads();
    const int x = bx + threadIdx.x;
    double du2 = 0.0, dv2 = 0.0;
#pragma unroll
    for (int r = 0; r < LS_RPT; ++r) {
        const int y = by + threadIdx.y + 4 * r;
        if (x < W && y < H) {
            const int lx = threadIdx.x + 1, ly = threadIdx.y + 4 * r + 1;
            size_t o = (size_t)y * W + x;
            float a, b;
            if (EXACT) {
                ls_point(su[ly - 1][lx], su[ly + 1][lx], su[ly][lx - 1], su[ly][lx + 1], su[ly - 1][lx - 1], su[ly - 1][lx + 1],
                         su[ly + 1][lx - 1], su[ly + 1][lx + 1], su[ly][lx],
                         sv[ly - 1][lx], sv[ly + 1][lx], sv[ly][lx - 1], sv[ly][lx + 1], sv[ly - 1][lx - 1], sv[ly - 1][lx + 1],
                         sv[ly + 1][lx - 1], sv[ly + 1][lx + 1], sv[ly][lx],
                         y > 0, y < H - 1, x > 0, x < W - 1,
                         f.IIx[o], f.IIy[o], f.II[o], f.Ixt[o], f.Iyt[o], f.B11[o], f.B12[o], f.B22[o], h, a, b);
            } else {
                const LsCo co{f.IIx[o], f.IIy[o], f.II[o], f.Ixt[o], f.Iyt[o], f.B11[o], f.B12[o], f.B22[o]};
                ls_fast_point(su[ly - 1][lx], su[ly + 1][lx], su[ly][lx - 1], su[ly][lx + 1], su[ly - 1][lx - 1], su[ly - 1][lx + 1],
                              su[ly + 1][lx - 1], su[ly + 1][lx + 1], su[ly][lx],
                              sv[ly - 1][lx], sv[ly + 1][lx], sv[ly][lx - 1], sv[ly][lx + 1], sv[ly - 1][lx - 1], sv[ly - 1][lx + 1],
                              sv[ly + 1][lx - 1], sv[ly + 1][lx + 1], sv[ly][lx],
                              y > 0, y < H - 1, x > 0, x < W - 1, co, h, a, b);
            }
            un[o] = a;
            vn[o] = b;
            double d1 = (double)__fsub_rn(a, su[ly][lx]), d2 = (double)__fsub_rn(b, sv[ly][lx]);
            du2 += d1 * d1;
            dv2 += d2 * d2;
        }
    }
    if (redo_i >= 0) return;
    du2 = ls_wave_sum(du2);
    dv2 = ls_wave_sum(dv2);
    if ((tid & 63) == 0) { red[0][tid >> 6] = du2; red[1][tid >> 6] = dv2; }
    __syncthreads();
    if (tid == 0) {
        size_t bidx = (size_t)(blockIdx.y - 1) * gridDim.x + blockIdx.x;
        partials[bidx] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        partials[nblocks + bidx] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

__global__ void k_ls_init_state(LsState *s) {
    s->err = 1e8; s->iters = 0; s->done[0] = s->done[1] = 0; s->result = 0; s->redo = 0; s->redo_src = 0;
}

__global__ void k_ls_select(LsBufs bufs, float *__restrict__ outu, float *__restrict__ outv, size_t n,
                            const LsState *__restrict__ state) {
    const float *su = bufs.u[state->result], *sv = bufs.v[state->result];
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) { outu[i] = su[i]; outv[i] = sv[i]; }
}

// ---- row-streaming pipeline (fast arithmetic): NL sweeps per launch ------------------------------------------
// Same scheme as the Horn-Schunck pipeline (ofri_hs_stream.hip): a workgroup walks down a strip of 128 columns one
// row per step; half-wavefronts are stages of two sweeps each that keep their rows in registers and hand rows on
// through a two-slot LDS ring; a loader wavefront streams the flow rows and the eight coefficient rows six rows ahead
// into LDS rings, where every stage reads a row's coefficients once for both of its sweeps.  Differences: eight
// coefficient planes (4 KB of LDS per row), so the depth that the LDS allows is 6 or 8 sweeps; the two border rules
// ('nearest' for the differences and averages, zero for the 8-neighbour sum) are applied by the stages themselves -
// first / last row by a workgroup-uniform branch, first / last column by selects in the two outermost strips - since
// a field extended by replication does not stay the replication of itself; every sweep's squared change is summed
// over the pixels the workgroup stores, for the stopping rule.
#define LT_RX 128                        // strip width (32 lanes x 4 pixels)
#define LT_HALO 8                        // columns lost on each side (a multiple of 4, at least the depth)
#define LT_OX (LT_RX - 2 * LT_HALO)      // stored width
#define LT_PF 6                          // rows the loader keeps in flight
template <int NL> struct LtCfg {
    static constexpr int NS = NL / 2;                          // pipeline stages (half-wavefronts)
    static constexpr int LAG = 3 * NS;                         // the last stage's second sweep runs this many rows behind the loader
    static constexpr int CR = LAG + 1;                         // coefficient ring rows: a stage reads rows rA and rA-1
    static constexpr int STAGE_THREADS = (NS * 32 + 63) / 64 * 64;
    static constexpr int THREADS = STAGE_THREADS + 64;         // + the loader wavefront
    static constexpr int LDS = CR * 8 * 32 * 16 + NS * 2 * 2 * 32 * 16;
    static constexpr int PER_CU = (160 * 1024) / LDS;          // workgroups per CU that the LDS allows
};
static_assert(LT_HALO % 4 == 0, "strips start on a multiple of 4 columns");

__device__ __forceinline__ float lt_left(float v) {     // lane-1 (wave_shr:1)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xF, 0xF, true));
}
__device__ __forceinline__ float lt_right(float v) {    // lane+1 (wave_shl:1)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xF, 0xF, true));
}

struct LtRow { float u[4], v[4]; };                 // a lane's four pixels of one row
struct LtCoef { float4 p[8]; };                     // IIx IIy II Ixt Iyt B11 B12 B22 of a lane's four pixels

__device__ __forceinline__ float lt_get(const float4 &q, int k) { return k == 0 ? q.x : (k == 1 ? q.y : (k == 2 ? q.z : q.w)); }

// One sweep of a row: a = row above, c = the row, b = row below.  top / bot: the row is the image's first / last
// (workgroup-uniform); isL / isR: the lane's first / last pixel is the image's first / last column (EDGE strips only);
// pass: copy the row instead (a launch of fewer sweeps than the pipeline is deep); cnt: this lane's pixels of this row
// count towards the sweep's squared change.
template <bool EDGE, bool FULL>
__device__ __forceinline__ LtRow lt_sweep(const LtRow &a, const LtRow &c, const LtRow &b, const LtCoef &g, float h, bool top, bool bot,
                                          bool isL, bool isR, bool pass, bool cnt, float &accu, float &accv) {
    float sU[4], dU[4], shU[4], thU[6], dxU[6], cxU[6], sV[4], dV[4], shV[4], thV[6], dxV[6], cxV[6];
    if (!(top || bot)) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            sU[k] = __fadd_rn(a.u[k], b.u[k]); dU[k] = __fsub_rn(b.u[k], a.u[k]); shU[k] = sU[k];
            sV[k] = __fadd_rn(a.v[k], b.v[k]); dV[k] = __fsub_rn(b.v[k], a.v[k]); shV[k] = sV[k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float nU = top ? c.u[k] : a.u[k], sUx = bot ? c.u[k] : b.u[k], nV = top ? c.v[k] : a.v[k], sVx = bot ? c.v[k] : b.v[k];
            sU[k] = __fadd_rn(nU, sUx); dU[k] = __fsub_rn(sUx, nU);
            sV[k] = __fadd_rn(nV, sVx); dV[k] = __fsub_rn(sVx, nV);
            shU[k] = __fadd_rn(top ? 0.0f : a.u[k], bot ? 0.0f : b.u[k]);
            shV[k] = __fadd_rn(top ? 0.0f : a.v[k], bot ? 0.0f : b.v[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        thU[k + 1] = __fadd_rn(shU[k], c.u[k]); thV[k + 1] = __fadd_rn(shV[k], c.v[k]);
        dxU[k + 1] = dU[k]; dxV[k + 1] = dV[k]; cxU[k + 1] = c.u[k]; cxV[k + 1] = c.v[k];
    }
    thU[0] = lt_left(thU[4]); thU[5] = lt_right(thU[1]); thV[0] = lt_left(thV[4]); thV[5] = lt_right(thV[1]);
    dxU[0] = lt_left(dxU[4]); dxU[5] = lt_right(dxU[1]); dxV[0] = lt_left(dxV[4]); dxV[5] = lt_right(dxV[1]);
    cxU[0] = lt_left(cxU[4]); cxU[5] = lt_right(cxU[1]); cxV[0] = lt_left(cxV[4]); cxV[5] = lt_right(cxV[1]);
    if (EDGE) {
        if (isL) { thU[0] = 0.0f; thV[0] = 0.0f; dxU[0] = dxU[1]; dxV[0] = dxV[1]; cxU[0] = cxU[1]; cxV[0] = cxV[1]; }
        if (isR) { thU[5] = 0.0f; thV[5] = 0.0f; dxU[5] = dxU[4]; dxV[5] = dxV[4]; cxU[5] = cxU[4]; cxV[5] = cxV[4]; }
    }
    LtRow o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const LsCo co{lt_get(g.p[0], k), lt_get(g.p[1], k), lt_get(g.p[2], k), lt_get(g.p[3], k),
                      lt_get(g.p[4], k), lt_get(g.p[5], k), lt_get(g.p[6], k), lt_get(g.p[7], k)};
        ls_fast_px(sU[k], dU[k], thU[k], thU[k + 2], shU[k], dxU[k], dxU[k + 2], cxU[k], cxU[k + 2],
                   sV[k], dV[k], thV[k], thV[k + 2], shV[k], dxV[k], dxV[k + 2], cxV[k], cxV[k + 2], co, h, o.u[k], o.v[k]);
        if (!FULL && pass) { o.u[k] = c.u[k]; o.v[k] = c.v[k]; }
        const float eu = __fsub_rn(o.u[k], c.u[k]), ev = __fsub_rn(o.v[k], c.v[k]);
        accu = cnt ? fmaf(eu, eu, accu) : accu;
        accv = cnt ? fmaf(ev, ev, accv) : accv;
    }
    return o;
}

struct LtStage {                 // registers of one pipeline stage (one lane)
    LtRow in0, in1;              // input rows rA-1, rA
    LtRow mid0, mid1;            // rows rA-2, rA-1 of the stage's first sweep
    float au0, av0, au1, av1;    // squared change of the stage's first / second sweep over this lane's stored pixels
    int rA;                      // row the first sweep produces at this step
    int cslot;                   // coefficient ring slot of row rA
};

template <int NL, bool FULL, bool EDGE>
__device__ __forceinline__ void lt_stage_step(LtStage &st, float4 (&ring)[LtCfg<NL>::NS][2][2][32], float4 (&coef)[LtCfg<NL>::CR][8][32],
                                              int stage, bool live, int l, bool passA, bool passB, bool store_lane, bool own_lane, bool isL,
                                              bool isR, int lane_off, float *__restrict__ rowU, float *__restrict__ rowV, float h, int H,
                                              int ya, int yb) {
    constexpr int NS = LtCfg<NL>::NS, CR = LtCfg<NL>::CR;
    const float4 nu4 = ring[stage][(st.rA + 1) & 1][0][l], nv4 = ring[stage][(st.rA + 1) & 1][1][l];   // newest input row rA+1
    LtCoef g, gprev;             // coefficients of rows rA and rA-1 (the slot before)
    const int pslot = st.cslot == 0 ? CR - 1 : st.cslot - 1;
#pragma unroll
    for (int i = 0; i < 8; ++i) g.p[i] = coef[st.cslot][i][l];
#pragma unroll
    for (int i = 0; i < 8; ++i) gprev.p[i] = coef[pslot][i][l];
    st.cslot = st.cslot + 1 == CR ? 0 : st.cslot + 1;
    LtRow nw;
    nw.u[0] = nu4.x; nw.u[1] = nu4.y; nw.u[2] = nu4.z; nw.u[3] = nu4.w;
    nw.v[0] = nv4.x; nw.v[1] = nv4.y; nw.v[2] = nv4.z; nw.v[3] = nv4.w;
    const int rA = st.rA, rB = st.rA - 1;
    const LtRow mid2 = lt_sweep<EDGE, FULL>(st.in0, st.in1, nw, g, h, rA == 0, rA == H - 1, isL, isR, passA,
                                            own_lane && rA >= ya && rA < yb, st.au0, st.av0);          // row rA of the first sweep
    const LtRow out = lt_sweep<EDGE, FULL>(st.mid0, st.mid1, mid2, gprev, h, rB == 0, rB == H - 1, isL, isR, passB,
                                           own_lane && rB >= ya && rB < yb, st.au1, st.av1);           // row rA-1 of the second sweep
    const float4 ou = make_float4(out.u[0], out.u[1], out.u[2], out.u[3]), ov = make_float4(out.v[0], out.v[1], out.v[2], out.v[3]);
    if (stage < NS - 1) {
        ring[stage + 1][rB & 1][0][l] = ou;
        ring[stage + 1][rB & 1][1][l] = ov;
    } else if (live && store_lane && rB >= ya && rB < yb) {
        *reinterpret_cast<float4 *>(rowU + lane_off) = ou;      // rowU, rowV: row rB of the destination (scalar address)
        *reinterpret_cast<float4 *>(rowV + lane_off) = ov;
    }
    st.in0 = st.in1; st.in1 = nw;
    st.mid0 = st.mid1; st.mid1 = mid2;
    ++st.rA;
}

// The loader wavefront's loop.  Lanes 0-31 carry u and coefficient planes 0-3, lanes 32-63 carry v and planes 4-7.
// Rows outside the image are loaded from the nearest row and lanes beyond the left / right edge from the nearest
// strip of four columns: what they hold is never used by a pixel of the image (the stages apply the border rules).
template <int NL>
__device__ __forceinline__ void lt_loader(const float *__restrict__ u, const float *__restrict__ v, const float *__restrict__ coefs,
                                          long long plane, float4 (&ring)[LtCfg<NL>::NS][2][2][32], float4 (&coef)[LtCfg<NL>::CR][8][32],
                                          int lane64, int x0, int rho0, int nsteps, int H, int W) {
    constexpr int CR = LtCfg<NL>::CR;
    const int lj = lane64 & 31, lset = lane64 >> 5;
    const int lx = clampi(x0 + 4 * lj, 0, W - 4);
    const float *fl = lset ? v : u;
    const float *cf = coefs + (lset ? 4 * plane : 0);
    int lslot = (rho0 + 64 * CR) % CR;
#define LT_LOAD(U_, ROW_)                                                                          \
    {                                                                                              \
        const long long o = (long long)clampi((ROW_), 0, H - 1) * W + lx;                          \
        qa##U_ = *reinterpret_cast<const float4 *>(fl + o);                                        \
        qb##U_ = *reinterpret_cast<const float4 *>(cf + o);                                        \
        qc##U_ = *reinterpret_cast<const float4 *>(cf + plane + o);                                \
        qd##U_ = *reinterpret_cast<const float4 *>(cf + 2 * plane + o);                            \
        qe##U_ = *reinterpret_cast<const float4 *>(cf + 3 * plane + o);                            \
    }
#define LT_Q(U_) float4 qa##U_, qb##U_, qc##U_, qd##U_, qe##U_; LT_LOAD(U_, rho0 + (U_)) asm volatile("" ::: "memory");
    LT_Q(0) LT_Q(1) LT_Q(2) LT_Q(3) LT_Q(4) LT_Q(5)
#undef LT_Q
    static_assert(LT_PF == 6, "the prefetch registers are spelled out for six rows");
#define LT_STEP(U_)                                                                \
    {                                                                              \
        const int rho = rho0 + t0 + (U_);                                          \
        ring[0][rho & 1][lset][lj] = qa##U_;                                       \
        coef[lslot][4 * lset][lj] = qb##U_;                                        \
        coef[lslot][4 * lset + 1][lj] = qc##U_;                                    \
        coef[lslot][4 * lset + 2][lj] = qd##U_;                                    \
        coef[lslot][4 * lset + 3][lj] = qe##U_;                                    \
        lslot = lslot + 1 == CR ? 0 : lslot + 1;                                   \
        LT_LOAD(U_, rho + LT_PF)                                                   \
        __syncthreads();                                                           \
    }
    for (int t0 = 0; t0 < nsteps; t0 += LT_PF) {
        LT_STEP(0) LT_STEP(1) LT_STEP(2) LT_STEP(3) LT_STEP(4) LT_STEP(5)
    }
#undef LT_STEP
#undef LT_LOAD
}

// Block k of the sweep loop: T <= NL sweeps from buffer k % 3 into buffer (k + 1) % 3, plus (workgroup 0) the check of
// block k-1.  coefs: the eight coefficient planes, `plane` floats apart.  partials: [2 * NL][tiles] sums of squares.
template <int NL, bool FULL>
__global__ void __launch_bounds__(LtCfg<NL>::THREADS, LtCfg<NL>::PER_CU)
k_ls_stream(LsBufs bufs, const float *__restrict__ coefs, long long plane, float h, int H, int W, int T, int tiles_x, int chunk_rows,
            LsState *state, double *__restrict__ partials, const double *__restrict__ partials_prev, int nparts_prev, int T_prev,
            int k, double inv_rc, double tol) {
    constexpr int NS = LtCfg<NL>::NS, CR = LtCfg<NL>::CR, LAG = LtCfg<NL>::LAG, ST = LtCfg<NL>::STAGE_THREADS;
    if (blockIdx.x == 0) {
        if (k > 0) ls_check(partials_prev, nparts_prev, T_prev, inv_rc, tol, state, k);
        return;
    }
    if (state->done[(k + 1) & 1]) return;
    __shared__ float4 ring[NS][2][2][32];         // ring[s]: input rows of stage s.  [row & 1][u|v][lane]
    __shared__ float4 coef[CR][8][32];            // [row slot][plane][lane]
    const int tid = threadIdx.x;
    // workgroups go to the 8 XCDs round-robin by index: renumber so that an XCD (one L2) gets neighbouring strips and chunks
    const int nb = gridDim.x - 1, j = blockIdx.x - 1, per = nb / 8;
    const int bid = j < per * 8 ? (j % 8) * per + j / 8 : j;
    const int strip = bid % tiles_x, chunk = bid / tiles_x;
    const int ya = chunk * chunk_rows, yb = min(H, ya + chunk_rows);
    const int x0 = strip * LT_OX - LT_HALO;       // image x of strip column 0
    const int rho0 = ya - NL;                     // first input row of the chunk
    const int nsteps = ((yb - ya) + NL + LAG + LT_PF - 1) / LT_PF * LT_PF;
    const float *u = bufs.u[k % 3], *v = bufs.v[k % 3];
    float *un = bufs.u[(k + 1) % 3], *vn = bufs.v[(k + 1) % 3];
    if (tid >= ST) {
        lt_loader<NL>(u, v, coefs, plane, ring, coef, tid - ST, x0, rho0, nsteps, H, W);
    } else {
        const int stage_id = tid >> 5, l = tid & 31;
        const bool live = stage_id < NS;                              // lanes that pad the last stage wavefront repeat the last stage and write nothing
        const int stage = live ? stage_id : NS - 1;
        LtStage st;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            st.in0.u[q] = st.in0.v[q] = st.in1.u[q] = st.in1.v[q] = 0.f;
            st.mid0.u[q] = st.mid0.v[q] = st.mid1.u[q] = st.mid1.v[q] = 0.f;
        }
        st.au0 = st.av0 = st.au1 = st.av1 = 0.f;
        st.rA = rho0 - 2 - 3 * stage;
        st.cslot = (st.rA + 64 * CR) % CR;
        const bool passA = 2 * stage >= T, passB = 2 * stage + 1 >= T;
        const int lane_x = x0 + 4 * l;
        const bool store_lane = l >= LT_HALO / 4 && l < 32 - LT_HALO / 4 && lane_x < W;
        const bool own_lane = store_lane && live;
        const bool edge = x0 < 0 || x0 + LT_RX > W;
        const bool isL = lane_x == 0, isR = lane_x + 4 == W;
#define LT_STEP(U_, EDGE_)                                                                                            \
    {                                                                                                                 \
        const int ro = rho0 - LAG + t + (U_);             /* row the last stage finishes at this step */              \
        lt_stage_step<NL, FULL, EDGE_>(st, ring, coef, stage, live, l, passA, passB, store_lane, own_lane, isL, isR, lane_x, \
                                       un + (long long)ro * W, vn + (long long)ro * W, h, H, ya, yb);               \
        __syncthreads();                                                                                              \
    }
        // one step per round: spelling out six steps (so that the row registers rotate by renaming, as the Horn-Schunck
        // pipeline does) makes the compiler keep over 300 registers live here
        if (edge) {
            for (int t = 0; t < nsteps; ++t) LT_STEP(0, true)
        } else {
            for (int t = 0; t < nsteps; ++t) LT_STEP(0, false)
        }
#undef LT_STEP
        // the stage's two sweeps: sums over its 32 lanes, one partial per workgroup and sweep
        double s0 = st.au0, s1 = st.av0, s2 = st.au1, s3 = st.av1;
        for (int o = 16; o > 0; o >>= 1) {
            s0 += __shfl_xor(s0, o, 32); s1 += __shfl_xor(s1, o, 32); s2 += __shfl_xor(s2, o, 32); s3 += __shfl_xor(s3, o, 32);
        }
        if (live && l == 0) {
            const size_t np = (size_t)nb;
            partials[(size_t)(4 * stage) * np + bid] = s0;
            partials[(size_t)(4 * stage + 1) * np + bid] = s1;
            partials[(size_t)(4 * stage + 2) * np + bid] = s2;
            partials[(size_t)(4 * stage + 3) * np + bid] = s3;
        }
    }
}

template <int NL>
static int lt_launch(ofri_ctx *ctx, const LsBufs &bufs, const float *coefs, long long plane, float h, int H, int W, int T,
                     LsState *state, double *partials, const double *pprev, int nparts_prev, int T_prev, int k, double inv_rc,
                     double tol, int *nparts) {
    const int tiles_x = ceil_div(W, LT_OX);
    int nchunks = LtCfg<NL>::PER_CU * ofri_ctx_ncu(ctx) / tiles_x;
    if (nchunks < 1) nchunks = 1;
    int chunk_rows = ceil_div(H, nchunks);
    if (chunk_rows < 16) chunk_rows = H < 16 ? H : 16;
    nchunks = ceil_div(H, chunk_rows);
    const int tiles = tiles_x * nchunks;
    if (nparts) *nparts = tiles;
    if (!state) return OFRI_OK;                    // size query
    if (T == NL) k_ls_stream<NL, true><<<tiles + 1, LtCfg<NL>::THREADS, 0, ctx->stream>>>(bufs, coefs, plane, h, H, W, T, tiles_x, chunk_rows, state, partials, pprev, nparts_prev, T_prev, k, inv_rc, tol);
    else k_ls_stream<NL, false><<<tiles + 1, LtCfg<NL>::THREADS, 0, ctx->stream>>>(bufs, coefs, plane, h, H, W, T, tiles_x, chunk_rows, state, partials, pprev, nparts_prev, T_prev, k, inv_rc, tol);
    OFRI_LAUNCH_CHECK();
    return OFRI_OK;
}

extern "C" int ofri_ls_set_mode(ofri_ctx *ctx, int mode) {
    OFRI_REQUIRE(ctx, "null context");
    OFRI_REQUIRE(mode == 0 || mode == 1 || mode == -1 || mode == 6 || mode == 8, "mode: 0 automatic, 1 one sweep per launch, 6 / 8 pipeline depth, -1 exact");
    ctx->ls_mode = mode;
    return OFRI_OK;
}

extern "C" int ofri_ls_compute(ofri_ctx *ctx, const float *im1, const float *im2, const float *u_in, const float *v_in,
                               float *u_out, float *v_out, float h, int maxit, double tol, int rows, int cols,
                               double *err_host, int *iters_host) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(im1 && im2 && u_in && v_in && u_out && v_out, "null pointer");
    OFRI_REQUIRE(rows > 0 && cols > 0 && maxit >= 0, "bad shape or maxit");
    const size_t n = (size_t)rows * cols, bytes = n * sizeof(float);
    const bool exact = ctx->ls_mode == -1;
    // the pipeline needs rows of whole 16-byte groups and pays on fields that fill the GPU; a forced depth takes any field
    const bool streamable = !exact && ctx->ls_mode != 1 && (cols & 3) == 0 && cols >= 8 && (n & 3) == 0;
    const int depth = !streamable ? 1 : (ctx->ls_mode == 8 ? 8 : (ctx->ls_mode == 6 ? 6 : (n >= (size_t)LS_STREAM_MIN_PIXELS ? LS_STREAM_DEPTH : 1)));
    PoolBuf planes(ctx), pp(ctx), st(ctx), parts(ctx);
    OFRI_TRY(planes.get(8 * bytes));
    OFRI_TRY(pp.get(6 * bytes));
    dim3 b(LS_TX, LS_TY / LS_RPT), g(ceil_div(cols, LS_TX), ceil_div(rows, LS_TY));
    const int nblocks = g.x * g.y;
    int ntiles = 0;
    if (depth == 8) OFRI_TRY(lt_launch<8>(ctx, LsBufs{}, nullptr, 0, h, rows, cols, 0, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, &ntiles));
    else if (depth == 6) OFRI_TRY(lt_launch<6>(ctx, LsBufs{}, nullptr, 0, h, rows, cols, 0, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, &ntiles));
    const size_t part_set = depth > 1 ? (size_t)2 * depth * ntiles : (size_t)2 * nblocks;
    OFRI_TRY(st.get(sizeof(LsState)));
    OFRI_TRY(parts.get(2 * part_set * sizeof(double)));     // two sets: a block writes one while the other is checked
    float *P = planes.as<float>();
    LsFields f{P, P + n, P + 2 * n, P + 3 * n, P + 4 * n, P + 5 * n, P + 6 * n, P + 7 * n};
    // im / max(im) per image
    OFRI_TRY(ofri_max_dev(ctx, im1, n, 2));
    OFRI_TRY(ofri_max_dev(ctx, im2, n, 3));
    {
        dim3 bs(64, 4), gs(ceil_div(cols, 64), ceil_div(rows, 4));
        k_ls_setup<<<gs, bs, 0, ctx->stream>>>(im1, im2, ofri_red_slot(ctx, 2), h, P, P + n, P + 2 * n, P + 3 * n, P + 4 * n,
                                               P + 5 * n, P + 6 * n, P + 7 * n, rows, cols);
        OFRI_LAUNCH_CHECK();
    }
    LsBufs bufs;
    for (int i = 0; i < 3; ++i) { bufs.u[i] = pp.as<float>() + 2 * i * n; bufs.v[i] = bufs.u[i] + n; }
    OFRI_HIP(hipMemcpyAsync(bufs.u[0], u_in, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    OFRI_HIP(hipMemcpyAsync(bufs.v[0], v_in, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    LsState *state = st.as<LsState>();
    k_ls_init_state<<<1, 1, 0, ctx->stream>>>(state);
    OFRI_LAUNCH_CHECK();
    const double inv_rc = 1.0 / ((double)rows * (double)cols);
    int k = 0, T_prev = 0, np_prev = 0;
    for (int left = maxit; left > 0; ++k) {
        const int T = left < depth ? left : depth;
        double *pk = parts.as<double>() + (size_t)(k & 1) * part_set, *pprev = parts.as<double>() + (size_t)((k + 1) & 1) * part_set;
        int np = nblocks;
        if (depth == 8) OFRI_TRY(lt_launch<8>(ctx, bufs, P, (long long)n, h, rows, cols, T, state, pk, pprev, np_prev, T_prev, k, inv_rc, tol, &np));
        else if (depth == 6) OFRI_TRY(lt_launch<6>(ctx, bufs, P, (long long)n, h, rows, cols, T, state, pk, pprev, np_prev, T_prev, k, inv_rc, tol, &np));
        else {
            if (exact) k_ls_sweep<true><<<dim3(g.x, g.y + 1), b, 0, ctx->stream>>>(bufs, f, h, rows, cols, state, pk, pprev, np_prev, T_prev, k, nblocks, inv_rc, tol, -1);
            else k_ls_sweep<false><<<dim3(g.x, g.y + 1), b, 0, ctx->stream>>>(bufs, f, h, rows, cols, state, pk, pprev, np_prev, T_prev, k, nblocks, inv_rc, tol, -1);
            OFRI_LAUNCH_CHECK();
        }
        left -= T; T_prev = T; np_prev = np;
    }
    if (k > 0) {            // the check of the last block
        const double *pprev = parts.as<double>() + (size_t)((k + 1) & 1) * part_set;
        k_ls_sweep<false><<<dim3(1, 1), b, 0, ctx->stream>>>(bufs, f, h, rows, cols, state, nullptr, pprev, np_prev, T_prev, k, nblocks, inv_rc, tol, -1);
        OFRI_LAUNCH_CHECK();
    }
    for (int i = 0; i + 1 < depth; ++i) {      // a stop inside a block: recompute its first sweeps (launches that exit at once otherwise)
        k_ls_sweep<false><<<dim3(g.x, g.y + 1), b, 0, ctx->stream>>>(bufs, f, h, rows, cols, state, nullptr, nullptr, 0, 0, 0, nblocks, inv_rc, tol, i);
        OFRI_LAUNCH_CHECK();
    }
    {
        size_t gsz = (n + 255) / 256;
        if (gsz > 4096) gsz = 4096;
        k_ls_select<<<(int)gsz, 256, 0, ctx->stream>>>(bufs, u_out, v_out, n, state);
        OFRI_LAUNCH_CHECK();
    }
    if (err_host || iters_host) {
        LsState hs;
        OFRI_HIP(hipMemcpyAsync(ctx->pinned, state, sizeof(LsState), hipMemcpyDeviceToHost, ctx->stream));
        OFRI_TRY(ofri_stream_wait(ctx));
        memcpy(&hs, ctx->pinned, sizeof(LsState));
        if (err_host) *err_host = hs.iters ? hs.err : 0.0;
        if (iters_host) *iters_host = hs.iters;
    }
    return OFRI_OK;
}

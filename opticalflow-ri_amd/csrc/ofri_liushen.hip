// Liu-Shen physics-based optical flow refinement.
// Reference behaviour: src/PhysicsBasedOpticalFlowLiuShen.py:47-158.
//
// All 3x3 stencils have weights that are exact in binary (+-1, +-1/2, +-1/4, 2), so a float64 sum of
// the float32 taps rounded once to float32 reproduces scipy.ndimage.convolve; the element-wise
// algebra around them is float32 with separately rounded products and sums, in the reference's order (exact mode).
// The default fast mode is float32 with fused multiply-adds in one pinned order (ls_fast_px).
// Axis convention of the reference: "x" is the ROW axis (D differentiates along rows), and u is the
// row-direction component.
#include "ofri_internal.h"
#include "ofri_prio.h"

#define LS_STREAM_DEPTH 6                 // sweeps per launch of the row-streaming pipeline in automatic mode
#define LS_STREAM_MIN_PIXELS (512 * 512)  // fields below this stay on the one-sweep kernel

// Device-side loop state.  Sweeps are launched in blocks of T (1 for the one-sweep kernels).  Buffers: 0..2 rotate
// (block k writes (k + 1) % 3 and, from the second block on, reads k % 3), 3 is the caller's input (read by the first
// block), 4 the caller's output (written by the last block).  The check of block k runs inside launch k + 1, which has
// then already started on trust: if the rule was met, its output is ignored and the launches after it exit at once.
// If the rule was met by sweep j < T of a block, the answer is the state after j sweeps of that block's input, which
// nothing has overwritten: one more launch at the end recomputes it (redo sweeps from buffer redo_src into buffer 4).
struct LsState {
    double err;             // total_error of the last counted iteration
    int iters;              // counted iterations
    int done[2];            // stopping rule met; the check that runs inside launch k writes done[k & 1]
    int result;             // buffer that holds the answer
    int redo, redo_src;     // sweeps to recompute at the end (0: none) and the buffer to start from
};
#define LS_IN 3
#define LS_OUT 4
struct LsBufs { float *u[5], *v[5]; };

// ---- set-up ------------------------------------------------------------------------------------
#define LS_MAX_PARTS 512
#define LSS_TX 64                        // tile of k_ls_setup (its block shape)
#define LSS_TY 4
__device__ inline float ls_wave_max(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, 64));
    return v;
}
// per-workgroup maxima of both images: part[b] and part[LS_MAX_PARTS + b]
__global__ void __launch_bounds__(256) k_ls_max2(const float *__restrict__ a, const float *__restrict__ b, size_t n, float *__restrict__ part) {
    __shared__ float sm[2][4];
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    float ma = -INFINITY, mb = -INFINITY;
    if (((((size_t)a) | ((size_t)b)) & 15) == 0) {          // 16-byte loads, four of each image in flight per thread
        const float4 *a4 = reinterpret_cast<const float4 *>(a), *b4 = reinterpret_cast<const float4 *>(b);
        const size_t n4 = n / 4;
        for (; i + st < n4; i += 2 * st) {
            const float4 p = a4[i], q = a4[i + st], r = b4[i], t = b4[i + st];
            ma = fmaxf(fmaxf(fmaxf(ma, p.x), fmaxf(p.y, p.z)), fmaxf(fmaxf(p.w, q.x), fmaxf(fmaxf(q.y, q.z), q.w)));
            mb = fmaxf(fmaxf(fmaxf(mb, r.x), fmaxf(r.y, r.z)), fmaxf(fmaxf(r.w, t.x), fmaxf(fmaxf(t.y, t.z), t.w)));
        }
        for (; i < n4; i += st) {
            const float4 p = a4[i], r = b4[i];
            ma = fmaxf(fmaxf(fmaxf(ma, p.x), fmaxf(p.y, p.z)), p.w);
            mb = fmaxf(fmaxf(fmaxf(mb, r.x), fmaxf(r.y, r.z)), r.w);
        }
        i = 4 * n4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // the last n % 4 values
    }
    for (; i < n; i += st) { ma = fmaxf(ma, a[i]); mb = fmaxf(mb, b[i]); }
    ma = ls_wave_max(ma); mb = ls_wave_max(mb);
    if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = ma; sm[1][threadIdx.x >> 6] = mb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[blockIdx.x] = fmaxf(fmaxf(sm[0][0], sm[0][1]), fmaxf(sm[0][2], sm[0][3]));
        part[LS_MAX_PARTS + blockIdx.x] = fmaxf(fmaxf(sm[1][0], sm[1][1]), fmaxf(sm[1][2], sm[1][3]));
    }
}

// Normalised images, the five coefficient fields and the inverse matrix, one thread per pixel; every workgroup first
// finishes the two maxima from the partial ones.  Planes written, EXACT: IIx IIy II Ixt Iyt B11 B12 B22 (the reference's
// fields); otherwise the forms the fast arithmetic uses, IIx/2 IIy/2 II/4 Ixt Iyt -B11 -B12 -B22 (exact scalings).
// The first thread also resets the loop state.
template <bool EXACT>
__global__ void __launch_bounds__(256)
k_ls_setup(const float *__restrict__ im1, const float *__restrict__ im2, const float *__restrict__ maxparts, int nparts,
           float h, float *__restrict__ P, size_t plane, int H, int W, LsState *state, int first_src) {
    __shared__ float sm[2][4];
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    {
        float ma = -INFINITY, mb = -INFINITY;
        for (int i = tid; i < nparts; i += 256) { ma = fmaxf(ma, maxparts[i]); mb = fmaxf(mb, maxparts[LS_MAX_PARTS + i]); }
        ma = ls_wave_max(ma); mb = ls_wave_max(mb);
        if ((tid & 63) == 0) { sm[0][tid >> 6] = ma; sm[1][tid >> 6] = mb; }
        __syncthreads();
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
        state->err = 1e8; state->iters = 0; state->done[0] = state->done[1] = 0; state->result = first_src; state->redo = 0; state->redo_src = 0;
    }
    const float m1 = fmaxf(fmaxf(sm[0][0], sm[0][1]), fmaxf(sm[0][2], sm[0][3])), m2 = fmaxf(fmaxf(sm[1][0], sm[1][1]), fmaxf(sm[1][2], sm[1][3]));
    // normalised tile with a one-pixel apron ('nearest' border): every pixel is divided once, not once per neighbour that
    // reads it (the set-up was bound by these divisions)
    __shared__ float t1[LSS_TY + 2][LSS_TX + 2], t2[LSS_TY + 2][LSS_TX + 2];
    const int bx = blockIdx.x * LSS_TX, by = blockIdx.y * LSS_TY;
    {
        constexpr int NT = (LSS_TY + 2) * (LSS_TX + 2), NLD = (NT + 255) / 256;
        float p1[NLD], p2[NLD];
#pragma unroll
        for (int e = 0; e < NLD; ++e) {
            const int i = min(tid + 256 * e, NT - 1), ly = i / (LSS_TX + 2), lx = i - ly * (LSS_TX + 2);
            const size_t o = (size_t)clampi(by - 1 + ly, 0, H - 1) * W + clampi(bx - 1 + lx, 0, W - 1);
            p1[e] = im1[o]; p2[e] = im2[o];
        }
#pragma unroll
        for (int e = 0; e < NLD; ++e) {
            const int i = tid + 256 * e, ly = i / (LSS_TX + 2), lx = i - ly * (LSS_TX + 2);
            if (i < NT) { t1[ly][lx] = __fdiv_rn(p1[e], m1); t2[ly][lx] = __fdiv_rn(p2[e], m2); }
        }
    }
    __syncthreads();
    int x = bx + threadIdx.x;
    int y = by + threadIdx.y;
    if (x >= W || y >= H) return;
    int ym = clampi(y - 1, 0, H - 1), yp = clampi(y + 1, 0, H - 1);
    int xm = clampi(x - 1, 0, W - 1), xp = clampi(x + 1, 0, W - 1);
    auto n1 = [&](int yy, int xx) { return t1[yy - by + 1][xx - bx + 1]; };
    auto n2 = [&](int yy, int xx) { return t2[yy - by + 1][xx - bx + 1]; };
    const float c = n1(y, x), up = n1(ym, x), dn = n1(yp, x), lf = n1(y, xm), rt = n1(y, xp);
    // (im2-im1)/dt - f with dt=1, f=0 at the four neighbours
    const float tu = __fsub_rn(n2(ym, x), up), td = __fsub_rn(n2(yp, x), dn), tl = __fsub_rn(n2(y, xm), lf), tr = __fsub_rn(n2(y, xp), rt);
    float Dq = (float)(((double)dn - (double)up) * 0.5);            // D: along rows
    float Dtq = (float)(((double)rt - (double)lf) * 0.5);           // D^T: along columns
    size_t o = (size_t)y * W + x;
    const float IIx = __fmul_rn(c, Dq), IIy = __fmul_rn(c, Dtq), II = __fmul_rn(c, c);
    float Dd = (float)(((double)td - (double)tu) * 0.5);
    float Dtd = (float)(((double)tr - (double)tl) * 0.5);
    P[3 * plane + o] = __fmul_rn(c, Dd);
    P[4 * plane + o] = __fmul_rn(c, Dtd);
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
    const float B11 = __fdiv_rn(A22, det), B12 = __fdiv_rn(-A12, det), B22 = __fdiv_rn(A11, det);
    if (EXACT) {
        P[o] = IIx; P[plane + o] = IIy; P[2 * plane + o] = II;
        P[5 * plane + o] = B11; P[6 * plane + o] = B12; P[7 * plane + o] = B22;
    } else {
        P[o] = __fmul_rn(0.5f, IIx); P[plane + o] = __fmul_rn(0.5f, IIy); P[2 * plane + o] = __fmul_rn(0.25f, II);
        P[5 * plane + o] = -B11; P[6 * plane + o] = -B12; P[7 * plane + o] = -B22;
    }
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
// and, with the coefficient planes stored as hx = IIx/2, hy = IIy/2, q = II/4, nB = -B (exact scalings, k_ls_setup),
//   bu = Ixt + hx (2 d_u + e_v) + hy d_v + q (4 s_u + m_v) + h H_u          [2 IIx Du + IIx D^T v + IIy Dv + II Fu + II Mv + h Hu + Ixt]
//   bv = Iyt + hy (d_u + 2 e_v) + hx e_u + q (m_u + 4 f_v) + h H_v          [IIy Du + IIx D^T u + 2 IIy D^T v + II Mu + II F^T v + h Hv + Iyt]
//   u' = nB11 bu + nB12 bv,   v' = nB22 bv + nB12 bu.
struct LsCo { float hx, hy, q, Ixt, Iyt, nB11, nB12, nB22; };
__device__ __forceinline__ void ls_fast_px(float sU, float dU, float thlU, float thrU, float shU, float dlU, float drU, float clU, float crU,
                                           float dV, float thlV, float thrV, float shV, float dlV, float drV, float clV, float crV,
                                           const LsCo &k, float h, float &un, float &vn) {
    const float Hu = __fadd_rn(__fadd_rn(thlU, thrU), shU), Hv = __fadd_rn(__fadd_rn(thlV, thrV), shV);
    const float mU = __fsub_rn(drU, dlU), mV = __fsub_rn(drV, dlV);
    const float eU = __fsub_rn(crU, clU), eV = __fsub_rn(crV, clV);
    const float fV = __fadd_rn(clV, crV);
    float bu = fmaf(k.hx, fmaf(2.0f, dU, eV), k.Ixt);
    bu = fmaf(k.hy, dV, bu);
    bu = fmaf(k.q, fmaf(4.0f, sU, mV), bu);
    bu = fmaf(h, Hu, bu);
    float bv = fmaf(k.hy, fmaf(2.0f, eV, dU), k.Iyt);
    bv = fmaf(k.hx, eU, bv);
    bv = fmaf(k.q, fmaf(4.0f, fV, mU), bv);
    bv = fmaf(h, Hv, bv);
    un = fmaf(k.nB11, bu, __fmul_rn(k.nB12, bv));
    vn = fmaf(k.nB22, bv, __fmul_rn(k.nB12, bu));
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
    ls_fast_px(s, d, thl, thr, sh, dl, dr, uW, uE, dv, thlv, thrv, shv, dlv, drv, vW, vE, k, h, un, vn);
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

// Counts the T (at most 8) sweeps of the block before this launch's from their per-workgroup partial sums,
// partials[(2 * j + {0, 1}) * nparts + w]; src / dst: the buffers that block read / wrote.  One workgroup (a multiple of
// 64 and at most 64 * LS_CHECK_WAVES threads: the pipelines run it with five wavefronts, the one-sweep kernel with four)
// inside launch k; its wavefronts sum different sweeps side by side.
struct LsPrev { const double *partials; int nparts, T, src, dst; };
#define LS_CHECK_WAVES 8
__device__ void ls_check(const LsPrev &pv, double inv_rc, double tol, LsState *state, int k) {
    __shared__ double sums[8][LS_CHECK_WAVES][2];        // [sweep][slice][u | v]; a slice index never reaches nw
    const int tid = threadIdx.y * blockDim.x + threadIdx.x, nthr = blockDim.x * blockDim.y, w = tid >> 6, ln = tid & 63;
    const int nw = min(nthr >> 6, LS_CHECK_WAVES);
    if (w >= nw) return;                       // before the barrier below, which only the counted wavefronts reach
    if (state->done[(k + 1) & 1]) {            // met before: hand the flag on
        if (tid == 0) state->done[k & 1] = 1;
        return;
    }
    // G wavefronts share a sweep's partial sums (all of them when the block is one sweep: a launch is not over before
    // its check is, and a field of 4096 x 4096 pixels has 32768 partial sums per sweep)
    const int G = nw >= pv.T ? nw / pv.T : 1, slice = w % G;
    for (int j = w / G; j < pv.T; j += nw / G) {
        double a = 0.0, b = 0.0, a2 = 0.0, b2 = 0.0;
        const double *pa = pv.partials + (size_t)(2 * j) * pv.nparts, *pb = pa + pv.nparts;
        int i = slice * 64 + ln;
        for (; i + 64 * G < pv.nparts; i += 128 * G) { a += pa[i]; b += pb[i]; a2 += pa[i + 64 * G]; b2 += pb[i + 64 * G]; }
        if (i < pv.nparts) { a += pa[i]; b += pb[i]; }
        a = ls_wave_sum(a + a2);
        b = ls_wave_sum(b + b2);
        if (ln == 0) { sums[j][slice][0] = a; sums[j][slice][1] = b; }
    }
    __syncthreads();
    if (tid == 0) {
        int stop = 0;
        const int Gs = nw >= pv.T ? nw / pv.T : 1;
        for (int j = 0; j < pv.T && !stop; ++j) {
            double ta = 0.0, tb = 0.0;
            for (int g = 0; g < Gs; ++g) { ta += sums[j][g][0]; tb += sums[j][g][1]; }
            const double err = ls_total_error(ta, tb, inv_rc);
            state->err = err;
            state->iters += 1;
            if (!(err > tol)) {
                stop = 1;
                if (j + 1 < pv.T) { state->redo = j + 1; state->redo_src = pv.src; state->result = LS_OUT; }
                else state->result = pv.dst;
            }
        }
        if (!stop) state->result = pv.dst;
        state->done[k & 1] = stop;
    }
}

// Launch k = sweep k (from buffer src into buffer dst) plus the stopping check of sweep k-1.  A 64x4-thread workgroup
// owns a 64 x 8 pixel tile (2 rows per thread); the u/v tile with a one-pixel apron is staged in LDS.  Row 0 of the grid
// holds no tile: its first workgroup runs the check of the sweep before (k > 0) while the tiles of this sweep are under
// way, so the check costs no launch of its own.  The tiles read the flag that the check inside the launch before wrote
// (done[(k-1) & 1]), never the one being written in this launch.  A launch with a 1 x 1 grid is the check alone (after
// the last sweep).
#define LS_TX 64
#define LS_TY 8
#define LS_RPT 2
template <bool EXACT>
__global__ void __launch_bounds__(256)
k_ls_sweep(LsBufs bufs, int src, int dst, LsFields f, float h, int H, int W, LsState *state, double *__restrict__ partials,
           LsPrev pv, int k, int nblocks, double inv_rc, double tol) {
    if (blockIdx.y == 0) {
        if (blockIdx.x == 0 && k > 0) ls_check(pv, inv_rc, tol, state, k);
        return;
    }
    if (state->done[(k + 1) & 1]) return;
    const float *__restrict__ u = bufs.u[src], *__restrict__ v = bufs.v[src];
    float *__restrict__ un = bufs.u[dst], *__restrict__ vn = bufs.v[dst];
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
                const LsCo co{f.IIx[o], f.IIy[o], f.II[o], f.Ixt[o], f.Iyt[o], f.B11[o], f.B12[o], f.B22[o]};   // the scaled forms
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

// the answer goes to the caller's arrays, unless the last block or the recomputing launch has put it there
__global__ void k_ls_select(LsBufs bufs, size_t n, const LsState *__restrict__ state) {
    const int r = state->result;
    if (r == LS_OUT) return;
    const float *su = bufs.u[r], *sv = bufs.v[r];
    float *outu = bufs.u[LS_OUT], *outv = bufs.v[LS_OUT];
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) { outu[i] = su[i]; outv[i] = sv[i]; }
}

// ---- row-streaming pipeline (fast arithmetic): NL sweeps per launch ------------------------------------------
// Same scheme as the Horn-Schunck pipeline (ofri_hs_stream.hip): a workgroup walks down a strip of 128 columns one
// row per step; its stages do two sweeps each, keep their rows in registers and hand rows on through a two-slot LDS
// ring; a loader wavefront streams the flow rows and the eight coefficient rows six rows ahead into LDS rings, where
// every stage reads a row's coefficients (and those of the row before) for its two sweeps.  Differences: eight
// coefficient planes (4 KB of LDS per row), so the depth that the LDS allows is 6 or 8 sweeps; a stage is a whole
// wavefront with PX = 2 pixels per lane (or half a wavefront with 4); the two border rules ('nearest' for the
// differences and averages, zero for the 8-neighbour sum) are applied by the stages themselves - first / last row where
// a stage meets it, first / last column by selects in the two outermost strips - since a field extended by replication
// does not stay the replication of itself; every sweep's squared change is summed over the pixels the workgroup
// stores, for the stopping rule.
#define LT_RX 128                        // strip width
#define LT_HALO 8                        // columns lost on each side (a multiple of 4, at least the depth)
#define LT_OX (LT_RX - 2 * LT_HALO)      // stored width
#define LT_PF 6                          // rows the loader keeps in flight
template <int NL, int PX> struct LtCfg {
    static constexpr int NS = NL / 2;                          // pipeline stages
    static constexpr int LN = LT_RX / PX;                      // lanes of a stage
    static constexpr int LAG = 3 * NS;                         // the last stage's second sweep runs this many rows behind the loader
    static constexpr int CR = LAG + 1;                         // coefficient ring rows: the last stage reads a row LAG - 1 steps after the loader wrote it
    static constexpr int STAGE_THREADS = (NS * LN + 63) / 64 * 64;
    static constexpr int THREADS = STAGE_THREADS + 64;         // + the loader wavefront
    static constexpr int LDS = CR * 8 * LT_RX * 4 + NS * 2 * 2 * LT_RX * 4;
    static constexpr int PER_CU = (160 * 1024) / LDS;          // workgroups per CU that the LDS allows
};
static_assert(LT_HALO % 4 == 0, "strips start on a multiple of 4 columns");

__device__ __forceinline__ float lt_left(float v) {     // lane-1 (wave_shr:1)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xF, 0xF, true));
}
__device__ __forceinline__ float lt_right(float v) {    // lane+1 (wave_shl:1)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xF, 0xF, true));
}

template <int PX> struct LtRow { float u[PX], v[PX]; };          // a lane's pixels of one row
template <int PX> struct LtCoef { float p[8][PX]; };             // the eight coefficient planes at a lane's pixels

template <int PX> __device__ __forceinline__ void lt_read(const float *__restrict__ lds, float (&out)[PX]) {
    if constexpr (PX == 4) { const float4 q = *reinterpret_cast<const float4 *>(lds); out[0] = q.x; out[1] = q.y; out[2] = q.z; out[3] = q.w; }
    else { const float2 q = *reinterpret_cast<const float2 *>(lds); out[0] = q.x; out[1] = q.y; }
}
template <int PX> __device__ __forceinline__ void lt_write(float *__restrict__ dst, const float (&v)[PX]) {
    if constexpr (PX == 4) *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
    else *reinterpret_cast<float2 *>(dst) = make_float2(v[0], v[1]);
}

// One sweep of a row: a = row above, c = the row, b = row below.  top / bot: the row is the image's first / last; isL /
// isR: the lane's first / last pixel is the image's first / last column (EDGE strips only); pass: copy the row instead
// (a launch of fewer sweeps than the pipeline is deep); cnt: this lane's pixels of this row count towards the sweep's
// squared change.
template <int PX, bool EDGE, bool FULL>
__device__ __forceinline__ LtRow<PX> lt_sweep(const LtRow<PX> &a, const LtRow<PX> &c, const LtRow<PX> &b, const LtCoef<PX> &g, float h,
                                              bool top, bool bot, bool isL, bool isR, bool pass, bool cnt, float &accu, float &accv) {
    float sU[PX], dU[PX], shU[PX], thU[PX + 2], dxU[PX + 2], cxU[PX + 2], dV[PX], shV[PX], thV[PX + 2], dxV[PX + 2], cxV[PX + 2];
    if (!(top || bot)) {
#pragma unroll
        for (int k = 0; k < PX; ++k) {
            sU[k] = __fadd_rn(a.u[k], b.u[k]); dU[k] = __fsub_rn(b.u[k], a.u[k]); shU[k] = sU[k];
            shV[k] = __fadd_rn(a.v[k], b.v[k]); dV[k] = __fsub_rn(b.v[k], a.v[k]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < PX; ++k) {
            const float nU = top ? c.u[k] : a.u[k], sUx = bot ? c.u[k] : b.u[k], nV = top ? c.v[k] : a.v[k], sVx = bot ? c.v[k] : b.v[k];
            sU[k] = __fadd_rn(nU, sUx); dU[k] = __fsub_rn(sUx, nU);
            dV[k] = __fsub_rn(sVx, nV);
            shU[k] = __fadd_rn(top ? 0.0f : a.u[k], bot ? 0.0f : b.u[k]);
            shV[k] = __fadd_rn(top ? 0.0f : a.v[k], bot ? 0.0f : b.v[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < PX; ++k) {
        thU[k + 1] = __fadd_rn(shU[k], c.u[k]); thV[k + 1] = __fadd_rn(shV[k], c.v[k]);
        dxU[k + 1] = dU[k]; dxV[k + 1] = dV[k]; cxU[k + 1] = c.u[k]; cxV[k + 1] = c.v[k];
    }
    thU[0] = lt_left(thU[PX]); thU[PX + 1] = lt_right(thU[1]); thV[0] = lt_left(thV[PX]); thV[PX + 1] = lt_right(thV[1]);
    dxU[0] = lt_left(dxU[PX]); dxU[PX + 1] = lt_right(dxU[1]); dxV[0] = lt_left(dxV[PX]); dxV[PX + 1] = lt_right(dxV[1]);
    cxU[0] = lt_left(cxU[PX]); cxU[PX + 1] = lt_right(cxU[1]); cxV[0] = lt_left(cxV[PX]); cxV[PX + 1] = lt_right(cxV[1]);
    if (EDGE) {
        if (isL) { thU[0] = 0.0f; thV[0] = 0.0f; dxU[0] = dxU[1]; dxV[0] = dxV[1]; cxU[0] = cxU[1]; cxV[0] = cxV[1]; }
        if (isR) { thU[PX + 1] = 0.0f; thV[PX + 1] = 0.0f; dxU[PX + 1] = dxU[PX]; dxV[PX + 1] = dxV[PX]; cxU[PX + 1] = cxU[PX]; cxV[PX + 1] = cxV[PX]; }
    }
    LtRow<PX> o;
    float ru = 0.0f, rv = 0.0f;                    // this row's squared change
#pragma unroll
    for (int k = 0; k < PX; ++k) {
        const LsCo co{g.p[0][k], g.p[1][k], g.p[2][k], g.p[3][k], g.p[4][k], g.p[5][k], g.p[6][k], g.p[7][k]};
        ls_fast_px(sU[k], dU[k], thU[k], thU[k + 2], shU[k], dxU[k], dxU[k + 2], cxU[k], cxU[k + 2],
                   dV[k], thV[k], thV[k + 2], shV[k], dxV[k], dxV[k + 2], cxV[k], cxV[k + 2], co, h, o.u[k], o.v[k]);
        if (!FULL && pass) { o.u[k] = c.u[k]; o.v[k] = c.v[k]; }
        const float eu = __fsub_rn(o.u[k], c.u[k]), ev = __fsub_rn(o.v[k], c.v[k]);
        ru = k == 0 ? __fmul_rn(eu, eu) : fmaf(eu, eu, ru);
        rv = k == 0 ? __fmul_rn(ev, ev) : fmaf(ev, ev, rv);
    }
    accu = __fadd_rn(accu, cnt ? ru : 0.0f);
    accv = __fadd_rn(accv, cnt ? rv : 0.0f);
    return o;
}

template <int PX> struct LtStage {   // registers of one pipeline stage (one lane)
    LtRow<PX> in0, in1;              // input rows rA-1, rA
    LtRow<PX> mid0, mid1;            // rows rA-2, rA-1 of the stage's first sweep
    LtCoef<PX> gprev;                // coefficients of row rA-1 (read from the ring one step ago)
    float au0, av0, au1, av1;        // squared change of the stage's first / second sweep over this lane's stored pixels
    int rA;                          // row the first sweep produces at this step
    int cslot;                       // coefficient ring slot of row rA
};

template <int NL, int PX, bool FULL, bool EDGE>
__device__ __forceinline__ void lt_stage_step(LtStage<PX> &st, float (&ring)[LtCfg<NL, PX>::NS][2][2][LT_RX], float (&coef)[LtCfg<NL, PX>::CR][8][LT_RX],
                                              int stage, bool live, int l, bool passA, bool passB, bool store_lane, bool own_lane, bool isL,
                                              bool isR, int lane_off, float *__restrict__ rowU, float *__restrict__ rowV, float h, int H,
                                              int ya, int yb) {
    constexpr int NS = LtCfg<NL, PX>::NS, CR = LtCfg<NL, PX>::CR;
    LtRow<PX> nw;                                                       // newest input row rA+1
    lt_read<PX>(&ring[stage][(st.rA + 1) & 1][0][l * PX], nw.u);
    lt_read<PX>(&ring[stage][(st.rA + 1) & 1][1][l * PX], nw.v);
    LtCoef<PX> g;                                                       // coefficients of row rA
#pragma unroll
    for (int i = 0; i < 8; ++i) lt_read<PX>(&coef[st.cslot][i][l * PX], g.p[i]);
    st.cslot = st.cslot + 1 == CR ? 0 : st.cslot + 1;
    const int rA = st.rA, rB = st.rA - 1;
    const LtRow<PX> mid2 = lt_sweep<PX, EDGE, FULL>(st.in0, st.in1, nw, g, h, rA == 0, rA == H - 1, isL, isR, passA,
                                                    own_lane && rA >= ya && rA < yb, st.au0, st.av0);          // row rA of the first sweep
    const LtRow<PX> out = lt_sweep<PX, EDGE, FULL>(st.mid0, st.mid1, mid2, st.gprev, h, rB == 0, rB == H - 1, isL, isR, passB,
                                                   own_lane && rB >= ya && rB < yb, st.au1, st.av1);           // row rA-1 of the second sweep
    if (stage < NS - 1) {
        lt_write<PX>(&ring[stage + 1][rB & 1][0][l * PX], out.u);
        lt_write<PX>(&ring[stage + 1][rB & 1][1][l * PX], out.v);
    } else if (live && store_lane && rB >= ya && rB < yb) {
        lt_write<PX>(rowU + lane_off, out.u);                           // rowU, rowV: row rB of the destination (scalar address)
        lt_write<PX>(rowV + lane_off, out.v);
    }
    st.in0 = st.in1; st.in1 = nw;
    st.mid0 = st.mid1; st.mid1 = mid2;
    st.gprev = g;
    ++st.rA;
}

// The loader wavefront's loop.  Lanes 0-31 carry u and coefficient planes 0-3, lanes 32-63 carry v and planes 4-7, four
// pixels each.  Rows outside the image are loaded from the nearest row and lanes beyond the left / right edge from the
// nearest group of four columns: what they hold is never used by a pixel of the image (the stages apply the border rules).
template <int NL, int PX>
__device__ __forceinline__ void lt_loader(const float *__restrict__ u, const float *__restrict__ v, const float *__restrict__ coefs,
                                          long long plane, float (&ring)[LtCfg<NL, PX>::NS][2][2][LT_RX], float (&coef)[LtCfg<NL, PX>::CR][8][LT_RX],
                                          int lane64, int x0, int rho0, int nsteps, int H, int W, const OfriPrio &pr, int block) {
    constexpr int CR = LtCfg<NL, PX>::CR;
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
#define LT_STEP(U_)                                                                                  \
    {                                                                                                \
        const int rho = rho0 + t0 + (U_);                                                            \
        *reinterpret_cast<float4 *>(&ring[0][rho & 1][lset][4 * lj]) = qa##U_;                       \
        *reinterpret_cast<float4 *>(&coef[lslot][4 * lset][4 * lj]) = qb##U_;                        \
        *reinterpret_cast<float4 *>(&coef[lslot][4 * lset + 1][4 * lj]) = qc##U_;                    \
        *reinterpret_cast<float4 *>(&coef[lslot][4 * lset + 2][4 * lj]) = qd##U_;                    \
        *reinterpret_cast<float4 *>(&coef[lslot][4 * lset + 3][4 * lj]) = qe##U_;                    \
        lslot = lslot + 1 == CR ? 0 : lslot + 1;                                                     \
        LT_LOAD(U_, rho + LT_PF)                                                                     \
        __syncthreads();                                                                             \
    }
    OfriPrioState ps = ofri_prio_begin(pr, nsteps, block);
    for (int t0 = 0; t0 < nsteps; t0 += LT_PF) {
        ofri_prio_turn(t0, pr, ps);
        LT_STEP(0) LT_STEP(1) LT_STEP(2) LT_STEP(3) LT_STEP(4) LT_STEP(5)
    }
#undef LT_STEP
#undef LT_LOAD
}

// Block k of the sweep loop: T <= NL sweeps from buffer src into buffer dst, plus (workgroup 0) the check of block k-1.
// coefs: the eight coefficient planes, `plane` floats apart.  partials: [2 * NL][tiles] sums of squares.
// REDO: the recomputing launch at the end instead (see LsState): T and src come from the state, nothing is counted.
template <int NL, int PX, bool FULL, bool REDO>
__global__ void __launch_bounds__((LtCfg<NL, PX>::THREADS), (LtCfg<NL, PX>::PER_CU))
k_ls_stream(LsBufs bufs, int src, int dst, const float *__restrict__ coefs, long long plane, float h, int H, int W, int T, int tiles_x,
            int chunk_rows, LsState *state, double *__restrict__ partials, LsPrev pv, int k, double inv_rc, double tol, OfriPrio pr) {
    constexpr int NS = LtCfg<NL, PX>::NS, CR = LtCfg<NL, PX>::CR, LAG = LtCfg<NL, PX>::LAG, ST = LtCfg<NL, PX>::STAGE_THREADS, LN = LtCfg<NL, PX>::LN;
    static_assert(LtCfg<NL, PX>::THREADS / 64 <= LS_CHECK_WAVES, "ls_check sizes its LDS for at most LS_CHECK_WAVES wavefronts");
    if (REDO) {
        if (blockIdx.x == 0) return;
        T = state->redo;
        if (T <= 0) return;
        src = state->redo_src;
    } else {
        if (blockIdx.x == 0) {
            if (k > 0) ls_check(pv, inv_rc, tol, state, k);
            return;
        }
        if (state->done[(k + 1) & 1]) return;
    }
    __shared__ __attribute__((aligned(16))) float ring[NS][2][2][LT_RX];      // ring[s]: input rows of stage s.  [row & 1][u|v][column]
    __shared__ __attribute__((aligned(16))) float coef[CR][8][LT_RX];         // [row slot][plane][column]
    const int tid = threadIdx.x;
    // workgroups go to the 8 XCDs round-robin by index: renumber so that an XCD (one L2) gets neighbouring strips and chunks
    const int nb = gridDim.x - 1, j = blockIdx.x - 1, per = nb / 8;
    const int bid = j < per * 8 ? (j % 8) * per + j / 8 : j;
    const int strip = bid % tiles_x, chunk = bid / tiles_x;
    const int ya = chunk * chunk_rows, yb = min(H, ya + chunk_rows);
    const int x0 = strip * LT_OX - LT_HALO;       // image x of strip column 0
    const int rho0 = ya - NL;                     // first input row of the chunk
    const int nsteps = ((yb - ya) + NL + LAG + LT_PF - 1) / LT_PF * LT_PF;
    const float *u = bufs.u[src], *v = bufs.v[src];
    float *un = bufs.u[dst], *vn = bufs.v[dst];
    if (tid >= ST) {
        lt_loader<NL, PX>(u, v, coefs, plane, ring, coef, tid - ST, x0, rho0, nsteps, H, W, pr, j);
    } else {
        const int stage_id = tid / LN, l = tid % LN;
        const bool live = stage_id < NS;                              // lanes that pad the last stage wavefront repeat the last stage and write nothing
        const int stage = live ? stage_id : NS - 1;
        LtStage<PX> st;
#pragma unroll
        for (int q = 0; q < PX; ++q) {
            st.in0.u[q] = st.in0.v[q] = st.in1.u[q] = st.in1.v[q] = 0.f;
            st.mid0.u[q] = st.mid0.v[q] = st.mid1.u[q] = st.mid1.v[q] = 0.f;
        }
        st.au0 = st.av0 = st.au1 = st.av1 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int q = 0; q < PX; ++q) st.gprev.p[i][q] = 0.f;
        st.rA = rho0 - 2 - 3 * stage;
        st.cslot = (st.rA + 64 * CR) % CR;
        const bool passA = 2 * stage >= T, passB = 2 * stage + 1 >= T;
        const int lane_x = x0 + PX * l;
        const bool store_lane = l >= LT_HALO / PX && l < LN - LT_HALO / PX && lane_x < W;
        const bool own_lane = store_lane && live;
        const bool edge = x0 < 0 || x0 + LT_RX > W;
        const bool isL = lane_x == 0, isR = lane_x + PX == W;
#define LT_STEP(EDGE_, U_)                                                                                            \
    {                                                                                                                 \
        const int ro = rho0 - LAG + t + (U_);             /* row the last stage finishes at this step */              \
        lt_stage_step<NL, PX, FULL, EDGE_>(st, ring, coef, stage, live, l, passA, passB, store_lane, own_lane, isL, isR, lane_x, \
                                           un + (long long)ro * W, vn + (long long)ro * W, h, H, ya, yb);           \
        __syncthreads();                                                                                              \
    }
        // two steps per round (nsteps is a multiple of six): the row registers take turns by renaming instead of moves.
        // Spelling out six steps, as the Horn-Schunck pipeline does, makes the compiler keep over 300 registers live here
        OfriPrioState ps = ofri_prio_begin(pr, nsteps, j);
        if (edge) {
            for (int t = 0; t < nsteps; t += 2) { ofri_prio_turn(t, pr, ps); LT_STEP(true, 0) LT_STEP(true, 1) }
        } else {
            for (int t = 0; t < nsteps; t += 2) { ofri_prio_turn(t, pr, ps); LT_STEP(false, 0) LT_STEP(false, 1) }
        }
#undef LT_STEP
        if (!REDO) {
            // the stage's two sweeps: sums over its lanes, one partial per workgroup and sweep
            double s0 = st.au0, s1 = st.av0, s2 = st.au1, s3 = st.av1;
            for (int o = LN / 2; o > 0; o >>= 1) {
                s0 += __shfl_xor(s0, o, LN); s1 += __shfl_xor(s1, o, LN); s2 += __shfl_xor(s2, o, LN); s3 += __shfl_xor(s3, o, LN);
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
}

// grid of the pipeline on an H x W field: strips x row chunks; as many chunks as fill the GPU once with the workgroups
// per CU that the LDS allows
template <int NL, int PX>
static void lt_grid(ofri_ctx *ctx, int H, int W, int *tiles_x, int *chunk_rows, int *tiles) {
    *tiles_x = ceil_div(W, LT_OX);
    int nchunks = LtCfg<NL, PX>::PER_CU * ofri_ctx_ncu(ctx) / *tiles_x / ctx->share;      // share: other contexts fill the rest of the GPU
    if (nchunks < 1) nchunks = 1;
    int cr = ceil_div(H, nchunks);
    if (cr < 16) cr = H < 16 ? H : 16;          // shorter chunks are mostly fill; longer ones leave CUs idle (24 to 64 rows measured no better)
    *chunk_rows = cr;
    *tiles = *tiles_x * ceil_div(H, cr);
}
template <int NL, int PX>
static int lt_launch(ofri_ctx *ctx, const LsBufs &bufs, int src, int dst, const float *coefs, long long plane, float h, int H, int W,
                     int T, LsState *state, double *partials, const LsPrev &pv, int k, double inv_rc, double tol, bool redo) {
    int tiles_x, chunk_rows, tiles;
    lt_grid<NL, PX>(ctx, H, W, &tiles_x, &chunk_rows, &tiles);
    constexpr int THR = LtCfg<NL, PX>::THREADS;
    const int ncu = ofri_ctx_ncu(ctx);
    // rotating issue priority (ofri_prio.h): 60 sweeps at 2048 x 2048 0.85 -> 0.75 ms, at 4096 x 4096 2.57 -> 2.28 ms
    const OfriPrio pr{ncu, tiles > ncu ? 4 : 0, 8};
    if (redo) k_ls_stream<NL, PX, false, true><<<tiles + 1, THR, 0, ctx->stream>>>(bufs, src, dst, coefs, plane, h, H, W, T, tiles_x, chunk_rows, state, partials, pv, k, inv_rc, tol, pr);
    else if (T == NL) k_ls_stream<NL, PX, true, false><<<tiles + 1, THR, 0, ctx->stream>>>(bufs, src, dst, coefs, plane, h, H, W, T, tiles_x, chunk_rows, state, partials, pv, k, inv_rc, tol, pr);
    else k_ls_stream<NL, PX, false, false><<<tiles + 1, THR, 0, ctx->stream>>>(bufs, src, dst, coefs, plane, h, H, W, T, tiles_x, chunk_rows, state, partials, pv, k, inv_rc, tol, pr);
    OFRI_LAUNCH_CHECK();
    return OFRI_OK;
}

#define LS_PX 2         // pixels per lane of the pipeline's stages

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
    OFRI_REQUIRE(u_out != v_out, "u_out and v_out are the same array");
    OFRI_REQUIRE(rows > 0 && cols > 0 && maxit >= 0, "bad shape or maxit");
    const size_t n = (size_t)rows * cols, bytes = n * sizeof(float);
    const bool exact = ctx->ls_mode == -1;
    // the pipeline needs rows of whole 16-byte groups and pays on fields that fill the GPU; a forced depth takes any field
    const bool aligned = ((((size_t)u_in | (size_t)v_in | (size_t)u_out | (size_t)v_out) & 15) == 0);
    const bool streamable = !exact && ctx->ls_mode != 1 && (cols & 3) == 0 && cols >= 8 && aligned;
    const int depth = !streamable ? 1 : (ctx->ls_mode == 8 ? 8 : (ctx->ls_mode == 6 ? 6 : (n >= (size_t)LS_STREAM_MIN_PIXELS ? LS_STREAM_DEPTH : 1)));
    PoolBuf planes(ctx), pp(ctx), st(ctx), parts(ctx), mx(ctx);
    OFRI_TRY(planes.get(8 * bytes));
    OFRI_TRY(pp.get((depth > 1 ? 6 : 4) * bytes));
    OFRI_TRY(mx.get(2 * LS_MAX_PARTS * sizeof(float)));
    dim3 b(LS_TX, LS_TY / LS_RPT), g(ceil_div(cols, LS_TX), ceil_div(rows, LS_TY));
    const int nblocks = g.x * g.y;
    int ntiles = 0, tx = 0, cr = 0;
    if (depth == 8) lt_grid<8, LS_PX>(ctx, rows, cols, &tx, &cr, &ntiles);
    else if (depth == 6) lt_grid<6, LS_PX>(ctx, rows, cols, &tx, &cr, &ntiles);
    const size_t part_set = depth > 1 ? (size_t)2 * depth * ntiles : (size_t)2 * nblocks;
    OFRI_TRY(st.get(sizeof(LsState)));
    OFRI_TRY(parts.get(2 * part_set * sizeof(double)));     // two sets: a block writes one while the other is checked
    float *P = planes.as<float>();
    LsFields f{P, P + n, P + 2 * n, P + 3 * n, P + 4 * n, P + 5 * n, P + 6 * n, P + 7 * n};
    LsState *state = st.as<LsState>();
    LsBufs bufs;
    for (int i = 0; i < 3; ++i) { bufs.u[i] = pp.as<float>() + 2 * (depth > 1 ? i : i & 1) * n; bufs.v[i] = bufs.u[i] + n; }
    bufs.u[LS_IN] = const_cast<float *>(u_in); bufs.v[LS_IN] = const_cast<float *>(v_in);       // read only
    bufs.u[LS_OUT] = u_out; bufs.v[LS_OUT] = v_out;
    // the first block reads the caller's arrays where they are, unless the output would overwrite them
    int first_src = LS_IN;
    if (u_out == u_in || u_out == v_in || v_out == u_in || v_out == v_in) {
        OFRI_TRY(ofri_dev_copy(ctx, bufs.u[0], u_in, bytes));
        OFRI_TRY(ofri_dev_copy(ctx, bufs.v[0], v_in, bytes));
        first_src = 0;
    }
    {   // im / max(im) per image, coefficient fields, loop state
        int gm = (int)((n + 255) / 256);
        if (gm > LS_MAX_PARTS) gm = LS_MAX_PARTS;
        k_ls_max2<<<gm, 256, 0, ctx->stream>>>(im1, im2, n, mx.as<float>());
        OFRI_LAUNCH_CHECK();
        dim3 bs(LSS_TX, LSS_TY), gs(ceil_div(cols, LSS_TX), ceil_div(rows, LSS_TY));
        if (exact) k_ls_setup<true><<<gs, bs, 0, ctx->stream>>>(im1, im2, mx.as<float>(), gm, h, P, n, rows, cols, state, first_src);
        else k_ls_setup<false><<<gs, bs, 0, ctx->stream>>>(im1, im2, mx.as<float>(), gm, h, P, n, rows, cols, state, first_src);
        OFRI_LAUNCH_CHECK();
    }
    const double inv_rc = 1.0 / ((double)rows * (double)cols);
    int k = 0;
    const int rot = depth > 1 ? 3 : 2;
    LsPrev pv{nullptr, 0, 0, 0, 0};
    for (int left = maxit; left > 0; ++k) {
        const int T = left < depth ? left : depth;
        // one-sweep blocks never stop inside a block, so nothing has to outlive the launch after it: two rotating pairs
        // (at 2048 x 2048 the third pair pushes the working set out of the 256 MB last-level cache)
        const int src = k == 0 ? first_src : k % rot, dst = left == T ? LS_OUT : (k + 1) % rot;
        double *pk = parts.as<double>() + (size_t)(k & 1) * part_set;
        if (depth == 8) OFRI_TRY((lt_launch<8, LS_PX>(ctx, bufs, src, dst, P, (long long)n, h, rows, cols, T, state, pk, pv, k, inv_rc, tol, false)));
        else if (depth == 6) OFRI_TRY((lt_launch<6, LS_PX>(ctx, bufs, src, dst, P, (long long)n, h, rows, cols, T, state, pk, pv, k, inv_rc, tol, false)));
        else {
            if (exact) k_ls_sweep<true><<<dim3(g.x, g.y + 1), b, 0, ctx->stream>>>(bufs, src, dst, f, h, rows, cols, state, pk, pv, k, nblocks, inv_rc, tol);
            else k_ls_sweep<false><<<dim3(g.x, g.y + 1), b, 0, ctx->stream>>>(bufs, src, dst, f, h, rows, cols, state, pk, pv, k, nblocks, inv_rc, tol);
            OFRI_LAUNCH_CHECK();
        }
        pv = LsPrev{pk, depth > 1 ? ntiles : nblocks, T, src, dst};
        left -= T;
    }
    if (k > 0) {            // the check of the last block
        k_ls_sweep<false><<<dim3(1, 1), b, 0, ctx->stream>>>(bufs, 0, 0, f, h, rows, cols, state, nullptr, pv, k, nblocks, inv_rc, tol);
        OFRI_LAUNCH_CHECK();
    }
    // a stop inside a block: recompute its first sweeps into the output (a launch that exits at once otherwise)
    if (depth == 8) OFRI_TRY((lt_launch<8, LS_PX>(ctx, bufs, 0, LS_OUT, P, (long long)n, h, rows, cols, 0, state, nullptr, pv, 0, inv_rc, tol, true)));
    else if (depth == 6) OFRI_TRY((lt_launch<6, LS_PX>(ctx, bufs, 0, LS_OUT, P, (long long)n, h, rows, cols, 0, state, nullptr, pv, 0, inv_rc, tol, true)));
    {
        size_t gsz = (n + 255) / 256;
        if (gsz > 4096) gsz = 4096;
        k_ls_select<<<(int)gsz, 256, 0, ctx->stream>>>(bufs, n, state);
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

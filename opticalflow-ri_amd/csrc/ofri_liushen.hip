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

struct LsState {            // lives in device memory
    double err;             // total_error of the last executed iteration
    int iters;              // executed iterations
    int done[2];            // stopping rule met; the check that runs inside launch k writes done[k & 1] (see k_ls_sweep)
};

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

__device__ inline double ls_wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// total_error = (|du| + |dv|) / (rows*cols) of sweep k-1 from its per-workgroup partial sums; stop when it is <= tol
// (reference loop: while err > tol and k < maxnum).  Run by one workgroup of 256 threads inside launch k.
__device__ void ls_check(const double *__restrict__ partials, int nblocks, double inv_rc, double tol, LsState *state, int k) {
    __shared__ double sm[2][4];
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    if (state->done[(k + 1) & 1]) {            // met before: hand the flag on
        if (tid == 0) state->done[k & 1] = 1;
        return;
    }
    double a = 0.0, b = 0.0;
    const double2 *p2 = reinterpret_cast<const double2 *>(partials);
    for (int i = tid; i < nblocks; i += 256) { double2 q = p2[i]; a += q.x; b += q.y; }
    a = ls_wave_sum(a);
    b = ls_wave_sum(b);
    if ((tid & 63) == 0) { sm[0][tid >> 6] = a; sm[1][tid >> 6] = b; }
    __syncthreads();
    if (tid == 0) {
        const double ta = (sm[0][0] + sm[0][1]) + (sm[0][2] + sm[0][3]), tb = (sm[1][0] + sm[1][1]) + (sm[1][2] + sm[1][3]);
        const double err = (sqrt(ta) + sqrt(tb)) * inv_rc;
        state->err = err;
        state->iters += 1;
        state->done[k & 1] = !(err > tol);
    }
}

// Launch k = sweep k plus the stopping check of sweep k-1.  A 64x4-thread workgroup owns a 64 x 8 pixel tile (2 rows
// per thread); the u/v tile with a one-pixel apron is staged in LDS.  Row 0 of the grid holds no tile: its first
// workgroup runs the check of the sweep before (k > 0) while the tiles of this sweep are under way, so the check costs
// no launch of its own.  Sweep k is therefore started without knowing whether sweep k-1 met the stopping rule: if it
// did, this sweep's output (into the other buffer of the pair) is ignored, and the launches after it exit at once.
// The tiles read the flag that the check inside the launch before wrote (done[(k-1) & 1]), never the one being
// written in this launch.  A launch with a 1 x 1 grid is the check alone (after the last sweep).
#define LS_TX 64
#define LS_TY 8
#define LS_RPT 2
__global__ void __launch_bounds__(256)
k_ls_sweep(const float *__restrict__ u, const float *__restrict__ v, float *__restrict__ un, float *__restrict__ vn,
           LsFields f, float h, int H, int W, LsState *state, double *__restrict__ partials,
           const double *__restrict__ partials_prev, int k, int nblocks, double inv_rc, double tol) {
    if (blockIdx.y == 0) {
        if (blockIdx.x == 0 && k > 0) ls_check(partials_prev, nblocks, inv_rc, tol, state, k);
        return;
    }
    if (state->done[(k + 1) & 1]) return;
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
            ls_point(su[ly - 1][lx], su[ly + 1][lx], su[ly][lx - 1], su[ly][lx + 1], su[ly - 1][lx - 1], su[ly - 1][lx + 1],
                     su[ly + 1][lx - 1], su[ly + 1][lx + 1], su[ly][lx],
                     sv[ly - 1][lx], sv[ly + 1][lx], sv[ly][lx - 1], sv[ly][lx + 1], sv[ly - 1][lx - 1], sv[ly - 1][lx + 1],
                     sv[ly + 1][lx - 1], sv[ly + 1][lx + 1], sv[ly][lx],
                     y > 0, y < H - 1, x > 0, x < W - 1,
                     f.IIx[o], f.IIy[o], f.II[o], f.Ixt[o], f.Iyt[o], f.B11[o], f.B12[o], f.B22[o], h, a, b);
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
        partials[2 * bidx] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        partials[2 * bidx + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

__global__ void k_ls_init_state(LsState *s) { s->err = 1e8; s->iters = 0; s->done[0] = s->done[1] = 0; }

// the result sits in buffer (iters & 1): 0 -> a, 1 -> b
__global__ void k_ls_select(const float *__restrict__ a0, const float *__restrict__ a1, const float *__restrict__ b0,
                            const float *__restrict__ b1, float *__restrict__ outu, float *__restrict__ outv, size_t n,
                            const LsState *__restrict__ state) {
    const bool odd = state->iters & 1;
    const float *su = odd ? b0 : a0, *sv = odd ? b1 : a1;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) { outu[i] = su[i]; outv[i] = sv[i]; }
}

extern "C" int ofri_ls_compute(ofri_ctx *ctx, const float *im1, const float *im2, const float *u_in, const float *v_in,
                               float *u_out, float *v_out, float h, int maxit, double tol, int rows, int cols,
                               double *err_host, int *iters_host) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(im1 && im2 && u_in && v_in && u_out && v_out, "null pointer");
    OFRI_REQUIRE(rows > 0 && cols > 0 && maxit >= 0, "bad shape or maxit");
    const size_t n = (size_t)rows * cols, bytes = n * sizeof(float);
    PoolBuf planes(ctx), pp(ctx), st(ctx), parts(ctx);
    OFRI_TRY(planes.get(8 * bytes));
    OFRI_TRY(pp.get(4 * bytes));
    dim3 b(LS_TX, LS_TY / LS_RPT), g(ceil_div(cols, LS_TX), ceil_div(rows, LS_TY));
    const int nblocks = g.x * g.y;
    OFRI_TRY(st.get(sizeof(LsState)));
    OFRI_TRY(parts.get((size_t)nblocks * 4 * sizeof(double)));     // two sets: a sweep writes one while the other is checked
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
    float *a0 = pp.as<float>(), *a1 = a0 + n, *b0 = a0 + 2 * n, *b1 = a0 + 3 * n;
    OFRI_HIP(hipMemcpyAsync(a0, u_in, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    OFRI_HIP(hipMemcpyAsync(a1, v_in, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    LsState *state = st.as<LsState>();
    k_ls_init_state<<<1, 1, 0, ctx->stream>>>(state);
    OFRI_LAUNCH_CHECK();
    const double inv_rc = 1.0 / ((double)rows * (double)cols);
    for (int k = 0; k < maxit; ++k) {
        const float *su = (k & 1) ? b0 : a0, *sv = (k & 1) ? b1 : a1;
        float *du = (k & 1) ? a0 : b0, *dv = (k & 1) ? a1 : b1;
        double *pk = parts.as<double>() + (size_t)(k & 1) * 2 * nblocks, *pprev = parts.as<double>() + (size_t)((k + 1) & 1) * 2 * nblocks;
        k_ls_sweep<<<dim3(g.x, g.y + 1), b, 0, ctx->stream>>>(su, sv, du, dv, f, h, rows, cols, state, pk, pprev, k, nblocks, inv_rc, tol);
        OFRI_LAUNCH_CHECK();
    }
    if (maxit > 0) {        // the check of the last sweep
        const double *pprev = parts.as<double>() + (size_t)((maxit + 1) & 1) * 2 * nblocks;
        k_ls_sweep<<<dim3(1, 1), b, 0, ctx->stream>>>(nullptr, nullptr, nullptr, nullptr, f, h, rows, cols, state, nullptr, pprev, maxit, nblocks, inv_rc, tol);
        OFRI_LAUNCH_CHECK();
    }
    {
        size_t gsz = (n + 255) / 256;
        if (gsz > 4096) gsz = 4096;
        k_ls_select<<<(int)gsz, 256, 0, ctx->stream>>>(a0, a1, b0, b1, u_out, v_out, n, state);
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

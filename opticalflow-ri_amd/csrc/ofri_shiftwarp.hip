// The optical-flow-equation warp of the pyramid driver (reference src/GenericPyramidalOpticalFlow.py:189-221,
// taken when biLinear is False).
//
//   1. every pixel moves to (y + floor(V + 0.5), x + floor(U + 0.5)); where several pixels land on one
//      destination the one latest in row-major order stays (NumPy's fancy-index assignment writes in
//      that order); destinations nothing lands on keep their value; an index in [-n, 0) counts from the
//      far edge, anything further out is the IndexError of the reference;
//   2. the fractional parts dU, dV are smoothed by the reference's own separable Gaussian
//      (src/gaussian_filter.py:87-90; 73 taps for the sigma and truncation the driver passes);
//   3. I -= d(I*dU)/dx + d(I*dV)/dy by forward differences, on all but the last row and column.
#include "ofri_internal.h"

#pragma clang fp contract(off)

// ---- step 1 -------------------------------------------------------------------------------------------
// oob[0], oob[1]: smallest row-major pixel index whose row / column destination is out of range
__global__ void k_shift_vote(const float *__restrict__ U, const float *__restrict__ V, int *__restrict__ win,
                             float *__restrict__ dU, float *__restrict__ dV, int *__restrict__ oob, int H, int W) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= W || y >= H) return;
    const int s = y * W + x;
    const float u = U[s], v = V[s];
    const float fu = floorf(__fadd_rn(u, 0.5f)), fv = floorf(__fadd_rn(v, 0.5f));
    dU[s] = __fsub_rn(u, fu);
    dV[s] = __fsub_rn(v, fv);
    // destinations as the reference forms them: int32(x + floor(...)) evaluated in float64
    const double tx = (double)x + (double)fu, ty = (double)y + (double)fv;
    const bool okx = tx >= -(double)W && tx < (double)W, oky = ty >= -(double)H && ty < (double)H;   // false for NaN
    if (!oky) atomicMin(&oob[0], s);
    if (!okx) atomicMin(&oob[1], s);
    if (!okx || !oky) return;
    int ix = (int)tx, iy = (int)ty;
    if (ix < 0) ix += W;
    if (iy < 0) iy += H;
    atomicMax(&win[iy * W + ix], s);
}
__global__ void k_shift_apply(const float *__restrict__ im, const int *__restrict__ win, float *__restrict__ out, int n) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n) return;
    const int s = win[d];
    out[d] = im[s >= 0 ? s : d];
}
// ---- step 3 -------------------------------------------------------------------------------------------
__global__ void k_ofe(const float *__restrict__ I, const float *__restrict__ dU, const float *__restrict__ dV,
                      float *__restrict__ out, int H, int W) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= W || y >= H) return;
    const size_t s = (size_t)y * W + x;
    float r = I[s];
    if (x < W - 1 && y < H - 1) {
        const float c = I[s];
        const float tx = __fsub_rn(__fmul_rn(I[s + 1], dU[s + 1]), __fmul_rn(c, dU[s]));
        const float ty = __fsub_rn(__fmul_rn(I[s + W], dV[s + W]), __fmul_rn(c, dV[s]));
        r = __fsub_rn(c, __fadd_rn(tx, ty));
    }
    out[s] = r;
}

extern "C" int ofri_warp_shift_ofe(ofri_ctx *ctx, const float *im, const float *U, const float *V, float *out, int rows,
                                   int cols, const float *taps_host, int ntaps) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(im && U && V && out, "null pointer");
    OFRI_REQUIRE(rows > 0 && cols > 0 && (long long)rows * cols < (1LL << 31), "bad shape %d x %d", rows, cols);
    OFRI_REQUIRE(taps_host && ntaps >= 1 && (ntaps & 1), "taps: odd count expected, got %d", ntaps);
    // this call reads the range check back: refuse a capture before anything is enqueued into it
    OFRI_REQUIRE(!ctx->capturing, "ofri_warp_shift_ofe waits for the GPU (its range check), which a captured sequence cannot do");
    const int H = rows, W = cols, n = rows * cols;
    const size_t bytes = (size_t)n * sizeof(float);
    PoolBuf win(ctx), dU(ctx), dV(ctx), shifted(ctx), oob(ctx);
    OFRI_TRY(win.get(bytes)); OFRI_TRY(dU.get(bytes)); OFRI_TRY(dV.get(bytes));
    OFRI_TRY(shifted.get(bytes)); OFRI_TRY(oob.get(2 * sizeof(int)));
    OFRI_TRY(ofri_dev_fill(ctx, win.p, 0xFF, bytes));             // -1: nothing landed here
    OFRI_TRY(ofri_dev_fill(ctx, oob.p, 0x7F, 2 * sizeof(int)));   // 0x7F7F7F7F: no offender
    dim3 b(64, 4), g(ceil_div(W, 64), ceil_div(H, 4));
    k_shift_vote<<<g, b, 0, ctx->stream>>>(U, V, win.as<int>(), dU.as<float>(), dV.as<float>(), oob.as<int>(), H, W);
    OFRI_LAUNCH_CHECK();
    int bad[2];
    OFRI_HIP(hipMemcpyAsync(bad, oob.p, sizeof(bad), hipMemcpyDeviceToHost, ctx->stream));
    OFRI_TRY(ofri_stream_wait(ctx));
    for (int axis = 0; axis < 2; ++axis) {
        if (bad[axis] == 0x7F7F7F7F) continue;
        // NumPy validates the row indices before the column indices and names the first offender
        float f;
        OFRI_HIP(hipMemcpy(&f, (axis == 0 ? V : U) + bad[axis], sizeof(float), hipMemcpyDeviceToHost));
        const double idx = (double)(axis == 0 ? bad[axis] / W : bad[axis] % W) + (double)floorf(f + 0.5f);
        return ofri_set_error(OFRI_ERANGE, "index %.0f is out of bounds for axis %d with size %d", idx, axis, axis == 0 ? H : W);
    }
    k_shift_apply<<<ceil_div(n, 256), 256, 0, ctx->stream>>>(im, win.as<int>(), shifted.as<float>(), n);
    OFRI_LAUNCH_CHECK();
    OFRI_TRY(ofri_gauss_sep(ctx, dU.as<float>(), dU.as<float>(), H, W, taps_host, ntaps));
    OFRI_TRY(ofri_gauss_sep(ctx, dV.as<float>(), dV.as<float>(), H, W, taps_host, ntaps));
    k_ofe<<<g, b, 0, ctx->stream>>>(shifted.as<float>(), dU.as<float>(), dV.as<float>(), out, H, W);
    OFRI_LAUNCH_CHECK();
    return OFRI_OK;
}

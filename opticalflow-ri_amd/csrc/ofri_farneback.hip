// Farneback polynomial-expansion optical flow kernels.
// Reference behaviour: src/optical_flow_farneback.cl:72-429 (OpenCV-derived OpenCL kernels) and their
// host wrappers src/Farneback_PyCL.py:211-457.  The kernels (ofri_farneback_kernels.h) follow the kernel text operation
// by operation in float32 and are compiled twice: in namespace fb_exact with separately rounded products and sums (no
// FMA contraction: the bits of the CPU restatement of the OpenCL text), in namespace fb_fast with a * b + c contracted
// into fused multiply-adds, which an OpenCL compiler is free to do with the reference's own source as well
// (optical_flow_farneback.cl:222-227, 244-247) and which takes a third of the operations out of the window filter.
// ofri_fb_set_mode picks one per context (fast by default).
#include "ofri_internal.h"

#define FB_MAX_KSH 100                 // MAX_KSIZE_HALF of the reference
struct FbTaps { float k[FB_MAX_KSH + 1]; };
struct FbPoly { float g[8], xg[8], xxg[8], ig[4]; };

// border index helpers of the reference (.cl:135-158): reflect-101 with a modulo guard
__device__ __forceinline__ int fb_low(int y, int last) { return abs(y) % (last + 1); }
__device__ __forceinline__ int fb_high(int y, int last) { return abs(last - abs(last - y)) % (last + 1); }
// in-range indices map to themselves under both helpers; only out-of-range ones pay for the modulo
__device__ __forceinline__ int fb_idx(int t, int last) { return (unsigned)t <= (unsigned)last ? t : fb_low(fb_high(t, last), last); }


// ---- updateMatrices border attenuation (.cl:254) ------------------------------------------------------
__constant__ float c_fb_border[6] = {0.14f, 0.14f, 0.4472f, 0.4472f, 0.4472f, 1.f};

// FB_PIXEL_CONTRACT: the contraction rule inside the per-pixel matrices step, which is instantiated for one pixel and
// for several in flight and must give the same bits in both: multiply-adds are fused only inside one expression there
// (the compiler's choice across statements depends on the surrounding code).
namespace fb_exact {
#pragma clang fp contract(off)
#define FB_PIXEL_CONTRACT _Pragma("clang fp contract(off)")
#include "ofri_farneback_kernels.h"
#undef FB_PIXEL_CONTRACT
}  // namespace fb_exact
namespace fb_fast {
#pragma clang fp contract(fast)
#define FB_PIXEL_CONTRACT _Pragma("clang fp contract(on)")
#include "ofri_farneback_kernels.h"
#undef FB_PIXEL_CONTRACT
}  // namespace fb_fast
#pragma clang fp contract(off)

extern "C" int ofri_fb_set_mode(ofri_ctx *ctx, int mode) {
    OFRI_REQUIRE(ctx, "null context");
    OFRI_REQUIRE(mode == 0 || mode == -1, "mode: 0 fast (fused multiply-adds), -1 exact (separately rounded operations)");
    ctx->fb_mode = mode;
    return OFRI_OK;
}

#define FB_DISPATCH(ctx, call) ((ctx) && (ctx)->fb_mode == -1 ? fb_exact::call : fb_fast::call)

extern "C" int ofri_fb_polyexp(ofri_ctx *ctx, const float *src, float *dst5, int rows, int cols, int polyN,
                               const float *g_host, const float *xg_host, const float *xxg_host, const float *ig_host) {
    return FB_DISPATCH(ctx, fbk_polyexp(ctx, src, dst5, rows, cols, polyN, g_host, xg_host, xxg_host, ig_host));
}
extern "C" int ofri_fb_blur_polyexp(ofri_ctx *ctx, const float *src, float *dst5, int rows, int cols,
                                    const float *blur_half_taps_host, int ksize_half, int polyN, const float *g_host,
                                    const float *xg_host, const float *xxg_host, const float *ig_host) {
    return FB_DISPATCH(ctx, fbk_blur_polyexp(ctx, src, dst5, rows, cols, blur_half_taps_host, ksize_half, polyN, g_host, xg_host, xxg_host, ig_host));
}
extern "C" int ofri_fb_blur(ofri_ctx *ctx, const float *src, float *dst, int rows, int cols,
                            const float *half_taps_host, int ksize_half) {
    return FB_DISPATCH(ctx, fbk_blur(ctx, src, dst, rows, cols, half_taps_host, ksize_half));
}
extern "C" int ofri_fb_blur5(ofri_ctx *ctx, const float *src5, float *dst5, int rows, int cols,
                             const float *half_taps_host, int ksize_half) {
    return FB_DISPATCH(ctx, fbk_blur5(ctx, src5, dst5, rows, cols, half_taps_host, ksize_half));
}
extern "C" int ofri_fb_box5(ofri_ctx *ctx, const float *src5, float *dst5, int rows, int cols, int ksize_half) {
    return FB_DISPATCH(ctx, fbk_box5(ctx, src5, dst5, rows, cols, ksize_half));
}
extern "C" int ofri_fb_update_matrices(ofri_ctx *ctx, const float *flowx, const float *flowy, const float *R0,
                                       const float *R1, float *M, int rows, int cols) {
    return FB_DISPATCH(ctx, fbk_update_matrices(ctx, flowx, flowy, R0, R1, M, rows, cols));
}
extern "C" int ofri_fb_update_flow(ofri_ctx *ctx, const float *M, float *flowx, float *flowy, int rows, int cols) {
    return FB_DISPATCH(ctx, fbk_update_flow(ctx, M, flowx, flowy, rows, cols));
}
extern "C" int ofri_fb_window5_update_flow(ofri_ctx *ctx, const float *M5, float *flowx, float *flowy, int rows, int cols,
                                           const float *half_taps_host, int ksize_half, int box) {
    return FB_DISPATCH(ctx, fbk_window5_update_flow(ctx, M5, flowx, flowy, rows, cols, half_taps_host, ksize_half, box));
}
extern "C" int ofri_fb_window5_flow_matrices(ofri_ctx *ctx, const float *M5, float *flowx, float *flowy, const float *R0,
                                             const float *R1, float *M5_next, int rows, int cols,
                                             const float *half_taps_host, int ksize_half, int box) {
    return FB_DISPATCH(ctx, fbk_window5_flow_matrices(ctx, M5, flowx, flowy, R0, R1, M5_next, rows, cols, half_taps_host, ksize_half, box));
}

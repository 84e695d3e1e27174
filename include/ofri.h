/*
 * ofri.h - C ABI of libofri_hip.so, the MI355X (gfx950) dense optical-flow engine.
 *
 * The reference (alexlib/OpticalFlow-RI) has no native/FFI boundary: its seam is the Python adapter
 * protocol  U, V, err = adapter.compute(im1, im2, U, V)  (src/GenericPyramidalOpticalFlow.py:256-290,
 * called at :406-411).  This header is what a ctypes binding underneath that protocol calls; every
 * entry point names the reference function it replaces.
 *
 * Conventions
 *  - every function returns 0 on success, a negative OFRI_E* code on failure; ofri_last_error()
 *    returns the calling thread's last message.  Nothing throws across this boundary.
 *  - images and flow fields are float32, row-major, dense (pitch == cols), shape (rows, cols);
 *    U is the x / column component, V the y / row component, as in the reference.
 *  - pointers named *_host are host memory; every other image pointer is DEVICE memory on the
 *    context's GPU (from ofri_alloc, hipMalloc, or a torch tensor's data_ptr()).
 *  - work is enqueued on the context's HIP stream and is asynchronous unless the function hands a
 *    value back to the host (err_host, iters_host, ofri_download, ofri_ctx_sync).
 *  - one host thread per context at a time; contexts are independent (one per GPU / per process).
 */
#ifndef OFRI_H
#define OFRI_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OFRI_OK          0
#define OFRI_EINVAL     -1   /* bad argument (null pointer, non-positive size, unsupported option) */
#define OFRI_EHIP       -2   /* a HIP runtime call failed; message carries hipGetErrorString */
#define OFRI_ENOMEM     -3   /* device allocation failed */
#define OFRI_ENODEVICE  -4   /* no usable GPU */
#define OFRI_ERANGE     -5   /* a data-dependent index left the array (the reference's IndexError) */

typedef struct ofri_ctx ofri_ctx;
typedef struct ofri_event ofri_event;

/* ---- library / context --------------------------------------------------------------------- */
int         ofri_version(void);
const char *ofri_last_error(void);
int         ofri_device_count(int *count);
/* stream == NULL: the context creates and owns a non-blocking stream; otherwise it borrows the
 * caller's hipStream_t (e.g. torch.cuda.current_stream().cuda_stream).  Replaces the
 * cl.Context / cl.CommandQueue creation of Farneback_PyCL.py:98-99,230 and
 * denseLucasKanade_PyCL.py:40-41 (platformID/deviceID -> HIP device ordinal). */
int ofri_ctx_create(int device, void *stream, ofri_ctx **out);
int ofri_ctx_destroy(ofri_ctx *ctx);
int ofri_ctx_sync(ofri_ctx *ctx);
int ofri_ctx_stream(ofri_ctx *ctx, void **stream);

/* pooled device memory: ofri_free returns the block to the context's pool, so steady-state
 * pipelines make no hipMalloc calls.  ofri_pool_trim releases the pool to the driver. */
int ofri_alloc(ofri_ctx *ctx, size_t bytes, void **dptr);
int ofri_free(ofri_ctx *ctx, void *dptr);
int ofri_pool_trim(ofri_ctx *ctx);
/* Allocation counters, for callers and tests that must show a code path allocates nothing: *runtime_allocs = allocating
 * HIP runtime calls this library has made in this process so far (hipMalloc, hipHostMalloc, event creation; all
 * contexts), *pool_requests = ofri_alloc calls on this context so far (served from the pool or not).  Either pointer may
 * be null.  No counterpart in the reference (PyOpenCL allocates a cl.Buffer per kernel call, Farneback_PyCL.py:225-229). */
int ofri_alloc_stats(ofri_ctx *ctx, unsigned long long *runtime_allocs, unsigned long long *pool_requests);
/* Debugging aid, no counterpart in the reference.  With OFRI_POOL_GUARD=<KiB> in the environment when the library is
 * loaded, every pool block carries that many KiB of 0xC3 bytes (0xFF with OFRI_POOL_GUARD_NAN=1: float NaNs, so that
 * reads outside an array reach the results) before and after it; this call waits for the device and counts the blocks,
 * of all contexts, whose bands were written to (*bad_blocks; the first one's size and the offending byte's offset from
 * the block's start in the optional outputs).  OFRI_EINVAL when the bands are off.  OFRI_POOL_POISON=1 fills every block
 * with 0xFF bytes when it is handed out, so that a kernel reading memory nobody wrote in that call shows up as NaNs. */
int ofri_pool_guard_check(ofri_ctx *ctx, int *bad_blocks, unsigned long long *first_bad_size, long long *first_bad_offset);
int ofri_upload(ofri_ctx *ctx, void *dst, const void *src_host, size_t bytes);     /* blocking */
int ofri_download(ofri_ctx *ctx, void *dst_host, const void *src, size_t bytes);   /* blocking */
int ofri_copy(ofri_ctx *ctx, void *dst, const void *src, size_t bytes);            /* device->device */
int ofri_zero(ofri_ctx *ctx, void *dst, size_t bytes);
/* uint8 / uint16 grey image -> float32 on device (examples do imread(...).astype(np.float32),
 * examples/PyHSchunck_Fs3_4.py:129-130) */
int ofri_u8_to_f32(ofri_ctx *ctx, const unsigned char *src, float *dst, size_t n);
int ofri_u16_to_f32(ofri_ctx *ctx, const unsigned short *src, float *dst, size_t n);

/* The 16-bit -> 8-bit range rule of benchmark_of_methods.py:134-137: if max(frame0) > 255 both frames
 * become float32(x / 65535 * 255); decided on the GPU, nothing blocks. */
int ofri_range_rule_16bit(ofri_ctx *ctx, float *frame0, float *frame1, size_t n);
/* pinned host memory and stream-ordered copies (no reference counterpart: the reference uploads every
 * operand synchronously per kernel call, Farneback_PyCL.py:225-235); ofri_ctx_wait orders one
 * context's stream after another's, so a copy context can feed a compute context */
int ofri_host_alloc(size_t bytes, void **hptr);
int ofri_host_free(void *hptr);
int ofri_upload_async(ofri_ctx *ctx, void *dst, const void *src_pinned_host, size_t bytes);
int ofri_download_async(ofri_ctx *ctx, void *dst_pinned_host, const void *src, size_t bytes);
int ofri_ctx_wait(ofri_ctx *ctx, ofri_ctx *other);

/* A sequence of calls captured once and replayed with one launch (a HIP graph on the context's stream).  No
 * reference counterpart: the reference pays a Python call and an OpenCL/Numba dispatch per step; below about
 * 2048 x 2048 pixels that, not the GPU, sets the frame rate.  Everything enqueued on the context between
 * ofri_graph_begin and ofri_graph_end is recorded instead of run; calls that wait for the GPU (err_host, iters_host,
 * ofri_download, ofri_ctx_sync, ofri_warp_shift_ofe) or that upload per-call host tables (Farneback, dense LK)
 * cannot be captured and make ofri_graph_end fail.  Run the same sequence once before capturing, so that its tables
 * exist and the pool holds its blocks.  Device memory obtained from ofri_alloc during the capture stays valid for the
 * graph's lifetime: blocks freed inside the capture become the graph's scratch memory and are handed to nobody else;
 * blocks still held at ofri_graph_end (the results) are the caller's to keep until the graph is destroyed.
 * ofri_graph_launch replays on the same context; inputs are whatever the captured pointers hold at that time. */
typedef struct ofri_graph ofri_graph;
int ofri_graph_begin(ofri_ctx *ctx);
int ofri_graph_end(ofri_ctx *ctx, ofri_graph **graph);
int ofri_graph_launch(ofri_ctx *ctx, ofri_graph *graph);
int ofri_graph_destroy(ofri_graph *graph);       /* before ofri_ctx_destroy */
/* A capture that runs with ofri_prof_enable(ctx, 1) records the timed sections (the Horn-Schunck sweep launches, see
 * ofri_prof_read) into the graph as event nodes: every replay times them again.  ofri_graph_prof_read waits for the
 * section's end in the graph's last replay and returns what ofri_prof_read returns.  No reference counterpart. */
int ofri_graph_prof_count(ofri_graph *graph, int *count);
int ofri_graph_prof_read(ofri_graph *graph, int idx, float *ms, int *launches, int *sweeps, int *rows, int *cols);

/* HIP events on the context's stream, for timing kernels from a host language */
int ofri_event_create(ofri_ctx *ctx, ofri_event **ev);
int ofri_event_record(ofri_ctx *ctx, ofri_event *ev);
int ofri_event_elapsed_ms(ofri_event *start, ofri_event *stop, float *ms);          /* syncs on stop */
int ofri_event_sync(ofri_event *ev);                                                /* host waits */
int ofri_event_destroy(ofri_event *ev);

/* ---- Gaussian pre-filter -------------------------------------------------------------------- */
/* convolveSeparableFilter (src/gaussian_filter.py:54-85): rows pass then columns pass, float32
 * sequential accumulation; border as the reference pads it (:63-66): mirrored with the edge sample
 * repeated on the low side, the last K/2 samples repeated in their original order on the high side.  dst may equal src (the
 * reference filters in place).  taps_host: ntaps float32 values (ntaps odd, <= 127,
 * ntaps/2 <= min(rows, cols)). */
int ofri_gauss_sep(ofri_ctx *ctx, const float *src, float *dst, int rows, int cols,
                   const float *taps_host, int ntaps);
/* prepareGaussianKernel (src/gaussian_filter.py:47-52) into taps_host[ksize] (host-side, no GPU) */
int ofri_gauss_taps(double sigma, int ksize, float *taps_host);

/* ---- Horn-Schunck --------------------------------------------------------------------------- */
/* computeDerivatives(im1=first, im2=second) (src/HornSchunck.py:107-127), 'mirror' border.
 * The adapter chain passes frame 1 as `first` and frame 0 as `second` (HornSchunck.py:37,73,84). */
int ofri_hs_derivs(ofri_ctx *ctx, const float *first, const float *second,
                   float *fx, float *fy, float *ft, int rows, int cols);
/* HS_helper + HS_helper2 (src/HornSchunck.py:52-71): niter Jacobi sweeps, result left in U, V.
 * Several sweeps are fused per launch (temporal blocking).  ofri_hs_set_fuse(ctx, t) selects how:
 *   t = 0   automatic: row-streaming pipeline (12 sweeps per launch) on fields of 2048 x 2048 pixels and
 *           more, tile kernel (up to 8 per launch) for the rest and for smaller fields;
 *   t = 1   one launch per sweep;  t = 2..8  tile kernel with at most t sweeps per launch;
 *   t = 16  row-streaming pipeline (strips of 128 columns) for every sweep, any field size;
 *   t = 32  the same with strips of 256 columns (what t = 0 picks where the row chunks stay long: 4096 x 4096);
 *   t = -1  exact mode: one launch per sweep with scipy's float64 tap accumulation and NumPy's unfused
 *           float32 element operations, bit-identical to the reference.
 * All settings but -1 give the same bits as each other. */
int ofri_hs_iterate(ofri_ctx *ctx, float *U, float *V, const float *fx, const float *fy,
                    const float *ft, float alpha, int niter, int rows, int cols);
int ofri_hs_set_fuse(ofri_ctx *ctx, int sweeps_per_launch);
/* No reference counterpart: a hint for callers that keep several frame pairs in flight on one GPU, each on a context
 * of its own (piv_batch.flow_lanes).  The row-streaming pipelines (Horn-Schunck, Liu-Shen) cut a field into as many row
 * chunks as fill the GPU once; every chunk recomputes the rows above it that its sweeps depend on (30 rows for 12
 * Horn-Schunck sweeps), which is the price of a short latency for one pair.  With this hint a launch takes 1/contexts
 * of those chunks and leaves the rest of the GPU to the other contexts' launches: less is recomputed and more pairs
 * finish per second (three contexts at 2048 x 2048: best with 2, worse than the default from 4 on).  Same bits.
 * Default 1. */
int ofri_set_share(ofri_ctx *ctx, int contexts);
/* Timing of the sweep launches, for roofline reports: while enabled, every ofri_hs_iterate call
 * brackets its sweep launches with two HIP events on the context's stream and appends a record.
 * ofri_prof_read syncs on record idx and returns its elapsed time, the number of kernel launches and
 * of Jacobi sweeps they covered, and the field shape.  ofri_prof_enable(ctx, 0/1) also clears records. */
int ofri_prof_enable(ofri_ctx *ctx, int on);
int ofri_prof_count(ofri_ctx *ctx, int *count);
int ofri_prof_read(ofri_ctx *ctx, int idx, float *ms, int *launches, int *sweeps, int *rows, int *cols);
/* HS as reached through HSOpticalFlowAlgoAdapter.compute (src/HornSchunck.py:35-37,73-105):
 * derivatives, niter sweeps from (U_in, V_in), error (|Un-U_in|_F + |Vn-V_in|_F)/(rows*cols).
 * U_out/V_out must not alias U_in/V_in.  U_in and V_in may both be NULL for a zero initial flow (what the
 * driver passes after a warp, src/GenericPyramidalOpticalFlow.py:224-225,361-362).  err_host may be NULL (then
 * nothing blocks). */
int ofri_hs_compute(ofri_ctx *ctx, const float *frame0, const float *frame1,
                    const float *U_in, const float *V_in, float *U_out, float *V_out,
                    float alpha, int niter, int rows, int cols, double *err_host);
/* ofri_hs_compute without any allocation.  No counterpart in the reference: it is for callers that cannot call
 * hipMalloc (code under HIP graph capture, tight serving loops, harnesses that forbid it).  Same arguments, same
 * kernels and same bits as ofri_hs_compute on the one-sweep and tile kernels; the scratch planes (two flow planes,
 * three coefficient planes, padded copies for widths that are not a multiple of 4, a zero start, norm partials) live in
 * a file-scope device array that every GPU holds until the process ends: 9 * OFRI_HS_STATIC_MAX_PIXELS floats + 16 KiB =
 * 36 MiB per GPU.  rows * (cols rounded up to a multiple of 4) must not exceed OFRI_HS_STATIC_MAX_PIXELS, else
 * OFRI_EINVAL.  Makes no hipMalloc / hipFree / pool call and creates no events (the ofri_prof_* records skip it).  All
 * calls on one GPU share that GPU's array: calls that overlap on one GPU (two contexts, two threads) race; calls on
 * one context are ordered by its stream. */
#define OFRI_HS_STATIC_MAX_PIXELS (1 << 20)
int ofri_hs_compute_static(ofri_ctx *ctx, const float *frame0, const float *frame1,
                           const float *U_in, const float *V_in, float *U_out, float *V_out,
                           float alpha, int niter, int rows, int cols, double *err_host);
/* The same step as the driver uses it when no second algorithm refines the level: the new flow is an increment
 * that the driver adds to the flow accumulated so far (Uaccum += U, src/GenericPyramidalOpticalFlow.py:413-414).
 * U_acc, V_acc are updated in place with the same float32 additions; on large fields the last sweep launch does
 * them as it stores, which saves a pass over four planes. */
int ofri_hs_compute_acc(ofri_ctx *ctx, const float *frame0, const float *frame1, const float *U_in,
                        const float *V_in, float *U_acc, float *V_acc, float alpha, int niter, int rows, int cols);
/* One pyramid level of the driver with Horn-Schunck as its only algorithm (src/GenericPyramidalOpticalFlow.py:
 * 200-201 warp, 374-375 pre-filter, 406 adapter step from a zero flow, 413-414 accumulation), on the level's
 * images as resized, before warp and filter.
 *   warp = 0 (first level): U_flow, V_flow receive the flow of the filtered pair.
 *   warp = 1: both images are warped by half of (U_flow, V_flow) in opposite directions, as ofri_warp_pair does,
 *             filtered, and the new flow is added to U_flow, V_flow.
 * taps_host: ntaps float32 filter taps as for ofri_gauss_sep (ntaps = 0: no filter).  Same bits as
 * ofri_warp_pair, ofri_gauss_sep twice and ofri_hs_compute(_acc) called one after the other; on large fields with
 * 3, 5 or 7 taps the warped and filtered images are never written to memory. */
int ofri_hs_level(ofri_ctx *ctx, const float *im1, const float *im2, float *U_flow, float *V_flow,
                  const float *taps_host, int ntaps, int warp, float alpha, int niter, int rows, int cols);

/* ---- Liu-Shen physics-based refinement --------------------------------------------------------- */
/* physicsBasedOpticalFlowLiuShen(im1, im2, h, U, V) (src/PhysicsBasedOpticalFlowLiuShen.py:82-158)
 * with its own argument convention: u is the ROW-direction component, v the column-direction one
 * (LiuShenOpticalFlowAlgoAdapter.compute swaps U/V in and out, :37-39).  Images are divided by their
 * maxima, the coefficient fields and the inverse matrix are built, then up to maxit fixed-point
 * sweeps run until total_error <= tol (reference: maxit 60, tol 1e-8).  The stopping rule is
 * evaluated on the GPU; nothing blocks unless err_host / iters_host is non-NULL. */
int ofri_ls_compute(ofri_ctx *ctx, const float *im1, const float *im2, const float *u_in,
                    const float *v_in, float *u_out, float *v_out, float h, int maxit, double tol,
                    int rows, int cols, double *err_host, int *iters_host);
/* Arithmetic and launch shape of the Liu-Shen sweeps (:141-156), like ofri_hs_set_fuse for Horn-Schunck:
 *   mode = 0   automatic: float32 arithmetic with fused multiply-adds in one pinned order; the row-streaming pipeline
 *              (6 sweeps per launch) on fields of 512 x 512 pixels and more whose width is a multiple of 4, one sweep
 *              per launch otherwise;
 *   mode = 1   the same arithmetic, one launch per sweep;   mode = 6 / 8   the pipeline at that depth, any field size;
 *   mode = -1  exact mode: scipy's float64 stencil sums and NumPy's separately rounded float32 products, one launch
 *              per sweep, the reference's bits.
 * All settings but -1 give the same flow bits as each other (about 1e-7 relative from the exact mode per call); the
 * stopping rule is the reference's in every mode (total_error as float32, at most maxit sweeps). */
int ofri_ls_set_mode(ofri_ctx *ctx, int mode);

/* ---- Farneback polynomial expansion --------------------------------------------------------------- */
/* Arithmetic of every ofri_fb_* kernel on this context:
 *   mode = 0   fast (default): float32 in the kernel text's order with a * b + c contracted into fused multiply-adds, as an
 *              OpenCL compiler may do with the reference's source (optical_flow_farneback.cl:222-227, 244-247);
 *   mode = -1  exact: products and sums rounded separately, the bits of the CPU restatement of the OpenCL text.
 * The two differ by about 1e-7 relative per kernel. */
int ofri_fb_set_mode(ofri_ctx *ctx, int mode);
/* 5-plane arrays (R0, R1, M) hold their planes stacked vertically: plane k is rows k*rows..(k+1)*rows-1
 * of a (5*rows, cols) float32 array, as in the reference (src/Farneback_PyCL.py:542-545).
 * polynomialExpansion (src/optical_flow_farneback.cl:72-133; host src/Farneback_PyCL.py:311-356):
 * g/xg/xxg_host are the non-negative halves (polyN+1 floats), ig_host = {ig11, ig03, ig33, ig55}. */
int ofri_fb_polyexp(ofri_ctx *ctx, const float *src, float *dst5, int rows, int cols, int polyN,
                    const float *g_host, const float *xg_host, const float *xxg_host, const float *ig_host);
/* gaussianBlur followed by polynomialExpansion of the blurred frame, the way Farneback_PyCL.compute prepares
 * each frame of a level (src/Farneback_PyCL.py:563-585) when the level has the frame's own size: one launch for
 * the 3-tap pre-blur (ksize_half 1), the two steps otherwise; same bits as ofri_fb_blur + ofri_fb_polyexp. */
int ofri_fb_blur_polyexp(ofri_ctx *ctx, const float *src, float *dst5, int rows, int cols,
                         const float *blur_half_taps_host, int ksize_half, int polyN, const float *g_host,
                         const float *xg_host, const float *xxg_host, const float *ig_host);
/* gaussianBlur / gaussianBlur5 (.cl:160-253; host :211-272): separable, reflect-101 borders,
 * half_taps_host = centre tap first, ksize_half+1 floats (ksize_half <= 100). */
int ofri_fb_blur(ofri_ctx *ctx, const float *src, float *dst, int rows, int cols,
                 const float *half_taps_host, int ksize_half);
int ofri_fb_blur5(ofri_ctx *ctx, const float *src5, float *dst5, int rows, int cols,
                  const float *half_taps_host, int ksize_half);
/* boxFilter5 (.cl:350-406; host :274-309): clamped borders, mean over (2*ksize_half+1)^2 */
int ofri_fb_box5(ofri_ctx *ctx, const float *src5, float *dst5, int rows, int cols, int ksize_half);
/* updateMatrices (.cl:256-348; host :391-423) */
int ofri_fb_update_matrices(ofri_ctx *ctx, const float *flowx, const float *flowy, const float *R0,
                            const float *R1, float *M, int rows, int cols);
/* updateFlow (.cl:408-429; host :358-389) */
int ofri_fb_update_flow(ofri_ctx *ctx, const float *M, float *flowx, float *flowy, int rows, int cols);
/* gaussianBlur5 (box != 0: boxFilter5) followed by updateFlow in one pass, the pairing used by
 * updateFlowGaussianBlur / updateFlowBoxFilter (src/Farneback_PyCL.py:425-457): the filtered
 * matrices are consumed in registers and never written. */
int ofri_fb_window5_update_flow(ofri_ctx *ctx, const float *M5, float *flowx, float *flowy, int rows,
                                int cols, const float *half_taps_host, int ksize_half, int box);
/* The same followed by updateMatrices for the next iteration, the body of the reference's flow loop when it is
 * not the last iteration (src/Farneback_PyCL.py:442-457: gaussianBlur5 / boxFilter5, updateFlow, updateMatrices):
 * one launch for ksize_half 16, the separate steps otherwise.  The new matrices go to M5_next (not M5); flowx and
 * flowy may both be NULL when the displacement of this iteration is not wanted. */
int ofri_fb_window5_flow_matrices(ofri_ctx *ctx, const float *M5, float *flowx, float *flowy, const float *R0,
                                  const float *R1, float *M5_next, int rows, int cols,
                                  const float *half_taps_host, int ksize_half, int box);

/* ---- dense Lucas-Kanade ----------------------------------------------------------------------------- */
/* lkDense (src/pyrlkDenseLargeW.cl:304-669) as launched by denseLucasKanade_PyCl.compute
 * (src/denseLucasKanade_PyCL.py:113-169): (2*half_window+1)^2 window (at most 32x32), Scharr
 * gradients of im1, up to niter Newton steps per pixel with im2 sampled bilinearly (clamp to edge),
 * step tolerance 0.01 px.  asym_* are the vorticity-enhancement window flags (:75-92; 0 when off).
 * Pixels whose window has a singular structure tensor keep their input displacement. */
int ofri_lk_compute(ofri_ctx *ctx, const float *im1, const float *im2, const float *U_in,
                    const float *V_in, float *U_out, float *V_out, int rows, int cols, int niter,
                    int half_window, int asym_left, int asym_right, int asym_top, int asym_bottom);
/*   mode = 0   fast (default): for the 27 x 27 window of the reference's examples (half_window 13, no asymmetry flags) a
 *              tile of 32 x 16 pixels per workgroup shares the staged images and gradients, a lane sums its own pixels'
 *              windows serially and interpolates the second image separably; pixels whose displaced window leaves the
 *              staged neighbourhood (more than 6 pixels from the tile's central displacement) are redone by the exact
 *              kernel.  Agrees with the exact mode to about 1e-5 px.  Other windows use the exact kernel.
 *   mode = -1  exact: one wavefront per pixel in the reference's lane layout and reduction order (.cl:304-669). */
int ofri_lk_set_mode(ofri_ctx *ctx, int mode);

/* ---- pyramid glue --------------------------------------------------------------------------- */
/* imresize (src/GenericPyramidalOpticalFlow.py:67-68): Pillow's scale-aware bicubic (a = -0.5) on
 * float32: horizontal then vertical pass, float64 accumulation, float32 intermediate. */
int ofri_resize_bicubic(ofri_ctx *ctx, const float *src, int rows_in, int cols_in,
                        float *dst, int rows_out, int cols_out);
/* Both frames of a pyramid level in one launch (src/GenericPyramidalOpticalFlow.py:336-343 resizes im1 and im2 one after the
 * other): the same numbers as two calls of ofri_resize_bicubic. */
int ofri_resize_bicubic_pair(ofri_ctx *ctx, const float *src1, const float *src2, int rows_in, int cols_in,
                             float *dst1, float *dst2, int rows_out, int cols_out);
/* Pillow BILINEAR resize used by Farneback_PyCL.imresize (src/Farneback_PyCL.py:62-63) */
int ofri_resize_bilinear(ofri_ctx *ctx, const float *src, int rows_in, int cols_in,
                         float *dst, int rows_out, int cols_out);
/* RectBivariateSpline flow upsampling (src/GenericPyramidalOpticalFlow.py:155-162): interpolating
 * bicubic (not-a-knot) spline with knots k/n evaluated at i/N, clamped at the last knot, float64,
 * rounded to float32 and then multiplied by post_scale in float32 (:169-172; pass 1.0f for none). */
int ofri_spline_upsample(ofri_ctx *ctx, const float *src, int rows_in, int cols_in,
                         float *dst, int rows_out, int cols_out, float post_scale);
/* The two components of a flow in one pair of launches (src/GenericPyramidalOpticalFlow.py:159-162,169-172 upsamples and
 * scales Uaccum and Vaccum one after the other): the same numbers as two calls of ofri_spline_upsample. */
int ofri_spline_upsample_pair(ofri_ctx *ctx, const float *src_u, const float *src_v, int rows_in, int cols_in,
                              float *dst_u, float *dst_v, int rows_out, int cols_out, float scale_u, float scale_v);
/* doBiLinearWarping (src/GenericPyramidalOpticalFlow.py:70-116): out = img sampled at (cy, cx),
 * clamp-to-edge bilinear, float64 weights. */
int ofri_warp_coords(ofri_ctx *ctx, const float *img, const float *cy, const float *cx,
                     float *out, int rows, int cols);
/* the symmetric warp of updateNextPyramidalLevel (:198-201): out = img sampled at
 * (y + half_sign*V/2, x + half_sign*U/2); half_sign = -1 for image 1, +1 for image 2. */
int ofri_warp_flow(ofri_ctx *ctx, const float *img, const float *U, const float *V,
                   float half_sign, float *out, int rows, int cols);
/* both warps of updateNextPyramidalLevel (:200-201) in one pass: out1 = im1 at (y - V/2, x - U/2),
 * out2 = im2 at (y + V/2, x + U/2). */
int ofri_warp_pair(ofri_ctx *ctx, const float *im1, const float *im2, const float *U, const float *V,
                   float *out1, float *out2, int rows, int cols);
/* the warp of updateNextPyramidalLevel with biLinear=False (src/GenericPyramidalOpticalFlow.py:189-221): pixels of
 * im move by (floor(V+0.5), floor(U+0.5)) (last writer in row-major order wins, negative destinations count from
 * the far edge), then out = shifted - (d(shifted*dU)/dx + d(shifted*dV)/dy) with the fractional parts dU, dV
 * smoothed by ofri_gauss_sep(taps) (the driver passes sigma 1.8, 73 taps).  A destination outside [-n, n) returns OFRI_ERANGE with the reference's
 * IndexError text in ofri_last_error().  out may equal im (the reference overwrites its argument). */
int ofri_warp_shift_ofe(ofri_ctx *ctx, const float *im, const float *U, const float *V, float *out,
                        int rows, int cols, const float *taps_host, int ntaps);
/* acc += x  (Uaccum += U, src/GenericPyramidalOpticalFlow.py:413-414);  y *= s */
int ofri_add_inplace(ofri_ctx *ctx, float *acc, const float *x, size_t n);
int ofri_scale_inplace(ofri_ctx *ctx, float *y, float s, size_t n);
/* (|a-b|_F) as float64 to the host (b may be NULL: |a|_F); max over a float32 array */
int ofri_norm_diff(ofri_ctx *ctx, const float *a, const float *b, size_t n, double *out_host);
int ofri_max(ofri_ctx *ctx, const float *a, size_t n, float *out_host);

#ifdef __cplusplus
}
#endif
#endif /* OFRI_H */

// Horn-Schunck sweeps as a row-streaming pipeline: 16 Jacobi sweeps per launch for large fields.
//
// A workgroup owns a vertical strip 128 pixels wide (96 stored, halo 16 each side) of a chunk of rows and
// walks down it one row per step.  Sixteen half-wavefronts are the sixteen sweep levels: at a step, level
// k turns rows r-1, r, r+1 of level k-1 (row r = newest row - 2 - 2k) into row r of sweep k+1, keeping
// the two older rows in registers and taking the new one, plus the row's fx, fy, ft and 1/den, from small
// LDS rings; a ninth wavefront streams the next rows from HBM into the rings, eight rows ahead.  One
// barrier per step.  Each input pixel is read once per 16 sweeps (times the strip overlap 128/96), which
// is what the tile kernel cannot reach: its halo grows in both directions.
//
// The planes are copies padded by 16 mirrored pixels ('mirror' border of the reference: a field extended
// by reflection stays the reflection of itself under the symmetric stencil), so the pipeline has no border
// cases and every 16-byte access is aligned.  Arithmetic is the shared sequence of ofri_hs_math.h, so the
// result is bit-identical to the other Horn-Schunck kernels.
#include "ofri_internal.h"
#include "ofri_hs_math.h"

#define SL_NL 16                         // levels = sweeps per launch = padding
#define SL_RX 128                        // strip width (32 lanes x 4 pixels)
#define SL_OX (SL_RX - 2 * SL_NL)        // stored width
#define SL_PF 8                          // rows the loader keeps in flight
#define SL_CR 34                         // coefficient ring rows: rows rho-32 .. rho are live
#define SL_THREADS (SL_NL * 32 + 64)

__device__ __forceinline__ float sl_left(float v) {     // lane-1 (wave_shr:1)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xF, 0xF, false));
}
__device__ __forceinline__ float sl_right(float v) {    // lane+1 (wave_shl:1)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xF, 0xF, false));
}
__device__ __forceinline__ int sl_refl(int i, int n) {
    if (n == 1) return 0;
    int p = 2 * n - 2;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - i;
}

struct SlPlanes { const float *U, *V, *fx, *fy, *ft, *rd; };   // padded planes, pointers at image pixel (0,0)

__device__ __forceinline__ void sl_step(bool loader, int t, int rho0, int H, int P, int x0, int lcol, int lset, int lj,
                                        const float *pa, const float *pb, const float *pc, float4 &qa, float4 &qb, float4 &qc,
                                        float4 (&ring)[SL_NL][4][2][32], float4 (&coef)[SL_CR][4][32], int &lslot, int level,
                                        int l, int &r, int &cslot, float (&w0u)[4], float (&w0v)[4], float (&w1u)[4],
                                        float (&w1v)[4], int T, int ya, int yb, bool store_lane, float *__restrict__ Uo,
                                        float *__restrict__ Vo, long long opitch) {
        if (loader) {
            const int rho = rho0 + t;
            ring[0][rho & 3][lset][lj & 31] = qa;
            coef[lslot][lset][lj & 31] = qb;
            coef[lslot][2 + lset][lj & 31] = qc;
            lslot = lslot + 1 == SL_CR ? 0 : lslot + 1;
            int row = min(rho + SL_PF, H + SL_NL - 1);
            long long o = (long long)row * P + x0 + lcol;
            qa = *reinterpret_cast<const float4 *>(pa + o);
            qb = *reinterpret_cast<const float4 *>(pb + o);
            qc = *reinterpret_cast<const float4 *>(pc + o);
        } else {
            // newest source row r+1, written one step ago
            const float4 nu4 = ring[level][(r + 1) & 3][0][l], nv4 = ring[level][(r + 1) & 3][1][l];
            const float nu[4] = {nu4.x, nu4.y, nu4.z, nu4.w}, nv[4] = {nv4.x, nv4.y, nv4.z, nv4.w};
            const float4 gx4 = coef[cslot][0][l], gy4 = coef[cslot][1][l], gt4 = coef[cslot][2][l], rd4 = coef[cslot][3][l];
            const float gx[4] = {gx4.x, gx4.y, gx4.z, gx4.w}, gy[4] = {gy4.x, gy4.y, gy4.z, gy4.w};
            const float gt[4] = {gt4.x, gt4.y, gt4.z, gt4.w}, rd[4] = {rd4.x, rd4.y, rd4.z, rd4.w};
            cslot = cslot + 1 == SL_CR ? 0 : cslot + 1;
            float su[6], sv[6], eu[6], ev[6];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                su[k + 1] = __fadd_rn(w0u[k], nu[k]); sv[k + 1] = __fadd_rn(w0v[k], nv[k]);
                eu[k + 1] = w1u[k]; ev[k + 1] = w1v[k];
            }
            su[0] = sl_left(su[4]); su[5] = sl_right(su[1]);
            sv[0] = sl_left(sv[4]); sv[5] = sl_right(sv[1]);
            eu[0] = sl_left(eu[4]); eu[5] = sl_right(eu[1]);
            ev[0] = sl_left(ev[4]); ev[5] = sl_right(ev[1]);
            float tu[4], tv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float ua = hs_avg(__fadd_rn(su[k + 1], __fadd_rn(eu[k], eu[k + 2])), __fadd_rn(su[k], su[k + 2]));
                float va = hs_avg(__fadd_rn(sv[k + 1], __fadd_rn(ev[k], ev[k + 2])), __fadd_rn(sv[k], sv[k + 2]));
                hs_update(ua, va, gx[k], gy[k], gt[k], rd[k], tu[k], tv[k]);
                if (level >= T) { tu[k] = w1u[k]; tv[k] = w1v[k]; }     // fewer than 16 sweeps asked: pass through
            }
            if (level < SL_NL - 1) {
                ring[level + 1][r & 3][0][l] = make_float4(tu[0], tu[1], tu[2], tu[3]);
                ring[level + 1][r & 3][1][l] = make_float4(tv[0], tv[1], tv[2], tv[3]);
            } else if (store_lane && r >= ya && r < yb) {
                long long o = (long long)r * opitch + x0 + 4 * l;
                *reinterpret_cast<float4 *>(Uo + o) = make_float4(tu[0], tu[1], tu[2], tu[3]);
                *reinterpret_cast<float4 *>(Vo + o) = make_float4(tv[0], tv[1], tv[2], tv[3]);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) { w0u[k] = w1u[k]; w0v[k] = w1v[k]; w1u[k] = nu[k]; w1v[k] = nv[k]; }
            ++r;
        }
}

__global__ void __launch_bounds__(SL_THREADS)
k_hs_stream(SlPlanes in, float *__restrict__ Uo, float *__restrict__ Vo, long long opitch, int P, int H, int W, int T,
            int tiles_x, int chunk_rows) {
    __shared__ float4 ring[SL_NL][4][2][32];      // ring[0]: input rows; ring[k+1]: rows of sweep k+1.  [slot][U|V][lane]
    __shared__ float4 coef[SL_CR][4][32];         // [row slot][fx|fy|ft|1/den][lane]
    const int tid = threadIdx.x;
    const int strip = blockIdx.x % tiles_x, chunk = blockIdx.x / tiles_x;
    const int ya = chunk * chunk_rows, yb = min(H, ya + chunk_rows);
    const int x0 = strip * SL_OX - SL_NL;          // image x of strip column 0
    const int rho0 = ya - SL_NL;                   // first input row of the chunk
    const int NS = (yb - ya) + 3 * SL_NL;          // steps until the last level has produced row yb-1
    const bool loader = tid >= SL_NL * 32;

    // ---- loader state -----------------------------------------------------------------------------
    const int lj = tid & 63, lcol = 4 * (lj & 31), lset = (lj >> 5) & 1;
    const float *pa = lset ? in.V : in.U, *pb = lset ? in.fy : in.fx, *pc = lset ? in.rd : in.ft;
    // rows in flight: named registers (an array here ends up in scratch memory, which serialises the prefetch)
#define SL_Q(U_) float4 qa##U_, qb##U_, qc##U_;
    SL_Q(0) SL_Q(1) SL_Q(2) SL_Q(3) SL_Q(4) SL_Q(5) SL_Q(6) SL_Q(7)
#undef SL_Q
    static_assert(SL_PF == 8, "the prefetch registers are spelled out for eight rows");
    // ---- level state ------------------------------------------------------------------------------
    const int level = tid >> 5, l = tid & 31;
    float w0u[4] = {0, 0, 0, 0}, w0v[4] = {0, 0, 0, 0}, w1u[4] = {0, 0, 0, 0}, w1v[4] = {0, 0, 0, 0};
    int r = rho0 - 2 - 2 * level;                              // row this level produces at step 0
    int cslot = (r + 64 * SL_CR) % SL_CR;                      // its coefficient ring slot
    int lslot = (rho0 + 64 * SL_CR) % SL_CR;                   // loader: slot of row rho
    const bool store_lane = l >= SL_NL / 4 && l < 32 - SL_NL / 4 && (x0 + 4 * l) < W;

    if (loader) {
#define SL_FILL(U_)                                                               \
    {                                                                             \
        const int row = min(rho0 + (U_), H + SL_NL - 1);                          \
        const long long o = (long long)row * P + x0 + lcol;                       \
        qa##U_ = *reinterpret_cast<const float4 *>(pa + o);                       \
        qb##U_ = *reinterpret_cast<const float4 *>(pb + o);                       \
        qc##U_ = *reinterpret_cast<const float4 *>(pc + o);                       \
    }
        SL_FILL(0) SL_FILL(1) SL_FILL(2) SL_FILL(3) SL_FILL(4) SL_FILL(5) SL_FILL(6) SL_FILL(7)
#undef SL_FILL
    }

    // one pipeline step; U_ names the prefetch registers of the row this step consumes
#define SL_STEP(U_)                                                                                       \
    {                                                                                                     \
        const int t = t0 + (U_);                                                                           \
        if (t < NS) {                                                                                     \
            sl_step(loader, t, rho0, H, P, x0, lcol, lset, lj, pa, pb, pc, qa##U_, qb##U_, qc##U_, ring, coef, lslot, level, l, r, cslot,  \
                       w0u, w0v, w1u, w1v, T, ya, yb, store_lane, Uo, Vo, opitch);                       \
            __syncthreads();                                                                              \
        }                                                                                                 \
    }
    for (int t0 = 0; t0 < NS; t0 += SL_PF) {
        SL_STEP(0) SL_STEP(1) SL_STEP(2) SL_STEP(3) SL_STEP(4) SL_STEP(5) SL_STEP(6) SL_STEP(7)
    }
#undef SL_STEP
}

// ---- padded copies ------------------------------------------------------------------------------------
// dst (pointer at image (0,0), pitch P) receives src mirrored into a border of SL_NL pixels; mode 1 writes
// 1/(alpha^2 + fx^2 + fy^2) instead (src = fx, src2 = fy)
__global__ void k_sl_pad(const float *__restrict__ src, const float *__restrict__ src2, float *__restrict__ dst, int P, int H,
                         int W, int mode, float a2) {
    int x = blockIdx.x * blockDim.x + threadIdx.x - SL_NL;
    int y = blockIdx.y * blockDim.y + threadIdx.y - SL_NL;
    if (x >= W + SL_NL || y >= H + SL_NL) return;
    size_t s = (size_t)sl_refl(y, H) * W + sl_refl(x, W);
    dst[(long long)y * P + x] = mode ? hs_rden(src[s], src2[s], a2) : src[s];
}
// refresh the mirrored border of a padded plane from its interior
__global__ void k_sl_repad(float *__restrict__ U, float *__restrict__ V, int P, int H, int W) {
    // frame pixels enumerated as: top and bottom bands (2*SL_NL rows x (W+2*SL_NL)), then left/right bands (H rows x 2*SL_NL)
    const long long nband = 2LL * SL_NL * (W + 2 * SL_NL), nside = (long long)H * 2 * SL_NL;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nband + nside) return;
    int x, y;
    if (i < nband) {
        int row = (int)(i / (W + 2 * SL_NL));
        x = (int)(i - (long long)row * (W + 2 * SL_NL)) - SL_NL;
        y = row < SL_NL ? row - SL_NL : H + (row - SL_NL);
    } else {
        long long j = i - nband;
        y = (int)(j / (2 * SL_NL));
        int c = (int)(j - (long long)y * 2 * SL_NL);
        x = c < SL_NL ? c - SL_NL : W + (c - SL_NL);
    }
    long long d = (long long)y * P + x, s = (long long)sl_refl(y, H) * P + sl_refl(x, W);
    U[d] = U[s];
    V[d] = V[s];
}

int ofri_hs_stream_sweeps(void) { return SL_NL; }

// niter sweeps from dense (Uin, Vin) to dense (Uout, Vout); all pointers 16-byte aligned, W % 4 == 0
int ofri_hs_stream_run(ofri_ctx *ctx, const float *Uin, const float *Vin, float *Uout, float *Vout, const float *fx,
                       const float *fy, const float *ft, float a2, int niter, int H, int W, int *launches,
                       hipEvent_t ev_begin, hipEvent_t ev_end) {
    const int tiles_x = ceil_div(W, SL_OX);
    const int P = tiles_x * SL_OX + 2 * SL_NL;                 // multiple of 4; covers the last strip
    const size_t rows = (size_t)H + 2 * SL_NL, plane = rows * P;
    PoolBuf buf(ctx);
    OFRI_TRY(buf.get(8 * plane * sizeof(float)));
    float *base = buf.as<float>();
    OFRI_HIP(hipMemsetAsync(base, 0, 8 * plane * sizeof(float), ctx->stream));   // columns past W+16 stay finite
    auto at00 = [&](int k) { return base + k * plane + (size_t)SL_NL * P + SL_NL; };
    float *pU[2] = {at00(0), at00(2)}, *pV[2] = {at00(1), at00(3)};
    float *pfx = at00(4), *pfy = at00(5), *pft = at00(6), *prd = at00(7);
    {
        dim3 b(64, 4), g(ceil_div(W + 2 * SL_NL, 64), ceil_div(H + 2 * SL_NL, 4));
        k_sl_pad<<<g, b, 0, ctx->stream>>>(Uin, nullptr, pU[0], P, H, W, 0, a2);
        k_sl_pad<<<g, b, 0, ctx->stream>>>(Vin, nullptr, pV[0], P, H, W, 0, a2);
        k_sl_pad<<<g, b, 0, ctx->stream>>>(fx, nullptr, pfx, P, H, W, 0, a2);
        k_sl_pad<<<g, b, 0, ctx->stream>>>(fy, nullptr, pfy, P, H, W, 0, a2);
        k_sl_pad<<<g, b, 0, ctx->stream>>>(ft, nullptr, pft, P, H, W, 0, a2);
        k_sl_pad<<<g, b, 0, ctx->stream>>>(fx, fy, prd, P, H, W, 1, a2);
        OFRI_LAUNCH_CHECK();
    }
    // one workgroup per CU (LDS bound): as many row chunks as fill the chip once
    hipDeviceProp_t prop;
    OFRI_HIP(hipGetDeviceProperties(&prop, ctx->device));
    int ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    int nchunks = ncu / tiles_x;
    if (nchunks < 1) nchunks = 1;
    int chunk_rows = ceil_div(H, nchunks);
    nchunks = ceil_div(H, chunk_rows);
    const int L = ceil_div(niter, SL_NL);
    int remaining = niter, cur = 0;
    if (ev_begin) OFRI_HIP(hipEventRecord(ev_begin, ctx->stream));
    for (int k = 0; k < L; ++k) {
        const int T = remaining < SL_NL ? remaining : SL_NL;
        const bool last = k == L - 1;
        SlPlanes in{pU[cur], pV[cur], pfx, pfy, pft, prd};
        float *uo = last ? Uout : pU[cur ^ 1], *vo = last ? Vout : pV[cur ^ 1];
        k_hs_stream<<<tiles_x * nchunks, SL_THREADS, 0, ctx->stream>>>(in, uo, vo, last ? (long long)W : (long long)P, P, H, W, T,
                                                                       tiles_x, chunk_rows);
        OFRI_LAUNCH_CHECK();
        if (!last) {
            long long n = 2LL * SL_NL * (W + 2 * SL_NL) + (long long)H * 2 * SL_NL;
            k_sl_repad<<<(int)((n + 255) / 256), 256, 0, ctx->stream>>>(uo, vo, P, H, W);
            OFRI_LAUNCH_CHECK();
        }
        cur ^= 1;
        remaining -= T;
    }
    if (ev_end) OFRI_HIP(hipEventRecord(ev_end, ctx->stream));
    if (launches) *launches = L;
    return OFRI_OK;
}

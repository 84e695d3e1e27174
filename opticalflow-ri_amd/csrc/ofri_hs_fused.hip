// Temporally blocked Horn-Schunck: T Jacobi sweeps per launch on a register-resident tile.
//
// A 256-thread workgroup owns a region of 128 x 64 pixels (output tile 112 x 48 plus a
// halo of 8).  Lane l of band b (one band = 32 lanes = 8 rows) keeps a column strip of 4 x 8 pixels of
// U, V and the scaled coefficients A, B (ofri_hs_math.h) in registers, C in LDS.  Per sweep the left/right neighbours come from the adjacent
// lanes (DPP wave shifts), the rows above/below the strip from a small double-buffered LDS array of band
// edge rows; one barrier per sweep.  Pixels within s of the region border are stale after s sweeps, which
// is why only the inner 112 x 48 pixels are stored after T <= 8 sweeps.  Region pixels outside the image
// are loaded from their mirror image ('mirror' border: a field extended by reflection stays the
// reflection of itself under the symmetric averaging stencil), so borders need no special case.
#include "ofri_internal.h"
#include "ofri_hs_math.h"

#define FT_HALO 8
#define FT_RX 128
#define FT_OX (FT_RX - 2 * FT_HALO)
#define FT_BANDS 8                   // bands of 32 lanes
// Rows per thread are a template parameter FT_S: 8 (region 128 x 64, 112 x 48 stored) for fields that fill the GPU,
// 4 (region 128 x 32, 112 x 16 stored) for small ones, where a launch has fewer workgroups than the GPU has CUs and
// its duration is the latency of one workgroup: half the rows per thread is half that latency.
#define FT_RY_OF(S) (FT_BANDS * (S))
#define FT_OY_OF(S) (FT_RY_OF(S) - 2 * FT_HALO)

// lane shifts with bound_ctrl: the wavefront's first / last lane reads 0 (a halo pixel, never stored)
__device__ __forceinline__ float lane_left(float v) {    // value held by lane-1 (wave_shr:1)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xF, 0xF, true));
}
__device__ __forceinline__ float lane_right(float v) {   // value held by lane+1 (wave_shl:1)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xF, 0xF, true));
}
// reflect-101 for any overshoot
__device__ __forceinline__ int refl101_any(int i, int n) {
    if (n == 1) return 0;
    int p = 2 * n - 2;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - i;
}

// One sweep of the workgroup's region, in place on the lane's strip; reads the band edge rows of buffer buf,
// writes the new ones to buffer buf ^ 1.
template <int FT_S>
__device__ __forceinline__ void ft_sweep(float4 (&edge)[2][FT_BANDS + 2][2][2][32], float4 (&ftile)[FT_RY_OF(FT_S)][32], int buf,
                                         int band, int l32, float (&cu)[FT_S][4], float (&cv)[FT_S][4],
                                         const float (&gx)[FT_S][4], const float (&gy)[FT_S][4]) {
        float4 upU = edge[buf][band][1][0][l32], upV = edge[buf][band][1][1][l32];          // bottom row of the band above
        float4 dnU = edge[buf][band + 2][0][0][l32], dnV = edge[buf][band + 2][0][1][l32];  // top row of the band below
        float pu[4] = {upU.x, upU.y, upU.z, upU.w}, pv[4] = {upV.x, upV.y, upV.z, upV.w};
#pragma unroll
        for (int r = 0; r < FT_S; ++r) {
            float nu[4], nv[4];
            if (r < FT_S - 1) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { nu[k] = cu[r + 1][k]; nv[k] = cv[r + 1][k]; }
            } else {
                nu[0] = dnU.x; nu[1] = dnU.y; nu[2] = dnU.z; nu[3] = dnU.w;
                nv[0] = dnV.x; nv[1] = dnV.y; nv[2] = dnV.z; nv[3] = dnV.w;
            }
            // vertical pair sums s and q = 2*centre + s for the lane's columns; q of the columns next to them
            // comes from the adjacent lanes
            float su[4], sv[4], qu[6], qv[6];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                su[k] = __fadd_rn(pu[k], nu[k]); sv[k] = __fadd_rn(pv[k], nv[k]);
                qu[k + 1] = hs_q(cu[r][k], su[k]); qv[k + 1] = hs_q(cv[r][k], sv[k]);
            }
            qu[0] = lane_left(qu[4]); qu[5] = lane_right(qu[1]);
            qv[0] = lane_left(qv[4]); qv[5] = lane_right(qv[1]);
            float tu[4], tv[4];
            const float4 g4 = ftile[band * FT_S + r][l32];
            const float gt[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // the neighbour lane's value goes first (a + b == b + a exactly): as first operand the lane shift
                // folds into the add
                const int p = k == 3 ? k + 2 : k, q = k == 3 ? k : k + 2;
                float ua = hs_avg(su[k], qu[p], qu[q]);
                float va = hs_avg(sv[k], qv[p], qv[q]);
                hs_update(ua, va, gx[r][k], gy[r][k], gt[k], tu[k], tv[k]);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) { pu[k] = cu[r][k]; pv[k] = cv[r][k]; cu[r][k] = tu[k]; cv[r][k] = tv[k]; }
        }
        edge[buf ^ 1][band + 1][0][0][l32] = make_float4(cu[0][0], cu[0][1], cu[0][2], cu[0][3]);
        edge[buf ^ 1][band + 1][0][1][l32] = make_float4(cv[0][0], cv[0][1], cv[0][2], cv[0][3]);
        edge[buf ^ 1][band + 1][1][0][l32] = make_float4(cu[FT_S - 1][0], cu[FT_S - 1][1], cu[FT_S - 1][2], cu[FT_S - 1][3]);
        edge[buf ^ 1][band + 1][1][1][l32] = make_float4(cv[FT_S - 1][0], cv[FT_S - 1][1], cv[FT_S - 1][2], cv[FT_S - 1][3]);
        __syncthreads();
}

template <int FT_S>
__global__ void __launch_bounds__(256, 2)
k_hs_fused(const float *__restrict__ U, const float *__restrict__ V, const float *__restrict__ fx,
           const float *__restrict__ fy, const float *__restrict__ ft, float *__restrict__ Un,
           float *__restrict__ Vn, int H, int W, int T, int tiles_x, int ntiles, int pin, int pc, int pout, int accumulate) {
    // [buffer][slot][top/bottom][U/V][lane]; slot b+1 belongs to band b, slots 0 and FT_BANDS+1 are never written
    __shared__ float4 edge[2][FT_BANDS + 2][2][2][32];
    constexpr int FT_RY = FT_RY_OF(FT_S), FT_OY = FT_OY_OF(FT_S);
    __shared__ float4 ftile[FT_RY][32];          // ft of the region: read-only after the load, kept out of registers
    const int tid = threadIdx.x, l32 = tid & 31, band = tid >> 5;

    // XCD-aware tile order: workgroups b, b+8, b+16, ... share an XCD (and its L2), so give each of the
    // 8 residue classes a contiguous run of tiles; neighbouring tiles share their halo reads
    int b = blockIdx.x, q = ntiles >> 3, r8 = ntiles & 7, xcd = b & 7;
    int t = xcd * q + (xcd < r8 ? xcd : r8) + (b >> 3);
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int gx0 = tx * FT_OX - FT_HALO + l32 * 4;
    const int gyb = ty * FT_OY - FT_HALO + band * FT_S;
    const bool xin = gx0 >= 0 && gx0 + 3 < W;

    float cu[FT_S][4], cv[FT_S][4], gx[FT_S][4], gy[FT_S][4];
#pragma unroll
    for (int r = 0; r < FT_S; ++r) {
        int y = gyb + r;
        if (xin && y >= 0 && y < H) {
            const size_t o = (size_t)y * pin + gx0, oc = (size_t)y * pc + gx0;
            float4 a = *reinterpret_cast<const float4 *>(U + o), bb = *reinterpret_cast<const float4 *>(V + o);
            float4 c = *reinterpret_cast<const float4 *>(fx + oc), d = *reinterpret_cast<const float4 *>(fy + oc);
            float4 e = *reinterpret_cast<const float4 *>(ft + oc);
            cu[r][0] = a.x; cu[r][1] = a.y; cu[r][2] = a.z; cu[r][3] = a.w;
            cv[r][0] = bb.x; cv[r][1] = bb.y; cv[r][2] = bb.z; cv[r][3] = bb.w;
            gx[r][0] = c.x; gx[r][1] = c.y; gx[r][2] = c.z; gx[r][3] = c.w;
            gy[r][0] = d.x; gy[r][1] = d.y; gy[r][2] = d.z; gy[r][3] = d.w;
            ftile[band * FT_S + r][l32] = e;
        } else {
            const int ry = refl101_any(y, H);
            const size_t ro = (size_t)ry * pin, rc = (size_t)ry * pc;
            float e[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int rx = refl101_any(gx0 + k, W);
                cu[r][k] = U[ro + rx]; cv[r][k] = V[ro + rx]; gx[r][k] = fx[rc + rx]; gy[r][k] = fy[rc + rx]; e[k] = ft[rc + rx];
            }
            ftile[band * FT_S + r][l32] = make_float4(e[0], e[1], e[2], e[3]);
        }
    }

    if (tid < 64) {   // the two slots nobody owns: rows outside the region, never valid, keep them finite
        float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        int s = (tid & 32) ? FT_BANDS + 1 : 0;
        edge[0][s][0][0][l32] = z; edge[0][s][0][1][l32] = z; edge[0][s][1][0][l32] = z; edge[0][s][1][1][l32] = z;
        edge[1][s][0][0][l32] = z; edge[1][s][0][1][l32] = z; edge[1][s][1][0][l32] = z; edge[1][s][1][1][l32] = z;
    }
    edge[0][band + 1][0][0][l32] = make_float4(cu[0][0], cu[0][1], cu[0][2], cu[0][3]);
    edge[0][band + 1][0][1][l32] = make_float4(cv[0][0], cv[0][1], cv[0][2], cv[0][3]);
    edge[0][band + 1][1][0][l32] = make_float4(cu[FT_S - 1][0], cu[FT_S - 1][1], cu[FT_S - 1][2], cu[FT_S - 1][3]);
    edge[0][band + 1][1][1][l32] = make_float4(cv[FT_S - 1][0], cv[FT_S - 1][1], cv[FT_S - 1][2], cv[FT_S - 1][3]);
    __syncthreads();

    // two sweeps per round, so that the buffer index is a constant and the strip's registers line up again at
    // the loop's back edge without copies
    int s = 0;
    for (; s + 1 < T; s += 2) {
        ft_sweep<FT_S>(edge, ftile, 0, band, l32, cu, cv, gx, gy);
        ft_sweep<FT_S>(edge, ftile, 1, band, l32, cu, cv, gx, gy);
    }
    if (s < T) ft_sweep<FT_S>(edge, ftile, 0, band, l32, cu, cv, gx, gy);

    // store the inner tile: bands 1..FT_BANDS-2, strips 2..29, clipped to the image.  A strip that straddles the right
    // edge (width not a multiple of 4) is stored whole: the destination's pitch is a multiple of 4, so its last
    // columns land in the row's padding.
    if (band >= FT_HALO / FT_S && band < FT_BANDS - FT_HALO / FT_S && l32 >= FT_HALO / 4 && l32 < 32 - FT_HALO / 4 && gx0 >= 0 && gx0 < W) {
#pragma unroll
        for (int r = 0; r < FT_S; ++r) {
            int y = gyb + r;
            if (y < H) {
                size_t o = (size_t)y * pout + gx0;
                float4 ou = make_float4(cu[r][0], cu[r][1], cu[r][2], cu[r][3]), ov = make_float4(cv[r][0], cv[r][1], cv[r][2], cv[r][3]);
                if (accumulate) {               // the destination holds the flow so far: add the increment to it
                    const float4 pu = *reinterpret_cast<const float4 *>(Un + o), pv = *reinterpret_cast<const float4 *>(Vn + o);
                    ou = make_float4(__fadd_rn(pu.x, ou.x), __fadd_rn(pu.y, ou.y), __fadd_rn(pu.z, ou.z), __fadd_rn(pu.w, ou.w));
                    ov = make_float4(__fadd_rn(pv.x, ov.x), __fadd_rn(pv.y, ov.y), __fadd_rn(pv.z, ov.z), __fadd_rn(pv.w, ov.w));
                }
                *reinterpret_cast<float4 *>(Un + o) = ou;
                *reinterpret_cast<float4 *>(Vn + o) = ov;
            }
        }
    }
}

// The fused kernel needs 16-byte aligned planes whose pitch is a multiple of 4 and at least 16 x 16 pixels; this is
// the test for dense planes (pitch = width).  Other widths go through padded copies (ofri_hs.hip).
bool ofri_hs_fused_ok(const void *const *ptrs, int nptrs, int H, int W) {
    size_t bits = 0;
    for (int i = 0; i < nptrs; ++i) bits |= (size_t)ptrs[i];
    return (bits & 15) == 0 && (W & 3) == 0 && W >= 16 && H >= 16;
}
int ofri_hs_fused_max_sweeps(void) { return FT_HALO; }

// One launch: T <= FT_HALO sweeps from (su, sv) into (du, dv); source and destination must differ.  Every
// pointer is at pixel (0,0) of its plane; pin, pc, pout are the row pitches (in floats, multiples of 4, at least W) of the
// source, coefficient and destination planes; fx, fy, ft hold the scaled coefficients A, B, C.  accumulate: the result is added to what (du, dv) hold.
int ofri_hs_fused_launch_pitched(ofri_ctx *ctx, const float *su, const float *sv, float *du, float *dv, const float *fx,
                                 const float *fy, const float *ft, int T, int H, int W, int pin, int pc, int pout,
                                 bool accumulate) {
    const int tiles_x = ceil_div(W, FT_OX);
    int tiles_y = ceil_div(H, FT_OY_OF(8)), ntiles = tiles_x * tiles_y;
    if (ntiles >= 2 * ofri_ctx_ncu(ctx)) {
        k_hs_fused<8><<<ntiles, 256, 0, ctx->stream>>>(su, sv, fx, fy, ft, du, dv, H, W, T, tiles_x, ntiles, pin, pc, pout, accumulate ? 1 : 0);
    } else {        // too few workgroups to fill the GPU: shorter ones
        tiles_y = ceil_div(H, FT_OY_OF(4)); ntiles = tiles_x * tiles_y;
        k_hs_fused<4><<<ntiles, 256, 0, ctx->stream>>>(su, sv, fx, fy, ft, du, dv, H, W, T, tiles_x, ntiles, pin, pc, pout, accumulate ? 1 : 0);
    }
    OFRI_LAUNCH_CHECK();
    return OFRI_OK;
}
int ofri_hs_fused_launch(ofri_ctx *ctx, const float *su, const float *sv, float *du, float *dv, const float *fx,
                         const float *fy, const float *ft, int T, int H, int W) {
    return ofri_hs_fused_launch_pitched(ctx, su, sv, du, dv, fx, fy, ft, T, H, W, W, W, W, false);
}

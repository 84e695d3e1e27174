// Kernels and launchers of the Farneback path; included twice by ofri_farneback.hip, inside namespace fb_exact (floating
// point contraction off) and inside namespace fb_fast (contraction on).  Not a stand-alone header.
// ---- polynomial expansion (.cl:72-133) -----------------------------------------------------------
#define PE_TX 64
#define PE_TY 8
template <int N>
__global__ void __launch_bounds__(256)
k_fb_polyexp(const float *__restrict__ src, float *__restrict__ dst, int H, int W, FbPoly c) {
    __shared__ float r0[PE_TY][PE_TX + 2 * N], r1[PE_TY][PE_TX + 2 * N], r2[PE_TY][PE_TX + 2 * N];
    const int bx = blockIdx.x * PE_TX, by = blockIdx.y * PE_TY, tid = threadIdx.y * 64 + threadIdx.x;
    // vertical pass; rows clamped (:98-99), columns clamped (:89)
    for (int i = tid; i < PE_TY * (PE_TX + 2 * N); i += 256) {
        int ly = i / (PE_TX + 2 * N), lx = i - ly * (PE_TX + 2 * N);
        int y = by + ly, x = clampi(bx + lx - N, 0, W - 1);
        float a = 0.f, b = 0.f, d = 0.f;
        if (y < H) {
            a = src[(size_t)y * W + x] * c.g[0];
#pragma unroll
            for (int k = 1; k <= N; ++k) {
                float t0 = src[(size_t)max(y - k, 0) * W + x];
                float t1 = src[(size_t)min(y + k, H - 1) * W + x];
                a += c.g[k] * (t0 + t1);
                b += c.xg[k] * (t1 - t0);
                d += c.xxg[k] * (t0 + t1);
            }
        }
        r0[ly][lx] = a; r1[ly][lx] = b; r2[ly][lx] = d;
    }
    __syncthreads();
    for (int i = tid; i < PE_TY * PE_TX; i += 256) {
        int ly = i / PE_TX, lx = i - ly * PE_TX;
        int y = by + ly, x = bx + lx;
        if (y >= H || x >= W) continue;
        const float *p0 = &r0[ly][lx + N], *p1 = &r1[ly][lx + N], *p2 = &r2[ly][lx + N];
        float b1 = c.g[0] * p0[0], b3 = c.g[0] * p1[0], b5 = c.g[0] * p2[0], b2 = 0.f, b4 = 0.f, b6 = 0.f;
#pragma unroll
        for (int k = 1; k <= N; ++k) {
            b1 += (p0[k] + p0[-k]) * c.g[k];
            b4 += (p0[k] + p0[-k]) * c.xxg[k];
            b2 += (p0[k] - p0[-k]) * c.xg[k];
            b3 += (p1[k] + p1[-k]) * c.g[k];
            b6 += (p1[k] - p1[-k]) * c.xg[k];
            b5 += (p2[k] + p2[-k]) * c.g[k];
        }
        size_t o = (size_t)y * W + x, P = (size_t)H * W;
        dst[o] = b3 * c.ig[0];
        dst[P + o] = b2 * c.ig[0];
        dst[2 * P + o] = b1 * c.ig[1] + b5 * c.ig[2];
        dst[3 * P + o] = b1 * c.ig[1] + b4 * c.ig[2];
        dst[4 * P + o] = b6 * c.ig[3];
    }
}

static int fbk_polyexp(ofri_ctx *ctx, const float *src, float *dst5, int rows, int cols, int polyN,
                               const float *g_host, const float *xg_host, const float *xxg_host, const float *ig_host) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(src && dst5 && g_host && xg_host && xxg_host && ig_host, "null pointer");
    OFRI_REQUIRE(rows > 0 && cols > 0, "bad shape");
    OFRI_REQUIRE(polyN == 5 || polyN == 7, "polyN must be 5 or 7");       // Farneback_PyCL.py:463
    FbPoly c;
    memset(&c, 0, sizeof(c));
    memcpy(c.g, g_host, (polyN + 1) * sizeof(float));
    memcpy(c.xg, xg_host, (polyN + 1) * sizeof(float));
    memcpy(c.xxg, xxg_host, (polyN + 1) * sizeof(float));
    memcpy(c.ig, ig_host, 4 * sizeof(float));
    dim3 b(64, 4), g(ceil_div(cols, PE_TX), ceil_div(rows, PE_TY));
    if (polyN == 5) k_fb_polyexp<5><<<g, b, 0, ctx->stream>>>(src, dst5, rows, cols, c);
    else k_fb_polyexp<7><<<g, b, 0, ctx->stream>>>(src, dst5, rows, cols, c);
    OFRI_LAUNCH_CHECK();
    return OFRI_OK;
}

// address spaces of the LDS-DMA loads (__builtin_amdgcn_global_load_lds)
typedef __attribute__((address_space(3))) void fb_lds_void;
typedef const __attribute__((address_space(1))) void fb_gbl_void;

// ---- pre-blur + polynomial expansion in one launch --------------------------------------------------
// The reference blurs every frame with a 3-tap kernel before expanding it (Farneback_PyCL.py:563-585; with the
// default single internal level the resize between the two is an identity).  Here a workgroup stages the source
// tile of its 64 x 16 outputs (polyN + 1 more on every side, reflect-101 like gaussianBlur), blurs it in LDS
// (vertical pass, then horizontal, the order and arithmetic of k_fb_window) and expands the blurred tile with the
// arithmetic of k_fb_polyexp, whose rows and columns are clamped to the image: same bits as the two kernels.
#define BP_TX 64
#define BP_TY 16
// DMA (N + 1 a multiple of 4, rows of a multiple of 4 pixels on a 16-byte aligned base, at least a tile wide): the source
// tile goes straight into LDS as 16-byte groups; tiles at the left and right border fill their mirror columns afterwards.
template <int N, bool DMA = false>
__global__ void __launch_bounds__(256)
k_fb_blur3_polyexp(const float *__restrict__ src, float *__restrict__ dst, int H, int W, float k0, float k1, FbPoly c) {
    constexpr int SW = BP_TX + 2 * N + 2, SH = BP_TY + 2 * N + 2;     // source tile
    constexpr int BW = BP_TX + 2 * N, BH = BP_TY + 2 * N;             // blurred tile: image rows by-N .., columns bx-N ..
    constexpr int NQ = (4 + 2 * N + 3) / 4, BWP = (BP_TX + 4 * NQ - 4 + 3) & ~3;   // 16-byte groups of a window; row pitch of r0, r1, r2
    __shared__ __attribute__((aligned(16))) float pool[SH * SW + BH * SW + 4];   // source tile + vertical pass (+ the 16 bytes the last task of the second blur pass reads beyond it); later r0, r1, r2
    __shared__ __attribute__((aligned(16))) float B[BH][SW];            // blurred tile (BW columns used)
    static_assert(3 * BP_TY * BWP <= SH * SW + BH * SW && BWP >= BW, "the three partial-sum planes fit where the blur's tiles were");
    float (*S)[SW] = reinterpret_cast<float (*)[SW]>(pool);
    float (*V)[SW] = reinterpret_cast<float (*)[SW]>(pool + SH * SW);
    float (*r0)[BWP] = reinterpret_cast<float (*)[BWP]>(pool);
    float (*r1)[BWP] = reinterpret_cast<float (*)[BWP]>(pool + BP_TY * BWP);
    float (*r2)[BWP] = reinterpret_cast<float (*)[BWP]>(pool + 2 * BP_TY * BWP);
    const int bx = blockIdx.x * BP_TX, by = blockIdx.y * BP_TY, tid = threadIdx.x;
    if (DMA) {
        static_assert(!DMA || ((N + 1) % 4 == 0 && SW % 4 == 0 && (SH * (SW / 4)) % 64 == 0 && SH * 8 == 256), "whole 16-byte groups, whole wavefronts");
        constexpr int NG = SH * (SW / 4), NK = (NG + 255) / 256;
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
        for (int k = 0; k < NK; ++k)
            if (256 * k + 64 * wv < NG) {
                const int g = 256 * k + tid, r = g / (SW / 4), c = 4 * (g - r * (SW / 4));
                const int y = fb_idx(by - N - 1 + r, H - 1), x = clampi(bx - N - 1 + c, 0, W - 4);
                __builtin_amdgcn_global_load_lds((fb_gbl_void *)(src + (size_t)y * W + x), (fb_lds_void *)(pool + 4 * (256 * k + 64 * wv)), 16, 0, 0);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (bx < N + 1 || bx + BP_TX + N + 1 > W) {
            __syncthreads();
            const int r = tid >> 3, j = tid & 7, xl = -1 - j, xr = W + j, cl = xl - (bx - N - 1), cr = xr - (bx - N - 1);
            if (cl >= 0) S[r][cl] = S[r][fb_idx(xl, W - 1) - (bx - N - 1)];
            if (cr < SW) S[r][cr] = S[r][fb_idx(xr, W - 1) - (bx - N - 1)];
        }
    } else {   // all of a thread's loads are issued before the first one is used (one at a time they cost a memory latency each)
        constexpr int NLD = (SH * SW + 255) / 256;
        float pre[NLD];
#pragma unroll
        for (int e = 0; e < NLD; ++e) {
            const int i = min(tid + 256 * e, SH * SW - 1), ly = i / SW, lx = i - ly * SW;
            const int y = fb_idx(by - N - 1 + ly, H - 1), x = fb_idx(bx - N - 1 + lx, W - 1);
            pre[e] = src[(size_t)y * W + x];
        }
#pragma unroll
        for (int e = 0; e < NLD; ++e) {
            const int i = tid + 256 * e;
            if (i < SH * SW) pool[i] = pre[e];
        }
    }
    __syncthreads();
    // both blur passes four columns per task (16-byte LDS accesses; the per-element form spent most of its instructions
    // on index arithmetic)
    static_assert(SW % 4 == 0 && (SH * SW) % 4 == 0, "rows of whole 16-byte groups");
    for (int i = tid; i < BH * (SW / 4); i += 256) {
        const int ly = i / (SW / 4), lx = 4 * (i - ly * (SW / 4));
        const float4 s0 = *reinterpret_cast<const float4 *>(&S[ly][lx]), s1 = *reinterpret_cast<const float4 *>(&S[ly + 1][lx]);
        const float4 s2 = *reinterpret_cast<const float4 *>(&S[ly + 2][lx]);
        float4 acc;
        acc.x = s1.x * k0; acc.x += (s0.x + s2.x) * k1;
        acc.y = s1.y * k0; acc.y += (s0.y + s2.y) * k1;
        acc.z = s1.z * k0; acc.z += (s0.z + s2.z) * k1;
        acc.w = s1.w * k0; acc.w += (s0.w + s2.w) * k1;
        *reinterpret_cast<float4 *>(&V[ly][lx]) = acc;
    }
    __syncthreads();
    for (int i = tid; i < BH * (SW / 4); i += 256) {
        const int ly = i / (SW / 4), lx = 4 * (i - ly * (SW / 4));
        // columns lx .. lx + 5 of the row; the last task of a row reads into the next row (the last row: into the pool's
        // spare 16 bytes) for outputs beyond column BW - 1, which nothing reads
        const float4 v0 = *reinterpret_cast<const float4 *>(&V[ly][lx]), v1 = *reinterpret_cast<const float4 *>(&V[ly][lx] + 4);
        float4 acc;
        acc.x = v0.y * k0; acc.x += (v0.x + v0.z) * k1;
        acc.y = v0.z * k0; acc.y += (v0.y + v0.w) * k1;
        acc.z = v0.w * k0; acc.z += (v0.z + v1.x) * k1;
        acc.w = v1.x * k0; acc.w += (v0.w + v1.y) * k1;
        *reinterpret_cast<float4 *>(&B[ly][lx]) = acc;
    }
    __syncthreads();
    // expansion, vertical pass; rows clamped (:98-99), columns clamped (:89).  Entries of B outside the image are never read.
    // A thread takes four stacked outputs of a column from a window of 4 + 2N blurred values in registers (a value of
    // the window serves every output that reaches it: the clamp belongs to the image row, not to the tap).
    static_assert(BP_TY % 4 == 0, "row groups of four");
    for (int i = tid; i < (BP_TY / 4) * BW; i += 256) {
        const int rg = i / BW, lx = i - rg * BW;
        const int col = clampi(bx + lx - N, 0, W - 1) - (bx - N);
        float w[4 + 2 * N];
#pragma unroll
        for (int j = 0; j < 4 + 2 * N; ++j) w[j] = B[clampi(by + 4 * rg + j - N, 0, H - 1) - (by - N)][col];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float a = 0.f, b = 0.f, d = 0.f;
            if (by + 4 * rg + r < H) {
                a = w[r + N] * c.g[0];
#pragma unroll
                for (int k = 1; k <= N; ++k) {
                    const float t0 = w[r + N - k], t1 = w[r + N + k];
                    a += c.g[k] * (t0 + t1);
                    b += c.xg[k] * (t1 - t0);
                    d += c.xxg[k] * (t0 + t1);
                }
            }
            r0[4 * rg + r][lx] = a; r1[4 * rg + r][lx] = b; r2[4 * rg + r][lx] = d;
        }
    }
    __syncthreads();
    // horizontal pass: four adjacent outputs per thread from windows of 4 + 2N partial sums (16-byte LDS reads)
    static_assert(BP_TY * BP_TX == 4 * 256 && BWP % 4 == 0 && 4 + 2 * N <= 4 * NQ, "four outputs per thread");
    {
        const int ly = tid >> 4, xg = tid & 15, y = by + ly;
        float c0[4 * NQ], c1[4 * NQ], c2[4 * NQ];
        const float4 *q0 = reinterpret_cast<const float4 *>(&r0[ly][4 * xg]), *q1 = reinterpret_cast<const float4 *>(&r1[ly][4 * xg]);
        const float4 *q2 = reinterpret_cast<const float4 *>(&r2[ly][4 * xg]);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float4 v0 = q0[q], v1 = q1[q], v2 = q2[q];
            c0[4 * q] = v0.x; c0[4 * q + 1] = v0.y; c0[4 * q + 2] = v0.z; c0[4 * q + 3] = v0.w;
            c1[4 * q] = v1.x; c1[4 * q + 1] = v1.y; c1[4 * q + 2] = v1.z; c1[4 * q + 3] = v1.w;
            c2[4 * q] = v2.x; c2[4 * q + 1] = v2.y; c2[4 * q + 2] = v2.z; c2[4 * q + 3] = v2.w;
        }
        float o0[4], o1[4], o2[4], o3[4], o4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float *p0 = &c0[r + N], *p1 = &c1[r + N], *p2 = &c2[r + N];
            float b1 = c.g[0] * p0[0], b3 = c.g[0] * p1[0], b5 = c.g[0] * p2[0], b2 = 0.f, b4 = 0.f, b6 = 0.f;
#pragma unroll
            for (int k = 1; k <= N; ++k) {
                b1 += (p0[k] + p0[-k]) * c.g[k];
                b4 += (p0[k] + p0[-k]) * c.xxg[k];
                b2 += (p0[k] - p0[-k]) * c.xg[k];
                b3 += (p1[k] + p1[-k]) * c.g[k];
                b6 += (p1[k] - p1[-k]) * c.xg[k];
                b5 += (p2[k] + p2[-k]) * c.g[k];
            }
            o0[r] = b3 * c.ig[0];
            o1[r] = b2 * c.ig[0];
            o2[r] = b1 * c.ig[1] + b5 * c.ig[2];
            o3[r] = b1 * c.ig[1] + b4 * c.ig[2];
            o4[r] = b6 * c.ig[3];
        }
        const int x = bx + 4 * xg;
        const size_t P = (size_t)H * W;
        if (y < H) {
            const size_t o = (size_t)y * W + x;
            if ((W & 3) == 0 && x + 3 < W && (((size_t)dst) & 15) == 0) {
                *reinterpret_cast<float4 *>(dst + o) = make_float4(o0[0], o0[1], o0[2], o0[3]);
                *reinterpret_cast<float4 *>(dst + P + o) = make_float4(o1[0], o1[1], o1[2], o1[3]);
                *reinterpret_cast<float4 *>(dst + 2 * P + o) = make_float4(o2[0], o2[1], o2[2], o2[3]);
                *reinterpret_cast<float4 *>(dst + 3 * P + o) = make_float4(o3[0], o3[1], o3[2], o3[3]);
                *reinterpret_cast<float4 *>(dst + 4 * P + o) = make_float4(o4[0], o4[1], o4[2], o4[3]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (x + r >= W) continue;
                    dst[o + r] = o0[r]; dst[P + o + r] = o1[r]; dst[2 * P + o + r] = o2[r]; dst[3 * P + o + r] = o3[r]; dst[4 * P + o + r] = o4[r];
                }
            }
        }
    }
}

// the border attenuation of updateMatrices (.cl:330-333): one factor per side, multiplied in this order
__device__ __forceinline__ float fb_border_scale(int x, int y, int H, int W) {
    return c_fb_border[min(x, 5)] * c_fb_border[min(y, 5)] * c_fb_border[min(W - x - 1, 5)] * c_fb_border[min(H - y - 1, 5)];
}

// The five matrix entries of N pixels (y, x) from their displacements (dx, dy) and the two expansions; scale = the pixels'
// border attenuation; pixels with on == false are computed on stand-in addresses and not stored.  Written without
// branches and in phases over the N pixels (addresses, every load, arithmetic, stores), so that the loads of all N pixels
// are in flight together: with a branch per pixel every pixel paid three memory latencies in turn.
template <int N>
__device__ __forceinline__ void fb_matrices_n(const int (&x)[N], const int (&y)[N], const float (&dx)[N], const float (&dy)[N],
                                              const float (&scale)[N], const bool (&on)[N], const float *__restrict__ R0,
                                              const float *__restrict__ R1, float *__restrict__ M, int H, int W) {
    FB_PIXEL_CONTRACT
    const size_t P = (size_t)H * W;
    size_t o[N], q[N];
    float fx[N], fy[N], c[N][5], g[N][5][4];
    bool in[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        o[i] = on[i] ? (size_t)y[i] * W + x[i] : 0;
        fx[i] = x[i] + dx[i]; fy[i] = y[i] + dy[i];
        // convert_int(floor(.)): saturate so that absurd displacements stay "outside" instead of wrapping
        const float flx = floorf(fx[i]), fly = floorf(fy[i]);
        const int x1 = (int)fminf(fmaxf(flx, -1e9f), 1e9f), y1 = (int)fminf(fmaxf(fly, -1e9f), 1e9f);
        fx[i] -= x1;
        fy[i] -= y1;
        in[i] = x1 >= 0 && y1 >= 0 && x1 < W - 1 && y1 < H - 1;
        q[i] = in[i] ? (size_t)y1 * W + x1 : 0;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const size_t d1 = in[i] ? 1 : 0, dW = in[i] ? (size_t)W : 0;          // outside: four reads of element 0, unused
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const float *p = R1 + k * P + q[i];
            g[i][k][0] = p[0]; g[i][k][1] = p[d1]; g[i][k][2] = p[dW]; g[i][k][3] = p[dW + d1];
            c[i][k] = R0[k * P + o[i]];
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float a00 = (1.f - fx[i]) * (1.f - fy[i]), a01 = fx[i] * (1.f - fy[i]), a10 = (1.f - fx[i]) * fy[i], a11 = fx[i] * fy[i];
        float b[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) b[k] = a00 * g[i][k][0] + a01 * g[i][k][1] + a10 * g[i][k][2] + a11 * g[i][k][3];
        float r2 = in[i] ? b[0] : 0.f, r3 = in[i] ? b[1] : 0.f;
        const float r4in = (c[i][2] + b[2]) * 0.5f, r5in = (c[i][3] + b[3]) * 0.5f, r6in = (c[i][4] + b[4]) * 0.25f;
        float r4 = in[i] ? r4in : c[i][2], r5 = in[i] ? r5in : c[i][3], r6 = in[i] ? r6in : c[i][4] * 0.5f;
        r2 = (c[i][0] - r2) * 0.5f;
        r3 = (c[i][1] - r3) * 0.5f;
        r2 += r4 * dy[i] + r6 * dx[i];
        r3 += r6 * dy[i] + r5 * dx[i];
        r2 *= scale[i]; r3 *= scale[i]; r4 *= scale[i]; r5 *= scale[i]; r6 *= scale[i];
        if (on[i]) {
            M[o[i]] = r4 * r4 + r6 * r6;
            M[P + o[i]] = (r4 + r5) * r6;
            M[2 * P + o[i]] = r5 * r5 + r6 * r6;
            M[3 * P + o[i]] = r4 * r2 + r6 * r3;
            M[4 * P + o[i]] = r6 * r2 + r5 * r3;
        }
    }
}

__device__ __forceinline__ void fb_matrices_at(int x, int y, float dx, float dy, const float *__restrict__ R0,
                                               const float *__restrict__ R1, float *__restrict__ M, int H, int W) {
    const int xs[1] = {x}, ys[1] = {y};
    const float dxs[1] = {dx}, dys[1] = {dy}, sc[1] = {fb_border_scale(x, y, H, W)};
    const bool on[1] = {true};
    fb_matrices_n<1>(xs, ys, dxs, dys, sc, on, R0, R1, M, H, W);
}

// ---- separable window filter on NP stacked planes (.cl:160-253, 350-406) -------------------------
// One workgroup: 64 x 16 outputs.  Per plane the (16+2k) x (64+2k) source tile goes to LDS once, the
// vertical pass writes 16 x (64+2k) partial sums to LDS, the horizontal pass leaves 4 results per
// thread in registers.  MODE 0: Gaussian taps, reflect-101 border.  MODE 1: box, clamped border.
// FLOW: instead of storing the NP = 5 filtered planes, solve the 2x2 system per pixel (.cl:408-429).
#define BL_TX 64
#define BL_TY 16
template <int NP, int MODE, bool FLOW>
__global__ void __launch_bounds__(256)
k_fb_window(const float *__restrict__ src, float *__restrict__ dst, float *__restrict__ flowx, float *__restrict__ flowy,
            int H, int W, int ksh, FbTaps taps) {
    extern __shared__ float lds[];
    const int TW = BL_TX + 2 * ksh, TH = BL_TY + 2 * ksh;
    float *tin = lds;                     // TH x TW
    float *tv = lds + (size_t)TH * TW;    // BL_TY x TW
    const int bx = blockIdx.x * BL_TX, by = blockIdx.y * BL_TY;
    const size_t P = (size_t)H * W;
    float res[NP][4];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const float *sp = src + p * P;
        for (int ly = threadIdx.y; ly < TH; ly += 4) {
            int y = by + ly - ksh;
            y = MODE == 0 ? fb_idx(y, H - 1) : clampi(y, 0, H - 1);
            for (int lx = threadIdx.x; lx < TW; lx += 64) {
                int x = bx + lx - ksh;
                x = MODE == 0 ? fb_idx(x, W - 1) : clampi(x, 0, W - 1);
                tin[ly * TW + lx] = sp[(size_t)y * W + x];
            }
        }
        __syncthreads();
        for (int ly = threadIdx.y; ly < BL_TY; ly += 4)
            for (int lx = threadIdx.x; lx < TW; lx += 64) {
                const float *c = tin + (size_t)(ly + ksh) * TW + lx;
                float acc = MODE == 0 ? c[0] * taps.k[0] : c[0];
                for (int j = 1; j <= ksh; ++j) {
                    float s = c[-j * TW] + c[j * TW];
                    acc += MODE == 0 ? s * taps.k[j] : s;
                }
                tv[ly * TW + lx] = acc;
            }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float *c = tv + (size_t)(threadIdx.y * 4 + r) * TW + threadIdx.x + ksh;
            float acc = MODE == 0 ? c[0] * taps.k[0] : c[0];
            for (int i = 1; i <= ksh; ++i) {
                float s = c[-i] + c[i];
                acc += MODE == 0 ? s * taps.k[i] : s;
            }
            res[p][r] = acc;
        }
        // the next plane's load overwrites tin only; tv is rewritten after the barrier that follows that load
    }
    const float box_inv = 1.f / ((1 + 2 * ksh) * (1 + 2 * ksh));
    const int x = bx + threadIdx.x;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int y = by + threadIdx.y * 4 + r;
        if (x >= W || y >= H) continue;
        size_t o = (size_t)y * W + x;
        float v[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) v[p] = MODE == 0 ? res[p][r] : res[p][r] * box_inv;
        if (FLOW) {
            float g11 = v[0], g12 = v[NP > 1 ? 1 : 0], g22 = v[NP > 2 ? 2 : 0], h1 = v[NP > 3 ? 3 : 0], h2 = v[NP > 4 ? 4 : 0];
            float detInv = 1.f / (g11 * g22 - g12 * g12 + 1e-3f);
            flowx[o] = (g11 * h2 - g12 * h1) * detInv;
            flowy[o] = (g22 * h1 - g12 * h2) * detInv;
        } else {
#pragma unroll
            for (int p = 0; p < NP; ++p) dst[p * P + o] = v[p];
        }
    }
}

// Same filter for the reference's default window (ksize_half = 16, windowSize 33), with register
// windows: a workgroup makes a 64 x 32 tile; the vertical pass gives each thread 4 stacked outputs of one
// column (36 LDS reads for 4 results instead of 132), three such columns per thread (8 row groups x 96 columns
// over 256 threads: every wavefront does the same work), the horizontal pass 4 adjacent outputs of one row
// fetched as nine 16-byte LDS reads, two rows per thread.  Taps are applied in the reference's order (centre,
// then pairs outward), so results are identical to k_fb_window.
// UM (with FLOW): the matrices of the next iteration are made from the new displacement on the spot (updateMatrices,
// a per-pixel step) and stored to Mnext; the displacement itself is stored only where flowx is given.
#define BW_TY 32
#ifndef FB_UM_NB
#define FB_UM_NB 2
#endif
#ifndef FB_W16_WAVES
#define FB_W16_WAVES 3
#endif
// DMA: the source tiles go from memory straight into LDS (16-byte LDS-DMA loads: no staging registers, no LDS store
// instructions), which brings the kernel under 128 registers, four workgroups per CU; the next plane's tile is fetched
// while the current plane's horizontal pass runs.  Needs rows of a multiple of 4 pixels on a 16-byte aligned base and a
// field at least one tile wide (fb_window_dma_ok); tiles at the left and right border fetch clamped 16-byte groups and
// fill in the border columns from the tile's own interior afterwards.
template <int NP, int MODE, bool FLOW, bool UM = false, bool DMA = false>
__global__ void __launch_bounds__(256, DMA ? 4 : FB_W16_WAVES)
k_fb_window16(const float *__restrict__ src, float *__restrict__ dst, float *__restrict__ flowx, float *__restrict__ flowy,
              int H, int W, FbTaps taps, int tiles_x, int ntiles, const float *__restrict__ R0 = nullptr,
              const float *__restrict__ R1 = nullptr, float *__restrict__ Mnext = nullptr) {
    constexpr int TY = BW_TY, KSH = 16, TW = BL_TX + 2 * KSH, TH = TY + 2 * KSH;      // 96 x 64
    constexpr int NLD = TH / 8 * 3;                                          // loads per thread and plane
    constexpr int NR = TY / 16;                                           // output rows per thread
    __shared__ __attribute__((aligned(16))) float tin[TH][TW];     // 16-byte groups land here by LDS-DMA
    __shared__ __attribute__((aligned(16))) float tv[TY][TW];
    // Workgroups go to the 8 XCDs round-robin by index and every XCD has its own L2: give each XCD a contiguous run of
    // tiles in raster order, so that the tiles whose 16-pixel aprons overlap are in flight on the same L2 (a tile reads
    // three times its own area; without this every apron came from memory again)
    const int b = blockIdx.x, q8 = ntiles >> 3, r8 = ntiles & 7, xcd = b & 7;
    const int t = xcd * q8 + (xcd < r8 ? xcd : r8) + (b >> 3);
    const int bx = (t % tiles_x) * BL_TX, by = (t / tiles_x) * TY, tid = threadIdx.x;
    const size_t P = (size_t)H * W;
    const int ly = tid >> 4, xg = tid & 15;                                  // horizontal pass: row (and row + 16), group of 4 columns
    float res[NP][NR][4] = {};
    // Register staging: this thread's samples of a plane's 64 x 96 source tile are rows (tid>>5) + 8s, columns
    // (tid&31) + 32q; all loads of a plane are issued together, and the next plane's are issued before the current
    // plane is filtered, so their latency hides behind the arithmetic.
    // DMA: the tile's 16-byte groups in row-major order, 256 per round over the workgroup's lanes.
    constexpr int NG = TH * (TW / 4), NK = (NG + 255) / 256;                 // 16-byte groups of the tile; loads per lane
    static_assert(NG % 64 == 0, "whole wavefronts of 16-byte groups");
    constexpr int NOFF = DMA ? NK : TH / 8 + 3;
    unsigned off[NOFF];                 // the launcher guarantees rows*cols < 2^32
    float pre[DMA ? 1 : NLD];
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool xborder = DMA && (bx < KSH || bx + BL_TX + KSH > W);
    if (DMA) {
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int g = min(256 * k + tid, NG - 1), r = g / (TW / 4), c = 4 * (g - r * (TW / 4));
            int y = by + r - KSH;
            y = MODE == 0 ? fb_idx(y, H - 1) : clampi(y, 0, H - 1);
            off[k] = (unsigned)y * (unsigned)W + (unsigned)clampi(bx + c - KSH, 0, W - 4);
        }
    } else {
#pragma unroll
        for (int sidx = 0; sidx < TH / 8; ++sidx) {
            int y = by + (tid >> 5) + 8 * sidx - KSH;
            y = MODE == 0 ? fb_idx(y, H - 1) : clampi(y, 0, H - 1);
            off[sidx] = (unsigned)y * (unsigned)W;
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            int x = bx + (tid & 31) + 32 * q - KSH;
            off[TH / 8 + q] = (unsigned)(MODE == 0 ? fb_idx(x, W - 1) : clampi(x, 0, W - 1));
        }
    }
    auto fetch = [&](const float *sp) {
        if (DMA) {
#pragma unroll
            for (int k = 0; k < NK; ++k)
                if (256 * k + 64 * wv < NG)     // the wavefront's 64 groups of this round lie inside the tile
                    __builtin_amdgcn_global_load_lds((fb_gbl_void *)(sp + off[k]), (fb_lds_void *)(&tin[0][0] + 4 * (256 * k + 64 * wv)), 16, 0, 0);
        } else {
#pragma unroll
            for (int e = 0; e < NLD; ++e) pre[e] = sp[off[e / 3] + off[TH / 8 + e % 3]];
        }
    };
    fetch(src);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        if (DMA) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();            // the tile has landed, and every wavefront is past the horizontal pass before
            if (xborder) {              // columns left of the field and right of it, from their mirror (clamp) columns
                const int r = tid >> 2;
#pragma unroll
                for (int j = 0; j < 4 && r < TH; ++j) {
                    const int jj = 4 * (tid & 3) + j, xl = -1 - jj, xr = W + jj;
                    const int cl = xl - bx + KSH, cr = xr - bx + KSH;
                    if (cl >= 0) tin[r][cl] = tin[r][(MODE == 0 ? fb_idx(xl, W - 1) : 0) - bx + KSH];
                    if (cr < TW) tin[r][cr] = tin[r][(MODE == 0 ? fb_idx(xr, W - 1) : W - 1) - bx + KSH];
                }
                __syncthreads();
            }
        } else {
#pragma unroll
            for (int sidx = 0; sidx < TH / 8; ++sidx)
#pragma unroll
                for (int q = 0; q < 3; ++q) tin[(tid >> 5) + 8 * sidx][(tid & 31) + 32 * q] = pre[sidx * 3 + q];
            __syncthreads();
            if (p + 1 < NP) fetch(src + (size_t)(p + 1) * P);
        }
#pragma unroll 1
        for (int i = tid; i < (TY / 4) * TW; i += 256) {
            const int rg = i / TW, col = i - rg * TW;
            float w[36];
#pragma unroll
            for (int t = 0; t < 36; ++t) w[t] = tin[rg * 4 + t][col];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float acc = MODE == 0 ? w[KSH + r] * taps.k[0] : w[KSH + r];
#pragma unroll
                for (int j = 1; j <= KSH; ++j) {
                    float sm = w[KSH + r - j] + w[KSH + r + j];
                    acc += MODE == 0 ? sm * taps.k[j] : sm;
                }
                tv[rg * 4 + r][col] = acc;
            }
        }
        __syncthreads();
        if (DMA && p + 1 < NP) fetch(src + (size_t)(p + 1) * P);    // the source tile's space is free now
#pragma unroll 1
        for (int h = 0; h < NR; ++h) {
            asm volatile("" ::: "memory");      // one row's window at a time in registers
            float c[36];
            const float4 *row4 = reinterpret_cast<const float4 *>(&tv[ly + 16 * h][4 * xg]);
#pragma unroll
            for (int q = 0; q < 9; ++q) { float4 v = row4[q]; c[4 * q] = v.x; c[4 * q + 1] = v.y; c[4 * q + 2] = v.z; c[4 * q + 3] = v.w; }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float acc = MODE == 0 ? c[KSH + k] * taps.k[0] : c[KSH + k];
#pragma unroll
                for (int i = 1; i <= KSH; ++i) {
                    float sm = c[KSH + k - i] + c[KSH + k + i];
                    acc += MODE == 0 ? sm * taps.k[i] : sm;
                }
                // the sum is made here: left alone the compiler moves a plane's whole horizontal pass past the following
                // barriers to where the result is first used and keeps the row's window in scratch memory until then
                asm volatile("" : "+v"(acc));
                // res indexed by h would live in scratch memory (h is a loop variable): select into registers instead
#pragma unroll
                for (int hh = 0; hh < NR; ++hh) res[p][hh][k] = h == hh ? acc : res[p][hh][k];
            }
        }
        // next plane: tin is rewritten before, tv after the barrier that follows the tin load
    }
    const float box_inv = 1.f / ((1 + 2 * KSH) * (1 + 2 * KSH));
    const int x0 = bx + 4 * xg;
    const bool vec_ok = FLOW ? (UM || ((((size_t)flowx | (size_t)flowy) & 15) == 0)) : ((((size_t)dst) & 15) == 0);
    // UM: the displacements of the tile go through LDS (the source tile's space: every thread is past its last read of
    // it), so that the per-pixel step after them runs with one pixel per lane along rows (coalesced gathers and stores)
    float (*tf)[TY][BL_TX] = reinterpret_cast<float (*)[TY][BL_TX]>(&tin[0][0]);
    static_assert(sizeof(float) * 2 * TY * BL_TX <= sizeof(float) * TH * TW, "the displacement tile fits in the source tile");
    const bool vec4 = (W & 3) == 0 && vec_ok;     // a thread's four pixels are one aligned 16-byte group
#pragma unroll
    for (int h = 0; h < NR; ++h) {
        const int y = by + ly + 16 * h;
        if (y >= H) continue;
        float ox[4], oy[4], ov[NP][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) { v[p] = MODE == 0 ? res[p][h][k] : res[p][h][k] * box_inv; ov[p][k] = v[p]; }
            if (FLOW) {
                float g11 = v[0], g12 = v[NP > 1 ? 1 : 0], g22 = v[NP > 2 ? 2 : 0], h1 = v[NP > 3 ? 3 : 0], h2 = v[NP > 4 ? 4 : 0];
                float detInv = 1.f / (g11 * g22 - g12 * g12 + 1e-3f);
                ox[k] = (g11 * h2 - g12 * h1) * detInv; oy[k] = (g22 * h1 - g12 * h2) * detInv;
            }
        }
        if (FLOW && UM) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { tf[0][ly + 16 * h][4 * xg + k] = ox[k]; tf[1][ly + 16 * h][4 * xg + k] = oy[k]; }
        } else if (vec4 && x0 + 3 < W) {
            const size_t o = (size_t)y * W + x0;
            if (FLOW) {
                *reinterpret_cast<float4 *>(flowx + o) = make_float4(ox[0], ox[1], ox[2], ox[3]);
                *reinterpret_cast<float4 *>(flowy + o) = make_float4(oy[0], oy[1], oy[2], oy[3]);
            } else {
#pragma unroll
                for (int p = 0; p < NP; ++p) *reinterpret_cast<float4 *>(dst + p * P + o) = make_float4(ov[p][0], ov[p][1], ov[p][2], ov[p][3]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int x = x0 + k;
                if (x >= W) continue;
                const size_t o = (size_t)y * W + x;
                if (FLOW) { flowx[o] = ox[k]; flowy[o] = oy[k]; }
                else {
#pragma unroll
                    for (int p = 0; p < NP; ++p) dst[p * P + o] = ov[p][k];
                }
            }
        }
    }
    if (FLOW && UM) {
        __syncthreads();
        constexpr int NB = FB_UM_NB;                            // pixels of a thread in flight together
        const int rx = tid & 63, x = bx + rx;
        const bool xin = x < W;
        const float bl = c_fb_border[min(xin ? x : 0, 5)], br = c_fb_border[min(xin ? W - x - 1 : 0, 5)];
#pragma unroll 1
        for (int j0 = 0; j0 < TY / 4; j0 += NB) {
            int xs[NB], ys[NB];
            float ux[NB], uy[NB], sc[NB];
            bool on[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int ry = __builtin_amdgcn_readfirstlane(tid >> 6) + 4 * (j0 + i), y = by + ry;     // one row per wavefront
                const int yc = min(y, H - 1);
                xs[i] = x; ys[i] = y; on[i] = xin && y < H;
                ux[i] = tf[0][ry][rx]; uy[i] = tf[1][ry][rx];
                sc[i] = bl * c_fb_border[min(yc, 5)] * br * c_fb_border[min(H - yc - 1, 5)];
                if (flowx && on[i]) { flowx[(size_t)y * W + x] = ux[i]; flowy[(size_t)y * W + x] = uy[i]; }
            }
            fb_matrices_n<NB>(xs, ys, ux, uy, sc, on, R0, R1, Mnext, H, W);
        }
    }
}

// the LDS-DMA variant's conditions: 16-byte groups of a row stay inside the row and are aligned
static inline bool fb_window_dma_ok(const float *src, int cols) { return (cols & 3) == 0 && cols >= BL_TX && (((size_t)src) & 15) == 0; }

template <int NP, int MODE, bool FLOW>
static int fb_window_launch(ofri_ctx *ctx, const float *src, float *dst, float *fx, float *fy, int rows, int cols,
                            const float *taps_host, int ksh) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(src && (FLOW ? (fx && fy) : dst != nullptr), "null pointer");
    OFRI_REQUIRE(rows > 0 && cols > 0, "bad shape");
    OFRI_REQUIRE(ksh >= 0 && ksh <= FB_MAX_KSH, "ksize_half out of range (0..%d)", FB_MAX_KSH);
    OFRI_REQUIRE(MODE == 1 || taps_host, "null taps");
    OFRI_REQUIRE(src != dst, "window filter cannot run in place");
    FbTaps t;
    memset(&t, 0, sizeof(t));
    if (MODE == 0) memcpy(t.k, taps_host, (ksh + 1) * sizeof(float));
    if (ksh == 16 && (size_t)rows * cols < 0xffffffffull) {
        const int tx16 = ceil_div(cols, BL_TX), nt16 = tx16 * ceil_div(rows, BW_TY);
        if (fb_window_dma_ok(src, cols)) k_fb_window16<NP, MODE, FLOW, false, true><<<nt16, 256, 0, ctx->stream>>>(src, dst, fx, fy, rows, cols, t, tx16, nt16);
        else k_fb_window16<NP, MODE, FLOW><<<nt16, 256, 0, ctx->stream>>>(src, dst, fx, fy, rows, cols, t, tx16, nt16);
        OFRI_LAUNCH_CHECK();
        return OFRI_OK;
    }
    size_t lds = ((size_t)(BL_TY + 2 * ksh) * (BL_TX + 2 * ksh) + (size_t)BL_TY * (BL_TX + 2 * ksh)) * sizeof(float);
    OFRI_REQUIRE(lds <= 160 * 1024, "window too large for LDS");
    auto kern = k_fb_window<NP, MODE, FLOW>;
    if (lds > 64 * 1024) OFRI_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 b(64, 4), g(ceil_div(cols, BL_TX), ceil_div(rows, BL_TY));
    kern<<<g, b, lds, ctx->stream>>>(src, dst, fx, fy, rows, cols, ksh, t);
    OFRI_LAUNCH_CHECK();
    return OFRI_OK;
}

static int fbk_blur(ofri_ctx *ctx, const float *src, float *dst, int rows, int cols,
                            const float *half_taps_host, int ksize_half) {
    return fb_window_launch<1, 0, false>(ctx, src, dst, nullptr, nullptr, rows, cols, half_taps_host, ksize_half);
}
static int fbk_blur5(ofri_ctx *ctx, const float *src5, float *dst5, int rows, int cols,
                             const float *half_taps_host, int ksize_half) {
    return fb_window_launch<5, 0, false>(ctx, src5, dst5, nullptr, nullptr, rows, cols, half_taps_host, ksize_half);
}
static int fbk_box5(ofri_ctx *ctx, const float *src5, float *dst5, int rows, int cols, int ksize_half) {
    return fb_window_launch<5, 1, false>(ctx, src5, dst5, nullptr, nullptr, rows, cols, nullptr, ksize_half);
}
static int fbk_window5_update_flow(ofri_ctx *ctx, const float *M5, float *flowx, float *flowy, int rows, int cols,
                                           const float *half_taps_host, int ksize_half, int box) {
    if (box) return fb_window_launch<5, 1, true>(ctx, M5, nullptr, flowx, flowy, rows, cols, nullptr, ksize_half);
    return fb_window_launch<5, 0, true>(ctx, M5, nullptr, flowx, flowy, rows, cols, half_taps_host, ksize_half);
}

__global__ void k_fb_update_matrices(const float *__restrict__ flowx, const float *__restrict__ flowy,
                                     const float *__restrict__ R0, const float *__restrict__ R1, float *__restrict__ M,
                                     int H, int W) {
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= W || y >= H) return;
    const size_t o = (size_t)y * W + x;
    fb_matrices_at(x, y, flowx[o], flowy[o], R0, R1, M, H, W);
}

static int fbk_update_matrices(ofri_ctx *ctx, const float *flowx, const float *flowy, const float *R0,
                                       const float *R1, float *M, int rows, int cols) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(flowx && flowy && R0 && R1 && M, "null pointer");
    OFRI_REQUIRE(rows > 0 && cols > 0, "bad shape");
    dim3 b(64, 4), g(ceil_div(cols, 64), ceil_div(rows, 4));
    k_fb_update_matrices<<<g, b, 0, ctx->stream>>>(flowx, flowy, R0, R1, M, rows, cols);
    OFRI_LAUNCH_CHECK();
    return OFRI_OK;
}

// One iteration of the reference's flow loop from the second on (Farneback_PyCL.py:442-457): window filter of the matrices
// + 2x2 solve + the next matrices in one launch for the default window (ksize_half 16, Gaussian or box); otherwise the
// separate steps.  flowx, flowy may be null when the displacement of this iteration is not wanted (the fused launch then
// does not store it); M5_next must not be M5.
static int fbk_window5_flow_matrices(ofri_ctx *ctx, const float *M5, float *flowx, float *flowy, const float *R0,
                                             const float *R1, float *M5_next, int rows, int cols,
                                             const float *half_taps_host, int ksize_half, int box) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(M5 && R0 && R1 && M5_next && M5_next != M5, "null or aliased pointer");
    OFRI_REQUIRE((flowx == nullptr) == (flowy == nullptr), "flowx and flowy must both be given or both be null");
    OFRI_REQUIRE(rows > 0 && cols > 0, "bad shape");
    OFRI_REQUIRE(box || half_taps_host, "null taps");
    if (ksize_half == 16 && (size_t)rows * cols < 0xffffffffull) {
        FbTaps t;
        memset(&t, 0, sizeof(t));
        if (!box) memcpy(t.k, half_taps_host, 17 * sizeof(float));
        const int tx16 = ceil_div(cols, BL_TX), nt16 = tx16 * ceil_div(rows, BW_TY);
        const bool dma = fb_window_dma_ok(M5, cols);
        if (box && dma) k_fb_window16<5, 1, true, true, true><<<nt16, 256, 0, ctx->stream>>>(M5, nullptr, flowx, flowy, rows, cols, t, tx16, nt16, R0, R1, M5_next);
        else if (box) k_fb_window16<5, 1, true, true><<<nt16, 256, 0, ctx->stream>>>(M5, nullptr, flowx, flowy, rows, cols, t, tx16, nt16, R0, R1, M5_next);
        else if (dma) k_fb_window16<5, 0, true, true, true><<<nt16, 256, 0, ctx->stream>>>(M5, nullptr, flowx, flowy, rows, cols, t, tx16, nt16, R0, R1, M5_next);
        else k_fb_window16<5, 0, true, true><<<nt16, 256, 0, ctx->stream>>>(M5, nullptr, flowx, flowy, rows, cols, t, tx16, nt16, R0, R1, M5_next);
        OFRI_LAUNCH_CHECK();
        return OFRI_OK;
    }
    PoolBuf tx(ctx), ty(ctx);
    if (!flowx) {
        OFRI_TRY(tx.get((size_t)rows * cols * sizeof(float))); OFRI_TRY(ty.get((size_t)rows * cols * sizeof(float)));
        flowx = tx.as<float>(); flowy = ty.as<float>();
    }
    OFRI_TRY(fbk_window5_update_flow(ctx, M5, flowx, flowy, rows, cols, half_taps_host, ksize_half, box));
    return fbk_update_matrices(ctx, flowx, flowy, R0, R1, M5_next, rows, cols);
}

// gaussianBlur followed by polynomialExpansion of the blurred image (same size): one launch for the 3-tap blur
// (ksize_half 1, the reference's pre-blur with one internal pyramid level), the two steps otherwise.
static int fbk_blur_polyexp(ofri_ctx *ctx, const float *src, float *dst5, int rows, int cols,
                                    const float *blur_half_taps_host, int ksize_half, int polyN, const float *g_host,
                                    const float *xg_host, const float *xxg_host, const float *ig_host) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(src && dst5 && blur_half_taps_host && g_host && xg_host && xxg_host && ig_host, "null pointer");
    OFRI_REQUIRE(rows > 0 && cols > 0, "bad shape");
    OFRI_REQUIRE(polyN == 5 || polyN == 7, "polyN must be 5 or 7");
    if (ksize_half != 1) {
        PoolBuf tmp(ctx);
        OFRI_TRY(tmp.get((size_t)rows * cols * sizeof(float)));
        OFRI_TRY(fbk_blur(ctx, src, tmp.as<float>(), rows, cols, blur_half_taps_host, ksize_half));
        return fbk_polyexp(ctx, tmp.as<float>(), dst5, rows, cols, polyN, g_host, xg_host, xxg_host, ig_host);
    }
    FbPoly c;
    memset(&c, 0, sizeof(c));
    memcpy(c.g, g_host, (polyN + 1) * sizeof(float));
    memcpy(c.xg, xg_host, (polyN + 1) * sizeof(float));
    memcpy(c.xxg, xxg_host, (polyN + 1) * sizeof(float));
    memcpy(c.ig, ig_host, 4 * sizeof(float));
    dim3 g(ceil_div(cols, BP_TX), ceil_div(rows, BP_TY));
    if (polyN == 5) k_fb_blur3_polyexp<5><<<g, 256, 0, ctx->stream>>>(src, dst5, rows, cols, blur_half_taps_host[0], blur_half_taps_host[1], c);
    else if (fb_window_dma_ok(src, cols)) k_fb_blur3_polyexp<7, true><<<g, 256, 0, ctx->stream>>>(src, dst5, rows, cols, blur_half_taps_host[0], blur_half_taps_host[1], c);
    else k_fb_blur3_polyexp<7><<<g, 256, 0, ctx->stream>>>(src, dst5, rows, cols, blur_half_taps_host[0], blur_half_taps_host[1], c);
    OFRI_LAUNCH_CHECK();
    return OFRI_OK;
}

// ---- updateFlow (.cl:408-429) -----------------------------------------------------------------------
__global__ void k_fb_update_flow(const float *__restrict__ M, float *__restrict__ flowx, float *__restrict__ flowy, size_t P) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < P; i += st) {
        float g11 = M[i], g12 = M[P + i], g22 = M[2 * P + i], h1 = M[3 * P + i], h2 = M[4 * P + i];
        float detInv = 1.f / (g11 * g22 - g12 * g12 + 1e-3f);
        flowx[i] = (g11 * h2 - g12 * h1) * detInv;
        flowy[i] = (g22 * h1 - g12 * h2) * detInv;
    }
}
static int fbk_update_flow(ofri_ctx *ctx, const float *M, float *flowx, float *flowy, int rows, int cols) {
    OFRI_TRY(ofri_check_ctx(ctx));
    OFRI_REQUIRE(M && flowx && flowy, "null pointer");
    OFRI_REQUIRE(rows > 0 && cols > 0, "bad shape");
    size_t P = (size_t)rows * cols, g = (P + 255) / 256;
    if (g > 4096) g = 4096;
    k_fb_update_flow<<<(int)g, 256, 0, ctx->stream>>>(M, flowx, flowy, P);
    OFRI_LAUNCH_CHECK();
    return OFRI_OK;
}

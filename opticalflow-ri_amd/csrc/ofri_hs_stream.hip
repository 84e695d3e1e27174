// Horn-Schunck sweeps as a row-streaming pipeline: 12 Jacobi sweeps per launch.
//
// A workgroup owns a vertical strip 128 pixels wide (104 stored, halo 12 each side) of a chunk of rows and
// walks down it one row per step.  Six half-wavefronts are six pipeline stages of two sweeps each: at
// a step, stage s turns rows r-1, r, r+1 of its input into row r of sweep 2s+1 and, from that row and the
// two it made in the steps before, row r-1 of sweep 2s+2, which it hands to stage s+1 through a two-slot
// LDS ring.  The rows a stage works on stay in its registers; a row's scaled coefficients A, B, C
// (ofri_hs_math.h) come from an LDS ring once per stage and serve both of its sweeps.  A fourth wavefront streams the next rows from HBM
// into the rings, six rows ahead.  One barrier per step; four workgroups (40 KB of LDS each) share a compute
// unit.  Each input pixel is read once per 12 sweeps (times the strip overlap 128/104), which the tile kernel cannot reach:
// its halo grows in both directions.
//
// Borders ('mirror' of the reference: a field extended by reflection stays the reflection of itself under the
// symmetric stencil) are the loader's business: rows outside the image are the reflected rows, and in the two
// outermost strips the lanes beyond the edge gather reflected columns; the stages never see a border.
// Arithmetic is the shared sequence of ofri_hs_math.h, so the result is bit-identical to the other
// Horn-Schunck kernels.
#include "ofri_internal.h"
#include "ofri_hs_math.h"
#include "ofri_hs_stream.h"
#include "ofri_prio.h"

#define SL_PF 6                          // rows the loader keeps in flight
// The depth is a template parameter.  Built: NL = 12 sweeps per launch (three stage wavefronts + loader, 40 KB of LDS,
// four workgroups per CU) for the bulk, and NL = 16 (four stage wavefronts + loader, 52 KB, three workgroups per CU) for
// a tail of 13 to 16 sweeps: 263 us per launch at 4096 x 4096 against 165 us for 12 sweeps, so 16.4 against 13.7 us per
// sweep, but cheaper than a launch of 12 and a launch for the rest (with four coefficient planes and two workgroups per
// CU it took 327 us).
// LN: lanes of a stage, 32 (a stage is half a wavefront, strips of 128 columns) or 64 (a whole wavefront, strips of 256
// columns: the sweeps' lost columns are 24 of 256 instead of 24 of 128, at twice the LDS per workgroup; pays where the
// row chunks stay long with two workgroups per CU, i.e. on 4096 x 4096 fields).
// K: sweeps per stage.  A stage keeps two rows per sweep and the coefficients of its last K - 1 rows in registers; per step it
// reads one row and one row of coefficients from LDS and writes one row, whatever K is, so the LDS traffic per sweep (which
// is what a step costs before any arithmetic: 47 KB per step and workgroup at K = 2 in wide strips, at 128 bytes per clock
// and CU) falls with 1 / K, against more registers per stage (K = 3: about 150).
template <int NL, int LN = 32, int K = 2> struct SlCfg {
    static_assert(NL % K == 0, "whole stages");
    static constexpr int NS = NL / K;                // pipeline stages
    static constexpr int RX = 4 * LN;                // strip width
    static constexpr int OX = RX - 2 * NL;           // stored width: the sweeps eat NL columns on each side
    static constexpr int LAG = (K + 1) * NS;         // the last stage's last sweep runs this many rows behind the loader
    static constexpr int CR = LAG;                   // coefficient ring rows: rows rho-(LAG-1) .. rho are live
    static constexpr int THREADS = (NS * LN + 63) / 64 * 64 + 2 * LN;     // stages (whole wavefronts) + the loader: two sets of LN lanes (U, A, C and V, B)
    static constexpr int LDS = (NS * 2 * 2 + CR * 3) * LN * 16;
    // workgroups per CU that the LDS (K = 2, LN = 32: 40 KB each at NL = 12, 52 KB at 16) and the registers allow (K = 2: 128,
    // four wavefronts per SIMD; K = 3: 168, three)
    static constexpr int PER_CU = K == 2 ? (LN == 64 ? (NL == 12 ? 2 : 1) : (NL == 12 ? 4 : 3))      // (16 deep in wide strips: 106 KB of LDS, one)
                                         : (LN == 64 ? 2 : 4);
    static constexpr int MIN_CHUNK = 128;            // chunk rows below which one workgroup per CU fewer is launched
};
#define SL_NL_MAIN 12
#ifndef SL_K_MAIN
#define SL_K_MAIN 2                      // sweeps per stage of the 12-sweep launches
#endif
#define SL_NL_DEEP 16

__device__ __forceinline__ float sl_left(float v) {     // lane-1 (wave_shr:1)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xF, 0xF, true));
}
__device__ __forceinline__ float sl_right(float v) {    // lane+1 (wave_shl:1)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xF, 0xF, true));
}
__device__ __forceinline__ int ceil_div_dev(int a, int b) { return (a + b - 1) / b; }
__device__ __forceinline__ int sl_refl(int i, int n) {
    if (n == 1) return 0;
    int p = 2 * n - 2;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - i;
}

#ifdef OFRI_SL_TRACE
// measurement build only (tools/hs_wg_trace.py): start, end and placement of every workgroup of the last launch
#define SL_TRACE_MAX 4096
__device__ unsigned long long g_sl_trace[SL_TRACE_MAX * 3];
extern "C" int ofri_debug_sl_trace(unsigned long long *host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_sl_trace), sizeof(unsigned long long) * 3 * (n < SL_TRACE_MAX ? n : SL_TRACE_MAX)) == hipSuccess ? 0 : 1;
}
extern "C" int ofri_debug_sl_trace_clear(void) {
    void *p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_sl_trace)) != hipSuccess) return 1;
    return hipMemset(p, 0, sizeof(unsigned long long) * 3 * SL_TRACE_MAX) == hipSuccess ? 0 : 1;
}
#endif
// SIMD (0..3) of the CU that the calling wavefront runs on
__device__ __forceinline__ int sl_simd_id() {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    return (hw >> 4) & 3;
}
// Which wavefronts of a workgroup load and which compute.  A stage wavefront issues about 120 vector instructions per
// step and a loader next to none, and a step ends when the busiest SIMD is done.  The hardware deals a workgroup's
// wavefronts over the SIMDs in the order 0 2 1 3 from wherever the workgroup before it on that CU stopped, so with
// fixed roles two workgroups of 6 + 2 wavefronts often put their loaders on the same SIMDs: 4 stage wavefronts on one
// SIMD and 2 on another, and a step takes the time of four.  The roles are therefore dealt at run time.  A loader goes
// to the SIMD that holds most of this workgroup's wavefronts (five wavefronts: the SIMD with two, which leaves one
// stage wavefront per SIMD); among equals to the SIMD this workgroup's dispatch round prefers (round = blockIdx / CUs:
// the workgroups that share a CU come from different rounds, so their loaders end up on different SIMDs and the
// stage wavefronts three per SIMD).  Returns the wavefront's role: 0 .. NW - NLW - 1 stages in wavefront order,
// NW - NLW .. NW - 1 loaders.  Any placement gives a valid set of roles; only the balance depends on it.
template <int NW, int NLW>
__device__ __forceinline__ int sl_wave_role(int *wave_simd, int round) {
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x / 64);
    if ((threadIdx.x & 63) == 0) wave_simd[w] = sl_simd_id();
    __syncthreads();
    unsigned cnt = 0;                              // wavefronts of this workgroup per SIMD, a byte each
    for (int i = 0; i < NW; ++i) cnt += 1u << (8 * wave_simd[i]);
    int pick0 = -1, pick1 = -1;
    for (int k = 0; k < NLW; ++k) {
        unsigned mx = 0;
        for (int q = 0; q < 4; ++q) mx = max(mx, (cnt >> (8 * q)) & 255u);
        const int start = NLW == 2 ? 2 * (round & 1) + k : (round & 3);
        int sd = start & 3;
        for (int j = 3; j >= 0; --j) {
            const int q = (start + j) & 3;
            if (((cnt >> (8 * q)) & 255u) == mx) sd = q;
        }
        int pk = -1;
        for (int i = 0; i < NW; ++i)
            if (wave_simd[i] == sd && i != pick0) pk = i;          // the last such wavefront
        if (pk < 0) pk = pick0 == NW - 1 ? NW - 2 : NW - 1;        // cannot happen; keeps the roles a permutation
        if (k == 0) pick0 = pk; else pick1 = pk;
        cnt -= 1u << (8 * sd);
    }
    if (w == pick0) return NW - NLW;
    if (NLW == 2 && w == pick1) return NW - 1;
    return w - (pick0 < w ? 1 : 0) - (NLW == 2 && pick1 < w ? 1 : 0);
}
struct SlPlanes { const float *U, *V, *fx, *fy, *ft; };        // dense planes (fx, fy, ft: scaled A, B, C); V, fy one plane after U, fx
// A pixel's (u, v) is one register pair and the sweeps run in the packed arithmetic of ofri_hs_math.h.
typedef hs_f2 f2;
struct SlRow { f2 p[4]; };                                     // (u, v) of a lane's four pixels of one row
struct SlCoef { f2 ab[4]; float c[4]; };                       // (A, B) and C of a lane's four pixels
__device__ __forceinline__ f2 sl_left2(f2 v) { return (f2){sl_left(v.x), sl_left(v.y)}; }
__device__ __forceinline__ f2 sl_right2(f2 v) { return (f2){sl_right(v.x), sl_right(v.y)}; }
__device__ __forceinline__ void sl_unpack(const float4 &lo, const float4 &hi, f2 (&p)[4]) {
    p[0] = (f2){lo.x, lo.y}; p[1] = (f2){lo.z, lo.w}; p[2] = (f2){hi.x, hi.y}; p[3] = (f2){hi.z, hi.w};
}

// one sweep of a row: a = row above, c = the row, b = row below; pass (only when !FULL) = copy the row instead
template <bool FULL>
__device__ __forceinline__ SlRow sl_sweep(const SlRow &a, const SlRow &c, const SlRow &b, const SlCoef &g, bool pass, f2 w12) {
    f2 s[4], q[6];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        s[k] = hs_pk_add(a.p[k], b.p[k]);                      // vertical pair sum
        q[k + 1] = hs_q_pk(c.p[k], s[k]);
    }
    q[0] = sl_left2(q[4]); q[5] = sl_right2(q[1]);
    SlRow o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        o.p[k] = hs_step_pk(s[k], q[k], q[k + 2], g.ab[k], g.c[k], w12);
        if (!FULL && pass) o.p[k] = c.p[k];
    }
    return o;
}

template <int K> struct SlStage {    // registers of one pipeline stage (one lane)
    SlRow r0[K], r1[K];              // sweep j of the stage (0-based) makes row rA-j: r0[j], r1[j] are rows rA-j-1, rA-j of its input
    SlCoef gp[K > 1 ? K - 1 : 1];    // coefficients of rows rA-1 .. rA-(K-1)
    float4 au0, av0, au1, av1, au2, av2;   // (accumulating store) what the destination holds in the next three rows to store
    int rA;                          // row the first sweep produces at this step
    int cslot;                       // coefficient ring slot of row rA
};

// one step of a stage: sweep j at row rA-j, j = 0 .. K-1, each from the row the sweep before it has just made.  ACC: the last
// stage adds its row to what the destination holds; those values are fetched three steps (rows) ahead by every lane of the
// wavefront that holds the last stage (LASTW), from addresses clamped into the chunk (accU, accV: the row three below rowU,
// rowV; lane_safe: the lane's columns clamped into the image), so that the loads are unconditional: a conditional load makes
// the rotation of the three registers a real move, which waits for the load just issued, and the compiler then waits for
// every store as well.
template <int NL, int LN, int K, bool FULL, bool ACC, bool LASTW>
__device__ __forceinline__ void sl_stage_step(SlStage<K> &st, float4 (&ring)[SlCfg<NL, LN, K>::NS][2][2][LN],
                                              float4 (&coef)[SlCfg<NL, LN, K>::CR][3][LN], int stage, int l, const bool (&pass)[K], bool all_pass,
                                              bool store_lane, int lane_off, float *__restrict__ rowU, float *__restrict__ rowV,
                                              bool row_ok, const float *__restrict__ accU, const float *__restrict__ accV, int lane_safe,
                                              bool idle, f2 w12) {
    constexpr int SL_NS = SlCfg<NL, LN, K>::NS, SL_CR = SlCfg<NL, LN, K>::CR;
    // newest input row rA+1, written one step ago
    // ring and coefficient rows hold a lane's pixels interleaved: [0] = u0 v0 u1 v1 (A0 B0 A1 B1), [1] = u2 v2 u3 v3
    const float4 n01 = ring[stage][(st.rA + 1) & 1][0][l], n23 = ring[stage][(st.rA + 1) & 1][1][l];
    SlCoef g;
    {
        const float4 g01 = coef[st.cslot][0][l], g23 = coef[st.cslot][1][l], gc = coef[st.cslot][2][l];
        sl_unpack(g01, g23, g.ab);
        g.c[0] = gc.x; g.c[1] = gc.y; g.c[2] = gc.z; g.c[3] = gc.w;
    }
    st.cslot = st.cslot + 1 == SL_CR ? 0 : st.cslot + 1;
    SlRow x[K + 1];                  // x[j]: the newest row of sweep j's input; x[K]: the row this stage hands on
    sl_unpack(n01, n23, x[0].p);
    if (idle || (!FULL && __builtin_amdgcn_ballot_w64(!all_pass) == 0)) {
        // idle: the rows this wavefront's stages would make at this step are outside what the chunk's result depends on
        // (the pipeline is filling or draining); or no stage of this wavefront has a sweep to do (fewer sweeps asked
        // than the pipeline is deep): hand the rows on untouched
#pragma unroll
        for (int j = 0; j < K; ++j) x[j + 1] = st.r1[j];
    } else {
#pragma unroll
        for (int j = 0; j < K; ++j) x[j + 1] = sl_sweep<FULL>(st.r0[j], st.r1[j], x[j], j == 0 ? g : st.gp[j > 0 ? j - 1 : 0], pass[j], w12);    // row rA-j of sweep j
    }
    const SlRow &out = x[K];
    const int rB = st.rA - (K - 1);
    float4 a0u, a0v;
    if (ACC && LASTW) {
        a0u = st.au0; a0v = st.av0;
        st.au0 = st.au1; st.av0 = st.av1; st.au1 = st.au2; st.av1 = st.av2;
        st.au2 = *reinterpret_cast<const float4 *>(accU + lane_safe);
        st.av2 = *reinterpret_cast<const float4 *>(accV + lane_safe);
    }
    if (stage < SL_NS - 1) {
        ring[stage + 1][rB & 1][0][l] = make_float4(out.p[0].x, out.p[0].y, out.p[1].x, out.p[1].y);
        ring[stage + 1][rB & 1][1][l] = make_float4(out.p[2].x, out.p[2].y, out.p[3].x, out.p[3].y);
    } else {
        float4 ou = make_float4(out.p[0].x, out.p[1].x, out.p[2].x, out.p[3].x), ov = make_float4(out.p[0].y, out.p[1].y, out.p[2].y, out.p[3].y);
        if (ACC && LASTW) {
            ou = make_float4(__fadd_rn(a0u.x, ou.x), __fadd_rn(a0u.y, ou.y), __fadd_rn(a0u.z, ou.z), __fadd_rn(a0u.w, ou.w));
            ov = make_float4(__fadd_rn(a0v.x, ov.x), __fadd_rn(a0v.y, ov.y), __fadd_rn(a0v.z, ov.z), __fadd_rn(a0v.w, ov.w));
        }
        if (store_lane && row_ok) {
            // rowU, rowV: the output row of this step (the same for the whole workgroup, so its address is scalar work)
            *reinterpret_cast<float4 *>(rowU + lane_off) = ou;
            *reinterpret_cast<float4 *>(rowV + lane_off) = ov;
        }
    }
#pragma unroll
    for (int j = 0; j < K; ++j) { st.r0[j] = st.r1[j]; st.r1[j] = x[j]; }
#pragma unroll
    for (int j = K - 2; j > 0; --j) st.gp[j] = st.gp[j - 1];
    if (K > 1) st.gp[0] = g;
    ++st.rA;
}

// ---- loader -------------------------------------------------------------------------------------------
// One lane's four pixels of a row, 'mirror' border: inside the image one 16-byte load; a lane that
// straddles or lies beyond the left / right edge gathers the reflected columns one by one.
struct SlCols { int x, x0, x1, x2, x3; bool inside; };    // a lane's first column, its four reflected columns
template <bool EDGE>
__device__ __forceinline__ float4 sl_load4(const float *__restrict__ row, const SlCols &c) {
    if (!EDGE || c.inside) return *reinterpret_cast<const float4 *>(row + c.x);
    return make_float4(row[c.x0], row[c.x1], row[c.x2], row[c.x3]);
}

// The loader's loop: two sets of LN lanes (two half-wavefronts, or two wavefronts for the wide strips).  Set 0 carries
// U, A, C; set 1 carries V, B (each one plane after its partner, so the row address is a scalar base plus a per-lane
// offset).  Rows outside the image are the reflected rows.  EDGE: the strip reaches over the left or right edge of
// the image.  ZERO: the flow is zero on input and is not read.
template <int NL, int LN, int K, bool EDGE, bool ZERO>
__device__ __forceinline__ void sl_loader(const SlPlanes &in, float4 (&ring)[SlCfg<NL, LN, K>::NS][2][2][LN],
                                          float4 (&coef)[SlCfg<NL, LN, K>::CR][3][LN], int ltid, int x0, int rho0, int NS, int H, int W,
                                          const OfriPrio &pr) {
    constexpr int SL_CR = SlCfg<NL, LN, K>::CR;
    const int lj = ltid % LN, lset = ltid / LN;                        // ltid: thread index within the loader
    const int lx = x0 + 4 * lj;
    SlCols cols;                                                       // column reflection does not depend on the row
    cols.x = lx; cols.inside = lx >= 0 && lx + 3 < W;
    cols.x0 = sl_refl(lx, W); cols.x1 = sl_refl(lx + 1, W); cols.x2 = sl_refl(lx + 2, W); cols.x3 = sl_refl(lx + 3, W);
    const long long pl = lset ? (long long)(in.V - in.U) : 0;          // plane stride for the second set
    int lslot = (rho0 + 64 * SL_CR) % SL_CR;                           // slot of the row written next
    // rows in flight: named registers (an array here ends up in scratch memory, which serialises the prefetch)
#define SL_LOAD(U_, ROW_)                                                                          \
    {                                                                                              \
        const long long o = (long long)sl_refl((ROW_), H) * W;   /* scalar */                      \
        qa##U_ = ZERO ? make_float4(0.f, 0.f, 0.f, 0.f) : sl_load4<EDGE>(in.U + o + pl, cols);    \
        qb##U_ = sl_load4<EDGE>(in.fx + o + pl, cols);                                            \
        if (!lset) qc##U_ = sl_load4<EDGE>(in.ft + o, cols);                                      \
    }
    // the compiler must keep the first fill in program order (oldest load = first row consumed), or it has
    // to wait for every load in flight at the top of each round
#define SL_Q(U_) float4 qa##U_, qb##U_, qc##U_ = make_float4(0.f, 0.f, 0.f, 0.f); SL_LOAD(U_, rho0 + (U_)) asm volatile("" ::: "memory");
    SL_Q(0) SL_Q(1) SL_Q(2) SL_Q(3) SL_Q(4) SL_Q(5)
#undef SL_Q
    static_assert(SL_PF == 6, "the prefetch registers are spelled out for six rows");
#define SL_STEP(U_)                                                                \
    {                                                                              \
        const int rho = rho0 + t0 + (U_);                                          \
        /* interleaved rows: set 0 (U, A) fills the even floats of [0] and [1], set 1 (V, B) the odd ones */ \
        {                                                                          \
            float *r0 = reinterpret_cast<float *>(&ring[0][rho & 1][0][lj]) + lset, *r1 = reinterpret_cast<float *>(&ring[0][rho & 1][1][lj]) + lset; \
            float *c0 = reinterpret_cast<float *>(&coef[lslot][0][lj]) + lset, *c1 = reinterpret_cast<float *>(&coef[lslot][1][lj]) + lset; \
            r0[0] = qa##U_.x; r0[2] = qa##U_.y; r1[0] = qa##U_.z; r1[2] = qa##U_.w; \
            c0[0] = qb##U_.x; c0[2] = qb##U_.y; c1[0] = qb##U_.z; c1[2] = qb##U_.w; \
        }                                                                          \
        if (!lset) coef[lslot][2][lj] = qc##U_;                                    \
        lslot = lslot + 1 == SL_CR ? 0 : lslot + 1;                                \
        SL_LOAD(U_, rho + SL_PF)                                                   \
        __syncthreads();                                                           \
    }
    OfriPrioState ps = ofri_prio_begin(pr, NS, blockIdx.x);
    for (int t0 = 0; t0 < NS; t0 += SL_PF) {
        ofri_prio_turn(t0, pr, ps);
        SL_STEP(0) SL_STEP(1) SL_STEP(2) SL_STEP(3) SL_STEP(4) SL_STEP(5)
    }
#undef SL_STEP
#undef SL_LOAD
}

// FULL: all NL sweeps are wanted (no copy-through selects in the inner loop).  ACC: the result is added to what
// (Uo, Vo) hold.  All planes are dense with W columns (W % 4 == 0, 16-byte aligned); zero_in: the input flow is
// zero and in.U, in.V are not read.
template <int NL, bool FULL, bool ACC, int LN = 32, int K = 2>
__global__ void __launch_bounds__((SlCfg<NL, LN, K>::THREADS), (K == 2 ? (LN == 64 && NL == 12 ? 2 : 1) : (SlCfg<NL, LN, K>::PER_CU)))
k_hs_stream(SlPlanes in, float *__restrict__ Uo, float *__restrict__ Vo, int H, int W, int T, int tiles_x, int chunk_rows,
            int zero_in, OfriPrio pr) {
    using Cfg = SlCfg<NL, LN, K>;
    constexpr int SL_NS = Cfg::NS, SL_CR = Cfg::CR, SL_LAG = Cfg::LAG, SL_OX = Cfg::OX, SL_NL = NL;
    constexpr int SL_RX = Cfg::RX;
    static_assert(Cfg::LDS * Cfg::PER_CU <= 160 * 1024, "the workgroups per CU must fit the LDS");
    __shared__ float4 ring[SL_NS][2][2][LN];      // ring[s]: input rows of stage s.  [row & 1][U|V][lane]
    __shared__ float4 coef[SL_CR][3][LN];         // [row slot][A|B|C][lane]
    constexpr int SL_NW = Cfg::THREADS / 64, SL_NLW = 2 * LN / 64;
    static_assert(Cfg::THREADS % 64 == 0 && (SL_NS * LN) % 64 == 0, "whole wavefronts");
    __shared__ int wave_simd[SL_NW];
    // thread index by role: stages first, then the loader (sl_wave_role)
    const int tid = sl_wave_role<SL_NW, SL_NLW>(wave_simd, pr.ncu > 0 ? blockIdx.x / pr.ncu : 0) * 64 + (threadIdx.x & 63);
#ifdef OFRI_SL_TRACE
    if (threadIdx.x == 0 && blockIdx.x < SL_TRACE_MAX) {
        g_sl_trace[3 * blockIdx.x] = wall_clock64();
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_sl_trace[3 * blockIdx.x + 2] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
    // Workgroups go to the 8 XCDs round-robin by index, and each XCD has its own L2.  Renumber them so that an
    // XCD gets a run of neighbouring strips and chunks: the columns and fill rows that neighbours share are then
    // fetched from memory once per XCD instead of once per workgroup.
    const int nb = gridDim.x, per = nb / 8;
    const int bid = blockIdx.x < per * 8 ? (blockIdx.x % 8) * per + blockIdx.x / 8 : blockIdx.x;
    const int strip = bid % tiles_x, chunk = bid / tiles_x;
    const int ya = chunk * chunk_rows, yb = min(H, ya + chunk_rows);
    const int x0 = strip * SL_OX - SL_NL;          // image x of strip column 0
    const int rho0 = ya - SL_NL;                   // first input row of the chunk
    // steps until the last sweep has produced row yb-1, rounded up to whole rounds of the prefetch registers: the
    // loader's loop then has no branch inside a round, and the compiler can leave the younger loads in flight
    // (with a branch per step it waits for all of them).  The extra steps store nothing.
    const int NS = ceil_div_dev((yb - ya) + SL_NL + SL_LAG, SL_PF) * SL_PF;

    // The loader and the stages run separate loops (their registers then overlap instead of adding up);
    // both execute one barrier per step, NS in all.
    if (tid >= SL_NS * LN) {
        const bool edge = x0 < 0 || x0 + SL_RX > W;
        if (edge) {
            if (zero_in) sl_loader<NL, LN, K, true, true>(in, ring, coef, tid - SL_NS * LN, x0, rho0, NS, H, W, pr);
            else sl_loader<NL, LN, K, true, false>(in, ring, coef, tid - SL_NS * LN, x0, rho0, NS, H, W, pr);
        } else {
            if (zero_in) sl_loader<NL, LN, K, false, true>(in, ring, coef, tid - SL_NS * LN, x0, rho0, NS, H, W, pr);
            else sl_loader<NL, LN, K, false, false>(in, ring, coef, tid - SL_NS * LN, x0, rho0, NS, H, W, pr);
        }
    } else {
        // ---- stage half-wavefronts ---------------------------------------------------------------------
        const int stage = tid / LN, l = tid % LN;
        SlStage<K> st;
#pragma unroll
        for (int j = 0; j < K; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                st.r0[j].p[k] = st.r1[j].p[k] = (f2){0.f, 0.f};
                st.gp[j > 0 ? j - 1 : 0].ab[k] = (f2){0.f, 0.f};
                st.gp[j > 0 ? j - 1 : 0].c[k] = 0.f;
            }
        const f2 w12 = hs_w12_pair();
        st.au0 = st.av0 = st.au1 = st.av1 = st.au2 = st.av2 = make_float4(0.f, 0.f, 0.f, 0.f);
        st.rA = rho0 - 2 - (K + 1) * stage;
        st.cslot = (st.rA + 64 * SL_CR) % SL_CR;
        bool pass[K], all_pass = true;                                    // fewer than NL sweeps asked: copy rows through
#pragma unroll
        for (int j = 0; j < K; ++j) { pass[j] = K * stage + j >= T; all_pass = all_pass && pass[j]; }
        const bool store_lane = l >= SL_NL / 4 && l < LN - SL_NL / 4 && (x0 + 4 * l) < W;
        const int lane_off = x0 + 4 * l;
        const int lane_safe = min(max(lane_off, 0), W - 4);
        // the wavefront that holds the last stage (it alone touches the destination)
        const bool lastw = __builtin_amdgcn_readfirstlane(tid / 64) == (SL_NS * LN - 1) / 64;
        // Steps at which this wavefront's stages have nothing to contribute.  Sweep k of a chunk is needed on rows
        // ya - (NL - k) .. yb - 1 + (NL - k) only; stage s makes row rA - j = rho0 - 2 - (K + 1) s - j + t of sweep K s + j + 1 at
        // step t, so it is needed for (2 K + 1) s + 3 <= t <= (yb - ya) + 2 NL + s.  Outside that window (the
        // pipeline's fill and drain) the wavefront skips the arithmetic and leaves the issue slots to the other
        // workgroups of the CU; what it hands on then is never used.
        const int wv = __builtin_amdgcn_readfirstlane(tid / 64);
        const int w_lo = (2 * K + 1) * ((wv * 64) / LN) + 3, w_hi = (yb - ya) + 2 * SL_NL + (wv * 64 + 63) / LN;
        // six steps per round (NS is a multiple): the row registers rotate by renaming instead of moves
#define SL_STEP(U_, LASTW_)                                                                                           \
    {                                                                                                                 \
        const int ro = rho0 - SL_LAG + t + (U_);          /* row the last stage finishes at this step */              \
        const long long ra = (long long)min(max(ro + 3, ya), yb - 1) * W;      /* row whose old values are fetched */  \
        const bool idle = t + (U_) < w_lo || t + (U_) > w_hi;                                                         \
        sl_stage_step<NL, LN, K, FULL, ACC, LASTW_>(st, ring, coef, stage, l, pass, all_pass, store_lane, lane_off, Uo + (long long)ro * W, \
                                     Vo + (long long)ro * W, ro >= ya && ro < yb, Uo + ra, Vo + ra, lane_safe, idle, w12); \
        __syncthreads();                                                                                              \
    }
        OfriPrioState ps = ofri_prio_begin(pr, NS, blockIdx.x);
        if (ACC && lastw) {
            // rows ro, ro + 1, ro + 2 of the first step (clamped into the chunk: what is fetched for rows outside is not used)
            const int r0c = rho0 - SL_LAG;
            const long long q0 = (long long)min(max(r0c, ya), yb - 1) * W, q1 = (long long)min(max(r0c + 1, ya), yb - 1) * W,
                            q2 = (long long)min(max(r0c + 2, ya), yb - 1) * W;
            st.au0 = *reinterpret_cast<const float4 *>(Uo + q0 + lane_safe); st.av0 = *reinterpret_cast<const float4 *>(Vo + q0 + lane_safe);
            st.au1 = *reinterpret_cast<const float4 *>(Uo + q1 + lane_safe); st.av1 = *reinterpret_cast<const float4 *>(Vo + q1 + lane_safe);
            st.au2 = *reinterpret_cast<const float4 *>(Uo + q2 + lane_safe); st.av2 = *reinterpret_cast<const float4 *>(Vo + q2 + lane_safe);
            for (int t = 0; t < NS; t += SL_PF) {
                ofri_prio_turn(t, pr, ps);
                SL_STEP(0, true) SL_STEP(1, true) SL_STEP(2, true) SL_STEP(3, true) SL_STEP(4, true) SL_STEP(5, true)
            }
        } else {
            for (int t = 0; t < NS; t += SL_PF) {
                ofri_prio_turn(t, pr, ps);
                SL_STEP(0, false) SL_STEP(1, false) SL_STEP(2, false) SL_STEP(3, false) SL_STEP(4, false) SL_STEP(5, false)
            }
        }
#undef SL_STEP
#ifdef OFRI_SL_TRACE
        if (tid == 0 && blockIdx.x < SL_TRACE_MAX) g_sl_trace[3 * blockIdx.x + 1] = wall_clock64();
#endif
    }
}

// ---- coefficient planes -------------------------------------------------------------------------------
// Scaled coefficients A, B, C (hs_scaled).  FRAMES: differentiated on the spot from the two frames (ca = first,
// cb = second image of computeDerivatives); otherwise ca, cb, cc are fx, fy, ft (and may be the output planes).
// The inputs are dense; po is the row pitch of the output planes.
template <bool FRAMES>
__global__ void k_sl_coefs(const float *ca, const float *cb, const float *cc, float *oa, float *ob, float *oc,
                           int H, int W, int po, float a2) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= W || y >= H) return;
    const size_t s = (size_t)y * W + x, d = (size_t)y * po + x;
    float gx, gy, gt;
    if (FRAMES) hs_derivs_at(ca, cb, y, x, H, W, gx, gy, gt);
    else { gx = ca[s]; gy = cb[s]; gt = cc[s]; }
    hs_scaled(gx, gy, gt, a2, oa[d], ob[d], oc[d]);
}

int ofri_hs_stream_sweeps(void) { return SL_NL_MAIN; }
int ofri_hs_stream_deep_sweeps(void) { return SL_NL_DEEP; }
// worth it where the row chunks are long against the pipeline's fill (2048 x 2048 pixels and up)
bool ofri_hs_stream_pays(int H, int W) { return (long long)H * W >= 4000000LL && (W & 3) == 0; }

size_t ofri_sl_bytes(int H, int W) { return 7 * (size_t)H * W * sizeof(float); }
void ofri_sl_layout(float *base, int H, int W, OfriSlSet *s) {
    const size_t plane = (size_t)H * W;
    s->U[0] = base; s->V[0] = base + plane; s->U[1] = base + 2 * plane; s->V[1] = base + 3 * plane;
    s->fx = base + 4 * plane; s->fy = base + 5 * plane; s->ft = base + 6 * plane;
    s->P = W;
    s->zero_in = 0;
}
// U, V dense (both null: zero flow, nothing is copied); coefficients either dense (fx, fy, ft) or, with
// fx == nullptr, differentiated from the frames
int ofri_sl_fill(ofri_ctx *ctx, OfriSlSet &s, const float *U, const float *V, const float *fx, const float *fy,
                 const float *ft, const float *first, const float *second, float a2, int H, int W) {
    const size_t bytes = (size_t)H * W * sizeof(float);
    s.zero_in = U == nullptr;
    if (U) {
        OFRI_TRY(ofri_dev_copy(ctx, s.U[0], U, bytes));
        OFRI_TRY(ofri_dev_copy(ctx, s.V[0], V, bytes));
    }
    return ofri_hs_scaled_coefs(ctx, fx, fy, ft, first, second, s.fx, s.fy, s.ft, a2, H, W);
}
int ofri_hs_scaled_coefs(ofri_ctx *ctx, const float *fx, const float *fy, const float *ft, const float *first,
                         const float *second, float *A, float *B, float *C, float a2, int H, int W, int pitch_out) {
    if (pitch_out <= 0) pitch_out = W;
    dim3 b(64, 4), g(ceil_div(W, 64), ceil_div(H, 4));
    if (fx) k_sl_coefs<false><<<g, b, 0, ctx->stream>>>(fx, fy, ft, A, B, C, H, W, pitch_out, a2);
    else k_sl_coefs<true><<<g, b, 0, ctx->stream>>>(first, second, nullptr, A, B, C, H, W, pitch_out, a2);
    OFRI_LAUNCH_CHECK();
    return OFRI_OK;
}
template <int NL, int LN, int K = 2>
static void sl_launch(ofri_ctx *ctx, const SlPlanes &in, float *uo, float *vo, int H, int W, int T, int zero_in, bool acc) {
    using Cfg = SlCfg<NL, LN, K>;
    const int tiles_x = ceil_div(W, Cfg::OX);
    // As many row chunks as fill the chip once with the workgroups per CU that the LDS allows.  Every chunk pays
    // NL + LAG rows of fill, so the last workgroup per CU is used only where the chunks stay long (4096 x 4096:
    // 161 against 170 us per launch; 2048 x 2048: 71 against 66 us with it).
    int nchunks = Cfg::PER_CU * ofri_ctx_ncu(ctx) / tiles_x;
    if (nchunks < 1) nchunks = 1;
    if (ceil_div(H, nchunks) < Cfg::MIN_CHUNK && Cfg::PER_CU > 2) nchunks = max(1, (Cfg::PER_CU - 1) * ofri_ctx_ncu(ctx) / tiles_x);
    nchunks = max(1, nchunks / ctx->share);           // other contexts fill the rest of the GPU (ofri_set_share)
    const int chunk_rows = ceil_div(H, nchunks);
    nchunks = ceil_div(H, chunk_rows);
    const int grid = tiles_x * nchunks, thr = Cfg::THREADS;
    const int ncu = ofri_ctx_ncu(ctx), per_cu = ceil_div(grid, ncu);
    // measured at 4096 x 4096 and 2048 x 2048: four priority levels, 8 turns per chunk for the 12-sweep launches (154 -> 140 us
    // and 58 -> 52 us per launch), 6 for the deeper one (232 -> 232 and 76 -> 69 us)
    const OfriPrio pr{ncu, per_cu > 1 ? 4 : 0, NL == SL_NL_MAIN ? 8 : 6};
    if (T == NL) {
        if (acc) k_hs_stream<NL, true, true, LN, K><<<grid, thr, 0, ctx->stream>>>(in, uo, vo, H, W, T, tiles_x, chunk_rows, zero_in, pr);
        else k_hs_stream<NL, true, false, LN, K><<<grid, thr, 0, ctx->stream>>>(in, uo, vo, H, W, T, tiles_x, chunk_rows, zero_in, pr);
    } else {
        if (acc) k_hs_stream<NL, false, true, LN, K><<<grid, thr, 0, ctx->stream>>>(in, uo, vo, H, W, T, tiles_x, chunk_rows, zero_in, pr);
        else k_hs_stream<NL, false, false, LN, K><<<grid, thr, 0, ctx->stream>>>(in, uo, vo, H, W, T, tiles_x, chunk_rows, zero_in, pr);
    }
}
// strips of 256 columns where two workgroups per CU still get long row chunks
static bool sl_wide_pays(ofri_ctx *ctx, int H, int W) {
    if (ctx->hs_fuse == 32) return true;                    // asked for (ofri_hs_set_fuse)
    if (ctx->hs_fuse == 16) return false;
    const int tiles_x = ceil_div(W, SlCfg<SL_NL_MAIN, 64, SL_K_MAIN>::OX);
    const int nchunks = max(1, SlCfg<SL_NL_MAIN, 64, SL_K_MAIN>::PER_CU * ofri_ctx_ncu(ctx) / tiles_x);
    return ceil_div(H, nchunks) >= 128;
}

// n sweeps starting from the pair *cur of the set (the last launch shorter if n is not a multiple of 12).
// to_dense: the last launch stores into (Uout, Vout), adding to what they hold if accumulate; otherwise the result
// stays in the set and *cur names its pair.
int ofri_sl_sweeps(ofri_ctx *ctx, OfriSlSet &s, int n, bool to_dense, bool accumulate, float *Uout, float *Vout,
                   int H, int W, int *cur, int *launches, bool deep_tail) {
    // the loader addresses V, B as U, A plus one common plane stride
    OFRI_REQUIRE(s.V[0] - s.U[0] == s.V[1] - s.U[1] && s.fy - s.fx == s.V[0] - s.U[0] &&
                     s.V[0] > s.U[0] && (s.V[0] - s.U[0]) < (1LL << 29),
                 "plane set is not laid out as U V U V A B C");
    OFRI_REQUIRE((W & 3) == 0 && H >= 1 && W >= 4, "row-streaming pipeline needs a width that is a multiple of 4");
    OFRI_REQUIRE(to_dense || !accumulate, "only a dense destination can be accumulated into");
    int remaining = n, c = *cur, L = 0;
    while (remaining > 0) {
        // deep_tail: when 13 to 16 sweeps are left, one launch of the 16-deep pipeline takes them (4096 x 4096: 263 us
        // against 165 us for 12 sweeps plus a launch of 130 to 190 us for the rest)
        const bool deep = deep_tail && remaining > SL_NL_MAIN && remaining <= SL_NL_DEEP;
        const int T = deep ? remaining : (remaining < SL_NL_MAIN ? remaining : SL_NL_MAIN);
        const bool dense = remaining == T && to_dense;
        SlPlanes in{s.U[c], s.V[c], s.fx, s.fy, s.ft};
        float *uo = dense ? Uout : s.U[c ^ 1], *vo = dense ? Vout : s.V[c ^ 1];
        // the deep launch in strips of 256 columns (eight stage wavefronts and two loaders, 106 KB of LDS, one workgroup per
        // CU) where the 12-sweep launches take them too: 16 sweeps at 4096 x 4096 in 180 us against 225 us in narrow strips
        if (deep && sl_wide_pays(ctx, H, W)) sl_launch<SL_NL_DEEP, 64>(ctx, in, uo, vo, H, W, T, s.zero_in, dense && accumulate);
        else if (deep) sl_launch<SL_NL_DEEP, 32>(ctx, in, uo, vo, H, W, T, s.zero_in, dense && accumulate);
        else if (sl_wide_pays(ctx, H, W)) sl_launch<SL_NL_MAIN, 64, SL_K_MAIN>(ctx, in, uo, vo, H, W, T, s.zero_in, dense && accumulate);
        else sl_launch<SL_NL_MAIN, 32, SL_K_MAIN>(ctx, in, uo, vo, H, W, T, s.zero_in, dense && accumulate);
        OFRI_LAUNCH_CHECK();
        s.zero_in = 0;
        if (!dense) c ^= 1;
        remaining -= T;
        ++L;
    }
    *cur = c;
    if (launches) *launches = L;
    return OFRI_OK;
}

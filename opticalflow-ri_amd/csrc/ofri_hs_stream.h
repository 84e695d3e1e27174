// Plane set of the Horn-Schunck row-streaming pipeline (ofri_hs_stream.hip): seven dense planes in one
// allocation, in the order U0 V0 U1 V1 A B C (the loader relies on V, B lying one plane after U, A).
// A, B, C are the scaled coefficients of ofri_hs_math.h (hs_scaled).  P is the row pitch in floats (= columns).
#pragma once
#include "ofri_internal.h"

struct OfriSlSet {
    float *U[2], *V[2];          // ping-pong pairs
    float *fx, *fy, *ft;         // scaled coefficients A, B, C
    int P;
    int zero_in;                 // pair 0 stands for a zero flow and holds nothing yet
};

int ofri_hs_stream_sweeps(void);                       // sweeps per launch
int ofri_hs_stream_deep_sweeps(void);                  // sweeps of the deeper launch that can take a tail (ofri_sl_sweeps, deep_tail)
bool ofri_hs_stream_pays(int H, int W);                // field large enough for the pipeline to win
size_t ofri_sl_bytes(int H, int W);                    // device bytes of a plane set
void ofri_sl_layout(float *base, int H, int W, OfriSlSet *s);
int ofri_sl_fill(ofri_ctx *ctx, OfriSlSet &s, const float *U, const float *V, const float *fx, const float *fy,
                 const float *ft, const float *first, const float *second, float a2, int H, int W);
int ofri_sl_sweeps(ofri_ctx *ctx, OfriSlSet &s, int n, bool to_dense, bool accumulate, float *Uout, float *Vout,
                   int H, int W, int *cur, int *launches, bool deep_tail = false);
// Scaled coefficient planes (A, B, C) for the fast kernels outside a plane set: from dense fx, fy, ft, or
// (fx == nullptr) differentiated from the two frames (first, second of computeDerivatives).  A, B, C may be fx, fy, ft.
// pitch_out: row pitch of A, B, C in floats (0: dense).
int ofri_hs_scaled_coefs(ofri_ctx *ctx, const float *fx, const float *fy, const float *ft, const float *first,
                         const float *second, float *A, float *B, float *C, float a2, int H, int W, int pitch_out = 0);
// the same from a pyramid level's images: warp (if U, V are given), pre-filter, derivatives in one pass
bool ofri_sl_level_ok(int ntaps, int H, int W);
int ofri_sl_fill_level(ofri_ctx *ctx, OfriSlSet &s, const float *first, const float *second, const float *U, const float *V,
                       const float *taps_host, int ntaps, float a2, int H, int W);

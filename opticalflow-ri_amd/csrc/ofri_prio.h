// Rotating issue priority for kernels whose workgroups walk long chunks in step with one barrier per row
// (the Horn-Schunck and Liu-Shen row-streaming pipelines).
#pragma once
#include <hip/hip_runtime.h>

// Issue priority of the calling wavefront (s_setprio takes an immediate).
__device__ __forceinline__ void ofri_setprio(int p) {
    switch (p & 3) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
    }
}
// Wavefronts are issued oldest first: of the workgroups that share a CU the one dispatched first runs ahead (it finishes
// its chunk in half the time of the last one), and all of them reach their barriers and LDS phases in step.  Rotating the
// issue priority over them (start value = dispatch round of the workgroup, blockIdx / CUs; `phases` turns while a
// workgroup walks its chunk; `mod` levels, 0 = leave the priority alone) evens their progress out and staggers them.
struct OfriPrio { int ncu, mod, phases; };
struct OfriPrioState { int phase, next, period; };
__device__ __forceinline__ OfriPrioState ofri_prio_begin(const OfriPrio &pr, int NS, int block) {
    OfriPrioState ps;
    ps.phase = pr.ncu > 0 ? block / pr.ncu : 0;
    ps.period = pr.phases > 0 ? max(1, NS / pr.phases) : NS + 1;
    ps.next = 0;
    return ps;
}
__device__ __forceinline__ void ofri_prio_turn(int t, const OfriPrio &pr, OfriPrioState &ps) {
    if (pr.mod > 0 && t >= ps.next) {
        ofri_setprio(ps.phase % pr.mod);
        ++ps.phase;
        ps.next += ps.period;
    }
}

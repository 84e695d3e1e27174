#!/usr/bin/env python3
"""Per-path timings of the non-headline configurations (Farneback, Liu-Shen, dense LK and their pyramids) on one
GPU, with HIP events on the context's stream.  Complements bench.py (which measures the headline config).

    python bench_paths.py [--size 2048] [--reps 3] [--cfg3-size 4096]
Prints one JSON line per path: {"path", "size", "ms", "mpix_s", "alg_bytes_per_px", "gbs", "frac_hbm"}.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "opticalflow-ri_amd")]
HBM = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=2048)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--lk-size", type=int, default=1024)
    ap.add_argument("--cfg3-size", type=int, default=4096, help="also time the 2-level Farneback pyramid at this size (0: skip)")
    a = ap.parse_args()
    import _ofri
    from piv_synth import synth_piv
    from HornSchunck import HSOpticalFlowAlgoAdapter
    from PhysicsBasedOpticalFlowLiuShen import LiuShenOpticalFlowAlgoAdapter, liushen_device
    from Farneback_PyCL import Farneback_PyCL
    from denseLucasKanade_PyCL import denseLucasKanade_PyCl
    from GenericPyramidalOpticalFlow import pyramidal_flow_device
    ctx = _ofri.context()

    def timed(fn, reps):
        fn(); ctx.sync()
        e0, e1 = ctx.event(), ctx.event()
        ctx.record(e0)
        for _ in range(reps):
            fn()
        ctx.record(e1)
        return ctx.elapsed_ms(e0, e1) / reps

    def report(path, n, ms, bpp=None):
        d = {"path": path, "size": n, "ms": round(ms, 4), "mpix_s": round(n * n / ms / 1e3, 1)}
        if bpp:
            gbs = bpp * n * n / (ms * 1e-3) / 1e9
            d.update({"alg_bytes_per_px": bpp, "gbs": round(gbs, 1), "frac_hbm": round(gbs / HBM, 3)})
        print(json.dumps(d), flush=True)

    n = a.size
    f0, f1 = synth_piv(n, n, seed=0)
    im0, im1 = ctx.upload(f0.astype(np.float32)), ctx.upload(f1.astype(np.float32))
    z = ctx.zeros((n, n))
    # Liu-Shen: 60 sweeps at 48 B/px/sweep + 40 B/px set-up
    ms = timed(lambda: liushen_device(im0, im1, 10, z, z, False), a.reps)
    report("liushen_60_sweeps", n, ms, 48 * 60 + 40)
    # Farneback single level (defaults 33/5/7/1.5): 5 x (window filter+solve 28 B/px, matrix update 68 B/px) + 2 poly-exp + pre-blur
    fb = Farneback_PyCL()
    ms = timed(lambda: fb.compute_device(im0, im1, z, z), a.reps)
    report("farneback_compute_1_level", n, ms, 5 * 28 + 5 * 68 + 2 * 24 + 2 * 8 + 2 * 8)
    ms = timed(lambda: pyramidal_flow_device(im0, im1, 0.0, Farneback_PyCL(), 2, 1, None), a.reps)
    report("cfg3_farneback_2_level_pyramid", n, ms)
    if a.cfg3_size and a.cfg3_size != n:      # the size the Farneback configuration is named on
        m = a.cfg3_size
        g0, g1 = synth_piv(m, m, seed=0)
        k0, k1 = ctx.upload(g0.astype(np.float32)), ctx.upload(g1.astype(np.float32))
        ms = timed(lambda: pyramidal_flow_device(k0, k1, 0.0, Farneback_PyCL(), 2, 1, None), a.reps)
        report("cfg3_farneback_2_level_pyramid", m, ms)
        # the same pair replayed from a captured graph (piv_plan.FlowPlan: the path for streams of equally sized pairs)
        from piv_plan import FlowPlan
        plan = FlowPlan((m, m), 0.0, lambda: Farneback_PyCL(), pyramidalLevels=2, ctx=ctx)
        plan.load(k0, k1)
        ms = timed(plan.run, a.reps)
        report("cfg3_farneback_2_level_pyramid_captured_graph", m, ms)
        plan.close()
        del k0, k1, plan
    ms = timed(lambda: pyramidal_flow_device(im0, im1, 0.0, Farneback_PyCL(), 2, 1, 0.48, LiuShenOpticalFlowAlgoAdapter(10)), a.reps)
    report("cfg4_farneback_plus_liushen_2_level", n, ms)
    ms = timed(lambda: pyramidal_flow_device(im0, im1, 3.4, HSOpticalFlowAlgoAdapter([21, 45], 100), 2, 1, 0.48,
                                             LiuShenOpticalFlowAlgoAdapter(5)), a.reps)
    report("cfg5_hs_plus_liushen_2_level_per_pair", n, ms)
    # dense LK (27x27 window, 5 Newton steps): VALU / L1 bound, no HBM fraction quoted
    m = a.lk_size
    g0, g1 = synth_piv(m, m, seed=1)
    j0, j1 = ctx.upload(g0.astype(np.float32)), ctx.upload(g1.astype(np.float32))
    zz = ctx.zeros((m, m))
    lk = denseLucasKanade_PyCl()
    ms = timed(lambda: lk.compute_device(j0, j1, zz, zz), max(1, a.reps // 2))
    report("dense_lk_27x27_5_iters", m, ms)


if __name__ == "__main__":
    main()

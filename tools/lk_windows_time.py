#!/usr/bin/env python3
"""Dense Lucas-Kanade: the tile-of-pixels kernel against the per-wavefront kernel for the windows it is built for
(13, 27, 31 samples) with and without the asymmetry flags; 5 iterations on a synthetic pair.
    python tools/lk_windows_time.py [size]      (default 1024)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "opticalflow-ri_amd")]
import _ofri  # noqa: E402
from piv_synth import synth_piv  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = _ofri.context()
lib = _ofri.lib()
f0, f1 = synth_piv(n, n, seed=0)
a, b = ctx.upload(f0.astype(np.float32)), ctx.upload(f1.astype(np.float32))
z = ctx.zeros((n, n))
U, V = ctx.empty((n, n)), ctx.empty((n, n))
for hw in (6, 13, 15):
    for asym in ((0, 0, 0, 0), (1, 0, 0, 1)):
        out = {}
        res = {}
        for mode in (-1, 0):
            ctx.set_lk_mode(mode)

            def run():
                _ofri.check(lib.ofri_lk_compute(ctx.handle, a.ptr, b.ptr, z.ptr, z.ptr, U.ptr, V.ptr, n, n, 5, hw, *asym))
            run(); ctx.sync()
            e0, e1 = ctx.event(), ctx.event()
            ctx.record(e0)
            for _ in range(3):
                run()
            ctx.record(e1)
            out[mode] = ctx.elapsed_ms(e0, e1) / 3
            res[mode] = (U.numpy().copy(), V.numpy().copy())
        ctx.set_lk_mode(0)
        d = np.sqrt(((res[0][0] - res[-1][0]) ** 2 + (res[0][1] - res[-1][1]) ** 2).sum() / ((res[-1][0] ** 2 + res[-1][1] ** 2).sum() + 1e-30))
        print(json.dumps({"size": n, "window": 2 * hw + 1, "asym": list(asym), "per_wavefront_ms": round(out[-1], 3), "tile_ms": round(out[0], 3),
                          "speedup": round(out[-1] / out[0], 2), "tile_Mpix_s": round(n * n / out[0] / 1e3, 1),
                          "rel_l2_tile_vs_per_wavefront": float(d), "largest_difference_px": float(max(np.abs(res[0][0] - res[-1][0]).max(), np.abs(res[0][1] - res[-1][1]).max()))}), flush=True)

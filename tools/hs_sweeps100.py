#!/usr/bin/env python3
"""A level's Horn-Schunck sweep loop (96 and 100 sweeps) under every launch shape."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "opticalflow-ri_amd")]
import _ofri
if len(sys.argv) > 1:
    _ofri.LIB_PATH = os.path.join(ROOT, 'opticalflow-ri_amd', sys.argv[1])
SIZES = tuple(int(a) for a in sys.argv[2:]) or (4096, 2048, 1024)
ctx = _ofri.Context(0); lib = _ofri.lib()
for n in SIZES:
    rng = np.random.default_rng(0)
    pl = [ctx.upload(rng.standard_normal((n, n)).astype(np.float32)) for _ in range(5)]
    for fuse in (16, 0, 32, 0, 32, 16):
        for T in (96, 100):
            ctx.set_hs_fuse(fuse)
            def run():
                _ofri.check(lib.ofri_hs_iterate(ctx.handle, pl[0].ptr, pl[1].ptr, pl[2].ptr, pl[3].ptr, pl[4].ptr, 10.0, T, n, n))
            try:
                for _ in range(6): run()
            except Exception as e:
                print(n, fuse, T, "error", e); continue
            ctx.sync(); ctx.prof_enable(True)
            for _ in range(5): run()
            ctx.sync(); recs = ctx.prof_records(); ctx.prof_enable(False)
            print("%d^2 fuse %2d T=%3d: %.1f us per call (%d launches)" % (n, fuse, T, sum(r[0] for r in recs) / 5 * 1e3, sum(r[1] for r in recs) // 5), flush=True)

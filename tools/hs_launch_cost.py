#!/usr/bin/env python3
"""Cost of one Horn-Schunck launch against the number of sweeps it carries, pipeline against tile kernel."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "opticalflow-ri_amd")]
import _ofri

if len(sys.argv) > 1:
    _ofri.LIB_PATH = os.path.join(ROOT, 'opticalflow-ri_amd', sys.argv[1])
ctx = _ofri.Context(0)
lib = _ofri.lib()
for n in (4096, 2048):
    rng = np.random.default_rng(0)
    pl = [ctx.upload(rng.standard_normal((n, n)).astype(np.float32)) for _ in range(5)]
    for fuse in (16, 8):
        for T in (2, 4, 6, 8, 10, 12):
            if fuse == 8 and T > 8:
                continue
            ctx.set_hs_fuse(fuse)
            def run():
                _ofri.check(lib.ofri_hs_iterate(ctx.handle, pl[0].ptr, pl[1].ptr, pl[2].ptr, pl[3].ptr, pl[4].ptr, 10.0, T, n, n))
            try:
                for _ in range(2):
                    run()
            except Exception as e:
                print(n, fuse, T, "error", e); continue
            ctx.sync()
            ctx.prof_enable(True)
            for _ in range(10):
                run()
            ctx.sync()
            recs = ctx.prof_records()
            ctx.prof_enable(False)
            ms = sum(r[0] for r in recs) / 10
            print("%d^2 %s T=%2d: %.1f us per call (%d launches/call)" % (n, "pipeline" if fuse == 16 else "tile    ", T, ms * 1e3, sum(r[1] for r in recs) // 10), flush=True)

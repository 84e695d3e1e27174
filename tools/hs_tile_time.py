#!/usr/bin/env python3
"""100 Horn-Schunck sweeps through the tile kernel (8 per launch) on fields below the pipeline's threshold."""
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
for n in (512, 1024, 1536, 2048):
    rng = np.random.default_rng(0)
    pl = [ctx.upload(rng.standard_normal((n, n)).astype(np.float32)) for _ in range(5)]
    ctx.set_hs_fuse(8)
    def run():
        _ofri.check(lib.ofri_hs_iterate(ctx.handle, pl[0].ptr, pl[1].ptr, pl[2].ptr, pl[3].ptr, pl[4].ptr, 10.0, 100, n, n))
    for _ in range(3):
        run()
    ctx.sync()
    for rep in range(2):
        ctx.prof_enable(True)
        for _ in range(10):
            run()
        ctx.sync()
        recs = ctx.prof_records()
        ctx.prof_enable(False)
        print("%d^2: %.1f us per 100 sweeps (%d launches)" % (n, sum(r[0] for r in recs) / 10 * 1e3, sum(r[1] for r in recs) // 10), flush=True)

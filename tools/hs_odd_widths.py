#!/usr/bin/env python3
"""Horn-Schunck sweep loop on widths that are not multiples of 4 (tile kernel on padded copies) against the nearest
aligned width and against one sweep per launch."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "opticalflow-ri_amd")]
import _ofri

ctx = _ofri.Context(0)
lib = _ofri.lib()
for shape in ((201, 313), (200, 312), (1023, 1023), (1024, 1024), (2047, 2047), (2048, 2048), (4095, 4095), (4096, 4096)):
    rng = np.random.default_rng(0)
    pl = [ctx.upload(rng.standard_normal(shape).astype(np.float32)) for _ in range(5)]
    for fuse in (0, 1):
        ctx.set_hs_fuse(fuse)
        run = lambda: _ofri.check(lib.ofri_hs_iterate(ctx.handle, pl[0].ptr, pl[1].ptr, pl[2].ptr, pl[3].ptr, pl[4].ptr, 10.0, 100, shape[0], shape[1]))
        run(); ctx.sync()
        e0, e1 = ctx.event(), ctx.event()
        ctx.record(e0)
        for _ in range(3):
            run()
        ctx.record(e1)
        ctx.sync()
        print("%5d x %5d  %s: %.3f ms per 100 sweeps" % (shape[0], shape[1], "automatic        " if fuse == 0 else "one sweep/launch ", ctx.elapsed_ms(e0, e1) / 3), flush=True)
    ctx.set_hs_fuse(0)

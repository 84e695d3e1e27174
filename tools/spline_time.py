#!/usr/bin/env python3
"""Development timing of the spline upsampling, 2048^2 -> 4096^2; optional argument: library file to load."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "opticalflow-ri_amd")]
import _ofri

if len(sys.argv) > 1:
    _ofri.LIB_PATH = os.path.join(ROOT, "opticalflow-ri_amd", sys.argv[1])
ctx = _ofri.Context(0)
lib = _ofri.lib()
n = 2048
F = (np.random.default_rng(1).standard_normal((n, n)).cumsum(0).cumsum(1) * 0.01).astype(np.float32)
d = ctx.upload(F)
o = ctx.empty((2 * n, 2 * n))
for _ in range(3):
    _ofri.check(lib.ofri_spline_upsample(ctx.handle, d.ptr, n, n, o.ptr, 2 * n, 2 * n, 2.0))
ctx.sync()
e0, e1 = ctx.event(), ctx.event()
ctx.record(e0)
for _ in range(20):
    _ofri.check(lib.ofri_spline_upsample(ctx.handle, d.ptr, n, n, o.ptr, 2 * n, 2 * n, 2.0))
ctx.record(e1)
ctx.sync()
print("%s: %.1f us per 2048^2 -> 4096^2 field" % (sys.argv[1] if len(sys.argv) > 1 else "default", ctx.elapsed_ms(e0, e1) / 20 * 1e3), flush=True)

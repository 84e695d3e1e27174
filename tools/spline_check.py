#!/usr/bin/env python3
"""Development check: the spline upsampling against SciPy (small sizes) and timing at 2048^2 -> 4096^2."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "opticalflow-ri_amd")]
import _ofri

ctx = _ofri.Context(0)
lib = _ofri.lib()
_vp, _i, _f = C.c_void_p, C.c_int, C.c_float
old = getattr(lib, "ofri_spline_upsample_twopass", None)
if old is not None:
    old.argtypes = [_vp, _vp, _i, _i, _vp, _i, _i, _f]; old.restype = _i


def run(fn, F, shape_out, scale=1.0):
    d = ctx.upload(F)
    o = ctx.empty(shape_out)
    _ofri.check(fn(ctx.handle, d.ptr, F.shape[0], F.shape[1], o.ptr, shape_out[0], shape_out[1], scale))
    return o.numpy()


from scipy.interpolate import RectBivariateSpline
cases = [((4, 4), (8, 8)), ((5, 7), (10, 14)), ((33, 34), (66, 68)), ((35, 130), (70, 260)), ((64, 128), (128, 256)),
         ((100, 156), (201, 313)), ((129, 257), (258, 514)), ((40, 50), (200, 333)), ((300, 200), (150, 100)),
         ((66, 255), (131, 509)), ((512, 512), (1024, 1024))]
for si, so in cases:
    rng = np.random.default_rng(si[0] * 100 + si[1])
    F = (rng.standard_normal(si).cumsum(0).cumsum(1) * 0.1).astype(np.float32)
    sp = RectBivariateSpline(np.arange(si[0]) / si[0], np.arange(si[1]) / si[1], F.astype(np.float64), kx=3, ky=3, s=0)
    want = sp(np.arange(so[0]) / so[0], np.arange(so[1]) / so[1]).astype(np.float32)
    got = run(lib.ofri_spline_upsample, F, so)
    e_new = np.abs(got - want).max() / max(1.0, np.abs(want).max())
    line = "%-12s -> %-12s new vs scipy %.2e" % (si, so, e_new)
    if old is not None:
        g2 = run(old, F, so)
        line += "  old vs scipy %.2e  new==old bits %.4f%%" % (np.abs(g2 - want).max() / max(1.0, np.abs(want).max()),
                                                             100.0 * np.mean(got.view(np.uint32) == g2.view(np.uint32)))
    print(line, flush=True)

n = 2048
F = (np.random.default_rng(1).standard_normal((n, n)).cumsum(0).cumsum(1) * 0.01).astype(np.float32)
d = ctx.upload(F)
o = ctx.empty((2 * n, 2 * n))
for name, fn in (("new", lib.ofri_spline_upsample), ("old", old)):
    if fn is None:
        continue
    for _ in range(3):
        _ofri.check(fn(ctx.handle, d.ptr, n, n, o.ptr, 2 * n, 2 * n, 2.0))
    ctx.sync()
    e0, e1 = ctx.event(), ctx.event()
    ctx.record(e0)
    for _ in range(20):
        _ofri.check(fn(ctx.handle, d.ptr, n, n, o.ptr, 2 * n, 2 * n, 2.0))
    ctx.record(e1)
    ctx.sync()
    print("%s: %.1f us per 2048^2 -> 4096^2 field" % (name, ctx.elapsed_ms(e0, e1) / 20 * 1e3), flush=True)
    res = o.numpy()
    if name == "new":
        keep = res
    else:
        print("new vs old at 4096^2: max abs diff %.3e, identical bits %.4f%%" % (np.abs(keep - res).max(), 100 * np.mean(keep.view(np.uint32) == res.view(np.uint32))))

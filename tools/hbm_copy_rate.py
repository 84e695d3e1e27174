#!/usr/bin/env python3
"""Device-to-device copy rate on the box (read + write bytes per second): the practical HBM ceiling that the
bandwidth-bound kernels are held against, next to the 8 TB/s of the data sheet."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "opticalflow-ri_amd")]
import _ofri

ctx = _ofri.Context(0)
lib = _ofri.lib()
for mb in (64, 256, 1024, 4096):
    n = mb << 20
    a, b = ctx.empty((n // 4,)), ctx.empty((n // 4,))
    _ofri.check(lib.ofri_zero(ctx.handle, a.ptr, n))
    _ofri.check(lib.ofri_copy(ctx.handle, b.ptr, a.ptr, n))
    ctx.sync()
    e0, e1 = ctx.event(), ctx.event()
    reps = 20
    ctx.record(e0)
    for _ in range(reps):
        _ofri.check(lib.ofri_copy(ctx.handle, b.ptr, a.ptr, n))
    ctx.record(e1)
    ctx.sync()
    ms = ctx.elapsed_ms(e0, e1) / reps
    print("copy of %5d MiB: %.3f ms, %.2f TB/s (read + write)" % (mb, ms, 2 * n / ms / 1e9), flush=True)

#!/usr/bin/env python3
"""Development timing of the Pillow-compatible bicubic resize, 4096^2 -> 2048^2, and a check against Pillow."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "opticalflow-ri_amd")]
import _ofri

ctx = _ofri.Context(0)
lib = _ofri.lib()
from PIL import Image
for (h, w), (ho, wo) in (((512, 768), (256, 384)), ((301, 447), (150, 223)), ((1000, 600), (333, 200))):
    F = (np.random.default_rng(h).random((h, w)) * 255).astype(np.float32)
    want = np.asarray(Image.fromarray(F, mode="F").resize((wo, ho), Image.BICUBIC))
    d, o = ctx.upload(F), ctx.empty((ho, wo))
    _ofri.check(lib.ofri_resize_bicubic(ctx.handle, d.ptr, h, w, o.ptr, ho, wo))
    print((h, w), "->", (ho, wo), "max abs diff vs Pillow %.3e" % np.abs(o.numpy() - want).max(), flush=True)
n = 4096
d = ctx.upload((np.random.default_rng(1).random((n, n)) * 255).astype(np.float32))
o = ctx.empty((n // 2, n // 2))
for _ in range(3):
    _ofri.check(lib.ofri_resize_bicubic(ctx.handle, d.ptr, n, n, o.ptr, n // 2, n // 2))
ctx.sync()
e0, e1 = ctx.event(), ctx.event()
ctx.record(e0)
for _ in range(20):
    _ofri.check(lib.ofri_resize_bicubic(ctx.handle, d.ptr, n, n, o.ptr, n // 2, n // 2))
ctx.record(e1)
ctx.sync()
print("resize 4096^2 -> 2048^2: %.1f us" % (ctx.elapsed_ms(e0, e1) / 20 * 1e3), flush=True)

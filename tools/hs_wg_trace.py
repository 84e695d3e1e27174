#!/usr/bin/env python3
"""Start, end and placement of every workgroup of one Horn-Schunck pipeline launch (needs a library built with
-DOFRI_SL_TRACE, path relative to the package as argv[1]): how evenly the chip is loaded during a launch."""
import ctypes
import os
import sys
from collections import Counter

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "opticalflow-ri_amd")]
import _ofri

_ofri.LIB_PATH = os.path.join(ROOT, 'opticalflow-ri_amd', sys.argv[1] if len(sys.argv) > 1 else 'libofri_trace.so')
ctx = _ofri.Context(0)
lib = _ofri.lib()
raw = ctypes.CDLL(_ofri.LIB_PATH)
raw.ofri_debug_sl_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
raw.ofri_debug_sl_trace_clear.argtypes = []
for n, fuse in ((4096, 16), (4096, 32), (2048, 16)):
    rng = np.random.default_rng(0)
    pl = [ctx.upload(rng.standard_normal((n, n)).astype(np.float32)) for _ in range(5)]
    ctx.set_hs_fuse(fuse)
    for k in range(4):
        if k == 3:
            ctx.sync(); raw.ofri_debug_sl_trace_clear()
        _ofri.check(lib.ofri_hs_iterate(ctx.handle, pl[0].ptr, pl[1].ptr, pl[2].ptr, pl[3].ptr, pl[4].ptr, 10.0, 12, n, n))
    ctx.sync()
    buf = np.zeros((4096, 3), dtype=np.uint64)
    assert raw.ofri_debug_sl_trace(buf.ctypes.data, 4096) == 0
    live = buf[:, 1] > 0
    t = buf[live]
    nwg = len(t)
    t0 = t[:, 0].min()
    start = (t[:, 0] - t0).astype(np.float64) / 100.0          # wall_clock64 ticks at 100 MHz -> us
    end = (t[:, 1] - t0).astype(np.float64) / 100.0
    dur = end - start
    hw = (t[:, 2] & np.uint64(0xffffffff)).astype(np.int64); xcc = (t[:, 2] >> np.uint64(32)).astype(np.int64)
    cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
    per_cu = Counter(cu.tolist())
    print("%d^2 fuse %d: %d workgroups on %d CUs (per CU: %s)" % (n, fuse, nwg, len(per_cu), dict(Counter(per_cu.values()))))
    print("  start: last %.1f us; end: first %.1f, median %.1f, last %.1f us; duration min %.1f median %.1f max %.1f us" %
          (start.max(), end.min(), np.median(end), end.max(), dur.min(), np.median(dur), dur.max()))
    print("  mean workgroup residency / launch span: %.2f" % (dur.sum() / nwg / end.max()))
    for x in range(8):
        m = xcc == x
        if m.any():
            print("  XCD %d: %4d workgroups, median duration %.1f, last end %.1f" % (x, m.sum(), np.median(dur[m]), end[m].max()))
    bidx = np.nonzero(live)[0]
    for lo in range(0, nwg, 128):
        m = (bidx >= lo) & (bidx < lo + 128)
        print("  workgroups %4d..%4d: end median %.1f (min %.1f max %.1f)" % (lo, lo + 127, np.median(end[m]), end[m].min(), end[m].max()))
    ranks = []
    for c in per_cu:
        m = np.nonzero(cu == c)[0]
        order = m[np.argsort(bidx[m])]
        ranks.append(tuple(np.argsort(np.argsort(end[order])).tolist()))
    print("  finishing order of a CU's workgroups listed by ascending index:", Counter(ranks).most_common(6))
    ox = 232 if fuse == 32 else 104
    tiles_x = -(-n // ox); per = nwg // 8
    bid = np.where(bidx < per * 8, (bidx % 8) * per + bidx // 8, bidx)
    strip = bid % tiles_x; chunk = bid // tiles_x
    late = np.argsort(-end)[:24]
    print("  latest workgroups (index, strip, chunk, end, co-residents on its CU):",
          [(int(bidx[i]), int(strip[i]), int(chunk[i]), round(float(end[i]), 1), per_cu[int(cu[i])]) for i in late])
    print("  rounds (index / 256) of the workgroups sharing a CU with the 8 latest:",
          [sorted((bidx[cu == cu[i]] >> 8).tolist()) for i in late[:8]])
    print("  CUs by the rounds they hold:", Counter(tuple(sorted((bidx[cu == c] >> 8).tolist())) for c in per_cu).most_common(8))
    for name, m in (("edge strips", (strip == 0) | (strip == tiles_x - 1)), ("inner strips", (strip > 0) & (strip < tiles_x - 1)),
                    ("first/last chunk", (chunk == 0) | (chunk == chunk.max())), ("last round", bidx >= 768 if nwg > 768 else bidx >= 256)):
        print("  %s: %d workgroups, duration median %.1f max %.1f" % (name, m.sum(), np.median(dur[m]), dur[m].max()))
    hist, edges = np.histogram(end, bins=8)
    print("  end-time histogram:", list(zip(np.round(edges[:-1], 0).tolist(), hist.tolist())))

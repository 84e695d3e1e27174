#!/usr/bin/env python3
"""Time of the Horn-Schunck pipeline launches (12 sweeps each) for one or more builds of the library:
    python tools/hs_stream_time.py [libofri_hip.so libofri_x.so ...]
Each library is measured in a process of its own (a process maps one build)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "opticalflow-ri_amd")]


def one(libname):
    import numpy as np
    import _ofri
    _ofri.LIB_PATH = os.path.join(ROOT, "opticalflow-ri_amd", libname)
    ctx = _ofri.Context(0)
    lib = _ofri.lib()
    out = []
    for n, sweeps in [tuple(map(int, x.split(':'))) for x in os.environ.get('HS_TIME_CASES', '4096:96,4096:16,4096:100,2048:100').split(',')]:
        rng = np.random.default_rng(0)
        pl = [ctx.upload(rng.standard_normal((n, n)).astype(np.float32)) for _ in range(5)]

        def run():
            _ofri.check(lib.ofri_hs_iterate(ctx.handle, pl[0].ptr, pl[1].ptr, pl[2].ptr, pl[3].ptr, pl[4].ptr, 10.0, sweeps, n, n))
        import time
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.6:       # the GPU raises its clocks over some hundred milliseconds of work
            for _ in range(4):
                run()
            ctx.sync()
        ctx.prof_enable(True)
        reps = 10
        for _ in range(reps):
            run()
        ctx.sync()
        recs = ctx.prof_records()
        ctx.prof_enable(False)
        ms = sum(r[0] for r in recs) / reps
        nl = sum(r[1] for r in recs) // reps
        out.append("%d^2 %3d sweeps: %7.1f us (%d launches, %.1f us each)" % (n, sweeps, ms * 1e3, nl, ms * 1e3 / nl))
        del pl
    print("%-28s %s" % (libname, " | ".join(out)), flush=True)


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--one":
        one(sys.argv[2])
    else:
        for name in (sys.argv[1:] or ["libofri_hip.so"]):
            subprocess.run([sys.executable, os.path.abspath(__file__), "--one", name], timeout=120)

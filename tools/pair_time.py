#!/usr/bin/env python3
"""Time of the headline frame pair (Horn-Schunck, 2 levels, 4096 x 4096, 100 sweeps per level) for one or more builds
of the library, replayed from a captured graph:
    python tools/pair_time.py [--size N] [libofri_hip.so libofri_x.so ...]
Each library is measured in a process of its own (a process maps one build).  With --one <lib> under rocprofv3 the
kernel trace of the same replays gives the per-kernel times."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "opticalflow-ri_amd")]


def one(libname, n):
    import numpy as np
    import _ofri
    _ofri.LIB_PATH = os.path.join(ROOT, "opticalflow-ri_amd", libname)
    from HornSchunck import HSOpticalFlowAlgoAdapter
    from piv_plan import FlowPlan
    from piv_synth import synth_piv
    ctx = _ofri.Context(0)
    a, b = synth_piv(n, n, 0)
    plan = FlowPlan((n, n), 3.4, lambda: HSOpticalFlowAlgoAdapter([21, 45], 100), pyramidalLevels=2, ctx=ctx, timed=bool(os.environ.get('OFRI_PAIR_TIMED')))
    plan.load(a, b)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.8:           # the GPU raises its clocks over some hundred milliseconds of work
        for _ in range(4):
            plan.run()
        ctx.sync()
    ts = []
    for _ in range(5):
        reps = 20
        ctx.sync()
        t = time.perf_counter()
        for _ in range(reps):
            plan.run()
        ctx.sync()
        ts.append((time.perf_counter() - t) / reps * 1e3)
    if os.environ.get('OFRI_PAIR_TIMED'):
        print("timed sections:", [(round(r[0] * 1e3, 1), r[1], r[2], r[3]) for r in plan.prof_records()])
    print("%-28s %d^2 pair: min %.4f ms, median %.4f ms  (%s)" % (libname, n, min(ts), sorted(ts)[2], " ".join("%.4f" % x for x in ts)), flush=True)


if __name__ == "__main__":
    argv = sys.argv[1:]
    n = 4096
    if argv[:1] == ["--size"]:
        n = int(argv[1]); argv = argv[2:]
    if argv[:1] == ["--one"]:
        one(argv[1], n)
    else:
        for name in (argv or ["libofri_hip.so"]):
            subprocess.run([sys.executable, os.path.abspath(__file__), "--size", str(n), "--one", name], timeout=200)

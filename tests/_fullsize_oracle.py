"""Oracle legs of tests/test_gpu_fullsize.py, run in CPU-only worker processes side by side (they never touch the
GPU): each returns the oracle's flow field of one BASELINE configuration on a seeded synthetic pair."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "opticalflow-ri_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def frames(n, seed):
    from piv_synth import synth_piv
    a, b = synth_piv(n, n, seed=seed)
    return a.astype(np.float32), b.astype(np.float32)


def oracle_flow(job):
    kind, n, seed = job
    from oracle import farneback as ofb, hs as ohs, liushen as ols, pyramid as opyr
    a, b = frames(n, seed)
    if kind == "hs2":            # HS Fs3.4, 2 levels, alphas [21,45], 100 sweeps per level
        U, V = opyr.pyramidal_flow(a, b, 3.4, ohs.HSAdapter([21, 45], 100), 2, 1, None)
    elif kind == "fb2":          # Farneback Fs0.0, 2 levels, adapter defaults
        U, V = opyr.pyramidal_flow(a, b, 0.0, ofb.FarnebackAdapter(), 2, 1, None)
    elif kind == "fb2ls":        # the same refined by Liu-Shen(10), Fs_opt 0.48
        U, V = opyr.pyramidal_flow(a, b, 0.0, ofb.FarnebackAdapter(), 2, 1, 0.48, ols.LiuShenAdapter(10))
    elif kind == "hs2ls":        # the batch configuration's per-pair call
        U, V = opyr.pyramidal_flow(a, b, 3.4, ohs.HSAdapter([21, 45], 100), 2, 1, 0.48, ols.LiuShenAdapter(5))
    else:
        raise ValueError(kind)
    return job, np.asarray(U, np.float32), np.asarray(V, np.float32)

#!/usr/bin/env python3
"""Liu-Shen: 60 sweeps per call under every launch shape (ofri_ls_set_mode), HIP events on the context's stream.
    python tools/ls_modes_time.py [sizes...]      (default 2048 1024 4096)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "opticalflow-ri_amd")]
import _ofri  # noqa: E402
from piv_synth import synth_piv  # noqa: E402
from PhysicsBasedOpticalFlowLiuShen import liushen_device  # noqa: E402

ctx = _ofri.context()
sizes = [int(a) for a in sys.argv[1:]] or [2048, 1024, 4096]
for n in sizes:
    f0, f1 = synth_piv(n, n, seed=0)
    im0, im1, z = ctx.upload(f0.astype(np.float32)), ctx.upload(f1.astype(np.float32)), ctx.zeros((n, n))
    for mode in (-1, 1, 6, 8, 0):
        ctx.set_ls_mode(mode)
        liushen_device(im0, im1, 5, z, z, False); ctx.sync()
        e0, e1 = ctx.event(), ctx.event()
        ctx.record(e0)
        for _ in range(5):
            liushen_device(im0, im1, 5, z, z, False)
        ctx.record(e1)
        ms = ctx.elapsed_ms(e0, e1) / 5
        print(json.dumps({"size": n, "mode": mode, "ms_per_call_60_sweeps": round(ms, 4), "us_per_sweep": round(ms / 60 * 1e3, 2),
                          "GBs_at_48B_px_sweep": round(48.0 * n * n * 60 / (ms * 1e-3) / 1e9, 1)}), flush=True)
    ctx.set_ls_mode(0)

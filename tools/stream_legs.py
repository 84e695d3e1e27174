#!/usr/bin/env python3
"""Which leg bounds the overlapped frame-pair stream (piv_ingest.stream_flows)?  Times, per 4096 x 4096 pair:
compute alone, upload + compute, compute + download, and all three."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "opticalflow-ri_amd")]
import _ofri
from piv_synth import synth_piv
from piv_ingest import PairFeeder, FlowDrain, stream_flows
from HornSchunck import HSOpticalFlowAlgoAdapter
from GenericPyramidalOpticalFlow import pyramidal_flow_device

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = 16
ctx = _ofri.context(0)
f0, f1 = synth_piv(n, n, seed=0)
flow = lambda a, b: pyramidal_flow_device(a, b, 3.4, HSOpticalFlowAlgoAdapter([21, 45], 100), 2, 1, None)
d0, d1 = ctx.upload(f0), ctx.upload(f1)


def timed(name, fn):
    fn(2)
    ctx.sync()
    t = time.perf_counter()
    fn(N)
    ctx.sync()
    dt = (time.perf_counter() - t) / N
    print("%-28s %.2f ms per pair  %.0f Mpix/s" % (name, dt * 1e3, n * n / dt / 1e6), flush=True)


def compute_only(k):
    for _ in range(k):
        flow(d0, d1)


def upload_compute(k):
    for a, b in PairFeeder(ctx, ((f0, f1) for _ in range(k))):
        flow(a, b)


def compute_download(k):
    drain, prev = FlowDrain(ctx), None
    for _ in range(k):
        t = drain.post(*flow(d0, d1))
        if prev is not None:
            drain.fetch(prev)
        prev = t
    drain.fetch(prev)


def download_only(k):
    U, V = flow(d0, d1)
    ctx.sync()
    drain = FlowDrain(ctx)
    t0 = time.perf_counter()
    for _ in range(k):
        drain.fetch(drain.post(U, V))
    print("   (download alone: %.2f ms per pair)" % ((time.perf_counter() - t0) / k * 1e3))


def all_three(k):
    for _ in stream_flows(ctx, ((f0, f1) for _ in range(k)), flow):
        pass


timed("compute", compute_only)
timed("upload + compute", upload_compute)
download_only(4)
timed("compute + download", compute_download)
timed("upload + compute + download", all_three)

"""A frame-pair pipeline captured once and replayed: for streams of equally sized pairs.

Below about 2048 x 2048 pixels the rate of ``genericPyramidalOpticalFlow`` is set by the host (one Python call
and one kernel launch per step of the driver), not by the GPU.  ``FlowPlan`` runs the driver once while the
context records its launches into a HIP graph, and every later pair costs two copies and one graph launch.  The
reference has no counterpart (its driver is re-interpreted per pair, GenericPyramidalOpticalFlow.py:256-416);
results are the same bits as ``genericPyramidalOpticalFlow`` with the same arguments.

    plan = FlowPlan((1024, 1024), 3.4, lambda: HSOpticalFlowAlgoAdapter([21, 45], 100), pyramidalLevels=2)
    for a, b in pairs:
        U, V = plan(a, b)            # NumPy in, NumPy out
"""
import numpy as np

import _ofri
from GenericPyramidalOpticalFlow import pyramidal_flow_device


class FlowPlan(object):
    """``make_adapter``: a callable that returns a fresh main adapter (adapters consume their parameter lists, so
    every run of the driver needs a new one).  Works for sequences that never wait for the GPU: all four adapters
    (Liu-Shen's stopping rule stays on the GPU; filter taps and window parameters are kernel arguments) with the
    bilinear warp or none.  Sequences that read something back (the biLinear=False warp's range check) raise
    OfriError at construction."""

    def __init__(self, shape, FILTER, make_adapter, pyramidalLevels=1, kLevels=1, FILTER_OPT=None, ctx=None, timed=False,
                 **driver_args):
        """``timed``: capture the Horn-Schunck sweep launches as timed sections (prof_records() after a run)."""
        self.ctx = ctx if ctx is not None else _ofri.context()
        self.shape = (int(shape[0]), int(shape[1]))
        self.im1, self.im2 = self.ctx.zeros(self.shape), self.ctx.zeros(self.shape)

        def run():
            return pyramidal_flow_device(self.im1, self.im2, FILTER, make_adapter(), int(pyramidalLevels), int(kLevels),
                                         FILTER_OPT, **driver_args)

        run()                               # builds the cached tables and fills the pool: the capture allocates nothing new
        if timed:
            self.ctx.prof_enable(True)
        try:
            self.ctx.graph_begin()
            try:
                self.U, self.V = run()
            except Exception:
                try:
                    self.ctx.graph_end().close()
                except _ofri.OfriError:
                    pass
                raise
            self.graph = self.ctx.graph_end()
        finally:
            if timed:
                self.ctx.prof_enable(False)

    def prof_records(self):
        """Timed sections of the last run (a plan made with timed=True): [(ms, launches, sweeps, rows, cols)]."""
        return self.graph.prof_records()

    def load(self, frame1, frame2):
        """Put a pair into the plan's input arrays: NumPy arrays (uint8, uint16 or anything float32 holds) or
        DeviceArrays of the plan's shape."""
        for dst, src in ((self.im1, frame1), (self.im2, frame2)):
            if isinstance(src, _ofri.DeviceArray):
                if src.shape != self.shape or src.dtype != np.float32:
                    raise ValueError("expected a float32 DeviceArray of shape %r" % (self.shape,))
                _ofri.check(_ofri.lib().ofri_copy(self.ctx.handle, dst.ptr, src.ptr, dst.nbytes))
            else:
                self.ctx.upload_into(dst, src)

    def run(self):
        """Replay on the loaded pair; returns the plan's own (U, V) DeviceArrays, overwritten by the next run."""
        self.graph.launch()
        return self.U, self.V

    def __call__(self, frame1, frame2):
        self.load(frame1, frame2)
        U, V = self.run()
        return U.numpy(), V.numpy()

    def close(self):
        self.graph.close()


class PlanPool(object):
    """Several FlowPlans on contexts (streams) of their own, fed round-robin: small frames do not fill the GPU one
    pair at a time (at 1024 x 1024 two pairs in flight give 1.8 times the rate of one).  ``flows(pairs)`` yields
    (U, V) NumPy arrays in the order of the pairs."""

    def __init__(self, shape, FILTER, make_adapter, in_flight=2, device=0, **plan_args):
        self.plans = [FlowPlan(shape, FILTER, make_adapter, ctx=_ofri.Context(device), **plan_args)
                      for _ in range(max(1, int(in_flight)))]

    def flows(self, pairs):
        busy = []                                     # plans whose result has not been collected, oldest first
        free = list(self.plans)
        for a, b in pairs:
            if not free:
                p = busy.pop(0)
                yield p.U.numpy(), p.V.numpy()        # waits for that plan's stream only
                free.append(p)
            p = free.pop(0)
            p.load(a, b)
            p.run()
            busy.append(p)
        for p in busy:
            yield p.U.numpy(), p.V.numpy()

    def close(self):
        for p in self.plans:
            p.close()
            p.ctx.close()


class StreamedPlan(object):
    """Host frame pairs in, host flow fields out, with upload, compute and download overlapped
    (piv_ingest.stream_flows) and the compute replayed from captured graphs.  Two plans alternate, so that the
    result of pair i can still be on its way to the host while pair i+1 is computed into the other plan's arrays;
    a plan is reused only after its last result has been fetched."""

    def __init__(self, shape, FILTER, make_adapter, ctx=None, **plan_args):
        self.ctx = ctx if ctx is not None else _ofri.context()
        self.plans = [FlowPlan(shape, FILTER, make_adapter, ctx=self.ctx, **plan_args) for _ in range(2)]

    def flows(self, pairs, range_rule=False):
        from piv_ingest import stream_flows
        turn = [0]

        def flow(im0, im1):
            p = self.plans[turn[0] & 1]
            turn[0] += 1
            p.load(im0, im1)
            return p.run()

        return stream_flows(self.ctx, pairs, flow, range_rule)

    def close(self):
        for p in self.plans:
            p.close()

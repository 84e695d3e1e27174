"""Frame-pair ingestion for batches: raw 8/16-bit frames go over PCIe as integers from pinned staging
buffers on a copy stream, are widened to float32 on the GPU, and the next pair's upload overlaps the
current pair's compute.  The reference reads one TIFF pair with ``imread(...).astype(np.float32)``
(examples/PyHSchunck_Fs3_4.py:129-130) and its benchmark script rescales 16-bit data into the 8-bit range
(benchmark_of_methods.py:134-137); both conversions are reproduced on the device."""
import ctypes as C
from concurrent.futures import ThreadPoolExecutor

import numpy as np

import _ofri

_copiers = None


def host_copy(dst, src, threads=4):
    """dst[...] = src for large frames, split by rows over a few threads (NumPy releases the interpreter lock while
    it copies): one core moves about 10 GB/s, which for a 4096 x 4096 pair is longer than the GPU needs for its flow."""
    global _copiers
    n = src.shape[0] if src.ndim else 0
    if src.nbytes < (4 << 20) or n < threads:
        dst[...] = src
        return
    if _copiers is None:
        _copiers = ThreadPoolExecutor(max_workers=threads, thread_name_prefix="ofri-stage")
    step = (n + threads - 1) // threads
    jobs = [_copiers.submit(np.copyto, dst[a:a + step], src[a:a + step]) for a in range(0, n, step)]
    for j in jobs:
        j.result()


def upload_pair(ctx, frame0, frame1, range_rule=False):
    """Upload two frames (uint8, uint16 or float arrays) and return float32 DeviceArrays."""
    lib = _ofri.lib()
    out = []
    for f in (frame0, frame1):
        f = np.ascontiguousarray(f)
        d = ctx.empty(f.shape)
        if f.dtype == np.uint8:
            raw = ctx.upload(f, np.uint8)
            _ofri.check(lib.ofri_u8_to_f32(ctx.handle, raw.ptr, d.ptr, f.size))
        elif f.dtype == np.uint16:
            raw = ctx.upload(f, np.uint16)
            _ofri.check(lib.ofri_u16_to_f32(ctx.handle, raw.ptr, d.ptr, f.size))
        else:
            d = ctx.upload(f)
        out.append(d)
    if range_rule:
        _ofri.check(lib.ofri_range_rule_16bit(ctx.handle, out[0].ptr, out[1].ptr, out[0].size))
    return out[0], out[1]


class PinnedBuffer(object):
    def __init__(self, nbytes):
        self.ptr = C.c_void_p()
        _ofri.check(_ofri.lib().ofri_host_alloc(int(nbytes), C.byref(self.ptr)))
        self.nbytes = int(nbytes)

    def array(self, shape, dtype):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        assert n <= self.nbytes
        return np.ctypeslib.as_array((C.c_ubyte * n).from_address(self.ptr.value)).view(dtype).reshape(shape)

    def __del__(self):
        try:
            if self.ptr:
                _ofri.lib().ofri_host_free(self.ptr)
        except Exception:
            pass
        self.ptr = None


class _PooledStage(object):
    """A staging block out of the process-wide pinned pool (kept for the next feeder instead of going back to the
    driver: allocating pinned memory costs milliseconds); PinnedBuffer when the pool is at its limit."""

    def __init__(self, nbytes):
        self._lease = _ofri._pinned.lease(int(nbytes))
        self._own = PinnedBuffer(nbytes) if self._lease is None else None
        self.ptr = C.c_void_p(self._lease.ptr) if self._lease is not None else self._own.ptr

    def array(self, shape, dtype):
        if self._own is not None:
            return self._own.array(shape, dtype)
        return np.asarray(self._lease).view(dtype).reshape(shape)


class PairFeeder(object):
    """Double-buffered feeder: ``for im0, im1 in PairFeeder(ctx, pairs): ...`` yields device-resident
    float32 frames of pair i while pair i+1 is already being copied on a second stream.  The staging of pair i+1
    (host copy into pinned memory, then the upload) runs on a worker thread, so the caller's thread is free to
    enqueue the work on pair i meanwhile."""

    def __init__(self, ctx, pairs, range_rule=False):
        self.ctx, self.pairs, self.range_rule = ctx, pairs, range_rule
        self.copy_ctx = _ofri.Context(ctx.device)
        self._stage = {}
        self._events = {}
        self._worker = ThreadPoolExecutor(max_workers=1, thread_name_prefix="ofri-feed")

    def _post(self, slot, pair):
        lib = _ofri.lib()
        f0, f1 = (np.ascontiguousarray(f) for f in pair)
        key = (slot, f0.shape, f0.dtype.str)
        if key not in self._stage:
            self._stage[key] = (_PooledStage(f0.nbytes), _PooledStage(f1.nbytes),
                                self.copy_ctx.empty(f0.shape, f0.dtype), self.copy_ctx.empty(f1.shape, f1.dtype))
        p0, p1, r0, r1 = self._stage[key]
        ev = self._events.get(key)
        if ev is not None:                       # the last upload out of these pinned buffers must have left the host
            _ofri.check(lib.ofri_event_sync(ev))
        host_copy(p0.array(f0.shape, f0.dtype), f0)
        host_copy(p1.array(f1.shape, f1.dtype), f1)
        _ofri.check(lib.ofri_upload_async(self.copy_ctx.handle, r0.ptr, p0.ptr, f0.nbytes))
        _ofri.check(lib.ofri_upload_async(self.copy_ctx.handle, r1.ptr, p1.ptr, f1.nbytes))
        if key not in self._events:
            self._events[key] = self.copy_ctx.event()
        self.copy_ctx.record(self._events[key])
        return r0, r1, f0.dtype

    def _widen(self, r0, r1, dtype):
        lib, ctx = _ofri.lib(), self.ctx
        _ofri.check(lib.ofri_ctx_wait(ctx.handle, self.copy_ctx.handle))     # compute stream waits for the copy
        outs = []
        for r in (r0, r1):
            d = ctx.empty(r.shape)
            if dtype == np.uint8:
                _ofri.check(lib.ofri_u8_to_f32(ctx.handle, r.ptr, d.ptr, r.size))
            elif dtype == np.uint16:
                _ofri.check(lib.ofri_u16_to_f32(ctx.handle, r.ptr, d.ptr, r.size))
            elif dtype == np.float32:
                _ofri.check(lib.ofri_copy(ctx.handle, d.ptr, r.ptr, r.nbytes))
            else:
                raise TypeError("frames must be uint8, uint16 or float32")
            outs.append(d)
        if self.range_rule:
            _ofri.check(lib.ofri_range_rule_16bit(ctx.handle, outs[0].ptr, outs[1].ptr, outs[0].size))
        return outs[0], outs[1]

    def __iter__(self):
        it = iter(self.pairs)
        try:
            nxt = self._worker.submit(self._post, 0, next(it))
        except StopIteration:
            return
        slot = 0
        try:
            while nxt is not None:
                cur = nxt.result()                   # the worker is idle from here until the next submit
                nxt = None
                im0, im1 = self._widen(*cur)
                # the staging slot we are about to refill was consumed by the widen kernels just enqueued:
                # order the copy stream after them
                slot ^= 1
                try:
                    pair = next(it)
                except StopIteration:
                    pair = None
                if pair is not None:
                    _ofri.check(_ofri.lib().ofri_ctx_wait(self.copy_ctx.handle, self.ctx.handle))
                    nxt = self._worker.submit(self._post, slot, pair)
                yield im0, im1
        finally:
            # a consumer that stops early leaves an upload under way: let it finish before the buffers can go
            self._worker.shutdown(wait=True)
            self.close()

    def close(self):
        """Release the copy stream, its events and its staging arrays.  The widen kernels on the compute stream read
        the raw frames, so that stream is drained first; the raw frames' blocks then leave with the copy context."""
        if self.copy_ctx is None:
            return
        lib = _ofri.lib()
        self.ctx.sync()
        self.copy_ctx.sync()
        for ev in self._events.values():
            lib.ofri_event_destroy(ev)
        self._events.clear()
        self._stage.clear()              # pinned blocks go back to the process-wide pool, raw frames to the copy context
        self.copy_ctx.close()
        self.copy_ctx = None


class FlowDrain(object):
    """Results leave the GPU on a copy stream of their own, into pooled pinned memory, while the next pair is
    being computed: ``t = drain.post(U, V)`` enqueues the copies behind the work already on the compute stream,
    ``drain.fetch(t)`` waits for them and returns ordinary ndarrays (the reference's scripts get their flow
    fields as host arrays, examples/PyHSchunck_Fs3_4.py:138-140)."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.copy_ctx = _ofri.Context(ctx.device)
        self._open = {}                  # tickets posted and not fetched: id -> ticket

    def post(self, *arrays):
        lib = _ofri.lib()
        _ofri.check(lib.ofri_ctx_wait(self.copy_ctx.handle, self.ctx.handle))     # the copy stream waits for the compute stream
        leases = []
        for a in arrays:
            lease = _ofri._pinned.lease(a.nbytes) if a.nbytes >= _ofri._PinnedPool.MIN_BYTES else None
            if lease is not None:
                _ofri.check(lib.ofri_download_async(self.copy_ctx.handle, C.c_void_p(lease.ptr), a.ptr, a.nbytes))
            leases.append(lease)
        ev = self.copy_ctx.event()
        self.copy_ctx.record(ev)
        # the device arrays stay referenced until fetch: their memory must not go back to the pool under the copy
        t = (arrays, leases, ev)
        self._open[id(t)] = t
        return t

    def fetch(self, ticket):
        arrays, leases, ev = ticket
        lib = _ofri.lib()
        self._open.pop(id(ticket), None)
        _ofri.check(lib.ofri_event_sync(ev))
        lib.ofri_event_destroy(ev)
        out = []
        for a, lease in zip(arrays, leases):
            if lease is None:                    # small result, or the caller holds the pinned pool's limit: plain copy
                out.append(a.numpy())
            else:
                out.append(np.asarray(lease).view(a.dtype).reshape(a.shape))
        return tuple(out)

    def close(self):
        """Wait for copies still under way (a consumer that stopped early leaves a ticket unfetched: its pinned blocks
        and device arrays must not be handed out again under the copy), then release the copy stream."""
        if self.copy_ctx is None:
            return
        lib = _ofri.lib()
        self.copy_ctx.sync()
        for _, _, ev in self._open.values():
            lib.ofri_event_destroy(ev)
        self._open.clear()
        self.copy_ctx.close()
        self.copy_ctx = None


def stream_flows(ctx, pairs, flow_fn, range_rule=False):
    """Flow fields of a sequence of host frame pairs with all three legs overlapped: while ``flow_fn(im0, im1)``
    (device frames in, a tuple of DeviceArrays out) works on pair i, pair i+1 is uploaded and the result of pair
    i-1 is copied back.  Yields the host results in pair order."""
    drain = FlowDrain(ctx)
    prev = None
    try:
        for im0, im1 in PairFeeder(ctx, pairs, range_rule):
            t = drain.post(*flow_fn(im0, im1))
            if prev is not None:
                yield drain.fetch(prev)
            prev = t
        if prev is not None:
            t, prev = prev, None
            yield drain.fetch(t)
    finally:
        drain.close()

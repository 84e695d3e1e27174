"""Frame-pair ingestion for batches: raw 8/16-bit frames go over PCIe as integers from pinned staging
buffers on a copy stream, are widened to float32 on the GPU, and the next pair's upload overlaps the
current pair's compute.  The reference reads one TIFF pair with ``imread(...).astype(np.float32)``
(examples/PyHSchunck_Fs3_4.py:129-130) and its benchmark script rescales 16-bit data into the 8-bit range
(benchmark_of_methods.py:134-137); both conversions are reproduced on the device."""
import ctypes as C

import numpy as np

import _ofri


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


class PairFeeder(object):
    """Double-buffered feeder: ``for im0, im1 in PairFeeder(ctx, pairs): ...`` yields device-resident
    float32 frames of pair i while pair i+1 is already being copied on a second stream."""

    def __init__(self, ctx, pairs, range_rule=False):
        self.ctx, self.pairs, self.range_rule = ctx, pairs, range_rule
        self.copy_ctx = _ofri.Context(ctx.device)
        self._stage = {}
        self._events = {}

    def _post(self, slot, pair):
        lib = _ofri.lib()
        f0, f1 = (np.ascontiguousarray(f) for f in pair)
        key = (slot, f0.shape, f0.dtype.str)
        if key not in self._stage:
            self._stage[key] = (PinnedBuffer(f0.nbytes), PinnedBuffer(f1.nbytes),
                                self.copy_ctx.empty(f0.shape, f0.dtype), self.copy_ctx.empty(f1.shape, f1.dtype))
        p0, p1, r0, r1 = self._stage[key]
        ev = self._events.get(key)
        if ev is not None:                       # the last upload out of these pinned buffers must have left the host
            _ofri.check(lib.ofri_event_sync(ev))
        p0.array(f0.shape, f0.dtype)[...] = f0
        p1.array(f1.shape, f1.dtype)[...] = f1
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
            nxt = self._post(0, next(it))
        except StopIteration:
            return
        slot = 0
        while nxt is not None:
            cur = nxt
            im0, im1 = self._widen(*cur)
            # the staging slot we are about to refill was consumed by the widen kernels just enqueued:
            # order the copy stream after them
            slot ^= 1
            try:
                pair = next(it)
                _ofri.check(_ofri.lib().ofri_ctx_wait(self.copy_ctx.handle, self.ctx.handle))
                nxt = self._post(slot, pair)
            except StopIteration:
                nxt = None
            yield im0, im1

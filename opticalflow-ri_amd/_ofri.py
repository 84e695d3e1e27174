"""ctypes binding of libofri_hip.so (the C ABI declared in include/ofri.h).

Host-side plumbing only: library loading, error translation, a per-device context and a thin
device-array handle.  All arithmetic happens in the HIP library; there is no CPU fallback - if the
library or a GPU is missing, importing works but the first call raises OfriError.
"""
import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OFRI_LIB_PATH") or os.path.join(_HERE, "libofri_hip.so")     # OFRI_LIB_PATH: another build of the library


class OfriError(RuntimeError):
    pass


_lib = None
_lock = threading.Lock()

_vp, _i, _f, _d, _sz = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t
_PROTOS = {
    "ofri_version": ([], _i),
    "ofri_last_error": ([], C.c_char_p),
    "ofri_device_count": ([C.POINTER(_i)], _i),
    "ofri_ctx_create": ([_i, _vp, C.POINTER(_vp)], _i),
    "ofri_ctx_destroy": ([_vp], _i),
    "ofri_ctx_sync": ([_vp], _i),
    "ofri_ctx_stream": ([_vp, C.POINTER(_vp)], _i),
    "ofri_alloc": ([_vp, _sz, C.POINTER(_vp)], _i),
    "ofri_free": ([_vp, _vp], _i),
    "ofri_pool_trim": ([_vp], _i),
    "ofri_alloc_stats": ([_vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)], _i),
    "ofri_pool_guard_check": ([_vp, C.POINTER(_i), C.POINTER(C.c_ulonglong), C.POINTER(C.c_longlong)], _i),
    "ofri_upload": ([_vp, _vp, _vp, _sz], _i),
    "ofri_download": ([_vp, _vp, _vp, _sz], _i),
    "ofri_copy": ([_vp, _vp, _vp, _sz], _i),
    "ofri_zero": ([_vp, _vp, _sz], _i),
    "ofri_u8_to_f32": ([_vp, _vp, _vp, _sz], _i),
    "ofri_u16_to_f32": ([_vp, _vp, _vp, _sz], _i),
    "ofri_range_rule_16bit": ([_vp, _vp, _vp, _sz], _i),
    "ofri_host_alloc": ([_sz, C.POINTER(_vp)], _i),
    "ofri_host_free": ([_vp], _i),
    "ofri_upload_async": ([_vp, _vp, _vp, _sz], _i),
    "ofri_download_async": ([_vp, _vp, _vp, _sz], _i),
    "ofri_ctx_wait": ([_vp, _vp], _i),
    "ofri_graph_begin": ([_vp], _i),
    "ofri_graph_end": ([_vp, C.POINTER(_vp)], _i),
    "ofri_graph_launch": ([_vp, _vp], _i),
    "ofri_graph_destroy": ([_vp], _i),
    "ofri_graph_prof_count": ([_vp, C.POINTER(_i)], _i),
    "ofri_graph_prof_read": ([_vp, _i, C.POINTER(_f), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)], _i),
    "ofri_event_create": ([_vp, C.POINTER(_vp)], _i),
    "ofri_event_record": ([_vp, _vp], _i),
    "ofri_event_elapsed_ms": ([_vp, _vp, C.POINTER(_f)], _i),
    "ofri_event_sync": ([_vp], _i),
    "ofri_event_destroy": ([_vp], _i),
    "ofri_gauss_sep": ([_vp, _vp, _vp, _i, _i, _vp, _i], _i),
    "ofri_gauss_taps": ([_d, _i, _vp], _i),
    "ofri_hs_derivs": ([_vp, _vp, _vp, _vp, _vp, _vp, _i, _i], _i),
    "ofri_hs_iterate": ([_vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _i, _i], _i),
    "ofri_hs_set_fuse": ([_vp, _i], _i),
    "ofri_set_share": ([_vp, _i], _i),
    "ofri_prof_enable": ([_vp, _i], _i),
    "ofri_prof_count": ([_vp, C.POINTER(_i)], _i),
    "ofri_prof_read": ([_vp, _i, C.POINTER(_f), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)], _i),
    "ofri_hs_compute": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _i, _i, C.POINTER(_d)], _i),
    "ofri_hs_compute_static": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _i, _i, C.POINTER(_d)], _i),
    "ofri_hs_compute_acc": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _i, _i], _i),
    "ofri_hs_level": ([_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _i, _i, _i], _i),
    "ofri_ls_compute": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _d, _i, _i, C.POINTER(_d), C.POINTER(_i)], _i),
    "ofri_ls_set_mode": ([_vp, _i], _i),
    "ofri_fb_set_mode": ([_vp, _i], _i),
    "ofri_fb_polyexp": ([_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp], _i),
    "ofri_fb_blur_polyexp": ([_vp, _vp, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp], _i),
    "ofri_fb_blur": ([_vp, _vp, _vp, _i, _i, _vp, _i], _i),
    "ofri_fb_blur5": ([_vp, _vp, _vp, _i, _i, _vp, _i], _i),
    "ofri_fb_box5": ([_vp, _vp, _vp, _i, _i, _i], _i),
    "ofri_fb_update_matrices": ([_vp, _vp, _vp, _vp, _vp, _vp, _i, _i], _i),
    "ofri_fb_update_flow": ([_vp, _vp, _vp, _vp, _i, _i], _i),
    "ofri_fb_window5_update_flow": ([_vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i], _i),
    "ofri_fb_window5_flow_matrices": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i], _i),
    "ofri_lk_compute": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i], _i),
    "ofri_lk_set_mode": ([_vp, _i], _i),
    "ofri_resize_bicubic": ([_vp, _vp, _i, _i, _vp, _i, _i], _i),
    "ofri_resize_bicubic_pair": ([_vp, _vp, _vp, _i, _i, _vp, _vp, _i, _i], _i),
    "ofri_resize_bilinear": ([_vp, _vp, _i, _i, _vp, _i, _i], _i),
    "ofri_spline_upsample": ([_vp, _vp, _i, _i, _vp, _i, _i, _f], _i),
    "ofri_spline_upsample_pair": ([_vp, _vp, _vp, _i, _i, _vp, _vp, _i, _i, _f, _f], _i),
    "ofri_warp_coords": ([_vp, _vp, _vp, _vp, _vp, _i, _i], _i),
    "ofri_warp_flow": ([_vp, _vp, _vp, _vp, _f, _vp, _i, _i], _i),
    "ofri_warp_pair": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i], _i),
    "ofri_warp_shift_ofe": ([_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i], _i),
    "ofri_add_inplace": ([_vp, _vp, _vp, _sz], _i),
    "ofri_scale_inplace": ([_vp, _vp, _f, _sz], _i),
    "ofri_norm_diff": ([_vp, _vp, _vp, _sz, C.POINTER(_d)], _i),
    "ofri_max": ([_vp, _vp, _sz, C.POINTER(_f)], _i),
}


def lib():
    """Load libofri_hip.so once; raise OfriError if it has not been built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise OfriError("HIP library not found at %s - build it with `make -C %s` "
                                    "(or python -c 'import __graft_entry__ as g; g.build()'); "
                                    "there is no CPU fallback" % (LIB_PATH, os.path.join(_HERE, "csrc")))
                l = C.CDLL(LIB_PATH)
                for name, (args, res) in _PROTOS.items():
                    fn = getattr(l, name, None)
                    if fn is None:
                        continue          # later additions are bound where they are used
                    fn.argtypes, fn.restype = args, res
                _lib = l
    return _lib


def bind(name, args, res=_i):
    """Declare an entry point outside the core table (used by the per-algorithm modules)."""
    fn = getattr(lib(), name)
    fn.argtypes, fn.restype = args, res
    _PROTOS.setdefault(name, (args, res))
    return fn


def check(rc):
    if rc != 0:
        msg = lib().ofri_last_error()
        if rc == -5:                                   # OFRI_ERANGE: the reference's IndexError
            raise IndexError(msg.decode("utf-8", "replace") if msg else "index out of bounds")
        raise OfriError("libofri_hip error %d: %s" % (rc, msg.decode("utf-8", "replace") if msg else "?"))


class _PinnedLease(object):
    """A block of pinned host memory that backs one ndarray handed to the caller.  The ndarray keeps
    this object alive through its .base chain; when the last view dies the block goes back to the pool."""

    __slots__ = ("ptr", "cap", "nbytes", "__weakref__")

    def __init__(self, ptr, cap, nbytes):
        self.ptr, self.cap, self.nbytes = ptr, cap, nbytes

    @property
    def __array_interface__(self):
        return {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 3}

    def __del__(self):
        try:
            _pinned.give_back(self.ptr, self.cap)
        except Exception:
            pass


class _PinnedPool(object):
    """Process-wide free lists of pinned host blocks for results copied back from the GPU.  A copy into
    pinned memory runs at the link rate (about 57 GB/s on MI355X hosts against 11 GB/s into a freshly
    allocated pageable array), and frame loops ask for the same sizes again and again."""

    MIN_BYTES = 1 << 20            # smaller results take the ordinary pageable path
    GRAIN = 1 << 21
    MAX_IDLE_BYTES = 2 << 30       # idle blocks kept for reuse; the rest goes back to the driver
    MAX_LIVE_BYTES = 16 << 30      # results the caller still holds; beyond this they are pageable again

    def __init__(self):
        self._free = {}
        self._idle = 0
        self._live = 0
        self._lock = threading.Lock()

    def lease(self, nbytes):
        """A block of at least nbytes, or None when the caller already holds MAX_LIVE_BYTES of results
        (a loop that keeps every flow field must not pin the whole host memory)."""
        cap = (nbytes + self.GRAIN - 1) // self.GRAIN * self.GRAIN
        with self._lock:
            if self._live + cap > self.MAX_LIVE_BYTES:
                return None
            self._live += cap
            lst = self._free.get(cap)
            if lst:
                self._idle -= cap
                return _PinnedLease(lst.pop(), cap, nbytes)
        p = _vp()
        if lib().ofri_host_alloc(cap, C.byref(p)) != 0:
            with self._lock:
                self._live -= cap
            return None
        return _PinnedLease(p.value, cap, nbytes)

    def give_back(self, ptr, cap):
        with self._lock:
            self._live -= cap
            if self._idle + cap <= self.MAX_IDLE_BYTES:
                self._free.setdefault(cap, []).append(ptr)
                self._idle += cap
                return
        lib().ofri_host_free(_vp(ptr))

    def trim(self):
        with self._lock:
            blocks = [p for lst in self._free.values() for p in lst]
            self._free.clear()
            self._idle = 0
        for p in blocks:
            lib().ofri_host_free(_vp(p))


_pinned = _PinnedPool()


class Context(object):
    """One GPU, one stream, one memory pool."""

    def __init__(self, device=0, stream=None):
        self._h = _vp()
        check(lib().ofri_ctx_create(int(device), _vp(stream) if stream else None, C.byref(self._h)))
        self.device = int(device)

    @property
    def handle(self):
        return self._h

    def sync(self):
        check(lib().ofri_ctx_sync(self._h))

    def stream(self):
        s = _vp()
        check(lib().ofri_ctx_stream(self._h, C.byref(s)))
        return s.value

    def set_hs_fuse(self, t):
        check(lib().ofri_hs_set_fuse(self._h, int(t)))

    def set_share(self, contexts):
        """Tell the row-streaming pipelines that ``contexts`` contexts keep this GPU busy together: longer row chunks,
        less recomputation, longer latency of a single pair (same bits)."""
        check(lib().ofri_set_share(self._h, int(contexts)))

    def set_fb_mode(self, mode):
        check(lib().ofri_fb_set_mode(self._h, int(mode)))

    def set_lk_mode(self, mode):
        check(lib().ofri_lk_set_mode(self._h, int(mode)))

    def set_ls_mode(self, mode):
        check(lib().ofri_ls_set_mode(self._h, int(mode)))

    def prof_enable(self, on=True):
        check(lib().ofri_prof_enable(self._h, 1 if on else 0))

    def prof_records(self):
        """[(ms, launches, sweeps, rows, cols)] of the Horn-Schunck sweep loops run since prof_enable."""
        n = _i(0)
        check(lib().ofri_prof_count(self._h, C.byref(n)))
        out = []
        for k in range(n.value):
            ms, la, sw, r, c = _f(), _i(), _i(), _i(), _i()
            check(lib().ofri_prof_read(self._h, k, C.byref(ms), C.byref(la), C.byref(sw), C.byref(r), C.byref(c)))
            out.append((ms.value, la.value, sw.value, r.value, c.value))
        return out

    def close(self):
        if self._h:
            lib().ofri_ctx_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        # every DeviceArray and Graph holds a reference to its context, so nothing of this context is alive any more
        try:
            self.close()
        except Exception:
            pass

    # ---- arrays
    def empty(self, shape, dtype=np.float32):
        return DeviceArray(self, shape, dtype)

    def zeros(self, shape, dtype=np.float32):
        a = DeviceArray(self, shape, dtype)
        check(lib().ofri_zero(self._h, a.ptr, a.nbytes))
        return a

    def upload(self, arr, dtype=np.float32):
        src = np.asarray(arr)
        if np.dtype(dtype) == np.float32 and src.dtype in (np.uint8, np.uint16) and src.size >= (1 << 16):
            # camera frames: send the raw samples and widen on the GPU (exact for both types) instead of
            # converting 4x / 2x the bytes on one host core first
            raw = self.upload(src, src.dtype)
            a = DeviceArray(self, src.shape, np.float32)
            widen = lib().ofri_u8_to_f32 if src.dtype == np.uint8 else lib().ofri_u16_to_f32
            check(widen(self._h, raw.ptr, a.ptr, src.size))
            return a
        arr = np.ascontiguousarray(arr, dtype=dtype)
        a = DeviceArray(self, arr.shape, dtype)
        check(lib().ofri_upload(self._h, a.ptr, arr.ctypes.data, arr.nbytes))
        return a

    def upload_into(self, dst, arr):
        """Copy a host array into the float32 DeviceArray ``dst`` (uint8/uint16 frames go raw and are widened on the GPU)."""
        src = np.asarray(arr)
        if tuple(src.shape) != dst.shape or dst.dtype != np.float32:
            raise ValueError("expected an array of shape %r for a float32 destination" % (dst.shape,))
        if src.dtype in (np.uint8, np.uint16) and src.size >= (1 << 16):
            raw = self.upload(src, src.dtype)
            widen = lib().ofri_u8_to_f32 if src.dtype == np.uint8 else lib().ofri_u16_to_f32
            check(widen(self._h, raw.ptr, dst.ptr, src.size))
            return
        src = np.ascontiguousarray(src, dtype=np.float32)
        check(lib().ofri_upload(self._h, dst.ptr, src.ctypes.data, src.nbytes))

    def wrap(self, ptr, shape, dtype=np.float32, keepalive=None):
        """Adopt external device memory (e.g. a torch tensor's data_ptr()); not freed by us."""
        return DeviceArray(self, shape, dtype, ptr=ptr, keepalive=keepalive)

    # ---- captured launch sequences
    def graph_begin(self):
        check(lib().ofri_graph_begin(self._h))

    def graph_end(self):
        g = _vp()
        check(lib().ofri_graph_end(self._h, C.byref(g)))
        return Graph(self, g)

    # ---- timing
    def event(self):
        e = _vp()
        check(lib().ofri_event_create(self._h, C.byref(e)))
        return e

    def record(self, ev):
        check(lib().ofri_event_record(self._h, ev))

    @staticmethod
    def elapsed_ms(a, b):
        ms = _f()
        check(lib().ofri_event_elapsed_ms(a, b, C.byref(ms)))
        return ms.value


class Graph(object):
    """A captured sequence of calls on a Context (Context.graph_begin / graph_end); launch() replays it."""

    def __init__(self, ctx, handle):
        self.ctx, self._g = ctx, handle

    def launch(self):
        check(lib().ofri_graph_launch(self.ctx.handle, self._g))

    def prof_records(self):
        """[(ms, launches, sweeps, rows, cols)] of the timed sections captured into the graph (the capture ran with
        Context.prof_enable on), as timed by the graph's last replay."""
        n = _i(0)
        check(lib().ofri_graph_prof_count(self._g, C.byref(n)))
        out = []
        for k in range(n.value):
            ms, la, sw, r, c = _f(0), _i(0), _i(0), _i(0), _i(0)
            check(lib().ofri_graph_prof_read(self._g, k, C.byref(ms), C.byref(la), C.byref(sw), C.byref(r), C.byref(c)))
            out.append((ms.value, la.value, sw.value, r.value, c.value))
        return out

    def close(self):
        if self._g and self.ctx._h:
            lib().ofri_graph_destroy(self._g)
        self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceArray(object):
    """A dense row-major array in GPU memory owned by a Context's pool."""

    def __init__(self, ctx, shape, dtype=np.float32, ptr=None, keepalive=None):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self._own = ptr is None
        self._keep = keepalive
        if ptr is None:
            p = _vp()
            check(lib().ofri_alloc(ctx.handle, max(self.nbytes, 1), C.byref(p)))
            self.ptr = p.value
        else:
            self.ptr = int(ptr)

    @property
    def size(self):
        return int(np.prod(self.shape))

    def numpy(self):
        """Copy to the host.  Large results land in pooled pinned memory that the returned ndarray owns
        (an ordinary writable array to the caller); small ones in a plain np.empty."""
        lease = _pinned.lease(self.nbytes) if self.nbytes >= _PinnedPool.MIN_BYTES else None
        if lease is not None:
            check(lib().ofri_download_async(self.ctx.handle, _vp(lease.ptr), self.ptr, self.nbytes))
            self.ctx.sync()
            return np.asarray(lease).view(self.dtype).reshape(self.shape)
        out = np.empty(self.shape, self.dtype)
        if self.nbytes:
            check(lib().ofri_download(self.ctx.handle, out.ctypes.data, self.ptr, self.nbytes))
        return out

    def copy(self):
        b = DeviceArray(self.ctx, self.shape, self.dtype)
        check(lib().ofri_copy(self.ctx.handle, b.ptr, self.ptr, self.nbytes))
        return b

    def __del__(self):
        try:
            if self._own and self.ptr and self.ctx._h:
                lib().ofri_free(self.ctx.handle, self.ptr)
        except Exception:
            pass
        self.ptr = None


_contexts = {}


def context(device=0):
    """Process-wide default context of a device (created on first use)."""
    c = _contexts.get(device)
    if c is None:
        c = _contexts[device] = Context(device)
    return c


def device_count():
    n = _i(0)
    rc = lib().ofri_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def as_f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)

#!/usr/bin/env python3
"""Host<->device copy rates through the C ABI: pageable against pinned buffers, 64 MB float32 fields."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "opticalflow-ri_amd")]
import _ofri

ctx = _ofri.Context(0)
lib = _ofri.lib()
n = 4096 * 4096
d = ctx.zeros((4096, 4096))
src = np.random.rand(4096, 4096).astype(np.float32)
hp = C.c_void_p()
_ofri.check(lib.ofri_host_alloc(n * 4, C.byref(hp)))
pinned = np.frombuffer((C.c_byte * (n * 4)).from_address(hp.value), dtype=np.float32).reshape(4096, 4096)


def t(label, fn, reps=5):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    dt = (time.perf_counter() - t0) / reps
    print("%-46s %7.2f ms  %6.1f GB/s" % (label, dt * 1e3, n * 4 / dt / 1e9), flush=True)


def up_pageable():
    _ofri.check(lib.ofri_upload(ctx.handle, d.ptr, src.ctypes.data, n * 4))


def up_pinned():
    _ofri.check(lib.ofri_upload_async(ctx.handle, d.ptr, hp, n * 4)); ctx.sync()


def up_staged():
    np.copyto(pinned, src); up_pinned()


def down_fresh():
    out = np.empty((4096, 4096), np.float32)
    _ofri.check(lib.ofri_download(ctx.handle, out.ctypes.data, d.ptr, n * 4))


touched = np.zeros((4096, 4096), np.float32)


def down_touched():
    _ofri.check(lib.ofri_download(ctx.handle, touched.ctypes.data, d.ptr, n * 4))


def down_pinned():
    _ofri.check(lib.ofri_download_async(ctx.handle, hp, d.ptr, n * 4)); ctx.sync()


def down_pinned_copy():
    down_pinned(); pinned.copy()


t("upload, pageable source", up_pageable)
t("upload, pinned source", up_pinned)
t("upload, memcpy to pinned + pinned copy", up_staged)
t("download into fresh np.empty", down_fresh)
t("download into touched pageable", down_touched)
t("download into pinned", down_pinned)
t("download into pinned + ndarray.copy()", down_pinned_copy)
t("astype uint8->float32 on the host (16M px)", lambda: src.astype(np.uint8).astype(np.float32), 2)

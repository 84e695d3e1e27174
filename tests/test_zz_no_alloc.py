"""The allocation-free Horn-Schunck entry point (ofri_hs_compute_static) against ofri_hs_compute, through the C ABI.

Both run the same kernels, so flow fields and the error value must agree bit for bit.  The scratch array is shared by
every call on a GPU and keeps the last call's contents: two calls in a row, the second on a smaller field, must each
match.  Collected last (the file name sorts after the other test files)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

_f, _i, _vp = C.c_float, C.c_int, C.c_void_p
ARGS = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _i, _i, C.POINTER(C.c_double)]


@pytest.fixture(scope="module")
def ctx():
    import _ofri
    if _ofri.device_count() < 1:
        pytest.fail("GPU tests need a HIP device and libofri_hip.so; no CPU fallback exists")
    return _ofri.context()


def _alloc_counts(ctx):
    """(allocating HIP runtime calls the library has made so far, ofri_alloc requests on this context so far)"""
    import _ofri
    a, b = C.c_ulonglong(0), C.c_ulonglong(0)
    _ofri.check(_ofri.lib().ofri_alloc_stats(ctx.handle, C.byref(a), C.byref(b)))
    return a.value, b.value


def test_static_entry_allocates_nothing(ctx):
    """The library counts its allocating runtime calls (hipMalloc, hipHostMalloc, event creation) and the requests to a
    context's pool: neither moves across static calls, both move across the pooled entry point."""
    import _ofri
    lib = _ofri.lib()
    static = _ofri.bind("ofri_hs_compute_static", ARGS)
    for shape, niter, prof in (((384, 512), 30, False), ((130, 203), 7, False), ((256, 256), 12, True)):
        a, b = _frames(shape, seed=3)
        rows, cols = shape
        da, db, Uo, Vo = ctx.upload(a), ctx.upload(b), ctx.empty(shape), ctx.empty(shape)
        err = C.c_double(-1.0)
        ctx.prof_enable(prof)                   # the timing records would create events: the static call skips them
        try:
            before = _alloc_counts(ctx)
            for _ in range(2):
                _ofri.check(static(ctx.handle, da.ptr, db.ptr, None, None, Uo.ptr, Vo.ptr, 21.0, niter, rows, cols, C.byref(err)))
            ctx.sync()
            assert _alloc_counts(ctx) == before, (shape, before, _alloc_counts(ctx))
            _ofri.check(lib.ofri_hs_compute(ctx.handle, da.ptr, db.ptr, None, None, Uo.ptr, Vo.ptr, 21.0, niter, rows, cols, C.byref(err)))
            after = _alloc_counts(ctx)
            assert after[1] > before[1]         # the pooled entry point does ask the pool: the counter is alive
        finally:
            ctx.prof_enable(False)


def _both(ctx, a, b, U0, V0, alpha, niter):
    """(U, V, err) from ofri_hs_compute and from ofri_hs_compute_static on the same device inputs."""
    import _ofri
    lib = _ofri.lib()
    static = _ofri.bind("ofri_hs_compute_static", ARGS)
    rows, cols = a.shape
    da, db = ctx.upload(a), ctx.upload(b)
    dU = ctx.upload(U0) if U0 is not None else None
    dV = ctx.upload(V0) if V0 is not None else None
    out = []
    for fn in (lib.ofri_hs_compute, static):
        Uo, Vo = ctx.empty((rows, cols)), ctx.empty((rows, cols))
        err = C.c_double(-1.0)
        _ofri.check(fn(ctx.handle, da.ptr, db.ptr, dU.ptr if dU else None, dV.ptr if dV else None, Uo.ptr, Vo.ptr,
                       float(alpha), int(niter), rows, cols, C.byref(err)))
        out.append((Uo.numpy(), Vo.numpy(), err.value))
    return out


def _frames(shape, seed):
    from piv_synth import synth_piv
    a, b = synth_piv(shape[0], shape[1], seed=seed)
    return a.astype(np.float32), b.astype(np.float32)


def test_static_entry_matches_pooled_entry_two_calls_in_a_row(ctx):
    rng = np.random.default_rng(5)
    # the larger field first, then smaller ones (stale scratch contents around them), a zero start (null pointers),
    # a width that is not a multiple of 4 (padded planes) and a field too small for the tile kernel
    cases = [((512, 640), 21.0, 37, True), ((192, 256), 45.0, 100, False), ((130, 203), 21.0, 19, True),
             ((96, 128), 1.0, 8, False), ((9, 13), 21.0, 5, True)]
    for shape, alpha, niter, start in cases:
        a, b = _frames(shape, seed=shape[0])
        U0 = rng.normal(0, 1, shape).astype(np.float32) if start else None
        V0 = rng.normal(0, 1, shape).astype(np.float32) if start else None
        (Up, Vp, ep), (Us, Vs, es) = _both(ctx, a, b, U0, V0, alpha, niter)
        assert np.isfinite(Up).all() and float(np.abs(Up).max()) > 0
        np.testing.assert_array_equal(Us, Up)
        np.testing.assert_array_equal(Vs, Vp)
        assert es == ep and es > 0


def test_static_entry_exact_mode_and_oracle(ctx):
    from oracle import hs as ohs
    a, b = _frames((128, 160), seed=11)
    ctx.set_hs_fuse(-1)
    try:
        (Up, Vp, ep), (Us, Vs, es) = _both(ctx, a, b, None, None, 21.0, 12)
    finally:
        ctx.set_hs_fuse(0)
    np.testing.assert_array_equal(Us, Up)
    np.testing.assert_array_equal(Vs, Vp)
    Z = np.zeros_like(a)
    Ur, Vr, er = ohs.hs(a, b, 21.0, 12, Z, Z.copy())
    np.testing.assert_array_equal(Us, Ur)                 # exact mode: the reference's bits
    np.testing.assert_array_equal(Vs, Vr)
    assert abs(es - er) <= 1e-5 * abs(er)                 # NumPy sums the float32 norms in float32


def test_static_entry_rejects_fields_beyond_its_scratch(ctx):
    import _ofri
    static = _ofri.bind("ofri_hs_compute_static", ARGS)
    rows, cols = 1024, 1028                                # 1024 * 1028 > 2**20
    d = ctx.empty((rows, cols))
    rc = static(ctx.handle, d.ptr, d.ptr, None, None, d.ptr, d.ptr, 21.0, 1, rows, cols, None)
    assert rc == -1
    assert b"limit" in _ofri.lib().ofri_last_error()
    rc = static(ctx.handle, d.ptr, d.ptr, None, None, d.ptr, d.ptr, 21.0, 1, 1024, 1022, None)   # 1022 pads to 1024: fits,
    assert rc == -1 and b"alias" in _ofri.lib().ofri_last_error()                                # and fails on the aliasing

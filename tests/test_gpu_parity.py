"""HIP path against the CPU oracle and the reference-generated fixtures.  Needs an MI355X.

Tolerances: integer-weight and float64-accumulated steps (derivatives, warp, Gaussian filter, exact
Horn-Schunck mode) are compared bit for bit; the fast float32 Horn-Schunck sweeps and everything
downstream of them to a relative L2 error of 1e-5 on single calls and 1e-4 (the stated target) on
whole pipelines.
"""
import numpy as np
import pytest

from conftest import load_gold, rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import _ofri
    if _ofri.device_count() < 1:
        pytest.fail("GPU tests need a HIP device and libofri_hip.so; no CPU fallback exists")
    return _ofri.context()


def flow_rel(U, V, Ur, Vr):
    return rel_l2(np.stack([U, V]), np.stack([Ur, Vr]))


# ------------------------------------------------------------------------------ Gaussian
def test_gauss_bit_exact(ctx):
    import gaussian_filter as gf
    g = load_gold("gauss")
    for key, (sig, K) in {"crop_f_3.4_3": (3.4, 3), "crop_f_0.48_5": (0.48, 5), "crop_f_1_7": (1.0, 7)}.items():
        a = g["crop"].copy()
        r = gf.gaussian_filterPx(a, sig, K)
        assert r is a
        np.testing.assert_array_equal(a, g[key])
    s = g["small"].copy()
    np.testing.assert_array_equal(gf.gaussian_filter(s, 1.8, 20.0), g["small_f_1.8_t20"])


def test_gauss_full_image_and_ragged(ctx, pair):
    import gaussian_filter as gf
    from oracle import gauss
    g = load_gold("gauss")
    f = gf.gaussian_filterPx(pair[0].copy(), 3.4, 3)
    np.testing.assert_array_equal(f[::2, ::2], g["full0_f_3.4_3_sub"])
    rng = np.random.default_rng(0)
    for shape, sig, K in (((1, 7), 1.0, 3), ((5, 1), 1.0, 3), ((3, 3), 2.0, 5), ((67, 131), 0.9, 9), ((70, 65), 2.0, 11)):
        a = rng.uniform(0, 255, shape).astype(np.float32)
        np.testing.assert_array_equal(gf.gaussian_filterPx(a.copy(), sig, K), gauss.filter_px(a.copy(), sig, K))
    import _ofri
    with pytest.raises(_ofri.OfriError):
        gf.gaussian_filterPx(np.zeros((2, 2), np.float32), 1.0, 7)      # image smaller than half the kernel


# ------------------------------------------------------------------------------ Horn-Schunck
def test_hs_derivatives_bit_exact(ctx):
    import HornSchunck as H
    h = load_gold("hs")
    fx, fy, ft = H.computeDerivatives(h["im1"], h["im0"])
    np.testing.assert_array_equal(fx, h["fx"]); np.testing.assert_array_equal(fy, h["fy"]); np.testing.assert_array_equal(ft, h["ft"])
    pg = load_gold("pair_parabolic01")
    a, b = pg["im1"].astype(np.float32)[7:60, 11:90].copy(), pg["im0"].astype(np.float32)[7:60, 11:90].copy()
    fx, fy, ft = H.computeDerivatives(a, b)
    np.testing.assert_array_equal(fx, h["crop_fx"]); np.testing.assert_array_equal(fy, h["crop_fy"]); np.testing.assert_array_equal(ft, h["crop_ft"])


CASES = [(21, 1, "rand"), (21, 10, "rand"), (21, 1, "zero"), (21, 10, "zero"), (21, 100, "zero"), (1.0, 100, "zero"), (45, 7, "rand")]


@pytest.mark.parametrize("alpha,nit,init", CASES)
def test_hs_exact_mode_matches_reference_bits(ctx, alpha, nit, init):
    import HornSchunck as H
    h = load_gold("hs")
    Z = np.zeros_like(h["U0"])
    Ui, Vi = (h["U0"], h["V0"]) if init == "rand" else (Z, Z)
    ctx.set_hs_fuse(-1)
    try:
        U, V, err = H.HS(h["im0"], h["im1"], alpha, nit, Ui.copy(), Vi.copy())
    finally:
        ctx.set_hs_fuse(0)
    key = "hs_a%g_n%d_%s" % (alpha, nit, init)
    np.testing.assert_array_equal(U, h[key + "_U"]); np.testing.assert_array_equal(V, h[key + "_V"])
    assert err == pytest.approx(float(h[key + "_err"]), rel=1e-5)


@pytest.mark.parametrize("fuse", [0, 1, 16])
@pytest.mark.parametrize("alpha,nit,init", CASES)
def test_hs_fast_mode_within_tolerance(ctx, fuse, alpha, nit, init):
    import HornSchunck as H
    h = load_gold("hs")
    Z = np.zeros_like(h["U0"])
    Ui, Vi = (h["U0"], h["V0"]) if init == "rand" else (Z, Z)
    ctx.set_hs_fuse(fuse)
    try:
        U, V, err = H.HS(h["im0"], h["im1"], alpha, nit, Ui.copy(), Vi.copy())
    finally:
        ctx.set_hs_fuse(0)
    key = "hs_a%g_n%d_%s" % (alpha, nit, init)
    assert flow_rel(U, V, h[key + "_U"], h[key + "_V"]) < 1e-5
    assert err == pytest.approx(float(h[key + "_err"]), rel=1e-4)


@pytest.mark.parametrize("alpha", [1e-3, 0.05, 300.0, 1e5])
def test_hs_fast_mode_over_the_range_of_alpha(ctx, alpha):
    """The fast kernels scale the coefficients by 1/sqrt(alpha^2 + fx^2 + fy^2) once instead of dividing per sweep: the
    tolerance must hold from a vanishing smoothness weight (the denominator is the squared gradient alone, zero where
    the frames are flat) to one that swamps the data term."""
    import HornSchunck as H
    from oracle import hs
    rng = np.random.default_rng(int(alpha * 1000) % 9973)
    a = rng.uniform(0, 255, (96, 160)).astype(np.float32); b = rng.uniform(0, 255, (96, 160)).astype(np.float32)
    a[40:56, 60:100] = 7.0; b[40:56, 60:100] = 7.0        # a flat patch: fx = fy = ft = 0 inside
    z = np.zeros_like(a)
    Ur, Vr, er = hs.hs(a, b, alpha, 25, z.copy(), z.copy())
    for fuse in (0, 1):
        ctx.set_hs_fuse(fuse)
        try:
            U, V, e = H.HS(a, b, alpha, 25, z.copy(), z.copy())
        finally:
            ctx.set_hs_fuse(0)
        assert np.isfinite(U).all() and np.isfinite(V).all()
        assert flow_rel(U, V, Ur, Vr) < 1e-5, (alpha, fuse)


def test_hs_adapter_semantics(ctx):
    from HornSchunck import HSOpticalFlowAlgoAdapter
    h = load_gold("hs")
    ad = HSOpticalFlowAlgoAdapter([3.0, 9.0], 25)
    z = np.zeros_like(h["adapter_crop_im0"])
    U, V, err = ad.compute(h["adapter_crop_im0"], h["adapter_crop_im1"], z, z.copy())
    assert ad.alphas == [3.0]
    assert U.dtype == np.float32 and U is not z
    assert flow_rel(U, V, h["adapter_crop_U"], h["adapter_crop_V"]) < 1e-5
    assert err == pytest.approx(float(h["adapter_crop_err"]), rel=1e-4)
    ad.compute(h["adapter_crop_im0"], h["adapter_crop_im1"], z, z)
    with pytest.raises(IndexError):
        ad.compute(h["adapter_crop_im0"], h["adapter_crop_im1"], z, z)


@pytest.mark.parametrize("shape", [(1, 1), (1, 9), (9, 1), (2, 2), (3, 5), (17, 260), (64, 256), (33, 1028), (130, 516)])
def test_hs_ragged_shapes_vs_oracle(ctx, shape):
    import HornSchunck as H
    from oracle import hs
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    a = rng.uniform(0, 255, shape).astype(np.float32); b = rng.uniform(0, 255, shape).astype(np.float32)
    U0 = rng.standard_normal(shape).astype(np.float32); V0 = rng.standard_normal(shape).astype(np.float32)
    Ur, Vr, er = hs.hs(a, b, 7.5, 13, U0.copy(), V0.copy())
    for fuse, tol in ((-1, 0.0), (1, 1e-5), (0, 1e-5), (16, 1e-5)):
        ctx.set_hs_fuse(fuse)
        try:
            U, V, e = H.HS(a, b, 7.5, 13, U0.copy(), V0.copy())
        finally:
            ctx.set_hs_fuse(0)
        if tol == 0.0:
            np.testing.assert_array_equal(U, Ur); np.testing.assert_array_equal(V, Vr)
        else:
            assert flow_rel(U, V, Ur, Vr) <= tol, (fuse, shape)


@pytest.mark.parametrize("shape,nit", [((1, 8), 5), ((5, 12), 16), ((40, 200), 33), ((300, 100), 17), ((257, 1000), 48),
                                       ((700, 96), 100), ((64, 2052), 20),
                                       # widths that are not multiples of 4: the tile kernel runs on padded copies
                                       ((16, 17), 9), ((40, 201), 33), ((300, 99), 17), ((130, 1023), 20), ((64, 2050), 12),
                                       ((113, 114), 8), ((48, 115), 24)])
def test_hs_launch_shapes_agree_bitwise(ctx, shape, nit):
    """One sweep per launch, the tile kernel (8 sweeps per launch) and the row-streaming pipeline
    (16 per launch) share one arithmetic sequence: their results are the same bits."""
    from HornSchunck import hs_iterate_device
    rng = np.random.default_rng(shape[0] * 7 + shape[1] + nit)
    host = [rng.standard_normal(shape).astype(np.float32) for _ in range(5)]
    out = {}
    for fuse in (1, 8, 16, 32):                  # 16 / 32: the pipeline with strips of 128 / 256 columns
        ctx.set_hs_fuse(fuse)
        try:
            d = [ctx.upload(x) for x in host]
            hs_iterate_device(d[0], d[1], d[2], d[3], d[4], 3.0, nit)
            out[fuse] = (d[0].numpy(), d[1].numpy())
        finally:
            ctx.set_hs_fuse(0)
    for fuse in (8, 16, 32):
        np.testing.assert_array_equal(out[fuse][0], out[1][0]); np.testing.assert_array_equal(out[fuse][1], out[1][1])



@pytest.mark.parametrize("nit", [24, 25, 27, 28, 29, 40, 600])
def test_hs_pipeline_tail_lengths_agree_bitwise(ctx, nit):
    """On fields large enough for the row-streaming pipeline the automatic mode splits the sweeps into launches of 12,
    a deeper last launch when 13 to 16 are left, and the tile kernel for other remainders: every split gives the bits of
    one sweep per launch.  600: the iteration count of the reference's own scripts
    (examples/PyHSchunck_Fs3_4_PyrLvls2.py:132), fifty launches of 12."""
    from HornSchunck import hs_iterate_device
    shape = (2048, 2052)
    rng = np.random.default_rng(nit)
    host = [rng.standard_normal(shape).astype(np.float32) for _ in range(5)]
    out = {}
    for fuse in (1, 0):
        ctx.set_hs_fuse(fuse)
        try:
            d = [ctx.upload(x) for x in host]
            hs_iterate_device(d[0], d[1], d[2], d[3], d[4], 3.0, nit)
            out[fuse] = (d[0].numpy(), d[1].numpy())
        finally:
            ctx.set_hs_fuse(0)
    assert out[0][0].tobytes() == out[1][0].tobytes() and out[0][1].tobytes() == out[1][1].tobytes()

@pytest.mark.parametrize("shape,nit,zero", [((64, 96), 20, False), ((2048, 2048), 28, True), ((2048, 2052), 28, False),
                                            ((2048, 2048), 24, True), ((2048, 2048), 36, False)])
def test_hs_accumulating_step_equals_step_plus_add(ctx, shape, nit, zero):
    """ofri_hs_compute_acc (the driver's Uaccum += U folded into the last launch's stores on large fields, a
    separate add elsewhere) gives the bits of ofri_hs_compute followed by the float32 addition."""
    from HornSchunck import HSOpticalFlowAlgoAdapter, hs_device
    from piv_synth import synth_piv
    a, b = synth_piv(shape[0], shape[1], seed=3)
    rng = np.random.default_rng(nit)
    acc = [rng.standard_normal(shape).astype(np.float32) for _ in range(2)]
    start = [None, None] if zero else [ctx.upload(rng.standard_normal(shape).astype(np.float32) * 0.1) for _ in range(2)]
    fa, fb = ctx.upload(a.astype(np.float32)), ctx.upload(b.astype(np.float32))
    Un, Vn, _ = hs_device(fa, fb, 21.0, nit, start[0], start[1], want_error=False)
    want = (acc[0] + Un.numpy(), acc[1] + Vn.numpy())
    Ua, Va = ctx.upload(acc[0]), ctx.upload(acc[1])
    HSOpticalFlowAlgoAdapter([21.0], nit).accumulate_device(fa, fb, start[0], start[1], Ua, Va)
    np.testing.assert_array_equal(Ua.numpy(), want[0]); np.testing.assert_array_equal(Va.numpy(), want[1])


@pytest.mark.parametrize("shape,nit,zero", [((700, 1000), 24, True), ((1030, 2052), 28, False), ((300, 520), 12, False)])
def test_hs_accumulating_store_in_wide_strips(ctx, shape, nit, zero):
    """The accumulating store of the pipeline's 256-column strips (the launch shape of 4096 x 4096 fields, forced here on
    smaller ones, ragged against the strips and chunks): the bits of one sweep per launch followed by the float32 addition."""
    from HornSchunck import HSOpticalFlowAlgoAdapter, hs_device
    from piv_synth import synth_piv
    a, b = synth_piv(shape[0], shape[1], seed=4)
    rng = np.random.default_rng(nit + shape[0])
    acc = [rng.standard_normal(shape).astype(np.float32) for _ in range(2)]
    start = [None, None] if zero else [ctx.upload(rng.standard_normal(shape).astype(np.float32) * 0.1) for _ in range(2)]
    fa, fb = ctx.upload(a.astype(np.float32)), ctx.upload(b.astype(np.float32))
    ctx.set_hs_fuse(1)
    try:
        Un, Vn, _ = hs_device(fa, fb, 21.0, nit, start[0], start[1], want_error=False)
    finally:
        ctx.set_hs_fuse(0)
    want = (acc[0] + Un.numpy(), acc[1] + Vn.numpy())
    Ua, Va = ctx.upload(acc[0]), ctx.upload(acc[1])
    ctx.set_hs_fuse(32)
    try:
        HSOpticalFlowAlgoAdapter([21.0], nit).accumulate_device(fa, fb, start[0], start[1], Ua, Va)
    finally:
        ctx.set_hs_fuse(0)
    np.testing.assert_array_equal(Ua.numpy(), want[0]); np.testing.assert_array_equal(Va.numpy(), want[1])


@pytest.mark.parametrize("shape,ksize,warp,nit", [((2048, 2048), 3, True, 28), ((2048, 2048), 3, False, 28), ((2050, 2052), 5, True, 16),
                                                   ((2049, 2052), 7, True, 28), ((2049, 2052), 7, False, 24), ((96, 128), 3, True, 20),
                                                   ((2048, 2048), 9, True, 28)])
def test_hs_whole_level_equals_its_steps(ctx, shape, ksize, warp, nit):
    """ofri_hs_level (warp, pre-filter, derivatives in one pass on large fields) against ofri_warp_pair,
    ofri_gauss_sep twice, ofri_hs_compute and the float32 accumulation one after the other: same bits, with a
    flow large enough to push samples over every edge of the image."""
    import _ofri
    from HornSchunck import HSOpticalFlowAlgoAdapter, hs_device
    from gaussian_filter import prepareGaussianKernel
    from piv_synth import synth_piv
    lib = _ofri.lib()
    H, W = shape
    a, b = synth_piv(H, W, seed=5)
    rng = np.random.default_rng(ksize + nit)
    flow = [(rng.standard_normal(shape) * 3.0).astype(np.float32) for _ in range(2)]
    taps = prepareGaussianKernel(3.4, ksize)
    da, db = ctx.upload(a.astype(np.float32)), ctx.upload(b.astype(np.float32))
    # the steps one after the other
    if warp:
        U0, V0 = ctx.upload(flow[0]), ctx.upload(flow[1])
        w1, w2 = ctx.empty(shape), ctx.empty(shape)
        _ofri.check(lib.ofri_warp_pair(ctx.handle, da.ptr, db.ptr, U0.ptr, V0.ptr, w1.ptr, w2.ptr, H, W))
    else:
        w1, w2 = da, db
    f1, f2 = ctx.empty(shape), ctx.empty(shape)
    for src, dst in ((w1, f1), (w2, f2)):
        _ofri.check(lib.ofri_gauss_sep(ctx.handle, src.ptr, dst.ptr, H, W, taps.ctypes.data, int(taps.size)))
    Un, Vn, _ = hs_device(f1, f2, 21.0, nit, None, None, want_error=False)
    want = (flow[0] + Un.numpy(), flow[1] + Vn.numpy()) if warp else (Un.numpy(), Vn.numpy())
    # the whole level
    Uf, Vf = (ctx.upload(flow[0]), ctx.upload(flow[1])) if warp else (ctx.empty(shape), ctx.empty(shape))
    HSOpticalFlowAlgoAdapter([21.0], nit).level_device(da, db, Uf, Vf, taps, warp)
    np.testing.assert_array_equal(Uf.numpy(), want[0]); np.testing.assert_array_equal(Vf.numpy(), want[1])


@pytest.mark.parametrize("shape", [(2048, 2048), (1030, 2052)])
def test_pipelines_give_the_same_bits_when_contexts_share_the_gpu(ctx, shape):
    """ofri_set_share: fewer, longer row chunks in the Horn-Schunck and Liu-Shen pipelines, same results."""
    import _ofri
    from PhysicsBasedOpticalFlowLiuShen import liushen_device
    lib = _ofri.lib()
    H, W = shape
    rng = np.random.default_rng(H + W)
    pl = [rng.standard_normal(shape).astype(np.float32) for _ in range(5)]
    a = rng.uniform(0, 255, shape).astype(np.float32); b = np.roll(a, 1, axis=1)
    out = []
    try:
        for share in (1, 3):
            ctx.set_share(share)
            d = [ctx.upload(p) for p in pl]
            _ofri.check(lib.ofri_hs_iterate(ctx.handle, d[0].ptr, d[1].ptr, d[2].ptr, d[3].ptr, d[4].ptr, 10.0, 40, H, W))
            U, V = liushen_device(ctx.upload(a), ctx.upload(b), 5.0, ctx.upload(pl[0]), ctx.upload(pl[1]), want_error=False)[:2]
            out.append([d[0].numpy(), d[1].numpy(), U.numpy(), V.numpy()])
    finally:
        ctx.set_share(1)
    assert lib.ofri_set_share(ctx.handle, 0) != 0
    for x, y, what in zip(out[0], out[1], ("HS u", "HS v", "LS u", "LS v")):
        assert x.tobytes() == y.tobytes(), what


def test_hs_zero_iterations_returns_input(ctx):
    import HornSchunck as H
    rng = np.random.default_rng(1)
    a = rng.uniform(0, 255, (20, 24)).astype(np.float32)
    U0 = rng.standard_normal((20, 24)).astype(np.float32)
    U, V, e = H.HS(a, a, 5.0, 0, U0, U0 * 2)
    np.testing.assert_array_equal(U, U0); np.testing.assert_array_equal(V, U0 * 2)
    assert e == 0.0


# ------------------------------------------------------------------------------ pyramid glue
def test_resize_vs_pillow_fixtures(ctx, pair):
    import GenericPyramidalOpticalFlow as G
    p = load_gold("pyramid")
    r = G.imresize(pair[0], (256, 256))
    assert np.abs(r[::2, ::2] - p["resize_256_sub"]).max() <= 1.6e-5
    assert np.abs(G.imresize(p["odd"], (156, 100)) - p["resize_odd"]).max() <= 1.6e-5
    assert np.abs(G.imresize(pair[0][:300, :450].copy(), (150, 100)) - p["resize_third"]).max() <= 1.6e-5
    np.testing.assert_array_equal(G.imresize(p["odd"], (313, 201)), p["odd"])          # same size: copy
    from oracle import pyramid
    up = G.imresize(p["odd"][:40, :50].copy(), (77, 93))                                # upscaling path
    assert np.abs(up - pyramid.resize_lib(p["odd"][:40, :50].copy(), (77, 93))).max() <= 3.1e-5


def test_spline_warp_next_level_vs_fixtures(ctx):
    import GenericPyramidalOpticalFlow as G
    p = load_gold("pyramid")
    np.testing.assert_array_equal(G.doBiLinearWarping(p["warp_img"], p["warp_cy"], p["warp_cx"], 1, "nearest"), p["warp_out"])
    r = G.updateNextPyramidalLevel(p["P0"].copy(), np.zeros((128, 128), np.float32), p["P1"].copy(), p["Uc"].copy(),
                                   p["Vc"].copy(), None, None, True, True, True)
    tol = 2e-6 * max(1.0, np.abs(p["up_Uacc"]).max())
    assert np.abs(r[2] - p["up_Uacc"]).max() <= tol and np.abs(r[3] - p["up_Vacc"]).max() <= tol
    assert rel_l2(r[0], p["up_im1"]) < 1e-6 and rel_l2(r[1], p["up_im2"]) < 1e-6
    assert not r[4].any() and not r[5].any()
    r = G.updateNextPyramidalLevel(p["P0"].copy(), np.zeros((128, 128), np.float32), p["P1"].copy(), p["Uc"].copy(),
                                   p["Vc"].copy(), None, None, False, True, False)
    assert np.abs(r[4] - p["nowarp_U"]).max() <= tol and np.abs(r[5] - p["nowarp_V"]).max() <= tol and not r[2].any()
    pg = load_gold("pair_parabolic01")
    r = G.updateNextPyramidalLevel(p["odd"].copy(), np.zeros((100, 156), np.float32), pg["im1"].astype(np.float32)[:201, :313].copy(),
                                   p["Uo"].copy(), p["Vo"].copy(), None, None, True, True, True)
    tol = 2e-6 * max(1.0, np.abs(p["odd_Uacc"]).max())
    assert np.abs(r[2] - p["odd_Uacc"]).max() <= tol and np.abs(r[3] - p["odd_Vacc"]).max() <= tol
    assert rel_l2(r[0], p["odd_im1"]) < 1e-6 and rel_l2(r[1], p["odd_im2"]) < 1e-6


def test_shift_warp_biLinear_false_vs_reference(ctx):
    """The optical-flow-equation warp (reference :189-221): colliding and wrapping destinations, the
    73-tap smoothing, the in-place update of the caller's array; then the reference's IndexError."""
    import GenericPyramidalOpticalFlow as G
    w = load_gold("shiftwarp")
    arg = w["im"].copy()
    r = G.updateNextPyramidalLevel(arg, np.zeros((120, 150), np.float32), w["im2"].copy(), w["U"].copy(), w["V"].copy(),
                                   None, None, True, False, False)
    assert r[0] is arg
    np.testing.assert_array_equal(r[0], w["out_im1"]); np.testing.assert_array_equal(r[1], w["out_im2"])
    np.testing.assert_array_equal(r[2], w["out_Uacc"]); np.testing.assert_array_equal(r[3], w["out_Vacc"])
    assert not r[4].any() and not r[5].any()
    # destinations past the far edge: same exception, same text as NumPy gives the reference
    for U, V in ((np.full((120, 150), 0.6, np.float32), np.zeros((120, 150), np.float32)),
                 (np.zeros((120, 150), np.float32), np.full((120, 150), 2.5, np.float32)),
                 (np.full((120, 150), -151.0, np.float32), np.full((120, 150), 7.5, np.float32))):
        xs, ys = np.meshgrid(np.arange(150, dtype=np.int32), np.arange(120, dtype=np.int32))
        with pytest.raises(IndexError) as want:
            w["im"].copy()[np.int32(ys + np.floor(V + 0.5)), np.int32(xs + np.floor(U + 0.5))] = 0
        with pytest.raises(IndexError) as got:
            G.updateNextPyramidalLevel(w["im"].copy(), np.zeros((120, 150), np.float32), w["im2"].copy(), U, V,
                                       None, None, True, False, False)
        assert str(got.value) == str(want.value)


def test_shift_warp_driver_vs_reference(ctx, pair):
    """Two levels with warping=True, biLinear=False (adapter defaults off).  The integer part of the
    shift is a step function of the flow, so agreement is asserted on all but a few pixels."""
    from HornSchunck import HSOpticalFlowAlgoAdapter as A
    from GenericPyramidalOpticalFlow import genericPyramidalOpticalFlow as flow
    w = load_gold("shiftwarp")
    a0, a1 = pair[0][:256, :256].copy(), pair[1][:256, :256].copy()
    ctx.set_hs_fuse(-1)
    try:
        U, V = flow(a0, a1, 3.4, A([21, 45], 30, False), 2, 1, None, None, True, False, True, False)
    finally:
        ctx.set_hs_fuse(0)
    bad = np.count_nonzero(a0 != w["drv_im1_after"])
    rel = flow_rel(U, V, w["drv_U"], w["drv_V"])
    print("shift-warp driver: rel L2", rel, "pixels of the overwritten image that differ:", bad)
    assert rel < 1e-3 and bad < 0.001 * a0.size
    assert np.count_nonzero(a0 != pair[0][:256, :256]) > 10000        # the caller's image was overwritten


@pytest.mark.parametrize("shape_in,shape_out", [((4, 4), (9, 7)), ((5, 6), (10, 12)), ((4, 70), (8, 141)), ((33, 35), (67, 70)),
                                                ((66, 67), (133, 131)), ((130, 40), (260, 83)), ((64, 64), (300, 100)),
                                                # around the column kernel's tiles (36 x 63 intervals), its row-weight window
                                                # (96 fine rows) and its passes (24 fine rows): factors 2, 5 and below 1
                                                ((37, 64), (74, 128)), ((38, 65), (76, 130)), ((73, 127), (146, 254)),
                                                ((74, 128), (148, 256)), ((45, 70), (225, 77)), ((150, 200), (60, 500)),
                                                ((200, 129), (401, 259))])
def test_spline_upsample_small_and_ragged(ctx, shape_in, shape_out):
    """FITPACK's interpolating spline (RectBivariateSpline, reference :152-162) on sizes around the solver's
    chunk boundaries, the smallest legal grid, and factors other than 2."""
    import GenericPyramidalOpticalFlow as G
    from oracle import pyramid
    rng = np.random.default_rng(shape_in[0] * 100 + shape_in[1])
    F = rng.standard_normal(shape_in).astype(np.float32).cumsum(0).cumsum(1).astype(np.float32) * 0.1
    want = pyramid.spline_upsample_lib(F, shape_out)
    got = G.spline_upsample_device(ctx.upload(F), shape_out).numpy()
    assert got.shape == tuple(shape_out)
    assert np.abs(got - want).max() <= 2e-6 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("shape_in,shape_out", [((4, 4), (8, 8)), ((38, 65), (76, 130)), ((45, 70), (225, 77)), ((150, 200), (60, 500)),
                                                ((600, 700), (1200, 1400))])
def test_pair_entry_points_equal_two_single_calls(ctx, shape_in, shape_out):
    """ofri_spline_upsample_pair and ofri_resize_bicubic_pair (both components / both frames in one launch) give the bits of
    two calls of the single-field entry points, on ragged shapes, other factors and sizes past one tile."""
    import _ofri
    import GenericPyramidalOpticalFlow as G
    lib = _ofri.lib()
    rng = np.random.default_rng(shape_in[0] + 7 * shape_in[1])
    a = rng.standard_normal(shape_in).astype(np.float32).cumsum(1).astype(np.float32)
    b = rng.standard_normal(shape_in).astype(np.float32).cumsum(0).astype(np.float32)
    da, db = ctx.upload(a), ctx.upload(b)
    ua, ub = ctx.empty(shape_out), ctx.empty(shape_out)
    _ofri.check(lib.ofri_spline_upsample_pair(ctx.handle, da.ptr, db.ptr, shape_in[0], shape_in[1], ua.ptr, ub.ptr, shape_out[0], shape_out[1], 2.0, 1.0))
    np.testing.assert_array_equal(ua.numpy(), G.spline_upsample_device(da, shape_out, 2.0).numpy())
    np.testing.assert_array_equal(ub.numpy(), G.spline_upsample_device(db, shape_out, 1.0).numpy())
    for shp in (shape_out, (max(1, shape_in[0] // 2), max(1, shape_in[1] // 2)), (max(1, shape_in[0] // 3), shape_in[1])):
        ra, rb = G.resize_pair_device(da, db, (shp[1], shp[0]))
        np.testing.assert_array_equal(ra.numpy(), G.resize_device(da, (shp[1], shp[0])).numpy())
        np.testing.assert_array_equal(rb.numpy(), G.resize_device(db, (shp[1], shp[0])).numpy())
    # aliased destinations are refused
    assert lib.ofri_spline_upsample_pair(ctx.handle, da.ptr, db.ptr, shape_in[0], shape_in[1], ua.ptr, ua.ptr, shape_out[0], shape_out[1], 1.0, 1.0) != 0
    assert lib.ofri_resize_bicubic_pair(ctx.handle, da.ptr, db.ptr, shape_in[0], shape_in[1], ua.ptr, ua.ptr, shape_out[0], shape_out[1]) != 0


# ------------------------------------------------------------------------------ whole driver
def _driver_case(name, U, V):
    d = load_gold("driver")
    s = int(d[name + "_sub"])
    rel = flow_rel(U[::s, ::s], V[::s, ::s], d[name + "_U"], d[name + "_V"])
    assert rel < 1e-4, (name, rel)
    assert float(np.linalg.norm(U.astype(np.float64))) == pytest.approx(d[name + "_Ustats"][1], rel=1e-4)
    return rel


def test_driver_cfg1_and_pyramids(ctx, pair):
    from HornSchunck import HSOpticalFlowAlgoAdapter as A
    from GenericPyramidalOpticalFlow import genericPyramidalOpticalFlow as flow
    I0, I1 = pair
    r = {}
    r["hs_L1_a21"] = _driver_case("hs_L1_a21", *flow(I0, I1, 3.4, A([21], 100), 1, 1, None))
    r["hs_L1_a1"] = _driver_case("hs_L1_a1", *flow(I0, I1, 3.4, A([1.0], 100), 1, 1, None))
    r["hs_L2"] = _driver_case("hs_L2", *flow(I0, I1, 3.4, A([21, 45], 100), 2, 1, None))
    r["hs_L3_nofilter"] = _driver_case("hs_L3_nofilter", *flow(I0, I1, 0.0, A([21, 45, 60], 30), 3, 1, None))
    r["hs_L2_k2"] = _driver_case("hs_L2_k2", *flow(I0[:256, :256].copy(), I1[:256, :256].copy(), 3.4, A([21, 21, 45, 45], 20), 2, 2, None))
    r["hs_L2_odd"] = _driver_case("hs_L2_odd", *flow(I0[:201, :313].copy(), I1[:201, :313].copy(), 2.0, A([21, 45], 40), 2, 1, None))
    print("driver rel L2 vs reference:", r)


def test_driver_accepts_foreign_adapter(ctx, pair):
    """Any object with the reference's adapter protocol works (here: the CPU oracle's adapter)."""
    from GenericPyramidalOpticalFlow import genericPyramidalOpticalFlow as flow
    from GenericPyramidalOpticalFlowWrapper import GenericPyramidalOpticalFlowWrapper as Wrap
    from oracle import hs
    I0, I1 = pair[0][:128, :128].copy(), pair[1][:128, :128].copy()
    from oracle import pyramid
    Ur, Vr = pyramid.pyramidal_flow(I0.copy(), I1.copy(), 3.4, hs.HSAdapter([21, 45], 10), 2, 1, None)
    U, V = flow(I0, I1, 3.4, hs.HSAdapter([21, 45], 10), 2, 1, None)
    assert flow_rel(U, V, Ur, Vr) < 1e-5
    U2, V2 = Wrap(hs.HSAdapter([21, 45], 10), filter_sigma=3.4, pyr_levels=2).calculateFlow(I0, I1)
    np.testing.assert_array_equal(U, U2)
    # the wrapper on frames that are already on the GPU
    from HornSchunck import HSOpticalFlowAlgoAdapter
    w = Wrap(HSOpticalFlowAlgoAdapter([21, 45, 21, 45], 10), filter_sigma=3.4, pyr_levels=2)
    assert (w.pyr_levels, w.k_levels, w.filter_opt, w.warping, w.bi_linear) == (2, 1, None, True, True)
    Uh, Vh = w.calculateFlow(I0, I1)
    Ud, Vd = w.calculateFlowDevice(ctx.upload(I0), ctx.upload(I1))
    np.testing.assert_array_equal(Ud.numpy(), Uh); np.testing.assert_array_equal(Vd.numpy(), Vh)


# ------------------------------------------------------------------------------ full-size properties
def test_full_size_properties_4096(ctx):
    """At the benchmark size the oracle is too slow for 100 sweeps, so check (a) three sweeps against
    the oracle, (b) launch-shape independence: fused sweeps == one sweep per launch bit for bit,
    (c) mirror symmetry: flipping every input left-right / up-down flips the output exactly."""
    import _ofri
    from piv_synth import synth_piv
    from oracle import hs
    import gaussian_filter as gf
    lib = _ofri.lib()
    a, b = synth_piv(4096, 4096, seed=0)
    a, b = a.astype(np.float32), b.astype(np.float32)

    def run(A, B, nit, fuse):
        ctx.set_hs_fuse(fuse)
        try:
            A_d, B_d = ctx.upload(A), ctx.upload(B)
            taps = gf.prepareGaussianKernel(3.4, 3)
            A_d, B_d = gf.filter_device(A_d, taps), gf.filter_device(B_d, taps)
            from HornSchunck import hs_device
            z = ctx.zeros(A.shape)
            U, V, e = hs_device(A_d, B_d, 21.0, nit, z, z, True)
            return U.numpy(), V.numpy(), A_d.numpy(), B_d.numpy()
        finally:
            ctx.set_hs_fuse(0)

    U1, V1, Af, Bf = run(a, b, 3, 1)
    Ur, Vr, _ = hs.hs(Af, Bf, 21.0, 3, np.zeros_like(Af), np.zeros_like(Af))
    assert flow_rel(U1, V1, Ur, Vr) < 1e-5
    Ua, Va, _, _ = run(a, b, 100, 0)
    Ub, Vb, _, _ = run(a, b, 100, 1)
    np.testing.assert_array_equal(Ua, Ub); np.testing.assert_array_equal(Va, Vb)
    assert np.isfinite(Ua).all() and np.abs(Ua).max() < 50
    # the flow of the centre rows points in -x with the Poiseuille magnitude's sign
    assert Ua[1800:2300].mean() < -0.5
    # (c) on the sweep itself: U, V, fx, fy, ft flipped together give the flipped result exactly
    from HornSchunck import hs_iterate_device
    rng = np.random.default_rng(7)
    F = [rng.standard_normal((1024, 4096)).astype(np.float32) for _ in range(5)]
    outs = []
    for flip in (lambda x: x, lambda x: x[:, ::-1], lambda x: x[::-1, :]):
        d = [ctx.upload(np.ascontiguousarray(flip(f))) for f in F]
        hs_iterate_device(d[0], d[1], d[2], d[3], d[4], 3.0, 17)
        outs.append((flip(d[0].numpy()), flip(d[1].numpy())))
    for k in (1, 2):
        np.testing.assert_array_equal(outs[0][0], outs[k][0]); np.testing.assert_array_equal(outs[0][1], outs[k][1])


# ------------------------------------------------------------------------------ Liu-Shen
def test_liushen_raw_and_adapter_vs_reference(ctx):
    import PhysicsBasedOpticalFlowLiuShen as L
    from oracle import gauss
    l = load_gold("liushen")
    pg = load_gold("pair_parabolic01")
    I0o = gauss.filter_px(pg["im0"].astype(np.float32), 0.48, 5)[:256, :256].copy()
    I1o = gauss.filter_px(pg["im1"].astype(np.float32), 0.48, 5)[:256, :256].copy()
    z = np.zeros((256, 256), np.float32)
    u, v, err = L.physicsBasedOpticalFlowLiuShen(I0o, I1o, 5, z, z.copy())
    r = flow_rel(u, v, l["raw_h5_zero_u"], l["raw_h5_zero_v"])
    print("liushen raw rel L2", r, "max abs", np.abs(u - l["raw_h5_zero_u"]).max())
    assert r < 1e-5
    assert err == pytest.approx(float(l["raw_h5_zero_err"]), rel=1e-3)
    a, b, e, it = L.liushen_device(ctx.upload(I0o), ctx.upload(I1o), 5, ctx.upload(z), ctx.upload(z), True)
    assert it == str(l["raw_h5_zero_log"]).count("Iteration:")
    # adapter: swaps the components in and out, ragged 57 x 71 crop, random initial flow
    U, V, err = L.LiuShenOpticalFlowAlgoAdapter(10).compute(l["crop_im0"].copy(), l["crop_im1"].copy(), l["crop_Uin"].copy(), l["crop_Vin"].copy())
    assert flow_rel(U, V, l["crop_U"], l["crop_V"]) < 1e-5
    assert err == pytest.approx(float(l["crop_err"]), rel=1e-3)


def test_liushen_stopping_rule_and_tiny_shapes(ctx):
    import PhysicsBasedOpticalFlowLiuShen as L
    from oracle import liushen
    rng = np.random.default_rng(3)
    # a flat image pair converges at once: coefficient fields vanish and u, v collapse to 0
    for shape in ((1, 1), (1, 6), (7, 1), (5, 9), (66, 130)):
        a = (rng.uniform(1, 255, shape)).astype(np.float32); b = (rng.uniform(1, 255, shape)).astype(np.float32)
        U0 = (0.1 * rng.standard_normal(shape)).astype(np.float32); V0 = (0.1 * rng.standard_normal(shape)).astype(np.float32)
        ur, vr, er, itr = liushen.liushen(a, b, 50.0, U0.copy(), V0.copy(), return_iters=True)
        ud, vd, ed, itd = L.liushen_device(ctx.upload(a), ctx.upload(b), 50.0, ctx.upload(U0), ctx.upload(V0), True)
        assert itd == itr, shape
        assert flow_rel(ud.numpy(), vd.numpy(), ur, vr) < 1e-5, shape
    flat = np.full((32, 48), 7.0, np.float32)
    ud, vd, ed, itd = L.liushen_device(ctx.upload(flat), ctx.upload(flat), 5.0, ctx.zeros(flat.shape), ctx.zeros(flat.shape), True)
    ur, vr, er, itr = liushen.liushen(flat, flat, 5.0, np.zeros_like(flat), np.zeros_like(flat), return_iters=True)
    assert itd == itr and itd < 60
    np.testing.assert_array_equal(ud.numpy(), ur)


@pytest.mark.parametrize("big,mode", [(False, 0), (True, 0), (False, -1), (False, 1), (False, 6), (True, 6), (False, 8)])
def test_liushen_stops_at_every_iteration_count(ctx, pair, big, mode):
    """The stopping check of a block of sweeps runs inside the next launch, which has then already started: whatever the
    count at which the rule is met (odd, even, inside a launch of several sweeps, the last allowed one), the result
    must be that sweep's, bit for bit, and the count the reference's loop would report."""
    import ctypes as C
    import _ofri
    lib = _ofri.lib()
    if big:
        I0, I1 = np.tile(pair[0], (2, 2))[:1000, :1032].copy(), np.tile(pair[1], (2, 2))[:1000, :1032].copy()
    else:
        I0, I1 = pair[0][100:230, 60:260].copy(), pair[1][100:230, 60:260].copy()
    rows, cols = I0.shape
    a, b, z = ctx.upload(I0), ctx.upload(I1), ctx.zeros(I0.shape)

    def run(maxit, tol):
        u, v = ctx.empty(I0.shape), ctx.empty(I0.shape)
        err, it = C.c_double(0.0), C.c_int(0)
        _ofri.check(lib.ofri_ls_compute(ctx.handle, a.ptr, b.ptr, z.ptr, z.ptr, u.ptr, v.ptr, 5.0, maxit, tol, rows, cols,
                                        C.byref(err), C.byref(it)))
        return u.numpy(), v.numpy(), err.value, it.value

    ctx.set_ls_mode(mode)
    try:
        fixed = {k: run(k, 0.0) for k in range(0, 16)}            # tol = 0: exactly k sweeps
        assert [fixed[k][3] for k in range(16)] == list(range(16))
        np.testing.assert_array_equal(fixed[0][0], 0.0)
        errs = [fixed[k][2] for k in range(1, 16)]
        assert all(e1 > e2 > 0 for e1, e2 in zip(errs, errs[1:])), errs    # the change per sweep shrinks on this pair
        for k in range(1, 15):
            tol = 0.5 * (fixed[k][2] + fixed[k + 1][2])           # err_k > tol > err_(k+1): first met by sweep k+1
            u, v, e, it = run(60, tol)
            assert it == k + 1 and e == fixed[k + 1][2]
            np.testing.assert_array_equal(u, fixed[k + 1][0]); np.testing.assert_array_equal(v, fixed[k + 1][1])
            u, v, e, it = run(k, tol)                             # the cap comes first
            assert it == k
            np.testing.assert_array_equal(u, fixed[k][0])
    finally:
        ctx.set_ls_mode(0)


@pytest.mark.parametrize("shape", [(130, 200), (256, 256), (1000, 1032), (64, 8), (3, 128), (1, 16), (257, 124), (40, 1024)])
def test_liushen_launch_shapes_agree_bitwise(ctx, shape):
    """One sweep per launch, the 6-deep and the 8-deep pipeline run the same float32 arithmetic: same bits and counts for
    any field shape (image edges inside a strip, strips wider than the image, fewer rows than the pipeline is deep) and
    any number of sweeps; the exact mode (the reference's bits) lies within 1e-5 of them."""
    from oracle import gauss, liushen
    from PhysicsBasedOpticalFlowLiuShen import liushen_device
    from piv_synth import synth_piv
    rng = np.random.default_rng(shape[0])
    H, W = shape
    f0, f1 = synth_piv(max(H, 8), max(W, 8), seed=H + W)
    I0 = gauss.filter_px(f0.astype(np.float32), 0.48, 5)[:H, :W].copy(); I1 = gauss.filter_px(f1.astype(np.float32), 0.48, 5)[:H, :W].copy()
    I0[0, 0] += 1.0                                               # never an all-zero crop
    U0 = (0.3 * rng.standard_normal(shape)).astype(np.float32); V0 = (0.3 * rng.standard_normal(shape)).astype(np.float32)
    d = [ctx.upload(x) for x in (I0, I1, U0, V0)]
    try:
        for maxit in (1, 5, 6, 7, 9, 13, 17, 60):
            res = {}
            for mode in (1, 6, 8, -1):
                ctx.set_ls_mode(mode)
                u, v, e, it = liushen_device(d[0], d[1], 5.0, d[2], d[3], True)[:4] if maxit == 60 else _ls_run(ctx, d, maxit)
                res[mode] = (u.numpy(), v.numpy(), e, it)
            for mode in (6, 8):
                np.testing.assert_array_equal(res[mode][0], res[1][0], err_msg="u, depth %d, %d sweeps" % (mode, maxit))
                np.testing.assert_array_equal(res[mode][1], res[1][1], err_msg="v, depth %d, %d sweeps" % (mode, maxit))
                assert res[mode][3] == res[1][3]
                assert res[mode][2] == pytest.approx(res[1][2], rel=1e-5)
            assert res[-1][3] == res[1][3]
            assert flow_rel(res[1][0], res[1][1], res[-1][0], res[-1][1]) < 1e-5, maxit
        ur, vr, er, itr = liushen.liushen(I0, I1, 5.0, U0.copy(), V0.copy(), return_iters=True)
        assert itr == res[-1][3]
        np.testing.assert_array_equal(res[-1][0], ur)             # exact mode: the oracle's bits
        np.testing.assert_array_equal(res[-1][1], vr)
    finally:
        ctx.set_ls_mode(0)


def _ls_run(ctx, d, maxit, tol=1e-8):
    import ctypes as C
    import _ofri
    rows, cols = d[0].shape
    u, v = ctx.empty((rows, cols)), ctx.empty((rows, cols))
    err, it = C.c_double(0.0), C.c_int(0)
    _ofri.check(_ofri.lib().ofri_ls_compute(ctx.handle, d[0].ptr, d[1].ptr, d[2].ptr, d[3].ptr, u.ptr, v.ptr, 5.0, maxit, tol,
                                            rows, cols, C.byref(err), C.byref(it)))
    return u, v, err.value, it.value


def test_driver_hs_plus_liushen_and_ls_main(ctx, pair):
    from HornSchunck import HSOpticalFlowAlgoAdapter as A
    from PhysicsBasedOpticalFlowLiuShen import LiuShenOpticalFlowAlgoAdapter as LS
    from GenericPyramidalOpticalFlow import genericPyramidalOpticalFlow as flow
    I0, I1 = pair
    r1 = _driver_case("hs_L2_ls", *flow(I0, I1, 3.4, A([21, 45], 100), 2, 1, 0.48, LS(5)))
    r2 = _driver_case("ls_main_L2", *flow(I0[:256, :256].copy(), I1[:256, :256].copy(), 3.4, LS(0.1), 2, 1, None, None, True, True, True, True))
    print("driver rel L2 vs reference: hs_L2_ls", r1, "ls_main_L2", r2)


# ------------------------------------------------------------------------------ Farneback
# The reference's OpenCL kernels cannot be executed where the fixtures were made (no OpenCL device), so
# these compare against the CPU restatement of the kernel text: parity unpinned by reference outputs.
def _fb_inputs(shape, seed):
    rng = np.random.default_rng(seed)
    from piv_synth import synth_piv
    a, b = synth_piv(shape[0], shape[1], seed=seed)
    return a.astype(np.float32), b.astype(np.float32), rng


@pytest.fixture(params=[-1, 0], ids=["exact", "fast"])
def fb_mode(request, ctx):
    """Farneback arithmetic: -1 = separately rounded operations (the oracle's bits), 0 = fused multiply-adds (default)."""
    ctx.set_fb_mode(request.param)
    yield request.param
    ctx.set_fb_mode(0)


def _fb_same(got, want, mode, what):
    """Exact mode: the oracle's bits.  Fast mode: within 1e-5 of the array's scale, element by element and in L2."""
    if mode == -1:
        np.testing.assert_array_equal(got, want, err_msg=what)
    else:
        scale = float(np.abs(want).max()) + 1e-30
        assert float(np.abs(got.astype(np.float64) - want).max()) <= 1e-5 * scale, what
        assert rel_l2(got, want) < 1e-5, what


@pytest.mark.parametrize("shape", [(64, 64), (37, 91), (130, 70)])
def test_farneback_kernels_vs_oracle(ctx, shape, fb_mode):
    from Farneback_PyCL import Farneback_PyCL
    from oracle import farneback as fb
    a, b, rng = _fb_inputs(shape, 11)
    H, W = shape
    same = lambda got, want, what: _fb_same(got, want, fb_mode, what)
    for polyN, sig in ((7, 1.5), (5, 1.1)):
        ad = Farneback_PyCL(polyN=polyN, polySigma=sig)
        ad.setPolynomialExpansionConsts()
        g, xg, xxg, i0, i1, i2, i3 = fb.prepare_gaussian(polyN, sig)
        ig = np.array([i0, i1, i2, i3]).astype(np.float32)
        RA = ad.polynomialExpansion(a, np.zeros((5 * H, W), np.float32))
        same(RA, fb.poly_expansion(a, polyN, g[polyN:], xg[polyN:], xxg[polyN:], ig), "polynomialExpansion")
    RB = ad.polynomialExpansion(b, np.zeros((5 * H, W), np.float32))
    RAo = fb.poly_expansion(a, polyN, g[polyN:], xg[polyN:], xxg[polyN:], ig); RBo = fb.poly_expansion(b, polyN, g[polyN:], xg[polyN:], xxg[polyN:], ig)
    # single-plane blur with the 3-tap and a wide kernel (wider than the image for the small shapes)
    for n in (3, 9, 33):
        ad.setGaussianBlurKernel(n, 0)
        want = fb.gaussian_blur(a, fb.blur_taps(n, 0), n // 2)
        same(ad.gaussianBlur(a, n / 2, np.zeros_like(a)), want, "gaussianBlur %d" % n)
    fx = (2.0 * rng.standard_normal(shape)).astype(np.float32); fy = (2.0 * rng.standard_normal(shape)).astype(np.float32)
    fx[0, 0] = -50.0; fy[-1, -1] = 1e6                       # far outside: fallback branch
    # every kernel is fed the oracle's version of its input, so that a difference is this kernel's own
    Mo = fb.update_matrices(fx, fy, RAo, RBo)
    same(ad.updateMatrices(fx, fy, RAo, RBo, np.zeros((5 * H, W), np.float32)), Mo, "updateMatrices")
    ad.setGaussianBlurKernel(33, 33 / 2 * 0.3)
    Mbo = fb.gaussian_blur5(Mo, fb.blur_taps(33, 4.95), 16)
    same(ad.gaussianBlur5(Mo, 33 / 2, np.zeros_like(Mo)), Mbo, "gaussianBlur5")
    same(ad.boxFilter5(Mo, 33 / 2, np.zeros_like(Mo)), fb.box_filter5(Mo, 16), "boxFilter5")
    ux, uy = ad.updateFlow(Mbo, np.zeros_like(a), np.zeros_like(a))
    uxo, uyo = fb.update_flow(Mbo)
    same(ux, uxo, "updateFlow x"); same(uy, uyo, "updateFlow y")


@pytest.mark.parametrize("shape", [(1, 1), (3, 5), (16, 64), (17, 65), (70, 130), (5, 300), (200, 9), (128, 256), (40, 68), (33, 1028)])
@pytest.mark.parametrize("polyN,ksh", [(7, 1), (5, 1), (7, 2)])
def test_farneback_blur_polyexp_equals_its_steps(ctx, shape, polyN, ksh, fb_mode):
    """Pre-blur and polynomial expansion in one call: the bits of ofri_fb_blur followed by ofri_fb_polyexp."""
    import _ofri
    from Farneback_PyCL import Farneback_PyCL
    lib = _ofri.lib()
    H, W = shape
    rng = np.random.default_rng(H * 17 + W + polyN)
    img = ctx.upload(rng.uniform(0, 255, shape).astype(np.float32))
    fb = Farneback_PyCL(polyN=polyN)
    fb.setPolynomialExpansionConsts()
    fb.setGaussianBlurKernel(2 * ksh + 1, 0.0 if ksh == 1 else 1.1)
    want = fb._polyexp(fb._blur(img, ksh)).numpy()
    got = fb._blur_polyexp(img, ksh).numpy()
    assert got.tobytes() == want.tobytes()


@pytest.mark.parametrize("shape,ksh,box", [((70, 130), 16, 0), ((33, 64), 16, 1), ((40, 100), 7, 0), ((128, 256), 16, 0)])
def test_farneback_fused_iteration_equals_its_steps(ctx, shape, ksh, box, fb_mode):
    """Window filter + updateFlow + updateMatrices in one call: the same bits as the three steps, with and without
    the displacement stored."""
    import _ofri
    from GaussianKernelBitExact import getGaussianKernelBitExact
    lib = _ofri.lib()
    H, W = shape
    rng = np.random.default_rng(H * 31 + W + ksh)
    M = ctx.upload((rng.standard_normal((5 * H, W)) * 3).astype(np.float32))
    R0 = ctx.upload(rng.standard_normal((5 * H, W)).astype(np.float32)); R1 = ctx.upload(rng.standard_normal((5 * H, W)).astype(np.float32))
    taps = np.ascontiguousarray(getGaussianKernelBitExact(2 * ksh + 1, 0.3 * ksh)[1][ksh:], dtype=np.float32)
    assert taps.shape == (ksh + 1,)
    fx, fy, Mref = ctx.empty(shape), ctx.empty(shape), ctx.empty((5 * H, W))
    _ofri.check(lib.ofri_fb_window5_update_flow(ctx.handle, M.ptr, fx.ptr, fy.ptr, H, W, taps.ctypes.data, ksh, box))
    _ofri.check(lib.ofri_fb_update_matrices(ctx.handle, fx.ptr, fy.ptr, R0.ptr, R1.ptr, Mref.ptr, H, W))
    gx, gy, Mn = ctx.zeros(shape), ctx.zeros(shape), ctx.zeros((5 * H, W))
    _ofri.check(lib.ofri_fb_window5_flow_matrices(ctx.handle, M.ptr, gx.ptr, gy.ptr, R0.ptr, R1.ptr, Mn.ptr, H, W, taps.ctypes.data, ksh, box))
    assert gx.numpy().tobytes() == fx.numpy().tobytes() and gy.numpy().tobytes() == fy.numpy().tobytes()
    assert Mn.numpy().tobytes() == Mref.numpy().tobytes()
    Mn2 = ctx.zeros((5 * H, W))
    _ofri.check(lib.ofri_fb_window5_flow_matrices(ctx.handle, M.ptr, None, None, R0.ptr, R1.ptr, Mn2.ptr, H, W, taps.ctypes.data, ksh, box))
    assert Mn2.numpy().tobytes() == Mref.numpy().tobytes()
    assert lib.ofri_fb_window5_flow_matrices(ctx.handle, M.ptr, None, None, R0.ptr, R1.ptr, M.ptr, H, W, taps.ctypes.data, ksh, box) != 0

@pytest.mark.parametrize("shape", [(33, 64), (70, 68), (31, 132), (64, 200), (100, 256), (40, 1028)])
@pytest.mark.parametrize("box", [0, 1])
def test_farneback_window_tiles_fetched_by_dma_or_through_registers(ctx, shape, box, fb_mode):
    """The 33 x 33 window fetches its source tiles straight into LDS when rows are 16-byte groups on an aligned base
    and through registers otherwise: the same planes at an address 4 bytes further take the second way, same bits
    (plain filter, filter + solve, and the fused iteration)."""
    import _ofri
    from GaussianKernelBitExact import getGaussianKernelBitExact
    lib = _ofri.lib()
    H, W = shape
    n = 5 * H * W
    rng = np.random.default_rng(H + 7 * W + box)
    host = (rng.standard_normal(n) * 3).astype(np.float32)
    M = ctx.upload(host)
    big = ctx.upload(np.concatenate([np.zeros(1, np.float32), host]))
    assert M.ptr % 16 == 0 and big.ptr % 16 == 0
    off = big.ptr + 4
    R0 = ctx.upload(rng.standard_normal((5 * H, W)).astype(np.float32)); R1 = ctx.upload(rng.standard_normal((5 * H, W)).astype(np.float32))
    taps = np.ascontiguousarray(getGaussianKernelBitExact(33, 4.95)[1][16:], dtype=np.float32)
    assert taps.shape == (17,)
    out = []
    for src in (M.ptr, off):
        d5, fx, fy, Mn = ctx.zeros((5 * H, W)), ctx.zeros(shape), ctx.zeros(shape), ctx.zeros((5 * H, W))
        if box:
            _ofri.check(lib.ofri_fb_box5(ctx.handle, src, d5.ptr, H, W, 16))
        else:
            _ofri.check(lib.ofri_fb_blur5(ctx.handle, src, d5.ptr, H, W, taps.ctypes.data, 16))
        _ofri.check(lib.ofri_fb_window5_update_flow(ctx.handle, src, fx.ptr, fy.ptr, H, W, taps.ctypes.data, 16, box))
        _ofri.check(lib.ofri_fb_window5_flow_matrices(ctx.handle, src, None, None, R0.ptr, R1.ptr, Mn.ptr, H, W, taps.ctypes.data, 16, box))
        out.append([a.numpy() for a in (d5, fx, fy, Mn)])
    for a, b, what in zip(out[0], out[1], ("planes", "flow x", "flow y", "next matrices")):
        assert a.tobytes() == b.tobytes(), what
    assert np.isfinite(out[0][0]).all()


@pytest.mark.parametrize("kw", [dict(), dict(useGaussian=False, windowSize=15, Niters=3, polyN=5, polySigma=1.1),
                                dict(pyramidalLevels=3)])
def test_farneback_compute_vs_oracle(ctx, kw, fb_mode):
    from Farneback_PyCL import Farneback_PyCL
    from oracle import farneback as fb
    a, b, rng = _fb_inputs((160, 192), 5)
    U0 = (0.3 * rng.standard_normal(a.shape)).astype(np.float32); V0 = (0.3 * rng.standard_normal(a.shape)).astype(np.float32)
    U, V, err = Farneback_PyCL(**kw).compute(a, b, U0.copy(), V0.copy())
    Ur, Vr, _ = fb.FarnebackAdapter(**kw).compute(a, b, U0.copy(), V0.copy())
    assert err == 'Unknown'
    r = flow_rel(U, V, Ur, Vr)
    print("farneback", kw, "mode", fb_mode, "rel L2 vs oracle", r)
    assert r < (1e-5 if fb_mode == -1 else 1e-4)


def test_farneback_in_driver_and_with_liushen(ctx, pair):
    from Farneback_PyCL import Farneback_PyCL
    from PhysicsBasedOpticalFlowLiuShen import LiuShenOpticalFlowAlgoAdapter as LS
    from GenericPyramidalOpticalFlow import genericPyramidalOpticalFlow as flow
    from oracle import farneback as fb, liushen, pyramid
    I0, I1 = pair[0][:256, :256].copy(), pair[1][:256, :256].copy()
    U, V = flow(I0, I1, 0.0, Farneback_PyCL(), 2, 1, None)
    Ur, Vr = pyramid.pyramidal_flow(I0.copy(), I1.copy(), 0.0, fb.FarnebackAdapter(), 2, 1, None)
    r1 = flow_rel(U, V, Ur, Vr)
    U, V = flow(I0, I1, 0.0, Farneback_PyCL(), 2, 1, 0.48, LS(10))
    Ur, Vr = pyramid.pyramidal_flow(I0.copy(), I1.copy(), 0.0, fb.FarnebackAdapter(), 2, 1, 0.48, liushen.LiuShenAdapter(10))
    r2 = flow_rel(U, V, Ur, Vr)
    print("driver farneback rel L2 vs oracle", r1, "farneback + liu-shen", r2)
    assert r1 < 1e-4 and r2 < 1e-4
    # physical sanity on the bundled Poiseuille pair: flow in -x, strongest near the centre rows
    assert U[100:156, 40:216].mean() < -1.0


# ------------------------------------------------------------------------------ dense Lucas-Kanade
# Parity unpinned by reference outputs (the kernel needs an OpenCL image sampler); compared against the
# CPU restatement of the kernel text, which uses the same exact float32 bilinear sampling.
@pytest.mark.parametrize("mode", [-1, 0], ids=["exact", "fast"])
@pytest.mark.parametrize("shape,hw,nit", [((48, 56), 13, 5), ((33, 41), 6, 3), ((40, 40), 15, 2), ((70, 101), 13, 5)])
def test_lk_vs_oracle(ctx, shape, hw, nit, mode):
    from denseLucasKanade_PyCL import denseLucasKanade_PyCl
    from oracle import lk
    from piv_synth import synth_piv
    a, b = synth_piv(shape[0], shape[1], seed=4, peak=-2.0)
    a, b = a.astype(np.float32), b.astype(np.float32)
    rng = np.random.default_rng(9)
    U0 = (0.2 * rng.standard_normal(shape)).astype(np.float32); V0 = (0.2 * rng.standard_normal(shape)).astype(np.float32)
    a[:6, :20] = 0.0                       # a flat corner: singular structure tensor only where the window is empty
    U0[5, 7] = 500.0                       # a start point far outside the image: the loop must bail out
    U0[20:24, 30:33] += 9.0                # displacements far from their neighbours': outside the tile kernel's staged neighbourhood
    V0[30, 10:14] -= 7.5
    ctx.set_lk_mode(mode)
    try:
        U, V, e = denseLucasKanade_PyCl(Niter=nit, halfWindow=hw).compute(a, b, U0.copy(), V0.copy())
    finally:
        ctx.set_lk_mode(0)
    Ur, Vr = lk.lk_dense(a, b, U0, V0, nit, hw)
    assert e is True and U.dtype == np.float32 and U.shape == shape
    bad = (np.abs(U - Ur) > 1e-4) | (np.abs(V - Vr) > 1e-4)
    sane = np.abs(Ur) < 100                # the bailed-out pixel keeps its 500 px and would swamp the norm
    r = flow_rel(U[sane], V[sane], Ur[sane], Vr[sane])
    print("lk", shape, hw, nit, "mode", mode, "rel L2 vs oracle", r, "pixels off by >1e-4:", int(bad.sum()), "largest", float(np.abs(U - Ur).max()))
    # exact mode: the oracle's arithmetic (a Newton loop's stopping test can still flip on a rounding difference);
    # fast mode: sums in another order, factored interpolation
    assert r < (1e-5 if mode == -1 else 1e-4) and bad.mean() < (1e-3 if mode == -1 else 1e-2)
    assert float(np.abs(U - Ur)[sane].max()) < 2e-3 and float(np.abs(V - Vr).max()) < 2e-3


def test_lk_asymmetric_windows_and_driver(ctx, pair):
    from denseLucasKanade_PyCL import denseLucasKanade_PyCl
    from GenericPyramidalOpticalFlow import genericPyramidalOpticalFlow as flow
    from oracle import lk, pyramid
    I0, I1 = pair[0][200:264, 100:180].copy(), pair[1][200:264, 100:180].copy()
    z = np.zeros_like(I0)
    ad = denseLucasKanade_PyCl(enableVorticityEnhancement=True)
    shear = np.fromfunction(lambda y, x: 0.05 * y, I0.shape).astype(np.float32)      # dU/dy > 0: positive "omega"
    assert ad.evaluateVorticityEnhancement(shear, z) == lk.vorticity_config(shear, z, True) == [1, 0, 0, 1]
    assert ad.evaluateVorticityEnhancement(-shear, z) == lk.vorticity_config(-shear, z, True) == [0, 1, 0, 1]
    assert ad.evaluateVorticityEnhancement(z, z) == [0, 0, 0, 0]
    for cfg in ([1, 0, 0, 1], [0, 1, 0, 1]):
        Un, Vn, _ = ad._run(ctx.upload(I0), ctx.upload(I1), ctx.upload(z), ctx.upload(z), cfg)
        Ur, Vr = lk.lk_dense(I0, I1, z, z, 5, 13, cfg)
        assert flow_rel(Un.numpy(), Vn.numpy(), Ur, Vr) < 1e-5
    # the other windows the tile kernel is built for (13 and 31 samples), every flag set at least once, both modes
    for hw, cfg in ((6, [1, 0, 0, 1]), (6, [0, 1, 1, 0]), (15, [1, 1, 1, 1]), (15, [0, 1, 0, 0])):
        adw = denseLucasKanade_PyCl(halfWindow=hw, Niter=4, enableVorticityEnhancement=True)
        Ur, Vr = lk.lk_dense(I0, I1, z, z, 4, hw, cfg)
        for mode in (0, -1):
            ctx.set_lk_mode(mode)
            try:
                Un, Vn, _ = adw._run(ctx.upload(I0), ctx.upload(I1), ctx.upload(z), ctx.upload(z), cfg)
            finally:
                ctx.set_lk_mode(0)
            r = flow_rel(Un.numpy(), Vn.numpy(), Ur, Vr)
            print("lk window", 2 * hw + 1, "flags", cfg, "mode", mode, "rel L2 vs oracle", r)
            assert r < (1e-5 if mode == -1 else 1e-4)
            assert float(np.abs(Un.numpy() - Ur).max()) < 2e-3 and float(np.abs(Vn.numpy() - Vr).max()) < 2e-3
    # through the pyramid driver with the adapter's defaults (no warping, final level not rescaled)
    J0, J1 = pair[0][128:256, 128:256].copy(), pair[1][128:256, 128:256].copy()
    U, V = flow(J0, J1, 2.0, denseLucasKanade_PyCl(), 2, 1, None)
    Ur, Vr = pyramid.pyramidal_flow(J0.copy(), J1.copy(), 2.0, lk.DenseLKAdapter(), 2, 1, None)
    r = flow_rel(U, V, Ur, Vr)
    print("driver dense LK rel L2 vs oracle", r)
    assert r < 1e-4


# ------------------------------------------------------------------------------ C ABI behaviour
def test_abi_rejects_bad_arguments_and_stays_usable(ctx):
    """Bad calls return a negative code with a message (no exception crosses the ABI, no kernel runs) and
    leave the context usable."""
    import ctypes as C
    import _ofri
    lib = _ofri.lib()
    a = ctx.upload(np.ones((16, 16), np.float32))
    b = ctx.empty((16, 16)); c = ctx.empty((16, 16))
    taps = np.array([0.25, 0.5, 0.25], np.float32)
    even = np.array([0.5, 0.5], np.float32)
    err = C.c_double()
    cases = [
        lib.ofri_gauss_sep(ctx.handle, None, b.ptr, 16, 16, taps.ctypes.data, 3),                        # null source
        lib.ofri_gauss_sep(ctx.handle, a.ptr, b.ptr, 0, 16, taps.ctypes.data, 3),                        # empty shape
        lib.ofri_gauss_sep(ctx.handle, a.ptr, b.ptr, 16, 16, even.ctypes.data, 2),                       # even tap count
        lib.ofri_hs_compute(ctx.handle, a.ptr, a.ptr, a.ptr, a.ptr, a.ptr, b.ptr, 1.0, 3, 16, 16, None),  # output aliases input
        lib.ofri_hs_compute(ctx.handle, a.ptr, a.ptr, a.ptr, None, b.ptr, c.ptr, 1.0, 3, 16, 16, None),   # only one of U, V null
        lib.ofri_hs_compute(ctx.handle, a.ptr, a.ptr, None, None, b.ptr, c.ptr, 1.0, -1, 16, 16, None),   # negative sweeps
        lib.ofri_hs_iterate(ctx.handle, b.ptr, c.ptr, a.ptr, a.ptr, None, 1.0, 3, 16, 16),               # null coefficient plane
        lib.ofri_spline_upsample(ctx.handle, a.ptr, 3, 16, b.ptr, 6, 32, 1.0),                           # fewer than 4 knots
        lib.ofri_resize_bicubic(ctx.handle, a.ptr, 16, 16, a.ptr, 8, 8),                                 # in place
        lib.ofri_hs_set_fuse(ctx.handle, 1000),
        lib.ofri_hs_compute(None, a.ptr, a.ptr, None, None, b.ptr, c.ptr, 1.0, 3, 16, 16, None),         # null context
    ]
    assert all(rc < 0 for rc in cases), cases
    assert lib.ofri_last_error()                                   # the last failure left its message
    # still works, and a zero start given as NULL equals one given as planes of zeros
    z = ctx.zeros((16, 16))
    assert lib.ofri_hs_compute(ctx.handle, a.ptr, a.ptr, None, None, b.ptr, c.ptr, 2.0, 5, 16, 16, C.byref(err)) == 0
    d, e = ctx.empty((16, 16)), ctx.empty((16, 16))
    assert lib.ofri_hs_compute(ctx.handle, a.ptr, a.ptr, z.ptr, z.ptr, d.ptr, e.ptr, 2.0, 5, 16, 16, None) == 0
    np.testing.assert_array_equal(b.numpy(), d.numpy()); np.testing.assert_array_equal(c.numpy(), e.numpy())


def test_hs_zero_start_as_null_on_streaming_sizes(ctx):
    """On fields that take the row-streaming pipeline a NULL start skips the zero planes; same bits."""
    from HornSchunck import hs_device
    from piv_synth import synth_piv
    a, b = synth_piv(2048, 2048, seed=3)
    A, B = ctx.upload(a.astype(np.float32)), ctx.upload(b.astype(np.float32))
    z = ctx.zeros(A.shape)
    for nit in (30, 24, 7):                                        # pipeline + tile rest, pipeline only, tile only
        U0, V0, e0 = hs_device(A, B, 21.0, nit, None, None, True)
        U1, V1, e1 = hs_device(A, B, 21.0, nit, z, z, True)
        np.testing.assert_array_equal(U0.numpy(), U1.numpy()); np.testing.assert_array_equal(V0.numpy(), V1.numpy())
        assert e0 == e1 and np.isfinite(U0.numpy()).all()

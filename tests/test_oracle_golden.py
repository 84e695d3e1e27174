"""The CPU oracle against fixtures produced by running the reference itself (oracle/make_golden.py)."""
import numpy as np
import pytest

from conftest import load_gold, rel_l2
from oracle import gauss, hs, liushen, pyramid


def test_gauss_taps():
    g = load_gold("gauss")
    for sig, K in ((3.4, 3), (2.0, 3), (0.48, 5), (1.0, 7)):
        np.testing.assert_array_equal(gauss.prepare_kernel(sig, K), g["taps_%g_%d" % (sig, K)])


def test_gauss_filter_bit_exact():
    g = load_gold("gauss")
    for key, (sig, K) in {"crop_f_3.4_3": (3.4, 3), "crop_f_0.48_5": (0.48, 5), "crop_f_1_7": (1.0, 7)}.items():
        np.testing.assert_array_equal(gauss.filter_px(g["crop"].copy(), sig, K), g[key])
    np.testing.assert_array_equal(gauss.filter_trunc(g["small"].copy(), 1.8, 20.0), g["small_f_1.8_t20"])


def test_gauss_in_place(pair):
    a = pair[0][:64, :64].copy()
    r = gauss.filter_px(a, 3.4, 3)
    assert r is a


def test_hs_derivatives():
    h = load_gold("hs")
    fx, fy, ft = hs.derivatives(h["im1"], h["im0"])
    for got, key in ((fx, "fx"), (fy, "fy"), (ft, "ft")):
        np.testing.assert_array_equal(got, h[key])


@pytest.mark.parametrize("alpha,nit,init", [(21, 1, "rand"), (21, 10, "rand"), (21, 1, "zero"), (21, 10, "zero"),
                                            (21, 100, "zero"), (1.0, 100, "zero"), (45, 7, "rand")])
def test_hs_iterations(alpha, nit, init):
    h = load_gold("hs")
    Z = np.zeros_like(h["U0"])
    Ui, Vi = (h["U0"], h["V0"]) if init == "rand" else (Z, Z)
    U, V, err = hs.hs(h["im0"], h["im1"], alpha, nit, Ui.copy(), Vi.copy())
    key = "hs_a%g_n%d_%s" % (alpha, nit, init)
    np.testing.assert_array_equal(U, h[key + "_U"])
    np.testing.assert_array_equal(V, h[key + "_V"])
    assert err == pytest.approx(float(h[key + "_err"]), rel=1e-12)


def test_hs_adapter_pops_last_alpha():
    h = load_gold("hs")
    ad = hs.HSAdapter([3.0, 9.0], 25)
    z = np.zeros_like(h["adapter_crop_im0"])
    U, V, err = ad.compute(h["adapter_crop_im0"], h["adapter_crop_im1"], z, z.copy())
    np.testing.assert_array_equal(U, h["adapter_crop_U"])
    np.testing.assert_array_equal(V, h["adapter_crop_V"])
    assert ad.alphas == list(h["adapter_alphas_left"])
    ad.compute(h["adapter_crop_im0"], h["adapter_crop_im1"], z, z)
    with pytest.raises(IndexError):
        ad.compute(h["adapter_crop_im0"], h["adapter_crop_im1"], z, z)


def test_resize(pair):
    p = load_gold("pyramid")
    np.testing.assert_array_equal(pyramid.resize_lib(pair[0], (256, 256))[::2, ::2], p["resize_256_sub"])
    np.testing.assert_array_equal(pyramid.resize_lib(p["odd"], (156, 100)), p["resize_odd"])
    # explicit restatement of Pillow's resampler: within half a float32 ulp of 255
    for src, wh, key in ((p["odd"], (156, 100), "resize_odd"), (pair[0][:300, :450].copy(), (150, 100), "resize_third")):
        e = pyramid.resize_explicit(src, wh)
        assert e.shape == p[key].shape
        assert np.abs(e - p[key]).max() <= 1.6e-5
    same = pyramid.resize_explicit(p["odd"], (313, 201))
    np.testing.assert_array_equal(same, p["odd"])


def test_spline_upsample():
    p = load_gold("pyramid")
    for src, shape, key, sc in ((p["Uc"], (256, 256), "up_Uacc", 2.0), (p["Uo"], (201, 313), "odd_Uacc", 313 / 156),
                                (p["Vo"], (201, 313), "odd_Vacc", 201 / 100)):
        lib = pyramid.spline_upsample_lib(src, shape) * np.float32(np.float32(sc))
        np.testing.assert_array_equal(lib, p[key])
        ex = pyramid.spline_upsample_explicit(src, shape) * np.float32(np.float32(sc))
        assert np.abs(ex - p[key]).max() <= 2e-6 * max(1.0, np.abs(p[key]).max())


def test_warp_and_next_level():
    p = load_gold("pyramid")
    np.testing.assert_array_equal(pyramid.warp_bilinear(p["warp_img"], p["warp_cy"], p["warp_cx"]), p["warp_out"])
    r = pyramid.next_level(p["P0"].copy(), (128, 128), p["P1"].copy(), p["Uc"].copy(), p["Vc"].copy(), True, True, True)
    for got, key in zip(r, ("up_im1", "up_im2", "up_Uacc", "up_Vacc", "up_U", "up_V")):
        np.testing.assert_array_equal(got, p[key])
    r = pyramid.next_level(p["P0"].copy(), (128, 128), p["P1"].copy(), p["Uc"].copy(), p["Vc"].copy(), False, True, False)
    np.testing.assert_array_equal(r[4], p["nowarp_U"])
    np.testing.assert_array_equal(r[5], p["nowarp_V"])
    assert not r[2].any()


def test_liushen_raw():
    l = load_gold("liushen")
    from oracle import gauss as G
    # coefficient matrices on the 256x256 corner of the optional-filtered frame 0
    # (inputs are rebuilt from the bundled pair exactly as the generator did)
    pairg = load_gold("pair_parabolic01")
    I0o = G.filter_px(pairg["im0"].astype(np.float32), 0.48, 5)[:256, :256].copy()
    I1o = G.filter_px(pairg["im1"].astype(np.float32), 0.48, 5)[:256, :256].copy()
    B11, B12, B22 = liushen.inverse_matrix(I0o / np.max(I0o), 5)
    np.testing.assert_array_equal(B11, l["B11"]); np.testing.assert_array_equal(B12, l["B12"])
    np.testing.assert_array_equal(B22, l["B22"])
    z = np.zeros((256, 256), np.float32)
    u, v, err, it = liushen.liushen(I0o, I1o, 5, z, z.copy(), return_iters=True)
    np.testing.assert_array_equal(u, l["raw_h5_zero_u"]); np.testing.assert_array_equal(v, l["raw_h5_zero_v"])
    assert err == pytest.approx(float(l["raw_h5_zero_err"]), rel=1e-6)
    assert it == str(l["raw_h5_zero_log"]).count("Iteration:")


def test_liushen_adapter_swaps_components():
    l = load_gold("liushen")
    U, V, err = liushen.LiuShenAdapter(10).compute(l["crop_im0"].copy(), l["crop_im1"].copy(), l["crop_Uin"].copy(), l["crop_Vin"].copy())
    np.testing.assert_array_equal(U, l["crop_U"]); np.testing.assert_array_equal(V, l["crop_V"])
    assert err == pytest.approx(float(l["crop_err"]), rel=1e-6)


def _check_driver(name, U, V, gold="driver"):
    d = load_gold(gold)
    s = int(d[name + "_sub"])
    np.testing.assert_array_equal(U[::s, ::s], d[name + "_U"]); np.testing.assert_array_equal(V[::s, ::s], d[name + "_V"])
    assert float(U.astype(np.float64).sum()) == pytest.approx(d[name + "_Ustats"][0], rel=1e-12, abs=1e-9)
    assert float(np.linalg.norm(V.astype(np.float64))) == pytest.approx(d[name + "_Vstats"][1], rel=1e-12)


def test_driver_hs_levels(pair):
    I0, I1 = pair
    _check_driver("hs_L1_a21", *pyramid.pyramidal_flow(I0.copy(), I1.copy(), 3.4, hs.HSAdapter([21], 100), 1, 1, None))
    _check_driver("hs_L2", *pyramid.pyramidal_flow(I0.copy(), I1.copy(), 3.4, hs.HSAdapter([21, 45], 100), 2, 1, None))
    _check_driver("hs_L3_nofilter", *pyramid.pyramidal_flow(I0.copy(), I1.copy(), 0.0, hs.HSAdapter([21, 45, 60], 30), 3, 1, None))


def test_driver_hs_with_the_scripts_own_iteration_count(pair):
    """Niter = 600 as in examples/PyHSchunck_Fs3_4_PyrLvls2.py:132."""
    I0, I1 = pair
    _check_driver("hs_L2_n600", *pyramid.pyramidal_flow(I0.copy(), I1.copy(), 3.4, hs.HSAdapter([21, 45], 600), 2, 1, None),
                  gold="driver_n600")


def test_driver_variants(pair):
    I0, I1 = pair
    _check_driver("hs_L2_k2", *pyramid.pyramidal_flow(I0[:256, :256].copy(), I1[:256, :256].copy(), 3.4,
                                                     hs.HSAdapter([21, 21, 45, 45], 20), 2, 2, None))
    _check_driver("hs_L2_odd", *pyramid.pyramidal_flow(I0[:201, :313].copy(), I1[:201, :313].copy(), 2.0,
                                                      hs.HSAdapter([21, 45], 40), 2, 1, None))
    _check_driver("ls_main_L2", *pyramid.pyramidal_flow(I0[:256, :256].copy(), I1[:256, :256].copy(), 3.4,
                                                       liushen.LiuShenAdapter(0.1), 2, 1, None, None, True, True, True, True))


def test_driver_hs_plus_liushen(pair):
    I0, I1 = pair
    _check_driver("hs_L2_ls", *pyramid.pyramidal_flow(I0.copy(), I1.copy(), 3.4, hs.HSAdapter([21, 45], 100), 2, 1, 0.48,
                                                     liushen.LiuShenAdapter(5)))


def test_farneback_host_tables():
    from oracle import farneback as fb
    f = load_gold("farneback_tables")
    for n, s in ((3, 0), (3, 0.5), (5, 0), (33, 4.95), (33, 16.5 * 0.3), (7, 1.2), (9, 0)):
        k = fb.bitexact_kernel(n, s)
        np.testing.assert_array_equal(k, f["bitexact_%d_%g" % (n, s)])
        np.testing.assert_array_equal(np.float32(k), f["bitexact_f32_%d_%g" % (n, s)])
    for n, s in ((7, 1.5), (5, 1.1), (5, 1.2), (7, 0.0)):
        r = fb.prepare_gaussian(n, s)
        for i, a in enumerate(r):
            np.testing.assert_array_equal(np.asarray(a), f["prep_%d_%g_%d" % (n, s, i)])


def test_farneback_oracle_recovers_poiseuille_profile():
    """No reference output exists for the OpenCL kernels (parity unpinned); the restatement must at least
    recover the known displacement field of a synthetic pair: u(y) parabolic in -x, v ~ 0."""
    from oracle import farneback as fb
    import sys, os
    from piv_synth import synth_piv, poiseuille_u
    a, b = synth_piv(128, 128, seed=2)
    a, b = a.astype(np.float32), b.astype(np.float32)
    z = np.zeros_like(a)
    U, V, err = fb.FarnebackAdapter().compute(a, b, z, z.copy())
    assert err == 'Unknown' and U.dtype == np.float32
    prof = U[24:104, 24:104].mean(axis=1)
    want = poiseuille_u(128)[24:104]
    assert np.abs(prof - want).max() < 0.6
    assert abs(V[24:104, 24:104].mean()) < 0.2


def test_lk_oracle_recovers_uniform_shift():
    """Parity unpinned (no OpenCL device for the reference kernel): the restatement must at least find a
    known sub-pixel translation of a textured image."""
    from oracle import lk
    rng = np.random.default_rng(0)
    base = rng.uniform(0, 255, (70, 80)).astype(np.float32)
    from scipy.ndimage import gaussian_filter, shift
    base = gaussian_filter(base, 1.5).astype(np.float32)
    moved = shift(base, (0.6, -1.3), order=3, mode='nearest').astype(np.float32)       # content moves by (+0.6 rows, -1.3 cols)
    z = np.zeros_like(base)
    U, V = lk.lk_dense(base[:40, :44], moved[:40, :44], z[:40, :44], z[:40, :44], 5, 13)
    assert abs(U[14:26, 14:30].mean() + 1.3) < 0.1 and abs(V[14:26, 14:30].mean() - 0.6) < 0.1
    flat = np.full((20, 20), 3.0, np.float32)
    U, V = lk.lk_dense(flat, flat, z[:20, :20] + 0.25, z[:20, :20], 5, 13)
    assert (U == 0.25).all() and (V == 0).all()                                          # singular windows keep their input


# ------------------------------------------------------------------------------ biLinear=False warp
def test_shift_warp_matches_reference_bits():
    from oracle import pyramid
    w = load_gold("shiftwarp")
    arg = w["im"].copy()
    r = pyramid.next_level(arg, (120, 150), w["im2"].copy(), w["U"].copy(), w["V"].copy(), True, False, False)
    assert r[0] is arg                                           # overwritten in place, as the reference does
    np.testing.assert_array_equal(r[0], w["out_im1"]); np.testing.assert_array_equal(r[1], w["out_im2"])
    np.testing.assert_array_equal(r[2], w["out_Uacc"]); np.testing.assert_array_equal(r[3], w["out_Vacc"])
    assert not r[4].any() and not r[5].any()
    assert np.count_nonzero(w["out_im1"] != w["im"]) > 1000


def test_shift_warp_driver_matches_reference_bits():
    from oracle import pyramid, hs
    w = load_gold("shiftwarp")
    pg = load_gold("pair_parabolic01")
    a0, a1 = pg["im0"].astype(np.float32)[:256, :256].copy(), pg["im1"].astype(np.float32)[:256, :256].copy()
    U, V = pyramid.pyramidal_flow(a0, a1, 3.4, hs.HSAdapter([21, 45], 30, False), 2, 1, None, None, True, False, True, False)
    np.testing.assert_array_equal(U, w["drv_U"]); np.testing.assert_array_equal(V, w["drv_V"])
    np.testing.assert_array_equal(a0, w["drv_im1_after"])       # the caller's first image is overwritten

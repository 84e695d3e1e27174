#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the reference itself.

TEST INFRASTRUCTURE ONLY.  Runs only in the build container, where the reference
tree is mounted read-only at /root/reference; nothing here travels to the GPU box
except the .npz/.npy files it writes.

The reference's HS / Liu-Shen / Gaussian modules do ``from numba import jit, njit``.
numba is not installed here, so this script puts a tiny stand-in package (decorators
that return the function unchanged, ``prange = range``) in a temp dir ahead of the
reference on sys.path.  The three decorated bodies (HornSchunck.py:52-59,
PhysicsBasedOpticalFlowLiuShen.py:75-80, gaussian_filter.py:24-45) are float32
element-wise / sequential loops that NumPy evaluates in float32 as well, so the
stand-in does not change the arithmetic.  The Farneback / dense-LK adapters need
PyOpenCL and an OpenCL device and cannot run here; only their pure-Python host
constant tables are captured (Farneback_PyCL.py:124-209, GaussianKernelBitExact.py).

Usage:  python oracle/make_golden.py            (writes tests/golden/*.npz)
"""
import contextlib
import io
import os
import sys
import tempfile

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

_SHIM = '''
def _passthrough(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return a[0]
    def deco(f):
        return f
    return deco
jit = njit = _passthrough
prange = range
class objmode(object):
    def __init__(self, *a, **k): pass
    def __enter__(self): return self
    def __exit__(self, *a): return False
'''


def _install_shims():
    d = tempfile.mkdtemp(prefix="ofri_shim_")
    os.makedirs(os.path.join(d, "numba"))
    with open(os.path.join(d, "numba", "__init__.py"), "w") as f:
        f.write(_SHIM)
    os.makedirs(os.path.join(d, "pyopencl"))
    for n in ("__init__.py", "array.py", "cltypes.py"):
        with open(os.path.join(d, "pyopencl", n), "w") as f:
            f.write("")
    sys.dont_write_bytecode = True
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.path[:0] = [d, os.path.join(REF, "src")]


def _quiet(fn, *a, **k):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        r = fn(*a, **k)
    return r, buf.getvalue()


def stats(a):
    a = np.asarray(a)
    return np.array([a.astype(np.float64).sum(), np.linalg.norm(a.astype(np.float64)),
                     float(a.min()), float(a.max())], dtype=np.float64)


def main():
    import warnings
    warnings.simplefilter("ignore")
    _install_shims()
    from PIL import Image
    import GenericPyramidalOpticalFlow as G
    import HornSchunck as H
    import PhysicsBasedOpticalFlowLiuShen as L
    import gaussian_filter as GF

    os.makedirs(OUT, exist_ok=True)
    base = os.path.join(REF, "examples", "testImages", "Bits08", "Ni06")
    I0u = np.array(Image.open(os.path.join(base, "parabolic01_0.tif")))
    I1u = np.array(Image.open(os.path.join(base, "parabolic01_1.tif")))
    assert I0u.dtype == np.uint8 and I0u.shape == (512, 512)
    np.savez_compressed(os.path.join(OUT, "pair_parabolic01.npz"), im0=I0u, im1=I1u)
    I0 = I0u.astype(np.float32)
    I1 = I1u.astype(np.float32)
    rng = np.random.default_rng(1234)

    # ---- Gaussian pre-filter (gaussian_filter.py:47-94) -------------------------------
    g = {}
    for sig, K in ((3.4, 3), (2.0, 3), (0.48, 5), (1.0, 7)):
        g["taps_%g_%d" % (sig, K)] = GF.prepareGaussianKernel(sig, K)
    crop = I0[100:196, 200:280].copy()                       # 96 x 80, ragged vs any tile
    g["crop"] = crop
    g["crop_f_3.4_3"] = GF.gaussian_filterPx(crop.copy(), 3.4, 3)
    g["crop_f_0.48_5"] = GF.gaussian_filterPx(crop.copy(), 0.48, 5)
    g["crop_f_1_7"] = GF.gaussian_filterPx(crop.copy(), 1.0, 7)
    # gaussian_filter(sigma, truncate) form used by the non-bilinear warp (K = 73)
    small = I0[300:340, 100:150].copy()                      # 40 x 50 < K: heavy border reuse
    g["small"] = small
    g["small_f_1.8_t20"] = GF.gaussian_filter(small.copy(), 1.8, 20.0)
    I0f = GF.gaussian_filterPx(I0.copy(), 3.4, 3)
    I1f = GF.gaussian_filterPx(I1.copy(), 3.4, 3)
    g["full0_f_3.4_3_sub"] = I0f[::2, ::2].copy()
    g["full0_f_3.4_3_stats"] = stats(I0f)
    np.savez_compressed(os.path.join(OUT, "gauss.npz"), **g)

    # ---- Horn-Schunck pieces (HornSchunck.py:52-127) ----------------------------------
    h = {}
    # adapter.compute(im1=frame0, im2=frame1) -> HS(im2=frame0, im1=frame1) -> computeDerivatives(frame1, frame0)
    A, B = I0f[96:256, 64:256].copy(), I1f[96:256, 64:256].copy()      # 160 x 192 crop of the filtered pair
    h["im0"], h["im1"] = A, B
    fx, fy, ft = H.computeDerivatives(B, A)
    h["fx"], h["fy"], h["ft"] = fx, fy, ft
    cfx, cfy, cft = H.computeDerivatives(I1[7:60, 11:90].copy(), I0[7:60, 11:90].copy())   # 53 x 79 odd sizes
    h["crop_fx"], h["crop_fy"], h["crop_ft"] = cfx, cfy, cft
    U0 = rng.standard_normal(A.shape).astype(np.float32)
    V0 = rng.standard_normal(A.shape).astype(np.float32)
    h["U0"], h["V0"] = U0, V0
    Z = np.zeros(A.shape, np.float32)
    for alpha, nit, init in ((21, 1, "rand"), (21, 10, "rand"), (21, 1, "zero"), (21, 10, "zero"),
                             (21, 100, "zero"), (1.0, 100, "zero"), (45, 7, "rand")):
        Ui, Vi = (U0, V0) if init == "rand" else (Z, Z)
        (U, V, err), _ = _quiet(H.HS, A, B, alpha, nit, Ui.copy(), Vi.copy())
        key = "hs_a%g_n%d_%s" % (alpha, nit, init)
        h[key + "_U"], h[key + "_V"], h[key + "_err"] = U, V, np.float64(err)
    # small ragged case straight through the adapter (pops alpha from the end)
    ad = H.HSOpticalFlowAlgoAdapter([3.0, 9.0], 25)
    a, b = I0[7:60, 11:90].copy(), I1[7:60, 11:90].copy()
    (U, V, err), _ = _quiet(ad.compute, a, b, np.zeros_like(a), np.zeros_like(a))
    h["adapter_crop_im0"], h["adapter_crop_im1"] = a, b
    h["adapter_crop_U"], h["adapter_crop_V"], h["adapter_crop_err"] = U, V, np.float64(err)
    h["adapter_alphas_left"] = np.array(ad.alphas, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "hs.npz"), **h)

    # ---- pyramid glue (GenericPyramidalOpticalFlow.py:67-235) -------------------------
    p = {}
    p["resize_256_sub"] = G.imresize(I0, (256, 256))[::2, ::2].copy()
    p["resize_256_stats"] = stats(G.imresize(I0, (256, 256)))
    odd = I0[:201, :313].copy()
    p["odd"] = odd
    p["resize_odd"] = G.imresize(odd, (int(np.round(313 * 0.5)), int(np.round(201 * 0.5))))   # (156, 100)
    p["resize_third"] = G.imresize(I0[:300, :450].copy(), (150, 100))
    # flow upsampling + scaling + symmetric warp, 256 -> 512
    yy, xx = np.mgrid[0:128, 0:128].astype(np.float32)
    Uc = (-2.0 * (1 - ((yy - 63.5) / 63.5) ** 2) + 0.3 * np.sin(xx / 17.0)).astype(np.float32)
    Vc = (0.4 * np.cos(yy / 23.0) * np.sin(xx / 31.0)).astype(np.float32)
    p["Uc"], p["Vc"] = Uc, Vc
    prev = np.zeros((128, 128), np.float32)
    P0, P1 = I0[128:384, 128:384].copy(), I1[128:384, 128:384].copy()
    p["P0"], p["P1"] = P0, P1
    (r, _o) = _quiet(G.updateNextPyramidalLevel, P0.copy(), prev, P1.copy(), Uc.copy(), Vc.copy(),
                     None, None, True, True, True)
    p["up_im1"], p["up_im2"], p["up_Uacc"], p["up_Vacc"], p["up_U"], p["up_V"] = r
    (r, _o) = _quiet(G.updateNextPyramidalLevel, P0.copy(), prev, P1.copy(), Uc.copy(), Vc.copy(),
                     None, None, False, True, False)
    p["nowarp_Uacc"], p["nowarp_U"], p["nowarp_V"] = r[2], r[4], r[5]
    # non-2x, non-square upsample: (100,156) -> (201,313)
    Uo = rng.standard_normal((100, 156)).astype(np.float32).cumsum(1).astype(np.float32) * 0.05
    Vo = rng.standard_normal((100, 156)).astype(np.float32).cumsum(0).astype(np.float32) * 0.05
    p["Uo"], p["Vo"] = Uo, Vo
    (r, _o) = _quiet(G.updateNextPyramidalLevel, odd.copy(), np.zeros((100, 156), np.float32),
                     I1[:201, :313].copy(), Uo.copy(), Vo.copy(), None, None, True, True, True)
    p["odd_im1"], p["odd_im2"], p["odd_Uacc"], p["odd_Vacc"] = r[0], r[1], r[2], r[3]
    # bilinear warp alone with large, out-of-range displacements (clamp path)
    cy = (np.mgrid[0:64, 0:80][0] + 9.0 * rng.standard_normal((64, 80))).astype(np.float32)
    cx = (np.mgrid[0:64, 0:80][1] + 9.0 * rng.standard_normal((64, 80))).astype(np.float32)
    wimg = I0[200:264, 100:180].copy()
    p["warp_img"], p["warp_cy"], p["warp_cx"] = wimg, cy, cx
    p["warp_out"] = G.doBiLinearWarping(wimg, cy, cx, 1, "nearest")
    np.savez_compressed(os.path.join(OUT, "pyramid.npz"), **p)

    # ---- Liu-Shen (PhysicsBasedOpticalFlowLiuShen.py:33-158) --------------------------
    l = {}
    I0o = GF.gaussian_filterPx(I0.copy(), 0.48, 5)
    I1o = GF.gaussian_filterPx(I1.copy(), 0.48, 5)
    (r, _o) = _quiet(L.physicsBasedOpticalFlowLiuShen, I0o[:256, :256].copy(), I1o[:256, :256].copy(), 5,
                     np.zeros((256, 256), np.float32), np.zeros((256, 256), np.float32))
    l["raw_h5_zero_u"], l["raw_h5_zero_v"], l["raw_h5_zero_err"] = r[0], r[1], np.float64(r[2])
    l["raw_h5_zero_log"] = np.array(_o)
    im = I0o[:256, :256] / np.max(I0o[:256, :256])
    B11, B12, B22 = L.generate_invmatrix(im, 5, 1)
    l["B11"], l["B12"], l["B22"] = B11, B12, B22
    ca, cb = I0o[40:97, 300:371].copy(), I1o[40:97, 300:371].copy()          # 57 x 71
    Uin = (0.5 * rng.standard_normal(ca.shape)).astype(np.float32)
    Vin = (0.5 * rng.standard_normal(ca.shape)).astype(np.float32)
    l["crop_im0"], l["crop_im1"], l["crop_Uin"], l["crop_Vin"] = ca, cb, Uin, Vin
    ad = L.LiuShenOpticalFlowAlgoAdapter(10)
    (r, _o) = _quiet(ad.compute, ca.copy(), cb.copy(), Uin.copy(), Vin.copy())
    l["crop_U"], l["crop_V"], l["crop_err"] = r[0], r[1], np.float64(r[2])
    l["crop_iters"] = np.int64(_o.count("Iteration:"))
    np.savez_compressed(os.path.join(OUT, "liushen.npz"), **l)

    # ---- whole-driver cases (GenericPyramidalOpticalFlow.py:238-416) ------------------
    d = {}

    def run(name, *a, **k):
        (r, o) = _quiet(G.genericPyramidalOpticalFlow, *a, **k)
        if r[0].size > 256 * 256:            # keep the committed fixture small: 1/4 regular subsample + full-field stats
            d[name + "_U"], d[name + "_V"] = r[0][::2, ::2].copy(), r[1][::2, ::2].copy()
            d[name + "_sub"] = np.int64(2)
        else:
            d[name + "_U"], d[name + "_V"] = r[0], r[1]
            d[name + "_sub"] = np.int64(1)
        d[name + "_Ustats"], d[name + "_Vstats"] = stats(r[0]), stats(r[1])
        d[name + "_log"] = np.array(o)
        print(name, stats(r[0]), stats(r[1]))

    run("hs_L1_a21", I0.copy(), I1.copy(), 3.4, H.HSOpticalFlowAlgoAdapter([21], 100), 1, 1, None)
    run("hs_L1_a1", I0.copy(), I1.copy(), 3.4, H.HSOpticalFlowAlgoAdapter([1.0], 100), 1, 1, None)
    run("hs_L2", I0.copy(), I1.copy(), 3.4, H.HSOpticalFlowAlgoAdapter([21, 45], 100), 2, 1, None)
    run("hs_L2_ls", I0.copy(), I1.copy(), 3.4, H.HSOpticalFlowAlgoAdapter([21, 45], 100), 2, 1, 0.48,
        L.LiuShenOpticalFlowAlgoAdapter(5))
    run("hs_L3_nofilter", I0.copy(), I1.copy(), 0.0, H.HSOpticalFlowAlgoAdapter([21, 45, 60], 30), 3, 1, None)
    run("hs_L2_k2", I0[:256, :256].copy(), I1[:256, :256].copy(), 3.4,
        H.HSOpticalFlowAlgoAdapter([21, 21, 45, 45], 20), 2, 2, None)
    run("hs_L2_odd", I0[:201, :313].copy(), I1[:201, :313].copy(), 2.0,
        H.HSOpticalFlowAlgoAdapter([21, 45], 40), 2, 1, None)
    # Liu-Shen as the main adapter (benchmark_of_methods.py:159-163): caller kwargs apply
    run("ls_main_L2", I0[:256, :256].copy(), I1[:256, :256].copy(), 3.4, L.LiuShenOpticalFlowAlgoAdapter(0.1),
        2, 1, None, None, True, True, True, True)
    np.savez_compressed(os.path.join(OUT, "driver.npz"), **d)

    # ---- the example script's own iteration count (examples/PyHSchunck_Fs3_4_PyrLvls2.py:132: Niter = 600) ----
    d = {}
    run("hs_L2_n600", I0.copy(), I1.copy(), 3.4, H.HSOpticalFlowAlgoAdapter([21, 45], 600), 2, 1, None)
    np.savez_compressed(os.path.join(OUT, "driver_n600.npz"), **d)

    # ---- optical-flow-equation warp, biLinear=False (GenericPyramidalOpticalFlow.py:189-221) --------
    w = {}
    wi = I0[60:180, 200:350].copy()                                        # 120 x 150
    yy, xx = np.mgrid[0:120, 0:150].astype(np.float32)
    # non-positive shifts: destinations wrap at the low edges, never leave the array at the high edges;
    # the gradients make several sources land on one destination and leave others untouched
    Uw = (-3.0 + 2.5 * np.sin(xx / 9.0 + yy / 31.0)).astype(np.float32)
    Vw = (-1.2 + 1.1 * np.cos(yy / 7.0 - xx / 23.0)).astype(np.float32)
    w["im"], w["U"], w["V"], w["im2"] = wi, Uw, Vw, I1[60:180, 200:350].copy()
    arg = wi.copy()
    (r, _o) = _quiet(G.updateNextPyramidalLevel, arg, np.zeros((120, 150), np.float32), w["im2"].copy(), Uw.copy(), Vw.copy(),
                     None, None, True, False, False)
    assert r[0] is arg                                                     # warped in place
    w["out_im1"], w["out_im2"], w["out_Uacc"], w["out_Vacc"] = r[0], r[1], r[2], r[3]
    # whole driver, two levels, adapter defaults off so that the caller's biLinear=False applies
    a0, a1 = I0[:256, :256].copy(), I1[:256, :256].copy()
    (r, _o) = _quiet(G.genericPyramidalOpticalFlow, a0, a1, 3.4, H.HSOpticalFlowAlgoAdapter([21, 45], 30, False), 2, 1,
                     None, None, True, False, True, False)
    w["drv_U"], w["drv_V"], w["drv_im1_after"] = r[0], r[1], a0
    w["drv_im1_changed"] = np.int64(np.count_nonzero(a0 != I0[:256, :256]))
    print("shift warp driver:", stats(r[0]), stats(r[1]), int(w["drv_im1_changed"]))
    np.savez_compressed(os.path.join(OUT, "shiftwarp.npz"), **w)

    # ---- Farneback host constant tables (pure Python parts only) ----------------------
    f = {}
    try:
        import GaussianKernelBitExact as GK
        import Farneback_PyCL as FB
        for n, s in ((3, 0), (3, 0.5), (5, 0), (33, 4.95), (33, 16.5 * 0.3), (7, 1.2), (9, 0)):
            _, kb = GK.getGaussianKernelBitExact(n, s)
            f["bitexact_%d_%g" % (n, s)] = np.asarray(kb, dtype=np.float64).ravel()
            f["bitexact_f32_%d_%g" % (n, s)] = np.float32(kb).ravel()
        for n, s in ((7, 1.5), (5, 1.1), (5, 1.2), (7, 0.0)):
            o = FB.Farneback_PyCL.__new__(FB.Farneback_PyCL)
            o.polyN, o.polySigma = n, s
            r = o.FarnebackPrepareGaussian()
            for i, a in enumerate(r):
                f["prep_%d_%g_%d" % (n, s, i)] = np.asarray(a)
    except Exception as e:                                   # pragma: no cover
        print("farneback host tables not captured:", repr(e))
    np.savez_compressed(os.path.join(OUT, "farneback_tables.npz"), **f)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()

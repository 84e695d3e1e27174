"""CPU-side checks of the boundary: the library loads, exports what include/ofri.h declares, reports
errors as codes + messages, and the host-only helpers agree with the oracle.  No GPU work."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

import _ofri  # noqa: E402
from oracle import gauss  # noqa: E402


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_ofri.LIB_PATH):
        entry.build()
    lib = _ofri.lib()
    names = entry.declared_symbols()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    unbound = [n for n in names if n not in _ofri._PROTOS]
    assert not unbound, "ctypes prototypes missing for %s" % unbound
    assert lib.ofri_version() >= 100


def test_error_convention_without_gpu_or_bad_args():
    lib = _ofri.lib()
    n = C.c_int(-1)
    lib.ofri_device_count(C.byref(n))
    assert n.value >= 0
    h = C.c_void_p()
    rc = lib.ofri_ctx_create(10 ** 6, None, C.byref(h))
    assert rc < 0 and not h.value
    assert lib.ofri_last_error()
    assert lib.ofri_ctx_sync(None) == -1            # OFRI_EINVAL, no crash
    assert b"null" in lib.ofri_last_error()
    if _ofri.device_count() == 0:
        with pytest.raises(_ofri.OfriError):
            _ofri.Context(0)


@pytest.mark.parametrize("sigma,K", [(3.4, 3), (2.0, 3), (0.48, 5), (1.0, 7), (1.8, 73), (0.7, 9), (5.3, 33), (2.5, 127)])
def test_gauss_taps_match_oracle(sigma, K):
    import gaussian_filter as gf
    np.testing.assert_array_equal(gf.prepareGaussianKernel(sigma, K), gauss.prepare_kernel(sigma, K))


def test_adapter_protocol_surface():
    from HornSchunck import HSOpticalFlowAlgoAdapter
    ad = HSOpticalFlowAlgoAdapter([1.0, 2.0], 10)
    assert ad.getAlgoName() == 'Horn-Schunck'
    assert ad.hasGenericPyramidalDefaults() is True
    assert ad.getGenericPyramidalDefaults() == {'warping': True, 'biLinear': True, 'scaling': True}
    assert HSOpticalFlowAlgoAdapter([1.0], 5, provideGenericPyramidalDefaults=False).hasGenericPyramidalDefaults() is False
    from GenericPyramidalOpticalFlowWrapper import GenericPyramidalOpticalFlowWrapper
    w = GenericPyramidalOpticalFlowWrapper(ad, filter_sigma=3.4, pyr_levels=2)
    assert w.pyr_levels == 2 and w.k_levels == 1 and w.warping is True and w.pyramidal_scaling is False


def test_synthetic_pair_is_deterministic_and_piv_like():
    from piv_synth import synth_piv, poiseuille_u
    a, b = synth_piv(128, 160, seed=5)
    a2, _ = synth_piv(128, 160, seed=5)
    np.testing.assert_array_equal(a, a2)
    assert a.dtype == np.uint8 and a.shape == (128, 160) and a.max() > 100 and 0.2 < (a > 0).mean() < 0.7
    u = poiseuille_u(128)
    assert abs(u[0]) < 1e-12 and abs(u.min() + 4.0) < 1e-3


def test_farneback_host_tables_match_reference_fixtures():
    """Host-side constant tables of the Farneback adapter need no GPU; build one without a context."""
    from conftest import load_gold
    from GaussianKernelBitExact import getGaussianKernelBitExact
    import Farneback_PyCL as FB
    f = load_gold("farneback_tables")
    for n, s in ((3, 0), (3, 0.5), (5, 0), (33, 4.95), (33, 16.5 * 0.3), (7, 1.2), (9, 0)):
        tot, k = getGaussianKernelBitExact(n, s)
        np.testing.assert_array_equal(np.float64(k), f["bitexact_%d_%g" % (n, s)])
    for n, s in ((7, 1.5), (5, 1.1), (5, 1.2), (7, 0.0)):
        o = FB.Farneback_PyCL.__new__(FB.Farneback_PyCL)
        o.polyN, o.polySigma = n, s
        for i, a in enumerate(o.FarnebackPrepareGaussian()):
            np.testing.assert_array_equal(np.asarray(a), f["prep_%d_%g_%d" % (n, s, i)])
    o = FB.Farneback_PyCL.__new__(FB.Farneback_PyCL)
    o.polyN, o.polySigma = 7, 1.5
    o.setPolynomialExpansionConsts()
    assert o.matrixG.shape == (1, 8) and o.matrixIG.dtype == np.float32
    o.setGaussianBlurKernel(33, 33 / 2 * 0.3)
    np.testing.assert_array_equal(o.matrixGKernel[0], np.float32(f["bitexact_33_4.95"])[16:])
    with pytest.raises(Exception):
        FB.Farneback_PyCL(windowSize=32)


def test_methods_report_table_names():
    """The ten rows of the reference's benchmark (benchmark_of_methods.py:143-148,197-201,251-255)."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import methods_report
    assert [r[0] for r in methods_report.method_table()] == [
        "HS_Fs0_0", "HS_Fs3_4", "HS_Fs3_4_PyrLvls2", "LiuSE_HS_Fs3_4_PyrLvls2",
        "LK_Fs2_0", "LK_Fs2_0_PyrLvls2", "LiuSE_LK_Fs2_0_PyrLvls2",
        "FB_Fs0_0", "FB_Fs0_0_PyrLvls2", "LiuSE_FB_Fs0_0_PyrLvls2"]
    assert [r[2:] for r in methods_report.method_table()][:4] == [(0.0, 1, False), (3.4, 1, False), (3.4, 2, False), (3.4, 2, True)]

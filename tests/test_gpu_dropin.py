"""The reference's example scripts, replayed against this package: bare-module imports from one directory
on sys.path, the scripts' own call sequence and their .mat writer (examples/PyHSchunck_Fs3_4_PyrLvls2.py,
examples/LiuSE_PyHSchunck_Fs3_4_PyrLvls2.py, examples/Farneback_Fs0_0_PyrLvls2.py,
examples/denseLK_Fs2_0_PyrLvls2.py).  The scripts themselves and skimage do not travel to the GPU box, so
the bundled frame pair comes from the committed fixture."""
import os

import numpy as np
import pytest

from conftest import load_gold, rel_l2

pytestmark = pytest.mark.gpu


def save_flow(U, V, filename):
    """Same record layout as the examples' save_flow (examples/PyHSchunck_Fs3_4.py:35-51)."""
    import scipy.io
    margins = {'top': 0, 'left': 0, 'bottom': 0, 'right': 0}
    results = {'u': U, 'v': V, 'iaWidth': 1, 'iaHeight': 1, 'margins': margins}
    parameters = {'overlapFactor': 1.0, 'imageHeight': np.size(U, 0), 'imageWidth': np.size(U, 1)}
    scipy.io.savemat(filename, mdict={'velocities': results, 'parameters': parameters})


def test_example_scripts_call_sequence(tmp_path, pair):
    from GenericPyramidalOpticalFlow import genericPyramidalOpticalFlow
    from PhysicsBasedOpticalFlowLiuShen import LiuShenOpticalFlowAlgoAdapter
    from HornSchunck import HSOpticalFlowAlgoAdapter
    from Farneback_PyCL import Farneback_PyCL
    from denseLucasKanade_PyCL import denseLucasKanade_PyCl
    import scipy.io
    Iold, Inew = pair
    d = load_gold("driver")
    # PyHSchunck_Fs3_4_PyrLvls2.py with Niter as in the fixture (the script uses 600)
    hsAdapter = HSOpticalFlowAlgoAdapter([21, 45], 100)
    [U, V] = genericPyramidalOpticalFlow(Iold, Inew, 3.4, hsAdapter, 2, 1, None)
    assert hsAdapter.alphas == []
    assert rel_l2(np.stack([U[::2, ::2], V[::2, ::2]]), np.stack([d["hs_L2_U"], d["hs_L2_V"]])) < 1e-4
    fn = os.path.join(str(tmp_path), 'PyHSchunck_Fs3_4_PyrLvls2.mat')
    save_flow(U, V, fn)
    m = scipy.io.loadmat(fn)
    np.testing.assert_array_equal(m['velocities']['u'][0, 0], U)
    assert int(np.ravel(m['parameters']['imageWidth'][0, 0])[0]) == 512
    # the same script with its own Niter = 600 (examples/PyHSchunck_Fs3_4_PyrLvls2.py:132)
    d6 = load_gold("driver_n600")
    [U, V] = genericPyramidalOpticalFlow(Iold, Inew, 3.4, HSOpticalFlowAlgoAdapter([21, 45], 600), 2, 1, None)
    assert rel_l2(np.stack([U[::2, ::2], V[::2, ::2]]), np.stack([d6["hs_L2_n600_U"], d6["hs_L2_n600_V"]])) < 1e-4
    # LiuSE_PyHSchunck_Fs3_4_PyrLvls2.py
    [U, V] = genericPyramidalOpticalFlow(Iold, Inew, 3.4, HSOpticalFlowAlgoAdapter([21, 45], 100), 2, 1, 0.48,
                                         LiuShenOpticalFlowAlgoAdapter(5))
    assert rel_l2(np.stack([U[::2, ::2], V[::2, ::2]]), np.stack([d["hs_L2_ls_U"], d["hs_L2_ls_V"]])) < 1e-4
    # Farneback_Fs0_0_PyrLvls2.py: against the oracle's run of the same call (the reference's OpenCL kernels cannot run
    # where the fixtures are made: parity unpinned by reference outputs), and the Poiseuille profile of the bundled pair
    from oracle import farneback as ofb, liushen as ols, lk as olk, pyramid as opyr
    [U, V] = genericPyramidalOpticalFlow(Iold, Inew, 0.0, Farneback_PyCL(platformID=0), 2, 1, None)
    Ur, Vr = opyr.pyramidal_flow(Iold.copy(), Inew.copy(), 0.0, ofb.FarnebackAdapter(), 2, 1, None)
    assert rel_l2(np.stack([U, V]), np.stack([Ur, Vr])) < 1e-4
    rows = U[:, 96:416].mean(axis=1)
    assert rows[256] < -3.0 and abs(rows[8]) < 1.5 and abs(V[64:448, 96:416].mean()) < 0.2
    # denseLK_Fs2_0_PyrLvls2.py (dense LK refined by Liu-Shen(10), no warping) on a crop the oracle can hold in memory
    lkAdapter = denseLucasKanade_PyCl(platformID=0, deviceID=0, Niter=5, halfWindow=13)
    c0, c1 = Iold[192:320, 192:320].copy(), Inew[192:320, 192:320].copy()
    [U, V] = genericPyramidalOpticalFlow(c0.copy(), c1.copy(), 2.0, lkAdapter, 2, 1, 0.48, LiuShenOpticalFlowAlgoAdapter(10), warping=False)
    Ur, Vr = opyr.pyramidal_flow(c0.copy(), c1.copy(), 2.0, olk.DenseLKAdapter(5, 13), 2, 1, 0.48, ols.LiuShenAdapter(10), warping=False)
    assert U.shape == (128, 128) and rel_l2(np.stack([U, V]), np.stack([Ur, Vr])) < 1e-4
    [U, V] = genericPyramidalOpticalFlow(Iold[128:384, 128:384].copy(), Inew[128:384, 128:384].copy(), 2.0,
                                         denseLucasKanade_PyCl(platformID=0, deviceID=0, Niter=5, halfWindow=13), 2, 1,
                                         0.48, LiuShenOpticalFlowAlgoAdapter(10), warping=False)
    assert U.shape == (256, 256) and np.isfinite(U).all() and U[100:156, 60:196].mean() < -1.0


def test_ingestion_matches_host_conversion():
    import _ofri
    from piv_ingest import PairFeeder, upload_pair
    ctx = _ofri.context()
    rng = np.random.default_rng(0)
    a8 = rng.integers(0, 256, (70, 90), dtype=np.uint8); b8 = rng.integers(0, 256, (70, 90), dtype=np.uint8)
    a16 = rng.integers(0, 65536, (70, 90), dtype=np.uint16); b16 = rng.integers(0, 65536, (70, 90), dtype=np.uint16)
    d0, d1 = upload_pair(ctx, a8, b8, range_rule=True)                 # max <= 255: untouched
    np.testing.assert_array_equal(d0.numpy(), a8.astype(np.float32)); np.testing.assert_array_equal(d1.numpy(), b8.astype(np.float32))
    d0, d1 = upload_pair(ctx, a16, b16, range_rule=True)               # 16-bit data: (x / 65535 * 255) in float32
    np.testing.assert_array_equal(d0.numpy(), (a16.astype(np.float32) / 65535.0 * 255.0).astype(np.float32))
    np.testing.assert_array_equal(d1.numpy(), (b16.astype(np.float32) / 65535.0 * 255.0).astype(np.float32))
    pairs = [(rng.integers(0, 256, (40, 52), dtype=np.uint8), rng.integers(0, 256, (40, 52), dtype=np.uint8)) for _ in range(5)]
    got = [(x.numpy(), y.numpy()) for x, y in PairFeeder(ctx, iter(pairs))]
    assert len(got) == 5
    for (x, y), (p, q) in zip(got, pairs):
        np.testing.assert_array_equal(x, p.astype(np.float32)); np.testing.assert_array_equal(y, q.astype(np.float32))
    assert list(PairFeeder(ctx, iter([]))) == []


def test_methods_report_runs_all_ten_configurations(tmp_path):
    """The reference's benchmark report (benchmark_of_methods.py:142-331) on a 160x160 crop of the bundled pair."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import methods_report
    out = os.path.join(str(tmp_path), "report")
    res = methods_report.main(["--crop", "160", "--out", out, "--mat", "--plots", "--dpi", "30", "--no-warm"])
    assert [r[0] for r in res] == [m[0] for m in methods_report.method_table()] and len(res) == 10
    for name, dt, U, V in res:
        assert U.shape == (160, 160) and np.isfinite(U).all() and np.isfinite(V).all(), name
        assert os.path.exists(os.path.join(out, name + ".mat")) and os.path.exists(os.path.join(out, name + ".png"))
    lines = open(os.path.join(out, "benchmark_summary.txt")).read().splitlines()
    assert lines[3].split()[:2] == ["Method", "Time"] and sum(1 for l in lines if l.startswith(("HS_", "LK_", "FB_", "LiuSE_"))) == 10
    assert os.path.exists(os.path.join(out, "execution_time_comparison.png"))
    by = {r[0]: r for r in res}
    assert by["HS_Fs3_4_PyrLvls2"][2][40:120, 30:130].mean() < -0.5      # flow of the bundled pair points in -x

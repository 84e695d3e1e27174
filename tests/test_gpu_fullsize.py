"""The BASELINE configurations at (or near) their own sizes against the CPU oracle, end to end.

The kernels that only run on large fields - the row-streaming Horn-Schunck pipeline and its deeper tail launch, the
fused warp + pre-filter + derivative kernel of a level, the spline and resize kernels at 2048 -> 4096 - are compared
here with the oracle itself, not with other HIP kernels.  The oracle legs (5 to 100 s of NumPy/SciPy each) run in
CPU-only worker processes side by side while this process drives the GPU.

Tolerance: relative L2 of [U; V] below 1e-4, the bar the task states for float32 flow fields; the exact Horn-Schunck
mode is compared bit for bit."""
import multiprocessing as mp

import numpy as np
import pytest

from conftest import rel_l2
from _fullsize_oracle import frames, oracle_flow

pytestmark = pytest.mark.gpu

JOBS = [("hs2", 4096, 0),        # the headline configuration exactly as bench.py runs it
        ("hs2", 2048, 1),        # the same pipeline a size down: exact mode bit for bit, fast mode to tolerance
        ("fb2", 1024, 0),        # Farneback 2-level pyramid
        ("fb2ls", 2048, 0),      # Farneback + Liu-Shen(10) at the size that configuration is named on
        ("hs2ls", 2048, 2)]      # one pair of the 512-pair batch configuration


@pytest.fixture(scope="module")
def oracle():
    pool = mp.get_context("spawn").Pool(len(JOBS))
    pending = {job: pool.apply_async(oracle_flow, (job,)) for job in JOBS}

    def get(job):
        _, U, V = pending[job].get(timeout=600)
        return U, V
    yield get
    pool.terminate()
    pool.join()


@pytest.fixture(scope="module")
def ctx():
    import _ofri
    if _ofri.device_count() < 1:
        pytest.fail("GPU tests need a HIP device and libofri_hip.so; no CPU fallback exists")
    return _ofri.context()


def _flow(ctx, n, seed, *driver_args):
    from GenericPyramidalOpticalFlow import pyramidal_flow_device
    a, b = frames(n, seed)
    U, V = pyramidal_flow_device(ctx.upload(a), ctx.upload(b), *driver_args)
    return U.numpy(), V.numpy()


def _check(tag, U, V, Ur, Vr, tol=1e-4, against="oracle"):
    r = rel_l2(np.stack([U, V]), np.stack([Ur, Vr]))
    worst = float(max(np.abs(U - Ur).max(), np.abs(V - Vr).max()))
    print("%s: rel L2 vs %s %.3e, largest difference %.3e px, |flow| rms %.3f px" %
          (tag, against, r, worst, float(np.sqrt((Ur.astype(np.float64) ** 2 + Vr.astype(np.float64) ** 2).mean()))))
    assert np.isfinite(U).all() and np.isfinite(V).all()
    assert r < tol, (tag, r)
    return r


def test_hs_two_level_2048_exact_bits_and_fast_tolerance(ctx, oracle):
    from HornSchunck import HSOpticalFlowAlgoAdapter
    Ur, Vr = oracle(("hs2", 2048, 1))
    ctx.set_hs_fuse(-1)
    try:
        U, V = _flow(ctx, 2048, 1, 3.4, HSOpticalFlowAlgoAdapter([21, 45], 100), 2, 1, None)
    finally:
        ctx.set_hs_fuse(0)
    np.testing.assert_array_equal(U, Ur)      # resize, spline, warp, pre-filter at these sizes and the exact sweeps
    np.testing.assert_array_equal(V, Vr)
    U, V = _flow(ctx, 2048, 1, 3.4, HSOpticalFlowAlgoAdapter([21, 45], 100), 2, 1, None)
    _check("HS 2-level 2048^2 fast mode", U, V, Ur, Vr)


def test_headline_configuration_4096_vs_oracle(ctx, oracle):
    """Fast mode, the launches bench.py times: level kernel, pipeline launches of 12 sweeps, the 16-deep tail with its
    accumulating stores, spline 2048 -> 4096, resize 4096 -> 2048."""
    import _ofri
    from HornSchunck import HSOpticalFlowAlgoAdapter
    from GenericPyramidalOpticalFlow import pyramidal_flow_device
    from piv_synth import synth_piv
    n = 4096
    f0, f1 = synth_piv(n, n, seed=0)
    lib = _ofri.lib()
    u0, u1 = ctx.upload(f0, np.uint8), ctx.upload(f1, np.uint8)         # bench.py's own ingestion: raw frames widened on the GPU
    im0, im1 = ctx.empty((n, n)), ctx.empty((n, n))
    _ofri.check(lib.ofri_u8_to_f32(ctx.handle, u0.ptr, im0.ptr, n * n))
    _ofri.check(lib.ofri_u8_to_f32(ctx.handle, u1.ptr, im1.ptr, n * n))
    Ua, Va = pyramidal_flow_device(im0, im1, 3.4, HSOpticalFlowAlgoAdapter([21, 45], 100), 2, 1, None)
    U, V = Ua.numpy(), Va.numpy()
    # the object bench.py times: the same pair replayed from the captured graph of piv_plan.FlowPlan (at this size with the
    # 256-column strips), twice, must give the kernel-by-kernel result bit for bit
    from piv_plan import FlowPlan
    plan = FlowPlan((n, n), 3.4, lambda: HSOpticalFlowAlgoAdapter([21, 45], 100), pyramidalLevels=2, ctx=ctx, timed=True)
    try:
        plan.load(im0, im1)
        for _ in range(2):
            Ug, Vg = plan.run()
            np.testing.assert_array_equal(Ug.numpy(), U)
            np.testing.assert_array_equal(Vg.numpy(), V)
        assert len(plan.prof_records()) > 0
    finally:
        plan.close()
    Ur, Vr = oracle(("hs2", 4096, 0))
    _check("HS 2-level 4096^2 (headline)", U, V, Ur, Vr)
    # the Poiseuille profile the frames were rendered with: -4 px at the centre row, 0 at the walls
    prof = U.mean(axis=1)
    assert -4.3 < prof[n // 2] < -3.5 and abs(prof[2]) < 0.6 and float(np.abs(V).mean()) < 0.1


def test_farneback_two_level_1024_vs_oracle(ctx, oracle):
    from Farneback_PyCL import Farneback_PyCL
    U, V = _flow(ctx, 1024, 0, 0.0, Farneback_PyCL(), 2, 1, None)
    _check("Farneback 2-level 1024^2", U, V, *oracle(("fb2", 1024, 0)))


def test_farneback_two_level_4096_fast_vs_exact_and_profile(ctx):
    """The Farneback configuration at its own size (4096 x 4096, 2 levels; examples/Farneback_Fs0_0_PyrLvls2.py:73).  The NumPy
    oracle would need many minutes there, so the check is size-independent: the default fast mode (fused multiply-adds)
    against the exact mode (separately rounded, bit-identical to the oracle at the sizes the oracle is run on) below
    1e-5, and the Poiseuille profile the frames were rendered with."""
    from Farneback_PyCL import Farneback_PyCL
    n = 4096
    U, V = _flow(ctx, n, 0, 0.0, Farneback_PyCL(), 2, 1, None)
    ctx.set_fb_mode(-1)
    try:
        Ue, Ve = _flow(ctx, n, 0, 0.0, Farneback_PyCL(), 2, 1, None)
    finally:
        ctx.set_fb_mode(0)
    r = _check("Farneback 2-level 4096^2 fast mode", U, V, Ue, Ve, tol=1e-5, against="exact mode")
    assert r > 0                                          # two different arithmetic paths did run
    prof = Ue.mean(axis=1)
    assert -4.3 < prof[n // 2] < -3.5 and abs(prof[8]) < 0.8 and float(np.abs(Ve).mean()) < 0.1
    q = n // 4                                            # parabola: three quarters of the centre value at the quarter rows
    assert abs(prof[q] / prof[n // 2] - 0.75) < 0.06 and abs(prof[3 * q] / prof[n // 2] - 0.75) < 0.06


def test_farneback_plus_liushen_2048_vs_oracle(ctx, oracle):
    from Farneback_PyCL import Farneback_PyCL
    from PhysicsBasedOpticalFlowLiuShen import LiuShenOpticalFlowAlgoAdapter
    U, V = _flow(ctx, 2048, 0, 0.0, Farneback_PyCL(), 2, 1, 0.48, LiuShenOpticalFlowAlgoAdapter(10))
    _check("Farneback + Liu-Shen(10) 2-level 2048^2", U, V, *oracle(("fb2ls", 2048, 0)))


def test_batch_pair_hs_plus_liushen_2048_vs_oracle(ctx, oracle):
    from HornSchunck import HSOpticalFlowAlgoAdapter
    from PhysicsBasedOpticalFlowLiuShen import LiuShenOpticalFlowAlgoAdapter
    U, V = _flow(ctx, 2048, 2, 3.4, HSOpticalFlowAlgoAdapter([21, 45], 100), 2, 1, 0.48, LiuShenOpticalFlowAlgoAdapter(5))
    _check("HS + Liu-Shen(5) 2-level 2048^2 (batch pair)", U, V, *oracle(("hs2ls", 2048, 2)))

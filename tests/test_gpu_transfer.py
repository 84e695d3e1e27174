"""Host <-> device copies of the Python host layer: results copied back into pooled pinned memory and
camera frames widened on the GPU must be indistinguishable, bit for bit, from the plain pageable copies."""
import ctypes as C
import gc

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def pageable_download(ctx, d):
    import _ofri
    out = np.empty(d.shape, d.dtype)
    _ofri.check(_ofri.lib().ofri_download(ctx.handle, out.ctypes.data, d.ptr, d.nbytes))
    return out


def test_large_results_come_back_in_pinned_memory_bit_identical():
    import _ofri
    ctx = _ofri.context(0)
    rng = np.random.default_rng(3)
    src = rng.standard_normal((1536, 1024)).astype(np.float32)      # 6 MB: above the pinned threshold
    src.ravel()[:4] = [np.nan, np.inf, -0.0, np.float32(1e-42)]      # payload bits must survive
    d = ctx.upload(src)
    a = d.numpy()
    assert isinstance(a.base, np.ndarray) and a.flags.writeable and a.flags.c_contiguous
    ref = pageable_download(ctx, d)
    assert a.tobytes() == ref.tobytes() == src.tobytes()
    # the array behaves like any other: views keep it alive, writes stay local
    v = a[100:200, 5:50]
    a[...] = 7.0
    del a
    gc.collect()
    assert np.all(v == 7.0)
    assert d.numpy().tobytes() == src.tobytes()
    # the block is reused once the last view is gone
    del v
    gc.collect()
    idle = _ofri._pinned._idle
    assert idle >= src.nbytes
    b = d.numpy()
    assert _ofri._pinned._idle == idle - (src.nbytes + _ofri._PinnedPool.GRAIN - 1) // _ofri._PinnedPool.GRAIN * _ofri._PinnedPool.GRAIN
    assert b.tobytes() == src.tobytes()


def test_small_and_float64_results():
    import _ofri
    ctx = _ofri.context(0)
    small = np.arange(12, dtype=np.float32).reshape(3, 4)
    assert np.array_equal(ctx.upload(small).numpy(), small)
    big64 = np.random.default_rng(0).standard_normal((600, 700))
    d = ctx.upload(big64, np.float64)
    assert d.numpy().dtype == np.float64 and d.numpy().tobytes() == big64.tobytes()


def test_results_beyond_the_live_cap_fall_back_to_pageable(monkeypatch):
    import _ofri
    ctx = _ofri.context(0)
    src = np.random.default_rng(1).random((1024, 1024)).astype(np.float32)
    d = ctx.upload(src)
    monkeypatch.setattr(_ofri._pinned, "MAX_LIVE_BYTES", 0)
    a = d.numpy()
    assert a.base is None and a.tobytes() == src.tobytes()


@pytest.mark.parametrize("dtype, top", [(np.uint8, 255), (np.uint16, 65535)])
def test_camera_frames_are_widened_on_the_gpu_exactly(dtype, top):
    import _ofri
    ctx = _ofri.context(0)
    rng = np.random.default_rng(5)
    f = rng.integers(0, top + 1, size=(513, 771), dtype=dtype)
    f.ravel()[:2] = [0, top]
    got = ctx.upload(f).numpy()
    assert got.dtype == np.float32 and got.tobytes() == f.astype(np.float32).tobytes()
    # non-contiguous input (a crop of a larger frame) goes the same way
    crop = f[3:500, 7:700]
    assert ctx.upload(crop).numpy().tobytes() == crop.astype(np.float32).tobytes()


def test_driver_accepts_uint8_frames_like_float32_frames(pair):
    from GenericPyramidalOpticalFlow import genericPyramidalOpticalFlow
    from HornSchunck import HSOpticalFlowAlgoAdapter
    a, b = pair
    a8, b8 = np.clip(a, 0, 255).astype(np.uint8), np.clip(b, 0, 255).astype(np.uint8)
    big = np.tile(a8, (2, 2)), np.tile(b8, (2, 2))      # large enough for the raw-frame path
    U1, V1 = genericPyramidalOpticalFlow(big[0], big[1], 3.4, HSOpticalFlowAlgoAdapter([21, 45], 20), 2, 1, None)
    U2, V2 = genericPyramidalOpticalFlow(big[0].astype(np.float32), big[1].astype(np.float32), 3.4,
                                         HSOpticalFlowAlgoAdapter([21, 45], 20), 2, 1, None)
    assert U1.tobytes() == U2.tobytes() and V1.tobytes() == V2.tobytes()


# ------------------------------------------------------------------------------ captured pipelines

def test_flow_plan_times_its_sweep_launches_at_every_replay():
    """A plan captured with timed=True holds the Horn-Schunck sweep launches as timed sections of its graph: every
    replay times them again (event nodes), the results stay the driver's bits, and the sections add up to less than
    the replay itself."""
    import _ofri
    from GenericPyramidalOpticalFlow import genericPyramidalOpticalFlow
    from HornSchunck import HSOpticalFlowAlgoAdapter
    from piv_plan import FlowPlan
    from piv_synth import synth_piv
    ctx = _ofri.context(0)
    shape = (2048, 2048)
    plan = FlowPlan(shape, 3.4, lambda: HSOpticalFlowAlgoAdapter([21, 45], 40), pyramidalLevels=2, timed=True)
    seen = []
    for seed in (1, 2):
        a, b = synth_piv(shape[0], shape[1], seed=seed)
        plan.load(a, b)
        e0, e1 = ctx.event(), ctx.event()
        ctx.record(e0)
        U, V = plan.run()
        ctx.record(e1)
        whole = ctx.elapsed_ms(e0, e1)
        recs = plan.prof_records()
        # two levels; the full-resolution one goes through the pipeline (launches of 12 sweeps and a deeper one),
        # the half-resolution one through the tile kernel
        assert sorted(set((r[3], r[4]) for r in recs)) == [(1024, 1024), (2048, 2048)]
        assert sum(r[2] for r in recs) == 80 and all(r[0] > 0 for r in recs)
        assert sum(r[0] for r in recs) < whole
        seen.append([r[0] for r in recs])
        Ur, Vr = genericPyramidalOpticalFlow(a.astype(np.float32), b.astype(np.float32), 3.4,
                                             HSOpticalFlowAlgoAdapter([21, 45], 40), 2, 1, None)
        assert U.numpy().tobytes() == Ur.tobytes() and V.numpy().tobytes() == Vr.tobytes()
    assert ctx.prof_records() == []                     # the sections belong to the graph, not to the context
    plan.close()

@pytest.mark.parametrize("shape,levels", [((256, 256), 2), ((200, 312), 1), ((2048, 2048), 2)])
def test_flow_plan_replays_the_driver_bit_for_bit(shape, levels):
    """FlowPlan (the driver's launches captured into a HIP graph) against genericPyramidalOpticalFlow on several
    pairs in a row, with other allocations and work on the same context in between."""
    import _ofri
    from GenericPyramidalOpticalFlow import genericPyramidalOpticalFlow
    from HornSchunck import HSOpticalFlowAlgoAdapter
    from piv_plan import FlowPlan
    from piv_synth import synth_piv
    ctx = _ofri.context(0)
    alphas = [21, 45][:levels]
    plan = FlowPlan(shape, 3.4, lambda: HSOpticalFlowAlgoAdapter(list(alphas), 30), pyramidalLevels=levels)
    rng = np.random.default_rng(7)
    for seed in (1, 2, 3):
        a, b = synth_piv(shape[0], shape[1], seed=seed)
        # memory handed out after the capture must not be the graph's scratch: fill some and check it afterwards
        guard_host = [rng.standard_normal(shape).astype(np.float32) for _ in range(4)]
        guard = [ctx.upload(g) for g in guard_host]
        U, V = plan(a, b)
        Ur, Vr = genericPyramidalOpticalFlow(a.astype(np.float32), b.astype(np.float32), 3.4,
                                             HSOpticalFlowAlgoAdapter(list(alphas), 30), levels, 1, None)
        assert U.tobytes() == Ur.tobytes() and V.tobytes() == Vr.tobytes()
        for g, h in zip(guard, guard_host):
            assert g.numpy().tobytes() == h.tobytes()
    plan.close()


def test_flow_plan_with_farneback_and_dense_lk(pair):
    """Their per-call constants (filter taps, expansion tables, window parameters) travel as kernel arguments, so
    the other two main adapters replay from a captured graph too."""
    from GenericPyramidalOpticalFlow import genericPyramidalOpticalFlow
    from Farneback_PyCL import Farneback_PyCL
    from denseLucasKanade_PyCL import denseLucasKanade_PyCl
    from piv_plan import FlowPlan
    for make, crop in ((lambda: Farneback_PyCL(), 256), (lambda: denseLucasKanade_PyCl(Niter=3, halfWindow=6), 96)):
        I0, I1 = pair[0][:crop, :crop].copy(), pair[1][:crop, :crop].copy()
        plan = FlowPlan(I0.shape, 2.0, make, pyramidalLevels=2)
        for k in range(2):
            a, b = np.roll(I0, 3 * k, axis=0), np.roll(I1, 3 * k, axis=0)
            U, V = plan(a, b)
            Ur, Vr = genericPyramidalOpticalFlow(a.copy(), b.copy(), 2.0, make(), 2, 1, None)
            assert U.tobytes() == Ur.tobytes() and V.tobytes() == Vr.tobytes()
        plan.close()


def test_flow_plan_with_liushen_refinement(pair):
    """Liu-Shen keeps its stopping rule on the GPU, so Horn-Schunck + Liu-Shen (the reference's LiuSE examples) is
    capturable too."""
    from GenericPyramidalOpticalFlow import genericPyramidalOpticalFlow
    from HornSchunck import HSOpticalFlowAlgoAdapter
    from PhysicsBasedOpticalFlowLiuShen import LiuShenOpticalFlowAlgoAdapter
    from piv_plan import FlowPlan
    a, b = pair[0][:192, :256].copy(), pair[1][:192, :256].copy()
    plan = FlowPlan(a.shape, 3.4, lambda: HSOpticalFlowAlgoAdapter([21, 45], 20), pyramidalLevels=2, FILTER_OPT=0.48,
                    optionalOFlowAlgoAdapter=LiuShenOpticalFlowAlgoAdapter(5))
    for _ in range(2):
        U, V = plan(a, b)
        Ur, Vr = genericPyramidalOpticalFlow(a.copy(), b.copy(), 3.4, HSOpticalFlowAlgoAdapter([21, 45], 20), 2, 1, 0.48,
                                             LiuShenOpticalFlowAlgoAdapter(5))
        assert U.tobytes() == Ur.tobytes() and V.tobytes() == Vr.tobytes()
    plan.close()


def test_flow_plan_refuses_sequences_that_wait_for_the_gpu():
    """The optical-flow-equation warp (biLinear=False) reads its range check back on the host."""
    import _ofri
    from HornSchunck import HSOpticalFlowAlgoAdapter
    from piv_plan import FlowPlan
    with pytest.raises(_ofri.OfriError):
        FlowPlan((64, 64), 3.4, lambda: HSOpticalFlowAlgoAdapter([21, 45], 10, False), pyramidalLevels=2, warping=True,
                 biLinear=False)
    # the context is usable afterwards
    ctx = _ofri.context(0)
    x = np.arange(12, dtype=np.float32).reshape(3, 4)
    assert np.array_equal(ctx.upload(x).numpy(), x)


def test_plan_pool_keeps_pair_order_and_bits():
    from GenericPyramidalOpticalFlow import genericPyramidalOpticalFlow
    from HornSchunck import HSOpticalFlowAlgoAdapter
    from piv_plan import PlanPool
    from piv_synth import synth_piv
    shape = (160, 224)
    pairs = [synth_piv(shape[0], shape[1], seed=s) for s in range(5)]
    pool = PlanPool(shape, 3.4, lambda: HSOpticalFlowAlgoAdapter([21, 45], 20), in_flight=3, pyramidalLevels=2)
    got = list(pool.flows(pairs))
    pool.close()
    assert len(got) == len(pairs)
    for (a, b), (U, V) in zip(pairs, got):
        Ur, Vr = genericPyramidalOpticalFlow(a.astype(np.float32), b.astype(np.float32), 3.4,
                                             HSOpticalFlowAlgoAdapter([21, 45], 20), 2, 1, None)
        assert U.tobytes() == Ur.tobytes() and V.tobytes() == Vr.tobytes()


def test_stream_flows_overlaps_copies_and_keeps_bits(pair):
    """Upload of the next pair, compute and download of the last result on three streams: same bits, same order
    as one pair at a time."""
    import _ofri
    from piv_ingest import stream_flows
    from HornSchunck import HSOpticalFlowAlgoAdapter
    from GenericPyramidalOpticalFlow import pyramidal_flow_device
    ctx = _ofri.context(0)
    I0, I1 = (np.clip(x, 0, 255).astype(np.uint8) for x in pair)
    pairs = [(np.roll(I0, k, axis=1), np.roll(I1, k, axis=1)) for k in range(5)]

    def flow(a, b):
        return pyramidal_flow_device(a, b, 3.4, HSOpticalFlowAlgoAdapter([21, 45], 20), 2, 1, None)

    want = []
    for f0, f1 in pairs:
        U, V = flow(ctx.upload(f0), ctx.upload(f1))
        want.append((pageable_download(ctx, U), pageable_download(ctx, V)))
    got = list(stream_flows(ctx, iter(pairs), flow))
    assert len(got) == len(want)
    for (U, V), (Ur, Vr) in zip(got, want):
        assert U.tobytes() == Ur.tobytes() and V.tobytes() == Vr.tobytes()
    assert list(stream_flows(ctx, iter([]), flow)) == []
    # a consumer that stops after the first result leaves nothing behind that breaks the next stream
    gen = stream_flows(ctx, iter(pairs), flow)
    U, V = next(gen)
    assert U.tobytes() == want[0][0].tobytes()
    gen.close()
    got = list(stream_flows(ctx, iter(pairs[:2]), flow))
    assert got[1][1].tobytes() == want[1][1].tobytes()
    # the same from two alternating captured plans
    from piv_plan import StreamedPlan
    sp = StreamedPlan(I0.shape, 3.4, lambda: HSOpticalFlowAlgoAdapter([21, 45], 20), ctx=ctx, pyramidalLevels=2)
    got = [(U.copy(), V.copy()) for U, V in sp.flows(iter(pairs))]
    sp.close()
    assert len(got) == len(want)
    for (U, V), (Ur, Vr) in zip(got, want):
        assert U.tobytes() == Ur.tobytes() and V.tobytes() == Vr.tobytes()


def test_blocks_a_captured_graph_writes_never_reach_the_pool_while_it_lives():
    """A block handed out during a capture and still held at ofri_graph_end is written by every replay: freeing it
    later must not hand it to anybody else before the graph is destroyed."""
    import ctypes as C
    import _ofri
    lib = _ofri.lib()
    ctx = _ofri.Context(0)
    try:
        n = 3 << 20
        src = ctx.upload(np.full(n // 4, 7.0, np.float32))
        warm = ctx.empty((n // 4,)); del warm                   # one block of that size in the pool
        ctx.graph_begin()
        held = ctx.empty((n // 4,))                             # handed out during the capture, kept past its end
        _ofri.check(lib.ofri_copy(ctx.handle, held.ptr, src.ptr, n))
        g = ctx.graph_end()
        p_held = held.ptr
        del held                                                # freed while the graph lives
        other = [ctx.empty((n // 4,)) for _ in range(3)]
        assert all(o.ptr != p_held for o in other)
        for o in other:
            _ofri.check(lib.ofri_zero(ctx.handle, o.ptr, n))
        g.launch()                                              # writes the held block: must not land in `other`
        ctx.sync()
        assert all(float(o.numpy().max()) == 0.0 for o in other)
        del other
        g.close()                                               # now the block is ordinary pool memory again
        again = [ctx.empty((n // 4,)) for _ in range(5)]
        assert any(a.ptr == p_held for a in again)
    finally:
        ctx.close()

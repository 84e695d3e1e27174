"""Multi-rank host logic on CPU: world_size 2 and 3 over gloo (no GPU work)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "opticalflow-ri_amd")]

from piv_batch import owner, partition  # noqa: E402


def test_partition_is_contiguous_and_balanced():
    assert partition(512, 8) == [(64 * r, 64 * (r + 1)) for r in range(8)]
    assert partition(5, 2) == [(0, 3), (3, 5)]
    assert partition(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
    assert partition(0, 3) == [(0, 0)] * 3
    for n in (1, 7, 64, 513):
        for w in (1, 2, 3, 8):
            p = partition(n, w)
            assert p[0][0] == 0 and p[-1][1] == n and all(a[1] == b[0] for a, b in zip(p, p[1:]))
            sizes = [b - a for a, b in p]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
            assert all(p[owner(i, n, w)][0] <= i < p[owner(i, n, w)][1] for i in range(n))
    with pytest.raises(ValueError):
        partition(3, 0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_pairs, q):
    import torch
    import torch.distributed as dist
    from piv_batch import gather_flows, max_over_ranks, partition
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a, b = partition(n_pairs, world)[rank]
        # stand-in for the per-pair pipeline: a field that encodes the pair index
        local = torch.stack([torch.full((2, 4, 6), float(p)) + torch.arange(6.0) / 100 for p in range(a, b)]) \
            if b > a else torch.zeros((0, 2, 4, 6))
        out = gather_flows(local, n_pairs, dst=0)
        slow = max_over_ranks(0.5 + rank)
        if rank == 0:
            q.put((out.numpy(), slow))
        else:
            assert out is None
            q.put(("ok", slow))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_pairs", [(2, 5), (2, 4), (3, 2)])
def test_gather_of_flow_fields_over_gloo(world, n_pairs):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_pairs, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = [r for r in res if not isinstance(r[0], str)]
    assert len(full) == 1
    out, slow = full[0]
    assert out.shape == (n_pairs, 2, 4, 6)
    for p in range(n_pairs):
        np.testing.assert_allclose(out[p, 0, 0], p + np.arange(6) / 100, rtol=0, atol=1e-6)
    assert all(abs(r[1] - (0.5 + world - 1)) < 1e-9 for r in res)


def _stub_pair(p):
    """Two 'frames' that encode the pair index."""
    a = np.full((4, 6), float(p), np.float32)
    return a, a + 0.5


def _stub_flow(im0, im1):
    """Stand-in for the per-pair pipeline (no GPU here): U, V are functions of both frames."""
    return im0 + np.arange(6, dtype=np.float32) / 100, im1 * 2


def _runner_worker(rank, world, port, n_pairs, chunk, q):
    import torch.distributed as dist
    from piv_batch import run_batch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        stats = {}
        out = run_batch(n_pairs, _stub_pair, _stub_flow, ctx=None, chunk=chunk, stats=stats, shape=(4, 6))
        own = run_batch(n_pairs, _stub_pair, _stub_flow, ctx=None, chunk=chunk, gather=False)
        q.put((rank, None if out is None else out.numpy(), own.numpy(), stats["pairs"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_pairs,chunk", [(2, 5, 2), (2, 8, 2), (2, 4, 8), (3, 2, 1), (3, 7, 3)])
def test_run_batch_shards_and_gathers_over_gloo(world, n_pairs, chunk):
    """The real runner with a stub flow function: contiguous blocks, per-chunk asynchronous gathers with two
    alternating send buffers, ragged last chunks, a rank without pairs, result in pair order on rank 0."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_runner_worker, args=(r, world, port, n_pairs, chunk, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    parts = partition(n_pairs, world)
    for rank, out, own, cnt in res:
        a, b = parts[rank]
        assert cnt == b - a and own.shape[0] == b - a
        assert (out is None) == (rank != 0)
        for k, p in enumerate(range(a, b)):
            U, V = _stub_flow(*_stub_pair(p))
            np.testing.assert_array_equal(own[k, 0], U)
            np.testing.assert_array_equal(own[k, 1], V)
    out = res[0][1]
    assert out.shape == (n_pairs, 2, 4, 6)
    for p in range(n_pairs):
        U, V = _stub_flow(*_stub_pair(p))
        np.testing.assert_array_equal(out[p, 0], U)
        np.testing.assert_array_equal(out[p, 1], V)


def test_run_batch_refuses_a_rank_without_pairs_unless_the_shape_is_given():
    from piv_batch import run_batch
    out = run_batch(3, _stub_pair, _stub_flow, ctx=None, chunk=2)          # no process group: the single-rank case
    assert tuple(out.shape) == (3, 2, 4, 6)
    np.testing.assert_array_equal(out[2, 1].numpy(), _stub_flow(*_stub_pair(2))[1])
    assert tuple(run_batch(0, _stub_pair, _stub_flow, ctx=None).shape)[:2] == (0, 2)
    with pytest.raises(ValueError):
        run_batch(-1, _stub_pair, _stub_flow)

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

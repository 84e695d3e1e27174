"""Batches of independent PIV frame pairs over the GPUs of one node.

Frame pairs share nothing (no halo, no state), so pair p of a batch of n goes to rank
``owner(p)``: contiguous blocks, the first ``n % world`` ranks taking one pair more.  The only exchange is
the final gather of the flow fields to one rank; it goes through ``torch.distributed`` (RCCL on GPUs,
gloo on CPU tensors).  The reference processes one pair per script run (examples/*.py) and has no
counterpart of this module.
"""


def partition(n_pairs, world):
    """[(start, stop)] per rank: contiguous, sizes differ by at most one, earlier ranks larger."""
    if n_pairs < 0 or world < 1:
        raise ValueError("n_pairs >= 0 and world >= 1 required")
    base, extra = divmod(n_pairs, world)
    out, s = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((s, s + n))
        s += n
    return out


def owner(pair, n_pairs, world):
    for r, (a, b) in enumerate(partition(n_pairs, world)):
        if a <= pair < b:
            return r
    raise IndexError(pair)


def gather_flows(local, n_pairs, dst=0, group=None):
    """Gather per-rank flow tensors (n_local, 2, H, W) into (n_pairs, 2, H, W) on rank ``dst`` in pair
    order; returns None elsewhere.  Shards of unequal length are padded to the longest for the
    collective and trimmed afterwards."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    parts = partition(n_pairs, world)
    longest = max(b - a for a, b in parts)
    a, b = parts[rank]
    if local.shape[0] != b - a:
        raise ValueError("rank %d holds %d pairs, expected %d" % (rank, local.shape[0], b - a))
    if longest == 0:
        return local.new_zeros((0,) + tuple(local.shape[1:])) if rank == dst else None
    send = local
    if b - a < longest:
        pad = local.new_zeros((longest - (b - a),) + tuple(local.shape[1:]))
        send = torch.cat([local, pad], 0)
    send = send.contiguous()
    recv = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, recv, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([recv[r][:pb - pa] for r, (pa, pb) in enumerate(parts)], 0)


def max_over_ranks(seconds, device=None, group=None):
    """The slowest rank's time: what a whole-job throughput figure must be divided by."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())

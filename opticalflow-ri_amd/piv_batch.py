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


# ------------------------------------------------------------------------------------------ batch runner
class _Job(object):
    """torch.distributed state of one run_batch call: world, rank and where tensors live."""

    def __init__(self, group, device):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.on else 1
        self.rank = dist.get_rank(group) if self.on else 0
        self.device = device                       # torch device of the flow tensors ('cpu' for host flows)


def _one_hip_runtime():
    """torch wheels bundle their own HIP runtime under the same soname as the system's.  Whichever is loaded first
    serves both torch and libofri_hip.so - unless libofri_hip.so came first: torch asks for its copy by another file
    name, a second runtime is mapped and finds no GPU.  So: import torch before the first _ofri call."""
    paths = set()
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    paths.add(line.split()[-1])
    except OSError:
        return
    if len(paths) > 1:
        raise RuntimeError("two HIP runtimes are mapped in this process (%s): import torch before the first use of "
                           "libofri_hip.so (any _ofri call), so that both share one runtime" % ", ".join(sorted(paths)))


def torch_context(device=0, make_current=True):
    """An _ofri.Context whose stream is a torch stream (made current unless ``make_current`` is false): the library's
    kernels, torch's copies and the RCCL collectives torch enqueues are then ordered against each other.  Returns
    (ctx, torch_stream); the context remembers the stream as ``ctx.torch_stream``, and run_batch orders its copies and
    collectives against that recorded stream, never against whatever stream happens to be current."""
    import torch
    import _ofri
    _one_hip_runtime()
    torch.cuda.set_device(device)
    stream = torch.cuda.Stream(device=device)
    if make_current:
        torch.cuda.set_stream(stream)
    ctx = _ofri.Context(device, stream.cuda_stream)
    ctx.torch_stream = stream
    return ctx, stream


def flow_lanes(n, make_flow, device=0):
    """``n`` independent (ctx, flow_fn) lanes on one GPU for run_batch(lanes=...): each its own stream, pool and captured
    graph, so that n pairs are in flight at once (the launches of one pair leave gaps and tails that another pair's fill;
    2048 x 2048 Horn-Schunck + Liu-Shen: 3 lanes give 1.4 times the pairs per second of one).  ``make_flow(ctx)``
    returns the lane's flow function (e.g. lambda c: hs_liushen_flow(c, shape)).  The caller's current torch stream is
    left as it is."""
    out = []
    for i in range(max(1, int(n))):
        ctx, _ = torch_context(device, make_current=False)
        # half as many row chunks per launch once another lane fills the rest of the GPU (64 pairs of 2048 x 2048 on three
        # lanes: 2670 -> 2955 Mpix/s; a third of the chunks 2890, a quarter 2670).  Before make_flow: a captured graph
        # keeps the launch shapes of its capture
        ctx.set_share(2 if int(n) >= 2 else 1)
        out.append((ctx, make_flow(ctx)))
    return out


def hs_liushen_flow(ctx, shape, niter=100, alphas=(21, 45), h=5, FILTER=3.4, FILTER_OPT=0.48, levels=2, graph=True,
                    liushen=True):
    """flow_fn(im0, im1) -> (U, V) DeviceArrays for the batch configuration: the call of
    examples/LiuSE_PyHSchunck_Fs3_4_PyrLvls2.py:133-139 (Horn-Schunck per level, refined by Liu-Shen on the level's
    images filtered with FILTER_OPT), replayed from a captured graph when ``graph``.  The arrays returned are the
    plan's own and are overwritten by the next call."""
    from HornSchunck import HSOpticalFlowAlgoAdapter
    from PhysicsBasedOpticalFlowLiuShen import LiuShenOpticalFlowAlgoAdapter
    from GenericPyramidalOpticalFlow import pyramidal_flow_device
    opt = {"FILTER_OPT": FILTER_OPT, "optionalOFlowAlgoAdapter": LiuShenOpticalFlowAlgoAdapter(h)} if liushen else {}
    make = lambda: HSOpticalFlowAlgoAdapter(list(alphas), niter)
    if graph:
        from piv_plan import FlowPlan
        plan = FlowPlan(shape, FILTER, make, pyramidalLevels=levels, ctx=ctx, **opt)

        def flow(im0, im1):
            plan.load(im0, im1)
            return plan.run()
        flow.close = plan.close
        return flow

    def flow(im0, im1):
        return pyramidal_flow_device(im0, im1, FILTER, make(), levels, 1, opt.get("FILTER_OPT"),
                                     opt.get("optionalOFlowAlgoAdapter"))
    flow.close = lambda: None
    return flow


def _device_frames(ctx, pair):
    """A pair as float32 DeviceArrays: host arrays are uploaded (8/16-bit frames raw, widened on the GPU), resident
    uint8/uint16 DeviceArrays are widened, float32 DeviceArrays pass through."""
    import numpy as np
    import _ofri
    out = []
    for f in pair:
        if isinstance(f, _ofri.DeviceArray):
            if f.dtype == np.float32:
                out.append(f)
                continue
            d = ctx.empty(f.shape)
            widen = _ofri.lib().ofri_u8_to_f32 if f.dtype == np.uint8 else _ofri.lib().ofri_u16_to_f32
            _ofri.check(widen(ctx.handle, f.ptr, d.ptr, f.size))
            out.append(d)
        else:
            out.append(ctx.upload(f))
    return out[0], out[1]


def run_batch(n_pairs, pair_fn, flow_fn, ctx=None, group=None, dst=0, chunk=8, gather=True, stats=None, shape=None,
              lanes=None):
    """Flow fields of a batch of independent frame pairs over the ranks of ``group``.

    Rank r works on its contiguous block ``partition(n_pairs, world)[r]``: for every pair p of the block,
    ``pair_fn(p)`` gives the two frames (host arrays, or DeviceArrays already resident on this rank's GPU) and
    ``flow_fn(im0, im1)`` their flow (U, V) - DeviceArrays on ``ctx`` (float32 frames in; see hs_liushen_flow), or
    NumPy arrays from any host function (then ``ctx`` is not used and tensors stay on the CPU: the gloo rehearsal).
    No collective inside the computation.  The only exchange: every ``chunk`` pairs each rank's new fields go to
    rank ``dst`` in one asynchronous gather (RCCL on GPU tensors) that overlaps the next chunk's computation; two
    send buffers alternate, and rank ``dst`` receives straight into its result tensor.

    Returns on rank ``dst`` a float32 tensor (n_pairs, 2, H, W) in pair order ([p, 0] = U, [p, 1] = V), None on the
    other ranks; with ``gather=False`` every rank returns its own block (n_local, 2, H, W).  Without an initialised
    process group the call is the single-rank case.  ``ctx`` must have been made by torch_context() when fields are
    gathered on the GPU or further lanes are given: the gather stream and the lanes are ordered against the streams the
    contexts recorded (``ctx.torch_stream``), whatever torch's current stream is.  DeviceArrays that ``pair_fn`` hands
    out must be complete on ctx's stream when run_batch is called (frames resident before the call: every lane waits
    for ctx's stream once, at the start) or be produced on the context of the lane that will use them.  ``shape``: (H, W) of a field;
    needed only when a rank may have no pairs (more ranks than pairs), otherwise taken from the first result.
    ``stats``: optional dict that receives 'elapsed_s' (this rank's time from its first pair to its last gather) and
    'pairs' (the rank's count).  ``lanes``: further (ctx, flow_fn) pairs on the same GPU (flow_lanes); the rank's k-th
    pair then runs on lane k modulo their number, (ctx, flow_fn) being lane 0, each lane on its own stream, so that
    several pairs are in flight; the fields and their order are the same."""
    import time
    import numpy as np
    import torch
    import _ofri
    lib = _ofri.lib() if ctx is not None else None
    job = _Job(group, None)
    L = [(ctx, flow_fn)] + (list(lanes) if (lanes and ctx is not None) else [])
    if any(c is None or not callable(f) or c.device != ctx.device for c, f in L[1:]):
        raise ValueError("lanes: (ctx, flow_fn) pairs on the GPU of ctx")
    lane_streams = [getattr(c, "torch_stream", None) for c, _ in L]
    if len(L) > 1 and any(st is None for st in lane_streams):
        raise ValueError("lanes need contexts made by torch_context() / flow_lanes()")
    for st in lane_streams[1:]:
        st.wait_stream(lane_streams[0])         # frames made resident through ctx before this call
    if n_pairs < 0 or not callable(pair_fn) or not callable(flow_fn):
        raise ValueError("run_batch(n_pairs >= 0, pair_fn(p), flow_fn(im0, im1))")
    parts = partition(n_pairs, job.world)
    a, b = parts[job.rank]
    n_local, longest = b - a, max(q - p for p, q in parts)
    chunk = max(1, min(int(chunk), max(longest, 1)))
    n_chunks = (longest + chunk - 1) // chunk
    do_gather = gather and job.on
    on_gpu = ctx is not None
    if do_gather and shape is None and min(q - p for p, q in parts) == 0:
        raise ValueError("run_batch: %d pairs over %d ranks leaves a rank without pairs; pass shape=(H, W)" % (n_pairs, job.world))
    t0 = time.perf_counter()

    state = {"send": None, "result": None, "local": None, "shape": None, "comm": None}
    pending = []            # gathers in flight, oldest first: (work handle, send buffer index)

    # A gloo group exchanges host tensors: fields computed on the GPU are then downloaded into host send buffers (gloo's own
    # staging of GPU tensors hung in its second gather when two ranks shared a GPU).  RCCL takes the GPU tensors as they are.
    host_exchange = on_gpu and do_gather and job.dist.get_backend(group) == "gloo"

    def setup(hw):
        """Buffers, once the field shape is known (every pair of a batch has the same shape)."""
        job.device = torch.device("cuda", ctx.device) if (on_gpu and not host_exchange) else torch.device("cpu")
        H, W = int(hw[0]), int(hw[1])
        state["shape"] = (H, W)
        if do_gather:
            state["send"] = [torch.empty((chunk, 2, H, W), dtype=torch.float32, device=job.device) for _ in range(2)]
            if job.rank == dst:     # rank r's chunk c lands in result[r, c*chunk:(c+1)*chunk]: a contiguous view each
                state["result"] = torch.empty((job.world, n_chunks * chunk, 2, H, W), dtype=torch.float32, device=job.device)
            if on_gpu and not host_exchange:
                if lane_streams[0] is None:
                    raise ValueError("run_batch: gathering GPU tensors needs a context made by torch_context() "
                                     "(its stream orders the send buffer's copies before the collective)")
                state["comm"] = torch.cuda.Stream(device=job.device)
        else:
            state["local"] = torch.empty((n_local, 2, H, W), dtype=torch.float32, device=job.device)

    def store(dst_t, U, V, lctx):
        """Copy one flow field into dst_t (2, H, W) in the order of the lane's stream."""
        if on_gpu and isinstance(U, _ofri.DeviceArray) and dst_t.is_cuda:
            _ofri.check(lib.ofri_copy(lctx.handle, dst_t[0].data_ptr(), U.ptr, U.nbytes))
            _ofri.check(lib.ofri_copy(lctx.handle, dst_t[1].data_ptr(), V.ptr, V.nbytes))
        elif isinstance(U, _ofri.DeviceArray):                   # host exchange: download (waits for the lane's stream)
            dst_t[0].copy_(torch.from_numpy(U.numpy()))
            dst_t[1].copy_(torch.from_numpy(V.numpy()))
        else:
            dst_t[0].copy_(torch.from_numpy(np.ascontiguousarray(U, dtype=np.float32)))
            dst_t[1].copy_(torch.from_numpy(np.ascontiguousarray(V, dtype=np.float32)))

    def post(c):
        """Start the gather of chunk c (every rank calls this for every c, whether or not it has pairs in it)."""
        buf = state["send"][c & 1]
        recv = [state["result"][r, c * chunk:(c + 1) * chunk] for r in range(job.world)] if job.rank == dst else None
        comm = state["comm"]
        if comm is not None:
            for st in lane_streams:                              # the chunk's fields are complete on every lane
                comm.wait_stream(st)
            with torch.cuda.stream(comm):
                w = job.dist.gather(buf, recv, dst=dst, group=job.group, async_op=True)
        else:
            w = job.dist.gather(buf, recv, dst=dst, group=job.group, async_op=True)
        pending.append(w)

    def reuse(c):
        """Before chunk c's send buffer is written again, the gather that last read it (chunk c - 2) must be over."""
        while len(pending) >= 2:
            w = pending.pop(0)
            if state["comm"] is not None:
                for st in lane_streams:                          # every lane's stream is ordered after the gather
                    with torch.cuda.stream(st):
                        w.wait()
            else:
                w.wait()

    if shape is not None:
        setup(shape)
    k = 0                   # pairs of this rank done so far
    for c in range(n_chunks):
        lo, hi = a + c * chunk, min(a + (c + 1) * chunk, b)
        if do_gather and state["send"] is not None:
            reuse(c)
        for p in range(lo, hi):
            pair = pair_fn(p)
            lctx, lflow = L[k % len(L)]
            if ctx is not None:
                pair = _device_frames(lctx, pair)
            U, V = lflow(pair[0], pair[1])
            if state["shape"] is None:
                setup(np.shape(U) if not isinstance(U, _ofri.DeviceArray) else U.shape)
            if do_gather:
                store(state["send"][c & 1][p - lo], U, V, lctx)
            else:
                store(state["local"][k], U, V, lctx)
            k += 1
        if do_gather:
            post(c)
    while pending:
        w = pending.pop(0)
        if state["comm"] is not None:
            with torch.cuda.stream(lane_streams[0]):
                w.wait()
        else:
            w.wait()
    if on_gpu:
        if state["comm"] is not None:
            lane_streams[0].wait_stream(state["comm"])
        for lctx, _ in L:
            lctx.sync()
    if stats is not None:
        stats["elapsed_s"] = time.perf_counter() - t0
        stats["pairs"] = n_local
    if not do_gather:
        if state["local"] is None:
            return torch.empty((0, 2, 0, 0), dtype=torch.float32)
        return state["local"]
    if job.rank != dst:
        return None
    res = state["result"]
    H, W = state["shape"]
    if all(q - p == n_chunks * chunk for p, q in parts):
        return res.view(n_pairs, 2, H, W)            # equal blocks that fill their chunks: already in pair order
    return torch.cat([res[r, :q - p] for r, (p, q) in enumerate(parts)], 0)

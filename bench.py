#!/usr/bin/env python3
"""Benchmark: Mpix/s per PIV frame pair through the 2-level Horn-Schunck pyramid at 4096x4096.

One step = one frame pair through the whole hot path with the reference's example parameters
(Gaussian pre-filter Fs 3.4, 2 pyramid levels, alphas [21, 45], 100 Jacobi sweeps per level), inputs
already resident in HBM as float32.  N GPUs = N ranks, each with its own pairs (weak scaling); with
N > 1 every pair's flow field is gathered to rank 0 over RCCL, overlapped with the next pair.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "opticalflow-ri_amd")]

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak
HS_BYTES_PER_PX_SWEEP = 28.0   # read U,V,fx,fy,ft + write U,V, float32


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=4096, help="frame edge in pixels (benchmark config: 4096)")
    ap.add_argument("--niter", type=int, default=100)
    ap.add_argument("--fuse", type=int, default=0, help="Jacobi sweeps per launch (0 = automatic)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--same-device", action="store_true", help="all ranks compute on GPU 0 (rehearsal on a one-GPU box)")
    ap.add_argument("--liushen", action="store_true", help="add the Liu-Shen refinement (h = 5, Fs_opt 0.48) after Horn-Schunck on "
                    "every level: with --size 2048 the per-pair work of the 512-pair batch configuration (not the headline metric)")
    ap.add_argument("--force-dist", action="store_true", help="take the multi-rank code path (process group, gather stream, "
                    "barriers) even with one rank: rehearsal of the N > 1 path on a one-GPU box")
    ap.add_argument("--with-transfer", action="store_true", help="also report the rate including H2D/D2H copies")
    ap.add_argument("--graph", action="store_true", help="replay the pair's launches from a captured HIP graph (piv_plan.FlowPlan)")
    ap.add_argument("--in-flight", type=int, default=1, help="with --graph: pairs per step, each on its own context/stream "
                    "(small frames do not fill the GPU one pair at a time)")
    return ap.parse_args()


def cpu_baseline(niter, n=2048, pairs=2):
    """The CPU oracle (NumPy/SciPy restatement of the reference's Numba/SciPy path, single thread) on a
    bounded sample of the same workload: the identical pipeline on two 2048x2048 synthetic pairs
    (each a quarter of the benchmark frame; the per-pixel cost of the CPU path does not depend on the size)."""
    from piv_synth import synth_piv
    from oracle import hs as ohs, pyramid as opyr
    frames = [tuple(f.astype(np.float32) for f in synth_piv(n, n, seed=k)) for k in range(pairs)]
    t = time.perf_counter()
    for a, b in frames:
        opyr.pyramidal_flow(a, b, 3.4, ohs.HSAdapter([21, 45], niter), 2, 1, None)
    dt = time.perf_counter() - t
    return {"value": pairs * n * n / dt / 1e6, "unit": "Mpix/s", "cores": 1, "kind": "port",
            "sample": "same 2-level HS pipeline (Fs3.4, alphas [21,45], %d sweeps/level) on %d %dx%d synthetic pairs, "
                      "%.1f s, NumPy/SciPy single thread" % (niter, pairs, n, n, dt)}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dist = torch = None
    multi = world > 1 or args.force_dist
    if multi:
        import torch
        import torch.distributed as dist
        if args.backend == "nccl":
            torch.cuda.set_device(0 if args.same_device else local)
        dist.init_process_group(args.backend)

    import _ofri
    from piv_synth import synth_piv
    from HornSchunck import HSOpticalFlowAlgoAdapter
    from GenericPyramidalOpticalFlow import pyramidal_flow_device

    H = W = args.size
    dev = 0 if args.same_device else local        # --same-device: rehearse several ranks on a one-GPU box
    stream = None
    if multi and args.backend == "nccl":
        stream = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(stream)
        ctx = _ofri.Context(dev, stream.cuda_stream)
    else:
        ctx = _ofri.Context(dev)
    ctx.set_hs_fuse(args.fuse)

    # synthetic pair of this rank, resident in HBM as float32 before the timed region
    f0, f1 = synth_piv(H, W, seed=rank)
    lib = _ofri.lib()
    u0, u1 = ctx.upload(f0, np.uint8), ctx.upload(f1, np.uint8)
    im0, im1 = ctx.empty((H, W)), ctx.empty((H, W))
    _ofri.check(lib.ofri_u8_to_f32(ctx.handle, u0.ptr, im0.ptr, H * W))
    _ofri.check(lib.ofri_u8_to_f32(ctx.handle, u1.ptr, im1.ptr, H * W))
    ctx.sync()

    # Every pair's flow field goes to rank 0 (the only exchange of the job).  The gather of pair i runs
    # on a side stream while pair i+1 is computed; two send buffers alternate.
    gather = multi and not args.no_gather
    gather_note = "none"
    pending = []
    if gather:
        try:
            tdev = "cuda" if args.backend == "nccl" else "cpu"
            out_t = [torch.empty((2, H, W), dtype=torch.float32, device=tdev) for _ in range(2)]
            recv = [torch.empty((2, H, W), dtype=torch.float32, device=tdev) for _ in range(world)] if rank == 0 else None
            comm_stream = torch.cuda.Stream(device=dev) if tdev == "cuda" else None
            gather_note = "%s gather to rank 0 per pair, overlapped with the next pair" % ("rccl" if tdev == "cuda" else args.backend)
        except Exception as e:                       # keep the compute measurement alive
            gather, gather_note = False, "disabled: %r" % (e,)

    opt_args = {}
    if args.liushen:
        from PhysicsBasedOpticalFlowLiuShen import LiuShenOpticalFlowAlgoAdapter
        opt_args = {"FILTER_OPT": 0.48, "optionalOFlowAlgoAdapter": LiuShenOpticalFlowAlgoAdapter(5)}
    plan = None
    extra = []                      # further (context, plan) pairs when several frame pairs are in flight
    if args.graph:
        from piv_plan import FlowPlan
        plan = FlowPlan((H, W), 3.4, lambda: HSOpticalFlowAlgoAdapter([21, 45], args.niter), pyramidalLevels=2, ctx=ctx, **opt_args)
        plan.load(im0, im1)
        for k in range(1, max(1, args.in_flight)):
            c = _ofri.Context(dev)
            pk = FlowPlan((H, W), 3.4, lambda: HSOpticalFlowAlgoAdapter([21, 45], args.niter), pyramidalLevels=2, ctx=c, **opt_args)
            pk.load(*synth_piv(H, W, seed=100 * rank + k))
            c.sync()
            extra.append((c, pk))
    pairs_per_step = 1 + len(extra)

    def step(i):
        if plan is not None:
            for _, pk in extra:
                pk.run()
            Ua, Va = plan.run()
        else:
            Ua, Va = pyramidal_flow_device(im0, im1, 3.4, HSOpticalFlowAlgoAdapter([21, 45], args.niter), 2, 1,
                                           opt_args.get("FILTER_OPT"), opt_args.get("optionalOFlowAlgoAdapter"))
        if gather:
            buf = out_t[i & 1]
            if len(pending) >= 2:
                pending.pop(0).wait()           # the buffer we are about to overwrite has been sent
            if comm_stream is not None:
                _ofri.check(lib.ofri_copy(ctx.handle, buf[0].data_ptr(), Ua.ptr, Ua.nbytes))
                _ofri.check(lib.ofri_copy(ctx.handle, buf[1].data_ptr(), Va.ptr, Va.nbytes))
                comm_stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(comm_stream):
                    pending.append(dist.gather(buf, recv, dst=0, async_op=True))
            else:
                buf[0].copy_(torch.from_numpy(Ua.numpy())); buf[1].copy_(torch.from_numpy(Va.numpy()))
                pending.append(dist.gather(buf, recv, dst=0, async_op=True))
        return Ua, Va

    def fence():
        while pending:
            pending.pop(0).wait()
        if multi and args.backend == "nccl":
            if gather:
                torch.cuda.current_stream().wait_stream(comm_stream)
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
        elif multi:
            ctx.sync()
            dist.barrier()
        else:
            ctx.sync()
        for c, _ in extra:
            c.sync()

    for i in range(args.warmup):
        step(i)
    fence()
    ctx.prof_enable(True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    dt = time.perf_counter() - t0
    recs = ctx.prof_records()
    ctx.prof_enable(False)

    if multi:
        from piv_batch import max_over_ranks
        dt = max_over_ranks(dt, device="cuda" if args.backend == "nccl" else None)

    result = None
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = world * pairs_per_step * H * W * args.steps / dt / 1e6
        # dominant kernel: the Jacobi sweep launches of the full-resolution level.  A Horn-Schunck call may use
        # two kernels there (row-streaming pipeline for the multiples of 12 sweeps, tile kernel for the rest);
        # the one with the larger share of the time is reported.
        full = [r for r in recs if r[3] == H and r[4] == W]
        groups = {}
        for r in full:
            # one sweep per launch; tile kernel: at most 8 per launch; row-streaming pipeline: 12 or 16
            # (a last pipeline launch of 13 to 16 sweeps is another kernel instantiation and is timed on its own)
            name = "k_hs_sweep" if r[2] == r[1] else ("k_hs_stream_deep" if r[2] > 12 * r[1] else
                                                      ("k_hs_stream" if r[2] > 8 * r[1] else "k_hs_fused"))
            groups.setdefault(name, []).append(r)
        kname, grp = max(groups.items(), key=lambda kv: sum(r[0] for r in kv[1])) if groups else ("none", [])
        hs_ms_all = sum(r[0] for r in full)
        ms = sum(r[0] for r in grp)
        launches = sum(r[1] for r in grp)
        sweeps = sum(r[2] for r in grp)
        per_launch_ms = ms / max(launches, 1)
        sweeps_per_launch = sweeps / max(launches, 1)
        alg_bytes = HS_BYTES_PER_PX_SWEEP * H * W * sweeps_per_launch
        achieved = alg_bytes / (per_launch_ms * 1e-3) / 1e9 if per_launch_ms > 0 else 0.0
        traffic = None
        try:        # HBM bytes per launch from the committed rocprofv3 PMC passes of this same command
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                traffic = json.load(f).get("%s@%dx%d" % (kname, H, W))
        except (OSError, ValueError):
            pass
        result = {
            "metric": "Mpix/s per PIV frame pair (HS 100-iter @4096^2)",
            "value": value, "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "HS Fs3.4, 2-level pyramid, %dx%d synthetic Poiseuille pair, alphas [21,45], "
                                   "%d sweeps/level%s" % (H, W, args.niter, " + Liu-Shen h=5 Fs_opt 0.48 on every level" if args.liushen else ""),
                       "pairs_per_step_per_gpu": pairs_per_step, "flow_gather": gather_note, "captured_graph": bool(args.graph),
                       "hs_sweeps_per_launch": sweeps_per_launch},
            "roofline": {"bound": "hbm", "kernel": "%s@%dx%d" % (kname, H, W), "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "launch_ms": per_launch_ms, "launches_timed": launches,
                         "sweeps_per_launch": sweeps_per_launch,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "kernel_share_of_step": (ms / args.steps) / ms_per_step if ms_per_step > 0 else None,
                         "hs_level2_share_of_step": (hs_ms_all / args.steps) / ms_per_step if ms_per_step > 0 else None},
        }
    if args.with_transfer and rank == 0 and world == 1:
        t = time.perf_counter()
        for i in range(3):
            a_d, b_d = ctx.upload(f0, np.uint8), ctx.upload(f1, np.uint8)
            _ofri.check(lib.ofri_u8_to_f32(ctx.handle, a_d.ptr, im0.ptr, H * W))
            _ofri.check(lib.ofri_u8_to_f32(ctx.handle, b_d.ptr, im1.ptr, H * W))
            Ua, Va = step(i)
            Ua.numpy(); Va.numpy()
        result["with_transfer_mpix_s"] = H * W * 3 / (time.perf_counter() - t) / 1e6
        # the same with the three legs overlapped (piv_ingest.stream_flows): upload of the next pair and download of
        # the last result on copy streams of their own
        from piv_ingest import stream_flows
        n_str = 16
        g0, g1 = np.ascontiguousarray(f0, dtype=np.uint8), np.ascontiguousarray(f1, dtype=np.uint8)
        flow = lambda a, b: pyramidal_flow_device(a, b, 3.4, HSOpticalFlowAlgoAdapter([21, 45], args.niter), 2, 1, None)
        for _ in stream_flows(ctx, ((g0, g1) for _ in range(2)), flow):
            pass
        t = time.perf_counter()
        for _ in stream_flows(ctx, ((g0, g1) for _ in range(n_str)), flow):
            pass
        result["with_transfer_streamed_mpix_s"] = H * W * n_str / (time.perf_counter() - t) / 1e6
        # and with the pair's launches replayed from captured graphs (piv_plan.StreamedPlan)
        from piv_plan import StreamedPlan
        sp = StreamedPlan((H, W), 3.4, lambda: HSOpticalFlowAlgoAdapter([21, 45], args.niter), ctx=ctx, pyramidalLevels=2)
        for _ in sp.flows((g0, g1) for _ in range(3)):
            pass
        t = time.perf_counter()
        for _ in sp.flows((g0, g1) for _ in range(n_str)):
            pass
        result["with_transfer_streamed_graph_mpix_s"] = H * W * n_str / (time.perf_counter() - t) / 1e6
        sp.close()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args.niter)
    elif rank == 0:
        result["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(result))
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

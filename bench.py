#!/usr/bin/env python3
"""Benchmark: Mpix/s per PIV frame pair through the 2-level Horn-Schunck pyramid at 4096x4096.

One step = one frame pair through the whole hot path with the reference's example parameters
(Gaussian pre-filter Fs 3.4, 2 pyramid levels, alphas [21, 45], 100 Jacobi sweeps per level), inputs
already resident in HBM as float32.  The pair's launches are captured once into a HIP graph (piv_plan.FlowPlan, the
library's path for streams of equally sized pairs) and a step replays it; the Horn-Schunck sweep launches are timed
sections of that graph (event nodes), so the roofline block is measured inside the timed steps.  ``--direct`` launches
kernel by kernel through the driver instead.  N GPUs = N ranks, each with its own pairs (weak scaling): the pairs are
independent, so the timed path has no exchange between ranks (barrier before and after, time = the slowest rank's);
``--gather`` adds the asynchronous gather of every pair's flow field to rank 0 over RCCL, overlapped with the next pair.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...          # no launcher: this process starts the N ranks itself and relays rank 0's line

Started with ``--gpus N`` (N > 1) and no WORLD_SIZE in the environment, the process never touches the GPU: it checks that
N devices are visible (exit code 2 with a message otherwise; ``--same-device`` puts all ranks on GPU 0 for rehearsals),
starts N fresh rank processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, relays rank 0's JSON line and exits
non-zero if any rank fails.

``--batch P`` measures the batch configuration instead: P independent 2048x2048 pairs (``--size``), Horn-Schunck
plus Liu-Shen per pair, sharded in contiguous blocks over the ranks, flow fields gathered to rank 0 per chunk
(piv_batch.run_batch); one step = the whole batch, reported with and without the gather.
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
HBM_COPY_GBS = 4700.0          # what a device-to-device copy of 4 GiB sustains on the box (read + write; tools/hbm_copy_rate.py);
                               # reported beside the figures, not the yardstick of "memory-bound" (a streaming kernel moves more)
HS_BYTES_PER_PX_SWEEP = 28.0   # one Jacobi sweep: read U,V and three coefficient planes + write U,V, float32
VALU_LANE_OPS_PER_S = 256 * 4 * 16 * 2.4e9      # 256 CUs x 4 SIMDs x one wavefront instruction per 4 cycles (the rate of fused
                                                # multiply-adds, tools/valu_operands.hip) x 2.4 GHz, one operation per lane
HS_OPS_PER_PX_SWEEP = 14       # vector operations per pixel and sweep of the fast kernels (ofri_hs_math.h)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=None, help="frame edge in pixels (headline: 4096; with --batch: 2048)")
    ap.add_argument("--niter", type=int, default=100)
    ap.add_argument("--spin-up", type=float, default=0.3, help="seconds of untimed steps before the warm-up steps: after an idle "
                    "spell (the frames are rendered on the host) the GPU takes about 25 ms of work to reach its clocks")
    ap.add_argument("--fuse", type=int, default=0, help="Jacobi sweeps per launch (0 = automatic)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-one-core-only", action="store_true", help="skip the all-cores leg of the CPU baseline")
    ap.add_argument("--no-gather", action="store_true", help="with --batch: leave out the gather of the flow fields to rank 0")
    ap.add_argument("--gather", action="store_true", help="headline mode with N > 1: also send every pair's flow field to rank 0 "
                    "(asynchronous RCCL gather overlapped with the next pair); off by default: the pairs are independent and "
                    "the timed path has no exchange step")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--same-device", action="store_true", help="all ranks compute on GPU 0 (rehearsal on a one-GPU box)")
    ap.add_argument("--liushen", action="store_true", help="add the Liu-Shen refinement (h = 5, Fs_opt 0.48) after Horn-Schunck on "
                    "every level: with --size 2048 the per-pair work of the 512-pair batch configuration (not the headline metric)")
    ap.add_argument("--force-dist", action="store_true", help="take the multi-rank code path (process group, gather stream, "
                    "barriers) even with one rank: rehearsal of the N > 1 path on a one-GPU box")
    ap.add_argument("--with-transfer", action="store_true", help="also report the rate including H2D/D2H copies")
    ap.add_argument("--graph", action="store_true", help="(default) replay the pair's launches from a captured HIP graph (piv_plan.FlowPlan)")
    ap.add_argument("--direct", action="store_true", help="launch kernel by kernel through the driver instead of replaying the captured "
                    "graph (the gaps between a pair's 24 launches then cost about 5 %% at 4096 x 4096)")
    ap.add_argument("--in-flight", type=int, default=0, help="with --graph: pairs per step, each on its own context/stream (default 1); "
                    "with --batch: pairs in flight per GPU (default 3) "
                    "(small frames do not fill the GPU one pair at a time)")
    ap.add_argument("--no-in-flight", action="store_true", help="skip the three-pairs-in-flight throughput figure")
    ap.add_argument("--batch", type=int, default=0, help="total number of pairs of the batch configuration (512 on 8 GPUs); 0 = headline")
    ap.add_argument("--chunk", type=int, default=8, help="with --batch: pairs per gather")
    ap.add_argument("--launch-check", action="store_true", help="ranks only rendezvous (process group, one all-reduce) and rank 0 "
                    "prints what it saw: checks the launcher and the argument path without any GPU work")
    ap.add_argument("--distinct", type=int, default=4, help="with --batch: distinct synthetic pairs generated per rank (cycled; "
                    "rendering one 2048x2048 pair takes 0.5 s of host time)")
    a = ap.parse_args()
    a.graph = not a.direct
    if a.size is None:
        a.size = 2048 if a.batch else 4096
    return a


# ------------------------------------------------------------------------------------------ CPU baseline
def _cpu_pair(args):
    """One pair through the CPU oracle; returns seconds (also the worker of the all-cores leg)."""
    n, niter, seed = args
    from piv_synth import synth_piv
    from oracle import hs as ohs, pyramid as opyr
    a, b = (f.astype(np.float32) for f in synth_piv(n, n, seed=seed))
    t = time.perf_counter()
    opyr.pyramidal_flow(a, b, 3.4, ohs.HSAdapter([21, 45], niter), 2, 1, None)
    return time.perf_counter() - t


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(niter, all_cores=True, n=2048):
    """The CPU oracle (NumPy/SciPy restatement of the reference's Numba/SciPy path; the timed loop is the reference's
    HS_helper, src/HornSchunck.py:62-71, plus the pyramid glue around it) on a bounded sample of the same workload:
    the identical pipeline on 2048x2048 synthetic pairs (a quarter of the benchmark frame; the per-pixel cost of
    the CPU path does not depend on the size - checked by one run at 4096x4096), single thread: median of three.
    All-cores leg: one such pair per worker process on min(16, nproc) cores at once (the reference itself is
    single-threaded; this is what a user gets by running independent pairs side by side)."""
    times = sorted(_cpu_pair((n, niter, k)) for k in range(3))
    med = times[1]
    t_full = _cpu_pair((4096, niter, 0))
    out = {"value": 4096 * 4096 / t_full / 1e6, "unit": "Mpix/s", "cores": 1, "kind": "port",
           "cpu_model": _cpu_model(), "nproc": os.cpu_count(),
           "runs_s": [round(t, 3) for t in times], "value_%d_median_of_3" % n: n * n / med / 1e6,
           "sample": "same 2-level HS pipeline (Fs3.4, alphas [21,45], %d sweeps/level), NumPy/SciPy single thread: one pair at the "
                     "metric's own size, 4096x4096 (%.1f s; `value`), beside the median of 3 runs on %dx%d synthetic pairs "
                     "(%.1f s each)" % (niter, t_full, n, n, med)}
    if all_cores:
        try:
            import multiprocessing as mp
            cores = max(1, min(16, os.cpu_count() or 1))
            with mp.get_context("spawn").Pool(cores) as pool:
                pool.map(_cpu_pair, [(256, 2, 0)] * cores)               # start the workers, import NumPy/SciPy
                t = time.perf_counter()
                pool.map(_cpu_pair, [(n, niter, 100 + k) for k in range(cores)], chunksize=1)
                dt = time.perf_counter() - t
            out["all_cores"] = {"value": cores * n * n / dt / 1e6, "unit": "Mpix/s", "cores": cores,
                                "sample": "%d worker processes, one %dx%d pair each, %.1f s" % (cores, n, n, dt)}
        except Exception as e:                                           # the single-thread figure stands on its own
            out["all_cores"] = {"error": repr(e)}
    return out


# ------------------------------------------------------------------------------------------ roofline block
def _committed(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def one_sweep_roofline(ctx, H, W, launches=20):
    """The plain one-sweep-per-launch kernel (k_hs_sweep_vec4) on the same field: the HBM-bound form of the update,
    28 B per pixel and launch, timed with the library's events on the context's stream."""
    import _ofri
    lib = _ofri.lib()
    planes = [ctx.zeros((H, W)) for _ in range(5)]
    ctx.set_hs_fuse(1)
    try:
        args = [p.ptr for p in planes] + [21.0]
        _ofri.check(lib.ofri_hs_iterate(ctx.handle, *args, 4, H, W))
        ctx.sync()
        ctx.prof_enable(True)
        _ofri.check(lib.ofri_hs_iterate(ctx.handle, *args, launches, H, W))
        recs = ctx.prof_records()
        ctx.prof_enable(False)
    finally:
        ctx.set_hs_fuse(0)
    ms = sum(r[0] for r in recs) / max(1, sum(r[1] for r in recs))
    gbs = HS_BYTES_PER_PX_SWEEP * H * W / (ms * 1e-3) / 1e9
    traffic = _committed("pmc_traffic.json").get("k_hs_sweep@%dx%d" % (H, W))
    return {"kernel": "k_hs_sweep_vec4@%dx%d" % (H, W), "bound": "hbm", "launch_ms": ms, "achieved": gbs,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "launches_timed": launches,
            "traffic": traffic, "traffic_GBs": (traffic / (ms * 1e-3) / 1e9) if traffic else None}


def roofline_block(recs, H, W, steps, ms_per_step, streaming=None):
    """Dominant kernel = the Jacobi sweep launches of the full-resolution level.  A Horn-Schunck call may use several
    kernels there (row-streaming pipeline for the multiples of 12 sweeps, a deeper launch or the tile kernel for the
    rest); the one with the largest share of the time is reported."""
    full = [r for r in recs if r[3] == H and r[4] == W]
    groups = {}
    for r in full:
        # one sweep per launch; tile kernel: at most 8 per launch; row-streaming pipeline: 12 (a last launch of 13 to 16
        # sweeps is another kernel instantiation and is timed on its own)
        name = "k_hs_sweep" if r[2] == r[1] else ("k_hs_stream_deep" if r[2] > 12 * r[1] else
                                                  ("k_hs_stream" if r[2] > 8 * r[1] else "k_hs_fused"))
        groups.setdefault(name, []).append(r)
    if not groups:
        return None
    kname, grp = max(groups.items(), key=lambda kv: sum(r[0] for r in kv[1]))
    hs_ms_all = sum(r[0] for r in full)
    ms = sum(r[0] for r in grp)
    launches = sum(r[1] for r in grp)
    sweeps = sum(r[2] for r in grp)
    launch_ms = ms / max(launches, 1)
    spl = sweeps / max(launches, 1)
    key = "%s@%dx%d" % (kname, H, W)
    blocked = spl > 1.5
    # HBM roofline of the launch: the bytes one launch has to move whatever it does on chip (every plane read once,
    # the flow written once).  For a one-sweep kernel this equals the per-sweep figure.
    alg_launch = HS_BYTES_PER_PX_SWEEP * H * W
    achieved = alg_launch / (launch_ms * 1e-3) / 1e9
    traffic = _committed("pmc_traffic.json").get(key)              # HBM bytes per launch from the committed PMC passes
    sq = _committed("pmc_sq.json").get(key)                         # issue fractions from the committed SQ pass
    useful = HS_OPS_PER_PX_SWEEP * H * W * spl / VALU_LANE_OPS_PER_S * 1e3          # ms of vector issue the sweeps need
    hbm_meas = (traffic / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None
    # A blocked launch is called memory-bound only when its counted traffic moves at (85 % of) the rate the HBM-bound
    # one-sweep kernel sustains on the same field in this very run (`streaming`: counted bytes over its measured launch
    # time, about 6.2 TB/s); below that the limit is on the chip: instruction issue, whose measured fractions
    # (simd_any_issue, simd_valu_issue) stand beside the label.
    stream_gbs = (streaming or {}).get("traffic_GBs")
    mem_bound = hbm_meas is not None and stream_gbs is not None and hbm_meas * HBM_PEAK_GBS >= 0.85 * stream_gbs
    block = {
        "bound": ("hbm" if mem_bound else "issue") if blocked else "hbm",
        "bound_basis": None if not blocked else
        "counted traffic moves at %s GB/s against %s GB/s of the one-sweep kernel in the same run; instructions issued per SIMD and quad-cycle at 2.4 GHz, all kinds: %s (scalar, LDS and memory instructions can issue beside a vector one, so this can pass 1), vector: %s" % (
            ("%.0f" % (hbm_meas * HBM_PEAK_GBS)) if hbm_meas is not None else "?", ("%.0f" % stream_gbs) if stream_gbs else "?",
            (sq or {}).get("simd_any_issue", "?"), (sq or {}).get("simd_valu_issue", "?")),
        "kernel": key, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
        "traffic": traffic, "launch_ms": launch_ms, "launches_timed": launches, "sweeps_per_launch": spl,
        "algorithmic_bytes_per_launch": alg_launch,
        "hbm_frac_measured": hbm_meas,
        "hbm_frac_of_copy_rate": (hbm_meas * HBM_PEAK_GBS / HBM_COPY_GBS) if hbm_meas is not None else None,
        # the same launch against the per-sweep byte model (what separate sweeps would have to move): above the HBM
        # peak by construction when sweeps are blocked on chip; not a roofline fraction
        "effective_per_sweep_GBs": HS_BYTES_PER_PX_SWEEP * H * W * spl / (launch_ms * 1e-3) / 1e9,
        "valu_useful_issue_frac": useful / launch_ms if launch_ms > 0 else None,
        "valu_issue_measured": sq,
        "kernel_share_of_step": (ms / steps) / ms_per_step if ms_per_step > 0 else None,
        "hs_full_level_share_of_step": (hs_ms_all / steps) / ms_per_step if ms_per_step > 0 else None,
    }
    return block


# ------------------------------------------------------------------------------------------ batch configuration
def run_batch_config(args, rank, world, local, dist, torch):
    """The 512-pair configuration at any pair count: contiguous blocks per rank, Horn-Schunck + Liu-Shen per pair from a
    captured graph, frames resident in HBM as uint8, one asynchronous gather per chunk.  Timed twice: with the gather
    (the job as specified) and without it."""
    import piv_batch
    from piv_synth import synth_piv
    H = W = args.size
    dev = 0 if args.same_device else local
    n_lanes = args.in_flight if args.in_flight > 0 else 3
    lanes = piv_batch.flow_lanes(n_lanes, lambda c: piv_batch.hs_liushen_flow(c, (H, W), niter=args.niter, graph=True), dev)
    (ctx, flow), more = lanes[0], lanes[1:]
    a, b = piv_batch.partition(args.batch, world)[rank]
    distinct = max(1, min(args.distinct, max(b - a, 1)))
    resident = []
    for k in range(distinct):
        f0, f1 = synth_piv(H, W, seed=a + k)
        resident.append((ctx.upload(f0, np.uint8), ctx.upload(f1, np.uint8)))
    pair_fn = lambda p: resident[(p - a) % distinct]

    def sync_all():
        ctx.sync()
        if dist is not None:
            if args.backend == "nccl":
                torch.cuda.synchronize()
            dist.barrier()
            if args.backend == "nccl":
                torch.cuda.synchronize()

    def timed(gather, steps):
        sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = piv_batch.run_batch(args.batch, pair_fn, flow, ctx=ctx, chunk=args.chunk, gather=gather, shape=(H, W), lanes=more)
        sync_all()
        dt = time.perf_counter() - t0
        if dist is not None:
            dt = piv_batch.max_over_ranks(dt, device="cuda" if args.backend == "nccl" else None)
        return dt, out

    for _ in range(args.warmup):
        piv_batch.run_batch(min(args.batch, world * args.chunk), lambda p: resident[0], flow, ctx=ctx, chunk=args.chunk,
                            gather=dist is not None and not args.no_gather, shape=(H, W), lanes=more)
    dt_g, out = timed(not args.no_gather, args.steps)
    ok = None
    if rank == 0 and out is not None:
        ok = bool(tuple(out.shape) == (args.batch, 2, H, W)) if (dist is not None and not args.no_gather) else True
    del out
    dt_n, _ = timed(False, args.steps)
    if rank != 0:
        return None
    px = args.batch * H * W * args.steps
    return {
        "metric": "Mpix/s per PIV frame pair (batch of %d pairs, HS + Liu-Shen)" % args.batch,
        "value": px / dt_g / 1e6, "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt_g / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic (%d distinct pairs per rank, cycled)" % distinct,
        "config": {"workload": "batch of %d independent %dx%d synthetic pairs, HS Fs3.4 2-level alphas [21,45] %d sweeps/level + "
                               "Liu-Shen h=5 Fs_opt 0.48 on every level, contiguous blocks per rank" % (args.batch, H, W, args.niter),
                   "pairs_per_gpu": [q - p for p, q in piv_batch.partition(args.batch, world)],
                   "flow_gather": ("none" if (dist is None or args.no_gather) else
                                   "%s gather to rank 0 every %d pairs, overlapped with the next chunk" %
                                   ("rccl" if args.backend == "nccl" else args.backend, args.chunk)),
                   "captured_graph": True, "pairs_in_flight_per_gpu": n_lanes, "gathered_shape_ok": ok},
        "without_gather": {"value": px / dt_n / 1e6, "ms_per_step": dt_n / args.steps * 1e3},
        "ms_per_pair_per_gpu": dt_g / args.steps * 1e3 / max(1, -(-args.batch // world)),
    }


# ------------------------------------------------------------------------------------------ own launcher
def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def visible_gpus():
    """Number of GPUs this process could use, without initialising any (the parent of the ranks must stay off the GPU)."""
    import torch
    return int(torch.cuda.device_count())


def launch_ranks(args, argv):
    """``--gpus N`` without a launcher: start N rank processes (one per GPU), relay rank 0's standard output, return the
    exit code (the first failing rank's; the other ranks are then terminated by their process ids)."""
    import subprocess
    n = args.gpus
    if not args.same_device and not args.launch_check:
        have = visible_gpus()
        if have < n:
            sys.stderr.write("bench.py: --gpus %d but %d GPU(s) visible; nothing measured (use --same-device to rehearse "
                             "%d ranks on one GPU)\n" % (n, have, n))
            return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    live = set(range(n))
    while live and rc == 0:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0:
                rc = code if code > 0 else 128 - code
                sys.stderr.write("bench.py: rank %d exited with code %d\n" % (r, code))
                break
        time.sleep(0.05)
    if rc != 0:
        for r in live:
            procs[r].terminate()
        for r in live:
            try:
                procs[r].wait(timeout=20)
            except subprocess.TimeoutExpired:
                procs[r].kill()
        return rc
    reader.join(timeout=10)
    text = (out0[0] if out0 else b"").decode("utf-8", "replace")
    lines = [ln for ln in text.splitlines() if ln.strip().startswith("{")]
    if not lines:
        sys.stderr.write("bench.py: rank 0 printed no result line\n")
        return 3
    sys.stdout.write(lines[-1] + "\n")
    sys.stdout.flush()
    return 0


def emit(result, real_stdout):
    """The one JSON line, on the process's original standard output."""
    sys.stdout.flush()
    os.write(real_stdout, (json.dumps(result) + "\n").encode())


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args, sys.argv[1:]))
    if os.environ.get("OFRI_BENCH_WATCHDOG"):       # seconds after which every thread's stack is dumped and the run ends
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["OFRI_BENCH_WATCHDOG"]), exit=True)
    # Everything but the result line goes to standard error: libraries print banners to file descriptor 1 (RCCL its
    # version block when the first communicator is made), and the contract is one JSON line on standard output.
    real_stdout = os.dup(1)
    sys.stdout.flush()
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dist = torch = None
    multi = world > 1 or args.force_dist
    if multi:
        import torch
        import torch.distributed as dist
        if args.force_dist and "RANK" not in os.environ:        # one rank started by hand: be its own rendezvous
            for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", str(_free_port()))):
                os.environ.setdefault(k, v)
        if args.launch_check:
            if os.environ.get("OFRI_BENCH_FAIL_RANK") == str(rank):      # test hook: a rank that dies before the rendezvous
                raise SystemExit(5)
            dist.init_process_group("gloo")
            t = torch.tensor([float(rank)])
            dist.all_reduce(t)
            if rank == 0:
                emit({"launch_check": True, "n_gpus": world, "rank_sum": float(t.item()),
                      "same_device": bool(args.same_device)}, real_stdout)
            dist.barrier()
            dist.destroy_process_group()
            return
        if not args.same_device and torch.cuda.device_count() <= local:
            raise SystemExit("bench.py: rank %d needs GPU %d but %d GPU(s) are visible (use --same-device to rehearse on one)"
                             % (rank, local, torch.cuda.device_count()))
        if args.backend == "nccl":
            torch.cuda.set_device(0 if args.same_device else local)
            # the device is named: the communicator is bound to it at once instead of guessing it from the rank
            dist.init_process_group(args.backend, device_id=torch.device("cuda", 0 if args.same_device else local))
        else:
            dist.init_process_group(args.backend)

    if args.batch:
        if not multi:
            import torch
        result = run_batch_config(args, rank, world, local, dist, torch)
        if rank == 0:
            result["cpu_baseline"] = None
            emit(result, real_stdout)
        if multi:
            dist.barrier()
            dist.destroy_process_group()
        return

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
    gather = multi and args.gather
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
    graph_note = None
    # The sweep launches are timed sections of the graph: every replay times them (event nodes on the context's stream).
    # Ranks under torch.distributed replay an untimed graph instead and time the sweep launches in two kernel-by-kernel
    # steps after the timed region: torch's wheel brings its own HIP runtime, which refused the event nodes when tried.
    timed_graph = args.graph and not multi
    if args.graph:
        from piv_plan import FlowPlan
        try:
            plan = FlowPlan((H, W), 3.4, lambda: HSOpticalFlowAlgoAdapter([21, 45], args.niter), pyramidalLevels=2, ctx=ctx,
                            timed=timed_graph, **opt_args)
        except _ofri.OfriError as e:
            plan, graph_note = None, "kernel by kernel: graph capture failed (%s)" % (str(e)[:120],)
    if plan is not None:
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

    # The GPU has idled while the frames were rendered: the first 10 or so pairs after that run at lower clocks (2.1-2.5 ms
    # against 1.9-2.0 ms per pair).  A stream of pairs, which is what the metric is about, does not see that; so the same
    # computation runs untimed for a moment before the warm-up steps (no gather).
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < args.spin_up:
        for _ in range(4):
            if plan is not None:
                plan.run()
            else:
                pyramidal_flow_device(im0, im1, 3.4, HSOpticalFlowAlgoAdapter([21, 45], args.niter), 2, 1,
                                      opt_args.get("FILTER_OPT"), opt_args.get("optionalOFlowAlgoAdapter"))
        ctx.sync()
    for i in range(args.warmup):
        step(i)
    fence()
    if plan is None:
        ctx.prof_enable(True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    dt = time.perf_counter() - t0
    if plan is None:
        recs, rec_steps = ctx.prof_records(), args.steps
        ctx.prof_enable(False)
    elif timed_graph:
        recs, rec_steps = plan.prof_records(), 1            # the graph's sections as its last replay timed them
    else:
        ctx.prof_enable(True)
        rec_steps = 2
        for _ in range(rec_steps):
            pyramidal_flow_device(im0, im1, 3.4, HSOpticalFlowAlgoAdapter([21, 45], args.niter), 2, 1,
                                  opt_args.get("FILTER_OPT"), opt_args.get("optionalOFlowAlgoAdapter"))
        ctx.sync()
        recs = ctx.prof_records()
        ctx.prof_enable(False)

    if multi:
        from piv_batch import max_over_ranks
        dt = max_over_ranks(dt, device="cuda" if args.backend == "nccl" else None)

    result = None
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = world * pairs_per_step * H * W * args.steps / dt / 1e6
        one_sweep = one_sweep_roofline(ctx, H, W) if (world == 1 and args.fuse == 0) else None
        roof = roofline_block(recs, H, W, rec_steps, ms_per_step, one_sweep)
        if roof is not None and plan is not None:
            roof["timed_in"] = ("the last of the %d replays of the captured graph" % args.steps if timed_graph else
                                "two kernel-by-kernel steps after the timed replays")
        if roof is not None and one_sweep is not None:
            roof["one_sweep_kernel"] = one_sweep
        result = {
            "metric": "Mpix/s per PIV frame pair (HS 100-iter @4096^2)",
            "value": value, "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "HS Fs3.4, 2-level pyramid, %dx%d synthetic Poiseuille pair, alphas [21,45], "
                                   "%d sweeps/level%s" % (H, W, args.niter, " + Liu-Shen h=5 Fs_opt 0.48 on every level" if args.liushen else ""),
                       "pairs_per_step_per_gpu": pairs_per_step, "spin_up_s": args.spin_up, "flow_gather": gather_note, "captured_graph": plan is not None,
                       "launch_mode_note": graph_note if not multi else
                       ((graph_note + "; ") if graph_note else "") +
                       "ranks under torch.distributed run on the HIP runtime of torch's wheel (a step is about 3 % slower there "
                       "than on the system's runtime that a single-process run uses, so the N = 1 value of a multi-rank launch is "
                       "not the single-process value); the timed steps replay an untimed captured graph, and the roofline block "
                       "comes from two kernel-by-kernel steps after the timed region",
                       "hs_sweeps_per_launch": roof["sweeps_per_launch"] if roof else None},
            "roofline": roof,
        }
    if rank == 0 and world == 1 and pairs_per_step == 1 and not args.no_in_flight:
        # throughput figure beside the headline: three pairs in flight, each replayed from its own captured graph on its
        # own stream (one pair's launches leave gaps and tails that another pair's fill); not `value`, which stays one
        # pair per step with the roofline measured inside it
        from piv_plan import FlowPlan
        lanes = []
        for k in range(3):
            c = ctx if k == 0 else _ofri.Context(dev)
            c.set_share(2)              # the lanes' launches take half the row chunks each (what piv_batch.flow_lanes sets)
            pk = FlowPlan((H, W), 3.4, lambda: HSOpticalFlowAlgoAdapter([21, 45], args.niter), pyramidalLevels=2, ctx=c, **opt_args)
            c.set_share(1)              # the captured graph keeps its launch shapes
            pk.load(*((im0, im1) if k == 0 else synth_piv(H, W, seed=100 + k)))
            c.sync()
            lanes.append((c, pk))
        def rounds(n, use=None):
            for _ in range(n):
                for _, pk in (use or lanes):
                    pk.run()
            for c, _ in lanes:
                c.sync()
        rounds(2)
        if plan is None:
            # one pair at a time from its captured graph: the headline's work without the gaps between its 24 launches
            t = time.perf_counter()
            rounds(args.steps, lanes[:1])
            dt1 = time.perf_counter() - t
            result["one_pair_captured_graph"] = {"value": H * W * args.steps / dt1 / 1e6, "unit": "Mpix/s",
                                                 "ms_per_pair": dt1 / args.steps * 1e3}
        else:
            # the same pair launched kernel by kernel through the driver (what a caller without a plan gets)
            def direct(n):
                for _ in range(n):
                    pyramidal_flow_device(im0, im1, 3.4, HSOpticalFlowAlgoAdapter([21, 45], args.niter), 2, 1,
                                          opt_args.get("FILTER_OPT"), opt_args.get("optionalOFlowAlgoAdapter"))
                ctx.sync()
            direct(2)
            t = time.perf_counter()
            direct(args.steps)
            dtd = time.perf_counter() - t
            result["kernel_by_kernel"] = {"value": H * W * args.steps / dtd / 1e6, "unit": "Mpix/s",
                                          "ms_per_pair": dtd / args.steps * 1e3}
        t = time.perf_counter()
        rounds(args.steps)
        dt3 = time.perf_counter() - t
        result["three_pairs_in_flight"] = {"value": 3 * H * W * args.steps / dt3 / 1e6, "unit": "Mpix/s",
                                           "ms_per_pair": dt3 / args.steps / 3 * 1e3, "captured_graphs": True}
        for c, pk in lanes:
            pk.close()
            if c is not ctx:
                c.close()
    if args.with_transfer and rank == 0 and world == 1:
        t = time.perf_counter()
        for i in range(3):
            a_d, b_d = ctx.upload(f0, np.uint8), ctx.upload(f1, np.uint8)
            _ofri.check(lib.ofri_u8_to_f32(ctx.handle, a_d.ptr, im0.ptr, H * W))
            _ofri.check(lib.ofri_u8_to_f32(ctx.handle, b_d.ptr, im1.ptr, H * W))
            if plan is not None:
                plan.load(im0, im1)
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
    if rank == 0:
        # The CPU leg runs last: a minute of host work in front of the GPU measurement left the GPU idle at low clocks
        # when the timed steps began.  Its all-cores part starts fresh worker processes (spawn), which never touch the GPU.
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            try:
                cpu = cpu_baseline(args.niter, all_cores=not args.cpu_one_core_only)
            except Exception as e:                      # the GPU figures stand on their own
                cpu = {"error": repr(e)}
        result["cpu_baseline"] = cpu
        emit(result, real_stdout)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

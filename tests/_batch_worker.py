"""Child process of tests/test_gpu_batch.py: the batch runner on one GPU over a real process group.

    python tests/_batch_worker.py <backend> <n_pairs> <chunk>

Runs piv_batch.run_batch (Horn-Schunck + Liu-Shen per pair, the 512-pair configuration's call at reduced size) with
world_size 1 on ``backend`` ('nccl' = RCCL) and compares every gathered field, bit for bit, with the driver called
pair by pair.  Exit code 0 and a last line 'BATCH-OK' on success."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "opticalflow-ri_amd")]


def no_group():
    """Single-rank call with no torch.distributed at all; frames resident in HBM as uint8 (the benchmark's layout)."""
    import torch  # noqa: F401  (before the first _ofri call: one HIP runtime for both)
    import _ofri
    import piv_batch
    from piv_synth import synth_piv
    from HornSchunck import HSOpticalFlowAlgoAdapter
    from GenericPyramidalOpticalFlow import pyramidal_flow_device
    ctx, stream = piv_batch.torch_context(0)
    H, W = 128, 192
    host = [synth_piv(H, W, seed=10 + p) for p in range(3)]
    resident = [(ctx.upload(a, np.uint8), ctx.upload(b, np.uint8)) for a, b in host]
    flow = piv_batch.hs_liushen_flow(ctx, (H, W), niter=16, liushen=False, graph=True)
    out = piv_batch.run_batch(3, lambda p: resident[p], flow, ctx=ctx, chunk=2)
    assert tuple(out.shape) == (3, 2, H, W) and out.is_cuda
    got = out.cpu().numpy()
    for p, (a, b) in enumerate(host):
        U, V = pyramidal_flow_device(ctx.upload(a), ctx.upload(b), 3.4, HSOpticalFlowAlgoAdapter([21, 45], 16), 2, 1, None)
        np.testing.assert_array_equal(got[p, 0], U.numpy())
        np.testing.assert_array_equal(got[p, 1], V.numpy())
    # three pairs in flight: lane k takes the rank's pairs k, k + 3, ...; same fields in the same order
    lanes = piv_batch.flow_lanes(2, lambda c: piv_batch.hs_liushen_flow(c, (H, W), niter=16, liushen=False, graph=True), 0)
    torch.cuda.set_stream(torch.cuda.default_stream())      # the runner orders against the contexts' own streams
    out3 = piv_batch.run_batch(3, lambda p: resident[p], flow, ctx=ctx, chunk=2, lanes=lanes)
    np.testing.assert_array_equal(out3.cpu().numpy(), got)
    for _, f in lanes:
        f.close()
    flow.close()
    print("BATCH-OK")


def cfg5_share(port):
    """One GPU's share of the 512-pair configuration at its own size and shape: 64 pairs of 2048 x 2048, Horn-Schunck +
    Liu-Shen per pair from captured graphs on three lanes, frames resident as uint8, one RCCL gather per chunk of 8
    (process group of one rank).  Every ninth field (pairs 0, 9, ... 63: every lane, every slot of a chunk, both send
    buffers) is compared bit for bit with the driver called on that pair alone."""
    import time
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    import piv_batch
    from piv_synth import synth_piv
    from HornSchunck import HSOpticalFlowAlgoAdapter
    from PhysicsBasedOpticalFlowLiuShen import LiuShenOpticalFlowAlgoAdapter
    from GenericPyramidalOpticalFlow import pyramidal_flow_device
    H = W = 2048
    n_pairs, distinct = 64, 8
    lanes = piv_batch.flow_lanes(3, lambda c: piv_batch.hs_liushen_flow(c, (H, W), graph=True), 0)
    (ctx, flow), more = lanes[0], lanes[1:]
    host = [synth_piv(H, W, seed=k) for k in range(distinct)]
    print("frames rendered", flush=True)
    resident = [(ctx.upload(a, np.uint8), ctx.upload(b, np.uint8)) for a, b in host]
    t = time.perf_counter()
    out = piv_batch.run_batch(n_pairs, lambda p: resident[p % distinct], flow, ctx=ctx, chunk=8, shape=(H, W), lanes=more)
    print("batch of %d pairs: %.1f ms per pair" % (n_pairs, (time.perf_counter() - t) * 1e3 / n_pairs), flush=True)
    assert tuple(out.shape) == (n_pairs, 2, H, W) and out.is_cuda
    assert bool(torch.isfinite(out).all())
    for p in range(0, n_pairs, 9):
        a, b = host[p % distinct]
        U, V = pyramidal_flow_device(ctx.upload(a.astype(np.float32)), ctx.upload(b.astype(np.float32)), 3.4,
                                     HSOpticalFlowAlgoAdapter([21, 45], 100), 2, 1, 0.48, LiuShenOpticalFlowAlgoAdapter(5))
        got = out[p].cpu().numpy()
        np.testing.assert_array_equal(got[0], U.numpy(), err_msg="pair %d U" % p)
        np.testing.assert_array_equal(got[1], V.numpy(), err_msg="pair %d V" % p)
        assert -4.5 < float(got[0][H // 2].mean()) < -3.3            # the Poiseuille pair's centre row
    # pairs that share a frame pair share a field, whichever lane and chunk slot computed them
    assert bool(torch.equal(out[3], out[3 + distinct])) and bool(torch.equal(out[5], out[5 + 7 * distinct]))
    for _, f in lanes:
        f.close()
    dist.barrier()
    dist.destroy_process_group()
    print("BATCH-OK")


def main():
    if sys.argv[1] == "nogroup":
        return no_group()
    if sys.argv[1] == "cfg5":
        return cfg5_share(sys.argv[2] if len(sys.argv) > 2 else "29534")
    backend, n_pairs, chunk = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", sys.argv[4] if len(sys.argv) > 4 else "29533")
    if backend == "nccl":
        torch.cuda.set_device(0)
    # before anything else touches the GPU; the device is named, so the communicator is not left guessing it from the rank
    dist.init_process_group(backend, rank=0, world_size=1,
                            **({"device_id": torch.device("cuda", 0)} if backend == "nccl" else {}))
    import _ofri
    import piv_batch
    from piv_synth import synth_piv
    from HornSchunck import HSOpticalFlowAlgoAdapter
    from PhysicsBasedOpticalFlowLiuShen import LiuShenOpticalFlowAlgoAdapter
    from GenericPyramidalOpticalFlow import pyramidal_flow_device

    H, W, niter = 192, 256, 24
    ctx, stream = piv_batch.torch_context(0)
    frames = [synth_piv(H, W, seed=p) for p in range(n_pairs)]
    for graph in (False, True):
        flow = piv_batch.hs_liushen_flow(ctx, (H, W), niter=niter, graph=graph)
        stats = {}
        out = piv_batch.run_batch(n_pairs, lambda p: frames[p], flow, ctx=ctx, chunk=chunk, stats=stats)
        assert tuple(out.shape) == (n_pairs, 2, H, W) and out.is_cuda, (out.shape, out.device)
        got = out.cpu().numpy()
        for p in range(n_pairs):
            a, b = ctx.upload(frames[p][0]), ctx.upload(frames[p][1])
            U, V = pyramidal_flow_device(a, b, 3.4, HSOpticalFlowAlgoAdapter([21, 45], niter), 2, 1, 0.48,
                                         LiuShenOpticalFlowAlgoAdapter(5))
            np.testing.assert_array_equal(got[p, 0], U.numpy(), err_msg="pair %d U (graph=%s)" % (p, graph))
            np.testing.assert_array_equal(got[p, 1], V.numpy(), err_msg="pair %d V (graph=%s)" % (p, graph))
            assert np.isfinite(got[p]).all() and np.abs(got[p, 0]).max() > 0.5
        assert stats["pairs"] == n_pairs and stats["elapsed_s"] > 0
        # the same without the exchange: the rank's own block
        loc = piv_batch.run_batch(n_pairs, lambda p: frames[p], flow, ctx=ctx, chunk=chunk, gather=False)
        np.testing.assert_array_equal(loc.cpu().numpy(), got)
        # several pairs in flight (each lane its own stream and graph), gathered chunk by chunk as before
        lanes = piv_batch.flow_lanes(2, lambda c: piv_batch.hs_liushen_flow(c, (H, W), niter=niter, graph=graph), 0)
        # torch's current stream is none of the lanes' from here on: the runner must order the gather against the
        # streams the contexts recorded, not against the ambient one
        torch.cuda.set_stream(torch.cuda.default_stream())
        out3 = piv_batch.run_batch(n_pairs, lambda p: frames[p], flow, ctx=ctx, chunk=chunk, lanes=lanes)
        got3 = out3.cpu().numpy()
        bad = [p for p in range(n_pairs) if not (np.array_equal(got3[p, 0], got[p, 0]) and np.array_equal(got3[p, 1], got[p, 1]))]
        assert not bad, "lanes (graph=%s): pairs %r differ from the one-lane run" % (graph, bad)
        for _, f in lanes:
            f.close()
        flow.close()
    t = piv_batch.max_over_ranks(1.25, device="cuda" if backend == "nccl" else None)
    assert t == 1.25
    dist.barrier()
    dist.destroy_process_group()
    print("BATCH-OK")


if __name__ == "__main__":
    main()

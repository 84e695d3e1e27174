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
    torch.cuda.set_stream(stream)
    out3 = piv_batch.run_batch(3, lambda p: resident[p], flow, ctx=ctx, chunk=2, lanes=lanes)
    np.testing.assert_array_equal(out3.cpu().numpy(), got)
    for _, f in lanes:
        f.close()
    flow.close()
    print("BATCH-OK")


def main():
    if sys.argv[1] == "nogroup":
        return no_group()
    backend, n_pairs, chunk = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", sys.argv[4] if len(sys.argv) > 4 else "29533")
    if backend == "nccl":
        torch.cuda.set_device(0)
    dist.init_process_group(backend, rank=0, world_size=1)       # before anything else touches the GPU
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
        torch.cuda.set_stream(stream)
        out3 = piv_batch.run_batch(n_pairs, lambda p: frames[p], flow, ctx=ctx, chunk=chunk, lanes=lanes)
        np.testing.assert_array_equal(out3.cpu().numpy(), got, err_msg="lanes (graph=%s)" % graph)
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

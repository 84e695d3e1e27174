"""The batch runner (piv_batch.run_batch) on the GPU: sharding is the identity with one rank, but the whole path
runs - resident frames widened on the GPU, per-pair flow, per-chunk asynchronous gather over RCCL into the result
tensor - and every gathered field must equal the driver's, pair by pair, bit for bit."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("n_pairs,chunk", [(5, 2), (4, 4)])
def test_run_batch_over_rccl_world_size_1(n_pairs, chunk):
    """In a child process, so that the process group is initialised before anything else touches the GPU."""
    _child("nccl", n_pairs, chunk)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_one_gpus_share_of_the_512_pair_configuration():
    """64 pairs of 2048 x 2048 on three lanes with the RCCL gather per chunk of 8: the runner at the configuration's own
    size and shape (examples/LiuSE_PyHSchunck_Fs3_4_PyrLvls2.py:133-139 per pair)."""
    _child("cfg5", str(_free_port()), timeout=280)


def _child(*argv, timeout=110):
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, "-u", os.path.join(HERE, "_batch_worker.py")] + [str(a) for a in argv],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    text = r.stdout.decode("utf-8", "replace")
    assert r.returncode == 0 and "BATCH-OK" in text, text[-3000:]


def test_run_batch_without_a_process_group_and_resident_frames():
    """Single-rank call with no torch.distributed at all; frames resident in HBM as uint8 (the benchmark's layout).
    A child process as well: torch must be imported before libofri_hip.so is loaded (one HIP runtime for both), and
    this process has loaded the library long ago."""
    _child("nogroup")

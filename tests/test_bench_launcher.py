"""bench.py started without a launcher: ``--gpus N`` must start N ranks itself or refuse loudly, never measure one GPU
and call it N (no GPU work here: the ranks only rendezvous over gloo)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(*argv, env=None, timeout=120):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + list(argv), env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          timeout=timeout)


def test_gpus_n_starts_n_ranks_and_relays_rank_0():
    r = _run("--gpus", "2", "--launch-check")
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = r.stdout.decode().strip().splitlines()
    assert len(lines) == 1, lines                       # one JSON line on standard output, nothing else
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["rank_sum"] == 1.0
    r = _run("--gpus", "3", "--launch-check", "--same-device")
    rec = json.loads(r.stdout.decode().strip())
    assert r.returncode == 0 and rec["n_gpus"] == 3 and rec["rank_sum"] == 3.0 and rec["same_device"]


def test_gpus_n_without_n_devices_exits_non_zero():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("two GPUs are visible here")
    r = _run("--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode == 2 and r.stdout == b""
    assert b"--gpus 2 but" in r.stderr and b"nothing measured" in r.stderr


def test_a_failing_rank_fails_the_launch():
    r = _run("--gpus", "2", "--launch-check", env={"OFRI_BENCH_FAIL_RANK": "1"})
    assert r.returncode == 5 and r.stdout == b""
    assert b"rank 1 exited with code 5" in r.stderr


def test_world_size_and_gpus_must_agree():
    r = _run("--gpus", "1", "--launch-check", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and b"--gpus 1 but WORLD_SIZE=2" in r.stderr

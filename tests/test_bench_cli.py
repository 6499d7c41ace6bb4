"""CPU: bench.py's launcher logic (no GPU here, so the ranks themselves stop at "needs a ROCm GPU"): `--gpus N` started
plainly must START N rank processes (torchrun as a child, before any GPU call), and a world size that differs from
--gpus must be refused instead of silently measuring fewer GPUs."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=240):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


@pytest.mark.timeout(300)
def test_gpus_n_spawns_n_ranks_and_fails_loudly_without_gpus():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    assert r.returncode != 0
    assert "starting 2 ranks" in r.stderr and "--nproc-per-node=2" in r.stderr
    # the ranks ran and refused to fall back to the CPU (torchrun terminates the sibling as soon as the first rank fails, so
    # the second refusal may or may not make it to stderr)
    assert r.stderr.count("needs a ROCm GPU") >= 1
    assert '"metric"' not in r.stdout                            # no result line from a failed run


def test_world_size_mismatch_is_refused():
    r = _run(["--gpus", "8", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "refusing to measure a different number of GPUs" in (r.stderr + r.stdout)
    r = _run(["--gpus", "1", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "refusing to measure a different number of GPUs" in (r.stderr + r.stdout)

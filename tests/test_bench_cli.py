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


@pytest.mark.timeout(600)
def test_reducer_trials_run_as_child_processes_before_the_timed_process():
    """--reducer auto under a data-parallel launch (here: the one-rank self-test): every candidate is tried in a child
    process of its own BEFORE the timed process touches the GPU.  Without a GPU each child stops at "needs a ROCm GPU"; the
    parent must have started all of them (plain / native / inorder / python), survive their failure, and then fail loudly itself --
    never print a result line."""
    r = _run(["--steps", "1", "--warmup", "0", "--no-cpu-baseline"], {"DVGR_DDP_SELFTEST": "1", "MASTER_PORT": "29711"}, timeout=500)
    assert r.returncode != 0
    for kind in ("plain", "native", "inorder", "python"):
        assert f"trial child '{kind}'" in r.stderr, r.stderr[-2000:]
    assert '"metric"' not in r.stdout
    # an explicit --reducer starts no children
    r = _run(["--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--reducer", "python"], {"DVGR_DDP_SELFTEST": "1"}, timeout=300)
    assert r.returncode != 0 and "trial child" not in r.stderr

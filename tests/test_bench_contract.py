"""bench.py's entry-point contract, as far as it can be checked without a GPU: it refuses to run without a device (no CPU path), a
--gpus N it cannot honour is an error and never a silent 1-GPU run, and a torchrun world that disagrees with --gpus is an error."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e, timeout=300)


def test_bench_needs_a_gpu_and_prints_nothing_to_stdout_when_it_fails():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("this is the no-GPU contract")
    p = _run(["--steps", "1", "--warmup", "0"])
    assert p.returncode != 0 and "no CPU path" in p.stderr and p.stdout.strip() == ""


def test_bench_gpus_n_without_n_devices_is_an_error():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("needs a box with fewer than two devices")
    p = _run(["--gpus", "2"])
    assert p.returncode == 2 and "--gpus 2" in p.stderr and p.stdout.strip() == ""


def test_bench_world_size_must_match_gpus():
    p = _run(["--gpus", "1"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode == 2 and "WORLD_SIZE" in p.stderr

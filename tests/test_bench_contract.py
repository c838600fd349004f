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


def test_committed_traffic_is_tied_to_the_machine_code_of_the_library():
    """roofline.traffic comes from PMC passes run earlier (profiles/collect.sh); bench.py reports it only while the library's gfx950 machine code is the code
    those passes ran.  The tie is a hash of the device code itself (mapad_amd/build.py: kernel_code_hash) — it must exist for the built library, be the same
    for two reads of it, and every entry that feeds the default line must carry it."""
    import json
    from mapad_amd import build
    build.build()
    h = build.kernel_code_hash()
    assert h and len(h) == 16 and int(h, 16) >= 0 and h == build.kernel_code_hash()
    assert build.kernel_code_hash(os.path.join(ROOT, "no_such_library.so")) is None
    traffic = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    for key in ("c4:3000000000:10000000", "c2:48000000:1000000", "c3:48000000:1000000"):
        assert len(traffic[key].get("kernel_code_sha16", "")) == 16 and traffic[key]["search_kernel"] > 0

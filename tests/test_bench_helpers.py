"""bench.py's host-side helpers (no GPU): what the container grants, and that the bench refuses to run without a device
instead of measuring something else."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpu_quota_is_none_or_a_positive_number_of_cpus():
    sys.path.insert(0, ROOT)
    import bench
    q = bench.cpu_quota()
    assert q is None or (isinstance(q, float) and 0 < q <= 4096)


def test_without_a_device_the_bench_fails_loudly():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a device is present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--cpu-sample", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert not any(line.startswith("{") for line in r.stdout.splitlines())      # no JSON line: nothing was measured

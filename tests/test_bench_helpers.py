"""bench.py's host-side helpers (no GPU): what the container grants, and that the bench refuses to run without a device
instead of measuring something else."""
import json
import os
import subprocess
import sys
import time

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
    lines = [json.loads(line) for line in r.stdout.splitlines() if line.startswith("{")]
    assert all(l["value"] is None and l["error"] for l in lines)                # nothing was measured, and the line (if any) says so


# ---- `bench.py --gpus N` outside a launcher: the spawner must not hang on a rank that dies
_CHILD = """
import os, sys, time, signal
r = int(os.environ["RANK"]); n = int(os.environ["WORLD_SIZE"])
open(os.path.join(os.environ["MBGC_BENCH_RUNDIR"], "rank%d.up" % r), "w").close()
mode = sys.argv[1]
if mode == "ok":
    time.sleep(0.3)
    if r == 0: print('{"metric": "m", "value": 1.0, "n_gpus": %d}' % n, flush=True)
elif mode == "kill3":
    if r == 3:
        time.sleep(1.0); os.kill(os.getpid(), signal.SIGKILL)
    time.sleep(600)                       # the others sit "inside a collective"
elif mode == "stuck":
    time.sleep(600)
elif mode == "exit7":
    if r == 1: sys.exit(7)
    time.sleep(600)
"""


def _spawn(mode, n, limit=None, grace=1.0):
    code = ("import sys; sys.path.insert(0, %r); import bench; bench.spawn_ranks(%d, argv=[sys.executable, '-c', %r, %r], grace=%r, limit=%r)"
            % (ROOT, n, _CHILD, mode, grace, limit))
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    return r, time.monotonic() - t0


def test_spawner_relays_rank_0s_line():
    r, dt = _spawn("ok", 4)
    assert r.returncode == 0, r.stderr
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0]["value"] == 1.0 and lines[0]["n_gpus"] == 4


def test_spawner_ends_the_run_when_a_rank_is_killed():
    """8 ranks, rank 3 dies on SIGKILL a second in, the others would wait ten minutes: the spawner returns within seconds,
    non-zero, with a line that says value null, which rank, and how many ranks had come up"""
    r, dt = _spawn("kill3", 8)
    assert r.returncode != 0 and dt < 30, (r.returncode, dt)
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0]["value"] is None and "rank 3" in lines[0]["error"] and "signal 9" in lines[0]["error"]
    assert lines[0]["rccl_ranks_seen"] == 8 and lines[0]["n_gpus"] == 8


def test_spawner_ends_the_run_when_a_rank_exits_with_an_error():
    r, dt = _spawn("exit7", 2)
    assert r.returncode == 7 and dt < 30
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines[0]["value"] is None and "rank 1" in lines[0]["error"] and "exit code 7" in lines[0]["error"]


def test_spawner_bounds_the_whole_run():
    r, dt = _spawn("stuck", 2, limit=2.0)
    assert r.returncode != 0 and dt < 30
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines[0]["value"] is None and "limit" in lines[0]["error"]


# ---- the lines bench.py adds beside the headline are put together from other programs' output: their format strings and regular
# expressions are exercised here with stand-ins for those programs (a percent sign in one of them cost a round-4 run its configs[1] line)
def _fake_tool(tmp_path, stdout, stderr):
    p = tmp_path / "fake-tool"
    p.write_text("#!/bin/sh\ncat <<'EOF_OUT'\n%s\nEOF_OUT\ncat >&2 <<'EOF_ERR'\n%s\nEOF_ERR\n" % (stdout, stderr))
    p.chmod(0o755)
    return str(p)


def test_the_lines_beside_the_headline_are_put_together(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    prof = ('kernel profile: {"load": {"ms": 7.0, "launches": 500}, "insert": {"ms": 12.0, "launches": 500}, "probe": {"ms": 0.0, "launches": 0}, '
            '"emit2": {"ms": 230.0, "launches": 500}, "resolve": {"ms": 140.0, "launches": 500}, "stitch": {"ms": 90.0, "launches": 500}, "emit": {"ms": 13.0, "launches": 500}}')
    out = "rounds of 28 targets; reference extension bytes dropped at the sliding window's end: 0\nfinal reference length: 1809744201\nexact matches total: 6954058\nswsMEM unmatched chars: 381336073\nfinal unmatched chars: 1"
    monkeypatch.setattr(bench, "TOOL", _fake_tool(tmp_path, out, "matching finished - 432 [ms]\n" + prof))
    (tmp_path / "meta.json").write_text(json.dumps({"genomes": 200, "bases": 909_000_000, "g0_bases": 5_000_000}))
    (tmp_path / "list.txt").write_text("")
    for rounds in (False, True):
        line = bench.config4_line(str(tmp_path), rounds=rounds)
        assert "error" not in line, line
        assert line["value"] == round(904_000_000 / 0.432 / 1e9, 4) and line["roofline"]["kernel"] == "k_resolve_blocks4"
        assert ("rounds of 28" in line["workload"]) == rounds and ("0.2-10 % divergence" in line["workload"]) != rounds
    # configs[1]: a child bench.py whose last line is the diagnostic line of `--no-emit`
    child = {"metric": "m", "value": 56.4, "unit": "Gbases/s", "ms_per_step": 1.4, "extension_bytes_dropped_per_step": 0.0,
             "config": {"max_ref_len": 1_280_000_000, "targets_per_step": 16}, "roofline": {"frac": 0.036}, "kernel_ms_per_launch": {"resolve": 0.92}}

    class R:
        returncode = 0
        stdout = json.dumps(child) + "\n"
        stderr = ""
    monkeypatch.setattr(bench.subprocess, "run", lambda *a, **k: R())
    line = bench.config1_line()
    assert "error" not in line, line
    assert line["value"] == 56.4 and "99% identity" in line["workload"] and "1.28e+09-byte" in line["workload"]

"""`mbgc-hip c --gpus N`: the C++ host with a round's targets sharded over N ranks (mbgc_amd/host/mgmp_sharded.cpp) and the
exchange of include/mbgc_exchange.h between them. The test box has one GPU, so the ranks share it and the bytes move
through host shared memory (`--exchange hostmem`: same calls, same order, same bytes as RCCL, which refuses two ranks on
one device); RCCL itself runs here with one rank. N ranks x R targets must equal one GPU with rounds of N x R — which
test_gpu_cli.py pins on the oracle-driven reference loop — byte for byte, in every stream."""
import json
import os
import subprocess

import pytest

from mbgc_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "mbgc_amd", "mbgc-hip")
STREAMS = ("literals", "mapOff", "mapOff5th", "mapLen", "gapDelta", "flags", "locksPos", "refExtSize")


def run_tool(args, cwd, env=None):
    r = subprocess.run(["timeout", "-k", "10", "280", TOOL] + args, cwd=cwd, capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    return r.stdout


def split(g, k):
    cuts = [0] + [g.size * i // k + (7 * i) % 13 for i in range(1, k)] + [g.size]
    return [g[cuts[i]:cuts[i + 1]] for i in range(k)]


def write_collection(tmp_path, n, length, div, contigs=2, seed=61):
    base = synth.base_codes(length, seed)
    paths = []
    for i in range(n):
        p = tmp_path / ("g%02d.fa" % i)
        with open(p, "wb") as f:
            d = (0.07 if i % 3 == 2 else 0.004) if div == "mixed" else div      # "mixed": kept targets between and behind stopped ones
            for j, c in enumerate(split(synth.genome(base, i, d), (1, 2, 3)[i % 3] if contigs == "ragged" else contigs)):
                f.write(synth.fasta_bytes(c, i * 10 + j))
        paths.append(str(p))
    (tmp_path / "list.txt").write_text("\n".join(paths) + "\n")


def dumps(tmp_path, prefix):
    return {k: (tmp_path / (prefix + "." + k)).read_bytes() for k in STREAMS}


# 0.3 %: every contig extends the reference, none with its reverse complement (from the third round on the round's finalize runs on the ranks' device-side
# verdicts); 1.5 %: with reverse complements; 0.1 %: none does; 6 %: contigs are given up as dissimilar and the round is redone from there on every rank;
# 11 / 12 files: the last round is short (a rank without targets) / full
@pytest.mark.parametrize("gpus,r,n,div,contigs", [(2, 2, 17, 0.003, 2), (2, 2, 11, 0.015, "ragged"), (2, 2, 12, 0.06, 2), (2, 3, 11, 0.001, 1),
                                                  (3, 1, 12, 0.015, "ragged"), (3, 2, 15, 0.06, "ragged"), (2, 3, 14, "mixed", 2), (3, 2, 17, "mixed", "ragged")])
def test_ranks_on_one_gpu_equal_the_single_gpu_rounds(tmp_path, gpus, r, n, div, contigs):
    write_collection(tmp_path, n, 70_000, div, contigs)
    one = run_tool(["c", "-R", str(gpus * r), "list.txt", "one"], str(tmp_path))
    many = run_tool(["c", "--gpus", str(gpus), "--exchange", "hostmem", "--shm-mb", "1", "-R", str(r), "list.txt", "many"], str(tmp_path))
    a, b = dumps(tmp_path, "one"), dumps(tmp_path, "many")
    for k in STREAMS:
        assert a[k] == b[k], k
    for line in ("exact matches total", "removed matches breaking gaps total", "swsMEM unmatched chars", "final unmatched chars"):
        assert [x for x in one.splitlines() if x.startswith(line)] == [x for x in many.splitlines() if x.startswith(line)], line
    spec = int([x for x in many.splitlines() if x.startswith("rounds finalized on the ranks' device-side verdicts")][0].split(":")[1])
    if div == 0.003:
        assert spec >= 2, many                              # 4 rounds of 2 x 2: the third and the fourth
    if div == 0.06:
        assert spec == 0


@pytest.mark.parametrize("gpus,r,empty", [(2, 1, (4, 9)), (2, 1, (3, 5, 7)), (3, 1, (6,)), (2, 2, (5, 6))])
def test_a_rank_whose_file_holds_no_record(tmp_path, gpus, r, empty):
    """files without a record among the targets (an empty file: kseq reports end of file at once, MGMP.cpp:16-35) with rounds of
    one or two targets per rank: in such a round one rank has nothing to match or emit, and in the next its emission slots stand
    differently from the other ranks' — the speculative finalize must come out the same on every rank (a rank that cannot
    queue it says so in the reduction of the verdicts), the streams must equal the single-GPU rounds'"""
    write_collection(tmp_path, 14, 70_000, 0.003, 1)
    for i in empty:
        (tmp_path / ("g%02d.fa" % i)).write_bytes(b"")
    one = run_tool(["c", "-R", str(gpus * r), "list.txt", "one"], str(tmp_path))
    many = run_tool(["c", "--gpus", str(gpus), "--exchange", "hostmem", "--shm-mb", "1", "-R", str(r), "list.txt", "many"], str(tmp_path))
    a, b = dumps(tmp_path, "one"), dumps(tmp_path, "many")
    for k in STREAMS:
        assert a[k] == b[k], k
    for line in ("exact matches total", "final unmatched chars"):
        assert [x for x in one.splitlines() if x.startswith(line)] == [x for x in many.splitlines() if x.startswith(line)], line


def test_window_sized_rounds_are_divided_among_the_ranks(tmp_path):
    """without -R a round is what the reference's rules let be in flight — min(64, window / largest target) — whatever the number of
    GPUs: two ranks take half of it each and write the bytes the single GPU writes (a 16 MiB buffer: window 1 MiB, 200 kbp genomes,
    rounds of 5 on one GPU, 2 per rank on two = rounds of 4: the single-GPU run is given that size to compare)"""
    write_collection(tmp_path, 40, 200_000, 0.003, 1)
    many = run_tool(["c", "--ref-factor", "4", "--gpus", "2", "--exchange", "hostmem", "--shm-mb", "2", "list.txt", "many"], str(tmp_path))
    assert "rounds of 2 targets per GPU; reference extension bytes dropped at the sliding window's end: 0" in many, many
    run_tool(["c", "--ref-factor", "4", "-R", "4", "list.txt", "one"], str(tmp_path))
    a, b = dumps(tmp_path, "one"), dumps(tmp_path, "many")
    for k in STREAMS:
        assert a[k] == b[k], k


def test_rccl_with_one_rank_equals_the_plain_loop(tmp_path):
    """the RCCL transport on the hardware at hand: one rank — communicators, the two collectives' streams, the on-stream
    reduction inside the speculative finalize, the gather — against the loop without an exchange"""
    write_collection(tmp_path, 13, 70_000, 0.003, 2)
    run_tool(["c", "-R", "3", "list.txt", "one"], str(tmp_path))
    out = run_tool(["c", "--gpus", "1", "-R", "3", "list.txt", "x"], str(tmp_path), env={"MBGC_HIP_EXCHANGE": "1"})
    a, b = dumps(tmp_path, "one"), dumps(tmp_path, "x")
    for k in STREAMS:
        assert a[k] == b[k], k
    assert "rounds finalized on the ranks' device-side verdicts: 3" in out or "verdicts: 2" in out, out


def test_bench_mode_with_ranks(tmp_path):
    write_collection(tmp_path, 17, 300_000, 0.003, 1)
    out = run_tool(["c", "--bench", "--warmup", "1", "--gpus", "2", "--exchange", "hostmem", "-R", "2", "list.txt", "x"], str(tmp_path))
    d = json.loads(out.strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["rounds"] == 3 and d["bases"] == 12 * 300_000 and d["value"] > 0
    assert d["rounds_finalized_on_device_verdicts"] >= 1


def test_a_wrapping_buffer_and_the_head_of_the_round(tmp_path):
    """a buffer that wraps several times (--ref-factor 1: 4 MiB for 1 Mbp genomes): after the wrap a round's locks stand
    one window ahead of the loading position, loadRef clips there, and the ranks exchange only that head of the round's
    extensions (mbgc_xchg_bcast_heads_begin) — streams equal to the single-GPU rounds'"""
    write_collection(tmp_path, 25, 1_000_000, 0.003, 1, seed=67)
    one = run_tool(["c", "--ref-factor", "1", "-R", "4", "list.txt", "one"], str(tmp_path))
    many = run_tool(["c", "--ref-factor", "1", "--gpus", "2", "--exchange", "hostmem", "--shm-mb", "1", "-R", "2", "list.txt", "many"], str(tmp_path))
    a, b = dumps(tmp_path, "one"), dumps(tmp_path, "many")
    for k in STREAMS:
        assert a[k] == b[k], k
    heads = int([x for x in many.splitlines() if x.startswith("rounds whose extension exchange carried only the loadable head")][0].split(":")[1])
    assert heads >= 2, many
    # the A/B for the head-only exchange: every round's extensions gathered whole (MBGC_HIP_GATHER_ALL=1) — the same bytes
    whole = run_tool(["c", "--ref-factor", "1", "--gpus", "2", "--exchange", "hostmem", "--shm-mb", "1", "-R", "2", "list.txt", "whole"], str(tmp_path),
                     env={"MBGC_HIP_GATHER_ALL": "1"})
    c = dumps(tmp_path, "whole")
    for k in STREAMS:
        assert a[k] == c[k], k
    assert "rounds whose extension exchange carried only the loadable head: 0" in whole


# ---- as many ranks as a one-GPU box lets share its card (five beside this process): the shape of BASELINE configs[3] — file per
# rank, replicated index, extensions exchanged in target order, streams to rank 0 — rehearsed over host shared memory. Eight ranks
# run over gloo on the CPU (tests/test_rounds_cpu.py::test_eight_ranks_gloo).
@pytest.mark.parametrize("gpus,r,n,div,contigs,factor", [(5, 1, 21, 0.003, 1, None), (5, 2, 26, 0.015, "ragged", None), (5, 1, 31, 0.003, 1, "1"),
                                                         (4, 1, 17, 0.06, 2, None)])
def test_five_ranks_on_one_gpu_equal_the_single_gpu_rounds(tmp_path, gpus, r, n, div, contigs, factor):
    """five (four) ranks: similar genomes (finalize on the ranks' verdicts), reverse-complement extensions with ragged contig
    layouts, a buffer that wraps several times (--ref-factor 1: the head-only exchange), dissimilar contigs that cut rounds
    between ranks — every stream equal to the single GPU's rounds of gpus x r"""
    write_collection(tmp_path, n, 300_000 if factor else 70_000, div, contigs)
    extra = ["--ref-factor", factor] if factor else []
    one = run_tool(["c"] + extra + ["-R", str(gpus * r), "list.txt", "one"], str(tmp_path))
    many = run_tool(["c"] + extra + ["--gpus", str(gpus), "--exchange", "hostmem", "--shm-mb", "2", "-R", str(r), "list.txt", "many"], str(tmp_path))
    a, b = dumps(tmp_path, "one"), dumps(tmp_path, "many")
    for k in STREAMS:
        assert a[k] == b[k], k
    for line in ("exact matches total", "final unmatched chars"):
        assert [x for x in one.splitlines() if x.startswith(line)] == [x for x in many.splitlines() if x.startswith(line)], line
    if factor:
        heads = int([x for x in many.splitlines() if x.startswith("rounds whose extension exchange carried only the loadable head")][0].split(":")[1])
        assert heads >= 1, many


def test_five_ranks_with_a_record_less_file(tmp_path):
    write_collection(tmp_path, 22, 70_000, 0.003, 1)
    for i in (4, 9, 13):
        (tmp_path / ("g%02d.fa" % i)).write_bytes(b"")
    one = run_tool(["c", "-R", "5", "list.txt", "one"], str(tmp_path))
    many = run_tool(["c", "--gpus", "5", "--exchange", "hostmem", "--shm-mb", "1", "-R", "1", "list.txt", "many"], str(tmp_path))
    a, b = dumps(tmp_path, "one"), dumps(tmp_path, "many")
    for k in STREAMS:
        assert a[k] == b[k], k


def _children(pid):
    out = subprocess.run(["ps", "-o", "pid=", "--ppid", str(pid)], capture_output=True, text=True).stdout.split()
    return [int(x) for x in out]


def test_a_killed_rank_ends_the_run_within_seconds(tmp_path):
    """kill -9 on one of four ranks in the middle of a run: the others must not sit in an exchange forever — rank 0 notices the
    child's abnormal end, raises the shared failure flag, every rank leaves, and the tool exits non-zero well within 30 s"""
    import signal
    import time
    write_collection(tmp_path, 61, 1_500_000, 0.003, 1)
    p = subprocess.Popen(["timeout", "-k", "10", "120", TOOL, "c", "--gpus", "4", "--exchange", "hostmem", "--shm-mb", "4", "-R", "1", "list.txt", "x"],
                         cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    t0 = time.monotonic()
    kids = []
    while time.monotonic() - t0 < 20 and len(kids) < 3:                # the tool (child of `timeout`) forks ranks 1..3
        tool = _children(p.pid)
        kids = _children(tool[0]) if tool else []
        time.sleep(0.05)
    assert len(kids) == 3, kids
    time.sleep(0.7)                                                     # into the rounds
    assert p.poll() is None, "the run ended before a rank could be killed: make it longer"
    os.kill(kids[1], signal.SIGKILL)
    t1 = time.monotonic()
    out, err = p.communicate(timeout=60)
    assert p.returncode != 0 and time.monotonic() - t1 < 30, (p.returncode, time.monotonic() - t1, err[-500:])


def test_bench_py_reports_a_killed_rank(tmp_path):
    """`python bench.py --gpus 4` (its own spawner; the four ranks on this one device over gloo): rank 3 killed in the middle —
    the run ends within 30 s, non-zero, and rank 0's place is taken by a line with "value": null, the rank, and how many ranks
    had come up"""
    import glob
    import signal
    import sys
    import time
    env = dict(os.environ, MBGC_BENCH_ONE_DEVICE="1", MBGC_BENCH_BACKEND="gloo", TMPDIR=str(tmp_path), MBGC_BENCH_PG_TIMEOUT="60")
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "150", "--warmup", "2", "--round", "1",
                          "--length", "1000000", "--cpu-sample", "0"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    t0 = time.monotonic()
    ups = []
    while time.monotonic() - t0 < 150 and len(ups) < 4 and p.poll() is None:
        ups = glob.glob(str(tmp_path / "mbgc_bench_*" / "rank*.up"))
        time.sleep(0.2)
    assert len(ups) == 4, (ups, p.poll())
    pids = [int(x) for x in open(glob.glob(str(tmp_path / "mbgc_bench_*" / "pids"))[0]).read().split()]
    time.sleep(1.0)                                                     # into the rounds
    assert p.poll() is None, "the run ended before a rank could be killed: make it longer"
    os.kill(pids[3], signal.SIGKILL)
    t1 = time.monotonic()
    out, err = p.communicate(timeout=90)
    dt = time.monotonic() - t1
    assert p.returncode != 0 and dt < 30, (p.returncode, dt, err[-800:])
    lines = [json.loads(l) for l in out.splitlines() if l.startswith("{")]
    assert lines and lines[-1]["value"] is None and "rank 3" in lines[-1]["error"] and lines[-1]["rccl_ranks_seen"] == 4, lines


def test_verify_is_refused_with_several_ranks(tmp_path):
    """--verify with --gpus N used to be accepted and silently skipped (the run ended with "verified on the device: 0 contigs" and
    exit code 0): it is refused now, with the way to get the check"""
    write_collection(tmp_path, 5, 70_000, 0.003, 1)
    r = subprocess.run([TOOL, "c", "--verify", "--gpus", "2", "--exchange", "hostmem", "-R", "1", "list.txt", "x"], cwd=str(tmp_path), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "--verify" in r.stderr and "single-GPU" in r.stderr, (r.returncode, r.stderr[-300:])

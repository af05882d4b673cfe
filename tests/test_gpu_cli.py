"""End to end through the C++ host classes (mbgc_amd/host: MBGC_Encoder / MultipleGenomeMatchingProcessor /
SlidingWindowSparseEMMatcher facade over the C ABI) and the mbgc-hip tool:
  * BASELINE.json configs[0]: the reference's three bundled Listeria genomes, `-t1` schedule — every raw
    stream must equal what the reference CLI (`mbgc-dev c -t1` + `v -D`, run in the build container)
    produced (digests in tests/golden/listeria/expected_t1.json);
  * the round schedule on synthetic multi-contig files against the oracle-driven reference loop."""
import hashlib
import json
import lzma
import os
import re
import subprocess

import numpy as np
import pytest

import _driver
import _orc
from mbgc_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "mbgc_amd", "mbgc-hip")
LIST = os.path.join(ROOT, "tests", "golden", "listeria")


def run_tool(args, cwd):
    r = subprocess.run([TOOL] + args, cwd=cwd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    return r.stdout


def test_listeria_t1_streams_equal_reference_cli(tmp_path):
    exp = json.load(open(os.path.join(LIST, "expected_t1.json")))
    names = []
    for f in exp["files"]:
        data = lzma.open(os.path.join(LIST, f + ".xz")).read()
        (tmp_path / f).write_bytes(data)
        names.append(str(tmp_path / f))
    (tmp_path / "seqlist.txt").write_text("\n".join(names) + "\n")
    out = run_tool(["c", "-t1", "seqlist.txt", "lm"], str(tmp_path))
    assert "exact matches total: 29731" in out
    assert "removed matches breaking gaps total: 1418" in out
    assert "swsMEM unmatched chars: 5009573" in out
    assert "final unmatched chars: 3652988" in out
    for name, e in exp["streams"].items():
        b = (tmp_path / ("lm." + name)).read_bytes()
        assert len(b) == e["bytes"], name
        assert hashlib.md5(b).hexdigest() == e["md5"], name


def test_listeria_m3_streams_equal_reference_cli(tmp_path):
    """`mbgc c -m3 -t1`: the max mode — sequential matching with its own margins and, at the end, the reverse-complement pass
    over the literal stream (rcMapOff / rcMapLen, SimpleSequenceMatcher::rcMatchSequence) — against the reference CLI's dumps"""
    exp = json.load(open(os.path.join(LIST, "expected_m3.json")))
    names = []
    for f in exp["files"]:
        (tmp_path / f).write_bytes(lzma.open(os.path.join(LIST, f + ".xz")).read())
        names.append(str(tmp_path / f))
    (tmp_path / "seqlist.txt").write_text("\n".join(names) + "\n")
    run_tool(["c", "-m", "3", "-t1", "seqlist.txt", "lm"], str(tmp_path))
    for name, e in exp["streams"].items():
        b = (tmp_path / ("lm." + name)).read_bytes()
        assert len(b) == e["bytes"], (name, len(b), e["bytes"])
        assert hashlib.md5(b).hexdigest() == e["md5"], name


def split(g, k):
    cuts = [0] + [g.size * i // k + (7 * i) % 13 for i in range(1, k)] + [g.size]
    return [g[cuts[i]:cuts[i + 1]] for i in range(k)]


def gzip_member(data):
    import gzip
    return gzip.compress(data, 6)


def gzip_open_bytes(path):
    import gzip
    return gzip.open(path).read()


def crlf(c):
    """the contig as kseq_read_lossless_fasta returns it from a CRLF file: the carriage returns stay (utils/kseq.h:233-274,
    ks_getuntil2 with loosy = false) — 80 bases, CR, 80 bases, CR ..."""
    n = c.size
    out = np.empty(n + (n + 79) // 80, dtype=np.uint8)
    k = 0
    for s0 in range(0, n, 80):
        seg = c[s0:s0 + 80]
        out[k:k + seg.size] = seg
        out[k + seg.size] = 13
        k += seg.size + 1
    return out[:k]


# divergence 1.5 %: every contig extends the reference (the speculative finalize applies from the second round on);
# 0.1 %: none does; 6 %: dissimilar contigs are given up and their targets matched again in units; "mixed": every third genome 7 %
# from the rest, the others 0.4 % — rounds of 6 and 7 with kept targets in front of, between and behind stopped ones (units of
# one and two targets); "crlf": CRLF line ends; "gz": gzip files
@pytest.mark.parametrize("args,rs,div", [(["-t1"], 0, 0.015), (["-R", "3"], 3, 0.015), (["-R", "8"], 8, 0.015), (["-R", "2"], 2, 0.001),
                                         (["-R", "3"], 3, 0.06), (["-R", "6"], 6, "mixed"), (["-R", "7"], 7, "mixed"), (["-R", "5"], 5, 0.06),
                                         (["-m", "0", "-R", "8"], 8, "mixed2"), (["-m", "2", "-R", "5"], 5, "mixed2"),
                                         (["-R", "3"], 3, "crlf"), (["-t1"], 0, "crlf"), (["-R", "3"], 3, "gz"), (["-t1"], 0, "gz")])
def test_synthetic_files_equal_oracle_driver(tmp_path, args, rs, div):
    base = synth.base_codes(70_000, 55)
    cr, gz = div == "crlf", div == "gz"
    mode = int(args[args.index("-m") + 1]) if "-m" in args else 1
    if div == "mixed":
        gs = [synth.genome(base, i, 0.07 if i % 3 == 2 else 0.004) for i in range(16)]
    elif div == "mixed2":       # two genomes of three far from the rest: -m0 goes on in units of up to five stopped targets, -m2 in units of one
        gs = [synth.genome(base, i, 0.004 if i % 3 == 1 else 0.07) for i in range(18)]
    else:
        gs = [synth.genome(base, i, 0.015 if cr or gz else div) for i in range(8)]
    files = [split(g, 2) for g in gs]
    paths = []
    for i, contigs in enumerate(files):
        p = tmp_path / ("g%02d.fa%s" % (i, ".gz" if gz else ""))
        with open(p, "wb") as f:
            for j, c in enumerate(contigs):
                data = synth.fasta_bytes(c, i * 10 + j)
                data = data.replace(b"\n", b"\r\n") if cr else data
                # gz: mgmpInOpen inflates what starts with the gzip magic, member after member (every record its own member
                # in the odd files, one member per file in the even ones)
                f.write(gzip_member(data) if gz and i % 2 else data)
        if gz and i % 2 == 0:
            p.write_bytes(gzip_member(p.read_bytes()))
        paths.append(str(p))
    if cr:
        files = [[crlf(c) for c in contigs] for contigs in files]
    (tmp_path / "list.txt").write_text("\n".join(paths) + "\n")
    run_tool(["c"] + args + ["list.txt", "out"], str(tmp_path))
    fsize = len(gzip_open_bytes(paths[0])) if gz else os.path.getsize(paths[0])      # (the reference sizes by the inflated bytes, MGMP.cpp:109)
    if rs == 0:
        lim, _ = _driver.ref_length_limit(len(files), fsize)
        o = _orc.OracleMatcher(lim)
        oe = _orc.OracleEmitter(o)
        res = _driver.encode_sequential(o, oe, files)
        streams = oe.streams()
        g0lit = files[0][0].tobytes() + b"\xa2"
    else:
        lim, _ = _driver.ref_length_limit(len(files), sum(c.size for c in files[0]), mode=mode)
        o = _orc.OracleMatcher(lim, skip_margin=24 if mode >= 2 else 16)
        res = _driver.encode_rounds(o, lambda: _orc.OracleEmitter(o, _orc.emit_params(mode)), files[0], files[1:], rs, _driver.Policy(mode))
        streams = res["streams"]
        g0lit = b"".join(c.tobytes() + b"\xa2" for c in files[0])
    got = {k: (tmp_path / ("out." + k)).read_bytes() for k in ("literals", "mapOff", "mapOff5th", "mapLen", "gapDelta", "flags", "locksPos", "refExtSize")}
    assert got["literals"] == g0lit + streams["literals"]
    for k in ("mapOff", "mapOff5th", "mapLen", "gapDelta", "flags"):
        assert got[k] == streams[k], k
    assert got["locksPos"] == res["locks"] and got["refExtSize"] == res["refExtSize"]


def test_malformed_fasta_is_refused(tmp_path):
    """validate_kseq_status (MGMP.cpp:16-35): the tool prints the reference's message and exits with a failure"""
    base = synth.base_codes(30_000, 56)
    good = tmp_path / "a.fa"
    good.write_bytes(synth.fasta_bytes(synth.genome(base, 0, 0.01), 0))
    ragged = tmp_path / "b.fa"
    ragged.write_bytes(b">x\nACGTACGT\nACG\nACGTACGT\n")               # a short line in the middle: not well-formed
    nofasta = tmp_path / "c.fa"
    nofasta.write_bytes(b"ACGT\n")
    for bad, msg in ((ragged, "inconsistent line length"), (nofasta, "expected FASTA format")):
        (tmp_path / "l.txt").write_text("%s\n%s\n" % (good, bad))
        r = subprocess.run([TOOL, "c", "-R", "2", "l.txt", "o"], cwd=str(tmp_path), capture_output=True, text=True, timeout=120)
        assert r.returncode != 0 and msg in r.stderr, r.stderr


def test_bench_mode_reports_throughput(tmp_path):
    """mbgc-hip c --bench: every round resident in HBM, the rounds after the warm-up timed (the C++ host on the same path
    bench.py measures through the Python driver)"""
    base = synth.base_codes(400_000, 57)
    paths = []
    for i in range(13):
        p = tmp_path / ("g%02d.fa" % i)
        p.write_bytes(synth.fasta_bytes(synth.genome(base, i, 0.01), i))
        paths.append(str(p))
    (tmp_path / "list.txt").write_text("\n".join(paths) + "\n")
    out = run_tool(["c", "--bench", "--warmup", "1", "-R", "4", "list.txt", "x"], str(tmp_path))
    d = json.loads(out.strip().splitlines()[-1])
    assert d["rounds"] == 2 and d["bases"] == 8 * 400_000 and d["value"] > 0


@pytest.mark.parametrize("blocks", [1, 2])
def test_backend_section_is_read_back_by_the_reference(tmp_path, blocks):
    """`--backend` (and `--backend-blocks 2`: twice the reference's block counts): the tool's streams through the job table + container framing of include/mbgc_backend.h with the
    reference's own PPMd7 / LZMA (oracle/_ref) as the leaf coders; the reference's reader
    (readCompressedCollectiveParallel, coders/CodersLib.cpp:417-478) must give back the dumps, and the section must weigh
    what the reference's archive of the same three genomes weighs (1 016 021 bytes) minus its name /
    header streams and parameters"""
    import ctypes as C
    import _refh
    if not _refh.available():
        pytest.skip("oracle/_ref not built")
    exp = json.load(open(os.path.join(LIST, "expected_t1.json")))
    names = []
    for f in exp["files"]:
        (tmp_path / f).write_bytes(lzma.open(os.path.join(LIST, f + ".xz")).read())
        names.append(str(tmp_path / f))
    (tmp_path / "seqlist.txt").write_text("\n".join(names) + "\n")
    out = run_tool(["c", "-t1", "--backend", os.path.join(ROOT, "oracle", "_ref", "libmbgc_coders.so"), "--backend-threads", "1",
                    "--backend-blocks", str(blocks), "seqlist.txt", "lm"], str(tmp_path))
    assert "backend:" in out and ("%d x the reference's blocks" % blocks) in out
    section = (tmp_path / "lm.collective").read_bytes()
    assert 900_000 < len(section) < (1_016_021 if blocks == 1 else 1_030_000)        # (models restarted twice as often: a few hundred bytes more)
    R = _refh.lib()
    order = [None, None, None, None, None, "factors", "literals", "locksPos", "gapDelta", "flags", "mapOff", "mapLen", "refExtSize"]
    sizes = (C.c_uint64 * len(order))()
    cap = 16 << 20
    buf = C.create_string_buffer(cap)
    R.refbk_read_collective.restype = C.c_uint64
    total = R.refbk_read_collective(section, C.c_uint64(len(section)), len(order), sizes, buf, C.c_uint64(cap))
    assert total <= cap
    at = 0
    for name, n in zip(order, sizes):
        got = buf.raw[at: at + n]
        at += n
        if name is None:
            assert n == 0
        elif name == "factors":
            assert got == bytes([128, 8] * 3)                           # one pair per file in the -t1 schedule (MGMP.cpp:252-255)
        else:
            assert got == (tmp_path / ("lm." + name)).read_bytes(), name


def test_backend_beside_the_matching_is_read_back_by_the_reference(tmp_path):
    """`--backend-overlap 1`: the incremental form of include/mbgc_backend.h under the round loop — blocks of 1 MiB of the
    split streams handed to the coder threads every eight targets while the rounds go on; the reference's reader gives back
    the streams the same run dumps"""
    import ctypes as C
    import _refh
    if not _refh.available():
        pytest.skip("oracle/_ref not built")
    base = synth.base_codes(1_500_000, 91)
    names = []
    for i in range(33):
        p = tmp_path / ("g%02d.fa" % i)
        p.write_bytes(synth.fasta_bytes(synth.genome(base, i, 0.03), i))
        names.append(str(p))
    (tmp_path / "list.txt").write_text("\n".join(names) + "\n")
    out = run_tool(["c", "-R", "4", "--backend", os.path.join(ROOT, "oracle", "_ref", "libmbgc_coders.so"), "--backend-threads", "4",
                    "--backend-overlap", "1", "list.txt", "ov"], str(tmp_path))
    m = re.search(r"blocks of 1 MiB, (\d+) of them coded while the matching ran", out)
    assert m, out
    section = (tmp_path / "ov.collective").read_bytes()
    R = _refh.lib()
    order = [None, None, None, None, None, "factors", "literals", "locksPos", "gapDelta", "flags", "mapOff", "mapLen", "refExtSize"]
    sizes = (C.c_uint64 * len(order))()
    cap = 64 << 20
    buf = C.create_string_buffer(cap)
    R.refbk_read_collective.restype = C.c_uint64
    total = R.refbk_read_collective(section, C.c_uint64(len(section)), len(order), sizes, buf, C.c_uint64(cap))
    assert total <= cap
    at = 0
    for name, n in zip(order, sizes):
        got = buf.raw[at: at + n]
        at += n
        if name is None:
            assert n == 0
        elif name != "factors":
            assert got == (tmp_path / ("ov." + name)).read_bytes(), name
    assert len((tmp_path / "ov.literals").read_bytes()) > (2 << 20)          # (more than one block of literals)
    plain = run_tool(["c", "-R", "4", "--backend", os.path.join(ROOT, "oracle", "_ref", "libmbgc_coders.so"), "--backend-threads", "4", "list.txt", "pl"],
                     str(tmp_path))
    assert "backend:" in plain
    for name in order:
        if name and name != "factors":
            assert (tmp_path / ("pl." + name)).read_bytes() == (tmp_path / ("ov." + name)).read_bytes(), name


def test_rounds_of_a_larger_collection_equal_the_oracle_loop(tmp_path):
    """41 files of 1 Mbp in rounds of 8 through the pipelined host (read-ahead thread, three round slots, two emissions in
    flight, the speculative finalize from the third round on) against the oracle-driven reference loop"""
    base = synth.base_codes(1_000_000, 58)
    gs = [synth.genome(base, i, 0.006) for i in range(41)]
    paths = []
    for i, g in enumerate(gs):
        p = tmp_path / ("g%02d.fa" % i)
        p.write_bytes(synth.fasta_bytes(g, i))
        paths.append(str(p))
    (tmp_path / "list.txt").write_text("\n".join(paths) + "\n")
    run_tool(["c", "-R", "8", "list.txt", "out"], str(tmp_path))
    lim, _ = _driver.ref_length_limit(len(gs), gs[0].size)
    o = _orc.OracleMatcher(lim)
    res = _driver.encode_rounds(o, lambda: _orc.OracleEmitter(o), [gs[0]], [[g] for g in gs[1:]], 8)
    got = {k: (tmp_path / ("out." + k)).read_bytes() for k in ("literals", "mapOff", "mapOff5th", "mapLen", "gapDelta", "flags", "locksPos", "refExtSize")}
    assert got["literals"] == gs[0].tobytes() + b"\xa2" + res["streams"]["literals"]
    for k in ("mapOff", "mapOff5th", "mapLen", "gapDelta", "flags"):
        assert got[k] == res["streams"][k], k
    assert got["locksPos"] == res["locks"] and got["refExtSize"] == res["refExtSize"]
    o.close()


def test_a_missing_file_in_a_later_round_is_reported(tmp_path):
    """the read-ahead thread meets a file that does not exist: the tool says so (MGMP.cpp:7-14's message) and fails"""
    base = synth.base_codes(30_000, 59)
    paths = []
    for i in range(7):
        p = tmp_path / ("g%02d.fa" % i)
        p.write_bytes(synth.fasta_bytes(synth.genome(base, i, 0.01), i))
        paths.append(str(p))
    paths.insert(5, str(tmp_path / "nowhere.fa"))
    (tmp_path / "l.txt").write_text("\n".join(paths) + "\n")
    r = subprocess.run([TOOL, "c", "-R", "2", "l.txt", "o"], cwd=str(tmp_path), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "cannot open file" in r.stderr and "nowhere.fa" in r.stderr, r.stderr


@pytest.mark.parametrize("rs", [1, 4, 7])
def test_a_wrapping_buffer_equals_the_oracle_loop(tmp_path, rs):
    """--ref-factor 1: a 4 MiB circular buffer for 1 Mbp genomes — it wraps every fourth target, loads are clipped at the
    oldest lock, an emission's text is overwritten while its second phase may still be waiting to be queued (the guard hands
    it over first) — 25 targets in rounds of 1, 4 and 7 through the pipelined host against the oracle-driven loop"""
    base = synth.base_codes(1_000_000, 67)
    gs = [synth.genome(base, i, 0.003) for i in range(26)]
    paths = []
    for i, g in enumerate(gs):
        p = tmp_path / ("g%02d.fa" % i)
        p.write_bytes(synth.fasta_bytes(g, i))
        paths.append(str(p))
    (tmp_path / "list.txt").write_text("\n".join(paths) + "\n")
    run_tool(["c", "--ref-factor", "1", "-R", str(rs), "list.txt", "out"], str(tmp_path))
    lim = 2 * max(gs[0].size, 1 << 21)                       # initMatcher with referenceFactor = 1 (MGMP.cpp:154-158)
    o = _orc.OracleMatcher(lim)
    res = _driver.encode_rounds(o, lambda: _orc.OracleEmitter(o), [gs[0]], [[g] for g in gs[1:]], rs)
    got = {k: (tmp_path / ("out." + k)).read_bytes() for k in ("literals", "mapOff", "mapOff5th", "mapLen", "gapDelta", "flags", "locksPos", "refExtSize")}
    assert got["literals"] == gs[0].tobytes() + b"\xa2" + res["streams"]["literals"]
    for k in ("mapOff", "mapOff5th", "mapLen", "gapDelta", "flags"):
        assert got[k] == res["streams"][k], k
    assert got["locksPos"] == res["locks"] and got["refExtSize"] == res["refExtSize"]
    assert o.loaded_ref_length() > lim                       # (it did wrap: rounds of 7 load the least, clipped at the oldest lock)
    o.close()


def test_rounds_sized_by_the_sliding_window_drop_nothing(tmp_path):
    """without -R the host sizes its rounds by the reference's own rules for targets in flight (the sliding window, at most 64:
    MGMP_Params::roundSize): 200 kbp genomes against a 16 MiB buffer (--ref-factor 4, window 1 MiB) make rounds of 5, the
    buffer wraps, no extension byte is dropped and the streams equal the oracle-driven loop with that round size; rounds of 8
    lose bytes at the window's end (SlidingWindowSparseEMMatcher.cpp:412-417,433) and the tool says how many"""
    base = synth.base_codes(200_000, 73)
    gs = [synth.genome(base, i, 0.004) for i in range(100)]
    paths = []
    for i, g in enumerate(gs):
        p = tmp_path / ("g%03d.fa" % i)
        p.write_bytes(synth.fasta_bytes(g, i))
        paths.append(str(p))
    (tmp_path / "list.txt").write_text("\n".join(paths) + "\n")
    out = run_tool(["c", "--ref-factor", "4", "list.txt", "out"], str(tmp_path))
    assert "rounds of 5 targets; reference extension bytes dropped at the sliding window's end: 0" in out, out
    lim = 4 * 2 * (1 << 21)
    o = _orc.OracleMatcher(lim)
    res = _driver.encode_rounds(o, lambda: _orc.OracleEmitter(o), [gs[0]], [[g] for g in gs[1:]], 5)
    assert o.loaded_ref_length() > lim                       # (it wrapped)
    got = {k: (tmp_path / ("out." + k)).read_bytes() for k in ("literals", "mapOff", "mapLen", "gapDelta", "flags", "locksPos", "refExtSize")}
    assert got["literals"] == gs[0].tobytes() + b"\xa2" + res["streams"]["literals"]
    for k in ("mapOff", "mapLen", "gapDelta", "flags"):
        assert got[k] == res["streams"][k], k
    assert got["locksPos"] == res["locks"] and got["refExtSize"] == res["refExtSize"]
    out8 = run_tool(["c", "--ref-factor", "4", "-R", "8", "list.txt", "out8"], str(tmp_path))
    dropped = int([x for x in out8.splitlines() if x.startswith("rounds of 8 targets")][0].rsplit(":", 1)[1])
    assert dropped > 1_000_000, out8


def test_repeated_runs_write_the_same_bytes(tmp_path):
    """the host's threads (file readers, the upload + parse thread, the thread that appends the streams) and the two
    emissions in flight leave no room for timing: four runs over a buffer that wraps, with different numbers of reader
    threads, one digest"""
    import hashlib
    base = synth.base_codes(1_000_000, 71)
    paths = []
    for i in range(31):
        p = tmp_path / ("g%02d.fa" % i)
        p.write_bytes(synth.fasta_bytes(synth.genome(base, i, 0.004), i))
        paths.append(str(p))
    (tmp_path / "list.txt").write_text("\n".join(paths) + "\n")
    digests = set()
    for readers in ("1", "8", "3", "8"):
        r = subprocess.run([TOOL, "c", "--ref-factor", "2", "-R", "5", "list.txt", "o"], cwd=str(tmp_path), capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, MBGC_HIP_READERS=readers))
        assert r.returncode == 0, r.stderr[-1000:]
        h = hashlib.md5()
        for k in ("literals", "mapOff", "mapOff5th", "mapLen", "gapDelta", "flags", "locksPos", "refExtSize"):
            h.update((tmp_path / ("o." + k)).read_bytes())
        digests.add(h.hexdigest())
    assert len(digests) == 1


@pytest.mark.parametrize("k1,args,rs", [(15, ["-t1"], 0), (9, ["-R", "3"], 3), (7, ["-R", "2"], 2)])
def test_an_odd_sampling_step_through_the_cpp_host(tmp_path, k1, args, rs):
    """`mbgc -s <odd k1>` (MGMP.cpp:170-176: the base matcher class, identity-encoded table entries): the C++ host with `-s k1` writes
    the streams the oracle, driven through the reference's target loop with the same sampling step, writes"""
    base = synth.base_codes(90_000, 77)
    files = [split(synth.genome(base, i, 0.012), 2) for i in range(8)]
    paths = []
    for i, contigs in enumerate(files):
        p = tmp_path / ("g%d.fa" % i)
        p.write_bytes(b"".join(synth.fasta_bytes(c, i * 10 + j) for j, c in enumerate(contigs)))
        paths.append(str(p))
    (tmp_path / "list.txt").write_text("\n".join(paths) + "\n")
    run_tool(["c", "-s", str(k1)] + args + ["list.txt", "out"], str(tmp_path))
    if rs == 0:
        lim, _ = _driver.ref_length_limit(len(files), os.path.getsize(paths[0]))
        o = _orc.OracleMatcher(lim, k1=k1)
        oe = _orc.OracleEmitter(o)
        res = _driver.encode_sequential(o, oe, files)
        streams = oe.streams()
        g0lit = files[0][0].tobytes() + b"\xa2"
    else:
        lim, _ = _driver.ref_length_limit(len(files), sum(c.size for c in files[0]))
        o = _orc.OracleMatcher(lim, k1=k1)
        res = _driver.encode_rounds(o, lambda: _orc.OracleEmitter(o), files[0], files[1:], rs)
        streams = res["streams"]
        g0lit = b"".join(c.tobytes() + b"\xa2" for c in files[0])
    got = {k: (tmp_path / ("out." + k)).read_bytes() for k in ("literals", "mapOff", "mapLen", "gapDelta", "flags", "locksPos", "refExtSize")}
    assert got["literals"] == g0lit + streams["literals"]
    for k in ("mapOff", "mapLen", "gapDelta", "flags"):
        assert got[k] == streams[k], k
    assert got["locksPos"] == res["locks"] and got["refExtSize"] == res["refExtSize"]

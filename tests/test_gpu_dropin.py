"""The drop-in, end to end (skipped where oracle/_ref lacks the reference builds):
  * `oracle/_ref/mbgc-dropin` IS the reference's CLI — its own main.cpp, MBGC_Encoder, MultipleGenomeMatchingProcessor, backend and
    archive writer — compiled with INTEGRATION.md's patch (oracle/dropin/: the matcher class replaced by the facade over
    libmbgc_hip.so, processMatches forwarding to swsem_emit). `mbgc-dropin c -t1` on the reference's three Listeria genomes must
    write the archive the stock tool writes (md5 79b8acfe...), and the stock `mbgc d` must give the files back;
  * a collection-level round trip for ROUNDS: the matcher-side streams of `mbgc-hip c -R r` (one GPU, and two ranks over the
    host-memory exchange) on a buffer that wraps, with dissimilar contigs retried and bytes clipped at the window's end, are
    put into an archive — header-side streams and the parameter block taken from the reference's own run on the same files,
    the collective section framed by include/mbgc_backend.h around the reference's coders — and decoded by the STOCK decoder
    (MBGC_Decoder.cpp:535-675,1064-1172: decodeInit, decodeTarget, loadRef with the recorded lock positions): every file must
    come back byte for byte."""
import ctypes as C
import hashlib
import lzma
import os
import struct
import subprocess

import numpy as np
import pytest

import _refh
from mbgc_amd import synth
from test_backend_jobs import LEAF_FN, NST, Params, reference_leaf

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "mbgc_amd", "mbgc-hip")
DROPIN = os.path.join(ROOT, "oracle", "_ref", "mbgc-dropin")
LIST = os.path.join(ROOT, "tests", "golden", "listeria")
needs_ref = pytest.mark.skipif(not (_refh.available() and os.access(_refh.REF_MBGC, os.X_OK) and os.access(_refh.REF_MBGC_DEV, os.X_OK)),
                               reason="reference builds (oracle/_ref) not on this host")


def run(cmd, cwd, **kw):
    r = subprocess.run(cmd, cwd=cwd, capture_output=True, text=True, timeout=600, **kw)
    assert r.returncode == 0, (cmd, r.stdout[-1500:], r.stderr[-1500:])
    return r.stdout


@needs_ref
@pytest.mark.skipif(not os.access(DROPIN, os.X_OK), reason="oracle/_ref/mbgc-dropin not built (make -C oracle dropin)")
def test_the_patched_reference_cli_writes_the_stock_archive(tmp_path):
    exp = __import__("json").load(open(os.path.join(LIST, "expected_t1.json")))
    for f in exp["files"]:
        (tmp_path / f).write_bytes(lzma.open(os.path.join(LIST, f + ".xz")).read())
    (tmp_path / "seqlist.txt").write_text("\n".join(exp["files"]) + "\n")
    run([DROPIN, "c", "-t1", "seqlist.txt", "hip.mbgc"], str(tmp_path))
    arch = (tmp_path / "hip.mbgc").read_bytes()
    assert hashlib.md5(arch).hexdigest() == "79b8acfe0ded3f371e381c72b7d7c2bb"        # the stock `mbgc c -t1` archive, SURVEY.md §8c
    os.mkdir(tmp_path / "out")
    run([_refh.REF_MBGC, "d", "hip.mbgc", "out"], str(tmp_path))
    for f in exp["files"]:
        assert (tmp_path / "out" / f).read_bytes() == (tmp_path / f).read_bytes(), f
    # the -m3 preset through the same binary (the reverse-complement pass stays the reference's own on this side)
    run([DROPIN, "c", "-m3", "-t1", "seqlist.txt", "hip3.mbgc"], str(tmp_path))
    run([_refh.REF_MBGC, "c", "-m3", "-t1", "seqlist.txt", "ref3.mbgc"], str(tmp_path))
    assert (tmp_path / "hip3.mbgc").read_bytes() == (tmp_path / "ref3.mbgc").read_bytes()


@needs_ref
@pytest.mark.skipif(not os.access(DROPIN, os.X_OK), reason="oracle/_ref/mbgc-dropin not built (make -C oracle dropin)")
def test_the_patched_cli_refuses_the_parallel_schedule(tmp_path):
    """the reference's default schedule calls matchTexts / processMatches from several worker threads on the one matcher
    (MGMP.cpp:520-555); the device handle serves one such pair at a time — the patched binary says so and exits instead of
    writing an archive whose streams its threads mixed up (advisor, round 3)"""
    exp = __import__("json").load(open(os.path.join(LIST, "expected_t1.json")))
    for f in exp["files"]:
        (tmp_path / f).write_bytes(lzma.open(os.path.join(LIST, f + ".xz")).read())
    (tmp_path / "seqlist.txt").write_text("\n".join(exp["files"]) + "\n")
    r = subprocess.run([DROPIN, "c", "seqlist.txt", "hip.mbgc"], cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "one matchTexts/processMatches pair at a time" in r.stderr, (r.returncode, r.stderr[-500:])


def split(g, k):
    cuts = [0] + [g.size * i // k + (7 * i) % 13 for i in range(1, k)] + [g.size]
    return [g[cuts[i]:cuts[i + 1]] for i in range(k)]


STATS = struct.Struct("<IQQQIQBQ")        # writeStats, MBGC_Encoder.cpp:734-743: filesCount, totalFilesLength, refG0InitPos, largestFileLength,
                                          # largestContigSize, refFinalTotalLength, refBuffersCount, literals' size


@needs_ref
@pytest.mark.parametrize("args", [["-R", "5"], ["-R", "1"], [], ["--gpus", "2", "--exchange", "hostmem", "--shm-mb", "4", "-R", "3"]])
def test_a_rounds_archive_is_decoded_by_the_stock_decoder(tmp_path, args):
    base = synth.base_codes(300_000, 83)
    names = []
    for i in range(41):
        div = (0.004, 0.015, 0.06, 0.002)[i % 4] if i else 0.0          # 6 %: given up as dissimilar and retried; 1.5 %: reverse complements loaded too
        g = synth.genome(base, i, div)
        name = "g%02d.fa" % i
        with open(tmp_path / name, "wb") as f:
            for j, c in enumerate(split(g, 1 + i % 3)):
                f.write(synth.fasta_bytes(c, i * 10 + j))
        names.append(name)
    (tmp_path / "list.txt").write_text("\n".join(names) + "\n")
    # the reference's own run on the same files, same buffer (-o 1: reference factor 2 -> 8 MiB, it wraps): its parameter block,
    # statistics and header-side streams (file names, sequence counts, header templates, headers, line lengths, the two factors per target)
    run([_refh.REF_MBGC_DEV, "c", "-m1", "-o", "1", "list.txt", "ref.mbgc"], str(tmp_path))
    run([_refh.REF_MBGC_DEV, "v", "-D", "ref.mbgc"], str(tmp_path))
    ref_arch = (tmp_path / "ref.mbgc").read_bytes()
    side = {i: (tmp_path / ("ref.mbgc_dump_%02d" % i)).read_bytes() for i in range(7, 13)}
    # this repo's matcher-side streams for the same collection in rounds
    out = run([TOOL, "c", "--ref-factor", "2"] + args + ["list.txt", "hip"], str(tmp_path))
    ref_len = int([x for x in out.splitlines() if x.startswith("final reference length")][0].split(":")[1])
    s = {k: (tmp_path / ("hip." + k)).read_bytes() for k in ("literals", "locksPos", "gapDelta", "flags", "mapOff", "mapOff5th", "mapLen", "refExtSize")}
    streams = [b""] * NST
    for i in range(6):
        streams[i] = side[7 + i]
    streams[6], streams[9], streams[10], streams[11] = s["literals"], s["locksPos"], s["gapDelta"], s["flags"]
    streams[12], streams[13], streams[14], streams[15] = s["mapOff"], s["mapOff5th"], s["mapLen"], s["refExtSize"]
    # the collective section: include/mbgc_backend.h's job table and framing around the reference's leaf coders
    import __graft_entry__ as g
    L = C.CDLL(os.path.join(ROOT, "mbgc_amd", "libmbgc_host.so"))
    L.mbgc_backend_last_error.restype = C.c_char_p
    p = Params(coderMode=1, ultraStreamsCompression=0, k=32, enableExtensionsWithMismatches=1, mismatchesWithExclusion=1, sequentialMatching=0,
               rcRedundancyRemoval=0, frugal64bitLenEncoding=1, lazyDecompressionSupport=1, refFinalTotalLength=ref_len, numberOfThreads=4, blocksScale=0)
    data = (C.c_char_p * NST)(*[bytes(x) for x in streams])
    size = (C.c_uint64 * NST)(*[len(x) for x in streams])
    cb, _ = reference_leaf()
    sec, n = C.c_void_p(), C.c_uint64()
    assert L.mbgc_backend_compress_streams(C.byref(p), data, size, cb, None, 0, C.byref(sec), C.byref(n)) == 0, L.mbgc_backend_last_error()
    section = C.string_at(sec, n.value)
    L.mbgc_backend_free(sec)
    # the parameter block is the reference run's (same options); its length: what the reference writes in front of a section with these parameters
    R = _refh.lib()
    prefix = C.c_uint64()
    dummy = (C.c_char_p * NST)(*[b"x"] * NST)
    dsize = (C.c_uint64 * NST)(*[1] * NST)
    assert R.refbk_archive(str(tmp_path / "dummy.mbgc").encode(), 1, 2, 1, C.c_uint64(ref_len), dummy, dsize, C.byref(prefix)) == 0
    P = prefix.value
    st = list(STATS.unpack(ref_arch[P - STATS.size: P]))
    assert st[0] == 41 and st[6] == 1, st                                  # (the block really is where it was taken to be)
    st[5], st[7] = ref_len, len(s["literals"])
    (tmp_path / "hip.mbgc").write_bytes(ref_arch[: P - STATS.size] + STATS.pack(*st) + section)
    os.mkdir(tmp_path / "out")
    run([_refh.REF_MBGC, "d", "hip.mbgc", "out"], str(tmp_path))
    for name in names:
        assert (tmp_path / "out" / name).read_bytes() == (tmp_path / name).read_bytes(), name
    if args[:2] == ["-R", "5"]:
        assert "extension bytes dropped at the sliding window's end: 0" not in out      # this case clips at the window's end, and still decodes

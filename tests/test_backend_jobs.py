"""SURVEY.md §8(f) row 3 — the backend's job table and container framing (include/mbgc_backend.h, mbgc_amd/host/
backend_jobs.cpp) against the reference's own prepareAndCompressStreams (mbgccoder/MBGC_Encoder.cpp:641-710 +
coders/CodersLib.cpp) compiled into oracle/_ref: the same streams, the reference's PPMd7 / LZMA behind the leaf callback
(the unchanged host backend of north_star) — the collective section must be the reference's, byte for byte, in every
compression mode. Runs on the CPU (host code only)."""
import ctypes as C
import os
import tempfile

import numpy as np
import pytest

import _refh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NST = 16


class Leaf(C.Structure):
    _fields_ = [("coder", C.c_int), ("level", C.c_int), ("lc", C.c_int), ("lp", C.c_int), ("pb", C.c_int), ("fb", C.c_int),
                ("algo", C.c_int), ("numThreads", C.c_int), ("dictSize", C.c_uint32), ("memSize", C.c_uint32), ("order", C.c_int)]


class Params(C.Structure):
    _fields_ = [("coderMode", C.c_int), ("ultraStreamsCompression", C.c_int), ("k", C.c_int), ("enableExtensionsWithMismatches", C.c_int),
                ("mismatchesWithExclusion", C.c_int), ("sequentialMatching", C.c_int), ("rcRedundancyRemoval", C.c_int),
                ("frugal64bitLenEncoding", C.c_int), ("lazyDecompressionSupport", C.c_int), ("refFinalTotalLength", C.c_uint64),
                ("numberOfThreads", C.c_int), ("blocksScale", C.c_int)]


LEAF_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(Leaf), C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64))


def host():
    import __graft_entry__ as g
    g.build()
    L = C.CDLL(os.path.join(ROOT, "mbgc_amd", "libmbgc_host.so"))
    L.mbgc_backend_last_error.restype = C.c_char_p
    return L


def reference_leaf():
    R = _refh.lib()
    calls = []

    def leaf(ctx, c, src, n, dest, cap, dest_len):
        c = c.contents
        calls.append((c.coder, c.order if c.coder == 3 else c.fb))
        return R.refbk_leaf(c.coder, c.level, C.c_uint32(c.dictSize), c.lc, c.lp, c.pb, c.fb, c.algo, c.numThreads, C.c_uint32(c.memSize),
                            c.order, C.c_void_p(src), C.c_uint64(n), C.c_void_p(dest), C.c_uint64(cap), dest_len)
    return LEAF_FN(leaf), calls


def synthetic_streams(seed, scale):
    """bytes with the statistics of the real streams (so that every job compresses, and splits into blocks once it passes
    2^20 bytes per block): DNA literals with marks, small gap deltas, sparse flags, 32-bit offsets, frugal lengths"""
    rng = np.random.default_rng(seed)
    def dna(n):
        return np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)]
    s = [b""] * NST
    s[0] = b"".join(b"genomes/sample_%05d.fna.gz\xbb" % i for i in range(40))
    s[1] = np.full(40, 1, dtype="<u4").tobytes()
    s[2] = b">NZ_CP0\xa5.1 Listeria monocytogenes strain \xa5 chromosome, complete genome"
    s[3] = b"".join(b"%05d\xa5N%d\xa5\xa2" % (rng.integers(0, 99999), i) for i in range(40))
    s[4] = np.full(40, 80, dtype="<u8").tobytes()
    s[5] = bytes([128, 8] * 40)
    lit = dna(int(5_000_000 * scale))
    lit[rng.integers(0, lit.size, lit.size // 3000)] = 0xA5
    comp = np.zeros(256, dtype=np.uint8)
    comp[list(b"ACGT")] = list(b"TGCA")
    for k in range(6):                                             # reverse-complement repeats for the -m3 pass to find
        a, b, n = int(lit.size * (0.1 + 0.05 * k)), int(lit.size * (0.5 + 0.07 * k)), 300 + 400 * k
        seg = lit[a:a + n].copy()
        seg[seg == 0xA5] = ord("A")
        lit[a:a + n] = seg
        lit[b:b + n] = comp[seg[::-1]]
    s[6] = lit.tobytes()
    s[9] = np.sort(rng.integers(0, 1 << 31, 40)).astype("<u8").tobytes()
    s[10] = rng.choice(np.arange(8, dtype=np.uint8), int(2_500_000 * scale), p=[.55, .2, .1, .05, .04, .03, .02, .01]).tobytes()
    s[11] = (rng.random(int(4_500_000 * scale)) < 0.03).astype(np.uint8).tobytes()
    s[12] = np.sort(rng.integers(0, 1 << 31, int(800_000 * scale))).astype("<u4").tobytes()
    s[13] = rng.integers(0, 2, 1000).astype(np.uint8).tobytes()
    s[14] = rng.geometric(0.01, int(3_000_000 * scale)).astype("<u2").tobytes()
    s[15] = np.full(40, 5_000_001 & 0xFFFF, dtype="<u2").tobytes()
    return s


def both(L, mode, flags, streams, threads=1, ref_total=1 << 31, blocks_scale=0):
    R = _refh.lib()
    data = (C.c_char_p * NST)(*[bytes(x) for x in streams])
    size = (C.c_uint64 * NST)(*[len(x) for x in streams])
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "a.mbgc").encode()
        prefix = C.c_uint64()
        assert R.refbk_archive(path, mode, flags, threads, C.c_uint64(ref_total), data, size, C.byref(prefix)) == 0
        ref = open(path, "rb").read()[prefix.value:]
    # the same parameters as MBGC_Params::setCompressionMode leaves them (MBGC_Params.h:886-922)
    p = Params(coderMode=mode, ultraStreamsCompression=flags & 1, k=32, enableExtensionsWithMismatches=1,
               mismatchesWithExclusion=0 if mode == 0 else 1, sequentialMatching=1 if flags & 4 or mode == 3 else 0,
               rcRedundancyRemoval=1 if mode == 3 else 0, frugal64bitLenEncoding=0 if mode == 0 else 1,
               lazyDecompressionSupport=1 if flags & 2 else 0, refFinalTotalLength=ref_total, numberOfThreads=threads, blocksScale=blocks_scale)
    # prepareHeadersStreams appends the file separator to the (one) file's templates
    mine_streams = list(streams)
    mine_streams[2] = streams[2] + b"\xbb"
    if mode == 3:
        # -m3: the reverse-complement pass rewrites the literals and makes the two rc streams first (MBGC_Encoder.cpp:636-638)
        import _orc
        mine_streams[6], mine_streams[7], mine_streams[8], _ = _orc.rc_match_sequence(np.frombuffer(streams[6], dtype=np.uint8), 55)
        assert len(mine_streams[7]) > 8 and len(mine_streams[6]) < len(streams[6])
    data = (C.c_char_p * NST)(*[bytes(x) for x in mine_streams])
    size = (C.c_uint64 * NST)(*[len(x) for x in mine_streams])
    cb, calls = reference_leaf()
    out, n = C.c_void_p(), C.c_uint64()
    rc = L.mbgc_backend_compress_streams(C.byref(p), data, size, cb, None, 0, C.byref(out), C.byref(n))
    assert rc == 0, L.mbgc_backend_last_error()
    mine = C.string_at(out, n.value)
    L.mbgc_backend_free(out)
    return ref, mine, calls


pytestmark = pytest.mark.skipif(not _refh.available(), reason="oracle/_ref not built")


@pytest.mark.parametrize("mode,flags,scale", [(1, 2, 1.0), (0, 2, 1.0), (2, 2, 1.0), (3, 2 | 4, 0.6), (1, 0, 0.05), (1, 2 | 1, 0.3)])
def test_collective_section_equals_the_reference(mode, flags, scale):
    """modes 0..3 (block counts, orders, LZMA levels), lazy on/off, streams below the block threshold, the compound coder
    of the headers stream under ultraStreamsCompression (mode 3 keeps it in one block)"""
    L = host()
    if flags & 1:
        mode = 3                                                   # (with more than one block the reference's compound props race)
    ref, mine, calls = both(L, mode, flags, synthetic_streams(7 + mode, scale))
    assert len(ref) > 1000
    assert mine == ref, (len(mine), len(ref), next(i for i in range(min(len(mine), len(ref))) if mine[i] != ref[i]))
    assert {c for c, _ in calls} == {1, 3}                         # both coder families were really used


def test_fifth_byte_stream_is_enrolled_beyond_4g_and_empty_streams_are_zero_lengths():
    L = host()
    s = synthetic_streams(3, 0.02)
    s[10] = b""                                                    # an empty stream is eight zero bytes in the archive
    ref, mine, _ = both(L, 1, 2, s, ref_total=(1 << 32) + 5)
    assert mine == ref
    ref2, mine2, _ = both(L, 1, 2, s, ref_total=1 << 31)
    assert mine2 == ref2 and len(ref2) < len(ref)


def test_more_blocks_than_the_references_are_read_back_by_the_references_reader():
    """blocksScale = 3: every split stream in three times the reference's blocks — other bytes than the reference writes, the
    same streams out of its readCompressedCollectiveParallel"""
    L = host()
    streams = synthetic_streams(11, 1.0)
    ref, mine, _ = both(L, 1, 2, streams, blocks_scale=3)
    assert mine != ref and abs(len(mine) - len(ref)) < 0.02 * len(ref)
    R = _refh.lib()
    R.refbk_read_collective.restype = C.c_uint64
    enrolled = [st for st in range(NST) if st not in (7, 8, 13)]           # -m1 below 2^32: no rc streams, no 5th byte
    sizes = (C.c_uint64 * len(enrolled))()
    cap = sum(len(x) for x in streams) + 64
    buf = C.create_string_buffer(cap)
    total = R.refbk_read_collective(mine, C.c_uint64(len(mine)), len(enrolled), sizes, buf, C.c_uint64(cap))
    assert total <= cap
    at = 0
    for st, n in zip(enrolled, sizes):
        want = streams[st] + (b"\xbb" if st == 2 else b"")
        assert buf.raw[at: at + n] == want, st
        at += n


def test_job_table_query():
    L = host()
    p = Params(coderMode=1, k=32, enableExtensionsWithMismatches=1, mismatchesWithExclusion=1, lazyDecompressionSupport=1,
               frugal64bitLenEncoding=1, numberOfThreads=8)
    blocks, leaf, prim = C.c_int(), Leaf(), Leaf()
    assert L.mbgc_backend_job(C.byref(p), 6, C.byref(blocks), C.byref(leaf), C.byref(prim)) == 0          # literals, -m1
    assert (blocks.value, leaf.coder, leaf.order, leaf.memSize, prim.coder) == (2, 3, 5, 192 << 20, 0)
    assert L.mbgc_backend_job(C.byref(p), 14, C.byref(blocks), C.byref(leaf), C.byref(prim)) == 0         # mapLen
    assert (blocks.value, leaf.coder, leaf.lp, leaf.pb, leaf.fb, leaf.numThreads) == (5, 1, 1, 1, 128, 2)
    assert L.mbgc_backend_job(C.byref(p), 7, None, None, None) == -1                                       # rc streams: -m3 only
    assert L.mbgc_backend_job(C.byref(p), 13, None, None, None) == -1                                      # 5th byte: beyond 2^32 only


def _params(mode, flags, ref_total=1 << 31, threads=1):
    return Params(coderMode=mode, ultraStreamsCompression=flags & 1, k=32, enableExtensionsWithMismatches=1,
                  mismatchesWithExclusion=0 if mode == 0 else 1, sequentialMatching=1 if flags & 4 or mode == 3 else 0,
                  rcRedundancyRemoval=1 if mode == 3 else 0, frugal64bitLenEncoding=0 if mode == 0 else 1,
                  lazyDecompressionSupport=1 if flags & 2 else 0, refFinalTotalLength=ref_total, numberOfThreads=threads, blocksScale=0)


def _stream_api(L):
    L.mbgc_backend_stream_open.restype = C.c_void_p
    L.mbgc_backend_stream_open.argtypes = [C.POINTER(Params), LEAF_FN, C.c_void_p, C.c_int, C.c_uint64]
    L.mbgc_backend_stream_feed.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_uint64]
    L.mbgc_backend_stream_finish.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.mbgc_backend_stream_close.argtypes = [C.c_void_p]


def fed_in_pieces(L, p, streams, block_bytes, threads, piece, ref_total, seed=1, linger=0.0):
    """every stream handed over `piece` bytes at a time, the streams taking turns at random; -> section, blocks coded before finish"""
    _stream_api(L)
    cb, _ = reference_leaf()
    h = L.mbgc_backend_stream_open(C.byref(p), cb, None, threads, block_bytes)
    assert h, L.mbgc_backend_last_error()
    rng = np.random.default_rng(seed)
    at = [0] * NST
    live = [st for st in range(NST) if len(streams[st])]
    while live:
        st = live[int(rng.integers(0, len(live)))]
        chunk = streams[st][at[st]: at[st] + piece]
        assert L.mbgc_backend_stream_feed(h, st, chunk, len(chunk)) == 0, L.mbgc_backend_last_error()
        at[st] += len(chunk)
        if at[st] >= len(streams[st]):
            live.remove(st)
    if linger:                                                      # (the matching that is still going on)
        import time
        time.sleep(linger)
    out, n, early = C.c_void_p(), C.c_uint64(), C.c_uint64()
    assert L.mbgc_backend_stream_finish(h, C.c_uint64(ref_total), C.byref(out), C.byref(n), C.byref(early)) == 0, L.mbgc_backend_last_error()
    section = C.string_at(out, n.value)
    L.mbgc_backend_free(out)
    L.mbgc_backend_stream_close(h)
    return section, early.value


def read_back(section, streams, enrolled):
    R = _refh.lib()
    R.refbk_read_collective.restype = C.c_uint64
    sizes = (C.c_uint64 * len(enrolled))()
    cap = sum(len(x) for x in streams) + 64
    buf = C.create_string_buffer(cap)
    total = R.refbk_read_collective(section, C.c_uint64(len(section)), len(enrolled), sizes, buf, C.c_uint64(cap))
    assert total <= cap
    at = 0
    for st, n in zip(enrolled, sizes):
        assert buf.raw[at: at + n] == streams[st], st
        at += n


def test_streams_fed_piece_by_piece_below_one_block_give_the_references_section():
    """the incremental form (the backend beside the matching): with every stream below one block nothing is cut and the
    section is the reference's own, byte for byte — fed in 100 kB pieces, streams interleaved"""
    L = host()
    streams = synthetic_streams(5, 0.15)
    ref, mine, _ = both(L, 1, 2, streams)
    assert mine == ref
    fed = list(streams)
    fed[2] = streams[2] + b"\xbb"
    section, early = fed_in_pieces(L, _params(1, 2), fed, 1 << 20, 2, 100_000, 1 << 31)
    assert section == ref and early == 0


@pytest.mark.parametrize("mode,flags,threads", [(1, 2, 3), (0, 2, 1), (3, 2 | 4, 2)])
def test_blocks_cut_while_the_streams_grow_are_read_back_by_the_references_reader(mode, flags, threads):
    """streams of several blocks, cut at 1 MiB as they arrive and coded by the pool while the feeding goes on: other block
    boundaries than the reference's parallelBlocksCompress chooses, the same streams out of its reader; the 5th-byte stream
    is fed and only enrolled by what finish() is told (beyond 2^32)"""
    L = host()
    streams = synthetic_streams(13 + mode, 0.8)
    fed = list(streams)
    fed[2] = streams[2] + b"\xbb"
    if mode == 3:
        import _orc
        fed[6], fed[7], fed[8], _ = _orc.rc_match_sequence(np.frombuffer(streams[6], dtype=np.uint8), 55)
    for ref_total in (1 << 31, (1 << 32) + 9):
        section, early = fed_in_pieces(L, _params(mode, flags, threads=threads), fed, 1 << 20, threads, 300_000, ref_total, linger=1.5)
        enrolled = [st for st in range(NST) if (st not in (7, 8) or mode == 3) and (st != 13 or ref_total > 0xFFFFFFFF)]
        read_back(section, fed, enrolled)
        ref, one_shot, _ = both(L, mode, flags, streams, threads=threads, ref_total=ref_total)
        assert one_shot == ref and section != ref and abs(len(section) - len(ref)) < 0.03 * len(ref)
        assert early > 0                                                       # (blocks were done before the end)

"""ctypes binding of the CPU oracle (oracle/liboracle.so). Test infrastructure only: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
NO_LOCK = 2 ** 64 - 1
SKIPPED = 2 ** 64 - 1
STREAM_NAMES = ("literals", "mapOff", "mapOff5th", "mapLen", "gapDelta", "flags")


class OrcMatch(C.Structure):
    _fields_ = [("posSrcText", C.c_uint64), ("length", C.c_uint64), ("posDestText", C.c_uint64),
                ("nextSrcRegionLoadingPos", C.c_uint64)]


class OrcBuf(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_uint8)), ("size", C.c_uint64), ("cap", C.c_uint64)]


class OrcStreams(C.Structure):
    _fields_ = [("s", OrcBuf * 6), ("unmatchedChars", C.c_uint64), ("extensionsMatchedChars", C.c_uint64),
                ("extensionsMismatches", C.c_uint64), ("totalMatched", C.c_uint64),
                ("totalDestOverlap", C.c_uint64), ("totalDestLen", C.c_uint64),
                ("removedGapBreakingMatches", C.c_uint64)]


class OrcEmitParams(C.Structure):
    _fields_ = [("enableExtensionsWithMismatches", C.c_int), ("mismatchesWithExclusion", C.c_int),
                ("lazyDecompressionSupport", C.c_int), ("enable40bitReference", C.c_int),
                ("frugal64bitLenEncoding", C.c_int), ("gapDepthOffsetEncoding", C.c_int),
                ("gapDepthMismatchesEncoding", C.c_int), ("gapBreakingMatchMinLength", C.c_uint64),
                ("mmsMatchBonus", C.c_int), ("mmsMismatchPenalty", C.c_int),
                ("mmsMismatchesScoreThreshold", C.c_int), ("mmsMismatchesInitialScore", C.c_int),
                ("allowedTargetsOutrunForDissimilarContigs", C.c_int),
                ("minimalLengthForDissimilarContigs", C.c_uint64),
                ("unmatchedFractionFactorTweakForDissimilarContigs", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(ORACLE_DIR, "liboracle.so")
        srcs = [os.path.join(ORACLE_DIR, f) for f in ("swsem_oracle.c", "emit_oracle.c", "oracle.h")]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "oracle"])
        L = C.CDLL(so)
        u64, vp, ci = C.c_uint64, C.c_void_p, C.c_int
        L.orc_hash.restype = C.c_uint32
        L.orc_hash.argtypes = [vp, ci]
        L.orc_matcher_create.restype = vp
        L.orc_matcher_create.argtypes = [u64, ci, ci, ci, ci]
        L.orc_matcher_destroy.argtypes = [vp]
        for n in ("orc_disable_sliding_window", "orc_disable_circular_buffer"):
            getattr(L, n).argtypes = [vp]
        L.orc_set_sliding_window_size.argtypes = [vp, ci]
        L.orc_load_ref.argtypes = [vp, vp, u64, ci, ci, ci]
        L.orc_load_separator.argtypes = [vp, ci]
        for n in ("orc_ref_length", "orc_loading_position", "orc_loaded_ref_length", "orc_max_ref_length",
                  "orc_acquire_lock"):
            getattr(L, n).restype = u64
            getattr(L, n).argtypes = [vp]
        L.orc_set_position.argtypes = [vp, u64, ci]
        L.orc_release_lock.restype = ci
        L.orc_release_lock.argtypes = [vp, u64]
        L.orc_hash_size.restype = C.c_uint32
        L.orc_hash_size.argtypes = [vp]
        L.orc_ht.restype = C.POINTER(C.c_uint32)
        L.orc_ht.argtypes = [vp]
        L.orc_ref.restype = C.POINTER(C.c_uint8)
        L.orc_ref.argtypes = [vp]
        L.orc_K.restype = ci
        L.orc_K.argtypes = [vp]
        L.orc_set_prefilter.argtypes = [vp, ci]
        L.orc_match_texts.restype = u64
        L.orc_match_texts.argtypes = [vp, vp, u64, C.c_uint32, u64, C.POINTER(C.POINTER(OrcMatch)), C.POINTER(u64)]
        L.orc_free.argtypes = [vp]
        L.orc_upper_reverse_complement.argtypes = [vp, u64, vp]
        L.orc_mismatch2code.restype = C.c_uint8
        L.orc_mismatch2code.argtypes = [C.c_uint8, C.c_uint8]
        L.orc_emit_params_default.argtypes = [C.POINTER(OrcEmitParams), ci]
        L.orc_streams_init.argtypes = [C.POINTER(OrcStreams)]
        L.orc_streams_free.argtypes = [C.POINTER(OrcStreams)]
        L.orc_process_matches.restype = u64
        L.orc_process_matches.argtypes = [vp, C.POINTER(OrcEmitParams), C.POINTER(OrcMatch), C.POINTER(u64), vp, u64,
                                          u64, ci, C.c_int64, C.c_int64, vp, u64, C.POINTER(OrcStreams)]
        _lib = L
    return _lib


def _bytes_ptr(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(C.c_void_p)


def orc_hash(b, K=28):
    a, p = _bytes_ptr(np.frombuffer(bytes(b), dtype=np.uint8))
    return lib().orc_hash(p, K)


class OracleMatcher:
    """Mirrors the reference's SlidingWindowExpSparseEMMatcher surface on the C restatement."""

    def __init__(self, max_ref_len, L=32, k1=16, k2=1, skip_margin=16):
        self.h = lib().orc_matcher_create(max_ref_len, L, k1, k2, skip_margin)
        assert self.h, "oracle: bad matcher parameters"
        self.last_stats = None

    def close(self):
        if self.h:
            lib().orc_matcher_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def disable_sliding_window(self): lib().orc_disable_sliding_window(self.h)
    def set_sliding_window_size(self, f): lib().orc_set_sliding_window_size(self.h, f)
    def disable_circular_buffer(self): lib().orc_disable_circular_buffer(self.h)

    def load_ref(self, text, load_rc=False, add_sep=True, sep=0):
        a, p = _bytes_ptr(text)
        lib().orc_load_ref(self.h, p, a.size, int(load_rc), int(add_sep), sep)

    def load_separator(self, sep=0): lib().orc_load_separator(self.h, sep)
    def ref_length(self): return lib().orc_ref_length(self.h)
    def loading_position(self): return lib().orc_loading_position(self.h)
    def loaded_ref_length(self): return lib().orc_loaded_ref_length(self.h)
    def max_ref_length(self): return lib().orc_max_ref_length(self.h)
    def set_position(self, pos, laps): lib().orc_set_position(self.h, pos, laps)
    def acquire_lock(self): return lib().orc_acquire_lock(self.h)
    def release_lock(self, v): return lib().orc_release_lock(self.h, v)
    def hash_size(self): return lib().orc_hash_size(self.h)
    def K(self): return lib().orc_K(self.h)
    def set_prefilter(self, on): lib().orc_set_prefilter(self.h, int(on))

    def ht(self):
        n = self.hash_size()
        return np.ctypeslib.as_array(lib().orc_ht(self.h), shape=(n,)).copy()

    def ref(self, n=None):
        n = self.ref_length() if n is None else n
        return np.ctypeslib.as_array(lib().orc_ref(self.h), shape=(n,)).copy()

    def match(self, q, min_len=32, lock=NO_LOCK):
        """-> (n,3) uint64 array of (posSrcText, length, posDestText)."""
        a, p = _bytes_ptr(q)
        out = C.POINTER(OrcMatch)()
        stats = (C.c_uint64 * 3)()
        n = lib().orc_match_texts(self.h, p, a.size, min_len, lock, C.byref(out), stats)
        self.last_stats = tuple(stats)
        res = np.zeros((n, 3), dtype=np.uint64)
        if n:
            raw = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint64)), shape=(n, 4))
            res[:] = raw[:, :3]
        lib().orc_free(out)
        return res


def emit_params(mode=1, **over):
    p = OrcEmitParams()
    lib().orc_emit_params_default(C.byref(p), mode)
    for k, v in over.items():
        setattr(p, k, v)
    return p


class OracleEmitter:
    """Per-target stream emission (MBGC_Encoder::processMatches) on the C restatement."""

    def __init__(self, matcher, params=None):
        self.m = matcher
        self.p = params if params is not None else emit_params(1)
        self.s = OrcStreams()
        lib().orc_streams_init(C.byref(self.s))

    def __del__(self):
        lib().orc_streams_free(C.byref(self.s))

    def process(self, matches, dest, lock=NO_LOCK, factor=128, processed=0, target_idx=0, loaded=None):
        a, p = _bytes_ptr(dest)
        n = len(matches)
        rows = np.zeros((max(n, 1), 4), dtype=np.uint64)                 # OrcMatch rows (the fourth field is filled by pass 1)
        if n:
            rows[:n, :3] = np.asarray(matches, dtype=np.uint64).reshape(n, 3)
        arr = C.cast(rows.ctypes.data_as(C.c_void_p), C.POINTER(OrcMatch))
        nn = C.c_uint64(n)
        ld = np.ascontiguousarray(loaded if loaded is not None else [0], dtype=np.uint64)
        r = lib().orc_process_matches(self.m.h, C.byref(self.p), arr, C.byref(nn), p, a.size, lock, factor,
                                      processed, target_idx, ld.ctypes.data_as(C.c_void_p), ld.size,
                                      C.byref(self.s))
        return r

    def put(self, which, data):
        b = bytes(data)
        lib().orc_buf_put.argtypes = [C.POINTER(OrcBuf), C.c_char_p, C.c_uint64]
        lib().orc_buf_put(C.byref(self.s.s[which]), b, len(b))

    def stream(self, which):
        b = self.s.s[which]
        return bytes(np.ctypeslib.as_array(b.data, shape=(b.size,))) if b.size else b""

    def streams(self):
        return {STREAM_NAMES[i]: self.stream(i) for i in range(6)}

    def counters(self):
        s = self.s
        return dict(unmatchedChars=s.unmatchedChars, extensionsMatchedChars=s.extensionsMatchedChars,
                    extensionsMismatches=s.extensionsMismatches, totalMatched=s.totalMatched,
                    totalDestLen=s.totalDestLen, removedGapBreakingMatches=s.removedGapBreakingMatches)


def decode_contig(ref, params, streams, lock=NO_LOCK, cap=None):
    """MBGC_Decoder::decodeSequenceAndReturnUnmatchedChars (oracle/decode_oracle.c) for the six streams of one contig
    (dict by STREAM_NAMES, bytes or uint8 arrays) against the reference bytes `ref` (a uint8 array, or a ctypes
    pointer to the matcher's buffer). Returns (contig bytes as uint8 array, unmatchedChars); raises on a malformed
    stream set."""
    arrs = [np.ascontiguousarray(np.frombuffer(streams[n], dtype=np.uint8) if isinstance(streams[n], (bytes, bytearray))
                                 else streams[n], dtype=np.uint8) for n in STREAM_NAMES]
    ptrs = (C.c_void_p * 6)(*[a.ctypes.data if a.size else None for a in arrs])
    sizes = (C.c_uint64 * 6)(*[a.size for a in arrs])
    if cap is None:
        cap = 1 << 28                                  # (untouched pages cost nothing)
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_uint64(0)
    refp = ref.ctypes.data_as(C.c_void_p) if isinstance(ref, np.ndarray) else ref
    f = lib().orc_decode_contig
    f.restype = C.c_int64
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p]
    r = f(refp, C.byref(params), ptrs, sizes, lock, out.ctypes.data_as(C.c_void_p), cap, C.byref(n))
    if r < 0:
        raise ValueError("orc_decode_contig: malformed streams (consumed into %d output bytes)" % n.value)
    return out[: n.value], int(r)


def fingerprint(matches):
    """FNV-style fingerprint of a match list, SURVEY.md §8c."""
    fp = 0xcbf29ce484222325
    for row in np.asarray(matches, dtype=np.uint64).reshape(-1, 3):
        for v in row:
            fp ^= int(v)
            fp = (fp * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return fp


# ---- the -m3 reverse-complement pass over the literal stream (oracle/rcmatch_oracle.c)
def rc_find_matches(seq, target=55, min_len=0xFFFFFFFF):
    """matches CopMEMMatcher::matchTexts pushes for the reverse-complemented sequence against itself, in push order:
    (n, 3) uint64 rows (posSrcText, length, posDestText in the reverse-complemented text) and (K, k1, k2, log2 hash size)"""
    a, p = _bytes_ptr(seq)
    L = lib()
    L.orc_rc_find_matches.restype = C.c_uint64
    L.orc_rc_find_matches.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.POINTER(OrcMatch)), C.POINTER(C.c_int), C.POINTER(C.c_uint64)]
    out = C.POINTER(OrcMatch)()
    params = (C.c_int * 4)()
    ext = C.c_uint64()
    n = L.orc_rc_find_matches(p, a.size, target, min_len, C.byref(out), params, C.byref(ext))
    assert n != 2 ** 64 - 1, "the reference exits on these parameters"
    res = np.zeros((n, 3), dtype=np.uint64)
    if n:
        res[:] = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint64)), shape=(n, 4))[:, :3]
    L.orc_free(out)
    return res, tuple(params), ext.value


def rc_match_sequence(seq, target=55, min_len=0xFFFFFFFF):
    """SimpleSequenceMatcher::rcMatchSequence: -> (rewritten sequence, rcMapOff, rcMapLen, (unique matches, matched, overlapped))"""
    a = np.array(seq, dtype=np.uint8, copy=True)
    L = lib()
    L.orc_rc_match_sequence.restype = C.c_uint64
    L.orc_rc_match_sequence.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(OrcBuf), C.POINTER(OrcBuf), C.POINTER(C.c_uint64)]
    off, ln = OrcBuf(), OrcBuf()
    st = (C.c_uint64 * 3)()
    n = L.orc_rc_match_sequence(a.ctypes.data_as(C.c_void_p), a.size, target, min_len, C.byref(off), C.byref(ln), st)
    assert n != 2 ** 64 - 1
    take = lambda b: bytes(np.ctypeslib.as_array(b.data, shape=(b.size,))) if b.size else b""
    res = (a[:n].tobytes(), take(off), take(ln), tuple(st))
    L.orc_free(off.data), L.orc_free(ln.data)
    return res


def rc_apply_matches(seq, matches, target=55, min_len=0xFFFFFFFF):
    """the post-processing half alone (rows as rc_find_matches returns them)"""
    a = np.array(seq, dtype=np.uint8, copy=True)
    L = lib()
    L.orc_rc_apply_matches.restype = C.c_uint64
    L.orc_rc_apply_matches.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(OrcBuf), C.POINTER(OrcBuf), C.POINTER(C.c_uint64)]
    n = len(matches)
    rows = np.zeros((max(n, 1), 4), dtype=np.uint64)
    if n:
        rows[:n, :3] = np.asarray(matches, dtype=np.uint64).reshape(n, 3)
    off, ln = OrcBuf(), OrcBuf()
    st = (C.c_uint64 * 3)()
    r = L.orc_rc_apply_matches(a.ctypes.data_as(C.c_void_p), a.size, rows.ctypes.data_as(C.c_void_p), n, target, min_len, C.byref(off), C.byref(ln), st)
    take = lambda b: bytes(np.ctypeslib.as_array(b.data, shape=(b.size,))) if b.size else b""
    res = (a[:r].tobytes(), take(off), take(ln), tuple(st))
    L.orc_free(off.data), L.orc_free(ln.data)
    return res

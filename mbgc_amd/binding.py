"""ctypes view of the C ABI in include/mbgc_swsem.h (libmbgc_hip.so, built by __graft_entry__.build()).

This is the Python-side mirror of the reference's SlidingWindowSparseEMMatcher surface
(matching/SlidingWindowSparseEMMatcher.h:88-124): same method names and argument meaning; errors the
reference reports with a message + exit(EXIT_FAILURE) surface here as SwsemError with that message.
There is no CPU fallback: a missing library or device raises."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MBGC_HIP_LIB", os.path.join(HERE, "libmbgc_hip.so"))   # override: A/B builds of the kernels
NO_LOCK = 2 ** 64 - 1
SKIPPED = 2 ** 64 - 1
STREAM_NAMES = ("literals", "mapOff", "mapOff5th", "mapLen", "gapDelta", "flags")
KERNEL_FAMILIES = ("load", "insert", "probe", "emit2", "resolve", "stitch", "emit")

EXPORTS = """swsem_last_error swsem_device_count swsem_device_numa_node swsem_create swsem_destroy swsem_set_stream swsem_synchronize
swsem_disable_sliding_window swsem_set_sliding_window_size swsem_disable_circular_buffer swsem_get_ref_length
swsem_get_loading_position swsem_get_loaded_ref_length swsem_get_max_ref_length swsem_get_sliding_window_size swsem_get_dropped_bytes swsem_set_position
swsem_acquire_lock swsem_release_lock swsem_get_K swsem_get_hash_size swsem_load_ref swsem_load_ref_dev
swsem_load_separator swsem_finalize_targets swsem_revcomp_dev swsem_match swsem_match_batch_dev swsem_batch_counts swsem_batch_matches
swsem_batch_fingerprint swsem_emit_params_default swsem_emit swsem_emit_batch swsem_emit_batch_begin swsem_emit_batch_begin_spec swsem_emit_batch_end swsem_emit_select swsem_emit_result swsem_emit_set_host_copy swsem_emit_unmatched swsem_emit_pack_dev swsem_emit_pack_dev_on swsem_emit_counters swsem_debug_copy_ref swsem_debug_write_ref swsem_debug_copy_ht swsem_debug_emit_stats
swsem_profile_enable swsem_profile_get swsem_batch_stats swsem_dev_malloc swsem_dev_free swsem_dev_upload swsem_dev_download swsem_dev_copy swsem_decode_contigs_dev swsem_emit_verify""".split()


class SpecFinalize(C.Structure):
    _fields_ = [("ntargets", C.c_int), ("ext_dev", C.c_void_p), ("ext_len", C.POINTER(C.c_uint64)), ("addSep", C.c_int), ("sep", C.c_int),
                ("lazySeparator", C.c_int), ("lockPos", C.POINTER(C.c_uint64)), ("loadedAfter", C.POINTER(C.c_uint64)),
                ("predExt", C.c_void_p), ("predRC", C.c_void_p), ("factor", C.c_int), ("rcFactor", C.c_int),
                ("gate_dev", C.c_void_p), ("exchange", C.c_void_p), ("exchange_ctx", C.c_void_p), ("veto", C.c_int)]


SPEC_EXCHANGE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p)


class SwsemError(RuntimeError):
    pass


class EmitParams(C.Structure):
    _fields_ = [("enableExtensionsWithMismatches", C.c_int), ("mismatchesWithExclusion", C.c_int),
                ("lazyDecompressionSupport", C.c_int), ("enable40bitReference", C.c_int),
                ("frugal64bitLenEncoding", C.c_int), ("gapDepthOffsetEncoding", C.c_int),
                ("gapDepthMismatchesEncoding", C.c_int), ("gapBreakingMatchMinLength", C.c_uint64),
                ("mmsMatchBonus", C.c_int), ("mmsMismatchPenalty", C.c_int),
                ("mmsMismatchesScoreThreshold", C.c_int), ("mmsMismatchesInitialScore", C.c_int),
                ("allowedTargetsOutrunForDissimilarContigs", C.c_int),
                ("minimalLengthForDissimilarContigs", C.c_uint64),
                ("unmatchedFractionFactorTweakForDissimilarContigs", C.c_int)]


class DecodeJob(C.Structure):
    _fields_ = [("stream_dev", C.c_void_p * 6), ("size", C.c_uint64 * 6), ("refLockPos", C.c_uint64), ("dest_dev", C.c_void_p),
                ("destCap", C.c_uint64)]


class Streams(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_uint8) * 6), ("size", C.c_uint64 * 6), ("unmatchedChars", C.c_uint64),
                ("extensionsMatchedChars", C.c_uint64), ("extensionsMismatches", C.c_uint64),
                ("totalMatched", C.c_uint64), ("removedGapBreakingMatches", C.c_uint64), ("nmatches", C.c_uint64)]


_lib = None


def lib():
    """The loaded library; raises if the HIP extension has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SwsemError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
        try:
            # torch bundles its own libamdhip64.so.7; loading it first makes this library bind to the
            # same HIP runtime instance (two runtimes in one process cannot share the GPU)
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        u64, vp, ci, pu64 = C.c_uint64, C.c_void_p, C.c_int, C.POINTER(C.c_uint64)
        L.swsem_last_error.restype = C.c_char_p
        L.swsem_create.argtypes = [C.POINTER(vp), u64, ci, ci, ci, ci, ci]
        L.swsem_destroy.argtypes = [vp]
        L.swsem_set_stream.argtypes = [vp, vp]
        L.swsem_synchronize.argtypes = [vp]
        for n in ("swsem_disable_sliding_window", "swsem_disable_circular_buffer"):
            getattr(L, n).argtypes = [vp]
            getattr(L, n).restype = None
        L.swsem_set_sliding_window_size.argtypes = [vp, ci]
        L.swsem_set_sliding_window_size.restype = None
        for n in ("swsem_get_ref_length", "swsem_get_loading_position", "swsem_get_loaded_ref_length",
                  "swsem_get_max_ref_length", "swsem_get_sliding_window_size", "swsem_get_dropped_bytes", "swsem_acquire_lock"):
            getattr(L, n).restype = u64
            getattr(L, n).argtypes = [vp]
        L.swsem_set_position.argtypes = [vp, u64, ci]
        L.swsem_set_position.restype = None
        L.swsem_release_lock.argtypes = [vp, u64]
        L.swsem_get_K.argtypes = [vp]
        L.swsem_get_hash_size.restype = C.c_uint32
        L.swsem_get_hash_size.argtypes = [vp]
        L.swsem_load_ref.argtypes = [vp, vp, u64, ci, ci, ci]
        L.swsem_load_ref_dev.argtypes = [vp, vp, u64, ci, ci, ci]
        L.swsem_load_separator.argtypes = [vp, ci]
        L.swsem_revcomp_dev.argtypes = [vp, vp, u64, vp]
        L.swsem_finalize_targets.argtypes = [vp, ci, vp, pu64, ci, ci, ci, pu64, pu64]
        L.swsem_match.argtypes = [vp, vp, u64, C.c_uint32, u64, C.POINTER(vp), pu64]
        L.swsem_match_batch_dev.argtypes = [vp, vp, pu64, ci, C.c_uint32, pu64]
        L.swsem_batch_counts.argtypes = [vp, pu64]
        L.swsem_batch_matches.argtypes = [vp, ci, vp, u64]
        L.swsem_batch_fingerprint.argtypes = [vp, pu64, pu64, pu64]
        L.swsem_emit_params_default.argtypes = [C.POINTER(EmitParams), ci]
        L.swsem_emit_params_default.restype = None
        L.swsem_emit.argtypes = [vp, C.POINTER(EmitParams), ci, u64, ci, C.c_int64, C.c_int64, vp, u64,
                                 C.POINTER(Streams)]
        L.swsem_emit_batch.argtypes = [vp, C.POINTER(EmitParams), ci, vp, vp, vp, vp, vp, vp, u64]
        L.swsem_emit_batch_begin.argtypes = L.swsem_emit_batch.argtypes
        L.swsem_emit_batch_end.argtypes = [vp]
        L.swsem_emit_batch_begin_spec.argtypes = L.swsem_emit_batch.argtypes + [C.POINTER(SpecFinalize), C.POINTER(C.c_int)]
        L.swsem_emit_select.argtypes = [vp, ci]
        L.swsem_emit_result.argtypes = [vp, ci, C.POINTER(Streams)]
        L.swsem_emit_set_host_copy.argtypes = [vp, ci]
        L.swsem_emit_set_host_copy.restype = None
        L.swsem_emit_pack_dev.argtypes = [vp, vp, u64, pu64, pu64]
        L.swsem_emit_pack_dev_on.argtypes = [vp, vp, u64, vp]
        L.swsem_emit_counters.argtypes = [vp, pu64]
        L.swsem_emit_unmatched.argtypes = [vp, pu64]
        L.swsem_emit_verify.argtypes = [vp, C.POINTER(ci), C.POINTER(ci), pu64]
        L.swsem_decode_contigs_dev.argtypes = [vp, C.POINTER(EmitParams), ci, vp, pu64, C.POINTER(C.c_int64)]
        L.swsem_debug_copy_ref.argtypes = [vp, u64, u64, vp]
        L.swsem_debug_write_ref.argtypes = [vp, u64, u64, vp]
        L.swsem_debug_copy_ht.argtypes = [vp, vp]
        L.swsem_profile_enable.argtypes = [vp, ci]
        L.swsem_profile_get.argtypes = [vp, C.POINTER(C.c_double), pu64]
        L.swsem_batch_stats.argtypes = [vp, pu64]
        _lib = L
    return _lib


def _chk(rc):
    if rc != 0:
        raise SwsemError("swsem error %d: %s" % (rc, lib().swsem_last_error().decode()))


def _bytes_ptr(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(C.c_void_p)


def emit_params(mode=1, **over):
    p = EmitParams()
    lib().swsem_emit_params_default(C.byref(p), mode)
    for k, v in over.items():
        setattr(p, k, v)
    return p


class SlidingWindowSparseEMMatcher:
    """HIP-backed matcher with the reference class's surface (Exp variant: even k1, k2 = 1)."""

    def __init__(self, max_ref_len, L=32, k1=16, k2=1, skip_margin=16, device=0):
        self.h = C.c_void_p()
        _chk(lib().swsem_create(C.byref(self.h), max_ref_len, L, k1, k2, skip_margin, device))

    def close(self):
        if getattr(self, "h", None):
            lib().swsem_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    # --- window / position API (host-side state, SlidingWindowSparseEMMatcher.h:93-116)
    def disable_sliding_window(self): lib().swsem_disable_sliding_window(self.h)
    def set_sliding_window_size(self, f): lib().swsem_set_sliding_window_size(self.h, f)
    def disable_circular_buffer(self): lib().swsem_disable_circular_buffer(self.h)
    def ref_length(self): return lib().swsem_get_ref_length(self.h)
    def loading_position(self): return lib().swsem_get_loading_position(self.h)
    def loaded_ref_length(self): return lib().swsem_get_loaded_ref_length(self.h)
    def max_ref_length(self): return lib().swsem_get_max_ref_length(self.h)
    def sliding_window_size(self): return lib().swsem_get_sliding_window_size(self.h)
    def dropped_bytes(self): return lib().swsem_get_dropped_bytes(self.h)

    def emit_stats(self):
        """Counters of the emission's pairing chain since the handle was made (diagnostics)."""
        out = (C.c_uint64 * 8)()
        lib().swsem_debug_emit_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        _chk(lib().swsem_debug_emit_stats(self.h, out))
        return {"foreign_boundary_steps": int(out[4]), "blocks_not_accepted": int(out[5]), "groups_replayed": int(out[6]),
                "blocks_given_up": int(out[7])}
    def set_position(self, pos, laps): lib().swsem_set_position(self.h, pos, laps)
    def acquire_lock(self): return lib().swsem_acquire_lock(self.h)
    def release_lock(self, v): _chk(lib().swsem_release_lock(self.h, v)); return 0
    def hash_size(self): return lib().swsem_get_hash_size(self.h)
    def K(self): return lib().swsem_get_K(self.h)
    def set_stream(self, stream_ptr): _chk(lib().swsem_set_stream(self.h, stream_ptr))
    def synchronize(self): _chk(lib().swsem_synchronize(self.h))

    # --- reference extension
    def load_ref(self, text, load_rc=False, add_sep=True, sep=0):
        a, p = _bytes_ptr(text)
        _chk(lib().swsem_load_ref(self.h, p, a.size, int(load_rc), int(add_sep), sep))

    def load_ref_dev(self, dev_ptr, n, load_rc=False, add_sep=True, sep=0):
        _chk(lib().swsem_load_ref_dev(self.h, dev_ptr, n, int(load_rc), int(add_sep), sep))

    def load_separator(self, sep=0): _chk(lib().swsem_load_separator(self.h, sep))

    def finalize_targets(self, ext_ptrs, ext_lens, locks, lazy=True, add_sep=True, sep=0):
        """loadRef(ext) [+ loadSeparator] + releaseWorkerMatchingLockPos for n targets in order; returns
        getLoadedRefLength() after each target."""
        n = len(ext_lens)
        ptrs = (C.c_void_p * n)(*[int(p) for p in ext_ptrs])
        lens = np.ascontiguousarray(ext_lens, dtype=np.uint64)
        lk = np.ascontiguousarray(locks, dtype=np.uint64)
        out = np.zeros(n, dtype=np.uint64)
        P = C.POINTER(C.c_uint64)
        _chk(lib().swsem_finalize_targets(self.h, n, ptrs, lens.ctypes.data_as(P), int(add_sep), sep, int(lazy),
                                          lk.ctypes.data_as(P), out.ctypes.data_as(P)))
        return out
    def revcomp_dev(self, src_ptr, n, dst_ptr): _chk(lib().swsem_revcomp_dev(self.h, src_ptr, n, dst_ptr))

    # --- matching
    def match(self, q, min_len=32, lock=NO_LOCK):
        """matchTexts: -> (n,3) uint64 rows (posSrcText, length, posDestText)."""
        a, p = _bytes_ptr(q)
        out, n = C.c_void_p(), C.c_uint64()
        _chk(lib().swsem_match(self.h, p, a.size, min_len, lock, C.byref(out), C.byref(n)))
        if n.value == 0:
            return np.zeros((0, 3), dtype=np.uint64)
        return np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint64)), shape=(n.value, 3)).copy()

    def match_batch_dev(self, dev_ptr, offsets, min_len=32, locks=None):
        offs = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = offs.size - 1
        lk = None
        if locks is not None:
            lk = np.ascontiguousarray(locks, dtype=np.uint64)
            assert lk.size == n
        _chk(lib().swsem_match_batch_dev(self.h, dev_ptr, offs.ctypes.data_as(C.POINTER(C.c_uint64)), n, min_len,
                                         lk.ctypes.data_as(C.POINTER(C.c_uint64)) if lk is not None else None))
        self._batch_n = n

    def batch_counts(self):
        out = np.zeros(self._batch_n, dtype=np.uint64)
        _chk(lib().swsem_batch_counts(self.h, out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return out

    def batch_matches(self, contig, count):
        out = np.zeros((int(count), 3), dtype=np.uint64)
        _chk(lib().swsem_batch_matches(self.h, contig, out.ctypes.data_as(C.c_void_p), int(count)))
        return out

    def batch_fingerprint(self):
        fp, tot, ln = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _chk(lib().swsem_batch_fingerprint(self.h, C.byref(fp), C.byref(tot), C.byref(ln)))
        return fp.value, tot.value, ln.value

    def batch_stats(self):
        s = (C.c_uint64 * 6)()
        _chk(lib().swsem_batch_stats(self.h, s))
        return dict(zip(("bases", "probes", "hits", "matches", "matched_len", "replayed_blocks"), s))

    # --- emission (MBGC_Encoder::processMatches)
    def emit(self, params, contig=0, lock=NO_LOCK, factor=128, processed=0, target_idx=0, loaded=None):
        ld = np.ascontiguousarray(loaded if loaded is not None else [0], dtype=np.uint64)
        st = Streams()
        _chk(lib().swsem_emit(self.h, C.byref(params), contig, lock, factor, processed, target_idx,
                              ld.ctypes.data_as(C.c_void_p), ld.size, C.byref(st)))
        streams = {}
        for i, name in enumerate(STREAM_NAMES):
            n = st.size[i]
            streams[name] = bytes(np.ctypeslib.as_array(st.data[i], shape=(n,))) if n else b""
        return st.unmatchedChars, streams, st

    def emit_batch(self, params, contigs=None, locks=None, factors=None, processed=None, target_idx=None, loaded=None, n=None,
                   _entry="swsem_emit_batch"):
        """processMatches for several contigs of the last batch at once; fetch with emit_result(k)."""
        def arr(x, dt):
            return None if x is None else np.ascontiguousarray(x, dtype=dt)
        ci, lk, fa = arr(contigs, np.int32), arr(locks, np.uint64), arr(factors, np.int32)
        pr, ti = arr(processed, np.int64), arr(target_idx, np.int64)
        ld = np.ascontiguousarray(loaded if loaded is not None else [0], dtype=np.uint64)
        cnt = n if n is not None else (ci.size if ci is not None else self._batch_n)
        ptr = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        _chk(getattr(lib(), _entry)(self.h, C.byref(params), cnt, ptr(ci), ptr(lk), ptr(fa), ptr(pr), ptr(ti), ptr(ld), ld.size))
        return cnt

    def emit_batch_begin(self, *a, **k):
        """emit_batch up to processMatches' return values (emit_unmatched); the stream bytes follow on a second
        HIP stream while the caller finalizes the round and starts the next one. emit_batch_end() waits."""
        return self.emit_batch(*a, _entry="swsem_emit_batch_begin", **k)

    def emit_batch_end(self): _chk(lib().swsem_emit_batch_end(self.h))

    def emit_batch_begin_spec(self, params, locks, factors, processed, target_idx, loaded, n, ext_ptrs, ext_lens, target_locks,
                              pred_ext, pred_rc, factor, rc_factor, lazy=True, add_sep=True, sep=0, gate=0, reduce=None,
                              verdict=None, veto=False):
        """emit_batch_begin plus a speculative finalize of the round's targets (see include/mbgc_swsem.h). Returns
        (applied, loaded_after): applied False means nothing was done and finalize_targets is still to be called.
        Several replicas: gate = device address of the int32 word the check writes, reduce(stream) queues its reduction
        (minimum over the replicas) on the handle's stream, verdict() returns the reduced word, veto = this replica's
        word is 0 whatever its pass 1 finds."""
        u64, P = np.uint64, C.POINTER(C.c_uint64)
        lk, fa = np.ascontiguousarray(locks, dtype=u64), np.ascontiguousarray(factors, dtype=np.int32)
        pr, ti = np.ascontiguousarray(processed, dtype=np.int64), np.ascontiguousarray(target_idx, dtype=np.int64)
        ld = np.ascontiguousarray(loaded if loaded is not None else [0], dtype=u64)
        nt = len(ext_lens)
        ptrs = (C.c_void_p * nt)(*[int(x) for x in ext_ptrs])
        lens = np.ascontiguousarray(ext_lens, dtype=u64)
        tl = np.ascontiguousarray(target_locks, dtype=u64)
        after = np.zeros(nt, dtype=u64)
        pe, prc = np.ascontiguousarray(pred_ext, dtype=np.uint8), np.ascontiguousarray(pred_rc, dtype=np.uint8)
        sp = SpecFinalize(nt, C.cast(ptrs, C.c_void_p), lens.ctypes.data_as(P), int(add_sep), sep, int(lazy), tl.ctypes.data_as(P),
                          after.ctypes.data_as(P), pe.ctypes.data_as(C.c_void_p), prc.ctypes.data_as(C.c_void_p), int(factor), int(rc_factor))
        failure = []
        if reduce is not None:
            def _cb(ctx, phase, gate_dev, stream):
                try:
                    if phase == 0:
                        reduce(stream)
                        return 0
                    return int(verdict())
                except BaseException as e:           # (an exception cannot cross the C frames)
                    failure.append(e)
                    return -1
            cb = SPEC_EXCHANGE(_cb)
            sp.gate_dev, sp.exchange, sp.veto = int(gate), C.cast(cb, C.c_void_p), int(veto)
        applied = C.c_int()
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        rc = lib().swsem_emit_batch_begin_spec(self.h, C.byref(params), n, None, vp(lk), vp(fa), vp(pr), vp(ti), vp(ld), ld.size,
                                               C.byref(sp), C.byref(applied))
        if failure:
            raise failure[0]
        _chk(rc)
        return bool(applied.value), after
    def emit_select(self, previous): _chk(lib().swsem_emit_select(self.h, int(previous)))

    def emit_set_host_copy(self, on): lib().swsem_emit_set_host_copy(self.h, int(on))

    def emit_verify(self):
        """device-side check of the selected emission: its streams decoded by the decoder's automaton against the query.
        -> (contigs that fail, first of them or -1, first differing byte)"""
        nbad, first, diff = C.c_int(), C.c_int(), C.c_uint64()
        _chk(lib().swsem_emit_verify(self.h, C.byref(nbad), C.byref(first), C.byref(diff)))
        return nbad.value, first.value, diff.value

    def decode_contigs_dev(self, params, jobs):
        """jobs: list of (six (device pointer, size) pairs, lock position, dest device pointer, dest capacity).
        -> (bytes written per contig, unmatchedChars per contig or -1)"""
        n = len(jobs)
        arr = (DecodeJob * n)()
        for k, (streams, lock, dest, cap) in enumerate(jobs):
            for st, (ptr, size) in enumerate(streams):
                arr[k].stream_dev[st] = ptr
                arr[k].size[st] = size
            arr[k].refLockPos, arr[k].dest_dev, arr[k].destCap = lock, dest, cap
        dl = np.zeros(n, dtype=np.uint64)
        un = np.zeros(n, dtype=np.int64)
        _chk(lib().swsem_decode_contigs_dev(self.h, C.byref(params), n, C.cast(arr, C.c_void_p), dl.ctypes.data_as(C.POINTER(C.c_uint64)),
                                            un.ctypes.data_as(C.POINTER(C.c_int64))))
        return dl, un

    def emit_pack_sizes(self, n):
        sizes = np.zeros(n * 6, dtype=np.uint64)
        tot = C.c_uint64()
        _chk(lib().swsem_emit_pack_dev(self.h, None, 0, sizes.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(tot)))
        return sizes.reshape(n, 6), tot.value

    def emit_pack_dev(self, dst_ptr, cap, stream=None):
        """stream: queue the copy there (0 = the handle's main stream) instead of waiting for it"""
        if stream is not None:
            _chk(lib().swsem_emit_pack_dev_on(self.h, dst_ptr, cap, C.c_void_p(stream)))
            return None
        tot = C.c_uint64()
        _chk(lib().swsem_emit_pack_dev(self.h, dst_ptr, cap, None, C.byref(tot)))
        return tot.value

    def emit_unmatched(self, n):
        out = np.zeros(n, dtype=np.uint64)
        _chk(lib().swsem_emit_unmatched(self.h, out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return out

    def emit_result(self, k):
        st = Streams()
        _chk(lib().swsem_emit_result(self.h, k, C.byref(st)))
        streams = {}
        for i, name in enumerate(STREAM_NAMES):
            sz = st.size[i]
            streams[name] = bytes(np.ctypeslib.as_array(st.data[i], shape=(sz,))) if sz else b""
        return st.unmatchedChars, streams, st

    # --- test / measurement hooks
    def ref(self, n=None, start=0):
        n = self.ref_length() if n is None else n
        out = np.zeros(n, dtype=np.uint8)
        _chk(lib().swsem_debug_copy_ref(self.h, start, n, out.ctypes.data_as(C.c_void_p)))
        return out

    def write_ref(self, start, data):
        a = np.ascontiguousarray(data, dtype=np.uint8)
        _chk(lib().swsem_debug_write_ref(self.h, start, a.size, a.ctypes.data_as(C.c_void_p)))

    def ht(self):
        out = np.zeros(self.hash_size(), dtype=np.uint32)
        _chk(lib().swsem_debug_copy_ht(self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def profile_enable(self, on=True): _chk(lib().swsem_profile_enable(self.h, int(on)))

    def profile_get(self):
        ms = (C.c_double * 7)()
        n = (C.c_uint64 * 7)()
        _chk(lib().swsem_profile_get(self.h, ms, n))
        return {KERNEL_FAMILIES[i]: (ms[i], n[i]) for i in range(7)}

"""ctypes binding of oracle/_ref/libswsem_ref.so — the *reference itself* compiled from
/root/reference (oracle/Makefile, oracle/ref_harness.cpp). Test infrastructure only. Present in the
build container (and shipped prebuilt to the GPU box); tests that need it skip when it is absent."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libswsem_ref.so")
REF_MBGC = os.path.join(ROOT, "oracle", "_ref", "mbgc")
REF_MBGC_DEV = os.path.join(ROOT, "oracle", "_ref", "mbgc-dev")
NO_LOCK = 2 ** 64 - 1

_lib = None


def available():
    return os.path.exists(REF_SO)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(REF_SO)
        u64, vp, ci = C.c_uint64, C.c_void_p, C.c_int
        L.refm_create.restype = vp
        L.refm_create.argtypes = [u64, ci, ci, ci, ci]
        L.refm_destroy.argtypes = [vp]
        L.refm_disable_sliding_window.argtypes = [vp]
        L.refm_disable_circular_buffer.argtypes = [vp]
        L.refm_set_sliding_window_size.argtypes = [vp, ci]
        L.refm_load_ref.argtypes = [vp, vp, u64, ci, ci, ci]
        L.refm_load_separator.argtypes = [vp, ci]
        for n in ("refm_ref_length", "refm_loading_position", "refm_loaded_ref_length", "refm_max_ref_length",
                  "refm_acquire_lock"):
            getattr(L, n).restype = u64
            getattr(L, n).argtypes = [vp]
        L.refm_release_lock.argtypes = [vp, u64]
        L.refm_set_position.argtypes = [vp, u64, ci]
        L.refm_hash_size.restype = C.c_uint32
        L.refm_hash_size.argtypes = [vp]
        L.refm_ht.restype = C.POINTER(C.c_uint32)
        L.refm_ht.argtypes = [vp]
        L.refm_ref.restype = C.POINTER(C.c_uint8)
        L.refm_ref.argtypes = [vp]
        L.refm_K.restype = ci
        L.refm_K.argtypes = [vp]
        L.refm_match.restype = u64
        L.refm_match.argtypes = [vp, vp, u64, C.c_uint32, u64, vp, u64]
        L.refe_create.restype = vp
        L.refe_create.argtypes = [vp, ci, ci, ci, ci]
        L.refe_destroy.argtypes = [vp]
        L.refe_set_processed_targets.argtypes = [vp, C.c_int64]
        L.refe_push_loaded_pos.argtypes = [vp, u64]
        L.refe_process_matches.restype = u64
        L.refe_process_matches.argtypes = [vp, vp, u64, vp, u64, ci, u64]
        L.refe_after_sequence.argtypes = [vp, ci]
        L.refe_after_target.argtypes = [vp, ci]
        L.refe_reset_target.argtypes = [vp, ci]
        L.refe_stream.restype = u64
        L.refe_stream.argtypes = [vp, ci, ci, vp, u64]
        _lib = L
    return _lib


def _bytes_ptr(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(C.c_void_p)


class RefMatcher:
    def __init__(self, max_ref_len, L=32, k1=16, k2=1, skip_margin=16):
        self.h = lib().refm_create(max_ref_len, L, k1, k2, skip_margin)

    def close(self):
        if self.h:
            lib().refm_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def disable_sliding_window(self): lib().refm_disable_sliding_window(self.h)
    def set_sliding_window_size(self, f): lib().refm_set_sliding_window_size(self.h, f)
    def disable_circular_buffer(self): lib().refm_disable_circular_buffer(self.h)

    def load_ref(self, text, load_rc=False, add_sep=True, sep=0):
        a, p = _bytes_ptr(text)
        lib().refm_load_ref(self.h, p, a.size, int(load_rc), int(add_sep), sep)

    def load_separator(self, sep=0): lib().refm_load_separator(self.h, sep)
    def ref_length(self): return lib().refm_ref_length(self.h)
    def loading_position(self): return lib().refm_loading_position(self.h)
    def loaded_ref_length(self): return lib().refm_loaded_ref_length(self.h)
    def max_ref_length(self): return lib().refm_max_ref_length(self.h)
    def set_position(self, pos, laps): lib().refm_set_position(self.h, pos, laps)
    def acquire_lock(self): return lib().refm_acquire_lock(self.h)
    def release_lock(self, v): lib().refm_release_lock(self.h, v)
    def hash_size(self): return lib().refm_hash_size(self.h)
    def K(self): return lib().refm_K(self.h)

    def ht(self):
        return np.ctypeslib.as_array(lib().refm_ht(self.h), shape=(self.hash_size(),)).copy()

    def ref(self, n=None):
        n = self.ref_length() if n is None else n
        return np.ctypeslib.as_array(lib().refm_ref(self.h), shape=(n,)).copy()

    def match(self, q, min_len=32, lock=NO_LOCK):
        a, p = _bytes_ptr(q)
        cap = max(1024, a.size // 16)
        while True:
            out = np.zeros((cap, 3), dtype=np.uint64)
            n = lib().refm_match(self.h, p, a.size, min_len, lock, out.ctypes.data_as(C.c_void_p), cap)
            if n <= cap:
                return out[:n].copy()
            cap = n


class RefEmitter:
    def __init__(self, matcher, mode=1, lazy=True, bit40=False, n_targets=1):
        self.m = matcher
        self.h = lib().refe_create(matcher.h, mode, int(lazy), int(bit40), n_targets)

    def __del__(self):
        if self.h:
            lib().refe_destroy(self.h)
            self.h = None

    def set_processed(self, n): lib().refe_set_processed_targets(self.h, n)
    def push_loaded_pos(self, v): lib().refe_push_loaded_pos(self.h, v)

    def process(self, matches, dest, target=0, lock=NO_LOCK):
        a, p = _bytes_ptr(np.array(dest, dtype=np.uint8, copy=True))
        m = np.ascontiguousarray(matches, dtype=np.uint64).reshape(-1, 3)
        return lib().refe_process_matches(self.h, m.ctypes.data_as(C.c_void_p), m.shape[0], p, a.size, target, lock)

    def after_sequence(self, t=0): lib().refe_after_sequence(self.h, t)
    def after_target(self, t=0): lib().refe_after_target(self.h, t)
    def reset_target(self, t=0): lib().refe_reset_target(self.h, t)

    def stream(self, which, target=0):
        n = lib().refe_stream(self.h, target, which, None, 0)
        buf = np.zeros(max(n, 1), dtype=np.uint8)
        lib().refe_stream(self.h, target, which, buf.ctypes.data_as(C.c_void_p), n)
        return bytes(buf[:n])

    def streams(self, target=0):
        names = ("literals", "mapOff", "mapOff5th", "mapLen", "gapDelta", "flags")
        return {names[i]: self.stream(i, target) for i in range(6)}


# ---- the -m3 reverse-complement pass (ref_harness.cpp: refrc_*), the reference's SimpleSequenceMatcher / CopMEMMatcher
def rc_find_matches(seq, target=55, min_len=0xFFFFFFFF):
    a, p = _bytes_ptr(seq)
    L = lib()
    L.refrc_find_matches.restype = C.c_uint64
    L.refrc_find_matches.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64]
    cap = max(1024, a.size // 8)
    out = np.zeros((cap, 3), dtype=np.uint64)
    n = L.refrc_find_matches(p, a.size, target, min_len, out.ctypes.data_as(C.c_void_p), cap)
    assert n <= cap
    return out[:n].copy()


def rc_match_sequence(seq, target=55, min_len=0xFFFFFFFF):
    a = np.array(seq, dtype=np.uint8, copy=True)
    L = lib()
    L.refrc_match_sequence.restype = C.c_uint64
    L.refrc_match_sequence.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p,
                                       C.POINTER(C.c_uint64), C.c_uint64]
    cap = max(4096, a.size)
    off, ln = np.zeros(cap, dtype=np.uint8), np.zeros(cap, dtype=np.uint8)
    no, nl = C.c_uint64(), C.c_uint64()
    n = L.refrc_match_sequence(a.ctypes.data_as(C.c_void_p), a.size, target, min_len, off.ctypes.data_as(C.c_void_p), C.byref(no),
                               ln.ctypes.data_as(C.c_void_p), C.byref(nl), cap)
    assert no.value <= cap and nl.value <= cap
    return a[:n].tobytes(), off[:no.value].tobytes(), ln[:nl.value].tobytes()

"""ctypes view of include/mbgc_copmem.h — the `-m3` reverse-complement pass over the literal stream
(SimpleSequenceMatcher::rcMatchSequence on CopMEMMatcher, matching/SimpleSequenceMatcher.cpp:165-176)."""
import ctypes as C

import numpy as np

from . import binding

EXPORTS = "mbgc_copmem_create mbgc_copmem_destroy mbgc_copmem_last_error mbgc_copmem_rc_matches mbgc_copmem_rc_match_sequence".split()
DEFAULT = 0xFFFFFFFF
_ready = False


def _lib():
    global _ready
    L = binding.lib()
    if not _ready:
        u64, vp, u32, P = C.c_uint64, C.c_void_p, C.c_uint32, C.POINTER
        L.mbgc_copmem_create.argtypes = [P(vp), C.c_int]
        L.mbgc_copmem_destroy.argtypes = [vp]
        L.mbgc_copmem_last_error.restype = C.c_char_p
        L.mbgc_copmem_rc_matches.argtypes = [vp, vp, u64, u32, u32, P(vp), P(u64), P(C.c_int)]
        L.mbgc_copmem_rc_match_sequence.argtypes = [vp, vp, u64, u32, u32, P(u64), P(vp), P(u64), P(vp), P(u64), P(u64)]
        _ready = True
    return L


class SimpleSequenceMatcher:
    """rcMatchSequence(sequence, rcMapOff, rcMapLen, targetMatchLength, minMatchLength) on the device."""

    def __init__(self, device=0):
        self.h = C.c_void_p()
        if _lib().mbgc_copmem_create(C.byref(self.h), device):
            raise binding.SwsemError(_lib().mbgc_copmem_last_error().decode())

    def close(self):
        if getattr(self, "h", None):
            _lib().mbgc_copmem_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def rc_matches(self, seq, target=55, min_len=DEFAULT):
        """-> ((n, 3) uint64 rows in push order: posSrcText, length, posDestText in the reverse-complemented text; (K, k1, k2, log2 hash size))"""
        a = np.ascontiguousarray(seq, dtype=np.uint8)
        out, n = C.c_void_p(), C.c_uint64()
        params = (C.c_int * 4)()
        r = _lib().mbgc_copmem_rc_matches(self.h, a.ctypes.data_as(C.c_void_p), a.size, target, min_len, C.byref(out), C.byref(n), params)
        if r:
            raise binding.SwsemError("copmem error %d: %s" % (r, _lib().mbgc_copmem_last_error().decode()))
        rows = np.zeros((n.value, 3), dtype=np.uint64)
        if n.value:
            rows[:] = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint64)), shape=(n.value, 3))
        return rows, tuple(params)

    def rc_match_sequence(self, seq, target=55, min_len=DEFAULT):
        """-> (rewritten sequence bytes, rcMapOff, rcMapLen, (unique matches, matched, overlapped))"""
        a = np.array(seq, dtype=np.uint8, copy=True)
        new_len, no, nl = C.c_uint64(), C.c_uint64(), C.c_uint64()
        off, ln = C.c_void_p(), C.c_void_p()
        st = (C.c_uint64 * 3)()
        r = _lib().mbgc_copmem_rc_match_sequence(self.h, a.ctypes.data_as(C.c_void_p), a.size, target, min_len, C.byref(new_len),
                                                 C.byref(off), C.byref(no), C.byref(ln), C.byref(nl), st)
        if r:
            raise binding.SwsemError("copmem error %d: %s" % (r, _lib().mbgc_copmem_last_error().decode()))
        take = lambda p, k: bytes(np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(k,))) if k else b""
        return a[:new_len.value].tobytes(), take(off, no.value), take(ln, nl.value), tuple(st)

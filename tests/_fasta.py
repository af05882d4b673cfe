"""ctypes helpers for the input stage: the C oracle (oracle/fasta_oracle.c) and, where oracle/_ref is built, the
reference's own kseq reader (oracle/ref_harness_fasta.cpp). Test infrastructure only."""
import ctypes as C

import numpy as np

import _orc
import _refh


class Rec(C.Structure):
    _fields_ = [("headerOff", C.c_uint64), ("headerLen", C.c_uint64), ("seqOff", C.c_uint64), ("seqLen", C.c_uint64)]


class RefRec(C.Structure):
    _fields_ = [("headerLen", C.c_uint64), ("seqOff", C.c_uint64), ("seqLen", C.c_uint64)]


def _u8(data):
    a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
    return a, a.ctypes.data_as(C.c_void_p)


def oracle_parse(data, uppercase=False):
    """-> dict(status, records=[(header bytes, sequence bytes)], dna_line_len, seq=all sequences back to back)"""
    L = _orc.lib()
    L.orc_fasta_parse.restype = C.c_uint64
    L.orc_fasta_parse.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.POINTER(Rec), C.c_uint64,
                                  C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
    a, p = _u8(data)
    out = np.zeros(max(a.size, 1), dtype=np.uint8)
    cap = a.size // 2 + 2
    recs = (Rec * cap)()
    nb, ll, st = C.c_uint64(), C.c_uint64(), C.c_int()
    n = L.orc_fasta_parse(p, a.size, int(uppercase), out.ctypes.data_as(C.c_void_p), recs, cap, C.byref(nb), C.byref(ll), C.byref(st))
    raw = a.tobytes()
    records = [(raw[r.headerOff: r.headerOff + r.headerLen], out[r.seqOff: r.seqOff + r.seqLen].tobytes()) for r in recs[:n]]
    return dict(status=st.value, records=records, dna_line_len=ll.value, seq=out[: nb.value].tobytes())


def ref_parse(data, uppercase=False):
    L = _refh.lib()
    L.reff_parse.restype = C.c_uint64
    L.reff_parse.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(RefRec), C.c_uint64,
                             C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
    a, p = _u8(data)
    out = np.zeros(max(a.size, 1), dtype=np.uint8)
    hdr = np.zeros(max(a.size, 1), dtype=np.uint8)
    cap = a.size // 2 + 2
    recs = (RefRec * cap)()
    nb, ll, st = C.c_uint64(), C.c_uint64(), C.c_int()
    n = L.reff_parse(p, a.size, int(uppercase), out.ctypes.data_as(C.c_void_p), hdr.ctypes.data_as(C.c_void_p), recs, cap,
                     C.byref(nb), C.byref(ll), C.byref(st))
    records, h = [], 0
    for r in recs[:n]:
        records.append((hdr[h: h + r.headerLen].tobytes(), out[r.seqOff: r.seqOff + r.seqLen].tobytes()))
        h += r.headerLen
    return dict(status=st.value, records=records, dna_line_len=ll.value, seq=out[: nb.value].tobytes())

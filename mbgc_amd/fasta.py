"""ctypes view of include/mbgc_fasta.h — the input stage (kseq_read_lossless_fasta for whole files in HBM)."""
import ctypes as C

import numpy as np

from . import binding

EXPORTS = "mbgc_fasta_create mbgc_fasta_destroy mbgc_fasta_last_error mbgc_fasta_parse_batch_dev mbgc_fasta_parse_host mbgc_fasta_host_alloc mbgc_fasta_host_free mbgc_fasta_upload".split()


class Record(C.Structure):
    _fields_ = [("headerOff", C.c_uint64), ("headerLen", C.c_uint64), ("seqOff", C.c_uint64), ("seqLen", C.c_uint64)]


def _lib():
    L = binding.lib()
    if not getattr(L, "_fasta_ready", False):
        L.mbgc_fasta_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
        L.mbgc_fasta_destroy.argtypes = [C.c_void_p]
        L.mbgc_fasta_last_error.restype = C.c_char_p
        L.mbgc_fasta_parse_batch_dev.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_int, C.c_int, C.c_void_p, C.c_uint64,
                                                 C.POINTER(C.c_uint64), C.POINTER(Record), C.c_uint64, C.POINTER(C.c_uint64),
                                                 C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
        L._fasta_ready = True
    return L


class FastaParser:
    def __init__(self, device=0):
        self.h = C.c_void_p()
        self._rec_cap = 4096
        if _lib().mbgc_fasta_create(C.byref(self.h), device):
            raise binding.SwsemError(_lib().mbgc_fasta_last_error().decode())

    def close(self):
        if self.h:
            _lib().mbgc_fasta_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def parse_batch_dev(self, files_ptr, file_offsets, out_ptr, out_cap, uppercase=False):
        """files_ptr: device buffer holding the files back to back, file f at [file_offsets[f], file_offsets[f+1]).
        -> dict(seq_base [nf+1], rec_base [nf+1], records (structured array), dna_line_len [nf], status [nf])"""
        offs = np.ascontiguousarray(file_offsets, dtype=np.uint64)
        nf = offs.size - 1
        P = C.POINTER(C.c_uint64)
        seq_base, rec_base = np.zeros(nf + 1, dtype=np.uint64), np.zeros(nf + 1, dtype=np.uint64)
        line, status = np.zeros(nf, dtype=np.uint64), np.zeros(nf, dtype=np.int32)
        rec_cap = max(self._rec_cap, nf)
        while True:
            recs = (Record * rec_cap)()
            r = _lib().mbgc_fasta_parse_batch_dev(self.h, files_ptr, offs.ctypes.data_as(P), nf, int(uppercase), out_ptr, out_cap,
                                                  seq_base.ctypes.data_as(P), recs, rec_cap, rec_base.ctypes.data_as(P),
                                                  line.ctypes.data_as(P), status.ctypes.data_as(C.POINTER(C.c_int)))
            if r == -104 and int(rec_base[-1]) > rec_cap:              # the table was too small: the call says how many it needs
                rec_cap = self._rec_cap = int(rec_base[-1])
                continue
            if r:
                raise binding.SwsemError(_lib().mbgc_fasta_last_error().decode())
            break
        n = int(rec_base[-1])
        arr = np.frombuffer(recs, dtype=[("headerOff", "<u8"), ("headerLen", "<u8"), ("seqOff", "<u8"), ("seqLen", "<u8")], count=n).copy()
        return dict(seq_base=seq_base, rec_base=rec_base, records=arr, dna_line_len=line, status=status)

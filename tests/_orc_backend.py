"""The CPU oracle behind the *device* surface of mbgc_amd.binding.SlidingWindowSparseEMMatcher
(match_batch_dev / emit_batch / emit_pack_dev / load_ref_dev / revcomp_dev on raw pointers), so the
round protocol (mbgc_amd.rounds) can be exercised without a GPU: world_size-2 gloo tests run it on
CPU tensors. Test infrastructure only."""
import ctypes as C

import numpy as np

import _driver
import _orc


def _view(ptr, n):
    if n == 0:
        return np.zeros(0, dtype=np.uint8)
    return np.ctypeslib.as_array(C.cast(int(ptr), C.POINTER(C.c_uint8)), shape=(int(n),))


class OracleDeviceMatcher:
    def __init__(self, max_ref_len, **kw):
        self.o = _orc.OracleMatcher(max_ref_len, **kw)
        self._contigs, self._locks, self._matches, self._em, self._em_prev, self._sel_prev = [], [], [], [], [], False

    # pass-through state API
    def __getattr__(self, name):
        return getattr(self.o, name)

    def load_ref_dev(self, ptr, n, load_rc=False, add_sep=True, sep=0):
        self.o.load_ref(_view(ptr, n).copy(), load_rc, add_sep, sep)

    def finalize_targets(self, ext_ptrs, ext_lens, locks, lazy=True, add_sep=True, sep=0):
        out = []
        for p, n, lk in zip(ext_ptrs, ext_lens, locks):
            if n:
                self.o.load_ref(_view(p, n).copy(), False, add_sep, sep)
            if lazy:
                self.o.load_separator(sep)
            out.append(self.o.loaded_ref_length())
            assert self.o.release_lock(int(lk)) == 0
        return np.array(out, dtype=np.uint64)

    def revcomp_dev(self, src, n, dst):
        _view(dst, n)[:] = _driver.revcomp(_view(src, n))

    def match_batch_dev(self, ptr, offsets, min_len=32, locks=None):
        offs = [int(x) for x in offsets]
        n = len(offs) - 1
        self._contigs = [_view(ptr + offs[i], offs[i + 1] - offs[i]).copy() for i in range(n)]
        self._locks = [int(x) for x in locks] if locks is not None else [_orc.NO_LOCK] * n
        self._matches = [self.o.match(c, min_len, lk) for c, lk in zip(self._contigs, self._locks)]

    def batch_counts(self):
        return np.array([len(m) for m in self._matches], dtype=np.uint64)

    def batch_matches(self, i, count):
        return self._matches[i][: int(count)]

    def emit_set_host_copy(self, on):
        pass

    def emit_batch(self, params, contigs=None, locks=None, factors=None, processed=None, target_idx=None, loaded=None, n=None):
        n = n if n is not None else len(self._contigs)
        self._em_prev, self._em = self._em, []
        for k in range(n):
            c = k if contigs is None else int(contigs[k])
            em = _orc.OracleEmitter(self.o, params)
            un = em.process(self._matches[c], self._contigs[c], int(locks[k]) if locks is not None else _orc.NO_LOCK,
                            int(factors[k]) if factors is not None else 128, int(processed[k]) if processed is not None else 0,
                            int(target_idx[k]) if target_idx is not None else 0, loaded)
            self._em.append((un, em))
        return n

    def emit_batch_begin(self, *a, **k):
        return self.emit_batch(*a, **k)

    def emit_batch_begin_spec(self, params, locks, factors, processed, target_idx, loaded, n, ext_ptrs, ext_lens, target_locks,
                              pred_ext, pred_rc, factor, rc_factor, lazy=True, add_sep=True, sep=0, gate=0, reduce=None,
                              verdict=None, veto=False):
        """the speculative finalize's contract (include/mbgc_swsem.h): emission, this replica's verdict on the prediction,
        the reduction over the replicas, then the finalize of all targets or nothing"""
        self.emit_batch(params, None, locks, factors, processed, target_idx, loaded, n)
        ok = not veto
        for k in range(n):
            un, ln = int(self._em[k][0]), len(self._contigs[k])
            ok = ok and un != _orc.SKIPPED and (un * factor > ln) == bool(pred_ext[k]) and (un * rc_factor > ln) == bool(pred_rc[k])
        if reduce is not None:
            np.ctypeslib.as_array(C.cast(int(gate), C.POINTER(C.c_int32)), shape=(1,))[0] = int(ok)
            reduce(None)
            ok = verdict() == 1 and ok
        if not ok:
            return False, np.zeros(len(ext_lens), dtype=np.uint64)
        return True, self.finalize_targets(ext_ptrs, ext_lens, target_locks, lazy, add_sep, sep)

    def emit_batch_end(self):
        pass

    def emit_select(self, previous):
        self._sel_prev = bool(previous)

    def _selected(self):
        return self._em_prev if self._sel_prev else self._em

    def emit_unmatched(self, n):
        return np.array([u for u, _ in self._em[:n]], dtype=np.uint64)

    def emit_pack_sizes(self, n):
        sizes = np.zeros((n, 6), dtype=np.uint64)
        for k, (un, em) in enumerate(self._selected()[:n]):
            if un != _orc.SKIPPED:
                sizes[k] = [len(em.stream(i)) for i in range(6)]
        return sizes, int(sizes.sum())

    def emit_pack_dev(self, dst, cap, stream=None):
        blob = b"".join(em.stream(i) for un, em in self._selected() if un != _orc.SKIPPED for i in range(6))
        assert len(blob) <= cap
        if blob:
            _view(dst, len(blob))[:] = np.frombuffer(blob, dtype=np.uint8)
        return len(blob)

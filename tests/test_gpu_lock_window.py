"""What an emission may read of the reference: everything except its own lock window [loading position at its start,
its matching-lock position) — candidates there are refused at match time (SlidingWindowSparseEMMatcher.cpp:212-220),
pairs do not span the lock (TextMatchers.h:46-50), the right extension stops at the loading position and the left one
at the lock (MBGC_Encoder.cpp:318-335, :379-384). That window is what the reference's loader overwrites while the workers
read, and what this library's finalize may overwrite while an emission's second phase is still running. Checked
deterministically: emit, fill the window with garbage, emit again from the same match rows, compare every stream byte
(then put the bytes back and go on) — over rounds with locks, a sliding window and several laps of the circular buffer."""
import numpy as np
import pytest

import _orc
from mbgc_amd import synth

pytestmark = pytest.mark.gpu


def collection(n, length, div, seed):
    base = synth.base_codes(length, seed)
    return [synth.genome(base, i, div) for i in range(n)]


def windows(pos1, lock, max_ref):
    """byte ranges of the lock window [pos1, lock), which may wrap around the buffer's end"""
    if lock > pos1:
        return [(pos1, min(lock, max_ref))]
    return [(pos1, max_ref), (1, lock)]


@pytest.mark.parametrize("lim,div,garbage", [(3_000_000, 0.004, 0xFF), (600_000, 0.004, 0xFF), (600_000, 0.004, 0x00),
                                               (450_000, 0.02, 0x41)])
def test_emission_does_not_read_its_lock_window(lim, div, garbage):
    import torch
    from mbgc_amd import binding
    gs = collection(17, 70_000, div, seed=lim % 97)
    R = 4
    h = binding.SlidingWindowSparseEMMatcher(lim)
    h.set_sliding_window_size(16)
    h.load_ref(gs[0], load_rc=True)
    p = binding.emit_params(1)
    loaded = [h.loaded_ref_length()]
    done, poisoned_bytes = 0, 0
    for r0 in range(1, len(gs), R):
        chunk = gs[r0:r0 + R]
        n = len(chunk)
        buf = torch.from_numpy(np.concatenate(chunk)).to("cuda:0")
        offs = np.zeros(n + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([c.size for c in chunk])
        torch.cuda.synchronize()
        locks = [h.acquire_lock() for _ in range(n)]
        pos1 = int(h.loading_position())
        h.match_batch_dev(buf.data_ptr(), offs, 32, locks)
        args = (p, None, locks, [128] * n, [done + t for t in range(n)], [done + t for t in range(n)], loaded)
        h.emit_batch(*args, n=n)
        first = [h.emit_result(c) for c in range(n)]
        saved = []
        for a, b in windows(pos1, int(min(locks)), int(h.max_ref_length())):
            if b > a:
                saved.append((a, h.ref(b - a, a)))
                h.write_ref(a, np.full(b - a, garbage, dtype=np.uint8))
                poisoned_bytes += b - a
        h.emit_batch(*args, n=n)
        for c in range(n):
            un, streams, _ = h.emit_result(c)
            assert un == first[c][0], (r0, c)
            for k in streams:
                assert streams[k] == first[c][1][k], (r0, c, k)
        for a, data in saved:
            h.write_ref(a, data)
        after = h.finalize_targets([buf.data_ptr() + int(offs[c]) for c in range(n)], [int(offs[c + 1] - offs[c]) for c in range(n)],
                                   locks, lazy=True)
        loaded += [int(x) for x in after]
        done += n
        torch.cuda.synchronize()
    assert poisoned_bytes > 100_000
    if lim < 1_000_000:
        assert h.loaded_ref_length() > lim                 # the loader has gone round the buffer
    h.close()


@pytest.mark.parametrize("seed,lim,div,R", [(1, 12_000_000, 0.004, 8), (2, 7_000_000, 0.01, 4)])
def test_rounds_over_several_laps_equal_the_oracle(seed, lim, div, R):
    """40 x 1 Mbp through the product's round protocol (finalize beside the running emission, speculative finalize,
    fingerprints told stale by their epochs) while the loader goes round the buffer: every stream byte, the lock and
    refExtSize bytes and the hash table equal the oracle's"""
    import torch
    import _driver
    from mbgc_amd import binding
    from mbgc_amd.rounds import RoundRunner
    base = synth.base_codes(1_000_000, seed)
    gs = [synth.genome(base, i, div) for i in range(41)]
    h = binding.SlidingWindowSparseEMMatcher(lim)
    o = _orc.OracleMatcher(lim)
    b = _driver.encode_rounds(o, lambda: _orc.OracleEmitter(o), [gs[0]], [[g] for g in gs[1:]], R)
    h.set_sliding_window_size(16)
    h.load_ref(gs[0], load_rc=True)
    runner = RoundRunner(h, 0, 1, None, "cuda:0", lazy=True, emit_params=binding.emit_params(1))
    runner.start()
    bufs = []
    for r0 in range(1, len(gs), R):
        chunk = gs[r0:r0 + R]
        offs = np.zeros(len(chunk) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([c.size for c in chunk])
        bufs.append((torch.from_numpy(np.concatenate(chunk)).to("cuda:0"), offs))
    torch.cuda.synchronize()
    for i, (buf, offs) in enumerate(bufs):
        runner.run_round(buf, offs, next_batch=bufs[i + 1] if i + 1 < len(bufs) else None)
    runner.flush()
    for k in b["streams"]:
        assert bytes(runner.streams[k]) == b["streams"][k], k
    assert bytes(runner.locks_stream) == b["locks"] and bytes(runner.ref_ext_sizes) == b["refExtSize"]
    assert h.loaded_ref_length() == o.loaded_ref_length() and h.loaded_ref_length() > lim
    assert np.array_equal(h.ht(), o.ht())
    h.close(); o.close()

"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical schedules.
Bit-exact bar: reference bytes, hash-table image and (posSrcText, length, posDestText) rows."""
import numpy as np
import pytest

import _orc
from mbgc_amd import synth

pytestmark = pytest.mark.gpu
NO_LOCK = _orc.NO_LOCK


@pytest.fixture(scope="module")
def binding():
    from mbgc_amd import binding as b
    assert b.lib().swsem_device_count() > 0, "no HIP device: the GPU tests must run on the MI355X box"
    return b


def pair(binding, max_len, **kw):
    return binding.SlidingWindowSparseEMMatcher(max_len, **kw), _orc.OracleMatcher(max_len, **kw)


def assert_same_state(h, o):
    assert h.loading_position() == o.loading_position()
    assert h.ref_length() == o.ref_length()
    assert h.loaded_ref_length() == o.loaded_ref_length()
    n = o.ref_length()
    assert np.array_equal(h.ref(n)[1:], o.ref(n)[1:])
    a, b = h.ht(), o.ht()
    assert np.array_equal(a, b), "HT image differs at %d buckets" % int((a != b).sum())


def small_collection(n, length, div=0.01, seed=7):
    base = synth.base_codes(length, seed)
    return [synth.genome(base, i, div) for i in range(n)]


@pytest.mark.parametrize("sizes", [(10275,), (10027,), (27,), (28,), (29,), (44,), (2075, 2076, 1), (4125, 17, 2049, 90000)])
def test_insert_tail_quirk(binding, sizes):
    rng = np.random.default_rng(sum(sizes))
    h, o = pair(binding, 1 << 20)
    for s in sizes:
        t = synth.ACGT[rng.integers(0, 4, s)]
        for m in (h, o):
            m.load_ref(t, load_rc=False, add_sep=True)
        assert_same_state(h, o)
    q = o.ref()[1:]
    assert np.array_equal(h.match(q), o.match(q))


def test_revcomp_all_bytes(binding):
    text = np.concatenate([np.arange(1, 256, dtype=np.uint8), np.frombuffer(b"ACGTNacgtnRYKMBDHVSWryk" * 9, dtype=np.uint8)])
    h, o = pair(binding, 1 << 20)
    for m in (h, o):
        m.load_ref(text, load_rc=True, add_sep=False)
    assert_same_state(h, o)


def test_sequential_collection(binding):
    gs = small_collection(8, 200_000, 0.01)
    h, o = pair(binding, 16_000_000)
    for m in (h, o):
        m.disable_sliding_window()
        m.load_ref(gs[0], load_rc=True)
    total = 0
    for g in gs[1:]:
        a, b = h.match(g), o.match(g)
        assert np.array_equal(a, b)
        total += len(b)
        for m in (h, o):
            m.load_ref(g)
    assert total > 10_000
    assert_same_state(h, o)


@pytest.mark.parametrize("div", [0.0, 0.001, 0.05, 0.25])
def test_divergence_extremes(binding, div):
    """identical genomes (one giant match, capped extension runs continued on demand) ... unrelated ones"""
    gs = small_collection(3, 300_000, div, seed=21)
    h, o = pair(binding, 8_000_000)
    for m in (h, o):
        m.load_ref(gs[0], load_rc=True)
    for g in gs[1:]:
        a, b = h.match(g), o.match(g)
        assert np.array_equal(a, b)
        for m in (h, o):
            m.load_ref(g)


def test_ragged_and_empty_queries(binding):
    gs = small_collection(2, 50_000, 0.01, seed=2)
    h, o = pair(binding, 1 << 20)
    for m in (h, o):
        m.load_ref(gs[0], load_rc=True)
    for n in (0, 1, 27, 28, 29, 31, 32, 33, 59, 60, 4095, 4096, 4097, 4096 + 27, 4096 + 28, 8192 + 27):
        q = gs[1][:n]
        assert np.array_equal(h.match(q), o.match(q)), n
    # unaligned starts inside a larger buffer
    for off in (1, 2, 3, 5):
        q = gs[1][off:off + 10_001]
        assert np.array_equal(h.match(q), o.match(q)), off


def test_low_complexity_and_non_acgt(binding):
    rng = np.random.default_rng(4)
    rep = np.frombuffer((b"A" * 3000 + b"ACACACAC" * 400 + b"N" * 500 + b"acgtn" * 300), dtype=np.uint8)
    rnd = synth.ACGT[rng.integers(0, 4, 20_000)]
    ref = np.concatenate([rep, rnd, rep[::-1].copy()])
    q = np.concatenate([rnd[:7000], rep, rnd[9000:], np.frombuffer(b"A" * 9000, dtype=np.uint8)])
    h, o = pair(binding, 1 << 20)
    for m in (h, o):
        m.load_ref(ref, load_rc=True)
    assert np.array_equal(h.match(q), o.match(q))
    assert_same_state(h, o)


@pytest.mark.parametrize("sequential", [True, False])
def test_wrap_quirk_and_locks(binding, sequential):
    rng = np.random.default_rng(5)
    base = rng.integers(0, 4, 60_000)
    h, o = pair(binding, 100_000)
    if sequential:
        h.disable_sliding_window(); o.disable_sliding_window()
    for step in range(14):
        g = base.copy()
        mask = rng.random(g.size) < 0.02
        g[mask] = (g[mask] + 1) & 3
        g = synth.ACGT[g][: int(rng.integers(20_000, 60_000))]
        lock = NO_LOCK
        if not sequential:
            lh, lo = h.acquire_lock(), o.acquire_lock()
            assert lh == lo
            lock = lo
        assert np.array_equal(h.match(g, 32, lock), o.match(g, 32, lock)), step
        for m in (h, o):
            m.load_ref(g, load_rc=bool(step % 3 == 0), add_sep=True)
            if step % 2:
                m.load_separator(0)
        if not sequential:
            h.release_lock(lock); o.release_lock(lock)
        assert_same_state(h, o)
    assert o.loaded_ref_length() > 150_000


def test_other_kmer_parameters(binding):
    gs = small_collection(3, 80_000, 0.02, seed=9)
    for L, k1, margin in ((32, 16, 24), (24, 8, 16), (40, 6, 16), (48, 16, 16), (32, 15, 16), (32, 7, 16), (40, 5, 24)):   # odd k1: identity-encoded entries (.h:74-76)
        h, o = pair(binding, 4_000_000, L=L, k1=k1, skip_margin=margin)
        assert h.K() == o.K() and h.hash_size() == o.hash_size()
        for m in (h, o):
            m.load_ref(gs[0], load_rc=True)
        for g in gs[1:]:
            assert np.array_equal(h.match(g, L), o.match(g, L)), (L, k1)
            for m in (h, o):
                m.load_ref(g)
        assert_same_state(h, o)
        h.close(); o.close()


def test_error_paths(binding):
    with pytest.raises(binding.SwsemError):
        binding.SlidingWindowSparseEMMatcher(1 << 20, k1=0)          # the sampling step is a positive integer (MBGC_Params.h:593-597)
    h = binding.SlidingWindowSparseEMMatcher(1 << 20)
    with pytest.raises(binding.SwsemError):
        h.match(np.zeros(100, dtype=np.uint8), min_len=16)           # minMatchLength < K
    h.set_sliding_window_size(16)
    with pytest.raises(binding.SwsemError):
        h.release_lock(12345)                                        # invalid worker lock value


def test_batch_round_on_device(binding):
    """A round of contigs resident in HBM (torch tensor) against the frozen reference."""
    import torch
    gs = small_collection(7, 150_000, 0.01, seed=31)
    h, o = pair(binding, 16_000_000)
    for m in (h, o):
        m.set_sliding_window_size(16)
        m.load_ref(gs[0], load_rc=True)
    contigs = [gs[1], gs[2][:70_001], gs[2][70_001:], gs[3], gs[4][:13], gs[5], gs[6]]
    offs = np.zeros(len(contigs) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([c.size for c in contigs])
    buf = torch.from_numpy(np.concatenate(contigs)).to("cuda:0")
    lock = [h.acquire_lock() for _ in contigs]
    lock_o = [o.acquire_lock() for _ in contigs]
    assert lock == lock_o
    torch.cuda.synchronize()
    h.match_batch_dev(buf.data_ptr(), offs, 32, lock)
    counts = h.batch_counts()
    exp = [o.match(c, 32, l) for c, l in zip(contigs, lock_o)]
    for i, e in enumerate(exp):
        assert counts[i] == len(e), i
        assert np.array_equal(h.batch_matches(i, counts[i]), e), i
    fp, tot, ln = h.batch_fingerprint()
    allm = np.concatenate(exp)
    assert fp == _orc.fingerprint(allm) and tot == len(allm) and ln == int(allm[:, 1].sum())


@pytest.mark.parametrize("env", [{"SWSEM_RESOLVE": "seq"}, {"SWSEM_RB": "1"}, {"SWSEM_RB": "2"}, {"SWSEM_RB": "16"}, {"SWSEM_RB": "64"},
                                 {"SWSEM_CHAINS": "1"}, {"SWSEM_CHAINS": "1", "SWSEM_RB": "1"}, {"SWSEM_CHAINS": "1", "SWSEM_RB": "5"},
                                 {"SWSEM_RB": "3"}])
def test_resolve_variants_agree_with_oracle(binding, env, monkeypatch):
    """one wave replaying a whole contig, block-parallel speculation at several block lengths (units of 1024 positions),
    four chains per wave (default) and one chain per wave, both launch orders: same rows"""
    for k, val in env.items():
        monkeypatch.setenv(k, val)
    for div, seed in ((0.01, 41), (0.0005, 42), (0.1, 43)):
        gs = small_collection(4, 120_000, div, seed=seed)
        h, o = pair(binding, 8_000_000)
        for m in (h, o):
            m.load_ref(gs[0], load_rc=True)
        for g in gs[1:]:
            assert np.array_equal(h.match(g), o.match(g)), (env, div)
            for m in (h, o):
                m.load_ref(g)
        h.close(); o.close()


def test_speculation_mostly_accepted(binding):
    """on the 99 %-identity regime nearly every resolve block is accepted as speculated"""
    import torch
    gs = small_collection(3, 1_000_000, 0.01, seed=77)
    h = binding.SlidingWindowSparseEMMatcher(16_000_000)
    h.load_ref(gs[0], load_rc=True)
    buf = torch.from_numpy(np.concatenate(gs[1:])).to("cuda:0")
    torch.cuda.synchronize()
    h.match_batch_dev(buf.data_ptr(), np.array([0, 1_000_000, 2_000_000], dtype=np.uint64), 32, None)
    h.batch_counts()
    st = h.batch_stats()
    nblocks = 2 * ((1_000_000 - 27 + 4095) // 4096 + 3) // 4
    assert st["replayed_blocks"] <= nblocks // 10, st


def test_the_devices_numa_node_is_what_sysfs_says(binding):
    """swsem_device_numa_node (the C++ host keeps its reader threads on that node): -1 or a node that exists"""
    import ctypes
    import os
    L = binding.lib()
    L.swsem_device_numa_node.argtypes = [ctypes.c_int]
    L.swsem_device_numa_node.restype = ctypes.c_int
    node = L.swsem_device_numa_node(0)
    assert node == -1 or os.path.isdir("/sys/devices/system/node/node%d" % node)
    assert L.swsem_device_numa_node(10_000) == -1

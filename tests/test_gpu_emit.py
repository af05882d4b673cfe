"""GPU parity of the stream emission (MBGC_Encoder::processMatches on the device) against the CPU
oracle, driven through the same target loops (tests/_driver.py). Bit-exact on all six streams."""
import numpy as np
import pytest

import _driver
import _orc
from mbgc_amd import synth

pytestmark = pytest.mark.gpu
NO_LOCK = _orc.NO_LOCK


@pytest.fixture(scope="module")
def binding():
    from mbgc_amd import binding as b
    assert b.lib().swsem_device_count() > 0
    return b


class HipEmitter:
    """_driver emitter on top of swsem_emit: consumes the match rows the handle holds from the
    preceding match call (they never leave HBM)."""

    def __init__(self, binding, matcher, params=None):
        self.b, self.m = binding, matcher
        self.p = params if params is not None else binding.emit_params(1)
        self.s = {k: b"" for k in binding.STREAM_NAMES}
        self.counters = dict(extensionsMatchedChars=0, extensionsMismatches=0, totalMatched=0, removedGapBreakingMatches=0)

    def process(self, m, contig, lock, factor, processed, target_idx, loaded):
        un, streams, st = self.m.emit(self.p, 0, lock, factor, processed, target_idx, loaded)
        if un != self.b.SKIPPED:
            for k in self.s:
                self.s[k] += streams[k]
            for k in self.counters:
                self.counters[k] += getattr(st, k)
        return un

    def put(self, which, data): self.s[self.b.STREAM_NAMES[which]] += bytes(data)
    def streams(self): return dict(self.s)


def small_collection(n, length, div=0.01, seed=7):
    base = synth.base_codes(length, seed)
    return [synth.genome(base, i, div) for i in range(n)]


def compare(a, b):
    for k in b:
        assert a[k] == b[k], "%s differs (%d vs %d bytes)" % (k, len(a[k]), len(b[k]))


@pytest.mark.parametrize("mode,lazy", [(1, True), (1, False), (0, True), (2, True)])
def test_emission_sequential(binding, mode, lazy):
    gs = small_collection(5, 150_000, 0.01, seed=11)
    files = [[g[:70_000], g[70_000:]] for g in gs]
    lim, _ = _driver.ref_length_limit(len(files), 150_000)
    margin = 24 if mode >= 2 else 16
    h = binding.SlidingWindowSparseEMMatcher(lim, skip_margin=margin)
    o = _orc.OracleMatcher(lim, skip_margin=margin)
    he = HipEmitter(binding, h, binding.emit_params(mode, lazyDecompressionSupport=int(lazy)))
    oe = _orc.OracleEmitter(o, _orc.emit_params(mode, lazyDecompressionSupport=int(lazy)))
    pol = _driver.Policy(mode)
    a = _driver.encode_sequential(h, he, files, pol, lazy=lazy)
    b = _driver.encode_sequential(o, oe, files, pol, lazy=lazy)
    assert a["locks"] == b["locks"] and a["refExtSize"] == b["refExtSize"]
    compare(he.streams(), oe.streams())
    oc = oe.counters()
    for k in he.counters:
        assert he.counters[k] == oc[k], k
    assert len(oe.streams()["flags"]) > 1000


@pytest.mark.parametrize("round_size", [1, 3, 8])
def test_emission_rounds_with_locks_and_wrap(binding, round_size):
    gs = small_collection(17, 120_000, 0.015, seed=3)
    g0 = [gs[0][:50_000], gs[0][50_000:]]
    targets = [[g[:40_000], g[40_000:]] for g in gs[1:]]
    lim = 900_000
    h = binding.SlidingWindowSparseEMMatcher(lim)
    o = _orc.OracleMatcher(lim)
    a = _driver.encode_rounds(h, lambda: HipEmitter(binding, h), g0, targets, round_size)
    b = _driver.encode_rounds(o, lambda: _orc.OracleEmitter(o), g0, targets, round_size)
    assert a["locks"] == b["locks"] and a["refExtSize"] == b["refExtSize"] and a["unmatched"] == b["unmatched"]
    compare(a["streams"], b["streams"])
    assert o.loaded_ref_length() > lim


@pytest.mark.parametrize("mode", [0, 2])
def test_rounds_that_go_on_in_units_of_other_sizes(binding, mode):
    """a round whose first pass gives contigs up goes on in units of allowedTargetsOutrunForDissimilarContigs + 1 stopped targets
    (tests/_driver.py): -m0 allows an outrun of 4 (units of up to five, and only the last three targets of a round of 8 can stop),
    -m2 none (units of one, every target but the first can). Two genomes of three are 7 % from the rest."""
    base = synth.base_codes(60_000, 77)
    gs = [synth.genome(base, i, 0.004 if i % 3 == 1 else 0.07) for i in range(18)]
    targets = [[g[:25_000], g[25_000:]] for g in gs[1:]]
    lim = 3_000_000
    margin = 24 if mode >= 2 else 16
    h = binding.SlidingWindowSparseEMMatcher(lim, skip_margin=margin)
    o = _orc.OracleMatcher(lim, skip_margin=margin)
    pol = _driver.Policy(mode)
    stops = []

    class Watch(_orc.OracleEmitter):
        def process(s, m, contig, lock, factor, processed, t, loaded):
            r = super().process(m, contig, lock, factor, processed, t, loaded)
            if r == _driver.SKIPPED:
                stops.append(t)
            return r
    a = _driver.encode_rounds(h, lambda: HipEmitter(binding, h, binding.emit_params(mode)), [gs[0]], targets, 8, pol)
    b = _driver.encode_rounds(o, lambda: Watch(o, _orc.emit_params(mode)), [gs[0]], targets, 8, pol)
    assert a["locks"] == b["locks"] and a["refExtSize"] == b["refExtSize"] and a["unmatched"] == b["unmatched"]
    compare(a["streams"], b["streams"])
    assert len(set(stops)) >= 3, stops
    assert all(t % 8 > pol.outrun for t in stops), stops                # (ENC.cpp:203: never with processed >= targetIdx - outrun)


def test_emission_mixed_alphabet_and_edges(binding):
    """lower case, N runs, IUPAC codes, a contig without matches, a tiny contig, an identical contig"""
    rng = np.random.default_rng(8)
    base = synth.ACGT[rng.integers(0, 4, 60_000)].copy()
    base[1000:1400] = ord("N")
    base[5000:5600] = np.frombuffer(b"acgtnRYKM" * 67, dtype=np.uint8)[:600]
    def mut(x, rate, seed):
        r = np.random.default_rng(seed)
        y = x.copy()
        idx = np.nonzero(r.random(x.size) < rate)[0]
        y[idx] = synth.ACGT[r.integers(0, 4, idx.size)]
        return y
    files = [[base], [mut(base, 0.02, 1)], [synth.ACGT[rng.integers(0, 4, 9_000)]], [base[:40]], [base.copy()],
             [mut(base, 0.3, 2)[:20_000], mut(base, 0.005, 3)]]
    lim, _ = _driver.ref_length_limit(len(files), 60_000)
    h = binding.SlidingWindowSparseEMMatcher(lim)
    o = _orc.OracleMatcher(lim)
    he, oe = HipEmitter(binding, h), _orc.OracleEmitter(o)
    _driver.encode_sequential(h, he, files)
    _driver.encode_sequential(o, oe, files)
    compare(he.streams(), oe.streams())


def test_emit_batch_matches_per_contig_calls(binding):
    import torch
    gs = small_collection(6, 100_000, 0.01, seed=13)
    h = binding.SlidingWindowSparseEMMatcher(8_000_000)
    o = _orc.OracleMatcher(8_000_000)
    for m in (h, o):
        m.set_sliding_window_size(16)
        m.load_ref(gs[0], load_rc=True)
    contigs = gs[1:]
    offs = np.zeros(len(contigs) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([c.size for c in contigs])
    buf = torch.from_numpy(np.concatenate(contigs)).to("cuda:0")
    locks = [h.acquire_lock() for _ in contigs]
    assert locks == [o.acquire_lock() for _ in contigs]
    torch.cuda.synchronize()
    h.match_batch_dev(buf.data_ptr(), offs, 32, locks)
    loaded = [h.loading_position()]
    p = binding.emit_params(1)
    n = h.emit_batch(p, locks=locks, factors=[128] * len(contigs), processed=[0] * len(contigs),
                     target_idx=list(range(len(contigs))), loaded=loaded)
    assert n == len(contigs)
    for i, c in enumerate(contigs):
        oe = _orc.OracleEmitter(o)
        m = o.match(c, 32, locks[i])
        un = oe.process(m, c, locks[i], 128, 0, i, loaded)
        got_un, streams, st = h.emit_result(i)
        if un == _orc.SKIPPED:
            assert got_un == binding.SKIPPED
            continue
        assert got_un == un
        compare(streams, oe.streams())


@pytest.mark.parametrize("lim,div", [(3_000_000, 0.012), (700_000, 0.012), (500_000, 0.07)])
def test_round_runner_equals_driver_rounds(binding, lim, div):
    """mbgc_amd.rounds.RoundRunner (the product's round protocol, device-resident, batched finalize with one
    insertion launch per round) against the reference loop restated in tests/_driver.py driven on the
    oracle: same streams, lock and refExtSize bytes, same hash table — also across circular wraps and
    with dissimilar-contig retries."""
    import torch
    from mbgc_amd.rounds import RoundRunner
    gs = small_collection(13, 90_000, div, seed=5)
    h = binding.SlidingWindowSparseEMMatcher(lim)
    o = _orc.OracleMatcher(lim)
    R = 4
    # oracle side
    b = _driver.encode_rounds(o, lambda: _orc.OracleEmitter(o), [gs[0]], [[g] for g in gs[1:]], R)
    # device side
    h.set_sliding_window_size(16)
    h.load_ref(gs[0], load_rc=True)
    runner = RoundRunner(h, 0, 1, None, "cuda:0", lazy=True, emit_params=binding.emit_params(1))
    runner.start()
    for r0 in range(1, len(gs), R):
        chunk = gs[r0:r0 + R]
        buf = torch.from_numpy(np.concatenate(chunk)).to("cuda:0")
        offs = np.zeros(len(chunk) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([c.size for c in chunk])
        torch.cuda.synchronize()
        runner.run_round(buf, offs)
    runner.flush()
    for k in b["streams"]:
        assert bytes(runner.streams[k]) == b["streams"][k], k
    assert bytes(runner.locks_stream) == b["locks"] and bytes(runner.ref_ext_sizes) == b["refExtSize"]
    assert h.loaded_ref_length() == o.loaded_ref_length()
    assert np.array_equal(h.ht(), o.ht())


@pytest.mark.parametrize("mode,lazy,lim", [(1, True, 4_000_000), (2, False, 4_000_000), (1, True, 600_000)])
def test_hip_streams_decode_back_to_the_contig(binding, mode, lazy, lim):
    """round trip through the decoder's automaton (oracle/decode_oracle.c, MBGC_Decoder.cpp:319-523): the HIP path's
    six streams, decoded against the HIP handle's own reference buffer, give the contig back (lim = 600 000: the
    circular buffer has wrapped)"""
    gs = small_collection(9 if lim < 1_000_000 else 5, 100_000, 0.012, seed=31 + mode)
    p = binding.emit_params(mode)
    p.lazyDecompressionSupport = int(lazy)
    po = _orc.emit_params(mode)
    po.lazyDecompressionSupport = int(lazy)
    h = binding.SlidingWindowSparseEMMatcher(lim, skip_margin=24 if mode >= 2 else 16)
    h.load_ref(gs[0], load_rc=True)
    loaded = [h.loaded_ref_length()]
    for t, g in enumerate(gs[1:]):
        for c in (g[:40_000], g[40_000:]):
            h.match(c)
            un, streams, _ = h.emit(p, 0, binding.NO_LOCK, 128, t, t, loaded)
            back, un2 = _orc.decode_contig(h.ref(h.max_ref_length()), po, streams, _orc.NO_LOCK)
            assert back.size == c.size and np.array_equal(back, c), (mode, lazy, t)
            assert un2 == (un & 0xFFFFFFFF)
            h.load_ref(c)
            if lazy:
                h.load_separator(0)
            loaded.append(h.loaded_ref_length())
    h.close()


@pytest.mark.parametrize("mode,lazy", [(1, True), (1, False), (0, True), (3, True)])
def test_long_gaps_between_paired_matches(binding, mode, lazy):
    """A stretch between two matches on one diagonal is a gap (MBGC_Encoder.cpp:262-270): extendMatchRight walks all of it,
    whatever its length (:338-371) — on the device a wave takes a long one together (ext_right_gap_wide). Divergent stretches
    of 100 B .. 70 kB that keep their length, one of them across a separator byte of the reference (where the comparison
    stops under lazy decompression), and the same with unmatched runs of plain literals that long."""
    rng = np.random.default_rng(41 + mode)
    base = synth.ACGT[rng.integers(0, 4, 400_000)]

    def diverged(g, spots, same_length=True):
        g = g.copy()
        out, at = [], 0
        for a, n in spots:
            out.append(g[at:a])
            out.append(synth.ACGT[rng.integers(0, 4, n if same_length else n + 7)])
            at = a + n
        out.append(g[at:])
        return np.concatenate(out)

    spots = [(20_000, 100), (40_000, 193), (60_000, 500), (90_000, 5_000), (150_000, 70_000), (300_000, 1_000)]
    files = [[base], [diverged(base, spots)], [diverged(base, spots[1:], same_length=False)], [diverged(base, spots[::2])]]
    lim, _ = _driver.ref_length_limit(len(files) + 1, base.size)
    margin = 24 if mode >= 2 else 16
    h = binding.SlidingWindowSparseEMMatcher(lim, skip_margin=margin)
    o = _orc.OracleMatcher(lim, skip_margin=margin)
    he = HipEmitter(binding, h, binding.emit_params(mode, lazyDecompressionSupport=int(lazy)))
    oe = _orc.OracleEmitter(o, _orc.emit_params(mode, lazyDecompressionSupport=int(lazy)))
    pol = _driver.Policy(mode)
    a = _driver.encode_sequential(h, he, files, pol, lazy=lazy)
    b = _driver.encode_sequential(o, oe, files, pol, lazy=lazy)
    loaded = a["loaded"]
    assert loaded == b["loaded"]
    # a contig that lies across a separator of the reference as it stands now, its middle (the separator's surroundings) diverged
    ref = h.ref(h.loaded_ref_length() + 1, 0)
    seps = np.nonzero(ref[1:] == 0)[0] + 1
    sep = int(seps[len(seps) // 2]) if seps.size else int(h.loaded_ref_length() // 2)      # (no separators in the buffer without lazy decompression)
    q = ref[sep - 3_000: sep + 3_000].copy()
    q[3_000 - 200: 3_000 + 200] = synth.ACGT[rng.integers(0, 4, 400)]
    mh, mo = h.match(q, 32, NO_LOCK), o.match(q, 32, NO_LOCK)
    assert np.array_equal(np.asarray(mh), np.asarray(mo)) and len(mo) >= 2
    assert he.process(mh, q, NO_LOCK, 128, 0, 0, loaded) == oe.process(mo, q, NO_LOCK, 128, 0, 0, loaded)
    compare(he.streams(), oe.streams())
    oc = oe.counters()
    for k in he.counters:
        assert he.counters[k] == oc[k], k
    assert len(oe.streams()["flags"]) > 140_000                      # (the long gaps are in the flags, a byte each)


@pytest.mark.parametrize("bonus,penalty,threshold,initial,mode", [(50, 50, 500, 125, 1), (10, 70, 300, 0, 1), (100, 30, 90, 60, 1), (1, 1, 4, 0, 0),
                                                                  (50, 50, 500, 500, 1), (0, 25, 2000, 10, 1)])
def test_extensions_across_divergent_stretches_with_other_scores(binding, bonus, penalty, threshold, initial, mode):
    """genomes a few percent apart: the extensions with mismatches run from one exact match to the next, hundreds of bytes to
    kilobytes — on the device the whole wave walks such a stretch, the score automaton (MBGC_Encoder.cpp:346-365, 404-421) as a
    prefix composition. The reference's scores (MBGC_Params.h:92-97) and others: a threshold reached early, late, at once, never."""
    rng = np.random.default_rng(bonus * 7 + penalty)
    base = synth.ACGT[rng.integers(0, 4, 300_000)]

    def drift(g, rate, burst):
        g = g.copy()
        m = rng.random(g.size) < rate
        for a in rng.integers(0, g.size - 4_000, 12):                      # stretches that diverge more than the rest
            m[a:a + 3_000] |= rng.random(3_000) < burst
        g[m] = synth.ACGT[rng.integers(0, 4, int(m.sum()))]
        return g

    files = [[base], [drift(base, 0.04, 0.25)], [drift(base, 0.07, 0.4)[:200_000], drift(base, 0.02, 0.15)[200_000:]], [drift(base, 0.10, 0.3)]]
    over = dict(mmsMatchBonus=bonus, mmsMismatchPenalty=penalty, mmsMismatchesScoreThreshold=threshold, mmsMismatchesInitialScore=initial)
    lim, _ = _driver.ref_length_limit(len(files), base.size)
    h = binding.SlidingWindowSparseEMMatcher(lim)
    o = _orc.OracleMatcher(lim)
    he = HipEmitter(binding, h, binding.emit_params(mode, **over))
    oe = _orc.OracleEmitter(o, _orc.emit_params(mode, **over))
    a = _driver.encode_sequential(h, he, files)
    b = _driver.encode_sequential(o, oe, files)
    assert a["refExtSize"] == b["refExtSize"]
    compare(he.streams(), oe.streams())
    oc = oe.counters()
    for k in he.counters:
        assert he.counters[k] == oc[k], k
    assert oc["extensionsMatchedChars"] > 100_000 or initial >= threshold

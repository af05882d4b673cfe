"""Structural fuzz: genomes related by substitutions, indels, block moves, inversions (reverse
complements), tandem duplications, N runs and soft-masked stretches, cut into contigs of very different
sizes — HIP path vs oracle through both target loops, bit-exact on every stream."""
import numpy as np
import pytest

import _driver
import _orc
from mbgc_amd import synth
from test_gpu_emit import HipEmitter, compare

pytestmark = pytest.mark.gpu
_COMP = {ord("A"): ord("T"), ord("C"): ord("G"), ord("G"): ord("C"), ord("T"): ord("A")}


@pytest.fixture(scope="module")
def binding():
    from mbgc_amd import binding as b
    assert b.lib().swsem_device_count() > 0
    return b


def mutate(rng, g, sub=0.01, n_events=12):
    g = g.copy()
    m = rng.random(g.size) < sub
    g[m] = synth.ACGT[rng.integers(0, 4, int(m.sum()))]
    for _ in range(n_events):
        kind = rng.integers(0, 7)
        n = g.size
        a = int(rng.integers(0, max(1, n - 2000)))
        ln = int(rng.integers(1, 1500))
        if kind == 0:      # deletion
            g = np.concatenate([g[:a], g[a + ln:]])
        elif kind == 1:    # insertion of random sequence
            g = np.concatenate([g[:a], synth.ACGT[rng.integers(0, 4, ln)], g[a:]])
        elif kind == 2:    # tandem duplication
            g = np.concatenate([g[:a + ln], g[a:a + ln], g[a + ln:]])
        elif kind == 3:    # inversion
            seg = g[a:a + ln][::-1].copy()
            seg = np.array([_COMP.get(int(x), int(x)) for x in seg], dtype=np.uint8)
            g = np.concatenate([g[:a], seg, g[a + ln:]])
        elif kind == 4:    # block move
            seg = g[a:a + ln].copy()
            rest = np.concatenate([g[:a], g[a + ln:]])
            b = int(rng.integers(0, rest.size))
            g = np.concatenate([rest[:b], seg, rest[b:]])
        elif kind == 5:    # N run
            g[a:a + min(ln, 300)] = ord("N")
        else:              # soft-masked (lower case) stretch
            g[a:a + ln] = np.char.lower(g[a:a + ln].view("S1")).view(np.uint8) if False else (g[a:a + ln] | 0x20)
    return g


def cut(rng, g, k):
    if k <= 1:
        return [g]
    cuts = np.sort(rng.integers(1, g.size - 1, k - 1))
    parts = np.split(g, cuts)
    return [p for p in parts if p.size]


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_fuzz_sequential(binding, seed):
    rng = np.random.default_rng(seed)
    base = synth.ACGT[rng.integers(0, 4, 60_000)]
    files = [cut(rng, mutate(rng, base, 0.005 * (1 + i % 3)), int(rng.integers(1, 6))) for i in range(6)]
    lim, _ = _driver.ref_length_limit(len(files), 60_000)
    h = binding.SlidingWindowSparseEMMatcher(lim)
    o = _orc.OracleMatcher(lim)
    he, oe = HipEmitter(binding, h), _orc.OracleEmitter(o)
    a = _driver.encode_sequential(h, he, files)
    b = _driver.encode_sequential(o, oe, files)
    for x, y in zip(a["matches"], b["matches"]):
        assert np.array_equal(x, y)
    compare(he.streams(), oe.streams())
    assert np.array_equal(h.ht(), o.ht())


# (1109, 4, 350000): the buffer wraps inside a round with retries, the first piece after the wrap is sampled off the grid
# and holds an N run — its entries send all-N lookups one byte to the left, where the bytes are equal (k_insert: fp_at)
@pytest.mark.parametrize("seed,round_size,lim", [(11, 3, 2_000_000), (12, 5, 350_000), (13, 2, 250_000), (1109, 4, 350_000)])
def test_fuzz_rounds(binding, seed, round_size, lim):
    rng = np.random.default_rng(seed)
    base = synth.ACGT[rng.integers(0, 4, 50_000)]
    gs = [cut(rng, mutate(rng, base, 0.004 * (1 + i % 4)), int(rng.integers(1, 5))) for i in range(9)]
    h = binding.SlidingWindowSparseEMMatcher(lim)
    o = _orc.OracleMatcher(lim)
    a = _driver.encode_rounds(h, lambda: HipEmitter(binding, h), gs[0], gs[1:], round_size)
    b = _driver.encode_rounds(o, lambda: _orc.OracleEmitter(o), gs[0], gs[1:], round_size)
    assert a["unmatched"] == b["unmatched"] and a["locks"] == b["locks"] and a["refExtSize"] == b["refExtSize"]
    compare(a["streams"], b["streams"])
    assert np.array_equal(h.ht(), o.ht())


def test_many_tiny_contigs_in_one_batch(binding):
    """hundreds of contigs from 1 base to a few kb in a single device-resident round"""
    import torch
    rng = np.random.default_rng(21)
    base = synth.ACGT[rng.integers(0, 4, 200_000)]
    h = binding.SlidingWindowSparseEMMatcher(4_000_000)
    o = _orc.OracleMatcher(4_000_000)
    for m in (h, o):
        m.load_ref(base, load_rc=True)
    mut = mutate(rng, base, 0.01, 20)
    sizes = [1, 2, 27, 28, 31, 32, 33, 64] + [int(x) for x in rng.integers(1, 4000, 300)]
    contigs, p = [], 0
    for s in sizes:
        contigs.append(mut[p:p + s].copy())
        p = (p + s) % (mut.size - 5000)
    offs = np.zeros(len(contigs) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([c.size for c in contigs])
    buf = torch.from_numpy(np.concatenate(contigs)).to("cuda:0")
    torch.cuda.synchronize()
    h.match_batch_dev(buf.data_ptr(), offs, 32, None)
    counts = h.batch_counts()
    loaded = [h.loading_position()]
    h.emit_batch(binding.emit_params(1), loaded=loaded, n=len(contigs))
    for i, c in enumerate(contigs):
        exp = o.match(c)
        assert counts[i] == len(exp), i
        if len(exp):
            assert np.array_equal(h.batch_matches(i, counts[i]), exp), i
        oe = _orc.OracleEmitter(o)
        un = oe.process(exp, c, _orc.NO_LOCK, 128, 0, 0, loaded)
        got_un, streams, _ = h.emit_result(i)
        assert got_un == un, i
        compare(streams, oe.streams())


def test_batch_beyond_65535_contigs(binding):
    """more contigs in one round than a grid's y extent: the emission kernels index blocks through a
    chunk-owner table, so ragged batches of any count launch one block per 256 rows and nothing else"""
    import torch
    rng = np.random.default_rng(33)
    base = synth.ACGT[rng.integers(0, 4, 300_000)]
    h = binding.SlidingWindowSparseEMMatcher(2_000_000)
    o = _orc.OracleMatcher(2_000_000)
    for m in (h, o):
        m.load_ref(base)
    mut = mutate(rng, base, 0.01, 10)
    n = 70_000
    sizes = rng.integers(20, 120, n)
    sizes[::5000] = 40_000                                  # a few long ones among the crumbs
    starts = rng.integers(0, mut.size - 40_000, n)
    contigs = [mut[a:a + s] for a, s in zip(starts, sizes)]
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(sizes)
    buf = torch.from_numpy(np.concatenate(contigs)).to("cuda:0")
    torch.cuda.synchronize()
    h.match_batch_dev(buf.data_ptr(), offs, 32, None)
    counts = h.batch_counts()
    loaded = [h.loading_position()]
    h.emit_batch(binding.emit_params(1), loaded=loaded, n=n)
    un = h.emit_unmatched(n)
    sizes6, total = h.emit_pack_sizes(n)
    exp_total = 0
    for i in list(range(0, n, 97)) + list(range(0, n, 5000)) + [n - 1]:
        exp = o.match(contigs[i])
        assert counts[i] == len(exp), i
        oe = _orc.OracleEmitter(o)
        eu = oe.process(exp, contigs[i], _orc.NO_LOCK, 128, 0, 0, loaded)
        assert un[i] == eu, i
        got_un, streams, _ = h.emit_result(i)
        compare(streams, oe.streams())
        assert [len(oe.stream(k)) for k in range(6)] == [int(x) for x in sizes6[i]], i
    assert total == int(sizes6.sum())


def genomes_with_runs(rng, length, nfiles):
    """related genomes whose common base holds runs of one letter (N runs among them) and of short periods"""
    base = synth.ACGT[rng.integers(0, 4, length)]
    for _ in range(int(rng.integers(2, 8))):
        a, ln = int(rng.integers(0, length - 700)), int(rng.integers(30, 600))
        per = int(rng.choice([1, 1, 1, 2, 3, 16]))
        unit = synth.ACGT[rng.integers(0, 4, per)] if rng.random() < 0.7 else np.full(per, ord("N"), dtype=np.uint8)
        base[a:a + ln] = np.resize(unit, ln)
    return [cut(rng, mutate(rng, base, 0.004 * (1 + i % 4)), int(rng.integers(1, 5))) for i in range(nfiles)]


FUZZ_ROUND = 4        # the build round these seeds belong to: every round fuzzes cases no round before it has seen


def _extra():
    import os
    return int(os.environ.get("MBGC_FUZZ_EXTRA", "100"))


def _seeds(first):
    """MBGC_FUZZ_EXTRA seeds (default 100) from `first` on, moved by the round number (MBGC_FUZZ_ROUND overrides it: any earlier
    round's cases can be run again); printed, so that a failure names its case"""
    import os
    rnd = int(os.environ.get("MBGC_FUZZ_ROUND", FUZZ_ROUND))
    lo = first + 100_000 * rnd
    print("fuzz seeds %d..%d (round %d)" % (lo, lo + _extra() - 1, rnd))
    return range(lo, lo + _extra())


def test_fuzz_small_buffers_and_runs_of_one_letter(binding):
    """MBGC_FUZZ_EXTRA cases (default 100, seeds of their own every round; 1000 ran clean at the end of round 2): buffers of 1.5 to
    20 genome lengths — most wrap several times, inside rounds with retries too —, runs of one letter, both target loops of the
    plain API"""
    for seed in _seeds(5000):
        rng = np.random.default_rng(seed)
        length = int(rng.choice([30_000, 50_000, 80_000]))
        gs = genomes_with_runs(rng, length, int(rng.integers(6, 14)))
        lim = int(rng.choice([3, 4, 6, 9, 40]) * length // 2 + rng.integers(0, 5000))
        rs = int(rng.integers(1, 8))
        h, o = binding.SlidingWindowSparseEMMatcher(lim), _orc.OracleMatcher(lim)
        a = _driver.encode_rounds(h, lambda: HipEmitter(binding, h), gs[0], gs[1:], rs)
        b = _driver.encode_rounds(o, lambda: _orc.OracleEmitter(o), gs[0], gs[1:], rs)
        assert a["unmatched"] == b["unmatched"] and a["locks"] == b["locks"] and a["refExtSize"] == b["refExtSize"], (seed, lim, rs)
        compare(a["streams"], b["streams"])
        assert np.array_equal(h.ht(), o.ht()), (seed, lim, rs)
        h.close(), o.close()
        h, o = binding.SlidingWindowSparseEMMatcher(lim), _orc.OracleMatcher(lim)
        he, oe = HipEmitter(binding, h), _orc.OracleEmitter(o)
        a, b = _driver.encode_sequential(h, he, gs), _driver.encode_sequential(o, oe, gs)
        for x, y in zip(a["matches"], b["matches"]):
            assert np.array_equal(x, y), (seed, lim)
        compare(he.streams(), oe.streams())
        assert np.array_equal(h.ht(), o.ht()), (seed, lim)
        h.close(), o.close()


def test_fuzz_the_pipelined_round_runner(binding):
    """the same kind of input through RoundRunner (batch calls, speculative finalize, second phase kept back behind the next
    resolve launch), targets of several contigs"""
    import torch
    from mbgc_amd.rounds import RoundRunner, round_schedule
    for seed in _seeds(7000):
        rng = np.random.default_rng(seed)
        length = int(rng.choice([30_000, 60_000]))
        gs = genomes_with_runs(rng, length, int(rng.integers(8, 20)))
        lim = int(rng.choice([3, 4, 6, 9, 40]) * length // 2 + rng.integers(0, 5000))
        rs = int(rng.integers(1, 7))
        h = binding.SlidingWindowSparseEMMatcher(lim)
        h.set_sliding_window_size(16)
        h.load_ref(np.concatenate(gs[0]), load_rc=True)
        runner = RoundRunner(h, 0, 1, None, "cuda:0", lazy=True, emit_params=binding.emit_params(1))
        runner.start()
        for rnd in round_schedule(len(gs) - 1, rs, 1):
            contigs = [c for t in rnd[0] for c in gs[1 + t]]
            tgt = [k for k, t in enumerate(rnd[0]) for _ in gs[1 + t]]
            buf = torch.from_numpy(np.concatenate(contigs)).to("cuda:0")
            offs = np.zeros(len(contigs) + 1, dtype=np.uint64)
            offs[1:] = np.cumsum([c.size for c in contigs])
            torch.cuda.synchronize()
            runner.run_round(buf, offs, targets=tgt)
        runner.flush()
        o = _orc.OracleMatcher(lim)
        res = _driver.encode_rounds(o, lambda: _orc.OracleEmitter(o), gs[0], gs[1:], rs)
        for k, v in res["streams"].items():
            assert bytes(runner.streams[k]) == v, (seed, lim, rs, k)
        assert bytes(runner.locks_stream) == res["locks"] and bytes(runner.ref_ext_sizes) == res["refExtSize"], (seed, lim, rs)
        assert np.array_equal(h.ht(), o.ht()), (seed, lim, rs)
        o.close(), h.close()

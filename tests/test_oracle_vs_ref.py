"""Pins the CPU oracle (oracle/*.c) against the reference itself (oracle/_ref, built from
/root/reference by oracle/Makefile). CPU-only; skipped where the reference build is absent."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

import _driver
import _orc
from mbgc_amd import synth

NO_LOCK = _orc.NO_LOCK


def small_collection(n, length, div=0.01, seed=7):
    base = synth.base_codes(length, seed)
    return [synth.genome(base, i, div) for i in range(n)]


def both(refh, max_len, **kw):
    return refh.RefMatcher(max_len, **kw), _orc.OracleMatcher(max_len, **kw)


def assert_same_state(r, o):
    assert r.loading_position() == o.loading_position()
    assert r.ref_length() == o.ref_length()
    assert r.loaded_ref_length() == o.loaded_ref_length()
    n = r.ref_length()
    assert np.array_equal(r.ref(n)[1:], o.ref(n)[1:])
    assert np.array_equal(r.ht(), o.ht())


def test_hash_kats():
    # SURVEY.md §8a row 2 (measured on the reference)
    assert _orc.orc_hash(b"A" * 28) == 0xd28afb14
    assert _orc.orc_hash(b"ACGT" * 7) == 0x6bdfbd14
    assert _orc.orc_hash(b"TGCCGCGGATTGGATAAAACACTGTAAA") == 0x48cf2a28


def test_params(refh):
    for L in (24, 32, 40, 48, 64):
        for lim in (1 << 20, 300_000_000):
            r, o = both(refh, lim, L=L)
            assert r.K() == o.K() and r.hash_size() == o.hash_size()
            r.close(); o.close()


def test_revcomp_lut_and_mismatch_table(refh):
    allb = np.arange(1, 256, dtype=np.uint8)
    text = np.concatenate([allb, np.frombuffer(b"ACGTNacgtn" * 10, dtype=np.uint8)])
    r, o = both(refh, 1 << 20)
    for m in (r, o):
        m.load_ref(text, load_rc=True, add_sep=False)
    assert_same_state(r, o)
    # table of coders/ContextAwareMismatchesCoder.h:13-17 as measured in SURVEY.md §8a row 9
    rows = {"A": [None, 2, 0, 1, 3], "C": [1, None, 2, 0, 3], "G": [0, 2, None, 1, 3], "T": [1, 0, 2, None, 3],
            "N": [1, 2, 3, 0, None]}
    for a, row in rows.items():
        for b, code in zip("ACGTN", row):
            got = _orc.lib().orc_mismatch2code(ord(a), ord(b))
            assert got == (ord(b) if code is None else code)
    assert _orc.lib().orc_mismatch2code(ord("a"), ord("C")) == ord("C")
    assert _orc.lib().orc_mismatch2code(0, ord("C")) == ord("C")


@pytest.mark.parametrize("sizes", [(10275,), (10027,), (27,), (28,), (29,), (44,), (2075, 2076, 1), (4125, 17, 2049, 90000)])
def test_insert_tail_quirk(refh, sizes):
    """processIgnoreCollisionsRef's main/tail sample sets."""
    rng = np.random.default_rng(sum(sizes))
    r, o = both(refh, 1 << 20)
    for s in sizes:
        t = synth.ACGT[rng.integers(0, 4, s)]
        for m in (r, o):
            m.load_ref(t, load_rc=False, add_sep=True)
        assert_same_state(r, o)
    q = o.ref()[1:]
    assert np.array_equal(r.match(q), o.match(q))


@pytest.mark.parametrize("sequential", [True, False])
def test_wrap_quirk_and_circular_buffer(refh, sequential):
    """Circular wrap: samplingPos reset to REF_SHIFT, clipping at swEnd, separators, dropped remainders."""
    rng = np.random.default_rng(5)
    base = rng.integers(0, 4, 60_000)
    r, o = both(refh, 100_000)
    if sequential:
        r.disable_sliding_window(); o.disable_sliding_window()
    for step in range(14):
        g = base.copy()
        mask = rng.random(g.size) < 0.02
        g[mask] = (g[mask] + 1) & 3
        g = synth.ACGT[g][: int(rng.integers(20_000, 60_000))]
        lock = NO_LOCK
        if not sequential:
            lr, lo = r.acquire_lock(), o.acquire_lock()
            assert lr == lo
            lock = lo
        mr, mo = r.match(g, 32, lock), o.match(g, 32, lock)
        assert np.array_equal(mr, mo), step
        rc = bool(step % 3 == 0)
        for m in (r, o):
            m.load_ref(g, load_rc=rc, add_sep=True)
            if step % 2:
                m.load_separator(0)
        if not sequential:
            r.release_lock(lock); assert o.release_lock(lock) == 0
        assert_same_state(r, o)
    assert o.loaded_ref_length() > 99_999 + 50_000  # wrapped


def test_locks_deque(refh):
    r, o = both(refh, 100_000)
    t = synth.ACGT[np.random.default_rng(1).integers(0, 4, 30_000)]
    held = []
    for i in range(12):
        a, b = r.acquire_lock(), o.acquire_lock()
        assert a == b
        held.append(a)
        for m in (r, o):
            m.load_ref(t, load_rc=False, add_sep=True)
        assert_same_state(r, o)
        if i % 3 == 2:
            for v in (held.pop(1), held.pop(0)):       # out-of-order then head release
                r.release_lock(v); assert o.release_lock(v) == 0
            assert_same_state(r, o)
    assert o.release_lock(12345) == -1


def test_prefilter_is_result_neutral():
    gs = small_collection(6, 100_000, 0.02)
    a, b = _orc.OracleMatcher(4_000_000), _orc.OracleMatcher(4_000_000)
    b.set_prefilter(False)
    for m in (a, b):
        m.load_ref(gs[0], load_rc=True)
    for g in gs[1:]:
        # ragged query ends exercise the stale l2/r2 values of .cpp:203-204
        for cut in (g.size, 31, 28, 57):
            assert np.array_equal(a.match(g[:cut]), b.match(g[:cut]))
        for m in (a, b):
            m.load_ref(g)


class RefEmitAdapter:
    """Gives the reference encoder harness the emitter interface of _driver."""

    def __init__(self, refh, matcher, n_targets=1, mode=1, lazy=True, bit40=False):
        self.e = refh.RefEmitter(matcher, mode=mode, lazy=lazy, bit40=bit40, n_targets=n_targets)
        self.pushed = 1
        self.extra = {}

    def process(self, m, contig, lock, factor, processed, t, loaded):
        for v in loaded[self.pushed:]:
            self.e.push_loaded_pos(int(v))
        self.pushed = len(loaded)
        self.e.set_processed(processed)
        return self.e.process(m, contig, t, lock)

    def view(self, t):
        ad = self

        class V:
            def process(s, m, contig, lock, factor, processed, tt, loaded): return ad.process(m, contig, lock, factor, processed, t, loaded)
            def put(s, which, data): ad.extra.setdefault((t, which), b""); ad.extra[(t, which)] += data  # noqa
            def streams(s):
                # separators were appended by the driver after all of the target's emissions, so
                # they belong at the end of the target's stream
                out = ad.e.streams(t)
                names = ("literals", "mapOff", "mapOff5th", "mapLen", "gapDelta", "flags")
                return {n: out[n] for n in names}
        return V()


@pytest.mark.parametrize("mode", [1, 2])
def test_emission_sequential(refh, mode):
    gs = small_collection(5, 150_000, 0.01, seed=11)
    files = [[g[:70_000], g[70_000:]] for g in gs]
    lim, _ = _driver.ref_length_limit(len(files), 150_000)
    r, o = both(refh, lim, skip_margin=24 if mode >= 2 else 16)
    pol = _driver.Policy(mode)

    class RefSeq:
        def __init__(s): s.a = RefEmitAdapter(refh, r, 1, mode); s.v = s.a.view(0)
        def process(s, *a): return s.v.process(*a)
        def put(s, which, data):
            # processAfterSequence / processAfterTarget in the reference
            s.a.e.after_sequence(0) if which == 0 else s.a.e.after_target(0)
        def streams(s): return s.a.e.streams(0)

    re_, oe = RefSeq(), _orc.OracleEmitter(o, _orc.emit_params(mode))
    a = _driver.encode_sequential(r, re_, files, pol)
    b = _driver.encode_sequential(o, oe, files, pol)
    assert a["locks"] == b["locks"] and a["refExtSize"] == b["refExtSize"]
    for x, y in zip(a["matches"], b["matches"]):
        assert np.array_equal(x, y)
    sa, sb = re_.streams(), oe.streams()
    for k in sa:
        assert sa[k] == sb[k], k
    assert len(sb["flags"]) > 1000 and len(sb["gapDelta"]) > 100
    assert_same_state(r, o)


@pytest.mark.parametrize("round_size", [1, 3, 8])
def test_emission_rounds_with_locks_and_wrap(refh, round_size):
    gs = small_collection(17, 120_000, 0.015, seed=3)
    g0 = [gs[0][:50_000], gs[0][50_000:]]
    targets = [[g[:40_000], g[40_000:]] for g in gs[1:]]
    lim = 900_000   # forces two wraps over 16 targets (each extends by >= 120 kB)
    r, o = both(refh, lim)
    ad = RefEmitAdapter(refh, r, n_targets=len(targets))
    cnt = {"t": 0}

    def make_ref():
        class E:
            def __init__(s): s.t = cnt["t"]; cnt["t"] += 1; s.v = ad.view(s.t)
            def process(s, *a): return s.v.process(*a)
            def put(s, which, data): ad.e.after_sequence(s.t) if which == 0 else ad.e.after_target(s.t)
            def streams(s): return ad.e.streams(s.t)
            def reset(s): ad.e.reset_target(s.t); return s
        return E()

    a = _driver.encode_rounds(r, make_ref, g0, targets, round_size)
    b = _driver.encode_rounds(o, lambda: _orc.OracleEmitter(o), g0, targets, round_size)
    assert a["locks"] == b["locks"] and a["refExtSize"] == b["refExtSize"] and a["unmatched"] == b["unmatched"]
    for x, y in zip(a["matches"], b["matches"]):
        assert np.array_equal(x, y)
    for k in a["streams"]:
        assert a["streams"][k] == b["streams"][k], k
    assert_same_state(r, o)
    assert o.loaded_ref_length() > lim


@pytest.mark.parametrize("mode,round_size", [(0, 8), (1, 6), (2, 5)])
def test_rounds_that_go_on_in_units(refh, mode, round_size):
    """the round drive of tests/_driver.py on collections whose rounds hold stopped targets in front of, between and behind kept
    ones (two genomes of three 7 % from the rest), with the reference's own processMatches behind it — units of five (-m0), two
    (-m1) and one (-m2) stopped targets; what the oracle restates must be what the reference does under that drive"""
    base = synth.base_codes(60_000, 77)
    gs = [synth.genome(base, i, 0.004 if i % 3 == 1 else 0.07) for i in range(18)]
    targets = [[g[:25_000], g[25_000:]] for g in gs[1:]]
    lim = 3_000_000
    margin = 24 if mode >= 2 else 16
    r, o = both(refh, lim, skip_margin=margin)
    pol = _driver.Policy(mode)
    ad = RefEmitAdapter(refh, r, n_targets=len(targets), mode=mode)
    cnt = {"t": 0}
    resets = []

    def make_ref():
        class E:
            def __init__(s): s.t = cnt["t"]; cnt["t"] += 1; s.v = ad.view(s.t)
            def process(s, *a): return s.v.process(*a)
            def put(s, which, data): ad.e.after_sequence(s.t) if which == 0 else ad.e.after_target(s.t)
            def streams(s): return ad.e.streams(s.t)
            def reset(s): resets.append(s.t); ad.e.reset_target(s.t); return s
        return E()

    a = _driver.encode_rounds(r, make_ref, [gs[0]], targets, round_size, pol)
    b = _driver.encode_rounds(o, lambda: _orc.OracleEmitter(o, _orc.emit_params(mode)), [gs[0]], targets, round_size, pol)
    assert a["locks"] == b["locks"] and a["refExtSize"] == b["refExtSize"] and a["unmatched"] == b["unmatched"]
    for x, y in zip(a["matches"], b["matches"]):
        assert np.array_equal(x, y)
    for k in a["streams"]:
        assert a["streams"][k] == b["streams"][k], k
    assert_same_state(r, o)
    assert len(resets) >= 3 and len(resets) < len(targets), resets      # (some targets were matched again, some kept the first pass)


@pytest.mark.parametrize("round_size,laps", [(1, 5), (4, 2)])
def test_five_laps_of_the_buffer(refh, round_size, laps):
    """the collection of tests/test_gpu_laps.py (five laps of a 2.4 MB buffer with rounds of 1; with rounds of 4 the lock
    window clips most of every extension and the loader gets less far): the restatement against the
    reference itself — matches, streams, locks, extension sizes, table — so that what the device path is compared with
    there (its lap tags drop stale entries without a visit) is the reference's own result"""
    gs = small_collection(121, 100_000, 0.01, seed=71)
    lim = 2_400_000
    r, o = both(refh, lim)
    targets = [[g] for g in gs[1:]]
    ad = RefEmitAdapter(refh, r, n_targets=len(targets))
    cnt = {"t": 0}

    def make_ref():
        class E:
            def __init__(s): s.t = cnt["t"]; cnt["t"] += 1; s.v = ad.view(s.t)
            def process(s, *a): return s.v.process(*a)
            def put(s, which, data): ad.e.after_sequence(s.t) if which == 0 else ad.e.after_target(s.t)
            def streams(s): return ad.e.streams(s.t)
            def reset(s): ad.e.reset_target(s.t); return s
        return E()

    a = _driver.encode_rounds(r, make_ref, [gs[0]], targets, round_size)
    b = _driver.encode_rounds(o, lambda: _orc.OracleEmitter(o), [gs[0]], targets, round_size)
    assert a["locks"] == b["locks"] and a["refExtSize"] == b["refExtSize"] and a["unmatched"] == b["unmatched"]
    for x, y in zip(a["matches"], b["matches"]):
        assert np.array_equal(x, y)
    for k in a["streams"]:
        assert a["streams"][k] == b["streams"][k], k
    assert_same_state(r, o)
    assert o.loaded_ref_length() > laps * lim


LISTERIA = "/root/reference/example-scripts"


@pytest.mark.skipif(not os.path.isdir(LISTERIA), reason="reference example data not on this host")
def test_listeria_streams_equal_reference_cli_dumps(refh, tmp_path):
    """`mbgc c -t1` on the three bundled Listeria genomes: archive md5 of SURVEY.md §8c, and every raw
    stream dumped by the developer build (`v -D`) equals driver + oracle."""
    names = sorted(f for f in os.listdir(LISTERIA) if f.endswith(".fna"))
    listing = tmp_path / "seqlist.txt"
    listing.write_text("".join(os.path.join(LISTERIA, n) + "\n" for n in names))
    arch = tmp_path / "lm.mbgc"
    subprocess.check_call([refh.REF_MBGC_DEV, "c", "-t1", str(listing), str(arch)], stdout=subprocess.DEVNULL)
    assert hashlib.md5(arch.read_bytes()).hexdigest() == "79b8acfe0ded3f371e381c72b7d7c2bb"
    subprocess.check_call([refh.REF_MBGC_DEV, "v", "-t1", "-D", str(arch)], stdout=subprocess.DEVNULL,
                          stderr=subprocess.DEVNULL, cwd=str(tmp_path))
    dump = {i: (tmp_path / ("lm.mbgc_dump_%02d" % i)).read_bytes() for i in (13, 14, 15, 16, 17, 18, 19)}
    files = [_driver.parse_fasta(os.path.join(LISTERIA, n)) for n in names]
    g0len = files[0][0].size                # G0 = first contig only (MGMP.cpp:91-100), kept as literals
    fsize = os.path.getsize(os.path.join(LISTERIA, names[0]))
    lim, _ = _driver.ref_length_limit(len(files), fsize)   # sized by the FILE size (MGMP.cpp:109)
    o = _orc.OracleMatcher(lim)
    oe = _orc.OracleEmitter(o)
    res = _driver.encode_sequential(o, oe, files)
    s = oe.streams()
    assert sum(len(m) for m in res["matches"]) == 29731
    assert dump[13][:g0len] == files[0][0].tobytes() and dump[13][g0len] == 0xA2   # processG0RefContig
    assert s["literals"] == dump[13][g0len + 1:]
    assert res["locks"] == dump[14]
    assert s["gapDelta"] == dump[15]
    assert s["flags"] == dump[16]
    assert s["mapOff"] == dump[17]
    assert s["mapLen"] == dump[18]
    assert res["refExtSize"] == dump[19]
    assert hashlib.md5(dump[15]).hexdigest().startswith("3731a2eb")
    assert hashlib.md5(dump[16]).hexdigest().startswith("14508a05")


def _fuzz_genomes(seed, length=50_000, nfiles=9):
    """tests/test_gpu_fuzz.py's generator without its GPU imports (test_fuzz_rounds, seed 1109: an N run in the piece that is
    loaded right after a wrap, inside a round with retries)"""
    src = open(os.path.join(os.path.dirname(__file__), "test_gpu_fuzz.py")).read()
    code = src[src.index("_COMP ="):src.index('@pytest.mark.parametrize("seed", [1, 2, 3, 4])')]
    code = code[:code.index("@pytest.fixture")] + code[code.index("def mutate"):]
    from mbgc_amd import synth
    g = {"np": np, "synth": synth}
    exec(code, g)
    rng = np.random.default_rng(seed)
    base = synth.ACGT[rng.integers(0, 4, length)]
    return [g["cut"](rng, g["mutate"](rng, base, 0.004 * (1 + i % 4)), int(rng.integers(1, 5))) for i in range(nfiles)]


def test_the_fuzz_case_with_an_n_run_behind_a_wrap(refh):
    """what the device path was wrong about (fingerprints of samples off the grid): the matches into the N run of
    the piece loaded right after the wrap ARE the reference's — restatement against the reference itself"""
    gs = _fuzz_genomes(1109)
    lim, round_size = 350_000, 4
    r, o = both(refh, lim)
    ad = RefEmitAdapter(refh, r, n_targets=len(gs) - 1)
    cnt = {"t": 0}

    def make_ref():
        class E:
            def __init__(s): s.t = cnt["t"]; cnt["t"] += 1; s.v = ad.view(s.t)
            def process(s, *a): return s.v.process(*a)
            def put(s, which, data): ad.e.after_sequence(s.t) if which == 0 else ad.e.after_target(s.t)
            def streams(s): return ad.e.streams(s.t)
            def reset(s): ad.e.reset_target(s.t); return s
        return E()

    a = _driver.encode_rounds(r, make_ref, gs[0], gs[1:], round_size)
    b = _driver.encode_rounds(o, lambda: _orc.OracleEmitter(o), gs[0], gs[1:], round_size)
    assert a["locks"] == b["locks"] and a["refExtSize"] == b["refExtSize"] and a["unmatched"] == b["unmatched"]
    for x, y in zip(a["matches"], b["matches"]):
        assert np.array_equal(x, y)
    for k in a["streams"]:
        assert a["streams"][k] == b["streams"][k], k
    assert_same_state(r, o)
    assert any(len(m) and (np.asarray(m)[:, 0] == 8276).any() for m in b["matches"])      # the matches in question


@pytest.mark.parametrize("k1,L,sequential", [(15, 32, True), (7, 32, False), (9, 40, False), (5, 32, True)])
def test_an_odd_sampling_step_takes_the_identity_encoded_matcher(refh, k1, L, sequential):
    """`mbgc -s <odd k1>`: MGMP.cpp:170-176 builds the base class SlidingWindowSparseEMMatcher — table entries are positions as they
    are (htEncodePos / htDecodePos the identity, .h:74-76), sampling starts at REF_SHIFT (.h:78). The restatement follows (k1ord =
    0): match rows, table image and loader state equal the reference's through several wraps, with locks, separators and
    reverse-complement loads (the reference side is built by refm_create as MGMP builds it)"""
    rng = np.random.default_rng(50 + k1)
    base = rng.integers(0, 4, 50_000)
    r, o = refh.RefMatcher(90_000, L=L, k1=k1), _orc.OracleMatcher(90_000, L=L, k1=k1)
    assert r.K() == o.K() and r.hash_size() == o.hash_size()
    if sequential:
        r.disable_sliding_window(); o.disable_sliding_window()
    nm = 0
    for step in range(12):
        g = base.copy()
        mask = rng.random(g.size) < 0.02
        g[mask] = (g[mask] + 1) & 3
        g = synth.ACGT[g][: int(rng.integers(15_000, 50_000))]
        lock = NO_LOCK
        if not sequential:
            lr, lo = r.acquire_lock(), o.acquire_lock()
            assert lr == lo
            lock = lo
        mr, mo = r.match(g, L, lock), o.match(g, L, lock)
        assert np.array_equal(mr, mo), step
        nm += len(mo)
        for m in (r, o):
            m.load_ref(g, load_rc=bool(step % 4 == 0), add_sep=True)
            if step % 2:
                m.load_separator(0)
        if not sequential:
            r.release_lock(lock); assert o.release_lock(lock) == 0
        assert_same_state(r, o)
    assert o.loaded_ref_length() > 89_999 + 40_000 and nm > 500      # wrapped (with locks the window clips what a round loads)

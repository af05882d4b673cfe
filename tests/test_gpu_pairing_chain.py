"""The emission's pairing chain (MBGC_Encoder.cpp:229-278) runs as speculative blocks of 16 matches, one lane each
(k_emit_meta_spec), accepted or replayed by k_emit_meta_stitch. On ordinary data every block is accepted and the
lanes never meet an inherited region boundary that is not the match's own, so the other paths get tests of their own:
blocks that fail (no warm-up: SWSEM_META_WARM=0), boundaries inherited across source regions (a diagonal that crosses
the loading position of a wrapped buffer), and more of those at once than a lane keeps (the block is given up and
replayed). Streams against the oracle, byte for byte; the counters say the paths were taken."""
import numpy as np
import pytest

import _driver
import _orc
from mbgc_amd import synth
from test_gpu_emit import HipEmitter, compare, small_collection

pytestmark = pytest.mark.gpu
NO_LOCK = _orc.NO_LOCK


@pytest.fixture(scope="module")
def binding():
    from mbgc_amd import binding as b
    assert b.lib().swsem_device_count() > 0
    return b


@pytest.mark.parametrize("mode,lazy,warm", [(1, True, 0), (1, False, 0), (2, True, 16), (0, True, 0)])
def test_blocks_that_fail_are_replayed_from_the_true_state(binding, monkeypatch, mode, lazy, warm):
    monkeypatch.setenv("SWSEM_META_WARM", str(warm))
    gs = small_collection(5, 150_000, 0.01, seed=21)
    files = [[g[:70_000], g[70_000:]] for g in gs]
    lim, _ = _driver.ref_length_limit(len(files), 150_000)
    margin = 24 if mode >= 2 else 16
    h = binding.SlidingWindowSparseEMMatcher(lim, skip_margin=margin)
    o = _orc.OracleMatcher(lim, skip_margin=margin)
    he = HipEmitter(binding, h, binding.emit_params(mode, lazyDecompressionSupport=int(lazy)))
    oe = _orc.OracleEmitter(o, _orc.emit_params(mode, lazyDecompressionSupport=int(lazy)))
    pol = _driver.Policy(mode)
    _driver.encode_sequential(h, he, files, pol, lazy=lazy)
    _driver.encode_sequential(o, oe, files, pol, lazy=lazy)
    compare(he.streams(), oe.streams())
    st = h.emit_stats()
    assert st["blocks_not_accepted"] > (50 if warm == 0 else 0) and st["groups_replayed"] > (20 if warm == 0 else 0), st


def test_rounds_with_locks_and_wrap_without_warm_up(binding, monkeypatch):
    monkeypatch.setenv("SWSEM_META_WARM", "0")
    gs = small_collection(12, 120_000, 0.015, seed=5)
    g0 = [gs[0][:50_000], gs[0][50_000:]]
    targets = [[g[:40_000], g[40_000:]] for g in gs[1:]]
    lim = 900_000
    h = binding.SlidingWindowSparseEMMatcher(lim)
    o = _orc.OracleMatcher(lim)
    a = _driver.encode_rounds(h, lambda: HipEmitter(binding, h), g0, targets, 4)
    b = _driver.encode_rounds(o, lambda: _orc.OracleEmitter(o), g0, targets, 4)
    assert a["locks"] == b["locks"] and a["refExtSize"] == b["refExtSize"]
    compare(a["streams"], b["streams"])
    assert h.emit_stats()["groups_replayed"] > 100


def _wrapped_pair(binding, lim, contigs):
    """The same loads on both matchers, whole buffer open (no sliding window): -> handles, refExtLoadedPosArr"""
    h = binding.SlidingWindowSparseEMMatcher(lim)
    o = _orc.OracleMatcher(lim)
    loaded = None
    for m in (h, o):
        m.disable_sliding_window()
        m.load_ref(contigs[0], load_rc=False, add_sep=True, sep=0)
        ld = [m.loading_position()]
        for c in contigs[1:]:
            before = m.loaded_ref_length()
            m.load_ref(c, load_rc=False, add_sep=True, sep=0)
            m.load_separator(0)
            ld.append(ld[-1] + m.loaded_ref_length() - before)
        assert loaded is None or loaded == ld
        loaded = ld
    assert h.loaded_ref_length() > lim and h.loading_position() == o.loading_position()
    return h, o, loaded


def _diagonals_query(ref, pos1, ndiag, length, seg=48, first=200, step=150):
    """Segments of `seg` bytes taken in turn from `ndiag` diagonals of the buffer; diagonal x crosses the loading position
    at query offset first + step * x: in front of it the newest region, behind it bytes of the lap before."""
    q = np.zeros(length, dtype=np.uint8)
    for s in range(length // seg):
        x = s % ndiag
        d = pos1 - (first + step * x)
        q[s * seg:(s + 1) * seg] = ref[s * seg + d:(s + 1) * seg + d]
    return q[:(length // seg) * seg]


@pytest.mark.parametrize("ndiag,given_up", [(2, False), (7, True)])
def test_boundaries_inherited_across_source_regions(binding, ndiag, given_up):
    rng = np.random.default_rng(77)
    lim = 600_000
    contigs = [synth.ACGT[rng.integers(0, 4, 70_000)] for _ in range(13)]      # 910 kB through a 600 kB buffer: one wrap
    h, o, loaded = _wrapped_pair(binding, lim, contigs)
    pos1 = h.loading_position()
    assert 20_000 < pos1 < lim - 20_000
    ref = h.ref(lim, 0)
    assert np.array_equal(ref, o.ref(lim))
    q = _diagonals_query(ref, pos1, ndiag, 12_000)
    he = HipEmitter(binding, h, binding.emit_params(1, lazyDecompressionSupport=1))
    oe = _orc.OracleEmitter(o, _orc.emit_params(1, lazyDecompressionSupport=1))
    mh, mo = h.match(q, 32, NO_LOCK), o.match(q, 32, NO_LOCK)
    assert np.array_equal(np.asarray(mh), np.asarray(mo)) and len(mo) > 150
    ua = he.process(mh, q, NO_LOCK, 128, 0, 0, loaded)
    ub = oe.process(mo, q, NO_LOCK, 128, 0, 0, loaded)
    assert ua == ub
    compare(he.streams(), oe.streams())
    assert len(oe.streams()["gapDelta"]) > 100 and any(oe.streams()["gapDelta"])
    st = h.emit_stats()
    assert st["foreign_boundary_steps"] > 20, st
    # (a chain that carries a foreign boundary carries it on for good: a warm-up that starts behind the crossing cannot know it)
    assert (st["blocks_given_up"] > 0) == given_up and st["groups_replayed"] > 0, st


@pytest.mark.parametrize("depth,depth_mism,lazy,ext,warm", [(64, 2, 1, 1, 64), (33, 5, 1, 1, 0), (7, 1, 0, 1, 64), (1, 0, 1, 1, 0), (0, 2, 1, 1, 64),
                                                            (64, 2, 0, 0, 0), (64, 64, 1, 1, 8), (2, 2, 0, 1, 64)])
def test_gap_depths_other_than_the_presets(binding, monkeypatch, depth, depth_mism, lazy, ext, warm):
    """gapDepthOffsetEncoding / gapDepthMismatchesEncoding away from MBGC's 64 / 2 (MBGC_Params.h:51-52), with and without lazy
    decompression and extensions, on genomes with duplications, moves and inversions (several diagonals interleaved)"""
    from test_gpu_fuzz import mutate
    monkeypatch.setenv("SWSEM_META_WARM", str(warm))
    rng = np.random.default_rng(1000 * depth + 10 * depth_mism + lazy)
    base = synth.ACGT[rng.integers(0, 4, 120_000)]
    gs = [base] + [mutate(rng, base, sub=0.012, n_events=25) for _ in range(4)]
    files = [[g[: g.size // 3], g[g.size // 3:]] for g in gs]
    lim, _ = _driver.ref_length_limit(len(files), 120_000)
    over = dict(lazyDecompressionSupport=lazy, enableExtensionsWithMismatches=ext, gapDepthOffsetEncoding=depth, gapDepthMismatchesEncoding=depth_mism)
    h = binding.SlidingWindowSparseEMMatcher(lim)
    o = _orc.OracleMatcher(lim)
    he = HipEmitter(binding, h, binding.emit_params(1, **over))
    oe = _orc.OracleEmitter(o, _orc.emit_params(1, **over))
    _driver.encode_sequential(h, he, files, lazy=bool(lazy))
    _driver.encode_sequential(o, oe, files, lazy=bool(lazy))
    compare(he.streams(), oe.streams())
    assert len(oe.streams()["mapLen"]) > 2000

"""Round trip: the six streams processMatches emits for a contig, decoded by the decoder's automaton
(oracle/decode_oracle.c = MBGC_Decoder::decodeSequenceAndReturnUnmatchedChars, MBGC_Decoder.cpp:319-523) against
the reference buffer the encoder matched against, give the contig back — a property that does not depend on the
input size. The decoder restatement is pinned by applying it to streams of the reference's own encoder
(oracle/_ref, when it is loadable: the build container) and to the encoder oracle, which is itself pinned to the
reference; the GPU tests then apply it to the HIP path's streams (test_gpu_emit.py, test_gpu_fullsize.py)."""
import numpy as np
import pytest

import _orc
from mbgc_amd import synth


def small_collection(n, length, div, seed):
    base = synth.base_codes(length, seed)
    return [synth.genome(base, i, div) for i in range(n)]


def roundtrip_oracle(gs, mode, lazy, lim, cut=None, lock_second=False):
    p = _orc.emit_params(mode)
    p.lazyDecompressionSupport = int(lazy)
    o = _orc.OracleMatcher(lim, skip_margin=24 if mode >= 2 else 16)
    o.load_ref(gs[0], load_rc=True)
    loaded = [o.loaded_ref_length()]
    nm = 0
    for t, g in enumerate(gs[1:]):
        contigs = [g] if cut is None else [g[:cut], g[cut:]]
        for c in contigs:
            rows = o.match(c)
            em = _orc.OracleEmitter(o, p)
            un = em.process(rows, c, _orc.NO_LOCK, 128, t, t, loaded)
            assert un != _orc.SKIPPED
            back, un2 = _orc.decode_contig(o.ref(o.max_ref_length()), p, em.streams(), _orc.NO_LOCK)
            assert back.size == c.size and np.array_equal(back, c), (mode, lazy, t)
            assert un2 == (un & 0xFFFFFFFF)
            nm += len(rows)
            o.load_ref(c)
            if lazy:
                o.load_separator(0)
            loaded.append(o.loaded_ref_length())
    o.close()
    return nm


@pytest.mark.parametrize("mode,lazy", [(1, True), (1, False), (0, True), (2, True), (2, False)])
@pytest.mark.parametrize("div", [0.01, 0.001, 0.08])
def test_decoder_inverts_the_encoder_oracle(mode, lazy, div):
    gs = small_collection(5, 120_000, div, seed=int(div * 10000) + mode)
    nm = roundtrip_oracle(gs, mode, lazy, 4_000_000, cut=50_000)
    assert nm > (50 if div > 0.05 else 400)


def test_decoder_inverts_the_encoder_oracle_after_a_wrap():
    """circular buffer: later targets overwrite the oldest text, matches point into every lap"""
    gs = small_collection(9, 100_000, 0.01, seed=5)
    roundtrip_oracle(gs, 1, True, 600_000)


def test_decoder_rejects_truncated_streams():
    gs = small_collection(2, 60_000, 0.01, seed=3)
    p = _orc.emit_params(1)
    o = _orc.OracleMatcher(2_000_000)
    o.load_ref(gs[0], load_rc=True)
    em = _orc.OracleEmitter(o, p)
    em.process(o.match(gs[1]), gs[1], _orc.NO_LOCK, 128, 0, 0, [o.loaded_ref_length()])
    s = em.streams()
    for k in ("mapLen", "flags", "mapOff"):
        t = dict(s)
        t[k] = t[k][:-1]
        with pytest.raises(ValueError):
            _orc.decode_contig(o.ref(o.max_ref_length()), p, t, _orc.NO_LOCK)
    o.close()


@pytest.fixture(scope="module")
def refh():
    import _refh
    if not _refh.available():
        pytest.skip("oracle/_ref not built (the reference is only present in the build container)")
    _refh.lib()
    return _refh


@pytest.mark.parametrize("mode", [1, 2])
def test_decoder_inverts_the_reference_encoder(refh, mode):
    """pins the decoder restatement: streams written by the reference's own MBGC_Encoder::processMatches"""
    gs = small_collection(5, 120_000, 0.01, seed=21 + mode)
    lim = 4_000_000
    r = refh.RefMatcher(lim, skip_margin=24 if mode >= 2 else 16)
    p = _orc.emit_params(mode)
    r.load_ref(gs[0], load_rc=True)
    for t, g in enumerate(gs[1:]):
        for c in (g[:70_000], g[70_000:]):
            rows = r.match(c)
            em = refh.RefEmitter(r, mode=mode, lazy=True, n_targets=1)
            em.set_processed(t)
            em.push_loaded_pos(r.loading_position())
            un = em.process(rows, c, 0, _orc.NO_LOCK)
            back, un2 = _orc.decode_contig(r.ref(lim), p, em.streams(0), _orc.NO_LOCK)
            assert np.array_equal(back, c), (mode, t)
            assert un2 == (un & 0xFFFFFFFF)
            r.load_ref(c)
    r.close()

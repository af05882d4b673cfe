"""The -m3 reverse-complement pass over the literal stream: the C restatement
(oracle/rcmatch_oracle.c) against the reference's own SimpleSequenceMatcher::rcMatchSequence / CopMEMMatcher compiled into
oracle/_ref (build container), and the decoder's inverse as a size-independent property."""
import numpy as np
import pytest

import _orc
import _rcdata

CASES = _rcdata.cases()


@pytest.mark.parametrize("name", sorted(CASES))
def test_matches_and_rewrite_equal_reference(refh, name):
    s = CASES[name]
    if s.size >= 55:
        got, params, _ = _orc.rc_find_matches(s)
        want = refh.rc_find_matches(s)
        assert params[:3] == (40, 5, 3)                               # L = 55 -> K = 40, k1 = 5, k2 = 3
        assert got.shape == want.shape and np.array_equal(got, want), name
    a = _orc.rc_match_sequence(s)
    b = refh.rc_match_sequence(s)
    assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2], name
    if name not in ("tiny",):
        assert (len(a[1]) > 0) == (name != "short" or len(a[1]) > 0)


@pytest.mark.parametrize("target,min_len", [(55, 0xFFFFFFFF), (32, 0xFFFFFFFF), (80, 60), (120, 0xFFFFFFFF), (55, 44)])
def test_other_lengths_equal_reference(refh, target, min_len):
    s = _rcdata.literal_like(150_000, 11, copies=50, longest=1500)
    got, _, _ = _orc.rc_find_matches(s, target, min_len)
    assert np.array_equal(got, refh.rc_find_matches(s, target, min_len))
    assert _orc.rc_match_sequence(s, target, min_len)[:3] == refh.rc_match_sequence(s, target, min_len)


def restore(seq, map_off, map_len, org_len):
    """SimpleSequenceMatcher::restoreRCMatchedSequence, matching/SimpleSequenceMatcher.cpp:178-211 (the decoder's inverse)"""
    def byte_frugal(buf, at):
        v, base = 0, 1
        while True:
            y = buf[at]; at += 1
            v += base * (y % 128); base *= 128
            if y < 128:
                return v, at
    comp = _rcdata.COMP.copy()
    out = bytearray()
    min_len, lp = byte_frugal(map_len, 0)
    op = 0
    for b in seq:
        if b != 0xA4:                                                 # RC_MATCH_MARK = '$' + 128
            out.append(b)
            continue
        src = int.from_bytes(map_off[op:op + 4], "little"); op += 4
        ln, lp = byte_frugal(map_len, lp)
        ln += min_len
        piece = bytes(out[src:src + ln]) if src + ln <= len(out) else None
        assert piece is not None
        out += bytes(comp[np.frombuffer(piece, dtype=np.uint8)][::-1])
    assert len(out) == org_len
    return bytes(out)


@pytest.mark.parametrize("name", ["planted", "long_copies", "manyfold"])
def test_rewritten_stream_restores(name):
    s = CASES[name].copy()
    s[s == 0xA4] = ord("A")                                           # (the mark itself must not occur in the input)
    s[(s >= ord("a")) & (s <= ord("z"))] = ord("C")                   # upper-case input: the restore complements with the upper table
    s[s == ord("N")] = ord("G")
    seq, off, ln, st = _orc.rc_match_sequence(s)
    assert st[1] > 0 and len(seq) < s.size
    assert restore(seq, off, ln, s.size) == s.tobytes()

"""The `-m3` reverse-complement pass over the literal stream on the device (mbgc_amd/csrc/copmem.hip through the C ABI of
include/mbgc_copmem.h) against the oracle (oracle/rcmatch_oracle.c, pinned on the reference's SimpleSequenceMatcher /
CopMEMMatcher): the matches in push order, the rewritten stream and the two maps, byte for byte."""
import numpy as np
import pytest

import _orc
import _rcdata

pytestmark = pytest.mark.gpu
CASES = _rcdata.cases()


@pytest.fixture(scope="module")
def ssm():
    from mbgc_amd import copmem
    m = copmem.SimpleSequenceMatcher()
    yield m
    m.close()


@pytest.mark.parametrize("name", sorted(CASES))
def test_matches_and_rewrite_equal_oracle(ssm, name):
    s = CASES[name]
    if s.size >= 55:
        got, params = ssm.rc_matches(s)
        want, wparams, _ = _orc.rc_find_matches(s)
        assert params == wparams
        assert got.shape == want.shape and np.array_equal(got, want), name
    assert ssm.rc_match_sequence(s) == _orc.rc_match_sequence(s), name


@pytest.mark.parametrize("target,min_len", [(55, 0xFFFFFFFF), (32, 0xFFFFFFFF), (80, 60), (120, 0xFFFFFFFF), (55, 44)])
def test_other_lengths_equal_oracle(ssm, target, min_len):
    s = _rcdata.literal_like(150_000, 11, copies=50, longest=1500)
    got, params = ssm.rc_matches(s, target, min_len)
    want, wparams, _ = _orc.rc_find_matches(s, target, min_len)
    assert params == wparams and np.array_equal(got, want)
    assert ssm.rc_match_sequence(s, target, min_len) == _orc.rc_match_sequence(s, target, min_len)


def test_reference_error_cases(ssm):
    from mbgc_amd import binding
    s = _rcdata.literal_like(10_000, 12)
    with pytest.raises(binding.SwsemError):
        ssm.rc_matches(s, 55, 20)                                     # "Minimal matching length too short!" (CopMEMMatcher.cpp:76-79)
    rows, params = ssm.rc_matches(s, 80, 24)                          # K follows the minimal length down ((24/4 - 1) * 4 = 20): no error
    assert params[0] == 20
    assert ssm.rc_match_sequence(s[:40]) == (s[:40].tobytes(), b"", b"", (0, 0, 0))   # shorter than the target: no matcher, empty maps


def test_literal_stream_sized_input(ssm):
    """a 64 MB stream (the literals of a few thousand 5 Mbp genomes) with a contig-sized reverse-complement copy in it:
    a match that spans thousands of query blocks, carried from block to block"""
    n = 64_000_000
    rng = np.random.default_rng(21)
    s = _rcdata.ACGT[rng.integers(0, 4, n)].copy()
    s[40_000_000:42_500_000] = _rcdata.revcomp(s[3_000_000:5_500_000])            # 2.5 Mbp, reverse-complemented
    for k in range(300):
        a, b, ln = int(rng.integers(0, n - 5000)), int(rng.integers(0, n - 5000)), int(rng.integers(60, 4000))
        s[b:b + ln] = _rcdata.revcomp(s[a:a + ln])
    got, params = ssm.rc_matches(s)
    want, wparams, _ = _orc.rc_find_matches(s)
    assert params == wparams and np.array_equal(got, want)
    a, b = ssm.rc_match_sequence(s), _orc.rc_match_sequence(s)
    assert a == b and a[3][1] > 2_400_000

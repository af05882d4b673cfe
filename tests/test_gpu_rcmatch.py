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


@pytest.mark.parametrize("target,min_len", [(55, 0xFFFFFFFF), (32, 0xFFFFFFFF), (120, 0xFFFFFFFF), (80, 0xFFFFFFFF), (55, 80)])
def test_other_lengths_equal_oracle(ssm, target, min_len):
    s = _rcdata.literal_like(150_000, 11, copies=50, longest=1500)
    got, params = ssm.rc_matches(s, target, min_len)
    want, wparams, _ = _orc.rc_find_matches(s, target, min_len)
    assert params == wparams and np.array_equal(got, want)
    assert ssm.rc_match_sequence(s, target, min_len) == _orc.rc_match_sequence(s, target, min_len)


def test_a_minimum_below_the_target_length_is_refused(ssm):
    """(80, 60) and (55, 44) equal the oracle on this file's data and differ from it on random inputs with runs of one letter
    (the reference reports whichever shorter matches its sampling happens upon): MBGC never asks for it, the device says no"""
    from mbgc_amd import binding
    s = _rcdata.literal_like(20_000, 13)
    for target, min_len in ((80, 60), (55, 44)):
        with pytest.raises(binding.SwsemError):
            ssm.rc_matches(s, target, min_len)


def test_fuzz_default_minimum(ssm):
    """random streams with planted, mutated and palindromic reverse-complement copies and runs of one letter, target lengths
    55 and 32 with the default minimum: matches, rewritten stream and maps against the oracle"""
    for seed in range(15000, 15040):
        rng = np.random.default_rng(seed)
        n = int(rng.choice([3_000, 40_000, 150_000, 400_000]))
        s = _rcdata.literal_like(n, seed, copies=int(rng.integers(0, 80)), longest=int(rng.choice([200, 1500, 6000, 30000])),
                                 mutate=float(rng.choice([0.0, 0.0, 0.005, 0.02])))
        if rng.random() < 0.5:
            a = int(rng.integers(0, max(1, n - 500)))
            s[a:a + int(rng.integers(60, 400))] = _rcdata.ACGT[int(rng.integers(0, 4))]
        if rng.random() < 0.3 and n > 5000:
            a = int(rng.integers(0, n - 3000))
            s[a + 700:a + 1400] = _rcdata.revcomp(s[a:a + 700].copy())
        target = int(rng.choice([55, 32]))
        got, params = ssm.rc_matches(s, target)
        want, wparams, _ = _orc.rc_find_matches(s, target)
        assert params == wparams and got.shape == want.shape and np.array_equal(got, want), seed
        assert ssm.rc_match_sequence(s, target) == _orc.rc_match_sequence(s, target), seed


def test_reference_error_cases(ssm):
    from mbgc_amd import binding
    s = _rcdata.literal_like(10_000, 12)
    with pytest.raises(binding.SwsemError):
        ssm.rc_matches(s, 55, 20)                                     # "Minimal matching length too short!" (CopMEMMatcher.cpp:76-79)
    rows, params = ssm.rc_matches(s, 24)                              # K follows the length down ((24/4 - 1) * 4 = 20): no error
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

"""Golden vectors generated from the reference itself (tests/golden/make_golden.py): they pin the CPU
oracle everywhere (-m "not gpu") and the HIP path on the GPU box (-m gpu), where /root/reference is absent."""
import hashlib
import os

import numpy as np
import pytest

import _driver
import _orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["seq_m1", "seq_m2", "rounds3_wrap", "rounds4_divergent", "seq_bit40", "rounds3_bit40", "seq_k15", "rounds3_wrap_k9", "rounds6_mixed"]


def load(case):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_inputs", os.path.join(GOLDEN, "make_golden.py"))
    # only the input recipe is needed (no reference import on this host): restate it here
    from mbgc_amd import synth
    d = np.load(os.path.join(GOLDEN, case + ".npz"))
    n, length, div, seed, lim, cpt, rs, mode = d["case"][:8]
    n, length, seed, lim, cpt, rs, mode = int(n), int(length), int(seed), int(lim), int(cpt), int(rs), int(mode)
    base = synth.base_codes(length, seed)
    # (divergence -1: every third genome 7 % from the rest, the others 0.4 % — make_golden.py's rounds6_mixed)
    gs = [synth.genome(base, i, float(div) if div >= 0 else (0.07 if i % 3 == 2 else 0.004)) for i in range(n)]
    if not lim:
        lim, _ = _driver.ref_length_limit(n, length)
    return d, gs, lim, cpt, rs, mode


def extras(d):
    """(loader position to start from, 40-bit offsets) of the cases that have them"""
    c = d["case"]
    return (int(c[8]), bool(c[9])) if len(c) > 8 else (0, False)


def k1_of(case):
    """the sampling step of the case (an odd one: the reference's base matcher class with identity-encoded table entries)"""
    c = np.load(os.path.join(GOLDEN, case + ".npz"))["case"]
    return int(c[10]) if len(c) > 10 else 16


def split(g, k):
    cuts = [0] + [g.size * i // k + (7 * i) % 13 for i in range(1, k)] + [g.size]
    return [g[cuts[i]:cuts[i + 1]] for i in range(k)]


def ht_digest(ht):
    nz = np.nonzero(ht)[0].astype(np.uint64)
    h = hashlib.sha256()
    h.update(nz.tobytes())
    h.update(ht[nz].astype(np.uint32).tobytes())
    return h.hexdigest()


def run_case(case, matcher, make_emitter):
    d, gs, lim, cpt, rs, mode = load(case)
    pol = _driver.Policy(mode)
    start, bit40 = extras(d)
    if start:
        matcher.set_position(start, 0)
    if rs == 0:
        em = make_emitter()
        res = _driver.encode_sequential(matcher, em, [split(g, cpt) for g in gs], pol)
        streams = em.streams()
    else:
        res = _driver.encode_rounds(matcher, make_emitter, split(gs[0], cpt), [split(g, cpt) for g in gs[1:]], rs, pol)
        streams = res["streams"]
    assert np.array_equal(np.concatenate(res["matches"]).astype(np.uint64), d["matches"]), "match rows"
    assert [len(m) for m in res["matches"]] == list(d["match_counts"])
    for k, v in streams.items():
        assert v == d["stream_" + k].tobytes(), k
    assert res["locks"] == d["locks"].tobytes() and res["refExtSize"] == d["refExtSize"].tobytes()
    assert matcher.loaded_ref_length() == int(d["loaded_ref_length"])
    assert ht_digest(matcher.ht()) == str(d["ht_sha256"])
    if bit40:                                         # the fifth offset byte is in use: some match lies beyond 2^32
        assert len(streams["mapOff5th"]) > 0 and max(streams["mapOff5th"]) == 1 and int(d["matches"][:, 0].max()) >= 1 << 32


def margin(case):
    return 24 if int(np.load(os.path.join(GOLDEN, case + ".npz"))["case"][7]) >= 2 else 16


@pytest.mark.parametrize("case", CASES)
def test_oracle_reproduces_reference_fixtures(case):
    d, gs, lim, cpt, rs, mode = load(case)
    o = _orc.OracleMatcher(lim, k1=k1_of(case), skip_margin=margin(case))
    run_case(case, o, lambda: _orc.OracleEmitter(o, _orc.emit_params(mode, enable40bitReference=int(extras(d)[1]))))


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_hip_reproduces_reference_fixtures(case):
    from mbgc_amd import binding
    from test_gpu_emit import HipEmitter
    d, gs, lim, cpt, rs, mode = load(case)
    h = binding.SlidingWindowSparseEMMatcher(lim, k1=k1_of(case), skip_margin=margin(case))
    run_case(case, h, lambda: HipEmitter(binding, h, binding.emit_params(mode, enable40bitReference=int(extras(d)[1]))))


def test_listeria_fingerprints_recorded():
    """what `mbgc c -t1` on the reference's bundled Listeria set produced in the build container
    (checked live by tests/test_oracle_vs_ref.py when /root/reference is present)"""
    fp = {"archive_md5": "79b8acfe0ded3f371e381c72b7d7c2bb", "archive_bytes": 1016021, "exact_matches": 29731,
          "gapDelta_md5_prefix": "3731a2eb", "flags_md5_prefix": "14508a05"}
    assert len(fp["archive_md5"]) == 32

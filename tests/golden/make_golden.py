#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REFERENCE ITSELF (oracle/_ref/libswsem_ref.so, compiled from
/root/reference by oracle/Makefile). Run in the build container:  python tests/golden/make_golden.py

Fixtures are data only: the inputs are regenerated from seeds by mbgc_amd.synth (numpy PCG64), the
expected outputs are what the reference's SlidingWindowExpSparseEMMatcher and
MBGC_Encoder::processMatches produced for them. They travel to the GPU box, where /root/reference
does not exist, and pin both the oracle (-m "not gpu") and the HIP path (-m gpu)."""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import _driver  # noqa: E402
import _refh  # noqa: E402
from mbgc_amd import synth  # noqa: E402
from test_oracle_vs_ref import RefEmitAdapter  # noqa: E402

CASES = {
    # name: (genomes, length, divergence, seed, max_ref_len, contigs per target, round size (0 = sequential), mode)
    "seq_m1": (6, 60_000, 0.01, 101, 0, 2, 0, 1),
    "seq_m2": (5, 50_000, 0.02, 102, 0, 1, 0, 2),
    "rounds3_wrap": (11, 50_000, 0.015, 103, 400_000, 2, 3, 1),
    "rounds4_divergent": (9, 40_000, 0.06, 104, 2_000_000, 1, 4, 1),
    # 40-bit reference offsets (MGMP.cpp:159-166, MBGC_Encoder.cpp:229-234): a buffer longer than 2^32 bytes, the loader
    # stood just below 2^32 (setPosition, SlidingWindowSparseEMMatcher.h:110-113) so that the collection's later genomes
    # lie beyond it — their matches carry a fifth offset byte (mapOff5th). Two more fields: start position, 40-bit flag.
    "seq_bit40": (6, 200_000, 0.012, 105, (1 << 32) + 6_000_000, 2, 0, 1, (1 << 32) - 500_000, 1),
    "rounds3_bit40": (8, 150_000, 0.012, 106, (1 << 32) + 6_000_000, 1, 3, 1, (1 << 32) - 400_000, 1),
    # an ODD sampling step (`mbgc -s 15`): MGMP.cpp:170-176 then builds the base class SlidingWindowSparseEMMatcher, whose table
    # entries hold positions as they are (htEncodePos / htDecodePos the identity, .h:74-76) and whose sampling starts at
    # REF_SHIFT. One more field: k1. Sequential, and rounds on a buffer that wraps several times.
    "seq_k15": (6, 60_000, 0.012, 107, 0, 2, 0, 1, 0, 0, 15),
    "rounds3_wrap_k9": (12, 50_000, 0.015, 108, 380_000, 2, 3, 1, 0, 0, 9),
    # divergence -1: every third genome 7 % from the rest, the others 0.4 % — rounds of 6 in which the fifth target gives a contig
    # up as dissimilar (MGMP.cpp:382-388) and the sixth, behind it, does not (it keeps what its worker found running ahead)
    "rounds6_mixed": (13, 40_000, -1, 109, 2_000_000, 2, 6, 1),
}


def split(g, k):
    cuts = [0] + [g.size * i // k + (7 * i) % 13 for i in range(1, k)] + [g.size]
    return [g[cuts[i]:cuts[i + 1]] for i in range(k)]


def inputs(case):
    n, length, div, seed, lim, cpt, rs, mode = CASES[case][:8]
    base = synth.base_codes(length, seed)
    gs = [synth.genome(base, i, div if div >= 0 else (0.07 if i % 3 == 2 else 0.004)) for i in range(n)]
    if not lim:
        lim, _ = _driver.ref_length_limit(n, length)
    return gs, lim, cpt, rs, mode


def extras(case):
    c = CASES[case]
    return (int(c[8]), bool(c[9])) if len(c) > 8 else (0, False)


def k1_of(case):
    c = CASES[case]
    return int(c[10]) if len(c) > 10 else 16


def ht_digest(ht):
    nz = np.nonzero(ht)[0].astype(np.uint64)
    h = hashlib.sha256()
    h.update(nz.tobytes())
    h.update(ht[nz].astype(np.uint32).tobytes())
    return h.hexdigest()


def run_reference(case):
    gs, lim, cpt, rs, mode = inputs(case)
    margin = 24 if mode >= 2 else 16
    r = _refh.RefMatcher(lim, k1=k1_of(case), skip_margin=margin)
    start, bit40 = extras(case)
    if start:
        r.set_position(start, 0)
    pol = _driver.Policy(mode)
    if rs == 0:
        files = [split(g, cpt) for g in gs]
        ad = RefEmitAdapter(_refh, r, 1, mode, bit40=bit40)

        class E:
            def process(s, *a): return ad.view(0).process(*a)
            def put(s, which, data): ad.e.after_sequence(0) if which == 0 else ad.e.after_target(0)
            def streams(s): return ad.e.streams(0)
        em = E()
        res = _driver.encode_sequential(r, em, files, pol)
        streams = em.streams()
    else:
        targets = [split(g, cpt) for g in gs[1:]]
        ad = RefEmitAdapter(_refh, r, n_targets=len(targets), mode=mode, bit40=bit40)
        cnt = {"t": 0}

        def make():
            class E:
                def __init__(s): s.t = cnt["t"]; cnt["t"] += 1; s.v = ad.view(s.t)
                def process(s, *a): return s.v.process(*a)
                def put(s, which, data): ad.e.after_sequence(s.t) if which == 0 else ad.e.after_target(s.t)
                def streams(s): return ad.e.streams(s.t)
                def reset(s): ad.e.reset_target(s.t); return s
            return E()
        res = _driver.encode_rounds(r, make, split(gs[0], cpt), targets, rs, pol)
        streams = res["streams"]
    out = {"stream_" + k: np.frombuffer(v, dtype=np.uint8) for k, v in streams.items()}
    out["locks"] = np.frombuffer(res["locks"], dtype=np.uint8)
    out["refExtSize"] = np.frombuffer(res["refExtSize"], dtype=np.uint8)
    out["matches"] = np.concatenate(res["matches"]).astype(np.uint64)
    out["match_counts"] = np.array([len(m) for m in res["matches"]], dtype=np.uint64)
    out["ht_sha256"] = np.array(ht_digest(r.ht()))
    out["loaded_ref_length"] = np.array(r.loaded_ref_length(), dtype=np.uint64)
    out["case"] = np.array(list(CASES[case]), dtype=np.float64)      # (2^32-sized values are exact in float64)
    return out


if __name__ == "__main__":
    assert _refh.available(), "build oracle/_ref first (make -C oracle ref)"
    for case in (sys.argv[1:] or CASES):
        out = run_reference(case)
        np.savez_compressed(os.path.join(HERE, case + ".npz"), **out)
        print(case, "matches", len(out["matches"]), "literals", out["stream_literals"].size, "flags", out["stream_flags"].size)

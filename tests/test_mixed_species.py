"""BASELINE.json configs[4]'s data shape — mixed-species collections (mbgc_amd/synth.py: MixedSpecies) in the `-m3` max
mode — on the CPU: the generator's contract, and the oracle driven through the reference's sequential target loop with
the `-m3` presets against what the reference's own CLI writes for the same files (`mbgc-dev c -m3` + `v -D`, built by
oracle/Makefile from /root/reference; skipped where that build is absent). The `-m gpu` counterpart at full size is
tests/test_gpu_configs4.py."""
import os
import subprocess

import numpy as np
import pytest

import _driver
import _orc
import _refh
from mbgc_amd import synth


def test_generator_is_a_function_of_seed_and_index():
    a, b = synth.MixedSpecies(length=40_000), synth.MixedSpecies(length=40_000)
    for i in (0, 3, 8, 17, 40, 73):
        x, y = a.contigs(i), b.contigs(i)
        assert len(x) == len(y) and all(np.array_equal(p, q) for p, q in zip(x, y))
    assert len(a.contigs(0)) == 1                                        # G0: one contig (the reference, MGMP.cpp:91-100)
    assert {len(a.contigs(i)) for i in range(1, 40)} == {1, 2, 3, 4}
    # species share nothing, genomes of a species do: 28-mers of genome 9 (species 1) found in genome 1 but not in genome 0
    def kmers(seq): return {seq[p:p + 28].tobytes() for p in range(0, seq.size - 28, 7)}
    g0, g1, g9 = (np.concatenate(a.contigs(i)) for i in (0, 1, 9))
    k9 = kmers(g9)
    assert not (k9 & kmers(g0))
    assert len(k9 & {g1[p:p + 28].tobytes() for p in range(g1.size - 28)}) > len(k9) // 20
    assert synth.mixed_genomes(a, [5, 6, 7], workers=2)[1][0].tobytes() == a.contigs(6)[0].tobytes()
    assert a.fasta(6).count(b">") == len(a.contigs(6))


def m3_oracle(files, lim):
    """`mbgc c -m3` restated with the oracle: sequential matching (MBGC_Params.h:915-916) with skipMargin 24 and the
    reverse-complement factor 128 (:908-913), then rcMatchSequence over the literal stream (MBGC_Encoder.cpp:636-638)"""
    o = _orc.OracleMatcher(lim, skip_margin=24)
    oe = _orc.OracleEmitter(o, _orc.emit_params(3, enable40bitReference=1 if lim > 0xFFFFFFFF else 0))
    res = _driver.encode_sequential(o, oe, files, _driver.Policy(mode=3))
    s = oe.streams()
    lit = files[0][0].tobytes() + b"\xa2" + s["literals"]
    lit2, rc_off, rc_len, _ = _orc.rc_match_sequence(np.frombuffer(lit, dtype=np.uint8), 55)
    out = dict(s, literals=lit2, rcMapOff=rc_off, rcMapLen=rc_len, locksPos=res["locks"], refExtSize=res["refExtSize"])
    return out, o


@pytest.mark.skipif(not (_refh.available() and os.access(_refh.REF_MBGC_DEV, os.X_OK)), reason="reference build (oracle/_ref) not on this host")
def test_m3_on_mixed_species_equals_the_reference_cli(tmp_path):
    coll = synth.MixedSpecies(species=4, strains=3, length=60_000)
    n = 44
    paths = []
    for i in range(n):
        p = tmp_path / ("m%03d.fa" % i)
        p.write_bytes(coll.fasta(i))
        paths.append(str(p))
    (tmp_path / "list.txt").write_text("\n".join(paths) + "\n")
    arch = tmp_path / "mx.mbgc"
    subprocess.check_call([_refh.REF_MBGC_DEV, "c", "-m3", "-t1", str(tmp_path / "list.txt"), str(arch)], stdout=subprocess.DEVNULL)
    subprocess.check_call([_refh.REF_MBGC_DEV, "v", "-t1", "-D", str(arch)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=str(tmp_path))
    # stream order of MBGC_Decoder.cpp:1085-1112 with rcRedundancyRemoval: literals 13, rcMapOff 14, rcMapLen 15, locks 16,
    # gapDelta 17, flags 18, mapOff 19, mapLen 20, refExtSize 21
    names = {13: "literals", 14: "rcMapOff", 15: "rcMapLen", 16: "locksPos", 17: "gapDelta", 18: "flags", 19: "mapOff", 20: "mapLen", 21: "refExtSize"}
    dump = {v: (tmp_path / ("mx.mbgc_dump_%02d" % k)).read_bytes() for k, v in names.items()}
    files = [coll.contigs(i) for i in range(n)]
    lim, _ = _driver.ref_length_limit(n, os.path.getsize(paths[0]), mode=3)
    got, _ = m3_oracle(files, lim)
    for k in names.values():
        assert got[k] == dump[k], k
    assert len(dump["rcMapOff"]) > 0                                     # the reverse-complemented contigs leave something for the pass

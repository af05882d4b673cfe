"""Input stage: kseq_read_lossless_fasta over whole files. The C oracle is pinned against
the reference's own kseq.h (oracle/_ref) on hand-made edge cases, random well-formed and damaged files and the
reference's example genomes; the HIP parser is compared with the oracle on the GPU box."""
import lzma
import os

import numpy as np
import pytest

import _fasta
import _refh

HERE = os.path.dirname(os.path.abspath(__file__))
LISTERIA = sorted(os.path.join(HERE, "golden", "listeria", f) for f in os.listdir(os.path.join(HERE, "golden", "listeria")) if f.endswith(".xz"))

EDGE = [
    b"",                                                   # empty file: no record, no error
    b">",                                                  # a lone '>' at the very end: EOF, no record
    b">h",                                                 # header without newline, no sequence
    b">h\n",
    b">h\nACGT",                                           # last line without newline
    b">h\nACGT\n",
    b">h\nACGT\nAC\n",                                     # shorter last line
    b">h\nAC\nACGT\n",                                     # longer last line: not well-formed
    b">h\nACGT\nACGT\n>g\nACGT\nA\n>k\nAC\n",            # several records, one line length
    b">h\nACGT\nACG\nACGT\n",                              # inconsistent inner line
    b">h\nACGT\n\nACGT\n",                                 # empty line inside a record
    b">h\nACGT\n\n",                                       # trailing empty line
    b">h\n\n>g\nAC\n",                                     # empty line right after a header
    b"ACGT\n>h\nAC\n",                                     # does not start with '>'
    b"\n>h\nAC\n",
    b">h\nAC>GT\nACTT\n",                                  # '>' inside a line is sequence
    b">h\nACGT\n>g\n>k\nACGT\n",                           # a record without sequence
    b">a b c\tdef\r\nACGT\r\nAC\r\n",                      # CR is data in lossless mode
    b">h\nacgtnACGT\nry\n",                                # case is kept unless uppercaseDNA
    b">h\nA\nC\nG\n>g\nTT\n",                              # last line of a later record longer than the line length
    b">h\nAAAA\n>g\nCC\nCC\n",                             # line length only learnt in a later record, earlier last line longer
    bytes([62, 104, 10, 200, 65, 255, 10, 65, 66, 10]),     # bytes >= 0x80
]


def same(a, b):
    assert a["status"] == b["status"], (a["status"], b["status"])
    assert a["records"] == b["records"]
    if a["status"] == 0:
        assert a["dna_line_len"] == b["dna_line_len"]
        assert a["seq"] == b["seq"]


def random_fasta(rng, well_formed=True):
    out = bytearray()
    width = int(rng.integers(1, 90))
    for r in range(int(rng.integers(1, 6))):
        out += b">" + bytes(rng.integers(32, 127, int(rng.integers(0, 40))).astype(np.uint8)).replace(b"\n", b" ") + b"\n"
        n = int(rng.integers(0, 5 * width + 3))
        seq = bytes(rng.choice(np.frombuffer(b"ACGTNacgtn>", dtype=np.uint8), n))
        lines = [seq[i:i + width] for i in range(0, n, width)]
        lines = [ln if not ln.startswith(b">") else b"A" + ln[1:] for ln in lines]
        if not well_formed and lines and rng.random() < 0.7:
            k = int(rng.integers(0, len(lines)))
            what = rng.integers(0, 3)
            if what == 0: lines.insert(k, b"")
            elif what == 1: lines[k] = lines[k] + b"A"
            else: lines[k] = lines[k][:-1]
        out += b"\n".join(lines)
        if lines and rng.random() < 0.9:
            out += b"\n"
    return bytes(out)


@pytest.mark.ref
@pytest.mark.skipif(not _refh.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("upper", [False, True])
def test_oracle_equals_reference_reader(upper):
    for f in EDGE:
        same(_fasta.oracle_parse(f, upper), _fasta.ref_parse(f, upper))
    rng = np.random.default_rng(5)
    for i in range(300):
        f = random_fasta(rng, well_formed=i % 3 != 0)
        same(_fasta.oracle_parse(f, upper), _fasta.ref_parse(f, upper))


@pytest.mark.ref
@pytest.mark.skipif(not _refh.available(), reason="oracle/_ref not built")
def test_oracle_equals_reference_reader_on_listeria():
    for path in LISTERIA:
        data = lzma.open(path).read()
        a, b = _fasta.oracle_parse(data), _fasta.ref_parse(data)
        same(a, b)
        assert a["status"] == 0 and a["dna_line_len"] == 80 and len(a["records"]) >= 1


def test_oracle_known_answers():
    """pinned facts that need no reference build"""
    r = _fasta.oracle_parse(b">h x\nACGT\nACGT\nAC\n>g\nTT\n")
    assert r["status"] == 0 and r["dna_line_len"] == 4
    assert r["records"] == [(b"h x", b"ACGTACGTAC"), (b"g", b"TT")]
    assert _fasta.oracle_parse(b">h\nAC\nACGT\n")["status"] == -4
    assert _fasta.oracle_parse(b"ACGT\n")["status"] == -3
    assert _fasta.oracle_parse(b">h\nacgt\n", True)["records"] == [(b"h", b"ACGT")]


# ---- the HIP input stage against the oracle (GPU box) ------------------------------------------------------
def hip_parse(files, upper=False):
    import torch
    from mbgc_amd import fasta
    blob = b"".join(files)
    offs = np.zeros(len(files) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(f) for f in files])
    dev = torch.from_numpy(np.frombuffer(blob + b"\0", dtype=np.uint8).copy()).to("cuda:0")
    out = torch.zeros(max(len(blob), 1), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    p = fasta.FastaParser()
    r = p.parse_batch_dev(dev.data_ptr(), offs, out.data_ptr(), out.numel(), upper)
    seq = out.cpu().numpy().tobytes()
    res = []
    for i, f in enumerate(files):
        recs = r["records"][int(r["rec_base"][i]): int(r["rec_base"][i + 1])]
        base = int(r["seq_base"][i])
        records = [(f[int(x["headerOff"]): int(x["headerOff"] + x["headerLen"])],
                    seq[base + int(x["seqOff"]): base + int(x["seqOff"] + x["seqLen"])]) for x in recs]
        res.append(dict(status=int(r["status"][i]), records=records, dna_line_len=int(r["dna_line_len"][i]),
                        seq=seq[base: int(r["seq_base"][i + 1])]))
    p.close()
    return res


def same_ok(h, o):
    """status always; everything else whenever the reference would carry on (status 0)"""
    assert h["status"] == o["status"], (h["status"], o["status"])
    if o["status"] == 0:
        assert h["records"] == o["records"]
        assert h["dna_line_len"] == o["dna_line_len"] and h["seq"] == o["seq"]


@pytest.mark.gpu
@pytest.mark.parametrize("upper", [False, True])
def test_hip_parser_edge_cases(upper):
    files = [f for f in EDGE]
    for h, f in zip(hip_parse(files, upper), files):
        same_ok(h, _fasta.oracle_parse(f, upper))


@pytest.mark.gpu
def test_hip_parser_random_batches():
    rng = np.random.default_rng(9)
    for it in range(6):
        files = [random_fasta(rng, well_formed=(i + it) % 4 != 0) for i in range(40)]
        for h, f in zip(hip_parse(files), files):
            same_ok(h, _fasta.oracle_parse(f))


@pytest.mark.gpu
def test_hip_parser_chunk_boundaries():
    """lines, headers and newlines placed around the 4096-byte chunk edges; very long lines and headers"""
    files = []
    for width in (1, 7, 60, 80, 4095, 4096, 4097, 10000):
        for hdr in (0, 5, 4090, 4094, 4095, 4096, 9000):
            for tail in (0, 1, width):
                body = (b"ACGTTGCA" * (3 * 4096 // 8 + 2))[: 3 * 4096 + 17]
                lines = [body[i:i + width] for i in range(0, len(body), width)]
                f = b">" + b"h" * hdr + b"\n" + b"\n".join(lines) + (b"\n" if tail else b"")
                f += b">second record\n" + body[:tail] + (b"\n" if tail else b"")
                files.append(f)
    for h, f in zip(hip_parse(files), files):
        same_ok(h, _fasta.oracle_parse(f))


@pytest.mark.gpu
def test_hip_parser_listeria_and_synthetic_round():
    from mbgc_amd import synth
    files = [lzma.open(p).read() for p in LISTERIA]
    base = synth.base_codes(300_000, 3)
    files += [synth.fasta_bytes(synth.genome(base, i, 0.01), i) for i in range(4)]
    for h, f in zip(hip_parse(files), files):
        o = _fasta.oracle_parse(f)
        assert o["status"] == 0
        same_ok(h, o)

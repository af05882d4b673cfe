"""Deterministic synthetic genome collections (BASELINE.json `configs`).

A collection is a random base genome over ACGT plus, per genome, i.i.d. substitutions at a given
divergence, so every genome is (1 - 2*div)-identical to every other one and the matcher's hot path
sees the 99 %-identity regime the headline metric is quoted on. numpy's PCG64 streams make the
bytes identical on every host.
"""
import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def base_codes(length=5_000_000, seed=12345):
    return np.random.default_rng(seed).integers(0, 4, length).astype(np.uint8)


def genome_codes(base, i, divergence=0.01):
    rng = np.random.default_rng(1000 + i)
    mask = rng.random(base.size) < divergence
    shift = rng.integers(1, 4, base.size).astype(np.uint8)
    out = base.copy()
    out[mask] = (out[mask] + shift[mask]) & 3
    return out


def genome(base, i, divergence=0.01):
    """Genome `i` of the collection as ASCII bytes (uint8 array)."""
    return ACGT[genome_codes(base, i, divergence)]


def collection(n, length=5_000_000, divergence=0.01, seed=12345):
    base = base_codes(length, seed)
    return [genome(base, i, divergence) for i in range(n)]


def fasta_bytes(seq, i, width=80):
    """One-contig FASTA file image of genome i (80-column lines, LF), as the survey's files."""
    hdr = (">synth%05d synthetic genome %d 99pct identity\n" % (i, i)).encode()
    n = seq.size
    full = n // width
    body = np.empty(n + full + (1 if n % width else 0), dtype=np.uint8)
    if full:
        blk = body[: full * (width + 1)].reshape(full, width + 1)
        blk[:, :width] = seq[: full * width].reshape(full, width)
        blk[:, width] = 10
    if n % width:
        body[full * (width + 1): -1] = seq[full * width:]
        body[-1] = 10
    return hdr + body.tobytes()

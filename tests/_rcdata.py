"""Literal-stream-like test sequences for the -m3 reverse-complement pass: random bases with planted reverse-complement
copies (short, long, overlapping, palindromic, many-fold), stream marks among them."""
import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.arange(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTacgt", b"TGCAtgca"):
    COMP[_a] = _b


def revcomp(x):
    return COMP[x][::-1]


def literal_like(n, seed, copies=40, longest=3000, marks=True, mutate=0.0):
    rng = np.random.default_rng(seed)
    s = ACGT[rng.integers(0, 4, n)].copy()
    for k in range(copies):
        ln = int(rng.integers(40, longest))
        if 2 * ln + 10 >= n:
            continue
        a = int(rng.integers(0, n - ln))
        b = int(rng.integers(0, n - ln))
        piece = revcomp(s[a:a + ln]).copy()
        if mutate:
            m = rng.random(ln) < mutate
            piece[m] = ACGT[rng.integers(0, 4, int(m.sum()))]
        s[b:b + ln] = piece
    if marks:                                          # match marks / sequence separators / a lower-case run / an N run
        for p in rng.integers(0, n, max(1, n // 5000)):
            s[p] = 0xA5 if p % 2 else 0xA2
        a = int(rng.integers(0, max(1, n - 200)))
        s[a:a + 60] = np.frombuffer(bytes(s[a:a + 60]).lower(), dtype=np.uint8)
        a = int(rng.integers(0, max(1, n - 200)))
        s[a:a + 30] = ord("N")
    return s


def cases():
    out = {
        "planted": literal_like(300_000, 1),
        "long_copies": literal_like(400_000, 2, copies=12, longest=20_000),          # matches far longer than a 256-sample block
        "mutated": literal_like(300_000, 3, copies=60, longest=2_000, mutate=0.01),
        "tiny": literal_like(54, 4, copies=0),                                         # shorter than the target length: untouched
        "short": literal_like(700, 5, copies=3, longest=200),                          # no full 256-sample block: tail loop only
    }
    # one 400-base unit planted 40 times forward and 40 times reverse-complemented: buckets beyond 13 entries
    rng = np.random.default_rng(6)
    s = ACGT[rng.integers(0, 4, 200_000)].copy()
    unit = ACGT[rng.integers(0, 4, 400)]
    for k in range(40):
        a = 1000 + k * 2300
        s[a:a + 400] = unit
        s[a + 1100:a + 1500] = revcomp(unit)
    out["manyfold"] = s
    # a palindromic region (equal to its own reverse complement) and a copy pair that overlaps itself
    s = ACGT[rng.integers(0, 4, 120_000)].copy()
    half = ACGT[rng.integers(0, 4, 900)]
    s[5000:5900] = half
    s[5900:6800] = revcomp(half)
    s[40_000:41_500] = revcomp(s[39_200:40_700]).copy()
    out["palindromes"] = s
    return out

#!/usr/bin/env python3
"""`mbgc-hip c` from FASTA files in the page cache to the streams in host memory, on synthetic 5 Mbp genomes: `-t1` and
`-m 3` (the sequential schedule: every target is matched against a reference that already holds the one before it,
MGMP.cpp:232-313) and rounds of 40 — the tool's own "matching finished" clock, the rounds' wall times (MBGC_HIP_TIMES=2)
and where the host threads' time went.
usage: cpp_host_sequential.py [targets=128]"""
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mbgc_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
d = tempfile.mkdtemp(prefix="mbgc_seq_", dir=os.environ.get("TMPDIR", "/tmp"))
base = synth.base_codes(5_000_000)
paths = []
for i, g in zip(range(n + 1), synth.genomes(base, list(range(n + 1)))):
    p = os.path.join(d, "s%05d.fa" % i)
    with open(p, "wb") as f:
        f.write(synth.fasta_bytes(g, i))
    paths.append(p)
with open(os.path.join(d, "list.txt"), "w") as f:
    f.write("\n".join(paths) + "\n")
out = {}
for name, args in (("rounds_of_40", ["-R", "40"]), ("t1", ["-t1"]), ("m3", ["-m", "3"])):
    t0 = time.time()
    r = subprocess.run([os.path.join(ROOT, "mbgc_amd", "mbgc-hip"), "c"] + args + [os.path.join(d, "list.txt"), os.path.join(d, "out")],
                       capture_output=True, text=True, env=dict(os.environ, MBGC_HIP_TIMES="2"))
    wall = time.time() - t0
    sys.stderr.write(r.stderr[-600:])
    m = re.search(r"matching finished - (\d+) \[ms\]", r.stderr)
    ms = int(m.group(1)) if m else None
    out[name] = dict(rc=r.returncode, wall_s=round(wall, 2), matching_ms=ms,
                     gbases_per_s=round(n * 5e6 / (ms / 1e3) / 1e9, 3) if ms else None)
    rounds = [float(x) for x in re.findall(r"round \d+: ([0-9.]+) ms", r.stderr)]
    if rounds:
        tail = sorted(rounds[len(rounds) // 2:])
        out[name]["round_ms"] = rounds
        out[name]["round_ms_median_of_second_half"] = tail[len(tail) // 2]
        out[name]["gbases_per_s_at_that_round_time"] = round(40 * 5e6 / (tail[len(tail) // 2] / 1e3) / 1e9, 2)
    where = [x.strip() for x in r.stderr.splitlines() if "reader threads:" in x]
    if where:
        out[name]["host_threads"] = where[-1]
print(json.dumps(dict(targets=n, genome_len=5_000_000, runs=out)))
subprocess.run(["rm", "-rf", d])

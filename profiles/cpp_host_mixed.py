#!/usr/bin/env python3
"""`mbgc-hip c` on the mixed-species collection (mbgc_amd/synth.py: MixedSpecies — BASELINE configs[4]'s kind of data: 8 unrelated
species x 4 strains, 0.2-10 % divergence, 1-4 contigs, reverse-complemented contigs, N runs) from FASTA files in the page cache:
the `-m 3` presets (sequential schedule) and `-m1` in window-sized rounds — the tool's own "matching finished" clock.
usage: cpp_host_mixed.py [genomes=200]"""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
# the files are written by forked workers (bench.py's helper, as for its configs[4] line): written by this one process they all
# sat in the page cache of its NUMA node, and the tool's reader threads — bound to the GPU's node — read them at half the rate
import bench  # noqa: E402
d = bench.write_mixed_species(n)
bases = json.load(open(os.path.join(d, "meta.json")))["bases"]
out = {}
runs = [("m3", ["-m", "3"]), ("m1_rounds", []), ("m1_rounds_again", []), ("m1_rounds_third", []), ("m1_t1", ["-t1"])]   # (the rounds three times: the first round's file reads and page-locked buffers make runs differ by tens of milliseconds)
if os.environ.get("MBGC_MIX_RUNS"):                  # e.g. "R2:-R 2,R4:-R 4": other command lines instead of the three above
    runs = [(x.split(":")[0], x.split(":")[1].split()) for x in os.environ["MBGC_MIX_RUNS"].split(",")]
for name, args in runs:
    t0 = time.time()
    r = subprocess.run([os.path.join(ROOT, "mbgc_amd", "mbgc-hip"), "c"] + args + [os.path.join(d, "list.txt"), os.path.join(d, "out")],
                       capture_output=True, text=True, env=dict(os.environ, MBGC_HIP_TIMES="1"))
    wall = time.time() - t0
    m = re.search(r"matching finished - (\d+) \[ms\]", r.stderr)
    ms = int(m.group(1)) if m else None
    um = re.search(r"final unmatched chars: (\d+)", r.stdout)
    rd = re.search(r"rounds of (\d+) targets", r.stdout)
    out[name] = dict(rc=r.returncode, wall_s=round(wall, 2), matching_ms=ms, gbases_per_s=round(bases / (ms / 1e3) / 1e9, 3) if ms else None,
                     final_unmatched_chars=int(um.group(1)) if um else None, targets_per_round=int(rd.group(1)) if rd else None)
    where = [x.strip() for x in r.stderr.splitlines() if "reader threads:" in x]
    if where:
        out[name]["host_threads"] = where[-1]
    if r.returncode:
        sys.stderr.write(r.stderr[-800:])
rounds = sorted(v["gbases_per_s"] for k, v in out.items() if k.startswith("m1_rounds") and v.get("gbases_per_s"))
print(json.dumps(dict(genomes=n, bases=bases, m1_rounds_median_gbases_per_s=rounds[len(rounds) // 2] if rounds else None, runs=out)))
subprocess.run(["rm", "-rf", d])

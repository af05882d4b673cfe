#!/usr/bin/env python3
"""What a round's size costs in compression ratio, on data that can show it: a mixed-species collection
(mbgc_amd/synth.py: MixedSpecies — divergent genomes, where a reference extension that is dropped is matches lost later) against
a circular buffer small enough to wrap several times. Once the buffer has wrapped the targets of a round share one lock
position one sliding window ahead of the loader, and loadRef drops what a round loads beyond it
(SlidingWindowSparseEMMatcher.cpp:361-378,412-417,433): rounds larger than the window lose reference.
  * the reference itself: `mbgc c -m1 -t1` (sequential: every target sees every earlier one) and with its default threads
    (its own parallel schedule), same buffer (`-o <order>`);
  * this repo: `mbgc-hip c --ref-factor F -R r --backend <the reference's leaf coders>` for r = the window's own size (no -R),
    1, 8, 32, 40, 64, 320: bytes of the collective section, extension bytes dropped, matching time.
usage: ratio_sweep.py [genomes=480] [genome length=1000000] [reference factor order=5] [backend threads=16]"""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 480
length = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
order = int(sys.argv[3]) if len(sys.argv) > 3 else 5
threads = sys.argv[4] if len(sys.argv) > 4 else "16"
REF = os.path.join(ROOT, "oracle", "_ref")
d = tempfile.mkdtemp(prefix="mbgc_sweep_", dir=os.environ.get("TMPDIR", "/tmp"))
coll = synth.MixedSpecies(length=length)
paths, bases = [], 0
for i, contigs in zip(range(n), synth.mixed_genomes(coll, range(n))):
    p = os.path.join(d, "m%05d.fa" % i)
    with open(p, "wb") as f:
        for c, seq in enumerate(contigs):
            f.write((">mixed%05d.%d\n" % (i, c)).encode() + synth.fasta_bytes(seq, i).split(b"\n", 1)[1])
            bases += seq.size
    paths.append(p)
lst = os.path.join(d, "list.txt")
with open(lst, "w") as f:
    f.write("\n".join(paths) + "\n")
g0 = sum(c.size for c in coll.contigs(0))
buf = (1 << order) * max(g0, 1 << 21) * 2
out = dict(collection="MixedSpecies(length=%d): %d genomes, 8 species x 4 strains, 0.2-10 %% divergence, 1-4 contigs" % (length, n), bases=bases,
           reference_factor=1 << order, buffer_bytes=buf, sliding_window_bytes=buf // 16, laps_of_the_buffer_at_most=round(bases / buf, 2))


def reference(extra, name):
    t0 = time.time()
    arch = os.path.join(d, name + ".mbgc")
    r = subprocess.run([os.path.join(REF, "mbgc"), "c", "-m1", "-o", str(order)] + extra + [lst, arch], capture_output=True, text=True)
    wall = time.time() - t0
    size = os.path.getsize(arch) if r.returncode == 0 else None
    return dict(command="mbgc c -m1 -o %d %s" % (order, " ".join(extra)), rc=r.returncode, wall_s=round(wall, 2), archive_bytes=size)


out["reference_t1"] = reference(["-t1"], "ref_t1")
out["reference_default_threads"] = reference([], "ref_par")
base_size = out["reference_t1"]["archive_bytes"]
rows = []
for R in (0, 1, 8, 32, 40, 64, 320):
    t0 = time.time()
    args = [os.path.join(ROOT, "mbgc_amd", "mbgc-hip"), "c", "--ref-factor", str(1 << order)] + (["-R", str(R)] if R else []) + \
           ["--backend", os.path.join(REF, "libmbgc_coders.so"), "--backend-threads", threads, lst, os.path.join(d, "hip%d" % R)]
    r = subprocess.run(args, capture_output=True, text=True)
    wall = time.time() - t0
    m = re.search(r"backend: (\d+) stream bytes to (\d+) in (\d+) ms", r.stdout)
    mm = re.search(r"matching finished - (\d+) \[ms\]", r.stderr)
    dr = re.search(r"rounds of (\d+) targets; reference extension bytes dropped at the sliding window's end: (\d+)", r.stdout)
    sec = int(m.group(2)) if m else None
    rows.append(dict(R="window" if R == 0 else R, targets_per_round=int(dr.group(1)) if dr else None, rc=r.returncode, wall_s=round(wall, 2),
                     matching_ms=int(mm.group(1)) if mm else None, extension_bytes_dropped=int(dr.group(2)) if dr else None,
                     stream_bytes=int(m.group(1)) if m else None, collective_section_bytes=sec,
                     section_over_reference_t1_archive=round(sec / base_size, 4) if sec and base_size else None))
    if r.returncode:
        sys.stderr.write(r.stderr[-800:])
out["this_repo"] = rows
out["note"] = ("the section holds the match / literal streams; the reference's archive also holds file names, headers and its parameter block "
               "(a few KB for this collection)")
print(json.dumps(out))
subprocess.run(["rm", "-rf", d])

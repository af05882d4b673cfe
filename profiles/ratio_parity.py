#!/usr/bin/env python3
"""Compression-ratio parity on synthetic 5 Mbp genomes (BASELINE.json's second clause): the reference's own `mbgc c -m1`
(oracle/_ref/mbgc, all hardware threads: its parallel schedule) against `mbgc-hip c --backend` (rounds sized by the sliding window), i.e. this repo's
streams through the job table + container of include/mbgc_backend.h with the reference's unchanged PPMd / LZMA as the leaf
coders. The two archives come from different (both admissible) schedules of the same encoder, so their sizes agree to a
fraction of a percent rather than byte for byte; byte identity is checked at the stream level by the tests.
usage: ratio_parity.py [targets=1000] [backend threads=16]"""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
threads = sys.argv[2] if len(sys.argv) > 2 else "16"
REF = os.path.join(ROOT, "oracle", "_ref")
d = tempfile.mkdtemp(prefix="mbgc_ratio_", dir=os.environ.get("TMPDIR", "/tmp"))
base = synth.base_codes(5_000_000)
paths = []
for i, g in zip(range(n + 1), synth.genomes(base, list(range(n + 1)))):
    p = os.path.join(d, "s%05d.fa" % i)
    with open(p, "wb") as f:
        f.write(synth.fasta_bytes(g, i))
    paths.append(p)
with open(os.path.join(d, "list.txt"), "w") as f:
    f.write("\n".join(paths) + "\n")
bases = (n + 1) * 5_000_000
out = dict(genomes=n + 1, bases=bases)
t0 = time.time()
r = subprocess.run([os.path.join(REF, "mbgc"), "c", "-m1", os.path.join(d, "list.txt"), os.path.join(d, "ref.mbgc")], capture_output=True, text=True)
wall = time.time() - t0
size = os.path.getsize(os.path.join(d, "ref.mbgc")) if r.returncode == 0 else None
out["reference"] = dict(command="mbgc c -m1 (all hardware threads)", rc=r.returncode, wall_s=round(wall, 2), archive_bytes=size,
                        bases_per_byte=round(bases / size, 2) if size else None)
for key, blocks, th in (("this_repo", "1", threads), ("this_repo_8x_blocks", "8", "64"), ("this_repo_backend_beside_the_matching_16MiB_blocks", "overlap", "64"),
                        ("this_repo_backend_beside_the_matching_4MiB_blocks", "overlap4", "64")):
    t0 = time.time()
    how = ["--backend-blocks", blocks] if not blocks.startswith("overlap") else ["--backend-overlap", "16" if blocks == "overlap" else "4"]
    r = subprocess.run([os.path.join(ROOT, "mbgc_amd", "mbgc-hip"), "c", "--backend", os.path.join(REF, "libmbgc_coders.so"), "--backend-threads", th] + how +
                       [os.path.join(d, "list.txt"), os.path.join(d, "hip")], capture_output=True, text=True)
    wall = time.time() - t0
    m = re.search(r"backend: (\d+) stream bytes to (\d+)(?: in|,) (\d+) ms", r.stdout)
    early = re.search(r"(\d+) of them coded while the matching ran", r.stdout)
    mm = re.search(r"matching finished - (\d+) \[ms\]", r.stderr)
    sec = int(m.group(2)) if m else None
    out[key] = dict(command="mbgc-hip c --backend <the reference's leaf coders, oracle/_ref/libmbgc_coders.so> --backend-threads %s %s" % (th, " ".join(how)), rc=r.returncode,
                    wall_s=round(wall, 2), matching_ms=int(mm.group(1)) if mm else None, stream_bytes=int(m.group(1)) if m else None,
                    collective_section_bytes=sec, backend_ms=int(m.group(3)) if m else None, bases_per_byte=round(bases / sec, 2) if sec else None,
                    section_over_reference_archive=round(sec / size, 4) if size and sec else None,
                    **({"blocks_coded_while_the_matching_ran": int(early.group(1))} if early else {}))
    if r.returncode:
        sys.stderr.write(r.stderr[-800:])
out["note"] = ("the section holds the match / literal streams; the file-name, header and line-length streams (a few KB for this collection) and the "
               "parameter block are the reference CLI's and are not in it")
print(json.dumps(out))
subprocess.run(["rm", "-rf", d])

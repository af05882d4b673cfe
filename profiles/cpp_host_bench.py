#!/usr/bin/env python3
"""The C++ host (mbgc_amd/host: MBGC_Encoder -> MultipleGenomeMatchingProcessor -> C ABI) on bench.py's workload:
writes the synthetic collection as FASTA files, runs `mbgc-hip c --bench --warmup W -R 40 list out` (every round
resident in HBM before the clock, streams left in HBM) and prints the tool's JSON line.
usage: cpp_host_bench.py [targets=1000] [round=40] [warmup_rounds=5]"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mbgc_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
rnd = int(sys.argv[2]) if len(sys.argv) > 2 else 40
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 5
d = tempfile.mkdtemp(prefix="mbgc_cpp_", dir=os.environ.get("TMPDIR", "/tmp"))
base = synth.base_codes(5_000_000)
t0 = time.time()
paths = []
for lo in range(0, n + 1, 100):
    ids = list(range(lo, min(n + 1, lo + 100)))
    for i, g in zip(ids, synth.genomes(base, ids)):
        p = os.path.join(d, "s%05d.fa" % i)
        with open(p, "wb") as f:
            f.write(synth.fasta_bytes(g, i))
        paths.append(p)
with open(os.path.join(d, "list.txt"), "w") as f:
    f.write("\n".join(paths) + "\n")
print("wrote %d files in %.1f s" % (len(paths), time.time() - t0), file=sys.stderr)
rc = 0
# the plain loop, then the same through the sharded loop's exchange with ONE rank over RCCL (communicators, the all-gather of
# the round's extensions on the bulk stream, the on-stream reduction inside the speculative finalize): what the exchange
# costs a rank before a second GPU is there
for env in ({}, {"MBGC_HIP_EXCHANGE": "1"}):
    t0 = time.time()
    r = subprocess.run([os.path.join(ROOT, "mbgc_amd", "mbgc-hip"), "c", "--bench", "--warmup", str(warm), "-R", str(rnd),
                        os.path.join(d, "list.txt"), os.path.join(d, "out")], capture_output=True, text=True, env=dict(os.environ, **env))
    print("tool %s: %.1f s wall, rc %d" % (env, time.time() - t0, r.returncode), file=sys.stderr)
    sys.stderr.write(r.stderr[-2000:])
    print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "")
    rc = rc or r.returncode
subprocess.run(["rm", "-rf", d])
sys.exit(rc)

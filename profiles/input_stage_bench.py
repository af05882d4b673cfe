#!/usr/bin/env python3
"""Times the device input stage (include/mbgc_fasta.h) on one round of configs[1]: 16 synthetic 5 Mbp FASTA files
(80-column lines) resident in HBM -> their contigs back to back in HBM. Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from mbgc_amd import fasta, synth

L, R = 5_000_000, 16
base = synth.base_codes(L)
files = [bytes(synth.fasta_bytes(synth.genome(base, 1 + i), 1 + i)) for i in range(R)]
blob = np.frombuffer(b"".join(files), dtype=np.uint8).copy()
offs = np.zeros(R + 1, dtype=np.uint64)
offs[1:] = np.cumsum([len(f) for f in files])
dev = torch.from_numpy(blob).to("cuda:0")
out = torch.empty(blob.size, dtype=torch.uint8, device="cuda:0")
torch.cuda.synchronize()
p = fasta.FastaParser()
for _ in range(3):
    r = p.parse_batch_dev(dev.data_ptr(), offs, out.data_ptr(), out.numel())
assert (r["status"] == 0).all() and (r["dna_line_len"] == 80).all() and int(r["seq_base"][-1]) == R * L
K = 20
t0 = time.perf_counter()
for _ in range(K):
    p.parse_batch_dev(dev.data_ptr(), offs, out.data_ptr(), out.numel())
dt = (time.perf_counter() - t0) / K
print(json.dumps({"what": "mbgc_fasta_parse_batch_dev, 16 x 5 Mbp FASTA (80 columns) in HBM -> contigs in HBM, call incl. its two host waits",
                  "ms_per_round": round(dt * 1e3, 3), "file_bytes": int(blob.size), "Gbases_per_s": round(R * L / dt / 1e9, 1),
                  "file_GB_per_s": round(blob.size / dt / 1e9, 1)}))

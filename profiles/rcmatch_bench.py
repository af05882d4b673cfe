#!/usr/bin/env python3
"""Times the `-m3` reverse-complement pass over a literal-stream-like sequence (include/mbgc_copmem.h: CopMEM index + query
scan on the device, post-processing on the host) next to the CPU restatement of the reference's single-thread path
(oracle/rcmatch_oracle.c) on this host, and checks that both give the same bytes. usage: rcmatch_bench.py [megabytes=256]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _orc  # noqa: E402
import _rcdata  # noqa: E402
from mbgc_amd import copmem  # noqa: E402

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = mb * 1_000_000
rng = np.random.default_rng(5)
s = _rcdata.ACGT[rng.integers(0, 4, n)].copy()
for k in range(n // 200_000):                       # reverse-complement copies of 60 .. 20 000 bases, about 2.5 % of the stream
    ln = int(rng.integers(60, 20_000))
    a, b = int(rng.integers(0, n - ln)), int(rng.integers(0, n - ln))
    s[b:b + ln] = _rcdata.revcomp(s[a:a + ln])
m = copmem.SimpleSequenceMatcher()
m.rc_matches(s[:10_000_000])                        # warm-up (allocations, code objects)
t0 = time.perf_counter(); rows, params = m.rc_matches(s); t_find = time.perf_counter() - t0
t0 = time.perf_counter(); got = m.rc_match_sequence(s); t_all = time.perf_counter() - t0
t0 = time.perf_counter(); want_rows, _, ext = _orc.rc_find_matches(s); t_cpu_find = time.perf_counter() - t0
t0 = time.perf_counter(); want = _orc.rc_match_sequence(s); t_cpu_all = time.perf_counter() - t0
assert np.array_equal(rows, want_rows) and got == want
print(json.dumps({"sequence_bytes": n, "params_K_k1_k2_log2hash": list(params), "matches_pushed": int(len(rows)), "matched_chars": int(got[3][1]),
                  "device_find_matches_s": round(t_find, 4), "device_rcMatchSequence_s": round(t_all, 4),
                  "device_GB_per_s": round(n / t_find / 1e9, 2),
                  "cpu_1thread_find_matches_s": round(t_cpu_find, 3), "cpu_1thread_rcMatchSequence_s": round(t_cpu_all, 3),
                  "cpu_character_extensions": int(ext), "identical_output": True,
                  "note": "device time includes the host-to-device copy of the sequence (it lives on the host, where the backend consumes it) "
                          "and the device-to-host copy of the matches; CPU = oracle/rcmatch_oracle.c, the reference's single-thread semantics"}))

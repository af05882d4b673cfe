#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (oracle/dropin): applies INTEGRATION.md's patch to a SCRATCH COPY of the reference (v2.1.5) — only
replacement lines live in this repo, none of the reference's:
  1. matching/SlidingWindowSparseEMMatcher.h  <- oracle/dropin/SlidingWindowSparseEMMatcher.h (the facade over the C ABI)
  2. mbgccoder/MBGC_Encoder.cpp lines 143-427 (processMatches, extendMatchRight, extendMatchLeft: the only users of
     getRef())  <- oracle/dropin/processMatches_hip.inc
matching/SlidingWindowSparseEMMatcher.cpp leaves the build (oracle/Makefile: `dropin`); everything else is untouched.
usage: apply_dropin.py <scratch copy of the reference tree>"""
import os
import shutil
import sys

here = os.path.dirname(os.path.abspath(__file__))
tree = sys.argv[1]
shutil.copyfile(os.path.join(here, "SlidingWindowSparseEMMatcher.h"), os.path.join(tree, "matching", "SlidingWindowSparseEMMatcher.h"))
enc = os.path.join(tree, "mbgccoder", "MBGC_Encoder.cpp")
lines = open(enc).read().split("\n")
first, last = 143, 427                                       # 1-based, inclusive
assert lines[first - 1].startswith("size_t MBGC_Encoder::processMatches("), "not the reference version this patch was written for (v2.1.5)"
assert lines[last - 1] == "}" and lines[last + 1].startswith("void MBGC_Encoder::loadFileNames"), "not the reference version this patch was written for (v2.1.5)"
new = open(os.path.join(here, "processMatches_hip.inc")).read().rstrip("\n").split("\n")
lines[first - 1: last] = new
# the C ABI's declarations for the replacement body
inc = next(i for i, l in enumerate(lines) if l.startswith("#include"))
lines.insert(inc, '#include "mbgc_swsem.h"')
lines.insert(inc, '#include <unistd.h>')                       # (_exit in the refusal of the parallel schedule)
open(enc, "w").write("\n".join(lines))
print("applied: matcher header replaced, MBGC_Encoder.cpp:%d-%d -> %d lines" % (first, last, len(new)))

#!/usr/bin/env python3
"""Headline benchmark: input Gbases/s of the MI355X match-finding path (BASELINE.json `metric`).

Workload (BASELINE.json configs[2] on one GPU, configs[3] on several: the SAME collection at every N): 1000 synthetic
5 Mbp genomes at 99 % identity, matched against the growing circular reference `mbgc c` sizes for that collection (G0 +
reverse complement preloaded; 2.56e9 bytes and a 2^28-entry table for 1001 files, MGMP.cpp:130-168 — at every N). A
*step* is one round: the round's targets hold their lock positions together, so a round is as large as the reference's
own rules let targets be in flight — what they load must fit the sliding window (1/16 of the buffer = 160 MB = 31
targets; SlidingWindowSparseEMMatcher.cpp:361-378,412-417,433: what goes beyond the window is dropped) and at most 64
(MGMP.cpp:374-375,532) — and NO extension byte is dropped (`extension_bytes_dropped_per_step` must be 0, or the line says
INVALID). One GPU: rounds of 31. Several GPUs: the total work of a step is the same at every N (strong scaling) — 24
targets, the largest multiple of 8 the window allows, dealt 12 / 6 / 3 per GPU (file-per-GPU) at N = 2 / 4 / 8; the one-GPU
run times that step too, beside its headline (`scaling_reference`), so the curve has its N = 1 point at the same step. Every
GPU matches its share of the round against its frozen replica — matchTexts + processMatches, six streams — then every
replica loads the round's extensions in target order (loadRef, hash insertion included). The run goes THROUGH the wrap of
the circular buffer (near target 510); step times before and after it are reported separately. With N > 1 the extension
bytes are exchanged over RCCL and the streams gathered to rank 0. Inputs are resident in HBM before the timed region. One
JSON line is printed by rank 0.

`python bench.py --gpus N` starts its own N ranks (children, before anything in this process touches a GPU) and watches
them: a rank that dies ends the run within seconds with a line that says `"value": null` and why (spawn_ranks); under
`python -m torch.distributed.run` (WORLD_SIZE set) it is one of the ranks, with a 120 s process-group timeout.

Beside the headline, N = 1 only and never inside its timed region (`--no-extras` skips them): `cpp_host` — the product's C++
host (`mbgc-hip c --bench`) on the same collection from FASTA files; `configs` — BASELINE configs[1] (128 genomes, matcher
only), the configs[4] data class and sizing (mixed-species genomes, `-m 3`, 4.5e9-byte reference with 40-bit offsets)
and the same genomes in `-m1` rounds (rounds on a divergent collection go on in units) through the C++ host, each
with its match-finding kernel's roofline; `cpu_baseline` — the reference's own OpenMP path.

Diagnostics (never the headline): --no-emit (matcher only), --from-host (queries cross PCIe inside the timed region),
--check (first round against the oracle); environment: MBGC_BENCH_MAX_REF (another buffer size), MBGC_BENCH_BLOCK_STATS
/ MBGC_BENCH_BLOCK_DUMP (per-block clocks of the last resolve launch), MBGC_BENCH_ONE_DEVICE + MBGC_BENCH_BACKEND=gloo
(several ranks on one GPU, a rehearsal of the N > 1 protocol), MBGC_BENCH_TIMEOUT / MBGC_BENCH_PG_TIMEOUT (seconds), and the
library's own switches (SWSEM_CHAINS, SWSEM_RB, SWSEM_RESOLVE, SWSEM_PROF_FAMS; see mbgc_amd/csrc/swsem_runtime.hip: swsem_create)."""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GENOME_LEN = 5_000_000
COLLECTION = 1000                    # targets of configs[2] / configs[3]
SW_FACTOR = 16                       # DEFAULT_REFERENCE_SLIDING_WINDOW_FACTOR, MGMP_Params.h:36
MAX_IN_FLIGHT = 64                   # matcherWorkingThreads x allowedTargetsOutrunFactor, MGMP.cpp:532, MGMP_Params.h:16,57


def targets_in_flight(max_ref, target_bytes):
    """targets that may hold their lock positions together: what they load (target + region separator each) fits the
    sliding window, and the reference never lets more than 64 run ahead of its finalizer"""
    return max(1, min(MAX_IN_FLIGHT, (max_ref // SW_FACTOR) // (target_bytes + 1)))


def ref_length_limit(files_count, basic_len):
    """the reference buffer `mbgc c` derives for a collection (loadG0Ref MGMP.cpp:130-134, initMatcher :152-168; -m1, RC
    in the reference, circular): 1001 files of 5 Mbp give 2.56e9 bytes; above 2048 files the rule asks for 5.12e9,
    which initMatcher squeezes to 2^32 - 1 + the excess / 16 and addresses with 40-bit offsets (mapOff5th)."""
    clz = 32 - int(files_count).bit_length()
    factor = 1 << min(12, max(5, 15 - clz // 3))
    lim = factor * max(basic_len, 1 << 21) * 2
    if lim > (0xFFFFFFFF << 8):
        lim = 0xFFFFFFFF << 8
    if lim > 0xFFFFFFFF:
        lim = 0xFFFFFFFF + (lim - 0xFFFFFFFF) // 16
    return lim


ALG_BYTES_PER_BASE = 5.5             # whole path, per input base
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md: 8 TB/s
# SURVEY.md §8(d) splits that figure by term; per kernel family (bytes per input base of a launch):
#   resolve k_resolve_blocks4: the query scan (1 B: the scan windows hash their K-mers from the query bytes), the table
#           probes the sequential loop performs (0.29 x 4 B), the reference bytes it compares (0.918 B) and the match rows
#           it writes (24 B x 9.6 k rows / 1 M bases)
#   load    extension copy, read + write (2 B);  insert  one 4-B table entry per 16 bases
#   emit    the six streams (0.14 B)
ALG_BYTES = {"resolve": 1.0 + 4 * 0.29 + 0.918 + 0.23, "stitch": 0.23, "load": 2.0, "insert": 0.25, "emit": 0.14}
KERNEL_OF = {"resolve": "k_resolve_blocks4", "stitch": "k_stitch_pre + k_stitch + k_gather", "load": "k_copy_multi",
             "insert": "k_insert_multi", "emit": "k_emit_* (13 launches)"}
# HBM-side bytes per launch of each kernel, from the PMC passes over THIS command (profiles/pmc_summary.py writes the
# file from rocprofv3's FETCH_SIZE / WRITE_SIZE passes, corrected as MI355X_MICROARCH.md prescribes); absent = null
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r04_pmc_traffic.json")


MIXED_GENOMES = 200                  # configs[4]'s kind of data, as many genomes as a bench line can afford (the full 10 000: profiles/configs4_run.py)
TOOL = os.path.join(ROOT, "mbgc_amd", "mbgc-hip")


def resolve_alg_bytes(bases, matched, matches):
    """SURVEY.md §8(d) for the match-finding kernel on ANY data: the query scan (1 B per base), the table probes the sequential
    loop performs (4 B each: it visits what no match lets it jump over — the unmatched positions, and per match about 8 in front
    of the hit and the 16 of the skip margin behind it: 0.31 per base on the 99 %-identity collection, where the oracle counts
    0.29), the reference bytes it compares (the matched length) and the 24-byte rows it writes."""
    probes = bases - matched + 24 * matches
    return bases + 4 * probes + matched + 24 * matches


def _mixed_one(args):
    d, i = args
    from mbgc_amd import synth
    coll = synth.MixedSpecies()
    with open(os.path.join(d, "m%05d.fa" % i), "wb") as f:
        f.write(coll.fasta(i))
    return sum(int(c.size) for c in coll.contigs(i))


def write_mixed_species(n):
    """n genomes of synth.MixedSpecies (8 unrelated species x 4 strains, 0.2-10 % divergence, 1-4 contigs, reverse-complemented
    contigs, N runs) as FASTA files, by forked workers — before this process touches the GPU"""
    import multiprocessing as mp
    import tempfile
    d = tempfile.mkdtemp(prefix="mbgc_bench_mix_", dir=os.environ.get("TMPDIR", "/tmp"))
    with mp.get_context("fork").Pool(min(16, os.cpu_count() or 1)) as p:
        sizes = p.map(_mixed_one, [(d, i) for i in range(n)], chunksize=2)
    with open(os.path.join(d, "list.txt"), "w") as f:
        f.write("".join(os.path.join(d, "m%05d.fa" % i) + "\n" for i in range(n)))
    with open(os.path.join(d, "meta.json"), "w") as f:
        json.dump({"genomes": n, "bases": sum(sizes), "g0_bases": sizes[0]}, f)
    return d


def _tool_profile(stderr):
    import re
    m = re.search(r"kernel profile: (\{.*\})", stderr)
    return json.loads(m.group(1)) if m else None


def cpp_host_line(fasta_dir, n_targets, warm, py_value):
    """the product's host — C++ classes over the C ABI, `mbgc-hip c --bench` — on the headline's collection (the same genomes as
    FASTA files: read, uploaded and parsed on the device before the clock starts; rounds sized by the window; timed behind
    `warm` warm-up rounds like the line above it)"""
    try:
        lst = os.path.join(fasta_dir, "list.txt")
        with open(lst, "w") as f:
            f.write("".join(synth_path(fasta_dir, i) + "\n" for i in range(n_targets + 1)))
        t0 = time.perf_counter()
        r = subprocess.run([TOOL, "c", "--bench", "--warmup", str(warm), lst, os.path.join(fasta_dir, "out")], capture_output=True, text=True,
                           timeout=300, env=dict(os.environ, MBGC_HIP_PROFILE="1"))
        wall = time.perf_counter() - t0
        if r.returncode != 0:
            return {"error": "mbgc-hip exited with %d: %s" % (r.returncode, r.stderr[-300:])}
        d = json.loads(r.stdout.strip().splitlines()[-1])
        out = {"command": "mbgc-hip c --bench --warmup %d <list of the same %d files>" % (warm, n_targets + 1), "value": d["value"], "unit": "Gbases/s",
               "ms_per_round": d["ms_per_round"], "rounds": d["rounds"], "targets_per_round": d["targets_per_round"],
               "extension_bytes_dropped": d.get("extension_bytes_dropped"), "max_ref_len": d.get("max_ref_len"),
               "over_the_python_line": round(d["value"] / py_value, 4) if py_value else None, "wall_seconds_of_the_tool": round(wall, 2)}
        prof = _tool_profile(r.stderr)
        if prof and prof["resolve"]["launches"]:
            ms = prof["resolve"]["ms"] / prof["resolve"]["launches"]
            ach = ALG_BYTES["resolve"] * d["targets_per_round"] * GENOME_LEN / (ms * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": KERNEL_OF["resolve"], "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": None, "avg_launch_ms": round(ms, 4), "alg_bytes_per_base": round(ALG_BYTES["resolve"], 3)}
            out["kernel_ms_per_launch"] = {k: round(v["ms"] / v["launches"], 4) for k, v in prof.items() if v["launches"]}
        return out
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, e)}


def synth_path(fasta_dir, i):
    from mbgc_amd import synth
    return synth.fasta_path(fasta_dir, i)


def config1_line():
    """BASELINE configs[1]: 128 synthetic 5 Mbp genomes, SlidingWindowSparseEMMatcher only (matchTexts + loadRef, no emission),
    against the 1.28e9-byte buffer `mbgc c` gives 129 files — this script again, as a child, in rounds of 16"""
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--no-emit", "--round", "16", "--steps", "7", "--warmup", "1", "--cpu-sample", "0",
                            "--no-extras"], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, MBGC_BENCH_MAX_REF=str(ref_length_limit(129, GENOME_LEN))))
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"config": "configs[1]", "error": "child bench exited with %d: %s" % (r.returncode, r.stderr[-300:])}
        d = json.loads(lines[-1])
        return {"config": "configs[1]", "workload": "configs[1]: 128 synthetic 5 Mbp genomes @99%% identity, SlidingWindowSparseEMMatcher only (matchTexts + loadRef per round, no "
                                                    "emission), %.4g-byte circular reference (the buffer `mbgc c` gives 129 files), rounds of 16; timed: targets 17..128" % d["config"]["max_ref_len"],
                "metric": d["metric"], "value": d["value"], "unit": d["unit"], "extension_bytes_dropped_per_step": d.get("extension_bytes_dropped_per_step"),
                "ms_per_step": d["ms_per_step"], "targets_per_step": d["config"]["targets_per_step"], "max_ref_len": d["config"]["max_ref_len"],
                "roofline": d["roofline"], "kernel_ms_per_launch": d["kernel_ms_per_launch"]}
    except Exception as e:
        return {"config": "configs[1]", "error": "%s: %s" % (type(e).__name__, e)}


def config4_line(mixed_dir, rounds=False):
    """rounds=True: the same files in the default presets (`-m1`) and window-sized rounds — rounds on a divergent collection: most
    targets of most rounds hold a contig MBGC calls dissimilar (MGMP.cpp:382-388), the round goes on in units.
    Otherwise BASELINE configs[4]'s data class and sizing on what a bench line can afford: MIXED_GENOMES mixed-species genomes through the C++
    host in the `-m 3` presets (sequential schedule, skip margin 24, reverse-complement pass) against the 4.5e9-byte buffer with
    40-bit offsets and 2^29 buckets that `mbgc c -m3` gives 10 001 files (--ref-factor 512) — files from the page cache, the tool's
    own clock over its matching phase (reading, upload, device parse, matchTexts, processMatches, loadRef)"""
    import re
    if not mixed_dir:
        return {"config": "configs[4]", "error": "no mixed-species files"}
    try:
        meta = json.load(open(os.path.join(mixed_dir, "meta.json")))
        r = subprocess.run([TOOL, "c"] + ([] if rounds else ["-m", "3", "--ref-factor", "512"]) + [os.path.join(mixed_dir, "list.txt"), os.path.join(mixed_dir, "out")],
                           capture_output=True, text=True, timeout=300, env=dict(os.environ, MBGC_HIP_PROFILE="1"))
        if r.returncode != 0:
            return {"config": "configs[4]", "error": "mbgc-hip exited with %d: %s" % (r.returncode, r.stderr[-300:])}
        ms = int(re.search(r"matching finished - (\d+) \[ms\]", r.stderr).group(1))
        matches = int(re.search(r"exact matches total: (\d+)", r.stdout).group(1))
        unmatched = int(re.search(r"swsMEM unmatched chars: (\d+)", r.stdout).group(1))
        ref_len = int(re.search(r"final reference length: (\d+)", r.stdout).group(1))
        bases = meta["bases"] - meta["g0_bases"]                      # the targets (the first file is the reference)
        out = {"config": "configs[4]", "workload": "%d mixed-species synthetic genomes (8 species x 4 strains, 0.2-10 %% divergence, 1-4 contigs), `mbgc-hip c -m 3 "
                                                   "--ref-factor 512`: sequential schedule, 4.5e9-byte reference with 40-bit offsets, 2^29 buckets; C++ host, files "
                                                   "from the page cache" % meta["genomes"],
               "metric": "input Gbases/s (compress path of the C++ host, -m3: file reading + device parse + match-finding + stream emission)",
               "value": 0}
        if rounds:
            rd = re.search(r"rounds of (\d+) targets", r.stdout)
            out = {"config": "configs[4] data in rounds", "workload": "the same %d mixed-species genomes, `mbgc-hip c` (-m1) in window-sized rounds of %s targets: a round "
                                                                       "whose first pass gives contigs up as dissimilar goes on in units of two targets; C++ host, files from "
                                                                       "the page cache" % (meta["genomes"], rd.group(1) if rd else "?"),
                   "metric": "input Gbases/s (compress path of the C++ host, -m1 rounds: file reading + device parse + match-finding + stream emission)"}
        out.update({
               "value": round(bases / (ms * 1e-3) / 1e9, 4), "unit": "Gbases/s", "matching_ms": ms, "target_bases": bases, "exact_matches": matches,
               "unmatched_chars_before_extensions": unmatched, "final_reference_length": ref_len})
        prof = _tool_profile(r.stderr)
        if prof and prof["resolve"]["ms"] > 0:
            alg = resolve_alg_bytes(bases, bases - unmatched, matches)
            ach = alg / (prof["resolve"]["ms"] * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": KERNEL_OF["resolve"], "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": None, "launches": prof["resolve"]["launches"],
                               "avg_launch_ms": round(prof["resolve"]["ms"] / prof["resolve"]["launches"], 4),
                               "alg_bytes_per_base": round(alg / bases, 3),
                               "share_of_the_matching_phase": round(prof["resolve"]["ms"] / ms, 3)}
            out["kernel_ms_total"] = {k: round(v["ms"], 2) for k, v in prof.items() if v["launches"]}
        return out
    except Exception as e:
        return {"config": "configs[4]", "error": "%s: %s" % (type(e).__name__, e)}


def kernel_source_sha():
    import hashlib
    h = hashlib.sha256()
    for n in ("swsem_resolve4.hip", "swsem_kernels.hip", "swsem_device.h"):
        h.update(open(os.path.join(ROOT, "mbgc_amd", "csrc", n), "rb").read())
    return h.hexdigest()[:16]


def measured_traffic(kernel_prefix, targets_per_launch, n_gpus=1):
    """(bytes per launch, where the figure comes from): the PMC passes are a rocprofv3 run of their own over this command, so
    the line says which commit's kernels they saw and whether the kernel sources have changed since (then: re-take them).
    The counters were taken on ONE GPU with `targets_per_launch` targets in a launch: another launch size, or a rank of a
    sharded run, has no measured figure (null, and the source says why)."""
    try:
        with open(TRAFFIC_FILE) as f:
            t = json.load(f)
        k = t["kernels"][kernel_prefix]
        src = {"file": os.path.relpath(TRAFFIC_FILE, ROOT), "commit": t.get("collected_at_commit"),
               "kernel_sources_changed_since": t.get("kernel_source_sha16") != kernel_source_sha()}
        if src["kernel_sources_changed_since"]:
            print("bench.py: %s was taken from other kernel sources than the ones built now: roofline.traffic is stale" % src["file"], file=sys.stderr)
        if n_gpus != t.get("n_gpus", 1):
            src["not_used"] = "counters were taken on %d GPU(s), this run has %d" % (t.get("n_gpus", 1), n_gpus)
            return None, src
        if t.get("targets_per_launch") != targets_per_launch:
            src["not_used"] = "counters were taken with %s targets per launch, this run has %d" % (t.get("targets_per_launch"), targets_per_launch)
            return None, src
        return int(k["bytes_per_launch"]), src
    except Exception:
        return None, None


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_single_thread(sample_targets, length, emit, max_ref):
    """G0(+RC) preloaded, then `sample_targets` targets matched (matchTexts), emitted (processMatches) and appended
    (loadRef) one after another on one core. Uses the reference's own code (oracle/_ref, prebuilt from /root/reference)
    when it is loadable, else the C restatement."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from mbgc_amd import synth
    import _orc
    kind, refh = "port", None
    try:
        import _refh
        if not _refh.available():
            raise OSError
        _refh.lib()
        refh, kind = _refh, "reference"
    except Exception:
        pass
    base = synth.base_codes(length)
    m = refh.RefMatcher(max_ref) if refh else _orc.OracleMatcher(max_ref)
    m.disable_sliding_window()
    m.load_ref(synth.genome(base, 0), load_rc=True)
    em = None
    if emit:
        em = refh.RefEmitter(m, mode=1, lazy=True, n_targets=1) if refh else _orc.OracleEmitter(m)
    loaded = [m.loading_position()]
    t = 0.0
    for i in range(1, sample_targets + 1):
        g = synth.genome(base, i)
        t0 = time.perf_counter()
        rows = m.match(g)
        if em is not None:
            if refh:
                em.process(rows, g, 0, _orc.NO_LOCK)
            else:
                em.process(rows, g, _orc.NO_LOCK, 128, 0, 0, loaded)
        m.load_ref(g)
        t += time.perf_counter() - t0
    m.close()
    what = "matchTexts + processMatches + loadRef" if emit else "matchTexts + loadRef"
    return dict(value=round(sample_targets * length / t / 1e9, 5), unit="Gbases/s", cores=1, kind=kind,
                sample="G0+RC preloaded, first %d targets of the collection (%.0f Mbases), %s, 1 thread"
                       % (sample_targets, sample_targets * length / 1e6, what))


def cpu_quota():
    """CPUs' worth of time the container grants this process (cgroup v2 cpu.max, v1 cfs quota), None when unlimited"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else round(int(q) / int(per), 2)
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else round(q / per, 2)
    except Exception:
        return None


def cpu_all_cores(sample_targets, length):
    """The reference's own parallel path (MGMP.cpp:520-555: worker threads + finalizer) as `mbgc c -m1` runs it: the
    tool built from /root/reference (oracle/_ref/mbgc) compresses G0 + `sample_targets` FASTA files with every CPU this process is granted (affinity and cgroup quota)
    of this host; the matching phase is what the tool itself reports between "processed reference dataset" and
    "matching finished" (file reading and kseq parsing included, backend compression excluded)."""
    import re
    import shutil
    import tempfile
    from mbgc_amd import synth
    tool = os.path.join(ROOT, "oracle", "_ref", "mbgc")
    if not os.access(tool, os.X_OK):
        return None
    hw = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)   # the hardware threads this process may run on
    quota = cpu_quota()
    # the threads the reference gets: what this process may really use. Under a CPU-time quota (a container's cpu.max) more
    # threads than that only spin against each other — on the MI355X hosts (quota 16 of 256 hardware threads) the
    # reference's matching phase takes 2.7 s with 256 threads and 1.9 s with 16 for the same 128 targets
    cores = max(1, min(hw, int(quota + 0.5))) if quota else hw
    d = tempfile.mkdtemp(prefix="mbgc_cpub_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        base = synth.base_codes(length)
        lst = os.path.join(d, "list.txt")
        with open(lst, "w") as l:
            for i in range(sample_targets + 1):
                fn = os.path.join(d, "s%05d.fa" % i)
                with open(fn, "wb") as f:
                    f.write(synth.fasta_bytes(synth.genome(base, i), i))
                l.write(fn + "\n")
        t0 = time.perf_counter()
        p = subprocess.run([tool, "c", "-m1", "-f", "-t", str(cores), "-T", str(cores), lst, os.path.join(d, "out.mbgc")],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        wall = time.perf_counter() - t0
        if p.returncode != 0:
            return None
        a = re.search(r"processed reference dataset - (\d+) \[ms\]", p.stdout)
        b = re.search(r"matching finished - (\d+) \[ms\]", p.stdout)
        ratio = re.search(r"compressed (\d+) bytes to (\d+)", p.stdout)
        if not (a and b):
            return None
        match_s = (int(b.group(1)) - int(a.group(1))) / 1e3
        out = dict(value=round(sample_targets * length / match_s / 1e9, 5), unit="Gbases/s", cores=cores, kind="reference",
                   sample="`mbgc c -m1 -t %d -T %d` (oracle/_ref, the reference's own OpenMP path) on G0 + the first %d targets "
                          "(%.0f Mbases): matching phase %.2f s, whole run %.2f s wall" %
                          (cores, cores, sample_targets, sample_targets * length / 1e6, match_s, wall),
                   whole_run_value=round(sample_targets * length / wall / 1e9, 5), cpu_model=cpu_model(), nproc=hw, cpu_quota=quota)
        if ratio:
            out["archive_bytes"] = int(ratio.group(2))
        # the same files through this repo's full encode: `mbgc-hip c --backend` (rounds sized by the window, the backend's job table
        # of include/mbgc_backend.h around the reference's own PPMd7 / LZMA leaf coders) — seconds and bytes beside the reference's
        hip, coders = os.path.join(ROOT, "mbgc_amd", "mbgc-hip"), os.path.join(ROOT, "oracle", "_ref", "libmbgc_coders.so")
        if os.access(hip, os.X_OK) and os.path.exists(coders):
            t0 = time.perf_counter()
            q = subprocess.run([hip, "c", "--backend", coders, "--backend-threads", str(cores), lst, os.path.join(d, "hip")],
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
            wall_hip = time.perf_counter() - t0
            mb = re.search(r"backend: (\d+) stream bytes to (\d+) in (\d+) ms", q.stdout)
            mm = re.search(r"matching finished - (\d+) \[ms\]", q.stderr)
            if q.returncode == 0 and mb:
                out["full_encode"] = dict(files=sample_targets + 1, command="mbgc-hip c --backend oracle/_ref/libmbgc_coders.so --backend-threads %d" % cores,
                                          seconds=round(wall_hip, 2), matching_ms=int(mm.group(1)) if mm else None, backend_ms=int(mb.group(3)),
                                          collective_section_bytes=int(mb.group(2)), reference_seconds=round(wall, 2),
                                          reference_archive_bytes=out.get("archive_bytes"),
                                          section_over_reference_archive=round(int(mb.group(2)) / out["archive_bytes"], 4) if out.get("archive_bytes") else None)
        return out
    except Exception:
        return None
    finally:
        shutil.rmtree(d, ignore_errors=True)


def cpu_baseline(sample_targets, length, emit, max_ref):
    """the CPU path on a bounded sample of the same workload, on this host (rank 0, N = 1): the reference's all-core
    path (the headline `value`, cores stated) and the one-thread harness beside it."""
    one = cpu_single_thread(min(sample_targets, 24), length, emit, max_ref)
    allc = cpu_all_cores(sample_targets, length) if emit else None
    if allc is None:
        one["cpu_model"], one["nproc"] = cpu_model(), os.cpu_count()
        return one
    allc["single_thread"] = one
    return allc


def error_line(n_gpus, steps, warm, msg, ranks_seen=None, extra=None):
    """the line rank 0 prints when a run cannot produce a value: the same keys, value null, what went wrong"""
    out = {"metric": "input Gbases/s (compress hot path, -m1: match-finding + stream emission)", "value": None, "unit": "Gbases/s",
           "n_gpus": n_gpus, "steps": steps, "warmup": warm, "ms_per_step": None, "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "u8", "data": "synthetic", "config": {"workload": "configs[%d]" % (3 if n_gpus > 1 else 2)},
           "error": msg, "rccl_ranks_seen": ranks_seen}
    if extra:
        out.update(extra)
    return json.dumps(out)


def spawn_ranks(n, argv=None, grace=5.0, limit=None):
    """`python bench.py --gpus N` outside a launcher: start N ranks of this script as children (this process has not
    touched a GPU and never does), relay rank 0's JSON line, exit with the worst return code.

    A rank that dies must not leave the run hanging: the others would sit inside a collective until the process group's
    timeout, and the caller would get no line at all. Every child is polled; on the first abnormal end the others get
    `grace` seconds to notice by themselves (a gloo peer does, an RCCL peer may not), then SIGTERM, then SIGKILL, and a JSON
    line with "value": null and the error is printed unless rank 0 already printed one. `limit` (MBGC_BENCH_TIMEOUT, seconds) bounds the whole run
    the same way. argv: the children's command (tests pass their own)."""
    import shutil
    import signal
    import tempfile
    import threading
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    if limit is None:
        limit = float(os.environ.get("MBGC_BENCH_TIMEOUT", "540"))
    rundir = tempfile.mkdtemp(prefix="mbgc_bench_", dir=os.environ.get("TMPDIR", "/tmp"))
    cmd = argv if argv is not None else [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", MBGC_BENCH_RUNDIR=rundir)
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      start_new_session=True))                  # (its own process group: workers it forks die with it)
    with open(os.path.join(rundir, "pids"), "w") as f:
        f.write(" ".join(str(p.pid) for p in procs) + "\n")
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()

    def stop(p, sig):
        try:
            os.killpg(p.pid, sig)
        except (ProcessLookupError, PermissionError):
            pass

    def on_signal(signum, _frame):                      # the spawner itself is told to end: the ranks go with it
        for p in procs:
            stop(p, signal.SIGKILL)
        sys.exit(128 + signum)
    for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sg, on_signal)

    t0, failed, failed_at, cause_rc = time.monotonic(), None, None, 0
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        now = time.monotonic()
        if failed is None:
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                r, c = bad[0]
                failed = "rank %d ended %s while the run was going on" % (r, "on signal %d" % -c if c < 0 else "with exit code %d" % c)
                failed_at, cause_rc = now, (c if c > 0 else 128 - c)
            elif now - t0 > limit:
                failed, failed_at = "the run exceeded its limit of %.0f s (MBGC_BENCH_TIMEOUT)" % limit, now - grace
        if failed is not None:
            if now - failed_at > grace + 5.0:
                for p in procs:
                    stop(p, signal.SIGKILL)
            elif now - failed_at > grace:
                for p in procs:
                    stop(p, signal.SIGTERM)
        time.sleep(0.1)
    reader.join(timeout=5)
    out = b"".join(c for c in chunks if c).decode(errors="replace")
    rc = cause_rc                                       # the rank that failed first, not the ones stopped because of it
    for p in procs:
        rc = rc or (p.returncode if p.returncode > 0 else (128 - p.returncode if p.returncode < 0 else 0))
    if failed is None and rc:
        failed = "a rank ended with exit code %d" % rc
    seen = len([f for f in os.listdir(rundir) if f.endswith(".up")])
    shutil.rmtree(rundir, ignore_errors=True)
    # rank 0's own line stands when it carries a value; an error line of its own (it noticed a dead peer inside a collective) gives
    # way to the spawner's, which knows WHICH rank ended first and how — rank 0's message rides along
    kept, rank0_error, has_value = [], None, False
    for l in out.splitlines():
        if l.startswith("{") and '"metric"' in l:
            try:
                d = json.loads(l)
                if d.get("value") is None and d.get("error"):
                    rank0_error = d["error"]
                    continue
                has_value = True
            except ValueError:
                pass
        kept.append(l)
    if kept:
        sys.stdout.write("\n".join(kept) + "\n")
    if failed is not None and not has_value:
        print(error_line(n, 0, 0, failed, ranks_seen=seen, extra={"rank0_error": rank0_error} if rank0_error else None))
    elif failed is None and rank0_error and not has_value:
        print(error_line(n, 0, 0, rank0_error, ranks_seen=seen))
    if failed is not None:
        print("bench.py: " + failed, file=sys.stderr)
    sys.stdout.flush()
    sys.exit(rc or (1 if failed is not None else 0))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--round", type=int, default=0, help="targets per GPU per step (default: what fits the sliding window, at most 64, divided by the GPUs)")
    ap.add_argument("--length", type=int, default=GENOME_LEN)
    ap.add_argument("--cpu-sample", type=int, default=128, help="targets timed on the CPU baseline (0 = skip)")
    ap.add_argument("--check", action="store_true", help="compare the first step's matches with the oracle")
    ap.add_argument("--no-emit", action="store_true", help="matcher only (no stream emission) inside the step")
    ap.add_argument("--no-extras", action="store_true",
                    help="N = 1 only: skip what is measured beside the headline (the constant-step line for the scaling curve, the C++ host "
                         "on the same collection, the other configs' lines)")
    ap.add_argument("--from-host", action="store_true",
                    help="diagnostic, not the headline: every timed round's queries start in pinned host memory and cross PCIe "
                         "inside the timed region (double-buffered on a copy stream); reported under its own metric name")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))

    from mbgc_amd import synth
    from mbgc_amd.rounds import round_schedule
    steps, warm = args.steps, args.warmup
    # the buffer `mbgc c` gives the collection of configs[2] / configs[3] (1000 targets + G0) at every N
    coll = COLLECTION
    max_ref = int(float(os.environ["MBGC_BENCH_MAX_REF"])) if "MBGC_BENCH_MAX_REF" in os.environ else ref_length_limit(1 + coll, args.length)
    cap = targets_in_flight(max_ref, args.length)
    # targets per step. One GPU: what the window lets be in flight (31). Several: the SAME step at every N, so that the curve
    # can be read — the largest multiple of 8 the window allows (24), dealt 12 / 6 / 3 per GPU at N = 2 / 4 / 8; the one-GPU run
    # times that step too, beside its headline ("scaling_reference").
    SCALE_STEP = max(8, cap // 8 * 8)
    if args.round > 0:
        R = args.round
    elif world == 1:
        R = max(1, min(cap, COLLECTION // (steps + warm)))
    else:
        R = max(1, min(SCALE_STEP, COLLECTION // (steps + warm)) // world)
    n_targets = (steps + warm) * R * world
    extras = (world == 1 and rank == 0 and not args.no_extras and not args.from_host and not args.no_emit and args.round <= 0 and
              args.length == GENOME_LEN and "MBGC_BENCH_MAX_REF" not in os.environ and not os.environ.get("MBGC_BENCH_SERIAL"))
    REF_WARM, REF_STEPS = 2, 6
    ref_R = min(SCALE_STEP, R)
    n_ref = (REF_WARM + REF_STEPS) * ref_R if extras and n_targets + (REF_WARM + REF_STEPS) * ref_R <= COLLECTION else 0
    # synthetic collection (SURVEY.md §8d recipe), generated by forked workers BEFORE this process touches the GPU:
    # this rank's targets of every round, one host array per round
    base = synth.base_codes(args.length)
    sched = round_schedule(n_targets, R, world)
    sched_ref = [[[n_targets + r0 + t for t in range(ref_R)]] for r0 in range(0, n_ref, ref_R)]
    mine_all = [1 + t for rnd in sched + sched_ref for t in rnd[rank]]
    if os.environ.get("MBGC_BENCH_SAME"):               # experiment: every target of a round is the same genome (perfect inter-target locality)
        mine_all = [1 + rnd[rank][0] for rnd in sched for t in rnd[rank]]
    fasta_dir = mixed_dir = None
    if extras:
        import tempfile
        fasta_dir = tempfile.mkdtemp(prefix="mbgc_bench_fa_", dir=os.environ.get("TMPDIR", "/tmp"))   # the same genomes as files, for the C++ host
    t_gen = time.perf_counter()
    gens = synth.genomes(base, mine_all, workers=max(1, min(16, (os.cpu_count() or 1) // world)),
                         fork=os.environ.get("MBGC_BENCH_GEN", "fork") != "thread",   # (threads under a profiler that has already opened the GPU)
                         fasta_dir=fasta_dir)
    host_rounds, k = [], 0
    for rnd in sched + sched_ref:
        cnt = len(rnd[rank])
        host_rounds.append(np.concatenate(gens[k: k + cnt]))
        for j in range(k, k + cnt):
            gens[j] = None
        k += cnt
    del gens
    if extras:
        with open(synth.fasta_path(fasta_dir, 0), "wb") as f:
            f.write(synth.fasta_bytes(synth.genome(base, 0), 0))
        try:
            mixed_dir = write_mixed_species(MIXED_GENOMES)
        except Exception as e:
            print("bench.py: no mixed-species files (%s)" % e, file=sys.stderr)
    t_gen = time.perf_counter() - t_gen

    import torch
    import torch.distributed as dist
    # rehearsal hooks (one-GPU box): MBGC_BENCH_ONE_DEVICE=1 puts every rank on cuda:0, MBGC_BENCH_BACKEND=gloo
    # replaces RCCL (which refuses two ranks on one device). The driver's runs use neither.
    if os.environ.get("MBGC_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("MBGC_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # MBGC_ROUNDS_FORCE_EXCHANGE=1 (diagnostics): one rank, but through every collective of the N > 1 protocol — RCCL itself
    # on a one-GPU box, and what the exchange costs a rank before a second GPU is there
    forced = world == 1 and os.environ.get("MBGC_ROUNDS_FORCE_EXCHANGE", "0") == "1"
    if forced:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        dist.init_process_group(backend, init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                                **({"device_id": dev} if backend == "nccl" else {}))
    if world > 1:
        # a rank that never arrives (or dies inside a collective) ends the others after this long, not after torch's ten minutes
        from datetime import timedelta
        pg_timeout = timedelta(seconds=float(os.environ.get("MBGC_BENCH_PG_TIMEOUT", "120")))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=pg_timeout)
        else:
            dist.init_process_group(backend, timeout=pg_timeout)
        if os.environ.get("MBGC_BENCH_RUNDIR"):                      # (spawn_ranks counts these for its error line)
            open(os.path.join(os.environ["MBGC_BENCH_RUNDIR"], "rank%d.up" % rank), "w").close()

    from mbgc_amd import binding
    from mbgc_amd.rounds import RoundRunner

    bit40 = max_ref > 0xFFFFFFFF
    m = binding.SlidingWindowSparseEMMatcher(max_ref, device=local_rank)
    stream = torch.cuda.current_stream()
    m.set_stream(stream.cuda_stream)
    m.set_sliding_window_size(16)
    g0 = torch.from_numpy(synth.genome(base, 0)).to(dev)
    torch.cuda.synchronize()
    m.load_ref_dev(g0.data_ptr(), g0.numel(), True, True, 0)

    # resident in HBM before the clock starts
    bufs = []
    for ri, rnd in enumerate(sched + sched_ref):
        offs = np.arange(len(rnd[rank]) + 1, dtype=np.uint64) * args.length
        bufs.append((torch.from_numpy(host_rounds[ri]).to(dev), offs))
        host_rounds[ri] = None
    torch.cuda.synchronize()
    hostbufs, copy_stream, slots, copied = None, None, None, None
    if args.from_host:
        hostbufs = [b.cpu().pin_memory() for b, _ in bufs]
        copy_stream = torch.cuda.Stream()
        # three slots: round s is matched in one, round s-1's emission still reads the bytes of another (its second phase
        # runs beside round s), round s+1 arrives in the third
        slots = [torch.empty_like(bufs[0][0]) for _ in range(3)]
        copied = [torch.cuda.Event() for _ in range(3)]

    emit = not args.no_emit
    ep = binding.emit_params(1, enable40bitReference=1 if bit40 else 0) if emit else None
    runner = RoundRunner(m, rank, world, None, dev, lazy=True, emit_params=ep, keep_streams=False)
    runner.start()
    tot_matches = 0

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for s in range(warm):
        runner.keep_streams = bool(args.check and s == 0)
        runner.run_round(*bufs[s], next_batch=bufs[s + 1] if s + 1 < len(bufs) else None)
        runner.flush()
        if args.check and s == 0 and rank == 0 and world == 1:
            check_against_oracle(runner, base, sched[0][0], args.length, emit, max_ref)
        runner.keep_streams = False
    m.profile_enable(True)
    dropped0 = m.dropped_bytes()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]      # step boundaries on the main stream
    laps_at = []                                                                   # laps of the circular buffer after each step
    barrier()
    t0 = time.perf_counter()
    marks[0].record()
    replayed = 0
    def stage(r):                                        # host -> device copy of round r's queries, on the copy stream
        with torch.cuda.stream(copy_stream):
            slots[r % 3].copy_(hostbufs[r], non_blocking=True)
            copied[r % 3].record(copy_stream)

    if args.from_host:
        stage(warm)
        for s in range(warm, warm + steps):
            torch.cuda.current_stream().wait_event(copied[s % 3])
            cur = (slots[s % 3], bufs[s][1])
            if s + 1 < warm + steps:
                copy_stream.wait_stream(torch.cuda.current_stream())    # (round s-2, the slot's last reader, is long done)
                stage(s + 1)
            tot_matches += int(runner.run_round(*cur).sum())
            replayed += m.batch_stats()["replayed_blocks"]
            marks[s - warm + 1].record()
            laps_at.append(m.ref_length() == m.max_ref_length())
    for s in range(warm, warm + steps) if not args.from_host else ():
        tot_matches += int(runner.run_round(*bufs[s], next_batch=bufs[s + 1] if s + 1 < len(bufs) else None).sum())
        replayed += m.batch_stats()["replayed_blocks"]
        if os.environ.get("MBGC_BENCH_SERIAL"):         # diagnostics: every kernel alone on the device (the line is marked)
            runner.flush()
            torch.cuda.synchronize()
        marks[s - warm + 1].record()
        laps_at.append(m.ref_length() == m.max_ref_length())
    runner.flush()                                     # the last round's emission (its second phase runs beside the next round)
    barrier()
    dt = time.perf_counter() - t0
    prof = m.profile_get()
    m.profile_enable(False)
    dropped = m.dropped_bytes() - dropped0
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
    # the step of the scaling curve (SCALE_STEP targets whatever N is) on this one GPU: the collection's next rounds, timed the same way
    scaling_ref = None
    if sched_ref:
        first = len(sched)
        d0 = m.dropped_bytes()
        for s_ in range(first, first + REF_WARM):
            runner.run_round(*bufs[s_], next_batch=bufs[s_ + 1])
        runner.flush()
        barrier()
        t1 = time.perf_counter()
        for s_ in range(first + REF_WARM, first + REF_WARM + REF_STEPS):
            runner.run_round(*bufs[s_], next_batch=bufs[s_ + 1] if s_ + 1 < len(bufs) else None)
        runner.flush()
        barrier()
        dt1 = time.perf_counter() - t1
        scaling_ref = {"targets_per_step": ref_R, "steps": REF_STEPS, "warmup": REF_WARM, "ms_per_step": round(dt1 / REF_STEPS * 1e3, 3),
                       "value": round(ref_R * args.length * REF_STEPS / dt1 / 1e9, 4), "unit": "Gbases/s",
                       "extension_bytes_dropped": m.dropped_bytes() - d0,
                       "what": "the step bench.py --gpus N times at every N > 1 (%d targets: the largest multiple of 8 the window allows), on one GPU; "
                               "targets %d..%d of the same collection" % (ref_R, n_targets + 1, n_targets + n_ref)}
    if os.environ.get("MBGC_BENCH_BLOCK_STATS"):          # diagnostics of the last round's resolve blocks, to stderr
        import ctypes as C
        from mbgc_amd import binding as _b
        buf = np.zeros(3 * 200000, dtype=np.uint64)
        nb = C.c_uint64()
        _b.lib().swsem_debug_block_times.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(C.c_uint64)]
        _b.lib().swsem_debug_block_times(m.h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), 200000, C.byref(nb))
        t = buf[: 3 * nb.value].reshape(-1, 3).astype(np.float64)
        print("resolve blocks %d: ticks mean %.0f max %.0f (shader clock), visits %d (mean %.0f/block), rows %d" %
              (nb.value, t[:, 0].mean(), t[:, 0].max(), t[:, 1].sum(), t[:, 1].mean(), t[:, 2].sum()), file=sys.stderr)
        if hasattr(_b.lib(), "swsem_debug_phases"):       # diagnostics build (-DSWSEM_DIAG_PHASES) only
            ph = np.zeros(8, dtype=np.uint64)
            _b.lib().swsem_debug_phases.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
            _b.lib().swsem_debug_phases(m.h, ph.ctypes.data_as(C.POINTER(C.c_uint64)))
            tot, tr, tv, nr, nv, tl, nl, nbk = [float(x) for x in ph]
            print("phases over %d block runs: total %.0f ticks/block; table-gather wait %.1f%% (%.0f refills/block, %.0f ticks each); "
                  "visit-load wait %.1f%% (%.0f visits/block, %.0f ticks each); run continuation %.1f%% (%.1f/block, %.0f ticks each); rest %.1f%%" %
                  (nbk, tot / nbk, 100 * tr / tot, nr / nbk, tr / max(nr, 1), 100 * tv / tot, nv / nbk, tv / max(nv, 1),
                   100 * tl / tot, nl / nbk, tl / max(nl, 1), 100 * (tot - tr - tv - tl) / tot), file=sys.stderr)
        if os.environ.get("MBGC_BENCH_BLOCK_DUMP"):
            np.save(os.environ["MBGC_BENCH_BLOCK_DUMP"], buf[: 3 * nb.value].reshape(-1, 3))
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())

    bases_step = R * world * args.length
    value = bases_step * steps / dt / 1e9
    if rank == 0:
        # the dominant kernel: the families that are a single kernel per launch (emission is 13 small kernels,
        # the second half of which runs beside other work on a second stream and is timed as elapsed time)
        dom = max(("resolve", "stitch", "insert", "load"), key=lambda k: prof[k][0])
        per_launch_ms = {k: (prof[k][0] / prof[k][1] if prof[k][1] else 0.0) for k in prof}
        dom_ms = per_launch_ms[dom]
        launch_bases = R * args.length                     # bases one launch of a family processes (this rank's round)
        alg = dict(ALG_BYTES)
        ach = alg[dom] * launch_bases / (dom_ms * 1e-3) / 1e9 if dom_ms else 0.0
        traffic, traffic_source = measured_traffic(KERNEL_OF[dom].split()[0], R, world)
        pre = [t for t, w in zip(step_ms, laps_at) if not w]
        post = [t for t, w in zip(step_ms, laps_at) if w]
        first_coll = 1 + warm * R * world
        out = {
            "metric": ("input Gbases/s (compress hot path, -m1: match-finding + stream emission)" if emit else
                       "input Gbases/s (compress path, SlidingWindowSparseEMMatcher only)") +
                      (" [queries cross PCIe inside the timed region]" if args.from_host else ""),
            "value": round(value, 4), "unit": "Gbases/s", "n_gpus": world, "steps": steps, "warmup": warm,
            "ms_per_step": round(dt / steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": ("configs[%d]: %d synthetic 5 Mbp genomes @99%% identity%s, %.4g-byte circular reference "
                                    "(the buffer `mbgc c` gives %d files); step = matchTexts%s + loadRef of one round of %d "
                                    "targets%s (the sliding window holds %d); timed: targets %d..%d of the collection, through the buffer's wrap") %
                                   (3 if world > 1 else 2, n_targets, " sharded file-per-GPU over %d GPUs" % world if world > 1 else "", float(max_ref),
                                    1 + coll, " + processMatches (six streams, %s)" %
                                    ("gathered to rank 0 over RCCL" if world > 1 else "left packed in HBM for the host backend") if emit else "",
                                    R * world, " (%d per GPU)" % R if world > 1 else "", cap, first_coll, n_targets),
                       "genome_len": args.length, "targets_per_step": R * world, "targets_in_flight_cap": cap, "max_ref_len": max_ref,
                       "hash_entries": m.hash_size(), "offsets_40bit": bit40,
                       "sharding": "file-per-GPU, all-gather of extensions" if world > 1 else "one GPU"},
            "roofline": {"bound": "hbm", "kernel": KERNEL_OF[dom], "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_source,
                         "traffic_GBs": round(traffic / (dom_ms * 1e-3) / 1e9, 1) if traffic and dom_ms else None,
                         "alg_bytes_per_base": round(alg[dom], 3), "avg_launch_ms": round(dom_ms, 4),
                         "whole_step_frac": round(ALG_BYTES_PER_BASE * value / world / HBM_PEAK_GBS, 5)},
            # reference extension bytes loadRef gave up at the window's end inside the timed region (.cpp:433): a round that fits drops none
            "extension_bytes_dropped_per_step": dropped / steps,
            # per step, this rank: the sharded part (match-finding + stitch + processMatches' first pass), the part every replica
            # repeats (loadRef: copies with the table insertion beside them), and the host's time in the exchange (N > 1, MBGC_ROUNDS_TRACE=1)
            "sharded_ms": round(per_launch_ms["resolve"] + per_launch_ms["stitch"] + per_launch_ms["emit"], 4),
            "replicated_ms": round(max(per_launch_ms["insert"], per_launch_ms["load"]), 4),
            "exchange_ms": (round(sum(v for k, v in runner.trace.items() if k == "top" or k.startswith("flush")) * 1e3 / (steps + warm), 4)
                            if runner.trace is not None and (world > 1 or forced) else None),
            "kernel_ms_per_launch": {k: round(v, 4) for k, v in per_launch_ms.items() if k != "probe"},
            "kernel_ms_total": {k: round(prof[k][0], 3) for k in prof if k != "probe"}, "dominant_kernel": dom,
            "ms_per_step_before_wrap": round(float(np.mean(pre)), 4) if pre else None, "steps_before_wrap": len(pre),
            "ms_per_step_after_wrap": round(float(np.mean(post)), 4) if post else None, "steps_after_wrap": len(post),
            "matches_per_step": tot_matches // steps, "replayed_resolve_blocks_per_step": replayed / steps,
            "stream_bytes_per_step": runner.stream_bytes // max(1, steps + warm),
            "round_finalizes_queued_behind_pass1": {"tried": runner.spec_local[0], "applied": runner.spec_local[1], "not_applied_at_try": runner.spec_local[2][:16]},
            "step_ms_min_median_max": [round(float(np.min(step_ms)), 3), round(float(np.median(step_ms)), 3), round(float(np.max(step_ms)), 3)],
            **({"step_ms": [round(t, 3) for t in step_ms]} if os.environ.get("MBGC_BENCH_STEP_MS") else {}),
            **({"scaling_reference": scaling_ref} if scaling_ref else {}),
            **({"INVALID": "MBGC_BENCH_SERIAL: the device was drained after every step (diagnostics)"} if os.environ.get("MBGC_BENCH_SERIAL") else {}),
        }
        if os.environ.get("MBGC_BENCH_BLOCK_TIMES"):                  # diagnostics: how even the last launch's resolve blocks were
            import ctypes
            L = binding.lib()
            L.swsem_debug_block_times.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
            cap = 1 << 18
            buf = np.zeros((cap, 3), dtype=np.uint64)
            nb = ctypes.c_uint64()
            if L.swsem_debug_block_times(m.h, buf.ctypes.data, cap, ctypes.byref(nb)) == 0:
                if os.environ.get("MBGC_BENCH_BLOCK_TIMES") not in ("1", ""):
                    np.save(os.environ["MBGC_BENCH_BLOCK_TIMES"], buf[: min(int(nb.value), cap)])
                t = np.sort(buf[: min(int(nb.value), cap), 0].astype(np.float64))
                if len(t):
                    out["resolve_block_ticks"] = {"blocks": int(nb.value), "mean": float(t.mean()), "median": float(t[len(t) // 2]),
                                                  "p90": float(t[int(len(t) * 0.9)]), "p99": float(t[int(len(t) * 0.99)]), "max": float(t[-1])}
        if os.environ.get("MBGC_BENCH_EMIT_STATS"):                   # diagnostics: the emission's pairing chain over the whole run
            out["pairing_chain"] = m.emit_stats()
        if os.environ.get("MBGC_BENCH_OCC"):                          # diagnostics: share of the table's buckets that hold an entry at the end of the run
            out["table_occupancy"] = round(float(np.count_nonzero(m.ht())) / m.hash_size(), 5)
        if world > 1 or forced:
            out["rccl_ranks_seen"] = dist.get_world_size()
            out["extension_allgathers_started_ahead"] = {"started": runner.pregathers[0], "used": runner.pregathers[1]}
            out["round_finalizes_queued_on_device_verdicts"] = {"tried": runner.spec_rounds[0], "applied": runner.spec_rounds[1]}
            out["extension_exchanges_cut_to_the_loadable_head"] = {"rounds": runner.head_gathers[0], "bytes_asked_for": runner.head_gathers[1],
                                                                   "bytes_of_those_rounds": runner.head_gathers[2]}
        if runner.trace is not None:
            out["host_ms_per_round"] = {k: round(v * 1e3 / (steps + warm), 3) for k, v in runner.trace.items()}
        if extras:
            # beside the headline, never inside its timed region: the product's C++ host on the same collection, the other configs
            runner = None
            m.close()
            del bufs
            torch.cuda.empty_cache()
            try:
                out["cpp_host"] = cpp_host_line(fasta_dir, n_targets, warm, value)
                out["configs"] = [config1_line(), config4_line(mixed_dir), config4_line(mixed_dir, rounds=True)]
            finally:
                import shutil
                for d_ in (fasta_dir, mixed_dir):
                    if d_:
                        shutil.rmtree(d_, ignore_errors=True)
        if world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.length, emit, max_ref)
        if dropped and not os.environ.get("MBGC_BENCH_ALLOW_DROPS"):
            # a round larger than the window discards part of its reference extensions: less replicated work, a worse ratio —
            # not the workload. Such a line is marked, never the headline.
            out["metric"] = "INVALID (reference extension bytes dropped at the sliding window's end): " + out["metric"]
            out["value"] = None
        print(json.dumps(out), flush=True)
    if world > 1 or forced:
        dist.destroy_process_group()


def check_against_oracle(runner, base, targets, length, emit, max_ref):
    """first round: the six streams (or, matcher only, the hash-table image) against the oracle driven
    through the reference's target loop (tests/_driver.py)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _driver
    import _orc
    from mbgc_amd import synth
    o = _orc.OracleMatcher(max_ref)
    if emit:
        op = _orc.emit_params(1, enable40bitReference=1 if max_ref > 0xFFFFFFFF else 0)
        res = _driver.encode_rounds(o, lambda: _orc.OracleEmitter(o, op), [synth.genome(base, 0)],
                                    [[synth.genome(base, 1 + t)] for t in targets], len(targets))
        for k, v in res["streams"].items():
            assert bytes(runner.streams[k]) == v, "stream %s differs from the oracle" % k
        assert bytes(runner.locks_stream) == res["locks"]
    else:
        o.set_sliding_window_size(16)
        o.load_ref(synth.genome(base, 0), load_rc=True)
        locks = [o.acquire_lock() for _ in targets]
        for t, lk in zip(targets, locks):
            o.load_ref(synth.genome(base, 1 + t))
            o.load_separator(0)
            o.release_lock(lk)
    assert np.array_equal(runner.m.ht(), o.ht()), "hash-table image differs from the oracle"
    print("check: first round identical to the oracle (%d targets, streams + hash table)" % len(targets), file=sys.stderr)


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException as e:                                         # rank 0 still prints a line: value null and what went wrong
        import traceback
        traceback.print_exc()
        if int(os.environ.get("RANK", "0")) == 0 and not isinstance(e, KeyboardInterrupt):
            seen = None
            try:
                import torch.distributed as dist
                seen = dist.get_world_size() if dist.is_initialized() else 0
            except Exception:
                pass
            print(error_line(int(os.environ.get("WORLD_SIZE", "1")), 0, 0, "%s: %s" % (type(e).__name__, e), ranks_seen=seen), flush=True)
        sys.stdout.flush()
        os._exit(1)                                                     # (no destructors of a half-dead process group: they can hang)

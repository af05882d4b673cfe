#!/usr/bin/env python3
"""Headline benchmark: input Gbases/s of the MI355X match-finding path (BASELINE.json `metric`).

Workload (BASELINE.json configs[1]): 128 synthetic 5 Mbp genomes at 99 % identity matched by the
SlidingWindowSparseEMMatcher path against the growing reference (G0 + reverse complement preloaded,
1.28e9-byte circular buffer, 2^27-entry table: the sizes `mbgc c` derives for 128 files,
MGMP.cpp:130-168). A *step* is one round: every GPU matches `--round` targets (default 16) against its
frozen replica, then every replica loads the round's extensions in target order (hash insertion
included). With N > 1 the targets are sharded file-per-GPU and the extension bytes are all-gathered
over RCCL; per-GPU work is fixed, so scaling is weak. Inputs are resident in HBM before the timed
region. One JSON line is printed by rank 0.

Diagnostics (never the headline): --no-emit (matcher only), --from-host (queries cross PCIe inside the timed region),
--check (first round against the oracle); environment: MBGC_BENCH_BLOCK_STATS / MBGC_BENCH_BLOCK_DUMP (per-block clocks
of the last resolve launch), MBGC_BENCH_NO_LOOKAHEAD, MBGC_BENCH_ONE_DEVICE + MBGC_BENCH_BACKEND=gloo (several ranks
on one GPU, a rehearsal of the N > 1 protocol), and the library's own switches (SWSEM_RB, SWSEM_PROBE, SWSEM_HASH,
SWSEM_RESOLVE, SWSEM_PROF_FAMS; see mbgc_amd/csrc/swsem_runtime.hip: swsem_create)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GENOME_LEN = 5_000_000
# 128 files -> referenceFactor 128 -> 128 * 5e6 * 2 (MGMP.cpp:130-158). MBGC_BENCH_MAX_REF (diagnostic): a smaller buffer,
# so that the circular reference wraps inside the run
MAX_REF_LEN = int(float(os.environ.get("MBGC_BENCH_MAX_REF", 1_280_000_000)))


def ref_length_limit(files_count, basic_len):
    """the reference buffer `mbgc c` derives for a collection (loadG0Ref MGMP.cpp:130-134, initMatcher :152-168; -m1, RC
    in the reference, circular): 129 files of 5 Mbp give configs[1]'s 1.28e9 bytes; the larger collections of the
    weak-scaling runs (N x 128 targets) get what the tool would give them (2.56e9 for 257..2048 files)."""
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
#   resolve k_resolve_blocks: the query scan (1 B: the scan windows hash their K-mers from the query bytes; with
#           SWSEM_HASH=pre that byte belongs to the "probe" family, k_probe<true>, instead), the table probes the
#           sequential loop performs (0.29 x 4 B), the reference bytes it compares (0.918 B) and the match rows it
#           writes (24 B x 766 k rows / 80 M bases)
#   load    extension copy, read + write (2 B);  insert  one 4-B table entry per 16 bases
#   emit    the six streams (0.14 B)
ALG_BYTES = {"probe": 1.0, "resolve": 4 * 0.29 + 0.918 + 0.23, "stitch": 0.23, "load": 2.0, "insert": 0.25, "emit": 0.14}
QUERY_SCAN_BYTES = 1.0               # moves to "resolve" when no hash kernel ran
# HBM bytes per input base each family really moves, from the PMC passes in profiles/r01_pmc_hbm_traffic.json
# (FETCH_SIZE + WRITE_SIZE of its largest launches = rounds of 16 x 5 Mbp, / 80 M bases)
TRAFFIC = {"probe": (41035.8 + 313186.5) * 1024 / 80e6, "resolve": (2958811.0 + 50837.9) * 1024 / 80e6,
           "stitch": (793.8 + 101.8 + 117.8 + 111.9 + 13241.4 + 18336.5) * 1024 / 80e6, "load": (39692.8 + 78735.5) * 1024 / 80e6,
           "insert": (353604.2 + 153432.2) * 1024 / 80e6,
           "emit": (24414.1 + 8602.6 + 1492.6 + 0.0 + 17843.9 + 33364.6 + 62.0 + 1.0 + 9646.2 + 1149.9 + 25.8 + 13.8 + 12625.5 +
                    118.3 + 3.2 + 0.8 + 21404.4 + 16375.0 + 62.4 + 65.3 + 21189.8 + 127.9 + 102860.9 + 16637.0 + 136645.7 +
                    30694.6) * 1024 / 80e6}
KERNEL_OF = {"probe": "k_probe<true> (K-mer hashes of the query, SWSEM_HASH=pre only)", "resolve": "k_resolve_blocks<2, false>",
             "stitch": "k_stitch_pre + k_stitch + k_gather", "load": "k_copy_multi", "insert": "k_insert_multi",
             "emit": "k_emit_* (13 launches)"}


def cpu_baseline(sample_targets, length, emit):
    """Time the CPU path on a bounded sample of the same workload (rank 0, N = 1): G0(+RC) preloaded,
    then `sample_targets` targets matched (matchTexts), emitted (processMatches) and appended (loadRef)
    one after another on one core. Uses the reference's own code (oracle/_ref, prebuilt from
    /root/reference) when it is loadable, else the C restatement."""
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
    m = refh.RefMatcher(MAX_REF_LEN) if refh else _orc.OracleMatcher(MAX_REF_LEN)
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
    return dict(value=sample_targets * length / t / 1e9, unit="Gbases/s", cores=1, kind=kind,
                sample="G0+RC preloaded, first %d of the 128 targets (%.0f Mbases), %s, 1 thread"
                       % (sample_targets, sample_targets * length / 1e6, what))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=7)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--round", type=int, default=16, help="targets per GPU per step")
    ap.add_argument("--length", type=int, default=GENOME_LEN)
    ap.add_argument("--cpu-sample", type=int, default=24, help="targets timed on the CPU baseline (0 = skip)")
    ap.add_argument("--check", action="store_true", help="compare the first step's matches with the oracle")
    ap.add_argument("--no-emit", action="store_true", help="matcher only (no stream emission) inside the step")
    ap.add_argument("--from-host", action="store_true",
                    help="diagnostic, not the headline: every timed round's queries start in pinned host memory and cross PCIe "
                         "inside the timed region (double-buffered on a copy stream); reported under its own metric name")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d): launch with torch.distributed.run" % (world, args.gpus))
    # rehearsal hooks (one-GPU box): MBGC_BENCH_ONE_DEVICE=1 puts every rank on cuda:0, MBGC_BENCH_BACKEND=gloo
    # replaces RCCL (which refuses two ranks on one device). The driver's runs use neither.
    if os.environ.get("MBGC_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("MBGC_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from mbgc_amd import binding, synth
    from mbgc_amd.rounds import RoundRunner, round_schedule

    R, steps, warm = args.round, args.steps, args.warmup
    n_targets = (steps + warm) * R * world
    base = synth.base_codes(args.length)
    # the buffer `mbgc c` would give this collection (129 files at the defaults: configs[1]'s 1.28e9 bytes)
    max_ref = MAX_REF_LEN if "MBGC_BENCH_MAX_REF" in os.environ else ref_length_limit(1 + n_targets, args.length)
    m = binding.SlidingWindowSparseEMMatcher(max_ref, device=local_rank)
    stream = torch.cuda.current_stream()
    m.set_stream(stream.cuda_stream)
    m.set_sliding_window_size(16)
    g0 = torch.from_numpy(synth.genome(base, 0)).to(dev)
    torch.cuda.synchronize()
    m.load_ref_dev(g0.data_ptr(), g0.numel(), True, True, 0)

    # this rank's targets of every round, resident in HBM before the clock starts
    sched = round_schedule(n_targets, R, world)
    bufs = []
    for rnd in sched:
        mine = rnd[rank]
        arr = np.concatenate([synth.genome(base, 1 + t) for t in mine])
        offs = np.arange(len(mine) + 1, dtype=np.uint64) * args.length
        bufs.append((torch.from_numpy(arr).to(dev), offs))
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
    runner = RoundRunner(m, rank, world, None, dev, lazy=True, emit_params=binding.emit_params(1) if emit else None,
                         keep_streams=False)
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
    barrier()
    t0 = time.perf_counter()
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
    for s in range(warm, warm + steps) if not args.from_host else ():
        # every timed step also hashes a following round's queries (the last one a round that is not matched here)
        tot_matches += int(runner.run_round(*bufs[s], next_batch=None if os.environ.get('MBGC_BENCH_NO_LOOKAHEAD') else bufs[(s + 1) % len(bufs)]).sum())
        replayed += m.batch_stats()["replayed_blocks"]
    runner.flush()                                     # the last round's emission (its second phase runs beside the next round)
    barrier()
    dt = time.perf_counter() - t0
    prof = m.profile_get()
    m.profile_enable(False)
    if os.environ.get("MBGC_BENCH_BLOCK_STATS"):          # diagnostics of the last round's resolve blocks, to stderr
        import ctypes as C
        from mbgc_amd import binding as _b
        buf = np.zeros(3 * 200000, dtype=np.uint64)
        nb = C.c_uint64()
        _b.lib().swsem_debug_block_times.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(C.c_uint64)]
        _b.lib().swsem_debug_block_times(m.h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), 200000, C.byref(nb))
        t = buf[: 3 * nb.value].reshape(-1, 3).astype(np.float64)
        print("resolve blocks %d: ticks mean %.0f max %.0f (100 MHz), visits %d (mean %.0f/block), rows %d" %
              (nb.value, t[:, 0].mean(), t[:, 0].max(), t[:, 1].sum(), t[:, 1].mean(), t[:, 2].sum()), file=sys.stderr)
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
        dom = max(("probe", "resolve", "stitch", "insert", "load"), key=lambda k: prof[k][0])
        per_launch_ms = {k: (prof[k][0] / prof[k][1] if prof[k][1] else 0.0) for k in prof}
        dom_ms = per_launch_ms[dom]
        launch_bases = R * args.length                     # bases one launch of a family processes (this rank's round)
        alg = dict(ALG_BYTES)
        if prof["probe"][1] == 0:                          # no hash kernel ran: the chains read the query themselves
            alg["resolve"] += QUERY_SCAN_BYTES
        ach = alg[dom] * launch_bases / (dom_ms * 1e-3) / 1e9 if dom_ms else 0.0
        out = {
            "metric": ("input Gbases/s (compress hot path, -m1: match-finding + stream emission)" if emit else
                       "input Gbases/s (compress path, SlidingWindowSparseEMMatcher only)") +
                      (" [queries cross PCIe inside the timed region]" if args.from_host else ""),
            "value": round(value, 4), "unit": "Gbases/s", "n_gpus": world, "steps": steps, "warmup": warm,
            "ms_per_step": round(dt / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": ("configs[1]: %d synthetic 5 Mbp genomes @99%% identity, %.3g-byte reference; step = "
                                    "matchTexts%s + loadRef of one round of %d targets/GPU") %
                                   (n_targets, float(max_ref), " + processMatches (six streams, gathered to rank 0)" if emit else "", R),
                       "genome_len": args.length, "targets_per_step": R * world, "max_ref_len": max_ref,
                       "hash_entries": m.hash_size(), "sharding": "file-per-GPU, all-gather of extensions"},
            "roofline": {"bound": "hbm", "kernel": KERNEL_OF[dom], "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
                         "traffic": round(TRAFFIC[dom] * launch_bases),
                         "traffic_GBs": round(TRAFFIC[dom] * launch_bases / (dom_ms * 1e-3) / 1e9, 1) if dom_ms else 0.0,
                         "alg_bytes_per_base": round(alg[dom], 3), "avg_launch_ms": round(dom_ms, 4),
                         "whole_step_frac": round(ALG_BYTES_PER_BASE * value / world / HBM_PEAK_GBS, 5)},
            "kernel_ms_per_launch": {k: round(v, 4) for k, v in per_launch_ms.items()},
            "kernel_ms_total": {k: round(prof[k][0], 3) for k in prof}, "dominant_kernel": dom,
            "matches_per_step": tot_matches // steps, "replayed_resolve_blocks_per_step": replayed / steps, "stream_bytes_per_step": runner.stream_bytes // max(1, steps + warm),
        }
        if world > 1:
            out["extension_allgathers_started_ahead"] = {"started": runner.pregathers[0], "used": runner.pregathers[1]}
        if world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.length, emit)
        print(json.dumps(out), flush=True)
    if world > 1:
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
        res = _driver.encode_rounds(o, lambda: _orc.OracleEmitter(o), [synth.genome(base, 0)],
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
    main()

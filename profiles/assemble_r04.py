#!/usr/bin/env python3
"""Turns gpurun_out/r04final (profiles/collect_r04.sh) into the tracked files under profiles/: the bench lines, rocprofv3's
per-kernel averages over the timed steps, the PMC traffic per launch (which bench.py reads for roofline.traffic) and the
SQ counters of the dominant kernel."""
import json
import os
import shutil
import sys

O = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r04final/"
P = "profiles/"
STEPS = 20


def kernel_source_sha():
    """the sources of the kernels these counters were taken from: bench.py says so when they have changed since"""
    import hashlib
    h = hashlib.sha256()
    for n in ("swsem_resolve4.hip", "swsem_kernels.hip", "swsem_device.h"):
        h.update(open(os.path.join("mbgc_amd", "csrc", n), "rb").read())
    return h.hexdigest()[:16]


def line(name):
    return json.loads([l for l in open(O + name) if l.startswith("{")][-1])


for src, dst in (("bench_full.json", "r04_bench_full_path.json"), ("bench_matcher.json", "r04_bench_matcher_only.json"),
                 ("bench_from_host.json", "r04_bench_from_host.json"), ("bench_rocprof.json", "r04_bench_under_rocprof.json"),
                 ("bench_n2_gloo.json", "r04_bench_two_ranks_one_gpu_gloo.json"), ("bench_round40.json", "r04_bench_rounds_of_40_dropping_bytes.json")):
    json.dump(line(src), open(P + dst, "w"), indent=1)
shutil.copy(O + "kernel_stats.csv", P + "r04_bench_kernel_stats.csv")
shutil.copy(O + "timed.json", P + "r04_bench_timed_kernel_avgs.json")
full = line("bench_full.json")
pre = full["steps_before_wrap"]                    # timed steps whose round ended before the wrap; the wrapping round itself still resolves pre-wrap
f, w, t = (json.load(open(O + n + ".json")) for n in ("FETCH_SIZE", "WRITE_SIZE", "tcc"))


def timed(vals, k):
    v = vals["values"][-k:] if k else []
    return sum(v) / len(v) if v else 0.0


def per_launch(name, counter_file, counter, k):
    e = counter_file.get(name, {}).get(counter)
    return timed(e, k) if e else 0.0


kernels = {}
names = sorted(set(f) | set(w))
res_false = [n for n in names if "k_resolve_blocks4<false>" in n]
res_true = [n for n in names if "k_resolve_blocks4<true>" in n]
n_false, n_true = pre + 1, STEPS - pre - 1
acc = {"fetch": 0.0, "write": 0.0, "hit": 0.0, "miss": 0.0}
for grp, k in ((res_false, n_false), (res_true, n_true)):
    for n in grp:
        acc["fetch"] += k * per_launch(n, f, "FETCH_SIZE", k)
        acc["write"] += k * per_launch(n, w, "WRITE_SIZE", k)
        acc["hit"] += k * per_launch(n, t, "TCC_HIT_sum", k)
        acc["miss"] += k * per_launch(n, t, "TCC_MISS_sum", k)
kernels["k_resolve_blocks4"] = {
    "bytes_per_launch": round((acc["fetch"] + acc["write"]) * 1024 / STEPS), "fetch_bytes": round(acc["fetch"] * 1024 / STEPS),
    "write_bytes": round(acc["write"] * 1024 / STEPS), "tcc_hit": round(acc["hit"] / STEPS), "tcc_miss": round(acc["miss"] / STEPS),
    "launches_averaged": {"k_resolve_blocks4<false>": n_false, "k_resolve_blocks4<true>": n_true}}
for key, pat, x2 in (("k_insert_multi", "k_insert_multi", False), ("k_copy_multi", "k_copy_multi", True), ("k_stitch_pre + k_stitch + k_gather", None, False)):
    if pat is None:
        grp = [n for n in names if any(s in n for s in ("k_stitch_pre", "k_stitch<", "k_gather"))]
    else:
        grp = [n for n in names if pat in n]
    fe = sum(per_launch(n, f, "FETCH_SIZE", STEPS) for n in grp) * 1024
    wr = sum(per_launch(n, w, "WRITE_SIZE", STEPS) for n in grp) * 1024
    kernels[key] = {"bytes_per_launch": round((2 * fe if x2 else fe) + wr), "fetch_bytes_raw": round(fe), "write_bytes": round(wr),
                    "fetch_doubled": x2}
out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum (separate passes) --kernel-trace over `python3 bench.py --cpu-sample 0` "
               "(the headline command: 1000 genomes, rounds of 31); bytes per launch averaged over the launches of the 20 timed steps. FETCH_SIZE counts "
               "64-byte requests of the L2 to the fabric (Infinity-Cache hits included); MI355X_MICROARCH.md's x2 correction applies to wide coalesced "
               "streaming reads (k_copy_multi: applied) and not to scattered 8-byte gathers (calibrated in round 1: 80 M independent gathers read 64 B each); "
               "k_resolve_blocks4 is 3/4 scattered gathers and 1/4 16-byte-per-lane window reads: its figure is the raw count, i.e. a lower bound (+ <= 25 %).",
       "command": "python3 bench.py --cpu-sample 0", "targets_per_launch": full["config"]["targets_per_step"], "n_gpus": 1, "kernels": kernels,
       "kernel_source_sha16": kernel_source_sha(), "collected_at_commit": os.popen("git rev-parse --short HEAD").read().strip()}
json.dump(out, open(P + "r04_pmc_traffic.json", "w"), indent=1)
sq = {}
for n in ("sq1", "sq2"):
    d = json.load(open(O + n + ".json"))
    for k, cs in d.items():
        if "k_resolve_blocks4" in k:
            sq.setdefault(k, {}).update({c: round(v["max"]) for c, v in cs.items()})
json.dump({"note": "rocprofv3 --pmc (two passes of SQ counters) --kernel-trace, `python3 bench.py --cpu-sample 0 --steps 3 --warmup 1`; per launch, max over "
                   "the launches. SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count in units of 4 cycles summed over waves.", "kernels": sq},
          open(P + "r04_resolve_sq_counters.json", "w"), indent=1)
shutil.copy(O + "timeline.txt", P + "r04_step_timeline.txt")
json.dump(json.loads(open(O + "cpp_host_mixed.json").read().strip().splitlines()[-1]), open(P + "r04_cpp_host_mixed_species.json", "w"), indent=1)
shutil.copy(O + "mixed_timeline.txt", P + "r04_cpp_host_mixed_species_timeline.txt")
print(json.dumps(kernels["k_resolve_blocks4"]))

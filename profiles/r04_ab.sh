# three runs of the headline bench under the given environment: bash profiles/r04_ab.sh <tag> [VAR=value ...]; prints ms per step of each and the median
R=$GRAFT_REPO_ROOT
tag=$1; shift
for kv in "$@"; do export "$kv"; done
mkdir -p $R/gpurun_out/ab
for i in 1 2 3; do
  MBGC_BENCH_STEP_MS=1 timeout -k 10 300 python3 $R/bench.py --cpu-sample 0 > $R/gpurun_out/ab/${tag}_$i.json 2> $R/gpurun_out/ab/${tag}_$i.err || exit 1
done
python3 - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
import os
R = os.environ["GRAFT_REPO_ROOT"]
ds = [json.load(open("%s/gpurun_out/ab/%s_%d.json" % (R, tag, i))) for i in (1, 2, 3)]
ms = sorted(d["ms_per_step"] for d in ds)
print(tag, "ms/step", [d["ms_per_step"] for d in ds], "median", ms[1], "Gbases/s", sorted(d["value"] for d in ds)[1],
      "resolve", [d["kernel_ms_per_launch"]["resolve"] for d in ds], "insert", [d["kernel_ms_per_launch"]["insert"] for d in ds],
      "pre/post wrap", [(d["ms_per_step_before_wrap"], d["ms_per_step_after_wrap"]) for d in ds])
PY

# one step's kernel timeline under rocprofv3: bash profiles/r04_timeline.sh <tag> [env assignments...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
for kv in "$@"; do export "$kv"; done
O=$R/gpurun_out/tl_$tag
rm -rf $O; mkdir -p $O
export MBGC_BENCH_GEN=thread
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/bench.py --cpu-sample 0 --steps 8 --warmup 5 > $O/bench.json 2>$O/kt.err || exit 1
python3 $R/profiles/timeline.py $O/kt > $R/gpurun_out/tl_$tag.txt
rm -rf $O/kt

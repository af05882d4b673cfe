cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
for kv in "$@"; do export "$kv"; done
O=$R/gpurun_out/st_$tag
rm -rf $O; mkdir -p $O
export MBGC_BENCH_GEN=thread
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/bench.py --cpu-sample 0 > $O/bench.json 2>$O/kt.err || exit 1
python3 $R/profiles/r04_steps.py $O/kt > $R/gpurun_out/st_$tag.txt
rm -rf $O/kt

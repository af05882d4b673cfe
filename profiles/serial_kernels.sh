cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/serial; rm -rf $O; mkdir -p $O
export MBGC_BENCH_GEN=thread MBGC_BENCH_SERIAL=1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --cpu-sample 0 --steps 10 --warmup 5 > $O/bench.json 2>$O/err.txt || exit 1
cd $R && python3 profiles/timed_stats.py $O/kt 10 > $O/timed.json; rm -rf $O/kt; cat $O/timed.json | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,v in sorted(d.items(), key=lambda kv:-kv[1]['avg_us_last_10']): print('%8.1f  %s'%(v['avg_us_last_10'],k[:90]))
"

# round 2, first GPU session: parity at configs[2], then launch order A/B (contig-major vs offset-major on one XCD)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02a
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
SWSEM_ORDER=contig MBGC_BENCH_BLOCK_STATS=1 timeout -k 10 300 python3 bench.py --cpu-sample 0 > $O/bench_contig.json 2>$O/bench_contig.err || exit 1
SWSEM_ORDER=xcd MBGC_BENCH_BLOCK_STATS=1 timeout -k 10 300 python3 bench.py --cpu-sample 0 > $O/bench_xcd.json 2>$O/bench_xcd.err || exit 1
cat $O/bench_contig.json $O/bench_xcd.json | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['value'], d['ms_per_step'], d['ms_per_step_before_wrap'], d['ms_per_step_after_wrap'], d['kernel_ms_per_launch'])"
cat $O/bench_contig.err $O/bench_xcd.err | grep -v amdgpu.ids
export MBGC_BENCH_GEN=thread
cd /tmp
for ord in contig xcd; do
  export SWSEM_ORDER=$ord
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$ord -- python3 $R/bench.py --cpu-sample 0 --steps 3 --warmup 1 --round 40 > $O/pmc_fetch_$ord.out 2>$O/pmc_fetch_$ord.err || exit 1
  timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_tcc_$ord -- python3 $R/bench.py --cpu-sample 0 --steps 3 --warmup 1 --round 40 > $O/pmc_tcc_$ord.out 2>$O/pmc_tcc_$ord.err || exit 1
  python3 $R/profiles/pmc_summary.py $O/pmc_fetch_$ord > $O/fetch_$ord.json
  python3 $R/profiles/pmc_summary.py $O/pmc_tcc_$ord > $O/tcc_$ord.json
  rm -rf $O/pmc_fetch_$ord $O/pmc_tcc_$ord
done
export SWSEM_ORDER=xcd
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --cpu-sample 0 > $O/bench_rocprof.json 2>$O/kt.err || exit 1
cd $R
python3 profiles/timed_stats.py $O/kt 20 > $O/timed.json
python3 profiles/kernel_stats.py $O/kt
python3 -c "
import json
for o in ('contig','xcd'):
    f=json.load(open('$O/fetch_%s.json'%o)); t=json.load(open('$O/tcc_%s.json'%o))
    for k in f:
        if 'resolve_blocks' in k or 'insert_multi' in k: print(o, k[:40], f[k], t.get(k))
"

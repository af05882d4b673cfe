set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r01f
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 300 python3 bench.py --check > $O/bench_check.json 2>$O/bench_check.err
timeout -k 10 300 python3 bench.py > $O/bench_full.json 2>$O/bench_full.err
timeout -k 10 200 python3 bench.py --no-emit --cpu-sample 0 > $O/bench_matcher.json 2>$O/bench_matcher.err
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --cpu-sample 0 > $O/bench_rocprof.json 2>$O/kt.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --cpu-sample 0 --steps 3 > $O/pmc_fetch.out 2>$O/pmc_fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --cpu-sample 0 --steps 3 > $O/pmc_write.out 2>$O/pmc_write.err
cd $R
python3 profiles/timed_stats.py $O/kt 7 > $O/timed.json
python3 profiles/pmc_summary.py $O/pmc_fetch > $O/fetch.json
python3 profiles/pmc_summary.py $O/pmc_write > $O/write.json
python3 profiles/kernel_stats.py $O/kt
cat $O/bench_full.json | tail -1

# round 4 measurements, two gpurun calls: bash profiles/collect_r04.sh a ; bash profiles/collect_r04.sh b  (then python3 profiles/assemble_r04.py here)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04final
mkdir -p $O
cd $R
if [ "$1" = a ]; then
timeout -k 10 600 python3 bench.py > $O/bench_full.json 2>$O/bench_full.err || exit 1
timeout -k 10 300 python3 bench.py --check --steps 2 --warmup 1 --cpu-sample 0 --no-extras > $O/bench_check.json 2>$O/bench_check.err || exit 1
timeout -k 10 300 python3 bench.py --no-emit --cpu-sample 0 --no-extras > $O/bench_matcher.json 2>$O/bench_matcher.err || exit 1
MBGC_BENCH_ALLOW_DROPS=1 timeout -k 10 300 python3 bench.py --round 40 --cpu-sample 0 --no-extras > $O/bench_round40.json 2>$O/bench_round40.err || exit 1
timeout -k 10 300 python3 bench.py --from-host --cpu-sample 0 --no-extras > $O/bench_from_host.json 2>$O/bench_from_host.err || exit 1
MBGC_BENCH_ONE_DEVICE=1 MBGC_BENCH_BACKEND=gloo timeout -k 10 500 python3 bench.py --gpus 2 --steps 6 --warmup 3 --cpu-sample 0 --no-extras > $O/bench_n2_gloo.json 2>$O/bench_n2_gloo.err || exit 1
fi
export MBGC_BENCH_GEN=thread
cd /tmp
if [ "$1" = a ]; then exit 0; fi
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --cpu-sample 0 --no-extras > $O/bench_rocprof.json 2>$O/kt.err || exit 1
python3 $R/profiles/timeline.py $O/kt > $O/timeline.txt 2>/dev/null
python3 $R/profiles/timed_stats.py $O/kt 20 > $O/timed.json
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv; rm -rf $O/kt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --cpu-sample 0 --no-extras > $O/pmc_$c.out 2>$O/pmc_$c.err || exit 1
done
timeout -k 10 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_tcc -- python3 $R/bench.py --cpu-sample 0 --no-extras > $O/pmc_tcc.out 2>$O/pmc_tcc.err || exit 1
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_sq1 -- python3 $R/bench.py --cpu-sample 0 --no-extras --steps 3 --warmup 1 > $O/pmc_sq1.out 2>$O/pmc_sq1.err || exit 1
timeout -k 10 400 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d $O/pmc_sq2 -- python3 $R/bench.py --cpu-sample 0 --no-extras --steps 3 --warmup 1 > $O/pmc_sq2.out 2>$O/pmc_sq2.err || exit 1
cd $R
timeout -k 10 400 python3 profiles/cpp_host_mixed.py 1200 > $O/cpp_host_mixed.json 2>$O/cpp_host_mixed.err || exit 1
bash profiles/r04_mixed_timeline.sh r04 1200 || exit 1
cp $R/gpurun_out/mix_r04.txt $O/mixed_timeline.txt
for c in FETCH_SIZE WRITE_SIZE tcc sq1 sq2; do python3 profiles/pmc_summary.py $O/pmc_$c 20 > $O/$c.json; rm -rf $O/pmc_$c; done
cat $O/cpp_host_mixed.json | cut -c1-600

# the mixed-species collection through `mbgc-hip c` under rocprofv3: bash profiles/r04_mixed_timeline.sh <tag> <genomes> [tool args...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; n=$2; shift; shift
O=$R/gpurun_out/mix_$tag
rm -rf $O; mkdir -p $O
D=$(python3 -c "
import sys; sys.path.insert(0, '$R')
import bench
print(bench.write_mixed_species($n))") || exit 1
export MBGC_HIP_TIMES=1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $R/mbgc_amd/mbgc-hip c "$@" $D/list.txt $D/out > $O/tool.out 2>$O/tool.err || { tail -5 $O/tool.err; exit 1; }
grep "matching finished\|reader threads" $O/tool.err > $R/gpurun_out/mix_$tag.txt
python3 $R/profiles/timeline_tool.py $O/kt >> $R/gpurun_out/mix_$tag.txt
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $R/gpurun_out/mix_${tag}_kernel_stats.csv
rm -rf $O/kt $D

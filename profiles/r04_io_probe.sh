# where the time in front of the first round goes: bash profiles/r04_io_probe.sh [genomes=400]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
n=${1:-400}
D=$(python3 -c "
import sys; sys.path.insert(0, '$R')
import bench
print(bench.write_mixed_species($n))") || exit 1
for i in 1 2 3 4; do
  MBGC_HIP_TIMES=1 $R/mbgc_amd/mbgc-hip c $D/list.txt $D/out 2>&1 >/dev/null | grep "first read-ahead\|loadG0Ref\|matching finished\|prepared by" | cut -c1-260
done
rm -rf $D

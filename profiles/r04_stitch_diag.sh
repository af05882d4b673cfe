# how the stitch accepts the resolve blocks on divergent and on similar data (SWSEM_DEBUG_STATS): bash profiles/r04_stitch_diag.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=$(python3 -c "
import sys; sys.path.insert(0, '$R')
import bench
print(bench.write_mixed_species(300))") || exit 1
for args in "-m 3" "-t1" ""; do
  echo "mbgc-hip c $args (300 mixed-species genomes)"
  SWSEM_DEBUG_STATS=1 MBGC_HIP_TIMES=1 $R/mbgc_amd/mbgc-hip c $args $D/list.txt $D/out 2>&1 >/dev/null | grep "swsem stitch\|matching finished"
done
rm -rf $D

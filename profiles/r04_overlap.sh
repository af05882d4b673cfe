# warm-up length of the resolve blocks on divergent data (SWSEM_OVERLAP=positions; unset = adapted): bash profiles/r04_overlap.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=$(python3 -c "
import sys; sys.path.insert(0, '$R')
import bench
print(bench.write_mixed_species(400))") || exit 1
for ov in "" 640 768 1024 "" 1024; do
  for args in "-m 3" ""; do
    echo "SWSEM_OVERLAP=$ov mbgc-hip c $args"
    if [ -n "$ov" ]; then export SWSEM_OVERLAP=$ov; else unset SWSEM_OVERLAP; fi
    SWSEM_DEBUG_STATS=1 $R/mbgc_amd/mbgc-hip c $args $D/list.txt $D/out 2>&1 >/dev/null | grep "swsem stitch\|matching finished"
  done
done
rm -rf $D

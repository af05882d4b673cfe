# shortest resolve block on small units (mixed-species collection through the C++ host): bash profiles/r04_rbmin.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in ${RBMINS:-2048 1024 512 2048 1024 512}; do
  echo "SWSEM_RB_MIN=$m"
  SWSEM_RB_MIN=$m MBGC_MIX_RUNS="m1_rounds:,m3:-m 3" timeout -k 10 300 python3 $R/profiles/cpp_host_mixed.py 400 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:(v['matching_ms'],v['gbases_per_s'],v['final_unmatched_chars']) for k,v in d['runs'].items()})"
done

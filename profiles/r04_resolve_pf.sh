# resolve launches of few waves with every window looked up one window ahead (SWSEM_RESOLVE_PF_MAX waves; 0 = never): bash profiles/r04_resolve_pf.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in ${PFMAXS:-0 2048 0 2048 100000}; do
  echo "SWSEM_RESOLVE_PF_MAX=$m"
  SWSEM_RESOLVE_PF_MAX=$m MBGC_MIX_RUNS="m1_rounds:,m3:-m 3,m1_t1:-t1" timeout -k 10 300 python3 $R/profiles/cpp_host_mixed.py 600 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:(v['matching_ms'],v['gbases_per_s'],v['final_unmatched_chars']) for k,v in d['runs'].items()})"
done

# the emission's byte automata with 16 tasks per wave on small batches (SWSEM_EMIT_THIN_MAX chunks; 0 = never): bash profiles/r04_emit_thin.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in 0 512 0 512 100000; do
  echo "SWSEM_EMIT_THIN_MAX=$m"
  SWSEM_EMIT_THIN_MAX=$m MBGC_MIX_RUNS="m1_rounds:,m3:-m 3,m1_t1:-t1" timeout -k 10 300 python3 $R/profiles/cpp_host_mixed.py 600 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:(v['matching_ms'],v['gbases_per_s'],v['final_unmatched_chars']) for k,v in d['runs'].items()})"
done

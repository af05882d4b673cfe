# the emission's pairing kernels beside the round's insertion (the product) or behind it (SWSEM_META_AFTER_INSERT=1): bash profiles/r04_meta_after_insert.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in 0 1 0 1 0 1; do
  SWSEM_META_AFTER_INSERT=$m timeout -k 10 300 python3 $R/bench.py --cpu-sample 0 --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('SWSEM_META_AFTER_INSERT=$m', d['value'], d['ms_per_step'], d['kernel_ms_per_launch'])"
done

# resolve blocks cut by predicted cost (SWSEM_COST_CUT=1, the default) against blocks of equal length (0), bench.py's headline: bash profiles/r04_cost_cut.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in 0 1 0 1 0 1; do
  SWSEM_COST_CUT=$m MBGC_BENCH_BLOCK_TIMES=1 timeout -k 10 300 python3 $R/bench.py --cpu-sample 0 --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
t=d.get('resolve_block_ticks',{})
print('SWSEM_COST_CUT=$m', d['value'], d['ms_per_step'], d['kernel_ms_per_launch']['resolve'], d['kernel_ms_per_launch']['stitch'], 'blocks', t.get('blocks'), 'mean %.0f max %.0f max/mean %.3f p99/mean %.3f' % (t['mean'], t['max'], t['max']/t['mean'], t['p99']/t['mean']), 'replayed', d.get('replayed_resolve_blocks_per_step'))"
done

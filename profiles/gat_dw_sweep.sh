#!/bin/bash
# k_bd_dw's grid (workgroups = heads x row ranges): rocprofv3 average per setting -> gpurun_out/gat_dw_sweep.log
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/gat_dw_sweep.log
: > $OUT
cd /tmp && export TMPDIR=/tmp
for nb in 256 512 768 1024; do
  rm -rf /tmp/dwsweep
  CSL_BD_DW_BLOCKS=$nb rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dwsweep -- python3 $R/profiles/e2e_only.py --steps 64 --warmup 16 --model gat --fanout 10,10,10 --hidden 32 --streams 32 > /tmp/dwsweep.log 2>&1 || { tail -5 /tmp/dwsweep.log; exit 1; }
  f=$(find /tmp/dwsweep -name '*kernel_stats.csv' | head -1)
  echo "== CSL_BD_DW_BLOCKS=$nb: $(python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'k_bd_dw' in r['Name'] or 'k_reduce_multi' in r['Name']: print(r['Name'].split('(')[0][-24:], 'avg us %.1f x%s;' % (float(r['AverageNs'])/1e3, r['Calls']), end=' ')
")" >> $OUT
done
cat $OUT

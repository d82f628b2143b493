#!/bin/bash
# k_bd_dw's grid (workgroups = heads x row ranges): rocprofv3 average per setting -> gpurun_out/gat_dw_sweep.log
# usage: gat_dw_sweep.sh [ENVVAR kernel-substring values...]   default: CSL_BD_DW_BLOCKS k_bd_dw 256 512 768 1024
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/gat_dw_sweep.log
: > $OUT
cd /tmp && export TMPDIR=/tmp
VAR=${1:-CSL_BD_DW_BLOCKS}; PAT=${2:-k_bd_dw}; shift; shift
VALS=${@:-256 512 768 1024}
for nb in $VALS; do
  rm -rf /tmp/dwsweep
  env $VAR=$nb true; export $VAR=$nb; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dwsweep -- python3 $R/profiles/e2e_only.py --steps 64 --warmup 16 --model gat --fanout 10,10,10 --hidden 32 --streams 32 > /tmp/dwsweep.log 2>&1 || { tail -5 /tmp/dwsweep.log; exit 1; }
  f=$(find /tmp/dwsweep -name '*kernel_stats.csv' | head -1)
  echo "== $VAR=$nb: $(python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if any(p in r['Name'] for p in '$PAT'.split('+')): print(r['Name'].split('(')[0][-24:], 'avg us %.1f x%s;' % (float(r['AverageNs'])/1e3, r['Calls']), end=' ')
")" >> $OUT
done
cat $OUT

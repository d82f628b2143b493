#!/bin/bash
# What k_sample costs at lower occupancy (an LDS-staged counting sort folded into it would take ~32 KB per workgroup):
# the kernel's dynamic LDS is padded through CSL_SAMPLE_LDS_PAD.  -> gpurun_out/sample_occupancy.log
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/sample_occupancy.log
: > $OUT
for pad in 0 8192 16384 24576 32768; do
  echo "== CSL_SAMPLE_LDS_PAD=$pad" >> $OUT
  CSL_SAMPLE_LDS_PAD=$pad python3 $R/bench.py --steps 20 --warmup 5 --e2e-steps 0 --e2e-gat-steps 0 --no-cpu-baseline --no-compat --no-live-pmc 2>/dev/null | python3 -c "
import sys, json
for line in sys.stdin:
    line=line.strip()
    if line.startswith('{'):
        d=json.loads(line)
        k=d.get('kernels',{})
        print('value %.3e  ms_per_step %.3f  k_sample %s  k_scatter %s' % (d['value'], d['ms_per_step'], round(k['k_sample']['avg_us'],1), round(k['k_scatter']['avg_us'],1)))
" >> $OUT || exit 1
done
cat $OUT

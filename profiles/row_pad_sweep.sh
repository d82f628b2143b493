#!/bin/bash
# GEMM row padding (cslicer.splitgnn.ROW_PAD, env CSLICER_ROW_PAD): the e2e steps per setting -> gpurun_out/row_pad_sweep.log
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/row_pad_sweep.log
: > $OUT
for pad in 4096 1024 512 256 128; do
  echo "== CSLICER_ROW_PAD=$pad" >> $OUT
  CSLICER_ROW_PAD=$pad python3 $R/profiles/e2e_only.py --steps 512 --warmup 64 --streams 64 2>/dev/null | grep e2e_only | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('  sage %.1f /s' % d['iters_per_sec'])" >> $OUT || exit 1
  CSLICER_ROW_PAD=$pad python3 $R/profiles/e2e_only.py --steps 256 --warmup 32 --model gat --fanout 10,10,10 --hidden 32 --streams 32 2>/dev/null | grep e2e_only | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('  gat  %.1f /s' % d['iters_per_sec'])" >> $OUT || exit 1
done
cat $OUT

#!/bin/bash
# A/B of the end-to-end step on one box: native step (one call per minibatch, direct hipBLASLt GEMMs) vs the same
# kernels issued from Python (autograd node, torch GEMMs with recorded TunableOp selections) vs the atomic-scatter backward
cd ${GRAFT_REPO_ROOT:-.}
python3 profiles/e2e_only.py --steps 64 --streams 32 --tuned > /dev/null 2>&1   # graph cache, first-touch
run() { python3 profiles/e2e_only.py --steps 512 --streams 32 --tuned 2>/dev/null | tail -1 | cut -c60-125; }
for rep in 1 2 3; do
  echo -n "native step:                 "; run
  echo -n "python step, by source:      "; CSLICER_PY_STEP=1 run
  echo -n "python step, atomic scatter: "; CSLICER_PY_STEP=1 CSLICER_NO_TRANSPOSE=1 run
done

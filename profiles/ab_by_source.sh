#!/bin/bash
# A/B of the end-to-end step on one box: direct hipBLASLt GEMMs vs torch's; input gradients gathered over the engine's
# slices by source vs the atomic scatter
cd ${GRAFT_REPO_ROOT:-.}
python3 profiles/e2e_only.py --steps 64 --streams 32 --tuned > /dev/null 2>&1   # graph cache, first-touch
run() { python3 profiles/e2e_only.py --steps 512 --streams 32 --tuned 2>/dev/null | tail -1 | cut -c60-125; }
for rep in 1 2 3; do
  echo -n "direct GEMMs, by source:  "; run
  echo -n "torch GEMMs,  by source:  "; CSLICER_TORCH_GEMMS=1 run
  echo -n "direct GEMMs, atomic:     "; CSLICER_NO_TRANSPOSE=1 run
  echo -n "torch GEMMs,  atomic:     "; CSLICER_TORCH_GEMMS=1 CSLICER_NO_TRANSPOSE=1 run
done

#!/bin/bash
# A/B of the end-to-end step: input gradients gathered over the engine's slices by source vs the atomic scatter
cd ${GRAFT_REPO_ROOT:-.}
python3 profiles/e2e_only.py --steps 64 --streams 32 --tuned > /dev/null 2>&1   # graph cache, first-touch
for rep in 1 2 3; do
  echo -n "by source: "; python3 profiles/e2e_only.py --steps 512 --streams 32 --tuned 2>/dev/null | tail -1 | cut -c60-130
  echo -n "atomic:    "; CSLICER_NO_TRANSPOSE=1 python3 profiles/e2e_only.py --steps 512 --streams 32 --tuned 2>/dev/null | tail -1 | cut -c60-130
done

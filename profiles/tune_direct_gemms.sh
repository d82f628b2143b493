#!/bin/bash
# Records the csl_gemm_f32 plans of the headline training step (run on the GPU box via gpurun): every solution of the
# library is timed per shape class (CSLICER_GEMM_TUNE=all), the chosen indices go to gpurun_out/gemm_plans_gfx950.txt
# (copy to occ-gnn_amd/cslicer/ to ship them).
cd ${GRAFT_REPO_ROOT:-.}
CSLICER_GEMM_PLANS=0 CSLICER_GEMM_TUNE=all CSLICER_GEMM_LOG=1 python3 - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys
sys.path.insert(0, "occ-gnn_amd")
sys.argv = ["e2e_only.py", "--steps", "256", "--streams", "32"]
exec(open("profiles/e2e_only.py").read())
from cslicer import aggr
aggr.gemm_save_plans("gpurun_out/gemm_plans_gfx950.txt")
print(open("gpurun_out/gemm_plans_gfx950.txt").read())
PY

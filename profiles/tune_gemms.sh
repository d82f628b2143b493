#!/bin/bash
# Records PyTorch TunableOp selections for the GEMM shapes of the end-to-end training step (run on the GPU box via
# gpurun); the result goes to gpurun_out/tunableop_gfx950.csv -> copy it to occ-gnn_amd/cslicer/tunableop_gfx950.csv.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/tune && cd $R/gpurun_out/tune
rm -f tunable*.csv
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=tunable.csv \
PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=300 python3 $R/profiles/e2e_only.py --steps 768 --warmup 64 2>&1 | tail -1
cp tunable0.csv $R/gpurun_out/tunableop_gfx950.csv
wc -l $R/gpurun_out/tunableop_gfx950.csv

#!/bin/bash
# Collects the rocprofv3 evidence for bench.py's default run (run on the GPU box via gpurun).
# usage: profiles/run_profiles.sh <tag>      -> gpurun_out/prof_<tag>/{trace,fetch,write}
set -e -o pipefail
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# --serial-rounds: one HIP stream, so each kernel's duration is its own (bench.py times kernels the same way)
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-compat --no-kernel-timing --serial-rounds --e2e-steps 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $R/bench.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $R/bench.py $ARGS > $OUT/write.log 2>&1
find $OUT -name '*.csv' | head -20

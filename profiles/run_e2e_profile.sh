#!/bin/bash
# rocprofv3 kernel trace of the end-to-end training step (run on the GPU box via gpurun).
# usage: profiles/run_e2e_profile.sh <tag> [extra args of e2e_only.py]   -> gpurun_out/prof_e2e_<tag>/
set -e -o pipefail
TAG=${1:-r2}
shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_e2e_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/profiles/e2e_only.py --steps 256 "$@" > $OUT/plain.log 2>&1
rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d $OUT/trace -- python3 $R/profiles/e2e_only.py --steps 256 "$@" > $OUT/trace.log 2>&1
python3 $R/profiles/e2e_summarize.py $OUT/trace 256 > $OUT/summary.md
cp $(find $OUT/trace -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
grep -h e2e_only $OUT/plain.log $OUT/trace.log > $OUT/rates.jsonl
# the raw trace is large: keep the summaries only
rm -rf $OUT/trace
head -30 $OUT/summary.md

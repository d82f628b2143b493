#!/bin/bash
# HBM traffic (rocprofv3 PMC: FETCH_SIZE and WRITE_SIZE, one pass each, nothing but --kernel-trace beside them) of the
# end-to-end training step's kernels, plus a plain kernel trace of the same command.  Run on the GPU box via gpurun:
#   profiles/run_e2e_pmc.sh <tag> [extra args of e2e_only.py]   -> gpurun_out/pmc_e2e_<tag>/summary.md
set -e -o pipefail
TAG=${1:-r3}
shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_e2e_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
STEPS=128
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/profiles/e2e_only.py --steps $STEPS --streams 64 "$@" > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $R/profiles/e2e_only.py --steps $STEPS --streams 64 "$@" > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $R/profiles/e2e_only.py --steps $STEPS --streams 64 "$@" > $OUT/write.log 2>&1
python3 $R/profiles/e2e_pmc_summarize.py $OUT "\`profiles/e2e_only.py --steps $STEPS --streams 64 $*\` (products-like graph, batch 1024, features 100, 47 classes)" > $OUT/summary.md
cp $(find $OUT/trace -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
grep -h e2e_only $OUT/trace.log $OUT/fetch.log $OUT/write.log > $OUT/rates.jsonl || true
rm -rf $OUT/trace $OUT/fetch $OUT/write
cat $OUT/summary.md

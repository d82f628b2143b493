#!/bin/bash
# k_sample over-fetch experiment (run on the GPU box via gpurun): FETCH_SIZE of the slicer kernels when all
# 128 streams slice DIFFERENT minibatches (the bench workload) vs the SAME minibatch (every row a stream touches is
# touched by the other 127 at the same time: the most any grouping of the streams' frontier nodes by row could save).
# usage: profiles/pmc_same_batch.sh <tag>   -> gpurun_out/pmc_same_<tag>/{normal,same}
set -e -o pipefail
TAG=${1:-r2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_same_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-compat --no-kernel-timing --serial-rounds --e2e-steps 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/normal -- python3 $R/bench.py $ARGS > $OUT/normal.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/same -- python3 $R/bench.py $ARGS --same-batch > $OUT/same.log 2>&1
python3 - $OUT <<'PY'
import collections, csv, glob, os, sys
out = sys.argv[1]
def agg(d):
    a = collections.OrderedDict()
    f = glob.glob(os.path.join(out, d, "*", "*counter_collection.csv"))[0]
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        x = a.setdefault(k, [0, 0.0]); x[0] += 1; x[1] += float(r["Counter_Value"])
    return a
n, s = agg("normal"), agg("same")
with open(os.path.join(out, "summary.md"), "w") as f:
    f.write("| kernel | launches | FETCH_SIZE MiB/launch (raw), 128 different minibatches | same minibatch on all 128 streams | ratio |\n|---|---|---|---|---|\n")
    for k, (c, v) in n.items():
        if k in s and k.startswith("k_") and c >= 10:
            a_, b_ = v / c / 1024, s[k][1] / s[k][0] / 1024
            f.write("| %s | %d | %.1f | %.1f | %.2f |\n" % (k, c, a_, b_, b_ / max(a_, 1e-9)))
print(open(os.path.join(out, "summary.md")).read())
PY
rm -rf $OUT/normal $OUT/same

#!/bin/bash
# SQ counters per launch for the kernels whose names contain one of the given substrings, one rocprofv3 --pmc pass per
# counter pair (nothing but --kernel-trace beside them).
# usage: sq_pmc.sh <out-name> <name1+name2+...> -- <python script and args>     -> gpurun_out/<out-name>.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
NAME=$1; PAT=$2; shift; shift; shift
OUT=$R/gpurun_out/$NAME.txt
: > $OUT
SCRIPT=$1; shift
[ -f "$R/$SCRIPT" ] && SCRIPT=$R/$SCRIPT
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT"; do
  rm -rf /tmp/sq_pmc
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/sq_pmc -- python3 $SCRIPT "$@" > /tmp/sq_pmc.log 2>&1 || { echo "== $grp: FAILED" >> $OUT; tail -2 /tmp/sq_pmc.log >> $OUT; continue; }
  f=$(find /tmp/sq_pmc -name '*counter_collection.csv' | head -1)
  python3 - "$f" "$PAT" <<'PY' >> $OUT
import csv, sys, collections
pats = sys.argv[2].split("+")
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40], r["Counter_Name"])
    agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
for (k, c), (n, v) in sorted(agg.items()):
    if any(p in k for p in pats):
        print("%-34s %-24s launches %4d  per launch %.4g" % (k, c, n, v / n))
PY
done
cat $OUT

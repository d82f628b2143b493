#!/bin/bash
# Extra PMC passes for diagnosing one kernel (write path / L2 / wave stalls). Run via gpurun.
# usage: profiles/pmc_extra.sh <tag>   -> gpurun_out/pmcx_<tag>/{wr,l2,sq}
set -e -o pipefail
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcx_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 6 --warmup 2 --no-cpu-baseline --no-compat --no-kernel-timing --serial-rounds --e2e-steps 0"
[[ "${PMCX_LEGS:-wr l2 sq}" == *wr* ]] && rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum --kernel-trace --output-format csv -d $OUT/wr -- python3 $R/bench.py $ARGS > $OUT/wr.log 2>&1
[[ "${PMCX_LEGS:-wr l2 sq}" == *l2* ]] && rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_WRITE_REQ_sum --kernel-trace --output-format csv -d $OUT/l2 -- python3 $R/bench.py $ARGS > $OUT/l2.log 2>&1
[[ "${PMCX_LEGS:-wr l2 sq}" == *sq* ]] && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/sq -- python3 $R/bench.py $ARGS > $OUT/sq.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
LEGS = [l for l in ('wr', 'l2', 'sq') if os.path.isdir(os.path.join(out, l))]
for leg in (LEGS):
    f = glob.glob(os.path.join(out, leg, "*", "*counter_collection.csv"))
    if not f:
        print(leg, "no data"); continue
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    seen = collections.Counter(); did = {}
    agg = collections.OrderedDict()
    for r in rows:
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if k not in ("k_emit", "k_sample", "k_bucket", "k_scatter", "k_count", "k_selfin", "k_degree"): continue
        key = (k, r["Dispatch_Id"])
        if key not in did:
            did[key] = seen[k] % 3; seen[k] += 1
        a = agg.setdefault((k, did[key], r["Counter_Name"]), [0, 0.0])
        a[0] += 1; a[1] += float(r["Counter_Value"])
    print("==", leg)
    for (k, l, c), (n, v) in agg.items():
        print("%-10s L%d %-36s %14.0f per launch" % (k, l, c, v / n))
PY

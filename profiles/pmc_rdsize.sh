#!/bin/bash
# Size mix of the L2 -> fabric read requests per kernel (32 / 64 / 128 B). Run via gpurun.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_rd
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-compat --no-kernel-timing --serial-rounds --e2e-steps 0 > $OUT.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if not k.startswith("k_"):
        continue
    agg.setdefault(k, collections.Counter())[r["Counter_Name"]] += float(r["Counter_Value"])
for k, a in agg.items():
    t = a.get("TCC_EA0_RDREQ_sum", 0)
    if t < 1e5:
        continue
    print("%-16s RDREQ %8.1f M/round  128B %5.1f%%  64B %5.1f%%  32B %5.1f%%" % (
        k, t / 1e6 / 8, 100 * a["TCC_EA0_RDREQ_128B_sum"] / t, 100 * a["TCC_EA0_RDREQ_64B_sum"] / t,
        100 * a["TCC_EA0_RDREQ_32B_sum"] / t))
PY

#!/bin/bash
# SQ counters of the fused gather -> fp32-MFMA forward (profiles/mfma_probe.py), one rocprofv3 --pmc pass per group (nothing
# but --kernel-trace beside them): how busy the matrix pipe is, and in how many of those cycles vector instructions execute
# beside it.  Variants: the kernel as shipped, without the multiply (CSLICER_MFMA_DBG=2) -> gpurun_out/mfma_pmc/
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/mfma_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  tag=$(echo $grp | tr ' ' '+')
  rm -rf /tmp/mfma_pmc
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/mfma_pmc -- python3 $R/profiles/mfma_probe.py --iters 20 > /tmp/mfma_pmc.log 2>&1 || { echo "== $tag: FAILED"; tail -3 /tmp/mfma_pmc.log; continue; }
  f=$(find /tmp/mfma_pmc -name '*counter_collection.csv' | head -1)
  python3 - "$f" "$tag" <<'PY' >> $OUT/counters.txt
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40], r["Counter_Name"])
    agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
for (k, c), (n, v) in sorted(agg.items()):
    if "sage_fwd_mfma" in k or "k_sage_cat" in k or "Cijk" in k:
        print("%-42s %-28s launches %4d  per launch %.4g" % (k, c, n, v / n))
PY
done
cat $OUT/counters.txt

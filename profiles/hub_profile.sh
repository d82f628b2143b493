cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/hubprof -- python3 $GRAFT_REPO_ROOT/profiles/hub_probe.py hub > $GRAFT_REPO_ROOT/gpurun_out/hubprof.log 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/hubprof -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-60s calls %6s total %9.1f ms avg %9.1f us max %9.1f us" % (r["Name"].replace("(anonymous namespace)::","")[:60], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, float(r["MaxNs"])/1e3))
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/hubprof

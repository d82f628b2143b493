#!/usr/bin/env python3
"""Per-launch timeline of the last engine round in a rocprofv3 kernel trace.
usage: python3 profiles/timeline.py <dir containing *_kernel_trace.csv> [index of the round, default -2]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))
        if "k_" in r["Kernel_Name"] and "pack" not in r["Kernel_Name"] and "mt19937" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a round starts at k_degree (layer 0; the later layers' degree pass is k_selfin_degree); with rounds in flight on
# several HIP streams, keep the launches of the stream that runs the last complete round
starts = [i for i, r in enumerate(rows) if "::k_degree(" in r["Kernel_Name"]]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
qid = rows[starts[which]].get("Queue_Id")
last = [r for r in rows[starts[which]:] if r.get("Queue_Id") == qid]
nxt = [i for i, r in enumerate(last) if i and "::k_degree(" in r["Kernel_Name"]]
if nxt:
    last = last[:nxt[0]]
t0 = int(last[0]["Start_Timestamp"])
tot = 0
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print("%-16s start=%8.1fus dur=%7.1fus blocks=%d" % (name, (s - t0) / 1e3, (e - s) / 1e3,
          int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) // int(r["Workgroup_Size_X"])))
    tot += e - s
print("sum of kernels %.1fus, wall %.1fus" % (tot / 1e3, (int(last[-1]["End_Timestamp"]) - t0) / 1e3))

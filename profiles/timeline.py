#!/usr/bin/env python3
"""Per-launch timeline of the last engine round in a rocprofv3 kernel trace.
usage: python3 profiles/timeline.py <dir containing *_kernel_trace.csv> [n_launches_per_round]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))
        if "k_" in r["Kernel_Name"] and "pack" not in r["Kernel_Name"] and "mt19937" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a round starts at k_seeds
starts = [i for i, r in enumerate(rows) if "k_seeds" in r["Kernel_Name"]]
last = rows[starts[-1]:]
t0 = int(last[0]["Start_Timestamp"])
tot = 0
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print("%-16s start=%8.1fus dur=%7.1fus blocks=%d" % (name, (s - t0) / 1e3, (e - s) / 1e3,
          int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) // int(r["Workgroup_Size_X"])))
    tot += e - s
print("sum of kernels %.1fus, wall %.1fus" % (tot / 1e3, (int(last[-1]["End_Timestamp"]) - t0) / 1e3))

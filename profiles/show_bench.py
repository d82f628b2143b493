#!/usr/bin/env python3
"""Pretty-print the JSON line bench.py emits (reads stdin or a file)."""
import json
import sys

txt = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
line = [l for l in txt.splitlines() if l.startswith("{")][-1]
d = json.loads(line)
print("%.2f Gedges/s  %.0f it/s  %.3f ms/step  (S=%s)" % (d["value"] / 1e9, d["iters_per_sec"], d["ms_per_step"], d["config"]["streams"]))
for k, v in d.get("kernels", {}).items():
    if v.get("launches"):
        print("  %-16s avg %8.1f us  x%d  %s" % (k, 1e3 * v["ms_total"] / v["launches"], v["launches"],
              ("%.0f GB/s alg" % v["achieved_GBs"]) if v.get("achieved_GBs") else ""))
if "roofline" in d:
    print("  roofline:", d["roofline"]["kernel"], "%.1f%% of 8 TB/s" % (100 * d["roofline"]["frac"]))
if "path_roofline" in d:
    print("  path: %.1f%% of 8 TB/s" % (100 * d["path_roofline"]["frac"]))
if "cpu_baseline" in d:
    c = d["cpu_baseline"]
    print("  cpu: %.3f Gedges/s on %d cores (%s)" % (c["value"] / 1e9, c["cores"], c["kind"]))
if "compat_path" in d:
    c = d["compat_path"]
    print("  compat (host lists):", ("%.0f samples/s" % c["samples_per_sec"]) if "samples_per_sec" in c else c)
if d.get("e2e"):
    e = d["e2e"]
    if "iters_per_sec" in e:
        print("  e2e: %.0f it/s (%.3f ms)" % (e["iters_per_sec"], e["ms_per_iter"]))
        r = e.get("roofline")
        if r:
            print("    GEMM %.1f TF/s (%.0f%% of %.0f), aggregation %.0f GB/s (%.0f%% of 8 TB/s) over the whole step" % (
                r["gemm"]["achieved"], 100 * r["gemm"]["frac"], r["gemm"]["peak"], r["aggregation"]["achieved"],
                100 * r["aggregation"]["frac"]))
    else:
        print("  e2e:", e)

#!/usr/bin/env python3
"""Condenses a rocprofv3 --kernel-trace --stats run of profiles/e2e_only.py into a markdown table:
per-kernel share of the GPU time, grouped (library GEMMs / aggregation kernels of libcslicer_hip /
slicer kernels / torch elementwise+optimizer), and the GPU-busy fraction of the timed region
(union of kernel intervals / window; the window is the ROCTX range `timed_region` of e2e_only.py,
from `--marker-trace`).

usage: python3 profiles/e2e_summarize.py <rocprof dir> <timed steps> > summary.md"""
import csv
import glob
import os
import sys


def group_of(name):
    n = name
    if "Cijk_" in n or "gemm" in n.lower() or "hipblaslt" in n.lower() or "rocblas" in n.lower():
        return "library GEMM"
    for k in ("k_spmm", "k_gather_rows", "k_scatter_add_rows", "k_div_rows", "k_gat_", "k_sage_", "k_csr_"):
        if k in n:
            return "aggregation (libcslicer_hip)"
    for k in ("k_sample", "k_emit", "k_bucket", "k_scatter", "k_count", "k_scan", "k_degree", "k_seeds", "k_selfin",
              "k_graph", "k_mt19937", "k_dupseeds", "k_pack"):
        if k in n:
            return "slicer (libcslicer_hip)"
    return "torch elementwise / optimizer / copies"


def main():
    d, steps = sys.argv[1], int(sys.argv[2])
    trace = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(trace)))
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
    mk = glob.glob(os.path.join(d, "**", "*marker_api_trace.csv"), recursive=True)
    w0 = w1 = None
    if mk:
        for r in csv.DictReader(open(mk[0])):
            if "timed_region" in (r.get("Function", "") + r.get("Message", "")):
                w0, w1 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if w0 is None:
        raise SystemExit("no `timed_region` marker in the trace (run rocprofv3 with --marker-trace)")
    iv = [(s_, min(e_, w1), n_) for s_, e_, n_ in iv if s_ < w1]
    t1 = w1
    busy, cur_s, cur_e = 0, None, None
    per, cnt = {}, {}
    for s, e, n in iv:
        if e <= w0:
            continue
        s = max(s, w0)
        per[n] = per.get(n, 0) + (e - s)
        cnt[n] = cnt.get(n, 0) + 1
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        busy += cur_e - cur_s
    window = t1 - w0
    tot = sum(per.values())
    steps_w = steps
    print("# e2e training step: rocprofv3 kernel trace of the timed region (%d steps)\n" % steps)
    print("window %.1f ms, GPU busy (union of kernel intervals) %.1f ms = **%.1f %%**; sum of kernel durations "
          "%.1f ms (overlap of the slicer's side streams with the step: %.2fx); ~%.3f ms of kernel time per step\n"
          % (window / 1e6, busy / 1e6, 100.0 * busy / window, tot / 1e6, tot / max(busy, 1), tot / 1e6 / steps_w))
    groups = {}
    for n, v in per.items():
        groups[group_of(n)] = groups.get(group_of(n), 0) + v
    print("| group | ms | share of kernel time |\n|---|---|---|")
    for g, v in sorted(groups.items(), key=lambda kv: -kv[1]):
        print("| %s | %.2f | %.1f %% |" % (g, v / 1e6, 100.0 * v / tot))
    print("\n| kernel | group | launches | total ms | avg us | share |\n|---|---|---|---|---|---|")
    for n, v in sorted(per.items(), key=lambda kv: -kv[1])[:40]:
        short = n if len(n) < 90 else n[:87] + "..."
        print("| `%s` | %s | %d | %.2f | %.1f | %.1f %% |" % (short, group_of(n), cnt[n], v / 1e6, v / 1e3 / cnt[n],
                                                             100.0 * v / tot))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Turn a gpurun_out/prof_<tag>/ directory (profiles/run_profiles.sh) into the
tracked evidence under profiles/: the rocprofv3 --stats kernel table and the
per-kernel FETCH_SIZE / WRITE_SIZE sums of the two PMC passes.

usage: python3 profiles/summarize.py gpurun_out/prof_r1a profiles/r1a
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def pmc(path):
    agg = collections.OrderedDict()
    files = glob.glob(os.path.join(path, "*", "*counter_collection.csv"))
    if not files:
        return agg, None
    cname = None
    for r in csv.DictReader(open(files[0])):
        k = short(r["Kernel_Name"])
        cname = r["Counter_Name"]
        a = agg.setdefault(k, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        a[2] = max(a[2], float(r["Counter_Value"]))
    return agg, cname


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(dst, exist_ok=True)
    stats = glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv"))[0]
    shutil.copy(stats, os.path.join(dst, "kernel_stats.csv"))
    for leg in ("trace", "fetch", "write"):
        log = os.path.join(src, leg + ".log")
        if os.path.exists(log):
            lines = [l for l in open(log) if l.startswith("{")]
            if lines:
                open(os.path.join(dst, "bench_%s.json" % leg), "w").write(lines[-1])
    rows = list(csv.DictReader(open(stats)))
    fetch, _ = pmc(os.path.join(src, "fetch"))
    write, _ = pmc(os.path.join(src, "write"))
    with open(os.path.join(dst, "summary.md"), "w") as f:
        f.write("# rocprofv3 summary: %s\n\n" % os.path.basename(src))
        f.write("Command: `profiles/run_profiles.sh` (bench.py --steps 20 --warmup 5 --no-cpu-baseline "
                "--no-kernel-timing --serial-rounds --e2e-steps 0; three separate runs: --kernel-trace --stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE).\n\n")
        f.write("FETCH_SIZE / WRITE_SIZE are rocprofv3's raw values in KiB, summed over the launches of the run "
                "(25 rounds x 3 layers = 75 launches per slicer kernel). FETCH_SIZE = TCC_EA0_RDREQ x 64 B; "
                "profiles/pmc_rdsize.sh shows > 99.8 % of every kernel's read requests are 128-B requests (random "
                "gathers pull whole lines too), so the bytes fetched are 2 x the raw value below "
                "(MI355X_MICROARCH.md's gfx950 correction). WRITE_SIZE is exact (64-B write requests).\n\n")
        f.write("| kernel | calls | total ms | avg us | max us | % | FETCH_SIZE MiB/launch (raw) | WRITE_SIZE MiB/launch |\n")
        f.write("|---|---|---|---|---|---|---|---|\n")
        for r in rows:
            k = short(r["Name"])
            fa = fetch.get(k)
            wa = write.get(k)
            f.write("| %s | %s | %.3f | %.1f | %.1f | %s | %s | %s |\n" % (
                k, r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                float(r["MaxNs"]) / 1e3, r["Percentage"],
                "%.1f" % (fa[1] / fa[0] / 1024) if fa else "-",
                "%.1f" % (wa[1] / wa[0] / 1024) if wa else "-"))
    json.dump({"fetch_KiB_sum": {k: v[1] for k, v in fetch.items()},
               "write_KiB_sum": {k: v[1] for k, v in write.items()},
               "calls": {k: v[0] for k, v in fetch.items()}},
              open(os.path.join(dst, "pmc.json"), "w"), indent=1)
    print(open(os.path.join(dst, "summary.md")).read())


if __name__ == "__main__":
    main()

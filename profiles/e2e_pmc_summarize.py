#!/usr/bin/env python3
"""Per-kernel time and HBM traffic of the end-to-end training step from profiles/run_e2e_pmc.sh's three rocprofv3 runs
(--kernel-trace --stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE).  FETCH_SIZE counts 64 B per read request while this
library's reads are 128-byte requests (profiles/pmc_rdsize.sh; MI355X_MICROARCH.md's gfx950 correction): bytes fetched =
2 x FETCH_SIZE.  WRITE_SIZE is exact.  Prints a markdown table: per launch, MiB and the GB/s they amount to."""
import collections
import csv
import glob
import os
import sys


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def pmc(path):
    agg = collections.OrderedDict()
    files = glob.glob(os.path.join(path, "*", "*counter_collection.csv"))
    for r in csv.DictReader(open(files[0])) if files else ():
        a = agg.setdefault(short(r["Kernel_Name"]), [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg


def main():
    src = sys.argv[1]
    stats = glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv"))[0]
    rows = list(csv.DictReader(open(stats)))
    fetch, write = pmc(os.path.join(src, "fetch")), pmc(os.path.join(src, "write"))
    print("# e2e training step: kernel time and HBM traffic per launch (rocprofv3 PMC)\n")
    what = sys.argv[2] if len(sys.argv) > 2 else ("`profiles/e2e_only.py --steps 128 --streams 64` (products-like graph, 15/10/5, "
                                                  "batch 1024, features 100,\nhidden 256, 47 classes)")
    print("Three runs of %s:" % what)
    print("`--kernel-trace --stats`, `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`.  fetched = 2 x FETCH_SIZE")
    print("(128-byte read requests counted as 64 B on gfx950), written = WRITE_SIZE; GB/s = (fetched + written) / average duration.\n")
    print("| kernel | launches | avg us | share | fetched MiB | written MiB | GB/s |")
    print("|---|---|---|---|---|---|---|")
    for r in rows:
        k = short(r["Name"])
        us = float(r["AverageNs"]) / 1e3
        fa, wa = fetch.get(k), write.get(k)
        if float(r["Percentage"]) < 0.3:
            continue
        f_mib = 2.0 * fa[1] / fa[0] / 1024 if fa else None      # KiB -> MiB, x2
        w_mib = wa[1] / wa[0] / 1024 if wa else None
        gbs = ((f_mib or 0) + (w_mib or 0)) * 1.048576 / us * 1e3 if (fa or wa) else None
        print("| `%s` | %s | %.1f | %s %% | %s | %s | %s |" % (
            k[:70], r["Calls"], us, r["Percentage"], "%.1f" % f_mib if fa else "-", "%.1f" % w_mib if wa else "-",
            "%.0f" % gbs if gbs else "-"))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Per-layer view of a profiles/run_profiles.sh directory: average duration of every slicer kernel by
layer (launch order within a kernel: layer = launch index mod n_layers), and FETCH_SIZE / WRITE_SIZE per
launch by layer from the two PMC passes.

usage: python3 profiles/per_layer.py gpurun_out/prof_<tag> [n_layers]   -> markdown on stdout
"""
import collections
import csv
import glob
import os
import sys


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def by_layer(path, value, L, start="Start_Timestamp"):
    files = glob.glob(os.path.join(path, "*", "*" + ("kernel_trace.csv" if value is None else "counter_collection.csv")))
    if not files:
        return {}
    rows = list(csv.DictReader(open(files[0])))
    key = start if start in rows[0] else "Dispatch_Id"
    rows.sort(key=lambda r: int(r[key]))
    seen = collections.Counter()
    out = collections.OrderedDict()
    for r in rows:
        k = short(r["Kernel_Name"])
        if not k.startswith("k_") or k in ("k_seeds", "k_mt19937_fill") or k.startswith("k_pack"):
            continue
        l = seen[k] % L
        seen[k] += 1
        x = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 if value is None else float(r[value])
        a = out.setdefault(k, [[0, 0.0] for _ in range(L)])
        a[l][0] += 1
        a[l][1] += x
    return out


def main():
    src = sys.argv[1]
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    dur = by_layer(os.path.join(src, "trace"), None, L)
    fetch = by_layer(os.path.join(src, "fetch"), "Counter_Value", L)
    write = by_layer(os.path.join(src, "write"), "Counter_Value", L)
    print("| kernel | " + " | ".join("layer %d us" % l for l in range(L)) + " | "
          + " | ".join("L%d fetch MiB (raw) / write MiB" % l for l in range(L)) + " |")
    print("|---" * (1 + 2 * L) + "|")
    tot = [0.0] * L
    for k, a in dur.items():
        d = [x[1] / max(x[0], 1) for x in a]
        for l in range(L):
            tot[l] += d[l]
        cells = []
        for l in range(L):
            f = fetch.get(k)
            w = write.get(k)
            cells.append("%s / %s" % ("%.1f" % (f[l][1] / max(f[l][0], 1) / 1024) if f else "-",
                                      "%.1f" % (w[l][1] / max(w[l][0], 1) / 1024) if w else "-"))
        print("| %s | " % k + " | ".join("%.1f" % x for x in d) + " | " + " | ".join(cells) + " |")
    print("| **sum** | " + " | ".join("%.1f" % x for x in tot) + " |" + " |" * L)


if __name__ == "__main__":
    main()

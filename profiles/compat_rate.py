#!/usr/bin/env python3
"""PCIe-inclusive rate of the compatibility path: the native `cslicer` module used
exactly like the reference's test_py.py (every sample's lists copied to host
vectors), products-like graph, fanout 10/10/10, 4 parts (the reference's constants).
usage: python3 profiles/compat_rate.py [workers] [batch] [samples]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "occ-gnn_amd"))
from conftest import load_native_module  # noqa: E402
from cslicer import l0  # noqa: E402


def main():
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    want = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    root = os.environ.get("CSLICER_BENCH_CACHE", "/tmp/cslicer_bench_cache")
    d = os.path.join(root, "l0", "products-like")
    if not os.path.exists(os.path.join(d, "meta.txt")):
        t0 = time.time()
        indptr, indices = l0.synth_graph(2_449_029, 50.5, seed=0)
        l0.write_l0(d, indptr, indices)
        print("wrote L0 dataset in %.1fs" % (time.time() - t0))
    m = load_native_module()
    t0 = time.time()
    csl = m.cslicer("products-like", 16, workers, 1, batch, data_root=os.path.join(root, "l0"))
    print("constructor (load + upload): %.1fs" % (time.time() - t0))
    n = min(want, csl.getNoSamples())
    s = csl.getSample()  # first sample includes the rng window fill
    t0 = time.time()
    edges = 0
    for _ in range(n - 1):
        s = csl.getSample()
    dt = time.time() - t0
    # touching attributes converts vectors to Python lists (as in the reference)
    t1 = time.time()
    tot = sum(len(s.layers[l][g].in_nodes) for l in range(3) for g in range(4))
    t_attr = time.time() - t1
    print("compat path: %d samples in %.3fs = %.1f samples/s (workers=%d, batch=%d); "
          "one sample's in_nodes -> Python lists: %d ids in %.1f ms" % (n - 1, dt, (n - 1) / dt, workers, batch, tot, 1e3 * t_attr))
    del csl


if __name__ == "__main__":
    main()

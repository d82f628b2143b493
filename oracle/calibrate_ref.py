#!/usr/bin/env python3
"""Speed calibration of the CPU oracle (oracle/cslicer_oracle.c, the "port")
against the UNMODIFIED reference (oracle/_ref/ref_harness), same graph, same
batches, the reference's constants (fanout 10/10/10, 4 parts).  TEST
INFRASTRUCTURE; needs /root/reference (build container only).

usage: python3 oracle/calibrate_ref.py [nodes] [mean_deg] [batch] [n_batches] [threads]
"""
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != HERE]
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "occ-gnn_amd"))
from cslicer import l0  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
    deg = float(sys.argv[2]) if len(sys.argv) > 2 else 36.0
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    nb = int(sys.argv[4]) if len(sys.argv) > 4 else 64
    T = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    indptr, indices = l0.synth_graph(n, deg, seed=0)
    perm = np.random.default_rng(1).permutation(n)
    batches = [perm[i * B:(i + 1) * B] for i in range(nb)]
    with tempfile.TemporaryDirectory() as td:
        d = os.path.join(td, "g")
        l0.write_l0(d, indptr, indices)
        words = [len(batches)]
        for b in batches:
            words.append(len(b))
            words.extend(int(x) for x in b)
        bp = os.path.join(td, "b.bin")
        np.array(words, dtype=np.int64).tofile(bp)
        r = subprocess.run([os.path.join(HERE, "_ref", "ref_harness"), "bench", d, bp, str(T)],
                           check=True, capture_output=True, text=True)
        ref_s = float(re.search(r"seconds=([0-9.]+)", r.stderr).group(1))
    sec, edges = orc.bench(indptr, indices, batches, n_parts=4, fanouts=(10, 10, 10), threads=T, deep_copy=True)
    print("graph N=%d E=%d  batch %d x %d  threads %d" % (n, indices.shape[0], B, nb, T))
    print("reference (unmodified, -O3 -DNDEBUG): %.3f s  %.2f iters/s" % (ref_s, nb / ref_s))
    print("oracle port (-O3):                    %.3f s  %.2f iters/s  %.1f Medges/s" % (sec, nb / sec, edges / sec / 1e6))
    print("port/reference speed ratio: %.3f" % (ref_s / sec))


if __name__ == "__main__":
    main()

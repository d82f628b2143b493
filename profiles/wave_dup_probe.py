#!/usr/bin/env python3
"""What a WAVE-level dedup (ballot / DS_SWIZZLE compare across the 64 lanes, the north_star's phrase) in front of the LDS
hash table of k_bucket could remove: the fraction of dedup-queue entries whose node id another lane of the same wave
already holds (csl_debug_wave_duplicates, environment CSL_WAVE_DUP_PROBE=1).
usage (GPU box):  CSL_WAVE_DUP_PROBE=1 python3 profiles/wave_dup_probe.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "occ-gnn_amd"))
os.environ.setdefault("CSL_WAVE_DUP_PROBE", "1")
from cslicer import _abi, l0  # noqa: E402

N, S, B, fan, P = 2_449_029, 32, 1024, (15, 10, 5), 4
cache = os.path.join(os.environ.get("CSLICER_BENCH_CACHE", "/tmp/cslicer_bench_cache"), "g_n2449029_d50.5_s0")
if os.path.exists(os.path.join(cache, "ok")):
    indptr, indices = np.load(os.path.join(cache, "indptr.npy")), np.load(os.path.join(cache, "indices.npy"))
else:
    indptr, indices = l0.synth_graph(N, 50.5, seed=0)
eng = _abi.Engine(indptr, indices, n_parts=P, fanouts=fan, max_batch=B, n_streams=S)
eng.set_nodes(np.random.default_rng(1).permutation(N))
rounds = 4
for k in range(rounds):
    eng.submit_round(k * S, B, S)
    eng.sync()
out = (C.c_uint64 * 3)()
L = _abi.load()
L.csl_debug_wave_duplicates.argtypes = [C.POINTER(C.c_uint64)]
rc = L.csl_debug_wave_duplicates(out)
assert rc == 0, rc
e, row, allr = int(out[0]), int(out[1]), int(out[2])
print("queue entries probed (first RC x BT of every bucket): %d over %d rounds of %d minibatches" % (e, rounds, S))
print("not the first of their id among the 64 entries of a wave's register row : %d = %.3f %%" % (row, 100.0 * row / max(e, 1)))
print("not the first of their id among all %d entries a wave holds of a bucket   : %d = %.3f %%" % (4 * 64, allr, 100.0 * allr / max(e, 1)))
eng.close()

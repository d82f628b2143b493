#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the UNMODIFIED reference cslicer.

TEST INFRASTRUCTURE.  Runs only in the build container (needs /root/reference
and oracle/_ref/ref_harness, see oracle/Makefile).  For each case it writes an
L0 dataset to a temp dir, lets the reference's own Dataset/Slicer/PySample code
process the batches (fanout 10/10/10, 4 parts, workload v%4, mt19937(5489) --
the reference's hard-coded constants) and stores inputs + every exported list as
a small .npz.  The fixtures are data only (inputs and expected outputs).

Usage:  python3 oracle/make_golden.py            # regenerate all fixtures
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "occ-gnn_amd"))
from cslicer import l0  # noqa: E402

HARNESS = os.path.join(HERE, "_ref", "ref_harness")
GOLD = os.path.join(ROOT, "tests", "golden")
MAGIC = 0x43534C4F52433031
LISTS = ["in_nodes", "indptr", "out_nodes", "owned_out_nodes", "indices", "self_ids_in", "self_ids_out"]


class Reader:
    def __init__(self, path):
        self.a = np.fromfile(path, dtype=np.int64)
        self.p = 0

    def word(self):
        v = int(self.a[self.p])
        self.p += 1
        return v

    def lst(self):
        n = self.word()
        v = self.a[self.p:self.p + n].copy()
        self.p += n
        return v


def parse_dump(path):
    r = Reader(path)
    assert r.word() == MAGIC
    nb = r.word()
    r.word()  # num_nodes
    out = {}
    for b in range(nb):
        out["b%d_seeds" % b] = r.lst()
        for l in range(3):
            k = "b%d_l%d_" % (b, l)
            out[k + "frontier"] = r.lst()
            out[k + "nbr_counts"] = r.lst()
            out[k + "nbr_flat"] = r.lst()
            out[k + "next_frontier"] = r.lst()
            out[k + "draws"] = np.array([r.word()], dtype=np.int64)
            for g in range(4):
                kg = k + "g%d_" % g
                out[kg + "gpu_id"] = np.array([r.word()], dtype=np.int64)
                for name in LISTS:
                    out[kg + name] = r.lst()
                for j in range(4):
                    out[kg + "from_ids%d" % j] = r.lst()
                for j in range(4):
                    out[kg + "to_ids%d" % j] = r.lst()
    assert r.p == r.a.shape[0], "trailing data in dump"
    return out


def run_reference(indptr, indices, batches):
    with tempfile.TemporaryDirectory() as td:
        d = os.path.join(td, "graph")
        l0.write_l0(d, indptr, indices)
        words = [len(batches)]
        for b in batches:
            words.append(len(b))
            words.extend(int(x) for x in b)
        bpath = os.path.join(td, "batches.bin")
        np.array(words, dtype=np.int64).tofile(bpath)
        opath = os.path.join(td, "out.bin")
        subprocess.run([HARNESS, "dump", d, bpath, opath], check=True,
                       stdout=subprocess.DEVNULL)
        return parse_dump(opath)


def graph_from_rows(rows):
    deg = np.array([len(r) for r in rows], dtype=np.int64)
    indptr = np.zeros(len(rows) + 1, dtype=np.int64)
    np.cumsum(deg, out=indptr[1:])
    indices = np.array([x for r in rows for x in r], dtype=np.int64)
    return indptr, indices


def case_toy40():
    # 40 nodes, mixed degrees below / at / above the fanout
    rng = np.random.default_rng(40)
    deg = rng.integers(0, 25, size=40)
    indptr, indices = l0.synth_graph(40, 0, seed=41, degrees=deg)
    batches = [[3, 17, 22, 39], [0, 8, 16, 31]]
    return indptr, indices, batches


def case_degree_edges():
    # isolated nodes, degree exactly 9 / 10 / 11, one hub
    n = 96
    deg = np.zeros(n, dtype=np.int64)
    deg[0:8] = 0
    deg[8:24] = 9
    deg[24:40] = 10
    deg[40:56] = 11
    deg[56:90] = np.arange(34) % 7 + 1
    deg[90:96] = 60
    indptr, indices = l0.synth_graph(n, 0, seed=7, degrees=deg)
    batches = [[0, 8, 24, 40, 90, 5, 57, 33], [1, 9, 25, 41, 91, 60, 61, 62], [95]]
    return indptr, indices, batches


def case_powerlaw2k():
    indptr, indices = l0.synth_graph(2000, 12.0, seed=2000)
    perm = np.random.default_rng(1).permutation(2000)
    batches = [perm[0:64].tolist(), perm[64:128].tolist(), perm[128:1152].tolist()]
    return indptr, indices, batches


def case_selfloops_multiedges():
    # not produced by the reference's converter (it strips self loops) but legal
    # input to Slicer: exercises the nd1 == nd2 branch of slice_layer
    # (slicer.cpp:33-35) for sampled neighbours, and repeated neighbours.
    rng = np.random.default_rng(99)
    rows = []
    n = 48
    for v in range(n):
        d = int(rng.integers(0, 16))
        r = sorted(int(x) for x in rng.integers(0, n, size=d))
        if v % 3 == 0 and d:
            r[0] = v  # self loop
            r = sorted(r)
        if v % 5 == 0 and d > 2:
            r[1] = r[2]  # multi-edge
        rows.append(r)
    indptr, indices = graph_from_rows(rows)
    batches = [[0, 3, 6, 9, 12, 15, 30, 45], [5, 10, 20, 40]]
    return indptr, indices, batches


def case_dense_small():
    # every node above the fanout: all rows take the random path; heavy dedup
    indptr, indices = l0.synth_graph(300, 0, seed=5, degrees=np.full(300, 40))
    perm = np.random.default_rng(2).permutation(300)
    batches = [perm[0:16].tolist(), perm[16:32].tolist()]
    return indptr, indices, batches


def case_duplicate_seeds():
    # repeated, non-adjacent seeds: BiPartite::reorder merges out_nodes
    # (bipartite.cpp:10) while indptr keeps one entry per push.  Pinned for the
    # oracle; the HIP engine rejects duplicate seeds (documented in DESIGN.md).
    indptr, indices = l0.synth_graph(200, 0, seed=11,
                                     degrees=np.random.default_rng(12).integers(0, 30, size=200))
    batches = [[5, 9, 5, 77, 9, 120]]
    return indptr, indices, batches


CASES = {
    "toy40": case_toy40,
    "degree_edges": case_degree_edges,
    "powerlaw2k": case_powerlaw2k,
    "selfloops_multiedges": case_selfloops_multiedges,
    "dense_small": case_dense_small,
    "duplicate_seeds": case_duplicate_seeds,
}


def list_hash(a):
    """64-bit digest of a list's int64 values, order-sensitive (blake2b of the little-endian bytes)."""
    import hashlib
    b = np.ascontiguousarray(np.asarray(a, dtype="<i8")).tobytes()
    return np.frombuffer(hashlib.blake2b(b, digest_size=8).digest(), dtype="<u8")[0]


def hashed_case():
    """A realistic-size case pinned by hashes only (the lists themselves would be megabytes):
    products-like degrees on 100k nodes, three consecutive minibatches of 1024 from one permutation.
    The graph is regenerated from its seed by the tests (cslicer.l0.synth_graph is deterministic)."""
    n, deg, seed = 100_000, 30.0, 77
    indptr, indices = l0.synth_graph(n, deg, seed=seed)
    perm = np.random.default_rng(5).permutation(n)
    batches = [perm[i * 1024:(i + 1) * 1024].tolist() for i in range(3)]
    out = run_reference(indptr, indices, batches)
    rec = {"graph": np.array([n, int(deg * 1000), seed], dtype=np.int64),
           "graph_csum": np.array([int(indptr.sum()), int(indices.sum())], dtype=np.int64),
           "perm_seed": np.array([5], dtype=np.int64)}
    names = LISTS + ["from_ids%d" % j for j in range(4)] + ["to_ids%d" % j for j in range(4)]
    for b in range(3):
        for l in range(3):
            k = "b%d_l%d_" % (b, l)
            rec[k + "next_frontier_hash"] = np.array([list_hash(out[k + "next_frontier"]), len(out[k + "next_frontier"])],
                                                     dtype=np.uint64)
            rec[k + "draws"] = out[k + "draws"]
            for g in range(4):
                hs = [[list_hash(out[k + "g%d_%s" % (g, nm)]), len(out[k + "g%d_%s" % (g, nm)])] for nm in names]
                rec[k + "g%d" % g] = np.array(hs, dtype=np.uint64)
    path = os.path.join(GOLD, "hashed_100k.npz")
    np.savez_compressed(path, **rec)
    print("%-24s nodes=%d edges=%d batches=3 -> %s (%d bytes)" % ("hashed_100k", n, indices.shape[0], path,
                                                                  os.path.getsize(path)))


def main():
    if not os.path.exists(HARNESS):
        sys.exit("build oracle/_ref/ref_harness first (make -C oracle ref)")
    os.makedirs(GOLD, exist_ok=True)
    for name, fn in CASES.items():
        indptr, indices, batches = fn()
        out = run_reference(indptr, indices, batches)
        out["indptr"] = indptr
        out["indices"] = indices
        out["n_batches"] = np.array([len(batches)], dtype=np.int64)
        # int32 storage keeps the fixtures small; every value fits (ids < 2^31)
        small = {}
        for k, v in out.items():
            v = np.asarray(v)
            assert v.size == 0 or (v.min() >= -2**31 and v.max() < 2**31)
            small[k] = v.astype(np.int32)
        path = os.path.join(GOLD, name + ".npz")
        np.savez_compressed(path, **small)
        print("%-24s nodes=%d edges=%d batches=%d -> %s (%d bytes)" % (
            name, indptr.shape[0] - 1, indices.shape[0], len(batches), path, os.path.getsize(path)))
    hashed_case()


if __name__ == "__main__":
    main()

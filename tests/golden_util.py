"""Helpers shared by the parity tests: load tests/golden/*.npz (vectors produced
by the unmodified reference, oracle/make_golden.py) and compare sample dicts."""
import os

import numpy as np

LIST_NAMES = ["in_nodes", "indptr", "out_nodes", "owned_out_nodes", "indices",
              "self_ids_in", "self_ids_out"]
GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["toy40", "degree_edges", "powerlaw2k", "selfloops_multiedges", "dense_small",
         "duplicate_seeds"]
# graph mode's specification (oracle orc_sample_graph) is only defined for distinct seeds
UNIQUE_SEED_CASES = [c for c in CASES if c != "duplicate_seeds"]


def load_case(name):
    d = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    g = {k: d[k].astype(np.int64) for k in d.files}
    nb = int(g["n_batches"][0])
    batches = []
    for b in range(nb):
        layers = []
        rec = {"seeds": g["b%d_seeds" % b], "layers": layers, "frontier": [], "nbr_counts": [],
               "nbr_flat": [], "draws": []}
        for l in range(3):
            k = "b%d_l%d_" % (b, l)
            rec["frontier"].append(g[k + "frontier"])
            rec["nbr_counts"].append(g[k + "nbr_counts"])
            rec["nbr_flat"].append(g[k + "nbr_flat"])
            rec["draws"].append(int(g[k + "draws"][0]))
            parts = []
            for p in range(4):
                kg = k + "g%d_" % p
                bp = {n: g[kg + n] for n in LIST_NAMES}
                bp["from_ids"] = [g[kg + "from_ids%d" % j] for j in range(4)]
                bp["to_ids"] = [g[kg + "to_ids%d" % j] for j in range(4)]
                bp["gpu_id"] = int(g[kg + "gpu_id"][0])
                parts.append(bp)
            layers.append(parts)
        rec["frontier"].append(g["b%d_l2_next_frontier" % b])
        batches.append(rec)
    return g["indptr"], g["indices"], batches


def assert_same_sample(got, want, what="", check_traversal=True, edge_stream=True):
    """Bit-exact comparison of two sample dicts (layers -> parts -> lists).

    check_traversal: also the frontiers, the draw counts and -- edge_stream=True -- the pre-dedup
    neighbour_sample stream (`nbr_counts` / `nbr_flat`).  A key `want` holds and `got` lacks is a FAILURE, not a
    skip: an engine that cannot export the stream (created without FLAG_KEEP_CANDIDATES, as the full-size cases
    are) is compared with an explicit edge_stream=False."""
    assert len(got["layers"]) == len(want["layers"]), what
    for l, (gl, wl) in enumerate(zip(got["layers"], want["layers"])):
        assert len(gl) == len(wl), what
        for p, (gb, wb) in enumerate(zip(gl, wl)):
            tag = "%s layer %d part %d " % (what, l, p)
            assert gb["gpu_id"] == wb["gpu_id"], tag + "gpu_id"
            for n in LIST_NAMES:
                np.testing.assert_array_equal(np.asarray(gb[n]), np.asarray(wb[n]), err_msg=tag + n)
            for j, (a, b) in enumerate(zip(gb["from_ids"], wb["from_ids"])):
                np.testing.assert_array_equal(np.asarray(a), np.asarray(b), err_msg=tag + "from_ids[%d]" % j)
            for j, (a, b) in enumerate(zip(gb["to_ids"], wb["to_ids"])):
                np.testing.assert_array_equal(np.asarray(a), np.asarray(b), err_msg=tag + "to_ids[%d]" % j)
    if check_traversal:
        for key in ("frontier",) + (("nbr_counts", "nbr_flat") if edge_stream else ()):
            if key not in want:
                continue
            assert key in got, "%s: `%s` missing from the sample under test" % (what, key)
            assert len(got[key]) == len(want[key]), what + key
            for l, (a, b) in enumerate(zip(got[key], want[key])):
                np.testing.assert_array_equal(np.asarray(a), np.asarray(b),
                                              err_msg="%s %s[%d]" % (what, key, l))
        if "draws" in want:
            assert "draws" in got, "%s: `draws` missing from the sample under test" % what
            assert list(got["draws"]) == list(want["draws"]), what + "draws"


def list_hash(a):
    """Same digest as oracle/make_golden.py::list_hash."""
    import hashlib
    b = np.ascontiguousarray(np.asarray(a, dtype="<i8")).tobytes()
    return int(np.frombuffer(hashlib.blake2b(b, digest_size=8).digest(), dtype="<u8")[0])


def load_hashed_case():
    """tests/golden/hashed_100k.npz: the unmodified reference on a 100k-node products-like graph,
    3 consecutive minibatches of 1024, every exported list pinned by (digest, length)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "occ-gnn_amd"))
    from cslicer import l0
    d = np.load(os.path.join(GOLDEN_DIR, "hashed_100k.npz"))
    n, deg1000, seed = (int(x) for x in d["graph"])
    indptr, indices = l0.synth_graph(n, deg1000 / 1000.0, seed=seed)
    assert [int(indptr.sum()), int(indices.sum())] == [int(x) for x in d["graph_csum"]], "generator drifted"
    perm = np.random.default_rng(int(d["perm_seed"][0])).permutation(n)
    batches = [perm[i * 1024:(i + 1) * 1024] for i in range(3)]
    return indptr, indices, batches, d


def assert_matches_hashed(sample, d, b, what=""):
    names = LIST_NAMES + ["from_ids%d" % j for j in range(4)] + ["to_ids%d" % j for j in range(4)]
    for l in range(3):
        for g in range(4):
            want = d["b%d_l%d_g%d" % (b, l, g)]
            bp = sample["layers"][l][g]
            for k, nm in enumerate(names):
                lst = bp[nm] if nm in bp else bp[nm[:-1]][int(nm[-1])]
                assert (list_hash(lst), len(lst)) == (int(want[k][0]), int(want[k][1])), \
                    "%s batch %d layer %d part %d %s" % (what, b, l, g, nm)
        if "frontier" in sample:
            nf = d["b%d_l%d_next_frontier_hash" % (b, l)]
            fr = sample["frontier"][l + 1]
            assert (list_hash(fr), len(fr)) == (int(nf[0]), int(nf[1])), "%s batch %d next frontier %d" % (what, b, l)
        if "draws" in sample:
            assert int(sample["draws"][l]) == int(d["b%d_l%d_draws" % (b, l)][0])

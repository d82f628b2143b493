"""Helpers shared by the parity tests: load tests/golden/*.npz (vectors produced
by the unmodified reference, oracle/make_golden.py) and compare sample dicts."""
import os

import numpy as np

LIST_NAMES = ["in_nodes", "indptr", "out_nodes", "owned_out_nodes", "indices",
              "self_ids_in", "self_ids_out"]
GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["toy40", "degree_edges", "powerlaw2k", "selfloops_multiedges", "dense_small",
         "duplicate_seeds"]
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


def assert_same_sample(got, want, what="", check_traversal=True):
    """Bit-exact comparison of two sample dicts (layers -> parts -> lists)."""
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
        for key in ("frontier", "nbr_counts", "nbr_flat"):
            if key in got and key in want:
                assert len(got[key]) == len(want[key]), what + key
                for l, (a, b) in enumerate(zip(got[key], want[key])):
                    np.testing.assert_array_equal(np.asarray(a), np.asarray(b),
                                                  err_msg="%s %s[%d]" % (what, key, l))
        if "draws" in got and "draws" in want:
            assert list(got["draws"]) == list(want["draws"]), what + "draws"

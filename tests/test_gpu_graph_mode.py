"""CSL_MODE_GRAPH: the object the reference meant to export (real slice CSR, per-peer boundary
lists).  The reference itself never builds it (bipartite.h:55-66, slicer.cpp:41-42), so parity is
against the sequential specification in the oracle (orc_sample_graph): bit-exact, plus structural
invariants that tie the graph object back to the strict object and to the input graph."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GKEYS = ["in_nodes", "out_nodes", "indptr", "indices", "owned_out_nodes", "self_ids_in", "self_ids_out",
         "owned_degree"]


@pytest.fixture(scope="module")
def abi():
    from cslicer import _abi
    _abi.load()
    return _abi


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def assert_same_graph(got, want, what=""):
    assert len(got["layers"]) == len(want["layers"])
    for l, (gl, wl) in enumerate(zip(got["layers"], want["layers"])):
        for p, (gb, wb) in enumerate(zip(gl, wl)):
            tag = "%s layer %d part %d " % (what, l, p)
            for k in GKEYS:
                np.testing.assert_array_equal(gb[k], wb[k], err_msg=tag + k)
            for j in range(len(wl)):
                np.testing.assert_array_equal(gb["from_ids"][j], wb["from_ids"][j], err_msg=tag + "from_ids[%d]" % j)
                np.testing.assert_array_equal(gb["to_ids"][j], wb["to_ids"][j], err_msg=tag + "to_ids[%d]" % j)
    for l, (a, b) in enumerate(zip(got["frontier"], want["frontier"])):
        np.testing.assert_array_equal(a, b, err_msg="%s frontier[%d]" % (what, l))


def check_graph_invariants(d, indptr, indices, P):
    """Every edge of the slice CSRs is a real edge of the input graph, boundary lists pair up."""
    for l, parts in enumerate(d["layers"]):
        fr = d["frontier"][l]
        edges = 0
        for g, bp in enumerate(parts):
            ip, ix = bp["indptr"], bp["indices"]
            assert ip[0] == 0 and len(ip) == len(bp["out_nodes"]) + 1 and ip[-1] == len(ix)
            assert (np.diff(ip) >= 0).all()
            assert ((ix >= 0) & (ix < len(bp["in_nodes"]))).all()
            edges += len(ix)
            src = bp["in_nodes"][ix]
            dst = np.repeat(bp["out_nodes"], np.diff(ip))
            assert (src % P == g).all()
            for k in range(0, len(src), max(1, len(src) // 200)):   # spot-check real adjacency
                row = indices[indptr[dst[k]]:indptr[dst[k] + 1]]
                assert src[k] in row
            own = bp["owned_out_nodes"]
            np.testing.assert_array_equal(bp["out_nodes"][own], fr[fr % P == g])
            np.testing.assert_array_equal(bp["in_nodes"][bp["self_ids_in"]], fr[fr % P == g])
            np.testing.assert_array_equal(bp["self_ids_out"], own)
        for g in range(P):
            for p in range(P):
                a, b = parts[g]["from_ids"][p], parts[p]["to_ids"][g]
                assert len(a) == len(b)
                if g == p:
                    assert len(a) == 0
                    continue
                np.testing.assert_array_equal(parts[g]["out_nodes"][a], parts[p]["out_nodes"][b])
                assert (parts[g]["out_nodes"][a] % P == p).all()
        # total degree of owned nodes == all edges of the layer
        assert sum(int(bp["owned_degree"].sum()) for bp in parts) == edges


def check_transposed(d, deepest_too=False):
    """FLAG_TRANSPOSE: t_indptr / t_indices of every layer but the deepest (FLAG_TRANSPOSE_ALL: of every layer) are the
    slice CSR + self lists sorted by source (cslicer_hip.h, CSL_T_INDPTR): per in node ~r of its self entry, then the
    out rows of its edges ascending."""
    L = len(d["layers"])
    for l, parts in enumerate(d["layers"]):
        for g, bp in enumerate(parts):
            tag = "layer %d part %d " % (l, g)
            if l == L - 1 and not deepest_too:
                assert len(bp["t_indptr"]) == 0 and len(bp["t_indices"]) == 0, tag
                continue
            n_in = len(bp["in_nodes"])
            if len(bp["out_nodes"]) == 0 and n_in == 0 and len(bp["t_indptr"]) == 0:
                continue   # an empty layer has no row pointers at all (like indptr)
            rows = np.repeat(np.arange(len(bp["out_nodes"]), dtype=np.int64), np.diff(bp["indptr"]))
            u = np.concatenate([bp["indices"].astype(np.int64), bp["self_ids_in"].astype(np.int64)])
            val = np.concatenate([rows, ~bp["self_ids_out"].astype(np.int64)])
            order = np.lexsort((val, u))
            want_ptr = np.concatenate([[0], np.cumsum(np.bincount(u, minlength=n_in))])
            np.testing.assert_array_equal(bp["t_indptr"], want_ptr, err_msg=tag + "t_indptr")
            got_idx, lens = bp["t_indices"].copy(), np.diff(want_ptr)
            for u in np.flatnonzero(lens > 128):         # a hub's list (> CSL_T_SORTED_MAX) comes in unspecified order
                got_idx[want_ptr[u]:want_ptr[u + 1]] = np.sort(got_idx[want_ptr[u]:want_ptr[u + 1]])
            np.testing.assert_array_equal(got_idx, val[order], err_msg=tag + "t_indices")
            assert bp["t_max_len"] == (int(lens.max()) if len(lens) else 0), tag


CONFIGS = [
    (3000, 20.0, 4, (10, 10, 10), 64, 2),
    (5000, 6.0, 1, (15, 10, 5), 100, 1),
    (4000, 25.0, 8, (15, 10, 5), 256, 2),
    (4000, 25.0, 3, (5, 5), 200, 3),
    (20000, 50.5, 4, (15, 10, 5), 1024, 2),
]


@pytest.mark.parametrize("cfg", CONFIGS, ids=[str(c) for c in CONFIGS])
def test_graph_mode_matches_specification(abi, orc, cfg):
    from cslicer import l0
    n, deg, P, fan, B, S = cfg
    indptr, indices = l0.synth_graph(n, deg, seed=n + P)
    perm = np.random.default_rng(3).permutation(n)
    # (with the slices by source: they must not disturb the lists of the specification)
    e = abi.Engine(indptr, indices, n_parts=P, fanouts=fan, max_batch=B, n_streams=S, mode=abi.MODE_GRAPH,
                   flags=abi.FLAG_TRANSPOSE)
    e.set_nodes(perm)
    oracles = [orc.Oracle(indptr, indices, n_parts=P, fanouts=fan) for _ in range(S)]
    for r in range(2):
        e.submit_round(r * S, B, S)
        for s in range(S):
            seeds = perm[(r * S + s) * B:(r * S + s + 1) * B]
            want = oracles[s].sample_graph(seeds)
            got = e.graph_dict(s)
            assert_same_graph(got, want, what="round %d stream %d" % (r, s))
            assert got["draws_total"] == want["draws_total"]
            check_graph_invariants(got, indptr, indices, P)
            check_transposed(got)
    e.close()


def test_transpose_all_covers_the_deepest_layer(abi, orc):
    from cslicer import l0
    indptr, indices = l0.synth_graph(5000, 18.0, seed=8)
    perm = np.random.default_rng(1).permutation(5000)
    with pytest.raises(abi.CslError):
        abi.Engine(indptr, indices, max_batch=64, mode=abi.MODE_GRAPH, flags=abi.FLAG_TRANSPOSE_ALL)   # widens TRANSPOSE only
    e = abi.Engine(indptr, indices, n_parts=3, fanouts=(6, 5, 4), max_batch=200, n_streams=2, mode=abi.MODE_GRAPH,
                   flags=abi.FLAG_TRANSPOSE | abi.FLAG_TRANSPOSE_ALL)
    e.set_nodes(perm)
    o = [orc.Oracle(indptr, indices, n_parts=3, fanouts=(6, 5, 4)) for _ in range(2)]
    e.submit_round(0, 200, 2)
    for s in range(2):
        got = e.graph_dict(s)
        assert_same_graph(got, o[s].sample_graph(perm[s * 200:(s + 1) * 200]), what="stream %d" % s)
        check_transposed(got, deepest_too=True)
    e.close()


def test_slices_by_source_with_hub_nodes(abi, orc):
    """Three hub nodes are neighbours of every node: each is the source of thousands of a minibatch's edges, i.e. its
    list in the slice by source is thousands of entries long: reported in t_max_len, left in unspecified order
    (CSL_T_SORTED_MAX), complete as a multiset."""
    n, deg = 30000, 12
    rng = np.random.default_rng(4)
    nb = rng.integers(0, n, size=(n, deg))
    nb[:, :3] = np.array([7, 11, 13])                       # the hubs
    nb[[7, 11, 13]] = rng.integers(100, n, size=(3, deg))   # (no self loops on them)
    indptr = np.arange(n + 1, dtype=np.int64) * deg
    indices = np.sort(nb, axis=1).reshape(-1).astype(np.int64)
    seeds = rng.permutation(n)[:1024]
    e = abi.Engine(indptr, indices, n_parts=2, fanouts=(8, 6), max_batch=1024, mode=abi.MODE_GRAPH,
                   flags=abi.FLAG_TRANSPOSE | abi.FLAG_TRANSPOSE_ALL)
    e.submit_seeds([seeds])
    got = e.graph_dict(0)
    assert_same_graph(got, orc.Oracle(indptr, indices, n_parts=2, fanouts=(8, 6)).sample_graph(seeds))
    check_transposed(got, deepest_too=True)
    longest = max(bp["t_max_len"] for parts in got["layers"] for bp in parts)
    assert longest > 1000, longest
    e.close()


def test_transpose_flag_needs_graph_mode_and_is_off_by_default(abi):
    from cslicer import l0
    indptr, indices = l0.synth_graph(500, 8.0, seed=1)
    with pytest.raises(abi.CslError):
        abi.Engine(indptr, indices, max_batch=16, flags=abi.FLAG_TRANSPOSE)     # strict mode: nothing to transpose
    e = abi.Engine(indptr, indices, max_batch=16, mode=abi.MODE_GRAPH)
    e.submit_seeds([np.arange(16)])
    d = e.graph_dict(0)
    assert all(len(bp["t_indptr"]) == 0 and len(bp["t_indices"]) == 0 for parts in d["layers"] for bp in parts)
    e.close()


def test_graph_mode_golden_graphs(abi, orc):
    # the reference's golden INPUTS (self loops, multi-edges, isolated nodes, degree 9/10/11):
    # sampling and frontiers must still be the reference's, the graph object the specification's
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from golden_util import UNIQUE_SEED_CASES, load_case
    for case in UNIQUE_SEED_CASES:
        indptr, indices, batches = load_case(case)
        mb = max(len(b["seeds"]) for b in batches)
        e = abi.Engine(indptr, indices, max_batch=mb, mode=abi.MODE_GRAPH)
        o = orc.Oracle(indptr, indices)
        for b, gold in enumerate(batches):
            e.submit_seeds([gold["seeds"]])
            got = e.graph_dict(0)
            assert_same_graph(got, o.sample_graph(gold["seeds"]), what="%s batch %d" % (case, b))
            for l in range(4):   # frontiers are the unmodified reference's
                np.testing.assert_array_equal(got["frontier"][l], gold["frontier"][l])
        e.close()


def test_strict_and_graph_share_sampling(abi):
    from cslicer import l0
    indptr, indices = l0.synth_graph(6000, 30.0, seed=4)
    perm = np.random.default_rng(9).permutation(6000)
    es = abi.Engine(indptr, indices, max_batch=128, n_streams=2)
    eg = abi.Engine(indptr, indices, max_batch=128, n_streams=2, mode=abi.MODE_GRAPH)
    for e in (es, eg):
        e.set_nodes(perm)
        e.submit_round(0, 128, 2)
    for s in range(2):
        ds, dg = es.sample_dict(s), eg.graph_dict(s)
        for l in range(4):
            np.testing.assert_array_equal(ds["frontier"][l], dg["frontier"][l])
        assert ds["draws_total"] == dg["draws_total"]
        for l in range(3):
            for g in range(4):
                a, b = ds["layers"][l][g], dg["layers"][l][g]
                # graph in_nodes = strict in_nodes + the slice's own frontier nodes (as sets: a self
                # entry can move a node's first occurrence forward)
                fr = ds["frontier"][l]
                want = np.union1d(a["in_nodes"], fr[fr % 4 == g])
                np.testing.assert_array_equal(np.sort(b["in_nodes"]), want)
                # strict out_nodes are a subsequence of graph out_nodes
                keep = np.isin(b["out_nodes"], a["out_nodes"])
                np.testing.assert_array_equal(b["out_nodes"][keep], a["out_nodes"])
    es.close()
    eg.close()


@pytest.mark.parametrize("mode", ["graph", "strict"])
def test_part_mask_materialises_only_the_asked_slices(mode):
    """csl_config.part_mask (one process per part asks for its own slice only): the lists of the parts in the
    mask, every size/offset in the meta, the frontiers and the draw counts equal the unmasked engine's."""
    from cslicer import _abi, l0
    indptr, indices = l0.synth_graph(5000, 22.0, seed=4)
    rng = np.random.default_rng(6)
    wl = rng.integers(0, 4, size=5000).astype(np.int32)
    perm = rng.permutation(5000)
    md = _abi.MODE_GRAPH if mode == "graph" else _abi.MODE_STRICT
    full = _abi.Engine(indptr, indices, n_parts=4, fanouts=(10, 5, 5), max_batch=300, n_streams=2, mode=md, workload=wl)
    full.set_nodes(perm)
    full.submit_round(0, 300, 2)
    for mask in (1 << 2, (1 << 0) | (1 << 3)):
        e = _abi.Engine(indptr, indices, n_parts=4, fanouts=(10, 5, 5), max_batch=300, n_streams=2, mode=md,
                        workload=wl, part_mask=mask)
        e.set_nodes(perm)
        e.submit_round(0, 300, 2)
        for s in range(2):
            m0, m1 = full.meta(s), e.meta(s)
            assert bytes(m0) == bytes(m1)                      # sizes, offsets, rng positions: all of it
            for l in range(3):
                np.testing.assert_array_equal(full.copy_frontier(l + 1, s), e.copy_frontier(l + 1, s))
                for g in range(4):
                    if not (mask >> g) & 1:
                        continue
                    for k in range(_abi.NUM_LISTS):
                        np.testing.assert_array_equal(full.copy_list(l, k, g, s, meta=m0), e.copy_list(l, k, g, s, meta=m1),
                                                      err_msg="mask %x layer %d part %d kind %d" % (mask, l, g, k))
        e.close()
    full.close()
    with pytest.raises(_abi.CslError):
        _abi.Engine(indptr, indices, n_parts=4, part_mask=1 << 5)

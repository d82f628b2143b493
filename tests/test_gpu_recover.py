"""Recovery from CSL_ERR_FRONTIER_CAP: the reference's vectors grow (cslicer/bipartite.h:55-66), a frontier that outgrows
a USER-given capacity poisons the sample here.  _abi.Engine then builds a fresh engine with the worst-case capacities and
replays its submissions: every sample, the overflowing one and all later ones, must be what an engine with enough room
produces (same mt19937 positions)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same(a, b):
    assert a["draws_total"] == b["draws_total"] and a["sampled_edges"] == b["sampled_edges"]
    for la, lb in zip(a["layers"], b["layers"]):
        for pa, pb in zip(la, lb):
            for k in ("in_nodes", "out_nodes", "owned_out_nodes", "self_ids_in", "self_ids_out"):
                np.testing.assert_array_equal(pa[k], pb[k])
    for fa, fb in zip(a["frontier"], b["frontier"]):
        np.testing.assert_array_equal(fa, fb)


def test_frontier_capacity_overflow_is_recovered_by_a_fresh_engine_and_replay():
    from cslicer import _abi, l0
    _abi.load()
    n, B, fan = 20000, 64, (10, 10, 10)
    indptr, indices = l0.synth_graph(n, 20.0, seed=4)
    perm = np.random.default_rng(0).permutation(n)
    kw = dict(n_parts=4, fanouts=fan, max_batch=B, n_streams=2, n_slots=2)
    ref = _abi.Engine(indptr, indices, **kw)
    # room for the first rounds' frontiers would need ~B * 11^l entries: 2000 is enough for layer 1, not for layer 2
    small = _abi.Engine(indptr, indices, frontier_cap=[B, 2000, 2000, 2000], **kw)
    for e in (ref, small):
        e.set_nodes(perm)
    want, got = [], []
    for r in range(3):
        for e, out in ((ref, want), (small, got)):
            e.submit_round(2 * r, B, 2, slot=r % 2)
            out.append([e.sample_dict(s, r % 2) for s in range(2)])
    assert small.recovered == 1 and ref.recovered == 0
    for ra, rb in zip(want, got):
        for a, b in zip(ra, rb):
            _same(a, b)
    # without recovery the overflow stays a loud error
    loud = _abi.Engine(indptr, indices, frontier_cap=[B, 2000, 2000, 2000], recover=False, **kw)
    loud.set_nodes(perm)
    loud.submit_round(0, B, 2, slot=0)
    with pytest.raises(_abi.CslError):
        loud.sample_dict(0, 0)
    for e in (ref, small, loud):
        e.close()

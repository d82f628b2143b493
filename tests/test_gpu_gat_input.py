"""The attention model's input layer as aggregate-then-project (include/cslicer_aggr.h: csl_gat_in_fwd_f32 /
csl_gat_in_bwd_f32, cslicer.aggr.GatInputLayer) against (a) the project-then-aggregate layer it replaces
(aggr.GatLayerLocal on the gathered rows: same definition, other association of the sums) and (b) the unsplit definition
in float64 on the ORACLE's traversal at BASELINE config 5's widths.  The reference has only a stub layer
(python/layers/dist_gatconv.py:3-6) and `attention_gather` (python/data/bipartite.py:75-80), no goldens: "parity
unpinned" by the reference.  Tolerances: outputs 1e-5 (north_star) against the fp32 layer, 1e-4 relative against
float64 through three layers; gradients within 1e-4 (fp32 layer) / 1e-3 (float64) of the tensor's largest entry."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from cslicer import _abi, aggr, splitgnn
    _abi.load()
    return _abi, aggr, splitgnn


def _scale(t):
    return max(1e-3, float(t.abs().max()))


@pytest.mark.parametrize("libgemm", [False, True])
@pytest.mark.parametrize("heads,D,F0,fan,B,elu", [(8, 32, 100, 10, 256, True), (4, 12, 20, 7, 64, False),
                                                  (1, 8, 128, 32, 50, True), (2, 16, 4, 3, 300, True),
                                                  (8, 4, 36, 17, 128, False), (4, 64, 128, 12, 100, True),
                                                  (2, 32, 64, 20, 77, False), (1, 16, 112, 5, 33, True)])
def test_input_layer_matches_the_projecting_layer(mods, heads, D, F0, fan, B, elu, libgemm, monkeypatch):
    """one layer, same slice, same parameters: output, weight / attention / bias gradients.  The block-diagonal projection
    runs on the MFMA kernels where they cover the shape (D in {16, 32, 64}) and as batched library GEMMs otherwise
    (libgemm: forced)"""
    abi, aggr, sg = mods
    if libgemm:
        if not aggr._lib().csl_gat_in_proj_ok(heads, F0, D):
            pytest.skip("the shape already runs on the library GEMMs")
        monkeypatch.setenv("CSLICER_GAT_IN_LIBGEMM", "1")
    from cslicer import l0
    torch.manual_seed(heads * 100 + fan)
    n = 5000
    indptr, indices = l0.synth_graph(n, 9.0, seed=fan)      # (degrees from 0 up: rows without edges occur)
    eng = abi.Engine(indptr, indices, n_parts=1, fanouts=(fan,), max_batch=B, mode=abi.MODE_GRAPH)
    eng.submit_seeds([np.random.default_rng(fan).permutation(n)[:B]])
    sl = sg.slices_of(eng)[0][0]
    assert aggr.gat_input_ok(heads, F0, fan)
    conv = sg.DistGATConv(F0, D, heads).cuda()
    with torch.no_grad():
        conv.bias.normal_(0, 0.1)
    table = torch.randn(n, F0, device="cuda")
    w = torch.randn(sl.n_out, heads * D, device="cuda")
    res = []
    for new in (True, False):
        conv.zero_grad()
        if new:
            out = aggr.GatInputLayer.apply(table, sl.in_nodes, conv.fc.weight, conv.attn_l, conv.attn_r, conv.bias, sl.indptr,
                                           sl.indices, sl.self_ids_in, sl.n_out, sl.n_edges, fan, conv.slope, elu, 0, False)
        else:
            x = table[sl.in_nodes.long()]
            out = aggr.GatLayerLocal.apply(x, conv.fc.weight, conv.attn_l, conv.attn_r, conv.bias, sl.indptr, sl.indices,
                                           sl.self_ids_in, sl.n_out, conv.slope, elu, 0, lambda gz, xp: gz.t() @ xp)
        (out * w).sum().backward()
        res.append((out.detach().clone(), [p.grad.clone() for p in conv.parameters()]))
    torch.testing.assert_close(res[0][0], res[1][0], rtol=1e-5, atol=1e-5 * _scale(res[1][0]))
    for (name, _), a_, b_ in zip(conv.named_parameters(), res[0][1], res[1][1]):
        torch.testing.assert_close(a_, b_, rtol=1e-4, atol=1e-4 * _scale(b_), msg=lambda m_: name + ": " + m_)
    eng.close()


def test_input_layer_without_edges_and_without_rows(mods):
    """a slice whose rows have no sampled edge gives the bias (ELU'd), zero attention gradients; no rows: no launch"""
    abi, aggr, sg = mods
    H, D, F0 = 8, 8, 16
    conv = sg.DistGATConv(F0, D, H).cuda()
    with torch.no_grad():
        conv.bias.normal_(0, 1.0)
    table = torch.randn(10, F0, device="cuda")
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device="cuda")                            # noqa: E731
    out = aggr.GatInputLayer.apply(table, i32([3, 4, 5]), conv.fc.weight, conv.attn_l, conv.attn_r, conv.bias,
                                   i32([0, 0, 0, 0]), i32([]), i32([0, 1, 2]), 3, 0, 4, conv.slope, True, 0, False)
    torch.testing.assert_close(out, torch.nn.functional.elu(conv.bias).expand(3, -1))
    out.sum().backward()
    assert float(conv.attn_l.grad.abs().max()) == 0.0 and float(conv.fc.weight.grad.abs().max()) == 0.0
    out = aggr.GatInputLayer.apply(table, i32([]), conv.fc.weight, conv.attn_l, conv.attn_r, conv.bias, i32([0]), i32([]),
                                   i32([]), 0, 0, 4, conv.slope, True, 0, False)
    assert out.shape == (0, H * D)


@pytest.mark.parametrize("heads,hidden,fan", [(8, 32, (10, 10, 10)), (4, 8, (5, 4, 3))])
def test_model_on_the_feature_table_matches_the_gathered_input(mods, heads, hidden, fan, monkeypatch):
    """DistGATModel.forward_parts on aggr.FeatureRows (deepest layer = GatInputLayer, no by-source slice for it) against
    the same model on the gathered, padded input (GatLayerLocal everywhere)"""
    abi, aggr, sg = mods
    from cslicer import l0
    torch.manual_seed(1)
    n, F0, classes, B = 60_000, 100, 47, 256
    indptr, indices = l0.synth_graph(n, 30.0, seed=1)
    table = torch.randn(n, F0, device="cuda")
    model = sg.DistGATModel(F0, hidden, classes, heads=heads, n_layers=3).cuda()
    with torch.no_grad():
        for conv in model.convs:
            conv.bias.normal_(0, 0.1)
    seeds = np.random.default_rng(2).permutation(n)[:B]
    w = torch.randn(B, classes, device="cuda")
    res = []
    for new in (True, False):
        eng = abi.Engine(indptr, indices, n_parts=1, fanouts=fan, max_batch=B, mode=abi.MODE_GRAPH,
                         flags=abi.FLAG_TRANSPOSE | (0 if new else abi.FLAG_TRANSPOSE_ALL))
        eng.submit_seeds([seeds])
        slices = sg.slices_of(eng)
        deep = slices[2][0]
        model.zero_grad()
        if new:
            x = aggr.FeatureRows(table, deep.in_nodes)
        else:
            x = aggr.padded_rows(deep.n_in, F0, sg.ROW_PAD, table.device)
            aggr.gather_rows(table, deep.in_nodes, out=x.t[:x.n])
        out = model.forward_parts(slices, {0: x})[0]
        (out * w).sum().backward()
        torch.cuda.synchronize()
        res.append((out.detach().clone(), [p.grad.clone() for p in model.parameters()]))
        eng.close()
    torch.testing.assert_close(res[0][0], res[1][0], rtol=1e-4, atol=1e-5 * _scale(res[1][0]))
    for (name, _), a_, b_ in zip(model.named_parameters(), res[0][1], res[1][1]):
        torch.testing.assert_close(a_, b_, rtol=1e-3, atol=2e-4 * _scale(b_), msg=lambda m_: name + ": " + m_)


def test_config5_widths_against_float64_on_the_oracle_traversal(mods):
    """BASELINE configs[4]'s layer shape on ONE part: features 100, 8 heads x 32, 47 classes, fanout 10/10/10, batch 1024 on
    a 400 k-node products-like graph, the model on the feature table (GatInputLayer + two GatLayerLocal), against the unsplit
    definition in float64 with index ops on the oracle's traversal of the same seeds."""
    abi, aggr, sg = mods
    from cslicer import l0
    from oracle import oracle as orc
    from test_gpu_gat import _dense_gat_vectorised
    import copy
    torch.manual_seed(2)
    n, F0, hidden, classes, B, heads, fan = 400_000, 100, 32, 47, 1024, 8, (10, 10, 10)
    indptr, indices = l0.synth_graph(n, 50.5, seed=0)
    seeds = np.random.default_rng(4).permutation(n)[:B]
    eng = abi.Engine(indptr, indices, n_parts=1, fanouts=fan, max_batch=B, mode=abi.MODE_GRAPH, flags=abi.FLAG_TRANSPOSE)
    eng.submit_seeds([seeds])
    slices = sg.slices_of(eng)
    feats = torch.randn(n, F0, device="cuda")
    model = sg.DistGATModel(F0, hidden, classes, heads=heads, n_layers=3).cuda()
    with torch.no_grad():
        for conv in model.convs:
            conv.bias.normal_(0, 0.1)
    out = model.forward_parts(slices, {0: aggr.FeatureRows(feats, slices[2][0].in_nodes)})[0]
    assert out.shape == (B, classes)
    w = torch.randn(n, classes, device="cuda")
    seeds_t = torch.from_numpy(seeds).cuda()
    (out * w[seeds_t]).sum().backward()
    got = [p.grad.clone() for p in model.parameters()]
    model64 = copy.deepcopy(model).double()
    model64.zero_grad()
    trav = orc.Oracle(indptr, indices, n_parts=1, fanouts=fan).sample(seeds)
    ref = _dense_gat_vectorised(model64, trav, feats.double())
    torch.testing.assert_close(out.detach(), ref[seeds_t].float(), rtol=1e-4, atol=1e-5)
    (ref[seeds_t] * w[seeds_t].double()).sum().backward()
    for (name, p_), gg in zip(model64.named_parameters(), got):
        torch.testing.assert_close(gg, p_.grad.float(), rtol=2e-3, atol=1e-3 * _scale(p_.grad),
                                   msg=lambda m_: "grad " + name + ": " + m_)
    eng.close()

"""GAT attention aggregation (include/cslicer_aggr.h: csl_gat_fwd_f32 / csl_gat_bwd_f32) and the split-parallel
DistGATConv / DistGATModel built on it (BASELINE config 5).  The reference has only a stub layer
(python/layers/dist_gatconv.py:3-6) and `attention_gather` (python/data/bipartite.py:75-80), no goldens:
"parity unpinned" by the reference; the results are pinned by plain torch fp32 computations of the same
definition.  Tolerance 1e-5 forward (north_star), 1e-4 on gradients."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from cslicer import _abi, aggr, splitgnn
    _abi.load()
    return _abi, aggr, splitgnn


def _torch_partial(el, er, z, indptr, indices, n_rows, H, D, slope):
    """(m, s, n) of the definition in cslicer_aggr.h with index ops; m detached like the kernel's."""
    deg = (indptr[1:] - indptr[:-1]).long()
    rows = torch.repeat_interleave(torch.arange(n_rows, device=z.device), deg)
    src = indices.long()
    score = torch.nn.functional.leaky_relu(el[src] + er[rows], slope)                   # [E, H]
    m = torch.full((n_rows, H), -1e30, device=z.device).scatter_reduce(
        0, rows[:, None].expand(-1, H), score.detach(), "amax", include_self=True)
    p = torch.exp(score - m[rows])
    s = torch.zeros((n_rows, H), device=z.device).index_add(0, rows, p)
    n = torch.zeros((n_rows, H, D), device=z.device).index_add(0, rows, p[:, :, None] * z[src].view(-1, H, D))
    return m, s, n.view(n_rows, H * D)


@pytest.mark.parametrize("H,D", [(8, 32), (4, 12), (1, 4), (3, 48), (8, 64), (2, 256)])
def test_gat_partial_aggregate_forward_backward(mods, H, D):
    _, aggr, _ = mods
    rng = np.random.default_rng(H * 100 + D)
    n_rows, n_src = 333, 211
    deg = rng.integers(0, 12, size=n_rows)
    deg[::9] = 0                                       # rows without local edges
    indptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.cumsum(deg, out=indptr[1:])
    indices = rng.integers(0, n_src, size=int(indptr[-1]))
    ip, ix = torch.from_numpy(indptr).int().cuda(), torch.from_numpy(indices).int().cuda()
    torch.manual_seed(D)
    leaves = [torch.randn(n_src, H, device="cuda"), torch.randn(n_rows, H, device="cuda"),
              torch.randn(n_src, H * D, device="cuda")]
    gs, gn = torch.randn(n_rows, H, device="cuda"), torch.randn(n_rows, H * D, device="cuda")
    res = []
    for which in ("hip", "torch"):
        el, er, z = [t.clone().requires_grad_() for t in leaves]
        if which == "hip":
            m, s, n = aggr.GatAggregate.apply(el, er, z, ip, ix, n_rows, H, D, 0.2)
        else:
            m, s, n = _torch_partial(el, er, z, ip, ix, n_rows, H, D, 0.2)
        ((s * gs).sum() + (n * gn).sum()).backward()
        res.append((m.detach(), s.detach(), n.detach(), el.grad, er.grad, z.grad))
    for name, a, b in zip(["m", "s", "n", "grad el", "grad er", "grad z"], res[0], res[1]):
        tol = 1e-5 if name in ("m", "s", "n") else 1e-4
        torch.testing.assert_close(a, b, rtol=tol, atol=tol * max(1.0, float(b.abs().max())), msg=lambda x: name + ": " + x)


def _dense_gat(model, trav, feats):
    """Unsplit torch fp32 GAT on the same sampled computation graph (CPU, python loops)."""
    L = len(trav["nbr_counts"])
    h = feats
    for k, conv in enumerate(model.convs):
        l = L - 1 - k
        fr = trav["frontier"][l]
        counts, flat = trav["nbr_counts"][l], trav["nbr_flat"][l]
        z = h @ conv.fc.weight.t()                       # the definition, not the layer's own project()
        zv = z.view(-1, conv.H, conv.D)
        el, er = (zv * conv.attn_l).sum(-1), (zv * conv.attn_r).sum(-1)
        rows = []
        pos = 0
        for i, nd1 in enumerate(fr):
            nb = flat[pos + 1:pos + counts[i]]
            pos += counts[i]
            nb = torch.as_tensor(nb[nb != nd1]).long()
            if len(nb):
                a = torch.softmax(torch.nn.functional.leaky_relu(el[nb] + er[nd1], conv.slope), dim=0)   # [deg, H]
                rows.append((a[:, :, None] * zv[nb]).sum(0).reshape(-1) + conv.bias)
            else:
                rows.append(conv.bias + 0 * z[0])
        new = torch.zeros(h.shape[0], conv.H * conv.D).index_copy(0, torch.as_tensor(np.asarray(fr)).long(), torch.stack(rows))
        if k + 1 < len(model.convs):
            h = torch.nn.functional.elu(new)
        else:
            h = new.view(-1, model.heads, conv.D).mean(1)[:, :model.n_classes]
    return h


@pytest.mark.parametrize("P,fan,heads", [(4, (10, 10, 10), 8), (2, (6, 4), 4), (1, (5, 5), 2)])
def test_split_parallel_gat_matches_dense_reference(mods, P, fan, heads):
    abi, aggr, sg = mods
    from cslicer import l0
    from oracle import oracle as orc
    torch.manual_seed(1)
    n, F0, hidden, classes, B = 2500, 20, 8, 7, 48
    indptr, indices = l0.synth_graph(n, 10.0, seed=6)
    seeds = np.random.default_rng(3).permutation(n)[:B]
    eng = abi.Engine(indptr, indices, n_parts=P, fanouts=fan, max_batch=B, mode=abi.MODE_GRAPH)
    eng.submit_seeds([seeds])
    slices = sg.slices_of(eng)
    L = len(fan)
    feats = torch.randn(n, F0)
    model = sg.DistGATModel(F0, hidden, classes, heads=heads, n_layers=L)
    with torch.no_grad():
        for conv in model.convs:
            conv.bias.normal_(0, 0.1)
    gm = sg.DistGATModel(F0, hidden, classes, heads=heads, n_layers=L).cuda()
    gm.load_state_dict(model.state_dict())
    x = {g: feats[slices[L - 1][g].in_nodes.cpu().long()].cuda().requires_grad_() for g in range(P)}
    out = gm.forward_parts(slices, x)
    trav = orc.Oracle(indptr, indices, n_parts=P, fanouts=fan).sample(seeds)
    fin = feats.clone().requires_grad_()
    ref = _dense_gat(model, trav, fin)
    for g in range(P):
        own = seeds[seeds % P == g]
        torch.testing.assert_close(out[g].detach().cpu(), ref[own].detach(), rtol=1e-5, atol=1e-5)
    w = torch.randn(n, classes)
    (ref[seeds] * w[seeds]).sum().backward()
    sum((out[g] * w[seeds[seeds % P == g]].cuda()).sum() for g in range(P)).backward()
    for (na, pa), (nb, pb) in zip(gm.named_parameters(), model.named_parameters()):
        torch.testing.assert_close(pa.grad.cpu(), pb.grad, rtol=1e-4, atol=1e-5, msg=lambda x: "grad " + na + ": " + x)
    for g in range(P):
        ids = slices[L - 1][g].in_nodes.cpu().long()
        torch.testing.assert_close(x[g].grad.cpu(), fin.grad[ids], rtol=1e-4, atol=1e-5)
    eng.close()


def _dense_gat_vectorised(model, trav, feats):
    """The unsplit definition on the same sampled computation graph with index ops (any device), for sizes the
    python-loop reference above cannot reach."""
    L = len(trav["nbr_counts"])
    dev = feats.device
    h = feats
    for k, conv in enumerate(model.convs):
        l = L - 1 - k
        fr = torch.as_tensor(np.asarray(trav["frontier"][l]), device=dev).long()
        counts = torch.as_tensor(np.asarray(trav["nbr_counts"][l]), device=dev).long()
        flat = torch.as_tensor(np.asarray(trav["nbr_flat"][l]), device=dev).long()
        owner = torch.repeat_interleave(fr, counts)                 # destination of every entry of the stream
        keep = flat != owner                                        # drops the leading self entry and sampled self loops
        dst, src = owner[keep], flat[keep]
        z = h @ conv.fc.weight.t()
        zv = z.view(-1, conv.H, conv.D)
        el, er = (zv * conv.attn_l).sum(-1), (zv * conv.attn_r).sum(-1)
        score = torch.nn.functional.leaky_relu(el[src] + er[dst], conv.slope)            # [E, H]
        n_all = h.shape[0]
        m = torch.full((n_all, conv.H), -1e30, device=dev, dtype=h.dtype).scatter_reduce(
            0, dst[:, None].expand(-1, conv.H), score.detach(), "amax", include_self=True)
        p = torch.exp(score - m[dst])
        ssum = torch.zeros((n_all, conv.H), device=dev, dtype=h.dtype).index_add(0, dst, p)
        num = torch.zeros((n_all, conv.H, conv.D), device=dev, dtype=h.dtype).index_add(0, dst, p[:, :, None] * zv[src])
        out = (num / ssum.clamp_min(1e-30)[:, :, None]).reshape(n_all, -1) + conv.bias
        new = torch.zeros_like(out).index_copy(0, fr, out[fr])
        if k + 1 < len(model.convs):
            h = torch.nn.functional.elu(new)
        else:
            h = new.view(-1, model.heads, conv.D).mean(1)[:, :model.n_classes]
    return h


@pytest.mark.parametrize("by_source", [False, True])
@pytest.mark.parametrize("heads,hidden,B", [(8, 32, 256), (2, 12, 64), (3, 8, 700)])
def test_fused_local_gat_layer_matches_the_node_by_node_path(mods, heads, hidden, B, by_source, monkeypatch):
    """aggr.GatLayerLocal (a whole layer + ELU as one autograd node: epilogue kernels, one gradient buffer for z)
    against the same model through the separate autograd nodes (CSLICER_NO_LOCAL_FUSE): logits, every parameter
    gradient and the input gradient."""
    abi, aggr, sg = mods
    from cslicer import l0
    torch.manual_seed(heads)
    n, F0, classes = 6000, 20, 7
    indptr, indices = l0.synth_graph(n, 12.0, seed=heads)
    # by_source: the engine emits the slices by source for EVERY layer and the fused layer's backward writes the
    # gradient of z row by row (csl_gat_bwd_t_f32) instead of scattering it with atomics (csl_gat_bwd_f32)
    eng = abi.Engine(indptr, indices, n_parts=1, fanouts=(5, 4, 3), max_batch=B, mode=abi.MODE_GRAPH,
                     flags=(abi.FLAG_TRANSPOSE | abi.FLAG_TRANSPOSE_ALL) if by_source else 0)
    eng.submit_seeds([np.random.default_rng(3).permutation(n)[:B]])
    slices = sg.slices_of(eng)
    assert all(bool(slices[l][0].t_indptr.numel()) == by_source for l in range(3))
    monkeypatch.setattr(sg, "ROW_PAD", 512)          # both the padded and the unpadded GEMM operand occur
    model = sg.DistGATModel(F0, hidden, classes, heads=heads, n_layers=3).cuda()
    with torch.no_grad():
        for conv in model.convs:
            conv.bias.normal_(0, 0.1)
    x0 = torch.randn(slices[2][0].n_in, F0, device="cuda")
    w = torch.randn(B, classes, device="cuda")
    res = []
    for fused in (True, False):
        monkeypatch.setattr(sg, "_NO_LOCAL_FUSE", not fused)
        model.zero_grad()
        x = x0.clone().requires_grad_()
        out = model.forward_parts(slices, {0: x})[0]
        assert (type(out.grad_fn).__name__ != "GatLayerLocalBackward") or fused
        (out * w).sum().backward()
        res.append((out.detach().clone(), x.grad.clone(), [p.grad.clone() for p in model.parameters()]))
    torch.testing.assert_close(res[0][0], res[1][0], rtol=1e-5, atol=1e-5)
    scale = lambda t: max(1.0, float(t.abs().max()))                                                # noqa: E731
    torch.testing.assert_close(res[0][1], res[1][1], rtol=1e-4, atol=1e-4 * scale(res[1][1]))
    for a_, b_ in zip(res[0][2], res[1][2]):
        torch.testing.assert_close(a_, b_, rtol=1e-4, atol=1e-4 * scale(b_))
    eng.close()


def test_gat_config5_shape_eight_parts(mods):
    """BASELINE configs[4] shape: 3-layer GAT, 8 heads x 32, fanout 10/10/10, batch 1024, EIGHT parts (all in this
    process; products-like degrees on a 400k-node graph so that the dense reference fits), against the unsplit
    definition computed with torch index ops in fp32 on the same sampled graph.  ~10^6 sampled edges and
    frontiers of ~10^5 nodes.  The reference runs in float64, so what is measured is the fp32 error of the
    split-parallel path itself: outputs within 1e-4 relative (1e-5 holds for the single aggregation kernel, tested
    above); gradients, which are sums of ~10^5-10^6 signed fp32 terms, within 1e-3 of the largest entry of the
    tensor (plus 2e-3 relative)."""
    abi, aggr, sg = mods
    from cslicer import l0
    from oracle import oracle as orc
    torch.manual_seed(2)
    n, F0, hidden, classes, B, P, heads, fan = 400_000, 100, 32, 47, 1024, 8, 8, (10, 10, 10)
    indptr, indices = l0.synth_graph(n, 50.5, seed=0)
    seeds = np.random.default_rng(4).permutation(n)[:B]
    eng = abi.Engine(indptr, indices, n_parts=P, fanouts=fan, max_batch=B, mode=abi.MODE_GRAPH)
    eng.submit_seeds([seeds])
    slices = sg.slices_of(eng)
    L = len(fan)
    feats = torch.randn(n, F0, device="cuda")
    model = sg.DistGATModel(F0, hidden, classes, heads=heads, n_layers=L).cuda()
    with torch.no_grad():
        for conv in model.convs:
            conv.bias.normal_(0, 0.1)
    x = {g: feats[slices[L - 1][g].in_nodes.long()].clone().requires_grad_() for g in range(P)}
    out = model.forward_parts(slices, x)
    assert sum(out[g].shape[0] for g in range(P)) == B
    w = torch.randn(n, classes, device="cuda")
    seeds_t = torch.from_numpy(seeds).cuda()
    sum((out[g] * w[seeds_t[seeds_t % P == g]]).sum() for g in range(P)).backward()
    got_grads = [p.grad.clone() for p in model.parameters()]
    import copy
    model64 = copy.deepcopy(model).double()
    model64.zero_grad()
    trav = orc.Oracle(indptr, indices, n_parts=P, fanouts=fan).sample(seeds)
    fin = feats.double().requires_grad_()
    ref = _dense_gat_vectorised(model64, trav, fin)
    for g in range(P):
        own = seeds_t[seeds_t % P == g]
        torch.testing.assert_close(out[g].detach(), ref[own].detach().float(), rtol=1e-4, atol=1e-5)
    (ref[seeds_t] * w[seeds_t].double()).sum().backward()
    for (name, p_), gg in zip(model64.named_parameters(), got_grads):
        scale = float(p_.grad.abs().max())
        torch.testing.assert_close(gg, p_.grad.float(), rtol=2e-3, atol=1e-3 * max(scale, 1e-3),
                                   msg=lambda m_: "grad " + name + ": " + m_)
    for g in range(P):
        ids = slices[L - 1][g].in_nodes.long()
        want = fin.grad[ids].float()
        torch.testing.assert_close(x[g].grad, want, rtol=2e-3, atol=1e-3 * max(float(want.abs().max()), 1e-3))
    eng.close()


@pytest.mark.parametrize("H,D,n", [(8, 32, 5000), (4, 12, 333), (1, 4, 70), (2, 256, 900), (3, 48, 1)])
def test_gat_logits_kernel_matches_torch(mods, H, D, n):
    """aggr.GatLogits (csl_gat_logits_fwd_f32 / _bwd_f32) against (z.view(n, H, D) * a).sum(-1) in torch."""
    _, aggr, _ = mods
    torch.manual_seed(H * D + n)
    z0 = torch.randn(n, H * D, device="cuda")
    al0, ar0 = torch.randn(H, D, device="cuda"), torch.randn(H, D, device="cuda")
    gl, gr = torch.randn(n, H, device="cuda"), torch.randn(n, H, device="cuda")
    res = []
    for which in ("hip", "torch"):
        z, al, ar = (t.clone().requires_grad_() for t in (z0, al0, ar0))
        if which == "hip":
            el, er = aggr.GatLogits.apply(z, al, ar)
        else:
            zv = z.view(n, H, D)
            el, er = (zv * al).sum(-1), (zv * ar).sum(-1)
        ((el * gl).sum() + (er * gr).sum()).backward()
        res.append((el.detach(), er.detach(), z.grad, al.grad, ar.grad))
    for name, a_, b_ in zip(["el", "er", "grad z", "grad a_l", "grad a_r"], res[0], res[1]):
        torch.testing.assert_close(a_, b_, rtol=1e-4, atol=1e-4 * max(1.0, float(b_.abs().max())), msg=lambda m_: name + ": " + m_)

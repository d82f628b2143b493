"""Aggregation kernels (include/cslicer_aggr.h) and the split-parallel GraphSAGE built on them,
against plain torch fp32 references.  The reference's Python consumer needs DGL (absent here,
ModuleNotFoundError) and ships no goldens: these results are pinned by analytic torch references
only ("parity unpinned" by the reference).  Tolerance: 1e-5 (north_star: aggregation outputs
within 1e-5 fp32)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-5, atol=1e-5)


@pytest.fixture(scope="module")
def mods():
    from cslicer import _abi, aggr, splitgnn
    _abi.load()
    return _abi, aggr, splitgnn


def _rand_csr(n_rows, n_src, max_deg, rng):
    deg = rng.integers(0, max_deg + 1, size=n_rows)
    indptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.cumsum(deg, out=indptr[1:])
    indices = rng.integers(0, n_src, size=int(indptr[-1])).astype(np.int64)
    return indptr, indices


@pytest.mark.parametrize("H", [1, 3, 4, 32, 100, 128, 256, 300])
def test_spmm_sum_forward_backward(mods, H):
    _, aggr, _ = mods
    rng = np.random.default_rng(H)
    n_rows, n_src = 777, 500
    indptr, indices = _rand_csr(n_rows, n_src, 15, rng)
    x = torch.randn(n_src, H, dtype=torch.float32)
    ip, ix = torch.from_numpy(indptr).int(), torch.from_numpy(indices).int()
    # torch fp32 reference: dense accumulation in edge order
    ref = torch.zeros(n_rows, H)
    rows = torch.repeat_interleave(torch.arange(n_rows), torch.from_numpy(np.diff(indptr)))
    ref.index_add_(0, rows, x[ix.long()])
    xg = x.cuda().requires_grad_()
    out = aggr.SpmmSum.apply(xg, ip.cuda(), ix.cuda(), n_rows)
    torch.testing.assert_close(out.cpu(), ref, **TOL)
    g = torch.randn(n_rows, H)
    out.backward(g.cuda())
    gref = torch.zeros(n_src, H)
    gref.index_add_(0, ix.long(), g[rows])
    torch.testing.assert_close(xg.grad.cpu(), gref, rtol=1e-5, atol=2e-5)
    # row subset (boundary rows first)
    sel = torch.from_numpy(rng.choice(n_rows, size=100, replace=False).astype(np.int64))
    part = aggr.spmm_sum(ip.cuda(), ix.cuda(), x.cuda(), n_rows, rows=sel.int().cuda())
    torch.testing.assert_close(part.cpu()[sel], ref[sel], **TOL)
    mask = torch.ones(n_rows, dtype=torch.bool)
    mask[sel] = False
    assert float(part.cpu()[mask].abs().sum()) == 0.0


@pytest.mark.parametrize("H", [5, 64, 128])
def test_gather_scatter_div_rows(mods, H):
    _, aggr, _ = mods
    rng = np.random.default_rng(7 + H)
    src = torch.randn(400, H)
    idx = torch.from_numpy(rng.choice(400, size=150, replace=False).astype(np.int64))
    idx_m = idx.clone()
    idx_m[::7] = -1
    got = aggr.gather_rows(src.cuda(), idx_m.int().cuda()).cpu()
    ref = src[idx_m.clamp(min=0)] * (idx_m >= 0).unsqueeze(1)
    torch.testing.assert_close(got, ref, rtol=0, atol=0)
    dst = torch.randn(400, H)
    add = torch.randn(150, H)
    got = aggr.scatter_add_rows_(dst.clone().cuda(), idx.int().cuda(), add.cuda()).cpu()
    ref = dst.clone()
    ref[idx] += add
    torch.testing.assert_close(got, ref, **TOL)
    deg = torch.from_numpy(rng.integers(0, 20, size=400).astype(np.int64))
    got = aggr.div_rows_(dst.clone().cuda(), deg.int().cuda()).cpu()
    torch.testing.assert_close(got, dst / deg.clamp(min=1).unsqueeze(1), **TOL)
    # autograd of the wrappers
    s = src.clone().cuda().requires_grad_()
    aggr.GatherRows.apply(s, idx.int().cuda()).sum().backward()
    gref = torch.zeros(400, H)
    gref[idx] = 1.0
    torch.testing.assert_close(s.grad.cpu(), gref, rtol=0, atol=0)


def test_attention_gather_matches_u_mul_v_sum(mods):
    _, aggr, _ = mods
    rng = np.random.default_rng(5)
    n_rows, n_src, H = 300, 200, 24
    indptr, indices = _rand_csr(n_rows, n_src, 9, rng)
    u, v = torch.randn(n_src, H), torch.randn(n_rows, H)
    ref = torch.zeros(n_rows, H)
    for r in range(n_rows):
        for e in range(indptr[r], indptr[r + 1]):
            ref[r] += u[indices[e]] * v[r]
    got = aggr.attention_gather(torch.from_numpy(indptr).int().cuda(), torch.from_numpy(indices).int().cuda(),
                                u.cuda(), v.cuda(), n_rows)
    torch.testing.assert_close(got.cpu(), ref, **TOL)


def _dense_reference(model, indptr, indices, trav, feats, P):
    """Unsplit torch fp32 GraphSAGE on the same sampled computation graph (CPU)."""
    L = len(trav["nbr_counts"])
    h = feats.clone()
    for k, conv in enumerate(model.convs):
        l = L - 1 - k
        fr = trav["frontier"][l]
        counts, flat = trav["nbr_counts"][l], trav["nbr_flat"][l]
        new = torch.zeros(h.shape[0], conv.fc.out_features)
        pos = 0
        for i, nd1 in enumerate(fr):
            nb = flat[pos + 1:pos + counts[i]]
            pos += counts[i]
            nb = nb[nb != nd1]
            neigh = h[nb].sum(0) / max(len(nb), 1) if len(nb) else torch.zeros(h.shape[1])
            new[nd1] = conv.fc(torch.cat([h[nd1], neigh]))
        h = torch.relu(new) if k + 1 < len(model.convs) else new
    return h


@pytest.mark.parametrize("P,fan", [(4, (10, 10, 10)), (2, (15, 10, 5)), (1, (5, 5))])
def test_split_parallel_sage_matches_dense_reference(mods, P, fan):
    abi, aggr, sg = mods
    from cslicer import l0
    from oracle import oracle as orc
    torch.manual_seed(0)
    n, F0, hidden, classes, B = 3000, 20, 16, 7, 64
    indptr, indices = l0.synth_graph(n, 12.0, seed=5)
    seeds = np.random.default_rng(2).permutation(n)[:B]
    eng = abi.Engine(indptr, indices, n_parts=P, fanouts=fan, max_batch=B, mode=abi.MODE_GRAPH)
    eng.submit_seeds([seeds])
    slices = sg.slices_of(eng)
    L = len(fan)
    feats = torch.randn(n, F0)
    model = sg.DistSAGEModel(F0, hidden, classes, n_layers=L)
    # each part reads only the features of nodes it owns
    x = {g: feats[slices[L - 1][g].in_nodes.cpu().long()].cuda().requires_grad_() for g in range(P)}
    for g in range(P):
        assert bool((slices[L - 1][g].in_nodes % P == g).all())
    gm = sg.DistSAGEModel(F0, hidden, classes, n_layers=L).cuda()
    gm.load_state_dict(model.state_dict())
    out = gm.forward_parts(slices, x)
    trav = orc.Oracle(indptr, indices, n_parts=P, fanouts=fan).sample(seeds)
    fin = feats.clone().requires_grad_()
    ref = _dense_reference(model, indptr, indices, trav, fin, P)
    for g in range(P):
        own = seeds[seeds % P == g]
        torch.testing.assert_close(out[g].detach().cpu(), ref[own].detach(), **TOL)
    # backward: same loss on both sides
    w = torch.randn(n, classes)
    loss_ref = (ref[seeds] * w[seeds]).sum()
    loss_ref.backward()
    loss = sum((out[g] * w[seeds[seeds % P == g]].cuda()).sum() for g in range(P))
    loss.backward()
    for (na, pa), (nb, pb) in zip(gm.named_parameters(), model.named_parameters()):
        torch.testing.assert_close(pa.grad.cpu(), pb.grad, rtol=1e-4, atol=1e-5, msg="grad " + na)
    for g in range(P):
        ids = slices[L - 1][g].in_nodes.cpu().long()
        torch.testing.assert_close(x[g].grad.cpu(), fin.grad[ids], rtol=1e-4, atol=1e-5)
    eng.close()


def test_split_k_linear_matches_torch():
    """splitgnn._SplitKLinear (library GEMMs + slab-wise weight gradient) against torch.nn.functional.linear:
    forward bit-identical (same addmm), gradients within 1e-4 relative (fp32, different summation order)."""
    import torch
    from cslicer import splitgnn
    torch.manual_seed(3)
    m, fin, fout = 2 * splitgnn.ROW_PAD, 200, 64
    x = torch.rand((m, fin), device="cuda", requires_grad=True)
    w = (torch.rand((fout, fin), device="cuda") - 0.5).requires_grad_(True)
    b = torch.rand((fout,), device="cuda", requires_grad=True)
    gy = torch.rand((m, fout), device="cuda")
    y0 = torch.nn.functional.linear(x, w, b)
    g0 = torch.autograd.grad(y0, (x, w, b), gy)
    y1 = splitgnn._SplitKLinear.apply(x, w, b)
    g1 = torch.autograd.grad(y1, (x, w, b), gy)
    assert torch.allclose(y0, y1, rtol=1e-6, atol=1e-6)
    for a_, b_ in zip(g0, g1):
        assert torch.allclose(a_, b_, rtol=1e-4, atol=1e-3 * float(a_.abs().max()) * 1e-1), float((a_ - b_).abs().max())


def test_finish_pads_rows_without_changing_results():
    """DistSageConv.finish pads the GEMM rows to a multiple of ROW_PAD above ROW_PAD rows: same output rows
    as the unpadded Linear."""
    import torch
    from cslicer import splitgnn
    torch.manual_seed(4)
    conv = splitgnn.DistSageConv(32, 16).cuda()
    m = splitgnn.ROW_PAD + 37

    class S(object):
        pass
    sl = S()
    sl.owned_out_nodes = torch.arange(m, dtype=torch.int32, device="cuda")
    sl.owned_degree = torch.randint(1, 9, (m,), dtype=torch.int32, device="cuda")
    sl.self_ids_in = torch.randint(0, m, (m,), dtype=torch.int32, device="cuda")
    agg = torch.rand((m, 32), device="cuda")
    x = torch.rand((m, 32), device="cuda")
    got = conv.finish(sl, agg, x)
    want = conv.fc(torch.cat([x[sl.self_ids_in.long()], agg / sl.owned_degree[:, None].float()], dim=1))
    assert got.shape == want.shape
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("m,fin,relu", [(500, 32, True), (500, 30, False), (4096 + 37, 32, True), (2 * 4096, 100, False)])
def test_fused_finish_matches_the_op_chain(m, fin, relu):
    """splitgnn._SageFinish (one autograd node: gathers into the GEMM operand, GEMM, ReLU) against the chain
    finish() + torch.relu built from the separate ops: forward 1e-5, gradients 1e-4 (slab-wise weight gradient)."""
    import torch
    from cslicer import splitgnn
    torch.manual_seed(m + fin)
    conv = splitgnn.DistSageConv(fin, 24).cuda()
    n_x, n_agg = m + 50, m + 20

    class S(object):
        pass
    sl = S()
    sl.owned_out_nodes = torch.randperm(n_agg, device="cuda")[:m].int()
    sl.owned_degree = torch.randint(0, 9, (m,), dtype=torch.int32, device="cuda")
    sl.self_ids_in = torch.randperm(n_x, device="cuda")[:m].int()
    sl.self_ids_in[::7] = -1                      # nodes never sampled as a source: zero self row
    agg0 = torch.rand((n_agg, fin), device="cuda")
    x0 = torch.rand((n_x, fin), device="cuda")
    gy = torch.rand((m, 24), device="cuda")
    outs = []
    for fused in (False, True):
        agg, x = agg0.clone().requires_grad_(True), x0.clone().requires_grad_(True)
        conv.zero_grad()
        if fused:
            y = conv.finish_fused(sl, agg, x, relu)
        else:
            y = conv.finish(sl, agg, x)
            y = torch.relu(y) if relu else y
        y.backward(gy)
        outs.append((y.detach(), agg.grad, x.grad, conv.fc.weight.grad.clone(), conv.fc.bias.grad.clone()))
    names = ["y", "grad agg", "grad x", "grad W", "grad b"]
    for nme, a_, b_ in zip(names, outs[0], outs[1]):
        tol = 1e-5 if nme == "y" else 1e-4
        assert torch.allclose(a_, b_, rtol=tol, atol=tol * max(1.0, float(a_.abs().max()))), (nme, float((a_ - b_).abs().max()))


def test_empty_lists_are_no_ops(mods):
    """A part without boundary rows hands empty index lists (null device pointers) to the kernels' C entry
    points: they return success and touch nothing (found with the RCCL world-of-one test)."""
    _, aggr, _ = mods
    x = torch.rand((10, 8), device="cuda")
    empty = torch.zeros(0, dtype=torch.int32, device="cuda")
    assert aggr.gather_rows(x, empty).shape == (0, 8)
    y = x.clone()
    aggr.scatter_add_rows_(y, empty, torch.zeros((0, 8), device="cuda"))
    assert torch.equal(x, y)
    ip = torch.zeros(11, dtype=torch.int32, device="cuda")
    out = torch.full((10, 8), 7.0, device="cuda")
    aggr.spmm_sum(ip, empty, x, 10, rows=empty, out=out)
    assert bool((out == 7.0).all())
    g = aggr.spmm_sum_bwd(ip, empty, torch.zeros((0, 8), device="cuda"), 10, rows=empty, compact=True)
    assert g.shape == (10, 8) and float(g.abs().sum()) == 0.0
    aggr.div_rows_(torch.zeros((0, 8), device="cuda"), empty)


@pytest.mark.parametrize("H,use_map,relu_in", [(4, False, False), (100, True, False), (256, False, True), (32, True, True)])
def test_sage_cat_forward_backward(mods, H, use_map, relu_in):
    """csl_sage_cat_f32 / csl_sage_cat_bwd_f32 (self gather + CSR mean straight into the GEMM operand, optional
    indirection through a row map, optional ReLU on load) against the torch op chain; 1e-5 forward, 1e-4 backward
    (fp32 atomics: summation order differs)."""
    _, aggr, _ = mods
    rng = np.random.default_rng(H)
    n, n_in, n_tab, n_pad = 700, 900, 1500, 1024
    indptr, indices = _rand_csr(n, n_in, 12, rng)
    self_ids = rng.permutation(n_in)[:n].astype(np.int32)
    self_ids[::11] = -1
    rowmap = rng.permutation(n_tab)[:n_in].astype(np.int32) if use_map else None
    x = torch.randn(n_tab if use_map else n_in, H)
    ip, ix, si = (torch.from_numpy(a).int().cuda() for a in (indptr, indices, self_ids))
    rm = torch.from_numpy(rowmap).cuda() if use_map else None
    xr = x.clone().requires_grad_()
    h = torch.relu(xr) if relu_in else xr
    src = h[torch.from_numpy(rowmap).long()] if use_map else h
    rows = torch.repeat_interleave(torch.arange(n), torch.from_numpy(np.diff(indptr)))
    summed = torch.zeros(n, H).index_add(0, rows, src[torch.from_numpy(indices).long()])
    deg = torch.from_numpy(np.diff(indptr)).clamp_min(1).float()
    selfrows = torch.where(torch.from_numpy(self_ids)[:, None] >= 0, src[torch.from_numpy(self_ids).clamp_min(0).long()],
                           torch.zeros(1, H))
    ref = torch.cat([selfrows, summed / deg[:, None]], dim=1)
    cat = aggr.sage_cat(x.cuda(), si, n, n_pad, indptr=ip, indices=ix, rowmap=rm, relu_in=relu_in)
    assert cat.shape == (n_pad, 2 * H) and float(cat[n:].abs().sum()) == 0.0
    torch.testing.assert_close(cat[:n].cpu(), ref.detach(), **TOL)
    if not use_map and not relu_in:
        g = torch.randn(n_pad, 2 * H)
        ref.backward(g[:n])
        gx = aggr.sage_cat_bwd(ip, ix, si, g.cuda(), n, n_in)
        torch.testing.assert_close(gx.cpu(), xr.grad, rtol=1e-4, atol=2e-5)
    # the merged-sums form (one part of several): agg[owned] / deg next to the self rows
    owned = torch.from_numpy(rng.permutation(n + 30)[:n].astype(np.int32)).cuda()
    degs = torch.from_numpy(rng.integers(0, 9, size=n).astype(np.int32)).cuda()
    agg = torch.randn(n + 30, H, device="cuda")
    cat2 = aggr.sage_cat(x.cuda(), si, n, n, owned=owned, deg=degs, agg=agg, rowmap=rm, relu_in=relu_in)
    torch.testing.assert_close(cat2[:, :H].cpu(), selfrows.detach(), **TOL)
    torch.testing.assert_close(cat2[:, H:], agg[owned.long()] / degs.clamp_min(1)[:, None].float(), **TOL)


@pytest.mark.parametrize("H,n,n_pad,masked", [(256, 5000, 8192, True), (47, 1024, 1024, False), (3, 130, 200, True),
                                              (100, 77, 77, True)])
def test_relu_bwd_colsum(mods, H, n, n_pad, masked):
    _, aggr, _ = mods
    torch.manual_seed(H)
    g = torch.randn(n, H, device="cuda")
    y = torch.randn(n_pad, H, device="cuda") if masked else None
    out, cs = aggr.relu_bwd_colsum(g, y, n, n_pad)
    want = g * (y[:n] > 0) if masked else g
    assert out.shape == (n_pad, H) and float(out[n:].abs().sum()) == 0.0
    assert torch.equal(out[:n], want)
    torch.testing.assert_close(cs, want.double().sum(0).float(), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("C_,use_map", [(47, False), (7, True), (172, False)])
def test_softmax_ce_matches_torch(mods, C_, use_map):
    _, aggr, _ = mods
    torch.manual_seed(C_)
    n, N = 1000, 5000
    logits = (4 * torch.randn(n, C_, device="cuda")).requires_grad_()
    ids = torch.randperm(N, device="cuda")[:n].int()
    labels = torch.randint(0, C_, (N,), device="cuda")
    rowmap = torch.randperm(N, device="cuda").int() if use_map else None
    lab = labels[rowmap[ids.long()].long()] if use_map else labels[ids.long()]
    ref_in = logits.detach().clone().requires_grad_()
    ref = torch.nn.functional.cross_entropy(ref_in, lab, reduction="sum") / n
    (ref * 3.0).backward()
    loss = aggr.SoftmaxCE.apply(logits, ids, labels, 1.0 / n, rowmap)
    (loss * 3.0).backward()
    torch.testing.assert_close(loss.detach(), ref.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(logits.grad, ref_in.grad, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("by_source,direct_gemms", [(False, False), (True, False), (True, True)])
def test_fused_local_model_reads_the_feature_table(mods, by_source, direct_gemms, monkeypatch):
    """DistSAGEModel.forward_local (the deepest layer indexing the resident feature table through the slice's
    in_nodes) == forward_parts on the gathered features, forward and weight gradients.  by_source: the engine emits
    the slices by source (FLAG_TRANSPOSE) and the model is ONE autograd node whose input gradients are gathered
    (csl_sage_cat_bwd_t_f32); otherwise one node per layer with the atomic scatter."""
    abi, aggr, sg = mods
    from cslicer import l0
    monkeypatch.setattr(sg, "_DIRECT_GEMMS", direct_gemms)   # csl_gemm_f32 instead of torch's GEMMs
    torch.manual_seed(1)
    n, F0, hidden, classes, B = 4000, 24, 32, 5, 128
    indptr, indices = l0.synth_graph(n, 15.0, seed=9)
    eng = abi.Engine(indptr, indices, n_parts=1, fanouts=(6, 5, 4), max_batch=B, mode=abi.MODE_GRAPH,
                     flags=abi.FLAG_TRANSPOSE if by_source else 0)
    eng.submit_seeds([np.random.default_rng(3).permutation(n)[:B]])
    slices = sg.slices_of(eng)
    assert bool(slices[0][0].t_indptr.numel()) == by_source and slices[2][0].t_indptr.numel() == 0
    feats = torch.randn(n, F0, device="cuda")
    model = sg.DistSAGEModel(F0, hidden, classes, n_layers=3).cuda()
    w = torch.randn(B, classes, device="cuda")
    grads = []
    for local in (True, False):
        model.zero_grad()
        if local:
            out = model.forward_local(slices, feats)
            assert (type(out.grad_fn).__name__ == "_SageModelLocalBackward") == by_source
        else:
            out = model.forward_parts(slices, {0: feats[slices[2][0].in_nodes.long()]})[0]
        (out * w).sum().backward()
        grads.append((out.detach().clone(), [p.grad.clone() for p in model.parameters()]))
    torch.testing.assert_close(grads[0][0], grads[1][0], **TOL)
    for a_, b_ in zip(grads[0][1], grads[1][1]):
        torch.testing.assert_close(a_, b_, rtol=1e-4, atol=1e-5)
    eng.close()


@pytest.mark.parametrize("B,row_pad", [(128, 64), (128, 0), (700, 512)])
def test_native_step_matches_the_autograd_path(mods, B, row_pad):
    """csl_sage_fwd_bwd_f32 (forward + loss + backward as one native call, direct hipBLASLt GEMMs) against the same
    model through torch autograd (_SageModelLocal + SoftmaxCE, torch GEMMs): loss and every parameter gradient."""
    abi, aggr, sg = mods
    from cslicer import l0
    torch.manual_seed(2)
    n, F0, hidden, classes = 5000, 24, 32, 7
    indptr, indices = l0.synth_graph(n, 15.0, seed=B)
    eng = abi.Engine(indptr, indices, n_parts=1, fanouts=(6, 5, 4), max_batch=B, mode=abi.MODE_GRAPH,
                     flags=abi.FLAG_TRANSPOSE)
    eng.submit_seeds([np.random.default_rng(3).permutation(n)[:B]])
    slices = sg.slices_of(eng)
    feats = torch.randn(n, F0, device="cuda")
    labels = torch.randint(0, classes, (n,), device="cuda")
    model = sg.DistSAGEModel(F0, hidden, classes, n_layers=3).cuda()
    out = model.forward_local(slices, feats)
    loss = aggr.SoftmaxCE.apply(out, slices[0][0].out_nodes, labels, 1.0 / B)
    loss.backward()
    want = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    step = aggr.SageStep(model, row_pad, 4)
    got_loss = torch.zeros(1, device="cuda")
    if B == 128 and row_pad == 0:
        # slices of an engine WITHOUT the slices by source are refused, not read as garbage
        e2 = abi.Engine(indptr, indices, n_parts=1, fanouts=(6, 5, 4), max_batch=B, mode=abi.MODE_GRAPH)
        e2.submit_seeds([np.random.default_rng(3).permutation(n)[:B]])
        s2 = sg.slices_of(e2)
        with pytest.raises(ValueError):
            step([s2[2][0], s2[1][0], s2[0][0]], feats, labels, 1.0 / B, got_loss)
        e2.close()
    for _ in range(2):   # the second call reuses workspace and GEMM plans
        step([slices[2][0], slices[1][0], slices[0][0]], feats, labels, 1.0 / B, got_loss)
        torch.testing.assert_close(got_loss[0], loss.detach(), rtol=1e-5, atol=1e-6)
        assert float((step.grads - want).abs().max()) <= 1e-4 * float(want.abs().max())
    eng.close()


def _hub_graph(n=20000, deg=10, seed=4):
    """a graph whose rows all point at three hubs"""
    rng = np.random.default_rng(seed)
    nb = rng.integers(0, n, size=(n, deg))
    nb[:, :3] = np.array([5, 9, 17])
    nb[[5, 9, 17]] = rng.integers(100, n, size=(3, deg))
    indptr = np.arange(n + 1, dtype=np.int64) * deg
    indices = np.sort(nb, axis=1).reshape(-1).astype(np.int64)
    return indptr, indices, rng


def test_native_step_with_hub_nodes_gathers_their_lists_cooperatively(mods, monkeypatch):
    """A graph whose rows all point at three hubs: the hubs' lists in the slices by source are thousands of entries
    long (t_max_len > CSL_T_SORTED_MAX).  One wave walking such a list is a serial crawl, so those rows are summed by a
    workgroup per segment of entries (csl_sage_cat_bwd_t_hub_f32) -- the layer keeps the gather form, no atomic scatter
    of the whole layer; same loss and gradients as the per-layer autograd path with the atomic scatter on the same
    slices (src/gnn/sage.cu:20-28 is that scatter in the reference)."""
    abi, aggr, sg = mods
    torch.manual_seed(4)
    n, F0, hidden, classes, B = 20000, 16, 32, 5, 512
    indptr, indices, rng = _hub_graph(n)
    eng = abi.Engine(indptr, indices, n_parts=1, fanouts=(6, 5, 4), max_batch=B, mode=abi.MODE_GRAPH,
                     flags=abi.FLAG_TRANSPOSE)
    eng.submit_seeds([rng.permutation(n)[:B]])
    slices = sg.slices_of(eng)
    assert max(slices[l][0].t_max_len for l in range(2)) > abi.T_SORTED_MAX
    feats = torch.randn(n, F0, device="cuda")
    labels = torch.randint(0, classes, (n,), device="cuda")
    model = sg.DistSAGEModel(F0, hidden, classes, n_layers=3).cuda()
    # reference: one autograd node per layer, input gradients scattered with atomics (no slice by source involved)
    monkeypatch.setattr(sg, "_NO_LOCAL_FUSE", False, raising=False)
    eng2 = abi.Engine(indptr, indices, n_parts=1, fanouts=(6, 5, 4), max_batch=B, mode=abi.MODE_GRAPH)
    eng2.submit_seeds([_hub_graph(n)[2].permutation(n)[:B]])
    slices2 = sg.slices_of(eng2)
    out2 = model.forward_local(slices2, feats)
    assert type(out2.grad_fn).__name__ != "_SageModelLocalBackward"
    loss2 = aggr.SoftmaxCE.apply(out2, slices2[0][0].out_nodes, labels, 1.0 / B)
    loss2.backward()
    want = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    # the one-node autograd path over the slices by source (hub rows cooperatively)
    model.zero_grad()
    out = model.forward_local(slices, feats)
    assert type(out.grad_fn).__name__ == "_SageModelLocalBackward"
    loss = aggr.SoftmaxCE.apply(out, slices[0][0].out_nodes, labels, 1.0 / B)
    loss.backward()
    got_py = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    torch.testing.assert_close(loss.detach(), loss2.detach(), rtol=1e-5, atol=1e-6)
    assert float((got_py - want).abs().max()) <= 1e-4 * float(want.abs().max())
    # the native step
    step = aggr.SageStep(model, 256, 4)
    got_loss = torch.zeros(1, device="cuda")
    step([slices[2][0], slices[1][0], slices[0][0]], feats, labels, 1.0 / B, got_loss)
    torch.testing.assert_close(got_loss[0], loss2.detach(), rtol=1e-5, atol=1e-6)
    assert float((step.grads - want).abs().max()) <= 1e-4 * float(want.abs().max())
    eng.close()
    eng2.close()


@pytest.mark.parametrize("H,masked", [(256, True), (64, False), (100, True)])
def test_hub_lists_by_source_match_the_atomic_scatter(mods, H, masked):
    """csl_sage_cat_bwd_t_hub_f32 on slices with hub lists (several segments per hub, hubs next to short lists) ==
    csl_sage_cat_bwd_f32 (atomics) + csl_relu_bwd_colsum_f32."""
    abi, aggr, sg = mods
    torch.manual_seed(H)
    n, B = 20000, 1024
    indptr, indices, rng = _hub_graph(n)
    eng = abi.Engine(indptr, indices, n_parts=1, fanouts=(8, 6), max_batch=B, mode=abi.MODE_GRAPH, flags=abi.FLAG_TRANSPOSE)
    eng.submit_seeds([rng.permutation(n)[:B]])
    sl = sg.slices_of(eng)[0][0]
    assert sl.t_max_len > 4 * abi.T_SORTED_MAX
    n_pad = sl.n_in + 91
    gcat = torch.randn(sl.n_out, 2 * H, device="cuda")
    y = torch.randn(n_pad, H, device="cuda") if masked else None
    out, cs = aggr.sage_cat_bwd_t(sl.t_indptr, sl.t_indices, sl.indptr, gcat, y, sl.n_in, n_pad, hub=True)
    gx = aggr.sage_cat_bwd(sl.indptr, sl.indices, sl.self_ids_in, gcat, sl.n_out, sl.n_in)
    want, wcs = aggr.relu_bwd_colsum(gx, y, sl.n_in, n_pad)
    # (hub rows sum thousands of N(0, 1) terms: compare relative to the row's magnitude)
    scale = want.abs().max(dim=1, keepdim=True).values.clamp(min=1.0)
    assert float(((out - want).abs() / scale).max()) <= 1e-5
    torch.testing.assert_close(cs, wcs, rtol=1e-4, atol=2e-3 * float(want.abs().max()))
    # a slice WITHOUT hubs through the hub entry point: identical to the plain kernel
    from cslicer import l0
    ip2, ix2 = l0.synth_graph(6000, 20.0, seed=2)
    e2 = abi.Engine(ip2, ix2, n_parts=1, fanouts=(8, 6), max_batch=256, mode=abi.MODE_GRAPH, flags=abi.FLAG_TRANSPOSE)
    e2.submit_seeds([np.random.default_rng(5).permutation(6000)[:256]])
    s2 = sg.slices_of(e2)[0][0]
    g2 = torch.randn(s2.n_out, 2 * H, device="cuda")
    a_, ca = aggr.sage_cat_bwd_t(s2.t_indptr, s2.t_indices, s2.indptr, g2, None, s2.n_in, s2.n_in + 5, hub=True)
    b_, cb = aggr.sage_cat_bwd_t(s2.t_indptr, s2.t_indices, s2.indptr, g2, None, s2.n_in, s2.n_in + 5)
    assert torch.equal(a_, b_)
    torch.testing.assert_close(ca, cb, rtol=1e-6, atol=1e-6)
    eng.close()
    e2.close()


@pytest.mark.parametrize("H,masked", [(256, True), (32, False), (100, True)])
def test_sage_cat_bwd_by_source_matches_atomic_scatter(mods, H, masked):
    """csl_sage_cat_bwd_t_f32 over the engine's slice by source == csl_sage_cat_bwd_f32 (atomics) followed by
    csl_relu_bwd_colsum_f32, and == a float64 dense reference."""
    abi, aggr, sg = mods
    from cslicer import l0
    torch.manual_seed(H)
    n, B = 6000, 256
    indptr, indices = l0.synth_graph(n, 20.0, seed=2)
    eng = abi.Engine(indptr, indices, n_parts=1, fanouts=(8, 6), max_batch=B, mode=abi.MODE_GRAPH,
                     flags=abi.FLAG_TRANSPOSE)
    eng.submit_seeds([np.random.default_rng(5).permutation(n)[:B]])
    sl = sg.slices_of(eng)[0][0]
    n_pad = sl.n_in + 37
    gcat = torch.randn(sl.n_out, 2 * H, device="cuda")
    y = torch.randn(n_pad, H, device="cuda") if masked else None
    out, cs = aggr.sage_cat_bwd_t(sl.t_indptr, sl.t_indices, sl.indptr, gcat, y, sl.n_in, n_pad)
    gx = aggr.sage_cat_bwd(sl.indptr, sl.indices, sl.self_ids_in, gcat, sl.n_out, sl.n_in)
    want, wcs = aggr.relu_bwd_colsum(gx, y, sl.n_in, n_pad)
    torch.testing.assert_close(out, want, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(cs, wcs, rtol=1e-4, atol=1e-3)
    # float64 reference
    ip = sl.indptr.long().cpu().numpy()
    deg = np.maximum(np.diff(ip), 1)
    rows = np.repeat(np.arange(sl.n_out), np.diff(ip))
    ref = torch.zeros(n_pad, H, dtype=torch.float64, device="cuda")
    g64 = gcat.double()
    ref.index_add_(0, sl.indices.long(), g64[torch.from_numpy(rows).cuda(), H:] /
                   torch.from_numpy(deg[rows]).cuda()[:, None].double())
    ref.index_add_(0, sl.self_ids_in.long(), g64[:, :H])
    if masked:
        ref = ref * (y > 0)
    torch.testing.assert_close(out.double(), ref, rtol=1e-5, atol=1e-5)
    assert float(out[sl.n_in:].abs().sum()) == 0.0
    out2, _ = aggr.sage_cat_bwd_t(sl.t_indptr, sl.t_indices, sl.indptr, gcat, y, sl.n_in, n_pad)
    assert torch.equal(out, out2)   # deterministic: every list has a fixed order
    eng.close()


@pytest.mark.parametrize("m,n,k,ta,tb,bias,relu", [
    (8192, 256, 200, False, True, True, True),      # Linear forward with bias + ReLU epilogue
    (4096, 47, 512, False, True, True, False),      # last layer: bias only
    (8192, 512, 256, False, False, False, False),   # input gradient
    (256, 200, 1000, True, False, False, False),    # weight gradient, un-slabbed
    (33, 7, 5, False, False, False, True),          # odd sizes, ReLU alone
])
def test_direct_gemm_matches_float64(mods, m, n, k, ta, tb, bias, relu):
    """csl_gemm_f32 (hipBLASLt, row-major wrapper, timed plan) against a float64 product: 1e-4 of the largest entry
    (fp32 accumulation over k <= 1000 terms)."""
    _, aggr, _ = mods
    torch.manual_seed(m + n)
    a = torch.randn((k, m) if ta else (m, k), device="cuda")
    b = torch.randn((n, k) if tb else (k, n), device="cuda")
    bv = torch.randn(n, device="cuda") if bias else None
    for _ in range(2):   # the second call takes the cached plan
        got = aggr.gemm(a, b, transa=ta, transb=tb, bias=bv, relu=relu)
        want = (a.double().t() if ta else a.double()) @ (b.double().t() if tb else b.double())
        if bias:
            want = want + bv.double()
        if relu:
            want = want.relu()
        assert got.shape == (m, n)
        assert float((got.double() - want).abs().max()) <= 1e-4 * float(want.abs().max())
    # strided rows: a column block of a wider matrix as the left operand
    if not ta:
        wide = torch.randn(m, k + 8, device="cuda")
        got = aggr.gemm(wide[:, 4:4 + k], b, transb=tb)
        want = wide[:, 4:4 + k].double() @ (b.double().t() if tb else b.double())
        assert float((got.double() - want).abs().max()) <= 1e-4 * float(want.abs().max())


def test_weight_grad_slabs_matches_float64(mods):
    _, aggr, _ = mods
    torch.manual_seed(3)
    rows, out_f, in_f = 32 * 640, 256, 200
    gy, x = torch.randn(rows, out_f, device="cuda"), torch.randn(rows, in_f, device="cuda")
    got = aggr.weight_grad_slabs(gy, x, 32)
    want = gy.double().t() @ x.double()
    assert got.shape == (out_f, in_f)
    assert float((got.double() - want).abs().max()) <= 1e-4 * float(want.abs().max())


def test_adam_matches_torch(mods):
    """aggr.Adam (csl_adam_f32, one launch for all tensors) against torch.optim.Adam over 20 steps."""
    _, aggr, _ = mods
    torch.manual_seed(5)
    shapes = [(256, 200), (256,), (47, 512), (47,), (3,), (1025, 3)]
    pa = [torch.randn(s, device="cuda").requires_grad_() for s in shapes]
    pb = [p.detach().clone().requires_grad_() for p in pa]
    oa, ob = aggr.Adam(pa, lr=3e-3), torch.optim.Adam(pb, lr=3e-3)
    for _ in range(20):
        for a_, b_ in zip(pa, pb):
            g = torch.randn_like(a_)
            a_.grad, b_.grad = g.clone(), g.clone()
        oa.step()
        ob.step()
    for a_, b_ in zip(pa, pb):
        torch.testing.assert_close(a_.detach(), b_.detach(), rtol=1e-5, atol=1e-6)


def test_scatter_add_rows_atomic_with_repeated_rows(mods):
    _, aggr, _ = mods
    torch.manual_seed(9)
    dst0 = torch.randn(300, 100, device="cuda")
    idx = torch.randint(0, 300, (2000,), device="cuda").int()
    idx[::17] = -1
    src = torch.randn(2000, 100, device="cuda")
    got = aggr.scatter_add_rows_atomic_(dst0.clone(), idx, src)
    keep = idx >= 0
    want = dst0.clone().index_add_(0, idx[keep].long(), src[keep])
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)

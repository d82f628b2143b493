"""Split-parallel GraphSAGE over the slices (CSL_MODE_GRAPH), torch + HIP kernels, no DGL.

Counterpart of the reference's Python consumer:
  DistSageConv.forward          python/layers/dist_sageconv.py:42-84
  DistSAGEModel                 python/models/factory.py:7-56
  BipartiteGraph ops            python/data/bipartite.py:61-99
  DistGraphConv (4x4 merge)     python/batch_slice_multi_gpu.py:40-76

Part g (one GPU) owns the nodes with workload g and their features.  Per GNN layer, on part g:

  agg   = SUM over the slice's CSR of source features        (local: sources are owned by g)
  send  = agg[from_ids[g][p]]  -> part p                     (partials for nodes owned by p)
  agg[to_ids[g][p]] += recv from p                           (boundary merge)
  h     = Linear(concat(x[self_ids_in], agg[owned] / degree))

The reference averages per-GPU means (`(local+remote)/2`, bipartite.py:98), which is not the
mean over the sampled neighbours; here partial SUMS are merged and divided once by the true
sampled degree.

Layer chaining needs no index translation: the rows a layer produces for part g (owned frontier
nodes, frontier order) are exactly, in order, the in_nodes of part g one hop closer to the seeds.

Every rank slices the SAME minibatch with its own engine (the slicer is deterministic and ~40x
faster than the training step): topology is recomputed, never communicated; only boundary
partial sums cross xGMI (one all-to-all per layer, on a side stream, overlapped with the rows
that do not leave the GPU).
"""
import torch
import torch.nn as nn

from . import _abi, _roctx, aggr


class _DevArray(object):
    """Zero-copy view of engine memory for torch.as_tensor (__cuda_array_interface__)."""

    def __init__(self, ptr, n, typestr="<i4"):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def _view_i32(ptr, n, device):
    if n == 0:
        return torch.zeros(0, dtype=torch.int32, device=device)
    return torch.as_tensor(_DevArray(ptr, n), device=device)


_SLICE_LISTS = {"in_nodes": _abi.IN_NODES, "out_nodes": _abi.OUT_NODES, "indptr": _abi.INDPTR, "indices": _abi.INDICES,
                "owned_out_nodes": _abi.OWNED_OUT_NODES, "self_ids_in": _abi.SELF_IDS_IN,
                "owned_degree": _abi.OWNED_DEGREE, "from_all": _abi.FROM_IDS, "to_all": _abi.TO_IDS,
                "t_indptr": _abi.T_INDPTR, "t_indices": _abi.T_INDICES}


class Slice(object):
    """One BiPartite of graph mode on the device: int32 tensors that alias the engine's arena.  The tensors are
    made when first asked for (a training step touches a third of the ~13 lists x 3 layers; every view costs a
    microsecond of host time); the sizes n_in / n_out / n_owned / n_edges come from the sample's meta.
    t_indptr / t_indices: the slice by source (engine flag FLAG_TRANSPOSE; empty otherwise and for the deepest layer).
    from_all / to_all: the per-peer lists back to back (receiver / sender order, the own one empty)."""

    __slots__ = ("part", "n_parts", "n_in", "n_out", "n_owned", "n_edges", "from_counts", "to_counts", "t_max_len",
                 "fanout", "_t", "_origin", "_lbase", "_lm") + \
        tuple(_SLICE_LISTS) + ("from_ids", "to_ids")

    def _seg(self, kind, lo, hi):
        o = self._origin + self._lbase[kind]
        return self._t[o + lo:o + hi]

    def ptr(self, kind):
        """device address of the part's list of `kind` (an _abi list kind) inside the engine's arena"""
        return self._t.data_ptr() + 4 * (self._origin + self._lbase[kind] + int(self._lm.off[kind][self.part]))

    def count(self, kind):
        """entries of the part's list of `kind`"""
        return int(self._lm.off[kind][self.part + 1]) - int(self._lm.off[kind][self.part])

    def __getattr__(self, name):          # only reached for a list that has not been made yet
        lm, g = self._lm, self.part
        kind = _SLICE_LISTS.get(name)
        if kind is not None:
            v = self._seg(kind, int(lm.off[kind][g]), int(lm.off[kind][g + 1]))
        elif name == "from_ids":
            f0 = int(lm.off[_abi.FROM_IDS][g])
            v = [self._seg(_abi.FROM_IDS, f0 + int(lm.pair_off[0][g][p]), f0 + int(lm.pair_off[0][g][p + 1]))
                 for p in range(self.n_parts)]
        elif name == "to_ids":
            t0 = int(lm.off[_abi.TO_IDS][g])
            v = [self._seg(_abi.TO_IDS, t0 + int(lm.pair_off[1][g][p]), t0 + int(lm.pair_off[1][g][p + 1]))
                 for p in range(self.n_parts)]
        else:
            raise AttributeError(name)
        object.__setattr__(self, name, v)
        return v


def _arena_tensors(eng, device):
    """One int32 tensor per layer over the engine's whole result arena, built once per engine:
    per-sample views are then plain tensor slices (torch.as_tensor on a raw pointer costs tens of
    microseconds; a training step needs ~40 views)."""
    cache = getattr(eng, "_arena_cache", None)
    if cache is None or cache[0] != device:
        layers = []
        for l in range(eng.n_layers):
            ptr, stride, lbase = eng.arena_info(l)
            t = torch.as_tensor(_DevArray(ptr, eng.n_slots * eng.n_streams * stride), device=device)
            layers.append((t, stride, lbase))
        cache = (device, layers)
        eng._arena_cache = cache
    return cache[1]


def slices_of(eng, stream=0, slot=0, parts=None, device=None, meta=None):
    """All layers x parts of one sample as `Slice`s (layer 0 = hop from the seeds).
    parts: iterable of part ids to materialise (default all)."""
    if eng.mode != _abi.MODE_GRAPH:
        raise ValueError("splitgnn needs an engine created with mode=MODE_GRAPH")
    device = device or torch.device("cuda", torch.cuda.current_device())
    m = meta if meta is not None else eng.meta(stream, slot)   # waits for the round that filled the slot
    P = eng.n_parts
    parts = range(P) if parts is None else parts
    arenas = _arena_tensors(eng, device)
    out = []
    for l in range(eng.n_layers):
        lm = m.layer[l]
        t, stride, lbase = arenas[l]
        origin = (slot * eng.n_streams + stream) * stride

        row = {}
        for g in parts:
            s = Slice()
            s.part, s.n_parts = g, P
            s._t, s._origin, s._lbase, s._lm = t, origin, lbase, lm
            s.n_in = int(lm.off[_abi.IN_NODES][g + 1]) - int(lm.off[_abi.IN_NODES][g])
            s.n_out = int(lm.off[_abi.OUT_NODES][g + 1]) - int(lm.off[_abi.OUT_NODES][g])
            s.n_owned = int(lm.off[_abi.OWNED_OUT_NODES][g + 1]) - int(lm.off[_abi.OWNED_OUT_NODES][g])
            s.n_edges = int(lm.off[_abi.INDICES][g + 1]) - int(lm.off[_abi.INDICES][g])
            # rows per peer of the boundary lists (from_ids[p] / to_ids[p]; the own entry is empty)
            s.from_counts = [int(lm.pair_off[0][g][p + 1]) - int(lm.pair_off[0][g][p]) for p in range(P)]
            s.to_counts = [int(lm.pair_off[1][g][p + 1]) - int(lm.pair_off[1][g][p]) for p in range(P)]
            s.t_max_len = int(lm.t_max_len[g])      # longest list of the slice by source (hubs: > _abi.T_SORTED_MAX)
            s.fanout = eng.fanouts[l]               # no row of the slice's CSR is longer
            row[g] = s
        out.append(row)
    return out


import os as _os
_NO_LOCAL_FUSE = bool(_os.environ.get("CSLICER_NO_LOCAL_FUSE"))   # A/B switch for the single-part fused layer
_GAT_RANK_AUTOGRAD = bool(_os.environ.get("CSLICER_GAT_RANK_AUTOGRAD"))   # A/B switch: a rank's GAT layer as separate autograd nodes
_NO_GAT_INPUT = bool(_os.environ.get("CSLICER_GAT_NO_INPUT_LAYER"))  # A/B switch: the deepest GAT layer projects its sources
# GEMM row counts are rounded up to a multiple of this (see DistSageConv.finish): shapes repeat, weight gradients split into
# SPLIT_K row slabs.  4096 (rounds 1-2) padded the 6-8 k-row middle layers by a third; the GEMM plans are kept per shape
# CLASS (long dimension blanked), so a finer quantum costs no tuning: profiles/row_pad_sweep.sh, 4096 / 1024 / 512 / 256:
# GraphSAGE 2,043 / 2,074 / 2,066 / 2,056, GAT 803 / 847 / 846 / 832 minibatches/s
ROW_PAD = int(_os.environ.get("CSLICER_ROW_PAD", "1024"))
SPLIT_K = 32     # the weight-gradient GEMM reduces over the rows in this many independent slabs


class _SplitKLinear(torch.autograd.Function):
    """y = x @ W^T + b for a tall x (rows a multiple of ROW_PAD).  Forward and the input gradient are the
    library's GEMMs; the WEIGHT gradient, a [out, rows] x [rows, in] product with a 10^5-long reduction and
    a 256 x 200 result, is what the library runs on ~32 workgroups of a 256-CU chip (0.29 ms per step in the
    rocprofv3 trace of the training step): it is computed as SPLIT_K batched slabs and summed instead."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return torch.addmm(bias, x, weight.t()) if bias is not None else x @ weight.t()

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gy @ weight if ctx.needs_input_grad[0] else None
        m = x.shape[0]
        gw = torch.bmm(gy.view(SPLIT_K, m // SPLIT_K, gy.shape[1]).transpose(1, 2),
                       x.view(SPLIT_K, m // SPLIT_K, x.shape[1])).sum(0)
        return gx, gw, (gy.sum(0) if ctx.has_bias else None)


class _SageFinish(torch.autograd.Function):
    """DistSageConv.finish (+ the ReLU after it) as ONE autograd node: csl_sage_cat_f32 writes the self rows and
    the degree-normalised merged sums straight into the two column blocks of the (row-padded) GEMM operand, the
    GEMM carries bias and ReLU in its epilogue, and the backward is csl_relu_bwd_colsum_f32 (mask + padding + bias
    gradient in one pass), three GEMMs and two row scatters instead of a chain of ~10 autograd nodes."""

    @staticmethod
    def forward(ctx, x, agg, weight, bias, self_ids_in, owned, deg, relu):
        m = owned.numel()
        mp = _pad_rows(m)
        cat = aggr.sage_cat(x, self_ids_in, m, mp, owned=owned, deg=deg, agg=agg)
        y = _linear_act(bias, cat, weight, relu)
        ctx.relu, ctx.m, ctx.n_x, ctx.n_agg = relu, m, x.shape[0], agg.shape[0]
        ctx.save_for_backward(cat, weight, self_ids_in, owned, deg, y if relu else None)
        return y[:m]

    @staticmethod
    def backward(ctx, gy):
        cat, weight, self_ids_in, owned, deg, y = ctx.saved_tensors
        m, mp, fin = ctx.m, cat.shape[0], cat.shape[1] // 2
        gyp, gb = aggr.relu_bwd_colsum(gy, y, m, mp)
        gw = _weight_grad(gyp, cat)
        gx = gagg = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            # one launch: zero fill, self rows and degree-normalised owned rows (both index lists are unique)
            gx, gagg = aggr.sage_cat_rows_bwd(self_ids_in, owned, deg, _input_grad(gyp, weight), m, ctx.n_x, ctx.n_agg,
                                              want_x=ctx.needs_input_grad[0], want_agg=ctx.needs_input_grad[1])
        return gx, gagg, gw, gb, None, None, None, None


def _pad_rows(m):
    return (m + ROW_PAD - 1) // ROW_PAD * ROW_PAD if m >= ROW_PAD else m


# CSLICER_DIRECT_GEMMS=1: the fused layers call hipBLASLt directly (aggr.gemm: cached plan per shape, the library's
# candidates timed on first use; ~5 us of host time per GEMM where torch's dispatcher + TunableOp lookup cost 25-35).
# Measured on the headline step (profiles/r2_e2e/README.md): the 48 candidates the library's heuristic offers are 6 %
# slower in sum than the recorded TunableOp selections (which search every solution), and a row count first seen in
# the timed region costs 16 ms of in-line timing -- 1.50-1.55 k minibatches/s against 1.70-1.78 k with torch's GEMMs
# on a step that is GPU-bound anyway.  Off by default; what a host without torch would use.
_DIRECT_GEMMS = bool(_os.environ.get("CSLICER_DIRECT_GEMMS"))


def _input_grad(gyp, weight):
    return aggr.gemm(gyp, weight) if _DIRECT_GEMMS and gyp.is_cuda else gyp @ weight


def _weight_grad(gyp, cat):
    """gy^T @ cat for a tall, row-padded pair: SPLIT_K batched slabs + a sum (see _SplitKLinear)."""
    mp = gyp.shape[0]
    if _DIRECT_GEMMS and gyp.is_cuda and (cat.shape[1] * gyp.shape[1]) % 4 == 0:
        if mp >= ROW_PAD and mp % SPLIT_K == 0:
            return aggr.weight_grad_slabs(gyp, cat, SPLIT_K)
        return aggr.gemm(gyp, cat, transa=True)
    if mp >= ROW_PAD and mp % SPLIT_K == 0:
        return torch.bmm(gyp.view(SPLIT_K, mp // SPLIT_K, gyp.shape[1]).transpose(1, 2),
                         cat.view(SPLIT_K, mp // SPLIT_K, cat.shape[1])).sum(0)
    return gyp.t() @ cat


def _linear_act(bias, cat, weight, relu):
    """cat @ W^T + b, with the ReLU in the GEMM's epilogue where the library offers it."""
    if _DIRECT_GEMMS and cat.is_cuda:
        return aggr.gemm(cat, weight, transb=True, bias=bias, relu=relu)
    if relu and _FUSED_EPILOGUE:
        return torch._addmm_activation(bias, cat, weight.t(), use_gelu=False)
    y = torch.addmm(bias, cat, weight.t())
    return y.relu_() if relu else y


_FUSED_EPILOGUE = hasattr(torch, "_addmm_activation") and not _os.environ.get("CSLICER_NO_FUSED_EPILOGUE")


class _SageLayerLocal(torch.autograd.Function):
    """A whole DistSageConv layer (+ ReLU) of a single part (nothing to exchange) as ONE autograd node over the
    fused HIP kernels: forward = csl_sage_cat_f32 (self gather + CSR mean straight into the GEMM operand; the
    deepest layer reads the resident feature table through `rowmap` = the slice's in_nodes, so the gathered
    input matrix never exists) + one GEMM with bias/ReLU epilogue; backward = csl_relu_bwd_colsum_f32 (ReLU
    mask + row padding + bias gradient), the slab-wise weight-gradient GEMM, the input-gradient GEMM and
    csl_sage_cat_bwd_f32.  dist_sageconv.py:42-84 for one GPU."""

    @staticmethod
    def forward(ctx, x, weight, bias, self_ids_in, indptr, indices, rowmap, n_out, n_src, relu):
        m, mp = n_out, _pad_rows(n_out)
        cat = aggr.sage_cat(x, self_ids_in, m, mp, indptr=indptr, indices=indices, rowmap=rowmap)
        y = _linear_act(bias, cat, weight, relu)
        ctx.m, ctx.n_src, ctx.relu = m, n_src, relu
        ctx.save_for_backward(cat, weight, y if relu else None, self_ids_in, indptr, indices)
        return y[:m]

    @staticmethod
    def backward(ctx, gy):
        cat, weight, y, self_ids_in, indptr, indices = ctx.saved_tensors
        m, mp = ctx.m, cat.shape[0]
        gyp, gb = aggr.relu_bwd_colsum(gy, y, m, mp)
        gw = _weight_grad(gyp, cat)
        gx = None
        if ctx.needs_input_grad[0]:
            gx = aggr.sage_cat_bwd(indptr, indices, self_ids_in, _input_grad(gyp, weight), m, ctx.n_src)
        return gx, gw, gb, None, None, None, None, None, None, None


class _SageModelLocal(torch.autograd.Function):
    """The WHOLE GraphSAGE model of a single part (one GPU holding every node) as one autograd node, so that the
    backward can fuse across layers: the input gradient of layer k is GATHERED over the slice by source the slicer
    emits (FLAG_TRANSPOSE; csl_sage_cat_bwd_t_f32: no atomics, no zero fill, deterministic) and in the same pass
    receives the ReLU mask of layer k-1, the row padding of its GEMM operand and its bias column sums.  Per layer:
    forward = csl_sage_cat_f32 + one GEMM (bias/ReLU epilogue); backward = weight-gradient GEMM (+ input-gradient
    GEMM and the fused gather for k > 0).  The deepest layer reads the resident feature table through its slice's
    in_nodes.  Arguments: feat table, n_layers, then per layer (weight, bias, Slice) in model order (deepest first)."""

    @staticmethod
    def forward(ctx, feat, n_layers, *args):
        ws, bs, sls = args[0::3], args[1::3], args[2::3]
        x, rowmap = feat, sls[0].in_nodes
        cats, ys = [], []
        for k in range(n_layers):
            sl = sls[k]
            m, mp = sl.n_out, _pad_rows(sl.n_out)
            cat = aggr.sage_cat(x, sl.self_ids_in, m, mp, indptr=sl.indptr, indices=sl.indices, rowmap=rowmap)
            y = _linear_act(bs[k], cat, ws[k], k + 1 < n_layers)
            cats.append(cat)
            ys.append(y)
            x, rowmap = y[:m], None
        ctx.n_layers, ctx.sls = n_layers, sls
        ctx.save_for_backward(*ws, *cats, *ys)
        return x

    @staticmethod
    def backward(ctx, g):
        L, sls = ctx.n_layers, ctx.sls
        t = ctx.saved_tensors
        ws, cats, ys = t[:L], t[L:2 * L], t[2 * L:]
        grads = [None] * (3 * L)
        top = sls[L - 1]
        gyp, gb = aggr.relu_bwd_colsum(g, None, top.n_out, cats[L - 1].shape[0])   # padding + bias sums (no mask)
        for k in range(L - 1, -1, -1):
            grads[3 * k] = _weight_grad(gyp, cats[k])
            grads[3 * k + 1] = gb
            if k == 0:
                break
            sl = sls[k]
            # gradient w.r.t. layer k-1's pre-activation output, padded like its GEMM operand, and its bias sums
            gyp, gb = aggr.sage_cat_bwd_t(sl.t_indptr, sl.t_indices, sl.indptr, _input_grad(gyp, ws[k]), ys[k - 1], sl.n_in,
                                          cats[k - 1].shape[0], hub=sl.t_max_len > _abi.T_SORTED_MAX)
        return (None, None) + tuple(grads)


class DistSageConv(nn.Module):
    """dist_sageconv.py:8-84: concat(self, aggregated neighbours) -> Linear(2*in, out)."""

    def __init__(self, in_feats, out_feats):
        super().__init__()
        self.fc = nn.Linear(2 * in_feats, out_feats)
        nn.init.xavier_uniform_(self.fc.weight, gain=nn.init.calculate_gain("relu"))  # dist_sageconv.py:35-41

    # -- the three phases of a layer on one part
    def local(self, sl, x):
        """sum-aggregate over the part's own CSR (BipartiteGraph.gather)."""
        return aggr.SpmmSum.apply(x, sl.indptr, sl.indices, sl.n_out)

    def boundary(self, sl, agg):
        """partials this part computed for nodes owned by peer p (pull_for_remotes)."""
        return [aggr.GatherRows.apply(agg, sl.from_ids[p]) if sl.from_ids[p].numel() else None
                for p in range(sl.n_parts)]

    def merge(self, sl, agg, recv):
        """add the peers' partials into the owned rows (push_from_remotes / mergeKernel)."""
        for p in range(sl.n_parts):
            if recv[p] is not None and sl.to_ids[p].numel():
                agg = aggr.ScatterAddRows.apply(agg, sl.to_ids[p], recv[p])
        return agg

    def finish_fused(self, sl, agg, x, relu):
        """finish (+ ReLU) as one autograd node (`_SageFinish`); same numbers as finish + torch.relu."""
        if x.shape[1] % 4:   # the fused operand kernel moves float4 columns
            y = self.finish(sl, agg, x)
            return torch.relu(y) if relu else y
        return _SageFinish.apply(x, agg, self.fc.weight, self.fc.bias, sl.self_ids_in, sl.owned_out_nodes,
                                 sl.owned_degree, relu)

    def layer_local(self, sl, x, relu, rowmap=None):
        """A whole layer of a single part (n_parts == 1: nothing to exchange) as one autograd node."""
        return _SageLayerLocal.apply(x, self.fc.weight, self.fc.bias, sl.self_ids_in, sl.indptr, sl.indices, rowmap,
                                     sl.n_out, sl.n_in, relu)

    def finish(self, sl, agg, x):
        """slice_owned_nodes + mean + self_gather + concat + Linear."""
        neigh = aggr.GatherMeanRows.apply(agg, sl.owned_out_nodes, sl.owned_degree)
        self_h = aggr.GatherRows.apply(x, sl.self_ids_in)
        # (splitting the Linear into two addmm over weight column blocks to avoid this concat was
        # measured 1.5x slower end to end: strided GEMM operands)
        cat = torch.cat([self_h, neigh], dim=1)
        # The row count is different in every minibatch, and hipBLASLt pays a heuristic search for every
        # GEMM shape it has not seen (80 us instead of 25 us of host time per call, three GEMMs per layer
        # with backward: profiles/gemm_shape_probe.py).  Rounding the rows up makes the shapes repeat.
        m = cat.shape[0]
        mp = (m + ROW_PAD - 1) // ROW_PAD * ROW_PAD
        if m >= ROW_PAD:
            if mp != m:
                cat = torch.nn.functional.pad(cat, (0, 0, 0, mp - m))
            return _SplitKLinear.apply(cat, self.fc.weight, self.fc.bias)[:m]
        return self.fc(cat)


class DistSAGEModel(nn.Module):
    """models/factory.py:7-56: n_layers DistSageConv, ReLU between, weights replicated on every part."""

    def __init__(self, in_feats, hidden, n_classes, n_layers=3):
        super().__init__()
        dims = [in_feats] + [hidden] * (n_layers - 1) + [n_classes]
        # convs[k] consumes engine layer n_layers-1-k (the deepest hop first)
        self.convs = nn.ModuleList([DistSageConv(dims[k], dims[k + 1]) for k in range(n_layers)])

    def forward_parts(self, slices, feats):
        """All parts in ONE process (validation / single GPU): slices[l][g], feats[g] = input
        features of slices[L-1][g].in_nodes.  Returns per part the logits of its owned seeds."""
        L = len(slices)
        parts = sorted(slices[0].keys())
        x = {g: feats[g] for g in parts}
        for k, conv in enumerate(self.convs):
            sl = slices[L - 1 - k]
            if len(parts) == 1 and sl[parts[0]].n_parts == 1 and not _NO_LOCAL_FUSE and x[parts[0]].shape[1] % 4 == 0:
                x = {parts[0]: conv.layer_local(sl[parts[0]], x[parts[0]], k + 1 < len(self.convs))}
                continue
            agg = {g: conv.local(sl[g], x[g]) for g in parts}
            send = {g: conv.boundary(sl[g], agg[g]) for g in parts}
            for g in parts:
                agg[g] = conv.merge(sl[g], agg[g], [send[p][g] if p != g else None for p in parts])
            x = {g: conv.finish_fused(sl[g], agg[g], x[g], k + 1 < len(self.convs)) for g in parts}
        return x

    def forward_local(self, slices, feat_table, part=0):
        """A single part holding every node (one GPU): each layer is one `_SageLayerLocal` node.  `feat_table` is
        the resident [N, F] feature matrix; the deepest layer indexes it through its slice's in_nodes."""
        L = len(slices)
        if L > 1 and all(slices[l][part].t_indptr.numel() for l in range(L - 1)):
            # the engine emitted the slices by source: one node for the model, gathered input gradients
            args = []
            for k, conv in enumerate(self.convs):
                args += [conv.fc.weight, conv.fc.bias, slices[L - 1 - k][part]]
            return _SageModelLocal.apply(feat_table, L, *args)
        x, rowmap = feat_table, slices[L - 1][part].in_nodes
        for k, conv in enumerate(self.convs):
            x = conv.layer_local(slices[L - 1 - k][part], x, k + 1 < len(self.convs), rowmap)
            rowmap = None
        return x

    def forward_rank(self, slices, feat, rank, comm, overlap=False):
        """One part per process: boundary partials go through `comm.all_to_all` (RCCL).
        overlap=True runs each layer's exchange on a side stream while the rows that stay on
        this GPU are aggregated (`_RankAggregate`); same numbers, different schedule."""
        L = len(slices)
        x = feat
        for k, conv in enumerate(self.convs):
            sl = slices[L - 1 - k][rank]
            # aggregation + boundary exchange + merge of a layer are ONE autograd node (`_RankAggregate`): 5 launches
            # forward whatever the number of peers (the op chain local / boundary / merge of forward_parts is
            # 2 + 2 (P - 1) launches and as many autograd nodes)
            x = conv.finish_fused(sl, _RankAggregate.apply(x, sl, comm, bool(overlap)), x, k + 1 < len(self.convs))
        return x



class DistGATConv(nn.Module):
    """Multi-head graph attention over the slices (BASELINE config 5).  The reference has no implementation
    (python/layers/dist_gatconv.py:3-6 is a stub; bipartite.py:75-80 `attention_gather` is the only piece), so
    the layer is DEFINED here, GATConv-style, and pinned against a dense torch computation ("parity unpinned"
    by the reference):

        z = W x;  el = <z, a_l>;  er = <z, a_r>   per head
        alpha(u -> v) = softmax over the sampled in-edges of v of LeakyReLU(el[u] + er[v])
        out[v] = sum_u alpha(u -> v) z[u] + bias          (a node without sampled neighbours gets the bias)

    Split-parallel form: part g holds z/el of the sources it owns and the edges out of them.  The owner of a
    destination sends its er to the parts that hold edges into it (the boundary lists, reverse direction), every
    part computes the partial softmax state (m, s, n) of its local edges in ONE fused HIP pass
    (aggr.GatAggregate), the partials travel to the owner like SAGE's partial sums and are merged there with the
    usual log-sum-exp rescaling; out = n / s."""

    def __init__(self, in_feats, out_feats, heads, slope=0.2):
        super().__init__()
        self.H, self.D, self.slope = heads, out_feats, slope
        self.fc = nn.Linear(in_feats, heads * out_feats, bias=False)
        self.attn_l = nn.Parameter(torch.empty(heads, out_feats))
        self.attn_r = nn.Parameter(torch.empty(heads, out_feats))
        self.bias = nn.Parameter(torch.zeros(heads * out_feats))
        gain = nn.init.calculate_gain("relu")
        nn.init.xavier_normal_(self.fc.weight, gain=gain)
        nn.init.xavier_normal_(self.attn_l, gain=gain)
        nn.init.xavier_normal_(self.attn_r, gain=gain)

    def project(self, x):
        """z = W x and the two attention logits per head, el = <z, a_l>, er = <z, a_r>: one GEMM and one row-wise
        HIP reduction over z (`aggr.GatLogits`; as two [rows, in] x [in, H] GEMMs on x the logits were the slowest
        kernels of the step: H = 8 output columns)."""
        w = self.fc.weight
        m = x.shape[0]
        if m >= ROW_PAD:  # tall operand: shapes that repeat + slab-wise weight gradient (see _SplitKLinear)
            mp = (m + ROW_PAD - 1) // ROW_PAD * ROW_PAD
            xp = torch.nn.functional.pad(x, (0, 0, 0, mp - m)) if mp != m else x
            z = _SplitKLinear.apply(xp, w, None)[:m]
        else:
            z = self.fc(x)
        el, er = aggr.GatLogits.apply(z, self.attn_l, self.attn_r)
        return z, el, er

    def forward_parts(self, sl, x, elu=False, pad_out=False):
        """sl[g]: Slice of part g, x[g]: features of sl[g].in_nodes.  Returns per part the [n_owned, H*D]
        output rows of the nodes it owns (frontier order); elu: apply the ELU that follows a hidden layer; pad_out
        (fused single-part layer only): hand the output on as an aggr.PaddedRows."""
        parts = sorted(sl.keys())
        if len(parts) == 1 and sl[parts[0]].n_parts == 1 and not _NO_LOCAL_FUSE and (
                isinstance(x[parts[0]], (aggr.PaddedRows, aggr.FeatureRows)) or x[parts[0]].is_cuda):
            # one part holding every node: the whole layer (+ ELU) as one autograd node.  The layers hand each other
            # row-padded buffers (aggr.PaddedRows), so that none copies its input into a padded GEMM operand again
            g = parts[0]
            xin = x[g]
            if isinstance(xin, aggr.FeatureRows):
                # the deepest layer on the resident feature table: aggregate the raw rows, project the destinations
                if (not _NO_GAT_INPUT and not xin.table.requires_grad      # (the layer returns no input gradient)
                        and aggr.gat_input_ok(self.H, xin.table.shape[1], sl[g].fanout)):
                    out = aggr.GatInputLayer.apply(xin.table, xin.rows, self.fc.weight, self.attn_l, self.attn_r, self.bias,
                                                   sl[g].indptr, sl[g].indices, sl[g].self_ids_in, sl[g].n_out,
                                                   sl[g].n_edges, sl[g].fanout, self.slope, bool(elu), ROW_PAD, bool(pad_out))
                    return {g: aggr.PaddedRows(out, sl[g].n_out) if pad_out else out}
                xp = aggr.padded_rows(xin.rows.numel(), xin.table.shape[1], ROW_PAD, xin.table.device)
                aggr.gather_rows(xin.table, xin.rows, out=xp.t[:xp.n])
                xin = xp
            padded = isinstance(xin, aggr.PaddedRows)
            out = aggr.GatLayerLocal.apply(xin.t if padded else xin, self.fc.weight, self.attn_l, self.attn_r, self.bias,
                                           sl[g].indptr, sl[g].indices, sl[g].self_ids_in, sl[g].n_out, self.slope,
                                           bool(elu), ROW_PAD, _weight_grad, sl[g].t_indptr, sl[g].t_indices,
                                           sl[g].t_max_len, xin.n if padded else None, bool(pad_out))
            return {g: aggr.PaddedRows(out, sl[g].n_out) if pad_out else out}
        x = {g: (aggr.gather_rows(v.table, v.rows) if isinstance(v, aggr.FeatureRows) else v) for g, v in x.items()}
        out = self._forward_parts(sl, {g: (v.t[:v.n] if isinstance(v, aggr.PaddedRows) else v) for g, v in x.items()})
        return {g: torch.nn.functional.elu(v) for g, v in out.items()} if elu else out

    def _forward_parts(self, sl, x):
        parts = sorted(sl.keys())
        H, D = self.H, self.D
        if len(parts) == 1 and sl[parts[0]].n_parts == 1:
            # one part holding every node: every out node is owned (owned_out_nodes = 0..n_out-1), nothing to merge
            g = parts[0]
            z, el, er = self.project(x[g])
            er_out = aggr.GatherRows.apply(er, sl[g].self_ids_in)
            _, S, N = aggr.GatAggregate.apply(el, er_out, z, sl[g].indptr, sl[g].indices, sl[g].n_out, H, D, self.slope)
            return {g: (N.view(-1, H, D) / S.clamp_min(1e-30).unsqueeze(-1)).reshape(-1, H * D) + self.bias}
        proj = {g: self.project(x[g]) for g in parts}
        # er of the destinations: owned rows from the part's own projection ...
        er_out = {}
        for g in parts:
            e = torch.zeros((sl[g].n_out, H), dtype=torch.float32, device=x[g].device)
            er_out[g] = e.index_copy(0, sl[g].owned_out_nodes.long(), proj[g][2][sl[g].self_ids_in.long()])
        er_own = dict(er_out)
        # ... rows owned by a peer from that peer (boundary lists, reverse direction of the partial sums)
        for g in parts:
            for p in parts:
                if p != g and sl[g].from_ids[p].numel():
                    er_out[g] = er_out[g].index_copy(0, sl[g].from_ids[p].long(), er_own[p][sl[p].to_ids[g].long()])
        part = {g: aggr.GatAggregate.apply(proj[g][1], er_out[g], proj[g][0], sl[g].indptr, sl[g].indices,
                                           sl[g].n_out, H, D, self.slope) for g in parts}
        out = {}
        for p in parts:
            M, S, N = part[p]
            N = N.view(-1, H, D)
            for g in parts:
                if g == p or not sl[p].to_ids[g].numel():
                    continue
                idx, src = sl[p].to_ids[g].long(), sl[g].from_ids[p].long()
                m2, s2, n2 = part[g][0][src], part[g][1][src], part[g][2].view(-1, H, D)[src]
                Mi = M[idx]
                Mn = torch.maximum(Mi, m2)
                a, b = torch.exp(Mi - Mn), torch.exp(m2 - Mn)
                S = S.index_copy(0, idx, S[idx] * a + s2 * b)
                N = N.index_copy(0, idx, N[idx] * a.unsqueeze(-1) + n2 * b.unsqueeze(-1))
                M = M.index_copy(0, idx, Mn)
            own = sl[p].owned_out_nodes.long()
            out[p] = (N[own] / S[own].clamp_min(1e-30).unsqueeze(-1)).reshape(-1, H * D) + self.bias
        return out


    def _forward_rank(self, sl, x, comm):
        """One part per process (DistGATConv.forward_rank): two boundary exchanges per layer.  On the GPU the layer is one
        autograd node (aggr.GatLayerRank: fused kernels, exchanges over the back-to-back per-peer lists);
        CSLICER_GAT_RANK_AUTOGRAD=1: the node-by-node form below (A/B switch, and what the CPU / gloo tests run)."""
        if x.is_cuda and not _GAT_RANK_AUTOGRAD:
            return aggr.GatLayerRank.apply(x, self.fc.weight, self.attn_l, self.attn_r, self.bias, sl, comm, self.slope, False)
        g, P, H, D = sl.part, sl.n_parts, self.H, self.D
        z, el, er = self.project(x)
        er_out = torch.zeros((sl.n_out, H), dtype=torch.float32, device=x.device)
        er_out = er_out.index_copy(0, sl.owned_out_nodes.long(), er[sl.self_ids_in.long()])
        # 1) owners -> holders of boundary edges: er of the destinations (reverse direction of the lists)
        send = [er_out[sl.to_ids[p].long()] if p != g and sl.to_ids[p].numel() else None for p in range(P)]
        recv, tie = comm.all_to_all(send, [0 if p == g else sl.from_ids[p].numel() for p in range(P)], H)
        er_out = er_out + tie
        for p in range(P):
            if recv[p] is not None:
                er_out = er_out.index_copy(0, sl.from_ids[p].long(), recv[p])
        M, S, N = aggr.GatAggregate.apply(el, er_out, z, sl.indptr, sl.indices, sl.n_out, H, D, self.slope)
        # 2) holders -> owners: the partial softmax state (m | s | n) of the boundary rows
        packed = torch.cat([M, S, N], dim=1)
        send = [packed[sl.from_ids[p].long()] if p != g and sl.from_ids[p].numel() else None for p in range(P)]
        recv, tie = comm.all_to_all(send, [0 if p == g else sl.to_ids[p].numel() for p in range(P)], 2 * H + H * D)
        S = S + tie
        N = N.view(-1, H, D)
        for p in range(P):
            if recv[p] is None:
                continue
            idx = sl.to_ids[p].long()
            m2, s2, n2 = recv[p][:, :H].detach(), recv[p][:, H:2 * H], recv[p][:, 2 * H:].reshape(-1, H, D)
            Mi = M[idx]
            Mn = torch.maximum(Mi, m2)
            a, b = torch.exp(Mi - Mn), torch.exp(m2 - Mn)
            S = S.index_copy(0, idx, S[idx] * a + s2 * b)
            N = N.index_copy(0, idx, N[idx] * a.unsqueeze(-1) + n2 * b.unsqueeze(-1))
            M = M.index_copy(0, idx, Mn)
        own = sl.owned_out_nodes.long()
        return (N[own] / S[own].clamp_min(1e-30).unsqueeze(-1)).reshape(-1, H * D) + self.bias


class DistGATModel(nn.Module):
    """n_layers DistGATConv: hidden layers concatenate their heads (ELU), the last layer averages them."""

    def __init__(self, in_feats, hidden, n_classes, heads=8, n_layers=3):
        super().__init__()
        dims = [in_feats] + [hidden * heads] * (n_layers - 1)
        # the aggregation kernel moves float4 columns: per-head widths are multiples of 4 (the class
        # dimension is padded, the extra logits are dropped)
        self.n_classes = n_classes
        outs = [hidden] * (n_layers - 1) + [(n_classes + 3) // 4 * 4]
        self.heads = heads
        self.convs = nn.ModuleList([DistGATConv(dims[k], outs[k], heads) for k in range(n_layers)])

    def forward_parts(self, slices, feats):
        L = len(slices)
        parts = sorted(slices[0].keys())
        x = {g: feats[g] for g in parts}
        fused = (len(parts) == 1 and slices[0][parts[0]].n_parts == 1 and not _NO_LOCAL_FUSE and
                 (isinstance(x[parts[0]], (aggr.PaddedRows, aggr.FeatureRows)) or x[parts[0]].is_cuda))
        for k, conv in enumerate(self.convs):
            last = k + 1 == len(self.convs)
            x = conv.forward_parts(slices[L - 1 - k], x, elu=not last, pad_out=fused and not last)
            if last:
                x = {g: x[g].view(-1, self.heads, conv.D).mean(1)[:, :self.n_classes] for g in parts}
        return x

    def forward_rank(self, slices, feat, rank, comm):
        """One part per process: `comm` (DistComm) carries the two exchanges of every layer over RCCL."""
        L = len(slices)
        x = feat
        for k, conv in enumerate(self.convs):
            x = conv._forward_rank(slices[L - 1 - k][rank], x, comm)
            if k + 1 < len(self.convs):
                x = torch.nn.functional.elu(x)
            else:
                x = x.view(-1, self.heads, conv.D).mean(1)[:, :self.n_classes]
        return x


class _RankAggregate(torch.autograd.Function):
    """local sum-aggregate + boundary exchange + merge of ONE layer on ONE rank (dist_sageconv.py:52-65:
    pull_for_remotes, the exchange, push_from_remotes), as one autograd node.  The boundary lists of a slice are
    contiguous over the peers, so one gather builds the send buffer and one (atomic) scatter-add merges what every
    peer sent.  overlap=True puts the all-to-all on the communicator's side stream while the rows that never
    leave the GPU are still being aggregated; same numbers, different schedule.

    forward   rows owned by peers first -> send buffer -> [all-to-all, side stream if overlap]
              || rows owned by this part -> wait -> add received partials into the owned rows
    backward  grads of received partials -> [reverse all-to-all, side stream if overlap]
              || backward of the owned rows -> wait -> backward of the peer-owned rows
    """

    @staticmethod
    def _exchange(comm, buf, send_counts, recv_counts, overlap):
        """all-to-all of row blocks; returns (received rows, event to wait for or None)."""
        if not overlap:
            return comm._exchange(buf, send_counts, recv_counts), None
        main, side = torch.cuda.current_stream(), comm.side_stream()
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ready)
            out = comm._exchange(buf, send_counts, recv_counts)
            done = torch.cuda.Event()
            done.record(side)
        buf.record_stream(side)
        return out, done

    @staticmethod
    def forward(ctx, x, sl, comm, overlap):
        g, P = sl.part, sl.n_parts
        send_counts = [0 if p == g else sl.from_ids[p].numel() for p in range(P)]
        recv_counts = [0 if p == g else sl.to_ids[p].numel() for p in range(P)]
        H = x.shape[1]
        agg = torch.empty((sl.n_out, H), dtype=torch.float32, device=x.device)
        # the partial sums of the rows peers own go straight into the send buffer; their rows of `agg` are never
        # written and never read (DistSageConv.finish takes the owned rows only)
        send_cat = aggr.spmm_sum_compact(sl.indptr, sl.indices, x, sl.from_all)
        recv_cat, done = _RankAggregate._exchange(comm, send_cat, send_counts, recv_counts, overlap)
        aggr.spmm_sum(sl.indptr, sl.indices, x, sl.n_out, rows=sl.owned_out_nodes, out=agg)   # overlaps
        if done is not None:
            torch.cuda.current_stream().wait_event(done)
            recv_cat.record_stream(torch.cuda.current_stream())
        aggr.scatter_add_rows_atomic_(agg, sl.to_all, recv_cat)   # an owned node may receive from several peers
        ctx.sl, ctx.comm, ctx.n_src, ctx.overlap = sl, comm, x.shape[0], overlap
        ctx.send_counts, ctx.recv_counts = send_counts, recv_counts
        return agg

    @staticmethod
    def backward(ctx, G):
        sl, comm = ctx.sl, ctx.comm
        G = G.contiguous()
        g_recv = aggr.gather_rows(G, sl.to_all)          # d loss / d (partials received from each peer)
        back, done = _RankAggregate._exchange(comm, g_recv, ctx.recv_counts, ctx.send_counts, ctx.overlap)
        gx = aggr.spmm_sum_bwd(sl.indptr, sl.indices, G, ctx.n_src, rows=sl.owned_out_nodes)   # overlaps
        if done is not None:
            torch.cuda.current_stream().wait_event(done)
            back.record_stream(torch.cuda.current_stream())
        # (the peer-owned rows of `agg` have no consumer on this rank: their gradient is what the peers send back)
        aggr.spmm_sum_bwd(sl.indptr, sl.indices, back, ctx.n_src, rows=sl.from_all, compact=True, out=gx)
        return gx, None, None, None


class _AllToAllRows(torch.autograd.Function):
    """Differentiable all_to_all_single of row blocks: backward is the reverse exchange."""

    @staticmethod
    def forward(ctx, comm, send_cat, send_counts, recv_counts):
        ctx.comm, ctx.send_counts, ctx.recv_counts = comm, send_counts, recv_counts
        return comm._exchange(send_cat, send_counts, recv_counts)

    @staticmethod
    def backward(ctx, g):
        return None, ctx.comm._exchange(g.contiguous(), ctx.recv_counts, ctx.send_counts), None, None


class DistComm(object):
    """Boundary exchange between parts = ranks of a torch.distributed group (backend "nccl" is
    RCCL on ROCm; "gloo" for CPU tests).  One all_to_all_single per layer."""

    def __init__(self, group=None, device=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.device = device if device is not None else (
            torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu"))
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._side = None

    def side_stream(self):
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        return self._side

    def _exchange(self, send_cat, send_counts, recv_counts):
        with _roctx.range("exchange"):
            return self._exchange_impl(send_cat, send_counts, recv_counts)

    def exchange_into(self, out, send_cat, send_counts, recv_counts):
        """the same exchange into a caller-owned [sum(recv_counts), H] tensor (the native rank step's workspace)"""
        if self.dist.get_backend(self.group) == "gloo" and send_cat.is_cuda:
            host = torch.empty(tuple(out.shape), dtype=send_cat.dtype)
            self.dist.all_to_all_single(host, send_cat.cpu(), output_split_sizes=list(recv_counts),
                                        input_split_sizes=list(send_counts), group=self.group)
            out.copy_(host)
            return out
        self.dist.all_to_all_single(out, send_cat, output_split_sizes=list(recv_counts),
                                    input_split_sizes=list(send_counts), group=self.group)
        return out

    def _exchange_impl(self, send_cat, send_counts, recv_counts):
        H = send_cat.shape[1]
        if self.dist.get_backend(self.group) == "gloo" and send_cat.is_cuda:
            # rehearsal backend (several ranks sharing one GPU): stage through host memory
            out = torch.empty((sum(recv_counts), H), dtype=send_cat.dtype)
            self.dist.all_to_all_single(out, send_cat.cpu(), output_split_sizes=list(recv_counts),
                                        input_split_sizes=list(send_counts), group=self.group)
            return out.to(send_cat.device)
        out = torch.empty((sum(recv_counts), H), dtype=send_cat.dtype, device=send_cat.device)
        self.dist.all_to_all_single(out, send_cat, output_split_sizes=list(recv_counts),
                                    input_split_sizes=list(send_counts), group=self.group)
        return out

    def all_to_all(self, send, recv_counts, H):
        ref = next((t for t in send if t is not None), None)
        device = ref.device if ref is not None else self.device
        send_counts = [0 if t is None else t.shape[0] for t in send]
        parts = [t for t in send if t is not None]
        send_cat = torch.cat(parts, dim=0) if parts else torch.zeros((0, H), dtype=torch.float32, device=device)
        if torch.is_grad_enabled() and not send_cat.requires_grad:
            send_cat = send_cat.detach().requires_grad_()
        out = _AllToAllRows.apply(self, send_cat, send_counts, list(recv_counts))
        recv, o = [], 0
        for c in recv_counts:
            recv.append(out[o:o + c] if c else None)
            o += c
        return recv, out.sum() * 0.0

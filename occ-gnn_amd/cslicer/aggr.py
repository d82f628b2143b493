"""Split-parallel aggregation ops over the slices: ctypes binding of include/cslicer_aggr.h
plus torch.autograd wrappers.  torch only provides device memory, streams and autograd
plumbing; the work is done by the HIP kernels in occ-gnn_amd/csrc/aggregate.hip.

Reference semantics: python/data/bipartite.py:61-99 (gather, self_gather, pull_for_remotes,
push_from_remotes), src/gnn/sage.cu:7-28, src/gnn/dist_sage.cu:193-199.
"""
import ctypes as C
import os

import torch

from . import _abi

SYMBOLS = ["csl_spmm_sum_f32", "csl_spmm_sum_bwd_f32", "csl_gather_rows_f32",
           "csl_scatter_add_rows_f32", "csl_div_rows_f32", "csl_gat_fwd_f32", "csl_gat_bwd_f32",
           "csl_sage_cat_f32", "csl_sage_cat_bwd_f32", "csl_relu_bwd_colsum_f32", "csl_softmax_ce_f32",
           "csl_relu_bwd_colsum_scratch", "csl_softmax_ce_scratch", "csl_adam_f32",
           "csl_scatter_add_rows_atomic_f32", "csl_gat_logits_fwd_f32", "csl_gat_logits_bwd_f32",
           "csl_gat_logits_bwd_scratch", "csl_spmm_sum_compact_f32", "csl_sage_cat_rows_bwd_f32",
           "csl_sage_cat_bwd_t_f32", "csl_sage_cat_bwd_t_scratch", "csl_gemm_f32", "csl_gemm_last_error",
           "csl_sum_slabs_f32", "csl_sage_fwd_bwd_f32", "csl_sage_fwd_bwd_workspace", "csl_sage_last_error",
           "csl_gemm_save_plans", "csl_gemm_load_plans", "csl_softmax_ce_partial_f32", "csl_reduce_multi_f32",
           "csl_sage_rank_fwd_bwd_f32", "csl_sage_rank_workspace", "csl_gat_logits_bwd_acc_f32", "csl_gat_finish_fwd_f32", "csl_gat_finish_bwd_f32", "csl_gat_finish_bwd_scratch",
           "csl_gat_bwd_t_f32", "csl_sage_fwd_mfma_f32", "csl_sage_fwd_mfma_scratch", "csl_sage_step_timing",
           "csl_sage_step_timing_read", "csl_sage_cat_bwd_t_hub_f32", "csl_sage_cat_bwd_t_hub_scratch",
           "csl_gat_bwd_t_fused_f32", "csl_gat_bwd_t_fused_scratch", "csl_sage_rank_g2_f32", "csl_scatter_rows_f32", "csl_spmm_sum_map_f32",
           "csl_gat_in_max_degree", "csl_gat_in_fwd_f32", "csl_gat_in_bwd_scratch", "csl_gat_in_bwd_f32", "csl_bias_elu_f32",
           "csl_elu_bwd_colsum_scratch", "csl_elu_bwd_colsum_f32", "csl_gat_in_proj_ok", "csl_gat_in_proj_fpad",
           "csl_gat_in_proj_f32", "csl_gat_in_proj_bwd_scratch", "csl_gat_in_proj_bwd_f32", "csl_gat_in_layer_fwd_scratch",
           "csl_gat_in_layer_fwd_f32", "csl_gat_in_layer_bwd_scratch", "csl_gat_in_layer_bwd_f32"]
_ready = False
_GAT_TORCH_MM = bool(os.environ.get("CSLICER_GAT_TORCH_MM"))


def _lib():
    global _ready
    L = _abi.load()
    if not _ready:
        vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int32
        L.csl_spmm_sum_f32.argtypes = [vp, vp, vp, i64, vp, i64, vp, i64, i32, vp]
        L.csl_spmm_sum_bwd_f32.argtypes = [vp, vp, vp, i64, vp, i64, i32, vp, i64, i32, vp]
        L.csl_spmm_sum_compact_f32.argtypes = [vp, vp, vp, i64, vp, i64, vp, i64, i32, vp]
        L.csl_sage_cat_rows_bwd_f32.argtypes = [vp, vp, vp, i64, vp, i64, vp, i64, vp, i64, i32, vp]
        L.csl_gather_rows_f32.argtypes = [vp, i64, vp, i64, vp, i64, i32, vp]
        L.csl_scatter_add_rows_f32.argtypes = [vp, i64, vp, i64, vp, i64, i32, vp]
        L.csl_div_rows_f32.argtypes = [vp, i64, vp, i64, i32, vp]
        L.csl_scatter_add_rows_atomic_f32.argtypes = [vp, i64, vp, i64, vp, i64, i32, vp]
        L.csl_gat_logits_fwd_f32.argtypes = [vp, vp, vp, i64, i32, i32, vp, vp, vp]
        L.csl_gat_logits_bwd_f32.argtypes = [vp, vp, vp, vp, vp, i64, i32, i32, vp, vp, vp, vp, vp]
        L.csl_gat_logits_bwd_scratch.argtypes = [i64, i32, i32]
        L.csl_gat_logits_bwd_scratch.restype = i64
        f32 = C.c_float
        L.csl_gat_fwd_f32.argtypes = [vp, vp, i64, vp, vp, vp, i32, i32, f32, vp, vp, vp, vp]
        L.csl_gat_bwd_f32.argtypes = [vp, vp, i64, vp, vp, vp, i32, i32, f32, vp, vp, vp, vp, vp, vp, vp]
        L.csl_sage_cat_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp, i64, vp, i64, i64, i64, vp, i64, i32, i32, vp]
        L.csl_sage_cat_bwd_f32.argtypes = [vp, vp, vp, i64, vp, i64, vp, i64, i64, i32, vp]
        L.csl_relu_bwd_colsum_f32.argtypes = [vp, i64, vp, i64, i64, i64, vp, i64, vp, vp, i32, vp]
        L.csl_relu_bwd_colsum_scratch.argtypes = [i64, i32]
        L.csl_relu_bwd_colsum_scratch.restype = i64
        L.csl_softmax_ce_f32.argtypes = [vp, i64, i64, i32, vp, vp, vp, f32, vp, vp, i64, vp, vp]
        L.csl_softmax_ce_scratch.argtypes = [i64]
        L.csl_softmax_ce_scratch.restype = i64
        L.csl_adam_f32.argtypes = [i32, vp, vp, vp, vp, vp, f32, f32, f32, f32, i64, vp]
        L.csl_sage_cat_bwd_t_f32.argtypes = [vp, vp, vp, vp, i64, vp, i64, i64, i64, vp, i64, vp, vp, i32, vp]
        L.csl_sage_cat_bwd_t_scratch.argtypes = [i64, i32]
        L.csl_sage_cat_bwd_t_scratch.restype = i64
        L.csl_sage_cat_bwd_t_hub_f32.argtypes = [vp, vp, i64, vp, vp, i64, vp, i64, i64, i64, vp, i64, vp, vp, i32, vp]
        L.csl_sage_cat_bwd_t_hub_scratch.argtypes = [i64, i32]
        L.csl_sage_cat_bwd_t_hub_scratch.restype = i64
        L.csl_gemm_f32.argtypes = [i32, i32, i64, i64, i64, vp, i64, i64, vp, i64, i64, vp, i64, i64, i32, vp, i32, vp]
        L.csl_gemm_last_error.restype = C.c_char_p
        L.csl_sum_slabs_f32.argtypes = [vp, i64, i32, vp, vp]
        L.csl_sage_fwd_bwd_workspace.argtypes = [i32, vp, vp, i64, i32]
        L.csl_sage_fwd_bwd_workspace.restype = i64
        L.csl_sage_fwd_bwd_f32.argtypes = [i32, vp, vp, vp, vp, vp, i64, vp, vp, vp, f32, i64, i32, vp, vp, vp, i64, vp]
        L.csl_sage_last_error.restype = C.c_char_p
        L.csl_gemm_save_plans.argtypes = [C.c_char_p]
        L.csl_sage_rank_workspace.argtypes = [i32, vp, vp, i64, i32]
        L.csl_sage_rank_workspace.restype = i64
        L.csl_sage_rank_fwd_bwd_f32.argtypes = [i32, vp, vp, vp, vp, vp, i64, vp, vp, vp, vp, f32, i64, i32, EXCHANGE_FN,
                                                EXCHANGE_WAIT_FN, vp, vp, vp, vp, i64, vp]
        L.csl_gat_logits_bwd_acc_f32.argtypes = [vp, vp, vp, vp, vp, i64, i32, i32, vp, i32, vp, vp, vp, vp]
        L.csl_gat_bwd_t_f32.argtypes = [vp, vp, i64, i64, vp, vp, vp, i32, i32, f32, vp, vp, vp, vp, vp, vp, vp]
        L.csl_gat_bwd_t_fused_f32.argtypes = [vp, vp, i64, i64, vp, vp, vp, i32, i32, f32, vp, vp, vp, vp, vp, vp, i64, vp, vp,
                                              vp, vp, vp, vp]
        L.csl_sage_rank_g2_f32.argtypes = [vp, vp, i64, vp, i64, vp, i64, i32, vp]
        L.csl_scatter_rows_f32.argtypes = [vp, i64, vp, i64, vp, i64, i32, vp]
        L.csl_spmm_sum_map_f32.argtypes = [vp, vp, vp, i64, vp, i64, vp, vp, i64, i32, i32, vp]
        L.csl_gat_bwd_t_fused_scratch.argtypes = [i64, i64, i32, i32]
        L.csl_gat_bwd_t_fused_scratch.restype = i64
        L.csl_gat_finish_fwd_f32.argtypes = [vp, vp, vp, i64, i32, i32, i32, vp, vp]
        L.csl_gat_finish_bwd_f32.argtypes = [vp, i64, vp, vp, vp, i64, i32, i32, i32, vp, vp, vp, vp, vp]
        L.csl_gat_finish_bwd_scratch.argtypes = [i64, i32, i32]
        L.csl_gat_finish_bwd_scratch.restype = i64
        L.csl_gemm_load_plans.argtypes = [C.c_char_p]
        L.csl_sage_step_timing.argtypes = [i32]
        L.csl_sage_step_timing_read.argtypes = [vp, vp]
        L.csl_sage_fwd_mfma_scratch.argtypes = [i32, i32]
        L.csl_sage_fwd_mfma_scratch.restype = i64
        L.csl_sage_fwd_mfma_f32.argtypes = [vp, vp, vp, vp, vp, i64, vp, i64, vp, i64, i64, i32, i32, i32, i32, vp, i64,
                                            vp, i64, vp, vp]
        L.csl_gat_in_fwd_f32.argtypes = [vp, vp, vp, vp, vp, i64, i32, vp, vp, i32, f32, i64, i64, i32, vp, vp, vp]
        L.csl_gat_in_bwd_scratch.argtypes = [i64, i32, i32]
        L.csl_gat_in_bwd_scratch.restype = i64
        L.csl_gat_in_bwd_f32.argtypes = [vp, vp, vp, vp, vp, i64, i32, vp, vp, i64, i64, i32, f32, i64, i64, i32, vp, vp, vp, vp]
        L.csl_bias_elu_f32.argtypes = [vp, i64, vp, i64, i32, i32, vp]
        L.csl_elu_bwd_colsum_scratch.argtypes = [i64, i32]
        L.csl_elu_bwd_colsum_scratch.restype = i64
        L.csl_elu_bwd_colsum_f32.argtypes = [vp, i64, vp, i64, i64, i32, i32, vp, i64, vp, vp, vp]
        L.csl_gat_in_proj_ok.argtypes = [i32, i32, i32]
        L.csl_gat_in_proj_fpad.argtypes = [i32]
        L.csl_gat_in_proj_f32.argtypes = [vp, vp, vp, i64, i32, i32, i32, i32, vp, i64, vp]
        L.csl_gat_in_proj_bwd_scratch.argtypes = [i32, i32, i32]
        L.csl_gat_in_proj_bwd_scratch.restype = i64
        L.csl_gat_in_proj_bwd_f32.argtypes = [vp, i64, vp, vp, i64, i32, i32, i32, vp, vp, vp, vp]
        L.csl_gat_in_layer_fwd_scratch.argtypes = [i32, i32]
        L.csl_gat_in_layer_fwd_scratch.restype = i64
        L.csl_gat_in_layer_fwd_f32.argtypes = [vp, vp, vp, vp, vp, i64, i32, vp, vp, vp, vp, i32, i32, f32, i32, i64, i64, i32,
                                               vp, vp, vp, i64, vp, vp]
        L.csl_gat_in_layer_bwd_scratch.argtypes = [i64, i32, i32, i32]
        L.csl_gat_in_layer_bwd_scratch.restype = i64
        L.csl_gat_in_layer_bwd_f32.argtypes = [vp, vp, vp, vp, vp, i64, i32, vp, vp, vp, i32, i32, f32, i32, i64, i64, i32,
                                               vp, vp, vp, i64, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp]
        _ready = True
    return L


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None and t.numel() else C.c_void_p(0)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """hipStream_t of torch's current stream.  The raw getter avoids building a torch.cuda.Stream object per
    kernel call (a training step makes ~25 of them: 0.13 ms of host time per step in the profile)."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(rc, what):
    if rc < 0:
        raise _abi.CslError(rc, what + " failed")


def _f32(x):
    if x.dtype != torch.float32 or not x.is_cuda:
        raise TypeError("expected a float32 CUDA tensor")
    return x if x.stride(-1) == 1 else x.contiguous()


def _i32(x):
    if x.dtype != torch.int32 or not x.is_cuda:
        raise TypeError("expected an int32 CUDA tensor (the slices' device index type)")
    return x.contiguous()


def spmm_sum(indptr, indices, x, n_rows, rows=None, out=None):
    """out[row] = sum of x[indices[e]] over the row's edges (BipartiteGraph.gather, sum form).
    rows: optional int32 list of the rows to compute (others untouched)."""
    x = _f32(x)
    H = x.shape[1]
    if out is None:
        out = torch.empty((n_rows, H), dtype=torch.float32, device=x.device)
        if rows is not None:
            out.zero_()
    n = n_rows if rows is None else rows.numel()
    _chk(_lib().csl_spmm_sum_f32(_p(_i32(indptr)), _p(_i32(indices)), _p(rows) if rows is not None else C.c_void_p(0),
                                 n, _p(x), x.stride(0), _p(out), out.stride(0), H, _stream()), "csl_spmm_sum_f32")
    return out


def spmm_sum_compact(indptr, indices, x, rows):
    """[len(rows), H]: the k-th row is the sum-aggregate of CSR row rows[k] (a send buffer of boundary partial sums)."""
    x = _f32(x)
    out = torch.empty((rows.numel(), x.shape[1]), dtype=torch.float32, device=x.device)
    _chk(_lib().csl_spmm_sum_compact_f32(_p(_i32(indptr)), _p(_i32(indices)), _p(_i32(rows)), rows.numel(), _p(x),
                                         x.stride(0), _p(out), out.stride(0), x.shape[1], _stream()),
         "csl_spmm_sum_compact_f32")
    return out


def sage_cat_rows_bwd(self_ids, owned, deg, gcat, n, n_x, n_agg, want_x=True, want_agg=True):
    """Gradients of sage_cat's merged-sums form: (gx [n_x, H] or None, gagg [n_agg, H] or None), zero fill inside."""
    gcat = _f32(gcat)
    H = gcat.shape[1] // 2
    gx = torch.empty((n_x, H), dtype=torch.float32, device=gcat.device) if want_x else None
    gagg = torch.empty((n_agg, H), dtype=torch.float32, device=gcat.device) if want_agg else None
    _chk(_lib().csl_sage_cat_rows_bwd_f32(_p(_i32(self_ids)), _p(_i32(owned)), _p(_i32(deg)), n, _p(gcat), gcat.stride(0),
                                          _p(gx), n_x, _p(gagg), n_agg, H, _stream()), "csl_sage_cat_rows_bwd_f32")
    return gx, gagg


def spmm_sum_bwd(indptr, indices, grad_out, n_src, rows=None, compact=False, out=None):
    """grad wrt the sources of spmm_sum.  rows: only these output rows contribute; compact: grad_out
    holds just those rows (k-th row of grad_out belongs to rows[k]).  Accumulates into `out` if given."""
    g = _f32(grad_out)
    gx = out if out is not None else torch.zeros((n_src, g.shape[1]), dtype=torch.float32, device=g.device)
    n = g.shape[0] if rows is None else rows.numel()
    _chk(_lib().csl_spmm_sum_bwd_f32(_p(_i32(indptr)), _p(_i32(indices)),
                                     _p(rows) if rows is not None else C.c_void_p(0), n, _p(g), g.stride(0),
                                     1 if compact else 0, _p(gx), gx.stride(0), g.shape[1], _stream()),
         "csl_spmm_sum_bwd_f32")
    return gx


def gather_rows(src, idx, out=None):
    """dst[k] = src[idx[k]] (zero row for idx -1): pull_for_remotes / self_gather.
    out: optional destination, [len(idx), H] with unit column stride (rows may be strided: a column
    block of a wider matrix)."""
    src = _f32(src)
    idx = _i32(idx)
    if out is None:
        dst = torch.empty((idx.numel(), src.shape[1]), dtype=torch.float32, device=src.device)
    else:
        dst = out
        if dst.shape != (idx.numel(), src.shape[1]) or dst.stride(-1) != 1 or dst.dtype != torch.float32:
            raise ValueError("out must be float32 [len(idx), H] with unit column stride")
    _chk(_lib().csl_gather_rows_f32(_p(src), src.stride(0), _p(idx), idx.numel(), _p(dst), dst.stride(0),
                                    src.shape[1], _stream()), "csl_gather_rows_f32")
    return dst


def scatter_add_rows_(dst, idx, src):
    """dst[idx[k]] += src[k] in place (idx unique): push_from_remotes / mergeKernel."""
    src = _f32(src)
    idx = _i32(idx)
    if dst.stride(-1) != 1:
        raise ValueError("dst must be row-contiguous")
    _chk(_lib().csl_scatter_add_rows_f32(_p(dst), dst.stride(0), _p(idx), idx.numel(), _p(src), src.stride(0),
                                         src.shape[1], _stream()), "csl_scatter_add_rows_f32")
    return dst


def scatter_add_rows_atomic_(dst, idx, src):
    """dst[idx[k]] += src[k] in place where idx may repeat (all peers' partial sums in one launch; fp32 atomics)."""
    src = _f32(src)
    idx = _i32(idx)
    if dst.stride(-1) != 1:
        raise ValueError("dst must be row-contiguous")
    _chk(_lib().csl_scatter_add_rows_atomic_f32(_p(dst), dst.stride(0), _p(idx), idx.numel(), _p(src), src.stride(0),
                                                src.shape[1], _stream()), "csl_scatter_add_rows_atomic_f32")
    return dst


def div_rows_(x, deg):
    """x[k] /= max(deg[k], 1) in place."""
    _chk(_lib().csl_div_rows_f32(_p(x), x.stride(0), _p(_i32(deg)), x.shape[0], x.shape[1], _stream()),
         "csl_div_rows_f32")
    return x


def sage_cat(x, self_ids, n, n_pad, indptr=None, indices=None, owned=None, deg=None, agg=None, rowmap=None,
             relu_in=False):
    """The operand of Linear(2*in, out) in one pass (csl_sage_cat_f32): [n_pad, 2*H] with
    cat[:, :H] = x[map(self_ids)], cat[:, H:] = mean over the CSR row of x[map(indices)] (indptr given) or
    agg[owned] / deg (indptr None).  Rows n..n_pad are zero."""
    x = _f32(x)
    H = x.shape[1]
    cat = torch.empty((n_pad, 2 * H), dtype=torch.float32, device=x.device)
    nul = C.c_void_p(0)
    if indptr is not None:
        args = (_p(_i32(indptr)), _p(_i32(indices)), _p(_i32(self_ids)), nul, _p(deg) if deg is not None else nul,
                _p(rowmap) if rowmap is not None else nul, _p(x), x.stride(0), nul, 0)
    else:
        agg = _f32(agg)
        args = (nul, nul, _p(_i32(self_ids)), _p(_i32(owned)), _p(_i32(deg)),
                _p(rowmap) if rowmap is not None else nul, _p(x), x.stride(0), _p(agg), agg.stride(0))
    _chk(_lib().csl_sage_cat_f32(*args, n, n_pad, _p(cat), cat.stride(0), H, 1 if relu_in else 0, _stream()),
         "csl_sage_cat_f32")
    return cat


def sage_fwd_mfma(x, self_ids, indptr, indices, weight, bias, n, n_pad, rowmap=None, relu_in=False, relu_out=False,
                  want_cat=False):
    """A GraphSAGE layer's forward as one kernel on the fp32 matrix cores (csl_sage_fwd_mfma_f32):
    y [n_pad, out] = act([x[map(self_ids)] | mean over the CSR row of x[map(indices)]] weight^T + bias); with want_cat
    also the operand [n_pad, 2 H] (what sage_cat returns).  weight [out, 2 H] (torch's Linear.weight)."""
    x, weight = _f32(x), _f32(weight)
    H, out = x.shape[1], weight.shape[0]
    L = _lib()
    nw = L.csl_sage_fwd_mfma_scratch(H, out)
    if nw < 0:
        raise ValueError("csl_sage_fwd_mfma_f32: unsupported widths (in %% 4 == 0, out <= 256): in %d, out %d" % (H, out))
    wpack = torch.empty((nw,), dtype=torch.float32, device=x.device)
    y = torch.empty((n_pad, out), dtype=torch.float32, device=x.device)
    cat = torch.empty((n_pad, 2 * H), dtype=torch.float32, device=x.device) if want_cat else None
    nul = C.c_void_p(0)
    _chk(L.csl_sage_fwd_mfma_f32(_p(_i32(indptr)), _p(_i32(indices)), _p(_i32(self_ids)),
                                 _p(rowmap) if rowmap is not None else nul, _p(x), x.stride(0), _p(weight),
                                 weight.stride(0), _p(_f32(bias)) if bias is not None else nul, n, n_pad, H, out,
                                 1 if relu_in else 0, 1 if relu_out else 0, _p(cat) if want_cat else nul,
                                 cat.stride(0) if want_cat else 0, _p(y), y.stride(0), _p(wpack), _stream()),
         "csl_sage_fwd_mfma_f32")
    return (y, cat) if want_cat else y


def sage_cat_bwd(indptr, indices, self_ids, gcat, n, n_src):
    """Gradient of sage_cat's CSR form w.r.t. x: a fresh [n_src, H] (zeroed inside the call, fp32 atomics)."""
    gcat = _f32(gcat)
    H = gcat.shape[1] // 2
    gx = torch.empty((n_src, H), dtype=torch.float32, device=gcat.device)
    _chk(_lib().csl_sage_cat_bwd_f32(_p(_i32(indptr)), _p(_i32(indices)), _p(_i32(self_ids)), n, _p(gcat),
                                     gcat.stride(0), _p(gx), gx.stride(0), n_src, H, _stream()),
         "csl_sage_cat_bwd_f32")
    return gx


def relu_bwd_colsum(g, y, n, n_pad):
    """(out [n_pad, H], colsum [H]): out[:n] = g masked by y > 0 (y None: unmasked), zero pad rows, column sums."""
    g = _f32(g)
    H = g.shape[1]
    out = torch.empty((n_pad, H), dtype=torch.float32, device=g.device)
    L = _lib()
    # colsum and the per-block partial sums share one allocation
    buf = torch.empty((H + max(int(L.csl_relu_bwd_colsum_scratch(n_pad, H)), 1),), dtype=torch.float32, device=g.device)
    _chk(L.csl_relu_bwd_colsum_f32(_p(g), g.stride(0), _p(y) if y is not None else C.c_void_p(0),
                                   y.stride(0) if y is not None else 0, n, n_pad, _p(out), out.stride(0),
                                   C.c_void_p(buf.data_ptr()), C.c_void_p(buf.data_ptr() + 4 * H), H, _stream()),
         "csl_relu_bwd_colsum_f32")
    return out, buf[:H]


def gemm(a, b, transa=False, transb=False, bias=None, relu=False):
    """op(a) @ op(b) (+ bias over the columns) (ReLU) for row-major fp32 matrices (rows may be strided), as one
    direct hipBLASLt call with a cached, timed plan per shape (csl_gemm_f32): ~5 us of host time per GEMM."""
    m, k = (a.shape[1], a.shape[0]) if transa else (a.shape[0], a.shape[1])
    n = b.shape[0] if transb else b.shape[1]
    if (b.shape[1] if transb else b.shape[0]) != k or a.stride(1) != 1 or b.stride(1) != 1:
        raise ValueError("gemm: inner dimensions differ or a column stride is not 1")
    out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    if m and n:
        L = _lib()
        rc = L.csl_gemm_f32(int(transa), int(transb), m, n, k, _p(a), a.stride(0), 0, _p(b), b.stride(0), 0,
                            _p(out), n, 0, 1, _p(bias) if bias is not None else C.c_void_p(0), int(relu), _stream())
        if rc < 0:
            raise _abi.CslError(rc, "csl_gemm_f32: " + L.csl_gemm_last_error().decode())
    return out


GEMM_PLANS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_plans_gfx950.txt")


def gemm_load_plans(path=None):
    """Recorded csl_gemm_f32 plans (solution index per shape class; profiles/tune_direct_gemms.sh made the shipped
    file with CSLICER_GEMM_TUNE=all): shapes they cover are planned without timing.  Returns the number of entries
    (0: no file).  CSLICER_GEMM_PLANS=<file> overrides the shipped file, CSLICER_GEMM_PLANS=0 disables it."""
    env = os.environ.get("CSLICER_GEMM_PLANS")
    if env == "0":
        return 0
    path = path or env or GEMM_PLANS
    if not os.path.exists(path):
        return 0
    return max(int(_lib().csl_gemm_load_plans(path.encode())), 0)


def gemm_save_plans(path):
    _chk(_lib().csl_gemm_save_plans(path.encode()), "csl_gemm_save_plans")


def weight_grad_slabs(gy, x, n_slabs):
    """gy^T @ x for tall row-major gy [rows, out], x [rows, in] (rows % n_slabs == 0): n_slabs independent row slabs
    as one batched GEMM, then their sum -- the 10^5-long reduction of a 256 x 200 result otherwise runs on a few
    dozen workgroups of the chip."""
    rows, out_f = gy.shape
    in_f = x.shape[1]
    rs = rows // n_slabs
    slabs = torch.empty((n_slabs + 1, out_f, in_f), dtype=torch.float32, device=gy.device)   # [n_slabs] = the sum
    L = _lib()
    rc = L.csl_gemm_f32(1, 0, out_f, in_f, rs, _p(gy), gy.stride(0), rs * gy.stride(0), _p(x), x.stride(0),
                        rs * x.stride(0), _p(slabs), in_f, out_f * in_f, n_slabs, C.c_void_p(0), 0, _stream())
    if rc < 0:
        raise _abi.CslError(rc, "csl_gemm_f32: " + L.csl_gemm_last_error().decode())
    _chk(L.csl_sum_slabs_f32(_p(slabs), out_f * in_f, n_slabs, C.c_void_p(slabs[n_slabs].data_ptr()), _stream()),
         "csl_sum_slabs_f32")
    return slabs[n_slabs]


def sage_cat_bwd_t(t_indptr, t_indices, indptr, gcat, y, n_src, n_pad, hub=False):
    """Gradient of sage_cat's CSR form w.r.t. x as a gather over the slice by source (engine flag FLAG_TRANSPOSE),
    masked by y > 0 (y None: unmasked), rows padded with zeros to n_pad; returns (out [n_pad, H], column sums [H]).
    hub: the slice has lists longer than T_SORTED_MAX entries (Slice.t_max_len): those rows are summed by many
    workgroups (csl_sage_cat_bwd_t_hub_f32)."""
    gcat = _f32(gcat)
    H = gcat.shape[1] // 2
    out = torch.empty((n_pad, H), dtype=torch.float32, device=gcat.device)
    L = _lib()
    nul = C.c_void_p(0)
    if hub:
        buf = torch.empty((H + max(int(L.csl_sage_cat_bwd_t_hub_scratch(n_pad, H)), 1),), dtype=torch.float32,
                          device=gcat.device)
        _chk(L.csl_sage_cat_bwd_t_hub_f32(_p(_i32(t_indptr)), _p(_i32(t_indices)), t_indices.numel(), _p(_i32(indptr)),
                                          _p(gcat), gcat.stride(0), _p(y) if y is not None else nul,
                                          y.stride(0) if y is not None else 0, n_src, n_pad, _p(out), out.stride(0),
                                          C.c_void_p(buf.data_ptr()), C.c_void_p(buf.data_ptr() + 4 * H), H, _stream()),
             "csl_sage_cat_bwd_t_hub_f32")
        return out, buf[:H]
    buf = torch.empty((H + max(int(L.csl_sage_cat_bwd_t_scratch(n_pad, H)), 1),), dtype=torch.float32,
                      device=gcat.device)
    _chk(L.csl_sage_cat_bwd_t_f32(_p(_i32(t_indptr)), _p(_i32(t_indices)), _p(_i32(indptr)), _p(gcat), gcat.stride(0),
                                  _p(y) if y is not None else nul, y.stride(0) if y is not None else 0,
                                  n_src, n_pad, _p(out), out.stride(0), C.c_void_p(buf.data_ptr()),
                                  C.c_void_p(buf.data_ptr() + 4 * H), H, _stream()), "csl_sage_cat_bwd_t_f32")
    return out, buf[:H]


class SageSlice(C.Structure):
    """csl_sage_slice (cslicer_aggr.h)"""
    _fields_ = [("indptr", C.c_void_p), ("indices", C.c_void_p), ("self_ids_in", C.c_void_p),
                ("t_indptr", C.c_void_p), ("t_indices", C.c_void_p), ("n_out", C.c_int64), ("n_in", C.c_int64),
                ("t_max_len", C.c_int64), ("t_entries", C.c_int64)]


STEP_GROUPS = ("fused_forward", "gemm", "aggregation", "other")


def step_timing(enable):
    """switch the per-group HIP-event timing of csl_sage_fwd_bwd_f32 on or off (cslicer_aggr.h)"""
    _chk(_lib().csl_sage_step_timing(1 if enable else 0), "csl_sage_step_timing")


def step_timing_read():
    """{group: (milliseconds, launches)} accumulated since the last read (waits for the recorded launches)"""
    ms = (C.c_double * len(STEP_GROUPS))()
    n = (C.c_int64 * len(STEP_GROUPS))()
    _chk(_lib().csl_sage_step_timing_read(ms, n), "csl_sage_step_timing_read")
    return {g: (float(ms[i]), int(n[i])) for i, g in enumerate(STEP_GROUPS)}


class SageStep(object):
    """Forward + cross-entropy + backward of a DistSAGEModel on one part as ONE native call per minibatch
    (csl_sage_fwd_bwd_f32): the gradients of W_0, b_0, W_1, ... land back to back in `self.grads` (what
    aggr.Adam.step(flat_grads=...) takes), the loss in the slot the caller names.  The parameters are used in place
    (their storage must not move: an optimizer that updates in place, as aggr.Adam does)."""

    def __init__(self, model, row_pad, n_slabs):
        ws, bs = [c.fc.weight for c in model.convs], [c.fc.bias for c in model.convs]
        self.L = len(ws)
        self.dims = [ws[0].shape[1] // 2] + [w.shape[0] for w in ws]
        for k, (w, b) in enumerate(zip(ws, bs)):
            if (w.dtype != torch.float32 or not w.is_cuda or not w.is_contiguous() or not b.is_contiguous()
                    or w.shape[1] != 2 * self.dims[k]):
                raise TypeError("contiguous float32 CUDA Linear(2*in, out) layers expected")
        self._params = ws + bs                      # kept alive: the pointer tables below alias them
        self._dims = (C.c_int32 * (self.L + 1))(*self.dims)
        self._w = (C.c_void_p * self.L)(*[w.data_ptr() for w in ws])
        self._b = (C.c_void_p * self.L)(*[b.data_ptr() for b in bs])
        self._sl = (SageSlice * self.L)()
        self.row_pad, self.n_slabs = int(row_pad), int(n_slabs)
        n_grad = sum(w.numel() + b.numel() for w, b in zip(ws, bs))
        self.grads = torch.empty((n_grad,), dtype=torch.float32, device=ws[0].device)
        self._ws = None
        gemm_load_plans()

    def __call__(self, slices, feat, labels, scale, loss_out):
        """slices: the part's `splitgnn.Slice`s in MODEL order (deepest hop first), from an engine with
        FLAG_TRANSPOSE; feat: resident [N, F] features; labels int64 [N]; loss_out: one-element float32 tensor."""
        A = _abi
        for k, s in enumerate(slices):
            c = self._sl[k]
            c.indptr, c.indices, c.self_ids_in = s.ptr(A.INDPTR), s.ptr(A.INDICES), s.ptr(A.SELF_IDS_IN)
            # (an EMPTY layer -- a rank's empty share of a short minibatch -- has no row pointers at all)
            if k and s.count(A.T_INDPTR) != s.n_in + 1 and (s.n_in or s.n_out or s.count(A.T_INDPTR)):
                raise ValueError("layer %d has no slice by source: create the engine with flags=FLAG_TRANSPOSE" % k)
            c.t_indptr, c.t_indices = (s.ptr(A.T_INDPTR), s.ptr(A.T_INDICES)) if k else (None, None)
            c.n_out, c.n_in, c.t_max_len = s.n_out, s.n_in, s.t_max_len
            c.t_entries = s.count(A.T_INDICES) if k else 0
        L = _lib()
        need = L.csl_sage_fwd_bwd_workspace(self.L, self._dims, self._sl, self.row_pad, self.n_slabs)
        if need < 0:
            raise _abi.CslError(need, "csl_sage_fwd_bwd_workspace: unsupported model or slices")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty((int(need * 1.25) + 1024,), dtype=torch.float32, device=feat.device)
        rc = L.csl_sage_fwd_bwd_f32(self.L, self._dims, self._sl, self._w, self._b, feat.data_ptr(), feat.stride(0),
                                    slices[0].ptr(A.IN_NODES), slices[-1].ptr(A.OUT_NODES), labels.data_ptr(),
                                    float(scale), self.row_pad, self.n_slabs, self.grads.data_ptr(),
                                    loss_out.data_ptr(), self._ws.data_ptr(), self._ws.numel(), _stream())
        if rc < 0:
            raise _abi.CslError(rc, "csl_sage_fwd_bwd_f32: " + L.csl_sage_last_error().decode() + " / " +
                                L.csl_gemm_last_error().decode())


class SageRankSlice(C.Structure):
    """csl_sage_rank_slice (cslicer_aggr.h)"""
    _fields_ = [("indptr", C.c_void_p), ("indices", C.c_void_p), ("self_ids_in", C.c_void_p),
                ("owned_out_nodes", C.c_void_p), ("owned_degree", C.c_void_p), ("from_all", C.c_void_p),
                ("to_all", C.c_void_p), ("n_out", C.c_int64), ("n_in", C.c_int64), ("n_owned", C.c_int64),
                ("n_from", C.c_int64), ("n_to", C.c_int64), ("t_indptr", C.c_void_p), ("t_indices", C.c_void_p),
                ("t_max_len", C.c_int64), ("t_entries", C.c_int64)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p)
EXCHANGE_WAIT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p)


class SageRankStep(object):
    """One rank's forward + loss + backward of the SPLIT-parallel GraphSAGE step as one native call
    (csl_sage_rank_fwd_bwd_f32); the 2 L - 1 boundary exchanges come back as callbacks and go through `comm`
    (splitgnn.DistComm: all_to_all_single over RCCL, gloo in the tests).  Gradients of this rank's share land in
    `self.grads` (W_0, b_0, ...) for the caller's all-reduce and aggr.Adam.step(flat_grads=...)."""

    def __init__(self, model, row_pad, n_slabs, comm, overlap=False):
        ws, bs = [c.fc.weight for c in model.convs], [c.fc.bias for c in model.convs]
        self.L = len(ws)
        self.dims = [ws[0].shape[1] // 2] + [w.shape[0] for w in ws]
        self._params = ws + bs
        # overlap: every exchange runs on the communicator's side stream while the rows that stay on this GPU are
        # aggregated (dist_sageconv.py:57-64); same numbers, different schedule
        self.overlap = bool(overlap)
        self._done = {}
        self._cb_wait = EXCHANGE_WAIT_FN(self._wait)
        self._dims = (C.c_int32 * (self.L + 1))(*self.dims)
        self._w = (C.c_void_p * self.L)(*[w.data_ptr() for w in ws])
        self._b = (C.c_void_p * self.L)(*[b.data_ptr() for b in bs])
        self._sl = (SageRankSlice * self.L)()
        self.row_pad, self.n_slabs, self.comm = int(row_pad), int(n_slabs), comm
        n_grad = sum(w.numel() + b.numel() for w, b in zip(ws, bs))
        self.grads = torch.empty((n_grad,), dtype=torch.float32, device=ws[0].device)
        self._ws = None
        self._cur = None
        self._exc = None
        self._cb = EXCHANGE_FN(self._exchange)     # (kept alive: the native call holds only the raw pointer)
        gemm_load_plans()

    def _view(self, ptr, rows, width):
        off = (int(ptr or 0) - self._ws.data_ptr()) // 4 if rows else 0
        return self._ws[off:off + rows * width].view(rows, width)

    def _exchange(self, user, layer, backward, src, dst, width, stream):
        try:
            s = self._cur[layer]
            send, recv = (s.to_counts, s.from_counts) if backward else (s.from_counts, s.to_counts)
            out, inp = self._view(dst, sum(recv), width), self._view(src, sum(send), width)
            # (a first real multi-GPU run must fail loudly rather than exchange misaligned rows: the per-peer counts
            # come from the sample's pair offsets, the list lengths from its list offsets, the peers from the group)
            n_from, n_to = s.count(_abi.FROM_IDS), s.count(_abi.TO_IDS)
            n_send, n_recv = (n_to, n_from) if backward else (n_from, n_to)
            world = getattr(self.comm, "world", len(send))
            if (sum(send) != n_send or sum(recv) != n_recv or len(send) != world or len(recv) != world
                    or min(list(send) + list(recv) + [0]) < 0 or send[s.part] != 0 or recv[s.part] != 0):
                raise RuntimeError("layer %d %s exchange of part %d: per-peer counts %r / %r against boundary lists of %d "
                                   "and %d rows, %d peers" % (layer, "backward" if backward else "forward", s.part,
                                                              list(send), list(recv), n_send, n_recv, world))
            if not self.overlap:
                self.comm.exchange_into(out, inp, send, recv)
                return 0
            main, side = torch.cuda.current_stream(), self.comm.side_stream()
            ready = torch.cuda.Event()
            ready.record(main)
            with torch.cuda.stream(side):
                side.wait_event(ready)
                self.comm.exchange_into(out, inp, send, recv)
                done = torch.cuda.Event()
                done.record(side)
            self._done[(layer, backward)] = done
            return 0
        except Exception as ex:      # an exception cannot cross the C frame: reported after the native call returns
            self._exc = ex
            return -1

    def _wait(self, user, layer, backward, stream):
        try:
            torch.cuda.current_stream().wait_event(self._done.pop((layer, backward)))
            return 0
        except Exception as ex:
            self._exc = ex
            return -1

    def __call__(self, slices, feat, feat_rows, seed_ids, label_rows, labels, scale, loss_out):
        """slices: this part's `splitgnn.Slice`s in MODEL order; feat: the rank's resident feature rows; feat_rows
        int32 [n_in of the deepest slice]: their local rows; seed_ids int32: global ids of the owned seeds."""
        A = _abi
        for k, s in enumerate(slices):
            c = self._sl[k]
            c.indptr, c.indices, c.self_ids_in = s.ptr(A.INDPTR), s.ptr(A.INDICES), s.ptr(A.SELF_IDS_IN)
            c.owned_out_nodes, c.owned_degree = s.ptr(A.OWNED_OUT_NODES), s.ptr(A.OWNED_DEGREE)
            c.from_all, c.to_all = s.ptr(A.FROM_IDS), s.ptr(A.TO_IDS)
            c.n_out, c.n_in, c.n_owned = s.n_out, s.n_in, s.n_owned
            c.n_from, c.n_to = sum(s.from_counts), sum(s.to_counts)
            # the part's slice by source, where the engine emitted it (FLAG_TRANSPOSE): the backward gathers over it
            if k and s.count(A.T_INDPTR) == s.n_in + 1 and s.n_in > 0:
                c.t_indptr, c.t_indices = s.ptr(A.T_INDPTR), s.ptr(A.T_INDICES)
                c.t_max_len, c.t_entries = s.t_max_len, s.count(A.T_INDICES)
            else:
                c.t_indptr, c.t_indices, c.t_max_len, c.t_entries = None, None, 0, 0
        L = _lib()
        need = L.csl_sage_rank_workspace(self.L, self._dims, self._sl, self.row_pad, self.n_slabs)
        if need < 0:
            raise _abi.CslError(need, "csl_sage_rank_workspace: unsupported model or slices")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty((int(need * 1.25) + 1024,), dtype=torch.float32, device=feat.device)
        self._cur, self._exc = slices, None
        rc = L.csl_sage_rank_fwd_bwd_f32(self.L, self._dims, self._sl, self._w, self._b, feat.data_ptr(), feat.stride(0),
                                         _p(feat_rows), _p(seed_ids), _p(label_rows) if label_rows is not None else None,
                                         labels.data_ptr(), float(scale), self.row_pad, self.n_slabs, self._cb,
                                         self._cb_wait if self.overlap else EXCHANGE_WAIT_FN(0), None,
                                         self.grads.data_ptr(), loss_out.data_ptr(), self._ws.data_ptr(),
                                         self._ws.numel(), _stream())
        self._cur = None
        if self._exc is not None:
            raise self._exc
        if rc < 0:
            raise _abi.CslError(rc, "csl_sage_rank_fwd_bwd_f32: " + L.csl_sage_last_error().decode() + " / " +
                                L.csl_gemm_last_error().decode())


class SoftmaxCE(torch.autograd.Function):
    """Cross-entropy summed over the rows and scaled (python/train.py:86), forward and backward in one HIP pass:
    label of row r = labels[rowmap[ids[r]]] (rowmap None: labels[ids[r]]); returns the scalar loss."""

    @staticmethod
    def forward(ctx, logits, ids, labels, scale, rowmap=None):
        logits = _f32(logits)
        n, Cn = logits.shape
        L = _lib()
        buf = torch.empty((1 + max(int(L.csl_softmax_ce_scratch(n)), 1),), dtype=torch.float32, device=logits.device)
        grad = torch.empty((n, Cn), dtype=torch.float32, device=logits.device)
        if labels.dtype != torch.int64:
            raise TypeError("labels must be int64")
        _chk(L.csl_softmax_ce_f32(_p(logits), logits.stride(0), n, Cn, _p(_i32(ids)),
                                  _p(rowmap) if rowmap is not None else C.c_void_p(0), _p(labels), float(scale),
                                  C.c_void_p(buf.data_ptr()), _p(grad), grad.stride(0),
                                  C.c_void_p(buf.data_ptr() + 4), _stream()), "csl_softmax_ce_f32")
        ctx.save_for_backward(grad)
        return buf[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None, None


class SpmmSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, indptr, indices, n_rows):
        ctx.save_for_backward(indptr, indices)
        ctx.n_src = x.shape[0]
        return spmm_sum(indptr, indices, x, n_rows)

    @staticmethod
    def backward(ctx, g):
        indptr, indices = ctx.saved_tensors
        return spmm_sum_bwd(indptr, indices, g.contiguous(), ctx.n_src), None, None, None


class GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, idx):
        ctx.save_for_backward(idx)
        ctx.n_src = src.shape[0]
        return gather_rows(src, idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        gs = torch.zeros((ctx.n_src, g.shape[1]), dtype=torch.float32, device=g.device)
        scatter_add_rows_(gs, idx, g.contiguous())
        return gs, None


class ScatterAddRows(torch.autograd.Function):
    """out = dst with src rows added at idx.  dst is updated in place (marked dirty)."""

    @staticmethod
    def forward(ctx, dst, idx, src):
        ctx.save_for_backward(idx)
        ctx.mark_dirty(dst)
        scatter_add_rows_(dst, idx, src)
        return dst

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        g = g.contiguous()
        return g, None, gather_rows(g, idx)


class GatherMeanRows(torch.autograd.Function):
    """dst[k] = src[idx[k]] / max(deg[k], 1): slice_owned_nodes + mean normalisation in two HIP
    launches (row gather, in-place degree division) instead of a chain of elementwise torch ops."""

    @staticmethod
    def forward(ctx, src, idx, deg):
        ctx.save_for_backward(idx, deg)
        ctx.n_src = src.shape[0]
        return div_rows_(gather_rows(src, idx), deg)

    @staticmethod
    def backward(ctx, g):
        idx, deg = ctx.saved_tensors
        g = div_rows_(g.contiguous().clone(), deg)
        gs = torch.zeros((ctx.n_src, g.shape[1]), dtype=torch.float32, device=g.device)
        scatter_add_rows_(gs, idx, g)
        return gs, None, None


class GatAggregate(torch.autograd.Function):
    """Partial edge-softmax aggregation of one slice (csl_gat_fwd_f32 / csl_gat_bwd_f32): returns
    (m [rows,H], s [rows,H], n [rows,H*D]); m is a stabiliser and is not differentiated (the merged
    result n/s does not depend on it)."""

    @staticmethod
    def forward(ctx, el, er, z, indptr, indices, n_rows, H, D, slope):
        el, er, z = _f32(el).contiguous(), _f32(er).contiguous(), _f32(z).contiguous()
        indptr, indices = _i32(indptr), _i32(indices)
        if el.shape != (z.shape[0], H) or er.shape != (n_rows, H) or z.shape[1] != H * D:
            raise ValueError("el [n_src,H], er [n_rows,H], z [n_src,H*D] expected")
        dev = z.device
        m = torch.empty((n_rows, H), dtype=torch.float32, device=dev)
        s = torch.empty((n_rows, H), dtype=torch.float32, device=dev)
        n = torch.empty((n_rows, H * D), dtype=torch.float32, device=dev)
        _chk(_lib().csl_gat_fwd_f32(_p(indptr), _p(indices), n_rows, _p(el), _p(er), _p(z), H, D, slope,
                                    _p(m), _p(s), _p(n), _stream()), "csl_gat_fwd_f32")
        ctx.save_for_backward(el, er, z, indptr, indices, m)
        ctx.cfg = (n_rows, H, D, slope)
        ctx.mark_non_differentiable(m)
        return m, s, n

    @staticmethod
    def backward(ctx, _gm, gs, gn):
        el, er, z, indptr, indices, m = ctx.saved_tensors
        n_rows, H, D, slope = ctx.cfg
        gs, gn = gs.contiguous(), gn.contiguous()
        g_el = torch.zeros_like(el)
        g_er = torch.empty_like(er)
        g_z = torch.zeros_like(z)
        _chk(_lib().csl_gat_bwd_f32(_p(indptr), _p(indices), n_rows, _p(el), _p(er), _p(z), H, D, slope, _p(m),
                                    _p(gs), _p(gn), _p(g_el), _p(g_er), _p(g_z), _stream()), "csl_gat_bwd_f32")
        return g_el, g_er, g_z, None, None, None, None, None, None


class GatLogits(torch.autograd.Function):
    """(el, er) [n, H] = <z[:, h, :], attn_l[h, :]>, <z[:, h, :], attn_r[h, :]> for z [n, H*D]: one row-wise HIP pass
    over z (csl_gat_logits_fwd_f32 / _bwd_f32) instead of two [n, in] x [in, H] library GEMMs with H = 8 columns."""

    @staticmethod
    def forward(ctx, z, attn_l, attn_r):
        z, al, ar = _f32(z).contiguous(), _f32(attn_l).contiguous(), _f32(attn_r).contiguous()
        H, D = al.shape
        n = z.shape[0]
        if z.shape[1] != H * D or ar.shape != al.shape:
            raise ValueError("z [n, H*D], attn_l / attn_r [H, D] expected")
        el = torch.empty((n, H), dtype=torch.float32, device=z.device)
        er = torch.empty((n, H), dtype=torch.float32, device=z.device)
        _chk(_lib().csl_gat_logits_fwd_f32(_p(z), _p(al), _p(ar), n, H, D, _p(el), _p(er), _stream()),
             "csl_gat_logits_fwd_f32")
        ctx.save_for_backward(z, al, ar)
        return el, er

    @staticmethod
    def backward(ctx, g_el, g_er):
        z, al, ar = ctx.saved_tensors
        H, D = al.shape
        n = z.shape[0]
        g_el, g_er = _f32(g_el).contiguous(), _f32(g_er).contiguous()
        g_z = torch.empty_like(z)
        L = _lib()
        buf = torch.empty((2 * H * D + max(int(L.csl_gat_logits_bwd_scratch(n, H, D)), 4),), dtype=torch.float32,
                          device=z.device)
        _chk(L.csl_gat_logits_bwd_f32(_p(z), _p(al), _p(ar), _p(g_el), _p(g_er), n, H, D, _p(g_z),
                                      C.c_void_p(buf.data_ptr()), C.c_void_p(buf.data_ptr() + 4 * H * D),
                                      C.c_void_p(buf.data_ptr() + 8 * H * D), _stream()), "csl_gat_logits_bwd_f32")
        return g_z, buf[:H * D].view(H, D), buf[H * D:2 * H * D].view(H, D)


class PaddedRows(object):
    """The first `n` rows of `t` [mp, width]: mp = n rounded up so that GEMM shapes repeat, rows n.. are zero.  What the
    fused GAT layers hand to each other, so that no layer copies its input into a padded buffer again."""
    __slots__ = ("t", "n")

    def __init__(self, t, n):
        self.t, self.n = t, int(n)


def padded_rows(n, width, row_pad, device):
    """an uninitialised PaddedRows buffer whose padding rows are zeroed"""
    mp = (n + row_pad - 1) // row_pad * row_pad if row_pad and n >= row_pad else n
    t = torch.empty((mp, width), dtype=torch.float32, device=device)
    if mp != n:
        t[n:].zero_()
    return PaddedRows(t, n)


class GatLayerLocal(torch.autograd.Function):
    """A whole DistGATConv layer (+ ELU) of a single part (nothing to exchange) as ONE autograd node:
    z = x W^T (library GEMM on the row-padded input), attention logits (csl_gat_logits_fwd_f32), destination logits
    (row gather), the fused edge-softmax aggregation (csl_gat_fwd_f32) and the epilogue act(n / s + bias)
    (csl_gat_finish_fwd_f32).  Backward: epilogue (csl_gat_finish_bwd_f32: g_n, g_s, bias gradient), aggregation
    (csl_gat_bwd_f32: atomics into ONE zeroed g_z), logits (csl_gat_logits_bwd_acc_f32 adds its share to the same
    g_z), then the two GEMMs.  As separate autograd nodes the same layer ran a dozen torch broadcast / reduction /
    accumulation kernels and three 90 MB gradient additions per step (profiles/r2_e2e_gat)."""

    @staticmethod
    def forward(ctx, x, weight, attn_l, attn_r, bias, indptr, indices, self_ids_in, n_out, slope, elu, row_pad,
                weight_grad, t_indptr=None, t_indices=None, t_max_len=0, n_rows=None, pad_out=False):
        """n_rows: x is ALREADY the row-padded buffer of a PaddedRows (its first n_rows rows are the input); pad_out:
        return the output as the padded buffer [pad(n_out), H*D] (zero rows behind n_out) for the next layer."""
        H, D = attn_l.shape
        Cw = H * D
        n_in, mp = x.shape[0], x.shape[0]
        if n_rows is not None:
            n_in = int(n_rows)
        elif row_pad and n_in >= row_pad:
            mp = (n_in + row_pad - 1) // row_pad * row_pad
        x = _f32(x)
        if n_rows is not None and x.is_contiguous():
            xp = x
        elif mp != n_in or not x.is_contiguous():
            xp = torch.empty((mp, x.shape[1]), dtype=torch.float32, device=x.device)
            xp[:n_in].copy_(x[:n_in])
            if mp != n_in:
                xp[n_in:].zero_()
        else:
            xp = x
        # (csl_gemm_f32: the algorithm is chosen by timing the library's candidates per shape class and size bucket; torch's
        # matmul takes the library's first suggestion, which for the 8 k-row layers is a 256 x 256 tile kernel on 32
        # workgroups: 44-68 us for ~1 GFLOP.  CSLICER_GAT_TORCH_MM=1: A/B switch)
        z = xp @ weight.t() if _GAT_TORCH_MM else gemm(xp, _f32(weight), transb=True)      # [mp, H*D]
        al, ar, b = _f32(attn_l).contiguous(), _f32(attn_r).contiguous(), _f32(bias).contiguous()
        indptr, indices, self_ids_in = _i32(indptr), _i32(indices), _i32(self_ids_in)
        dev = x.device
        L = _lib()
        el = torch.empty((n_in, H), dtype=torch.float32, device=dev)
        er = torch.empty((n_in, H), dtype=torch.float32, device=dev)
        _chk(L.csl_gat_logits_fwd_f32(_p(z), _p(al), _p(ar), n_in, H, D, _p(el), _p(er), _stream()),
             "csl_gat_logits_fwd_f32")
        er_out = gather_rows(er, self_ids_in)                       # logits of the destinations, [n_out, H]
        m = torch.empty((n_out, H), dtype=torch.float32, device=dev)
        s = torch.empty((n_out, H), dtype=torch.float32, device=dev)
        n = torch.empty((n_out, Cw), dtype=torch.float32, device=dev)
        _chk(L.csl_gat_fwd_f32(_p(indptr), _p(indices), n_out, _p(el), _p(er_out), _p(z), H, D, slope, _p(m), _p(s), _p(n),
                               _stream()), "csl_gat_fwd_f32")
        out = padded_rows(n_out, Cw, row_pad, dev).t if pad_out else torch.empty((n_out, Cw), dtype=torch.float32,
                                                                                  device=dev)
        _chk(L.csl_gat_finish_fwd_f32(_p(n), _p(s), _p(b), n_out, H, D, 1 if elu else 0, _p(out), _stream()),
             "csl_gat_finish_fwd_f32")
        ctx.save_for_backward(xp, weight, al, ar, z, el, er_out, m, s, n, out, indptr, indices, self_ids_in)
        ctx.cfg = (n_in, n_out, H, D, slope, bool(elu), weight_grad)
        ctx.x_rows = x.shape[0]
        # the slice by source (engine flags FLAG_TRANSPOSE | FLAG_TRANSPOSE_ALL): the backward then writes the gradient
        # of z row by row instead of scattering it with atomics into a zeroed buffer
        ctx.by_source = ((_i32(t_indptr), _i32(t_indices)) if t_indptr is not None and t_indptr.numel() == n_in + 1
                         and t_max_len <= _abi.T_SORTED_MAX else None)     # (a hub's list: the atomic form)
        return out

    @staticmethod
    def backward(ctx, g):
        xp, weight, al, ar, z, el, er_out, m, s, n, out, indptr, indices, self_ids_in = ctx.saved_tensors
        n_in, n_out, H, D, slope, elu, weight_grad = ctx.cfg
        Cw, dev = H * D, z.device
        L = _lib()
        g = _f32(g)
        if g.stride(-1) != 1 or g.stride(0) % 4:
            g = g.contiguous()
        g_n = torch.empty((n_out, Cw), dtype=torch.float32, device=dev)
        g_s = torch.empty((n_out, H), dtype=torch.float32, device=dev)
        buf = torch.empty((3 * Cw + max(int(L.csl_gat_finish_bwd_scratch(n_out, H, D)),
                                        2 * int(L.csl_gat_logits_bwd_scratch(n_in, H, D)), 4),), dtype=torch.float32,
                          device=dev)
        g_bias, g_al, g_ar = buf[:Cw], buf[Cw:2 * Cw].view(H, D), buf[2 * Cw:3 * Cw].view(H, D)
        scratch = C.c_void_p(buf.data_ptr() + 12 * Cw)
        _chk(L.csl_gat_finish_bwd_f32(_p(g), g.stride(0), _p(out), _p(n), _p(s), n_out, H, D, 1 if elu else 0, _p(g_n),
                                      _p(g_s), _p(g_bias), scratch, _stream()), "csl_gat_finish_bwd_f32")
        # ONE gradient buffer for z (padded like the GEMM operand): the aggregation's share, then the logits' share
        if ctx.by_source is not None and not os.environ.get("CSLICER_GAT_NO_FOLD"):
            # by source, with the logits' backward folded in: g_z is written complete in one pass over the source rows
            # plus a small one over the destinations (csl_gat_bwd_t_fused_f32)
            tptr, trow = ctx.by_source
            g_z = torch.empty((z.shape[0], Cw), dtype=torch.float32, device=dev)     # every row is written
            g_er_out = torch.empty((n_out, H), dtype=torch.float32, device=dev)
            fs = torch.empty((max(int(L.csl_gat_bwd_t_fused_scratch(z.shape[0], n_out, H, D)), 4),), dtype=torch.float32,
                             device=dev)
            _chk(L.csl_gat_bwd_t_fused_f32(_p(tptr), _p(trow), n_in, z.shape[0], _p(el), _p(er_out), _p(z), H, D, slope, _p(m),
                                           _p(g_s), _p(g_n), _p(al), _p(ar), _p(self_ids_in), n_out, _p(g_er_out), _p(g_z),
                                           C.c_void_p(g_al.data_ptr()), C.c_void_p(g_ar.data_ptr()), _p(fs), _stream()),
                 "csl_gat_bwd_t_fused_f32")
        else:
            if ctx.by_source is not None:
                tptr, trow = ctx.by_source
                g_z = torch.empty((z.shape[0], Cw), dtype=torch.float32, device=dev)     # every row is written
                g_el = torch.empty((n_in, H), dtype=torch.float32, device=dev)
                g_er_out = torch.zeros((n_out, H), dtype=torch.float32, device=dev)
                _chk(L.csl_gat_bwd_t_f32(_p(tptr), _p(trow), n_in, z.shape[0], _p(el), _p(er_out), _p(z), H, D, slope, _p(m),
                                         _p(g_s), _p(g_n), _p(g_el), _p(g_er_out), _p(g_z), _stream()), "csl_gat_bwd_t_f32")
            else:
                g_z = torch.zeros((z.shape[0], Cw), dtype=torch.float32, device=dev)
                g_el = torch.zeros((n_in, H), dtype=torch.float32, device=dev)
                g_er_out = torch.empty((n_out, H), dtype=torch.float32, device=dev)
                _chk(L.csl_gat_bwd_f32(_p(indptr), _p(indices), n_out, _p(el), _p(er_out), _p(z), H, D, slope, _p(m), _p(g_s),
                                       _p(g_n), _p(g_el), _p(g_er_out), _p(g_z), _stream()), "csl_gat_bwd_f32")
            g_er = torch.zeros((n_in, H), dtype=torch.float32, device=dev)
            scatter_add_rows_(g_er, self_ids_in, g_er_out)              # (a node is the self source of one destination)
            _chk(L.csl_gat_logits_bwd_acc_f32(_p(z), _p(al), _p(ar), _p(g_el), _p(g_er), n_in, H, D, _p(g_z), 1,
                                              C.c_void_p(g_al.data_ptr()), C.c_void_p(g_ar.data_ptr()), scratch, _stream()),
                 "csl_gat_logits_bwd_acc_f32")
        gw = weight_grad(g_z, xp)
        gx = None
        if ctx.needs_input_grad[0]:
            gx = g_z @ weight if _GAT_TORCH_MM else gemm(g_z, weight)   # rows behind n_in are zero (g_z's are)
            if gx.shape[0] != ctx.x_rows:
                gx = gx[:ctx.x_rows]
        return gx, gw, g_al, g_ar, g_bias, None, None, None, None, None, None, None, None, None, None, None, None, None


class FeatureRows(object):
    """The input of the deepest layer as (resident feature table, rows): row s of the layer's input is
    table[rows[s]] (rows = the slice's in_nodes).  A layer that can read its input through the map (GatInputLayer)
    takes this instead of a gathered matrix."""
    __slots__ = ("table", "rows")

    def __init__(self, table, rows):
        self.table, self.rows = table, rows


def gat_input_ok(H, F, fanout):
    """whether GatInputLayer covers a layer of H heads on F input features whose rows have <= fanout edges"""
    return H in (1, 2, 4, 8) and F % 4 == 0 and 4 <= F <= 128 and 0 < fanout <= int(_lib().csl_gat_in_max_degree())


def _gemm_batched(transa, transb, m, n, k, a, lda, sa, b, ldb, sb, c, ldc, sc, batch):
    L = _lib()
    rc = L.csl_gemm_f32(int(transa), int(transb), m, n, k, _p(a), lda, sa, _p(b), ldb, sb, _p(c), ldc, sc, batch,
                        C.c_void_p(0), 0, _stream())
    if rc < 0:
        raise _abi.CslError(rc, "csl_gemm_f32: " + L.csl_gemm_last_error().decode())


class GatInputLayer(torch.autograd.Function):
    """A DistGATConv layer (+ ELU) of a single part whose input is the FEATURE TABLE (no input gradient), as
    aggregate-then-project (csrc/gat_input.hip): the attention logits and the weighted sum are linear in x, so
        v_l = W_h^T a_l, v_r = W_h^T a_r;  agg[v, h] = sum_u alpha_h(u -> v) x[u]  (csl_gat_in_fwd_f32, raw rows through
        `rows`);  out[v, h] = W_h agg[v, h] + bias  (H GEMMs over the n_out DESTINATIONS)
    equals GatLayerLocal on the gathered rows up to fp32 rounding, with a tenth of its flops and without the projected
    source matrix, the gathered input matrix or the layer's slice by source.  Backward: dW_h = g_h^T agg_h, dagg_h = g_h W_h,
    one pass over the edges for the gradients of v_l / v_r (csl_gat_in_bwd_f32), then the chain rule through v = W^T a.
    Where the fp32-MFMA projection kernels cover the shape (csl_gat_in_proj_ok: D in {16, 32, 64}) each direction is ONE
    native call (csl_gat_in_layer_fwd_f32 / _bwd_f32); otherwise the products are strided-batched csl_gemm_f32 calls and
    the small pieces torch ops."""

    @staticmethod
    def forward(ctx, table, rows, weight, attn_l, attn_r, bias, indptr, indices, self_ids_in, n_out, n_edges, max_deg, slope,
                elu, row_pad, pad_out):
        """max_deg: no row of the CSR has more edges (the slicer's fanout of the layer)"""
        H, D = attn_l.shape
        F, Cw = table.shape[1], H * D
        table, weight = _f32(table), _f32(weight).contiguous()
        al, ar, b = _f32(attn_l).contiguous(), _f32(attn_r).contiguous(), _f32(bias).contiguous()
        indptr, indices, self_ids_in = _i32(indptr), _i32(indices), _i32(self_ids_in)
        rows = _i32(rows) if rows is not None else None
        dev = table.device
        L = _lib()
        agg = torch.empty((n_out, H * F), dtype=torch.float32, device=dev)
        alpha = torch.empty((max(n_edges, 1), H), dtype=torch.float32, device=dev)
        out = padded_rows(n_out, Cw, row_pad, dev).t if pad_out else torch.empty((n_out, Cw), dtype=torch.float32,
                                                                                  device=dev)
        # out[:, h*D:(h+1)*D] = agg[:, h*F:(h+1)*F] @ W_h^T + bias, ELU: the H matrices interleaved in agg / out rows
        mfma = bool(L.csl_gat_in_proj_ok(H, F, D)) and not os.environ.get("CSLICER_GAT_IN_LIBGEMM")
        if mfma:
            # the whole forward as one native call: v_l / v_r, the edge pass, the projection on the fp32 matrix cores
            buf = torch.empty((max(int(L.csl_gat_in_layer_fwd_scratch(H, F)), 4),), dtype=torch.float32, device=dev)
            _chk(L.csl_gat_in_layer_fwd_f32(_p(indptr), _p(indices), _p(self_ids_in), _p(rows), _p(table), table.stride(0), F,
                                            _p(weight), _p(al), _p(ar), _p(b), H, D, slope, 1 if elu else 0, n_out, n_edges,
                                            max_deg, _p(agg), _p(alpha), _p(out), Cw, _p(buf), _stream()),
                 "csl_gat_in_layer_fwd_f32")
        else:
            Wv = weight.view(H, D, F)
            vl = torch.einsum("hdf,hd->hf", Wv, al).contiguous()
            vr = torch.einsum("hdf,hd->hf", Wv, ar).contiguous()
            _chk(L.csl_gat_in_fwd_f32(_p(indptr), _p(indices), _p(self_ids_in), _p(rows), _p(table), table.stride(0), F,
                                      _p(vl), _p(vr), H, slope, n_out, n_edges, max_deg, _p(agg), _p(alpha), _stream()),
                 "csl_gat_in_fwd_f32")
            if n_out:
                _gemm_batched(0, 1, n_out, D, F, agg, H * F, F, weight, F, D * F, out, Cw, D, H)
                _chk(L.csl_bias_elu_f32(_p(out), Cw, _p(b), n_out, Cw, 1 if elu else 0, _stream()), "csl_bias_elu_f32")
        ctx.save_for_backward(table, rows, weight, al, ar, agg, alpha, out, indptr, indices, self_ids_in)
        ctx.cfg = (n_out, n_edges, max_deg, H, D, F, slope, bool(elu), mfma)
        return out

    @staticmethod
    def backward(ctx, g):
        table, rows, weight, al, ar, agg, alpha, out, indptr, indices, self_ids_in = ctx.saved_tensors
        n_out, n_edges, max_deg, H, D, F, slope, elu, mfma = ctx.cfg
        Cw, dev = H * D, table.device
        L = _lib()
        g = _f32(g)
        if g.stride(-1) != 1 or g.stride(0) % 4:
            g = g.contiguous()
        gg = torch.empty((n_out, Cw), dtype=torch.float32, device=dev)
        g_bias = torch.empty((Cw,), dtype=torch.float32, device=dev)
        FP = int(L.csl_gat_in_proj_fpad(F)) if mfma else F     # head stride of dagg (whole 16-column tiles on the MFMA path)
        gW = torch.empty((H, D, F), dtype=torch.float32, device=dev)
        dagg = torch.empty((n_out, H * FP), dtype=torch.float32, device=dev)
        if mfma:
            # the whole backward as one native call (five kernels, one second-stage launch, the chain rule through v = W^T a)
            g_a = torch.empty((2, H, D), dtype=torch.float32, device=dev)
            buf = torch.empty((max(int(L.csl_gat_in_layer_bwd_scratch(n_out, H, F, D)), 4),), dtype=torch.float32, device=dev)
            _chk(L.csl_gat_in_layer_bwd_f32(_p(indptr), _p(indices), _p(self_ids_in), _p(rows), _p(table), table.stride(0), F,
                                            _p(weight), _p(al), _p(ar), H, D, slope, 1 if elu else 0, n_out, n_edges, max_deg,
                                            _p(agg), _p(alpha), _p(out), Cw, _p(g), g.stride(0), _p(gg), _p(dagg),
                                            C.c_void_p(gW.data_ptr()), C.c_void_p(g_a[0].data_ptr()),
                                            C.c_void_p(g_a[1].data_ptr()), _p(g_bias), _p(buf), _stream()),
                 "csl_gat_in_layer_bwd_f32")
            return (None, None, gW.view(Cw, F), g_a[0], g_a[1], g_bias) + (None,) * 10
        buf = torch.empty((max(int(L.csl_elu_bwd_colsum_scratch(n_out, Cw)), int(L.csl_gat_in_bwd_scratch(n_out, H, F)), 4),),
                          dtype=torch.float32, device=dev)
        _chk(L.csl_elu_bwd_colsum_f32(_p(g), g.stride(0), _p(out), Cw, n_out, Cw, 1 if elu else 0, _p(gg), Cw, _p(g_bias),
                                      _p(buf), _stream()), "csl_elu_bwd_colsum_f32")
        g_v = torch.empty((2, H, F), dtype=torch.float32, device=dev)
        # dW_h = g_h^T agg_h  [D, F] (the sum runs over the n_out rows);  dagg_h = g_h W_h  [n_out, F]
        if n_out:
            _gemm_batched(1, 0, D, F, n_out, gg, Cw, D, agg, H * F, F, gW, F, D * F, H)
            _gemm_batched(0, 0, n_out, F, D, gg, Cw, D, weight, F, D * F, dagg, H * F, F, H)
        else:
            gW.zero_()
        _chk(L.csl_gat_in_bwd_f32(_p(indptr), _p(indices), _p(self_ids_in), _p(rows), _p(table), table.stride(0), F, _p(alpha),
                                  _p(dagg), H * FP, FP, H, slope, n_out, n_edges, max_deg, C.c_void_p(g_v[0].data_ptr()),
                                  C.c_void_p(g_v[1].data_ptr()), _p(buf), _stream()), "csl_gat_in_bwd_f32")
        # chain rule through v_l[h] = W_h^T a_l[h] (and v_r)
        Wv = weight.view(H, D, F)
        gW = gW + al.unsqueeze(2) * g_v[0].unsqueeze(1) + ar.unsqueeze(2) * g_v[1].unsqueeze(1)
        g_a = torch.einsum("hdf,shf->shd", Wv, g_v)
        return (None, None, gW.view(Cw, F), g_a[0], g_a[1], g_bias) + (None,) * 10


class GatLayerRank(torch.autograd.Function):
    """A whole DistGATConv layer of ONE part = one process (DistGATConv.forward_rank) as one autograd node: projection
    (csl_gemm_f32), logits (csl_gat_logits_fwd_f32), the partial edge-softmax state of the local edges (csl_gat_fwd_f32), the
    two boundary exchanges and the merge at the owners, the epilogue (csl_gat_finish_fwd_f32); the backward by hand with
    the same kernels' backward halves and the three reverse exchanges.  The exchanges are `comm._exchange` calls over the
    slice's back-to-back per-peer lists (to_all / from_all): nothing loops over the peers.  As separate autograd nodes
    (DistGATConv._forward_rank_autograd) the layer was ~150 small launches forward + backward and a rank's share of an
    8-GPU minibatch a 3.7 ms step whatever its size (profiles/r3_rank/gat_world_of_one.md).

    Merge at the owner of a row, over its own partial state and the peers' (m_p, s_p, n_p):
        M = max_p m_p;  c_p = exp(m_p - M);  S = sum_p c_p s_p;  N = sum_p c_p n_p;  out = N / S + bias.
    M and the m_p are stabilisers (the result does not depend on them): no gradient flows through them, so
        g_N = g_out / S,  g_S = -<g_out, N> / S^2  (csl_gat_finish_bwd_f32),  g_n_p = c_p g_N,  g_s_p = c_p g_S,
    which travel back to the holders of the edges, whose local backward (csl_gat_bwd_f32) needs only its own m_p."""

    @staticmethod
    def forward(ctx, x, weight, attn_l, attn_r, bias, sl, comm, slope, elu):
        H, D = attn_l.shape
        Cw = H * D
        x = _f32(x).contiguous()
        weight = _f32(weight).contiguous()
        al, ar, b = _f32(attn_l).contiguous(), _f32(attn_r).contiguous(), _f32(bias).contiguous()
        n_in, n_out, dev = x.shape[0], sl.n_out, x.device
        L = _lib()
        indptr, indices = _i32(sl.indptr), _i32(sl.indices)
        owned, self_in = sl.owned_out_nodes.long(), sl.self_ids_in.long()
        to_all, from_all = _i32(sl.to_all), _i32(sl.from_all)
        tl, fl = to_all.long(), from_all.long()
        to_counts, from_counts = list(sl.to_counts), list(sl.from_counts)
        multi = comm is not None     # (a world of one runs the exchanges too, with empty lists: the proxy pays their host cost)
        z = gemm(x, weight, transb=True) if n_in else torch.zeros((0, Cw), dtype=torch.float32, device=dev)
        el = torch.empty((n_in, H), dtype=torch.float32, device=dev)
        er = torch.empty((n_in, H), dtype=torch.float32, device=dev)
        _chk(L.csl_gat_logits_fwd_f32(_p(z), _p(al), _p(ar), n_in, H, D, _p(el), _p(er), _stream()), "csl_gat_logits_fwd_f32")
        # logits of the destinations: own rows from the own projection, the others from their owners
        er_out = torch.zeros((n_out, H), dtype=torch.float32, device=dev)
        er_out.index_copy_(0, owned, er.index_select(0, self_in))
        if multi:
            recv = comm._exchange(er_out.index_select(0, tl), to_counts, from_counts)
            er_out.index_copy_(0, fl, recv)
        m = torch.empty((n_out, H), dtype=torch.float32, device=dev)
        s_ = torch.empty((n_out, H), dtype=torch.float32, device=dev)
        n_ = torch.empty((n_out, Cw), dtype=torch.float32, device=dev)
        _chk(L.csl_gat_fwd_f32(_p(indptr), _p(indices), n_out, _p(el), _p(er_out), _p(z), H, D, slope, _p(m), _p(s_), _p(n_),
                               _stream()), "csl_gat_fwd_f32")
        # partial states of the boundary rows to their owners, merged there
        c_own, c_recv = None, None
        S, N = s_, n_
        if multi:
            pack = torch.cat([m, s_, n_], dim=1)
            recv = comm._exchange(pack.index_select(0, fl), from_counts, to_counts)          # [len(to_all), 2H + Cw]
            M = m.clone()
            if tl.numel():
                M.scatter_reduce_(0, tl.unsqueeze(1).expand(-1, H), recv[:, :H], "amax", include_self=True)
            c_own = torch.exp(m - M)
            S = s_ * c_own
            N = (n_.view(n_out, H, D) * c_own.unsqueeze(2)).view(n_out, Cw)
            if tl.numel():
                c_recv = torch.exp(recv[:, :H] - M.index_select(0, tl))
                S.index_add_(0, tl, recv[:, H:2 * H] * c_recv)
                N.view(n_out, H, D).index_add_(0, tl, recv[:, 2 * H:2 * H + Cw].reshape(-1, H, D) * c_recv.unsqueeze(2))
        S_o = S.index_select(0, owned).contiguous()
        N_o = N.index_select(0, owned).contiguous()
        n_owned = owned.numel()
        out = torch.empty((n_owned, Cw), dtype=torch.float32, device=dev)
        _chk(L.csl_gat_finish_fwd_f32(_p(N_o), _p(S_o), _p(b), n_owned, H, D, 1 if elu else 0, _p(out), _stream()),
             "csl_gat_finish_fwd_f32")
        ctx.save_for_backward(x, weight, al, ar, z, el, er_out, m, S_o, N_o, out, indptr, indices, owned, self_in, tl, fl,
                              c_own if c_own is not None else torch.empty(0, device=dev),
                              c_recv if c_recv is not None else torch.empty(0, device=dev))
        ctx.cfg = (n_in, n_out, H, D, slope, bool(elu), comm if multi else None, to_counts, from_counts)
        return out

    @staticmethod
    def backward(ctx, g):
        (x, weight, al, ar, z, el, er_out, m, S_o, N_o, out, indptr, indices, owned, self_in, tl, fl, c_own,
         c_recv) = ctx.saved_tensors
        n_in, n_out, H, D, slope, elu, comm, to_counts, from_counts = ctx.cfg
        Cw, dev = H * D, x.device
        L = _lib()
        g = _f32(g)
        if g.stride(-1) != 1 or g.stride(0) % 4:
            g = g.contiguous()
        n_owned = owned.numel()
        g_n_o = torch.empty((n_owned, Cw), dtype=torch.float32, device=dev)
        g_s_o = torch.empty((n_owned, H), dtype=torch.float32, device=dev)
        buf = torch.empty((3 * Cw + max(int(L.csl_gat_finish_bwd_scratch(n_owned, H, D)),
                                        2 * int(L.csl_gat_logits_bwd_scratch(n_in, H, D)), 4),), dtype=torch.float32,
                          device=dev)
        g_bias, g_al, g_ar = buf[:Cw], buf[Cw:2 * Cw].view(H, D), buf[2 * Cw:3 * Cw].view(H, D)
        scratch = C.c_void_p(buf.data_ptr() + 12 * Cw)
        _chk(L.csl_gat_finish_bwd_f32(_p(g), g.stride(0), _p(out), _p(N_o), _p(S_o), n_owned, H, D, 1 if elu else 0, _p(g_n_o),
                                      _p(g_s_o), _p(g_bias), scratch, _stream()), "csl_gat_finish_bwd_f32")
        # gradients of the merged state in out-row order, then of every contribution: c_p times them
        g_s = torch.zeros((n_out, H), dtype=torch.float32, device=dev)
        g_n = torch.zeros((n_out, Cw), dtype=torch.float32, device=dev)
        g_s.index_copy_(0, owned, g_s_o)
        g_n.index_copy_(0, owned, g_n_o)
        if comm is not None:
            send = torch.cat([g_s.index_select(0, tl) * c_recv,
                              (g_n.index_select(0, tl).view(-1, H, D) * c_recv.unsqueeze(2)).view(-1, Cw)], dim=1) \
                if tl.numel() else torch.zeros((0, H + Cw), dtype=torch.float32, device=dev)
            recv = comm._exchange(send, to_counts, from_counts)                                # [len(from_all), H + Cw]
            g_s.mul_(c_own)
            g_n.view(n_out, H, D).mul_(c_own.unsqueeze(2))
            if fl.numel():
                g_s.index_copy_(0, fl, recv[:, :H])
                g_n.index_copy_(0, fl, recv[:, H:])
        # the local edges
        g_el = torch.zeros((n_in, H), dtype=torch.float32, device=dev)
        g_er_out = torch.empty((n_out, H), dtype=torch.float32, device=dev)
        g_z = torch.zeros((n_in, Cw), dtype=torch.float32, device=dev)
        _chk(L.csl_gat_bwd_f32(_p(indptr), _p(indices), n_out, _p(el), _p(er_out), _p(z), H, D, slope, _p(m), _p(g_s), _p(g_n),
                               _p(g_el), _p(g_er_out), _p(g_z), _stream()), "csl_gat_bwd_f32")
        # the destinations' logits: boundary rows back to their owners (a row may come back from several peers)
        if comm is not None:
            recv = comm._exchange(g_er_out.index_select(0, fl), from_counts, to_counts)
            if tl.numel():
                g_er_out.index_add_(0, tl, recv)
        g_er = torch.zeros((n_in, H), dtype=torch.float32, device=dev)
        g_er.index_copy_(0, self_in, g_er_out.index_select(0, owned))
        _chk(L.csl_gat_logits_bwd_acc_f32(_p(z), _p(al), _p(ar), _p(g_el), _p(g_er), n_in, H, D, _p(g_z), 1,
                                          C.c_void_p(g_al.data_ptr()), C.c_void_p(g_ar.data_ptr()), scratch, _stream()),
             "csl_gat_logits_bwd_acc_f32")
        gw = gemm(g_z, x, transa=True) if n_in else torch.zeros_like(weight)
        gx = gemm(g_z, weight) if ctx.needs_input_grad[0] and n_in else (
            torch.zeros_like(x) if ctx.needs_input_grad[0] else None)
        return gx, gw, g_al, g_ar, g_bias, None, None, None, None


def attention_gather(indptr, indices, u_in, v_in, n_rows):
    """BipartiteGraph.attention_gather (python/data/bipartite.py:75-80): out[v] = sum over in-edges
    (u, v) of u_in[u] * v_in[v] (DGL u_mul_v + sum).  v_in[v] does not depend on u, so this is the
    sum-aggregate scaled row-wise: one SpMM launch plus an elementwise product."""
    return SpmmSum.apply(u_in, indptr, indices, n_rows) * v_in


class Adam(object):
    """torch.optim.Adam's update (lr, betas, eps; no weight decay, no amsgrad: what python/train.py:83 uses) for a
    handful of fp32 CUDA parameters in ONE HIP launch per step (csl_adam_f32).  Same interface as far as the
    trainer needs it: zero_grad(set_to_none=True), step(); the moments live in `state`."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params]
        if not self.params or len(self.params) > 24:
            raise ValueError("1..24 parameter tensors")
        for p in self.params:
            if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                raise TypeError("contiguous float32 CUDA parameters expected")
        self.lr, self.betas, self.eps, self.t = float(lr), (float(betas[0]), float(betas[1])), float(eps), 0
        self.state = [(torch.zeros_like(p), torch.zeros_like(p)) for p in self.params]
        n = len(self.params)
        self._p = (C.c_void_p * n)(*[p.data_ptr() for p in self.params])
        self._m = (C.c_void_p * n)(*[m.data_ptr() for m, _ in self.state])
        self._v = (C.c_void_p * n)(*[v.data_ptr() for _, v in self.state])
        self._n = (C.c_int64 * n)(*[p.numel() for p in self.params])
        self._g = (C.c_void_p * n)()

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def step(self, flat_grads=None):
        """flat_grads: optional contiguous fp32 tensor holding every parameter's gradient back to back (in
        parameter order) -- the all-reduced buffer of a multi-rank job is used in place, not copied back."""
        o = 0
        for k, p in enumerate(self.params):
            if flat_grads is not None:
                self._g[k] = flat_grads.data_ptr() + 4 * o
                o += p.numel()
                continue
            g = p.grad
            if g is None:
                raise RuntimeError("parameter %d has no gradient" % k)
            if not g.is_contiguous():
                g = p.grad = g.contiguous()
            self._g[k] = g.data_ptr()
        self.t += 1
        _chk(_lib().csl_adam_f32(len(self.params), self._p, self._g, self._m, self._v, self._n, self.lr,
                                 self.betas[0], self.betas[1], self.eps, self.t, _stream()), "csl_adam_f32")

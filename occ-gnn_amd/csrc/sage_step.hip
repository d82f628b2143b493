// sage_step.hip -- forward, loss and backward of the GraphSAGE model of ONE part (a GPU that holds every node) for one
// minibatch as ONE call behind the C ABI (cslicer_aggr.h, csl_sage_fwd_bwd_f32): the sequence of this library's fused
// kernels and plain GEMMs that python/train.py:56-88 + python/layers/dist_sageconv.py:42-84 amount to on one GPU.
//
// Why a native sequencer: the step is ~25 kernel launches and 8 GEMMs of 5-90 us each.  Issued from Python (one ctypes
// call or torch op each, an autograd graph around them) they cost 0.55-0.75 ms of host time per step, more than the
// 0.6 ms the GPU needs, and the rate then follows the host's speed (1.2-1.8 k minibatches/s from box to box).  Issued
// from here a step is ~0.15 ms of host time and the GPU sets the pace.
//
// Per model layer k (deepest hop first; slice k = the engine's layer n_layers-1-k, graph mode, FLAG_TRANSPOSE):
//   forward   cat_k = [x[self] | mean_{CSR row} x[src]]       csl_sage_cat_f32 (k = 0 reads the resident feature table
//             y_k   = cat_k W_k^T + b_k (ReLU for k < L-1)     through the slice's in_nodes)      + csl_gemm_f32
//   loss      csl_softmax_ce_partial_f32 on y_{L-1} (forward, gradient = the top layer's padded gy, bias column sums)
//   backward  gW_k  = gy_k^T cat_k (row slabs)                 csl_gemm_f32 (batched)
//             gcat  = gy_k W_k                                 csl_gemm_f32
//             gy_{k-1}, gb_{k-1} = gather of gcat over the slice by source, ReLU mask of y_{k-1}, row padding and
//                                  bias column sums in the same pass                   csl_sage_cat_bwd_t_f32
//   the second stage of every reduction above (bias sums, slab sums, the loss): ONE launch (csl_reduce_multi_f32)
// Rows are padded to a multiple of `row_pad` so that GEMM shapes repeat from minibatch to minibatch.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "cslicer_aggr.h"
#include "cslicer_hip.h"

namespace {

thread_local char s_err[200];

// every layer defers a bias sum and possibly a weight-gradient slab sum, the step one loss sum: one launch finishes them
static_assert(2 * CSL_MAX_LAYERS + 1 <= CSL_REDUCE_MULTI_MAX, "csl_reduce_multi_f32 takes the step's second stages in one call");

// GEMM row counts repeat from minibatch to minibatch (a plan per shape: gemm_lt.hip): multiples of row_pad for tall
// operands, of 256 below that (a rank of an N-GPU job sees a few hundred to a few thousand rows per layer)
inline int64_t pad_rows(int64_t m, int64_t row_pad) {
  if (row_pad <= 0) return m;
  const int64_t q = m >= row_pad ? row_pad : (row_pad < 256 ? row_pad : 256);
  return (m + q - 1) / q * q;
}
inline int64_t up4(int64_t x) { return (x + 3) & ~(int64_t)3; }  // every buffer starts 16-byte aligned

struct Layout {
  int64_t cat[CSL_MAX_LAYERS], y[CSL_MAX_LAYERS], gy[CSL_MAX_LAYERS], gcat[CSL_MAX_LAYERS], mp[CSL_MAX_LAYERS];
  bool hub[CSL_MAX_LAYERS];
  // first stages of the step's reductions, each in a buffer of its own: they are all finished by ONE launch at the end
  int64_t slabs[CSL_MAX_LAYERS];    // [n_slabs][out][2 in] of a layer whose weight gradient is computed in row slabs
  int64_t bpart[CSL_MAX_LAYERS];    // [blocks][out]: per-block column sums behind gb_k (k < L-1: from layer k+1's gather)
  int64_t bblocks[CSL_MAX_LAYERS];
  int64_t lpart, lblocks;           // [blocks]: the loss
  int64_t wpack;                    // W_0 in MFMA operand order (csl_sage_fwd_mfma_f32), -1: the deepest layer is not fused
  bool slabbed[CSL_MAX_LAYERS];
  bool top_cols;                    // the softmax pass also leaves the top layer's bias column sums
  int64_t g, scratch, total;
};

// bump allocation of the step's buffers (floats); returns false for an unsupported model
bool lay_out(int32_t L, const int32_t* dims, const csl_sage_slice* sl, int64_t row_pad, int32_t n_slabs, Layout& o) {
  if (L < 1 || L > CSL_MAX_LAYERS || n_slabs < 1) return false;
  int64_t at = 0, scratch = 0;
  for (int k = 0; k < L; k++) {
    const int64_t in = dims[k], out = dims[k + 1];
    if (in < 4 || in % 4 != 0 || out < 1 || sl[k].n_out < 0 || sl[k].n_in < 0) return false;
    if (k + 1 < L && out % 4 != 0) return false;  // a hidden width feeds the next layer's float4 kernels
    if (k > 0 && sl[k].n_in != sl[k - 1].n_out) return false;  // layer k's sources are layer k-1's outputs
    const int64_t mp = pad_rows(sl[k].n_out, row_pad);
    o.mp[k] = mp;
    o.cat[k] = at, at += up4(mp * 2 * in);
    o.y[k] = at, at += up4(mp * out);
    o.gy[k] = at, at += up4(mp * out);
    o.gcat[k] = at, at += k > 0 ? up4(mp * 2 * in) : 0;
    o.hub[k] = k > 0 && sl[k].t_max_len > CSL_T_SORTED_MAX;   // a hub list: its rows are gathered by many workgroups
    const int64_t wn = out * 2 * in;
    o.slabbed[k] = row_pad > 0 && mp >= row_pad && n_slabs > 1 && mp % n_slabs == 0 && wn % 4 == 0;
    o.slabs[k] = at, at += o.slabbed[k] ? up4(wn * n_slabs) : 0;
    // gb_k's first stage: written by layer k+1's gather (k < L-1; twice the blocks when that layer has hub lists), or by
    // the loss pass / relu_bwd_colsum (k = L-1)
    const bool hub_above = k + 1 < L && sl[k + 1].t_max_len > CSL_T_SORTED_MAX;
    const int64_t part = k + 1 < L ? (hub_above ? csl_sage_cat_bwd_t_hub_scratch(mp, (int32_t)out)
                                                : csl_sage_cat_bwd_t_scratch(mp, (int32_t)out)) : 0;
    o.bblocks[k] = out > 0 ? part / out : 0;
    o.bpart[k] = at, at += up4(part);
  }
  {
    const int k = L - 1;
    const int64_t C = dims[L], rows = o.mp[k];
    o.top_cols = C <= 256;
    o.lblocks = (rows + 3) / 4;
    o.lpart = at, at += up4(o.lblocks);
    if (o.top_cols) {
      o.bblocks[k] = o.lblocks;
      o.bpart[k] = at, at += up4(o.lblocks * C);
    } else {
      scratch = csl_relu_bwd_colsum_scratch(rows, (int32_t)C);
    }
    // the gradient w.r.t. the logits IS the top layer's (padded) gy when the loss pass pads it; else a buffer of its own
    o.g = o.top_cols ? o.gy[k] : at;
    at += o.top_cols ? 0 : up4(sl[k].n_out * C);
  }
  o.scratch = at, at += up4(scratch);
  {
    // the deepest layer as ONE kernel (gather -> fp32 MFMA -> bias + ReLU) where its widths allow
    const int64_t wp = getenv("CSLICER_NO_MFMA_FWD") ? -1 : csl_sage_fwd_mfma_scratch(dims[0], dims[1]);
    o.wpack = wp > 0 ? at : -1;
    if (wp > 0) at += up4(wp);
  }
  o.total = at;
  return true;
}

// ---- diagnostics: device time of the step's launches by group (csl_sage_step_timing / _read): HIP events around every
// launch on the step's own stream, read back (and the stream drained) by the caller.  Off by default: no cost.
struct TimedSpan {
  int group;
  hipEvent_t e0, e1;
};
bool g_timing = false;
std::vector<TimedSpan> g_spans;       // recorded, not yet read
std::vector<hipEvent_t> g_free;       // event pool
std::mutex g_tmu;
hipEvent_t take_event() {
  if (!g_free.empty()) {
    hipEvent_t e = g_free.back();
    g_free.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
struct SpanGuard {   // records e0 now and e1 when it goes out of scope
  hipStream_t st;
  TimedSpan sp;
  bool on;
  SpanGuard(int group, void* stream) : st((hipStream_t)stream), on(g_timing) {
    if (!on) return;
    std::lock_guard<std::mutex> lk(g_tmu);
    sp.group = group, sp.e0 = take_event(), sp.e1 = take_event();
    (void)hipEventRecord(sp.e0, st);
  }
  ~SpanGuard() {
    if (!on) return;
    (void)hipEventRecord(sp.e1, st);
    std::lock_guard<std::mutex> lk(g_tmu);
    g_spans.push_back(sp);
  }
};
#define TSTEP(group, x)              \
  do {                               \
    SpanGuard sg_((group), stream);  \
    STEP(x);                         \
  } while (0)

#define STEP(x)                                                              \
  do {                                                                       \
    const int rc_ = (x);                                                     \
    if (rc_ < 0) {                                                           \
      snprintf(s_err, sizeof(s_err), "%s failed (%d), layer %d", #x, rc_, k); \
      return rc_;                                                            \
    }                                                                        \
  } while (0)

}  // namespace

extern "C" {

const char* csl_sage_last_error(void) { return s_err; }

int csl_sage_step_timing(int32_t enable) {
  std::lock_guard<std::mutex> lk(g_tmu);
  g_timing = enable != 0;
  return CSL_OK;
}

int csl_sage_step_timing_read(double* ms, int64_t* launches) {
  if (!ms || !launches) return CSL_E_INVALID;
  std::lock_guard<std::mutex> lk(g_tmu);
  for (int g = 0; g < CSL_STEP_GROUPS; g++) ms[g] = 0.0, launches[g] = 0;
  int rc = CSL_OK;
  for (const TimedSpan& sp : g_spans) {
    float t = 0.f;
    if (hipEventSynchronize(sp.e1) != hipSuccess || hipEventElapsedTime(&t, sp.e0, sp.e1) != hipSuccess) rc = CSL_E_HIP;
    if (sp.group >= 0 && sp.group < CSL_STEP_GROUPS) ms[sp.group] += t, launches[sp.group]++;
    g_free.push_back(sp.e0), g_free.push_back(sp.e1);
  }
  g_spans.clear();
  return rc;
}

int64_t csl_sage_fwd_bwd_workspace(int32_t n_layers, const int32_t* dims, const csl_sage_slice* slices, int64_t row_pad,
                                   int32_t n_slabs) {
  Layout o;
  if (!dims || !slices || !lay_out(n_layers, dims, slices, row_pad, n_slabs, o)) return CSL_E_INVALID;
  return o.total;
}

int csl_sage_fwd_bwd_f32(int32_t n_layers, const int32_t* dims, const csl_sage_slice* sl, const float* const* weights,
                         const float* const* biases, const float* feat, int64_t ldf, const int32_t* feat_rows,
                         const int32_t* seed_ids, const int64_t* labels, float scale, int64_t row_pad, int32_t n_slabs,
                         float* grads, float* loss, float* workspace, int64_t workspace_floats, void* stream) {
  int k = -1;
  s_err[0] = 0;
  Layout o;
  if (!dims || !sl || !weights || !biases || !grads || !loss || !lay_out(n_layers, dims, sl, row_pad, n_slabs, o)) {
    snprintf(s_err, sizeof(s_err), "bad argument or unsupported model (widths must be multiples of 4, 1..%d layers, "
             "n_in of a layer = n_out of the layer below)", CSL_MAX_LAYERS);
    return CSL_E_INVALID;
  }
  if (o.total > workspace_floats || (o.total > 0 && (!workspace || ((uintptr_t)workspace & 15)))) {
    snprintf(s_err, sizeof(s_err), "workspace: %lld floats needed, %lld given", (long long)o.total, (long long)workspace_floats);
    return CSL_E_INVALID;
  }
  const int L = n_layers;
  float* ws = workspace;
  // where each parameter's gradient sits in the flat buffer: W_0, b_0, W_1, b_1, ...
  float *gW[CSL_MAX_LAYERS], *gb[CSL_MAX_LAYERS];
  {
    int64_t at = 0;
    for (int j = 0; j < L; j++) {
      gW[j] = grads + at, at += (int64_t)dims[j + 1] * 2 * dims[j];
      gb[j] = grads + at, at += dims[j + 1];
    }
  }
  // ---- forward
  for (k = 0; k < L; k++) {
    const int32_t in = dims[k], out = dims[k + 1];
    const int64_t m = sl[k].n_out, mp = o.mp[k];
    const float* x = k == 0 ? feat : ws + o.y[k - 1];
    if (k == 0 && o.wpack >= 0) {
      // gather [self | mean] into LDS, multiply on the fp32 matrix cores, bias + ReLU on the way out; the operand is
      // also written (the weight gradient reads it), but never read back by the forward
      TSTEP(CSL_STEP_FUSED_FWD, csl_sage_fwd_mfma_f32(sl[0].indptr, sl[0].indices, sl[0].self_ids_in, feat_rows, feat, ldf, weights[0],
                                 2 * (int64_t)in, biases[0], m, mp, in, out, 0, L > 1 ? 1 : 0, ws + o.cat[0],
                                 2 * (int64_t)in, ws + o.y[0], out, ws + o.wpack, stream));
      continue;
    }
    TSTEP(CSL_STEP_AGGREGATION, csl_sage_cat_f32(sl[k].indptr, sl[k].indices, sl[k].self_ids_in, nullptr, nullptr, k == 0 ? feat_rows : nullptr, x,
                          k == 0 ? ldf : (int64_t)in, nullptr, 0, m, mp, ws + o.cat[k], 2 * (int64_t)in, in, 0, stream));
    TSTEP(CSL_STEP_GEMM, csl_gemm_f32(0, 1, mp, out, 2 * (int64_t)in, ws + o.cat[k], 2 * (int64_t)in, 0, weights[k], 2 * (int64_t)in, 0,
                      ws + o.y[k], out, 0, 1, biases[k], k + 1 < L ? 1 : 0, stream));
  }
  // the second stages, collected: (source, blocks, width, destination)
  // (a step defers at most 2 L + 1 second stages: see the static_assert next to CSL_MAX_LAYERS below)
  const float* r_src[CSL_REDUCE_MULTI_MAX];
  float* r_dst[CSL_REDUCE_MULTI_MAX];
  int64_t r_nblk[CSL_REDUCE_MULTI_MAX];
  int32_t r_h[CSL_REDUCE_MULTI_MAX];
  int nr = 0;
  auto defer = [&](const float* src, int64_t nblk, int32_t h, float* dst) {
    if (nr >= CSL_REDUCE_MULTI_MAX) return;   // (cannot happen: asserted at compile time)
    r_src[nr] = src, r_nblk[nr] = nblk, r_h[nr] = h, r_dst[nr] = dst;
    nr++;
  };
  // ---- loss and its gradient w.r.t. the logits (written as the top layer's padded gy), bias column sums alongside
  k = L - 1;
  {
    const int32_t C = dims[L];
    const int64_t m = sl[k].n_out;
    if (o.top_cols) {
      TSTEP(CSL_STEP_OTHER, csl_softmax_ce_partial_f32(ws + o.y[k], C, m, o.mp[k], C, seed_ids, nullptr, labels, scale, ws + o.gy[k], C,
                                      ws + o.lpart, ws + o.bpart[k], stream));
      defer(ws + o.bpart[k], o.bblocks[k], C, gb[k]);
    } else {
      TSTEP(CSL_STEP_OTHER, csl_softmax_ce_partial_f32(ws + o.y[k], C, m, m, C, seed_ids, nullptr, labels, scale, ws + o.g, C, ws + o.lpart,
                                      nullptr, stream));
      TSTEP(CSL_STEP_AGGREGATION, csl_relu_bwd_colsum_f32(ws + o.g, C, nullptr, 0, m, o.mp[k], ws + o.gy[k], C, gb[k], ws + o.scratch, C, stream));
    }
    defer(ws + o.lpart, o.top_cols ? o.lblocks : (m + 3) / 4, 1, loss);  // (blocks of four rows the loss pass covered)
  }
  // ---- backward
  for (k = L - 1; k >= 0; k--) {
    const int32_t in = dims[k], out = dims[k + 1];
    const int64_t mp = o.mp[k], wn = (int64_t)out * 2 * in;
    if (mp == 0) {
      if (hipMemsetAsync(gW[k], 0, sizeof(float) * wn, (hipStream_t)stream) != hipSuccess) return CSL_E_HIP;
    } else if (o.slabbed[k]) {
      const int64_t rs = mp / n_slabs;
      TSTEP(CSL_STEP_GEMM, csl_gemm_f32(1, 0, out, 2 * (int64_t)in, rs, ws + o.gy[k], out, rs * out, ws + o.cat[k], 2 * (int64_t)in,
                        rs * 2 * in, ws + o.slabs[k], 2 * (int64_t)in, wn, n_slabs, nullptr, 0, stream));
      defer(ws + o.slabs[k], n_slabs, (int32_t)wn, gW[k]);
    } else {
      TSTEP(CSL_STEP_GEMM, csl_gemm_f32(1, 0, out, 2 * (int64_t)in, mp, ws + o.gy[k], out, 0, ws + o.cat[k], 2 * (int64_t)in, 0, gW[k],
                        2 * (int64_t)in, 0, 1, nullptr, 0, stream));
    }
    if (k == 0) break;
    if (!sl[k].t_indptr || !sl[k].t_indices) {
      snprintf(s_err, sizeof(s_err), "layer %d has no slice by source (engine flag CSL_FLAG_TRANSPOSE)", k);
      return CSL_E_INVALID;
    }
    TSTEP(CSL_STEP_GEMM, csl_gemm_f32(0, 0, mp, 2 * (int64_t)in, out, ws + o.gy[k], out, 0, weights[k], 2 * (int64_t)in, 0, ws + o.gcat[k],
                      2 * (int64_t)in, 0, 1, nullptr, 0, stream));
    // gradient w.r.t. layer k-1's pre-activation output (= this layer's input x), padded like its GEMM operand;
    // its column sums (gb_{k-1}) stay as per-block partials
    if (o.hub[k]) {
      // a hub's list in the slice by source is thousands of entries, one wave's serial walk (1.3 ms instead of 30 us per
      // launch on a Zipf graph: profiles/hub_probe.py): those rows are summed by a workgroup per segment of entries
      TSTEP(CSL_STEP_AGGREGATION, csl_sage_cat_bwd_t_hub_f32(sl[k].t_indptr, sl[k].t_indices, sl[k].t_entries, sl[k].indptr,
                                                              ws + o.gcat[k], 2 * (int64_t)in, ws + o.y[k - 1], in, sl[k].n_in,
                                                              o.mp[k - 1], ws + o.gy[k - 1], in, nullptr, ws + o.bpart[k - 1],
                                                              in, stream));
    } else
    TSTEP(CSL_STEP_AGGREGATION, csl_sage_cat_bwd_t_f32(sl[k].t_indptr, sl[k].t_indices, sl[k].indptr, ws + o.gcat[k], 2 * (int64_t)in,
                                ws + o.y[k - 1], in, sl[k].n_in, o.mp[k - 1], ws + o.gy[k - 1], in, nullptr,
                                ws + o.bpart[k - 1], in, stream));
    if (o.bblocks[k - 1] > 0) defer(ws + o.bpart[k - 1], o.bblocks[k - 1], in, gb[k - 1]);
    else if (hipMemsetAsync(gb[k - 1], 0, sizeof(float) * in, (hipStream_t)stream) != hipSuccess) return CSL_E_HIP;
  }
  // ---- every deferred second stage (bias sums, weight-gradient slabs, the loss) in one launch
  k = -1;
  TSTEP(CSL_STEP_OTHER, csl_reduce_multi_f32(nr, r_src, r_nblk, r_h, r_dst, stream));
  return CSL_OK;
}

}  // extern "C"

// ===== one rank of the SPLIT-PARALLEL step (python/train.py + dist_sageconv.py:42-84 with several GPUs) =============
// The same sequencing for part g of P: per layer the partial sums of the boundary rows go to their owners and come
// back merged (pull_for_remotes / push_from_remotes, dist_sageconv.py:52-65), backward mirrors it.  The exchange itself
// is the CALLER's (a callback: torch.distributed's all_to_all_single over RCCL, or gloo in the tests) -- this file has no
// communicator; everything between two exchanges is issued from here, so a step costs the host the 2 L callbacks plus
// ~0.25 ms instead of ~1.2 ms of interpreter + autograd work, which is what bounds a rank once the GPU work is split N ways.
namespace {

struct RankLayout {
  int64_t x0, send[CSL_MAX_LAYERS], recv[CSL_MAX_LAYERS], agg[CSL_MAX_LAYERS], cat[CSL_MAX_LAYERS], y[CSL_MAX_LAYERS],
      gy[CSL_MAX_LAYERS], gcat[CSL_MAX_LAYERS], gx[CSL_MAX_LAYERS], g2[CSL_MAX_LAYERS], slabs[CSL_MAX_LAYERS], bpart[CSL_MAX_LAYERS],
      bblocks[CSL_MAX_LAYERS], mp[CSL_MAX_LAYERS];
  bool slabbed[CSL_MAX_LAYERS];
  int64_t lpart, lblocks, total;
  int64_t wpack;   // W_0 in MFMA operand order when the deepest layer has no boundary rows on this part (else -1)
};

bool rank_lay_out(int32_t L, const int32_t* dims, const csl_sage_rank_slice* sl, int64_t row_pad, int32_t n_slabs,
                  RankLayout& o) {
  if (L < 1 || L > CSL_MAX_LAYERS || n_slabs < 1) return false;
  int64_t at = 0;
  o.x0 = 0;   // (the deepest layer reads the resident feature rows through feat_rows: no gathered input matrix)
  for (int k = 0; k < L; k++) {
    const int64_t in = dims[k], out = dims[k + 1];
    const csl_sage_rank_slice& s = sl[k];
    if (in < 4 || in % 4 != 0 || out < 1 || (k + 1 < L && out % 4 != 0)) return false;
    if (s.n_out < 0 || s.n_in < 0 || s.n_owned < 0 || s.n_from < 0 || s.n_to < 0 || s.n_owned > s.n_out) return false;
    if (k > 0 && s.n_in != sl[k - 1].n_owned) return false;  // a layer's sources are the nodes this part owns below
    const int64_t mp = pad_rows(s.n_owned, row_pad);
    o.mp[k] = mp;
    o.send[k] = at, at += up4(s.n_from * in);   // forward: partial sums of peer-owned rows; backward: their gradients back
    o.recv[k] = at, at += up4(s.n_to * in);     // forward: partials received for owned rows; backward: their gradients
    o.agg[k] = at, at += up4(s.n_out * in);     // forward: merged sums (owned rows); backward: their gradient
    o.cat[k] = at, at += up4(mp * 2 * in);
    o.y[k] = at, at += up4(mp * out);
    o.gy[k] = at, at += up4(mp * out);
    o.gcat[k] = at, at += up4(mp * 2 * in);
    const bool by_src = k > 0 && s.t_indptr && s.t_indices;      // this layer's input gradient is gathered by source
    o.gx[k] = at, at += (k > 0 && !by_src) ? up4(s.n_in * in) : 0;
    o.g2[k] = at, at += by_src ? up4(s.n_out * 2 * in) : 0;
    const int64_t wn = out * 2 * in;
    o.slabbed[k] = row_pad > 0 && mp >= row_pad && n_slabs > 1 && mp % n_slabs == 0 && wn % 4 == 0;
    o.slabs[k] = at, at += o.slabbed[k] ? up4(wn * n_slabs) : 0;
    // gb_k's first stage is written by layer k+1's backward: the mask pass behind the atomic scatter, or the gather by source
    const bool src_above = k + 1 < L && sl[k + 1].t_indptr && sl[k + 1].t_indices;
    const int64_t part = k + 1 < L ? (!src_above ? csl_relu_bwd_colsum_scratch(mp, (int32_t)out)
                                     : sl[k + 1].t_max_len > CSL_T_SORTED_MAX ? csl_sage_cat_bwd_t_hub_scratch(mp, (int32_t)out)
                                                                              : csl_sage_cat_bwd_t_scratch(mp, (int32_t)out)) : 0;
    o.bblocks[k] = part / out;
    o.bpart[k] = at, at += up4(part);
  }
  const int64_t C = dims[L];
  if (C > 256) return false;  // (the loss pass leaves the bias column sums: csl_softmax_ce_partial_f32)
  o.lblocks = (o.mp[L - 1] + 3) / 4;
  o.lpart = at, at += up4(o.lblocks);
  o.bblocks[L - 1] = o.lblocks;
  o.bpart[L - 1] = at, at += up4(o.lblocks * C);
  {
    // a deepest layer WITHOUT boundary rows on this part (every out row owned, nothing sent or received: a world of one,
    // or a partition that keeps a minibatch's neighbourhoods local) is the single-GPU layer: one fused kernel
    const csl_sage_rank_slice& s0 = sl[0];
    const bool local = s0.n_from == 0 && s0.n_to == 0 && s0.n_owned == s0.n_out && !getenv("CSLICER_NO_MFMA_FWD");
    const int64_t wp = local ? csl_sage_fwd_mfma_scratch(dims[0], dims[1]) : -1;
    o.wpack = wp > 0 ? at : -1;
    if (wp > 0) at += up4(wp);
  }
  o.total = at;
  return true;
}

}  // namespace

extern "C" {

int64_t csl_sage_rank_workspace(int32_t n_layers, const int32_t* dims, const csl_sage_rank_slice* slices, int64_t row_pad,
                                int32_t n_slabs) {
  RankLayout o;
  if (!dims || !slices || !rank_lay_out(n_layers, dims, slices, row_pad, n_slabs, o)) return CSL_E_INVALID;
  return o.total;
}

int csl_sage_rank_fwd_bwd_f32(int32_t n_layers, const int32_t* dims, const csl_sage_rank_slice* sl,
                              const float* const* weights, const float* const* biases, const float* feat, int64_t ldf,
                              const int32_t* feat_rows, const int32_t* seed_ids, const int32_t* label_rows,
                              const int64_t* labels, float scale, int64_t row_pad, int32_t n_slabs,
                              csl_exchange_fn exchange, csl_exchange_wait_fn wait, void* user, float* grads, float* loss,
                              float* workspace, int64_t workspace_floats, void* stream) {
  int k = -1;
  s_err[0] = 0;
  RankLayout o;
  if (!dims || !sl || !weights || !biases || !grads || !loss || !exchange ||
      !rank_lay_out(n_layers, dims, sl, row_pad, n_slabs, o)) {
    snprintf(s_err, sizeof(s_err), "bad argument or unsupported model (widths multiples of 4, <= 256 classes, 1..%d layers, "
             "n_in of a layer = n_owned of the layer below)", CSL_MAX_LAYERS);
    return CSL_E_INVALID;
  }
  if (o.total > workspace_floats || (o.total > 0 && (!workspace || ((uintptr_t)workspace & 15)))) {
    snprintf(s_err, sizeof(s_err), "workspace: %lld floats needed, %lld given", (long long)o.total, (long long)workspace_floats);
    return CSL_E_INVALID;
  }
  const int L = n_layers;
  float* ws = workspace;
  float *gW[CSL_MAX_LAYERS], *gb[CSL_MAX_LAYERS];
  {
    int64_t at = 0;
    for (int j = 0; j < L; j++) {
      gW[j] = grads + at, at += (int64_t)dims[j + 1] * 2 * dims[j];
      gb[j] = grads + at, at += dims[j + 1];
    }
  }
#define XCHG(layer, backward, src, dst, width)                                                     \
  do {                                                                                             \
    const int rc_ = exchange(user, (layer), (backward), (src), (dst), (width), stream);            \
    if (rc_ < 0) {                                                                                 \
      snprintf(s_err, sizeof(s_err), "the exchange callback failed (%d), layer %d, %s", rc_, (layer), \
               (backward) ? "backward" : "forward");                                               \
      return rc_;                                                                                  \
    }                                                                                              \
  } while (0)
  // `wait` given: `exchange` only STARTS the exchange (on a stream of the caller's) and `wait` makes `stream` wait for
  // it -- called right before the received rows are first used, so the rows that never leave the GPU are aggregated
  // while the boundary rows travel (dist_sageconv.py:57-64 on a side stream)
#define XWAIT(layer, backward)                                                                     \
  do {                                                                                             \
    if (wait) {                                                                                    \
      const int rc_ = wait(user, (layer), (backward), stream);                                     \
      if (rc_ < 0) {                                                                               \
        snprintf(s_err, sizeof(s_err), "the exchange-wait callback failed (%d), layer %d", rc_, (layer)); \
        return rc_;                                                                                \
      }                                                                                            \
    }                                                                                              \
  } while (0)
  // ---- forward
  for (k = 0; k < L; k++) {
    const int32_t in = dims[k], out = dims[k + 1];
    const csl_sage_rank_slice& s = sl[k];
    // the deepest layer reads the resident feature rows through feat_rows (no gathered input matrix)
    const float* x = k == 0 ? feat : ws + o.y[k - 1];
    const int64_t ldx = k == 0 ? ldf : (int64_t)in;
    const int32_t* map = k == 0 ? feat_rows : nullptr;
    if (k == 0 && o.wpack >= 0) {
      // no boundary rows: the single-GPU layer as one kernel (gather -> fp32 MFMA -> bias + ReLU); the CSR degree IS the
      // true degree, the owned rows ARE the out rows
      STEP(csl_sage_fwd_mfma_f32(s.indptr, s.indices, s.self_ids_in, feat_rows, feat, ldf, weights[0], 2 * (int64_t)in,
                                 biases[0], s.n_owned, o.mp[0], in, out, 0, L > 1 ? 1 : 0, ws + o.cat[0], 2 * (int64_t)in,
                                 ws + o.y[0], out, ws + o.wpack, stream));
      continue;
    }
    // partial sums of the rows peers own, straight into the send buffer; then the rows this part owns
    STEP(csl_spmm_sum_map_f32(s.indptr, s.indices, s.from_all, s.n_from, x, ldx, map, ws + o.send[k], in, in, 1, stream));
    XCHG(k, 0, ws + o.send[k], ws + o.recv[k], in);
    STEP(csl_spmm_sum_map_f32(s.indptr, s.indices, s.owned_out_nodes, s.n_owned, x, ldx, map, ws + o.agg[k], in, in, 0, stream));
    XWAIT(k, 0);
    STEP(csl_scatter_add_rows_atomic_f32(ws + o.agg[k], in, s.to_all, s.n_to, ws + o.recv[k], in, in, stream));
    STEP(csl_sage_cat_f32(nullptr, nullptr, s.self_ids_in, s.owned_out_nodes, s.owned_degree, map, x, ldx, ws + o.agg[k],
                          in, s.n_owned, o.mp[k], ws + o.cat[k], 2 * (int64_t)in, in, 0, stream));
    STEP(csl_gemm_f32(0, 1, o.mp[k], out, 2 * (int64_t)in, ws + o.cat[k], 2 * (int64_t)in, 0, weights[k], 2 * (int64_t)in, 0,
                      ws + o.y[k], out, 0, 1, biases[k], k + 1 < L ? 1 : 0, stream));
  }
  // (a step defers at most 2 L + 1 second stages: see the static_assert next to CSL_MAX_LAYERS below)
  const float* r_src[CSL_REDUCE_MULTI_MAX];
  float* r_dst[CSL_REDUCE_MULTI_MAX];
  int64_t r_nblk[CSL_REDUCE_MULTI_MAX];
  int32_t r_h[CSL_REDUCE_MULTI_MAX];
  int nr = 0;
  auto defer = [&](const float* src, int64_t nblk, int32_t h, float* dst) {
    if (nr >= CSL_REDUCE_MULTI_MAX) return;   // (cannot happen: asserted at compile time)
    r_src[nr] = src, r_nblk[nr] = nblk, r_h[nr] = h, r_dst[nr] = dst;
    nr++;
  };
  // ---- loss over the seeds this part owns (the caller's `scale` = 1 / seeds of the WHOLE minibatch)
  k = L - 1;
  {
    const int32_t C = dims[L];
    STEP(csl_softmax_ce_partial_f32(ws + o.y[k], C, sl[k].n_owned, o.mp[k], C, seed_ids, label_rows, labels, scale,
                                    ws + o.gy[k], C, ws + o.lpart, ws + o.bpart[k], stream));
    defer(ws + o.bpart[k], o.bblocks[k], C, gb[k]);
    defer(ws + o.lpart, o.lblocks, 1, loss);
  }
  // ---- backward
  for (k = L - 1; k >= 0; k--) {
    const int32_t in = dims[k], out = dims[k + 1];
    const csl_sage_rank_slice& s = sl[k];
    const int64_t mp = o.mp[k], wn = (int64_t)out * 2 * in;
    if (mp == 0) {
      if (hipMemsetAsync(gW[k], 0, sizeof(float) * wn, (hipStream_t)stream) != hipSuccess) return CSL_E_HIP;
    } else if (o.slabbed[k]) {
      const int64_t rs = mp / n_slabs;
      STEP(csl_gemm_f32(1, 0, out, 2 * (int64_t)in, rs, ws + o.gy[k], out, rs * out, ws + o.cat[k], 2 * (int64_t)in,
                        rs * 2 * in, ws + o.slabs[k], 2 * (int64_t)in, wn, n_slabs, nullptr, 0, stream));
      defer(ws + o.slabs[k], n_slabs, (int32_t)wn, gW[k]);
    } else {
      STEP(csl_gemm_f32(1, 0, out, 2 * (int64_t)in, mp, ws + o.gy[k], out, 0, ws + o.cat[k], 2 * (int64_t)in, 0, gW[k],
                        2 * (int64_t)in, 0, 1, nullptr, 0, stream));
    }
    if (k == 0) break;   // (no gradient flows into the input features; the deepest layer's exchange has no backward)
    STEP(csl_gemm_f32(0, 0, mp, 2 * (int64_t)in, out, ws + o.gy[k], out, 0, weights[k], 2 * (int64_t)in, 0, ws + o.gcat[k],
                      2 * (int64_t)in, 0, 1, nullptr, 0, stream));
    if (s.t_indptr && s.t_indices) {
      // BY SOURCE: the operand gradient goes into out-row order (mean half / true degree), the rows peers own get theirs
      // from the reverse exchange, and ONE gather over the part's slice by source writes the input gradient with the ReLU
      // mask of the layer below, its padding and its bias sums -- no atomics, no zero fill
      float* g2 = ws + o.g2[k];
      STEP(csl_sage_rank_g2_f32(s.owned_out_nodes, s.owned_degree, s.n_owned, ws + o.gcat[k], 2 * (int64_t)in, g2, 2 * (int64_t)in,
                                in, stream));
      // what this part received forward gets its gradient back (the mean halves of the owned rows it was merged into)
      STEP(csl_gather_rows_f32(g2 + in, 2 * (int64_t)in, s.to_all, s.n_to, ws + o.recv[k], in, in, stream));
      XCHG(k, 1, ws + o.recv[k], ws + o.send[k], in);
      XWAIT(k, 1);
      STEP(csl_scatter_rows_f32(g2 + in, 2 * (int64_t)in, s.from_all, s.n_from, ws + o.send[k], in, in, stream));
      if (o.mp[k - 1] > 0) {
        if (s.t_max_len > CSL_T_SORTED_MAX)
          STEP(csl_sage_cat_bwd_t_hub_f32(s.t_indptr, s.t_indices, s.t_entries, nullptr, g2, 2 * (int64_t)in, ws + o.y[k - 1], in,
                                          s.n_in, o.mp[k - 1], ws + o.gy[k - 1], in, nullptr, ws + o.bpart[k - 1], in, stream));
        else
          STEP(csl_sage_cat_bwd_t_f32(s.t_indptr, s.t_indices, nullptr, g2, 2 * (int64_t)in, ws + o.y[k - 1], in, s.n_in,
                                      o.mp[k - 1], ws + o.gy[k - 1], in, nullptr, ws + o.bpart[k - 1], in, stream));
      }
      defer(ws + o.bpart[k - 1], o.mp[k - 1] > 0 ? o.bblocks[k - 1] : 0, in, gb[k - 1]);
      continue;
    }
    // operand gradient -> self rows of gx and owned rows of the merged sums' gradient (both zeroed inside)
    STEP(csl_sage_cat_rows_bwd_f32(s.self_ids_in, s.owned_out_nodes, s.owned_degree, s.n_owned, ws + o.gcat[k],
                                   2 * (int64_t)in, ws + o.gx[k], s.n_in, ws + o.agg[k], s.n_out, in, stream));
    // what this part received forward gets its gradient back; the reverse exchange returns the gradients of the
    // partial sums this part sent
    STEP(csl_gather_rows_f32(ws + o.agg[k], in, s.to_all, s.n_to, ws + o.recv[k], in, in, stream));
    XCHG(k, 1, ws + o.recv[k], ws + o.send[k], in);
    STEP(csl_spmm_sum_bwd_f32(s.indptr, s.indices, s.owned_out_nodes, s.n_owned, ws + o.agg[k], in, 0, ws + o.gx[k], in, in,
                              stream));
    XWAIT(k, 1);
    STEP(csl_spmm_sum_bwd_f32(s.indptr, s.indices, s.from_all, s.n_from, ws + o.send[k], in, 1, ws + o.gx[k], in, in,
                              stream));
    // ReLU mask of the layer below, padding of its GEMM operand, its bias column sums (first stage)
    if (o.mp[k - 1] > 0)
      STEP(csl_relu_bwd_colsum_f32(ws + o.gx[k], in, ws + o.y[k - 1], in, s.n_in, o.mp[k - 1], ws + o.gy[k - 1], in,
                                   nullptr, ws + o.bpart[k - 1], in, stream));
    defer(ws + o.bpart[k - 1], o.mp[k - 1] > 0 ? o.bblocks[k - 1] : 0, in, gb[k - 1]);
  }
  k = -1;
  STEP(csl_reduce_multi_f32(nr, r_src, r_nblk, r_h, r_dst, stream));
#undef XCHG
#undef XWAIT
  return CSL_OK;
}

}  // extern "C"

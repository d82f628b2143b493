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
//   loss      csl_softmax_ce_f32 on y_{L-1} (forward and gradient in one pass)
//   backward  gW_k  = gy_k^T cat_k (row slabs + sum)           csl_gemm_f32 (batched) + csl_sum_slabs_f32
//             gcat  = gy_k W_k                                 csl_gemm_f32
//             gy_{k-1}, gb_{k-1} = gather of gcat over the slice by source, ReLU mask of y_{k-1}, row padding and
//                                  bias column sums in the same pass                   csl_sage_cat_bwd_t_f32
// Rows are padded to a multiple of `row_pad` so that GEMM shapes repeat from minibatch to minibatch.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "cslicer_aggr.h"
#include "cslicer_hip.h"

namespace {

thread_local char s_err[200];

inline int64_t pad_rows(int64_t m, int64_t row_pad) {
  return (row_pad > 0 && m >= row_pad) ? (m + row_pad - 1) / row_pad * row_pad : m;
}
inline int64_t up4(int64_t x) { return (x + 3) & ~(int64_t)3; }  // every buffer starts 16-byte aligned

struct Layout {
  int64_t cat[CSL_MAX_LAYERS], y[CSL_MAX_LAYERS], gy[CSL_MAX_LAYERS], gcat[CSL_MAX_LAYERS], mp[CSL_MAX_LAYERS];
  int64_t g, slabs, scratch, total;
};

// bump allocation of the step's buffers (floats); returns false for an unsupported model
bool lay_out(int32_t L, const int32_t* dims, const csl_sage_slice* sl, int64_t row_pad, int32_t n_slabs, Layout& o) {
  if (L < 1 || L > CSL_MAX_LAYERS || n_slabs < 1) return false;
  int64_t at = 0, scratch = 0, slabs = 0;
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
    const int64_t s1 = csl_relu_bwd_colsum_scratch(mp, (int32_t)out), s2 = csl_sage_cat_bwd_t_scratch(mp, (int32_t)out);
    if (s1 > scratch) scratch = s1;
    if (s2 > scratch) scratch = s2;
    if (out * 2 * in * n_slabs > slabs) slabs = out * 2 * in * n_slabs;
  }
  const int64_t s3 = csl_softmax_ce_scratch(sl[L - 1].n_out);
  if (s3 > scratch) scratch = s3;
  o.g = at, at += up4(sl[L - 1].n_out * dims[L]);
  o.slabs = at, at += up4(slabs);
  o.scratch = at, at += up4(scratch);
  o.total = at;
  return true;
}

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
    STEP(csl_sage_cat_f32(sl[k].indptr, sl[k].indices, sl[k].self_ids_in, nullptr, nullptr, k == 0 ? feat_rows : nullptr, x,
                          k == 0 ? ldf : (int64_t)in, nullptr, 0, m, mp, ws + o.cat[k], 2 * (int64_t)in, in, 0, stream));
    STEP(csl_gemm_f32(0, 1, mp, out, 2 * (int64_t)in, ws + o.cat[k], 2 * (int64_t)in, 0, weights[k], 2 * (int64_t)in, 0,
                      ws + o.y[k], out, 0, 1, biases[k], k + 1 < L ? 1 : 0, stream));
  }
  // ---- loss and its gradient w.r.t. the logits; padded + bias column sums for the top layer
  k = L - 1;
  {
    const int32_t C = dims[L];
    const int64_t m = sl[k].n_out;
    STEP(csl_softmax_ce_f32(ws + o.y[k], C, m, C, seed_ids, nullptr, labels, scale, loss, ws + o.g, C, ws + o.scratch, stream));
    STEP(csl_relu_bwd_colsum_f32(ws + o.g, C, nullptr, 0, m, o.mp[k], ws + o.gy[k], C, gb[k], ws + o.scratch, C, stream));
  }
  // ---- backward
  for (k = L - 1; k >= 0; k--) {
    const int32_t in = dims[k], out = dims[k + 1];
    const int64_t mp = o.mp[k], wn = (int64_t)out * 2 * in;
    if (mp == 0) {
      if (hipMemsetAsync(gW[k], 0, sizeof(float) * wn, (hipStream_t)stream) != hipSuccess) return CSL_E_HIP;
    } else if (row_pad > 0 && mp >= row_pad && mp % n_slabs == 0 && wn % 4 == 0 && n_slabs > 1 && ((uintptr_t)gW[k] & 15) == 0) {
      const int64_t rs = mp / n_slabs;
      STEP(csl_gemm_f32(1, 0, out, 2 * (int64_t)in, rs, ws + o.gy[k], out, rs * out, ws + o.cat[k], 2 * (int64_t)in,
                        rs * 2 * in, ws + o.slabs, 2 * (int64_t)in, wn, n_slabs, nullptr, 0, stream));
      STEP(csl_sum_slabs_f32(ws + o.slabs, wn, n_slabs, gW[k], stream));
    } else {
      STEP(csl_gemm_f32(1, 0, out, 2 * (int64_t)in, mp, ws + o.gy[k], out, 0, ws + o.cat[k], 2 * (int64_t)in, 0, gW[k],
                        2 * (int64_t)in, 0, 1, nullptr, 0, stream));
    }
    if (k == 0) break;
    if (!sl[k].t_indptr || !sl[k].t_indices) {
      snprintf(s_err, sizeof(s_err), "layer %d has no slice by source (engine flag CSL_FLAG_TRANSPOSE)", k);
      return CSL_E_INVALID;
    }
    STEP(csl_gemm_f32(0, 0, mp, 2 * (int64_t)in, out, ws + o.gy[k], out, 0, weights[k], 2 * (int64_t)in, 0, ws + o.gcat[k],
                      2 * (int64_t)in, 0, 1, nullptr, 0, stream));
    // gradient w.r.t. layer k-1's pre-activation output (= this layer's input x), padded like its GEMM operand
    STEP(csl_sage_cat_bwd_t_f32(sl[k].t_indptr, sl[k].t_indices, sl[k].indptr, ws + o.gcat[k], 2 * (int64_t)in,
                                ws + o.y[k - 1], in, sl[k].n_in, o.mp[k - 1], ws + o.gy[k - 1], in, gb[k - 1],
                                ws + o.scratch, in, stream));
  }
  return CSL_OK;
}

}  // extern "C"

/* cslicer_aggr.h -- C ABI of the split-parallel aggregation kernels that consume the
 * slices (CSL_MODE_GRAPH) on each GPU.  Part of libcslicer_hip.so.
 *
 * Reference semantics (the Python split trainer, DGL-based, and the older CUDA prototype):
 *   csl_spmm_sum_f32          BipartiteGraph.gather            python/data/bipartite.py:61-67
 *                             aggregate_nodeWise               src/gnn/sage.cu:7-18
 *   csl_spmm_sum_bwd_f32      aggregate_edgeWise (backward)    src/gnn/sage.cu:20-28
 *   csl_gather_rows_f32       pull_for_remotes / self_gather   python/data/bipartite.py:82-91
 *   csl_scatter_add_rows_f32  push_from_remotes / mergeKernel  python/data/bipartite.py:93-99,
 *                                                              src/gnn/dist_sage.cu:193-199
 *   csl_div_rows_f32          mean normalisation by the true degree (the reference averages
 *                             per-GPU means, bipartite.py:98; sums + one division is the
 *                             mathematically intended mean)
 *
 *   csl_gat_fwd_f32 /         attention aggregation of BASELINE config 5 (GAT).  The reference has only
 *   csl_gat_bwd_f32           BipartiteGraph.attention_gather (bipartite.py:75-80) and a stub layer
 *                             (layers/dist_gatconv.py:3-6): "parity unpinned", defined against torch.
 *
 * All pointers are DEVICE pointers; index arrays are int32 (the slices' device type); `stream`
 * is a hipStream_t.  fp32 throughout.  Returns 0 or a negative CSL_E_* code (cslicer_hip.h).
 */
#ifndef CSLICER_AGGR_H
#define CSLICER_AGGR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* out[r, :] = sum over e in [indptr[row], indptr[row+1]) of x[indices[e], :]
 * row = rows ? rows[r] : r, r in [0, n_rows).  x has `ldx` floats per row, out `ldo`; H <= ldx, ldo. */
int csl_spmm_sum_f32(const int32_t* indptr, const int32_t* indices, const int32_t* rows, int64_t n_rows,
                     const float* x, int64_t ldx, float* out, int64_t ldo, int32_t H, void* stream);

/* grad_x[indices[e], :] += grad_out[q, :] for every edge e of row = rows ? rows[r] : r,
 * r in [0, n_rows); q = compact ? r : row (compact: grad_out holds only the listed rows). fp32 atomics. */
int csl_spmm_sum_bwd_f32(const int32_t* indptr, const int32_t* indices, const int32_t* rows, int64_t n_rows,
                         const float* grad_out, int64_t ldg, int32_t compact, float* grad_x, int64_t ldx, int32_t H,
                         void* stream);

/* dst[k, :] = idx[k] >= 0 ? src[idx[k], :] : 0 */
int csl_gather_rows_f32(const float* src, int64_t lds, const int32_t* idx, int64_t n, float* dst, int64_t ldd,
                        int32_t H, void* stream);

/* dst[idx[k], :] += src[k, :]; idx must not repeat inside one call */
int csl_scatter_add_rows_f32(float* dst, int64_t ldd, const int32_t* idx, int64_t n, const float* src, int64_t lds,
                             int32_t H, void* stream);

/* x[k, :] /= max(deg[k], 1) */
int csl_div_rows_f32(float* x, int64_t ldx, const int32_t* deg, int64_t n, int32_t H, void* stream);

/* GAT partial attention over one slice's CSR (sources owned by this part), per destination row r, head h:
 *   score_e = LeakyReLU(el[src_e,h] + er[r,h]; slope);  m = max_e score_e (-1e30 for an empty row);
 *   s = sum_e exp(score_e - m);  n[r,h,:] = sum_e exp(score_e - m) * z[src_e,h,:].
 * el [n_src,H], er/m/s [n_rows,H], z [n_src,H*D], n [n_rows,H*D], all dense row-major, 16-byte aligned;
 * D % 4 == 0, D <= 256.  The owner of r merges the parts' (m,s,n) and divides. */
int csl_gat_fwd_f32(const int32_t* indptr, const int32_t* indices, int64_t n_rows, const float* el, const float* er,
                    const float* z, int32_t H, int32_t D, float slope, float* m_out, float* s_out, float* n_out,
                    void* stream);

/* Gradients of (s, n) above: g_el [n_src,H] and g_z [n_src,H*D] are ACCUMULATED with fp32 atomics (zero them
 * first), g_er [n_rows,H] is written.  m_in is the forward's m (a stabiliser, not differentiated). */
int csl_gat_bwd_f32(const int32_t* indptr, const int32_t* indices, int64_t n_rows, const float* el, const float* er,
                    const float* z, int32_t H, int32_t D, float slope, const float* m_in, const float* g_s,
                    const float* g_n, float* g_el, float* g_er, float* g_z, void* stream);

#ifdef __cplusplus
}
#endif
#endif

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

/* the same with the k-th listed row written to out[k] (rows != NULL): a send buffer of boundary partial sums */
int csl_spmm_sum_compact_f32(const int32_t* indptr, const int32_t* indices, const int32_t* rows, int64_t n_rows,
                             const float* x, int64_t ldx, float* out, int64_t ldo, int32_t H, void* stream);

/* csl_spmm_sum_f32 / _compact_f32 (compact != 0) over a RESIDENT table: the source row of index s is x[rowmap[s]] (rowmap
 * NULL: x[s]).  The deepest layer of a rank reads its feature rows in place (python/data/bipartite.py:61-67 on the
 * owner's feature shard) instead of gathering them into an input matrix first. */
int csl_spmm_sum_map_f32(const int32_t* indptr, const int32_t* indices, const int32_t* rows, int64_t n_rows, const float* x,
                         int64_t ldx, const int32_t* rowmap, float* out, int64_t ldo, int32_t H, int32_t compact, void* stream);

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

/* the same where idx MAY repeat (push_from_remotes for all peers in one launch: several parts send a partial sum
 * for the same owned node): fp32 atomics, the order of the additions is not deterministic */
int csl_scatter_add_rows_atomic_f32(float* dst, int64_t ldd, const int32_t* idx, int64_t n, const float* src, int64_t lds,
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


/* The same gradients BY SOURCE, over the slicer's slice by source (cslicer_hip.h CSL_T_INDPTR / CSL_T_INDICES with
 * CSL_FLAG_TRANSPOSE_ALL; self entries, negative, are skipped): g_z [n_pad, H*D] and g_el [n_src, H] are WRITTEN (each
 * row once: nothing to pre-zero, no atomics on them; rows [n_src, n_pad) of g_z are zeroed), g_er [n_rows, H] is
 * ACCUMULATED with fp32 atomics (zero it first: H floats per edge instead of H*D). */
int csl_gat_bwd_t_f32(const int32_t* t_indptr, const int32_t* t_indices, int64_t n_src, int64_t n_pad, const float* el,
                      const float* er, const float* z, int32_t H, int32_t D, float slope, const float* m_in,
                      const float* g_s, const float* g_n, float* g_el, float* g_er, float* g_z, void* stream);

/* csl_gat_bwd_t_f32 with the attention logits' backward (csl_gat_logits_bwd_acc_f32) folded in: g_z [n_pad, H*D] is
 * WRITTEN complete -- the aggregation's share + g_el a_l (inside the pass over the source rows, where z[u] and the finished
 * g_el[u] are in registers) + g_er a_r (a second small pass over the n_out destination rows, whose z row is their self
 * source self_ids_in[r]) --, g_attn_l / g_attn_r [H, D] are the sums over the rows (two-stage), g_er_out [n_out, H] is
 * zeroed and accumulated here (an output for callers that want it).  No pass re-reads z or read-modify-writes all of g_z.
 * scratch: csl_gat_bwd_t_fused_scratch(n_pad, n_out, H, D) floats. */
int64_t csl_gat_bwd_t_fused_scratch(int64_t n_pad, int64_t n_out, int32_t H, int32_t D);
int csl_gat_bwd_t_fused_f32(const int32_t* t_indptr, const int32_t* t_indices, int64_t n_src, int64_t n_pad, const float* el,
                            const float* er_out, const float* z, int32_t H, int32_t D, float slope, const float* m_in,
                            const float* g_s, const float* g_n, const float* attn_l, const float* attn_r,
                            const int32_t* self_ids_in, int64_t n_out, float* g_er_out, float* g_z, float* g_attn_l,
                            float* g_attn_r, float* scratch, void* stream);

/* ---- the attention model's INPUT layer, aggregate-then-project (csrc/gat_input.hip) ----
 * DistGATConv on a layer whose input takes no gradient (the feature table): logits and weighted sum are linear in x, so
 *   v_l[h] = W_h^T attn_l[h], v_r likewise [H, F];  el[u, h] = <x[u], v_l[h]>;  er[v, h] = <x[v], v_r[h]>
 *   alpha_h(u -> v) = softmax over the sampled in-edges of v of LeakyReLU(el[u, h] + er[v, h])
 *   agg[v, h, :] = sum_u alpha_h(u -> v) x[u]           (then out[v, h, :] = W_h agg[v, h, :] + bias: a batched csl_gemm_f32)
 * give the same layer without projecting the sources (python/data/bipartite.py:75-80 is the reference's only piece of it).
 * csl_gat_in_fwd_f32: agg [n_out, H * F] and alpha [n_edges, H] (kept for the backward; its sign bit = the logit was <= 0)
 * from the slice's CSR (indptr [n_out + 1], indices [n_edges], self_ids [n_out]: the destination's own source row or -1)
 * over x read through rowmap (source s = row rowmap[s] of x; NULL: row s).  H in {1, 2, 4, 8}; F % 4 == 0, F <= 128;
 * every row has at most max_deg <= csl_gat_in_max_degree() = 32 edges (the slicer's fanout; the kernel instance is chosen by
 * it and keeps a row's feature rows in registers; longer rows would be CUT, so the caller must not pass them); x / v_l / v_r / agg 16-byte aligned, ldx % 4 == 0.
 * csl_gat_in_bwd_f32: g_vl, g_vr [H, F] from dagg (the gradient of agg; element [r, h, f] at r * ld_r + h * ld_h + f):
 *   dalpha = <dagg[r, h], x[u]>;  dlogit = alpha (dalpha - sum_e alpha dalpha) * LeakyReLU';
 *   g_vl[h] = sum_e dlogit x[u_e];  g_vr[h] = sum_r (sum_e dlogit) x[self(r)]     (two-stage sums, no atomics).
 * scratch: csl_gat_in_bwd_scratch(n_out, H, F) floats. */
int32_t csl_gat_in_max_degree(void);
int csl_gat_in_fwd_f32(const int32_t* indptr, const int32_t* indices, const int32_t* self_ids, const int32_t* rowmap,
                       const float* x, int64_t ldx, int32_t F, const float* vl, const float* vr, int32_t H, float slope,
                       int64_t n_out, int64_t n_edges, int32_t max_deg, float* agg, float* alpha, void* stream);
int64_t csl_gat_in_bwd_scratch(int64_t n_out, int32_t H, int32_t F);
int csl_gat_in_bwd_f32(const int32_t* indptr, const int32_t* indices, const int32_t* self_ids, const int32_t* rowmap,
                       const float* x, int64_t ldx, int32_t F, const float* alpha, const float* dagg, int64_t ld_r,
                       int64_t ld_h, int32_t H, float slope, int64_t n_out, int64_t n_edges, int32_t max_deg, float* g_vl,
                       float* g_vr, float* scratch, void* stream);
/* The block-diagonal projection of the destinations and its gradients on the fp32 matrix cores (v_mfma_f32_16x16x4_f32; a
 * wave owns a head, W_h stationary in registers, operands straight from global memory in operand order):
 *   csl_gat_in_proj_f32:      out[r, h*D + d] = act(sum_f agg[r, h*F + f] W[h*D + d, f] + bias[h*D + d])   (act: ELU if elu)
 *   csl_gat_in_proj_bwd_f32:  dagg[r, h*FP + f] = sum_d gg[r, h*D + d] W[h*D + d, f]   (FP = csl_gat_in_proj_fpad(F): whole
 *                             16-column tiles per head, columns >= F zero: pass ld_r = H * FP, ld_h = FP to csl_gat_in_bwd_f32)
 *                             gW[h*D + d, f] = sum_r gg[r, h*D + d] agg[r, h*F + f]
 * agg [n, H*F], W [H*D, F] (torch's Linear.weight), gg [n, ldg].  csl_gat_in_proj_ok(H, F, D): H in {1,2,4,8}, F % 4 == 0,
 * F <= 128, D in {16, 32, 64}; other shapes: the same three products as strided-batched csl_gemm_f32 calls.
 * scratch: csl_gat_in_proj_bwd_scratch(H, F, D) floats. */
int32_t csl_gat_in_proj_ok(int32_t H, int32_t F, int32_t D);
int32_t csl_gat_in_proj_fpad(int32_t F);
int csl_gat_in_proj_f32(const float* agg, const float* W, const float* bias, int64_t n, int32_t H, int32_t F, int32_t D,
                        int32_t elu, float* out, int64_t ldo, void* stream);
int64_t csl_gat_in_proj_bwd_scratch(int32_t H, int32_t F, int32_t D);
int csl_gat_in_proj_bwd_f32(const float* gg, int64_t ldg, const float* agg, const float* W, int64_t n, int32_t H, int32_t F,
                            int32_t D, float* dagg, float* gW, float* scratch, void* stream);
/* The whole input layer as ONE call per direction (what aggr.GatInputLayer issues where csl_gat_in_proj_ok holds):
 *   forward : v_l / v_r from W and attn_* (scratch: csl_gat_in_layer_fwd_scratch(H, F) floats), csl_gat_in_fwd_f32,
 *             csl_gat_in_proj_f32 -> agg [n_out, H*F], alpha [n_edges, H] (both kept for the backward), out [n_out, ldo]
 *   backward: from g [n_out, ldg] (the gradient of out): ELU' and the bias gradient, dagg / gW (csl_gat_in_proj_bwd_f32), the
 *             edge pass (csl_gat_in_bwd_f32), ONE second-stage launch for all their sums, then the chain rule through
 *             v = W_h^T a -> gW [H*D, F] (complete), g_al / g_ar [H, D], g_bias [H*D].  gg [n_out, H*D] and
 *             dagg [n_out, H * csl_gat_in_proj_fpad(F)] are work buffers; scratch: csl_gat_in_layer_bwd_scratch(...) floats. */
int64_t csl_gat_in_layer_fwd_scratch(int32_t H, int32_t F);
int csl_gat_in_layer_fwd_f32(const int32_t* indptr, const int32_t* indices, const int32_t* self_ids, const int32_t* rowmap,
                             const float* x, int64_t ldx, int32_t F, const float* W, const float* attn_l, const float* attn_r,
                             const float* bias, int32_t H, int32_t D, float slope, int32_t elu, int64_t n_out, int64_t n_edges,
                             int32_t max_deg, float* agg, float* alpha, float* out, int64_t ldo, float* scratch, void* stream);
int64_t csl_gat_in_layer_bwd_scratch(int64_t n_out, int32_t H, int32_t F, int32_t D);
int csl_gat_in_layer_bwd_f32(const int32_t* indptr, const int32_t* indices, const int32_t* self_ids, const int32_t* rowmap,
                             const float* x, int64_t ldx, int32_t F, const float* W, const float* attn_l, const float* attn_r,
                             int32_t H, int32_t D, float slope, int32_t elu, int64_t n_out, int64_t n_edges, int32_t max_deg,
                             const float* agg, const float* alpha, const float* out, int64_t ldo, const float* g, int64_t ldg,
                             float* gg, float* dagg, float* gW, float* g_al, float* g_ar, float* g_bias, float* scratch,
                             void* stream);
/* y[r, 0:C) = act(y[r, 0:C) + bias) in place for r < n (act = ELU, alpha 1, when elu != 0; C % 4 == 0), and its
 * backward: out[r, :] = g[r, :] * act'(y[r, :]) (from the activation's OUTPUT: y > 0 ? 1 : y + 1), colsum[C] = the
 * column sums of out (the bias gradient; two-stage).  C <= 256; scratch: csl_elu_bwd_colsum_scratch(n, C) floats. */
int csl_bias_elu_f32(float* y, int64_t ldy, const float* bias, int64_t n, int32_t C, int32_t elu, void* stream);
int64_t csl_elu_bwd_colsum_scratch(int64_t n, int32_t C);
int csl_elu_bwd_colsum_f32(const float* g, int64_t ldg, const float* y, int64_t ldy, int64_t n, int32_t C, int32_t elu,
                           float* out, int64_t ldo, float* colsum, float* scratch, void* stream);

/* GAT attention logits (DistGATConv.project): el[r, h] = <z[r, h, :], attn_l[h, :]>, er likewise; z [n, H*D],
 * attn_* [H, D], el/er [n, H]; D % 4 == 0, D <= 256, 16-byte aligned.  Backward: g_z [n, H*D] is WRITTEN
 * (g_el a_l + g_er a_r), g_attn_l / g_attn_r [H, D] are the sums over the rows (two-stage, no atomics);
 * scratch: csl_gat_logits_bwd_scratch(n, H, D) floats. */
int csl_gat_logits_fwd_f32(const float* z, const float* attn_l, const float* attn_r, int64_t n, int32_t H, int32_t D,
                           float* el, float* er, void* stream);
int64_t csl_gat_logits_bwd_scratch(int64_t n, int32_t H, int32_t D);
int csl_gat_logits_bwd_f32(const float* z, const float* attn_l, const float* attn_r, const float* g_el,
                           const float* g_er, int64_t n, int32_t H, int32_t D, float* g_z, float* g_attn_l,
                           float* g_attn_r, float* scratch, void* stream);

/* the same with g_z ACCUMULATED (accumulate != 0: g_z += ...), on top of csl_gat_bwd_f32's share of the gradient of z:
 * one buffer and no separate addition for z's two consumers */
int csl_gat_logits_bwd_acc_f32(const float* z, const float* attn_l, const float* attn_r, const float* g_el,
                               const float* g_er, int64_t n, int32_t H, int32_t D, float* g_z, int32_t accumulate,
                               float* g_attn_l, float* g_attn_r, float* scratch, void* stream);

/* GAT layer epilogue (DistGATConv: out[v] = sum_u alpha(u -> v) z[u] + bias, ELU between layers) from the aggregation's
 * (s [n,H], n [n,H*D]):  out[r, h, :] = act(n[r, h, :] / max(s[r, h], 1e-30) + bias[h, :]),  act = ELU if elu else identity.
 * Backward: p = g .* act'(out);  g_n = p / s;  g_s[r, h] = -sum_d g_n[r, h, d] n[r, h, d] / s[r, h];  g_bias = column
 * sums of p (two-stage; scratch: csl_gat_finish_bwd_scratch(n, H, D) floats).  D % 4 == 0, D <= 256, dense rows (g may
 * have a leading dimension ldg), 16-byte aligned. */
int csl_gat_finish_fwd_f32(const float* n_in, const float* s_in, const float* bias, int64_t n, int32_t H, int32_t D,
                           int32_t elu, float* out, void* stream);
int64_t csl_gat_finish_bwd_scratch(int64_t n, int32_t H, int32_t D);
int csl_gat_finish_bwd_f32(const float* g, int64_t ldg, const float* out, const float* n_in, const float* s_in, int64_t n,
                           int32_t H, int32_t D, int32_t elu, float* g_n, float* g_s, float* g_bias, float* scratch,
                           void* stream);

/* ---- fused GraphSAGE layer pieces (DistSageConv.forward, python/layers/dist_sageconv.py:42-84) ----
 *
 * csl_sage_cat_f32: the operand of Linear(2*in, out) in one pass (self_gather + gather/mean + concat,
 * dist_sageconv.py:66-80; python/data/bipartite.py:61-91):
 *   cat[r, 0:H)  = act(x[map(self_ids[r])])                                   (zero row for self_ids[r] = -1)
 *   cat[r, H:2H) = sum_{e in CSR row r} act(x[map(indices[e])]) / max(deg_r, 1)     when indptr != NULL
 *                = agg[owned[r]] / max(deg[r], 1)                                   when indptr == NULL
 * deg_r = deg ? deg[r] : (indptr[r+1] - indptr[r]); map(i) = rowmap ? rowmap[i] : i (the deepest layer reads the
 * resident feature table through the slice's in_nodes); act = ReLU if relu_in (x is then the previous layer's
 * pre-activation output).  Rows [n, n_pad) of cat are zeroed.  H % 4 == 0; x, agg, cat 16-byte aligned with
 * leading dimensions that are multiples of 4. */
int csl_sage_cat_f32(const int32_t* indptr, const int32_t* indices, const int32_t* self_ids, const int32_t* owned,
                     const int32_t* deg, const int32_t* rowmap, const float* x, int64_t ldx, const float* agg,
                     int64_t lda, int64_t n, int64_t n_pad, float* cat, int64_t ldc, int32_t H, int32_t relu_in,
                     void* stream);

/* gradient of the CSR form above w.r.t. x: gx [n_src, ldx] is ZEROED here, then
 *   gx[self_ids[r]] += gcat[r, 0:H),  gx[indices[e]] += gcat[r, H:2H) / max(indptr[r+1]-indptr[r], 1)  (fp32 atomics).
 * (With relu_in the caller applies the ReLU mask of the layer below to gx: csl_relu_bwd_colsum_f32.) */
int csl_sage_cat_bwd_f32(const int32_t* indptr, const int32_t* indices, const int32_t* self_ids, int64_t n,
                         const float* gcat, int64_t ldg, float* gx, int64_t ldx, int64_t n_src, int32_t H,
                         void* stream);

/* gradient of the merged-sums form (indptr == NULL) of csl_sage_cat_f32: gx [n_x, H] and gagg [n_agg, H] (dense,
 * either may be NULL) are ZEROED here, then gx[self_ids[r]] = gcat[r, 0:H) and
 * gagg[owned[r]] = gcat[r, H:2H) / max(deg[r], 1); self_ids (where >= 0) and owned are unique.  H % 4 == 0. */
int csl_sage_cat_rows_bwd_f32(const int32_t* self_ids, const int32_t* owned, const int32_t* deg, int64_t n,
                              const float* gcat, int64_t ldg, float* gx, int64_t n_x, float* gagg, int64_t n_agg,
                              int32_t H, void* stream);

/* Backward of csl_sage_cat_f32 (CSR form) as a gather over the slice by source (cslicer_hip.h CSL_T_INDPTR /
 * CSL_T_INDICES, engine flag CSL_FLAG_TRANSPOSE), with the ReLU mask of the layer below, the row padding of its GEMM
 * operand and its bias column sums in the same pass:
 *   out[u, :] = mask_u .* sum over t in t_indices[t_indptr[u] .. t_indptr[u+1]) of
 *               ( t < 0 ? gcat[~t, 0:H) : gcat[t, H:2H) / max(indptr[t+1] - indptr[t], 1) ),   mask_u = y ? y[u, :] > 0 : 1
 * for u < n_src, zero rows up to n_pad; colsum[c] = sum_u out[u, c].  No atomics, nothing to pre-zero, deterministic.
 * indptr: the slice's CSR row pointers (the forward's mean divisors); NULL: no division (gcat's mean half is already
 * divided: the rank step).  H % 4 == 0.
 * scratch: csl_sage_cat_bwd_t_scratch(n_pad, H) floats = [blocks][H] per-block column sums.  colsum == NULL: they are
 * left there for the caller's own second stage (csl_reduce_multi_f32 with nblk = scratch floats / H). */
int64_t csl_sage_cat_bwd_t_scratch(int64_t n_pad, int32_t H);
int csl_sage_cat_bwd_t_f32(const int32_t* t_indptr, const int32_t* t_indices, const int32_t* indptr, const float* gcat,
                           int64_t ldg, const float* y, int64_t ldy, int64_t n_src, int64_t n_pad, float* out,
                           int64_t ldo, float* colsum, float* scratch, int32_t H, void* stream);

/* Pieces of the split-parallel rank step's backward by source (csl_sage_rank_fwd_bwd_f32):
 * csl_sage_rank_g2_f32: g2[owned[j], 0:H) = gcat[j, 0:H), g2[owned[j], H:2H) = gcat[j, H:2H) / max(deg[j], 1): the operand
 * gradient from owned-row order into out-row order, the mean half divided by the TRUE degree (rows of other owners are
 * filled from the reverse exchange with csl_scatter_rows_f32); csl_sage_cat_bwd_t_f32 with indptr = NULL then gathers
 * over it without dividing again.  csl_scatter_rows_f32: dst[idx[k], 0:H) = src[k, 0:H), idx unique.  H % 4 == 0. */
int csl_sage_rank_g2_f32(const int32_t* owned, const int32_t* deg, int64_t n_owned, const float* gcat, int64_t ldg, float* g2,
                         int64_t ld2, int32_t H, void* stream);
int csl_scatter_rows_f32(float* dst, int64_t ldd, const int32_t* idx, int64_t n, const float* src, int64_t lds, int32_t H,
                         void* stream);

/* The same where the slice by source has HUB lists (csl_layer_meta.t_max_len > CSL_T_SORTED_MAX: a node that thousands
 * of the minibatch's rows sampled; the reference's backward walks every edge with its own thread for the same reason,
 * src/gnn/sage.cu:20-28): rows with longer lists are summed by many workgroups, a segment of the slice's entries each,
 * and added to the row with fp32 atomics (the order of those few adds per element is the only non-determinism); every
 * other row exactly as in csl_sage_cat_bwd_t_f32, same masks, padding and column sums.  t_entries = t_indptr[n_src]
 * (csl_layer_meta.off[CSL_T_INDICES]: the host knows it).  scratch: csl_sage_cat_bwd_t_hub_scratch(n_pad, H) floats =
 * [2 * blocks][H] per-block column sums (colsum == NULL leaves them there: nblk = scratch floats / H). */
int64_t csl_sage_cat_bwd_t_hub_scratch(int64_t n_pad, int32_t H);
int csl_sage_cat_bwd_t_hub_f32(const int32_t* t_indptr, const int32_t* t_indices, int64_t t_entries, const int32_t* indptr,
                               const float* gcat, int64_t ldg, const float* y, int64_t ldy, int64_t n_src, int64_t n_pad,
                               float* out, int64_t ldo, float* colsum, float* scratch, int32_t H, void* stream);

/* out[r, :] = (y == NULL || y[r, :] > 0) ? g[r, :] : 0 for r < n, zero rows for n <= r < n_pad;
 * colsum[c] = sum_r out[r, c]: ReLU backward + row padding of the GEMM operand + bias gradient in one pass
 * (two-stage reduction, no atomics, nothing to pre-zero).  scratch: csl_relu_bwd_colsum_scratch(n_pad, H) floats =
 * [blocks][H] per-block sums; colsum == NULL leaves them there for the caller's own second stage (csl_reduce_multi_f32). */
int64_t csl_relu_bwd_colsum_scratch(int64_t n_pad, int32_t H);
int csl_relu_bwd_colsum_f32(const float* g, int64_t ldg, const float* y, int64_t ldy, int64_t n, int64_t n_pad,
                            float* out, int64_t ldo, float* colsum, float* scratch, int32_t H, void* stream);

/* cross-entropy (python/train.py:86), forward and backward in one pass:
 * *loss = -scale * sum_r log softmax(logits[r])[label_r]; grad[r, :] = scale * (softmax(logits[r]) - onehot(label_r));
 * label_r = labels[rowmap ? rowmap[ids[r]] : ids[r]] (int64 labels, int32 node ids).
 * scratch: csl_softmax_ce_scratch(n) floats. */
int64_t csl_softmax_ce_scratch(int64_t n);
int csl_softmax_ce_f32(const float* logits, int64_t ldl, int64_t n, int32_t C, const int32_t* ids, const int32_t* rowmap,
                       const int64_t* labels, float scale, float* loss, float* grad, int64_t ldgr, float* scratch,
                       void* stream);

/* The same with the two-stage reductions left open, for a caller that finishes every reduction of a step in one
 * launch (csl_reduce_multi_f32): rows [n, n_pad) of grad are zeroed (GEMM operand padding); loss_partial[blocks] and,
 * if given (C <= 256), col_partial[blocks][C] (column sums of grad: the bias gradient) with blocks = ceil(n_pad / 4). */
int csl_softmax_ce_partial_f32(const float* logits, int64_t ldl, int64_t n, int64_t n_pad, int32_t C, const int32_t* ids,
                               const int32_t* rowmap, const int64_t* labels, float scale, float* grad, int64_t ldgr,
                               float* loss_partial, float* col_partial, void* stream);

/* dst[j][c] = sum over b < nblk[j] of src[j][b * H[j] + c], c < H[j], for count <= CSL_REDUCE_MULTI_MAX (12) jobs in ONE launch: the second
 * stage of a step's two-stage reductions (column sums, the loss with H = 1, the row slabs of a weight gradient).
 * src / nblk / H / dst are HOST arrays. */
#define CSL_REDUCE_MULTI_MAX 12
int csl_reduce_multi_f32(int32_t count, const float* const* src, const int64_t* nblk, const int32_t* H, float* const* dst,
                         void* stream);

/* torch.optim.Adam's update (python/train.py:83; no weight decay, no amsgrad) for up to 24 parameter tensors in one
 * launch.  params / grads / exp_avg / exp_avg_sq: HOST arrays of `count` device pointers, numel[t] elements each;
 * step = 1 for the first update (bias corrections 1 - beta^step are computed on the host). */
int csl_adam_f32(int32_t count, float* const* params, const float* const* grads, float* const* exp_avg,
                 float* const* exp_avg_sq, const int64_t* numel, float lr, float beta1, float beta2, float eps,
                 int64_t step, void* stream);

/* ---- the step's plain GEMMs (Linear forward / input gradient / weight gradient, dist_sageconv.py:34,80) ----
 * Row-major fp32:  C[b] (m x n, ldc) = op(A[b]) (m x k) . op(B[b]) (k x n)  (+ bias[n] added to every row) (ReLU),
 * b < batch, consecutive matrices stride_* elements apart (ignored for batch == 1); op(X) = X or X^T (trans* != 0:
 * X is stored [k, m] resp. [n, k]).  A direct hipBLASLt call with a cached plan per shape; the algorithm of a shape
 * is chosen by timing the library's candidates on its first use (C is overwritten several times then; environment
 * CSLICER_GEMM_TUNE=0 takes the library's first suggestion, CSLICER_GEMM_LOG=1 reports each choice).  The library
 * is bound at run time (the libhipblaslt.so.1 already in the process, else ROCm's).  csl_gemm_last_error: details
 * of the last failure.
 * ONE device and ONE stream per process: the library handle and the 128 MB workspace every plan shares are created by
 * the first call, on its device, for its stream; a call from another device or stream returns CSL_E_STATE (two GEMMs
 * on different streams would use the same workspace).  The trainer's contract: one GPU per process, one training
 * stream. */
int csl_gemm_f32(int32_t transa, int32_t transb, int64_t m, int64_t n, int64_t k, const float* A, int64_t lda,
                 int64_t stride_a, const float* B, int64_t ldb, int64_t stride_b, float* C, int64_t ldc, int64_t stride_c,
                 int32_t batch, const float* bias, int32_t relu, void* stream);
const char* csl_gemm_last_error(void);
/* Recorded plans.  CSLICER_GEMM_TUNE=all times EVERY solution the library has for a shape class (what a framework's
 * offline tuner does: ~0.3 s per class) instead of the heuristic's candidates; csl_gemm_save_plans writes the chosen
 * solution index of every class seen so far to a text file, csl_gemm_load_plans (returns the number of entries) makes
 * later plans of those classes take the recorded solution without timing anything.  The indices belong to one library
 * version (named in the file's header); with another version the file is ignored. */
int csl_gemm_save_plans(const char* path);
int csl_gemm_load_plans(const char* path);

/* out[c] = sum over b < n_slabs of slabs[b * n + c], c < n (n % 4 == 0, 16-byte aligned): the reduction of a weight
 * gradient computed as n_slabs independent row slabs (a batched csl_gemm_f32 with transa) */
int csl_sum_slabs_f32(const float* slabs, int64_t n, int32_t n_slabs, float* out, void* stream);

/* ---- a GraphSAGE layer's forward as ONE kernel on the fp32 matrix cores (csrc/sage_mfma.hip) ----
 * DistSageConv.forward (python/layers/dist_sageconv.py:66-80: self_gather, gather / mean over the slice CSR of
 * python/data/bipartite.py:61-67, concat, Linear(2*in, out)) without the gathered operand ever being read back from HBM:
 *   y[r, 0:out) = act_out( [ act_in(x[map(self_ids[r])]) | mean_{e in CSR row r} act_in(x[map(indices[e])]) ] . W^T + bias )
 * for r < n (a zero self block for self_ids[r] = -1, a zero mean for an empty row); rows [n, n_pad) get act_out(bias), what
 * the two-kernel form (csl_sage_cat_f32's zero rows + csl_gemm_f32) leaves there.  map(i) = rowmap ? rowmap[i] : i;
 * act_* = ReLU when relu_* != 0.  W [out, ldw >= 2 H] row-major (torch's Linear.weight), bias [out] or NULL.
 * cat != NULL: the operand [n_pad, ldc >= 2 H] is also written (the backward's weight-gradient GEMM reads it).
 * v_mfma_f32_32x32x2_f32: exact fp32, a k-ordered fmaf chain per output.  H % 4 == 0, out <= 256, ldx / ldw / ldc
 * multiples of 4, x / W / cat / wpack 16-byte aligned.  wpack: csl_sage_fwd_mfma_scratch(H, out) floats (W re-packed
 * in MFMA operand order, rewritten by every call). */
int64_t csl_sage_fwd_mfma_scratch(int32_t H, int32_t out);
int csl_sage_fwd_mfma_f32(const int32_t* indptr, const int32_t* indices, const int32_t* self_ids, const int32_t* rowmap,
                          const float* x, int64_t ldx, const float* W, int64_t ldw, const float* bias, int64_t n,
                          int64_t n_pad, int32_t H, int32_t out, int32_t relu_in, int32_t relu_out, float* cat,
                          int64_t ldc, float* y, int64_t ldy, float* wpack, void* stream);

/* ---- one training step of the GraphSAGE model on ONE part (python/train.py:56-88 on a GPU that holds every node) ----
 * Forward, cross-entropy and backward for one minibatch as one call: the fused kernels above + csl_gemm_f32, issued
 * from native code (a step is ~25 launches and 8 GEMMs of 5-90 us: issued one by one from an interpreter they cost more
 * host time than the GPU needs for them).
 *
 * Model: n_layers x Linear(2 * dims[k] -> dims[k+1]) over [self | mean of sampled neighbours], ReLU between (DistSageConv,
 * models/factory.py:7-56); layer k = 0 is the DEEPEST hop.  slices[k]: the engine's graph-mode slice of layer
 * n_layers-1-k, part 0 of 1, device pointers (csl_list_device_ptr / csl_arena_info); t_indptr / t_indices
 * (CSL_FLAG_TRANSPOSE) are needed for k >= 1.  n_in of layer k must equal n_out of layer k-1.
 * weights[k] [dims[k+1], 2*dims[k]] row-major, biases[k] [dims[k+1]]; feat: the resident feature table [N, ldf], read
 * through feat_rows = slices[0]'s in_nodes; seed_ids = the top slice's out_nodes (label of a seed: labels[id]).
 * Results: *loss (device) = -scale * sum over the seeds of log softmax(logits)[label]; grads (device): the gradients of
 * W_0, b_0, W_1, b_1, ... back to back.  Rows are padded to multiples of row_pad (0: no padding) so that GEMM shapes
 * repeat; weight gradients of padded layers are reduced in n_slabs row slabs.  dims[0 .. n_layers-1] % 4 == 0.
 * workspace: csl_sage_fwd_bwd_workspace(...) floats for THESE slice sizes, 16-byte aligned.
 * csl_sage_last_error: what failed (thread-local). */
typedef struct {
  const int32_t* indptr;
  const int32_t* indices;
  const int32_t* self_ids_in;
  const int32_t* t_indptr;
  const int32_t* t_indices;
  int64_t n_out;
  int64_t n_in;
  /* csl_layer_meta.t_max_len of the slice: above CSL_T_SORTED_MAX (a hub node's list: one wave would walk thousands of
   * entries) the layer's input gradient is gathered by csl_sage_cat_bwd_t_hub_f32 (hub rows by many workgroups) */
  int64_t t_max_len;
  /* entries of the slice by source (t_indptr[n_in]): csl_layer_meta.off[CSL_T_INDICES] of the part */
  int64_t t_entries;
} csl_sage_slice;
int64_t csl_sage_fwd_bwd_workspace(int32_t n_layers, const int32_t* dims, const csl_sage_slice* slices, int64_t row_pad,
                                   int32_t n_slabs);
int csl_sage_fwd_bwd_f32(int32_t n_layers, const int32_t* dims, const csl_sage_slice* slices, const float* const* weights,
                         const float* const* biases, const float* feat, int64_t ldf, const int32_t* feat_rows,
                         const int32_t* seed_ids, const int64_t* labels, float scale, int64_t row_pad, int32_t n_slabs,
                         float* grads, float* loss, float* workspace, int64_t workspace_floats, void* stream);
const char* csl_sage_last_error(void);

/* Diagnostics: device time of csl_sage_fwd_bwd_f32's launches by group, measured with HIP events around every launch on
 * the step's own stream (what bench.py's e2e.roofline is computed from: work of a group / time of that group's kernels,
 * not / the step's wall time).  csl_sage_step_timing(1) switches the recording on (it costs two event records per launch:
 * a measurement pass, not the timed region), csl_sage_step_timing_read waits for the recorded launches, returns the
 * milliseconds and launch counts accumulated since the last read in ms[CSL_STEP_GROUPS] / launches[CSL_STEP_GROUPS] and
 * clears them. */
#define CSL_STEP_FUSED_FWD 0    /* csl_sage_fwd_mfma_f32 (its W re-pack included) */
#define CSL_STEP_GEMM 1         /* csl_gemm_f32: the library GEMMs */
#define CSL_STEP_AGGREGATION 2  /* csl_sage_cat_f32, csl_sage_cat_bwd_t_f32 / csl_sage_cat_bwd_f32, csl_relu_bwd_colsum_f32 */
#define CSL_STEP_OTHER 3        /* loss, second stages of the reductions */
#define CSL_STEP_GROUPS 4
int csl_sage_step_timing(int32_t enable);
int csl_sage_step_timing_read(double* ms, int64_t* launches);

/* ---- one rank of the SPLIT-PARALLEL training step (python/train.py + dist_sageconv.py:42-84 on several GPUs) ----
 * Part g of P, one process per part: forward, loss over the seeds this part owns and backward of the GraphSAGE model
 * for one minibatch as one call.  Per layer the partial sums of the out rows that PEERS own are sent to their owners
 * and the partials of the rows this part owns are received and merged (pull_for_remotes / push_from_remotes,
 * dist_sageconv.py:52-65); backward runs the reverse exchange.  The exchange is the caller's: `exchange(user, layer,
 * backward, src, dst, width, stream)` must move row blocks of `width` floats between the parts --
 *   forward  (backward = 0): src = [n_from, width] rows grouped by owner (send_counts), dst = [n_to, width] rows grouped
 *                            by sender (recv_counts);
 *   backward (backward = 1): the reverse: src = [n_to, width] (recv_counts), dst = [n_from, width] (send_counts)
 * -- ordered on `stream` (an all_to_all_single over RCCL in production), and return 0 or a negative code.
 * slices[k] (model order, deepest first): the graph-mode slice of part g (csl_config.part_mask), device pointers;
 * from_all / to_all: the per-peer boundary lists back to back; n_in of layer k = n_owned of layer k-1.
 * feat: this part's resident feature rows, read through feat_rows[n_in of layer 0] (local row of every in node);
 * seed_ids [n_owned of the top layer]: global ids of the owned seeds, label of a seed = labels[label_rows ?
 * label_rows[id] : id]; scale = 1 / (seeds of the WHOLE minibatch).  grads: W_0, b_0, ... of THIS part's share (the
 * caller all-reduces them); *loss: this part's share.  Widths multiples of 4, at most 256 classes.
 * workspace: csl_sage_rank_workspace(...) floats. */
typedef struct {
  const int32_t* indptr;
  const int32_t* indices;
  const int32_t* self_ids_in;
  const int32_t* owned_out_nodes;
  const int32_t* owned_degree;
  const int32_t* from_all;
  const int32_t* to_all;
  int64_t n_out, n_in, n_owned, n_from, n_to;
  /* optional (layers k >= 1): the part's slice by source (engine flag CSL_FLAG_TRANSPOSE with csl_config.part_mask): the
   * backward then GATHERS the input gradient over it (csl_sage_rank_g2_f32 + csl_sage_cat_bwd_t_f32: no atomics, no zero
   * fill, ReLU mask / padding / bias sums in the same pass) instead of scattering it; NULL: the atomic scatter form */
  const int32_t* t_indptr;
  const int32_t* t_indices;
  int64_t t_max_len, t_entries;
} csl_sage_rank_slice;
typedef int (*csl_exchange_fn)(void* user, int32_t layer, int32_t backward, const float* src, float* dst, int32_t width,
                               void* stream);
/* optional second callback: with it, `exchange` may merely START the exchange on a stream of the caller's (after making
 * that stream wait for what `stream` holds so far) and `wait` makes `stream` wait for its end; the step calls it right
 * before the received rows are first used, so the rows that stay on the GPU are aggregated while the others travel */
typedef int (*csl_exchange_wait_fn)(void* user, int32_t layer, int32_t backward, void* stream);
int64_t csl_sage_rank_workspace(int32_t n_layers, const int32_t* dims, const csl_sage_rank_slice* slices, int64_t row_pad,
                                int32_t n_slabs);
int csl_sage_rank_fwd_bwd_f32(int32_t n_layers, const int32_t* dims, const csl_sage_rank_slice* slices,
                              const float* const* weights, const float* const* biases, const float* feat, int64_t ldf,
                              const int32_t* feat_rows, const int32_t* seed_ids, const int32_t* label_rows,
                              const int64_t* labels, float scale, int64_t row_pad, int32_t n_slabs,
                              csl_exchange_fn exchange, csl_exchange_wait_fn wait, void* user, float* grads, float* loss,
                              float* workspace, int64_t workspace_floats, void* stream);

#ifdef __cplusplus
}
#endif
#endif

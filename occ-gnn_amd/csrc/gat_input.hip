// gat_input.hip -- the INPUT layer of the multi-head attention model (BASELINE config 5), aggregate-then-project (gfx950).
//
// DistGATConv (cslicer/splitgnn.py; the reference's python/layers/dist_gatconv.py is a stub, python/data/bipartite.py:75-80
// `attention_gather` its only piece) projects every SOURCE row first: z = W x, el = <z, a_l>, er = <z, a_r>, then
// out[v] = sum_u alpha(u -> v) z[u].  At the deepest layer that is a projection of ~10 source rows per destination
// (config 5: 0.5 M rows x 100 -> 256, 25.6 GFLOP forward and the same again for the weight gradient) of which nine tenths
// only exist to be averaged.  Both the logits and the sum are LINEAR in x, so for a layer whose input takes no gradient
// (the feature table) the same numbers come from the raw rows:
//     v_l[h] = W_h^T a_l[h]  (H x F, tiny)          el[u, h] = <x[u], v_l[h]>      er[v, h] = <x[v], v_r[h]>
//     agg[v, h, :] = sum_u alpha_h(u -> v) x[u]     out[v, h, :] = W_h agg[v, h, :] + bias
// i.e. an attention-weighted sum of RAW feature rows per head (HBM-bound gather, this file), then a block-diagonal
// projection of the destinations only (H small GEMMs over n_out rows, csl_gemm_f32 batched): 10x fewer flops, no projected
// source matrix, no gathered input matrix (x is read through the slice's in_nodes), no by-source slice for this layer.
// Backward: dW_h = g_h^T agg_h and dagg_h = g_h W_h (batched GEMMs), then ONE pass over the edges (k_gatin_bwd) turns dagg
// into the gradients of v_l and v_r: a source's logit is recomputed per EDGE in the forward (sources of a sampled layer are
// nearly all distinct), so its gradient needs no per-source accumulation either.
//
// Layout: a destination row belongs to 32 lanes (two rows per wave), lane q holds columns 4q..4q+3 of a feature row
// (F <= 128, F % 4 == 0); the H x (32 / H) dot products of a group of 32 / H edges are reduced across the 32 lanes by a
// halving exchange (31 shuffles for 32 values) that leaves value i = edge_in_group * H + head in lane i.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "cslicer_aggr.h"
#include "cslicer_hip.h"

namespace {

constexpr int BLK = 256;
constexpr int RPB = BLK / 32;   // destination rows per block and pass
constexpr int GATIN_MAX_DEG = 32;

__device__ __forceinline__ float dot4(const float4 a, const float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ void fma4(float4& acc, const float s, const float4 v) {
  acc.x += s * v.x, acc.y += s * v.y, acc.z += s * v.z, acc.w += s * v.w;
}

// Sum of V values (V = 2^k <= 32) over the 32 lanes of a row group.  While more than one value is left a step halves
// them: a lane keeps the half its bit selects and receives the partner's share of it.  Returns in lane q the total of
// value q >> (5 - k).
template <int V>
__device__ __forceinline__ float treduce32(float (&v)[V], const int q) {
  int n = V;
#pragma unroll
  for (int d = 16; d >= 1; d >>= 1) {
    // the selection as bit arithmetic (one v_bfi_b32): written `up ? v[i + hn] : v[i]` the compiler turns it into ONE load
    // of v[i + (up ? hn : 0)], a per-lane index into a register array, and lowers that to a chain of 31 compares and
    // selects per value (5,800 of each in a kernel of 350 fmas)
    const unsigned m = (q & d) ? 0xffffffffu : 0u;
    if (n > 1) {
      const int hn = n >> 1;
#pragma unroll
      for (int i = 0; i < V / 2; i++) {
        if (i < hn) {
          const unsigned lo = __float_as_uint(v[i]), hi = __float_as_uint(v[i + hn]);
          const float keep = __uint_as_float((hi & m) | (lo & ~m));
          const float send = __uint_as_float((lo & m) | (hi & ~m));
          v[i] = keep + __shfl_xor(send, d);
        }
      }
      n = hn;
    } else {
      v[0] += __shfl_xor(v[0], d);
    }
  }
  return v[0];
}

template <int H>
struct Hlog;
template <> struct Hlog<1> { static constexpr int k = 0; };
template <> struct Hlog<2> { static constexpr int k = 1; };
template <> struct Hlog<4> { static constexpr int k = 2; };
template <> struct Hlog<8> { static constexpr int k = 3; };

__device__ __forceinline__ float leaky(const float x, const float slope) { return x > 0.f ? x : slope * x; }

// A kernel instance covers rows of at most ME edges (a multiple of the 32 / H edges of a reduction group) and keeps their
// feature rows in registers between its two passes over the edges: ME = 12 (48 registers) covers config 5's fanout of 10,
// ME = 32 anything the layer accepts.

// The index chain of a row, ONE hop per level for all its edges: lane q takes edge q (indices, then rowmap), the row ids are
// handed out with shuffles.  (An edge at a time it was three dependent memory round trips per group of edges and again per
// group in the second pass: 0.9 ms for the 110 k rows of config 5's deepest layer at two waves per SIMD.)
struct RowIdx {
  int e0, deg, sid;
  int xrow_q;   // lane q: feature row of edge q (of edge 0 for q >= deg)
  int srow;     // feature row of the destination itself (0 if it has none)
};
__device__ __forceinline__ RowIdx load_row_idx(const int* __restrict__ indptr, const int* __restrict__ indices,
                                               const int* __restrict__ self_ids, const int* __restrict__ rowmap,
                                               const long long r, const bool rowok, const int q, const int me) {
  RowIdx ri;
  ri.e0 = 0, ri.deg = 0, ri.sid = -1;
  if (rowok) {
    ri.e0 = indptr[r];
    ri.deg = indptr[r + 1] - ri.e0;
    if (ri.deg > me) ri.deg = me;   // (memory safety only: the host side refuses longer rows)
    ri.sid = self_ids[r];
  }
  const int src = indices[q < ri.deg ? ri.e0 + q : (ri.deg > 0 ? ri.e0 : 0)];   // unconditional, always a valid entry
  ri.srow = ri.sid >= 0 ? (rowmap ? rowmap[ri.sid] : ri.sid) : 0;
  ri.xrow_q = rowmap ? rowmap[src] : src;
  return ri;
}

// forward: agg[r, h, :] = sum_e alpha[e, h] x[src_e];  alpha[e, h] (its sign bit: the logit was <= 0) is kept for the
// backward.  Rows with more than GATIN_MAX_DEG edges are refused by the host side (the slicer's fanout bounds them).
template <int H, int ME>
__global__ __launch_bounds__(BLK) __attribute__((amdgpu_waves_per_eu(ME <= 16 ? 3 : 1, 8))) void k_gatin_fwd(const int* __restrict__ indptr, const int* __restrict__ indices,
                                                   const int* __restrict__ self_ids, const int* __restrict__ rowmap,
                                                   const float* __restrict__ x, long long ldx, int F,
                                                   const float* __restrict__ vl, const float* __restrict__ vr, float slope,
                                                   long long n_out, float* __restrict__ agg, float* __restrict__ alpha) {
  constexpr int EPG = 32 / H;              // edges per reduction group
  constexpr int NG = ME / EPG;             // groups of a row
  static_assert(ME % EPG == 0 && ME <= GATIN_MAX_DEG, "whole groups");
  __shared__ __attribute__((aligned(16))) float s_al[RPB][ME * H];
  const int q = threadIdx.x & 31, g = threadIdx.x >> 5, half = threadIdx.x & 32;
  const bool on = 4 * q < F;
  const int col = on ? 4 * q : 0;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const long long stride = (long long)gridDim.x * RPB;
  // v_r is used once per row: it lives in LDS (kept in registers across the passes it cost 32 of them and a wave per SIMD)
  __shared__ float4 s_vr[H][32];
  if (g == 0) {
#pragma unroll
    for (int h = 0; h < H; h++) {
      float4 w = *reinterpret_cast<const float4*>(vr + h * F + col);
      if (!on) w = zero4;
      s_vr[h][q] = w;
    }
  }
  float4 wl[H];
#pragma unroll
  for (int h = 0; h < H; h++) {
    wl[h] = *reinterpret_cast<const float4*>(vl + h * F + col);
    if (!on) wl[h] = zero4;
  }
  __syncthreads();
  // a workgroup walks rows blockIdx.x * RPB + g, + stride, ...: the index chain of the NEXT pass's row (three dependent
  // round trips) is requested while this pass's row is worked on
  RowIdx nx = load_row_idx(indptr, indices, self_ids, rowmap, (long long)blockIdx.x * RPB + g,
                           (long long)blockIdx.x * RPB + g < n_out, q, ME);
  for (long long rb = (long long)blockIdx.x * RPB; rb < n_out; rb += stride) {   // block-uniform
  const long long r = rb + g;
  const bool rowok = r < n_out;
  const RowIdx ri = nx;
  const int e0 = ri.e0, deg = ri.deg;
  const int degmax = max(deg, __shfl_xor(deg, 32));          // wave-uniform
  // every feature row the first pass needs is requested before any is used
  float4 xe[ME];
#pragma unroll
  for (int j = 0; j < ME; j++) {
    const long long row = __shfl(ri.xrow_q, half | j);
    xe[j] = *reinterpret_cast<const float4*>(x + row * ldx + col);
  }
  float4 xs = *reinterpret_cast<const float4*>(x + ri.srow * ldx + col);
  nx = load_row_idx(indptr, indices, self_ids, rowmap, r + stride, r + stride < n_out, q, ME);
  if (!on || ri.sid < 0) xs = zero4;
#pragma unroll
  for (int j = 0; j < ME; j++)
    if (!on || j >= deg) xe[j] = zero4;
  float er_mine;
  {
    float pv[H];
#pragma unroll
    for (int h = 0; h < H; h++) pv[h] = dot4(xs, s_vr[h][q]);
    const float t = treduce32<H>(pv, q);                     // lane q: head q >> (5 - k)
    er_mine = __shfl(t, half | ((q & (H - 1)) << (5 - Hlog<H>::k)));  // -> head q % H
  }
  float lg[NG];
  unsigned negbits = 0;
#pragma unroll
  for (int c = 0; c < NG; c++) {
    lg[c] = -1e30f;
    if (c * EPG < degmax) {                                  // wave-uniform
      float v[32];
#pragma unroll
      for (int j = 0; j < EPG; j++)
#pragma unroll
        for (int h = 0; h < H; h++) v[j * H + h] = dot4(xe[c * EPG + j], wl[h]);
      const float t = treduce32<32>(v, q);                   // lane q: edge q / H of the group, head q % H
      __builtin_amdgcn_sched_barrier(0);                     // (groups interleaved by the scheduler: 32 more registers each)
      const float raw = t + er_mine;
      if (c * EPG + q / H < deg) {
        lg[c] = leaky(raw, slope);
        if (!(raw > 0.f)) negbits |= 1u << c;
      }
    }
  }
  // softmax over the row's edges, per head: lanes q, q ^ H, q ^ 2H, ... hold the same head
  float m = lg[0];
#pragma unroll
  for (int c = 1; c < NG; c++) m = fmaxf(m, lg[c]);
#pragma unroll
  for (int d = H; d < 32; d <<= 1) m = fmaxf(m, __shfl_xor(m, d));
  float p[NG], s = 0.f;
#pragma unroll
  for (int c = 0; c < NG; c++) {
    p[c] = c * EPG + q / H < deg ? __expf(lg[c] - m) : 0.f;
    s += p[c];
  }
#pragma unroll
  for (int d = H; d < 32; d <<= 1) s += __shfl_xor(s, d);
  const float inv = s > 0.f ? 1.f / s : 0.f;
#pragma unroll
  for (int c = 0; c < NG; c++) {
    const float a = p[c] * inv;
    s_al[g][c * 32 + q] = a;
    if (c * EPG + q / H < deg)
      alpha[(long long)(e0 + c * EPG) * H + q] = ((negbits >> c) & 1u) ? -a : a;   // 32 consecutive floats per group
  }
  __syncthreads();
  float4 acc[H];
#pragma unroll
  for (int h = 0; h < H; h++) acc[h] = zero4;
#pragma unroll
  for (int j = 0; j < ME; j++) {
    if (j < degmax) {                                        // wave-uniform
      asm volatile("" ::: "memory");   // (all 12 x 8 weights read ahead would cost 96 registers)
#pragma unroll
      for (int h = 0; h < H; h++) fma4(acc[h], s_al[g][j * H + h], xe[j]);   // (zero weights and rows for j >= deg)
    }
  }
  if (rowok && on) {
#pragma unroll
    for (int h = 0; h < H; h++) *reinterpret_cast<float4*>(agg + (r * H + h) * F + col) = acc[h];
  }
  __syncthreads();   // s_al is rewritten by the next pass
  }
}

// backward: from dagg[r, h, :] (row stride ld_r, head stride ld_h) and the kept alpha, the gradients of v_l and v_r as
// per-block partial sums part_l / part_r [blocks][H * F] (second stage: csl_reduce_multi_f32).
//   dalpha[e, h] = <dagg[r, h], x[src_e]>      dlogit = alpha (dalpha - sum_e alpha dalpha)      draw = dlogit * leaky'
//   dv_l[h] += draw[e, h] x[src_e]             dv_r[h] += (sum_e draw[e, h]) x[self(r)]
template <int H, int ME>
__global__ __launch_bounds__(BLK) void k_gatin_bwd(const int* __restrict__ indptr, const int* __restrict__ indices,
                                                   const int* __restrict__ self_ids, const int* __restrict__ rowmap,
                                                   const float* __restrict__ x, long long ldx, int F,
                                                   const float* __restrict__ alpha, const float* __restrict__ dagg,
                                                   long long ld_r, long long ld_h, float slope, long long n_out,
                                                   long long rows_per_block, float* __restrict__ part_l,
                                                   float* __restrict__ part_r) {
  constexpr int EPG = 32 / H;
  constexpr int NG = ME / EPG;
  static_assert(ME % EPG == 0 && ME <= GATIN_MAX_DEG, "whole groups");
  __shared__ __attribute__((aligned(16))) float s_dr[RPB][ME * H];
  __shared__ __attribute__((aligned(16))) float s_der[RPB][8];
  __shared__ float4 s_red[RPB][32];
  __shared__ float4 s_accr[RPB][H][32];   // the v_r side's sums live in LDS: 32 registers less per lane
  const int q = threadIdx.x & 31, g = threadIdx.x >> 5, half = threadIdx.x & 32;
  const bool on = 4 * q < F;
  const int col = on ? 4 * q : 0;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r_end = r0 + rows_per_block < n_out ? r0 + rows_per_block : n_out;
  float4 accl[H];
#pragma unroll
  for (int h = 0; h < H; h++) {
    accl[h] = zero4;
    s_accr[g][h][q] = zero4;
  }
  // the index chain of the NEXT pass's row is requested while this pass's row is worked on
  RowIdx nx = load_row_idx(indptr, indices, self_ids, rowmap, r0 + g, r0 + g < r_end, q, ME);
  for (long long rb = r0; rb < r_end; rb += RPB) {           // block-uniform
    const long long r = rb + g;
    const bool rowok = r < r_end;
    const RowIdx ri = nx;
    const int e0 = ri.e0, deg = ri.deg;
    const int degmax = max(deg, __shfl_xor(deg, 32));
    float4 xe[ME];
#pragma unroll
    for (int j = 0; j < ME; j++) {
      const long long row = __shfl(ri.xrow_q, half | j);
      xe[j] = *reinterpret_cast<const float4*>(x + row * ldx + col);
    }
    float4 xs = *reinterpret_cast<const float4*>(x + ri.srow * ldx + col);
    const long long rr = rowok ? r : 0;
    float4 da[H];
#pragma unroll
    for (int h = 0; h < H; h++) da[h] = *reinterpret_cast<const float4*>(dagg + rr * ld_r + h * ld_h + col);
    float av[NG];
#pragma unroll
    for (int c = 0; c < NG; c++) {
      const bool mine = c * EPG + q / H < deg;
      av[c] = alpha[mine ? (long long)(e0 + c * EPG) * H + q : 0];
      if (!mine) av[c] = 0.f;
    }
    nx = load_row_idx(indptr, indices, self_ids, rowmap, r + RPB, r + RPB < r_end, q, ME);
    if (!on || ri.sid < 0) xs = zero4;
#pragma unroll
    for (int j = 0; j < ME; j++)
      if (!on || j >= deg) xe[j] = zero4;
#pragma unroll
    for (int h = 0; h < H; h++)
      if (!on || !rowok) da[h] = zero4;
    float dal[NG];
    float tsum = 0.f;
#pragma unroll
    for (int c = 0; c < NG; c++) {
      dal[c] = 0.f;
      if (c * EPG < degmax) {                                // wave-uniform
        float v[32];
#pragma unroll
        for (int j = 0; j < EPG; j++)
#pragma unroll
          for (int h = 0; h < H; h++) v[j * H + h] = dot4(xe[c * EPG + j], da[h]);
        dal[c] = treduce32<32>(v, q);
        __builtin_amdgcn_sched_barrier(0);                   // (groups interleaved by the scheduler: 32 more registers each)
        tsum += fabsf(av[c]) * dal[c];
      }
    }
#pragma unroll
    for (int d = H; d < 32; d <<= 1) tsum += __shfl_xor(tsum, d);
    float der = 0.f;
#pragma unroll
    for (int c = 0; c < NG; c++) {
      // (the sign bit of alpha: the logit was <= 0; -0.0f keeps it for a weight that underflowed)
      const float dr = fabsf(av[c]) * (dal[c] - tsum) * ((__float_as_uint(av[c]) >> 31) ? slope : 1.f);
      s_dr[g][c * 32 + q] = dr;
      der += dr;
    }
#pragma unroll
    for (int d = H; d < 32; d <<= 1) der += __shfl_xor(der, d);
    if (q < H) s_der[g][q] = der;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ME; j++) {
      if (j < degmax) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int h = 0; h < H; h++) fma4(accl[h], s_dr[g][j * H + h], xe[j]);
      }
    }
#pragma unroll
    for (int h = 0; h < H; h++) {
      float4 t = s_accr[g][h][q];
      fma4(t, s_der[g][h], xs);
      s_accr[g][h][q] = t;
    }
    __syncthreads();   // s_dr / s_der are rewritten by the next pass
  }
  // the block's sums: over its RPB row groups, head by head
#pragma unroll
  for (int side = 0; side < 2; side++) {
#pragma unroll
    for (int h = 0; h < H; h++) {
      s_red[g][q] = side ? s_accr[g][h][q] : accl[h];
      __syncthreads();
      if (g == 0) {
        float4 t = s_red[0][q];
#pragma unroll
        for (int k = 1; k < RPB; k++) {
          const float4 u = s_red[k][q];
          t.x += u.x, t.y += u.y, t.z += u.z, t.w += u.w;
        }
        if (on) *reinterpret_cast<float4*>((side ? part_r : part_l) + ((long long)blockIdx.x * H + h) * F + col) = t;
      }
      __syncthreads();
    }
  }
}

// y[r, c] = act(y[r, c] + bias[c]) in place (ELU with alpha = 1 when elu), C % 4 == 0
__global__ __launch_bounds__(BLK) void k_bias_elu(float* __restrict__ y, long long ldy, const float* __restrict__ bias,
                                                  long long n, int C, int elu) {
  const int qpr = C / 4;
  const long long i = (long long)blockIdx.x * BLK + threadIdx.x;
  if (i >= n * qpr) return;
  const long long r = i / qpr;
  const int c = (int)(i - r * qpr) * 4;
  float4 v = *reinterpret_cast<float4*>(y + r * ldy + c);
  const float4 b = *reinterpret_cast<const float4*>(bias + c);
  v.x += b.x, v.y += b.y, v.z += b.z, v.w += b.w;
  if (elu) {
    v.x = v.x > 0.f ? v.x : expm1f(v.x);
    v.y = v.y > 0.f ? v.y : expm1f(v.y);
    v.z = v.z > 0.f ? v.z : expm1f(v.z);
    v.w = v.w > 0.f ? v.w : expm1f(v.w);
  }
  *reinterpret_cast<float4*>(y + r * ldy + c) = v;
}

// out[r, :] = g[r, :] * act'(y[r, :]) (ELU: y > 0 ? 1 : y + 1) and the per-block column sums of out (the bias gradient);
// a block takes rows_per_block rows, thread t the float4 column t % (C/4) of every (BLK / (C/4))-th row
__global__ __launch_bounds__(BLK) void k_elu_bwd_colsum(const float* __restrict__ g, long long ldg,
                                                        const float* __restrict__ y, long long ldy, long long n, int C,
                                                        int elu, float* __restrict__ out, long long ldo,
                                                        float* __restrict__ part, long long rows_per_block) {
  __shared__ float4 s_p[BLK];
  const int qpr = C / 4;             // <= 64
  const int rows_at_once = BLK / qpr;
  const int cq = threadIdx.x % qpr, rq = threadIdx.x / qpr;
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r_end = r0 + rows_per_block < n ? r0 + rows_per_block : n;
  float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
  if (rq < rows_at_once) {
    for (long long r = r0 + rq; r < r_end; r += rows_at_once) {
      float4 v = *reinterpret_cast<const float4*>(g + r * ldg + 4 * cq);
      if (elu) {
        const float4 o = *reinterpret_cast<const float4*>(y + r * ldy + 4 * cq);
        v.x *= o.x > 0.f ? 1.f : o.x + 1.f;
        v.y *= o.y > 0.f ? 1.f : o.y + 1.f;
        v.z *= o.z > 0.f ? 1.f : o.z + 1.f;
        v.w *= o.w > 0.f ? 1.f : o.w + 1.f;
      }
      *reinterpret_cast<float4*>(out + r * ldo + 4 * cq) = v;
      sum.x += v.x, sum.y += v.y, sum.z += v.z, sum.w += v.w;
    }
  }
  s_p[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x < qpr) {
    float4 t = s_p[threadIdx.x];
    for (int k = 1; k < rows_at_once; k++) {
      const float4 u = s_p[k * qpr + threadIdx.x];
      t.x += u.x, t.y += u.y, t.z += u.z, t.w += u.w;
    }
    *reinterpret_cast<float4*>(part + (long long)blockIdx.x * C + 4 * threadIdx.x) = t;
  }
}

// ---- the block-diagonal projection of the destinations and its two gradients on the fp32 matrix cores ----
//   proj  out[r, h, d]  = act(sum_f agg[r, h, f] W[h, d, f] + bias[h, d])
//   dagg  dagg[r, h, f] = sum_d gg[r, h, d] W[h, d, f]
//   dW    gW[h, d, f]   = sum_r gg[r, h, d] agg[r, h, f]
// H independent GEMMs of [n, F] x [F, D] with F = 100, D = 32 at config 5: 2.8 GFLOP over 0.2 GB, memory-bound; as one
// strided-batched library call each they took 105 / 118 / 176 us (the library's tiles are made for one large matrix).
// Here a wave owns one head and walks 16-row tiles with v_mfma_f32_16x16x4_f32, W_h stationary in its registers.  Operands
// come straight from global memory in operand order: the sum over k may run in any order, so lane (m, kq) takes
// k = 16 j + 4 kq + i (j = chunk, i = 0..3): ONE float4 load per chunk, 64 contiguous bytes per row and instruction.
// No LDS, no barrier.  Rows past n repeat row n - 1 (same values stored twice: benign), k past F / f past F are zeros.
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int BD_WAVES = 4;   // per block

template <int KT, int NT>   // KT = ceil(F / 16) k-chunks, NT = D / 16 column tiles
__global__ __launch_bounds__(64 * BD_WAVES) void k_bd_proj(const float* __restrict__ agg, const float* __restrict__ W,
                                                          const float* __restrict__ bias, float* __restrict__ out,
                                                          long long ldo, long long n, int H, int F, int elu) {
  constexpr int D = 16 * NT;
  const int l = threadIdx.x & 63, m = l & 15, kq = l >> 4;
  const long long gw = (long long)blockIdx.x * BD_WAVES + (threadIdx.x >> 6), GW = (long long)gridDim.x * BD_WAVES;
  const int h = (int)(gw % H);
  const long long tiles = (n + 15) / 16;
  float b[KT][4][NT];
#pragma unroll
  for (int j = 0; j < KT; j++) {
    const int k0 = 16 * j + 4 * kq;
#pragma unroll
    for (int t = 0; t < NT; t++) {
      float4 w = *reinterpret_cast<const float4*>(W + ((long long)h * D + 16 * t + m) * F + (k0 < F ? k0 : 0));
      if (k0 >= F) w = make_float4(0.f, 0.f, 0.f, 0.f);
      b[j][0][t] = w.x, b[j][1][t] = w.y, b[j][2][t] = w.z, b[j][3][t] = w.w;
    }
  }
  float bv[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) bv[t] = bias[h * D + 16 * t + m];
  auto load_a = [&](const long long tile, float4 (&a)[KT]) {
    long long row = tile * 16 + m;
    if (row >= n) row = n - 1;
#pragma unroll
    for (int j = 0; j < KT; j++) {
      const int k0 = 16 * j + 4 * kq;
      a[j] = *reinterpret_cast<const float4*>(agg + (row * H + h) * F + (k0 < F ? k0 : 0));
      if (k0 >= F) a[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  float4 nxt[KT];
  long long tile = gw / H;
  const long long step = GW / H;
  if (tile < tiles) load_a(tile, nxt);
  for (; tile < tiles; tile += step) {
    float4 a[KT];
#pragma unroll
    for (int j = 0; j < KT; j++) a[j] = nxt[j];
    load_a(tile + step < tiles ? tile + step : tile, nxt);   // (the last tile re-reads itself: no branch around the loads)
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < KT; j++) {
      const float av[4] = {a[j].x, a[j].y, a[j].z, a[j].w};
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], b[j][i][t], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      long long r = tile * 16 + 4 * kq + i;
      if (r >= n) r = n - 1;
#pragma unroll
      for (int t = 0; t < NT; t++) {
        float v = acc[t][i] + bv[t];
        if (elu) v = v > 0.f ? v : expm1f(v);
        out[r * ldo + h * D + 16 * t + m] = v;
      }
    }
  }
}

// dagg[r, h, 0:16 KT) (head stride 16 KT: whole tiles, no masked stores; the columns >= F are zeros)
template <int KT, int NT>
__global__ __launch_bounds__(64 * BD_WAVES) void k_bd_dagg(const float* __restrict__ gg, long long ldg,
                                                          const float* __restrict__ W, float* __restrict__ dagg,
                                                          long long n, int H, int F) {
  constexpr int D = 16 * NT, FP = 16 * KT;
  const int l = threadIdx.x & 63, m = l & 15, kq = l >> 4;
  const long long gw = (long long)blockIdx.x * BD_WAVES + (threadIdx.x >> 6), GW = (long long)gridDim.x * BD_WAVES;
  const int h = (int)(gw % H);
  const long long tiles = (n + 15) / 16;
  float b[NT][4][KT];   // B[k = d][col = f] = W[h, d, f]
#pragma unroll
  for (int j = 0; j < NT; j++)
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int t = 0; t < KT; t++) {
        const int f = 16 * t + m;
        const float w = W[((long long)h * D + 16 * j + 4 * kq + i) * F + (f < F ? f : 0)];
        b[j][i][t] = f < F ? w : 0.f;
      }
  auto load_a = [&](const long long tile, float4 (&a)[NT]) {
    long long row = tile * 16 + m;
    if (row >= n) row = n - 1;
#pragma unroll
    for (int j = 0; j < NT; j++) a[j] = *reinterpret_cast<const float4*>(gg + row * ldg + h * D + 16 * j + 4 * kq);
  };
  float4 nxt[NT];
  long long tile = gw / H;
  const long long step = GW / H;
  if (tile < tiles) load_a(tile, nxt);
  for (; tile < tiles; tile += step) {
    float4 a[NT];
#pragma unroll
    for (int j = 0; j < NT; j++) a[j] = nxt[j];
    load_a(tile + step < tiles ? tile + step : tile, nxt);
    f32x4 acc[KT];
#pragma unroll
    for (int t = 0; t < KT; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NT; j++) {
      const float av[4] = {a[j].x, a[j].y, a[j].z, a[j].w};
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int t = 0; t < KT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], b[j][i][t], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      long long r = tile * 16 + 4 * kq + i;
      if (r >= n) r = n - 1;
#pragma unroll
      for (int t = 0; t < KT; t++) dagg[(r * H + h) * FP + 16 * t + m] = acc[t][i];
    }
  }
}

// gW partial sums: a wave owns head h and the rows [rg * rows_per, (rg + 1) * rows_per); k = the rows, four per MFMA.
// The rows of a tile need not be consecutive d, nor its columns consecutive f: M-tile i takes d = NT m + i and column tile
// (j2, j) takes f = 64 j2 + 4 n + j, so that a lane's operands of ALL tiles are NT consecutive floats of gg and one float4
// (two for F > 64) of agg: 3 load instructions per four rows instead of 9 four-byte ones.  (A memory instruction costs this
// chip's L1 ~27 cycles plus ~8 per 64-byte sector it touches, whatever its width -- measured across these kernels; with
// 256-byte instructions the kernel was bound by that fixed part: 97 us.  Splitting the tiles over more waves to raise the
// occupancy doubled the instructions and took 214 us; a software pipeline of the loads 120.)
// part[rg][h][d][f]; second stage: csl_reduce_multi_f32 over the rg.
template <int FT, int NT>   // FT = ceil(F / 64) float4 loads per row, NT = D / 16
__global__ __launch_bounds__(64 * BD_WAVES) void k_bd_dw(const float* __restrict__ gg, long long ldg,
                                                        const float* __restrict__ agg, float* __restrict__ part,
                                                        long long n, int H, int F, long long rows_per) {
  constexpr int D = 16 * NT, CT = 4 * FT;
  typedef float avec __attribute__((ext_vector_type(NT)));
  const int l = threadIdx.x & 63, m = l & 15, kq = l >> 4;
  const long long gw = (long long)blockIdx.x * BD_WAVES + (threadIdx.x >> 6);
  const int h = (int)(gw % H);
  const long long rg = gw / H;
  const long long r_lo = rg * rows_per, r_hi = r_lo + rows_per < n ? r_lo + rows_per : n;
  f32x4 acc[NT][CT];
#pragma unroll
  for (int i = 0; i < NT; i++)
#pragma unroll
    for (int t = 0; t < CT; t++) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bool col_ok[FT];
#pragma unroll
  for (int j2 = 0; j2 < FT; j2++) col_ok[j2] = 64 * j2 + 4 * m < F;
  constexpr int U = 4;   // k-steps (of four rows) whose operands are requested together
  for (long long r0 = r_lo; r0 < r_hi; r0 += 4 * U) {
    avec a[U];
    float4 b[U][FT];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const long long r = r0 + 4 * u + kq;
      const bool ok = r < r_hi;
      const long long rr = ok ? r : r_lo;
      a[u] = *reinterpret_cast<const avec*>(gg + rr * ldg + h * D + NT * m);
      if (!ok) a[u] = (avec)(0.f);
#pragma unroll
      for (int j2 = 0; j2 < FT; j2++) {
        b[u][j2] = *reinterpret_cast<const float4*>(agg + (rr * H + h) * F + (col_ok[j2] ? 64 * j2 + 4 * m : 0));
        if (!col_ok[j2]) b[u][j2] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int j2 = 0; j2 < FT; j2++) {
        const float bv[4] = {b[u][j2].x, b[u][j2].y, b[u][j2].z, b[u][j2].w};
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
          for (int i = 0; i < NT; i++)
            acc[i][4 * j2 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i], bv[j], acc[i][4 * j2 + j], 0, 0, 0);
      }
  }
  float* dst = part + (rg * H + h) * (long long)D * F;
#pragma unroll
  for (int i = 0; i < NT; i++)
#pragma unroll
    for (int j2 = 0; j2 < FT; j2++)
#pragma unroll
      for (int c = 0; c < 4; c++) {
        // C row 4 kq + c of M-tile i is d = NT (4 kq + c) + i; the lane's four column tiles are consecutive f
        const int d = NT * (4 * kq + c) + i;
        if (col_ok[j2])
          *reinterpret_cast<float4*>(dst + (long long)d * F + 64 * j2 + 4 * m) =
              make_float4(acc[i][4 * j2][c], acc[i][4 * j2 + 1][c], acc[i][4 * j2 + 2][c], acc[i][4 * j2 + 3][c]);
      }
}

constexpr int BD_BLOCKS = 512;   // 2048 waves: two workgroups per CU, a multiple of every supported head count
int bd_blocks() {   // workgroups of k_bd_proj / k_bd_dagg (CSL_BD_BLOCKS: measurement knob, an even number)
  static const int v = getenv("CSL_BD_BLOCKS") ? atoi(getenv("CSL_BD_BLOCKS")) : BD_BLOCKS;
  return v < 2 ? 2 : (v > 4096 ? 4096 : v / 2 * 2);
}

int bd_kt(int F) { return F <= 64 ? 4 : (F <= 112 ? 7 : 8); }
bool bd_ok(int H, int F, int D) { return (H == 1 || H == 2 || H == 4 || H == 8) && F >= 4 && F % 4 == 0 && F <= 128 && (D == 16 || D == 32 || D == 64); }

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
int done() { return hipGetLastError() == hipSuccess ? CSL_OK : CSL_E_HIP; }

long long bwd_rows(long long n) {   // destination rows per workgroup of k_gatin_bwd / k_elu_bwd_colsum: <= ~1024 workgroups
  long long rpb = (n + 1023) / 1024;
  rpb = (rpb + RPB - 1) / RPB * RPB;
  return rpb < RPB ? RPB : rpb;
}

bool heads_ok(int H) { return H == 1 || H == 2 || H == 4 || H == 8; }

// v_l[h, f] = sum_d W[h*D + d, f] attn_l[h, d] (and v_r): the attention vectors pulled back through the projection
__global__ __launch_bounds__(BLK) void k_gatin_vlr(const float* __restrict__ W, const float* __restrict__ al,
                                                   const float* __restrict__ ar, int H, int D, int F,
                                                   float* __restrict__ vlr) {
  const int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= H * F) return;
  const int h = i / F, f = i - h * F;
  float sl = 0.f, sr = 0.f;
  for (int d = 0; d < D; d++) {
    const float w = W[((long long)h * D + d) * F + f];
    sl += w * al[h * D + d];
    sr += w * ar[h * D + d];
  }
  vlr[i] = sl;
  vlr[H * F + i] = sr;
}

// the chain rule through v = W_h^T a, a wave per row (h, d) of W:
//   gW[h*D + d, f] += a_l[h, d] g_vl[h, f] + a_r[h, d] g_vr[h, f];   g_al[h, d] = sum_f W[h*D + d, f] g_vl[h, f]  (g_ar likewise)
__global__ __launch_bounds__(BLK) void k_gatin_chain(const float* __restrict__ W, const float* __restrict__ al,
                                                     const float* __restrict__ ar, const float* __restrict__ gv, int H, int D,
                                                     int F, float* __restrict__ gW, float* __restrict__ g_al,
                                                     float* __restrict__ g_ar) {
  const int row = blockIdx.x * (BLK / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= H * D) return;
  const int h = row / D;
  const float a_l = al[row], a_r = ar[row];
  float sl = 0.f, sr = 0.f;
  for (int f = lane; f < F; f += 64) {
    const float w = W[(long long)row * F + f], vl = gv[h * F + f], vr = gv[(H + h) * F + f];
    gW[(long long)row * F + f] += a_l * vl + a_r * vr;
    sl += w * vl;
    sr += w * vr;
  }
  for (int o = 32; o > 0; o >>= 1) {
    sl += __shfl_down(sl, o);
    sr += __shfl_down(sr, o);
  }
  if (lane == 0) {
    g_al[row] = sl;
    g_ar[row] = sr;
  }
}

// A layer call collects the second stages of its kernels' two-stage sums and runs them as ONE csl_reduce_multi_f32 launch.
struct DeferredSums {
  int count;
  const float* src[8];
  int64_t nblk[8];
  int32_t H[8];
  float* dst[8];
};
thread_local DeferredSums* g_defer = nullptr;
int reduce_or_defer(int count, const float* const* src, const int64_t* nblk, const int32_t* Hs, float* const* dst, void* stream) {
  if (!g_defer) return csl_reduce_multi_f32(count, src, nblk, Hs, dst, stream);
  for (int j = 0; j < count; j++) {
    if (g_defer->count >= 8) return CSL_E_STATE;
    const int k = g_defer->count++;
    g_defer->src[k] = src[j], g_defer->nblk[k] = nblk[j], g_defer->H[k] = Hs[j], g_defer->dst[k] = dst[j];
  }
  return CSL_OK;
}
long long up4(long long v) { return (v + 3) / 4 * 4; }

}  // namespace

extern "C" {

int32_t csl_gat_in_max_degree(void) { return GATIN_MAX_DEG; }

int csl_gat_in_fwd_f32(const int32_t* indptr, const int32_t* indices, const int32_t* self_ids, const int32_t* rowmap,
                       const float* x, int64_t ldx, int32_t F, const float* vl, const float* vr, int32_t H, float slope,
                       int64_t n_out, int64_t n_edges, int32_t max_deg, float* agg, float* alpha, void* stream) {
  if (n_out < 0 || n_edges < 0 || max_deg < 0 || max_deg > GATIN_MAX_DEG || !heads_ok(H) || F < 4 || F % 4 != 0 || F > 128 || ldx % 4 != 0 || ldx < F) return CSL_E_INVALID;
  if (n_out == 0) return CSL_OK;
  if (!indptr || !self_ids || !x || !vl || !vr || !agg || !aligned16(x) || !aligned16(vl) || !aligned16(vr) || !aligned16(agg))
    return CSL_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  if (n_edges == 0) {   // (no index to read: every row is empty)
    return hipMemsetAsync(agg, 0, sizeof(float) * (size_t)n_out * H * F, st) == hipSuccess ? CSL_OK : CSL_E_HIP;
  }
  if (!indices || !alpha) return CSL_E_INVALID;
  long long need = (n_out + RPB - 1) / RPB;
  const unsigned blocks = (unsigned)(need < 768 ? need : 768);   // three workgroups per CU (all resident) walk the rows
#define LAUNCH_GIF(HH, MM)                                                                                             \
  hipLaunchKernelGGL((k_gatin_fwd<HH, MM>), dim3(blocks), dim3(BLK), 0, st, indptr, indices, self_ids, rowmap, x,          \
                     (long long)ldx, (int)F, vl, vr, slope, (long long)n_out, agg, alpha)
  switch (H) {
    case 1: LAUNCH_GIF(1, 32); break;
    case 2: if (max_deg <= 16) LAUNCH_GIF(2, 16); else LAUNCH_GIF(2, 32); break;
    case 4: if (max_deg <= 16) LAUNCH_GIF(4, 16); else LAUNCH_GIF(4, 32); break;
    default: if (max_deg <= 12) LAUNCH_GIF(8, 12); else LAUNCH_GIF(8, 32); break;
  }
#undef LAUNCH_GIF
  return done();
}

int64_t csl_gat_in_bwd_scratch(int64_t n_out, int32_t H, int32_t F) {
  if (n_out <= 0) return 0;
  const long long rpb = bwd_rows(n_out);
  return 2 * ((n_out + rpb - 1) / rpb) * (long long)H * F;
}

int csl_gat_in_bwd_f32(const int32_t* indptr, const int32_t* indices, const int32_t* self_ids, const int32_t* rowmap,
                       const float* x, int64_t ldx, int32_t F, const float* alpha, const float* dagg, int64_t ld_r,
                       int64_t ld_h, int32_t H, float slope, int64_t n_out, int64_t n_edges, int32_t max_deg, float* g_vl,
                       float* g_vr, float* scratch, void* stream) {
  if (n_out < 0 || n_edges < 0 || max_deg < 0 || max_deg > GATIN_MAX_DEG || !heads_ok(H) || F < 4 || F % 4 != 0 || F > 128 || ldx % 4 != 0 || ldx < F || ld_r % 4 != 0 ||
      ld_h % 4 != 0)
    return CSL_E_INVALID;
  if (!g_vl || !g_vr || !aligned16(g_vl) || !aligned16(g_vr)) return CSL_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  if (n_out == 0 || n_edges == 0) {   // no edge: no attention, no gradient through the logits
    if (hipMemsetAsync(g_vl, 0, sizeof(float) * (size_t)H * F, st) != hipSuccess) return CSL_E_HIP;
    return hipMemsetAsync(g_vr, 0, sizeof(float) * (size_t)H * F, st) == hipSuccess ? CSL_OK : CSL_E_HIP;
  }
  if (!indptr || !indices || !self_ids || !x || !alpha || !dagg || !scratch || !aligned16(x) || !aligned16(dagg) ||
      !aligned16(scratch))
    return CSL_E_INVALID;
  const long long rpb = bwd_rows(n_out);
  const long long blocks = (n_out + rpb - 1) / rpb;
  float* part_l = scratch;
  float* part_r = scratch + blocks * H * F;
#define LAUNCH_GIB(HH, MM)                                                                                             \
  hipLaunchKernelGGL((k_gatin_bwd<HH, MM>), dim3((unsigned)blocks), dim3(BLK), 0, st, indptr, indices, self_ids, rowmap, x, \
                     (long long)ldx, (int)F, alpha, dagg, (long long)ld_r, (long long)ld_h, slope, (long long)n_out, rpb, \
                     part_l, part_r)
  switch (H) {
    case 1: LAUNCH_GIB(1, 32); break;
    case 2: if (max_deg <= 16) LAUNCH_GIB(2, 16); else LAUNCH_GIB(2, 32); break;
    case 4: if (max_deg <= 16) LAUNCH_GIB(4, 16); else LAUNCH_GIB(4, 32); break;
    default: if (max_deg <= 12) LAUNCH_GIB(8, 12); else LAUNCH_GIB(8, 32); break;
  }
#undef LAUNCH_GIB
  if (hipGetLastError() != hipSuccess) return CSL_E_HIP;
  const float* src[2] = {part_l, part_r};
  const int64_t nblk[2] = {blocks, blocks};
  const int32_t Hs[2] = {H * F, H * F};
  float* dst[2] = {g_vl, g_vr};
  return reduce_or_defer(2, src, nblk, Hs, dst, stream);
}

int csl_bias_elu_f32(float* y, int64_t ldy, const float* bias, int64_t n, int32_t C, int32_t elu, void* stream) {
  if (n < 0 || C < 4 || C % 4 != 0 || ldy < C || ldy % 4 != 0) return CSL_E_INVALID;
  if (n == 0) return CSL_OK;
  if (!y || !bias || !aligned16(y) || !aligned16(bias)) return CSL_E_INVALID;
  const long long total = n * (C / 4);
  hipLaunchKernelGGL(k_bias_elu, dim3((unsigned)((total + BLK - 1) / BLK)), dim3(BLK), 0, (hipStream_t)stream, y,
                     (long long)ldy, bias, (long long)n, (int)C, (int)elu);
  return done();
}

int64_t csl_elu_bwd_colsum_scratch(int64_t n, int32_t C) {
  if (n <= 0) return 0;
  const long long rpb = bwd_rows(n);
  return ((n + rpb - 1) / rpb) * (long long)C;
}

int csl_elu_bwd_colsum_f32(const float* g, int64_t ldg, const float* y, int64_t ldy, int64_t n, int32_t C, int32_t elu,
                           float* out, int64_t ldo, float* colsum, float* scratch, void* stream) {
  if (n < 0 || C < 4 || C % 4 != 0 || C > 256 || ldg < C || ldg % 4 != 0 || ldo < C || ldo % 4 != 0 || !colsum) return CSL_E_INVALID;
  if (elu && (ldy < C || ldy % 4 != 0)) return CSL_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) return hipMemsetAsync(colsum, 0, sizeof(float) * (size_t)C, st) == hipSuccess ? CSL_OK : CSL_E_HIP;
  if (!g || !out || !scratch || (elu && !y) || !aligned16(g) || !aligned16(out) || !aligned16(scratch) || (elu && !aligned16(y)))
    return CSL_E_INVALID;
  const long long rpb = bwd_rows(n);
  const long long blocks = (n + rpb - 1) / rpb;
  hipLaunchKernelGGL(k_elu_bwd_colsum, dim3((unsigned)blocks), dim3(BLK), 0, st, g, (long long)ldg, y, (long long)ldy,
                     (long long)n, (int)C, (int)elu, out, (long long)ldo, scratch, rpb);
  if (hipGetLastError() != hipSuccess) return CSL_E_HIP;
  const float* src[1] = {scratch};
  const int64_t nblk[1] = {blocks};
  const int32_t Hs[1] = {C};
  float* dst[1] = {colsum};
  return reduce_or_defer(1, src, nblk, Hs, dst, stream);
}

int32_t csl_gat_in_proj_ok(int32_t H, int32_t F, int32_t D) { return bd_ok(H, F, D) ? 1 : 0; }
int32_t csl_gat_in_proj_fpad(int32_t F) { return 16 * bd_kt(F); }

#define BD_DISPATCH(KERNEL, ...)                                                                   \
  do {                                                                                             \
    const int kt = bd_kt(F), nt = D / 16;                                                          \
    if (kt == 4 && nt == 1) hipLaunchKernelGGL((KERNEL<4, 1>), grid, block, 0, st, __VA_ARGS__);   \
    else if (kt == 4 && nt == 2) hipLaunchKernelGGL((KERNEL<4, 2>), grid, block, 0, st, __VA_ARGS__); \
    else if (kt == 4) hipLaunchKernelGGL((KERNEL<4, 4>), grid, block, 0, st, __VA_ARGS__);         \
    else if (kt == 7 && nt == 1) hipLaunchKernelGGL((KERNEL<7, 1>), grid, block, 0, st, __VA_ARGS__); \
    else if (kt == 7 && nt == 2) hipLaunchKernelGGL((KERNEL<7, 2>), grid, block, 0, st, __VA_ARGS__); \
    else if (kt == 7) hipLaunchKernelGGL((KERNEL<7, 4>), grid, block, 0, st, __VA_ARGS__);         \
    else if (nt == 1) hipLaunchKernelGGL((KERNEL<8, 1>), grid, block, 0, st, __VA_ARGS__);         \
    else if (nt == 2) hipLaunchKernelGGL((KERNEL<8, 2>), grid, block, 0, st, __VA_ARGS__);         \
    else hipLaunchKernelGGL((KERNEL<8, 4>), grid, block, 0, st, __VA_ARGS__);                      \
  } while (0)

int csl_gat_in_proj_f32(const float* agg, const float* W, const float* bias, int64_t n, int32_t H, int32_t F, int32_t D,
                        int32_t elu, float* out, int64_t ldo, void* stream) {
  if (n < 0 || !bd_ok(H, F, D) || ldo < (int64_t)H * D) return CSL_E_INVALID;
  if (n == 0) return CSL_OK;
  if (!agg || !W || !bias || !out || !aligned16(agg) || !aligned16(W)) return CSL_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(bd_blocks()), block(64 * BD_WAVES);
  BD_DISPATCH(k_bd_proj, agg, W, bias, out, (long long)ldo, (long long)n, (int)H, (int)F, (int)elu);
  return done();
}

constexpr int BD_DW_BLOCKS_MAX = 1024;
int bd_dw_blocks() {   // workgroups of k_bd_dw
  // (profiles/gat_dw_sweep.sh at config 5's shape: 256 / 512 / 768 / 1024 workgroups 142 / 95 / 83 / 101 us: three waves per
  // SIMD, all resident)
  static const int v = getenv("CSL_BD_DW_BLOCKS") ? atoi(getenv("CSL_BD_DW_BLOCKS")) : 768;
  return v < 8 ? 8 : (v > BD_DW_BLOCKS_MAX ? BD_DW_BLOCKS_MAX : v / 8 * 8);
}
int64_t csl_gat_in_proj_bwd_scratch(int32_t H, int32_t F, int32_t D) {
  return (int64_t)(BD_DW_BLOCKS_MAX * BD_WAVES / (H > 0 ? H : 1)) * H * D * F;
}

int csl_gat_in_proj_bwd_f32(const float* gg, int64_t ldg, const float* agg, const float* W, int64_t n, int32_t H, int32_t F,
                            int32_t D, float* dagg, float* gW, float* scratch, void* stream) {
  if (n < 0 || !bd_ok(H, F, D) || ldg < (int64_t)H * D || ldg % 4 != 0) return CSL_E_INVALID;
  if (!gW) return CSL_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) return hipMemsetAsync(gW, 0, sizeof(float) * (size_t)H * D * F, st) == hipSuccess ? CSL_OK : CSL_E_HIP;
  if (!gg || !agg || !W || !dagg || !scratch || !aligned16(gg) || !aligned16(agg) || !aligned16(W) || !aligned16(scratch) ||
      !aligned16(gW))
    return CSL_E_INVALID;
  const dim3 grid(bd_blocks()), block(64 * BD_WAVES);
  BD_DISPATCH(k_bd_dagg, gg, (long long)ldg, W, dagg, (long long)n, (int)H, (int)F);
  const long long ranges = (long long)bd_dw_blocks() * BD_WAVES / H;
  long long rows_per = (n + ranges - 1) / ranges;
  rows_per = (rows_per + 15) / 16 * 16;
  const dim3 grid_dw((unsigned)bd_dw_blocks());
#define LAUNCH_DW(FF, NN)                                                                                          \
  hipLaunchKernelGGL((k_bd_dw<FF, NN>), grid_dw, block, 0, st, gg, (long long)ldg, agg, scratch, (long long)n, (int)H, (int)F, \
                     rows_per)
  if (F <= 64) {
    if (D == 16) LAUNCH_DW(1, 1); else if (D == 32) LAUNCH_DW(1, 2); else LAUNCH_DW(1, 4);
  } else {
    if (D == 16) LAUNCH_DW(2, 1); else if (D == 32) LAUNCH_DW(2, 2); else LAUNCH_DW(2, 4);
  }
#undef LAUNCH_DW
  if (hipGetLastError() != hipSuccess) return CSL_E_HIP;
  const float* src[1] = {scratch};
  const int64_t nblk[1] = {ranges};
  const int32_t Hs[1] = {H * D * F};
  float* dst[1] = {gW};
  return reduce_or_defer(1, src, nblk, Hs, dst, stream);
}

/* ---- the whole layer as one call per direction ---- */
int64_t csl_gat_in_layer_fwd_scratch(int32_t H, int32_t F) { return 2 * (int64_t)H * F; }

int csl_gat_in_layer_fwd_f32(const int32_t* indptr, const int32_t* indices, const int32_t* self_ids, const int32_t* rowmap,
                             const float* x, int64_t ldx, int32_t F, const float* W, const float* attn_l, const float* attn_r,
                             const float* bias, int32_t H, int32_t D, float slope, int32_t elu, int64_t n_out, int64_t n_edges,
                             int32_t max_deg, float* agg, float* alpha, float* out, int64_t ldo, float* scratch, void* stream) {
  if (!bd_ok(H, F, D) || !W || !attn_l || !attn_r || !scratch || !aligned16(scratch)) return CSL_E_INVALID;
  hipLaunchKernelGGL(k_gatin_vlr, dim3((unsigned)((H * F + BLK - 1) / BLK)), dim3(BLK), 0, (hipStream_t)stream, W, attn_l, attn_r,
                     (int)H, (int)D, (int)F, scratch);
  int rc = csl_gat_in_fwd_f32(indptr, indices, self_ids, rowmap, x, ldx, F, scratch, scratch + (size_t)H * F, H, slope, n_out,
                              n_edges, max_deg, agg, alpha, stream);
  if (rc != CSL_OK) return rc;
  return csl_gat_in_proj_f32(agg, W, bias, n_out, H, F, D, elu, out, ldo, stream);
}

int64_t csl_gat_in_layer_bwd_scratch(int64_t n_out, int32_t H, int32_t F, int32_t D) {
  return up4(csl_elu_bwd_colsum_scratch(n_out, H * D)) + up4(csl_gat_in_proj_bwd_scratch(H, F, D)) +
         up4(csl_gat_in_bwd_scratch(n_out, H, F)) + 2 * (int64_t)H * F;
}

int csl_gat_in_layer_bwd_f32(const int32_t* indptr, const int32_t* indices, const int32_t* self_ids, const int32_t* rowmap,
                             const float* x, int64_t ldx, int32_t F, const float* W, const float* attn_l, const float* attn_r,
                             int32_t H, int32_t D, float slope, int32_t elu, int64_t n_out, int64_t n_edges, int32_t max_deg,
                             const float* agg, const float* alpha, const float* out, int64_t ldo, const float* g, int64_t ldg,
                             float* gg, float* dagg, float* gW, float* g_al, float* g_ar, float* g_bias, float* scratch,
                             void* stream) {
  if (!bd_ok(H, F, D) || !W || !attn_l || !attn_r || !gW || !g_al || !g_ar || !g_bias || !scratch || !aligned16(scratch) || n_out < 0)
    return CSL_E_INVALID;
  const int C = H * D, FP = 16 * bd_kt(F);
  float* s_elu = scratch;
  float* s_dw = s_elu + up4(csl_elu_bwd_colsum_scratch(n_out, C));
  float* s_in = s_dw + up4(csl_gat_in_proj_bwd_scratch(H, F, D));
  float* g_v = s_in + up4(csl_gat_in_bwd_scratch(n_out, H, F));
  DeferredSums jobs;
  jobs.count = 0;
  g_defer = &jobs;
  int rc = csl_elu_bwd_colsum_f32(g, ldg, out, ldo, n_out, C, elu, gg, C, g_bias, s_elu, stream);
  if (rc == CSL_OK) rc = csl_gat_in_proj_bwd_f32(gg, C, agg, W, n_out, H, F, D, dagg, gW, s_dw, stream);
  if (rc == CSL_OK)
    rc = csl_gat_in_bwd_f32(indptr, indices, self_ids, rowmap, x, ldx, F, alpha, dagg, (int64_t)H * FP, FP, H, slope, n_out,
                            n_edges, max_deg, g_v, g_v + (size_t)H * F, s_in, stream);
  g_defer = nullptr;
  if (rc != CSL_OK) return rc;
  if (jobs.count) {
    rc = csl_reduce_multi_f32(jobs.count, jobs.src, jobs.nblk, jobs.H, jobs.dst, stream);
    if (rc != CSL_OK) return rc;
  }
  hipLaunchKernelGGL(k_gatin_chain, dim3((unsigned)((C + BLK / 64 - 1) / (BLK / 64))), dim3(BLK), 0, (hipStream_t)stream, W, attn_l,
                     attn_r, g_v, (int)H, (int)D, (int)F, gW, g_al, g_ar);
  return done();
}

}  // extern "C"

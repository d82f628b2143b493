// aggregate.hip -- split-parallel aggregation over the slices (gfx950).
//
// The downstream of the slicer on every GPU: sum-aggregate source features over
// the slice CSR, pull/push boundary rows, normalise by the true degree.  All of
// it is HBM-bound row gathering (1 flop per 4 bytes): no MFMA here; the dense
// Linear(2*in, out) that follows is a library GEMM.
//
// Layout: feature rows are contiguous fp32; a group of G = ceil(H/4) lanes
// (rounded up to a power of two) owns one output row and moves it as float4;
// 64/G rows share a wave so short feature rows still fill every lane.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>

#include "cslicer_aggr.h"
#include "cslicer_hip.h"

namespace {

constexpr int BLK = 256;

__device__ __forceinline__ float4 ld4(const float* p, int c, int H) {
  // c = first column of this lane's quad; tail quads are read element-wise
  if (c + 3 < H) return *reinterpret_cast<const float4*>(p + c);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < H) v.x = p[c];
  if (c + 1 < H) v.y = p[c + 1];
  if (c + 2 < H) v.z = p[c + 2];
  return v;
}
__device__ __forceinline__ void st4(float* p, int c, int H, float4 v) {
  if (c + 3 < H) {
    *reinterpret_cast<float4*>(p + c) = v;
    return;
  }
  if (c < H) p[c] = v.x;
  if (c + 1 < H) p[c + 1] = v.y;
  if (c + 2 < H) p[c + 2] = v.z;
}
__device__ __forceinline__ void add4(float4& a, const float4 b) {
  a.x += b.x;
  a.y += b.y;
  a.z += b.z;
  a.w += b.w;
}

// G lanes per row (power of two, <= 64); quads beyond G*4 columns are looped
template <int G>
__global__ __launch_bounds__(BLK) void k_spmm_sum(const int* __restrict__ indptr,
                                                  const int* __restrict__ indices,
                                                  const int* __restrict__ rows, long long n_rows,
                                                  const float* __restrict__ x, long long ldx, float* __restrict__ out,
                                                  long long ldo, int H, int vec_ok, int compact,
                                                  const int* __restrict__ rowmap) {
  // rowmap: x is a resident table read through it (x row of source s = rowmap[s]): the deepest layer of a rank reads its
  // feature rows in place instead of gathering them into an input matrix first
  constexpr int RPB = BLK / G;  // rows per block
  const int lane = threadIdx.x % G;
  const long long r = (long long)blockIdx.x * RPB + threadIdx.x / G;
  if (r >= n_rows) return;
  const long long row = rows ? rows[r] : r;
  const long long orow = compact ? r : row;  // compact: the k-th listed row goes to out[k] (a send buffer)
  const long long e0 = indptr[row], e1 = indptr[row + 1];
  for (int c = lane * 4; c < H; c += G * 4) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    long long e = e0;
    // four source rows in flight per lane
    for (; e + 4 <= e1; e += 4) {
      long long s0 = indices[e], s1 = indices[e + 1], s2 = indices[e + 2], s3 = indices[e + 3];
      if (rowmap) s0 = rowmap[s0], s1 = rowmap[s1], s2 = rowmap[s2], s3 = rowmap[s3];
      float4 v0, v1, v2, v3;
      if (vec_ok) {
        v0 = *reinterpret_cast<const float4*>(x + s0 * ldx + c);
        v1 = *reinterpret_cast<const float4*>(x + s1 * ldx + c);
        v2 = *reinterpret_cast<const float4*>(x + s2 * ldx + c);
        v3 = *reinterpret_cast<const float4*>(x + s3 * ldx + c);
      } else {
        v0 = ld4(x + s0 * ldx, c, H);
        v1 = ld4(x + s1 * ldx, c, H);
        v2 = ld4(x + s2 * ldx, c, H);
        v3 = ld4(x + s3 * ldx, c, H);
      }
      // fixed association: the sum is in edge order, reproducible run to run
      add4(acc, v0);
      add4(acc, v1);
      add4(acc, v2);
      add4(acc, v3);
    }
    for (; e < e1; e++) {
      long long s0 = indices[e];
      if (rowmap) s0 = rowmap[s0];
      add4(acc, vec_ok ? *reinterpret_cast<const float4*>(x + s0 * ldx + c) : ld4(x + s0 * ldx, c, H));
    }
    if (vec_ok) {
      *reinterpret_cast<float4*>(out + orow * ldo + c) = acc;
    } else {
      st4(out + orow * ldo, c, H, acc);
    }
  }
}

// backward of the sum-aggregate: one wave per output row walks the row's edges and adds
// the row's gradient into each source row, 64 consecutive floats per atomic instruction
// (the shape the chip's memory-side float atomics run fastest at)
__global__ __launch_bounds__(BLK) void k_spmm_sum_bwd(const int* __restrict__ indptr,
                                                      const int* __restrict__ indices,
                                                      const int* __restrict__ rows, long long n_rows,
                                                      const float* __restrict__ g, long long ldg, int compact,
                                                      float* __restrict__ gx, long long ldx, int H) {
  const int lane = threadIdx.x & 63;
  const long long r = (long long)blockIdx.x * (BLK / 64) + (threadIdx.x >> 6);
  if (r >= n_rows) return;
  const long long row = rows ? rows[r] : r;
  const long long q = compact ? r : row;
  const long long e0 = indptr[row], e1 = indptr[row + 1];
  for (int c = lane; c < H; c += 64) {
    const float v = g[q * ldg + c];
    for (long long e = e0; e < e1; e++) atomicAdd(gx + indices[e] * ldx + c, v);
  }
}

template <int G>
__global__ __launch_bounds__(BLK) void k_gather_rows(const float* __restrict__ src, long long lds,
                                                     const int* __restrict__ idx, long long n,
                                                     float* __restrict__ dst, long long ldd, int H, int vec_ok) {
  constexpr int RPB = BLK / G;
  const int lane = threadIdx.x % G;
  const long long k = (long long)blockIdx.x * RPB + threadIdx.x / G;
  if (k >= n) return;
  const long long s = idx[k];
  for (int c = lane * 4; c < H; c += G * 4) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (s >= 0) v = vec_ok ? *reinterpret_cast<const float4*>(src + s * lds + c) : ld4(src + s * lds, c, H);
    if (vec_ok) {
      *reinterpret_cast<float4*>(dst + k * ldd + c) = v;
    } else {
      st4(dst + k * ldd, c, H, v);
    }
  }
}

template <int G>
__global__ __launch_bounds__(BLK) void k_scatter_add_rows(float* __restrict__ dst, long long ldd,
                                                          const int* __restrict__ idx, long long n,
                                                          const float* __restrict__ src, long long lds, int H,
                                                          int vec_ok) {
  constexpr int RPB = BLK / G;
  const int lane = threadIdx.x % G;
  const long long k = (long long)blockIdx.x * RPB + threadIdx.x / G;
  if (k >= n) return;
  const long long d = idx[k];
  if (d < 0) return;
  for (int c = lane * 4; c < H; c += G * 4) {
    float4 a = vec_ok ? *reinterpret_cast<const float4*>(dst + d * ldd + c) : ld4(dst + d * ldd, c, H);
    add4(a, vec_ok ? *reinterpret_cast<const float4*>(src + k * lds + c) : ld4(src + k * lds, c, H));
    if (vec_ok) {
      *reinterpret_cast<float4*>(dst + d * ldd + c) = a;
    } else {
      st4(dst + d * ldd, c, H, a);
    }
  }
}

// dst[idx[k], :] += src[k, :] where idx MAY repeat (the partial sums several peers send for the same owned node,
// merged in one launch): fp32 atomics, one wave per row, 64 consecutive floats per instruction
__global__ __launch_bounds__(BLK) void k_scatter_add_rows_atomic(float* __restrict__ dst, long long ldd,
                                                                 const int* __restrict__ idx, long long n,
                                                                 const float* __restrict__ src, long long lds, int H) {
  const int lane = threadIdx.x & 63;
  const long long k = (long long)blockIdx.x * (BLK / 64) + (threadIdx.x >> 6);
  if (k >= n) return;
  const long long d = idx[k];
  if (d < 0) return;
  for (int c = lane; c < H; c += 64) atomicAdd(dst + d * ldd + c, src[k * lds + c]);
}

template <int G>
__global__ __launch_bounds__(BLK) void k_div_rows(float* __restrict__ x, long long ldx,
                                                  const int* __restrict__ deg, long long n, int H, int vec_ok) {
  constexpr int RPB = BLK / G;
  const int lane = threadIdx.x % G;
  const long long k = (long long)blockIdx.x * RPB + threadIdx.x / G;
  if (k >= n) return;
  const long long d = deg[k];
  const float inv = 1.0f / (float)(d > 1 ? d : 1);
  for (int c = lane * 4; c < H; c += G * 4) {
    float4 a = vec_ok ? *reinterpret_cast<const float4*>(x + k * ldx + c) : ld4(x + k * ldx, c, H);
    a.x *= inv;
    a.y *= inv;
    a.z *= inv;
    a.w *= inv;
    if (vec_ok) {
      *reinterpret_cast<float4*>(x + k * ldx + c) = a;
    } else {
      st4(x + k * ldx, c, H, a);
    }
  }
}


// ---- GAT: fused edge-softmax aggregation over one slice's CSR, split-parallel form.
// For destination row r and head h, over the row's LOCAL edges (sources owned by this part):
//   score_e = LeakyReLU(el[src_e, h] + er[r, h])        m = max_e score_e
//   s = sum_e exp(score_e - m)                          n[h, :] = sum_e exp(score_e - m) * z[src_e, h, :]
// (m, s, n) are the partial softmax state the owner of r merges across parts (log-sum-exp), then out = n / s.
// One wave per row, a lane owns 4 consecutive feature columns (so D % 4 == 0, D <= 256);
// rows have <= fanout edges, so both passes stay in cache.  HBM-bound gather of z rows: no MFMA.
__device__ __forceinline__ float leaky(float x, float slope) { return x > 0.f ? x : slope * x; }

__global__ __launch_bounds__(BLK) void k_gat_fwd(const int* __restrict__ indptr, const int* __restrict__ indices,
                                                 long long n_rows, const float* __restrict__ el,
                                                 const float* __restrict__ er, const float* __restrict__ z, int H, int D,
                                                 float slope, float* __restrict__ m_out, float* __restrict__ s_out,
                                                 float* __restrict__ n_out) {
  const int lane = threadIdx.x & 63;
  const long long r = (long long)blockIdx.x * (BLK / 64) + (threadIdx.x >> 6);
  if (r >= n_rows) return;
  const int C = H * D;
  const int lpc = (64 / (D / 4)) * (D / 4);  // lanes per chunk: whole heads only
  const int e0 = indptr[r], e1 = indptr[r + 1];
  constexpr int EK = 16;
  if (e1 - e0 <= EK && C <= lpc * 4) {
    // The common case (a sampled row: <= fanout edges; one chunk of columns): every load of the row is requested before
    // the first is used -- lane j takes edge j's source id, the ids are handed out with shuffles, then the EK logits and
    // the EK z rows are in flight together.  (Edge by edge it was two dependent round trips per edge and pass: 17-21 us
    // for the 1 k- and 8 k-row layers of config 5's model whatever their size.)
    const int deg = e1 - e0;
    const int c = lane * 4;
    const bool on = lane < lpc && c < C;
    const int h = on ? c / D : 0;
    if (deg == 0) {   // (no edge: nothing to read; the state of an empty softmax)
      if (on) {
        *reinterpret_cast<float4*>(n_out + r * C + c) = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c % D == 0) {
          m_out[r * H + h] = -1e30f;
          s_out[r * H + h] = 0.f;
        }
      }
      return;
    }
    const int src_l = indices[e0 + (lane < deg ? lane : 0)];
    const float erv = er[r * H + h];
    float elv[EK];
    float4 zv[EK];
#pragma unroll
    for (int e = 0; e < EK; e++) {
      const long long src = __shfl(src_l, e < deg ? e : 0);
      elv[e] = el[src * H + h];
      zv[e] = *reinterpret_cast<const float4*>(z + src * C + (on ? c : 0));
    }
    float m = -1e30f;
#pragma unroll
    for (int e = 0; e < EK; e++)
      if (e < deg) m = fmaxf(m, leaky(elv[e] + erv, slope));
    float s = 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int e = 0; e < EK; e++) {
      if (e < deg) {
        const float p = expf(leaky(elv[e] + erv, slope) - m);
        s += p;
        acc.x += p * zv[e].x, acc.y += p * zv[e].y, acc.z += p * zv[e].z, acc.w += p * zv[e].w;
      }
    }
    if (on) {
      *reinterpret_cast<float4*>(n_out + r * C + c) = acc;
      if (c % D == 0) {
        m_out[r * H + h] = m;
        s_out[r * H + h] = s;
      }
    }
    return;
  }
  for (int c0 = 0; c0 < C; c0 += lpc * 4) {
    const int c = c0 + lane * 4;
    const bool on = lane < lpc && c < C;
    const int h = on ? c / D : 0;
    const float erv = er[r * H + h];
    float m = -1e30f;
    for (int e = e0; e < e1; e++) m = fmaxf(m, leaky(el[(long long)indices[e] * H + h] + erv, slope));
    float s = 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int e = e0; e < e1; e++) {
      const long long src = indices[e];
      const float p = expf(leaky(el[src * H + h] + erv, slope) - m);
      s += p;
      if (on) {
        const float4 zv = *reinterpret_cast<const float4*>(z + src * C + c);
        acc.x += p * zv.x;
        acc.y += p * zv.y;
        acc.z += p * zv.z;
        acc.w += p * zv.w;
      }
    }
    if (on) {
      *reinterpret_cast<float4*>(n_out + r * C + c) = acc;
      if (c % D == 0) {
        m_out[r * H + h] = m;
        s_out[r * H + h] = s;
      }
    }
  }
}

// Backward of (s, n) w.r.t. el, er, z (m is a stabiliser: the merged result does not depend on it).
// g_el and g_z are accumulated with fp32 atomics (caller zeroes them), g_er is written.
__global__ __launch_bounds__(BLK) void k_gat_bwd(const int* __restrict__ indptr, const int* __restrict__ indices,
                                                 long long n_rows, const float* __restrict__ el,
                                                 const float* __restrict__ er, const float* __restrict__ z, int H, int D,
                                                 float slope, const float* __restrict__ m_in,
                                                 const float* __restrict__ g_s, const float* __restrict__ g_n,
                                                 float* g_el, float* __restrict__ g_er, float* g_z) {
  const int lane = threadIdx.x & 63;
  const long long r = (long long)blockIdx.x * (BLK / 64) + (threadIdx.x >> 6);
  if (r >= n_rows) return;
  const int C = H * D;
  const int gsz = D / 4;              // lanes of one head
  const int lpc = (64 / gsz) * gsz;   // lanes per chunk: whole heads only
  const int gl = lane % gsz;          // position inside the head's lane group
  const int e0 = indptr[r], e1 = indptr[r + 1];
  for (int c0 = 0; c0 < C; c0 += lpc * 4) {
    const int c = c0 + lane * 4;
    const bool on = lane < lpc && c < C;
    const int h = on ? c / D : 0;
    const float erv = er[r * H + h], mh = m_in[r * H + h], gs = g_s[r * H + h];
    const float4 gn = on ? *reinterpret_cast<const float4*>(g_n + r * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    // The z gradient leaves in a second register layout: lane l owns columns c0 + l + 64 k, so that every atomic
    // instruction adds 64 CONSECUTIVE floats (the shape the memory-side float atomics run fastest at; the float4
    // layout above made each instruction touch every fourth float: 1.9 ms of the 7 ms GAT step).  The row's
    // gradient is loaded once more in that layout; the edge's weight p of a column's head comes from a lane of
    // the float4 layout that owns that head (all lanes of a head hold the same p).
    const int chunk_end = c0 + lpc * 4 < C ? c0 + lpc * 4 : C;
    float gn2[4];
    int p_lane[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int col = c0 + lane + 64 * k;
      gn2[k] = col < chunk_end ? g_n[r * C + col] : 0.f;
      p_lane[k] = col < chunk_end ? ((col / D) * D - c0) / 4 : 0;
    }
    float ger = 0.f;
    for (int e = e0; e < e1; e++) {
      const long long src = indices[e];
      const float raw = el[src * H + h] + erv;
      const float p = expf(leaky(raw, slope) - mh);
      float dot = 0.f;
      if (on) {
        const float4 zv = *reinterpret_cast<const float4*>(z + src * C + c);
        dot = gn.x * zv.x + gn.y * zv.y + gn.z * zv.z + gn.w * zv.w;
      }
      // sum over the head's lane group (any size): segmented tree, then the leader's value for all
      for (int o = 1; o < gsz; o <<= 1) {
        const float t = __shfl_down(dot, o);
        if (gl + o < gsz) dot += t;
      }
      dot = __shfl(dot, lane - gl);
      const float gsc = (gs + dot) * p * (raw > 0.f ? 1.f : slope);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const float pk = __shfl(p, p_lane[k]);
        const int col = c0 + lane + 64 * k;
        if (col < chunk_end) atomicAdd(g_z + src * C + col, pk * gn2[k]);
      }
      if (on && c % D == 0) {
        atomicAdd(g_el + src * H + h, gsc);
        ger += gsc;
      }
    }
    if (on && c % D == 0) g_er[r * H + h] = ger;
  }
}

// The same backward BY SOURCE (the slicer's CSL_T_INDPTR / CSL_T_INDICES, CSL_FLAG_TRANSPOSE_ALL): one wave per source
// row u walks the destinations of its edges, so g_z[u, :] and g_el[u, :] are accumulated in registers and WRITTEN once --
// no zero fill of the 0.7 GB gradient of z, no fp32 atomics on it; what is left to atomics is g_er (H floats per edge
// instead of H * D; the caller zeroes it).  Rows [n_src, n_pad) of g_z are zeroed (GEMM operand padding).
__global__ __launch_bounds__(BLK) void k_gat_bwd_t(const int* __restrict__ tptr, const int* __restrict__ trow,
                                                   long long n_src, long long n_pad, const float* __restrict__ el,
                                                   const float* __restrict__ er, const float* __restrict__ z, int H, int D,
                                                   float slope, const float* __restrict__ m_in,
                                                   const float* __restrict__ g_s, const float* __restrict__ g_n,
                                                   float* __restrict__ g_el, float* g_er, float* __restrict__ g_z) {
  const int lane = threadIdx.x & 63;
  const long long u = (long long)blockIdx.x * (BLK / 64) + (threadIdx.x >> 6);
  if (u >= n_pad) return;
  const int C = H * D, gsz = D / 4, lpc = (64 / gsz) * gsz, gl = lane % gsz;
  const int j0 = u < n_src ? tptr[u] : 0, j1 = u < n_src ? tptr[u + 1] : 0;
  for (int c0 = 0; c0 < C; c0 += lpc * 4) {
    const int c = c0 + lane * 4;
    const bool on = lane < lpc && c < C;
    const int h = on ? c / D : 0;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), zv = acc;
    float elu = 0.f, gel = 0.f;
    if (u < n_src) {
      elu = el[u * H + h];
      if (on) zv = *reinterpret_cast<const float4*>(z + u * C + c);
    }
    for (int j = j0; j < j1; j++) {
      const int r = trow[j];  // (wave-uniform)
      if (r < 0) continue;    // the node's self entry: attention runs over the sampled edges only
      const float raw = elu + er[(long long)r * H + h];
      const float p = expf(leaky(raw, slope) - m_in[(long long)r * H + h]);
      float dot = 0.f;
      float4 gn = make_float4(0.f, 0.f, 0.f, 0.f);
      if (on) {
        gn = *reinterpret_cast<const float4*>(g_n + (long long)r * C + c);
        dot = gn.x * zv.x + gn.y * zv.y + gn.z * zv.z + gn.w * zv.w;
      }
      for (int o = 1; o < gsz; o <<= 1) {  // sum over the head's lane group
        const float t = __shfl_down(dot, o);
        if (gl + o < gsz) dot += t;
      }
      dot = __shfl(dot, lane - gl);
      const float gsc = (g_s[(long long)r * H + h] + dot) * p * (raw > 0.f ? 1.f : slope);
      acc.x += p * gn.x, acc.y += p * gn.y, acc.z += p * gn.z, acc.w += p * gn.w;
      if (on && c % D == 0) {
        gel += gsc;
        atomicAdd(g_er + (long long)r * H + h, gsc);
      }
    }
    if (on) {
      *reinterpret_cast<float4*>(g_z + u * C + c) = acc;
      if (u < n_src && c % D == 0) g_el[u * H + h] = gel;
    }
  }
}

// k_gat_bwd_t with the logits' backward folded in (csl_gat_bwd_t_fused_f32).  A source row's g_el[u, h] is complete when
// its wave has walked its list, and z[u, :] is already in the wave's registers, so the el half of DistGATConv.project's
// backward -- g_z[u, h, :] += g_el[u, h] a_l[h, :] and g_attn_l[h, :] += g_el[u, h] z[u, h, :] -- costs no memory pass of
// its own (k_gat_logits_bwd re-read z and read-modify-wrote all of g_z: 112 us per layer in the config-5 step).  The er
// half depends on g_er of the DESTINATIONS, complete only when every source row has been walked: it runs afterwards over
// the destination rows alone (k_gat_logits_bwd_dst: n_out rows, a tenth of the sources).  A workgroup takes
// rows_per_block source rows, a wave every fourth of them; part_l[block][C] = the block's share of g_attn_l.
__global__ __launch_bounds__(BLK) void k_gat_bwd_t2(const int* __restrict__ tptr, const int* __restrict__ trow,
                                                    long long n_src, long long n_pad, const float* __restrict__ el,
                                                    const float* __restrict__ er, const float* __restrict__ z, int H, int D,
                                                    float slope, const float* __restrict__ m_in,
                                                    const float* __restrict__ g_s, const float* __restrict__ g_n,
                                                    const float* __restrict__ al, float* g_er, float* __restrict__ g_z,
                                                    float* __restrict__ part_l, long long rows_per_block) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int C = H * D, gsz = D / 4, lpc = (64 / gsz) * gsz, gl = lane % gsz;
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r_end = r0 + rows_per_block < n_pad ? r0 + rows_per_block : n_pad;
  __shared__ float4 s_l[BLK];
  for (int c0 = 0; c0 < C; c0 += lpc * 4) {   // block-uniform
    const int c = c0 + lane * 4;
    const bool on = lane < lpc && c < C;
    const int h = on ? c / D : 0;
    float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f), pl = a4;
    if (on) a4 = *reinterpret_cast<const float4*>(al + c);
    for (long long u = r0 + w; u < r_end; u += BLK / 64) {
      const int j0 = u < n_src ? tptr[u] : 0, j1 = u < n_src ? tptr[u + 1] : 0;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), zv = acc;
      float elu = 0.f, gel = 0.f;
      if (u < n_src) {
        elu = el[u * H + h];
        if (on) zv = *reinterpret_cast<const float4*>(z + u * C + c);
      }
      for (int j = j0; j < j1; j++) {
        const int r = trow[j];  // (wave-uniform)
        if (r < 0) continue;    // the node's self entry: attention runs over the sampled edges only
        const float raw = elu + er[(long long)r * H + h];
        const float p = expf(leaky(raw, slope) - m_in[(long long)r * H + h]);
        float dot = 0.f;
        float4 gn = make_float4(0.f, 0.f, 0.f, 0.f);
        if (on) {
          gn = *reinterpret_cast<const float4*>(g_n + (long long)r * C + c);
          dot = gn.x * zv.x + gn.y * zv.y + gn.z * zv.z + gn.w * zv.w;
        }
        for (int o = 1; o < gsz; o <<= 1) {  // sum over the head's lane group
          const float t = __shfl_down(dot, o);
          if (gl + o < gsz) dot += t;
        }
        dot = __shfl(dot, lane - gl);
        const float gsc = (g_s[(long long)r * H + h] + dot) * p * (raw > 0.f ? 1.f : slope);
        acc.x += p * gn.x, acc.y += p * gn.y, acc.z += p * gn.z, acc.w += p * gn.w;
        gel += gsc;   // (every lane of the head's group carries the same value)
        if (on && c % D == 0) atomicAdd(g_er + (long long)r * H + h, gsc);
      }
      if (on) {
        acc.x += gel * a4.x, acc.y += gel * a4.y, acc.z += gel * a4.z, acc.w += gel * a4.w;   // + g_el a_l
        *reinterpret_cast<float4*>(g_z + u * C + c) = acc;
        pl.x += gel * zv.x, pl.y += gel * zv.y, pl.z += gel * zv.z, pl.w += gel * zv.w;       // g_attn_l's share
      }
    }
    __syncthreads();
    s_l[threadIdx.x] = pl;
    __syncthreads();
    if (w == 0 && on) {
      for (int k = 1; k < BLK / 64; k++) add4(pl, s_l[k * 64 + lane]);
      *reinterpret_cast<float4*>(part_l + (long long)blockIdx.x * C + c) = pl;
    }
  }
}

// the er half: destination r's logit gradient goes to the z row of its self source
__global__ __launch_bounds__(BLK) void k_gat_logits_bwd_dst(const float* __restrict__ z, const float* __restrict__ ar,
                                                            const int* __restrict__ self_ids, const float* __restrict__ g_er,
                                                            long long n_out, int H, int D, float* __restrict__ g_z,
                                                            float* __restrict__ part_r, long long rows_per_block) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int C = H * D, gsz = D / 4, lpc = (64 / gsz) * gsz;
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r_end = r0 + rows_per_block < n_out ? r0 + rows_per_block : n_out;
  __shared__ float4 s_r[BLK];
  for (int c0 = 0; c0 < C; c0 += lpc * 4) {
    const int c = c0 + lane * 4;
    const bool on = lane < lpc && c < C;
    const int h = on ? c / D : 0;
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f), pr = b4;
    if (on) b4 = *reinterpret_cast<const float4*>(ar + c);
    for (long long r = r0 + w; r < r_end; r += BLK / 64) {
      const long long u = self_ids[r];
      if (!on || u < 0) continue;
      const float ge = g_er[r * H + h];
      const float4 zv = *reinterpret_cast<const float4*>(z + u * C + c);
      float4 o = *reinterpret_cast<const float4*>(g_z + u * C + c);
      o.x += ge * b4.x, o.y += ge * b4.y, o.z += ge * b4.z, o.w += ge * b4.w;
      *reinterpret_cast<float4*>(g_z + u * C + c) = o;
      pr.x += ge * zv.x, pr.y += ge * zv.y, pr.z += ge * zv.z, pr.w += ge * zv.w;
    }
    __syncthreads();
    s_r[threadIdx.x] = pr;
    __syncthreads();
    if (w == 0 && on) {
      for (int k = 1; k < BLK / 64; k++) add4(pr, s_r[k * 64 + lane]);
      *reinterpret_cast<float4*>(part_r + (long long)blockIdx.x * C + c) = pr;
    }
  }
}

// ---- fused GraphSAGE layer pieces (one part per GPU and the single-GPU trainer) --------------------------
// k_sage_cat: builds the operand of Linear(2*in, out) in ONE pass (dist_sageconv.py:66-80: self_gather, gather,
// slice_owned_nodes, mean, concat):
//   cat[r, 0:H)   = act(x[map(self_ids[r])])
//   cat[r, H:2H)  = 1/max(deg_r,1) * sum over the CSR row of act(x[map(indices[e])])      (indptr != null)
//                 = 1/max(deg[r],1) * agg[owned[r]]                                       (indptr == null: the sums
//                                                                         were merged across parts beforehand)
// map(i) = rowmap ? rowmap[i] : i  (the deepest layer reads the resident feature table through the slice's
// in_nodes, so the gathered input matrix is never materialised); act = ReLU when relu_in (the previous
// layer's pre-activation output is consumed directly: no separate activation pass).  Rows [n, n_pad) are zeroed.
template <int G>
__global__ __launch_bounds__(BLK) void k_sage_cat(const int* __restrict__ indptr, const int* __restrict__ indices,
                                                  const int* __restrict__ self_ids, const int* __restrict__ owned,
                                                  const int* __restrict__ deg, const int* __restrict__ rowmap,
                                                  const float* __restrict__ x, long long ldx,
                                                  const float* __restrict__ agg, long long lda, long long n,
                                                  long long n_pad, float* __restrict__ cat, long long ldc, int H,
                                                  int relu_in) {
  constexpr int RPB = BLK / G;
  const int lane = threadIdx.x % G;
  const long long r = (long long)blockIdx.x * RPB + threadIdx.x / G;
  if (r >= n_pad) return;
  float* out = cat + r * ldc;
  if (r >= n) {
    for (int c = lane * 4; c < 2 * H; c += G * 4) *reinterpret_cast<float4*>(out + c) = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  const float lo = relu_in ? 0.f : -__builtin_inff();
  const long long sid = self_ids[r];
  const long long srow = sid < 0 ? -1 : (rowmap ? (long long)rowmap[sid] : sid);
  long long e0 = 0, e1 = 0;
  if (indptr) {
    e0 = indptr[r];
    e1 = indptr[r + 1];
  }
  const long long d = deg ? (long long)deg[r] : (e1 - e0);
  const float inv = 1.0f / (float)(d > 1 ? d : 1);
  for (int c = lane * 4; c < H; c += G * 4) {
    float4 sv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (srow >= 0) {
      sv = *reinterpret_cast<const float4*>(x + srow * ldx + c);
      sv.x = fmaxf(sv.x, lo), sv.y = fmaxf(sv.y, lo), sv.z = fmaxf(sv.z, lo), sv.w = fmaxf(sv.w, lo);
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (indptr) {
      long long e = e0;
      for (; e + 4 <= e1; e += 4) {  // four source rows in flight per lane, fixed summation order
        long long s0 = indices[e], s1 = indices[e + 1], s2 = indices[e + 2], s3 = indices[e + 3];
        if (rowmap) s0 = rowmap[s0], s1 = rowmap[s1], s2 = rowmap[s2], s3 = rowmap[s3];
        float4 v0 = *reinterpret_cast<const float4*>(x + s0 * ldx + c);
        float4 v1 = *reinterpret_cast<const float4*>(x + s1 * ldx + c);
        float4 v2 = *reinterpret_cast<const float4*>(x + s2 * ldx + c);
        float4 v3 = *reinterpret_cast<const float4*>(x + s3 * ldx + c);
        acc.x += fmaxf(v0.x, lo), acc.y += fmaxf(v0.y, lo), acc.z += fmaxf(v0.z, lo), acc.w += fmaxf(v0.w, lo);
        acc.x += fmaxf(v1.x, lo), acc.y += fmaxf(v1.y, lo), acc.z += fmaxf(v1.z, lo), acc.w += fmaxf(v1.w, lo);
        acc.x += fmaxf(v2.x, lo), acc.y += fmaxf(v2.y, lo), acc.z += fmaxf(v2.z, lo), acc.w += fmaxf(v2.w, lo);
        acc.x += fmaxf(v3.x, lo), acc.y += fmaxf(v3.y, lo), acc.z += fmaxf(v3.z, lo), acc.w += fmaxf(v3.w, lo);
      }
      for (; e < e1; e++) {
        long long s0 = indices[e];
        if (rowmap) s0 = rowmap[s0];
        const float4 v0 = *reinterpret_cast<const float4*>(x + s0 * ldx + c);
        acc.x += fmaxf(v0.x, lo), acc.y += fmaxf(v0.y, lo), acc.z += fmaxf(v0.z, lo), acc.w += fmaxf(v0.w, lo);
      }
    } else {
      acc = *reinterpret_cast<const float4*>(agg + (long long)owned[r] * lda + c);
    }
    acc.x *= inv, acc.y *= inv, acc.z *= inv, acc.w *= inv;
    *reinterpret_cast<float4*>(out + c) = sv;
    *reinterpret_cast<float4*>(out + H + c) = acc;
  }
}

// gradient of k_sage_cat's merged-sums form (several parts): plain stores, the index lists are unique
template <int G>
__global__ __launch_bounds__(BLK) void k_sage_cat_rows_bwd(const int* __restrict__ self_ids, const int* __restrict__ owned,
                                                           const int* __restrict__ deg, long long n,
                                                           const float* __restrict__ gcat, long long ldg,
                                                           float* __restrict__ gx, float* __restrict__ gagg, int H) {
  constexpr int RPB = BLK / G;
  const int lane = threadIdx.x % G;
  const long long r = (long long)blockIdx.x * RPB + threadIdx.x / G;
  if (r >= n) return;
  const long long sid = self_ids[r], ow = owned[r];
  const long long d = deg[r];
  const float inv = 1.0f / (float)(d > 1 ? d : 1);
  for (int c = lane * 4; c < H; c += G * 4) {
    const float4 a = *reinterpret_cast<const float4*>(gcat + r * ldg + c);
    float4 b = *reinterpret_cast<const float4*>(gcat + r * ldg + H + c);
    b.x *= inv, b.y *= inv, b.z *= inv, b.w *= inv;
    if (gx && sid >= 0) *reinterpret_cast<float4*>(gx + sid * (long long)H + c) = a;
    if (gagg) *reinterpret_cast<float4*>(gagg + ow * (long long)H + c) = b;
  }
}

// k_sage_cat_bwd: gradient of k_sage_cat (CSR form) w.r.t. x, accumulated with fp32 atomics into a zeroed gx:
//   gx[self_ids[r]] += gcat[r, 0:H),   gx[indices[e]] += gcat[r, H:2H) / max(deg_r, 1) for every edge of row r.
// One wave per output row, 64 consecutive floats per atomic instruction (as k_spmm_sum_bwd).  When the layer
// consumed a pre-activation input (relu_in) the caller masks gx afterwards (k_relu_bwd_colsum of the layer below).
__global__ __launch_bounds__(BLK) void k_sage_cat_bwd(const int* __restrict__ indptr, const int* __restrict__ indices,
                                                      const int* __restrict__ self_ids, long long n,
                                                      const float* __restrict__ gcat, long long ldg,
                                                      float* __restrict__ gx, long long ldx, int H) {
  const int lane = threadIdx.x & 63;
  const long long r = (long long)blockIdx.x * (BLK / 64) + (threadIdx.x >> 6);
  if (r >= n) return;
  const long long e0 = indptr[r], e1 = indptr[r + 1];
  const float inv = 1.0f / (float)(e1 - e0 > 1 ? e1 - e0 : 1);
  const long long sid = self_ids[r];
  for (int c = lane; c < H; c += 64) {
    if (sid >= 0) atomicAdd(gx + sid * ldx + c, gcat[r * ldg + c]);
    const float v = gcat[r * ldg + H + c] * inv;
    for (long long e = e0; e < e1; e++) atomicAdd(gx + (long long)indices[e] * ldx + c, v);
  }
}

// k_relu_bwd_colsum: out[r, :] = y ? (y[r, :] > 0 ? g[r, :] : 0) : g[r, :] for r < n, zero rows up to n_pad, and
// the column sums of `out` (the bias gradient): ReLU backward, the row padding of the GEMM operand and the bias
// reduction in one pass over the gradient.  Two stages, no atomics and nothing to pre-zero: every block leaves the
// column sums of its RB_ROWS rows in partial[block][H], k_colsum_finish adds the blocks up.
constexpr int RB_U = 4;       // rows a thread has in flight
// rows per block: 128 (enough blocks in flight to cover the memory latency), more only beyond 2048 blocks
__host__ __device__ inline long long rb_rows(long long n_pad) {
  long long r = 128;
  while ((n_pad + r - 1) / r > 2048) r *= 2;
  return r;
}
template <int G>
__global__ __launch_bounds__(BLK) void k_relu_bwd_colsum(const float* __restrict__ g, long long ldg,
                                                         const float* __restrict__ y, long long ldy, long long n,
                                                         long long n_pad, float* __restrict__ out, long long ldo,
                                                         float* __restrict__ partial, int H, int vec,
                                                         long long rows_per_block) {
  constexpr int RPB = BLK / G;  // row groups of the block
  const int lane = threadIdx.x % G, sub = threadIdx.x / G;
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r_end = r0 + rows_per_block < n_pad ? r0 + rows_per_block : n_pad;
  __shared__ float s_sum[BLK * 4];
  // (the column loop is block-uniform: every thread reaches the barriers)
  for (int c0 = 0; c0 < H; c0 += G * 4) {
    const int c = c0 + lane * 4;
    const bool full = vec && c + 3 < H;  // aligned float4 accesses, else element-wise with bounds
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long long rb = r0 + sub; rb < r_end; rb += RPB * RB_U) {
      float4 v[RB_U], a[RB_U];
#pragma unroll
      for (int u = 0; u < RB_U; u++) {
        const long long r = rb + (long long)u * RPB;
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        a[u] = make_float4(1.f, 1.f, 1.f, 1.f);
        if (r < n) {
          if (full) {
            v[u] = *reinterpret_cast<const float4*>(g + r * ldg + c);
            if (y) a[u] = *reinterpret_cast<const float4*>(y + r * ldy + c);
          } else {
            if (c < H) v[u].x = g[r * ldg + c];
            if (c + 1 < H) v[u].y = g[r * ldg + c + 1];
            if (c + 2 < H) v[u].z = g[r * ldg + c + 2];
            if (c + 3 < H) v[u].w = g[r * ldg + c + 3];
            if (y) {
              if (c < H) a[u].x = y[r * ldy + c];
              if (c + 1 < H) a[u].y = y[r * ldy + c + 1];
              if (c + 2 < H) a[u].z = y[r * ldy + c + 2];
              if (c + 3 < H) a[u].w = y[r * ldy + c + 3];
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < RB_U; u++) {
        const long long r = rb + (long long)u * RPB;
        if (r >= r_end) continue;
        float4 w = v[u];
        w.x = a[u].x > 0.f ? w.x : 0.f, w.y = a[u].y > 0.f ? w.y : 0.f, w.z = a[u].z > 0.f ? w.z : 0.f, w.w = a[u].w > 0.f ? w.w : 0.f;
        add4(acc, w);
        if (full) {
          *reinterpret_cast<float4*>(out + r * ldo + c) = w;
        } else {
          if (c < H) out[r * ldo + c] = w.x;
          if (c + 1 < H) out[r * ldo + c + 1] = w.y;
          if (c + 2 < H) out[r * ldo + c + 2] = w.z;
          if (c + 3 < H) out[r * ldo + c + 3] = w.w;
        }
      }
    }
    // the RPB row groups of the block hold partial sums of the same columns
    __syncthreads();
    reinterpret_cast<float4*>(s_sum)[threadIdx.x] = acc;
    __syncthreads();
    if (sub == 0) {
      for (int k = 1; k < RPB; k++) add4(acc, reinterpret_cast<float4*>(s_sum)[k * G + lane]);
      float* p = partial + (long long)blockIdx.x * H;
      if (c < H) p[c] = acc.x;
      if (c + 1 < H) p[c + 1] = acc.y;
      if (c + 2 < H) p[c + 2] = acc.z;
      if (c + 3 < H) p[c + 3] = acc.w;
    }
  }
}
// out[c] = sum over the blocks of partial[block][c]: 4 x 64 threads per 64 columns, the four groups split the
// blocks; eight independent loads in flight per thread
__global__ __launch_bounds__(BLK) void k_colsum_finish(const float* __restrict__ partial, long long nblk, int H,
                                                       float* __restrict__ out) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
  __shared__ float s_q[BLK];
  float acc = 0.f;
  if (c < H) {
    constexpr int Q = BLK / 64, U = 8;
    long long b = q;
    for (; b + (U - 1) * Q < nblk; b += U * Q) {
      float v[U];
#pragma unroll
      for (int u = 0; u < U; u++) v[u] = partial[(b + u * Q) * H + c];
#pragma unroll
      for (int u = 0; u < U; u++) acc += v[u];
    }
    for (; b < nblk; b += Q) acc += partial[b * H + c];
  }
  s_q[threadIdx.x] = acc;
  __syncthreads();
  if (q == 0 && c < H) out[c] = s_q[threadIdx.x] + s_q[threadIdx.x + 64] + s_q[threadIdx.x + 128] + s_q[threadIdx.x + 192];
}

// k_sage_cat_bwd_t: the backward of k_sage_cat's CSR form as a GATHER over the slice by source the slicer emits
// (CSL_FLAG_TRANSPOSE: cslicer_hip.h, CSL_T_INDPTR).  k_sage_cat_bwd scatters with fp32 atomics into a zeroed buffer
// (middle layer of the headline model: 84 MB of zero fill + 100 MB of L2 atomics, then k_relu_bwd_colsum reads and
// rewrites all of it); here every source-gradient row is written exactly once, so the ReLU mask of the layer below,
// the row padding of its GEMM operand and its bias column sums ride on the same pass:
//   out[u, :] = mask_u .* sum_j ( t < 0 ? gcat[~t, 0:H) : gcat[t, H:2H) / max(deg_t, 1) ),  t = trow[j], j over u's list;
//   deg_t = indptr[t+1] - indptr[t] (the forward's mean divisor);  mask_u = y ? y[u, :] > 0 : 1;
// rows [n_src, n_pad) zero; per-block column sums to `partial` (k_colsum_finish).  The order of a list is fixed by the
// slicer (sorted), so the result is deterministic, unlike the atomic scatter's.
// G lanes per source row, a float4 column chunk per lane and pass: G = H / 4 up to a whole wave, so that a row's index
// entries are read once (H = 256: one wave per row, 1.85 -> 1.91 k minibatches/s against 16 lanes x 4 passes)
template <int G>
__global__ __launch_bounds__(BLK) void k_sage_cat_bwd_t(const int* __restrict__ tptr, const int* __restrict__ trow,
                                                        const int* __restrict__ indptr, const float* __restrict__ gcat,
                                                        long long ldg, const float* __restrict__ y, long long ldy,
                                                        long long n_src, long long n_pad, float* __restrict__ out,
                                                        long long ldo, float* __restrict__ partial, int H,
                                                        long long rows_per_block, int thr) {
  // thr > 0: rows whose list is longer than thr entries are LEFT AT ZERO here (hub nodes: one wave walking thousands of
  // entries); k_sage_cat_bwd_t_hub / _hubfin fill them in
  constexpr int RPB = BLK / G;
  const int lane = threadIdx.x % G, sub = threadIdx.x / G;
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r_end = r0 + rows_per_block < n_pad ? r0 + rows_per_block : n_pad;
  __shared__ float s_sum[BLK * 4];
  for (int c0 = 0; c0 < H; c0 += G * 4) {  // block-uniform
    const int c = c0 + lane * 4;
    float4 colacc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long long u = r0 + sub; u < r_end; u += RPB) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      if (u < n_src && c < H && !(thr > 0 && tptr[u + 1] - tptr[u] > thr)) {
        const int j0 = tptr[u], j1 = tptr[u + 1];
        float4 a = make_float4(1.f, 1.f, 1.f, 1.f);
        if (y) a = *reinterpret_cast<const float4*>(y + u * ldy + c);
        int j = j0;
        for (; j + 2 <= j1; j += 2) {  // two contributions in flight
          const int ta = trow[j], tb = trow[j + 1];
          const int ra = ta < 0 ? ~ta : ta, rb = tb < 0 ? ~tb : tb;
          // (a self entry's divisor load is issued too and ignored: no branch between the loads)
          // (indptr == null: the caller's rows are already divided -- the split-parallel rank step)
          const int da = indptr ? indptr[ra + 1] - indptr[ra] : 1, db = indptr ? indptr[rb + 1] - indptr[rb] : 1;
          const float4 va = *reinterpret_cast<const float4*>(gcat + (long long)ra * ldg + (ta < 0 ? 0 : H) + c);
          const float4 vb = *reinterpret_cast<const float4*>(gcat + (long long)rb * ldg + (tb < 0 ? 0 : H) + c);
          const float wa = ta < 0 ? 1.f : 1.f / (float)(da > 1 ? da : 1);
          const float wb = tb < 0 ? 1.f : 1.f / (float)(db > 1 ? db : 1);
          acc.x += wa * va.x, acc.y += wa * va.y, acc.z += wa * va.z, acc.w += wa * va.w;
          acc.x += wb * vb.x, acc.y += wb * vb.y, acc.z += wb * vb.z, acc.w += wb * vb.w;
        }
        if (j < j1) {
          const int ta = trow[j];
          const int ra = ta < 0 ? ~ta : ta;
          const int da = indptr ? indptr[ra + 1] - indptr[ra] : 1;
          const float4 va = *reinterpret_cast<const float4*>(gcat + (long long)ra * ldg + (ta < 0 ? 0 : H) + c);
          const float wa = ta < 0 ? 1.f : 1.f / (float)(da > 1 ? da : 1);
          acc.x += wa * va.x, acc.y += wa * va.y, acc.z += wa * va.z, acc.w += wa * va.w;
        }
        acc.x = a.x > 0.f ? acc.x : 0.f, acc.y = a.y > 0.f ? acc.y : 0.f, acc.z = a.z > 0.f ? acc.z : 0.f, acc.w = a.w > 0.f ? acc.w : 0.f;
        add4(colacc, acc);
      }
      if (c < H) *reinterpret_cast<float4*>(out + u * ldo + c) = acc;
    }
    __syncthreads();
    reinterpret_cast<float4*>(s_sum)[threadIdx.x] = colacc;
    __syncthreads();
    if (sub == 0 && c < H) {
      for (int k = 1; k < RPB; k++) add4(colacc, reinterpret_cast<float4*>(s_sum)[k * G + lane]);
      *reinterpret_cast<float4*>(partial + (long long)blockIdx.x * H + c) = colacc;
    }
  }
}
__host__ __device__ inline long long tb_rows(long long n_pad) {
  long long r = 16;   // (64 until round 3: the 6 k-row top layer of the headline model ran on 110 workgroups, 17 us)
  while ((n_pad + r - 1) / r > 2048) r *= 2;
  return r;
}

// ---- the split-parallel rank step's backward by source (sage_step.hip, csl_sage_rank_fwd_bwd_f32).  The operand
// gradient gcat [n_owned, 2H] is in OWNED-row order; the slice by source names OUT rows.  g2 [n_out, 2H] is the same in
// out-row order with the mean half already divided by the TRUE degree (the forward's divisor: the merged sums of all
// parts) -- G2[owned[j], 0:H) = gcat[j, 0:H), G2[owned[j], H:2H) = gcat[j, H:2H) / max(deg[j], 1); the mean halves of the
// rows peers own arrive from the reverse exchange (k_scatter_rows_set) -- so that csl_sage_cat_bwd_t_f32 with
// indptr = NULL gathers the input gradient over it: no atomics, no zero fill, ReLU mask + padding + bias sums in the pass.
template <int G>
__global__ __launch_bounds__(BLK) void k_rank_g2(const int* __restrict__ owned, const int* __restrict__ deg, long long n,
                                                 const float* __restrict__ gcat, long long ldg, float* __restrict__ g2,
                                                 long long ld2, int H) {
  constexpr int RPB = BLK / G;
  const int lane = threadIdx.x % G;
  const long long j = (long long)blockIdx.x * RPB + threadIdx.x / G;
  if (j >= n) return;
  const long long o = owned[j];
  const int d = deg[j];
  const float inv = 1.0f / (float)(d > 1 ? d : 1);
  for (int c = lane * 4; c < H; c += G * 4) {
    const float4 a = *reinterpret_cast<const float4*>(gcat + j * ldg + c);
    float4 m = *reinterpret_cast<const float4*>(gcat + j * ldg + H + c);
    m.x *= inv, m.y *= inv, m.z *= inv, m.w *= inv;
    *reinterpret_cast<float4*>(g2 + o * ld2 + c) = a;
    *reinterpret_cast<float4*>(g2 + o * ld2 + H + c) = m;
  }
}

// dst[idx[k], 0:H) = src[k, 0:H)   (idx unique)
template <int G>
__global__ __launch_bounds__(BLK) void k_scatter_rows_set(float* __restrict__ dst, long long ldd, const int* __restrict__ idx,
                                                          long long n, const float* __restrict__ src, long long lds, int H) {
  constexpr int RPB = BLK / G;
  const int lane = threadIdx.x % G;
  const long long k = (long long)blockIdx.x * RPB + threadIdx.x / G;
  if (k >= n) return;
  const long long o = idx[k];
  for (int c = lane * 4; c < H; c += G * 4)
    *reinterpret_cast<float4*>(dst + o * ldd + c) = *reinterpret_cast<const float4*>(src + k * lds + c);
}

// ---- hub lists.  A node that thousands of the minibatch's rows sampled has a list of thousands of entries in the slice
// by source (power-law graphs; src/gnn/sage.cu:20-28 walks every edge with its own thread for the same reason).
// k_sage_cat_bwd_t leaves such rows at zero (thr); here the ENTRIES of the slice are cut into segments of HUB_SEG, a
// workgroup per segment: it finds the rows its segment overlaps (binary search in t_indptr), and for every hub row among
// them sums its part of the list -- G lanes per column quad, the BLK / G lane groups striding over the entries, one LDS
// reduction -- and adds the partial sum to the row with fp32 atomics (a hub of 5,000 entries: 10 segments, 10 adds
// per element; the order of those adds is the only non-determinism left).  Segments without a hub entry cost two loads.
constexpr int HUB_SEG = 512;
template <int G>
__global__ __launch_bounds__(BLK) void k_sage_cat_bwd_t_hub(const int* __restrict__ tptr, const int* __restrict__ trow,
                                                            const int* __restrict__ indptr, const float* __restrict__ gcat,
                                                            long long ldg, long long n_src, float* __restrict__ out,
                                                            long long ldo, int H, int thr) {
  constexpr int RPB = BLK / G;
  const int lane = threadIdx.x % G, sub = threadIdx.x / G;
  const long long total = tptr[n_src];
  const long long e0 = (long long)blockIdx.x * HUB_SEG, e1 = e0 + HUB_SEG < total ? e0 + HUB_SEG : total;
  if (e0 >= e1) return;
  long long lo = 0, hi = n_src;   // the first row whose list ends beyond e0 (block-uniform: every thread searches)
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    if (tptr[mid + 1] > e0) hi = mid; else lo = mid + 1;
  }
  long long lo2 = lo, hi2 = n_src;   // the first row whose list starts at or beyond e1
  while (lo2 < hi2) {
    const long long mid = (lo2 + hi2) >> 1;
    if (tptr[mid] >= e1) hi2 = mid; else lo2 = mid + 1;
  }
  // the hub rows among [lo, lo2): looked for by all threads at once (a segment of short lists overlaps hundreds of rows)
  __shared__ int s_hub[HUB_SEG / (CSL_T_SORTED_MAX + 1) + 4];   // (at most 2 + 510 / 129 lists longer than the threshold overlap a segment)
  __shared__ int s_nhub;
  __shared__ float4 s_part[BLK];
  if (threadIdx.x == 0) s_nhub = 0;
  __syncthreads();
  for (long long u = lo + threadIdx.x; u < lo2; u += BLK)
    if (tptr[u + 1] - tptr[u] > thr) s_hub[atomicAdd(&s_nhub, 1)] = (int)(u - lo);
  __syncthreads();
  const int nhub = s_nhub;
  for (int q = 0; q < nhub; q++) {
    const long long u = lo + s_hub[q];
    const long long j0 = tptr[u], j1 = tptr[u + 1];
    const long long a0 = j0 > e0 ? j0 : e0, a1 = j1 < e1 ? j1 : e1;
    for (int c0 = 0; c0 < H; c0 += G * 4) {   // block-uniform
      const int c = c0 + lane * 4;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < H) {
        for (long long j = a0 + sub; j < a1; j += RPB) {
          const int t = trow[j];
          const int r = t < 0 ? ~t : t;
          const int d = indptr ? indptr[r + 1] - indptr[r] : 1;
          const float4 v = *reinterpret_cast<const float4*>(gcat + (long long)r * ldg + (t < 0 ? 0 : H) + c);
          const float w = t < 0 ? 1.f : 1.f / (float)(d > 1 ? d : 1);
          acc.x += w * v.x, acc.y += w * v.y, acc.z += w * v.z, acc.w += w * v.w;
        }
      }
      __syncthreads();
      s_part[threadIdx.x] = acc;
      __syncthreads();
      if (sub == 0 && c < H) {
        for (int k = 1; k < RPB; k++) add4(acc, s_part[k * G + lane]);
        float* o = out + u * ldo + c;
        atomicAdd(o, acc.x), atomicAdd(o + 1, acc.y), atomicAdd(o + 2, acc.z), atomicAdd(o + 3, acc.w);
      }
    }
  }
}

// the hub rows' ReLU mask and their share of the column sums (everything k_sage_cat_bwd_t does after a row's sum)
template <int G>
__global__ __launch_bounds__(BLK) void k_sage_cat_bwd_t_hubfin(const int* __restrict__ tptr, const float* __restrict__ y,
                                                               long long ldy, long long n_src, float* __restrict__ out,
                                                               long long ldo, float* __restrict__ partial, int H,
                                                               long long rows_per_block, int thr) {
  constexpr int RPB = BLK / G;
  const int lane = threadIdx.x % G, sub = threadIdx.x / G;
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r_end = r0 + rows_per_block < n_src ? r0 + rows_per_block : n_src;
  __shared__ float4 s_sum[BLK];
  for (int c0 = 0; c0 < H; c0 += G * 4) {
    const int c = c0 + lane * 4;
    float4 colacc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long long u = r0 + sub; u < r_end; u += RPB) {
      if (c < H && tptr[u + 1] - tptr[u] > thr) {
        float4 v = *reinterpret_cast<const float4*>(out + u * ldo + c);
        if (y) {
          const float4 a = *reinterpret_cast<const float4*>(y + u * ldy + c);
          v.x = a.x > 0.f ? v.x : 0.f, v.y = a.y > 0.f ? v.y : 0.f, v.z = a.z > 0.f ? v.z : 0.f, v.w = a.w > 0.f ? v.w : 0.f;
          *reinterpret_cast<float4*>(out + u * ldo + c) = v;
        }
        add4(colacc, v);
      }
    }
    __syncthreads();
    s_sum[threadIdx.x] = colacc;
    __syncthreads();
    if (sub == 0 && c < H) {
      for (int k = 1; k < RPB; k++) add4(colacc, s_sum[k * G + lane]);
      *reinterpret_cast<float4*>(partial + (long long)blockIdx.x * H + c) = colacc;
    }
  }
}

// k_softmax_ce: cross-entropy of one minibatch (train.py:86 loss_fn) forward AND backward in one pass:
//   loss = -scale * sum_r log softmax(logits[r])[label_r],   grad[r, :] = scale * (softmax(logits[r]) - onehot(label_r))
// label_r = labels[rowmap ? rowmap[ids[r]] : ids[r]] (ids = the seeds' node ids).  One wave per row (C <= 4096); a
// block leaves the loss of its four rows in partial[block], k_colsum_finish (H = 1) adds the blocks up.
constexpr int SM_CMAX = 256;  // widest row whose per-block column sums fit the LDS tile below
__global__ __launch_bounds__(BLK) void k_softmax_ce(const float* __restrict__ logits, long long ldl, long long n,
                                                    long long n_pad, int C, const int* __restrict__ ids,
                                                    const int* __restrict__ rowmap, const long long* __restrict__ labels,
                                                    float scale, float* __restrict__ partial, float* __restrict__ grad,
                                                    long long ldgr, float* __restrict__ colpart) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long long r = (long long)blockIdx.x * (BLK / 64) + w;
  __shared__ float s_l[BLK / 64];
  __shared__ float s_c[BLK / 64][SM_CMAX];
  float mine = 0.f;
  if (r < n) {
    const float* z = logits + r * ldl;
    float m = -__builtin_inff();
    for (int c = lane; c < C; c += 64) m = fmaxf(m, z[c]);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += expf(z[c] - m);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const long long node = ids[r];
    const long long lab = labels[rowmap ? (long long)rowmap[node] : node];
    const float inv = 1.0f / s;
    for (int c = lane; c < C; c += 64) {
      const float gv = scale * (expf(z[c] - m) * inv - (c == lab ? 1.f : 0.f));
      grad[r * ldgr + c] = gv;
      if (colpart) s_c[w][c] = gv;
    }
    mine = scale * (logf(s) + m - z[lab]);
  } else if (r < n_pad) {
    // padding rows of the GEMM operand the gradient becomes
    for (int c = lane; c < C; c += 64) grad[r * ldgr + c] = 0.f;
  }
  if (colpart && r >= n)
    for (int c = lane; c < C; c += 64) s_c[w][c] = 0.f;
  if (lane == 0) s_l[w] = mine;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = s_l[0] + s_l[1] + s_l[2] + s_l[3];
  // the block's column sums of the gradient (the bias gradient's first stage)
  if (colpart)
    for (int c = threadIdx.x; c < C; c += BLK)
      colpart[(long long)blockIdx.x * C + c] = (s_c[0][c] + s_c[1][c]) + (s_c[2][c] + s_c[3][c]);
}

// dst_j[c] = sum over b < nblk_j of src_j[b * H_j + c] for up to RM_MAX jobs in ONE launch: the second stage of every
// two-stage reduction of a training step (bias column sums, the loss, the row slabs of the weight gradients) when the
// results are only needed at the step's end -- one launch instead of one 5-9 us launch each.
constexpr int RM_MAX = CSL_REDUCE_MULTI_MAX;
struct ReduceJobs {
  const float* src[RM_MAX];
  float* dst[RM_MAX];
  long long nblk[RM_MAX];
  int H[RM_MAX];
  int first_block[RM_MAX + 1];
  int vec[RM_MAX];  // H % 4 == 0 and 16-byte aligned operands: float4 columns
  int count;
};
// A block is RM_L lanes along the columns x RM_G groups along the partial rows: a bias job has ~10^3 partial rows and only
// a few hundred columns, so the rows must be split finely (with 64 lanes x 4 groups its 320-long serial sums were 25 us of
// a 29 us launch); a slab job has 32 rows and 10^5 columns and is happy either way.
constexpr int RM_L = 16, RM_G = BLK / RM_L;
template <typename V>
__device__ __forceinline__ void reduce_job(const float* __restrict__ partial, float* __restrict__ dst, long long nblk,
                                           int H, int block_in_job) {
  constexpr int W = sizeof(V) / 4, U = 4;
  const int lane = threadIdx.x % RM_L, q = threadIdx.x / RM_L;
  const int c = (block_in_job * RM_L + lane) * W;
  __shared__ V s_q[BLK];
  V acc = {};
  if (c < H) {
    long long b = q;
    for (; b + (U - 1) * RM_G < nblk; b += U * RM_G) {
      V v[U];
#pragma unroll
      for (int u = 0; u < U; u++) v[u] = *reinterpret_cast<const V*>(partial + (b + u * RM_G) * H + c);
#pragma unroll
      for (int u = 0; u < U; u++) acc = acc + v[u];
    }
    for (; b < nblk; b += RM_G) acc = acc + *reinterpret_cast<const V*>(partial + b * H + c);
  }
  s_q[threadIdx.x] = acc;
  __syncthreads();
  for (int o = RM_G / 2; o > 0; o >>= 1) {  // tree over the groups
    if (q < o) s_q[threadIdx.x] = s_q[threadIdx.x] + s_q[threadIdx.x + o * RM_L];
    __syncthreads();
  }
  if (q == 0 && c < H) *reinterpret_cast<V*>(dst + c) = s_q[threadIdx.x];
}
__global__ __launch_bounds__(BLK) void k_reduce_multi(ReduceJobs jb) {
  int j = 0;
  while (j + 1 < jb.count && (int)blockIdx.x >= jb.first_block[j + 1]) j++;
  const int bj = (int)blockIdx.x - jb.first_block[j];
  if (jb.vec[j]) reduce_job<float4>(jb.src[j], jb.dst[j], jb.nblk[j], jb.H[j], bj);
  else reduce_job<float>(jb.src[j], jb.dst[j], jb.nblk[j], jb.H[j], bj);
}

// ---- Adam (python/train.py:83 torch.optim.Adam, no weight decay, no amsgrad) over every parameter tensor of the
// model in ONE launch: the model has six small tensors, the library's for-each form is 1-8 launches of 20-40 us.
//   m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= (lr / (1-b1^t)) * m / (sqrt(v) / sqrt(1-b2^t) + eps)
constexpr int ADAM_MAX = 24;
struct AdamArgs {
  float* p[ADAM_MAX];
  const float* g[ADAM_MAX];
  float* m[ADAM_MAX];
  float* v[ADAM_MAX];
  long long first_block[ADAM_MAX + 1];  // blocks of ADAM_CHUNK elements, tensors back to back
  long long n[ADAM_MAX];
  int count;
};
constexpr int ADAM_CHUNK = 1024;
__global__ __launch_bounds__(BLK) void k_adam(AdamArgs a, float b1, float b2, float step_size, float inv_sqrt_bc2,
                                              float eps) {
  int t = 0;
  while (t + 1 < a.count && (long long)blockIdx.x >= a.first_block[t + 1]) t++;
  const long long base = ((long long)blockIdx.x - a.first_block[t]) * ADAM_CHUNK;
  float* __restrict__ p = a.p[t];
  const float* __restrict__ g = a.g[t];
  float* __restrict__ m = a.m[t];
  float* __restrict__ v = a.v[t];
  for (long long i = base + threadIdx.x; i < base + ADAM_CHUNK && i < a.n[t]; i += BLK) {
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
  }
}

// ---- GAT attention logits: el[r, h] = <z[r, h, :], a_l[h, :]>, er likewise (DistGATConv.project).  A row-wise
// reduction over z (one HBM pass, 1 KB per row at 8 heads x 32); as `[rows, in] x [in, H]` library GEMMs the same
// numbers took 0.2-0.8 ms per launch (H = 8 columns: MT32x16x512 tiles), 1.7 ms of the 5.9 ms GAT step.
// One wave per row, a lane owns 4 consecutive columns (D % 4 == 0), heads are groups of D / 4 lanes.
__global__ __launch_bounds__(BLK) void k_gat_logits_fwd(const float* __restrict__ z, const float* __restrict__ al,
                                                        const float* __restrict__ ar, long long n, int H, int D,
                                                        float* __restrict__ el, float* __restrict__ er) {
  const int lane = threadIdx.x & 63;
  const long long r = (long long)blockIdx.x * (BLK / 64) + (threadIdx.x >> 6);
  if (r >= n) return;
  const int C = H * D, gsz = D / 4, lpc = (64 / gsz) * gsz, gl = lane % gsz;
  for (int c0 = 0; c0 < C; c0 += lpc * 4) {
    const int c = c0 + lane * 4;
    const bool on = lane < lpc && c < C;
    float dl = 0.f, dr = 0.f;
    if (on) {
      const float4 zv = *reinterpret_cast<const float4*>(z + r * C + c);
      const float4 a = *reinterpret_cast<const float4*>(al + c), b = *reinterpret_cast<const float4*>(ar + c);
      dl = zv.x * a.x + zv.y * a.y + zv.z * a.z + zv.w * a.w;
      dr = zv.x * b.x + zv.y * b.y + zv.z * b.z + zv.w * b.w;
    }
    for (int o = 1; o < gsz; o <<= 1) {  // segmented tree over the head's lane group
      const float tl = __shfl_down(dl, o), tr = __shfl_down(dr, o);
      if (gl + o < gsz) dl += tl, dr += tr;
    }
    if (on && gl == 0) {
      el[r * H + c / D] = dl;
      er[r * H + c / D] = dr;
    }
  }
}
// backward: g_z[r, h, :] = g_el[r, h] a_l[h, :] + g_er[r, h] a_r[h, :]; per-block partial sums of
// g_a_l[h, :] = sum_r g_el[r, h] z[r, h, :] (and g_a_r) for k_colsum_finish
__global__ __launch_bounds__(BLK) void k_gat_logits_bwd(const float* __restrict__ z, const float* __restrict__ al,
                                                        const float* __restrict__ ar, const float* __restrict__ g_el,
                                                        const float* __restrict__ g_er, long long n, int H, int D,
                                                        float* __restrict__ g_z, float* __restrict__ part_l,
                                                        float* __restrict__ part_r, long long rows_per_block,
                                                        int accumulate) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int C = H * D, gsz = D / 4, lpc = (64 / gsz) * gsz;
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r_end = r0 + rows_per_block < n ? r0 + rows_per_block : n;
  __shared__ float s_l[BLK * 4], s_r[BLK * 4];
  for (int c0 = 0; c0 < C; c0 += lpc * 4) {  // block-uniform
    const int c = c0 + lane * 4;
    const bool on = lane < lpc && c < C;
    const int h = on ? c / D : 0;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a, pl = a, pr = a;
    if (on) {
      a = *reinterpret_cast<const float4*>(al + c);
      b = *reinterpret_cast<const float4*>(ar + c);
    }
    for (long long r = r0 + w; r < r_end; r += BLK / 64) {
      if (!on) continue;
      const float gl_ = g_el[r * H + h], gr_ = g_er[r * H + h];
      const float4 zv = *reinterpret_cast<const float4*>(z + r * C + c);
      float4 o;
      o.x = gl_ * a.x + gr_ * b.x, o.y = gl_ * a.y + gr_ * b.y, o.z = gl_ * a.z + gr_ * b.z, o.w = gl_ * a.w + gr_ * b.w;
      if (accumulate) add4(o, *reinterpret_cast<const float4*>(g_z + r * C + c));  // on top of the aggregation's share
      *reinterpret_cast<float4*>(g_z + r * C + c) = o;
      pl.x += gl_ * zv.x, pl.y += gl_ * zv.y, pl.z += gl_ * zv.z, pl.w += gl_ * zv.w;
      pr.x += gr_ * zv.x, pr.y += gr_ * zv.y, pr.z += gr_ * zv.z, pr.w += gr_ * zv.w;
    }
    __syncthreads();
    reinterpret_cast<float4*>(s_l)[threadIdx.x] = pl;
    reinterpret_cast<float4*>(s_r)[threadIdx.x] = pr;
    __syncthreads();
    if (w == 0 && on) {
      for (int k = 1; k < BLK / 64; k++) {
        add4(pl, reinterpret_cast<float4*>(s_l)[k * 64 + lane]);
        add4(pr, reinterpret_cast<float4*>(s_r)[k * 64 + lane]);
      }
      *reinterpret_cast<float4*>(part_l + (long long)blockIdx.x * C + c) = pl;
      *reinterpret_cast<float4*>(part_r + (long long)blockIdx.x * C + c) = pr;
    }
  }
}

// ---- GAT layer epilogue: out = act(n / s + bias) from the aggregation's (s, n) (DistGATConv: out[v] = sum_u alpha z[u]
// + bias; ELU between layers), and its backward -- with torch ops this was a division, an add, an ELU and their
// autograd nodes with broadcasts and reductions (a dozen launches per layer).  One wave per row, a lane owns 4
// consecutive columns, heads are groups of D / 4 lanes (the layout of k_gat_logits_*).
__global__ __launch_bounds__(BLK) void k_gat_finish_fwd(const float* __restrict__ nsum, const float* __restrict__ ssum,
                                                        const float* __restrict__ bias, long long n, int H, int D,
                                                        int elu, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long long r = (long long)blockIdx.x * (BLK / 64) + (threadIdx.x >> 6);
  if (r >= n) return;
  const int C = H * D, gsz = D / 4, lpc = (64 / gsz) * gsz;
  for (int c0 = 0; c0 < C; c0 += lpc * 4) {
    const int c = c0 + lane * 4;
    if (lane >= lpc || c >= C) continue;
    const float inv = 1.f / fmaxf(ssum[r * H + c / D], 1e-30f);
    float4 v = *reinterpret_cast<const float4*>(nsum + r * C + c);
    const float4 b = *reinterpret_cast<const float4*>(bias + c);
    v.x = v.x * inv + b.x, v.y = v.y * inv + b.y, v.z = v.z * inv + b.z, v.w = v.w * inv + b.w;
    if (elu) v.x = v.x > 0.f ? v.x : expm1f(v.x), v.y = v.y > 0.f ? v.y : expm1f(v.y), v.z = v.z > 0.f ? v.z : expm1f(v.z), v.w = v.w > 0.f ? v.w : expm1f(v.w);
    *reinterpret_cast<float4*>(out + r * C + c) = v;
  }
}
// backward: p = g .* act'(out) (ELU: out > 0 ? 1 : out + 1);  g_n = p / s;  g_s[r, h] = -sum_d g_n n / s;
// per-block column sums of p (the bias gradient's first stage) to part[block][C]
__global__ __launch_bounds__(BLK) void k_gat_finish_bwd(const float* __restrict__ g, long long ldg,
                                                        const float* __restrict__ out, const float* __restrict__ nsum,
                                                        const float* __restrict__ ssum, long long n, int H, int D, int elu,
                                                        float* __restrict__ g_n, float* __restrict__ g_s,
                                                        float* __restrict__ part, long long rows_per_block) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int C = H * D, gsz = D / 4, lpc = (64 / gsz) * gsz, gl = lane % gsz;
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r_end = r0 + rows_per_block < n ? r0 + rows_per_block : n;
  __shared__ float s_p[BLK * 4];
  for (int c0 = 0; c0 < C; c0 += lpc * 4) {  // block-uniform
    const int c = c0 + lane * 4;
    const bool on = lane < lpc && c < C;
    const int h = on ? c / D : 0;
    float4 colacc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long long r = r0 + w; r < r_end; r += BLK / 64) {  // wave-uniform
      float dot = 0.f;
      if (on) {
        float4 p = *reinterpret_cast<const float4*>(g + r * ldg + c);
        if (elu) {
          const float4 o = *reinterpret_cast<const float4*>(out + r * C + c);
          p.x *= o.x > 0.f ? 1.f : o.x + 1.f, p.y *= o.y > 0.f ? 1.f : o.y + 1.f;
          p.z *= o.z > 0.f ? 1.f : o.z + 1.f, p.w *= o.w > 0.f ? 1.f : o.w + 1.f;
        }
        add4(colacc, p);
        const float inv = 1.f / fmaxf(ssum[r * H + h], 1e-30f);
        const float4 nv = *reinterpret_cast<const float4*>(nsum + r * C + c);
        p.x *= inv, p.y *= inv, p.z *= inv, p.w *= inv;
        *reinterpret_cast<float4*>(g_n + r * C + c) = p;
        dot = -(p.x * nv.x + p.y * nv.y + p.z * nv.z + p.w * nv.w) * inv;
      }
      for (int o = 1; o < gsz; o <<= 1) {  // segmented tree over the head's lane group
        const float t = __shfl_down(dot, o);
        if (gl + o < gsz) dot += t;
      }
      if (on && gl == 0) g_s[r * H + h] = dot;
    }
    __syncthreads();
    reinterpret_cast<float4*>(s_p)[threadIdx.x] = colacc;
    __syncthreads();
    if (w == 0 && on) {
      for (int k = 1; k < BLK / 64; k++) add4(colacc, reinterpret_cast<float4*>(s_p)[k * 64 + lane]);
      *reinterpret_cast<float4*>(part + (long long)blockIdx.x * C + c) = colacc;
    }
  }
}

int group_for(int H) {
  int q = (H + 3) / 4, g = 1;
  while (g < q && g < 64) g <<= 1;
  return g;
}
bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
int vec_ok(const void* a, long long lda, const void* b, long long ldb, int H) {
  return (H % 4 == 0) && (lda % 4 == 0) && (ldb % 4 == 0) && aligned16(a) && aligned16(b);
}

#define DISPATCH_G(G, KERNEL, grid_rows, ...)                                                        \
  switch (G) {                                                                                       \
    case 1: hipLaunchKernelGGL(KERNEL<1>, dim3((unsigned)(((grid_rows) + BLK - 1) / BLK)), dim3(BLK), 0, st, __VA_ARGS__); break; \
    case 2: hipLaunchKernelGGL(KERNEL<2>, dim3((unsigned)(((grid_rows) + BLK / 2 - 1) / (BLK / 2))), dim3(BLK), 0, st, __VA_ARGS__); break; \
    case 4: hipLaunchKernelGGL(KERNEL<4>, dim3((unsigned)(((grid_rows) + BLK / 4 - 1) / (BLK / 4))), dim3(BLK), 0, st, __VA_ARGS__); break; \
    case 8: hipLaunchKernelGGL(KERNEL<8>, dim3((unsigned)(((grid_rows) + BLK / 8 - 1) / (BLK / 8))), dim3(BLK), 0, st, __VA_ARGS__); break; \
    case 16: hipLaunchKernelGGL(KERNEL<16>, dim3((unsigned)(((grid_rows) + BLK / 16 - 1) / (BLK / 16))), dim3(BLK), 0, st, __VA_ARGS__); break; \
    case 32: hipLaunchKernelGGL(KERNEL<32>, dim3((unsigned)(((grid_rows) + BLK / 32 - 1) / (BLK / 32))), dim3(BLK), 0, st, __VA_ARGS__); break; \
    default: hipLaunchKernelGGL(KERNEL<64>, dim3((unsigned)(((grid_rows) + BLK / 64 - 1) / (BLK / 64))), dim3(BLK), 0, st, __VA_ARGS__); break; \
  }

int done() { return hipGetLastError() == hipSuccess ? CSL_OK : CSL_E_HIP; }

}  // namespace

extern "C" {

static int spmm_sum_impl(const int32_t* indptr, const int32_t* indices, const int32_t* rows, int64_t n_rows,
                         const float* x, int64_t ldx, float* out, int64_t ldo, int32_t H, int compact, void* stream,
                         const int32_t* rowmap = nullptr) {
  if (n_rows == 0) return CSL_OK;  // nothing to do: empty lists come with null pointers
  if (n_rows < 0 || H < 1 || !indptr || !out || ldx < H || ldo < H || (compact && !rows)) return CSL_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const int G = group_for(H), v = vec_ok(x, ldx, out, ldo, H);
  DISPATCH_G(G, k_spmm_sum, n_rows, indptr, indices, rows,
             (long long)n_rows, x, (long long)ldx, out, (long long)ldo, (int)H, v, compact, rowmap);
  return done();
}

int csl_spmm_sum_map_f32(const int32_t* indptr, const int32_t* indices, const int32_t* rows, int64_t n_rows, const float* x,
                         int64_t ldx, const int32_t* rowmap, float* out, int64_t ldo, int32_t H, int32_t compact, void* stream) {
  return spmm_sum_impl(indptr, indices, rows, n_rows, x, ldx, out, ldo, H, compact ? 1 : 0, stream, rowmap);
}

int csl_spmm_sum_f32(const int32_t* indptr, const int32_t* indices, const int32_t* rows, int64_t n_rows,
                     const float* x, int64_t ldx, float* out, int64_t ldo, int32_t H, void* stream) {
  return spmm_sum_impl(indptr, indices, rows, n_rows, x, ldx, out, ldo, H, 0, stream);
}

int csl_spmm_sum_compact_f32(const int32_t* indptr, const int32_t* indices, const int32_t* rows, int64_t n_rows,
                             const float* x, int64_t ldx, float* out, int64_t ldo, int32_t H, void* stream) {
  return spmm_sum_impl(indptr, indices, rows, n_rows, x, ldx, out, ldo, H, 1, stream);
}

// gradient of csl_sage_cat_f32's merged-sums form: gx [n_x, H] and gagg [n_agg, H] are zeroed here, then
// gx[self_ids[r]] = gcat[r, 0:H) and gagg[owned[r]] = gcat[r, H:2H) / max(deg[r], 1) (both index lists are unique)
int csl_gat_bwd_t_f32(const int32_t* t_indptr, const int32_t* t_indices, int64_t n_src, int64_t n_pad, const float* el,
                      const float* er, const float* z, int32_t H, int32_t D, float slope, const float* m_in,
                      const float* g_s, const float* g_n, float* g_el, float* g_er, float* g_z, void* stream) {
  if (n_src < 0 || n_pad < n_src || H < 1 || D < 4 || D % 4 != 0 || D > 256) return CSL_E_INVALID;
  if (n_pad == 0) return CSL_OK;
  if (!g_z || !aligned16(g_z) || (n_src > 0 && (!t_indptr || !el || !z || !g_el || !aligned16(z)))) return CSL_E_INVALID;
  if (n_src > 0 && (!er || !m_in || !g_s || !g_n || !g_er || !aligned16(g_n))) return CSL_E_INVALID;
  hipLaunchKernelGGL(k_gat_bwd_t, dim3((unsigned)((n_pad + BLK / 64 - 1) / (BLK / 64))), dim3(BLK), 0, (hipStream_t)stream,
                     t_indptr, t_indices, (long long)n_src, (long long)n_pad, el, er, z, (int)H, (int)D, slope, m_in, g_s,
                     g_n, g_el, g_er, g_z);
  return done();
}

static long long gat_t_rows(long long n) {   // source rows per workgroup of k_gat_bwd_t2: at most ~2048 workgroups
  long long r = 16;                        // (each leaves a row of partial sums for the second stage to read)
  // (CSL_GAT_T_BLOCKS: measurement knob; 1024 / 2048 / 4096 workgroups: 32.7 / 26.5 / 26.9 us for the 62 k-row layer, the
  // second stage 9.5 / 11.2 / 14.6)
  static const long long cap = getenv("CSL_GAT_T_BLOCKS") ? atoll(getenv("CSL_GAT_T_BLOCKS")) : 2048;
  while ((n + r - 1) / r > cap) r *= 2;
  return r;
}

int64_t csl_gat_bwd_t_fused_scratch(int64_t n_pad, int64_t n_out, int32_t H, int32_t D) {
  if (n_pad < 0 || n_out < 0 || H < 1 || D < 4 || D % 4 != 0 || D > 256) return CSL_E_INVALID;
  const long long r1 = gat_t_rows(n_pad), r2 = gat_t_rows(n_out);
  return ((n_pad + r1 - 1) / r1 + (n_out + r2 - 1) / r2) * (int64_t)H * D;
}

int csl_gat_bwd_t_fused_f32(const int32_t* t_indptr, const int32_t* t_indices, int64_t n_src, int64_t n_pad, const float* el,
                            const float* er_out, const float* z, int32_t H, int32_t D, float slope, const float* m_in,
                            const float* g_s, const float* g_n, const float* attn_l, const float* attn_r,
                            const int32_t* self_ids_in, int64_t n_out, float* g_er_out, float* g_z, float* g_attn_l,
                            float* g_attn_r, float* scratch, void* stream) {
  if (n_src < 0 || n_pad < n_src || n_out < 0 || H < 1 || D < 4 || D % 4 != 0 || D > 256 || !g_attn_l || !g_attn_r)
    return CSL_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const int C = H * D;
  const long long r1 = gat_t_rows(n_pad), r2 = gat_t_rows(n_out);
  const long long b1 = (n_pad + r1 - 1) / r1, b2 = (n_out + r2 - 1) / r2;
  if (b1 > 0) {
    if (!g_z || !aligned16(g_z) || !scratch || !aligned16(scratch) || !attn_l || !aligned16(attn_l)) return CSL_E_INVALID;
    if (n_src > 0 && (!t_indptr || !el || !z || !aligned16(z) || !er_out || !m_in || !g_s || !g_n || !aligned16(g_n) || !g_er_out))
      return CSL_E_INVALID;
    if (n_out > 0 && hipMemsetAsync(g_er_out, 0, sizeof(float) * (size_t)n_out * H, st) != hipSuccess) return CSL_E_HIP;
    hipLaunchKernelGGL(k_gat_bwd_t2, dim3((unsigned)b1), dim3(BLK), 0, st, t_indptr, t_indices, (long long)n_src,
                       (long long)n_pad, el, er_out, z, (int)H, (int)D, slope, m_in, g_s, g_n, attn_l, g_er_out, g_z, scratch, r1);
  }
  if (b2 > 0) {
    if (!attn_r || !aligned16(attn_r) || !self_ids_in) return CSL_E_INVALID;
    hipLaunchKernelGGL(k_gat_logits_bwd_dst, dim3((unsigned)b2), dim3(BLK), 0, st, z, attn_r, self_ids_in, g_er_out,
                       (long long)n_out, (int)H, (int)D, g_z, scratch + b1 * C, r2);
  }
  {
    // both second stages in one launch
    const float* src[2] = {scratch, scratch + b1 * C};
    const int64_t nb[2] = {b1, b2};
    const int32_t hh[2] = {C, C};
    float* dst[2] = {g_attn_l, g_attn_r};
    return csl_reduce_multi_f32(2, src, nb, hh, dst, stream);
  }
}

int csl_gat_finish_fwd_f32(const float* n_in, const float* s_in, const float* bias, int64_t n, int32_t H, int32_t D,
                           int32_t elu, float* out, void* stream) {
  if (n < 0 || H < 1 || D < 4 || D % 4 != 0 || D > 256) return CSL_E_INVALID;
  if (n == 0) return CSL_OK;
  if (!n_in || !s_in || !bias || !out || !aligned16(n_in) || !aligned16(bias) || !aligned16(out)) return CSL_E_INVALID;
  hipLaunchKernelGGL(k_gat_finish_fwd, dim3((unsigned)((n + BLK / 64 - 1) / (BLK / 64))), dim3(BLK), 0, (hipStream_t)stream,
                     n_in, s_in, bias, (long long)n, (int)H, (int)D, (int)(elu != 0), out);
  return done();
}

// rows per workgroup of k_gat_finish_bwd: a wave walks its rows one after the other (a dependent shuffle chain per row), so a
// small layer wants MANY small workgroups: with rb_rows' 128 the 8 k-row and 1 k-row layers of config 5's model ran on 64
// and 8 workgroups, 31-41 us each whatever their size (profiles/r3_gat_in)
static long long gat_finish_rows(long long n) {
  long long r = (n + 1023) / 1024;
  r = (r + 3) / 4 * 4;
  return r < 4 ? 4 : (r > 128 ? rb_rows(n) : r);
}

int64_t csl_gat_finish_bwd_scratch(int64_t n, int32_t H, int32_t D) {
  const long long rpb = gat_finish_rows(n);
  return ((n + rpb - 1) / rpb) * (int64_t)H * D;
}

int csl_gat_finish_bwd_f32(const float* g, int64_t ldg, const float* out, const float* n_in, const float* s_in, int64_t n,
                           int32_t H, int32_t D, int32_t elu, float* g_n, float* g_s, float* g_bias, float* scratch,
                           void* stream) {
  if (n < 0 || H < 1 || D < 4 || D % 4 != 0 || D > 256 || !g_bias) return CSL_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const int C = H * D;
  const long long rpb = gat_finish_rows(n);
  const long long blocks = (n + rpb - 1) / rpb;
  if (blocks > 0) {
    if (!g || ldg < C || ldg % 4 != 0 || !n_in || !s_in || !g_n || !g_s || !scratch || (elu && !out) || !aligned16(g) ||
        !aligned16(n_in) || !aligned16(g_n) || !aligned16(scratch) || (out && !aligned16(out)))
      return CSL_E_INVALID;
    hipLaunchKernelGGL(k_gat_finish_bwd, dim3((unsigned)blocks), dim3(BLK), 0, st, g, (long long)ldg, out, n_in, s_in,
                       (long long)n, (int)H, (int)D, (int)(elu != 0), g_n, g_s, scratch, rpb);
  }
  hipLaunchKernelGGL(k_colsum_finish, dim3((unsigned)((C + 63) / 64)), dim3(BLK), 0, st, scratch, blocks, C, g_bias);
  return done();
}

int csl_sage_cat_rows_bwd_f32(const int32_t* self_ids, const int32_t* owned, const int32_t* deg, int64_t n,
                              const float* gcat, int64_t ldg, float* gx, int64_t n_x, float* gagg, int64_t n_agg,
                              int32_t H, void* stream);

int csl_spmm_sum_bwd_f32(const int32_t* indptr, const int32_t* indices, const int32_t* rows, int64_t n_rows,
                         const float* grad_out, int64_t ldg, int32_t compact, float* grad_x, int64_t ldx, int32_t H,
                         void* stream) {
  if (n_rows == 0) return CSL_OK;
  if (n_rows < 0 || H < 1 || !indptr || !grad_out || !grad_x) return CSL_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_spmm_sum_bwd, dim3((unsigned)((n_rows + BLK / 64 - 1) / (BLK / 64))), dim3(BLK), 0, st,
                     indptr, indices, rows, (long long)n_rows,
                     grad_out, (long long)ldg, (int)compact, grad_x, (long long)ldx, (int)H);
  return done();
}

int csl_gather_rows_f32(const float* src, int64_t lds, const int32_t* idx, int64_t n, float* dst, int64_t ldd,
                        int32_t H, void* stream) {
  if (n == 0) return CSL_OK;
  if (n < 0 || H < 1 || !idx || !dst) return CSL_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const int G = group_for(H), v = vec_ok(src, lds, dst, ldd, H);
  DISPATCH_G(G, k_gather_rows, n, src, (long long)lds, idx, (long long)n, dst, (long long)ldd, (int)H, v);
  return done();
}

int csl_scatter_add_rows_f32(float* dst, int64_t ldd, const int32_t* idx, int64_t n, const float* src, int64_t lds,
                             int32_t H, void* stream) {
  if (n == 0) return CSL_OK;
  if (n < 0 || H < 1 || !idx || !dst || !src) return CSL_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const int G = group_for(H), v = vec_ok(src, lds, dst, ldd, H);
  DISPATCH_G(G, k_scatter_add_rows, n, dst, (long long)ldd, idx, (long long)n, src, (long long)lds,
             (int)H, v);
  return done();
}

int csl_scatter_add_rows_atomic_f32(float* dst, int64_t ldd, const int32_t* idx, int64_t n, const float* src, int64_t lds,
                                    int32_t H, void* stream) {
  if (n == 0) return CSL_OK;
  if (n < 0 || H < 1 || !idx || !dst || !src) return CSL_E_INVALID;
  hipLaunchKernelGGL(k_scatter_add_rows_atomic, dim3((unsigned)((n + BLK / 64 - 1) / (BLK / 64))), dim3(BLK), 0,
                     (hipStream_t)stream, dst, (long long)ldd, idx, (long long)n, src, (long long)lds, (int)H);
  return done();
}

int csl_div_rows_f32(float* x, int64_t ldx, const int32_t* deg, int64_t n, int32_t H, void* stream) {
  if (n == 0) return CSL_OK;
  if (n < 0 || H < 1 || !x || !deg) return CSL_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const int G = group_for(H), v = vec_ok(x, ldx, x, ldx, H);
  DISPATCH_G(G, k_div_rows, n, x, (long long)ldx, deg, (long long)n, (int)H, v);
  return done();
}

static int gat_args_ok(const void* a, const void* b, const void* c, int32_t H, int32_t D) {
  if (H < 1 || D < 4 || D % 4 != 0 || D > 256) return 0;  // a head is at most one wave of float4 lanes
  return aligned16(a) && aligned16(b) && aligned16(c);
}

int csl_gat_fwd_f32(const int32_t* indptr, const int32_t* indices, int64_t n_rows, const float* el, const float* er,
                    const float* z, int32_t H, int32_t D, float slope, float* m_out, float* s_out, float* n_out,
                    void* stream) {
  if (n_rows == 0) return CSL_OK;
  if (n_rows < 0 || !indptr || !er || !m_out || !s_out || !n_out || !gat_args_ok(z, n_out, n_out, H, D))
    return CSL_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_gat_fwd, dim3((unsigned)((n_rows + BLK / 64 - 1) / (BLK / 64))), dim3(BLK), 0, st, indptr,
                     indices, (long long)n_rows, el, er, z, (int)H, (int)D, slope, m_out, s_out, n_out);
  return done();
}

int csl_gat_bwd_f32(const int32_t* indptr, const int32_t* indices, int64_t n_rows, const float* el, const float* er,
                    const float* z, int32_t H, int32_t D, float slope, const float* m_in, const float* g_s,
                    const float* g_n, float* g_el, float* g_er, float* g_z, void* stream) {
  if (n_rows == 0) return CSL_OK;
  if (n_rows < 0 || !indptr || !er || !m_in || !g_s || !g_n || !g_er || !gat_args_ok(z, g_n, g_z, H, D))
    return CSL_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_gat_bwd, dim3((unsigned)((n_rows + BLK / 64 - 1) / (BLK / 64))), dim3(BLK), 0, st, indptr,
                     indices, (long long)n_rows, el, er, z, (int)H, (int)D, slope, m_in, g_s, g_n, g_el, g_er, g_z);
  return done();
}


int csl_sage_cat_f32(const int32_t* indptr, const int32_t* indices, const int32_t* self_ids, const int32_t* owned,
                     const int32_t* deg, const int32_t* rowmap, const float* x, int64_t ldx, const float* agg,
                     int64_t lda, int64_t n, int64_t n_pad, float* cat, int64_t ldc, int32_t H, int32_t relu_in,
                     void* stream) {
  if (n_pad == 0) return CSL_OK;
  if (n < 0 || n_pad < n || H < 4 || H % 4 != 0 || !cat || ldc < 2 * (int64_t)H || ldc % 4 != 0 || !aligned16(cat))
    return CSL_E_INVALID;
  if (n > 0) {
    if (!self_ids || !x || ldx < H || ldx % 4 != 0 || !aligned16(x)) return CSL_E_INVALID;
    if (!indptr && (!owned || !deg || !agg || lda < H || lda % 4 != 0 || !aligned16(agg))) return CSL_E_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  const int G = group_for(H);
  DISPATCH_G(G, k_sage_cat, n_pad, indptr, indices, self_ids, owned, deg, rowmap, x, (long long)ldx, agg, (long long)lda,
             (long long)n, (long long)n_pad, cat, (long long)ldc, (int)H, (int)relu_in);
  return done();
}

int csl_sage_cat_bwd_f32(const int32_t* indptr, const int32_t* indices, const int32_t* self_ids, int64_t n,
                         const float* gcat, int64_t ldg, float* gx, int64_t ldx, int64_t n_src, int32_t H,
                         void* stream) {
  if (n < 0 || n_src < 0 || H < 1 || !gx || ldx < H) return CSL_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  if (n_src > 0 && hipMemsetAsync(gx, 0, sizeof(float) * (size_t)n_src * (size_t)ldx, st) != hipSuccess) return CSL_E_HIP;
  if (n == 0) return CSL_OK;
  if (!indptr || !self_ids || !gcat || ldg < 2 * (int64_t)H) return CSL_E_INVALID;
  hipLaunchKernelGGL(k_sage_cat_bwd, dim3((unsigned)((n + BLK / 64 - 1) / (BLK / 64))), dim3(BLK), 0, st, indptr, indices,
                     self_ids, (long long)n, gcat, (long long)ldg, gx, (long long)ldx, (int)H);
  return done();
}

int64_t csl_relu_bwd_colsum_scratch(int64_t n_pad, int32_t H) {
  const long long rpb = rb_rows(n_pad);
  return ((n_pad + rpb - 1) / rpb) * (int64_t)(H > 0 ? H : 0);
}

int64_t csl_sage_cat_bwd_t_scratch(int64_t n_pad, int32_t H) {
  const long long rpb = tb_rows(n_pad);
  return ((n_pad + rpb - 1) / rpb) * (int64_t)(H > 0 ? H : 0);
}

int csl_sage_cat_bwd_t_f32(const int32_t* t_indptr, const int32_t* t_indices, const int32_t* indptr, const float* gcat,
                           int64_t ldg, const float* y, int64_t ldy, int64_t n_src, int64_t n_pad, float* out,
                           int64_t ldo, float* colsum, float* scratch, int32_t H, void* stream) {
  if (n_src < 0 || n_pad < n_src || H < 4 || H % 4 != 0) return CSL_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const long long rpb = tb_rows(n_pad);
  const long long blocks = (n_pad + rpb - 1) / rpb;
  if (blocks > 0) {
    if (!out || !scratch || ldo < H || ldo % 4 != 0 || !aligned16(out)) return CSL_E_INVALID;
    if (n_src > 0 && (!t_indptr || !t_indices || !gcat || ldg < 2 * (int64_t)H || ldg % 4 != 0 || !aligned16(gcat)))
      return CSL_E_INVALID;
    if (y && (ldy < H || ldy % 4 != 0 || !aligned16(y))) return CSL_E_INVALID;
#define LAUNCH_BWD_T(G)                                                                                           \
  hipLaunchKernelGGL(k_sage_cat_bwd_t<G>, dim3((unsigned)blocks), dim3(BLK), 0, st, t_indptr, t_indices, indptr, gcat, \
                     (long long)ldg, y, (long long)ldy, (long long)n_src, (long long)n_pad, out, (long long)ldo, scratch, \
                     (int)H, rpb, 0)
    if (H > 128) LAUNCH_BWD_T(64);
    else if (H > 64) LAUNCH_BWD_T(32);
    else LAUNCH_BWD_T(16);
#undef LAUNCH_BWD_T
  }
  // colsum == NULL: the per-block sums stay in scratch[blocks][H] for the caller's own second stage (csl_reduce_multi_f32)
  if (colsum) hipLaunchKernelGGL(k_colsum_finish, dim3((unsigned)((H + 63) / 64)), dim3(BLK), 0, st, scratch, blocks, (int)H, colsum);
  return done();
}

int csl_sage_rank_g2_f32(const int32_t* owned, const int32_t* deg, int64_t n_owned, const float* gcat, int64_t ldg, float* g2,
                         int64_t ld2, int32_t H, void* stream) {
  if (n_owned == 0) return CSL_OK;
  if (n_owned < 0 || H < 4 || H % 4 != 0 || !owned || !deg || !gcat || !g2 || ldg < 2 * (int64_t)H || ld2 < 2 * (int64_t)H ||
      ldg % 4 != 0 || ld2 % 4 != 0 || !aligned16(gcat) || !aligned16(g2))
    return CSL_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const int G = group_for(H);
  DISPATCH_G(G, k_rank_g2, n_owned, owned, deg, (long long)n_owned, gcat, (long long)ldg, g2, (long long)ld2, (int)H);
  return done();
}

int csl_scatter_rows_f32(float* dst, int64_t ldd, const int32_t* idx, int64_t n, const float* src, int64_t lds, int32_t H,
                         void* stream) {
  if (n == 0) return CSL_OK;
  if (n < 0 || H < 4 || H % 4 != 0 || !dst || !idx || !src || ldd < H || lds < H || ldd % 4 != 0 || lds % 4 != 0 ||
      !aligned16(dst) || !aligned16(src))
    return CSL_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const int G = group_for(H);
  DISPATCH_G(G, k_scatter_rows_set, n, dst, (long long)ldd, idx, (long long)n, src, (long long)lds, (int)H);
  return done();
}

int64_t csl_sage_cat_bwd_t_hub_scratch(int64_t n_pad, int32_t H) {
  if (n_pad < 0 || H < 4 || H % 4 != 0) return CSL_E_INVALID;
  const long long rpb = tb_rows(n_pad);
  return 2 * ((n_pad + rpb - 1) / rpb) * H;   // the ordinary rows' blocks, then the hub rows'
}

int csl_sage_cat_bwd_t_hub_f32(const int32_t* t_indptr, const int32_t* t_indices, int64_t t_entries, const int32_t* indptr,
                               const float* gcat, int64_t ldg, const float* y, int64_t ldy, int64_t n_src, int64_t n_pad,
                               float* out, int64_t ldo, float* colsum, float* scratch, int32_t H, void* stream) {
  if (n_src < 0 || n_pad < n_src || H < 4 || H % 4 != 0 || t_entries < 0) return CSL_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const long long rpb = tb_rows(n_pad);
  const long long blocks = (n_pad + rpb - 1) / rpb;
  if (blocks > 0) {
    if (!out || !scratch || ldo < H || ldo % 4 != 0 || !aligned16(out)) return CSL_E_INVALID;
    if (n_src > 0 && (!t_indptr || !t_indices || !gcat || ldg < 2 * (int64_t)H || ldg % 4 != 0 || !aligned16(gcat)))
      return CSL_E_INVALID;
    if (y && (ldy < H || ldy % 4 != 0 || !aligned16(y))) return CSL_E_INVALID;
    const int thr = CSL_T_SORTED_MAX;
    const long long segs = (t_entries + HUB_SEG - 1) / HUB_SEG;
    float* part2 = scratch + blocks * H;
#define LAUNCH_HUB(G)                                                                                                  \
  do {                                                                                                                 \
    hipLaunchKernelGGL(k_sage_cat_bwd_t<G>, dim3((unsigned)blocks), dim3(BLK), 0, st, t_indptr, t_indices, indptr, gcat, \
                       (long long)ldg, y, (long long)ldy, (long long)n_src, (long long)n_pad, out, (long long)ldo,     \
                       scratch, (int)H, rpb, thr);                                                                     \
    if (segs > 0 && n_src > 0)                                                                                         \
      hipLaunchKernelGGL(k_sage_cat_bwd_t_hub<G>, dim3((unsigned)segs), dim3(BLK), 0, st, t_indptr, t_indices, indptr,  \
                         gcat, (long long)ldg, (long long)n_src, out, (long long)ldo, (int)H, thr);                    \
    hipLaunchKernelGGL(k_sage_cat_bwd_t_hubfin<G>, dim3((unsigned)blocks), dim3(BLK), 0, st, t_indptr, y, (long long)ldy, \
                       (long long)n_src, out, (long long)ldo, part2, (int)H, rpb, thr);                                \
  } while (0)
    if (H > 128) LAUNCH_HUB(64);
    else if (H > 64) LAUNCH_HUB(32);
    else LAUNCH_HUB(16);
#undef LAUNCH_HUB
  }
  if (colsum) hipLaunchKernelGGL(k_colsum_finish, dim3((unsigned)((H + 63) / 64)), dim3(BLK), 0, st, scratch, 2 * blocks, (int)H, colsum);
  return done();
}

int csl_relu_bwd_colsum_f32(const float* g, int64_t ldg, const float* y, int64_t ldy, int64_t n, int64_t n_pad,
                            float* out, int64_t ldo, float* colsum, float* scratch, int32_t H, void* stream) {
  if (n < 0 || n_pad < n || H < 1) return CSL_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const long long rpb = rb_rows(n_pad);
  const long long blocks = (n_pad + rpb - 1) / rpb;
  if (blocks > 0) {
    if (!scratch || !out || ldo < H || (n > 0 && (!g || ldg < H)) || (y && ldy < H)) return CSL_E_INVALID;
    const int G = group_for(H);
    const int vec = (H % 4 == 0) && (ldg % 4 == 0) && (ldo % 4 == 0) && (!y || ldy % 4 == 0) && aligned16(g) &&
                    aligned16(out) && (!y || aligned16(y));
#define LAUNCH_RBC(GG)                                                                                         \
  hipLaunchKernelGGL(k_relu_bwd_colsum<GG>, dim3((unsigned)blocks), dim3(BLK), 0, st, g, (long long)ldg, y,    \
                     (long long)ldy, (long long)n, (long long)n_pad, out, (long long)ldo, scratch, (int)H, vec, rpb)
    switch (G) {
      case 1: case 2: case 4: LAUNCH_RBC(4); break;
      case 8: LAUNCH_RBC(8); break;
      case 16: LAUNCH_RBC(16); break;
      case 32: LAUNCH_RBC(32); break;
      default: LAUNCH_RBC(64); break;
    }
#undef LAUNCH_RBC
  }
  // colsum == NULL: the per-block sums stay in scratch[blocks][H] for the caller's own second stage
  if (colsum) hipLaunchKernelGGL(k_colsum_finish, dim3((unsigned)((H + 63) / 64)), dim3(BLK), 0, st, scratch, blocks, (int)H, colsum);
  return done();
}

int64_t csl_softmax_ce_scratch(int64_t n) { return (n + BLK / 64 - 1) / (BLK / 64); }

int csl_softmax_ce_f32(const float* logits, int64_t ldl, int64_t n, int32_t C, const int32_t* ids, const int32_t* rowmap,
                       const int64_t* labels, float scale, float* loss, float* grad, int64_t ldgr, float* scratch,
                       void* stream) {
  if (n < 0 || C < 1 || !loss) return CSL_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const long long blocks = (n + BLK / 64 - 1) / (BLK / 64);
  if (blocks > 0) {
    if (!logits || !ids || !labels || !grad || !scratch || ldl < C || ldgr < C) return CSL_E_INVALID;
    hipLaunchKernelGGL(k_softmax_ce, dim3((unsigned)blocks), dim3(BLK), 0, st, logits, (long long)ldl, (long long)n,
                       (long long)n, (int)C, ids, rowmap, (const long long*)labels, scale, scratch, grad, (long long)ldgr,
                       (float*)nullptr);
  }
  hipLaunchKernelGGL(k_colsum_finish, dim3(1), dim3(BLK), 0, st, scratch, blocks, 1, loss);
  return done();
}

int csl_softmax_ce_partial_f32(const float* logits, int64_t ldl, int64_t n, int64_t n_pad, int32_t C, const int32_t* ids,
                               const int32_t* rowmap, const int64_t* labels, float scale, float* grad, int64_t ldgr,
                               float* loss_partial, float* col_partial, void* stream) {
  if (n < 0 || n_pad < n || C < 1 || (col_partial && C > SM_CMAX)) return CSL_E_INVALID;
  const long long blocks = (n_pad + BLK / 64 - 1) / (BLK / 64);
  if (blocks > 0) {
    if (!grad || !loss_partial || ldgr < C || (n > 0 && (!logits || !ids || !labels || ldl < C))) return CSL_E_INVALID;
    hipLaunchKernelGGL(k_softmax_ce, dim3((unsigned)blocks), dim3(BLK), 0, (hipStream_t)stream, logits, (long long)ldl,
                       (long long)n, (long long)n_pad, (int)C, ids, rowmap, (const long long*)labels, scale, loss_partial,
                       grad, (long long)ldgr, col_partial);
  }
  return done();
}

int csl_reduce_multi_f32(int32_t count, const float* const* src, const int64_t* nblk, const int32_t* H, float* const* dst,
                         void* stream) {
  if (count < 0 || count > RM_MAX) return CSL_E_INVALID;
  if (count == 0) return CSL_OK;
  if (!src || !nblk || !H || !dst) return CSL_E_INVALID;
  ReduceJobs jb;
  int at = 0;
  jb.count = 0;
  for (int j = 0; j < count; j++) {
    if (H[j] < 0 || nblk[j] < 0) return CSL_E_INVALID;
    if (H[j] == 0) continue;
    if (!dst[j] || (nblk[j] > 0 && !src[j])) return CSL_E_INVALID;
    const int k = jb.count++;
    jb.src[k] = src[j], jb.dst[k] = dst[j], jb.nblk[k] = nblk[j], jb.H[k] = H[j];
    jb.vec[k] = H[j] % 4 == 0 && aligned16(dst[j]) && (nblk[j] == 0 || aligned16(src[j]));
    jb.first_block[k] = at;
    at += jb.vec[k] ? (H[j] + 4 * RM_L - 1) / (4 * RM_L) : (H[j] + RM_L - 1) / RM_L;
  }
  jb.first_block[jb.count] = at;
  if (at > 0) hipLaunchKernelGGL(k_reduce_multi, dim3((unsigned)at), dim3(BLK), 0, (hipStream_t)stream, jb);
  return done();
}

int csl_adam_f32(int32_t count, float* const* params, const float* const* grads, float* const* exp_avg,
                 float* const* exp_avg_sq, const int64_t* numel, float lr, float beta1, float beta2, float eps,
                 int64_t step, void* stream) {
  if (count < 0 || count > ADAM_MAX || step < 1) return CSL_E_INVALID;
  if (count == 0) return CSL_OK;
  if (!params || !grads || !exp_avg || !exp_avg_sq || !numel) return CSL_E_INVALID;
  AdamArgs a;
  long long blocks = 0;
  for (int t = 0; t < count; t++) {
    if (numel[t] < 0 || (numel[t] > 0 && (!params[t] || !grads[t] || !exp_avg[t] || !exp_avg_sq[t]))) return CSL_E_INVALID;
    a.p[t] = params[t];
    a.g[t] = grads[t];
    a.m[t] = exp_avg[t];
    a.v[t] = exp_avg_sq[t];
    a.n[t] = numel[t];
    a.first_block[t] = blocks;
    blocks += (numel[t] + ADAM_CHUNK - 1) / ADAM_CHUNK;
  }
  a.first_block[count] = blocks;
  a.count = count;
  if (blocks == 0) return CSL_OK;
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(BLK), 0, (hipStream_t)stream, a, beta1, beta2,
                     (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)), eps);
  return done();
}

int csl_gat_logits_fwd_f32(const float* z, const float* attn_l, const float* attn_r, int64_t n, int32_t H, int32_t D,
                           float* el, float* er, void* stream) {
  if (n == 0) return CSL_OK;
  if (n < 0 || !el || !er || !gat_args_ok(z, attn_l, attn_r, H, D)) return CSL_E_INVALID;
  hipLaunchKernelGGL(k_gat_logits_fwd, dim3((unsigned)((n + BLK / 64 - 1) / (BLK / 64))), dim3(BLK), 0, (hipStream_t)stream,
                     z, attn_l, attn_r, (long long)n, (int)H, (int)D, el, er);
  return done();
}

int64_t csl_gat_logits_bwd_scratch(int64_t n, int32_t H, int32_t D) {
  const long long rpb = gat_finish_rows(n);
  return 2 * ((n + rpb - 1) / rpb) * (int64_t)H * D;
}

int csl_gat_logits_bwd_f32(const float* z, const float* attn_l, const float* attn_r, const float* g_el,
                           const float* g_er, int64_t n, int32_t H, int32_t D, float* g_z, float* g_attn_l,
                           float* g_attn_r, float* scratch, void* stream) {
  return csl_gat_logits_bwd_acc_f32(z, attn_l, attn_r, g_el, g_er, n, H, D, g_z, 0, g_attn_l, g_attn_r, scratch, stream);
}

int csl_gat_logits_bwd_acc_f32(const float* z, const float* attn_l, const float* attn_r, const float* g_el,
                               const float* g_er, int64_t n, int32_t H, int32_t D, float* g_z, int32_t accumulate,
                               float* g_attn_l, float* g_attn_r, float* scratch, void* stream) {
  if (n < 0 || !g_attn_l || !g_attn_r || H < 1 || D < 4 || D % 4 != 0 || D > 256) return CSL_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const int C = H * D;
  const long long rpb = gat_finish_rows(n);   // (small layers: many small workgroups, as the epilogue's backward)
  const long long blocks = (n + rpb - 1) / rpb;
  if (blocks > 0) {
    if (!g_el || !g_er || !g_z || !scratch || !gat_args_ok(z, attn_l, attn_r, H, D) || !aligned16(g_z) ||
        !aligned16(scratch))
      return CSL_E_INVALID;
    hipLaunchKernelGGL(k_gat_logits_bwd, dim3((unsigned)blocks), dim3(BLK), 0, st, z, attn_l, attn_r, g_el, g_er,
                       (long long)n, (int)H, (int)D, g_z, scratch, scratch + blocks * C, rpb, (int)(accumulate != 0));
  }
  hipLaunchKernelGGL(k_colsum_finish, dim3((unsigned)((C + 63) / 64)), dim3(BLK), 0, st, scratch, blocks, C, g_attn_l);
  hipLaunchKernelGGL(k_colsum_finish, dim3((unsigned)((C + 63) / 64)), dim3(BLK), 0, st, scratch + blocks * C, blocks, C,
                     g_attn_r);
  return done();
}

int csl_sage_cat_rows_bwd_f32(const int32_t* self_ids, const int32_t* owned, const int32_t* deg, int64_t n,
                              const float* gcat, int64_t ldg, float* gx, int64_t n_x, float* gagg, int64_t n_agg,
                              int32_t H, void* stream) {
  if (n < 0 || n_x < 0 || n_agg < 0 || H < 4 || H % 4 != 0) return CSL_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  if (gx && n_x > 0 && hipMemsetAsync(gx, 0, sizeof(float) * (size_t)n_x * H, st) != hipSuccess) return CSL_E_HIP;
  if (gagg && n_agg > 0 && hipMemsetAsync(gagg, 0, sizeof(float) * (size_t)n_agg * H, st) != hipSuccess) return CSL_E_HIP;
  if (n == 0 || (!gx && !gagg)) return CSL_OK;
  if (!self_ids || !owned || !deg || !gcat || ldg < 2 * (int64_t)H || ldg % 4 != 0 || !aligned16(gcat) ||
      (gx && !aligned16(gx)) || (gagg && !aligned16(gagg)))
    return CSL_E_INVALID;
  const int G = group_for(H);
  DISPATCH_G(G, k_sage_cat_rows_bwd, n, self_ids, owned, deg, (long long)n, gcat, (long long)ldg, gx, gagg, (int)H);
  return done();
}

}  // extern "C"

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

#include <cstdint>

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
                                                  long long ldo, int H, int vec_ok) {
  constexpr int RPB = BLK / G;  // rows per block
  const int lane = threadIdx.x % G;
  const long long r = (long long)blockIdx.x * RPB + threadIdx.x / G;
  if (r >= n_rows) return;
  const long long row = rows ? rows[r] : r;
  const long long e0 = indptr[row], e1 = indptr[row + 1];
  for (int c = lane * 4; c < H; c += G * 4) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    long long e = e0;
    // four source rows in flight per lane
    for (; e + 4 <= e1; e += 4) {
      const long long s0 = indices[e], s1 = indices[e + 1], s2 = indices[e + 2], s3 = indices[e + 3];
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
      const long long s0 = indices[e];
      add4(acc, vec_ok ? *reinterpret_cast<const float4*>(x + s0 * ldx + c) : ld4(x + s0 * ldx, c, H));
    }
    if (vec_ok) {
      *reinterpret_cast<float4*>(out + row * ldo + c) = acc;
    } else {
      st4(out + row * ldo, c, H, acc);
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
      if (on) {
        float* gz = g_z + src * C + c;
        atomicAdd(gz + 0, p * gn.x);
        atomicAdd(gz + 1, p * gn.y);
        atomicAdd(gz + 2, p * gn.z);
        atomicAdd(gz + 3, p * gn.w);
        if (c % D == 0) {
          atomicAdd(g_el + src * H + h, gsc);
          ger += gsc;
        }
      }
    }
    if (on && c % D == 0) g_er[r * H + h] = ger;
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

int csl_spmm_sum_f32(const int32_t* indptr, const int32_t* indices, const int32_t* rows, int64_t n_rows,
                     const float* x, int64_t ldx, float* out, int64_t ldo, int32_t H, void* stream) {
  if (n_rows == 0) return CSL_OK;  // nothing to do: empty lists come with null pointers
  if (n_rows < 0 || H < 1 || !indptr || !out || ldx < H || ldo < H) return CSL_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const int G = group_for(H), v = vec_ok(x, ldx, out, ldo, H);
  DISPATCH_G(G, k_spmm_sum, n_rows, indptr, indices, rows,
             (long long)n_rows, x, (long long)ldx, out, (long long)ldo, (int)H, v);
  return done();
}

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

}  // extern "C"

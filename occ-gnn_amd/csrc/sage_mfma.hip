// sage_mfma.hip -- a GraphSAGE layer's forward as ONE kernel on the fp32 matrix cores (gfx950):
//
//   y[r, :] = act( [ x[map(self_ids[r])] | mean over the CSR row r of x[map(indices[e])] ] . W^T + bias )
//
// i.e. DistSageConv.forward of python/layers/dist_sageconv.py:66-80 (self_gather, gather, mean, concat, Linear(2*in, out))
// on the slice CSR of python/data/bipartite.py:61-67, with the gathered operand staged in LDS instead of HBM.  For the
// deepest layer of the products step (82 k rows x [100 | 100] x 256) the two-kernel form moved 65 MB of operand out to HBM
// and back in (csl_sage_cat_f32 58 us + library GEMM 96 us); here a workgroup gathers 32*MT rows into LDS, multiplies
// them by the whole of W with v_mfma_f32_32x32x2_f32 (exact fp32: a k-ordered fmaf chain) and stores y with bias and
// ReLU applied.  The operand is optionally ALSO written out (cat: the backward's weight-gradient GEMM reads it), which
// costs the store but no read.
//
// Work split.  256 threads = 4 waves.  A wave owns two n-tiles (64 output columns) and all MT m-tiles of the block:
// MT*2 accumulators of 16 registers.  A (the gathered rows) comes from LDS, row-major with a leading dimension of
// 4*odd floats so that the 16-lane groups of a ds_read_b128 down a column of rows touch 64 different banks.  B comes
// straight from L2 into registers, one 16-byte load per lane per n-tile per 8 k: W is re-packed once per call
// (k_pack_w, 200 KB) so that this load is contiguous per wave.  The MFMA sums over k in any order as long as A and B
// agree, so one float4 of a lane feeds FOUR consecutive MFMAs: lanes 0-31 (k index 0 of the instruction) hold
// k = 8q .. 8q+3 of their row / column, lanes 32-63 (k index 1) hold k = 8q+4 .. 8q+7, and MFMA j of the group uses
// element j of both.
//
// Occupancy: MT = 2 at in <= 104 (52 KB of LDS: three workgroups per CU, one gathering while the others multiply),
// MT = 1 above (a 32-row tile).  All of it is plain HIP; the C ABI is in cslicer_aggr.h.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "cslicer_aggr.h"
#include "cslicer_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TB = 256;   // threads per workgroup
constexpr int EC = 8;     // edges of a row requested together

struct FwdArgs {
  const int* indptr;
  const int* indices;
  const int* self_ids;
  const int* rowmap;
  const float* x;
  long long ldx;
  const float4* wp;   // packed W: [K2/8][NTp][64 lanes] float4
  const float* bias;
  float* cat;         // optional [n_pad][ldc]
  long long ldc;
  float* y;           // [n_pad][ldy]
  long long ldy;
  long long n, n_pad;
  int H, out, NTp, relu_in, relu_out, lda;
  int dbg;   // diagnostics (CSLICER_MFMA_DBG): 1 = no gather (LDS left as is), 2 = no multiply
};

// W [out][ldw] -> wp[q][nt][lane].j = W[32 nt + (lane & 31)][8 q + 4 (lane >> 5) + j]   (zero beyond `out`)
__global__ __launch_bounds__(TB) void k_pack_w(const float* __restrict__ W, long long ldw, int out, int KQ, int NTp,
                                               float4* __restrict__ wp) {
  const long long i = (long long)blockIdx.x * TB + threadIdx.x;
  if (i >= (long long)KQ * NTp * 64) return;
  const int lane = (int)(i & 63);
  const long long t = i >> 6;
  const int nt = (int)(t % NTp), q = (int)(t / NTp);
  const int n = 32 * nt + (lane & 31), k = 8 * q + 4 * (lane >> 5);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (n < out) v = *reinterpret_cast<const float4*>(W + (long long)n * ldw + k);
  wp[i] = v;
}

__device__ __forceinline__ float4 relu4(float4 v, float lo) {
  v.x = fmaxf(v.x, lo), v.y = fmaxf(v.y, lo), v.z = fmaxf(v.z, lo), v.w = fmaxf(v.w, lo);
  return v;
}
__device__ __forceinline__ void acc4(float4& a, const float4 b) { a.x += b.x, a.y += b.y, a.z += b.z, a.w += b.w; }

// [self | mean] quads of row r at column c (the summation order of k_sage_cat: edge order, then one multiply)
struct RowRef {
  long long srow;      // feature-table row of the node itself, -1: none
  long long e0, e1;    // CSR range
  float inv;
};

__device__ __forceinline__ RowRef row_ref(const FwdArgs& a, long long r) {
  RowRef o;
  o.srow = -1, o.e0 = o.e1 = 0, o.inv = 1.f;
  if (r < a.n) {
    const long long sid = a.self_ids[r];
    o.srow = sid < 0 ? -1 : (a.rowmap ? (long long)a.rowmap[sid] : sid);
    o.e0 = a.indptr[r];
    o.e1 = a.indptr[r + 1];
    const long long d = o.e1 - o.e0;
    o.inv = 1.0f / (float)(d > 1 ? d : 1);
  }
  return o;
}

template <int MT>
__global__ __launch_bounds__(TB, 3) void k_sage_fwd_mfma(const FwdArgs a) {
  constexpr int BM = 32 * MT;
  extern __shared__ float4 smem4[];
  float* A = reinterpret_cast<float*>(smem4);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long row0 = (long long)blockIdx.x * BM;
  const int H = a.H, lda = a.lda;
  const float lo = a.relu_in ? 0.f : -__builtin_inff();
  const unsigned tl_start = (unsigned)__builtin_amdgcn_s_memrealtime();   // (diagnostics, dbg & 64)

  // ---- gather: 32 lanes per row, 8 rows per pass, two passes in flight (their loads are issued together)
  if (!(a.dbg & 1)) {
    const int g = tid >> 5, gl = tid & 31;
    for (int p = 0; p < BM / 16; p++) {
      const int rl0 = p * 16 + g, rl1 = rl0 + 8;
      const RowRef q0 = row_ref(a, row0 + rl0), q1 = row_ref(a, row0 + rl1);
      for (int c = gl * 4; c < H; c += 128) {
        // first EC edges of both rows: every load independent of the others
        int s0[EC], s1[EC];
#pragma unroll
        for (int u = 0; u < EC; u++) {
          s0[u] = q0.e0 + u < q0.e1 ? a.indices[q0.e0 + u] : -1;
          s1[u] = q1.e0 + u < q1.e1 ? a.indices[q1.e0 + u] : -1;
        }
        if (a.rowmap) {
#pragma unroll
          for (int u = 0; u < EC; u++) {
            if (s0[u] >= 0) s0[u] = a.rowmap[s0[u]];
            if (s1[u] >= 0) s1[u] = a.rowmap[s1[u]];
          }
        }
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 v0[EC], v1[EC];
        float4 sv0 = z4, sv1 = z4;
        if (q0.srow >= 0) sv0 = *reinterpret_cast<const float4*>(a.x + q0.srow * a.ldx + c);
        if (q1.srow >= 0) sv1 = *reinterpret_cast<const float4*>(a.x + q1.srow * a.ldx + c);
#pragma unroll
        for (int u = 0; u < EC; u++) {
          v0[u] = z4, v1[u] = z4;
          if (s0[u] >= 0) v0[u] = *reinterpret_cast<const float4*>(a.x + (long long)s0[u] * a.ldx + c);
          if (s1[u] >= 0) v1[u] = *reinterpret_cast<const float4*>(a.x + (long long)s1[u] * a.ldx + c);
        }
        if (q0.srow >= 0) sv0 = relu4(sv0, lo);
        if (q1.srow >= 0) sv1 = relu4(sv1, lo);
        float4 m0 = z4, m1 = z4;
#pragma unroll
        for (int u = 0; u < EC; u++) {
          if (s0[u] >= 0) acc4(m0, relu4(v0[u], lo));
          if (s1[u] >= 0) acc4(m1, relu4(v1[u], lo));
        }
        // rows with more than EC edges (shallower layers' fanouts)
        for (long long e = q0.e0 + EC; e < q0.e1; e++) {
          long long s = a.indices[e];
          if (a.rowmap) s = a.rowmap[s];
          acc4(m0, relu4(*reinterpret_cast<const float4*>(a.x + s * a.ldx + c), lo));
        }
        for (long long e = q1.e0 + EC; e < q1.e1; e++) {
          long long s = a.indices[e];
          if (a.rowmap) s = a.rowmap[s];
          acc4(m1, relu4(*reinterpret_cast<const float4*>(a.x + s * a.ldx + c), lo));
        }
        m0.x *= q0.inv, m0.y *= q0.inv, m0.z *= q0.inv, m0.w *= q0.inv;
        m1.x *= q1.inv, m1.y *= q1.inv, m1.z *= q1.inv, m1.w *= q1.inv;
        *reinterpret_cast<float4*>(A + rl0 * lda + c) = sv0;
        *reinterpret_cast<float4*>(A + rl0 * lda + H + c) = m0;
        *reinterpret_cast<float4*>(A + rl1 * lda + c) = sv1;
        *reinterpret_cast<float4*>(A + rl1 * lda + H + c) = m1;
        if (a.cat) {
          if (row0 + rl0 < a.n_pad) {
            float* o = a.cat + (row0 + rl0) * a.ldc;
            *reinterpret_cast<float4*>(o + c) = sv0;
            *reinterpret_cast<float4*>(o + H + c) = m0;
          }
          if (row0 + rl1 < a.n_pad) {
            float* o = a.cat + (row0 + rl1) * a.ldc;
            *reinterpret_cast<float4*>(o + c) = sv1;
            *reinterpret_cast<float4*>(o + H + c) = m1;
          }
        }
      }
    }
  }
  const unsigned tl_gathered = (unsigned)__builtin_amdgcn_s_memrealtime();
  __syncthreads();

  // ---- multiply: this wave's two n-tiles x MT m-tiles over K2 = 2 H
  const int nt0 = 2 * wave;
  if (nt0 >= a.NTp || (a.dbg & 2)) return;   // (narrow layers: fewer than 8 n-tiles; no barrier follows)
  const int KQ = (2 * H) / 8;
  const int h = lane >> 5, l31 = lane & 31;
  f32x16 acc[MT][2];
#pragma unroll
  for (int mt = 0; mt < MT; mt++)
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int i = 0; i < 16; i++) acc[mt][t][i] = 0.f;
  const float* ab = A + l31 * lda + 4 * h;
  const float4* wb = a.wp + (long long)nt0 * 64 + lane;
  const long long wstep = (long long)a.NTp * 64;
  float4 b0 = wb[0], b1 = wb[64];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int q = 0; q < KQ; q++) {
    float4 nb0 = b0, nb1 = b1;
    if (q + 1 < KQ) {
      nb0 = wb[(q + 1) * wstep];
      nb1 = wb[(q + 1) * wstep + 64];
    }
    float4 av[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) av[mt] = *reinterpret_cast<const float4*>(ab + mt * 32 * lda + 8 * q);
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
      acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt].x, b0.x, acc[mt][0], 0, 0, 0);
      acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt].x, b1.x, acc[mt][1], 0, 0, 0);
    }
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
      acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt].y, b0.y, acc[mt][0], 0, 0, 0);
      acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt].y, b1.y, acc[mt][1], 0, 0, 0);
    }
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
      acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt].z, b0.z, acc[mt][0], 0, 0, 0);
      acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt].z, b1.z, acc[mt][1], 0, 0, 0);
    }
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
      acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt].w, b0.w, acc[mt][0], 0, 0, 0);
      acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt].w, b1.w, acc[mt][1], 0, 0, 0);
    }
    b0 = nb0, b1 = nb1;
  }

  const unsigned tl_loop_end = (unsigned)__builtin_amdgcn_s_memrealtime();
  if (a.dbg & 32) {   // diagnostics: shader cycles and 100 MHz ticks of the multiply loop, in place of the tile's output
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
      a.y[(row0 + wave) * a.ldy + 0] = (float)(t1 - t0);
      a.y[(row0 + wave) * a.ldy + 1] = (float)(r1 - r0);
    }
    return;
  }
  // ---- epilogue: C/D of a 32x32 tile: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  const bool whole = row0 + BM <= a.n_pad;   // (uniform: every row of the tile exists)
#pragma unroll
  for (int t = 0; t < 2; t++) {
    const int col = 32 * (nt0 + t) + l31;
    if (col >= a.out) continue;
    const float bv = a.bias ? a.bias[col] : 0.f;
    float* yc = a.y + row0 * a.ldy + col;
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int rl = mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        float v = acc[mt][t][i] + bv;
        if (a.relu_out) v = fmaxf(v, 0.f);
        if ((a.dbg & 4) && v != 12345.678f) continue;
        if (whole || row0 + rl < a.n_pad) yc[(long long)rl * a.ldy] = v;
      }
    }
  }
  if (a.dbg & 64) {   // diagnostics: this wave's timeline (10 ns ticks) and where it ran, in place of output words
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned tl_end = (unsigned)__builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
      unsigned* o = reinterpret_cast<unsigned*>(a.y + (row0 + wave) * a.ldy);
      o[0] = tl_start, o[1] = tl_gathered, o[2] = (unsigned)r0, o[3] = tl_loop_end, o[4] = tl_end;
      o[5] = __builtin_amdgcn_s_getreg(63492), o[6] = __builtin_amdgcn_s_getreg(63508);
    }
  }
}

inline int lda_for(int H) { return 2 * H + 4; }  // 2 H is a multiple of 8, so (2 H + 4) / 4 is odd
inline int ntp_for(int out) { return ((out + 31) / 32 + 1) & ~1; }
// rows per workgroup: 64 while three workgroups fit a CU's 160 KB of LDS, else 32
inline int mt_for(int H) { return 64 * lda_for(H) * 4 <= 53 * 1024 ? 2 : 1; }

}  // namespace

extern "C" {

int64_t csl_sage_fwd_mfma_scratch(int32_t H, int32_t out) {
  if (H < 4 || H % 4 != 0 || out < 1 || out > 256) return CSL_E_INVALID;
  return (int64_t)(2 * H / 8) * ntp_for(out) * 64 * 4;
}

int csl_sage_fwd_mfma_f32(const int32_t* indptr, const int32_t* indices, const int32_t* self_ids, const int32_t* rowmap,
                          const float* x, int64_t ldx, const float* W, int64_t ldw, const float* bias, int64_t n,
                          int64_t n_pad, int32_t H, int32_t out, int32_t relu_in, int32_t relu_out, float* cat,
                          int64_t ldc, float* y, int64_t ldy, float* wpack, void* stream) {
  if (n_pad == 0) return CSL_OK;
  if (n < 0 || n_pad < n || H < 4 || H % 4 != 0 || out < 1 || out > 256 || !W || ldw < 2 * (int64_t)H || ldw % 4 != 0 ||
      !y || ldy < out || !wpack || ((uintptr_t)W & 15) || ((uintptr_t)wpack & 15))
    return CSL_E_INVALID;
  if (n > 0 && (!indptr || !self_ids || !x || ldx < H || ldx % 4 != 0 || ((uintptr_t)x & 15)))
    return CSL_E_INVALID;
  if (cat && (ldc < 2 * (int64_t)H || ldc % 4 != 0 || ((uintptr_t)cat & 15))) return CSL_E_INVALID;
  int dbg = 0;
  {
    const char* e = getenv("CSLICER_MFMA_DBG");
    dbg = e ? atoi(e) : 0;
  }
  const int lda = lda_for(H), NTp = ntp_for(out), KQ = 2 * H / 8, MT = (dbg & 8) ? 1 : mt_for(H);
  const size_t lds = (size_t)32 * MT * lda * sizeof(float);
  if (lds > 160 * 1024) return CSL_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const long long nw = (long long)KQ * NTp * 64;
  hipLaunchKernelGGL(k_pack_w, dim3((unsigned)((nw + TB - 1) / TB)), dim3(TB), 0, st, W, (long long)ldw, (int)out, KQ, NTp,
                     reinterpret_cast<float4*>(wpack));
  FwdArgs a;
  a.indptr = indptr, a.indices = indices, a.self_ids = self_ids, a.rowmap = rowmap;
  a.x = x, a.ldx = ldx, a.wp = reinterpret_cast<const float4*>(wpack), a.bias = bias;
  a.cat = cat, a.ldc = ldc, a.y = y, a.ldy = ldy, a.n = n, a.n_pad = n_pad;
  a.H = H, a.out = out, a.NTp = NTp, a.relu_in = relu_in, a.relu_out = relu_out, a.lda = lda;
  a.dbg = dbg;
  if (dbg & 16) {
    int nb = -1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, MT == 2 ? (const void*)k_sage_fwd_mfma<2> : (const void*)k_sage_fwd_mfma<1>, TB, lds);
    fprintf(stderr, "[sage_mfma] MT %d lds %zu blocks/CU %d\n", MT, lds, nb);
  }
  const int BM = 32 * MT;
  const unsigned grid = (unsigned)((n_pad + BM - 1) / BM);
  if (MT == 2) {
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void*)k_sage_fwd_mfma<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return CSL_E_HIP;
    hipLaunchKernelGGL(k_sage_fwd_mfma<2>, dim3(grid), dim3(TB), lds, st, a);
  } else {
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void*)k_sage_fwd_mfma<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return CSL_E_HIP;
    hipLaunchKernelGGL(k_sage_fwd_mfma<1>, dim3(grid), dim3(TB), lds, st, a);
  }
  return hipGetLastError() == hipSuccess ? CSL_OK : CSL_E_HIP;
}

}  // extern "C"

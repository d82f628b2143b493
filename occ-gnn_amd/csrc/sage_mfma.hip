// sage_mfma.hip -- a GraphSAGE layer's forward as ONE kernel on the fp32 matrix cores (gfx950):
//
//   y[r, :] = act( [ x[map(self_ids[r])] | mean over the CSR row r of x[map(indices[e])] ] . W^T + bias )
//
// i.e. DistSageConv.forward of python/layers/dist_sageconv.py:66-80 (self_gather, gather, mean, concat, Linear(2*in, out))
// on the slice CSR of python/data/bipartite.py:61-67, with the gathered operand staged in LDS instead of HBM.  For the
// deepest layer of the products step (82 k rows x [100 | 100] x 256) the two-kernel form moved 65 MB of operand out to HBM
// and back in (csl_sage_cat_f32 58 us + library GEMM 96 us).
//
// Structure: ONE persistent workgroup per CU, 12 waves with fixed roles.
//   * 8 producer waves gather tile after tile of 32 output rows into an LDS ring of two operand tiles (and, optionally,
//     write the operand out: the backward's weight-gradient GEMM reads it).  The gather of a tile is the end of a chain
//     of four dependent loads (CSR row pointers -> indices -> feature-table row map -> feature rows); the chain is
//     software-pipelined over FOUR tiles, so every step issues all four stages' loads together and waits once.
//   * 4 consumer waves (one per SIMD) multiply the previous tile by the whole of W with v_mfma_f32_32x32x2_f32 (exact
//     fp32: a k-ordered fmaf chain per output), add the bias, apply the ReLU and leave the 32 x out result in a second LDS
//     ring; the producers store it as whole rows in the next step.  A consumer therefore has no store in its vector-memory
//     queue: its only global traffic is the B operand, prefetched two steps of k ahead.
//   One workgroup barrier per step.  (A first version -- every workgroup gathers its tile, then multiplies it, three
//   workgroups per CU -- ran all workgroups in lockstep: 150 us, the gather and multiply phases never overlapped and
//   the accumulator stores of a whole chip at once took 12-25 us per tile: profiles/r3_mfma/.)
//
// Operands.  A (the gathered rows) comes from LDS, row-major with a leading dimension of 4*odd floats so that the
// 16-lane groups of a ds_read_b128 down a column of rows touch 64 different banks.  B comes straight from L2 into
// registers, one 16-byte load per lane per n-tile per 8 k: W is re-packed once per call (k_pack_w) so that this load is
// contiguous per wave.  The MFMA sums over k in any order as long as A and B agree, so one float4 of a lane feeds FOUR
// consecutive MFMAs: lanes 0-31 (k index 0 of the instruction) hold k = 8q .. 8q+3 of their row / column, lanes 32-63
// (k index 1) hold k = 8q+4 .. 8q+7, and MFMA j of the group uses element j of both.
//
// All of it is plain HIP; the C ABI is in cslicer_aggr.h.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "cslicer_aggr.h"
#include "cslicer_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NCONS = 4;                    // consumer (MFMA) waves: one per SIMD
constexpr int NPROD = 8;                    // producer (gather / store) waves
constexpr int TBP = 64 * (NCONS + NPROD);   // 768 threads
constexpr int PT = 64 * NPROD;              // producer threads
constexpr int BM = 32;                      // rows of a tile: one MFMA m-tile
constexpr int EC = 6;                       // edges of a row whose feature rows are requested together
constexpr int EMAX = 16;                    // resolved neighbour rows staged per output row (longer rows: slow path)
constexpr int IDXW = EMAX + 2;              // [degree, self row, neighbour rows]

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
  int H, out, NTp, relu_in, relu_out, lda, ldo, n_tiles, vec_y;
  // the index pipeline loads unconditionally (at element 0 where a lane has nothing to fetch): never-null stand-ins
  const int* indptr_ld;
  const int* indices_ld;
  const int* self_ld;
  const int* rowmap_ld;
  int dbg;            // diagnostics (CSLICER_MFMA_DBG): 1 = no feature loads, 2 = no multiply
};

// W [out][ldw] -> wp[q][nt][lane].j = W[32 nt + (lane & 31)][8 q + 4 (lane >> 5) + j]   (zero beyond `out`)
__global__ __launch_bounds__(256) void k_pack_w(const float* __restrict__ W, long long ldw, int out, int KQ, int NTp,
                                                float4* __restrict__ wp) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
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

// ---- producer: one step.  Tile ids of this workgroup: b, b + G, b + 2 G, ... (S of them); step t gathers tile t's
// feature rows (stage F), resolves tile t+1's rows through the row map (R), reads tile t+2's indices (I) and tile
// t+3's row pointers (P), and stores the result tile t-2 the consumers left in LDS.
struct ProdState {
  int p_e0, p_deg, p_sid;    // stage P -> I: CSR range and self id of this thread's row
  int i_raw, i_sid, i_deg;   // stage I -> R: this thread's edge slot (raw index), self id, degree
};

__device__ __forceinline__ void producer_step(const FwdArgs& a, int t, int S, int b, int G, int j, float* Abuf,
                                              float* Obuf, int* Ibuf, ProdState& st) {
  const int H = a.H, lda = a.lda, ldo = a.ldo;
  const float lo = a.relu_in ? 0.f : -__builtin_inff();
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);

  // ---- the result tile t-2: LDS -> y, whole rows
  if (t >= 2 && t - 2 < S) {
    const float* O = Obuf + ((t - 2) & 1) * BM * ldo;
    const long long row0 = (long long)(b + (long long)(t - 2) * G) * BM;
    const int q4 = ldo >> 2;   // float4 per staged row
    for (int e = j; e < BM * q4; e += PT) {
      const int rl = e / q4, c = (e - rl * q4) * 4;
      const long long r = row0 + rl;
      if (r >= a.n_pad || c >= a.out) continue;
      const float4 v = *reinterpret_cast<const float4*>(O + rl * ldo + c);
      float* yp = a.y + r * a.ldy + c;
      if (a.vec_y) {
        *reinterpret_cast<float4*>(yp) = v;
      } else {
        yp[0] = v.x;
        if (c + 1 < a.out) yp[1] = v.y;
        if (c + 2 < a.out) yp[2] = v.z;
        if (c + 3 < a.out) yp[3] = v.w;
      }
    }
  }

  // ---- stage F: feature rows of tile t.  32 lanes per row, this group's rows g and g + 16, both in flight.
  const bool doF = t >= 0 && t < S;
  const int g = j >> 5, gl = j & 31;
  const int* I0 = Ibuf + (t & 1) * BM * IDXW + g * IDXW;
  const int* I1 = I0 + 16 * IDXW;
  int deg0 = 0, deg1 = 0, sr0 = -1, sr1 = -1;
  int s0[EC], s1[EC];
  float4 v0[EC], v1[EC], sv0 = z4, sv1 = z4;
  int c = gl * 4;
#pragma unroll
  for (int u = 0; u < EC; u++) s0[u] = -1, s1[u] = -1, v0[u] = z4, v1[u] = z4;
  if (doF) {
    deg0 = I0[0], sr0 = I0[1], deg1 = I1[0], sr1 = I1[1];
#pragma unroll
    for (int u = 0; u < EC; u++) {
      s0[u] = u < deg0 ? I0[2 + u] : -1;
      s1[u] = u < deg1 ? I1[2 + u] : -1;
    }
    if (c < H && !(a.dbg & 1)) {
      if (sr0 >= 0) sv0 = *reinterpret_cast<const float4*>(a.x + (long long)sr0 * a.ldx + c);
      if (sr1 >= 0) sv1 = *reinterpret_cast<const float4*>(a.x + (long long)sr1 * a.ldx + c);
#pragma unroll
      for (int u = 0; u < EC; u++) {
        if (s0[u] >= 0) v0[u] = *reinterpret_cast<const float4*>(a.x + (long long)s0[u] * a.ldx + c);
        if (s1[u] >= 0) v1[u] = *reinterpret_cast<const float4*>(a.x + (long long)s1[u] * a.ldx + c);
      }
    }
  }

  // ---- the index pipeline: thread j is (row pr, edge slot ps) of the tile
  const int pr = j >> 4, ps = j & 15;
  // Every load below is issued unconditionally, at element 0 where the lane (or the whole step) has nothing to fetch,
  // and its value is selected after stage F has been finished: a branch around a load, divergent or uniform, ends the
  // compiler's scheduling region and it then waits for the whole queue before the next stage's loads are issued.
  // stage R: tile t+1's raw indices -> feature-table rows
  const bool doR = t + 1 >= 0 && t + 1 < S;
  const int m_nbr = a.rowmap_ld[(a.rowmap && st.i_raw >= 0) ? st.i_raw : 0];
  const int m_self = a.rowmap_ld[(a.rowmap && st.i_sid >= 0) ? st.i_sid : 0];
  // stage I: tile t+2's indices
  const int m_raw = a.indices_ld[(a.indices && ps < st.p_deg) ? (long long)st.p_e0 + ps : 0];
  // stage P: tile t+3's row pointers and self ids
  const long long rP = (long long)(b + (long long)(t + 3) * G) * BM + pr;
  const bool okP = t + 3 < S && rP < a.n;   // (t + 3 >= 0 always)
  const long long rc = okP ? rP : 0;
  const int m_e0 = a.indptr_ld[rc], m_e1 = a.indptr_ld[rc + 1], m_sid = a.self_ld[rc];

  // ---- finish stage F (its loads are the oldest in the queue)
  if (doF) {
    const long long row0 = (long long)(b + (long long)t * G) * BM;
    float* A = Abuf + (t & 1) * BM * lda;
    const float inv0 = 1.0f / (float)(deg0 > 1 ? deg0 : 1), inv1 = 1.0f / (float)(deg1 > 1 ? deg1 : 1);
    for (; c < H; c += 128) {
      if (c != gl * 4 && !(a.dbg & 1)) {   // rows wider than 128 columns: further quads of the same rows
        sv0 = z4, sv1 = z4;
        if (sr0 >= 0) sv0 = *reinterpret_cast<const float4*>(a.x + (long long)sr0 * a.ldx + c);
        if (sr1 >= 0) sv1 = *reinterpret_cast<const float4*>(a.x + (long long)sr1 * a.ldx + c);
#pragma unroll
        for (int u = 0; u < EC; u++) {
          v0[u] = z4, v1[u] = z4;
          if (s0[u] >= 0) v0[u] = *reinterpret_cast<const float4*>(a.x + (long long)s0[u] * a.ldx + c);
          if (s1[u] >= 0) v1[u] = *reinterpret_cast<const float4*>(a.x + (long long)s1[u] * a.ldx + c);
        }
      }
      if (sr0 >= 0) sv0 = relu4(sv0, lo);
      if (sr1 >= 0) sv1 = relu4(sv1, lo);
      float4 m0 = z4, m1 = z4;
#pragma unroll
      for (int u = 0; u < EC; u++) {
        if (s0[u] >= 0) acc4(m0, relu4(v0[u], lo));
        if (s1[u] >= 0) acc4(m1, relu4(v1[u], lo));
      }
      // rows with more than EC edges (shallower layers' fanouts): staged rows up to EMAX, then through the index arrays
      if (!(a.dbg & 1)) {
        for (int u = EC; u < deg0; u++) {
          long long s;
          if (u < EMAX) {
            s = I0[2 + u];
          } else {
            s = a.indices[(long long)a.indptr[row0 + g] + u];
            if (a.rowmap) s = a.rowmap[s];
          }
          acc4(m0, relu4(*reinterpret_cast<const float4*>(a.x + s * a.ldx + c), lo));
        }
        for (int u = EC; u < deg1; u++) {
          long long s;
          if (u < EMAX) {
            s = I1[2 + u];
          } else {
            s = a.indices[(long long)a.indptr[row0 + g + 16] + u];
            if (a.rowmap) s = a.rowmap[s];
          }
          acc4(m1, relu4(*reinterpret_cast<const float4*>(a.x + s * a.ldx + c), lo));
        }
      }
      m0.x *= inv0, m0.y *= inv0, m0.z *= inv0, m0.w *= inv0;
      m1.x *= inv1, m1.y *= inv1, m1.z *= inv1, m1.w *= inv1;
      *reinterpret_cast<float4*>(A + g * lda + c) = sv0;
      *reinterpret_cast<float4*>(A + g * lda + H + c) = m0;
      *reinterpret_cast<float4*>(A + (g + 16) * lda + c) = sv1;
      *reinterpret_cast<float4*>(A + (g + 16) * lda + H + c) = m1;
      if (a.cat && !(a.dbg & 64)) {
        if (row0 + g < a.n_pad) {
          float* o = a.cat + (row0 + g) * a.ldc;
          *reinterpret_cast<float4*>(o + c) = sv0;
          *reinterpret_cast<float4*>(o + H + c) = m0;
        }
        if (row0 + g + 16 < a.n_pad) {
          float* o = a.cat + (row0 + g + 16) * a.ldc;
          *reinterpret_cast<float4*>(o + c) = sv1;
          *reinterpret_cast<float4*>(o + H + c) = m1;
        }
      }
    }
  }

  // ---- hand the pipeline on
  if (doR) {
    int* In = Ibuf + ((t + 1) & 1) * BM * IDXW + pr * IDXW;
    In[2 + ps] = st.i_raw >= 0 ? (a.rowmap ? m_nbr : st.i_raw) : -1;
    if (ps == 0) In[0] = st.i_deg, In[1] = st.i_sid >= 0 ? (a.rowmap ? m_self : st.i_sid) : -1;
  }
  st.i_raw = (a.indices && ps < st.p_deg) ? m_raw : -1, st.i_sid = st.p_sid, st.i_deg = st.p_deg;
  st.p_e0 = okP ? m_e0 : 0, st.p_deg = okP ? m_e1 - m_e0 : 0, st.p_sid = okP ? m_sid : -1;
}

// ---- consumer: multiply tile t-1 (LDS) by this wave's two n-tiles of W, leave act(. + bias) in the result ring.
// B of k-group q sits in buffer q % 3 and is requested two groups ahead, A of group q in buffer q % 2, one ahead: the
// loop is unrolled by six so that every buffer index is a constant (a rotation by register moves made the compiler
// wait for each load right behind its issue).  Groups 0 and 1 of B never change: they stay in registers (rb).
struct ConsRegs {
  float4 rb[2][2];   // B of k-groups 0 and 1, n-tiles 0 and 1 of this wave
  float bv0, bv1;    // bias of this lane's two columns
};

#define MFMA8(AV, B0, B1)                                                   \
  acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).x, (B0).x, acc0, 0, 0, 0); \
  acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).x, (B1).x, acc1, 0, 0, 0); \
  acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).y, (B0).y, acc0, 0, 0, 0); \
  acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).y, (B1).y, acc1, 0, 0, 0); \
  acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).z, (B0).z, acc0, 0, 0, 0); \
  acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).z, (B1).z, acc1, 0, 0, 0); \
  acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).w, (B0).w, acc0, 0, 0, 0); \
  acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).w, (B1).w, acc1, 0, 0, 0);

__device__ __forceinline__ void consumer_step(const FwdArgs& a, int t, int wave, int lane, const float* Abuf, float* Obuf,
                                              const ConsRegs& cr) {
  const int nt0 = 2 * wave;
  const int lda = a.lda, ldo = a.ldo;
  const float* A = Abuf + ((t - 1) & 1) * BM * lda;
  float* O = Obuf + ((t - 1) & 1) * BM * ldo;
  const int KQ = (2 * a.H) / 8;
  const int h = lane >> 5, l31 = lane & 31;
  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; i++) acc0[i] = cr.bv0, acc1[i] = cr.bv1;   // C starts as the bias of the lane's column
  const float* ab = A + l31 * lda + 4 * h;
  const float4* wb = a.wp + (long long)nt0 * 64 + lane;
  const long long wstep = (long long)a.NTp * 64;
  float4 bb[3][2], aa[2];
  bb[0][0] = cr.rb[0][0], bb[0][1] = cr.rb[0][1], bb[1][0] = cr.rb[1][0], bb[1][1] = cr.rb[1][1];
  bb[2][0] = bb[0][0], bb[2][1] = bb[0][1];
  aa[0] = *reinterpret_cast<const float4*>(ab);
  aa[1] = aa[0];
  int q0 = 0;
  for (; q0 + 6 <= KQ; q0 += 6) {   // (loads past the last group re-read it: no branch, no drained queue)
#pragma unroll
    for (int i = 0; i < 6; i++) {
      const int q = q0 + i, qb = q + 2 < KQ ? q + 2 : KQ - 1, qa = q + 1 < KQ ? q + 1 : KQ - 1;
      if (!(a.dbg & 256)) {
        bb[(i + 2) % 3][0] = wb[qb * wstep];
        bb[(i + 2) % 3][1] = wb[qb * wstep + 64];
      }
      if (!(a.dbg & 512)) aa[(i + 1) % 2] = *reinterpret_cast<const float4*>(ab + 8 * qa);
      MFMA8(aa[i % 2], bb[i % 3][0], bb[i % 3][1])
    }
  }
#pragma unroll
  for (int i = 0; i < 5; i++) {   // the last KQ % 6 groups
    const int q = q0 + i;
    if (q < KQ) {
      const int qb = q + 2 < KQ ? q + 2 : KQ - 1, qa = q + 1 < KQ ? q + 1 : KQ - 1;
      bb[(i + 2) % 3][0] = wb[qb * wstep];
      bb[(i + 2) % 3][1] = wb[qb * wstep + 64];
      aa[(i + 1) % 2] = *reinterpret_cast<const float4*>(ab + 8 * qa);
      MFMA8(aa[i % 2], bb[i % 3][0], bb[i % 3][1])
    }
  }
  // C/D of a 32x32 tile: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  const int col0 = 32 * nt0 + l31, col1 = col0 + 32;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const int rl = (i & 3) + 8 * (i >> 2) + 4 * h;
    float u0 = acc0[i], u1 = acc1[i];
    if (a.relu_out) u0 = fmaxf(u0, 0.f), u1 = fmaxf(u1, 0.f);
    O[rl * ldo + col0] = u0;
    O[rl * ldo + col1] = u1;
  }
}
#undef MFMA8

__global__ __launch_bounds__(TBP) void k_sage_fwd_mfma(const FwdArgs a) {
  extern __shared__ float4 smem4[];
  float* Abuf = reinterpret_cast<float*>(smem4);               // [2][BM][lda]
  float* Obuf = Abuf + 2 * BM * a.lda;                         // [2][BM][ldo]
  int* Ibuf = reinterpret_cast<int*>(Obuf + 2 * BM * a.ldo);   // [2][BM][IDXW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = gridDim.x, b = blockIdx.x;
  const int S = (a.n_tiles - b + G - 1) / G;   // this workgroup's tiles: b, b + G, ...
  if (wave >= NCONS) {
    ProdState st;
    st.p_e0 = 0, st.p_deg = 0, st.p_sid = -1, st.i_raw = -1, st.i_sid = -1, st.i_deg = 0;
    const int j = tid - 64 * NCONS;
    for (int t = -3; t <= S + 1; t++) {
      const unsigned s0 = (unsigned)__builtin_amdgcn_s_memrealtime();
      producer_step(a, t, S, b, G, j, Abuf, Obuf, Ibuf, st);
      const unsigned s1 = (unsigned)__builtin_amdgcn_s_memrealtime();
      __syncthreads();
      if ((a.dbg & 64) && j == 0 && t + 3 < 64) {   // diagnostics: 10 ns ticks of this step, into the (unused) cat buffer
        unsigned* o = reinterpret_cast<unsigned*>(a.cat) + ((b * 64 + (t + 3)) * 2 + 1) * 4;
        o[0] = s0, o[1] = s1, o[2] = (unsigned)__builtin_amdgcn_s_memrealtime(), o[3] = S;
      }
    }
  } else {
    __builtin_amdgcn_s_setprio(2);
    const int nt0 = 2 * wave;
    const bool work = nt0 < a.NTp && !(a.dbg & 2);   // (narrow layers: fewer than 8 n-tiles)
    ConsRegs cr;
    cr.bv0 = cr.bv1 = 0.f;
    cr.rb[0][0] = cr.rb[0][1] = cr.rb[1][0] = cr.rb[1][1] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (work) {
      const float4* wb = a.wp + (long long)nt0 * 64 + lane;
      const long long wstep = (long long)a.NTp * 64;
      const int KQ = (2 * a.H) / 8;
      cr.rb[0][0] = wb[0], cr.rb[0][1] = wb[64];
      if (KQ > 1) cr.rb[1][0] = wb[wstep], cr.rb[1][1] = wb[wstep + 64];
      const int col0 = 32 * nt0 + (lane & 31), col1 = col0 + 32;
      if (a.bias && col0 < a.out) cr.bv0 = a.bias[col0];
      if (a.bias && col1 < a.out) cr.bv1 = a.bias[col1];
    }
    for (int t = -3; t <= S + 1; t++) {
      const unsigned s0 = (unsigned)__builtin_amdgcn_s_memrealtime();
      if (work && t >= 1 && t <= S) consumer_step(a, t, wave, lane, Abuf, Obuf, cr);
      const unsigned s1 = (unsigned)__builtin_amdgcn_s_memrealtime();
      __syncthreads();
      if ((a.dbg & 64) && tid == 0 && t + 3 < 64) {
        unsigned* o = reinterpret_cast<unsigned*>(a.cat) + ((b * 64 + (t + 3)) * 2 + 0) * 4;
        o[0] = s0, o[1] = s1, o[2] = (unsigned)__builtin_amdgcn_s_memrealtime(), o[3] = S;
      }
    }
  }
}

inline int lda_for(int H) { return 2 * H + 4; }  // 2 H is a multiple of 8, so (2 H + 4) / 4 is odd
inline int ntp_for(int out) { return ((out + 31) / 32 + 1) & ~1; }
inline size_t lds_for(int H, int out) {
  return (size_t)2 * BM * (lda_for(H) + ntp_for(out) * 32) * sizeof(float) + (size_t)2 * BM * IDXW * sizeof(int);
}

}  // namespace

extern "C" {

int64_t csl_sage_fwd_mfma_scratch(int32_t H, int32_t out) {
  if (H < 4 || H % 4 != 0 || out < 1 || out > 256 || lds_for(H, out) > 160 * 1024) return CSL_E_INVALID;
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
  if (n > 0 && (!indptr || !self_ids || !x || ldx < H || ldx % 4 != 0 || ((uintptr_t)x & 15))) return CSL_E_INVALID;
  if (cat && (ldc < 2 * (int64_t)H || ldc % 4 != 0 || ((uintptr_t)cat & 15))) return CSL_E_INVALID;
  const size_t lds = lds_for(H, out);
  if (lds > 160 * 1024) return CSL_E_INVALID;
  if ((n_pad + BM - 1) / BM > 0x7fffffffLL / BM) return CSL_E_INVALID;
  int dbg = 0;
  {
    const char* e = getenv("CSLICER_MFMA_DBG");
    dbg = e ? atoi(e) : 0;
  }
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return CSL_E_HIP;
    n_cu = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
  }
  const int NTp = ntp_for(out), KQ = 2 * H / 8;
  hipStream_t st = (hipStream_t)stream;
  const long long nw = (long long)KQ * NTp * 64;
  hipLaunchKernelGGL(k_pack_w, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, st, W, (long long)ldw, (int)out, KQ, NTp,
                     reinterpret_cast<float4*>(wpack));
  FwdArgs a;
  a.indptr = indptr, a.indices = indices, a.self_ids = self_ids, a.rowmap = rowmap;
  a.x = x, a.ldx = ldx, a.wp = reinterpret_cast<const float4*>(wpack), a.bias = bias;
  a.cat = cat, a.ldc = ldc, a.y = y, a.ldy = ldy, a.n = n, a.n_pad = n_pad;
  a.H = H, a.out = out, a.NTp = NTp, a.relu_in = relu_in, a.relu_out = relu_out, a.lda = lda_for(H), a.ldo = NTp * 32;
  a.n_tiles = (int)((n_pad + BM - 1) / BM);
  a.vec_y = (out % 4 == 0 && ldy % 4 == 0 && ((uintptr_t)y & 15) == 0) ? 1 : 0;
  a.indptr_ld = (indptr && n > 0) ? indptr : reinterpret_cast<const int*>(wpack);
  a.self_ld = (self_ids && n > 0) ? self_ids : reinterpret_cast<const int*>(wpack);
  a.indices_ld = indices ? indices : reinterpret_cast<const int*>(wpack);
  a.rowmap_ld = rowmap ? rowmap : reinterpret_cast<const int*>(wpack);
  a.dbg = dbg;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)k_sage_fwd_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return CSL_E_HIP;
    attr_set = true;
  }
  const unsigned grid = (unsigned)(a.n_tiles < n_cu ? a.n_tiles : n_cu);
  hipLaunchKernelGGL(k_sage_fwd_mfma, dim3(grid), dim3(TBP), lds, st, a);
  return hipGetLastError() == hipSuccess ? CSL_OK : CSL_E_HIP;
}

}  // extern "C"

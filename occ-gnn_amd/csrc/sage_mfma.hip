// sage_mfma.hip -- a GraphSAGE layer's forward as ONE kernel on the fp32 matrix cores (gfx950):
//
//   y[r, :] = act( [ x[map(self_ids[r])] | mean over the CSR row r of x[map(indices[e])] ] . W^T + bias )
//
// i.e. DistSageConv.forward of python/layers/dist_sageconv.py:66-80 (self_gather, gather, mean, concat, Linear(2*in, out))
// on the slice CSR of python/data/bipartite.py:61-67, with the gathered operand staged in LDS instead of HBM.  For the
// deepest layer of the products step (82 k rows x [100 | 100] x 256) the two-kernel form moved 65 MB of operand out to HBM
// and back in (csl_sage_cat_f32 58 us + library GEMM 96 us).
//
// Structure: ONE persistent workgroup per CU, 8 waves with fixed roles, two per SIMD, 256 registers each.
//   * 4 producer waves gather tile after tile of 32 output rows into an LDS ring of two operand tiles (and, optionally,
//     write the operand out: the backward's weight-gradient GEMM reads it).  The gather of a tile is the end of a chain
//     of four dependent loads (CSR row pointers -> indices -> feature-table row map -> feature rows); the chain is
//     software-pipelined over FOUR tiles, so every step issues all four stages' loads together and waits once.
//   * 4 consumer waves (one per SIMD) multiply the previous tile by W with v_mfma_f32_32x32x2_f32 (exact fp32: a
//     k-ordered fmaf chain per output, started from the bias), apply the ReLU and leave the 32 x out result in a second
//     LDS ring; the producers store it as whole rows in the next step.  A wave owns 64 output columns, and its slice of W
//     is STATIONARY ON THE CU: 8 registers per lane per k-group of 8 for the first KS groups (19 of the 25 at in = 100:
//     152 registers next to 32 of accumulators), the next KL groups in LDS (6: 48 KB), and only what is beyond that
//     (layers wider than in = 100) streamed from L2 two groups ahead.  At in = 100 a consumer touches no global memory
//     at all in the steady state.
//   One workgroup barrier per step.
// What came before (profiles/r3_mfma/): (v1) every workgroup gathers its tile, then multiplies it, three workgroups per
// CU: all workgroups ran in lockstep, 150 us, the gather and multiply phases never overlapped and the accumulator stores
// of a whole chip at once took 12-25 us per tile.  (v2) these roles with B streamed from L2 by the consumers: 142 us, a
// consumer's B loads sat in the CU's memory pipeline behind the producers' HBM gathers (12.3 us per 32-row tile with the
// gather running, 9.2 without, 6.9 with no loads in the loop at all).
//
// Operands.  A (the gathered rows) comes from LDS, row-major with a leading dimension of 4*odd floats so that the
// 16-lane groups of a ds_read_b128 down a column of rows touch 64 different banks.  W is re-packed once per call
// (k_pack_w) so that a lane's B values of a k-group are one contiguous 16-byte load.  The MFMA sums over k in any order
// as long as A and B agree, so one float4 of a lane feeds FOUR consecutive MFMAs: lanes 0-31 (k index 0 of the
// instruction) hold k = 8q .. 8q+3 of their row / column, lanes 32-63 (k index 1) hold k = 8q+4 .. 8q+7, and MFMA j of
// the group uses element j of both.
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
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// raw buffer descriptors (stride 0) for the consumers' stores: 32-bit byte offsets (one lane offset per tile + the row as a
// scalar offset: no 64-bit vector arithmetic per store), and an offset beyond num_records stores nothing.
// (Loads do not go this way: __builtin_amdgcn_raw_buffer_load_b128 is lowered to a ONE-dword load by ROCm 7.2's clang 22.)
constexpr unsigned SRD_FLAGS = 0x00020000;
constexpr unsigned OOB = 0xF0000000u;   // beyond every buffer this path accepts (< 3.75 GB)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, SRD_FLAGS);
}
constexpr int ZROW4 = 80;                   // float4 of the zero row (320 floats: in <= 298 is all that fits the LDS)
constexpr int NCONS = 4;                    // consumer (MFMA) waves: one per SIMD
constexpr int NPROD = 4;                    // producer (gather / store) waves: one per SIMD
constexpr int TBP = 64 * (NCONS + NPROD);   // 512 threads: two waves per SIMD, 256 registers each
constexpr int PT = 64 * NPROD;              // producer threads
constexpr int BM = 32;                      // rows of a tile: one MFMA m-tile
constexpr int EC = 6;                       // edges of a row whose feature rows are requested together
constexpr int NR = 4;                       // rows a group of 32 producer lanes gathers at once (BM / (PT / 32))
constexpr int EMAX = 16;                    // resolved neighbour rows staged per output row (longer rows: generic path)
constexpr int IDXW = 3 * EC + 2;            // staged per output row: [degree, address of the self row, of the neighbour rows]:
                                            // every slot the fast path's three edge passes can load, so that the slots beyond
                                            // EMAX hold the zero row's address too (they are loaded from unconditionally)
static_assert(3 * EC >= EMAX, "the staged row covers every slot a fast-path pass loads");
typedef unsigned long long StagedT;          // (64-bit: an address, the row of zeros where there is nothing to fetch)
// (an address that went through LDS as an integer is loaded from as GLOBAL memory, said explicitly: a generic pointer
// would make it a flat load, which counts in both wait counters and completes out of order)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const f32x4 __attribute__((address_space(1)))* GlobalF4;
__device__ __forceinline__ float4 ldg4(StagedT addr) {
  const f32x4 v = *reinterpret_cast<GlobalF4>((uintptr_t)addr);
  return make_float4(v.x, v.y, v.z, v.w);
}
constexpr int NSL = EMAX * BM / PT;         // edge slots of a row per producer thread in the index pipeline (2)

struct FwdArgs {
  const int* indptr;
  const int* indices;
  const int* self_ids;
  const int* rowmap;
  const float* x;
  long long ldx;
  const float4* wp;   // packed W: [K2/8][NTp][64 lanes] float4
  const float* zero;  // a row of zeros behind it (as wide as the widest accepted layer): what a lane with nothing to fetch loads
  const float* bias;
  float* cat;         // optional [n_pad][ldc]
  long long ldc;
  float* y;           // [n_pad][ldy]
  long long ldy;
  long long n, n_pad;
  int H, out, NTp, relu_in, relu_out, lda, n_tiles;
  // the index pipeline loads unconditionally (at element 0 where a lane has nothing to fetch): never-null stand-ins
  const int* indptr_ld;
  const int* indices_ld;
  const int* self_ld;
  const int* rowmap_ld;
  unsigned y_bytes;   // n_pad * ldy * 4 if < 4 GB, else 0
  unsigned cat_bytes;
  int dbg;            // diagnostics (CSLICER_MFMA_DBG): 1 = no feature loads, 2 = no multiply, 64 = step stamps into cat
};

// W [out][ldw] -> wp[q][nt][lane].j = W[32 nt + (lane & 31)][8 q + 4 (lane >> 5) + j]   (zero beyond `out`)
__global__ __launch_bounds__(256) void k_pack_w(const float* __restrict__ W, long long ldw, int out, int KQ, int NTp,
                                                float4* __restrict__ wp) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)KQ * NTp * 64) {
    if (i < (long long)KQ * NTp * 64 + ZROW4) wp[i] = make_float4(0.f, 0.f, 0.f, 0.f);   // the zero row behind the packed W
    return;
  }
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

// ---- producers.  Tile ids of this workgroup: b, b + G, b + 2 G, ... (S of them).  Step t finishes tile t's feature
// rows (stage F), resolves tile t+2's rows through the row map (R), reads tile t+3's indices (I) and tile t+4's row
// pointers (P), and stores the result tile t-2 the consumers left in LDS.
struct ProdState {
  int p_e0, p_deg, p_sid;         // stage P -> I: CSR range and self id of this thread's row
  int i_raw[NSL], i_sid, i_deg;   // stage I -> R: this thread's edge slots (raw indices), self id, degree
  float4 r[2][2 * (EC + 1)];      // stage F in flight (fast path): two chunks of [two rows x (self + EC edges)] quads
};

// the index pipeline: thread j is row pr of the tile, edge slots ps and ps + 8.  Its loads are issued unconditionally,
// at element 0 where the lane (or the whole step) has nothing to fetch, and the values are selected later (pipe_finish):
// a branch around a load ends the compiler's scheduling region.
struct PipeLoads {
  int m_nbr[NSL], m_self, m_raw[NSL], m_e0, m_e1, m_sid;
  bool okP;
};
__device__ __forceinline__ void pipe_issue(const FwdArgs& a, int t, int S, int b, int G, int j, const ProdState& st,
                                           PipeLoads& pl) {
  const int pr = j >> 3, ps = j & 7;
#pragma unroll
  for (int k = 0; k < NSL; k++) pl.m_nbr[k] = a.rowmap_ld[(a.rowmap && st.i_raw[k] >= 0) ? st.i_raw[k] : 0];
  pl.m_self = a.rowmap_ld[(a.rowmap && st.i_sid >= 0) ? st.i_sid : 0];
#pragma unroll
  for (int k = 0; k < NSL; k++)
    pl.m_raw[k] = a.indices_ld[(a.indices && ps + 8 * k < st.p_deg) ? (long long)st.p_e0 + ps + 8 * k : 0];
  const long long rP = (long long)(b + (long long)(t + 4) * G) * BM + pr;
  pl.okP = t + 4 < S && rP < a.n;   // (t + 4 >= 0 always)
  const long long rc = pl.okP ? rP : 0;
  pl.m_e0 = a.indptr_ld[rc], pl.m_e1 = a.indptr_ld[rc + 1], pl.m_sid = a.self_ld[rc];
}
// The staged entries are ADDRESSES of the feature-table rows (of the zero row where there is none), computed here once per
// (row, slot): each of the 32 lanes that later fetch a row adds its column and loads -- one vector instruction per load
// instead of a 64-bit multiply-add and a select in every lane (the fp32 MFMA and the vector ALU share the SIMD's fp32
// lanes on this chip: every vector instruction a producer issues is taken from the consumer beside it).
__device__ __forceinline__ void pipe_finish(const FwdArgs& a, int t, int S, int j, StagedT* Ibuf, ProdState& st,
                                            const PipeLoads& pl) {
  const int pr = j >> 3, ps = j & 7;
  if (t + 2 >= 0 && t + 2 < S) {
    StagedT* Iw = Ibuf + ((t + 6) % 3) * BM * IDXW + pr * IDXW;   // tile t+2's (tile T lives in slot (T + 4) % 3)
    const StagedT zero = (StagedT)(uintptr_t)a.zero;
#pragma unroll
    for (int k = 0; k < NSL; k++) {
      const long long row = a.rowmap ? pl.m_nbr[k] : st.i_raw[k];
      Iw[2 + ps + 8 * k] = st.i_raw[k] >= 0 ? (StagedT)(uintptr_t)(a.x + row * a.ldx) : zero;
    }
    if (ps == 0) {
      const long long row = a.rowmap ? pl.m_self : st.i_sid;
      Iw[0] = (StagedT)(unsigned)st.i_deg;
      Iw[1] = st.i_sid >= 0 ? (StagedT)(uintptr_t)(a.x + row * a.ldx) : zero;
    }
  }
#pragma unroll
  for (int k = 0; k < NSL; k++) st.i_raw[k] = (a.indices && ps + 8 * k < st.p_deg) ? pl.m_raw[k] : -1;
  st.i_sid = st.p_sid, st.i_deg = st.p_deg;
  st.p_e0 = pl.okP ? pl.m_e0 : 0, st.p_deg = pl.okP ? pl.m_e1 - pl.m_e0 : 0, st.p_sid = pl.okP ? pl.m_sid : -1;
}

// ---- the fast path: rows of at most EP * EC edges, at most 128 columns.  A tile is 2 EP chunks of loads -- chunk
// (P, p): rows g + 16 P and g + 16 P + 8 of this group, their self rows (p = 0) and edges [EC p, EC p + EC) -- over two
// register sets; as soon as a chunk has been consumed its set takes the loads of the chunk after the next, of this tile
// or of the next one, so two chunks of loads are in flight at every moment, barriers included.  No loop with a memory
// operation in it between the issue and the use of a load: the compiler then waits with counted vmcnt instead of
// draining the queue.
template <int EP>
__device__ __forceinline__ void chunk_issue(const StagedT* I_tile, int g, unsigned cb, int k, float4 (&r)[2 * (EC + 1)]) {
  // every load is issued: a slot with nothing to fetch (no self row, a shorter row) holds the address of a row of zeros.
  // No branch around a load (a join the compiler's wait insertion does not count across), no select afterwards.
  const int P = k / EP, p = k % EP;
#pragma unroll
  for (int w = 0; w < 2; w++) {
    const StagedT* I = I_tile + (g + 16 * P + 8 * w) * IDXW;
    float4* rr = r + w * (EC + 1);
    if (p == 0) rr[0] = ldg4(I[1] + cb);
#pragma unroll
    for (int e = 0; e < EC; e++) rr[1 + e] = ldg4(I[2 + EC * p + e] + cb);
  }
}

template <int EP, bool RELU>
__device__ __forceinline__ void producer_step_fast(const FwdArgs& a, int t, int S, int b, int G, int j, float* Abuf,
                                                   StagedT* Ibuf, ProdState& st) {
  const int H = a.H, lda = a.lda;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  PipeLoads pl;
  pipe_issue(a, t, S, b, G, j, st, pl);
  const bool doF = t >= 0 && t < S, doN = t + 1 >= 0 && t + 1 < S;
  const int g = j >> 5, gl = j & 31, c = gl * 4;
  const unsigned cb = c < H ? (unsigned)c * 4u : 0u;    // (lanes beyond the row's width fetch its first quad again: same line)
  const StagedT* Ib = Ibuf + ((t + 4) % 3) * BM * IDXW;    // tile t's staged rows (initial values before tile 0's are there)
  const StagedT* In = doN ? Ibuf + ((t + 5) % 3) * BM * IDXW : Ib;    // tile t+1's (the last step loads the last tile again)
  float* A = Abuf + (t & 1) * BM * lda;
  float4 m[2];
  m[0] = m[1] = z4;
#pragma unroll
  for (int k = 0; k < 2 * EP; k++) {
    const int P = k / EP, p = k % EP;
    float4(&r)[2 * (EC + 1)] = st.r[k % 2];
#pragma unroll
    for (int w = 0; w < 2; w++) {
      const int rl = g + 16 * P + 8 * w;
      const float4* rr = r + w * (EC + 1);
      if (p == 0) {
        m[w] = z4;
        const float4 s4 = RELU ? relu4(rr[0], 0.f) : rr[0];
        if (doF && c < H) *reinterpret_cast<float4*>(A + rl * lda + c) = s4;
      }
#pragma unroll
      for (int e = 0; e < EC; e++) acc4(m[w], RELU ? relu4(rr[1 + e], 0.f) : rr[1 + e]);
      if (p == EP - 1) {
        const int deg = (int)Ib[rl * IDXW];
        const float inv = 1.0f / (float)(deg > 1 ? deg : 1);
        float4 mm = m[w];
        mm.x *= inv, mm.y *= inv, mm.z *= inv, mm.w *= inv;
        if (doF && c < H) *reinterpret_cast<float4*>(A + rl * lda + H + c) = mm;
      }
    }
    // the set takes the chunk after the next: of this tile, or of the next one
    if (k + 2 < 2 * EP) chunk_issue<EP>(Ib, g, cb, k + 2, r);
    else chunk_issue<EP>(In, g, cb, k + 2 - 2 * EP, r);
  }
  pipe_finish(a, t, S, j, Ibuf, st, pl);
}

// ---- the generic path (rows of any length, any width): stage F of tile t is issued and waited for inside step t
__device__ __forceinline__ void producer_step_slow(const FwdArgs& a, int t, int S, int b, int G, int j, float* Abuf,
                                                   StagedT* Ibuf, ProdState& st) {
  const int H = a.H, lda = a.lda;
  const float lo = a.relu_in ? 0.f : -__builtin_inff();
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  PipeLoads pl;
  pipe_issue(a, t, S, b, G, j, st, pl);
  if (t >= 0 && t < S) {
    const int g = j >> 5, gl = j & 31;
    const StagedT* Ib = Ibuf + ((t + 4) % 3) * BM * IDXW;
    const long long row0 = (long long)(b + (long long)t * G) * BM;
    float* A = Abuf + (t & 1) * BM * lda;
    for (int c = gl * 4; c < H; c += 128) {
#pragma unroll
      for (int u = 0; u < NR; u++) {
        const int rl = g + 8 * u;
        const StagedT* I = Ib + rl * IDXW;
        const int deg = (int)I[0];
        float4 s4 = z4, m = z4;
        if (!(a.dbg & 1)) {
          s4 = relu4(ldg4(I[1] + 4u * (unsigned)c), lo);   // (the zero row stays zero)
          for (int e = 0; e < deg; e++) {
            StagedT src;
            if (e < EMAX) {
              src = I[2 + e];
            } else {
              long long s = a.indices[(long long)a.indptr[row0 + rl] + e];
              if (a.rowmap) s = a.rowmap[s];
              src = (StagedT)(uintptr_t)(a.x + s * a.ldx);
            }
            acc4(m, relu4(ldg4(src + 4u * (unsigned)c), lo));
          }
        }
        const float inv = 1.0f / (float)(deg > 1 ? deg : 1);
        m.x *= inv, m.y *= inv, m.z *= inv, m.w *= inv;
        *reinterpret_cast<float4*>(A + rl * lda + c) = s4;
        *reinterpret_cast<float4*>(A + rl * lda + H + c) = m;
      }
    }
  }
  pipe_finish(a, t, S, j, Ibuf, st, pl);
}

template <int MODE, bool RELU>   // 1, 2, 3: fast path with that many edge passes; 0: generic
__device__ __forceinline__ void producer_loop(const FwdArgs& a, int S, int b, int G, int j, float* Abuf, StagedT* Ibuf,
                                              ProdState& st) {
  // (no global store anywhere in a producer, diagnostics included: with stores and loads both pending in the queue the
  // compiler stops counting and waits with vmcnt(0), which would drain the next tile's loads at every use of this one's)
  for (int t = -4; t <= S + 1; t++) {
    if (MODE == 0) producer_step_slow(a, t, S, b, G, j, Abuf, Ibuf, st);
    else producer_step_fast<(MODE > 0 ? MODE : 1), RELU>(a, t, S, b, G, j, Abuf, Ibuf, st);
    __syncthreads();
  }
}

// ---- consumer: multiply tile t-1 (LDS) by this wave's two n-tiles of W and store act(. + bias).
// k-groups [0, KS) of B are in registers (ws), groups [KS, KS + KL) in LDS (Wl: what the register file cannot hold next to
// the accumulators; read one group ahead like A), groups beyond are streamed from L2: group q in buffer (q - KS - KL) % 3,
// requested two groups ahead (the first two before the stationary part starts), the loop unrolled by six so that every
// buffer index is a constant (a rotation by register moves made the compiler wait for each load right behind its issue).
// A of group q sits in buffer q % 2, read one group ahead.
#define MFMA8(AV, B0, B1)                                                   \
  acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).x, (B0).x, acc0, 0, 0, 0); \
  acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).x, (B1).x, acc1, 0, 0, 0); \
  acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).y, (B0).y, acc0, 0, 0, 0); \
  acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).y, (B1).y, acc1, 0, 0, 0); \
  acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).z, (B0).z, acc0, 0, 0, 0); \
  acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).z, (B1).z, acc1, 0, 0, 0); \
  acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).w, (B0).w, acc0, 0, 0, 0); \
  acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32((AV).w, (B1).w, acc1, 0, 0, 0);

template <int KS, int KL, bool STREAM>
__device__ __forceinline__ void consumer_step(const FwdArgs& a, int t, int tile, int wave, int lane, const float* Abuf,
                                              const float4* Wl, const float4 (&ws)[KS > 0 ? KS : 1][2], float bv0, float bv1) {
  constexpr int K0 = KS + KL;   // first streamed group
  const int nt0 = 2 * wave;
  const int lda = a.lda;
  const float* A = Abuf + ((t - 1) & 1) * BM * lda;
  const int KQ = (2 * a.H) / 8;
  const int h = lane >> 5, l31 = lane & 31;
  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; i++) acc0[i] = bv0, acc1[i] = bv1;   // C starts as the bias of the lane's column
  // (a tile of padding rows only -- the caller rounds its row count up for the GEMMs that follow -- is act(bias): nothing
  // to multiply)
  if ((long long)tile * BM < a.n) {
  const float* ab = A + l31 * lda + 4 * h;
  float4 aa[2];
  aa[0] = *reinterpret_cast<const float4*>(ab);
  aa[1] = aa[0];
  constexpr int NB = STREAM ? 3 : 1;
  float4 bb[NB][2];
  const float4* wb = a.wp + (long long)nt0 * 64 + lane;
  const long long wstep = (long long)a.NTp * 64;
  if (STREAM) {
    const int q1 = K0 < KQ ? K0 : KQ - 1, q2 = K0 + 1 < KQ ? K0 + 1 : KQ - 1;
    bb[0][0] = wb[q1 * wstep], bb[0][1] = wb[q1 * wstep + 64];
    bb[1 % NB][0] = wb[q2 * wstep], bb[1 % NB][1] = wb[q2 * wstep + 64];
  }
  // groups in registers
#pragma unroll
  for (int q = 0; q < ((a.dbg & 16) ? 0 : KS); q++) {
    const int qa = q + 1 < KQ ? q + 1 : KQ - 1;
    aa[(q + 1) % 2] = *reinterpret_cast<const float4*>(ab + 8 * qa);
    MFMA8(aa[q % 2], ws[q][0], ws[q][1])
  }
  // groups in LDS: Wl[k][wave][n-tile][lane].  A real loop (two groups per trip): unrolled, the compiler hoists every
  // LDS read to the top and spills
  if (KL > 0) {
    const float4* wl = Wl + wave * 128 + lane;
    float4 lb[2][2];
    lb[0][0] = wl[0], lb[0][1] = wl[64];
#pragma unroll 1
    for (int k0 = 0; k0 < KL; k0 += 2) {
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const int k = k0 + i;
        if (k < KL) {
          const int q = KS + k, qa = q + 1 < KQ ? q + 1 : KQ - 1, kn = k + 1 < KL ? k + 1 : KL - 1;
          lb[(i + 1) % 2][0] = wl[kn * (NCONS * 128)], lb[(i + 1) % 2][1] = wl[kn * (NCONS * 128) + 64];
          aa[(KS + i + 1) % 2] = *reinterpret_cast<const float4*>(ab + 8 * qa);
          MFMA8(aa[(KS + i) % 2], lb[i % 2][0], lb[i % 2][1])
        }
      }
    }
  }
  if (STREAM) {
    int q0 = K0;
    for (; q0 + 6 <= KQ; q0 += 6) {   // (loads past the last group re-read it: no branch, no drained queue)
#pragma unroll
      for (int i = 0; i < 6; i++) {
        const int q = q0 + i, qb = q + 2 < KQ ? q + 2 : KQ - 1, qa = q + 1 < KQ ? q + 1 : KQ - 1;
        bb[(i + 2) % NB][0] = wb[qb * wstep];
        bb[(i + 2) % NB][1] = wb[qb * wstep + 64];
        aa[(K0 + i + 1) % 2] = *reinterpret_cast<const float4*>(ab + 8 * qa);
        MFMA8(aa[(K0 + i) % 2], bb[i % NB][0], bb[i % NB][1])
      }
    }
#pragma unroll
    for (int i = 0; i < 5; i++) {   // the last (KQ - K0) % 6 groups
      const int q = q0 + i;
      if (q < KQ) {
        const int qb = q + 2 < KQ ? q + 2 : KQ - 1, qa = q + 1 < KQ ? q + 1 : KQ - 1;
        bb[(i + 2) % NB][0] = wb[qb * wstep];
        bb[(i + 2) % NB][1] = wb[qb * wstep + 64];
        aa[(K0 + i + 1) % 2] = *reinterpret_cast<const float4*>(ab + 8 * qa);
        MFMA8(aa[(K0 + i) % 2], bb[i % NB][0], bb[i % NB][1])
      }
    }
  }
  }
  // C/D of a 32x32 tile: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5): a store instruction writes
  // two 128-byte row segments.  Nothing waits for these stores (a consumer has no load in its queue behind them).
  const long long row0 = (long long)tile * BM;
  const int col0 = 32 * nt0 + l31, col1 = col0 + 32;
  if (a.y_bytes) {
    // 32-bit buffer addressing: one lane offset per tile, the row of each register as a scalar offset; a row beyond
    // n_pad is beyond the descriptor's range and is not stored
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(a.y, a.y_bytes);
    const unsigned base = ((unsigned)(row0 + 4 * h) * (unsigned)a.ldy + (unsigned)col0) * 4u;
    const unsigned v0 = col0 < a.out ? base : OOB, v1 = col1 < a.out ? base + 128u : OOB;
    const unsigned rowb = (unsigned)a.ldy * 4u;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int rl = (i & 3) + 8 * (i >> 2);
      float u0 = acc0[i], u1 = acc1[i];
      if (a.relu_out) u0 = fmaxf(u0, 0.f), u1 = fmaxf(u1, 0.f);
      if ((a.dbg & 8) && u0 != 12345.678f) continue;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, u0), ry, v0, rl * rowb, 0);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, u1), ry, v1, rl * rowb, 0);
    }
    return;
  }
  const bool whole = row0 + BM <= a.n_pad;
  float* y0 = a.y + (row0 + 4 * h) * a.ldy;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const int rl = (i & 3) + 8 * (i >> 2);
    float u0 = acc0[i], u1 = acc1[i];
    if (a.relu_out) u0 = fmaxf(u0, 0.f), u1 = fmaxf(u1, 0.f);
    if (whole || row0 + 4 * h + rl < a.n_pad) {
      if (col0 < a.out) y0[(long long)rl * a.ldy + col0] = u0;
      if (col1 < a.out) y0[(long long)rl * a.ldy + col1] = u1;
    }
  }
}
#undef MFMA8

// the operand tile itself, if the caller wants it (the backward's weight-gradient GEMM reads it): LDS -> cat, 8 rows per
// consumer wave
__device__ __forceinline__ void store_operand(const FwdArgs& a, int t, int tile, int wave, int lane, const float* Abuf) {
  if (!a.cat || (a.dbg & 64)) return;
  const int lda = a.lda;
  const float* A = Abuf + ((t - 1) & 1) * BM * lda;
  const long long row0 = (long long)tile * BM;
  const int q4 = (2 * a.H) >> 2;   // float4 per row
  if (a.cat_bytes) {
    // a row per trip, lanes across it: no division, the row as a scalar offset, rows beyond n_pad out of range
    const __amdgpu_buffer_rsrc_t rc = make_rsrc(a.cat, a.cat_bytes);
    const unsigned rowb = (unsigned)a.ldc * 4u;
    for (int c4 = lane; c4 < q4; c4 += 64) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int rl = 8 * wave + i;
        const float4 v = *reinterpret_cast<const float4*>(A + rl * lda + c4 * 4);
        const u32x4 u = {__builtin_bit_cast(unsigned, v.x), __builtin_bit_cast(unsigned, v.y), __builtin_bit_cast(unsigned, v.z),
                         __builtin_bit_cast(unsigned, v.w)};
        __builtin_amdgcn_raw_buffer_store_b128(u, rc, (unsigned)c4 * 16u, (unsigned)(row0 + rl) * rowb, 0);
      }
    }
    return;
  }
  for (int e = lane; e < 8 * q4; e += 64) {
    const int rl = 8 * wave + e / q4, c = (e % q4) * 4;
    if (row0 + rl < a.n_pad)
      *reinterpret_cast<float4*>(a.cat + (row0 + rl) * a.ldc + c) = *reinterpret_cast<const float4*>(A + rl * lda + c);
  }
}

// KS k-groups of W stationary in the consumers' registers, the next KL in LDS; STREAM: the layer has more, streamed from L2
template <int KS, int KL, bool STREAM>
__global__ __launch_bounds__(TBP) void k_sage_fwd_mfma(const FwdArgs a) {
  extern __shared__ float4 smem4[];
  float* Abuf = reinterpret_cast<float*>(smem4);               // [2][BM][lda]
  StagedT* Ibuf = reinterpret_cast<StagedT*>(Abuf + 2 * BM * a.lda);   // [3][BM][IDXW]
  float4* Wl = reinterpret_cast<float4*>(Ibuf + 4 * BM * IDXW);   // [KL][NCONS][2][64]  (16-byte aligned)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = gridDim.x, b = blockIdx.x;
  const int S = (a.n_tiles - b + G - 1) / G;   // this workgroup's tiles: b, b + G, ...
  // the longest row among this workgroup's tiles decides the producers' path (one look at the row pointers; uniform)
  __shared__ int s_maxdeg;
  if (tid == 0) s_maxdeg = 0;
  __syncthreads();
  if (wave >= NCONS) {
    const int j = tid - 64 * NCONS;
    int md = 0;
    for (int i = j; i < S * BM; i += PT) {
      const long long r = (long long)(b + (long long)(i / BM) * G) * BM + (i % BM);
      if (r < a.n) md = max(md, a.indptr[r + 1] - a.indptr[r]);
    }
    if (md > 0) atomicMax(&s_maxdeg, md);
    // staged rows start as "no self row, no edges" (the prologue steps read them)
    for (int i = j; i < 3 * BM * IDXW; i += PT) Ibuf[i] = (i % IDXW) == 0 ? (StagedT)0 : (StagedT)(uintptr_t)a.zero;
  }
  __syncthreads();
  if (wave >= NCONS) {
    const int j = tid - 64 * NCONS;
    const int md = s_maxdeg;
    if (a.dbg & 2048) __builtin_amdgcn_s_setprio(3);
    ProdState st;
    st.p_e0 = 0, st.p_deg = 0, st.p_sid = -1, st.i_sid = -1, st.i_deg = 0;
#pragma unroll
    for (int k = 0; k < NSL; k++) st.i_raw[k] = -1;
#pragma unroll
    for (int k = 0; k < 2; k++)
#pragma unroll
      for (int e = 0; e < 2 * (EC + 1); e++) st.r[k][e] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.H > 128 || md > 3 * EC || md > EMAX || (a.dbg & 4096)) producer_loop<0, false>(a, S, b, G, j, Abuf, Ibuf, st);
    else if (a.relu_in) {
      if (md <= EC) producer_loop<1, true>(a, S, b, G, j, Abuf, Ibuf, st);
      else if (md <= 2 * EC) producer_loop<2, true>(a, S, b, G, j, Abuf, Ibuf, st);
      else producer_loop<3, true>(a, S, b, G, j, Abuf, Ibuf, st);
    } else {
      if (md <= EC) producer_loop<1, false>(a, S, b, G, j, Abuf, Ibuf, st);
      else if (md <= 2 * EC) producer_loop<2, false>(a, S, b, G, j, Abuf, Ibuf, st);
      else producer_loop<3, false>(a, S, b, G, j, Abuf, Ibuf, st);
    }
  } else {
    if (!(a.dbg & 1024)) __builtin_amdgcn_s_setprio(2);
    const int nt0 = 2 * wave;
    const bool work = nt0 < a.NTp && !(a.dbg & 2);   // (narrow layers: fewer than 8 n-tiles)
    float4 ws[KS > 0 ? KS : 1][2];
    float bv0 = 0.f, bv1 = 0.f;
    {
      // (a wave without columns of its own reads n-tile 0: the loads stay unconditional)
      const float4* wb = a.wp + (long long)(work ? nt0 : 0) * 64 + lane;
      const long long wstep = (long long)a.NTp * 64;
#pragma unroll
      for (int q = 0; q < KS; q++) ws[q][0] = wb[q * wstep], ws[q][1] = wb[q * wstep + 64];
#pragma unroll
      for (int k = 0; k < KL; k++) {   // this wave's own slice of the LDS part (read back by this wave only)
        Wl[k * (NCONS * 128) + wave * 128 + lane] = wb[(KS + k) * wstep];
        Wl[k * (NCONS * 128) + wave * 128 + 64 + lane] = wb[(KS + k) * wstep + 64];
      }
      if (KS == 0) ws[0][0] = ws[0][1] = make_float4(0.f, 0.f, 0.f, 0.f);
      const int col0 = 32 * nt0 + (lane & 31), col1 = col0 + 32;
      if (a.bias && col0 < a.out) bv0 = a.bias[col0];
      if (a.bias && col1 < a.out) bv1 = a.bias[col1];
    }
    for (int t = -4; t <= S + 1; t++) {
      const unsigned s0 = (unsigned)__builtin_amdgcn_s_memrealtime();
      if (t >= 1 && t <= S) {
        if (work) consumer_step<KS, KL, STREAM>(a, t, b + (t - 1) * G, wave, lane, Abuf, Wl, ws, bv0, bv1);
        store_operand(a, t, b + (t - 1) * G, wave, lane, Abuf);
      }
      const unsigned s1 = (unsigned)__builtin_amdgcn_s_memrealtime();
      __syncthreads();
      if ((a.dbg & 64) && tid == 0 && t + 4 < 64) {
        unsigned* o = reinterpret_cast<unsigned*>(a.cat) + ((b * 64 + (t + 4)) * 2 + 0) * 4;
        o[0] = s0, o[1] = s1, o[2] = (unsigned)__builtin_amdgcn_s_memrealtime(), o[3] = S;
      }
    }
  }
}

inline int lda_for(int H) { return 2 * H + 4; }  // 2 H is a multiple of 8, so (2 H + 4) / 4 is odd
inline int ntp_for(int out) { return ((out + 31) / 32 + 1) & ~1; }
inline size_t lds_for(int H, int out, int KL) {
  return (size_t)2 * BM * lda_for(H) * sizeof(float) + (size_t)4 * BM * IDXW * sizeof(StagedT) + 64 +
         (size_t)KL * NCONS * 128 * sizeof(float4);
}
// how W is held: KS k-groups in the consumers' registers (8 registers each), KL in LDS (8 KB each), the rest streamed.
// in = 100 (25 groups): 19 + 6, nothing streamed.
inline void split_for(int KQ, int& KS, int& KL) {
  KS = KQ == 25 ? 19 : KQ == 24 ? 18 : KQ >= 16 ? 16 : 0;
  KL = (KQ == 24 || KQ == 25) ? 6 : 0;
}

template <int KS, int KL, bool STREAM>
int launch(const FwdArgs& a, unsigned grid, size_t lds, hipStream_t st) {
  static size_t attr_lds = 48 * 1024;   // (more dynamic LDS than the default limit must be asked for, once per size)
  if (lds > attr_lds) {
    if (hipFuncSetAttribute((const void*)k_sage_fwd_mfma<KS, KL, STREAM>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return CSL_E_HIP;
    attr_lds = lds;
  }
  hipLaunchKernelGGL((k_sage_fwd_mfma<KS, KL, STREAM>), dim3(grid), dim3(TBP), lds, st, a);
  return hipGetLastError() == hipSuccess ? CSL_OK : CSL_E_HIP;
}

}  // namespace

extern "C" {

int64_t csl_sage_fwd_mfma_scratch(int32_t H, int32_t out) {
  int KS, KL;
  split_for(2 * H / 8, KS, KL);
  if (H < 4 || H % 4 != 0 || out < 1 || out > 256 || lds_for(H, out, KL) > 160 * 1024 - 64) return CSL_E_INVALID;
  return (int64_t)(2 * H / 8) * ntp_for(out) * 64 * 4 + 4 * ZROW4;   // packed W + a zero row
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
  int KS, KL;
  split_for(2 * H / 8, KS, KL);
  const size_t lds = lds_for(H, out, KL);
  if (lds > 160 * 1024 - 64) return CSL_E_INVALID;
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
  hipLaunchKernelGGL(k_pack_w, dim3((unsigned)((nw + ZROW4 + 255) / 256)), dim3(256), 0, st, W, (long long)ldw, (int)out, KQ, NTp,
                     reinterpret_cast<float4*>(wpack));
  FwdArgs a;
  a.indptr = indptr, a.indices = indices, a.self_ids = self_ids, a.rowmap = rowmap;
  a.x = x, a.ldx = ldx, a.wp = reinterpret_cast<const float4*>(wpack), a.zero = wpack + nw * 4, a.bias = bias;
  a.cat = cat, a.ldc = ldc, a.y = y, a.ldy = ldy, a.n = n, a.n_pad = n_pad;
  a.H = H, a.out = out, a.NTp = NTp, a.relu_in = relu_in, a.relu_out = relu_out, a.lda = lda_for(H);
  a.n_tiles = (int)((n_pad + BM - 1) / BM);
  a.indptr_ld = (indptr && n > 0) ? indptr : reinterpret_cast<const int*>(wpack);
  a.self_ld = (self_ids && n > 0) ? self_ids : reinterpret_cast<const int*>(wpack);
  a.indices_ld = indices ? indices : reinterpret_cast<const int*>(wpack);
  a.rowmap_ld = rowmap ? rowmap : reinterpret_cast<const int*>(wpack);
  {
    const unsigned long long yb = (unsigned long long)n_pad * (unsigned long long)ldy * 4ull;
    const unsigned long long cb = cat ? (unsigned long long)n_pad * (unsigned long long)ldc * 4ull : 0ull;
    a.y_bytes = yb < OOB ? (unsigned)yb : 0u;
    a.cat_bytes = (cb > 0 && cb < OOB) ? (unsigned)cb : 0u;
  }
  a.dbg = dbg;
  const unsigned grid = (unsigned)(a.n_tiles < n_cu ? a.n_tiles : n_cu);
  if (KQ == 25) return launch<19, 6, false>(a, grid, lds, st);
  if (KQ == 24) return launch<18, 6, false>(a, grid, lds, st);
  if (KQ == 16) return launch<16, 0, false>(a, grid, lds, st);
  if (KQ > 16) return launch<16, 0, true>(a, grid, lds, st);
  return launch<0, 0, true>(a, grid, lds, st);
}

}  // extern "C"

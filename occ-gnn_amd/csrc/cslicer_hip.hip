// cslicer_hip.hip -- MI355X (gfx950) cslicer engine: kernels + C ABI.
//
// What the reference does sequentially per minibatch (cslicer/slicer.cpp:25-64,
// bipartite.cpp:3-17, util/duplicate.cpp:14-39) is restated here as ten
// data-parallel passes per layer over S minibatches ("streams") at once.
// Order-dependent semantics (first-occurrence dedup, consecutive-dedup lists,
// the single mt19937 stream) are recovered from each element's position in the
// reference's traversal order:
//
//   candidate position  c = i * (fanout+1) + slot     (i = index in frontier,
//                                                      slot 0 = the node itself)
//   first occurrence    = smallest c                   (candidates are hash-
//                                                      partitioned into buckets;
//                                                      each bucket is resolved in
//                                                      an LDS hash table, the
//                                                      DuplicateRemover mask)
//   stable lists        = exclusive scans of flags in c / i order
//   rng word of (i,j)   = base + fanout * #{i' < i : deg(i') >= fanout} + j
//
// HBM layout (all device memory, per engine):
//   off32    u32[N+1]    row offsets (E < 2^32)          one gather per node (else rowinfo u64[N]:
//                                                       (row offset << 24) | degree)
//   indices  u32[E]      CSR neighbours (ids < 2^31)
//   wl       u8[N]       owner part (absent => v % P)
//   rng ring u32[2^k]    mt19937 outputs, absolute position & mask
//   per stream: frontier / candidate / flag scratch, bucket queue of
//   {node id, position} pairs, tile counters; result arena (int64 lists in
//   BiPartite layout, see include/cslicer_hip.h).  Nothing per stream scales
//   with N: the dedup state lives in LDS.
//
// No CPU fallback exists in this file: every entry point either runs the HIP
// kernels or returns an error.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <deque>
#include <vector>

#include "cslicer_hip.h"

namespace {

constexpr uint32_t UNSET = 0xFFFFFFFFu;

constexpr int TN = 256;  // frontier nodes per tile == threads per block
// launch bounds of the 256-thread kernels: the compiler is asked for 8 waves per SIMD.  Left alone it spends ~90
// SGPRs on the by-value LArgs and the hardware then admits 7 blocks per CU, not 8 (MI355X_MICROARCH.md, residency);
// with the bound it stays at 78: k_sample 193 -> 185 us, k_emit 183 -> 175 us.  -DCSL_OCC7 restores the old bound.
#ifdef CSL_OCC7
#define CSL_LB256 __launch_bounds__(256)
#else
#define CSL_LB256 __launch_bounds__(256, 8)
#endif
constexpr int CSL_MAX_SETS = 4;  // rounds in flight at once (HIP streams / scratch sets)
constexpr int NW = TN / 64;
// tuning constants (overridable with -D for sweeps; defaults measured best on MI355X, round 1)
#ifndef CSL_TPB
#define CSL_TPB 4
#endif
#ifndef CSL_HLOG
#define CSL_HLOG 12
#endif
#ifndef CSL_QMEAN
#define CSL_QMEAN 2048
#endif
#ifndef CSL_SCT
#define CSL_SCT 4096  // (8192 until the kernels ran at 8 blocks per CU: profiles/SWEEP_r2.md)
#endif
#ifndef CSL_SU
#define CSL_SU 1
#endif
constexpr int TPB = CSL_TPB;      // frontier tiles per k_sample block on large layers
constexpr int HLOG = CSL_HLOG;    // LDS hash table: 2^HLOG slots x 12 B = 48 KiB per block
constexpr int HCAP = 1 << HLOG;
constexpr int QMEAN = CSL_QMEAN;  // target candidates per bucket (table load <= 0.5)
constexpr int SCT = CSL_SCT;      // candidates per k_scatter block
constexpr int DEG_BITS = 24;
constexpr uint32_t DEG_MASK = (1u << DEG_BITS) - 1;

// ---- tile counter kinds -------------------------------------------------
// kinds 0..1 are produced by k_degree, 2.. by k_sample / k_flag
constexpr int K_NEED = 0;   // nodes with deg >= fanout (rng consumers)
constexpr int K_EDGES = 1;  // sum of min(deg, fanout)
constexpr int K_NEWF = 2;   // first occurrences -> next frontier
__host__ __device__ constexpr int K_IN(int P, int g) { return 3 + g; }
__host__ __device__ constexpr int K_OUT(int P, int g) { return 3 + P + g; }
__host__ __device__ constexpr int K_OWNED(int P, int g) { return 3 + 2 * P + g; }
__host__ __device__ constexpr int K_SELF(int P, int g) { return 3 + 3 * P + g; }
__host__ __device__ constexpr int K_TO(int P, int g) { return 3 + 4 * P + g; }
__host__ __device__ constexpr int K_FROM(int P, int g) { return 3 + 5 * P + g; }
// graph mode only: edges per source part (CSR sizes) and boundary pairs (sender g, receiver p)
__host__ __device__ constexpr int K_ECNT(int P, int g) { return 3 + 6 * P + g; }
__host__ __device__ constexpr int K_PAIR(int P, int g, int p) { return 3 + 7 * P + g * P + p; }
__host__ __device__ constexpr int NKINDS(int P, bool graph) { return graph ? 3 + 7 * P + P * P : 3 + 6 * P; }
constexpr int MAXK = 3 + 7 * CSL_MAX_PARTS + CSL_MAX_PARTS * CSL_MAX_PARTS;

enum { KN_SEEDS = 0, KN_DEGREE, KN_SCAN_A, KN_SAMPLE, KN_SCAN_Q, KN_SCATTER, KN_BUCKET, KN_COUNT, KN_SCAN_B,
       KN_EMIT, KN_SELFIN, KN_MT, KN_EDGES, KN_DUPSEEDS, KN_SELFIN_DEGREE };
const char* const kKernelNames[CSL_NUM_KERNELS] = {"k_seeds",   "k_degree", "k_scan_need", "k_sample",
                                                   "k_scan_buckets", "k_scatter", "k_bucket", "k_count",
                                                   "k_scan_lists", "k_emit",  "k_selfin",  "k_mt19937_fill",
                                                   "k_graph", "k_dupseeds", "k_selfin_degree"};

// Everything a layer's kernels need; passed by value.
struct LArgs {
  // graph
  const unsigned long long* rowinfo;  // u64[N]: (row offset << 24) | degree        (graphs with E >= 2^32)
  const uint32_t* off32;              // u32[N+1]: row offsets (E < 2^32): half the table, more of it in L2
  const uint32_t* indices;
  const uint8_t* wl;
  uint32_t N, P;
  // rng
  const uint32_t* ring;
  unsigned long long ring_mask, gen_lo, gen_hi;
  unsigned long long* rngpos;   // [S] running position
  unsigned long long* rngbase;  // [S] base of the current layer
  unsigned long long* acc;      // [S][2] running totals since creation: sampled edges, minibatches
  // per-stream scratch (stride = elements per stream)
  const uint32_t* fr_in;        // [S][fr_in_stride]
  uint32_t* fr_out;             // [S][fr_out_stride]
  size_t fr_in_stride, fr_out_stride;
  uint32_t fr_out_cap;
  unsigned long long* ninfo;    // [S][fcap]
  uint32_t* hasedge;            // [S][fcap]
  uint32_t* selfpos;            // [S][fcap]
  uint32_t* firstpos;           // [S][fcap] position of the node's first edge occurrence, or UNSET
  size_t fcap;
  uint32_t* cand;               // [S][ccap]
  uint8_t* cflag;               // [S][ccap]
  uint32_t* crank;              // [S][ccap] in-node rank of first-edge candidates
  uint2* queue;                 // [S][ccap] bucketed {node id, position | self marker}
  size_t ccap;
  uint32_t* nbk;                // [S] buckets in use this layer
  uint32_t* bcnt;               // [S][nbmax+1] bucket sizes -> offsets
  uint32_t* bcur;               // [S][nbmax]   scatter cursors
  uint32_t nbmax;
  uint32_t* tcnt;               // [S][nk][tmax]
  uint32_t tmax, nk;
  uint32_t* fsize;              // [S][CSL_MAX_LAYERS+1] frontier sizes
  csl_sample_meta* meta;        // [S] (slot already applied)
  // result arena of this layer (slot applied): int32 [S][arena_stride] (ids and local
  // indices are < 2^31; host exports widen to the reference's `long`)
  int* arena;
  size_t arena_stride;
  size_t list_base[CSL_NUM_LISTS];  // element offset of each kind inside a stream's arena
  uint32_t layer, fanout, W;
  unsigned long long wmagic;  // ceil(2^40 / W): exact c / W for c < 2^31, W < 512
  uint32_t S;
  uint32_t tpb;               // frontier tiles per k_sample block
  uint32_t* boff;             // [S][nbmax+1] bucket offsets inside the queue (k_scatter's block 0 -> k_bucket)
  uint32_t* ticket;           // [S][2] last-block-done tickets (k_degree, k_count)
  uint32_t pmagic;            // floor(2^32 / P) (P >= 2), see mod_parts()
  uint32_t pmask;             // bit g: the lists of part g are written (csl_config.part_mask; all ones = every part)
  uint32_t last;              // 1 on the final layer (no next frontier to prepare)
  // repeated seed ids (layer 0 only; bipartite.cpp:3-17 on a batch with duplicates): see k_dupseeds
  uint32_t* dupflag;          // [S] != 0: the stream's minibatch holds a seed id more than once
  uint32_t* seedrep;          // [S][fcap0] index of the first seed with the same id
  uint32_t* dupfirst;         // [S][fcap0*P] scratch of k_dupseeds
  uint32_t* dupout;           // [S][fcap0*P] scratch of k_dupseeds
  size_t fcap0;
  unsigned long long* rngend; // [S] the stream's mt19937 position after this round (per scratch set)
  // CSL_FLAG_KEEP_CANDIDATES: raw neighbour_sample stream of this layer, [S][ccap] (debug export)
  uint32_t* candk;
  // CSL_MODE_GRAPH extras
  uint32_t graph;             // 0 strict, 1 graph
  uint8_t* ecnt;              // [S][fcap*P] edges of node i whose source is owned by g
  uint32_t* srcpos;           // [S][ccap]   position of the first occurrence of an edge's source
  uint32_t* tcur;             // [S][ccap]   CSL_FLAG_TRANSPOSE, this layer: edges (+ self entry) per in node, then fill cursors; else null
};

constexpr uint32_t SELF_BIT = 0x80000000u;

// v % P for the runtime constant P (pyfrontend.cpp:57 workload_map[j] = j % 4) without the ~25-instruction
// software division: q = floor(v * floor(2^32 / P) / 2^32) is the quotient or one short of it.
__device__ __forceinline__ uint32_t mod_parts(const LArgs& a, uint32_t v) {
  uint32_t r = v - __umulhi(v, a.pmagic) * a.P;
  if (r >= a.P) r -= a.P;
  if (r >= a.P) r -= a.P;
  return r;
}
__device__ __forceinline__ uint32_t owner(const LArgs& a, uint32_t v) {
  return a.wl ? (uint32_t)a.wl[v] : mod_parts(a, v);
}
// row offset and degree of node v, in ninfo's packed form (the gather of slicer.cpp:8-9).  With 32-bit
// offsets the lookup table is 4 B per node instead of 8: the random gathers of a layer touch the same
// number of entries but a line holds twice as many and more of the table stays in the 4 MiB L2s.
__device__ __forceinline__ unsigned long long row_lookup(const LArgs& a, uint32_t v) {
  if (a.off32) {
    const uint32_t lo = a.off32[v], hi = a.off32[v + 1];
    return ((unsigned long long)lo << 24) | (unsigned long long)(hi - lo);
  }
  return a.rowinfo[v];
}
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ uint32_t div_w(const LArgs& a, uint32_t c) {
  return (uint32_t)(((unsigned long long)c * a.wmagic) >> 40);
}
// XCD-aware block mapping.  Workgroups are dealt round-robin over the 8 XCDs
// (block id % 8), each XCD with its own L2.  All blocks of one stream are given
// the same id % 8, in consecutive order, so a stream's scratch (candidates,
// flags, bucket queue) is touched through ONE L2: scattered narrow stores merge
// into whole lines there before they reach HBM.  Speed only, never correctness.
// grid.x = 8 * ceil(S/8) * per_stream.
__device__ __forceinline__ bool xcd_block_at(const LArgs& a, uint32_t id, uint32_t grid, uint32_t& x, uint32_t& s) {
  const uint32_t groups = (a.S + 7u) >> 3;
  const uint32_t per_stream = grid / (8u * groups);
  const uint32_t j = id >> 3;
  const uint32_t sl = j / per_stream;
  x = j - sl * per_stream;
  s = sl * 8u + (id & 7u);
  return s < a.S;
}
__device__ __forceinline__ bool xcd_block(const LArgs& a, uint32_t& x, uint32_t& s) {
  return xcd_block_at(a, blockIdx.x, gridDim.x, x, s);
}
// Last-block-done: every block of a stream's share of a launch publishes its tile counters, then takes a ticket;
// the block that draws the last one runs the stream's scan in the same launch instead of a one-block-per-stream
// kernel of its own.  Hand-off form of MI355X_MICROARCH.md (inter-workgroup visibility, first table row): the
// counters are stored AND loaded `sc1` (write-through / L1-bypassing: __hip_atomic_store/load, relaxed, agent
// scope), every storing wave drains its stores (`s_waitcnt vmcnt(0)`) before the workgroup barrier behind which ONE
// lane adds to the stream's ticket counter (agent-scope atomic); the workgroup whose add came last, told by the
// value the add returned, loads -- its other waves after a workgroup barrier the adding wave joins.  No agent fence:
// a `__threadfence()` here writes back the whole XCD L2 from every thread (k_count: 30 us -> 670 us).
__device__ __forceinline__ void st_sc1(uint32_t* p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t ld_sc1(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool last_block_of_stream(uint32_t* ticket, uint32_t expected) {
  __shared__ uint32_t s_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's counter stores have left the CU
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = t == expected - 1u;
    if (t == expected - 1u) {
      st_sc1(ticket, 0u);  // ready for the next launch (no other block touches it any more)
      // belt and braces: the scan only issues sc1 loads, which do not look at this CU's L1; the acquire
      // (buffer_inv sc1, one wave of one block per stream) makes a plain load correct as well
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  return s_last != 0u;
}
__device__ __forceinline__ unsigned long long lt_mask() {
  return (1ull << lane_id()) - 1ull;
}
// bucket of a node id: multiplicative hash, uniform over [0, nb)
__device__ __forceinline__ uint32_t bucket_of(uint32_t v, uint32_t nb) {
  return __umulhi(v * 0x9E3779B1u, nb);
}
// which pass of an oversized bucket resolves an id (k_bucket): a third mix of the id
__device__ __forceinline__ uint32_t pass_of(uint32_t v, uint32_t npass) {
  uint32_t x = v * 0x27D4EB2Fu;
  x ^= x >> 13;
  return __umulhi(x * 0x165667B1u, npass);
}
// slot inside a bucket's LDS table: independent mix of the same id
__device__ __forceinline__ uint32_t slot_of(uint32_t v) {
  uint32_t x = v * 0x85EBCA6Bu;
  x ^= x >> 15;
  x *= 0xC2B2AE35u;
  return x >> (32 - HLOG);
}

// ---- scan_body: exclusive scan of tile counters kinds [k_lo, k_hi) of one stream, then the per-layer bookkeeping
// that depends on the totals.  Run by the LAST block of the stream in k_degree (PHASE 0: rng consumers, rng base,
// bucket geometry) and in k_count (PHASE 1: list offsets, next frontier size): see last_block_of_stream().
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t x, uint32_t& total) {
  uint32_t incl = x;
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t y = __shfl_up(incl, o);
    if ((int)lane_id() >= o) incl += y;
  }
  total = __shfl(incl, 63);
  return incl - x;
}

// Inclusive wave64 prefix sum on the VALU's DPP path (no LDS traffic, unlike __shfl_up): Hillis-Steele
// inside each row of 16 lanes (row_shr 1,2,4,8), then lane 15 of rows 0/2 into rows 1/3 (row_bcast15),
// then lane 31 into the upper half (row_bcast31) -- the sequence LLVM's atomic optimizer emits on GFX9.
// Packed counters (several small fields in one word) scan in one go as long as no field overflows.
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t x) {
  int v = (int)x;
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
  return (uint32_t)v;
}
// bit g of x (g < 4) -> byte g
__device__ __forceinline__ uint32_t spread4(uint32_t x) { return ((x & 0xFu) * 0x00204081u) & 0x01010101u; }

constexpr int SCAN_CH = 12;    // tiles a lane keeps in registers (64 x 12 tiles = 196 k frontier nodes)

template <int PHASE>
__device__ __forceinline__ void scan_body(const LArgs& a, const int s) {
  // (every word another block of this launch wrote is read `sc1`: see last_block_of_stream)
  const uint32_t F = ld_sc1(&a.fsize[s * (CSL_MAX_LAYERS + 1) + a.layer]);
  const uint32_t ntiles = (F + TN - 1) / TN;
  const int k_lo = PHASE == 0 ? 0 : 2;
  const int k_hi = PHASE == 0 ? 2 : (int)a.nk;
  __shared__ uint32_t s_tot[MAXK];
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const uint32_t chunk = (ntiles + 63) / 64;  // consecutive tiles per lane
  for (int k = k_lo + w; k < k_hi; k += nw) {
    uint32_t* p = a.tcnt + ((size_t)s * a.nk + k) * a.tmax;
    if (chunk <= (uint32_t)SCAN_CH) {
      // all of the lane's counters are requested at once; one wave scan of the lane sums
      const uint32_t t0 = lane_id() * chunk;
      uint32_t x[SCAN_CH], sum = 0;
#pragma unroll
      for (int j = 0; j < SCAN_CH; j++) {
        x[j] = ((uint32_t)j < chunk && t0 + j < ntiles) ? ld_sc1(&p[t0 + j]) : 0u;
        sum += x[j];
      }
      uint32_t tot;
      uint32_t run = wave_excl_scan(sum, tot);
#pragma unroll
      for (int j = 0; j < SCAN_CH; j++) {
        if ((uint32_t)j < chunk && t0 + j < ntiles) p[t0 + j] = run;
        run += x[j];
      }
      if (lane_id() == 0) s_tot[k] = tot;
    } else {
      uint32_t run = 0;
      for (uint32_t t0 = 0; t0 < ntiles; t0 += 64) {
        const uint32_t t = t0 + lane_id();
        const uint32_t x = t < ntiles ? ld_sc1(&p[t]) : 0;
        uint32_t tot;
        const uint32_t ex = wave_excl_scan(x, tot);
        if (t < ntiles) p[t] = run + ex;
        run += tot;
      }
      if (lane_id() == 0) s_tot[k] = run;
    }
  }
  if (PHASE == 0) {
    // bucket geometry of this layer: enough buckets that a bucket's distinct
    // ids stay well below the LDS table; bucket counters start at zero
    const unsigned long long C = (unsigned long long)F * a.W;
    uint32_t nb = (uint32_t)((C + QMEAN - 1) / QMEAN);
    if (nb < 1) nb = 1;
    if (nb > a.nbmax) nb = a.nbmax;
    if (threadIdx.x == 0) a.nbk[s] = F ? nb : 0;
    for (uint32_t b = threadIdx.x; b <= a.nbmax; b += blockDim.x) {
      a.bcnt[(size_t)s * (a.nbmax + 1) + b] = 0;
      if (b < a.nbmax) a.bcur[(size_t)s * a.nbmax + b] = 0;  // k_scatter's cursors count from the bucket's offset
    }
  }
  __syncthreads();
  csl_layer_meta& m = a.meta[s].layer[a.layer];
  if (PHASE == 0) {
    if (threadIdx.x == 0) {
      const unsigned long long base = a.rngpos[s];
      const unsigned long long draws = (unsigned long long)s_tot[K_NEED] * a.fanout;
      a.rngbase[s] = base;
      a.rngpos[s] = base + draws;
      a.rngend[s] = base + draws;  // per scratch set: what the round's position snapshot copies
      if (a.layer == 0) a.meta[s].rng_begin = base;
      a.meta[s].rng_end = base + draws;
      a.acc[2 * s] += s_tot[K_EDGES];
      if (a.layer == 0 && F) a.acc[2 * s + 1] += 1;
      m.frontier = F;
      m.draws = (uint32_t)draws;
      m.sampled_edges = s_tot[K_EDGES];
    }
  } else {
    // the layer's offset tables: one thread per list kind (and per part for the boundary pairs), not one for all
    const int P = (int)a.P;
    const uint32_t t = threadIdx.x;
    auto pair_sum = [&](int g, bool from) -> uint32_t {
      uint32_t x = 0;
      for (int p = 0; p < P; p++) x += from ? s_tot[K_PAIR(P, g, p)] : s_tot[K_PAIR(P, p, g)];
      return x;
    };
    auto tot_of = [&](int kind, int g) -> uint32_t {
      switch (kind) {
        case CSL_IN_NODES: return s_tot[K_IN(P, g)];
        case CSL_OUT_NODES: return s_tot[K_OUT(P, g)];
        case CSL_OWNED_OUT_NODES: return s_tot[K_OWNED(P, g)];
        case CSL_SELF_IDS_IN:
        case CSL_SELF_IDS_OUT: return s_tot[K_SELF(P, g)];
        case CSL_TO_IDS: return a.graph ? pair_sum(g, false) : s_tot[K_TO(P, g)];
        case CSL_FROM_IDS: return a.graph ? pair_sum(g, true) : s_tot[K_FROM(P, g)];
        case CSL_INDPTR: return a.graph ? (F ? s_tot[K_OUT(P, g)] + 1 : 0u) : 0u;
        case CSL_INDICES: return a.graph ? s_tot[K_ECNT(P, g)] : 0u;
        case CSL_T_INDPTR: return a.tcur ? (F ? s_tot[K_IN(P, g)] + 1 : 0u) : 0u;
        case CSL_T_INDICES: return a.tcur ? s_tot[K_ECNT(P, g)] + s_tot[K_SELF(P, g)] : 0u;
        default: return a.graph ? s_tot[K_OWNED(P, g)] : 0u;  // CSL_OWNED_DEGREE
      }
    };
    if (t < (uint32_t)CSL_NUM_LISTS) {
      uint32_t run = 0;
      for (int g = 0; g < P; g++) {
        m.off[t][g] = run;
        run += tot_of((int)t, g);
      }
      for (int g = P; g <= CSL_MAX_PARTS; g++) m.off[t][g] = run;
    } else if (t == 16) {
      uint32_t nf = s_tot[K_NEWF];
      if (nf > a.fr_out_cap) {
        atomicOr(&a.meta[s].error, (uint32_t)CSL_ERR_FRONTIER_CAP);
        nf = a.fr_out_cap;
      }
      m.next_frontier = nf;
      a.fsize[s * (CSL_MAX_LAYERS + 1) + a.layer + 1] = nf;
    } else if (t == 17) {
      for (int g = 0; g < CSL_MAX_PARTS; g++)
        m.indptr_len[g] = g < P ? tot_of(a.graph ? CSL_INDPTR : CSL_OUT_NODES, g) : 0u;
    } else if (a.graph && t >= 32 && t < 32u + (uint32_t)P) {
      const int g = (int)t - 32;
      uint32_t fr = 0, to = 0;
      for (int p = 0; p < P; p++) {
        m.pair_off[0][g][p] = fr;  // (g -> p) inside slice g's from_ids
        m.pair_off[1][g][p] = to;  // (p -> g) inside slice g's to_ids
        fr += s_tot[K_PAIR(P, g, p)];
        to += s_tot[K_PAIR(P, p, g)];
      }
      for (int p = P; p <= CSL_MAX_PARTS; p++) {
        m.pair_off[0][g][p] = fr;
        m.pair_off[1][g][p] = to;
      }
    }
  }
}

// ---- k_degree: Slicer::get_sample's copy of the batch into `in` (slicer.cpp:70-74; layer 0 only) and the first
// half of Slicer::neighbour_sample (slicer.cpp:8-9): row offset and degree of every frontier node; counts rng
// consumers and sampled edges per tile; the stream's last block then runs scan_body<0>.
struct BatchDesc {
  long long offset;  // into the node array
  int count;
  int pad;
};

__device__ __forceinline__ void degree_body(const LArgs& a, const uint32_t bx, const uint32_t s, const uint32_t per_stream,
                                            const long long* __restrict__ nodes, const BatchDesc* __restrict__ desc) {
  // one wave per tile, four consecutive nodes per lane: no LDS, no barrier before the hand-over
  uint32_t F;
  long long noff = 0;
  if (a.layer == 0) {
    const BatchDesc d = desc[s];
    F = (uint32_t)d.count;
    noff = d.offset;
    if (bx == 0 && threadIdx.x == 0) {
      st_sc1(&a.fsize[s * (CSL_MAX_LAYERS + 1)], F);  // read by the stream's last block in this launch
      for (uint32_t l = 1; l <= CSL_MAX_LAYERS; l++) a.fsize[s * (CSL_MAX_LAYERS + 1) + l] = 0;
      a.dupflag[s] = 0;  // set by k_bucket of layer 0, read by k_dupseeds: both later launches on this HIP stream
      // (meta[s].error was zeroed by a memset on the stream before this launch: a reset in here would race
      // with the other blocks' atomicOr of CSL_ERR_SEED_RANGE)
      a.meta[s].n_seeds = F;
    }
  } else {
    F = a.fsize[s * (CSL_MAX_LAYERS + 1) + a.layer];
  }
  const uint32_t tile = bx * NW + (threadIdx.x >> 6);
  if (tile * TN < F) {
    const uint32_t i0 = tile * TN + lane_id() * 4;
    unsigned long long ri[4] = {0, 0, 0, 0};
    if (a.layer == 0) {
      uint32_t v[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        v[j] = UNSET;
        if (i0 + j < F) {
          long long id = nodes[noff + i0 + j];
          if (id < 0 || id >= (long long)a.N) {
            atomicOr(&a.meta[s].error, (uint32_t)CSL_ERR_SEED_RANGE);
            id = 0;
          }
          v[j] = (uint32_t)id;
        }
      }
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (v[j] != UNSET) ri[j] = row_lookup(a, v[j]);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (v[j] != UNSET) {
          const_cast<uint32_t*>(a.fr_in)[s * a.fr_in_stride + i0 + j] = v[j];
          a.ninfo[s * a.fcap + i0 + j] = ri[j];
        }
      }
    } else {
      // gathered by the previous layer's k_emit
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (i0 + j < F) ri[j] = a.ninfo[s * a.fcap + i0 + j];
    }
    uint32_t need = 0, ne = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (i0 + j < F) {
        const uint32_t deg = (uint32_t)(ri[j] & DEG_MASK);
        need += deg >= a.fanout;
        ne += deg < a.fanout ? deg : a.fanout;
      }
    }
    for (int o = 32; o > 0; o >>= 1) {
      need += __shfl_down(need, o);
      ne += __shfl_down(ne, o);
    }
    if (lane_id() == 0) {
      st_sc1(&a.tcnt[((size_t)s * a.nk + K_NEED) * a.tmax + tile], need);
      st_sc1(&a.tcnt[((size_t)s * a.nk + K_EDGES) * a.tmax + tile], ne);
    }
  }
  // blocks without a tile leave at once; block 0 always stays (an empty minibatch still needs its bookkeeping)
  uint32_t nb_act = ((F + TN - 1) / TN + NW - 1) / NW;
  if (nb_act < 1) nb_act = 1;
  if (nb_act > per_stream) nb_act = per_stream;
  if (bx >= nb_act) return;
  if (last_block_of_stream(a.ticket + 2 * s, nb_act)) scan_body<0>(a, (int)s);
}

__global__ CSL_LB256 void k_degree(LArgs a, const long long* __restrict__ nodes,
                                               const BatchDesc* __restrict__ desc) {
  uint32_t bx, s;
  if (!xcd_block(a, bx, s)) return;
  degree_body(a, bx, s, gridDim.x / (8u * ((a.S + 7u) >> 3)), nodes, desc);
}

// mt19937's output tempering (applied by k_mt19937_temper)
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

// ---- k_sample: second half of neighbour_sample (slicer.cpp:10-21) + the
// bookkeeping side of the slice_layer inner loop (slicer.cpp:31-44): writes the
// candidate stream, ORs the owner part of each sampled neighbour into the
// node's part mask, counts the per-node list memberships and the size of each
// dedup bucket.  A block walks TPB consecutive tiles so its bucket histogram
// (LDS) is flushed once per 1024 nodes.
__global__ CSL_LB256 void k_sample(LArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t s_bh[];  // [nb] bucket histogram
  uint32_t bx, s;
  if (!xcd_block(a, bx, s)) return;
  const uint32_t F = a.fsize[s * (CSL_MAX_LAYERS + 1) + a.layer];
  if (bx * a.tpb * TN >= F) return;
  __shared__ uint32_t s_v[TN];
  __shared__ unsigned long long s_ri[TN];
  __shared__ uint32_t s_rng[TN];
  __shared__ uint32_t s_hb[TN];
  __shared__ uint8_t s_to[TN];  // owner part of the tile's nodes
  __shared__ uint32_t s_wn[NW];
  __shared__ uint32_t s_cnt[5 * CSL_MAX_PARTS];
  __shared__ uint32_t s_ec[TN * CSL_MAX_PARTS];                                 // graph: edges per (node, source part)
  __shared__ uint32_t s_gcnt[CSL_MAX_PARTS + CSL_MAX_PARTS * CSL_MAX_PARTS];    // graph: ECNT[g], PAIR[g][p]
  const uint32_t n = threadIdx.x;
  const uint32_t f = a.fanout, W = a.W;
  const uint32_t P = a.P;
  const uint32_t nb = a.nbk[s];
  for (uint32_t b = n; b < nb; b += TN) s_bh[b] = 0;
  const unsigned long long rbase = a.rngbase[s];
  for (uint32_t sub = 0; sub < a.tpb; sub++) {
    const uint32_t tile = bx * a.tpb + sub;
    if (tile * TN >= F) break;
    const uint32_t i = tile * TN + n;
    // phase 1: stage the tile's nodes, rank the rng consumers
    uint32_t v = 0, need = 0;
    unsigned long long ri = 0;
    if (i < F) {
      v = a.fr_in[s * a.fr_in_stride + i];
      ri = a.ninfo[s * a.fcap + i];
      need = (uint32_t)(ri & DEG_MASK) >= f;
    }
    const unsigned long long bm = __ballot(need);
    __syncthreads();  // previous sub-tile done with the LDS arrays
    if (lane_id() == 0) s_wn[n >> 6] = __popcll(bm);
    if (n < 5 * CSL_MAX_PARTS) s_cnt[n] = 0;
    s_v[n] = v;
    s_ri[n] = ri;
    s_hb[n] = 0;
    s_to[n] = i < F ? (uint8_t)owner(a, v) : 0;
    if (a.graph) {
      for (uint32_t k = n; k < TN * P; k += TN) s_ec[k] = 0;
      if (n < CSL_MAX_PARTS + CSL_MAX_PARTS * CSL_MAX_PARTS) s_gcnt[n] = 0;
    }
    __syncthreads();
    {
      uint32_t r = __popcll(bm & lt_mask());
      for (uint32_t w = 0; w < (n >> 6); w++) r += s_wn[w];
      const uint32_t tb = a.tcnt[((size_t)s * a.nk + K_NEED) * a.tmax + tile];
      s_rng[n] = need ? (tb + r) * f : UNSET;
    }
    __syncthreads();
    // phase 2: candidates, coalesced over c.  Each thread keeps SU candidates in
    // flight: all their rng words are requested, then all their neighbour ids,
    // before any is consumed (the loads are dependent pairs of HBM round trips).
    const uint32_t nodes_here = (F - tile * TN) < (uint32_t)TN ? (F - tile * TN) : (uint32_t)TN;
    const uint32_t ncand = nodes_here * W;
    const size_t cbase = (size_t)s * a.ccap + (size_t)tile * TN * W;
    constexpr int SU = CSL_SU;
    const uint32_t dq = TN / W, dr = TN - dq * W;  // k += TN  =>  node += dq, slot += dr (+carry)
    uint32_t nn = n / W, slot = n - nn * W;
    for (uint32_t k0 = n; k0 < ncand; k0 += TN * SU) {
      uint32_t nnu[SU], slu[SU], vvu[SU], degu[SU], val[SU], rnd[SU];
      unsigned long long addr[SU], rpos[SU];
      bool live[SU], gat[SU], rq[SU];
#pragma unroll
      for (int u = 0; u < SU; u++) {
        live[u] = k0 + u * TN < ncand;
        nnu[u] = nn;
        slu[u] = slot;
        nn += dq;
        slot += dr;
        if (slot >= W) {
          slot -= W;
          nn++;
        }
        val[u] = UNSET;
        gat[u] = false;
        rq[u] = false;
        addr[u] = 0;
        rpos[u] = 0;
        vvu[u] = 0;
        degu[u] = 1;
        rnd[u] = 0;
        if (live[u]) {
          vvu[u] = s_v[nnu[u]];
          if (slu[u] == 0) {
            val[u] = vvu[u];
          } else {
            const unsigned long long r2 = s_ri[nnu[u]];
            const uint32_t deg = (uint32_t)(r2 & DEG_MASK);
            const uint32_t j = slu[u] - 1;
            degu[u] = deg;
            addr[u] = r2 >> DEG_BITS;
            if (deg < f) {
              gat[u] = j < deg;
              addr[u] += j;
            } else {
              gat[u] = true;
              rq[u] = true;
              rpos[u] = rbase + s_rng[nnu[u]] + j;
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < SU; u++) {
        if (rq[u]) {
          if (rpos[u] >= a.gen_lo && rpos[u] < a.gen_hi) {
            rnd[u] = a.ring[rpos[u] & a.ring_mask];
          } else {
            atomicOr(&a.meta[s].error, (uint32_t)CSL_ERR_RNG_WINDOW);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < SU; u++) {
        // (non-temporal loads here were measured 5 % slower: picks of one row share lines)
        if (gat[u]) val[u] = a.indices[addr[u] + (rq[u] ? rnd[u] % degu[u] : 0u)];
      }
#pragma unroll
      for (int u = 0; u < SU; u++) {
        if (live[u]) {
          // The flag byte every candidate gets here is what k_bucket would find in the common case: a self entry
          // new to the frontier, an edge candidate the first occurrence of its node (in the frontier and among its
          // slice's in-nodes) and not a frontier node itself.  k_bucket then only stores the exceptions (a fifth of
          // the entries) instead of one scattered byte per first occurrence (most entries): its evaluate phase was
          // bound by exactly those stores.  bit0 new-frontier, bit1 first-in-node, bits 2-4 owner part.
          uint32_t fl = 0;
          if (slu[u] == 0) {
            atomicAdd(&s_bh[bucket_of(val[u], nb)], 1u);
            fl = (a.graph ? 3u : 1u) | ((uint32_t)s_to[nnu[u]] << 2);
          } else if (val[u] != UNSET) {
            if (val[u] == vvu[u]) {
              // a sampled self loop only re-adds the self edge (slicer.cpp:33-35,
              // bipartite.h:34): it is neither an edge nor new to the frontier
              if (a.candk) a.candk[cbase + k0 + u * TN] = val[u];  // (the raw stream keeps it)
              val[u] = UNSET;
            } else {
              const uint32_t og = owner(a, val[u]);
              atomicOr(&s_hb[nnu[u]], 1u << og);
              if (a.graph) atomicAdd(&s_ec[nnu[u] * P + og], 1u);
              atomicAdd(&s_bh[bucket_of(val[u], nb)], 1u);
              fl = 3u | (og << 2);
              // graph mode: an edge's source position is its own position unless k_bucket finds an earlier one
              if (a.graph) a.srcpos[cbase + k0 + u * TN] = tile * TN * W + k0 + u * TN;
            }
          }
          a.cand[cbase + k0 + u * TN] = val[u];
          a.cflag[cbase + k0 + u * TN] = (uint8_t)fl;  // k_bucket corrects the exceptions
          if (a.candk && (val[u] != UNSET || slu[u] == 0 || !gat[u])) a.candk[cbase + k0 + u * TN] = val[u];
        }
      }
    }
    __syncthreads();
    // phase 3: per-node list memberships (bipartite.h:33-66 push conditions)
    uint32_t hb = 0, to = 0;
    const bool act = i < F;
    if (act) {
      hb = s_hb[n];
      to = s_to[n];
      if (!a.graph) a.firstpos[s * a.fcap + i] = UNSET;  // k_bucket stores it for nodes that are sampled as a neighbour too
      // graph mode: a node is always an out node of its own slice
      if (a.graph) hb |= 1u << to;
      a.hasedge[s * a.fcap + i] = hb;
    }
    if (a.graph) {
      for (uint32_t g = 0; g < P; g++) {
        const uint32_t ec = act ? s_ec[n * P + g] : 0u;  // <= fanout <= 255
        if (act) a.ecnt[(s * a.fcap + i) * P + g] = (uint8_t)ec;
        uint32_t x = ec;
        for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o);
        if (lane_id() == 0 && x) atomicAdd(&s_gcnt[g], x);
        const bool hasg = act && ((hb >> g) & 1u) && to != g;
        for (uint32_t p = 0; p < P; p++) {
          const uint32_t c_pair = __popcll(__ballot(hasg && to == p));
          if (lane_id() == 0 && c_pair) atomicAdd(&s_gcnt[CSL_MAX_PARTS + g * CSL_MAX_PARTS + p], c_pair);
        }
      }
    }
    for (uint32_t g = 0; g < P; g++) {
      const bool own = act && to == g;
      const bool has = act && ((hb >> g) & 1u);
      const uint32_t c_out = __popcll(__ballot(has));
      const uint32_t c_owned = __popcll(__ballot(own && has));
      const uint32_t c_self = __popcll(__ballot(own));
      const uint32_t c_to = __popcll(__ballot(own && (hb & ~(1u << g)) != 0));
      const uint32_t c_from = __popcll(__ballot(has && !own));
      if (lane_id() == 0) {
        if (c_out) atomicAdd(&s_cnt[0 * CSL_MAX_PARTS + g], c_out);
        if (c_owned) atomicAdd(&s_cnt[1 * CSL_MAX_PARTS + g], c_owned);
        if (c_self) atomicAdd(&s_cnt[2 * CSL_MAX_PARTS + g], c_self);
        if (c_to) atomicAdd(&s_cnt[3 * CSL_MAX_PARTS + g], c_to);
        if (c_from) atomicAdd(&s_cnt[4 * CSL_MAX_PARTS + g], c_from);
      }
    }
    __syncthreads();
    if (n < 5 * P) {
      const uint32_t kind5 = n / P, g = n - kind5 * P;
      a.tcnt[((size_t)s * a.nk + (K_OUT(P, 0) + kind5 * P + g)) * a.tmax + tile] = s_cnt[kind5 * CSL_MAX_PARTS + g];
    }
    if (a.graph) {
      if (n < P) a.tcnt[((size_t)s * a.nk + K_ECNT(P, n)) * a.tmax + tile] = s_gcnt[n];
      if (n < P * P) {
        const uint32_t g = n / P, p = n - g * P;
        a.tcnt[((size_t)s * a.nk + K_PAIR(P, g, p)) * a.tmax + tile] = s_gcnt[CSL_MAX_PARTS + g * CSL_MAX_PARTS + p];
      }
    }
  }
  __syncthreads();
  uint32_t* gcnt = a.bcnt + (size_t)s * (a.nbmax + 1);
  for (uint32_t b = n; b < nb; b += TN) {
    const uint32_t c = s_bh[b];
    if (c) atomicAdd(&gcnt[b], c);
  }
}

// ---- k_scatter: partitions the candidate stream into the dedup buckets.
// Self entries carry the node's frontier index, edge entries their position;
// holes (short rows, sampled self loops) never reach a bucket.  A block counting-
// sorts its SCT candidates by bucket in LDS first, so the pairs of one bucket
// leave as one contiguous run (random 8-B stores run at ~90 G/s on this chip,
// runs of 8-16 at 400-600 G/s: profiles/microbench/RESULTS.md).
__global__ CSL_LB256 void k_scatter(LArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];  // hist[nb] loff[nb] gbase[nb] staged[2*SCT]
  uint32_t bx, s;
  if (!xcd_block(a, bx, s)) return;
  const uint32_t F = a.fsize[s * (CSL_MAX_LAYERS + 1) + a.layer];
  const uint32_t W = a.W;
  const unsigned long long C = (unsigned long long)F * W;
  const unsigned long long base = (unsigned long long)bx * SCT;
  if (base >= C) return;
  const uint32_t nb = a.nbk[s];
  uint32_t* s_hist = s_dyn;
  uint32_t* s_loff = s_dyn + nb;
  uint32_t* s_gbase = s_dyn + 2 * nb;
  uint2* s_stage = reinterpret_cast<uint2*>(s_dyn + ((3 * nb + 1) & ~1u));
  __shared__ uint32_t s_part[TN];
  const uint32_t n = threadIdx.x;
  for (uint32_t b = n; b < nb; b += TN) s_hist[b] = 0;
  __syncthreads();
  const uint32_t* cand = a.cand + (size_t)s * a.ccap + base;
  const uint32_t cnt = (C - base) < (unsigned long long)SCT ? (uint32_t)(C - base) : (uint32_t)SCT;
  constexpr int CU = 8;
  // pass A: bucket histogram of this block's candidates
  for (uint32_t k0 = n; k0 < cnt; k0 += TN * CU) {
    uint32_t v[CU];
#pragma unroll
    for (int u = 0; u < CU; u++) v[u] = k0 + u * TN < cnt ? cand[k0 + u * TN] : UNSET;
#pragma unroll
    for (int u = 0; u < CU; u++)
      if (v[u] != UNSET) atomicAdd(&s_hist[bucket_of(v[u], nb)], 1u);
  }
  __syncthreads();
  // exclusive scan of the histogram (LDS offsets) + one global reservation per bucket
  const uint32_t chunk = (nb + TN - 1) / TN;
  const uint32_t b_lo = n * chunk < nb ? n * chunk : nb;
  const uint32_t b_hi = b_lo + chunk < nb ? b_lo + chunk : nb;
  uint32_t part = 0;
  for (uint32_t b = b_lo; b < b_hi; b++) part += s_hist[b];
  s_part[n] = part;
  __syncthreads();
  if (n < 64) {
    // one wave scans the 256 partial sums, 4 per lane
    uint32_t p4[4], sum = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      p4[j] = s_part[n * 4 + j];
      sum += p4[j];
    }
    uint32_t tot;
    uint32_t ex = wave_excl_scan(sum, tot);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      s_part[n * 4 + j] = ex;
      ex += p4[j];
    }
  }
  __syncthreads();
  // Queue offset of every bucket = exclusive scan of the stream's bucket sizes (k_sample's histogram).  Every
  // block scans them for itself (a few hundred counters out of L2: cheaper than a one-block-per-stream kernel
  // between k_sample and here); block 0 leaves the offsets in `boff` for k_bucket.
  const uint32_t* gcnt = a.bcnt + (size_t)s * (a.nbmax + 1);
  uint32_t gpart = 0;
  for (uint32_t b = b_lo; b < b_hi; b++) gpart += gcnt[b];
  const uint32_t lrun = s_part[n];
  __syncthreads();  // s_part is reused
  s_part[n] = gpart;
  __syncthreads();
  if (n < 64) {
    uint32_t p4[4], sum = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      p4[j] = s_part[n * 4 + j];
      sum += p4[j];
    }
    uint32_t tot;
    uint32_t ex = wave_excl_scan(sum, tot);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      s_part[n * 4 + j] = ex;
      ex += p4[j];
    }
  }
  __syncthreads();
  uint32_t* cur = a.bcur + (size_t)s * a.nbmax;  // zeroed by scan_body<0>: reservations count from the bucket's offset
  uint32_t* boff = a.boff + (size_t)s * (a.nbmax + 1);
  {
    uint32_t run = lrun, grun = s_part[n];
    for (uint32_t b = b_lo; b < b_hi; b++) {
      const uint32_t h = s_hist[b];
      s_loff[b] = run;
      s_gbase[b] = h ? grun + atomicAdd(&cur[b], h) : 0;
      s_hist[b] = 0;
      run += h;
      if (bx == 0) boff[b] = grun;
      grun += gcnt[b];
    }
    if (bx == 0 && n == TN - 1) boff[nb] = grun;  // the last thread's range ends at nb: the queue's length
  }
  __syncthreads();
  // pass B: place every pair at its sorted LDS position
  for (uint32_t k0 = n; k0 < cnt; k0 += TN * CU) {
    uint32_t v[CU];
#pragma unroll
    for (int u = 0; u < CU; u++) v[u] = k0 + u * TN < cnt ? cand[k0 + u * TN] : UNSET;
#pragma unroll
    for (int u = 0; u < CU; u++) {
      if (v[u] == UNSET) continue;
      const uint32_t c = (uint32_t)base + k0 + u * TN;
      const uint32_t i = div_w(a, c);
      const uint32_t b = bucket_of(v[u], nb);
      const uint32_t p = s_loff[b] + atomicAdd(&s_hist[b], 1u);
      s_stage[p] = make_uint2(v[u], c == i * W ? (SELF_BIT | i) : c);
    }
  }
  __syncthreads();
  // write-out: consecutive threads, consecutive pairs of a bucket, consecutive addresses
  uint2* q = a.queue + (size_t)s * a.ccap;
  const uint32_t staged = s_loff[nb - 1] + s_hist[nb - 1];  // pass B left every bucket's count in s_hist
  for (uint32_t k = n; k < staged; k += TN) {
    const uint2 e = s_stage[k];
    const uint32_t b = bucket_of(e.x, nb);
    q[s_gbase[b] + (k - s_loff[b])] = e;
  }
}

// ---- k_bucket: DuplicateRemover in LDS.  One block owns one bucket of one
// stream: an open-addressing table {node id -> min edge position, frontier
// index}.  From it every candidate learns whether it is the first occurrence
// of its node (a) in the next frontier (out_dr mask, slicer.cpp:45-49) and
// (b) among the in_nodes of its slice (order_and_remove_duplicates,
// bipartite.cpp:4).  flag byte: bit0 new-frontier, bit1 first-in-node,
// bits2-4 owner part, bit5 the node is also a frontier node of this layer
// (k_sample pre-writes the common-case flag; only exceptions are stored here).
__device__ __forceinline__ uint32_t ht_find(const uint32_t* h_key, uint32_t val) {
  uint32_t h = slot_of(val);
  for (uint32_t probes = 0; probes < (uint32_t)HCAP; probes++) {
    const uint32_t k = h_key[h];
    if (k == val) return h;
    if (k == UNSET) return UNSET;
    h = (h + 1) & (HCAP - 1);
  }
  return UNSET;
}

#ifndef CSL_BT
#define CSL_BT 512
#endif
#ifndef CSL_RC
#define CSL_RC 4
#endif
constexpr int BT = CSL_BT;  // threads per k_bucket block: 3 blocks x 8 waves share a CU's LDS
constexpr int RC = CSL_RC;  // queue entries a thread keeps in registers between the two phases

__device__ __forceinline__ uint32_t ht_insert(uint32_t* h_key, uint32_t val) {
  uint32_t h = slot_of(val);
  for (uint32_t probes = 0; probes < (uint32_t)HCAP; probes++) {
    uint32_t kk = h_key[h];
    if (kk == UNSET) kk = atomicCAS(&h_key[h], UNSET, val);
    if (kk == UNSET || kk == val) return h;
    h = (h + 1) & (HCAP - 1);
  }
  return UNSET;
}

#ifndef CSL_BPB
#define CSL_BPB 4
#endif
constexpr int BPB = CSL_BPB;  // buckets a block resolves one after the other; the next one's entries are in flight meanwhile

// HAS_WL: owner parts come from the workload table (a load) instead of v % P.  The kernel is specialised on it
// because a load that MAY be pending in a register makes the compiler wait for the whole memory counter --
// i.e. for the acknowledgement of every earlier scattered flag store -- before each next store: that wait,
// not LDS or the stores themselves, was most of this kernel's time (8 us per bucket).
// CSL_WAVE_DUP_PROBE=1 (measurement only, csl_debug_wave_duplicates): how many queue entries a WAVE-level dedup in front of
// the LDS table could remove -- entries whose id another lane of the same wave already holds -- [0] entries seen, [1] not the
// first of their id among the 64 entries of one register row, [2] among all RC x 64 entries a wave holds of a bucket
__device__ unsigned long long g_wave_dup[3];

template <bool HAS_WL, bool PROBE = false>
__global__ __launch_bounds__(BT) void k_bucket(LArgs a) {
  uint32_t bx, s;
  if (!xcd_block(a, bx, s)) return;
  const uint32_t nbk = a.nbk[s];
  uint32_t b = bx * BPB;
  if (b >= nbk) return;
  __shared__ __attribute__((aligned(16))) uint32_t h_key[HCAP];
  __shared__ __attribute__((aligned(16))) uint32_t h_epos[HCAP];
  __shared__ __attribute__((aligned(16))) uint32_t h_self[HCAP];
  const uint32_t n = threadIdx.x;
  const uint32_t* off = a.boff + (size_t)s * (a.nbmax + 1);
  const uint2* qs = a.queue + (size_t)s * a.ccap;
  const uint32_t W = a.W;
  uint32_t q0 = off[b], q1 = off[b + 1];
  uint2 e[RC];
#pragma unroll
  for (int r = 0; r < RC; r++) e[r] = n + r * BT < q1 - q0 ? qs[q0 + n + r * BT] : make_uint2(UNSET, 0u);
  // what an entry leaves in its slot: the self entry its frontier index, an edge entry its position (min)
  auto record = [&](const uint2 ee, const uint32_t h) {
    if (ee.y & SELF_BIT) {
      // h_self = the node's FIRST frontier index.  Only the seed layer can hold a node twice (later frontiers
      // are deduplicated): strict mode then follows bipartite.cpp:3-17 (k_dupseeds); graph mode, whose
      // specification is only defined for distinct seeds, refuses the minibatch.
      // (a repeated id shows in the evaluate phase: every occurrence but the first reads a smaller index back)
      atomicMin(&h_self[h], ee.y & ~SELF_BIT);
      // graph mode: the self entry is a source of its own slice
      if (a.graph) atomicMin(&h_epos[h], (ee.y & ~SELF_BIT) * W);
    } else {
      atomicMin(&h_epos[h], ee.y);
    }
  };
  auto insert = [&](const uint2 ee) -> uint32_t {
    const uint32_t h = ht_insert(h_key, ee.x);
    if (h == UNSET) {
      atomicOr(&a.meta[s].error, (uint32_t)CSL_ERR_BUCKET_FULL);
    } else {
      record(ee, h);
    }
    return h;
  };
  uint8_t* cflag = a.cflag + (size_t)s * a.ccap;
  auto part_of = [&](const uint32_t v) -> uint32_t { return HAS_WL ? (uint32_t)a.wl[v] : mod_parts(a, v); };
  // what an entry learns from its slot (epos = first edge position of the node, self = its frontier index);
  // g = owner part of the node
  auto judge = [&](const uint2 ee, const uint32_t epos, const uint32_t self, const uint32_t g) {
    if (ee.y & SELF_BIT) {
      const uint32_t i = ee.y & ~SELF_BIT;
      const uint32_t c = i * W;
      if (self != i) {  // the id was in the minibatch before
        if (a.graph) atomicOr(&a.meta[s].error, (uint32_t)CSL_ERR_DUP_SEED);
        else a.dupflag[s] = 1u;
      }
      // k_sample gave every self entry the flag of the common case (new to the frontier; graph mode: also the
      // first source occurrence): only the exceptions are stored here, as 0
      if (a.graph) {
        const uint32_t fe = epos == c;  // epos already includes the self entry
        if (!fe) cflag[c] = 0;
        a.firstpos[s * a.fcap + i] = epos;
      } else {
        // new to the frontier: no edge occurrence before it (UNSET compares greater than any position) and,
        // for a repeated seed id, the first of its self entries (slicer.cpp:45-49)
        const uint32_t newf = epos > c && self == i;
        if (!newf) cflag[c] = 0;
        if (a.layer == 0) a.seedrep[s * a.fcap0 + i] = self;
        if (epos != UNSET) a.firstpos[s * a.fcap + i] = epos;  // (k_sample left UNSET)
      }
    } else {
      // edge candidates carry "first occurrence, not a frontier node" (3 | g << 2) from k_sample
      const uint32_t c = ee.y;
      const uint32_t fe = epos == c;
      if (!fe) {
        cflag[c] = 0;
      } else if (!a.graph && self != UNSET) {
        const uint32_t newf = (unsigned long long)self * W > c;
        cflag[c] = (uint8_t)(newf | 2u | (g << 2) | 32u);  // bit 5: k_emit keeps its in-node rank for k_selfin
      }
      if (a.graph && !fe) a.srcpos[(size_t)s * a.ccap + c] = epos;  // (k_sample stored c itself)
    }
  };
  auto evaluate = [&](const uint2 ee, const uint32_t h) { judge(ee, h_epos[h], h_self[h], part_of(ee.x)); };
  for (int j = 0; j < BPB; j++, b++) {
    if (b >= nbk) break;  // block-uniform
    const uint2* q = qs + q0;
    const uint32_t cnt = q1 - q0;
    if (PROBE) {
      unsigned n_e = 0, n_row = 0, n_all = 0;
      const uint32_t lane = lane_id();
#pragma unroll
      for (int r = 0; r < RC; r++) {
        bool later_row = false, later_all = false;
        for (uint32_t sl = 0; sl < 64; sl++) {
#pragma unroll
          for (int r2 = 0; r2 < RC; r2++) {
            const uint32_t other = __shfl(e[r2].x, sl);
            if (other == e[r].x && (r2 < r || (r2 == r && sl < lane))) {
              later_all = true;
              if (r2 == r) later_row = true;
            }
          }
        }
        if (e[r].x != UNSET) {
          n_e++;
          n_row += later_row;
          n_all += later_all;
        }
      }
      atomicAdd(&g_wave_dup[0], (unsigned long long)n_e);
      atomicAdd(&g_wave_dup[1], (unsigned long long)n_row);
      atomicAdd(&g_wave_dup[2], (unsigned long long)n_all);
    }
    // The next bucket's first entries are requested before this one is resolved.  The loads are issued
    // UNCONDITIONALLY (idle lanes and the last bucket re-read element 0): a load inside a branch would make
    // the compiler wait for the whole memory counter at the next use of any loaded register, prefetch included.
    uint2 en[RC];
    const bool more = j + 1 < BPB && b + 1 < nbk;
    const uint32_t q1n = off[more ? b + 2 : b + 1];
#pragma unroll
    for (int r = 0; r < RC; r++) {
      const uint32_t k = n + r * BT;
      const bool ok = more && k < q1n - q1;
      en[r] = qs[ok ? q1 + k : 0u];
      if (!ok) en[r] = make_uint2(UNSET, 0u);
    }
    auto clear_table = [&]() {
      // 16-byte LDS stores: 6 per thread instead of 24
      const uint4 u4 = make_uint4(UNSET, UNSET, UNSET, UNSET);
      for (uint32_t i = n; i < (uint32_t)HCAP / 4; i += BT) {
        reinterpret_cast<uint4*>(h_key)[i] = u4;
        reinterpret_cast<uint4*>(h_epos)[i] = u4;
        reinterpret_cast<uint4*>(h_self)[i] = u4;
      }
    };
    if (cnt > (uint32_t)HCAP) {  // block-uniform
      // More entries than table slots: the bucket's ids MIGHT not fit (it takes ids chosen against the bucket hash:
      // for hashed ids a bucket of twice the mean is a 45-sigma event).  Resolved in several passes over the queue,
      // each taking the ids of one class of an independent hash, so the sample stays exact instead of being
      // flagged CSL_ERR_BUCKET_FULL; only ids that also collide under the second hash can still overflow a pass.
      const uint32_t npass = (cnt + (uint32_t)HCAP / 2 - 1) / ((uint32_t)HCAP / 2);
      for (uint32_t pass = 0; pass < npass; pass++) {
        clear_table();
        __syncthreads();
        for (uint32_t k = n; k < cnt; k += BT) {
          const uint2 ee = q[k];
          if (pass_of(ee.x, npass) == pass) insert(ee);
        }
        __syncthreads();
        for (uint32_t k = n; k < cnt; k += BT) {
          const uint2 ee = q[k];
          if (pass_of(ee.x, npass) != pass) continue;
          const uint32_t h = ht_find(h_key, ee.x);
          if (h != UNSET) evaluate(ee, h);
        }
        __syncthreads();
      }
    } else {
    clear_table();
    __syncthreads();
    // (one entry after the other: probing all four register entries together, reads then the CAS of the empty
    // slots, was measured slower twice -- 204 us in round 1, 215 us in round 2 against 158 -- and probing with the
    // CAS alone 196 us: an id's many occurrences then hit its slot with atomics, which serialise)
    uint32_t hs[RC];
#pragma unroll
    for (int r = 0; r < RC; r++) hs[r] = e[r].x != UNSET ? insert(e[r]) : UNSET;
    for (uint32_t k = n + RC * BT; k < cnt; k += BT) insert(q[k]);
    __syncthreads();
    {
      // owner parts first (with a workload table these are loads: all requested before the first store), and
      // every slot value before the first flag store
      uint32_t og[RC], ep[RC], sf[RC];
#pragma unroll
      for (int r = 0; r < RC; r++) {
        og[r] = hs[r] != UNSET ? part_of(e[r].x) : 0u;
        ep[r] = hs[r] != UNSET ? h_epos[hs[r]] : UNSET;
        sf[r] = hs[r] != UNSET ? h_self[hs[r]] : UNSET;
      }
#pragma unroll
      for (int r = 0; r < RC; r++)
        if (hs[r] != UNSET) judge(e[r], ep[r], sf[r], og[r]);
    }
    for (uint32_t k = n + RC * BT; k < cnt; k += BT) {
      const uint2 ee = q[k];
      const uint32_t h = ht_find(h_key, ee.x);
      if (h != UNSET) evaluate(ee, h);  // UNSET: table overflow, already flagged
    }
    }
    if (!more) break;
    __syncthreads();  // the table is rebuilt for the next bucket
#pragma unroll
    for (int r = 0; r < RC; r++) e[r] = en[r];
    q0 = q1;
    q1 = q1n;
  }
}

// ---- k_count: per-tile counts of the two candidate-level flags, in
// traversal order, for the list scans.  One wave per tile, four flag bytes per
// lane per load.
__device__ __forceinline__ void count_tile(const LArgs& a, const uint32_t s, const uint32_t tile, const uint32_t F);

constexpr int CNT_T = 1024;  // k_count: 16 waves = 16 tiles per block; the stream's last block scans 3+6P kinds with them
__global__ __launch_bounds__(CNT_T) void k_count(LArgs a) {
  uint32_t bx, s;
  if (!xcd_block(a, bx, s)) return;
  const uint32_t F = a.fsize[s * (CSL_MAX_LAYERS + 1) + a.layer];
  const uint32_t tile = bx * (CNT_T / 64) + (threadIdx.x >> 6);
  if (tile * TN < F) count_tile(a, s, tile, F);
  // the stream's last block turns the tile counts into list offsets (was a kernel of its own)
  const uint32_t per_stream = gridDim.x / (8u * ((a.S + 7u) >> 3));
  uint32_t nb_act = ((F + TN - 1) / TN + CNT_T / 64 - 1) / (CNT_T / 64);
  if (nb_act < 1) nb_act = 1;
  if (nb_act > per_stream) nb_act = per_stream;
  if (bx >= nb_act) return;  // blocks without a tile leave at once; block 0 always stays
  if (last_block_of_stream(a.ticket + 2 * s + 1, nb_act)) scan_body<1>(a, (int)s);
}

__device__ __forceinline__ void count_tile(const LArgs& a, const uint32_t s, const uint32_t tile, const uint32_t F) {
  const uint32_t W = a.W, P = a.P;
  const uint32_t nodes_here = (F - tile * TN) < (uint32_t)TN ? (F - tile * TN) : (uint32_t)TN;
  const uint32_t nbytes = nodes_here * W;
  // tile bases are multiples of 256 bytes: word loads are aligned
  const uint32_t* fw = reinterpret_cast<const uint32_t*>(a.cflag + (size_t)s * a.ccap + (size_t)tile * TN * W);
  // first-in-node counts of the parts are 16-bit fields of two 64-bit accumulators (a lane sees at most
  // 4*W <= 1024 flags): one shift and one add per flag byte instead of a compare chain per part
  uint32_t c_new = 0;
  unsigned long long acc_lo = 0, acc_hi = 0;  // parts 0-3, 4-7
  for (uint32_t o = lane_id() * 4; o < nbytes; o += 256) {
    uint32_t w = fw[o >> 2];
    if (o + 4 > nbytes) w &= 0xFFFFFFFFu >> (8 * (o + 4 - nbytes));  // bytes past the tile's candidates
    c_new += __popc(w & 0x01010101u);
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const uint32_t fl = (w >> (8 * b)) & 0xFFu;
      const unsigned long long one = (unsigned long long)((fl >> 1) & 1u) << (16u * ((fl >> 2) & 3u));
      if (P > 4 && (fl & 16u)) acc_hi += one; else acc_lo += one;
    }
  }
  uint32_t cnt[1 + CSL_MAX_PARTS];
  if (nbytes < 65536u) {
    // a tile total fits its 16-bit field: reduce the packed words
    for (int o = 32; o > 0; o >>= 1) {
      c_new += __shfl_down(c_new, o);
      acc_lo += __shfl_down(acc_lo, o);
      if (P > 4) acc_hi += __shfl_down(acc_hi, o);
    }
    cnt[0] = c_new;
#pragma unroll
    for (int g = 0; g < 4; g++) {
      cnt[1 + g] = (uint32_t)(acc_lo >> (16 * g)) & 0xFFFFu;
      cnt[5 + g] = (uint32_t)(acc_hi >> (16 * g)) & 0xFFFFu;
    }
  } else {
    cnt[0] = c_new;
#pragma unroll
    for (int g = 0; g < 4; g++) {
      cnt[1 + g] = (uint32_t)(acc_lo >> (16 * g)) & 0xFFFFu;
      cnt[5 + g] = (uint32_t)(acc_hi >> (16 * g)) & 0xFFFFu;
    }
#pragma unroll
    for (int k = 0; k < 1 + CSL_MAX_PARTS; k++) {
      uint32_t x = cnt[k];
      for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o);
      cnt[k] = x;
    }
  }
  if (lane_id() == 0) {
#pragma unroll
    for (int k = 0; k < 1 + CSL_MAX_PARTS; k++)
      if (k < 1 + (int)P) st_sc1(&a.tcnt[((size_t)s * a.nk + (K_NEWF + k)) * a.tmax + tile], cnt[k]);
  }
}

// ---- k_emit: stable compaction of everything into the BiPartite lists
// (bipartite.h:9-26) and of the next frontier.  Positions come from the tile
// scans (k_scan) plus ballot ranks inside the tile, so every list is in the
// reference's push order.
#ifndef CSL_EP
#define CSL_EP 8
#endif
constexpr int EP = CSL_EP;  // candidate steps whose flag and id a k_emit thread preloads

__global__ CSL_LB256 void k_emit(LArgs a) {
  uint32_t tile, s;
  if (!xcd_block(a, tile, s)) return;
  const uint32_t F = a.fsize[s * (CSL_MAX_LAYERS + 1) + a.layer];
  if (tile * TN >= F) return;
  const uint32_t n = threadIdx.x;
  const uint32_t w = __builtin_amdgcn_readfirstlane(n >> 6);  // wave-uniform, kept in an SGPR
  const uint32_t lane = lane_id();
  const uint32_t W = a.W, P = a.P;
  const csl_layer_meta& m = a.meta[s].layer[a.layer];
  int* ar = a.arena + (size_t)s * a.arena_stride;
  const uint32_t* tc = a.tcnt + (size_t)s * a.nk * a.tmax;
#define TB(kind) tc[(size_t)(kind)*a.tmax + tile]
  // per-wave counts (<= 64) are BYTES of one word per kind: a wave's prefix over the waves before it is
  // one masked v_sad_u8 instead of a loop of LDS reads
  static_assert(NW == 4, "four wave counts per word");
  __shared__ uint32_t s_wc[2][1 + CSL_MAX_PARTS];   // a step's counts (double-buffered)
  __shared__ uint32_t s_run[2][1 + CSL_MAX_PARTS];  // list position of the step's first candidate
  __shared__ uint32_t s_wn[5 * 2 * NW];             // node-level counts: [kind][word][wave]
  const uint32_t below = (1u << (8u * w)) - 1u;     // byte lanes of the waves before this one
#define WBYTE(word) (reinterpret_cast<uint8_t*>(&(word))[w])
#define BYTESUM(x) __builtin_amdgcn_sad_u8((x), 0u, 0u)
  __shared__ uint32_t s_tb[5 * CSL_MAX_PARTS];         // tile bases of the node-level lists
  __shared__ uint32_t s_mo[6][CSL_MAX_PARTS];          // part offsets: in_nodes + the five node-level lists
  // Everything that does not depend on this block's own stores is requested NOW: the tile bases and part
  // offsets (a load of them between two stores would wait for a full memory round trip every step, the
  // compiler cannot prove they do not alias the lists) and the node-level phase's inputs.
  if (n < 1 + P) s_run[0][n] = n == 0 ? TB(K_NEWF) : TB(K_IN(P, n - 1));
  if (n >= 64 && n < 64 + 5 * P) s_tb[n - 64] = TB(K_OUT(P, 0) + (n - 64));  // kinds OUT, OWNED, SELF, TO, FROM
  if (n >= 128 && n < 128 + 6 * CSL_MAX_PARTS) {
    const uint32_t k6 = (n - 128) / CSL_MAX_PARTS, g = (n - 128) % CSL_MAX_PARTS;
    const int list = k6 == 0 ? CSL_IN_NODES : k6 == 1 ? CSL_OUT_NODES : k6 == 2 ? CSL_OWNED_OUT_NODES
                   : k6 == 3 ? CSL_SELF_IDS_OUT : k6 == 4 ? CSL_TO_IDS : CSL_FROM_IDS;
    s_mo[k6][g] = g < P ? m.off[list][g] : 0u;
  }
  const uint32_t i = tile * TN + n;
  const bool act = i < F;
  uint32_t v = 0, hb = 0;
  if (act) {
    v = a.fr_in[s * a.fr_in_stride + i];
    hb = a.hasedge[s * a.fcap + i];
  }
  // ---- candidate-level lists: next frontier, in_nodes.  Step `it` covers candidates it*TN..it*TN+TN-1
  // of the tile; traversal order = (step, wave, lane): ballot ranks inside the wave, per-wave counts
  // through LDS, ONE barrier per step (counts and running positions are double-buffered).
  const uint32_t nodes_here = (F - tile * TN) < (uint32_t)TN ? (F - tile * TN) : (uint32_t)TN;
  const uint32_t ncand = nodes_here * W;
  const uint32_t iters = (ncand + TN - 1) / TN;
  const size_t cbase = (size_t)s * a.ccap + (size_t)tile * TN * W;
  const uint32_t nf_cap = m.next_frontier;  // already clamped to capacity
  const unsigned long long lt = lt_mask();
  uint32_t* fr_out = a.fr_out + s * a.fr_out_stride;
  unsigned long long* ninfo = a.ninfo + s * a.fcap;
  int* in_list = ar + a.list_base[CSL_IN_NODES];
  const uint8_t* __restrict__ cflag = a.cflag;
  const uint32_t* __restrict__ cand = a.cand;
  for (uint32_t c0 = 0; c0 < iters; c0 += EP) {
    // flags AND ids of the chunk's EP steps are requested up front: an id is only used where a flag is
    // set, but waiting for the flag first puts one more memory round trip on the block's critical path
    // (182 vs 202 us per launch)
    uint32_t pf[EP], pv[EP];
#pragma unroll
    for (int j = 0; j < EP; j++) {
      const uint32_t k = (c0 + j) * TN + n;
      pf[j] = k < ncand ? cflag[cbase + k] : 0u;
      pv[j] = k < ncand ? cand[cbase + k] : 0u;
    }
#pragma unroll
    for (int j = 0; j < EP; j++) {
      if (c0 + j >= iters) break;  // block-uniform
      const uint32_t b = j & 1;    // EP is even: the buffer parity carries over from chunk to chunk
      const uint32_t fl = pf[j], val = pv[j];
      const uint32_t newf = fl & 1u, fe = (fl >> 1) & 1u, g = (fl >> 2) & 7u;
      const unsigned long long m0 = __ballot(newf);
      const uint32_t r0 = __popcll(m0 & lt);
      uint32_t rE = 0;
      if (lane == 0) WBYTE(s_wc[b][0]) = (uint8_t)__popcll(m0);
#pragma unroll
      for (uint32_t gg = 0; gg < CSL_MAX_PARTS; gg++) {
        if (gg < P) {
          const unsigned long long mg = __ballot(fe && g == gg);
          if (g == gg) rE = __popcll(mg & lt);
          if (lane == 0) WBYTE(s_wc[b][1 + gg]) = (uint8_t)__popcll(mg);
        }
      }
      __syncthreads();
      if (n < 1 + P) s_run[b ^ 1][n] = s_run[b][n] + BYTESUM(s_wc[b][n]);  // the next step's first candidate
      if (newf) {
        const uint32_t p = s_run[b][0] + r0 + BYTESUM(s_wc[b][0] & below);
        if (p < nf_cap) {
          fr_out[p] = val;
          // the next layer's row lookup rides on this pass (slicer.cpp:8-9)
          if (!a.last) ninfo[p] = row_lookup(a, val);
        }
      }
#ifndef CSL_ABLATE_EMIT_IN
      if (fe && ((a.pmask >> g) & 1u)) {
        const uint32_t p = s_run[b][1 + g] + rE + BYTESUM(s_wc[b][1 + g] & below);  // local index inside slice g's in_nodes
        in_list[s_mo[0][g] + p] = (int)val;
        if (a.tcur) a.tcur[(size_t)s * a.ccap + s_mo[0][g] + p] = 0;  // k_graph counts the in node's edges here
        // DuplicateRemover::replace's lookup value (mask[v]-1): read back by k_selfin for candidates whose
        // node is in the frontier (flag bit 5) and by k_graph for every edge
        if (a.graph || (fl & 32u)) a.crank[cbase + (c0 + j) * TN + n] = p;
      }
#endif
    }
  }
  static_assert(EP % 2 == 0, "buffer parity");
  // ---- node-level lists: out_nodes, owned_out_nodes, self_ids_out, to_ids, from_ids.  A node's
  // membership in the P lists of a kind is one word of byte counters (parts 0-3; a second word for parts
  // 4-7): ONE wave scan per kind and word ranks it in all parts at once; a wave's total goes to LDS as one
  // word, and the counts of the (<= 3) waves before it still fit a byte (<= 192 + 63).
  const uint32_t to = act ? owner(a, v) : 0u;
  const uint32_t ob = act ? 1u << to : 0u;
  const uint32_t oth = hb & ~ob;                                   // parts other than the owner with an edge
  const bool hi = P > 4;                                           // block-uniform
  const uint32_t one_to = act ? 1u << (8u * (to & 3u)) : 0u;       // the owner's byte ...
  const bool to_hi = to >= 4;                                      // ... of the second word?
  uint32_t wd[5][2];  // kinds OUT, OWNED, SELF, TO, FROM (the order of s_tb / the tile counters)
  wd[0][0] = spread4(hb);
  wd[4][0] = spread4(oth);
  wd[1][0] = (!to_hi && (hb & ob)) ? one_to : 0u;
  wd[2][0] = !to_hi ? one_to : 0u;
  wd[3][0] = (!to_hi && oth) ? one_to : 0u;
  wd[0][1] = wd[1][1] = wd[2][1] = wd[3][1] = wd[4][1] = 0;
  if (hi) {
    wd[0][1] = spread4(hb >> 4);
    wd[4][1] = spread4(oth >> 4);
    wd[1][1] = (to_hi && (hb & ob)) ? one_to : 0u;
    wd[2][1] = to_hi ? one_to : 0u;
    wd[3][1] = (to_hi && oth) ? one_to : 0u;
  }
#pragma unroll
  for (int k = 0; k < 5; k++) {
#pragma unroll
    for (int h = 0; h < 2; h++) {
      if (h == 0 || hi) {
        const uint32_t incl = wave_incl_scan_dpp(wd[k][h]);
        if (lane == 63) s_wn[(k * 2 + h) * NW + w] = incl;  // the wave's counts
        wd[k][h] = incl - wd[k][h];                         // ranks inside the wave
      }
    }
  }
  __syncthreads();
  const uint32_t m0 = w > 0 ? 0xFFFFFFFFu : 0u, m1 = w > 1 ? 0xFFFFFFFFu : 0u, m2 = w > 2 ? 0xFFFFFFFFu : 0u;
#pragma unroll
  for (int k = 0; k < 5; k++) {
#pragma unroll
    for (int h = 0; h < 2; h++) {
      if (h == 0 || hi) {
        const uint32_t* c = &s_wn[(k * 2 + h) * NW];
        wd[k][h] += (c[0] & m0) + (c[1] & m1) + (c[2] & m2);  // + the waves before this one
      }
    }
  }
#define RANK(k, g) ((wd[k][(g) >> 2] >> (8u * ((g)&3u))) & 0xFFu)
#ifndef CSL_ABLATE_EMIT_NODES
  if (act) {
    int outrank_to = -1;
#pragma unroll
    for (uint32_t g = 0; g < CSL_MAX_PARTS; g++) {
      if (g < P && ((hb >> g) & 1u)) {
        const uint32_t p = RANK(0, g) + s_tb[0 * P + g];  // local index inside slice g's out_nodes
        if (g == to) outrank_to = p;
        if (!((a.pmask >> g) & 1u)) continue;  // slice g is not materialised here
        ar[a.list_base[CSL_OUT_NODES] + s_mo[1][g] + p] = (int)v;
        if (g != to && !a.graph) {
          const uint32_t q = RANK(4, g) + s_tb[4 * P + g];
          ar[a.list_base[CSL_FROM_IDS] + s_mo[5][g] + q] = (int)p;
        }
      }
    }
    const uint32_t tsh = 8u * (to & 3u);
#define RANK_TO(k) (((to_hi ? wd[k][1] : wd[k][0]) >> tsh) & 0xFFu)
    const bool mine = (a.pmask >> to) & 1u;  // the node's own slice is materialised here
    {
      const uint32_t q = RANK_TO(2) + s_tb[2 * P + to];
      const uint32_t pos = s_mo[3][to] + q;
      if (mine) ar[a.list_base[CSL_SELF_IDS_OUT] + pos] = outrank_to;
      a.selfpos[s * a.fcap + i] = mine ? pos : UNSET;  // k_selfin fills self_ids_in at the same place
    }
    if (outrank_to >= 0 && mine) {
      const uint32_t q = RANK_TO(1) + s_tb[1 * P + to];
      ar[a.list_base[CSL_OWNED_OUT_NODES] + s_mo[2][to] + q] = outrank_to;
    }
    if (!a.graph && oth != 0 && mine) {
      const uint32_t q = RANK_TO(3) + s_tb[3 * P + to];
      ar[a.list_base[CSL_TO_IDS] + s_mo[4][to] + q] = outrank_to;
    }
#undef RANK_TO
  }
#else
  if (act) a.selfpos[s * a.fcap + i] = 0;
#endif
#undef RANK
#undef TB
#undef WBYTE
#undef BYTESUM
}

// ---- k_dupseeds (strict mode, layer 0, only for a minibatch that holds a seed id more than once): the
// node-level lists of BiPartite::reorder (bipartite.cpp:10-16) for a frontier WITH repeated ids.  The
// reference samples every occurrence (its own draws), and then
//   * out_nodes of slice g keeps the FIRST occurrence of an id among the entries with an edge from g
//     (order_and_remove_duplicates, duplicate.cpp:14-26),
//   * owned_out_nodes / self_ids_* / to_ids / from_ids keep one entry per push; a push is skipped when the
//     previous push to the SAME list carried the same id (bipartite.h:33-66 `back() == nd1`), i.e. when the
//     previous member entry of that list is another occurrence of the id,
//   * every value is replace()d by the id's index in the slice's out_nodes, or -1 (duplicate.cpp:35-39):
//     ALL occurrences of an id point to the same out-node, whichever occurrence pushed it.
// With distinct seeds this is what k_emit wrote (no skip ever applies, every entry is its own first
// occurrence), so k_emit's lists stand and this kernel returns at once.  Otherwise it rewrites the five
// node-level lists of the stream's layer 0 (they only get shorter) and their offsets in the meta.
// One block per stream; entries are walked in chunks of DS_T with carried scans.  in_nodes, self_ids_in's
// values and the next frontier do not depend on it (k_bucket / k_emit / k_selfin).
constexpr int DS_T = 256;
__device__ __forceinline__ uint32_t ds_ld(const uint32_t* p) {
  // written earlier in this kernel by other threads (plain stores or L2 atomics): read past the vector L1
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// exclusive block scan of one value per thread; IS_MAX: running maximum (identity 0), else sum
template <bool IS_MAX>
__device__ __forceinline__ uint32_t ds_block_scan(uint32_t x, uint32_t* s_w, uint32_t& total) {
  const uint32_t lane = lane_id(), w = threadIdx.x >> 6;
  uint32_t incl = x;
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t y = __shfl_up(incl, o);
    if ((int)lane >= o) incl = IS_MAX ? (incl > y ? incl : y) : incl + y;
  }
  uint32_t excl = __shfl_up(incl, 1);
  if (lane == 0) excl = 0;
  __syncthreads();  // s_w free again
  if (lane == 63) s_w[w] = incl;
  __syncthreads();
  uint32_t before = 0, all = 0;
  for (uint32_t k = 0; k < DS_T / 64; k++) {
    const uint32_t t = s_w[k];
    if (k < w) before = IS_MAX ? (before > t ? before : t) : before + t;
    all = IS_MAX ? (all > t ? all : t) : all + t;
  }
  total = all;
  return IS_MAX ? (excl > before ? excl : before) : excl + before;
}

__global__ __launch_bounds__(DS_T) void k_dupseeds(LArgs a) {
  const uint32_t s = blockIdx.x;
  if (!a.dupflag[s]) return;
  const uint32_t F = a.fsize[s * (CSL_MAX_LAYERS + 1)], P = a.P, n = threadIdx.x;
  const uint32_t* fr = a.fr_in + s * a.fr_in_stride;
  const uint32_t* hbv = a.hasedge + s * a.fcap;
  const uint32_t* rep = a.seedrep + s * a.fcap0;
  uint32_t* firstg = a.dupfirst + (size_t)s * a.fcap0 * P;  // [entry][part] first occurrence with an edge from g
  uint32_t* outidx = a.dupout + (size_t)s * a.fcap0 * P;    // [entry][part] its index in out_nodes of g
  uint32_t* selfpos = a.selfpos + s * a.fcap;
  int* ar = a.arena + (size_t)s * a.arena_stride;
  csl_layer_meta& m = a.meta[s].layer[0];
  __shared__ uint32_t s_w[DS_T / 64];
  __shared__ uint32_t s_cnt[6][CSL_MAX_PARTS];  // kinds OUT, OWNED, SELF, TO, FROM, out_nodes pushes (= indptr ones)
  __shared__ uint32_t s_off[5][CSL_MAX_PARTS + 1];
  for (uint32_t k = n; k < F * P; k += DS_T) firstg[k] = UNSET;
  __syncthreads();
  for (uint32_t i = n; i < F; i += DS_T) {
    const uint32_t h = hbv[i], r = rep[i];
    for (uint32_t g = 0; g < P; g++)
      if ((h >> g) & 1u) atomicMin(&firstg[(size_t)r * P + g], i);
  }
  __syncthreads();
  const uint32_t chunks = (F + DS_T - 1) / DS_T;
  // pass 0: out_nodes indices and the size of every list; pass 1: the lists
  for (int pass = 0; pass < 2; pass++) {
    for (uint32_t g = 0; g < P; g++) {
      for (int kind = pass == 0 ? 0 : 1; kind < (pass == 0 ? 6 : 5); kind++) {
        uint32_t run = 0, prev_run = 0;  // carried over the chunks: members so far, last member entry + 1
        for (uint32_t ch = 0; ch < chunks; ch++) {
          const uint32_t i = ch * DS_T + n;
          const bool act = i < F;
          uint32_t v = 0, h = 0, r = 0, to = 0;
          if (act) {
            v = fr[i];
            h = hbv[i];
            r = rep[i];
            to = owner(a, v);
          }
          const bool bit = act && ((h >> g) & 1u), own = act && to == g;
          bool mem;
          if (kind == 0) mem = bit && ds_ld(&firstg[(size_t)r * P + g]) == i;
          else if (kind == 1) mem = own && bit;
          else if (kind == 2) mem = own;
          else if (kind == 3) mem = own && (h & ~(1u << g)) != 0;
          else if (kind == 4) mem = bit && !own;
          else mem = bit;  // out_nodes pushes before reorder (bipartite.h:59-62): each leaves a `1` in indptr
          if (kind != 0) {
            // bipartite.h `back() == nd1`: the previous member entry of this list is the same id
            uint32_t tot;
            uint32_t pm = ds_block_scan<true>(mem ? i + 1 : 0u, s_w, tot);
            if (pm < prev_run) pm = prev_run;
            if (tot > prev_run) prev_run = tot;
            if (mem && pm != 0 && rep[pm - 1] == r) mem = false;
          }
          uint32_t tot;
          const uint32_t rank = run + ds_block_scan<false>(mem ? 1u : 0u, s_w, tot);
          run += tot;
          if (pass == 0) {
            if (kind == 0 && mem) outidx[(size_t)i * P + g] = rank;
          } else if (act) {
            // index of the id in slice g's out_nodes (any occurrence may have pushed it), or -1
            const uint32_t fo = ds_ld(&firstg[(size_t)r * P + g]);
            const int oi = fo == UNSET ? -1 : (int)ds_ld(&outidx[(size_t)fo * P + g]);
            if (kind == 1 && mem) ar[a.list_base[CSL_OWNED_OUT_NODES] + s_off[1][g] + rank] = oi;
            if (kind == 2) {
              if (mem) ar[a.list_base[CSL_SELF_IDS_OUT] + s_off[2][g] + rank] = oi;
              if (own) selfpos[i] = mem ? s_off[2][g] + rank : UNSET;  // where k_selfin puts self_ids_in
            }
            if (kind == 3 && mem) ar[a.list_base[CSL_TO_IDS] + s_off[3][g] + rank] = oi;
            if (kind == 4 && mem) ar[a.list_base[CSL_FROM_IDS] + s_off[4][g] + rank] = oi;
          }
        }
        if (pass == 0 && n == 0) s_cnt[kind][g] = run;
      }
    }
    __syncthreads();
    if (pass == 0) {
      if (n < 5) {
        uint32_t run = 0;
        for (uint32_t g = 0; g <= CSL_MAX_PARTS; g++) {
          s_off[n][g] = run;
          if (g < P) run += s_cnt[n][g];
        }
        const int list = n == 0 ? CSL_OUT_NODES : n == 1 ? CSL_OWNED_OUT_NODES : n == 2 ? CSL_SELF_IDS_OUT
                       : n == 3 ? CSL_TO_IDS : CSL_FROM_IDS;
        for (uint32_t g = 0; g <= CSL_MAX_PARTS; g++) {
          m.off[list][g] = s_off[n][g];
          if (n == 2) m.off[CSL_SELF_IDS_IN][g] = s_off[n][g];
        }
      }
      if (n >= 64 && n < 64 + P) m.indptr_len[n - 64] = s_cnt[5][n - 64];
      __syncthreads();
      // out_nodes themselves: the first occurrences, in frontier order (their ranks are known now)
      for (uint32_t i = n; i < F; i += DS_T) {
        const uint32_t v = fr[i], h = hbv[i], r = rep[i];
        for (uint32_t g = 0; g < P; g++)
          if (((h >> g) & 1u) && ds_ld(&firstg[(size_t)r * P + g]) == i)
            ar[a.list_base[CSL_OUT_NODES] + s_off[0][g] + ds_ld(&outidx[(size_t)i * P + g])] = (int)v;
      }
      __syncthreads();
    }
  }
}

// ---- k_graph (CSL_MODE_GRAPH only): what BiPartite::add_edge was meant to build
// (bipartite.h:55-66) and what slice_layer meant to record per peer
// (slicer.cpp:40-43): CSR row pointers and local source indices of every slice,
// the (sender, receiver) boundary lists, and the mean divisor of owned nodes.
// One thread per frontier node; runs after k_emit (needs the in-node ranks).
__global__ __launch_bounds__(TN) void k_graph(LArgs a) {
  uint32_t tile, s;
  if (!xcd_block(a, tile, s)) return;
  const uint32_t F = a.fsize[s * (CSL_MAX_LAYERS + 1) + a.layer];
  if (tile * TN >= F) return;
  const uint32_t n = threadIdx.x, w = n >> 6;
  const uint32_t W = a.W, P = a.P;
  const csl_layer_meta& m = a.meta[s].layer[a.layer];
  int* ar = a.arena + (size_t)s * a.arena_stride;
  const uint32_t* tc = a.tcnt + (size_t)s * a.nk * a.tmax;
#define TB(kind) tc[(size_t)(kind)*a.tmax + tile]
  __shared__ uint32_t s_wo[NW][CSL_MAX_PARTS];                  // out-node counts per wave
  __shared__ uint32_t s_we[NW][CSL_MAX_PARTS];                  // edge counts per wave
  __shared__ uint32_t s_wp[NW][CSL_MAX_PARTS * CSL_MAX_PARTS];  // pair counts per wave
  const uint32_t i = tile * TN + n;
  const bool act = i < F;
  uint32_t v = 0, hb = 0, to = 0;
  if (act) {
    v = a.fr_in[s * a.fr_in_stride + i];
    hb = a.hasedge[s * a.fcap + i];
    to = owner(a, v);
  }
  uint32_t r_out[CSL_MAX_PARTS], ec[CSL_MAX_PARTS], r_ec[CSL_MAX_PARTS], r_pair[CSL_MAX_PARTS];
  const unsigned long long lt = lt_mask();
#pragma unroll
  for (uint32_t g = 0; g < CSL_MAX_PARTS; g++) {
    r_out[g] = ec[g] = r_ec[g] = r_pair[g] = 0;
    if (g < P) {
      const bool has = act && ((hb >> g) & 1u);
      const unsigned long long b_out = __ballot(has);
      r_out[g] = __popcll(b_out & lt);
      ec[g] = act ? (uint32_t)a.ecnt[(s * a.fcap + i) * P + g] : 0u;
      uint32_t tot;
      r_ec[g] = wave_excl_scan(ec[g], tot);
      if (lane_id() == 0) {
        s_wo[w][g] = __popcll(b_out);
        s_we[w][g] = tot;
      }
      // boundary pair (sender g, receiver p): nodes owned by p with an edge from g
      for (uint32_t p = 0; p < P; p++) {
        const unsigned long long b_pair = __ballot(has && to == p && p != g);
        if (to == p) r_pair[g] = __popcll(b_pair & lt);
        if (lane_id() == 0) s_wp[w][g * CSL_MAX_PARTS + p] = __popcll(b_pair);
      }
    }
  }
  __syncthreads();
  if (!act) return;
  uint32_t outrank[CSL_MAX_PARTS], rs[CSL_MAX_PARTS];
  uint32_t deg = 0;
#pragma unroll
  for (uint32_t g = 0; g < CSL_MAX_PARTS; g++) {
    outrank[g] = rs[g] = 0;
    if (g < P) {
      uint32_t p = r_out[g], q = r_ec[g];
      for (uint32_t ww = 0; ww < w; ww++) {
        p += s_wo[ww][g];
        q += s_we[ww][g];
      }
      outrank[g] = p + TB(K_OUT(P, g));
      rs[g] = q + TB(K_ECNT(P, g));  // first index of this node's row in slice g's indices
      deg += ec[g];
      if (((hb >> g) & 1u) && ((a.pmask >> g) & 1u))
        ar[a.list_base[CSL_INDPTR] + m.off[CSL_INDPTR][g] + outrank[g] + 1] = (int)(rs[g] + ec[g]);
      if (i == 0) ar[a.list_base[CSL_INDPTR] + m.off[CSL_INDPTR][g]] = 0;
    }
  }
#pragma unroll
  for (uint32_t g = 0; g < CSL_MAX_PARTS; g++) {
    if (g < P && g != to && ((hb >> g) & 1u)) {
      uint32_t q = r_pair[g];
      for (uint32_t ww = 0; ww < w; ww++) q += s_wp[ww][g * CSL_MAX_PARTS + to];
      q += TB(K_PAIR(P, g, to));
      if ((a.pmask >> g) & 1u)
        ar[a.list_base[CSL_FROM_IDS] + m.off[CSL_FROM_IDS][g] + m.pair_off[0][g][to] + q] = (int)outrank[g];
      if ((a.pmask >> to) & 1u)
        ar[a.list_base[CSL_TO_IDS] + m.off[CSL_TO_IDS][to] + m.pair_off[1][to][g] + q] = (int)outrank[to];
    }
  }
  // mean divisor, next to owned_out_nodes (same order as self_ids_*)
  if ((a.pmask >> to) & 1u) {
    const uint32_t q = a.selfpos[s * a.fcap + i] - m.off[CSL_SELF_IDS_OUT][to];
    ar[a.list_base[CSL_OWNED_DEGREE] + m.off[CSL_OWNED_DEGREE][to] + q] = (int)deg;
  }
  // the node's edges, sampling order: local index of each source inside its slice
  const size_t cb = (size_t)s * a.ccap + (size_t)i * W;
  for (uint32_t slot = 1; slot < W; slot++) {
    const uint32_t val = a.cand[cb + slot];
    if (val == UNSET) continue;
    const uint32_t g = owner(a, val);
    if (!((a.pmask >> g) & 1u)) continue;  // the edge lives in a slice that is not materialised here
    const uint32_t rank = a.crank[(size_t)s * a.ccap + a.srcpos[cb + slot]];
    uint32_t pos = 0;
#pragma unroll
    for (uint32_t gg = 0; gg < CSL_MAX_PARTS; gg++) {
      if (gg == g) {
        pos = rs[gg];
        rs[gg]++;
      }
    }
    ar[a.list_base[CSL_INDICES] + m.off[CSL_INDICES][g] + pos] = (int)rank;
    if (a.tcur) atomicAdd(&a.tcur[(size_t)s * a.ccap + m.off[CSL_IN_NODES][g] + rank], 1u);
  }
  // the slice by source (k_transpose) also lists the node's self entry: the in-node rank k_selfin stores
  if (a.tcur && ((a.pmask >> to) & 1u)) {
    const uint32_t selfrank = a.crank[(size_t)s * a.ccap + a.firstpos[s * a.fcap + i]];
    atomicAdd(&a.tcur[(size_t)s * a.ccap + m.off[CSL_IN_NODES][to] + selfrank], 1u);
  }
#undef TB
}

// ---- the slice by SOURCE (CSL_FLAG_TRANSPOSE).  k_graph left the number of entries of every in node of slice g of stream
// s in tcur; four small grid-wide passes turn that into t_indptr / t_indices (one 1024-thread block per slice did all
// of it in round 2's first version: 0.35 ms per launch for the trainer's middle layer, 4 ms for a deepest layer of 0.7 M
// in nodes -- one CU per stream, a tenth of a GAT training step on the side stream):
//   k_tsum   sum of every tile of TT in nodes                              grid (tiles, S * P)
//   k_tptr   tile base = sum of the tile sums before it; exclusive scan inside the tile -> t_indptr, tcur := cursor
//   k_tfill  every out row (and self entry) drops its row id at its source's cursor (atomic: any order)
//   k_tsort  every in node's short list sorted -> deterministic: ~r of the self entry first, then the out rows ascending
constexpr int TT = 2048;  // in nodes per tile: 8 per thread
struct TSlice {
  uint32_t n_in, n_out, n_self;
  uint32_t* cnt;
  int* tptr;
  int* trow;
  const int *indptr, *indices, *self_in, *self_out;
};
__device__ __forceinline__ bool tslice(const LArgs& a, TSlice& t, uint32_t& s, uint32_t& g) {
  s = blockIdx.y / a.P, g = blockIdx.y - s * a.P;
  if (!((a.pmask >> g) & 1u)) return false;
  if (a.fsize[s * (CSL_MAX_LAYERS + 1) + a.layer] == 0) return false;
  const csl_layer_meta& m = a.meta[s].layer[a.layer];
  int* ar = a.arena + (size_t)s * a.arena_stride;
  t.n_in = m.off[CSL_IN_NODES][g + 1] - m.off[CSL_IN_NODES][g];
  t.n_out = m.off[CSL_OUT_NODES][g + 1] - m.off[CSL_OUT_NODES][g];
  t.n_self = m.off[CSL_SELF_IDS_IN][g + 1] - m.off[CSL_SELF_IDS_IN][g];
  t.cnt = a.tcur + (size_t)s * a.ccap + m.off[CSL_IN_NODES][g];
  t.tptr = ar + a.list_base[CSL_T_INDPTR] + m.off[CSL_T_INDPTR][g];
  t.trow = ar + a.list_base[CSL_T_INDICES] + m.off[CSL_T_INDICES][g];
  t.indptr = ar + a.list_base[CSL_INDPTR] + m.off[CSL_INDPTR][g];
  t.indices = ar + a.list_base[CSL_INDICES] + m.off[CSL_INDICES][g];
  t.self_in = ar + a.list_base[CSL_SELF_IDS_IN] + m.off[CSL_SELF_IDS_IN][g];
  t.self_out = ar + a.list_base[CSL_SELF_IDS_OUT] + m.off[CSL_SELF_IDS_OUT][g];
  return true;
}
__device__ __forceinline__ uint32_t block_sum(uint32_t x, uint32_t* s_w) {  // 256 threads
  for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o);
  if (lane_id() == 0) s_w[threadIdx.x >> 6] = x;
  __syncthreads();
  const uint32_t tot = s_w[0] + s_w[1] + s_w[2] + s_w[3];
  __syncthreads();
  return tot;
}
__global__ __launch_bounds__(TN) void k_tsum(LArgs a, uint32_t* __restrict__ tsum, uint32_t ttiles) {
  TSlice t;
  uint32_t s, g;
  if (!tslice(a, t, s, g) || blockIdx.x * TT >= t.n_in) return;
  __shared__ uint32_t s_w[NW];
  const uint32_t u0 = blockIdx.x * TT + threadIdx.x * 8;
  uint32_t x = 0, mx = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const uint32_t c = u0 + k < t.n_in ? t.cnt[u0 + k] : 0u;
    x += c;
    mx = c > mx ? c : mx;
  }
  const uint32_t tot = block_sum(x, s_w);
  if (threadIdx.x == 0) tsum[(size_t)blockIdx.y * ttiles + blockIdx.x] = tot;
  // the slice's longest list (csl_layer_meta.t_max_len; the round's memset zeroed it)
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t y = __shfl_down(mx, o);
    mx = y > mx ? y : mx;
  }
  if (lane_id() == 0 && mx) atomicMax(&a.meta[s].layer[a.layer].t_max_len[g], mx);
}
__global__ __launch_bounds__(TN) void k_tptr(LArgs a, const uint32_t* __restrict__ tsum, uint32_t ttiles) {
  TSlice t;
  uint32_t s, g;
  if (!tslice(a, t, s, g)) return;
  if (t.n_in == 0) {  // a part without a node in this layer: its row pointers are the single 0
    if (blockIdx.x == 0 && threadIdx.x == 0) t.tptr[0] = 0;
    return;
  }
  if (blockIdx.x * TT >= t.n_in) return;
  __shared__ uint32_t s_w[NW];
  const uint32_t* ts = tsum + (size_t)blockIdx.y * ttiles;
  uint32_t before = 0;
  for (uint32_t k = threadIdx.x; k < blockIdx.x; k += TN) before += ts[k];
  const uint32_t base = block_sum(before, s_w);
  const uint32_t u0 = blockIdx.x * TT + threadIdx.x * 8;
  uint32_t c[8], x = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    c[k] = u0 + k < t.n_in ? t.cnt[u0 + k] : 0u;
    x += c[k];
  }
  uint32_t wtot;
  uint32_t run = wave_excl_scan(x, wtot);
  if (lane_id() == 0) s_w[threadIdx.x >> 6] = wtot;
  __syncthreads();
  for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) run += s_w[w];
  run += base;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    if (u0 + k < t.n_in) {
      t.tptr[u0 + k] = (int)run;
      t.cnt[u0 + k] = run;  // the fill cursor
    }
    run += c[k];
  }
  // the slice's last in node closes the row pointers
  if (u0 <= t.n_in - 1 && t.n_in - 1 < u0 + 8) t.tptr[t.n_in] = (int)run;
}
__global__ __launch_bounds__(TN) void k_tfill(LArgs a) {
  TSlice t;
  uint32_t s, g;
  if (!tslice(a, t, s, g)) return;
  const uint32_t r = blockIdx.x * TN + threadIdx.x;
  if (r < t.n_self) {
    const int u = t.self_in[r];
    if (u >= 0) t.trow[atomicAdd(&t.cnt[u], 1u)] = ~t.self_out[r];
  }
  if (r < t.n_out) {
    const int e1 = t.indptr[r + 1];
    for (int e = t.indptr[r]; e < e1; e++) t.trow[atomicAdd(&t.cnt[t.indices[e]], 1u)] = (int)r;
  }
}
__global__ __launch_bounds__(TN) void k_tsort(LArgs a) {
  TSlice t;
  uint32_t s, g;
  if (!tslice(a, t, s, g)) return;
  const uint32_t u = blockIdx.x * TN + threadIdx.x;
  if (u >= t.n_in) return;
  const int j0 = t.tptr[u], j1 = t.tptr[u + 1];
  int* v = t.trow + j0;
  const int len = j1 - j0;
  if (len <= 24) {
    for (int j = 1; j < len; j++) {  // insertion sort: one or two entries on average
      const int x = v[j];
      int k = j - 1;
      while (k >= 0 && v[k] > x) {
        v[k + 1] = v[k];
        k--;
      }
      v[k + 1] = x;
    }
    return;
  }
  // a hub (a source that thousands of the minibatch's rows sampled) is left as it is: a single thread sorting 10^4
  // entries in global memory took 10-25 ms of the round (profiles/hub_probe.py), and a consumer does not gather over
  // such a list anyway (cslicer_hip.h, CSL_T_SORTED_MAX / t_max_len)
  if (len > CSL_T_SORTED_MAX) return;
  // heap sort, O(len log len) whatever the order
  auto sift = [&](int root, int end) {
    const int x = v[root];
    for (;;) {
      int child = 2 * root + 1;
      if (child >= end) break;
      if (child + 1 < end && v[child + 1] > v[child]) child++;
      if (v[child] <= x) break;
      v[root] = v[child];
      root = child;
    }
    v[root] = x;
  };
  for (int i = len / 2 - 1; i >= 0; i--) sift(i, len);
  for (int end = len - 1; end > 0; end--) {
    const int top = v[0];
    v[0] = v[end];
    v[end] = top;
    sift(0, end);
  }
}

// ---- k_selfin: self_ids_in (replace(self_ids_in), bipartite.cpp:6): the
// in-node rank of each frontier node inside its own slice, -1 if it was never
// sampled as a neighbour.
__device__ __forceinline__ void selfin_body(const LArgs& a, const uint32_t bx, const uint32_t s) {
  // four nodes per thread (stride TN, coalesced), all gathers in flight before the stores
  const uint32_t F = a.fsize[s * (CSL_MAX_LAYERS + 1) + a.layer];
  const uint32_t i0 = bx * 4 * TN + threadIdx.x;
  if (bx * 4 * TN >= F) return;
  int* ar = a.arena + (size_t)s * a.arena_stride + a.list_base[CSL_SELF_IDS_IN];
  const uint32_t* __restrict__ crank = a.crank + (size_t)s * a.ccap;
  uint32_t fp[4], sp[4];
  int r[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const uint32_t i = i0 + j * TN;
    fp[j] = i < F ? a.firstpos[s * a.fcap + i] : UNSET;
    sp[j] = i < F ? a.selfpos[s * a.fcap + i] : UNSET;
  }
#pragma unroll
  for (int j = 0; j < 4; j++) r[j] = fp[j] == UNSET ? -1 : (int)crank[fp[j]];
#pragma unroll
  for (int j = 0; j < 4; j++)
    if (sp[j] != UNSET) ar[sp[j]] = r[j];
}

__global__ CSL_LB256 void k_selfin(LArgs a) {
  uint32_t bx, s;
  if (!xcd_block(a, bx, s)) return;
  selfin_body(a, bx, s);
}
// k_selfin of layer l and k_degree of layer l+1 both depend on layer l's k_emit only and are small: one launch.
// Blocks [0, n_degree) are the next layer's k_degree (incl. its last-block scan), the rest this layer's k_selfin.
__global__ CSL_LB256 void k_selfin_degree(LArgs a, LArgs nx, uint32_t n_degree) {
  uint32_t bx, s;
  if (blockIdx.x < n_degree) {
    if (!xcd_block_at(nx, blockIdx.x, n_degree, bx, s)) return;
    degree_body(nx, bx, s, n_degree / (8u * ((nx.S + 7u) >> 3)), nullptr, nullptr);
  } else {
    if (!xcd_block_at(a, blockIdx.x - n_degree, gridDim.x - n_degree, bx, s)) return;
    selfin_body(a, bx, s);
  }
}

// ---- k_mt19937_fill: the std::mt19937 stream (slicer.h:33), generated on the
// device into a ring.  One workgroup; the 624-word twist splits into three
// dependent phases of independent words (i+397 wraps at 227).
__device__ __forceinline__ uint32_t mt_twist(uint32_t u, uint32_t l) {
  const uint32_t y = (u & 0x80000000u) | (l & 0x7fffffffu);
  return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
constexpr int MTT = 256;

// The recurrence x[n+624] = x[n+397] ^ twist(x[n], x[n+1]) lets 227 words be computed side by side.  Lane i < 227 owns
// the words i, 227+i and 454+i of every 624-word block: the new word 227+i needs the new word i, and 454+i the new
// 227+i -- the lane's own results -- and everything else a lane reads belongs to the PREVIOUS block.  So one block is
// one LDS round trip (the other lanes' old words) and ONE workgroup barrier, not three dependent phases (round 1's
// kernel: 1.03 G words/s, and at S <= 16 the slicer waited for it: profiles/small_s_trace).  The barrier does not
// wait for the ring stores (s_waitcnt lgkmcnt only).  The one workgroup is ALU-issue-bound, so it stores the words
// untempered; k_mt19937_temper, a chip-wide pass behind it on the same HIP stream, applies the output function in place
// (k_sample is issue-bound too: tempering each draw there cost it 7 %).  Word 623 needs the new word 0: lane 169 recomputes that one.
__global__ __launch_bounds__(MTT) void k_mt19937_fill(uint32_t* state, uint32_t* ring, unsigned long long ring_mask,
                                                     unsigned long long pos0, uint32_t nblocks) {
  __shared__ uint32_t buf[2][624 + 8];
  const uint32_t t = threadIdx.x;
  for (uint32_t i = t; i < 624; i += MTT) buf[0][i] = state[i];
  __syncthreads();
  const bool act = t < 227, has2 = t < 170;
  uint32_t v0 = 0, v1 = 0, v2 = 0;
  if (act) {
    v0 = buf[0][t];
    v1 = buf[0][227 + t];
    if (has2) v2 = buf[0][454 + t];
  }
  uint32_t cur = 0;
  const uint32_t m = (uint32_t)ring_mask;
  uint32_t p = (uint32_t)pos0 + t;
  for (uint32_t blk = 0; blk < nblocks; blk++) {
    const uint32_t* A = buf[cur];
    uint32_t* B = buf[cur ^ 1];
    if (act) {
      const uint32_t a1 = A[t + 1], a397 = A[t + 397], a228 = A[t + 228];
      const uint32_t a455 = t < 169 ? A[t + 455] : 0u;
      // the new word 0, for word 623 (every lane reads the three words -- LDS broadcasts -- in the same round trip:
      // a branch for lane 169 alone would cost its wave, and with it the barrier, a second one)
      const uint32_t b0 = A[397] ^ mt_twist(A[0], A[1]);
      const uint32_t n0 = a397 ^ mt_twist(v0, a1);
      const uint32_t n1 = n0 ^ mt_twist(v1, a228);
      const uint32_t n2 = n1 ^ mt_twist(v2, t == 169 ? b0 : a455);
      B[t] = n0;
      B[227 + t] = n1;
      if (has2) B[454 + t] = n2;
      v0 = n0;
      v1 = n1;
      v2 = n2;
      // (32-bit index arithmetic is exact: the ring has at most 2^31 words)
      ring[p & m] = n0;
      ring[(p + 227u) & m] = n1;
      if (has2) ring[(p + 454u) & m] = n2;
      p += 624u;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    cur ^= 1;
  }
  __syncthreads();
  for (uint32_t i = t; i < 624; i += MTT) state[i] = buf[cur][i];
}

__global__ __launch_bounds__(256) void k_mt19937_temper(uint32_t* ring, uint32_t m, uint32_t p0, uint32_t nwords) {
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < nwords; i += gridDim.x * 256u)
    ring[(p0 + i) & m] = mt_temper(ring[(p0 + i) & m]);
}

// one generator chunk: nblocks * 624 words from position pos0 on `st`
void launch_mt_fill(hipStream_t st, uint32_t* state, uint32_t* ring, uint64_t ring_words, uint64_t pos0, uint32_t nblocks) {
  hipLaunchKernelGGL(k_mt19937_fill, dim3(1), dim3(MTT), 0, st, state, ring, (unsigned long long)(ring_words - 1),
                     (unsigned long long)pos0, nblocks);
  uint64_t nwords = (uint64_t)nblocks * 624u;
  if (nwords > ring_words) {  // (csl_rng_peek may run the generator over more than one lap: only the last lap is left)
    pos0 += nwords - ring_words;
    nwords = ring_words;
  }
  hipLaunchKernelGGL(k_mt19937_temper, dim3((unsigned)((nwords + 1023u) / 1024u)), dim3(256), 0, st, ring,
                     (uint32_t)(ring_words - 1), (uint32_t)pos0, (uint32_t)nwords);
}

// packs the reference's int64 CSR into rowinfo / u32 indices on the device
__global__ void k_pack_rows(const long long* indptr, unsigned long long* rowinfo, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const unsigned long long o = (unsigned long long)indptr[i];
    const unsigned long long d = (unsigned long long)(indptr[i + 1] - indptr[i]);
    rowinfo[i] = (o << DEG_BITS) | d;
  }
}
__global__ void k_pack_off32(const long long* indptr, uint32_t* off32, size_t n1) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n1; i += stride) off32[i] = (uint32_t)indptr[i];
}
// csl_fetch_sample*: gathers one sample's lists (a segment per layer and list kind, spread over the arenas)
// into ONE contiguous device buffer, so that the sample leaves the GPU in a single D2H copy instead of ~30
struct FetchSegs {
  const int* src[CSL_MAX_LAYERS * CSL_NUM_LISTS];
  uint32_t dst[CSL_MAX_LAYERS * CSL_NUM_LISTS];
  uint32_t n[CSL_MAX_LAYERS * CSL_NUM_LISTS];
  int count;
};
__global__ __launch_bounds__(256) void k_pack_sample(FetchSegs f, int* out) {
  const int* src = f.src[blockIdx.y];
  int* dst = out + f.dst[blockIdx.y];
  const uint32_t n = f.n[blockIdx.y];
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) dst[i] = src[i];
}
__global__ void k_pack_indices(const long long* src, uint32_t* dst, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = (uint32_t)src[i];
}
__global__ void k_pack_wl(const int* src, uint8_t* dst, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = (uint8_t)src[i];
}

// ------------------------------------------------------------------ host side
thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIPCHECK(x)                                                                          \
  do {                                                                                       \
    hipError_t _e = (x);                                                                     \
    if (_e != hipSuccess)                                                                    \
      return fail(CSL_E_HIP, "%s failed: %s (%s:%d)", #x, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

struct TimedEvent {
  hipEvent_t a, b;
  int kernel;
};

}  // namespace

struct csl_engine {
  csl_config cfg;
  int S, P, L, slots;
  uint32_t N;
  size_t E;
  // rounds alternate between two HIP streams and two scratch sets (when there are >= 2 result
  // slots): the latency-bound small layers of round r+1 run beside the big last layer of round r.
  // `stream` aliases streams[0].
  hipStream_t stream = nullptr, rng_stream = nullptr;
  hipStream_t streams[CSL_MAX_SETS] = {};
  int nsets = 1;
  hipEvent_t rng_event = nullptr;
  hipEvent_t chain_event = nullptr;  // recorded after a round's last rng-position update
  bool chain_valid = false;
  std::vector<hipEvent_t> slot_event;  // recorded after the round that fills a result slot
  std::vector<char> slot_pending;
  std::vector<hipEvent_t> desc_event;  // recorded after a slot's batch descriptors were uploaded
  std::vector<char> desc_inflight;
  // graph
  unsigned long long* rowinfo = nullptr;
  uint32_t* off32 = nullptr;
  uint32_t* indices = nullptr;
  uint8_t* wl = nullptr;
  // node order
  long long* nodes = nullptr;
  int64_t n_nodes = 0;
  long long* seedbuf = nullptr;  // staging for csl_submit_seeds
  size_t seedbuf_cap = 0;
  // rng
  uint32_t* ring = nullptr;
  uint32_t* mt_state = nullptr;
  uint64_t ring_words = 0, gen_hi = 0;
  unsigned long long* rngpos = nullptr;
  unsigned long long* rngbase = nullptr;
  unsigned long long* acc = nullptr;
  std::vector<uint64_t> pos_ub, pos_lb;
  uint64_t worst_draws = 0;
  // the generator runs ahead in chunks; a round waits only for the chunk that covers its draws
  struct RngChunk {
    uint64_t hi;
    hipEvent_t ev;
  };
  std::deque<RngChunk> rng_chunks;
  std::vector<hipEvent_t> rng_event_pool;
  // asynchronous snapshots of the streams' positions keep the host's bounds tight without syncing
  static constexpr int NSNAP = 8;
  unsigned long long* snap_host = nullptr;  // pinned [NSNAP][S]
  hipEvent_t snap_ev[NSNAP] = {};
  uint64_t snap_round[NSNAP] = {};
  bool snap_pending[NSNAP] = {};
  uint64_t rounds_submitted = 0;
  uint64_t snap_next = 0;  // oldest round whose snapshot has not been applied yet
  // capacities
  size_t fcap[CSL_MAX_LAYERS + 1];  // frontier capacity entering layer l
  size_t fcap_max = 0, ccap_max = 0;
  uint32_t tmax = 0, nk = 0;
  // scratch
  uint32_t* fr[CSL_MAX_LAYERS + 1] = {};
  unsigned long long* ninfo = nullptr;
  uint32_t* hasedge = nullptr;
  uint32_t* selfpos = nullptr;
  uint32_t* firstpos = nullptr;
  uint32_t* cand = nullptr;
  uint8_t* cflag = nullptr;
  uint32_t* crank = nullptr;
  uint8_t* ecnt = nullptr;
  uint32_t* srcpos = nullptr;
  uint32_t* tcur = nullptr;
  uint32_t* tsum = nullptr;     // CSL_FLAG_TRANSPOSE: [nsets][S][P][ttiles_max] in-node tile sums
  size_t ttiles_max = 0;
  uint32_t* dupflag = nullptr;   // [nsets][S]
  uint32_t* seedrep = nullptr;   // [nsets][S][fcap0]
  uint32_t* dupfirst = nullptr;  // [nsets][S][fcap0*P]
  uint32_t* dupout = nullptr;
  unsigned long long* rngend = nullptr;  // [nsets][S]
  uint32_t* candk = nullptr;     // CSL_FLAG_KEEP_CANDIDATES: [slots][L][S][ccap_max]
  uint2* queue = nullptr;
  uint32_t* nbk = nullptr;
  uint32_t* bcnt = nullptr;
  uint32_t* bcur = nullptr;
  uint32_t* boff = nullptr;    // [nsets][S][nbmax+1]
  uint32_t* ticket = nullptr;  // [nsets][S][2]
  uint32_t nbmax = 0;
  size_t scatter_lds = 0;
  uint32_t* tcnt = nullptr;
  uint32_t* fsize = nullptr;
  // results
  csl_sample_meta* meta = nullptr;  // [slots][S]
  int* arena[CSL_MAX_LAYERS] = {};  // [slots][S][arena_stride[l]], int32
  size_t arena_stride[CSL_MAX_LAYERS];
  size_t list_base[CSL_MAX_LAYERS][CSL_NUM_LISTS];
  size_t list_cap[CSL_MAX_LAYERS][CSL_NUM_LISTS];
  // batch descriptors
  BatchDesc* desc_dev = nullptr;   // [slots][S]
  BatchDesc* desc_host = nullptr;  // pinned, [slots][S]
  // pinned staging for csl_fetch_sample
  int* fetch_stage = nullptr;       // pinned int32 staging of the sample handed out last
  int* fetch_stage2[2] = {nullptr, nullptr};  // two buffers: the caller may still read one while the next fills
  int* fetch_pack = nullptr;        // device: the sample's lists back to back (k_pack_sample)
  int fetch_flip = 0;
  long long* fetch_host = nullptr;  // widened copy handed to the caller
  size_t fetch_cap = 0;
  hipStream_t copy_stream = nullptr;
  // host mirror of meta for fetches
  std::vector<csl_sample_meta> meta_host;
  std::vector<char> meta_valid;  // per slot
  int64_t dev_bytes = 0;
  bool dirty = false;  // work submitted since last sync
  // timing
  bool timing = false;
  std::vector<TimedEvent> timed;
  std::vector<hipEvent_t> event_pool;
  double t_ms[CSL_NUM_KERNELS] = {};
  int64_t t_n[CSL_NUM_KERNELS] = {};
};

namespace {

template <typename T>
int dmalloc(csl_engine* e, T** p, size_t count) {
  size_t bytes = count * sizeof(T);
  if (bytes == 0) bytes = sizeof(T);
  hipError_t r = hipMalloc((void**)p, bytes);
  if (r != hipSuccess) return fail(CSL_E_NOMEM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(r));
  e->dev_bytes += (int64_t)bytes;
  return 0;
}
#define DMALLOC(p, n)                      \
  do {                                     \
    int _r = dmalloc(e, &(p), (n));        \
    if (_r) return _r;                     \
  } while (0)

hipEvent_t get_event(csl_engine* e) {
  if (!e->event_pool.empty()) {
    hipEvent_t ev = e->event_pool.back();
    e->event_pool.pop_back();
    return ev;
  }
  hipEvent_t ev;
  hipEventCreate(&ev);
  return ev;
}

struct Timed {
  csl_engine* e;
  TimedEvent te;
  bool on;
  Timed(csl_engine* e_, int kernel, hipStream_t st) : e(e_), on(e_->timing) {
    if (on) {
      te.kernel = kernel;
      te.a = get_event(e);
      te.b = get_event(e);
      hipEventRecord(te.a, st);
      stream = st;
    }
  }
  hipStream_t stream = nullptr;
  ~Timed() {
    if (on) {
      hipEventRecord(te.b, stream);
      e->timed.push_back(te);
    }
  }
};

int collect_timing(csl_engine* e) {
  for (auto& te : e->timed) {
    float ms = 0.f;
    HIPCHECK(hipEventSynchronize(te.b));
    HIPCHECK(hipEventElapsedTime(&ms, te.a, te.b));
    e->t_ms[te.kernel] += ms;
    e->t_n[te.kernel] += 1;
    e->event_pool.push_back(te.a);
    e->event_pool.push_back(te.b);
  }
  e->timed.clear();
  return 0;
}

// host init of the mt19937 state (ISO C++ [rand.eng.mers] seeding)
void mt_seed_host(uint32_t seed, uint32_t* mt) {
  mt[0] = seed;
  for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
}

int refresh_positions(csl_engine* e) {
  // exact stream positions; only valid when the device is idle
  std::vector<unsigned long long> tmp(e->S);
  HIPCHECK(hipMemcpy(tmp.data(), e->rngpos, sizeof(unsigned long long) * e->S, hipMemcpyDeviceToHost));
  for (int s = 0; s < e->S; s++) e->pos_ub[s] = e->pos_lb[s] = tmp[s];
  for (int k = 0; k < csl_engine::NSNAP; k++) e->snap_pending[k] = false;
  e->snap_next = e->rounds_submitted;
  return 0;
}

// Position snapshots are applied strictly in round order: round q's end positions become the streams' lower
// bounds only when the snapshots of ALL rounds <= q have completed, i.e. when no round that may still read
// ring words below them is in flight (rounds overlap on different HIP streams; later rounds only read at
// or above round q's end positions).
void apply_snapshot(csl_engine* e, int k) {
  const uint64_t after = e->rounds_submitted - (e->snap_round[k] + 1);  // rounds submitted since
  const unsigned long long* sn = e->snap_host + (size_t)k * e->S;
  for (int s = 0; s < e->S; s++) {
    if (sn[s] > e->pos_lb[s]) e->pos_lb[s] = sn[s];
    const uint64_t ub = sn[s] + after * e->worst_draws;
    if (ub < e->pos_ub[s] && ub >= e->pos_lb[s]) e->pos_ub[s] = ub;
  }
  e->snap_pending[k] = false;
  e->snap_next = e->snap_round[k] + 1;
}
int snap_slot_of(csl_engine* e, uint64_t round) {
  const int k = (int)(round % csl_engine::NSNAP);
  return (e->snap_pending[k] && e->snap_round[k] == round) ? k : -1;
}
void poll_snapshots(csl_engine* e) {
  while (e->snap_next < e->rounds_submitted) {
    const int k = snap_slot_of(e, e->snap_next);
    if (k < 0) {  // (dropped by refresh_positions)
      e->snap_next++;
      continue;
    }
    if (hipEventQuery(e->snap_ev[k]) != hipSuccess) break;
    apply_snapshot(e, k);
  }
}
// blocking form: everything up to and including `round` (a snapshot buffer is about to be reused)
int retire_snapshots(csl_engine* e, uint64_t round) {
  while (e->snap_next <= round) {
    const int k = snap_slot_of(e, e->snap_next);
    if (k < 0) {
      e->snap_next++;
      continue;
    }
    HIPCHECK(hipEventSynchronize(e->snap_ev[k]));
    apply_snapshot(e, k);
  }
  return 0;
}

hipEvent_t rng_get_event(csl_engine* e) {
  if (!e->rng_event_pool.empty()) {
    hipEvent_t ev = e->rng_event_pool.back();
    e->rng_event_pool.pop_back();
    return ev;
  }
  hipEvent_t ev;
  hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  return ev;
}

void rng_bounds(csl_engine* e, uint64_t* need_hi, uint64_t* lo) {
  *need_hi = 0;
  *lo = UINT64_MAX;
  for (int s = 0; s < e->S; s++) {
    if (e->pos_ub[s] + e->worst_draws > *need_hi) *need_hi = e->pos_ub[s] + e->worst_draws;
    if (e->pos_lb[s] < *lo) *lo = e->pos_lb[s];
  }
}

// Keep the ring filled up to LOOKAHEAD beyond what the next round may draw, in chunks on the rng
// stream; hand back the event of the chunk that covers the next round (`*wait`, may be null when
// that chunk is known complete) and the position up to which the round may read (`*safe_hi`).
// The generator only ever overwrites words below every stream's lower bound.
int ensure_rng(csl_engine* e, hipEvent_t* wait, uint64_t* safe_hi) {
  poll_snapshots(e);
  uint64_t need_hi, lo;
  rng_bounds(e, &need_hi, &lo);
  uint64_t lookahead = 4 * e->worst_draws;
  if (lookahead < (1ull << 20)) lookahead = 1ull << 20;
  if (lookahead > e->ring_words / 4) lookahead = e->ring_words / 4;
  uint64_t chunk = 2 * e->worst_draws;
  if (chunk < (1ull << 18)) chunk = 1ull << 18;
  if (chunk > e->ring_words / 8) chunk = e->ring_words / 8;
  chunk = (chunk / 624 + 1) * 624;
  for (int attempt = 0; attempt < 2; attempt++) {
    while (e->gen_hi < need_hi + lookahead) {
      uint64_t target = e->gen_hi + chunk;
      if (target > lo + e->ring_words) target = lo + e->ring_words;
      const uint32_t nblocks = target > e->gen_hi ? (uint32_t)((target - e->gen_hi) / 624) : 0;  // round down
      if (nblocks == 0) break;  // the ring is full relative to the slowest stream
      {
        Timed t(e, KN_MT, e->rng_stream);
        launch_mt_fill(e->rng_stream, e->mt_state, e->ring, e->ring_words, e->gen_hi, nblocks);
      }
      HIPCHECK(hipGetLastError());
      e->gen_hi += (uint64_t)nblocks * 624ull;
      csl_engine::RngChunk c;
      c.hi = e->gen_hi;
      c.ev = rng_get_event(e);
      HIPCHECK(hipEventRecord(c.ev, e->rng_stream));
      e->rng_chunks.push_back(c);
    }
    if (e->gen_hi >= need_hi) break;
    if (attempt == 1)
      return fail(CSL_E_INVALID, "mt19937 ring too small: streams span %llu words, ring %llu (raise rng_ring_log2)",
                  (unsigned long long)(need_hi - lo), (unsigned long long)e->ring_words);
    // the bounds drifted apart (or the ring is tight): settle everything and take exact positions
    for (int k = 0; k < e->nsets; k++) HIPCHECK(hipStreamSynchronize(e->streams[k]));
    e->dirty = false;
    int r = collect_timing(e);
    if (r) return r;
    r = refresh_positions(e);
    if (r) return r;
    rng_bounds(e, &need_hi, &lo);
  }
  // retire chunks nobody can need any more, find the one this round waits for
  while (e->rng_chunks.size() > 1 && e->rng_chunks.front().hi <= lo &&
         hipEventQuery(e->rng_chunks.front().ev) == hipSuccess) {
    e->rng_event_pool.push_back(e->rng_chunks.front().ev);
    e->rng_chunks.pop_front();
  }
  *wait = nullptr;
  *safe_hi = e->gen_hi;
  for (const auto& c : e->rng_chunks) {
    if (c.hi >= need_hi) {
      *wait = c.ev;
      *safe_hi = c.hi;
      break;
    }
  }
  return 0;
}

int run_round(csl_engine* e, const long long* nodes_dev, int32_t n_batches, int32_t slot) {
  const int S = e->S, L = e->L;
  hipEvent_t rng_wait = nullptr;
  uint64_t rng_safe_hi = 0;
  int r = ensure_rng(e, &rng_wait, &rng_safe_hi);
  if (r) return r;
  // per-kernel timing wants undisturbed kernels: timed rounds all run on stream 0
  const int set = (e->nsets > 1 && !e->timing) ? (slot % e->nsets) : 0;
  hipStream_t st = e->streams[set];
  const size_t sF = (size_t)set * S * e->fcap_max, sC = (size_t)set * S * e->ccap_max;
  uint32_t* fsize = e->fsize + (size_t)set * S * (CSL_MAX_LAYERS + 1);
  // wait for the generator chunk that covers this round's draws (requested rounds ago, normally done)
  if (rng_wait) HIPCHECK(hipStreamWaitEvent(st, rng_wait, 0));
  // the generator may only overwrite ring words below every stream's position
  const unsigned long long gen_lo = e->gen_hi > e->ring_words ? e->gen_hi - e->ring_words : 0;
  BatchDesc* dd = e->desc_dev + (size_t)slot * S;
  HIPCHECK(hipMemcpyAsync(dd, e->desc_host + (size_t)slot * S, sizeof(BatchDesc) * S, hipMemcpyHostToDevice, st));
  HIPCHECK(hipEventRecord(e->desc_event[slot], st));
  e->desc_inflight[slot] = 1;
  csl_sample_meta* meta = e->meta + (size_t)slot * S;
  HIPCHECK(hipMemsetAsync(meta, 0, sizeof(csl_sample_meta) * (size_t)S, st));  // error bits and stale layer tables
  LArgs A[CSL_MAX_LAYERS];
  for (int l = 0; l < L; l++) {
    LArgs& a = A[l];
    memset(&a, 0, sizeof(a));
    a.rowinfo = e->rowinfo;
    a.off32 = e->off32;
    a.indices = e->indices;
    a.wl = e->wl;
    a.N = e->N;
    a.P = (uint32_t)e->P;
    a.ring = e->ring;
    a.ring_mask = e->ring_words - 1;
    a.gen_lo = gen_lo;
    a.gen_hi = rng_safe_hi;
    a.rngpos = e->rngpos;
    a.rngbase = e->rngbase + (size_t)set * S;
    a.acc = e->acc;
    a.fr_in = e->fr[l] + (size_t)slot * S * e->fcap[l];
    a.fr_out = e->fr[l + 1] + (size_t)slot * S * e->fcap[l + 1];
    a.fr_in_stride = e->fcap[l];
    a.fr_out_stride = e->fcap[l + 1];
    a.fr_out_cap = (uint32_t)e->fcap[l + 1];
    a.ninfo = e->ninfo + sF;
    a.hasedge = e->hasedge + sF;
    a.selfpos = e->selfpos + sF;
    a.firstpos = e->firstpos + sF;
    a.fcap = e->fcap_max;
    a.cand = e->cand + sC;
    a.cflag = e->cflag + sC;
    a.crank = e->crank + sC;
    a.queue = e->queue + sC;
    a.ccap = e->ccap_max;
    a.nbk = e->nbk + (size_t)set * S;
    a.bcnt = e->bcnt + (size_t)set * S * (e->nbmax + 1);
    a.bcur = e->bcur + (size_t)set * S * e->nbmax;
    a.nbmax = e->nbmax;
    a.tcnt = e->tcnt + (size_t)set * S * e->nk * e->tmax;
    a.tmax = e->tmax;
    a.nk = e->nk;
    a.fsize = fsize;
    a.meta = meta;
    a.arena = e->arena[l] + (size_t)slot * S * e->arena_stride[l];
    a.arena_stride = e->arena_stride[l];
    for (int k = 0; k < CSL_NUM_LISTS; k++) a.list_base[k] = e->list_base[l][k];
    a.layer = (uint32_t)l;
    a.fanout = (uint32_t)e->cfg.fanout[l];
    a.W = a.fanout + 1;
    a.wmagic = ((1ull << 40) + a.W - 1) / a.W;
    a.S = (uint32_t)S;
    a.last = l == L - 1 ? 1u : 0u;
    a.pmask = e->cfg.part_mask ? e->cfg.part_mask : 0xFFFFFFFFu;
    a.pmagic = e->P > 1 ? (uint32_t)((1ull << 32) / (unsigned)e->P) : 0xFFFFFFFFu;  // (P == 1: q = v - 1, r = 1 -> 0)
    a.tpb = (unsigned)((e->fcap[l] + TN - 1) / TN) > 128 ? TPB : 1;  // small layers are latency-bound: one tile per block
    a.boff = e->boff + (size_t)set * S * (e->nbmax + 1);
    a.ticket = e->ticket + (size_t)set * S * 2;
    a.graph = e->cfg.mode == CSL_MODE_GRAPH ? 1u : 0u;
    a.ecnt = e->ecnt ? e->ecnt + sF * e->P : nullptr;
    a.srcpos = e->srcpos ? e->srcpos + sC : nullptr;
    a.tcur = (e->tcur && (l < L - 1 || (e->cfg.flags & CSL_FLAG_TRANSPOSE_ALL))) ? e->tcur + sC : nullptr;
    a.dupflag = e->dupflag + (size_t)set * S;
    a.fcap0 = e->fcap[0];
    a.seedrep = e->seedrep + (size_t)set * S * e->fcap[0];
    a.dupfirst = e->dupfirst ? e->dupfirst + (size_t)set * S * e->fcap[0] * e->P : nullptr;
    a.dupout = e->dupout ? e->dupout + (size_t)set * S * e->fcap[0] * e->P : nullptr;
    a.rngend = e->rngend + (size_t)set * S;
    a.candk = e->candk ? e->candk + ((size_t)slot * L + l) * S * e->ccap_max : nullptr;
  }
  const dim3 blk(TN);
  const unsigned xg = 8u * (unsigned)((S + 7) / 8);  // see xcd_block()
  auto tiles_of = [&](int l) { return (unsigned)((e->fcap[l] + TN - 1) / TN); };
  auto degree_blocks = [&](int l) { return xg * ((tiles_of(l) + NW - 1) / NW); };
  // A round is 6 launches per layer (+ k_graph / k_dupseeds): k_degree [batch copy on layer 0, row lookups, its last
  // block per stream scans the rng consumers] . k_sample . k_scatter [bucket offsets scanned in place] . k_bucket .
  // k_count [last block per stream: list offsets] . k_emit . k_selfin -- the latter in ONE launch with the next
  // layer's k_degree (both depend on k_emit only).
  // The stream's mt19937 position is handed from round to round by k_degree's scan: this round's first one waits
  // for the previous round's last one.
  if (e->nsets > 1 && e->chain_valid) HIPCHECK(hipStreamWaitEvent(st, e->chain_event, 0));
  {
    Timed t(e, KN_DEGREE, st);
    hipLaunchKernelGGL(k_degree, dim3(degree_blocks(0)), blk, 0, st, A[0], nodes_dev, (const BatchDesc*)dd);
  }
  for (int l = 0; l < L; l++) {
    const LArgs& a = A[l];
    const unsigned tiles_in = tiles_of(l);
    const size_t ccap_l = (size_t)tiles_in * TN * a.W;
    const dim3 grid_in(xg * tiles_in);
    const dim3 grid_sample(xg * ((tiles_in + a.tpb - 1) / a.tpb));
    const dim3 grid_scatter(xg * (unsigned)((ccap_l + SCT - 1) / SCT));
    unsigned nb_l = (unsigned)((ccap_l + QMEAN - 1) / QMEAN);
    if (nb_l > e->nbmax) nb_l = e->nbmax;
    const dim3 grid_bucket(xg * ((nb_l + BPB - 1) / BPB));
    const size_t lds_hist = (size_t)e->nbmax * sizeof(uint32_t);
    if (l == L - 1 && e->nsets > 1) {
      // (recorded after the launch that holds the round's last position update: the last layer's k_degree)
      HIPCHECK(hipEventRecord(e->chain_event, st));
      e->chain_valid = true;
    }
    {
      Timed t(e, KN_SAMPLE, st);
      // CSL_SAMPLE_LDS_PAD=<bytes>: measurement knob (profiles/r3_sample_occupancy): what k_sample costs at the occupancy
      // an LDS-staged counting sort inside it (k_scatter folded in: 32 KB per workgroup) would leave it
      static const size_t lds_pad = getenv("CSL_SAMPLE_LDS_PAD") ? (size_t)atol(getenv("CSL_SAMPLE_LDS_PAD")) : 0;
      hipLaunchKernelGGL(k_sample, grid_sample, blk, lds_hist + lds_pad, st, a);
    }
    if (l == L - 1) {
      // Snapshot of the streams' positions AFTER this round (tightens the host's bounds without a sync).
      // It is taken from the scratch set's own copy and only after the round's last k_sample: once the
      // snapshot's event has completed, nothing of this round reads the ring any more, so -- applied in
      // round order (poll_snapshots) -- these positions are the floor below which the generator may recycle.
      const int k = (int)(e->rounds_submitted % csl_engine::NSNAP);
      if (e->snap_pending[k]) {
        int r2 = retire_snapshots(e, e->snap_round[k]);
        if (r2) return r2;
      }
      HIPCHECK(hipMemcpyAsync(e->snap_host + (size_t)k * S, a.rngend, sizeof(unsigned long long) * S,
                              hipMemcpyDeviceToHost, st));
      HIPCHECK(hipEventRecord(e->snap_ev[k], st));
      e->snap_round[k] = e->rounds_submitted;
      e->snap_pending[k] = true;
    }
    {
      Timed t(e, KN_SCATTER, st);
      hipLaunchKernelGGL(k_scatter, grid_scatter, blk, e->scatter_lds, st, a);
    }
    {
      Timed t(e, KN_BUCKET, st);
      static const bool probe = getenv("CSL_WAVE_DUP_PROBE") != nullptr;
      if (probe) hipLaunchKernelGGL((k_bucket<false, true>), grid_bucket, dim3(BT), 0, st, a);   // (v % P owners only)
      else if (a.wl) hipLaunchKernelGGL(k_bucket<true>, grid_bucket, dim3(BT), 0, st, a);
      else hipLaunchKernelGGL(k_bucket<false>, grid_bucket, dim3(BT), 0, st, a);
    }
    {
      Timed t(e, KN_COUNT, st);
      hipLaunchKernelGGL(k_count, dim3(xg * ((tiles_in + CNT_T / 64 - 1) / (CNT_T / 64))), dim3(CNT_T), 0, st, a);
    }
    {
      Timed t(e, KN_EMIT, st);
      hipLaunchKernelGGL(k_emit, grid_in, blk, 0, st, a);
    }
    if (a.graph) {
      Timed t(e, KN_EDGES, st);
      hipLaunchKernelGGL(k_graph, grid_in, blk, 0, st, a);
    } else if (l == 0) {
      // repeated seed ids (bipartite.cpp:3-17): returns at once for a stream whose seeds are distinct
      Timed t(e, KN_DUPSEEDS, st);
      hipLaunchKernelGGL(k_dupseeds, dim3(S), dim3(DS_T), 0, st, a);
    }
    const unsigned n_selfin = xg * ((tiles_in + 3) / 4);
    if (l + 1 < L) {
      Timed t(e, KN_SELFIN_DEGREE, st);
      const unsigned n_degree = degree_blocks(l + 1);
      hipLaunchKernelGGL(k_selfin_degree, dim3(n_degree + n_selfin), blk, 0, st, a, A[l + 1], n_degree);
    } else {
      Timed t(e, KN_SELFIN, st);
      hipLaunchKernelGGL(k_selfin, dim3(n_selfin), blk, 0, st, a);
    }
    // (after k_selfin: the self entries come from self_ids_in)
    if (a.tcur) {
      const unsigned sp = (unsigned)(S * e->P);
      const unsigned in_cap = (unsigned)ccap_l;  // in nodes of a slice <= candidates of the layer
      const unsigned ttiles = (in_cap + TT - 1) / TT;
      uint32_t* tsum = e->tsum + (size_t)set * S * e->P * e->ttiles_max;
      hipLaunchKernelGGL(k_tsum, dim3(ttiles, sp), blk, 0, st, a, tsum, (uint32_t)e->ttiles_max);
      hipLaunchKernelGGL(k_tptr, dim3(ttiles, sp), blk, 0, st, a, (const uint32_t*)tsum, (uint32_t)e->ttiles_max);
      // (a slice has at most list_cap[OUT_NODES] out rows, and no more self entries than that)
      hipLaunchKernelGGL(k_tfill, dim3((unsigned)((e->list_cap[l][CSL_OUT_NODES] + TN - 1) / TN), sp), blk, 0, st, a);
      hipLaunchKernelGGL(k_tsort, dim3((in_cap + TN - 1) / TN, sp), blk, 0, st, a);
    }
  }
  HIPCHECK(hipGetLastError());
  for (int s = 0; s < n_batches && s < S; s++) e->pos_ub[s] += e->worst_draws;
  e->rounds_submitted++;
  HIPCHECK(hipEventRecord(e->slot_event[slot], st));
  e->slot_pending[slot] = 1;
  e->dirty = true;
  e->meta_valid[slot] = 0;
  return 0;
}

}  // namespace

extern "C" {

const char* csl_last_error(void) { return g_err; }
int csl_abi_version(void) { return CSL_ABI_VERSION; }
const char* csl_kernel_name(int32_t k) { return (k >= 0 && k < CSL_NUM_KERNELS) ? kKernelNames[k] : ""; }

int csl_debug_wave_duplicates(uint64_t* out3) {
  if (!out3) return CSL_E_INVALID;
  unsigned long long v[3];
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(v, HIP_SYMBOL(g_wave_dup), sizeof(v)) != hipSuccess)
    return CSL_E_HIP;
  for (int i = 0; i < 3; i++) out3[i] = v[i];
  return CSL_OK;
}

void csl_destroy(csl_engine* e) {
  if (!e) return;
  hipSetDevice(e->cfg.device);
  for (int k = 0; k < CSL_MAX_SETS; k++)
    if (e->streams[k]) hipStreamSynchronize(e->streams[k]);
  if (e->rng_stream) hipStreamSynchronize(e->rng_stream);
  for (auto& te : e->timed) {
    hipEventDestroy(te.a);
    hipEventDestroy(te.b);
  }
  for (auto ev : e->event_pool) hipEventDestroy(ev);
  void* ptrs[] = {e->rowinfo, e->off32, e->indices, e->wl,      e->nodes, e->seedbuf, e->ring,  e->mt_state, e->rngpos,
                  e->rngbase, e->ninfo,   e->hasedge, e->selfpos, e->firstpos, e->cand, e->cflag, e->crank,
                  e->queue,   e->nbk,     e->bcnt,    e->bcur,    e->tcnt,     e->fsize, e->meta, e->desc_dev,
                  e->ecnt,    e->srcpos,  e->tcur, e->tsum, e->acc,     e->dupflag, e->seedrep, e->dupfirst, e->dupout, e->rngend,
                  e->candk,   e->boff,    e->ticket};
  for (void* p : ptrs)
    if (p) hipFree(p);
  for (int l = 0; l <= CSL_MAX_LAYERS; l++)
    if (e->fr[l]) hipFree(e->fr[l]);
  for (int l = 0; l < CSL_MAX_LAYERS; l++)
    if (e->arena[l]) hipFree(e->arena[l]);
  if (e->desc_host) hipHostFree(e->desc_host);
  for (int k = 0; k < 2; k++)
    if (e->fetch_stage2[k]) hipHostFree(e->fetch_stage2[k]);
  if (e->fetch_pack) hipFree(e->fetch_pack);
  free(e->fetch_host);
  if (e->copy_stream) hipStreamDestroy(e->copy_stream);
  if (e->rng_event) hipEventDestroy(e->rng_event);
  if (e->chain_event) hipEventDestroy(e->chain_event);
  for (auto& c : e->rng_chunks) hipEventDestroy(c.ev);
  for (auto ev : e->rng_event_pool) hipEventDestroy(ev);
  for (int k = 0; k < csl_engine::NSNAP; k++)
    if (e->snap_ev[k]) hipEventDestroy(e->snap_ev[k]);
  if (e->snap_host) hipHostFree(e->snap_host);
  for (auto ev : e->slot_event) hipEventDestroy(ev);
  for (auto ev : e->desc_event) hipEventDestroy(ev);
  for (int k = 0; k < CSL_MAX_SETS; k++)
    if (e->streams[k]) hipStreamDestroy(e->streams[k]);
  if (e->rng_stream) hipStreamDestroy(e->rng_stream);
  delete e;
}

static int create_impl(const csl_config* cfg, csl_engine* e) {
  e->cfg = *cfg;
  const int S = e->S = cfg->n_streams, P = e->P = cfg->n_parts, L = e->L = cfg->n_layers;
  e->slots = cfg->n_slots;
  e->N = (uint32_t)cfg->num_nodes;
  e->E = (size_t)cfg->num_edges;
  HIPCHECK(hipSetDevice(cfg->device));
  e->nsets = (cfg->flags & CSL_FLAG_SERIAL_ROUNDS) ? 1 : (cfg->n_slots < CSL_MAX_SETS ? cfg->n_slots : CSL_MAX_SETS);
  for (int k = 0; k < e->nsets; k++) HIPCHECK(hipStreamCreateWithFlags(&e->streams[k], hipStreamNonBlocking));
  e->stream = e->streams[0];
  HIPCHECK(hipStreamCreateWithFlags(&e->rng_stream, hipStreamNonBlocking));
  HIPCHECK(hipEventCreateWithFlags(&e->rng_event, hipEventDisableTiming));
  HIPCHECK(hipEventCreateWithFlags(&e->chain_event, hipEventDisableTiming));
  const size_t N = e->N;
  // ---- graph upload + packing (int64 CSR -> rowinfo / u32 indices)
  const bool small_offsets = e->E < (1ull << 32) && !getenv("CSLICER_ROWINFO64");  // (env: force the 8-byte table)
  if (small_offsets) DMALLOC(e->off32, N + 1);
  else DMALLOC(e->rowinfo, N);
  DMALLOC(e->indices, e->E);
  {
    long long* tmp = nullptr;
    const size_t chunk = (size_t)64 << 20;  // elements per staging chunk
    const size_t tmp_n = (N + 1 > chunk ? N + 1 : chunk);
    hipError_t r = hipMalloc((void**)&tmp, tmp_n * sizeof(long long));
    if (r != hipSuccess) return fail(CSL_E_NOMEM, "staging hipMalloc failed: %s", hipGetErrorString(r));
    hipError_t er = hipMemcpy(tmp, cfg->indptr, (N + 1) * sizeof(long long), hipMemcpyHostToDevice);
    if (er == hipSuccess) {
      if (small_offsets) hipLaunchKernelGGL(k_pack_off32, dim3(1024), dim3(256), 0, e->stream, tmp, e->off32, N + 1);
      else hipLaunchKernelGGL(k_pack_rows, dim3(1024), dim3(256), 0, e->stream, tmp, e->rowinfo, N);
      er = hipStreamSynchronize(e->stream);
    }
    for (size_t o = 0; er == hipSuccess && o < e->E; o += chunk) {
      const size_t n = e->E - o < chunk ? e->E - o : chunk;
      er = hipMemcpy(tmp, cfg->indices + o, n * sizeof(long long), hipMemcpyHostToDevice);
      if (er != hipSuccess) break;
      hipLaunchKernelGGL(k_pack_indices, dim3(2048), dim3(256), 0, e->stream, tmp, e->indices + o, n);
      er = hipStreamSynchronize(e->stream);
    }
    if (er == hipSuccess && cfg->workload) {
      DMALLOC(e->wl, N);
      er = hipMemcpy(tmp, cfg->workload, N * sizeof(int), hipMemcpyHostToDevice);
      if (er == hipSuccess) {
        hipLaunchKernelGGL(k_pack_wl, dim3(1024), dim3(256), 0, e->stream, (const int*)tmp, e->wl, N);
        er = hipStreamSynchronize(e->stream);
      }
    }
    hipFree(tmp);
    if (er != hipSuccess) return fail(CSL_E_HIP, "graph upload failed: %s", hipGetErrorString(er));
  }
  // ---- capacities
  e->fcap[0] = (size_t)cfg->max_batch;
  e->worst_draws = 0;
  for (int l = 0; l < L; l++) {
    size_t worst = e->fcap[l] * (size_t)(cfg->fanout[l] + 1);
    if (worst > N) worst = N;
    size_t cap = cfg->frontier_cap[l + 1] > 0 ? (size_t)cfg->frontier_cap[l + 1] : worst;
    if (cap > worst) cap = worst;
    e->fcap[l + 1] = cap;
    e->worst_draws += (uint64_t)e->fcap[l] * (uint64_t)cfg->fanout[l];
  }
  e->fcap_max = 0;
  e->ccap_max = 0;
  for (int l = 0; l < L; l++) {
    if (e->fcap[l] > e->fcap_max) e->fcap_max = e->fcap[l];
    const size_t c = ((e->fcap[l] + TN - 1) / TN) * TN * (size_t)(cfg->fanout[l] + 1);
    if (c > e->ccap_max) e->ccap_max = c;
    if (c >= 0x80000000ull) return fail(CSL_E_INVALID, "candidate positions exceed 31 bits");
  }
  e->tmax = (uint32_t)((e->fcap_max + TN - 1) / TN);
  e->nk = (uint32_t)NKINDS(P, cfg->mode == CSL_MODE_GRAPH);
  // ---- per-stream scratch
  for (int l = 0; l <= L; l++) DMALLOC(e->fr[l], (size_t)e->slots * S * e->fcap[l]);  // [slot][S][fcap]
  DMALLOC(e->ninfo, e->nsets * (size_t)S * e->fcap_max);
  DMALLOC(e->hasedge, e->nsets * (size_t)S * e->fcap_max);
  DMALLOC(e->selfpos, e->nsets * (size_t)S * e->fcap_max);
  DMALLOC(e->firstpos, e->nsets * (size_t)S * e->fcap_max);
  DMALLOC(e->cand, e->nsets * (size_t)S * e->ccap_max);
  DMALLOC(e->cflag, e->nsets * (size_t)S * e->ccap_max);
  DMALLOC(e->crank, e->nsets * (size_t)S * e->ccap_max);
  if (cfg->mode == CSL_MODE_GRAPH) {
    DMALLOC(e->ecnt, e->nsets * (size_t)S * e->fcap_max * P);
    DMALLOC(e->srcpos, e->nsets * (size_t)S * e->ccap_max);
    if ((cfg->flags & CSL_FLAG_TRANSPOSE) && (L > 1 || (cfg->flags & CSL_FLAG_TRANSPOSE_ALL))) {
      DMALLOC(e->tcur, e->nsets * (size_t)S * e->ccap_max);
      e->ttiles_max = (e->ccap_max + TT - 1) / TT;
      DMALLOC(e->tsum, e->nsets * (size_t)S * P * e->ttiles_max);
    }
  }
  DMALLOC(e->queue, e->nsets * (size_t)S * e->ccap_max);
  DMALLOC(e->dupflag, e->nsets * (size_t)S);
  DMALLOC(e->seedrep, e->nsets * (size_t)S * e->fcap[0]);
  if (cfg->mode == CSL_MODE_STRICT) {
    DMALLOC(e->dupfirst, e->nsets * (size_t)S * e->fcap[0] * P);
    DMALLOC(e->dupout, e->nsets * (size_t)S * e->fcap[0] * P);
  }
  DMALLOC(e->rngend, e->nsets * (size_t)S);
  HIPCHECK(hipMemsetAsync(e->rngend, 0, sizeof(unsigned long long) * e->nsets * S, e->stream));
  if (cfg->flags & CSL_FLAG_KEEP_CANDIDATES) DMALLOC(e->candk, (size_t)e->slots * L * S * e->ccap_max);
  e->nbmax = (uint32_t)((e->ccap_max + QMEAN - 1) / QMEAN);
  if (e->nbmax < 1) e->nbmax = 1;
  if (e->nbmax > 8192)
    return fail(CSL_E_INVALID, "a layer may hold %zu candidates per minibatch; this build supports %d",
                e->ccap_max, 8192 * QMEAN);
  e->scatter_lds = (((size_t)3 * e->nbmax + 1) & ~(size_t)1) * sizeof(uint32_t) + (size_t)SCT * sizeof(uint2);
  if (e->scatter_lds > 150 * 1024) return fail(CSL_E_INVALID, "k_scatter needs %zu bytes of LDS", e->scatter_lds);
  HIPCHECK(hipFuncSetAttribute((const void*)k_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->scatter_lds));
  DMALLOC(e->nbk, e->nsets * (size_t)S);
  DMALLOC(e->bcnt, e->nsets * (size_t)S * (e->nbmax + 1));
  DMALLOC(e->bcur, e->nsets * (size_t)S * e->nbmax);
  DMALLOC(e->boff, e->nsets * (size_t)S * (e->nbmax + 1));
  DMALLOC(e->ticket, e->nsets * (size_t)S * 2);
  HIPCHECK(hipMemsetAsync(e->ticket, 0, sizeof(uint32_t) * e->nsets * S * 2, e->stream));
  DMALLOC(e->tcnt, e->nsets * (size_t)S * e->nk * e->tmax);
  DMALLOC(e->fsize, e->nsets * (size_t)S * (CSL_MAX_LAYERS + 1));
  DMALLOC(e->rngpos, (size_t)S);
  DMALLOC(e->rngbase, e->nsets * (size_t)S);
  DMALLOC(e->acc, (size_t)2 * S);
  HIPCHECK(hipMemsetAsync(e->acc, 0, sizeof(unsigned long long) * 2 * S, e->stream));
  HIPCHECK(hipMemsetAsync(e->rngpos, 0, sizeof(unsigned long long) * S, e->stream));
  HIPCHECK(hipMemsetAsync(e->fsize, 0, sizeof(uint32_t) * e->nsets * S * (CSL_MAX_LAYERS + 1), e->stream));
  // ---- result arenas
  DMALLOC(e->meta, (size_t)e->slots * S);
  HIPCHECK(hipMemsetAsync(e->meta, 0, sizeof(csl_sample_meta) * e->slots * S, e->stream));
  for (int l = 0; l < L; l++) {
    const size_t F = e->fcap[l], f = (size_t)cfg->fanout[l];
    const size_t edges = F * f;
    size_t outs = F * (size_t)(P < (int)f ? P : (int)f);
    if (outs > edges) outs = edges;
    size_t cap[CSL_NUM_LISTS];
    cap[CSL_IN_NODES] = edges;
    cap[CSL_OUT_NODES] = outs;
    cap[CSL_OWNED_OUT_NODES] = F;
    cap[CSL_SELF_IDS_IN] = F;
    cap[CSL_SELF_IDS_OUT] = F;
    cap[CSL_TO_IDS] = F;
    cap[CSL_FROM_IDS] = outs;
    cap[CSL_INDPTR] = cap[CSL_INDICES] = cap[CSL_OWNED_DEGREE] = cap[CSL_T_INDPTR] = cap[CSL_T_INDICES] = 0;
    if (cfg->mode == CSL_MODE_GRAPH) {
      // every owned node is an in node and an out node of its slice
      size_t outs_g = F * (size_t)(P < (int)f + 1 ? P : (int)f + 1);
      if (outs_g > edges + F) outs_g = edges + F;
      cap[CSL_IN_NODES] = edges + F;
      cap[CSL_OUT_NODES] = outs_g;
      cap[CSL_TO_IDS] = outs;
      cap[CSL_FROM_IDS] = outs;
      cap[CSL_INDPTR] = outs_g + (size_t)P;
      cap[CSL_INDICES] = edges;
      cap[CSL_OWNED_DEGREE] = F;
      if ((cfg->flags & CSL_FLAG_TRANSPOSE) && (l < L - 1 || (cfg->flags & CSL_FLAG_TRANSPOSE_ALL))) {
        cap[CSL_T_INDPTR] = edges + F + (size_t)P;
        cap[CSL_T_INDICES] = edges + F;
      }
    }
    size_t o = 0;
    for (int k = 0; k < CSL_NUM_LISTS; k++) {
      e->list_base[l][k] = o;
      e->list_cap[l][k] = cap[k];
      o += (cap[k] + 3) & ~(size_t)3;  // keep 16-byte alignment of every list
    }
    e->arena_stride[l] = o;
    DMALLOC(e->arena[l], (size_t)e->slots * S * o);
  }
  DMALLOC(e->desc_dev, (size_t)e->slots * S);
  HIPCHECK(hipHostMalloc((void**)&e->desc_host, sizeof(BatchDesc) * e->slots * S, hipHostMallocDefault));
  e->meta_host.resize((size_t)e->slots * S);
  e->meta_valid.assign(e->slots, 0);
  e->slot_event.resize(e->slots);
  e->slot_pending.assign(e->slots, 0);
  e->desc_event.resize(e->slots);
  e->desc_inflight.assign(e->slots, 0);
  for (int k = 0; k < e->slots; k++) {
    HIPCHECK(hipEventCreateWithFlags(&e->slot_event[k], hipEventDisableTiming));
    HIPCHECK(hipEventCreateWithFlags(&e->desc_event[k], hipEventDisableTiming));
  }
  // ---- rng
  const uint32_t lg = cfg->rng_ring_log2 ? cfg->rng_ring_log2 : 26;
  if (lg > 31) return fail(CSL_E_INVALID, "rng_ring_log2=%u: at most 31", lg);
  e->ring_words = 1ull << lg;
  if (e->ring_words < 4 * e->worst_draws + 2 * 624)
    return fail(CSL_E_INVALID, "rng_ring_log2=%u too small: one round may draw %llu words", lg,
                (unsigned long long)e->worst_draws);
  DMALLOC(e->ring, e->ring_words);
  DMALLOC(e->mt_state, 624);
  {
    uint32_t st[624];
    mt_seed_host(cfg->rng_seed, st);
    HIPCHECK(hipMemcpy(e->mt_state, st, sizeof(st), hipMemcpyHostToDevice));
  }
  e->pos_ub.assign(S, 0);
  e->pos_lb.assign(S, 0);
  e->gen_hi = 0;
  HIPCHECK(hipHostMalloc((void**)&e->snap_host, sizeof(unsigned long long) * csl_engine::NSNAP * S, hipHostMallocDefault));
  for (int k = 0; k < csl_engine::NSNAP; k++) HIPCHECK(hipEventCreateWithFlags(&e->snap_ev[k], hipEventDisableTiming));
  HIPCHECK(hipStreamSynchronize(e->stream));
  return 0;
}

int csl_create(const csl_config* cfg, csl_engine** out) {
  if (!cfg || !out) return fail(CSL_E_INVALID, "null argument");
  *out = nullptr;
  if (cfg->abi_version != CSL_ABI_VERSION) return fail(CSL_E_INVALID, "abi_version mismatch");
  if (cfg->n_parts < 1 || cfg->n_parts > CSL_MAX_PARTS) return fail(CSL_E_INVALID, "n_parts must be 1..%d", CSL_MAX_PARTS);
  if (cfg->n_layers < 1 || cfg->n_layers > CSL_MAX_LAYERS) return fail(CSL_E_INVALID, "n_layers must be 1..%d", CSL_MAX_LAYERS);
  if (cfg->n_streams < 1 || cfg->n_streams > 1024) return fail(CSL_E_INVALID, "n_streams must be 1..1024");
  if (cfg->n_slots < 1 || cfg->n_slots > 16) return fail(CSL_E_INVALID, "n_slots must be 1..16");
  if (cfg->max_batch < 1) return fail(CSL_E_INVALID, "max_batch must be >= 1");
  if (cfg->mode != CSL_MODE_STRICT && cfg->mode != CSL_MODE_GRAPH) return fail(CSL_E_INVALID, "unknown mode %d", cfg->mode);
  if ((cfg->flags & CSL_FLAG_TRANSPOSE) && cfg->mode != CSL_MODE_GRAPH)
    return fail(CSL_E_INVALID, "CSL_FLAG_TRANSPOSE needs CSL_MODE_GRAPH (strict mode has no edges to transpose)");
  if ((cfg->flags & CSL_FLAG_TRANSPOSE_ALL) && !(cfg->flags & CSL_FLAG_TRANSPOSE))
    return fail(CSL_E_INVALID, "CSL_FLAG_TRANSPOSE_ALL only widens CSL_FLAG_TRANSPOSE");
  if (cfg->part_mask >> cfg->n_parts) return fail(CSL_E_INVALID, "part_mask 0x%x names parts beyond n_parts", cfg->part_mask);
  if (!cfg->indptr || (!cfg->indices && cfg->num_edges > 0)) return fail(CSL_E_INVALID, "graph arrays missing");
  // the reference keeps ids in `int` (bipartite.h:55): node ids are only defined below 2^31.  Row offsets:
  // the reference's `int offset` (slicer.cpp:9) wraps at 2^31 edges, i.e. its behaviour is undefined beyond;
  // here offsets are 64-bit up to the 40 bits the packed row table holds (u32 offsets below 2^32 edges,
  // (offset << 24 | degree) above), so graphs the reference cannot index still slice as it meant to.
  if (cfg->num_nodes < 1 || cfg->num_nodes >= (1ll << 31)) return fail(CSL_E_INVALID, "num_nodes must be in [1, 2^31)");
  if (cfg->num_edges < 0 || cfg->num_edges >= (1ll << 40)) return fail(CSL_E_INVALID, "num_edges must be in [0, 2^40)");
  for (int l = 0; l < cfg->n_layers; l++)
    if (cfg->fanout[l] < 1 || cfg->fanout[l] > 255) return fail(CSL_E_INVALID, "fanout[%d] must be 1..255", l);
  if (cfg->indptr[0] != 0 || cfg->indptr[cfg->num_nodes] != cfg->num_edges)
    return fail(CSL_E_INVALID, "indptr[0] must be 0 and indptr[N] must equal num_edges");
  for (int64_t v = 0; v < cfg->num_nodes; v++) {
    const int64_t d = cfg->indptr[v + 1] - cfg->indptr[v];
    if (d < 0 || d > (int64_t)DEG_MASK) return fail(CSL_E_INVALID, "row %lld: degree %lld unsupported", (long long)v, (long long)d);
  }
  for (int64_t k = 0; k < cfg->num_edges; k++)
    if (cfg->indices[k] < 0 || cfg->indices[k] >= cfg->num_nodes)
      return fail(CSL_E_INVALID, "indices[%lld]=%lld outside [0,num_nodes)", (long long)k, (long long)cfg->indices[k]);
  if (cfg->workload)
    for (int64_t v = 0; v < cfg->num_nodes; v++)
      if (cfg->workload[v] < 0 || cfg->workload[v] >= cfg->n_parts)
        return fail(CSL_E_INVALID, "workload[%lld]=%d outside [0,%d)", (long long)v, cfg->workload[v], cfg->n_parts);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return fail(CSL_E_HIP, "no HIP device available (this engine has no CPU path)");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(CSL_E_INVALID, "device %d out of range", cfg->device);
  csl_engine* e = new csl_engine();
  int r = create_impl(cfg, e);
  if (r) {
    char keep[sizeof(g_err)];
    memcpy(keep, g_err, sizeof(keep));
    csl_destroy(e);
    memcpy(g_err, keep, sizeof(keep));
    return r;
  }
  *out = e;
  return 0;
}

int csl_set_nodes(csl_engine* e, const int64_t* host_nodes, int64_t n) {
  if (!e || !host_nodes || n < 0) return fail(CSL_E_INVALID, "bad argument");
  HIPCHECK(hipSetDevice(e->cfg.device));
  for (int k = 0; k < e->nsets; k++) HIPCHECK(hipStreamSynchronize(e->streams[k]));
  if (e->nodes) {
    hipFree(e->nodes);
    e->dev_bytes -= (int64_t)(e->n_nodes * sizeof(long long));
    e->nodes = nullptr;
  }
  DMALLOC(e->nodes, (size_t)n);
  HIPCHECK(hipMemcpy(e->nodes, host_nodes, sizeof(long long) * (size_t)n, hipMemcpyHostToDevice));
  e->n_nodes = n;
  return 0;
}

int csl_submit_round(csl_engine* e, int64_t first_batch, int32_t batch_size, int32_t n_batches, int32_t slot) {
  if (!e) return fail(CSL_E_INVALID, "null engine");
  if (!e->nodes) return fail(CSL_E_STATE, "csl_set_nodes has not been called");
  if (slot < 0 || slot >= e->slots) return fail(CSL_E_INVALID, "slot out of range");
  if (n_batches < 0 || n_batches > e->S) return fail(CSL_E_INVALID, "n_batches must be 0..n_streams");
  if (batch_size < 1 || batch_size > e->cfg.max_batch) return fail(CSL_E_INVALID, "batch_size must be 1..max_batch");
  if (first_batch < 0) return fail(CSL_E_INVALID, "first_batch < 0");
  HIPCHECK(hipSetDevice(e->cfg.device));
  // the pinned descriptor block of this slot is read when the earlier round's copy EXECUTES, not when
  // it was enqueued: a host running ahead must not overwrite it before then
  if (e->desc_inflight[slot]) {
    HIPCHECK(hipEventSynchronize(e->desc_event[slot]));
    e->desc_inflight[slot] = 0;
  }
  BatchDesc* d = e->desc_host + (size_t)slot * e->S;
  for (int s = 0; s < e->S; s++) {
    long long off = (first_batch + s) * (long long)batch_size;
    long long cnt = 0;
    if (s < n_batches && off < e->n_nodes) cnt = e->n_nodes - off < batch_size ? e->n_nodes - off : batch_size;
    d[s].offset = cnt ? off : 0;
    d[s].count = (int)cnt;
    d[s].pad = 0;
  }
  return run_round(e, e->nodes, n_batches, slot);
}

int csl_submit_seeds(csl_engine* e, const int64_t* seeds, const int64_t* offsets, int32_t n_batches, int32_t slot) {
  if (!e || !offsets) return fail(CSL_E_INVALID, "null argument");
  if (slot < 0 || slot >= e->slots) return fail(CSL_E_INVALID, "slot out of range");
  if (n_batches < 0 || n_batches > e->S) return fail(CSL_E_INVALID, "n_batches must be 0..n_streams");
  HIPCHECK(hipSetDevice(e->cfg.device));
  const int64_t total = offsets[n_batches] - offsets[0];
  if (total < 0) return fail(CSL_E_INVALID, "offsets not ascending");
  if (e->desc_inflight[slot]) {
    HIPCHECK(hipEventSynchronize(e->desc_event[slot]));
    e->desc_inflight[slot] = 0;
  }
  BatchDesc* d = e->desc_host + (size_t)slot * e->S;
  for (int s = 0; s < e->S; s++) {
    d[s].offset = 0;
    d[s].count = 0;
    d[s].pad = 0;
    if (s < n_batches) {
      const int64_t c = offsets[s + 1] - offsets[s];
      if (c < 0 || c > e->cfg.max_batch) return fail(CSL_E_INVALID, "minibatch %d has %lld seeds (max_batch %d)", s, (long long)c, e->cfg.max_batch);
      d[s].offset = offsets[s] - offsets[0];
      d[s].count = (int)c;
    }
  }
  // the staging buffer is reused: wait for rounds that may still read it
  for (int k = 0; k < e->nsets; k++) HIPCHECK(hipStreamSynchronize(e->streams[k]));
  if ((size_t)total > e->seedbuf_cap) {
    if (e->seedbuf) {
      hipFree(e->seedbuf);
      e->dev_bytes -= (int64_t)(e->seedbuf_cap * sizeof(long long));
      e->seedbuf = nullptr;
    }
    size_t cap = (size_t)e->S * (size_t)e->cfg.max_batch;
    DMALLOC(e->seedbuf, cap);
    e->seedbuf_cap = cap;
  }
  if (total) HIPCHECK(hipMemcpy(e->seedbuf, seeds + offsets[0], sizeof(long long) * (size_t)total, hipMemcpyHostToDevice));
  return run_round(e, e->seedbuf, n_batches, slot);
}

int csl_sync(csl_engine* e) {
  if (!e) return fail(CSL_E_INVALID, "null engine");
  HIPCHECK(hipSetDevice(e->cfg.device));
  for (int k = 0; k < e->nsets; k++) HIPCHECK(hipStreamSynchronize(e->streams[k]));
  HIPCHECK(hipStreamSynchronize(e->rng_stream));
  e->dirty = false;
  std::fill(e->slot_pending.begin(), e->slot_pending.end(), 0);
  int r = collect_timing(e);
  if (r) return r;
  return refresh_positions(e);
}

static int load_meta(csl_engine* e, int32_t slot) {
  // wait for the round that filled THIS slot only; later rounds keep running
  if (e->slot_pending[slot]) {
    HIPCHECK(hipEventSynchronize(e->slot_event[slot]));
    e->slot_pending[slot] = 0;
  }
  if (!e->meta_valid[slot]) {
    HIPCHECK(hipMemcpy(e->meta_host.data() + (size_t)slot * e->S, e->meta + (size_t)slot * e->S,
                       sizeof(csl_sample_meta) * e->S, hipMemcpyDeviceToHost));
    e->meta_valid[slot] = 1;
  }
  return 0;
}

int csl_get_meta(csl_engine* e, int32_t slot, int32_t stream, csl_sample_meta* out) {
  if (!e || !out) return fail(CSL_E_INVALID, "null argument");
  if (slot < 0 || slot >= e->slots || stream < 0 || stream >= e->S) return fail(CSL_E_INVALID, "slot/stream out of range");
  HIPCHECK(hipSetDevice(e->cfg.device));
  int r = load_meta(e, slot);
  if (r) return r;
  *out = e->meta_host[(size_t)slot * e->S + stream];
  if (out->error) return fail(CSL_E_DEVICE, "device flagged error bits 0x%x in slot %d stream %d", out->error, slot, stream);
  return 0;
}

int64_t csl_copy_list(csl_engine* e, int32_t slot, int32_t stream, int32_t layer, int32_t kind, int32_t part,
                      int64_t* dst, int64_t cap) {
  if (!e) return fail(CSL_E_INVALID, "null engine");
  if (slot < 0 || slot >= e->slots || stream < 0 || stream >= e->S || layer < 0 || layer >= e->L || kind < 0 ||
      kind >= CSL_NUM_LISTS || part < 0 || part >= e->P)
    return fail(CSL_E_INVALID, "index out of range");
  hipSetDevice(e->cfg.device);
  int r = load_meta(e, slot);
  if (r) return r;
  const csl_sample_meta& sm = e->meta_host[(size_t)slot * e->S + stream];
  if (sm.error) return fail(CSL_E_DEVICE, "device flagged error bits 0x%x", sm.error);
  const csl_layer_meta& m = sm.layer[layer];
  const int64_t lo = m.off[kind][part], hi = m.off[kind][part + 1];
  const int64_t n = hi - lo;
  if (n > cap) return fail(CSL_E_INVALID, "destination too small: need %lld", (long long)n);
  if (n > 0) {
    const int* src = e->arena[layer] + ((size_t)slot * e->S + stream) * e->arena_stride[layer] +
                     e->list_base[layer][kind] + lo;
    std::vector<int> tmp((size_t)n);
    hipError_t er = hipMemcpy(tmp.data(), src, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost);
    if (er != hipSuccess) return fail(CSL_E_HIP, "hipMemcpy failed: %s", hipGetErrorString(er));
    for (int64_t i = 0; i < n; i++) dst[i] = (int64_t)tmp[(size_t)i];  // widen to the reference's `long`
  }
  return n;
}

static int fetch_stage(csl_engine* e, int32_t slot, int32_t stream, csl_sample_meta* meta,
                       int64_t seg[CSL_MAX_LAYERS][CSL_NUM_LISTS], size_t* total) {
  if (!e || !meta || !seg) return fail(CSL_E_INVALID, "null argument");
  if (slot < 0 || slot >= e->slots || stream < 0 || stream >= e->S) return fail(CSL_E_INVALID, "slot/stream out of range");
  HIPCHECK(hipSetDevice(e->cfg.device));
  int r = load_meta(e, slot);
  if (r) return r;
  const csl_sample_meta& sm = e->meta_host[(size_t)slot * e->S + stream];
  *meta = sm;
  if (sm.error) return fail(CSL_E_DEVICE, "device flagged error bits 0x%x in slot %d stream %d", sm.error, slot, stream);
  if (!e->fetch_pack) {
    size_t cap = 0;
    for (int l = 0; l < e->L; l++) cap += e->arena_stride[l];
    for (int k = 0; k < 2; k++) HIPCHECK(hipHostMalloc((void**)&e->fetch_stage2[k], cap * sizeof(int), hipHostMallocDefault));
    HIPCHECK(hipMalloc((void**)&e->fetch_pack, cap * sizeof(int)));
    e->fetch_host = (long long*)malloc(cap * sizeof(long long));
    if (!e->fetch_host) return fail(CSL_E_NOMEM, "host allocation failed");
    e->fetch_cap = cap;
    HIPCHECK(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
  }
  e->fetch_flip ^= 1;
  e->fetch_stage = e->fetch_stage2[e->fetch_flip];
  FetchSegs f;
  f.count = 0;
  size_t o = 0;
  uint32_t longest = 0;
  for (int l = 0; l < e->L; l++) {
    const int* base = e->arena[l] + ((size_t)slot * e->S + stream) * e->arena_stride[l];
    for (int k = 0; k < CSL_NUM_LISTS; k++) {
      const size_t n = sm.layer[l].off[k][e->P];
      seg[l][k] = (int64_t)o;
      if (n) {
        f.src[f.count] = base + e->list_base[l][k];
        f.dst[f.count] = (uint32_t)o;
        f.n[f.count] = (uint32_t)n;
        f.count++;
        if (n > longest) longest = (uint32_t)n;
      }
      o += n;
    }
  }
  if (o) {
    unsigned gx = (longest + 256 * 8 - 1) / (256 * 8);
    if (gx < 1) gx = 1;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(k_pack_sample, dim3(gx, (unsigned)f.count), dim3(256), 0, e->copy_stream, f, e->fetch_pack);
    HIPCHECK(hipMemcpyAsync(e->fetch_stage, e->fetch_pack, o * sizeof(int), hipMemcpyDeviceToHost, e->copy_stream));
  }
  HIPCHECK(hipStreamSynchronize(e->copy_stream));
  *total = o;
  return 0;
}

int csl_fetch_sample(csl_engine* e, int32_t slot, int32_t stream, csl_sample_meta* meta, const int64_t** host_ptr,
                     int64_t seg[CSL_MAX_LAYERS][CSL_NUM_LISTS]) {
  if (!host_ptr) return fail(CSL_E_INVALID, "null argument");
  size_t o = 0;
  int r = fetch_stage(e, slot, stream, meta, seg, &o);
  if (r) return r;
  for (size_t i = 0; i < o; i++) e->fetch_host[i] = (long long)e->fetch_stage[i];  // widen to `long`
  *host_ptr = (const int64_t*)e->fetch_host;
  return 0;
}

int csl_fetch_sample32(csl_engine* e, int32_t slot, int32_t stream, csl_sample_meta* meta, const int32_t** host_ptr,
                       int64_t seg[CSL_MAX_LAYERS][CSL_NUM_LISTS]) {
  if (!host_ptr) return fail(CSL_E_INVALID, "null argument");
  size_t o = 0;
  int r = fetch_stage(e, slot, stream, meta, seg, &o);
  if (r) return r;
  *host_ptr = (const int32_t*)e->fetch_stage;
  return 0;
}

int csl_list_device_ptr(csl_engine* e, int32_t slot, int32_t stream, int32_t layer, int32_t kind, const int32_t** out) {
  if (!e || !out) return fail(CSL_E_INVALID, "null argument");
  if (slot < 0 || slot >= e->slots || stream < 0 || stream >= e->S || layer < 0 || layer >= e->L || kind < 0 ||
      kind >= CSL_NUM_LISTS)
    return fail(CSL_E_INVALID, "index out of range");
  *out = (const int32_t*)(e->arena[layer] + ((size_t)slot * e->S + stream) * e->arena_stride[layer] +
                          e->list_base[layer][kind]);
  return 0;
}

int csl_arena_info(csl_engine* e, int32_t layer, const int32_t** base, int64_t* stride,
                   int64_t list_base[CSL_NUM_LISTS]) {
  if (!e || !base || !stride || !list_base) return fail(CSL_E_INVALID, "null argument");
  if (layer < 0 || layer >= e->L) return fail(CSL_E_INVALID, "layer out of range");
  *base = (const int32_t*)e->arena[layer];
  *stride = (int64_t)e->arena_stride[layer];
  for (int k = 0; k < CSL_NUM_LISTS; k++) list_base[k] = (int64_t)e->list_base[layer][k];
  return 0;
}

int csl_frontier_device_ptr(csl_engine* e, int32_t slot, int32_t stream, int32_t layer, const uint32_t** out) {
  if (!e || !out) return fail(CSL_E_INVALID, "null argument");
  if (slot < 0 || slot >= e->slots || stream < 0 || stream >= e->S || layer < 0 || layer > e->L)
    return fail(CSL_E_INVALID, "index out of range");
  *out = e->fr[layer] + ((size_t)slot * e->S + stream) * e->fcap[layer];
  return 0;
}

int64_t csl_copy_frontier(csl_engine* e, int32_t slot, int32_t stream, int32_t layer, int64_t* dst, int64_t cap) {
  if (!e) return fail(CSL_E_INVALID, "null engine");
  if (slot < 0 || slot >= e->slots || stream < 0 || stream >= e->S || layer < 0 || layer > e->L)
    return fail(CSL_E_INVALID, "index out of range");
  hipSetDevice(e->cfg.device);
  int r = load_meta(e, slot);
  if (r) return r;
  const csl_sample_meta& sm = e->meta_host[(size_t)slot * e->S + stream];
  const int64_t n = layer < e->L ? sm.layer[layer].frontier : sm.layer[e->L - 1].next_frontier;
  if (n > cap) return fail(CSL_E_INVALID, "destination too small: need %lld", (long long)n);
  std::vector<uint32_t> tmp((size_t)n);
  if (n) {
    hipError_t er = hipMemcpy(tmp.data(), e->fr[layer] + ((size_t)slot * e->S + stream) * e->fcap[layer],
                              sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost);
    if (er != hipSuccess) return fail(CSL_E_HIP, "hipMemcpy failed: %s", hipGetErrorString(er));
  }
  for (int64_t i = 0; i < n; i++) dst[i] = (int64_t)tmp[(size_t)i];
  return n;
}

int64_t csl_copy_candidates(csl_engine* e, int32_t slot, int32_t stream, int32_t layer, int64_t* flat, int64_t cap_flat,
                            int64_t* counts, int64_t cap_counts) {
  if (!e || !flat || !counts) return fail(CSL_E_INVALID, "null argument");
  if (!e->candk) return fail(CSL_E_STATE, "engine was not created with CSL_FLAG_KEEP_CANDIDATES");
  if (slot < 0 || slot >= e->slots || stream < 0 || stream >= e->S || layer < 0 || layer >= e->L)
    return fail(CSL_E_INVALID, "index out of range");
  hipSetDevice(e->cfg.device);
  int r = load_meta(e, slot);
  if (r) return r;
  const csl_sample_meta& sm = e->meta_host[(size_t)slot * e->S + stream];
  const size_t F = sm.layer[layer].frontier, W = (size_t)e->cfg.fanout[layer] + 1;
  if ((int64_t)F > cap_counts) return fail(CSL_E_INVALID, "counts too small: need %zu", F);
  std::vector<uint32_t> tmp(F * W);
  if (F) {
    const uint32_t* src = e->candk + (((size_t)slot * e->L + layer) * e->S + stream) * e->ccap_max;
    hipError_t er = hipMemcpy(tmp.data(), src, sizeof(uint32_t) * F * W, hipMemcpyDeviceToHost);
    if (er != hipSuccess) return fail(CSL_E_HIP, "hipMemcpy failed: %s", hipGetErrorString(er));
  }
  int64_t o = 0;
  for (size_t i = 0; i < F; i++) {
    int64_t c = 0;
    for (size_t j = 0; j < W; j++) {
      const uint32_t v = tmp[i * W + j];
      if (v == UNSET) continue;  // a row shorter than the fanout: slots past its degree
      if (o >= cap_flat) return fail(CSL_E_INVALID, "flat too small");
      flat[o++] = (int64_t)v;
      c++;
    }
    counts[i] = c;
  }
  return o;
}

int csl_hip_stream(csl_engine* e, int32_t slot, void** out) {
  if (!e || !out) return fail(CSL_E_INVALID, "null argument");
  if (slot < 0 || slot >= e->slots) return fail(CSL_E_INVALID, "slot out of range");
  *out = (void*)e->streams[e->nsets > 1 ? (slot % e->nsets) : 0];
  return 0;
}

int csl_timing_enable(csl_engine* e, int32_t on) {
  if (!e) return fail(CSL_E_INVALID, "null engine");
  int r = csl_sync(e);
  if (r) return r;
  e->timing = on != 0;
  for (int k = 0; k < CSL_NUM_KERNELS; k++) {
    e->t_ms[k] = 0;
    e->t_n[k] = 0;
  }
  return 0;
}

int csl_timing_read(csl_engine* e, double* ms_total, int64_t* launches) {
  if (!e || !ms_total || !launches) return fail(CSL_E_INVALID, "null argument");
  int r = csl_sync(e);
  if (r) return r;
  for (int k = 0; k < CSL_NUM_KERNELS; k++) {
    ms_total[k] = e->t_ms[k];
    launches[k] = e->t_n[k];
  }
  return 0;
}

int csl_rng_peek(csl_engine* e, uint64_t pos, uint32_t* dst, int64_t n) {
  if (!e || !dst || n < 0) return fail(CSL_E_INVALID, "bad argument");
  HIPCHECK(hipSetDevice(e->cfg.device));
  int r = csl_sync(e);
  if (r) return r;
  // generate on demand (test hook): extend the window to cover [pos, pos+n)
  if ((uint64_t)n > e->ring_words / 2) return fail(CSL_E_INVALID, "peek larger than half the ring");
  while (e->gen_hi < pos + (uint64_t)n) {
    uint64_t words = pos + (uint64_t)n - e->gen_hi;
    uint32_t nblocks = (uint32_t)((words + 623) / 624);
    launch_mt_fill(e->rng_stream, e->mt_state, e->ring, e->ring_words, e->gen_hi, nblocks);
    e->gen_hi += (uint64_t)nblocks * 624ull;
    HIPCHECK(hipStreamSynchronize(e->rng_stream));
  }
  if (pos + e->ring_words < e->gen_hi) return fail(CSL_E_INVALID, "position already overwritten in the ring");
  const uint64_t start = pos & (e->ring_words - 1);
  const uint64_t first = (uint64_t)n < e->ring_words - start ? (uint64_t)n : e->ring_words - start;
  if (first) HIPCHECK(hipMemcpy(dst, e->ring + start, sizeof(uint32_t) * first, hipMemcpyDeviceToHost));
  if ((uint64_t)n > first)
    HIPCHECK(hipMemcpy(dst + first, e->ring, sizeof(uint32_t) * ((uint64_t)n - first), hipMemcpyDeviceToHost));
  return 0;
}

int64_t csl_device_bytes(csl_engine* e) { return e ? e->dev_bytes : 0; }

int csl_totals(csl_engine* e, uint64_t* sampled_edges, uint64_t* minibatches) {
  if (!e) return fail(CSL_E_INVALID, "null engine");
  int r = csl_sync(e);
  if (r) return r;
  std::vector<unsigned long long> tmp((size_t)2 * e->S);
  HIPCHECK(hipMemcpy(tmp.data(), e->acc, sizeof(unsigned long long) * tmp.size(), hipMemcpyDeviceToHost));
  uint64_t ed = 0, mb = 0;
  for (int s = 0; s < e->S; s++) {
    ed += tmp[2 * s];
    mb += tmp[2 * s + 1];
  }
  if (sampled_edges) *sampled_edges = ed;
  if (minibatches) *minibatches = mb;
  return 0;
}

}  // extern "C"

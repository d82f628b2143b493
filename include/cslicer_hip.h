/* cslicer_hip.h -- C ABI of the MI355X-native cslicer engine (libcslicer_hip.so).
 *
 * Drop-in boundary for the reference's cslicer hot path.  The reference has no
 * C ABI of its own: its boundary is the pybind11 module `cslicer`
 * (cslicer/pyfrontend.cpp:116-148) over the C++ objects Slicer / Sample /
 * BiPartite / PySample.  Each entry point below names the reference interface
 * it stands in for.  Plain pointers and sizes only; no torch / pybind types.
 *
 * Execution model: the engine runs S independent "streams".  Stream s is one
 * reference worker (one `Slicer`, WorkerPool.cpp:29-33): it owns its dedup
 * tables and its own std::mt19937(seed) position (slicer.h:33).  One *round*
 * slices up to S minibatches at once, batch k of the round on stream k, all in
 * the same kernel launches.  With S = 1 the engine is the single-worker
 * reference; with S > 1 it is the reference with S workers and the
 * deterministic assignment "batch b -> worker b % S" (the stale
 * cslicer/driver.cpp:69-71 deals batches the same way).
 *
 * Results of a round stay in device memory (result slot `slot`), in the object
 * layout of BiPartite (bipartite.h:9-26): per layer and per list kind one
 * array (int32 on the device) holding the lists of parts 0..P-1 back to back, plus an offset
 * table (csl_layer_meta).  csl_copy_list / csl_list_device_ptr hand them out.
 *
 * All functions return 0 on success, a negative CSL_E_* code on failure;
 * csl_last_error() gives the message (per calling thread).  Nothing here falls
 * back to a CPU path.  An engine is driven by one thread at a time (the
 * reference's CSlicer is likewise single-consumer); different engines are
 * independent.
 */
#ifndef CSLICER_HIP_H
#define CSLICER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CSL_MAX_PARTS 8
#define CSL_MAX_LAYERS 4
#define CSL_ABI_VERSION 5

enum {
  CSL_OK = 0,
  CSL_E_INVALID = -1,   /* bad argument / unsupported configuration */
  CSL_E_HIP = -2,       /* a HIP runtime call failed */
  CSL_E_NOMEM = -3,
  CSL_E_STATE = -4,     /* call sequence error */
  CSL_E_DEVICE = -5     /* the device flagged an error in a sample (see csl_sample_meta.error) */
};

/* bits of csl_sample_meta.error (set by the kernels, never silently ignored) */
enum {
  CSL_ERR_RNG_WINDOW = 1,     /* a draw fell outside the generated mt19937 window */
  CSL_ERR_DUP_SEED = 2,       /* repeated seed id in one minibatch, CSL_MODE_GRAPH only (its specification is
                                 defined for distinct seeds); CSL_MODE_STRICT follows the reference:
                                 every occurrence is sampled, the lists are merged as bipartite.cpp:3-17 does */
  CSL_ERR_SEED_RANGE = 4,     /* seed id outside [0, num_nodes) */
  CSL_ERR_FRONTIER_CAP = 8,   /* a frontier outgrew its configured capacity */
  CSL_ERR_BUCKET_FULL = 16    /* an LDS dedup table overflowed: since round 2 only for ids that collide under BOTH the
                                 bucket hash and the pass hash of an oversized bucket (k_bucket resolves a bucket with
                                 more entries than table slots in several passes) */
};

/* list kinds: members of BiPartite (bipartite.h:9-26) */
enum {
  CSL_IN_NODES = 0,         /* global ids, first-occurrence order (bipartite.cpp:4) */
  CSL_OUT_NODES = 1,        /* global ids */
  CSL_OWNED_OUT_NODES = 2,  /* local index into out_nodes */
  CSL_SELF_IDS_IN = 3,      /* local index into in_nodes or -1 */
  CSL_SELF_IDS_OUT = 4,     /* local index into out_nodes or -1 */
  CSL_TO_IDS = 5,           /* to_ids[gpu_id]  (only the own index is ever filled, slicer.cpp:41) */
  CSL_FROM_IDS = 6,         /* from_ids[gpu_id] (slicer.cpp:42) */
  /* CSL_MODE_STRICT: `indptr` is len(out_nodes) ones and `indices` is empty in the
   * reference's exported object (bipartite.h:55-66 never builds the CSR); the host
   * binding synthesises both, the three kinds below stay empty on the device.
   * CSL_MODE_GRAPH fills them (see csl_config.mode). */
  CSL_INDPTR = 7,           /* graph mode: CSR row pointers over out_nodes, len(out_nodes)+1 per part */
  CSL_INDICES = 8,          /* graph mode: local index into in_nodes of every edge's source */
  CSL_OWNED_DEGREE = 9,     /* graph mode: edges of each owned out node over ALL parts (mean divisor) */
  /* graph mode + CSL_FLAG_TRANSPOSE, every layer but the deepest (with CSL_FLAG_TRANSPOSE_ALL: every layer): the slice
   * by SOURCE, for the backward pass of the aggregation as a gather (each source-gradient row written once: no atomics,
   * nothing to pre-zero).
   * For in node u of slice g, entries [t_indptr[u], t_indptr[u+1]) of t_indices hold, ascending: ~r (negative) when u
   * is the self source of owned out node r (self_ids_in/self_ids_out), then the out row r of every edge u -> r of the
   * slice's CSR (a row twice when the edge was sampled twice).  Lists longer than CSL_T_SORTED_MAX entries -- hub nodes
   * that thousands of the minibatch's rows sampled -- hold the same entries in UNSPECIFIED order;
   * csl_layer_meta.t_max_len tells a consumer whether a slice has such a list (a gather over one of them is one
   * wave's serial walk: consumers fall back to their scatter form for that layer). */
  CSL_T_INDPTR = 10,        /* len(in_nodes)+1 per part */
  CSL_T_INDICES = 11,       /* len(indices) + len(self_ids_in) per part */
  CSL_NUM_LISTS = 12
};
enum {
  CSL_T_SORTED_MAX = 128    /* longest list of a slice by source that is sorted */
};

/* csl_config.mode */
enum {
  /* bit-exact reproduction of the object the reference exports (default) */
  CSL_MODE_STRICT = 0,
  /* the object the reference meant to export: same sampling, same frontiers, but
   *  - a real CSR per slice (indptr/indices; sampled self loops dropped),
   *  - every frontier node owned by g is an out node AND an in node of slice g
   *    (its self entry counts as a source), so self_ids_* never hold -1,
   *  - from_ids / to_ids per ordered pair of parts: slice g's from_ids segment is
   *    split by receiver p (csl_layer_meta.pair_off[0][g][p]), slice p's to_ids
   *    segment by sender g (pair_off[1][p][g]); list (g -> p) has the same length
   *    and node order on both sides,
   *  - owned_degree.
   * Specification: oracle/cslicer_oracle.c::orc_sample_graph. */
  CSL_MODE_GRAPH = 1
};

typedef struct {
  uint32_t abi_version;      /* CSL_ABI_VERSION */
  int32_t device;            /* HIP device ordinal */
  /* graph: the reference's Dataset arrays (dataset.h:39-40), host memory */
  int64_t num_nodes;
  int64_t num_edges;         /* < 2^40 (the reference's `int offset`, slicer.cpp:9, is only defined below 2^31) */
  const int64_t* indptr;     /* [num_nodes + 1] */
  const int64_t* indices;    /* [num_edges] */
  const int32_t* workload;   /* [num_nodes] owner part of each node, or NULL => v % n_parts
                                (pyfrontend.cpp:57 builds workload_map[j] = j % 4) */
  int32_t n_parts;           /* 1..CSL_MAX_PARTS; reference: 4 (sample.h:8) */
  int32_t n_layers;          /* 1..CSL_MAX_LAYERS; reference: 3 (slicer.cpp:75) */
  int32_t fanout[CSL_MAX_LAYERS]; /* layer 0 = hop from the seeds; reference: 10 (slicer.cpp:10,15) */
  int32_t max_batch;         /* largest minibatch (seeds) */
  int32_t n_streams;         /* S, concurrent minibatches per round (reference: worker threads) */
  int32_t n_slots;           /* result slots (rounds whose results can be alive at once), >= 1 */
  uint32_t rng_seed;         /* reference: 5489 (default-constructed std::mt19937) */
  uint32_t rng_ring_log2;    /* log2 of the device mt19937 window in 32-bit words (0 => 26) */
  /* optional frontier capacity per layer input (0 => worst case batch*prod(fanout+1)) */
  int64_t frontier_cap[CSL_MAX_LAYERS + 1];
  int32_t mode;              /* CSL_MODE_STRICT (0) or CSL_MODE_GRAPH */
  int32_t flags;             /* CSL_FLAG_* */
  /* one process per part: bit g set = the lists of part g are written; 0 = all parts.  Sampling, dedup and every
   * size/offset in the meta are unaffected (they are global properties of the minibatch); the list CONTENTS of a
   * part that is masked out are undefined.  A rank of the split-parallel trainer asks for its own part only. */
  uint32_t part_mask;
  uint32_t reserved;         /* 0 */
} csl_config;

/* csl_config.flags */
enum {
  /* run every round on one HIP stream (no overlap of consecutive rounds): per-kernel
   * durations are then not stretched by a neighbouring round; used for profiling */
  CSL_FLAG_SERIAL_ROUNDS = 1,
  /* keep every layer's raw neighbour_sample stream (slicer.cpp:6-22: the node itself, then its sampled
   * neighbours, sampled self loops included) per result slot for csl_copy_candidates; test/debug only:
   * costs n_slots x n_layers x n_streams x (largest candidate capacity) x 4 bytes of device memory */
  CSL_FLAG_KEEP_CANDIDATES = 2,
  /* CSL_MODE_GRAPH: also emit the slices by source (CSL_T_INDPTR / CSL_T_INDICES) for every layer but the deepest
   * (whose sources are input features: no gradient flows to them) */
  CSL_FLAG_TRANSPOSE = 4,
  /* with CSL_FLAG_TRANSPOSE: the deepest layer too (a consumer whose aggregation acts on projected features -- GAT --
   * needs the gradient of every layer's sources) */
  CSL_FLAG_TRANSPOSE_ALL = 8
};

typedef struct {
  uint32_t frontier;         /* |in| of this layer (slice_layer's `in`, slicer.cpp:25) */
  uint32_t next_frontier;    /* |out| */
  uint32_t draws;            /* mt19937 outputs consumed by this layer */
  uint32_t sampled_edges;    /* neighbour_sample entries excluding the leading self entry */
  /* list g of kind k occupies [off[k][g], off[k][g+1]) of the kind's array */
  uint32_t off[CSL_NUM_LISTS][CSL_MAX_PARTS + 1];
  /* graph mode: pair_off[0][g][p] = start of (g -> p) inside slice g's from_ids
   * segment, pair_off[1][p][g] = start of (g -> p) inside slice p's to_ids segment */
  uint32_t pair_off[2][CSL_MAX_PARTS][CSL_MAX_PARTS + 1];
  /* strict mode: length of slice g's exported `indptr`: one `1` per out_nodes PUSH (bipartite.h:59-62), which
   * reorder() does not deduplicate.  Equals len(out_nodes) unless the minibatch repeats a seed id (layer 0).
   * graph mode: len(out_nodes) + 1 when the layer is not empty (the CSR row pointers). */
  uint32_t indptr_len[CSL_MAX_PARTS];
  /* CSL_FLAG_TRANSPOSE: length of the longest list of slice g by source (0 without the flag) */
  uint32_t t_max_len[CSL_MAX_PARTS];
} csl_layer_meta;

typedef struct {
  uint32_t error;            /* CSL_ERR_* bits, 0 = good */
  uint32_t n_seeds;
  uint64_t rng_begin;        /* stream's mt19937 position before / after this sample */
  uint64_t rng_end;
  csl_layer_meta layer[CSL_MAX_LAYERS];
} csl_sample_meta;

typedef struct csl_engine csl_engine;

const char* csl_last_error(void);
int csl_abi_version(void);

/* CSlicer::CSlicer + WorkerPool::WorkerPool + Slicer::Slicer (pyfrontend.cpp:41-70,
 * WorkerPool.cpp:4-35, slicer.h:41-70): uploads the CSR, allocates the per-stream
 * dedup tables (DuplicateRemover, util/duplicate.cpp:8-12) and starts the
 * device mt19937 stream. */
int csl_create(const csl_config* cfg, csl_engine** out);
void csl_destroy(csl_engine* e);

/* WorkerPool::training_nodes (WorkerPool.cpp:12-16,40): the epoch's node order,
 * uploaded once; rounds then take consecutive minibatches from it. */
int csl_set_nodes(csl_engine* e, const int64_t* host_nodes, int64_t n);

/* WorkerPool::run's batch loop + Slicer::run (WorkerPool.cpp:41-50,
 * slicer.cpp:120-140) for n_batches <= S minibatches: minibatch k =
 * nodes[(first_batch+k)*batch : ...] goes to stream k.  Asynchronous. */
int csl_submit_round(csl_engine* e, int64_t first_batch, int32_t batch_size, int32_t n_batches,
                     int32_t slot);

/* Slicer::get_sample(vector<long>&) (slicer.cpp:69-81) with explicit seeds:
 * minibatch k = seeds[offsets[k] .. offsets[k+1]) (host memory). Asynchronous. */
int csl_submit_seeds(csl_engine* e, const int64_t* seeds, const int64_t* offsets, int32_t n_batches,
                     int32_t slot);

/* blocks until every submitted round has finished (ConQueue::pop_object,
 * util/conqueue.h:42-60, is the reference's blocking point) */
int csl_sync(csl_engine* e);

/* PySample(Sample*) (pybipartite.cpp:49-66): sizes/offsets of one sample */
int csl_get_meta(csl_engine* e, int32_t slot, int32_t stream, csl_sample_meta* out);

/* PyBipartite(BiPartite*) (pybipartite.cpp:10-43): copy list `kind` of part
 * `part` of `layer` into dst (host memory, int64).  Returns the length, or a
 * negative error; cap = capacity of dst in elements. */
int64_t csl_copy_list(csl_engine* e, int32_t slot, int32_t stream, int32_t layer, int32_t kind,
                      int32_t part, int64_t* dst, int64_t cap);

/* PySample(Sample*) in one call: every list of one sample copied to a pinned host
 * buffer the engine owns (valid until the next csl_fetch_sample on this engine).
 * seg[l][k] = element offset in *host_ptr of kind k of layer l (its parts back to
 * back, split with meta->layer[l].off[k][..]).  One batch of async copies and one
 * wait instead of layers x parts x kinds blocking copies. */
int csl_fetch_sample(csl_engine* e, int32_t slot, int32_t stream, csl_sample_meta* meta,
                     const int64_t** host_ptr, int64_t seg[CSL_MAX_LAYERS][CSL_NUM_LISTS]);

/* Same without the widening pass: *host_ptr is the pinned int32 staging buffer itself.  Two staging
 * buffers alternate, so the pointer stays valid until the SECOND next fetch: a caller may widen sample k
 * into its own `long` vectors on one thread while another thread already fetches sample k+1.  (The sample
 * leaves the GPU as one packed D2H copy: a small kernel gathers its ~30 list segments first.) */
int csl_fetch_sample32(csl_engine* e, int32_t slot, int32_t stream, csl_sample_meta* meta,
                       const int32_t** host_ptr, int64_t seg[CSL_MAX_LAYERS][CSL_NUM_LISTS]);

/* zero-copy surface: device pointer of the kind's array (all parts back to back,
 * offsets in csl_layer_meta.off).  On the device the lists are int32 (ids and local
 * indices are < 2^31, -1 sentinels kept): half the store traffic of the reference's
 * `long`; the host exports above (csl_copy_list, csl_fetch_sample) widen to int64. */
int csl_list_device_ptr(csl_engine* e, int32_t slot, int32_t stream, int32_t layer, int32_t kind,
                        const int32_t** out);
/* the whole result arena of `layer` at once, for consumers that build their views once:
 * *base = int32 array of [n_slots][n_streams][*stride] elements; list kind k of (slot, stream)
 * starts at ((slot * n_streams + stream) * stride + list_base[k]) */
int csl_arena_info(csl_engine* e, int32_t layer, const int32_t** base, int64_t* stride,
                   int64_t list_base[CSL_NUM_LISTS]);

/* device pointer of the frontier entering `layer` (0..n_layers; n_layers = the
 * nodes whose features the model reads); uint32 ids; length in meta */
int csl_frontier_device_ptr(csl_engine* e, int32_t slot, int32_t stream, int32_t layer, const uint32_t** out);
int64_t csl_copy_frontier(csl_engine* e, int32_t slot, int32_t stream, int32_t layer, int64_t* dst,
                          int64_t cap);

/* Slicer::neighbour_sample's output vectors (slicer.cpp:6-22), the pre-dedup edge stream of `layer`: for every
 * frontier node in order its `neighbors` vector (the node itself first), concatenated into flat[], lengths in
 * counts[] (one per frontier node).  Needs CSL_FLAG_KEEP_CANDIDATES.  Returns the number of ids written to
 * flat, or a negative error.  (Debug/parity export: the reference itself never exports the stream.) */
int64_t csl_copy_candidates(csl_engine* e, int32_t slot, int32_t stream, int32_t layer, int64_t* flat,
                            int64_t cap_flat, int64_t* counts, int64_t cap_counts);

/* HIP stream (a hipStream_t) the rounds of result slot `slot` are launched on, for callers
 * that order their own work after a round.  With >= 2 slots, rounds alternate between two
 * streams (and two scratch sets) so that consecutive rounds overlap on the GPU. */
int csl_hip_stream(csl_engine* e, int32_t slot, void** out);

/* time the dominant kernels of the last rounds with HIP events on the
 * engine's own stream: enable, run rounds, read back per-kernel totals */
#define CSL_NUM_KERNELS 15
int csl_timing_enable(csl_engine* e, int32_t on);
int csl_timing_read(csl_engine* e, double* ms_total /*[CSL_NUM_KERNELS]*/,
                    int64_t* launches /*[CSL_NUM_KERNELS]*/);
const char* csl_kernel_name(int32_t k);

/* measurement hook (environment CSL_WAVE_DUP_PROBE=1 makes the dedup kernel count): out3 = { queue entries seen, entries
 * that are not the first of their node id among the 64 entries a wave holds in one register row, ... among all the entries
 * a wave holds of a bucket } since the library was loaded -- what a wave-level (ballot / swizzle) dedup in front of the LDS
 * table could remove (north_star; profiles/r3_wave_dup/).  Synchronises the device. */
int csl_debug_wave_duplicates(uint64_t* out3);

/* device mt19937 stream self-test hook: copies words [pos, pos+n) to dst */
int csl_rng_peek(csl_engine* e, uint64_t pos, uint32_t* dst, int64_t n);

/* totals since csl_create over all streams (waits for submitted rounds): sampled edges
 * (neighbour_sample entries without the leading self entry) and minibatches sliced */
int csl_totals(csl_engine* e, uint64_t* sampled_edges, uint64_t* minibatches);

/* device memory the engine holds, bytes */
int64_t csl_device_bytes(csl_engine* e);

#ifdef __cplusplus
}
#endif
#endif

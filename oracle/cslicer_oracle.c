/* oracle/cslicer_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the reference's cslicer hot path (sequential, same
 * data structures and cost profile: growable id vectors + dense int masks),
 * generalised only in its constants (fanout per layer, number of layers,
 * number of parts, optional workload table).  With fanout={10,10,10},
 * n_parts=4, workload=NULL (v % 4) it reproduces the reference bit for bit;
 * that is pinned by tests/test_oracle_golden.py against tests/golden/ (vectors
 * produced by the unmodified reference, see oracle/make_golden.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (occ-gnn_amd/) never does.
 *
 * Each function cites the reference file:line it restates (paths relative to
 * the reference's cslicer/ directory).
 */
#define _POSIX_C_SOURCE 200809L /* clock_gettime */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define ORC_MAX_PARTS 8
#define ORC_MAX_LAYERS 8

/* ------------------------------------------------------------------ vectors */
typedef struct {
  int64_t* p;
  size_t n, cap;
} vec;

static void vec_push(vec* v, int64_t x) {
  if (v->n == v->cap) {
    v->cap = v->cap ? v->cap * 2 : 16; /* libstdc++ doubling growth */
    v->p = (int64_t*)realloc(v->p, v->cap * sizeof(int64_t));
  }
  v->p[v->n++] = x;
}
static void vec_clear(vec* v) { v->n = 0; } /* capacity kept, like std::vector::clear */
static void vec_free(vec* v) {
  free(v->p);
  v->p = NULL;
  v->n = v->cap = 0;
}
static void vec_assign(vec* d, const vec* s) {
  vec_clear(d);
  for (size_t i = 0; i < s->n; i++) vec_push(d, s->p[i]);
}

/* ------------------------------------------------------------------ mt19937
 * std::mt19937 (ISO C++ [rand.eng.mers]), default seed 5489u (slicer.h:33 --
 * the engine member is default-constructed and never reseeded).  The 10000th
 * output of a default-constructed engine is 4123659995 (checked in tests). */
typedef struct {
  uint32_t mt[624];
  int idx;
} mt19937;

static void mt_seed(mt19937* m, uint32_t seed) {
  m->mt[0] = seed;
  for (int i = 1; i < 624; i++)
    m->mt[i] = 1812433253u * (m->mt[i - 1] ^ (m->mt[i - 1] >> 30)) + (uint32_t)i;
  m->idx = 624;
}
static uint32_t mt_next(mt19937* m) {
  if (m->idx >= 624) {
    for (int i = 0; i < 624; i++) {
      uint32_t y = (m->mt[i] & 0x80000000u) | (m->mt[(i + 1) % 624] & 0x7fffffffu);
      m->mt[i] = m->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    m->idx = 0;
  }
  uint32_t y = m->mt[m->idx++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

/* ---------------------------------------------------------- DuplicateRemover
 * util/duplicate.h:5-21, util/duplicate.cpp:8-45 */
typedef struct {
  int* mask; /* int mask[num_nodes], duplicate.h:10 */
  vec used_nodes;
} dupr;

static void dupr_init(dupr* d, int64_t n) { /* duplicate.cpp:8-12 */
  d->mask = (int*)calloc((size_t)(n > 0 ? n : 1), sizeof(int));
  memset(&d->used_nodes, 0, sizeof(vec));
}
static void dupr_free(dupr* d) {
  free(d->mask);
  vec_free(&d->used_nodes);
}
static void dupr_order_and_remove_duplicates(dupr* d, vec* nodes) { /* duplicate.cpp:14-26 */
  int i = 0;
  for (size_t k = 0; k < nodes->n; k++) {
    int64_t nd1 = nodes->p[k];
    if (d->mask[nd1] == 0) {
      nodes->p[i] = nd1;
      i++;
      d->mask[nd1] = i;
      vec_push(&d->used_nodes, nd1);
    }
  }
  nodes->n = (size_t)i;
}
static void dupr_clear(dupr* d) { /* duplicate.cpp:28-33 */
  for (size_t k = 0; k < d->used_nodes.n; k++) d->mask[d->used_nodes.p[k]] = 0;
  vec_clear(&d->used_nodes);
}
static void dupr_replace(dupr* d, vec* v) { /* duplicate.cpp:35-39 */
  for (size_t i = 0; i < v->n; i++) v->p[i] = d->mask[v->p[i]] - 1;
}

/* ------------------------------------------------------------------ BiPartite
 * bipartite.h:7-84 */
typedef struct {
  vec in_nodes, indptr, out_nodes, owned_out_nodes, indices;
  vec from_ids[ORC_MAX_PARTS], to_ids[ORC_MAX_PARTS];
  vec self_ids_in, self_ids_out;
  int gpu_id;
} bipartite;

static void bp_add_self_edge(bipartite* b, int64_t nd1) { /* bipartite.h:33-39 */
  if (b->self_ids_in.n != 0 && b->self_ids_in.p[b->self_ids_in.n - 1] == nd1) return;
  vec_push(&b->self_ids_in, nd1);
  vec_push(&b->self_ids_out, nd1);
}
static void bp_add_from_node(bipartite* b, int64_t nd1, int gpu_id) { /* bipartite.h:41-46 */
  vec* v = &b->from_ids[gpu_id];
  if (v->n != 0 && v->p[v->n - 1] == nd1) return;
  vec_push(v, nd1);
}
static void bp_add_to_node(bipartite* b, int64_t nd1, int gpu_id) { /* bipartite.h:48-53 */
  vec* v = &b->to_ids[gpu_id];
  if (v->n != 0 && v->p[v->n - 1] == nd1) return;
  vec_push(v, nd1);
}
/* bipartite.h:55-66.  The parameters are `int`: ids are truncated to 32 bits.
 * Lines 64-65 (`indptr[l] = indptr[l] + 1` with l == indptr.size()) write one
 * element past the vector's size; the slot is either overwritten by the next
 * push_back(1) or never becomes visible, so the exported indptr is all ones.
 * The restatement keeps the visible effect and omits the out-of-bounds store. */
static void bp_add_edge(bipartite* b, int nd1, int nd2, int islocal) {
  if (islocal && (b->owned_out_nodes.n == 0 || b->owned_out_nodes.p[b->owned_out_nodes.n - 1] != nd1))
    vec_push(&b->owned_out_nodes, nd1);
  if (b->out_nodes.n == 0 || b->out_nodes.p[b->out_nodes.n - 1] != nd1) {
    vec_push(&b->out_nodes, nd1);
    vec_push(&b->indptr, 1);
  }
  vec_push(&b->in_nodes, nd2);
}
static void bp_refresh(bipartite* b, int n_parts) { /* bipartite.h:68-81 */
  for (int i = 0; i < n_parts; i++) {
    vec_clear(&b->from_ids[i]);
    vec_clear(&b->to_ids[i]);
  }
  vec_clear(&b->indptr);
  vec_clear(&b->indices);
  vec_clear(&b->self_ids_in);
  vec_clear(&b->self_ids_out);
  vec_clear(&b->in_nodes);
  vec_clear(&b->out_nodes);
  vec_clear(&b->owned_out_nodes);
}
static void bp_reorder(bipartite* b, dupr* dr, int n_parts) { /* bipartite.cpp:3-17 */
  dupr_order_and_remove_duplicates(dr, &b->in_nodes);
  dupr_replace(dr, &b->indices);
  dupr_replace(dr, &b->self_ids_in);
  dupr_clear(dr);

  dupr_order_and_remove_duplicates(dr, &b->out_nodes);
  dupr_replace(dr, &b->owned_out_nodes);
  dupr_replace(dr, &b->self_ids_out);
  for (int i = 0; i < n_parts; i++) {
    dupr_replace(dr, &b->from_ids[i]);
    dupr_replace(dr, &b->to_ids[i]);
  }
  dupr_clear(dr);
}
static void bp_free(bipartite* b) {
  vec_free(&b->in_nodes);
  vec_free(&b->indptr);
  vec_free(&b->out_nodes);
  vec_free(&b->owned_out_nodes);
  vec_free(&b->indices);
  vec_free(&b->self_ids_in);
  vec_free(&b->self_ids_out);
  for (int i = 0; i < ORC_MAX_PARTS; i++) {
    vec_free(&b->from_ids[i]);
    vec_free(&b->to_ids[i]);
  }
}

/* ------------------------------------------------------------------ Slicer
 * sample.h:6-33 (Layer/Sample), slicer.h:12-91 */
typedef struct orc {
  const int64_t* indptr;
  const int64_t* indices;
  int64_t num_nodes;
  const int32_t* workload; /* NULL => v % n_parts (pyfrontend.cpp:57) */
  int n_parts, n_layers;
  int fanout[ORC_MAX_LAYERS];
  bipartite layers[ORC_MAX_LAYERS][ORC_MAX_PARTS]; /* Sample.layers[l].bipartite[g] */
  dupr dr, out_dr;                                  /* slicer.h:58-59 */
  vec in, out, neighbors;                           /* slicer.h:26-29 */
  mt19937 rng;                                      /* slicer.h:33 */
  uint64_t draws;                                   /* engine outputs consumed so far */
  /* capture (not in the reference): per-layer traversal record for tests */
  int capture;
  vec frontier[ORC_MAX_LAYERS + 1];
  vec nbr_counts[ORC_MAX_LAYERS];
  vec nbr_flat[ORC_MAX_LAYERS];
  uint64_t layer_draws[ORC_MAX_LAYERS];
  uint64_t sampled_edges; /* non-self entries pushed by neighbour_sample, last sample */
  /* graph ("fixed") mode, see orc_sample_graph below */
  struct gbp {
    vec in_nodes, out_nodes, indptr, indices, owned_out_nodes, self_ids_in, self_ids_out, owned_degree;
    vec from_ids[ORC_MAX_PARTS], to_ids[ORC_MAX_PARTS];
  } glayers[ORC_MAX_LAYERS][ORC_MAX_PARTS];
  int* gmask; /* scratch: in-node rank + 1 per node, one slice at a time */
  int* grow;  /* scratch: out-node rank + 1 per node */
} orc;

static inline int orc_workload(const orc* o, int64_t v) {
  return o->workload ? o->workload[v] : (int)(v % o->n_parts);
}

/* slicer.cpp:6-22 */
static inline void orc_neighbour_sample(orc* o, int64_t nd1, vec* neighbors, int fanout) {
  vec_push(neighbors, nd1);
  int64_t nbs = o->indptr[nd1 + 1] - o->indptr[nd1];
  int offset = (int)o->indptr[nd1]; /* `int offset`, slicer.cpp:9 */
  if (nbs < fanout) {
    for (int i = 0; i < nbs; i++) vec_push(neighbors, o->indices[offset + i]);
  } else {
    for (int i = 0; i < fanout; i++) {
      /* `int rand_nb = engine() % nbs`: result_type is unsigned long on LP64,
       * so the modulo is evaluated as unsigned long % (unsigned long)nbs */
      int rand_nb = (int)((uint64_t)mt_next(&o->rng) % (uint64_t)nbs);
      o->draws++;
      vec_push(neighbors, o->indices[offset + rand_nb]);
    }
  }
}

/* slicer.cpp:25-64 */
static void orc_slice_layer(orc* o, vec* in, vec* out, int l) {
  bipartite* bp = o->layers[l];
  const int fanout = o->fanout[l];
  uint64_t draws0 = o->draws;
  for (size_t k = 0; k < in->n; k++) {
    int64_t nd1 = in->p[k];
    vec_clear(&o->neighbors);
    orc_neighbour_sample(o, nd1, &o->neighbors, fanout);
    o->sampled_edges += o->neighbors.n - 1;
    if (o->capture) {
      vec_push(&o->nbr_counts[l], (int64_t)o->neighbors.n);
      for (size_t j = 0; j < o->neighbors.n; j++) vec_push(&o->nbr_flat[l], o->neighbors.p[j]);
    }
    int to = orc_workload(o, nd1);
    for (size_t j = 0; j < o->neighbors.n; j++) {
      int64_t nd2 = o->neighbors.p[j];
      if (nd1 == nd2) {
        bp_add_self_edge(&bp[to], nd1);
      } else {
        int from = orc_workload(o, nd2);
        if (to == from) {
          bp_add_edge(&bp[to], (int)nd1, (int)nd2, 1);
        } else {
          bp_add_edge(&bp[from], (int)nd1, (int)nd2, 0);
          bp_add_to_node(&bp[to], nd1, to);       /* index is the OWN id, slicer.cpp:41 */
          bp_add_from_node(&bp[from], nd1, from); /* slicer.cpp:42 */
        }
      }
      if (o->out_dr.mask[nd2] == 0) { /* slicer.cpp:45-49 */
        vec_push(out, nd2);
        o->out_dr.mask[nd2] = 1;
        vec_push(&o->out_dr.used_nodes, nd2);
      }
    }
  }
  for (int i = 0; i < o->n_parts; i++) bp_reorder(&bp[i], &o->dr, o->n_parts); /* slicer.cpp:52-54 */
  dupr_clear(&o->out_dr);                                                       /* slicer.cpp:55 */
  o->layer_draws[l] = o->draws - draws0;
}

/* Slicer::clear slicer.cpp:114-118 + Slicer::get_sample slicer.cpp:69-81 (as
 * called from Slicer::run slicer.cpp:130-131) */
int orc_sample(orc* o, const int64_t* seeds, int64_t n) {
  for (int l = 0; l < o->n_layers; l++)
    for (int g = 0; g < o->n_parts; g++) bp_refresh(&o->layers[l][g], o->n_parts);
  o->sampled_edges = 0;
  if (o->capture) {
    for (int l = 0; l < o->n_layers; l++) {
      vec_clear(&o->nbr_counts[l]);
      vec_clear(&o->nbr_flat[l]);
      vec_clear(&o->frontier[l]);
    }
    vec_clear(&o->frontier[o->n_layers]);
  }
  vec_clear(&o->in);
  for (int64_t i = 0; i < n; i++) vec_push(&o->in, seeds[i]);
  for (int l = 0; l < o->n_layers; l++) {
    if (o->capture) vec_assign(&o->frontier[l], &o->in);
    orc_slice_layer(o, &o->in, &o->out, l);
    vec_assign(&o->in, &o->out); /* in.clear(); in = out; out.clear(); */
    vec_clear(&o->out);
  }
  if (o->capture) vec_assign(&o->frontier[o->n_layers], &o->in);
  return 0;
}

/* Slicer ctor slicer.h:41-70 (two masks; the unused target_nodes array and the
 * debug prints are not restated) */
orc* orc_create(const int64_t* indptr, const int64_t* indices, int64_t num_nodes,
                const int32_t* workload, int n_parts, int n_layers, const int32_t* fanouts,
                uint32_t seed, int capture) {
  if (n_parts < 1 || n_parts > ORC_MAX_PARTS || n_layers < 1 || n_layers > ORC_MAX_LAYERS) return NULL;
  orc* o = (orc*)calloc(1, sizeof(orc));
  o->indptr = indptr;
  o->indices = indices;
  o->num_nodes = num_nodes;
  o->workload = workload;
  o->n_parts = n_parts;
  o->n_layers = n_layers;
  for (int l = 0; l < n_layers; l++) o->fanout[l] = fanouts[l];
  for (int l = 0; l < n_layers; l++)
    for (int g = 0; g < n_parts; g++) o->layers[l][g].gpu_id = g; /* sample.h:12 */
  dupr_init(&o->dr, num_nodes);
  dupr_init(&o->out_dr, num_nodes);
  mt_seed(&o->rng, seed);
  o->capture = capture;
  return o;
}

void orc_destroy(orc* o) {
  if (!o) return;
  for (int l = 0; l < ORC_MAX_LAYERS; l++) {
    for (int g = 0; g < ORC_MAX_PARTS; g++) bp_free(&o->layers[l][g]);
    vec_free(&o->nbr_counts[l]);
    vec_free(&o->nbr_flat[l]);
  }
  for (int l = 0; l <= ORC_MAX_LAYERS; l++) vec_free(&o->frontier[l]);
  for (int l = 0; l < ORC_MAX_LAYERS; l++)
    for (int g = 0; g < ORC_MAX_PARTS; g++) {
      struct gbp* b = &o->glayers[l][g];
      vec_free(&b->in_nodes); vec_free(&b->out_nodes); vec_free(&b->indptr); vec_free(&b->indices);
      vec_free(&b->owned_out_nodes); vec_free(&b->self_ids_in); vec_free(&b->self_ids_out);
      vec_free(&b->owned_degree);
      for (int p = 0; p < ORC_MAX_PARTS; p++) { vec_free(&b->from_ids[p]); vec_free(&b->to_ids[p]); }
    }
  free(o->gmask);
  free(o->grow);
  dupr_free(&o->dr);
  dupr_free(&o->out_dr);
  vec_free(&o->in);
  vec_free(&o->out);
  vec_free(&o->neighbors);
  free(o);
}

/* ------------------------------------------------------------------ graph mode
 * NOT in the reference: the object the reference *meant* to export (a real CSR
 * per slice; its BiPartite::add_edge never builds one, bipartite.h:55-66, and
 * from_ids/to_ids are only filled at the own index, slicer.cpp:41-42).  This is
 * the specification of the engine's CSL_MODE_GRAPH, restated sequentially so
 * the HIP path can be checked bit for bit.  Sampling (rng stream, candidate
 * order, next frontier) is exactly the strict path's.  Per layer and slice g
 * (g owns the SOURCE nodes of its edges):
 *   in_nodes[g]   first-occurrence order over { self entry of every frontier
 *                 node owned by g } U { sampled neighbours owned by g }
 *   out_nodes[g]  frontier nodes owned by g or with >= 1 edge from a g-owned
 *                 source, frontier order
 *   indptr/indices[g]  CSR over out_nodes[g]; indices = in-node rank of each
 *                 edge's source, sampling order (sampled self loops dropped)
 *   owned_out_nodes[g] = self_ids_out[g]  rows of out_nodes[g] owned by g
 *   self_ids_in[g]     their ranks in in_nodes[g]
 *   owned_degree[g]    their edge counts over ALL slices (mean divisor)
 *   from_ids[g][p]     rows of out_nodes[g] owned by p != g (partials to send)
 *   to_ids[p][g]       rows of out_nodes[p] they are added to, same order
 */
static void orc_graph_layer(orc* o, const vec* in, int l) {
  const int P = o->n_parts;
  const vec* counts = &o->nbr_counts[l];
  const vec* flat = &o->nbr_flat[l];
  for (int g = 0; g < P; g++) {
    struct gbp* b = &o->glayers[l][g];
    vec_clear(&b->in_nodes); vec_clear(&b->out_nodes); vec_clear(&b->indptr); vec_clear(&b->indices);
    vec_clear(&b->owned_out_nodes); vec_clear(&b->self_ids_in); vec_clear(&b->self_ids_out);
    vec_clear(&b->owned_degree);
    for (int p = 0; p < P; p++) { vec_clear(&b->from_ids[p]); vec_clear(&b->to_ids[p]); }
  }
  for (int g = 0; g < P; g++) {
    struct gbp* b = &o->glayers[l][g];
    /* pass 1: in_nodes ranks */
    size_t k = 0;
    for (size_t i = 0; i < in->n; i++) {
      int64_t nd1 = in->p[i];
      size_t c = (size_t)counts->p[i];
      for (size_t j = 0; j < c; j++) {
        int64_t nd2 = flat->p[k + j];
        int take = (j == 0) ? (orc_workload(o, nd1) == g) : (nd2 != nd1 && orc_workload(o, nd2) == g);
        if (take && o->gmask[nd2] == 0) {
          vec_push(&b->in_nodes, nd2);
          o->gmask[nd2] = (int)b->in_nodes.n;
        }
      }
      k += c;
    }
    /* pass 2: rows */
    k = 0;
    vec_push(&b->indptr, 0);
    for (size_t i = 0; i < in->n; i++) {
      int64_t nd1 = in->p[i];
      size_t c = (size_t)counts->p[i];
      int own = orc_workload(o, nd1) == g;
      size_t e0 = b->indices.n;
      for (size_t j = 1; j < c; j++) {
        int64_t nd2 = flat->p[k + j];
        if (nd2 != nd1 && orc_workload(o, nd2) == g) vec_push(&b->indices, o->gmask[nd2] - 1);
      }
      if (own || b->indices.n > e0) {
        vec_push(&b->out_nodes, nd1);
        vec_push(&b->indptr, (int64_t)b->indices.n);
        if (own) {
          vec_push(&b->owned_out_nodes, (int64_t)b->out_nodes.n - 1);
          vec_push(&b->self_ids_out, (int64_t)b->out_nodes.n - 1);
          vec_push(&b->self_ids_in, o->gmask[nd1] - 1);
          int64_t deg = 0;
          for (size_t j = 1; j < c; j++) deg += flat->p[k + j] != nd1;
          vec_push(&b->owned_degree, deg);
        }
      }
      k += c;
    }
    for (size_t q = 0; q < b->in_nodes.n; q++) o->gmask[b->in_nodes.p[q]] = 0;
  }
  /* pass 3: boundary lists, per ordered pair (sender g, receiver p) */
  for (int p = 0; p < P; p++) {
    struct gbp* bp = &o->glayers[l][p];
    for (size_t r = 0; r < bp->out_nodes.n; r++) o->grow[bp->out_nodes.p[r]] = (int)r + 1;
    for (int g = 0; g < P; g++) {
      if (g == p) continue;
      struct gbp* bg = &o->glayers[l][g];
      for (size_t r = 0; r < bg->out_nodes.n; r++) {
        int64_t nd1 = bg->out_nodes.p[r];
        if (orc_workload(o, nd1) != p) continue;
        vec_push(&bg->from_ids[p], (int64_t)r);
        vec_push(&bp->to_ids[g], (int64_t)o->grow[nd1] - 1);
      }
    }
    for (size_t r = 0; r < bp->out_nodes.n; r++) o->grow[bp->out_nodes.p[r]] = 0;
  }
}

int orc_sample_graph(orc* o, const int64_t* seeds, int64_t n) {
  if (!o->capture) return -1; /* needs the traversal record */
  if (!o->gmask) {
    o->gmask = (int*)calloc((size_t)(o->num_nodes > 0 ? o->num_nodes : 1), sizeof(int));
    o->grow = (int*)calloc((size_t)(o->num_nodes > 0 ? o->num_nodes : 1), sizeof(int));
  }
  orc_sample(o, seeds, n); /* same rng consumption, same frontiers */
  for (int l = 0; l < o->n_layers; l++) orc_graph_layer(o, &o->frontier[l], l);
  return 0;
}

/* ------------------------------------------------------------------ accessors */
enum {
  ORC_IN_NODES = 0,
  ORC_INDPTR = 1,
  ORC_OUT_NODES = 2,
  ORC_OWNED_OUT_NODES = 3,
  ORC_INDICES = 4,
  ORC_SELF_IDS_IN = 5,
  ORC_SELF_IDS_OUT = 6,
  ORC_FROM_IDS = 7, /* sub = index */
  ORC_TO_IDS = 8,   /* sub = index */
  ORC_FRONTIER = 9, /* part ignored; layer 0..n_layers (capture only) */
  ORC_NBR_COUNTS = 10,
  ORC_NBR_FLAT = 11,
  /* graph mode (orc_sample_graph) */
  ORC_G_IN_NODES = 100,
  ORC_G_OUT_NODES = 101,
  ORC_G_INDPTR = 102,
  ORC_G_INDICES = 103,
  ORC_G_OWNED_OUT_NODES = 104,
  ORC_G_SELF_IDS_IN = 105,
  ORC_G_SELF_IDS_OUT = 106,
  ORC_G_OWNED_DEGREE = 107,
  ORC_G_FROM_IDS = 108, /* sub = receiver */
  ORC_G_TO_IDS = 109    /* sub = sender */
};

static const vec* orc_pick(const orc* o, int layer, int part, int which, int sub) {
  static const vec empty = {NULL, 0, 0};
  if (which == ORC_FRONTIER) return (layer >= 0 && layer <= o->n_layers) ? &o->frontier[layer] : &empty;
  if (layer < 0 || layer >= o->n_layers) return &empty;
  if (which == ORC_NBR_COUNTS) return &o->nbr_counts[layer];
  if (which == ORC_NBR_FLAT) return &o->nbr_flat[layer];
  if (part < 0 || part >= o->n_parts) return &empty;
  if (which >= ORC_G_IN_NODES) {
    const struct gbp* gb = &o->glayers[layer][part];
    switch (which) {
      case ORC_G_IN_NODES: return &gb->in_nodes;
      case ORC_G_OUT_NODES: return &gb->out_nodes;
      case ORC_G_INDPTR: return &gb->indptr;
      case ORC_G_INDICES: return &gb->indices;
      case ORC_G_OWNED_OUT_NODES: return &gb->owned_out_nodes;
      case ORC_G_SELF_IDS_IN: return &gb->self_ids_in;
      case ORC_G_SELF_IDS_OUT: return &gb->self_ids_out;
      case ORC_G_OWNED_DEGREE: return &gb->owned_degree;
      case ORC_G_FROM_IDS: return (sub >= 0 && sub < o->n_parts) ? &gb->from_ids[sub] : &empty;
      case ORC_G_TO_IDS: return (sub >= 0 && sub < o->n_parts) ? &gb->to_ids[sub] : &empty;
    }
    return &empty;
  }
  const bipartite* b = &o->layers[layer][part];
  switch (which) {
    case ORC_IN_NODES: return &b->in_nodes;
    case ORC_INDPTR: return &b->indptr;
    case ORC_OUT_NODES: return &b->out_nodes;
    case ORC_OWNED_OUT_NODES: return &b->owned_out_nodes;
    case ORC_INDICES: return &b->indices;
    case ORC_SELF_IDS_IN: return &b->self_ids_in;
    case ORC_SELF_IDS_OUT: return &b->self_ids_out;
    case ORC_FROM_IDS: return (sub >= 0 && sub < o->n_parts) ? &b->from_ids[sub] : &empty;
    case ORC_TO_IDS: return (sub >= 0 && sub < o->n_parts) ? &b->to_ids[sub] : &empty;
  }
  return &empty;
}
int64_t orc_list_len(const orc* o, int layer, int part, int which, int sub) {
  return (int64_t)orc_pick(o, layer, part, which, sub)->n;
}
const int64_t* orc_list_ptr(const orc* o, int layer, int part, int which, int sub) {
  return orc_pick(o, layer, part, which, sub)->p;
}
uint64_t orc_draws_total(const orc* o) { return o->draws; }
uint64_t orc_layer_draws(const orc* o, int layer) { return o->layer_draws[layer]; }
uint64_t orc_sampled_edges(const orc* o) { return o->sampled_edges; }

/* mt19937 self-check hook for tests: n-th output (1-based) for a seed */
uint32_t orc_mt19937_nth(uint32_t seed, uint64_t n) {
  mt19937 m;
  mt_seed(&m, seed);
  uint32_t y = 0;
  for (uint64_t i = 0; i < n; i++) y = mt_next(&m);
  return y;
}
void orc_mt19937_fill(uint32_t seed, uint32_t* out, uint64_t n) {
  mt19937 m;
  mt_seed(&m, seed);
  for (uint64_t i = 0; i < n; i++) out[i] = mt_next(&m);
}

/* ------------------------------------------------------------------ timing
 * cpu_baseline leg ("port"): T slicers, one thread each, every one with its own
 * masks and its own mt19937(seed) as WorkerPool.cpp:29-33 creates them; batch b
 * goes to worker b % T.  With deep_copy != 0 every exported list is copied once
 * per sample as PySample/PyBipartite do (pybipartite.cpp:10-66). */
typedef struct {
  orc* o;
  const int64_t* seeds;
  const int64_t* offs;
  int64_t n_batches;
  int tid, T, deep_copy;
  uint64_t edges;
  uint64_t sink;
} bench_arg;

static uint64_t deep_copy_sample(const orc* o) {
  uint64_t s = 0;
  for (int l = 0; l < o->n_layers; l++)
    for (int g = 0; g < o->n_parts; g++)
      for (int which = 0; which <= ORC_TO_IDS; which++) {
        int subs = (which >= ORC_FROM_IDS) ? o->n_parts : 1;
        for (int sub = 0; sub < subs; sub++) {
          const vec* v = orc_pick(o, l, g, which, sub);
          if (!v->n) continue;
          int64_t* c = (int64_t*)malloc(v->n * sizeof(int64_t));
          memcpy(c, v->p, v->n * sizeof(int64_t));
          s += (uint64_t)c[v->n - 1];
          free(c);
        }
      }
  return s;
}

static void* bench_thread(void* p) {
  bench_arg* a = (bench_arg*)p;
  for (int64_t b = a->tid; b < a->n_batches; b += a->T) {
    orc_sample(a->o, a->seeds + a->offs[b], a->offs[b + 1] - a->offs[b]);
    a->edges += a->o->sampled_edges;
    if (a->deep_copy) a->sink += deep_copy_sample(a->o);
  }
  return NULL;
}

double orc_bench(const int64_t* indptr, const int64_t* indices, int64_t num_nodes,
                 const int32_t* workload, int n_parts, int n_layers, const int32_t* fanouts,
                 uint32_t seed, const int64_t* seeds, const int64_t* batch_offsets,
                 int64_t n_batches, int threads, int deep_copy, uint64_t* sampled_edges_out) {
  if (threads < 1) threads = 1;
  bench_arg* args = (bench_arg*)calloc((size_t)threads, sizeof(bench_arg));
  pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
  for (int t = 0; t < threads; t++) {
    args[t].o = orc_create(indptr, indices, num_nodes, workload, n_parts, n_layers, fanouts, seed, 0);
    args[t].seeds = seeds;
    args[t].offs = batch_offsets;
    args[t].n_batches = n_batches;
    args[t].tid = t;
    args[t].T = threads;
    args[t].deep_copy = deep_copy;
  }
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, bench_thread, &args[t]);
  for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  uint64_t edges = 0;
  for (int t = 0; t < threads; t++) {
    edges += args[t].edges;
    orc_destroy(args[t].o);
  }
  if (sampled_edges_out) *sampled_edges_out = edges;
  free(args);
  free(th);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

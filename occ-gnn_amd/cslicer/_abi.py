"""ctypes binding of libcslicer_hip.so (include/cslicer_hip.h).

The HIP library is the only implementation: if it is missing or cannot be
loaded this module raises -- there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# CSLICER_LIB: alternative build of the same library (tuning sweeps); default is the in-tree build
LIB_PATH = os.environ.get("CSLICER_LIB") or os.path.join(os.path.dirname(HERE), "lib", "libcslicer_hip.so")

MAX_PARTS = 8
MAX_LAYERS = 4
ABI_VERSION = 5
NUM_LISTS = 12
NUM_KERNELS = 15
(IN_NODES, OUT_NODES, OWNED_OUT_NODES, SELF_IDS_IN, SELF_IDS_OUT, TO_IDS, FROM_IDS,
 INDPTR, INDICES, OWNED_DEGREE, T_INDPTR, T_INDICES) = range(12)
MODE_STRICT, MODE_GRAPH = 0, 1
FLAG_SERIAL_ROUNDS = 1
FLAG_KEEP_CANDIDATES = 2
FLAG_TRANSPOSE = 4   # graph mode: also the slices by source (T_INDPTR / T_INDICES), every layer but the deepest
FLAG_TRANSPOSE_ALL = 8   # ... the deepest too (with FLAG_TRANSPOSE)
T_SORTED_MAX = 128       # lists of a slice by source up to this length are sorted; longer ones (hubs) are not
LIST_KINDS = {
    "in_nodes": IN_NODES, "out_nodes": OUT_NODES, "owned_out_nodes": OWNED_OUT_NODES,
    "self_ids_in": SELF_IDS_IN, "self_ids_out": SELF_IDS_OUT, "to_ids": TO_IDS, "from_ids": FROM_IDS,
}
ERR_BITS = {1: "RNG_WINDOW", 2: "DUP_SEED", 4: "SEED_RANGE", 8: "FRONTIER_CAP", 16: "BUCKET_FULL"}

# every symbol include/cslicer_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "csl_last_error", "csl_abi_version", "csl_create", "csl_destroy", "csl_set_nodes",
    "csl_submit_round", "csl_submit_seeds", "csl_sync", "csl_get_meta", "csl_copy_list",
    "csl_list_device_ptr", "csl_frontier_device_ptr", "csl_copy_frontier", "csl_hip_stream",
    "csl_timing_enable", "csl_timing_read", "csl_kernel_name", "csl_rng_peek", "csl_device_bytes",
    "csl_debug_wave_duplicates",
    "csl_fetch_sample", "csl_fetch_sample32", "csl_totals", "csl_arena_info", "csl_copy_candidates",
]


class Config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("device", C.c_int32),
        ("num_nodes", C.c_int64),
        ("num_edges", C.c_int64),
        ("indptr", C.POINTER(C.c_int64)),
        ("indices", C.POINTER(C.c_int64)),
        ("workload", C.POINTER(C.c_int32)),
        ("n_parts", C.c_int32),
        ("n_layers", C.c_int32),
        ("fanout", C.c_int32 * MAX_LAYERS),
        ("max_batch", C.c_int32),
        ("n_streams", C.c_int32),
        ("n_slots", C.c_int32),
        ("rng_seed", C.c_uint32),
        ("rng_ring_log2", C.c_uint32),
        ("frontier_cap", C.c_int64 * (MAX_LAYERS + 1)),
        ("mode", C.c_int32),
        ("flags", C.c_int32),
        ("part_mask", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class LayerMeta(C.Structure):
    _fields_ = [
        ("frontier", C.c_uint32),
        ("next_frontier", C.c_uint32),
        ("draws", C.c_uint32),
        ("sampled_edges", C.c_uint32),
        ("off", (C.c_uint32 * (MAX_PARTS + 1)) * NUM_LISTS),
        ("pair_off", ((C.c_uint32 * (MAX_PARTS + 1)) * MAX_PARTS) * 2),
        ("indptr_len", C.c_uint32 * MAX_PARTS),
        ("t_max_len", C.c_uint32 * MAX_PARTS),
    ]


class SampleMeta(C.Structure):
    _fields_ = [
        ("error", C.c_uint32),
        ("n_seeds", C.c_uint32),
        ("rng_begin", C.c_uint64),
        ("rng_end", C.c_uint64),
        ("layer", LayerMeta * MAX_LAYERS),
    ]


class CslError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("cslicer_hip error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load():
    """Load libcslicer_hip.so; raises if it is absent (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s not found: build it with `make -C occ-gnn_amd/csrc` (hipcc, gfx950). "
            "The cslicer engine has no CPU implementation." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    p64 = C.POINTER(C.c_int64)
    vp = C.c_void_p
    L.csl_last_error.restype = C.c_char_p
    L.csl_abi_version.restype = C.c_int
    L.csl_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.csl_destroy.argtypes = [vp]
    L.csl_destroy.restype = None
    L.csl_set_nodes.argtypes = [vp, p64, C.c_int64]
    L.csl_submit_round.argtypes = [vp, C.c_int64, C.c_int32, C.c_int32, C.c_int32]
    L.csl_submit_seeds.argtypes = [vp, p64, p64, C.c_int32, C.c_int32]
    L.csl_sync.argtypes = [vp]
    L.csl_get_meta.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(SampleMeta)]
    L.csl_copy_list.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, p64, C.c_int64]
    L.csl_copy_list.restype = C.c_int64
    L.csl_list_device_ptr.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp)]
    L.csl_fetch_sample.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(SampleMeta), C.POINTER(p64),
                                   C.POINTER((C.c_int64 * NUM_LISTS) * MAX_LAYERS)]
    L.csl_frontier_device_ptr.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp)]
    L.csl_copy_frontier.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, p64, C.c_int64]
    L.csl_copy_frontier.restype = C.c_int64
    L.csl_hip_stream.argtypes = [vp, C.c_int32, C.POINTER(vp)]
    L.csl_copy_candidates.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, p64, C.c_int64, p64, C.c_int64]
    L.csl_copy_candidates.restype = C.c_int64
    L.csl_timing_enable.argtypes = [vp, C.c_int32]
    L.csl_timing_read.argtypes = [vp, C.POINTER(C.c_double), p64]
    L.csl_kernel_name.argtypes = [C.c_int32]
    L.csl_kernel_name.restype = C.c_char_p
    L.csl_rng_peek.argtypes = [vp, C.c_uint64, C.POINTER(C.c_uint32), C.c_int64]
    L.csl_totals.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.csl_arena_info.argtypes = [vp, C.c_int32, C.POINTER(vp), C.POINTER(C.c_int64), C.POINTER(C.c_int64 * NUM_LISTS)]
    L.csl_device_bytes.argtypes = [vp]
    L.csl_device_bytes.restype = C.c_int64
    if L.csl_abi_version() != ABI_VERSION:
        raise ImportError("libcslicer_hip.so ABI version mismatch")
    _lib = L
    return L


def _check(rc):
    if rc < 0:
        raise CslError(rc, load().csl_last_error().decode())
    return rc


ERR_FRONTIER_CAP = 8
E_DEVICE = -5    # CSL_E_DEVICE: the device flagged an error in a sample (csl_sample_meta.error)


class Engine:
    """Owns one csl_engine. Thin, argument-for-argument wrapper of the C ABI.

    Recovery from CSL_ERR_FRONTIER_CAP (the reference's vectors simply grow, bipartite.h:55-66; here a frontier that
    outgrows a USER-given `frontier_cap` poisons the sample, loudly): with `frontier_cap` given and `recover=True`
    (default) the wrapper keeps every submission since creation; the first sample that reports the overflow makes it
    build a FRESH engine with the worst-case capacities (never a re-exec, nothing shared with the old one) and replay
    the submissions, so that every stream's mt19937 position -- and with it every later sample -- is what an engine
    created with enough room would have produced.  `self.recovered` counts how often that happened (at most once: the
    worst case cannot overflow)."""

    def __init__(self, indptr, indices, n_parts=4, fanouts=(10, 10, 10), max_batch=1024,
                 n_streams=1, n_slots=1, workload=None, device=0, rng_seed=5489,
                 rng_ring_log2=0, frontier_cap=None, mode=MODE_STRICT, flags=0, part_mask=0, recover=True):
        self._h = None
        self._kw = dict(n_parts=n_parts, fanouts=fanouts, max_batch=max_batch, n_streams=n_streams, n_slots=n_slots,
                        workload=workload, device=device, rng_seed=rng_seed, rng_ring_log2=rng_ring_log2, mode=mode,
                        flags=flags, part_mask=part_mask)
        self._log = [] if (frontier_cap is not None and recover) else None   # submissions since creation
        self.recovered = 0
        self._create(indptr, indices, frontier_cap=frontier_cap, **self._kw)

    def _create(self, indptr, indices, n_parts, fanouts, max_batch, n_streams, n_slots, workload, device, rng_seed,
                rng_ring_log2, frontier_cap, mode, flags, part_mask):
        L = load()
        self.indptr = np.ascontiguousarray(indptr, dtype=np.int64)
        self.indices = np.ascontiguousarray(indices, dtype=np.int64)
        self.workload = None if workload is None else np.ascontiguousarray(workload, dtype=np.int32)
        self.n_parts, self.n_layers = int(n_parts), len(fanouts)
        self.fanouts = tuple(int(f) for f in fanouts)
        self.n_streams, self.n_slots, self.max_batch = int(n_streams), int(n_slots), int(max_batch)
        cfg = Config()
        cfg.abi_version = ABI_VERSION
        cfg.device = device
        cfg.num_nodes = self.indptr.shape[0] - 1
        cfg.num_edges = self.indices.shape[0]
        cfg.indptr = self.indptr.ctypes.data_as(C.POINTER(C.c_int64))
        cfg.indices = self.indices.ctypes.data_as(C.POINTER(C.c_int64))
        cfg.workload = (self.workload.ctypes.data_as(C.POINTER(C.c_int32))
                        if self.workload is not None else None)
        cfg.n_parts = self.n_parts
        cfg.n_layers = self.n_layers
        if self.n_layers > MAX_LAYERS:
            raise ValueError("at most %d layers" % MAX_LAYERS)
        for l, f in enumerate(self.fanouts):
            cfg.fanout[l] = f
        cfg.max_batch = self.max_batch
        cfg.n_streams = self.n_streams
        cfg.n_slots = self.n_slots
        cfg.rng_seed = rng_seed
        cfg.rng_ring_log2 = rng_ring_log2
        cfg.mode = mode
        cfg.flags = flags
        cfg.part_mask = part_mask
        self.part_mask = part_mask
        self.mode = mode
        self.flags = flags
        if frontier_cap is not None:
            for l, c in enumerate(frontier_cap):
                cfg.frontier_cap[l] = int(c)
        h = C.c_void_p()
        _check(L.csl_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self.num_nodes = int(cfg.num_nodes)

    def close(self):
        if self._h is not None:
            load().csl_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _recover(self, err):
        """a sample overflowed a user-given frontier capacity: fresh engine, worst-case capacities, same history"""
        if self._log is None:
            raise err
        log, self._log = self._log, None          # (the replay below must not log again; the worst case cannot overflow)
        self.close()
        self._create(self.indptr, self.indices, frontier_cap=None, **self._kw)
        for what, args in log:
            getattr(self, what)(*args)
        self.recovered += 1

    # -- submission
    def set_nodes(self, nodes):
        nodes = np.ascontiguousarray(nodes, dtype=np.int64)
        if self._log is not None:
            self._log.append(("set_nodes", (nodes.copy(),)))   # (callers shuffle their node order in place)
        _check(load().csl_set_nodes(self._h, nodes.ctypes.data_as(C.POINTER(C.c_int64)), nodes.shape[0]))

    def submit_round(self, first_batch, batch_size, n_batches=None, slot=0):
        n_batches = self.n_streams if n_batches is None else n_batches
        if self._log is not None:
            self._log.append(("submit_round", (first_batch, batch_size, n_batches, slot)))
        _check(load().csl_submit_round(self._h, first_batch, batch_size, n_batches, slot))

    def submit_seeds(self, batches, slot=0):
        if self._log is not None:
            self._log.append(("submit_seeds", ([np.array(b, dtype=np.int64) for b in batches], slot)))
        flat = np.ascontiguousarray(
            np.concatenate([np.asarray(b, dtype=np.int64).reshape(-1) for b in batches])
            if len(batches) else np.zeros(0, dtype=np.int64))
        offs = np.zeros(len(batches) + 1, dtype=np.int64)
        np.cumsum([len(b) for b in batches], out=offs[1:])
        if flat.shape[0] == 0:
            flat = np.zeros(1, dtype=np.int64)
        _check(load().csl_submit_seeds(self._h, flat.ctypes.data_as(C.POINTER(C.c_int64)),
                                       offs.ctypes.data_as(C.POINTER(C.c_int64)), len(batches), slot))

    def sync(self):
        _check(load().csl_sync(self._h))

    # -- results
    def meta(self, stream=0, slot=0):
        m = SampleMeta()
        try:
            _check(load().csl_get_meta(self._h, slot, stream, C.byref(m)))
        except CslError as err:
            if err.code != E_DEVICE or not (int(m.error) & ERR_FRONTIER_CAP) or self._log is None:
                raise
            self._recover(err)
            m = SampleMeta()
            _check(load().csl_get_meta(self._h, slot, stream, C.byref(m)))
        return m

    def copy_list(self, layer, kind, part, stream=0, slot=0, meta=None):
        m = meta if meta is not None else self.meta(stream, slot)
        n = int(m.layer[layer].off[kind][part + 1]) - int(m.layer[layer].off[kind][part])
        out = np.empty(max(n, 1), dtype=np.int64)
        got = _check(load().csl_copy_list(self._h, slot, stream, layer, kind, part,
                                          out.ctypes.data_as(C.POINTER(C.c_int64)), out.shape[0]))
        return out[:got]

    def copy_frontier(self, layer, stream=0, slot=0, meta=None):
        m = meta if meta is not None else self.meta(stream, slot)
        n = (m.layer[layer].frontier if layer < self.n_layers
             else m.layer[self.n_layers - 1].next_frontier)
        out = np.empty(max(int(n), 1), dtype=np.int64)
        got = _check(load().csl_copy_frontier(self._h, slot, stream, layer,
                                              out.ctypes.data_as(C.POINTER(C.c_int64)), out.shape[0]))
        return out[:got]

    def copy_candidates(self, layer, stream=0, slot=0, meta=None):
        """(flat, counts): the layer's raw neighbour_sample stream (needs flags=FLAG_KEEP_CANDIDATES)."""
        m = meta if meta is not None else self.meta(stream, slot)
        F = int(m.layer[layer].frontier)
        flat = np.empty(max(F * (self.fanouts[layer] + 1), 1), dtype=np.int64)
        counts = np.empty(max(F, 1), dtype=np.int64)
        n = _check(load().csl_copy_candidates(self._h, slot, stream, layer, flat.ctypes.data_as(C.POINTER(C.c_int64)),
                                              flat.shape[0], counts.ctypes.data_as(C.POINTER(C.c_int64)),
                                              counts.shape[0]))
        return flat[:n], counts[:F]

    def list_device_ptr(self, layer, kind, stream=0, slot=0):
        p = C.c_void_p()
        _check(load().csl_list_device_ptr(self._h, slot, stream, layer, kind, C.byref(p)))
        return p.value

    def frontier_device_ptr(self, layer, stream=0, slot=0):
        p = C.c_void_p()
        _check(load().csl_frontier_device_ptr(self._h, slot, stream, layer, C.byref(p)))
        return p.value

    def hip_stream(self, slot=0):
        p = C.c_void_p()
        _check(load().csl_hip_stream(self._h, slot, C.byref(p)))
        return p.value

    def fetch_sample(self, stream=0, slot=0):
        """All lists of one sample through csl_fetch_sample: {(layer, kind): [parts...]} of numpy copies."""
        m = SampleMeta()
        ptr = C.POINTER(C.c_int64)()
        seg = ((C.c_int64 * NUM_LISTS) * MAX_LAYERS)()
        _check(load().csl_fetch_sample(self._h, slot, stream, C.byref(m), C.byref(ptr), C.byref(seg)))
        out = {}
        for l in range(self.n_layers):
            for k in range(NUM_LISTS):
                tot = int(m.layer[l].off[k][self.n_parts])
                flat = (np.ctypeslib.as_array(ptr, shape=(int(seg[l][k]) + tot,))[int(seg[l][k]):].copy()
                        if tot else np.zeros(0, dtype=np.int64))
                out[(l, k)] = [flat[int(m.layer[l].off[k][g]):int(m.layer[l].off[k][g + 1])]
                               for g in range(self.n_parts)]
        return m, out

    def sample_dict(self, stream=0, slot=0):
        """One sample as a dict of numpy lists (the shape the parity tests compare)."""
        m = self.meta(stream, slot)
        out = {"layers": [], "frontier": [], "draws": [], "sampled_edges": 0}
        for l in range(self.n_layers):
            parts = []
            for g in range(self.n_parts):
                get = lambda k: self.copy_list(l, k, g, stream, slot, m)  # noqa: E731
                bp = {
                    "in_nodes": get(IN_NODES),
                    "out_nodes": get(OUT_NODES),
                    "owned_out_nodes": get(OWNED_OUT_NODES),
                    "self_ids_in": get(SELF_IDS_IN),
                    "self_ids_out": get(SELF_IDS_OUT),
                    "indices": np.zeros(0, dtype=np.int64),
                    "gpu_id": g,
                }
                # bipartite.h:55-66: one `1` per out_nodes push (not deduplicated by reorder), CSR never built
                bp["indptr"] = np.ones(int(m.layer[l].indptr_len[g]), dtype=np.int64)
                empty = np.zeros(0, dtype=np.int64)
                bp["from_ids"] = [get(FROM_IDS) if j == g else empty for j in range(self.n_parts)]
                bp["to_ids"] = [get(TO_IDS) if j == g else empty for j in range(self.n_parts)]
                parts.append(bp)
            out["layers"].append(parts)
            out["draws"].append(int(m.layer[l].draws))
            out["sampled_edges"] += int(m.layer[l].sampled_edges)
        for l in range(self.n_layers + 1):
            out["frontier"].append(self.copy_frontier(l, stream, slot, m))
        out["draws_total"] = int(m.rng_end)
        out["rng_begin"] = int(m.rng_begin)
        if self.flags & FLAG_KEEP_CANDIDATES:
            out["nbr_flat"], out["nbr_counts"] = [], []
            for l in range(self.n_layers):
                flat, counts = self.copy_candidates(l, stream, slot, m)
                out["nbr_flat"].append(flat)
                out["nbr_counts"].append(counts)
        return out

    def graph_dict(self, stream=0, slot=0):
        """One sample of a CSL_MODE_GRAPH engine: real slice CSR + per-peer boundary lists."""
        if self.mode != MODE_GRAPH:
            raise ValueError("engine was not created with mode=MODE_GRAPH")
        m, lists = self.fetch_sample(stream, slot)
        out = {"layers": [], "frontier": [], "sampled_edges": 0}
        P = self.n_parts
        for l in range(self.n_layers):
            parts = []
            lm = m.layer[l]
            for g in range(P):
                bp = {
                    "in_nodes": lists[(l, IN_NODES)][g], "out_nodes": lists[(l, OUT_NODES)][g],
                    "indptr": lists[(l, INDPTR)][g], "indices": lists[(l, INDICES)][g],
                    "owned_out_nodes": lists[(l, OWNED_OUT_NODES)][g],
                    "self_ids_in": lists[(l, SELF_IDS_IN)][g], "self_ids_out": lists[(l, SELF_IDS_OUT)][g],
                    "owned_degree": lists[(l, OWNED_DEGREE)][g], "gpu_id": g,
                    "t_indptr": lists[(l, T_INDPTR)][g], "t_indices": lists[(l, T_INDICES)][g],
                    "t_max_len": int(lm.t_max_len[g]),
                }
                fr, to = lists[(l, FROM_IDS)][g], lists[(l, TO_IDS)][g]
                bp["from_ids"] = [fr[int(lm.pair_off[0][g][p]):int(lm.pair_off[0][g][p + 1])] for p in range(P)]
                bp["to_ids"] = [to[int(lm.pair_off[1][g][p]):int(lm.pair_off[1][g][p + 1])] for p in range(P)]
                parts.append(bp)
            out["layers"].append(parts)
            out["sampled_edges"] += int(lm.sampled_edges)
        for l in range(self.n_layers + 1):
            out["frontier"].append(self.copy_frontier(l, stream, slot, m))
        out["draws_total"] = int(m.rng_end)
        return out

    # -- measurement / test hooks
    def timing_enable(self, on=True):
        _check(load().csl_timing_enable(self._h, 1 if on else 0))

    def timing_read(self):
        ms = (C.c_double * NUM_KERNELS)()
        n = (C.c_int64 * NUM_KERNELS)()
        _check(load().csl_timing_read(self._h, ms, n))
        L = load()
        return {L.csl_kernel_name(k).decode(): (float(ms[k]), int(n[k])) for k in range(NUM_KERNELS)}

    def rng_peek(self, pos, n):
        out = np.empty(n, dtype=np.uint32)
        _check(load().csl_rng_peek(self._h, pos, out.ctypes.data_as(C.POINTER(C.c_uint32)), n))
        return out

    def arena_info(self, layer):
        """(device pointer, elements per (slot, stream), [list base offsets]) of a layer's result arena."""
        p, stride, lb = C.c_void_p(), C.c_int64(0), (C.c_int64 * NUM_LISTS)()
        _check(load().csl_arena_info(self._h, layer, C.byref(p), C.byref(stride), C.byref(lb)))
        return p.value, int(stride.value), [int(x) for x in lb]

    def totals(self):
        """(sampled edges, minibatches) sliced since creation, all streams; waits for submitted rounds."""
        ed, mb = C.c_uint64(0), C.c_uint64(0)
        _check(load().csl_totals(self._h, C.byref(ed), C.byref(mb)))
        return int(ed.value), int(mb.value)

    def device_bytes(self):
        return int(load().csl_device_bytes(self._h))

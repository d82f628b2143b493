"""ctypes binding of oracle/liboracle.so -- CPU ORACLE, TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product path (occ-gnn_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")

LIST_NAMES = ["in_nodes", "indptr", "out_nodes", "owned_out_nodes", "indices",
              "self_ids_in", "self_ids_out"]
IN_NODES, INDPTR, OUT_NODES, OWNED_OUT_NODES, INDICES, SELF_IDS_IN, SELF_IDS_OUT = range(7)
FROM_IDS, TO_IDS, FRONTIER, NBR_COUNTS, NBR_FLAT = 7, 8, 9, 10, 11
GRAPH_LISTS = {"in_nodes": 100, "out_nodes": 101, "indptr": 102, "indices": 103, "owned_out_nodes": 104,
               "self_ids_in": 105, "self_ids_out": 106, "owned_degree": 107}
G_FROM_IDS, G_TO_IDS = 108, 109

_lib = None


def build():
    subprocess.run(["make", "-s", "-f", os.path.join(HERE, "Makefile"), LIB], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(LIB)
        p64 = C.POINTER(C.c_int64)
        p32 = C.POINTER(C.c_int32)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [p64, p64, C.c_int64, p32, C.c_int, C.c_int, p32, C.c_uint32, C.c_int]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_sample.argtypes = [C.c_void_p, p64, C.c_int64]
        L.orc_sample_graph.argtypes = [C.c_void_p, p64, C.c_int64]
        L.orc_list_len.restype = C.c_int64
        L.orc_list_len.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_list_ptr.restype = p64
        L.orc_list_ptr.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_draws_total.restype = C.c_uint64
        L.orc_draws_total.argtypes = [C.c_void_p]
        L.orc_layer_draws.restype = C.c_uint64
        L.orc_layer_draws.argtypes = [C.c_void_p, C.c_int]
        L.orc_sampled_edges.restype = C.c_uint64
        L.orc_sampled_edges.argtypes = [C.c_void_p]
        L.orc_mt19937_nth.restype = C.c_uint32
        L.orc_mt19937_nth.argtypes = [C.c_uint32, C.c_uint64]
        L.orc_mt19937_fill.argtypes = [C.c_uint32, C.POINTER(C.c_uint32), C.c_uint64]
        L.orc_bench.restype = C.c_double
        L.orc_bench.argtypes = [p64, p64, C.c_int64, p32, C.c_int, C.c_int, p32, C.c_uint32,
                                p64, p64, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
        _lib = L
    return _lib


def _p64(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def _p32(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32)) if a is not None else None


def mt19937_stream(n, seed=5489):
    out = np.empty(n, dtype=np.uint32)
    lib().orc_mt19937_fill(seed, out.ctypes.data_as(C.POINTER(C.c_uint32)), n)
    return out


class Oracle:
    """One reference `Slicer` (one worker): own masks, own mt19937(seed)."""

    def __init__(self, indptr, indices, n_parts=4, fanouts=(10, 10, 10), workload=None,
                 seed=5489, capture=True):
        self.indptr = np.ascontiguousarray(indptr, dtype=np.int64)
        self.indices = np.ascontiguousarray(indices, dtype=np.int64)
        self.workload = None if workload is None else np.ascontiguousarray(workload, dtype=np.int32)
        self.fanouts = np.ascontiguousarray(fanouts, dtype=np.int32)
        self.n_parts = int(n_parts)
        self.n_layers = int(self.fanouts.shape[0])
        self.num_nodes = self.indptr.shape[0] - 1
        self._h = lib().orc_create(_p64(self.indptr), _p64(self.indices), self.num_nodes,
                                   _p32(self.workload), self.n_parts, self.n_layers,
                                   _p32(self.fanouts), seed, 1 if capture else 0)
        if not self._h:
            raise ValueError("orc_create failed")

    def close(self):
        if self._h:
            lib().orc_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def _get(self, layer, part, which, sub=0):
        n = lib().orc_list_len(self._h, layer, part, which, sub)
        if n == 0:
            return np.zeros(0, dtype=np.int64)
        p = lib().orc_list_ptr(self._h, layer, part, which, sub)
        return np.ctypeslib.as_array(p, shape=(n,)).copy()

    def sample(self, seeds):
        """Run clear()+get_sample(seeds); return a dict of every exported list."""
        seeds = np.ascontiguousarray(seeds, dtype=np.int64)
        lib().orc_sample(self._h, _p64(seeds), seeds.shape[0])
        out = {"layers": [], "frontier": [], "nbr_counts": [], "nbr_flat": [], "draws": []}
        for l in range(self.n_layers):
            parts = []
            for g in range(self.n_parts):
                bp = {name: self._get(l, g, i) for i, name in enumerate(LIST_NAMES)}
                bp["from_ids"] = [self._get(l, g, FROM_IDS, j) for j in range(self.n_parts)]
                bp["to_ids"] = [self._get(l, g, TO_IDS, j) for j in range(self.n_parts)]
                bp["gpu_id"] = g
                parts.append(bp)
            out["layers"].append(parts)
            out["nbr_counts"].append(self._get(l, 0, NBR_COUNTS))
            out["nbr_flat"].append(self._get(l, 0, NBR_FLAT))
            out["draws"].append(int(lib().orc_layer_draws(self._h, l)))
        for l in range(self.n_layers + 1):
            out["frontier"].append(self._get(l, 0, FRONTIER))
        out["sampled_edges"] = int(lib().orc_sampled_edges(self._h))
        out["draws_total"] = int(lib().orc_draws_total(self._h))
        return out


def _sample_graph(self, seeds):
    """Graph ("fixed") mode: real slice CSR + per-peer boundary lists (the engine's
    CSL_MODE_GRAPH specification, see orc_sample_graph in cslicer_oracle.c)."""
    seeds = np.ascontiguousarray(seeds, dtype=np.int64)
    if lib().orc_sample_graph(self._h, _p64(seeds), seeds.shape[0]) != 0:
        raise RuntimeError("orc_sample_graph needs capture=True")
    out = {"layers": [], "frontier": []}
    for l in range(self.n_layers):
        parts = []
        for g in range(self.n_parts):
            bp = {name: self._get(l, g, code) for name, code in GRAPH_LISTS.items()}
            bp["from_ids"] = [self._get(l, g, G_FROM_IDS, p) for p in range(self.n_parts)]
            bp["to_ids"] = [self._get(l, g, G_TO_IDS, p) for p in range(self.n_parts)]
            bp["gpu_id"] = g
            parts.append(bp)
        out["layers"].append(parts)
    for l in range(self.n_layers + 1):
        out["frontier"].append(self._get(l, 0, FRONTIER))
    out["sampled_edges"] = int(lib().orc_sampled_edges(self._h))
    out["draws_total"] = int(lib().orc_draws_total(self._h))
    return out


Oracle.sample_graph = _sample_graph


def bench(indptr, indices, batches, n_parts=4, fanouts=(10, 10, 10), workload=None, seed=5489,
          threads=1, deep_copy=True):
    """Time the oracle over `batches` (list of int64 arrays). Returns (seconds, sampled_edges)."""
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    indices = np.ascontiguousarray(indices, dtype=np.int64)
    fan = np.ascontiguousarray(fanouts, dtype=np.int32)
    wl = None if workload is None else np.ascontiguousarray(workload, dtype=np.int32)
    flat = np.ascontiguousarray(np.concatenate([np.asarray(b, dtype=np.int64) for b in batches]))
    offs = np.zeros(len(batches) + 1, dtype=np.int64)
    np.cumsum([len(b) for b in batches], out=offs[1:])
    edges = C.c_uint64(0)
    sec = lib().orc_bench(_p64(indptr), _p64(indices), indptr.shape[0] - 1, _p32(wl), n_parts,
                          fan.shape[0], _p32(fan), seed, _p64(flat), _p64(offs), len(batches),
                          threads, 1 if deep_copy else 0, C.byref(edges))
    return sec, int(edges.value)

"""L0 on-disk dataset format of the reference (reader/writer) + synthetic graphs.

Format (reference writer python/utils/convert_dgl_dataset.py:99-127, reference
reader cslicer/dataset.cpp:18-113):

    <dir>/meta.txt            key=value lines: num_nodes, num_edges, feature_dim,
                              csum_features, csum_labels, csum_offsets,
                              csum_edges, num_classes
    <dir>/indptr.bin          int64[N+1]
    <dir>/indices.bin         int64[E]
    <dir>/features.bin        float32[N*feature_dim]   (unused by the slicer)
    <dir>/labels.bin          int32[N]                 (unused by the slicer)
    <dir>/partition_map_opt.bin int32[N]               (loaded, ignored: the
                              reference uses v % 4, cslicer/pyfrontend.cpp:57)

The reference reader skips a final line that has no trailing newline
(dataset.cpp:75), so the writer always terminates every line with '\n'.

OGB datasets are not available offline; `synth_graph` generates graphs of the
same shape: pareto(alpha=1.5) degrees scaled to the requested
mean, uniform random neighbours, rows sorted, no self loops.
"""
import os

import numpy as np

PRESETS = {
    # name: (num_nodes, mean_degree, feature_dim, num_classes)
    "arxiv-like": (169_343, 6.9, 128, 40),
    "products-like": (2_449_029, 50.5, 100, 47),
    "papers-like": (111_059_956, 14.5, 128, 172),
}


def synth_degrees(num_nodes, mean_deg, rng, alpha=1.5, d0=1, cap=None):
    """deg = min(floor(pareto(alpha) * s + d0), cap), s tuned so mean(deg) ~= mean_deg."""
    if cap is None:
        cap = int(min(num_nodes - 1, 20_000))
    x = rng.pareto(alpha, num_nodes)
    lo, hi = 0.0, float(max(mean_deg, 1.0)) * 64.0

    def degs(s):
        return np.minimum(np.floor(x * s + d0), cap).astype(np.int64)

    for _ in range(60):
        mid = 0.5 * (lo + hi)
        if degs(mid).mean() < mean_deg:
            lo = mid
        else:
            hi = mid
    return degs(0.5 * (lo + hi))


def synth_graph(num_nodes, mean_deg, seed=0, alpha=1.5, degrees=None, sort_rows=True):
    """Return (indptr int64[N+1], indices int64[E]) of a synthetic CSR graph."""
    rng = np.random.default_rng(seed)
    if degrees is None:
        deg = synth_degrees(num_nodes, mean_deg, rng, alpha=alpha)
    else:
        deg = np.asarray(degrees, dtype=np.int64)
        assert deg.shape == (num_nodes,)
    indptr = np.zeros(num_nodes + 1, dtype=np.int64)
    np.cumsum(deg, out=indptr[1:])
    num_edges = int(indptr[-1])
    rows = np.repeat(np.arange(num_nodes, dtype=np.int64), deg)
    cols = rng.integers(0, num_nodes, size=num_edges, dtype=np.int64)
    # no self loops (convert_dgl_dataset.py:45 removes them)
    if num_nodes > 1:
        hit = cols == rows
        cols[hit] = (cols[hit] + 1) % num_nodes
    if sort_rows and num_edges:
        # rows sorted (convert_dgl_dataset.py:47); rows are already grouped, so
        # sorting row*N+col keeps the grouping and orders within a row.
        key = rows * np.int64(num_nodes) + cols
        key.sort()
        cols = key - rows * np.int64(num_nodes)
    return indptr, cols


def synth_graph_big(num_nodes, mean_deg, seed=0, alpha=1.5, chunk=4_000_000):
    """synth_graph for graphs with 10^9 edges (papers-like): same degree law and uniform neighbours, rows NOT
    sorted, built node-chunk by node-chunk so that only the result itself (indptr + indices) is ever held;
    the degree scale is tuned on the first two million draws instead of all of them."""
    rng = np.random.default_rng(seed)
    cap = int(min(num_nodes - 1, 20_000))
    x = rng.pareto(alpha, num_nodes)
    sub = x[:2_000_000]
    lo, hi = 0.0, float(max(mean_deg, 1.0)) * 64.0
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        if np.minimum(np.floor(sub * mid + 1), cap).mean() < mean_deg:
            lo = mid
        else:
            hi = mid
    deg = np.minimum(np.floor(x * (0.5 * (lo + hi)) + 1), cap).astype(np.int64)
    del x
    indptr = np.zeros(num_nodes + 1, dtype=np.int64)
    np.cumsum(deg, out=indptr[1:])
    cols = np.empty(int(indptr[-1]), dtype=np.int64)
    for a in range(0, num_nodes, chunk):
        b = min(num_nodes, a + chunk)
        e0, e1 = int(indptr[a]), int(indptr[b])
        c = rng.integers(0, num_nodes, size=e1 - e0, dtype=np.int64)
        rows = np.repeat(np.arange(a, b, dtype=np.int64), deg[a:b])
        hit = c == rows                                    # no self loops (convert_dgl_dataset.py:45)
        c[hit] = (c[hit] + 1) % num_nodes
        cols[e0:e1] = c
    return indptr, cols


def synth_preset(name, seed=0):
    n, d, _, _ = PRESETS[name]
    return synth_graph(n, d, seed=seed)


def write_l0(path, indptr, indices, features=None, labels=None, partition=None,
             feature_dim=None, num_classes=2):
    """Write an L0 dataset directory readable by the reference's Dataset class."""
    os.makedirs(path, exist_ok=True)
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    indices = np.ascontiguousarray(indices, dtype=np.int64)
    n = indptr.shape[0] - 1
    if features is None:
        feature_dim = 1 if feature_dim is None else feature_dim
        features = np.zeros((n, feature_dim), dtype=np.float32)
    features = np.ascontiguousarray(features, dtype=np.float32).reshape(n, -1)
    if labels is None:
        labels = np.zeros(n, dtype=np.int32)
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    if partition is None:
        partition = (np.arange(n) % 4).astype(np.int32)
    partition = np.ascontiguousarray(partition, dtype=np.int32)
    indptr.tofile(os.path.join(path, "indptr.bin"))
    indices.tofile(os.path.join(path, "indices.bin"))
    features.tofile(os.path.join(path, "features.bin"))
    labels.tofile(os.path.join(path, "labels.bin"))
    partition.tofile(os.path.join(path, "partition_map_opt.bin"))
    meta = {
        "num_nodes": n,
        "num_edges": int(indices.shape[0]),
        "feature_dim": int(features.shape[1]),
        "csum_features": int(features.sum(dtype=np.float64)),
        "csum_labels": int(labels.sum(dtype=np.int64)),
        "csum_offsets": int(indptr.sum(dtype=np.int64)),
        "csum_edges": int(indices.sum(dtype=np.int64)),
        "num_classes": int(num_classes),
    }
    with open(os.path.join(path, "meta.txt"), "w") as f:
        for k, v in meta.items():
            f.write("%s=%d\n" % (k, v))
    return meta


def read_meta(path):
    meta = {}
    with open(os.path.join(path, "meta.txt")) as f:
        for line in f:
            if not line.endswith("\n"):
                break  # dataset.cpp:75: an unterminated last line is dropped
            line = line.strip()
            if not line:
                continue
            k, _, v = line.partition("=")
            meta[k] = int(v)
    return meta


def read_l0(path, mmap=True, check=True):
    """Read the graph part of an L0 directory. Returns (indptr, indices, meta)."""
    meta = read_meta(path)
    n, e = meta["num_nodes"], meta["num_edges"]
    if mmap:
        indptr = np.memmap(os.path.join(path, "indptr.bin"), dtype=np.int64, mode="r", shape=(n + 1,))
        indices = np.memmap(os.path.join(path, "indices.bin"), dtype=np.int64, mode="r", shape=(e,))
    else:
        indptr = np.fromfile(os.path.join(path, "indptr.bin"), dtype=np.int64, count=n + 1)
        indices = np.fromfile(os.path.join(path, "indices.bin"), dtype=np.int64, count=e)
    if check:
        # dataset.cpp:27,35 checksum asserts (compiled out in the reference's
        # setup.py build; enforced here, loudly)
        if int(np.sum(indptr, dtype=np.int64)) != meta["csum_offsets"]:
            raise ValueError("indptr checksum mismatch in %s" % path)
        if int(np.sum(indices, dtype=np.int64)) != meta["csum_edges"]:
            raise ValueError("indices checksum mismatch in %s" % path)
    return indptr, indices, meta


def from_edge_list(num_nodes, src, dst, symmetric=False, rows="in"):
    """Edge list -> L0 CSR the way the reference converter prepares it
    (python/utils/convert_dgl_dataset.py:42-49): self loops removed (:45), rows sorted (:47),
    duplicate edges kept.  rows="in": row v lists the sources of edges u -> v (what a GNN samples
    from); rows="out": row u lists the destinations.  The reference takes DGL's `adj()`, whose
    orientation changed between DGL versions; for the symmetric OGB graphs it converts both are the
    same matrix.  symmetric=True adds the reverse of every edge first."""
    src = np.asarray(src, dtype=np.int64).reshape(-1)
    dst = np.asarray(dst, dtype=np.int64).reshape(-1)
    if src.shape != dst.shape:
        raise ValueError("src and dst differ in length")
    if src.size and (min(src.min(), dst.min()) < 0 or max(src.max(), dst.max()) >= num_nodes):
        raise ValueError("edge endpoint outside [0, num_nodes)")
    if symmetric:
        src, dst = np.concatenate([src, dst]), np.concatenate([dst, src])
    keep = src != dst                                   # convert_dgl_dataset.py:45
    src, dst = src[keep], dst[keep]
    if rows == "out":
        src, dst = dst, src
    elif rows != "in":
        raise ValueError("rows must be 'in' or 'out'")
    key = dst * np.int64(num_nodes) + src               # group by destination, sort sources in a row
    key.sort()
    rows, cols = key // np.int64(num_nodes), key % np.int64(num_nodes)
    indptr = np.zeros(num_nodes + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows, minlength=num_nodes), out=indptr[1:])
    return indptr, cols


def _main(argv):
    """python -m cslicer.l0 convert <edges.npz|edges.npy> <out_dir> [--symmetric] [--num-nodes N]
    edges.npz: arrays `src`, `dst` (optionally `features`, `labels`, `partition`, `num_nodes`);
    edges.npy: int array of shape [2, E] or [E, 2]."""
    import argparse
    ap = argparse.ArgumentParser(prog="cslicer.l0")
    sub = ap.add_subparsers(dest="cmd", required=True)
    c = sub.add_parser("convert")
    c.add_argument("edges")
    c.add_argument("out_dir")
    c.add_argument("--symmetric", action="store_true")
    c.add_argument("--num-nodes", type=int, default=None)
    a = ap.parse_args(argv)
    feats = labels = part = None
    if a.edges.endswith(".npz"):
        z = np.load(a.edges)                              # allow_pickle stays False
        src, dst = z["src"], z["dst"]
        feats = z["features"] if "features" in z.files else None
        labels = z["labels"] if "labels" in z.files else None
        part = z["partition"] if "partition" in z.files else None
        n = a.num_nodes or (int(z["num_nodes"]) if "num_nodes" in z.files else None)
    else:
        e = np.load(a.edges)
        e = e if e.shape[0] == 2 else e.T
        src, dst, n = e[0], e[1], a.num_nodes
    if n is None:
        n = int(max(src.max(), dst.max())) + 1
    indptr, indices = from_edge_list(n, src, dst, symmetric=a.symmetric)
    classes = int(labels.max()) + 1 if labels is not None else 2
    meta = write_l0(a.out_dir, indptr, indices, features=feats, labels=labels, partition=part, num_classes=classes)
    print("wrote %s: %s" % (a.out_dir, meta))


if __name__ == "__main__":
    import sys
    _main(sys.argv[1:])

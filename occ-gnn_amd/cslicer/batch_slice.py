"""Forward-only split-parallel pass over an epoch, the counterpart of the reference's
python/batch_slice_multi_gpu.py (DGL NodeDataLoader -> Python 4x4 block splitter ->
DistGraphConv with cross-GPU merge, :128-246), without DGL: the slices come from the HIP engine
(CSL_MODE_GRAPH), the aggregation from the HIP kernels.

All parts run in ONE process on one GPU here (the reference is one process driving 4 GPUs); the
multi-process form is cslicer.train.  Prints the four lines experiments/exp5/populate_table.py:22-25
parses:

    forward_time_per_epoch:<s>     local sum-aggregation over the slice CSRs      (AGGR)
    merge_time per epoch:<s>       adding remote partials into owned rows         (MERGE)
    data transfer:<s>              pulling boundary rows for the peers            (MOVE)
    graph splitting time:<s>       sampling + slicing on the GPU                  (SLICE)
"""
import argparse
import sys
import time

import numpy as np
import torch

from . import _abi, aggr, l0, splitgnn


def run_epoch(eng, feats_dev, n_parts, batch, streams, n_batches, hidden_conv=None):
    """One pass over the engine's node order. Returns seconds: (slice, forward, move, merge)."""
    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
    t_slice = t_fwd = t_move = t_merge = 0.0
    rounds = (n_batches + streams - 1) // streams
    L = eng.n_layers
    for r in range(rounds):
        nb = min(streams, n_batches - r * streams)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.submit_round(r * streams, batch, nb, slot=r & 1)
        eng.sync()
        t_slice += time.perf_counter() - t0
        for s in range(nb):
            sl = splitgnn.slices_of(eng, s, r & 1)
            deep = sl[L - 1]
            x = {g: feats_dev[deep[g].in_nodes.long()] for g in range(n_parts)}
            a, b, c, d = ev(), ev(), ev(), ev()
            a.record()
            agg = {g: aggr.spmm_sum(deep[g].indptr, deep[g].indices, x[g], deep[g].n_out) for g in range(n_parts)}
            b.record()
            send = {g: [aggr.gather_rows(agg[g], deep[g].from_ids[p]) if deep[g].from_ids[p].numel() else None
                        for p in range(n_parts)] for g in range(n_parts)}
            c.record()
            for g in range(n_parts):
                for p in range(n_parts):
                    if p != g and send[p][g] is not None:
                        aggr.scatter_add_rows_(agg[g], deep[g].to_ids[p], send[p][g])
            d.record()
            d.synchronize()
            t_fwd += a.elapsed_time(b) * 1e-3
            t_move += b.elapsed_time(c) * 1e-3
            t_merge += c.elapsed_time(d) * 1e-3
    return t_slice, t_fwd, t_move, t_merge


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--graph", default="synthetic", help="L0 directory, or 'synthetic'")
    ap.add_argument("--nodes", type=int, default=200_000)
    ap.add_argument("--mean-deg", type=float, default=20.0)
    ap.add_argument("--fsize", type=int, default=128)
    ap.add_argument("--batch-size", type=int, default=4096)        # batch_slice_multi_gpu.py defaults
    ap.add_argument("--fan-out", default="10,10,10")
    ap.add_argument("--num-epochs", type=int, default=2)
    ap.add_argument("--n-parts", type=int, default=4)
    ap.add_argument("--streams", type=int, default=8)
    a = ap.parse_args(argv)
    if a.graph == "synthetic":
        indptr, indices = l0.synth_graph(a.nodes, a.mean_deg, seed=0)
    else:
        indptr, indices, _ = l0.read_l0(a.graph, mmap=False)
    n = indptr.shape[0] - 1
    fan = tuple(int(x) for x in a.fan_out.split(","))
    eng = _abi.Engine(indptr, indices, n_parts=a.n_parts, fanouts=fan, max_batch=a.batch_size, n_streams=a.streams,
                      n_slots=2, mode=_abi.MODE_GRAPH)
    feats = torch.rand((n, a.fsize), dtype=torch.float32, device="cuda")
    n_batches = (n + a.batch_size - 1) // a.batch_size
    print("total batches", n / a.batch_size)
    tot = np.zeros(4)
    t1 = time.time()
    for i in range(a.num_epochs):
        print("epoch", i, time.time() - t1)
        eng.set_nodes(np.random.default_rng(i).permutation(n))
        res = run_epoch(eng, feats, a.n_parts, a.batch_size, a.streams, n_batches)
        if i != 0 or a.num_epochs == 1:      # the reference does not count the first epoch (:230)
            tot += np.array(res)
    div = max(a.num_epochs - 1, 1)
    print("Total time :", time.time() - t1)
    print("forward_time_per_epoch:{}".format(tot[1] / div))
    print("merge_time per epoch:{}".format(tot[3] / div))
    print("data transfer:{}".format(tot[2] / div))
    print("graph splitting time:{}".format(tot[0] / div))
    eng.close()


if __name__ == "__main__":
    main(sys.argv[1:])

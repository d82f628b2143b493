#!/usr/bin/env python3
"""How the training step copes with HUB nodes (sources that thousands of a minibatch's rows sample): a graph whose rows
draw half of their neighbours from 1000 popular nodes (Zipf) against the uniform graph of the same size; native step
(input gradients gathered over the slices by source: a hub's list is walked by one wave) vs CSLICER_PY_STEP=1
CSLICER_NO_TRANSPOSE=1 (atomic scatter: a hub's row takes thousands of atomics).
usage (gpurun): python3 profiles/hub_probe.py"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "occ-gnn_amd"))


def graph(n, deg, hubs, seed=0):
    rng = np.random.default_rng(seed)
    nb = rng.integers(0, n, size=(n, deg))
    if hubs:
        pop = rng.permutation(n)[:1000]
        w = 1.0 / np.arange(1, 1001)
        nb[:, :deg // 2] = pop[rng.choice(1000, size=(n, deg // 2), p=w / w.sum())]
    indptr = np.arange(n + 1, dtype=np.int64) * deg
    return indptr, np.sort(nb, axis=1).reshape(-1).astype(np.int64)


def run(hubs):
    import torch
    from cslicer.train import Trainer, synthetic_node_data
    n = 1_000_000
    indptr, indices = graph(n, 40, hubs)
    feats, labels = synthetic_node_data(n, 100, 47)
    t = Trainer(indptr, indices, feats, labels, 47, fanouts=(15, 10, 5), batch=1024, streams=32, hidden=256)
    t.set_nodes(np.random.default_rng(1).permutation(n))
    t.run(64)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t.run(256, first_batch=64)
    dt = time.perf_counter() - t0
    print("%s graph, %s: %.0f minibatches/s" % ("hub" if hubs else "uniform", "native step (by source)" if t.native else
                                                "python step (atomic scatter)", 256 / dt), flush=True)
    t.close()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1] == "hub")
    else:
        for env in ({}, {"CSLICER_PY_STEP": "1", "CSLICER_NO_TRANSPOSE": "1"}):
            for kind in ("uniform", "hub"):
                subprocess.run([sys.executable, os.path.abspath(__file__), kind], env=dict(os.environ, **env))

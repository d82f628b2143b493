#!/usr/bin/env python3
"""Host-side profile (cProfile) of the end-to-end training step: where a step spends its Python/launch time.
usage: python3 profiles/e2e_hostprofile.py [steps] [rank]   (rank: the one-process-per-part path, RCCL world of one)"""
import cProfile
import os
import pstats
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "occ-gnn_amd"))
from cslicer import l0  # noqa: E402
from cslicer.train import Trainer, synthetic_node_data  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cache = os.path.join(os.environ.get("CSLICER_BENCH_CACHE", "/tmp/cslicer_bench_cache"), "g_n2449029_d50.5_s0")
if os.path.exists(os.path.join(cache, "ok")):
    indptr, indices = np.load(os.path.join(cache, "indptr.npy")), np.load(os.path.join(cache, "indices.npy"))
else:
    indptr, indices = l0.synth_graph(2_449_029, 50.5, seed=0)
n = indptr.shape[0] - 1
feats, labels = synthetic_node_data(n, 100, 47)
dist = None
if len(sys.argv) > 2 and sys.argv[2] == "rank":
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29578")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from cslicer.train import use_tuned_gemms  # noqa: E402
use_tuned_gemms()
gat = os.environ.get("HP_MODEL", "sage") == "gat"     # HP_MODEL=gat HP_BATCH=128: a rank's share of config 5 on 8 GPUs
t = Trainer(indptr, indices, feats, labels, 47, fanouts=(10, 10, 10) if gat else (15, 10, 5),
            batch=int(os.environ.get("HP_BATCH", "1024")), streams=32, hidden=32 if gat else 256,
            model="gat" if gat else "sage", rank_path=dist is not None, dist=dist)
t.set_nodes(np.random.default_rng(1).permutation(n))
t.run(64)
import time  # noqa: E402
import torch  # noqa: E402
torch.cuda.synchronize()
t0 = time.perf_counter()
t.run(steps, first_batch=64)
print('un-profiled: %.3f ms/step' % ((time.perf_counter() - t0) / steps * 1e3))
pr = cProfile.Profile()
pr.enable()
t.run(steps, first_batch=64 + steps)
pr.disable()
st = pstats.Stats(pr)
print("profiled: %.3f ms/step (host only; includes cProfile overhead)" % (st.total_tt / steps * 1e3))
st.sort_stats("tottime").print_stats(40)

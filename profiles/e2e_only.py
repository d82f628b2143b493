#!/usr/bin/env python3
"""The end-to-end training leg of bench.py on its own (same shapes: products-like graph, fanout 15/10/5,
batch 1024, features 100, hidden 256, 47 classes, one part on one GPU), for rocprofv3:

    rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 profiles/e2e_only.py --steps 256

Prints one JSON line: ms per step of the timed region."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "occ-gnn_amd"))
from cslicer import _roctx, l0  # noqa: E402
from cslicer.train import Trainer, synthetic_node_data  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=256)
ap.add_argument("--warmup", type=int, default=32)
ap.add_argument("--model", default="sage")
ap.add_argument("--hidden", type=int, default=256)
ap.add_argument("--heads", type=int, default=8)
ap.add_argument("--fanout", default="15,10,5")
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--streams", type=int, default=8, help="minibatches the trainer's engine slices per round")
ap.add_argument("--rank-path", action="store_true",
                help="the one-process-per-part code path (RCCL collectives with a world of one) instead of the fused "
                     "single-part one: what a rank of a multi-GPU job runs per step, minus the peers")
ap.add_argument("--overlap", action="store_true")
ap.add_argument("--tuned", action="store_true", help="use the recorded TunableOp GEMM selections (cslicer.train.use_tuned_gemms)")
a = ap.parse_args()
import torch  # noqa: E402

cache = os.path.join(os.environ.get("CSLICER_BENCH_CACHE", "/tmp/cslicer_bench_cache"), "g_n2449029_d50.5_s0")
if os.path.exists(os.path.join(cache, "ok")):
    indptr, indices = np.load(os.path.join(cache, "indptr.npy")), np.load(os.path.join(cache, "indices.npy"))
else:
    indptr, indices = l0.synth_graph(2_449_029, 50.5, seed=0)
    os.makedirs(cache, exist_ok=True)
    np.save(os.path.join(cache, "indptr.npy"), indptr)
    np.save(os.path.join(cache, "indices.npy"), indices)
    open(os.path.join(cache, "ok"), "w").write("ok\n")
n = indptr.shape[0] - 1
feats, labels = synthetic_node_data(n, 100, 47)
fan = tuple(int(x) for x in a.fanout.split(","))
if a.tuned:
    from cslicer.train import use_tuned_gemms
    print("tuned GEMM selections:", use_tuned_gemms(), file=sys.stderr)
dist = None
if a.rank_path:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = Trainer(indptr, indices, feats, labels, 47, fanouts=fan, batch=a.batch, streams=a.streams, hidden=a.hidden,
            model=a.model, heads=a.heads, rank_path=a.rank_path, dist=dist, overlap=a.overlap)
del feats
t.set_nodes(np.random.default_rng(1).permutation(n))
t.run(a.warmup)
torch.cuda.synchronize()
t0 = time.perf_counter()
_roctx.push("timed_region")      # profiles/e2e_summarize.py windows the kernel trace with this range
t.run(a.steps, first_batch=a.warmup)
torch.cuda.synchronize()
_roctx.pop()
dt = time.perf_counter() - t0
print(json.dumps({"e2e_only": True, "model": a.model, "steps": a.steps, "ms_per_step": 1e3 * dt / a.steps,
                  "iters_per_sec": a.steps / dt, "rank_path": a.rank_path, "overlap": a.overlap, "streams": a.streams,
                  "units_per_step": [{k: round(v / max(t.steps_done, 1), 1) for k, v in u.items()} for u in t.units]}))
t.close()
if dist is not None:
    dist.destroy_process_group()

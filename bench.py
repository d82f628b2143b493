#!/usr/bin/env python3
"""bench.py -- cslicer hot path on MI355X: sampled edges/s and minibatch iters/s.

Workload (BASELINE.json configs[1]): ogbn-products-shaped synthetic graph
(N=2,449,029, mean degree 50.5, pareto degrees), 3-layer
GraphSAGE fanout 15/10/5 (layer 0 = hop from the seeds), minibatch 1024,
4 parts with workload v % 4, S engine streams (= S reference workers, each its
own mt19937(5489)).  One *step* = one round = S minibatches sliced in the same
kernel launches.  Inputs (CSR, node permutation, mt19937 window) are resident
in HBM before the timed region.

Multi-GPU: one process per GPU (torch.distributed, backend nccl == RCCL).  The
slicer shards by minibatch: rank r slices its own minibatches, no data-path
collective ("weak" scaling; `value`).  The split-parallel training step that
consumes the slices (one part per GPU, RCCL all-to-all of boundary partial sums
per layer + gradient all-reduce) runs after it on the same ranks and is reported
as `e2e` / `e2e_iters_per_sec` ("strong" scaling of one global minibatch).

Launch forms: under torch.distributed.run (RANK/WORLD_SIZE in the environment)
this process is one rank.  `python bench.py --gpus N` with N > 1 and no
WORLD_SIZE starts the N ranks itself as a CHILD torch.distributed.run (decided
before anything here imports torch or touches the GPU) and relays rank 0's line.

Prints ONE JSON line on rank 0.  Exit code: 0 only if every leg finished; a
failed or timed-out e2e leg still prints the line (with `e2e.error`) and exits 3.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "occ-gnn_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
FP32_MFMA_PEAK_TFS = 157.3  # MI355X dense fp32 matrix peak (the trainer's GEMMs are fp32, like the reference's)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--streams", type=int, default=128, help="S: minibatches per round")
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--fanout", type=str, default="15,10,5")
    ap.add_argument("--parts", type=int, default=4)
    ap.add_argument("--nodes", type=int, default=2_449_029)
    ap.add_argument("--mean-deg", type=float, default=50.5)
    ap.add_argument("--graph-seed", type=int, default=0)
    ap.add_argument("--unsorted-rows", action="store_true", help="skip the per-row sort of the generator (huge graphs)")
    ap.add_argument("--no-graph-cache", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline leg")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--mode", choices=("strict", "graph"), default="strict",
                    help="strict = the reference's exported lists (headline); graph = real slice CSR, what the trainer consumes")
    ap.add_argument("--slots", type=int, default=3, help="result slots = rounds in flight (engine streams/scratch sets)")
    ap.add_argument("--serial-rounds", action="store_true",
                    help="no overlap of consecutive rounds (profiling: undisturbed per-kernel durations)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (rehearsal on one GPU)")
    ap.add_argument("--e2e-steps", type=int, default=2048,
                    help="end-to-end training minibatches for the secondary iters/s figure (0 = skip)")
    ap.add_argument("--e2e-gat-steps", type=int, default=256,
                    help="one GPU: training minibatches of the GAT leg (BASELINE config 5's shape: 3 layers, 8 heads x 32, "
                         "fanout 10/10/10, batch 1024; reported as e2e_gat; 0 = skip)")
    ap.add_argument("--hub-probe", action="store_true",
                    help="one GPU: also train on a Zipf 'hub' graph and the uniform graph of the same size (profiles/hub_probe.py: "
                         "1 M nodes, degree 40, half of every row's neighbours drawn from 1000 popular nodes) and report both "
                         "rates as e2e_hub_probe: by-source lists of thousands of entries are gathered cooperatively")
    ap.add_argument("--no-e2e-multi", action="store_true", help="several GPUs: skip the split-parallel training leg")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not measure roofline.traffic in this run (two short child runs of this workload under rocprofv3 "
                         "--pmc FETCH_SIZE / WRITE_SIZE); the committed profiles/<LATEST>/pmc.json is then quoted instead")
    ap.add_argument("--e2e-overlap", action="store_true",
                    help="several GPUs: run every boundary exchange of the split-parallel step on a side HIP stream while the "
                         "rows that stay on the GPU are aggregated (same numbers; off by default until it can be measured on "
                         "a multi-GPU node)")
    ap.add_argument("--no-e2e-dp", action="store_true",
                    help="several GPUs: skip the data-parallel training leg (e2e.data_parallel)")
    ap.add_argument("--e2e-timeout", type=float, default=180.0, help="several GPUs: watchdog of the e2e leg, seconds")
    ap.add_argument("--e2e-multi", action="store_true", help="(default now; kept for older command lines)")
    ap.add_argument("--e2e-model", choices=("sage", "gat"), default="sage",
                    help="sage: BASELINE configs 1-3 (default); gat: config 5's layer (8 heads, hidden = per-head width)")
    ap.add_argument("--e2e-heads", type=int, default=8)
    ap.add_argument("--e2e-hidden", type=int, default=256)
    ap.add_argument("--e2e-feat", type=int, default=100)
    ap.add_argument("--e2e-classes", type=int, default=47)
    ap.add_argument("--e2e-streams", type=int, default=64, help="minibatches the trainer's engine slices per round")
    ap.add_argument("--no-tuned-gemms", action="store_true",
                    help="e2e leg: hipBLASLt's default heuristic instead of the recorded TunableOp selections")
    ap.add_argument("--no-compat", action="store_true", help="skip the reference-surface (host lists) leg")
    ap.add_argument("--same-batch", action="store_true",
                    help="experiment (profiles/pmc_same_batch.sh): every stream slices the SAME minibatch, so all S "
                         "streams visit the same rows at the same time -- the ceiling of what grouping the streams' "
                         "frontier nodes by row could save in k_sample's fetches; results are not a throughput figure")
    ap.add_argument("--selftest-dist", choices=("ok", "e2e-fail", "e2e-hang"), default=None,
                    help="no GPU work at all: rendezvous (gloo), reductions, the guarded e2e leg with a stand-in body, "
                         "the single JSON line and the shutdown path -- what tests/test_bench_launch.py runs on CPU")
    return ap.parse_args()


def get_graph(args, rank, barrier):
    """Synthetic L0-shaped graph; rank 0 generates and caches, others load."""
    from cslicer import l0
    if args.no_graph_cache:
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            raise SystemExit("--no-graph-cache is single-rank only")
        return l0.synth_graph(args.nodes, args.mean_deg, seed=args.graph_seed, sort_rows=not args.unsorted_rows)
    key = "g_n%d_d%g_s%d%s" % (args.nodes, args.mean_deg, args.graph_seed, "_u" if args.unsorted_rows else "")
    cache = os.path.join(os.environ.get("CSLICER_BENCH_CACHE", "/tmp/cslicer_bench_cache"), key)
    ok = os.path.join(cache, "ok")
    if rank == 0 and not os.path.exists(ok):
        os.makedirs(cache, exist_ok=True)
        t0 = time.time()
        indptr, indices = l0.synth_graph(args.nodes, args.mean_deg, seed=args.graph_seed,
                                         sort_rows=not args.unsorted_rows)
        np.save(os.path.join(cache, "indptr.npy"), indptr)
        np.save(os.path.join(cache, "indices.npy"), indices)
        open(ok, "w").write("ok\n")
        sys.stderr.write("[bench] generated graph N=%d E=%d in %.1fs\n" % (args.nodes, indices.shape[0], time.time() - t0))
    barrier()
    indptr = np.load(os.path.join(cache, "indptr.npy"))
    indices = np.load(os.path.join(cache, "indices.npy"), mmap_mode="r")
    return indptr, np.ascontiguousarray(indices)


def path_bytes_reference_types(meta_layers, P):
    """SURVEY.md 8(d) algorithmic bytes of one minibatch, reference data types
    (int64 ids/CSR, int32 workload & masks)."""
    B = 0
    for m in meta_layers:
        F, E, D, U = m["F"], m["E"], m["D"], m["U"]
        Ug = m["in_total"]
        O = m["list_total"]
        B += F * (8 + 16 + 4) + E * (8 + 4) + D * 4 + (E + F) * 4 + U * (4 + 8) + E * 4 + Ug * (4 + 8) + 8 * O
        B += 4 * (U + Ug + m["out_total"])
    return B


def kernel_bytes_device_layout(name, m, P, layer=0, last=False, nxt=None):
    """Algorithmic bytes one launch of `name` must move for one minibatch-layer,
    in the engine's own HBM layout (u32 ids, int32 lists; host exports widen to int64).
    Stated per term in DESIGN.md section 5.  nxt = the next layer's units (k_selfin_degree)."""
    F, E, D, U, C = m["F"], m["E"], m["D"], m["U"], m["C"]
    Q = E + F  # bucket queue entries: one per real candidate
    if name == "k_degree":
        # only layer 0 has a k_degree launch of its own: batch ids (int64), row lookup, frontier + ninfo stores
        return F * (8 + 8 + 4 + 8) if layer == 0 else 0
    if name == "k_selfin":
        return F * (4 + 4 + 4 + 4) if last else 0
    if name == "k_selfin_degree":
        # this layer's k_selfin and the next layer's k_degree (which reads the ninfo k_emit prepared) in one launch
        return 0 if last or nxt is None else F * (4 + 4 + 4 + 4) + nxt["F"] * 8
    if name == "k_sample":
        # ids + ninfo, rng words, neighbour gather, candidate + flag store, part mask
        return F * (4 + 8) + D * 4 + E * 4 + C * (4 + 1) + F * 4
    if name == "k_scatter":
        return C * 4 + Q * 8                   # candidates in, {id, position} pairs out
    if name == "k_bucket":
        return Q * 8 + Q * 1 + F * 4           # pairs in (second pass hits L2), flag bytes, first positions
    if name == "k_count":
        return C * 1
    if name == "k_emit":
        # flags, ids of flagged candidates, next frontier, in_nodes + ranks, node lists,
        # and (not on the last layer) the next layer's rowinfo gather + ninfo store
        # (lists are int32 on the device)
        # (the in-node rank `crank` is stored only for candidates whose node is in the frontier: <= F)
        return (C * 1 + (U + m["in_total"]) * 4 + U * 4 + m["in_total"] * 4 + min(F, m["in_total"]) * 4
                + F * (4 + 4) + m["node_lists"] * 4 + F * 4 + (0 if last else U * 16))
    return 0


def compat_leg(indptr, indices, args, B, samples=384, workers=32):
    """The drop-in path with the reference's own surface (PCIe inclusive): the native pybind11 module `cslicer`
    used exactly like the reference's test_py.py -- cslicer(name, queue, workers, epochs, batch) then getSample() in
    a loop, every sample's 3 x 4 BiPartites deep-copied into host `long` vectors -- at the reference's constants
    (fanout 10/10/10, 4 parts, v % 4).  Never `value`: a host-list rate next to the device-resident one."""
    import glob
    import importlib.util
    from cslicer import l0
    hits = glob.glob(os.path.join(ROOT, "occ-gnn_amd", "pybind", "cslicer*.so"))
    if not hits:
        raise ImportError("native cslicer module not built")
    spec = importlib.util.spec_from_file_location("cslicer", hits[0])   # (PyInit_cslicer; not registered in sys.modules)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    root = os.path.join(os.environ.get("CSLICER_BENCH_CACHE", "/tmp/cslicer_bench_cache"), "l0")
    name = "n%d_d%g_s%d" % (args.nodes, args.mean_deg, args.graph_seed)
    if not os.path.exists(os.path.join(root, name, "meta.txt")):
        l0.write_l0(os.path.join(root, name), indptr, indices)
    t0 = time.perf_counter()
    csl = mod.cslicer(name, 16, workers, 1, B, data_root=root)
    t_load = time.perf_counter() - t0
    n = min(samples, csl.getNoSamples())
    s = csl.getSample()          # the first sample includes the mt19937 window fill
    t0 = time.perf_counter()
    for _ in range(n - 1):
        s = csl.getSample()
    dt = time.perf_counter() - t0
    ids = sum(len(s.layers[l][g].in_nodes) for l in range(3) for g in range(4))
    del s, csl
    return {"samples_per_sec": (n - 1) / dt, "samples": n - 1, "workers": workers,
            "config": "native pybind module `cslicer`, fanout 10/10/10, 4 parts (v %% 4), minibatch %d, host `long` "
                      "lists per sample (PCIe + deep copy inclusive)" % B,
            "in_nodes_of_last_sample": ids, "constructor_seconds": t_load}


def pmc_counter_sum(csv_path, counter, kernel):
    """(sum of the counter's values in rocprofv3's unit -- KiB for FETCH_SIZE / WRITE_SIZE --, launches) of `kernel` (name
    without namespace, return type and argument list) in a rocprofv3 counter_collection.csv"""
    import csv
    total, n = 0.0, 0
    for r in csv.DictReader(open(csv_path)):
        nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if nm == kernel and r["Counter_Name"] == counter:
            total += float(r["Counter_Value"])
            n += 1
    return total, n


def live_pmc_traffic(args, kernel, rounds=8, warm=2, timeout=240):
    """HBM-side traffic of `kernel`, bytes per launch, measured NOW: two child runs of this same workload under rocprofv3,
    one per counter (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, each with --kernel-trace only: MI355X_MICROARCH.md's recipe),
    rounds on one HIP stream.  FETCH_SIZE counts 64-B units of read requests that are all 128 B on this path
    (profiles/pmc_rdsize.sh), hence 2 x FETCH_SIZE + WRITE_SIZE.  Returns (bytes per launch, launches) or raises."""
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        raise RuntimeError("rocprofv3 not found")
    child = [sys.executable, os.path.abspath(__file__), "--steps", str(rounds), "--warmup", str(warm), "--no-cpu-baseline",
             "--no-compat", "--no-kernel-timing", "--no-live-pmc", "--serial-rounds", "--e2e-steps", "0", "--gpus", "1",
             "--streams", str(args.streams), "--batch", str(args.batch), "--parts", str(args.parts), "--fanout", args.fanout,
             "--nodes", str(args.nodes), "--mean-deg", str(args.mean_deg), "--graph-seed", str(args.graph_seed),
             "--mode", args.mode]
    if args.unsorted_rows:
        child.append("--unsorted-rows")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["TMPDIR"] = env.get("TMPDIR") or "/tmp"
    tot = {}
    with tempfile.TemporaryDirectory(prefix="bench_pmc_", dir="/tmp") as tmp:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            subprocess.run([exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--"] + child,
                           cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout, check=True)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                raise RuntimeError("rocprofv3 wrote no counter file for " + counter)
            kib, n = pmc_counter_sum(files[0], counter, kernel)
            if n == 0:
                raise RuntimeError("no launch of %s in the %s pass" % (kernel, counter))
            tot[counter] = (kib, n)
    n = tot["FETCH_SIZE"][1]
    return 1024.0 * (2.0 * tot["FETCH_SIZE"][0] / n + tot["WRITE_SIZE"][0] / tot["WRITE_SIZE"][1]), n


def launch_plan(gpus, env):
    """How this invocation runs: ("rank", world) when it is one rank of a job somebody else launched (or the
    single-GPU case), ("spawn", gpus) when it has to start the ranks itself."""
    if "WORLD_SIZE" in env:
        return "rank", int(env["WORLD_SIZE"])
    if gpus > 1:
        return "spawn", gpus
    return "rank", 1


def self_launch(gpus, argv):
    """`python bench.py --gpus N` (N > 1) outside torch.distributed.run: start the N ranks as a child
    process tree.  The parent has not imported torch and never touches the GPU; rank 0's JSON line goes
    straight to the inherited stdout.  Returns the child's exit code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    sys.stderr.write("[bench] --gpus %d without WORLD_SIZE: launching %s\n" % (gpus, " ".join(cmd[1:9])))
    return subprocess.call(cmd)


E2E_FAILED_RC = 3


def run_guarded(leg, out, rank, world, timeout, store=None, partial=None):
    """Run the multi-rank e2e leg under a watchdog.  On success rank 0 gets the result in out["e2e"].  If the leg
    raises on some rank, or does not finish within `timeout` seconds (a collective some other rank never joined),
    rank 0 prints the line with `e2e.error` and every process exits with E2E_FAILED_RC -- never 0: the process
    group may be wedged, so there is no orderly shutdown on that path.

    torch.distributed.run kills the surviving ranks as soon as one exits non-zero, so a failing rank other than 0
    first tells rank 0 through the rendezvous store (`store`, a c10d Store) and waits for its acknowledgement
    (the line is out) before it leaves; the watchdogs poll the store, so nobody sits out the whole timeout."""
    import threading
    finished = threading.Event()
    k_fail, k_ack = "bench_e2e_failed", "bench_e2e_line_printed"

    def store_do(fn, default=None):
        try:
            return fn() if store is not None else default
        except Exception:
            return default

    def give_up(msg):
        if rank == 0:
            # (`partial`: what the leg had finished before it failed -- the split-parallel figures survive a failure
            # of the data-parallel leg that follows them)
            out["e2e"] = dict(partial or {}, error=msg)
            out["e2e_iters_per_sec"] = (partial or {}).get("iters_per_sec")
            sys.stdout.write(json.dumps(out) + "\n")
            sys.stdout.flush()
            store_do(lambda: store.set(k_ack, "1"))
            time.sleep(0.5)      # let the others read the acknowledgement before the launcher reaps everybody
        else:
            store_do(lambda: store.set(k_fail, "rank %d: %s" % (rank, msg)))
            t_end = time.time() + 30.0
            while store is not None and time.time() < t_end and not store_do(lambda: store.check([k_ack]), True):
                time.sleep(0.1)
        sys.stderr.write("[bench] rank %d: e2e leg failed: %s\n" % (rank, msg))
        sys.stderr.flush()
        os._exit(E2E_FAILED_RC)

    def watchdog():
        t_end = time.time() + timeout
        while not finished.wait(0.25):
            if store_do(lambda: store.check([k_fail]), False):
                give_up(store_do(lambda: store.get(k_fail).decode(), "a peer rank failed"))
            if time.time() > t_end:
                give_up("the %d-GPU e2e leg did not finish within %.0f s" % (world, timeout))

    threading.Thread(target=watchdog, daemon=True).start()
    try:
        res = leg()
    except Exception as ex:  # (the collective of a rank whose peer left may raise too: same exit)
        give_up(repr(ex)[:300])
    finished.set()
    if rank == 0:
        out["e2e"] = res


def default_store(dist):
    try:
        return dist.distributed_c10d._get_default_store()
    except Exception:
        return None


def finish(out, rank, dist):
    """Rank 0 prints the single JSON line; then ONE barrier and the process group goes."""
    if rank == 0 and "iters_per_sec" in (out.get("e2e") or {}):
        # first-class: the end-to-end half of the headline metric (strong scaling on several GPUs)
        out["e2e_iters_per_sec"] = out["e2e"]["iters_per_sec"]
        out["e2e_scaling"] = "strong" if out.get("n_gpus", 1) > 1 else None
        dp = out["e2e"].get("data_parallel") or {}
        if "iters_per_sec" in dp:     # several GPUs: the same step data-parallel (cslicer.train.DataParallelTrainer)
            out["e2e_data_parallel_iters_per_sec"] = dp["iters_per_sec"]
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def selftest_dist(args, rank, world):
    """--selftest-dist: the distributed plumbing of this script with no GPU in sight (gloo)."""
    import torch.distributed as dist
    from cslicer import shard
    dist.init_process_group(backend="gloo")
    dt = shard.max_over_ranks(0.001 * (rank + 1), dist, "cpu")
    units = shard.sum_over_ranks(10.0, dist, "cpu")
    out = {"metric": "selftest", "n_gpus": world, "value": units / dt, "selftest": args.selftest_dist,
           "dist": {"backend": dist.get_backend(), "world_size": dist.get_world_size()}}

    def leg():
        if args.selftest_dist == "e2e-fail" and rank == world - 1:
            raise RuntimeError("stand-in failure on rank %d" % rank)
        if args.selftest_dist == "e2e-hang" and rank == world - 1:
            time.sleep(3600)
        dist.barrier()       # the others wait in a collective, as they would in the real leg
        return {"iters_per_sec": 1.0}

    if world > 1:
        run_guarded(leg, out, rank, world, args.e2e_timeout, default_store(dist))
    else:
        out["e2e"] = leg()
    finish(out, rank, dist)



def host_cpu_info():
    """CPU model, physical/logical core counts and the CPUs this process may actually use (affinity mask and
    cgroup quota): what the cpu_baseline leg runs on (BASELINE.md section 2 asks for model and core count)."""
    model, phys = "unknown", set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                pid = v
            elif k == "core id":
                cid = v
            elif not k and pid is not None:
                phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    logical = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = logical
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        pass
    if quota:
        usable = max(1, min(usable, int(quota + 0.5)))
    return {"model": model, "logical_cpus": logical, "physical_cores": len(phys) or None,
            "usable_cpus": usable, "cgroup_cpu_quota": quota}


def main():
    args = parse()
    how, world = launch_plan(args.gpus, os.environ)
    if how == "spawn":
        sys.exit(self_launch(world, sys.argv[1:]))
    fan = tuple(int(x) for x in args.fanout.split(","))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.selftest_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        return selftest_dist(args, rank, world)
    import torch
    dist = None
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs an MI355X: the cslicer engine has no CPU path")
    device = local_rank % ndev
    red_dev = "cuda" if args.dist_backend == "nccl" else "cpu"
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(device)
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend=args.dist_backend)
    if args.gpus != world and rank == 0 and world > 1:
        sys.stderr.write("[bench] --gpus %d but WORLD_SIZE %d; using WORLD_SIZE\n" % (args.gpus, world))

    def barrier():
        if dist is not None:
            dist.barrier()

    from cslicer import _abi
    _abi.load()  # raises if the HIP library is missing: no fallback
    indptr, indices = get_graph(args, rank, barrier)
    N = indptr.shape[0] - 1
    S, B, P = args.streams, args.batch, args.parts
    perm = np.random.default_rng(1).permutation(N).astype(np.int64)
    eng_flags = _abi.FLAG_SERIAL_ROUNDS if args.serial_rounds else 0
    NS = args.slots
    eng_mode = _abi.MODE_GRAPH if args.mode == "graph" else _abi.MODE_STRICT
    eng = _abi.Engine(indptr, indices, n_parts=P, fanouts=fan, max_batch=B, n_streams=S, n_slots=NS,
                      device=device, flags=eng_flags, mode=eng_mode)
    eng.set_nodes(perm)
    from cslicer import shard
    n_rounds, n_batches = shard.rounds_per_epoch(N, B, S)  # full rounds only: every step does S minibatches

    def run_round(step):
        # weak scaling: rank r takes its own rounds (cslicer/shard.py); wraps around the epoch
        first, nb = shard.batches_of_round(shard.round_of(step, rank, world, n_rounds), S)
        if args.same_batch:
            k = step % max(1, n_batches)
            eng.submit_seeds([perm[k * B:(k + 1) * B]] * S, slot=step % NS)
            return
        eng.submit_round(first, B, nb, slot=step % NS)

    def timed(nsteps, first_step):
        barrier()
        eng.sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(nsteps):
            run_round(first_step + k)
        eng.sync()
        torch.cuda.synchronize()
        barrier()
        return time.perf_counter() - t0

    for w in range(args.warmup):
        run_round(w)
    eng.sync()
    edges0, iters0 = eng.totals()        # device-side running totals: exact units of the timed region
    dt = shard.max_over_ranks(timed(args.steps, args.warmup), dist, red_dev)
    edges1, iters1 = eng.totals()

    # ---- units processed: read the last two rounds' metas (both slots)
    def slot_stats(slot):
        layers = [dict(F=0, E=0, D=0, U=0, C=0, in_total=0, out_total=0, node_lists=0, list_total=0) for _ in fan]
        for s in range(S):
            m = eng.meta(s, slot)
            for l, f in enumerate(fan):
                lm = m.layer[l]
                d = layers[l]
                d["F"] += lm.frontier
                d["E"] += lm.sampled_edges
                d["D"] += lm.draws
                d["U"] += lm.next_frontier
                d["C"] += lm.frontier * (f + 1)
                tot = [int(lm.off[k][P]) for k in range(_abi.NUM_LISTS)]
                d["in_total"] += tot[_abi.IN_NODES]
                d["out_total"] += tot[_abi.OUT_NODES]
                d["node_lists"] += sum(tot) - tot[_abi.IN_NODES]
                d["list_total"] += sum(tot) + tot[_abi.OUT_NODES]  # + indptr ones
        return layers

    stats = slot_stats((args.warmup + args.steps - 1) % NS)
    edges_per_round = sum(d["E"] for d in stats)
    total_edges = shard.sum_over_ranks(edges1 - edges0, dist, red_dev)
    iters = int(shard.sum_over_ranks(iters1 - iters0, dist, red_dev))
    value = total_edges / dt

    out = {
        "metric": "sampled_edges_per_sec",
        "value": value,
        "unit": "edges/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "iters_per_sec": iters / dt,
        "config": {
            "workload": "products-like synthetic (N=%d, mean degree %g, pareto degrees, seed %d), GraphSAGE fanout %s "
                        "(layer 0 = seeds' hop), minibatch %d, %d parts (v %% %d), engine-only (sample+slice)%s"
                        % (N, args.mean_deg, args.graph_seed, "/".join(map(str, fan)), B, P, P,
                           ", graph mode (real slice CSR + boundary lists)" if args.mode == "graph" else ""),
            "streams": S,
            "minibatches_per_step": S,
            "round_overlap": not args.serial_rounds,
            "device_lists": "int32 (host exports widen to the reference's int64)",
            "sampled_edges_per_minibatch": edges_per_round / S,
            "engine_device_bytes": eng.device_bytes(),
        },
    }

    # ---- end-to-end minibatch rate: slice + feature gather + forward/backward + Adam, one part per GPU
    def e2e_leg():
        from cslicer.train import Trainer, synthetic_node_data, use_tuned_gemms
        tuned = (not args.no_tuned_gemms) and use_tuned_gemms()
        # every rank generates only the feature/label rows of the nodes it owns (counter-based generator)
        feats = lambda own: synthetic_node_data(N, args.e2e_feat, args.e2e_classes, seed=0, rows=own)[0]  # noqa: E731
        labels = lambda own: synthetic_node_data(N, 1, args.e2e_classes, seed=0, rows=own)[1]            # noqa: E731
        tr = Trainer(indptr, indices, feats, labels, args.e2e_classes, rank=rank, world=world, fanouts=fan,
                     batch=B, streams=args.e2e_streams, hidden=args.e2e_hidden, device=device, dist=dist,
                     model=args.e2e_model, heads=args.e2e_heads, feat_dim=args.e2e_feat,
                     overlap=bool(args.e2e_overlap and world > 1 and args.e2e_model == "sage"))
        tr.set_nodes(perm)
        # warm-up (allocator, rng window, GEMM plans).  Steady state: every call slices one round ahead -- the warm-up
        # slices the timed call's first round, the timed call slices the first round of a (never trained) successor -- so
        # the timed region slices exactly as many rounds as it trains
        after = (48 + args.e2e_steps) % tr.n_batches
        tr.run(48, then=(48, args.e2e_steps))
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.reset_units()
        tr.run(args.e2e_steps, first_batch=48, then=(after, args.e2e_streams))
        torch.cuda.synchronize()
        barrier()
        t_e2e = shard.max_over_ranks(time.perf_counter() - t0, dist, red_dev)
        work = tr.step_work(args.e2e_steps) if args.e2e_model == "sage" else None
        tr_native = getattr(tr, "native", None) is not None or getattr(tr, "native_rank", None) is not None
        # ---- per-group device time of the step's own kernels: one more round with HIP events around every launch of the
        # native step (csl_sage_step_timing), outside the timed region.  The fractions below are work of a group / the
        # time of THAT group's kernels -- what the kernels sustain -- not / the step's wall time.
        groups = None
        if work is not None and getattr(tr, "native", None) is not None and world == 1:
            from cslicer import aggr
            n_t = args.e2e_streams
            aggr.step_timing(True)
            tr.reset_units()
            tr.run(n_t, first_batch=after)
            torch.cuda.synchronize()
            groups = aggr.step_timing_read()
            aggr.step_timing(False)
            work_t = tr.step_work(n_t)
        fused = bool(getattr(tr, "fused_deepest_layer", lambda: False)())
        tr.close()
        roof = None
        if work is not None:
            step_s = t_e2e / args.e2e_steps
            roof = {"whole_step": {
                "gemm_TFLOPs": work["gemm_flops"] / step_s / 1e12, "aggregation_GBs": work["aggregation_bytes"] / step_s / 1e9,
                "note": "algorithmic work of rank 0's step / the step's WALL time (GEMMs, HBM-bound gathers and small "
                        "kernels run one after the other, so these are not kernel rates)"}}
            if groups is not None:
                us = {g: 1e3 * groups[g][0] / n_t for g in groups}
                roof["us_per_step"] = us
                roof["launches_per_step"] = {g: groups[g][1] / n_t for g in groups}
                if us["gemm"] > 0:
                    a_ = work_t["gemm"]["flops"] / (us["gemm"] * 1e-6) / 1e12
                    roof["gemm"] = {"bound": "mfma", "what": "library fp32 GEMMs (hipBLASLt)", "flops_per_step": work_t["gemm"]["flops"],
                                    "achieved": a_, "peak": FP32_MFMA_PEAK_TFS, "unit": "TFLOP/s", "frac": a_ / FP32_MFMA_PEAK_TFS}
                if us["fused_forward"] > 0:
                    a_ = work_t["fused_forward"]["flops"] / (us["fused_forward"] * 1e-6) / 1e12
                    b_ = work_t["fused_forward"]["bytes"] / (us["fused_forward"] * 1e-6) / 1e9
                    roof["fused_forward"] = {"bound": "mfma", "what": "csl_sage_fwd_mfma_f32: gather -> fp32 MFMA -> bias + ReLU, deepest layer",
                                             "flops_per_step": work_t["fused_forward"]["flops"], "achieved": a_,
                                             "peak": FP32_MFMA_PEAK_TFS, "unit": "TFLOP/s", "frac": a_ / FP32_MFMA_PEAK_TFS,
                                             "hbm_bytes_per_step": work_t["fused_forward"]["bytes"], "hbm_GBs": b_,
                                             "hbm_frac": b_ / HBM_PEAK_GBS}
                if us["aggregation"] > 0:
                    b_ = work_t["aggregation"]["bytes"] / (us["aggregation"] * 1e-6) / 1e9
                    roof["aggregation"] = {"bound": "hbm", "what": "csl_sage_cat_f32 / csl_sage_cat_bwd_t_f32 (every row counted once)",
                                           "bytes_per_step": work_t["aggregation"]["bytes"], "achieved": b_, "peak": HBM_PEAK_GBS,
                                           "unit": "GB/s", "frac": b_ / HBM_PEAK_GBS}
                roof["note"] = ("work of a kernel group / device time of that group's kernels, HIP events around every launch of "
                                "the native step in a separate pass of %d steps (csl_sage_step_timing); PMC traffic of the "
                                "aggregation kernels: profiles/r3_e2e/" % n_t)
        return {
            "roofline": roof,
            # one native call per minibatch (csl_sage_fwd_bwd_f32, direct hipBLASLt GEMMs timed per shape) or the
            # kernels issued from Python through torch autograd (then with torch's GEMMs and these selections)
            "native_step": tr_native,
            "fused_deepest_layer": fused,
            "exchange_overlap": bool(args.e2e_overlap and world > 1 and args.e2e_model == "sage"),
            "tuned_gemm_selections": bool(tuned) and not tr_native,
            "iters_per_sec": args.e2e_steps / t_e2e, "ms_per_iter": 1e3 * t_e2e / args.e2e_steps,
            "steps": args.e2e_steps,
            "config": "split-parallel %s fanout %s, batch %d (global), %d part(s) = %d GPU(s), features %d, "
                      "hidden %d, classes %d, fp32, Adam; slice+gather+fwd+bwd+step; every call slices one round ahead, so the timed "
                      "region slices as many rounds as it trains" % (
                          "GraphSAGE" if args.e2e_model == "sage" else "GAT (%d heads)" % args.e2e_heads,
                          "/".join(map(str, fan)), B, world, world, args.e2e_feat, args.e2e_hidden, args.e2e_classes),
            "scaling": "strong",
        }

    # ---- BASELINE config 5's model on ONE GPU: 3-layer GAT, 8 heads x 32, fanout 10/10/10, batch 1024 (the reference has a
    # stub layer only, python/layers/dist_gatconv.py:3-6 + bipartite.py:75-80: defined GATConv-style here, "parity
    # unpinned"); the attention aggregation is HBM-bound gather work; the deepest layer aggregates the RAW feature rows per
    # head and projects the destinations only (csrc/gat_input.hip), the other layers' projections are library GEMMs
    def e2e_gat_leg():
        from cslicer.train import Trainer, synthetic_node_data
        gfan, heads, hid, n_st = (10, 10, 10), 8, 32, 32
        feats = lambda own: synthetic_node_data(N, args.e2e_feat, args.e2e_classes, seed=0, rows=own)[0]  # noqa: E731
        labels = lambda own: synthetic_node_data(N, 1, args.e2e_classes, seed=0, rows=own)[1]            # noqa: E731
        tr = Trainer(indptr, indices, feats, labels, args.e2e_classes, rank=0, world=1, fanouts=gfan, batch=B, streams=n_st,
                     hidden=hid, device=device, dist=None, model="gat", heads=heads, feat_dim=args.e2e_feat)
        tr.set_nodes(perm)
        steps = args.e2e_gat_steps
        after = (32 + steps) % tr.n_batches
        tr.run(32, then=(32, steps))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.run(steps, first_batch=32, then=(after, n_st))   # (slices as many rounds as it trains)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        units = [{k: v / max(tr.steps_done, 1) for k, v in u.items()} for u in tr.units]   # per step; layer 0 = deepest
        input_layer = "aggregate-then-project (csl_gat_in_*)" if getattr(tr, "gat_input", False) else "project-then-aggregate"
        tr.close()
        return {"iters_per_sec": steps / dt, "ms_per_iter": 1e3 * dt / steps, "steps": steps,
                "units_per_step": units, "input_layer": input_layer,
                "config": "3-layer GAT, %d heads x %d, fanout %s, batch %d, 1 part = 1 GPU, features %d, classes %d, fp32, "
                          "Adam; slice+gather+fwd+bwd+step (BASELINE config 5's shape on one GPU)" % (
                              heads, hid, "/".join(map(str, gfan)), B, args.e2e_feat, args.e2e_classes)}

    # ---- the same step DATA-parallel (several GPUs only): every GPU holds graph + features, trains 1/N of each minibatch
    # with the single-GPU native step, one gradient all-reduce per step (cslicer.train.DataParallelTrainer)
    def e2e_dp_leg(kind="sage"):
        """kind "gat": BASELINE config 5's model (8 heads x 32, fanout 10/10/10) data-parallel -- a rank's split-parallel GAT
        step is launch-bound whatever its share, so this is the form in which the attention model uses
        several GPUs of a node whose HBM holds graph + features"""
        from cslicer.train import DataParallelTrainer, synthetic_node_data
        feats = lambda own: synthetic_node_data(N, args.e2e_feat, args.e2e_classes, seed=0, rows=own)[0]  # noqa: E731
        labels = lambda own: synthetic_node_data(N, 1, args.e2e_classes, seed=0, rows=own)[1]            # noqa: E731
        gat = kind == "gat"
        dfan, dhid, dstreams, dsteps = ((10, 10, 10), 32, 32, args.e2e_gat_steps) if gat else (
            fan, args.e2e_hidden, args.e2e_streams, args.e2e_steps)
        tr = DataParallelTrainer(indptr, indices, feats, labels, args.e2e_classes, rank, world, dist, batch=B,
                                 fanouts=dfan, streams=dstreams, hidden=dhid, device=device, feat_dim=args.e2e_feat,
                                 model=kind, heads=8)
        tr.set_nodes(perm)
        warm = 48 if not gat else 32
        after = (warm + dsteps) % tr.n_batches
        tr.run(warm, then=(warm, dsteps))
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.run(dsteps, first_batch=warm, then=(after, dstreams))   # (slices as many rounds as it trains)
        torch.cuda.synchronize()
        barrier()
        t_dp = shard.max_over_ranks(time.perf_counter() - t0, dist, red_dev)
        tr.close()
        return {"iters_per_sec": dsteps / t_dp, "ms_per_iter": 1e3 * t_dp / dsteps,
                "steps": dsteps, "scaling": "strong",
                "config": "data-parallel %s fanout %s, batch %d (global; %d seeds per GPU), %d GPU(s), whole graph + "
                          "feature table on every GPU, %s + one gradient all-reduce per step; features %d, "
                          "hidden %d, classes %d, fp32, Adam" % (
                              "GAT (8 heads)" if gat else "GraphSAGE", "/".join(map(str, dfan)), B, (B + world - 1) // world,
                              world, "single-GPU attention step" if gat else "native step", args.e2e_feat, dhid,
                              args.e2e_classes)}

    live_pmc_kernel = None
    if rank == 0:
        # ---- per-kernel HIP-event timing pass (engine's own stream) for the roofline
        if not args.no_kernel_timing:
            eng.timing_enable(True)
            t_ev = time.perf_counter()
            for k in range(args.steps):
                run_round(args.warmup + args.steps + k)
            eng.sync()
            t_ev = time.perf_counter() - t_ev
            tim = eng.timing_read()
            eng.timing_enable(False)
            stats2 = slot_stats((args.warmup + 2 * args.steps - 1) % NS)
            per_kernel = {}
            for name, (ms, n) in tim.items():
                if n == 0 or name in ("k_mt19937_fill", "k_dupseeds", "k_graph"):
                    if n:
                        per_kernel[name] = {"ms_total": ms, "launches": n, "avg_us": 1e3 * ms / n}
                    continue
                nbytes = sum(kernel_bytes_device_layout(name, d, P, l, l == len(stats2) - 1,
                                                        stats2[l + 1] if l + 1 < len(stats2) else None)
                             for l, d in enumerate(stats2)) * args.steps
                per_kernel[name] = {"ms_total": ms, "launches": n, "avg_us": 1e3 * ms / n,
                                    "alg_bytes_per_launch": nbytes / n,
                                    "achieved_GBs": nbytes / (ms * 1e-3) / 1e9 if ms > 0 else None}
            dom = max((k for k in per_kernel if "achieved_GBs" in per_kernel[k]),
                      key=lambda k: per_kernel[k]["ms_total"])
            d = per_kernel[dom]
            out["roofline"] = {
                "bound": "hbm", "kernel": dom, "achieved": d["achieved_GBs"], "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": d["achieved_GBs"] / HBM_PEAK_GBS, "traffic": None,
                "avg_launch_us": d["avg_us"], "alg_bytes_per_launch": d["alg_bytes_per_launch"],
            }
            # HBM traffic of the same kernel from the committed rocprofv3 PMC passes
            # (profiles/<LATEST>/pmc.json: separate --pmc FETCH_SIZE / WRITE_SIZE runs of this
            # command at the same stream count; KiB)
            try:
                latest = open(os.path.join(ROOT, "profiles", "LATEST")).read().strip()
                pmc = json.load(open(os.path.join(ROOT, "profiles", latest, "pmc.json")))
                bj = json.loads(open(os.path.join(ROOT, "profiles", latest, "bench_fetch.json")).read())
                if bj["config"]["streams"] == S and dom in pmc["calls"]:
                    calls = pmc["calls"][dom]
                    # FETCH_SIZE = TCC_EA0_RDREQ x 64 B; a separate PMC pass (profiles/pmc_rdsize.sh) shows that
                    # > 99.8 % of this path's read requests are 128-B requests, gathers included, so the bytes
                    # actually fetched are 2 x FETCH_SIZE for every kernel (the guide's gfx950 correction)
                    out["roofline"]["traffic"] = 1024.0 * (2.0 * pmc["fetch_KiB_sum"][dom] + pmc["write_KiB_sum"][dom]) / calls
                    out["roofline"]["traffic_source"] = ("profiles/%s/pmc.json (2 x FETCH_SIZE + WRITE_SIZE; all read "
                                                         "requests are 128 B: profiles/pmc_rdsize.sh)" % latest)
                    out["roofline"]["traffic_note"] = (
                        "traffic above the algorithmic bytes is whole 128-B lines fetched for 4-20-byte random gathers "
                        "(neighbour picks in `indices`, row lookups), not re-reads: DESIGN.md section 6, Request budget")
            except Exception:
                pass
            out["kernels"] = per_kernel
            out["timing_pass_ms_per_step"] = 1e3 * t_ev / args.steps
            live_pmc_kernel = dom
        # whole path against the HBM roof, SURVEY 8(d) byte formula (reference data types)
        per_iter = [{k: v / S for k, v in d.items()} for d in stats]
        pb = path_bytes_reference_types(per_iter, P)
        out["path_roofline"] = {
            "alg_bytes_per_minibatch": pb,
            "achieved_GBs": pb * (args.steps * S) / dt / 1e9 if world == 1 else pb * iters / dt / 1e9 / world,
            "peak_GBs": HBM_PEAK_GBS,
        }
        out["path_roofline"]["frac"] = out["path_roofline"]["achieved_GBs"] / HBM_PEAK_GBS

        # ---- CPU baseline: the oracle (port of the reference algorithm) on host cores
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle as orc
            host = host_cpu_info()
            cores = host["usable_cpus"]      # every CPU this process may use (affinity mask / cgroup quota)
            probe = [perm[i * B:(i + 1) * B] for i in range(2)]
            sec1, e1 = orc.bench(indptr, indices, probe, n_parts=P, fanouts=fan, threads=1)
            per_iter_s = sec1 / 2
            nb = int(max(cores, min(n_batches, args.cpu_seconds / per_iter_s)))
            nb = (nb // cores) * cores
            sample = [perm[i * B:(i + 1) * B] for i in range(nb)]
            secT, eT = orc.bench(indptr, indices, sample, n_parts=P, fanouts=fan, threads=cores)
            out["cpu_baseline"] = {
                "value": eT / secT, "unit": "edges/s", "cores": cores, "kind": "port",
                "sample": "%d minibatches of the same workload (first %d of the permutation), %d threads, one "
                          "slicer per thread, incl. per-sample deep copy; %.2fs" % (nb, nb, cores, secT),
                "iters_per_sec": nb / secT,
                "host": host,
                "single_thread": {"value": e1 / sec1, "iters_per_sec": 2 / sec1, "sample": "2 minibatches"},
            }
            secN, eN = orc.bench(indptr, indices, sample, n_parts=P, fanouts=fan, threads=cores, deep_copy=False)
            out["cpu_baseline"]["without_deep_copy"] = {"value": eN / secN, "iters_per_sec": nb / secN}
            # the UNMODIFIED reference (oracle/_ref/ref_harness, built from /root/reference in the build
            # container) at its hard-coded constants (fanout 10/10/10, 4 parts), beside the port on the
            # same minibatches: how conservative the "port" baseline is on THIS host
            harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
            if os.path.exists(harness):
                try:
                    import re
                    import subprocess
                    from cslicer import l0
                    d = os.path.join(os.environ.get("CSLICER_BENCH_CACHE", "/tmp/cslicer_bench_cache"),
                                     "l0_n%d_d%g_s%d" % (args.nodes, args.mean_deg, args.graph_seed))
                    if not os.path.exists(os.path.join(d, "meta.txt")):
                        l0.write_l0(d, indptr, indices)
                    nb_ref = 4 * cores
                    batches = [perm[i * B:(i + 1) * B] for i in range(nb_ref)]
                    words = [nb_ref]
                    for b_ in batches:
                        words.append(len(b_))
                        words.extend(int(x) for x in b_)
                    bp = os.path.join(d, "bench_batches.bin")
                    np.array(words, dtype=np.int64).tofile(bp)
                    r = subprocess.run([harness, "bench", d, bp, str(cores)], capture_output=True, text=True,
                                       timeout=300)
                    ref_s = float(re.search(r"seconds=([0-9.]+)", r.stderr).group(1))
                    port_s, port_e = orc.bench(indptr, indices, batches, n_parts=4, fanouts=(10, 10, 10), threads=cores)
                    out["cpu_baseline"]["reference_calibration"] = {
                        "kind": "reference", "config": "fanout 10/10/10, 4 parts (the reference's constants), "
                        "%d minibatches of %d, %d threads" % (nb_ref, B, cores),
                        "reference_iters_per_sec": nb_ref / ref_s, "reference_edges_per_sec": port_e / ref_s,
                        "port_iters_per_sec": nb_ref / port_s, "port_over_reference": ref_s / port_s,
                    }
                except Exception as ex:  # the calibration is optional evidence, never fatal
                    out["cpu_baseline"]["reference_calibration"] = {"error": repr(ex)[:200]}
    if eng is not None:
        eng.close()
        eng = None
    if rank == 0 and world == 1 and live_pmc_kernel and not args.no_live_pmc:
        # roofline.traffic measured in THIS run (the engine above is closed: the children get the GPU's memory)
        try:
            name = "k_bucket<false>" if live_pmc_kernel == "k_bucket" else live_pmc_kernel   # (no workload table here)
            traffic, launches = live_pmc_traffic(args, name)
            out["roofline"]["traffic"] = traffic
            out["roofline"]["traffic_source"] = (
                "measured in this run: two child runs of this workload under rocprofv3 (--pmc FETCH_SIZE, --pmc WRITE_SIZE; "
                "%d launches each), 2 x FETCH_SIZE + WRITE_SIZE (every read request of this path is 128 B: "
                "profiles/pmc_rdsize.sh)" % launches)
        except Exception as ex:   # the committed PMC summary quoted above stays
            out["roofline"]["traffic_live_error"] = repr(ex)[:200]
    if rank == 0 and world == 1 and not args.no_compat:
        try:
            out["compat_path"] = compat_leg(indptr, indices, args, B)
        except Exception as ex:   # optional evidence, never fatal
            out["compat_path"] = {"error": repr(ex)[:200]}
    # The e2e leg runs LAST.  On several GPUs it is a job of RCCL collectives (all_to_all_single per layer +
    # gradient all-reduce) and runs under a watchdog: a collective that never completes, or a rank that raises,
    # must not take the slicer's result with it, and must not look like a success either.  Rank 0 then prints
    # the line with `e2e.error` and EVERY rank leaves with a non-zero code (torch.distributed.run reports it).
    if dist is not None:
        out["dist"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                       "launcher": "torch.distributed.run"}
    if args.e2e_steps > 0 and (world == 1 or not args.no_e2e_multi):
        if world == 1:
            out["e2e"] = e2e_leg()
            if args.hub_probe:
                try:
                    import subprocess
                    probe = os.path.join(ROOT, "profiles", "hub_probe.py")
                    res = {}
                    for kind in ("uniform", "hub"):
                        o = subprocess.run([sys.executable, probe, kind], capture_output=True, text=True, timeout=600).stdout
                        res[kind] = float(o.strip().splitlines()[-1].split(":")[1].split()[0])
                    res["hub_over_uniform"] = res["hub"] / res["uniform"]
                    res["config"] = "native step, 1 M nodes, degree 40, fanout 15/10/5, batch 1024 (profiles/hub_probe.py)"
                    out["e2e_hub_probe"] = res
                except Exception as ex:
                    out["e2e_hub_probe"] = {"error": repr(ex)[:300]}
            if args.e2e_gat_steps > 0 and args.e2e_model == "sage":
                try:
                    out["e2e_gat"] = e2e_gat_leg()
                except Exception as ex:   # a second, optional figure: never takes the line with it
                    out["e2e_gat"] = {"error": repr(ex)[:300]}
        else:
            partial = {}

            def both_legs():
                res = e2e_leg()                      # split-parallel: the reference's design (north_star)
                partial.update(res)
                if not args.no_e2e_dp and args.e2e_model == "sage":
                    res["data_parallel"] = e2e_dp_leg()
                    partial.update(res)
                    if args.e2e_gat_steps > 0:
                        res["data_parallel_gat"] = e2e_dp_leg("gat")
                return res
            run_guarded(both_legs, out, rank, world, args.e2e_timeout, default_store(dist), partial)
    finish(out, rank, dist)


if __name__ == "__main__":
    main()

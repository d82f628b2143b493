"""End-to-end split-parallel training step (cslicer.train.Trainer) on the GPU box.
(a) one rank: the loss of a learnable synthetic task goes down;
(b) two ranks (two processes sharing the GPU, gloo rehearsal backend, boundary all-to-all +
    gradient all-reduce) reproduce the single-process two-part run step for step."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _task(n=6000, F0=24, classes=5, seed=3):
    from cslicer import l0
    indptr, indices = l0.synth_graph(n, 14.0, seed=seed)
    rng = np.random.default_rng(seed)
    feats = rng.random((n, F0), dtype=np.float32)
    labels = np.argmax(feats[:, :classes], axis=1).astype(np.int64)   # learnable from the self features
    perm = rng.permutation(n)
    return indptr, indices, feats, labels, perm


def test_single_rank_training_learns():
    from cslicer.train import Trainer
    indptr, indices, feats, labels, perm = _task()
    t = Trainer(indptr, indices, feats, labels, 5, rank=0, world=1, fanouts=(10, 5), batch=256, streams=4,
                hidden=32, lr=1e-2)
    t.set_nodes(perm)
    losses = t.run(60)
    assert len(losses) == 60 and all(np.isfinite(losses))
    assert np.mean(losses[-10:]) < 0.7 * np.mean(losses[:5]), (losses[:5], losses[-10:])
    rep = t.report()
    for key in ("avg forward time", "batch slice time", "cache refresh time"):
        assert key in rep
    t.close()


def test_native_step_trains_like_the_python_step(monkeypatch):
    """The single-GPU GraphSAGE trainer runs its step as one native call (csl_sage_fwd_bwd_f32 + csl_adam_f32);
    CSLICER_PY_STEP=1 issues the same kernels through torch autograd with torch's GEMMs.  Same minibatches, same
    initial weights: the loss curves agree (fp32 GEMM algorithms differ: 2e-3 relative after 12 Adam steps)."""
    from cslicer.train import Trainer
    indptr, indices, feats, labels, perm = _task()
    curves = []
    for py in (False, True):
        if py:
            monkeypatch.setenv("CSLICER_PY_STEP", "1")
        t = Trainer(indptr, indices, feats, labels, 5, rank=0, world=1, fanouts=(10, 5), batch=256, streams=4,
                    hidden=32, lr=1e-2, seed=7)
        assert (t.native is None) == py
        t.set_nodes(perm)
        curves.append(t.run(12))
        t.close()
    np.testing.assert_allclose(curves[0], curves[1], rtol=2e-3)


def _dp_rank_main(rank, world, port, q, kind="sage"):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "occ-gnn_amd"))
    sys.path.insert(0, os.path.join(root, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cslicer.train import DataParallelTrainer
        from test_gpu_train import _task
        indptr, indices, feats, labels, perm = _task()
        # batch 250 over 3 ranks: chunks of 84 / 84 / 82; the epoch's last minibatch is short
        t = DataParallelTrainer(indptr, indices, feats, labels, 5, rank, world, dist, batch=250, fanouts=(10, 5),
                                streams=3, hidden=16, lr=1e-2, seed=3, model=kind, heads=2)
        assert (t.native is not None) == (kind == "sage")
        # the ranks' mt19937 streams differ (rank-dependent seed): the same seed list draws different neighbourhoods
        from cslicer import splitgnn
        t.eng.submit_seeds([perm[:64]], slot=0)
        probe = splitgnn.slices_of(t.eng, 0, 0)[1][0].in_nodes.cpu().numpy().copy()
        t.set_nodes(perm[:250 * 7 + 100 * (world > 2)])
        losses = t.run(24)
        tl = torch.tensor(losses, dtype=torch.float64)
        dist.all_reduce(tl)     # a minibatch's loss = sum of the ranks' shares (each already divided by its seed count)
        w = torch.cat([p.detach().reshape(-1).cpu() for p in t.model.parameters()])
        t.close()
        dist.barrier()
        q.put((rank, tl.tolist(), w.numpy(), probe))
        dist.destroy_process_group()
    except Exception as ex:
        q.put((rank, "error: " + repr(ex), None, None))
        raise


@pytest.mark.parametrize("world,kind", [(2, "sage"), (3, "sage"), (2, "gat")])
def test_data_parallel_ranks_stay_identical_and_learn(world, kind):
    """DataParallelTrainer: every rank trains its chunk of each minibatch with the native step (GraphSAGE) or the single-GPU
    attention step, gradients are summed by one all-reduce (gloo here, two / three processes on the one GPU): the replicas'
    weights stay bit-identical and the loss falls."""
    import torch.multiprocessing as mp
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_rank_main, args=(r, world, port, q, kind)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
    for g in got:
        assert not isinstance(g[1], str), g[1]
    for g in got[1:]:
        np.testing.assert_array_equal(g[2], got[0][2])
        np.testing.assert_allclose(g[1], got[0][1])
        # every rank samples its own chunk with its own mt19937 stream (seed 5489 + 7919 rank)
        assert g[3].shape != got[0][3].shape or not np.array_equal(g[3], got[0][3])
    # Everything here is a function of the seeds (node order, weights, one mt19937 stream per rank, a deterministic
    # step and reduction), so the trajectory is the same in every run; "learns" is read off windows of six steps of the
    # 24, not off single noisy minibatch losses
    losses = got[0][1]
    # (the attention model, 2 heads x 16, moves more slowly on this task: 1.678 -> 1.623 over the 24 steps)
    assert all(np.isfinite(losses)) and np.mean(losses[-6:]) < (0.9 if kind == "sage" else 0.98) * np.mean(losses[:6]), losses


def test_data_parallel_world_of_one_is_the_single_gpu_trainer():
    """dp_world = 1: seeds handed over as lists instead of ranges of the uploaded node order, nothing else differs."""
    from cslicer.train import DataParallelTrainer, Trainer
    indptr, indices, feats, labels, perm = _task()
    kw = dict(fanouts=(10, 5), streams=4, hidden=32, lr=1e-2, seed=7)
    a = Trainer(indptr, indices, feats, labels, 5, rank=0, world=1, batch=256, **kw)
    b = DataParallelTrainer(indptr, indices, feats, labels, 5, 0, 1, None, batch=256, **kw)
    for t in (a, b):
        t.set_nodes(perm)
    la, lb = a.run(10), b.run(10)
    np.testing.assert_array_equal(la, lb)
    a.close()
    b.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _table(n, world, seed=11):
    """A node -> part table that is NOT v % world (the METIS-map use case).  With three parts the last one owns
    only ~0.5 % of the nodes: most minibatches then hold no seed of that rank and its slices are nearly empty
    (empty lists, zero-row GEMMs, a rank without a loss term: it must still join every collective)."""
    rng = np.random.default_rng(seed)
    if world == 3:
        return rng.choice(3, size=n, p=[0.55, 0.445, 0.005]).astype(np.int32)
    return rng.integers(0, world, size=n).astype(np.int32)


def _rank_main(rank, world, port, q, overlap=False, kind="sage", table=False):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "occ-gnn_amd"))
    sys.path.insert(0, os.path.join(root, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _rank_body(rank, world, q, overlap, kind, dist, table)
    except Exception as ex:      # the parent must hear about it instead of waiting for the queue
        q.put((rank, "error: " + repr(ex), None))
        raise


def _rank_body(rank, world, q, overlap, kind, dist, table=False):
    from cslicer.train import Trainer
    from test_gpu_train import _table, _task
    indptr, indices, feats, labels, perm = _task()
    wl = _table(indptr.shape[0] - 1, world) if table else None
    # with a table the rank is handed loaders, not the full arrays: it must ask for its own rows only
    asked = []

    def load(a):
        def f(own):
            asked.append(np.asarray(own))
            return a[own]
        return f
    t = Trainer(indptr, indices, load(feats) if table else feats, load(labels) if table else labels, 5, rank=rank,
                world=world, fanouts=(10, 5), batch=128, streams=2, hidden=16, lr=1e-2, dist=dist, overlap=overlap,
                model=kind, heads=2, workload=wl, feat_dim=feats.shape[1])
    # GraphSAGE: one native call per step, the exchanges as callbacks (on the side stream with overlap); GAT: one fused
    # autograd node per layer (aggr.GatLayerRank)
    assert (t.native_rank is not None) == (kind == "sage")
    if table:
        assert len(asked) == 2 and all(bool((wl[o] == rank).all()) and len(o) == int((wl == rank).sum()) for o in asked)
    t.set_nodes(perm)
    losses = t.run(4)
    tl = torch.tensor(losses, dtype=torch.float64)
    dist.all_reduce(tl)     # global minibatch loss = sum of the ranks' shares
    w = torch.cat([p.detach().reshape(-1).cpu() for p in t.model.parameters()])
    t.close()
    dist.barrier()
    q.put((rank, tl.tolist(), w.numpy()))
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap,kind,table,world", [(False, "sage", False, 2), (True, "sage", False, 2),
                                                      (False, "gat", False, 2), (False, "sage", True, 2),
                                                      (True, "sage", True, 2), (False, "sage", True, 3),
                                                      (False, "gat", True, 3), (False, "sage", False, 4),
                                                      (True, "sage", True, 4)],
                         ids=["sequential", "side-stream-overlap", "gat", "partition-table", "partition-table-overlap",
                              "three-ranks-one-nearly-empty", "three-ranks-one-nearly-empty-gat", "four-ranks",
                              "four-ranks-table-overlap"])
def test_two_ranks_match_single_process_two_parts(overlap, kind, table, world):
    import torch.multiprocessing as mp
    from cslicer import _abi, splitgnn
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, q, overlap, kind, table)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda x: x[0])
    for r_ in res:
        assert not isinstance(r_[1], str), r_[1]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # single-process reference: same engine config, both parts in this process
    indptr, indices, feats, labels, perm = _task()
    dev = torch.device("cuda", 0)
    eng = _abi.Engine(indptr, indices, n_parts=world, fanouts=(10, 5), max_batch=128, n_streams=2, n_slots=2,
                      mode=_abi.MODE_GRAPH, workload=_table(indptr.shape[0] - 1, world) if table else None)
    eng.set_nodes(perm)
    torch.manual_seed(0)
    if kind == "gat":
        model = splitgnn.DistGATModel(feats.shape[1], 16, 5, heads=2, n_layers=2).to(dev)
    else:
        model = splitgnn.DistSAGEModel(feats.shape[1], 16, 5, n_layers=2).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    ft, lt = torch.from_numpy(feats).to(dev), torch.from_numpy(labels).to(dev)
    ref_losses = []
    for r in range(2):
        eng.submit_round(r * 2, 128, 2, slot=r & 1)
        for s in range(2):
            sl = splitgnn.slices_of(eng, s, r & 1)
            x = {g: ft[sl[1][g].in_nodes.long()] for g in range(world)}
            out = model.forward_parts(sl, x)
            loss = 0
            for g in range(world):
                seeds = sl[0][g].out_nodes[sl[0][g].owned_out_nodes.long()].long()
                loss = loss + torch.nn.functional.cross_entropy(out[g], lt[seeds], reduction="sum")
            loss = loss / 128
            opt.zero_grad()
            loss.backward()
            opt.step()
            ref_losses.append(float(loss))
    w_ref = torch.cat([p.detach().reshape(-1).cpu() for p in model.parameters()]).numpy()
    eng.close()
    # GraphSAGE: the ranks run the same kernels in the same order as the single-process reference.  GAT: a rank's layer is
    # one fused autograd node (aggr.GatLayerRank: csl_gemm_f32 with a timed algorithm choice, other summation orders than
    # the node-by-node reference), its gradients agree with the node-by-node form to 1e-4 of the largest entry
    # (test_fused_gat_rank_layer_matches_its_autograd_form) -- and Adam at lr 1e-2 turns noise-level gradient entries into
    # lr-sized steps, so the trajectories agree to 1e-3, not 1e-5
    for rank, losses, w in res:
        if kind == "gat":
            np.testing.assert_allclose(losses, ref_losses, rtol=1e-3, atol=1e-6)
            # Weights: Adam divides every entry's gradient by its own magnitude, so an entry whose gradient is 1e-4 of the
            # tensor's largest (padded class columns, a nearly empty rank's share: a sixth of the entries here) moves by lr
            # times the RELATIVE rounding difference of the two summation orders; four steps bound the walk by 4 lr, the mean
            # stays small.  (The gradients themselves are compared entry by entry in the A/B test named above.)
            d = np.abs(w - w_ref)
            assert d.max() <= 4 * 1e-2 + 1e-6 and d.mean() < 2e-3, (float(d.max()), float(d.mean()))
        else:
            np.testing.assert_allclose(losses, ref_losses, rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(w, w_ref, rtol=1e-4, atol=1e-5)


def test_batch_slice_script_prints_exp5_keys(capsys):
    """cslicer.batch_slice (counterpart of python/batch_slice_multi_gpu.py): the four lines
    experiments/exp5/populate_table.py:22-25 parses must match its regexes."""
    import re
    from cslicer import batch_slice
    batch_slice.main(["--nodes", "20000", "--mean-deg", "10", "--fsize", "16", "--batch-size", "1024",
                      "--fan-out", "10,10", "--num-epochs", "2", "--streams", "4"])
    out = capsys.readouterr().out
    for pat in (r"forward_time_per_epoch:(\d+\.\d+)", r"merge_time per epoch:(\d+\.\d+)",
                r"data transfer:(\d+\.\d+)", r"graph splitting time:(\d+\.\d+)"):
        m = re.findall(pat, out)
        assert len(m) == 1 and float(m[0]) >= 0.0, (pat, out)


@pytest.mark.parametrize("model", ["gcn", "gat"])
def test_train_cli_prints_reference_keys(capsys, model):
    """`python -m cslicer.train` with the reference's argument names (python/train.py:109-131) runs a few
    minibatches and prints the keys experiments/exp6/occ.py:21-23 parses."""
    from cslicer import train
    train.main(["--graph", "synthetic", "--model-name", model, "--fan-out", "5,10", "--num-layers", "2",
                "--num-hidden", "32", "--num-heads", "2", "--batch-size", "256", "--num-epochs", "1", "--max-steps", "4",
                "--cache-per", "0.25", "--num-workers", "0"])
    out = capsys.readouterr().out
    for key in ("avg forward time", "batch slice time", "cache refresh time"):
        assert key in out


def test_train_cli_on_an_l0_directory_with_partition_file(capsys, tmp_path):
    """`python -m cslicer.train --graph <L0 dir> --partition file`: features/labels are read memory-mapped (own rows
    only), ownership comes from partition_map_opt.bin (all zeros here: one rank), the epoch's tail minibatch is
    trained once (7 minibatches of 300 over 2000 nodes: 6 full + one of 200)."""
    from cslicer import l0, train
    n = 2000
    indptr, indices = l0.synth_graph(n, 9.0, seed=4)
    rng = np.random.default_rng(0)
    feats = rng.random((n, 12), dtype=np.float32)
    labels = rng.integers(0, 3, size=n).astype(np.int32)
    d = str(tmp_path / "tiny")
    l0.write_l0(d, indptr, indices, features=feats, labels=labels, partition=np.zeros(n, dtype=np.int32), num_classes=3)
    train.main(["--graph", d, "--partition", "file", "--fan-out", "4,6", "--num-layers", "2", "--num-hidden", "16",
                "--batch-size", "300", "--num-epochs", "1"])
    out = capsys.readouterr().out
    assert "epoch 0: 7 minibatches" in out and "avg forward time" in out


def _nccl_single_rank(port, q, kind):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "occ-gnn_amd"))
    sys.path.insert(0, os.path.join(root, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from cslicer.train import Trainer
    from test_gpu_train import _task
    indptr, indices, feats, labels, perm = _task()
    out = []
    try:
        _nccl_runs(Trainer, dist, indptr, indices, feats, labels, perm, kind, out)
    except Exception as ex:      # the parent must hear about it instead of waiting for the queue
        q.put(("error", repr(ex), None))
        raise
    q.put(out)
    dist.destroy_process_group()


def _nccl_runs(Trainer, dist, indptr, indices, feats, labels, perm, kind, out):
    for rank_path, overlap in ((False, False), (True, False), (True, True)):
        if overlap and kind != "sage":
            out.append(out[-1])
            continue
        t = Trainer(indptr, indices, feats, labels, 5, rank=0, world=1, fanouts=(10, 5), batch=128, streams=2,
                    hidden=16, lr=1e-2, dist=dist, model=kind, heads=2, rank_path=rank_path, overlap=overlap)
        t.set_nodes(perm)
        out.append(t.run(4))
        t.close()
    if kind == "sage":
        # the data-parallel trainer's one collective -- the all-reduce of the flat gradient buffer -- over RCCL too
        # (with a world of one it is skipped by default: forced here); same numbers as the plain single-GPU run
        from cslicer.train import DataParallelTrainer
        t = DataParallelTrainer(indptr, indices, feats, labels, 5, 0, 1, dist, batch=128, fanouts=(10, 5), streams=2,
                                hidden=16, lr=1e-2)
        t.grad_sync = lambda flat: dist.all_reduce(flat)
        t.set_nodes(perm)
        dp = t.run(4)
        t.close()
        np.testing.assert_allclose(dp, out[0], rtol=1e-6)


@pytest.mark.parametrize("kind", ["sage", "gat"])
def test_rccl_calls_on_one_gpu(kind):
    """The one-process-per-part code path with the REAL backend (nccl = RCCL) and a world of one: the
    all_to_all_single / all_reduce calls, their device tensors and their empty splits go through RCCL, and the
    losses equal the collective-free single-part path."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_single_rank, args=(_free_port(), q, kind))
    p.start()
    got = q.get(timeout=240)
    p.join(timeout=120)
    assert got[0] != "error", got[1]
    plain, ranked, overlapped = got
    assert p.exitcode == 0
    # GAT: the single-part path is the fused layer node (different summation orders in its reductions; gradients
    # agree to 1e-4, tests/test_gpu_gat.py), and Adam turns noise-level gradient entries into lr-sized steps
    tol = 1e-3 if kind == "gat" else 1e-5
    np.testing.assert_allclose(ranked, plain, rtol=tol, atol=1e-6)
    np.testing.assert_allclose(overlapped, plain, rtol=tol, atol=1e-6)   # side-stream exchange schedule (SAGE)


def _gat_layer_ab_main(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "occ-gnn_amd"))
    sys.path.insert(0, os.path.join(root, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cslicer import _abi, splitgnn
        from test_gpu_train import _table, _task
        indptr, indices, feats, labels, perm = _task()
        dev = torch.device("cuda", 0)
        eng = _abi.Engine(indptr, indices, n_parts=world, fanouts=(10, 5), max_batch=128, n_streams=1, mode=_abi.MODE_GRAPH,
                          workload=_table(indptr.shape[0] - 1, world), part_mask=1 << rank)
        eng.submit_seeds([perm[:128]])
        slices = splitgnn.slices_of(eng, parts=[rank])
        sl = slices[1][rank]                                          # the deepest layer's slice of this rank
        torch.manual_seed(0)
        model = splitgnn.DistGATModel(feats.shape[1], 16, 5, heads=2, n_layers=2).to(dev)
        with torch.no_grad():
            for conv in model.convs:
                conv.bias.normal_(0, 0.1)
        comm = splitgnn.DistComm(device=dev)
        x0 = torch.from_numpy(feats).to(dev)[sl.in_nodes.long()]
        w = torch.randn(slices[0][rank].n_owned, 5, generator=torch.Generator().manual_seed(rank)).to(dev)
        res = []
        for fused in (True, False):
            splitgnn._GAT_RANK_AUTOGRAD = not fused
            model.zero_grad()
            x = x0.clone().requires_grad_()
            out = model.forward_rank(slices, x, rank, comm)
            (out * w).sum().backward()
            res.append([out.detach().cpu().numpy(), x.grad.cpu().numpy()] + [p.grad.cpu().numpy() for p in model.parameters()])
        eng.close()
        dist.barrier()
        q.put((rank, res))
    except Exception as ex:
        q.put((rank, "error: " + repr(ex)))
        raise
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_fused_gat_rank_layer_matches_its_autograd_form(world):
    """aggr.GatLayerRank (a rank's whole DistGATConv layer as one autograd node: fused kernels, exchanges over the
    back-to-back lists, the backward by hand) against the node-by-node form, a two-layer model on the same slices, ranks
    over gloo on one GPU: output, input gradient and every parameter gradient of every rank."""
    import torch.multiprocessing as mp
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gat_layer_ab_main, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t_: t_[0])
    for p in procs:
        p.join(timeout=120)
    names = ["out", "grad x"] + ["grad %s of layer %d" % (n_, k) for k in range(2) for n_ in ("attn_l", "attn_r", "bias", "weight")]
    for rank, r_ in res:
        assert not isinstance(r_, str), r_
        assert len(r_[0]) == len(names)
        for name, a_, b_ in zip(names, r_[0], r_[1]):
            # (a nearly empty rank's gradients are sums over a handful of rows: entries of 1e-8 are rounding noise)
            scale = max(1e-1, float(np.abs(b_).max()))
            np.testing.assert_allclose(a_, b_, rtol=1e-4, atol=1e-5 * scale, err_msg="rank %d %s" % (rank, name))

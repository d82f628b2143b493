"""End-to-end split-parallel GraphSAGE training step on the slices: the counterpart of the
reference's python/train.py:19-106 (Adam, cross-entropy, 3-layer DistSAGEModel), one process per
GPU over torch.distributed (RCCL) instead of one process driving 4 GPUs with
torch.nn.parallel.replicate/gather.

Per rank g (= part g): an engine in CSL_MODE_GRAPH slices S minibatches per round (every rank
slices the same minibatches, deterministically; nothing about topology is communicated), the rank
keeps the features and labels of the nodes it owns, runs DistSAGEModel.forward_rank on its slice,
cross-entropy on the seeds it owns, backward (boundary gradients travel the reverse all-to-all),
all-reduce of the replicated weights' gradients, Adam.

The print keys of train.py:101-103 are kept (`avg forward time`, `batch slice time`,
`cache refresh time`; there is no feature cache here: features are resident in HBM, so the last
one is reported as 0).
"""
import time

import numpy as np
import torch

from . import _abi, _roctx, aggr, splitgnn


class Trainer(object):
    def __init__(self, indptr, indices, features, labels, n_classes, rank=0, world=1, fanouts=(15, 10, 5),
                 batch=1024, streams=8, hidden=256, lr=1e-3, device=0, dist=None, seed=0, overlap=False,
                 model="sage", heads=8, rank_path=None):
        """features: float32 [N, F] (host, the rank keeps only the rows it owns); labels int64 [N]."""
        self.rank, self.world, self.dist = rank, world, dist
        self.P = world
        self.dev = torch.device("cuda", device)
        torch.cuda.set_device(self.dev)
        N = indptr.shape[0] - 1
        self.N, self.B, self.S, self.L = N, batch, streams, len(fanouts)
        self.eng = _abi.Engine(indptr, indices, n_parts=self.P, fanouts=fanouts, max_batch=batch,
                               n_streams=streams, n_slots=2, device=device, mode=_abi.MODE_GRAPH)
        own = np.arange(rank, N, self.P)
        self.feat = torch.from_numpy(np.ascontiguousarray(features[own])).to(self.dev)   # row v // P of owner v % P
        self.labels = torch.from_numpy(np.ascontiguousarray(labels[own])).to(self.dev)
        torch.manual_seed(seed)      # identical replicated weights on every rank
        self.kind = model
        if model == "sage":
            self.model = splitgnn.DistSAGEModel(features.shape[1], hidden, n_classes, n_layers=self.L).to(self.dev)
        elif model == "gat":
            self.model = splitgnn.DistGATModel(features.shape[1], hidden, n_classes, heads=heads,
                                               n_layers=self.L).to(self.dev)
        else:
            raise ValueError("model must be 'sage' or 'gat'")
        try:     # one fused kernel per step (the for-each form is eight small launches, ~0.1 ms of GPU time)
            self.opt = torch.optim.Adam(self.model.parameters(), lr=lr, fused=True)
        except (RuntimeError, TypeError):
            self.opt = torch.optim.Adam(self.model.parameters(), lr=lr)
        # rank_path=True forces the one-process-per-part code (collectives included) even for a single part:
        # a way to run the RCCL calls on a one-GPU box
        self.rank_path = (world > 1) if rank_path is None else bool(rank_path)
        self.comm = splitgnn.DistComm(device=self.dev) if self.rank_path else None
        self.overlap = overlap
        self.t_forward = self.t_slice = 0.0
        self.steps_done = 0

    def set_nodes(self, nodes):
        self.eng.set_nodes(nodes)
        self.n_batches = (len(nodes) + self.B - 1) // self.B

    def _step(self, stream, slot):
        # ROCTX ranges = the reference's nvtx annotations (python/train.py:68, dist_sageconv.py:52-65)
        t0 = time.perf_counter()
        _roctx.push("slice")
        meta = self.eng.meta(stream, slot)
        slices = splitgnn.slices_of(self.eng, stream, slot, parts=[self.rank], device=self.dev, meta=meta)
        _roctx.pop()
        self.t_slice += time.perf_counter() - t0
        deep = slices[self.L - 1][self.rank]
        top = slices[0][self.rank]
        n_seeds = int(meta.n_seeds)
        fused = (not self.rank_path and self.kind == "sage" and self.P == 1 and self.feat.shape[1] % 4 == 0
                 and not splitgnn._NO_LOCAL_FUSE)
        if fused:
            # one GPU holding every node: the deepest layer reads the resident feature table through the slice's
            # in_nodes (no gathered input matrix), every layer is one fused node, the loss is one HIP pass
            t1 = time.perf_counter()
            _roctx.push("forward")
            logits = self.model.forward_local(slices, self.feat)
            loss = aggr.SoftmaxCE.apply(logits, top.out_nodes, self.labels, 1.0 / max(n_seeds, 1))
            _roctx.pop()
        else:
            # gather of owned input features (row v // P of the owner v % P), int32 indices, float4 row kernel
            _roctx.push("gather")
            rows = deep.in_nodes if self.P == 1 else torch.div(deep.in_nodes, self.P, rounding_mode="floor")
            x = aggr.gather_rows(self.feat, rows)
            _roctx.pop()
            t1 = time.perf_counter()
            _roctx.push("forward")
            if self.rank_path:
                if self.kind == "gat":
                    logits = self.model.forward_rank(slices, x, self.rank, self.comm)
                else:
                    logits = self.model.forward_rank(slices, x, self.rank, self.comm, overlap=self.overlap)
            else:
                logits = self.model.forward_parts(slices, {0: x})[0]
            seeds = top.out_nodes[top.owned_out_nodes.long()]  # the seeds this rank owns, frontier order
            y = self.labels[(seeds // self.P).long()]
            # mean over the WHOLE minibatch: sum of local losses / global seed count
            loss = torch.nn.functional.cross_entropy(logits, y, reduction="sum") / max(n_seeds, 1)
            _roctx.pop()
        self.t_forward += time.perf_counter() - t1
        _roctx.push("backward")
        self.opt.zero_grad(set_to_none=True)
        loss.backward()
        _roctx.pop()
        if self.rank_path:
            _roctx.push("grad_allreduce")
            # (a rank whose share of the minibatch produced no gradient for a parameter still takes part)
            flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1)
                              for p in self.model.parameters()])
            self.dist.all_reduce(flat)                    # replicated weights: sum of per-rank gradients
            o = 0
            for p in self.model.parameters():
                n = p.numel()
                if p.grad is None:
                    p.grad = flat[o:o + n].view_as(p).clone()
                else:
                    p.grad.copy_(flat[o:o + n].view_as(p))
                o += n
            _roctx.pop()
        _roctx.push("optimizer")
        self.opt.step()
        _roctx.pop()
        self.steps_done += 1
        return loss.detach()

    def run(self, n_steps, first_batch=0):
        """n_steps minibatches, S per engine round; the next round is sliced while this one trains."""
        losses = []
        rounds = (n_steps + self.S - 1) // self.S
        per_epoch = max(1, self.n_batches // self.S)

        def submit(r):
            torch.cuda.current_stream().synchronize()     # the slot's previous consumer has finished
            first = ((first_batch // self.S + r) % per_epoch) * self.S
            self.eng.submit_round(first, self.B, self.S, slot=r & 1)

        submit(0)
        done = 0
        for r in range(rounds):
            if r + 1 < rounds:
                submit(r + 1)
            for s in range(self.S):
                if done >= n_steps:
                    break
                losses.append(self._step(s, r & 1))
                done += 1
        torch.cuda.synchronize()
        return [float(x) for x in losses]

    def report(self):
        n = max(self.steps_done, 1)
        # keys of python/train.py:101-103 (parsed by experiments/exp6/occ.py:21-23)
        return ("avg forward time: %.6f sec\nbatch slice time: %.6f sec\ncache refresh time: %.6f sec"
                % (self.t_forward / n, self.t_slice / n, 0.0))

    def close(self):
        self.eng.close()


def synthetic_node_data(num_nodes, feat_dim, n_classes, seed=0):
    """features f32 U[0,1) and random labels (datasets are not available offline)."""
    rng = np.random.default_rng(seed)
    feats = rng.random((num_nodes, feat_dim), dtype=np.float32)
    labels = rng.integers(0, n_classes, size=num_nodes).astype(np.int64)
    return feats, labels


def main(argv=None):
    """Command line with the argument names of the reference's python/train.py:109-131 (same spelling, same
    defaults where they apply); one process per GPU under torchrun, or a single process.

        python -m cslicer.train --graph synthetic --model-name gcn --fan-out 5,10,15 --batch-size 1024
        python -m torch.distributed.run --nproc-per-node 4 --master-addr 127.0.0.1 -m cslicer.train --graph <L0 dir>

    --fan-out follows the reference (DGL) convention, input side first (python/train.py:128,
    batch_slice_multi_gpu.py:202): the LAST number is the hop from the seeds.  --graph: an L0 directory
    (cslicer.l0), a preset name (arxiv-like, products-like, papers-like) or `synthetic`.
    Accepted and ignored (no counterpart here): --cache-per (features are resident in HBM), --num-workers,
    --dropout, --debug, --log-every, --eval-every."""
    import argparse
    import os
    ap = argparse.ArgumentParser("split-parallel training on MI355X")
    ap.add_argument("--graph", type=str, default="synthetic")
    ap.add_argument("--log-every", type=int, default=20)
    ap.add_argument("--eval-every", type=int, default=5)
    ap.add_argument("--lr", type=float, default=0.01)
    ap.add_argument("--num-workers", type=int, default=0)
    ap.add_argument("--debug", type=bool, default=False)
    ap.add_argument("--cache-per", type=float)
    ap.add_argument("--model-name", default="gcn", help="gcn|gat")
    ap.add_argument("--num-epochs", type=int, default=2)
    ap.add_argument("--num-hidden", type=int, default=256)
    ap.add_argument("--num-layers", type=int, default=3)
    ap.add_argument("--num-heads", type=int, default=8)
    ap.add_argument("--fan-out", type=str, default="10,10,25")
    ap.add_argument("--batch-size", type=int, default=1032)
    ap.add_argument("--dropout", type=float, default=0)
    ap.add_argument("--max-steps", type=int, default=0, help="(extra) stop an epoch after this many minibatches")
    a = ap.parse_args(argv)
    from . import l0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group(backend=os.environ.get("CSLICER_DIST_BACKEND", "nccl"))
    if a.graph == "synthetic":
        indptr, indices = l0.synth_graph(200_000, 20.0, seed=0)
        feats, labels = synthetic_node_data(indptr.shape[0] - 1, 128, 40, seed=0)
        n_classes = 40
    elif a.graph in l0.PRESETS:
        n, d, fdim, n_classes = l0.PRESETS[a.graph]
        indptr, indices = l0.synth_graph(n, d, seed=0)
        feats, labels = synthetic_node_data(n, fdim, n_classes, seed=0)
    else:
        indptr, indices, meta = l0.read_l0(a.graph, mmap=False)
        n = meta["num_nodes"]
        feats = np.fromfile(os.path.join(a.graph, "features.bin"), dtype=np.float32).reshape(n, meta["feature_dim"])
        labels = np.fromfile(os.path.join(a.graph, "labels.bin"), dtype=np.int32).astype(np.int64)
        n_classes = meta["num_classes"]
    fan = tuple(int(x) for x in a.fan_out.split(","))[::-1]          # engine order: layer 0 = hop from the seeds
    if len(fan) != a.num_layers:
        fan = fan[:a.num_layers] if len(fan) > a.num_layers else fan
    kind = "gat" if a.model_name == "gat" else "sage"
    hidden = a.num_hidden // a.num_heads if kind == "gat" else a.num_hidden
    tr = Trainer(indptr, indices, feats, labels, n_classes, rank=rank, world=world, fanouts=fan, batch=a.batch_size,
                 streams=8, hidden=max(4, hidden // 4 * 4), lr=a.lr, device=local, dist=dist, model=kind, heads=a.num_heads)
    n = indptr.shape[0] - 1
    for epoch in range(a.num_epochs):
        tr.set_nodes(np.random.default_rng(epoch).permutation(n))
        steps = tr.n_batches if a.max_steps <= 0 else min(a.max_steps, tr.n_batches)
        t0 = time.time()
        losses = tr.run(steps)
        if rank == 0:
            print("epoch %d: %d minibatches in %.2f s, loss %.4f -> %.4f" % (epoch, steps, time.time() - t0, losses[0], losses[-1]))
    if rank == 0:
        print(tr.report())
    tr.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    import sys
    main(sys.argv[1:])

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
import os
import time

import numpy as np
import torch

from . import _abi, _roctx, aggr, shard, splitgnn


class Trainer(object):
    # result slots of the engine: round r lives in slot r % SLOTS.  Three, so that the slot the NEXT round is sliced
    # into was last read two rounds ago: waiting for that round's end event never drains the training stream (with
    # two slots the wait was for the steps just enqueued, an idle gap on the GPU once per round).
    SLOTS = 3

    def __init__(self, indptr, indices, features, labels, n_classes, rank=0, world=1, fanouts=(15, 10, 5),
                 batch=1024, streams=8, hidden=256, lr=1e-3, device=0, dist=None, seed=0, overlap=False,
                 model="sage", heads=8, rank_path=None, workload=None, feat_dim=None, rng_seed=5489):
        """Part `rank` of `world`.  Ownership = the engine's workload table (`workload` int32 [N], the METIS map of
        python/utils/sampler.py:64-134 / partition_map_opt.bin; None = v % world like pyfrontend.cpp:57): the rank
        keeps the feature and label rows of the nodes it owns, in ascending node order.

        features / labels: either host arrays over ALL nodes (float32 [N, F], int64 [N]; the rank copies its own
        rows) or callables `f(own_ids) -> rows` so that a rank never holds more than its share (papers100M:
        57 GB of features over 8 ranks); with a callable `features`, pass `feat_dim`."""
        self.rank, self.world, self.dist = rank, world, dist
        self.P = world
        self.dev = torch.device("cuda", device)
        torch.cuda.set_device(self.dev)
        N = indptr.shape[0] - 1
        self.N, self.B, self.S, self.L = N, batch, streams, len(fanouts)
        # rank_path=True forces the one-process-per-part code (collectives included) even for a single part:
        # a way to run the RCCL calls on a one-GPU box
        self.rank_path = (world > 1) if rank_path is None else bool(rank_path)
        if workload is not None:
            workload = np.ascontiguousarray(workload, dtype=np.int32)
            if workload.shape != (N,) or workload.min() < 0 or workload.max() >= world:
                raise ValueError("workload must be int32 [num_nodes] with values in [0, world)")
        # one process per part: only this rank's slices are materialised (the sampling itself is replicated).
        # One GPU holding every node (the fused GraphSAGE path): the engine also emits the slices by source, over
        # which the backward gathers its input gradients (CSLICER_NO_TRANSPOSE=1: atomic scatter instead, A/B switch).
        import os
        by_source = (not self.rank_path and self.P == 1 and model == "sage" and len(fanouts) > 1
                     and not os.environ.get("CSLICER_NO_TRANSPOSE"))
        eng_flags = _abi.FLAG_TRANSPOSE if by_source else 0
        # one process per part, GraphSAGE, native rank step: the part's slices by source too (the backward gathers its input
        # gradients over them instead of scattering them with atomics; CSLICER_NO_TRANSPOSE=1: A/B switch)
        if (self.rank_path and model == "sage" and len(fanouts) > 1 and not os.environ.get("CSLICER_NO_TRANSPOSE")
                and not os.environ.get("CSLICER_PY_STEP")):
            eng_flags = _abi.FLAG_TRANSPOSE
        if (not self.rank_path and self.P == 1 and model == "gat" and not os.environ.get("CSLICER_NO_TRANSPOSE")):
            # GAT aggregates PROJECTED features: every layer's sources take a gradient -- the deepest layer's too, unless
            # it runs aggregate-then-project on the raw feature rows (aggr.GatInputLayer: no source gradient at all)
            F_in = features.shape[1] if feat_dim is None and not callable(features) else feat_dim
            self.gat_input = (not splitgnn._NO_GAT_INPUT and not splitgnn._NO_LOCAL_FUSE and F_in is not None
                              and aggr.gat_input_ok(heads, F_in, fanouts[-1]))
            eng_flags = _abi.FLAG_TRANSPOSE | (0 if self.gat_input else _abi.FLAG_TRANSPOSE_ALL)
        self.eng = _abi.Engine(indptr, indices, n_parts=self.P, fanouts=fanouts, max_batch=batch,
                               n_streams=streams, n_slots=self.SLOTS, device=device, mode=_abi.MODE_GRAPH,
                               workload=workload, part_mask=(1 << rank) if self.rank_path else 0,
                               flags=eng_flags, rng_seed=rng_seed)
        if workload is None:
            own = np.arange(rank, N, self.P, dtype=np.int64)           # owner v % P holds v at local row v // P
        else:
            own = np.flatnonzero(workload == rank).astype(np.int64)   # ascending: local row = rank among owned
        self.n_own = own.shape[0]
        take = (lambda a: a(own)) if callable(features) else (lambda a: a[own] if self.P > 1 else a)
        f_own = np.ascontiguousarray(take(features), dtype=np.float32)
        l_own = np.ascontiguousarray((labels(own) if callable(labels) else (labels[own] if self.P > 1 else labels)),
                                     dtype=np.int64)
        if f_own.shape[0] != self.n_own or l_own.shape[0] != self.n_own:
            raise ValueError("features / labels do not cover the rank's %d nodes" % self.n_own)
        F = f_own.shape[1] if feat_dim is None else feat_dim
        self.feat = torch.from_numpy(f_own).to(self.dev)
        self.labels = torch.from_numpy(l_own).to(self.dev)
        del f_own, l_own
        # global node id -> local row of the owner (-1 elsewhere); a single part holds every node at its own id
        self.local_row = None
        if self.P > 1:
            lr_ = np.full(N, -1, dtype=np.int32)
            lr_[own] = np.arange(self.n_own, dtype=np.int32)
            self.local_row = torch.from_numpy(lr_).to(self.dev)
        torch.manual_seed(seed)      # identical replicated weights on every rank
        self.kind = model
        if model == "sage":
            self.model = splitgnn.DistSAGEModel(F, hidden, n_classes, n_layers=self.L).to(self.dev)
        elif model == "gat":
            self.model = splitgnn.DistGATModel(F, hidden, n_classes, heads=heads, n_layers=self.L).to(self.dev)
        else:
            raise ValueError("model must be 'sage' or 'gat'")
        # torch.optim.Adam's update in one HIP launch per step (the library's for-each form is eight small
        # launches, ~0.1 ms of GPU time per step; its fused form one of 42 us for these six small tensors)
        self.opt = aggr.Adam(list(self.model.parameters()), lr=lr)
        # the fused single-GPU GraphSAGE step as one native call per minibatch (CSLICER_PY_STEP=1: the same kernels
        # issued from Python through an autograd node, A/B switch and what the tests compare it with)
        self.native, self.grad_sync = None, None
        if (by_source and F % 4 == 0 and hidden % 4 == 0 and not os.environ.get("CSLICER_PY_STEP")
                and not splitgnn._NO_LOCAL_FUSE):
            self.native = aggr.SageStep(self.model, splitgnn.ROW_PAD, splitgnn.SPLIT_K)
            self._loss_ring, self._ring_at = torch.zeros((4096,), dtype=torch.float32, device=self.dev), 0
        self.comm = splitgnn.DistComm(device=self.dev) if self.rank_path else None
        self.overlap = overlap
        # one process per part: the same idea -- everything between two boundary exchanges is issued by one native call,
        # the exchanges come back as callbacks into self.comm, on its side stream with overlap=True (CSLICER_PY_STEP=1: the
        # autograd path)
        self.native_rank = None
        if (self.rank_path and model == "sage" and F % 4 == 0 and hidden % 4 == 0 and n_classes <= 256
                and not os.environ.get("CSLICER_PY_STEP")):
            self.native_rank = aggr.SageRankStep(self.model, splitgnn.ROW_PAD, splitgnn.SPLIT_K, self.comm,
                                                 overlap=overlap)
            self._loss_ring, self._ring_at = torch.zeros((4096,), dtype=torch.float32, device=self.dev), 0
        self.t_forward = self.t_slice = 0.0
        self.steps_done = 0
        self._round_base, self._slot_done, self._ahead = 0, [None] * self.SLOTS, None
        # units of the steps done (measurement only): per model layer k the output rows, source rows and edges
        self.units = [{"rows": 0, "src": 0, "edges": 0} for _ in range(self.L)]

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
        for k in range(self.L):
            sl = slices[self.L - 1 - k][self.rank]
            u = self.units[k]
            u["rows"] += sl.n_owned
            u["src"] += sl.n_in
            u["edges"] += sl.n_edges
        n_seeds = int(meta.n_seeds)
        fused = (not self.rank_path and self.kind == "sage" and self.P == 1 and self.feat.shape[1] % 4 == 0
                 and not splitgnn._NO_LOCAL_FUSE)
        if self.native_rank is not None:
            _roctx.push("step_native_rank")
            loss = self._loss_ring[self._ring_at:self._ring_at + 1]
            self._ring_at += 1
            rows = deep.in_nodes if self.local_row is None else self.local_row[deep.in_nodes.long()]
            seeds = top.out_nodes[top.owned_out_nodes.long()]          # the seeds this rank owns, frontier order
            self.native_rank([slices[self.L - 1 - k][self.rank] for k in range(self.L)], self.feat, rows, seeds,
                             self.local_row, self.labels, 1.0 / max(n_seeds, 1), loss)
            self.dist.all_reduce(self.native_rank.grads)               # replicated weights: sum of the ranks' shares
            self.opt.step(flat_grads=self.native_rank.grads)
            _roctx.pop()
            self.steps_done += 1
            return loss
        if self.native is not None:
            # forward, loss, backward: one native call; the optimizer: a second one on the flat gradient buffer
            _roctx.push("step_native")
            loss = self._loss_ring[self._ring_at:self._ring_at + 1]    # (run() sized the ring for its steps)
            self._ring_at += 1
            self.native([slices[self.L - 1 - k][self.rank] for k in range(self.L)], self.feat, self.labels,
                        1.0 / self._loss_den(stream, slot, n_seeds), loss)
            if self.grad_sync is not None:
                self.grad_sync(self.native.grads)         # (data-parallel: sum over the ranks' shares of the minibatch)
            self.opt.step(flat_grads=self.native.grads)
            _roctx.pop()
            self.steps_done += 1
            return loss
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
            rows = deep.in_nodes if self.P == 1 else self.local_row[deep.in_nodes.long()]
            if self.kind == "gat" and getattr(self, "gat_input", False):
                x = aggr.FeatureRows(self.feat, rows)       # the deepest layer reads the table through in_nodes
            elif self.kind == "gat" and not self.rank_path and self.P == 1 and not splitgnn._NO_LOCAL_FUSE:
                # straight into the row-padded buffer the fused GAT layer multiplies (no second copy of 0.3 GB)
                x = aggr.padded_rows(rows.numel(), self.feat.shape[1], splitgnn.ROW_PAD, self.dev)
                aggr.gather_rows(self.feat, rows, out=x.t[:x.n])
            else:
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
            # mean over the WHOLE minibatch: sum of local losses / global seed count
            if self.kind == "sage" and logits.shape[0] > 0:
                loss = aggr.SoftmaxCE.apply(logits, seeds, self.labels, 1.0 / max(n_seeds, 1), self.local_row)
            else:
                seeds = seeds.long()
                y = self.labels[seeds if self.P == 1 else self.local_row[seeds].long()]
                den = max(n_seeds, 1) if self.rank_path else self._loss_den(stream, slot, n_seeds)   # (data-parallel: global)
                loss = torch.nn.functional.cross_entropy(logits, y, reduction="sum") / den
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
            _roctx.pop()
            _roctx.push("optimizer")
            self.opt.step(flat_grads=flat)                # the reduced buffer is used in place
            _roctx.pop()
        elif self.grad_sync is not None:
            # data-parallel replicas on the autograd path (GAT): one all-reduce of the flat gradient, as the native step's
            _roctx.push("grad_allreduce")
            flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1)
                              for p in self.model.parameters()])
            self.grad_sync(flat)
            _roctx.pop()
            _roctx.push("optimizer")
            self.opt.step(flat_grads=flat)
            _roctx.pop()
        else:
            _roctx.push("optimizer")
            self.opt.step()
            _roctx.pop()
        self.steps_done += 1
        return loss.detach()

    def run(self, n_steps, first_batch=0, then=None):
        """n_steps minibatches starting at minibatch `first_batch` of the node order (wrapping around the epoch),
        up to S per engine round; the next round is sliced while this one trains.  A round never crosses the end
        of the epoch: the last round of an epoch holds the remaining n_batches % S minibatches, the last of them
        possibly short -- every minibatch of the epoch is trained exactly once.

        then = (first_batch, n_steps) of the run() call that will follow: its first round is sliced while this call's
        last round trains, so the follow-up call starts without waiting for the slicer (the caller MUST make that call
        next: the round is already submitted and has advanced the streams' mt19937 positions)."""
        losses = []
        plan = shard.round_plan(self.n_batches, self.S, first_batch, n_steps)
        if not plan:
            return losses

        if self.native is not None or self.native_rank is not None:
            if self._loss_ring.numel() < n_steps:
                self._loss_ring = torch.zeros((n_steps,), dtype=torch.float32, device=self.dev)
            self._ring_at = 0
        base = self._round_base        # rounds keep rotating through the result slots from call to call
        done = self._slot_done         # per slot: event behind the last step that read the slot

        def submit(r, entry):
            slot = (base + r) % self.SLOTS
            ev = done[slot]
            if ev is not None:
                ev.synchronize()                          # the slot's previous consumer has finished
            self._submit(entry[0], entry[1], slot)

        ahead, self._ahead = self._ahead, None
        if ahead is not None and ahead != plan[0]:
            raise RuntimeError("run(then=...) announced the round %r, this call starts with %r" % (ahead, plan[0]))
        if ahead is None:
            submit(0, plan[0])
        nxt = shard.round_plan(self.n_batches, self.S, then[0], then[1])[:1] if then else []
        for r in range(len(plan)):
            if r + 1 < len(plan):
                submit(r + 1, plan[r + 1])
            elif nxt:
                submit(r + 1, nxt[0])
                self._ahead = nxt[0]
            slot = (base + r) % self.SLOTS
            for s in range(plan[r][1]):
                losses.append(self._step(s, slot))
            ev = torch.cuda.Event()
            ev.record()
            done[slot] = ev
        self._round_base = (base + len(plan)) % self.SLOTS
        torch.cuda.synchronize()
        return [float(x) for x in losses]

    def _submit(self, first, n, slot):
        """minibatches [first, first + n) of the node order into result slot `slot`"""
        self.eng.submit_round(first, self.B, n, slot=slot)

    def _loss_den(self, stream, slot, n_seeds):
        """what the summed loss of this rank's seeds is divided by: the minibatch's seed count"""
        return max(n_seeds, 1)

    def reset_units(self):
        for u in self.units:
            u["rows"] = u["src"] = u["edges"] = 0

    def fused_deepest_layer(self):
        """whether the native step runs the deepest layer as the one fused gather -> fp32-MFMA kernel"""
        if self.native is None or os.environ.get("CSLICER_NO_MFMA_FWD"):
            return False
        fout, fin2 = self.model.convs[0].fc.weight.shape
        return aggr._lib().csl_sage_fwd_mfma_scratch(fin2 // 2, fout) > 0

    def step_work(self, steps):
        """Algorithmic work of one training step on this rank (fp32), by the groups csl_sage_step_timing measures, from
        the slices actually trained (units per model layer k: m output rows, src source rows, e edges):
          gemm          flops of the library GEMMs: forward (k >= 1, or every layer when the deepest is not fused),
                        weight gradient (every layer), input gradient (k >= 1), 2 m 2in out each
          fused_forward the deepest layer as one kernel: 2 m 2in out flops; HBM bytes: (m + e) in 4 of gathered
                        feature-table rows + m out 4 of result + m 2in 4 of operand kept for the weight gradient
          aggregation   HBM bytes of the gather kernels, every row counted ONCE (the per-edge re-reads of a layer's
                        sources are served by L2: the whole source matrix is a few tens of MB):
                        forward k >= 1: src in 4 read + m 2in 4 written; backward k >= 1 (by source): m 2in 4 read
                        + src in 4 (ReLU mask) read + src in 4 written
        The returned dict also has the totals `gemm_flops` (fused forward included) and `aggregation_bytes`."""
        fused = self.fused_deepest_layer()
        w = {"gemm": {"flops": 0.0}, "fused_forward": {"flops": 0.0, "bytes": 0.0}, "aggregation": {"bytes": 0.0}}
        for k, (u, conv) in enumerate(zip(self.units, self.model.convs)):
            if not hasattr(conv, "fc"):
                return None
            fout, fin2 = conv.fc.weight.shape
            fin = fin2 // 2
            m, src, e = u["rows"] / steps, u["src"] / steps, u["edges"] / steps
            g = 2.0 * m * fin2 * fout
            w["gemm"]["flops"] += g                                            # weight gradient
            if k > 0:
                w["gemm"]["flops"] += 2 * g                                    # forward + input gradient
                w["aggregation"]["bytes"] += (src * fin + m * fin2) * 4 + (m * fin2 + 2 * src * fin) * 4
            elif fused:
                w["fused_forward"]["flops"] += g
                w["fused_forward"]["bytes"] += (m + e) * fin * 4 + m * fout * 4 + m * fin2 * 4
            else:
                w["gemm"]["flops"] += g
                w["aggregation"]["bytes"] += (m + e) * fin * 4 + m * fin2 * 4
        w["gemm_flops"] = w["gemm"]["flops"] + w["fused_forward"]["flops"]
        w["aggregation_bytes"] = w["aggregation"]["bytes"] + w["fused_forward"]["bytes"]
        return w

    def report(self):
        n = max(self.steps_done, 1)
        # keys of python/train.py:101-103 (parsed by experiments/exp6/occ.py:21-23)
        return ("avg forward time: %.6f sec\nbatch slice time: %.6f sec\ncache refresh time: %.6f sec"
                % (self.t_forward / n, self.t_slice / n, 0.0))

    def close(self):
        self.eng.close()


class DataParallelTrainer(Trainer):
    """The same model trained DATA-parallel: every GPU holds the whole graph and feature table (288 GB of HBM per
    MI355X: ogbn-products is 1.5 GB, papers100M 64 GB), samples and trains its 1/W share of every minibatch with the
    single-GPU native step (GraphSAGE; the attention model: the single-GPU autograd step with its fused layers), and one
    all-reduce (RCCL) sums the 0.7 MB of gradients.  Not the reference's design -- its
    trainer is split-parallel (python/train.py, `Trainer` above with world > 1) -- but what the same slicer + step give
    when a GPU is large enough to hold everything: no per-layer boundary exchange, the only collective is the gradient
    all-reduce.  A minibatch of B seeds is dealt in contiguous chunks of ceil(B / W); the loss is the sum over a
    rank's seeds / B, so the reduced gradient is the whole minibatch's (each chunk samples its own neighbourhoods)."""

    def __init__(self, indptr, indices, features, labels, n_classes, dp_rank, dp_world, dist, batch=1024, **kw):
        self.dp_rank, self.dp_world, self.dp_dist, self.global_B = int(dp_rank), int(dp_world), dist, int(batch)
        self.chunk = (self.global_B + self.dp_world - 1) // self.dp_world
        # every rank samples its OWN chunk of a minibatch: the ranks' mt19937 streams must differ, or rank r and rank r'
        # consume the same draws for their chunks and the sampling noise of the reduced gradient does not average out as
        # 1 / W (the split-parallel trainer is the opposite case: every rank must slice the SAME minibatch, same seed)
        kw.setdefault("rng_seed", 5489 + 7919 * self.dp_rank)
        super().__init__(indptr, indices, features, labels, n_classes, rank=0, world=1, batch=self.chunk, dist=None,
                         rank_path=False, **kw)
        if self.native is None and self.kind != "gat":
            raise ValueError("the data-parallel trainer runs the native GraphSAGE step (feature and hidden widths "
                             "multiples of 4, at least two layers) or the single-GPU GAT step")
        if self.dp_world > 1:
            self.grad_sync = lambda flat: self.dp_dist.all_reduce(flat)
        self._den = {}

    def set_nodes(self, nodes):
        self.nodes = np.ascontiguousarray(nodes, dtype=np.int64)
        self.n_batches = (len(self.nodes) + self.global_B - 1) // self.global_B

    def _submit(self, first, n, slot):
        mine = []
        for j in range(n):
            a, b, total = shard.dp_chunk(first + j, self.global_B, len(self.nodes), self.dp_rank, self.dp_world)
            mine.append(self.nodes[a:b])
            self._den[(j, slot)] = max(total, 1)
        self.eng.submit_seeds(mine, slot=slot)

    def _loss_den(self, stream, slot, n_seeds):
        return self._den[(stream, slot)]


TUNED_GEMMS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tunableop_gfx950.csv")


def use_tuned_gemms(path=TUNED_GEMMS):
    """Let torch pick the library GEMM kernels recorded in `path` (PyTorch TunableOp results for the trainer's
    shapes on gfx950, written by profiles/tune_gemms.sh) instead of hipBLASLt's default heuristic: the deepest
    layer's forward GEMM runs in 81 us instead of 100, its weight gradient in 73 instead of 83.  Recorded
    selections only, no tuning at run time; shapes that are not in the file keep the default.  Process-wide."""
    if not os.path.exists(path) or not hasattr(torch.cuda, "tunable"):
        return False
    torch.cuda.tunable.enable(True)
    torch.cuda.tunable.tuning_enable(False)
    if hasattr(torch.cuda.tunable, "write_file_on_exit"):
        torch.cuda.tunable.write_file_on_exit(False)      # the recorded file is an input, never rewritten
    torch.cuda.tunable.set_filename(path)
    return True


def _mix64(x):
    """splitmix64 finaliser on uint64 arrays: a counter-based generator, so any subset of rows can be produced
    without the rows before it."""
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def synthetic_node_data(num_nodes, feat_dim, n_classes, seed=0, rows=None):
    """features f32 U[0,1) and random labels (datasets are not available offline).  Every value is
    a hash of (seed, node id, column), so `rows` (node ids) yields exactly the rows of the full matrix: each
    rank generates only the nodes it owns."""
    ids = np.arange(num_nodes, dtype=np.uint64) if rows is None else np.asarray(rows).astype(np.uint64)
    base = np.uint64((int(seed) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
    feats = np.empty((ids.shape[0], feat_dim), dtype=np.float32)
    step = max(1, (1 << 24) // max(feat_dim, 1))
    cols = np.arange(feat_dim, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        for lo in range(0, ids.shape[0], step):
            h = _mix64((ids[lo:lo + step, None] * np.uint64(feat_dim) + cols + base) * np.uint64(0x9E3779B97F4A7C15))
            feats[lo:lo + step] = (h >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))
        labels = (_mix64((ids + base) * np.uint64(0xD6E8FEB86659FD93)) % np.uint64(n_classes)).astype(np.int64)
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
    ap.add_argument("--partition", choices=("mod", "file"), default="mod",
                    help="(extra) node ownership: `mod` = v %% world (pyfrontend.cpp:57), `file` = the L0 directory's "
                         "partition_map_opt.bin (the METIS map of python/utils/sampler.py:64-134; its values must be "
                         "< the number of ranks)")
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
    workload, fdim = None, None
    if a.graph == "synthetic" or a.graph in l0.PRESETS:
        n, d, fdim, n_classes = (200_000, 20.0, 128, 40) if a.graph == "synthetic" else l0.PRESETS[a.graph]
        indptr, indices = l0.synth_graph(n, d, seed=0)
        # every rank generates only the rows of the nodes it owns
        feats = lambda own: synthetic_node_data(n, fdim, n_classes, seed=0, rows=own)[0]    # noqa: E731
        labels = lambda own: synthetic_node_data(n, 1, n_classes, seed=0, rows=own)[1]      # noqa: E731
        if a.partition == "file":
            raise SystemExit("--partition file needs an L0 directory")
    else:
        indptr, indices, meta = l0.read_l0(a.graph, mmap=False)
        n, fdim = meta["num_nodes"], meta["feature_dim"]
        # memory-mapped: a rank touches only the rows it owns
        fmap = np.memmap(os.path.join(a.graph, "features.bin"), dtype=np.float32, mode="r", shape=(n, fdim))
        lmap = np.memmap(os.path.join(a.graph, "labels.bin"), dtype=np.int32, mode="r", shape=(n,))
        feats = lambda own: np.asarray(fmap[own])                                            # noqa: E731
        labels = lambda own: np.asarray(lmap[own]).astype(np.int64)                          # noqa: E731
        n_classes = meta["num_classes"]
        if a.partition == "file":
            workload = np.fromfile(os.path.join(a.graph, "partition_map_opt.bin"), dtype=np.int32)
    fan = tuple(int(x) for x in a.fan_out.split(","))[::-1]          # engine order: layer 0 = hop from the seeds
    if len(fan) != a.num_layers:
        fan = fan[:a.num_layers] if len(fan) > a.num_layers else fan
    kind = "gat" if a.model_name == "gat" else "sage"
    hidden = a.num_hidden // a.num_heads if kind == "gat" else a.num_hidden
    tr = Trainer(indptr, indices, feats, labels, n_classes, rank=rank, world=world, fanouts=fan, batch=a.batch_size,
                 streams=8, hidden=max(4, hidden // 4 * 4), lr=a.lr, device=local, dist=dist, model=kind, heads=a.num_heads,
                 workload=workload, feat_dim=fdim)
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

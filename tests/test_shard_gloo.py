"""Multi-rank host logic on CPU: two processes, torch.distributed gloo, 127.0.0.1.
Checks that the minibatch sharding bench.py uses deals every minibatch of an
epoch to exactly one rank, and that the timing/unit reductions agree on all ranks."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "occ-gnn_amd"))
    from cslicer import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N, B, S = 10_000, 64, 8
    plan = shard.epoch_plan(N, B, S, world)
    mine = [(f, n) for (r, f, n) in plan if r == rank]
    got = sorted(b for f, n in mine for b in range(f, f + n))
    # exchange: every rank learns every rank's minibatches
    gathered = [None] * world
    dist.all_gather_object(gathered, got)
    everything = sorted(b for g in gathered for b in g)
    n_batches = (N + B - 1) // B
    ok_cover = everything == list(range(n_batches))
    # weak-scaling steps: ranks never collide within a step
    n_rounds, _ = shard.rounds_per_epoch(N, B, S)
    steps = [shard.round_of(k, rank, world, n_rounds) for k in range(5)]
    all_steps = [None] * world
    dist.all_gather_object(all_steps, steps)
    ok_disjoint = all(len({all_steps[r][k] for r in range(world)}) == world for k in range(5))
    t = shard.max_over_ranks(1.0 + rank, dist)
    e = shard.sum_over_ranks(100.0 * (rank + 1), dist)
    dist.barrier()
    q.put((rank, ok_cover, ok_disjoint, t, e))
    dist.destroy_process_group()


def test_two_rank_sharding_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_cover, ok_disjoint, t, e in res:
        assert ok_cover, "rank %d: epoch plan does not cover every minibatch exactly once" % rank
        assert ok_disjoint
        assert t == 2.0           # max over ranks of (1.0, 2.0)
        assert e == 300.0         # sum over ranks


def test_epoch_plan_single_rank_and_tail():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "occ-gnn_amd"))
    from cslicer import shard
    plan = shard.epoch_plan(1000, 100, 4, 1)           # 10 minibatches: 2 full rounds + tail of 2
    assert plan == [(0, 0, 4), (0, 4, 4), (0, 8, 2)]
    plan3 = shard.epoch_plan(1000, 100, 4, 3)
    assert [r for r, _, _ in plan3] == [0, 1, 2]
    assert shard.round_of(3, 1, 4, 5) == (3 * 4 + 1) % 5


def test_round_plan_covers_every_minibatch_once():
    """Trainer.run's rounds (cslicer/shard.py::round_plan): 194 minibatches with 8 streams = 24 full rounds + one
    round of 2; a second epoch starts over at minibatch 0; an unaligned start works too."""
    from cslicer import shard
    plan = shard.round_plan(194, 8, 0, 194)
    assert len(plan) == 25 and plan[-1] == (192, 2) and all(k == 8 for _, k in plan[:-1])
    seen = [b + j for b, k in plan for j in range(k)]
    assert seen == list(range(194))
    two = shard.round_plan(194, 8, 0, 2 * 194)
    seen = [b + j for b, k in two for j in range(k)]
    assert seen == list(range(194)) * 2
    part = shard.round_plan(10, 4, 7, 9)          # 7 8 9 | 0 1 2 3 | 4 5
    assert part == [(7, 3), (0, 4), (4, 2)]
    assert shard.round_plan(0, 4, 0, 5) == [] and shard.round_plan(5, 4, 0, 0) == []


def test_data_parallel_chunks_cover_every_minibatch_once():
    """shard.dp_chunk: the ranks' chunks of a minibatch are disjoint, in order, and together the whole minibatch --
    also for the short last minibatch of the node order, where trailing ranks get an empty chunk."""
    from cslicer import shard
    for n_nodes, B, W in [(1000, 250, 3), (1000, 256, 4), (1024, 1024, 8), (10, 4, 3), (7, 7, 1), (100, 33, 5)]:
        n_batches = (n_nodes + B - 1) // B
        seen = []
        for b in range(n_batches):
            total_expected = min(B, n_nodes - b * B)
            at = b * B
            for r in range(W):
                lo, hi, total = shard.dp_chunk(b, B, n_nodes, r, W)
                assert total == total_expected
                assert lo == at and hi >= lo and hi - lo <= (B + W - 1) // W
                seen.extend(range(lo, hi))
                at = hi
            assert at == b * B + total_expected
        assert seen == list(range(n_nodes))

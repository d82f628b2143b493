"""How minibatches are dealt to GPUs (ranks) and engine streams.

The cslicer path shards by minibatch: there is no exchange step between the
slicers of different GPUs, so ranks only agree on WHICH minibatches each one
takes.  Round k of the job is dealt to rank k % world (the same round-robin the
reference's driver uses for worker threads, cslicer/driver.cpp:69-71, one level
up); inside a round, minibatch j goes to engine stream j.
"""


def rounds_per_epoch(num_nodes, batch_size, streams):
    """Full rounds in one pass over the node order (a short tail round is dealt
    separately by `tail_round`)."""
    n_batches = (num_nodes + batch_size - 1) // batch_size
    return n_batches // streams, n_batches


def round_of(step, rank, world, n_rounds):
    """Global round index rank `rank` slices at its local step `step` (wraps
    around the epoch so a weak-scaling run of any length stays in range)."""
    if n_rounds < 1:
        raise ValueError("need at least one full round")
    return (step * world + rank) % n_rounds


def batches_of_round(round_idx, streams):
    """First minibatch index and count of a full round."""
    return round_idx * streams, streams


def epoch_plan(num_nodes, batch_size, streams, world):
    """Every (rank, first_batch, n_batches) of one epoch, each minibatch exactly once."""
    full, n_batches = rounds_per_epoch(num_nodes, batch_size, streams)
    plan = []
    for r in range(full):
        plan.append((r % world, r * streams, streams))
    tail = n_batches - full * streams
    if tail:
        plan.append((full % world, full * streams, tail))
    return plan


def round_plan(n_batches, streams, first_batch, n_steps):
    """(first minibatch, count) of the engine rounds that train `n_steps` minibatches starting at `first_batch`,
    wrapping around the epoch.  A round holds at most `streams` minibatches and never crosses the end of the
    epoch, so the tail n_batches % streams of every epoch is one short round and each minibatch of an epoch is
    taken exactly once (the reference's WorkerPool likewise hands out every batch once per epoch,
    WorkerPool.cpp:41-50)."""
    plan = []
    if n_batches < 1 or n_steps < 1:
        return plan
    b, left = first_batch % n_batches, n_steps
    while left > 0:
        k = min(streams, n_batches - b, left)
        plan.append((b, k))
        b = (b + k) % n_batches
        left -= k
    return plan


def dp_chunk(batch_index, global_batch, n_nodes, rank, world):
    """Data-parallel dealing of ONE minibatch: positions [lo, hi) of the node order that rank `rank` of `world` trains
    of minibatch `batch_index` (contiguous chunks of ceil(global_batch / world); the last minibatch of the order may
    be short, a chunk may be empty), and the minibatch's total seed count (the loss divisor on every rank)."""
    lo = batch_index * global_batch
    hi = min(lo + global_batch, n_nodes)
    chunk = (global_batch + world - 1) // world
    a = min(lo + rank * chunk, hi)
    return a, min(a + chunk, hi), max(hi - lo, 0)


def max_over_ranks(seconds, dist=None, device=None):
    """The job's wall time is the slowest rank's (bench.py contract)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    import torch
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, dist=None, device=None):
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())

"""The boundary exchange of the split-parallel model on CPU: two processes, gloo, 127.0.0.1.
Row blocks must arrive in peer order and gradients must travel the reverse route, including on a
rank that sends or receives nothing (its backward exchange must still run, or the peers deadlock)."""
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
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cslicer.splitgnn import DistComm
    comm = DistComm(device=torch.device("cpu"))
    H = 3
    ok = True
    # case 1: rank 0 sends 2 rows to rank 1, rank 1 sends 3 rows to rank 0
    n_send = [0, 2] if rank == 0 else [3, 0]
    n_recv = [0, 3] if rank == 0 else [2, 0]
    send = [None if n == 0 else (torch.arange(n * H, dtype=torch.float32).reshape(n, H) + 100 * rank).requires_grad_()
            for n in n_send]
    recv, tie = comm.all_to_all(send, n_recv, H)
    peer = 1 - rank
    want = torch.arange(n_recv[peer] * H, dtype=torch.float32).reshape(n_recv[peer], H) + 100 * peer
    ok &= recv[rank] is None and torch.equal(recv[peer].detach(), want)
    loss = (recv[peer] * (rank + 1)).sum() + tie
    loss.backward()
    # my sent rows were multiplied by (peer rank + 1) on the peer
    ok &= torch.equal(send[peer].grad, torch.full((n_send[peer], H), float(peer + 1)))
    # case 2: only rank 0 sends; rank 1 receives; both must finish backward
    n_send = [0, 4] if rank == 0 else [0, 0]
    n_recv = [0, 0] if rank == 0 else [4, 0]
    send = [None if n == 0 else torch.ones(n, H, requires_grad=True) for n in n_send]
    recv, tie = comm.all_to_all(send, n_recv, H)
    base = torch.zeros(1, requires_grad=True)
    loss = base.sum() + tie + (recv[0].sum() * 2 if rank == 1 else 0.0)
    loss.backward()
    if rank == 0:
        ok &= torch.equal(send[1].grad, torch.full((4, H), 2.0))
    dist.barrier()
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_boundary_exchange_two_ranks_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res

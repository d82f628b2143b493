#!/usr/bin/env python3
"""Host cost of torch Linear (hipBLASLt) when the row count changes every call, as it does in the
split-parallel training step (one GEMM per layer per minibatch, M = number of owned frontier nodes),
against row counts rounded up to a multiple of 4096 (shapes repeat, the heuristic cache hits)."""
import time

import torch

torch.manual_seed(0)
dev = "cuda"
lin = torch.nn.Linear(200, 256).to(dev)
rng = torch.Generator().manual_seed(1)
Ms = [int(x) for x in torch.randint(100_000, 120_000, (200,), generator=rng)]
big = torch.rand((131072, 200), device=dev)


def run(pad, backward):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for M in Ms:
        Mp = (M + pad - 1) // pad * pad if pad else M
        x = big[:Mp]
        if backward:
            x = x.detach().requires_grad_(True)
        y = lin(x)
        if backward:
            y.sum().backward()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / len(Ms) * 1e6, (t2 - t0) / len(Ms) * 1e6


lin(big[:1000]).sum().backward()          # one-time library initialisation out of the way
torch.cuda.synchronize()
for backward in (False, True):
    for pad in (0, 4096, 0, 4096):
        # fresh row counts every time: nothing is in the heuristic cache unless the padding put it there
        Ms = [int(x) for x in torch.randint(100_000, 120_000, (200,), generator=rng)]
        h, t = run(pad, backward)
        print("backward=%s pad=%5d: host issue %7.1f us/call, total %7.1f us/call" % (backward, pad, h, t))

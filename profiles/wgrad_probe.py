#!/usr/bin/env python3
"""Weight-gradient GEMM of a layer (gy^T @ cat: 256 x 200 or 256 x 512 result, 10^4-10^5-long reduction): the library's own
split-K candidates (one un-slabbed csl_gemm_f32, every solution timed) against row slabs + csl_sum_slabs_f32.
Run via gpurun:  CSLICER_GEMM_PLANS=0 CSLICER_GEMM_TUNE=all CSLICER_GEMM_LOG=1 python3 profiles/wgrad_probe.py
Round 2: deepest layer 200 us un-slabbed against 82-84 us with 16/32/64 slabs; middle layer 44 against 32-42."""
import sys, os, time
sys.path.insert(0, "occ-gnn_amd")
import torch
from cslicer import aggr
torch.manual_seed(0)
for rows in (81920, 12288):
    out_f, in_f = 256, (200 if rows == 81920 else 512)
    gy, x = torch.randn(rows, out_f, device="cuda"), torch.randn(rows, in_f, device="cuda")
    a = aggr.gemm(gy, x, transa=True)          # un-slabbed: the library's own split-K candidates
    b = aggr.weight_grad_slabs(gy, x, 32)
    print(rows, "max diff", float((a - b).abs().max()), flush=True)
    for name, f in (("direct", lambda: aggr.gemm(gy, x, transa=True)), ("slabs32", lambda: aggr.weight_grad_slabs(gy, x, 32)),
                    ("slabs64", lambda: aggr.weight_grad_slabs(gy, x, 64)), ("slabs16", lambda: aggr.weight_grad_slabs(gy, x, 16))):
        f(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): f()
        torch.cuda.synchronize(); print("  %s: %.1f us" % (name, (time.perf_counter() - t0) / 50 * 1e6), flush=True)

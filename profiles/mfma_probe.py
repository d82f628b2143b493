"""Deepest-layer forward at the bench shape: the fused gather -> fp32-MFMA kernel (csl_sage_fwd_mfma_f32) against the
two-kernel form (csl_sage_cat_f32 + library GEMM).  Prints microseconds per call (HIP events on torch's stream, which is
the stream both are launched on) and the TFLOP/s / gather GB/s they amount to.

    python profiles/mfma_probe.py [--rows 82000] [--deg 5] [--table 2449029] [--feat 100] [--out 256] [--no-cat]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "occ-gnn_amd"))
from cslicer import _abi, aggr  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=82000)
    ap.add_argument("--src", type=int, default=450000)
    ap.add_argument("--deg", type=int, default=5)
    ap.add_argument("--table", type=int, default=2449029)
    ap.add_argument("--feat", type=int, default=100)
    ap.add_argument("--out", type=int, default=256)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--relu-in", action="store_true")
    a = ap.parse_args()
    _abi.load()
    rng = np.random.default_rng(0)
    n, H, out = a.rows, a.feat, a.out
    n_pad = (n + 255) // 256 * 256
    dev = "cuda"
    deg = np.minimum(rng.integers(1, a.deg + 1, size=n) + (rng.random(n) < 0.8) * a.deg, a.deg)
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(deg, out=indptr[1:])
    indices = torch.from_numpy(rng.integers(0, a.src, size=int(indptr[-1])).astype(np.int32)).to(dev)
    indptr = torch.from_numpy(indptr.astype(np.int32)).to(dev)
    self_ids = torch.from_numpy(rng.integers(0, a.src, size=n).astype(np.int32)).to(dev)
    rowmap = torch.from_numpy(rng.choice(a.table, size=a.src, replace=False).astype(np.int32)).to(dev) if a.table else None
    x = torch.rand((a.table or a.src, H), device=dev)
    W = (torch.rand((out, 2 * H), device=dev) - 0.5) / np.sqrt(2 * H)
    b = torch.rand((out,), device=dev)
    E = int(indices.numel())

    def two():
        cat = aggr.sage_cat(x, self_ids, n, n_pad, indptr=indptr, indices=indices, rowmap=rowmap, relu_in=a.relu_in)
        return aggr.gemm(cat, W, transb=True, bias=b, relu=True)

    def fused(want_cat):
        return aggr.sage_fwd_mfma(x, self_ids, indptr, indices, W, b, n, n_pad, rowmap=rowmap, relu_in=a.relu_in,
                                  relu_out=True, want_cat=want_cat)

    def timeit(f):
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            f()
        e1.record()
        torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / a.iters

    if int(os.environ.get("CSLICER_MFMA_DBG", "0")) & 32:
        y = fused(False).cpu()
        bm = 64 if H <= 104 and not int(os.environ["CSLICER_MFMA_DBG"]) & 8 else 32
        t = torch.stack([y[w::bm, :2] for w in range(4)])      # [wave, tile, (cycles, 100 MHz ticks)]
        cyc, tick = t[..., 0].flatten(), t[..., 1].flatten()
        print("multiply loop: median %.0f shader cycles, %.0f ticks of 10 ns -> %.2f GHz; min/max cycles %.0f / %.0f" % (
            cyc.median(), tick.median(), float(cyc.median() / tick.median()) / 10, cyc.min(), cyc.max()))
        return
    if int(os.environ.get("CSLICER_MFMA_DBG", "0")) & 64:
        for _ in range(3):
            fused(True)
        torch.cuda.synchronize()
        _, cat = fused(True)
        w = cat.flatten()[:256 * 64 * 2 * 4].view(torch.int32).cpu().numpy().astype(np.int64).reshape(256, 64, 2, 4) & 0xFFFFFFFF
        S = int(w[0, 0, 0, 3])
        nst = S + 6
        t0 = w[:, 0, :, 0].min()
        print("steps per workgroup %d (tiles %d); kernel span %.1f us" % (nst, S, (w[:, :nst, :, 2].max() - t0) / 100))
        for role, name in ((0, "consumer"),):
            work = (w[:, :nst, role, 1] - w[:, :nst, role, 0]) / 100.0
            wait = (w[:, :nst, role, 2] - w[:, :nst, role, 1]) / 100.0
            print("  %s: work per step (us, median over workgroups): %s" % (name, " ".join("%.2f" % x for x in np.median(work, 0))))
            print("  %s: barrier wait per step:                      %s" % (name, " ".join("%.2f" % x for x in np.median(wait, 0))))
        print("  workgroup 0 step ends (us): %s" % " ".join("%.1f" % ((x - t0) / 100) for x in w[0, :nst, 0, 2]))
        return
    y2 = two()
    y1 = fused(False)
    err = float((y1 - y2).abs().max())
    flops = 2.0 * n_pad * 2 * H * out
    gbytes = (n + E) * H * 4
    print("rows %d (padded %d), edges %d, in %d, out %d; max |fused - two-kernel| = %.3g" % (n, n_pad, E, H, out, err))
    for name, f in (("cat + library GEMM", two), ("fused, operand also stored", lambda: fused(True)),
                    ("fused", lambda: fused(False))):
        us = timeit(f)
        print("%-28s %8.1f us   %6.1f TFLOP/s   gather %6.0f GB/s" % (name, us, flops / us / 1e6, gbytes / us / 1e3))


if __name__ == "__main__":
    main()

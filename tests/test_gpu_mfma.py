"""The fused GraphSAGE layer forward on the fp32 matrix cores (csl_sage_fwd_mfma_f32, csrc/sage_mfma.hip):
gather [self | mean of the CSR row] into LDS, multiply by W with v_mfma_f32_32x32x2_f32, bias + ReLU in the epilogue.

Reference: DistSageConv.forward, python/layers/dist_sageconv.py:66-80 over the slice CSR of
python/data/bipartite.py:61-67.  The reference ships no fixtures for it ("parity unpinned" by the reference): pinned
here against a dense torch float64 computation of the same formula, tolerance 1e-5 (north_star: aggregation outputs
within 1e-5 fp32), and exactly (integers) where the data allow it -- the operand-layout check the MFMA guide asks for
(asymmetric W, every output compared).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def aggr():
    from cslicer import _abi, aggr
    _abi.load()
    return aggr


def _case(rng, n, n_src, H, out, max_deg, table_rows=None, no_self_every=0, integers=False):
    deg = rng.integers(0, max_deg + 1, size=n)
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(deg, out=indptr[1:])
    indices = rng.integers(0, n_src, size=int(indptr[-1])).astype(np.int64)
    self_ids = rng.integers(0, n_src, size=n).astype(np.int64)
    if no_self_every:
        self_ids[::no_self_every] = -1
    rowmap = None
    rows = n_src
    if table_rows:
        rowmap = rng.choice(table_rows, size=n_src, replace=False).astype(np.int64)
        rows = table_rows
    if integers:
        x = rng.integers(-3, 4, size=(rows, H)).astype(np.float32)
        W = rng.integers(-2, 3, size=(out, 2 * H)).astype(np.float32)
        W[np.arange(out), np.arange(out) % (2 * H)] += 5.0      # asymmetric
        b = rng.integers(-4, 5, size=out).astype(np.float32)
    else:
        x = rng.standard_normal((rows, H)).astype(np.float32)
        W = (rng.standard_normal((out, 2 * H)) / np.sqrt(2 * H)).astype(np.float32)
        b = rng.standard_normal(out).astype(np.float32)
    return indptr, indices, self_ids, rowmap, x, W, b


def _reference(indptr, indices, self_ids, rowmap, x, W, b, n, n_pad, relu_in, relu_out, mean=True):
    """float64: returns (y [n_pad, out], cat [n_pad, 2H])"""
    x = torch.from_numpy(x).double()
    if relu_in:
        x = x.clamp(min=0)
    mp = (lambda i: torch.from_numpy(rowmap)[i]) if rowmap is not None else (lambda i: i)
    H = x.shape[1]
    cat = torch.zeros(n_pad, 2 * H, dtype=torch.float64)
    sid = torch.from_numpy(self_ids)
    has = sid >= 0
    cat[:n][has, :H] = x[mp(sid[has])]
    deg = torch.from_numpy(np.diff(indptr))
    rows = torch.repeat_interleave(torch.arange(n), deg)
    agg = torch.zeros(n, H, dtype=torch.float64)
    agg.index_add_(0, rows, x[mp(torch.from_numpy(indices))])
    cat[:n, H:] = agg / (deg.clamp(min=1).double().unsqueeze(1) if mean else 1.0)
    y = cat @ torch.from_numpy(W).double().t() + torch.from_numpy(b).double()
    if relu_out:
        y = y.clamp(min=0)
    return y, cat


def _run(aggr, case, n, n_pad, relu_in, relu_out, want_cat):
    indptr, indices, self_ids, rowmap, x, W, b = case
    dev = "cuda"
    return aggr.sage_fwd_mfma(torch.from_numpy(x).to(dev), torch.from_numpy(self_ids).int().to(dev),
                              torch.from_numpy(indptr).int().to(dev), torch.from_numpy(indices).int().to(dev),
                              torch.from_numpy(W).to(dev), torch.from_numpy(b).to(dev), n, n_pad,
                              rowmap=torch.from_numpy(rowmap).int().to(dev) if rowmap is not None else None,
                              relu_in=relu_in, relu_out=relu_out, want_cat=want_cat)


@pytest.mark.parametrize("H,out", [(4, 1), (8, 32), (100, 256), (100, 47), (104, 200), (128, 64), (128, 256), (36, 33), (256, 64), (256, 256), (96, 256)])
def test_operand_layout_exact_on_integers(aggr, H, out):
    """Small integers: every product and partial sum is exact in fp32, so the fused kernel must reproduce the float64
    result bit for bit wherever the mean is exact (degrees 0, 1, 2, 4: divisions by powers of two)."""
    rng = np.random.default_rng(100 * H + out)
    n, n_src = 333, 150
    case = list(_case(rng, n, n_src, H, out, 0, integers=True))
    # degrees drawn from {0, 1, 2, 4}
    deg = rng.choice([0, 1, 2, 4], size=n)
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(deg, out=indptr[1:])
    case[0], case[1] = indptr, rng.integers(0, n_src, size=int(indptr[-1])).astype(np.int64)
    case[2][::5] = -1
    n_pad = 384
    y, cat = _run(aggr, case, n, n_pad, False, False, True)
    yr, cr = _reference(*case, n, n_pad, False, False)
    assert torch.equal(cat.cpu().double(), cr)
    assert torch.equal(y.cpu().double(), yr)


@pytest.mark.parametrize("H,out,max_deg,relu_in,relu_out,table", [
    (100, 256, 5, False, True, 5000),      # the deepest layer of the products step: feature table through in_nodes
    (100, 256, 20, False, True, None),     # rows longer than the 8 edges requested together
    (128, 256, 10, True, True, None),      # a middle layer: previous pre-activation output, ReLU on the way in
    (256, 256, 10, True, True, None),      # ... at the bench's hidden width (rows wider than 128 columns: generic producers)
    (100, 256, 12, False, True, 4000),     # two edge passes per row
    (100, 256, 16, False, True, 4000),     # three
    (100, 256, 40, False, False, 3000),    # rows longer than the 16 staged entries: the index arrays are read directly
    (256, 47, 15, True, False, None),      # the top layer: 47 classes (two n-tiles, the second one partial)
    (4, 3, 3, False, False, 97),
    (64, 130, 9, True, True, 700),
])
def test_random_shapes_match_float64(aggr, H, out, max_deg, relu_in, relu_out, table):
    rng = np.random.default_rng(H * 1000 + out + max_deg)
    for n, n_pad in ((1, 1), (63, 64), (1000, 1024), (2049, 2304)):
        n_src = max(2, n // 2 + 3)
        case = _case(rng, n, n_src, H, out, max_deg, table_rows=table and max(table, n_src), no_self_every=7)
        y, cat = _run(aggr, case, n, n_pad, relu_in, relu_out, True)
        yr, cr = _reference(*case, n, n_pad, relu_in, relu_out)
        torch.testing.assert_close(cat.cpu().double(), cr, rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(y.cpu().double(), yr, rtol=1e-5, atol=1e-5)
        y2 = _run(aggr, case, n, n_pad, relu_in, relu_out, False)
        assert torch.equal(y2, y)


def test_matches_the_two_kernel_form_bit_for_bit_in_the_operand(aggr):
    """cat is what csl_sage_cat_f32 builds (same summation order: edge order, one multiply by 1/deg) and y is within
    1e-5 of csl_sage_cat_f32 + the library GEMM, at the bench shape's widths."""
    rng = np.random.default_rng(9)
    H, out, n, n_pad, n_src, table = 100, 256, 20000, 20224, 90000, 400000
    case = _case(rng, n, n_src, H, out, 5, table_rows=table)
    indptr, indices, self_ids, rowmap, x, W, b = case
    y, cat = _run(aggr, case, n, n_pad, False, True, True)
    dev = "cuda"
    cat2 = aggr.sage_cat(torch.from_numpy(x).to(dev), torch.from_numpy(self_ids).int().to(dev), n, n_pad,
                         indptr=torch.from_numpy(indptr).int().to(dev), indices=torch.from_numpy(indices).int().to(dev),
                         rowmap=torch.from_numpy(rowmap).int().to(dev))
    assert torch.equal(cat, cat2)
    y2 = aggr.gemm(cat2, torch.from_numpy(W).to(dev), transb=True, bias=torch.from_numpy(b).to(dev), relu=True)
    torch.testing.assert_close(y, y2, rtol=1e-5, atol=1e-5)
    yr, _ = _reference(*case, n, n_pad, False, True)
    torch.testing.assert_close(y.cpu().double(), yr, rtol=1e-5, atol=1e-5)


def test_rejects_unsupported_widths(aggr):
    from cslicer import _abi
    x = torch.zeros(8, 6, device="cuda")
    ip = torch.zeros(2, dtype=torch.int32, device="cuda")
    ix = torch.zeros(1, dtype=torch.int32, device="cuda")
    sid = torch.zeros(1, dtype=torch.int32, device="cuda")
    with pytest.raises(ValueError):
        aggr.sage_fwd_mfma(x, sid, ip, ix, torch.zeros(4, 12, device="cuda"), None, 1, 1)       # in % 4 != 0
    with pytest.raises(ValueError):
        aggr.sage_fwd_mfma(torch.zeros(8, 8, device="cuda"), sid, ip, ix, torch.zeros(300, 16, device="cuda"), None, 1, 1)
    with pytest.raises(ValueError):      # two operand tiles of this width do not fit a CU's 160 KB of LDS
        aggr.sage_fwd_mfma(torch.zeros(8, 320, device="cuda"), sid, ip, ix, torch.zeros(256, 640, device="cuda"), None, 1, 1)

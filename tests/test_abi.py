"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports
every entry point include/cslicer_hip.h declares, struct layouts agree, and
argument validation fails loudly without a GPU (no compute is attempted)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from cslicer import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "cslicer_hip.h")


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(csl_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = _abi.load()
    names = header_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(L, n), "libcslicer_hip.so does not export %s" % n
    assert sorted(_abi.SYMBOLS) == names


def test_aggregation_symbols_exported():
    from cslicer import aggr
    L = _abi.load()
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "cslicer_aggr.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(csl_[a-z_0-9]+)\s*\(", src)))
    assert names == sorted(aggr.SYMBOLS)
    for n in names:
        assert hasattr(L, n)


def test_abi_version_and_struct_layout():
    L = _abi.load()
    assert L.csl_abi_version() == _abi.ABI_VERSION
    # csl_layer_meta: 4 u32 + 10*(8+1) u32 offsets + 2*8*(8+1) pair offsets + 8 indptr lengths
    assert C.sizeof(_abi.LayerMeta) == 4 * (4 + 12 * 9 + 2 * 8 * 9 + 8 + 8)
    assert C.sizeof(_abi.SampleMeta) == 8 + 16 + 4 * C.sizeof(_abi.LayerMeta)
    assert L.csl_kernel_name(3).decode() == "k_sample"
    names = [L.csl_kernel_name(k).decode() for k in range(_abi.NUM_KERNELS)]
    assert "k_mt19937_fill" in names and "k_graph" in names and len(set(names)) == _abi.NUM_KERNELS


def test_header_cites_reference_interfaces():
    src = open(HEADER).read()
    for cite in ("pyfrontend.cpp:41-70", "slicer.cpp:69-81", "pybipartite.cpp:49-66",
                 "WorkerPool.cpp:41-50", "bipartite.h:9-26"):
        assert cite in src


def test_create_rejects_bad_config_loudly():
    indptr = np.array([0, 1, 2], dtype=np.int64)
    indices = np.array([1, 0], dtype=np.int64)
    with pytest.raises(_abi.CslError):
        _abi.Engine(indptr, indices, n_parts=9)
    with pytest.raises(_abi.CslError):
        _abi.Engine(indptr, indices, fanouts=(0, 10))
    with pytest.raises(_abi.CslError):
        _abi.Engine(np.array([0, 1, 3], dtype=np.int64), indices)  # indptr[N] != E
    with pytest.raises(_abi.CslError):
        _abi.Engine(indptr, indices, workload=np.array([0, 7], dtype=np.int32), n_parts=4)


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    indptr = np.array([0, 1, 2], dtype=np.int64)
    indices = np.array([1, 0], dtype=np.int64)
    with pytest.raises(_abi.CslError) as ei:
        _abi.Engine(indptr, indices)
    assert "no CPU path" in str(ei.value) or "HIP" in str(ei.value)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "occ-gnn_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".c")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt, os.path.join(dp, f)


def test_gat_input_layer_host_side_contract():
    """The attention input layer's pure-host entry points (include/cslicer_aggr.h, csrc/gat_input.hip): which shapes the
    MFMA projection covers, the padded head stride of dagg, scratch sizes, and argument validation that returns before any
    HIP call (no GPU here)."""
    from cslicer import aggr
    L = aggr._lib()
    assert L.csl_gat_in_max_degree() == 32
    assert L.csl_gat_in_proj_ok(8, 100, 32) == 1 and L.csl_gat_in_proj_ok(4, 128, 64) == 1 and L.csl_gat_in_proj_ok(1, 4, 16) == 1
    assert L.csl_gat_in_proj_ok(3, 100, 32) == 0        # heads: a power of two <= 8
    assert L.csl_gat_in_proj_ok(8, 100, 12) == 0        # D: whole 16-row tiles
    assert L.csl_gat_in_proj_ok(8, 132, 32) == 0 and L.csl_gat_in_proj_ok(8, 102, 32) == 0
    assert [L.csl_gat_in_proj_fpad(f) for f in (4, 64, 68, 100, 112, 116, 128)] == [64, 64, 112, 112, 112, 128, 128]
    assert L.csl_gat_in_bwd_scratch(0, 8, 100) == 0 and L.csl_gat_in_bwd_scratch(61850, 8, 100) % (2 * 800) == 0
    assert L.csl_gat_in_layer_fwd_scratch(8, 100) == 1600
    assert L.csl_gat_in_layer_bwd_scratch(61850, 8, 100, 32) >= (L.csl_elu_bwd_colsum_scratch(61850, 256) +
                                                                 L.csl_gat_in_proj_bwd_scratch(8, 100, 32) +
                                                                 L.csl_gat_in_bwd_scratch(61850, 8, 100) + 1600)
    null, st = C.c_void_p(0), C.c_void_p(0)
    # H = 3, F % 4 != 0, F > 128, a row longer than the kernels hold: refused, nothing launched
    assert L.csl_gat_in_fwd_f32(null, null, null, null, null, 100, 100, null, null, 3, 0.2, 10, 10, 10, null, null, st) == -1
    assert L.csl_gat_in_fwd_f32(null, null, null, null, null, 104, 102, null, null, 8, 0.2, 10, 10, 10, null, null, st) == -1
    assert L.csl_gat_in_fwd_f32(null, null, null, null, null, 132, 132, null, null, 8, 0.2, 10, 10, 10, null, null, st) == -1
    assert L.csl_gat_in_fwd_f32(null, null, null, null, null, 100, 100, null, null, 8, 0.2, 10, 10, 33, null, null, st) == -1
    assert L.csl_gat_in_fwd_f32(null, null, null, null, null, 100, 100, null, null, 8, 0.2, 10, 10, 10, null, null, st) == -1  # null x
    assert L.csl_gat_in_proj_f32(null, null, null, 10, 8, 100, 12, 1, null, 96, st) == -1
    assert L.csl_gat_in_layer_fwd_f32(null, null, null, null, null, 100, 100, null, null, null, null, 8, 12, 0.2, 1, 10, 10, 10,
                                      null, null, null, 96, null, st) == -1
    assert aggr.gat_input_ok(8, 100, 10) and not aggr.gat_input_ok(8, 100, 33) and not aggr.gat_input_ok(3, 100, 10)

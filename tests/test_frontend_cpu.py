"""Host-side logic of the pybind-surface mirror that needs no GPU."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_module_surface_matches_reference_names():
    import cslicer
    # dir(cslicer) of the reference module: bipatite, cslicer, sample, test_list, test_pyfront
    for n in ("bipatite", "cslicer", "sample", "test_list", "test_pyfront"):
        assert hasattr(cslicer, n)


def test_test_pyfront_and_test_list():
    import cslicer
    s = cslicer.test_pyfront()
    assert len(s.layers) == 3 and all(len(r) == 4 for r in s.layers)
    for l in range(3):
        for g in range(4):
            b = s.layers[l][g]
            assert b.gpu_id == g
            for n in ("in_nodes", "indptr", "out_nodes", "owned_out_nodes", "indices",
                      "self_ids_in", "self_ids_out"):
                assert getattr(b, n) == []
            assert b.from_ids == [[], [], [], []] and b.to_ids == [[], [], [], []]
    # pyfrontend.cpp:107: `l.append(10)` acts on a py::list HANDLE, i.e. on the caller's list
    # (checked on the compiled reference module: test_list([7, 8]) leaves the caller's list [7, 8, 10])
    arg = [7, 8]
    assert cslicer.test_list(arg) == [1, 2, 3, 4]
    assert arg == [7, 8, 10]
    import pytest
    with pytest.raises(TypeError):
        cslicer.test_list((1, 2))        # pybind11's py::list parameter takes a list only


def test_bipatite_attribute_value_semantics():
    import cslicer
    b = cslicer.bipatite(2, 4)
    b.in_nodes = [5, 6]
    got = b.in_nodes
    got.append(7)            # mutating the returned list does not touch the object
    assert b.in_nodes == [5, 6]
    b.from_ids = [[1], [], [2, 3], []]
    assert b.from_ids[2] == [2, 3]


def test_epoch_shuffle_is_libstdcxx_random_shuffle(tmp_path):
    # the reference calls std::random_shuffle with glibc's never-seeded rand()
    # (WorkerPool.cpp:40); pin the restatement against that very library call
    src = tmp_path / "rs.cpp"
    src.write_text(textwrap.dedent("""
        #include <algorithm>
        #include <cstdio>
        #include <vector>
        int main() {
          std::vector<long> v(1000);
          for (long i = 0; i < 1000; i++) v[i] = i;
          for (int epoch = 0; epoch < 2; epoch++) {
            std::random_shuffle(v.begin(), v.end());
            for (long x : v) printf("%ld ", x);
            printf("\\n");
          }
        }"""))
    exe = tmp_path / "rs"
    subprocess.run(["g++", "-std=c++14", "-O1", "-w", str(src), "-o", str(exe)], check=True)
    want = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r)
        from cslicer.frontend import epoch_shuffle
        v = np.arange(1000, dtype=np.int64)
        for _ in range(2):
            epoch_shuffle(v)
            print(" ".join(str(int(x)) for x in v) + " ")
        """ % os.path.join(ROOT, "occ-gnn_amd"))
    got = subprocess.run([sys.executable, "-c", code], check=True, capture_output=True, text=True).stdout
    assert got.split() == want.split()


def test_l0_roundtrip_and_meta(tmp_path):
    from cslicer import l0
    indptr, indices = l0.synth_graph(500, 8.0, seed=3)
    assert indptr[0] == 0 and indptr[-1] == indices.shape[0]
    rows = np.repeat(np.arange(500), np.diff(indptr))
    assert not np.any(rows == indices)                      # no self loops
    for v in (0, 17, 499):
        r = indices[indptr[v]:indptr[v + 1]]
        assert np.all(np.diff(r) >= 0)                      # rows sorted
    d = str(tmp_path / "g")
    meta = l0.write_l0(d, indptr, indices)
    assert meta["csum_offsets"] == int(indptr.sum()) and meta["csum_edges"] == int(indices.sum())
    a, b, m = l0.read_l0(d, mmap=False)
    np.testing.assert_array_equal(a, indptr)
    np.testing.assert_array_equal(b, indices)
    assert m["num_nodes"] == 500
    assert open(os.path.join(d, "meta.txt")).read().endswith("\n")


def test_native_pybind_module_surface():
    # the C++ host side: same module name and members as the reference's pybind11 module
    from conftest import load_native_module
    m = load_native_module()
    assert sorted(n for n in dir(m) if not n.startswith("_")) == ["bipatite", "cslicer", "sample",
                                                                  "test_list", "test_pyfront"]
    s = m.test_pyfront()
    assert len(s.layers) == 3 and all(len(r) == 4 for r in s.layers)
    b = s.layers[2][3]
    assert b.gpu_id == 3 and b.in_nodes == [] and b.from_ids == [[], [], [], []]
    b.in_nodes = [4, 5]          # def_readwrite: assignment works, reads copy
    got = b.in_nodes
    got.append(6)
    assert b.in_nodes == [4, 5]
    arg = [7, 8]
    assert m.test_list(arg) == [1, 2, 3, 4] and arg == [7, 8, 10]   # pyfrontend.cpp:107 mutates the caller's list
    import inspect
    doc = m.cslicer.__init__.__doc__
    for a in ("name", "queue_size", "no_worker_threads", "number_of_epochs", "minibatch_size"):
        assert a in doc


def test_native_module_missing_dataset_raises(tmp_path):
    from conftest import load_native_module
    m = load_native_module()
    import pytest
    with pytest.raises(RuntimeError):
        m.cslicer("nope", 16, 2, 1, 64, data_root=str(tmp_path))


def test_l0_converter_from_edge_list(tmp_path):
    # counterpart of python/utils/convert_dgl_dataset.py:42-49: in-neighbour CSR, no self loops, rows sorted
    from cslicer import l0
    src = np.array([0, 1, 2, 2, 3, 3, 1, 4])
    dst = np.array([1, 2, 2, 0, 0, 1, 2, 4])          # (2,2) and (4,4) are self loops
    indptr, indices = l0.from_edge_list(5, src, dst)
    assert indptr.tolist() == [0, 2, 4, 6, 6, 6]
    assert indices.tolist() == [2, 3, 0, 3, 1, 1]       # duplicates (1->2 twice) kept, rows sorted
    ip2, ix2 = l0.from_edge_list(5, src, dst, symmetric=True)
    assert ip2[-1] == 2 * 6 and (np.diff(ip2) >= 0).all()
    np.savez(tmp_path / "e.npz", src=src, dst=dst, num_nodes=np.int64(5))
    import subprocess
    r = subprocess.run([sys.executable, "-m", "cslicer.l0", "convert", str(tmp_path / "e.npz"), str(tmp_path / "out")],
                       env=dict(os.environ, PYTHONPATH=os.path.join(ROOT, "occ-gnn_amd")), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    a, b, m = l0.read_l0(str(tmp_path / "out"), mmap=False)
    np.testing.assert_array_equal(a, indptr)
    np.testing.assert_array_equal(b, indices)
    with pytest.raises(ValueError):
        l0.from_edge_list(3, [0, 5], [1, 2])

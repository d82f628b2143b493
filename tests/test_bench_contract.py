"""bench.py's JSON line (the committed one from the last GPU run, profiles/<LATEST>/) carries every
field the driver's contract asks for, with consistent values."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest():
    latest = open(os.path.join(ROOT, "profiles", "LATEST")).read().strip()
    path = os.path.join(ROOT, "profiles", latest, "bench_default.json.log")
    line = [l for l in open(path) if l.startswith("{")][-1]
    return json.loads(line)


def test_bench_json_contract():
    d = _latest()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "sampled_edges_per_sec" and d["unit"] == "edges/s" and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert "15/10/5" in d["config"]["workload"] and "2449029" in d["config"]["workload"]
    # value and ms_per_step describe the same timed region
    edges_per_step = d["config"]["sampled_edges_per_minibatch"] * d["config"]["minibatches_per_step"]
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 / edges_per_step - 1.0) < 0.02
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
    assert d["value"] > 50 * c["value"]      # sanity: the GPU path is not the CPU path in disguise


def test_rocprof_stats_agree_with_bench_kernel_time():
    """DESIGN/contract: the rocprofv3 --stats average of the dominant kernel (same command, same stream
    count, serial rounds) agrees with the HIP-event average bench.py reports."""
    import csv
    d = _latest()
    latest = open(os.path.join(ROOT, "profiles", "LATEST")).read().strip()
    rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", latest, "kernel_stats.csv"))))
    dom = d["roofline"]["kernel"]
    row = [r for r in rows if ("::" + dom + "(") in r["Name"]][0]
    prof_us = float(row["AverageNs"]) / 1e3
    assert abs(prof_us / d["roofline"]["avg_launch_us"] - 1.0) < 0.10, (prof_us, d["roofline"]["avg_launch_us"])


def test_bench_flags_parse():
    """The command lines DESIGN/README quote, and the driver's own, are accepted by bench.py's parser."""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    old = sys.argv
    try:
        for argv in (["--gpus", "8", "--steps", "20", "--warmup", "5"],
                     ["--mode", "graph"],
                     ["--fanout", "10,10,10", "--e2e-model", "gat", "--e2e-hidden", "32"],
                     ["--no-e2e-multi", "--e2e-timeout", "30"],
                     ["--serial-rounds", "--no-cpu-baseline", "--no-kernel-timing", "--e2e-steps", "0"],
                     ["--no-compat", "--no-tuned-gemms", "--e2e-streams", "16", "--same-batch"],
                     ["--gpus", "2", "--selftest-dist", "ok"]):
            sys.argv = ["bench.py"] + argv
            a = mod.parse()
            assert a.steps > 0 and a.mode in ("strict", "graph")
    finally:
        sys.argv = old


def test_split_k_linear_on_cpu():
    """splitgnn._SplitKLinear is plain torch: its slab-wise weight gradient equals F.linear's on the CPU too."""
    import sys
    import torch
    sys.path.insert(0, os.path.join(ROOT, "occ-gnn_amd"))
    from cslicer import splitgnn
    torch.manual_seed(0)
    m = splitgnn.ROW_PAD
    x = torch.rand((m, 24), dtype=torch.float64, requires_grad=True)
    w = torch.rand((8, 24), dtype=torch.float64, requires_grad=True)
    b = torch.rand((8,), dtype=torch.float64, requires_grad=True)
    gy = torch.rand((m, 8), dtype=torch.float64)
    g0 = torch.autograd.grad(torch.nn.functional.linear(x, w, b), (x, w, b), gy)
    g1 = torch.autograd.grad(splitgnn._SplitKLinear.apply(x, w, b), (x, w, b), gy)
    for a_, b_ in zip(g0, g1):
        assert torch.allclose(a_, b_, rtol=1e-12, atol=1e-12)
    (gx, gw) = torch.autograd.grad(splitgnn._SplitKLinear.apply(x, w, None), (x, w), gy)
    assert torch.allclose(gw, g0[1], rtol=1e-12, atol=1e-12)


def test_pmc_counter_sum_reads_a_rocprofv3_counter_file(tmp_path):
    """bench.pmc_counter_sum: per-kernel sum and launch count from rocprofv3's counter_collection.csv (what
    roofline.traffic is computed from in the run)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    f = tmp_path / "1_counter_collection.csv"
    f.write_text(
        "Correlation_Id,Dispatch_Id,Agent_Id,Queue_Id,Process_Id,Thread_Id,Grid_Size,Kernel_Id,Kernel_Name,"
        "Workgroup_Size,LDS_Block_Size,Scratch_Size,VGPR_Count,Accum_VGPR_Count,SGPR_Count,Counter_Name,Counter_Value,"
        "Start_Timestamp,End_Timestamp\n"
        '1,1,4,1,9,9,1024,7,"(anonymous namespace)::k_sample((anonymous namespace)::LArgs)",256,0,0,48,0,80,FETCH_SIZE,1000.5,1,2\n'
        '2,2,4,1,9,9,1024,7,"(anonymous namespace)::k_sample((anonymous namespace)::LArgs)",256,0,0,48,0,80,FETCH_SIZE,2000,3,4\n'
        '3,3,4,1,9,9,1024,8,"void (anonymous namespace)::k_bucket<false>((anonymous namespace)::LArgs)",512,0,0,46,0,80,FETCH_SIZE,50,5,6\n'
        '4,4,4,1,9,9,1024,7,"(anonymous namespace)::k_sample((anonymous namespace)::LArgs)",256,0,0,48,0,80,WRITE_SIZE,7,7,8\n')
    assert bench.pmc_counter_sum(str(f), "FETCH_SIZE", "k_sample") == (3000.5, 2)
    assert bench.pmc_counter_sum(str(f), "FETCH_SIZE", "k_bucket<false>") == (50.0, 1)
    assert bench.pmc_counter_sum(str(f), "WRITE_SIZE", "k_emit") == (0.0, 0)

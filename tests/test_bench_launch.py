"""bench.py's launch forms and its shutdown path, on CPU (gloo, two ranks).

`--selftest-dist` runs everything of bench.py that is not GPU work: the launcher decision, the child
torch.distributed.run the parent starts for `--gpus N` without WORLD_SIZE, rendezvous, the reductions of
cslicer/shard.py, the guarded e2e leg (stand-in body), the single JSON line and barrier + destroy."""
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _mod():
    spec = importlib.util.spec_from_file_location("bench_mod", BENCH)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _run(argv, env_extra=None, timeout=180):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, BENCH] + argv, capture_output=True, text=True, timeout=timeout, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r.returncode, [json.loads(l) for l in lines], r.stderr


def test_launch_plan():
    m = _mod()
    assert m.launch_plan(1, {}) == ("rank", 1)
    assert m.launch_plan(8, {}) == ("spawn", 8)                      # the form the driver used for N = 1
    assert m.launch_plan(8, {"WORLD_SIZE": "8"}) == ("rank", 8)      # under torch.distributed.run
    assert m.launch_plan(1, {"WORLD_SIZE": "4"}) == ("rank", 4)      # the environment wins over the flag
    assert m.E2E_FAILED_RC != 0


def test_self_launch_two_ranks_clean_exit():
    rc, lines, err = _run(["--gpus", "2", "--selftest-dist", "ok", "--e2e-timeout", "60"])
    assert rc == 0, err[-2000:]
    assert len(lines) == 1, lines
    d = lines[0]
    assert d["n_gpus"] == 2 and d["dist"]["world_size"] == 2
    assert d["e2e_iters_per_sec"] == 1.0 and d["e2e_scaling"] == "strong"
    assert d["value"] == 2 * 10.0 / 0.002          # sum of units over ranks / max of times over ranks


def test_under_torchrun_two_ranks_clean_exit():
    """The driver's own form: python -m torch.distributed.run ... bench.py --gpus 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29547", BENCH, "--gpus", "2",
                        "--selftest-dist", "ok"], capture_output=True, text=True, timeout=180, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


def test_failed_e2e_leg_prints_line_and_exits_nonzero():
    """A rank other than 0 raises inside the e2e leg: rank 0 still prints the single line (with e2e.error),
    and the job's exit code is not 0."""
    rc, lines, err = _run(["--gpus", "2", "--selftest-dist", "e2e-fail", "--e2e-timeout", "60"])
    assert rc != 0
    assert len(lines) == 1, (lines, err[-2000:])
    assert "stand-in failure" in lines[0]["e2e"]["error"] and lines[0]["e2e_iters_per_sec"] is None


def test_hung_e2e_leg_times_out_nonzero():
    rc, lines, err = _run(["--gpus", "2", "--selftest-dist", "e2e-hang", "--e2e-timeout", "3"])
    assert rc != 0
    assert len(lines) == 1, (lines, err[-2000:])
    assert "did not finish" in lines[0]["e2e"]["error"]


def test_host_cpu_info():
    h = _mod().host_cpu_info()
    assert h["usable_cpus"] >= 1 and h["logical_cpus"] >= h["usable_cpus"] and isinstance(h["model"], str)

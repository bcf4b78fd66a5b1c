"""bench.py's command-line contract (driver: `python bench.py --gpus N --steps K --warmup W`) and its JSON line.
CPU: the flags parse and the script refuses to run without a GPU (no CPU fallback).  GPU: one short run prints ONE
JSON line with every key the contract names, a roofline object and a cpu_baseline object."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
BENCH = os.path.join(ROOT, "bench.py")


def test_bench_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       cwd=ROOT, timeout=300)
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)
    h = subprocess.run([sys.executable, BENCH, "--help"], capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert h.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--batch", "--precision", "--no-cpu-baseline", "--config", "--stream"):
        assert flag in h.stdout


def test_gpus_n_without_n_gpus_exits_nonzero():
    """`python bench.py --gpus 2` started WITHOUT a launcher must start 2 ranks itself or refuse: on a box with fewer
    than 2 GPUs it exits non-zero with a clear message (it used to run 1 GPU silently and label it n_gpus 1)."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("2+ GPUs present")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       cwd=ROOT, timeout=300, env=env)
    assert r.returncode != 0
    assert "--gpus 2" in r.stderr and "GPU(s) are visible" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]          # no metric line under a wrong label


def test_self_launch_plumbing_two_ranks_gloo():
    """The self-launch path end to end on CPU: plain `python bench.py --gpus 2` (no WORLD_SIZE) re-executes itself
    under torch.distributed.run with 2 ranks on 127.0.0.1; the ranks rendezvous (gloo), gather their maps to rank 0
    with the bench's AsyncGather and rank 0 checks every gathered map.  No inference, no metric line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--batch", "3", "--selftest-plumbing"], capture_output=True,
                       text=True, cwd=ROOT, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    assert lines[0] == {"plumbing": "ok", "n_gpus": 2, "world_size_seen": 2, "self_launched": True}


def test_launcher_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "1"], capture_output=True, text=True, cwd=ROOT,
                       timeout=300, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_bench_json_line_contract():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "4"],
                       capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "verified", "end_to_end"):
        assert k in d, k
    assert d["verified"] is True and 0 <= d["epe_vs_oracle_px"] < 1e-3      # timed batch vs single pair AND vs the CPU oracle
    e2e = d["end_to_end"]
    assert e2e["unit"] == "pairs/s" and e2e["value"] > 0 and e2e["async_single_pair"]["value"] > 0
    assert e2e["h2d_bytes_per_pair"] == 3 * 1280 * 720 and e2e["d2h_bytes_per_pair"] == 4 * 1280 * 720
    assert d["unit"] == "pairs/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 4 * 2 / (d["ms_per_step"] * 2 / 1e3)) < 1e-6 * d["value"] + 1e-9
    roof = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in roof, k
    assert roof["bound"] in ("hbm", "mfma") and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    cpu = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cpu, k
    assert cpu["kind"] in ("port", "reference") and cpu["value"] > 0

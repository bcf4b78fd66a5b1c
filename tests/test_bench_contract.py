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
    for flag in ("--gpus", "--steps", "--warmup", "--batch", "--precision", "--no-cpu-baseline"):
        assert flag in h.stdout


@pytest.mark.gpu
def test_bench_json_line_contract():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "4"],
                       capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
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

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
    d = lines[0]
    assert (d["plumbing"], d["n_gpus"], d["world_size_seen"], d["self_launched"]) == ("ok", 2, 2, True)
    # no RCCL on this box: the job must say so and run on, not die (dist.choose_gather: rccl -> ipc -> gloo)
    assert d["gather"]["requested"] == "rccl" and d["gather"]["mode"] == "ipc" and d["gather"]["fallback"] is True
    assert "FALLBACK from rccl" in d["parallelism"] and d["gather"]["attempts"][0]["ok"] is False
    assert [p["rank"] for p in d["per_rank"]] == [0, 1]
    assert all(p["ms_per_step"] > 0 and p["gather_wait_ms_per_step"] >= 0 for p in d["per_rank"])


def _plumbing(extra, env_extra, ranks=2, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(ranks), "--batch", "3", "--selftest-plumbing"] + extra,
                       capture_output=True, text=True, cwd=ROOT, timeout=timeout, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return lines[0]


def test_a_hung_communicator_ends_in_a_labelled_fallback_not_in_silence():
    """if RCCL init or the first gather hangs, the run must still yield a line.  Rank 1's probe is made
    to hang (SN_BENCH_FAULT=rccl_hang@1); the watchdog (2 s here) expires, ALL ranks agree to leave RCCL, the job runs on
    the ipc gather, says so, and exits cleanly although a thread is still stuck."""
    d = _plumbing([], {"SN_BENCH_FAULT": "rccl_hang@1", "SN_BENCH_WATCHDOG_S": "2"})
    assert d["plumbing"] == "ok"
    assert d["gather"]["mode"] == "ipc" and d["gather"]["fallback"] is True and "FALLBACK from rccl" in d["parallelism"]


def test_fallback_chain_reaches_the_host_gather():
    d = _plumbing([], {"SN_BENCH_FAULT": "ipc_error@1"}, ranks=3)
    assert d["plumbing"] == "ok" and d["n_gpus"] == 3
    assert [a["mode"] for a in d["gather"]["attempts"]] == ["rccl", "ipc", "gloo"]
    assert [a["ok"] for a in d["gather"]["attempts"]] == [False, False, True]
    assert "injected failure" in d["gather"]["attempts"][1]["detail"]
    assert d["gather"]["mode"] == "gloo" and "gloo-gather-via-host" in d["parallelism"] and "FALLBACK" in d["parallelism"]


def test_requested_modes_are_not_labelled_as_fallbacks():
    d = _plumbing(["--gather", "ipc"], {})
    assert d["gather"] == {"requested": "ipc", "mode": "ipc", "fallback": False, "attempts": d["gather"]["attempts"]}
    assert "FALLBACK" not in d["parallelism"] and "ipc-peer-pull" in d["parallelism"]
    d = _plumbing(["--dist-backend", "gloo"], {})
    assert d["gather"]["mode"] == "gloo" and d["gather"]["fallback"] is False and "FALLBACK" not in d["parallelism"]


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

"""The N > 1 path with the REAL engine on a one-GPU box (BASELINE configs[3] / [4] are 8-GPU configurations; the driver's
scaling run is the only place they execute for real).  `bench.py --gpus 2 --dist-backend gloo --device-map 0,0` starts
two ranks that both drive GPU 0: each builds its own engine, infers its own seeded shard, the int32 maps are gathered to
rank 0 (staged through host memory — gloo has no device gather), and rank 0 verifies the gathered maps of rank 1 against
its OWN inference of rank 1's inputs.  A rank / device / shard mix-up that would zero an 8-GPU result fails here."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
BENCH = os.path.join(ROOT, "bench.py")


def _run(extra, timeout=1200, env=None, drop=()):
    skip = ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT") + tuple(drop)
    base = {k: v for k, v in os.environ.items() if k not in skip}
    base.update(env or {})
    env = base
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, BENCH] + extra, capture_output=True, text=True, cwd=ROOT, timeout=timeout, env=env)
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    return r, lines


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_real_engine():
    r, lines = _run(["--gpus", "2", "--dist-backend", "gloo", "--device-map", "0,0", "--batch", "4", "--steps", "2",
                     "--warmup", "1", "--no-cpu-baseline", "--no-end-to-end"])
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1
    d = lines[0]
    assert d["n_gpus"] == 2 and d["world_size_seen"] == 2 and d["self_launched"] is True
    assert d["verified"] is True and "gathered maps of ranks 1..1" in d["verification"]
    assert d["config"]["devices"] == [0, 0] and "gloo" in d["config"]["parallelism"]
    assert d["config"]["pairs_per_gpu_per_step"] == 4 and d["steps"] == 2
    assert d["value"] > 0 and abs(d["value"] - 2 * 4 * 2 / (d["ms_per_step"] * 2 / 1e3)) < 1e-6 * d["value"]


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_peer_pull_gather():
    """`--gather ipc` (dist.PeerPullGather): rank 1 exports its two output buffer sets as HIP IPC handles, rank 0 opens them and
    PULLS rank 1's maps with device-to-device copies on its own copy stream (no collective, no receive kernel); two ranks on
    one GPU can open each other's handles, so the whole path — export, open, ready / free handshake, pulls overlapping the
    next step, flush — runs here with the real engine, and rank 0 verifies the pulled maps against its own inference of rank
    1's inputs bit for bit, after the timed steps AND after the further steps of the long measurement."""
    r, lines = _run(["--gpus", "2", "--dist-backend", "gloo", "--device-map", "0,0", "--gather", "ipc", "--batch", "4", "--steps", "5",
                     "--warmup", "2", "--no-cpu-baseline", "--no-end-to-end"])
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1
    d = lines[0]
    assert d["n_gpus"] == 2 and d["world_size_seen"] == 2
    assert d["verified"] is True and "gathered maps of ranks 1..1" in d["verification"]
    assert "ipc-peer-pull" in d["config"]["parallelism"] and d["config"]["devices"] == [0, 0]
    assert d["value"] > 0


@pytest.mark.gpu
def test_rccl_refusing_the_devices_ends_in_a_labelled_ipc_fallback():
    """Two ranks on ONE GPU with the default --dist-backend nccl: RCCL refuses duplicate devices — a REAL communicator
    failure on this box.  The job must not die without a line: all ranks agree to leave RCCL, the maps
    travel by the ipc gather, the line says so, carries every rank's clocks, and still verifies bit for bit."""
    r, lines = _run(["--gpus", "2", "--device-map", "0,0", "--batch", "4", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                     "--no-end-to-end", "--no-long"], env={"SN_BENCH_WATCHDOG_S": "45"})
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1
    d = lines[0]
    g = d["gather"]
    assert g["requested"] == "rccl" and g["mode"] == "ipc" and g["fallback"] is True and g["attempts"][0]["ok"] is False
    assert "FALLBACK from rccl" in d["config"]["parallelism"] and "ipc-peer-pull" in d["config"]["parallelism"]
    assert d["verified"] is True and "gathered maps of ranks 1..1" in d["verification"]
    pr = d["per_rank"]
    assert [p["rank"] for p in pr] == [0, 1]
    for p in pr:
        assert 0 < p["engine_ms_per_step"] <= p["ms_per_step"] + 1e-6
        assert p["gather_wait_ms_per_step"] >= 0 and p["gather_issue_ms_per_step"] >= 0
    assert d["root_extra_ms_per_step_over_median_rank"] is not None
    assert d["refine_stats"]["same_on_all_ranks"] is True and d["config"]["precision_selected"] == "f16"


@pytest.mark.gpu
def test_a_hung_rccl_probe_with_the_real_engine():
    """The watchdog with the real engine: rank 1's RCCL probe hangs (injected), the deadline (5 s) expires, the job runs on
    the ipc gather and exits although a thread is still stuck in the probe."""
    r, lines = _run(["--gpus", "2", "--device-map", "0,0", "--batch", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                     "--no-end-to-end", "--no-long"], env={"SN_BENCH_FAULT": "rccl_hang@1", "SN_BENCH_WATCHDOG_S": "5"})
    assert r.returncode == 0, r.stderr[-3000:]
    d = lines[0]
    assert d["gather"]["mode"] == "ipc" and d["gather"]["fallback"] is True and d["verified"] is True


@pytest.mark.gpu
def test_three_ranks_ipc_gather_shuts_down_cleanly():
    """Teardown order of the IPC exports (round 5's ipc3.log ended with torch's "Producer process has been terminated before
    all shared CUDA tensors released"): the root drops its views and releases the opened mappings, barrier, only then the
    producers let go; the root itself exports nothing.  Three ranks on one GPU, no warning."""
    r, lines = _run(["--gpus", "3", "--dist-backend", "gloo", "--device-map", "0,0,0", "--gather", "ipc", "--batch", "2",
                     "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-end-to-end", "--no-long"])
    assert r.returncode == 0, r.stderr[-3000:]
    d = lines[0]
    assert d["n_gpus"] == 3 and d["verified"] is True and "gathered maps of ranks 1..2" in d["verification"]
    assert d["gather"]["mode"] == "ipc" and d["gather"]["fallback"] is False
    assert "Producer process has been terminated" not in r.stderr, r.stderr[-3000:]
    assert "CudaIPCTypes" not in r.stderr, r.stderr[-3000:]


@pytest.mark.gpu
def test_timed_workload_batch_64_at_the_metric_size(model_factory, oracle, weights_blob):
    """BASELINE configs[2] — the workload bench.py times: 64 pairs of 1280x720 D=192 through the fp16 path in one call.
    Every pair equals its single-pair inference bit for bit (batching, pieces, tower chunks and streams change nothing)
    and one pair is within the north-star bound of the CPU oracle."""
    import numpy as np
    from hobot_stereonet_amd import api, synth
    w, h, d = 1280, 720, 192
    base = [synth.model_input_i8(w, h, d, 300 + s) for s in range(4)]
    xs = np.stack([np.roll(base[k % 4], 16 * (k // 4), axis=2) for k in range(64)])
    with api.StereoNetHIP(model_factory(w, h, d), precision=api.PREC_F16, max_batch=64) as eng:
        disp, raw = eng.infer(xs)
        for k in (0, 1, 17, 38, 63):
            d1, r1 = eng.infer(xs[k])
            assert (r1 == raw[k]).all() and (d1 == disp[k]).all(), k
    assert (raw >= 0).all() and np.isfinite(disp).all()
    odisp = oracle.forward(weights_blob, xs[17], d)[0]
    assert float(np.abs(disp[17] - odisp).mean()) < 1e-3


@pytest.mark.gpu
def test_stream_mode_contract_c5():
    """BASELINE configs[4]'s mode on one GPU: bench.py --config c5 --stream (sustained host-to-host stream of the
    hierarchical-refinement model at 1242x375 D=256)."""
    r, lines = _run(["--config", "c5", "--stream", "2", "--batch", "8"])
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1
    d = lines[0]
    assert d["config"]["width"] == 1242 and d["config"]["height"] == 375 and d["config"]["dmax"] == 256
    assert d["config"]["refine"] == "multi" and d["config"]["refine_levels"] == 4
    assert d["stream"]["h2d_bytes_per_pair"] == 6 * 1242 * 375 and d["stream"]["d2h_bytes_per_pair"] == 4 * 1242 * 375
    assert d["value"] > 0 and d["timed_seconds"] >= 2.0 and d["unit"] == "pairs/s"
    # the maps the pipeline delivered to host memory were checked: both buffer sets == the synchronous path bit for bit,
    # one frame within the bound of the CPU oracle
    assert d["verified"] is True and "bit-exact" in d["verification"] and "CPU oracle" in d["verification"]
    assert d["epe_vs_oracle_px"] is not None and d["epe_vs_oracle_px"] < 1e-3


@pytest.mark.gpu
def test_stream_mode_verified_c2_nv12():
    """The sustained stream at the metric's shape with FeedImg's side-by-side NV12 ingest: seeded stereo frames, the host
    outputs of both buffer sets equal sn_preprocess_sbs_nv12_batch + sn_infer_batch on the same frames, and frame 0 is within
    the bound of the oracle's own split / pre-process / network."""
    r, lines = _run(["--config", "c2", "--stream", "2", "--batch", "8"])
    assert r.returncode == 0, r.stderr[-3000:]
    d = lines[0]
    assert d["stream"]["h2d_bytes_per_pair"] == 3 * 1280 * 720 and "NV12" in d["stream"]["ingest"]
    assert d["verified"] is True and "sn_preprocess_sbs_nv12_batch" in d["verification"]
    assert d["epe_vs_oracle_px"] is not None and d["epe_vs_oracle_px"] < 1e-3


@pytest.mark.gpu
def test_stream_mode_two_ranks_every_rank_verifies():
    """N = 2 on one GPU: each rank streams its own seeded shard and verifies its own host outputs; `verified` is the AND."""
    r, lines = _run(["--gpus", "2", "--dist-backend", "gloo", "--device-map", "0,0", "--config", "c5", "--stream", "1",
                     "--batch", "4", "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-3000:]
    d = lines[0]
    assert d["n_gpus"] == 2 and d["verified"] is True and "all 2 ranks" in d["verification"]


@pytest.mark.gpu
def test_multi_rank_stream_mode_runs_on_default_priority_streams():
    """bench.py raises the engine's stream priority only where a gather runs beside it.  The sustained-stream mode has none
    (ranks stream independently), and high-priority pipeline streams cost the host-to-host path 30 %: two
    ranks on one GPU must report default-priority streams and the rate they reach with the variable forced off; the gather
    mode reports high-priority streams."""
    base = ["--gpus", "2", "--dist-backend", "gloo", "--device-map", "0,0", "--batch", "4", "--no-cpu-baseline"]
    vals = {}
    for tag, env in (("unset", {}), ("forced_off", {"SN_STREAM_PRIORITY": "0"})):
        r, lines = _run(base + ["--config", "c5", "--stream", "2"], env=env, drop=("SN_STREAM_PRIORITY",))
        assert r.returncode == 0, r.stderr[-3000:]
        d = lines[0]
        assert d["n_gpus"] == 2 and d["verified"] is True
        assert d["config"]["stream_priority_high"] is False, tag
        vals[tag] = d["value"]
    print(f"two ranks on one GPU, stream mode: {vals}")
    assert abs(vals["unset"] - vals["forced_off"]) < 0.10 * vals["forced_off"], vals
    r, lines = _run(base + ["--steps", "2", "--warmup", "1", "--no-end-to-end"], drop=("SN_STREAM_PRIORITY",))
    assert r.returncode == 0, r.stderr[-3000:]
    assert lines[0]["config"]["stream_priority_high"] is True


@pytest.mark.gpu
def test_default_bench_end_to_end_figures_are_verified():
    """The default mode's host-to-host figures (batched stream + the async single-pair path) are checked against the
    synchronous path before they are reported."""
    r, lines = _run(["--batch", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-long"])
    assert r.returncode == 0, r.stderr[-3000:]
    d = lines[0]
    assert d["verified"] is True
    assert d["end_to_end"]["verified"] is True and d["end_to_end"]["async_single_pair"]["verified"] is True


@pytest.mark.gpu
def test_hierarchical_batch_larger_than_a_piece(model_factory, oracle, weights_multi):
    """Hierarchical model at KITTI size, batch > piece: the coarse levels run once per low-resolution piece in chunks of
    rb * 4^level pairs; 21 pairs with piece 8 leave ragged tails at every level."""
    import numpy as np
    from hobot_stereonet_amd import api, synth
    w, h, d = 1242, 375, 256
    base = [synth.model_input_i8(w, h, d, 400 + s) for s in range(3)]
    xs = np.stack([np.roll(base[k % 3], 8 * (k // 3), axis=2) for k in range(21)])
    with api.StereoNetHIP(model_factory(w, h, d, multi=True), precision=api.PREC_F16, max_batch=21, piece=8) as eng:
        disp, raw = eng.infer(xs)
        for k in (0, 7, 8, 15, 20):
            d1, r1 = eng.infer(xs[k])
            assert (r1 == raw[k]).all() and (d1 == disp[k]).all(), k
    odisp = oracle.forward(weights_multi, xs[8], d)[0]
    assert float(np.abs(disp[8] - odisp).mean()) < 1e-3


@pytest.mark.gpu
def test_rccl_gather_path_single_rank():
    """The RCCL form of the gather (backend "nccl", device tensors, async_op) executes for real, with the one rank a
    one-GPU box can host: AsyncGather / gather_to_root / run_sharded on device tensors through RCCL — the code the 8-GPU run
    uses, minus the peers."""
    import socket
    import torch
    import torch.distributed as dist
    from hobot_stereonet_amd import dist as sdist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        dev = torch.device("cuda", 0)
        a = torch.arange(5 * 7 * 9, dtype=torch.int32, device=dev).reshape(5, 7, 9)
        g1, g2 = sdist.AsyncGather(a, dst=0), sdist.AsyncGather(a + 1, dst=0)     # two in flight, waited out of order
        r2, r1 = g2.wait(), g1.wait()
        assert len(r1) == 1 and r1[0].is_cuda and torch.equal(r1[0], a) and torch.equal(r2[0], a + 1)
        out = sdist.run_sharded(5, lambda b, e: a[b:e], dst=0)
        assert torch.equal(out, a)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_emulated_root_ingress_runs_and_verifies():
    """bench.py --emulate-root-ingress G: the gather root's receive side emulated on one GPU (G - 1 shard-sized copies per
    step through sn_dbg_copy_limited on a side stream, overlapping the next step).  The batch must still verify."""
    r, lines = _run(["--batch", "4", "--steps", "3", "--warmup", "1", "--emulate-root-ingress", "4", "--ingress-workgroups", "8",
                     "--no-cpu-baseline", "--no-end-to-end", "--no-long"])
    assert r.returncode == 0, r.stderr[-3000:]
    d = lines[0]
    assert d["verified"] is True and d["n_gpus"] == 1
    assert d["emulated_root_ingress"]["copies_per_step"] == 3 and d["emulated_root_ingress"]["bytes_per_copy"] == 4 * 1280 * 720 * 4


@pytest.mark.gpu
def test_limited_copy_hook_copies():
    import torch
    from hobot_stereonet_amd import api
    lib = api.load_library()
    dev = torch.device("cuda", 0)
    src = torch.arange(1 << 20, dtype=torch.int32, device=dev)
    dst = torch.zeros_like(src)
    st = torch.cuda.Stream(device=dev)
    assert lib.sn_dbg_copy_limited(dst.data_ptr(), src.data_ptr(), src.numel() * 4, 3, st.cuda_stream) == 0
    st.synchronize()
    assert torch.equal(src, dst)
    assert lib.sn_dbg_copy_limited(dst.data_ptr(), src.data_ptr(), 12, 3, st.cuda_stream) != 0      # not a multiple of 16

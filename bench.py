#!/usr/bin/env python3
"""Benchmark of the StereoNet hot path on MI355X — BASELINE.json metric: stereo pairs/s (+ ms/frame)
at 1280x720, D=192.

    python bench.py --gpus N --steps K --warmup W [--batch B] [--no-cpu-baseline]

A step = one pass of the hot path (int8 model input -> int32 wire output + float disparity) over
one batch of B synthetic pairs per GPU, inputs already resident in HBM.  N>1: one process per GPU
(torch.distributed, backend nccl = RCCL), pairs sharded with no data-path collective; the step ends
with the north-star's single exchange, an RCCL gather of the int32 disparity maps to rank 0.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

W, H, D = 1280, 720, 192
MFMA_F32_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: exact-fp32 MFMA peak
MFMA_F16_PEAK_TFLOPS = 2500.0     # dense fp16 MFMA peak
HBM_PEAK_GBS = 8000.0


def cpu_baseline(blob, x_one):
    """The CPU-float oracle (oracle/, a 'port': the reference has no CPU implementation of the network)
    timed on this box's host cores on a bounded sample: 3 passes of one 1280x720 D=192 pair."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    oracle_py.forward(blob, x_one, D)            # warm-up (thread pool, page faults)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        oracle_py.forward(blob, x_one, D)
        ts.append(time.perf_counter() - t0)
    med = sorted(ts)[1]
    return {"value": 1.0 / med, "unit": "pairs/s", "cores": oracle_py.num_threads(), "kind": "port",
            "sample": "3 passes of one 1280x720 D=192 pair through oracle/stereonet_oracle.c (median), 1 warm-up",
            "ms_per_frame": med * 1e3}


def pmc_traffic(kernel_match="k_ref_conv_f16"):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC summary
    (profiles/r*_pmc_traffic.json, made by scripts/pmc_traffic.py from separate rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE passes with the gfx950 2x FETCH correction).  None if no summary matches."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
    except Exception:
        return None, None
    if d.get("dominant_match") != kernel_match or not d.get("dominant_avg_hbm_bytes_per_launch"):
        return None, None
    per_pair = float(d["dominant_avg_hbm_bytes_per_launch"]) / float(d.get("pairs_per_launch", 1))
    return per_pair, os.path.relpath(files[-1], ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="pairs per GPU per step")
    ap.add_argument("--refine-chunk", type=int, default=2)
    ap.add_argument("--precision", choices=["fp32", "f16", "f16x3"], default="f16",
                    help="fp32 = exact-fp32 MFMA everywhere; f16 = fp16 tower + 22-bit split low-res convs; f16x3 = split operands everywhere")
    ap.add_argument("--piece", type=int, default=8, help="pairs per low-res piece of the two-stream pipeline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    from hobot_stereonet_amd import api, synth, weights

    blob = weights.synthetic(0)
    B = args.batch
    tmp = tempfile.mkdtemp(prefix=f"snbench{rank}_")
    model = os.path.join(tmp, "bench.snw")
    weights.save_snw(model, blob, W, H, D)
    prec = {"f16": api.PREC_F16, "f16x3": api.PREC_F16X3, "fp32": api.PREC_FP32}[args.precision]
    eng = api.StereoNetHIP(model, device=local_rank, max_batch=B, refine_chunk=args.refine_chunk, precision=prec,
                           piece=args.piece)

    # synthetic shard for this rank: distinct seeds per pair; a few distinct pairs tiled to B
    uniq = min(B, 4)
    host = np.stack([synth.model_input_i8(W, H, D, rank * 1000 + i) for i in range(uniq)])
    host = np.concatenate([host] * ((B + uniq - 1) // uniq))[:B]
    dev = torch.device("cuda", local_rank)
    x = torch.from_numpy(host).to(dev)
    raws = [torch.empty((B, H, W), dtype=torch.int32, device=dev) for _ in range(2)]   # double buffered: the
    raw = raws[0]                                                                       # gather of step i overlaps
    disp = torch.empty((B, H, W), dtype=torch.float32, device=dev)                      # the compute of step i+1
    from hobot_stereonet_amd import dist as sdist

    pending = [None, None]
    counter = [0]

    def step():
        i = counter[0] & 1
        counter[0] += 1
        if pending[i] is not None:          # the gather that still reads raws[i] (issued two steps ago)
            pending[i].wait()
            pending[i] = None
        st = torch.cuda.current_stream().cuda_stream
        eng.infer_device(B, x.data_ptr(), raws[i].data_ptr(), disp.data_ptr(), st)
        if world > 1:      # the path's one exchange: int32 disparity maps -> rank 0 (RCCL over xGMI)
            pending[i] = sdist.AsyncGather(raws[i], dst=0)

    def sync_all():
        for i in range(2):
            if pending[i] is not None:
                pending[i].wait()
                pending[i] = None
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-stage / dominant-kernel timing with HIP events on the launch stream (untimed extra pass)
    roof = None
    stage = None
    if rank == 0:
        eng.set_profiling(True)
        reps = 3
        acc = None
        for _ in range(reps):
            eng.infer_device(B, x.data_ptr(), raw.data_ptr(), disp.data_ptr(), 0)   # own stream, synchronous
            ms = eng.stage_ms()
            acc = ms if acc is None else {k: acc[k] + ms[k] for k in ms}
        stage = {k: v / reps for k, v in acc.items()}
        eng.set_profiling(False)
        dk = eng.dominant_kernel()
        launch_ms = stage["refine_conv"] / dk["launches"]
        tflops = dk["flops_per_launch"] / (launch_ms * 1e-3) / 1e12
        gbs = dk["bytes_per_launch"] / (launch_ms * 1e-3) / 1e9
        if args.precision == "fp32":     # exact-fp32 MFMA: compute-bound
            roof = {"bound": "mfma", "achieved": tflops, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": tflops / MFMA_F32_PEAK_TFLOPS}
        else:                            # fp16 MFMA at AI ~ 125 FLOP/B: HBM-bound
            roof = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS}
        traffic, traffic_src = (None, None)
        if args.precision == "f16":
            traffic, traffic_src = pmc_traffic()
            if traffic is not None:
                traffic *= args.refine_chunk           # summary is per pair; a launch covers refine_chunk pairs
        roof.update({"traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC: 2*FETCH_SIZE + WRITE_SIZE, KiB->B)",
                     "traffic_source": traffic_src, "kernel": dk["name"], "avg_launch_ms": launch_ms,
                     "launches_per_refine_chunk": dk["launches"], "algorithmic_bytes_per_launch": dk["bytes_per_launch"],
                     "algorithmic_flops_per_launch": dk["flops_per_launch"], "tflops": tflops, "gbytes_per_s": gbs,
                     "mfma_peak_tflops": MFMA_F32_PEAK_TFLOPS if args.precision == "fp32" else MFMA_F16_PEAK_TFLOPS})

    if rank == 0:
        pairs = B * world * args.steps
        value = pairs / elapsed
        out = {
            "metric": "stereo pairs/s at 1280x720 D=192 (ms/frame alongside)", "value": value, "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_frame": elapsed / (B * args.steps) * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "f16": "f16 (refinement MFMA operands; fp32 accumulate; low-res branch f32)",
                      "f16x3": "f16x3 (hi/lo split fp16 operands, 3 MFMAs per product ~ 22-bit; fp32 accumulate)"}[args.precision],
            "data": "synthetic (seeded stereo pairs, seeded random SN-K4 weights)",
            "config": {"workload": f"BASELINE configs[1]/[2] shape: 1280x720 D=192, {B} synthetic pairs/GPU/step resident in HBM",
                       "pairs_per_gpu_per_step": B, "width": W, "height": H, "dmax": D, "precision": args.precision,
                       "refine_chunk": args.refine_chunk, "piece": args.piece, "parallelism": f"shard{world}+rccl-gather" if world > 1 else "1gpu"},
            "gflop_per_pair": eng.flops_per_pair / 1e9,
            "model_tflops": value * eng.flops_per_pair / 1e12 / world,
            "stage_ms_per_step": stage,
            "stage_note": "serialised profiling pass; features/aggregate = first piece of `piece` pairs only, "
                          "refine_conv = the 12 tower launches of the first refine chunk, refine/total = whole batch",
            "roofline": roof,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(blob, host[0])
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

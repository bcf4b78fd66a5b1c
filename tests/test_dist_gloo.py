"""N>1 path on CPU: world_size-2 gloo processes shard a batch of pairs, compute their shard (the oracle
stands in for the GPU engine here — it is the checker, the sharding/gather logic is what is under test) and
gather to rank 0; the gathered maps must be bit-identical to the single-process result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

W, H, D = 64, 48, 32


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, out_path):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import oracle_py
    from hobot_stereonet_amd import dist as sdist, synth, weights
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    blob = weights.synthetic(0)

    def infer_shard(begin, end):
        outs = [oracle_py.forward(blob, synth.model_input_i8(W, H, D, 50 + i), D)[1] for i in range(begin, end)]
        return torch.from_numpy(np.stack(outs)) if outs else torch.empty((0, H, W), dtype=torch.int32)

    got = sdist.run_sharded(n, infer_shard, dst=0)
    if rank == 0:
        np.save(out_path, got.numpy())
    else:
        assert got is None
    # the bench's non-blocking form (equal shards): two gathers in flight, waited out of order
    a = torch.full((2, 3), rank * 10 + 1, dtype=torch.int32)
    b = torch.full((2, 3), rank * 10 + 2, dtype=torch.int32)
    ga, gb = sdist.AsyncGather(a, dst=0), sdist.AsyncGather(b, dst=0)
    rb, ra = gb.wait(), ga.wait()
    if rank == 0:
        assert [int(t[0, 0]) for t in ra] == [r * 10 + 1 for r in range(world)]
        assert [int(t[0, 0]) for t in rb] == [r * 10 + 2 for r in range(world)]
    else:
        assert ra is None and rb is None
    # the root's receive lists allocated once and reused step after step (bench.py's two buffer sets)
    bufs = [sdist.AsyncGather.alloc_root_buffers(a, dst=0) for _ in range(2)]
    assert (bufs[0] is None) == (rank != 0)
    for step in range(4):
        src = torch.full((2, 3), rank * 100 + step, dtype=torch.int32)
        got_s = sdist.AsyncGather(src, dst=0, bufs=bufs[step & 1]).wait()
        if rank == 0:
            assert got_s is bufs[step & 1] and [int(t[1, 2]) for t in got_s] == [r * 100 + step for r in range(world)]
    if rank == 0:
        with pytest.raises(ValueError):
            sdist.AsyncGather(a, dst=0, bufs=bufs[0][:1])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [4, 5, 1])
def test_sharded_gather_matches_single_process(tmp_path, oracle, weights_blob, n):
    from hobot_stereonet_amd import synth
    out = str(tmp_path / "g.npy")
    mp.spawn(_worker, args=(2, _free_port(), n, out), nprocs=2, join=True)
    got = np.load(out)
    exp = np.stack([oracle.forward(weights_blob, synth.model_input_i8(W, H, D, 50 + i), D)[1] for i in range(n)])
    assert got.shape == exp.shape and (got == exp).all()


def test_shard_ranges_cover_everything():
    from hobot_stereonet_amd.dist import shard_counts, shard_range
    for n in (0, 1, 7, 8, 64, 513):
        for world in (1, 2, 3, 8):
            ranges = [shard_range(n, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            assert max(shard_counts(n, world)) - min(shard_counts(n, world)) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _pull_worker(rank, world, port, steps):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from hobot_stereonet_amd import dist as sdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = sdist.PeerPullGather((3, 4, 5), torch.int32, "cpu", dst=0)
    seen = {}
    for k in range(steps):
        i = g.begin()
        assert i == k % 2
        if rank == 0 and k >= 2:
            # begin(k) has released step k-2: its pull is complete, and nothing newer has been pulled into that set yet
            got = g.result(i)
            assert [int(t[0, 0, 0]) for t in got[1:]] == [r * 1000 + (k - 2) for r in range(1, world)]
            seen[k - 2] = True
        g.local[i].fill_(rank * 1000 + k)           # "the engine writes this step's maps"
        g.end()
    g.flush()
    if rank == 0:
        for s in range(min(2, steps)):
            last = max(k for k in range(steps) if k % 2 == s)
            got = g.result(s)
            assert len(got) == world and got[0] is g.local[s]
            assert [int(t[2, 3, 4]) for t in got] == [r * 1000 + last for r in range(world)]
        assert len(seen) == max(0, steps - 2)
    else:
        assert g.result(0) is None
    g.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,steps", [(2, 5), (3, 4), (2, 1)])
def test_peer_pull_gather_protocol(world, steps):
    """PeerPullGather (the zero-CU gather: the root pulls every peer's exported buffer sets) on CPU tensors shared through
    /dev/shm: double buffering, the ready / free handshake, flush, result order."""
    mp.spawn(_pull_worker, args=(world, _free_port(), steps), nprocs=world, join=True)

"""Multi-GPU data-parallel execution of the hot path: independent stereo pairs are sharded across ranks
(one process per GPU), there is NO collective on the data path, and the only exchange is the final gather
of the int32 disparity maps to rank 0 (RCCL over xGMI with backend "nccl", gloo on CPU in the tests).

The reference has no multi-device path; the unit of work here is what it already treats
as independent: one frame = one Run() (stereonet_infer/src/stereonet_node.cpp:144,812).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [begin, end) of n pairs for `rank`; the first n % world ranks get one extra pair."""
    if n < 0 or world <= 0 or not (0 <= rank < world):
        raise ValueError("bad shard arguments")
    q, r = divmod(n, world)
    begin = rank * q + min(rank, r)
    return begin, begin + q + (1 if rank < r else 0)


def shard_counts(n: int, world: int) -> List[int]:
    return [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]


def gather_to_root(local: torch.Tensor, counts: List[int], dst: int = 0) -> Optional[torch.Tensor]:
    """Gathers per-rank [count_r, ...] tensors to `dst`, returning the [sum(counts), ...] tensor there
    (None elsewhere).  Ragged shards are padded to the largest shard for the collective and trimmed."""
    world = dist.get_world_size()
    rank = dist.get_rank()
    if len(counts) != world or local.shape[0] != counts[rank]:
        raise ValueError("counts do not match the local shard")
    cmax = max(counts)
    send = local
    if local.shape[0] != cmax:
        send = torch.zeros((cmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        send[:local.shape[0]] = local
    send = send.contiguous()
    bufs = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)


class AsyncGather:
    """gather_to_root for equal shards without blocking the compute stream: the collective is enqueued with
    async_op=True (RCCL runs it on its own stream behind an event on the current stream), so the next batch's
    kernels overlap the transfer over xGMI.  Call wait() before reusing the source tensor.

    gloo has no gather for device tensors: a device-resident shard is staged through host memory (a synchronous copy,
    then the same asynchronous gather on the host copies) and rank `dst` gets its buffers back on the shard's device.
    That form exists so that the whole multi-rank path — real engine, real maps — can run where RCCL cannot, e.g. two
    ranks on ONE GPU in the tests; it is not a throughput path."""

    def __init__(self, local: torch.Tensor, dst: int = 0, bufs: Optional[List[torch.Tensor]] = None):
        """bufs: rank `dst`'s receive list (world tensors shaped like `local`), allocated ONCE by a caller that gathers
        every step (alloc_root_buffers) — world x 236 MB at the metric's shard size is not something to allocate per
        step; None allocates a fresh list (one-shot callers).

        Aliasing contract of `bufs`: wait() returns the caller's OWN list, so the result of one gather is overwritten by
        the next gather into the same list — nothing orders that write behind consumers of the previous result.  A caller
        that keeps results across steps gives every gather in flight its own list (bench.py: one per buffer set) and is
        done with a list's contents before it starts the next gather into it.
        Staged mode (gloo with device tensors) gathers HOST copies: device-resident `bufs` cannot receive them, so passing
        them raises instead of being dropped silently; host-resident `bufs` (pinned or not) are used as the receive list
        and wait() returns fresh device copies of them."""
        world, rank = dist.get_world_size(), dist.get_rank()
        self.device = local.device
        self.staged = local.is_cuda and dist.get_backend() == "gloo"
        send = local.contiguous()
        if self.staged:
            send = send.cpu()                 # waits for the producing stream
        if rank != dst:
            self.bufs = None
        elif bufs is not None:
            if len(bufs) != world or any(b.shape != send.shape or b.dtype != send.dtype or b.device != send.device for b in bufs):
                raise ValueError("preallocated gather buffers do not match the shard" +
                                 (" (staged gloo gather of device tensors: the receive list must live in host memory)" if self.staged else ""))
            self.bufs = bufs
        else:
            self.bufs = [torch.empty_like(send) for _ in range(world)]
        self.work = dist.gather(send, self.bufs, dst=dst, async_op=True)

    @staticmethod
    def alloc_root_buffers(local: torch.Tensor, dst: int = 0) -> Optional[List[torch.Tensor]]:
        """The receive list of one gather on rank `dst` (None on the other ranks): on the shard's device, or in host
        memory when the gather will be staged (gloo with device tensors)."""
        if dist.get_rank() != dst:
            return None
        staged = local.is_cuda and dist.get_backend() == "gloo"
        dev = torch.device("cpu") if staged else local.device
        return [torch.empty(local.shape, dtype=local.dtype, device=dev) for _ in range(dist.get_world_size())]

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
            if self.staged and self.bufs is not None:
                self.bufs = [b.to(self.device) for b in self.bufs]
        return self.bufs


def run_sharded(n: int, infer_shard: Callable[[int, int], torch.Tensor], dst: int = 0) -> Optional[torch.Tensor]:
    """infer_shard(begin, end) -> tensor [end-begin, H, W] for this rank's pairs; returns all n maps on dst."""
    world, rank = dist.get_world_size(), dist.get_rank()
    begin, end = shard_range(n, rank, world)
    local = infer_shard(begin, end)
    return gather_to_root(local, shard_counts(n, world), dst)

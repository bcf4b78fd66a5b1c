"""Multi-GPU data-parallel execution of the hot path: independent stereo pairs are sharded across ranks
(one process per GPU), there is NO collective on the data path, and the only exchange is the final gather
of the int32 disparity maps to rank 0 (RCCL over xGMI with backend "nccl", gloo on CPU in the tests).

The reference has no multi-device path; the unit of work here is what it already treats
as independent: one frame = one Run() (stereonet_infer/src/stereonet_node.cpp:144,812).
"""
from __future__ import annotations

import os
import threading
import time
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

    def __init__(self, local: torch.Tensor, dst: int = 0, bufs: Optional[List[torch.Tensor]] = None, group=None):
        """group: the process group of the exchange (None = the default group; bench.py keeps a gloo default group as its
        control plane and passes the RCCL group chosen by choose_gather).
        bufs: rank `dst`'s receive list (world tensors shaped like `local`), allocated ONCE by a caller that gathers
        every step (alloc_root_buffers) — world x 236 MB at the metric's shard size is not something to allocate per
        step; None allocates a fresh list (one-shot callers).

        Aliasing contract of `bufs`: wait() returns the caller's OWN list, so the result of one gather is overwritten by
        the next gather into the same list — nothing orders that write behind consumers of the previous result.  A caller
        that keeps results across steps gives every gather in flight its own list (bench.py: one per buffer set) and is
        done with a list's contents before it starts the next gather into it.
        Staged mode (gloo with device tensors) gathers HOST copies: device-resident `bufs` cannot receive them, so passing
        them raises instead of being dropped silently; host-resident `bufs` (pinned or not) are used as the receive list
        and wait() returns fresh device copies of them."""
        world, rank = dist.get_world_size(group), dist.get_rank()
        self.device = local.device
        self.staged = local.is_cuda and dist.get_backend(group) == "gloo"
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
        self.work = dist.gather(send, self.bufs, dst=dst, async_op=True, group=group)

    @staticmethod
    def alloc_root_buffers(local: torch.Tensor, dst: int = 0, group=None) -> Optional[List[torch.Tensor]]:
        """The receive list of one gather on rank `dst` (None on the other ranks): on the shard's device, or in host
        memory when the gather will be staged (gloo with device tensors)."""
        if dist.get_rank() != dst:
            return None
        staged = local.is_cuda and dist.get_backend(group) == "gloo"
        dev = torch.device("cpu") if staged else local.device
        return [torch.empty(local.shape, dtype=local.dtype, device=dev) for _ in range(dist.get_world_size(group))]

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



class PeerPullGather:
    """The gather with NOTHING running on the root's compute units: every rank exports its two
    output buffer sets ONCE (HIP IPC memory handles), and the root PULLS each peer's maps with plain device-to-device
    copies on a copy stream of its own — between devices those are copy-engine (SDMA) transfers over the peer's xGMI
    link: no receive kernel, no workgroups taken from the towers.  `AsyncGather` over RCCL stays the default (north_star:
    "RCCL over xGMI only for the final gather"); this is the A/B form (`bench.py --gather ipc`): round 4's emulated root
    ingress measured RCCL-style receive kernels at -9 % for the root at G = 8.

    Two ranks on ONE GPU can open each other's handles, so the whole path runs on a one-GPU box (`--device-map 0,0`);
    CPU tensors take the same protocol through files in /dev/shm (world-size-2 gloo test, tests/test_dist_gloo.py).

    The caller writes step k's maps into `self.local[k % sets]` between begin() and end():
        i = g.begin()                       # -> buffer set of this step
        ... enqueue the step that writes g.local[i] on the current stream ...
        g.end()
        ...
        g.flush()                           # after the last step; then, on the root:
        maps = g.result(i)                  # [rank 0's set i, rank 1's, ...] as of the last step that used set i
    Protocol (tiny control messages on a gloo group; no device collective at all):
      peer   begin(k): wait for the root's "free k - sets" (its pull of the step that last wrote this set has completed)
             end(k):   record an event behind step k; host-wait for step k-1's event, tell the root "ready k-1"
      root   begin(k): host-wait for its pulls of step k - sets, tell every peer "free k - sets"
             end(k):   take "ready k-1" from every peer, enqueue the pulls of step k-1's set on the copy stream
    so a rank's host runs at most one step ahead of its device and the pulls of step k-1 overlap the compute of step k."""

    def __init__(self, shape, dtype, device, dst: int = 0, sets: int = 2):
        if sets < 2:
            # with one set the root's begin(k) would release step k-1 before it has taken "ready k-1" or enqueued that pull
            raise ValueError("PeerPullGather needs at least two buffer sets")
        self.world, self.rank, self.dst = dist.get_world_size(), dist.get_rank(), dst
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.sets = sets
        self.shape, self.dtype = tuple(int(d) for d in shape), dtype
        self.ctl = dist.new_group(backend="gloo")            # control plane: eight bytes per step and peer
        self._paths: List[str] = []
        self.is_root = self.rank == dst
        self.local = [self._alloc(s) for s in range(sets)]
        # Only the PEERS export: an exported allocation is reference counted by torch's CUDA-IPC plumbing and every export
        # expects one consumer; the root's own buffers are read first-hand, and a handle of them that nobody opens kept
        # the root's allocation in torch's limbo until the "Producer process has been terminated before all shared CUDA
        # tensors released" warning at exit (gpurun_out/ipc3.log, round 5).
        # A rank that cannot export / open reports it instead of raising on its own: the constructor is collective, and a
        # lone exception would leave the other ranks in the barrier below (choose_gather falls back on the error).
        handles, err = None, ""
        if not self.is_root:
            try:
                if _fault("ipc_error", self.rank):
                    raise RuntimeError("injected failure (SN_BENCH_FAULT=ipc_error)")
                handles = [self._export(s) for s in range(sets)]
            except Exception as e:                           # noqa: BLE001 (reported to every rank below)
                err = f"rank {self.rank} could not export its buffers: {e!r}"
        table = [None] * self.world if self.is_root else None
        dist.gather_object((handles, err), table, dst=dst, group=self.ctl)
        self.peer_views = None
        self.recv = None
        self.copy_stream = None
        verdict = [""]
        if self.is_root:
            try:
                bad = [e for r, (_, e) in enumerate(table) if r != dst and e]
                if bad:
                    raise RuntimeError("; ".join(bad))
                if _fault("ipc_error", self.rank):
                    raise RuntimeError("injected failure (SN_BENCH_FAULT=ipc_error)")
                self.peer_views = [[self._open(h) for h in hs] if r != dst else None for r, (hs, _) in enumerate(table)]
                self.recv = [[torch.empty(self.shape, dtype=dtype, device=self.device) if r != dst else None
                              for r in range(self.world)] for _ in range(sets)]
                if self.cuda:
                    self.copy_stream = torch.cuda.Stream(device=self.device)
            except Exception as e:                           # noqa: BLE001
                verdict[0] = f"{e!r}"
                self.peer_views = None
        dist.broadcast_object_list(verdict, src=dst, group=self.ctl)
        if verdict[0]:
            self._teardown()
            raise RuntimeError(f"peer-pull gather cannot be set up: {verdict[0]}")
        self.pull_ev = [torch.cuda.Event() if self.cuda else None for _ in range(sets)]
        self.step_ev = [torch.cuda.Event() if self.cuda else None for _ in range(sets)]
        self.k = 0                   # steps begun
        self.ended = 0               # steps ended
        self.announced = 0           # peer: "ready" messages sent / root: steps whose pulls are enqueued
        self.released = 0            # root: "free" messages sent / peer: received
        dist.barrier(group=self.ctl)

    # ---- buffers and handles -------------------------------------------------------------------------------------
    def _numel(self):
        n = 1
        for d in self.shape:
            n *= d
        return n

    def _alloc(self, s):
        if self.cuda:
            return torch.empty(self.shape, dtype=self.dtype, device=self.device)
        import os
        path = f"/dev/shm/sn_peerpull_{os.getpid()}_{id(self) & 0xffffff:x}_{s}"
        n = self._numel()
        t = torch.from_file(path, shared=True, size=max(n, 1), dtype=self.dtype)[:n].view(self.shape)
        self._paths.append(path)
        return t

    def _export(self, s):
        if self.cuda:
            from torch.multiprocessing.reductions import reduce_tensor
            fn, args = reduce_tensor(self.local[s])          # hipIpcGetMemHandle behind torch's CUDA-IPC plumbing
            return ("cuda", fn, args)
        return ("file", self._paths[s])

    def _open(self, h):
        if h[0] == "cuda":
            return h[1](*h[2])                               # hipIpcOpenMemHandle: a tensor on the PEER's device
        n = self._numel()
        return torch.from_file(h[1], shared=True, size=max(n, 1), dtype=self.dtype)[:n].view(self.shape)

    # ---- control messages ----------------------------------------------------------------------------------------
    def _send(self, value, to):
        dist.send(torch.tensor([value], dtype=torch.int64), dst=to, group=self.ctl)

    def _recv(self, frm):
        t = torch.zeros(1, dtype=torch.int64)
        dist.recv(t, src=frm, group=self.ctl)
        return int(t[0])

    def _peers(self):
        return [r for r in range(self.world) if r != self.dst]

    # ---- root side -----------------------------------------------------------------------------------------------
    def _release_through(self, step):           # root: announce steps [released, step] free once their pulls are done
        while self.released <= step:
            s = self.released % self.sets
            if self.cuda:
                self.pull_ev[s].synchronize()
            for r in self._peers():
                self._send(self.released, r)
            self.released += 1

    def _pull_through(self, step):              # root: enqueue the pulls of steps [announced, step]
        while self.announced <= step:
            s = self.announced % self.sets
            for r in self._peers():
                got = self._recv(r)
                if got != self.announced:
                    raise RuntimeError(f"peer {r} announced step {got}, expected {self.announced}")
            if self.cuda:
                with torch.cuda.stream(self.copy_stream):
                    for r in self._peers():
                        self.recv[s][r].copy_(self.peer_views[r][s], non_blocking=True)
                    self.pull_ev[s].record(self.copy_stream)
            else:
                for r in self._peers():
                    self.recv[s][r].copy_(self.peer_views[r][s])
            self.announced += 1

    # ---- peer side -----------------------------------------------------------------------------------------------
    def _announce_through(self, step):          # peer: tell the root steps [announced, step] have finished on the device
        while self.announced <= step:
            s = self.announced % self.sets
            if self.cuda:
                self.step_ev[s].synchronize()
            self._send(self.announced, self.dst)
            self.announced += 1

    # ---- the per-step calls --------------------------------------------------------------------------------------
    def begin(self) -> int:
        k = self.k
        if self.k != self.ended:
            raise RuntimeError("begin() without end()")
        if k >= self.sets:
            if self.is_root:
                self._release_through(k - self.sets)
            else:
                while self.released <= k - self.sets:
                    got = self._recv(self.dst)
                    if got != self.released:
                        raise RuntimeError(f"root released step {got}, expected {self.released}")
                    self.released += 1
        self.k += 1
        return k % self.sets

    def end(self) -> None:
        k = self.ended
        if self.k != k + 1:
            raise RuntimeError("end() without begin()")
        if self.cuda:
            self.step_ev[k % self.sets].record(torch.cuda.current_stream(self.device))
        self.ended += 1
        if k >= 1:
            if self.is_root:
                self._pull_through(k - 1)
            else:
                self._announce_through(k - 1)

    def flush(self) -> None:
        """After the last step: every announced / pulled / released counter catches up; the root's copy stream is idle."""
        last = self.ended - 1
        if last < 0:
            return
        if self.is_root:
            self._pull_through(last)
            self._release_through(last)
        else:
            self._announce_through(last)
            while self.released <= last:
                got = self._recv(self.dst)
                if got != self.released:
                    raise RuntimeError(f"root released step {got}, expected {self.released}")
                self.released += 1

    def result(self, s: int) -> Optional[List[torch.Tensor]]:
        """Root, after flush(): the maps of the last step that used set s, rank order (its own set first-hand)."""
        if not self.is_root:
            return None
        if self.cuda:
            self.pull_ev[s].synchronize()
        return [self.local[s] if r == self.dst else self.recv[s][r] for r in range(self.world)]

    def close(self) -> None:
        """Teardown in the order the exported allocations need: (1) the root drops every view of the peers' buffers and
        has torch release the opened IPC mappings, (2) barrier, (3) only then the producers let go of the exported
        buffers, (4) barrier, control group destroyed."""
        import gc
        self.peer_views = None
        self.recv = None
        gc.collect()                          # the IPC mappings go with the last reference to the opened tensors
        if self.cuda:
            torch.cuda.synchronize(self.device)
            torch.cuda.ipc_collect()          # torch caches opened IPC allocations: release them before the producers exit
        dist.barrier(group=self.ctl)          # nobody unlinks / frees while a peer may still map it
        self._teardown()

    def _teardown(self) -> None:
        import gc
        self.local = None                     # the caller's own references (bench.py: raws) go first
        gc.collect()
        if self.cuda:
            torch.cuda.synchronize(self.device)
            torch.cuda.ipc_collect()          # producer side: blocks whose consumers are gone leave torch's limbo
        for p in self._paths:
            try:
                os.unlink(p)
            except OSError:
                pass
        self._paths = []
        if self.ctl is not None:
            dist.barrier(group=self.ctl)
            dist.destroy_process_group(self.ctl)
            self.ctl = None


def _fault(name: str, rank: int) -> bool:
    """SN_BENCH_FAULT=name[@rank][,name...]: fault injection for the gather set-up (tests, and a rehearsal of the first
    real multi-GPU run): rccl_hang, rccl_error, ipc_error."""
    for item in os.environ.get("SN_BENCH_FAULT", "").split(","):
        item = item.strip()
        if not item:
            continue
        n, _, r = item.partition("@")
        if n == name and (not r or int(r) == rank):
            return True
    return False


class GatherPlan:
    """How the maps reach rank `dst` in this job, and how that was decided.
        mode      "rccl" (AsyncGather on `group`, an RCCL process group), "ipc" (`pull`, a PeerPullGather) or "gloo"
                  (AsyncGather on the default gloo group, staged through host memory)
        attempts  one record per mode tried: {"mode", "ok", "seconds", "detail"}; the first ok one is in force
        hung      a probe thread is still stuck in a collective: the process must leave through os._exit"""

    def __init__(self, requested):
        self.requested, self.mode, self.group, self.pull, self.attempts, self.hung = requested, None, None, None, [], False

    @property
    def fallback(self) -> bool:
        return self.mode != self.requested

    def label(self, world: int) -> str:
        base = {"rccl": f"shard{world}+rccl-gather",
                "ipc": f"shard{world}+ipc-peer-pull-gather (root pulls the peers' exported buffers, copy stream; control messages over gloo)",
                "gloo": f"shard{world}+gloo-gather-via-host (functional check, not a scaling figure)"}[self.mode]
        if self.fallback:
            why = "; ".join(f"{a['mode']}: {a['detail']}" for a in self.attempts if not a["ok"])
            base += f" [FALLBACK from {self.requested}: {why}]"
        return base


def _agree(ok: bool) -> bool:
    """True only if every rank says so (control plane = the default gloo group, CPU tensors)."""
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item())


def _probe_rccl(device, dst: int, timeout_s: float):
    """Creates an RCCL group and runs ONE small gather on it in a helper thread, with a deadline: communicator creation
    and the first collective are where a multi-GPU job hangs or fails (no 8-GPU node has been available to any round,
    so the first real run must not be able to end without a line).  -> (group or None, detail, hung)"""
    from datetime import timedelta
    world, rank = dist.get_world_size(), dist.get_rank()
    box = {"ok": False, "err": "", "group": None}

    def run():
        try:
            if _fault("rccl_hang", rank):
                time.sleep(10 ** 6)
            if _fault("rccl_error", rank):
                raise RuntimeError("injected failure (SN_BENCH_FAULT=rccl_error)")
            t = torch.full((4,), rank + 1, dtype=torch.int32, device=device)
            bufs = [torch.empty_like(t) for _ in range(world)] if rank == dst else None
            dist.gather(t, bufs, dst=dst, group=box["group"])
            if torch.device(device).type == "cuda":
                torch.cuda.synchronize(device)
            if rank == dst and any(int(b[0]) != r + 1 for r, b in enumerate(bufs)):
                raise RuntimeError("the probe gather returned wrong values")
            box["ok"] = True
        except Exception as e:                               # noqa: BLE001
            box["err"] = repr(e)[:300]

    try:
        # (a long collective timeout: torch's own watchdog must not tear the process down while the fallback runs)
        box["group"] = dist.new_group(backend="nccl", timeout=timedelta(minutes=60))
    except Exception as e:                                   # noqa: BLE001
        if not _fault("rccl_hang", rank):                    # (an injected hang is played out below even where RCCL cannot start)
            return None, f"communicator group: {repr(e)[:300]}", False
    th = threading.Thread(target=run, daemon=True, name="sn-rccl-probe")
    th.start()
    th.join(timeout_s)
    if th.is_alive():
        return None, f"no answer from the first gather within {timeout_s:.0f} s", True
    if not box["ok"]:
        return None, box["err"] or "failed", False
    return box["group"], "ok", False


def choose_gather(requested: str, device, shape, dtype, dst: int = 0, timeout_s: float = 60.0) -> GatherPlan:
    """Sets up the job's one exchange (gather of the maps to `dst`), falling back instead of dying: rccl -> ipc -> gloo.
    Collective: every rank calls it with the same arguments; the DEFAULT process group must be gloo (control plane).  Every
    mode is tried under a deadline and accepted only if ALL ranks succeeded; the plan records what was tried and why it was
    left (bench.py prints it in config.parallelism and `gather`)."""
    if dist.get_backend() != "gloo":
        raise RuntimeError("choose_gather wants a gloo default process group as its control plane")
    order = {"rccl": ["rccl", "ipc", "gloo"], "ipc": ["ipc", "gloo"], "gloo": ["gloo"]}[requested]
    plan = GatherPlan(requested)
    for mode in order:
        t0 = time.perf_counter()
        ok, detail = False, ""
        if mode == "rccl":
            group, detail, hung = _probe_rccl(device, dst, timeout_s)
            plan.hung = plan.hung or hung
            ok = group is not None
            if _agree(ok):
                plan.group = group
            elif ok:
                ok, detail = False, "another rank's probe failed"
        elif mode == "ipc":
            try:
                plan.pull = PeerPullGather(shape, dtype, device, dst=dst)
                ok, detail = True, "ok"
            except Exception as e:                           # noqa: BLE001 (PeerPullGather fails on every rank together)
                plan.pull, detail = None, repr(e)[:300]
            if not _agree(ok) and ok:
                ok, detail = False, "another rank could not set it up"
                plan.pull.close()
                plan.pull = None
        else:
            ok, detail = _agree(True), "ok"
        plan.attempts.append({"mode": mode, "ok": ok, "seconds": round(time.perf_counter() - t0, 3), "detail": detail})
        if ok:
            plan.mode = mode
            return plan
    raise RuntimeError("no gather mode could be set up")

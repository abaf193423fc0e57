"""Batch / multi-GPU driver: independent images, one process per GPU, no data-path collective.

The reference has no multi-GPU mode (its gpu_id argument only prints device properties,
seamlessClone_imp.cu:243-250).  Each clone is independent, so a batch shards
as image i -> rank i mod world; the only cross-rank traffic is the timing barrier and the
max-over-ranks of the elapsed time, which run over torch.distributed (gloo: CPU scalars only,
nothing on xGMI).  torch is imported lazily and only when world_size > 1.
"""
from __future__ import annotations

import os
import time


def shard_indices(n_items: int, rank: int, world: int) -> list[int]:
    """Round-robin ownership: image i belongs to rank i % world (ragged tails allowed)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    return list(range(rank, n_items, world))


def parse_cpulist(text: str) -> list[int]:
    """'0-15,64-79' (the kernel's cpulist format) -> sorted core numbers."""
    out = []
    for part in text.replace("\n", "").split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-", 1)
            out.extend(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return sorted(set(out))


def rank_cores(local_rank: int, world: int, allowed: list[int], gpu_local_cpulist: str | None = None,
               ranks_sharing: int | None = None) -> list[int]:
    """Host cores of one rank when `world` ranks (one per GPU) share a node.  `allowed`: the cores this process may use now.
    With the kernel's list of cores local to the rank's GPU (`gpu_local_cpulist`, /sys/bus/pci/devices/<bdf>/local_cpulist):
    the allowed cores among them, split evenly among the ranks that share that list -- on an 8-GPU node the GPUs come four
    per socket, so by default world / 2 ranks share one (`ranks_sharing`), rank r being number r % ranks_sharing among them.
    Without it, or when it names none of the allowed cores: an even contiguous split of `allowed` (core numbers are
    socket-major on Linux, so rank r of 8 still lands on socket r // 4).  Never returns an empty list unless `allowed` is."""
    allowed = sorted(set(allowed))
    if world <= 1 or not allowed:
        return allowed
    if not (0 <= local_rank < world):
        raise ValueError(f"bad local rank {local_rank}/{world}")
    pool, k, n = allowed, local_rank, world
    if gpu_local_cpulist:
        local = [c for c in parse_cpulist(gpu_local_cpulist) if c in set(allowed)]
        if local and len(local) < len(allowed):
            n = ranks_sharing if ranks_sharing else max(1, world // max(1, round(len(allowed) / len(local))))
            pool, k = local, local_rank % n
    per = len(pool) // n
    if per == 0:                      # more ranks than cores: share
        return [pool[k % len(pool)]]
    return pool[k * per:(k + 1) * per]


def _spawn_once(world: int, cmd: list[str], extra_env: dict | None, poll_s: float):
    import socket
    import subprocess
    import tempfile
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    errs = [tempfile.TemporaryFile(mode="w+") for _ in range(world)]     # every rank's stderr, kept for the failure report
    # rank 0's stdout goes to a temporary FILE, not a pipe: a pipe is only read after every rank has exited, so a rank 0 that
    # prints more than the pipe buffer (64 KiB: a long JSON line, library warnings) would block in write() while the others
    # wait for it in a barrier
    with tempfile.TemporaryFile(mode="w+") as out0:
        for r in range(world):
            env = dict(os.environ)
            env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1",
                        "MASTER_PORT": str(port)})
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if extra_env:
                env.update(extra_env)
            procs.append(subprocess.Popen(cmd, env=env, stdout=out0 if r == 0 else subprocess.DEVNULL, stderr=errs[r], text=True))
        rc = 0
        first_failed = None
        pending = set(range(world))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    first_failed = r
                    for q in pending:                      # a rank died: the rest would hang in the next barrier
                        procs[q].terminate()
            if pending:
                time.sleep(poll_s)
        out0.seek(0)
        report = {"rc": rc, "first_failed_rank": first_failed, "codes": [p.returncode for p in procs], "stderr": []}
        for e in errs:
            e.seek(0)
            report["stderr"].append(e.read()[-4000:])
            e.close()
        return rc, out0.read(), report


# what a lost rendezvous looks like in a rank's stderr (gloo / c10d TCP store): the only failures worth a second launch
# Only the stderr of the rank that failed FIRST is classified: when a rank dies of anything else (a GPU fault, an import
# error) the survivors' gloo errors ("timed out", "Connection reset", "TCPStore ...") would match generic marks and repeat a
# faulting GPU run.  The marks are the ones a taken port or a store that was never reachable produces.
_RENDEZVOUS_MARKS = ("address already in use", "EADDRINUSE", "Connection refused", "connection refused")


def _rendezvous_failure(report: dict) -> bool:
    r = report.get("first_failed_rank")
    if r is None or not (0 <= r < len(report["stderr"])):
        return False
    return any(m in report["stderr"][r] for m in _RENDEZVOUS_MARKS)


def spawn_ranks(world: int, cmd: list[str], extra_env: dict | None = None, poll_s: float = 0.05, retries: int = 1):
    """Start `world` fresh processes running `cmd`, one per rank, with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
    MASTER_PORT set (what torch.distributed.run would export), and wait for them.  Returns (rc, rank-0 stdout): rc is 0
    only if every rank exited 0; when one fails the others are terminated (they would wait in a barrier for ever).
    The rendezvous port is found by binding port 0 and closing the socket, so another process can take it before the
    ranks bind it: a launch that fails before rank 0 has printed anything AND whose ranks' stderr shows a rendezvous error
    (port taken, connection refused, store time-out) is repeated (`retries` times) on a new port.  Any other failure (a GPU
    fault, an import error, an out-of-memory kill) is NOT retried.  Either way every failed attempt is reported on this
    process's stderr: exit code of every rank, which one failed first, the tail of each rank's stderr -- a first failure
    never disappears behind a retry.  `spawn_ranks.last_report` keeps the last attempt's report.
    The caller must not have initialised the GPU -- nothing in here does: the children are plain subprocesses started
    with Popen (fork + exec of a process that never loaded the HIP library), never an exec of an initialised process."""
    import sys
    attempt = 0
    while True:
        rc, out0, report = _spawn_once(world, cmd, extra_env, poll_s)
        spawn_ranks.last_report = report
        if rc == 0:
            return rc, out0
        attempt += 1
        retry = retries > 0 and not out0.strip() and _rendezvous_failure(report)
        sys.stderr.write(f"[spawn_ranks] attempt {attempt}: rank {report['first_failed_rank']} failed first, exit codes {report['codes']}"
                         f"{' -- rendezvous error, launching again on a new port' if retry else ''}\n")
        for r, e in enumerate(report["stderr"]):
            if e.strip():
                sys.stderr.write(f"[spawn_ranks] ---- rank {r} stderr (tail) ----\n{e.rstrip()}\n")
        sys.stderr.flush()
        if not retry:
            return rc, out0
        retries -= 1


spawn_ranks.last_report = None


class Comm:
    """Barrier + max-reduction over ranks.  world == 1 needs no torch at all."""

    def __init__(self, rank: int | None = None, world: int | None = None, backend: str = "gloo"):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else rank
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
        self.local_rank = int(os.environ.get("LOCAL_RANK", str(self.rank)))
        self._dist = None
        if self.world > 1:
            import torch.distributed as dist
            if not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29531")
                dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world)
            self._dist = dist

    def barrier(self):
        if self._dist is not None:
            self._dist.barrier()

    def max(self, value: float) -> float:
        if self._dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())

    def sum(self, value: float) -> float:
        if self._dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return float(t.item())

    def gather(self, value: float) -> list[float]:
        """Every rank's value, in rank order, on every rank (CPU scalars over gloo)."""
        if self._dist is None:
            return [float(value)]
        import torch
        t = torch.zeros(self.world, dtype=torch.float64)
        t[self.rank] = float(value)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return [float(x) for x in t.tolist()]

    def close(self):
        if self._dist is not None and self._dist.is_initialized():
            self._dist.destroy_process_group()
            self._dist = None


def timed_region(comm: Comm, sync, body, own: list | None = None):
    """barrier + device sync | body() | device sync + barrier; returns max-over-ranks seconds.  `own` (a list) receives this
    rank's own elapsed seconds, so that a scaling run can show imbalance between the ranks beside the maximum."""
    comm.barrier()
    sync()
    t0 = time.perf_counter()
    body()
    sync()
    dt = time.perf_counter() - t0
    comm.barrier()
    if own is not None:
        own.append(dt)
    return comm.max(dt)


class StreamPool:
    """K library instances on one GPU (one HIP stream each) driven by K host threads.

    Clones are independent, and a single clone leaves the GPU idle during its latency-bound
    phases (coarse multigrid levels, the bounding-box read-back), so running several at once
    raises throughput on small and medium ROIs.  ctypes releases the GIL around every C call, so
    plain Python threads are enough; a C++ host does the same with one instance per std::thread
    (INTEGRATION.md)."""

    def __init__(self, gpu_id: int = 0, streams: int = 4, **solver):
        from concurrent.futures import ThreadPoolExecutor
        from . import capi
        self.instances = [capi.Instance(gpu_id) for _ in range(max(1, streams))]
        for inst in self.instances:
            if solver:
                inst.set_solver(**solver)
        self._pool = ThreadPoolExecutor(max_workers=len(self.instances))

    def map(self, fn, items):
        """fn(instance, item) for every item; item i runs on instance i % K.  Returns results in order."""
        items = list(items)
        k = len(self.instances)

        def lane(j):
            inst = self.instances[j]
            return [(i, fn(inst, items[i])) for i in range(j, len(items), k)]

        out = [None] * len(items)
        for part in self._pool.map(lane, range(k)):
            for i, r in part:
                out[i] = r
        return out

    def sync(self):
        for inst in self.instances:
            inst.sync()

    def close(self):
        self._pool.shutdown(wait=True)
        for inst in self.instances:
            inst.destroy()
        self.instances = []


def run_batch(instance, items, clone_one):
    """Run clone_one(instance, item) for every item this rank owns; returns units processed."""
    n = 0
    for it in items:
        clone_one(instance, it)
        n += 1
    return n

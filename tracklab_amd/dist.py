"""Stream-parallel multi-GPU plumbing: one process per GPU, streams partitioned round-robin, NO data-path
collective. The only communication is (a) the barrier / max-time reduction around a timed region and (b) one
per-epoch SUM all-reduce of the HOTA sufficient statistics (tracklab_amd.hota.pack, ~1.1 KB, latency-bound).
Backend "nccl" is RCCL over xGMI on ROCm; "gloo" covers the same code path on CPU in the tests.
"""
from __future__ import annotations

import os

import numpy as np
import torch


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend: str | None = None):
    """Initialise torch.distributed from the torchrun environment (no-op for a single process)."""
    world, rank, local_rank = env_world()
    if world == 1 and os.environ.get("TLK_FORCE_DIST") != "1":
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)
    return dist


def init_single(backend: str | None = None):
    """A process group of ONE rank without a torchrun environment (bench.py's N = 1 line): the barrier, the max / sum reductions and the
    all-gather of the per-rank rates then run on the job's real backend (RCCL on the GPU box) instead of being skipped, so the collective path
    the multi-GPU job depends on has executed at least once on hardware.  Rendezvous on 127.0.0.1 and a free port."""
    import socket
    import torch.distributed as dist
    if dist.is_initialized():
        return dist
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    kw = {}
    if backend == "nccl":
        torch.cuda.set_device(0)
        kw["device_id"] = torch.device("cuda", 0)
    dist.init_process_group(backend, init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, **kw)
    return dist


def streams_for_rank(n_streams_total: int, rank: int, world: int) -> list[int]:
    """stream s -> rank s mod world: whole videos, never frames, are sharded."""
    return [s for s in range(n_streams_total) if s % world == rank]


def allreduce_sum(vec: np.ndarray, dist=None, device=None) -> np.ndarray:
    if dist is None:
        return np.asarray(vec, dtype=np.float64)
    t = torch.as_tensor(np.asarray(vec, dtype=np.float64), device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def allreduce_max(value: float, dist=None, device=None) -> float:
    if dist is None:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def even_cpu_slice(local_rank: int, local_world: int, cpus=None) -> list[int]:
    """Host CPUs of one rank when the GPU's NUMA node is not known (/sys without the PCI tree): the allowed CPUs split evenly, in order, among
    the node's ranks -- disjoint slices, so that 8 ranks' loader / pinned-memory threads do not all roam over every core."""
    cpus = sorted(os.sched_getaffinity(0)) if cpus is None else sorted(cpus)
    if local_world <= 1 or len(cpus) < local_world:
        return cpus
    per = len(cpus) // local_world
    return cpus[local_rank * per:(local_rank + 1) * per]


def numa_cpu_slice(node_cpus, peers_on_node: list[int], me: int, allowed=None) -> list[int]:
    """CPUs of rank `me` among the ranks `peers_on_node` (sorted device indices) that share one NUMA node's `node_cpus`."""
    allowed = sorted(set(node_cpus) & (set(allowed) if allowed is not None else os.sched_getaffinity(0)))
    if not allowed or me not in peers_on_node or len(peers_on_node) <= 1:
        return allowed
    per = max(1, len(allowed) // len(peers_on_node))
    k = peers_on_node.index(me)
    return allowed[k * per:(k + 1) * per] or allowed


class serialized:
    """Cross-process mutual exclusion on one node (an flock'ed file): ``with serialized("tune"):`` lets the ranks of a job pass ONE AT A TIME.
    bench.py wraps each rank's warm-up in it when world > 1 -- eight concurrent hipGraph captures, hipBLASLt / MIOpen tuning passes and pinned
    allocations on one host were never run together on hardware; serial warm-up costs seconds and removes the question.
    The lock file lives in ``TLK_LOCK_DIR`` (default: the system temp directory) and is keyed by the user id and the job's MASTER_PORT, so that
    two jobs on one host do not wait for each other and a file another user left behind is never opened.  A lock file that
    cannot be opened or locked (read-only temp directory, foreign owner) does not cost the job its run: the section then runs unlocked and
    ``locked`` is False.  ``order`` (a list) receives (event, time) pairs for the tests."""

    def __init__(self, name: str, order=None):
        import tempfile
        key = os.environ.get("MASTER_PORT", "solo")
        uid = os.getuid() if hasattr(os, "getuid") else 0
        self.path = os.path.join(os.environ.get("TLK_LOCK_DIR", tempfile.gettempdir()), f"tlk_{name}_{uid}_{key}.lock")
        self.order = order
        self.locked = False
        self._f = None

    def __enter__(self):
        import fcntl
        import time
        try:
            self._f = open(self.path, "a+")
            fcntl.flock(self._f, fcntl.LOCK_EX)
            self.locked = True
        except OSError:
            if self._f is not None:
                self._f.close()
            self._f = None
        if self.order is not None:
            self.order.append(("enter", time.time()))
        return self

    def __exit__(self, *exc):
        import fcntl
        import time
        if self.order is not None:
            self.order.append(("exit", time.time()))
        if self._f is not None:
            try:
                fcntl.flock(self._f, fcntl.LOCK_UN)
            finally:
                self._f.close()
                self._f = None
        self.locked = False
        return False


def placement_is_sound(placement, local_world: int, host_cpus: int) -> bool:
    """rank_placement rows [rank, numa node, cpus]: every rank of a multi-rank job is pinned to a non-empty CPU set, and the sets of one node can
    be disjoint (their sizes sum to at most the host's CPUs).  bench.py reports it; a False on hardware means ranks roam over each other's cores."""
    if local_world <= 1:
        return True
    if any(p[2] <= 0 for p in placement):
        return False
    return sum(p[2] for p in placement) <= max(host_cpus, 1) * max(1, len(placement) // max(local_world, 1))

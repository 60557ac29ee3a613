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

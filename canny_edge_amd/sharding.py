"""Host logic for running the hot path on several GPUs of one node.

Frames are independent (the reference processes one frame at a time, src/main.cpp:120-137), so the
multi-GPU path is a partition, not a collective: rank r of N takes the contiguous frame range
``shard_range(n_frames, r, N)`` and nothing crosses xGMI.  ``torch.distributed`` is used only for the
barrier and the MAX-over-ranks of the timed region that ``bench.py`` reports.
"""
from __future__ import annotations

import os
from typing import Tuple

from . import capi


def rank_env() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) as torch.distributed.run exports them."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def shard_range(n_frames: int, rank: int, world: int) -> Tuple[int, int]:
    """Frame range [begin, end) of shard ``rank``; identical to what canny_hip_canny_multi_gpu uses."""
    return capi.shard_range(n_frames, rank, world)


def max_over_ranks(seconds: float, device=None) -> float:
    """MAX of a per-rank duration over the process group (identity when not initialised)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def aggregate_throughput(units_per_rank: int, world: int, seconds_max: float) -> float:
    """Whole-job units per second: every rank processed ``units_per_rank`` in ``seconds_max``."""
    return world * units_per_rank / seconds_max

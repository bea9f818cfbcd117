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


# ---- per-rank reporting of the host->host legs (bench.py, N ranks) -------------------------------------------------
def gather_rows(row: dict) -> list:
    """One row per rank, in rank order, on every rank (identity list when the process group is not initialised)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [row]
    rows = [None] * dist.get_world_size()
    dist.all_gather_object(rows, row)
    return rows


def numa_nodes_of(arr):
    """NUMA node(s) the pages of a host array live on: {node: resident pages}, summed over every mapping of
    /proc/self/maps that overlaps the array (the kernel splits a big allocation into several mappings when parts of
    it get different flags) with the per-node counts of /proc/self/numa_maps; huge pages count as one page each.
    None if unknown."""
    try:
        lo_a, hi_a = arr.ctypes.data, arr.ctypes.data + max(1, arr.nbytes)
        starts = set()
        with open("/proc/self/maps") as f:
            for line in f:
                lo, _, hi = line.split()[0].partition("-")
                if int(lo, 16) < hi_a and int(hi, 16) > lo_a:
                    starts.add(int(lo, 16))
        if not starts:
            return None
        nodes = {}
        with open("/proc/self/numa_maps") as f:
            for line in f:
                parts = line.split()
                if int(parts[0], 16) not in starts:
                    continue
                for tok in parts[1:]:
                    k, _, v = tok[1:].partition("=")
                    if tok.startswith("N") and k.isdigit() and v.isdigit():
                        nodes[int(k)] = nodes.get(int(k), 0) + int(v)
        return nodes or None
    except (OSError, ValueError, IndexError):
        return None


def h2h_rank_row(rank: int, pixels: int, out_bytes: int, seconds: float, src=None, out=None) -> dict:
    """This rank's row of a host->host batch: its own median call (not the MAX), its link rates in each direction
    (one byte per pixel goes up, out_bytes come down) and where its pinned buffers live."""
    return {"rank": rank, "ms_per_batch": round(seconds * 1e3, 3), "Mpixels_per_s": round(pixels / seconds / 1e6, 1),
            "h2d_GBps": round(pixels / seconds / 1e9, 2), "d2h_GBps": round(out_bytes / seconds / 1e9, 2),
            "pinned_src_numa_pages": numa_nodes_of(src) if src is not None else None,
            "pinned_out_numa_pages": numa_nodes_of(out) if out is not None else None,
            "cpus_bound": len(os.sched_getaffinity(0))}


def config5_result(frames_per_round: int, rounds: int, pixels_per_round: int, rank: int, world: int,
                   seconds_own: float, seconds_max: float) -> dict:
    """BASELINE config 5's per-GPU share through the host pipeline: `rounds` batches of `frames_per_round` frames per
    GPU re-using the same pinned buffers; weak scaling (every rank moves the same number of frames), aggregate over
    the MAX-over-ranks time, one row per rank beside it."""
    return {"what": f"{rounds} x {frames_per_round} = {rounds * frames_per_round} 4K frames per GPU, pinned host u8 in -> "
                    "pinned host s16 maps out, buffers re-used per round (BASELINE.json configs[4]: 1024 per GPU)",
            "frames_per_gpu": rounds * frames_per_round,
            "value": round(aggregate_throughput(pixels_per_round * rounds, world, seconds_max) / 1e6, 1),
            "unit": "Mpixels/s", "seconds": round(seconds_max, 4),
            "per_rank": gather_rows({"rank": rank, "seconds": round(seconds_own, 4),
                                     "Mpixels_per_s": round(pixels_per_round * rounds / seconds_own / 1e6, 1)})}

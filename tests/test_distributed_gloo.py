"""N>1 path on the CPU: gloo ranks (2 and 3 of them, even, uneven and empty shards) shard a batch with the product's sharding rules, each rank runs
its shard (here through the oracle, since there is no GPU in the build container), and rank 0 checks
that the gathered result equals the single-process result and that the timing reduction is a MAX."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from canny_edge_amd import sharding
from canny_edge_amd.synth import synth_frame

H, W = 48, 64


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_q, N_FRAMES):
    import oracle
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert sharding.rank_env() == (rank, world, rank)
        frames = np.stack([synth_frame(H, W, 100 + i) for i in range(N_FRAMES)])
        b, e = sharding.shard_range(N_FRAMES, rank, world)
        mine = np.stack([oracle.canny(f, 1.0, 50, 150) for f in frames[b:e]]) if e > b else \
            np.empty((0, H, W), np.int16)
        dist.barrier()
        slow = sharding.max_over_ranks(1.0 + rank)            # rank 1 pretends to be slower
        padded = np.zeros((N_FRAMES, H, W), np.int16)
        padded[b:e] = mine
        t = torch.from_numpy(padded.astype(np.int32))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)               # disjoint shards: SUM == concatenation
        if rank == 0:
            want = np.stack([oracle.canny(f, 1.0, 50, 150) for f in frames])
            out_q.put((bool(np.array_equal(t.numpy().astype(np.int16), want)), slow, (b, e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_frames,world,range0", [(5, 2, (0, 3)), (7, 3, (0, 3)), (2, 3, (0, 1))],
                         ids=["5_frames_2_ranks", "7_frames_3_ranks", "2_frames_3_ranks_one_empty_shard"])
def test_rank_sharding_and_max_timing(n_frames, world, range0):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, n_frames)) for r in range(world)]
    for p in procs:
        p.start()
    ok, slow, rng0 = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok, "gathered shards differ from the single-process result"
    assert slow == float(world), "timing must be the MAX over ranks"
    assert rng0 == range0
    # every frame belongs to exactly one rank, ranges are contiguous and ordered, sizes differ by at most one
    ranges = [sharding.shard_range(n_frames, r, world) for r in range(world)]
    assert ranges[0][0] == 0 and ranges[-1][1] == n_frames
    assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    sizes = [e - b for b, e in ranges]
    assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    assert sharding.aggregate_throughput(10, 2, 4.0) == 5.0


def _report_worker(rank, world, port, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        src = np.zeros((2, H, W), np.uint8)
        out = np.zeros((2, H, W), np.int16)
        seconds = 0.010 * (1 + rank)                          # rank 1 pretends to be slower
        rows = sharding.gather_rows(sharding.h2h_rank_row(rank, src.size, out.nbytes, seconds, src, out))
        t_max = sharding.max_over_ranks(seconds)
        c5 = sharding.config5_result(frames_per_round=2, rounds=8, pixels_per_round=src.size, rank=rank, world=world,
                                     seconds_own=8 * seconds, seconds_max=8 * t_max)
        if rank == 0:
            out_q.put((rows, c5, t_max))
    finally:
        dist.destroy_process_group()


def test_per_rank_rows_and_config5_at_world_2():
    """The N-rank bench line (bench.py host_to_host): one row per rank in rank order with that rank's OWN time, link
    rates and buffer placement, and config 5 (1024 frames per GPU as 8 rounds) aggregated over the MAX time."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_report_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    rows, c5, t_max = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    px = 2 * H * W
    assert [r["rank"] for r in rows] == [0, 1]
    for r in rows:
        assert set(r) >= {"rank", "ms_per_batch", "Mpixels_per_s", "h2d_GBps", "d2h_GBps", "pinned_src_numa_pages",
                          "pinned_out_numa_pages", "cpus_bound"}
        assert r["ms_per_batch"] == 10.0 * (1 + r["rank"])            # its own time, not the MAX
        assert r["d2h_GBps"] == round(2 * r["h2d_GBps"], 2) or abs(r["d2h_GBps"] - 2 * r["h2d_GBps"]) < 0.02
        assert r["cpus_bound"] >= 1
        assert r["pinned_src_numa_pages"] is None or all(v > 0 for v in r["pinned_src_numa_pages"].values())
    assert t_max == 0.020
    assert c5["frames_per_gpu"] == 16 and c5["unit"] == "Mpixels/s"
    assert [r["rank"] for r in c5["per_rank"]] == [0, 1]
    assert c5["seconds"] == round(8 * t_max, 4)
    assert c5["value"] == round(world * 8 * px / (8 * t_max) / 1e6, 1)    # aggregate over the MAX-over-ranks time
    assert c5["per_rank"][1]["seconds"] == 2 * c5["per_rank"][0]["seconds"]
    # single process: the helpers degrade to one row
    solo = sharding.gather_rows({"rank": 0})
    assert solo == [{"rank": 0}]

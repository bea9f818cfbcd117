#!/usr/bin/env python3
"""Headline benchmark: Canny throughput on 4K gray frames + HBM roofline of the fused Sobel+NMS pass
(BASELINE.json `metric`).

Which number is which (round 3; the two readings of "end-to-end" are both in the line, each under its own name):
  value        -- DEVICE-RESIDENT whole-pipeline rate: frames already in HBM when the timed region starts, edge maps
                  left in HBM.  That is what this bench's contract defines `value` to be ("inputs already resident in
                  HBM when the timed region starts ... the PCIe-inclusive rate is never `value`"), and the `metric`
                  string says so.
  end_to_end   -- SURVEY.md 8(d) Metric 1, the reference's own notion (src/cuda.cu:83-101, src/utils.cpp:435,479):
                  host u8 frame in -> host s16 edge map out, H2D and D2H inside the timed region, same run.  The
                  details (u8 / bit maps, per-rank rows, NUMA placement, config 5) are in `host_to_host`.

    python bench.py --gpus N --steps K --warmup W

One process per GPU (the driver launches N>1 through torch.distributed.run).  A *step* is one pass of
the whole hot path (gaussian -> fused Sobel+NMS -> hysteresis) over this rank's batch of F resident
4K frames; frames are independent, so ranks share nothing and the only collective is the timing
MAX / barrier (weak scaling: F frames per GPU whatever N is).

Workload (BASELINE.json configs[1]): 3840x2160 gray, sigma 1.4, thresholds 50/150, synthetic frames
(canny_edge_amd.synth, seed 42+i), F = 128 frames per GPU per step: the working set (1 GB u8 in, 2 GB s16 per
plane) is far beyond the 256 MB Infinity Cache and a step (2.4 ms) is long against its fixed costs -- kernel
tails, sweep launches, one host round trip: 64 frames per step run 7 % slower per frame, 256 another 3 % faster.
Untimed before the W warm-up steps: a parity spot check against the oracle and --spinup-seconds of steps that bring
the device to its steady clocks.

The JSON line also carries
  roofline     -- the Sobel+NMS kernel of the timed region at SURVEY.md 8(d)'s 4 B/px (`frac`; `frac_own_bytes` prices
                  the 4.25 B/px it really moves), with `copy_probe`: a plain device copy of the same 2 + 2 B/px timed
                  in the same run -- what a 1:1 read/write stream reaches on THIS box (boxes differ by up to 14 %).
                  Algorithmic bytes / average launch time measured
                  with HIP events inside the timed region (attached to the kernel's dispatch on the launch stream,
                  every 4th step: see the comment at the timed loop), vs 8 TB/s.  canny() runs it with the
                  hysteresis threshold-classify step inside (2 B/px s16 in + 2 B/px s16 provisional edge map +
                  0.25 B/px bit-planes out); `roofline_sobel_nms_s16` is the stage-API form SURVEY.md 8(d) prices at 4 B/px (s16 in,
                  s16 out), timed the same way on the same batch right after the timed region;
                  `roofline_other_kernels` prices the other kernels of the step the same way
  host_to_host -- SURVEY.md 8(d) Metric 1 as the reference's GPU path pays it (src/cuda.cu:83-101: every frame
                  crosses PCIe both ways): --h2h-frames 4K frames from pinned host memory through
                  canny_hip_canny_batch (s16 maps, the reference's plane type) and canny_hip_canny_batch_u8, wall
                  time of the median call including H2D and D2H, GB/s per direction against the PCIe 5 x16 link (63 GB/s spec), a single
                  frame's latency, the pageable-buffer rate (five calls listed), per-rank rows with the NUMA node
                  of each rank's pinned buffers, and `config5`: 1024 frames per GPU as 8 x 128 through the same
                  pinned buffers (BASELINE config 5's per-GPU share).
  cpu_baseline -- the CPU oracle (a faithful single-thread restatement of the reference's utils.cpp;
                  the reference itself cannot be compiled here) timed on a bounded sample, rank 0, N=1
torch is used only for device memory, the stream and torch.distributed.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
PCIE_PEAK_GBS = 63.0   # PCIe Gen5 x16 per direction, spec (MI355X_MICROARCH.md, "Host link")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=128, help="4K frames resident per GPU per step")
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--sigma", type=float, default=1.4)
    ap.add_argument("--min-val", type=int, default=50)
    ap.add_argument("--max-val", type=int, default=150)
    ap.add_argument("--cpu-frames", type=int, default=32,
                    help="frames the CPU baseline times (rank 0, N=1): ~0.37 s each, i.e. ~12 s by default")
    ap.add_argument("--fuse-classify", type=int, default=1, choices=(0, 1),
                    help="0: canny() runs Sobel+NMS and the hysteresis classify pass as separate kernels (A/B)")
    ap.add_argument("--stream", type=int, default=0, choices=(0, 1, 2),
                    help="1: steps go through canny_hip_dev_canny_stream (the host learns about step i's convergence "
                         "after it has queued the Gaussian of step i+1; two alternating output buffers); 2: the same "
                         "with the sweeps on a second stream beside that Gaussian; 0: plain canny_hip_dev_canny calls")
    ap.add_argument("--spinup-seconds", type=float, default=0.6,
                    help="untimed steps run for this long before the W warm-up steps: the device reaches its steady "
                         "clocks only after a few tenths of a second of load (3 warm-up steps = 7 ms: 2.46 ms per "
                         "step; 200: 2.40)")
    ap.add_argument("--h2h-frames", type=int, default=128,
                    help="frames per host-to-host batch (pinned host u8 in -> pinned host edge maps out); 0 skips it")
    ap.add_argument("--h2h-reps", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="skip the pre-timing parity spot check")
    return ap.parse_args()


def cpu_baseline(frames, sigma, lo, hi, n_sample):
    """Time the CPU oracle on a bounded sample of the same workload (single thread)."""
    import oracle
    n_sample = max(1, n_sample)
    t = 0.0
    px = 0
    stage = {"gaussian": 0.0, "sobel": 0.0, "nms": 0.0, "hysteresis": 0.0}
    for i in range(n_sample):  # the batch cycles through len(frames) distinct frames, and so does the sample
        f = frames[i % len(frames)]
        r = oracle.canny(f, sigma, lo, hi, stages=True)
        t += r["seconds"]["total"]
        for k in stage:
            stage[k] += r["seconds"][k]
        px += f.size
    return {
        "value": round(px / t / 1e6, 3), "unit": "Mpixels/s", "cores": 1, "kind": "port",
        "sample": f"{n_sample} of the benchmark's 4K frames, 4 stages timed like src/utils.cpp:435-479, "
                  f"gcc -O2 -ffp-contract=off, host has {os.cpu_count()} logical cores",
        "seconds": round(t, 3),
        "stage_share": {k: round(v / t, 3) for k, v in stage.items()},
    }


def host_to_host(ctx, np, base_np, args, rank, world, sync_max, check):
    """SURVEY.md 8(d) Metric 1: host u8 frames in -> host edge maps out, H2D and D2H inside the timed region
    (one-time setup -- context, pipelines, pinned buffers -- outside, as the metric defines it)."""
    from canny_edge_amd import sharding
    H, W, n = args.height, args.width, args.h2h_frames
    px = n * H * W
    compact = ctx.get_option("tune_batch_compact") == 0  # s16 / u8 maps cross the link as bit maps (round 3 default)
    res = {"frames_per_gpu": n, "height": H, "width": W, "sigma": args.sigma, "reps": args.h2h_reps,
           "pcie_peak_GBps_per_direction": PCIE_PEAK_GBS,
           "transfer": ("compact: the finished s16 / u8 map is packed to 1 bit per pixel on the device, the bits cross "
                        "PCIe (1/16 of the s16 map) and host threads write the caller's plane from them"
                        if compact else "the map itself is downloaded"),
           "what": "canny_hip_canny_batch / _u8 / _bits on pinned host buffers: 3-stream chunk pipeline "
                   "(upload | kernels | download), wall time of the median of --h2h-reps calls, MAX over ranks; "
                   "s16 = the reference's short maps, u8 = the same 0/255 as bytes, bits = 1 bit per pixel"}
    if True:  # the bench's own context: its stream is idle here, and every extra stream costs a hardware queue
        src = ctx.pinned_array((n, H, W), np.uint8)
        for i in range(n):
            src[i] = base_np[i % len(base_np)]
        outs = {"s16": ctx.pinned_array((n, H, W), np.int16), "u8": ctx.pinned_array((n, H, W), np.uint8),
                "bits": ctx.pinned_array((n, H, (W + 7) // 8), np.uint8)}
        for name, out in outs.items():
            u8, bits = name == "u8", name == "bits"
            out[...] = 1
            for _ in range(2):  # builds pipelines and staging; lets the link clocks settle after the compute phase
                ctx.canny_batch(src, args.sigma, args.min_val, args.max_val, out=out, u8=u8, bits=bits)
            sync_max(0.0)
            calls = []
            for _ in range(args.h2h_reps):  # every call timed; the figure is the MEDIAN call (MAX over ranks)
                t0 = time.perf_counter()
                ctx.canny_batch(src, args.sigma, args.min_val, args.max_val, out=out, u8=u8, bits=bits)
                calls.append(time.perf_counter() - t0)
            t_own = sorted(calls)[len(calls) // 2]
            t = sync_max(t_own)
            wire_out = n * H * ((W + 7) // 8) if (compact or bits) else out.nbytes  # bytes that cross the link downwards
            out_bytes_per_px = wire_out / px
            h2d, d2h = px / t / 1e9, wire_out / t / 1e9
            # one row per rank (not only the MAX): this rank's median call, its link rates, where its buffers live
            per_rank = sharding.gather_rows(sharding.h2h_rank_row(rank, px, wire_out, t_own, src, out))
            res[name] = {"value": round(px * world / t / 1e6, 1), "unit": "Mpixels/s", "ms_per_batch": round(t * 1e3, 3),
                         "ms_per_call": [round(c * 1e3, 2) for c in calls], "per_rank": per_rank,
                         "h2d_GBps_per_gpu": round(h2d, 2), "d2h_GBps_per_gpu": round(d2h, 2),
                         "link_frac": round(max(h2d, d2h) / PCIE_PEAK_GBS, 4),
                         "host_plane_write_GBps_per_gpu": round(out.nbytes / t / 1e9, 2),
                         "bytes_over_link_per_px": round(1 + out_bytes_per_px, 4)}
            # one frame at a time (latency): pinned in, pinned out
            one_in, one_out = src[:1], out[:1]
            ctx.canny_batch(one_in, args.sigma, args.min_val, args.max_val, out=one_out, u8=u8, bits=bits)
            t0 = time.perf_counter()
            for _ in range(10):
                ctx.canny_batch(one_in, args.sigma, args.min_val, args.max_val, out=one_out, u8=u8, bits=bits)
            res[name]["single_frame_ms"] = round((time.perf_counter() - t0) / 10 * 1e3, 4)
        if check:
            import oracle
            ok = True
            for i in (0, n - 1):
                want = oracle.canny(base_np[i % len(base_np)], args.sigma, args.min_val, args.max_val)
                ok = ok and bool(np.array_equal(outs["s16"][i], want)) and \
                    bool(np.array_equal(outs["u8"][i], want.astype(np.uint8))) and \
                    bool(np.array_equal(outs["bits"][i], np.packbits(want != 0, axis=-1)))
            res["parity_checked"] = ok
            if not ok:
                raise SystemExit("bench.py: host-to-host edge maps differ from the oracle -- refusing to report")
        # BASELINE config 5's per-GPU share: 1024 4K frames per GPU through the host pipeline, as 8 x n frames re-using
        # the pinned buffers (eight ranks then need 8 x 3.2 GB of pinned memory, not 8 x 25 GB).  Weak scaling: every
        # rank moves 8 * n frames whatever the world size; the aggregate uses the MAX-over-ranks time.
        rounds = max(1, 1024 // n) if n >= 16 else 1
        out16 = outs["s16"]
        t0 = time.perf_counter()
        for _ in range(rounds):
            ctx.canny_batch(src, args.sigma, args.min_val, args.max_val, out=out16)
        t_own = time.perf_counter() - t0
        t = sync_max(t_own)
        res["config5"] = sharding.config5_result(n, rounds, px, rank, world, t_own, t)
        # pageable caller buffers (what a caller that never heard of pinned memory gets), s16: five calls, all listed
        # (one call of a fresh process measured anything between 10.7 and 19.4 Gpix/s).  The arrays
        # are allocated AND first touched after this process bound itself to the GPU's local CPUs (main()), so they
        # and the library's staging threads -- which inherit the caller's mask -- share a NUMA node.
        pg_in = np.array(src[: max(1, n // 2)])
        pg_out = np.zeros(pg_in.shape, np.int16)  # zeros: touched here, not page-faulted inside the timed call
        ctx.canny_batch(pg_in, args.sigma, args.min_val, args.max_val, out=pg_out)
        pg_calls = []
        for _ in range(5):
            t0 = time.perf_counter()
            ctx.canny_batch(pg_in, args.sigma, args.min_val, args.max_val, out=pg_out)
            pg_calls.append(time.perf_counter() - t0)
        t = sync_max(sorted(pg_calls)[2])
        res["s16_pageable"] = {"value": round(pg_in.size * world / t / 1e6, 1), "unit": "Mpixels/s",
                               "frames_per_gpu": int(pg_in.shape[0]), "ms_per_batch": round(t * 1e3, 3),
                               "ms_per_call": [round(c * 1e3, 2) for c in pg_calls],
                               "numa_pages_in": sharding.numa_nodes_of(pg_in),
                               "numa_pages_out": sharding.numa_nodes_of(pg_out)}
        # the same caller-allocated buffers page-locked once with canny_hip_host_register (what a maintainer of the
        # reference would do with its new[] frames): the pinned path
        t0 = time.perf_counter()
        ctx.host_register(pg_in)
        ctx.host_register(pg_out)
        t_reg = time.perf_counter() - t0
        try:
            ctx.canny_batch(pg_in, args.sigma, args.min_val, args.max_val, out=pg_out)
            t0 = time.perf_counter()
            ctx.canny_batch(pg_in, args.sigma, args.min_val, args.max_val, out=pg_out)
            t = sync_max(time.perf_counter() - t0)
        finally:
            ctx.host_unregister(pg_in)
            ctx.host_unregister(pg_out)
        res["s16_registered"] = {"value": round(pg_in.size * world / t / 1e6, 1), "unit": "Mpixels/s",
                                 "frames_per_gpu": int(pg_in.shape[0]), "ms_per_batch": round(t * 1e3, 3),
                                 "register_ms_once": round(t_reg * 1e3, 2)}
    res["GPU_MAX_HW_QUEUES"] = os.environ.get("GPU_MAX_HW_QUEUES")
    res["cpus_of_this_process"] = len(os.sched_getaffinity(0))
    return res


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))   # same triple as sharding.rank_env()
    if world != args.gpus and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)

    # HIP multiplexes a process's streams onto 4 hardware queues by default; torch brings streams of its own, and the
    # host->host pipeline needs its upload, compute and download streams on separate queues (include/canny_hip.h)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import numpy as np
    import torch  # first: its HIP runtime is the one the C-ABI library then binds to
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the Canny hot path has no CPU fallback")
    # CANNY_BENCH_REHEARSE=1: a dry run of the N-rank code path on a box with fewer GPUs than ranks -- ranks share the
    # cards round-robin and rendezvous over gloo (RCCL refuses two ranks on one device).  Its numbers mean nothing and the
    # line says so ("rehearsal": true); the driver never sets it.
    rehearse = os.environ.get("CANNY_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = None if rehearse else dev  # where the MAX-over-ranks tensor lives
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from canny_edge_amd import capi, sharding
    from canny_edge_amd.synth import synth_frame

    # one process per GPU: run (and first-touch host buffers) on the CPUs local to this rank's GPU -- a buffer on the other
    # socket costs the host->host pipeline a third to a half of the PCIe rate
    local_cpus = capi.device_local_cpus(local_rank)
    if local_cpus:
        try:
            cpus = set()
            for part in local_cpus.split(","):
                a, _, b = part.partition("-")
                cpus.update(range(int(a), int(b or a) + 1))
            os.sched_setaffinity(0, cpus & os.sched_getaffinity(0) or os.sched_getaffinity(0))
        except (OSError, ValueError):
            pass

    H, W, F = args.height, args.width, args.frames
    distinct = min(16, F)
    base_np = np.stack([synth_frame(H, W, 42 + 1000 * rank + i) for i in range(distinct)])
    base = torch.from_numpy(base_np).to(dev)
    idx = torch.arange(F, device=dev) % distinct
    d_img = base[idx].contiguous()                       # [F, H, W] uint8, resident in HBM
    d_edges = torch.empty((F, H, W), dtype=torch.int16, device=dev)
    d_edges_b = torch.empty((F, H, W), dtype=torch.int16, device=dev) if args.stream else None
    del base
    torch.cuda.synchronize()

    ctx = capi.Context(local_rank)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)
    ctx.set_option("fuse_classify", args.fuse_classify)
    ctx.set_option("stream_overlap", 1 if args.stream == 2 else 0)

    def plain_step():
        ctx.dev_canny(d_img.data_ptr(), args.sigma, args.min_val, args.max_val, H, W, F, d_edges.data_ptr())

    calls = [0]

    def stream_step():
        # a stream of batches: the edge map of call i is complete once call i+1 (or the flush) has returned, so
        # consecutive calls write alternating buffers as a consumer of the maps would need them to
        out = d_edges if calls[0] % 2 == 0 else d_edges_b
        calls[0] += 1
        ctx.dev_canny_stream(d_img.data_ptr(), args.sigma, args.min_val, args.max_val, H, W, F, out.data_ptr())

    step = stream_step if args.stream else plain_step

    def drain():
        if args.stream:
            ctx.dev_canny_stream_flush()
        torch.cuda.synchronize()

    # parity spot check (outside the timed region): frame 0 of this rank against the oracle
    parity = None
    if not args.no_check and rank == 0:
        import oracle
        step()
        drain()
        got = d_edges[0].cpu().numpy()
        want = oracle.canny(base_np[0], args.sigma, args.min_val, args.max_val)
        parity = bool(np.array_equal(got, want))
        if not parity:
            raise SystemExit("bench.py: HIP edge map differs from the oracle -- refusing to report a number")

    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < args.spinup_seconds:  # clock spin-up, untimed (see --spinup-seconds)
        for _ in range(8):
            step()
        drain()
    for _ in range(args.warmup):
        step()
    drain()
    # Inside the timed region only the roofline kernel carries HIP events, and that pair is ATTACHED to the kernel's
    # dispatch (hipExtLaunchKernel start/stop events: the kernel's own begin/end timestamps) rather than recorded
    # before and after it: a recorded pair puts two barrier packets into the stream and cost ~0.03-0.06 ms per step,
    # with all stages bracketed ~0.08 ms (3 %).  The other stages are timed in a second, untimed pass of the same K
    # steps right after.
    # ... and only every 4th step carries that pair: even an attached pair makes the dispatch wait for the stream to
    # drain (~0.04 ms per step it is on); `launches_timed` says how many launches the average is over.
    ctx.set_option("profile_stage_mask", 1 << capi.STAGE_SOBEL_NMS)
    sample_every = max(1, min(4, args.steps // 3))
    ctx.set_option("profile_sample_interval", sample_every)
    ctx.profile_enable(True)
    ctx.profile_reset()

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()  # the last step's sweeps are inside the timed region
    elapsed = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
        elapsed = sharding.max_over_ranks(elapsed, red_dev)

    sn_ms_total, sn_launches = ctx.profile_get(capi.STAGE_SOBEL_NMS)
    ctx.set_option("profile_stage_mask", 0)  # all stages
    ctx.set_option("profile_sample_interval", 1)
    ctx.profile_reset()
    for _ in range(args.steps):
        step()
    drain()
    stages = {}
    for sid, name in enumerate(capi.STAGE_NAMES[:5]):
        ms, n = ctx.profile_get(sid)
        stages[name] = {"ms_per_step": round(ms / max(1, args.steps), 4), "launch_groups": n,
                        "timed": "second pass, all stages bracketed by events"}
    stages["sobel_nms"] = {"ms_per_step": round(sn_ms_total / max(1, sn_launches), 4), "launch_groups": sn_launches,
                           "timed": f"inside the timed region, on every {sample_every}th step" if sample_every > 1
                           else "inside the timed region, on every step"}
    ctx.profile_enable(False)
    hyst_sweeps = ctx.last_hysteresis_iterations

    px_per_step = F * H * W
    value = sharding.aggregate_throughput(px_per_step * args.steps, world, elapsed) / 1e6

    # the same steps as plain blocking calls (each returns with its edge map complete), for comparison
    plain = None
    if args.stream:
        plain_step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            plain_step()
        torch.cuda.synchronize()
        el_plain = time.perf_counter() - t1
        if world > 1:
            dist.barrier()
            el_plain = sharding.max_over_ranks(el_plain, red_dev)
        plain = {"value": round(sharding.aggregate_throughput(px_per_step * args.steps, world, el_plain) / 1e6, 1),
                 "unit": "Mpixels/s", "ms_per_step": round(el_plain / args.steps * 1e3, 4),
                 "what": "same workload through canny_hip_dev_canny (no overlap between consecutive steps)"}

    # ---- roofline -------------------------------------------------------------------------------------
    # PMC-measured HBM bytes per launch (profiles/traffic_sobel_nms.json, made by tools/pmc_passes.sh +
    # tools/pmc_summary.py on this same workload); null when the file does not cover this shape.
    measured = {}
    tpath = os.path.join(ROOT, "profiles", "traffic_sobel_nms.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            # the PMC figure belongs to the kernel source it was measured on: stale after any edit of that file
            import hashlib
            src_sha = hashlib.sha256(open(os.path.join(ROOT, "canny_edge_amd", "csrc",
                                                       "canny_sobel_nms_march.hip"), "rb").read()).hexdigest()
            if tj.get("frames") == F and tj.get("height") == H and tj.get("width") == W and \
                    tj.get("kernel_source_sha256") == src_sha:
                measured = tj.get("kernels", {})
        except Exception:
            measured = {}

    def roof(kernel, what, bytes_per_px, total_ms, launches, extra=None):
        """total_ms = summed device time of the `launches` launches of this kernel that carried events; every
        launch covers the whole batch of F frames (one launch per step)."""
        avg = total_ms / launches if launches else 0.0
        alg = bytes_per_px * px_per_step
        ach = alg / (avg * 1e-3) / 1e9 if avg > 0 else 0.0
        m = measured.get(kernel, {})
        traffic = m.get("hbm_bytes_per_launch") if m.get("frames_per_launch") == F else None
        r = {"kernel": what, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
             "algorithmic_bytes_per_px": bytes_per_px, "algorithmic_bytes_per_launch": int(alg),
             "avg_launch_ms": round(avg, 4), "launches_timed": launches, "launches_per_step": 1,
             "frames_per_launch": F}
        if extra:
            r.update(extra)
        return r

    fused = stages["hyst_classify"]["launch_groups"] == 0  # canny() ran Sobel+NMS with the classify step inside
    u8_plane = bool(ctx.get_option("last_canny_smoothed_u8"))  # the smoothed plane between the two kernels was bytes
    sn_ms, sn_n = sn_ms_total, sn_launches  # the launches of the timed region that carried events
    if fused and u8_plane:
        # Round 3 default: the Gaussian hands the fused kernel a u8 plane ((short)(sum/count) lies in [0,255],
        # src/utils.cpp:62).  The kernel of the timed region then moves 1 B/px in + 2 B/px provisional s16 edge map +
        # 2/8 B/px bit-planes out = 3.25 B/px, and is priced with THOSE bytes; the 4 B/px pass SURVEY.md 8(d) prices
        # (s16 in, s16 out) and the fused s16-input form are timed right after the region, below.
        roofline = roof("sobel_nms_classify_u8in",
                        "fused Sobel+NMS+threshold-classify on the u8 smoothed plane (u8 in; s16 edge map + "
                        "strong/connectable bit-planes out) -- what canny() runs", 3.25, sn_ms, sn_n,
                        {"limiter": "VALU issue (30.5 instructions per pixel, ~3.0 SIMD cycles each: PMC) beside an "
                                    "HBM stream that a plain copy moves 1.2-1.4x faster (copy_probe), see DESIGN.md"})
    elif fused:
        # s16 smoothed in (2 B/px); out: the provisional s16 edge map (2 B/px, completed in place by the propagation
        # sweeps -- there is no finalize pass) and the two 1-bit hysteresis planes (2/8 B/px).  `frac` / `achieved`
        # price it at SURVEY.md 8(d)'s 4 B/px; `frac_own_bytes` at the 4.25 B/px it really moves.
        roofline = roof("sobel_nms_classify",
                        "fused Sobel+NMS+threshold-classify (s16 smoothed in; s16 edge map + strong/connectable "
                        "bit-planes out), priced at SURVEY 8(d)'s 4 B/px", 4.0, sn_ms, sn_n,
                        {"limiter": "HBM stream (a plain copy of the same bytes: copy_probe) with VALU issue as "
                                    "co-limiter, see DESIGN.md"})
        own = roof("sobel_nms_classify", "", 4.25, sn_ms, sn_n)
        roofline["frac_own_bytes"] = own["frac"]
        roofline["achieved_own_bytes"] = own["achieved"]
        roofline["own_bytes_per_px"] = 4.25
    else:
        roofline = roof("sobel_nms", "fused Sobel+NMS (s16 smoothed in, s16 suppressed magnitude out)", 4.0,
                        sn_ms, sn_n)

    # SURVEY.md 8(d) prices "the fused Sobel+NMS pass" at 4 B/px (s16 in, s16 out).  That kernel is what the
    # stage API (nonmaximalSuppression after sobelOperator) runs; time it on the same resident batch, same
    # HIP-event mechanism, right after the timed region, so that both figures come from one run.
    d_sm = torch.empty((F, H, W), dtype=torch.int16, device=dev)
    ctx.dev_gaussian(d_img.data_ptr(), args.sigma, H, W, F, d_sm.data_ptr())
    ctx.dev_sobel_nms(d_sm.data_ptr(), H, W, F, d_edges.data_ptr())
    torch.cuda.synchronize()
    ctx.profile_enable(True)
    ctx.profile_reset()
    for _ in range(args.steps):
        ctx.dev_sobel_nms(d_sm.data_ptr(), H, W, F, d_edges.data_ptr())
    torch.cuda.synchronize()
    ms16, n16 = ctx.profile_get(capi.STAGE_SOBEL_NMS)
    ctx.profile_enable(False)
    # What does a plain device copy of the same 2 B/px in + 2 B/px out reach on THIS box, in THIS run?  (The roofline's
    # peak is the 8 TB/s spec; copies on MI355X boxes of this pool measured 4.8-6.2 TB/s: tools/probe_march_pattern.hip.)
    copy_ms = ctx.probe_copy(d_sm.data_ptr(), d_edges.data_ptr(), F * H * W * 2, launches=max(5, args.steps))
    copy_gbs = 4.0 * px_per_step / (copy_ms * 1e-3) / 1e9
    copy_probe = {"what": "plain device copy of one s16 plane of the batch into another (2 B/px read + 2 B/px written, "
                          "16 B per thread, one 4 KB chunk per workgroup), same run, HIP events attached to the dispatch",
                  "avg_launch_ms": round(copy_ms, 4), "achieved": round(copy_gbs, 1), "unit": "GB/s",
                  "frac_of_peak": round(copy_gbs / HBM_PEAK_GBS, 4)}
    roofline["copy_probe"] = copy_probe
    roofline["time_over_copy"] = round(roofline["avg_launch_ms"] / copy_ms, 3) if copy_ms > 0 else None
    del d_sm
    # The fused kernel in its s16-INPUT form (canny() with the option smoothed_u8 = 0: the pass whose bytes are
    # SURVEY 8(d)'s 4 B/px plus the bit-planes), same batch, same events, right after the region.
    roofline_fused_s16in = None
    if fused and u8_plane:
        ctx.set_option("smoothed_u8", 0)
        ctx.set_option("profile_stage_mask", 1 << capi.STAGE_SOBEL_NMS)
        plain_step()
        torch.cuda.synchronize()
        ctx.profile_enable(True)
        ctx.profile_reset()
        for _ in range(args.steps):
            plain_step()
        torch.cuda.synchronize()
        msf, nf = ctx.profile_get(capi.STAGE_SOBEL_NMS)
        ctx.profile_enable(False)
        ctx.set_option("profile_stage_mask", 0)
        ctx.set_option("smoothed_u8", 1)
        roofline_fused_s16in = roof("sobel_nms_classify", "fused Sobel+NMS+threshold-classify on the s16 smoothed plane "
                                    "(canny() with smoothed_u8 = 0), priced at SURVEY 8(d)'s 4 B/px", 4.0, msf, nf,
                                    {"timed": "after the timed region, same batch, HIP events on every launch"})
        own = roof("sobel_nms_classify", "", 4.25, msf, nf)
        roofline_fused_s16in["frac_own_bytes"] = own["frac"]
        roofline_fused_s16in["own_bytes_per_px"] = 4.25
        roofline_fused_s16in["time_over_copy"] = round(roofline_fused_s16in["avg_launch_ms"] / copy_ms, 3) if copy_ms > 0 else None
    roofline_s16 = roof("sobel_nms", "fused Sobel+NMS, stage-API form (s16 smoothed in, s16 suppressed magnitude out)",
                        4.0, ms16, n16, {"timed": "after the timed region, same batch, HIP events"})
    roofline_s16["time_over_copy"] = round(roofline_s16["avg_launch_ms"] / copy_ms, 3) if copy_ms > 0 else None

    per_kernel = {
        "gaussian": roof("gaussian_u8out" if u8_plane else "gaussian",
                         "separable Gaussian, rows+columns in one kernel (u8 in, %s out)" % ("u8" if u8_plane else "s16"),
                         2.0 if u8_plane else 3.0,
                         stages["gaussian"]["ms_per_step"] * args.steps, stages["gaussian"]["launch_groups"],
                         {"limiter": "VALU issue: separately rounded f32 mul/add chains (bit-exactness)"}),
    }
    if stages["hyst_finalize"]["launch_groups"]:
        per_kernel["hyst_finalize"] = roof("hyst_finalize", "hysteresis finalize (1-bit plane in, s16 edge map out)",
                                           2.125, stages["hyst_finalize"]["ms_per_step"] * args.steps,
                                           stages["hyst_finalize"]["launch_groups"])
    if not fused:
        per_kernel["hyst_classify"] = roof("hyst_classify", "hysteresis classify (s16 in, two 1-bit planes out)", 2.25,
                                           stages["hyst_classify"]["ms_per_step"] * args.steps,
                                           stages["hyst_classify"]["launch_groups"])

    # ---- host -> host (Metric 1 of SURVEY.md 8(d)); after every device-resident timing: the link-bound batches let the
    # clocks drop ------------------------------------------------------
    def sync_max(t):
        if world > 1:
            dist.barrier()
            return sharding.max_over_ranks(t, red_dev)
        return t

    h2h = None
    if args.h2h_frames > 0:
        ctx.set_stream(0)  # the context's own non-blocking stream: torch's current stream is the legacy default stream
        h2h = host_to_host(ctx, np, base_np, args, rank, world, sync_max, check=(not args.no_check and rank == 0))
        ctx.set_stream(stream.cuda_stream)
        h2h["gpu_local_cpus"] = local_cpus

    end_to_end = None
    if h2h:
        end_to_end = {"metric": "Mpixels/s end-to-end Canny (4K gray): SURVEY.md 8(d) Metric 1 -- host u8 frame in -> host "
                                "s16 edge map out, H2D and D2H inside the timed region (as src/cuda.cu:83-101 pays them)",
                      "value": h2h["s16"]["value"], "unit": "Mpixels/s", "ms_per_batch": h2h["s16"]["ms_per_batch"],
                      "frames_per_gpu": h2h["frames_per_gpu"], "n_gpus": world,
                      "bound": "PCIe %s: %.1f GB/s per GPU = %.0f %% of the %.0f GB/s link spec (the other direction: %.1f GB/s)" % (
                          ("H2D", h2h["s16"]["h2d_GBps_per_gpu"], 100 * h2h["s16"]["link_frac"], PCIE_PEAK_GBS,
                           h2h["s16"]["d2h_GBps_per_gpu"]) if h2h["s16"]["h2d_GBps_per_gpu"] >= h2h["s16"]["d2h_GBps_per_gpu"]
                          else ("D2H", h2h["s16"]["d2h_GBps_per_gpu"], 100 * h2h["s16"]["link_frac"], PCIE_PEAK_GBS,
                                h2h["s16"]["h2d_GBps_per_gpu"])),
                      "transfer": h2h["transfer"],
                      "u8_maps": h2h["u8"]["value"], "bit_maps": h2h["bits"]["value"]}
    out = {
        "metric": "Mpixels/s device-resident Canny (4K gray, frames in HBM, whole pipeline); % HBM roofline on "
                  "Sobel+NMS in `roofline`; the end-to-end (host->host, PCIe-inclusive) rate is `end_to_end`",
        "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32+s16",
        "scope": "device-resident: frames already in HBM when the timed region starts, edge maps left in HBM (the bench "
                 "contract's definition of `value`); `end_to_end` is the PCIe-inclusive Metric 1 of the same run",
        "end_to_end": end_to_end,
        "data": "synthetic",
        "config": {"workload": f"{F}x {W}x{H} gray frames per GPU per step, sigma={args.sigma}, "
                               f"thresholds {args.min_val}/{args.max_val}, inputs resident in HBM",
                   "frames_per_gpu": F, "height": H, "width": W, "sigma": args.sigma,
                   "calls": ("canny_hip_dev_canny_stream: the host checks step i's convergence after queueing the "
                             "Gaussian of step i+1; all K steps complete inside the timed region") if args.stream
                   else "canny_hip_dev_canny (asynchronous: K calls queued back to back, the timed region ends with a device "
                             "synchronize)",
                   "sharding": "independent frames per GPU, no collective"},
        "host_to_host": h2h,
        "roofline": roofline,
        "roofline_sobel_nms_s16": roofline_s16,
        "roofline_sobel_nms_classify_s16in": roofline_fused_s16in,
        "roofline_other_kernels": per_kernel,
        "stages": stages,
        "plain_calls": plain,
        "hysteresis_sweeps": hyst_sweeps,
        "parity_checked": parity,
    }
    if rehearse:
        out["rehearsal"] = True
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(base_np, args.sigma, args.min_val, args.max_val, args.cpu_frames)
        out["vs_cpu_baseline"] = {"device_resident": round(value / out["cpu_baseline"]["value"], 1),
                                  "end_to_end_s16": round(h2h["s16"]["value"] / out["cpu_baseline"]["value"], 1)
                                  if h2h else None,
                                  "note": "GPU / 1-thread CPU oracle; says nothing about kernel quality (roofline does). "
                                          "`vs_baseline` stays null: BASELINE.md publishes no number for this metric"}
        if end_to_end:
            end_to_end["vs_cpu_baseline"] = out["vs_cpu_baseline"]["end_to_end_s16"]
    else:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

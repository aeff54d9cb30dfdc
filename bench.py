#!/usr/bin/env python3
"""bench.py -- images/sec through resize + thumbnail + watermark on a batch of 1920x1080 RGBA
frames resident in HBM (BASELINE.json configs[2]; --workload resize gives configs[1]).

One process per GPU.  For N > 1 the driver launches this file under torch.distributed.run; the
frames are independent, so every rank runs the same batch size on its own GPU with no data-path
collective (weak scaling) and torch.distributed (gloo, CPU tensors) only carries the barrier and
the max-over-ranks of the timed region.

A "step" is one pass of the hot path over the whole batch (one fused launch).  Prints ONE JSON
line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is achievable


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=1024, help="frames per GPU per step")
    ap.add_argument("--workload", choices=["full", "resize", "full-keepaspect", "wm", "thumb", "resize-wm"], default="full")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--pool", type=int, default=32, help="distinct seeded frames tiled over the batch")
    ap.add_argument("--cpu-sample", type=int, default=384, help="frames timed on the CPU oracle (0 = skip)")
    ap.add_argument("--check", action="store_true", help="compare frame 0 with the oracle after the run")
    ap.add_argument("--mixed", type=int, default=0, metavar="N",
                    help="BASELINE config 5 instead: N frames per GPU of mixed sizes (480p-8K, seed 0x51), full "
                         "pipeline, pull scheduling over a shared largest-first queue (work stealing across ranks)")
    return ap.parse_args()


MIXED_SIZES = [(854, 480), (1280, 720), (1920, 1080), (2560, 1440), (3840, 2160), (7680, 4320)]


def run_mixed(args, ipx, shard, rank, local_rank, world):
    """Config 5: frames of six sizes drawn uniformly (seed 0x51); every rank holds a small seeded pool per
    size in HBM; work items are chunks of equal-size frames (~256 MB of source each), ordered largest
    first, claimed with an atomic counter in the rendezvous store."""
    import numpy as np
    import torch.distributed as dist
    from helpers import DEFAULT_COL, rgba_frames, text_glyphs
    rng = np.random.default_rng(0x51)
    total = args.mixed * world
    draw = rng.integers(0, len(MIXED_SIZES), total)
    ctx = ipx.Context(device=local_rank % max(1, ipx.device_count()))
    plans, bufs, chunk_of = {}, {}, {}
    for si, (w, h) in enumerate(MIXED_SIZES):
        fb = w * h * 4
        chunk_of[si] = max(1, min(64, (256 << 20) // fb))
        gs = ctx.glyphset(text_glyphs(w, h), DEFAULT_COL)
        pl = ctx.plan(w, h, resize=(1024, 768, True), thumbnail=(200, True), watermark=gs)
        n = chunk_of[si]
        src = ctx.alloc(n * fb)
        one = rgba_frames(1, w, h, seed=0x1F00D + si)
        for i in range(n):
            src.upload(one, offset=i * fb)
        i_ = pl.info
        bufs[si] = (src, ctx.alloc(n * i_.resize_bytes), ctx.alloc(n * i_.thumb_bytes), ctx.alloc(n * i_.wm_bytes))
        plans[si] = pl
    # items: (size index, frame count), largest total bytes first
    items = []
    for si in range(len(MIXED_SIZES)):
        cnt = int((draw == si).sum())
        while cnt > 0:
            m = min(cnt, chunk_of[si])
            items.append((si, m))
            cnt -= m
    order = shard.lpt_order([shard.frame_cost(*MIXED_SIZES[si]) * m for si, m in items])
    q = shard.WorkQueue(len(order), chunk=1, key="ipx_mixed", store=shard.default_store())

    def run_item(si, m):
        src, res, th, wm = bufs[si]
        plans[si].run_dev(m, src.ptr, res.ptr, th.ptr, wm.ptr)
        ctx.sync()     # pull scheduling: claim again only when this chunk is done

    for si in range(len(MIXED_SIZES)):   # warm every plan once
        run_item(si, 1)
    ctx.device_sync()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    mine = 0
    while True:
        c = q.claim()
        if c is None:
            break
        for pos in c:
            si, m = items[order[pos]]
            run_item(si, m)
            mine += m
    ctx.device_sync()
    if world > 1:
        dist.barrier()
    frames_done, elapsed = shard.aggregate(mine, time.perf_counter() - t0)
    counts = [mine]
    if world > 1:
        import torch
        t = torch.zeros(world, dtype=torch.int64)
        t[rank] = mine
        dist.all_reduce(t)
        counts = t.tolist()
    if rank == 0:
        print(json.dumps({
            "metric": "images/sec (resize+thumb+watermark) on mixed-size batch (480p-8K), work stealing",
            "value": round(frames_done / elapsed, 1), "unit": "images/sec", "n_gpus": world, "steps": 1, "warmup": 1,
            "ms_per_step": round(elapsed * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic: one seeded frame per size tiled per chunk, resident in HBM",
            "config": {"workload": "%d frames of sizes %s drawn uniformly (seed 0x51), full pipeline keep_aspect=true"
                                   % (int(frames_done), MIXED_SIZES), "items": len(items),
                       "frames_per_rank": counts, "sharding": "pull scheduling, largest first, atomic counter in the store"},
            "roofline": None, "cpu_baseline": None}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the ranks as children (no exec after
    GPU init: nothing here has touched the GPU yet)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"),
           os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


def cpu_baseline(pool, glyphs, col, resize, thumb, want, nsample):
    """The CPU oracle (scalar C restatement of the reference's loops, one frame per thread, as Go's
    loops are single-threaded per image) on a bounded sample of the same workload."""
    import concurrent.futures as cf
    import oracle
    cores = max(1, min(len(os.sched_getaffinity(0)), 64))

    def one(i):
        oracle.process(pool[i % len(pool)], resize=resize or (1, 1, False), thumb=thumb or (1, False),
                       glyphs=glyphs, col=col, want=want)

    one(0)  # warm: page in the library
    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(cores) as ex:
        list(ex.map(one, range(nsample)))
    dt = time.perf_counter() - t0
    # the reference deploys WORKER_CONCURRENCY=3 goroutines (.env.example:38)
    n3 = max(3, min(nsample, 24))
    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(3) as ex:
        list(ex.map(one, range(n3)))
    dt3 = time.perf_counter() - t0
    return {"value": round(nsample / dt, 2), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "%d frames of the same workload through oracle/ipx_oracle.c (C restatement of the "
                      "Go loops, -O2, no FMA), %d threads, one frame per thread; %.1f s wall" % (nsample, cores, dt),
            "at_reference_concurrency_3": round(n3 / dt3, 2)}


def main():
    args = parse()
    if args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        self_launch(args)

    import numpy as np
    import imageprocessor_amd as ipx
    from imageprocessor_amd import shard
    from helpers import DEFAULT_COL, rgba_frames, text_glyphs

    # one process per GPU; gloo carries the barrier and the max-over-ranks only (no data-path collective)
    rank, local_rank, world = shard.init_from_env()
    if args.mixed:
        return run_mixed(args, ipx, shard, rank, local_rank, world)
    dist = None
    if world > 1:
        import torch.distributed as dist

    sw, sh, F = args.width, args.height, args.frames
    resize = {"full": (1024, 768, False), "resize": (1024, 768, False), "full-keepaspect": (1024, 768, True),
              "wm": None, "thumb": None, "resize-wm": (1024, 768, False)}[args.workload]
    thumb = None if args.workload in ("resize", "wm", "resize-wm") else (200, True)
    do_wm = args.workload not in ("resize", "thumb")

    ctx = ipx.Context(device=local_rank % max(1, ipx.device_count()))
    glyphs = text_glyphs(sw, sh) if do_wm else []
    gs = ctx.glyphset(glyphs, DEFAULT_COL) if do_wm else None
    plan = ctx.plan(sw, sh, resize=resize, thumbnail=thumb, watermark=gs)
    info = plan.info

    # synthetic batch: `pool` seeded opaque frames (seed 0x1F00D + rank), tiled over F slots in HBM
    P = min(args.pool, F)
    pool = rgba_frames(P, sw, sh, seed=0x1F00D + rank)
    fbytes = sw * sh * 4
    src = ctx.alloc(F * fbytes)
    src.upload(pool)
    for i in range(P, F, P):
        m = min(P, F - i)
        ctx.copy_d2d(src.ptr + i * fbytes, src.ptr, m * fbytes)
    # experiment switch: IPX_BENCH_PAD=bytes widens the OUTPUT frame strides (a 1024x768 frame is exactly 3 MiB)
    pad = int(os.environ.get("IPX_BENCH_PAD", "0"))
    res = ctx.alloc(F * (info.resize_bytes + pad)) if info.resize_bytes else None
    th = ctx.alloc(F * (info.thumb_bytes + pad)) if info.thumb_bytes else None
    wm = ctx.alloc(F * info.wm_bytes) if info.wm_bytes else None

    def step():
        plan.run_dev(F, src.ptr, res.ptr if res else None, th.ptr if th else None, wm.ptr if wm else None,
                     resize_frame_stride=info.resize_bytes + pad, thumb_frame_stride=info.thumb_bytes + pad)

    L = ipx.lib()
    for _ in range(args.warmup):
        step()
    ctx.device_sync()
    if dist:
        dist.barrier()
    K = args.steps
    evs = [(L.ipx_event_create(ctx.handle), L.ipx_event_create(ctx.handle)) for _ in range(K)]
    t0 = time.perf_counter()
    for k in range(K):
        L.ipx_event_record(ctx.handle, evs[k][0], None)   # HIP events on the stream the kernel runs on
        step()
        L.ipx_event_record(ctx.handle, evs[k][1], None)
    ctx.device_sync()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    import ctypes
    launch_ms = []
    for e0, e1 in evs:
        ms = ctypes.c_float()
        L.ipx_event_elapsed_ms(ctx.handle, e0, e1, ctypes.byref(ms))
        launch_ms.append(ms.value)
        L.ipx_event_destroy(ctx.handle, e0)
        L.ipx_event_destroy(ctx.handle, e1)
    frames_done, elapsed = shard.aggregate(F * K, elapsed)   # frames summed over ranks, time = max over ranks

    if args.check and rank == 0:
        import oracle
        want = oracle.process(pool[0], resize=resize or (1, 1, False), thumb=thumb or (1, False), glyphs=glyphs, col=DEFAULT_COL,
                              want=[k for k, b in (("resize", res), ("thumbnail", th), ("watermark", wm)) if b])
        if res:
            assert np.array_equal(res.download((info.resize_h, info.resize_w, 4)), want["resize"])
        if th:
            assert np.array_equal(th.download((info.thumb_h, info.thumb_w, 4)), want["thumbnail"])
        if wm:
            assert np.array_equal(wm.download((sh, sw, 4)), want["watermark"])
        print("check ok", file=sys.stderr)

    if rank == 0 and os.environ.get("IPX_BENCH_TRACE"):
        print("launch_ms:", " ".join("%.3f" % m for m in launch_ms), file=sys.stderr)
    if rank == 0:
        avg_ms = sum(launch_ms) / len(launch_ms)
        alg = info.algorithmic_bytes * F  # source read once + each output written once
        achieved = alg / (avg_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                with open(tpath) as f:
                    tj = json.load(f)
                key = "%s_%dx%d_%d" % (args.workload, sw, sh, F)
                traffic = tj.get(key, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "images/sec (resize+thumb+watermark) on 1080p batch at 1/2/4/8 MI355X",
            "value": round(frames_done / elapsed, 1),
            "unit": "images/sec",
            "n_gpus": world,
            "steps": K,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / K * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic: %d seeded opaque RGBA8 frames per GPU tiled over %d slots resident in HBM" % (P, F),
            "config": {"workload": "%d x %dx%d RGBA8, %s" % (F, sw, sh, {
                "full": "full pipeline: resize 1024x768 (keep_aspect=false) + thumbnail 200 crop + watermark 16 glyphs",
                "resize": "resize to 1024x768 only",
                "full-keepaspect": "full pipeline, product-default keep_aspect=true (1024x576)",
                "wm": "watermark only (copy + 16 glyphs)", "thumb": "thumbnail 200 crop only",
                "resize-wm": "resize 1024x768 + watermark"}[args.workload]),
                "frames_per_gpu": F, "sharding": "independent frames, round-robin by rank, no collective",
                "arithmetic": "u8 pixels; taps interpolated in f64 (exact fp32 on dyadic axes), composite in u32"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": "band_pipe_kernel", "algorithmic_bytes_per_launch": alg,
                         "avg_launch_ms": round(avg_ms, 4)},
        }
        if world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(pool, glyphs, DEFAULT_COL, resize, thumb,
                                               [k for k, b in (("resize", res), ("thumbnail", th), ("watermark", wm)) if b],
                                               args.cpu_sample)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)

    if dist:
        dist.barrier()
        dist.destroy_process_group()
    plan.close()
    ctx.close()


if __name__ == "__main__":
    main()

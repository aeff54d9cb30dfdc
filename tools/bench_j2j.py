#!/usr/bin/env python3
"""Compressed in, compressed out alone (ipx_plan_run_jpeg_jpeg): n 1080p 4:2:0 q85 files -> decode + resize 1024x576 + thumbnail 200 + watermark
+ three jpeg.Encode on the GPU.  For rocprofv3: the kernel sums against the wall time say how much of a call is GPU work.
usage: tools/bench_j2j.py [files] [reps] [concurrent callers]; IPX_BENCH_PROGRESSIVE=1: progressive files (their scans are decoded on host threads)"""
import io
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from PIL import Image  # noqa: E402

import imageprocessor_amd as ipx  # noqa: E402
from helpers import DEFAULT_COL, text_glyphs  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sw, sh = 1920, 1080
ctx = ipx.Context(lanes=int(os.environ.get("IPX_BENCH_LANES", "5")), lane_bytes=1 << 30)
gs = ctx.glyphset(text_glyphs(sw, sh), DEFAULT_COL)
plan = ctx.plan(sw, sh, resize=(1024, 768, True), thumbnail=(200, True), watermark=gs)
yy, xx = np.mgrid[0:sh, 0:sw]
files = []
for k in range(4):
    base = np.stack([np.sin(xx / (40.0 + 7 * k)) * 90 + 128, np.cos(yy / (31.0 + 5 * k)) * 90 + 128, ((xx + 2 * yy) / 6.0 + 40 * k) % 256], -1)
    img = (base + np.random.default_rng(k).normal(0, 6, (sh, sw, 3))).clip(0, 255).astype(np.uint8)
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, "JPEG", quality=85, progressive=bool(os.environ.get("IPX_BENCH_PROGRESSIVE")))
    files.append(buf.getvalue())
batch = [files[i % 4] for i in range(n)]
plan.run_jpeg_jpeg(batch[:64], copy=False)
best = 1e9
for _ in range(reps):
    t0 = time.perf_counter()
    lens, st = plan.run_jpeg_jpeg(batch, copy=False)
    best = min(best, time.perf_counter() - t0)
    if os.environ.get("IPX_BENCH_VERBOSE"):
        print("  repetition: %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
assert not any(st)
print(("progressive " if os.environ.get("IPX_BENCH_PROGRESSIVE") else "") + "JPEG files (%.0f KB) -> three JPEG streams: %d files in %.1f ms = %.0f images/s" % (len(files[0]) / 1e3, n, best * 1e3, n / best))
callers = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if callers:
    import threading

    def call():
        out, st = plan.run_jpeg_jpeg(batch, copy=False)
        assert not any(st)
    for rep in range(reps):
        ths = [threading.Thread(target=call) for _ in range(callers)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        dt = time.perf_counter() - t0
        print("%d concurrent callers x %d files, repetition %d: %.1f ms = %.0f images/s" % (callers, n, rep, dt * 1e3, callers * n / dt), flush=True)

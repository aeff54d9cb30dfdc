#!/usr/bin/env python3
"""Every repetition of the host legs, not the best one: stalls (allocation, pinning) hide behind best-of-N.
usage: tools/bench_reps.py [frames] [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import imageprocessor_amd as ipx  # noqa: E402
from helpers import DEFAULT_COL, rgba_frames, text_glyphs  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 16
sw, sh = 1920, 1080
ctx = ipx.Context(lanes=4, lane_bytes=1 << 30)
gs = ctx.glyphset(text_glyphs(sw, sh), DEFAULT_COL)
plan = ctx.plan(sw, sh, resize=(1024, 768, False), thumbnail=(200, True), watermark=gs)
src = ctx.host_alloc((n, sh, sw, 4))
src[:] = np.resize(rgba_frames(4, sw, sh, seed=3), src.shape)
dev = ctx.alloc(src.nbytes).upload(src)
for label, fn in (("run_host_jpeg (frames in host memory -> three streams)", lambda: plan.run_host_jpeg(src, copy=False)),
                  ("jpeg_encode_batch_dev (frames in HBM -> streams)", lambda: ctx.jpeg_encode_batch_dev(dev.ptr, sw, sh, n, 85, copy=False)[1]()),
                  ("run_host (pixels in, pixels out)", lambda: plan.run_host(src))):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    print("%-58s %d frames, ms per repetition: %s" % (label, n, " ".join("%.0f" % t for t in ts)), flush=True)

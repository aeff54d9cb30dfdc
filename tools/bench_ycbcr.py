#!/usr/bin/env python3
"""Throughput of the decoded-JPEG batch path (ipx_plan_run_dev_ycbcr): n x 1920x1080 4:2:0 frames resident in HBM -> resize 1024x576 +
thumbnail 200 + watermark through the one-pass kernel ks_fused_kernel<KS_YCC>; IPX_FUSED=0 times the per-output
kernels instead."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import imageprocessor_amd as ipx  # noqa: E402
from helpers import DEFAULT_COL, text_glyphs  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ops = sys.argv[2] if len(sys.argv) > 2 else "full"     # full | wm | resize | thumb | resize-wm
w, h, ratio = 1920, 1080, 2
cw, ch = (w + 1) // 2, (h + 1) // 2
ctx = ipx.Context()
gs = ctx.glyphset(text_glyphs(w, h), DEFAULT_COL)
plan = ctx.plan(w, h, resize=(1024, 768, True) if ops in ("full", "resize", "resize-wm") else None,
                thumbnail=(200, True) if ops in ("full", "thumb") else None, watermark=gs if ops in ("full", "wm", "resize-wm") else None)
i = plan.info
rng = np.random.default_rng(3)
pool = 8
sets = int(sys.argv[3]) if len(sys.argv) > 3 else 1    # buffer sets, each with new addresses: the run time depends on where the buffers land
keep = []
alg = n * (w * h * 1.5 + i.resize_bytes + i.thumb_bytes + i.wm_bytes)
for _ in range(sets):
    y = ctx.alloc(n * w * h).upload(np.resize(rng.integers(0, 256, (pool, h, w), dtype=np.uint8), (n, h, w)))
    cb = ctx.alloc(n * cw * ch).upload(np.resize(rng.integers(0, 256, (pool, ch, cw), dtype=np.uint8), (n, ch, cw)))
    cr = ctx.alloc(n * cw * ch).upload(np.resize(rng.integers(0, 256, (pool, ch, cw), dtype=np.uint8), (n, ch, cw)))
    res, th, wm = ctx.alloc(n * i.resize_bytes), ctx.alloc(n * i.thumb_bytes), ctx.alloc(n * i.wm_bytes)
    keep.append((y, cb, cr, res, th, wm))

    def step():
        plan.run_dev_ycbcr(n, y.ptr, cb.ptr, cr.ptr, ratio, w, cw, w * h, cw * ch, res.ptr, th.ptr, wm.ptr)

    for _ in range(3):
        step()
    ctx.device_sync()
    ms = min(ctx.timed(step) for _ in range(10))
    print("ycbcr 4:2:0 batch (" + ops + "): %d frames in %.3f ms = %.0f images/s; algorithmic %.1f GB/s (1.5 B/px in + outputs = %.2f MB per frame)"
          % (n, ms, n / ms * 1e3, alg / ms / 1e6, alg / n / 1e6), flush=True)

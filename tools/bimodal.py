#!/usr/bin/env python3
"""Does the run time of the full 1080p pipeline depend on where the batch buffers lie?  Several buffer sets in one process (earlier ones
kept, so each set gets new addresses), the same plan, per-launch times of each.  usage: tools/bimodal.py [sets] [frames]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import imageprocessor_amd as ipx  # noqa: E402
from helpers import DEFAULT_COL, text_glyphs  # noqa: E402

sets = int(sys.argv[1]) if len(sys.argv) > 1 else 4
F = int(sys.argv[2]) if len(sys.argv) > 2 else 512
sw, sh = 1920, 1080
ctx = ipx.Context()
gs = ctx.glyphset(text_glyphs(sw, sh), DEFAULT_COL)
plan = ctx.plan(sw, sh, resize=(1024, 768, False), thumbnail=(200, True), watermark=gs)
info = plan.info
rng = np.random.default_rng(1)
pool = rng.integers(0, 256, (8, sh, sw, 4), dtype=np.uint8)
pool[..., 3] = 255
keep = []
for s in range(sets):
    src = ctx.alloc(F * sw * sh * 4)
    for i in range(0, F, 8):
        src.upload(pool[:min(8, F - i)], offset=i * sw * sh * 4) if hasattr(src, "upload") and False else None
    src.upload(pool)
    for i in range(8, F, 8):
        ctx.copy_d2d(src.ptr + i * sw * sh * 4, src.ptr, min(8, F - i) * sw * sh * 4)
    res, th, wm = ctx.alloc(F * info.resize_bytes), ctx.alloc(F * info.thumb_bytes), ctx.alloc(F * info.wm_bytes)
    keep.append((src, res, th, wm))

    def step():
        plan.run_dev(F, src.ptr, res.ptr, th.ptr, wm.ptr)

    for _ in range(3):
        step()
    ctx.device_sync()
    ms = [ctx.timed(step) for _ in range(8)]
    print("set %d  src %#x res %#x th %#x wm %#x   ms/launch: %s" % (s, src.ptr, res.ptr, th.ptr, wm.ptr, " ".join("%.3f" % m for m in ms)), flush=True)
    if s % 2 == 1:
        keep.append(ctx.alloc((3 + s) << 20))      # shift the next set by an odd number of MiB

#!/usr/bin/env python3
"""The pixels-to-pixels host leg of bench.py alone (256 x 1080p RGBA frames in pinned memory per call), for sweeping the knobs of
run_host_packed: IPX_HOST_CHUNK, IPX_COPY_PIECE_MB, IPX_HOST_STAGED, IPX_LANES."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import imageprocessor_amd as ipx
from helpers import DEFAULT_COL, text_glyphs

n, sw, sh = int(os.environ.get("N", 256)), 1920, 1080
lanes = int(os.environ.get("IPX_LANES", 5))
ctx = ipx.Context(device=0, lanes=lanes, lane_bytes=1 << 30)
gs = ctx.glyphset(text_glyphs(sw, sh), DEFAULT_COL)
plan = ctx.plan(sw, sh, resize=(1024, 768, False), thumbnail=(200, True), watermark=gs)
i = plan.info
src = ctx.host_alloc((n, sh, sw, 4))
src[:] = np.random.default_rng(1).integers(0, 256, (1, sh, sw, 4), dtype=np.uint8)
src[..., 3] = 255
outs = {"resize": ctx.host_alloc((n, i.resize_h, i.resize_w, 4)), "thumbnail": ctx.host_alloc((n, i.thumb_h, i.thumb_w, 4)),
        "watermark": ctx.host_alloc((n, i.wm_h, i.wm_w, 4))}
plan.run_host(src, out=outs)
ms = []
for _ in range(5):
    t0 = time.perf_counter()
    plan.run_host(src, out=outs)
    ms.append((time.perf_counter() - t0) * 1e3)
b = n * (sw * sh * 4 + i.resize_bytes + i.thumb_bytes + i.wm_bytes)
best = min(ms)
print("chunk=%s piece=%s staged=%s lanes=%d: best %.2f ms = %.0f images/s, %.1f GB/s both ways; all %s" % (
    os.environ.get("IPX_HOST_CHUNK"), os.environ.get("IPX_COPY_PIECE_MB"), os.environ.get("IPX_HOST_STAGED"), lanes, best, n / best * 1e3, b / best / 1e6,
    [round(v, 1) for v in ms]), flush=True)

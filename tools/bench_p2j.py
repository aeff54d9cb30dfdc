#!/usr/bin/env python3
"""bench.py's pixels -> three JPEG streams leg alone (256 x 1080p RGBA frames in pinned memory, ipx_plan_run_host_jpeg), for sweeping
IPX_HOST_CHUNK_JPEG and IPX_LANES."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import bench
import imageprocessor_amd as ipx
from helpers import DEFAULT_COL, text_glyphs

n, sw, sh = 256, 1920, 1080
ctx = ipx.Context(device=0, lanes=int(os.environ.get("IPX_LANES", 5)), lane_bytes=1 << 30)
gs = ctx.glyphset(text_glyphs(sw, sh), DEFAULT_COL)
plan = ctx.plan(sw, sh, resize=(1024, 768, False), thumbnail=(200, True), watermark=gs)
frames = bench.photo_like(4, sw, sh)
src = ctx.host_alloc((n, sh, sw, 4))
for k in range(n):
    src[k] = frames[k % 4]
plan.run_host_jpeg(src, 85, copy=False)
ms = []
for _ in range(5):
    t0 = time.perf_counter()
    plan.run_host_jpeg(src, 85, copy=False)
    ms.append((time.perf_counter() - t0) * 1e3)
print("chunk=%s lanes=%s: best %.1f ms = %.0f images/s (%.1f GB/s up); all %s" % (os.environ.get("IPX_HOST_CHUNK_JPEG"), os.environ.get("IPX_LANES"), min(ms), n / min(ms) * 1e3,
      n * sw * sh * 4 / min(ms) / 1e6, [round(v, 1) for v in ms]), flush=True)

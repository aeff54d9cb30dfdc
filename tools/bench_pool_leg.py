#!/usr/bin/env python3
"""bench.py's pool leg alone (256 x 1080p RGBA frames as one pixel job on device 0), for sweeping POOL_LANES / POOL_LANE_MB /
IPX_HOST_DIRECT."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import imageprocessor_amd as ipx
from helpers import DEFAULT_COL, text_glyphs

n, sw, sh = int(os.environ.get("N", 256)), 1920, 1080
lanes, lane_mb = int(os.environ.get("POOL_LANES", 0)), int(os.environ.get("POOL_LANE_MB", 0))
glyphs = text_glyphs(sw, sh)
if os.environ.get("PRE_CTX"):            # a context that lived (and was closed) before the pool is made
    c0 = ipx.Context(device=0)
    if os.environ["PRE_CTX"] == "plan":
        p0 = c0.plan(sw, sh, resize=(1024, 768, False), thumbnail=(200, True), watermark=True)
        p0.close()
    c0.close()
with ipx.Pool(devices=(0,), lanes_per_device=lanes, lane_bytes=lane_mb << 20) as pool:
    src = pool.host_alloc(0, (n, sh, sw, 4))
    src[:] = np.random.default_rng(1).integers(0, 256, (1, sh, sw, 4), dtype=np.uint8)
    src[..., 3] = 255
    outs = dict(resize=pool.host_alloc(0, (n, 768, 1024, 4)), thumbnail=pool.host_alloc(0, (n, 200, 200, 4)), watermark=pool.host_alloc(0, (n, sh, sw, 4)))
    def run():
        pool.submit(src, resize=(1024, 768, False), thumbnail=(200, True), glyphs=glyphs, col=DEFAULT_COL, out=outs).wait()
    run()
    ms = []
    for _ in range(5):
        t0 = time.perf_counter()
        run()
        ms.append((time.perf_counter() - t0) * 1e3)
    best = min(ms)
    print("lanes=%d lane_mb=%d direct=%s: best %.2f ms = %.0f images/s; all %s" % (lanes, lane_mb, os.environ.get("IPX_HOST_DIRECT"), best, n / best * 1e3, [round(v, 1) for v in ms]), flush=True)

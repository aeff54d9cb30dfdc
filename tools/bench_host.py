#!/usr/bin/env python3
"""End-to-end rate of the host boundary: frames in pinned host memory -> H2D -> fused kernel -> D2H
-> pinned host memory, pipelined over the context's lanes (ipx_plan_run_host).  This is the
PCIe-inclusive figure quoted in DESIGN.md; it is never bench.py's `value`."""
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
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
sw, sh = 1920, 1080
ctx = ipx.Context(lanes=lanes, lane_bytes=1 << 30)
gs = ctx.glyphset(text_glyphs(sw, sh), DEFAULT_COL)
plan = ctx.plan(sw, sh, resize=(1024, 768, False), thumbnail=(200, True), watermark=gs)
i = plan.info
for pinned in (True, False):
    alloc = ctx.host_alloc if pinned else (lambda shape: np.empty(shape, np.uint8))
    src = alloc((n, sh, sw, 4))
    src[:] = np.resize(rgba_frames(8, sw, sh, seed=3), src.shape)
    out = {"resize": alloc((n, i.resize_h, i.resize_w, 4)), "thumbnail": alloc((n, i.thumb_h, i.thumb_w, 4)),
           "watermark": alloc((n, i.wm_h, i.wm_w, 4))}
    plan.run_host(src, out=out)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        plan.run_host(src, out=out)
        best = min(best, time.perf_counter() - t0)
    gb = n * (sw * sh * 4 + i.resize_bytes + i.thumb_bytes + i.wm_bytes) / 1e9
    print("host->host %s memory, %d lanes: %d frames in %.1f ms = %.0f images/s, %.1f GB/s over PCIe (both directions summed)"
          % ("pinned" if pinned else "pageable", lanes, n, best * 1e3, n / best, gb / best))

# the same leg with jpeg.Encode on the GPU: only the finished streams come back
src = ctx.host_alloc((n, sh, sw, 4))
yy, xx = np.mgrid[0:sh, 0:sw]
for k in range(n):   # photograph-like content (smooth + mild texture): noise would make the streams unrealistically large
    if k < 4:
        base = np.stack([np.sin(xx / (40.0 + 7 * k)) * 90 + 128, np.cos(yy / (31.0 + 5 * k)) * 90 + 128, ((xx + 2 * yy) / 6.0 + 40 * k) % 256], -1)
        src[k, ..., :3] = (base + np.random.default_rng(k).normal(0, 6, (sh, sw, 3))).clip(0, 255)
        src[k, ..., 3] = 255
    else:
        src[k] = src[k % 4]
plan.run_host_jpeg(src[:8], copy=False)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    lens = plan.run_host_jpeg(src, copy=False)
    best = min(best, time.perf_counter() - t0)
kb = sum(sum(v) for v in lens.values()) / n / 1e3
print("host frames -> three JPEG streams (operators + jpeg.Encode on the GPU), pinned source, %d lanes: %d frames in %.1f ms = %.0f images/s; "
      "%.0f KB of streams per frame come back instead of %.1f MB of pixels" % (lanes, n, best * 1e3, n / best, kb,
                                                                               (i.resize_bytes + i.thumb_bytes + i.wm_bytes) / 1e6))

# decoded JPEGs: 4:2:0 planes in (1.5 B per pixel up), three streams out
cw, chh = sw // 2, sh // 2
yp, cbp, crp = ctx.host_alloc((n, sh, sw)), ctx.host_alloc((n, chh, cw)), ctx.host_alloc((n, chh, cw))
for k in range(n):
    f = src[k % 4].astype(np.int32)
    r, g, b = f[..., 0], f[..., 1], f[..., 2]
    yp[k] = ((19595 * r + 38470 * g + 7471 * b + 32768) >> 16).clip(0, 255)
    cbp[k] = (((-11056 * r - 21712 * g + 32768 * b + (257 << 15)) >> 16).clip(0, 255))[::2, ::2]
    crp[k] = (((32768 * r - 27440 * g - 5328 * b + (257 << 15)) >> 16).clip(0, 255))[::2, ::2]
plan2 = ctx.plan(sw, sh, resize=(1024, 768, True), thumbnail=(200, True), watermark=gs)
plan2.run_host_ycbcr_jpeg(yp[:8], cbp[:8], crp[:8], 2, copy=False)
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    lens = plan2.run_host_ycbcr_jpeg(yp, cbp, crp, 2, copy=False)
    best = min(best, time.perf_counter() - t0)
kb = sum(sum(v) for v in lens.values()) / n / 1e3
print("decoded-JPEG planes (4:2:0) -> three JPEG streams, pinned source, %d lanes: %d frames in %.1f ms = %.0f images/s; 3.1 MB up, %.0f KB down per frame"
      % (lanes, n, best * 1e3, n / best, kb))

# compressed in, compressed out: the worker's whole per-message job on the GPU
import io
from PIL import Image
files, files_rst = [], []
for k in range(4):
    for dst, kw in ((files, {}), (files_rst, {"restart_marker_rows": 1})):
        buf = io.BytesIO()
        Image.fromarray(src[k, ..., :3]).save(buf, "JPEG", quality=85, **kw)
        dst.append(buf.getvalue())
for label, pool_ in (("no restart markers", files), ("one restart interval per MCU row", files_rst)):
    for m in (n, 4 * n):
        batch = [pool_[i % 4] for i in range(m)]
        plan2.run_jpeg_jpeg(batch[:64], copy=False)
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            lens, st = plan2.run_jpeg_jpeg(batch, copy=False)
            best = min(best, time.perf_counter() - t0)
        assert not any(st)
        print("JPEG files (%s, %.0f KB) -> decode + operators + three encodes on the GPU: %d files in %.1f ms = %.0f images/s"
              % (label, len(batch[0]) / 1e3, m, best * 1e3, m / best))

# two callers at once (the worker's goroutines): each call leases its own lane, so one batch decodes while the other encodes
import threading
batch = [files[i % 4] for i in range(1024)]
def call():
    out, st = plan2.run_jpeg_jpeg(batch, copy=False)
    assert not any(st)
for nthreads in (2, 3):
    best = 1e9
    for _ in range(2):
        ths = [threading.Thread(target=call) for _ in range(nthreads)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        best = min(best, time.perf_counter() - t0)
    print("JPEG files -> JPEG streams, %d concurrent callers x 1024 files: %.1f ms = %.0f images/s" % (nthreads, best * 1e3, nthreads * 1024 / best))

#!/usr/bin/env python3
"""image.Decode throughput for baseline JPEG batches on the GPU (ipx_jpeg_decode_batch: compressed bytes in host memory ->
*image.YCbCr planes in HBM), next to the CPU oracle and libjpeg (Pillow) on one host thread.
usage: tools/bench_jpeg_dec.py [frames ...]"""
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
import oracle  # noqa: E402

w, h = 1920, 1080
yy, xx = np.mgrid[0:h, 0:w]
pool, pool_rst = [], []
for k in range(4):
    base = np.stack([np.sin(xx / (40.0 + 7 * k)) * 90 + 128, np.cos(yy / (31.0 + 5 * k)) * 90 + 128, ((xx + 2 * yy) / 6.0 + 40 * k) % 256], -1)
    img = (base + np.random.default_rng(k).normal(0, 6, (h, w, 3))).clip(0, 255).astype(np.uint8)
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, "JPEG", quality=85)
    pool.append(buf.getvalue())
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, "JPEG", quality=85, restart_marker_rows=1)
    pool_rst.append(buf.getvalue())
print("files: %dx%d 4:2:0 q85, %.0f KB each" % (w, h, sum(len(p) for p in pool) / 4 / 1e3))
ctx = ipx.Context()
for n, src, label in [(int(a), pool, "no restart markers") for a in sys.argv[1:] or [256, 1024, 4096]] + [(int(a), pool_rst, "one restart interval per MCU row") for a in sys.argv[1:] or [256, 1024]]:
    files = [src[i % 4] for i in range(n)]
    ctx.jpeg_decode_batch(files[:64], download=False)[0]["free"]()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        info, st = ctx.jpeg_decode_batch(files, download=False)
        dt = time.perf_counter() - t0
        info["free"]()
        best = min(best, dt)
    assert not any(st)
    print("GPU decode, %s, batch of %5d: %.1f ms = %.0f frames/s (parse + pack + upload + Huffman + IDCT, planes left in HBM)" % (label, n, best * 1e3, n / best))
t0 = time.perf_counter()
for i in range(4):
    ref = oracle.jpeg_decode(pool[i])
dt = (time.perf_counter() - t0) / 4
t0 = time.perf_counter()
for i in range(8):
    im = Image.open(io.BytesIO(pool[i % 4]))
    im.draft("YCbCr", (w, h))
    im.load()
dp = (time.perf_counter() - t0) / 8
print("one host thread: oracle (Go's decoder restated) %.1f ms per frame = %.0f frames/s; libjpeg-turbo (Pillow) %.1f ms = %.0f frames/s" % (dt * 1e3, 1 / dt, dp * 1e3, 1 / dp))

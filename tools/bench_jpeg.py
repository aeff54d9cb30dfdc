#!/usr/bin/env python3
"""jpeg.Encode throughput: the transform kernel (frames and coefficients resident in HBM) and the whole batch
encode (transform + coefficient download + host entropy coding), next to the CPU oracle (Go's writer restated).
usage: tools/bench_jpeg.py [frames] [width height]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import imageprocessor_amd as ipx  # noqa: E402
import oracle  # noqa: E402
from helpers import rgba_frames  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
ctx = ipx.Context()
# photographs, not noise: a smooth base with mild texture so that the entropy coder sees realistic run lengths
yy, xx = np.mgrid[0:h, 0:w]
pool = []
for k in range(4):
    base = np.stack([np.sin(xx / (40.0 + 7 * k)) * 90 + 128, np.cos(yy / (31.0 + 5 * k)) * 90 + 128, ((xx + 2 * yy) / 6.0 + 40 * k) % 256], -1)
    tex = np.random.default_rng(k).normal(0, 6, (h, w, 3))
    f = np.concatenate([(base + tex).clip(0, 255), np.full((h, w, 1), 255.0)], -1).astype(np.uint8)
    pool.append(f)
fb = w * h * 4
src = ctx.alloc(n * fb)
for i in range(n):
    src.upload(pool[i % 4], offset=i * fb)
per = ipx.lib().ipx_jpeg_coef_count(w, h) * 2
coefs = ctx.alloc(n * per)


def step():
    ctx.jpeg_fdct_dev(src.ptr, w, h, n, coefs.ptr, 85)


for _ in range(3):
    step()
ctx.device_sync()
ms = min(ctx.timed(step) for _ in range(10))
alg = n * (fb + per)
print("jpeg transform kernel: %d x %dx%d in %.3f ms = %.0f frames/s; %.1f GB/s (4 B/px in + 3 B/px out)" % (n, w, h, ms, n / ms * 1e3, alg / ms / 1e6))
for mode, label in (("0", "entropy coding on the GPU"), ("1", "entropy coding on all host threads")):
    os.environ["IPX_JPEG_HOST_ENTROPY"] = mode
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        outs, release = ctx.jpeg_encode_batch_dev(src.ptr, w, h, n, 85, copy=False)
        best = min(best, time.perf_counter() - t0)
        sizes = [len(o) for o in outs]
        last = bytes(outs[min(n, 8) - 1])
        release()
    print("batch encode from HBM to host streams, %s: %d frames in %.1f ms = %.0f frames/s; %.0f KB per stream"
          % (label, n, best * 1e3, n / best, sum(sizes) / n / 1e3))
os.environ["IPX_JPEG_HOST_ENTROPY"] = "0"
m = min(n, 8)
t0 = time.perf_counter()
for i in range(m):
    ref = oracle.jpeg_encode_rgba(pool[i % 4], 85)
dt = time.perf_counter() - t0
print("CPU oracle (Go's writer restated, one thread): %.1f ms per frame = %.1f frames/s; stream equal: %s" % (dt / m * 1e3, m / dt, ref == last))

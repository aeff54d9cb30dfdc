#!/usr/bin/env python3
"""What leaving png.Encode on the host costs (DESIGN.md section 7, N3's PNG row): one thread, one 1920x1080 RGB frame,
(a) libpng through Pillow at compress_level 6 (adaptive filtering), (b) zlib level 6 alone on rows filtered with the heuristic
png.Encode uses (per row the filter with the smallest sum of absolute values; Go's compress/flate at its default level does the same
class of work as zlib level 6: hash chains + lazy matching).  Neither emits Go's bytes -- that is the point of the row staying in Go."""
import io
import sys
import time
import zlib

import numpy as np
from PIL import Image

sh, sw = 1080, 1920
yy, xx = np.mgrid[0:sh, 0:sw]
for name, img in (("photograph-like", (np.stack([np.sin(xx / 40.0) * 90 + 128, np.cos(yy / 31.0) * 90 + 128, ((xx + 2 * yy) / 6.0) % 256], -1)
                                       + np.random.default_rng(0).normal(0, 6, (sh, sw, 3))).clip(0, 255).astype(np.uint8)),
                  ("flat graphics", ((xx // 64 + yy // 64) % 5 * 50).astype(np.uint8)[..., None].repeat(3, -1))):
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 3:
        buf = io.BytesIO()
        Image.fromarray(img).save(buf, "PNG", compress_level=6)
        n += 1
    t_pil = (time.perf_counter() - t0) / n
    # the filter heuristic on whole rows (numpy), then zlib alone
    a = img.reshape(sh, sw * 3).astype(np.int16)
    left = np.concatenate([np.zeros((sh, 3), np.int16), a[:, :-3]], 1)
    up = np.concatenate([np.zeros((1, sw * 3), np.int16), a[:-1]], 0)
    ul = np.concatenate([np.zeros((1, sw * 3), np.int16), left[:-1]], 0)
    p = left + up - ul
    pa, pb, pc = np.abs(p - left), np.abs(p - up), np.abs(p - ul)
    paeth = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, ul))
    cands = [a, a - left, a - up, a - (left + up) // 2, a - paeth]
    cost = np.stack([np.abs(((c + 128) & 255) - 128).sum(1) for c in cands])
    best = cost.argmin(0)
    rows = np.stack([(c & 255).astype(np.uint8) for c in cands])[best, np.arange(sh)]
    raw = np.concatenate([best.astype(np.uint8)[:, None], rows], 1).tobytes()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 3:
        z = zlib.compress(raw, 6)
        n += 1
    t_z = (time.perf_counter() - t0) / n
    print("%-16s 1920x1080 RGB: libpng level 6 %.0f ms per frame and thread (%.1f frames/s, %d KB); zlib level 6 alone on the filtered rows %.0f ms (%d KB)"
          % (name, t_pil * 1e3, 1 / t_pil, len(buf.getvalue()) // 1024, t_z * 1e3, len(z) // 1024))

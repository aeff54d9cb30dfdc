#!/usr/bin/env python3
"""The micro-batcher under load (ipx_batcher_*): S submitter threads, each with ONE file in flight at a time as a goroutine of the
reference has (internal/worker/worker.go:112-149), 1080p 4:2:0 q85 uploads, resize 1024x576 + thumbnail 200 + watermark, three JPEG
streams back.  Prints images/s and the p50 / p99 latency of a file (submit -> its objects), and how the batcher grouped the files.
usage: tools/bench_batcher.py [files per submitter] [max_batch] [max_wait_us] [submitters ...]"""
import io
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from PIL import Image  # noqa: E402

import imageprocessor_amd as ipx  # noqa: E402
from helpers import DEFAULT_COL, text_glyphs  # noqa: E402

per = int(sys.argv[1]) if len(sys.argv) > 1 else 64
max_batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
max_wait = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
subs = [int(v) for v in sys.argv[4:]] or [3, 16, 64]
sw, sh = 1920, 1080
yy, xx = np.mgrid[0:sh, 0:sw]
files = []
for k in range(4):
    base = np.stack([np.sin(xx / (40.0 + 7 * k)) * 90 + 128, np.cos(yy / (31.0 + 5 * k)) * 90 + 128, ((xx + 2 * yy) / 6.0 + 40 * k) % 256], -1)
    img = (base + np.random.default_rng(k).normal(0, 6, (sh, sw, 3))).clip(0, 255).astype(np.uint8)
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, "JPEG", quality=85)
    files.append(buf.getvalue())
glyphs = text_glyphs(sw, sh)
ops = dict(resize=(1024, 768, True), thumbnail=(200, True), glyphs=glyphs, col=DEFAULT_COL)
with ipx.Pool(devices=(0,)) as pool, ipx.Batcher(pool, max_batch=max_batch, max_wait_us=max_wait, quality=85) as b:
    for t in [b.submit(files[i % 4], sw, sh, **ops) for i in range(32)]:      # plans, glyph set, lanes warm
        b.wait(t)
    for S in subs:
        lat = []
        mu = threading.Lock()
        before = b.stats()

        def work(k):
            mine = []
            for i in range(per):
                t0 = time.perf_counter()
                st, out = b.wait(b.submit(files[(k + i) % 4], sw, sh, **ops))
                mine.append(time.perf_counter() - t0)
                assert st == 0 and out["resize"]
            with mu:
                lat.extend(mine)
        ts = [threading.Thread(target=work, args=(k,)) for k in range(S)]
        t0 = time.perf_counter()
        [t.start() for t in ts]
        [t.join() for t in ts]
        dt = time.perf_counter() - t0
        lat.sort()
        st = b.stats()
        nb = st["batches"] - before["batches"]
        print("%3d submitters x %d files: %7.0f images/s; latency p50 %.2f ms, p99 %.2f ms; %d batches (mean %.1f files; %d by size, %d by timer, %d when idle)"
              % (S, per, S * per / dt, lat[len(lat) // 2] * 1e3, lat[min(len(lat) - 1, int(len(lat) * 0.99))] * 1e3, nb, S * per / max(1, nb),
                 st["flushed_by_size"] - before["flushed_by_size"], st["flushed_by_timer"] - before["flushed_by_timer"],
                 st["flushed_when_idle"] - before["flushed_when_idle"]), flush=True)

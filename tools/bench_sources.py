#!/usr/bin/env python3
"""Throughput of the batch paths for the other source types image.Decode returns, frames resident in HBM, 1920x1080,
resize 1024x768 + thumbnail 200 + watermark, through the one-pass kernel (ks_fused_kernel) and, with IPX_FUSED=0, the
per-output kernels:
  *image.NRGBA     ipx_plan_run_dev_nrgba       *image.Gray  ipx_plan_run_dev_gray       *image.Paletted  ipx_plan_run_dev_paletted (expansion + NRGBA)
  NRGBA64 / Gray16 / CMYK  ipx_plan_run_dev_deep (expansion to 16-bit taps, then the kernel on them)
usage: tools/bench_sources.py [frames]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import imageprocessor_amd as ipx  # noqa: E402
from helpers import DEFAULT_COL, text_glyphs  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
w, h = 1920, 1080
ctx = ipx.Context()
gs = ctx.glyphset(text_glyphs(w, h), DEFAULT_COL)
plan = ctx.plan(w, h, resize=(1024, 768, True), thumbnail=(200, True), watermark=gs)
i = plan.info
rng = np.random.default_rng(3)
pool = 4
res, th, wm = ctx.alloc(n * i.resize_bytes), ctx.alloc(n * i.thumb_bytes), ctx.alloc(n * i.wm_bytes)
outs = i.resize_bytes + i.thumb_bytes + i.wm_bytes


def timed(step, label, in_bytes):
    for _ in range(2):
        step()
    ctx.device_sync()
    ms = min(ctx.timed(step) for _ in range(6))
    alg = n * (in_bytes + outs)
    print("%-62s %4d frames in %7.3f ms = %7.0f images/s; algorithmic %6.1f GB/s (%.2f MB per frame)"
          % (label, n, ms, n / ms * 1e3, alg / ms / 1e6, alg / n / 1e6), flush=True)


nr = ctx.alloc(n * w * h * 4).upload(np.resize(rng.integers(0, 256, (pool, h, w, 4), dtype=np.uint8), (n, h, w, 4)))
paths = (("1", "one-pass kernel"), ("0", "per-output kernels"))
for fused, label in paths:
    os.environ["IPX_FUSED"] = fused
    timed(lambda: plan.run_dev_nrgba(n, nr.ptr, res.ptr, th.ptr, wm.ptr), "NRGBA, " + label, w * h * 4)
del nr
gr = ctx.alloc(n * w * h).upload(np.resize(rng.integers(0, 256, (pool, h, w), dtype=np.uint8), (n, h, w)))
for fused, label in paths:
    os.environ["IPX_FUSED"] = fused
    timed(lambda: plan.run_dev_gray(n, gr.ptr, w, w * h, res.ptr, th.ptr, wm.ptr), "Gray, " + label, w * h)
os.environ["IPX_FUSED"] = "1"
pal = rng.integers(0, 256, (n, 256, 4), dtype=np.uint8)
pal[..., 3] = 255
dp = ctx.alloc(pal.nbytes).upload(pal)
timed(lambda: plan.run_dev_paletted(n, gr.ptr, w, w * h, dp.ptr, res.ptr, th.ptr, wm.ptr), "Paletted, palette expansion + one-pass kernel (NRGBA)", w * h + 1024)
del gr
for name, kind, bpp in (("NRGBA64", ipx.DEEP_NRGBA64, 8), ("Gray16", ipx.DEEP_GRAY16, 2), ("CMYK", ipx.DEEP_CMYK, 4)):
    dsrc = ctx.alloc(n * w * h * bpp).upload(np.resize(rng.integers(0, 256, (pool, h, w * bpp), dtype=np.uint8), (n, h, w * bpp)))
    for fused, label in paths:
        os.environ["IPX_FUSED"] = fused
        timed(lambda: plan.run_dev_deep(n, kind, dsrc.ptr, w * bpp, w * h * bpp, res.ptr, th.ptr, wm.ptr), name + ", tap expansion + " + label, w * h * bpp)
    os.environ["IPX_FUSED"] = "1"
    del dsrc

#!/usr/bin/env python3
"""Latency of the per-operator seam (ipx_processor_process on one decoded frame: thumbnail + resize + watermark, pixels in and out), the
call INTEGRATION.md tells a maintainer to wire first -- alone, and while a host batch of the same context is running.
usage: tools/bench_seam.py [calls]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import imageprocessor_amd as ipx  # noqa: E402
from imageprocessor_amd import operations as ops  # noqa: E402
from helpers import rgba_frames, text_glyphs  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 200
w, h = 1920, 1080
src = rgba_frames(1, w, h, seed=7)[0]
font = ops.Font(lambda text, size: 300, lambda text, size, px, py, fw=None, fh=None: text_glyphs(w, h))
task = {"ID": "t", "ImageID": "i", "Format": "jpeg", "Operations": [
    {"Type": "thumbnail", "Parameters": {"size": 200.0, "crop_to_fit": True}},
    {"Type": "resize", "Parameters": {"width": 1024.0, "height": 768.0, "keep_aspect": True}},
    {"Type": "watermark", "Parameters": {"text": "x", "opacity": 0.5, "position": "bottom-right"}}]}
ctx = ipx.Context(device=0)
ip = ops.ImageProcessor(ctx, font)


def run(label, src=src):
    lat = []
    for _ in range(calls):
        t0 = time.perf_counter()
        res, err = ip.Process(task, src, "jpeg")
        lat.append((time.perf_counter() - t0) * 1e3)
        assert err is None
    lat.sort()
    print("%-44s median %.2f ms  p90 %.2f  max %.2f  (%d calls, one 1080p frame, three operators)" % (label, lat[len(lat) // 2], lat[int(len(lat) * 0.9)], lat[-1], calls), flush=True)


ip.Process(task, src, "jpeg")
run("single calls, idle context")
n = 256
frames = ctx.host_alloc((n, h, w, 4))
frames[:] = np.resize(rgba_frames(4, w, h, seed=5), frames.shape)
plan = ctx.plan(w, h, resize=(1024, 768, True), thumbnail=(200, True), watermark=True)
outs = {"resize": ctx.host_alloc((n, 576, 1024, 4)), "thumbnail": ctx.host_alloc((n, 200, 200, 4)), "watermark": ctx.host_alloc((n, h, w, 4))}
plan.run_host(frames, out=outs)
t0 = time.perf_counter()
plan.run_host(frames, out=outs)
print("one 256-frame host batch alone: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
state = {"go": True}


def batches():
    while state["go"]:
        plan.run_host(frames, out=outs)


th = threading.Thread(target=batches)
th.start()
time.sleep(0.05)
run("single calls while host batches run")
pinned_src = ctx.host_alloc((h, w, 4))
pinned_src[:] = src
run("... with the call's source frame in pinned memory", pinned_src)
state["go"] = False
th.join()

# the same beside batches that never touch the link (frames and outputs resident in HBM): what of the above is contention for CUs
dsrc = ctx.alloc(n * h * w * 4)
dsrc.upload(frames)
dres, dth, dwm = ctx.alloc(n * 576 * 1024 * 4), ctx.alloc(n * 200 * 200 * 4), ctx.alloc(n * h * w * 4)
st = ctx.stream()
state["go"] = True


def dev_batches():
    while state["go"]:
        plan.run_dev(n, dsrc.ptr, dres.ptr, dth.ptr, dwm.ptr, stream=st)
        ctx.sync(st)


th = threading.Thread(target=dev_batches)
th.start()
time.sleep(0.05)
run("single calls while device batches run")
state["go"] = False
th.join()

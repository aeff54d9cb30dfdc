#!/usr/bin/env python3
"""Random geometries through the one-pass kernel scaler (RGBA, NRGBA, Gray, Paletted, YCbCr at every subsampling ratio, the deep types), random tilings of it and the per-output kernels, against the
oracle's routines for the type: frame sizes (mostly multiples of 4: the fused kernels' domain), resize / thumbnail parameters, tile shapes.
usage: tools/fuzz_sources.py [trials] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import imageprocessor_amd as ipx  # noqa: E402
import oracle  # noqa: E402
from helpers import DEFAULT_COL, text_glyphs  # noqa: E402
from test_sources_gpu import _expect_ycbcr_ops  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
ctx = ipx.Context()
bad = 0


def expect(kind, src, resize, thumb, glyphs, w, h):
    """(resize, thumbnail, watermark) as the reference's helpers treat a source of this type"""
    if kind == "rgba":
        o = oracle.process(src, resize=resize, thumb=thumb, glyphs=glyphs, col=DEFAULT_COL)
        return o["resize"], o["thumbnail"], o["watermark"]
    if kind == "ycbcr":
        o = _expect_ycbcr_ops(src[0], src[1], src[2], src[3], resize, thumb, glyphs, DEFAULT_COL)
        return o["resize"], o["thumbnail"], o["watermark"]
    if kind == "gray":
        rgba = np.dstack([src] * 3 + [np.full_like(src, 255)])
        o = oracle.process(rgba, resize=resize, thumb=thumb, glyphs=glyphs, col=DEFAULT_COL)
        return o["resize"], o["thumbnail"], o["watermark"]
    if kind.startswith("deep"):
        pix, dk = src
        scale = lambda dw, dh, sr=None: oracle.scale_bilinear_deep(pix, dk, dw, dh, sr=sr)       # noqa: E731
        draw = lambda: oracle.draw_deep(np.zeros((h, w, 4), np.uint8), (0, 0, w, h), pix, dk)    # noqa: E731
    elif kind == "nrgba":
        scale = lambda dw, dh, sr=None: oracle.scale_bilinear_nrgba(src, dw, dh, sr=sr)          # noqa: E731
        draw = lambda: oracle.draw_nrgba(np.zeros((h, w, 4), np.uint8), (0, 0, w, h), src)       # noqa: E731
    else:
        idx, p16 = src
        scale = lambda dw, dh, sr=None: oracle.scale_bilinear_paletted(idx, p16, dw, dh, sr=sr)  # noqa: E731
        draw = lambda: oracle.draw_paletted(np.zeros((h, w, 4), np.uint8), (0, 0, w, h), idx, p16)  # noqa: E731
    nw, nh = oracle.resize_dims(w, h, *resize)
    crop, tw, th = oracle.thumb_geometry(w, h, *thumb)
    if thumb[1]:
        cs_w, cs_h = crop[2] - crop[0], crop[3] - crop[1]
        t = oracle.scale_bilinear(scale(cs_w, cs_h, sr=crop), tw, th)
    else:
        t = scale(tw, th)
    return scale(nw, nh), t, oracle.composite_glyphs(draw(), glyphs, DEFAULT_COL)


for trial in range(trials):
    w = int(rng.choice([4 * rng.integers(1, 40), 4 * rng.integers(40, 520), rng.integers(1, 300)]))
    h = int(rng.choice([rng.integers(1, 40), rng.integers(40, 600)]))
    resize = (int(rng.integers(1, 1300)), int(rng.integers(1, 900)), bool(rng.integers(0, 2)))
    thumb = (int(rng.integers(1, 300)), bool(rng.integers(0, 2)))
    nw, nh = oracle.resize_dims(w, h, *resize)
    _, tw0, th0 = oracle.thumb_geometry(w, h, *thumb)
    if nw < 1 or nh < 1 or tw0 < 1 or th0 < 1 or max(nw, nh, tw0, th0) > 65535:      # (beyond 65535 pixels a side: IPX_ERR_UNSUPPORTED by design)
        continue
    knobs = ("IPX_KS_STRIPS", "IPX_KS_SPLIT_ROWS", "IPX_KS_SPLIT", "IPX_KS_SPEC", "IPX_FUSED", "IPX_KS_FAST", "IPX_KS_FIX_CAP")
    for k in knobs:
        os.environ.pop(k, None)
    # the arithmetic route: the float pass (with lists that overflow now and then), or float64 throughout
    pick = rng.random()
    if pick < 0.6:
        if rng.random() < 0.3:
            os.environ["IPX_KS_FIX_CAP"] = str(int(rng.choice([1, 8, 100, 1000])))
    elif pick < 0.8:
        os.environ["IPX_KS_FAST"] = "0"
    if rng.random() < 0.4:
        os.environ["IPX_KS_STRIPS"] = str(int(rng.choice([1, 2, 3, 5, 9])))          # tilings of the one-pass kernel (read when the plan is made)
    if rng.random() < 0.4:
        os.environ["IPX_KS_SPLIT_ROWS"] = str(int(rng.choice([4, 9, 17, 64, 200])))
    if rng.random() < 0.5:
        os.environ["IPX_KS_SPLIT"] = str(int(rng.integers(0, 2)))                     # one segment per frame / many
    if rng.random() < 0.2:
        os.environ["IPX_KS_SPEC"] = "0"                                               # the general four-channel kernel alone
    if rng.random() < 0.1:
        os.environ["IPX_FUSED"] = "0"                                                 # per-output kernels
    kind = ["nrgba", "gray", "paletted", "ycbcr", "rgba", "deep"][trial % 6]
    n = int(rng.integers(1, 4))
    glyphs = text_glyphs(w, h, n=5, width_px=min(60, w), height_px=min(20, h))
    gs = ctx.glyphset(glyphs, DEFAULT_COL)
    plan = ctx.plan(w, h, resize=resize, thumbnail=thumb, watermark=gs)
    if kind == "deep":
        dk = int(rng.integers(0, 4))
        if dk == oracle.DEEP_GRAY16:
            vals = rng.integers(0, 65536, (n, h, w), dtype=np.uint16)
        elif dk == oracle.DEEP_CMYK:
            vals = rng.integers(0, 256, (n, h, w, 4), dtype=np.uint8)
        else:
            vals = rng.integers(0, 65536, (n, h, w, 4), dtype=np.uint16)
            if rng.random() < 0.3:
                vals[..., 3] = 0xffff
            if dk == oracle.DEEP_RGBA64:
                vals[..., :3] = np.minimum(vals[..., :3], vals[..., 3:])
        pix = np.stack([oracle.deep_pix(vals[i], dk) for i in range(n)])
        got = plan.run_host_deep(pix, dk)
        srcs = [(pix[i], dk) for i in range(n)]
    elif kind == "nrgba":
        frames = rng.integers(0, 256, (n, h, w, 4), dtype=np.uint8)
        if rng.random() < 0.3:
            frames[..., 3] = 255
        got = plan.run_host_nrgba(frames)
        srcs = [frames[i] for i in range(n)]
    elif kind == "gray":
        frames = rng.integers(0, 256, (n, h, w), dtype=np.uint8)
        got = plan.run_host_gray(frames)
        srcs = [frames[i] for i in range(n)]
    elif kind == "rgba":
        frames = rng.integers(0, 256, (n, h, w, 4), dtype=np.uint8)
        if rng.random() < 0.7:
            frames[..., :3] = np.minimum(frames[..., :3], frames[..., 3:])      # premultiplied, as image.RGBA holds it (else: colours above alpha, clamped by the crop copy)
        if rng.random() < 0.5:
            frames[..., 3] = 255
        got = plan.run_host(frames)
        srcs = [frames[i] for i in range(n)]
    elif kind == "ycbcr":
        ratio = int(rng.integers(0, 4))
        chh, cww = oracle.chroma_shape(w, h, ratio)
        yp = rng.integers(0, 256, (n, h, w), dtype=np.uint8)
        cbp, crp = rng.integers(0, 256, (n, chh, cww), dtype=np.uint8), rng.integers(0, 256, (n, chh, cww), dtype=np.uint8)
        got = plan.run_host_ycbcr(yp, cbp, crp, ratio)
        srcs = [(yp[i], cbp[i], crp[i], ratio) for i in range(n)]
    else:
        idx = rng.integers(0, 256, (n, h, w), dtype=np.uint8)
        pal = rng.integers(0, 256, (n, 256, 4), dtype=np.uint8)
        pal[:, : int(rng.integers(0, 257)), 3] = 255
        got = plan.run_host_paletted(idx, pal)
        srcs = [(idx[i], oracle.palette16(pal[i], "nrgba")) for i in range(n)]
    for i in range(n):
        want = expect(kind, srcs[i], resize, thumb, glyphs, w, h)
        for key, wv in zip(("resize", "thumbnail", "watermark"), want):
            if key in got and not np.array_equal(got[key][i], wv):
                bad += 1
                print("MISMATCH trial", trial, kind, key, "frame", i, "%dx%d" % (w, h), "resize", resize, "thumb", thumb,
                      {e: os.environ.get(e) for e in knobs}, flush=True)
                break
    plan.close()
    gs.close()
    if trial % 20 == 19:
        print("...", trial + 1, "trials", bad, "mismatches", flush=True)
print("done:", trials, "trials", bad, "mismatches")

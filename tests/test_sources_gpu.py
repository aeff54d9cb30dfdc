"""Source-type variants: *image.NRGBA (PNG with alpha) and *image.YCbCr (JPEG)
sources through the per-operation seam, HIP vs the CPU oracle and the committed known answers.
Bit-exact.  PARITY UNPINNED against Go itself (oracle/ipx_oracle.h)."""
import json
import os

import numpy as np
import pytest

import oracle
from helpers import rgba_frames

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "kats.json")) as f:
    CASES = json.load(f)["cases"]


@pytest.fixture(scope="module")
def ctx():
    import imageprocessor_amd as m
    c = m.Context(lanes=2)
    yield c
    c.close()


def _frame(flat, w, h):
    return np.array(flat, np.uint8).reshape(h, w, 4)


def _planes(img):
    w, h, ratio = img["w"], img["h"], img["ratio"]
    chh, cw = oracle.chroma_shape(w, h, ratio)
    return (np.array(img["y"], np.uint8).reshape(h, w), np.array(img["cb"], np.uint8).reshape(chh, cw),
            np.array(img["cr"], np.uint8).reshape(chh, cw), ratio)


def test_golden_nrgba_and_ycbcr(ctx):
    n = 0
    for c in CASES:
        k = c["kind"]
        if k not in ("draw_nrgba", "scale_nrgba", "draw_ycbcr", "scale_ycbcr"):
            continue
        dst = _frame(c["dst"], c["dw"], c["dh"]).copy()
        if k == "draw_nrgba":
            ctx.draw_nrgba(dst, c["r"], _frame(c["src"], c["sw"], c["sh"]), c["sp"], c["op"])
        elif k == "scale_nrgba":
            ctx.scale_bilinear_nrgba(_frame(c["src"], c["sw"], c["sh"]), c["dw"], c["dh"], sr=c["sr"], dr=c["dr"],
                                     op=c["op"], dst=dst)
        elif k == "draw_ycbcr":
            ctx.draw_ycbcr(dst, c["r"], *_planes(c["img"]), c["sp"])
        else:
            ctx.scale_bilinear_ycbcr(*_planes(c["img"]), c["dw"], c["dh"], sr=c["sr"], dr=c["dr"], dst=dst)
        np.testing.assert_array_equal(dst, _frame(c["expect"], c["dw"], c["dh"]), err_msg=c["name"])
        n += 1
    assert n >= 25


def _rand_ycbcr(w, h, ratio, seed):
    rng = np.random.default_rng(seed)
    chh, cw = oracle.chroma_shape(w, h, ratio)
    return (rng.integers(0, 256, (h, w), dtype=np.uint8), rng.integers(0, 256, (chh, cw), dtype=np.uint8),
            rng.integers(0, 256, (chh, cw), dtype=np.uint8))


@pytest.mark.parametrize("ratio", [0, 1, 2, 3], ids=["444", "422", "420", "440"])
def test_ycbcr_reference_shapes(ctx, ratio):
    """A decoded 1080p JPEG through the three helpers: resize 1024x576, the thumbnail's crop copy +
    200x200 scale (thumbnail.go:128-131), and the watermark's full-frame draw.Draw (watermark.go:92)."""
    w, h = 1920, 1080
    y, cb, cr = _rand_ycbcr(w, h, ratio, 100 + ratio)
    np.testing.assert_array_equal(ctx.scale_bilinear_ycbcr(y, cb, cr, ratio, 1024, 576),
                                  oracle.scale_bilinear_ycbcr(y, cb, cr, ratio, 1024, 576))
    crop = (420, 0, 1500, 1080)
    cropped = ctx.scale_bilinear_ycbcr(y, cb, cr, ratio, 1080, 1080, sr=crop)          # equal size: Copy -> DrawYCbCr
    np.testing.assert_array_equal(cropped, oracle.scale_bilinear_ycbcr(y, cb, cr, ratio, 1080, 1080, sr=crop))
    np.testing.assert_array_equal(ctx.scale_bilinear(cropped, 200, 200), oracle.scale_bilinear(cropped, 200, 200))
    full = np.zeros((h, w, 4), np.uint8)
    np.testing.assert_array_equal(ctx.draw_ycbcr(full, (0, 0, w, h), y, cb, cr, ratio),
                                  oracle.draw_ycbcr(np.zeros((h, w, 4), np.uint8), (0, 0, w, h), y, cb, cr, ratio))
    # odd sizes and an odd source origin (x/2, y/2 chroma indexing), upscale
    y, cb, cr = _rand_ycbcr(333, 251, ratio, 7)
    for dw, dh, sr in ((100, 90, None), (640, 480, None), (50, 50, (13, 7, 320, 240))):
        np.testing.assert_array_equal(ctx.scale_bilinear_ycbcr(y, cb, cr, ratio, dw, dh, sr=sr),
                                      oracle.scale_bilinear_ycbcr(y, cb, cr, ratio, dw, dh, sr=sr))


def test_nrgba_vs_oracle(ctx):
    src = rgba_frames(1, 640, 360, seed=5, opaque=False, premul=False)[0]      # straight alpha, as png.Decode gives
    for dw, dh, sr, op, used in ((1024, 576, None, 0, False), (200, 200, (140, 0, 500, 360), 0, False),
                                 (97, 33, None, 0, True), (97, 33, None, 1, True), (640, 360, None, 0, True)):
        dst0 = rgba_frames(1, dw, dh, seed=9, opaque=False)[0] if used else np.zeros((dh, dw, 4), np.uint8)
        np.testing.assert_array_equal(ctx.scale_bilinear_nrgba(src, dw, dh, sr=sr, op=op, dst=dst0.copy()),
                                      oracle.scale_bilinear_nrgba(src, dw, dh, sr=sr, op=op, dst=dst0.copy()))
    opaque = src.copy()
    opaque[..., 3] = 255                                                        # Over + opaque source => Src
    dst0 = rgba_frames(1, 50, 40, seed=2, opaque=False)[0]
    np.testing.assert_array_equal(ctx.scale_bilinear_nrgba(opaque, 50, 40, dst=dst0.copy()),
                                  oracle.scale_bilinear_nrgba(opaque, 50, 40, dst=dst0.copy()))
    for op in (0, 1):                                                           # the watermark's draw.Draw / a Copy
        d0 = rgba_frames(1, 640, 360, seed=4, opaque=False)[0]
        np.testing.assert_array_equal(ctx.draw_nrgba(d0.copy(), (5, 5, 700, 400), src, (2, 3), op),
                                      oracle.draw_nrgba(d0.copy(), (5, 5, 700, 400), src, (2, 3), op))

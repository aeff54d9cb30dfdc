"""GPU: ipx_plan_run_dev_deep / ipx_plan_run_host_deep -- batches of *image.NRGBA64, *image.RGBA64, *image.Gray16 (16-bit PNGs) and
*image.CMYK (four-component JPEGs) frames, the image types image.Decode (image_processor.go:47) returns besides the ones the
other entries take.  Per operator as the reference's helpers treat the type: resizeImage (resize.go:121-125) interpolates
src.At(x, y).RGBA() at 16 bits (scale_RGBA_Image_*), the crop copy (thumbnail.go:128-130) and draw.Draw (watermark.go:92) keep the top
byte (drawRGBA / drawCMYK).  Bit-exact against the oracle's restatement of those routines (tests/test_deep_oracle.py pins it);
the fused converted-tile kernel (reading Go's Pix, or the expanded taps) and the three-kernel path."""
import numpy as np
import pytest

import imageprocessor_amd as ipa
import oracle
from oracle import DEEP_CMYK, DEEP_GRAY16, DEEP_NRGBA64, DEEP_RGBA64

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = ipa.Context()
    yield c
    c.close()


def _frames(kind, n, h, w, seed):
    rng = np.random.default_rng(seed)
    if kind == DEEP_GRAY16:
        v = rng.integers(0, 65536, (n, h, w), dtype=np.uint16)
    elif kind == DEEP_CMYK:
        v = rng.integers(0, 256, (n, h, w, 4), dtype=np.uint8)
    else:
        v = rng.integers(0, 65536, (n, h, w, 4), dtype=np.uint16)
        v[0, :, : w // 2, 3] = 0xffff
        v[0, 0, :8, 3] = [0, 1, 2, 0xff, 0x100, 0xff00, 0xfffe, 0xffff]
        if kind == DEEP_RGBA64:
            v[..., :3] = np.minimum(v[..., :3], v[..., 3:4])                 # premultiplied, as the PNG decoder makes them
            v[n - 1] |= 0                                                    # (kept: the last frame too)
    return np.stack([oracle.deep_pix(v[k], kind) for k in range(n)])


def _expect(pix, kind, w, h, resize, thumb, glyphs, col):
    nw, nh = oracle.resize_dims(w, h, *resize)
    want_r = oracle.scale_bilinear_deep(pix, kind, nw, nh)
    crop, tw, thh = oracle.thumb_geometry(w, h, *thumb)
    if thumb[1]:
        cs = crop[2] - crop[0]
        cropped = oracle.scale_bilinear_deep(pix, kind, cs, cs, sr=crop)        # equal sizes: Copy -> drawRGBA, Over onto zeros
        want_t = oracle.scale_bilinear(cropped, tw, thh)
    else:
        want_t = oracle.scale_bilinear_deep(pix, kind, tw, thh)
    want_w = oracle.composite_glyphs(oracle.draw_deep(np.zeros((h, w, 4), np.uint8), (0, 0, w, h), pix, kind), glyphs, col)
    return want_r, want_t, want_w


@pytest.mark.parametrize("kind", [DEEP_NRGBA64, DEEP_RGBA64, DEEP_GRAY16, DEEP_CMYK], ids=["nrgba64", "rgba64", "gray16", "cmyk"])
@pytest.mark.parametrize("case", [(640, 360, 3, (1024, 768, True), (200, True)), (333, 251, 2, (200, 100, False), (64, False)),
                                  (200, 200, 2, (200, 200, False), (100, True)), (1280, 720, 2, (500, 333, False), (200, True))],
                         ids=lambda c: "%dx%d" % (c[0], c[1]))
@pytest.mark.parametrize("path", ["taps", "split", "three", "f64", "cap"], ids=["one-pass", "one-pass-split-strips", "per-output", "one-pass-float64", "one-pass-short-lists"])
def test_deep_batch_plan(ctx, kind, case, path, monkeypatch):
    """The three ways a batch can go: the converted-tile kernel reading Go's Pix (16-byte aligned frames, widths that are multiples of
    4), the same kernel on the expanded frames of taps, and expansion + the three-kernel path (any shape)."""
    from helpers import DEFAULT_COL, text_glyphs
    monkeypatch.setenv("IPX_FUSED", "0" if path == "three" else "1")
    if path == "split":
        monkeypatch.setenv("IPX_KS_STRIPS", "2"); monkeypatch.setenv("IPX_KS_SPLIT", "1"); monkeypatch.setenv("IPX_KS_SPLIT_ROWS", "29")
    if path == "f64":
        monkeypatch.setenv("IPX_KS_FAST", "0")          # no float pass (NRGBA64, Gray16 and CMYK take it by default; RGBA64 never)
    if path == "cap":
        monkeypatch.setenv("IPX_KS_FIX_CAP", "13")      # the float pass's lists fill up: the frames' items are redone in float64
    w, h, n, resize, thumb = case
    pix = _frames(kind, n, h, w, seed=w + kind)
    glyphs = text_glyphs(w, h, n=6, width_px=min(150, w), height_px=min(30, h))
    gs = ctx.glyphset(glyphs, DEFAULT_COL)
    plan = ctx.plan(w, h, resize=resize, thumbnail=thumb, watermark=gs)
    i = plan.info
    src = ctx.alloc(pix.nbytes).upload(pix)
    res, th, wm = ctx.alloc(n * i.resize_bytes), ctx.alloc(n * i.thumb_bytes), ctx.alloc(n * i.wm_bytes)
    plan.run_dev_deep(n, kind, src.ptr, pix.shape[2], pix.shape[1] * pix.shape[2], res.ptr, th.ptr, wm.ptr)
    ctx.sync()
    got_r, got_t, got_w = res.download((n, i.resize_h, i.resize_w, 4)), th.download((n, i.thumb_h, i.thumb_w, 4)), wm.download((n, h, w, 4))
    for k in range(n):
        want_r, want_t, want_w = _expect(pix[k], kind, w, h, resize, thumb, glyphs, DEFAULT_COL)
        np.testing.assert_array_equal(got_r[k], want_r, err_msg="resize %d" % k)
        np.testing.assert_array_equal(got_t[k], want_t, err_msg="thumbnail %d" % k)
        np.testing.assert_array_equal(got_w[k], want_w, err_msg="watermark %d" % k)
    plan.close()
    gs.close()


def test_deep_1080p_and_single_outputs(ctx):
    """The bench geometry (1920x1080 -> 1024x768, thumbnail 200, watermark) on NRGBA64 frames, and plans that want one output only."""
    from helpers import DEFAULT_COL, text_glyphs
    w, h, n = 1920, 1080, 2
    pix = _frames(DEEP_NRGBA64, n, h, w, seed=9)
    glyphs = text_glyphs(w, h, n=8)
    gs = ctx.glyphset(glyphs, DEFAULT_COL)
    src = ctx.alloc(pix.nbytes).upload(pix)
    want = [_expect(pix[k], DEEP_NRGBA64, w, h, (1024, 768, False), (200, True), glyphs, DEFAULT_COL) for k in range(n)]
    for kw in ({"resize": (1024, 768, False), "thumbnail": (200, True), "watermark": gs}, {"resize": (1024, 768, False)}, {"thumbnail": (200, True)},
               {"watermark": gs}):
        plan = ctx.plan(w, h, **kw)
        i = plan.info
        res, th, wm = ctx.alloc(max(1, n * i.resize_bytes)), ctx.alloc(max(1, n * i.thumb_bytes)), ctx.alloc(max(1, n * i.wm_bytes))
        plan.run_dev_deep(n, DEEP_NRGBA64, src.ptr, pix.shape[2], pix.shape[1] * pix.shape[2], res.ptr if "resize" in kw else None,
                          th.ptr if "thumbnail" in kw else None, wm.ptr if "watermark" in kw else None)
        ctx.sync()
        for k in range(n):
            if "resize" in kw:
                np.testing.assert_array_equal(res.download((n, i.resize_h, i.resize_w, 4))[k], want[k][0])
            if "thumbnail" in kw:
                np.testing.assert_array_equal(th.download((n, i.thumb_h, i.thumb_w, 4))[k], want[k][1])
            if "watermark" in kw:
                np.testing.assert_array_equal(wm.download((n, h, w, 4))[k], want[k][2])
        plan.close()
    gs.close()


def test_deep_host_frames_and_bad_arguments(ctx, monkeypatch):
    """ipx_plan_run_host_deep: frames in host memory chunked over the lanes (several chunks, a ragged last one); argument errors."""
    from helpers import DEFAULT_COL, text_glyphs
    monkeypatch.setenv("IPX_HOST_CHUNK", "2")
    w, h, n, resize, thumb = 320, 200, 5, (200, 120, False), (64, True)
    glyphs = text_glyphs(w, h, n=6, width_px=150, height_px=30)
    gs = ctx.glyphset(glyphs, DEFAULT_COL)
    plan = ctx.plan(w, h, resize=resize, thumbnail=thumb, watermark=gs)
    for kind in (DEEP_NRGBA64, DEEP_GRAY16, DEEP_CMYK):
        pix = _frames(kind, n, h, w, seed=40 + kind)
        out = plan.run_host_deep(pix, kind)
        for k in range(n):
            want_r, want_t, want_w = _expect(pix[k], kind, w, h, resize, thumb, glyphs, DEFAULT_COL)
            np.testing.assert_array_equal(out["resize"][k], want_r)
            np.testing.assert_array_equal(out["thumbnail"][k], want_t)
            np.testing.assert_array_equal(out["watermark"][k], want_w)
    pix = _frames(DEEP_GRAY16, 1, h, w, seed=1)
    with pytest.raises(ipa.IpxError):
        plan.run_host_deep(pix, 7)                                                   # unknown type
    with pytest.raises(ipa.IpxError):
        plan.run_host_deep(pix[:, :, : w], DEEP_GRAY16)                              # rows shorter than the plan's width
    plan.close()
    gs.close()


@pytest.mark.parametrize("kind", [DEEP_NRGBA64, DEEP_RGBA64, DEEP_GRAY16, DEEP_CMYK], ids=["nrgba64", "rgba64", "gray16", "cmyk"])
def test_deep_per_operation_seam(ctx, kind):
    """ipx_scale_bilinear_deep / ipx_draw_deep: any rectangles, Src and Over onto a frame that already holds pixels (Over becomes Src
    only when the source's Opaque() holds), odd sizes, sub-rectangles, equal sizes (Copy)."""
    rng = np.random.default_rng(70 + kind)
    w, h = 157, 93
    pix = _frames(kind, 2, h, w, seed=5 + kind)
    under = rng.integers(0, 256, (64, 80, 4), dtype=np.uint8)
    under[..., :3] = np.minimum(under[..., :3], under[..., 3:4])
    for k in range(2):                                   # frame 0 of the alpha types is half opaque, frame 1 not at all
        for op in (oracle.OP_SRC, oracle.OP_OVER):
            for dw, dh, sr, dr in ((80, 64, None, None), (80, 64, (10, 5, 150, 90), (4, 3, 70, 60)), (80, 64, (20, 10, 60, 50), (0, 0, 40, 40)), (80, 64, None, (-10, -6, 120, 80))):
                want = oracle.scale_bilinear_deep(pix[k], kind, dw, dh, sr=sr, dr=dr, op=op, dst=under.copy())
                got = ctx.scale_bilinear_deep(pix[k], kind, dw, dh, sr=sr, dr=dr, op=op, dst=under.copy())
                np.testing.assert_array_equal(got, want, err_msg="scale op %d %r %r" % (op, sr, dr))
            want = oracle.draw_deep(under.copy(), (5, 7, 75, 60), pix[k], kind, sp=(3, 2), op=op)
            got = ctx.draw_deep(under.copy(), (5, 7, 75, 60), pix[k], kind, sp=(3, 2), op=op)
            np.testing.assert_array_equal(got, want, err_msg="draw op %d" % op)
    if kind in (DEEP_NRGBA64, DEEP_RGBA64):                 # a fully opaque frame: Over == Src through Opaque()
        v = rng.integers(0, 65536, (h, w, 4), dtype=np.uint16)
        v[..., 3] = 0xffff
        op_pix = oracle.deep_pix(v, kind)
        want = oracle.scale_bilinear_deep(op_pix, kind, 80, 64, op=oracle.OP_OVER, dst=under.copy())
        np.testing.assert_array_equal(ctx.scale_bilinear_deep(op_pix, kind, 80, 64, op=oracle.OP_OVER, dst=under.copy()), want)
    with pytest.raises(ipa.IpxError):
        ctx.scale_bilinear_deep(pix[0], 9, 10, 10)


def test_gray16_rows_that_are_8_byte_aligned_only(ctx):
    """636 pixels of Gray16 are 1272 bytes: rows (and, with 251 rows, frames) on 8-byte but not 16-byte boundaries -- the fused
    kernel's chunk loads for this type are 8 bytes, so such frames still take it."""
    from helpers import DEFAULT_COL, text_glyphs
    w, h, n = 636, 251, 3
    pix = _frames(DEEP_GRAY16, n, h, w, seed=3)
    glyphs = text_glyphs(w, h, n=6, width_px=150, height_px=30)
    gs = ctx.glyphset(glyphs, DEFAULT_COL)
    plan = ctx.plan(w, h, resize=(300, 200, False), thumbnail=(100, True), watermark=gs)
    out = plan.run_host_deep(pix, DEEP_GRAY16)
    i = plan.info
    src = ctx.alloc(pix.nbytes).upload(pix)
    res, th, wm = ctx.alloc(n * i.resize_bytes), ctx.alloc(n * i.thumb_bytes), ctx.alloc(n * i.wm_bytes)
    plan.run_dev_deep(n, DEEP_GRAY16, src.ptr, pix.shape[2], pix.shape[1] * pix.shape[2], res.ptr, th.ptr, wm.ptr)
    ctx.sync()
    got_r, got_t, got_w = res.download((n, i.resize_h, i.resize_w, 4)), th.download((n, i.thumb_h, i.thumb_w, 4)), wm.download((n, h, w, 4))
    for k in range(n):
        want_r, want_t, want_w = _expect(pix[k], DEEP_GRAY16, w, h, (300, 200, False), (100, True), glyphs, DEFAULT_COL)
        for got, want in ((got_r[k], want_r), (got_t[k], want_t), (got_w[k], want_w), (out["resize"][k], want_r), (out["thumbnail"][k], want_t), (out["watermark"][k], want_w)):
            np.testing.assert_array_equal(got, want)
    plan.close()
    gs.close()

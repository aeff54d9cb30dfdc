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


def _expect_ycbcr_ops(y, cb, cr, ratio, resize, thumb, glyphs, col):
    """What the reference's three helpers give on one *image.YCbCr, via the oracle."""
    h, w = y.shape
    nw, nh = oracle.resize_dims(w, h, *resize)
    out = {"resize": oracle.scale_bilinear_ycbcr(y, cb, cr, ratio, nw, nh)}
    crop, tw, th = oracle.thumb_geometry(w, h, *thumb)
    if thumb[1]:   # cropAndResize: Scale of equal size = DrawYCbCr copy, then the RGBA scale
        cs = crop[2] - crop[0]
        cropped = oracle.scale_bilinear_ycbcr(y, cb, cr, ratio, cs, cs, sr=crop)
        out["thumbnail"] = oracle.scale_bilinear(cropped, tw, th)
    else:          # resizeImage straight from the YCbCr source
        out["thumbnail"] = oracle.scale_bilinear_ycbcr(y, cb, cr, ratio, tw, th)
    wm = oracle.draw_ycbcr(np.zeros((h, w, 4), np.uint8), (0, 0, w, h), y, cb, cr, ratio)
    out["watermark"] = oracle.composite_glyphs(wm, glyphs, col)
    return out


@pytest.mark.parametrize("case", [(1920, 1080, 2, 2, (1024, 768, True), (200, True)),
                                  (640, 480, 3, 2, (1024, 768, True), (200, True)),
                                  (854, 480, 2, 1, (1024, 768, False), (120, False)),
                                  (333, 251, 2, 0, (200, 100, True), (64, True)),
                                  (200, 200, 1, 3, (200, 200, False), (200, True)),
                                  # shapes the fused planar kernel takes (width % 4 == 0), every subsampling ratio
                                  (1280, 720, 2, 0, (1024, 768, True), (200, True)),
                                  (1280, 720, 2, 1, (1024, 768, False), (200, False)),     # non-crop thumbnail: 16-bit taps
                                  (640, 361, 2, 2, (1000, 700, False), (64, True)),        # odd height, non-dyadic resize
                                  (640, 360, 2, 3, (320, 180, True), (200, True)),
                                  (2560, 1440, 1, 2, (1024, 768, True), (200, True)),      # two column blocks
                                  (1920, 1080, 1, 2, (3840, 2160, False), (200, True))],   # upscale: falls back (too many columns)
                         ids=lambda c: "%dx%d r%d" % (c[0], c[1], c[3]))
@pytest.mark.parametrize("fused", ["1", "0", "split", "f64", "cap"], ids=["one-pass", "per-output", "one-pass-split-strips", "one-pass-float64", "one-pass-short-lists"])
def test_ycbcr_batch_plan(ctx, case, fused, monkeypatch):
    monkeypatch.setenv("IPX_FUSED", "0" if fused == "0" else "1")
    if fused == "split":
        monkeypatch.setenv("IPX_KS_STRIPS", "3"); monkeypatch.setenv("IPX_KS_SPLIT", "1"); monkeypatch.setenv("IPX_KS_SPLIT_ROWS", "23")
    if fused == "f64":
        monkeypatch.setenv("IPX_KS_FAST", "0")          # no float pass
    if fused == "cap":
        monkeypatch.setenv("IPX_KS_FIX_CAP", "9")       # the float pass's lists fill up: the frames' items are redone in float64
    from helpers import DEFAULT_COL, text_glyphs
    w, h, n, ratio, resize, thumb = case
    planes = [_rand_ycbcr(w, h, ratio, 50 + i) for i in range(n)]
    y = np.stack([p[0] for p in planes]); cb = np.stack([p[1] for p in planes]); cr = np.stack([p[2] for p in planes])
    glyphs = text_glyphs(w, h, n=6, width_px=min(150, w), height_px=min(30, h))
    gs = ctx.glyphset(glyphs, DEFAULT_COL)
    plan = ctx.plan(w, h, resize=resize, thumbnail=thumb, watermark=gs)
    got = plan.run_host_ycbcr(y, cb, cr, ratio)
    for i in range(n):
        want = _expect_ycbcr_ops(y[i], cb[i], cr[i], ratio, resize, thumb, glyphs, DEFAULT_COL)
        for k in ("resize", "thumbnail", "watermark"):
            np.testing.assert_array_equal(got[k][i], want[k], err_msg="%s frame %d" % (k, i))
    # subsets: thumbnail only (conversion goes to scratch), resize only (no conversion at all)
    for kw, keys in ((dict(resize=None, thumbnail=thumb, watermark=None), ("thumbnail",)),
                     (dict(resize=resize, thumbnail=None, watermark=None), ("resize",))):
        p2 = ctx.plan(w, h, **kw)
        g2 = p2.run_host_ycbcr(y, cb, cr, ratio)
        want = _expect_ycbcr_ops(y[0], cb[0], cr[0], ratio, resize, thumb, glyphs, DEFAULT_COL)
        for k in keys:
            np.testing.assert_array_equal(g2[k][0], want[k])
        p2.close()
    plan.close()
    gs.close()


@pytest.mark.parametrize("case", [(640, 360, 3, (1024, 768, True), (200, True)), (333, 251, 2, (200, 100, False), (64, False)),
                                  (200, 200, 2, (200, 200, False), (100, True)), (1920, 1080, 2, (1024, 768, False), (200, False)),
                                  (1280, 720, 2, (500, 333, False), (200, True))], ids=lambda c: "%dx%d" % (c[0], c[1]))
@pytest.mark.parametrize("fused", ["conv", "split", "0", "f64", "cap"], ids=["one-pass", "one-pass-split-strips", "per-output", "one-pass-float64", "one-pass-short-lists"])
def test_nrgba_batch_plan(ctx, case, fused, monkeypatch):
    """ipx_plan_run_dev_nrgba: a batch of *image.NRGBA frames (PNGs with alpha), per operator as the reference's helpers treat the
    type: 16-bit premultiplied taps for resizeImage, drawNRGBA* first for the crop thumbnail and the watermark.  The converted-tile
    kernel (every source pixel premultiplied once, ipx_band_conv.hip), the per-tap kernel it falls back to (ipx_band_nrgba.hip; both
    need widths that are multiples of 4) and the three-kernel path."""
    from helpers import DEFAULT_COL, text_glyphs
    monkeypatch.setenv("IPX_FUSED", "0" if fused == "0" else "1")
    if fused == "split":
        monkeypatch.setenv("IPX_KS_STRIPS", "2"); monkeypatch.setenv("IPX_KS_SPLIT", "1"); monkeypatch.setenv("IPX_KS_SPLIT_ROWS", "31")
    if fused == "f64":
        monkeypatch.setenv("IPX_KS_FAST", "0")          # no float pass (all four channels are sums here: the alpha too)
    if fused == "cap":
        monkeypatch.setenv("IPX_KS_FIX_CAP", "11")      # the float pass's lists fill up: the frames' items are redone in float64
    w, h, n, resize, thumb = case
    rng = np.random.default_rng(w)
    frames = rng.integers(0, 256, (n, h, w, 4), dtype=np.uint8)          # non-premultiplied: colour may exceed alpha
    frames[0, :, : w // 2, 3] = 255
    glyphs = text_glyphs(w, h, n=6, width_px=min(150, w), height_px=min(30, h))
    gs = ctx.glyphset(glyphs, DEFAULT_COL)
    plan = ctx.plan(w, h, resize=resize, thumbnail=thumb, watermark=gs)
    i = plan.info
    src = ctx.alloc(frames.nbytes).upload(frames)
    res, th, wm = ctx.alloc(n * i.resize_bytes), ctx.alloc(n * i.thumb_bytes), ctx.alloc(n * i.wm_bytes)
    plan.run_dev_nrgba(n, src.ptr, res.ptr, th.ptr, wm.ptr)
    ctx.sync()
    got = {"resize": res.download((n, i.resize_h, i.resize_w, 4)), "thumbnail": th.download((n, i.thumb_h, i.thumb_w, 4)),
           "watermark": wm.download((n, h, w, 4))}
    for k in range(n):
        nw, nh = oracle.resize_dims(w, h, *resize)
        np.testing.assert_array_equal(got["resize"][k], oracle.scale_bilinear_nrgba(frames[k], nw, nh), err_msg="resize %d" % k)
        crop, tw, thh = oracle.thumb_geometry(w, h, *thumb)
        if thumb[1]:
            cs = crop[2] - crop[0]
            cropped = oracle.scale_bilinear_nrgba(frames[k], cs, cs, sr=crop)       # equal size: Copy = drawNRGBAOver onto zeros
            want_t = oracle.scale_bilinear(cropped, tw, thh)
        else:
            want_t = oracle.scale_bilinear_nrgba(frames[k], tw, thh)
        np.testing.assert_array_equal(got["thumbnail"][k], want_t, err_msg="thumbnail %d" % k)
        want_w = oracle.draw_nrgba(np.zeros((h, w, 4), np.uint8), (0, 0, w, h), frames[k])
        np.testing.assert_array_equal(got["watermark"][k], oracle.composite_glyphs(want_w, glyphs, DEFAULT_COL), err_msg="watermark %d" % k)
    plan.close()
    gs.close()


@pytest.mark.parametrize("kind", ["gif", "png-trns"])
def test_paletted_batch_plan(ctx, kind):
    """ipx_plan_run_dev_paletted: *image.Paletted frames (GIF uploads, palette PNGs).  The oracle follows the GENERIC upstream routines
    (scale_RGBA_Image_*, drawRGBA reading Palette[i].RGBA()); the product expands to NRGBA8 and takes the NRGBA pass -- the outputs
    must be the same bytes.  "gif": opaque color.RGBA entries plus a transparent index (the zero colour); "png-trns": color.NRGBA."""
    from helpers import DEFAULT_COL, text_glyphs
    w, h, n, resize, thumb = (640, 360, 3, (320, 180, False), (64, True)) if kind == "gif" else (333, 251, 3, (200, 100, False), (64, True))
    rng = np.random.default_rng(12)
    idx = rng.integers(0, 256, (n, h, w), dtype=np.uint8)
    idx[:, 40:90, 30:200] = 7                                    # flat areas, as palette images have
    pal = rng.integers(0, 256, (n, 256, 4), dtype=np.uint8)
    if kind == "gif":
        pal[..., 3] = 255
        pal[:, 7] = 0                                            # the transparent index: color.RGBA{}
        pal16 = [oracle.palette16(pal[k], "rgba") for k in range(n)]
    else:
        pal[:, :64, 3] = 255
        pal16 = [oracle.palette16(pal[k], "nrgba") for k in range(n)]
    glyphs = text_glyphs(w, h, n=6, width_px=150, height_px=30)
    gs = ctx.glyphset(glyphs, DEFAULT_COL)
    for th in (thumb, (64, False)):
        plan = ctx.plan(w, h, resize=resize, thumbnail=th, watermark=gs)
        i = plan.info
        d_idx, d_pal = ctx.alloc(idx.nbytes).upload(idx), ctx.alloc(pal.nbytes).upload(pal)
        res, tho, wm = ctx.alloc(n * i.resize_bytes), ctx.alloc(n * i.thumb_bytes), ctx.alloc(n * i.wm_bytes)
        plan.run_dev_paletted(n, d_idx.ptr, w, w * h, d_pal.ptr, res.ptr, tho.ptr, wm.ptr)
        ctx.sync()
        got = {"resize": res.download((n, i.resize_h, i.resize_w, 4)), "thumbnail": tho.download((n, i.thumb_h, i.thumb_w, 4)),
               "watermark": wm.download((n, h, w, 4))}
        for k in range(n):
            nw, nh = oracle.resize_dims(w, h, *resize)
            np.testing.assert_array_equal(got["resize"][k], oracle.scale_bilinear_paletted(idx[k], pal16[k], nw, nh), err_msg="resize %d" % k)
            crop, tw, thh = oracle.thumb_geometry(w, h, *th)
            if th[1]:
                cs = crop[2] - crop[0]
                cropped = oracle.scale_bilinear_paletted(idx[k], pal16[k], cs, cs, sr=crop)     # equal sizes: Copy -> drawRGBA, Over onto zeros
                want_t = oracle.scale_bilinear(cropped, tw, thh)
            else:
                want_t = oracle.scale_bilinear_paletted(idx[k], pal16[k], tw, thh)
            np.testing.assert_array_equal(got["thumbnail"][k], want_t, err_msg="thumbnail %d" % k)
            want_w = oracle.draw_paletted(np.zeros((h, w, 4), np.uint8), (0, 0, w, h), idx[k], pal16[k])
            np.testing.assert_array_equal(got["watermark"][k], oracle.composite_glyphs(want_w, glyphs, DEFAULT_COL), err_msg="watermark %d" % k)
        plan.close()
    gs.close()


def test_host_variants_of_the_other_source_types(ctx, monkeypatch):
    """ipx_plan_run_host_{nrgba,gray,paletted}: frames in host memory, chunked over the lanes (several chunks per lane, a ragged last one);
    the same bytes as the oracle's routines for the type."""
    from helpers import DEFAULT_COL, text_glyphs
    monkeypatch.setenv("IPX_HOST_CHUNK", "2")
    w, h, n, resize, thumb = 320, 200, 7, (200, 120, False), (64, True)
    rng = np.random.default_rng(31)
    glyphs = text_glyphs(w, h, n=6, width_px=150, height_px=30)
    gs = ctx.glyphset(glyphs, DEFAULT_COL)
    plan = ctx.plan(w, h, resize=resize, thumbnail=thumb, watermark=gs)
    crop, tw, th = oracle.thumb_geometry(w, h, *thumb)
    cs = crop[2] - crop[0]
    zeros = lambda: np.zeros((h, w, 4), np.uint8)   # noqa: E731

    nrgba = rng.integers(0, 256, (n, h, w, 4), dtype=np.uint8)
    got = plan.run_host_nrgba(nrgba)
    for k in range(n):
        np.testing.assert_array_equal(got["resize"][k], oracle.scale_bilinear_nrgba(nrgba[k], resize[0], resize[1]))
        np.testing.assert_array_equal(got["thumbnail"][k], oracle.scale_bilinear(oracle.scale_bilinear_nrgba(nrgba[k], cs, cs, sr=crop), tw, th))
        np.testing.assert_array_equal(got["watermark"][k], oracle.composite_glyphs(oracle.draw_nrgba(zeros(), (0, 0, w, h), nrgba[k]), glyphs, DEFAULT_COL))

    gray = rng.integers(0, 256, (n, h, w), dtype=np.uint8)
    got = plan.run_host_gray(gray)
    for k in range(n):
        rgba = np.dstack([gray[k]] * 3 + [np.full_like(gray[k], 255)])
        want = oracle.process(rgba, resize=resize, thumb=thumb, glyphs=glyphs, col=DEFAULT_COL)
        for key in ("resize", "thumbnail", "watermark"):
            np.testing.assert_array_equal(got[key][k], want[key], err_msg="gray %s %d" % (key, k))

    pal = rng.integers(0, 256, (n, 256, 4), dtype=np.uint8)
    pal[:, :128, 3] = 255
    got = plan.run_host_paletted(gray, pal)
    for k in range(n):
        p16 = oracle.palette16(pal[k], "nrgba")
        np.testing.assert_array_equal(got["resize"][k], oracle.scale_bilinear_paletted(gray[k], p16, resize[0], resize[1]))
        np.testing.assert_array_equal(got["thumbnail"][k], oracle.scale_bilinear(oracle.scale_bilinear_paletted(gray[k], p16, cs, cs, sr=crop), tw, th))
        np.testing.assert_array_equal(got["watermark"][k], oracle.composite_glyphs(oracle.draw_paletted(zeros(), (0, 0, w, h), gray[k], p16), glyphs, DEFAULT_COL))
    plan.close()
    gs.close()

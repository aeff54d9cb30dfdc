"""The glyph mask producer (csrc/ipx_font.cpp: truetype.Parse + freetype.Context.DrawString).

Host logic, no GPU.  PARITY UNPINNED against the Go library (no Go toolchain, the module and the Go Regular
face are not in the image); pinned here by
  * hand-derivable coverage values on a font built for the purpose (axis-aligned boxes with 1/64-pixel
    edges, a triangle through pixel corners),
  * fontTools for the tables (cmap, hmtx, kern),
  * an independent pure-Python model of the same published algorithm (oracle/ft_model.py), bit for bit,
  * FreeType itself (Pillow), within a tolerance, since Pillow hints and this path does not.
"""
import io
import os

import numpy as np
import pytest

import imageprocessor_amd as ipx
from imageprocessor_amd.operations import TrueTypeFont

DEJAVU = next((p for p in ("/usr/share/fonts/truetype/dejavu/DejaVuSans.ttf",
                           "/usr/local/lib/python3.10/dist-packages/matplotlib/mpl-data/fonts/ttf/DejaVuSans.ttf")
               if os.path.exists(p)), None)
needs_dejavu = pytest.mark.skipif(DEJAVU is None, reason="DejaVuSans.ttf not in this image")
DEFAULT_TEXT = "© ImageProcessor"   # domain.DefaultWatermarkText


def build_font(glyphs, upem=4096, advance=64 * 20):
    """A TrueType font with the given {char: [contour, ...]} outlines (all points on-curve unless a point is
    (x, y, 0)); upem 4096 at font size 64 (scale 4096 in 26.6) makes one font unit one 1/64 pixel."""
    from fontTools.fontBuilder import FontBuilder
    from fontTools.pens.ttGlyphPen import TTGlyphPen
    from fontTools.ttLib.tables._g_l_y_f import Glyph
    names = [".notdef"] + ["g%d" % i for i in range(len(glyphs))]
    fb = FontBuilder(upem, isTTF=True)
    fb.setupGlyphOrder(names)
    fb.setupCharacterMap({ord(ch): "g%d" % i for i, ch in enumerate(glyphs)})
    tab = {".notdef": Glyph()}
    for i, (ch, contours) in enumerate(glyphs.items()):
        pen = TTGlyphPen(None)
        for c in contours:
            on = [p for p in c]
            pen.moveTo(on[0][:2])
            k = 1
            while k < len(on):
                if len(on[k]) == 3 and on[k][2] == 0:   # off-curve control followed by an on-curve point
                    pen.qCurveTo(on[k][:2], on[(k + 1) % len(on)][:2])   # a trailing control closes onto the start
                    k += 2
                else:
                    pen.lineTo(on[k][:2])
                    k += 1
            pen.closePath()
        tab["g%d" % i] = pen.glyph()
    fb.setupGlyf(tab)
    fb.setupHorizontalMetrics({n: (advance, tab[n].xMin if hasattr(tab[n], "xMin") and tab[n].numberOfContours else 0) for n in names})
    fb.setupHorizontalHeader(ascent=upem, descent=0)
    fb.setupNameTable({"familyName": "ipxtest", "styleName": "Regular"})
    fb.setupOS2()
    fb.setupPost()
    buf = io.BytesIO()
    fb.save(buf)
    return buf.getvalue()


def box(x0, y0, x1, y1):
    # clockwise in y-up font space (TrueType's filled direction)
    return [(x0, y0), (x0, y1), (x1, y1), (x1, y0)]


def render_one(font, ch, size=64, px=0, py=40, w=64, h=64):
    gl, _ = font.draw_string(ch, size, px, py, w, h)
    canvas = np.zeros((h, w), np.uint8)
    for g in gl:
        x0, y0, x1, y1 = g["dr"]
        mx, my = g["mp"]
        canvas[y0:y1, x0:x1] = g["mask"][my:my + y1 - y0, mx:mx + x1 - x0]
    return canvas


def test_axis_aligned_boxes_have_exact_coverage():
    """A box with edges at multiples of 1/64 px covers wx*wy/4096 of a pixel: the 12-bit alpha is
    min(wx*wy, 4095) and the mask byte its top 8 bits (areaToAlpha, AlphaSrcPainter)."""
    # box from (2.5, 1.25) to (9.75, 6.5) pixels above the baseline, in 1/64 px units
    x0, y0, x1, y1 = 160, 80, 624, 416
    f = TrueTypeFont(build_font({"A": [box(x0, y0, x1, y1)]}))
    got = render_one(f, "A", py=40)
    want = np.zeros_like(got)
    for Y in range(64):
        for X in range(64):
            wx = max(0, min(x1, (X + 1) * 64) - max(x0, X * 64))
            # pixel row Y spans font y in [(40 - Y - 1) * 64, (40 - Y) * 64]
            wy = max(0, min(y1, (40 - Y) * 64) - max(y0, (40 - Y - 1) * 64))
            want[Y, X] = min(wx * wy, 4095) >> 4
    np.testing.assert_array_equal(got, want)
    assert got.max() == 255 and got[33, 3] == 0x80 and got[33, 2] == 0x40 and got[38, 9] == (48 * 48) >> 4
    f.close()


def test_triangle_through_pixel_corners():
    """Right triangle (0,0) (8,0) (0,8) px: pixels under the diagonal are full, pixels ON it are exactly half
    covered (alpha12 2048 -> byte 0x80), everything else empty: the sloped scan path distributes exactly."""
    f = TrueTypeFont(build_font({"T": [[(0, 0), (0, 512), (512, 0)]]}))
    got = render_one(f, "T", py=40)
    want = np.zeros_like(got)
    for Y in range(32, 40):
        r = 39 - Y            # rows above the baseline: 0 at the bottom
        want[Y, :7 - r] = 255
        want[Y, 7 - r] = 0x80
    np.testing.assert_array_equal(got, want)
    f.close()


def test_subpixel_x_positions_and_mp_quirk():
    """DrawString keeps 4 horizontal sub-pixel positions (p.X & 63 selects the rasterisation) and passes
    mp = (0, dr.Min.Y - glyphRect.Min.Y): x is NOT adjusted when the clip cuts a glyph's left side."""
    ttf = build_font({"A": [box(0, 0, 640, 640)], "B": [box(16, 0, 656, 640)]}, advance=16)
    f = TrueTypeFont(ttf)
    # "AB...": B starts a quarter pixel in (advance 16/64) and is itself offset by 16/64: its left edge sits at x = 0.5 px
    gl, endx = f.draw_string("AB", 64, 3, 20, 100, 100)
    assert endx == (3 << 6) + 32
    a, b = gl
    assert a["dr"] == (3, 10, 13, 20) and (a["mask"] == 255).all()
    assert b["dr"][0] == 3 and b["mask"][0, 0] == 0x80 and b["mask"][0, 1] == 255 and b["mask"][0, -1] == 0x80
    # clip on the left and top: dr shrinks, mp.y follows, mp.x stays 0 (so the mask's LEFT columns are used)
    gl, _ = f.draw_string("B", 64, -4, 5, 100, 100)
    g = gl[0]
    assert g["dr"] == (0, 0, 7, 5) and g["mp"] == (0, 5)
    f.close()


def test_context_glyph_cache_reuses_the_first_rasterisation_in_a_bucket():
    """(c *Context) glyph of golang/freetype: 4 sub-pixel slots per glyph, and a hit needs only `e.valid && e.glyph == glyph` -- a
    repeated glyph whose fx falls into the same quarter-pixel bucket gets the mask rasterised at the FIRST fx of that bucket
    (addTextWatermark makes a fresh Context per call, watermark.go:98, so the cache lives for one DrawString).  At font size 64
    one font unit is 1/64 px: an advance of 645 units moves fx by 5 per glyph: 0, 5, 10, 15 (bucket 0), 20 (bucket 1) ..."""
    from oracle import ft_model
    ttf = build_font({"A": [box(0, 0, 645 - 64, 640)]}, advance=645)
    f = TrueTypeFont(ttf)
    gl, _ = f.draw_string("AAAAAA", 64, 2, 20, 200, 100)
    assert len(gl) == 6
    # fx = 0 for the first glyph: its left edge is whole, and the three glyphs that follow in bucket 0 carry the SAME mask although
    # a rasterisation at their own fx (5, 10, 15) would shade the left column
    first = gl[0]["mask"]
    assert (first[:, 0] == 255).all()
    for k in (1, 2, 3):
        np.testing.assert_array_equal(gl[k]["mask"], first, err_msg="glyph %d" % k)
    # the fifth glyph opens bucket 1 and is rasterised at fx = 20: a partially covered left column
    assert 0 < gl[4]["mask"][0, 0] < 255 and not np.array_equal(gl[4]["mask"], first)
    np.testing.assert_array_equal(gl[5]["mask"], gl[4]["mask"])          # fx = 25: bucket 1 again
    # positions still advance by the true pen position (the cached offset is relative to the integer part)
    assert [g["dr"][0] for g in gl] == [2 + (645 * k) // 64 for k in range(6)]
    # the Python model carries the same rule
    want, _ = ft_model.draw_string(ft_model.Font(io.BytesIO(ttf)), "AAAAAA", 64, 2, 20, 200, 100)
    for g, w_ in zip(gl, want):
        np.testing.assert_array_equal(g["mask"], w_["mask"])
        assert g["dr"] == w_["dr"]
    f.close()


def test_quadratic_contour_area():
    """Four quadratic arcs with the controls at the corners of the bounding square: the ink must match the
    analytic area (10/3 R^2) to 0.5 %, i.e. Add2's subdivision is fine enough and closed."""
    R, c = 640, 700
    k = R   # control points at the corners of the bounding square: a "squircle" of quadratic arcs
    contour = [(c + R, c), (c + R, c - k, 0), (c, c - R), (c - R, c - k, 0), (c - R, c), (c - R, c + k, 0), (c, c + R),
               (c + R, c + k, 0)]
    f = TrueTypeFont(build_font({"O": [contour]}))
    got = render_one(f, "O", py=40, w=40, h=48)
    # per quadrant: chord triangle R^2/2 + the arc's segment, 2/3 of the control triangle R^2/2 => 5/6 R^2
    area_px = (10.0 / 3) * (R / 64.0) ** 2
    assert abs(got.astype(np.float64).sum() / 255 - area_px) / area_px < 0.005
    f.close()


@needs_dejavu
def test_tables_against_fonttools():
    from fontTools.ttLib import TTFont
    tt = TTFont(DEJAVU)
    f = TrueTypeFont.from_file(DEJAVU)
    # parseCmap stops at the first Unicode subtable: here (0,3), format 4, BMP only -- U+1F600 is .notdef for Go
    cmap = next(t for t in tt["cmap"].tables if (t.platformID, t.platEncID) == (0, 3)).cmap
    assert 0x1F600 in tt.getBestCmap() and 0x1F600 not in cmap
    upem = tt["head"].unitsPerEm
    for cp in list(range(32, 127)) + [0xA9, 0xE9, 0x416, 0x20AC, 0x1F600, 0xFFFF]:
        want = tt.getGlyphID(cmap[cp]) if cp in cmap else 0
        assert f.index(cp) == want, hex(cp)
        for size in (12, 36, 17.3):
            scale = int(0.5 + size * 64)   # truetype.NewFace
            adv = tt["hmtx"][tt.getGlyphName(want)][0]
            # advance = scale(s*(xmin-lsb+aw)) - scale(s*(xmin-lsb)); xmin == lsb in this font for simple glyphs
            g = tt["glyf"][tt.getGlyphName(want)]
            off = (g.xMin if g.numberOfContours else 0) - tt["hmtx"][tt.getGlyphName(want)][1]

            def sc(x):
                return (x + upem // 2) // upem if x >= 0 else -((-x + upem // 2) // upem)
            assert f.glyph_advance(cp, size) == sc(scale * (off + adv)) - sc(scale * off)
    kt = tt["kern"].kernTables[0].kernTable
    for (a, b), v in list(kt.items())[:200]:
        ra = [k for k, n in cmap.items() if n == a]
        rb = [k for k, n in cmap.items() if n == b]
        if ra and rb and ra[0] < 0x10000 and rb[0] < 0x10000:
            x = 36 * 64 * v
            want = (x + upem // 2) // upem if x >= 0 else -((-x + upem // 2) // upem)
            assert f.kern(chr(ra[0]), chr(rb[0]), 36) == want
    assert f.kern("x", "x", 36) == 0
    f.close()


@needs_dejavu
@pytest.mark.parametrize("size", [36, 12, 17.3, 96])
def test_masks_match_python_model(size):
    """csrc/ipx_font.cpp against oracle/ft_model.py on real outlines (simple, compound, transformed-compound
    glyphs), all four sub-pixel positions, bit for bit."""
    from oracle import ft_model
    model = ft_model.Font(DEJAVU)
    f = TrueTypeFont.from_file(DEJAVU)
    text = DEFAULT_TEXT + " gjQ@&%éÅЖǄ≠½"
    for frac_px, py in ((0, 150), (1, 150)):
        got, endx = f.draw_string(text, size, 7 + frac_px, py, 4000, 400)
        want, wendx = ft_model.draw_string(model, text, size, 7 + frac_px, py, 4000, 400)
        assert endx == wendx
        assert len(got) == len(want)
        for g, w, ch in zip(got, want, [c for c in text if c != " "]):
            assert g["dr"] == w["dr"] and g["mp"] == w["mp"], ch
            np.testing.assert_array_equal(g["mask"], w["mask"], err_msg=repr(ch))
    assert f.text_width(text, size) == ft_model.text_width(model, text, size)
    f.close()


@needs_dejavu
def test_clipped_text_matches_model():
    from oracle import ft_model
    model = ft_model.Font(DEJAVU)
    f = TrueTypeFont.from_file(DEJAVU)
    for px, py, w, h in ((-10, 20, 120, 30), (100, 300, 160, 310), (5, 5, 64, 64)):
        got, _ = f.draw_string(DEFAULT_TEXT, 36, px, py, w, h)
        want, _ = ft_model.draw_string(model, DEFAULT_TEXT, 36, px, py, w, h)
        assert [(g["dr"], g["mp"]) for g in got] == [(g["dr"], g["mp"]) for g in want]
        for g, w_ in zip(got, want):
            np.testing.assert_array_equal(g["mask"], w_["mask"])
    f.close()


@needs_dejavu
def test_close_to_freetype_via_pillow():
    """FreeType (through Pillow) hints and snaps advances to whole pixels, so only glyph-level agreement is
    asked for: per-glyph ink within 6 % and bounding boxes within one pixel."""
    from PIL import Image, ImageDraw, ImageFont
    f = TrueTypeFont.from_file(DEJAVU)
    pf = ImageFont.truetype(DEJAVU, 36)
    for ch in "IPmgeo©cs":
        gl, _ = f.draw_string(ch, 36, 10, 60, 100, 100)
        mine = np.zeros((100, 100), np.uint8)
        g = gl[0]
        x0, y0, x1, y1 = g["dr"]
        mine[y0:y1, x0:x1] = g["mask"][g["mp"][1]:g["mp"][1] + y1 - y0, :x1 - x0]
        im = Image.new("L", (100, 100), 0)
        ImageDraw.Draw(im).text((10, 60), ch, font=pf, fill=255, anchor="ls")
        ref = np.asarray(im)
        a, b = mine.astype(np.float64).sum(), ref.astype(np.float64).sum()
        assert abs(a - b) / b < 0.06, (ch, a, b)
        ys, xs = np.nonzero(mine > 32)
        ry, rx = np.nonzero(ref > 32)
        assert abs(ys.min() - ry.min()) <= 1 and abs(ys.max() - ry.max()) <= 1, ch
        assert abs(xs.min() - rx.min()) <= 1 and abs(xs.max() - rx.max()) <= 1, ch
    f.close()


def test_parse_errors_are_the_nil_font():
    """truetype.Parse failing leaves the Watermarker without a font (watermark.go:31-34): create fails with text."""
    for blob in (b"", b"\x00\x01\x00\x00" + b"\x00" * 8, b"OTTO" + b"\x00" * 60, b"nonsense" * 10):
        with pytest.raises(ipx.IpxError):
            TrueTypeFont(blob)


def test_utf8_iteration_like_go_range():
    """`for _, r := range s`: every invalid byte is one U+FFFD (here: .notdef, same advance each)."""
    f = TrueTypeFont(build_font({"A": [box(0, 0, 64, 64)]}, advance=64))
    w_ok, _ = f.text_width("AA", 64)
    out, n, endx = (ipx._lib.C.POINTER(ipx._lib.Glyph)(), ipx._lib.C.c_int(), ipx._lib.C.c_int32())
    L = ipx.lib()
    wid = ipx._lib.C.c_int32()
    L.ipx_font_text_width(f.handle, b"A\xff\xc3A", 64.0, ipx._lib.C.byref(wid), None)
    assert wid.value == 2 * (w_ok // 2) + 2 * f.glyph_advance(0xFFFD, 64)
    f.close()

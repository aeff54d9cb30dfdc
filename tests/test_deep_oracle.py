"""CPU: the oracle's routines for the 16-bit image types of Go's PNG decoder (*image.NRGBA64, *image.RGBA64, *image.Gray16) and for
*image.CMYK (oracle/ipx_oracle.c "deep sources"; SURVEY.md section 8(f) N2 tail; resize.go:121-125 takes them through the generic
scale_RGBA_Image_* routines, watermark.go:92 through image/draw's drawRGBA / drawCMYK).

The reference holds no fixture for these types and Go cannot run here: parity unpinned, like the rest of the pixel path.  What IS
checked: (a) the taps equal color.{NRGBA64,RGBA64,Gray16,CMYK}.RGBA() restated independently in numpy; (b) a deep frame whose
channels are 8-bit values widened by 0x101 goes through the deep routines to the SAME bytes as the 8-bit type's own routines (the
KAT-pinned scale_RGBA_NRGBA_* / scale_RGBA_RGBA_* / drawNRGBA* restatements) -- (c*0x101)*(a*0x101)/0xffff == (c*0x101)*a/0xff
exactly; (c) hand-computed known answers."""
import numpy as np
import pytest

import oracle
from oracle import DEEP_CMYK, DEEP_GRAY16, DEEP_NRGBA64, DEEP_RGBA64


def _np_taps(values, kind):
    v = np.asarray(values).astype(np.uint64)
    if kind == DEEP_GRAY16:
        return np.stack([v, v, v, np.full_like(v, 0xffff)], -1).astype(np.uint16)
    if kind == DEEP_CMYK:
        w = 0xffff - v[..., 3] * 0x101
        rgb = [(0xffff - v[..., c] * 0x101) * w // 0xffff for c in range(3)]
        return np.stack(rgb + [np.full_like(w, 0xffff)], -1).astype(np.uint16)
    if kind == DEEP_NRGBA64:
        a = v[..., 3:4]
        return np.concatenate([v[..., :3] * a // 0xffff, a], -1).astype(np.uint16)
    return v.astype(np.uint16)


@pytest.mark.parametrize("kind", [DEEP_NRGBA64, DEEP_RGBA64, DEEP_GRAY16, DEEP_CMYK], ids=["nrgba64", "rgba64", "gray16", "cmyk"])
def test_taps_are_the_colour_types_rgba(kind):
    rng = np.random.default_rng(kind)
    h, w = 37, 53
    if kind == DEEP_GRAY16:
        vals = rng.integers(0, 65536, (h, w), dtype=np.uint16)
        vals[0, :4] = [0, 1, 0xfffe, 0xffff]
    elif kind == DEEP_CMYK:
        vals = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        vals[0, 0] = 0; vals[0, 1] = 255; vals[0, 2] = [0, 0, 0, 255]; vals[0, 3] = [255, 255, 255, 0]
    else:
        vals = rng.integers(0, 65536, (h, w, 4), dtype=np.uint16)
        vals[0, 0] = 0xffff; vals[0, 1] = 0; vals[0, 2] = [0xffff, 0xffff, 0xffff, 1]; vals[0, 3] = [1, 2, 3, 0xffff]
        vals[1, :, 3] = 0xffff
    pix = oracle.deep_pix(vals, kind)
    np.testing.assert_array_equal(oracle.deep_taps(pix, kind), _np_taps(vals, kind))
    # draw.Draw(Src) onto an RGBA frame keeps the top byte of every tap (drawRGBA; drawCMYK = CMYKToRGB)
    got = oracle.draw_deep(np.full((h, w, 4), 77, np.uint8), (0, 0, w, h), pix, kind)
    np.testing.assert_array_equal(got, (_np_taps(vals, kind) >> 8).astype(np.uint8))
    # Over onto a zeroed frame stores the same bytes (a = (m - sa) * 0x101 multiplies zeros)
    got = oracle.draw_deep(np.zeros((h, w, 4), np.uint8), (0, 0, w, h), pix, kind, op=oracle.OP_OVER)
    np.testing.assert_array_equal(got, (_np_taps(vals, kind) >> 8).astype(np.uint8))


def test_widened_8bit_frames_take_the_8bit_routines_bytes():
    rng = np.random.default_rng(5)
    h, w = 61, 83
    px = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    px[:, : w // 3, 3] = 255
    wide = px.astype(np.uint16) * 0x101
    n64 = oracle.deep_pix(wide, DEEP_NRGBA64)
    for dw, dh, sr in ((40, 30, None), (100, 90, None), (w, h, None), (25, 25, (10, 5, 60, 55))):
        np.testing.assert_array_equal(oracle.scale_bilinear_deep(n64, DEEP_NRGBA64, dw, dh, sr=sr), oracle.scale_bilinear_nrgba(px, dw, dh, sr=sr))
    dst = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    dst[..., :3] = np.minimum(dst[..., :3], dst[..., 3:4])
    for op in (oracle.OP_SRC, oracle.OP_OVER):
        np.testing.assert_array_equal(oracle.draw_deep(dst.copy(), (0, 0, w, h), n64, DEEP_NRGBA64, op=op),
                                      oracle.draw_nrgba(dst.copy(), (0, 0, w, h), px, op=op))
    # premultiplied: RGBA64 against the RGBA routines
    pm = px.copy()
    pm[..., :3] = np.minimum(pm[..., :3], pm[..., 3:4])
    r64 = oracle.deep_pix(pm.astype(np.uint16) * 0x101, DEEP_RGBA64)
    for dw, dh in ((40, 30), (100, 90)):
        np.testing.assert_array_equal(oracle.scale_bilinear_deep(r64, DEEP_RGBA64, dw, dh), oracle.scale_bilinear(pm, dw, dh))
        under = np.ascontiguousarray(np.resize(dst, (dh, dw, 4)))           # Over a frame that already holds pixels (the source is not opaque)
        np.testing.assert_array_equal(oracle.scale_bilinear_deep(r64, DEEP_RGBA64, dw, dh, dst=under.copy()), oracle.scale_bilinear(pm, dw, dh, dst=under.copy()))
    # Gray16 against an opaque gray RGBA frame
    g = rng.integers(0, 256, (h, w), dtype=np.uint8)
    g16 = oracle.deep_pix(g.astype(np.uint16) * 0x101, DEEP_GRAY16)
    rgba = np.stack([g, g, g, np.full_like(g, 255)], -1)
    np.testing.assert_array_equal(oracle.scale_bilinear_deep(g16, DEEP_GRAY16, 33, 47), oracle.scale_bilinear(rgba, 33, 47))


def test_known_answers():
    # one NRGBA64 pixel pair scaled 2 -> 4 wide: taps c*a/0xffff, tent weights (scale 0.5, support 1), ftou's round to nearest, >> 8
    vals = np.array([[[0x8000, 0x4000, 0xffff, 0x8000], [0xffff, 0x0000, 0x0001, 0xffff]]], np.uint16)
    pix = oracle.deep_pix(vals, DEEP_NRGBA64)
    taps = oracle.deep_taps(pix, DEEP_NRGBA64)
    assert taps.tolist() == [[[0x8000 * 0x8000 // 0xffff, 0x4000 * 0x8000 // 0xffff, 0xffff * 0x8000 // 0xffff, 0x8000], [0xffff, 0, 1, 0xffff]]]
    out = oracle.scale_bilinear_deep(pix, DEEP_NRGBA64, 4, 1)
    # centres -0.25, 0.25, 0.75, 1.25: dx = 0 sees only tap 0 (tap -1 does not exist, tap 1 is at t = 1.25), dx = 1 weights 0.75 / 0.25, dx = 2 0.25 / 0.75,
    # dx = 3 only tap 1 -- every weight dyadic, so the float64 sums are exact.  The source is not opaque: Over onto zeros.
    t0, t1 = taps[0, 0].astype(np.float64), taps[0, 1].astype(np.float64)
    want = [(t0.astype(np.uint32) >> 8), ((0.75 * t0 + 0.25 * t1 + 0.5).astype(np.uint32) >> 8), ((0.25 * t0 + 0.75 * t1 + 0.5).astype(np.uint32) >> 8), (t1.astype(np.uint32) >> 8)]
    assert out[0].tolist() == [list(map(int, x)) for x in want]
    # CMYK: (c, m, y, k) = (0, 128, 255, 64): w = 0xffff - 64*0x101; r = 0xffff * w / 0xffff = w, ...
    cm = oracle.deep_pix(np.array([[[0, 128, 255, 64]]], np.uint8), DEEP_CMYK)
    wv = 0xffff - 64 * 0x101
    assert oracle.deep_taps(cm, DEEP_CMYK)[0, 0].tolist() == [wv, (0xffff - 128 * 0x101) * wv // 0xffff, 0, 0xffff]

"""*image.Paletted sources: the generic upstream routines (oracle: scale_RGBA_Image_*, drawRGBA on Palette[i].RGBA()) give the bytes of the
NRGBA routines on the expanded pixels for every palette the GIF / PNG decoders build -- the equivalence ipx_plan_run_dev_paletted rests on."""
import numpy as np
import pytest

import oracle


@pytest.mark.parametrize("kind", ["rgba", "nrgba"])
def test_generic_paletted_routines_equal_nrgba_routines_on_expanded_pixels(kind):
    rng = np.random.default_rng(3)
    pal = rng.integers(0, 256, (256, 4), dtype=np.uint8)
    if kind == "rgba":
        pal[:, 3] = 255
        pal[9] = 0                       # a GIF's transparent index
    else:
        pal[:50, 3] = 255
        pal[50:60, 3] = 0                # colour under zero alpha
    idx = rng.integers(0, 256, (61, 83), dtype=np.uint8)
    p16 = oracle.palette16(pal, kind)
    if kind == "rgba":
        np.testing.assert_array_equal(p16, oracle.palette16(pal, "nrgba"))    # opaque and zero entries: both colour types agree
    px = pal[idx]
    for dw, dh in ((31, 23), (200, 150), (83, 61)):
        for op in (oracle.OP_SRC, oracle.OP_OVER):
            base = rng.integers(0, 256, (dh, dw, 4), dtype=np.uint8)
            a = oracle.scale_bilinear_paletted(idx, p16, dw, dh, op=op, dst=base.copy())
            b = oracle.scale_bilinear_nrgba(px, dw, dh, op=op, dst=base.copy())
            np.testing.assert_array_equal(a, b)
    for op in (oracle.OP_SRC, oracle.OP_OVER):
        base = rng.integers(0, 256, (61, 83, 4), dtype=np.uint8)
        a = oracle.draw_paletted(base.copy(), (5, 4, 70, 50), idx, p16, sp=(3, 2), op=op)
        b = oracle.draw_nrgba(base.copy(), (5, 4, 70, 50), px, sp=(3, 2), op=op)
        np.testing.assert_array_equal(a, b)


def test_paletted_known_answers():
    """Hand-computed: a two-colour palette, 2x1 -> 4x1.  Entry 0 = NRGBA(200, 100, 50, 128): RGBA() = (200*0x101*128/0xff, ...) =
    (25800, 12900, 6450, 32896); entry 1 = opaque white.  Destination x = 0 clamps to tap 0, x = 3 to tap 1; x = 1 sits at sx = 0.25."""
    pal = np.zeros((256, 4), np.uint8)
    pal[0] = (200, 100, 50, 128)
    pal[1] = 255
    p16 = oracle.palette16(pal, "nrgba")
    assert tuple(int(v) for v in p16[0]) == (200 * 0x101 * 128 // 0xff, 100 * 0x101 * 128 // 0xff, 50 * 0x101 * 128 // 0xff, 128 * 0x101)
    assert tuple(int(v) for v in p16[1]) == (0xffff,) * 4
    out = oracle.scale_bilinear_paletted(np.array([[0, 1]], np.uint8), p16, 4, 1, op=oracle.OP_SRC)
    assert tuple(out[0, 0]) == tuple(int(v) >> 8 for v in p16[0])
    assert tuple(out[0, 3]) == (255, 255, 255, 255)
    want = tuple(int(0.75 * float(p16[0][c]) + 0.25 * 65535.0) >> 8 for c in range(4))
    assert tuple(int(v) for v in out[0, 1]) == want

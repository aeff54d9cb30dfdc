"""The decision rule of the one-pass kernel's float pass (csrc/ipx_ks_fused.hip, ks_float_eps in csrc/ipx_ks_host.cpp), on the CPU.

The kernel stores a byte from float32 sums only when T = (V' + 0.5) / 256 lies at least (nx + ny + 4) * 2^-24 * T away from every
integer; every other value is recomputed in float64.  This test restates both sides in numpy -- the reference's float64 walk (products
rounded before they are added, * invTotalWeightFFFF, the vertical sums, * invTotalWeight, ftou, >> 8: tests/golden/make_kats.py's
model, vectorised) and the float pass (weights folded and rounded to float32, one fused multiply-add per term) -- and checks on
millions of values, adversarial ones among them, that WHENEVER the rule decides, the float byte IS the float64 byte.  It also counts how
often the rule does not decide (the share of pixels the exact pass has to recompute)."""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from make_kats import new_distrib  # noqa: E402


def _fma32(a, b, c):
    """fl32(a * b + c) with ONE rounding, for float32 arrays whose products are exact in float64 (a: integers below 2^16)."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def _axis(dw, sw):
    d = new_distrib(dw, sw)
    ntap = max(len(c) for c, _, _ in d)
    lo = np.array([c[0][0] for c, _, _ in d])
    w = np.zeros((dw, ntap))
    for i, (c, _, _) in enumerate(d):
        w[i, :len(c)] = [x[1] for x in c]
    itw = np.array([t for _, t, _ in d])
    itwf = np.array([t for _, _, t in d])
    return lo, w, itw, itwf, ntap


def _both_ways(src, dw, dh, unit):
    """src: sh x sw array of taps as the reference weights them (16-bit values: bytes * 0x101, or any 16-bit value).
    -> (byte64, byte32, decided): the reference's byte, the float pass's byte, and where the float pass decides."""
    sh, sw = src.shape
    xlo, wx, _, xitwf, nx = _axis(dw, sw)
    ylo, wy, yitw, _, ny = _axis(dh, sh)
    cols = np.minimum(xlo[:, None] + np.arange(nx)[None, :], sw - 1)          # padded taps carry weight 0
    rows = np.minimum(ylo[:, None] + np.arange(ny)[None, :], sh - 1)
    # ---- the reference: scaleX into tmp (float64, product rounded, then added), scaleY, ftou, >> 8
    tmp = np.zeros((sh, dw))
    for t in range(nx):
        tmp = tmp + src[:, cols[:, t]].astype(np.float64) * wx[None, :, t]
    tmp = tmp * xitwf[None, :]
    acc = np.zeros((dh, dw))
    for t in range(ny):
        acc = acc + tmp[rows[:, t], :] * wy[:, t][:, None]
    f = acc * yitw[:, None]
    i = (0xFFFF * f + 0.5).astype(np.int64)
    byte64 = np.clip(i, 0, 0xFFFF) >> 8
    # ---- the float pass: folded float32 weights, one fma per term; taps are what the tile holds (bytes when unit carries the 0x101)
    taps = (src // 0x101 if unit == 257 else src).astype(np.float32)
    wxf = (wx * xitwf[:, None] * 65535.0 * unit).astype(np.float32)
    wyf = (wy * yitw[:, None]).astype(np.float32)
    tmpf = np.zeros((sh, dw), np.float32)
    for t in range(nx):
        tmpf = _fma32(taps[:, cols[:, t]], np.broadcast_to(wxf[None, :, t], (sh, dw)), tmpf)
    q = np.zeros((dh, dw), np.float32)
    for t in range(ny):
        q = _fma32(tmpf[rows[:, t], :], np.broadcast_to(wyf[:, t][:, None], (dh, dw)), q)
    T = _fma32(q, np.full_like(q, 1.0 / 256.0), np.full_like(q, 0.5 / 256.0))
    feps = np.float32(np.nextafter(np.float32((nx + ny + 4) / 16777216.0), np.float32(1.0)))
    d = T - np.rint(T)
    decided = ~(np.abs(d) < feps * T)
    byte32 = np.floor(T).astype(np.int64)
    assert byte32.max() <= 255
    return byte64, byte32, decided


def test_the_float_pass_never_decides_wrongly():
    rng = np.random.default_rng(20261005)
    total = undecided = 0
    for sw, sh, dw, dh in ((1920, 270, 1024, 192), (1080, 270, 200, 50), (960, 128, 480, 64), (640, 96, 160, 24), (300, 64, 512, 96),
                           (1920, 132, 256, 12)):
        for kind in ("noise", "smooth", "flat-with-steps"):
            if kind == "noise":
                b = rng.integers(0, 256, (sh, sw))
            elif kind == "smooth":
                yy, xx = np.mgrid[0:sh, 0:sw]
                b = ((np.sin(xx / 37.0) + np.cos(yy / 23.0)) * 60 + 128 + rng.normal(0, 2, (sh, sw))).clip(0, 255).astype(np.int64)
            else:
                b = np.repeat(rng.integers(0, 256, (sh, (sw + 7) // 8)), 8, axis=1)[:, :sw]
            b64, b32, dec = _both_ways(b * 0x101, dw, dh, 257)
            assert np.array_equal(b64[dec], b32[dec]), (sw, sh, dw, dh, kind)
            total += dec.size
            undecided += int((~dec).sum())
        # 16-bit taps (the converted sources' tiles)
        v = rng.integers(0, 65536, (sh, sw))
        b64, b32, dec = _both_ways(v, dw, dh, 1)
        assert np.array_equal(b64[dec], b32[dec]), (sw, sh, dw, dh, "16-bit")
        total += dec.size
        undecided += int((~dec).sum())
    assert total > 1_000_000
    assert undecided / total < 0.004, undecided / total            # a few in a thousand at most go to the exact pass


def test_values_exactly_on_a_boundary_are_never_decided():
    """Exact 2:1 downscales: the weights are (1, 3, 3, 1) / 8 per axis, so sum(tap * weight) + 0.5 is EXACTLY a multiple of 256 for about
    one value in 3000 of a random frame; the reference's byte there is decided by float64 roundings.  The rule must leave every one of
    them (and it must be right wherever it decides)."""
    rng = np.random.default_rng(7)
    b = rng.integers(0, 256, (512, 1024))
    b64, b32, dec = _both_ways(b * 0x101, 512, 256, 257)
    assert np.array_equal(b64[dec], b32[dec])
    wy = np.array([1, 3, 3, 1])
    pad = np.pad(b, ((1, 1), (1, 1)), mode="edge")
    rows = sum(wy[k] * pad[k:k + 512:2] for k in range(4))[:256]
    cols = sum(wy[k] * rows[:, k:k + 1024:2] for k in range(4))[:, :512]
    on = (257 * cols + 32) % (256 * 64) == 0
    on[0, :] = on[-1, :] = False                  # (edge rows and columns have renormalised weights of their own)
    on[:, 0] = on[:, -1] = False
    assert on.sum() > 20, on.sum()
    assert not dec[on].any()


def test_margin_constant_matches_the_library():
    """ks_float_eps (csrc/ipx_ks_host.cpp) and this file use the same constant, and the library refuses outputs with more than 100 taps."""
    src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "imageprocessor_amd", "csrc", "ipx_ks_host.cpp")).read()
    assert "(nx + ny + 4) * (1.0 / 16777216.0)" in src and "nx + ny > 100 ? 0.f" in src
    assert math.isclose(float(np.float32(13 / 16777216.0)), 13 / 16777216.0, rel_tol=1e-7)

#!/usr/bin/env python3
"""Writes tests/golden/kats.json: known answers for the pixel hot path.

The reference (sj-shoff/ImageProcessor) has no tests or fixtures and cannot be run here
(Go, no toolchain), so these vectors are NOT reference output -- PARITY UNPINNED.  They are
derived two ways, both independent of oracle/ipx_oracle.c and of the HIP kernels:

  * "hand": literal numbers worked out from the published formulas (SURVEY.md section 8c,
    K1-K10) in integer / exact-fraction arithmetic;
  * "model": a pure-Python loop model of the same routines (Python floats are IEEE doubles
    and CPython never fuses a*b+c), run on small seeded inputs.

The JSON layout takes real Go-generated vectors without change: every case is
{"kind", inputs..., "expect"} with pixel data as flat byte lists.

Run:  python tests/golden/make_kats.py
"""
import json
import os
import random
from fractions import Fraction

HERE = os.path.dirname(os.path.abspath(__file__))
OVER, SRC = 0, 1
M = 0xFFFF
U32 = 0xFFFFFFFF


# ---- pure-Python models ---------------------------------------------------------------------

def trunc_i32(v):
    return int(v)  # Go int32(float64): toward zero


def model_opaque(src, sw, sh):
    return all(src[(y * sw + x) * 4 + 3] == 0xFF for y in range(sh) for x in range(sw))


def model_draw(dst, dw, dh, r, src, sw, sh, sp, op):
    """image/draw.DrawMask, nil mask, RGBA <- RGBA (clip + drawCopyOver / drawCopySrc)."""
    x0, y0, x1, y1 = r
    ox, oy = x0, y0
    spx, spy = sp
    x0, y0, x1, y1 = max(x0, 0), max(y0, 0), min(x1, dw), min(y1, dh)
    if x0 >= x1 or y0 >= y1:
        return
    bx0, by0, bx1, by1 = ox - spx, oy - spy, sw + ox - spx, sh + oy - spy
    x0, y0, x1, y1 = max(x0, bx0), max(y0, by0), min(x1, bx1), min(y1, by1)
    if x0 >= x1 or y0 >= y1:
        return
    spx += x0 - ox
    spy += y0 - oy
    for y in range(y1 - y0):
        for x in range(x1 - x0):
            di = ((y0 + y) * dw + x0 + x) * 4
            si = ((spy + y) * sw + spx + x) * 4
            if op == SRC:
                dst[di:di + 4] = src[si:si + 4]
                continue
            sa = src[si + 3] * 0x101
            a = ((M - sa) * 0x101) & U32
            for c in range(4):
                s = src[si + c] * 0x101
                dst[di + c] = ((((dst[di + c] * a) & U32) // M + s) >> 8) & 0xFF


def new_distrib(dw, sw):
    """x/image/draw newDistrib for BiLinear = &Kernel{1, tent}: per destination index (contribs, 1/total, 1/total/0xffff)."""
    import math
    scale = float(sw) / float(dw)
    half_width, kernel_arg_scale = 1.0, 1.0
    if scale > 1:
        half_width *= scale
        kernel_arg_scale = 1 / scale
    out = []
    for x in range(dw):
        center = (float(x) + 0.5) * scale - 0.5
        i = max(int(math.floor(center - half_width)), 0)
        j = int(math.ceil(center + half_width))
        if j > sw:
            j = max(sw, i)
        contribs, total = [], 0.0
        for coord in range(i, j):
            t = abs((center - float(coord)) * kernel_arg_scale)
            if t >= 1.0:
                continue
            w = 1 - t
            if w == 0:
                continue
            total += w
            contribs.append((coord, w))
        total = 1 / total
        out.append((contribs, total, total / 0xFFFF))
    return out


def ftou(f):
    i = int(0xFFFF * f + 0.5)
    return 0xFFFF if i > 0xFFFF else (i if i > 0 else 0)


def model_kernel_scale(dst, dw, dh, dr, tap, sr, op, alpha_one=False):
    """kernelScaler.Scale after its preamble: scaleX_<type> into tmp, then scaleY_RGBA_{Src,Over}.
    tap(x, y) -> 16-bit premultiplied RGBA of the source pixel; op already switched by opaque()."""
    ax0, ay0, ax1, ay1 = max(dr[0], 0), max(dr[1], 0), min(dr[2], dw), min(dr[3], dh)
    if ax0 >= ax1 or ay0 >= ay1 or sr[0] >= sr[2] or sr[1] >= sr[3]:
        return
    ax0, ax1, ay0, ay1 = ax0 - dr[0], ax1 - dr[0], ay0 - dr[1], ay1 - dr[1]
    zdw, zdh, zsw, zsh = dr[2] - dr[0], dr[3] - dr[1], sr[2] - sr[0], sr[3] - sr[1]
    hz, vt = new_distrib(zdw, zsw), new_distrib(zdh, zsh)
    tmp = []
    for y in range(zsh):
        for contribs, _, itw_ffff in hz:
            p = [0.0, 0.0, 0.0, 0.0]
            for coord, w in contribs:
                t = tap(sr[0] + coord, sr[1] + y)
                for c in range(4):
                    p[c] += float(t[c]) * w
            q = [v * itw_ffff for v in p]
            if alpha_one:
                q[3] = 1.0
            tmp.append(q)
    for dx in range(ax0, ax1):
        for dy in range(ay0, ay1):
            contribs, itw, _ = vt[dy]
            p = [0.0, 0.0, 0.0, 0.0]
            for coord, w in contribs:
                t = tmp[coord * zdw + dx]
                for c in range(4):
                    p[c] += t[c] * w
            for c in range(3):
                if p[c] > p[3]:
                    p[c] = p[3]
            q = [ftou(v * itw) for v in p]
            di = ((dr[1] + dy) * dw + dr[0] + dx) * 4
            if op == SRC:
                for c in range(4):
                    dst[di + c] = (q[c] >> 8) & 0xFF
            else:
                pa1 = ((0xFFFF - q[3]) * 0x101) & U32
                for c in range(4):
                    dst[di + c] = ((((dst[di + c] * pa1) & U32) // 0xFFFF + q[c]) >> 8) & 0xFF


def model_scale(dst, dw, dh, dr, src, sw, sh, sr, op):
    """x/image/draw BiLinear.Scale (Kernel.Scale) for *image.RGBA <- *image.RGBA: scaleX_RGBA + scaleY_RGBA_{Src,Over}.
    Equal sizes are NOT simplified to Copy (only nnInterpolator / ablInterpolator.Scale do that)."""
    if op == OVER and model_opaque(src, sw, sh):
        op = SRC

    def tap(x, y):
        i = (y * sw + x) * 4
        return [src[i] * 0x101, src[i + 1] * 0x101, src[i + 2] * 0x101, src[i + 3] * 0x101]
    model_kernel_scale(dst, dw, dh, dr, tap, sr, op)


def model_glyphs(dst, dw, dh, glyphs, col):
    """image/draw drawGlyphOver per glyph, uint32 wrap-around kept."""
    sr, sg, sb, sa = [v * 0x101 for v in col]
    for g in glyphs:
        x0, y0, x1, y1 = g["dr"]
        ox, oy = x0, y0
        mpx, mpy = g["mp"]
        mw, mh = g["mw"], g["mh"]
        x0, y0, x1, y1 = max(x0, 0), max(y0, 0), min(x1, dw), min(y1, dh)
        if x0 >= x1 or y0 >= y1:
            continue
        bx0, by0, bx1, by1 = ox - mpx, oy - mpy, mw + ox - mpx, mh + oy - mpy
        x0, y0, x1, y1 = max(x0, bx0), max(y0, by0), min(x1, bx1), min(y1, by1)
        if x0 >= x1 or y0 >= y1:
            continue
        mpx += x0 - ox
        mpy += y0 - oy
        for y in range(y1 - y0):
            for x in range(x1 - x0):
                ma = g["mask"][(mpy + y) * mw + mpx + x]
                if ma == 0:
                    continue
                ma |= ma << 8
                a = ((M - (sa * ma) // M) * 0x101) & U32
                di = ((y0 + y) * dw + x0 + x) * 4
                for c, s in enumerate((sr, sg, sb, sa)):
                    dst[di + c] = ((((dst[di + c] * a + s * ma) & U32) // M) >> 8) & 0xFF


# ---- source-type variants: *image.NRGBA and *image.YCbCr sources ------------------------

def tap_nrgba(src, sw, x, y):
    i = (y * sw + x) * 4
    a = src[i + 3] * 0x101
    return [src[i] * a // 0xFF, src[i + 1] * a // 0xFF, src[i + 2] * a // 0xFF, a]


def chroma_shape(w, h, ratio):
    cw = (w + 1) // 2 if ratio in (1, 2) else w
    ch = (h + 1) // 2 if ratio in (2, 3) else h
    return cw, ch


def coff(ratio, cw, x, y):
    return {0: y * cw + x, 1: y * cw + x // 2, 2: (y // 2) * cw + x // 2, 3: (y // 2) * cw + x}[ratio]


def tap_ycbcr(img, x, y):
    """color.YCbCr.RGBA as x/image/draw inlines it: 16-bit, clamped, alpha 0xffff."""
    w, cw = img["w"], chroma_shape(img["w"], img["h"], img["ratio"])[0]
    ci = coff(img["ratio"], cw, x, y)
    yy1 = img["y"][y * w + x] * 0x10101
    cb1, cr1 = img["cb"][ci] - 128, img["cr"][ci] - 128
    out = [(yy1 + 91881 * cr1) >> 8, (yy1 - 22554 * cb1 - 46802 * cr1) >> 8, (yy1 + 116130 * cb1) >> 8]
    return [min(max(v, 0), 0xFFFF) for v in out] + [0xFFFF]


def ycbcr_to_rgb8(img, x, y):
    """color.YCbCrToRGB (imageutil.DrawYCbCr): 8-bit with the overflow trick."""
    w, cw = img["w"], chroma_shape(img["w"], img["h"], img["ratio"])[0]
    ci = coff(img["ratio"], cw, x, y)
    yy1 = img["y"][y * w + x] * 0x10101
    cb1, cr1 = img["cb"][ci] - 128, img["cr"][ci] - 128
    out = []
    for v in (yy1 + 91881 * cr1, yy1 - 22554 * cb1 - 46802 * cr1, yy1 + 116130 * cb1):
        out.append(v >> 16 if 0 <= v < (1 << 24) else (0 if v < 0 else 255))
    return out + [255]


def model_scale_taps(dst, dw, dh, dr, tap, sw, sh, sr, op, alpha_one=False):
    """Kernel.Scale with a tap function returning 16-bit premultiplied RGBA (scaleX_NRGBA / scaleX_YCbCr4xx / scaleX_Image)."""
    model_kernel_scale(dst, dw, dh, dr, tap, sr, op, alpha_one)


def model_draw_nrgba(dst, dw, dh, r, src, sw, sh, sp, op):
    x0, y0, x1, y1 = r
    ox, oy = x0, y0
    spx, spy = sp
    x0, y0, x1, y1 = max(x0, 0), max(y0, 0), min(x1, dw), min(y1, dh)
    x0, y0, x1, y1 = max(x0, ox - spx), max(y0, oy - spy), min(x1, sw + ox - spx), min(y1, sh + oy - spy)
    if x0 >= x1 or y0 >= y1:
        return
    spx += x0 - ox
    spy += y0 - oy
    for y in range(y1 - y0):
        for x in range(x1 - x0):
            di = ((y0 + y) * dw + x0 + x) * 4
            t = tap_nrgba(src, sw, spx + x, spy + y)
            if op == SRC:
                for c in range(4):
                    dst[di + c] = (t[c] >> 8) & 0xFF
            else:
                a = ((M - t[3]) * 0x101) & U32
                for c in range(4):
                    dst[di + c] = ((((dst[di + c] * a) & U32) // M + t[c]) >> 8) & 0xFF


def rnd_ycbcr(rng, w, h, ratio):
    cw, ch = chroma_shape(w, h, ratio)
    return {"w": w, "h": h, "ratio": ratio, "y": [rng.randrange(256) for _ in range(w * h)],
            "cb": [rng.randrange(256) for _ in range(cw * ch)], "cr": [rng.randrange(256) for _ in range(cw * ch)]}


# ---- cases ------------------------------------------------------------------------------------

def rnd_frame(rng, w, h, opaque=False, premul=True):
    out = []
    for _ in range(w * h):
        a = 255 if opaque else rng.randrange(256)
        if premul:
            out += [rng.randrange(a + 1), rng.randrange(a + 1), rng.randrange(a + 1), a]
        else:
            out += [rng.randrange(256), rng.randrange(256), rng.randrange(256), a]
    return out


def scale_case(name, origin, sw, sh, src, dw, dh, sr=None, dr=None, op=OVER, dst=None, expect=None):
    sr = list(sr or (0, 0, sw, sh))
    dr = list(dr or (0, 0, dw, dh))
    dst0 = list(dst) if dst is not None else [0] * (dw * dh * 4)
    if expect is None:
        expect = list(dst0)
        model_scale(expect, dw, dh, dr, src, sw, sh, sr, op)
    return {"kind": "scale", "name": name, "origin": origin, "sw": sw, "sh": sh, "src": src,
            "dw": dw, "dh": dh, "sr": sr, "dr": dr, "op": op, "dst": dst0, "expect": expect}


def main():
    rng = random.Random(0x1F00D)
    cases = []

    # K1: a constant frame scales to the same constant, any geometry (hand)
    for (sw, sh, dw, dh, v) in [(7, 5, 3, 4, 200), (3, 3, 8, 8, 1), (27, 11, 5, 2, 255), (4, 4, 3, 3, 0)]:
        src = [v] * (sw * sh * 4)
        cases.append(scale_case("K1 constant %d %dx%d->%dx%d" % (v, sw, sh, dw, dh), "hand",
                                sw, sh, src, dw, dh, expect=[v] * (dw * dh * 4)))

    # K2: 2x2 -> 1x1: scale 2, centre 0.5, both taps at t = 0.25 with weight 0.75 on either axis: the plain mean.
    # taps 10,20,30,41: mean*257 = 6489.25; ftou: int(6489.25 + 0.5) = 6489 -> >> 8 = 25 (hand)
    src = [10] * 4 + [20] * 4 + [30] * 4 + [41] * 4
    cases.append(scale_case("K2 2x2->1x1", "hand", 2, 2, src, 1, 1, expect=[25] * 4))

    def tent_axis(dw_, sw_):
        """the tent weights of one axis in exact rationals: per destination index [(coord, weight)], weights summing to 1"""
        import math
        scale = Fraction(sw_, dw_)
        hw, kas = (scale, 1 / scale) if scale > 1 else (Fraction(1), Fraction(1))
        out = []
        for x in range(dw_):
            center = (Fraction(x) + Fraction(1, 2)) * scale - Fraction(1, 2)
            i = max(math.floor(center - hw), 0)
            j = min(math.ceil(center + hw), sw_)
            ws = [(c, 1 - abs(center - c) * kas) for c in range(i, max(i, j)) if abs(center - c) * kas < 1]
            tot = sum(w for _, w in ws)
            out.append([(c, w / tot) for c, w in ws])
        return out

    def hand_scale(src, sw_, sh_, dw_, dh_):
        """exact value of the two-pass tent scaler (opaque source, Src), then ftou's int(v + 0.5) >> 8"""
        hx, vy = tent_axis(dw_, sw_), tent_axis(dh_, sh_)
        out = []
        for dy in range(dh_):
            for dx in range(dw_):
                for c in range(4):
                    v = sum(wy * sum(wx * src[(y * sw_ + x) * 4 + c] * 257 for x, wx in hx[dx]) for y, wy in vy[dy])
                    assert abs((v + Fraction(1, 2)) - round(v + Fraction(1, 2))) > Fraction(1, 10**6), "too close to a rounding step to call by hand"
                    out.append(min(int(v + Fraction(1, 2)), 0xFFFF) >> 8)
        return out

    # K2b: 4 -> 2 columns (scale 2): dx = 0 sees columns 0,1,2 with tent weights .75,.75,.25 (column -1 does not exist, so the
    # weights are renormalised by 1/1.75): (3a + 3b + c)/7; dx = 1 sees 1,2,3 with .25,.75,.75: (b + 3c + 3d)/7 (hand).
    # The 2-tap ApproxBiLinear would give (a+b)/2, (c+d)/2 here -- this case tells the two apart.
    src = [7] * 3 + [255] + [70] * 3 + [255] + [140] * 3 + [255] + [250] * 3 + [255]
    exp = [(int(Fraction(3 * 7 + 3 * 70 + 140, 7) * 257 + Fraction(1, 2)) >> 8)] * 3 + [255] + \
          [(int(Fraction(70 + 3 * 140 + 3 * 250, 7) * 257 + Fraction(1, 2)) >> 8)] * 3 + [255]
    assert exp == [53, 53, 53, 255, 177, 177, 177, 255] and exp == hand_scale(src, 4, 1, 2, 1)
    cases.append(scale_case("K2b 4x1->2x1 tent weights (3a+3b+c)/7", "hand", 4, 1, src, 2, 1, expect=None))
    assert cases[-1]["expect"] == exp, (cases[-1]["expect"], exp)

    # K3: 1920 -> 1024 columns (scale 1.875): 3 or 4 taps per column, in exact rationals (hand, via Fraction); the float64 model
    # must agree wherever the exact value is not within 1e-6 of a rounding step (asserted inside hand_scale)
    sw, dw = 1920, 1024
    row = [(x * 7 + 3) & 0xFF for x in range(sw)]
    src = []
    for x in range(sw):
        src += [row[x], (row[x] * 3) & 0xFF, 255 - row[x], 255]
    exp = hand_scale(src, sw, 1, dw, 1)
    cases.append(scale_case("K3 1920x1->1024x1 ramp", "hand~", sw, 1, src, dw, 1, expect=None))
    assert cases[-1]["expect"] == exp
    ntaps = [len(t) for t in tent_axis(1024, 1920)]
    assert min(ntaps) == 3 and max(ntaps) == 4 and max(len(t) for t in tent_axis(200, 1080)) == 11

    # K4: upscale 2 -> 5 columns (scale 0.4, support stays 1): columns 0 and 4 see one tap, 1..3 blend the two (hand, rationals)
    src = [0, 0, 0, 255, 200, 100, 50, 255]
    exp = hand_scale(src, 2, 1, 5, 1)
    cases.append(scale_case("K4 2x1->5x1 edges", "hand~", 2, 1, src, 5, 1, expect=None))
    assert cases[-1]["expect"] == exp, (cases[-1]["expect"], exp)
    assert exp[0:4] == [0, 0, 0, 255] and exp[16:20] == [200, 100, 50, 255]

    # K4b: the thumbnail's geometry in small: 27 -> 5 on both axes (scale 5.4, up to 11 x 11 taps per pixel), opaque (hand, rationals)
    src = []
    for i in range(27 * 27):
        v = (i * 37 + 11) & 0xFF
        src += [v, (v * 5 + 1) & 0xFF, 255 - v, 255]
    exp = hand_scale(src, 27, 27, 5, 5)
    cases.append(scale_case("K4b 27x27->5x5 (x5.4 as 1080->200)", "hand~", 27, 27, src, 5, 5, expect=None))
    assert cases[-1]["expect"] == exp

    # K4c: equal sizes are an identity for premultiplied pixels (one tap of weight 1 per axis; NOT a Copy: an invalid pixel with
    # a colour above its alpha is clamped to the alpha by scaleY's "if pr > pa") (hand)
    src = [10, 20, 30, 40, 200, 100, 50, 255, 0, 0, 0, 0, 90, 60, 30, 20]
    cases.append(scale_case("K4c equal size: identity, colour clamped to alpha", "hand", 4, 1, src, 4, 1,
                            expect=[10, 20, 30, 40, 200, 100, 50, 255, 0, 0, 0, 0, 20, 20, 20, 20]))
    chk = [0] * 16
    model_scale(chk, 4, 1, [0, 0, 4, 1], src, 4, 1, [0, 0, 4, 1], OVER)
    assert chk == cases[-1]["expect"], chk

    # model-derived scale cases
    def add_model(name, sw, sh, dw, dh, opaque=False, **kw):
        cases.append(scale_case(name, "model", sw, sh, rnd_frame(rng, sw, sh, opaque), dw, dh, **kw))

    add_model("down 13x9->5x4 alpha", 13, 9, 5, 4)
    add_model("down 27x27->5x5 (x5.4 as 1080->200)", 27, 27, 5, 5, opaque=True)
    add_model("up 5x4->13x11", 5, 4, 13, 11)
    add_model("crop sr 16x10 (3,2,11,10)->4x4", 16, 10, 4, 4, sr=(3, 2, 11, 10))
    add_model("1-wide source 1x6->4x3", 1, 6, 4, 3)
    add_model("1x1 source ->3x3", 1, 1, 3, 3)
    add_model("down 32x18->17x10 src op", 32, 18, 17, 10, op=SRC)
    add_model("dr inside dst", 9, 7, 12, 10, dr=(2, 1, 9, 8))
    add_model("dr clipped by dst", 9, 7, 6, 6, dr=(-3, -2, 9, 8))
    add_model("equal size over a used frame", 6, 5, 6, 5, dst=rnd_frame(rng, 6, 5))
    add_model("equal size crop", 12, 9, 5, 5, sr=(4, 2, 9, 7))
    add_model("over onto non-zero dst, translucent src", 11, 8, 6, 5, dst=rnd_frame(rng, 6, 5))
    add_model("over onto non-zero dst, opaque src (switches to Src)", 11, 8, 6, 5, opaque=True,
              dst=rnd_frame(rng, 6, 5))
    add_model("src op onto non-zero dst", 11, 8, 6, 5, op=SRC, dst=rnd_frame(rng, 6, 5))

    # draw (DrawMask without a mask)
    for name, op, r, sp in [("draw src full", SRC, (0, 0, 8, 6), (0, 0)),
                            ("draw over offset", OVER, (2, 1, 7, 6), (1, 0)),
                            ("draw over clipped", OVER, (-2, -1, 12, 9), (0, 0)),
                            ("draw src clipped by source", SRC, (3, 3, 8, 6), (5, 4))]:
        sw, sh, dw, dh = 8, 6, 8, 6
        src = rnd_frame(rng, sw, sh)
        dst0 = rnd_frame(rng, dw, dh)
        exp = list(dst0)
        model_draw(exp, dw, dh, r, src, sw, sh, sp, op)
        cases.append({"kind": "draw", "name": name, "origin": "model", "sw": sw, "sh": sh,
                      "src": src, "dw": dw, "dh": dh, "r": list(r), "sp": list(sp), "op": op,
                      "dst": dst0, "expect": exp})

    # K5: geometry (hand)
    for ow, oh, w, h, keep, nw, nh in [(854, 480, 1024, 768, 1, 1024, 575), (1080, 1920, 1024, 768, 1, 432, 768),
                                       (333, 500, 1024, 768, 1, 511, 768), (1920, 1080, 1024, 768, 1, 1024, 576),
                                       (640, 480, 1024, 768, 1, 1024, 768), (1920, 1080, 1024, 768, 0, 1024, 768),
                                       (3840, 2160, 1024, 768, 1, 1024, 576), (7680, 4320, 1024, 768, 1, 1024, 576)]:
        cases.append({"kind": "resize_dims", "origin": "hand", "ow": ow, "oh": oh, "w": w, "h": h,
                      "keep_aspect": keep, "expect": [nw, nh]})
    for ow, oh, size, crop, rect, nw, nh in [(640, 480, 200, 1, (80, 0, 560, 480), 200, 200),
                                             (854, 480, 200, 1, (187, 0, 667, 480), 200, 200),
                                             (1920, 1080, 200, 1, (420, 0, 1500, 1080), 200, 200),
                                             (3840, 2160, 200, 1, (840, 0, 3000, 2160), 200, 200),
                                             (7680, 4320, 200, 1, (1680, 0, 6000, 4320), 200, 200),
                                             (1080, 1920, 200, 1, (0, 420, 1080, 1500), 200, 200),
                                             (500, 500, 200, 1, (0, 0, 500, 500), 200, 200),
                                             (1920, 1080, 200, 0, (0, 0, 1920, 1080), 355, 200),
                                             (1080, 1920, 200, 0, (0, 0, 1080, 1920), 200, 355)]:
        cases.append({"kind": "thumb_geometry", "origin": "hand", "ow": ow, "oh": oh, "size": size,
                      "crop_to_fit": crop, "expect": {"crop": list(rect), "nw": nw, "nh": nh}})
    # watermark.go:116-148 with width 300, 36 pt (height 44 = ceil(36*64*1.2 / 64)) on 1920x1080
    cases.append({"kind": "text_height", "origin": "hand", "font_size": 36, "expect": 44})
    cases.append({"kind": "text_height", "origin": "hand", "font_size": 12.5, "expect": 15})
    for pos, x, y in [("top-left", 20, 64), ("top-right", 1600, 64), ("top-center", 810, 64),
                      ("bottom-left", 20, 1060), ("bottom-right", 1600, 1060),
                      ("bottom-center", 810, 1060), ("center", 810, 562), ("nonsense", 1600, 1060)]:
        cases.append({"kind": "anchor", "origin": "hand", "position": pos, "w": 1920, "h": 1080,
                      "width_px": 300, "height_px": 44, "expect": [x, y]})
    for s, op_, rgba, err in [("255,255,255", 0.5, (255, 255, 255, 127), False),
                              ("10, 20 ,30", 1.0, (10, 20, 30, 255), False),
                              ("300,-5,7,64", 0.5, (255, 0, 7, 64), False),
                              ("1,2,3,x", 0.25, (1, 2, 3, 63), False),
                              ("1,2", 0.5, (0, 0, 0, 127), True),
                              ("red,0,0", 0.5, (0, 0, 0, 127), True),
                              ("", 0.5, (0, 0, 0, 127), True),
                              ("1,2,3,4,5", 0.5, (0, 0, 0, 127), True)]:
        cases.append({"kind": "parse_color", "origin": "hand", "s": s, "opacity": op_,
                      "expect": {"rgba": list(rgba), "error": err}})

    # K6-K10: drawGlyphOver with the default colour (255,255,255,127) (hand)
    col = [255, 255, 255, 127]
    ladder = [(0, 255), (1, 0), (2, 1), (3, 1), (4, 2), (128, 64), (254, 127), (255, 128)]
    d = []
    e = []
    for dv, ev in ladder:
        d += [dv, dv, dv, dv]
        e += [ev, ev, ev, ev]
    e[3] = 127  # K6: alpha of d=0 is 32639 >> 8
    # alpha channel: (d*a + sa*ma)/m >> 8 with sa = 32639 -- worked by the model below, while the
    # RGB ladder is the SURVEY's hand-derived K8 list
    g = {"mask": [255] * len(ladder), "mw": len(ladder), "mh": 1, "dr": [0, 0, len(ladder), 1], "mp": [0, 0]}
    chk = list(d)
    model_glyphs(chk, len(ladder), 1, [g], col)
    for i in range(len(ladder)):
        assert chk[i * 4:i * 4 + 3] == e[i * 4:i * 4 + 3], (i, chk[i * 4:i * 4 + 4], e[i * 4:i * 4 + 4])
        e[i * 4 + 3] = chk[i * 4 + 3]
    assert e[3] == 127
    cases.append({"kind": "glyphs", "name": "K6-K8 full-coverage ladder", "origin": "hand",
                  "dw": len(ladder), "dh": 1, "dst": d, "glyphs": [g], "col": col, "expect": e})
    # K9/K10 partial coverage, d in (0,100,255), RGB from SURVEY; alpha from the model
    for mv, rgb in [(0, (0, 100, 255)), (1, (1, 101, 0)), (64, (64, 152, 32)), (128, (128, 203, 64)),
                    (200, (200, 5, 100))]:
        d = [0, 0, 0, 0, 100, 100, 100, 100, 255, 255, 255, 255]
        g = {"mask": [mv] * 3, "mw": 3, "mh": 1, "dr": [0, 0, 3, 1], "mp": [0, 0]}
        chk = list(d)
        model_glyphs(chk, 3, 1, [g], col)
        for i in range(3):
            assert chk[i * 4:i * 4 + 3] == [rgb[i]] * 3, (mv, i, chk)
        cases.append({"kind": "glyphs", "name": "K9 mask=%d" % mv, "origin": "hand", "dw": 3, "dh": 1,
                      "dst": d, "glyphs": [g], "col": col, "expect": chk})

    # model-derived glyph runs: clipping, mask points, overlapping glyphs applied in order
    dw, dh = 24, 12
    dst0 = rnd_frame(rng, dw, dh, opaque=True)
    glyphs = []
    for (mw, mh, dr, mp) in [(7, 9, (1, 2, 8, 11), (0, 0)), (6, 8, (6, 3, 12, 11), (0, 0)),
                             (8, 8, (-3, -2, 5, 6), (0, 0)), (9, 7, (18, 8, 27, 15), (0, 0)),
                             (10, 10, (10, 0, 16, 6), (2, 3)), (5, 5, (11, 1, 16, 6), (0, 2))]:
        mask = [rng.choice([0, 0, 255, rng.randrange(256)]) for _ in range(mw * mh)]
        glyphs.append({"mask": mask, "mw": mw, "mh": mh, "dr": list(dr), "mp": list(mp)})
    for col_ in ([255, 255, 255, 127], [0, 0, 0, 127], [12, 200, 99, 255], [40, 30, 20, 64]):
        exp = list(dst0)
        model_glyphs(exp, dw, dh, glyphs, col_)
        cases.append({"kind": "glyphs", "name": "clip+overlap col=%s" % col_, "origin": "model",
                      "dw": dw, "dh": dh, "dst": dst0, "glyphs": glyphs, "col": col_, "expect": exp})

    # ---- NRGBA and YCbCr sources ---------------------------------------------------------------------
    # hand: NRGBA (255,128,0,128) under Src: a16 = 32896; r = 255*32896/255 = 32896 -> 128;
    # g = 128*32896/255 = 16512 -> 64; b = 0; a -> 128.
    cases.append({"kind": "draw_nrgba", "name": "hand premultiply", "origin": "hand", "sw": 1, "sh": 1,
                  "src": [255, 128, 0, 128], "dw": 1, "dh": 1, "r": [0, 0, 1, 1], "sp": [0, 0], "op": SRC,
                  "dst": [9, 9, 9, 9], "expect": [128, 64, 0, 128]})
    for name, op in (("nrgba draw src", SRC), ("nrgba draw over", OVER)):
        src = rnd_frame(rng, 7, 5, premul=False)
        dst0 = rnd_frame(rng, 9, 6)
        exp = list(dst0)
        model_draw_nrgba(exp, 9, 6, (1, 1, 10, 8), src, 7, 5, (0, 1), op)
        cases.append({"kind": "draw_nrgba", "name": name, "origin": "model", "sw": 7, "sh": 5, "src": src, "dw": 9,
                      "dh": 6, "r": [1, 1, 10, 8], "sp": [0, 1], "op": op, "dst": dst0, "expect": exp})
    for name, sw, sh, dw, dh, op, opaque, used, sr in (("nrgba down over zero dst", 13, 9, 5, 4, OVER, False, False, None),
                                                         ("nrgba up src", 5, 4, 11, 9, SRC, False, True, None),
                                                         ("nrgba over used dst", 12, 8, 7, 5, OVER, False, True, (1, 1, 11, 7)),
                                                         ("nrgba opaque over used dst -> Src", 12, 8, 7, 5, OVER, True, True, None)):
        src = rnd_frame(rng, sw, sh, opaque=opaque, premul=False)
        dst0 = rnd_frame(rng, dw, dh) if used else [0] * (dw * dh * 4)
        sr_ = list(sr or (0, 0, sw, sh))
        exp = list(dst0)
        op_eff = SRC if (op == OVER and opaque) else op
        model_scale_taps(exp, dw, dh, (0, 0, dw, dh), lambda x, y: tap_nrgba(src, sw, x, y), sw, sh, sr_, op_eff)
        cases.append({"kind": "scale_nrgba", "name": name, "origin": "model", "sw": sw, "sh": sh, "src": src, "dw": dw,
                      "dh": dh, "sr": sr_, "dr": [0, 0, dw, dh], "op": op, "dst": dst0, "expect": exp})
    # hand: grey (Cb = Cr = 128) converts to R = G = B = Y in both the 8-bit and the 16-bit formula;
    # and Y=76, Cb=85, Cr=255: r = 5000268 + 91881*127 = 16669155 -> 254; g = 26236 -> 0; b = 6678 -> 0.
    grey = {"w": 4, "h": 4, "ratio": 2, "y": [(i * 17) & 255 for i in range(16)], "cb": [128] * 4, "cr": [128] * 4}
    cases.append({"kind": "draw_ycbcr", "name": "hand grey 4:2:0", "origin": "hand", "img": grey, "dw": 4, "dh": 4,
                  "r": [0, 0, 4, 4], "sp": [0, 0], "dst": [0] * 64,
                  "expect": sum([[v, v, v, 255] for v in grey["y"]], [])})
    red = {"w": 1, "h": 1, "ratio": 0, "y": [76], "cb": [85], "cr": [255]}
    cases.append({"kind": "draw_ycbcr", "name": "hand red", "origin": "hand", "img": red, "dw": 1, "dh": 1,
                  "r": [0, 0, 1, 1], "sp": [0, 0], "dst": [0] * 4, "expect": [254, 0, 0, 255]})
    for ratio in (0, 1, 2, 3):
        img = rnd_ycbcr(rng, 11, 9, ratio)
        dst0 = rnd_frame(rng, 11, 9)
        exp = list(dst0)
        r, sp = (2, 1, 10, 8), (1, 1)          # odd source origin: exercises the x/2, y/2 chroma indexing
        for y in range(r[3] - r[1]):
            for x in range(r[2] - r[0]):
                di = ((r[1] + y) * 11 + r[0] + x) * 4
                exp[di:di + 4] = ycbcr_to_rgb8(img, sp[0] + x, sp[1] + y)
        cases.append({"kind": "draw_ycbcr", "name": "draw ratio %d" % ratio, "origin": "model", "img": img, "dw": 11,
                      "dh": 9, "r": list(r), "sp": list(sp), "dst": dst0, "expect": exp})
        for dw, dh, sr in ((5, 4, (0, 0, 11, 9)), (13, 14, (0, 0, 11, 9)), (4, 3, (1, 1, 10, 8))):
            exp = [0] * (dw * dh * 4)
            model_scale_taps(exp, dw, dh, (0, 0, dw, dh), lambda x, y: tap_ycbcr(img, x, y), 11, 9, list(sr), SRC, alpha_one=True)
            cases.append({"kind": "scale_ycbcr", "name": "scale ratio %d -> %dx%d" % (ratio, dw, dh), "origin": "model",
                          "img": img, "dw": dw, "dh": dh, "sr": list(sr), "dr": [0, 0, dw, dh], "dst": [0] * (dw * dh * 4),
                          "expect": exp})

    with open(os.path.join(HERE, "kats.json"), "w") as f:
        json.dump({"note": "hand/model-derived known answers; NOT reference output (parity unpinned)",
                   "cases": cases}, f, separators=(",", ":"))
    print("wrote %d cases" % len(cases))


if __name__ == "__main__":
    main()

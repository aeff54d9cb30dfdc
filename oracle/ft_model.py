"""ft_model.py -- TEST INFRASTRUCTURE: a pure-Python model of the glyph mask producer.

A second, independent transcription of the published algorithm of github.com/golang/freetype @ e2365dfdc4a0
(go.mod:42; the module is not under /root/reference and there is no Go toolchain: PARITY UNPINNED), used only
by tests/ to cross-check csrc/ipx_font.cpp.  Differences in construction that make the cross-check worth
something: the font tables are read by fontTools (not by a hand-written parser), numbers are Python ints with
explicit int32 wrapping, the cell store is a dict per scanline (not linked lists), spans are painted from a
sorted cell list.  Function names follow the Go sources:

  truetype/truetype.go  (f *Font) scale, HMetric, Kern          -> Font.scale / hmetric / kern
  truetype/glyph.go     GlyphBuf.Load / load / loadCompound     -> Font.load_glyph
  freetype.go           drawContour, rasterize, glyph, DrawString -> draw_contour / rasterize / draw_string
  raster/raster.go      Add1, Add2, scan, areaToAlpha, Rasterize -> Raster.*
  reference call sites  operations/watermark.go:98-118,151
"""
import numpy as np


def _i32(v):
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v & 0x80000000 else v


def _div(a, b):  # Go / C integer division: truncation toward zero
    q = abs(a) // abs(b)
    return q if (a < 0) == (b < 0) else -q


def _mod(a, b):  # Go / C remainder: sign of the dividend
    return a - b * _div(a, b)


class Font:
    def __init__(self, path):
        from fontTools.ttLib import TTFont
        self.tt = TTFont(path, lazy=False)
        self.upem = self.tt["head"].unitsPerEm
        self.order = self.tt.getGlyphOrder()
        self.glyf = self.tt["glyf"]
        self.hmtx = self.tt["hmtx"]
        # the subtable Go's parseCmap settles on: first Unicode (0,3)/(0,4) wins, else the last Microsoft one
        pick = None
        for t in self.tt["cmap"].tables:
            key = (t.platformID, t.platEncID)
            if key in ((0, 3), (0, 4)):
                pick = t
                break
            if key in ((3, 0), (3, 1), (3, 10)):
                pick = t
        self.cmap = pick.cmap
        self.kern_pairs = {}
        if "kern" in self.tt and self.tt["kern"].kernTables:
            k0 = self.tt["kern"].kernTables[0]
            self.kern_pairs = {(self.tt.getGlyphID(a), self.tt.getGlyphID(b)): v for (a, b), v in k0.kernTable.items()}
        h = self.tt["head"]
        self.bounds = (h.xMin, h.yMin, h.xMax, h.yMax)

    def scale(self, x):
        if x >= 0:
            x += self.upem // 2
        else:
            x -= self.upem // 2
        return _div(x, self.upem)

    def index(self, rune):
        name = self.cmap.get(rune)
        return self.tt.getGlyphID(name) if name else 0

    def kern(self, scale, i0, i1):
        v = self.kern_pairs.get((i0, i1))
        return self.scale(_i32(scale * v)) if v is not None else 0

    def load_glyph(self, scale, gid):
        """-> (points [(x, y, on)], ends, advance), scaled 26.6, y up: GlyphBuf.Load with font.HintingNone"""
        state = {"metrics_set": False, "pp1x": 0, "phantom": (0, 0)}
        pts, ends = [], []
        self._load(scale, gid, True, pts, ends, state, 0)
        if state["pp1x"]:
            pts = [(x - state["pp1x"], y, on) for x, y, on in pts]
        return pts, ends, state["phantom"][1] - state["phantom"][0]

    def _load(self, scale, gid, use_my_metrics, pts, ends, st, depth):
        assert depth < 32
        name = self.order[gid]
        g = self.glyf[name]
        adv, lsb = self.hmtx[name]
        xmin = getattr(g, "xMin", 0) if g.numberOfContours != 0 else 0
        ph = (self.scale(_i32(scale * (xmin - lsb))), self.scale(_i32(scale * (xmin - lsb + adv))))
        if g.numberOfContours == 0:
            st["phantom"] = ph
            return
        if g.numberOfContours < 0:
            saved_outer = None
            for c in g.components:
                saved = st["phantom"]
                np0 = len(pts)
                umm = bool(c.flags & 0x200)
                self._load(scale, self.tt.getGlyphID(c.glyphName), use_my_metrics and umm, pts, ends, st, depth + 1)
                if not umm:
                    st["phantom"] = saved
                if hasattr(c, "transform"):
                    # F2Dot14 matrix [[xx, xy], [yx, yy]] as fontTools holds it; Go keeps the raw int16s
                    t = [int(round(v * 16384)) for v in (c.transform[0][0], c.transform[0][1], c.transform[1][0], c.transform[1][1])]
                    for j in range(np0, len(pts)):
                        x, y, on = pts[j]
                        nx = ((x * t[0] + (1 << 13)) >> 14) + ((y * t[2] + (1 << 13)) >> 14)
                        ny = ((x * t[1] + (1 << 13)) >> 14) + ((y * t[3] + (1 << 13)) >> 14)
                        pts[j] = (nx, ny, on)
                dx, dy = self.scale(_i32(scale * c.x)), self.scale(_i32(scale * c.y))
                if c.flags & 0x4:
                    dx, dy = (dx + 32) & ~63, (dy + 32) & ~63
                for j in range(np0, len(pts)):
                    x, y, on = pts[j]
                    pts[j] = (x + dx, y + dy, on)
            if not st["metrics_set"]:
                st["phantom"] = ph
            pp = ph[0]
        else:
            np0 = len(pts)
            coords, endpts, flags = g.coordinates, g.endPtsOfContours, g.flags
            for (x, y), fl in zip(coords, flags):
                pts.append((self.scale(_i32(scale * x)), self.scale(_i32(scale * y)), fl & 1))
            ends.extend(np0 + e + 1 for e in endpts)
            pp = ph[0]
            if use_my_metrics:
                st["phantom"] = ph
        if use_my_metrics and not st["metrics_set"]:
            st["metrics_set"] = True
            st["pp1x"] = pp


class Raster:
    def __init__(self, width, height):
        self.width, self.height = max(width, 0), max(height, 0)
        ss2 = 32
        if self.width > 24 or self.height > 24:
            ss2 *= 2
            if self.width > 120 or self.height > 120:
                ss2 *= 2
        self.ss2 = ss2
        self.rows = [dict() for _ in range(self.height)]   # xi -> [area, cover]
        self.a = (0, 0)
        self.xi = self.yi = 0
        self.area = self.cover = 0

    def save_cell(self):
        if self.area or self.cover:
            if 0 <= self.yi < self.height:
                xi = -1 if self.xi < 0 else min(self.xi, self.width)
                c = self.rows[self.yi].setdefault(xi, [0, 0])
                c[0] += self.area
                c[1] += self.cover
            self.area = self.cover = 0

    def set_cell(self, xi, yi):
        if (xi, yi) != (self.xi, self.yi):
            self.save_cell()
            self.xi, self.yi = xi, yi

    def scan(self, yi, x0, y0f, x1, y1f):
        x0i = _div(x0, 64); x0f = x0 - 64 * x0i
        x1i = _div(x1, 64); x1f = x1 - 64 * x1i
        if y0f == y1f:
            self.set_cell(x1i, yi)
            return
        dx, dy = x1 - x0, y1f - y0f
        if x0i == x1i:
            self.area += (x0f + x1f) * dy
            self.cover += dy
            return
        if dx > 0:
            p, q, edge0, edge1, step = (64 - x0f) * dy, dx, 0, 64, 1
        else:
            p, q, edge0, edge1, step = x0f * dy, -dx, 64, 0, -1
        ydelta, yrem = _div(p, q), _mod(p, q)
        if yrem < 0:
            ydelta -= 1; yrem += q
        xi, y = x0i, y0f
        self.area += (x0f + edge1) * ydelta
        self.cover += ydelta
        xi += step; y += ydelta
        self.set_cell(xi, yi)
        if xi != x1i:
            p = 64 * (y1f - y + ydelta)
            full, frem = _div(p, q), _mod(p, q)
            if frem < 0:
                full -= 1; frem += q
            yrem -= q
            while xi != x1i:
                ydelta = full
                yrem += frem
                if yrem >= 0:
                    ydelta += 1; yrem -= q
                self.area += 64 * ydelta
                self.cover += ydelta
                xi += step; y += ydelta
                self.set_cell(xi, yi)
        ydelta = y1f - y
        self.area += (edge0 + x1f) * ydelta
        self.cover += ydelta

    def start(self, p):
        self.set_cell(_div(p[0], 64), _div(p[1], 64))
        self.a = p

    def add1(self, b):
        x0, y0 = self.a
        x1, y1 = b
        dx, dy = x1 - x0, y1 - y0
        y0i = _div(y0, 64); y0f = y0 - 64 * y0i
        y1i = _div(y1, 64); y1f = y1 - 64 * y1i
        if y0i == y1i:
            self.scan(y0i, x0, y0f, x1, y1f)
        elif dx == 0:
            edge0, edge1, step = (0, 64, 1) if dy > 0 else (64, 0, -1)
            x0i = _div(x0, 64)
            x0f2 = (x0 - 64 * x0i) * 2
            yi = y0i
            dcover = edge1 - y0f
            self.area += x0f2 * dcover; self.cover += dcover
            yi += step
            self.set_cell(x0i, yi)
            dcover = edge1 - edge0
            while yi != y1i:
                self.area += x0f2 * dcover; self.cover += dcover
                yi += step
                self.set_cell(x0i, yi)
            dcover = y1f - edge0
            self.area += x0f2 * dcover; self.cover += dcover
        else:
            if dy > 0:
                p, q, edge0, edge1, step = (64 - y0f) * dx, dy, 0, 64, 1
            else:
                p, q, edge0, edge1, step = y0f * dx, -dy, 64, 0, -1
            xdelta, xrem = _div(p, q), _mod(p, q)
            if xrem < 0:
                xdelta -= 1; xrem += q
            x, yi = x0, y0i
            self.scan(yi, x, y0f, x + xdelta, edge1)
            x += xdelta; yi += step
            self.set_cell(_div(x, 64), yi)
            if yi != y1i:
                p = 64 * dx
                full, frem = _div(p, q), _mod(p, q)
                if frem < 0:
                    full -= 1; frem += q
                xrem -= q
                while yi != y1i:
                    xdelta = full
                    xrem += frem
                    if xrem >= 0:
                        xdelta += 1; xrem -= q
                    self.scan(yi, x, edge0, x + xdelta, edge1)
                    x += xdelta; yi += step
                    self.set_cell(_div(x, 64), yi)
            self.scan(yi, x, edge0, x1, y1f)
        self.a = b

    def add2(self, b, c):
        a = self.a
        dev = _div(max(abs(a[0] - 2 * b[0] + c[0]), abs(a[1] - 2 * b[1] + c[1])), self.ss2)
        nsplit = 0
        while dev > 0:
            dev = _div(dev, 4); nsplit += 1
        # recursive form of Go's explicit stack: left half first, each leaf = two chords through the midpoint
        def rec(p0, p1, p2, s):   # p0 = END point, p2 = start point (Go's pStack order)
            if s > 0:
                m1 = (_div(p0[0] + p1[0], 2), _div(p0[1] + p1[1], 2))
                m3 = (_div(p2[0] + p1[0], 2), _div(p2[1] + p1[1], 2))
                m2 = (_div(m1[0] + m3[0], 2), _div(m1[1] + m3[1], 2))
                rec(m2, m3, p2, s - 1)   # the half nearer the start is drawn first
                rec(p0, m1, m2, s - 1)
            else:
                mid = (_div(p0[0] + 2 * p1[0] + p2[0], 4), _div(p0[1] + 2 * p1[1] + p2[1], 4))
                self.add1(mid)
                self.add1(p0)
        rec(c, b, a, nsplit)

    @staticmethod
    def area_to_alpha(area):
        a = abs((area + 1) >> 1) & 0x1FFF
        if a > 0x1000:
            a = 0x2000 - a
        elif a == 0x1000:
            a = 0x0FFF
        return (a << 4 | a >> 8) & 0xFFFF

    def rasterize(self, mw, mh):
        self.save_cell()
        mask = np.zeros((mh, mw), np.uint8)

        def paint(y, x0, x1, alpha):
            x0, x1 = max(x0, 0), min(x1, self.width)
            if alpha and x0 < x1 and 0 <= y < mh:
                mask[y, max(x0, 0):min(x1, mw)] = alpha >> 8
        for yi, row in enumerate(self.rows):
            xi, cover = 0, 0
            for cx in sorted(row):
                area, cov = row[cx]
                if cover and cx > xi:
                    paint(yi, xi, cx, self.area_to_alpha(cover * 128))
                cover += cov
                paint(yi, cx, cx + 1, self.area_to_alpha(cover * 128 - area))
                xi = cx + 1
        return mask


def draw_contour(r, ps, dx, dy):
    if not ps:
        return
    start = (dx + ps[0][0], dy - ps[0][1])
    if ps[0][2]:
        others = ps[1:]
    else:
        last = (dx + ps[-1][0], dy - ps[-1][1])
        if ps[-1][2]:
            start, others = last, ps[:-1]
        else:
            start, others = (_div(start[0] + last[0], 2), _div(start[1] + last[1], 2)), ps
    r.start(start)
    q0, on0 = start, True
    for x, y, on in others:
        q = (dx + x, dy - y)
        if on:
            if on0:
                r.add1(q)
            else:
                r.add2(q0, q)
        elif not on0:
            r.add2(q0, (_div(q0[0] + q[0], 2), _div(q0[1] + q[1], 2)))
        q0, on0 = q, on
    if on0:
        r.add1(start)
    else:
        r.add2(q0, start)


def context_scale(size):
    return int(size * 72.0 * (64.0 / 72.0))


def face_scale(size):
    return int(0.5 + (size * 72.0 * 64 / 72))


def rasterize(font, scale, gid, fx, fy):
    """-> (advance, mask, (offx, offy)): (c *Context) rasterize"""
    pts, ends, adv = font.load_glyph(scale, gid)
    if pts:
        bx0, bx1 = min(p[0] for p in pts), max(p[0] for p in pts)
        by0, by1 = min(p[1] for p in pts), max(p[1] for p in pts)
    else:
        bx0 = bx1 = by0 = by1 = 0
    xmin, ymin = (fx + bx0) >> 6, (fy - by1) >> 6
    xmax, ymax = (fx + bx1 + 0x3F) >> 6, (fy - by0 + 0x3F) >> 6
    fx -= xmin << 6
    fy -= ymin << 6
    b = [font.scale(_i32(scale * v)) for v in font.bounds]
    rw = ((b[2] + 63) >> 6) - (b[0] >> 6)
    rh = ((-(b[1] - 63)) >> 6) - ((-b[3]) >> 6)
    r = Raster(rw, rh)
    e0 = 0
    for e1 in ends:
        draw_contour(r, pts[e0:e1], fx, fy)
        e0 = e1
    return adv, r.rasterize(xmax - xmin, ymax - ymin), (xmin, ymin)


def text_width(font, text, size):
    scale = face_scale(size)
    w = sum(font.load_glyph(scale, font.index(ord(ch)))[2] for ch in text)
    return w, (w + 63) >> 6


def draw_string(font, text, size, px, py, clip_w, clip_h):
    """-> ([{"mask", "dr", "mp"}], end X 26.6): (c *Context) DrawString at freetype.Pt(px, py)"""
    scale = context_scale(size)
    X, Y = px << 6, py << 6
    prev = None
    out = []
    # (c *Context) glyph: 256 * 4 * 1 cache slots per Context (one per DrawString here: addTextWatermark makes a fresh Context,
    # watermark.go:98).  A hit needs `e.valid && e.glyph == glyph` only, so a repeated glyph whose fx lands in the same quarter-pixel
    # bucket reuses the mask, offset and advance rasterised at the FIRST fx.
    cache = {}
    for ch in text:
        idx = font.index(ord(ch))
        if prev is not None:
            X += font.kern(scale, prev, idx)
        fx, fy = X & 63, Y & 63
        t = ((fx // 16) * 1 + fy // 64) * 256 + idx % 256
        if t not in cache or cache[t][0] != idx:
            cache[t] = (idx, rasterize(font, scale, idx, fx, fy))
        adv, mask, (ox, oy) = cache[t][1]
        gx0, gy0 = ox + (X >> 6), oy + (Y >> 6)
        X += adv
        mh, mw = mask.shape
        dr = (max(gx0, 0), max(gy0, 0), min(gx0 + mw, clip_w), min(gy0 + mh, clip_h))
        if mw > 0 and mh > 0 and dr[0] < dr[2] and dr[1] < dr[3]:
            out.append({"mask": mask, "dr": dr, "mp": (0, dr[1] - gy0)})
        prev = idx
    return out, X

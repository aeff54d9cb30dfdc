// ipx_font.cpp -- the glyph mask producer of the watermark.
//
// What the reference does (operations/watermark.go): NewWatermarker parses a TrueType font
// (truetype.Parse, :29-38); addTextWatermark measures the text with a truetype face
// (face.GlyphAdvance per rune, NO kerning, :105-115) and draws it with a fresh freetype.Context
// (DPI 72, no hinting, :98-104, c.DrawString(text, pt) :151).  DrawString makes one
// draw.DrawMask(dst, dr, uniform, ZP, *image.Alpha mask, mp, Over) call per rune; this file produces
// exactly that list of (mask, dr, mp) -- the ipx_glyph array the composite kernels consume -- on the
// host.  Glyphs are tiny (16 runes of ~20 x 40 px for the default text) and cached, so this is host
// work by design; the per-pixel composite is the GPU's.
//
// The arithmetic lives in github.com/golang/freetype @ e2365dfdc4a0 (go.mod:42), which is NOT under
// /root/reference.  Restated here from its published algorithm, function by function:
//   truetype/truetype.go  Parse, parseCmap/Head/Hhea/Kern/Maxp, Index, HMetric, Kern, scale, Bounds
//   truetype/glyph.go     GlyphBuf.Load, load, loadSimple, loadCompound, addPhantomsAndScale (HintingNone)
//   truetype/face.go      NewFace (scale = Int26_6(0.5 + size*dpi*64/72)), GlyphAdvance
//   freetype.go           Context.recalc (scale = Int26_6(size*dpi*64/72)), drawContour, rasterize,
//                         glyph (4 x-subpixel positions), DrawString, Pt
//   raster/raster.go      Rasterizer.SetBounds, Start, Add1, Add2, scan, areaToAlpha (even-odd), Rasterize
//   raster/paint.go       AlphaSrcPainter.Paint
// PARITY UNPINNED: no Go toolchain and no copy of that module or of the Go Regular face exist in the
// build image, so nothing here could be compared with the Go library's output; tests pin it against
// hand-derivable coverage values, an independent Python model and FreeType (Pillow) within tolerance.
//
// Bytecode hinting (font.HintingFull) is not built: the reference never enables it.
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/ipx.h"

namespace ipx { void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2))); }

namespace {

typedef int32_t f26;  // fixed.Int26_6 (an int32 in Go: products wrap)

inline f26 mul26(f26 a, f26 b) { return (f26)((uint32_t)a * (uint32_t)b); }

struct Slice {
    const uint8_t *p = nullptr;
    size_t n = 0;
    bool has(size_t off, size_t len) const { return off <= n && len <= n - off; }
    uint32_t u16(size_t i) const { return (uint32_t)p[i] << 8 | p[i + 1]; }
    uint32_t u32(size_t i) const { return (uint32_t)p[i] << 24 | (uint32_t)p[i + 1] << 16 | (uint32_t)p[i + 2] << 8 | p[i + 3]; }
    int32_t i16(size_t i) const { return (int16_t)u16(i); }
};

struct CMapRange { uint32_t start, end, delta, offset; };

struct GlyphPoint { f26 x, y; uint32_t flags; };

struct HMetric { f26 advance, lsb; };

}  // namespace

// truetype.Font: table slices and the values parse* derive from them
struct ipx_font {
    std::vector<uint8_t> data;
    Slice cmap, glyf, head, hhea, hmtx, kern, loca, maxp, cmap_indexes;
    std::vector<CMapRange> cm;
    int loca_long = 0, n_glyph = 0, n_hmetric = 0, n_kern = 0;
    int32_t units_per_em = 0;
    f26 bx0 = 0, by0 = 0, bx1 = 0, by1 = 0;  // head.{xMin,yMin,xMax,yMax}, FUnits
    int32_t ascent = 0, descent = 0;

    // rasterised masks: freetype.Context.cache keeps them per Context (one per addTextWatermark call);
    // a mask depends only on (context scale, glyph, fx, fy), so sharing them across calls changes nothing
    struct Mask { int w = 0, h = 0, offx = 0, offy = 0; f26 advance = 0; std::vector<uint8_t> pix; };
    std::mutex mu;
    std::map<std::tuple<f26, uint32_t, f26, f26>, std::shared_ptr<Mask>> masks;

    // (f *Font) scale: x / fUnitsPerEm rounded to nearest, half away from zero
    f26 scale(f26 x) const
    {
        if (x >= 0) x += units_per_em / 2;
        else x -= units_per_em / 2;
        return x / units_per_em;
    }
};

namespace {

// ---- truetype.Parse --------------------------------------------------------------------------
struct ParseError { std::string text; };

void fail(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
void fail(const char *fmt, ...)
{
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw ParseError{buf};
}

void parse_cmap(ipx_font &f)
{
    const Slice &c = f.cmap;
    if (c.n < 4) fail("freetype: invalid TrueType format: cmap too short");
    const int nsub = (int)c.u16(2);
    if (c.n < (size_t)(8 * nsub + 4)) fail("freetype: invalid TrueType format: cmap too short");
    size_t offset = 0;
    bool found = false;
    for (int i = 0, x = 4; i < nsub; i++, x += 8) {
        const uint32_t pid_psid = c.u32(x), o = c.u32(x + 4);
        // the Unicode encodings win outright; a Microsoft one is kept unless Unicode shows up later
        if (pid_psid == 0x00000003 || pid_psid == 0x00000004) { offset = o; found = true; break; }
        if (pid_psid == 0x00030000 || pid_psid == 0x00030001 || pid_psid == 0x0003000a) { offset = o; found = true; }
    }
    if (!found) fail("freetype: unsupported TrueType feature: cmap encoding");
    if (offset <= 0 || offset > c.n) fail("freetype: invalid TrueType format: bad cmap offset");
    if (!c.has(offset, 2)) fail("freetype: invalid TrueType format: bad cmap offset");
    const uint32_t format = c.u16(offset);
    if (format == 4) {
        if (!c.has(offset, 14)) fail("freetype: invalid TrueType format: cmap too short");
        if (c.u16(offset + 4) != 0) fail("freetype: unsupported TrueType feature: language: %u", c.u16(offset + 4));
        const int seg_x2 = (int)c.u16(offset + 6);
        if (seg_x2 % 2 == 1) fail("freetype: invalid TrueType format: bad segCountX2: %d", seg_x2);
        const int seg = seg_x2 / 2;
        offset += 14;
        if (!c.has(offset, (size_t)8 * seg + 2)) fail("freetype: invalid TrueType format: cmap too short");
        f.cm.assign(seg, CMapRange{0, 0, 0, 0});
        for (int i = 0; i < seg; i++, offset += 2) f.cm[i].end = c.u16(offset);
        offset += 2;
        for (int i = 0; i < seg; i++, offset += 2) f.cm[i].start = c.u16(offset);
        for (int i = 0; i < seg; i++, offset += 2) f.cm[i].delta = c.u16(offset);
        for (int i = 0; i < seg; i++, offset += 2) f.cm[i].offset = c.u16(offset);
        f.cmap_indexes.p = c.p + offset;
        f.cmap_indexes.n = c.n - offset;
        return;
    }
    if (format == 12) {
        if (!c.has(offset, 16)) fail("freetype: invalid TrueType format: cmap too short");
        if (c.u16(offset + 2) != 0) fail("freetype: invalid TrueType format: cmap format");
        const uint32_t length = c.u32(offset + 4), language = c.u32(offset + 8), ngroups = c.u32(offset + 12);
        if (language != 0) fail("freetype: unsupported TrueType feature: language: %u", language);
        if (length != 12 * ngroups + 16) fail("freetype: invalid TrueType format: inconsistent cmap length");
        offset += 16;
        if (!c.has(offset, (size_t)12 * ngroups)) fail("freetype: invalid TrueType format: cmap too short");
        f.cm.assign(ngroups, CMapRange{0, 0, 0, 0});
        for (uint32_t i = 0; i < ngroups; i++, offset += 12) {
            f.cm[i].start = c.u32(offset);
            f.cm[i].end = c.u32(offset + 4);
            f.cm[i].delta = c.u32(offset + 8) - f.cm[i].start;
        }
        return;
    }
    fail("freetype: unsupported TrueType feature: cmap format: %u", format);
}

void parse_head(ipx_font &f)
{
    if (f.head.n != 54) fail("freetype: invalid TrueType format: bad head length: %zu", f.head.n);
    f.units_per_em = (int32_t)f.head.u16(18);
    if (f.units_per_em == 0) fail("freetype: invalid TrueType format: unitsPerEm 0");
    f.bx0 = f.head.i16(36); f.by0 = f.head.i16(38); f.bx1 = f.head.i16(40); f.by1 = f.head.i16(42);
    const uint32_t fmt = f.head.u16(50);
    if (fmt == 0) f.loca_long = 0;
    else if (fmt == 1) f.loca_long = 1;
    else fail("freetype: invalid TrueType format: bad indexToLocFormat: %u", fmt);
}

void parse_hhea(ipx_font &f)
{
    if (f.hhea.n != 36) fail("freetype: invalid TrueType format: bad hhea length: %zu", f.hhea.n);
    f.ascent = f.hhea.i16(4);
    f.descent = f.hhea.i16(6);
    f.n_hmetric = (int)f.hhea.u16(34);
    if ((size_t)(4 * f.n_hmetric + 2 * (f.n_glyph - f.n_hmetric)) != f.hmtx.n)
        fail("freetype: invalid TrueType format: bad hmtx length: %zu", f.hmtx.n);
}

void parse_kern(ipx_font &f)
{
    // only the "old" (Windows) table layout and only its first, horizontal format-0 subtable, as Go
    if (f.kern.n == 0) return;
    if (f.kern.n < 18) fail("freetype: invalid TrueType format: kern data too short");
    if (f.kern.u16(0) != 0) fail("freetype: unsupported TrueType feature: kern version: %u", f.kern.u16(0));
    if (f.kern.u16(2) == 0) fail("freetype: unsupported TrueType feature: kern nTables: 0");
    const int length = (int)f.kern.u16(6);
    const uint32_t coverage = f.kern.u16(8);
    if (coverage != 0x0001) fail("freetype: unsupported TrueType feature: kern coverage: 0x%04x", coverage);
    f.n_kern = (int)f.kern.u16(10);
    if (6 * f.n_kern != length - 14) fail("freetype: invalid TrueType format: bad kern table length");
    if (!f.kern.has(18, (size_t)6 * f.n_kern)) fail("freetype: invalid TrueType format: bad kern table length");
}

void parse_maxp(ipx_font &f)
{
    if (f.maxp.n != 32) fail("freetype: invalid TrueType format: bad maxp length: %zu", f.maxp.n);
    f.n_glyph = (int)f.maxp.u16(4);
}

void parse_font(ipx_font &f)
{
    Slice d{f.data.data(), f.data.size()};
    if (d.n < 12) fail("freetype: invalid TrueType format: TTF data is too short");
    size_t offset = 0;
    uint32_t magic = d.u32(0);
    if (magic == 0x74746366) {  // "ttcf": the first font of the collection (truetype.Parse == parse(ttf, 0))
        if (d.n < 16) fail("freetype: invalid TrueType format: TTC data is too short");
        const uint32_t nfonts = d.u32(8);
        if (nfonts == 0) fail("freetype: invalid TrueType format: bad TTC count");
        offset = d.u32(12);
        if (!d.has(offset, 12)) fail("freetype: invalid TrueType format: bad TTC offset");
        magic = d.u32(offset);
    }
    if (magic == 0x4f54544f) fail("freetype: unsupported TrueType feature: OpenType fonts with CFF data");
    if (magic != 0x00010000) fail("freetype: invalid TrueType format: bad TTF version");
    const int ntab = (int)d.u16(offset + 4);
    if (!d.has(offset + 12, (size_t)16 * ntab)) fail("freetype: invalid TrueType format: TTF data is too short");
    for (int i = 0; i < ntab; i++) {
        const size_t x = offset + 12 + 16 * (size_t)i;
        const uint32_t tag = d.u32(x), to = d.u32(x + 8), tl = d.u32(x + 12);
        if (!d.has(to, tl)) fail("freetype: invalid TrueType format: bad table offset or length");
        const Slice s{d.p + to, tl};
        switch (tag) {
        case 0x636d6170: f.cmap = s; break;
        case 0x676c7966: f.glyf = s; break;
        case 0x68656164: f.head = s; break;
        case 0x68686561: f.hhea = s; break;
        case 0x686d7478: f.hmtx = s; break;
        case 0x6b65726e: f.kern = s; break;
        case 0x6c6f6361: f.loca = s; break;
        case 0x6d617870: f.maxp = s; break;
        default: break;  // cvt, fpgm, prep, hdmx, vmtx, name, OS/2: hinting / metrics the watermark never reads
        }
    }
    parse_head(f);
    parse_maxp(f);
    parse_cmap(f);
    parse_kern(f);
    parse_hhea(f);
    const size_t need = f.loca_long ? 4 * ((size_t)f.n_glyph + 1) : 2 * ((size_t)f.n_glyph + 1);
    if (f.loca.n < need) fail("freetype: invalid TrueType format: bad loca length");
}

// (f *Font) Index
uint32_t glyph_index(const ipx_font &f, uint32_t c)
{
    const int n = (int)f.cm.size();
    for (int i = 0, j = n; i < j;) {
        const int h = i + (j - i) / 2;
        const CMapRange &r = f.cm[h];
        if (c < r.start) j = h;
        else if (r.end < c) i = h + 1;
        else if (r.offset == 0) return (c + r.delta) & 0xffffu;
        else {
            const long long off = (long long)r.offset + 2 * ((long long)h - n + (long long)(c - r.start));
            if (off >= 0 && (size_t)off + 2 <= f.cmap_indexes.n) return f.cmap_indexes.u16((size_t)off);
            return 0;
        }
    }
    return 0;
}

HMetric unscaled_hmetric(const ipx_font &f, uint32_t i)
{
    const int j = (int)i;
    if (j < 0 || f.n_glyph <= j) return HMetric{0, 0};
    if (j >= f.n_hmetric) {
        const int p = 4 * (f.n_hmetric - 1);
        return HMetric{(f26)f.hmtx.u16(p), (f26)f.hmtx.i16(p + 2 * j + 2)};
    }
    return HMetric{(f26)f.hmtx.u16(4 * j), (f26)f.hmtx.i16(4 * j + 2)};
}

// (f *Font) Kern
f26 kern(const ipx_font &f, f26 scale, uint32_t i0, uint32_t i1)
{
    if (f.n_kern == 0) return 0;
    const uint32_t g = i0 << 16 | i1;
    int lo = 0, hi = f.n_kern;
    while (lo < hi) {
        const int i = (lo + hi) / 2;
        const uint32_t ig = f.kern.u32(18 + 6 * i);
        if (ig < g) lo = i + 1;
        else if (ig > g) hi = i;
        else return f.scale(mul26(scale, (f26)f.kern.i16(22 + 6 * i)));
    }
    return 0;
}

// ---- truetype.GlyphBuf (font.HintingNone) ----------------------------------------------------
struct GlyphBuf {
    const ipx_font *font = nullptr;
    f26 scale = 0;
    std::vector<GlyphPoint> points;
    std::vector<int> ends;
    f26 advance = 0;
    f26 bx0 = 0, by0 = 0, bx1 = 0, by1 = 0;  // control box, y up
    f26 pp1x = 0;
    GlyphPoint phantom[4];
    bool metrics_set = false;

    void add_phantoms_and_scale(size_t np1)
    {
        for (int k = 0; k < 4; k++) points.push_back(phantom[k]);
        for (size_t i = np1; i < points.size(); i++) {
            points[i].x = font->scale(mul26(scale, points[i].x));
            points[i].y = font->scale(mul26(scale, points[i].y));
        }
    }

    void load_simple(const Slice &g, int ne)
    {
        size_t offset = 10;
        if (!g.has(offset, (size_t)2 * ne + 2)) fail("freetype: invalid TrueType format: glyf data too short");
        const size_t np0 = points.size();
        for (int i = 0; i < ne; i++, offset += 2) ends.push_back(1 + (int)g.u16(offset));
        const int np = ne > 0 ? ends.back() : 0;  // points of this glyph
        if (np < 0 || np > 65536) fail("freetype: invalid TrueType format: bad point count");
        points.resize(np0 + np, GlyphPoint{0, 0, 0});
        const size_t instr = g.u16(offset);
        offset += 2;
        if (!g.has(offset, instr)) fail("freetype: invalid TrueType format: glyf data too short");
        offset += instr;   // the bytecode program is not run without hinting
        // flags, with repeats
        for (int i = 0; i < np;) {
            if (!g.has(offset, 1)) fail("freetype: invalid TrueType format: glyf data too short");
            const uint32_t c = g.p[offset++];
            points[np0 + i++].flags = c;
            if (c & 0x08) {
                if (!g.has(offset, 1)) fail("freetype: invalid TrueType format: glyf data too short");
                int count = g.p[offset++];
                for (; count > 0; count--) {
                    if (i >= np) fail("freetype: invalid TrueType format: glyf flag repeat overrun");
                    points[np0 + i++].flags = c;
                }
            }
        }
        // x then y deltas: 0x02/0x04 = one byte, 0x10/0x20 = positive byte or (if long) "same as previous"
        int16_t x = 0;
        for (int i = 0; i < np; i++) {
            const uint32_t fl = points[np0 + i].flags;
            if (fl & 0x02) {
                if (!g.has(offset, 1)) fail("freetype: invalid TrueType format: glyf data too short");
                const int16_t dx = g.p[offset++];
                x = (int16_t)(fl & 0x10 ? x + dx : x - dx);
            } else if (!(fl & 0x10)) {
                if (!g.has(offset, 2)) fail("freetype: invalid TrueType format: glyf data too short");
                x = (int16_t)(x + (int16_t)g.u16(offset));
                offset += 2;
            }
            points[np0 + i].x = (f26)x;
        }
        int16_t y = 0;
        for (int i = 0; i < np; i++) {
            const uint32_t fl = points[np0 + i].flags;
            if (fl & 0x04) {
                if (!g.has(offset, 1)) fail("freetype: invalid TrueType format: glyf data too short");
                const int16_t dy = g.p[offset++];
                y = (int16_t)(fl & 0x20 ? y + dy : y - dy);
            } else if (!(fl & 0x20)) {
                if (!g.has(offset, 2)) fail("freetype: invalid TrueType format: glyf data too short");
                y = (int16_t)(y + (int16_t)g.u16(offset));
                offset += 2;
            }
            points[np0 + i].y = (f26)y;
        }
    }

    void load_compound(uint32_t recursion, const Slice &g, bool use_my_metrics)
    {
        enum { kWords = 1, kXY = 2, kRound = 4, kScale = 8, kMore = 32, kXYScale = 64, k2x2 = 128, kUseMyMetrics = 512 };
        const size_t np_outer = points.size();
        size_t offset = 10;
        for (;;) {
            if (!g.has(offset, 4)) fail("freetype: invalid TrueType format: glyf data too short");
            const uint32_t flags = g.u16(offset), component = g.u16(offset + 2);
            f26 dx, dy;
            int16_t tr[4] = {0, 0, 0, 0};
            bool has_tr = false;
            if (flags & kWords) {
                if (!g.has(offset, 8)) fail("freetype: invalid TrueType format: glyf data too short");
                dx = g.i16(offset + 4); dy = g.i16(offset + 6);
                offset += 8;
            } else {
                if (!g.has(offset, 6)) fail("freetype: invalid TrueType format: glyf data too short");
                dx = (int8_t)g.p[offset + 4]; dy = (int8_t)g.p[offset + 5];
                offset += 6;
            }
            if (!(flags & kXY)) fail("freetype: unsupported TrueType feature: compound glyph transform vector");
            if (flags & (kScale | kXYScale | k2x2)) {
                has_tr = true;
                if (flags & kScale) {
                    if (!g.has(offset, 2)) fail("freetype: invalid TrueType format: glyf data too short");
                    tr[0] = (int16_t)g.u16(offset); tr[3] = tr[0];
                    offset += 2;
                } else if (flags & kXYScale) {
                    if (!g.has(offset, 4)) fail("freetype: invalid TrueType format: glyf data too short");
                    tr[0] = (int16_t)g.u16(offset); tr[3] = (int16_t)g.u16(offset + 2);
                    offset += 4;
                } else {
                    if (!g.has(offset, 8)) fail("freetype: invalid TrueType format: glyf data too short");
                    for (int k = 0; k < 4; k++) tr[k] = (int16_t)g.u16(offset + 2 * k);
                    offset += 8;
                }
            }
            GlyphPoint saved[4];
            memcpy(saved, phantom, sizeof saved);
            const size_t np0 = points.size();
            load(recursion + 1, component, use_my_metrics && (flags & kUseMyMetrics));
            if (!(flags & kUseMyMetrics)) memcpy(phantom, saved, sizeof saved);
            if (has_tr) {
                for (size_t j = np0; j < points.size(); j++) {
                    GlyphPoint &p = points[j];
                    const f26 nx = (f26)(((int64_t)p.x * tr[0] + (1 << 13)) >> 14) + (f26)(((int64_t)p.y * tr[2] + (1 << 13)) >> 14);
                    const f26 ny = (f26)(((int64_t)p.x * tr[1] + (1 << 13)) >> 14) + (f26)(((int64_t)p.y * tr[3] + (1 << 13)) >> 14);
                    p.x = nx; p.y = ny;
                }
            }
            dx = font->scale(mul26(scale, dx));
            dy = font->scale(mul26(scale, dy));
            if (flags & kRound) { dx = (dx + 32) & ~63; dy = (dy + 32) & ~63; }
            for (size_t j = np0; j < points.size(); j++) { points[j].x += dx; points[j].y += dy; }
            if (!(flags & kMore)) break;
        }
        // only the four phantom points are new and unscaled here: the components arrived scaled
        add_phantoms_and_scale(points.size());
        if (!metrics_set) memcpy(phantom, &points[points.size() - 4], sizeof phantom);
        points.resize(points.size() - 4);
        (void)np_outer;
    }

    void load(uint32_t recursion, uint32_t i, bool use_my_metrics)
    {
        if (recursion >= 32) fail("freetype: unsupported TrueType feature: excessive compound glyph recursion");
        if ((int)i >= font->n_glyph) fail("freetype: invalid TrueType format: glyph index out of range");
        uint32_t g0, g1;
        if (!font->loca_long) { g0 = 2 * font->loca.u16(2 * i); g1 = 2 * font->loca.u16(2 * i + 2); }
        else { g0 = font->loca.u32(4 * (size_t)i); g1 = font->loca.u32(4 * (size_t)i + 4); }
        Slice g;
        int ne = 0;
        f26 bxmin = 0, bymax = 0;
        if ((uint64_t)g0 + 10 <= g1) {
            if (!font->glyf.has(g0, g1 - g0)) fail("freetype: invalid TrueType format: bad glyf offset");
            g = Slice{font->glyf.p + g0, g1 - g0};
            ne = g.i16(0);
            bxmin = g.i16(2);
            bymax = g.i16(8);
        }
        const HMetric uhm = unscaled_hmetric(*font, i);
        // the vertical phantom points (vmtx) never reach a value the watermark reads: only X is kept exact
        phantom[0] = GlyphPoint{bxmin - uhm.lsb, 0, 0};
        phantom[1] = GlyphPoint{bxmin - uhm.lsb + uhm.advance, 0, 0};
        phantom[2] = GlyphPoint{uhm.advance / 2, bymax, 0};
        phantom[3] = GlyphPoint{uhm.advance / 2, bymax, 0};
        f26 pp = 0;
        if (g.n == 0) {
            add_phantoms_and_scale(points.size());
            memcpy(phantom, &points[points.size() - 4], sizeof phantom);
            points.resize(points.size() - 4);
            return;
        }
        if (ne < 0) {
            if (ne != -1) fail("freetype: unsupported TrueType feature: negative number of contours");
            pp = font->scale(mul26(scale, bxmin - uhm.lsb));
            load_compound(recursion, g, use_my_metrics);
        } else {
            const size_t np0 = points.size(), ne0 = ends.size();
            load_simple(g, ne);
            add_phantoms_and_scale(np0);
            pp = points[points.size() - 4].x;
            if (use_my_metrics) memcpy(phantom, &points[points.size() - 4], sizeof phantom);
            points.resize(points.size() - 4);
            if (np0 != 0)
                for (size_t k = ne0; k < ends.size(); k++) ends[k] += (int)np0;
        }
        if (use_my_metrics && !metrics_set) { metrics_set = true; pp1x = pp; }
    }

    // (g *GlyphBuf) Load(f, scale, i, font.HintingNone)
    void Load(const ipx_font *f, f26 sc, uint32_t i)
    {
        font = f; scale = sc;
        points.clear(); ends.clear();
        pp1x = 0; metrics_set = false;
        memset(phantom, 0, sizeof phantom);
        load(0, i, true);
        if (pp1x != 0)
            for (auto &p : points) p.x -= pp1x;
        advance = phantom[1].x - phantom[0].x;
        if (points.empty()) { bx0 = by0 = bx1 = by1 = 0; return; }
        bx0 = bx1 = points[0].x; by0 = by1 = points[0].y;
        for (size_t k = 1; k < points.size(); k++) {
            const GlyphPoint &p = points[k];
            if (bx0 > p.x) bx0 = p.x; else if (bx1 < p.x) bx1 = p.x;
            if (by0 > p.y) by0 = p.y; else if (by1 < p.y) by1 = p.y;
        }
    }
};

// ---- raster.Rasterizer -------------------------------------------------------------------------
struct P26 { f26 x, y; };

struct Rasterizer {
    struct Cell { int xi; long long area, cover; int next; };
    int width = 0, split_scale2 = 32;
    P26 a{0, 0};
    int xi = 0, yi = 0;
    long long area = 0, cover = 0;
    std::vector<Cell> cell;
    std::vector<int> cell_index;

    void SetBounds(int w, int h)
    {
        if (w < 0) w = 0;
        if (h < 0) h = 0;
        int ss2 = 32;  // the C FreeType 2.4.0 heuristic
        if (w > 24 || h > 24) { ss2 *= 2; if (w > 120 || h > 120) ss2 *= 2; }
        width = w; split_scale2 = ss2;
        cell_index.assign(h, -1);
        Clear();
    }
    void Clear()
    {
        a = P26{0, 0}; xi = yi = 0; area = cover = 0;
        cell.clear();
        for (auto &c : cell_index) c = -1;
    }
    int find_cell()
    {
        if (yi < 0 || yi >= (int)cell_index.size()) return -1;
        int x = xi;
        if (x < 0) x = -1;
        else if (x > width) x = width;
        int i = cell_index[yi], prev = -1;
        while (i != -1 && cell[i].xi <= x) {
            if (cell[i].xi == x) return i;
            prev = i; i = cell[i].next;
        }
        const int c = (int)cell.size();
        cell.push_back(Cell{x, 0, 0, i});
        if (prev == -1) cell_index[yi] = c;
        else cell[prev].next = c;
        return c;
    }
    void save_cell()
    {
        if (area != 0 || cover != 0) {
            const int i = find_cell();
            if (i != -1) { cell[i].area += area; cell[i].cover += cover; }
            area = 0; cover = 0;
        }
    }
    void set_cell(int x, int y)
    {
        if (xi != x || yi != y) { save_cell(); xi = x; yi = y; }
    }
    // area / coverage of scanline y for the piece from (x0, y0f) to (x1, y1f), y fractions within the row
    void scan(int y_i, f26 x0, f26 y0f, f26 x1, f26 y1f)
    {
        const int x0i = (int)x0 / 64;
        const f26 x0f = x0 - (f26)(64 * x0i);
        const int x1i = (int)x1 / 64;
        const f26 x1f = x1 - (f26)(64 * x1i);
        if (y0f == y1f) { set_cell(x1i, y_i); return; }
        const f26 dx = x1 - x0, dy = y1f - y0f;
        if (x0i == x1i) { area += (long long)mul26(x0f + x1f, dy); cover += dy; return; }
        f26 p, q, edge0, edge1;
        int xi_delta;
        if (dx > 0) { p = mul26(64 - x0f, dy); q = dx; edge0 = 0; edge1 = 64; xi_delta = 1; }
        else { p = mul26(x0f, dy); q = -dx; edge0 = 64; edge1 = 0; xi_delta = -1; }
        f26 y_delta = p / q, y_rem = p % q;
        if (y_rem < 0) { y_delta -= 1; y_rem += q; }
        int x = x0i;
        f26 y = y0f;
        area += (long long)mul26(x0f + edge1, y_delta);
        cover += y_delta;
        x += xi_delta; y += y_delta;
        set_cell(x, y_i);
        if (x != x1i) {
            p = mul26(64, y1f - y + y_delta);
            f26 full_delta = p / q, full_rem = p % q;
            if (full_rem < 0) { full_delta -= 1; full_rem += q; }
            y_rem -= q;
            while (x != x1i) {
                y_delta = full_delta;
                y_rem += full_rem;
                if (y_rem >= 0) { y_delta += 1; y_rem -= q; }
                area += (long long)mul26(64, y_delta);
                cover += y_delta;
                x += xi_delta; y += y_delta;
                set_cell(x, y_i);
            }
        }
        y_delta = y1f - y;
        area += (long long)mul26(edge0 + x1f, y_delta);
        cover += y_delta;
    }
    void Start(P26 p)
    {
        set_cell((int)(p.x / 64), (int)(p.y / 64));
        a = p;
    }
    void Add1(P26 b)
    {
        const f26 x0 = a.x, y0 = a.y, x1 = b.x, y1 = b.y;
        const f26 dx = x1 - x0, dy = y1 - y0;
        const int y0i = (int)y0 / 64;
        const f26 y0f = y0 - (f26)(64 * y0i);
        const int y1i = (int)y1 / 64;
        const f26 y1f = y1 - (f26)(64 * y1i);
        if (y0i == y1i) {
            scan(y0i, x0, y0f, x1, y1f);
        } else if (dx == 0) {
            // vertical: area / cover directly
            f26 edge0, edge1;
            int yi_delta;
            if (dy > 0) { edge0 = 0; edge1 = 64; yi_delta = 1; }
            else { edge0 = 64; edge1 = 0; yi_delta = -1; }
            const int x0i = (int)x0 / 64;
            int y = y0i;
            const long long x0f2 = ((long long)x0 - 64LL * x0i) * 2;
            long long dcover = edge1 - y0f, darea = x0f2 * dcover;
            area += darea; cover += dcover;
            y += yi_delta;
            set_cell(x0i, y);
            dcover = edge1 - edge0; darea = x0f2 * dcover;
            while (y != y1i) {
                area += darea; cover += dcover;
                y += yi_delta;
                set_cell(x0i, y);
            }
            dcover = y1f - edge0; darea = x0f2 * dcover;
            area += darea; cover += dcover;
        } else {
            f26 p, q, edge0, edge1;
            int yi_delta;
            if (dy > 0) { p = mul26(64 - y0f, dx); q = dy; edge0 = 0; edge1 = 64; yi_delta = 1; }
            else { p = mul26(y0f, dx); q = -dy; edge0 = 64; edge1 = 0; yi_delta = -1; }
            f26 x_delta = p / q, x_rem = p % q;
            if (x_rem < 0) { x_delta -= 1; x_rem += q; }
            f26 x = x0;
            int y = y0i;
            scan(y, x, y0f, x + x_delta, edge1);
            x += x_delta; y += yi_delta;
            set_cell((int)x / 64, y);
            if (y != y1i) {
                p = mul26(64, dx);
                f26 full_delta = p / q, full_rem = p % q;
                if (full_rem < 0) { full_delta -= 1; full_rem += q; }
                x_rem -= q;
                while (y != y1i) {
                    x_delta = full_delta;
                    x_rem += full_rem;
                    if (x_rem >= 0) { x_delta += 1; x_rem -= q; }
                    scan(y, x, edge0, x + x_delta, edge1);
                    x += x_delta; y += yi_delta;
                    set_cell((int)x / 64, y);
                }
            }
            scan(y, x, edge0, x1, y1f);
        }
        a = b;
    }
    // quadratic segment a -> c with control b: split by how far b sits from the chord's midpoint
    void Add2(P26 b, P26 c)
    {
        const f26 ddx = a.x - 2 * b.x + c.x, ddy = a.y - 2 * b.y + c.y;
        const f26 adx = ddx < 0 ? -ddx : ddx, ady = ddy < 0 ? -ddy : ddy;
        f26 dev = (adx > ady ? adx : ady) / (f26)split_scale2;
        int nsplit = 0;
        while (dev > 0) { dev /= 4; nsplit++; }
        const int kMaxNsplit = 16;
        if (nsplit > kMaxNsplit) fail("freetype/raster: Add2 nsplit too large: %d", nsplit);  // Go panics here
        P26 ps[2 * kMaxNsplit + 3];
        int ss[kMaxNsplit + 1];
        int i = 0;
        ss[0] = nsplit;
        ps[0] = c; ps[1] = b; ps[2] = a;
        while (i >= 0) {
            const int s = ss[i];
            P26 *p = ps + 2 * i;
            if (s > 0) {
                const f26 mx = p[1].x;
                p[4].x = p[2].x;
                p[3].x = (p[4].x + mx) / 2;
                p[1].x = (p[0].x + mx) / 2;
                p[2].x = (p[1].x + p[3].x) / 2;
                const f26 my = p[1].y;
                p[4].y = p[2].y;
                p[3].y = (p[4].y + my) / 2;
                p[1].y = (p[0].y + my) / 2;
                p[2].y = (p[1].y + p[3].y) / 2;
                ss[i] = s - 1;
                ss[i + 1] = s - 1;
                i++;
            } else {
                const f26 midx = (p[0].x + 2 * p[1].x + p[2].x) / 4;
                const f26 midy = (p[0].y + 2 * p[1].y + p[2].y) / 4;
                Add1(P26{midx, midy});
                Add1(p[0]);
                i--;
            }
        }
    }
    // even-odd (Rasterizer.UseNonZeroWinding is left false by freetype.NewContext)
    static uint32_t area_to_alpha(long long ar)
    {
        long long v = (ar + 1) >> 1;
        if (v < 0) v = -v;
        uint32_t alpha = (uint32_t)v;
        alpha &= 0x1fff;
        if (alpha > 0x1000) alpha = 0x2000 - alpha;
        else if (alpha == 0x1000) alpha = 0x0fff;
        return alpha << 4 | alpha >> 8;
    }
    // Rasterize(raster.NewAlphaSrcPainter(mask)): spans straight into an A8 mask of mw x mh
    void Rasterize(uint8_t *mask, int mw, int mh)
    {
        save_cell();
        auto paint = [&](int y, int x0, int x1, uint32_t alpha) {
            if (y < 0 || y >= mh) return;
            if (x0 < 0) x0 = 0;
            if (x1 > mw) x1 = mw;
            if (x0 >= x1) return;
            memset(mask + (size_t)y * mw + x0, (int)(alpha >> 8), (size_t)(x1 - x0));
        };
        for (int y = 0; y < (int)cell_index.size(); y++) {
            int x = 0;
            long long cov = 0;
            for (int c = cell_index[y]; c != -1; c = cell[c].next) {
                if (cov != 0 && cell[c].xi > x) {
                    const uint32_t alpha = area_to_alpha(cov * 64 * 2);
                    if (alpha != 0) {
                        int xi0 = x, xi1 = cell[c].xi;
                        if (xi0 < 0) xi0 = 0;
                        if (xi1 >= width) xi1 = width;
                        if (xi0 < xi1) paint(y, xi0, xi1, alpha);
                    }
                }
                cov += cell[c].cover;
                const uint32_t alpha = area_to_alpha(cov * 64 * 2 - cell[c].area);
                x = cell[c].xi + 1;
                if (alpha != 0) {
                    int xi0 = cell[c].xi, xi1 = x;
                    if (xi0 < 0) xi0 = 0;
                    if (xi1 >= width) xi1 = width;
                    if (xi0 < xi1) paint(y, xi0, xi1, alpha);
                }
            }
        }
    }
};

// ---- freetype.Context (the parts DrawString runs) ------------------------------------------------
// (c *Context) drawContour: points are y-up, the rasteriser is y-down, offset (dx, dy)
void draw_contour(Rasterizer &r, const GlyphPoint *ps, int n, f26 dx, f26 dy)
{
    if (n == 0) return;
    P26 start{dx + ps[0].x, dy - ps[0].y};
    const GlyphPoint *others;
    int nothers;
    if (ps[0].flags & 1) { others = ps + 1; nothers = n - 1; }
    else {
        const P26 last{dx + ps[n - 1].x, dy - ps[n - 1].y};
        if (ps[n - 1].flags & 1) { start = last; others = ps; nothers = n - 1; }
        else { start = P26{(start.x + last.x) / 2, (start.y + last.y) / 2}; others = ps; nothers = n; }
    }
    r.Start(start);
    P26 q0 = start;
    bool on0 = true;
    for (int k = 0; k < nothers; k++) {
        const P26 q{dx + others[k].x, dy - others[k].y};
        const bool on = others[k].flags & 1;
        if (on) {
            if (on0) r.Add1(q);
            else r.Add2(q0, q);
        } else if (!on0) {
            r.Add2(q0, P26{(q0.x + q.x) / 2, (q0.y + q.y) / 2});
        }
        q0 = q; on0 = on;
    }
    if (on0) r.Add1(start);
    else r.Add2(q0, start);
}

const int kMaxMaskSide = 8192;  // a guard the reference does not have (it would try to allocate)

// (c *Context) rasterize(glyph, fx, fy), fx / fy in [0, 64)
std::shared_ptr<ipx_font::Mask> rasterize(const ipx_font &f, f26 scale, uint32_t glyph, f26 fx, f26 fy)
{
    GlyphBuf gb;
    gb.Load(&f, scale, glyph);
    const int xmin = (int)(fx + gb.bx0) >> 6, ymin = (int)(fy - gb.by1) >> 6;
    const int xmax = (int)(fx + gb.bx1 + 0x3f) >> 6, ymax = (int)(fy - gb.by0 + 0x3f) >> 6;
    if (xmin > xmax || ymin > ymax) fail("freetype: negative sized glyph");
    if (xmax - xmin > kMaxMaskSide || ymax - ymin > kMaxMaskSide) fail("freetype: glyph mask larger than %d pixels", kMaxMaskSide);
    fx -= (f26)(xmin << 6);
    fy -= (f26)(ymin << 6);
    // (c *Context) recalc: the rasteriser's bounds hold the font's bounding box at this scale
    Rasterizer r;
    {
        const f26 b0x = f.scale(mul26(scale, f.bx0)), b0y = f.scale(mul26(scale, f.by0));
        const f26 b1x = f.scale(mul26(scale, f.bx1)), b1y = f.scale(mul26(scale, f.by1));
        const int rxmin = +(int)b0x >> 6, rymin = -(int)b1y >> 6;
        const int rxmax = +(int)(b1x + 63) >> 6, rymax = -(int)(b0y - 63) >> 6;
        if (rxmax - rxmin > 4 * kMaxMaskSide || rymax - rymin > 4 * kMaxMaskSide) fail("freetype: font size too large");
        r.SetBounds(rxmax - rxmin, rymax - rymin);
    }
    int e0 = 0;
    for (int e1 : gb.ends) {
        if (e1 < e0 || e1 > (int)gb.points.size()) fail("freetype: invalid TrueType format: bad contour end");
        draw_contour(r, gb.points.data() + e0, e1 - e0, fx, fy);
        e0 = e1;
    }
    auto m = std::make_shared<ipx_font::Mask>();
    m->w = xmax - xmin; m->h = ymax - ymin; m->offx = xmin; m->offy = ymin; m->advance = gb.advance;
    m->pix.assign((size_t)m->w * m->h + 1, 0);
    r.Rasterize(m->pix.data(), m->w, m->h);
    return m;
}

// A memo of rasterize(glyph, fx, fy) across calls.  This is NOT freetype's own glyph cache (that one lives per Context and is
// modelled in ipx_font_draw_string, where it changes results); it only saves rasterising the same (glyph, fx, fy) again.
std::shared_ptr<ipx_font::Mask> cached_mask(ipx_font &f, f26 scale, uint32_t glyph, f26 fx, f26 fy)
{
    const auto key = std::make_tuple(scale, glyph, fx, fy);
    {
        std::lock_guard<std::mutex> lk(f.mu);
        auto it = f.masks.find(key);
        if (it != f.masks.end()) return it->second;
    }
    auto m = rasterize(f, scale, glyph, fx, fy);
    std::lock_guard<std::mutex> lk(f.mu);
    if (f.masks.size() > 4096) f.masks.clear();
    f.masks[key] = m;
    return m;
}

// `for _, r := range s`: UTF-8 decoding with U+FFFD for every invalid byte
std::vector<uint32_t> runes(const char *s)
{
    std::vector<uint32_t> out;
    const unsigned char *p = (const unsigned char *)(s ? s : "");
    const size_t n = strlen((const char *)p);
    for (size_t i = 0; i < n;) {
        const unsigned c = p[i];
        if (c < 0x80) { out.push_back(c); i++; continue; }
        int need = 0;
        uint32_t r = 0, lo = 0x80, hi = 0xbf;
        if (c >= 0xc2 && c <= 0xdf) { need = 1; r = c & 0x1f; }
        else if (c >= 0xe0 && c <= 0xef) { need = 2; r = c & 0x0f; if (c == 0xe0) lo = 0xa0; if (c == 0xed) hi = 0x9f; }
        else if (c >= 0xf0 && c <= 0xf4) { need = 3; r = c & 0x07; if (c == 0xf0) lo = 0x90; if (c == 0xf4) hi = 0x8f; }
        else { out.push_back(0xfffd); i++; continue; }
        bool ok = i + (size_t)need < n;   // a truncated sequence is one U+FFFD per byte
        if (ok) {
            for (int k = 1; k <= need; k++) {
                const unsigned cc = p[i + k];
                const uint32_t l = k == 1 ? lo : 0x80, h = k == 1 ? hi : 0xbf;
                if (cc < l || cc > h) { ok = false; break; }
                r = r << 6 | (cc & 0x3f);
            }
        }
        if (!ok) { out.push_back(0xfffd); i++; continue; }
        out.push_back(r);
        i += need + 1;
    }
    return out;
}

// results of the last draw on this thread (goroutines migrate between OS threads, but a result is consumed
// by the call that asked for it before that call returns)
struct DrawResult {
    std::vector<std::shared_ptr<ipx_font::Mask>> keep;
    std::vector<ipx_glyph> glyphs;
};
thread_local DrawResult t_result;

f26 face_scale(double size) { return (f26)(0.5 + (size * 72.0 * 64 / 72)); }        // truetype.NewFace, DPI 72
f26 context_scale(double size) { return (f26)(size * 72.0 * (64.0 / 72.0)); }      // freetype.Context.recalc, DPI 72

bool size_ok(double size)
{
    if (!(size > 0) || size > 2000) { ipx::set_error("font: size %g outside (0, 2000]", size); return false; }
    return true;
}

}  // namespace

extern "C" {

int ipx_font_create(const uint8_t *ttf, size_t len, ipx_font **out)
{
    if (!ttf || !out) { ipx::set_error("ipx_font_create: null argument"); return IPX_ERR_INVALID; }
    *out = nullptr;
    std::unique_ptr<ipx_font> f(new ipx_font);
    f->data.assign(ttf, ttf + len);
    try {
        parse_font(*f);
    } catch (const ParseError &e) {
        ipx::set_error("%s", e.text.c_str());
        return IPX_ERR_INVALID;
    }
    *out = f.release();
    return IPX_OK;
}

void ipx_font_destroy(ipx_font *f) { delete f; }

int ipx_font_glyph_index(const ipx_font *f, uint32_t rune)
{
    return f ? (int)glyph_index(*f, rune) : 0;
}

int ipx_font_glyph_advance(const ipx_font *f, uint32_t rune, double font_size, int32_t *advance26_6)
{
    if (!f || !advance26_6) { ipx::set_error("ipx_font_glyph_advance: null argument"); return IPX_ERR_INVALID; }
    if (!size_ok(font_size)) return IPX_ERR_INVALID;
    try {
        GlyphBuf gb;
        gb.Load(f, face_scale(font_size), glyph_index(*f, rune));
        *advance26_6 = gb.advance;
    } catch (const ParseError &e) {
        ipx::set_error("%s", e.text.c_str());
        return IPX_ERR_INVALID;
    }
    return IPX_OK;
}

int ipx_font_kern(const ipx_font *f, uint32_t rune0, uint32_t rune1, double font_size, int32_t *kern26_6)
{
    if (!f || !kern26_6) { ipx::set_error("ipx_font_kern: null argument"); return IPX_ERR_INVALID; }
    if (!size_ok(font_size)) return IPX_ERR_INVALID;
    *kern26_6 = kern(*f, context_scale(font_size), glyph_index(*f, rune0), glyph_index(*f, rune1));
    return IPX_OK;
}

int ipx_font_text_width(const ipx_font *f, const char *text, double font_size, int32_t *width26_6, int *width_px)
{
    if (!f) { ipx::set_error("font not loaded"); return IPX_ERR_INVALID; }
    if (!size_ok(font_size)) return IPX_ERR_INVALID;
    const f26 scale = face_scale(font_size);
    f26 w = 0;
    GlyphBuf gb;
    for (uint32_t r : runes(text)) {
        try {
            gb.Load(f, scale, glyph_index(*f, r));
            w += gb.advance;   // `if ok`: a glyph that fails to load adds nothing
        } catch (const ParseError &) {
        }
    }
    if (width26_6) *width26_6 = w;
    if (width_px) *width_px = (int)((w + 0x3f) >> 6);   // fixed.Int26_6.Ceil
    return IPX_OK;
}

int ipx_font_draw_string(ipx_font *f, const char *text, double font_size, int px, int py, int clip_w, int clip_h,
                         const ipx_glyph **out, int *n, int32_t *end_x26_6)
{
    if (!f) { ipx::set_error("font not loaded"); return IPX_ERR_INVALID; }
    if (!out || !n) { ipx::set_error("ipx_font_draw_string: null argument"); return IPX_ERR_INVALID; }
    if (!size_ok(font_size)) return IPX_ERR_INVALID;
    DrawResult &res = t_result;
    res.keep.clear(); res.glyphs.clear();
    const f26 scale = context_scale(font_size);
    f26 X = (f26)((uint32_t)px << 6), Y = (f26)((uint32_t)py << 6);   // freetype.Pt
    uint32_t prev = 0;
    bool has_prev = false;
    // (c *Context) glyph keeps a cache of nGlyphs * nXFractions * nYFractions = 256 * 4 * 1 entries, and addTextWatermark makes a
    // fresh Context per call (watermark.go:98), so the cache lives exactly as long as this DrawString.  Slot t = ((fx / 16) * 1 +
    // fy / 64) * 256 + glyph % 256; a hit needs `e.valid && e.glyph == glyph` only -- NOT an equal fx: a glyph that comes again with
    // its fx in the same quarter-pixel bucket reuses the mask (and offset, and advance) rasterised at the FIRST fx.  The default
    // text "© ImageProcessor" repeats o, r, s and e, so this decides antialiasing pixels of the reference's watermark.
    struct CacheEntry { bool valid = false; uint32_t glyph = 0; std::shared_ptr<ipx_font::Mask> mask; };
    std::vector<CacheEntry> cache(256 * 4);
    try {
        for (uint32_t r : runes(text)) {
            const uint32_t index = glyph_index(*f, r);
            if (has_prev) X += kern(*f, scale, prev, index);
            const int ix = (int)(X >> 6), iy = (int)(Y >> 6);
            const f26 fx = X & 0x3f, fy = Y & 0x3f;
            CacheEntry &e = cache[(size_t)(((int)fx / 16) * 1 + (int)fy / 64) * 256 + index % 256];
            if (!(e.valid && e.glyph == index)) {
                e.mask = cached_mask(*f, scale, index, fx, fy);     // a ParseError leaves the slot as it was, like Go's early return
                e.valid = true; e.glyph = index;
            }
            auto m = e.mask;
            X += m->advance;
            // glyphRect = mask.Bounds().Add(offset + (ix, iy)); dr = clip ∩ glyphRect
            const int gx0 = m->offx + ix, gy0 = m->offy + iy, gx1 = gx0 + m->w, gy1 = gy0 + m->h;
            int dx0 = gx0 > 0 ? gx0 : 0, dy0 = gy0 > 0 ? gy0 : 0;
            int dx1 = gx1 < clip_w ? gx1 : clip_w, dy1 = gy1 < clip_h ? gy1 : clip_h;
            const bool glyph_empty = gx0 >= gx1 || gy0 >= gy1;
            if (!glyph_empty && dx0 < dx1 && dy0 < dy1) {
                ipx_glyph g;
                g.mask = m->pix.data();
                g.mw = m->w; g.mh = m->h; g.mstride = m->w;
                g.dr = ipx_rect{dx0, dy0, dx1, dy1};
                g.mpx = 0;                 // DrawString passes image.Point{0, dr.Min.Y - glyphRect.Min.Y}: x is NOT adjusted
                g.mpy = dy0 - gy0;
                res.glyphs.push_back(g);
                res.keep.push_back(m);
            }
            prev = index; has_prev = true;
        }
    } catch (const ParseError &e) {
        res.keep.clear(); res.glyphs.clear();
        ipx::set_error("%s", e.text.c_str());
        return IPX_ERR_INVALID;
    }
    *out = res.glyphs.data();
    *n = (int)res.glyphs.size();
    if (end_x26_6) *end_x26_6 = X;
    return IPX_OK;
}

void ipx_font_release_thread(void)
{
    t_result.keep.clear();
    t_result.glyphs.clear();
}

// ---- the ipx_text_rasterizer seam of the operator entry points -----------------------------------
static int font_measure(void *user, const char *text, double font_size, int *width_px)
{
    return ipx_font_text_width((const ipx_font *)user, text, font_size, nullptr, width_px);
}
static int font_glyphs(void *user, const char *text, double font_size, int px, int py, int w, int h,
                       const ipx_glyph **out, int *n)
{
    return ipx_font_draw_string((ipx_font *)user, text, font_size, px, py, w, h, out, n, nullptr);
}
static void font_release(void *) { ipx_font_release_thread(); }

int ipx_font_rasterizer(ipx_font *f, ipx_text_rasterizer *out)
{
    if (!f || !out) { ipx::set_error("font not loaded"); return IPX_ERR_INVALID; }
    out->user = f;
    out->measure = font_measure;
    out->glyphs = font_glyphs;
    out->release = font_release;
    return IPX_OK;
}

}  // extern "C"

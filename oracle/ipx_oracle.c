/*
 * ipx_oracle.c -- CPU restatement of the reference's pixel hot path (plain C, scalar).
 *
 * TEST INFRASTRUCTURE ONLY (see ipx_oracle.h).  PARITY UNPINNED: the reference holds no
 * golden vectors for this path and cannot be built here (Go; no toolchain, no module cache).
 *
 * Build with -ffp-contract=off: the reference is built for GOAMD64=v1 (dockerfile:12-13),
 * where the Go compiler never fuses x*y+z, so every float64 product below is rounded
 * before it is added.
 *
 * Reference call sites (under internal/usecase/processor/operations/):
 *   resize.go:121-125      resizeImage      -> x/image/draw BiLinear.Scale(..., Over, nil)
 *   thumbnail.go:114-132   cropAndResize    -> BiLinear.Scale (equal sizes: one tap of weight 1 per axis) + resizeImage
 *   watermark.go:90-92     draw.Draw(result, bounds, img, ZP, draw.Src)
 *   watermark.go:151       freetype DrawString -> draw.DrawMask(dst, dr, Uniform, ZP, Alpha, mp, Over)
 * Upstream routines restated: x/image@v0.33.0 draw/scale.go -- BiLinear = &Kernel{1, tent}, Kernel.Scale ->
 * newDistrib, kernelScaler.Scale, opaque -- and draw/impl.go scaleX_{RGBA,NRGBA,Gray,YCbCr4xx,Image},
 * scaleY_RGBA_{Src,Over}, ftou; Go 1.24 image/draw/draw.go clip, DrawMask, drawCopyOver, drawCopySrc,
 * drawGlyphOver; image/image.go (*RGBA).Opaque.
 *
 * Rounds 1-2 restated ablInterpolator (x/image's ApproxBiLinear) here by mistake: BiLinear is the tent KERNEL,
 * a two-pass float64 scaler whose tap count grows with the downscale ratio.  thumbnail.go:129's equal-size
 * Scale runs through the same kernel scaler (a single tap of weight 1 per axis): only nnInterpolator /
 * ablInterpolator.Scale simplify equal sizes to Copy.
 */
#include "ipx_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- image.Rectangle helpers (image/geom.go) ---------------------------------------- */

static int rect_empty(ipxo_rect r) { return r.x0 >= r.x1 || r.y0 >= r.y1; }

static ipxo_rect rect_intersect(ipxo_rect r, ipxo_rect s)
{
    if (r.x0 < s.x0) r.x0 = s.x0;
    if (r.y0 < s.y0) r.y0 = s.y0;
    if (r.x1 > s.x1) r.x1 = s.x1;
    if (r.y1 > s.y1) r.y1 = s.y1;
    if (rect_empty(r)) { ipxo_rect z = {0, 0, 0, 0}; return z; }
    return r;
}

static ipxo_rect rect_add(ipxo_rect r, int dx, int dy)
{
    r.x0 += dx; r.x1 += dx; r.y0 += dy; r.y1 += dy;
    return r;
}

/* ---- geometry ------------------------------------------------------------------------ */

/* resize.go:61-75.  Go's int(float64) truncates toward zero. */
void ipxo_resize_dims(int ow, int oh, int w, int h, int keep_aspect, int *nw, int *nh)
{
    if (keep_aspect) {
        double width_ratio = (double)w / (double)ow;
        double height_ratio = (double)h / (double)oh;
        double ratio = width_ratio < height_ratio ? width_ratio : height_ratio; /* math.Min */
        *nw = (int)((double)ow * ratio);
        *nh = (int)((double)oh * ratio);
    } else {
        *nw = w;
        *nh = h;
    }
}

/* thumbnail.go:48-65 and :114-127 */
void ipxo_thumb_geometry(int ow, int oh, int size, int crop_to_fit, ipxo_rect *crop, int *nw,
                         int *nh)
{
    if (crop_to_fit) {
        int cx, cy, cs;
        if (ow > oh) { cs = oh; cx = (ow - oh) / 2; cy = 0; }
        else         { cs = ow; cx = 0; cy = (oh - ow) / 2; }
        crop->x0 = cx; crop->y0 = cy; crop->x1 = cx + cs; crop->y1 = cy + cs;
        *nw = size; *nh = size;
    } else {
        crop->x0 = 0; crop->y0 = 0; crop->x1 = ow; crop->y1 = oh;
        if (ow > oh) {
            *nh = size;
            *nw = (int)((double)ow * (double)size / (double)oh);
        } else {
            *nw = size;
            *nh = (int)((double)oh * (double)size / (double)ow);
        }
    }
}

/* watermark.go:116-118: fixed.Int26_6(fontSize*64*1.2).Ceil() */
int ipxo_text_height_px(double font_size)
{
    int32_t h = (int32_t)(font_size * 64 * 1.2);
    return (int)((h + 0x3f) >> 6);
}

/* watermark.go:121-148; freetype.Pt(x, y) keeps whole pixels, returned here as such */
void ipxo_watermark_anchor(const char *position, int w, int h, int width_px, int height_px,
                           int *px, int *py)
{
    const int margin = 20;
    if (!strcmp(position, "top-left"))            { *px = margin;                 *py = margin + height_px; }
    else if (!strcmp(position, "top-right"))      { *px = w - width_px - margin;  *py = margin + height_px; }
    else if (!strcmp(position, "top-center"))     { *px = (w - width_px) / 2;     *py = margin + height_px; }
    else if (!strcmp(position, "bottom-left"))    { *px = margin;                 *py = h - margin; }
    else if (!strcmp(position, "bottom-right"))   { *px = w - width_px - margin;  *py = h - margin; }
    else if (!strcmp(position, "bottom-center"))  { *px = (w - width_px) / 2;     *py = h - margin; }
    else if (!strcmp(position, "center"))         { *px = (w - width_px) / 2;     *py = (h + height_px) / 2; }
    else                                          { *px = w - width_px - margin;  *py = h - margin; }
}

/* strconv.Atoi: optional sign, decimal digits only, at least one digit */
static int go_atoi(const char *s, size_t n, long *out)
{
    size_t i = 0;
    int neg = 0;
    long v = 0;
    if (n == 0) return -1;
    if (s[0] == '+' || s[0] == '-') { neg = s[0] == '-'; i = 1; if (n == 1) return -1; }
    for (; i < n; i++) {
        if (s[i] < '0' || s[i] > '9') return -1;
        if (v > 900000000000000000L) return -1; /* Atoi reports a range error near 2^63 */
        v = v * 10 + (s[i] - '0');
    }
    *out = neg ? -v : v;
    return 0;
}

static int clampi(long v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : (int)v); }

/* watermark.go:159-186 with the fallback of :93-97 folded in */
int ipxo_parse_color(const char *s, double opacity, uint8_t rgba[4])
{
    char buf[256];
    const char *part[8];
    size_t plen[8];
    size_t n = 0, i;
    int nparts = 0;
    uint8_t oa = (uint8_t)(int32_t)(255 * opacity);
    long r, g, b, a;

    for (i = 0; s[i] && n + 1 < sizeof buf; i++)
        if (s[i] != ' ') buf[n++] = s[i]; /* strings.ReplaceAll(colorStr, " ", "") */
    buf[n] = 0;
    part[0] = buf; nparts = 1;
    for (i = 0; i < n; i++)
        if (buf[i] == ',') {
            plen[nparts - 1] = (size_t)(&buf[i] - part[nparts - 1]);
            if (nparts == 8) goto bad;
            part[nparts++] = &buf[i + 1];
        }
    plen[nparts - 1] = (size_t)(&buf[n] - part[nparts - 1]);
    if (nparts != 3 && nparts != 4) goto bad;
    if (go_atoi(part[0], plen[0], &r) || go_atoi(part[1], plen[1], &g) ||
        go_atoi(part[2], plen[2], &b))
        goto bad;
    rgba[0] = (uint8_t)clampi(r, 0, 255);
    rgba[1] = (uint8_t)clampi(g, 0, 255);
    rgba[2] = (uint8_t)clampi(b, 0, 255);
    if (nparts == 4 && go_atoi(part[3], plen[3], &a) == 0) rgba[3] = (uint8_t)clampi(a, 0, 255);
    else rgba[3] = oa;
    return 0;
bad:
    rgba[0] = rgba[1] = rgba[2] = 0; /* watermark.go:96: black with the opacity alpha */
    rgba[3] = oa;
    return 1;
}

/* ---- image/draw: DrawMask with a nil mask, *image.RGBA <- *image.RGBA ----------------- */

void ipxo_draw_rgba8(uint8_t *dst, int dw, int dh, int dstride, ipxo_rect r,
                     const uint8_t *src, int sw, int sh, int sstride, int spx, int spy, int op)
{
    const uint32_t m = 0xffff;
    ipxo_rect db = {0, 0, dw, dh}, sb = {0, 0, sw, sh};
    int ox = r.x0, oy = r.y0, x, y;
    /* draw.clip */
    r = rect_intersect(r, db);
    r = rect_intersect(r, rect_add(sb, ox - spx, oy - spy));
    if (rect_empty(r)) return;
    spx += r.x0 - ox;
    spy += r.y0 - oy;
    for (y = 0; y < r.y1 - r.y0; y++) {
        uint8_t *d = dst + (size_t)(r.y0 + y) * dstride + (size_t)r.x0 * 4;
        const uint8_t *s = src + (size_t)(spy + y) * sstride + (size_t)spx * 4;
        if (op == IPXO_OP_SRC) { /* drawCopySrc */
            memmove(d, s, (size_t)(r.x1 - r.x0) * 4);
            continue;
        }
        for (x = 0; x < r.x1 - r.x0; x++, d += 4, s += 4) { /* drawCopyOver */
            uint32_t sr_ = (uint32_t)s[0] * 0x101;
            uint32_t sg_ = (uint32_t)s[1] * 0x101;
            uint32_t sb_ = (uint32_t)s[2] * 0x101;
            uint32_t sa_ = (uint32_t)s[3] * 0x101;
            uint32_t a = (m - sa_) * 0x101;
            d[0] = (uint8_t)(((uint32_t)d[0] * a / m + sr_) >> 8);
            d[1] = (uint8_t)(((uint32_t)d[1] * a / m + sg_) >> 8);
            d[2] = (uint8_t)(((uint32_t)d[2] * a / m + sb_) >> 8);
            d[3] = (uint8_t)(((uint32_t)d[3] * a / m + sa_) >> 8);
        }
    }
}

/* ---- x/image/draw: BiLinear.Scale = (&Kernel{1, tent}).Scale ---------------------------------------
 *
 * draw/scale.go:
 *   BiLinear = &Kernel{1, func(t float64) float64 { return 1 - t }}
 *   (q *Kernel).Scale(dst, dr, src, sr, op, opts) = q.newScaler(dr.Dx(), dr.Dy(), sr.Dx(), sr.Dy(), false).Scale(...)
 *   newDistrib(q, dw, sw): per destination index the contributing source indices and their weights
 *   kernelScaler.Scale: adr / opaque() / Uniform shortcut, scaleX_<src type> into tmp [][4]float64 (dw x sh),
 *     then scaleY_RGBA_{Src,Over} down the columns of tmp.
 */

typedef void (*tap_fn)(const void *src, int x, int y, uint32_t out[4]); /* 16-bit premultiplied RGBA */

typedef struct { int32_t i, j; double inv_total, inv_total_ffff; } ks_source;
typedef struct { int32_t coord; double weight; } ks_contrib;
typedef struct { ks_source *sources; ks_contrib *contribs; } ks_distrib;

/* newDistrib with q.Support = 1 and q.At(t) = 1 - t.  0 ok, -1 out of memory. */
static int ks_new_distrib(ks_distrib *d, int32_t dw, int32_t sw)
{
    const double support = 1.0;
    double scale = (double)sw / (double)dw;
    double half_width = support, kernel_arg_scale = 1.0;
    int32_t x, n = 0;
    size_t nc = 0;
    /* When shrinking, broaden the effective kernel support so that every source pixel is visited. */
    if (scale > 1) {
        half_width *= scale;
        kernel_arg_scale = 1 / scale;
    }
    d->sources = (ks_source *)malloc(sizeof(ks_source) * (size_t)(dw > 0 ? dw : 1));
    if (!d->sources) return -1;
    /* first pass: i, j = range of source indices; inv_total temporarily holds the centre */
    for (x = 0; x < dw; x++) {
        double center = ((double)x + 0.5) * scale - 0.5;
        int32_t i = (int32_t)floor(center - half_width);
        int32_t j;
        if (i < 0) i = 0;
        j = (int32_t)ceil(center + half_width);
        if (j > sw) {
            j = sw;
            if (j < i) j = i;
        }
        d->sources[x].i = i; d->sources[x].j = j; d->sources[x].inv_total = center;
        n += j - i;
    }
    d->contribs = (ks_contrib *)malloc(sizeof(ks_contrib) * (size_t)(n > 0 ? n : 1));
    if (!d->contribs) { free(d->sources); return -1; }
    for (x = 0; x < dw; x++) {
        ks_source b = d->sources[x];
        double total = 0.0;
        int32_t l = (int32_t)nc, coord;
        for (coord = b.i; coord < b.j; coord++) {
            double t = (b.inv_total - (double)coord) * kernel_arg_scale, weight;
            if (t < 0) t = -t;               /* scale.go's own abs() */
            if (t >= support) continue;
            weight = 1 - t;                  /* BiLinear's At */
            if (weight == 0) continue;
            total += weight;
            d->contribs[nc].coord = coord; d->contribs[nc].weight = weight; nc++;
        }
        total = 1 / total;
        d->sources[x].i = l; d->sources[x].j = (int32_t)nc;
        d->sources[x].inv_total = total;
        d->sources[x].inv_total_ffff = total / 0xffff;
    }
    return 0;
}

static void ks_free(ks_distrib *d) { free(d->sources); free(d->contribs); }

static uint32_t ks_ftou(double f) /* impl.go ftou */
{
    int32_t i = (int32_t)(0xffff * f + 0.5);
    if (i > 0xffff) return 0xffff;
    if (i > 0) return (uint32_t)i;
    return 0;
}

/* kernelScaler.Scale after its adr / opaque() preamble.  adr is relative to dr.Min.  alpha_one: the source type's scaleX writes
 * a literal 1 into tmp's alpha (scaleX_Gray, scaleX_YCbCr4xx) instead of the weighted sum of 0xffff taps.
 * 0 ok, -3 out of memory. */
static int kernel_scale(uint8_t *dst, int dstride, ipxo_rect dr, ipxo_rect adr, const void *src, tap_fn tap, ipxo_rect sr, int op,
                        int alpha_one)
{
    const int32_t dw = dr.x1 - dr.x0, dh = dr.y1 - dr.y0, sw = sr.x1 - sr.x0, sh = sr.y1 - sr.y0;
    ks_distrib hz, vt;
    double (*tmp)[4];
    int32_t x, y, dx, k;
    size_t t = 0;
    if (ks_new_distrib(&hz, dw, sw)) return -3;
    if (ks_new_distrib(&vt, dh, sh)) { ks_free(&hz); return -3; }
    tmp = (double (*)[4])malloc(sizeof(double[4]) * (size_t)dw * (size_t)sh);   /* makeTmpBuf: z.dw * z.sh */
    if (!tmp) { ks_free(&hz); ks_free(&vt); return -3; }

    /* scaleX_*: distributes the source image's columns over the temporary image */
    for (y = 0; y < sh; y++) {
        for (x = 0; x < dw; x++) {
            const ks_source *s = &hz.sources[x];
            double pr = 0, pg = 0, pb = 0, pa = 0;
            for (k = s->i; k < s->j; k++) {
                const ks_contrib *c = &hz.contribs[k];
                uint32_t p[4];
                tap(src, sr.x0 + c->coord, sr.y0 + y, p);
                pr += (double)p[0] * c->weight;
                pg += (double)p[1] * c->weight;
                pb += (double)p[2] * c->weight;
                pa += (double)p[3] * c->weight;
            }
            tmp[t][0] = pr * s->inv_total_ffff;
            tmp[t][1] = pg * s->inv_total_ffff;
            tmp[t][2] = pb * s->inv_total_ffff;
            tmp[t][3] = alpha_one ? 1.0 : pa * s->inv_total_ffff;
            t++;
        }
    }

    /* scaleY_RGBA_{Src,Over}: distributes the temporary image's rows over the destination image */
    for (dx = adr.x0; dx < adr.x1; dx++) {
        uint8_t *d = dst + (size_t)(dr.y0 + adr.y0) * dstride + (size_t)(dr.x0 + dx) * 4;
        for (y = adr.y0; y < adr.y1; y++, d += dstride) {
            const ks_source *s = &vt.sources[y];
            double pr = 0, pg = 0, pb = 0, pa = 0;
            for (k = s->i; k < s->j; k++) {
                const ks_contrib *c = &vt.contribs[k];
                const double *p = tmp[(size_t)c->coord * dw + dx];
                pr += p[0] * c->weight;
                pg += p[1] * c->weight;
                pb += p[2] * c->weight;
                pa += p[3] * c->weight;
            }
            if (pr > pa) pr = pa;
            if (pg > pa) pg = pa;
            if (pb > pa) pb = pa;
            if (op == IPXO_OP_SRC) {
                d[0] = (uint8_t)(ks_ftou(pr * s->inv_total) >> 8);
                d[1] = (uint8_t)(ks_ftou(pg * s->inv_total) >> 8);
                d[2] = (uint8_t)(ks_ftou(pb * s->inv_total) >> 8);
                d[3] = (uint8_t)(ks_ftou(pa * s->inv_total) >> 8);
            } else {
                uint32_t pr0 = ks_ftou(pr * s->inv_total), pg0 = ks_ftou(pg * s->inv_total);
                uint32_t pb0 = ks_ftou(pb * s->inv_total), pa0 = ks_ftou(pa * s->inv_total);
                uint32_t pa1 = (0xffff - pa0) * 0x101;
                d[0] = (uint8_t)(((uint32_t)d[0] * pa1 / 0xffff + pr0) >> 8);
                d[1] = (uint8_t)(((uint32_t)d[1] * pa1 / 0xffff + pg0) >> 8);
                d[2] = (uint8_t)(((uint32_t)d[2] * pa1 / 0xffff + pb0) >> 8);
                d[3] = (uint8_t)(((uint32_t)d[3] * pa1 / 0xffff + pa0) >> 8);
            }
        }
    }
    free(tmp);
    ks_free(&hz);
    ks_free(&vt);
    return 0;
}

/* the preamble of kernelScaler.Scale shared by every source type: adr, then the caller's opaque() test */
static int ks_adr(int dw, int dh, ipxo_rect dr, ipxo_rect sr, int sw, int sh, ipxo_rect *adr)
{
    ipxo_rect db = {0, 0, dw, dh};
    *adr = rect_intersect(db, dr);
    if (rect_empty(*adr) || rect_empty(sr)) return 1;              /* nothing to do */
    *adr = rect_add(*adr, -dr.x0, -dr.y0);
    if (sr.x0 < 0 || sr.y0 < 0 || sr.x1 > sw || sr.y1 > sh) return -1; /* scaleX_Image on a partly outside sr: not restated */
    return 0;
}

/* image.(*RGBA).Opaque over the whole source image, as draw/scale.go opaque() asks */
static int rgba_opaque(const uint8_t *src, int sw, int sh, int sstride)
{
    int x, y;
    for (y = 0; y < sh; y++)
        for (x = 0; x < sw; x++)
            if (src[(size_t)y * sstride + (size_t)x * 4 + 3] != 0xff) return 0;
    return 1;
}

typedef struct { const uint8_t *pix; int stride; } rgba_src;

static void tap_rgba(const void *s, int x, int y, uint32_t out[4]) /* scaleX_RGBA: Pix * 0x101 */
{
    const rgba_src *n = (const rgba_src *)s;
    const uint8_t *p = n->pix + (size_t)y * n->stride + (size_t)x * 4;
    out[0] = (uint32_t)p[0] * 0x101; out[1] = (uint32_t)p[1] * 0x101;
    out[2] = (uint32_t)p[2] * 0x101; out[3] = (uint32_t)p[3] * 0x101;
}

int ipxo_scale_bilinear_rgba8(uint8_t *dst, int dw, int dh, int dstride, ipxo_rect dr,
                              const uint8_t *src, int sw, int sh, int sstride, ipxo_rect sr,
                              int op)
{
    ipxo_rect adr;
    rgba_src n;
    int rc = ks_adr(dw, dh, dr, sr, sw, sh, &adr);
    if (rc) return rc < 0 ? -1 : 0;
    if (op == IPXO_OP_OVER && rgba_opaque(src, sw, sh, sstride)) op = IPXO_OP_SRC;
    n.pix = src; n.stride = sstride;
    return kernel_scale(dst, dstride, dr, adr, &n, tap_rgba, sr, op, 0);
}

/* ---- image/draw: drawGlyphOver, one call per glyph in string order --------------------- */

void ipxo_composite_glyphs_rgba8(uint8_t *dst, int dw, int dh, int dstride,
                                 const ipxo_glyph *glyphs, int n, const uint8_t col[4])
{
    const uint32_t m = 0xffff;
    /* color.RGBA.RGBA(): each 8-bit field widened with v |= v<<8; no premultiplication */
    const uint32_t sr = (uint32_t)col[0] * 0x101, sg = (uint32_t)col[1] * 0x101,
                   sb = (uint32_t)col[2] * 0x101, sa = (uint32_t)col[3] * 0x101;
    ipxo_rect db = {0, 0, dw, dh};
    int g, x, y;
    for (g = 0; g < n; g++) {
        const ipxo_glyph *gl = &glyphs[g];
        ipxo_rect r = gl->dr, mb = {0, 0, gl->mw, gl->mh};
        int ox = r.x0, oy = r.y0, mpx = gl->mpx, mpy = gl->mpy;
        /* draw.clip; the Uniform source is unbounded */
        r = rect_intersect(r, db);
        r = rect_intersect(r, rect_add(mb, ox - mpx, oy - mpy));
        if (rect_empty(r)) continue;
        mpx += r.x0 - ox;
        mpy += r.y0 - oy;
        for (y = 0; y < r.y1 - r.y0; y++) {
            uint8_t *d = dst + (size_t)(r.y0 + y) * dstride + (size_t)r.x0 * 4;
            const uint8_t *mk = gl->mask + (size_t)(mpy + y) * gl->mstride + mpx;
            for (x = 0; x < r.x1 - r.x0; x++, d += 4) {
                uint32_t ma = mk[x];
                uint32_t a;
                if (ma == 0) continue;
                ma |= ma << 8;
                a = (m - (sa * ma / m)) * 0x101;
                d[0] = (uint8_t)(((uint32_t)d[0] * a + sr * ma) / m >> 8);
                d[1] = (uint8_t)(((uint32_t)d[1] * a + sg * ma) / m >> 8);
                d[2] = (uint8_t)(((uint32_t)d[2] * a + sb * ma) / m >> 8);
                d[3] = (uint8_t)(((uint32_t)d[3] * a + sa * ma) / m >> 8);
            }
        }
    }
}

/* ---- image_processor.go:64-65,104-117: each operator applied to the original frame ----- */

int ipxo_process_rgba8(const ipxo_pipeline *p, const uint8_t *src, int sw, int sh, int sstride,
                       uint8_t *resize_out, uint8_t *thumb_out, uint8_t *wm_out)
{
    ipxo_rect full = {0, 0, sw, sh};
    if (resize_out) { /* resize.go:61-75,121-125 */
        int nw, nh;
        ipxo_rect dr;
        ipxo_resize_dims(sw, sh, p->resize_w, p->resize_h, p->keep_aspect, &nw, &nh);
        if (nw < 0 || nh < 0) return -2; /* image.NewRGBA would panic */
        memset(resize_out, 0, (size_t)nw * nh * 4);
        dr.x0 = 0; dr.y0 = 0; dr.x1 = nw; dr.y1 = nh;
        if (ipxo_scale_bilinear_rgba8(resize_out, nw, nh, nw * 4, dr, src, sw, sh, sstride, full,
                                      IPXO_OP_OVER))
            return -1;
    }
    if (thumb_out) { /* thumbnail.go:48-65,114-132 */
        int nw, nh;
        ipxo_rect crop, dr;
        ipxo_thumb_geometry(sw, sh, p->thumb_size, p->crop_to_fit, &crop, &nw, &nh);
        memset(thumb_out, 0, (size_t)nw * nh * 4);
        dr.x0 = 0; dr.y0 = 0; dr.x1 = nw; dr.y1 = nh;
        if (p->crop_to_fit) {
            int cs = crop.x1 - crop.x0;
            ipxo_rect cr = {0, 0, cs, cs};
            uint8_t *cropped = (uint8_t *)calloc((size_t)cs * cs, 4);
            if (!cropped) return -3;
            ipxo_scale_bilinear_rgba8(cropped, cs, cs, cs * 4, cr, src, sw, sh, sstride, crop,
                                      IPXO_OP_OVER);
            ipxo_scale_bilinear_rgba8(thumb_out, nw, nh, nw * 4, dr, cropped, cs, cs, cs * 4, cr,
                                      IPXO_OP_OVER);
            free(cropped);
        } else {
            ipxo_scale_bilinear_rgba8(thumb_out, nw, nh, nw * 4, dr, src, sw, sh, sstride, full,
                                      IPXO_OP_OVER);
        }
    }
    if (wm_out) { /* watermark.go:90-92,151 */
        memset(wm_out, 0, (size_t)sw * sh * 4);
        ipxo_draw_rgba8(wm_out, sw, sh, sw * 4, full, src, sw, sh, sstride, 0, 0, IPXO_OP_SRC);
        ipxo_composite_glyphs_rgba8(wm_out, sw, sh, sw * 4, p->glyphs, p->n_glyphs, p->col);
    }
    return 0;
}

/* ======================================================================================================
 * Source-type variants.  Same kernel scaler, different scaleX_<type> tap fetch.
 * Upstream routines restated: x/image@v0.33.0 draw/impl.go scaleX_NRGBA, scaleX_YCbCr{444,422,420,440},
 * scaleX_Gray, scaleX_Image; Go 1.24 image/draw drawNRGBAOver / drawNRGBASrc;
 * image/internal/imageutil DrawYCbCr; image/color YCbCr.RGBA / YCbCrToRGB.
 * ====================================================================================================== */

/* ---- *image.NRGBA ---------------------------------------------------------------------------------- */

typedef struct { const uint8_t *pix; int stride; } nrgba_src;

static void tap_nrgba(const void *s, int x, int y, uint32_t out[4])
{
    const nrgba_src *n = (const nrgba_src *)s;
    const uint8_t *p = n->pix + (size_t)y * n->stride + (size_t)x * 4;
    uint32_t a = (uint32_t)p[3] * 0x101;
    out[0] = (uint32_t)p[0] * a / 0xff;
    out[1] = (uint32_t)p[1] * a / 0xff;
    out[2] = (uint32_t)p[2] * a / 0xff;
    out[3] = a;
}

void ipxo_draw_nrgba8(uint8_t *dst, int dw, int dh, int dstride, ipxo_rect r,
                      const uint8_t *src, int sw, int sh, int sstride, int spx, int spy, int op)
{
    const uint32_t m = 0xffff;
    ipxo_rect db = {0, 0, dw, dh}, sb = {0, 0, sw, sh};
    int ox = r.x0, oy = r.y0, x, y;
    r = rect_intersect(r, db);
    r = rect_intersect(r, rect_add(sb, ox - spx, oy - spy));
    if (rect_empty(r)) return;
    spx += r.x0 - ox;
    spy += r.y0 - oy;
    for (y = 0; y < r.y1 - r.y0; y++) {
        uint8_t *d = dst + (size_t)(r.y0 + y) * dstride + (size_t)r.x0 * 4;
        const uint8_t *s = src + (size_t)(spy + y) * sstride + (size_t)spx * 4;
        for (x = 0; x < r.x1 - r.x0; x++, d += 4, s += 4) {
            uint32_t sa = (uint32_t)s[3] * 0x101;
            uint32_t sr_ = (uint32_t)s[0] * sa / 0xff, sg_ = (uint32_t)s[1] * sa / 0xff, sb_ = (uint32_t)s[2] * sa / 0xff;
            if (op == IPXO_OP_SRC) { /* drawNRGBASrc */
                d[0] = (uint8_t)(sr_ >> 8); d[1] = (uint8_t)(sg_ >> 8); d[2] = (uint8_t)(sb_ >> 8); d[3] = (uint8_t)(sa >> 8);
            } else {                 /* drawNRGBAOver */
                uint32_t a = (m - sa) * 0x101;
                d[0] = (uint8_t)(((uint32_t)d[0] * a / m + sr_) >> 8);
                d[1] = (uint8_t)(((uint32_t)d[1] * a / m + sg_) >> 8);
                d[2] = (uint8_t)(((uint32_t)d[2] * a / m + sb_) >> 8);
                d[3] = (uint8_t)(((uint32_t)d[3] * a / m + sa) >> 8);
            }
        }
    }
}

int ipxo_scale_bilinear_nrgba8(uint8_t *dst, int dw, int dh, int dstride, ipxo_rect dr,
                               const uint8_t *src, int sw, int sh, int sstride, ipxo_rect sr, int op)
{
    ipxo_rect adr;
    nrgba_src n;
    int rc = ks_adr(dw, dh, dr, sr, sw, sh, &adr);
    if (rc) return rc < 0 ? -1 : 0;
    if (op == IPXO_OP_OVER && rgba_opaque(src, sw, sh, sstride)) op = IPXO_OP_SRC; /* (*NRGBA).Opaque: same scan */
    n.pix = src; n.stride = sstride;
    return kernel_scale(dst, dstride, dr, adr, &n, tap_nrgba, sr, op, 0);          /* scaleX_NRGBA */
}

/* ---- *image.YCbCr ---------------------------------------------------------------------------------- */

static size_t ycbcr_coff(const ipxo_ycbcr *s, int x, int y) /* image.(*YCbCr).COffset with Rect.Min = 0 */
{
    switch (s->ratio) {
    case 1: return (size_t)y * s->cstride + (size_t)(x / 2);          /* 4:2:2 */
    case 2: return (size_t)(y / 2) * s->cstride + (size_t)(x / 2);    /* 4:2:0 */
    case 3: return (size_t)(y / 2) * s->cstride + (size_t)x;          /* 4:4:0 */
    default: return (size_t)y * s->cstride + (size_t)x;               /* 4:4:4 */
    }
}

static void tap_ycbcr(const void *sv, int x, int y, uint32_t out[4]) /* color.YCbCr.RGBA, inlined upstream */
{
    const ipxo_ycbcr *s = (const ipxo_ycbcr *)sv;
    size_t ci = ycbcr_coff(s, x, y);
    int yy1 = (int)s->y[(size_t)y * s->ystride + x] * 0x10101;
    int cb1 = (int)s->cb[ci] - 128, cr1 = (int)s->cr[ci] - 128;
    int r = (yy1 + 91881 * cr1) >> 8;
    int g = (yy1 - 22554 * cb1 - 46802 * cr1) >> 8;
    int b = (yy1 + 116130 * cb1) >> 8;
    out[0] = (uint32_t)(r < 0 ? 0 : r > 0xffff ? 0xffff : r);
    out[1] = (uint32_t)(g < 0 ? 0 : g > 0xffff ? 0xffff : g);
    out[2] = (uint32_t)(b < 0 ? 0 : b > 0xffff ? 0xffff : b);
    out[3] = 0xffff;
}

void ipxo_draw_ycbcr(uint8_t *dst, int dw, int dh, int dstride, ipxo_rect r, const ipxo_ycbcr *src,
                     int spx, int spy)
{
    ipxo_rect db = {0, 0, dw, dh}, sb = {0, 0, src->w, src->h};
    int ox = r.x0, oy = r.y0, x, y;
    r = rect_intersect(r, db);
    r = rect_intersect(r, rect_add(sb, ox - spx, oy - spy));
    if (rect_empty(r)) return;
    spx += r.x0 - ox;
    spy += r.y0 - oy;
    for (y = 0; y < r.y1 - r.y0; y++) {
        uint8_t *d = dst + (size_t)(r.y0 + y) * dstride + (size_t)r.x0 * 4;
        for (x = 0; x < r.x1 - r.x0; x++, d += 4) { /* color.YCbCrToRGB */
            size_t ci = ycbcr_coff(src, spx + x, spy + y);
            int32_t yy1 = (int32_t)src->y[(size_t)(spy + y) * src->ystride + spx + x] * 0x10101;
            int32_t cb1 = (int32_t)src->cb[ci] - 128, cr1 = (int32_t)src->cr[ci] - 128;
            int32_t rr = yy1 + 91881 * cr1, gg = yy1 - 22554 * cb1 - 46802 * cr1, bb = yy1 + 116130 * cb1;
            rr = ((uint32_t)rr & 0xff000000u) == 0 ? rr >> 16 : ~(rr >> 31);
            gg = ((uint32_t)gg & 0xff000000u) == 0 ? gg >> 16 : ~(gg >> 31);
            bb = ((uint32_t)bb & 0xff000000u) == 0 ? bb >> 16 : ~(bb >> 31);
            d[0] = (uint8_t)rr; d[1] = (uint8_t)gg; d[2] = (uint8_t)bb; d[3] = 255;
        }
    }
}

int ipxo_scale_bilinear_ycbcr(uint8_t *dst, int dw, int dh, int dstride, ipxo_rect dr,
                              const ipxo_ycbcr *src, ipxo_rect sr)
{
    ipxo_rect adr;
    int rc = ks_adr(dw, dh, dr, sr, src->w, src->h, &adr);
    if (rc) return rc < 0 ? -1 : 0;
    /* a YCbCr image is opaque: Over becomes Src.  scaleX_YCbCr4xx writes tmp alpha = 1 */
    return kernel_scale(dst, dstride, dr, adr, src, tap_ycbcr, sr, IPXO_OP_SRC, 1);
}

/* ---- *image.Paletted (GIF uploads, palette PNGs) -----------------------------------------------------
 * No specialised routine exists upstream for this source type, so the GENERIC ones run:
 *   x/image@v0.33.0 draw/impl.go scaleX_Image (scaleX_RGBA64Image): every tap is src.At(x, y).RGBA(), i.e. the
 *     palette entry's 16-bit premultiplied colour, then the same float64 lerps and the same stores;
 *   Go 1.24 image/draw drawRGBA (the fallback of DrawMask for dst *image.RGBA): with a nil mask ma = m and
 *     Src stores uint8(sr*ma/m >> 8), Over stores uint8((dr*a + sr*ma)/m >> 8) with a = (m - sa*ma/m)*0x101.
 * pal16[i] = Palette[i].RGBA() is computed by the caller from the entry's concrete colour type (color.RGBA
 * from the GIF decoder and from PNGs without tRNS: c*0x101; color.NRGBA from PNGs with tRNS:
 * (c*0x101)*a/0xff, alpha a*0x101) -- oracle/__init__.py palette16. */
typedef struct { const uint8_t *pix; int stride; const uint16_t (*pal)[4]; } pal_src;

static void tap_paletted(const void *s, int x, int y, uint32_t out[4])
{
    const pal_src *p = (const pal_src *)s;
    const uint16_t *c = p->pal[p->pix[(size_t)y * p->stride + x]];
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

void ipxo_draw_paletted(uint8_t *dst, int dw, int dh, int dstride, ipxo_rect r, const uint8_t *src, int sw, int sh,
                        int sstride, const uint16_t pal16[256][4], int spx, int spy, int op)
{
    const uint32_t m = 0xffff, ma = 0xffff;
    ipxo_rect db = {0, 0, dw, dh}, sb = {0, 0, sw, sh};
    int ox = r.x0, oy = r.y0, x, y, c;
    r = rect_intersect(r, db);
    r = rect_intersect(r, rect_add(sb, ox - spx, oy - spy));
    if (rect_empty(r)) return;
    spx += r.x0 - ox;
    spy += r.y0 - oy;
    for (y = 0; y < r.y1 - r.y0; y++) {
        uint8_t *d = dst + (size_t)(r.y0 + y) * dstride + (size_t)r.x0 * 4;
        for (x = 0; x < r.x1 - r.x0; x++, d += 4) {
            const uint16_t *s = pal16[src[(size_t)(spy + y) * sstride + spx + x]];
            if (op == IPXO_OP_SRC) {
                for (c = 0; c < 4; c++) d[c] = (uint8_t)((uint32_t)s[c] * ma / m >> 8);
            } else {
                uint32_t a = (m - ((uint32_t)s[3] * ma / m)) * 0x101;
                for (c = 0; c < 4; c++) d[c] = (uint8_t)(((uint32_t)d[c] * a + (uint32_t)s[c] * ma) / m >> 8);
            }
        }
    }
}

/* image.(*Paletted).Opaque: only the entries some pixel uses count */
static int paletted_opaque(const uint8_t *src, int sw, int sh, int sstride, const uint16_t pal16[256][4])
{
    int present[256] = {0}, x, y, i;
    for (y = 0; y < sh; y++)
        for (x = 0; x < sw; x++) present[src[(size_t)y * sstride + x]] = 1;
    for (i = 0; i < 256; i++)
        if (present[i] && pal16[i][3] != 0xffff) return 0;
    return 1;
}

int ipxo_scale_bilinear_paletted(uint8_t *dst, int dw, int dh, int dstride, ipxo_rect dr, const uint8_t *src, int sw, int sh,
                                 int sstride, const uint16_t pal16[256][4], ipxo_rect sr, int op)
{
    ipxo_rect adr;
    pal_src p;
    int rc = ks_adr(dw, dh, dr, sr, sw, sh, &adr);
    if (rc) return rc < 0 ? -1 : 0;
    if (op == IPXO_OP_OVER && paletted_opaque(src, sw, sh, sstride, pal16)) op = IPXO_OP_SRC;
    p.pix = src; p.stride = sstride; p.pal = pal16;
    return kernel_scale(dst, dstride, dr, adr, &p, tap_paletted, sr, op, 0);       /* scaleX_Image / scaleX_RGBA64Image */
}

/* ---- the 16-bit image types of the PNG decoder and *image.CMYK ("deep" sources) -----------------------------
 * image.Decode returns *image.NRGBA64 (16-bit truecolour / gray with alpha or tRNS), *image.RGBA64 (16-bit truecolour), *image.Gray16
 * (16-bit gray) for PNGs and *image.CMYK for four-component JPEGs (image_processor.go:47).  No routine of x/image/draw or image/draw
 * specialises on the first three: scaleX_Image (scaleX_RGBA64Image) reads every tap as src.At(x, y).RGBA() and DrawMask falls to drawRGBA
 * (restated for *image.Paletted above; the same code runs here).  *image.CMYK has drawCMYK = color.CMYKToRGB per pixel, which is the
 * top byte of color.CMYK.RGBA().  Pix layouts are Go's: big-endian 16-bit channels (R G B A / Y), C M Y K bytes.
 *   color.NRGBA64.RGBA: c * a / 0xffff, alpha a        color.RGBA64.RGBA: as stored
 *   color.Gray16.RGBA:  (y, y, y, 0xffff)               color.CMYK.RGBA:   w = 0xffff - k*0x101; (0xffff - c*0x101) * w / 0xffff, alpha 0xffff */
int ipxo_deep_bpp(int kind) { return kind == IPXO_DEEP_GRAY16 ? 2 : (kind == IPXO_DEEP_CMYK ? 4 : 8); }

typedef struct { const uint8_t *pix; int stride, kind; } deep_src;

static void deep_rgba(int kind, const uint8_t *p, uint32_t out[4])
{
    if (kind == IPXO_DEEP_GRAY16) {
        out[0] = out[1] = out[2] = (uint32_t)p[0] << 8 | p[1];
        out[3] = 0xffff;
    } else if (kind == IPXO_DEEP_CMYK) {
        const uint32_t w = 0xffff - (uint32_t)p[3] * 0x101;
        out[0] = (0xffff - (uint32_t)p[0] * 0x101) * w / 0xffff;
        out[1] = (0xffff - (uint32_t)p[1] * 0x101) * w / 0xffff;
        out[2] = (0xffff - (uint32_t)p[2] * 0x101) * w / 0xffff;
        out[3] = 0xffff;
    } else {
        const uint32_t r = (uint32_t)p[0] << 8 | p[1], g = (uint32_t)p[2] << 8 | p[3], b = (uint32_t)p[4] << 8 | p[5], a = (uint32_t)p[6] << 8 | p[7];
        if (kind == IPXO_DEEP_NRGBA64) { out[0] = r * a / 0xffff; out[1] = g * a / 0xffff; out[2] = b * a / 0xffff; }
        else { out[0] = r; out[1] = g; out[2] = b; }
        out[3] = a;
    }
}

static void tap_deep(const void *s, int x, int y, uint32_t out[4])
{
    const deep_src *d = (const deep_src *)s;
    deep_rgba(d->kind, d->pix + (size_t)y * d->stride + (size_t)x * ipxo_deep_bpp(d->kind), out);
}

/* At(x, y).RGBA() of every pixel, as little-endian uint16 quadruples (what the product's expansion kernel must produce) */
void ipxo_deep_taps(uint16_t *out, const uint8_t *src, int sw, int sh, int sstride, int kind)
{
    int x, y, c;
    for (y = 0; y < sh; y++)
        for (x = 0; x < sw; x++) {
            uint32_t t[4];
            deep_rgba(kind, src + (size_t)y * sstride + (size_t)x * ipxo_deep_bpp(kind), t);
            for (c = 0; c < 4; c++) out[((size_t)y * sw + x) * 4 + c] = (uint16_t)t[c];
        }
}

void ipxo_draw_deep(uint8_t *dst, int dw, int dh, int dstride, ipxo_rect r, const uint8_t *src, int sw, int sh, int sstride, int kind,
                    int spx, int spy, int op)
{
    const uint32_t m = 0xffff, ma = 0xffff;
    const int bpp = ipxo_deep_bpp(kind);
    ipxo_rect db = {0, 0, dw, dh}, sb = {0, 0, sw, sh};
    int ox = r.x0, oy = r.y0, x, y, c;
    r = rect_intersect(r, db);
    r = rect_intersect(r, rect_add(sb, ox - spx, oy - spy));
    if (rect_empty(r)) return;
    spx += r.x0 - ox;
    spy += r.y0 - oy;
    for (y = 0; y < r.y1 - r.y0; y++) {
        uint8_t *d = dst + (size_t)(r.y0 + y) * dstride + (size_t)r.x0 * 4;
        for (x = 0; x < r.x1 - r.x0; x++, d += 4) {
            uint32_t s[4];
            deep_rgba(kind, src + (size_t)(spy + y) * sstride + (size_t)(spx + x) * bpp, s);
            if (kind == IPXO_DEEP_CMYK) {           /* drawCMYK, for either op: CMYKToRGB, alpha 255 */
                d[0] = (uint8_t)(s[0] >> 8); d[1] = (uint8_t)(s[1] >> 8); d[2] = (uint8_t)(s[2] >> 8); d[3] = 255;
            } else if (op == IPXO_OP_SRC) {         /* drawRGBA */
                for (c = 0; c < 4; c++) d[c] = (uint8_t)(s[c] * ma / m >> 8);
            } else {
                uint32_t a = (m - (s[3] * ma / m)) * 0x101;
                for (c = 0; c < 4; c++) d[c] = (uint8_t)(((uint32_t)d[c] * a + s[c] * ma) / m >> 8);
            }
        }
    }
}

/* (*NRGBA64).Opaque / (*RGBA64).Opaque: both alpha bytes 0xff everywhere; Gray16 and CMYK are opaque */
static int deep_opaque(const uint8_t *src, int sw, int sh, int sstride, int kind)
{
    int x, y;
    if (kind == IPXO_DEEP_GRAY16 || kind == IPXO_DEEP_CMYK) return 1;
    for (y = 0; y < sh; y++)
        for (x = 0; x < sw; x++) {
            const uint8_t *p = src + (size_t)y * sstride + (size_t)x * 8;
            if (p[6] != 0xff || p[7] != 0xff) return 0;
        }
    return 1;
}

int ipxo_scale_bilinear_deep(uint8_t *dst, int dw, int dh, int dstride, ipxo_rect dr, const uint8_t *src, int sw, int sh, int sstride,
                             int kind, ipxo_rect sr, int op)
{
    ipxo_rect adr;
    deep_src p;
    int rc = ks_adr(dw, dh, dr, sr, sw, sh, &adr);
    if (rc) return rc < 0 ? -1 : 0;
    if (op == IPXO_OP_OVER && deep_opaque(src, sw, sh, sstride, kind)) op = IPXO_OP_SRC;
    p.pix = src; p.stride = sstride; p.kind = kind;
    return kernel_scale(dst, dstride, dr, adr, &p, tap_deep, sr, op, 0);           /* scaleX_Image / scaleX_RGBA64Image */
}

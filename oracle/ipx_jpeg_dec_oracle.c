/* ipx_jpeg_dec_oracle.c -- TEST INFRASTRUCTURE: scalar restatement of Go's image/jpeg decoder (baseline part).
 *
 * The reference decodes every upload with image.Decode (image_processor.go:47); for a JPEG that is Go 1.24's
 * image/jpeg (go.mod:3; reader.go, scan.go, huffman.go, idct.go), absent from /root/reference and not runnable here
 * (no Go toolchain): PARITY UNPINNED against Go itself.  Restated, function by function:
 *   decode            marker loop: SOI, DQT, SOF0 / SOF1, DHT, DRI, SOS, APPn / COM skipped, Adobe / JFIF noted (isRGB)
 *   processSOF        8-bit precision; 3 components, Y sampling (1|2) x (1|2), chroma 1 x 1 -> 4:4:4 / 4:2:2 / 4:2:0 / 4:4:0;
 *                     or 1 component -> *image.Gray (its sampling factors count as 1 x 1, 8 x 8 MCUs)
 *   makeImg           image.NewYCbCr(Rect(0, 0, 8*h0*mxx, 8*v0*myy), ratio).SubImage(Rect(0, 0, w, h)): MCU-padded strides
 *   processSOS        one interleaved scan; DC prediction; F.2.2.1 / F.2.2.2 symbol decoding; restart intervals
 *   huffman.go        canonical codes from BITS / HUFFVAL; receiveExtend
 *   reconstructBlock  b[unzig[zig]] *= qt[zig]; idct; +128, clip, store
 *   idct.go           the Chen-Wang 32-bit integer IDCT of the MPEG-2 reference decoder (w1..w7, r2 = 181)
 * Out of this oracle's scope (the product reports them unsupported and the worker keeps Go's CPU path for such files):
 * progressive (SOF2), CMYK / RGB JPEGs, 4:1:1 / 4:1:0, multi-scan baseline files, 12-bit precision.
 * Pins: decode(encode(x)) reproduces the pinned encoder's coefficients exactly (tests/test_jpeg_decode.py), and
 * libjpeg (Pillow) decodes the same files to within the known +-1..2 of a different IDCT.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int w, h, ratio;          /* ratio: 0 = 4:4:4, 1 = 4:2:2, 2 = 4:2:0, 3 = 4:4:0 (image.YCbCrSubsampleRatio); 4 = *image.Gray (y only) */
    int ystride, cstride, yrows, crows;
    uint8_t *y, *cb, *cr;
    int dc_wide;              /* some DC value left the int16 range (Go keeps int32 and decodes on; the GPU pipeline reports such files unsupported) */
} ipxo_decoded;

static const uint8_t k_unzig[64] = {
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

/* ---- idct.go ---- */
#define W1 2841
#define W2 2676
#define W3 2408
#define W5 1609
#define W6 1108
#define W7 565
#define R2 181

static void idct(int32_t *src)
{
    for (int y = 0; y < 8; y++) {
        int32_t *s = src + 8 * y;
        if (s[1] == 0 && s[2] == 0 && s[3] == 0 && s[4] == 0 && s[5] == 0 && s[6] == 0 && s[7] == 0) {
            int32_t dc = (int32_t)((uint32_t)s[0] << 3);
            for (int i = 0; i < 8; i++) s[i] = dc;
            continue;
        }
        int32_t x0 = (int32_t)((uint32_t)s[0] << 11) + 128, x1 = (int32_t)((uint32_t)s[4] << 11), x2 = s[6], x3 = s[2], x4 = s[1], x5 = s[7], x6 = s[5], x7 = s[3];
        int32_t x8 = W7 * (x4 + x5);
        x4 = x8 + (W1 - W7) * x4;
        x5 = x8 - (W1 + W7) * x5;
        x8 = W3 * (x6 + x7);
        x6 = x8 - (W3 - W5) * x6;
        x7 = x8 - (W3 + W5) * x7;
        x8 = x0 + x1;
        x0 -= x1;
        x1 = W6 * (x3 + x2);
        x2 = x1 - (W2 + W6) * x2;
        x3 = x1 + (W2 - W6) * x3;
        x1 = x4 + x6;
        x4 -= x6;
        x6 = x5 + x7;
        x5 -= x7;
        x7 = x8 + x3;
        x8 -= x3;
        x3 = x0 + x2;
        x0 -= x2;
        x2 = (R2 * (x4 + x5) + 128) >> 8;
        x4 = (R2 * (x4 - x5) + 128) >> 8;
        s[0] = (x7 + x1) >> 8; s[1] = (x3 + x2) >> 8; s[2] = (x0 + x4) >> 8; s[3] = (x8 + x6) >> 8;
        s[4] = (x8 - x6) >> 8; s[5] = (x0 - x4) >> 8; s[6] = (x3 - x2) >> 8; s[7] = (x7 - x1) >> 8;
    }
    for (int x = 0; x < 8; x++) {
        int32_t *s = src + x;
        int32_t y0 = (int32_t)((uint32_t)s[0] << 8) + 8192, y1 = (int32_t)((uint32_t)s[32] << 8), y2 = s[48], y3 = s[16], y4 = s[8], y5 = s[56], y6 = s[40], y7 = s[24];
        int32_t y8 = W7 * (y4 + y5) + 4;
        y4 = (y8 + (W1 - W7) * y4) >> 3;
        y5 = (y8 - (W1 + W7) * y5) >> 3;
        y8 = W3 * (y6 + y7) + 4;
        y6 = (y8 - (W3 - W5) * y6) >> 3;
        y7 = (y8 - (W3 + W5) * y7) >> 3;
        y8 = y0 + y1;
        y0 -= y1;
        y1 = W6 * (y3 + y2) + 4;
        y2 = (y1 - (W2 + W6) * y2) >> 3;
        y3 = (y1 + (W2 - W6) * y3) >> 3;
        y1 = y4 + y6;
        y4 -= y6;
        y6 = y5 + y7;
        y5 -= y7;
        y7 = y8 + y3;
        y8 -= y3;
        y3 = y0 + y2;
        y0 -= y2;
        y2 = (R2 * (y4 + y5) + 128) >> 8;
        y4 = (R2 * (y4 - y5) + 128) >> 8;
        s[0] = (y7 + y1) >> 14; s[8] = (y3 + y2) >> 14; s[16] = (y0 + y4) >> 14; s[24] = (y8 + y6) >> 14;
        s[32] = (y8 - y6) >> 14; s[40] = (y0 - y4) >> 14; s[48] = (y3 - y2) >> 14; s[56] = (y7 - y1) >> 14;
    }
}

/* ---- huffman.go ---- */
typedef struct {
    int ok;
    int32_t mincode[16], maxcode[16], valptr[16];
    uint8_t vals[256];
    int nvals;
} huff;

typedef struct {
    const uint8_t *p, *end;
    uint32_t acc;
    int n;
    int err;
} bitrd;

static int next_bit(bitrd *b)
{
    if (b->n == 0) {
        if (b->p >= b->end) { b->err = 1; return 0; }
        uint8_t c = *b->p++;
        if (c == 0xff) {
            if (b->p >= b->end || *b->p != 0x00) { b->err = 1; return 0; }   /* a marker inside entropy-coded data */
            b->p++;
        }
        b->acc = c; b->n = 8;
    }
    b->n--;
    return (b->acc >> b->n) & 1;
}
static int decode_huff(bitrd *b, const huff *h)
{
    int32_t code = 0;
    for (int i = 0; i < 16; i++) {
        code = code << 1 | next_bit(b);
        if (b->err) return 0;
        if (h->maxcode[i] >= 0 && code <= h->maxcode[i] && code >= h->mincode[i]) return h->vals[h->valptr[i] + code - h->mincode[i]];
    }
    b->err = 1;   /* "bad Huffman code" */
    return 0;
}
static int32_t receive_extend(bitrd *b, int t)
{
    int32_t x = 0;
    for (int i = 0; i < t; i++) x = x << 1 | next_bit(b);
    if (t && x < (1 << (t - 1))) x += (int32_t)((uint32_t)-1 << t) + 1;
    return x;
}

static uint32_t be16(const uint8_t *p) { return (uint32_t)p[0] << 8 | p[1]; }

void ipxo_decoded_free(ipxo_decoded *d) { free(d->y); free(d->cb); free(d->cr); memset(d, 0, sizeof *d); }

/* 0 ok; -1 malformed; -2 valid JPEG outside this restatement (see the header); coefs (may be NULL) receives the quantised
 * coefficients of every block in scan order, natural (de-zig-zagged) index order, 64 int16 per block */
int ipxo_jpeg_decode(const uint8_t *data, size_t len, ipxo_decoded *out, int16_t *coefs, size_t coefs_cap)
{
    memset(out, 0, sizeof *out);
    if (len < 4 || data[0] != 0xff || data[1] != 0xd8) return -1;
    uint16_t quant[4][64];
    int have_q[4] = {0, 0, 0, 0};
    huff hf[2][4];
    memset(hf, 0, sizeof hf);
    int w = 0, h = 0, ncomp = 0, ch[3] = {0}, cv[3] = {0}, ctq[3] = {0}, cid[3] = {0};
    int ri = 0, jfif = 0, adobe_valid = 0, adobe_transform = 0;
    size_t i = 2;
    for (;;) {
        if (i + 2 > len) return -1;
        if (data[i] != 0xff) return -1;
        while (i + 1 < len && data[i + 1] == 0xff) i++;   /* fill bytes */
        if (i + 2 > len) return -1;
        const int m = data[i + 1];
        i += 2;
        if (m == 0xd9) return -1;                 /* EOI before any scan: "missing SOS marker" */
        if (m == 0x00 || (m >= 0xd0 && m <= 0xd7)) continue;
        if (i + 2 > len) return -1;
        const size_t n = be16(data + i);
        if (n < 2 || i + n > len) return -1;
        const uint8_t *s = data + i + 2;
        const size_t sn = n - 2;
        if (m == 0xc0 || m == 0xc1) {
            if (ncomp) return -1;                 /* "multiple SOF markers" */
            if (sn < 6) return -1;
            if (s[0] != 8) return -2;             /* precision */
            h = (int)be16(s + 1); w = (int)be16(s + 3); ncomp = s[5];
            if (ncomp != 3 && ncomp != 1) return ncomp == 4 ? -2 : -1;
            if (sn != (size_t)(6 + 3 * ncomp) || w <= 0 || h <= 0) return -1;
            for (int c = 0; c < ncomp; c++) {
                cid[c] = s[6 + 3 * c]; ch[c] = s[7 + 3 * c] >> 4; cv[c] = s[7 + 3 * c] & 15; ctq[c] = s[8 + 3 * c];
                if (ctq[c] > 3 || ch[c] < 1 || ch[c] > 4 || cv[c] < 1 || cv[c] > 4) return -1;
                for (int j = 0; j < c; j++)
                    if (cid[j] == cid[c]) return -1;   /* processSOF: "repeated component identifier" (B.2.2) */
            }
            if (ncomp == 1) { ch[0] = cv[0] = 1; }        /* processSOF: "the component's (h, v) is effectively always (1, 1)" */
            else if (ch[1] != 1 || cv[1] != 1 || ch[2] != 1 || cv[2] != 1 || ch[0] > 2 || cv[0] > 2) return -2;
        } else if (m == 0xc2) {
            return -2;                            /* progressive */
        } else if (m == 0xc4) {
            size_t k = 0;
            while (k < sn) {
                if (k + 17 > sn) return -1;
                const int tc = s[k] >> 4, th = s[k] & 15;
                if (tc > 1 || th > 3) return -1;
                huff *t = &hf[tc][th];
                int total = 0;
                for (int b = 0; b < 16; b++) total += s[k + 1 + b];
                if (total == 0 || total > 256 || k + 17 + (size_t)total > sn) return -1;
                memcpy(t->vals, s + k + 17, (size_t)total);
                t->nvals = total;
                int32_t code = 0, idx = 0;
                for (int b = 0; b < 16; b++) {
                    const int cnt = s[k + 1 + b];
                    code <<= 1;
                    if (cnt == 0) { t->maxcode[b] = -1; t->mincode[b] = -1; t->valptr[b] = -1; continue; }
                    t->mincode[b] = code; t->valptr[b] = idx;
                    code += cnt; idx += cnt;
                    t->maxcode[b] = code - 1;
                }
                t->ok = 1;
                k += 17 + (size_t)total;
            }
        } else if (m == 0xdb) {
            size_t k = 0;
            while (k < sn) {
                const int pq = s[k] >> 4, tq = s[k] & 15;
                if (tq > 3 || pq > 1) return -1;
                const size_t need = pq ? 128 : 64;
                if (k + 1 + need > sn) return -1;
                for (int z = 0; z < 64; z++) quant[tq][z] = pq ? (uint16_t)be16(s + k + 1 + 2 * z) : s[k + 1 + z];
                have_q[tq] = 1;
                k += 1 + need;
            }
        } else if (m == 0xdd) {
            if (sn != 2) return -1;
            ri = (int)be16(s);
        } else if (m == 0xe0) {
            if (sn >= 5 && !memcmp(s, "JFIF\0", 5)) jfif = 1;
        } else if (m == 0xee) {
            if (sn >= 12 && !memcmp(s, "Adobe", 5)) { adobe_valid = 1; adobe_transform = s[11]; }
        } else if (m == 0xda) {
            if (!ncomp) return -1;
            if (sn < 1 || s[0] != ncomp) return sn >= 1 && s[0] >= 1 && s[0] <= 3 ? -2 : -1;
            if (sn != (size_t)(4 + 2 * ncomp)) return -1;
            int td[3] = {0, 0, 0}, ta[3] = {0, 0, 0};
            for (int c = 0; c < ncomp; c++) {
                if (s[1 + 2 * c] != cid[c]) return -2;       /* components out of frame order */
                td[c] = s[2 + 2 * c] >> 4; ta[c] = s[2 + 2 * c] & 15;
                if (td[c] > 3 || ta[c] > 3 || !hf[0][td[c]].ok || !hf[1][ta[c]].ok || !have_q[ctq[c]]) return -1;
            }
            /* isRGB: not JFIF and (Adobe transform "unknown" or component ids 'R','G','B') */
            if (ncomp == 3 && !jfif && ((adobe_valid && adobe_transform == 0) || (cid[0] == 'R' && cid[1] == 'G' && cid[2] == 'B'))) return -2;
            const int h0 = ch[0], v0 = cv[0];
            const int mxx = (w + 8 * h0 - 1) / (8 * h0), myy = (h + 8 * v0 - 1) / (8 * v0);
            out->w = w; out->h = h;
            out->ratio = ncomp == 1 ? 4 : (h0 == 1 ? (v0 == 1 ? 0 : 3) : (v0 == 1 ? 1 : 2));
            out->ystride = 8 * h0 * mxx; out->yrows = 8 * v0 * myy;
            out->cstride = 8 * mxx; out->crows = 8 * myy;
            out->y = (uint8_t *)calloc((size_t)out->ystride * out->yrows, 1);
            out->cb = (uint8_t *)calloc((size_t)out->cstride * out->crows, 1);
            out->cr = (uint8_t *)calloc((size_t)out->cstride * out->crows, 1);
            if (!out->y || !out->cb || !out->cr) { ipxo_decoded_free(out); return -3; }
            bitrd br = {data + i + n, data + len, 0, 0, 0};
            int32_t dc[3] = {0, 0, 0};
            int mcu = 0, expected_rst = 0xd0;
            size_t nblk = 0;
            for (int my = 0; my < myy; my++)
                for (int mx = 0; mx < mxx; mx++) {
                    for (int c = 0; c < ncomp; c++) {
                        const int hi = ch[c], vi = cv[c];
                        for (int j = 0; j < hi * vi; j++) {
                            const int bx = hi * mx + j % hi, by = vi * my + j / hi;
                            int32_t b[64];
                            memset(b, 0, sizeof b);
                            int t = decode_huff(&br, &hf[0][td[c]]);
                            if (br.err || t > 16) { ipxo_decoded_free(out); return -1; }
                            dc[c] += receive_extend(&br, t);
                            if (dc[c] < -32768 || dc[c] > 32767) out->dc_wide = 1;
                            b[0] = dc[c];
                            for (int zig = 1; zig < 64; zig++) {
                                const int v = decode_huff(&br, &hf[1][ta[c]]);
                                if (br.err) { ipxo_decoded_free(out); return -1; }
                                const int r = v >> 4, sz = v & 15;
                                if (sz) {
                                    zig += r;
                                    if (zig > 63) break;
                                    b[k_unzig[zig]] = receive_extend(&br, sz);
                                } else {
                                    if (r != 15) break;
                                    zig += 15;
                                }
                            }
                            if (br.err) { ipxo_decoded_free(out); return -1; }
                            if (coefs && (nblk + 1) * 64 <= coefs_cap)
                                for (int z = 0; z < 64; z++) coefs[nblk * 64 + z] = (int16_t)b[z];
                            nblk++;
                            const uint16_t *qt = quant[ctq[c]];
                            for (int zig = 0; zig < 64; zig++) b[k_unzig[zig]] *= qt[zig];
                            idct(b);
                            uint8_t *dst = c == 0 ? out->y : (c == 1 ? out->cb : out->cr);
                            const int stride = c == 0 ? out->ystride : out->cstride;
                            dst += 8 * ((size_t)by * stride + bx);
                            for (int yy = 0; yy < 8; yy++)
                                for (int xx = 0; xx < 8; xx++) {
                                    int32_t v2 = b[8 * yy + xx];
                                    dst[yy * stride + xx] = (uint8_t)(v2 < -128 ? 0 : (v2 > 127 ? 255 : v2 + 128));
                                }
                        }
                    }
                    mcu++;
                    if (ri > 0 && mcu % ri == 0 && mcu < mxx * myy) {
                        /* the restart marker follows on the next byte boundary; Go resets bits and DC predictions */
                        if (br.p + 2 > br.end || br.p[0] != 0xff || br.p[1] != expected_rst) { ipxo_decoded_free(out); return -2; }
                        br.p += 2;
                        expected_rst = expected_rst == 0xd7 ? 0xd0 : expected_rst + 1;
                        br.n = 0; br.acc = 0;
                        dc[0] = dc[1] = dc[2] = 0;
                    }
                }
            return 0;
        }
        i += n;
    }
}

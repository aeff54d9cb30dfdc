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
    int dc_wide;              /* some DC value (or, with several scans, any coefficient) left the int16 range (Go keeps int32 and decodes on; the GPU pipeline reports such files unsupported) */
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

/* ============================================================================================================================
 * The whole of Go's decoder for the files the reference accepts beyond single-scan baseline: progressive (SOF2: spectral selection,
 * successive approximation, EOB runs, refinement passes) and multi-scan sequential files, with Go's marker loop -- garbage between
 * segments skipped, "\xff\x00" outside a scan ignored, stray RSTn ignored, EOI REQUIRED (a file that ends after its last scan is
 * io.ErrUnexpectedEOF in Go).  Restated from Go 1.24 image/jpeg: reader.go (decode, processSOF, processDQT, processDRI, applyBlack
 * not needed), huffman.go (processDHT, decodeHuffman, receiveExtend, decodeBit, decodeBits), scan.go (processSOS, refine,
 * refineNonZeroes, reconstructProgressiveImage, reconstructBlock).  Same scope of pixel formats as above (3 components with 1x1
 * chroma and Y up to 2x2, or 1 component); everything else is -2.
 * Pins (tests/test_jpeg_progressive.py): a progressive file and the baseline file libjpeg writes for the same image carry the same
 * quantised coefficients, so this decoder must reproduce the baseline decoder's planes EXACTLY on the visible region; and libjpeg
 * (Pillow) decodes the progressive files to within the +-2 of its different IDCT.
 * ============================================================================================================================ */
typedef struct {
    int ncodes;                      /* 0: uninitialised table ("uninitialized Huffman table" when a scan uses it) */
    int32_t mincode[16], maxcode[16], valptr[16];
    uint8_t vals[256];
} huff2;

typedef struct {
    const uint8_t *data;
    size_t len, pos;                 /* pos: next unread byte of the file */
    uint32_t acc; int nbits;         /* the entropy decoder's bit buffer (d.bits) */
    int err;                         /* -1 malformed, -2 unsupported */
    /* frame */
    int w, h, ncomp, progressive, baseline;
    int cid[3], ch[3], cv[3], ctq[3];
    uint16_t quant[4][64];
    huff2 hf[2][4];
    int ri, jfif, adobe_valid, adobe_transform;
    uint16_t eobrun;
    int mxx, myy;
    int32_t *prog[3];                /* progCoeffs: blocks of 64 int32 per component, NULL until a scan touches the component */
    ipxo_decoded *out;
    int have_img;
} jdec;

static void d_fail(jdec *d, int code) { if (!d->err) d->err = code; }

/* readByteStuffedByte: inside entropy-coded data 0xff 0x00 is a 0xff; 0xff followed by anything else is a marker the entropy decoder
 * must not consume ("missing 0xff00 sequence"), and running out of file is "short Huffman data" */
static int d_next_bit(jdec *d)
{
    if (d->nbits == 0) {
        if (d->pos >= d->len) { d_fail(d, -1); return 0; }
        uint8_t c = d->data[d->pos];
        if (c == 0xff) {
            if (d->pos + 1 >= d->len || d->data[d->pos + 1] != 0x00) { d_fail(d, -1); return 0; }
            d->pos += 2;
        } else d->pos += 1;
        d->acc = c; d->nbits = 8;
    }
    d->nbits--;
    return (d->acc >> d->nbits) & 1;
}
static uint32_t d_bits(jdec *d, int n)    /* decodeBits */
{
    uint32_t x = 0;
    for (int i = 0; i < n && !d->err; i++) x = x << 1 | (uint32_t)d_next_bit(d);
    return x;
}
static int d_huff(jdec *d, const huff2 *h)   /* decodeHuffman */
{
    if (h->ncodes == 0) { d_fail(d, -1); return 0; }
    int32_t code = 0;
    for (int i = 0; i < 16; i++) {
        code = code << 1 | d_next_bit(d);
        if (d->err) return 0;
        if (h->maxcode[i] >= 0 && code <= h->maxcode[i] && code >= h->mincode[i]) return h->vals[h->valptr[i] + code - h->mincode[i]];
    }
    d_fail(d, -1);                    /* "bad Huffman code" */
    return 0;
}
static int32_t d_receive_extend(jdec *d, int t)
{
    int32_t x = (int32_t)d_bits(d, t);
    if (t && x < (1 << (t - 1))) x += (int32_t)((uint32_t)-1 << t) + 1;
    return x;
}

static void d_reconstruct(jdec *d, int32_t *blk, int bx, int by, int c)
{
    int32_t b[64];
    memcpy(b, blk, sizeof b);
    for (int z = 0; z < 64; z++) if (b[z] < -32768 || b[z] > 32767) d->out->dc_wide = 1;   /* Go's int32 decodes on; the product hands such a file back */
    const uint16_t *qt = d->quant[d->ctq[c]];
    for (int zig = 0; zig < 64; zig++) b[k_unzig[zig]] = (int32_t)((uint32_t)b[k_unzig[zig]] * (uint32_t)qt[zig]);
    idct(b);
    ipxo_decoded *o = d->out;
    uint8_t *dst = c == 0 ? o->y : (c == 1 ? o->cb : o->cr);
    const int stride = c == 0 ? o->ystride : o->cstride;
    dst += 8 * ((size_t)by * stride + bx);
    for (int yy = 0; yy < 8; yy++)
        for (int xx = 0; xx < 8; xx++) {
            const int32_t v = b[8 * yy + xx];
            dst[yy * stride + xx] = (uint8_t)(v < -128 ? 0 : (v > 127 ? 255 : v + 128));
        }
}

/* The product holds coefficients in int16 (Go: int32) and hands a file back as soon as ANY value it writes does not fit -- also one a
 * later scan would overwrite.  The flag follows the same rule, at the same writes, so that the verdicts can be compared. */
#define D_WIDE(d, v) do { if ((v) < -32768 || (v) > 32767) (d)->out->dc_wide = 1; } while (0)

/* refineNonZeroes: refine the non-zero entries of b in zig-zag order; if nz >= 0 the first nz zero entries are skipped over */
static int32_t d_refine_nonzeroes(jdec *d, int32_t *b, int32_t zig, int32_t zig_end, int32_t nz, int32_t delta)
{
    for (; zig <= zig_end; zig++) {
        const int u = k_unzig[zig];
        if (b[u] == 0) {
            if (nz == 0) break;
            nz--;
            continue;
        }
        const int bit = d_next_bit(d);
        if (d->err) return 0;
        if (!bit) continue;
        if (b[u] >= 0) b[u] += delta; else b[u] -= delta;
        D_WIDE(d, b[u]);
    }
    return zig;
}

/* refine: a successive approximation refinement block (G.1.2) */
static void d_refine(jdec *d, int32_t *b, const huff2 *h, int32_t zig_start, int32_t zig_end, int32_t delta)
{
    if (zig_start == 0) {            /* refining a DC component is trivial */
        const int bit = d_next_bit(d);
        if (!d->err && bit) { b[0] |= delta; D_WIDE(d, b[0]); }
        return;
    }
    int32_t zig = zig_start;
    if (d->eobrun == 0) {
        for (; zig <= zig_end; zig++) {
            int32_t z = 0;
            const int value = d_huff(d, h);
            if (d->err) return;
            const int val0 = value >> 4, val1 = value & 0x0f;
            if (val1 == 0) {
                if (val0 != 0x0f) {
                    d->eobrun = (uint16_t)(1u << val0);
                    if (val0 != 0) d->eobrun |= (uint16_t)d_bits(d, val0);
                    if (d->err) return;
                    break;
                }
            } else if (val1 == 1) {
                z = delta;
                const int bit = d_next_bit(d);
                if (d->err) return;
                if (!bit) z = -z;
            } else { d_fail(d, -1); return; }      /* "unexpected Huffman code" */
            zig = d_refine_nonzeroes(d, b, zig, zig_end, val0, delta);
            if (d->err) return;
            if (zig > zig_end) { d_fail(d, -1); return; }   /* "too many coefficients" */
            if (z != 0) b[k_unzig[zig]] = z;
        }
    }
    if (d->eobrun > 0) {
        d->eobrun--;
        (void)d_refine_nonzeroes(d, b, zig, zig_end, -1, delta);
    }
}

static int d_make_img(jdec *d)
{
    ipxo_decoded *o = d->out;
    const int h0 = d->ch[0], v0 = d->cv[0];
    o->w = d->w; o->h = d->h;
    o->ratio = d->ncomp == 1 ? 4 : (h0 == 1 ? (v0 == 1 ? 0 : 3) : (v0 == 1 ? 1 : 2));
    o->ystride = 8 * h0 * d->mxx; o->yrows = 8 * v0 * d->myy;
    o->cstride = 8 * d->mxx; o->crows = 8 * d->myy;
    o->y = (uint8_t *)calloc((size_t)o->ystride * o->yrows, 1);
    o->cb = (uint8_t *)calloc((size_t)o->cstride * o->crows, 1);
    o->cr = (uint8_t *)calloc((size_t)o->cstride * o->crows, 1);
    d->have_img = 1;
    return o->y && o->cb && o->cr;
}

static void d_sos(jdec *d, const uint8_t *s, size_t n)
{
    if (d->ncomp == 0) { d_fail(d, -1); return; }                                 /* "missing SOF marker" */
    if (n < 6 || (size_t)(4 + 2 * d->ncomp) < n || n % 2 != 0) { d_fail(d, -1); return; }
    const int ncomp = s[0];
    if (n != (size_t)(4 + 2 * ncomp)) { d_fail(d, -1); return; }
    int comp_index[3] = {0, 0, 0}, td[3] = {0, 0, 0}, ta[3] = {0, 0, 0}, total_hv = 0;
    for (int i = 0; i < ncomp; i++) {
        int ci = -1;
        for (int j = 0; j < d->ncomp; j++) if (s[1 + 2 * i] == d->cid[j]) ci = j;
        if (ci < 0) { d_fail(d, -1); return; }                                    /* "unknown component selector" */
        comp_index[i] = ci;
        for (int j = 0; j < i; j++) if (comp_index[j] == ci) { d_fail(d, -1); return; }   /* "repeated component selector" */
        total_hv += d->ch[ci] * d->cv[ci];
        td[i] = s[2 + 2 * i] >> 4; ta[i] = s[2 + 2 * i] & 0x0f;
        if (td[i] > 3 || (d->baseline && td[i] > 1) || ta[i] > 3 || (d->baseline && ta[i] > 1)) { d_fail(d, -1); return; }
    }
    if (d->ncomp > 1 && total_hv > 10) { d_fail(d, -1); return; }
    int32_t zig_start = 0, zig_end = 63;
    uint32_t ah = 0, al = 0;
    if (d->progressive) {
        zig_start = s[1 + 2 * ncomp]; zig_end = s[2 + 2 * ncomp];
        ah = s[3 + 2 * ncomp] >> 4; al = s[3 + 2 * ncomp] & 0x0f;
        if ((zig_start == 0 && zig_end != 0) || zig_start > zig_end || zig_end >= 64) { d_fail(d, -1); return; }
        if (zig_start != 0 && ncomp != 1) { d_fail(d, -1); return; }
        if (ah != 0 && ah != al + 1) { d_fail(d, -1); return; }
    }
    const int h0 = d->ch[0], v0 = d->cv[0];
    d->mxx = (d->w + 8 * h0 - 1) / (8 * h0); d->myy = (d->h + 8 * v0 - 1) / (8 * v0);
    if (!d->have_img && !d_make_img(d)) { d_fail(d, -3); return; }
    if (d->progressive)
        for (int i = 0; i < ncomp; i++) {
            const int ci = comp_index[i];
            if (!d->prog[ci]) {
                d->prog[ci] = (int32_t *)calloc((size_t)d->mxx * d->myy * d->ch[ci] * d->cv[ci] * 64, sizeof(int32_t));
                if (!d->prog[ci]) { d_fail(d, -3); return; }
            }
        }
    d->acc = 0; d->nbits = 0;
    int mcu = 0, expected_rst = 0xd0, block_count = 0;
    int32_t dc[3] = {0, 0, 0};
    for (int my = 0; my < d->myy; my++)
        for (int mx = 0; mx < d->mxx; mx++) {
            for (int i = 0; i < ncomp; i++) {
                const int ci = comp_index[i], hi = d->ch[ci], vi = d->cv[ci];
                for (int j = 0; j < hi * vi; j++) {
                    int bx, by;
                    if (ncomp != 1) { bx = hi * mx + j % hi; by = vi * my + j / hi; }
                    else {
                        const int q = d->mxx * hi;
                        bx = block_count % q; by = block_count / q;
                        block_count++;
                        if (bx * 8 >= d->w || by * 8 >= d->h) continue;
                    }
                    int32_t local[64];
                    int32_t *b = local;
                    if (d->progressive) b = d->prog[ci] + ((size_t)by * d->mxx * hi + bx) * 64;
                    else memset(local, 0, sizeof local);
                    if (ah != 0) {
                        d_refine(d, b, &d->hf[1][ta[i]], zig_start, zig_end, (int32_t)(1u << al));
                        if (d->err) return;
                    } else {
                        int32_t zig = zig_start;
                        if (zig == 0) {
                            zig++;
                            const int value = d_huff(d, &d->hf[0][td[i]]);
                            if (d->err) return;
                            if (value > 16) { d_fail(d, -2); return; }            /* UnsupportedError("excessive DC component") */
                            dc[ci] += d_receive_extend(d, value);
                            if (d->err) return;
                            if (dc[ci] < -32768 || dc[ci] > 32767) d->out->dc_wide = 1;
                            b[0] = (int32_t)((uint32_t)dc[ci] << al);
                            D_WIDE(d, b[0]);
                        }
                        if (zig <= zig_end && d->eobrun > 0) d->eobrun--;
                        else {
                            const huff2 *h = &d->hf[1][ta[i]];
                            for (; zig <= zig_end; zig++) {
                                const int value = d_huff(d, h);
                                if (d->err) return;
                                const int val0 = value >> 4, val1 = value & 0x0f;
                                if (val1 != 0) {
                                    zig += val0;
                                    if (zig > zig_end) break;
                                    const int32_t ac = d_receive_extend(d, val1);
                                    if (d->err) return;
                                    b[k_unzig[zig]] = (int32_t)((uint32_t)ac << al);
                                    D_WIDE(d, b[k_unzig[zig]]);
                                } else {
                                    if (val0 != 0x0f) {
                                        d->eobrun = (uint16_t)(1u << val0);
                                        if (val0 != 0) d->eobrun |= (uint16_t)d_bits(d, val0);
                                        if (d->err) return;
                                        d->eobrun--;
                                        break;
                                    }
                                    zig += 0x0f;
                                }
                            }
                        }
                    }
                    if (d->progressive) continue;          /* reconstructed after EOI, from the accumulated coefficients */
                    d_reconstruct(d, b, bx, by, ci);
                }
            }
            mcu++;
            if (d->ri > 0 && mcu % d->ri == 0 && mcu < d->mxx * d->myy) {
                /* the RSTn marker follows on the next byte; anything else makes Go search for it (findRST), which this restatement
                 * reports as unsupported: the product does not guess at resynchronisation either */
                if (d->pos + 2 > d->len) { d_fail(d, -1); return; }
                if (d->data[d->pos] != 0xff || d->data[d->pos + 1] != expected_rst) { d_fail(d, -2); return; }
                d->pos += 2;
                expected_rst = expected_rst == 0xd7 ? 0xd0 : expected_rst + 1;
                d->acc = 0; d->nbits = 0;
                dc[0] = dc[1] = dc[2] = 0;
                d->eobrun = 0;
            }
        }
}

static void d_free(jdec *d) { for (int c = 0; c < 3; c++) free(d->prog[c]); }

/* 0 ok; -1 malformed; -2 valid for Go but outside this restatement; -3 out of memory */
int ipxo_jpeg_decode_full(const uint8_t *data, size_t len, ipxo_decoded *out)
{
    memset(out, 0, sizeof *out);
    jdec D;
    jdec *d = &D;
    memset(d, 0, sizeof D);
    d->data = data; d->len = len; d->out = out;
    if (len < 2 || data[0] != 0xff || data[1] != 0xd8) return -1;
    d->pos = 2;
    int seen_eoi = 0;
    while (!d->err) {
        /* marker loop of decode(): two bytes; bytes that are not 0xff are skipped ("libjpeg is liberal in what it accepts") */
        if (d->pos + 2 > len) { d_fail(d, -1); break; }            /* io.ErrUnexpectedEOF: no EOI */
        uint8_t t0 = data[d->pos], t1 = data[d->pos + 1];
        d->pos += 2;
        while (t0 != 0xff) {
            t0 = t1;
            if (d->pos >= len) { d_fail(d, -1); break; }
            t1 = data[d->pos++];
        }
        if (d->err) break;
        int marker = t1;
        if (marker == 0) continue;                                  /* "\xff\x00": extraneous data */
        while (marker == 0xff) {                                    /* fill bytes */
            if (d->pos >= len) { d_fail(d, -1); break; }
            marker = data[d->pos++];
        }
        if (d->err) break;
        if (marker == 0xd9) { seen_eoi = 1; break; }
        if (marker >= 0xd0 && marker <= 0xd7) continue;             /* a stray restart marker after the last interval */
        if (d->pos + 2 > len) { d_fail(d, -1); break; }
        const int n = (int)be16(data + d->pos) - 2;
        d->pos += 2;
        if (n < 0) { d_fail(d, -1); break; }
        if (d->pos + (size_t)n > len) { d_fail(d, -1); break; }
        const uint8_t *s = data + d->pos;
        const size_t sn = (size_t)n;
        d->pos += sn;                                               /* (a scan's entropy data follows its header: d_sos reads on from here) */
        if (marker == 0xc0 || marker == 0xc1 || marker == 0xc2) {
            d->baseline = marker == 0xc0; d->progressive = marker == 0xc2;
            if (d->ncomp) { d_fail(d, -1); break; }
            if (sn == 9) d->ncomp = 1; else if (sn == 15) d->ncomp = 3; else { d_fail(d, -2); break; }   /* 4 components (CMYK): unsupported here */
            if (s[0] != 8) { d_fail(d, -2); break; }
            d->h = (int)be16(s + 1); d->w = (int)be16(s + 3);
            if (s[5] != d->ncomp) { d_fail(d, -1); break; }
            for (int c = 0; c < d->ncomp && !d->err; c++) {
                d->cid[c] = s[6 + 3 * c];
                for (int j = 0; j < c; j++) if (d->cid[j] == d->cid[c]) d_fail(d, -1);
                d->ctq[c] = s[8 + 3 * c];
                if (d->ctq[c] > 3) d_fail(d, -1);
                int hh = s[7 + 3 * c] >> 4, vv = s[7 + 3 * c] & 15;
                if (hh < 1 || hh > 4 || vv < 1 || vv > 4) d_fail(d, -1);
                else if (hh == 3 || vv == 3) d_fail(d, -2);
                if (d->ncomp == 1) { hh = 1; vv = 1; }
                d->ch[c] = hh; d->cv[c] = vv;
            }
            if (d->err) break;
            if (d->ncomp == 3 && (d->ch[1] != 1 || d->cv[1] != 1 || d->ch[2] != 1 || d->cv[2] != 1 || d->ch[0] > 2 || d->cv[0] > 2)) { d_fail(d, -2); break; }
            if (d->w <= 0 || d->h <= 0) { d_fail(d, -1); break; }
        } else if (marker == 0xc4) {
            size_t k = 0;
            while (k < sn && !d->err) {
                if (sn - k < 17) { d_fail(d, -1); break; }
                const int tc = s[k] >> 4, th = s[k] & 15;
                if (tc > 1 || th > 3 || (d->baseline && th > 1)) { d_fail(d, -1); break; }
                huff2 *t = &d->hf[tc][th];
                int total = 0;
                for (int b = 0; b < 16; b++) total += s[k + 1 + b];
                if (total == 0 || total > 256 || k + 17 + (size_t)total > sn) { t->ncodes = 0; d_fail(d, -1); break; }
                t->ncodes = total;
                memcpy(t->vals, s + k + 17, (size_t)total);
                int32_t code = 0, idx = 0;
                for (int b = 0; b < 16; b++) {
                    const int cnt = s[k + 1 + b];
                    code <<= 1;
                    if (cnt == 0) { t->maxcode[b] = -1; t->mincode[b] = -1; t->valptr[b] = -1; continue; }
                    t->mincode[b] = code; t->valptr[b] = idx;
                    code += cnt; idx += cnt;
                    t->maxcode[b] = code - 1;
                }
                k += 17 + (size_t)total;
            }
        } else if (marker == 0xdb) {
            size_t k = 0;
            while (k < sn && !d->err) {
                const int pq = s[k] >> 4, tq = s[k] & 15;
                if (tq > 3 || pq > 1) { d_fail(d, -1); break; }
                const size_t need = pq ? 128 : 64;
                if (k + 1 + need > sn) { d_fail(d, -1); break; }
                for (int z = 0; z < 64; z++) d->quant[tq][z] = pq ? (uint16_t)be16(s + k + 1 + 2 * z) : s[k + 1 + z];
                k += 1 + need;
            }
        } else if (marker == 0xdd) {
            if (sn != 2) { d_fail(d, -1); break; }
            d->ri = (int)be16(s);
        } else if (marker == 0xe0) {
            if (sn >= 5 && !memcmp(s, "JFIF\0", 5)) d->jfif = 1;
        } else if (marker == 0xee) {
            if (sn >= 12 && !memcmp(s, "Adobe", 5)) { d->adobe_valid = 1; d->adobe_transform = s[11]; }
        } else if (marker == 0xda) {
            d_sos(d, s, sn);
        } else if ((marker >= 0xe0 && marker <= 0xef) || marker == 0xfe) {
            /* APPn / COM: ignored */
        } else if (marker < 0xc0) { d_fail(d, -1); }                /* "unknown marker" */
        else d_fail(d, -2);                                        /* UnsupportedError("unknown marker"): arithmetic coding, lossless, DNL ... */
    }
    if (!d->err && !seen_eoi) d_fail(d, -1);
    if (!d->err && !d->have_img) d_fail(d, -1);                     /* "missing SOS marker" */
    if (!d->err && d->ncomp == 3 && !d->jfif &&
        ((d->adobe_valid && d->adobe_transform == 0) || (d->cid[0] == 'R' && d->cid[1] == 'G' && d->cid[2] == 'B')))
        d_fail(d, -2);                                             /* isRGB: convertToRGB is outside this restatement */
    if (!d->err && d->progressive) {
        /* reconstructProgressiveImage: only blocks that hold image pixels; the rest of the MCU-padded planes stays zero */
        for (int c = 0; c < d->ncomp; c++) {
            if (!d->prog[c]) continue;
            const int v = 8 * d->cv[0] / d->cv[c], hh = 8 * d->ch[0] / d->ch[c], stride = d->mxx * d->ch[c];
            for (int by = 0; by * v < d->h; by++)
                for (int bx = 0; bx * hh < d->w; bx++) d_reconstruct(d, d->prog[c] + ((size_t)by * stride + bx) * 64, bx, by, c);
        }
    }
    d_free(d);
    if (d->err) { ipxo_decoded_free(out); return d->err; }
    return 0;
}

/* ipx_jpeg_oracle.c -- TEST INFRASTRUCTURE: scalar restatement of Go's image/jpeg encoder.
 *
 * The reference ends every operator with jpeg.Encode(buf, img, &jpeg.Options{Quality: 85})
 * (operations/resize.go:80, thumbnail.go:70, watermark.go:68,73,76; domain.DefaultJPEGQuality) on the
 * *image.RGBA its helpers produced.  The encoder is Go 1.24 stdlib (go.mod:3), whose source is not under
 * /root/reference and cannot be run here (no Go toolchain): PARITY UNPINNED against Go itself.  What this file
 * restates, function by function (image/jpeg/writer.go, image/jpeg/fdct.go, image/color/ycbcr.go):
 *   Encode            SOI, writeDQT (both tables, zig-zag order), writeSOF0 (Y 2x2, Cb / Cr 1x1; Gray: one
 *                     component), writeDHT (the four Annex K tables; two for Gray), writeSOS, EOI
 *   writeSOS          16x16 MCUs in raster order: four Y blocks, then scale() of the Cb and Cr blocks
 *   rgbaToYCbCr       edge pixels replicated (sx = min(x, xmax)); color.RGBToYCbCr on the stored (premultiplied)
 *                     R, G, B bytes; alpha ignored
 *   scale             2x2 box, (sum + 2) >> 2
 *   fdct              libjpeg's jfdctint (slow-but-accurate integer), level shift inside, output scaled by 8
 *   writeBlock        div(b[unzig[zig]], 8*quant[zig]) rounding half away from zero; DC delta; AC run lengths
 *   emitHuffRLE/emit  magnitude category + bits; 0xff -> 0xff 0x00 stuffing; final pad emit(0x7f, 7)
 * The Gray path (one component, 8x8 MCUs) is restated too, although the reference never encodes a Gray image:
 * it is the configuration on which libjpeg (through Pillow) produces the SAME scan bytes, which pins fdct,
 * the quantiser, the Huffman tables and the bit packer against an independent implementation (tests/test_jpeg.py).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- tables ---------------------------------------------------------------------------------- */
static const uint8_t k_zigzag[64] = { /* natural index of the zig-th coefficient (Go: unzig) */
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

/* ITU-T T.81 Annex K.1, natural order; Go stores them already zig-zagged (unscaledQuant) */
static const uint8_t k_quant_natural[2][64] = {
    {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
     18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99},
    {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
     99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99}};

/* Annex K.3: theHuffmanSpec = {luminance DC, luminance AC, chrominance DC, chrominance AC} */
typedef struct { uint8_t count[16]; const uint8_t *value; int nvalue; } huff_spec;
static const uint8_t k_dc_values[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t k_ac_lum_values[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91,
    0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a,
    0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53,
    0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79,
    0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5,
    0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9,
    0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2,
    0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
static const uint8_t k_ac_chr_values[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14,
    0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17,
    0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a,
    0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78,
    0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7,
    0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2,
    0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
static const huff_spec k_spec[4] = {
    {{0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0}, k_dc_values, 12},
    {{0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 125}, k_ac_lum_values, 162},
    {{0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0}, k_dc_values, 12},
    {{0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 119}, k_ac_chr_values, 162}};

/* ---- encoder state ------------------------------------------------------------------------------ */
typedef struct {
    uint8_t *buf; size_t len, cap; int err;
    uint32_t bits, nbits;
    uint8_t quant[2][64];          /* zig-zag order */
    uint32_t lut[4][256];          /* huffmanLUT: code | length << 24 */
} enc;

static void put(enc *e, uint8_t b)
{
    if (e->len == e->cap) {
        size_t nc = e->cap ? e->cap * 2 : 4096;
        uint8_t *nb = (uint8_t *)realloc(e->buf, nc);
        if (!nb) { e->err = 1; return; }
        e->buf = nb; e->cap = nc;
    }
    e->buf[e->len++] = b;
}
static void put_n(enc *e, const uint8_t *p, int n) { for (int i = 0; i < n; i++) put(e, p[i]); }

/* (e *encoder) emit */
static void emit(enc *e, uint32_t bits, uint32_t nbits)
{
    nbits += e->nbits;
    bits <<= 32 - nbits;
    bits |= e->bits;
    while (nbits >= 8) {
        uint8_t b = (uint8_t)(bits >> 24);
        put(e, b);
        if (b == 0xff) put(e, 0x00);
        bits <<= 8;
        nbits -= 8;
    }
    e->bits = bits; e->nbits = nbits;
}
static void emit_huff(enc *e, int h, int32_t value)
{
    uint32_t x = e->lut[h][value];
    emit(e, x & ((1u << 24) - 1), x >> 24);
}
static uint32_t bit_count(uint32_t a)  /* Go's bitCount table: bits needed for a, a < 256 */
{
    uint32_t n = 0;
    while (a) { n++; a >>= 1; }
    return n;
}
static void emit_huff_rle(enc *e, int h, int32_t run, int32_t value)
{
    int32_t a = value, b = value;
    if (a < 0) { a = -value; b = value - 1; }
    uint32_t nbits = a < 0x100 ? bit_count((uint32_t)a) : 8 + bit_count((uint32_t)a >> 8);
    emit_huff(e, h, run << 4 | (int32_t)nbits);
    if (nbits > 0) emit(e, (uint32_t)b & ((1u << nbits) - 1), nbits);
}

/* huffmanLUT.init */
static void lut_init(uint32_t *lut, const huff_spec *s)
{
    uint32_t code = 0;
    int k = 0;
    memset(lut, 0, 256 * sizeof(uint32_t));
    for (int i = 0; i < 16; i++) {
        uint32_t nbits = (uint32_t)(i + 1) << 24;
        for (int j = 0; j < s->count[i]; j++) { lut[s->value[k]] = nbits | code; code++; k++; }
        code <<= 1;
    }
}

static int32_t div_round(int32_t a, int32_t b)  /* writer.go div */
{
    if (a >= 0) return (a + (b >> 1)) / b;
    return -((-a + (b >> 1)) / b);
}

/* ---- fdct.go ------------------------------------------------------------------------------------ */
#define FIX_0_298631336 2446
#define FIX_0_390180644 3196
#define FIX_0_541196100 4433
#define FIX_0_765366865 6270
#define FIX_0_899976223 7373
#define FIX_1_175875602 9633
#define FIX_1_501321110 12299
#define FIX_1_847759065 15137
#define FIX_1_961570560 16069
#define FIX_2_053119869 16819
#define FIX_2_562915447 20995
#define FIX_3_072711026 25172
#define CONST_BITS 13
#define PASS1_BITS 2
#define CENTER 128

static void fdct(int32_t *b)
{
    for (int y = 0; y < 8; y++) {   /* pass 1: rows */
        int32_t *s = b + 8 * y;
        int32_t x0 = s[0], x1 = s[1], x2 = s[2], x3 = s[3], x4 = s[4], x5 = s[5], x6 = s[6], x7 = s[7];
        int32_t tmp0 = x0 + x7, tmp1 = x1 + x6, tmp2 = x2 + x5, tmp3 = x3 + x4;
        int32_t tmp10 = tmp0 + tmp3, tmp12 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp13 = tmp1 - tmp2;
        tmp0 = x0 - x7; tmp1 = x1 - x6; tmp2 = x2 - x5; tmp3 = x3 - x4;
        s[0] = (tmp10 + tmp11 - 8 * CENTER) << PASS1_BITS;
        s[4] = (tmp10 - tmp11) << PASS1_BITS;
        int32_t z1 = (tmp12 + tmp13) * FIX_0_541196100;
        z1 += 1 << (CONST_BITS - PASS1_BITS - 1);
        s[2] = (z1 + tmp12 * FIX_0_765366865) >> (CONST_BITS - PASS1_BITS);
        s[6] = (z1 - tmp13 * FIX_1_847759065) >> (CONST_BITS - PASS1_BITS);
        tmp10 = tmp0 + tmp3; tmp11 = tmp1 + tmp2; tmp12 = tmp0 + tmp2; tmp13 = tmp1 + tmp3;
        z1 = (tmp12 + tmp13) * FIX_1_175875602;
        z1 += 1 << (CONST_BITS - PASS1_BITS - 1);
        tmp0 *= FIX_1_501321110; tmp1 *= FIX_3_072711026; tmp2 *= FIX_2_053119869; tmp3 *= FIX_0_298631336;
        tmp10 *= -FIX_0_899976223; tmp11 *= -FIX_2_562915447; tmp12 *= -FIX_0_390180644; tmp13 *= -FIX_1_961570560;
        tmp12 += z1; tmp13 += z1;
        s[1] = (tmp0 + tmp10 + tmp12) >> (CONST_BITS - PASS1_BITS);
        s[3] = (tmp1 + tmp11 + tmp13) >> (CONST_BITS - PASS1_BITS);
        s[5] = (tmp2 + tmp11 + tmp12) >> (CONST_BITS - PASS1_BITS);
        s[7] = (tmp3 + tmp10 + tmp13) >> (CONST_BITS - PASS1_BITS);
    }
    for (int x = 0; x < 8; x++) {   /* pass 2: columns; PASS1_BITS removed, the factor 8 stays */
        int32_t *s = b + x;
        int32_t tmp0 = s[0] + s[56], tmp1 = s[8] + s[48], tmp2 = s[16] + s[40], tmp3 = s[24] + s[32];
        int32_t tmp10 = tmp0 + tmp3 + (1 << (PASS1_BITS - 1)), tmp12 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp13 = tmp1 - tmp2;
        tmp0 = s[0] - s[56]; tmp1 = s[8] - s[48]; tmp2 = s[16] - s[40]; tmp3 = s[24] - s[32];
        s[0] = (tmp10 + tmp11) >> PASS1_BITS;
        s[32] = (tmp10 - tmp11) >> PASS1_BITS;
        int32_t z1 = (tmp12 + tmp13) * FIX_0_541196100;
        z1 += 1 << (CONST_BITS + PASS1_BITS - 1);
        s[16] = (z1 + tmp12 * FIX_0_765366865) >> (CONST_BITS + PASS1_BITS);
        s[48] = (z1 - tmp13 * FIX_1_847759065) >> (CONST_BITS + PASS1_BITS);
        tmp10 = tmp0 + tmp3; tmp11 = tmp1 + tmp2; tmp12 = tmp0 + tmp2; tmp13 = tmp1 + tmp3;
        z1 = (tmp12 + tmp13) * FIX_1_175875602;
        z1 += 1 << (CONST_BITS + PASS1_BITS - 1);
        tmp0 *= FIX_1_501321110; tmp1 *= FIX_3_072711026; tmp2 *= FIX_2_053119869; tmp3 *= FIX_0_298631336;
        tmp10 *= -FIX_0_899976223; tmp11 *= -FIX_2_562915447; tmp12 *= -FIX_0_390180644; tmp13 *= -FIX_1_961570560;
        tmp12 += z1; tmp13 += z1;
        s[8] = (tmp0 + tmp10 + tmp12) >> (CONST_BITS + PASS1_BITS);
        s[24] = (tmp1 + tmp11 + tmp13) >> (CONST_BITS + PASS1_BITS);
        s[40] = (tmp2 + tmp11 + tmp12) >> (CONST_BITS + PASS1_BITS);
        s[56] = (tmp3 + tmp10 + tmp13) >> (CONST_BITS + PASS1_BITS);
    }
}

/* (e *encoder) writeBlock; coefs (may be NULL) receives the 64 quantised values in zig-zag order */
static int32_t write_block(enc *e, int32_t *b, int q, int32_t prev_dc, int16_t *coefs)
{
    fdct(b);
    int32_t dc = div_round(b[0], 8 * (int32_t)e->quant[q][0]);
    if (coefs) coefs[0] = (int16_t)dc;
    emit_huff_rle(e, 2 * q + 0, 0, dc - prev_dc);
    int h = 2 * q + 1;
    int32_t run = 0;
    for (int zig = 1; zig < 64; zig++) {
        int32_t ac = div_round(b[k_zigzag[zig]], 8 * (int32_t)e->quant[q][zig]);
        if (coefs) coefs[zig] = (int16_t)ac;
        if (ac == 0) run++;
        else {
            while (run > 15) { emit_huff(e, h, 0xf0); run -= 16; }
            emit_huff_rle(e, h, run, ac);
            run = 0;
        }
    }
    if (run > 0) emit_huff(e, h, 0x00);
    return dc;
}

/* color.RGBToYCbCr */
static void rgb_to_ycbcr(uint8_t r, uint8_t g, uint8_t b, int32_t *yy, int32_t *cb, int32_t *cr)
{
    int32_t r1 = r, g1 = g, b1 = b;
    *yy = (19595 * r1 + 38470 * g1 + 7471 * b1 + (1 << 15)) >> 16;
    int32_t c = -11056 * r1 - 21712 * g1 + 32768 * b1 + (257 << 15);
    if (((uint32_t)c & 0xff000000u) == 0) c >>= 16; else c = ~(c >> 31);
    *cb = (uint8_t)c;
    c = 32768 * r1 - 27440 * g1 - 5328 * b1 + (257 << 15);
    if (((uint32_t)c & 0xff000000u) == 0) c >>= 16; else c = ~(c >> 31);
    *cr = (uint8_t)c;
}

/* scale: the 16x16 region held in four blocks -> one 8x8 block */
static void scale(int32_t *dst, int32_t src[4][64])
{
    for (int i = 0; i < 4; i++) {
        int off = (i & 2) << 4 | (i & 1) << 2;
        for (int y = 0; y < 4; y++)
            for (int x = 0; x < 4; x++) {
                int j = 16 * y + 2 * x;
                int32_t sum = src[i][j] + src[i][j + 1] + src[i][j + 8] + src[i][j + 9];
                dst[8 * y + x + off] = (sum + 2) >> 2;
            }
    }
}

static void header(enc *e, int w, int h, int ncomp)
{
    static const uint8_t soi[2] = {0xff, 0xd8};
    put_n(e, soi, 2);
    /* writeDQT: both tables, always */
    { const uint8_t m[4] = {0xff, 0xdb, 0x00, 2 + 2 * 65}; put_n(e, m, 4); }
    for (int i = 0; i < 2; i++) { put(e, (uint8_t)i); put_n(e, e->quant[i], 64); }
    /* writeSOF0 */
    { int len = 8 + 3 * ncomp;
      const uint8_t m[10] = {0xff, 0xc0, (uint8_t)(len >> 8), (uint8_t)len, 8, (uint8_t)(h >> 8), (uint8_t)h, (uint8_t)(w >> 8), (uint8_t)w, (uint8_t)ncomp};
      put_n(e, m, 10); }
    if (ncomp == 1) { const uint8_t c[3] = {1, 0x11, 0x00}; put_n(e, c, 3); }
    else { const uint8_t c[9] = {1, 0x22, 0x00, 2, 0x11, 0x01, 3, 0x11, 0x01}; put_n(e, c, 9); }
    /* writeDHT */
    { int nspec = ncomp == 1 ? 2 : 4, len = 2;
      for (int i = 0; i < nspec; i++) len += 1 + 16 + k_spec[i].nvalue;
      const uint8_t m[4] = {0xff, 0xc4, (uint8_t)(len >> 8), (uint8_t)len};
      put_n(e, m, 4);
      static const uint8_t tc_th[4] = {0x00, 0x10, 0x01, 0x11};
      for (int i = 0; i < nspec; i++) { put(e, tc_th[i]); put_n(e, k_spec[i].count, 16); put_n(e, k_spec[i].value, k_spec[i].nvalue); } }
    /* SOS header */
    if (ncomp == 1) { const uint8_t s[10] = {0xff, 0xda, 0x00, 0x08, 0x01, 0x01, 0x00, 0x00, 0x3f, 0x00}; put_n(e, s, 10); }
    else { const uint8_t s[14] = {0xff, 0xda, 0x00, 0x0c, 0x03, 0x01, 0x00, 0x02, 0x11, 0x03, 0x11, 0x00, 0x3f, 0x00}; put_n(e, s, 14); }
}

static int enc_init(enc *e, int quality)
{
    memset(e, 0, sizeof *e);
    if (quality < 1) quality = 1; else if (quality > 100) quality = 100;
    int scale_ = quality < 50 ? 5000 / quality : 200 - quality * 2;
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 64; j++) {
            int x = k_quant_natural[i][k_zigzag[j]];
            x = (x * scale_ + 50) / 100;
            if (x < 1) x = 1; else if (x > 255) x = 255;
            e->quant[i][j] = (uint8_t)x;
        }
    for (int i = 0; i < 4; i++) lut_init(e->lut[i], &k_spec[i]);
    return 0;
}

/* quantisation tables as DQT carries them (zig-zag order) */
void ipxo_jpeg_quant(int quality, uint8_t out[2][64])
{
    enc e;
    enc_init(&e, quality);
    memcpy(out, e.quant, 128);
}

/* jpeg.Encode(w, *image.RGBA, &jpeg.Options{Quality: quality}).  Returns 0 and a malloc'd buffer; coefs (may be NULL)
 * receives every block's quantised coefficients in scan order (6 x 64 int16 per MCU: Y0 Y1 Y2 Y3 Cb Cr). */
int ipxo_jpeg_encode_rgba8(const uint8_t *pix, int w, int h, int stride, int quality, uint8_t **out, size_t *out_len, int16_t *coefs)
{
    enc e;
    if (w <= 0 || h <= 0 || w >= 1 << 16 || h >= 1 << 16) return -1;   /* "jpeg: image is too large to encode" */
    enc_init(&e, quality);
    header(&e, w, h, 3);
    int32_t prev_y = 0, prev_cb = 0, prev_cr = 0;
    int32_t b[64], cb[4][64], cr[4][64];
    for (int y = 0; y < h; y += 16)
        for (int x = 0; x < w; x += 16) {
            for (int i = 0; i < 4; i++) {
                int px = x + (i & 1) * 8, py = y + (i & 2) * 4;
                for (int j = 0; j < 8; j++) {   /* rgbaToYCbCr */
                    int sj = py + j > h - 1 ? h - 1 : py + j;
                    for (int k = 0; k < 8; k++) {
                        int sx = px + k > w - 1 ? w - 1 : px + k;
                        const uint8_t *p = pix + (size_t)sj * stride + (size_t)sx * 4;
                        rgb_to_ycbcr(p[0], p[1], p[2], &b[8 * j + k], &cb[i][8 * j + k], &cr[i][8 * j + k]);
                    }
                }
                prev_y = write_block(&e, b, 0, prev_y, coefs);
                if (coefs) coefs += 64;
            }
            scale(b, cb);
            prev_cb = write_block(&e, b, 1, prev_cb, coefs);
            if (coefs) coefs += 64;
            scale(b, cr);
            prev_cr = write_block(&e, b, 1, prev_cr, coefs);
            if (coefs) coefs += 64;
        }
    emit(&e, 0x7f, 7);
    put(&e, 0xff); put(&e, 0xd9);
    if (e.err) { free(e.buf); return -2; }
    *out = e.buf; *out_len = e.len;
    return 0;
}

/* jpeg.Encode on an *image.Gray (nComponent 1, 8x8 MCUs, grayToY with replicated edges) */
int ipxo_jpeg_encode_gray8(const uint8_t *pix, int w, int h, int stride, int quality, uint8_t **out, size_t *out_len)
{
    enc e;
    if (w <= 0 || h <= 0 || w >= 1 << 16 || h >= 1 << 16) return -1;
    enc_init(&e, quality);
    header(&e, w, h, 1);
    int32_t prev = 0, b[64];
    for (int y = 0; y < h; y += 8)
        for (int x = 0; x < w; x += 8) {
            for (int j = 0; j < 8; j++) {
                int sj = y + j > h - 1 ? h - 1 : y + j;
                for (int k = 0; k < 8; k++) {
                    int sx = x + k > w - 1 ? w - 1 : x + k;
                    b[8 * j + k] = pix[(size_t)sj * stride + sx];
                }
            }
            prev = write_block(&e, b, 0, prev, NULL);
        }
    emit(&e, 0x7f, 7);
    put(&e, 0xff); put(&e, 0xd9);
    if (e.err) { free(e.buf); return -2; }
    *out = e.buf; *out_len = e.len;
    return 0;
}

void ipxo_free(void *p) { free(p); }

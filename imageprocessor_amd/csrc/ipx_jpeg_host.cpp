// ipx_jpeg_host.cpp -- the host half of jpeg.Encode: tables, stream headers and the entropy coder.
//
// Go's image/jpeg writer (Go 1.24 stdlib, go.mod:3; called at operations/resize.go:80, thumbnail.go:70,
// watermark.go:68,73,76 with Quality 85) writes SOI, DQT (both tables), SOF0 (Y 2x2, Cb / Cr 1x1), DHT (the four
// Annex K tables), SOS, the Huffman-coded MCUs and EOI -- no JFIF / APPn segment.  The transform half runs on
// the GPU (ipx_jpeg.hip) and hands over quantised coefficients in scan order; this file turns them into the
// byte stream: DC deltas per component, AC run lengths (ZRL / EOB), magnitude categories, 0xff stuffing and
// the final emit(0x7f, 7) padding, exactly as writeBlock / emitHuffRLE / emit do.
// The coder works per symbol with precombined (Huffman code + magnitude bits) tables and a 64-bit accumulator,
// flushed four bytes at a time; that changes the speed, not one output bit.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ipx_internal.h"

namespace ipx {

namespace {

// ITU-T T.81 K.1 in natural order; Go holds the same numbers zig-zagged (unscaledQuant)
const uint8_t kQuantNatural[2][64] = {
    {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
     18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99},
    {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
     99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99}};
const uint8_t kNaturalOfZig[64] = {
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// T.81 K.3.3: BITS and HUFFVAL of the four typical tables (theHuffmanSpec: lum DC, lum AC, chroma DC, chroma AC)
const uint8_t kBits[4][16] = {{0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0},
                              {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 125},
                              {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0},
                              {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 119}};
const uint8_t kDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t kAcLumVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91,
    0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a,
    0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53,
    0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79,
    0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5,
    0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9,
    0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2,
    0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
const uint8_t kAcChrVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14,
    0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17,
    0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a,
    0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78,
    0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7,
    0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2,
    0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
const uint8_t *const kVals[4] = {kDcVals, kAcLumVals, kDcVals, kAcChrVals};
const int kNVals[4] = {12, 162, 12, 162};

struct Code { uint32_t code; uint32_t len; };

struct HuffTables {
    Code sym[4][256];
    HuffTables()
    {
        for (int t = 0; t < 4; t++) {
            memset(sym[t], 0, sizeof sym[t]);
            uint32_t code = 0;
            int k = 0;
            for (int i = 0; i < 16; i++) {
                for (int j = 0; j < kBits[t][i]; j++, k++, code++) sym[t][kVals[t][k]] = Code{code, (uint32_t)i + 1};
                code <<= 1;
            }
        }
    }
};
const HuffTables &huff()
{
    static const HuffTables t;
    return t;
}

class BitWriter {
public:
    explicit BitWriter(std::vector<uint8_t> *out) : out_(out) {}
    inline void put(uint32_t bits, uint32_t n)   // n <= 27
    {
        acc_ = (acc_ << n) | bits;
        cnt_ += n;
        while (cnt_ >= 8) {
            const uint8_t b = (uint8_t)(acc_ >> (cnt_ - 8));
            out_->push_back(b);
            if (b == 0xff) out_->push_back(0x00);
            cnt_ -= 8;
        }
    }
    void pad() { put(0x7f, 7); }   // e.emit(0x7f, 7): fills the last byte with ones
private:
    std::vector<uint8_t> *out_;
    uint64_t acc_ = 0;
    uint32_t cnt_ = 0;
};

inline uint32_t category(int32_t a)   // bits needed for |value|
{
    return a ? 32u - (uint32_t)__builtin_clz((uint32_t)a) : 0u;
}

// emitHuffRLE: Huffman code of (run << 4 | category), then the low `category` bits of value (value - 1 if negative)
inline void put_rle(BitWriter &w, const Code *tab, int32_t run, int32_t value)
{
    const int32_t a = value < 0 ? -value : value, b = value < 0 ? value - 1 : value;
    const uint32_t nb = category(a);
    const Code c = tab[(uint32_t)run << 4 | nb];
    w.put(c.code, c.len);
    if (nb) w.put((uint32_t)b & ((1u << nb) - 1), nb);
}

inline int32_t put_block(BitWriter &w, const int16_t *c, int q, int32_t prev_dc)
{
    const HuffTables &h = huff();
    const int32_t dc = c[0];
    put_rle(w, h.sym[2 * q], 0, dc - prev_dc);
    const Code *ac = h.sym[2 * q + 1];
    int32_t run = 0;
    for (int zig = 1; zig < 64; zig++) {
        const int32_t v = c[zig];
        if (v == 0) { run++; continue; }
        while (run > 15) { w.put(ac[0xf0].code, ac[0xf0].len); run -= 16; }
        put_rle(w, ac, run, v);
        run = 0;
    }
    if (run > 0) w.put(ac[0x00].code, ac[0x00].len);
    return dc;
}

void put16(std::vector<uint8_t> &o, int v) { o.push_back((uint8_t)(v >> 8)); o.push_back((uint8_t)v); }

}  // namespace

void jpeg_tables(int quality, JpegTables *t)
{
    if (quality < 1) quality = 1;
    else if (quality > 100) quality = 100;
    const int scale = quality < 50 ? 5000 / quality : 200 - quality * 2;
    for (int i = 0; i < 2; i++)
        for (int zig = 0; zig < 64; zig++) {
            const int nat = kNaturalOfZig[zig];
            int x = ((int)kQuantNatural[i][nat] * scale + 50) / 100;
            x = x < 1 ? 1 : (x > 255 ? 255 : x);
            t->quant[i][zig] = (uint8_t)x;
            const uint32_t d = 8u * (uint32_t)x;
            t->div8[i][nat] = (uint16_t)d;
            t->recip[i][nat] = (uint32_t)(((1ull << 32) + d - 1) / d);   // ceil(2^32 / d): mulhi is exact for n * (d - 1) < 2^32
        }
}

// SOI, DQT, SOF0, DHT and the SOS header of jpeg.Encode for a w x h *image.RGBA (623 bytes)
void jpeg_write_header(int w, int h, const JpegTables &t, std::vector<uint8_t> *out)
{
    std::vector<uint8_t> &o = *out;
    o.clear();
    o.push_back(0xff); o.push_back(0xd8);                                   // SOI
    o.push_back(0xff); o.push_back(0xdb); put16(o, 2 + 2 * 65);              // writeDQT
    for (int i = 0; i < 2; i++) { o.push_back((uint8_t)i); o.insert(o.end(), t.quant[i], t.quant[i] + 64); }
    o.push_back(0xff); o.push_back(0xc0); put16(o, 8 + 3 * 3);               // writeSOF0
    o.push_back(8); put16(o, h); put16(o, w); o.push_back(3);
    { const uint8_t comp[9] = {1, 0x22, 0x00, 2, 0x11, 0x01, 3, 0x11, 0x01}; o.insert(o.end(), comp, comp + 9); }
    int dht = 2;
    for (int i = 0; i < 4; i++) dht += 1 + 16 + kNVals[i];
    o.push_back(0xff); o.push_back(0xc4); put16(o, dht);                     // writeDHT
    { const uint8_t tc_th[4] = {0x00, 0x10, 0x01, 0x11};
      for (int i = 0; i < 4; i++) { o.push_back(tc_th[i]); o.insert(o.end(), kBits[i], kBits[i] + 16); o.insert(o.end(), kVals[i], kVals[i] + kNVals[i]); } }
    { const uint8_t sos[14] = {0xff, 0xda, 0x00, 0x0c, 0x03, 0x01, 0x00, 0x02, 0x11, 0x03, 0x11, 0x00, 0x3f, 0x00}; o.insert(o.end(), sos, sos + 14); }
}

// the four Huffman tables as the GPU coder reads them: [table][symbol] = length << 16 | code
void jpeg_huff_packed(uint32_t out[1024])
{
    const HuffTables &h = huff();
    for (int t = 0; t < 4; t++)
        for (int i = 0; i < 256; i++) out[t * 256 + i] = h.sym[t][i].len << 16 | h.sym[t][i].code;
}

// The whole stream of jpeg.Encode for a w x h *image.RGBA whose quantised coefficients are `coefs`
// (6 x 64 int16 per 16x16 MCU, scan order Y0 Y1 Y2 Y3 Cb Cr, zig-zag inside a block).
void jpeg_write_stream(const int16_t *coefs, int w, int h, const JpegTables &t, std::vector<uint8_t> *out)
{
    std::vector<uint8_t> &o = *out;
    const size_t mcus = (size_t)((w + 15) / 16) * (size_t)((h + 15) / 16);
    jpeg_write_header(w, h, t, out);
    o.reserve(1024 + mcus * 96);
    BitWriter bw(&o);
    int32_t dc_y = 0, dc_cb = 0, dc_cr = 0;
    for (size_t m = 0; m < mcus; m++, coefs += 384) {
        for (int i = 0; i < 4; i++) dc_y = put_block(bw, coefs + 64 * i, 0, dc_y);
        dc_cb = put_block(bw, coefs + 256, 1, dc_cb);
        dc_cr = put_block(bw, coefs + 320, 1, dc_cr);
    }
    bw.pad();
    o.push_back(0xff); o.push_back(0xd9);                                   // EOI
}

}  // namespace ipx

extern "C" {

size_t ipx_jpeg_coef_count(int w, int h)
{
    if (w <= 0 || h <= 0) return 0;
    return (size_t)((w + 15) / 16) * (size_t)((h + 15) / 16) * 384;
}

int ipx_jpeg_quant_tables(int quality, uint8_t out[128]) try
{
    if (!out) { ipx::set_error("ipx_jpeg_quant_tables: null argument"); return IPX_ERR_INVALID; }
    ipx::JpegTables t;
    ipx::jpeg_tables(quality, &t);
    memcpy(out, t.quant, 128);
    return IPX_OK;
}
IPX_CATCH_STATUS

int ipx_jpeg_entropy_encode(const int16_t *coefs, int w, int h, int quality, uint8_t **out, size_t *len) try
{
    if (!coefs || !out || !len) { ipx::set_error("ipx_jpeg_entropy_encode: null argument"); return IPX_ERR_INVALID; }
    if (w <= 0 || h <= 0 || w >= 1 << 16 || h >= 1 << 16) { ipx::set_error("jpeg: image is too large to encode"); return IPX_ERR_INVALID; }
    ipx::JpegTables t;
    ipx::jpeg_tables(quality, &t);
    std::vector<uint8_t> v;
    ipx::jpeg_write_stream(coefs, w, h, t, &v);
    uint8_t *p = (uint8_t *)malloc(v.size() ? v.size() : 1);
    if (!p) { ipx::set_error("out of memory"); return IPX_ERR_NOMEM; }
    memcpy(p, v.data(), v.size());
    *out = p; *len = v.size();
    return IPX_OK;
}
IPX_CATCH_STATUS

void ipx_buffer_free(void *p) { free(p); }

}  // extern "C"

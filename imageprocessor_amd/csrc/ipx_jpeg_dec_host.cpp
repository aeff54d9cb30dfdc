// ipx_jpeg_dec_host.cpp -- the host part of image.Decode for JPEGs: marker parsing and table preparation.
//
// The reference decodes every upload with image.Decode (image_processor.go:47); for JPEG files that is Go's image/jpeg
// (reader.go: decode, processSOF, processDHT, processDQT, processDRI, processSOS; Go 1.24 stdlib, go.mod:3).  The
// entropy decoding and the inverse transform run on the GPU (ipx_jpeg_dec.hip); what stays here is what is serial and
// tiny: walk the markers, collect the tables the single scan refers to, and lay them out the way the kernels read them
// (an 8-bit first-level Huffman table plus the canonical mincode / maxcode arrays for longer codes; quantisers
// de-zig-zagged).  Anything outside the baseline subset the kernels implement is reported as IPX_ERR_UNSUPPORTED for
// that image, so that the worker keeps Go's CPU path for it: progressive (SOF2), CMYK / RGB (Adobe) files,
// 4:1:1 / 4:1:0 and other sampling factors, several scans, 12-bit samples, Huffman table ids above 1.
#include <cstring>

#include "ipx_internal.h"

namespace ipx {

namespace {

const uint8_t kUnzig[64] = {
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct RawHuff { bool ok = false; uint8_t counts[16]; uint8_t vals[256]; int nvals = 0; };

inline uint32_t be16(const uint8_t *p) { return (uint32_t)p[0] << 8 | p[1]; }

// huffman.go's table, in the two forms the kernel uses
bool build_huff(const RawHuff &r, uint16_t lut[256], int32_t maxcode[18], int32_t valoff[18], uint8_t vals[256], uint32_t bound[8])
{
    memset(lut, 0, 256 * sizeof(uint16_t));
    memcpy(vals, r.vals, 256);
    int32_t code = 0, idx = 0;
    for (int len = 1; len <= 16; len++) {
        code <<= 1;
        const int cnt = r.counts[len - 1];
        if (cnt == 0) { maxcode[len] = -1; valoff[len] = 0; if (len >= 9) bound[len - 9] = (uint32_t)code << (16 - len); continue; }
        if (code + cnt > (1 << len)) return false;            // "bad Huffman table": more codes than the length can hold
        valoff[len] = idx - code;                              // symbol index = valoff + code
        if (len <= 8)
            for (int k = 0; k < cnt; k++) {
                const int c = code + k, sym = r.vals[idx + k];
                for (int fill = 0; fill < (1 << (8 - len)); fill++) lut[(c << (8 - len)) | fill] = (uint16_t)(len << 8 | sym);
            }
        code += cnt; idx += cnt;
        maxcode[len] = code - 1;
        if (len >= 9) bound[len - 9] = (uint32_t)code << (16 - len);   // canonical codes: a 16-bit window below this has a code of at most len bits
    }
    maxcode[0] = -1; valoff[0] = 0; maxcode[17] = 0x7fffffff; valoff[17] = 0;
    return true;
}

}  // namespace

// 0 ok; IPX_ERR_INVALID malformed; IPX_ERR_UNSUPPORTED a valid file outside the GPU subset
int jpeg_parse(const uint8_t *d, size_t len, JpegDecInfo *info, JpegDecTables *tab)
{
    memset(info, 0, sizeof *info);
    if (len < 4 || d[0] != 0xff || d[1] != 0xd8) return IPX_ERR_INVALID;
    uint16_t quant[4][64];
    bool have_q[4] = {false, false, false, false};
    RawHuff hf[2][4];
    int ncomp = 0, ch[3] = {0, 0, 0}, cv[3] = {0, 0, 0}, ctq[3] = {0, 0, 0}, cid[3] = {0, 0, 0};
    bool jfif = false, adobe = false;
    int adobe_transform = 0;
    size_t i = 2;
    for (;;) {
        if (i + 2 > len || d[i] != 0xff) return IPX_ERR_INVALID;
        while (i + 1 < len && d[i + 1] == 0xff) i++;
        if (i + 2 > len) return IPX_ERR_INVALID;
        const int m = d[i + 1];
        i += 2;
        if (m == 0xd9) return IPX_ERR_INVALID;                 // "missing SOS marker"
        if (m == 0x00 || (m >= 0xd0 && m <= 0xd7)) continue;
        if (i + 2 > len) return IPX_ERR_INVALID;
        const size_t n = be16(d + i);
        if (n < 2 || i + n > len) return IPX_ERR_INVALID;
        const uint8_t *s = d + i + 2;
        const size_t sn = n - 2;
        switch (m) {
        case 0xc0: case 0xc1: {
            if (ncomp || sn < 6) return IPX_ERR_INVALID;
            if (s[0] != 8) return IPX_ERR_UNSUPPORTED;
            info->h = (int)be16(s + 1); info->w = (int)be16(s + 3);
            ncomp = s[5];
            if (ncomp == 4) return IPX_ERR_UNSUPPORTED;
            if ((ncomp != 3 && ncomp != 1) || sn != (size_t)(6 + 3 * ncomp) || info->w <= 0 || info->h <= 0) return IPX_ERR_INVALID;
            if ((long long)info->w * info->h > (1LL << 28)) return IPX_ERR_UNSUPPORTED;   // 268 Mpixel: beyond any batch this path is meant for
            for (int c = 0; c < ncomp; c++) {
                cid[c] = s[6 + 3 * c]; ch[c] = s[7 + 3 * c] >> 4; cv[c] = s[7 + 3 * c] & 15; ctq[c] = s[8 + 3 * c];
                if (ctq[c] > 3 || ch[c] < 1 || ch[c] > 4 || cv[c] < 1 || cv[c] > 4) return IPX_ERR_INVALID;
                for (int j = 0; j < c; j++) if (cid[j] == cid[c]) return IPX_ERR_INVALID;   // "repeated component identifier"
            }
            if (ncomp == 1) { ch[0] = cv[0] = 1; break; }   // processSOF: a single component is non-interleaved, its (h, v) is effectively (1, 1)
            if (ch[1] != 1 || cv[1] != 1 || ch[2] != 1 || cv[2] != 1 || ch[0] > 2 || cv[0] > 2) return IPX_ERR_UNSUPPORTED;
            break;
        }
        case 0xc2: return IPX_ERR_UNSUPPORTED;
        case 0xc4: {
            size_t k = 0;
            while (k < sn) {
                if (k + 17 > sn) return IPX_ERR_INVALID;
                const int tc = s[k] >> 4, th = s[k] & 15;
                if (tc > 1 || th > 3) return IPX_ERR_INVALID;
                RawHuff &t = hf[tc][th];
                int total = 0;
                for (int b = 0; b < 16; b++) { t.counts[b] = s[k + 1 + b]; total += t.counts[b]; }
                if (total == 0 || total > 256 || k + 17 + (size_t)total > sn) return IPX_ERR_INVALID;
                memset(t.vals, 0, sizeof t.vals);
                memcpy(t.vals, s + k + 17, (size_t)total);
                t.nvals = total; t.ok = true;
                k += 17 + (size_t)total;
            }
            break;
        }
        case 0xdb: {
            size_t k = 0;
            while (k < sn) {
                const int pq = s[k] >> 4, tq = s[k] & 15;
                if (tq > 3 || pq > 1) return IPX_ERR_INVALID;
                const size_t need = pq ? 128 : 64;
                if (k + 1 + need > sn) return IPX_ERR_INVALID;
                for (int z = 0; z < 64; z++) quant[tq][z] = pq ? (uint16_t)be16(s + k + 1 + 2 * z) : s[k + 1 + z];
                have_q[tq] = true;
                k += 1 + need;
            }
            break;
        }
        case 0xdd:
            if (sn != 2) return IPX_ERR_INVALID;
            info->ri = (int)be16(s);
            break;
        case 0xe0: if (sn >= 5 && !memcmp(s, "JFIF\0", 5)) jfif = true; break;
        case 0xee: if (sn >= 12 && !memcmp(s, "Adobe", 5)) { adobe = true; adobe_transform = s[11]; } break;
        case 0xda: {
            if (!ncomp) return IPX_ERR_INVALID;
            if (sn < 1) return IPX_ERR_INVALID;
            if (s[0] != ncomp) return s[0] >= 1 && s[0] <= 3 ? IPX_ERR_UNSUPPORTED : IPX_ERR_INVALID;   // a frame coded in several scans
            if (sn != (size_t)(4 + 2 * ncomp)) return IPX_ERR_INVALID;
            int td[3] = {0, 0, 0}, ta[3] = {0, 0, 0};
            for (int c = 0; c < ncomp; c++) {
                if (s[1 + 2 * c] != cid[c]) return IPX_ERR_UNSUPPORTED;
                td[c] = s[2 + 2 * c] >> 4; ta[c] = s[2 + 2 * c] & 15;
                if (td[c] > 3 || ta[c] > 3) return IPX_ERR_INVALID;
                if (td[c] > 1 || ta[c] > 1) return IPX_ERR_UNSUPPORTED;
                if (!hf[0][td[c]].ok || !hf[1][ta[c]].ok || !have_q[ctq[c]]) return IPX_ERR_INVALID;
                info->td[c] = (uint8_t)td[c]; info->ta[c] = (uint8_t)(2 + ta[c]);   // kernel table slots: 0,1 = DC; 2,3 = AC
            }
            if (ncomp == 3 && !jfif && ((adobe && adobe_transform == 0) || (cid[0] == 'R' && cid[1] == 'G' && cid[2] == 'B'))) return IPX_ERR_UNSUPPORTED;
            info->h0 = ch[0]; info->v0 = cv[0];
            info->ncomp = ncomp;
            info->ratio = ncomp == 1 ? IPX_GRAY : ch[0] == 1 ? (cv[0] == 1 ? IPX_YCBCR_444 : IPX_YCBCR_440) : (cv[0] == 1 ? IPX_YCBCR_422 : IPX_YCBCR_420);
            info->scan_off = i + n;
            info->scan_len = len - (i + n);
            for (int tc = 0; tc < 2; tc++)
                for (int th = 0; th < 2; th++) {
                    const int slot = tc * 2 + th;
                    if (!hf[tc][th].ok) {
                        memset(tab->lut[slot], 0, sizeof tab->lut[slot]);
                        for (int l = 0; l < 18; l++) { tab->maxcode[slot][l] = -1; tab->valoff[slot][l] = 0; }
                        tab->maxcode[slot][17] = 0x7fffffff;
                        memset(tab->bound[slot], 0, sizeof tab->bound[slot]);
                        continue;
                    }
                    if (!build_huff(hf[tc][th], tab->lut[slot], tab->maxcode[slot], tab->valoff[slot], tab->vals[slot], tab->bound[slot])) return IPX_ERR_INVALID;
                }
            for (int c = 0; c < 3; c++)
                for (int zig = 0; zig < 64; zig++) tab->qnat[c][kUnzig[zig]] = c < ncomp ? quant[ctq[c]][zig] : 0;
            return IPX_OK;
        }
        default: break;   // APPn, COM, ...: skipped
        }
        i += n;
    }
}

}  // namespace ipx

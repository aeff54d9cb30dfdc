// ipx_jpeg_dec_host.cpp -- the host part of image.Decode for JPEGs: marker parsing and table preparation.
//
// The reference decodes every upload with image.Decode (image_processor.go:47); for JPEG files that is Go's image/jpeg
// (reader.go: decode, processSOF, processDHT, processDQT, processDRI, processSOS; Go 1.24 stdlib, go.mod:3).  The
// entropy decoding and the inverse transform run on the GPU (ipx_jpeg_dec.hip); what stays here is what is serial and
// tiny: walk the markers, collect the tables the single scan refers to, and lay them out the way the kernels read them
// (an 8-bit first-level Huffman table plus the canonical mincode / maxcode arrays for longer codes; quantisers
// de-zig-zagged).  Anything outside the baseline subset the kernels implement is reported as IPX_ERR_UNSUPPORTED for
// that image, so that the worker keeps Go's CPU path for it: CMYK / RGB (Adobe) files, 4:1:1 / 4:1:0 and other sampling factors,
// 12-bit samples.  Progressive files, files coded in several scans and Huffman table ids above 1 are marked host_scans: their
// scans are decoded on the host (ipx_jpeg_dec_prog.cpp) and only the transform onwards runs on the GPU.  A file has to end the way
// Go's marker loop wants it to (EOI after the last scan), or it is malformed here as it is there.
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
    memset(quant, 0, sizeof quant);                            // a table no DQT defined is all zero in Go's decoder, and the file decodes (to flat grey)
    RawHuff hf[2][4];
    int ncomp = 0, ch[3] = {0, 0, 0}, cv[3] = {0, 0, 0}, ctq[3] = {0, 0, 0}, cid[3] = {0, 0, 0};
    bool jfif = false, adobe = false;
    int adobe_transform = 0;
    size_t i = 2;
    for (;;) {
        if (i + 2 > len) return IPX_ERR_INVALID;
        if (d[i] != 0xff) { i++; continue; }                   // decode(): bytes that belong to no segment are skipped ("libjpeg is liberal")
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
        case 0xc0: case 0xc1: case 0xc2: {
            if (m == 0xc2) info->host_scans = info->progressive = 1;   // progressive: the scans refine each other (ipx_jpeg_dec_prog.cpp)
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
                // processSOF refuses a factor of 3 for every component, a single one included -- before it sets that one's (h, v) to (1, 1)
                // (found by tools/fuzz_corrupt.py: a Gray file whose V_1 a bit flip had made 3 decoded here and fails in Go)
                if (ch[c] == 3 || cv[c] == 3) return IPX_ERR_UNSUPPORTED;
            }
            if (ncomp == 1) { ch[0] = cv[0] = 1; break; }   // processSOF: a single component is non-interleaved, its (h, v) is effectively (1, 1)
            if (ch[1] != 1 || cv[1] != 1 || ch[2] != 1 || cv[2] != 1 || ch[0] > 2 || cv[0] > 2) return IPX_ERR_UNSUPPORTED;
            break;
        }
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
            info->h0 = ch[0]; info->v0 = cv[0];
            info->ncomp = ncomp;
            info->ratio = ncomp == 1 ? IPX_GRAY : ch[0] == 1 ? (cv[0] == 1 ? IPX_YCBCR_444 : IPX_YCBCR_440) : (cv[0] == 1 ? IPX_YCBCR_422 : IPX_YCBCR_420);
            if (info->host_scans) return IPX_OK;               // the frame is known; the scans are the host decoder's
            if (s[0] != ncomp) {                               // a sequential frame coded in several scans
                if (s[0] < 1 || s[0] > 3) return IPX_ERR_INVALID;
                info->host_scans = 1;
                return IPX_OK;
            }
            if (sn != (size_t)(4 + 2 * ncomp)) return IPX_ERR_INVALID;
            int td[3] = {0, 0, 0}, ta[3] = {0, 0, 0};
            for (int c = 0; c < ncomp; c++) {
                if (s[1 + 2 * c] != cid[c]) { info->host_scans = 1; return IPX_OK; }           // components out of frame order
                td[c] = s[2 + 2 * c] >> 4; ta[c] = s[2 + 2 * c] & 15;
                if (td[c] > 3 || ta[c] > 3) return IPX_ERR_INVALID;
                if (td[c] > 1 || ta[c] > 1) { info->host_scans = 1; return IPX_OK; }           // extended sequential: four tables per class
                if (!hf[0][td[c]].ok || !hf[1][ta[c]].ok) return IPX_ERR_INVALID;   // "uninitialized Huffman table"
                info->td[c] = (uint8_t)td[c]; info->ta[c] = (uint8_t)(2 + ta[c]);   // kernel table slots: 0,1 = DC; 2,3 = AC
                // processSOS runs ONE block loop for sequential and progressive scans: an AC symbol (r < 15, s = 0) starts an end-of-band
                // run in a baseline file too, and the next blocks of the scan lose their AC part.  No encoder writes such a symbol into a
                // sequential file, the GPU decoder reads it as a plain EOB -- so a (damaged) table that holds one goes to the host decoder.
                const RawHuff &act = hf[1][ta[c]];
                for (int v = 0; v < act.nvals; v++)
                    if ((act.vals[v] & 15) == 0 && (act.vals[v] >> 4) != 0 && (act.vals[v] >> 4) != 15) { info->host_scans = 1; return IPX_OK; }
            }
            if (ncomp == 3 && !jfif && ((adobe && adobe_transform == 0) || (cid[0] == 'R' && cid[1] == 'G' && cid[2] == 'B'))) return IPX_ERR_UNSUPPORTED;
            info->scan_off = i + n;
            {
                // The scan ends at the first marker that is not a restart marker; what follows has to be what Go's marker loop accepts
                // up to EOI (decode() keeps reading segments until it sees one: a file that stops after its scan is
                // io.ErrUnexpectedEOF there, and a task the reference marks failed).  Another SOS makes it a multi-scan file.
                const uint8_t *sd = d + info->scan_off;
                const size_t rest = len - info->scan_off;
                size_t k = 0, end = rest;
                while (k + 1 < rest) {
                    const uint8_t *q = (const uint8_t *)memchr(sd + k, 0xff, rest - 1 - k);
                    if (!q) break;
                    k = (size_t)(q - sd);
                    const uint8_t m2 = sd[k + 1];
                    if (m2 == 0x00 || (m2 >= 0xd0 && m2 <= 0xd7)) { k += 2; continue; }
                    end = k;
                    break;
                }
                info->scan_len = end;
                size_t p = info->scan_off + end;
                for (;;) {                                     // the tail: Go's marker loop, segments skipped by their length
                    if (p + 2 > len) return IPX_ERR_INVALID;   // no EOI
                    if (d[p] != 0xff) { p++; continue; }       // bytes that belong to no segment are skipped
                    const int m3 = d[p + 1];
                    if (m3 == 0xff) { p++; continue; }         // fill byte
                    p += 2;
                    if (m3 == 0xd9) break;                     // EOI
                    if (m3 == 0x00 || (m3 >= 0xd0 && m3 <= 0xd7)) continue;
                    if (m3 == 0xda) { info->host_scans = 1; return IPX_OK; }
                    if (m3 == 0xc0 || m3 == 0xc1 || m3 == 0xc2) return IPX_ERR_INVALID;        // "multiple SOF markers"
                    if (!((m3 >= 0xe0 && m3 <= 0xef) || m3 == 0xfe || m3 == 0xc4 || m3 == 0xdb || m3 == 0xdd)) return m3 < 0xc0 ? IPX_ERR_INVALID : IPX_ERR_UNSUPPORTED;
                    if (p + 2 > len) return IPX_ERR_INVALID;
                    const size_t n3 = be16(d + p);
                    if (n3 < 2 || p + n3 > len) return IPX_ERR_INVALID;
                    p += n3;
                }
            }
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
        default:          // APPn and COM are skipped; anything else is "unknown marker": a FormatError below SOF0, an UnsupportedError above
            if (!((m >= 0xe0 && m <= 0xef) || m == 0xfe)) return m < 0xc0 ? IPX_ERR_INVALID : IPX_ERR_UNSUPPORTED;
            break;
        }
        i += n;
    }
}

}  // namespace ipx

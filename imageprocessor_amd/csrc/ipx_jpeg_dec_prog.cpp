// ipx_jpeg_dec_prog.cpp -- image.Decode for the JPEGs whose entropy coding does not parallelise like a single baseline scan does:
// progressive files (SOF2: spectral selection, successive approximation, EOB runs, refinement passes -- the common web case) and
// sequential files coded in several scans or with more than two Huffman tables per class.
//
// The reference decodes every upload with image.Decode (image_processor.go:47) = Go's image/jpeg; its scan.go walks the scans one
// after the other, every scan refining the coefficient blocks the earlier ones left (d.progCoeffs), and reconstructs the image after
// EOI (reconstructProgressiveImage).  A refinement pass reads one bit per already non-zero coefficient, so a scan's bit positions
// depend on every scan before it: there is no independent piece to hand to a GPU lane.  The split here: THIS file decodes the scans
// to quantised coefficients on the host (one file per thread, ipx_jpeg_decode_batch runs many at once), in the layout the GPU's
// IDCT kernel reads; the coefficients go up (6.3 MB per 1080p 4:2:0 file, still less than its pixels), and dequantisation, the
// integer IDCT, every operator and jpeg.Encode run on the GPU as for baseline files.
//
// Restated from Go 1.24 image/jpeg (reader.go: decode's marker loop, processSOF / DQT / DHT / DRI; scan.go: processSOS, refine,
// refineNonZeroes; huffman.go: decodeHuffman, receiveExtend, decodeBit(s)): garbage between segments is skipped, "\xff\x00" outside
// a scan is ignored, a stray RSTn is ignored, EOI is required.  What Go does and this path does not is reported, never guessed at:
// IPX_ERR_UNSUPPORTED for a restart marker that is not where it belongs (Go resynchronises), quantisation tables redefined between
// the scans of a sequential file (Go dequantises per scan), coefficients beyond int16 -- the worker keeps Go's CPU path for those.
#include <cstring>
#include <vector>

#include "ipx_internal.h"

namespace ipx {

namespace {

const uint8_t kUnzig[64] = {
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

inline uint32_t be16(const uint8_t *p) { return (uint32_t)p[0] << 8 | p[1]; }

// a Huffman table: 9-bit first level (length << 8 | symbol), canonical bounds for longer codes
struct Huff {
    int ncodes = 0;                                // 0: never defined ("uninitialized Huffman table" when a scan decodes with it)
    uint16_t look[512];
    int32_t mincode[17], maxcode[17], valptr[17];  // by code length 1..16; maxcode -1: no code of that length
    uint8_t vals[256];
};

// Known divergence on DAMAGED tables (parity unpinned there): Go's processDHT fills an 8-bit first-level table with `base := uint8(code << (7 - i))`,
// so the codes of an over-subscribed length wrap around modulo 256 and later codes overwrite earlier entries; here such codes are left out of
// the (9-bit) first level and only match through the per-length ranges.  A valid table (Kraft sum <= 1) decodes the same either way; a file
// whose DHT is over-full can decode to different garbage than Go's before both fail.  tools/fuzz_corrupt.py (26 000 files) has not produced one.
void huff_build(Huff &h, const uint8_t counts[16], const uint8_t *vals, int total)
{
    h.ncodes = total;
    memset(h.look, 0, sizeof h.look);
    memcpy(h.vals, vals, (size_t)total);
    int32_t code = 0, idx = 0;
    for (int len = 1; len <= 16; len++) {
        code <<= 1;
        const int cnt = counts[len - 1];
        h.mincode[len] = h.maxcode[len] = h.valptr[len] = -1;
        if (!cnt) continue;
        h.mincode[len] = code; h.valptr[len] = idx;
        if (len <= 9)
            for (int k = 0; k < cnt; k++) {
                const int c = (code + k) << (9 - len);
                if (c + (1 << (9 - len)) > 512) break;      // an over-full length: such codes never match in the first level
                for (int f = 0; f < (1 << (9 - len)); f++) h.look[c | f] = (uint16_t)(len << 8 | vals[idx + k]);
            }
        code += cnt; idx += cnt;
        h.maxcode[len] = code - 1;
    }
}

// The entropy decoder's view of the file.  The accumulator is topped up greedily (up to 64 bits) but never across a marker: fill()
// stops in front of an 0xff that is not followed by 0x00, so whatever is left of a scan's bytes when the scan ends lies before the
// next marker and the marker loop skips it exactly like Go's does from wherever its own look-ahead had stopped.  A demand for bits the
// file does not hold in front of the marker is the error, as in Go ("missing 0xff00 sequence" / "short Huffman data").
struct Bits {
    const uint8_t *d;
    size_t len, pos;
    uint64_t acc = 0;
    int n = 0;
    bool err = false;
    bool fill()   // one more byte; false at a marker or at the end of the file
    {
        if (pos >= len) return false;
        const uint8_t c = d[pos];
        if (c == 0xff) {
            if (pos + 1 >= len || d[pos + 1] != 0) return false;
            pos += 2;
        } else pos++;
        acc = acc << 8 | c; n += 8;
        return true;
    }
    void refill() { while (n <= 56 && fill()) {} }
    int bit()
    {
        if (n == 0) { refill(); if (n == 0) { err = true; return 0; } }
        n--;
        return (int)(acc >> n) & 1;
    }
    uint32_t bits(int k)      // k <= 32
    {
        if (n < k) { refill(); if (n < k) { err = true; return 0; } }
        n -= k;
        return (uint32_t)((acc >> n) & ((1ull << k) - 1));
    }
    int huff(const Huff &h)
    {
        if (!h.ncodes) { err = true; return 0; }
        // first level when nine bits are at hand (or can be had without touching a marker); the bit-serial walk otherwise, which asks
        // for no more bits than the code has -- the end of a scan decodes the same way Go's slow path does
        if (n < 9) refill();
        if (n >= 9) {
            const uint16_t e = h.look[(acc >> (n - 9)) & 511];
            if (e) { n -= e >> 8; return e & 255; }
        }
        int32_t code = 0;
        for (int l = 1; l <= 16; l++) {
            code = code << 1 | bit();
            if (err) return 0;
            if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
        }
        err = true;     // "bad Huffman code"
        return 0;
    }
    int32_t receive_extend(int t)
    {
        if (!t) return 0;
        const int32_t x = (int32_t)bits(t);
        return x < (1 << (t - 1)) ? x + (int32_t)((uint32_t)-1 << t) + 1 : x;
    }
};

struct Decoder {
    const uint8_t *d;
    size_t len;
    int w = 0, h = 0, ncomp = 0;
    bool progressive = false, baseline = false;
    int cid[3], ch[3], cv[3], ctq[3];
    uint16_t quant[4][64];
    Huff hf[2][4];
    int ri = 0;
    bool jfif = false, adobe = false;
    int adobe_transform = 0;
    uint16_t eobrun = 0;
    int mxx = 0, myy = 0, scans = 0;
    bool dqt_after_scan = false;
    // The coefficients live where the IDCT will read them from the start: int16, natural order, blocks in the scan order of an interleaved
    // baseline scan ([MCU][block of the MCU][64]); the DC term sits in element 0 until the end (jpeg_host_decode moves it to its dense
    // array).  Go keeps int32: a value that does not fit sets `wide` at the write that produces it and the file is handed back.
    int16_t *out = nullptr;           // the caller's slice: out_blocks blocks of 64 (pinned memory the upload reads from)
    size_t out_blocks = 0;
    bool out_ready = false;           // zeroed at the first scan, once the frame is known to have the geometry the caller expects
    bool wide = false;
    std::vector<uint64_t> nzmask;     // per block: which zig-zag positions hold a non-zero coefficient (progressive files)
    int bpm = 0, ybl = 0;             // blocks per MCU in the output layout, luma blocks per MCU
    int status = IPX_OK;

    void fail(int s) { if (status == IPX_OK) status = s; }
    void put(int16_t &dst, int32_t v) { if (v != (int32_t)(int16_t)v) wide = true; dst = (int16_t)v; }
    size_t block_index(int k, int bx, int by)   // block (bx, by) of component k's grid in the output layout
    {
        // sampling factors are 1 or 2 here (everything else was refused at the frame header): shifts and masks, no division -- this
        // runs once per block and scan
        const int sx = ch[k] - 1, sy = cv[k] - 1;
        return ((size_t)(by >> sy) * mxx + (bx >> sx)) * bpm + (k == 0 ? (size_t)((by & sy) << sx) + (bx & sx) : (size_t)ybl + k - 1);
    }

    // refineNonZeroes: one bit per coefficient of [zig, zig_end] that is already non-zero, up to where the nz-th zero has been passed
    // (nz < 0: to the end); returns that position.  Go walks the band coefficient by coefficient; the mask of non-zero positions
    // (zig-zag order) that every write keeps lets this go straight to the ones that take a bit -- most blocks of a refinement scan
    // have none -- and the update is arithmetic on the bit, not a branch on it (the bits are as good as random).
    int32_t refine_nonzeroes(Bits &br, int16_t *b, uint64_t m, int32_t zig, int32_t zig_end, int32_t nz, int32_t delta)
    {
        const uint64_t range = (~0ull << zig) & (~0ull >> (63 - zig_end));
        int32_t stop = zig_end + 1;
        if (nz >= 0) {
            uint64_t z = ~m & range;
            for (; nz > 0 && z; nz--) z &= z - 1;
            if (z) stop = __builtin_ctzll(z);
        }
        uint64_t todo = m & range & (stop >= 64 ? ~0ull : (1ull << stop) - 1);
        while (todo) {
            const int cnt = __builtin_popcountll(todo);
            const int k = cnt < 24 ? cnt : 24;
            const uint32_t v = br.bits(k);
            if (br.err) return 0;
            for (int j = k - 1; j >= 0; j--) {
                int16_t &c = b[kUnzig[__builtin_ctzll(todo)]];
                todo &= todo - 1;
                const int32_t sgn = (int32_t)c >> 31;                                 // -1 for a negative coefficient
                const int32_t step = ((delta ^ sgn) - sgn) & -(int32_t)((v >> j) & 1);  // +-delta when the bit is set
                put(c, (int32_t)c + step);
            }
        }
        return stop;
    }

    void refine(Bits &br, int16_t *b, uint64_t &m, const Huff &h, int32_t zig_start, int32_t zig_end, int32_t delta)
    {
        if (zig_start == 0) {
            if (br.bit()) put(b[0], (int32_t)b[0] | delta);
            return;
        }
        int32_t zig = zig_start;
        if (eobrun == 0) {
            for (; zig <= zig_end; zig++) {
                int32_t z = 0;
                const int value = br.huff(h);
                if (br.err) return;
                const int run = value >> 4, size = value & 15;
                if (size == 0) {
                    if (run != 15) {
                        eobrun = (uint16_t)(1u << run);
                        if (run) eobrun |= (uint16_t)br.bits(run);
                        break;
                    }
                } else if (size == 1) {
                    z = br.bit() ? delta : -delta;
                } else { br.err = true; return; }                      // "unexpected Huffman code"
                if (br.err) return;
                zig = refine_nonzeroes(br, b, m, zig, zig_end, run, delta);
                if (br.err) return;
                if (zig > zig_end) { br.err = true; return; }          // "too many coefficients"
                if (z) { put(b[kUnzig[zig]], z); m |= 1ull << zig; }
            }
        }
        if (eobrun > 0) {
            eobrun--;
            if (zig <= zig_end) (void)refine_nonzeroes(br, b, m, zig, zig_end, -1, delta);
        }
    }

    // processSOS: the header at s (sn bytes), the entropy-coded data from *pos on; *pos ends where the marker loop continues
    void scan(const uint8_t *s, size_t sn, size_t *pos)
    {
        if (!ncomp) return fail(IPX_ERR_INVALID);
        if (sn < 6 || (size_t)(4 + 2 * ncomp) < sn || sn % 2) return fail(IPX_ERR_INVALID);
        const int ns = s[0];
        if (sn != (size_t)(4 + 2 * ns)) return fail(IPX_ERR_INVALID);
        int ci[3] = {0, 0, 0}, td[3] = {0, 0, 0}, ta[3] = {0, 0, 0}, total_hv = 0;
        for (int i = 0; i < ns; i++) {
            int k = -1;
            for (int j = 0; j < ncomp; j++) if (s[1 + 2 * i] == cid[j]) k = j;
            if (k < 0) return fail(IPX_ERR_INVALID);
            for (int j = 0; j < i; j++) if (ci[j] == k) return fail(IPX_ERR_INVALID);
            ci[i] = k;
            total_hv += ch[k] * cv[k];
            td[i] = s[2 + 2 * i] >> 4; ta[i] = s[2 + 2 * i] & 15;
            if (td[i] > 3 || ta[i] > 3 || (baseline && (td[i] > 1 || ta[i] > 1))) return fail(IPX_ERR_INVALID);
        }
        if (ncomp > 1 && total_hv > 10) return fail(IPX_ERR_INVALID);
        int32_t zs = 0, ze = 63;
        uint32_t ah = 0, al = 0;
        if (progressive) {
            zs = s[1 + 2 * ns]; ze = s[2 + 2 * ns]; ah = s[3 + 2 * ns] >> 4; al = s[3 + 2 * ns] & 15;
            if ((zs == 0 && ze != 0) || zs > ze || ze >= 64) return fail(IPX_ERR_INVALID);
            if (zs != 0 && ns != 1) return fail(IPX_ERR_INVALID);
            if (ah != 0 && ah != al + 1) return fail(IPX_ERR_INVALID);
        }
        if (!out_ready) {
            ybl = ch[0] * cv[0];
            bpm = ncomp == 1 ? 1 : ybl + 2;
            if ((size_t)mxx * myy * bpm != out_blocks) return fail(IPX_ERR_UNSUPPORTED);       // not the frame the header parser saw
            memset(out, 0, out_blocks * 64 * sizeof(int16_t));
            if (progressive) nzmask.assign(out_blocks, 0);
            out_ready = true;
        }
        scans++;
        Bits br{d, len, *pos};
        int32_t dc[3] = {0, 0, 0};
        int mcu = 0, expected = 0xd0;
        int nbx = 0, nby = 0;                 // the next block of a non-interleaved scan, which walks the component's own block grid
        const int q1 = ns == 1 ? mxx * ch[ci[0]] : 0;
        uint64_t no_mask = 0;
        for (int my = 0; my < myy; my++)
            for (int mx = 0; mx < mxx; mx++) {
                for (int i = 0; i < ns; i++) {
                    const int k = ci[i], hi = ch[k], vi = cv[k];
                    for (int j = 0; j < hi * vi; j++) {
                        int bx, by;
                        if (ns != 1) { bx = hi * mx + (j & (hi - 1)); by = vi * my + (hi == 2 ? j >> 1 : j); }
                        else {
                            bx = nbx; by = nby;
                            if (++nbx == q1) { nbx = 0; nby++; }
                            if (bx * 8 >= w || by * 8 >= h) continue;      // a non-interleaved scan carries no data for blocks outside the image
                        }
                        const size_t gb = block_index(k, bx, by);
                        int16_t *b = out + gb * 64;
                        uint64_t &m = progressive ? nzmask[gb] : no_mask;
                        if (!progressive && scans > 1) {                   // a sequential scan starts from an empty block (b = block{})
                            // (the blocks of one component are written by one scan in a well-formed file; a second scan over the same
                            // component replaces them, as Go's reconstructBlock overwrites the pixels)
                            memset(b, 0, 64 * sizeof(int16_t));
                        }
                        if (ah != 0) {
                            refine(br, b, m, hf[1][ta[i]], zs, ze, (int32_t)(1u << al));
                        } else {
                            int32_t zig = zs;
                            if (zig == 0) {
                                zig++;
                                const int t = br.huff(hf[0][td[i]]);
                                if (br.err) break;
                                if (t > 16) { fail(IPX_ERR_UNSUPPORTED); return; }     // "excessive DC component"
                                dc[k] = (int32_t)((uint32_t)dc[k] + (uint32_t)br.receive_extend(t));   // Go's int32 wraps; a signed overflow here would be undefined
                                put(b[0], (int32_t)((uint32_t)dc[k] << al));
                            }
                            if (zig <= ze && eobrun > 0) eobrun--;
                            else {
                                const Huff &ac = hf[1][ta[i]];
                                for (; zig <= ze; zig++) {
                                    const int value = br.huff(ac);
                                    if (br.err) break;
                                    const int run = value >> 4, size = value & 15;
                                    if (size) {
                                        zig += run;
                                        if (zig > ze) break;
                                        put(b[kUnzig[zig]], (int32_t)((uint32_t)br.receive_extend(size) << al));
                                        m |= 1ull << zig;
                                    } else {
                                        if (run != 15) {
                                            eobrun = (uint16_t)(1u << run);
                                            if (run) eobrun |= (uint16_t)br.bits(run);
                                            eobrun--;
                                            break;
                                        }
                                        zig += 15;
                                    }
                                }
                            }
                        }
                        if (br.err) { fail(IPX_ERR_INVALID); return; }
                    }
                    if (br.err) { fail(IPX_ERR_INVALID); return; }
                }
                mcu++;
                if (ri > 0 && mcu % ri == 0 && mcu < mxx * myy) {
                    // whole bytes still waiting in the accumulator are bytes between the interval's data and the marker: Go's reader,
                    // which takes bytes as late as it can, would stand in front of them and go looking for the marker (findRST)
                    if (br.n >= 8) {
                        size_t at = br.pos;                                 // where that reader stands: in front of the waiting bytes
                        for (int held = br.n / 8; held > 0; held--) at -= at >= 2 && d[at - 1] == 0 && d[at - 2] == 0xff ? 2 : 1;
                        fail(at + 2 > len ? IPX_ERR_INVALID : IPX_ERR_UNSUPPORTED);
                        return;
                    }
                    if (br.pos + 2 > len) { fail(IPX_ERR_INVALID); return; }
                    if (d[br.pos] != 0xff || d[br.pos + 1] != expected) { fail(IPX_ERR_UNSUPPORTED); return; }   // Go would search for the marker
                    br.pos += 2;
                    expected = expected == 0xd7 ? 0xd0 : expected + 1;
                    br.acc = 0; br.n = 0;
                    dc[0] = dc[1] = dc[2] = 0;
                    eobrun = 0;
                }
            }
        *pos = br.pos;
    }

    void run()
    {
        if (len < 2 || d[0] != 0xff || d[1] != 0xd8) return fail(IPX_ERR_INVALID);
        size_t pos = 2;
        bool eoi = false;
        while (status == IPX_OK) {
            if (pos + 2 > len) return fail(IPX_ERR_INVALID);                // no EOI: io.ErrUnexpectedEOF
            uint8_t t0 = d[pos], t1 = d[pos + 1];
            pos += 2;
            while (t0 != 0xff) {                                            // bytes that belong to no segment are skipped
                t0 = t1;
                if (pos >= len) return fail(IPX_ERR_INVALID);
                t1 = d[pos++];
            }
            int m = t1;
            if (m == 0) continue;
            while (m == 0xff) {
                if (pos >= len) return fail(IPX_ERR_INVALID);
                m = d[pos++];
            }
            if (m == 0xd9) { eoi = true; break; }
            if (m >= 0xd0 && m <= 0xd7) continue;
            if (pos + 2 > len) return fail(IPX_ERR_INVALID);
            const int n = (int)be16(d + pos) - 2;
            pos += 2;
            if (n < 0 || pos + (size_t)n > len) return fail(IPX_ERR_INVALID);
            const uint8_t *s = d + pos;
            const size_t sn = (size_t)n;
            pos += sn;
            switch (m) {
            case 0xc0: case 0xc1: case 0xc2: {
                baseline = m == 0xc0; progressive = m == 0xc2;
                if (ncomp) return fail(IPX_ERR_INVALID);
                if (sn == 9) ncomp = 1; else if (sn == 15) ncomp = 3; else return fail(IPX_ERR_UNSUPPORTED);
                if (s[0] != 8) return fail(IPX_ERR_UNSUPPORTED);
                h = (int)be16(s + 1); w = (int)be16(s + 3);
                if (s[5] != ncomp) return fail(IPX_ERR_INVALID);
                for (int c = 0; c < ncomp; c++) {
                    cid[c] = s[6 + 3 * c];
                    for (int j = 0; j < c; j++) if (cid[j] == cid[c]) return fail(IPX_ERR_INVALID);
                    ctq[c] = s[8 + 3 * c];
                    if (ctq[c] > 3) return fail(IPX_ERR_INVALID);
                    int hh = s[7 + 3 * c] >> 4, vv = s[7 + 3 * c] & 15;
                    if (hh < 1 || hh > 4 || vv < 1 || vv > 4) return fail(IPX_ERR_INVALID);
                    if (ncomp == 1) hh = vv = 1;      // reader.go normalises a one-component file first: its factors, 3 included, mean nothing
                    if (hh == 3 || vv == 3) return fail(IPX_ERR_UNSUPPORTED);
                    ch[c] = hh; cv[c] = vv;
                }
                if (ncomp == 3 && (ch[1] != 1 || cv[1] != 1 || ch[2] != 1 || cv[2] != 1 || ch[0] > 2 || cv[0] > 2)) return fail(IPX_ERR_UNSUPPORTED);
                if (w <= 0 || h <= 0) return fail(IPX_ERR_INVALID);
                if ((long long)w * h > (1LL << 28)) return fail(IPX_ERR_UNSUPPORTED);
                mxx = (w + 8 * ch[0] - 1) / (8 * ch[0]); myy = (h + 8 * cv[0] - 1) / (8 * cv[0]);
                break;
            }
            case 0xc4: {
                size_t k = 0;
                while (k < sn) {
                    if (sn - k < 17) return fail(IPX_ERR_INVALID);
                    const int tc = s[k] >> 4, th = s[k] & 15;
                    if (tc > 1 || th > 3 || (baseline && th > 1)) return fail(IPX_ERR_INVALID);
                    int total = 0;
                    for (int b = 0; b < 16; b++) total += s[k + 1 + b];
                    if (total == 0 || total > 256 || k + 17 + (size_t)total > sn) return fail(IPX_ERR_INVALID);
                    huff_build(hf[tc][th], s + k + 1, s + k + 17, total);
                    k += 17 + (size_t)total;
                }
                break;
            }
            case 0xdb: {
                if (scans && !progressive) dqt_after_scan = true;
                size_t k = 0;
                while (k < sn) {
                    const int pq = s[k] >> 4, tq = s[k] & 15;
                    if (tq > 3 || pq > 1) return fail(IPX_ERR_INVALID);
                    const size_t need = pq ? 128 : 64;
                    if (k + 1 + need > sn) return fail(IPX_ERR_INVALID);
                    for (int z = 0; z < 64; z++) quant[tq][z] = pq ? (uint16_t)be16(s + k + 1 + 2 * z) : s[k + 1 + z];
                    k += 1 + need;
                }
                break;
            }
            case 0xdd:
                if (sn != 2) return fail(IPX_ERR_INVALID);
                ri = (int)be16(s);
                break;
            case 0xe0: if (sn >= 5 && !memcmp(s, "JFIF\0", 5)) jfif = true; break;
            case 0xee: if (sn >= 12 && !memcmp(s, "Adobe", 5)) { adobe = true; adobe_transform = s[11]; } break;
            case 0xda: scan(s, sn, &pos); break;
            default:
                if ((m >= 0xe0 && m <= 0xef) || m == 0xfe) break;         // APPn, COM
                return fail(m < 0xc0 ? IPX_ERR_INVALID : IPX_ERR_UNSUPPORTED);
            }
        }
        if (status == IPX_OK && !eoi) fail(IPX_ERR_INVALID);
        if (status == IPX_OK && !scans) fail(IPX_ERR_INVALID);              // "missing SOS marker"
        if (status == IPX_OK && ncomp == 3 && !jfif && ((adobe && adobe_transform == 0) || (cid[0] == 'R' && cid[1] == 'G' && cid[2] == 'B')))
            fail(IPX_ERR_UNSUPPORTED);                                      // isRGB
        if (status == IPX_OK && dqt_after_scan) fail(IPX_ERR_UNSUPPORTED);  // Go dequantises a sequential file scan by scan
    }
};

}  // namespace

// Decodes every scan of a JPEG on the host.  On IPX_OK: info describes the frame like jpeg_parse does (td / ta / scan_* unused), coefs
// holds nblk * 64 int16 in the IDCT kernel's layout -- blocks in MCU order (Y blocks of the MCU row-major, then Cb, Cr), natural
// coefficient order, element 0 left zero -- dcs the nblk DC terms, qnat the FINAL quantisation tables per component in natural order
// (Go dequantises a progressive image after EOI), and *progressive says whether reconstructProgressiveImage's rule applies (blocks
// that hold no image pixel are not reconstructed: they stay zero in the MCU-padded planes).
int jpeg_host_decode(const uint8_t *d, size_t len, JpegDecInfo *info, int16_t *coefs, int16_t *dcs, size_t nblk, uint16_t qnat[3][64],
                     bool *progressive)
{
    Decoder D;
    memset(D.quant, 0, sizeof D.quant);
    memset(D.cid, 0, sizeof D.cid); memset(D.ch, 0, sizeof D.ch); memset(D.cv, 0, sizeof D.cv); memset(D.ctq, 0, sizeof D.ctq);
    D.d = d; D.len = len;
    D.out = coefs; D.out_blocks = nblk;
    D.run();
    if (D.status != IPX_OK) return D.status;
    if (!D.out_ready) return IPX_ERR_INVALID;                                // (no scan at all: the marker parser refuses such a file before)
    memset(info, 0, sizeof *info);
    info->w = D.w; info->h = D.h; info->h0 = D.ch[0]; info->v0 = D.cv[0]; info->ncomp = D.ncomp; info->ri = 0;
    info->ratio = D.ncomp == 1 ? IPX_GRAY : D.ch[0] == 1 ? (D.cv[0] == 1 ? IPX_YCBCR_444 : IPX_YCBCR_440) : (D.cv[0] == 1 ? IPX_YCBCR_422 : IPX_YCBCR_420);
    *progressive = D.progressive;
    if (D.wide) return IPX_ERR_UNSUPPORTED;                                  // Go keeps int32; the GPU pipeline holds int16
    for (size_t gb = 0; gb < nblk; gb++) { dcs[gb] = coefs[gb * 64]; coefs[gb * 64] = 0; }
    for (int c = 0; c < 3; c++)
        for (int zig = 0; zig < 64; zig++) qnat[c][kUnzig[zig]] = c < D.ncomp ? D.quant[D.ctq[c]][zig] : 0;
    return IPX_OK;
}

}  // namespace ipx

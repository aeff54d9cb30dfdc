// ipx_jpeg_dec.hip -- image.Decode for baseline JPEGs on the GPU: entropy decoding and reconstruction.
//
// Go's decoder (image/jpeg scan.go processSOS / decodeHuffman / receiveExtend, reader.go reconstructBlock, idct.go)
// walks the scan serially: Huffman symbols have no length prefix, so a block's position in the byte stream is only
// known once everything before it has been decoded.  Two kernels:
//
//   jpeg_huff_kernel   one LANE per image: 64 images advance through their scans in lock step (the instruction stream
//                      -- fetch bits, table look-up, extend, store -- is the same for every lane; only rare branches
//                      diverge).  Each lane keeps its four 8-bit first-level Huffman tables in LDS (2 KiB per image,
//                      128 KiB per wave), falls back to the canonical mincode / maxcode search for longer codes, and
//                      writes the non-zero quantised coefficients, de-zig-zagged, into a zeroed int16 array.  0xff00
//                      unstuffing follows F.1.2.3.  Where a file has restart intervals (DRI), the host cuts its scan at the
//                      RSTn markers and every interval becomes a lane of its own: Go's processSOS resets the DC
//                      predictions and the bit reader there, so the pieces are independent.  A single image gains nothing here: a lane needs ~0.4 us per
//                      symbol step (a dependent chain of ~100 instructions, and a wave steps at the pace of its slowest
//                      lane), i.e. ~0.5 s for a 1080p file whatever the batch size, so throughput = batch / 0.5 s until
//                      the chip is full (16 k images with per-lane tables, more with shared ones).  Measured without the
//                      coefficient stores and with 16-byte scan reads the time barely moves: it is instruction latency,
//                      not memory.  The remedy is parallelism inside a file (restart intervals where present,
//                      self-synchronising sub-sequences otherwise), not a faster lane.
//   jpeg_idct_kernel   parallel over blocks: b[unzig[zig]] *= qt[zig]; idct (the Chen-Wang integer transform of
//                      idct.go, row pass in registers, column pass through LDS); level shift, clip, 8-byte row stores
//                      into the MCU-padded planes of image.NewYCbCr.
// The planes feed band_conv_kernel directly (ipx_plan_run_dev_ycbcr): decoded pixels never leave HBM.
#include "ipx_internal.h"

namespace ipx {

namespace {

constexpr int kLutStride = 2048 + 8;   // bytes of LDS per lane: 4 x 256 x uint16, padded off the bank period

__constant__ uint8_t c_unzig[64] = {
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// The scan bytes of one image, read 16 at a time: a lane's loads are the latency that nothing else on its SIMD hides, so
// they are made rare.  scan data starts 16-byte aligned in the blob and the blob is padded by 16 bytes.
struct BitReader {
    const uint4 *base;
    uint32_t pos = 0, len = 0;        // bytes consumed / available
    unsigned long long lo = 0, hi = 0;
    int have = 0;                     // bytes left in lo / hi
    unsigned long long acc = 0;
    int cnt = 0;
    bool stop = false;     // a marker or the end of the data was reached: no more bytes
    bool err = false;

    __device__ __forceinline__ void load_chunk()
    {
        const uint4 v = base[pos >> 4];   // have == 0 only at multiples of 16
        lo = (unsigned long long)v.y << 32 | v.x;
        hi = (unsigned long long)v.w << 32 | v.z;
        have = 16;
    }
    __device__ __forceinline__ uint32_t peek_byte()
    {
        if (have == 0) load_chunk();
        return (uint32_t)lo & 0xffu;
    }
    __device__ __forceinline__ uint32_t peek_second()   // the byte after the next one
    {
        if (have >= 2) return (uint32_t)(lo >> 8) & 0xffu;
        return ((const uint8_t *)base)[pos + 1];         // straddles a chunk: rare
    }
    __device__ __forceinline__ void advance()
    {
        if (have == 0) load_chunk();
        lo = (lo >> 8) | (hi << 56);
        hi >>= 8;
        have--;
        pos++;
    }
    __device__ __forceinline__ void refill()
    {
        // four bytes at once when none of them is 0xff (no stuffing, no marker) and they are all in the buffer
        if (cnt <= 32 && have >= 4 && pos + 4 <= len) {
            const uint32_t w4 = (uint32_t)lo, inv = ~w4;
            if (((inv - 0x01010101u) & ~inv & 0x80808080u) == 0) {
                acc = (acc << 32) | __builtin_bswap32(w4);
                cnt += 32;
                lo = (lo >> 32) | (hi << 32);
                hi >>= 32;
                have -= 4;
                pos += 4;
                if (cnt > 56) return;
            }
        }
        while (cnt <= 56 && !stop) {
            if (pos >= len) { stop = true; break; }
            const uint32_t c = peek_byte();
            if (c == 0xff) {
                if (pos + 1 >= len || peek_second() != 0x00) { stop = true; break; }   // marker: left unread
                advance();
            }
            advance();
            acc = (acc << 8) | c;
            cnt += 8;
        }
    }
    // next 16 bits, zero padded past the end of the data
    __device__ __forceinline__ uint32_t peek16()
    {
        if (cnt < 16) refill();
        return cnt >= 16 ? (uint32_t)(acc >> (cnt - 16)) & 0xffffu : (uint32_t)(acc << (16 - cnt)) & 0xffffu;
    }
    __device__ __forceinline__ void skip(int n)
    {
        if (n > cnt) { err = true; cnt = 0; return; }
        cnt -= n;
    }
    __device__ __forceinline__ int receive_extend(int t)   // huffman.go receiveExtend
    {
        if (t == 0) return 0;
        const uint32_t v = peek16() >> (16 - t);
        skip(t);
        return (int)v < (1 << (t - 1)) ? (int)v + (int)(0xffffffffu << t) + 1 : (int)v;
    }
};

// the canonical tables of the long codes (9..16 bits): in LDS for the shared-table kernel, in global memory otherwise
struct SlowTables { const int32_t *maxcode, *valoff; const uint8_t *vals; };   // [4][18], [4][18], [4][256]

__device__ __forceinline__ int decode_symbol(BitReader &br, const uint16_t *lut, const SlowTables &st, int slot)
{
    const uint32_t bits = br.peek16();
    const uint32_t e = lut[bits >> 8];
    if (e) { br.skip((int)(e >> 8)); return (int)(e & 0xffu); }
    for (int len = 9; len <= 16; len++) {     // canonical search for the long codes
        const int code = (int)(bits >> (16 - len));
        if (code <= st.maxcode[slot * 18 + len]) {
            br.skip(len);
            return st.vals[slot * 256 + ((st.valoff[slot * 18 + len] + code) & 255)];
        }
    }
    br.err = true;    // "bad Huffman code"
    return 0;
}

// SHARED: every image of the batch carries the same Huffman tables (the usual case: Annex K tables, or one encoder's
// output), so one 3.7 KiB copy per (single-wave) workgroup serves all lanes and many waves fit on a CU; otherwise each lane
// keeps its own first-level tables (128 KiB per wave, one wave per CU).
template <bool SHARED>
__global__ __launch_bounds__(64) void jpeg_huff_kernel(JpegDecArgs a)
{
    extern __shared__ uint4 lds_raw[];
    uint8_t *lds = (uint8_t *)lds_raw;
    const int lane = threadIdx.x;
    const int item = blockIdx.x * 64 + lane;
    const bool live = item < a.nitems && a.img[item].valid;
    const int img = live ? (int)a.img[item].img : 0;
    uint16_t *lut = (uint16_t *)(SHARED ? lds : lds + lane * kLutStride);
    const JpegDecTables *tab = a.tab + (SHARED ? a.first_valid : (live ? img : 0));
    uint8_t *unz;
    SlowTables st{&tab->maxcode[0][0], &tab->valoff[0][0], &tab->vals[0][0]};
    if (SHARED) {
        const uint4 *src = (const uint4 *)&tab->lut[0][0];
        for (int i = lane; i < 128; i += 64) ((uint4 *)lds)[i] = src[i];   // the four first-level tables, 2 KiB
        unz = lds + 2048;
        unz[lane] = c_unzig[lane];
        int32_t *mc = (int32_t *)(lds + 2048 + 64), *vo = mc + 72;
        uint8_t *vl = (uint8_t *)(vo + 72);
        for (int i = lane; i < 72; i += 64) { mc[i] = (&tab->maxcode[0][0])[i]; vo[i] = (&tab->valoff[0][0])[i]; }
        for (int i = lane; i < 256; i += 64) ((uint32_t *)vl)[i] = ((const uint32_t *)&tab->vals[0][0])[i];
        st = SlowTables{mc, vo, vl};
    } else {
        if (live) {
            const uint4 *src = (const uint4 *)&tab->lut[0][0];
            uint2 *dst = (uint2 *)lut;                   // kLutStride keeps 8-byte alignment
            for (int i = 0; i < 128; i++) { const uint4 v = src[i]; dst[2 * i] = make_uint2(v.x, v.y); dst[2 * i + 1] = make_uint2(v.z, v.w); }
        }
        unz = lds + 64 * kLutStride;                     // the zig-zag -> natural map, shared (lanes index it divergently)
        unz[lane] = c_unzig[lane];
    }
    __syncthreads();
    if (!live) return;                                   // no barrier below
    const JpegDecImage im = a.img[item];
    BitReader br;
    br.base = (const uint4 *)(a.blob + im.scan_off);
    br.len = im.scan_len;
    for (int k = 0; k < (int)im.pad; k++) br.advance();   // the piece starts inside its first 16-byte chunk
    int16_t *coefs = a.coefs + (size_t)img * a.nblk * 64;
    const int ybl = a.ybl, bpm = a.bpm;
    // table slots and DC predictions as scalars selected by the component: arrays indexed by c would live in scratch memory
    // (a global-memory round trip per block)
    const int td0 = im.td[0], td1 = im.td[1], td2 = im.td[2], ta0 = im.ta[0], ta1 = im.ta[1], ta2 = im.ta[2];
    int dc0 = 0, dc1 = 0, dc2 = 0;
    int status = 0;
    bool wide = false;
    for (int m = (int)im.first_mcu; m < (int)(im.first_mcu + im.n_mcu); m++) {
        for (int bi = 0; bi < bpm; bi++) {
            const int c = bi < ybl ? 0 : bi - ybl + 1;
            int16_t *b = coefs + ((size_t)m * bpm + bi) * 64;
            const int td = c == 0 ? td0 : (c == 1 ? td1 : td2), ta = c == 0 ? ta0 : (c == 1 ? ta1 : ta2);
            const int t = decode_symbol(br, lut + td * 256, st, td);
            if (t > 16) { br.err = true; break; }        // "excessive DC component"
            const int diff = br.receive_extend(t);
            const int dcv = (c == 0 ? dc0 : (c == 1 ? dc1 : dc2)) + diff;
            if (c == 0) dc0 = dcv; else if (c == 1) dc1 = dcv; else dc2 = dcv;
            if (dcv < -32768 || dcv > 32767) wide = true;   // Go keeps int32 and decodes on: not representable here (kJpegStatusLast)
            a.dcs[(size_t)img * a.nblk + (size_t)m * bpm + bi] = (int16_t)dcv;
            const uint16_t *aclut = lut + ta * 256;
            for (int zig = 1; zig < 64; zig++) {
                const int v = decode_symbol(br, aclut, st, ta);
                const int r = v >> 4, sz = v & 15;
                if (sz) {
                    zig += r;
                    if (zig > 63) break;
                    const int ac = br.receive_extend(sz);
                    b[unz[zig]] = (int16_t)ac;
                } else {
                    if (r != 15) break;
                    zig += 15;
                }
                if (br.err) break;
            }
            if (br.err) break;
        }
        if (br.err) { status = IPX_ERR_INVALID; break; }
    }
    // An interval that does not end exactly at its marker (damaged data: too few or too many bits) is where Go's processSOS starts
    // searching for the next RSTn (findRST); that heuristic is not restated here -- the file goes back to the CPU path.
    if (!status && im.strict_end && !(br.cnt < 8 && br.pos >= br.len)) status = IPX_ERR_UNSUPPORTED;
    // Pieces of one image report into one word (zeroed by the host).  A sequential decoder stops at the FIRST interval that fails, so the
    // earliest piece's verdict is the image's: negative keys ordered by first_mcu, combined with atomicMin (jpeg_status_of unpacks).
    if (status) atomicMin(&a.status[img], jpeg_status_key(im.first_mcu, status));
    else if (wide) atomicMin(&a.status[img], jpeg_status_key(kJpegStatusLast, IPX_ERR_UNSUPPORTED));
}

// ---- reconstruction -----------------------------------------------------------------------------------------
constexpr int W1 = 2841, W2 = 2676, W3 = 2408, W5 = 1609, W6 = 1108, W7 = 565, R2 = 181;

__device__ __forceinline__ void idct_row(int (&s)[8])   // idct.go, horizontal pass (the all-zero-AC shortcut gives the same values)
{
    int x0 = (int)((uint32_t)s[0] << 11) + 128, x1 = (int)((uint32_t)s[4] << 11), x2 = s[6], x3 = s[2], x4 = s[1], x5 = s[7], x6 = s[5], x7 = s[3];
    int x8 = W7 * (x4 + x5);
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
__device__ __forceinline__ void idct_col(int (&s)[8])   // vertical pass
{
    int y0 = (int)((uint32_t)s[0] << 8) + 8192, y1 = (int)((uint32_t)s[4] << 8), y2 = s[6], y3 = s[2], y4 = s[1], y5 = s[7], y6 = s[5], y7 = s[3];
    int y8 = W7 * (y4 + y5) + 4;
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
    s[0] = (y7 + y1) >> 14; s[1] = (y3 + y2) >> 14; s[2] = (y0 + y4) >> 14; s[3] = (y8 + y6) >> 14;
    s[4] = (y8 - y6) >> 14; s[5] = (y0 - y4) >> 14; s[6] = (y3 - y2) >> 14; s[7] = (y7 - y1) >> 14;
}

// A workgroup takes kIdctMcus consecutive MCUs of ONE MCU row (32 for one-component files): up to 48 blocks, eight threads each.
// The pixels go to LDS at their place in the MCU row's strip of each plane, and the strip is written out in 8-byte pieces that are
// consecutive across lanes -- a wave's store covers a few contiguous row segments.  (The first version stored each thread's eight
// pixels straight from the block: 64 lanes, 64 different cache lines per store instruction, 398 M eight-byte write transactions per
// 1024 1080p files; the kernel ran at 2.8 TB/s of its 9.5 GB.)  Index arithmetic: one division by the blocks-per-MCU count, a
// constant per branch (1, 3, 4 or 6); per-image bases in scalar registers plus 32-bit offsets.
constexpr int kIdctMcus = 8, kIdctMaxBlocks = 48, kIdctThreads = kIdctMaxBlocks * 8;
__global__ __launch_bounds__(kIdctThreads) void jpeg_idct_kernel(JpegDecArgs a, JpegPlanes pl, int wgs_per_row, int mcus_per_wg)
{
    __shared__ int ws[kIdctMaxBlocks * 72];
    __shared__ __attribute__((aligned(8))) uint8_t ob[kIdctMaxBlocks * 64];     // the strips: Y (8 * v0 rows), then Cb, then Cr (8 rows each)
    const int t = threadIdx.x, blk = t >> 3, r = t & 7;
    const int img = blockIdx.y;
    const int vflags = pl.valid[img];
    if (!vflags) return;                                  // uniform over the workgroup
    const bool prog = (vflags & 2) != 0;                  // reconstructProgressiveImage: blocks that hold no image pixel stay zero
    const int ybl = a.ybl, bpm = a.bpm;
    const int my = (int)blockIdx.x / wgs_per_row, mx0 = ((int)blockIdx.x - my * wgs_per_row) * mcus_per_wg;      // scalar
    int mxl, bi;                                          // MCU within the workgroup, block within the MCU
    switch (bpm) {                                        // uniform
    case 1: mxl = blk; bi = 0; break;
    case 3: mxl = blk / 3; bi = blk - mxl * 3; break;
    case 4: mxl = blk >> 2; bi = blk & 3; break;
    default: mxl = blk / 6; bi = blk - mxl * 6; break;    // 6 (4:2:0)
    }
    const int mx = mx0 + mxl;
    const bool live = mxl < mcus_per_wg && mx < a.mxx;
    const int c = bi < ybl ? 0 : bi - ybl + 1;
    const uint32_t gb = (uint32_t)((my * a.mxx + mx) * bpm + bi);
    const int16_t *coefs = a.coefs + (size_t)img * a.nblk * 64;       // per image: scalar
    const int16_t *dcs = a.dcs + (size_t)img * a.nblk;
    const uint16_t *qnat = &a.tab[img].qnat[0][0];
    const int ypitch = 8 * a.h0 * mcus_per_wg, cpitch = 8 * mcus_per_wg;   // bytes per strip row
    const int ybytes = 8 * a.v0 * ypitch, cbytes = 8 * cpitch;
    int s[8];
    if (live) {
        const uint4 v = *(const uint4 *)(coefs + (gb * 64u + (uint32_t)r * 8u));
        const uint4 qv = *(const uint4 *)(qnat + ((uint32_t)c * 64u + (uint32_t)r * 8u));
        const uint32_t wv[4] = {v.x, v.y, v.z, v.w}, qw[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
        for (int i = 0; i < 8; i++) s[i] = (int)(int16_t)(wv[i >> 1] >> (16 * (i & 1))) * (int)((qw[i >> 1] >> (16 * (i & 1))) & 0xffffu);   // b[unzig[zig]] *= qt[zig]
        if (r == 0) s[0] = (int)dcs[gb] * (int)(qw[0] & 0xffffu);                                        // the DC values live in their own dense array
        idct_row(s);
#pragma unroll
        for (int i = 0; i < 8; i++) ws[blk * 72 + r * 8 + i] = s[i];
    }
    __syncthreads();
    if (live) {
#pragma unroll
        for (int i = 0; i < 8; i++) s[i] = ws[blk * 72 + i * 8 + r];   // column r
        idct_col(s);
        // pixel (row i, column r) of the block -> its place in the strip (h0, v0 are 1 or 2)
        int base, pitch;
        if (c == 0) {
            const int bxl = a.h0 == 2 ? bi & 1 : 0, byl = a.h0 == 2 ? bi >> 1 : bi;
            pitch = ypitch; base = byl * 8 * ypitch + (a.h0 * mxl + bxl) * 8 + r;
        } else {
            pitch = cpitch; base = ybytes + (c - 1) * cbytes + mxl * 8 + r;
        }
#pragma unroll
        for (int i = 0; i < 8; i++) ob[base + i * pitch] = (uint8_t)(min(max(s[i], -128), 127) + 128);   // level shift, clip
    }
    __syncthreads();
    // the strips, 8 bytes (one block row) per thread and step
    const int mcus = min(mcus_per_wg, a.mxx - mx0);                    // MCUs of this workgroup that exist
    const int ypieces = ybytes >> 3, cpieces = cbytes >> 3, npieces = a.bpm == 1 ? ypieces : ypieces + 2 * cpieces;
    uint8_t *const py = pl.y + (size_t)img * pl.y_fs, *const pcb = pl.cb + (size_t)img * pl.c_fs, *const pcr = pl.cr + (size_t)img * pl.c_fs;
    for (int p = t; p < npieces; p += kIdctThreads) {
        int q = p, plane_i = 0;
        if (q >= ypieces) { q -= ypieces; plane_i = 1; if (q >= cpieces) { q -= cpieces; plane_i = 2; } }
        const int ppr = (plane_i == 0 ? ypitch : cpitch) >> 3;          // pieces per strip row (a power of two: 8, 16 or 32)
        const int row = q >> (31 - __builtin_clz((unsigned)ppr)), col = q & (ppr - 1);
        if (col >= mcus * (plane_i == 0 ? a.h0 : 1)) continue;          // beyond the last MCU of the row
        const int bx = (plane_i == 0 ? mx0 * a.h0 : mx0) + col, y = (plane_i == 0 ? my * 8 * a.v0 : my * 8) + row;
        const bool inside = plane_i == 0 ? bx * 8 < a.w && (y & ~7) < a.h : bx * 8 * a.h0 < a.w && my * 8 * a.v0 < a.h;   // the block's test (scan.go)
        uint2 v = *(const uint2 *)(ob + (p << 3));
        if (prog && !inside) v = make_uint2(0u, 0u);      // image.NewYCbCr's zeros: Go never writes these blocks of a progressive image
        uint8_t *plane = plane_i == 0 ? py : (plane_i == 1 ? pcb : pcr);
        const uint32_t stride = (uint32_t)(plane_i == 0 ? pl.ystride : pl.cstride);
        *(uint2 *)(plane + ((uint32_t)y * stride + (uint32_t)bx * 8u)) = v;
    }
}

}  // namespace

hipError_t launch_jpeg_huff(const JpegDecArgs &a, hipStream_t s)
{
    if (a.shared_tables) {
        hipLaunchKernelGGL(jpeg_huff_kernel<true>, dim3((a.nitems + 63) / 64), dim3(64), 2048 + 64 + 2 * 72 * 4 + 1024, s, a);
        return hipGetLastError();
    }
    const size_t lds = (size_t)64 * kLutStride + 64;
    static KernelLaunchCache cache;
    hipError_t e = cache.prepare((const void *)jpeg_huff_kernel<false>, 64, lds, nullptr);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(jpeg_huff_kernel<false>, dim3((a.nitems + 63) / 64), dim3(64), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_jpeg_idct(const JpegDecArgs &a, const JpegPlanes &pl, hipStream_t s)
{
    if (a.bpm != 1 && a.bpm != 3 && a.bpm != 4 && a.bpm != 6) return hipErrorInvalidValue;      // (the parser admits no other sampling)
    const int mcus_per_wg = a.bpm == 1 ? 32 : kIdctMcus;                     // 32, 24, 32 or 48 blocks
    const int wgs_per_row = (a.mxx + mcus_per_wg - 1) / mcus_per_wg;
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3((unsigned)(wgs_per_row * a.myy), a.n), dim3(kIdctThreads), 0, s, a, pl, wgs_per_row, mcus_per_wg);
    return hipGetLastError();
}

}  // namespace ipx

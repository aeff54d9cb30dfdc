// ipx_jpeg_dec_par.hip -- Huffman decoding that is parallel INSIDE a scan (no restart markers needed).
//
// jpeg_huff_kernel (ipx_jpeg_dec.hip) gives a scan to one lane, and a lane needs ~0.45 s for a 1080p file.  The
// classic remedy for variable-length codes is used here: Huffman streams resynchronise by themselves.  A decoder that
// starts at an arbitrary bit with a guessed state produces garbage for a while, but once it happens to sit on a true
// symbol boundary with the true (block-in-MCU, zig-zag index) state it stays correct for ever.  So:
//
//   0. unstuff the scan is copied without its stuffed 0x00 bytes and cut at the first marker (count per 1 KiB chunk, scan,
//              copy): after that a position is a plain bit offset, a sub-sequence is a fixed 1 KiB window, and a lane
//              refills its accumulator with one aligned 32-bit LDS read -- no per-byte 0xff tests in the hot loop (with
//              them a wave spent ~8000 cycles per symbol step, because some lane was always in the byte loop);
//   1. spec    one lane per sub-sequence decodes it from a guessed state (block 0, DC expected) -- except sub-sequence 0,
//              whose state is known -- and records where and in which state it leaves: exit[t];
//   2. sync    lane t takes exit[t-1] as its entry; if that differs from the entry it used last time it decodes its
//              sub-sequence again.  Repeat until no entry changes.  entry[0] is true, so by induction every entry is true at
//              the fixed point; because of the self-synchronisation most exits are already right after step 1 and the loop
//              ends after a handful of rounds (the host reads one counter per round);
//   3. write   block ends per sub-sequence -> exclusive scan = index of the block each lane starts in; the lanes decode once
//              more, now storing coefficients (DC as the difference Go's processSOS would add to its running value);
//   4. dc      per image and component, the running sum of the DC differences (F.2.1.3.1).
// After that the coefficient array is what jpeg_huff_kernel would have left, and jpeg_idct_kernel finishes the job.
// A workgroup is one wave = 64 consecutive sub-sequences of ONE image, with that image's Huffman tables in LDS.  The scan
// words of a large batch are read through L1 / L2 (a lane walks its own sub-sequence sequentially): staging 64 KiB-rows in LDS
// (IPX_JPEG_PAR_STAGE=1, row stride sub + 4 bytes against bank aliasing) leaves two waves per CU and is 2.2x slower than the occupancy
// it costs.  A small batch (every wave resident at once, 256-byte rows) is staged: its waves sit alone on their SIMDs and would wait
// for a global load almost every symbol (ipx_jpeg_runtime.hip, where stage_rows is set).
#include <algorithm>
#include <cstdlib>

#include "ipx_internal.h"

namespace ipx {

namespace {

constexpr int kSub = 1024;            // bytes of scan per sub-sequence at most (JpegParArgs::sub: 256 and 512 for small batches)
constexpr int kRowPad = 4;            // a staged sub-sequence takes sub + 4 LDS bytes: lanes at the same depth of their rows hit different banks
constexpr uint32_t kEnd = 0xffffffffu;

__constant__ uint8_t c_unzig_par[64] = {
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// (unstuffed bit position, block-in-MCU index, next zig-zag index; 0 = DC expected)
__device__ __forceinline__ unsigned long long pack_state(uint32_t p, int c, int z) { return (unsigned long long)p | (unsigned long long)c << 32 | (unsigned long long)z << 40; }

struct Tables {          // LDS
    const uint16_t *lut; const int32_t *maxcode, *valoff; const uint8_t *vals, *unz;
    const uint4 *bound;  // [slot][2]: the eight length bounds for codes of 9..16 bits
};

// Reads the UNSTUFFED scan of one image: bit position p <-> word p >> 5, most significant bit first within each byte.
struct Reader {
    const uint8_t *lds;      // staged rows of this workgroup's 64 sub-sequences (+ 16 bytes)
    const uint8_t *g;        // the unstuffed scan in global memory (zero padded), for the words outside the staged window
    uint32_t wg_base, wg_bytes;
    uint32_t sub_shift = 10, sub_mask = kSub - 1, row = kSub + kRowPad;   // staged rows: sub-sequence i >> sub_shift starts at LDS byte (i >> sub_shift) * row
    uint32_t ubits;          // length of the unstuffed scan in bits
    uint32_t widx = 0;       // next 32-bit word
    unsigned long long acc = 0;
    int cnt = 0;

    // Global reads go 16 bytes at a time (scans and pieces start 16-byte aligned): with a 4-byte load per refill and a 1 KiB stride between
    // lanes every 128-byte line was fetched from HBM many times over (PMC: 5.8 GB per sync round for a 0.35 GB scan).
    uint4 buf = make_uint4(0, 0, 0, 0);
    uint32_t buf_q = 0xffffffffu;     // index of the 16-byte group in buf
    __device__ __forceinline__ uint32_t word_at(uint32_t wi)
    {
        const uint32_t i = wi * 4 - wg_base;
        uint32_t wv;
        if (i < wg_bytes) wv = *(const uint32_t *)(lds + (i >> sub_shift) * row + (i & sub_mask));
        else {
            if ((wi >> 2) != buf_q) { buf_q = wi >> 2; buf = *(const uint4 *)(g + (size_t)buf_q * 16); }
            // (two selects and a shift: as a chain of ternaries this became two levels of exec-mask branches)
            const unsigned long long half = wi & 2u ? ((unsigned long long)buf.w << 32 | buf.z) : ((unsigned long long)buf.y << 32 | buf.x);
            wv = (uint32_t)(half >> ((wi & 1u) * 32u));
        }
        return __builtin_bswap32(wv);
    }
    // The word for the NEXT refill is loaded when the current one is consumed: its latency overlaps the ~5 symbols the current word lasts
    // (what matters where few waves share a SIMD: one lane per restart interval gives about one wave per SIMD).
    uint32_t nextw = 0;
    __device__ __forceinline__ void refill()
    {
        if (cnt <= 32) { acc = (acc << 32) | nextw; widx++; cnt += 32; nextw = word_at(widx); }
    }
    __device__ __forceinline__ uint32_t upos() const { return widx * 32u - (uint32_t)cnt; }
    __device__ __forceinline__ bool exhausted() const { return upos() >= ubits; }
    __device__ __forceinline__ void seek(uint32_t p)
    {
        widx = p >> 5;
        acc = 0; cnt = 0;
        nextw = word_at(widx);
        refill();
        cnt -= (int)(p & 31u);
    }
};

// checkpoints per sub-sequence, a sixteenth of it apart (512 bits for 1 KiB).  A re-decode ends at the first checkpoint where it meets its previous trajectory, and a
// wave waits for its slowest lane: with three checkpoints (512, 1536, 4096 bits) round 1 took 1.7 ms per 1024 1080p files against
// 2.65 for the full speculative round, with fifteen 1.12 (the slowest of 64 lanes needs ~3000 bits as a rule)
constexpr int kCk = 15;
constexpr unsigned long long kNoState = ~0ull;           // no packed state has its upper 16 bits set
// (512 bits apart in a 1 KiB sub-sequence, 128 in a 256-byte one: sixteen intervals whatever the length)
__device__ __forceinline__ uint32_t ck_bits(int k, int sub_bytes) { return (uint32_t)(sub_bytes >> 1) * (uint32_t)(k + 1); }

struct NoSink {
    __device__ __forceinline__ void dc(int) {}
    __device__ __forceinline__ void ac(int, int) {}
    __device__ __forceinline__ bool end_block() { return true; }
    __device__ __forceinline__ void bad() {}
};

// Decode from state (c, z) until the position reaches uend <= ubits (or the data ends).  Returns the exit state; *ends counts
// blocks that ended.  sink.end_block() returning false stops the lane (all blocks of the image are done).
// One accumulator refill and one table lookup per symbol: the top 32 bits hold the code (<= 16 bits) AND the value bits that
// follow it (<= 16), so receiveExtend needs no second look at the stream; the bounds test runs once per symbol on the sum.
// Huffman table slots of the three components as scalars (arrays indexed by the component would live in scratch memory: a global-memory
// round trip per symbol -- measured on the first piece kernel: 12.5 ms instead of 3)
struct Slots {
    uint32_t packed;     // four bits per slot: DC tables of the three components, then their AC tables
    template <class Im>
    __device__ __forceinline__ static Slots of(const Im &im)
    {
        return Slots{(uint32_t)im.td[0] | (uint32_t)im.td[1] << 4 | (uint32_t)im.td[2] << 8 | (uint32_t)im.ta[0] << 12 | (uint32_t)im.ta[1] << 16 | (uint32_t)im.ta[2] << 20};
    }
};

// The 64 lanes of a wave are in 64 different states (DC or AC, short or long code, refill or not, end of block or not), so the wave
// walks every arm of this loop on every step and what a step costs is the loop's instruction count, not its memory accesses (a wave
// alone on its SIMD: ~0.8 us per symbol with the first, branchy form -- 350 instructions, 40 branches, half of them exec-mask
// bookkeeping).  Hence one arm for DC and AC with selects, and branches only where memory is touched (refill, the long codes, the sink).
template <class Sink>
__device__ __forceinline__ unsigned long long run(Reader &r, const Tables &T, const Slots im, int bpm, int ybl, int c, int z, uint32_t uend,
                                                  Sink &sink, uint32_t *ends)
{
    uint32_t nend = 0;
    unsigned long long out;
    const unsigned long long kOut = pack_state(kEnd, 0, 0);
    for (;;) {
        const uint32_t p = r.upos();
        if (p >= uend) { out = p >= r.ubits ? kOut : pack_state(p, c, z); break; }
        const bool dcs = z == 0;
        const int comp = c < ybl ? 0 : c - ybl + 1;
        const int slot = (int)((im.packed >> (comp * 4 + (dcs ? 0 : 12))) & 15u);
        r.refill();
        const uint32_t bits = (uint32_t)(r.acc >> (r.cnt - 32));        // cnt >= 33 after the refill; past the end the buffer holds zeros
        const uint32_t e = T.lut[slot * 256 + (bits >> 24)];
        int len = (int)(e >> 8), sym = (int)(e & 0xffu);
        if (!e) {
            // a code of 9..16 bits: canonical codes are ordered, so its length is 9 + the number of length bounds the 16-bit window has
            // passed -- eight compares on two LDS reads instead of a loop of dependent table reads that every lane of the wave waits for
            const uint32_t x = bits >> 16;
            const uint4 b0 = T.bound[slot * 2], b1 = T.bound[slot * 2 + 1];
            len = 9 + (int)(x >= b0.x) + (int)(x >= b0.y) + (int)(x >= b0.z) + (int)(x >= b0.w) + (int)(x >= b1.x) + (int)(x >= b1.y) + (int)(x >= b1.z);
            const int at = slot * 256 + ((T.valoff[slot * 18 + len] + (int)(bits >> (32 - len))) & 255);
            sym = x < b1.w ? (int)T.vals[at] : -1;
        }
        const bool nocode = sym < 0;                                     // no such code: a speculative decoder just moves on by one bit
        const int run_ = sym >> 4, sz = sym & 15;
        // value bits after the code (receiveExtend).  AC: Go does zig += val0; if zig > zigEnd { break } -- the value bits stay unread
        int nbits = dcs ? (sym <= 16 ? sym : 0) : (sz && z + run_ <= 63 ? sz : 0);
        nbits = nocode ? 0 : nbits;
        len = nocode ? 1 : len;
        r.cnt -= len + nbits;
        if (r.upos() > r.ubits) { out = kOut; break; }                   // the data ended inside this symbol
        const uint32_t v = ((bits << len) >> 1) >> (31 - nbits);         // nbits == 0: 0
        const int val = v < ((1u << nbits) >> 1) ? (int)v + 1 - (1 << nbits) : (int)v;
        const bool bad = nocode || (dcs && sym > 16);
        bool block_done = false;
        if (bad) sink.bad();                                             // (the state stays as it is)
        else if (dcs) { sink.dc(val); z = 1; }
        else {
            // sz != 0: skip run_ zeros and store; (15, 0): sixteen zeros; (r, 0): end of block
            const int zz = z + (sz ? run_ : (run_ == 15 ? 16 : 64));
            if (sz && zz <= 63) sink.ac(zz, val);
            z = zz + (sz ? 1 : 0);
            block_done = z > 63;
        }
        if (block_done) {
            nend++;
            c = c + 1 == bpm ? 0 : c + 1;
            z = 0;
            if (!sink.end_block()) { out = kOut; break; }
        }
    }
    *ends = nend;
    return out;
}

__device__ __forceinline__ Tables stage(uint8_t *lds, const JpegParArgs &a, const JpegParImage &im, int py, int lane, int first_sub, Reader &r)
{
    // tables of this image, then the workgroup's 64 sub-sequences (+ 16 bytes of look-ahead)
    const JpegDecTables *tab = a.tab + im.img;
    uint8_t *tb = lds;
    for (int i = lane; i < 128; i += 64) ((uint4 *)tb)[i] = ((const uint4 *)&tab->lut[0][0])[i];
    uint8_t *unz = tb + 2048;
    unz[lane] = c_unzig_par[lane];
    int32_t *mc = (int32_t *)(tb + 2048 + 64), *vo = mc + 72;
    uint8_t *vl = (uint8_t *)(vo + 72);
    for (int i = lane; i < 72; i += 64) { mc[i] = (&tab->maxcode[0][0])[i]; vo[i] = (&tab->valoff[0][0])[i]; }
    for (int i = lane; i < 256; i += 64) ((uint32_t *)vl)[i] = ((const uint32_t *)&tab->vals[0][0])[i];
    uint32_t *bd = (uint32_t *)(tb + 3712);                   // 2048 + 64 + 2 * 288 + 1024, 16-byte aligned
    if (lane < 32) bd[lane] = (&tab->bound[0][0])[lane];
    uint8_t *rows = tb + 4096;
    const uint8_t *scan = a.ublob + im.scan_off;           // unstuffed copy: same offsets as the packed scans, zero padded
    const uint32_t base = (uint32_t)first_sub * (uint32_t)a.sub;
    const uint32_t cap = (im.scan_len + 15u) & ~15u;       // the unstuffed scan is no longer than the stuffed one
    const uint32_t avail = a.stage_rows == 1 && cap > base ? min(cap - base, (uint32_t)(64 * a.sub + 16)) : 0u;
    r.sub_shift = 31u - (uint32_t)__builtin_clz((uint32_t)a.sub); r.sub_mask = (uint32_t)a.sub - 1u; r.row = (uint32_t)a.sub + kRowPad;
    for (uint32_t ch = lane; ch < (avail >> 4); ch += 64) {
        const uint4 v = *(const uint4 *)(scan + base + ch * 16);
        const uint32_t i = ch * 16;
        uint32_t *d = (uint32_t *)(rows + (i >> r.sub_shift) * r.row + (i & r.sub_mask));   // 16-byte pieces never straddle a row; rows are only 4-byte aligned
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    r.lds = rows; r.g = scan; r.wg_base = base; r.wg_bytes = avail; r.ubits = a.ulen[py] * 8u;
    return Tables{(const uint16_t *)tb, mc, vo, vl, unz, (const uint4 *)bd};
}

// tables, 64 rows and the 16 bytes of look-ahead in a 65th
constexpr size_t par_lds(int sub) { return 4096 + (size_t)65 * (size_t)(sub + kRowPad) + 64; }
constexpr size_t kParLds = par_lds(kSub);

// step 0a: stuffed zeros per 1 KiB chunk of the scan, and where the scan ends (the first 0xff not followed by 0x00)
__global__ __launch_bounds__(256) void par_count_kernel(JpegParArgs a)
{
    const JpegParImage im = a.img[blockIdx.y];
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= (int)im.nsub) return;
    const uint8_t *scan = a.blob + im.scan_off;
    const uint32_t b0 = (uint32_t)t * (uint32_t)a.sub, b1 = min(b0 + (uint32_t)a.sub, im.scan_len);
    uint32_t n = 0, first_marker = 0xffffffffu;
    uint32_t prev = b0 > 0 ? scan[b0 - 1] : 0u;
    for (uint32_t j = b0; j < b1; j += 16) {                 // scans start 16-byte aligned (and are padded), chunks are multiples of 16
        const uint4 v4 = *(const uint4 *)(scan + j);         // 16 bytes per load: 4-byte loads at a 1 KiB lane stride refetch every line from HBM
        const uint32_t w4[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t wv = w4[q];
            const uint32_t inv = ~wv;
            if (prev != 0xff && ((inv - 0x01010101u) & ~inv & 0x80808080u) == 0 && j + 4 * q + 4 <= b1) { prev = wv >> 24; continue; }   // no 0xff here
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t c = (wv >> (8 * k)) & 0xffu;
                if (j + 4 * q + k < b1) {
                    n += (prev == 0xff && c == 0x00);
                    if (prev == 0xff && c != 0x00) first_marker = min(first_marker, j + 4 * q + k - 1);
                    prev = c;
                }
            }
        }
    }
    if (b1 == im.scan_len && prev == 0xff) first_marker = min(first_marker, im.scan_len - 1);   // a lone 0xff at the very end
    a.stuffed[im.sub_off + t] = n;
    if (first_marker != 0xffffffffu) atomicMin(a.scan_end + blockIdx.y, first_marker);
}

// step 0b: the scan without its stuffed zeros, up to the marker (a.stuffed holds the exclusive scan of the counts)
__global__ __launch_bounds__(256) void par_unstuff_kernel(JpegParArgs a)
{
    const JpegParImage im = a.img[blockIdx.y];
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= (int)im.nsub) return;
    const uint8_t *scan = a.blob + im.scan_off;
    uint8_t *dst = a.ublob + im.scan_off;
    const uint32_t end = min(a.scan_end[blockIdx.y], im.scan_len);
    const uint32_t b0 = (uint32_t)t * (uint32_t)a.sub, b1 = min(b0 + (uint32_t)a.sub, end);
    uint32_t o = b0 - a.stuffed[im.sub_off + t];
    uint32_t prev = b0 > 0 ? scan[b0 - 1] : 0u;
    // whole input words (chunks start 4-byte aligned) through a small byte queue; output words are stored whole only when
    // this thread owns all four bytes and they are aligned, the ragged ends go out byte by byte (the neighbours' do too)
    unsigned long long q = 0;
    int nq = 0;
    uint32_t j = b0;
    auto push = [&](uint32_t c) { q |= (unsigned long long)c << (8 * nq); nq++; };
    auto drain = [&](bool all) {
        while (nq >= 4 || (all && nq > 0)) {
            if (nq >= 4 && (o & 3u) == 0) { *(uint32_t *)(dst + o) = (uint32_t)q; q >>= 32; nq -= 4; o += 4; }
            else { dst[o++] = (uint8_t)q; q >>= 8; nq--; }
        }
    };
    auto word = [&](uint32_t wv) {
        const uint32_t inv = ~wv;
        if (prev != 0xff && ((inv - 0x01010101u) & ~inv & 0x80808080u) == 0) {   // no 0xff in sight: nothing to drop
            q |= (unsigned long long)wv << (8 * nq);
            nq += 4;
            prev = wv >> 24;
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t c = (wv >> (8 * k)) & 0xffu;
                if (!(prev == 0xff && c == 0x00)) push(c);
                prev = c;
            }
        }
        drain(false);
    };
    for (; j + 16 <= b1; j += 16) {                          // 16 bytes per load (see par_count_kernel)
        const uint4 v4 = *(const uint4 *)(scan + j);
        word(v4.x); word(v4.y); word(v4.z); word(v4.w);
    }
    for (; j + 4 <= b1; j += 4) word(*(const uint32_t *)(scan + j));
    for (; j < b1; j++) {
        const uint32_t c = scan[j];
        if (!(prev == 0xff && c == 0x00)) push(c);
        prev = c;
        drain(false);
    }
    drain(true);
    if (b0 < end && b1 == end) a.ulen[blockIdx.y] = o;       // the chunk that holds the last byte of the scan
}

// steps 1 and 2.  round 0: speculative; round >= 1: re-decode where the entry changed.
// A round >= 1 is a fixed point WITHIN the wave first: lane t takes lane t - 1's exit of this very pass (a shuffle) and decodes again
// while any entry in the wave still changes; only the first lane of a wave depends on what another wave left in the round before.  With
// 256-byte sub-sequences a stream needs several of them to fall into step, and one host round per sub-sequence (a launch and a wait
// each) was most of a small batch's decode: 5 rounds for 8 files, 2 now.
__global__ __launch_bounds__(64) void par_sync_kernel(JpegParArgs a, int round)
{
    extern __shared__ uint4 lds_raw[];
    const JpegParImage im = a.img[blockIdx.y];
    const int first = blockIdx.x * 64, lane = threadIdx.x, t = first + lane;
    if (first >= (int)im.nsub) return;
    Reader r;
    const Tables T = stage((uint8_t *)lds_raw, a, im, blockIdx.y, lane, first, r);
    const bool live = t < (int)im.nsub;
    const size_t s = im.sub_off + (live ? t : 0);
    const unsigned long long *exit_prev = round & 1 ? a.exit_a : a.exit_b;
    unsigned long long *exit_next = round & 1 ? a.exit_b : a.exit_a;
    const uint32_t ustart = (uint32_t)t * (uint32_t)a.sub * 8u, uend = min(ustart + (uint32_t)a.sub * 8u, r.ubits);
    unsigned long long *ck = a.ck_state + s * kCk;
    uint32_t *cke = a.ck_ends + s * kCk;

    // one decode of this lane's sub-sequence from `entry`; again = there was an earlier decode whose checkpoints may end this one early.
    // Checkpoints: the decoder state at the first symbol boundary at or after ustart + ck_bits(k).  A re-decode that arrives at a
    // checkpoint in the state the previous decode of this sub-sequence had there decodes the rest exactly as before: it stops, keeps
    // the old exit and adds the old block-end count of the remainder.  Streams re-synchronise within a few hundred bits as a rule, so a
    // re-decode costs a fraction of the first one.
    auto decode = [&](unsigned long long entry, bool again, unsigned long long old_exit, uint32_t old_total, uint32_t *total) -> unsigned long long {
        const uint32_t p = (uint32_t)entry;
        uint32_t ends = 0;
        unsigned long long out;
        int k = 0;
        if (p == kEnd || p >= uend) out = p >= r.ubits ? pack_state(kEnd, 0, 0) : entry;   // nothing of this sub-sequence is left to decode
        else {
            r.seek(p);
            NoSink sink;
            int c = (int)(entry >> 32) & 0xff, z = (int)(entry >> 40) & 0xff;
            bool ended = false;
            out = entry;
            for (; k < kCk; k++) {
                const uint32_t limit = ustart + ck_bits(k, a.sub);
                if (limit >= uend) break;                               // the remaining checkpoints lie beyond this sub-sequence
                if (p >= limit) { ck[k] = kNoState; continue; }         // entered beyond it: not on this trajectory
                uint32_t e1;
                out = run(r, T, Slots::of(im), a.bpm, a.ybl, c, z, limit, sink, &e1);
                ends += e1;
                if ((uint32_t)out == kEnd) { ended = true; break; }
                if (again && ck[k] == out) {
                    const uint32_t old_at = cke[k], shift = ends - old_at;
                    cke[k] = ends;
                    for (int j = k + 1; j < kCk; j++) cke[j] += shift;
                    *total = ends + (old_total - old_at);
                    return old_exit;
                }
                ck[k] = out; cke[k] = ends;
                c = (int)(out >> 32) & 0xff; z = (int)(out >> 40) & 0xff;
            }
            if (!ended) {
                uint32_t e1;
                out = run(r, T, Slots::of(im), a.bpm, a.ybl, c, z, uend, sink, &e1);
                ends += e1;
            }
        }
        for (int j = k; j < kCk; j++) ck[j] = kNoState;                 // checkpoints this decode did not reach
        *total = ends;
        return out;
    };

    if (round == 0) {
        if (!live) return;
        const unsigned long long entry = ustart < r.ubits ? pack_state(ustart, 0, 0) : pack_state(kEnd, 0, 0);   // t == 0: the truth; t > 0: a guess
        a.entry[s] = entry;
        uint32_t total = 0;
        exit_next[s] = decode(entry, false, 0, 0, &total);
        a.ends[s] = total;
        return;
    }
    unsigned long long entry = live ? a.entry[s] : 0, mine = live ? exit_prev[s] : 0;
    uint32_t total = live ? a.ends[s] : 0;
    const unsigned long long left = live && t > 0 && lane == 0 ? exit_prev[s - 1] : 0;     // what the wave before left in the round before
    bool any_change = false;
    for (int pass = 0; pass < 64; pass++) {                                               // (an entry moves at most one lane per pass)
        const unsigned long long up = __shfl_up(mine, 1);
        const unsigned long long want = lane == 0 ? left : up;
        const bool changed = live && t > 0 && want != entry;
        if (!__any(changed)) break;
        if (changed) {
            entry = want;
            any_change = true;
            mine = decode(entry, true, mine, total, &total);
        }
    }
    if (!live) return;
    if (any_change) { atomicAdd(a.changed, 1u); a.entry[s] = entry; a.ends[s] = total; }
    exit_next[s] = mine;
}

struct CoefSink {
    int16_t *coefs; int16_t *dcs; const uint8_t *unz;
    uint32_t g, nblk;
    int *status;
    __device__ __forceinline__ void dc(int d)
    {
        if (g >= nblk) return;
        if (d < -32768 || d > 32767) { atomicMin(status, jpeg_status_key(kJpegStatusLast, IPX_ERR_UNSUPPORTED)); return; }   // see kJpegStatusLast
        dcs[g] = (int16_t)d;
    }
    bool drop = false;                   // (diagnostic: IPX_JPEG_PAR_STAGE=2 runs the write pass without its coefficient stores)
    __device__ __forceinline__ void ac(int z, int v) { if (g < nblk && !drop) coefs[(size_t)g * 64 + unz[z]] = (int16_t)v; }
    __device__ __forceinline__ bool end_block() { g++; return g < nblk; }
    __device__ __forceinline__ void bad() { if (g < nblk) atomicMin(status, jpeg_status_key(0, IPX_ERR_INVALID)); }   // "bad Huffman code" in real data
};

// step 3: a.ends holds the exclusive scan of the block ends = the block each lane starts in
__global__ __launch_bounds__(64) void par_write_kernel(JpegParArgs a)
{
    extern __shared__ uint4 lds_raw[];
    const JpegParImage im = a.img[blockIdx.y];
    const int first = blockIdx.x * 64, lane = threadIdx.x, t = first + lane;
    if (first >= (int)im.nsub) return;
    Reader r;
    const Tables T = stage((uint8_t *)lds_raw, a, im, blockIdx.y, lane, first, r);
    if (t >= (int)im.nsub) return;
    const size_t s = im.sub_off + t;
    const unsigned long long entry = a.entry[s];
    const uint32_t p = (uint32_t)entry;
    const uint32_t uend = min(((uint32_t)t + 1) * (uint32_t)a.sub * 8u, r.ubits);
    if (p == kEnd || p >= uend) return;
    CoefSink sink{a.coefs + (size_t)im.img * a.nblk * 64, a.dcs + (size_t)im.img * a.nblk, T.unz, a.ends[(size_t)blockIdx.y * a.max_nsub + t], (uint32_t)a.nblk, a.status + im.img};
    sink.drop = a.stage_rows == 2;
    if (sink.g >= sink.nblk) return;
    r.seek(p);
    uint32_t ends;
    (void)run(r, T, Slots::of(im), a.bpm, a.ybl, (int)(entry >> 32) & 0xff, (int)(entry >> 40) & 0xff, uend, sink, &ends);
}

// step 4: DC differences -> DC values, per component in scan order; also the "scan ran out of data" verdict.
// Tiles of 1024 MCUs, one per thread: the running sums of the three components by wave shuffles and sixteen wave sums in LDS, the next
// tile's loads in flight meanwhile.  (The first form gave a thread nmcu / 1024 consecutive MCUs and walked them twice, one dependent
// 2-byte load after the other: 79 us for ONE 1080p file, 0.94 ms per 1024 files with 256 threads per image.)
constexpr int kDcThreads = 1024;
constexpr int kDcBlocks = 10;         // blocks per MCU at most (B.2.3: the sampling factors of a scan's components sum to <= 10)
__global__ __launch_bounds__(kDcThreads) void par_dc_kernel(JpegParArgs a)
{
    __shared__ int wsum[2][16][3];
    const JpegParImage im = a.img[blockIdx.x];
    int16_t *dcs = a.dcs + (size_t)im.img * a.nblk;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, bpm = a.bpm, ybl = a.ybl, nmcu = a.nblk / bpm;
    if (t == 0 && a.total_ends[blockIdx.x] < (uint32_t)a.nblk) atomicMin(a.status + im.img, jpeg_status_key(0, IPX_ERR_INVALID));   // "short Huffman data"
    auto load = [&](int m, int (&x)[kDcBlocks]) {
#pragma unroll
        for (int bi = 0; bi < kDcBlocks; bi++) x[bi] = bi < bpm && m < nmcu ? (int)dcs[(size_t)m * bpm + bi] : 0;
    };
    int carry0 = 0, carry1 = 0, carry2 = 0, nxt[kDcBlocks];
    bool bad = false;
    load(t, nxt);
    int flip = 0;
    for (int base = 0; base < nmcu; base += kDcThreads, flip ^= 1) {
        const int m = base + t;
        int x[kDcBlocks];
#pragma unroll
        for (int bi = 0; bi < kDcBlocks; bi++) x[bi] = nxt[bi];
        if (base + kDcThreads < nmcu) load(m + kDcThreads, nxt);
        int mine0 = 0, mine1 = 0, mine2 = 0;          // (scalars and selects: arrays indexed by the component would live in scratch memory)
#pragma unroll
        for (int bi = 0; bi < kDcBlocks; bi++) {
            mine0 += bi < ybl ? x[bi] : 0;
            mine1 += bi == ybl ? x[bi] : 0;
            mine2 += bi > ybl ? x[bi] : 0;            // (x is 0 beyond bpm)
        }
        int inc0 = mine0, inc1 = mine1, inc2 = mine2;
        for (int d = 1; d < 64; d <<= 1) {
            const int y0 = __shfl_up(inc0, d), y1 = __shfl_up(inc1, d), y2 = __shfl_up(inc2, d);
            if (lane >= d) { inc0 += y0; inc1 += y1; inc2 += y2; }
        }
        if (lane == 63) { wsum[flip][wave][0] = inc0; wsum[flip][wave][1] = inc1; wsum[flip][wave][2] = inc2; }
        __syncthreads();                               // (two copies of wsum: a fast wave may be a tile ahead of a slow one's reads)
        int run0 = carry0 + inc0 - mine0, run1 = carry1 + inc1 - mine1, run2 = carry2 + inc2 - mine2;
        for (int w = 0; w < 16; w++) {
            const int q0 = wsum[flip][w][0], q1 = wsum[flip][w][1], q2 = wsum[flip][w][2];
            if (w < wave) { run0 += q0; run1 += q1; run2 += q2; }
            carry0 += q0; carry1 += q1; carry2 += q2;
        }
        if (m < nmcu) {
#pragma unroll
            for (int bi = 0; bi < kDcBlocks; bi++) {
                if (bi < bpm) {
                    run0 += bi < ybl ? x[bi] : 0;
                    run1 += bi == ybl ? x[bi] : 0;
                    run2 += bi > ybl ? x[bi] : 0;
                    const int r = bi < ybl ? run0 : (bi == ybl ? run1 : run2);
                    bad |= r < -32768 || r > 32767;
                    dcs[(size_t)m * bpm + bi] = (int16_t)r;
                }
            }
        }
    }
    if (bad) atomicMin(a.status + im.img, jpeg_status_key(kJpegStatusLast, IPX_ERR_UNSUPPORTED));   // a DC value beyond int16: see kJpegStatusLast
}

// ---- pieces with a known start state: restart intervals, and scans too short to be worth speculating on ------------------------
// One lane per piece, as jpeg_huff_kernel (ipx_jpeg_dec.hip) has it, but through the reader above: the piece is first copied without its
// stuffed zeros, so the decode loop refills with aligned words and never tests bytes (that test, divergent across the wave, is what made
// the byte-wise kernel spend thousands of cycles per symbol step).
__global__ __launch_bounds__(256) void piece_unstuff_kernel(JpegDecArgs a, uint8_t *ublob, uint32_t *ulen)
{
    const int item = blockIdx.x * 256 + threadIdx.x;
    if (item >= a.nitems) return;
    const JpegDecImage im = a.img[item];
    if (!im.valid) { ulen[item] = 0; return; }
    const uint8_t *src = a.blob + im.scan_off + im.pad;
    const uint32_t n = im.scan_len - im.pad;
    uint8_t *dst = ublob + im.uoff;
    uint32_t o = 0, prev = 0, j = 0;
    unsigned long long q = 0;
    int nq = 0;
    bool marker = false;
    auto drain = [&](bool all) {
        while (nq >= 4 || (all && nq > 0)) {
            if (nq >= 4) { *(uint32_t *)(dst + o) = (uint32_t)q; q >>= 32; nq -= 4; o += 4; }   // o stays a multiple of 4 until the final bytes
            else { dst[o++] = (uint8_t)q; q >>= 8; nq--; }
        }
    };
    auto byte = [&](uint32_t c) {                         // false: a marker -- the entropy-coded data ended one byte earlier
        if (prev == 0xff) {
            if (c != 0x00) return false;
            prev = 0x100;                                 // the stuffed zero is dropped and is nobody's predecessor
            return true;
        }
        q |= (unsigned long long)c << (8 * nq); nq++;
        prev = c;
        return true;
    };
    for (; j < n && ((uintptr_t)(src + j) & 3); j++) { if (!byte(src[j])) { marker = true; break; } drain(false); }
    if (!marker)
        for (; j + 4 <= n; j += 4) {
            const uint32_t wv = *(const uint32_t *)(src + j);
            const uint32_t inv = ~wv;
            if (prev != 0xff && ((inv - 0x01010101u) & ~inv & 0x80808080u) == 0) {   // no 0xff in sight: nothing to drop
                q |= (unsigned long long)wv << (8 * nq);
                nq += 4;
                prev = wv >> 24;
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (!marker && !byte((wv >> (8 * k)) & 0xffu)) marker = true;
                if (marker) break;
            }
            drain(false);
        }
    if (!marker)
        for (; j < n; j++) { if (!byte(src[j])) { marker = true; break; } drain(false); }
    // the 0xff that opened the marker (or a lone 0xff at the very end) is not data
    if (prev == 0xff && nq > 0) { nq--; q &= ~(0xffull << (8 * nq)); }
    else if (prev == 0xff && o > 0) o--;                  // it had already gone out: step back over it (the reader is bounded by ulen)
    drain(true);
    ulen[item] = o;
}

struct PieceSink {
    int16_t *coefs; int16_t *dcs; const uint8_t *unz;
    uint32_t g, g_end;
    int c, bpm, ybl;                 // block within the MCU, for the DC predictions
    int dc0, dc1, dc2;
    bool err, wide;
    __device__ __forceinline__ void dc(int d)
    {
        // every prediction is updated arithmetically: written as v = (k == 0 ? dc0 : k == 1 ? dc1 : dc2) + d the compiler turns the
        // three members into an indexed array in scratch memory (a round trip and a vmcnt(0) per block)
        const int k = c < ybl ? 0 : c - ybl + 1;
        dc0 += k == 0 ? d : 0; dc1 += k == 1 ? d : 0; dc2 += k == 2 ? d : 0;
        const int v = (k == 0 ? dc0 : 0) + (k == 1 ? dc1 : 0) + (k == 2 ? dc2 : 0);
        if (v < -32768 || v > 32767) { wide = true; return; }  // Go keeps int32 and decodes on: not representable here (kJpegStatusLast)
        dcs[g] = (int16_t)v;
    }
    __device__ __forceinline__ void ac(int z, int v) { coefs[(size_t)g * 64 + unz[z]] = (int16_t)v; }
    __device__ __forceinline__ bool end_block() { g++; c = c + 1 == bpm ? 0 : c + 1; return g < g_end && !err; }
    __device__ __forceinline__ void bad() { err = true; }   // "bad Huffman code" / "excessive DC component": nothing speculative here
};

__global__ __launch_bounds__(64) void piece_decode_kernel(JpegDecArgs a, const uint8_t *ublob, const uint32_t *ulen)
{
    __shared__ __attribute__((aligned(16))) uint8_t tb[4096];
    const int lane = threadIdx.x, item = blockIdx.x * 64 + lane;
    const JpegDecTables *tab = a.tab + a.img[blockIdx.x * 64].tab_img;   // the host pads table groups to whole workgroups
    for (int i = lane; i < 128; i += 64) ((uint4 *)tb)[i] = ((const uint4 *)&tab->lut[0][0])[i];
    uint8_t *unz = tb + 2048;
    unz[lane] = c_unzig_par[lane];
    int32_t *mc = (int32_t *)(tb + 2048 + 64), *vo = mc + 72;
    uint8_t *vl = (uint8_t *)(vo + 72);
    for (int i = lane; i < 72; i += 64) { mc[i] = (&tab->maxcode[0][0])[i]; vo[i] = (&tab->valoff[0][0])[i]; }
    for (int i = lane; i < 256; i += 64) ((uint32_t *)vl)[i] = ((const uint32_t *)&tab->vals[0][0])[i];
    uint32_t *bd = (uint32_t *)(tb + 3712);
    if (lane < 32) bd[lane] = (&tab->bound[0][0])[lane];
    __syncthreads();
    if (item >= a.nitems) return;
    // field by field: a per-lane copy of the whole struct (byte arrays inside) ends up in scratch memory
    const JpegDecImage *ip = a.img + item;
    const uint32_t n_mcu = ip->n_mcu, first_mcu = ip->first_mcu, img = ip->img;
    if (!ip->valid || n_mcu == 0) return;
    const Slots slots = Slots::of(*ip);
    const bool strict_end = ip->strict_end != 0;
    const Tables T{(const uint16_t *)tb, mc, vo, vl, unz, (const uint4 *)bd};
    Reader r;
    r.lds = nullptr; r.g = ublob + ip->uoff; r.wg_base = 0; r.wg_bytes = 0; r.ubits = ulen[item] * 8u;
    r.seek(0);
    const uint32_t g0 = first_mcu * (uint32_t)a.bpm;
    PieceSink sink{a.coefs + (size_t)img * a.nblk * 64, a.dcs + (size_t)img * a.nblk, unz, g0, g0 + n_mcu * (uint32_t)a.bpm,
                   0, a.bpm, a.ybl, 0, 0, 0, false, false};
    uint32_t ends;
    (void)run(r, T, slots, a.bpm, a.ybl, 0, 0, r.ubits, sink, &ends);
    int status = 0;
    if (sink.err || sink.g < sink.g_end) status = IPX_ERR_INVALID;          // a bad code, or the data ran out before the last block
    // An interval that does not end exactly at its marker (damaged data: too few or too many bits) is where Go's processSOS starts
    // searching for the next RSTn (findRST); that heuristic is not restated here -- the file goes back to the CPU path.
    else if (strict_end && r.ubits - r.upos() >= 8u) status = IPX_ERR_UNSUPPORTED;
    if (status) atomicMin(&a.status[img], jpeg_status_key(first_mcu, status));
    else if (sink.wide) atomicMin(&a.status[img], jpeg_status_key(kJpegStatusLast, IPX_ERR_UNSUPPORTED));
}

}  // namespace

int jpeg_par_sub_bytes() { return kSub; }
int jpeg_par_checkpoints() { return kCk; }

hipError_t launch_jpeg_pieces(const JpegDecArgs &a, uint8_t *ublob, uint32_t *ulen, hipStream_t s)
{
    if (a.nitems <= 0) return hipSuccess;
    hipLaunchKernelGGL(piece_unstuff_kernel, dim3((a.nitems + 255) / 256), dim3(256), 0, s, a, ublob, ulen);
    hipLaunchKernelGGL(piece_decode_kernel, dim3((a.nitems + 63) / 64), dim3(64), 0, s, a, (const uint8_t *)ublob, (const uint32_t *)ulen);
    return hipGetLastError();
}

hipError_t launch_par_count(const JpegParArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(par_count_kernel, dim3((a.max_nsub + 255) / 256, a.nimg), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_par_unstuff(const JpegParArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(par_unstuff_kernel, dim3((a.max_nsub + 255) / 256, a.nimg), dim3(256), 0, s, a);
    return hipGetLastError();
}
hipError_t launch_par_sync(const JpegParArgs &a, int round, hipStream_t s)
{
    static KernelLaunchCache sync_cache, write_cache;
    hipError_t e = sync_cache.prepare((const void *)par_sync_kernel, 64, kParLds, nullptr);
    if (e == hipSuccess) e = write_cache.prepare((const void *)par_write_kernel, 64, kParLds, nullptr);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(par_sync_kernel, dim3((a.max_nsub + 63) / 64, a.nimg), dim3(64), a.stage_rows == 1 ? par_lds(a.sub) : 4096, s, a, round);
    return hipGetLastError();
}
hipError_t launch_par_write(const JpegParArgs &a, hipStream_t s)
{
    // dynamic LDS beyond the 4 KiB of tables only limits how many waves share a CU: the lanes' scattered 2-byte coefficient stores keep one
    // 128-byte line each in flight, and with 2048 lanes per CU those lines do not fit the XCD's 4 MiB L2 -- every line is then written
    // back (and read for the merge) several times
    static const int write_lds = [] { const char *e = getenv("IPX_JPEG_WRITE_LDS"); return e ? atoi(e) : 16384; }();   // measured per 1024 x 1080p files: 4 KiB 8.7 ms, 10 KiB 8.4, 16 KiB 7.9, 20 KiB 8.1, 40 KiB 11.7
    hipLaunchKernelGGL(par_write_kernel, dim3((a.max_nsub + 63) / 64, a.nimg), dim3(64), a.stage_rows == 1 ? std::max(par_lds(a.sub), (size_t)write_lds) : (size_t)std::max(4096, write_lds), s, a);
    return hipGetLastError();
}
hipError_t launch_par_dc(const JpegParArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(par_dc_kernel, dim3(a.nimg), dim3(kDcThreads), 0, s, a);
    return hipGetLastError();
}

}  // namespace ipx

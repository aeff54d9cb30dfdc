// ipx_jpeg_entropy.hip -- the entropy-coding half of jpeg.Encode on the GPU.
//
// After ipx_jpeg.hip the quantised coefficients of a batch sit in HBM (6.3 MB per 1080p frame); shipping them to
// the host costs more PCIe time than the pixels' worth and the host's Huffman loop runs at ~250 frames/s per core
// (tools/bench_jpeg.py), five hundred times slower than the transform.  So the scan is coded where the data is
// and only the finished streams (~0.3 MB per 1080p frame) cross the link.
//
// What Go's writer does serially (image/jpeg/writer.go: writeBlock, emitHuffRLE, emit) is a pure function of the
// coefficients once three dependencies are cut:
//   * the DC delta needs the previous block of the same component   -> read from the coefficient array;
//   * a block's bits start where the previous block's end           -> pass A sizes every block, an exclusive
//                                                                      scan places it, pass B writes it there;
//   * emit() stuffs a 0x00 after every 0xff byte                    -> count per 256-byte chunk, scan, pass D
//                                                                      copies each chunk to its stuffed place.
// Pass A / B: one thread per 8x8 block; the workgroup's 256 blocks are staged in LDS (block stride 33 words, so
// that 64 lanes reading "their" coefficient hit 64 different banks), a 64-bit non-zero mask is built and only the
// non-zero coefficients are visited (ffs), which is what keeps lanes from idling through zero runs.  Codes come
// from an LDS copy of the four Annex K tables (len << 16 | code).  Pass B ORs big-endian words into a zeroed
// buffer with atomics: words shared by neighbouring blocks need no ownership rule.
// The padding of writeSOS's final emit(0x7f, 7) -- ones up to the byte boundary -- is written by the frame's last
// block.  Pass D also places the stream header (SOI .. SOS, built on the host) and EOI.
#include "ipx_internal.h"

namespace ipx {

namespace {

constexpr int kBlkWords = 33;   // LDS words per staged block (32 + 1)

struct HuffLds { uint32_t t[4][256]; };

__device__ __forceinline__ void load_tables(HuffLds &h, const uint32_t *g)
{
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) (&h.t[0][0])[i] = g[i];
}

// stage the 256 (or fewer) blocks of this workgroup: coalesced 16-byte loads, 33-word stride in LDS
__device__ __forceinline__ void stage_blocks(uint32_t *lds, const int16_t *coefs, long long first, long long nblk_total)
{
    const uint4 *src = (const uint4 *)(coefs + first * 64);
    const int chunks = (int)min((long long)256, nblk_total - first) * 8;   // 16-byte chunks, 8 per block
    for (int c = threadIdx.x; c < chunks; c += blockDim.x) {
        const uint4 v = src[c];
        uint32_t *d = lds + (c >> 3) * kBlkWords + (c & 7) * 4;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
}

__device__ __forceinline__ int coef_at(const uint32_t *blk, int zig)
{
    const uint32_t wv = blk[zig >> 1];
    return (int)(int16_t)((zig & 1) ? (wv >> 16) : (wv & 0xffffu));
}

// writeBlock's symbol sequence for one block; sink(code, nbits) per symbol part
template <class Sink>
__device__ __forceinline__ void code_block(const uint32_t *blk, int prev_dc, int q, const HuffLds &h, Sink &sink)
{
    auto rle = [&](const uint32_t *tab, int run, int value) {   // emitHuffRLE
        const int a = value < 0 ? -value : value, b = value < 0 ? value - 1 : value;
        const int nb = a ? 32 - __clz(a) : 0;
        const uint32_t e = tab[(run << 4 | nb) & 255];
        // Huffman code, then the low nb bits of b: one sink call, at most 16 + 11 bits
        sink((e & 0xffffu) << nb | ((uint32_t)b & ((1u << nb) - 1)), (e >> 16) + nb);
    };
    unsigned long long mask = 0;
#pragma unroll
    for (int i = 0; i < 32; i++) {
        const uint32_t wv = blk[i];
        mask |= (unsigned long long)((wv & 0xffffu) != 0) << (2 * i);
        mask |= (unsigned long long)((wv >> 16) != 0) << (2 * i + 1);
    }
    rle(h.t[2 * q], 0, coef_at(blk, 0) - prev_dc);
    const uint32_t *ac = h.t[2 * q + 1];
    mask &= ~1ull;
    int prev = 0;
    while (mask) {
        const int zig = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        int run = zig - prev - 1;
        prev = zig;
        while (run > 15) { sink(ac[0xf0] & 0xffffu, ac[0xf0] >> 16); run -= 16; }
        rle(ac, run, coef_at(blk, zig));
    }
    if (prev != 63) sink(ac[0] & 0xffffu, ac[0] >> 16);
}

// DC of the previous block of the same component (0 at the start of the scan)
__device__ __forceinline__ int prev_dc_of(const int16_t *frame_coefs, int b)
{
    const int m = b / 6, j = b - m * 6;
    int p;
    if (j == 0) p = m > 0 ? (m - 1) * 6 + 3 : -1;
    else if (j < 4) p = b - 1;
    else p = m > 0 ? b - 6 : -1;
    return p < 0 ? 0 : (int)frame_coefs[(size_t)p * 64];
}

struct LenSink {
    uint32_t bits = 0;
    __device__ __forceinline__ void operator()(uint32_t, uint32_t n) { bits += n; }
};

// pass A: bit length of every block
__global__ __launch_bounds__(256) void jpeg_len_kernel(const int16_t *coefs, int nblk, const uint32_t *tables, uint32_t *len)
{
    __shared__ HuffLds h;
    __shared__ uint32_t lds[256 * kBlkWords];
    const int f = blockIdx.y;
    const int16_t *fc = coefs + (size_t)f * nblk * 64;
    const int first = blockIdx.x * 256;
    load_tables(h, tables);
    stage_blocks(lds, fc, first, nblk);
    __syncthreads();
    const int b = first + threadIdx.x;
    if (b >= nblk) return;
    LenSink s;
    code_block(lds + threadIdx.x * kBlkWords, prev_dc_of(fc, b), (b % 6) < 4 ? 0 : 1, h, s);
    len[(size_t)f * nblk + b] = s.bits;
}

// pass A': the AC part of len[] comes from the transform kernel (ipx_jpeg.hip, step 2b); this adds the DC symbol
__global__ __launch_bounds__(256) void jpeg_dclen_kernel(const int16_t *dcq, int nblk, const uint32_t *tables, uint32_t *len)
{
    const int f = blockIdx.y, b = blockIdx.x * 256 + threadIdx.x;
    if (b >= nblk) return;
    const int16_t *fd = dcq + (size_t)f * nblk;
    const int m = b / 6, j = b - m * 6;
    int p;                                                   // the previous block of the same component (prev_dc_of)
    if (j == 0) p = m > 0 ? (m - 1) * 6 + 3 : -1;
    else if (j < 4) p = b - 1;
    else p = m > 0 ? b - 6 : -1;
    const int diff = (int)fd[b] - (p < 0 ? 0 : (int)fd[p]);
    const int a = diff < 0 ? -diff : diff;
    const int nb = a ? 32 - __clz(a) : 0;
    len[(size_t)f * nblk + b] += (tables[(j < 4 ? 0 : 2) * 256 + nb] >> 16) + (uint32_t)nb;
}

struct BitSink {
    uint32_t *out;          // big-endian words of the frame's unstuffed stream
    unsigned long long acc = 0;
    uint32_t cnt, widx;
    __device__ __forceinline__ BitSink(uint32_t *o, unsigned long long pos) : out(o), cnt((uint32_t)(pos & 31)), widx((uint32_t)(pos >> 5)) {}
    __device__ __forceinline__ void operator()(uint32_t code, uint32_t n)
    {
        acc = (acc << n) | code;
        cnt += n;
        if (cnt >= 32) {
            atomicOr(out + widx, __builtin_bswap32((uint32_t)(acc >> (cnt - 32))));
            widx++;
            cnt -= 32;
        }
    }
    __device__ __forceinline__ void flush()
    {
        if (cnt > 0) atomicOr(out + widx, __builtin_bswap32((uint32_t)(acc << (32 - cnt))));
    }
};

// pass B: every block's bits to their place (off = exclusive scan of len); the last block pads with ones
__global__ __launch_bounds__(256) void jpeg_bits_kernel(const int16_t *coefs, int nblk, const uint32_t *tables, const uint32_t *off,
                                                        const uint32_t *total_bits, const unsigned long long *ubase, uint8_t *ustream)
{
    __shared__ HuffLds h;
    __shared__ uint32_t lds[256 * kBlkWords];
    const int f = blockIdx.y;
    const int16_t *fc = coefs + (size_t)f * nblk * 64;
    const int first = blockIdx.x * 256;
    load_tables(h, tables);
    stage_blocks(lds, fc, first, nblk);
    __syncthreads();
    const int b = first + threadIdx.x;
    if (b >= nblk) return;
    BitSink s((uint32_t *)(ustream + ubase[f]), off[(size_t)f * nblk + b]);
    code_block(lds + threadIdx.x * kBlkWords, prev_dc_of(fc, b), (b % 6) < 4 ? 0 : 1, h, s);
    if (b == nblk - 1) {
        const uint32_t pad = (8 - (total_bits[f] & 7)) & 7;   // emit(0x7f, 7): only the bits that complete a byte get out
        if (pad) s((1u << pad) - 1, pad);
    }
    s.flush();
}

// exclusive scan of one frame's array per workgroup, in place; sum -> total[frame].
// Tiles of 4096 values, four consecutive ones per thread: wave scan by shuffles, the sixteen wave sums through LDS, the next tile's
// loads in flight while this one is scanned.  (The first form gave every thread n / 1024 CONSECUTIVE values and walked them twice, one
// dependent strided load after the other: 78 us for the 48 600 blocks of one 1080p frame -- a sixth of a small call's encode.)
__global__ __launch_bounds__(1024) void scan_kernel(uint32_t *v, int n, uint32_t *total)
{
    __shared__ uint32_t wsum[2][16];
    uint32_t *a = v + (size_t)blockIdx.x * n;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    auto load = [&](int i, uint32_t (&x)[4]) {
        for (int k = 0; k < 4; k++) x[k] = i + k < n ? a[i + k] : 0u;
    };
    uint32_t carry = 0, nxt[4];
    load(4 * t, nxt);
    int flip = 0;
    for (int base = 0; base < n; base += 4096, flip ^= 1) {
        const int i = base + 4 * t;
        const uint32_t x0 = nxt[0], x1 = nxt[1], x2 = nxt[2], x3 = nxt[3];
        if (base + 4096 < n) load(i + 4096, nxt);
        const uint32_t mine = x0 + x1 + x2 + x3;
        uint32_t inc = mine;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(inc, d);
            if (lane >= d) inc += y;
        }
        if (lane == 63) wsum[flip][wave] = inc;
        __syncthreads();                       // (two copies of wsum: a fast wave may write the next tile's sum while a slow one still reads this tile's)
        uint32_t before = 0, all = 0;
        for (int w = 0; w < 16; w++) { const uint32_t q = wsum[flip][w]; before += w < wave ? q : 0u; all += q; }
        uint32_t run = carry + before + inc - mine;
        if (i < n) a[i] = run;
        run += x0; if (i + 1 < n) a[i + 1] = run;
        run += x1; if (i + 2 < n) a[i + 2] = run;
        run += x2; if (i + 3 < n) a[i + 3] = run;
        carry += all;
    }
    if (t == 0) total[blockIdx.x] = carry;
}

constexpr int kChunk = 64;    // unstuffed bytes per thread in the stuffing passes

// pass C: 0xff bytes per chunk of the unstuffed stream
__global__ __launch_bounds__(256) void jpeg_ffcount_kernel(const uint8_t *ustream, const unsigned long long *ubase, const uint32_t *ubytes,
                                                           int max_chunks, uint32_t *ffcount)
{
    const int f = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= max_chunks) return;
    const uint32_t nb = ubytes[f];
    uint32_t cnt = 0;
    const uint32_t b0 = (uint32_t)c * kChunk;
    if (b0 < nb) {
        const uint32_t *p = (const uint32_t *)(ustream + ubase[f] + b0);   // chunks are word aligned; the tail word is zero padded
        const uint32_t words = (min(nb - b0, (uint32_t)kChunk) + 3) >> 2;
        for (uint32_t i = 0; i < words; i++) {
            const uint32_t wv = p[i];
            cnt += ((wv & 0xffu) == 0xffu) + (((wv >> 8) & 0xffu) == 0xffu) + (((wv >> 16) & 0xffu) == 0xffu) + ((wv >> 24) == 0xffu);
        }
    }
    ffcount[(size_t)f * max_chunks + c] = cnt;
}

// pass D: header, stuffed scan, EOI.  A thread stuffs its 64-byte chunk into LDS, at the place its output has inside the workgroup's
// (contiguous) piece of the stream, shifted so that 16-byte boundaries of the LDS buffer are 16-byte boundaries of the stream; the
// piece then goes out in whole 16-byte stores, byte stores only for its two ragged ends (the neighbouring workgroups write the bytes
// next to them).  The first version stored every byte from the chunk loop: 470 M one-byte stores per 1024 x 3 streams, 1.9 ms.
__global__ __launch_bounds__(256) void jpeg_stuff_kernel(const uint8_t *ustream, const unsigned long long *ubase, const uint32_t *ubytes,
                                                         int max_chunks, const uint32_t *ffoff, const uint8_t *header, int hdr_len,
                                                         const unsigned long long *obase, uint8_t *ostream)
{
    __shared__ __attribute__((aligned(16))) uint8_t buf[256 * 2 * kChunk + 32];
    __shared__ uint32_t piece_end;
    const int f = blockIdx.y, t = threadIdx.x, c = blockIdx.x * 256 + t;
    uint8_t *o = ostream + obase[f];
    if (blockIdx.x == 0)
        for (int i = t; i < hdr_len; i += 256) o[i] = header[i];
    const uint32_t nb = ubytes[f], w0 = (uint32_t)blockIdx.x * 256u * kChunk;          // first input byte of the workgroup
    if (w0 >= nb || (int)(blockIdx.x * 256) >= max_chunks) return;                     // uniform
    const size_t out0 = (size_t)hdr_len + w0 + ffoff[(size_t)f * max_chunks + blockIdx.x * 256];   // where the workgroup's piece starts in the frame's stream
    const uint32_t phase = (uint32_t)((uintptr_t)(o + out0) & 15u);
    if (t == 0) piece_end = 0;
    __syncthreads();
    const uint32_t b0 = (uint32_t)c * kChunk;
    if (c < max_chunks && b0 < nb) {
        const uint32_t *p = (const uint32_t *)(ustream + ubase[f] + b0);   // word aligned; the tail word is zero padded
        const uint32_t n = min(nb - b0, (uint32_t)kChunk);
        uint32_t at = phase + (b0 - w0) + (ffoff[(size_t)f * max_chunks + c] - ffoff[(size_t)f * max_chunks + blockIdx.x * 256]);
        uint32_t wv[kChunk / 4];
#pragma unroll
        for (int i = 0; i < kChunk / 4; i++) wv[i] = (uint32_t)i * 4 < n ? p[i] : 0u;
#pragma unroll
        for (int i = 0; i < kChunk / 4; i++) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if ((uint32_t)(i * 4 + k) < n) {
                    const uint8_t v = (uint8_t)(wv[i] >> (8 * k));
                    buf[at++] = v;
                    if (v == 0xff) buf[at++] = 0x00;
                }
            }
        }
        if (b0 + n == nb) { buf[at++] = 0xff; buf[at++] = 0xd9; }   // EOI
        if (b0 + n == nb || t == 255) piece_end = at;                // the last chunk of the piece (chunks are consecutive: exactly one thread)
    }
    __syncthreads();
    const uint32_t end = piece_end;                                   // buffer offset one past the piece
    uint8_t *g = o + out0 - phase;                                    // 16-byte aligned; buffer offset i <-> g + i
    for (uint32_t k = (uint32_t)t * 16u; k < end; k += 256u * 16u) {
        if (k >= phase && k + 16 <= end) *(uint4 *)(g + k) = *(const uint4 *)(buf + k);
        else
            for (uint32_t i = max(k, phase); i < min(k + 16, end); i++) g[i] = buf[i];
    }
}

}  // namespace

hipError_t launch_jpeg_dclen(const int16_t *dcq, int nblk, int n, const uint32_t *tables, uint32_t *len, hipStream_t s)
{
    hipLaunchKernelGGL(jpeg_dclen_kernel, dim3((nblk + 255) / 256, n), dim3(256), 0, s, dcq, nblk, tables, len);
    return hipGetLastError();
}

hipError_t launch_jpeg_len(const int16_t *coefs, int nblk, int n, const uint32_t *tables, uint32_t *len, hipStream_t s)
{
    hipLaunchKernelGGL(jpeg_len_kernel, dim3((nblk + 255) / 256, n), dim3(256), 0, s, coefs, nblk, tables, len);
    return hipGetLastError();
}
hipError_t launch_jpeg_bits(const int16_t *coefs, int nblk, int n, const uint32_t *tables, const uint32_t *off, const uint32_t *total_bits,
                            const unsigned long long *ubase, uint8_t *ustream, hipStream_t s)
{
    hipLaunchKernelGGL(jpeg_bits_kernel, dim3((nblk + 255) / 256, n), dim3(256), 0, s, coefs, nblk, tables, off, total_bits, ubase, ustream);
    return hipGetLastError();
}
hipError_t launch_scan(uint32_t *v, int per_frame, int n, uint32_t *total, hipStream_t s)
{
    hipLaunchKernelGGL(scan_kernel, dim3(n), dim3(1024), 0, s, v, per_frame, total);
    return hipGetLastError();
}
int jpeg_chunk_bytes() { return kChunk; }
hipError_t launch_jpeg_ffcount(const uint8_t *ustream, const unsigned long long *ubase, const uint32_t *ubytes, int max_chunks, int n,
                               uint32_t *ffcount, hipStream_t s)
{
    hipLaunchKernelGGL(jpeg_ffcount_kernel, dim3((max_chunks + 255) / 256, n), dim3(256), 0, s, ustream, ubase, ubytes, max_chunks, ffcount);
    return hipGetLastError();
}
hipError_t launch_jpeg_stuff(const uint8_t *ustream, const unsigned long long *ubase, const uint32_t *ubytes, int max_chunks, int n,
                             const uint32_t *ffoff, const uint8_t *header, int hdr_len, const unsigned long long *obase, uint8_t *ostream,
                             hipStream_t s)
{
    hipLaunchKernelGGL(jpeg_stuff_kernel, dim3((max_chunks + 255) / 256, n), dim3(256), 0, s, ustream, ubase, ubytes, max_chunks, ffoff, header,
                       hdr_len, obase, ostream);
    return hipGetLastError();
}

}  // namespace ipx

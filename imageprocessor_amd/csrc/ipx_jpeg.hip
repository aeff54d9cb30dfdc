// ipx_jpeg.hip -- the transform half of jpeg.Encode on the GPU (encoder side).
//
// Every operator of the reference ends in jpeg.Encode(buf, img, &jpeg.Options{Quality: 85}) on the *image.RGBA it
// produced (operations/resize.go:80, thumbnail.go:70, watermark.go:68), and on the CPU worker that encode is
// where most of the time goes.  Go's encoder (image/jpeg/writer.go, Go 1.24 stdlib, go.mod:3)
// per 16x16 MCU: rgbaToYCbCr (color.RGBToYCbCr on the stored bytes, edge pixels replicated), scale() (2x2 box,
// (sum + 2) >> 2) for Cb / Cr, fdct (libjpeg's jfdctint, level shift inside, output x 8) and
// div(b[unzig[zig]], 8 * quant[zig]) rounding half away from zero.  This kernel does exactly that for a batch of
// frames resident in HBM and leaves the quantised coefficients, int16 in zig-zag order, 6 x 64 per MCU in scan
// order (Y0 Y1 Y2 Y3 Cb Cr); the entropy coder (ipx_jpeg_host.cpp) turns them into the byte stream.
//
// Work split: a workgroup of 256 threads takes 8 horizontally adjacent MCUs (128 x 16 pixels, 48 blocks).
//   1. thread t loads the 8 pixels of row t/16 of x-block t%16 (two 16-byte loads; a row of 16 threads reads 512
//      contiguous bytes), converts them, and already holds one ROW of a Y block: pass 1 of the DCT runs in
//      registers.  Chroma: horizontal pair sums in the thread, vertical sums by lane ^ 16 (rows 2y / 2y+1 sit in
//      the same wave), the two half rows of an MCU's chroma row meet by lane ^ 1; even-row threads then run
//      pass 1 for Cb (even x-block) or Cr (odd x-block).  Pass-1 rows go to LDS (block stride 72 ints: no more
//      than the unavoidable 2-way bank conflict for the column reads).
//   2. barrier; 384 column tasks (48 blocks x 8 columns) run pass 2 from LDS, quantise with an exact
//      multiply-high reciprocal, and scatter the int16 results in zig-zag order into an LDS image of the output;
//   3. barrier; the 6 KiB of the workgroup leave with coalesced 16-byte stores.
// Bound: HBM (4 B/pixel in, 3 B/pixel out; ~60 integer ops per pixel).
#include "ipx_internal.h"

namespace ipx {

namespace {

constexpr int kMcuPerWg = 8;
constexpr int kBlkStride = 72;   // ints per block in LDS (64 + 8: column reads of 8 blocks spread over all banks)

__constant__ uint8_t c_zig_of_natural[64] = {   // natural index -> position in the zig-zag sequence
    0, 1, 5, 6, 14, 15, 27, 28, 2, 4, 7, 13, 16, 26, 29, 42, 3, 8, 12, 17, 25, 30, 41, 43, 9, 11, 18, 24, 31, 40, 44, 53,
    10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};

constexpr int FIX_0_298631336 = 2446, FIX_0_390180644 = 3196, FIX_0_541196100 = 4433, FIX_0_765366865 = 6270;
constexpr int FIX_0_899976223 = 7373, FIX_1_175875602 = 9633, FIX_1_501321110 = 12299, FIX_1_847759065 = 15137;
constexpr int FIX_1_961570560 = 16069, FIX_2_053119869 = 16819, FIX_2_562915447 = 20995, FIX_3_072711026 = 25172;
constexpr int kConstBits = 13, kPass1Bits = 2;

// one 8-point pass of fdct.go; PASS2 = column pass (descale by kPass1Bits more, rounding folded into tmp10)
template <bool PASS2>
__device__ __forceinline__ void fdct8(int (&s)[8])
{
    const int x0 = s[0], x1 = s[1], x2 = s[2], x3 = s[3], x4 = s[4], x5 = s[5], x6 = s[6], x7 = s[7];
    int tmp0 = x0 + x7, tmp1 = x1 + x6, tmp2 = x2 + x5, tmp3 = x3 + x4;
    int tmp10 = tmp0 + tmp3 + (PASS2 ? 1 << (kPass1Bits - 1) : 0), tmp12 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp13 = tmp1 - tmp2;
    tmp0 = x0 - x7; tmp1 = x1 - x6; tmp2 = x2 - x5; tmp3 = x3 - x4;
    constexpr int sh = PASS2 ? kConstBits + kPass1Bits : kConstBits - kPass1Bits;
    if (PASS2) { s[0] = (tmp10 + tmp11) >> kPass1Bits; s[4] = (tmp10 - tmp11) >> kPass1Bits; }
    else { s[0] = (tmp10 + tmp11 - 8 * 128) << kPass1Bits; s[4] = (tmp10 - tmp11) << kPass1Bits; }
    int z1 = (tmp12 + tmp13) * FIX_0_541196100;
    z1 += 1 << (sh - 1);
    s[2] = (z1 + tmp12 * FIX_0_765366865) >> sh;
    s[6] = (z1 - tmp13 * FIX_1_847759065) >> sh;
    tmp10 = tmp0 + tmp3; tmp11 = tmp1 + tmp2; tmp12 = tmp0 + tmp2; tmp13 = tmp1 + tmp3;
    z1 = (tmp12 + tmp13) * FIX_1_175875602;
    z1 += 1 << (sh - 1);
    tmp0 *= FIX_1_501321110; tmp1 *= FIX_3_072711026; tmp2 *= FIX_2_053119869; tmp3 *= FIX_0_298631336;
    tmp10 *= -FIX_0_899976223; tmp11 *= -FIX_2_562915447; tmp12 *= -FIX_0_390180644; tmp13 *= -FIX_1_961570560;
    tmp12 += z1; tmp13 += z1;
    s[1] = (tmp0 + tmp10 + tmp12) >> sh;
    s[3] = (tmp1 + tmp11 + tmp13) >> sh;
    s[5] = (tmp2 + tmp11 + tmp12) >> sh;
    s[7] = (tmp3 + tmp10 + tmp13) >> sh;
}

// color.RGBToYCbCr on one stored pixel (alpha ignored, as rgbaToYCbCr does)
__device__ __forceinline__ void rgb_to_ycc(uint32_t px, int &yy, int &cb, int &cr)
{
    const int r = (int)(px & 0xffu), g = (int)((px >> 8) & 0xffu), b = (int)((px >> 16) & 0xffu);
    yy = (19595 * r + 38470 * g + 7471 * b + (1 << 15)) >> 16;
    // Go: if uint32(c)&0xff000000 == 0 { c >>= 16 } else { c = ^(c >> 31) }, then uint8(c): clamp(c >> 16, 0, 255)
    cb = min(max((-11056 * r - 21712 * g + 32768 * b + (257 << 15)) >> 16, 0), 255);
    cr = min(max((32768 * r - 27440 * g - 5328 * b + (257 << 15)) >> 16, 0), 255);
}

__global__ __launch_bounds__(256) void jpeg_fdct_kernel(JpegArgs a)
{
    __shared__ int ws[48 * kBlkStride];
    __shared__ __attribute__((aligned(16))) int16_t so[48 * 64];
    __shared__ uint8_t aclen[2][256];       // code lengths of the two AC tables (only when the lengths are wanted)
    const int t = threadIdx.x;
    if (a.aclen)
        for (int i = t; i < 512; i += 256) aclen[i >> 8][i & 255] = (uint8_t)(a.huff[(2 * (i >> 8) + 1) * 256 + (i & 255)] >> 16);
    const int r = t >> 4, xb = t & 15;
    const int mcu0 = blockIdx.x * kMcuPerWg;
    const int x0 = mcu0 * 16 + xb * 8, y = blockIdx.y * 16 + r;
    const uint8_t *frame = a.src + (size_t)blockIdx.z * a.frame_stride;

    // ---- 1. load + colour conversion -------------------------------------------------------------------
    uint32_t px[8];
    const bool inside = mcu0 * 16 + 128 <= a.w && blockIdx.y * 16 + 16 <= a.h;   // uniform over the workgroup
    if (inside && a.aligned16) {
        const uint4 *p = (const uint4 *)(frame + (size_t)y * a.stride + (size_t)x0 * 4);
        const uint4 v0 = p[0], v1 = p[1];
        px[0] = v0.x; px[1] = v0.y; px[2] = v0.z; px[3] = v0.w; px[4] = v1.x; px[5] = v1.y; px[6] = v1.z; px[7] = v1.w;
    } else {
        const int sy = min(y, a.h - 1);   // rgbaToYCbCr: edge pixels replicated
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int sx = min(x0 + i, a.w - 1);
            px[i] = *(const uint32_t *)(frame + (size_t)sy * a.stride + (size_t)sx * 4);
        }
    }
    int yy[8], cb[8], cr[8];
#pragma unroll
    for (int i = 0; i < 8; i++) rgb_to_ycc(px[i], yy[i], cb[i], cr[i]);

    // Y: this thread holds row (r & 7) of Y block (r >> 3) * 2 + (xb & 1) of MCU xb >> 1
    fdct8<false>(yy);
    {
        const int blk = (xb >> 1) * 4 + (r >> 3) * 2 + (xb & 1);
        int *w = ws + blk * kBlkStride + (r & 7) * 8;
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = yy[i];
    }
    // chroma: scale() = (c[2y][2x] + c[2y][2x+1] + c[2y+1][2x] + c[2y+1][2x+1] + 2) >> 2
    int hb[4], hr[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        hb[i] = cb[2 * i] + cb[2 * i + 1];
        hr[i] = cr[2 * i] + cr[2 * i + 1];
        hb[i] = (hb[i] + __shfl_xor(hb[i], 16) + 2) >> 2;   // row r ^ 1 of the same x-block
        hr[i] = (hr[i] + __shfl_xor(hr[i], 16) + 2) >> 2;
    }
    // even x-blocks gather the Cb row of their MCU, odd x-blocks the Cr row: exchange the halves not kept
    int c8[8];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int give = (xb & 1) ? hb[i] : hr[i];
        const int got = __shfl_xor(give, 1);
        if (xb & 1) { c8[i] = got; c8[4 + i] = hr[i]; }     // Cr: left half from the even neighbour
        else { c8[i] = hb[i]; c8[4 + i] = got; }            // Cb: right half from the odd neighbour
    }
    if ((r & 1) == 0) {
        fdct8<false>(c8);
        const int blk = 32 + (xb & 1) * 8 + (xb >> 1);
        int *w = ws + blk * kBlkStride + (r >> 1) * 8;
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = c8[i];
    }
    __syncthreads();

    // ---- 2. column pass + quantisation -------------------------------------------------------------------
    for (int k = t; k < 48 * 8; k += 256) {
        const int blk = k >> 3, col = k & 7;
        const int q = blk < 32 ? 0 : 1;
        int s[8];
#pragma unroll
        for (int i = 0; i < 8; i++) s[i] = ws[blk * kBlkStride + i * 8 + col];
        fdct8<true>(s);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int nat = i * 8 + col;
            const uint32_t d = a.div8[q][nat];               // 8 * quant, natural order
            const int v = s[i];
            const uint32_t mag = (uint32_t)(v < 0 ? -v : v) + (d >> 1);
            const int qv = (int)__umulhi(mag, a.recip[q][nat]);   // exact floor(mag / d) for mag < 2^20 (checked on the host)
            so[blk * 64 + c_zig_of_natural[nat]] = (int16_t)(v < 0 ? -qv : qv);
        }
    }
    __syncthreads();

    // ---- 2b. one thread per block: bits its AC symbols will take (writeBlock's run-length coding, sized only), and its DC term.
    // 48 lanes of the first wave: spreading them over the four waves was measured slower (1.9 against 1.45 ms per 256 frames) ----
    if (a.aclen && t < 48) {
        const int m = t / 6, j = t - m * 6;
        const int mw2 = (a.w + 15) >> 4;
        if (mcu0 + m < mw2) {
            const int blk = j < 4 ? m * 4 + j : (j == 4 ? 32 + m : 40 + m);
            const int q = j < 4 ? 0 : 1;
            const uint32_t *bw = (const uint32_t *)(so + blk * 64);
            unsigned long long mask = 0;
#pragma unroll
            for (int i = 0; i < 32; i++) {
                const uint32_t wv = bw[i];
                mask |= (unsigned long long)((wv & 0xffffu) != 0) << (2 * i);
                mask |= (unsigned long long)((wv >> 16) != 0) << (2 * i + 1);
            }
            mask &= ~1ull;
            uint32_t bits = 0;
            int prev = 0;
            while (mask) {
                const int zig = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const int run = zig - prev - 1;
                prev = zig;
                const int v = so[blk * 64 + zig];
                const int nb = 32 - __clz(v < 0 ? -v : v);
                bits += (uint32_t)(run >> 4) * aclen[q][0xf0] + aclen[q][((run & 15) << 4 | nb) & 255] + (uint32_t)nb;
            }
            if (prev != 63) bits += aclen[q][0];
            const size_t gb = ((size_t)blockIdx.z * a.mcus_per_frame + (size_t)blockIdx.y * mw2 + mcu0) * 6 + t;
            a.aclen[gb] = bits;
            a.dcq[gb] = so[blk * 64];
        }
    }

    // ---- 3. coalesced store: MCU m of this workgroup = Y blocks 4m..4m+3, Cb block 32+m, Cr block 40+m ----
    const int mw = (a.w + 15) >> 4;
    int16_t *out = a.coefs + ((size_t)blockIdx.z * a.mcus_per_frame + (size_t)blockIdx.y * mw + mcu0) * 384;
    for (int c = t; c < 48 * 8; c += 256) {               // 16-byte chunks: 8 per block
        const int ob = c >> 3, part = c & 7;                // ob = output block index within the workgroup: m * 6 + j
        const int m = ob / 6, j = ob - m * 6;
        if (mcu0 + m >= mw) continue;
        const int blk = j < 4 ? m * 4 + j : (j == 4 ? 32 + m : 40 + m);
        *(uint4 *)(out + ob * 64 + part * 8) = *(const uint4 *)(so + blk * 64 + part * 8);
    }
}

}  // namespace

hipError_t launch_jpeg_fdct(const JpegArgs &a, int n, hipStream_t s)
{
    const int mw = (a.w + 15) >> 4, mh = (a.h + 15) >> 4;
    dim3 grid((mw + kMcuPerWg - 1) / kMcuPerWg, mh, n);
    hipLaunchKernelGGL(jpeg_fdct_kernel, grid, dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace ipx

// ipx_kernels.hip -- gfx950 (CDNA4) kernels of the pixel path.  No MFMA: every kernel here is a
// streaming kernel bounded by HBM bandwidth (the scaler lives in ipx_ks_generic.hip / ipx_ks_fused.hip).
//
// Arithmetic contracts (restating what the reference's libraries compute):
//   * copy: image/draw drawCopySrc / drawCopyOver.
//   * glyph composite: image/draw drawGlyphOver, uint32 wrap-around preserved.
// The file is compiled with -ffp-contract=off and the pragma below; check with
// `llvm-objdump -d` that the scale kernels hold no v_fma_f64.
#include <algorithm>

#include "ipx_internal.h"

#pragma clang fp contract(off)

#include "ipx_device.h"

namespace ipx {

namespace {

// image.(*RGBA).Opaque over the whole source: *flag (preset to 1) is cleared by any alpha != 0xff
__global__ __launch_bounds__(256) void opaque_scan_kernel(const uint8_t *src, int sw, int sh,
                                                          int sstride, int *flag)
{
    const int y = blockIdx.y;
    bool bad = false;
    for (int x = blockIdx.x * 256 + threadIdx.x; x < sw; x += gridDim.x * 256)
        bad |= src[(size_t)y * sstride + (size_t)x * 4 + 3] != 0xff;
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicAnd(flag, 0);
}

// (*image.NRGBA64).Opaque / (*image.RGBA64).Opaque on the frame of taps: the alpha tap is the stored alpha for both types
__global__ __launch_bounds__(256) void opaque_scan_tap64_kernel(const uint8_t *src, int sw, int sh, int sstride, int *flag)
{
    const int y = blockIdx.y;
    bool bad = false;
    for (int x = blockIdx.x * 256 + threadIdx.x; x < sw; x += gridDim.x * 256)
        bad |= *(const uint16_t *)(src + (size_t)y * sstride + (size_t)x * 8 + 6) != 0xffffu;
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicAnd(flag, 0);
}

// ---------------------------------------------------------------------------------------------
// draw: DrawMask with a nil mask on pre-clipped rectangles
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void draw_src_vec_kernel(uint4 *__restrict__ dst, int dstride,
                                                           const uint4 *__restrict__ src,
                                                           int sstride, int wchunks, int h)
{
    const int y = blockIdx.y;
    if (y >= h) return;
    const uint4 *s = (const uint4 *)((const uint8_t *)src + (size_t)y * sstride);
    uint4 *d = (uint4 *)((uint8_t *)dst + (size_t)y * dstride);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < wchunks; i += gridDim.x * 256) d[i] = s[i];
}

__global__ __launch_bounds__(256) void draw_px_kernel(uint8_t *dst, int dstride, const uint8_t *src,
                                                      int sstride, int w, int h, int op)
{
    const int y = blockIdx.y;
    if (y >= h) return;
    for (int x = blockIdx.x * 256 + threadIdx.x; x < w; x += gridDim.x * 256) {
        const uint32_t s = *(const uint32_t *)(src + (size_t)y * sstride + (size_t)x * 4);
        uint32_t *dp = (uint32_t *)(dst + (size_t)y * dstride + (size_t)x * 4);
        if (op == IPX_OP_SRC) { *dp = s; continue; }
        // drawCopyOver: a = (m - sa)*0x101; d = uint8((d*a/m + s) >> 8), s widened by 0x101
        const uint32_t d = *dp;
        const uint32_t sa = (s >> 24) * 0x101u;
        const uint32_t al = (kM - sa) * 0x101u;
        const uint32_t r = (((d & 0xffu) * al / kM + (s & 0xffu) * 0x101u) >> 8) & 0xffu;
        const uint32_t g = ((((d >> 8) & 0xffu) * al / kM + ((s >> 8) & 0xffu) * 0x101u) >> 8) & 0xffu;
        const uint32_t b = ((((d >> 16) & 0xffu) * al / kM + ((s >> 16) & 0xffu) * 0x101u) >> 8) & 0xffu;
        const uint32_t a = (((d >> 24) * al / kM + sa) >> 8) & 0xffu;
        *dp = r | (g << 8) | (b << 16) | (a << 24);
    }
}

// drawNRGBASrc / drawNRGBAOver: premultiply each source pixel, then as drawCopySrc / drawCopyOver
__global__ __launch_bounds__(256) void draw_nrgba_kernel(uint8_t *dst, int dstride, const uint8_t *src,
                                                         int sstride, int w, int h, int op, size_t dst_fs, size_t src_fs)
{
    const int y = blockIdx.y;
    if (y >= h) return;
    dst += blockIdx.z * dst_fs; src += blockIdx.z * src_fs;   // frame of a batch
    for (int x = blockIdx.x * 256 + threadIdx.x; x < w; x += gridDim.x * 256) {
        const uint32_t s = *(const uint32_t *)(src + (size_t)y * sstride + (size_t)x * 4);
        uint32_t *dp = (uint32_t *)(dst + (size_t)y * dstride + (size_t)x * 4);
        const uint32_t sa = (s >> 24) * 0x101u;
        const uint32_t sr = (s & 0xffu) * sa / 0xffu, sg = ((s >> 8) & 0xffu) * sa / 0xffu, sb = ((s >> 16) & 0xffu) * sa / 0xffu;
        if (op == IPX_OP_SRC) { *dp = (sr >> 8) | (sg & 0xff00u) | ((sb & 0xff00u) << 8) | ((sa & 0xff00u) << 16); continue; }
        const uint32_t d = *dp;
        const uint32_t al = (kM - sa) * 0x101u;
        const uint32_t r = (((d & 0xffu) * al / kM + sr) >> 8) & 0xffu;
        const uint32_t g = ((((d >> 8) & 0xffu) * al / kM + sg) >> 8) & 0xffu;
        const uint32_t b = ((((d >> 16) & 0xffu) * al / kM + sb) >> 8) & 0xffu;
        const uint32_t a = (((d >> 24) * al / kM + sa) >> 8) & 0xffu;
        *dp = r | (g << 8) | (b << 16) | (a << 24);
    }
}

// The deep source types (*image.NRGBA64 / RGBA64 / Gray16 from 16-bit PNGs, *image.CMYK from four-component JPEGs): no routine of
// x/image/draw or image/draw specialises on them, every consumer reads src.At(x, y).RGBA().  One pass turns Go's Pix (big-endian
// 16-bit channels; C M Y K bytes) into those taps, four little-endian uint16 per pixel:
//   color.NRGBA64.RGBA  c * a / 0xffff, alpha a            color.RGBA64.RGBA  as stored
//   color.Gray16.RGBA   (y, y, y, 0xffff)                   color.CMYK.RGBA    w = 0xffff - k * 0x101; (0xffff - c * 0x101) * w / 0xffff, alpha 0xffff
__device__ __forceinline__ uint32_t be16x2(uint32_t v) { return __builtin_amdgcn_perm(0u, v, 0x02030001u); }   // two big-endian uint16 -> lo | hi << 16
template <int KIND>
__global__ __launch_bounds__(256) void deep_expand_kernel(uint8_t *dst, size_t dst_fs, const uint8_t *src, int sstride, size_t src_fs, int w, int h)
{
    const int y = blockIdx.y;
    dst += blockIdx.z * dst_fs; src += blockIdx.z * src_fs;
    const uint8_t *row = src + (size_t)y * sstride;
    uint2 *out = (uint2 *)(dst + (size_t)y * w * 8);
    for (int x = blockIdx.x * 256 + threadIdx.x; x < w; x += gridDim.x * 256) {
        uint2 o;
        if (KIND == IPX_DEEP_GRAY16) {
            const uint32_t v = (uint32_t)row[2 * x] << 8 | row[2 * x + 1];
            o.x = v | v << 16; o.y = v | 0xffff0000u;
        } else if (KIND == IPX_DEEP_CMYK) {
            const uint32_t p = *(const uint32_t *)(row + (size_t)x * 4);           // rows of 4-byte pixels: 4-aligned by the entry's check
            const uint32_t wk = 0xffffu - (p >> 24) * 0x101u;
            const uint32_t r = (0xffffu - (p & 0xffu) * 0x101u) * wk / 0xffffu, g = (0xffffu - ((p >> 8) & 0xffu) * 0x101u) * wk / 0xffffu,
                           b = (0xffffu - ((p >> 16) & 0xffu) * 0x101u) * wk / 0xffffu;
            o.x = r | g << 16; o.y = b | 0xffff0000u;
        } else {
            const uint16_t *p = (const uint16_t *)(row + (size_t)x * 8);           // (2-aligned by the entry's check)
            const uint32_t rg = be16x2((uint32_t)p[0] | (uint32_t)p[1] << 16), ba = be16x2((uint32_t)p[2] | (uint32_t)p[3] << 16);
            if (KIND == IPX_DEEP_NRGBA64) {
                const uint32_t a = ba >> 16;
                const uint32_t r = (rg & 0xffffu) * a / 0xffffu, g = (rg >> 16) * a / 0xffffu, b = (ba & 0xffffu) * a / 0xffffu;
                o.x = r | g << 16; o.y = b | a << 16;
            } else { o.x = rg; o.y = ba; }
        }
        out[x] = o;
    }
}

// image/draw drawRGBA with a nil mask (and drawCMYK, whose CMYKToRGB is the top byte of color.CMYK.RGBA with alpha 0xffff) on frames of
// taps: Src keeps the top byte of every tap; Over: a = (m - sa) * 0x101, d = uint8((d * a + s * m) / m >> 8), uint32 wrap-around as in Go
__global__ __launch_bounds__(256) void draw_tap64_kernel(uint8_t *dst, int dstride, const uint8_t *src, int sstride, int w, int h, int op,
                                                         size_t dst_fs, size_t src_fs)
{
    const int y = blockIdx.y;
    if (y >= h) return;
    dst += blockIdx.z * dst_fs; src += blockIdx.z * src_fs;
    for (int x = blockIdx.x * 256 + threadIdx.x; x < w; x += gridDim.x * 256) {
        const uint2 t = *(const uint2 *)(src + (size_t)y * sstride + (size_t)x * 8);
        uint32_t *dp = (uint32_t *)(dst + (size_t)y * dstride + (size_t)x * 4);
        if (op == IPX_OP_SRC) { *dp = __builtin_amdgcn_perm(t.y, t.x, 0x07050301u); continue; }
        const uint32_t d = *dp;
        const uint32_t sr = t.x & 0xffffu, sg = t.x >> 16, sb = t.y & 0xffffu, sa = t.y >> 16;
        const uint32_t al = (kM - sa) * 0x101u;
        const uint32_t r = ((((d & 0xffu) * al + sr * kM) / kM) >> 8) & 0xffu;
        const uint32_t g = (((((d >> 8) & 0xffu) * al + sg * kM) / kM) >> 8) & 0xffu;
        const uint32_t b = (((((d >> 16) & 0xffu) * al + sb * kM) / kM) >> 8) & 0xffu;
        const uint32_t a = ((((d >> 24) * al + sa * kM) / kM) >> 8) & 0xffu;
        *dp = r | (g << 8) | (b << 16) | (a << 24);
    }
}

// imageutil.DrawYCbCr: color.YCbCrToRGB per pixel (8 bit, the uint32 overflow test of the Go code), A = 255
__global__ __launch_bounds__(256) void draw_ycbcr_kernel(uint8_t *dst, int dstride, const uint8_t *yp, int ystride,
                                                         const uint8_t *cb, const uint8_t *cr, int cstride, int ratio,
                                                         int spx, int spy, int w, int h, size_t dst_fs, size_t y_fs,
                                                         size_t c_fs)
{
    const int y = blockIdx.y;
    if (y >= h) return;
    dst += blockIdx.z * dst_fs; yp += blockIdx.z * y_fs; cb += blockIdx.z * c_fs; cr += blockIdx.z * c_fs;
    for (int x = blockIdx.x * 256 + threadIdx.x; x < w; x += gridDim.x * 256) {
        const int sx = spx + x, sy = spy + y;
        const int cx = (ratio == IPX_YCBCR_422 || ratio == IPX_YCBCR_420) ? sx / 2 : sx;
        const int cy = (ratio == IPX_YCBCR_420 || ratio == IPX_YCBCR_440) ? sy / 2 : sy;
        const size_t ci = (size_t)cy * cstride + cx;
        const int32_t yy1 = (int32_t)yp[(size_t)sy * ystride + sx] * 0x10101;
        const int32_t cb1 = (int32_t)cb[ci] - 128, cr1 = (int32_t)cr[ci] - 128;
        int32_t r = yy1 + 91881 * cr1, g = yy1 - 22554 * cb1 - 46802 * cr1, b = yy1 + 116130 * cb1;
        r = ((uint32_t)r & 0xff000000u) == 0 ? r >> 16 : ~(r >> 31);
        g = ((uint32_t)g & 0xff000000u) == 0 ? g >> 16 : ~(g >> 31);
        b = ((uint32_t)b & 0xff000000u) == 0 ? b >> 16 : ~(b >> 31);
        *(uint32_t *)(dst + (size_t)y * dstride + (size_t)x * 4) =
            ((uint32_t)r & 0xffu) | (((uint32_t)g & 0xffu) << 8) | (((uint32_t)b & 0xffu) << 16) | 0xff000000u;
    }
}

// Stand-alone composite over the bounding box of the clipped glyph rectangles (the text pass after every band kernel, and the
// per-operation seam).  One wave covers a 64-pixel row segment.  The glyph table goes to LDS once per workgroup; the list is walked
// four glyphs at a time: rectangle tests from LDS, the mask bytes of the glyphs that hold the pixel loaded together, then the composites
// in string order -- a step costs one global latency instead of one per glyph (the first version loaded descriptor and mask per
// glyph and pixel, each a full round trip: 156 us per 1024 frames for a 16-glyph text).  A ballot skips steps no lane of the segment
// touches, and untouched pixels are never written.
constexpr int kCompositeRows = 4;   // pixels per thread (one column, 4 rows apart): the table load and the launch overhead of a block serve 16 rows
__global__ __launch_bounds__(256) void composite_kernel(uint8_t *dst, int dstride, size_t frame_stride,
                                                        const DevGlyph *__restrict__ gl, int n,
                                                        Rect bbox, uint32_t sr, uint32_t sg,
                                                        uint32_t sb, uint32_t sa)
{
    __shared__ DevGlyph tab[kMaxGlyphs];
    const int lt = (int)(threadIdx.y * 64 + threadIdx.x);
    const int x = bbox.x0 + (int)(blockIdx.x * 64 + threadIdx.x);
    const int ybase = bbox.y0 + (int)(blockIdx.y * (4 * kCompositeRows) + threadIdx.y);
    uint8_t *frame = dst + blockIdx.z * frame_stride;
    // the pixels first (their loads fly while the table arrives): a thread takes kCompositeRows pixels of one column, 4 rows apart
    uint32_t d[kCompositeRows], d0[kCompositeRows];
#pragma unroll
    for (int r = 0; r < kCompositeRows; r++) {
        const int y = ybase + 4 * r;
        d0[r] = d[r] = x < bbox.x1 && y < bbox.y1 ? *(const uint32_t *)(frame + (size_t)y * dstride + (size_t)x * 4) : 0u;
    }
    for (int g = lt; g < n; g += 256) tab[g] = gl[g];
    __syncthreads();
    for (int g = 0; g < n; g += 4) {
        uint32_t m[4][kCompositeRows];
        bool any = false;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const DevGlyph G = tab[min(g + j, n - 1)];
            const bool inx = g + j < n && x >= G.x0 && x < G.x1;
#pragma unroll
            for (int r = 0; r < kCompositeRows; r++) {
                const int y = ybase + 4 * r;
                const bool in = inx && y >= G.y0 && y < G.y1 && y < bbox.y1;
                m[j][r] = in ? G.mask[(size_t)(y - G.y0) * G.mstride + (x - G.x0)] : 0u;
                any |= in;
            }
        }
        if (!__any(any)) continue;  // wave-uniform skip
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < kCompositeRows; r++)
                if (m[j][r]) d[r] = glyph_over(d[r], m[j][r], sr, sg, sb, sa);
    }
#pragma unroll
    for (int r = 0; r < kCompositeRows; r++) {
        const int y = ybase + 4 * r;
        if (x < bbox.x1 && y < bbox.y1 && d[r] != d0[r]) *(uint32_t *)(frame + (size_t)y * dstride + (size_t)x * 4) = d[r];
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
hipError_t launch_draw_nrgba(uint8_t *dst, int dstride, const uint8_t *src, int sstride, int w, int h, int op,
                             hipStream_t s, int nframes, size_t dst_fs, size_t src_fs)
{
    if (w <= 0 || h <= 0 || nframes <= 0) return hipSuccess;
    dim3 grid(min(8, (w + 255) / 256), h, nframes);
    hipLaunchKernelGGL(draw_nrgba_kernel, grid, dim3(256), 0, s, dst, dstride, src, sstride, w, h, op, dst_fs, src_fs);
    return hipGetLastError();
}

hipError_t launch_deep_expand(uint8_t *dst, size_t dst_fs, const uint8_t *src, int sstride, size_t src_fs, int kind, int w, int h, int nframes, hipStream_t s)
{
    if (w <= 0 || h <= 0 || nframes <= 0) return hipSuccess;
    dim3 grid(min(2, (w + 255) / 256), h, nframes);   // a few pixels per thread: a workgroup per 256 pixels was launch-bound
    switch (kind) {
    case IPX_DEEP_NRGBA64: hipLaunchKernelGGL(deep_expand_kernel<IPX_DEEP_NRGBA64>, grid, dim3(256), 0, s, dst, dst_fs, src, sstride, src_fs, w, h); break;
    case IPX_DEEP_RGBA64: hipLaunchKernelGGL(deep_expand_kernel<IPX_DEEP_RGBA64>, grid, dim3(256), 0, s, dst, dst_fs, src, sstride, src_fs, w, h); break;
    case IPX_DEEP_GRAY16: hipLaunchKernelGGL(deep_expand_kernel<IPX_DEEP_GRAY16>, grid, dim3(256), 0, s, dst, dst_fs, src, sstride, src_fs, w, h); break;
    case IPX_DEEP_CMYK: hipLaunchKernelGGL(deep_expand_kernel<IPX_DEEP_CMYK>, grid, dim3(256), 0, s, dst, dst_fs, src, sstride, src_fs, w, h); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_draw_tap64(uint8_t *dst, int dstride, const uint8_t *src, int sstride, int w, int h, int op, hipStream_t s, int nframes,
                             size_t dst_fs, size_t src_fs)
{
    if (w <= 0 || h <= 0 || nframes <= 0) return hipSuccess;
    dim3 grid(min(2, (w + 255) / 256), h, nframes);   // a few pixels per thread: a workgroup per 256 pixels was launch-bound
    hipLaunchKernelGGL(draw_tap64_kernel, grid, dim3(256), 0, s, dst, dstride, src, sstride, w, h, op, dst_fs, src_fs);
    return hipGetLastError();
}

// image/draw drawGray (and scale_RGBA_Gray_Src's taps): a Gray source pixel is (y, y, y, 0xff)
__global__ __launch_bounds__(256) void gray_expand_kernel(uint8_t *dst, size_t dst_fs, const uint8_t *src, int sstride, size_t src_fs, int w, int h)
{
    const int y = blockIdx.y;
    dst += blockIdx.z * dst_fs; src += blockIdx.z * src_fs;
    const uint8_t *row = src + (size_t)y * sstride;
    uint32_t *out = (uint32_t *)(dst + (size_t)y * w * 4);
    for (int x = blockIdx.x * 256 + threadIdx.x; x < w; x += gridDim.x * 256) out[x] = (uint32_t)row[x] * 0x010101u | 0xff000000u;
}

hipError_t launch_gray_expand(uint8_t *dst, size_t dst_fs, const uint8_t *src, int sstride, size_t src_fs, int w, int h, int n, hipStream_t s)
{
    hipLaunchKernelGGL(gray_expand_kernel, dim3(std::min(8, (w + 255) / 256), h, n), dim3(256), 0, s, dst, dst_fs, src, sstride, src_fs, w, h);
    return hipGetLastError();
}

// *image.Paletted (GIF uploads, palette PNGs): no specialised routine upstream, the generic ones read Palette[i].RGBA() per tap / pixel.
// For the entries the decoders produce (opaque color.RGBA, the zero colour, color.NRGBA) that is the premultiplication the NRGBA
// routines apply to (R, G, B, A) itself, so the indices are expanded to NRGBA8 here and the frames take the NRGBA pass.
__global__ __launch_bounds__(256) void palette_expand_kernel(uint8_t *dst, size_t dst_fs, const uint8_t *src, int sstride, size_t src_fs,
                                                             const uint32_t *palettes, int w, int h, int rows_per_block, int vec)
{
    __shared__ uint32_t pal[256];
    pal[threadIdx.x] = palettes[(size_t)blockIdx.y * 256 + threadIdx.x];
    __syncthreads();
    dst += blockIdx.y * dst_fs; src += blockIdx.y * src_fs;
    const int y0 = blockIdx.x * rows_per_block, y1 = min(h, y0 + rows_per_block);
    for (int y = y0; y < y1; y++) {
        const uint8_t *row = src + (size_t)y * sstride;
        uint32_t *out = (uint32_t *)(dst + (size_t)y * w * 4);
        if (vec) {           // four indices per load, sixteen bytes per store
            for (int c = threadIdx.x; c < (w >> 2); c += 256) {
                const uint32_t i4 = ((const uint32_t *)row)[c];
                ((uint4 *)out)[c] = make_uint4(pal[i4 & 0xffu], pal[(i4 >> 8) & 0xffu], pal[(i4 >> 16) & 0xffu], pal[i4 >> 24]);
            }
        } else {
            for (int x = threadIdx.x; x < w; x += 256) out[x] = pal[row[x]];
        }
    }
}

hipError_t launch_palette_expand(uint8_t *dst, size_t dst_fs, const uint8_t *src, int sstride, size_t src_fs, const uint8_t *palettes, int w,
                                 int h, int n, hipStream_t s)
{
    const int rows = 8;      // per workgroup: the 1 KiB palette load is shared by 8 rows
    const int vec = !(w & 3) && !((((uintptr_t)src) | (uintptr_t)sstride | src_fs) & 3) && !((((uintptr_t)dst) | dst_fs) & 15);
    hipLaunchKernelGGL(palette_expand_kernel, dim3((h + rows - 1) / rows, n), dim3(256), 0, s, dst, dst_fs, src, sstride, src_fs,
                       (const uint32_t *)palettes, w, h, rows, vec);
    return hipGetLastError();
}

hipError_t launch_draw_ycbcr(uint8_t *dst, int dstride, const uint8_t *y, int ystride, const uint8_t *cb,
                             const uint8_t *cr, int cstride, int ratio, int spx, int spy, int w, int h, hipStream_t s,
                             int nframes, size_t dst_fs, size_t y_fs, size_t c_fs)
{
    if (w <= 0 || h <= 0 || nframes <= 0) return hipSuccess;
    dim3 grid(min(8, (w + 255) / 256), h, nframes);
    hipLaunchKernelGGL(draw_ycbcr_kernel, grid, dim3(256), 0, s, dst, dstride, y, ystride, cb, cr, cstride, ratio, spx,
                       spy, w, h, dst_fs, y_fs, c_fs);
    return hipGetLastError();
}

hipError_t launch_opaque_scan(const uint8_t *src, int sw, int sh, int sstride, int *flag,
                              hipStream_t s)
{
    hipError_t e = hipMemsetAsync(flag, 1, sizeof(int), s);  // any non-zero value means "opaque"
    if (e != hipSuccess) return e;
    if (sw <= 0 || sh <= 0) return hipSuccess;
    dim3 grid(min(8, (sw + 255) / 256), sh);
    hipLaunchKernelGGL(opaque_scan_kernel, grid, dim3(256), 0, s, src, sw, sh, sstride, flag);
    return hipGetLastError();
}

hipError_t launch_opaque_scan_tap64(const uint8_t *src, int sw, int sh, int sstride, int *flag, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(flag, 1, sizeof(int), s);
    if (e != hipSuccess) return e;
    if (sw <= 0 || sh <= 0) return hipSuccess;
    dim3 grid(min(8, (sw + 255) / 256), sh);
    hipLaunchKernelGGL(opaque_scan_tap64_kernel, grid, dim3(256), 0, s, src, sw, sh, sstride, flag);
    return hipGetLastError();
}

hipError_t launch_draw(uint8_t *dst, int dstride, const uint8_t *src, int sstride, int w, int h,
                       int op, hipStream_t s)
{
    if (w <= 0 || h <= 0) return hipSuccess;
    const bool vec = op == IPX_OP_SRC && (w & 3) == 0 &&
                     ((((uintptr_t)dst) | ((uintptr_t)src) | (uintptr_t)dstride | (uintptr_t)sstride) & 15) == 0;
    if (vec) {
        const int wc = w / 4;
        dim3 grid(min(8, (wc + 255) / 256), h);
        hipLaunchKernelGGL(draw_src_vec_kernel, grid, dim3(256), 0, s, (uint4 *)dst, dstride,
                           (const uint4 *)src, sstride, wc, h);
    } else {
        dim3 grid(min(8, (w + 255) / 256), h);
        hipLaunchKernelGGL(draw_px_kernel, grid, dim3(256), 0, s, dst, dstride, src, sstride, w, h, op);
    }
    return hipGetLastError();
}

// ---- a plain streaming copy: the ceiling of the box at hand (ipx_stream_copy) ----
// Every workgroup copies a contiguous region of its own, four 16-byte loads in flight per lane: 2048 separate sequential streams.  That
// is the faster of the two obvious shapes on MI355X (tools/ubench_streams.hip: 5.0 - 5.3 TB/s against 4.8 - 4.9 for a grid-stride copy
// in which the whole grid moves one window), and it is the shape of the band kernels' own traffic.
__global__ __launch_bounds__(256) void stream_copy_kernel(const uint4 *__restrict__ s, uint4 *__restrict__ d, size_t n)
{
    const size_t per = (n + gridDim.x - 1) / gridDim.x;
    const size_t i0 = (size_t)blockIdx.x * per, i1 = i0 + per < n ? i0 + per : n;
    size_t i = i0 + threadIdx.x;
    for (; i + 768 < i1; i += 1024) {
        const uint4 a = s[i], b = s[i + 256], c = s[i + 512], e = s[i + 768];
        d[i] = a; d[i + 256] = b; d[i + 512] = c; d[i + 768] = e;
    }
    for (; i < i1; i += 256) d[i] = s[i];
}

hipError_t launch_stream_copy(void *dst, const void *src, size_t bytes, hipStream_t s)
{
    const size_t n = bytes / 16;
    const unsigned blocks = (unsigned)std::min<size_t>(2048, (n + 255) / 256);
    hipLaunchKernelGGL(stream_copy_kernel, dim3(blocks), dim3(256), 0, s, (const uint4 *)src, (uint4 *)dst, n);
    return hipGetLastError();
}

hipError_t launch_composite(uint8_t *dst, int dstride, size_t frame_stride, int nframes,
                            const DevGlyph *glyphs_dev, int n, Rect bbox, uint32_t sr, uint32_t sg,
                            uint32_t sb, uint32_t sa, hipStream_t s)
{
    if (n <= 0 || bbox.empty() || nframes <= 0) return hipSuccess;
    dim3 block(64, 4), grid((bbox.dx() + 63) / 64, (bbox.dy() + 4 * kCompositeRows - 1) / (4 * kCompositeRows), nframes);
    hipLaunchKernelGGL(composite_kernel, grid, block, 0, s, dst, dstride, frame_stride, glyphs_dev, n,
                       bbox, sr, sg, sb, sa);
    return hipGetLastError();
}

}  // namespace ipx

// ipx_kernels.hip -- gfx950 (CDNA4) kernels of the pixel path.  No MFMA: every kernel here is a
// streaming / gather kernel bounded by HBM bandwidth (and, for the bilinear taps, by FP64 VALU).
//
// Arithmetic contracts (restating what the reference's libraries compute):
//   * bilinear scale: x/image/draw ablInterpolator, scale_RGBA_RGBA_{Src,Over}: taps widened to
//     16 bit (v*0x101), three float64 lerps with every product rounded before the add (the
//     reference's amd64 build never fuses), truncation to uint32, >> 8.
//   * copy: image/draw drawCopySrc / drawCopyOver.
//   * glyph composite: image/draw drawGlyphOver, uint32 wrap-around preserved.
// The file is compiled with -ffp-contract=off and the pragma below; check with
// `llvm-objdump -d` that the scale kernels hold no v_fma_f64.
#include "ipx_internal.h"

#pragma clang fp contract(off)

namespace ipx {

namespace {

constexpr uint32_t kM = 0xffffu;

// v * 0x101 for byte `C` of a packed RGBA dword, as float64
template <int C>
__device__ __forceinline__ double widen(uint32_t px)
{
    // v_perm_b32: bytes {0, 0, b, b} -> b * 0x101 in one instruction
    constexpr uint32_t sel = 0x0c0c0000u | (uint32_t)C | ((uint32_t)C << 8);
    return (double)__builtin_amdgcn_perm(0u, px, sel);
}

// one output channel: the three lerps of scale_RGBA_RGBA_*, products rounded separately
template <int C>
__device__ __forceinline__ uint32_t lerp_channel(uint32_t p00, uint32_t p10, uint32_t p01,
                                                 uint32_t p11, double xw0, double xw1, double yw0,
                                                 double yw1)
{
    const double s00 = widen<C>(p00), s10 = widen<C>(p10);
    const double s01 = widen<C>(p01), s11 = widen<C>(p11);
    const double top = xw0 * s00 + xw1 * s10;
    const double bot = xw0 * s01 + xw1 * s11;
    const double v = yw0 * top + yw1 * bot;
    return (uint32_t)v;  // truncation, as Go's uint32(float64)
}

__device__ __forceinline__ uint32_t pack_src(uint32_t pr, uint32_t pg, uint32_t pb, uint32_t pa)
{
    // uint8(p >> 8) per channel; p <= 0xffff
    return (pr >> 8) | (pg & 0xff00u) | ((pb & 0xff00u) << 8) | ((pa & 0xff00u) << 16);
}

__device__ __forceinline__ uint32_t blend_over(uint32_t d, uint32_t pr, uint32_t pg, uint32_t pb,
                                               uint32_t pa)
{
    // scale_RGBA_RGBA_Over: dst = uint8((uint32(dst)*pa1/0xffff + p) >> 8), pa1 = (0xffff-pa)*0x101
    const uint32_t pa1 = (kM - pa) * 0x101u;
    const uint32_t r = (((d & 0xffu) * pa1 / kM + pr) >> 8) & 0xffu;
    const uint32_t g = ((((d >> 8) & 0xffu) * pa1 / kM + pg) >> 8) & 0xffu;
    const uint32_t b = ((((d >> 16) & 0xffu) * pa1 / kM + pb) >> 8) & 0xffu;
    const uint32_t a = (((d >> 24) * pa1 / kM + pa) >> 8) & 0xffu;
    return r | (g << 8) | (b << 16) | (a << 24);
}

// ---------------------------------------------------------------------------------------------
// Generic scale: any rectangles, Src or Over, one thread per destination pixel, taps straight
// from global memory.  This is the per-operation seam; the batched path uses band_kernel.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void scale_generic_kernel(ScaleArgs a)
{
    const int dx = a.adr_x0 + (int)(blockIdx.x * 64 + threadIdx.x);
    const int dy = a.adr_y0 + (int)(blockIdx.y * 4 + threadIdx.y);
    if (dx >= a.adr_x1 || dy >= a.adr_y1) return;

    const double sy = ((double)dy + 0.5) * a.yscale - 0.5;
    int sy0 = (int)sy;
    double yf0 = sy - (double)sy0;
    double yf1 = 1 - yf0;
    int sy1 = sy0 + 1;
    if (sy < 0) { sy0 = 0; sy1 = 0; yf0 = 0; yf1 = 1; }
    else if (sy1 > a.ssh - 1) { sy0 = a.ssh - 1; sy1 = a.ssh - 1; yf0 = 1; yf1 = 0; }

    const double sx = ((double)dx + 0.5) * a.xscale - 0.5;
    int sx0 = (int)sx;
    double xf0 = sx - (double)sx0;
    double xf1 = 1 - xf0;
    int sx1 = sx0 + 1;
    if (sx < 0) { sx0 = 0; sx1 = 0; xf0 = 0; xf1 = 1; }
    else if (sx1 > a.ssw - 1) { sx0 = a.ssw - 1; sx1 = a.ssw - 1; xf0 = 1; xf1 = 0; }

    const uint8_t *row0 = a.src + (size_t)(a.sr_y0 + sy0) * a.sstride;
    const uint8_t *row1 = a.src + (size_t)(a.sr_y0 + sy1) * a.sstride;
    const uint32_t p00 = *(const uint32_t *)(row0 + (size_t)(a.sr_x0 + sx0) * 4);
    const uint32_t p10 = *(const uint32_t *)(row0 + (size_t)(a.sr_x0 + sx1) * 4);
    const uint32_t p01 = *(const uint32_t *)(row1 + (size_t)(a.sr_x0 + sx0) * 4);
    const uint32_t p11 = *(const uint32_t *)(row1 + (size_t)(a.sr_x0 + sx1) * 4);

    const uint32_t pr = lerp_channel<0>(p00, p10, p01, p11, xf1, xf0, yf1, yf0);
    const uint32_t pg = lerp_channel<1>(p00, p10, p01, p11, xf1, xf0, yf1, yf0);
    const uint32_t pb = lerp_channel<2>(p00, p10, p01, p11, xf1, xf0, yf1, yf0);
    const uint32_t pa = lerp_channel<3>(p00, p10, p01, p11, xf1, xf0, yf1, yf0);

    uint32_t *d = (uint32_t *)(a.dst + (size_t)(a.dr_y0 + dy) * a.dstride + (size_t)(a.dr_x0 + dx) * 4);
    int op = a.op;
    if (op == IPX_OP_OVER && a.opaque_flag && *a.opaque_flag) op = IPX_OP_SRC;  // draw/scale.go opaque()
    *d = op == IPX_OP_SRC ? pack_src(pr, pg, pb, pa) : blend_over(*d, pr, pg, pb, pa);
}

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

// ---------------------------------------------------------------------------------------------
// drawGlyphOver for one pixel and one mask value; uint32 arithmetic wraps exactly as in Go
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t glyph_over(uint32_t d, uint32_t mask8, uint32_t sr, uint32_t sg,
                                               uint32_t sb, uint32_t sa)
{
    const uint32_t ma = mask8 | (mask8 << 8);
    const uint32_t a = (kM - (sa * ma / kM)) * 0x101u;
    const uint32_t r = (((d & 0xffu) * a + sr * ma) / kM >> 8) & 0xffu;
    const uint32_t g = ((((d >> 8) & 0xffu) * a + sg * ma) / kM >> 8) & 0xffu;
    const uint32_t b = ((((d >> 16) & 0xffu) * a + sb * ma) / kM >> 8) & 0xffu;
    const uint32_t al = (((d >> 24) * a + sa * ma) / kM >> 8) & 0xffu;
    return r | (g << 8) | (b << 16) | (al << 24);
}

// all glyphs, in string order, on the pixel (x, y)
__device__ __forceinline__ uint32_t glyph_run(uint32_t d, int x, int y, const DevGlyph *__restrict__ gl,
                                              int n, uint32_t sr, uint32_t sg, uint32_t sb,
                                              uint32_t sa)
{
    for (int g = 0; g < n; g++) {
        const DevGlyph G = gl[g];
        if (x >= G.x0 && x < G.x1 && y >= G.y0 && y < G.y1) {
            const uint32_t m = G.mask[(size_t)(y - G.y0) * G.mstride + (x - G.x0)];
            if (m) d = glyph_over(d, m, sr, sg, sb, sa);
        }
    }
    return d;
}

// Stand-alone composite over the bounding box of the clipped glyph rectangles.  One wave covers
// a 64-pixel row segment: the glyph table is wave-uniform (scalar loads), a ballot skips glyphs
// no lane of the segment touches, and the untouched pixels are never written.
__global__ __launch_bounds__(256) void composite_kernel(uint8_t *dst, int dstride, size_t frame_stride,
                                                        const DevGlyph *__restrict__ gl, int n,
                                                        Rect bbox, uint32_t sr, uint32_t sg,
                                                        uint32_t sb, uint32_t sa)
{
    const int x = bbox.x0 + (int)(blockIdx.x * 64 + threadIdx.x);
    const int y = bbox.y0 + (int)(blockIdx.y * 4 + threadIdx.y);
    const bool live = x < bbox.x1 && y < bbox.y1;
    uint32_t *p = (uint32_t *)(dst + blockIdx.z * frame_stride + (size_t)y * dstride + (size_t)x * 4);
    uint32_t d = live ? *p : 0u;
    const uint32_t d0 = d;
    for (int g = 0; g < n; g++) {
        const DevGlyph G = gl[g];
        const bool in = live && x >= G.x0 && x < G.x1 && y >= G.y0 && y < G.y1;
        if (!__any(in)) continue;  // wave-uniform skip
        if (in) {
            const uint32_t m = G.mask[(size_t)(y - G.y0) * G.mstride + (x - G.x0)];
            if (m) d = glyph_over(d, m, sr, sg, sb, sa);
        }
    }
    if (live && d != d0) *p = d;
}

// ---------------------------------------------------------------------------------------------
// Fused band kernel: one pass over the source frame produces the watermark copy (+ composite)
// and every scaled output.
//
// A workgroup owns source rows [r0, r1) x columns [c0, c1) of one frame.  It loads those plus one
// halo row and one halo column into LDS with 16-byte coalesced row loads, writes the owned pixels
// to the watermark frame on the way, and then produces every destination pixel of each scaled
// output whose tap pair starts inside the owned block (the pair's second row / column is at most
// the halo).  Which destination rows / columns those are is tabulated on the host with the same
// float64 arithmetic (row_begin / col_begin), as are the taps (AxisTap).
// ---------------------------------------------------------------------------------------------
struct BandGeom {
    int f, b, cb;
};

__device__ __forceinline__ uint32_t lds_u32(const uint8_t *lds, int off)
{
    return *(const uint32_t *)(lds + off);
}

__global__ __launch_bounds__(256, 2) void band_kernel(BandArgs a)
{
    extern __shared__ uint4 lds_raw[];
    uint8_t *lds = (uint8_t *)lds_raw;

    // XCD-aware order: workgroups that share blockIdx % 8 share an XCD (and its L2); give each XCD
    // a contiguous run of (frame, band) blocks so a band's halo row is the neighbour's L2 line.
    const int per_frame = a.nbands * a.ncolblk;
    const int total = per_frame * a.nframes;
    int bid = blockIdx.x;
    {
        const int per_xcd = total >> 3;
        const int body = per_xcd << 3;
        if (bid < body) bid = (bid & 7) * per_xcd + (bid >> 3);
    }
    const int f = bid / per_frame;
    const int rem = bid - f * per_frame;
    const int b = rem / a.ncolblk;
    const int cb = rem - b * a.ncolblk;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    const int r0 = b * a.band_rows;
    const int r1 = min(r0 + a.band_rows, a.sh);
    const int c0 = cb * a.blk_cols;
    const int c1 = min(c0 + a.blk_cols, a.sw);
    const int rows_ld = min(r1 + 1, a.sh) - r0;          // owned rows + halo row
    const int cols_ld = min(c1 + 1, a.sw) - c0;          // owned columns + halo column
    const int pitch = (a.blk_cols + 4) * 4;              // LDS bytes per tile row
    const int nchunk = (cols_ld + 3) >> 2;               // 16-byte chunks per tile row
    const int own_rows = r1 - r0;
    const int own_cols = c1 - c0;

    const uint8_t *sframe = a.src + (size_t)f * a.src_frame_stride;
    uint8_t *wframe = a.wm ? a.wm + (size_t)f * a.wm_frame_stride : nullptr;

    const bool src16 = ((((uintptr_t)sframe) | (uintptr_t)a.sstride) & 15) == 0;
    const bool wm16 = wframe && ((((uintptr_t)wframe) | (uintptr_t)a.wm_stride) & 15) == 0;
    const bool any_glyph = a.nglyphs > 0 && wframe;

    // ---- phase 1: rows -> LDS (+ watermark copy of the owned block) -------------------------
    for (int ry = wave; ry < rows_ld; ry += 4) {
        const int y = r0 + ry;
        const uint8_t *srow = sframe + (size_t)y * a.sstride + (size_t)c0 * 4;
        uint8_t *lrow = lds + ry * pitch;
        uint8_t *wrow = wframe ? wframe + (size_t)y * a.wm_stride + (size_t)c0 * 4 : nullptr;
        const bool own_row = ry < own_rows;
        const bool grow = any_glyph && own_row && y >= a.gbox.y0 && y < a.gbox.y1;
        constexpr int U = 4;
        for (int j0 = lane; j0 < nchunk; j0 += 64 * U) {
            uint4 v[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int j = j0 + u * 64;
                v[u] = make_uint4(0, 0, 0, 0);
                if (j < nchunk) {
                    const int px = j * 4;
                    if (src16 && c0 + px + 4 <= a.sw) {
                        v[u] = *(const uint4 *)(srow + px * 4);
                    } else {
                        const uint32_t *s32 = (const uint32_t *)(srow + px * 4);
                        const int lim = a.sw - (c0 + px);
                        if (lim > 0) v[u].x = s32[0];
                        if (lim > 1) v[u].y = s32[1];
                        if (lim > 2) v[u].z = s32[2];
                        if (lim > 3) v[u].w = s32[3];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int j = j0 + u * 64;
                if (j >= nchunk) continue;
                const int px = j * 4;
                *(uint4 *)(lrow + px * 4) = v[u];
                if (!wrow || !own_row || px >= own_cols) continue;
                uint4 o = v[u];
                if (grow && c0 + px + 4 > a.gbox.x0 && c0 + px < a.gbox.x1) {
                    const int x = c0 + px;
                    o.x = glyph_run(o.x, x + 0, y, a.glyphs, a.nglyphs, a.cr, a.cg, a.cb, a.ca);
                    o.y = glyph_run(o.y, x + 1, y, a.glyphs, a.nglyphs, a.cr, a.cg, a.cb, a.ca);
                    o.z = glyph_run(o.z, x + 2, y, a.glyphs, a.nglyphs, a.cr, a.cg, a.cb, a.ca);
                    o.w = glyph_run(o.w, x + 3, y, a.glyphs, a.nglyphs, a.cr, a.cg, a.cb, a.ca);
                }
                if (wm16 && px + 4 <= own_cols) {
                    *(uint4 *)(wrow + px * 4) = o;
                } else {
                    uint32_t *w32 = (uint32_t *)(wrow + px * 4);
                    const int lim = own_cols - px;
                    if (lim > 0) w32[0] = o.x;
                    if (lim > 1) w32[1] = o.y;
                    if (lim > 2) w32[2] = o.z;
                    if (lim > 3) w32[3] = o.w;
                }
            }
        }
    }
    if (a.nscale == 0) return;
    __syncthreads();

    // ---- phase 2: scaled outputs from the LDS tile ---------------------------------------------
    for (int k = 0; k < a.nscale; k++) {
        const ScaleOut &S = a.sc[k];
        if (!S.out) continue;
        const int dyA = S.row_begin[b], dyB = S.row_begin[b + 1];
        const int dxA = S.col_begin[cb], dxB = S.col_begin[cb + 1];
        if (dyA >= dyB) continue;
        uint8_t *oframe = S.out + (size_t)f * S.frame_stride;
        for (int dx = dxA + tid; dx < dxB; dx += 256) {
            const AxisTap tx = S.xt[dx];
            const int lx = (S.sr_x0 + tx.base - c0) * 4;
            for (int dy = dyA; dy < dyB; dy++) {
                const AxisTap ty = S.yt[dy];  // wave-uniform: scalar loads
                const int off = (S.sr_y0 + ty.base - r0) * pitch + lx;
                const uint32_t p00 = lds_u32(lds, off), p10 = lds_u32(lds, off + 4);
                const uint32_t p01 = lds_u32(lds, off + pitch), p11 = lds_u32(lds, off + pitch + 4);
                const uint32_t pr = lerp_channel<0>(p00, p10, p01, p11, tx.w0, tx.w1, ty.w0, ty.w1);
                const uint32_t pg = lerp_channel<1>(p00, p10, p01, p11, tx.w0, tx.w1, ty.w0, ty.w1);
                const uint32_t pb = lerp_channel<2>(p00, p10, p01, p11, tx.w0, tx.w1, ty.w0, ty.w1);
                const uint32_t pa = lerp_channel<3>(p00, p10, p01, p11, tx.w0, tx.w1, ty.w0, ty.w1);
                *(uint32_t *)(oframe + (size_t)dy * S.ostride + (size_t)dx * 4) = pack_src(pr, pg, pb, pa);
            }
        }
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
hipError_t launch_scale_generic(const ScaleArgs &a, hipStream_t s)
{
    const int w = a.adr_x1 - a.adr_x0, h = a.adr_y1 - a.adr_y0;
    if (w <= 0 || h <= 0) return hipSuccess;
    dim3 block(64, 4), grid((w + 63) / 64, (h + 3) / 4);
    hipLaunchKernelGGL(scale_generic_kernel, grid, block, 0, s, a);
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

hipError_t launch_composite(uint8_t *dst, int dstride, size_t frame_stride, int nframes,
                            const DevGlyph *glyphs_dev, int n, Rect bbox, uint32_t sr, uint32_t sg,
                            uint32_t sb, uint32_t sa, hipStream_t s)
{
    if (n <= 0 || bbox.empty() || nframes <= 0) return hipSuccess;
    dim3 block(64, 4), grid((bbox.dx() + 63) / 64, (bbox.dy() + 3) / 4, nframes);
    hipLaunchKernelGGL(composite_kernel, grid, block, 0, s, dst, dstride, frame_stride, glyphs_dev, n,
                       bbox, sr, sg, sb, sa);
    return hipGetLastError();
}

size_t band_lds_bytes(int band_rows, int blk_cols)
{
    return (size_t)(band_rows + 1) * (size_t)(blk_cols + 4) * 4;
}

hipError_t launch_band(const BandArgs &a, hipStream_t s)
{
    const long long total = (long long)a.nbands * a.ncolblk * a.nframes;
    if (total <= 0) return hipSuccess;
    if (total > 0x7fffffffLL) return hipErrorInvalidValue;
    const size_t lds = band_lds_bytes(a.band_rows, a.blk_cols);
    static thread_local size_t lds_set = 0;
    if (lds > lds_set) {
        hipError_t e = hipFuncSetAttribute((const void *)band_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_set = lds;
    }
    hipLaunchKernelGGL(band_kernel, dim3((unsigned)total), dim3(256), lds, s, a);
    return hipGetLastError();
}

}  // namespace ipx

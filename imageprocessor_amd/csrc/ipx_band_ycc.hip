// ipx_band_ycc.hip -- the fused band kernel for decoded JPEG batches (*image.YCbCr sources).
//
// Same decomposition as band_pipe_kernel (ipx_band.hip): a persistent workgroup walks (frame, band, column
// block) items, keeps the next item's loads in flight while it computes the current one from LDS, and one
// pass over the source produces the watermark frame and both scaled outputs.  What differs is the source:
// three planes (Y at 1 byte per pixel, Cb / Cr subsampled), so a 1080p 4:2:0 frame is 3.1 MB of reads instead
// of 8.3 MB, and the reference's per-operator conversion rules (image_processor.go:47 hands every operator
// the *image.YCbCr itself):
//   * watermark: draw.Draw(result, b, img, Point{}, draw.Src) (watermark.go:92) = imageutil.DrawYCbCr, the
//     8-bit color.YCbCrToRGB per pixel -> done in registers on the way to the store;
//   * crop thumbnail: the equal-size Scale of cropAndResize (thumbnail.go:128-130) is a Copy = DrawYCbCr too,
//     and resizeImage then scales that RGBA8 copy with scale_RGBA_RGBA_* -> taps converted to RGBA8 (mode 1);
//   * resize and the non-crop thumbnail: resizeImage on the YCbCr itself = scale_RGBA_YCbCr4xx_Src, every TAP
//     converted to 16-bit RGB (color.YCbCr.RGBA inlined, clamped) and interpolated in float64 -> mode 0; on
//     dyadic axes the float64 value is sum(w*tap) / 2^(kx+ky) exactly, computed here in u32.
// The LDS tile holds one packed dword (Y, Cb, Cr, 0) per source pixel -- chroma replicated -- so the tile,
// tap tables and item geometry are those of the RGBA kernel and every consumer converts the taps it reads.
//
// Bound: HBM writes (1080p 4:2:0, full pipeline: 3.1 MB in, 11.6 MB out per frame) with the integer colour
// conversion close behind (about 13 VALU ops per watermark pixel, 85 per resized pixel).
#include <algorithm>
#include <cstdlib>

#include "ipx_internal.h"

#pragma clang fp contract(off)

#include "ipx_device.h"
#include "ipx_band_common.h"

namespace ipx {

namespace {

constexpr int kNT = 512;    // threads per workgroup: one 4-pixel chunk per thread and tile row
constexpr int kRows = 9;    // tile rows incl. the halo row

// imageutil.DrawYCbCr / color.YCbCrToRGB for one pixel given the chroma products (shared by the pixels that
// share a chroma sample).  Go: if uint32(r)&0xff000000 == 0 { r >>= 16 } else { r = ^(r >> 31) } == clamp(r >> 16)
struct Chroma8 { int r, g, b; };
__device__ __forceinline__ Chroma8 chroma_products(int cb, int cr)
{
    const int cb1 = cb - 128, cr1 = cr - 128;
    return Chroma8{91881 * cr1, -22554 * cb1 - 46802 * cr1, 116130 * cb1};
}
__device__ __forceinline__ uint32_t rgba8_of(int y, const Chroma8 &c)
{
    const int yy1 = y * 0x10101;
    const uint32_t r = (uint32_t)min(max((yy1 + c.r) >> 16, 0), 255);
    const uint32_t g = (uint32_t)min(max((yy1 + c.g) >> 16, 0), 255);
    const uint32_t b = (uint32_t)min(max((yy1 + c.b) >> 16, 0), 255);
    // Packed with v_perm_b32 on purpose.  Written as r | g << 8 | b << 16, hipcc (ROCm 7.2) selects
    // v_ashr_pk_u8_i32 for the clamp-and-pack of r and g and then ORs b << 16 into the same register assuming its
    // upper half is zero; on gfx950 the instruction leaves the destination's upper 16 bits as they were (the
    // raw green value), which showed up as green bits leaking into blue.
    const uint32_t rg = __builtin_amdgcn_perm(g, r, 0x0c0c0400u);            // {r, g, 0, 0}
    return __builtin_amdgcn_perm(b | 0xff00u, rg, 0x05040100u);              // {r, g, b, 0xff}
}
// a packed (Y, Cb, Cr, 0) tap -> RGBA8
__device__ __forceinline__ uint32_t ycc_rgba8(uint32_t p)
{
    return rgba8_of((int)(p & 0xffu), chroma_products((int)((p >> 8) & 0xffu), (int)((p >> 16) & 0xffu)));
}
// a packed tap -> 16-bit RGB as scale_RGBA_YCbCr4xx_Src converts it
struct Rgb16 { uint32_t r, g, b; };
__device__ __forceinline__ Rgb16 ycc_rgb16(uint32_t p)
{
    const int yy1 = (int)(p & 0xffu) * 0x10101;
    const int cb1 = (int)((p >> 8) & 0xffu) - 128, cr1 = (int)((p >> 16) & 0xffu) - 128;
    Rgb16 t;
    t.r = (uint32_t)min(max((yy1 + 91881 * cr1) >> 8, 0), 0xffff);
    t.g = (uint32_t)min(max((yy1 - 22554 * cb1 - 46802 * cr1) >> 8, 0), 0xffff);
    t.b = (uint32_t)min(max((yy1 + 116130 * cb1) >> 8, 0), 0xffff);
    return t;
}
__device__ __forceinline__ uint32_t lerp16_f64(uint32_t s00, uint32_t s10, uint32_t s01, uint32_t s11, double xw0,
                                               double xw1, double yw0, double yw1)
{
    const double top = xw0 * (double)s00 + xw1 * (double)s10;
    const double bot = xw0 * (double)s01 + xw1 * (double)s11;
    return (uint32_t)(yw0 * top + yw1 * bot);
}
// dyadic axes, integer weights x0 + x1 = 2^kx <= 256, y0 + y1 = 2^ky <= 256: exact in u32, operands < 2^24
__device__ __forceinline__ uint32_t lerp16_int(uint32_t s00, uint32_t s10, uint32_t s01, uint32_t s11, uint32_t x0,
                                               uint32_t x1, uint32_t y0, uint32_t y1, int sh)
{
    const uint32_t top = __umul24(x0, s00) + __umul24(x1, s10);
    const uint32_t bot = __umul24(x0, s01) + __umul24(x1, s11);
    return (__umul24(y0, top) + __umul24(y1, bot)) >> sh;
}

struct Stage {
    uint32_t y[kRows];
    uint32_t cb[kRows], cr[kRows];   // HS: two samples in the low half; VS: only the first (kRows + 1) / 2 are used
};

template <int HS, int VS>
__device__ __forceinline__ void issue_tile_ycc(const YccArgs &A, const Tile &t, int f, bool valid, int tid, Stage &st)
{
    const BandArgs &a = A.b;
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(A.y + (size_t)f * A.y_fs), 0, (a.sh - 1) * A.ystride + a.sw, 0x00020000);
    const int cbytes = (A.ch - 1) * A.cstride + A.cw;
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void *)(A.cb + (size_t)f * A.c_fs), 0, cbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void *)(A.cr + (size_t)f * A.c_fs), 0, cbytes, 0x00020000);
    const int rows = valid ? t.rows_ld : 0;
    const bool in_tile = tid < t.nchunk;
    const int yoff = t.r0 * A.ystride + t.c0 + tid * 4;
#pragma unroll
    for (int r = 0; r < kRows; r++)   // r < rows is wave-uniform
        st.y[r] = __builtin_amdgcn_raw_buffer_load_b32(yrs, in_tile && r < rows ? yoff + r * A.ystride : kOOB, 0, 0);
    constexpr int NCR = VS ? (kRows + 1) / 2 : kRows;
    const int crows = rows > 0 ? ((rows - 1) >> VS) + 1 : 0;
    const int coff = (t.r0 >> VS) * A.cstride + ((t.c0 + tid * 4) >> HS);
#pragma unroll
    for (int j = 0; j < NCR; j++) {
        const int off = in_tile && j < crows ? coff + j * A.cstride : kOOB;
        if (HS) {
            st.cb[j] = (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(brs, off, 0, 0);
            st.cr[j] = (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rrs, off, 0, 0);
        } else {
            st.cb[j] = __builtin_amdgcn_raw_buffer_load_b32(brs, off, 0, 0);
            st.cr[j] = __builtin_amdgcn_raw_buffer_load_b32(rrs, off, 0, 0);
        }
    }
}

// staged planes -> packed tile in LDS, and the owned pixels converted to RGBA8 -> watermark frame
template <int HS, int VS>
__device__ __forceinline__ void drain_tile_ycc(const YccArgs &A, const Tile &t, int f, int tid, const Stage &st,
                                               uint8_t *lds, bool any_glyph)
{
    const BandArgs &a = A.b;
    uint8_t *wframe = a.wm ? a.wm + (size_t)f * a.wm_frame_stride : nullptr;
    const int wm_bytes = wframe ? (a.sh - 1) * a.wm_stride + a.sw * 4 : 0;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)wframe, 0, wm_bytes, 0x00020000);
    const bool gl_rows = any_glyph && t.r0 < a.gbox.y1 && t.r1 > a.gbox.y0;   // wave-uniform
    const bool in_tile = tid < t.nchunk;
    const bool owned = wframe && tid * 4 < t.own_cols;
    const int woff = t.r0 * a.wm_stride + t.c0 * 4 + tid * 16;
    const int loff = tid * 16;
    constexpr int NCR = VS ? (kRows + 1) / 2 : kRows;
#pragma unroll
    for (int j = 0; j < NCR; j++) {
        const uint32_t cbw = st.cb[j], crw = st.cr[j];
        uint32_t cbs[4], crs[4];   // per pixel of the chunk
        if (HS) {
            cbs[0] = cbs[1] = cbw & 0xffu; cbs[2] = cbs[3] = (cbw >> 8) & 0xffu;
            crs[0] = crs[1] = crw & 0xffu; crs[2] = crs[3] = (crw >> 8) & 0xffu;
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) { cbs[i] = (cbw >> (8 * i)) & 0xffu; crs[i] = (crw >> (8 * i)) & 0xffu; }
        }
        Chroma8 cp[4];
        if (HS) { cp[0] = cp[1] = chroma_products((int)cbs[0], (int)crs[0]); cp[2] = cp[3] = chroma_products((int)cbs[2], (int)crs[2]); }
        else {
#pragma unroll
            for (int i = 0; i < 4; i++) cp[i] = chroma_products((int)cbs[i], (int)crs[i]);
        }
#pragma unroll
        for (int rr = 0; rr < (VS ? 2 : 1); rr++) {
            const int r = VS ? 2 * j + rr : j;
            if (r >= kRows) continue;
            const uint32_t yw = st.y[r];
            v4u packed, rgba;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t yv = (yw >> (8 * i)) & 0xffu;
                packed[i] = yv | (cbs[i] << 8) | (crs[i] << 16);
                rgba[i] = rgba8_of((int)yv, cp[i]);
            }
            if (r < t.rows_ld && in_tile) *(v4u *)(lds + r * t.pitch + loff) = packed;
            int off = r < t.own_rows && owned ? woff + r * a.wm_stride : kOOB;
            // chunks that meet the text box are written by the composite step
            if (gl_rows && chunk_in_textbox(a, t.c0 + tid * 4, t.r0 + r)) off = kOOB;
            __builtin_amdgcn_raw_buffer_store_b128(rgba, wrs, off, 0, 0);
        }
    }
}

__device__ __forceinline__ void glyph_phase_ycc(const BandArgs &a, const Tile &t, uint8_t *wframe, const uint8_t *lds, int tid)
{
    const int gy0 = max(a.gbox.y0, t.r0), gy1 = min(a.gbox.y1, t.r1);
    const int gx0 = max(a.gbox.x0 & ~3, t.c0), gx1 = min((a.gbox.x1 + 3) & ~3, t.c1);  // whole skipped chunks
    const int gw = gx1 - gx0, gn = gw * (gy1 - gy0);
    for (int i = tid; i < gn; i += kNT) {
        const int yy = i / gw, x = gx0 + (i - yy * gw), y = gy0 + yy;
        uint32_t d = ycc_rgba8(lds_u32(lds, (y - t.r0) * t.pitch + (x - t.c0) * 4));
        d = glyph_run(d, x, y, a.glyphs, a.nglyphs, a.cr, a.cg, a.cb, a.ca);
        *(uint32_t *)(wframe + (size_t)y * a.wm_stride + (size_t)x * 4) = d;
    }
}

// One scaled output from the packed tile.  mode 0: taps -> 16-bit RGB, then interpolate (scale_RGBA_YCbCr4xx_Src);
// mode 1: taps -> RGBA8 first (the crop copy), then scale_RGBA_RGBA_Src.
template <int NX, bool FP>
__device__ __forceinline__ void scale_out_ycc(const BandArgs &a, int k, int mode, const Tile &t, int f, const uint8_t *lds,
                                              const AxisTap *ytap_k, int tid, const OutCols<NX, FP> &o, int dyA, int dyB)
{
    if (k >= a.nscale || dyA >= dyB) return;
    const ScaleOut &S = a.sc[k];
    uint8_t *oframe = S.out + (size_t)f * S.frame_stride;
    const int ybias = S.sr_y0 - t.r0, xbias = S.sr_x0 - t.c0;
    const int rows = min(kYChunk, dyB - dyA);
    bool live[NX];
    int lx[NX];
    uint32_t *op[NX];
#pragma unroll
    for (int i = 0; i < NX; i++) {
        const int dx = o.dxA + tid + kNT * i;
        live[i] = dx < o.dxB;
        lx[i] = (xbias + o.tx[i].base) * 4;
        op[i] = (uint32_t *)(oframe + (size_t)dyA * S.ostride + (size_t)dx * 4);
    }
    const bool dyadic = !FP || S.dyadic_shift >= 0;
    if (dyadic && mode == 0) {
        const int sh = S.dyadic_shift + 8;
        uint32_t x0[NX], x1[NX];
#pragma unroll
        for (int i = 0; i < NX; i++) { x0[i] = (uint32_t)o.tx[i].f0; x1[i] = (uint32_t)o.tx[i].f1; }
        for (int r = 0; r < rows; r++) {
            const int rowoff = (ybias + __builtin_amdgcn_readfirstlane(ytap_k[r].base)) * t.pitch;
            const uint32_t y0 = (uint32_t)ytap_k[r].f0, y1 = (uint32_t)ytap_k[r].f1;
            uint32_t p[NX][4];
#pragma unroll
            for (int i = 0; i < NX; i++) {
                const int off = rowoff + lx[i];
                p[i][0] = lds_u32(lds, off); p[i][1] = lds_u32(lds, off + 4);
                p[i][2] = lds_u32(lds, off + t.pitch); p[i][3] = lds_u32(lds, off + t.pitch + 4);
            }
#pragma unroll
            for (int i = 0; i < NX; i++) {
                const Rgb16 t00 = ycc_rgb16(p[i][0]), t10 = ycc_rgb16(p[i][1]), t01 = ycc_rgb16(p[i][2]), t11 = ycc_rgb16(p[i][3]);
                const uint32_t pr = lerp16_int(t00.r, t10.r, t01.r, t11.r, x0[i], x1[i], y0, y1, sh);
                const uint32_t pg = lerp16_int(t00.g, t10.g, t01.g, t11.g, x0[i], x1[i], y0, y1, sh);
                const uint32_t pb = lerp16_int(t00.b, t10.b, t01.b, t11.b, x0[i], x1[i], y0, y1, sh);
                if (live[i]) *op[i] = pr | (pg << 8) | (pb << 16) | 0xff000000u;
                op[i] = (uint32_t *)((uint8_t *)op[i] + S.ostride);
            }
        }
    } else if (dyadic) {
        const int sh = S.dyadic_shift + 8;
        for (int r = 0; r < rows; r++) {
            const int rowoff = (ybias + __builtin_amdgcn_readfirstlane(ytap_k[r].base)) * t.pitch;
            const float yf0 = ytap_k[r].f0, yf1 = ytap_k[r].f1;
#pragma unroll
            for (int i = 0; i < NX; i++) {
                const int off = rowoff + lx[i];
                const uint32_t p00 = ycc_rgba8(lds_u32(lds, off)), p10 = ycc_rgba8(lds_u32(lds, off + 4));
                const uint32_t p01 = ycc_rgba8(lds_u32(lds, off + t.pitch)), p11 = ycc_rgba8(lds_u32(lds, off + t.pitch + 4));
                const uint32_t v = lerp_dyadic(p00, p10, p01, p11, o.tx[i].f0, o.tx[i].f1, yf0, yf1, sh);
                if (live[i]) *op[i] = v;
                op[i] = (uint32_t *)((uint8_t *)op[i] + S.ostride);
            }
        }
    } else if constexpr (FP) {
        for (int r = 0; r < rows; r++) {
            const int rowoff = (ybias + __builtin_amdgcn_readfirstlane(ytap_k[r].base)) * t.pitch;
            const double yw0 = ytap_k[r].w0, yw1 = ytap_k[r].w1;
#pragma unroll
            for (int i = 0; i < NX; i++) {
                const int off = rowoff + lx[i];
                const uint32_t q00 = lds_u32(lds, off), q10 = lds_u32(lds, off + 4);
                const uint32_t q01 = lds_u32(lds, off + t.pitch), q11 = lds_u32(lds, off + t.pitch + 4);
                const double xw0 = o.tx[i].w0, xw1 = o.tx[i].w1;
                uint32_t v;
                if (mode == 0) {
                    const Rgb16 t00 = ycc_rgb16(q00), t10 = ycc_rgb16(q10), t01 = ycc_rgb16(q01), t11 = ycc_rgb16(q11);
                    const uint32_t pr = lerp16_f64(t00.r, t10.r, t01.r, t11.r, xw0, xw1, yw0, yw1);
                    const uint32_t pg = lerp16_f64(t00.g, t10.g, t01.g, t11.g, xw0, xw1, yw0, yw1);
                    const uint32_t pb = lerp16_f64(t00.b, t10.b, t01.b, t11.b, xw0, xw1, yw0, yw1);
                    v = pack_src(pr, pg, pb, 0xffffu);
                } else {
                    const uint32_t p00 = ycc_rgba8(q00), p10 = ycc_rgba8(q10), p01 = ycc_rgba8(q01), p11 = ycc_rgba8(q11);
                    const uint32_t pr = lerp_channel<0>(p00, p10, p01, p11, xw0, xw1, yw0, yw1);
                    const uint32_t pg = lerp_channel<1>(p00, p10, p01, p11, xw0, xw1, yw0, yw1);
                    const uint32_t pb = lerp_channel<2>(p00, p10, p01, p11, xw0, xw1, yw0, yw1);
                    const uint32_t pa = lerp_channel<3>(p00, p10, p01, p11, xw0, xw1, yw0, yw1);
                    v = pack_src(pr, pg, pb, pa);
                }
                if (live[i]) *op[i] = v;
                op[i] = (uint32_t *)((uint8_t *)op[i] + S.ostride);
            }
        }
    }
}

struct ItemY {
    int f, b, cb;
    Tile t;
    int dyA[2], dyB[2];
};

__device__ __forceinline__ void item_setup_ycc(const BandArgs &a, ItemY &it, bool valid)
{
    it.t = make_tile(a, it.b, it.cb);
#pragma unroll
    for (int k = 0; k < 2; k++) {
        it.dyA[k] = a.nscale > 0 ? a.sc[k].row_begin[it.b] : 0;
        it.dyB[k] = valid && k < a.nscale ? a.sc[k].row_begin[it.b + 1] : it.dyA[k];
    }
}

template <int NX0, bool FP0, int NX1, bool FP1, int HS, int VS>
__global__ __launch_bounds__(kNT) void band_ycc_kernel(YccArgs A)
{
    extern __shared__ uint4 lds_raw[];
    uint8_t *lds = (uint8_t *)lds_raw;
    const BandArgs &a = A.b;
    const int tid = threadIdx.x;

    const int per_cb = a.nframes * a.nbands;
    const int items = per_cb * a.ncolblk;
    const int G = (int)gridDim.x;
    // grid-interleaved, XCD-contiguous slots (band_pipe_kernel's pipe_order 1)
    const int bid = blockIdx.x;
    int idx = (G & 7) == 0 ? (bid & 7) * (G >> 3) + (bid >> 3) : bid;
    if (idx >= items) return;

    AxisTap *ytap = (AxisTap *)(lds + (a.band_rows + 1) * ((a.blk_cols + 4) * 4));  // [2][kYChunk]
    const bool any_glyph = a.nglyphs > 0 && a.wm;

    auto decode = [&](int i, ItemY &it) {
        it.cb = i / per_cb;
        it.f = (i - it.cb * per_cb) / a.nbands;
        it.b = i - it.cb * per_cb - it.f * a.nbands;
    };
    ItemY cur;
    decode(idx, cur);
    item_setup_ycc(a, cur, true);

    OutCols<NX0, FP0> o0;
    OutCols<NX1, FP1> o1;
    if (a.nscale > 0) { load_xtaps<NX0, FP0, kNT>(a, 0, cur.cb, tid, o0); load_xtaps<NX1, FP1, kNT>(a, 1, cur.cb, tid, o1); }

    Stage st;
    v4u ty_stage[2][2];
    auto issue_ytaps = [&](const ItemY &it) {
        if (a.nscale > 0) {
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const v4u *yp = (const v4u *)&a.sc[k].yt[min(it.dyA[k] + tid, a.sc[k].dh - 1)];
                ty_stage[k][0] = yp[0]; ty_stage[k][1] = yp[1];
            }
        }
    };
    issue_tile_ycc<HS, VS>(A, cur.t, cur.f, true, tid, st);
    issue_ytaps(cur);

    for (;;) {
        // A: staged planes -> packed LDS tile + converted watermark pixels; y taps -> LDS
        drain_tile_ycc<HS, VS>(A, cur.t, cur.f, tid, st, lds, any_glyph);
        if (a.nscale > 0) {
#pragma unroll
            for (int k = 0; k < 2; k++)
                if (tid < min(cur.dyB[k] - cur.dyA[k], kYChunk)) {
                    v4u *yl = (v4u *)&ytap[k * kYChunk + tid];
                    yl[0] = ty_stage[k][0]; yl[1] = ty_stage[k][1];
                }
        }
        __syncthreads();

        // B: the next item's loads
        ItemY nxt;
        const bool has_next = idx + G < items;
        if (has_next) decode(idx + G, nxt);
        else { nxt.b = cur.b; nxt.f = cur.f; nxt.cb = cur.cb; }
        item_setup_ycc(a, nxt, has_next);
        issue_tile_ycc<HS, VS>(A, nxt.t, nxt.f, has_next, tid, st);
        issue_ytaps(nxt);

        // C: the current item from LDS
        if (any_glyph && tile_meets_textbox(a, cur.t))
            glyph_phase_ycc(a, cur.t, a.wm + (size_t)cur.f * a.wm_frame_stride, lds, tid);
        if (a.nscale > 0) {
            scale_out_ycc<NX0, FP0>(a, 0, A.mode[0], cur.t, cur.f, lds, ytap, tid, o0, cur.dyA[0], cur.dyB[0]);
            scale_out_ycc<NX1, FP1>(a, 1, A.mode[1], cur.t, cur.f, lds, ytap + kYChunk, tid, o1, cur.dyA[1], cur.dyB[1]);
        }
        __syncthreads();

        if (!has_next) break;
        if (nxt.cb != cur.cb && a.nscale > 0) {
            load_xtaps<NX0, FP0, kNT>(a, 0, nxt.cb, tid, o0);
            load_xtaps<NX1, FP1, kNT>(a, 1, nxt.cb, tid, o1);
        }
        cur = nxt;
        idx += G;
    }
}

template <int NX0, bool FP0, int NX1, bool FP1, int HS, int VS>
hipError_t launch_ycc(const YccArgs &A, long long items, size_t lds, hipStream_t s)
{
    static KernelLaunchCache cache;
    int resident = 1;
    auto kern = band_ycc_kernel<NX0, FP0, NX1, FP1, HS, VS>;
    hipError_t e = cache.prepare((const void *)kern, kNT, lds, &resident);
    if (e != hipSuccess) return e;
    const long long grid = std::min<long long>(items, (long long)A.b.cus * std::min(A.b.pipe_wgs, resident));
    if (getenv("IPX_DEBUG") && cache.first_report()) {
        fprintf(stderr, "[ipx] band_ycc_kernel<%d,%d,%d,%d,hs%d,vs%d>: tile %d rows x %d cols, lds %zu B, resident %d/CU, grid %lld, items %lld\n",
                NX0, (int)FP0, NX1, (int)FP1, HS, VS, A.b.band_rows, A.b.blk_cols, lds, resident, grid, items);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kNT), lds, s, A);
    return hipGetLastError();
}

template <int HS, int VS>
hipError_t launch_ycc_cfg(const YccArgs &A, long long items, size_t lds, hipStream_t s, bool *matched)
{
    const BandArgs &a = A.b;
    // a.nx_out counts blocks of 256 destination columns per column block; a 512-thread workgroup serves two each
    const int need0 = a.nscale > 0 ? (a.nx_out[0] + 1) / 2 : 0, need1 = a.nscale > 1 ? (a.nx_out[1] + 1) / 2 : 0;
    const bool fp0 = a.nscale > 0 && a.sc[0].dyadic_shift < 0;
    *matched = true;
    if (need0 <= 2 && !fp0 && need1 <= 1) return launch_ycc<2, false, 1, true, HS, VS>(A, items, lds, s);
    if (need0 <= 2 && need1 <= 1) return launch_ycc<2, true, 1, true, HS, VS>(A, items, lds, s);
    *matched = false;
    return hipSuccess;
}

}  // namespace

// Tile shapes and alignments the fused YCbCr kernel is built for; anything else takes the three-kernel path.
bool band_ycc_supported(const YccArgs &A)
{
    const BandArgs &a = A.b;
    const int hs = A.ratio == IPX_YCBCR_422 || A.ratio == IPX_YCBCR_420, vs = A.ratio == IPX_YCBCR_420 || A.ratio == IPX_YCBCR_440;
    if ((a.sw & 3) || a.band_rows + 1 > kRows || a.blk_cols / 4 + 1 > kNT || (a.blk_cols & 3)) return false;
    if (vs && (a.band_rows & 1)) return false;
    if ((((uintptr_t)A.y) | (uintptr_t)A.ystride | A.y_fs) & 3) return false;
    const uintptr_t cal = hs ? 1 : 3;
    if ((((uintptr_t)A.cb) | ((uintptr_t)A.cr) | (uintptr_t)A.cstride | A.c_fs) & cal) return false;
    if (a.wm && ((((uintptr_t)a.wm) | a.wm_frame_stride | (uintptr_t)a.wm_stride) & 15)) return false;
    return true;
}

hipError_t launch_band_ycc(const YccArgs &A, hipStream_t s, bool *matched)
{
    const BandArgs &a = A.b;
    *matched = false;
    const long long total = (long long)a.nbands * a.ncolblk * a.nframes;
    if (total <= 0) { *matched = true; return hipSuccess; }
    if (total > 0x7fffffffLL || !band_ycc_supported(A)) return hipSuccess;
    const size_t lds = band_lds_bytes(a.band_rows, a.blk_cols);
    switch (A.ratio) {
    case IPX_YCBCR_444: return launch_ycc_cfg<0, 0>(A, total, lds, s, matched);
    case IPX_YCBCR_422: return launch_ycc_cfg<1, 0>(A, total, lds, s, matched);
    case IPX_YCBCR_420: return launch_ycc_cfg<1, 1>(A, total, lds, s, matched);
    case IPX_YCBCR_440: return launch_ycc_cfg<0, 1>(A, total, lds, s, matched);
    default: return hipSuccess;
    }
}

}  // namespace ipx

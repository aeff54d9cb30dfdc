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
// v_ashr_pk_u8_i32 (new on gfx950): the low half of the result is {sat_u8(a >> 16), sat_u8(b >> 16)} -- shift, clamp and pack of two
// channels in one instruction; the upper half is left as it was, so consumers read the low half only.  (hipcc's own pattern for
// `r | g << 8 | b << 16` uses this instruction and assumes a zeroed upper half: green bits leaked into blue.  The builtin of the same
// name masks its result with an extra v_and; inline asm plus a v_perm_b32 that picks the two low bytes needs neither.)
__device__ __forceinline__ uint32_t ashr16_pk_u8(int a, int b)
{
    uint32_t d;
    asm("v_ashr_pk_u8_i32 %0, %1, %2, 16" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t rgba8_of(int y, const Chroma8 &c)
{
    const int yy1 = y * 0x10101;
    const uint32_t rg = ashr16_pk_u8(yy1 + c.r, yy1 + c.g), ba = ashr16_pk_u8(yy1 + c.b, 255 << 16);
    return __builtin_amdgcn_perm(ba, rg, 0x05040100u);   // {r, g, b, 0xff}
}
// a packed (Y, Cb, Cr, 0) tap -> RGBA8
__device__ __forceinline__ uint32_t ycc_rgba8(uint32_t p)
{
    return rgba8_of((int)(p & 0xffu), chroma_products((int)((p >> 8) & 0xffu), (int)((p >> 16) & 0xffu)));
}
// how the shared scale paths (ipx_band_common.h, scale_out_conv) read a packed (Y, Cb, Cr, 0) tile dword
struct YccConv {
    static constexpr int NC = 3;    // the converted alpha is 0xffff for every tap: the output alpha is 0xff
    // a packed tap -> 16-bit RGB as scale_RGBA_YCbCr4xx_Src converts it (color.YCbCr.RGBA inlined, clamped)
    static __device__ __forceinline__ void tap16(uint32_t p, uint32_t (&c)[3])
    {
        const int yy1 = (int)(p & 0xffu) * 0x10101;
        const int cb1 = (int)((p >> 8) & 0xffu) - 128, cr1 = (int)((p >> 16) & 0xffu) - 128;
        c[0] = (uint32_t)min(max((yy1 + 91881 * cr1) >> 8, 0), 0xffff);
        c[1] = (uint32_t)min(max((yy1 - 22554 * cb1 - 46802 * cr1) >> 8, 0), 0xffff);
        c[2] = (uint32_t)min(max((yy1 + 116130 * cb1) >> 8, 0), 0xffff);
    }
    static __device__ __forceinline__ uint32_t rgba8(uint32_t p) { return ycc_rgba8(p); }
};

typedef const __attribute__((address_space(4))) int *ConstIntsY;

struct Stage {
    uint32_t y[kRows];
    uint32_t cb[kRows], cr[kRows];   // HS: two samples in the low half; VS: only the first (kRows + 1) / 2 are used
};

// The tile loads of one item.  Clipping is the descriptors' job: each plane's descriptor starts at the tile's first row and ends with
// its last one, so row slots past the tile (or the frame) fall out of range by themselves and return 0; a thread whose chunk lies
// outside the tile carries an out-of-range base offset.  valid = false: empty descriptors.
struct PlaneBases { const uint8_t *y, *cb, *cr; };    // the three planes of one frame of the batch
__device__ __forceinline__ PlaneBases plane_bases(const YccArgs &A, int f)
{
    return PlaneBases{A.y + (size_t)f * A.y_fs, A.cb + (size_t)f * A.c_fs, A.cr + (size_t)f * A.c_fs};
}

template <int HS, int VS>
__device__ __forceinline__ void issue_tile_ycc(const YccArgs &A, const Tile &t, const PlaneBases &pb, bool valid, int tid, Stage &st)
{
    const BandArgs &a = A.b;
    const int crow0 = t.r0 >> VS, crows = valid ? ((t.rows_ld - 1) >> VS) + 1 : 0;
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(pb.y + (size_t)t.r0 * A.ystride), 0, valid ? (t.rows_ld - 1) * A.ystride + a.sw : 0, 0x00020000);
    const int cbytes = crows > 0 ? (crows - 1) * A.cstride + A.cw : 0;
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void *)(pb.cb + (size_t)crow0 * A.cstride), 0, cbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void *)(pb.cr + (size_t)crow0 * A.cstride), 0, cbytes, 0x00020000);
    const bool in_tile = tid < t.nchunk;
    const int yoff = in_tile ? t.c0 + tid * 4 : kOOB;
#pragma unroll
    for (int r = 0; r < kRows; r++) st.y[r] = __builtin_amdgcn_raw_buffer_load_b32(yrs, yoff + r * A.ystride, 0, 0);
    constexpr int NCR = VS ? (kRows + 1) / 2 : kRows;
    const int coff = in_tile ? (t.c0 + tid * 4) >> HS : kOOB;
#pragma unroll
    for (int j = 0; j < NCR; j++) {
        const int off = coff + j * A.cstride;
        if (HS) {
            st.cb[j] = (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(brs, off, 0, 0);
            st.cr[j] = (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rrs, off, 0, 0);
        } else {
            st.cb[j] = __builtin_amdgcn_raw_buffer_load_b32(brs, off, 0, 0);
            st.cr[j] = __builtin_amdgcn_raw_buffer_load_b32(rrs, off, 0, 0);
        }
    }
}

// staged planes -> packed tile in LDS (kRows rows are allocated; rows past the tile hold what their loads returned and are never
// read), and the owned pixels converted to RGBA8 -> watermark frame.  The last tile row is never an owned one (band_rows + 1 <=
// kRows), so it is neither converted nor stored.
template <int HS, int VS>
__device__ __forceinline__ void drain_tile_ycc(const YccArgs &A, const Tile &t, uint8_t *wframe, int tid, const Stage &st,
                                               uint8_t *lds, bool any_glyph)
{
    const BandArgs &a = A.b;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(a.wm ? wframe + (size_t)t.r0 * a.wm_stride : nullptr), 0,
        a.wm ? (t.own_rows - 1) * a.wm_stride + a.sw * 4 : 0, 0x00020000);
    const bool gl_rows = any_glyph && t.r0 < a.gbox.y1 && t.r1 > a.gbox.y0;   // wave-uniform
    const bool in_tile = tid < t.nchunk;
    const int x = t.c0 + tid * 4;
    const int woff = tid * 4 < t.own_cols ? x * 4 : kOOB;
    const bool in_box = gl_rows && x + 4 > a.gbox.x0 && x < a.gbox.x1;
    const int loff = tid * 16;
    constexpr int NCR = VS ? (kRows + 1) / 2 : kRows;
#pragma unroll
    for (int j = 0; j < NCR; j++) {
        const uint32_t cbw = st.cb[j], crw = st.cr[j];
        // per pixel of the chunk: the chroma pair as (cb | cr << 8) for the packed tile dword, and the chroma products for the
        // watermark pixel (shared by the pixels that share a sample)
        uint32_t c2[4];
        Chroma8 cp[4];
        if (HS) {
            c2[0] = c2[1] = __builtin_amdgcn_perm(crw, cbw, 0x0c0c0400u);
            c2[2] = c2[3] = __builtin_amdgcn_perm(crw, cbw, 0x0c0c0501u);
            cp[0] = cp[1] = chroma_products((int)(cbw & 0xffu), (int)(crw & 0xffu));
            cp[2] = cp[3] = chroma_products((int)((cbw >> 8) & 0xffu), (int)((crw >> 8) & 0xffu));
        } else {
            c2[0] = __builtin_amdgcn_perm(crw, cbw, 0x0c0c0400u); c2[1] = __builtin_amdgcn_perm(crw, cbw, 0x0c0c0501u);
            c2[2] = __builtin_amdgcn_perm(crw, cbw, 0x0c0c0602u); c2[3] = __builtin_amdgcn_perm(crw, cbw, 0x0c0c0703u);
#pragma unroll
            for (int i = 0; i < 4; i++) cp[i] = chroma_products((int)((cbw >> (8 * i)) & 0xffu), (int)((crw >> (8 * i)) & 0xffu));
        }
#pragma unroll
        for (int rr = 0; rr < (VS ? 2 : 1); rr++) {
            const int r = VS ? 2 * j + rr : j;
            if (r >= kRows) continue;
            const uint32_t yw = st.y[r];
            v4u packed;     // (Y, Cb, Cr, 0): one v_perm_b32 per pixel
            packed[0] = __builtin_amdgcn_perm(c2[0], yw, 0x0c050400u); packed[1] = __builtin_amdgcn_perm(c2[1], yw, 0x0c050401u);
            packed[2] = __builtin_amdgcn_perm(c2[2], yw, 0x0c050402u); packed[3] = __builtin_amdgcn_perm(c2[3], yw, 0x0c050403u);
            if (in_tile) *(v4u *)(lds + r * t.pitch + loff) = packed;
            if (r == kRows - 1 || !a.wm) continue;
            v4u rgba;
#pragma unroll
            for (int i = 0; i < 4; i++) rgba[i] = rgba8_of((int)((yw >> (8 * i)) & 0xffu), cp[i]);
            // chunks that meet the text box are written by the composite step
            const bool skip = in_box && t.r0 + r >= a.gbox.y0 && t.r0 + r < a.gbox.y1;
            __builtin_amdgcn_raw_buffer_store_b128(rgba, wrs, skip ? kOOB : woff + r * a.wm_stride, 0, 0);
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

struct ItemY {
    int f, b, cb;
    Tile t;
    int dyA[2], dyB[2];
};

__device__ __forceinline__ void item_setup_ycc(const BandArgs &a, ItemY &it, bool valid)
{
    it.t = make_tile(a, it.b, it.cb);
#pragma unroll
    for (int k = 0; k < 2; k++) {   // scalar loads: the tables are read through the constant address space
        const ConstIntsY rb = (ConstIntsY)(uintptr_t)a.sc[k].row_begin;
        it.dyA[k] = a.nscale > 0 ? rb[it.b] : 0;
        it.dyB[k] = valid && k < a.nscale ? rb[it.b + 1] : it.dyA[k];
    }
}

template <int NX0, bool FP0, int NX1, bool FP1, int HS, int VS>
__global__ __launch_bounds__(kNT, kNT / 128) void band_ycc_kernel(YccArgs A)
{
    extern __shared__ uint4 lds_raw[];
    uint8_t *lds = (uint8_t *)lds_raw;
    const BandArgs &a = A.b;
    const int tid = threadIdx.x;

    // one contiguous run of (column block, frame, band) items per workgroup, entered at an offset of its own (band_pipe_kernel has the why)
    const int per_cb = a.nframes * a.nbands;
    const int items = per_cb * a.ncolblk;
    const int G = (int)gridDim.x;
    const int per = (items + G - 1) / G;
    const int idx0 = blockIdx.x * per, idx_end = min(items, idx0 + per);
    if (idx0 >= idx_end) return;
    int idx = idx0 + (int)((blockIdx.x * 67u) % (unsigned)(idx_end - idx0));
    int left = idx_end - idx0;

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
    PlaneBases pb = plane_bases(A, cur.f);        // of the item whose loads go out next
    OutBases ob = out_bases(a, cur.f);            // of the item being drained / computed
    issue_tile_ycc<HS, VS>(A, cur.t, pb, true, tid, st);

    for (;;) {
        // A: staged planes -> packed LDS tile + converted watermark pixels
        drain_tile_ycc<HS, VS>(A, cur.t, ob.wm, tid, st, lds, any_glyph);
        __syncthreads();

        // B: the next item's loads
        ItemY nxt;
        const bool has_next = left > 1;
        nxt.b = cur.b; nxt.f = cur.f; nxt.cb = cur.cb;
        if (has_next) {
            if (idx + 1 == idx_end) { idx = idx0 - 1; decode(idx0, nxt); }     // wrap to the start of the run (once per launch)
            else {
                nxt.b = cur.b + 1;
                if (nxt.b == a.nbands) { nxt.b = 0; if (++nxt.f == a.nframes) { nxt.f = 0; ++nxt.cb; } }
            }
        }
        item_setup_ycc(a, nxt, has_next);
        if (nxt.f != cur.f) pb = plane_bases(A, nxt.f);
        issue_tile_ycc<HS, VS>(A, nxt.t, pb, has_next, tid, st);

        // C: the current item from LDS
        if (any_glyph && tile_meets_textbox(a, cur.t))
            glyph_phase_ycc(a, cur.t, ob.wm, lds, tid);
        if (a.nscale > 0) {
            scale_out_conv<NX0, FP0, kNT, YccConv>(a, 0, A.mode[0], cur.t, ob.o0, lds, tid, o0, cur.dyA[0], cur.dyB[0]);
            scale_out_conv<NX1, FP1, kNT, YccConv>(a, 1, A.mode[1], cur.t, ob.o1, lds, tid, o1, cur.dyA[1], cur.dyB[1]);
        }
        __syncthreads();

        if (!has_next) break;
        if (nxt.cb != cur.cb && a.nscale > 0) {
            load_xtaps<NX0, FP0, kNT>(a, 0, nxt.cb, tid, o0);
            load_xtaps<NX1, FP1, kNT>(a, 1, nxt.cb, tid, o1);
        }
        if (nxt.f != cur.f) ob = out_bases(a, nxt.f);
        cur = nxt;
        idx++;
        left--;
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
    const size_t lds = (size_t)kRows * (size_t)(a.blk_cols + 4) * 4;     // the packed tile alone (y taps come through scalar loads)
    switch (A.ratio) {
    case IPX_YCBCR_444: return launch_ycc_cfg<0, 0>(A, total, lds, s, matched);
    case IPX_YCBCR_422: return launch_ycc_cfg<1, 0>(A, total, lds, s, matched);
    case IPX_YCBCR_420: return launch_ycc_cfg<1, 1>(A, total, lds, s, matched);
    case IPX_YCBCR_440: return launch_ycc_cfg<0, 1>(A, total, lds, s, matched);
    default: return hipSuccess;
    }
}

}  // namespace ipx

// ipx_band_conv.hip -- the fused band kernel for the decoded source types that the reference CONVERTS per operator: *image.YCbCr planes
// (decoded JPEGs), *image.Gray (one plane) and *image.NRGBA frames (PNGs with alpha; *image.Paletted after palette expansion) --
// SURVEY.md 8(f) N2.
//
// Same decomposition as band_pipe_kernel (ipx_band.hip): a persistent workgroup walks (frame, band, column block) items, keeps the next
// item's loads in flight while it computes the current one from LDS, and one pass over the source produces the watermark frame and
// both scaled outputs.  What differs is the source, and the reference's per-operator conversion rules for it (image_processor.go:47
// hands every operator the decoded image itself).  For *image.YCbCr -- three planes, Y at 1 byte per pixel, Cb / Cr subsampled, so a
// 1080p 4:2:0 frame is 3.1 MB of reads instead of 8.3 MB:
//   * watermark: draw.Draw(result, b, img, Point{}, draw.Src) (watermark.go:92) = imageutil.DrawYCbCr, the 8-bit color.YCbCrToRGB;
//   * crop thumbnail: the equal-size Scale of cropAndResize (thumbnail.go:128-130) is a Copy = DrawYCbCr too, and resizeImage then
//     scales that RGBA8 copy with scale_RGBA_RGBA_* -> RGBA8 taps (mode 1);
//   * resize and the non-crop thumbnail: resizeImage on the YCbCr itself = scale_RGBA_YCbCr4xx_Src, every TAP converted to 16-bit RGB
//     (color.YCbCr.RGBA inlined, clamped) and interpolated in float64 -> mode 0; on dyadic axes the float64 value is
//     sum(w*tap) / 2^(kx+ky) exactly, computed here in u32.
// All three read the SAME clamped 24-bit value per channel, v = clamp(yy1 + chroma term, 0, 0xffffff): the 8-bit pixel is its top
// byte (Go: r >> 16, or 0 / 0xff outside), the 16-bit tap its top two bytes (Go: r >> 8 clamped to 0..0xffff).  So every source pixel
// is converted ONCE, on its way from the staging registers to LDS (chroma terms shared by the pixels that share a sample), and the
// tile holds converted taps: two planes of a dword per pixel, R16 | G16 << 16 and B16 | A16 << 16 (A16 = 0xffff for YCbCr / Gray).  The
// scale steps then cost what they cost for an RGBA source -- a v_perm_b32 pairs the channel of two neighbouring taps for one
// v_dot2_u32_u16 with the packed x weights -- instead of converting 2 taps per output column and tile row again.  *image.NRGBA works
// the same way: drawNRGBASrc's pixel is the top byte of scale_RGBA_NRGBA_*'s premultiplied 16-bit tap (convert_px_nrgba).
//
// The price is LDS: 8 bytes per pixel.  The plan carries a second tiling for this kernel (PlanGeom conv, ipx_runtime.hip): column
// blocks of at most 1020 pixels x 8 rows, tile rows of a fixed 4096 bytes per plane, 72 KB per tile, two workgroups per CU.  A
// workgroup is 512 threads: thread = (4-pixel chunk, half); slot_row() below has the row assignment and the halo carry.
//
// Bound: instruction issue (with every store dropped the YCbCr kernel is as slow as with them); the text is composited
// by composite_kernel afterwards because the fused composite cost the item loop its scalar registers (IPX_FUSED_GLYPHS_CONV).
#include <algorithm>
#include <cstdlib>

#include <type_traits>

#include "ipx_internal.h"

#pragma clang fp contract(off)

#include "ipx_device.h"
#include "ipx_band_common.h"

#ifndef IPX_DIAG
#define IPX_DIAG 0      // -DIPX_DIAG=1 (tools/build_diag.sh): ablations by BandArgs::dbg, never in a timed or shipped build
#endif

namespace ipx {

namespace {

constexpr int kNT = 512;        // threads per workgroup
constexpr int kCPR = kNT / 2;   // 4-pixel chunks per tile row: thread = (chunk, half of the tile's rows)
constexpr int kRows = 9;        // tile rows incl. the halo row
static_assert(kCPR * 16 == kConvTilePitch, "a tile row holds one 16-byte slot per chunk index");
constexpr int kYS = 5;          // rows a thread stages: the four rows of its half (slots 0..3) + row 0 (slot 4: lower half, when not carried)

// Chroma terms of color.YCbCrToRGB / color.YCbCr.RGBA for one chroma sample, the -128 folded into the constants.
struct Chroma { int r, g, b; };
__device__ __forceinline__ Chroma chroma_terms(int cb, int cr)
{
    return Chroma{__mul24(cr, 91881) - 91881 * 128, __mul24(cb, -22554) + (__mul24(cr, -46802) + (22554 + 46802) * 128),
                  __mul24(cb, 116130) - 116130 * 128};
}
// one pixel: the clamped 24-bit channels -> the two tile dwords
struct Px { uint32_t lo, hi; };
__device__ __forceinline__ Px convert_px(uint32_t y, const Chroma &c)
{
    const int yy1 = (int)__umul24(y, 0x10101u);
    const int r = min(max(yy1 + c.r, 0), 0xffffff), g = min(max(yy1 + c.g, 0), 0xffffff), b = min(max(yy1 + c.b, 0), 0xffffff);
    Px p;
    p.lo = __builtin_amdgcn_perm((uint32_t)g, (uint32_t)r, 0x06050201u);     // {r.1, r.2, g.1, g.2}
    p.hi = __builtin_amdgcn_perm(0u, (uint32_t)b, 0x0d0d0201u);              // {b.1, b.2, 0xff, 0xff}
    return p;
}
// the RGBA8 pixel of imageutil.DrawYCbCr = the top byte of every channel: {r.2, g.2, b.2, 0xff}
__device__ __forceinline__ uint32_t rgba8_of(uint32_t lo, uint32_t hi) { return __builtin_amdgcn_perm(hi, lo, 0x0d050301u); }
// *image.NRGBA: a tap as scale_RGBA_NRGBA_* reads it -- a16 = a * 0x101, c16 = c * a16 / 0xff -- and the pixel of drawNRGBASrc =
// the top byte of each (c * a16 / 0xff >> 8, alpha a16 >> 8 = a): the same sharing as for YCbCr, one premultiplication per SOURCE pixel
__device__ __forceinline__ Px convert_px_nrgba(uint32_t p)
{
    const uint32_t a16 = (p >> 24) * 0x101u;
    const uint32_t r = (p & 0xffu) * a16 / 0xffu, g = ((p >> 8) & 0xffu) * a16 / 0xffu, b = ((p >> 16) & 0xffu) * a16 / 0xffu;
    return Px{r | g << 16, b | a16 << 16};
}

// how the shared scale paths (ipx_band_common.h) read the two-plane tile
struct YccConv {
    static constexpr int NC = 3;    // the converted alpha is 0xffff for every tap: the output alpha is 0xff
    static __device__ __forceinline__ int plane(const Tile &t) { return kRows * t.pitch; }
    static __device__ __forceinline__ uint32_t rgba8_of(uint32_t lo, uint32_t hi) { return ipx::rgba8_of(lo, hi); }
    static __device__ __forceinline__ uint32_t rgba8_at(const uint8_t *lds, int off, int plane)
    {
        return rgba8_of(lds_u32(lds, off), lds_u32(lds, off + plane));
    }
    static __device__ __forceinline__ void tap16_at(const uint8_t *lds, int off, int plane, uint32_t (&c)[3])
    {
        const uint32_t lo = lds_u32(lds, off), hi = lds_u32(lds, off + plane);
        c[0] = lo & 0xffffu; c[1] = lo >> 16; c[2] = hi & 0xffffu;
    }
    static __device__ __forceinline__ void h16(const uint8_t *lds, int off, int plane, uint32_t iw, uint32_t (&h)[3])
    {
        const uint32_t lo0 = lds_u32(lds, off), lo1 = lds_u32(lds, off + 4), hi0 = lds_u32(lds, off + plane), hi1 = lds_u32(lds, off + plane + 4);
        h[0] = dot2_u16(__builtin_amdgcn_perm(lo1, lo0, 0x05040100u), iw);   // [tap0.r | tap1.r << 16] . [x0 | x1 << 16]
        h[1] = dot2_u16(__builtin_amdgcn_perm(lo1, lo0, 0x07060302u), iw);
        h[2] = dot2_u16(__builtin_amdgcn_perm(hi1, hi0, 0x05040100u), iw);
    }
};

// the same for a tile of premultiplied NRGBA taps: R16 | G16 << 16 and B16 | A16 << 16 (alpha is a channel like the others)
struct NrgbaConv2 {
    static constexpr int NC = 4;
    static __device__ __forceinline__ int plane(const Tile &t) { return kRows * t.pitch; }
    static __device__ __forceinline__ uint32_t rgba8_of(uint32_t lo, uint32_t hi) { return __builtin_amdgcn_perm(hi, lo, 0x07050301u); }
    static __device__ __forceinline__ uint32_t rgba8_at(const uint8_t *lds, int off, int plane)
    {
        return rgba8_of(lds_u32(lds, off), lds_u32(lds, off + plane));
    }
    static __device__ __forceinline__ void tap16_at(const uint8_t *lds, int off, int plane, uint32_t (&c)[4])
    {
        const uint32_t lo = lds_u32(lds, off), hi = lds_u32(lds, off + plane);
        c[0] = lo & 0xffffu; c[1] = lo >> 16; c[2] = hi & 0xffffu; c[3] = hi >> 16;
    }
    static __device__ __forceinline__ void h16(const uint8_t *lds, int off, int plane, uint32_t iw, uint32_t (&h)[4])
    {
        const uint32_t lo0 = lds_u32(lds, off), lo1 = lds_u32(lds, off + 4), hi0 = lds_u32(lds, off + plane), hi1 = lds_u32(lds, off + plane + 4);
        h[0] = dot2_u16(__builtin_amdgcn_perm(lo1, lo0, 0x05040100u), iw);
        h[1] = dot2_u16(__builtin_amdgcn_perm(lo1, lo0, 0x07060302u), iw);
        h[2] = dot2_u16(__builtin_amdgcn_perm(hi1, hi0, 0x05040100u), iw);
        h[3] = dot2_u16(__builtin_amdgcn_perm(hi1, hi0, 0x07060302u), iw);
    }
};

typedef const __attribute__((address_space(4))) int *ConstIntsY;

// Tile row of a thread's staging slot s.  A tile is rows 0..8 of its band (8 = the halo row = row 0 of the next band).  Slots 0..3 are rows
// 4*half + 1 .. 4*half + 4: the lower half of the workgroup converts rows 1..4, the upper half rows 5..8.  Row 0 has two sources: when the
// workgroup's previous item was the band above (the common case in a run), row 0 IS that item's row 8 and still sits converted in the
// tile -- it is copied from row slot 8 to row slot 0 inside LDS (**halo carry**: not loaded, not converted again, and its watermark
// pixels were stored by the previous item); otherwise the lower half loads and converts it through slot 4.
__device__ __forceinline__ int slot_row(int half, int s) { return s < 4 ? 4 * half + s + 1 : 0; }

// ---- the two sources --------------------------------------------------------------------------------------------------------------
// A source type says how a thread stages its rows (Stage, issue) and how the staged rows become converted taps (rows: calls
// row(s, lo, hi) for each of the thread's staging slots, s = 4 only when the thread owns row 0).  Clipping is the descriptors' job: a
// descriptor starts at the tile's first row and ends with its last one, so row slots past the tile (or the frame) fall out of range by
// themselves and return 0; a thread whose chunk lies outside the tile carries an out-of-range base offset.  valid = false: empty
// descriptors.  carry: row 0 is not loaded.

template <int HS, int VS>
struct YccSrc {          // *image.YCbCr planes (and *image.Gray as Y + a stride-0 row of 128s)
    typedef YccArgs Args;
    typedef YccConv Conv;
    static constexpr int NCS = VS ? 3 : kYS;   // chroma rows a thread stages (VS: one per pair of y rows)
    struct Stage {
        uint32_t y[kYS];
        uint32_t cb[NCS], cr[NCS];             // HS: two samples in the low half
    };
    struct Bases { const uint8_t *y, *cb, *cr; };    // the three planes of one frame of the batch
    static __device__ __forceinline__ Bases bases(const Args &A, int f)
    {
        return Bases{A.y + (size_t)f * A.y_fs, A.cb + (size_t)f * A.c_fs, A.cr + (size_t)f * A.c_fs};
    }
    static __device__ __forceinline__ void issue(const Args &A, const Tile &t, const Bases &pb, bool valid, bool carry, int chunk, int half, Stage &st)
    {
        const BandArgs &a = A.b;
        // descriptors over the whole plane of the frame (rows past the frame's last one fall out of range; rows of the tile that exist
        // are always readable), the tile's first row in the offset: no 64-bit address arithmetic per item
        const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void *)pb.y, 0, (a.sh - 1) * A.ystride + a.sw, 0x00020000);
        const int cbytes = (A.ch - 1) * A.cstride + A.cw;
        const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void *)pb.cb, 0, cbytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void *)pb.cr, 0, cbytes, 0x00020000);
        const bool in_tile = valid && chunk < t.nchunk;
        const bool row0_mine = half == 0 && !carry;                           // wave-uniform
        const int yoff = in_tile ? t.r0 * A.ystride + t.c0 + chunk * 4 : kOOB;
#pragma unroll
        for (int s = 0; s < kYS; s++) {
            const bool mine = s < 4 || row0_mine;
            st.y[s] = __builtin_amdgcn_raw_buffer_load_b32(yrs, mine ? yoff + slot_row(half, s) * A.ystride : kOOB, 0, 0);   // (row in the VGPR offset: the range check ignores the scalar one)
        }
        const int coff = in_tile ? (t.r0 >> VS) * A.cstride + ((t.c0 + chunk * 4) >> HS) : kOOB;
#pragma unroll
        for (int j = 0; j < NCS; j++) {
            // chroma row of slot j: VS: rows 2*half + j (they serve rows 4*half + 1 .. 4*half + 4, and chroma row 0 serves row 0); else the row of y slot j
            const int crow = VS ? 2 * half + j : slot_row(half, j);
            const bool mine = VS || j < 4 || row0_mine;
            const int off = mine ? coff + crow * A.cstride : kOOB;
            if (HS) {
                st.cb[j] = (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(brs, off, 0, 0);
                st.cr[j] = (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rrs, off, 0, 0);
            } else {
                st.cb[j] = __builtin_amdgcn_raw_buffer_load_b32(brs, off, 0, 0);
                st.cr[j] = __builtin_amdgcn_raw_buffer_load_b32(rrs, off, 0, 0);
            }
        }
    }
    template <class RowFn>
    static __device__ __forceinline__ void rows(const Stage &st, bool row0_mine, RowFn row)
    {
        auto terms = [&](int j, Chroma (&cp)[4]) {             // per pixel of the chunk (shared by the pixels that share a sample)
            const uint32_t cbw = st.cb[j], crw = st.cr[j];
            if (HS) {
                cp[0] = cp[1] = chroma_terms((int)(cbw & 0xffu), (int)(crw & 0xffu));
                cp[2] = cp[3] = chroma_terms((int)((cbw >> 8) & 0xffu), (int)((crw >> 8) & 0xffu));
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++) cp[i] = chroma_terms((int)((cbw >> (8 * i)) & 0xffu), (int)((crw >> (8 * i)) & 0xffu));
            }
        };
        auto conv = [&](int s, const Chroma (&cp)[4]) {
            const uint32_t yw = st.y[s];
            v4u lo, hi;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const Px p = convert_px((yw >> (8 * i)) & 0xffu, cp[i]);
                lo[i] = p.lo; hi[i] = p.hi;
            }
            row(s, lo, hi);
        };
        Chroma cp[4];
        if (VS) {                                              // y slots in the order of their chroma slots: 0 | (row 0) | 1 2 | 3
            terms(0, cp); conv(0, cp);
            if (row0_mine) conv(4, cp);
            terms(1, cp); conv(1, cp); conv(2, cp);
            terms(2, cp); conv(3, cp);
        } else {
#pragma unroll
            for (int s = 0; s < 4; s++) { terms(s, cp); conv(s, cp); }
            if (row0_mine) { terms(4, cp); conv(4, cp); }
        }
    }
    static __device__ __forceinline__ void touch(Stage &st)    // (diagnostic stamps: the wait for the staged loads on its own)
    {
#pragma unroll
        for (int q = 0; q < kYS; q++) asm volatile("" : "+v"(st.y[q]));
#pragma unroll
        for (int q = 0; q < NCS; q++) asm volatile("" : "+v"(st.cb[q]), "+v"(st.cr[q]));
    }
};

struct GraySrc {         // *image.Gray: the Y plane alone.  drawGray and scale_RGBA_Gray_Src read a pixel as (y, y, y, 0xff); as 16-bit taps
                         // that is y * 0x101 per colour channel (color.Gray.RGBA) -- one multiply per tile dword, no chroma arithmetic
    typedef YccArgs Args;
    typedef YccConv Conv;
    struct Stage { uint32_t y[kYS]; };
    typedef const uint8_t *Bases;
    static __device__ __forceinline__ Bases bases(const Args &A, int f) { return A.y + (size_t)f * A.y_fs; }
    static __device__ __forceinline__ void issue(const Args &A, const Tile &t, const Bases &py, bool valid, bool carry, int chunk, int half, Stage &st)
    {
        const BandArgs &a = A.b;
        const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void *)py, 0, (a.sh - 1) * A.ystride + a.sw, 0x00020000);
        const bool row0_mine = half == 0 && !carry;                           // wave-uniform
        const int yoff = valid && chunk < t.nchunk ? t.r0 * A.ystride + t.c0 + chunk * 4 : kOOB;
#pragma unroll
        for (int s = 0; s < kYS; s++) {
            const bool mine = s < 4 || row0_mine;
            st.y[s] = __builtin_amdgcn_raw_buffer_load_b32(yrs, mine ? yoff + slot_row(half, s) * A.ystride : kOOB, 0, 0);
        }
    }
    template <class RowFn>
    static __device__ __forceinline__ void rows(const Stage &st, bool row0_mine, RowFn row)
    {
        auto conv = [&](int s) {
            const uint32_t yw = st.y[s];
            v4u lo, hi;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                lo[i] = __builtin_amdgcn_perm(0u, yw, 0x01010101u * (uint32_t)i);                  // {y, y, y, y}: R16 | G16 << 16 = y * 0x101 twice
                hi[i] = __builtin_amdgcn_perm(0u, yw, 0x0d0d0000u | (uint32_t)(i | i << 8));       // {y, y, 0xff, 0xff}: B16 | 0xffff << 16
            }
            row(s, lo, hi);
        };
#pragma unroll
        for (int s = 0; s < 4; s++) conv(s);
        if (row0_mine) conv(4);
    }
    static __device__ __forceinline__ void touch(Stage &st)
    {
#pragma unroll
        for (int q = 0; q < kYS; q++) asm volatile("" : "+v"(st.y[q]));
    }
};

struct NrgbaSrc {        // *image.NRGBA frames (PNGs with alpha; *image.Paletted frames after their palette expansion)
    typedef NrgbaArgs Args;
    typedef NrgbaConv2 Conv;
    struct Stage { v4u px[kYS]; };
    typedef const uint8_t *Bases;
    static __device__ __forceinline__ Bases bases(const Args &A, int f) { return A.b.src + (size_t)f * A.b.src_frame_stride; }
    static __device__ __forceinline__ void issue(const Args &A, const Tile &t, const Bases &sframe, bool valid, bool carry, int chunk, int half, Stage &st)
    {
        const BandArgs &a = A.b;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)sframe, 0, (a.sh - 1) * a.sstride + a.sw * 4, 0x00020000);
        const bool row0_mine = half == 0 && !carry;                           // wave-uniform
        const int off = valid && chunk < t.nchunk ? t.r0 * a.sstride + t.c0 * 4 + chunk * 16 : kOOB;
#pragma unroll
        for (int s = 0; s < kYS; s++) {
            const bool mine = s < 4 || row0_mine;
            st.px[s] = __builtin_amdgcn_raw_buffer_load_b128(rs, mine ? off + slot_row(half, s) * a.sstride : kOOB, 0, 0);
        }
    }
    template <class RowFn>
    static __device__ __forceinline__ void rows(const Stage &st, bool row0_mine, RowFn row)
    {
        auto conv = [&](int s) {
            v4u lo, hi;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const Px p = convert_px_nrgba(st.px[s][i]);
                lo[i] = p.lo; hi[i] = p.hi;
            }
            row(s, lo, hi);
        };
#pragma unroll
        for (int s = 0; s < 4; s++) conv(s);
        if (row0_mine) conv(4);
    }
    static __device__ __forceinline__ void touch(Stage &st)
    {
#pragma unroll
        for (int q = 0; q < kYS; q++) asm volatile("" : "+v"(st.px[q]));
    }
};

struct Tap64Src {        // frames of ready-made taps: At(x, y).RGBA() as four little-endian uint16 per pixel -- what deep_expand_kernel
                         // makes of *image.NRGBA64 / RGBA64 / Gray16 / CMYK frames.  The two dwords of a pixel ARE the tile's two planes.
    typedef NrgbaArgs Args;
    typedef NrgbaConv2 Conv;
    struct Stage { v4u a[kYS], b[kYS]; };          // pixels 0 1 | 2 3 of the chunk
    typedef const uint8_t *Bases;
    static __device__ __forceinline__ Bases bases(const Args &A, int f) { return A.b.src + (size_t)f * A.b.src_frame_stride; }
    static __device__ __forceinline__ void issue(const Args &A, const Tile &t, const Bases &sframe, bool valid, bool carry, int chunk, int half, Stage &st)
    {
        const BandArgs &a = A.b;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)sframe, 0, (a.sh - 1) * a.sstride + a.sw * 8, 0x00020000);
        const bool row0_mine = half == 0 && !carry;                           // wave-uniform
        const int off = valid && chunk < t.nchunk ? t.r0 * a.sstride + t.c0 * 8 + chunk * 32 : kOOB;
#pragma unroll
        for (int s = 0; s < kYS; s++) {
            const bool mine = s < 4 || row0_mine;
            const int o = mine ? off + slot_row(half, s) * a.sstride : kOOB;
            st.a[s] = __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, 0);
            st.b[s] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((uint32_t)o + 16u), 0, 0);     // (kOOB + 16 stays out of range)
        }
    }
    template <class RowFn>
    static __device__ __forceinline__ void rows(const Stage &st, bool row0_mine, RowFn row)
    {
        auto conv = [&](int s) {
            const v4u lo = {st.a[s][0], st.a[s][2], st.b[s][0], st.b[s][2]}, hi = {st.a[s][1], st.a[s][3], st.b[s][1], st.b[s][3]};
            row(s, lo, hi);
        };
#pragma unroll
        for (int s = 0; s < 4; s++) conv(s);
        if (row0_mine) conv(4);
    }
    static __device__ __forceinline__ void touch(Stage &st)
    {
#pragma unroll
        for (int q = 0; q < kYS; q++) asm volatile("" : "+v"(st.a[q]), "+v"(st.b[q]));
    }
};

// floor(x / 0xffff) for x <= 0xffff * 0xffff (checked over every quotient boundary): color.NRGBA64.RGBA and color.CMYK.RGBA divide by it
__device__ __forceinline__ uint32_t div_ffff(uint32_t x) { return (x + (x >> 16) + 1u) >> 16; }

typedef uint32_t v2u __attribute__((ext_vector_type(2)));

template <int KIND>
struct DeepSrc {         // *image.NRGBA64 / RGBA64 / Gray16 / CMYK frames as Go's Pix holds them (big-endian 16-bit channels; C M Y K bytes),
                         // converted to At(x, y).RGBA() on the way into the tile -- the conversions of deep_expand_kernel (ipx_kernels.hip), per
                         // source pixel, without the trip through HBM
    typedef NrgbaArgs Args;
    static constexpr bool kAlpha = KIND == IPX_DEEP_NRGBA64 || KIND == IPX_DEEP_RGBA64;     // the other two are opaque: three channels to interpolate
    typedef typename std::conditional<kAlpha, NrgbaConv2, YccConv>::type Conv;
    static constexpr int BPP = KIND == IPX_DEEP_GRAY16 ? 2 : (KIND == IPX_DEEP_CMYK ? 4 : 8);
    struct Stage { uint32_t w[kYS][BPP]; };        // a chunk of 4 pixels is BPP dwords
    typedef const uint8_t *Bases;
    static __device__ __forceinline__ Bases bases(const Args &A, int f) { return A.b.src + (size_t)f * A.b.src_frame_stride; }
    static __device__ __forceinline__ void issue(const Args &A, const Tile &t, const Bases &sframe, bool valid, bool carry, int chunk, int half, Stage &st)
    {
        const BandArgs &a = A.b;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)sframe, 0, (a.sh - 1) * a.sstride + a.sw * BPP, 0x00020000);
        const bool row0_mine = half == 0 && !carry;                           // wave-uniform
        const int off = valid && chunk < t.nchunk ? t.r0 * a.sstride + (t.c0 + chunk * 4) * BPP : kOOB;
#pragma unroll
        for (int s = 0; s < kYS; s++) {
            const bool mine = s < 4 || row0_mine;
            const int o = mine ? off + slot_row(half, s) * a.sstride : kOOB;
            if constexpr (BPP == 2) {
                const v2u v = __builtin_amdgcn_raw_buffer_load_b64(rs, o, 0, 0);
                st.w[s][0] = v[0]; st.w[s][1] = v[1];
            } else {
                const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; i++) st.w[s][i] = v[i];
                if constexpr (BPP == 8) {
                    const v4u u = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((uint32_t)o + 16u), 0, 0);     // (kOOB + 16 stays out of range)
#pragma unroll
                    for (int i = 0; i < 4; i++) st.w[s][4 + i] = u[i];
                }
            }
        }
    }
    static __device__ __forceinline__ Px convert(const uint32_t (&w)[BPP], int i)
    {
        Px p;
        if constexpr (KIND == IPX_DEEP_GRAY16) {
            const uint32_t v = w[i >> 1];
            p.lo = __builtin_amdgcn_perm(0u, v, i & 1 ? 0x02030203u : 0x00010001u);         // {y.lo, y.hi} twice: R16 | G16 << 16
            p.hi = __builtin_amdgcn_perm(0u, v, i & 1 ? 0x0d0d0203u : 0x0d0d0001u);         // B16 | 0xffff << 16
        } else if constexpr (KIND == IPX_DEEP_CMYK) {
            const uint32_t v = w[i];
            const uint32_t wk = 0xffffu - (v >> 24) * 0x101u;
            const uint32_t r = div_ffff(__umul24(0xffffu - (v & 0xffu) * 0x101u, wk)), g = div_ffff(__umul24(0xffffu - ((v >> 8) & 0xffu) * 0x101u, wk)),
                           b = div_ffff(__umul24(0xffffu - ((v >> 16) & 0xffu) * 0x101u, wk));
            p.lo = r | g << 16; p.hi = b | 0xffff0000u;
        } else {
            const uint32_t rg = __builtin_amdgcn_perm(0u, w[2 * i], 0x02030001u), ba = __builtin_amdgcn_perm(0u, w[2 * i + 1], 0x02030001u);   // big-endian pairs -> lo | hi << 16
            if constexpr (KIND == IPX_DEEP_NRGBA64) {
                const uint32_t al = ba >> 16;
                const uint32_t r = div_ffff(__umul24(rg & 0xffffu, al)), g = div_ffff(__umul24(rg >> 16, al)), b = div_ffff(__umul24(ba & 0xffffu, al));
                p.lo = r | g << 16; p.hi = b | (ba & 0xffff0000u);
            } else { p.lo = rg; p.hi = ba; }
        }
        return p;
    }
    template <class RowFn>
    static __device__ __forceinline__ void rows(const Stage &st, bool row0_mine, RowFn row)
    {
        auto conv = [&](int s) {
            v4u lo, hi;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const Px p = convert(st.w[s], i);
                lo[i] = p.lo; hi[i] = p.hi;
            }
            row(s, lo, hi);
        };
#pragma unroll
        for (int s = 0; s < 4; s++) conv(s);
        if (row0_mine) conv(4);
    }
    static __device__ __forceinline__ void touch(Stage &st)
    {
#pragma unroll
        for (int q = 0; q < kYS; q++)
#pragma unroll
            for (int i = 0; i < BPP; i++) asm volatile("" : "+v"(st.w[q][i]));
    }
};

// staged rows -> converted tile in LDS (kRows rows are allocated; rows past the tile hold what their loads returned and are never
// read), and the top bytes of every row converted here -> watermark frame (rows 1..8: the halo row's pixels are stored by the item that
// converts it; row 0 only where it is converted, i.e. not under carry).
template <class Src>
__device__ __forceinline__ void drain_tile_conv(const BandArgs &a, const Tile &t, uint8_t *wframe, bool carry, int chunk, int half,
                                                const typename Src::Stage &st, uint8_t *lds, bool any_glyph)
{
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void *)wframe, 0,                                                        // the whole frame: rows past its last one are clipped
#if IPX_DIAG
        a.wm && !(a.dbg & 32) ? (a.sh - 1) * a.wm_stride + a.sw * 4 : 0, 0x00020000);   // 32: every store dropped by the descriptor
#else
        a.wm ? (a.sh - 1) * a.wm_stride + a.sw * 4 : 0, 0x00020000);
#endif
    // (the text box test takes the halo row along: its chunks inside the box are the next band's composite step's to write)
    const bool gl_rows = any_glyph && t.r0 < a.gbox.y1 && t.r0 + t.rows_ld > a.gbox.y0;   // wave-uniform
    const int x = t.c0 + chunk * 4;
    const int woff = chunk * 4 < t.own_cols ? t.r0 * a.wm_stride + x * 4 : kOOB;
    // chunks that meet the text box are written by the composite step: their store offset gets the top bit (beyond any frame)
    const uint32_t in_box = gl_rows && x + 4 > a.gbox.x0 && x < a.gbox.x1 ? 0x80000000u : 0u;
    const int loff = chunk * 16, plane = kRows * t.pitch;
    if (carry && half) {                                   // row 8 of the previous item -> row 0, by the thread that overwrites this chunk of row 8 below
        const v4u lo8 = *(const v4u *)(lds + (kRows - 1) * t.pitch + loff), hi8 = *(const v4u *)(lds + plane + (kRows - 1) * t.pitch + loff);
        *(v4u *)(lds + loff) = lo8;
        *(v4u *)(lds + plane + loff) = hi8;
    }
    Src::rows(st, half == 0 && !carry, [&](int s, const v4u &lo, const v4u &hi) {
        const int r = slot_row(half, s);
#if IPX_DIAG
        if (a.dbg & 8) return;                                            // 8: neither LDS writes nor watermark stores
#endif
        // (every chunk index has a slot in the fixed-pitch tile row: threads past the tile's columns write what their loads returned)
        *(v4u *)(lds + r * t.pitch + loff) = lo;
        *(v4u *)(lds + plane + r * t.pitch + loff) = hi;
        if (!a.wm) return;
        v4u rgba;
#pragma unroll
        for (int i = 0; i < 4; i++) rgba[i] = Src::Conv::rgba8_of(lo[i], hi[i]);
        const uint32_t row_in_box = t.r0 + r >= a.gbox.y0 && t.r0 + r < a.gbox.y1 ? ~0u : 0u;       // scalar
#if IPX_DIAG
        if (a.dbg & 16) return;                                           // 16: no watermark stores
#endif
        __builtin_amdgcn_raw_buffer_store_b128(rgba, wrs, (int)((uint32_t)(woff + r * a.wm_stride) | (in_box & row_in_box)), 0, 0);
    });
}

struct ItemY {
    int f, b, cb;
    Tile t;
    int dyA[2], dyB[2];
};

__device__ __forceinline__ void item_setup_conv(const BandArgs &a, ItemY &it, bool valid)
{
    it.t = make_tile(a, it.b, it.cb);
    it.t.pitch = kConvTilePitch;       // one slot per chunk index: no column test on the way into LDS, constant row offsets
    band_out_rows(a, it.b, valid, it.dyA, it.dyB);
}

template <int NX0, bool FP0, int NX1, bool FP1, class Src>
__global__ __launch_bounds__(kNT, kNT / 128) void band_conv_kernel(typename Src::Args A)
{
    extern __shared__ uint4 lds_raw[];
    uint8_t *lds = (uint8_t *)lds_raw;
    const BandArgs &a = A.b;
    const int tid = threadIdx.x, chunk = tid & (kCPR - 1), half = __builtin_amdgcn_readfirstlane(tid / kCPR);   // (half is wave-uniform)

    // one contiguous run of (column block, frame, band) items per workgroup, entered at an offset of its own (band_pipe_kernel has the why)
    const int per_cb = a.nframes * a.nbands;
    const int items = per_cb * a.ncolblk;
    const int G = (int)gridDim.x;
    const int per = (items + G - 1) / G;
    const int idx0 = blockIdx.x * per, idx_end = min(items, idx0 + per);
    if (idx0 >= idx_end) return;
    int idx = idx0 + (int)((blockIdx.x * 67u) % (unsigned)(idx_end - idx0));
    int left = idx_end - idx0;

    const bool any_glyph = IPX_FUSED_GLYPHS_CONV && a.nglyphs > 0 && a.wm;

    auto decode = [&](int i, ItemY &it) {
        it.cb = i / per_cb;
        it.f = (i - it.cb * per_cb) / a.nbands;
        it.b = i - it.cb * per_cb - it.f * a.nbands;
    };
    ItemY cur;
    decode(idx, cur);
    item_setup_conv(a, cur, true);

    OutCols<NX0, FP0> o0;
    OutCols<NX1, FP1> o1;
    if (a.nscale > 0) { load_xtaps<NX0, FP0, kNT>(a, 0, cur.cb, tid, o0); load_xtaps<NX1, FP1, kNT>(a, 1, cur.cb, tid, o1); }

    typename Src::Stage st;
    typename Src::Bases pb = Src::bases(A, cur.f);   // of the item whose loads go out next
    OutBases ob = out_bases(a, cur.f);                // of the item being drained / computed
    Src::issue(A, cur.t, pb, true, false, chunk, half, st);
    bool cur_carry = false;                           // row 0 of `cur` is row 8 of the item this workgroup processed just before

    // In-kernel phase stamps exist only in the diagnostic build (-DIPX_DIAG=1, tools/build_diag.sh); the shipped kernel executes none.
#if IPX_DIAG
    unsigned long long acc[6] = {0, 0, 0, 0, 0, 0};
#define IPX_STAMP(i) do { if (a.stamps) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc[i] += now_ - tprev; tprev = now_; } } while (0)
    unsigned long long tprev = a.stamps ? __builtin_amdgcn_s_memtime() : 0;
#else
#define IPX_STAMP(i) do { } while (0)
#endif
    for (;;) {
        // A: staged rows -> converted LDS tile + watermark pixels
#if IPX_DIAG
        if (a.stamps) { Src::touch(st); IPX_STAMP(0); }
#endif
        drain_tile_conv<Src>(a, cur.t, ob.wm, cur_carry, chunk, half, st, lds, any_glyph);
        IPX_STAMP(1);
        __syncthreads();
        IPX_STAMP(2);

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
        item_setup_conv(a, nxt, has_next);
        if (nxt.f != cur.f) pb = Src::bases(A, nxt.f);
        const bool nxt_carry = has_next && a.band_rows + 1 == kRows && nxt.b == cur.b + 1 && nxt.f == cur.f && nxt.cb == cur.cb;
        Src::issue(A, nxt.t, pb, has_next, nxt_carry, chunk, half, st);
        IPX_STAMP(3);

        // C: the current item from LDS
        if (any_glyph && tile_meets_textbox(a, cur.t))
            glyph_phase<kNT, kRows - 1, typename Src::Conv>(a, cur.t, ob.wm, lds, tid);
#if IPX_DIAG
        if (!(a.dbg & 1))                                                     // 1: skip scaling
#endif
        if (a.nscale > 0) {
            scale_out_conv<NX0, FP0, kNT, typename Src::Conv>(a, 0, A.mode[0], cur.t, ob.o0, lds, tid, o0, cur.dyA[0], cur.dyB[0]);
            scale_out_conv<NX1, FP1, kNT, typename Src::Conv>(a, 1, A.mode[1], cur.t, ob.o1, lds, tid, o1, cur.dyA[1], cur.dyB[1]);
        }
        IPX_STAMP(4);
        __syncthreads();
        IPX_STAMP(5);

        if (!has_next) break;
        if (nxt.cb != cur.cb && a.nscale > 0) {
            load_xtaps<NX0, FP0, kNT>(a, 0, nxt.cb, tid, o0);
            load_xtaps<NX1, FP1, kNT>(a, 1, nxt.cb, tid, o1);
        }
        if (nxt.f != cur.f) ob = out_bases(a, nxt.f);
        cur = nxt;
        cur_carry = nxt_carry;
        idx++;
        left--;
    }
#if IPX_DIAG
    if (a.stamps && (tid & 63) == 0)
        for (int i = 0; i < 6; i++) atomicAdd(&a.stamps[i], acc[i]);
#endif
#undef IPX_STAMP
}

template <int NX0, bool FP0, int NX1, bool FP1, class Src>
hipError_t launch_conv(const typename Src::Args &A, const char *what, long long items, size_t lds, hipStream_t s)
{
    static KernelLaunchCache cache;
    int resident = 1;
    auto kern = band_conv_kernel<NX0, FP0, NX1, FP1, Src>;
    hipError_t e = cache.prepare((const void *)kern, kNT, lds, &resident);
    if (e != hipSuccess) return e;
    const long long grid = std::min<long long>(items, (long long)A.b.cus * std::min(A.b.pipe_wgs, resident));
    if (getenv("IPX_DEBUG") && cache.first_report()) {
        fprintf(stderr, "[ipx] band_conv_kernel<%d,%d,%d,%d,%s>: tile %d rows x %d cols, lds %zu B, resident %d/CU, grid %lld, items %lld\n",
                NX0, (int)FP0, NX1, (int)FP1, what, A.b.band_rows, A.b.blk_cols, lds, resident, grid, items);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kNT), lds, s, A);
    return hipGetLastError();
}

template <class Src>
hipError_t launch_conv_cfg(const typename Src::Args &A, const char *what, long long items, size_t lds, hipStream_t s, bool *matched)
{
    const BandArgs &a = A.b;
    // a.nx_out counts blocks of 256 destination columns per column block; the 512-thread workgroup serves two each
    const int need0 = a.nscale > 0 ? (a.nx_out[0] + 1) / 2 : 0, need1 = a.nscale > 1 ? (a.nx_out[1] + 1) / 2 : 0;
    const bool fp0 = a.nscale > 0 && a.sc[0].dyadic_shift < 0;
    *matched = true;
    if (need0 <= 1 && !fp0 && need1 <= 1) return launch_conv<1, false, 1, true, Src>(A, what, items, lds, s);
    if (need0 <= 1 && need1 <= 1) return launch_conv<1, true, 1, true, Src>(A, what, items, lds, s);
    *matched = false;
    return hipSuccess;
}

bool conv_tiling_ok(const BandArgs &a)
{
    if ((a.sw & 3) || a.band_rows + 1 > kRows || (a.blk_cols + 4) * 4 > kConvTilePitch || (a.blk_cols & 3)) return false;
    if (a.wm && ((((uintptr_t)a.wm) | a.wm_frame_stride | (uintptr_t)a.wm_stride) & 15)) return false;
    return true;
}

}  // namespace

// Tile shapes and alignments band_conv_kernel is built for on YCbCr planes; anything else takes the three-kernel path.
bool band_ycc_supported(const YccArgs &A)
{
    const BandArgs &a = A.b;
    const int hs = A.ratio == IPX_YCBCR_422 || A.ratio == IPX_YCBCR_420, vs = A.ratio == IPX_YCBCR_420 || A.ratio == IPX_YCBCR_440;
    if (!conv_tiling_ok(a)) return false;
    if (vs && (a.band_rows & 1)) return false;
    if ((((uintptr_t)A.y) | (uintptr_t)A.ystride | A.y_fs) & 3) return false;
    const uintptr_t cal = hs ? 1 : 3;
    if ((((uintptr_t)A.cb) | ((uintptr_t)A.cr) | (uintptr_t)A.cstride | A.c_fs) & cal) return false;
    return true;
}

hipError_t launch_band_ycc(const YccArgs &A, hipStream_t s, bool *matched)
{
    const BandArgs &a = A.b;
    *matched = false;
    const long long total = (long long)a.nbands * a.ncolblk * a.nframes;
    if (total <= 0) { *matched = true; return hipSuccess; }
    const size_t lds = 2 * (size_t)kRows * kConvTilePitch;     // the two planes of the converted tile (y taps come through scalar loads)
    if (A.ratio == IPX_GRAY) {                                 // the Y plane alone (ipx_plan_run_dev_gray, Gray JPEGs)
        if (total > 0x7fffffffLL || !conv_tiling_ok(a) || ((((uintptr_t)A.y) | (uintptr_t)A.ystride | A.y_fs) & 3)) return hipSuccess;
        return launch_conv_cfg<GraySrc>(A, "gray", total, lds, s, matched);
    }
    if (total > 0x7fffffffLL || !band_ycc_supported(A)) return hipSuccess;
    switch (A.ratio) {
    case IPX_YCBCR_444: return launch_conv_cfg<YccSrc<0, 0>>(A, "ycc 4:4:4", total, lds, s, matched);
    case IPX_YCBCR_422: return launch_conv_cfg<YccSrc<1, 0>>(A, "ycc 4:2:2", total, lds, s, matched);
    case IPX_YCBCR_420: return launch_conv_cfg<YccSrc<1, 1>>(A, "ycc 4:2:0", total, lds, s, matched);
    case IPX_YCBCR_440: return launch_conv_cfg<YccSrc<0, 1>>(A, "ycc 4:4:0", total, lds, s, matched);
    default: return hipSuccess;
    }
}

// *image.NRGBA frames through the same kernel (premultiplied 16-bit taps in the tile); the plan's `conv` tiling.  Not matched: the
// per-tap kernel of ipx_band_nrgba.hip on the plan's other tiling, then the three-kernel path.
hipError_t launch_band_nrgba_conv(const NrgbaArgs &A, hipStream_t s, bool *matched)
{
    const BandArgs &a = A.b;
    *matched = false;
    const long long total = (long long)a.nbands * a.ncolblk * a.nframes;
    if (total <= 0) { *matched = true; return hipSuccess; }
    if (total > 0x7fffffffLL || !conv_tiling_ok(a)) return hipSuccess;
    if ((((uintptr_t)a.src) | (uintptr_t)a.sstride | a.src_frame_stride) & 15) return hipSuccess;
    return launch_conv_cfg<NrgbaSrc>(A, "nrgba", total, 2 * (size_t)kRows * kConvTilePitch, s, matched);
}

// The deep source types straight from Go's Pix (a.src / a.sstride / a.src_frame_stride describe it).  Not matched (tile shape, alignment):
// the expansion pass and launch_band_tap64_conv, then the three-kernel path.
hipError_t launch_band_deep(const NrgbaArgs &A, int kind, hipStream_t s, bool *matched)
{
    const BandArgs &a = A.b;
    *matched = false;
    const long long total = (long long)a.nbands * a.ncolblk * a.nframes;
    if (total <= 0) { *matched = true; return hipSuccess; }
    if (total > 0x7fffffffLL || !conv_tiling_ok(a)) return hipSuccess;
    const uintptr_t al = kind == IPX_DEEP_GRAY16 ? 7 : 15;          // a chunk's load: 8 bytes of Gray16, 16 (twice for the 64-bit types) otherwise
    if ((((uintptr_t)a.src) | (uintptr_t)a.sstride | a.src_frame_stride) & al) return hipSuccess;
    const size_t lds = 2 * (size_t)kRows * kConvTilePitch;
    switch (kind) {
    case IPX_DEEP_NRGBA64: return launch_conv_cfg<DeepSrc<IPX_DEEP_NRGBA64>>(A, "nrgba64", total, lds, s, matched);
    case IPX_DEEP_RGBA64: return launch_conv_cfg<DeepSrc<IPX_DEEP_RGBA64>>(A, "rgba64", total, lds, s, matched);
    case IPX_DEEP_GRAY16: return launch_conv_cfg<DeepSrc<IPX_DEEP_GRAY16>>(A, "gray16", total, lds, s, matched);
    case IPX_DEEP_CMYK: return launch_conv_cfg<DeepSrc<IPX_DEEP_CMYK>>(A, "cmyk", total, lds, s, matched);
    default: return hipSuccess;
    }
}

// Frames of 16-bit taps (the deep source types after their expansion, ipx_plan_run_dev_deep): a.src / a.sstride / a.src_frame_stride
// describe 8-byte pixels.  Not matched: the three-kernel path.
hipError_t launch_band_tap64_conv(const NrgbaArgs &A, hipStream_t s, bool *matched)
{
    const BandArgs &a = A.b;
    *matched = false;
    const long long total = (long long)a.nbands * a.ncolblk * a.nframes;
    if (total <= 0) { *matched = true; return hipSuccess; }
    if (total > 0x7fffffffLL || !conv_tiling_ok(a)) return hipSuccess;
    if ((((uintptr_t)a.src) | (uintptr_t)a.sstride | a.src_frame_stride) & 15) return hipSuccess;
    if ((long long)(a.sh - 1) * a.sstride + (long long)a.sw * 8 > 0x7fffffffLL - 2 * kConvTilePitch) return hipSuccess;   // one descriptor spans the frame
    return launch_conv_cfg<Tap64Src>(A, "tap64", total, 2 * (size_t)kRows * kConvTilePitch, s, matched);
}

}  // namespace ipx

// ipx_band_common.h -- device code shared by the band kernels (ipx_band.hip): the tile geometry of a
// work item, the tap lerps, and the composite / scale steps that read the LDS tile.
// Include after `#pragma clang fp contract(off)` and ipx_device.h.
#pragma once

namespace ipx {
namespace {

#ifndef IPX_DIAG_STORES
#ifdef IPX_DIAG
#define IPX_DIAG_STORES IPX_DIAG
#else
#define IPX_DIAG_STORES 0
#endif
#endif
#ifndef IPX_WAVE_SKIP
#define IPX_WAVE_SKIP 1
#endif

constexpr int kYChunk = 64;  // destination rows whose y taps sit in LDS at a time
constexpr int kLoadU = 4;    // 16-byte loads in flight per lane in band_kernel's phase 1

typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
constexpr int kOOB = 0x7fffffff;  // a byte offset beyond any frame: buffer loads return 0, stores are dropped

// Exact lerp for dyadic weights: x0+x1 = 2^kx, y0+y1 = 2^ky (small integers held in floats), taps
// as plain bytes.  sum = y0*(x0*t00 + x1*t10) + y1*(x0*t01 + x1*t11) < 2^(8+kx+ky) <= 2^24, so fp32
// (fused or not) is exact; the reference's float64 value is 257*sum / 2^(kx+ky), truncated, >> 8.
// sh = kx + ky + 8.
__device__ __forceinline__ uint32_t lerp_dyadic(uint32_t p00, uint32_t p10, uint32_t p01,
                                                uint32_t p11, float x0, float x1, float y0, float y1,
                                                int sh)
{
    const v2f X0 = {x0, x0}, X1 = {x1, x1}, Y0 = {y0, y0}, Y1 = {y1, y1};
    auto lo = [](uint32_t p) { return v2f{(float)(p & 0xffu), (float)((p >> 8) & 0xffu)}; };
    auto hi = [](uint32_t p) { return v2f{(float)((p >> 16) & 0xffu), (float)(p >> 24)}; };
    const v2f top_l = __builtin_elementwise_fma(X1, lo(p10), X0 * lo(p00));
    const v2f bot_l = __builtin_elementwise_fma(X1, lo(p11), X0 * lo(p01));
    const v2f top_h = __builtin_elementwise_fma(X1, hi(p10), X0 * hi(p00));
    const v2f bot_h = __builtin_elementwise_fma(X1, hi(p11), X0 * hi(p01));
    const v2f vl = __builtin_elementwise_fma(Y1, bot_l, Y0 * top_l);
    const v2f vh = __builtin_elementwise_fma(Y1, bot_h, Y0 * top_h);
    const uint32_t r = (uint32_t)vl.x, g = (uint32_t)vl.y, b = (uint32_t)vh.x, a = (uint32_t)vh.y;
    return ((r * 257u) >> sh) | (((g * 257u) >> sh) << 8) | (((b * 257u) >> sh) << 16) |
           (((a * 257u) >> sh) << 24);
}

__device__ __forceinline__ uint32_t lds_u32(const uint8_t *lds, int off)
{
    return *(const uint32_t *)(lds + off);
}

struct Tile {
    int r0, r1, c0, c1;          // owned rows / columns
    int rows_ld, cols_ld;        // with the halo row / column
    int own_rows, own_cols, pitch, nchunk;
};

__device__ __forceinline__ Tile make_tile(const BandArgs &a, int b, int cb)
{
    Tile t;
    t.r0 = b * a.band_rows;
    t.r1 = min(t.r0 + a.band_rows, a.sh);
    t.c0 = cb * a.blk_cols;
    t.c1 = min(t.c0 + a.blk_cols, a.sw);
    t.rows_ld = min(t.r1 + 1, a.sh) - t.r0;
    t.cols_ld = min(t.c1 + 1, a.sw) - t.c0;
    t.pitch = (a.blk_cols + 4) * 4;                // LDS bytes per tile row
    t.nchunk = (t.cols_ld + 3) >> 2;               // 16-byte chunks per tile row
    t.own_rows = t.r1 - t.r0;
    t.own_cols = t.c1 - t.c0;
    return t;
}

__device__ __forceinline__ bool chunk_in_textbox(const BandArgs &a, int x, int y)
{
    return y >= a.gbox.y0 && y < a.gbox.y1 && x + 4 > a.gbox.x0 && x < a.gbox.x1;
}

__device__ __forceinline__ bool tile_meets_textbox(const BandArgs &a, const Tile &t)
{
    return t.r0 < a.gbox.y1 && t.r1 > a.gbox.y0 && t.c0 < a.gbox.x1 && t.c1 > a.gbox.x0;
}

// Step 3: glyph composite over the tile's share of the text box, source pixels from LDS (read as RGBA8 through Conv).
// A thread takes one pixel COLUMN of that share and keeps the column's pixels (one per tile row, MAXR >= owned rows) in registers:
// the glyph list is walked once per column, not once per pixel, four descriptors per step (scalar loads, issued together), and a
// glyph whose rectangle holds the column loads its mask bytes for all rows at once.  Per pixel the glyphs still apply in string order.
// (The first version walked the list per pixel: a scalar load and a test per glyph and pixel, every one a full latency -- 25 us per
// tile that meets the text, 7% of a 1080p run.)
typedef const __attribute__((address_space(4))) DevGlyph *ConstGlyphs;
#ifndef IPX_GLYPH_STEP
#define IPX_GLYPH_STEP 4
#endif
constexpr int kGlyphStep = IPX_GLYPH_STEP;   // glyph descriptors fetched together (scalar loads): latency against SGPR pressure

template <int NT, int MAXR, class Conv>
__device__ __forceinline__ void glyph_phase(const BandArgs &a, const Tile &t, uint8_t *wframe, const uint8_t *lds, int tid)
{
    const int by0 = max(a.gbox.y0, t.r0), by1 = min(a.gbox.y1, t.r1);
    const int gx0 = max(a.gbox.x0 & ~3, t.c0), gx1 = min((a.gbox.x1 + 3) & ~3, t.c1);  // whole skipped chunks
    const int plane = Conv::plane(t);
    const ConstGlyphs gl = (ConstGlyphs)(uintptr_t)a.glyphs;
    for (int gy0 = by0; gy0 < by1; gy0 += MAXR) {          // (one round: MAXR covers the rows a tile owns, except in band_kernel's tall tiles)
        const int gy1 = min(gy0 + MAXR, by1);
        for (int x = gx0 + tid; x < gx1; x += NT) {
            uint32_t d[MAXR];
#pragma unroll
            for (int r = 0; r < MAXR; r++) d[r] = Conv::rgba8_at(lds, (min(gy0 + r, gy1 - 1) - t.r0) * t.pitch + (x - t.c0) * 4, plane);
            for (int g = 0; g < a.nglyphs; g += kGlyphStep) {
                int rx0[kGlyphStep], rx1[kGlyphStep], ry0[kGlyphStep], ry1[kGlyphStep], ms[kGlyphStep];
                const uint8_t *mk[kGlyphStep];
#pragma unroll
                for (int j = 0; j < kGlyphStep; j++) {
                    const int gi = min(g + j, a.nglyphs - 1);
                    rx0[j] = gl[gi].x0; rx1[j] = g + j < a.nglyphs ? gl[gi].x1 : gl[gi].x0;   // (an empty rectangle past the end of the list)
                    ry0[j] = gl[gi].y0; ry1[j] = min(gl[gi].y1, gy1); ms[j] = gl[gi].mstride; mk[j] = gl[gi].mask;
                }
#pragma unroll
                for (int j = 0; j < kGlyphStep; j++) {
                    if (x < rx0[j] || x >= rx1[j]) continue;
                    uint32_t m[MAXR];
#pragma unroll
                    for (int r = 0; r < MAXR; r++) {
                        const int y = gy0 + r;
                        m[r] = y >= ry0[j] && y < ry1[j] ? mk[j][(size_t)(y - ry0[j]) * ms[j] + (x - rx0[j])] : 0u;
                    }
#pragma unroll
                    for (int r = 0; r < MAXR; r++)
                        if (m[r]) d[r] = glyph_over(d[r], m[r], a.cr, a.cg, a.cb, a.ca);
                }
            }
#pragma unroll
            for (int r = 0; r < MAXR; r++)
                if (gy0 + r < gy1) *(uint32_t *)(wframe + (size_t)(gy0 + r) * a.wm_stride + (size_t)x * 4) = d[r];
        }
    }
}

// Destination rows [dyA, dyB) of band b per scaled output: four scalar loads issued together (one wait, far downstream), not a load and a
// wait per value -- that chain was 2000 cycles of every item's issue phase.  (a.sc[1] mirrors a.sc[0] when one output is scaled.)
typedef const __attribute__((address_space(4))) int *ConstRowBegin;
__device__ __forceinline__ void band_out_rows(const BandArgs &a, int b, bool valid, int (&dyA)[2], int (&dyB)[2])
{
    dyA[0] = dyA[1] = dyB[0] = dyB[1] = 0;
    if (a.nscale > 0) {
        const ConstRowBegin rb0 = (ConstRowBegin)(uintptr_t)a.sc[0].row_begin, rb1 = (ConstRowBegin)(uintptr_t)a.sc[1].row_begin;
        const int a0 = rb0[b], b0 = rb0[b + 1], a1 = rb1[b], b1 = rb1[b + 1];
        dyA[0] = a0; dyA[1] = a1;
        dyB[0] = valid ? b0 : a0;
        dyB[1] = valid && a.nscale > 1 ? b1 : a1;
    }
}

// Where the frames of one batch index start: 64-bit products of the frame index, recomputed only when the index changes (a
// workgroup walks the bands of a frame one after the other), not per item.
struct OutBases { uint8_t *wm, *o0, *o1; };
__device__ __forceinline__ OutBases out_bases(const BandArgs &a, int f)
{
    OutBases b;
    b.wm = a.wm ? a.wm + (size_t)f * a.wm_frame_stride : nullptr;
    b.o0 = a.sc[0].out + (size_t)f * a.sc[0].frame_stride;
    b.o1 = a.sc[1].out + (size_t)f * a.sc[1].frame_stride;
    return b;
}

// ---- per-output state a thread keeps in registers -------------------------------------------------
// A thread serves destination columns dxA + tid + NT*i, i < NX, of one scaled output (NT = threads of
// the workgroup).  FP = the
// output may need the float64 lerp (non-dyadic axis), so the float64 x weights are kept as well; an
// output known to be dyadic keeps two floats and an index per column.
template <bool FP>
struct XTapT;
template <>
struct XTapT<true> {
    double w0, w1;
    float f0, f1;
    int base;
    uint32_t iw;
};
template <>
struct XTapT<false> {
    float f0, f1;
    int base;
    uint32_t iw;
};

template <int NX, bool FP>
struct OutCols {
    XTapT<FP> tx[NX];
    int dxA, dxB;
};

// x taps of column block cb.  Loads are unconditional (index clamped into the table).
// (a.sc[1] mirrors a.sc[0] when only one output is scaled, so the loads are always legal.)
template <int NX, bool FP, int NT = 256>
__device__ __forceinline__ void load_xtaps(const BandArgs &a, int k, int cb, int tid, OutCols<NX, FP> &o)
{
    const ScaleOut &S = a.sc[k];
    o.dxA = S.col_begin[cb];
    o.dxB = k < a.nscale ? S.col_begin[cb + 1] : o.dxA;
#pragma unroll
    for (int i = 0; i < NX; i++) {
        const AxisTap *p = &S.xt[min(o.dxA + tid + NT * i, S.dw - 1)];
        if constexpr (FP) { o.tx[i].w0 = p->w0; o.tx[i].w1 = p->w1; }
        o.tx[i].f0 = p->f0; o.tx[i].f1 = p->f1; o.tx[i].base = p->base; o.tx[i].iw = p->iw;
    }
}

// ---- the packed-integer lerp (dyadic axes, RGBA8 taps) -------------------------------------------------
// Same exact value as lerp_dyadic -- sum = y0*(x0*t00 + x1*t10) + y1*(x0*t01 + x1*t11), byte = (257*sum) >> (kx+ky+8) -- in
// integer arithmetic shaped for few instructions:
//   H   one tile row lerped horizontally for a thread's column, two channels per register (16-bit lanes: 8 + kx <= 16 bits):
//       h_rb = x0*(p0 & 0x00ff00ff) + x1*(p1 & 0x00ff00ff), h_ga likewise on (p >> 8).  A thread's columns are fixed, so an H row
//       is computed once and serves the output row above and the one below it (vertical scale < 2: 1.4 H rows per output row
//       at 1080 -> 768 instead of 2);
//   V   per channel one v_dot2_u32_u16 on the pair [H_top.c | H_bot.c << 16] and the packed weights [y0 | y1 << 16];
//   out byte = mul_hi_u24(sum', 257 << (24 - k')): the y weights are pre-scaled so that k' = kx + ky' >= 9, which puts the byte
//       (257*sum) >> (k + 8) into bits 32..39 of a 24 x 24 bit product.
// Straight-line: every lane computes, dead lanes store to an out-of-range buffer offset.  Rows are walked in order; their tap
// rows never decrease, so two H rows (top, bottom) with a scalar tag are all the state there is.
struct HRow { uint32_t rb, ga; };

// How a scale path reads a tap from the LDS tile.  A tile is one or two planes of a dword per pixel with the same pitch (plane(t) =
// byte distance between them); `off` is the byte offset of the pixel within a plane.
struct TapAsIs {   // the tile holds RGBA8 pixels
    static __device__ __forceinline__ int plane(const Tile &) { return 0; }
    static __device__ __forceinline__ uint32_t rgba8_at(const uint8_t *lds, int off, int) { return lds_u32(lds, off); }
};

template <class Conv = TapAsIs>
__device__ __forceinline__ HRow h_row(const uint8_t *lds, int off, int plane, uint32_t x0, uint32_t x1)
{
    const uint32_t p0 = Conv::rgba8_at(lds, off, plane), p1 = Conv::rgba8_at(lds, off + 4, plane);
    HRow h;
    h.rb = __umul24(p0 & 0x00ff00ffu, x0) + __umul24(p1 & 0x00ff00ffu, x1);
    // v_perm_b32: bytes {p.1, 0, p.3, 0} = (p >> 8) & 0x00ff00ff in one instruction
    h.ga = __umul24(__builtin_amdgcn_perm(0u, p0, 0x0c030c01u), x0) + __umul24(__builtin_amdgcn_perm(0u, p1, 0x0c030c01u), x1);
    return h;
}

typedef unsigned short v2us __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t dot2_u16(uint32_t a, uint32_t b)
{
    return __builtin_amdgcn_udot2(__builtin_bit_cast(v2us, a), __builtin_bit_cast(v2us, b), 0u, false);
}
// out[i] = byte r | byte g << 8 | byte b << 16 | byte a << 24 with byte c = mul_hi_u24(sum[i][c], m) (<= 255): four instructions per pixel,
// the multiply writing its low byte straight into byte c of the result (SDWA dst_sel, the other bytes preserved).  gfx940+ need one
// wait state between a dst_sel write and a VALU read of that register (LLVM's DstSelForwardingHazard; the compiler cannot see into
// inline asm): the NX pixels of a thread are interleaved inside ONE asm statement per channel, so a register's next use is at least
// NX >= 2 instructions away, and a lone pixel gets an s_nop.
template <int NX>
__device__ __forceinline__ void finish_bytes(const uint32_t (&sum)[NX][4], uint32_t m, uint32_t (&out)[NX])
{
    static_assert(NX == 1 || NX == 2 || NX == 4, "column counts the band kernels are built for");
#define IPX_SDWA(B) "v_mul_hi_u32_u24_sdwa %0, %1, %2 dst_sel:BYTE_" #B " dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n"
    if constexpr (NX == 1) {
        asm("v_mul_hi_u32_u24_e32 %0, %1, %2" : "=v"(out[0]) : "v"(m), "v"(sum[0][0]));
        asm(IPX_SDWA(1) "s_nop 0" : "+v"(out[0]) : "v"(m), "v"(sum[0][1]));
        asm(IPX_SDWA(2) "s_nop 0" : "+v"(out[0]) : "v"(m), "v"(sum[0][2]));
        asm(IPX_SDWA(3) "s_nop 0" : "+v"(out[0]) : "v"(m), "v"(sum[0][3]));
    } else if constexpr (NX == 2) {
#define IPX_SDWA2(B) "v_mul_hi_u32_u24_sdwa %0, %2, %3 dst_sel:BYTE_" #B " dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n" \
                     "v_mul_hi_u32_u24_sdwa %1, %2, %4 dst_sel:BYTE_" #B " dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
        asm("v_mul_hi_u32_u24_e32 %0, %2, %3\nv_mul_hi_u32_u24_e32 %1, %2, %4" : "=&v"(out[0]), "=&v"(out[1]) : "v"(m), "v"(sum[0][0]), "v"(sum[1][0]));
        asm(IPX_SDWA2(1) : "+v"(out[0]), "+v"(out[1]) : "v"(m), "v"(sum[0][1]), "v"(sum[1][1]));
        asm(IPX_SDWA2(2) : "+v"(out[0]), "+v"(out[1]) : "v"(m), "v"(sum[0][2]), "v"(sum[1][2]));
        asm(IPX_SDWA2(3) : "+v"(out[0]), "+v"(out[1]) : "v"(m), "v"(sum[0][3]), "v"(sum[1][3]));
#undef IPX_SDWA2
    } else {
#define IPX_SDWA4(B) "v_mul_hi_u32_u24_sdwa %0, %4, %5 dst_sel:BYTE_" #B " dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n" \
                     "v_mul_hi_u32_u24_sdwa %1, %4, %6 dst_sel:BYTE_" #B " dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n" \
                     "v_mul_hi_u32_u24_sdwa %2, %4, %7 dst_sel:BYTE_" #B " dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n" \
                     "v_mul_hi_u32_u24_sdwa %3, %4, %8 dst_sel:BYTE_" #B " dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
        asm("v_mul_hi_u32_u24_e32 %0, %4, %5\nv_mul_hi_u32_u24_e32 %1, %4, %6\nv_mul_hi_u32_u24_e32 %2, %4, %7\nv_mul_hi_u32_u24_e32 %3, %4, %8"
            : "=&v"(out[0]), "=&v"(out[1]), "=&v"(out[2]), "=&v"(out[3]) : "v"(m), "v"(sum[0][0]), "v"(sum[1][0]), "v"(sum[2][0]), "v"(sum[3][0]));
        asm(IPX_SDWA4(1) : "+v"(out[0]), "+v"(out[1]), "+v"(out[2]), "+v"(out[3]) : "v"(m), "v"(sum[0][1]), "v"(sum[1][1]), "v"(sum[2][1]), "v"(sum[3][1]));
        asm(IPX_SDWA4(2) : "+v"(out[0]), "+v"(out[1]), "+v"(out[2]), "+v"(out[3]) : "v"(m), "v"(sum[0][2]), "v"(sum[1][2]), "v"(sum[2][2]), "v"(sum[3][2]));
        asm(IPX_SDWA4(3) : "+v"(out[0]), "+v"(out[1]), "+v"(out[2]), "+v"(out[3]) : "v"(m), "v"(sum[0][3]), "v"(sum[1][3]), "v"(sum[2][3]), "v"(sum[3][3]));
#undef IPX_SDWA4
    }
#undef IPX_SDWA
}

struct IntRow { uint32_t ctl, yw; };
typedef const __attribute__((address_space(4))) IntRow *ConstRows;
template <int NX, bool FP, int NT, class Conv = TapAsIs>
__device__ __forceinline__ void scale_rows_int(const ScaleOut &S, const Tile &t, const uint8_t *lds, int tid,
                                               const OutCols<NX, FP> &o, __amdgpu_buffer_rsrc_t ors, int dyA, int dyB)
{
    const int xbias = S.sr_x0 - t.c0, plane = Conv::plane(t);
    const uint32_t m3 = S.imul;
    uint32_t x0[NX], x1[NX];
    int lx[NX], voff[NX];
#pragma unroll
    for (int i = 0; i < NX; i++) {
        const int dx = o.dxA + tid + NT * i;
        lx[i] = (xbias + o.tx[i].base) * 4;
        x0[i] = o.tx[i].iw & 0xffffu; x1[i] = o.tx[i].iw >> 16;
        voff[i] = dx < o.dxB ? dx * 4 : kOOB;
    }
    HRow top[NX], bot[NX];
#pragma unroll
    for (int i = 0; i < NX; i++) top[i] = bot[i] = HRow{0u, 0u};
    // the row table is read through the constant address space: scalar loads, one row ahead of the arithmetic
    ConstRows yr = (ConstRows)(uintptr_t)S.yrow + dyA;
    IntRow nx = {yr[0].ctl, yr[0].yw};
    int soff = dyA * S.ostride;
    for (int n = dyB - dyA; n > 0; n--) {
        const IntRow row = nx;
        ++yr;
        nx = IntRow{yr[0].ctl, yr[0].yw};            // (the table carries one entry past the last row)
        const int off = (int)(row.ctl & 0x0fffffffu);
        if (row.ctl >> 28) {                         // 0: the same pair of tile rows as the row before (an upscaled axis)
            if (row.ctl >> 29) {                     // 2: a gap -- the row that becomes the upper one is not at hand
#pragma unroll
                for (int i = 0; i < NX; i++) bot[i] = h_row<Conv>(lds, off + lx[i], plane, x0[i], x1[i]);
            }
#pragma unroll
            for (int i = 0; i < NX; i++) { top[i] = bot[i]; bot[i] = h_row<Conv>(lds, off + t.pitch + lx[i], plane, x0[i], x1[i]); }
        }
        uint32_t sum[NX][4], v[NX];
#pragma unroll
        for (int i = 0; i < NX; i++) {               // [top.c | bot.c << 16] per channel, then y0 * top.c + y1 * bot.c
            sum[i][0] = dot2_u16(__builtin_amdgcn_perm(bot[i].rb, top[i].rb, 0x05040100u), row.yw);
            sum[i][1] = dot2_u16(__builtin_amdgcn_perm(bot[i].ga, top[i].ga, 0x05040100u), row.yw);
            sum[i][2] = dot2_u16(__builtin_amdgcn_perm(bot[i].rb, top[i].rb, 0x07060302u), row.yw);
            sum[i][3] = dot2_u16(__builtin_amdgcn_perm(bot[i].ga, top[i].ga, 0x07060302u), row.yw);
        }
        finish_bytes<NX>(sum, m3, v);
#pragma unroll
        for (int i = 0; i < NX; i++) __builtin_amdgcn_raw_buffer_store_b32(v[i], ors, voff[i], soff, 0);
        soff += S.ostride;
    }
}

// Step 2 for one scaled output: its destination rows [dyA, dyB) x this thread's columns, from the
// LDS tile.  Global memory is touched with stores only: on gfx9 loads and stores share vmcnt in issue
// order, and a vector load in here would wait for every pixel store before it.  The y taps are wave-uniform: they are
// read from the plan's table through the constant address space, i.e. with scalar loads into SGPRs (lgkmcnt, not vmcnt; no
// registers or LDS spent on staging them), the next row's tap while the current row computes.  Rows outermost, the NX columns
// innermost and unrolled, so the NX x 4 tap reads are issued together and their LDS latency overlaps.  Straight-line: every lane
// reads (clamped taps keep the addresses inside the tile) and computes; a lane without a column stores to an out-of-range buffer
// offset, which the hardware drops -- no exec-mask branch around the arithmetic.
typedef const __attribute__((address_space(4))) AxisTap *ConstTaps;
__device__ __forceinline__ ConstTaps const_taps(const AxisTap *p) { return (ConstTaps)(uintptr_t)p; }
struct YTapF32 { int base; float f0, f1; };
struct YTapF64 { int base; double w0, w1; };

template <int NX, bool FP, int NT = 256>
__device__ __forceinline__ void scale_out(const BandArgs &a, int k, const Tile &t, uint8_t *oframe, const uint8_t *lds,
                                          int tid, const OutCols<NX, FP> &o, int dyA, int dyB)
{
    if (k >= a.nscale || dyA >= dyB) return;
    // a wave none of whose lanes has a destination column leaves (wave-uniform: the thumbnail's 200 columns keep 4 of a workgroup's 8
    // waves busy, 2 of 8 on the narrow tiles of band_conv_kernel; the others ran the whole row loop to store nothing)
#if IPX_WAVE_SKIP
    if (o.dxA + __builtin_amdgcn_readfirstlane(tid & ~63) >= o.dxB) return;
#endif
    const ScaleOut &S = a.sc[k];
#if IPX_DIAG_STORES
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void *)oframe, 0, (a.dbg & 32) ? 0 : S.dh * S.ostride, 0x00020000);   // 32: every store dropped
#else
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void *)oframe, 0, S.dh * S.ostride, 0x00020000);
#endif
    const ConstTaps yt = const_taps(S.yt);
    if (S.imul) {   // dyadic axes within the packed-integer limits: the common case (1080p, 4K, 8K, 720p, 1440p to 1024 wide)
        scale_rows_int<NX, FP, NT>(S, t, lds, tid, o, ors, dyA, dyB);
        return;
    }
    const int ybias = S.sr_y0 - t.r0, xbias = S.sr_x0 - t.c0;
    int lx[NX], voff[NX];
#pragma unroll
    for (int i = 0; i < NX; i++) {
        const int dx = o.dxA + tid + NT * i;
        lx[i] = (xbias + o.tx[i].base) * 4;
        voff[i] = dx < o.dxB ? dx * 4 : kOOB;
    }
    int soff = dyA * S.ostride;
    if (!FP || S.dyadic_shift >= 0) {
        // both axes dyadic but beyond the integer path's lanes: every product and sum is exact in fp32 (8 + kx + ky <= 24 bits)
        // and 257 * sum / 2^(kx+ky) is the reference's float64 value
        const int sh = S.dyadic_shift + 8;
        YTapF32 nx = {yt[dyA].base, yt[dyA].f0, yt[dyA].f1};
        for (int dy = dyA; dy < dyB; dy++) {
            const YTapF32 y = nx;
            const int dn = min(dy + 1, dyB - 1);
            nx = YTapF32{yt[dn].base, yt[dn].f0, yt[dn].f1};
            const int rowoff = (ybias + y.base) * t.pitch;
            uint32_t p[NX][4];
#pragma unroll
            for (int i = 0; i < NX; i++) {
                const int off = rowoff + lx[i];
                p[i][0] = lds_u32(lds, off); p[i][1] = lds_u32(lds, off + 4);
                p[i][2] = lds_u32(lds, off + t.pitch); p[i][3] = lds_u32(lds, off + t.pitch + 4);
            }
#pragma unroll
            for (int i = 0; i < NX; i++) {
                const uint32_t v = lerp_dyadic(p[i][0], p[i][1], p[i][2], p[i][3], o.tx[i].f0, o.tx[i].f1, y.f0, y.f1, sh);
                __builtin_amdgcn_raw_buffer_store_b32(v, ors, voff[i], soff, 0);
            }
            soff += S.ostride;
        }
    } else if constexpr (FP) {
        YTapF64 nx = {yt[dyA].base, yt[dyA].w0, yt[dyA].w1};
        for (int dy = dyA; dy < dyB; dy++) {
            const YTapF64 y = nx;
            const int dn = min(dy + 1, dyB - 1);
            nx = YTapF64{yt[dn].base, yt[dn].w0, yt[dn].w1};
            const int rowoff = (ybias + y.base) * t.pitch;
#pragma unroll
            for (int i = 0; i < NX; i++) {
                const int off = rowoff + lx[i];
                const uint32_t p00 = lds_u32(lds, off), p10 = lds_u32(lds, off + 4);
                const uint32_t p01 = lds_u32(lds, off + t.pitch), p11 = lds_u32(lds, off + t.pitch + 4);
                const double xw0 = o.tx[i].w0, xw1 = o.tx[i].w1;
                const uint32_t pr = lerp_channel<0>(p00, p10, p01, p11, xw0, xw1, y.w0, y.w1);
                const uint32_t pg = lerp_channel<1>(p00, p10, p01, p11, xw0, xw1, y.w0, y.w1);
                const uint32_t pb = lerp_channel<2>(p00, p10, p01, p11, xw0, xw1, y.w0, y.w1);
                const uint32_t pa = lerp_channel<3>(p00, p10, p01, p11, xw0, xw1, y.w0, y.w1);
                __builtin_amdgcn_raw_buffer_store_b32(pack_src(pr, pg, pb, pa), ors, voff[i], soff, 0);
            }
            soff += S.ostride;
        }
    }
}

// ---- scaled outputs of the kernels whose tile holds a converted source type (ipx_band_conv.hip, ipx_band_nrgba.hip) ----
// Conv::NC        3: the converted alpha is constant 0xffff (YCbCr), 4: it is a channel like the others (NRGBA)
// Conv::tap16_at  a tile pixel -> the 16-bit channels the reference's scale_RGBA_<type>_* interpolates (mode 0)
// Conv::h16       the horizontal step of mode 0 on dyadic axes: h.c = x0 * tap(off).c + x1 * tap(off + 4).c, iw = x0 | x1 << 16
// Conv::rgba8_at  a tile pixel -> the RGBA8 pixel of the reference's copy / draw routine for the type (mode 1: the crop thumbnail scales
//                 the RGBA8 copy cropAndResize made, thumbnail.go:128-131)
// DwordConv<D>    all of these for a tile of one dword per pixel that D::tap16 / D::rgba8 convert when a tap is read
// Mode 0 on dyadic axes (kx, ky <= 8) is exact in u32: h.c = x0*t0.c + x1*t1.c (< 2^24), sum.c = y0'*h_top.c + y1'*h_bot.c with the y
// weights scaled by 2^(16-kx-ky), so that the output byte is the TOP byte of the 32-bit sum; the H rows are reused between output
// rows exactly as in the packed-integer lerp above (two converted taps per H row instead of four per pixel).
template <class D>
struct DwordConv {
    static __device__ __forceinline__ int plane(const Tile &) { return 0; }
    static __device__ __forceinline__ uint32_t rgba8_at(const uint8_t *lds, int off, int) { return D::rgba8(lds_u32(lds, off)); }
    template <int N>
    static __device__ __forceinline__ void tap16_at(const uint8_t *lds, int off, int, uint32_t (&c)[N]) { D::tap16(lds_u32(lds, off), c); }
    template <int N>
    static __device__ __forceinline__ void h16(const uint8_t *lds, int off, int, uint32_t iw, uint32_t (&h)[N])
    {
        uint32_t t0[N], t1[N];
        D::tap16(lds_u32(lds, off), t0);
        D::tap16(lds_u32(lds, off + 4), t1);
        const uint32_t x0 = iw & 0xffffu, x1 = iw >> 16;
#pragma unroll
        for (int j = 0; j < N; j++) h[j] = __umul24(x0, t0[j]) + __umul24(x1, t1[j]);
    }
};

struct IntRow16 { uint32_t ctl, y0, y1, pad; };
typedef const __attribute__((address_space(4))) IntRow16 *ConstRows16;

template <int NX, bool FP, int NT, class Conv>
__device__ __forceinline__ void scale_rows_int16(const ScaleOut &S, const Tile &t, const uint8_t *lds, int tid,
                                                 const OutCols<NX, FP> &o, __amdgpu_buffer_rsrc_t ors, int dyA, int dyB)
{
    constexpr int NC = Conv::NC;
    const int xbias = S.sr_x0 - t.c0, plane = Conv::plane(t);
    int lx[NX], voff[NX];
#pragma unroll
    for (int i = 0; i < NX; i++) {
        const int dx = o.dxA + tid + NT * i;
        lx[i] = (xbias + o.tx[i].base) * 4;
        voff[i] = dx < o.dxB ? dx * 4 : kOOB;
    }
    uint32_t top[NX][NC], bot[NX][NC];
#pragma unroll
    for (int i = 0; i < NX; i++)
#pragma unroll
        for (int j = 0; j < NC; j++) top[i][j] = bot[i][j] = 0;
    auto h16 = [&](int off, int i, uint32_t (&h)[NC]) { Conv::h16(lds, off, plane, o.tx[i].iw, h); };
    ConstRows16 yr = (ConstRows16)(uintptr_t)S.yrow16 + dyA;
    IntRow16 nx = {yr[0].ctl, yr[0].y0, yr[0].y1, 0};
    int soff = dyA * S.ostride;
    for (int n = dyB - dyA; n > 0; n--) {
        const IntRow16 row = nx;
        ++yr;
        nx = IntRow16{yr[0].ctl, yr[0].y0, yr[0].y1, 0};     // (the table carries one entry past the last row)
        const int off = (int)(row.ctl & 0x0fffffffu);
        if (row.ctl >> 28) {
            if (row.ctl >> 29) {
#pragma unroll
                for (int i = 0; i < NX; i++) h16(off + lx[i], i, bot[i]);
            }
#pragma unroll
            for (int i = 0; i < NX; i++) {
#pragma unroll
                for (int j = 0; j < NC; j++) top[i][j] = bot[i][j];
                h16(off + t.pitch + lx[i], i, bot[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < NX; i++) {
            uint32_t sum[NC];
#pragma unroll
            for (int j = 0; j < NC; j++) sum[j] = __umul24(row.y0, top[i][j]) + __umul24(row.y1, bot[i][j]);
            const uint32_t rg = __builtin_amdgcn_perm(sum[1], sum[0], 0x0c0c0703u);               // {r, g, 0, 0}: the top bytes
            uint32_t v;
            if constexpr (NC == 3) v = __builtin_amdgcn_perm(sum[2], rg, 0x0d070100u);            // {r, g, b, 0xff}
            else v = rg | __builtin_amdgcn_perm(sum[3], sum[2], 0x07030c0cu);
            __builtin_amdgcn_raw_buffer_store_b32(v, ors, voff[i], soff, 0);
        }
        soff += S.ostride;
    }
}



__device__ __forceinline__ uint32_t lerp16_f64(uint32_t s00, uint32_t s10, uint32_t s01, uint32_t s11, double xw0, double xw1, double yw0,
                                               double yw1)
{
    const double top = xw0 * (double)s00 + xw1 * (double)s10;
    const double bot = xw0 * (double)s01 + xw1 * (double)s11;
    return (uint32_t)(yw0 * top + yw1 * bot);
}

template <int NX, bool FP, int NT, class Conv>
__device__ __forceinline__ void scale_out_conv(const BandArgs &a, int k, int mode, const Tile &t, uint8_t *oframe, const uint8_t *lds, int tid,
                                               const OutCols<NX, FP> &o, int dyA, int dyB)
{
    if (k >= a.nscale || dyA >= dyB) return;
    // a wave none of whose lanes has a destination column leaves (wave-uniform: the thumbnail's 200 columns keep 4 of a workgroup's 8
    // waves busy, 2 of 8 on the narrow tiles of band_conv_kernel; the others ran the whole row loop to store nothing)
#if IPX_WAVE_SKIP
    if (o.dxA + __builtin_amdgcn_readfirstlane(tid & ~63) >= o.dxB) return;
#endif
    const ScaleOut &S = a.sc[k];
#if IPX_DIAG_STORES
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void *)oframe, 0, (a.dbg & 32) ? 0 : S.dh * S.ostride, 0x00020000);   // 32: every store dropped
#else
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void *)oframe, 0, S.dh * S.ostride, 0x00020000);
#endif
    if (mode == 0 && S.yrow16) { scale_rows_int16<NX, FP, NT, Conv>(S, t, lds, tid, o, ors, dyA, dyB); return; }
    if (mode == 1 && S.imul) { scale_rows_int<NX, FP, NT, Conv>(S, t, lds, tid, o, ors, dyA, dyB); return; }
    const ConstTaps yt = const_taps(S.yt);
    const int ybias = S.sr_y0 - t.r0, xbias = S.sr_x0 - t.c0, plane = Conv::plane(t);
    int lx[NX], voff[NX];
#pragma unroll
    for (int i = 0; i < NX; i++) {
        const int dx = o.dxA + tid + NT * i;
        lx[i] = (xbias + o.tx[i].base) * 4;
        voff[i] = dx < o.dxB ? dx * 4 : kOOB;
    }
    int soff = dyA * S.ostride;
    if (mode == 1 && (!FP || S.dyadic_shift >= 0)) {       // dyadic beyond the integer path's lanes: exact fp32 on the converted RGBA8 taps
        const int sh = S.dyadic_shift + 8;
        for (int dy = dyA; dy < dyB; dy++) {
            const int rowoff = (ybias + yt[dy].base) * t.pitch;
            const float yf0 = yt[dy].f0, yf1 = yt[dy].f1;
#pragma unroll
            for (int i = 0; i < NX; i++) {
                const int off = rowoff + lx[i];
                const uint32_t p00 = Conv::rgba8_at(lds, off, plane), p10 = Conv::rgba8_at(lds, off + 4, plane);
                const uint32_t p01 = Conv::rgba8_at(lds, off + t.pitch, plane), p11 = Conv::rgba8_at(lds, off + t.pitch + 4, plane);
                __builtin_amdgcn_raw_buffer_store_b32(lerp_dyadic(p00, p10, p01, p11, o.tx[i].f0, o.tx[i].f1, yf0, yf1, sh), ors, voff[i], soff, 0);
            }
            soff += S.ostride;
        }
    } else if constexpr (FP) {                              // the reference's float64 lerp (a host-side rule turns mode 0 on dyadic axes beyond 8 bits into this as well)
        for (int dy = dyA; dy < dyB; dy++) {
            const int rowoff = (ybias + yt[dy].base) * t.pitch;
            const double yw0 = yt[dy].w0, yw1 = yt[dy].w1;
#pragma unroll
            for (int i = 0; i < NX; i++) {
                const int off = rowoff + lx[i];
                const double xw0 = o.tx[i].w0, xw1 = o.tx[i].w1;
                uint32_t v;
                if (mode == 0) {
                    uint32_t t00[Conv::NC], t10[Conv::NC], t01[Conv::NC], t11[Conv::NC], pc[4];
                    Conv::tap16_at(lds, off, plane, t00); Conv::tap16_at(lds, off + 4, plane, t10);
                    Conv::tap16_at(lds, off + t.pitch, plane, t01); Conv::tap16_at(lds, off + t.pitch + 4, plane, t11);
#pragma unroll
                    for (int j = 0; j < Conv::NC; j++) pc[j] = lerp16_f64(t00[j], t10[j], t01[j], t11[j], xw0, xw1, yw0, yw1);
                    if constexpr (Conv::NC == 3) pc[3] = 0xffffu;
                    v = pack_src(pc[0], pc[1], pc[2], pc[3]);
                } else {
                    const uint32_t p00 = Conv::rgba8_at(lds, off, plane), p10 = Conv::rgba8_at(lds, off + 4, plane);
                    const uint32_t p01 = Conv::rgba8_at(lds, off + t.pitch, plane), p11 = Conv::rgba8_at(lds, off + t.pitch + 4, plane);
                    const uint32_t pr = lerp_channel<0>(p00, p10, p01, p11, xw0, xw1, yw0, yw1);
                    const uint32_t pg = lerp_channel<1>(p00, p10, p01, p11, xw0, xw1, yw0, yw1);
                    const uint32_t pb = lerp_channel<2>(p00, p10, p01, p11, xw0, xw1, yw0, yw1);
                    const uint32_t pa = lerp_channel<3>(p00, p10, p01, p11, xw0, xw1, yw0, yw1);
                    v = pack_src(pr, pg, pb, pa);
                }
                __builtin_amdgcn_raw_buffer_store_b32(v, ors, voff[i], soff, 0);
            }
            soff += S.ostride;
        }
    }
}

}  // namespace
}  // namespace ipx

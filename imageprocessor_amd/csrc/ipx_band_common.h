// ipx_band_common.h -- device code shared by the band kernels (ipx_band.hip, ipx_ring.hip): the tile
// geometry of a work item, the tap lerps, and the composite / scale steps that read the LDS tile.
// Include after `#pragma clang fp contract(off)` and ipx_device.h.
#pragma once

namespace ipx {
namespace {

#ifndef IPX_ROW_UNROLL
#define IPX_ROW_UNROLL 1
#endif
constexpr int kYChunk = 64;  // destination rows whose y taps sit in LDS at a time
constexpr int kLoadU = 4;    // 16-byte loads in flight per lane in band_kernel's phase 1

typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
constexpr int kOOB = 0x7fffffff;  // a byte offset beyond any frame
#ifndef IPX_AUX_LOAD
#define IPX_AUX_LOAD 0
#endif
#ifndef IPX_AUX_WM
#define IPX_AUX_WM 0
#endif
#ifndef IPX_NT_PX
#define IPX_NT_PX 0
#endif
#ifndef IPX_WM_FROM_LDS
#define IPX_WM_FROM_LDS 0
#endif
__device__ __forceinline__ void store_px(uint32_t *p, uint32_t v)
{
    if (IPX_NT_PX) __builtin_nontemporal_store(v, p); else *p = v;
}

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

// the part of an AxisTap a thread keeps in registers for its destination column
struct XTap {
    double w0, w1;
    float f0, f1;
    int base;
};

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

// x taps of column block cb for this thread: destination columns dxA + tid + NT*i.
// (a.sc[1] mirrors a.sc[0] when only one output is scaled, so both loads are always legal.)
template <int NX, int NT = 256>
__device__ __forceinline__ void load_xtaps(const BandArgs &a, int cb, int tid, XTap (&tx)[2][NX],
                                           int (&dxA)[2], int (&dxB)[2])
{
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const ScaleOut &S = a.sc[k];
        dxA[k] = S.col_begin[cb];
        dxB[k] = k < a.nscale ? S.col_begin[cb + 1] : dxA[k];
#pragma unroll
        for (int i = 0; i < NX; i++) {
            const AxisTap xt = S.xt[min(dxA[k] + tid + NT * i, S.dw - 1)];
            tx[k][i].w0 = xt.w0; tx[k][i].w1 = xt.w1; tx[k][i].f0 = xt.f0; tx[k][i].f1 = xt.f1;
            tx[k][i].base = xt.base;
        }
    }
}

// Step 3: glyph composite over the block's share of the text box, source pixels from LDS.
template <int NT = 256>
__device__ __forceinline__ void glyph_phase(const BandArgs &a, const Tile &t, uint8_t *wframe,
                                            const uint8_t *lds, int tid)
{
    const int gy0 = max(a.gbox.y0, t.r0), gy1 = min(a.gbox.y1, t.r1);
    const int gx0 = max(a.gbox.x0 & ~3, t.c0), gx1 = min((a.gbox.x1 + 3) & ~3, t.c1);  // whole skipped chunks
    const int gw = gx1 - gx0, gn = gw * (gy1 - gy0);
    for (int i = tid; i < gn; i += NT) {
        const int yy = i / gw, x = gx0 + (i - yy * gw), y = gy0 + yy;
        uint32_t d = lds_u32(lds, (y - t.r0) * t.pitch + (x - t.c0) * 4);
        d = glyph_run(d, x, y, a.glyphs, a.nglyphs, a.cr, a.cg, a.cb, a.ca);
        *(uint32_t *)(wframe + (size_t)y * a.wm_stride + (size_t)x * 4) = d;
    }
}

__device__ __forceinline__ bool tile_meets_textbox(const BandArgs &a, const Tile &t)
{
    return t.r0 < a.gbox.y1 && t.r1 > a.gbox.y0 && t.c0 < a.gbox.x1 && t.c1 > a.gbox.x0;
}

// Step 2: scaled outputs from the LDS tile.  Global memory is touched with stores only: on gfx9
// loads and stores share vmcnt in issue order, and a load in here would wait for every pixel store
// before it.  The y taps of the block's destination rows sit in LDS (wave-uniform reads).
template <int NX, int NT = 256, bool REFILL = true>
__device__ __forceinline__ void scale_phase(const BandArgs &a, const Tile &t, int f, const uint8_t *lds,
                                            AxisTap *ytap, int tid, const XTap (&tx)[2][NX],
                                            const int (&dxA)[2], const int (&dxB)[2],
                                            const int (&dyA)[2], const int (&dyB)[2])
{
#pragma unroll
    for (int k = 0; k < 2; k++) {
        if (k >= a.nscale || dyA[k] >= dyB[k]) continue;
        const ScaleOut &S = a.sc[k];
        uint8_t *oframe = S.out + (size_t)f * S.frame_stride;
        const int ybias = S.sr_y0 - t.r0, xbias = S.sr_x0 - t.c0;
        for (int chunk = dyA[k]; chunk < dyB[k]; chunk += kYChunk) {
            const int rows = min(kYChunk, dyB[k] - chunk);
            if (REFILL && chunk != dyA[k]) {  // more destination rows than one chunk: refill the y taps
                __syncthreads();
                ytap[k * kYChunk + (tid & (kYChunk - 1))] = S.yt[min(chunk + (tid & (kYChunk - 1)), S.dh - 1)];
                __syncthreads();
            }
            const AxisTap *yt = ytap + k * kYChunk;
#pragma unroll
            for (int i = 0; i < NX; i++) {
                const int dx = dxA[k] + tid + NT * i;
                if (dx >= dxB[k]) continue;
                const XTap &X = tx[k][i];
                const int lx = (xbias + X.base) * 4;
                uint32_t *op = (uint32_t *)(oframe + (size_t)chunk * S.ostride + (size_t)dx * 4);
                if (S.dyadic_shift >= 0) {
                    // both axes dyadic: every product and sum is exact in fp32 (8 + kx + ky <= 24
                    // bits) and 257 * sum / 2^(kx+ky) is the reference's float64 value
                    const int sh = S.dyadic_shift + 8;
#pragma unroll IPX_ROW_UNROLL
                    for (int r = 0; r < rows; r++, op = (uint32_t *)((uint8_t *)op + S.ostride)) {
                        const int ybase = __builtin_amdgcn_readfirstlane(yt[r].base);
                        const float yf0 = yt[r].f0, yf1 = yt[r].f1;
                        const int off = (ybias + ybase) * t.pitch + lx;
                        const uint32_t p00 = lds_u32(lds, off), p10 = lds_u32(lds, off + 4);
                        const uint32_t p01 = lds_u32(lds, off + t.pitch), p11 = lds_u32(lds, off + t.pitch + 4);
                        store_px(op, lerp_dyadic(p00, p10, p01, p11, X.f0, X.f1, yf0, yf1, sh));
                    }
                } else {
                    for (int r = 0; r < rows; r++, op = (uint32_t *)((uint8_t *)op + S.ostride)) {
                        const int ybase = yt[r].base;
                        const double yw0 = yt[r].w0, yw1 = yt[r].w1;
                        const int off = (ybias + ybase) * t.pitch + lx;
                        const uint32_t p00 = lds_u32(lds, off), p10 = lds_u32(lds, off + 4);
                        const uint32_t p01 = lds_u32(lds, off + t.pitch), p11 = lds_u32(lds, off + t.pitch + 4);
                        const uint32_t pr = lerp_channel<0>(p00, p10, p01, p11, X.w0, X.w1, yw0, yw1);
                        const uint32_t pg = lerp_channel<1>(p00, p10, p01, p11, X.w0, X.w1, yw0, yw1);
                        const uint32_t pb = lerp_channel<2>(p00, p10, p01, p11, X.w0, X.w1, yw0, yw1);
                        const uint32_t pa = lerp_channel<3>(p00, p10, p01, p11, X.w0, X.w1, yw0, yw1);
                        store_px(op, pack_src(pr, pg, pb, pa));
                    }
                }
            }
        }
    }
}

}  // namespace
}  // namespace ipx

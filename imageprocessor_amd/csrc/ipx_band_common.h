// ipx_band_common.h -- device code shared by the band kernels (ipx_band.hip): the tile geometry of a
// work item, the tap lerps, and the composite / scale steps that read the LDS tile.
// Include after `#pragma clang fp contract(off)` and ipx_device.h.
#pragma once

namespace ipx {
namespace {

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
};
template <>
struct XTapT<false> {
    float f0, f1;
    int base;
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
        o.tx[i].f0 = p->f0; o.tx[i].f1 = p->f1; o.tx[i].base = p->base;
    }
}

// Step 2 for one scaled output: its destination rows [dyA, dyB) x this thread's columns, from the
// LDS tile.  Global memory is touched with stores only: on gfx9 loads and stores share vmcnt in issue
// order, and a load in here would wait for every pixel store before it.  The y taps of the rows sit
// in LDS (wave-uniform reads).  Rows outermost, the NX columns innermost and unrolled: a row's y tap
// is fetched once for NX pixels and the NX x 4 tap reads are issued together, so their LDS latency
// overlaps.  All reads are unconditional (clamped taps keep the addresses inside the tile); only the
// store is predicated.
template <int NX, bool FP, bool REFILL, int NT = 256>
__device__ __forceinline__ void scale_out(const BandArgs &a, int k, const Tile &t, int f, const uint8_t *lds,
                                          AxisTap *ytap_k, int tid, const OutCols<NX, FP> &o, int dyA, int dyB)
{
    if (k >= a.nscale || dyA >= dyB) return;
    const ScaleOut &S = a.sc[k];
    uint8_t *oframe = S.out + (size_t)f * S.frame_stride;
    const int ybias = S.sr_y0 - t.r0, xbias = S.sr_x0 - t.c0;
    for (int chunk = dyA; chunk < dyB; chunk += kYChunk) {
        const int rows = min(kYChunk, dyB - chunk);
        if (REFILL && chunk != dyA) {  // more destination rows than one chunk: refill the y taps
            __syncthreads();
            ytap_k[tid & (kYChunk - 1)] = S.yt[min(chunk + (tid & (kYChunk - 1)), S.dh - 1)];
            __syncthreads();
        }
        bool live[NX];
        int lx[NX];
        uint32_t *op[NX];
#pragma unroll
        for (int i = 0; i < NX; i++) {
            const int dx = o.dxA + tid + NT * i;
            live[i] = dx < o.dxB;
            lx[i] = (xbias + o.tx[i].base) * 4;
            op[i] = (uint32_t *)(oframe + (size_t)chunk * S.ostride + (size_t)dx * 4);
        }
        if (!FP || S.dyadic_shift >= 0) {
            // both axes dyadic: every product and sum is exact in fp32 (8 + kx + ky <= 24 bits) and
            // 257 * sum / 2^(kx+ky) is the reference's float64 value
            const int sh = S.dyadic_shift + 8;
            for (int r = 0; r < rows; r++) {
                const int rowoff = (ybias + __builtin_amdgcn_readfirstlane(ytap_k[r].base)) * t.pitch;
                const float yf0 = ytap_k[r].f0, yf1 = ytap_k[r].f1;
                uint32_t p[NX][4];
#pragma unroll
                for (int i = 0; i < NX; i++) {
                    const int off = rowoff + lx[i];
                    p[i][0] = lds_u32(lds, off); p[i][1] = lds_u32(lds, off + 4);
                    p[i][2] = lds_u32(lds, off + t.pitch); p[i][3] = lds_u32(lds, off + t.pitch + 4);
                }
#pragma unroll
                for (int i = 0; i < NX; i++) {
                    const uint32_t v = lerp_dyadic(p[i][0], p[i][1], p[i][2], p[i][3], o.tx[i].f0, o.tx[i].f1, yf0, yf1, sh);
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
                    const uint32_t p00 = lds_u32(lds, off), p10 = lds_u32(lds, off + 4);
                    const uint32_t p01 = lds_u32(lds, off + t.pitch), p11 = lds_u32(lds, off + t.pitch + 4);
                    const double xw0 = o.tx[i].w0, xw1 = o.tx[i].w1;
                    const uint32_t pr = lerp_channel<0>(p00, p10, p01, p11, xw0, xw1, yw0, yw1);
                    const uint32_t pg = lerp_channel<1>(p00, p10, p01, p11, xw0, xw1, yw0, yw1);
                    const uint32_t pb = lerp_channel<2>(p00, p10, p01, p11, xw0, xw1, yw0, yw1);
                    const uint32_t pa = lerp_channel<3>(p00, p10, p01, p11, xw0, xw1, yw0, yw1);
                    if (live[i]) *op[i] = pack_src(pr, pg, pb, pa);
                    op[i] = (uint32_t *)((uint8_t *)op[i] + S.ostride);
                }
            }
        }
    }
}

}  // namespace
}  // namespace ipx

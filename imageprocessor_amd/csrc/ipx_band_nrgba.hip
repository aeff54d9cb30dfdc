// ipx_band_nrgba.hip -- the fused band kernel for *image.NRGBA batches (PNGs with alpha; *image.Paletted frames after their
// palette expansion).
//
// Same decomposition as band_pipe_kernel (ipx_band.hip) and band_ycc_kernel (ipx_band_ycc.hip): a persistent workgroup walks
// (frame, band, column block) items, keeps the next item's loads in flight while it computes the current one from LDS, and one
// pass over the source produces the watermark frame and both scaled outputs.  The LDS tile holds the source pixels as they are
// (non-premultiplied R, G, B, A) and every consumer applies the conversion the reference's routine for its operator applies
// (image_processor.go:47 hands each operator the *image.NRGBA itself):
//   * watermark: draw.Draw(result, b, img, Point{}, draw.Src) (watermark.go:92) = image/draw drawNRGBASrc: sa = a*0x101,
//     c*sa/0xff >> 8 per channel -> done in registers on the way to the store;
//   * crop thumbnail: the equal-size Scale of cropAndResize (thumbnail.go:128-130) is a Copy = drawNRGBAOver onto a zeroed
//     frame (the same bytes as drawNRGBASrc), and resizeImage then scales that RGBA8 copy with scale_RGBA_RGBA_* -> taps
//     premultiplied to RGBA8 first (mode 1);
//   * resize and the non-crop thumbnail: resizeImage on the NRGBA itself = scale_RGBA_NRGBA_*: every TAP premultiplied to
//     16 bit (a16 = a*0x101, c*a16/0xff) and interpolated in float64 -> mode 0; on dyadic axes the float64 value is
//     sum(w*tap) / 2^(kx+ky) exactly, computed here in u32.  Over onto the zeroed output is Src.
//
// Bound: HBM (1080p, full pipeline: 8.3 MB in, 11.6 MB out per frame), with the premultiplication (two integer divisions by
// 0xff per channel and tap, as multiply-high) close behind for the resized output.
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

// drawNRGBASrc for one pixel: sa = a*0x101; c = uint8(c*sa/0xff >> 8); alpha = uint8(sa >> 8) = a
__device__ __forceinline__ uint32_t nrgba_rgba8(uint32_t p)
{
    const uint32_t sa = (p >> 24) * 0x101u;
    const uint32_t r = ((p & 0xffu) * sa / 0xffu) >> 8, g = (((p >> 8) & 0xffu) * sa / 0xffu) >> 8, b = (((p >> 16) & 0xffu) * sa / 0xffu) >> 8;
    return r | (g << 8) | (b << 16) | (p & 0xff000000u);
}
// a tap as scale_RGBA_NRGBA_* reads it
struct Rgba16 { uint32_t r, g, b, a; };
__device__ __forceinline__ Rgba16 nrgba_tap16(uint32_t p)
{
    Rgba16 t;
    t.a = (p >> 24) * 0x101u;
    t.r = (p & 0xffu) * t.a / 0xffu;
    t.g = ((p >> 8) & 0xffu) * t.a / 0xffu;
    t.b = ((p >> 16) & 0xffu) * t.a / 0xffu;
    return t;
}
__device__ __forceinline__ uint32_t lerp16_f64(uint32_t s00, uint32_t s10, uint32_t s01, uint32_t s11, double xw0, double xw1, double yw0,
                                               double yw1)
{
    const double top = xw0 * (double)s00 + xw1 * (double)s10;
    const double bot = xw0 * (double)s01 + xw1 * (double)s11;
    return (uint32_t)(yw0 * top + yw1 * bot);
}
// dyadic axes, integer weights x0 + x1 = 2^kx <= 256, y0 + y1 = 2^ky <= 256: exact in u32, operands < 2^24
__device__ __forceinline__ uint32_t lerp16_int(uint32_t s00, uint32_t s10, uint32_t s01, uint32_t s11, uint32_t x0, uint32_t x1, uint32_t y0,
                                               uint32_t y1, int sh)
{
    const uint32_t top = __umul24(x0, s00) + __umul24(x1, s10);
    const uint32_t bot = __umul24(x0, s01) + __umul24(x1, s11);
    return (__umul24(y0, top) + __umul24(y1, bot)) >> sh;
}

struct Stage { v4u px[kRows]; };

__device__ __forceinline__ void issue_tile(const BandArgs &a, const Tile &t, int f, bool valid, int tid, Stage &st)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(a.src + (size_t)f * a.src_frame_stride), 0,
                                                                        (a.sh - 1) * a.sstride + a.sw * 4, 0x00020000);
    const int rows = valid ? t.rows_ld : 0;
    const bool in_tile = tid < t.nchunk;
    const int off = t.r0 * a.sstride + t.c0 * 4 + tid * 16;
#pragma unroll
    for (int r = 0; r < kRows; r++)   // r < rows is wave-uniform
        st.px[r] = __builtin_amdgcn_raw_buffer_load_b128(rs, in_tile && r < rows ? off + r * a.sstride : kOOB, 0, 0);
}

// staged pixels -> LDS tile as they are, and the owned pixels premultiplied -> watermark frame
__device__ __forceinline__ void drain_tile(const BandArgs &a, const Tile &t, int f, int tid, const Stage &st, uint8_t *lds, bool any_glyph)
{
    uint8_t *wframe = a.wm ? a.wm + (size_t)f * a.wm_frame_stride : nullptr;
    const int wm_bytes = wframe ? (a.sh - 1) * a.wm_stride + a.sw * 4 : 0;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)wframe, 0, wm_bytes, 0x00020000);
    const bool gl_rows = any_glyph && t.r0 < a.gbox.y1 && t.r1 > a.gbox.y0;   // wave-uniform
    const bool in_tile = tid < t.nchunk;
    const bool owned = wframe && tid * 4 < t.own_cols;
    const int woff = t.r0 * a.wm_stride + t.c0 * 4 + tid * 16;
    const int loff = tid * 16;
#pragma unroll
    for (int r = 0; r < kRows; r++) {
        const v4u p = st.px[r];
        if (r < t.rows_ld && in_tile) *(v4u *)(lds + r * t.pitch + loff) = p;
        v4u rgba;
#pragma unroll
        for (int i = 0; i < 4; i++) rgba[i] = nrgba_rgba8(p[i]);
        int off = r < t.own_rows && owned ? woff + r * a.wm_stride : kOOB;
        if (gl_rows && chunk_in_textbox(a, t.c0 + tid * 4, t.r0 + r)) off = kOOB;   // chunks that meet the text box are written by the composite step
        __builtin_amdgcn_raw_buffer_store_b128(rgba, wrs, off, 0, 0);
    }
}

__device__ __forceinline__ void glyph_phase_nrgba(const BandArgs &a, const Tile &t, uint8_t *wframe, const uint8_t *lds, int tid)
{
    const int gy0 = max(a.gbox.y0, t.r0), gy1 = min(a.gbox.y1, t.r1);
    const int gx0 = max(a.gbox.x0 & ~3, t.c0), gx1 = min((a.gbox.x1 + 3) & ~3, t.c1);  // whole skipped chunks
    const int gw = gx1 - gx0, gn = gw * (gy1 - gy0);
    for (int i = tid; i < gn; i += kNT) {
        const int yy = i / gw, x = gx0 + (i - yy * gw), y = gy0 + yy;
        uint32_t d = nrgba_rgba8(lds_u32(lds, (y - t.r0) * t.pitch + (x - t.c0) * 4));
        d = glyph_run(d, x, y, a.glyphs, a.nglyphs, a.cr, a.cg, a.cb, a.ca);
        *(uint32_t *)(wframe + (size_t)y * a.wm_stride + (size_t)x * 4) = d;
    }
}

// One scaled output from the tile.  mode 0: taps -> 16-bit premultiplied, then interpolate (scale_RGBA_NRGBA_*);
// mode 1: taps -> premultiplied RGBA8 first (the crop copy), then scale_RGBA_RGBA_*.
template <int NX, bool FP>
__device__ __forceinline__ void scale_out_nrgba(const BandArgs &a, int k, int mode, const Tile &t, int f, const uint8_t *lds,
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
                const Rgba16 t00 = nrgba_tap16(p[i][0]), t10 = nrgba_tap16(p[i][1]), t01 = nrgba_tap16(p[i][2]), t11 = nrgba_tap16(p[i][3]);
                const uint32_t pr = lerp16_int(t00.r, t10.r, t01.r, t11.r, x0[i], x1[i], y0, y1, sh);
                const uint32_t pg = lerp16_int(t00.g, t10.g, t01.g, t11.g, x0[i], x1[i], y0, y1, sh);
                const uint32_t pb = lerp16_int(t00.b, t10.b, t01.b, t11.b, x0[i], x1[i], y0, y1, sh);
                const uint32_t pa = lerp16_int(t00.a, t10.a, t01.a, t11.a, x0[i], x1[i], y0, y1, sh);
                if (live[i]) *op[i] = pr | (pg << 8) | (pb << 16) | (pa << 24);
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
                const uint32_t p00 = nrgba_rgba8(lds_u32(lds, off)), p10 = nrgba_rgba8(lds_u32(lds, off + 4));
                const uint32_t p01 = nrgba_rgba8(lds_u32(lds, off + t.pitch)), p11 = nrgba_rgba8(lds_u32(lds, off + t.pitch + 4));
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
                    const Rgba16 t00 = nrgba_tap16(q00), t10 = nrgba_tap16(q10), t01 = nrgba_tap16(q01), t11 = nrgba_tap16(q11);
                    const uint32_t pr = lerp16_f64(t00.r, t10.r, t01.r, t11.r, xw0, xw1, yw0, yw1);
                    const uint32_t pg = lerp16_f64(t00.g, t10.g, t01.g, t11.g, xw0, xw1, yw0, yw1);
                    const uint32_t pb = lerp16_f64(t00.b, t10.b, t01.b, t11.b, xw0, xw1, yw0, yw1);
                    const uint32_t pa = lerp16_f64(t00.a, t10.a, t01.a, t11.a, xw0, xw1, yw0, yw1);
                    v = pack_src(pr, pg, pb, pa);
                } else {
                    const uint32_t p00 = nrgba_rgba8(q00), p10 = nrgba_rgba8(q10), p01 = nrgba_rgba8(q01), p11 = nrgba_rgba8(q11);
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

struct ItemN {
    int f, b, cb;
    Tile t;
    int dyA[2], dyB[2];
};

__device__ __forceinline__ void item_setup(const BandArgs &a, ItemN &it, bool valid)
{
    it.t = make_tile(a, it.b, it.cb);
#pragma unroll
    for (int k = 0; k < 2; k++) {
        it.dyA[k] = a.nscale > 0 ? a.sc[k].row_begin[it.b] : 0;
        it.dyB[k] = valid && k < a.nscale ? a.sc[k].row_begin[it.b + 1] : it.dyA[k];
    }
}

template <int NX0, bool FP0, int NX1, bool FP1>
__global__ __launch_bounds__(kNT) void band_nrgba_kernel(NrgbaArgs A)
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

    auto decode = [&](int i, ItemN &it) {
        it.cb = i / per_cb;
        it.f = (i - it.cb * per_cb) / a.nbands;
        it.b = i - it.cb * per_cb - it.f * a.nbands;
    };
    ItemN cur;
    decode(idx, cur);
    item_setup(a, cur, true);

    OutCols<NX0, FP0> o0;
    OutCols<NX1, FP1> o1;
    if (a.nscale > 0) { load_xtaps<NX0, FP0, kNT>(a, 0, cur.cb, tid, o0); load_xtaps<NX1, FP1, kNT>(a, 1, cur.cb, tid, o1); }

    Stage st;
    v4u ty_stage[2][2];
    auto issue_ytaps = [&](const ItemN &it) {
        if (a.nscale > 0) {
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const v4u *yp = (const v4u *)&a.sc[k].yt[min(it.dyA[k] + tid, a.sc[k].dh - 1)];
                ty_stage[k][0] = yp[0]; ty_stage[k][1] = yp[1];
            }
        }
    };
    issue_tile(a, cur.t, cur.f, true, tid, st);
    issue_ytaps(cur);

    for (;;) {
        // A: staged pixels -> LDS tile + premultiplied watermark pixels; y taps -> LDS
        drain_tile(a, cur.t, cur.f, tid, st, lds, any_glyph);
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
        ItemN nxt;
        const bool has_next = idx + G < items;
        if (has_next) decode(idx + G, nxt);
        else { nxt.b = cur.b; nxt.f = cur.f; nxt.cb = cur.cb; }
        item_setup(a, nxt, has_next);
        issue_tile(a, nxt.t, nxt.f, has_next, tid, st);
        issue_ytaps(nxt);

        // C: the current item from LDS
        if (any_glyph && tile_meets_textbox(a, cur.t))
            glyph_phase_nrgba(a, cur.t, a.wm + (size_t)cur.f * a.wm_frame_stride, lds, tid);
        if (a.nscale > 0) {
            scale_out_nrgba<NX0, FP0>(a, 0, A.mode[0], cur.t, cur.f, lds, ytap, tid, o0, cur.dyA[0], cur.dyB[0]);
            scale_out_nrgba<NX1, FP1>(a, 1, A.mode[1], cur.t, cur.f, lds, ytap + kYChunk, tid, o1, cur.dyA[1], cur.dyB[1]);
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

template <int NX0, bool FP0, int NX1, bool FP1>
hipError_t launch_cfg(const NrgbaArgs &A, long long items, size_t lds, hipStream_t s)
{
    static KernelLaunchCache cache;
    int resident = 1;
    auto kern = band_nrgba_kernel<NX0, FP0, NX1, FP1>;
    hipError_t e = cache.prepare((const void *)kern, kNT, lds, &resident);
    if (e != hipSuccess) return e;
    const long long grid = std::min<long long>(items, (long long)A.b.cus * std::min(A.b.pipe_wgs, resident));
    if (getenv("IPX_DEBUG") && cache.first_report()) {
        fprintf(stderr, "[ipx] band_nrgba_kernel<%d,%d,%d,%d>: tile %d rows x %d cols, lds %zu B, resident %d/CU, grid %lld, items %lld\n", NX0,
                (int)FP0, NX1, (int)FP1, A.b.band_rows, A.b.blk_cols, lds, resident, grid, items);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kNT), lds, s, A);
    return hipGetLastError();
}

}  // namespace

// Tile shapes and alignments the fused NRGBA kernel is built for; anything else takes the three-kernel path.
bool band_nrgba_supported(const NrgbaArgs &A)
{
    const BandArgs &a = A.b;
    if ((a.sw & 3) || a.band_rows + 1 > kRows || a.blk_cols / 4 + 1 > kNT || (a.blk_cols & 3)) return false;
    if ((((uintptr_t)a.src) | (uintptr_t)a.sstride | a.src_frame_stride) & 15) return false;
    if (a.wm && ((((uintptr_t)a.wm) | a.wm_frame_stride | (uintptr_t)a.wm_stride) & 15)) return false;
    return true;
}

hipError_t launch_band_nrgba(const NrgbaArgs &A, hipStream_t s, bool *matched)
{
    const BandArgs &a = A.b;
    *matched = false;
    const long long total = (long long)a.nbands * a.ncolblk * a.nframes;
    if (total <= 0) { *matched = true; return hipSuccess; }
    if (total > 0x7fffffffLL || !band_nrgba_supported(A)) return hipSuccess;
    const size_t lds = band_lds_bytes(a.band_rows, a.blk_cols);
    // a.nx_out counts blocks of 256 destination columns per column block; a 512-thread workgroup serves two each
    const int need0 = a.nscale > 0 ? (a.nx_out[0] + 1) / 2 : 0, need1 = a.nscale > 1 ? (a.nx_out[1] + 1) / 2 : 0;
    const bool fp0 = a.nscale > 0 && a.sc[0].dyadic_shift < 0;
    *matched = true;
    if (need0 <= 2 && !fp0 && need1 <= 1) return launch_cfg<2, false, 1, true>(A, total, lds, s);
    if (need0 <= 2 && need1 <= 1) return launch_cfg<2, true, 1, true>(A, total, lds, s);
    *matched = false;
    return hipSuccess;
}

}  // namespace ipx

// ipx_band_nrgba.hip -- the PER-TAP band kernel for *image.NRGBA batches (PNGs with alpha; *image.Paletted frames after their
// palette expansion).  Since round 2 such batches take band_conv_kernel<..., NrgbaSrc> (ipx_band_conv.hip: every
// source pixel premultiplied once into a tile of 16-bit taps); this kernel is what they fall back to when the plan has no `conv`
// tiling or IPX_NRGBA_CONV=0 says so, and the tests keep it exercised.
//
// Same decomposition as band_pipe_kernel (ipx_band.hip) and band_conv_kernel (ipx_band_conv.hip): a persistent workgroup walks
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
// how the shared scale paths (ipx_band_common.h, scale_out_conv) read a tile dword (a non-premultiplied pixel)
struct NrgbaTaps {
    // a tap as scale_RGBA_NRGBA_* reads it: a16 = a * 0x101, c * a16 / 0xff
    static __device__ __forceinline__ void tap16(uint32_t p, uint32_t (&c)[4])
    {
        c[3] = (p >> 24) * 0x101u;
        c[0] = (p & 0xffu) * c[3] / 0xffu;
        c[1] = ((p >> 8) & 0xffu) * c[3] / 0xffu;
        c[2] = ((p >> 16) & 0xffu) * c[3] / 0xffu;
    }
    static __device__ __forceinline__ uint32_t rgba8(uint32_t p) { return nrgba_rgba8(p); }
};
struct NrgbaConv : DwordConv<NrgbaTaps> { static constexpr int NC = 4; };

typedef const __attribute__((address_space(4))) int *ConstIntsN;

struct Stage { v4u px[kRows]; };

// The tile loads of one item; clipping by the descriptor (it spans the tile's rows), an out-of-range base offset for a thread without
// a chunk.  carry: slot 0 takes the chunk the thread holds in its last slot (the previous band's halo row is this band's first row)
// and only rows 1 .. kRows-1 are loaded, so every source row is read from HBM once.
__device__ __forceinline__ void issue_tile(const BandArgs &a, const Tile &t, const uint8_t *sframe, bool valid, bool carry, int tid, Stage &st)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(sframe + (size_t)t.r0 * a.sstride), 0, valid ? (t.rows_ld - 1) * a.sstride + a.sw * 4 : 0, 0x00020000);
    const int off = tid < t.nchunk ? t.c0 * 4 + tid * 16 : kOOB;
    if (carry) {
        st.px[0] = st.px[kRows - 1];
#pragma unroll
        for (int r = 1; r < kRows; r++) st.px[r] = __builtin_amdgcn_raw_buffer_load_b128(rs, off + r * a.sstride, 0, 0);
    } else {
#pragma unroll
        for (int r = 0; r < kRows; r++) st.px[r] = __builtin_amdgcn_raw_buffer_load_b128(rs, off + r * a.sstride, 0, 0);
    }
}

// staged pixels -> LDS tile as they are (kRows rows are allocated), and the owned pixels premultiplied -> watermark frame.  The last
// tile row is never an owned one (band_rows + 1 <= kRows): neither premultiplied nor stored.
__device__ __forceinline__ void drain_tile(const BandArgs &a, const Tile &t, uint8_t *wframe, int tid, const Stage &st, uint8_t *lds, bool any_glyph)
{
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(a.wm ? wframe + (size_t)t.r0 * a.wm_stride : nullptr), 0,
        a.wm ? (t.own_rows - 1) * a.wm_stride + a.sw * 4 : 0, 0x00020000);
    const bool gl_rows = any_glyph && t.r0 < a.gbox.y1 && t.r1 > a.gbox.y0;   // wave-uniform
    const bool in_tile = tid < t.nchunk;
    const int x = t.c0 + tid * 4;
    const int woff = tid * 4 < t.own_cols ? x * 4 : kOOB;
    const bool in_box = gl_rows && x + 4 > a.gbox.x0 && x < a.gbox.x1;
    const int loff = tid * 16;
    if (in_tile) {
#pragma unroll
        for (int r = 0; r < kRows; r++) *(v4u *)(lds + r * t.pitch + loff) = st.px[r];
    }
    if (!a.wm) return;
#pragma unroll
    for (int r = 0; r < kRows - 1; r++) {
        const v4u p = st.px[r];
        v4u rgba;
#pragma unroll
        for (int i = 0; i < 4; i++) rgba[i] = nrgba_rgba8(p[i]);
        const bool skip = in_box && t.r0 + r >= a.gbox.y0 && t.r0 + r < a.gbox.y1;   // chunks that meet the text box are written by the composite step
        __builtin_amdgcn_raw_buffer_store_b128(rgba, wrs, skip ? kOOB : woff + r * a.wm_stride, 0, 0);
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
    band_out_rows(a, it.b, valid, it.dyA, it.dyB);
}

template <int NX0, bool FP0, int NX1, bool FP1>
__global__ __launch_bounds__(kNT, kNT / 128) void band_nrgba_kernel(NrgbaArgs A)
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

    const bool any_glyph = IPX_FUSED_GLYPHS_RGBA && a.nglyphs > 0 && a.wm;
    const bool can_carry = a.band_rows + 1 == kRows;

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
    const uint8_t *sframe = a.src + (size_t)cur.f * a.src_frame_stride;     // of the item whose loads go out next
    OutBases ob = out_bases(a, cur.f);                                      // of the item being drained / computed
    issue_tile(a, cur.t, sframe, true, false, tid, st);

    for (;;) {
        // A: staged pixels -> LDS tile + premultiplied watermark pixels
        drain_tile(a, cur.t, ob.wm, tid, st, lds, any_glyph);
        __syncthreads();

        // B: the next item's loads
        ItemN nxt;
        const bool has_next = left > 1;
        nxt.b = cur.b; nxt.f = cur.f; nxt.cb = cur.cb;
        if (has_next) {
            if (idx + 1 == idx_end) { idx = idx0 - 1; decode(idx0, nxt); }     // wrap to the start of the run (once per launch)
            else {
                nxt.b = cur.b + 1;
                if (nxt.b == a.nbands) { nxt.b = 0; if (++nxt.f == a.nframes) { nxt.f = 0; ++nxt.cb; } }
            }
        }
        item_setup(a, nxt, has_next);
        if (nxt.f != cur.f) sframe = a.src + (size_t)nxt.f * a.src_frame_stride;
        issue_tile(a, nxt.t, sframe, has_next, can_carry && has_next && nxt.b == cur.b + 1 && nxt.f == cur.f && nxt.cb == cur.cb, tid, st);

        // C: the current item from LDS
        if (any_glyph && tile_meets_textbox(a, cur.t))
            glyph_phase<kNT, kRows - 1, NrgbaConv>(a, cur.t, ob.wm, lds, tid);
        if (a.nscale > 0) {
            scale_out_conv<NX0, FP0, kNT, NrgbaConv>(a, 0, A.mode[0], cur.t, ob.o0, lds, tid, o0, cur.dyA[0], cur.dyB[0]);
            scale_out_conv<NX1, FP1, kNT, NrgbaConv>(a, 1, A.mode[1], cur.t, ob.o1, lds, tid, o1, cur.dyA[1], cur.dyB[1]);
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
    const size_t lds = (size_t)kRows * (size_t)(a.blk_cols + 4) * 4;     // the tile alone (y taps come through scalar loads)
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

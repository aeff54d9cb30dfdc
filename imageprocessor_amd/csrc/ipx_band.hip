// ipx_band.hip -- the fused band kernels: one pass over each source frame produces the watermark
// frame (copy + glyph composite) and every scaled output (resize, thumbnail).
//
// Reference semantics: every operator of image_processor.go:64-65 reads the ORIGINAL decoded
// frame, so resizeImage (resize.go:121-125), cropAndResize (thumbnail.go:114-132) and the
// draw.Draw + DrawString of addTextWatermark (watermark.go:90-92,151) are independent functions of
// one source and can share its single trip through HBM.
//
// Decomposition: a work item is source rows [r0, r1) x columns [c0, c1) of one frame.  For it
//   1. those pixels plus one halo row and one halo column go to LDS with 16-byte coalesced row
//      loads, and the owned pixels go to the watermark frame on the way;
//   2. every destination pixel of each scaled output whose tap pair STARTS inside the owned block
//      (its second row / column is at most the halo) is produced from LDS;
//   3. where the block meets the text box, the glyphs are composited over the block's pixels (from
//      LDS) and those watermark pixels written (step 1 skipped them).
// Which destination rows / columns start in which block, and the taps themselves, are tabulated on
// the host with the reference's float64 arithmetic (ipx_runtime.hip, build_axis_taps).
//
// Two kernels share these steps:
//   band_pipe_kernel  persistent: a workgroup walks a contiguous run of items and keeps the NEXT
//                     item's tile in flight (global -> registers) while it computes the current one
//                     from LDS, so HBM never idles behind the lerp.  Needs 16-byte aligned rows.
//                     This is the throughput path.
//   band_kernel       one workgroup per item, any alignment or width; the fallback.
//
// Bound: HBM.  Per 1080p frame 8.29 MB in, 11.6 MB out; the lerp is ~0.1 flop/B.
//
// Code-shape rule used throughout: NO global load or store sits under a lane-divergent condition.
// hipcc (ROCm 7.2) ends such a branch with s_waitcnt vmcnt(0), which serialises a wave's memory
// operations (measured: a tile copy drops from 5.9 to 3.5 TB/s).  Frames are addressed through
// buffer descriptors instead and an idle lane gets an out-of-range offset (load returns 0, store
// is dropped), so the instruction stream is straight-line and vmcnt can be counted.
#include <algorithm>
#include <cstdlib>

#include "ipx_internal.h"

#ifndef IPX_DIAG
#define IPX_DIAG 0
#endif

#pragma clang fp contract(off)

#include "ipx_device.h"
#include "ipx_band_common.h"

namespace ipx {

namespace {

// ---------------------------------------------------------------------------------------------
// band_kernel: one workgroup per item.  Step 1 here: wave w takes tile rows w, w+4, ...; a "slot"
// is one 1 KiB wave-load (64 lanes x 16 B) of such a row; kLoadU slots are issued back to back
// before any is consumed.  FAST: 16-byte aligned rows and a frame width that is a multiple of 4.
// ---------------------------------------------------------------------------------------------
template <bool FAST>
__device__ __forceinline__ void stage_tile(const BandArgs &a, const Tile &t, const uint8_t *sframe,
                                           uint8_t *wframe, uint8_t *lds, int lane, int wave,
                                           bool any_glyph)
{
    const int frame_bytes = (a.sh - 1) * a.sstride + a.sw * 4;
    const int wm_bytes = wframe ? (a.sh - 1) * a.wm_stride + a.sw * 4 : 0;
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc((void *)sframe, 0, frame_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)wframe, 0, wm_bytes, 0x00020000);
    const int cpl = (t.nchunk + 63) >> 6;                 // slots per row
    const int nrow_w = (t.rows_ld - wave + 3) >> 2;       // rows of this wave (may be <= 0)
    int ri = 0, ci = 0;
    while (ri < nrow_w) {
        v4u v[kLoadU];
        int rr[kLoadU], cc[kLoadU];
#pragma unroll
        for (int u = 0; u < kLoadU; u++) {
            rr[u] = ri; cc[u] = ci;
            const int ry = wave + 4 * ri;
            const int px = (ci * 64 + lane) * 4;
            const bool ok = ry < t.rows_ld && px < t.cols_ld;
            const int off = (t.r0 + ry) * a.sstride + (t.c0 + px) * 4;
            if (FAST) {
                v[u] = __builtin_amdgcn_raw_buffer_load_b128(srs, ok ? off : kOOB, 0, 0);
            } else {
                v[u].x = __builtin_amdgcn_raw_buffer_load_b32(srs, ok ? off : kOOB, 0, 0);
                v[u].y = __builtin_amdgcn_raw_buffer_load_b32(srs, ok && px + 1 < t.cols_ld ? off + 4 : kOOB, 0, 0);
                v[u].z = __builtin_amdgcn_raw_buffer_load_b32(srs, ok && px + 2 < t.cols_ld ? off + 8 : kOOB, 0, 0);
                v[u].w = __builtin_amdgcn_raw_buffer_load_b32(srs, ok && px + 3 < t.cols_ld ? off + 12 : kOOB, 0, 0);
            }
            if (++ci == cpl) { ci = 0; ++ri; }
        }
#pragma unroll
        for (int u = 0; u < kLoadU; u++) {
            const int ry = wave + 4 * rr[u];
            const int px = (cc[u] * 64 + lane) * 4;
            const bool ok = ry < t.rows_ld && px < t.cols_ld;
            const int y = t.r0 + ry, x = t.c0 + px;
            if (ok) *(v4u *)(lds + ry * t.pitch + px * 4) = v[u];
            // owned pixels go to the watermark frame, except chunks that meet the text box: those
            // are written by the composite step
            bool w = ok && wframe && ry < t.own_rows && px < t.own_cols;
            if (any_glyph && chunk_in_textbox(a, x, y)) w = false;
            const int woff = y * a.wm_stride + x * 4;
            if (FAST) {
                __builtin_amdgcn_raw_buffer_store_b128(v[u], wrs, w ? woff : kOOB, 0, 0);
            } else {
                __builtin_amdgcn_raw_buffer_store_b32(v[u].x, wrs, w ? woff : kOOB, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(v[u].y, wrs, w && px + 1 < t.own_cols ? woff + 4 : kOOB, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(v[u].z, wrs, w && px + 2 < t.own_cols ? woff + 8 : kOOB, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(v[u].w, wrs, w && px + 3 < t.own_cols ? woff + 12 : kOOB, 0, 0);
            }
        }
    }
}

template <int NX>
__global__ __launch_bounds__(256) void band_kernel(BandArgs a)
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
    const Tile t = make_tile(a, b, cb);

    // taps first: the oldest loads of the wave, so nothing later waits behind a store
    int dyA[2] = {0, 0}, dyB[2] = {0, 0};
    OutCols<NX, true> o0, o1;
    if (a.nscale > 0) {
        load_xtaps<NX, true>(a, 0, cb, tid, o0);
        load_xtaps<NX, true>(a, 1, cb, tid, o1);
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const ScaleOut &S = a.sc[k];
            dyA[k] = S.row_begin[b];
            dyB[k] = k < a.nscale ? S.row_begin[b + 1] : dyA[k];
        }
    }

    const uint8_t *sframe = a.src + (size_t)f * a.src_frame_stride;
    uint8_t *wframe = a.wm ? a.wm + (size_t)f * a.wm_frame_stride : nullptr;
    const bool any_glyph = IPX_FUSED_GLYPHS_RGBA && a.nglyphs > 0 && wframe;
    const bool fast = (a.sw & 3) == 0 && ((((uintptr_t)sframe) | (uintptr_t)a.sstride) & 15) == 0 &&
                      (!wframe || ((((uintptr_t)wframe) | (uintptr_t)a.wm_stride) & 15) == 0);

    if (fast) stage_tile<true>(a, t, sframe, wframe, lds, tid & 63, tid >> 6, any_glyph);
    else stage_tile<false>(a, t, sframe, wframe, lds, tid & 63, tid >> 6, any_glyph);

    const bool glyph_tile = any_glyph && tile_meets_textbox(a, t);
    if (a.nscale == 0 && !glyph_tile) return;
    __syncthreads();
    // the composite goes first so that its few loads are not queued behind the pixel stores
    if (glyph_tile) glyph_phase<256, 8, TapAsIs>(a, t, wframe, lds, tid);
    if (a.nscale > 0) {
        scale_out<NX, true>(a, 0, t, a.sc[0].out + (size_t)f * a.sc[0].frame_stride, lds, tid, o0, dyA[0], dyB[0]);
        scale_out<NX, true>(a, 1, t, a.sc[1].out + (size_t)f * a.sc[1].frame_stride, lds, tid, o1, dyA[1], dyB[1]);
    }
}

// ---------------------------------------------------------------------------------------------
// band_pipe_kernel: persistent, software-pipelined.  Items are ordered (column block, frame, band)
// and each workgroup walks ONE contiguous run of them, so consecutive items are consecutive bands of
// one frame and the x taps rarely change.  Per item:
//     A  the staged tile (registers, loaded during the previous item's compute) -> LDS, and the
//        owned chunks -> watermark frame;                                   barrier
//     B  issue the NEXT item's tile loads into the same registers (ROWS*CH x 16 B per thread)
//     C  composite + scale the current item from LDS, pixel stores;         barrier
// The tile is laid over the workgroup as ROWS x CH slots per thread: slot (r, h) is tile row r,
// chunk tid + NT*h, so every address is base + r*stride + constant: no per-slot division.
//
// Every source row is read from HBM ONCE: the halo row of a band is the first row of the next band,
// which the same workgroup processes next, and the thread that loaded a chunk of it still holds that
// chunk in its last staging slot -- it moves to slot 0 and only ROWS-1 rows are loaded (8 of 9: 11 % fewer
// read requests, and no reliance on the halo row surviving in an L2 that 64 workgroups stream through).
//
// Clipping is the buffer descriptor's job: the source descriptor of an item starts at the tile's first
// row and ends with its last one, the watermark descriptor covers the owned rows, so row slots beyond
// either fall out of range by themselves (loads return 0, stores are dropped) and the per-slot
// offset is one add; a thread whose chunk lies outside the tile (or, for stores, in the halo column)
// carries an out-of-range base offset.  What depends only on the column block (those base offsets, the
// LDS offsets, the x taps) is recomputed when the column block changes, i.e. almost never.
// Template parameters: NXk = destination columns per thread of output k (NT*NXk per column block),
// FPk = output k may be non-dyadic (float64 lerp, float64 x weights in registers).
// ---------------------------------------------------------------------------------------------
struct Item {
    int f, b, cb;
    int r0, own_rows, rows_ld;
    int dyA[2], dyB[2];
};

typedef const __attribute__((address_space(4))) int *ConstInts;

__device__ __forceinline__ void item_rows(const BandArgs &a, Item &it, bool valid)
{
    it.r0 = it.b * a.band_rows;
    it.own_rows = min(a.band_rows, a.sh - it.r0);
    it.rows_ld = min(it.own_rows + 1, a.sh - it.r0);
    band_out_rows(a, it.b, valid, it.dyA, it.dyB);
}

// what a thread keeps per column block
template <int CH>
struct ColState {
    int c0, c1, cols_ld, nchunk, own_cols;
    int ld_voff[CH];    // byte offset of chunk h from the start of a frame row; kOOB: the chunk is outside the tile
    int st_voff[CH];    // the same for the watermark store; kOOB: not owned (the halo column)
    int loff[CH];       // LDS offset within a tile row
    bool in_tile[CH];
    bool in_box[CH];    // the chunk's columns meet the text box
};

template <int CH, int NT>
__device__ __forceinline__ void col_setup(const BandArgs &a, int cb, int tid, ColState<CH> &cs)
{
    cs.c0 = cb * a.blk_cols;
    cs.c1 = min(cs.c0 + a.blk_cols, a.sw);
    cs.cols_ld = min(cs.c1 + 1, a.sw) - cs.c0;
    cs.nchunk = (cs.cols_ld + 3) >> 2;
    cs.own_cols = cs.c1 - cs.c0;
#pragma unroll
    for (int h = 0; h < CH; h++) {
        const int ch = tid + NT * h, x = cs.c0 + ch * 4;
        cs.in_tile[h] = ch < cs.nchunk;
        cs.loff[h] = ch * 16;
        cs.ld_voff[h] = cs.in_tile[h] ? x * 4 : kOOB;
        cs.st_voff[h] = ch * 4 < cs.own_cols ? x * 4 : kOOB;
        cs.in_box[h] = x + 4 > a.gbox.x0 && x < a.gbox.x1;
    }
}

template <int CH>
__device__ __forceinline__ Tile tile_of(const BandArgs &a, const Item &it, const ColState<CH> &cs)
{
    Tile t;
    t.r0 = it.r0; t.r1 = it.r0 + it.own_rows; t.c0 = cs.c0; t.c1 = cs.c1;
    t.rows_ld = it.rows_ld; t.cols_ld = cs.cols_ld;
    t.pitch = (a.blk_cols + 4) * 4; t.nchunk = cs.nchunk;
    t.own_rows = it.own_rows; t.own_cols = cs.own_cols;
    return t;
}

// B: the tile loads of item `it`.  carry: slot 0 takes the chunk this thread holds in its last slot (the previous band's halo
// row is this band's first row) and only rows 1 .. ROWS-1 are loaded.  valid = false: an empty descriptor, every load out of range.
template <int ROWS, int CH, int NT>
__device__ __forceinline__ void issue_tile(const BandArgs &a, const Item &it, const uint8_t *sframe, const ColState<CH> &cs, bool valid,
                                           bool carry, v4u (&stage)[ROWS * CH])
{
    const int records = valid ? (it.rows_ld - 1) * a.sstride + a.sw * 4 : 0;
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc((void *)(sframe + (size_t)it.r0 * a.sstride), 0, records, 0x00020000);
    if (carry) {
#pragma unroll
        for (int h = 0; h < CH; h++) stage[h] = stage[(ROWS - 1) * CH + h];
#pragma unroll
        for (int r = 1; r < ROWS; r++) {
#pragma unroll
            for (int h = 0; h < CH; h++) stage[r * CH + h] = __builtin_amdgcn_raw_buffer_load_b128(srs, cs.ld_voff[h] + r * a.sstride, 0, 0);
        }
    } else {
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
#pragma unroll
            for (int h = 0; h < CH; h++) stage[r * CH + h] = __builtin_amdgcn_raw_buffer_load_b128(srs, cs.ld_voff[h] + r * a.sstride, 0, 0);
        }
    }
}

// A: registers -> LDS tile (ROWS rows are allocated; rows past the frame's end hold the zeros their loads returned and are never
// read) and the owned chunks -> watermark frame.  Chunks that meet the text box are written by the composite step instead.
template <int ROWS, int CH, int NT>
__device__ __forceinline__ void drain_tile(const BandArgs &a, const Item &it, uint8_t *wframe, const ColState<CH> &cs,
                                           const v4u (&stage)[ROWS * CH], uint8_t *lds, bool any_glyph)
{
    const int pitch = (a.blk_cols + 4) * 4;
#pragma unroll
    for (int h = 0; h < CH; h++) {
        if (cs.in_tile[h]) {
#pragma unroll
            for (int r = 0; r < ROWS; r++) *(v4u *)(lds + r * pitch + cs.loff[h]) = stage[r * CH + h];
        }
    }
    if (!a.wm) return;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(wframe + (size_t)it.r0 * a.wm_stride), 0, (it.own_rows - 1) * a.wm_stride + a.sw * 4, 0x00020000);
    const bool gl_rows = any_glyph && it.r0 < a.gbox.y1 && it.r0 + it.own_rows > a.gbox.y0;   // wave-uniform
    if (!gl_rows) {
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
#pragma unroll
            for (int h = 0; h < CH; h++) __builtin_amdgcn_raw_buffer_store_b128(stage[r * CH + h], wrs, cs.st_voff[h] + r * a.wm_stride, 0, 0);
        }
    } else {
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            const bool row_in = it.r0 + r >= a.gbox.y0 && it.r0 + r < a.gbox.y1;
#pragma unroll
            for (int h = 0; h < CH; h++)
                __builtin_amdgcn_raw_buffer_store_b128(stage[r * CH + h], wrs, row_in && cs.in_box[h] ? kOOB : cs.st_voff[h] + r * a.wm_stride, 0, 0);
        }
    }
}

template <int NX0, bool FP0, int NX1, bool FP1, int ROWS, int CH, int NT>
__global__ __launch_bounds__(NT, NT / 128) void band_pipe_kernel(BandArgs a)   // two workgroups per CU (LDS-bound): 128 VGPRs at 512 threads
{
    extern __shared__ uint4 lds_raw[];
    uint8_t *lds = (uint8_t *)lds_raw;
    const int tid = threadIdx.x;

    const int per_cb = a.nframes * a.nbands;
    const int items = per_cb * a.ncolblk;
    const int G = (int)gridDim.x;
    const int per = (items + G - 1) / G;
    const int idx0 = blockIdx.x * per, idx_end = min(items, idx0 + per);
    if (idx0 >= idx_end) return;
    // pipe_order 1: a workgroup enters its run at an offset of its own and wraps around at the end.  Runs are whole frames more often
    // than not (1024 frames x 135 bands over 512 workgroups = 2 frames each), and with every workgroup at the same band of "its"
    // frame at the same time the chip's 512 streams sit at addresses a multiple of the frame size apart -- how well that spreads
    // over the HBM channels depended on where the allocations happened to land (3.5 or 4.0 ms per launch, per process).
    int idx = idx0 + (a.pipe_order ? (int)((blockIdx.x * 67u) % (unsigned)(idx_end - idx0)) : 0);
    int left = idx_end - idx0;

    const bool any_glyph = IPX_FUSED_GLYPHS_RGBA && a.nglyphs > 0 && a.wm;
    const bool can_carry = a.band_rows + 1 == ROWS;    // the halo row sits in the last staging slot

    Item cur;
    cur.cb = idx / per_cb;
    cur.f = (idx - cur.cb * per_cb) / a.nbands;
    cur.b = idx - cur.cb * per_cb - cur.f * a.nbands;
    item_rows(a, cur, true);
    ColState<CH> cs;
    col_setup<CH, NT>(a, cur.cb, tid, cs);

    OutCols<NX0, FP0> o0;
    OutCols<NX1, FP1> o1;
    if (a.nscale > 0) { load_xtaps<NX0, FP0, NT>(a, 0, cur.cb, tid, o0); load_xtaps<NX1, FP1, NT>(a, 1, cur.cb, tid, o1); }

    v4u stage[ROWS * CH];
#if IPX_DIAG
    const bool do_loads = !(a.dbg & 2);
#else
    const bool do_loads = true;
#endif
    const uint8_t *sframe = a.src + (size_t)cur.f * a.src_frame_stride;     // of the item whose loads go out next
    OutBases ob = out_bases(a, cur.f);                                      // of the item being drained / computed
    issue_tile<ROWS, CH, NT>(a, cur, sframe, cs, do_loads, false, stage);

    // In-kernel phase stamps exist only in the diagnostic build (-DIPX_DIAG=1, tools/build_diag.sh);
    // the shipped kernel executes none.
#if IPX_DIAG
    unsigned long long acc[5] = {0, 0, 0, 0, 0};
#define IPX_STAMP(i) do { if (a.stamps) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc[i] += now_ - tprev; tprev = now_; } } while (0)
    unsigned long long tprev = a.stamps ? __builtin_amdgcn_s_memtime() : 0;
#else
#define IPX_STAMP(i) do { } while (0)
#endif
    for (;;) {
        // A: staged tile -> LDS (+ watermark copy)
        drain_tile<ROWS, CH, NT>(a, cur, ob.wm, cs, stage, lds, any_glyph);
        IPX_STAMP(0);
        __syncthreads();
        IPX_STAMP(1);

        // B: the next item's loads go out now and land while C computes.  Past the end of the run there is no item: the same
        // loads are issued against an empty descriptor, which keeps this block free of branches around memory operations.
        Item nxt;
        const bool has_next = left > 1;
        nxt.b = cur.b; nxt.f = cur.f; nxt.cb = cur.cb;
        if (has_next) {
            if (idx + 1 == idx_end) {                    // wrap to the start of the run (once per launch)
                idx = idx0 - 1;
                nxt.cb = idx0 / per_cb;
                nxt.f = (idx0 - nxt.cb * per_cb) / a.nbands;
                nxt.b = idx0 - nxt.cb * per_cb - nxt.f * a.nbands;
            } else {
                nxt.b = cur.b + 1;
                if (nxt.b == a.nbands) { nxt.b = 0; if (++nxt.f == a.nframes) { nxt.f = 0; ++nxt.cb; } }
            }
        }
        item_rows(a, nxt, has_next);
        if (nxt.f != cur.f) sframe = a.src + (size_t)nxt.f * a.src_frame_stride;
        const Tile t = tile_of<CH>(a, cur, cs);          // before the column state may move on
        if (nxt.cb != cur.cb) {                          // (the staged tile of `cur` is in LDS by now; cs serves the loads below)
            // the current item still needs its own x taps in C: they stay in o0 / o1 until the end of the iteration
            ColState<CH> ncs;
            col_setup<CH, NT>(a, nxt.cb, tid, ncs);
            issue_tile<ROWS, CH, NT>(a, nxt, sframe, ncs, has_next && do_loads, false, stage);
            cs = ncs;
        } else {
            issue_tile<ROWS, CH, NT>(a, nxt, sframe, cs, has_next && do_loads, can_carry && has_next && nxt.b == cur.b + 1 && nxt.f == cur.f, stage);
        }
        IPX_STAMP(2);

        // C: the current item from LDS
        if (any_glyph && tile_meets_textbox(a, t))
            glyph_phase<NT, (ROWS - 1 > 8 ? 8 : ROWS - 1), TapAsIs>(a, t, ob.wm, lds, tid);
#if IPX_DIAG
        if (a.nscale > 0 && !(a.dbg & 1)) {
#else
        if (a.nscale > 0) {
#endif
            scale_out<NX0, FP0, NT>(a, 0, t, ob.o0, lds, tid, o0, cur.dyA[0], cur.dyB[0]);
            scale_out<NX1, FP1, NT>(a, 1, t, ob.o1, lds, tid, o1, cur.dyA[1], cur.dyB[1]);
        }
        IPX_STAMP(3);
        __syncthreads();
        IPX_STAMP(4);

        if (!has_next) break;
        if (nxt.cb != cur.cb && a.nscale > 0) {
            load_xtaps<NX0, FP0, NT>(a, 0, nxt.cb, tid, o0);
            load_xtaps<NX1, FP1, NT>(a, 1, nxt.cb, tid, o1);
        }
        if (nxt.f != cur.f) ob = out_bases(a, nxt.f);
        cur = nxt;
        idx++;
        left--;
    }
#if IPX_DIAG
    if (a.stamps && (tid & 63) == 0 && tid < 256)
        for (int i = 0; i < 5; i++) atomicAdd(&a.stamps[i], acc[i]);
#endif
#undef IPX_STAMP
}

template <int NX>
hipError_t launch_nx(const BandArgs &a, unsigned total, size_t lds, hipStream_t s)
{
    static KernelLaunchCache cache;
    hipError_t e = cache.prepare((const void *)band_kernel<NX>, 256, lds, nullptr);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(band_kernel<NX>, dim3(total), dim3(256), lds, s, a);
    return hipGetLastError();
}

template <int NX0, bool FP0, int NX1, bool FP1, int ROWS, int CH, int NT>
hipError_t launch_pipe(const BandArgs &a, long long items, size_t lds, hipStream_t s)
{
    // a persistent grid must be fully resident: size it from the occupancy the runtime reports for
    // this instantiation and LDS size, not from the LDS arithmetic alone (VGPRs may bind first)
    static KernelLaunchCache cache;
    int resident = 1;
    auto kern = band_pipe_kernel<NX0, FP0, NX1, FP1, ROWS, CH, NT>;
    lds = (size_t)ROWS * (size_t)(a.blk_cols + 4) * 4;     // the tile alone: ROWS rows (y taps come through scalar loads)
    hipError_t e = cache.prepare((const void *)kern, NT, lds, &resident);
    if (e != hipSuccess) return e;
    const long long grid = std::min<long long>(items, (long long)a.cus * std::min(a.pipe_wgs, resident));
    if (getenv("IPX_DEBUG") && cache.first_report()) {
        fprintf(stderr, "[ipx] band_pipe_kernel<%d,%d,%d,%d,%d,%d,%d>: tile %d rows x %d cols, lds %zu B, resident %d/CU, grid %lld, items %lld\n",
                NX0, (int)FP0, NX1, (int)FP1, ROWS, CH, NT, a.band_rows, a.blk_cols, lds, resident, grid, items);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), lds, s, a);
    return hipGetLastError();
}

// column / lerp configurations built for the pipelined kernel, cheapest first
template <int ROWS, int CH, int NT>
hipError_t launch_pipe_cfg(const BandArgs &a, long long items, size_t lds, hipStream_t s, bool *matched)
{
    // a.nx_out counts blocks of 256 destination columns; a thread of an NT-wide workgroup serves NT-strided ones
    constexpr int W = NT / 256, N4 = 4 / W;   // N4 x NT = 1024 columns per block
    const int need0 = a.nscale > 0 ? (a.nx_out[0] + W - 1) / W : 0, need1 = a.nscale > 1 ? (a.nx_out[1] + W - 1) / W : 0;
    const bool fp0 = a.nscale > 0 && a.sc[0].dyadic_shift < 0, fp1 = a.nscale > 1 && a.sc[1].dyadic_shift < 0;
    *matched = true;
    if (need0 <= 1 && need1 <= 1) return launch_pipe<1, true, 1, true, ROWS, CH, NT>(a, items, lds, s);
    if (need0 <= N4 && !fp0 && need1 <= 1) return launch_pipe<N4, false, 1, true, ROWS, CH, NT>(a, items, lds, s);
    if (need0 <= N4 && need1 <= 1) return launch_pipe<N4, true, 1, true, ROWS, CH, NT>(a, items, lds, s);
    if (need0 <= N4 && need1 <= N4 && !fp0 && !fp1) return launch_pipe<N4, false, N4, false, ROWS, CH, NT>(a, items, lds, s);
    *matched = false;
    return hipSuccess;
}

}  // namespace

size_t band_lds_bytes(int band_rows, int blk_cols)
{
    return (size_t)(band_rows + 1) * (size_t)(blk_cols + 4) * 4 + 2 * kYChunk * sizeof(AxisTap);
}

// tile shapes (rows incl. halo, 256-chunk column groups) the pipelined kernel is built for
bool band_pipe_shape(int band_rows, int blk_cols, int *rows, int *ch)
{
    const int r = band_rows + 1, c = blk_cols / 4 + 1;
    if (c <= 256 && r <= 9) { *rows = 9; *ch = 1; return true; }
    if (c <= 256 && r <= 17) { *rows = 17; *ch = 1; return true; }
    if (c <= 512 && r <= 9) { *rows = 9; *ch = 2; return true; }
    return false;
}

hipError_t launch_band(const BandArgs &a, hipStream_t s)
{
    const long long total = (long long)a.nbands * a.ncolblk * a.nframes;
    if (total <= 0) return hipSuccess;
    if (total > 0x7fffffffLL) return hipErrorInvalidValue;
    const size_t lds = band_lds_bytes(a.band_rows, a.blk_cols);
    int rows = 0, ch = 0;
    if (a.pipe_wgs > 0 && band_pipe_shape(a.band_rows, a.blk_cols, &rows, &ch)) {
        bool matched = false;
        hipError_t e = hipSuccess;
        // wide 9-row tiles: 512 threads per workgroup (one 16-byte chunk per thread and row) put twice the
        // waves on a CU for the same LDS, which is what the latency-bound lerp phase wants
        const bool wide = a.pipe_nt == 512;
        if (rows == 9 && ch == 1) e = launch_pipe_cfg<9, 1, 256>(a, total, lds, s, &matched);
        else if (rows == 17 && ch == 1) e = launch_pipe_cfg<17, 1, 256>(a, total, lds, s, &matched);
        else if (wide) e = launch_pipe_cfg<9, 1, 512>(a, total, lds, s, &matched);
        else e = launch_pipe_cfg<9, 2, 256>(a, total, lds, s, &matched);
        if (matched) return e;
    }
    const int nx = std::max(a.nx_out[0], a.nx_out[1]);
    if (nx <= 1) return launch_nx<1>(a, (unsigned)total, lds, s);
    if (nx <= 2) return launch_nx<2>(a, (unsigned)total, lds, s);
    return launch_nx<kBandNX>(a, (unsigned)total, lds, s);
}

}  // namespace ipx

// ipx_ring.hip -- band_ring_kernel: the fused band pass with a dedicated loader wave.
//
// Same work items and arithmetic as ipx_band.hip (see there for the reference mapping); what
// differs is who moves the source tile.  On gfx9 a wave's loads and stores retire through ONE
// in-order counter (vmcnt), so a wave that both prefetches tiles and stores pixels has to drain
// its stores before it can know its prefetch has landed: the memory pipeline of the workgroup
// empties once per item.  Here the roles are split:
//
//   wave 8 (loader)      issues the NEXT item's tile as LDS-DMA (buffer_load_dwordx4 ... lds: HBM ->
//                        LDS with no register stop), waits for exactly those loads, parks the item's
//                        y taps in LDS, and meets the others at one s_barrier per item.  It never
//                        stores to global memory, so its vmcnt counts loads only.
//   waves 0-7 (workers)  after the barrier, copy the owned pixels LDS -> watermark frame, composite
//                        the glyphs, and produce the scaled outputs from LDS.  They never wait on a
//                        global load inside the loop, so their stores are never drained.
//
// Two LDS tile buffers alternate: the loader fills buffer (i+1)&1 while the workers read buffer i&1.
// One workgroup of 576 threads per CU (2 x 68 KiB tiles), persistent over a contiguous run of items.
#include <algorithm>
#include <cstdlib>

#include "ipx_internal.h"

#pragma clang fp contract(off)

#include "ipx_device.h"
#include "ipx_band_common.h"

namespace ipx {

namespace {

constexpr int kWorkers = 512;             // 8 waves
constexpr int kRingThreads = kWorkers + 64;

typedef __attribute__((address_space(3))) void *lds_ptr_t;

struct RingItem {
    int f, b, cb;
    Tile t;
    int nq;            // 16-byte chunks of the tile
    uint32_t magic;    // ceil(2^32 / nchunk)
    int dyA[2], dyB[2];
};

__device__ __forceinline__ void ring_item_setup(const BandArgs &a, RingItem &it)
{
    it.t = make_tile(a, it.b, it.cb);
    it.t.pitch = it.t.nchunk * 16;          // the DMA lays the tile out as a flat run of chunks
    it.nq = it.t.rows_ld * it.t.nchunk;
    it.magic = 0xffffffffu / (uint32_t)it.t.nchunk + 1u;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        it.dyA[k] = a.nscale > 0 ? a.sc[k].row_begin[it.b] : 0;
        it.dyB[k] = k < a.nscale ? a.sc[k].row_begin[it.b + 1] : it.dyA[k];
    }
}

__device__ __forceinline__ void ring_advance(const BandArgs &a, RingItem &it)
{
    if (++it.b == a.nbands) { it.b = 0; if (++it.f == a.nframes) { it.f = 0; ++it.cb; } }
}

// loader: the whole tile as 1 KiB LDS-DMA pieces, then the item's y taps into registers
__device__ __forceinline__ void loader_issue(const BandArgs &a, const RingItem &it, int lane, uint8_t *buf,
                                             v4u (&ty_stage)[2][2])
{
    const int frame_bytes = (a.sh - 1) * a.sstride + a.sw * 4;
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(a.src + (size_t)it.f * a.src_frame_stride), 0, frame_bytes, 0x00020000);
    const int base = it.t.r0 * a.sstride + it.t.c0 * 4;
    const int pieces = (it.nq + 63) >> 6;
    for (int k = 0; k < pieces; k++) {
        const int q = k * 64 + lane;
        const int row = (int)__umulhi((uint32_t)q, it.magic);
        const int ch = q - row * it.t.nchunk;
        // lanes past the end of the tile get an out-of-range offset; what they deposit lands in the
        // padding behind the tile (the buffer is a whole number of 1 KiB pieces)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(srs, (lds_ptr_t)(buf + k * 1024), 16,
                                                 q < it.nq ? base + row * a.sstride + ch * 16 : 0x7fffffff, 0, 0, 0);
    }
    if (a.nscale > 0) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const v4u *yp = (const v4u *)&a.sc[k].yt[min(it.dyA[k] + lane, a.sc[k].dh - 1)];
            ty_stage[k][0] = yp[0]; ty_stage[k][1] = yp[1];
        }
    }
}

template <int NX>
__global__ __launch_bounds__(kRingThreads) void band_ring_kernel(BandArgs a, int tile_bytes)
{
    extern __shared__ uint4 lds_raw[];
    uint8_t *lds = (uint8_t *)lds_raw;
    uint8_t *tile[2] = {lds, lds + tile_bytes};
    AxisTap *ytap[2] = {(AxisTap *)(lds + 2 * tile_bytes), (AxisTap *)(lds + 2 * tile_bytes) + 2 * kYChunk};

    const int tid = threadIdx.x;
    const int per_cb = a.nframes * a.nbands;
    const int items = per_cb * a.ncolblk;
    const int per = (items + (int)gridDim.x - 1) / (int)gridDim.x;
    const int idx0 = blockIdx.x * per;
    const int n = min(items, idx0 + per) - idx0;
    if (n <= 0) return;

    RingItem it;
    it.cb = idx0 / per_cb;
    it.f = (idx0 - it.cb * per_cb) / a.nbands;
    it.b = idx0 - it.cb * per_cb - it.f * a.nbands;
    ring_item_setup(a, it);

    if (tid >= kWorkers) {
        // ---------------------------------------------------------------- loader wave
        const int lane = tid - kWorkers;
        v4u ty_stage[2][2];
        loader_issue(a, it, lane, tile[0], ty_stage);
        for (int i = 0; i < n; i++) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // item i has landed (loads only)
            if (a.nscale > 0) {
#pragma unroll
                for (int k = 0; k < 2; k++)
                    if (lane < min(it.dyB[k] - it.dyA[k], kYChunk)) {
                        v4u *yl = (v4u *)&ytap[i & 1][k * kYChunk + lane];
                        yl[0] = ty_stage[k][0]; yl[1] = ty_stage[k][1];
                    }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                            // workers are done with item i-1
            if (i + 1 < n) {
                ring_advance(a, it);
                ring_item_setup(a, it);
                loader_issue(a, it, lane, tile[(i + 1) & 1], ty_stage);
            }
        }
        return;
    }

    // -------------------------------------------------------------------- worker waves
    const bool any_glyph = a.nglyphs > 0 && a.wm;
    XTap tx[2][NX];
    int dxA[2] = {0, 0}, dxB[2] = {0, 0};
    int cb_loaded = it.cb;
    if (a.nscale > 0) load_xtaps<NX, kWorkers>(a, it.cb, tid, tx, dxA, dxB);

    for (int i = 0; i < n; i++) {
        if (i) { ring_advance(a, it); ring_item_setup(a, it); }
        if (it.cb != cb_loaded && a.nscale > 0) { load_xtaps<NX, kWorkers>(a, it.cb, tid, tx, dxA, dxB); cb_loaded = it.cb; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // my reads of the other buffer are done
        __builtin_amdgcn_s_barrier();                                 // item i is in tile[i & 1]
        const uint8_t *buf = tile[i & 1];

        // owned pixels LDS -> watermark frame (chunks that meet the text box are left to the composite)
        if (a.wm) {
            uint8_t *wframe = a.wm + (size_t)it.f * a.wm_frame_stride;
            const int wm_bytes = (a.sh - 1) * a.wm_stride + a.sw * 4;
            const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)wframe, 0, wm_bytes, 0x00020000);
            const int wbase = it.t.r0 * a.wm_stride + it.t.c0 * 4;
            for (int q0 = 0; q0 < it.nq; q0 += kWorkers * 4) {
                v4u v[4];
                int off[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int q = q0 + u * kWorkers + tid;
                    const int qc = min(q, it.nq - 1);
                    const int row = (int)__umulhi((uint32_t)qc, it.magic);
                    const int ch = qc - row * it.t.nchunk;
                    v[u] = *(const v4u *)(buf + qc * 16);
                    bool w = q < it.nq && row < it.t.own_rows && ch * 4 < it.t.own_cols;
                    if (any_glyph && chunk_in_textbox(a, it.t.c0 + ch * 4, it.t.r0 + row)) w = false;
                    off[u] = w ? wbase + row * a.wm_stride + ch * 16 : 0x7fffffff;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) __builtin_amdgcn_raw_buffer_store_b128(v[u], wrs, off[u], 0, 0);
            }
            if (any_glyph && tile_meets_textbox(a, it.t)) glyph_phase<kWorkers>(a, it.t, wframe, buf, tid);
        }
        if (a.nscale > 0)
            scale_phase<NX, kWorkers, false>(a, it.t, it.f, buf, ytap[i & 1], tid, tx, dxA, dxB, it.dyA, it.dyB);
    }
}

template <int NX>
hipError_t launch_ring_nx(const BandArgs &a, long long items, int tile_bytes, size_t lds, hipStream_t s)
{
    static thread_local size_t lds_set = 0;
    if (lds != lds_set) {
        hipError_t e = hipFuncSetAttribute((const void *)band_ring_kernel<NX>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_set = lds;
    }
    const long long grid = std::min<long long>(items, a.cus);
    static thread_local bool said = false;
    if (!said && getenv("IPX_DEBUG")) {
        said = true;
        fprintf(stderr, "[ipx] band_ring_kernel<%d>: tile %d rows x %d cols, lds %zu B, grid %lld, items %lld\n", NX,
                a.band_rows, a.blk_cols, lds, grid, items);
    }
    hipLaunchKernelGGL(band_ring_kernel<NX>, dim3((unsigned)grid), dim3(kRingThreads), lds, s, a, tile_bytes);
    return hipGetLastError();
}

}  // namespace

// whole 1 KiB DMA pieces per tile buffer
static int ring_tile_bytes(int band_rows, int blk_cols)
{
    const int bytes = (band_rows + 1) * (blk_cols / 4 + 1) * 16;
    return (bytes + 1023) & ~1023;
}

size_t ring_lds_bytes(int band_rows, int blk_cols)
{
    return 2 * (size_t)ring_tile_bytes(band_rows, blk_cols) + 2 * 2 * kYChunk * sizeof(AxisTap);
}

hipError_t launch_ring(const BandArgs &a, hipStream_t s)
{
    const long long total = (long long)a.nbands * a.ncolblk * a.nframes;
    if (total <= 0) return hipSuccess;
    const int tb = ring_tile_bytes(a.band_rows, a.blk_cols);
    const size_t lds = ring_lds_bytes(a.band_rows, a.blk_cols);
    if (a.nx <= 2) return launch_ring_nx<1>(a, total, tb, lds, s);   // a.nx counts 256 columns, the workers are 512
    return launch_ring_nx<2>(a, total, tb, lds, s);
}

}  // namespace ipx

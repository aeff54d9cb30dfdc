// ipx_ks_fused.hip -- the one-pass kernel of the batched path: resize + thumbnail + watermark copy from ONE read of each source frame.
//
// What it replaces: resizeImage (resize.go:121-125) and cropAndResize (thumbnail.go:114-132), i.e. xdraw.BiLinear.Scale = x/image/draw's
// kernelScaler with the tent kernel, plus the full-frame draw.Draw of addTextWatermark (watermark.go:90-92); the text itself goes on in
// composite_kernel afterwards.  Every operator reads the ORIGINAL frame (image_processor.go:64-65), so they share the read.
//
// Structure.  Item = (frame, strip of source columns, segment of source rows); one workgroup per item.
//   * The workgroup streams its rows top to bottom in groups of B.  A group's pixels are loaded 16 bytes per lane one group ahead into
//     registers, go to the LDS tile and -- the rows the segment owns -- to the watermark frame.
//   * Waves have roles: wave w serves one scaled output, and each of its lanes owns `cpl` destination COLUMNS of it for the whole item.
//     Per group and column the lane runs scaleX on the B rows in LDS (taps from LDS, weights from an LDS table read once per group and
//     tap: sum += float64(tap) * weight in source order, then * invTotalWeightFFFF) and feeds each result straight into scaleY's running
//     sums: a destination row dy accumulates in register set dy % NACC while the source rows of its range stream by, in source-row
//     order, exactly as scaleY_RGBA_Src walks a column of tmp.  The row table (LDS, staged with the pixels) says per source row what each
//     accumulator gets and whether a destination row is complete -- then it is finished (clamp to alpha, * invTotalWeight, ftou, >> 8),
//     stored, and the accumulator cleared.  The float64 image tmp of the reference never exists, and no sum is ever split or reordered.
//   * Bound: FP64 VALU (v_cvt_f64_u32 / v_mul_f64 / v_add_f64 at 16 lanes per clock and SIMD): 1080p -> 1024x768 + 200x200 costs about
//     4 M horizontal tap-pixels per output kind and frame, 12 (opaque) or 16 float64-rate operations each.  HBM traffic stays the
//     algorithmic minimum (source once, outputs once).
//
// NCH = 3: the frame is assumed opaque (alpha 0xff everywhere, as every decoded JPEG and every RGB PNG is).  Then scaleX's alpha sums
// are data-independent, no colour can exceed alpha (rounding is monotonic: each product and each partial sum of a colour is <= the one of
// alpha, term by term), so the clamp never fires and the stored alpha is 0xff: the alpha channel costs nothing.  The kernel checks
// the assumption on every pixel it stages; an item that meets alpha != 0xff gives up and is redone by the NCH = 4 kernel (`redo`).
#include <algorithm>
#include <cstdlib>

#include "ipx_ks.h"

#pragma clang fp contract(off)

#include "ipx_device.h"

namespace ipx {
namespace {

#ifndef IPX_DIAG
#define IPX_DIAG 0
#endif
#if IPX_DIAG
#define KS_DIAG_ARGS , tsum, tlast
#define KS_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tsum[i] += t_ - tlast; tlast = t_; } while (0)
#else
#define KS_DIAG_ARGS
#define KS_STAMP(i) do { } while (0)
#endif
constexpr int kOOB = 0x7fffffff;

// Workgroup barrier for LDS traffic only.  __syncthreads() also waits for every global store the wave has in flight (vmcnt(0)): here
// that was the round trip of the watermark and destination-row stores at every group, a fifth of the run.  Nothing in this kernel
// communicates between waves through global memory, so LDS visibility (lgkmcnt) is all the barrier has to order.
__device__ __forceinline__ void ks_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t ks_ftou8(double f)   // uint8(ftou(f) >> 8)
{
    int i = (int)(0xffff * f + 0.5);
    i = min(max(i, 0), 0xffff);
    return (uint32_t)i >> 8;
}

// channel C of the tap as the 16-bit value scaleX_RGBA weights, as float64.  CROP: the crop copy's `if pr > pa { pr = pa }` first.
template <int C, bool CROP>
__device__ __forceinline__ double ks_chan(uint32_t px)
{
    if (!CROP || C == 3) return widen<C>(px);
    const uint32_t c = (px >> (8 * C)) & 0xffu, al = px >> 24;
    return (double)(min(c, al) * 0x101u);
}

template <int NCH, int NACC>
struct KsCol {                 // one destination column of a lane
    double q[NACC][NCH];       // scaleY's running sums
    double itwf;               // invTotalWeightFFFF of the column
    int xb;                    // LDS byte offset of the column's first tap within a tile row
    int wofs;                  // LDS byte offset of the column's first weight
    int ooff;                  // byte offset of the column in a destination row; kOOB = the lane has no such column
};

// scaleX on the B rows of the tile for one column, then scaleY's accumulation and, where a destination row completes, its store
template <int NCH, int NACC, int B, bool CROP>
__device__ __forceinline__ void ks_column(const uint8_t *lds, const uint8_t *tile, KsCol<NCH, NACC> &c, int ntap, int wstride, int pitch,
                                          const uint8_t *rows, __amdgpu_buffer_rsrc_t ors, int ostride
#if IPX_DIAG
                                          , unsigned long long *tsum, unsigned long long &tlast
#endif
                                          )
{
    double acc[B][NCH];
#pragma unroll
    for (int r = 0; r < B; r++)
#pragma unroll
        for (int k = 0; k < NCH; k++) acc[r][k] = 0.0;
    const uint8_t *tap = tile + c.xb;
    const uint8_t *wp = lds + c.wofs;
    // two taps per iteration: their ten LDS reads are in flight together
    auto one_tap = [&](uint32_t px, double w, int r) {
        acc[r][0] += ks_chan<0, CROP>(px) * w;
        if (NCH > 1) acc[r][1 % NCH] += ks_chan<1, CROP>(px) * w;
        if (NCH > 2) acc[r][2 % NCH] += ks_chan<2, CROP>(px) * w;
        if (NCH > 3) acc[r][3 % NCH] += ks_chan<3, CROP>(px) * w;
    };
    int t = 0;
    for (; t + 2 <= ntap; t += 2) {
        const double w0 = *(const double *)wp, w1 = *(const double *)(wp + wstride);
        uint32_t p0[B], p1[B];
#pragma unroll
        for (int r = 0; r < B; r++) { p0[r] = *(const uint32_t *)(tap + r * pitch); p1[r] = *(const uint32_t *)(tap + r * pitch + 4); }
#pragma unroll
        for (int r = 0; r < B; r++) one_tap(p0[r], w0, r);
#pragma unroll
        for (int r = 0; r < B; r++) one_tap(p1[r], w1, r);     // (a column's taps in source order, row by row)
        tap += 8;
        wp += 2 * wstride;
    }
    if (t < ntap) {
        const double w0 = *(const double *)wp;
        uint32_t p0[B];
#pragma unroll
        for (int r = 0; r < B; r++) p0[r] = *(const uint32_t *)(tap + r * pitch);
#pragma unroll
        for (int r = 0; r < B; r++) one_tap(p0[r], w0, r);
    }
    KS_STAMP(4);                                         // scaleX of the group's rows for this column
    typedef KsRowT<NACC> Row;
    // the group's row entries in one batch of LDS reads (read one by one where they are used, every read was a round trip of its own)
    double rw[B][NACC], ritw[B][NACC];
    int remit[B][NACC];
#pragma unroll
    for (int r = 0; r < B; r++) {
        const Row *row = (const Row *)(rows + r * sizeof(Row));
#pragma unroll
        for (int p = 0; p < NACC; p++) { rw[r][p] = row->w[p]; ritw[r][p] = row->itw[p]; remit[r][p] = row->emit[p]; }
    }
#pragma unroll
    for (int r = 0; r < B; r++) {
        double tmp[NCH];
#pragma unroll
        for (int k = 0; k < NCH; k++) tmp[k] = acc[r][k] * c.itwf;
#pragma unroll
        for (int p = 0; p < NACC; p++) {
            const double w = rw[r][p];
#pragma unroll
            for (int k = 0; k < NCH; k++) c.q[p][k] += tmp[k] * w;     // w = 0 for an accumulator this row does not feed: x + 0 == x
            const int dy = __builtin_amdgcn_readfirstlane(remit[r][p]);
            if (dy >= 0) {                                             // wave-uniform
                const double s = ritw[r][p];
                uint32_t px;
                if (NCH == 4) {
                    double pa = c.q[p][3 % NCH];
                    double pr = c.q[p][0], pg = c.q[p][1 % NCH], pb = c.q[p][2 % NCH];
                    if (pr > pa) pr = pa;
                    if (pg > pa) pg = pa;
                    if (pb > pa) pb = pa;
                    px = ks_ftou8(pr * s) | ks_ftou8(pg * s) << 8 | ks_ftou8(pb * s) << 16 | ks_ftou8(pa * s) << 24;
                } else {
                    px = ks_ftou8(c.q[p][0] * s) | ks_ftou8(c.q[p][1 % NCH] * s) << 8 | ks_ftou8(c.q[p][2 % NCH] * s) << 16 | 0xff000000u;
                }
                __builtin_amdgcn_raw_buffer_store_b32(px, ors, c.ooff, dy * ostride, 0);
#pragma unroll
                for (int k = 0; k < NCH; k++) c.q[p][k] = 0.0;
            }
        }
    }
}

template <int NCH, int NACC, int B, bool OPQ>
__global__ __launch_bounds__(kKsMaxThreads) void ks_fused_kernel(KsFusedArgs a)
{
    extern __shared__ __align__(16) uint8_t lds[];
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int item = (int)blockIdx.x;
    if (!OPQ && a.redo && !a.redo[item]) return;
    const int seg = item % a.nseg;
    item /= a.nseg;
    const int strip = item % a.nstrips, frame = item / a.nstrips;
    const KsStrip st = a.strips[strip];
    const KsSeg sg = a.segs[seg];
    const int pitch = a.pitch, CH = pitch >> 4;
    typedef KsRowT<NACC> Row;
    constexpr int RW = (int)(sizeof(Row) / 4);       // dwords per row entry

    const uint8_t *sframe = a.src + (size_t)frame * a.src_fs;
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc((void *)sframe, 0, (a.sh - 1) * a.sstride + a.sw * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)(a.wm ? a.wm + (size_t)frame * a.wm_fs : nullptr), 0,
                                                                        a.wm ? (a.sh - 1) * a.wm_stride + a.sw * 4 : 0, 0x00020000);

    // ---- what this thread stages per group: up to kKsMaxStage 16-byte chunks of the tile, and one dword of the row entries ----
    int s_lds[kKsMaxStage], s_src[kKsMaxStage], s_wm[kKsMaxStage], s_row[kKsMaxStage];   // (the frame width is a multiple of 4: whole chunks)
#pragma unroll
    for (int i = 0; i < kKsMaxStage; i++) {
        const int q = tid + i * a.nthreads;
        const int row = q / CH, ch = q - row * CH, x = st.t0 + ch * 4;
        const bool in = q < B * CH && x < a.sw;
        s_row[i] = in ? row : -1;
        s_lds[i] = row * pitch + ch * 16;
        s_src[i] = in ? row * a.sstride + x * 4 : kOOB;
        s_wm[i] = in && x >= st.c0 && x < st.c1 ? row * a.wm_stride + x * 4 : kOOB; // owned columns only (c0 and c1 are multiples of 4)
    }
    const int rk = tid / (B * RW), ri = tid - rk * (B * RW);                         // row entries: dword ri of output rk's B entries
    const bool rstage = rk < a.nout;
    const uint32_t *rsrc = rstage ? (const uint32_t *)a.o[rk].rows + (size_t)a.o[rk].rowoff[seg] * RW + ri : nullptr;

    // ---- this lane's destination columns ----
    int role = -1, wk = 0;
    {
        const int rb = a.wave_role[wv & 15];
        if (rb != 0xff) { role = rb >> 4; wk = rb & 15; }
    }
    role = __builtin_amdgcn_readfirstlane(role);
    wk = __builtin_amdgcn_readfirstlane(wk);
    // the waves of the output with the longer tap loop are the critical path of every group: they go first when several waves of a
    // SIMD are ready, the others fill the gaps
    if (a.nout > 1 && role >= 0 && a.o[role].ntap * a.o[role].cpl > a.o[1 - role].ntap * a.o[1 - role].cpl) __builtin_amdgcn_s_setprio(2);
    KsCol<NCH, NACC> col[kKsMaxCpl];
    __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void *)nullptr, 0, 0, 0x00020000);
    int ntap = 0, wstride = 0, ostride = 0, cpl = 0;
    bool crop = false;
    const uint8_t *rows_lds = lds + a.lds_rows;
    const int tile_bytes = B * pitch, rows_bytes = 2 * B * (int)sizeof(Row);   // the tile and the row entries are double-buffered
    if (role >= 0) {
        const KsFusedOut &o = a.o[role];
        ors = __builtin_amdgcn_make_buffer_rsrc((void *)(o.out + (size_t)frame * o.frame_stride), 0, o.obytes, 0x00020000);
        ntap = o.ntap; wstride = o.wcols * 8; ostride = o.ostride; cpl = o.cpl;
        crop = o.kind == IPX_SRC_RGBA_CROP;
        rows_lds += role * B * (int)sizeof(Row);
        const int cb = o.colb[strip], ce = o.colb[strip + 1];
        // the strip's weight table -> LDS (every wave of the role copies a share)
        {
            const double *wsrc = o.wx + (size_t)strip * o.ntap * o.wcols;
            double *wdst = (double *)(lds + a.lds_w[role]);
            const int n = o.ntap * o.wcols;
            for (int i = wk * 64 + lane; i < n; i += o.waves * 64) wdst[i] = wsrc[i];
        }
#pragma unroll
        for (int j = 0; j < kKsMaxCpl; j++) {
            const int slot = wk * 64 + lane + j * 64 * o.waves, dx = cb + slot;
            const bool has = j < o.cpl && dx < ce;
#pragma unroll
            for (int p = 0; p < NACC; p++)
#pragma unroll
                for (int k = 0; k < NCH; k++) col[j].q[p][k] = 0.0;
            col[j].itwf = has ? o.itwf[dx] : 0.0;
            col[j].xb = has ? (o.sr_x0 + o.xlo[dx] - st.t0) * 4 : 0;
            col[j].wofs = a.lds_w[role] + (has ? slot : 0) * 8;
            col[j].ooff = has ? dx * 4 : kOOB;
        }
    }

    const int ngroups = (sg.r1 - sg.ys + B - 1) / B;
    u32x4 stage[kKsMaxStage];
    uint32_t rstg = 0;
    auto issue = [&](int g) {
        const int y0 = sg.ys + g * B;
#pragma unroll
        for (int i = 0; i < kKsMaxStage; i++) {
            const bool ok = s_row[i] >= 0 && y0 + s_row[i] < sg.r1;
            stage[i] = __builtin_amdgcn_raw_buffer_load_b128(srs, ok ? s_src[i] : kOOB, y0 * a.sstride, 0);
        }
        if (rstage) rstg = rsrc[(size_t)g * B * RW];
    };
    issue(0);
    bool bad = false;
#if IPX_DIAG
    unsigned long long tsum[6] = {0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif
    for (int g = 0; g < ngroups; g++) {
#if IPX_DIAG
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        KS_STAMP(0);                                     // waiting for the group's staged loads
#endif
        const int y0 = sg.ys + g * B;
        // ---- registers -> LDS tile and watermark frame ----
#pragma unroll
        for (int i = 0; i < kKsMaxStage; i++) {
            if (s_row[i] < 0) continue;
            const u32x4 v = stage[i];
            *(u32x4 *)(lds + (g & 1) * tile_bytes + s_lds[i]) = v;
            const int y = y0 + s_row[i];
            const bool live = y < sg.r1;
            if (OPQ && live) bad |= ((v.x & v.y & v.z & v.w) >> 24) != 0xffu;
            if (a.wm && live && y >= sg.r0) __builtin_amdgcn_raw_buffer_store_b128(v, wrs, s_wm[i], y0 * a.wm_stride, 0);
        }
        if (rstage) ((uint32_t *)(lds + a.lds_rows + (g & 1) * rows_bytes))[rk * B * RW + ri] = rstg;
        KS_STAMP(1);                                     // registers -> LDS, watermark stores
        if (OPQ && g == 0) {
            if (__syncthreads_or(bad)) {                 // not an opaque frame (seen in the first rows already): the general kernel redoes the item
                if (tid == 0) a.redo[blockIdx.x] = 1;
                return;
            }
        } else ks_barrier();
        KS_STAMP(2);                                     // barrier
        if (g + 1 < ngroups) issue(g + 1);
        KS_STAMP(3);                                     // issuing the next group's loads
        // ---- scaleX on the tile, scaleY's sums, finished destination rows ----
        // (one barrier per group: a wave that writes buffer g & 1 two groups on has passed the barrier of group g + 1, which every wave
        // reaches only after its arithmetic on group g)
        if (role >= 0) {
            const uint8_t *tile = lds + (g & 1) * tile_bytes, *rws = rows_lds + (g & 1) * rows_bytes;
            if (crop) {
#pragma unroll
                for (int j = 0; j < kKsMaxCpl; j++)
                    if (j < cpl) ks_column<NCH, NACC, B, true>(lds, tile, col[j], ntap, wstride, pitch, rws, ors, ostride KS_DIAG_ARGS);
            } else {
#pragma unroll
                for (int j = 0; j < kKsMaxCpl; j++)
                    if (j < cpl) ks_column<NCH, NACC, B, false>(lds, tile, col[j], ntap, wstride, pitch, rws, ors, ostride KS_DIAG_ARGS);
            }
        }
        KS_STAMP(5);                                     // scaleY's sums and finished rows
    }
#if IPX_DIAG
    if (a.stamps && lane == 0)
        for (int i = 0; i < 6; i++) atomicAdd(&a.stamps[(role < 0 ? 2 : role) * 8 + i], tsum[i]);
#endif
    if (OPQ) {                                           // any pixel with alpha != 0xff in the rest of the item: redo it as well
        const int any = __syncthreads_or(bad);
        if (tid == 0) a.redo[blockIdx.x] = any;
    }
}

template <int NCH, int NACC, bool OPQ>
hipError_t launch_one(const KsFusedPlan &p, const KsFusedArgs &a, int nitems, hipStream_t s)
{
    static KernelLaunchCache cache;
    auto fn = ks_fused_kernel<NCH, NACC, kKsRows, OPQ>;
    hipError_t e = cache.prepare((const void *)fn, p.nthreads, (size_t)p.lds_bytes, nullptr);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fn, dim3(nitems), dim3(p.nthreads), (size_t)p.lds_bytes, s, a);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_ks_fused(const KsFusedPlan &p, KsFusedArgs &a, int cus, hipStream_t s, bool *matched)
{
    *matched = false;
    if (!p.ok || a.nframes <= 0) return hipSuccess;
    if (a.sw & 3) return hipSuccess;                               // whole 16-byte chunks only
    // frames are addressed dword-wise through buffer descriptors
    if ((((uintptr_t)a.src) | (uintptr_t)a.sstride | a.src_fs) & 3) return hipSuccess;
    if (a.wm && ((((uintptr_t)a.wm) | (uintptr_t)a.wm_stride | a.wm_fs) & 3)) return hipSuccess;
    for (int k = 0; k < a.nout; k++)
        if (a.o[k].kind != IPX_SRC_RGBA && a.o[k].kind != IPX_SRC_RGBA_CROP) return hipSuccess;
    // large batches: one segment per frame (no row is staged twice); small ones: enough items to fill the chip
    const char *ev = getenv("IPX_KS_SPLIT");                       // test knob: 1 = always the split segmentation, 0 = never
    const bool whole = ev && *ev ? atoi(ev) == 0 : (long long)a.nframes * p.nstrips >= 2LL * cus;
    const KsFusedGeom &g = whole || p.split.nseg <= 1 ? p.whole : p.split;
    a.nstrips = p.nstrips; a.nseg = g.nseg; a.nthreads = p.nthreads; a.pitch = p.pitch;
    a.strips = p.strips; a.segs = g.segs;
    a.lds_rows = p.lds_rows;
    for (int i = 0; i < a.nout; i++) {                             // a.o[i].pk: which of the plan's outputs this is
        const int k = a.o[i].pk;
        const KsFusedPlan::Out &po = p.o[k];
        KsFusedOut &o = a.o[i];
        o.ntap = po.ntap; o.waves = po.waves; o.cpl = po.cpl; o.wcols = po.wcols;
        o.wx = po.wx; o.itwf = po.itwf; o.xlo = po.xlo; o.colb = po.colb;
        o.rows = g.rows[k]; o.rowoff = g.rowoff[k];
        a.lds_w[i] = p.lds_w[k];
    }
    // deal the roles out: the waves of the output with fewer waves are spread evenly among the others
    {
        for (int w = 0; w < 16; w++) a.wave_role[w] = 0xff;
        const int n0 = a.nout > 0 ? a.o[0].waves : 0, n1 = a.nout > 1 ? a.o[1].waves : 0, n = n0 + n1;
        int i0 = 0, i1 = 0;
        for (int w = 0; w < n && w < 16; w++) {
            // output 1 takes slot w when its share of the slots so far falls behind
            const bool one = i1 < n1 && (i0 >= n0 || (long long)(i1 + 1) * n <= (long long)(w + 1) * n1);
            if (one) a.wave_role[w] = (uint8_t)(1 << 4 | i1++);
            else a.wave_role[w] = (uint8_t)(0 << 4 | i0++);
        }
    }
    const long long nitems = (long long)a.nframes * p.nstrips * g.nseg;
    if (nitems > 0x7fffffffLL) return hipSuccess;
    *matched = true;
    hipError_t e;
    const bool spec = a.redo != nullptr;        // the caller provides the redo flags when it wants the speculative opaque pass first
    if (p.nacc == 2) {
        if (spec) {
            e = launch_one<3, 2, true>(p, a, (int)nitems, s);
            if (e == hipSuccess) e = launch_one<4, 2, false>(p, a, (int)nitems, s);
        } else e = launch_one<4, 2, false>(p, a, (int)nitems, s);
    } else {
        if (spec) {
            e = launch_one<3, 4, true>(p, a, (int)nitems, s);
            if (e == hipSuccess) e = launch_one<4, 4, false>(p, a, (int)nitems, s);
        } else e = launch_one<4, 4, false>(p, a, (int)nitems, s);
    }
    return e;
}

}  // namespace ipx

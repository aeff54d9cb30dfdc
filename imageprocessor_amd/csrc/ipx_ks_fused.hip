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
//
// The float pass (FAST = true; RGBA frames taken as opaque, YCbCr, Gray, NRGBA, and the 16-bit / CMYK types whose colours stay under their alpha).  The float64 arithmetic above is what bounds the kernel, and a
// byte of output needs almost none of it: the same sums in float -- taps are exact in float, one fused multiply-add per term, both
// normalisations folded into the weights -- land within (nx + ny + 3) 2^-24 of the reference's value, relatively (derivation at
// ks_float_eps, ipx_ks_host.cpp), and the byte is floor((V + 0.5) / 256): unless V + 0.5 lies that close to a multiple of 256, the
// float result IS the reference's byte.  The pass checks exactly that for every channel it stores (the fraction of t / 256 against
// feps times t / 256); a pixel with a channel it cannot decide -- about one in a thousand -- goes on its frame's list, and ks_fix_kernel recomputes
// the listed pixels in float64, operation by operation as the reference does.  A frame whose list is full hands the item to the
// float64 kernel through `redo`, like an item that is not opaque.  Results are the reference's bits either way; IPX_KS_FAST=0 runs
// float64 throughout.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

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

// ---- source types: how four pixels of a tile row come from HBM, go to LDS (as the 16-bit values scaleX weights) and to the watermark frame ----
enum { KS_RGBA = 0, KS_NRGBA = 1, KS_YCC = 2, KS_GRAY = 3, KS_TAP64 = 4 };
// bytes per pixel of the LDS tile: RGBA keeps the packed bytes (a tap is byte * 0x101), the converting types keep four 16-bit taps,
// Gray one (y * 0x101)
template <int SRC> struct KsPx { static constexpr int bytes = SRC == KS_RGBA ? 4 : SRC == KS_GRAY ? 2 : 8; };
// tap modes of an output (KsFusedOut::kind folded per source type)
enum { KS_TAP_PLAIN = 0,   // the value in the tile
       KS_TAP_CLAMP = 1,   // RGBA tile: min(c, a) -- the crop copy's clamp of a colour above its alpha
       KS_TAP_TOP = 2,     // 16-bit tile: (v >> 8) * 0x101 -- the crop copy's 8-bit pixel widened again
       KS_TAP_MINTOP = 3 };// 16-bit tile: (min(c, a) >> 8) * 0x101 (RGBA64 may hold a colour above its alpha)

struct KsStageRegs { uint32_t v[4]; };

// color.YCbCr.RGBA as scaleX_YCbCr4xx inlines it, for one pixel with the chroma terms of its sample: 16-bit values, clamped
__device__ __forceinline__ void ks_ycc16(uint32_t yb, int rt, int gt, int bt, uint32_t &r, uint32_t &g, uint32_t &b)
{
    const int yy1 = (int)(yb * 0x10101u);
    r = (uint32_t)min(max((yy1 + rt) >> 8, 0), 0xffff);
    g = (uint32_t)min(max((yy1 + gt) >> 8, 0), 0xffff);
    b = (uint32_t)min(max((yy1 + bt) >> 8, 0), 0xffff);
}

// converts the staged registers of one chunk (four pixels) into LDS taps at `dst` and the four RGBA8 pixels draw.Draw would write
template <int SRC>
__device__ __forceinline__ u32x4 ks_convert(const KsStageRegs &s, uint8_t *dst, int hs)
{
    u32x4 px;
    if (SRC == KS_RGBA) {
        px.x = s.v[0]; px.y = s.v[1]; px.z = s.v[2]; px.w = s.v[3];
        *(u32x4 *)dst = px;
    } else if (SRC == KS_GRAY) {
        const uint32_t y4 = s.v[0];
        uint32_t lo = __builtin_amdgcn_perm(0u, y4, 0x01010000u), hi = __builtin_amdgcn_perm(0u, y4, 0x03030202u);   // y * 0x101 per pixel
        *(uint2 *)dst = make_uint2(lo, hi);
        px.x = __builtin_amdgcn_perm(0u, y4, 0x0d000000u); px.y = __builtin_amdgcn_perm(0u, y4, 0x0d010101u);       // (y, y, y, 0xff)
        px.z = __builtin_amdgcn_perm(0u, y4, 0x0d020202u); px.w = __builtin_amdgcn_perm(0u, y4, 0x0d030303u);
    } else if (SRC == KS_TAP64) {
        // (two 16-byte loads per chunk: v[] holds the first half here, the caller passes the second through `hs`-less overload)
        px = u32x4{0, 0, 0, 0};
    } else if (SRC == KS_NRGBA) {
        uint32_t t[8];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t p = s.v[i], a16 = (p >> 24) * 0x101u;   // scaleX_NRGBA: c * a16 / 0xff
            const uint32_t r = (p & 0xffu) * a16 / 0xffu, g = ((p >> 8) & 0xffu) * a16 / 0xffu, b = ((p >> 16) & 0xffu) * a16 / 0xffu;
            t[2 * i] = r | g << 16; t[2 * i + 1] = b | a16 << 16;
        }
        *(u32x4 *)dst = u32x4{t[0], t[1], t[2], t[3]};
        *(u32x4 *)(dst + 16) = u32x4{t[4], t[5], t[6], t[7]};
        px.x = __builtin_amdgcn_perm(t[1], t[0], 0x07050301u); px.y = __builtin_amdgcn_perm(t[3], t[2], 0x07050301u);   // drawNRGBASrc: the top bytes
        px.z = __builtin_amdgcn_perm(t[5], t[4], 0x07050301u); px.w = __builtin_amdgcn_perm(t[7], t[6], 0x07050301u);
    } else {   // KS_YCC: v[0] = four luma bytes, v[1] / v[2] = the Cb / Cr bytes of their samples (two when hs, else four)
        uint32_t t[8];
        // the chroma terms once per SAMPLE: two pixels share one where the planes are subsampled horizontally (hs is uniform; with the
        // sample index computed per pixel the compiler multiplied every pixel's terms anew: 6.8 multiplies per pixel instead of 3)
        int rt[4], gt[4], bt[4];
        if (hs) {
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const int cb1 = (int)((s.v[1] >> (8 * c)) & 0xffu) - 128, cr1 = (int)((s.v[2] >> (8 * c)) & 0xffu) - 128;
                rt[2 * c] = rt[2 * c + 1] = 91881 * cr1;
                gt[2 * c] = gt[2 * c + 1] = -22554 * cb1 - 46802 * cr1;
                bt[2 * c] = bt[2 * c + 1] = 116130 * cb1;
            }
        } else {
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int cb1 = (int)((s.v[1] >> (8 * c)) & 0xffu) - 128, cr1 = (int)((s.v[2] >> (8 * c)) & 0xffu) - 128;
                rt[c] = 91881 * cr1; gt[c] = -22554 * cb1 - 46802 * cr1; bt[c] = 116130 * cb1;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint32_t r, g, b;
            ks_ycc16((s.v[0] >> (8 * i)) & 0xffu, rt[i], gt[i], bt[i], r, g, b);
            t[2 * i] = r | g << 16; t[2 * i + 1] = b | 0xffff0000u;
        }
        *(u32x4 *)dst = u32x4{t[0], t[1], t[2], t[3]};
        *(u32x4 *)(dst + 16) = u32x4{t[4], t[5], t[6], t[7]};
        px.x = __builtin_amdgcn_perm(t[1], t[0], 0x07050301u); px.y = __builtin_amdgcn_perm(t[3], t[2], 0x07050301u);   // DrawYCbCr: the top bytes
        px.z = __builtin_amdgcn_perm(t[5], t[4], 0x07050301u); px.w = __builtin_amdgcn_perm(t[7], t[6], 0x07050301u);
    }
    return px;
}

// the 16-bit values of one tap as float64, NCH channels (1: gray, 3: colours of an opaque source, 4: colours and alpha)
// what one tap is in LDS: the packed pixel (RGBA), one 16-bit value (Gray), four 16-bit values
template <int SRC> struct KsTapRaw { typedef uint2 type; };
template <> struct KsTapRaw<KS_RGBA> { typedef uint32_t type; };
template <> struct KsTapRaw<KS_GRAY> { typedef uint16_t type; };
template <int SRC, int NCH, int MODE>
__device__ __forceinline__ void ks_fetch(typename KsTapRaw<SRC>::type raw, double (&v)[NCH])
{
    if constexpr (SRC == KS_RGBA) {
        const uint32_t px = raw;
        if (MODE == KS_TAP_CLAMP && NCH == 4) {
            const uint32_t al = px >> 24;
            v[0] = (double)(min(px & 0xffu, al) * 0x101u);
            v[1 % NCH] = (double)(min((px >> 8) & 0xffu, al) * 0x101u);
            v[2 % NCH] = (double)(min((px >> 16) & 0xffu, al) * 0x101u);
            v[3 % NCH] = (double)(al * 0x101u);
        } else {
            v[0] = widen<0>(px);
            if (NCH > 1) v[1 % NCH] = widen<1>(px);
            if (NCH > 2) v[2 % NCH] = widen<2>(px);
            if (NCH > 3) v[3 % NCH] = widen<3>(px);
        }
    } else if constexpr (SRC == KS_GRAY) {
        v[0] = (double)(uint32_t)raw;
    } else {
        // one 8-byte LDS read per tap (three 16-bit reads per tap loaded the LDS pipe more than the two extractions cost the VALU)
        const uint2 t = raw;
        uint32_t c[4];
        if (MODE == KS_TAP_TOP) {                          // (v >> 8) * 0x101 straight from the packed halves
            c[0] = __builtin_amdgcn_perm(0u, t.x, 0x0c0c0101u); c[1] = __builtin_amdgcn_perm(0u, t.x, 0x0c0c0303u);
            c[2] = __builtin_amdgcn_perm(0u, t.y, 0x0c0c0101u); c[3] = __builtin_amdgcn_perm(0u, t.y, 0x0c0c0303u);
        } else {
            c[0] = t.x & 0xffffu; c[1] = t.x >> 16; c[2] = t.y & 0xffffu; c[3] = t.y >> 16;
            if (MODE == KS_TAP_MINTOP) {
#pragma unroll
                for (int k = 0; k < 3; k++) c[k] = __builtin_amdgcn_perm(0u, min(c[k], c[3]), 0x0c0c0101u);
                c[3] = __builtin_amdgcn_perm(0u, c[3], 0x0c0c0101u);
            }
        }
#pragma unroll
        for (int k = 0; k < NCH; k++) v[k] = (double)c[k];
    }
}

// the same tap as float (the float pass); an RGBA tile's bytes stay bytes -- the weights carry the 0x101
template <int SRC, int NCH, int MODE>
__device__ __forceinline__ void ks_fetchf(typename KsTapRaw<SRC>::type raw, float (&v)[NCH])
{
    if constexpr (SRC == KS_RGBA) {
        v[0] = (float)(raw & 0xffu);                                   // v_cvt_f32_ubyte0 .. 3
        if (NCH > 1) v[1 % NCH] = (float)((raw >> 8) & 0xffu);
        if (NCH > 2) v[2 % NCH] = (float)((raw >> 16) & 0xffu);
        if (NCH > 3) v[3 % NCH] = (float)(raw >> 24);
    } else if constexpr (SRC == KS_GRAY) {
        v[0] = (float)(uint32_t)raw;
    } else {
        const uint2 t = raw;
        uint32_t c[4];
        if (MODE == KS_TAP_TOP) {                          // the top bytes themselves (v_cvt_f32_ubyte1 / 3): this output's weights carry the 0x101
            c[0] = (t.x >> 8) & 0xffu; c[1] = t.x >> 24; c[2] = (t.y >> 8) & 0xffu; c[3] = t.y >> 24;
        } else {
            c[0] = t.x & 0xffffu; c[1] = t.x >> 16; c[2] = t.y & 0xffffu; c[3] = t.y >> 16;
        }
#pragma unroll
        for (int k = 0; k < NCH; k++) v[k] = (float)c[k];
    }
}

template <int NCH, int NACC, class T = double>
struct KsCol {                 // one destination column of a lane
    T q[NACC][NCH];            // scaleY's running sums
    double itwf;               // invTotalWeightFFFF of the column (float64 kernels)
    int xb;                    // LDS byte offset of the column's first tap within a tile row
    int wofs;                  // LDS byte offset of the column's first weight
    int ooff;                  // byte offset of the column in a destination row; kOOB = the lane has no such column
    __device__ __forceinline__ void clear()
    {
#pragma unroll
        for (int p = 0; p < NACC; p++)
#pragma unroll
            for (int k = 0; k < NCH; k++) q[p][k] = 0;
    }
};
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int NCH, int NACC>
struct KsCol<NCH, NACC, float> {   // the float pass: accumulators 2p and 2p + 1 of a channel side by side (one v_pk_fma_f32 feeds both)
    f32x2 q2[NCH][NACC / 2];
    double itwf;
    int xb, wofs, ooff;
    __device__ __forceinline__ void clear()
    {
#pragma unroll
        for (int k = 0; k < NCH; k++)
#pragma unroll
            for (int h = 0; h < NACC / 2; h++) q2[k][h] = f32x2{0.f, 0.f};
    }
};

// scaleX on the B rows of the tile for one column, then scaleY's accumulation and, where a destination row completes, its store.
// AONE: the source type's scaleX writes tmp alpha = 1 (Gray, YCbCr): scaleY's alpha sum is the row's `ones`, and colours are clamped to it.
template <int SRC, int NCH, int NACC, int B, int MODE, bool AONE>
__device__ __forceinline__ void ks_column(const uint8_t *lds, const uint8_t *tile, KsCol<NCH, NACC> &c, int ntap, int wstride, int pitch,
                                          const uint8_t *rows, __amdgpu_buffer_rsrc_t ors, int ostride
#if IPX_DIAG
                                          , unsigned long long *tsum, unsigned long long &tlast
#endif
                                          )
{
    constexpr int PXB = KsPx<SRC>::bytes;
    double acc[B][NCH];
#pragma unroll
    for (int r = 0; r < B; r++)
#pragma unroll
        for (int k = 0; k < NCH; k++) acc[r][k] = 0.0;
    const uint8_t *tap = tile + c.xb;
    const uint8_t *wp = lds + c.wofs;
    // Two taps per iteration.  Their LDS reads are issued together and kept together (sched_barrier: left alone, the scheduler spreads
    // them between the arithmetic of the previous row, each followed by a wait of its own).  A column's taps are added in source order.
    typedef typename KsTapRaw<SRC>::type Raw;
    auto one_tap = [&](Raw raw, double w, int r) {
        double v[NCH];
        ks_fetch<SRC, NCH, MODE>(raw, v);
#pragma unroll
        for (int k = 0; k < NCH; k++) acc[r][k] += v[k] * w;
    };
    auto taps = [&](auto nt) {                           // nt taps: all their LDS reads first, then the arithmetic tap by tap
        constexpr int NT = decltype(nt)::value;
        double w[NT];
        Raw px[NT][B];
#pragma unroll
        for (int i = 0; i < NT; i++) {
            w[i] = *(const double *)(wp + i * wstride);
#pragma unroll
            for (int r = 0; r < B; r++) px[i][r] = *(const Raw *)(tap + r * pitch + i * PXB);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NT; i++)
#pragma unroll
            for (int r = 0; r < B; r++) one_tap(px[i][r], w[i], r);
        tap += NT * PXB;
        wp += NT * wstride;
    };
    int t = 0;
    constexpr int TU = SRC == KS_RGBA || SRC == KS_GRAY ? 4 : 2;   // (8-byte taps: two at a time keep the registers)
    for (; t + TU <= ntap; t += TU) taps(std::integral_constant<int, TU>());
    if (TU > 2 && t + 2 <= ntap) { taps(std::integral_constant<int, 2>()); t += 2; }
    if (t < ntap) taps(std::integral_constant<int, 1>());
    KS_STAMP(4);                                         // scaleX of the group's rows for this column
    typedef KsRowT<NACC> Row;
    // the group's row entries in one batch of LDS reads (read one by one where they are used, every read was a round trip of its own)
    double rw[B][NACC], ritw[B][NACC], rone[B][NACC];
    int remit[B][NACC];
#pragma unroll
    for (int r = 0; r < B; r++) {
        const Row *row = (const Row *)(rows + r * sizeof(Row));
#pragma unroll
        for (int p = 0; p < NACC; p++) {
            rw[r][p] = row->w[p]; ritw[r][p] = row->itw[p]; remit[r][p] = row->emit[p];
            if (AONE) rone[r][p] = row->ones[p];
        }
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
                } else if (AONE) {
                    const double pa = rone[r][p];                      // what scaleY sums from a column of tmp alphas that are all 1
                    double pr = c.q[p][0], pg = c.q[p][1 % NCH], pb = c.q[p][2 % NCH];
                    if (pr > pa) pr = pa;
                    if (NCH == 1) px = ks_ftou8(pr * s) * 0x010101u | ks_ftou8(pa * s) << 24;
                    else {
                        if (pg > pa) pg = pa;
                        if (pb > pa) pb = pa;
                        px = ks_ftou8(pr * s) | ks_ftou8(pg * s) << 8 | ks_ftou8(pb * s) << 16 | ks_ftou8(pa * s) << 24;
                    }
                } else if (NCH == 1) {
                    px = ks_ftou8(c.q[p][0] * s) * 0x010101u | 0xff000000u;   // an opaque gray frame: three equal colour sums, alpha 0xff
                } else {
                    // an opaque frame: no colour sum can exceed the alpha sum (rounding is monotonic), alpha is 0xff
                    px = ks_ftou8(c.q[p][0] * s) | ks_ftou8(c.q[p][1 % NCH] * s) << 8 | ks_ftou8(c.q[p][2 % NCH] * s) << 16 | 0xff000000u;
                }
                __builtin_amdgcn_raw_buffer_store_b32(px, ors, c.ooff, dy * ostride, 0);
#pragma unroll
                for (int k = 0; k < NCH; k++) c.q[p][k] = 0.0;
            }
        }
    }
}

// The pixels a wave of the float pass could not decide wait in LDS (kKsOpenPerWave per wave) and go to the frame's list in HBM a batch
// at a time: the append needs the list's old count back, and a wave that waits for an atomic's return waits for every load it has in
// flight as well -- the next group's pixels.  One such wait per hundred pixels, not one per pixel.
struct KsOpen {
    uint2 *wave;               // this wave's entries in LDS
    int n, room;               // how many (wave-uniform); how many fit
    uint2 *list; int *count; int cap;   // the frame's list in HBM
    bool full;                 // the list had no room for an entry of this lane
};
__device__ __forceinline__ void ks_open_flush(KsOpen &o)
{
    if (o.n == 0) return;                                              // wave-uniform
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    int base = 0;
    if (lane == 0) base = atomicAdd(o.count, o.n);
    base = __builtin_amdgcn_readfirstlane(base);
    for (int i = lane; i < o.n; i += 64) {
        if (base + i < o.cap) o.list[base + i] = o.wave[i];
        else o.full = true;
    }
    o.n = 0;
}

// The float pass's columns: the same walk with float sums, for the NC columns of a lane TOGETHER -- their LDS reads are issued in one
// batch (twice as many in flight per wait: with three waves per SIMD it is the LDS round trips that the arithmetic waits for) and the
// group's row entries are read once for both.
template <int SRC, int NCH, int NACC, int B, int MODE, int NC>
__device__ __forceinline__ void ks_columns_fast(const uint8_t *lds, const uint8_t *tile, KsCol<NCH, NACC, float> *c, int ntap, int wstride, int pitch,
                                                const uint8_t *rows, __amdgpu_buffer_rsrc_t ors, int ostride, float feps, bool split2, KsOpen &op
#if IPX_DIAG
                                                , unsigned long long *tsum, unsigned long long &tlast
#endif
                                                )
{
    constexpr int PXB = KsPx<SRC>::bytes;
    static_assert(B % 2 == 0 && NACC % 2 == 0, "rows and accumulators go in pairs");
    f32x2 acc[NC][B / 2][NCH];          // rows r and r + 1 of a channel side by side: one v_pk_fma_f32 adds a tap to both
    const uint8_t *tap[NC], *wp[NC];
#pragma unroll
    for (int j = 0; j < NC; j++) {
#pragma unroll
        for (int r = 0; r < B / 2; r++)
#pragma unroll
            for (int k = 0; k < NCH; k++) acc[j][r][k] = f32x2{0.f, 0.f};
        tap[j] = tile + c[j].xb;
        wp[j] = lds + c[j].wofs;
    }
    typedef typename KsTapRaw<SRC>::type Raw;
    auto taps = [&](auto nt) {
        constexpr int NT = decltype(nt)::value;
        float w[NC][NT];
        Raw px[NC][NT][B];
#pragma unroll
        for (int j = 0; j < NC; j++)
#pragma unroll
            for (int i = 0; i < NT; i++) {
                w[j][i] = *(const float *)(wp[j] + i * wstride);
#pragma unroll
                for (int r = 0; r < B; r++) px[j][i][r] = *(const Raw *)(tap[j] + r * pitch + i * PXB);
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NC; j++) {
#pragma unroll
            for (int i = 0; i < NT; i++)
#pragma unroll
                for (int r = 0; r < B; r += 2) {
                    float v0[NCH], v1[NCH];
                    ks_fetchf<SRC, NCH, MODE>(px[j][i][r], v0);
                    ks_fetchf<SRC, NCH, MODE>(px[j][i][r + 1], v1);
                    const f32x2 ww{w[j][i], w[j][i]};
#pragma unroll
                    for (int k = 0; k < NCH; k++) acc[j][r / 2][k] = __builtin_elementwise_fma(f32x2{v0[k], v1[k]}, ww, acc[j][r / 2][k]);
                }
            tap[j] += NT * PXB;
            wp[j] += NT * wstride;
        }
    };
    int t = 0;
    constexpr int TU = SRC == KS_RGBA || SRC == KS_GRAY ? 4 : 2;   // (8-byte taps: two at a time keep the registers)
    for (; t + TU <= ntap; t += TU) taps(std::integral_constant<int, TU>());
    if (TU > 2 && t + 2 <= ntap) { taps(std::integral_constant<int, 2>()); t += 2; }
    for (; t < ntap; t++) taps(std::integral_constant<int, 1>());
    if (NC == 1 && split2) {
        // two lanes per column: each summed half of the taps (the order of a float sum is free), the halves meet here -- v_add with a
        // DPP operand, quad_perm [1, 0, 3, 2]; both lanes go on with the total, the even one owns the column's stores
        auto other = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)); };
#pragma unroll
        for (int r = 0; r < B / 2; r++)
#pragma unroll
            for (int k = 0; k < NCH; k++) {
                acc[0][r][k].x += other(acc[0][r][k].x);
                acc[0][r][k].y += other(acc[0][r][k].y);
            }
    }
    KS_STAMP(4);                                         // scaleX of the group's rows for these columns
    typedef KsRowT<NACC> Row;
    f32x2 rw[B][NACC / 2];
    int remit[B][NACC];
#pragma unroll
    for (int r = 0; r < B; r++) {
        const Row *row = (const Row *)(rows + r * sizeof(Row));
#pragma unroll
        for (int p = 0; p < NACC; p++) remit[r][p] = row->emit[p];
#pragma unroll
        for (int h = 0; h < NACC / 2; h++) rw[r][h] = f32x2{row->wf[2 * h], row->wf[2 * h + 1]};
    }
#pragma unroll
    for (int r = 0; r < B; r++) {
#pragma unroll
        for (int h = 0; h < NACC / 2; h++)                             // the row's value into both accumulators of a pair
#pragma unroll
            for (int j = 0; j < NC; j++)
#pragma unroll
                for (int k = 0; k < NCH; k++) {
                    const float v = acc[j][r / 2][k][r & 1];
                    c[j].q2[k][h] = __builtin_elementwise_fma(f32x2{v, v}, rw[r][h], c[j].q2[k][h]);
                }
#pragma unroll
        for (int p = 0; p < NACC; p++) {
            const int dy = __builtin_amdgcn_readfirstlane(remit[r][p]);
            if (dy >= 0) {                                             // wave-uniform
#pragma unroll
                for (int j = 0; j < NC; j++) {
                    // T = (V' + 0.5) / 256 in one rounding (scaling by 2^-8 is exact); the byte is floor(T) -- below 256 for every tap count
                    // the float pass takes (ks_float_eps) -- and the channel is open when T lies within feps * T of an integer (T - rint(T)
                    // is exact).  v_cvt_pk_u8_f32 converts an integer-valued float and drops it into its byte of the pixel.
                    uint32_t px = 0xff000000u;
                    bool open = false;                                 // a channel too close to a multiple of 256 to call
#pragma unroll
                    for (int k = 0; k < NCH; k++) {
                        const float u = __builtin_fmaf(c[j].q2[k][p / 2][p & 1], 1.0f / 256.0f, 0.5f / 256.0f);
                        const float d = u - __builtin_rintf(u);
                        open |= __builtin_fabsf(d) < feps * u;
                        px = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_floorf(u), (uint32_t)k, px);
                        c[j].q2[k][p / 2][p & 1] = 0.f;
                    }
                    if (NCH == 1) px = (px & 0xffu) * 0x010101u | 0xff000000u;
                    __builtin_amdgcn_raw_buffer_store_b32(px, ors, c[j].ooff, dy * ostride, 0);
                    open = open && c[j].ooff != kOOB;
                    const unsigned long long m = __ballot(open);
                    if (m) {                                           // wave-uniform; about one store in thirty
                        const int k = __popcll(m);
                        if (op.n + k > op.room) ks_open_flush(op);
                        const int slot = op.n + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));   // open lanes below this one
                        if (open) op.wave[slot] = make_uint2((uint32_t)dy, (uint32_t)c[j].ooff >> 2);
                        op.n += k;
                    }
                }
            }
        }
    }
}

template <int SRC, int NCH, int NACC, int B, bool OPQ, bool RAG, bool FAST>
__global__ __launch_bounds__(kKsMaxThreads) void ks_fused_kernel(KsFusedArgs a)
{
    typedef std::conditional_t<FAST, float, double> Sum;
    constexpr int CPLM = NACC == 2 ? kKsMaxCpl : 1;   // columns per lane: four accumulators per column leave registers for one (ks_fused_plan knows)
    extern __shared__ __align__(16) uint8_t lds[];
    constexpr int PXB = KsPx<SRC>::bytes;
    constexpr int CHB = 4 * PXB;                      // LDS bytes of a chunk (four pixels)
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int item = (int)blockIdx.x;
    if (!OPQ && !FAST && a.redo && !a.redo[item]) return;
    const int seg = item % a.nseg;
    item /= a.nseg;
    const int strip = item % a.nstrips, frame = item / a.nstrips;
    const KsStrip st = a.strips[strip];
    const KsSeg sg = a.segs[seg];
    const int pitch = a.pitch, CH = pitch / CHB;
    typedef KsRowT<NACC> Row;
    constexpr int RW = (int)(sizeof(Row) / 4);       // dwords per row entry
    constexpr int SPX = SRC == KS_RGBA || SRC == KS_NRGBA ? 4 : SRC == KS_TAP64 ? 8 : 1;   // source bytes per pixel (luma plane for YCbCr / Gray)
    const int hs = SRC == KS_YCC && (a.ratio == IPX_YCBCR_422 || a.ratio == IPX_YCBCR_420), vs = SRC == KS_YCC && (a.ratio == IPX_YCBCR_420 || a.ratio == IPX_YCBCR_440);

    const uint8_t *sframe = a.src + (size_t)frame * a.src_fs;
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc((void *)sframe, 0, (a.sh - 1) * a.sstride + a.sw * SPX, 0x00020000);
    const int chh = vs ? (a.sh + 1) / 2 : a.sh, cww = hs ? (a.sw + 1) / 2 : a.sw;
    const __amdgpu_buffer_rsrc_t cbrs = __builtin_amdgcn_make_buffer_rsrc((void *)(SRC == KS_YCC ? a.cb + (size_t)frame * a.c_fs : nullptr), 0,
                                                                         SRC == KS_YCC ? (chh - 1) * a.cstride + cww : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t crrs = __builtin_amdgcn_make_buffer_rsrc((void *)(SRC == KS_YCC ? a.cr + (size_t)frame * a.c_fs : nullptr), 0,
                                                                         SRC == KS_YCC ? (chh - 1) * a.cstride + cww : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)(a.wm ? a.wm + (size_t)frame * a.wm_fs : nullptr), 0,
                                                                        a.wm ? (a.sh - 1) * a.wm_stride + a.sw * 4 : 0, 0x00020000);

    // ---- what this thread stages per group: up to kKsMaxStage chunks of four pixels, and one dword of the row entries ----
    int s_lds[kKsMaxStage], s_src[kKsMaxStage], s_wm[kKsMaxStage], s_row[kKsMaxStage], s_cx[kKsMaxStage], s_np[kKsMaxStage];
    constexpr bool ragged = RAG && (SRC == KS_RGBA || SRC == KS_NRGBA);   // the frame's last chunk of a row holds fewer than four pixels (a.sw & 3)
#pragma unroll
    for (int i = 0; i < kKsMaxStage; i++) {
        const int q = tid + i * a.nthreads;
        const int row = q / CH, ch = q - row * CH, x = st.t0 + ch * 4;
        const bool in = q < B * CH && x < a.sw;
        s_row[i] = in ? row : -1;
        s_lds[i] = row * pitch + ch * CHB;
        s_src[i] = in ? row * a.sstride + x * SPX : kOOB;
        s_cx[i] = in ? (hs ? x >> 1 : x) : kOOB;         // chroma byte offset within its row
        s_np[i] = !ragged ? 4 : in ? min(4, a.sw - x) : 0;   // pixels of the chunk inside the frame
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
    KsCol<NCH, NACC, Sum> col[CPLM];
    __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void *)nullptr, 0, 0, 0x00020000);
    int ntap = 0, wstride = 0, ostride = 0, cpl = 0, mode = KS_TAP_PLAIN;
    float feps = 0.f;
    bool split2 = false;                                 // the float pass: two lanes per column of this wave's output
    bool aone = false;
    const uint8_t *rows_lds = lds + a.lds_rows;
    const int dbuf = a.dbuf;
    const int tile_bytes = B * pitch, rows_bytes = 2 * B * (int)sizeof(Row);   // a second tile and a second set of row entries when dbuf
    if (role >= 0) {
        const KsFusedOut &o = a.o[role];
        ors = __builtin_amdgcn_make_buffer_rsrc((void *)(o.out + (size_t)frame * o.frame_stride), 0, o.obytes, 0x00020000);
        split2 = FAST && o.split == 2;
        ntap = FAST ? (split2 ? o.ntapf >> 1 : o.ntapf) : o.ntap;      // taps per LANE
        wstride = o.wcols * (FAST ? 4 : 8); ostride = o.ostride; cpl = o.cpl; feps = o.feps;
        mode = o.mode; aone = o.aone != 0;
        rows_lds += role * B * (int)sizeof(Row);
        const int cb = o.colb[strip], ce = o.colb[strip + 1];
        // the strip's weight table -> LDS (every wave of the role copies a share)
        {
            const int n = (FAST ? o.ntapf : o.ntap) * o.wcols;
            if (FAST) {
                const float *wsrc = o.wxf + (size_t)strip * o.ntapf * o.wcols;
                float *wdst = (float *)(lds + a.lds_w[role]);
                for (int i = wk * 64 + lane; i < n; i += o.waves * 64) wdst[i] = wsrc[i];
            } else {
                const double *wsrc = o.wx + (size_t)strip * o.ntap * o.wcols;
                double *wdst = (double *)(lds + a.lds_w[role]);
                for (int i = wk * 64 + lane; i < n; i += o.waves * 64) wdst[i] = wsrc[i];
            }
        }
#pragma unroll
        for (int j = 0; j < CPLM; j++) {
            const int slot = wk * 64 + lane + j * 64 * o.waves;
            const int part = split2 ? slot & 1 : 0, cslot = split2 ? slot >> 1 : slot, dx = cb + cslot;   // (two lanes per column: which half of the taps)
            const bool has = j < o.cpl && dx < ce;
            col[j].clear();
            col[j].itwf = has ? o.itwf[dx] : 0.0;
            col[j].xb = has ? (o.sr_x0 + o.xlo[dx] - st.t0 + part * ntap) * PXB : 0;
            col[j].wofs = a.lds_w[role] + (has ? cslot : 0) * (FAST ? 4 : 8) + part * ntap * wstride;
            col[j].ooff = has && part == 0 ? dx * 4 : kOOB;
        }
    }

    const int ngroups = (sg.r1 - sg.ys + B - 1) / B;
    KsStageRegs stage[kKsMaxStage];
    uint32_t stage2[SRC == KS_TAP64 ? kKsMaxStage : 1][4];   // the second 16 bytes of a chunk of 8-byte pixels
    uint32_t rstg = 0;
    auto issue = [&](int g) {
        const int y0 = sg.ys + g * B;
#pragma unroll
        for (int i = 0; i < kKsMaxStage; i++) {
            const bool ok = s_row[i] >= 0 && y0 + s_row[i] < sg.r1;
            const int off = ok ? s_src[i] : kOOB;
            if (SRC == KS_RGBA || SRC == KS_NRGBA || SRC == KS_TAP64) {
                u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(srs, ragged && s_np[i] < 4 ? kOOB : off, y0 * a.sstride, 0);
                if (ragged) {   // (uniform) the short chunk pixel by pixel: a 16-byte load would read past the row, and past the frame in its last row
                    const int np = ok ? s_np[i] : 4;
                    v.x |= __builtin_amdgcn_raw_buffer_load_b32(srs, np < 4 ? off : kOOB, y0 * a.sstride, 0);
                    v.y |= __builtin_amdgcn_raw_buffer_load_b32(srs, np == 2 || np == 3 ? off + 4 : kOOB, y0 * a.sstride, 0);
                    v.z |= __builtin_amdgcn_raw_buffer_load_b32(srs, np == 3 ? off + 8 : kOOB, y0 * a.sstride, 0);
                }
                stage[i].v[0] = v.x; stage[i].v[1] = v.y; stage[i].v[2] = v.z; stage[i].v[3] = v.w;
                if (SRC == KS_TAP64) {
                    const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(srs, ok ? s_src[i] + 16 : kOOB, y0 * a.sstride, 0);
                    stage2[SRC == KS_TAP64 ? i : 0][0] = u.x; stage2[SRC == KS_TAP64 ? i : 0][1] = u.y;
                    stage2[SRC == KS_TAP64 ? i : 0][2] = u.z; stage2[SRC == KS_TAP64 ? i : 0][3] = u.w;
                }
            } else {
                stage[i].v[0] = __builtin_amdgcn_raw_buffer_load_b32(srs, off, y0 * a.sstride, 0);     // four luma bytes
                if (SRC == KS_YCC) {
                    const int cy = vs ? (y0 + s_row[i]) >> 1 : y0 + s_row[i];
                    const int coff = ok ? cy * a.cstride + s_cx[i] : kOOB;
                    if (hs) {
                        stage[i].v[1] = (uint32_t)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(cbrs, coff, 0, 0);
                        stage[i].v[2] = (uint32_t)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(crrs, coff, 0, 0);
                    } else {
                        stage[i].v[1] = __builtin_amdgcn_raw_buffer_load_b32(cbrs, coff, 0, 0);
                        stage[i].v[2] = __builtin_amdgcn_raw_buffer_load_b32(crrs, coff, 0, 0);
                    }
                }
            }
        }
        if (rstage) rstg = rsrc[(size_t)g * B * RW];
    };
    issue(0);
    bool bad = false;
    KsOpen open;
    if (FAST) {
        const int pk = role >= 0 ? a.o[role].pk : 0;
        open.wave = (uint2 *)(lds + a.lds_open) + wv * a.open_per_wave;
        open.n = 0; open.room = a.open_per_wave; open.full = false;
        open.list = a.fix + (size_t)frame * a.fix_stride + (pk ? a.fix_cap[0] : 0); open.count = a.fix_count + 2 * frame + pk; open.cap = a.fix_cap[pk];
    }
#if IPX_DIAG
    unsigned long long tsum[6] = {0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif
    for (int g = 0; g < ngroups; g++) {
#if IPX_DIAG
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        KS_STAMP(0);                                     // waiting for the group's staged loads
#endif
        const int y0 = sg.ys + g * B, buf = dbuf ? g & 1 : 0;
        // ---- registers -> taps in the LDS tile, pixels in the watermark frame ----
#pragma unroll
        for (int i = 0; i < kKsMaxStage; i++) {
            if (s_row[i] < 0) continue;
            uint8_t *dst = lds + buf * tile_bytes + s_lds[i];
            u32x4 v;
            if (SRC == KS_TAP64) {
                const uint32_t *s0 = stage[i].v, *s1 = stage2[SRC == KS_TAP64 ? i : 0];
                *(u32x4 *)dst = u32x4{s0[0], s0[1], s0[2], s0[3]};
                *(u32x4 *)(dst + 16) = u32x4{s1[0], s1[1], s1[2], s1[3]};
                v.x = __builtin_amdgcn_perm(s0[1], s0[0], 0x07050301u); v.y = __builtin_amdgcn_perm(s0[3], s0[2], 0x07050301u);   // drawRGBA with Src: the top bytes
                v.z = __builtin_amdgcn_perm(s1[1], s1[0], 0x07050301u); v.w = __builtin_amdgcn_perm(s1[3], s1[2], 0x07050301u);
            } else v = ks_convert<SRC>(stage[i], dst, hs);
            const int y = y0 + s_row[i];
            const bool live = y < sg.r1;
            if (OPQ && live) {
                uint32_t m = v.x & v.y & v.z & v.w;
                if (ragged && s_np[i] < 4) m = v.x & (s_np[i] > 1 ? v.y : ~0u) & (s_np[i] > 2 ? v.z : ~0u);
                bad |= (m >> 24) != 0xffu;
            }
            if (a.wm && live && y >= sg.r0) {
                if (!ragged) __builtin_amdgcn_raw_buffer_store_b128(v, wrs, s_wm[i], y0 * a.wm_stride, 0);
                else {
                    const int np = s_np[i], wo = s_wm[i];
                    __builtin_amdgcn_raw_buffer_store_b128(v, wrs, np == 4 ? wo : kOOB, y0 * a.wm_stride, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(v.x, wrs, np < 4 ? wo : kOOB, y0 * a.wm_stride, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(v.y, wrs, (np == 2 || np == 3) && wo != kOOB ? wo + 4 : kOOB, y0 * a.wm_stride, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(v.z, wrs, np == 3 && wo != kOOB ? wo + 8 : kOOB, y0 * a.wm_stride, 0);
                }
            }
        }
        if (rstage) ((uint32_t *)(lds + a.lds_rows + buf * rows_bytes))[rk * B * RW + ri] = rstg;
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
        if (role >= 0) {
            const uint8_t *tile = lds + buf * tile_bytes, *rws = rows_lds + buf * rows_bytes;
#define KS_COLS(MODE, AONE)                                                                                                              \
    _Pragma("unroll") for (int j = 0; j < CPLM; j++)                                                                                 \
        if (j < cpl) ks_column<SRC, NCH, NACC, B, MODE, AONE>(lds, tile, col[j], ntap, wstride, pitch, rws, ors, ostride KS_DIAG_ARGS)
#define KS_COLSF(MODE)                                                                                                                   \
    do {                                                                                                                                 \
        if (CPLM > 1 && cpl > 1) ks_columns_fast<SRC, NCH, NACC, B, MODE, CPLM>(lds, tile, col, ntap, wstride, pitch, rws, ors, ostride, feps, false, open KS_DIAG_ARGS); \
        else if (cpl > 0) ks_columns_fast<SRC, NCH, NACC, B, MODE, 1>(lds, tile, col, ntap, wstride, pitch, rws, ors, ostride, feps, split2, open KS_DIAG_ARGS);          \
    } while (0)
            if constexpr (FAST) {
                // (the alpha of these sources never reaches the sums: opaque RGBA, YCbCr and Gray store 0xff)
                if constexpr (SRC == KS_YCC) {
                    if (!aone) { KS_COLSF(KS_TAP_TOP); }
                    else { KS_COLSF(KS_TAP_PLAIN); }
                } else if constexpr (SRC == KS_TAP64) {
                    // (only for taps whose colours never exceed their alpha: then min(c, a) is c and the crop copy's tap is the top byte)
                    if (mode == KS_TAP_MINTOP) { KS_COLSF(KS_TAP_TOP); }
                    else { KS_COLSF(KS_TAP_PLAIN); }
                } else if constexpr (SRC == KS_NRGBA) {
                    // (a colour tap of an NRGBA pixel, c * a16 / 0xff, never exceeds its alpha tap: the reference's clamp of a colour
                    // sum to the alpha sum never fires, and all four channels are plain sums)
                    if (mode == KS_TAP_TOP) { KS_COLSF(KS_TAP_TOP); }
                    else { KS_COLSF(KS_TAP_PLAIN); }
                } else { KS_COLSF(KS_TAP_PLAIN); }
            } else if (SRC == KS_RGBA) {
                if (NCH == 4 && mode == KS_TAP_CLAMP) { KS_COLS(KS_TAP_CLAMP, false); }
                else { KS_COLS(KS_TAP_PLAIN, false); }
            } else if (SRC == KS_GRAY) {
                if (aone) { KS_COLS(KS_TAP_PLAIN, true); }
                else { KS_COLS(KS_TAP_PLAIN, false); }
            } else if (SRC == KS_YCC) {
                if (aone) { KS_COLS(KS_TAP_PLAIN, true); }
                else { KS_COLS(KS_TAP_TOP, false); }
            } else if (SRC == KS_NRGBA) {
                if (mode == KS_TAP_TOP) { KS_COLS(KS_TAP_TOP, false); }
                else { KS_COLS(KS_TAP_PLAIN, false); }
            } else {
                if (mode == KS_TAP_MINTOP) { KS_COLS(KS_TAP_MINTOP, false); }
                else { KS_COLS(KS_TAP_PLAIN, false); }
            }
#undef KS_COLS
#undef KS_COLSF
        }
        KS_STAMP(5);                                     // scaleY's sums and finished rows
        // one tile buffer: everyone is done with it before the next group overwrites it.  Two: a wave that writes buffer g & 1 two groups
        // on has passed the barrier of group g + 1, which every wave reaches only after its arithmetic on group g
        if (!dbuf) ks_barrier();
    }
#if IPX_DIAG
    if (a.stamps && lane == 0)
        for (int i = 0; i < 6; i++) atomicAdd(&a.stamps[(role < 0 ? 2 : role) * 8 + i], tsum[i]);
#endif
    if (FAST) {
        ks_open_flush(open);
        bad |= open.full;
    }
    if (OPQ || FAST) {                                   // any pixel with alpha != 0xff in the rest of the item, or a full list: redo it as well
        const int any = __syncthreads_or(bad);
        if (tid == 0) a.redo[blockIdx.x] = any;
    }
}

template <int SRC, int NCH, int NACC, bool OPQ, bool RAG, bool FAST>
hipError_t launch_rag(const KsFusedPlan &p, const KsFusedArgs &a, int nitems, hipStream_t s)
{
    static KernelLaunchCache cache;
    auto fn = ks_fused_kernel<SRC, NCH, NACC, kKsRows, OPQ, RAG, FAST>;
    int resident = 0;
    const int lds_bytes = FAST ? p.fast.lds_bytes : p.lds_bytes;
    hipError_t e = cache.prepare((const void *)fn, p.nthreads, (size_t)lds_bytes, getenv("IPX_KS_DEBUG") ? &resident : nullptr);
    if (e != hipSuccess) return e;
    if (resident) fprintf(stderr, "[ipx ks] %d workgroups of %d threads with %d bytes of LDS resident per CU\n", resident, p.nthreads, lds_bytes);
    hipLaunchKernelGGL(fn, dim3(nitems), dim3(p.nthreads), (size_t)lds_bytes, s, a);
    return hipGetLastError();
}

template <int SRC, int NCH, int NACC, bool OPQ, bool FAST = false>
hipError_t launch_one(const KsFusedPlan &p, const KsFusedArgs &a, int nitems, hipStream_t s)
{
    if constexpr (SRC == KS_RGBA || SRC == KS_NRGBA) {
        if (a.sw & 3) return launch_rag<SRC, NCH, NACC, OPQ, true, FAST>(p, a, nitems, s);
    }
    return launch_rag<SRC, NCH, NACC, OPQ, false, FAST>(p, a, nitems, s);
}

template <int SRC, int NCH>
hipError_t launch_src(const KsFusedPlan &p, const KsFusedArgs &a, int nitems, hipStream_t s)
{
    return p.nacc == 2 ? launch_one<SRC, NCH, 2, false>(p, a, nitems, s) : launch_one<SRC, NCH, 4, false>(p, a, nitems, s);
}

}  // namespace

hipError_t launch_ks_fused(const KsFusedPlan &p, KsFusedArgs &a, const KsFix *fix, int cus, hipStream_t s, bool *matched)
{
    *matched = false;
    if (!p.ok || a.nframes <= 0) return hipSuccess;
    // frames are addressed through buffer descriptors with aligned dword (chroma: word) loads
    if ((((uintptr_t)a.src) | (uintptr_t)a.sstride | a.src_fs) & 3) return hipSuccess;
    if (a.wm && ((((uintptr_t)a.wm) | (uintptr_t)a.wm_stride | a.wm_fs) & 3)) return hipSuccess;
    int src = KS_RGBA;
    switch (a.src_kind) {
    case IPX_SRC_RGBA: src = KS_RGBA; break;
    case IPX_SRC_NRGBA: src = KS_NRGBA; break;
    case IPX_SRC_TAP64: src = KS_TAP64; break;
    case IPX_SRC_YCBCR: {
        src = a.cstride == 0 ? KS_GRAY : KS_YCC;                  // a stride-0 chroma row of 128s: a Gray frame (ipx_plan_run_dev_gray)
        if (src == KS_YCC) {
            const bool hs = a.ratio == IPX_YCBCR_422 || a.ratio == IPX_YCBCR_420;
            const uintptr_t al = hs ? 1 : 3;
            if ((((uintptr_t)a.cb) | (uintptr_t)a.cr | (uintptr_t)a.cstride | a.c_fs) & al) return hipSuccess;
        }
        break;
    }
    default: return hipSuccess;
    }
    if ((a.sw & 3) && src != KS_RGBA && src != KS_NRGBA) return hipSuccess;   // ragged rows: only where a pixel is a dword
    for (int i = 0; i < a.nout; i++) {                             // the tap kind of each output as a mode of this source type's tile
        KsFusedOut &o = a.o[i];
        o.aone = 0; o.mode = KS_TAP_PLAIN;
        switch (o.kind) {
        case IPX_SRC_RGBA: case IPX_SRC_NRGBA: case IPX_SRC_TAP64: break;
        case IPX_SRC_YCBCR: o.aone = 1; break;
        case IPX_SRC_RGBA_CROP: o.mode = KS_TAP_CLAMP; break;
        case IPX_SRC_NRGBA_CROP: o.mode = KS_TAP_TOP; break;
        case IPX_SRC_YCBCR_CROP: o.mode = src == KS_GRAY ? KS_TAP_PLAIN : KS_TAP_TOP; break;   // gray: (y * 0x101 >> 8) * 0x101 is y * 0x101 again
        case IPX_SRC_TAP64_CROP: o.mode = KS_TAP_MINTOP; break;
        default: return hipSuccess;
        }
    }
    // large batches: one segment per frame (no row is staged twice); small ones: enough items to fill the chip
    const char *ev = getenv("IPX_KS_SPLIT");                       // test knob: 1 = always the split segmentation, 0 = never
    const bool whole = ev && *ev ? atoi(ev) == 0 : (long long)a.nframes * p.nstrips >= 2LL * cus;
    const KsFusedGeom &g = whole || p.split.nseg <= 1 ? p.whole : p.split;
    a.nstrips = p.nstrips; a.nseg = g.nseg; a.nthreads = p.nthreads; a.pitch = p.pitch;
    a.strips = p.strips; a.segs = g.segs;
    // where things live in LDS: the float64 kernels' layout, or the float pass's own (float weight tables, lists of undecided pixels)
    auto layout = [&](bool fl) {
        a.dbuf = fl ? p.fast.dbuf : p.dbuf;
        a.lds_rows = fl ? p.fast.lds_rows : p.lds_rows;
        a.lds_open = fl ? p.fast.lds_open : 0; a.open_per_wave = fl ? p.fast.open_per_wave : 0;
        for (int i = 0; i < a.nout; i++) a.lds_w[i] = fl ? p.fast.lds_w[a.o[i].pk] : p.lds_w[a.o[i].pk];
    };
    for (int i = 0; i < a.nout; i++) {                             // a.o[i].pk: which of the plan's outputs this is
        const int k = a.o[i].pk;
        const KsFusedPlan::Out &po = p.o[k];
        KsFusedOut &o = a.o[i];
        o.ntap = po.ntap; o.waves = po.waves; o.cpl = po.cpl; o.wcols = po.wcols;
        o.wx = po.wx; o.itwf = po.itwf; o.xlo = po.xlo; o.colb = po.colb; o.wxf = po.wxf; o.feps = po.feps; o.split = po.split; o.ntapf = po.ntapf;
        o.rows = g.rows[k]; o.rowoff = g.rowoff[k];
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
    const int n = (int)nitems;
    if (getenv("IPX_KS_DEBUG"))
        fprintf(stderr, "[ipx ks] src %d nacc %d frames %d strips %d segs %d threads %d pitch %d dbuf %d lds %d (float pass: dbuf %d lds %d) | out0 ntap %d waves %d cpl %d wcols %d | out1 ntap %d waves %d cpl %d wcols %d\n", src, p.nacc,
                a.nframes, a.nstrips, a.nseg, a.nthreads, a.pitch, p.dbuf, p.lds_bytes, p.fast.dbuf, p.fast.lds_bytes, a.o[0].ntap, a.o[0].waves, a.o[0].cpl, a.o[0].wcols, a.nout > 1 ? a.o[1].ntap : 0,
                a.nout > 1 ? a.o[1].waves : 0, a.nout > 1 ? a.o[1].cpl : 0, a.nout > 1 ? a.o[1].wcols : 0);
    // the float pass where the source type has one and the caller brought lists: float kernel, the listed pixels in float64, then the
    // float64 kernel on the items the float kernel gave up
    bool fast = fix && fix->list && a.redo && (src == KS_RGBA || src == KS_YCC || src == KS_GRAY || src == KS_NRGBA || (src == KS_TAP64 && a.taps_le_alpha));
    for (int i = 0; i < a.nout; i++) fast = fast && a.o[i].feps > 0.f;       // (an output with more taps than the margin's derivation covers)
    fast = fast && p.fast.lds_bytes > 0;
    layout(false);
    a.fix = fast ? fix->list : nullptr; a.fix_count = fast ? fix->count : nullptr;
    a.fix_cap[0] = fast ? fix->cap[0] : 0; a.fix_cap[1] = fast ? fix->cap[1] : 0; a.fix_stride = a.fix_cap[0] + a.fix_cap[1];
    auto exact_listed = [&]() {
        hipError_t e = hipSuccess;
        for (int i = 0; i < a.nout && e == hipSuccess; i++) {
            const KsFusedOut &o = a.o[i];
            KsGenArgs g{};
            g.dst = o.out; g.dstride = o.ostride; g.dst_fs = o.frame_stride;
            g.adr_x1 = o.dw; g.adr_y1 = o.dh; g.sr_x0 = o.sr_x0; g.sr_y0 = o.sr_y0;
            g.ax = fix->ax[o.pk]; g.ay = fix->ay[o.pk]; g.op = IPX_OP_SRC; g.kind = o.kind;
            g.src = a.src; g.sstride = a.sstride; g.src_fs = a.src_fs;
            g.cb = a.cb; g.cr = a.cr; g.cstride = a.cstride; g.ratio = a.ratio; g.c_fs = a.c_fs;
            g.nframes = a.nframes;
            e = launch_ks_fix(g, fix->list + (o.pk ? fix->cap[0] : 0), (size_t)a.fix_stride, fix->count + o.pk, 2, fix->cap[o.pk], s);
        }
        return e;
    };
    hipError_t e = hipSuccess;
    switch (src) {
    case KS_NRGBA:
        if (!fast) { a.redo = nullptr; return launch_src<KS_NRGBA, 4>(p, a, n, s); }
        layout(true);
        e = p.nacc == 2 ? launch_one<KS_NRGBA, 4, 2, false, true>(p, a, n, s) : launch_one<KS_NRGBA, 4, 4, false, true>(p, a, n, s);
        layout(false);
        if (e == hipSuccess) e = exact_listed();
        if (e == hipSuccess) e = launch_src<KS_NRGBA, 4>(p, a, n, s);
        return e;
    case KS_TAP64:
        if (!fast) { a.redo = nullptr; return launch_src<KS_TAP64, 4>(p, a, n, s); }
        layout(true);
        e = p.nacc == 2 ? launch_one<KS_TAP64, 4, 2, false, true>(p, a, n, s) : launch_one<KS_TAP64, 4, 4, false, true>(p, a, n, s);
        layout(false);
        if (e == hipSuccess) e = exact_listed();
        if (e == hipSuccess) e = launch_src<KS_TAP64, 4>(p, a, n, s);
        return e;
    case KS_YCC:
        if (!fast) { a.redo = nullptr; return launch_src<KS_YCC, 3>(p, a, n, s); }
        layout(true);
        e = p.nacc == 2 ? launch_one<KS_YCC, 3, 2, false, true>(p, a, n, s) : launch_one<KS_YCC, 3, 4, false, true>(p, a, n, s);
        layout(false);
        if (e == hipSuccess) e = exact_listed();
        if (e == hipSuccess) e = launch_src<KS_YCC, 3>(p, a, n, s);
        return e;
    case KS_GRAY:
        if (!fast) { a.redo = nullptr; return launch_src<KS_GRAY, 1>(p, a, n, s); }
        layout(true);
        e = p.nacc == 2 ? launch_one<KS_GRAY, 1, 2, false, true>(p, a, n, s) : launch_one<KS_GRAY, 1, 4, false, true>(p, a, n, s);
        layout(false);
        if (e == hipSuccess) e = exact_listed();
        if (e == hipSuccess) e = launch_src<KS_GRAY, 1>(p, a, n, s);
        return e;
    default: break;
    }
    if (!a.redo) return launch_src<KS_RGBA, 4>(p, a, n, s);        // the general kernel alone
    // the speculative opaque pass first; the general kernel then redoes the items that met a pixel with alpha != 0xff
    if (fast) {
        layout(true);
        e = p.nacc == 2 ? launch_one<KS_RGBA, 3, 2, true, true>(p, a, n, s) : launch_one<KS_RGBA, 3, 4, true, true>(p, a, n, s);
        layout(false);
        if (e == hipSuccess) e = exact_listed();
    } else e = p.nacc == 2 ? launch_one<KS_RGBA, 3, 2, true>(p, a, n, s) : launch_one<KS_RGBA, 3, 4, true>(p, a, n, s);
    if (e == hipSuccess) e = launch_src<KS_RGBA, 4>(p, a, n, s);
    return e;
}

}  // namespace ipx

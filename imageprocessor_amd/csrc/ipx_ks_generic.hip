// ipx_ks_generic.hip -- xdraw.BiLinear.Scale (x/image/draw kernelScaler) for any rectangles, either op and every source type: the
// per-operation seam (resize.go:121-125, thumbnail.go:128-131) and the fallback of the batched path for shapes the fused kernel
// (ipx_ks_fused.hip) is not built for.
//
// One thread per destination pixel.  The reference fills tmp[sy][dx] once per source row and destination column and then walks the
// columns; a destination pixel depends on tmp[sy][dx] for the rows sy of its vertical range only, so the thread recomputes those few
// tmp values itself -- the same float64 operations in the same order (horizontal sum in source-column order, times invTotalWeightFFFF;
// vertical sum in source-row order; clamp to alpha; times invTotalWeight; ftou) -- and no float64 image ever goes to HBM.  The price
// is that a tmp value is computed by every destination row that uses it (two for a downscale).  Bound: FP64 VALU and L2 gathers; this
// is not the throughput path.
#include <algorithm>
#include <cstdlib>

#include "ipx_ks.h"

#pragma clang fp contract(off)

#include "ipx_device.h"

namespace ipx {
namespace {

struct Tap4 { uint32_t r, g, b, a; };

// the four 16-bit values scaleX_<type> weights for the source pixel (x, y); see ipx_ks.h for the kinds
template <int KIND>
__device__ __forceinline__ Tap4 ks_tap(const KsGenArgs &a, int x, int y)
{
    Tap4 t;
    if (KIND == IPX_SRC_YCBCR || KIND == IPX_SRC_YCBCR_CROP) {
        const int cx = (a.ratio == IPX_YCBCR_422 || a.ratio == IPX_YCBCR_420) ? x / 2 : x;
        const int cy = (a.ratio == IPX_YCBCR_420 || a.ratio == IPX_YCBCR_440) ? y / 2 : y;
        const size_t ci = (size_t)cy * a.cstride + cx;
        const int yy1 = (int)a.src[(size_t)y * a.sstride + x] * 0x10101;
        const int cb1 = (int)a.cb[ci] - 128, cr1 = (int)a.cr[ci] - 128;
        t.r = (uint32_t)min(max((yy1 + 91881 * cr1) >> 8, 0), 0xffff);
        t.g = (uint32_t)min(max((yy1 - 22554 * cb1 - 46802 * cr1) >> 8, 0), 0xffff);
        t.b = (uint32_t)min(max((yy1 + 116130 * cb1) >> 8, 0), 0xffff);
        t.a = 0xffffu;
        if (KIND == IPX_SRC_YCBCR_CROP) { t.r = (t.r >> 8) * 0x101u; t.g = (t.g >> 8) * 0x101u; t.b = (t.b >> 8) * 0x101u; }
    } else if (KIND == IPX_SRC_TAP64 || KIND == IPX_SRC_TAP64_CROP) {
        const uint2 p = *(const uint2 *)(a.src + (size_t)y * a.sstride + (size_t)x * 8);
        t.r = p.x & 0xffffu; t.g = p.x >> 16; t.b = p.y & 0xffffu; t.a = p.y >> 16;
        if (KIND == IPX_SRC_TAP64_CROP) {
            t.r = (min(t.r, t.a) >> 8) * 0x101u; t.g = (min(t.g, t.a) >> 8) * 0x101u; t.b = (min(t.b, t.a) >> 8) * 0x101u;
            t.a = (t.a >> 8) * 0x101u;
        }
    } else {
        const uint32_t p = *(const uint32_t *)(a.src + (size_t)y * a.sstride + (size_t)x * 4);
        if (KIND == IPX_SRC_NRGBA || KIND == IPX_SRC_NRGBA_CROP) {
            t.a = (p >> 24) * 0x101u;
            t.r = (p & 0xffu) * t.a / 0xffu;
            t.g = ((p >> 8) & 0xffu) * t.a / 0xffu;
            t.b = ((p >> 16) & 0xffu) * t.a / 0xffu;
            if (KIND == IPX_SRC_NRGBA_CROP) { t.r = (t.r >> 8) * 0x101u; t.g = (t.g >> 8) * 0x101u; t.b = (t.b >> 8) * 0x101u; }
        } else {
            const uint32_t al = p >> 24;
            uint32_t r = p & 0xffu, g = (p >> 8) & 0xffu, b = (p >> 16) & 0xffu;
            if (KIND == IPX_SRC_RGBA_CROP) { r = min(r, al); g = min(g, al); b = min(b, al); }
            t.r = r * 0x101u; t.g = g * 0x101u; t.b = b * 0x101u; t.a = al * 0x101u;
        }
    }
    return t;
}

__device__ __forceinline__ uint32_t ks_ftou(double f)   // impl.go ftou
{
    const int i = (int)(0xffff * f + 0.5);
    return i > 0xffff ? 0xffffu : (i > 0 ? (uint32_t)i : 0u);
}

// one destination pixel (dx, dy relative to dr.Min) of the frame `a` points at
template <int KIND>
__device__ __forceinline__ void ks_pixel(const KsGenArgs &a, int dx, int dy)
{
    const int xlo = a.ax.lo[dx], xn = a.ax.cnt[dx], ylo = a.ay.lo[dy], yn = a.ay.cnt[dy];
    const double *wx = a.ax.w + (size_t)dx * a.ax.ntap, *wy = a.ay.w + (size_t)dy * a.ay.ntap;
    const double xs = a.ax.itwffff[dx], ys = a.ay.itw[dy];
    constexpr bool alpha_one = KIND == IPX_SRC_YCBCR;   // scaleX_YCbCr4xx (and scaleX_Gray) store a literal 1 as tmp alpha

    double qr = 0, qg = 0, qb = 0, qa = 0;
    for (int j = 0; j < yn; j++) {
        double pr = 0, pg = 0, pb = 0, pa = 0;
        for (int i = 0; i < xn; i++) {
            const Tap4 t = ks_tap<KIND>(a, a.sr_x0 + xlo + i, a.sr_y0 + ylo + j);
            const double w = wx[i];
            pr += (double)t.r * w;
            pg += (double)t.g * w;
            pb += (double)t.b * w;
            if (!alpha_one) pa += (double)t.a * w;
        }
        const double w = wy[j];
        qr += (pr * xs) * w;
        qg += (pg * xs) * w;
        qb += (pb * xs) * w;
        qa += (alpha_one ? 1.0 : pa * xs) * w;
    }
    if (qr > qa) qr = qa;
    if (qg > qa) qg = qa;
    if (qb > qa) qb = qa;
    const uint32_t pr0 = ks_ftou(qr * ys), pg0 = ks_ftou(qg * ys), pb0 = ks_ftou(qb * ys), pa0 = ks_ftou(qa * ys);

    uint32_t *d = (uint32_t *)(a.dst + (size_t)(a.dr_y0 + dy) * a.dstride + (size_t)(a.dr_x0 + dx) * 4);
    int op = a.op;
    if (op == IPX_OP_OVER && a.opaque_flag && *a.opaque_flag) op = IPX_OP_SRC;  // draw/scale.go opaque()
    *d = op == IPX_OP_SRC ? pack_src(pr0, pg0, pb0, pa0) : blend_over(*d, pr0, pg0, pb0, pa0);   // scaleY_RGBA_Src / _Over
}

template <int KIND>
__global__ __launch_bounds__(256) void ks_generic_kernel(KsGenArgs a)
{
    a.dst += blockIdx.z * a.dst_fs;      // frame of a batch (all zero for a single frame)
    a.src += blockIdx.z * a.src_fs;
    if (KIND == IPX_SRC_YCBCR || KIND == IPX_SRC_YCBCR_CROP) { a.cb += blockIdx.z * a.c_fs; a.cr += blockIdx.z * a.c_fs; }
    const int dx = a.adr_x0 + (int)(blockIdx.x * 64 + threadIdx.x);
    const int dy = a.adr_y0 + (int)(blockIdx.y * 4 + threadIdx.y);
    if (dx >= a.adr_x1 || dy >= a.adr_y1) return;
    ks_pixel<KIND>(a, dx, dy);
}

// The exact pass behind the one-pass kernel's float pass (ipx_ks_fused.hip): per frame (blockIdx.y) a list of the pixels whose byte the
// float sums could not decide; the ones of output `k` are recomputed here, in float64 and in the reference's order.
// R lanes per pixel (R = 8, 16 or 64, at least the vertical tap count where that is at most 64): lane j of a pixel's group walks source
// row j from left to right -- scaleX's sum for tmp[row][dx], in source-column order -- and the group's values are then added in
// source-row order, one after the other as scaleY does, in every lane of the group (the first one stores).  A thread per pixel would
// walk nx * ny taps one after the other: 1936 dependent steps for an 8K frame's thumbnail, 0.7 ms for a batch of two frames.
template <int KIND>
__global__ __launch_bounds__(256) void ks_fix_kernel(KsGenArgs a, const uint2 *list, size_t list_stride, const int *count, int count_stride, int cap, int R)
{
    const int frame = (int)blockIdx.y, n = min(count[(size_t)frame * count_stride], cap);
    list += (size_t)frame * list_stride;
    a.dst += frame * a.dst_fs;
    a.src += frame * a.src_fs;
    if (KIND == IPX_SRC_YCBCR || KIND == IPX_SRC_YCBCR_CROP) { a.cb += frame * a.c_fs; a.cr += frame * a.c_fs; }
    const int lane = (int)threadIdx.x & 63, wv = (int)threadIdx.x >> 6;
    const int per_wave = 64 / R, sub = lane / R, sl = lane - sub * R, per_block = ((int)blockDim.x >> 6) * per_wave;
    constexpr bool alpha_one = KIND == IPX_SRC_YCBCR;
    for (int i0 = (int)blockIdx.x * per_block; i0 < n; i0 += (int)gridDim.x * per_block) {
        const int i = i0 + wv * per_wave + sub;
        bool live = i < n;
        const uint2 e = live ? list[i] : make_uint2(0u, 0u);
        const int dy = (int)e.x, dx = (int)e.y;
        live = live && dx < a.adr_x1 && dy < a.adr_y1;
        const int xlo = live ? a.ax.lo[dx] : 0, xn = live ? a.ax.cnt[dx] : 0, ylo = live ? a.ay.lo[dy] : 0, yn = live ? a.ay.cnt[dy] : 0;
        const double *wx = a.ax.w + (size_t)(live ? dx : 0) * a.ax.ntap, *wy = a.ay.w + (size_t)(live ? dy : 0) * a.ay.ntap;
        const double xs = live ? a.ax.itwffff[dx] : 0.0;
        double qr = 0, qg = 0, qb = 0, qa = 0;
        for (int jb = 0; __any(jb < yn); jb += R) {                      // (wave-uniform) R source rows at a time, top to bottom
            const int j = jb + sl;
            double pr = 0, pg = 0, pb = 0, pa = 0;
            if (j < yn)
#pragma unroll 4
                for (int t = 0; t < xn; t++) {                            // (the taps' loads do not wait for one another)
                    const Tap4 tp = ks_tap<KIND>(a, a.sr_x0 + xlo + t, a.sr_y0 + ylo + j);
                    const double w = wx[t];
                    pr += (double)tp.r * w;
                    pg += (double)tp.g * w;
                    pb += (double)tp.b * w;
                    if (!alpha_one) pa += (double)tp.a * w;
                }
            const double tr = pr * xs, tg = pg * xs, tb = pb * xs, ta = alpha_one ? 1.0 : pa * xs;
            const double wl = j < yn ? wy[j] : 0.0;                       // the row's vertical weight travels with its sums (a load per
                                                                          // step of the loop below was a memory round trip per row)
            for (int jj = 0; jj < R; jj++) {                              // every lane of the group adds the group's rows in order
                const int from = sub * R + jj;
                const double vr = __shfl(tr, from), vg = __shfl(tg, from), vb = __shfl(tb, from), va = __shfl(ta, from);
                const double w = __shfl(wl, from);
                if (jb + jj < yn) {
                    qr += vr * w;
                    qg += vg * w;
                    qb += vb * w;
                    qa += va * w;
                }
            }
        }
        if (live && sl == 0) {
            const double ys = a.ay.itw[dy];
            if (qr > qa) qr = qa;
            if (qg > qa) qg = qa;
            if (qb > qa) qb = qa;
            const uint32_t pr0 = ks_ftou(qr * ys), pg0 = ks_ftou(qg * ys), pb0 = ks_ftou(qb * ys), pa0 = ks_ftou(qa * ys);
            *(uint32_t *)(a.dst + (size_t)(a.dr_y0 + dy) * a.dstride + (size_t)(a.dr_x0 + dx) * 4) = pack_src(pr0, pg0, pb0, pa0);
        }
    }
}

}  // namespace

hipError_t launch_ks_generic(const KsGenArgs &a, hipStream_t s)
{
    const int w = a.adr_x1 - a.adr_x0, h = a.adr_y1 - a.adr_y0;
    if (w <= 0 || h <= 0) return hipSuccess;
    dim3 block(64, 4), grid((w + 63) / 64, (h + 3) / 4, a.nframes > 0 ? a.nframes : 1);
    switch (a.kind) {
    case IPX_SRC_NRGBA: hipLaunchKernelGGL(ks_generic_kernel<IPX_SRC_NRGBA>, grid, block, 0, s, a); break;
    case IPX_SRC_YCBCR: hipLaunchKernelGGL(ks_generic_kernel<IPX_SRC_YCBCR>, grid, block, 0, s, a); break;
    case IPX_SRC_TAP64: hipLaunchKernelGGL(ks_generic_kernel<IPX_SRC_TAP64>, grid, block, 0, s, a); break;
    case IPX_SRC_RGBA_CROP: hipLaunchKernelGGL(ks_generic_kernel<IPX_SRC_RGBA_CROP>, grid, block, 0, s, a); break;
    case IPX_SRC_NRGBA_CROP: hipLaunchKernelGGL(ks_generic_kernel<IPX_SRC_NRGBA_CROP>, grid, block, 0, s, a); break;
    case IPX_SRC_YCBCR_CROP: hipLaunchKernelGGL(ks_generic_kernel<IPX_SRC_YCBCR_CROP>, grid, block, 0, s, a); break;
    case IPX_SRC_TAP64_CROP: hipLaunchKernelGGL(ks_generic_kernel<IPX_SRC_TAP64_CROP>, grid, block, 0, s, a); break;
    default: hipLaunchKernelGGL(ks_generic_kernel<IPX_SRC_RGBA>, grid, block, 0, s, a); break;
    }
    return hipGetLastError();
}

hipError_t launch_ks_fix(const KsGenArgs &a, const uint2 *list, size_t list_stride, const int *count, int count_stride, int cap, hipStream_t s)
{
    if (a.nframes <= 0 || cap <= 0) return hipSuccess;
    const int R = a.ay.ntap <= 8 ? 8 : a.ay.ntap <= 16 ? 16 : 64;
    // about two thousand blocks in all: two per frame for a batch of 1024 (a photograph leaves a few hundred pixels of an output on
    // its frame's list: a block takes 32 or 16 per pass), more per frame when the batch is small.  Measured per 1024 x 1080p
    // (tools/fix_grid.sh): 8192 blocks 113 + 96 us (most of them find nothing to do), 2048 105 + 63, 1024-thread blocks 108 - 250.
    constexpr int threads = 256, budget = 2048;
    const int per_block = (threads / 64) * (64 / R), most = (cap + per_block - 1) / per_block;
    dim3 block(threads), grid(std::max(1, std::min(most, std::max(1, budget / a.nframes))), a.nframes);
    switch (a.kind) {
    case IPX_SRC_YCBCR: hipLaunchKernelGGL(ks_fix_kernel<IPX_SRC_YCBCR>, grid, block, 0, s, a, list, list_stride, count, count_stride, cap, R); break;
    case IPX_SRC_YCBCR_CROP: hipLaunchKernelGGL(ks_fix_kernel<IPX_SRC_YCBCR_CROP>, grid, block, 0, s, a, list, list_stride, count, count_stride, cap, R); break;
    case IPX_SRC_RGBA_CROP: hipLaunchKernelGGL(ks_fix_kernel<IPX_SRC_RGBA_CROP>, grid, block, 0, s, a, list, list_stride, count, count_stride, cap, R); break;
    case IPX_SRC_RGBA: hipLaunchKernelGGL(ks_fix_kernel<IPX_SRC_RGBA>, grid, block, 0, s, a, list, list_stride, count, count_stride, cap, R); break;
    case IPX_SRC_NRGBA: hipLaunchKernelGGL(ks_fix_kernel<IPX_SRC_NRGBA>, grid, block, 0, s, a, list, list_stride, count, count_stride, cap, R); break;
    case IPX_SRC_NRGBA_CROP: hipLaunchKernelGGL(ks_fix_kernel<IPX_SRC_NRGBA_CROP>, grid, block, 0, s, a, list, list_stride, count, count_stride, cap, R); break;
    case IPX_SRC_TAP64: hipLaunchKernelGGL(ks_fix_kernel<IPX_SRC_TAP64>, grid, block, 0, s, a, list, list_stride, count, count_stride, cap, R); break;
    case IPX_SRC_TAP64_CROP: hipLaunchKernelGGL(ks_fix_kernel<IPX_SRC_TAP64_CROP>, grid, block, 0, s, a, list, list_stride, count, count_stride, cap, R); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace ipx

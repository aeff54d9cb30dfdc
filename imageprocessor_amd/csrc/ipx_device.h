// ipx_device.h -- device-side arithmetic shared by the kernels.  Include after
// `#pragma clang fp contract(off)`: the float64 lerp must round every product before the add.
#pragma once

#include "ipx_internal.h"

namespace ipx {
namespace {

constexpr uint32_t kM = 0xffffu;

// v * 0x101 for byte `C` of a packed RGBA dword, as float64
template <int C>
__device__ __forceinline__ double widen(uint32_t px)
{
    // v_perm_b32: bytes {0, 0, b, b} -> b * 0x101 in one instruction
    constexpr uint32_t sel = 0x0c0c0000u | (uint32_t)C | ((uint32_t)C << 8);
    return (double)__builtin_amdgcn_perm(0u, px, sel);
}

// one output channel: the three lerps of scale_RGBA_RGBA_*, products rounded separately
template <int C>
__device__ __forceinline__ uint32_t lerp_channel(uint32_t p00, uint32_t p10, uint32_t p01,
                                                 uint32_t p11, double xw0, double xw1, double yw0,
                                                 double yw1)
{
    const double s00 = widen<C>(p00), s10 = widen<C>(p10);
    const double s01 = widen<C>(p01), s11 = widen<C>(p11);
    const double top = xw0 * s00 + xw1 * s10;
    const double bot = xw0 * s01 + xw1 * s11;
    const double v = yw0 * top + yw1 * bot;
    return (uint32_t)v;  // truncation, as Go's uint32(float64)
}

__device__ __forceinline__ uint32_t pack_src(uint32_t pr, uint32_t pg, uint32_t pb, uint32_t pa)
{
    // uint8(p >> 8) per channel; p <= 0xffff
    return (pr >> 8) | (pg & 0xff00u) | ((pb & 0xff00u) << 8) | ((pa & 0xff00u) << 16);
}

__device__ __forceinline__ uint32_t blend_over(uint32_t d, uint32_t pr, uint32_t pg, uint32_t pb,
                                               uint32_t pa)
{
    // scale_RGBA_RGBA_Over: dst = uint8((uint32(dst)*pa1/0xffff + p) >> 8), pa1 = (0xffff-pa)*0x101
    const uint32_t pa1 = (kM - pa) * 0x101u;
    const uint32_t r = (((d & 0xffu) * pa1 / kM + pr) >> 8) & 0xffu;
    const uint32_t g = ((((d >> 8) & 0xffu) * pa1 / kM + pg) >> 8) & 0xffu;
    const uint32_t b = ((((d >> 16) & 0xffu) * pa1 / kM + pb) >> 8) & 0xffu;
    const uint32_t a = (((d >> 24) * pa1 / kM + pa) >> 8) & 0xffu;
    return r | (g << 8) | (b << 16) | (a << 24);
}

// ---------------------------------------------------------------------------------------------
// drawGlyphOver for one pixel and one mask value; uint32 arithmetic wraps exactly as in Go
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t glyph_over(uint32_t d, uint32_t mask8, uint32_t sr, uint32_t sg,
                                               uint32_t sb, uint32_t sa)
{
    const uint32_t ma = mask8 | (mask8 << 8);
    const uint32_t a = (kM - (sa * ma / kM)) * 0x101u;
    const uint32_t r = (((d & 0xffu) * a + sr * ma) / kM >> 8) & 0xffu;
    const uint32_t g = ((((d >> 8) & 0xffu) * a + sg * ma) / kM >> 8) & 0xffu;
    const uint32_t b = ((((d >> 16) & 0xffu) * a + sb * ma) / kM >> 8) & 0xffu;
    const uint32_t al = (((d >> 24) * a + sa * ma) / kM >> 8) & 0xffu;
    return r | (g << 8) | (b << 16) | (al << 24);
}

// all glyphs, in string order, on the pixel (x, y)
__device__ __forceinline__ uint32_t glyph_run(uint32_t d, int x, int y, const DevGlyph *__restrict__ gl,
                                              int n, uint32_t sr, uint32_t sg, uint32_t sb,
                                              uint32_t sa)
{
    for (int g = 0; g < n; g++) {
        const DevGlyph G = gl[g];
        if (x >= G.x0 && x < G.x1 && y >= G.y0 && y < G.y1) {
            const uint32_t m = G.mask[(size_t)(y - G.y0) * G.mstride + (x - G.x0)];
            if (m) d = glyph_over(d, m, sr, sg, sb, sa);
        }
    }
    return d;
}


}  // namespace
}  // namespace ipx

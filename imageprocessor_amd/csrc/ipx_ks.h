// ipx_ks.h -- the kernel scaler: x/image/draw's BiLinear, i.e. (&Kernel{1, tent}).Scale (draw/scale.go), as tables and launchers.
//
// resize.go:123 and thumbnail.go:129 call xdraw.BiLinear.Scale.  BiLinear is NOT the 2x2-tap ApproxBiLinear (ablInterpolator) that
// rounds 1 and 2 of this library implemented; it is the tent kernel run through kernelScaler: per axis a distribution of source
// indices over destination indices (newDistrib: support 1, widened by the downscale ratio, weights 1 - t renormalised), a horizontal
// pass into a float64 image (scaleX_<source type>: sum of tap * weight in source order, times 1 / (total * 0xffff)) and a vertical pass
// down its columns (scaleY_RGBA_{Src,Over}: sum of tmp * weight in source-row order, colour clamped to alpha, times 1 / total, ftou's
// int32(0xffff * f + 0.5), >> 8).  Every product is rounded before it is added (GOAMD64=v1 never fuses), so the kernels here do the same
// float64 operations in the same order and are compiled with contraction off.
#pragma once

#include <vector>

#include "ipx_internal.h"

namespace ipx {

// ---- one axis of newDistrib, on the host ------------------------------------------------------------------------------------------
// For the tent the contributing source indices of a destination index are one contiguous range: t = |centre - coord| * argscale grows
// away from the centre, `t >= Support` cuts both ends, and a weight 1 - t is never zero inside (ks_build_axis checks it).
struct KsAxis {
    int dw = 0, sw = 0;
    int ntap = 0;                       // most contributions of any destination index
    std::vector<int32_t> lo, cnt;       // first contributing source index (relative to the source rectangle) and how many
    std::vector<double> w;              // [dw][ntap], in source order, padded with zeros (x + 0 * v == x for finite v >= 0)
    std::vector<double> itw, itwffff;   // 1 / total and that / 0xffff (source.invTotalWeight, invTotalWeightFFFF)
    std::vector<double> ones;           // the sum of 1.0 * w in source order: what scaleY makes of a tmp alpha that scaleX set to 1 (Gray, YCbCr)
};
// false: dw or sw not positive, or an axis the contiguous-range form cannot hold (never for the tent)
bool ks_build_axis(int dw, int sw, KsAxis *out);

// ---- the same axis in HBM ------------------------------------------------------------------------------------------------------------
struct KsAxisDev {
    const int32_t *lo, *cnt;
    const double *w, *itw, *itwffff, *ones;
    int ntap;
};
size_t ks_axis_bytes(const KsAxis &a);
// lays the axis out at host blob `h` (ks_axis_bytes long, 16-byte aligned) so that the device copy at `d` is described by *out
void ks_axis_pack(const KsAxis &a, uint8_t *h, const uint8_t *d, KsAxisDev *out);

// ---- how a source pixel becomes the four 16-bit values scaleX weights (ScaleArgs::kind of the generic kernel, KsFusedArgs::mode) ------
// mode 0 kinds: what x/image's scaleX_<type> reads from the source image itself
//   IPX_SRC_RGBA   scaleX_RGBA      c * 0x101
//   IPX_SRC_NRGBA  scaleX_NRGBA     a16 = a * 0x101; c * a16 / 0xff
//   IPX_SRC_YCBCR  scaleX_YCbCr4xx  color.YCbCr.RGBA inlined, clamped to 16 bit; tmp alpha = 1
//   IPX_SRC_TAP64  scaleX_Image     At(x, y).RGBA() as four uint16 (deep sources after deep_expand_kernel)
// mode 1 kinds: cropAndResize (thumbnail.go:128-131) first scales the crop rectangle into an RGBA frame of the SAME size -- one tap of
// weight 1 per axis, so every pixel goes through scaleX / scaleY alone: colour clamped to alpha, ftou, >> 8 -- and resizeImage then
// runs scaleX_RGBA on that 8-bit frame.  The tap of the second scale is therefore (top byte of the clamped mode-0 value) * 0x101:
enum {
    IPX_SRC_RGBA_CROP = 8,    // min(c, a) * 0x101 (identity for premultiplied pixels)
    IPX_SRC_NRGBA_CROP = 9,   // (c * a16 / 0xff >> 8) * 0x101, alpha a * 0x101
    IPX_SRC_YCBCR_CROP = 10,  // top byte of the clamped 16-bit conversion (= color.YCbCrToRGB) * 0x101, alpha 0xffff (summed, not 1)
    IPX_SRC_TAP64_CROP = 11   // (min(c, a) >> 8) * 0x101, alpha (a >> 8) * 0x101
};
inline int ks_crop_kind(int kind)
{
    return kind == IPX_SRC_NRGBA ? IPX_SRC_NRGBA_CROP : kind == IPX_SRC_YCBCR ? IPX_SRC_YCBCR_CROP : kind == IPX_SRC_TAP64 ? IPX_SRC_TAP64_CROP
                                                                                                                           : IPX_SRC_RGBA_CROP;
}

// ---- generic launcher: any rectangles, either op, any kind; one thread per destination pixel (ipx_ks_generic.hip) ---------------------
struct KsGenArgs {
    uint8_t *dst; int dstride; size_t dst_fs;
    int dr_x0, dr_y0;
    int adr_x0, adr_y0, adr_x1, adr_y1;   // affected rectangle, relative to dr.Min
    int sr_x0, sr_y0;
    KsAxisDev ax, ay;
    int op;                               // IPX_OP_*
    const int *opaque_flag;               // device int: nonzero when the whole source is opaque (Over only)
    int kind;
    const uint8_t *src; int sstride; size_t src_fs;
    const uint8_t *cb, *cr; int cstride, ratio; size_t c_fs;
    int nframes;
};
hipError_t launch_ks_generic(const KsGenArgs &a, hipStream_t s);

}  // namespace ipx

// =====================================================================================================================================
// The one-pass kernel (ipx_ks_fused.hip): every operator of a batch from ONE read of each source frame (image_processor.go:64-65: all
// operators take the ORIGINAL frame).  A workgroup streams a strip of source columns from top to bottom in groups of B rows through LDS,
// stores the rows to the watermark frame on the way, and keeps the kernel scaler's state in registers: a lane owns destination columns,
// runs scaleX for them on the rows in LDS and feeds the result straight into scaleY's running sums (the float64 image tmp never exists).
// =====================================================================================================================================
namespace ipx {

struct KsStrip { int c0, c1, t0, tw; };   // owned source columns [c0, c1) (watermark stores); the LDS tile holds columns [t0, t0 + tw)
struct KsSeg { int ys, r0, r1; };         // rows [ys, r1) are streamed; rows [r0, r1) are owned (stored to the watermark frame)
// What one source row does to scaleY's running sums of one output.  A destination row dy keeps its sums in accumulator dy % NACC (at
// most NACC destination rows are fed by one source row); w = 0: that accumulator gets nothing from this row; emit >= 0: this was the
// last row of destination row `emit` -- finish it (times itw, clamp to the alpha `ones` for sources whose tmp alpha is 1) and clear.
// wf: the weight of the float pass (ipx_ks_fused.hip, "the float pass"): float(w * invTotalWeight of the destination row).
template <int NACC> struct KsRowT { double w[NACC], itw[NACC], ones[NACC]; int32_t emit[NACC]; float wf[NACC]; };
static_assert(sizeof(KsRowT<2>) == 64 && sizeof(KsRowT<4>) == 128, "row entries are copied dword-wise");

struct KsFusedOut {
    uint8_t *out; size_t frame_stride; int ostride, obytes;   // destination frames (tightly packed rows)
    int dw, dh, sr_x0, sr_y0;
    int kind;                  // tap kind of the source pixels for this output (IPX_SRC_* / *_CROP)
    int mode, aone;            // filled by the launcher from `kind`: the tap mode on this source type's tile; tmp alpha = 1 (Gray, YCbCr)
    int pk;                    // which output of the plan this is (0 resize, 1 thumbnail); the launcher fills everything below from it
    int ntap;                  // horizontal taps per destination column, padded with zero weights
    int waves, cpl;            // wave roles: `waves` waves of 64 lanes, `cpl` columns per lane
    int wcols;                 // columns per tap row of the LDS weight table (most destination columns any strip owns)
    const double *wx;          // [strip][ntap][wcols] horizontal weights
    const float *wxf;          // the float pass: float(w * invTotalWeightFFFF * 0xffff [* 0x101 where the tile holds bytes]), [strip][ntapf][wcols]
    int split, ntapf;          // the float pass: lanes per column (1, or 2: adjacent lanes sum half of a column's taps each -- float sums need
                               // no order -- and add the halves), and the tap rows of wxf (ntap rounded up to a multiple of split, zero padded)
    float feps;                // the float pass: a channel whose T = (value + 0.5) / 256 has a fraction within feps * T of 0 or 1 is not decided
    const double *itwf;        // [dw] invTotalWeightFFFF
    const int32_t *xlo;        // [dw] first tap, relative to the source rectangle
    const int32_t *colb;       // [nstrips + 1] first destination column owned by each strip
    const void *rows;          // KsRowT<NACC> entries, segment after segment, each padded to a multiple of B
    const int32_t *rowoff;     // [nseg] first entry of each segment
};
struct KsFusedArgs {
    const uint8_t *src; size_t src_fs; int sstride, sw, sh;   // pixels, or the luma plane
    int src_kind;              // IPX_SRC_*: how `src` is laid out
    int taps_le_alpha;         // IPX_SRC_TAP64: no colour tap exceeds its alpha tap, so the reference's clamp never fires (the float pass asks)
    const uint8_t *cb, *cr; int cstride, ratio; size_t c_fs;   // IPX_SRC_YCBCR: the chroma planes (cstride 0: a Gray frame)
    uint8_t *wm; size_t wm_fs; int wm_stride;
    int nframes, nstrips, nseg, nthreads, pitch, dbuf;
    const KsStrip *strips; const KsSeg *segs;
    int nout; KsFusedOut o[2];
    int lds_w[2], lds_rows, lds_open;   // byte offsets in LDS: weight tables, staged row entries, the float pass's undecided pixels (the tile is at 0)
    int open_per_wave;         // the float pass: entries of a wave's share of lds_open
    uint8_t wave_role[16];     // per wave: output index << 4 | index among that output's waves; 0xff = no role (stages pixels only).  The roles
                               // are interleaved so that the waves of one output spread over the SIMDs however the hardware deals waves out
    unsigned long long *stamps; // diagnostic build (-DIPX_DIAG=1) only: per-phase cycle sums over all waves, else NULL
    int *redo;                 // speculative kernels (opaque, float): one int per item, 1 = the item met a pixel with alpha != 0xff or
                               // overflowed the frame's list and is to be redone; general kernel: only items with redo[item] != 0 run (NULL: all)
    uint2 *fix; int *fix_count; int fix_cap[2], fix_stride;   // the float pass: per frame and output (of the plan: 0 resize, 1 thumbnail) a
                               // list of (dy, dx) of the pixels it could not decide -- output k's at fix + frame * fix_stride + (k ? fix_cap[0] : 0),
                               // fix_cap[k] entries -- and how many were appended, fix_count[2 * frame + k]
};
// what the exact per-pixel pass after the float pass needs per output: the axes in HBM (NULL list: no float pass)
struct KsFix { uint2 *list = nullptr; int *count = nullptr; int cap[2] = {0, 0}; KsAxisDev ax[2], ay[2]; };
// one segmentation of the frame's rows and the row tables cut for it
struct KsFusedGeom { int nseg = 0; const KsSeg *segs = nullptr; const void *rows[2] = {nullptr, nullptr}; const int32_t *rowoff[2] = {nullptr, nullptr}; };
struct KsFusedPlan {
    bool ok = false;
    int nacc = 2, rows = 4, pitch = 0, nstrips = 0, nthreads = 0, nstg = 0, dbuf = 0;   // dbuf: two tile buffers fit in LDS (one barrier per group)
    int lds_w[2] = {0, 0}, lds_rows = 0, lds_open = 0, lds_bytes = 0;
    // the float pass's own layout: float weight tables are half the size, so two tile buffers fit where the float64 kernels have room for one
    struct Lds { int dbuf = 0, lds_w[2] = {0, 0}, lds_rows = 0, lds_open = 0, open_per_wave = 0, lds_bytes = 0; } fast;
    const KsStrip *strips = nullptr;
    struct Out { int ntap = 0, waves = 0, cpl = 0, wcols = 0; const double *wx = nullptr, *itwf = nullptr; const int32_t *xlo = nullptr, *colb = nullptr;
                 const float *wxf = nullptr; float feps = 0; int split = 1, ntapf = 0; } o[2];
    KsFusedGeom whole, split;   // one segment per frame (large batches) / segments of about kKsSplitRows rows (small ones)
};
constexpr int kKsSplitRows = 96;
constexpr int kKsRows = 4;        // B: source rows per group
constexpr int kKsMaxStage = 3;    // 16-byte chunks a thread stages per group
constexpr int kKsMaxWaves = 12;   // waves per workgroup (768 threads: three waves per SIMD, 168 registers each)
constexpr int kKsMaxThreads = 64 * kKsMaxWaves;
constexpr int kKsMaxCpl = 2;
constexpr int kKsSplitTaps = 16;      // horizontal taps per column from which the float pass may give a column to two lanes
constexpr int kKsOpenPerWave = 128;   // the float pass: undecided pixels a wave collects in LDS before they go to the frame's list (64 where that lets a second tile buffer in; at least one entry per lane)
// cuts the frame into strips and segments, assigns wave roles and lays every table out in `blob` (appended, 16-byte aligned offsets;
// the pointers in *out are OFFSETS into the blob until ks_fused_rebase adds the device address).  sc[k] = nullptr: output absent.
// top_taps: this output's taps on a tile of 16-bit values are top bytes times 0x101 (the crop thumbnail of YCbCr frames): the float
// pass reads the byte and its weights carry the 0x101
struct KsFusedIn { int dw, dh, sr_x0, sr_y0; const KsAxis *hx, *hy; int top_taps = 0; };
bool ks_fused_plan(int sw, int sh, const KsFusedIn *sc0, const KsFusedIn *sc1, int px_bytes, std::vector<uint8_t> *blob, KsFusedPlan *out);
void ks_fused_rebase(KsFusedPlan *p, const uint8_t *dev_blob);
// *matched = false: nothing launched (shape, alignment or kind the kernel is not built for)
// fix: lists for the float pass (RGBA with `redo`, YCbCr, Gray sources), or NULL: float64 throughout
hipError_t launch_ks_fused(const KsFusedPlan &p, KsFusedArgs &a, const KsFix *fix, int cus, hipStream_t s, bool *matched);
// the exact pass over the float pass's lists: every listed pixel of one output (described by `g`, one frame per blockIdx.y) recomputed.
// list / count: the output's list and counter of frame 0; frame f's are list + f * list_stride and count[f * count_stride]
hipError_t launch_ks_fix(const KsGenArgs &g, const uint2 *list, size_t list_stride, const int *count, int count_stride, int cap, hipStream_t s);

}  // namespace ipx

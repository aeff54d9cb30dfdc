// ipx_ks_host.cpp -- newDistrib of x/image/draw (draw/scale.go) for BiLinear = &Kernel{1, func(t) { return 1 - t }}, on the host.
//
// The weights are part of the result's bits, so they are computed here with the reference's own float64 expressions, in its order
// (this file is compiled with -ffp-contract=off like everything else; there is nothing to fuse anyway), and the kernels only read them.
// Reference call sites: resize.go:121-125 (resizeImage), thumbnail.go:128-131 (cropAndResize).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "ipx_ks.h"

namespace ipx {

bool ks_build_axis(int dw, int sw, KsAxis *out)
{
    if (dw <= 0 || sw <= 0) return false;
    const double support = 1.0;                       // BiLinear.Support
    const double scale = (double)sw / (double)dw;
    double half_width = support, kernel_arg_scale = 1.0;
    if (scale > 1) {                                  // shrinking: widen the support so that every source pixel is visited
        half_width *= scale;
        kernel_arg_scale = 1 / scale;
    }
    KsAxis &a = *out;
    a.dw = dw; a.sw = sw; a.ntap = 0;
    a.lo.assign(dw, 0); a.cnt.assign(dw, 0); a.itw.assign(dw, 0.0); a.itwffff.assign(dw, 0.0); a.ones.assign(dw, 0.0);
    std::vector<std::vector<double>> ws(dw);
    for (int x = 0; x < dw; x++) {
        const double center = ((double)x + 0.5) * scale - 0.5;
        int32_t i = (int32_t)std::floor(center - half_width);
        if (i < 0) i = 0;
        int32_t j = (int32_t)std::ceil(center + half_width);
        if (j > sw) {
            j = sw;
            if (j < i) j = i;
        }
        double total = 0.0;
        int first = -1, last = -1;
        for (int32_t coord = i; coord < j; coord++) {
            double t = (center - (double)coord) * kernel_arg_scale;
            if (t < 0) t = -t;
            if (t >= support) continue;
            const double weight = 1 - t;              // BiLinear.At
            if (weight == 0) continue;
            if (first < 0) first = coord;
            else if (coord != last + 1) return false; // a hole in the range: not the tent
            last = coord;
            total += weight;
            ws[x].push_back(weight);
        }
        if (first < 0) return false;                  // no contribution at all: 1 / 0 upstream, cannot happen with sw >= 1
        a.lo[x] = first;
        a.cnt[x] = (int32_t)ws[x].size();
        if (a.cnt[x] > a.ntap) a.ntap = a.cnt[x];
        total = 1 / total;
        a.itw[x] = total;
        a.itwffff[x] = total / 0xffff;
        double ones = 0.0;                            // scaleY's `pa += p[3] * c.weight` over a column of tmp alphas that are all 1
        for (double wv : ws[x]) ones += 1.0 * wv;
        a.ones[x] = ones;
    }
    a.w.assign((size_t)dw * a.ntap, 0.0);
    for (int x = 0; x < dw; x++)
        for (size_t k = 0; k < ws[x].size(); k++) a.w[(size_t)x * a.ntap + k] = ws[x][k];
    return true;
}

static size_t al16(size_t v) { return (v + 15) & ~(size_t)15; }

size_t ks_axis_bytes(const KsAxis &a)
{
    return 2 * al16((size_t)a.dw * sizeof(int32_t)) + al16(a.w.size() * sizeof(double)) + 3 * al16((size_t)a.dw * sizeof(double));
}

void ks_axis_pack(const KsAxis &a, uint8_t *h, const uint8_t *d, KsAxisDev *out)
{
    size_t off = 0;
    auto put = [&](const void *p, size_t bytes) {
        memcpy(h + off, p, bytes);
        const uint8_t *dev = d + off;
        off += al16(bytes);
        return dev;
    };
    out->lo = (const int32_t *)put(a.lo.data(), (size_t)a.dw * sizeof(int32_t));
    out->cnt = (const int32_t *)put(a.cnt.data(), (size_t)a.dw * sizeof(int32_t));
    out->w = (const double *)put(a.w.data(), a.w.size() * sizeof(double));
    out->itw = (const double *)put(a.itw.data(), (size_t)a.dw * sizeof(double));
    out->itwffff = (const double *)put(a.itwffff.data(), (size_t)a.dw * sizeof(double));
    out->ones = (const double *)put(a.ones.data(), (size_t)a.dw * sizeof(double));
    out->ntap = a.ntap;
}

}  // namespace ipx

// =====================================================================================================================================
// Tiling of a frame for the one-pass kernel (ipx_ks_fused.hip) and its tables.
// =====================================================================================================================================
namespace ipx {

// The float pass's margin (ipx_ks_fused.hip, "the float pass"), relative to the value.
// Reference, in real numbers: X_i = sum_j tap_ij a_j (a = horizontal weights over their total), V = sum_i X_i b_i, result
// min(floor(V + 0.5), 0xffff) >> 8 = floor((V + 0.5) / 256) capped at 255; its float64 evaluation is within 1e-9 units of that for any
// tap count below 10^4.  The float pass evaluates the same sums with taps exact in float (at most 16 bits), weights a' = fl(a), b' = fl(b)
// (relative error u = 2^-24 each) and one fused multiply-add per term.  Taps and weights are not negative, so partial sums only grow:
// every rounding of a sum is at most u times the FINAL value of that sum.
//   |X'_i - X_i| <= nx roundings of at most u X'_i + the weights' u X_i                      <= (nx + 1) u X_i (1 + 1e-5)
//   |V' - V|     <= ny roundings of at most u V' + sum b'_i |X'_i - X_i| + the weights' u V  <= (nx + ny + 2) u V (1 + 1e-4)
//   t = fl(V' + 0.5): one more rounding of at most u t                                        <= (nx + ny + 3) u t (1 + 1e-4)
// (sum b_i X_i = V: a bright row's larger error enters with its weight.)  T = t / 256 and its fraction are exact.  So when that
// fraction is at least (nx + ny + 4) u T away from 0 and from 1 -- one spare u T for the factors, the reference's own rounding
// (t >= 0.5, so u t >= 3e-8) and the rounding of the product that forms the margin -- floor(T) IS the reference's byte; a channel
// that is not goes on the frame's list and is recomputed in float64, operation by operation as the reference does it (ks_fix_kernel).
// The margin grows with the value: a dark pixel is decided almost always, a white one has (nx + ny + 4) / 65536 of a byte on either side.
// 0: no float pass for an output with more than 100 taps per pixel (a downscale by 25 and more on both axes; 8K to a 200-pixel thumbnail
// has 88).  The factors above hold far beyond that, but V' <= 65535 (1 + 104 u) keeps T below 256: the kernel needs no clamp.
float ks_float_eps(int nx, int ny) { return nx + ny > 100 ? 0.f : std::nextafter((float)((nx + ny + 4) * (1.0 / 16777216.0)), 1.0f); }

namespace {

size_t blob_put(std::vector<uint8_t> *blob, const void *src, size_t bytes)
{
    const size_t off = (blob->size() + 15) & ~(size_t)15;
    blob->resize(off + bytes);
    if (bytes) memcpy(blob->data() + off, src, bytes);
    return off;
}
template <class T> const T *as_off(size_t off) { return (const T *)(uintptr_t)off; }

// most destination indices one source index feeds
int axis_max_active(const KsAxis &a)
{
    int best = 0, first = 0;
    for (int y = 0; y < a.sw; y++) {
        while (first < a.dw && a.lo[first] + a.cnt[first] <= y) first++;
        int n = 0;
        for (int d = first; d < a.dw && a.lo[d] <= y; d++) n += (y < a.lo[d] + a.cnt[d]);
        if (n > best) best = n;
    }
    return best;
}

// rows of one segmentation for one output: entries for source rows [ys, r1) of every segment, each segment padded to a multiple of B
template <int NACC>
void build_rows(const KsFusedIn &o, const std::vector<KsSeg> &segs, int B, std::vector<KsRowT<NACC>> *rows, std::vector<int32_t> *rowoff)
{
    const KsAxis &hy = *o.hy;
    rows->clear(); rowoff->clear();
    int d0 = 0;
    for (const KsSeg &sg : segs) {
        rowoff->push_back((int32_t)rows->size());
        const int n = ((sg.r1 - sg.ys + B - 1) / B) * B;
        std::vector<KsRowT<NACC>> e(n);
        for (auto &r : e)
            for (int p = 0; p < NACC; p++) { r.w[p] = 0; r.itw[p] = 0; r.ones[p] = 0; r.emit[p] = -1; r.wf[p] = 0; }
        // destination rows owned by this segment: their LAST source row lies in [r0, r1)
        while (d0 < o.dh && o.sr_y0 + hy.lo[d0] + hy.cnt[d0] - 1 < sg.r0) d0++;
        for (int d = d0; d < o.dh && o.sr_y0 + hy.lo[d] + hy.cnt[d] - 1 < sg.r1; d++) {
            const int p = d % NACC;
            for (int k = 0; k < hy.cnt[d]; k++) {
                const int y = o.sr_y0 + hy.lo[d] + k - sg.ys;   // >= 0 by the choice of ys
                e[y].w[p] = hy.w[(size_t)d * hy.ntap + k];
                e[y].wf[p] = (float)(hy.w[(size_t)d * hy.ntap + k] * hy.itw[d]);
                if (k == hy.cnt[d] - 1) { e[y].emit[p] = d; e[y].itw[p] = hy.itw[d]; e[y].ones[p] = hy.ones[d]; }
            }
        }
        rows->insert(rows->end(), e.begin(), e.end());
    }
}

void make_segs(int sh, int nseg, const KsFusedIn *const sc[2], std::vector<KsSeg> *segs)
{
    segs->clear();
    for (int i = 0; i < nseg; i++) {
        KsSeg s;
        s.r0 = (int)((long long)sh * i / nseg); s.r1 = (int)((long long)sh * (i + 1) / nseg);
        s.ys = s.r0;
        for (int k = 0; k < 2; k++) {
            if (!sc[k]) continue;
            const KsAxis &hy = *sc[k]->hy;
            for (int d = 0; d < sc[k]->dh; d++) {               // first destination row whose last source row is in the segment
                const int last = sc[k]->sr_y0 + hy.lo[d] + hy.cnt[d] - 1;
                if (last < s.r0) continue;
                if (last < s.r1) s.ys = std::min(s.ys, sc[k]->sr_y0 + hy.lo[d]);
                break;
            }
        }
        segs->push_back(s);
    }
}

}  // namespace

bool ks_fused_plan(int sw, int sh, const KsFusedIn *sc0, const KsFusedIn *sc1, int px_bytes, std::vector<uint8_t> *blob, KsFusedPlan *out)
{
    const KsFusedIn *const sc[2] = {sc0, sc1};
    KsFusedPlan &P = *out;
    P = KsFusedPlan();
    const int B = kKsRows;
    P.rows = B;
    int nacc = 1;
    for (int k = 0; k < 2; k++)
        if (sc[k]) nacc = std::max(nacc, axis_max_active(*sc[k]->hy));
    if (nacc > 4) return false;                                   // a large upscale: per-output kernels
    P.nacc = nacc <= 2 ? 2 : 4;
    const size_t row_bytes = P.nacc == 2 ? sizeof(KsRowT<2>) : sizeof(KsRowT<4>);
    const size_t lds_budget = (size_t)150 << 10;

    // test knobs: IPX_KS_STRIPS = start with that many strips; IPX_KS_SPLIT_ROWS = rows per segment of the split segmentation
    const char *ev = getenv("IPX_KS_STRIPS");
    const int first_strips = ev && atoi(ev) > 0 ? std::min(atoi(ev), 64) : 1;
    ev = getenv("IPX_KS_SPLIT_ROWS");
    // a segment re-stages the rows its first destination rows reach back to (as many as the vertical tap count): keep that a small share
    int ytaps = 1;
    for (int k = 0; k < 2; k++) if (sc[k]) ytaps = std::max(ytaps, sc[k]->hy->ntap);
    const int split_rows = ev && atoi(ev) > 0 ? atoi(ev) : std::max(kKsSplitRows, 6 * ytaps);
    for (int nstrips = first_strips; nstrips <= 64; nstrips++) {
        const int wc = std::max(4, (((sw + nstrips - 1) / nstrips) + 3) & ~3);
        const int ns = (sw + wc - 1) / wc;
        if (ns != nstrips && nstrips > first_strips) continue;    // this count yields the same strips as a smaller one
        std::vector<KsStrip> strips(ns);
        std::vector<int32_t> colb[2];
        int wcols[2] = {0, 0}, twmax = 0;
        for (int c = 0; c < ns; c++) { strips[c].c0 = c * wc; strips[c].c1 = std::min(sw, (c + 1) * wc); strips[c].t0 = strips[c].c0; strips[c].tw = strips[c].c1 - strips[c].c0; }
        for (int k = 0; k < 2; k++) {
            if (!sc[k]) continue;
            const KsAxis &hx = *sc[k]->hx;
            colb[k].assign(ns + 1, 0);
            int d = 0;
            for (int c = 0; c <= ns; c++) {                        // first destination column whose first tap lies in strip c or beyond
                while (c < ns && d < sc[k]->dw && sc[k]->sr_x0 + hx.lo[d] < strips[c].c0) d++;
                colb[k][c] = c == ns ? sc[k]->dw : d;
            }
        }
        // Whole waves: the output with the most columns gets strip boundaries at multiples of 64 columns (a strip of 513 columns costs
        // a ninth wave for one column).  Its columns next to a boundary may then tap outside the strip's own source columns: the tile
        // grows to hold them (t0 below c0), the watermark stores stay with [c0, c1).
        {
            const int pk = sc[0] && (!sc[1] || sc[0]->dw >= sc[1]->dw) ? 0 : 1;
            if (sc[pk] && ns > 1)
                for (int c = 1; c < ns; c++) {
                    int target = (colb[pk][c] + 32) / 64 * 64;
                    target = std::max(target, colb[pk][c - 1]);
                    colb[pk][c] = std::min(target, sc[pk]->dw);
                }
        }
        for (int k = 0; k < 2; k++) {
            if (!sc[k]) continue;
            const KsAxis &hx = *sc[k]->hx;
            for (int c = 0; c < ns; c++) {
                const int n = colb[k][c + 1] - colb[k][c];
                wcols[k] = std::max(wcols[k], n);
                if (n > 0) {
                    const int first = sc[k]->sr_x0 + hx.lo[colb[k][c]], last = sc[k]->sr_x0 + hx.lo[colb[k][c + 1] - 1] + hx.ntap + (hx.ntap >= kKsSplitTaps ? 1 : 0);   // padded taps stay in the tile (one more where the float pass may halve an odd count)
                    const int t1 = std::max(strips[c].t0 + strips[c].tw, last);
                    strips[c].t0 = std::min(strips[c].t0, first & ~3);
                    strips[c].tw = t1 - strips[c].t0;
                }
            }
        }
        for (auto &s : strips) { s.tw = (s.tw + 3) & ~3; twmax = std::max(twmax, s.tw); }
        const int pitch = twmax * px_bytes;

        // wave roles: W_k waves for output k, each lane cpl_k columns; the busiest SIMD (waves dealt round-robin) decides
        int bestW[2] = {0, 0}, bestcpl[2] = {0, 0}, bestS[2] = {1, 1};
        double bestT = 1e300;
        const bool two = sc[0] && sc[1] && wcols[0] > 0 && wcols[1] > 0;
        // An output with many taps per column (a thumbnail from 4K: 22, from 8K: 44) makes one lane's tap loop the workgroup's longest
        // chain while most lanes idle: the float pass may give such a column to TWO adjacent lanes, half of the taps each (S = 2; then
        // one column pair per lane pair: cpl = 1).  The float64 kernels keep one lane per column on the same waves.
        static const int split_knob = [] { const char *e = getenv("IPX_KS_TAPSPLIT"); return e ? atoi(e) : -1; }();   // test knob: 0 never, 1 wherever allowed
        for (int s0 = 1; s0 <= 2; s0++)
        for (int s1 = 1; s1 <= 2; s1++) {
            const int S[2] = {s0, s1};
            bool allowed = true;
            double percol[2] = {0, 0};
            for (int k = 0; k < 2; k++) {
                if (S[k] == 2 && (!sc[k] || sc[k]->hx->ntap < kKsSplitTaps || split_knob == 0)) allowed = false;
                if (S[k] == 1 && sc[k] && sc[k]->hx->ntap >= kKsSplitTaps && split_knob == 1) allowed = false;
                if (sc[k]) percol[k] = 12.0 * ((sc[k]->hx->ntap + S[k] - 1) / S[k]) + 30.0 + (S[k] == 2 ? 8.0 : 0.0);
            }
            if (!allowed) continue;
            for (int nw = 1; nw <= kKsMaxWaves; nw++) {
                for (int w0 = two ? 1 : (sc[0] && wcols[0] > 0 ? nw : 0); w0 <= (two ? nw - 1 : (sc[0] && wcols[0] > 0 ? nw : 0)); w0++) {
                    const int W[2] = {w0, nw - w0};
                    int cpl[2] = {0, 0};
                    bool ok = true;
                    for (int k = 0; k < 2; k++) {
                        if (!sc[k] || wcols[k] <= 0) { ok = ok && W[k] == 0; continue; }
                        if (W[k] <= 0) { ok = false; continue; }
                        cpl[k] = (wcols[k] * S[k] + 64 * W[k] - 1) / (64 * W[k]);
                        if (cpl[k] > (P.nacc == 2 && S[k] == 1 ? kKsMaxCpl : 1)) ok = false;   // four accumulators per column, or two lanes per column: one column per lane
                    }
                    if (!ok) continue;
                    double simd[4] = {0, 0, 0, 0};
                    for (int i = 0; i < nw; i++) { const int k = i < W[0] ? 0 : 1; simd[i & 3] += cpl[k] * percol[k]; }
                    // (a workgroup whose waves do not divide by the four SIMDs loads them unevenly, and two such workgroups on a CU stack
                    // their surplus on the same SIMDs: measured 30 % slower with 6 waves than with 12)
                    const double T = (std::max(std::max(simd[0], simd[1]), std::max(simd[2], simd[3])) + 1e-3 * nw) * (nw % 4 ? 1.25 : 1.0);
                    if (T < bestT) { bestT = T; bestW[0] = W[0]; bestW[1] = W[1]; bestcpl[0] = cpl[0]; bestcpl[1] = cpl[1]; bestS[0] = S[0]; bestS[1] = S[1]; }
                }
            }
        }
        const bool any_cols = (sc[0] && wcols[0] > 0) || (sc[1] && wcols[1] > 0);
        if (any_cols && bestT >= 1e299) continue;                  // too many columns per strip for kKsMaxWaves waves: more strips
        int nthreads = any_cols ? 64 * (bestW[0] + bestW[1]) : 256;   // no scaled output: the watermark copy alone
        nthreads = std::max(nthreads, 256);
        const int chunks = B * (pitch / (4 * px_bytes));          // chunks of four pixels
        if (chunks > kKsMaxStage * nthreads) {
            if (chunks <= kKsMaxStage * kKsMaxThreads) nthreads = ((chunks + kKsMaxStage - 1) / kKsMaxStage + 63) & ~63;
            else continue;
        }
        // two tile buffers if they fit (one barrier per group), else one
        size_t rest = 0;
        int lds_w[2] = {0, 0};
        for (int k = 0; k < 2; k++)
            if (sc[k]) rest += (size_t)sc[k]->hx->ntap * wcols[k] * sizeof(double);
        rest += 2 * 2 * (size_t)B * row_bytes + 64;                // two buffers of row entries for two outputs (+ slack)
        int dbuf = 1;
        if (2 * (size_t)B * pitch + rest > lds_budget) dbuf = 0;
        if ((size_t)(dbuf + 1) * B * pitch + rest > lds_budget) continue;
        size_t lds = (size_t)(dbuf + 1) * B * pitch;
        for (int k = 0; k < 2; k++) {
            lds_w[k] = (int)lds;
            if (sc[k]) lds += (size_t)sc[k]->hx->ntap * wcols[k] * sizeof(double);
        }
        const int lds_rows = (int)lds;
        lds += 2 * 2 * (size_t)B * row_bytes + 64;
        // the float pass's layout: float weights, the waves' lists of undecided pixels behind the row entries; two tile buffers where
        // they fit in the CU's 160 KB with lists of 128 or 64 entries per wave, else one
        KsFusedPlan::Lds F;
        {
            size_t wf = 0;
            for (int k = 0; k < 2; k++) if (sc[k]) wf += ((size_t)(sc[k]->hx->ntap + 1) * wcols[k] * sizeof(float) + 15) & ~(size_t)15;   // (+ 1: a split output's padded tap row)
            const size_t rows_b = 2 * 2 * (size_t)B * row_bytes + 64, cu = ((size_t)160 << 10) - 512;
            F.dbuf = 0; F.open_per_wave = kKsOpenPerWave;
            for (int per : {kKsOpenPerWave, 64})                   // (never under 64: one ballot may add an entry per lane)
                if (2 * (size_t)B * pitch + wf + rows_b + 16 + (size_t)kKsMaxWaves * per * sizeof(uint2) <= cu) { F.dbuf = 1; F.open_per_wave = per; break; }
            size_t at = (size_t)(F.dbuf + 1) * B * pitch;
            for (int k = 0; k < 2; k++) {
                F.lds_w[k] = (int)at;
                if (sc[k]) at += ((size_t)(sc[k]->hx->ntap + 1) * wcols[k] * sizeof(float) + 15) & ~(size_t)15;
            }
            F.lds_rows = (int)at;
            at += rows_b;
            at = (at + 15) & ~(size_t)15;
            F.lds_open = (int)at;
            at += (size_t)kKsMaxWaves * F.open_per_wave * sizeof(uint2);
            F.lds_bytes = (int)at;
            if (at > cu) { F = KsFusedPlan::Lds(); }            // (cannot happen while the float64 layout fits: it is larger)
        }

        // ---- accepted: lay the tables out ----
        P.nstrips = ns; P.pitch = pitch; P.nthreads = nthreads; P.dbuf = dbuf;
        P.nstg = (chunks + nthreads - 1) / nthreads;
        P.lds_w[0] = lds_w[0]; P.lds_w[1] = lds_w[1]; P.lds_rows = lds_rows; P.lds_bytes = (int)lds;
        P.fast = F;
        if (const char *e = getenv("IPX_KS_FAST_DBUF")) if (atoi(e) == 0 && P.fast.dbuf) {   // test knob: the float pass with one tile buffer
            P.fast.dbuf = 0;
        }
        P.strips = as_off<KsStrip>(blob_put(blob, strips.data(), strips.size() * sizeof(KsStrip)));
        for (int k = 0; k < 2; k++) {
            if (!sc[k]) continue;
            const KsAxis &hx = *sc[k]->hx;
            KsFusedPlan::Out &o = P.o[k];
            o.ntap = hx.ntap; o.waves = bestW[k]; o.cpl = bestcpl[k]; o.wcols = wcols[k];
            std::vector<double> wx((size_t)ns * hx.ntap * std::max(1, wcols[k]), 0.0);
            for (int c = 0; c < ns; c++)
                for (int i = 0; i < colb[k][c + 1] - colb[k][c]; i++)
                    for (int t = 0; t < hx.ntap; t++)
                        wx[((size_t)c * hx.ntap + t) * wcols[k] + i] = hx.w[(size_t)(colb[k][c] + i) * hx.ntap + t];
            o.wx = as_off<double>(blob_put(blob, wx.data(), wx.size() * sizeof(double)));
            // the float pass: weights with both normalisations folded in, in the unit of the 16-bit result (a byte tile's taps are bytes)
            o.split = bestS[k]; o.ntapf = (hx.ntap + o.split - 1) / o.split * o.split;
            std::vector<float> wxf((size_t)ns * o.ntapf * std::max(1, wcols[k]), 0.f);
            const double unit = px_bytes == 4 || (px_bytes == 8 && sc[k]->top_taps) ? 65535.0 * 257.0 : 65535.0;
            for (int c = 0; c < ns; c++)
                for (int i = 0; i < colb[k][c + 1] - colb[k][c]; i++)
                    for (int t = 0; t < hx.ntap; t++)
                        wxf[((size_t)c * o.ntapf + t) * wcols[k] + i] = (float)(hx.w[(size_t)(colb[k][c] + i) * hx.ntap + t] * hx.itwffff[colb[k][c] + i] * unit);
            o.wxf = as_off<float>(blob_put(blob, wxf.data(), wxf.size() * sizeof(float)));
            o.feps = ks_float_eps(hx.ntap, sc[k]->hy->ntap);
            o.itwf = as_off<double>(blob_put(blob, hx.itwffff.data(), hx.itwffff.size() * sizeof(double)));
            o.xlo = as_off<int32_t>(blob_put(blob, hx.lo.data(), hx.lo.size() * sizeof(int32_t)));
            o.colb = as_off<int32_t>(blob_put(blob, colb[k].data(), colb[k].size() * sizeof(int32_t)));
        }
        for (int gi = 0; gi < 2; gi++) {
            KsFusedGeom &g = gi ? P.split : P.whole;
            std::vector<KsSeg> segs;
            make_segs(sh, gi ? std::max(1, (sh + split_rows / 2) / split_rows) : 1, sc, &segs);
            g.nseg = (int)segs.size();
            g.segs = as_off<KsSeg>(blob_put(blob, segs.data(), segs.size() * sizeof(KsSeg)));
            for (int k = 0; k < 2; k++) {
                if (!sc[k]) continue;
                std::vector<int32_t> rowoff;
                if (P.nacc == 2) {
                    std::vector<KsRowT<2>> rows;
                    build_rows<2>(*sc[k], segs, B, &rows, &rowoff);
                    g.rows[k] = as_off<void>(blob_put(blob, rows.data(), rows.size() * sizeof(rows[0])));
                } else {
                    std::vector<KsRowT<4>> rows;
                    build_rows<4>(*sc[k], segs, B, &rows, &rowoff);
                    g.rows[k] = as_off<void>(blob_put(blob, rows.data(), rows.size() * sizeof(rows[0])));
                }
                g.rowoff[k] = as_off<int32_t>(blob_put(blob, rowoff.data(), rowoff.size() * sizeof(int32_t)));
            }
        }
        P.ok = true;
        return true;
    }
    return false;
}

void ks_fused_rebase(KsFusedPlan *p, const uint8_t *d)
{
    auto fix = [&](auto *&ptr) { ptr = (std::remove_reference_t<decltype(ptr)>)(const void *)(d + (uintptr_t)ptr); };
    fix(p->strips);
    for (int k = 0; k < 2; k++) {
        if (!p->o[k].wx) continue;                // output absent (offset 0 is the strips table, never an output's)
        fix(p->o[k].wx); fix(p->o[k].itwf); fix(p->o[k].xlo); fix(p->o[k].colb); fix(p->o[k].wxf);
    }
    for (KsFusedGeom *g : {&p->whole, &p->split}) {
        fix(g->segs);
        for (int k = 0; k < 2; k++) {
            if (!g->rows[k] && !g->rowoff[k]) continue;
            fix(g->rows[k]); fix(g->rowoff[k]);
        }
    }
}

}  // namespace ipx

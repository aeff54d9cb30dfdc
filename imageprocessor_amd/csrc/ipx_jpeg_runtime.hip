// ipx_jpeg_runtime.hip -- the ABI entries of the codec legs: jpeg.Encode and image.Decode for JPEG on the GPU, and the legs that
// chain them with the operators (host frames -> streams, planes -> streams, files -> streams).  Kernels: ipx_jpeg.hip,
// ipx_jpeg_entropy.hip, ipx_jpeg_dec.hip, ipx_jpeg_dec_par.hip; host halves: ipx_jpeg_host.cpp, ipx_jpeg_dec_host.cpp.
#include <atomic>
#include <chrono>
#include <functional>
#include <memory>
#include <string>
#include <thread>

#include "ipx_runtime_internal.h"
#include "ipx_threads.h"

// ---- jpeg.Encode: the entries that touch the device (tables / entropy coder: ipx_jpeg_host.cpp) ----------

extern "C" {

static int fdct_rgba8(ipx_ctx *ctx, hipStream_t s, const uint8_t *src, int w, int h, int stride, size_t frame_stride, int n, int quality,
                      int16_t *coefs, const uint32_t *huff, uint32_t *aclen, int16_t *dcq);

int ipx_dev_jpeg_fdct_rgba8(ipx_ctx *ctx, void *stream, const uint8_t *src, int w, int h, int stride, size_t frame_stride,
                            int n, int quality, int16_t *coefs) try
{
    IPX_ENTER(ctx);
    return fdct_rgba8(ctx, stream ? (hipStream_t)stream : ctx->stream, src, w, h, stride, frame_stride, n, quality, coefs, nullptr, nullptr, nullptr);
}
IPX_CATCH_STATUS

}  // extern "C"

// huff / aclen / dcq: all or none -- the transform kernel then also sizes every block's AC symbols and leaves the DC terms in a dense array
static int fdct_rgba8(ipx_ctx *ctx, hipStream_t s, const uint8_t *src, int w, int h, int stride, size_t frame_stride, int n, int quality,
                      int16_t *coefs, const uint32_t *huff, uint32_t *aclen, int16_t *dcq)
{
    (void)ctx;
    if (!src || !coefs || n < 0 || w <= 0 || h <= 0 || (long long)stride < (long long)w * 4) {
        set_error("ipx_dev_jpeg_fdct_rgba8: bad argument");
        return IPX_ERR_INVALID;
    }
    if (w >= 1 << 16 || h >= 1 << 16) { set_error("jpeg: image is too large to encode"); return IPX_ERR_INVALID; }
    if (n == 0) return IPX_OK;
    if (n > 65535) { set_error("ipx_dev_jpeg_fdct_rgba8: at most 65535 frames per call"); return IPX_ERR_UNSUPPORTED; }
    JpegTables t;
    jpeg_tables(quality, &t);
    JpegArgs a;
    a.src = src; a.frame_stride = frame_stride; a.stride = stride; a.w = w; a.h = h;
    a.aligned16 = ((((uintptr_t)src) | (uintptr_t)stride | frame_stride) & 15) == 0;
    a.coefs = coefs; a.mcus_per_frame = ((w + 15) / 16) * ((h + 15) / 16);
    a.huff = huff; a.aclen = aclen; a.dcq = dcq;
    memcpy(a.recip, t.recip, sizeof a.recip);
    memcpy(a.div8, t.div8, sizeof a.div8);
    IPX_HIP(launch_jpeg_fdct(a, n, s));
    return IPX_OK;
}

// host entropy coding of a downloaded coefficient batch (IPX_JPEG_HOST_ENTROPY=1, and the reference point of tools/bench_jpeg.py)
static int jpeg_batch_host_entropy(ipx_ctx *ctx, Lane &lane, const int16_t *dcoefs, int w, int h, int n, int quality,
                                   uint8_t **blob, size_t *offs, size_t *lens)
{
    const size_t per = ipx_jpeg_coef_count(w, h) * sizeof(int16_t);
    hipStream_t s = lane.stream;
    int16_t *host = nullptr;
    IPX_HIP(hipHostMalloc((void **)&host, per * n, hipHostMallocDefault));
    hipError_t e = hipMemcpyAsync(host, dcoefs, per * n, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { (void)hipHostFree(host); set_error("coefficient download failed: %s", hipGetErrorString(e)); return IPX_ERR_HIP; }
    JpegTables t;
    jpeg_tables(quality, &t);
    std::vector<std::vector<uint8_t>> streams(n);
    std::atomic<int> failed{IPX_OK};
    HostPool::instance().parallel_for(n, 16, [&](int i) {
        const int rc = guarded_status([&] { jpeg_write_stream(host + (per / sizeof(int16_t)) * (size_t)i, w, h, t, &streams[i]); }, nullptr);
        if (rc) failed = rc;
    });
    (void)hipHostFree(host);
    if (failed) { set_error("entropy coding on the host failed (out of memory)"); return failed; }
    size_t total = 0;
    for (int i = 0; i < n; i++) { offs[i] = total; lens[i] = streams[i].size(); total += (lens[i] + 15) & ~(size_t)15; }
    uint8_t *b = (uint8_t *)ipx_host_alloc(ctx, total ? total : 1);
    if (!b) return IPX_ERR_NOMEM;
    for (int i = 0; i < n; i++) memcpy(b + offs[i], streams[i].data(), lens[i]);
    *blob = b;
    return IPX_OK;
}


extern "C" {

}  // extern "C"

// n frames in HBM -> streams in one pinned block, everything on stream s; dcoefs = n * ipx_jpeg_coef_count int16 of scratch
static int jpeg_encode_core(ipx_ctx *ctx, hipStream_t s, int16_t *dcoefs, const uint8_t *src, int w, int h, int stride, size_t frame_stride,
                            int n, int quality, uint8_t **blob, size_t *offs, size_t *lens);

extern "C" {

int ipx_jpeg_encode_batch_dev(ipx_ctx *ctx, const uint8_t *src, int w, int h, int stride, size_t frame_stride, int n,
                              int quality, uint8_t **blob, size_t *offs, size_t *lens) try
{
    IPX_ENTER(ctx);
    if (!blob || !offs || !lens || n < 0) { set_error("ipx_jpeg_encode_batch_dev: bad argument"); return IPX_ERR_INVALID; }
    *blob = nullptr;
    if (n == 0) return IPX_OK;
    const size_t per = ipx_jpeg_coef_count(w, h) * sizeof(int16_t);
    LaneLease lane(ctx);
    int rc = lane_reserve(lane.get(), per * n);
    if (rc) return rc;
    int16_t *dcoefs = (int16_t *)lane->dev;
    if (env_int("IPX_JPEG_HOST_ENTROPY", 0)) {
        rc = ipx_dev_jpeg_fdct_rgba8(ctx, lane->stream, src, w, h, stride, frame_stride, n, quality, dcoefs);
        if (rc) return rc;
        return jpeg_batch_host_entropy(ctx, lane.get(), dcoefs, w, h, n, quality, blob, offs, lens);
    }
    return jpeg_encode_core(ctx, lane->stream, dcoefs, src, w, h, stride, frame_stride, n, quality, blob, offs, lens);
}
IPX_CATCH_STATUS

}  // extern "C"

// jpeg.Encode of up to three sets of n frames (the three operators' outputs of one batch) with THREE waits for the device in all: every
// set's transform + symbol sizing, one read-back of the sizes; every set's bit packing + 0xff count, one read-back; every set's byte
// stuffing into one block, one download.  (One set after the other, as rounds 1 and 2 ran it, is nine waits: 5.5 ms for a batch of 8.)
struct JpegEncSet { int16_t *dcoefs; const uint8_t *src; int w, h, stride; size_t frame_stride; size_t *offs, *lens; };   // offs / lens: [n], into *blob
static int jpeg_encode_sets(ipx_ctx *ctx, hipStream_t s, const JpegEncSet *sets, int K, int n, int quality, uint8_t **blob)
{
    *blob = nullptr;
    if (K <= 0 || K > 3 || n <= 0) return IPX_OK;
    // every way out of this function waits for the stream: the queued copies read and write the vectors below
    struct SyncOnExit { hipStream_t s; ~SyncOnExit() { (void)hipStreamSynchronize(s); } } sync_on_exit{s};
    const bool trace = env_int("IPX_DEBUG_J2J", 0) != 0;
    const auto te0 = std::chrono::steady_clock::now();
    auto ems = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - te0).count(); };
    double t_q1 = 0, t_s1 = 0, t_q2 = 0, t_s2 = 0, t_q3 = 0;

    // ---- entropy coding on the GPU: size, scan, place, stuff (ipx_jpeg_entropy.hip) ----
    JpegTables t;
    jpeg_tables(quality, &t);
    uint32_t packed[1024];
    jpeg_huff_packed(packed);
    AsyncFree mem{s, {}};
    uint32_t *d_tab;
    IPX_HIP(mem.get(&d_tab, sizeof packed));
    IPX_HIP(hipMemcpyAsync(d_tab, packed, sizeof packed, hipMemcpyHostToDevice, s));
    struct Dev {
        int nblk = 0, max_chunks = 0;
        std::vector<uint8_t> hdr;
        uint32_t *d_len = nullptr, *d_ubytes = nullptr, *d_ff = nullptr;
        unsigned long long *d_ubase = nullptr, *d_obase = nullptr;
        uint8_t *d_hdr = nullptr;
    } dv[3];
    uint32_t *d_tot, *d_fftot;                    // [K][n]
    IPX_HIP(mem.get(&d_tot, (size_t)K * n * 4));
    IPX_HIP(mem.get(&d_fftot, (size_t)K * n * 4));
    if (trace) fprintf(stderr, "[ipx]   tables queued at %.2f ms\n", ems());
    for (int k = 0; k < K; k++) {
        const JpegEncSet &o = sets[k];
        Dev &d = dv[k];
        d.nblk = (int)(ipx_jpeg_coef_count(o.w, o.h) * sizeof(int16_t) / 128);
        jpeg_write_header(o.w, o.h, t, &d.hdr);
        IPX_HIP(mem.get(&d.d_len, (size_t)n * d.nblk * 4));
        IPX_HIP(mem.get(&d.d_ubytes, (size_t)n * 4));
        IPX_HIP(mem.get(&d.d_ubase, (size_t)n * 8));
        IPX_HIP(mem.get(&d.d_obase, (size_t)n * 8));
        IPX_HIP(mem.get(&d.d_hdr, d.hdr.size()));
        IPX_HIP(hipMemcpyAsync(d.d_hdr, d.hdr.data(), d.hdr.size(), hipMemcpyHostToDevice, s));
        if (env_int("IPX_JPEG_FUSED_LEN", 1)) {
            // the transform kernel sizes the AC symbols of every block while it has the block in LDS; a small kernel adds the DC symbols
            int16_t *d_dcq;
            IPX_HIP(mem.get(&d_dcq, (size_t)n * d.nblk * 2));
            int rc = fdct_rgba8(ctx, s, o.src, o.w, o.h, o.stride, o.frame_stride, n, quality, o.dcoefs, d_tab, d.d_len, d_dcq);
            if (rc) return rc;
            IPX_HIP(launch_jpeg_dclen(d_dcq, d.nblk, n, d_tab, d.d_len, s));
        } else {
            int rc = fdct_rgba8(ctx, s, o.src, o.w, o.h, o.stride, o.frame_stride, n, quality, o.dcoefs, nullptr, nullptr, nullptr);
            if (rc) return rc;
            IPX_HIP(launch_jpeg_len(o.dcoefs, d.nblk, n, d_tab, d.d_len, s));     // the earlier separate pass over the coefficients
        }
        IPX_HIP(launch_scan(d.d_len, d.nblk, n, d_tot + (size_t)k * n, s));
        if (trace) fprintf(stderr, "[ipx]   set %d (%dx%d) transform and sizes queued at %.2f ms\n", k, o.w, o.h, ems());
    }
    std::vector<uint32_t> tot((size_t)K * n), ubytes((size_t)K * n), ff((size_t)K * n);
    IPX_HIP(hipMemcpyAsync(tot.data(), d_tot, (size_t)K * n * 4, hipMemcpyDeviceToHost, s));
    t_q1 = ems();
    IPX_HIP(hipStreamSynchronize(s));
    t_s1 = ems();
    std::vector<unsigned long long> ubase((size_t)K * n), obase((size_t)K * n);
    unsigned long long utotal = 0;
    const int chunk = jpeg_chunk_bytes();
    size_t ff_words = 0;
    for (int k = 0; k < K; k++) {
        uint32_t umax = 0;
        for (int i = 0; i < n; i++) {
            const size_t j = (size_t)k * n + i;
            ubytes[j] = (tot[j] + 7) / 8;
            ubase[j] = utotal;
            utotal += align256((size_t)ubytes[j] + 8);
            umax = std::max(umax, ubytes[j]);
        }
        dv[k].max_chunks = (int)((umax + chunk - 1) / chunk);
        ff_words += (size_t)n * dv[k].max_chunks;
    }
    uint8_t *d_ustream = nullptr, *d_ostream = nullptr;
    uint32_t *d_ffall = nullptr;
    IPX_HIP(mem.get(&d_ustream, (size_t)utotal));
    IPX_HIP(mem.get(&d_ffall, std::max<size_t>(ff_words, 1) * 4));
    IPX_HIP(hipMemsetAsync(d_ustream, 0, (size_t)utotal, s));
    {
        size_t at = 0;
        for (int k = 0; k < K; k++) {
            Dev &d = dv[k];
            d.d_ff = d_ffall + at;
            at += (size_t)n * d.max_chunks;
            IPX_HIP(hipMemcpyAsync(d.d_ubase, ubase.data() + (size_t)k * n, (size_t)n * 8, hipMemcpyHostToDevice, s));
            IPX_HIP(hipMemcpyAsync(d.d_ubytes, ubytes.data() + (size_t)k * n, (size_t)n * 4, hipMemcpyHostToDevice, s));
            IPX_HIP(launch_jpeg_bits(sets[k].dcoefs, d.nblk, n, d_tab, d.d_len, d_tot + (size_t)k * n, d.d_ubase, d_ustream, s));
            IPX_HIP(launch_jpeg_ffcount(d_ustream, d.d_ubase, d.d_ubytes, d.max_chunks, n, d.d_ff, s));
            IPX_HIP(launch_scan(d.d_ff, d.max_chunks, n, d_fftot + (size_t)k * n, s));
        }
    }
    IPX_HIP(hipMemcpyAsync(ff.data(), d_fftot, (size_t)K * n * 4, hipMemcpyDeviceToHost, s));
    t_q2 = ems();
    IPX_HIP(hipStreamSynchronize(s));
    t_s2 = ems();
    unsigned long long ototal = 0;
    for (int k = 0; k < K; k++)
        for (int i = 0; i < n; i++) {
            const size_t j = (size_t)k * n + i;
            sets[k].lens[i] = dv[k].hdr.size() + ubytes[j] + ff[j] + 2;
            obase[j] = ototal;
            sets[k].offs[i] = (size_t)ototal;
            ototal += (sets[k].lens[i] + 15) & ~(size_t)15;
        }
    IPX_HIP(mem.get(&d_ostream, (size_t)ototal));
    for (int k = 0; k < K; k++) {
        Dev &d = dv[k];
        IPX_HIP(hipMemcpyAsync(d.d_obase, obase.data() + (size_t)k * n, (size_t)n * 8, hipMemcpyHostToDevice, s));
        IPX_HIP(launch_jpeg_stuff(d_ustream, d.d_ubase, d.d_ubytes, d.max_chunks, n, d.d_ff, d.d_hdr, (int)d.hdr.size(), d.d_obase, d_ostream, s));
    }
    uint8_t *host = (uint8_t *)ipx_host_alloc(ctx, (size_t)ototal ? (size_t)ototal : 1);   // pinned: the download runs at link speed
    if (!host) { (void)hipStreamSynchronize(s); return IPX_ERR_NOMEM; }
    hipError_t e = hipMemcpyAsync(host, d_ostream, (size_t)ototal, hipMemcpyDeviceToHost, s);
    t_q3 = ems();
    { const hipError_t e2 = hipStreamSynchronize(s); if (e == hipSuccess) e = e2; }          // (ubase / obase are read by the queued copies until here)
    if (trace)
        fprintf(stderr, "[ipx] encode of %d sets x %d frames: sized (queued %.2f ms, done %.2f), packed (queued %.2f, done %.2f), streams of %.1f MB queued %.2f, down at %.2f\n",
                K, n, t_q1, t_s1, t_q2, t_s2, (double)ototal / 1e6, t_q3, ems());
    if (e != hipSuccess) { (void)ipx_host_free(ctx, host); set_error("stream download failed: %s", hipGetErrorString(e)); return IPX_ERR_HIP; }
    *blob = host;
    return IPX_OK;
}

static int jpeg_encode_core(ipx_ctx *ctx, hipStream_t s, int16_t *dcoefs, const uint8_t *src, int w, int h, int stride, size_t frame_stride,
                            int n, int quality, uint8_t **blob, size_t *offs, size_t *lens)
{
    const JpegEncSet one{dcoefs, src, w, h, stride, frame_stride, offs, lens};
    return jpeg_encode_sets(ctx, s, &one, 1, n, quality, blob);
}

// ---- host frames in, JPEG streams out: the worker's whole GPU leg ----------------------------------------
struct ipx_jpeg_result { std::vector<uint8_t *> blobs; };

extern "C" {

void ipx_jpeg_result_free(ipx_ctx *ctx, ipx_jpeg_result *r)
{
    if (!r) return;
    for (uint8_t *b : r->blobs) (void)ipx_host_free(ctx, b);
    delete r;
}

}  // extern "C"

// one implementation for both source kinds: ysrc == nullptr -> RGBA frames at src
static int run_host_jpeg_impl(ipx_ctx *ctx, const ipx_plan *pl, int n, const uint8_t *src, int sstride, size_t src_frame_stride,
                              const ipx_ycbcr_batch *ysrc, int quality, ipx_bytes *resize_out, ipx_bytes *thumb_out, ipx_bytes *wm_out,
                              ipx_jpeg_result **result)
{
    *result = nullptr;
    if (n == 0) return IPX_OK;
    const int sw = pl->p.sw, sh = pl->p.sh;
    const int cw = ysrc ? ((ysrc->ratio == IPX_YCBCR_422 || ysrc->ratio == IPX_YCBCR_420) ? (sw + 1) / 2 : sw) : 0;
    const int ch = ysrc ? ((ysrc->ratio == IPX_YCBCR_420 || ysrc->ratio == IPX_YCBCR_440) ? (sh + 1) / 2 : sh) : 0;
    const size_t yb = ysrc ? align256((size_t)sw * sh) : 0, cbb = ysrc ? align256((size_t)cw * ch) : 0;
    const size_t fsrc = ysrc ? yb + 2 * cbb : align256((size_t)sw * sh * 4);
    const size_t fres = resize_out ? align256(pl->info.resize_bytes) : 0, fth = thumb_out ? align256(pl->info.thumb_bytes) : 0;
    const size_t fwm = wm_out ? align256(pl->info.wm_bytes) : 0;
    const size_t cres = fres ? ipx_jpeg_coef_count(pl->info.resize_w, pl->info.resize_h) * 2 : 0;
    const size_t cth = fth ? ipx_jpeg_coef_count(pl->info.thumb_w, pl->info.thumb_h) * 2 : 0;
    const size_t cwm = fwm ? ipx_jpeg_coef_count(sw, sh) * 2 : 0;
    const size_t ccoef[3] = {align256(cres), align256(cth), align256(cwm)};     // a coefficient buffer per output (jpeg_encode_sets)
    const size_t per_frame = fsrc + fres + fth + fwm + ccoef[0] + ccoef[1] + ccoef[2];
    // every lane runs its own host thread: upload, operators, the three encodes together (two small read-backs) --
    // the threads block independently, so copies and kernels of different chunks overlap
    std::vector<Lane *> lanes;
    {
        std::unique_lock<std::mutex> lk(ctx->mu);
        ctx->cv.wait(lk, [&] { for (auto &l : ctx->lanes) if (l.busy) return false; return true; });
        for (auto &l : ctx->lanes) { l.busy = true; lanes.push_back(&l); }
    }
    const int nl = (int)lanes.size();
    int chunk = std::max(1, (n + 2 * nl - 1) / (2 * nl));
    chunk = (int)std::min<size_t>((size_t)chunk, std::max<size_t>(1, ctx->lane_bytes / per_frame));
    chunk = std::max(1, std::min(chunk, env_int("IPX_HOST_CHUNK_JPEG", 32)));
    std::unique_ptr<ipx_jpeg_result> res(new ipx_jpeg_result);
    std::mutex res_mu;
    std::atomic<int> next{0};
    std::atomic<int> status{IPX_OK};
    std::string err_text;
    const int nchunks = (n + chunk - 1) / chunk;
    auto worker = [&](Lane *l) {
        if (hipSetDevice(ctx->device) != hipSuccess) { status = IPX_ERR_HIP; return; }
        int rc = lane_reserve(*l, per_frame * chunk + 256);
        std::vector<size_t> offs(3 * (size_t)chunk), lens(3 * (size_t)chunk);
        for (int c = next.fetch_add(1); !rc && c < nchunks && status == IPX_OK; c = next.fetch_add(1)) {
            const int i0 = c * chunk, m = std::min(chunk, n - i0);
            uint8_t *dsrc = (uint8_t *)(((uintptr_t)l->dev + 255) & ~(uintptr_t)255);
            uint8_t *dres = fres ? dsrc + fsrc * chunk : nullptr;
            uint8_t *dth = fth ? dsrc + (fsrc + fres) * chunk : nullptr;
            uint8_t *dwm = fwm ? dsrc + (fsrc + fres + fth) * chunk : nullptr;
            uint8_t *cbase = dsrc + (fsrc + fres + fth + fwm) * chunk;
            hipError_t e = hipSuccess;
            if (ysrc) {
                // planes of the chunk: [m x Y][m x Cb][m x Cr]
                uint8_t *dy = dsrc, *dcb = dy + yb * chunk, *dcr = dcb + cbb * chunk;
                auto up = [&](uint8_t *d, size_t dfs, int w, int h, const uint8_t *hsrc, int hstride, size_t hfs) {
                    if (hstride == w && hfs == dfs) return hipMemcpyAsync(d, hsrc + hfs * i0, dfs * m, hipMemcpyHostToDevice, l->stream);
                    hipError_t r = hipSuccess;
                    for (int i = 0; i < m && r == hipSuccess; i++)
                        r = hipMemcpy2DAsync(d + dfs * i, w, hsrc + hfs * (size_t)(i0 + i), hstride, w, h, hipMemcpyHostToDevice, l->stream);
                    return r;
                };
                e = up(dy, yb, sw, sh, ysrc->y, ysrc->ystride, ysrc->y_frame_stride);
                if (e == hipSuccess) e = up(dcb, cbb, cw, ch, ysrc->cb, ysrc->cstride, ysrc->c_frame_stride);
                if (e == hipSuccess) e = up(dcr, cbb, cw, ch, ysrc->cr, ysrc->cstride, ysrc->c_frame_stride);
                if (e != hipSuccess) { set_error("upload failed: %s", hipGetErrorString(e)); rc = IPX_ERR_HIP; break; }
                ipx_ycbcr_batch d;
                d.y = dy; d.cb = dcb; d.cr = dcr; d.ystride = sw; d.cstride = cw; d.y_frame_stride = yb; d.c_frame_stride = cbb;
                d.ratio = ysrc->ratio;
                rc = ipx_plan_run_dev_ycbcr(ctx, l->stream, pl, m, &d, dres, fres, dth, fth, dwm, fwm);
            } else {
                if (sstride == sw * 4 && src_frame_stride == fsrc)
                    e = hipMemcpyAsync(dsrc, src + (size_t)i0 * src_frame_stride, fsrc * m, hipMemcpyHostToDevice, l->stream);
                else
                    for (int i = 0; i < m && e == hipSuccess; i++)
                        e = hipMemcpy2DAsync(dsrc + fsrc * i, (size_t)sw * 4, src + (size_t)(i0 + i) * src_frame_stride, sstride,
                                             (size_t)sw * 4, sh, hipMemcpyHostToDevice, l->stream);
                if (e != hipSuccess) { set_error("upload failed: %s", hipGetErrorString(e)); rc = IPX_ERR_HIP; break; }
                rc = ipx_plan_run_dev(ctx, l->stream, pl, m, dsrc, sw * 4, fsrc, dres, fres, dth, fth, dwm, fwm);
            }
            struct Out { uint8_t *dev; size_t fs; int w, h; ipx_bytes *dst; };
            const Out outs[3] = {{dres, fres, pl->info.resize_w, pl->info.resize_h, resize_out},
                                 {dth, fth, pl->info.thumb_w, pl->info.thumb_h, thumb_out},
                                 {dwm, fwm, sw, sh, wm_out}};
            if (rc) break;
            JpegEncSet sets[3];
            const Out *who[3];
            int K = 0;
            size_t cat = 0;
            for (int k = 0; k < 3; k++) {
                const Out &o = outs[k];
                if (o.dev && o.w > 0 && o.h > 0) {
                    sets[K] = JpegEncSet{(int16_t *)(cbase + cat * chunk), o.dev, o.w, o.h, o.w * 4, o.fs, offs.data() + (size_t)K * chunk, lens.data() + (size_t)K * chunk};
                    who[K++] = &o;
                }
                cat += ccoef[k];
            }
            uint8_t *blob = nullptr;
            rc = jpeg_encode_sets(ctx, l->stream, sets, K, m, quality, &blob);
            if (rc) break;
            for (int k = 0; k < K; k++)
                for (int i = 0; i < m; i++) { who[k]->dst[i0 + i].data = blob + sets[k].offs[i]; who[k]->dst[i0 + i].len = sets[k].lens[i]; }
            if (blob) {
                std::lock_guard<std::mutex> lk(res_mu);
                res->blobs.push_back(blob);
            }
        }
        if (rc) {
            std::lock_guard<std::mutex> lk(res_mu);
            if (status == IPX_OK) { status = rc; err_text = ipx_last_error(); }
        }
        (void)hipStreamSynchronize(l->stream);
    };
    auto guarded = [&](Lane *l) {
        std::string text;
        const int rc = guarded_status([&] { worker(l); }, &text);
        if (!rc) return;
        (void)hipStreamSynchronize(l->stream);   // whatever the worker queued may still be reading the lane's scratch
        std::lock_guard<std::mutex> lk(res_mu);
        if (status == IPX_OK) { status = rc; err_text = text; }
    };
    HostPool::instance().parallel_for(nl, nl, [&](int i) { guarded(lanes[i]); });   // one host thread per lane (they block on their streams)
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        for (auto *l : lanes) l->busy = false;
    }
    ctx->cv.notify_all();
    if (status != IPX_OK) {
        ipx_jpeg_result_free(ctx, res.release());
        set_error("%s", err_text.c_str());
        return status;
    }
    *result = res.release();
    return IPX_OK;
}

extern "C" {

int ipx_plan_run_host_jpeg(ipx_ctx *ctx, const ipx_plan *pl, int n, const uint8_t *src, int sstride, size_t src_frame_stride,
                           int quality, ipx_bytes *resize_out, ipx_bytes *thumb_out, ipx_bytes *wm_out, ipx_jpeg_result **result) try
{
    IPX_ENTER(ctx);
    if (!pl || n < 0 || !src || !result || (long long)sstride < (long long)pl->p.sw * 4) {
        set_error("ipx_plan_run_host_jpeg: bad argument");
        return IPX_ERR_INVALID;
    }
    return run_host_jpeg_impl(ctx, pl, n, src, sstride, src_frame_stride, nullptr, quality, resize_out, thumb_out, wm_out, result);
}
IPX_CATCH_STATUS

int ipx_plan_run_host_ycbcr_jpeg(ipx_ctx *ctx, const ipx_plan *pl, int n, const ipx_ycbcr_batch *src, int quality,
                                 ipx_bytes *resize_out, ipx_bytes *thumb_out, ipx_bytes *wm_out, ipx_jpeg_result **result) try
{
    IPX_ENTER(ctx);
    if (!pl || n < 0 || !result || !src || !src->y || !src->cb || !src->cr || src->ratio < 0 || src->ratio > IPX_YCBCR_440 ||
        src->ystride < pl->p.sw) {
        set_error("ipx_plan_run_host_ycbcr_jpeg: bad argument");
        return IPX_ERR_INVALID;
    }
    return run_host_jpeg_impl(ctx, pl, n, nullptr, 0, 0, src, quality, resize_out, thumb_out, wm_out, result);
}
IPX_CATCH_STATUS

int ipx_jpeg_encode_rgba8(ipx_ctx *ctx, const uint8_t *pix, int w, int h, int stride, int quality, uint8_t **out, size_t *len) try
{
    IPX_ENTER(ctx);
    if (!pix || !out || !len || w <= 0 || h <= 0 || (long long)stride < (long long)w * 4) {
        set_error("ipx_jpeg_encode_rgba8: bad argument");
        return IPX_ERR_INVALID;
    }
    if (w >= 1 << 16 || h >= 1 << 16) { set_error("jpeg: image is too large to encode"); return IPX_ERR_INVALID; }
    const size_t fbytes = align256((size_t)w * h * 4), per = ipx_jpeg_coef_count(w, h) * sizeof(int16_t);
    std::vector<int16_t> host(per / sizeof(int16_t));
    {
        LaneLease lane(ctx);
        int rc = lane_reserve(lane.get(), fbytes + per);
        if (rc) return rc;
        hipStream_t s = lane->stream;
        IPX_HIP(hipMemcpy2DAsync(lane->dev, (size_t)w * 4, pix, stride, (size_t)w * 4, h, hipMemcpyHostToDevice, s));
        rc = ipx_dev_jpeg_fdct_rgba8(ctx, s, lane->dev, w, h, w * 4, fbytes, 1, quality, (int16_t *)(lane->dev + fbytes));
        if (rc) { (void)hipStreamSynchronize(s); return rc; }
        IPX_HIP(hipMemcpyAsync(host.data(), lane->dev + fbytes, per, hipMemcpyDeviceToHost, s));
        IPX_HIP(hipStreamSynchronize(s));
    }
    return ipx_jpeg_entropy_encode(host.data(), w, h, quality, out, len);
}
IPX_CATCH_STATUS

}  // extern "C"


// ---- image.Decode for JPEG batches -----------------------------------------------------------------------
struct ipx_jpeg_planes { std::vector<void *> dev; hipStream_t stream = nullptr; };   // stream-ordered allocations of `stream`

extern "C" {

void ipx_jpeg_planes_free(ipx_ctx *ctx, ipx_jpeg_planes *o)
{
    if (!o) return;
    if (ctx) (void)hipSetDevice(ctx->device);
    // stream-ordered, like the allocation: hipMalloc / hipFree wait for EVERY stream of the device, and with several decodes in flight on
    // lanes of their own each such call waited for all the others' kernels (four concurrent parts: 0.9 s per decode instead of 0.03 s)
    for (void *p : o->dev) (void)hipFreeAsync(p, o->stream);
    delete o;
}

static int decode_batch(ipx_ctx *ctx, hipStream_t s, Lane *lane, bool planes_in_lane, const ipx_bytes *jpegs, int n, int *w, int *h,
                        ipx_ycbcr_batch *planes, int *status, ipx_jpeg_planes **owner);

int ipx_jpeg_decode_batch(ipx_ctx *ctx, void *stream, const ipx_bytes *jpegs, int n, int *w, int *h, ipx_ycbcr_batch *planes,
                          int *status, ipx_jpeg_planes **owner) try
{
    IPX_ENTER(ctx);
    if (!jpegs || n < 0 || !w || !h || !planes || !status || !owner) { set_error("ipx_jpeg_decode_batch: bad argument"); return IPX_ERR_INVALID; }
    *owner = nullptr;
    memset(planes, 0, sizeof *planes);
    if (n == 0) return IPX_OK;
    if (n > 65535) { set_error("ipx_jpeg_decode_batch: at most 65535 files per call"); return IPX_ERR_UNSUPPORTED; }
    // the scratch comes out of a lane's decode buffer for the duration of the call; the planes are the caller's (stream-ordered allocations)
    LaneLease lane(ctx);
    return decode_batch(ctx, stream ? (hipStream_t)stream : ctx->stream, env_int("IPX_JPEG_LANE_ARENA", 1) ? &lane.get() : nullptr, false, jpegs, n, w, h,
                        planes, status, owner);
}
IPX_CATCH_STATUS

// lane != NULL: scratch is bumped out of the lane's decode buffer (no allocation in the steady state); planes_in_lane: the planes too --
// the caller then holds the lane for as long as it uses them and *owner has nothing to free
static int decode_batch(ipx_ctx *ctx, hipStream_t s, Lane *lane, bool planes_in_lane, const ipx_bytes *jpegs, int n, int *w, int *h,
                        ipx_ycbcr_batch *planes, int *status, ipx_jpeg_planes **owner)
{
    std::vector<JpegDecInfo> info(n);
    std::vector<JpegDecTables> tabs(n);
    std::vector<JpegDecImage> items;
    std::vector<uint8_t> valid(n, 0);
    std::vector<size_t> blob_off(n, 0);
    // host preparation runs on a few threads: parsing is trivial, but finding the RSTn markers and packing the scans walk
    // every compressed byte (0.3 GB for a thousand 1080p files)
    std::atomic<int> prep_failed{IPX_OK};     // an exception inside a preparation thread (allocation): checked after each parallel_for
    // (the process-wide pool of ipx_threads.h: sized from the CPUs this process may use, no thread started per call)
    auto parallel_for = [&](int count, const std::function<void(int)> &fn) {            // light items: a thread per eight of them
        HostPool::instance().parallel_for(count, std::max(1, std::min(count / 8, 16)), [&](int i) {
            const int rc = guarded_status([&] { fn(i); }, nullptr);
            if (rc) prep_failed = rc;
        });
    };
    auto parallel_for_each = [&](int count, const std::function<void(int)> &fn) {      // heavy items (a file's scans): a thread each, up to 16
        HostPool::instance().parallel_for(count, 16, [&](int i) {
            const int rc = guarded_status([&] { fn(i); }, nullptr);
            if (rc) prep_failed = rc;
        });
    };
    // pieces of a scan: the whole scan, or one per restart interval.  Inside entropy-coded data 0xff is followed by 0x00 or by a
    // marker, so every 0xff 0xd0..0xd7 pair is an RSTn.
    std::vector<std::vector<uint32_t>> marks(n);
    // files whose scans are walked on the host (progressive, several scans): decoded further down, once the batch's geometry is known,
    // straight into one pinned block in the IDCT kernel's layout (slot hslot[i]): the upload then is plain DMA.  (Uploading from the
    // pageable vectors the first version decoded into took a quarter of such a call: 6.2 MB of dense coefficients per 1080p file.)
    std::vector<int> hslot(n, -1);
    const auto td0 = std::chrono::steady_clock::now();
    auto dms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - td0).count(); };
    double t_parse = 0, t_alloc = 0, t_pin = 0, t_pack = 0, t_launch = 0;
    parallel_for(n, [&](int i) {
        status[i] = !jpegs[i].data ? IPX_ERR_INVALID : (jpegs[i].len >= ((size_t)1 << 30) ? IPX_ERR_UNSUPPORTED : jpeg_parse(jpegs[i].data, jpegs[i].len, &info[i], &tabs[i]));
        if (status[i] != IPX_OK) return;
        if (info[i].host_scans) return;      // its frame header is known; the scans wait for the batch's geometry
        const JpegDecInfo &I = info[i];
        const int nmcu = ((I.w + 8 * I.h0 - 1) / (8 * I.h0)) * ((I.h + 8 * I.v0 - 1) / (8 * I.v0));
        if (I.ri <= 0 || nmcu <= I.ri) return;
        const uint8_t *sd = jpegs[i].data + I.scan_off;
        int expected = 0;
        for (size_t k = 0; k + 1 < I.scan_len;) {
            const uint8_t *q = (const uint8_t *)memchr(sd + k, 0xff, I.scan_len - 1 - k);
            if (!q) break;
            k = (size_t)(q - sd);
            const uint8_t m2 = sd[k + 1];
            if (m2 == 0x00) { k += 2; continue; }
            if (m2 < 0xd0 || m2 > 0xd7) break;                        // EOI or another marker: the scan ends here
            if (m2 != 0xd0 + expected) { status[i] = IPX_ERR_UNSUPPORTED; return; }
            marks[i].push_back((uint32_t)k);
            expected = (expected + 1) & 7;
            k += 2;
        }
        if ((int)marks[i].size() != (nmcu + I.ri - 1) / I.ri - 1) status[i] = IPX_ERR_UNSUPPORTED;   // Go would try to resynchronise
    });
    t_parse = dms();
    if (prep_failed) { set_error("jpeg decode: host preparation failed"); return prep_failed; }
    int ref = -1;
    size_t blob_bytes = 0, piece_ubytes = 0;
    // table classes: tab_of[i] = the first image of the batch whose Huffman tables equal image i's (most batches have one class, the
    // Annex K tables every encoder defaults to); the piece kernels share one set per workgroup, so pieces are grouped by class
    std::vector<int> tab_of(n, -1);
    {
        std::vector<int> reps;
        auto same = [&](int x, int y) {
            return !memcmp(tabs[x].lut, tabs[y].lut, sizeof tabs[x].lut) && !memcmp(tabs[x].maxcode, tabs[y].maxcode, sizeof tabs[x].maxcode) &&
                   !memcmp(tabs[x].valoff, tabs[y].valoff, sizeof tabs[x].valoff) && !memcmp(tabs[x].vals, tabs[y].vals, sizeof tabs[x].vals);
        };
        for (int i = 0; i < n; i++) {
            if (status[i] != IPX_OK || info[i].host_scans) continue;
            for (size_t k = reps.size(); k-- > 0 && tab_of[i] < 0;)     // newest first: neighbours tend to match
                if (same(i, reps[k])) tab_of[i] = reps[k];
            if (tab_of[i] < 0) { tab_of[i] = i; reps.push_back(i); }
            if (reps.size() > 64) break;                                   // a batch of hand-optimised tables: not worth the quadratic search
        }
        for (int i = 0; i < n; i++) if (status[i] == IPX_OK && tab_of[i] < 0) tab_of[i] = i;
    }
    std::vector<JpegParImage> par;
    const bool use_par = env_int("IPX_JPEG_PAR", 1) != 0;
    // Sub-sequences of the scans that are decoded in parallel (ipx_jpeg_dec_par.hip): 1 KiB each for a large batch.  A lane walks its
    // sub-sequence symbol by symbol, so a pass over a small batch takes as long as ONE sub-sequence takes while the chip idles (8
    // files: 44 waves, 1.3 ms per pass, three passes); shorter ones until the batch fills about eight waves per CU.
    int par_sub = jpeg_par_sub_bytes();
    {
        const int forced = env_int("IPX_JPEG_PAR_SUB", 0);
        if (forced == 128 || forced == 256 || forced == 512 || forced == 1024) par_sub = forced;
        else {
            size_t total = 0;
            for (int i = 0; i < n; i++) if (status[i] == IPX_OK && !info[i].host_scans) total += info[i].scan_len;
            while (par_sub > 256 && total / (size_t)par_sub < (size_t)131072) par_sub >>= 1;     // (128 measured no better: 3.0 against 3.1 ms for 8 files, worse for 64)
        }
    }
    for (int i = 0; i < n; i++) {
        if (status[i] == IPX_OK) {
            if (ref < 0 && (*w <= 0 || (info[i].w == *w && info[i].h == *h))) ref = i;
            if (ref >= 0 && (info[i].w != info[ref].w || info[i].h != info[ref].h || info[i].h0 != info[ref].h0 || info[i].v0 != info[ref].v0 ||
                             info[i].ncomp != info[ref].ncomp))
                status[i] = IPX_ERR_UNSUPPORTED;
            else if (ref < 0) status[i] = IPX_ERR_UNSUPPORTED;   // a size other than the one asked for
        }
        if (status[i] != IPX_OK) continue;
        if (info[i].host_scans) { valid[i] = info[i].progressive ? 3 : 1; continue; }     // coefficients come from the host; bit 1: progressive
        const JpegDecInfo &I = info[i];
        const int nmcu = ((I.w + 8 * I.h0 - 1) / (8 * I.h0)) * ((I.h + 8 * I.v0 - 1) / (8 * I.v0));
        auto push = [&](size_t a0, size_t a1, int m0, int cnt) {
            JpegDecImage it;
            memset(&it, 0, sizeof it);
            it.scan_off = blob_bytes + (a0 & ~(size_t)15); it.scan_len = (uint32_t)(a1 - (a0 & ~(size_t)15));
            it.img = (uint32_t)i; it.first_mcu = (uint32_t)m0; it.n_mcu = (uint32_t)cnt;
            memcpy(it.td, I.td, 3); memcpy(it.ta, I.ta, 3);
            it.valid = 1;
            it.pad = (uint8_t)(a0 & 15);           // bytes to skip: pieces start 16-byte aligned for the kernel's chunk loads
            it.uoff = piece_ubytes;                // its unstuffed copy (launch_jpeg_pieces): a region of its own
            it.tab_img = (uint32_t)tab_of[i];
            piece_ubytes += ((a1 - a0) + 15 + 16) & ~(size_t)15;
            items.push_back(it);
        };
        size_t start = 0;
        int mcu = 0;
        for (uint32_t k : marks[i]) { push(start, k, mcu, I.ri); items.back().strict_end = 1; mcu += I.ri; start = (size_t)k + 2; }
        if (marks[i].empty() && use_par && I.scan_len >= (size_t)4 * jpeg_par_sub_bytes() && I.scan_len < ((size_t)1 << 28)) {   // (the same files whatever par_sub is)
            // a long scan without restart markers: decoded in parallel inside the scan (ipx_jpeg_dec_par.hip)
            JpegParImage pi;
            memset(&pi, 0, sizeof pi);
            pi.scan_off = blob_bytes; pi.scan_len = (uint32_t)I.scan_len; pi.img = (uint32_t)i;
            pi.nsub = (uint32_t)((I.scan_len + par_sub - 1) / par_sub);
            memcpy(pi.td, I.td, 3); memcpy(pi.ta, I.ta, 3);
            par.push_back(pi);
        } else {
            push(start, I.scan_len, mcu, nmcu - mcu);
        }
        valid[i] = 1;
        blob_off[i] = blob_bytes;
        blob_bytes += (I.scan_len + 15 + 16) & ~(size_t)15;
    }
    if (ref < 0) return IPX_OK;
    {
        // group the pieces by table class (stable: image order inside a class), each group padded to whole workgroups of 64
        bool one_class = true;
        for (auto &it : items) one_class = one_class && it.tab_img == items[0].tab_img;
        if (!one_class) {
            std::stable_sort(items.begin(), items.end(), [](const JpegDecImage &x, const JpegDecImage &y) { return x.tab_img < y.tab_img; });
            std::vector<JpegDecImage> grouped;
            JpegDecImage pad;
            memset(&pad, 0, sizeof pad);
            for (size_t k = 0; k < items.size(); k++) {
                if (k && items[k].tab_img != items[k - 1].tab_img)
                    while (grouped.size() & 63) { pad.tab_img = items[k - 1].tab_img; grouped.push_back(pad); }
                grouped.push_back(items[k]);
            }
            items.swap(grouped);
        }
    }
    const JpegDecInfo &R = info[ref];
    *w = R.w; *h = R.h;
    JpegDecArgs a{};
    a.n = n; a.h0 = R.h0; a.v0 = R.v0; a.w = R.w; a.h = R.h;
    a.mxx = (R.w + 8 * R.h0 - 1) / (8 * R.h0); a.myy = (R.h + 8 * R.v0 - 1) / (8 * R.v0);
    const bool gray = R.ncomp == 1;                        // *image.Gray: one block per MCU, no chroma planes
    a.ybl = R.h0 * R.v0; a.bpm = gray ? 1 : a.ybl + 2;
    a.nblk = a.mxx * a.myy * a.bpm;
    JpegPlanes pl{};
    pl.ystride = 8 * R.h0 * a.mxx; pl.cstride = 8 * a.mxx;
    pl.y_fs = align256((size_t)pl.ystride * 8 * R.v0 * a.myy); pl.c_fs = gray ? 0 : align256((size_t)pl.cstride * 8 * a.myy);

    std::unique_ptr<ipx_jpeg_planes> own(new ipx_jpeg_planes);
    own->stream = s;
    // scratch of this call: stream-ordered, or bumped out of the lane's decode buffer
    AsyncFree mem{s, {}};
    if (lane) {
        size_t subs = 0;
        for (auto &pi : par) subs = std::max(subs, (size_t)pi.nsub);
        subs *= par.size();
        const size_t est = (pl.y_fs + 2 * pl.c_fs) * n + (size_t)n * a.nblk * 130 + 3 * (blob_bytes + 1024) + piece_ubytes +
                           items.size() * (sizeof(JpegDecImage) + 8) + (size_t)n * (sizeof(JpegDecTables) + 64) + par.size() * (sizeof(JpegParImage) + 64) +
                           subs * 96 + ((size_t)4 << 20);
        const int rr = lane_reserve_dec(*lane, est);
        if (rr) return rr;
        mem.arena = lane->dec; mem.cap = lane->dec_bytes;
    }
    auto dalloc = [&](void **p, size_t bytes) {
        if (lane && planes_in_lane) return mem.get((uint8_t **)p, bytes);   // first requests of the call and counted in est: they always fit
        hipError_t e = hipMallocAsync(p, bytes ? bytes : 1, s);
        if (e == hipSuccess) own->dev.push_back(*p);
        return e;
    };
    auto fail = [&](hipError_t e, const char *what) {
        set_error("%s: %s", what, hipGetErrorString(e));
        ipx_jpeg_planes_free(ctx, own.release());
        return IPX_ERR_HIP;
    };
    hipError_t e;
    if ((e = dalloc((void **)&pl.y, pl.y_fs * n)) != hipSuccess) return fail(e, "plane allocation");
    if (!gray && (e = dalloc((void **)&pl.cb, pl.c_fs * n)) != hipSuccess) return fail(e, "plane allocation");
    if (!gray && (e = dalloc((void **)&pl.cr, pl.c_fs * n)) != hipSuccess) return fail(e, "plane allocation");
    uint8_t *d_blob; JpegDecImage *d_img; JpegDecTables *d_tab; int16_t *d_coefs; int *d_status;
    if ((e = mem.get(&d_blob, blob_bytes + 16)) != hipSuccess) return fail(e, "scratch allocation");
    uint8_t *d_valid;
    if ((e = mem.get(&d_img, sizeof(JpegDecImage) * items.size())) != hipSuccess) return fail(e, "scratch allocation");
    if ((e = mem.get(&d_valid, (size_t)n)) != hipSuccess) return fail(e, "scratch allocation");
    if ((e = mem.get(&d_tab, sizeof(JpegDecTables) * n)) != hipSuccess) return fail(e, "scratch allocation");
    if ((e = mem.get(&d_coefs, (size_t)n * a.nblk * 128)) != hipSuccess) return fail(e, "scratch allocation");
    if ((e = mem.get(&d_status, sizeof(int) * n)) != hipSuccess) return fail(e, "scratch allocation");
    int16_t *d_dcs;
    if ((e = mem.get(&d_dcs, (size_t)n * a.nblk * 2 + 16)) != hipSuccess) return fail(e, "scratch allocation");
    t_alloc = dms();
    uint8_t *hblob = (uint8_t *)ipx_host_alloc(ctx, blob_bytes + 16);
    if (!hblob) { ipx_jpeg_planes_free(ctx, own.release()); return IPX_ERR_NOMEM; }
    // the host-decoded files of the batch (progressive, several scans): their scans are walked on the preparation threads, further
    // down, in groups of kHostGroup files through a pinned block of two groups -- one uploads while the next decodes.  (One block for
    // all of them was 6.3 MB per 1080p file, 1.6 GB for a part of 256 progressive files, per part and per feeder of a pool.)
    int nhost = 0;
    for (int i = 0; i < n; i++) if (valid[i] && info[i].host_scans) hslot[i] = nhost++;
    int16_t *hpin = nullptr;
    const size_t hcoef_words = (size_t)a.nblk * 64, hslot_words = hcoef_words + a.nblk;
    const int kHostGroup = std::max(1, env_int("IPX_JPEG_HOST_GROUP", 16));
    std::vector<int> hfiles;
    for (int i = 0; i < n; i++) if (hslot[i] >= 0) hfiles.push_back(i);
    if (nhost) {
        hpin = (int16_t *)ipx_host_alloc(ctx, (size_t)2 * std::min(nhost, kHostGroup) * hslot_words * sizeof(int16_t));
        if (!hpin) {   // no pinned memory for them: these files stay on the caller's CPU path, the rest of the batch goes on
            for (int i : hfiles) { status[i] = IPX_ERR_UNSUPPORTED; valid[i] = 0; }
            hfiles.clear();
            clear_error();
        }
    }
    t_pin = dms();
    parallel_for(n, [&](int i) { if (valid[i] && !info[i].host_scans) memcpy(hblob + blob_off[i], jpegs[i].data + info[i].scan_off, info[i].scan_len); });
    if (prep_failed) { (void)ipx_host_free(ctx, hblob); if (hpin) (void)ipx_host_free(ctx, hpin); ipx_jpeg_planes_free(ctx, own.release()); set_error("jpeg decode: host preparation failed"); return prep_failed; }
    t_pack = dms();
    e = hipMemcpyAsync(d_blob, hblob, blob_bytes, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_img, items.data(), sizeof(JpegDecImage) * items.size(), hipMemcpyHostToDevice, s);
    pl.valid = d_valid;
    a.nitems = (int)items.size();
    if (e == hipSuccess) e = hipMemsetAsync(d_coefs, 0, (size_t)n * a.nblk * 128, s);
    if (e == hipSuccess) e = hipMemsetAsync(d_status, 0, sizeof(int) * n, s);
    if (e == hipSuccess) e = hipMemsetAsync(d_dcs, 0, (size_t)n * a.nblk * 2, s);
    a.blob = d_blob; a.img = d_img; a.tab = d_tab; a.coefs = d_coefs; a.status = d_status; a.dcs = d_dcs;
    // host-decoded files: group by group, scans walked on the pool's threads into one half of the pinned block, coefficients copied into
    // their slots (after the memsets, same stream) while the next group decodes into the other half
    {
        hipEvent_t hev[2] = {nullptr, nullptr};
        bool hused[2] = {false, false};
        for (int g0 = 0, gi = 0; g0 < (int)hfiles.size() && e == hipSuccess; g0 += kHostGroup, gi++) {
            const int half = gi & 1, cnt = std::min(kHostGroup, (int)hfiles.size() - g0);
            int16_t *base = hpin + (size_t)half * std::min(nhost, kHostGroup) * hslot_words;
            if (hused[half]) e = hipEventSynchronize(hev[half]);          // the copies of two groups ago have left this half
            if (e != hipSuccess) break;
            parallel_for_each(cnt, [&](int j) {
                const int i = hfiles[g0 + j];
                bool prog = false;
                JpegDecInfo full;
                const int rc = jpeg_host_decode(jpegs[i].data, jpegs[i].len, &full, base + (size_t)j * hslot_words, base + (size_t)j * hslot_words + hcoef_words,
                                                (size_t)a.nblk, tabs[i].qnat, &prog);
                if (rc != IPX_OK) { status[i] = rc; valid[i] = 0; }
            });
            if (prep_failed) break;
            for (int j = 0; j < cnt && e == hipSuccess; j++) {
                const int i = hfiles[g0 + j];
                if (!valid[i]) continue;
                e = hipMemcpyAsync(d_coefs + (size_t)i * hcoef_words, base + (size_t)j * hslot_words, hcoef_words * 2, hipMemcpyHostToDevice, s);
                if (e == hipSuccess) e = hipMemcpyAsync(d_dcs + (size_t)i * a.nblk, base + (size_t)j * hslot_words + hcoef_words, (size_t)a.nblk * 2, hipMemcpyHostToDevice, s);
            }
            if (e == hipSuccess && !hev[half]) e = hipEventCreateWithFlags(&hev[half], hipEventDisableTiming);
            if (e == hipSuccess) { e = hipEventRecord(hev[half], s); hused[half] = true; }
        }
        for (hipEvent_t ev : hev) if (ev) (void)hipEventDestroy(ev);
        if (prep_failed) {
            (void)hipStreamSynchronize(s);
            (void)ipx_host_free(ctx, hblob); if (hpin) (void)ipx_host_free(ctx, hpin); ipx_jpeg_planes_free(ctx, own.release());
            set_error("jpeg decode: host preparation failed");
            return prep_failed;
        }
    }
    // (which files are decodable is final only now: a host-decoded file may have failed in its scans; and the host decoder fills the
    // quantisation tables of its files)
    if (e == hipSuccess) e = hipMemcpyAsync(d_valid, valid.data(), (size_t)n, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_tab, tabs.data(), sizeof(JpegDecTables) * n, hipMemcpyHostToDevice, s);
    int ref_gpu = -1;                                        // the first image the Huffman kernels decode: the one whose tables a shared-table launch carries
    for (int i = 0; i < n && ref_gpu < 0; i++) if (valid[i] && !info[i].host_scans) ref_gpu = i;
    a.first_valid = ref_gpu >= 0 ? ref_gpu : ref;
    a.shared_tables = env_int("IPX_JPEG_SHARED_TABLES", 1);
    for (int i = 0; i < n && a.shared_tables && ref_gpu >= 0; i++)
        if (valid[i] && !info[i].host_scans &&
            (memcmp(tabs[i].lut, tabs[ref_gpu].lut, sizeof tabs[i].lut) || memcmp(tabs[i].maxcode, tabs[ref_gpu].maxcode, sizeof tabs[i].maxcode) ||
             memcmp(tabs[i].valoff, tabs[ref_gpu].valoff, sizeof tabs[i].valoff) || memcmp(tabs[i].vals, tabs[ref_gpu].vals, sizeof tabs[i].vals)))
            a.shared_tables = 0;
    if (e == hipSuccess && a.nitems > 0) {
        if (env_int("IPX_JPEG_PIECE", 1)) {
            uint8_t *d_upieces; uint32_t *d_ulen;
            e = mem.get(&d_upieces, piece_ubytes + 64);
            if (e == hipSuccess) e = mem.get(&d_ulen, sizeof(uint32_t) * items.size());
            if (e == hipSuccess) e = launch_jpeg_pieces(a, d_upieces, d_ulen, s);
        } else {
            e = launch_jpeg_huff(a, s);     // the earlier kernel: byte-wise reader, per-lane tables when the files of the batch carry different ones
        }
    }
    if (e == hipSuccess && !par.empty()) {
        JpegParArgs P{};
        P.blob = d_blob; P.tab = d_tab; P.nimg = (int)par.size(); P.bpm = a.bpm; P.ybl = a.ybl; P.nblk = a.nblk;
        P.coefs = d_coefs; P.status = d_status; P.dcs = d_dcs;
        P.sub = par_sub;
        for (auto &pi : par) P.max_nsub = std::max(P.max_nsub, (int)pi.nsub);
        for (size_t k = 0; k < par.size(); k++) par[k].sub_off = k * (size_t)P.max_nsub;
        const size_t nsubs = par.size() * (size_t)P.max_nsub;
        // The scan bytes of a wave's 64 sub-sequences staged in LDS, or read through L1 / L2.  A big batch hides the latency of the global
        // reads behind its other waves and loses more to the occupancy the rows cost (1024 x 1080p, 1 KiB rows: 109 ms staged against
        // 49); in a small one a SIMD has one wave, and that wave waits for a global load nearly every symbol, because some lane of the 64
        // crosses a 16-byte group each step.  tools/par_stage.sh, 1080p files: 1 file 2.8 -> 2.6 ms, 8 files 3.5 -> 3.4, 64 files (86 k
        // sub-sequences) 7.4 -> 6.9; 256 files (345 k) 17.7 -> 20.0.  Staged while every wave of the batch is resident at once.
        P.stage_rows = env_int("IPX_JPEG_PAR_STAGE", -1);
        if (P.stage_rows < 0) P.stage_rows = nsubs <= (size_t)env_int("IPX_JPEG_PAR_STAGE_SUBS", 98304) ? 1 : 0;
        JpegParImage *d_par = nullptr; uint32_t *d_tot = nullptr;
        if (e == hipSuccess) e = mem.get(&d_par, sizeof(JpegParImage) * par.size());
        if (e == hipSuccess) e = mem.get(&P.stuffed, nsubs * 4);
        if (e == hipSuccess) e = mem.get(&P.entry, nsubs * 8);
        if (e == hipSuccess) e = mem.get(&P.exit_a, nsubs * 8);
        if (e == hipSuccess) e = mem.get(&P.exit_b, nsubs * 8);
        if (e == hipSuccess) e = mem.get(&P.ends, nsubs * 4);
        if (e == hipSuccess) e = mem.get(&P.ck_state, nsubs * 8 * jpeg_par_checkpoints());
        if (e == hipSuccess) e = mem.get(&P.ck_ends, nsubs * 4 * jpeg_par_checkpoints());
        if (e == hipSuccess) e = mem.get(&P.total_ends, par.size() * 4);
        if (e == hipSuccess) e = mem.get(&d_tot, par.size() * 4);
        if (e == hipSuccess) e = mem.get(&P.changed, 4);
        P.img = d_par;
        if (e == hipSuccess) e = hipMemcpyAsync(d_par, par.data(), sizeof(JpegParImage) * par.size(), hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipMemsetAsync(P.stuffed, 0, nsubs * 4, s);
        if (e == hipSuccess) e = hipMemsetAsync(P.entry, 0xff, nsubs * 8, s);
        if (e == hipSuccess) e = hipMemsetAsync(P.ends, 0, nsubs * 4, s);
        if (e == hipSuccess) e = hipMemsetAsync(P.ck_state, 0xff, nsubs * 8 * jpeg_par_checkpoints(), s);
        if (e == hipSuccess) e = hipMemsetAsync(P.ck_ends, 0, nsubs * 4 * jpeg_par_checkpoints(), s);
        if (e == hipSuccess) e = mem.get(&P.ublob, blob_bytes + 64);
        if (e == hipSuccess) e = mem.get(&P.scan_end, par.size() * 4);
        if (e == hipSuccess) e = mem.get(&P.ulen, par.size() * 4);
        if (e == hipSuccess) e = hipMemsetAsync(P.ublob, 0, blob_bytes + 64, s);
        if (e == hipSuccess) e = hipMemsetAsync(P.scan_end, 0xff, par.size() * 4, s);
        if (e == hipSuccess) e = hipMemsetAsync(P.ulen, 0, par.size() * 4, s);
        if (e == hipSuccess) e = launch_par_count(P, s);
        if (e == hipSuccess) e = launch_scan(P.stuffed, P.max_nsub, P.nimg, d_tot, s);
        if (e == hipSuccess) e = launch_par_unstuff(P, s);
        if (e == hipSuccess) e = launch_par_sync(P, 0, s);
        bool converged = false;
        const int max_rounds = env_int("IPX_JPEG_PAR_ROUNDS", 96);
        for (int round = 1; e == hipSuccess && round <= max_rounds; round++) {
            uint32_t changed = 0;
            e = hipMemsetAsync(P.changed, 0, 4, s);
            if (e == hipSuccess) e = launch_par_sync(P, round, s);
            if (e == hipSuccess) e = hipMemcpyAsync(&changed, P.changed, 4, hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (getenv("IPX_DEBUG")) fprintf(stderr, "[ipx] jpeg par sync round %d: %u entries changed\n", round, changed);
            if (e == hipSuccess && changed == 0) { converged = true; break; }
        }
        if (e == hipSuccess && !converged) {
            // a scan that never settled (it would take a pathological file): hand these images to the serial kernel
            std::vector<JpegDecImage> serial;
            for (auto &pi : par) {
                JpegDecImage it;
                memset(&it, 0, sizeof it);
                it.scan_off = pi.scan_off; it.scan_len = pi.scan_len; it.img = pi.img; it.first_mcu = 0; it.n_mcu = (uint32_t)(a.mxx * a.myy);
                memcpy(it.td, pi.td, 3); memcpy(it.ta, pi.ta, 3);
                it.valid = 1;
                serial.push_back(it);
            }
            JpegDecImage *d_serial;
            e = mem.get(&d_serial, sizeof(JpegDecImage) * serial.size());
            if (e == hipSuccess) e = hipMemcpyAsync(d_serial, serial.data(), sizeof(JpegDecImage) * serial.size(), hipMemcpyHostToDevice, s);
            JpegDecArgs a2 = a;
            a2.img = d_serial; a2.nitems = (int)serial.size();
            if (e == hipSuccess) e = launch_jpeg_huff(a2, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);   // `serial` must outlive the copy
        } else if (e == hipSuccess) {
            e = launch_scan(P.ends, P.max_nsub, P.nimg, P.total_ends, s);
            if (e == hipSuccess) e = launch_par_write(P, s);
            if (e == hipSuccess) e = launch_par_dc(P, s);
        }
    }
    if (e == hipSuccess) e = launch_jpeg_idct(a, pl, s);
    t_launch = dms();
    std::vector<int> dev_status(n, 0);
    if (e == hipSuccess) e = hipMemcpyAsync(dev_status.data(), d_status, sizeof(int) * n, hipMemcpyDeviceToHost, s);
    // unconditionally: after a failed enqueue the copies and kernels queued before it may still be reading the pinned blob, the host
    // tables and the lane's arena, all of which are handed back below
    { const hipError_t e2 = hipStreamSynchronize(s); if (e == hipSuccess) e = e2; }
    if ((getenv("IPX_DEBUG") && dms() > 200.0) || env_int("IPX_DEBUG_J2J", 0))
        fprintf(stderr, "[ipx] decode of %d files: parsed at %.1f ms, device scratch at %.1f, pinned block at %.1f, packed at %.1f, launched at %.1f, finished at %.1f\n", n, t_parse, t_alloc, t_pin, t_pack, t_launch, dms());
    (void)ipx_host_free(ctx, hblob);
    if (hpin) (void)ipx_host_free(ctx, hpin);
    if (e != hipSuccess) return fail(e, "jpeg decode");
    for (int i = 0; i < n; i++)
        if (status[i] == IPX_OK && dev_status[i]) status[i] = jpeg_status_of(dev_status[i]);
    planes->y = pl.y; planes->cb = pl.cb; planes->cr = pl.cr;
    planes->ystride = pl.ystride; planes->cstride = pl.cstride;
    planes->y_frame_stride = pl.y_fs; planes->c_frame_stride = pl.c_fs;
    planes->ratio = R.ratio;
    *owner = own.release();
    return IPX_OK;
}

}  // extern "C"


// ---- compressed in, compressed out: image.Decode, the operators and jpeg.Encode without leaving the GPU ------
extern "C" {

static int run_jpeg_jpeg_one(ipx_ctx *ctx, const ipx_plan *pl, int n, const ipx_bytes *files, int quality, ipx_bytes *resize_out,
                             ipx_bytes *thumb_out, ipx_bytes *wm_out, int *status, ipx_jpeg_result **result)
{
    IPX_ENTER(ctx);
    *result = nullptr;
    if (n == 0) return IPX_OK;
    const int sw = pl->p.sw, sh = pl->p.sh;
    for (int i = 0; i < n; i++) {
        if (resize_out) resize_out[i] = ipx_bytes{nullptr, 0};
        if (thumb_out) thumb_out[i] = ipx_bytes{nullptr, 0};
        if (wm_out) wm_out[i] = ipx_bytes{nullptr, 0};
    }
    const bool dbg = getenv("IPX_DEBUG") != nullptr;
    struct Slot {
        ipx_ctx *c;
        explicit Slot(ipx_ctx *ctx) : c(ctx)
        {
            const int cap = std::max(1, env_int("IPX_JPEG_JPEG_PARALLEL", 4));
            std::unique_lock<std::mutex> lk(c->mu);
            c->cv.wait(lk, [&] { return c->jj_active < cap; });
            c->jj_active++;
        }
        ~Slot()
        {
            { std::lock_guard<std::mutex> lk(c->mu); c->jj_active--; }
            c->cv.notify_all();
        }
    } slot(ctx);
    const auto t0 = std::chrono::steady_clock::now();
    auto ms_since = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
    LaneLease lane(ctx);
    const double t_lane = ms_since(t0);
    hipStream_t s = lane->stream;
    int w = sw, h = sh;
    ipx_ycbcr_batch planes;
    ipx_jpeg_planes *owner = nullptr;
    memset(&planes, 0, sizeof planes);
    int rc = n > 65535 ? IPX_ERR_UNSUPPORTED : decode_batch(ctx, s, env_int("IPX_JPEG_LANE_ARENA", 1) ? &lane.get() : nullptr, true, files, n, &w, &h, &planes, status, &owner);
    const double t_dec = ms_since(t0);
    if (rc) return rc;
    if (!planes.y) return IPX_OK;                         // nothing decodable: every status says why
    struct Guard { ipx_ctx *c; ipx_jpeg_planes *o; ~Guard() { ipx_jpeg_planes_free(c, o); } } guard{ctx, owner};
    const size_t fres = resize_out ? align256(pl->info.resize_bytes) : 0, fth = thumb_out ? align256(pl->info.thumb_bytes) : 0;
    const size_t fwm = wm_out ? align256(pl->info.wm_bytes) : 0;
    const size_t cres = fres ? ipx_jpeg_coef_count(pl->info.resize_w, pl->info.resize_h) * 2 : 0;
    const size_t cth = fth ? ipx_jpeg_coef_count(pl->info.thumb_w, pl->info.thumb_h) * 2 : 0;
    const size_t cwm = fwm ? ipx_jpeg_coef_count(sw, sh) * 2 : 0;
    // (a coefficient buffer per output: the three encodes run stage by stage together, jpeg_encode_sets)
    const size_t ccoef[3] = {align256(cres), align256(cth), align256(cwm)};
    const size_t per_frame = fres + fth + fwm + ccoef[0] + ccoef[1] + ccoef[2];
    if (per_frame == 0) return IPX_OK;
    const int chunk = std::max(1, std::min(n, env_int("IPX_JPEG_JPEG_CHUNK", 256)));
    rc = lane_reserve(lane.get(), per_frame * chunk + 256);
    if (rc) return rc;
    const double t_res = ms_since(t0);
    std::unique_ptr<ipx_jpeg_result> res(new ipx_jpeg_result);
    std::vector<size_t> offs(3 * (size_t)chunk), lens(3 * (size_t)chunk);
    for (int i0 = 0; i0 < n && !rc; i0 += chunk) {
        const int m = std::min(chunk, n - i0);
        uint8_t *base = (uint8_t *)(((uintptr_t)lane->dev + 255) & ~(uintptr_t)255);
        uint8_t *dres = fres ? base : nullptr, *dth = fth ? base + fres * chunk : nullptr, *dwm = fwm ? base + (fres + fth) * chunk : nullptr;
        uint8_t *cbase = base + (fres + fth + fwm) * chunk;
        ipx_ycbcr_batch d = planes;
        d.y += planes.y_frame_stride * i0;
        if (d.cb) { d.cb += planes.c_frame_stride * i0; d.cr += planes.c_frame_stride * i0; }
        if (planes.ratio == IPX_GRAY) rc = ipx_plan_run_dev_gray(ctx, s, pl, m, d.y, d.ystride, d.y_frame_stride, dres, fres, dth, fth, dwm, fwm);
        else rc = ipx_plan_run_dev_ycbcr(ctx, s, pl, m, &d, dres, fres, dth, fth, dwm, fwm);
        if (rc) break;
        struct Out { uint8_t *dev; size_t fs; int w, h; ipx_bytes *dst; };
        const Out outs[3] = {{dres, fres, pl->info.resize_w, pl->info.resize_h, resize_out},
                             {dth, fth, pl->info.thumb_w, pl->info.thumb_h, thumb_out},
                             {dwm, fwm, sw, sh, wm_out}};
        JpegEncSet sets[3];
        const Out *who[3];
        int K = 0;
        size_t cat = 0;
        for (int k = 0; k < 3; k++) {
            const Out &o = outs[k];
            if (o.dev && o.w > 0 && o.h > 0) {
                sets[K] = JpegEncSet{(int16_t *)(cbase + cat * chunk), o.dev, o.w, o.h, o.w * 4, o.fs, offs.data() + (size_t)K * chunk, lens.data() + (size_t)K * chunk};
                who[K++] = &o;
            }
            cat += ccoef[k];
        }
        uint8_t *blob = nullptr;
        rc = jpeg_encode_sets(ctx, s, sets, K, m, quality, &blob);
        if (rc) break;
        if (blob) res->blobs.push_back(blob);
        for (int k = 0; k < K; k++)
            for (int i = 0; i < m; i++)
                if (status[i0 + i] == IPX_OK) { who[k]->dst[i0 + i].data = blob + sets[k].offs[i]; who[k]->dst[i0 + i].len = sets[k].lens[i]; }
    }
    (void)hipStreamSynchronize(s);
    if (dbg) fprintf(stderr, "[ipx] jpeg->jpeg part of %d files: lane after %.1f ms, decoded at %.1f, scratch at %.1f, done at %.1f\n", n, t_lane, t_dec, t_res, ms_since(t0));
    if (rc) { ipx_jpeg_result_free(ctx, res.release()); return rc; }
    *result = res.release();
    return IPX_OK;
}


int ipx_plan_run_jpeg_jpeg(ipx_ctx *ctx, const ipx_plan *pl, int n, const ipx_bytes *files, int quality, ipx_bytes *resize_out,
                           ipx_bytes *thumb_out, ipx_bytes *wm_out, int *status, ipx_jpeg_result **result) try
{
    IPX_ENTER(ctx);
    if (!pl || n < 0 || !files || !status || !result) { set_error("ipx_plan_run_jpeg_jpeg: bad argument"); return IPX_ERR_INVALID; }
    *result = nullptr;
    // a large batch is cut into parts that run on lanes of their own, one host thread each: while one part is in its (host-paced)
    // encode read-backs another decodes.  Two callers with 1024 files each measured 15 k images/s against 11.7 k for one.
    // Four parts of 256 measured 9 % above two of 512 or three of 341 (tools/j2j_parts.sh: 55.3 ms against 60.4 per 1024 files); more
    // than four gain nothing.  One lane stays free for a per-operator call that arrives while the batch runs.
    const int nl = (int)ctx->lanes.size();
    const int parts = std::max(1, std::min({nl >= 3 ? nl - 1 : nl, n / std::max(1, env_int("IPX_JPEG_JPEG_PART", 256)), env_int("IPX_JPEG_JPEG_MAXPARTS", 4)}));
    if (parts == 1) return run_jpeg_jpeg_one(ctx, pl, n, files, quality, resize_out, thumb_out, wm_out, status, result);
    std::vector<ipx_jpeg_result *> res(parts, nullptr);
    std::vector<int> rcs(parts, IPX_OK);
    std::vector<std::string> errs(parts);
    auto work = [&](int k) {
        const int i0 = (int)((long long)n * k / parts), i1 = (int)((long long)n * (k + 1) / parts);
        rcs[k] = run_jpeg_jpeg_one(ctx, pl, i1 - i0, files + i0, quality, resize_out ? resize_out + i0 : nullptr, thumb_out ? thumb_out + i0 : nullptr,
                                   wm_out ? wm_out + i0 : nullptr, status + i0, &res[k]);
        if (rcs[k]) errs[k] = ipx_last_error();
    };
    auto guarded = [&](int k) { const int rc = guarded_status([&] { work(k); }, &errs[k]); if (rc) rcs[k] = rc; };
    HostPool::instance().parallel_for(parts, parts, [&](int k) { guarded(k); });      // one host thread per part (each drives a lane)
    std::unique_ptr<ipx_jpeg_result> all(new ipx_jpeg_result);
    int rc = IPX_OK;
    for (int k = 0; k < parts; k++) {
        if (res[k]) { all->blobs.insert(all->blobs.end(), res[k]->blobs.begin(), res[k]->blobs.end()); delete res[k]; }
        if (rcs[k] && !rc) { rc = rcs[k]; set_error("%s", errs[k].c_str()); }
    }
    if (rc) { ipx_jpeg_result_free(ctx, all.release()); return rc; }
    *result = all.release();
    return IPX_OK;
}
IPX_CATCH_STATUS

}  // extern "C"

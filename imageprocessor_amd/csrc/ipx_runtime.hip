// ipx_runtime.hip -- context, staging lanes, glyph sets, plans and the C ABI of include/ipx.h.
//
// Threading model (worker.go:88-96: WORKER_CONCURRENCY goroutines share one processor): a
// context is safe to call from any number of OS threads.  Host-pointer calls borrow one of
// `lanes` staging lanes (stream + device scratch), blocking only when all lanes are busy;
// device-pointer calls are plain asynchronous launches on the caller's stream.
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <utility>
#include <vector>

#include "ipx_runtime_internal.h"

#ifndef IPX_DIAG
#define IPX_DIAG 0
#endif

namespace {

// Scale's argument checks and dispatch decisions, shared by host- and device-pointer entries
struct ScalePrep {
    bool copy;      // equal sizes: Copy -> DrawMask
    bool empty;
    Rect adr;       // relative to dr.Min
    double xscale, yscale;
};

int scale_prepare(int dw, int dh, const Rect &dr, int sw, int sh, const Rect &sr, int op, ScalePrep *o)
{
    if (op != IPX_OP_OVER && op != IPX_OP_SRC) { set_error("scale: unknown op %d", op); return IPX_ERR_INVALID; }
    o->copy = dr.dx() == sr.dx() && dr.dy() == sr.dy();
    o->empty = false;
    if (o->copy) return IPX_OK;
    Rect adr = Rect{0, 0, dw, dh}.intersect(dr);
    if (adr.empty() || sr.empty()) { o->empty = true; return IPX_OK; }
    if (sr.x0 < 0 || sr.y0 < 0 || sr.x1 > sw || sr.y1 > sh) {
        set_error("scale: source rectangle (%d,%d)-(%d,%d) leaves the %dx%d source; the reference's "
                  "generic Image path is not covered", sr.x0, sr.y0, sr.x1, sr.y1, sw, sh);
        return IPX_ERR_UNSUPPORTED;
    }
    o->adr = adr.shifted(-dr.x0, -dr.y0);
    o->yscale = (double)sr.dy() / (double)dr.dy();
    o->xscale = (double)sr.dx() / (double)dr.dx();
    return IPX_OK;
}

// dst / src are device pointers at pixel (0,0)
// a source image resident in HBM: RGBA / NRGBA pixels, or the three planes of a YCbCr image
struct DevSrc {
    int kind = IPX_SRC_RGBA;
    const uint8_t *pix = nullptr;   // pixels, or the Y plane
    int stride = 0;                 // bytes per row of pix
    const uint8_t *cb = nullptr, *cr = nullptr;
    int cstride = 0, ratio = 0;
    int w = 0, h = 0;
    int nframes = 1;                // a batch: frames frame_stride / c_frame_stride bytes apart
    size_t frame_stride = 0, c_frame_stride = 0;
};

int dev_draw_src(hipStream_t s, uint8_t *dst, int dw, int dh, int dstride, Rect r, const DevSrc &src, int spx,
                 int spy, int op, size_t dst_fs = 0)
{
    if (op != IPX_OP_OVER && op != IPX_OP_SRC) { set_error("draw: unknown op %d", op); return IPX_ERR_INVALID; }
    int mx = 0, my = 0;
    if (!draw_clip(r, dw, dh, true, src.w, src.h, spx, spy, false, 0, 0, mx, my)) return IPX_OK;
    uint8_t *d = dst + (size_t)r.y0 * dstride + (size_t)r.x0 * 4;
    if (src.kind == IPX_SRC_YCBCR)   // opaque source: Over == Src (image/draw.DrawMask's YCbCr arm)
        IPX_HIP(launch_draw_ycbcr(d, dstride, src.pix, src.stride, src.cb, src.cr, src.cstride, src.ratio, spx, spy,
                                  r.dx(), r.dy(), s, src.nframes, dst_fs, src.frame_stride, src.c_frame_stride));
    else if (src.kind == IPX_SRC_NRGBA)
        IPX_HIP(launch_draw_nrgba(d, dstride, src.pix + (size_t)spy * src.stride + (size_t)spx * 4, src.stride, r.dx(),
                                  r.dy(), op, s, src.nframes, dst_fs, src.frame_stride));
    else if (src.kind == IPX_SRC_TAP64)
        IPX_HIP(launch_draw_tap64(d, dstride, src.pix + (size_t)spy * src.stride + (size_t)spx * 8, src.stride, r.dx(), r.dy(), op, s,
                                  src.nframes, dst_fs, src.frame_stride));
    else
        IPX_HIP(launch_draw(d, dstride, src.pix + (size_t)spy * src.stride + (size_t)spx * 4, src.stride, r.dx(), r.dy(),
                            op, s));
    return IPX_OK;
}

int dev_scale_src(hipStream_t s, int *flag, uint8_t *dst, int dw, int dh, int dstride, const Rect &dr,
                  const DevSrc &src, const Rect &sr, int op, size_t dst_fs = 0)
{
    ScalePrep pr;
    int rc = scale_prepare(dw, dh, dr, src.w, src.h, sr, op, &pr);
    if (rc) return rc;
    if (pr.copy) return dev_draw_src(s, dst, dw, dh, dstride, dr, src, sr.x0, sr.y0, op, dst_fs);
    if (pr.empty) return IPX_OK;
    if (src.kind == IPX_SRC_YCBCR) op = IPX_OP_SRC;   // (*image.YCbCr).Opaque() is always true
    if (op == IPX_OP_OVER && src.kind == IPX_SRC_TAP64) IPX_HIP(launch_opaque_scan_tap64(src.pix, src.w, src.h, src.stride, flag, s));
    else if (op == IPX_OP_OVER) IPX_HIP(launch_opaque_scan(src.pix, src.w, src.h, src.stride, flag, s));  // RGBA and NRGBA: alpha scan
    ScaleArgs a;
    a.dst = dst; a.dstride = dstride; a.src = src.pix; a.sstride = src.stride;
    a.dr_x0 = dr.x0; a.dr_y0 = dr.y0;
    a.adr_x0 = pr.adr.x0; a.adr_y0 = pr.adr.y0; a.adr_x1 = pr.adr.x1; a.adr_y1 = pr.adr.y1;
    a.sr_x0 = sr.x0; a.sr_y0 = sr.y0; a.ssw = sr.dx(); a.ssh = sr.dy();
    a.xscale = pr.xscale; a.yscale = pr.yscale;
    a.op = op; a.opaque_flag = op == IPX_OP_OVER ? flag : nullptr;
    a.kind = src.kind; a.cb = src.cb; a.cr = src.cr; a.cstride = src.cstride; a.ratio = src.ratio;
    a.nframes = src.nframes; a.src_fs = src.frame_stride; a.c_fs = src.c_frame_stride; a.dst_fs = dst_fs;
    IPX_HIP(launch_scale_generic(a, s));
    return IPX_OK;
}

DevSrc rgba_src(const uint8_t *p, int w, int h, int stride, int kind = IPX_SRC_RGBA)
{
    DevSrc d;
    d.kind = kind; d.pix = p; d.stride = stride; d.w = w; d.h = h;
    return d;
}

// dst / src are device pointers at pixel (0,0)
int dev_draw(hipStream_t s, uint8_t *dst, int dw, int dh, int dstride, Rect r, const uint8_t *src,
             int sw, int sh, int sstride, int spx, int spy, int op)
{
    return dev_draw_src(s, dst, dw, dh, dstride, r, rgba_src(src, sw, sh, sstride), spx, spy, op);
}

int dev_scale(hipStream_t s, int *flag, uint8_t *dst, int dw, int dh, int dstride, const Rect &dr,
              const uint8_t *src, int sw, int sh, int sstride, const Rect &sr, int op)
{
    return dev_scale_src(s, flag, dst, dw, dh, dstride, dr, rgba_src(src, sw, sh, sstride), sr, op);
}

// clip every glyph against a dw x dh frame (image/draw.clip) and cache the device table
int glyphs_for_frame(const ipx_glyphset *gs, int dw, int dh, ClippedGlyphs *out)
{
    std::lock_guard<std::mutex> lk(gs->mu);
    auto it = gs->clipped.find({dw, dh});
    if (it != gs->clipped.end()) { *out = it->second; return IPX_OK; }
    std::vector<DevGlyph> tab;
    Rect bb{0, 0, 0, 0};
    for (const GlyphHost &g : gs->g) {
        Rect r = g.dr;
        int spx = 0, spy = 0, mpx = g.mpx, mpy = g.mpy;
        if (!draw_clip(r, dw, dh, false, 0, 0, spx, spy, true, g.mw, g.mh, mpx, mpy)) continue;
        DevGlyph d;
        d.mask = gs->masks_dev + g.mask_off + (size_t)mpy * g.mw + mpx;
        d.mstride = g.mw;
        d.x0 = r.x0; d.y0 = r.y0; d.x1 = r.x1; d.y1 = r.y1;
        if (tab.empty()) bb = r;
        else {
            bb.x0 = std::min(bb.x0, r.x0); bb.y0 = std::min(bb.y0, r.y0);
            bb.x1 = std::max(bb.x1, r.x1); bb.y1 = std::max(bb.y1, r.y1);
        }
        tab.push_back(d);
    }
    ClippedGlyphs c;
    c.n = (int)tab.size();
    c.bbox = bb;
    if (c.n) {
        IPX_HIP(hipMalloc((void **)&c.dev, tab.size() * sizeof(DevGlyph)));
        IPX_HIP(hipMemcpy(c.dev, tab.data(), tab.size() * sizeof(DevGlyph), hipMemcpyHostToDevice));
    }
    gs->clipped[{dw, dh}] = c;
    *out = c;
    return IPX_OK;
}

int dev_composite(hipStream_t s, uint8_t *dst, int dw, int dh, int dstride, size_t frame_stride,
                  int nframes, const ipx_glyphset *gs)
{
    ClippedGlyphs c;
    int rc = glyphs_for_frame(gs, dw, dh, &c);
    if (rc) return rc;
    IPX_HIP(launch_composite(dst, dstride, frame_stride, nframes, c.dev, c.n, c.bbox,
                             gs->col[0] * 0x101u, gs->col[1] * 0x101u, gs->col[2] * 0x101u,
                             gs->col[3] * 0x101u, s));
    return IPX_OK;
}

}  // namespace

// =============================================================================================
extern "C" {

int ipx_device_count(void) try
{
    clear_error();
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); set_error("hipGetDeviceCount: %s", hipGetErrorString(e)); return IPX_ERR_NODEVICE; }
    return n;
}
IPX_CATCH_STATUS

int ipx_create(const ipx_config *cfg, ipx_ctx **out) try
{
    clear_error();
    if (!out) { set_error("ipx_create: null out"); return IPX_ERR_INVALID; }
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        set_error("no HIP device visible: the pixel path has no CPU fallback");
        return IPX_ERR_NODEVICE;
    }
    int dev = cfg ? cfg->device : -1;
    if (dev < 0) dev = env_int("IPX_DEVICE", env_int("LOCAL_RANK", 0) % ndev);
    if (dev >= ndev) { set_error("device %d out of range (%d visible)", dev, ndev); return IPX_ERR_INVALID; }
    IPX_HIP(hipSetDevice(dev));
    hipDeviceProp_t prop;
    IPX_HIP(hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; libipx carries gfx950 code only", dev, prop.gcnArchName);
        return IPX_ERR_NODEVICE;
    }
    ipx_ctx *c = new (std::nothrow) ipx_ctx;
    if (!c) { set_error("out of memory"); return IPX_ERR_NOMEM; }
    c->device = dev;
    c->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    int lanes = cfg && cfg->lanes > 0 ? cfg->lanes : env_int("IPX_LANES", 5);
    c->lane_bytes = cfg && cfg->lane_bytes ? cfg->lane_bytes : (size_t)64 << 20;
    c->host_cache_limit = (size_t)std::max(0, env_int("IPX_HOST_CACHE_MB", 8192)) << 20;
    c->lanes.resize(lanes);
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    for (size_t i = 0; i < c->lanes.size(); i++) {
        Lane &l = c->lanes[i];
        // the last lane is the one a batch leaves free: single-frame calls land on it, ahead of the batch's queued work
        if (e == hipSuccess)
            e = i + 1 == c->lanes.size() && c->lanes.size() >= 3 ? hipStreamCreateWithPriority(&l.stream, hipStreamNonBlocking, prio_hi)
                                                                 : hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMalloc((void **)&l.flag, sizeof(int));
    }
    if (e == hipSuccess) {
        // stream-ordered scratch (the codecs' coefficient arrays, gigabytes per batch) comes from the device's default pool; with the
        // default release threshold of 0 every synchronisation hands the freed memory back to the driver and the next batch maps it
        // again.  Keep up to IPX_POOL_KEEP_GB (64) in the pool.
        hipMemPool_t pool = nullptr;
        if (hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess && pool) {
            uint64_t keep = (uint64_t)std::max(0, env_int("IPX_POOL_KEEP_GB", 64)) << 30;
            (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
        }
        (void)hipGetLastError();
    }
    if (e == hipSuccess) e = hipMalloc((void **)&c->flat_chroma, ipx_ctx::kFlatChromaBytes);
    if (e == hipSuccess) e = hipMemset(c->flat_chroma, 128, ipx_ctx::kFlatChromaBytes);
    if (e != hipSuccess) {
        set_error("context setup failed: %s", hipGetErrorString(e));
        ipx_destroy(c);
        return IPX_ERR_HIP;
    }
    *out = c;
    return IPX_OK;
}
IPX_CATCH_STATUS

void ipx_destroy(ipx_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    for (auto &l : c->lanes) {
        if (l.stream) { (void)hipStreamSynchronize(l.stream); (void)hipStreamDestroy(l.stream); }
        if (l.dev) (void)hipFree(l.dev);
        if (l.pin) (void)hipHostFree(l.pin);
        if (l.dec) (void)hipFree(l.dec);
        if (l.flag) (void)hipFree(l.flag);
    }
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    if (c->flat_chroma) (void)hipFree(c->flat_chroma);
    for (auto &kv : c->plan_cache) ipx_plan_destroy(c, kv.second.second);     // (their glyph sets go with them)
    c->plan_cache.clear();
    for (auto &b : c->host_free_blocks) (void)hipHostFree(b.second);
    delete c;
}

// ---- memory ------------------------------------------------------------------------------------
void *ipx_host_alloc(ipx_ctx *ctx, size_t bytes)
{
    clear_error();
    if (!ctx || !bytes) { set_error("ipx_host_alloc: bad argument"); return nullptr; }
    const size_t want = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);   // 1 MiB classes
    {
        std::lock_guard<std::mutex> lk(ctx->host_mu);
        auto it = ctx->host_free_blocks.lower_bound(want);
        if (it != ctx->host_free_blocks.end() && it->first <= 2 * want + ((size_t)4 << 20)) {
            void *p = it->second;
            ctx->host_cached -= it->first;
            ctx->host_size[p] = it->first;
            ctx->host_free_blocks.erase(it);
            ctx->host_lru.erase(std::find(ctx->host_lru.begin(), ctx->host_lru.end(), p));
            return p;
        }
    }
    (void)hipSetDevice(ctx->device);
    void *p = nullptr;
    if (getenv("IPX_DEBUG")) fprintf(stderr, "[ipx] pinning %zu MiB (cache holds %zu MiB in %zu blocks)\n", want >> 20, ctx->host_cached >> 20, ctx->host_free_blocks.size());
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e != hipSuccess) { (void)hipGetLastError(); set_error("hipHostMalloc(%zu): %s", want, hipGetErrorString(e)); return nullptr; }
    std::lock_guard<std::mutex> lk(ctx->host_mu);
    ctx->host_size[p] = want;
    return p;
}

int ipx_host_free(ipx_ctx *ctx, void *p) try
{
    IPX_ENTER(ctx);
    if (!p) return IPX_OK;
    std::vector<void *> victims;     // unpinned outside the lock
    bool cached = false;
    {
        std::lock_guard<std::mutex> lk(ctx->host_mu);
        auto it = ctx->host_size.find(p);
        if (it != ctx->host_size.end()) {
            const size_t sz = it->second;
            ctx->host_size.erase(it);
            if (sz <= ctx->host_cache_limit) {
                // keep the block; make room by dropping the blocks that have gone unused the longest (a cache full of sizes nobody
                // asks for any more would otherwise make every later call pin fresh memory, ~0.2 ms per MB)
                while (ctx->host_cached + sz > ctx->host_cache_limit && !ctx->host_lru.empty()) {
                    void *old = ctx->host_lru.front();
                    ctx->host_lru.pop_front();
                    for (auto fb = ctx->host_free_blocks.begin(); fb != ctx->host_free_blocks.end(); ++fb)
                        if (fb->second == old) { ctx->host_cached -= fb->first; ctx->host_free_blocks.erase(fb); break; }
                    victims.push_back(old);
                }
                ctx->host_free_blocks.emplace(sz, p);
                ctx->host_lru.push_back(p);
                ctx->host_cached += sz;
                cached = true;
            }
        }
    }
    hipError_t e = hipSuccess;
    for (void *v : victims) { hipError_t e2 = hipHostFree(v); if (e == hipSuccess) e = e2; }
    if (!cached) { hipError_t e2 = hipHostFree(p); if (e == hipSuccess) e = e2; }
    IPX_HIP(e);
    return IPX_OK;
}
IPX_CATCH_STATUS

void *ipx_dev_alloc(ipx_ctx *ctx, size_t bytes)
{
    clear_error();
    if (!ctx || !bytes) { set_error("ipx_dev_alloc: bad argument"); return nullptr; }
    (void)hipSetDevice(ctx->device);
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) { (void)hipGetLastError(); set_error("hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); return nullptr; }
    return p;
}

int ipx_dev_free(ipx_ctx *ctx, void *p) try
{
    IPX_ENTER(ctx);
    if (p) IPX_HIP(hipFree(p));
    return IPX_OK;
}
IPX_CATCH_STATUS

int ipx_memcpy_h2d(ipx_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes) try
{
    IPX_ENTER(ctx);
    if (bytes) IPX_HIP(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
    return IPX_OK;
}
IPX_CATCH_STATUS

int ipx_memcpy_d2h(ipx_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes) try
{
    IPX_ENTER(ctx);
    if (bytes) IPX_HIP(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
    return IPX_OK;
}
IPX_CATCH_STATUS

int ipx_memcpy_d2d(ipx_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes) try
{
    IPX_ENTER(ctx);
    if (bytes) IPX_HIP(hipMemcpy(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice));
    return IPX_OK;
}
IPX_CATCH_STATUS

int ipx_stream_copy(ipx_ctx *ctx, void *stream, void *dst_dev, const void *src_dev, size_t bytes) try
{
    IPX_ENTER(ctx);
    if (((uintptr_t)dst_dev | (uintptr_t)src_dev | bytes) & 15) { set_error("ipx_stream_copy: pointers and size must be multiples of 16"); return IPX_ERR_INVALID; }
    if (bytes) IPX_HIP(launch_stream_copy(dst_dev, src_dev, bytes, stream ? (hipStream_t)stream : ctx->stream));
    return IPX_OK;
}
IPX_CATCH_STATUS

int ipx_device_sync(ipx_ctx *ctx) try
{
    IPX_ENTER(ctx);
    IPX_HIP(hipDeviceSynchronize());
    return IPX_OK;
}
IPX_CATCH_STATUS

int ipx_stream_sync(ipx_ctx *ctx, void *stream) try
{
    IPX_ENTER(ctx);
    IPX_HIP(hipStreamSynchronize(stream ? (hipStream_t)stream : ctx->stream));
    return IPX_OK;
}
IPX_CATCH_STATUS

void *ipx_stream_create(ipx_ctx *ctx)
{
    clear_error();
    if (!ctx) return nullptr;
    (void)hipSetDevice(ctx->device);
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); set_error("hipStreamCreate failed"); return nullptr; }
    return st;
}

int ipx_stream_destroy(ipx_ctx *ctx, void *stream) try
{
    IPX_ENTER(ctx);
    if (!stream) return IPX_OK;
    IPX_HIP(hipStreamSynchronize((hipStream_t)stream));
    IPX_HIP(hipStreamDestroy((hipStream_t)stream));
    return IPX_OK;
}
IPX_CATCH_STATUS

void *ipx_event_create(ipx_ctx *ctx)
{
    clear_error();
    if (!ctx) return nullptr;
    (void)hipSetDevice(ctx->device);
    hipEvent_t ev = nullptr;
    if (hipEventCreate(&ev) != hipSuccess) { (void)hipGetLastError(); set_error("hipEventCreate failed"); return nullptr; }
    return ev;
}

int ipx_event_record(ipx_ctx *ctx, void *event, void *stream) try
{
    IPX_ENTER(ctx);
    if (!event) { set_error("ipx_event_record: null event"); return IPX_ERR_INVALID; }
    IPX_HIP(hipEventRecord((hipEvent_t)event, stream ? (hipStream_t)stream : ctx->stream));
    return IPX_OK;
}
IPX_CATCH_STATUS

int ipx_event_elapsed_ms(ipx_ctx *ctx, void *start, void *stop, float *ms) try
{
    IPX_ENTER(ctx);
    if (!start || !stop || !ms) { set_error("ipx_event_elapsed_ms: bad argument"); return IPX_ERR_INVALID; }
    IPX_HIP(hipEventSynchronize((hipEvent_t)stop));
    IPX_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return IPX_OK;
}
IPX_CATCH_STATUS

void ipx_event_destroy(ipx_ctx *ctx, void *event)
{
    if (!ctx || !event) return;
    (void)hipSetDevice(ctx->device);
    (void)hipEventDestroy((hipEvent_t)event);
}

// ---- device-pointer operations ---------------------------------------------------------------------
int ipx_dev_scale_bilinear_rgba8(ipx_ctx *ctx, void *stream, uint8_t *dst, int dw, int dh,
                                 int dstride, ipx_rect dr, const uint8_t *src, int sw, int sh,
                                 int sstride, ipx_rect sr, int op) try
{
    IPX_ENTER(ctx);
    IPX_FRAME("ipx_dev_scale_bilinear_rgba8", "destination", dst, dw, dh, dstride);
    IPX_FRAME("ipx_dev_scale_bilinear_rgba8", "source", src, sw, sh, sstride);
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    int *flag = nullptr;
    if (op == IPX_OP_OVER) {
        // the opaque() flag must outlive the launch: one device int per call, freed stream-ordered
        IPX_HIP(hipMallocAsync((void **)&flag, sizeof(int), s));
    }
    int rc = dev_scale(s, flag, dst, dw, dh, dstride, to_rect(dr), src, sw, sh, sstride, to_rect(sr), op);
    if (flag) (void)hipFreeAsync(flag, s);
    return rc;
}
IPX_CATCH_STATUS

int ipx_dev_draw_rgba8(ipx_ctx *ctx, void *stream, uint8_t *dst, int dw, int dh, int dstride,
                       ipx_rect r, const uint8_t *src, int sw, int sh, int sstride, int spx, int spy,
                       int op) try
{
    IPX_ENTER(ctx);
    IPX_FRAME("ipx_dev_draw_rgba8", "destination", dst, dw, dh, dstride);
    IPX_FRAME("ipx_dev_draw_rgba8", "source", src, sw, sh, sstride);
    return dev_draw(stream ? (hipStream_t)stream : ctx->stream, dst, dw, dh, dstride, to_rect(r), src,
                    sw, sh, sstride, spx, spy, op);
}
IPX_CATCH_STATUS

int ipx_glyphset_create(ipx_ctx *ctx, const ipx_glyph *glyphs, int n, const uint8_t col[4],
                        ipx_glyphset **out) try
{
    IPX_ENTER(ctx);
    if (!out || n < 0 || (n && !glyphs) || !col) { set_error("ipx_glyphset_create: bad argument"); return IPX_ERR_INVALID; }
    if (n > kMaxGlyphs) { set_error("ipx_glyphset_create: %d glyphs exceed the limit of %d", n, kMaxGlyphs); return IPX_ERR_UNSUPPORTED; }
    *out = nullptr;
    ipx_glyphset *gs = new (std::nothrow) ipx_glyphset;
    if (!gs) { set_error("out of memory"); return IPX_ERR_NOMEM; }
    gs->device = ctx->device;
    memcpy(gs->col, col, 4);
    std::vector<uint8_t> blob;
    for (int i = 0; i < n; i++) {
        const ipx_glyph &g = glyphs[i];
        if (g.mw < 0 || g.mh < 0 || (g.mw && g.mh && (!g.mask || g.mstride < g.mw))) {
            set_error("ipx_glyphset_create: glyph %d has a bad mask", i);
            delete gs;
            return IPX_ERR_INVALID;
        }
        GlyphHost h;
        h.mask_off = blob.size();
        h.mw = g.mw; h.mh = g.mh; h.dr = to_rect(g.dr); h.mpx = g.mpx; h.mpy = g.mpy;
        for (int y = 0; y < g.mh; y++) blob.insert(blob.end(), g.mask + (size_t)y * g.mstride, g.mask + (size_t)y * g.mstride + g.mw);
        gs->g.push_back(h);
    }
    gs->masks_bytes = blob.size();
    if (!blob.empty()) {
        hipError_t e = hipMalloc((void **)&gs->masks_dev, blob.size());
        if (e == hipSuccess) e = hipMemcpy(gs->masks_dev, blob.data(), blob.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            set_error("glyph mask upload failed: %s", hipGetErrorString(e));
            ipx_glyphset_destroy(ctx, gs);
            return IPX_ERR_HIP;
        }
    }
    *out = gs;
    return IPX_OK;
}
IPX_CATCH_STATUS

void ipx_glyphset_destroy(ipx_ctx *ctx, ipx_glyphset *gs)
{
    if (!gs) return;
    (void)hipSetDevice(ctx ? ctx->device : gs->device);
    for (auto &kv : gs->clipped) if (kv.second.dev) (void)hipFree(kv.second.dev);
    if (gs->masks_dev) (void)hipFree(gs->masks_dev);
    delete gs;
}

int ipx_dev_composite_glyphs_rgba8(ipx_ctx *ctx, void *stream, uint8_t *dst, int dw, int dh,
                                   int dstride, const ipx_glyphset *gs) try
{
    IPX_ENTER(ctx);
    if (!gs) { set_error("ipx_dev_composite_glyphs_rgba8: bad argument"); return IPX_ERR_INVALID; }
    IPX_FRAME("ipx_dev_composite_glyphs_rgba8", "destination", dst, dw, dh, dstride);
    return dev_composite(stream ? (hipStream_t)stream : ctx->stream, dst, dw, dh, dstride, 0, 1, gs);
}
IPX_CATCH_STATUS

// ---- host-pointer operations: stage through a lane ------------------------------------------------
int ipx_scale_bilinear_rgba8(ipx_ctx *ctx, uint8_t *dst, int dw, int dh, int dstride, ipx_rect dr,
                             const uint8_t *src, int sw, int sh, int sstride, ipx_rect sr, int op) try
{
    IPX_ENTER(ctx);
    IPX_FRAME("ipx_scale_bilinear_rgba8", "destination", dst, dw, dh, dstride);
    IPX_FRAME("ipx_scale_bilinear_rgba8", "source", src, sw, sh, sstride);
    ScalePrep pr;
    int rc = scale_prepare(dw, dh, to_rect(dr), sw, sh, to_rect(sr), op, &pr);
    if (rc) return rc;
    if (pr.empty || dw == 0 || dh == 0) return IPX_OK;
    LaneLease lane(ctx);
    const size_t sbytes = align256((size_t)sw * sh * 4), dbytes = align256((size_t)dw * dh * 4);
    rc = lane_reserve(lane.get(), sbytes + dbytes);
    if (rc) return rc;
    uint8_t *dsrc = lane->dev, *ddst = lane->dev + sbytes;
    hipStream_t s = lane->stream;
    if (sw && sh) IPX_HIP(hipMemcpy2DAsync(dsrc, (size_t)sw * 4, src, sstride, (size_t)sw * 4, sh, hipMemcpyHostToDevice, s));
    // Over (and the Copy path's Over) reads the destination; Src leaves pixels outside adr untouched
    IPX_HIP(hipMemcpy2DAsync(ddst, (size_t)dw * 4, dst, dstride, (size_t)dw * 4, dh, hipMemcpyHostToDevice, s));
    rc = dev_scale(s, lane->flag, ddst, dw, dh, dw * 4, to_rect(dr), dsrc, sw, sh, sw * 4, to_rect(sr), op);
    if (rc) { (void)hipStreamSynchronize(s); return rc; }
    IPX_HIP(hipMemcpy2DAsync(dst, dstride, ddst, (size_t)dw * 4, (size_t)dw * 4, dh, hipMemcpyDeviceToHost, s));
    IPX_HIP(hipStreamSynchronize(s));
    return IPX_OK;
}
IPX_CATCH_STATUS

int ipx_draw_rgba8(ipx_ctx *ctx, uint8_t *dst, int dw, int dh, int dstride, ipx_rect r,
                   const uint8_t *src, int sw, int sh, int sstride, int spx, int spy, int op) try
{
    IPX_ENTER(ctx);
    IPX_FRAME("ipx_draw_rgba8", "destination", dst, dw, dh, dstride);
    IPX_FRAME("ipx_draw_rgba8", "source", src, sw, sh, sstride);
    if (op != IPX_OP_OVER && op != IPX_OP_SRC) { set_error("draw: unknown op %d", op); return IPX_ERR_INVALID; }
    if (!dw || !dh || !sw || !sh) return IPX_OK;
    LaneLease lane(ctx);
    const size_t sbytes = align256((size_t)sw * sh * 4), dbytes = align256((size_t)dw * dh * 4);
    int rc = lane_reserve(lane.get(), sbytes + dbytes);
    if (rc) return rc;
    uint8_t *dsrc = lane->dev, *ddst = lane->dev + sbytes;
    hipStream_t s = lane->stream;
    IPX_HIP(hipMemcpy2DAsync(dsrc, (size_t)sw * 4, src, sstride, (size_t)sw * 4, sh, hipMemcpyHostToDevice, s));
    IPX_HIP(hipMemcpy2DAsync(ddst, (size_t)dw * 4, dst, dstride, (size_t)dw * 4, dh, hipMemcpyHostToDevice, s));
    rc = dev_draw(s, ddst, dw, dh, dw * 4, to_rect(r), dsrc, sw, sh, sw * 4, spx, spy, op);
    if (rc) { (void)hipStreamSynchronize(s); return rc; }
    IPX_HIP(hipMemcpy2DAsync(dst, dstride, ddst, (size_t)dw * 4, (size_t)dw * 4, dh, hipMemcpyDeviceToHost, s));
    IPX_HIP(hipStreamSynchronize(s));
    return IPX_OK;
}
IPX_CATCH_STATUS

int ipx_composite_glyphs_rgba8(ipx_ctx *ctx, uint8_t *dst, int dw, int dh, int dstride,
                               const ipx_glyph *glyphs, int n, const uint8_t col[4]) try
{
    IPX_ENTER(ctx);
    if (n < 0 || (n && !glyphs) || !col) { set_error("ipx_composite_glyphs_rgba8: bad argument"); return IPX_ERR_INVALID; }
    IPX_FRAME("ipx_composite_glyphs_rgba8", "destination", dst, dw, dh, dstride);
    if (!n || !dw || !dh) return IPX_OK;
    ipx_glyphset *gs = nullptr;
    int rc = ipx_glyphset_create(ctx, glyphs, n, col, &gs);
    if (rc) return rc;
    {
        LaneLease lane(ctx);
        rc = lane_reserve(lane.get(), align256((size_t)dw * dh * 4));
        hipStream_t s = lane->stream;
        hipError_t e = hipSuccess;
        if (!rc) {
            e = hipMemcpy2DAsync(lane->dev, (size_t)dw * 4, dst, dstride, (size_t)dw * 4, dh, hipMemcpyHostToDevice, s);
            if (e == hipSuccess) rc = dev_composite(s, lane->dev, dw, dh, dw * 4, 0, 1, gs);
            if (e == hipSuccess && !rc)
                e = hipMemcpy2DAsync(dst, dstride, lane->dev, (size_t)dw * 4, (size_t)dw * 4, dh, hipMemcpyDeviceToHost, s);
            hipError_t e2 = hipStreamSynchronize(s);
            if (e == hipSuccess) e = e2;
            if (e != hipSuccess && !rc) { set_error("composite staging failed: %s", hipGetErrorString(e)); rc = IPX_ERR_HIP; }
        }
    }
    ipx_glyphset_destroy(ctx, gs);
    return rc;
}
IPX_CATCH_STATUS

}  // extern "C"

// ---- source-type variants: host pointers, staged through a lane ----------------------------------------
namespace {

// dst goes up and down; the source planes go up; `run` launches on the lane's stream
template <typename F>
int stage_and_run(ipx_ctx *ctx, uint8_t *dst, int dw, int dh, int dstride, size_t src_bytes, F &&upload_and_run)
{
    LaneLease lane(ctx);
    const size_t dbytes = align256((size_t)dw * dh * 4);
    int rc = lane_reserve(lane.get(), dbytes + src_bytes + 1024);
    if (rc) return rc;
    uint8_t *ddst = lane->dev, *dsrc = lane->dev + dbytes;
    hipStream_t s = lane->stream;
    IPX_HIP(hipMemcpy2DAsync(ddst, (size_t)dw * 4, dst, dstride, (size_t)dw * 4, dh, hipMemcpyHostToDevice, s));
    rc = upload_and_run(s, lane->flag, ddst, dsrc);
    if (rc) { (void)hipStreamSynchronize(s); return rc; }
    IPX_HIP(hipMemcpy2DAsync(dst, dstride, ddst, (size_t)dw * 4, (size_t)dw * 4, dh, hipMemcpyDeviceToHost, s));
    IPX_HIP(hipStreamSynchronize(s));
    return IPX_OK;
}

bool ycbcr_ok(const ipx_ycbcr *y, int *cw, int *ch)
{
    if (!y || !y->y || !y->cb || !y->cr || y->w <= 0 || y->h <= 0 || y->ratio < 0 || y->ratio > IPX_YCBCR_440) return false;
    *cw = (y->ratio == IPX_YCBCR_422 || y->ratio == IPX_YCBCR_420) ? (y->w + 1) / 2 : y->w;   // image.NewYCbCr
    *ch = (y->ratio == IPX_YCBCR_420 || y->ratio == IPX_YCBCR_440) ? (y->h + 1) / 2 : y->h;
    return y->ystride >= y->w && y->cstride >= *cw;
}

int upload_ycbcr(hipStream_t s, const ipx_ycbcr *y, int cw, int ch, uint8_t *dsrc, DevSrc *out)
{
    const size_t yb = align256((size_t)y->w * y->h), cbytes = align256((size_t)cw * ch);
    IPX_HIP(hipMemcpy2DAsync(dsrc, y->w, y->y, y->ystride, y->w, y->h, hipMemcpyHostToDevice, s));
    IPX_HIP(hipMemcpy2DAsync(dsrc + yb, cw, y->cb, y->cstride, cw, ch, hipMemcpyHostToDevice, s));
    IPX_HIP(hipMemcpy2DAsync(dsrc + yb + cbytes, cw, y->cr, y->cstride, cw, ch, hipMemcpyHostToDevice, s));
    out->kind = IPX_SRC_YCBCR; out->pix = dsrc; out->stride = y->w;
    out->cb = dsrc + yb; out->cr = dsrc + yb + cbytes; out->cstride = cw; out->ratio = y->ratio;
    out->w = y->w; out->h = y->h;
    return IPX_OK;
}

}  // namespace

extern "C" {

int ipx_scale_bilinear_nrgba8(ipx_ctx *ctx, uint8_t *dst, int dw, int dh, int dstride, ipx_rect dr,
                              const uint8_t *src, int sw, int sh, int sstride, ipx_rect sr, int op) try
{
    IPX_ENTER(ctx);
    IPX_FRAME("ipx_scale_bilinear_nrgba8", "destination", dst, dw, dh, dstride);
    IPX_FRAME("ipx_scale_bilinear_nrgba8", "source", src, sw, sh, sstride);
    if (!dw || !dh || !sw || !sh) return IPX_OK;
    return stage_and_run(ctx, dst, dw, dh, dstride, align256((size_t)sw * sh * 4), [&](hipStream_t s, int *flag, uint8_t *ddst, uint8_t *dsrc) -> int {
        IPX_HIP(hipMemcpy2DAsync(dsrc, (size_t)sw * 4, src, sstride, (size_t)sw * 4, sh, hipMemcpyHostToDevice, s));
        return dev_scale_src(s, flag, ddst, dw, dh, dw * 4, to_rect(dr), rgba_src(dsrc, sw, sh, sw * 4, IPX_SRC_NRGBA), to_rect(sr), op);
    });
}
IPX_CATCH_STATUS

int ipx_draw_nrgba8(ipx_ctx *ctx, uint8_t *dst, int dw, int dh, int dstride, ipx_rect r, const uint8_t *src,
                    int sw, int sh, int sstride, int spx, int spy, int op) try
{
    IPX_ENTER(ctx);
    IPX_FRAME("ipx_draw_nrgba8", "destination", dst, dw, dh, dstride);
    IPX_FRAME("ipx_draw_nrgba8", "source", src, sw, sh, sstride);
    if (!dw || !dh || !sw || !sh) return IPX_OK;
    return stage_and_run(ctx, dst, dw, dh, dstride, align256((size_t)sw * sh * 4), [&](hipStream_t s, int *, uint8_t *ddst, uint8_t *dsrc) -> int {
        IPX_HIP(hipMemcpy2DAsync(dsrc, (size_t)sw * 4, src, sstride, (size_t)sw * 4, sh, hipMemcpyHostToDevice, s));
        return dev_draw_src(s, ddst, dw, dh, dw * 4, to_rect(r), rgba_src(dsrc, sw, sh, sw * 4, IPX_SRC_NRGBA), spx, spy, op);
    });
}
IPX_CATCH_STATUS

// a deep frame from host memory -> frame of taps in the lane's scratch (the Pix copy sits behind it)
static int upload_deep(hipStream_t s, const uint8_t *src, int sw, int sh, int sstride, int kind, uint8_t *dsrc, DevSrc *out)
{
    const int bpp = kind == IPX_DEEP_GRAY16 ? 2 : (kind == IPX_DEEP_CMYK ? 4 : 8);
    uint8_t *pix = dsrc + align256((size_t)sw * sh * 8);
    IPX_HIP(hipMemcpy2DAsync(pix, (size_t)sw * bpp, src, sstride, (size_t)sw * bpp, sh, hipMemcpyHostToDevice, s));
    IPX_HIP(launch_deep_expand(dsrc, 0, pix, sw * bpp, 0, kind, sw, sh, 1, s));
    out->kind = IPX_SRC_TAP64; out->pix = dsrc; out->stride = sw * 8; out->w = sw; out->h = sh;
    return IPX_OK;
}
static int deep_args_status(const char *who, const void *dst, int dw, int dh, int dstride, const void *src, int sw, int sh, int sstride, int kind)
{
    if (kind != IPX_DEEP_NRGBA64 && kind != IPX_DEEP_RGBA64 && kind != IPX_DEEP_GRAY16 && kind != IPX_DEEP_CMYK) {
        set_error("%s: unknown source type %d", who, kind);
        return IPX_ERR_INVALID;
    }
    int rc = frame_status(who, "destination", dst, dw, dh, dstride);
    if (!rc) rc = frame_status(who, "source", src, sw, sh, sstride, kind == IPX_DEEP_GRAY16 ? 2 : (kind == IPX_DEEP_CMYK ? 4 : 8));
    if (!rc) rc = frame_status(who, "source", src, sw, sh, (long long)sw * 8, 8);       // the frame of taps
    return rc;
}

int ipx_scale_bilinear_deep(ipx_ctx *ctx, uint8_t *dst, int dw, int dh, int dstride, ipx_rect dr, const uint8_t *src, int sw, int sh,
                            int sstride, int kind, ipx_rect sr, int op) try
{
    IPX_ENTER(ctx);
    const int rc = deep_args_status("ipx_scale_bilinear_deep", dst, dw, dh, dstride, src, sw, sh, sstride, kind);
    if (rc) return rc;
    if (!dw || !dh || !sw || !sh) return IPX_OK;
    return stage_and_run(ctx, dst, dw, dh, dstride, align256((size_t)sw * sh * 8) * 2, [&](hipStream_t s, int *flag, uint8_t *ddst, uint8_t *dsrc) -> int {
        DevSrc t;
        const int r2 = upload_deep(s, src, sw, sh, sstride, kind, dsrc, &t);
        return r2 ? r2 : dev_scale_src(s, flag, ddst, dw, dh, dw * 4, to_rect(dr), t, to_rect(sr), op);
    });
}
IPX_CATCH_STATUS

int ipx_draw_deep(ipx_ctx *ctx, uint8_t *dst, int dw, int dh, int dstride, ipx_rect r, const uint8_t *src, int sw, int sh, int sstride,
                  int kind, int spx, int spy, int op) try
{
    IPX_ENTER(ctx);
    const int rc = deep_args_status("ipx_draw_deep", dst, dw, dh, dstride, src, sw, sh, sstride, kind);
    if (rc) return rc;
    if (!dw || !dh || !sw || !sh) return IPX_OK;
    return stage_and_run(ctx, dst, dw, dh, dstride, align256((size_t)sw * sh * 8) * 2, [&](hipStream_t s, int *, uint8_t *ddst, uint8_t *dsrc) -> int {
        DevSrc t;
        const int r2 = upload_deep(s, src, sw, sh, sstride, kind, dsrc, &t);
        return r2 ? r2 : dev_draw_src(s, ddst, dw, dh, dw * 4, to_rect(r), t, spx, spy, op);
    });
}
IPX_CATCH_STATUS

int ipx_scale_bilinear_ycbcr(ipx_ctx *ctx, uint8_t *dst, int dw, int dh, int dstride, ipx_rect dr,
                             const ipx_ycbcr *src, ipx_rect sr) try
{
    IPX_ENTER(ctx);
    int cw = 0, ch = 0;
    IPX_FRAME("ipx_scale_bilinear_ycbcr", "destination", dst, dw, dh, dstride);
    if (!ycbcr_ok(src, &cw, &ch)) { set_error("ipx_scale_bilinear_ycbcr: bad arguments"); return IPX_ERR_INVALID; }
    if (!frame_span_ok(src->w, src->h, src->ystride, 1)) { set_error("ipx_scale_bilinear_ycbcr: source planes beyond the addressable span"); return IPX_ERR_UNSUPPORTED; }
    if (!dw || !dh) return IPX_OK;
    const size_t bytes = align256((size_t)src->w * src->h) + 2 * align256((size_t)cw * ch);
    return stage_and_run(ctx, dst, dw, dh, dstride, bytes, [&](hipStream_t s, int *flag, uint8_t *ddst, uint8_t *dsrc) -> int {
        DevSrc d;
        int rc = upload_ycbcr(s, src, cw, ch, dsrc, &d);
        if (rc) return rc;
        return dev_scale_src(s, flag, ddst, dw, dh, dw * 4, to_rect(dr), d, to_rect(sr), IPX_OP_SRC);
    });
}
IPX_CATCH_STATUS

int ipx_draw_ycbcr(ipx_ctx *ctx, uint8_t *dst, int dw, int dh, int dstride, ipx_rect r, const ipx_ycbcr *src,
                   int spx, int spy) try
{
    IPX_ENTER(ctx);
    int cw = 0, ch = 0;
    IPX_FRAME("ipx_draw_ycbcr", "destination", dst, dw, dh, dstride);
    if (!ycbcr_ok(src, &cw, &ch)) { set_error("ipx_draw_ycbcr: bad arguments"); return IPX_ERR_INVALID; }
    if (!frame_span_ok(src->w, src->h, src->ystride, 1)) { set_error("ipx_draw_ycbcr: source planes beyond the addressable span"); return IPX_ERR_UNSUPPORTED; }
    if (!dw || !dh) return IPX_OK;
    const size_t bytes = align256((size_t)src->w * src->h) + 2 * align256((size_t)cw * ch);
    return stage_and_run(ctx, dst, dw, dh, dstride, bytes, [&](hipStream_t s, int *, uint8_t *ddst, uint8_t *dsrc) -> int {
        DevSrc d;
        int rc = upload_ycbcr(s, src, cw, ch, dsrc, &d);
        if (rc) return rc;
        return dev_draw_src(s, ddst, dw, dh, dw * 4, to_rect(r), d, spx, spy, IPX_OP_SRC);
    });
}
IPX_CATCH_STATUS

}  // extern "C"

extern "C" {

// ---- plans -------------------------------------------------------------------------------------------
int ipx_plan_create(ipx_ctx *ctx, const ipx_plan_params *p, ipx_plan **out) try
{
    IPX_ENTER(ctx);
    if (!p || !out) { set_error("ipx_plan_create: bad argument"); return IPX_ERR_INVALID; }
    *out = nullptr;
    if (p->sw <= 0 || p->sh <= 0) { set_error("ipx_plan_create: frame size %dx%d", p->sw, p->sh); return IPX_ERR_INVALID; }
    if (!frame_span_ok(p->sw, p->sh, (long long)p->sw * 4, 4)) {
        set_error("ipx_plan_create: a %dx%d frame is beyond the 2 GiB / 65535-pixel span the kernels address", p->sw, p->sh);
        return IPX_ERR_UNSUPPORTED;
    }
    ipx_plan *pl = new (std::nothrow) ipx_plan;
    if (!pl) { set_error("out of memory"); return IPX_ERR_NOMEM; }
    pl->p = *p;
    int rc = IPX_OK;
    const int sw = p->sw, sh = p->sh;
    if (p->do_resize) {
        int nw, nh;
        rc = ipx_resize_dims(sw, sh, p->resize_w, p->resize_h, p->keep_aspect, &nw, &nh);
        if (rc) { delete pl; return rc; }
        if (!frame_span_ok(nw, nh, (long long)nw * 4, 4)) {
            set_error("ipx_plan_create: a %dx%d resize output is beyond the 2 GiB / 65535-pixel span the kernels address", nw, nh);
            delete pl;
            return IPX_ERR_UNSUPPORTED;
        }
        pl->sc[0].on = true; pl->sc[0].dw = nw; pl->sc[0].dh = nh; pl->sc[0].sr = Rect{0, 0, sw, sh};
        pl->info.resize_w = nw; pl->info.resize_h = nh;
        pl->info.resize_bytes = (size_t)nw * nh * 4;
    }
    if (p->do_thumbnail) {
        int nw, nh;
        ipx_rect crop;
        rc = ipx_thumb_geometry(sw, sh, p->thumb_size, p->crop_to_fit, &crop, &nw, &nh);
        if (rc) { delete pl; return rc; }
        if (!frame_span_ok(nw, nh, (long long)nw * 4, 4)) {
            set_error("ipx_plan_create: a %dx%d thumbnail is beyond the 2 GiB / 65535-pixel span the kernels address", nw, nh);
            delete pl;
            return IPX_ERR_UNSUPPORTED;
        }
        pl->sc[1].on = true; pl->sc[1].dw = nw; pl->sc[1].dh = nh; pl->sc[1].sr = to_rect(crop);
        pl->info.thumb_w = nw; pl->info.thumb_h = nh; pl->info.thumb_crop = crop;
        pl->info.thumb_bytes = (size_t)nw * nh * 4;
    }
    if (p->do_watermark) {
        pl->info.wm_w = sw; pl->info.wm_h = sh;
        pl->info.wm_bytes = (size_t)sw * sh * 4;
        if (p->glyphs) {
            rc = glyphs_for_frame(p->glyphs, sw, sh, &pl->glyphs);
            if (rc) { delete pl; return rc; }
        }
    }
    pl->info.algorithmic_bytes = (size_t)sw * sh * 4 + pl->info.resize_bytes + pl->info.thumb_bytes + pl->info.wm_bytes;

    // The band kernel needs a tap pair per axis (source extents >= 2) and non-empty outputs; an
    // output with a zero dimension (resize.go:70-72 has no guard) is simply empty.
    pl->fused = true;
    for (auto &s : pl->sc)
        if (s.on && s.dw > 0 && s.dh > 0 && (s.sr.dx() < 2 || s.sr.dy() < 2)) pl->fused = false;
    if (env_int("IPX_NO_FUSE", 0)) pl->fused = false;
    if (!pl->fused) { *out = pl; return IPX_OK; }

    // taps per axis, with the reference's float64 arithmetic; dyadic axes get the exact fp32 path
    std::vector<AxisTap> xt[2], yt[2];
    for (int k = 0; k < 2; k++) {
        PlanScale &s = pl->sc[k];
        s.dyadic_shift = -1;
        if (!s.on || s.dw <= 0 || s.dh <= 0) continue;
        xt[k].resize(s.dw); yt[k].resize(s.dh);
        build_axis_taps(s.sr.dx(), s.dw, 0, s.dw, xt[k].data());
        build_axis_taps(s.sr.dy(), s.dh, 0, s.dh, yt[k].data());
        if (!env_int("IPX_NO_DYADIC", 0)) {
            const int kx = axis_dyadic_bits(xt[k].data(), s.dw, 12, 1);   // kx >= 1 keeps kx + ky >= 1 for the packed-integer lerp
            const int ky = axis_dyadic_bits(yt[k].data(), s.dh, 12);
            if (kx >= 0 && ky >= 0 && kx + ky <= 16) { s.dyadic_shift = kx + ky; s.kx = kx; s.ky = ky; }  // 8 + kx + ky <= 24 bits
        }
    }

    // Tilings of the source frame (PlanGeom) and the tables that depend on them.  A tiling = owned columns per workgroup (multiple of
    // 4 pixels = 16 B) and owned rows; a column block may hold at most out_cols destination columns of any scaled output.
    //   pl->g     tiles of one dword per pixel: band_pipe_kernel (RGBA sources) and band_nrgba_kernel; 72 KB of LDS, two workgroups per CU
    //   pl->conv  tiles of two dwords per pixel (converted taps): band_conv_kernel; at most 1020 columns x 8 rows, again two per CU
    struct HostGeom {
        int bc = 0, br = 0;
        std::vector<int> rb[2], cbv[2];
        std::vector<uint32_t> yr[2], y16[2];
        size_t off_rb[2] = {0, 0}, off_cb[2] = {0, 0}, off_yr[2] = {0, 0}, off_y16[2] = {0, 0};
    };
    auto build_geom = [&](PlanGeom &g, HostGeom &hg, int max_cols, size_t lds_budget, int px_bytes, int rows_cap, int out_cols) -> bool {
        for (;;) {
            const int ncb = (sw + max_cols - 1) / max_cols;
            const int bc = std::max(4, ((sw + ncb - 1) / ncb + 3) & ~3);
            int br = (int)(lds_budget / ((size_t)(bc + 4) * px_bytes)) - 1;
            br = std::max(1, std::min(br, rows_cap > 0 ? rows_cap : env_int("IPX_BAND_ROWS_MAX", bc / 4 + 1 > 256 ? 8 : 16)));  // shapes of band_pipe_shape
            if (rows_cap <= 0 && env_int("IPX_BAND_ROWS", 0) > 0) br = env_int("IPX_BAND_ROWS", 0);
            br = std::min(br, sh);
            if (rows_cap > 0 && br > 1) br &= ~1;              // whole chroma rows per band (4:2:0, 4:4:0)
            hg.bc = bc; hg.br = br;
            g.blk_cols = bc; g.band_rows = br;
            g.ncolblk = (sw + bc - 1) / bc;
            g.nbands = (sh + br - 1) / br;
            int widest = 0;
            g.most_rows = 0;
            g.nx_out[0] = g.nx_out[1] = 0;
            for (int k = 0; k < 2; k++) {
                PlanScale &sk = pl->sc[k];
                if (xt[k].empty()) continue;
                hg.rb[k].assign(g.nbands + 1, 0); hg.cbv[k].assign(g.ncolblk + 1, 0);
                int d = 0;
                for (int b2 = 0; b2 <= g.nbands; b2++) {  // first output row whose tap pair starts in band b2 or below
                    while (d < sk.dh && sk.sr.y0 + yt[k][d].base < b2 * br) d++;
                    hg.rb[k][b2] = b2 == g.nbands ? sk.dh : d;
                }
                d = 0;
                for (int c = 0; c <= g.ncolblk; c++) {
                    while (d < sk.dw && sk.sr.x0 + xt[k][d].base < c * bc) d++;
                    hg.cbv[k][c] = c == g.ncolblk ? sk.dw : d;
                }
                int wk = 0;
                for (int c = 0; c < g.ncolblk; c++) wk = std::max(wk, hg.cbv[k][c + 1] - hg.cbv[k][c]);
                widest = std::max(widest, wk);
                g.nx_out[k] = (wk + 255) / 256;
                for (int b2 = 0; b2 < g.nbands; b2++) g.most_rows = std::max(g.most_rows, hg.rb[k][b2 + 1] - hg.rb[k][b2]);
            }
            if (widest <= out_cols) return true;
            if (bc <= 4) return false;                         // enormous upscale: per-operation kernels
            max_cols = std::max(4, (int)((long long)bc * out_cols / widest) & ~3);
            if (max_cols >= bc) max_cols = bc - 4;
        }
    };
    HostGeom hg[2];
    if (!build_geom(pl->g, hg[0], std::max(4, env_int("IPX_BLK_COLS", 2044)) & ~3,
                    ((size_t)std::max(8, env_int("IPX_LDS_KB", 72)) << 10) - band_lds_bytes(-1, -4), 4, 0, 256 * kBandNX)) {
        pl->fused = false;
        *out = pl;
        return IPX_OK;
    }
    pl->g.ok = true;
    pl->conv.ok = build_geom(pl->conv, hg[1], std::max(4, std::min(env_int("IPX_CONV_BLK_COLS", 1020), 1020)) & ~3, (size_t)80 << 10, 8, 8, 512);

    // host tables -> one device blob
    std::vector<uint8_t> blob;
    auto put = [&](const void *src, size_t bytes) {
        const size_t off = (blob.size() + 15) & ~(size_t)15;
        blob.resize(off + bytes);
        memcpy(blob.data() + off, src, bytes);
        return off;
    };
    size_t off_xt[2] = {0, 0}, off_yt[2] = {0, 0};
    for (int k = 0; k < 2; k++) {
        if (xt[k].empty()) continue;
        off_xt[k] = put(xt[k].data(), xt[k].size() * sizeof(AxisTap));
        off_yt[k] = put(yt[k].data(), yt[k].size() * sizeof(AxisTap));
        PlanScale &sk = pl->sc[k];
        // the packed-integer lerp's row table (ScaleOut::yrow): RGBA8 taps, kx <= 8 (16-bit lanes hold 8 + kx bits), ky <= 12
        const bool int_rows = !(sk.dyadic_shift < 1 || sk.kx > 8 || sk.ky > 12 || env_int("IPX_NO_INTLERP", 0));
        const int kk = std::max(sk.dyadic_shift, 9), ysh = kk - sk.dyadic_shift;   // ky + ysh <= 15: the scaled weights stay 16-bit lanes
        if (int_rows) sk.imul = 257u << (24 - kk);
        for (int gi = 0; gi < 2; gi++) {
            PlanGeom &g = gi ? pl->conv : pl->g;
            HostGeom &h = hg[gi];
            if (!g.ok) continue;
            h.off_rb[k] = put(h.rb[k].data(), h.rb[k].size() * sizeof(int));
            h.off_cb[k] = put(h.cbv[k].data(), h.cbv[k].size() * sizeof(int));
            if (!int_rows) continue;
            const uint32_t pitch = gi ? (uint32_t)kConvTilePitch : (uint32_t)(h.bc + 4) * 4;      // of one plane of the tile
            h.yr[k].assign((size_t)(sk.dh + 1) * 2, 0);
            for (int d = 0; d < sk.dh; d++) {
                const int row = sk.sr.y0 + yt[k][d].base, band = row / h.br;
                uint32_t code = 2;                                   // the first row of a band finds nothing at hand
                if (d > h.rb[k][band]) {
                    const int step = yt[k][d].base - yt[k][d - 1].base;
                    code = step == 0 ? 0 : step == 1 ? 1 : 2;
                }
                h.yr[k][2 * d] = (uint32_t)(row - band * h.br) * pitch | code << 28;
                h.yr[k][2 * d + 1] = yt[k][d].iw << ysh;
            }
            h.off_yr[k] = put(h.yr[k].data(), h.yr[k].size() * sizeof(uint32_t));
            if (sk.kx <= 8 && sk.ky <= 8) {   // the same walk for 16-bit converted taps (ScaleOut::yrow16)
                h.y16[k].assign((size_t)(sk.dh + 1) * 4, 0);
                for (int d = 0; d < sk.dh; d++) {
                    h.y16[k][4 * d] = h.yr[k][2 * d];
                    h.y16[k][4 * d + 1] = (yt[k][d].iw & 0xffffu) << (16 - sk.dyadic_shift);
                    h.y16[k][4 * d + 2] = (yt[k][d].iw >> 16) << (16 - sk.dyadic_shift);
                }
                h.off_y16[k] = put(h.y16[k].data(), h.y16[k].size() * sizeof(uint32_t));
            }
        }
    }
    if (!blob.empty()) {
        hipError_t e = hipMalloc((void **)&pl->blob, blob.size());
        if (e == hipSuccess) e = hipMemcpy(pl->blob, blob.data(), blob.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            set_error("plan table upload failed: %s", hipGetErrorString(e));
            ipx_plan_destroy(ctx, pl);
            return IPX_ERR_HIP;
        }
        for (int k = 0; k < 2; k++) {
            PlanScale &sk = pl->sc[k];
            if (!sk.on || sk.dw <= 0 || sk.dh <= 0) continue;
            sk.xt = (AxisTap *)(pl->blob + off_xt[k]);
            sk.yt = (AxisTap *)(pl->blob + off_yt[k]);
            for (int gi = 0; gi < 2; gi++) {
                PlanGeom &g = gi ? pl->conv : pl->g;
                const HostGeom &h = hg[gi];
                if (!g.ok) continue;
                g.row_begin[k] = (int *)(pl->blob + h.off_rb[k]);
                g.col_begin[k] = (int *)(pl->blob + h.off_cb[k]);
                if (!h.yr[k].empty()) g.yrow[k] = (uint32_t *)(pl->blob + h.off_yr[k]);
                if (!h.y16[k].empty()) g.yrow16[k] = (uint32_t *)(pl->blob + h.off_y16[k]);
            }
        }
    }
    *out = pl;
    return IPX_OK;
}
IPX_CATCH_STATUS

void ipx_plan_destroy(ipx_ctx *ctx, ipx_plan *plan)
{
    if (!plan) return;
    if (ctx) (void)hipSetDevice(ctx->device);
    if (plan->blob) (void)hipFree(plan->blob);
    if (plan->thumb_only) ipx_plan_destroy(ctx, plan->thumb_only);
    if (plan->owned_gs) ipx_glyphset_destroy(ctx, plan->owned_gs);
    delete plan;
}

// ---- plans by content --------------------------------------------------------------------------------------------------------
// The per-operator entries (ipx_*_process, ipx_processor_process) and the pool describe their operators per call; building a glyph set
// and a plan per call cost three hipMalloc / hipFree pairs, each of which waits for every stream of the device.  The context keeps
// plans (with their glyph sets) by content instead: operator parameters, colour, and every glyph's rectangle and mask bytes.
static constexpr size_t kMaxCachedPlans = 256;

static int ops_key(const ipx_pool_ops &in, std::string *key)
{
    if (in.n_glyphs < 0 || (in.n_glyphs && !in.glyphs)) { set_error("bad glyph list"); return IPX_ERR_INVALID; }
    key->assign((const char *)&in, offsetof(ipx_pool_ops, glyphs));
    key->append((const char *)in.col, 4);
    for (int i = 0; i < in.n_glyphs; i++) {
        const ipx_glyph &g = in.glyphs[i];
        if (g.mw < 0 || g.mh < 0 || (g.mw && g.mh && (!g.mask || g.mstride < g.mw))) { set_error("glyph %d has a bad mask", i); return IPX_ERR_INVALID; }
        key->append((const char *)&g.mw, sizeof(int32_t) * 2);
        key->append((const char *)&g.dr, sizeof g.dr);
        key->append((const char *)&g.mpx, sizeof(int32_t) * 2);
        for (int y = 0; y < g.mh; y++) key->append((const char *)g.mask + (size_t)y * g.mstride, (size_t)g.mw);
    }
    return IPX_OK;
}

int ipx_plan_acquire(ipx_ctx *ctx, const ipx_pool_ops *ops, ipx_plan **plan, int *cached) try
{
    IPX_ENTER(ctx);
    if (!ops || !plan || !cached) { set_error("ipx_plan_acquire: bad argument"); return IPX_ERR_INVALID; }
    *plan = nullptr; *cached = 0;
    std::string key;
    int rc = ops_key(*ops, &key);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(ctx->plan_mu);      // (a miss builds under the lock: two callers with the same new content build once)
    const bool use_cache = env_int("IPX_PLAN_CACHE", 1) != 0;     // 0: a plan per call, as before the cache existed (tools/bench_seam.py)
    auto it = ctx->plan_cache.find(key);
    if (use_cache && it != ctx->plan_cache.end()) { *plan = it->second.second; *cached = 1; return IPX_OK; }
    ipx_glyphset *gs = nullptr;
    if (ops->do_watermark && ops->n_glyphs > 0) {
        rc = ipx_glyphset_create(ctx, ops->glyphs, ops->n_glyphs, ops->col, &gs);
        if (rc) return rc;
    }
    ipx_plan_params pp;
    memset(&pp, 0, sizeof pp);
    pp.sw = ops->sw; pp.sh = ops->sh;
    pp.do_resize = ops->do_resize; pp.resize_w = ops->resize_w; pp.resize_h = ops->resize_h; pp.keep_aspect = ops->keep_aspect;
    pp.do_thumbnail = ops->do_thumbnail; pp.thumb_size = ops->thumb_size; pp.crop_to_fit = ops->crop_to_fit;
    pp.do_watermark = ops->do_watermark; pp.glyphs = gs;
    ipx_plan *pl = nullptr;
    rc = ipx_plan_create(ctx, &pp, &pl);
    if (rc) { if (gs) ipx_glyphset_destroy(ctx, gs); return rc; }
    pl->owned_gs = gs;
    if (use_cache && ctx->plan_cache.size() < kMaxCachedPlans) { ctx->plan_cache.emplace(std::move(key), std::make_pair(gs, pl)); *cached = 1; }
    *plan = pl;
    return IPX_OK;
}
IPX_CATCH_STATUS

void ipx_plan_release(ipx_ctx *ctx, ipx_plan *plan, int cached)
{
    if (plan && !cached) ipx_plan_destroy(ctx, plan);   // a plan the full cache did not take lives for one call
}

int ipx_plan_query(const ipx_plan *plan, ipx_plan_info *info) try
{
    clear_error();
    if (!plan || !info) { set_error("ipx_plan_query: bad argument"); return IPX_ERR_INVALID; }
    *info = plan->info;
    return IPX_OK;
}
IPX_CATCH_STATUS

// the source rows of a batch entry: stride given by the caller, size by the plan
static int plan_src_status(const char *who, const ipx_plan *pl, long long stride, int bpp)
{
    if (frame_span_ok(pl->p.sw, pl->p.sh, stride, bpp)) return IPX_OK;
    set_error("%s: %dx%d frames with a row stride of %lld bytes are beyond the 2 GiB span the kernels address", who, pl->p.sw, pl->p.sh, stride);
    return IPX_ERR_UNSUPPORTED;
}
#define IPX_PLAN_SRC(who, pl, stride, bpp) do { const int rc_ = plan_src_status(who, pl, stride, bpp); if (rc_) return rc_; } while (0)

static bool glyphs_separate(bool fused_kernel);

int ipx_plan_run_dev(ipx_ctx *ctx, void *stream, const ipx_plan *pl, int n, const uint8_t *src,
                     int sstride, size_t src_frame_stride, uint8_t *resize_out,
                     size_t resize_frame_stride, uint8_t *thumb_out, size_t thumb_frame_stride,
                     uint8_t *wm_out, size_t wm_frame_stride) try
{
    IPX_ENTER(ctx);
    if (!pl || n < 0 || !src || (long long)sstride < (long long)pl->p.sw * 4) {
        set_error("ipx_plan_run_dev: bad argument");
        return IPX_ERR_INVALID;
    }
    IPX_PLAN_SRC("ipx_plan_run_dev", pl, sstride, 4);
    if (n == 0) return IPX_OK;
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    const int sw = pl->p.sw, sh = pl->p.sh;
    uint8_t *outs[2] = {pl->sc[0].on ? resize_out : nullptr, pl->sc[1].on ? thumb_out : nullptr};
    const size_t ostr[2] = {resize_frame_stride, thumb_frame_stride};
    uint8_t *wm = pl->p.do_watermark ? wm_out : nullptr;
    const uint32_t col[4] = {pl->p.glyphs ? pl->p.glyphs->col[0] * 0x101u : 0, pl->p.glyphs ? pl->p.glyphs->col[1] * 0x101u : 0,
                             pl->p.glyphs ? pl->p.glyphs->col[2] * 0x101u : 0, pl->p.glyphs ? pl->p.glyphs->col[3] * 0x101u : 0};

    if (pl->fused) {
        BandArgs a{};
        a.src = src; a.src_frame_stride = src_frame_stride; a.sstride = sstride;
        a.sw = sw; a.sh = sh;
        a.band_rows = pl->g.band_rows; a.nbands = pl->g.nbands;
        a.blk_cols = pl->g.blk_cols; a.ncolblk = pl->g.ncolblk;
        a.nframes = n;
        // the persistent pipelined kernel needs 16-byte aligned rows on both frames and a tile of at
        // most kPipeMaxSlots chunks per thread; otherwise one workgroup per item
        a.pipe_wgs = 0;
        a.pipe_nt = 256;
        a.pipe_order = 0;
        a.cus = ctx->cus;
        a.dbg = 0;
        a.stamps = nullptr;
#if IPX_DIAG
        a.dbg = env_int("IPX_DBG", 0);
        static unsigned long long *stamp_buf = nullptr;   // IPX_STAMPS=1: phase stamps, never in a timed run
        if (env_int("IPX_STAMPS", 0)) {
            if (!stamp_buf) IPX_HIP(hipMalloc((void **)&stamp_buf, 8 * sizeof(unsigned long long)));
            IPX_HIP(hipMemsetAsync(stamp_buf, 0, 8 * sizeof(unsigned long long), s));
            a.stamps = stamp_buf;
        }
#endif
        const bool aligned = (sw & 3) == 0 && ((((uintptr_t)src) | (uintptr_t)sstride | src_frame_stride) & 15) == 0 &&
                             (!wm || ((((uintptr_t)wm) | wm_frame_stride) & 15) == 0);
        int pr = 0, pc = 0;
        if (aligned && env_int("IPX_PIPE", 1) && pl->g.most_rows <= 64 && band_pipe_shape(pl->g.band_rows, pl->g.blk_cols, &pr, &pc)) {
            a.pipe_wgs = std::max(1, env_int("IPX_PIPE_WGS", 8));  // clamped to what is resident at launch
            a.pipe_nt = env_int("IPX_PIPE_NT", 512) == 512 ? 512 : 256;
            a.pipe_order = -1;   // chosen below, once the operators are known
        }
        a.wm = wm; a.wm_frame_stride = wm_frame_stride; a.wm_stride = sw * 4;
        a.nscale = 0;
        a.nx_out[0] = a.nx_out[1] = 0;
        for (int k = 0; k < 2; k++) {
            const PlanScale &ps = pl->sc[k];
            if (!outs[k] || ps.dw <= 0 || ps.dh <= 0) continue;
            ScaleOut &o = a.sc[a.nscale++];
            o.out = outs[k]; o.frame_stride = ostr[k]; o.ostride = ps.dw * 4;
            o.dw = ps.dw; o.dh = ps.dh; o.sr_x0 = ps.sr.x0; o.sr_y0 = ps.sr.y0;
            o.xt = ps.xt; o.yt = ps.yt; o.row_begin = pl->g.row_begin[k]; o.col_begin = pl->g.col_begin[k];
            o.dyadic_shift = ps.dyadic_shift;
            o.imul = ps.imul; o.yrow = pl->g.yrow[k];
            a.nx_out[a.nscale - 1] = pl->g.nx_out[k];
        }
        if (a.nscale == 1) a.sc[1] = a.sc[0];  // keeps the kernel's unconditional tap loads legal
        const bool sep_glyphs = wm && pl->glyphs.n > 0 && glyphs_separate(IPX_FUSED_GLYPHS_RGBA);
        a.glyphs = pl->glyphs.dev; a.nglyphs = wm && !sep_glyphs ? pl->glyphs.n : 0; a.gbox = pl->glyphs.bbox;
        a.cr = col[0]; a.cg = col[1]; a.cb = col[2]; a.ca = col[3];
        if (!wm && a.nscale == 0) return IPX_OK;
        // every workgroup of the persistent kernel walks one contiguous run of items; 1 = it enters the run at an offset of its own
        // (IPX_PIPE_ORDER=0: every run from its first item -- bimodal from process to process, see band_pipe_kernel)
        if (a.pipe_order < 0) a.pipe_order = env_int("IPX_PIPE_ORDER", 1) ? 1 : 0;
        IPX_HIP(launch_band(a, s));
        if (sep_glyphs)
            IPX_HIP(launch_composite(wm, sw * 4, wm_frame_stride, n, pl->glyphs.dev, pl->glyphs.n, pl->glyphs.bbox, col[0], col[1], col[2], col[3], s));
        if (a.stamps) {
            unsigned long long h[8];
            IPX_HIP(hipMemcpyAsync(h, a.stamps, sizeof h, hipMemcpyDeviceToHost, s));
            IPX_HIP(hipStreamSynchronize(s));
            const double items = (double)pl->g.nbands * pl->g.ncolblk * n, waves = 4.0;
            fprintf(stderr, "[ipx stamps] cycles per item per wave: drain %.0f  barrier1 %.0f  issue %.0f  compute %.0f  barrier2 %.0f\n",
                    h[0] / items / waves, h[1] / items / waves, h[2] / items / waves, h[3] / items / waves, h[4] / items / waves);
        }
        return IPX_OK;
    }

    // unfused fallback (1-pixel-wide sources and the like): the per-operation kernels, frame by frame
    for (int i = 0; i < n; i++) {
        const uint8_t *f = src + (size_t)i * src_frame_stride;
        for (int k = 0; k < 2; k++) {
            const PlanScale &ps = pl->sc[k];
            if (!outs[k] || ps.dw <= 0 || ps.dh <= 0) continue;
            uint8_t *o = outs[k] + (size_t)i * ostr[k];
            IPX_HIP(hipMemsetAsync(o, 0, (size_t)ps.dw * ps.dh * 4, s));  // image.NewRGBA
            int rc = dev_scale(s, nullptr, o, ps.dw, ps.dh, ps.dw * 4, Rect{0, 0, ps.dw, ps.dh}, f, sw, sh,
                               sstride, ps.sr, IPX_OP_SRC);  // Over on a zeroed frame == Src
            if (rc) return rc;
        }
        if (wm) {
            uint8_t *o = wm + (size_t)i * wm_frame_stride;
            int rc = dev_draw(s, o, sw, sh, sw * 4, Rect{0, 0, sw, sh}, f, sw, sh, sstride, 0, 0, IPX_OP_SRC);
            if (rc) return rc;
        }
    }
    if (wm && pl->glyphs.n)
        IPX_HIP(launch_composite(wm, sw * 4, wm_frame_stride, n, pl->glyphs.dev, pl->glyphs.n, pl->glyphs.bbox,
                                 col[0], col[1], col[2], col[3], s));
    return IPX_OK;
}
IPX_CATCH_STATUS

// Frames in host memory, any packed source type: chunks over the lanes so that H2D of one chunk, the kernel of another and D2H of
// a third overlap.  kind: IPX_SRC_RGBA / IPX_SRC_NRGBA (4 bytes per pixel), IPX_GRAY (1), kPalettedKind (1 + 1 KiB palette per frame).
constexpr int kPalettedKind = 100;
constexpr int kDeepKind = 200;    // + IPX_DEEP_*
static int deep_bpp(int kind) { return kind == IPX_DEEP_GRAY16 ? 2 : (kind == IPX_DEEP_CMYK ? 4 : 8); }
static int run_host_packed(ipx_ctx *ctx, const ipx_plan *pl, int n, int kind, const uint8_t *src, int sstride, size_t src_frame_stride,
                           const uint8_t *palettes, uint8_t *resize_out, size_t resize_frame_stride, uint8_t *thumb_out,
                           size_t thumb_frame_stride, uint8_t *wm_out, size_t wm_frame_stride, const char *who)
{
    const int sw = pl->p.sw, sh = pl->p.sh;
    const int bpp = kind >= kDeepKind ? deep_bpp(kind - kDeepKind) : (kind == IPX_SRC_RGBA || kind == IPX_SRC_NRGBA ? 4 : 1);
    // device-side frame strides: tight when that keeps rows 16-byte aligned (then a whole chunk moves
    // with one copy per direction and buffer), padded to 256 otherwise
    auto dstride = [](size_t bytes) { return (bytes & 15) == 0 ? bytes : align256(bytes); };
    const size_t fsrc = dstride((size_t)sw * sh * bpp);
    const size_t fpal = kind == kPalettedKind ? 1024 : 0;
    const size_t fres = resize_out ? dstride(pl->info.resize_bytes) : 0;
    const size_t fth = thumb_out ? dstride(pl->info.thumb_bytes) : 0;
    const size_t fwm = wm_out ? dstride(pl->info.wm_bytes) : 0;
    const size_t per_frame = fsrc + fpal + fres + fth + fwm;
    // chunk the batch so that H2D of one chunk, the kernel of another and D2H of a third overlap on
    // different lanes (one stream each); several chunks per lane keep all three engines busy
    const int nl = (int)ctx->lanes.size();
    int chunk = std::max(1, (n + 4 * nl - 1) / (4 * nl));
    chunk = (int)std::min<size_t>((size_t)chunk, std::max<size_t>(1, ctx->lane_bytes / per_frame));
    chunk = std::max(1, std::min(chunk, env_int("IPX_HOST_CHUNK", 16)));

    // This call is the pipeline: it takes the lanes -- all but one when the context has three or more, so that a single-frame call
    // (the per-operator seam: one chunk, one lane) is served WHILE a batch runs instead of behind it; a call of one chunk takes one.
    const int nchunks = (n + chunk - 1) / chunk;
    const int want = nchunks <= 1 ? 1 : std::max(1, nl >= 3 ? nl - 1 : nl);
    std::vector<Lane *> lanes;
    const bool trace = want == 1 && env_int("IPX_DEBUG_SEAM", 0);
    const auto t_in = std::chrono::steady_clock::now();
    auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    {
        std::unique_lock<std::mutex> lk(ctx->mu);
        auto free_lanes = [&] { int k = 0; for (auto &l : ctx->lanes) k += !l.busy; return k; };
        ctx->cv.wait(lk, [&] { return free_lanes() >= want; });
        if (want == 1) {   // a single chunk: the last free lane (the high-priority one when it is free)
            for (size_t i = ctx->lanes.size(); i-- > 0;) if (!ctx->lanes[i].busy) { ctx->lanes[i].busy = true; lanes.push_back(&ctx->lanes[i]); break; }
        } else {
            for (auto &l : ctx->lanes) if (!l.busy && (int)lanes.size() < want) { l.busy = true; lanes.push_back(&l); }
        }
    }
    int rc = IPX_OK;
    hipError_t e = hipSuccess;
    const double t_lock = trace ? ms_since(t_in) : 0;
    for (auto *l : lanes) {
        rc = lane_reserve(*l, per_frame * chunk + 256);
        if (rc) break;
    }
    const double t_res = trace ? ms_since(t_in) : 0;
    const bool src_tight = sstride == sw * bpp && src_frame_stride == fsrc;
    // A single small frame in PAGEABLE memory (the per-operator seam from a caller that does not pin): the runtime's staged path for
    // small pageable copies blocks the enqueueing thread behind other streams' work every few contexts (a 640x360 call then takes as
    // long as the batch running beside it; tools/seam_hunt.py), larger ones it pins on the fly and never did.  Such a call goes
    // through a pinned bounce buffer of its lane instead: two host memcpys of < 1 ms.
    bool bounce = false;
    if (!rc && nchunks == 1 && n == 1 && kind != kPalettedKind && per_frame <= ((size_t)8 << 20)) {
        hipPointerAttribute_t at;
        const bool pinned = hipPointerGetAttributes(&at, src) == hipSuccess && at.type == hipMemoryTypeHost;
        (void)hipGetLastError();
        if (!pinned) {
            Lane &l = *lanes[0];
            if (l.pin_bytes < per_frame) {
                if (l.pin) (void)hipHostFree(l.pin);
                l.pin = nullptr; l.pin_bytes = 0;
                if (hipHostMalloc((void **)&l.pin, per_frame + (per_frame >> 2), hipHostMallocDefault) == hipSuccess) l.pin_bytes = per_frame + (per_frame >> 2);
                else (void)hipGetLastError();
            }
            bounce = l.pin != nullptr;
        }
    }
    auto d2h = [&](uint8_t *host, size_t host_stride, const uint8_t *dev, size_t dev_stride, size_t bytes, int i0, int m,
                   hipStream_t st) {
        if (!dev || !bytes) return hipSuccess;
        if (host_stride == dev_stride)  // tight on both sides: one copy for the chunk
            return hipMemcpyAsync(host + (size_t)i0 * host_stride, dev, dev_stride * (m - 1) + bytes, hipMemcpyDeviceToHost, st);
        hipError_t r = hipSuccess;
        for (int i = 0; i < m && r == hipSuccess; i++)
            r = hipMemcpyAsync(host + (size_t)(i0 + i) * host_stride, dev + dev_stride * i, bytes, hipMemcpyDeviceToHost, st);
        return r;
    };
    for (int i0 = 0, c = 0; !rc && e == hipSuccess && i0 < n; i0 += chunk, c++) {
        Lane &l = *lanes[c % lanes.size()];
        const int m = std::min(chunk, n - i0);
        uint8_t *dsrc = (uint8_t *)(((uintptr_t)l.dev + 255) & ~(uintptr_t)255);
        uint8_t *dres = fres ? dsrc + fsrc * chunk : nullptr;
        uint8_t *dth = fth ? dsrc + (fsrc + fres) * chunk : nullptr;
        uint8_t *dwm = fwm ? dsrc + (fsrc + fres + fth) * chunk : nullptr;
        uint8_t *dpal = fpal ? dsrc + (fsrc + fres + fth + fwm) * chunk : nullptr;
        // the lane's stream serialises reuse of its scratch: chunk c waits for chunk c - lanes
        if (bounce) {
            for (int y = 0; y < sh; y++) memcpy(l.pin + (size_t)y * sw * bpp, src + (size_t)y * sstride, (size_t)sw * bpp);
            e = hipMemcpyAsync(dsrc, l.pin, (size_t)sw * sh * bpp, hipMemcpyHostToDevice, l.stream);
        } else if (src_tight) {
            e = hipMemcpyAsync(dsrc, src + (size_t)i0 * src_frame_stride, fsrc * m, hipMemcpyHostToDevice, l.stream);
        } else {
            for (int i = 0; i < m && e == hipSuccess; i++)
                e = hipMemcpy2DAsync(dsrc + fsrc * i, (size_t)sw * bpp, src + (size_t)(i0 + i) * src_frame_stride, sstride,
                                     (size_t)sw * bpp, sh, hipMemcpyHostToDevice, l.stream);
        }
        if (e == hipSuccess && dpal) e = hipMemcpyAsync(dpal, palettes + (size_t)i0 * 1024, (size_t)m * 1024, hipMemcpyHostToDevice, l.stream);
        if (e != hipSuccess) break;
        switch (kind) {
        case IPX_SRC_RGBA: rc = ipx_plan_run_dev(ctx, l.stream, pl, m, dsrc, sw * 4, fsrc, dres, fres, dth, fth, dwm, fwm); break;
        case IPX_SRC_NRGBA: rc = ipx_plan_run_dev_nrgba(ctx, l.stream, pl, m, dsrc, sw * 4, fsrc, dres, fres, dth, fth, dwm, fwm); break;
        case IPX_GRAY: rc = ipx_plan_run_dev_gray(ctx, l.stream, pl, m, dsrc, sw, fsrc, dres, fres, dth, fth, dwm, fwm); break;
        case kPalettedKind: rc = ipx_plan_run_dev_paletted(ctx, l.stream, pl, m, dsrc, sw, fsrc, dpal, dres, fres, dth, fth, dwm, fwm); break;
        default: rc = ipx_plan_run_dev_deep(ctx, l.stream, pl, m, kind - kDeepKind, dsrc, sw * bpp, fsrc, dres, fres, dth, fth, dwm, fwm); break;
        }
        if (rc) break;
        if (bounce) {          // one copy of the three outputs (they follow the source in the lane's scratch), handed out after the sync below
            if (fres + fth + fwm) e = hipMemcpyAsync(l.pin + fsrc, dsrc + fsrc, fres + fth + fwm, hipMemcpyDeviceToHost, l.stream);
            continue;
        }
        e = d2h(resize_out, resize_frame_stride, dres, fres, pl->info.resize_bytes, i0, m, l.stream);
        if (e == hipSuccess) e = d2h(thumb_out, thumb_frame_stride, dth, fth, pl->info.thumb_bytes, i0, m, l.stream);
        if (e == hipSuccess) e = d2h(wm_out, wm_frame_stride, dwm, fwm, pl->info.wm_bytes, i0, m, l.stream);
    }
    const double t_enq = trace ? ms_since(t_in) : 0;
    for (auto *l : lanes) {
        hipError_t e2 = hipStreamSynchronize(l->stream);
        if (e == hipSuccess) e = e2;
    }
    if (bounce && !rc && e == hipSuccess) {
        const uint8_t *p = lanes[0]->pin + fsrc;
        if (resize_out) memcpy(resize_out, p, pl->info.resize_bytes);
        if (thumb_out) memcpy(thumb_out, p + fres, pl->info.thumb_bytes);
        if (wm_out) memcpy(wm_out, p + fres + fth, pl->info.wm_bytes);
    }
    if (trace)
        fprintf(stderr, "[ipx seam] lane %d of %d: lock %.2f ms, reserve %.2f, enqueued %.2f, done %.2f\n", (int)(lanes[0] - &ctx->lanes[0]), nl, t_lock, t_res,
                t_enq, ms_since(t_in));
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        for (auto *l : lanes) l->busy = false;
    }
    ctx->cv.notify_all();
    if (!rc && e != hipSuccess) { set_error("%s: %s", who, hipGetErrorString(e)); rc = IPX_ERR_HIP; }
    return rc;
}

int ipx_plan_run_host(ipx_ctx *ctx, const ipx_plan *pl, int n, const uint8_t *src, int sstride,
                      size_t src_frame_stride, uint8_t *resize_out, size_t resize_frame_stride,
                      uint8_t *thumb_out, size_t thumb_frame_stride, uint8_t *wm_out,
                      size_t wm_frame_stride) try
{
    IPX_ENTER(ctx);
    if (!pl || n < 0 || !src || (long long)sstride < (long long)pl->p.sw * 4) {
        set_error("ipx_plan_run_host: bad argument");
        return IPX_ERR_INVALID;
    }
    IPX_PLAN_SRC("ipx_plan_run_host", pl, sstride, 4);
    if (n == 0) return IPX_OK;
    return run_host_packed(ctx, pl, n, IPX_SRC_RGBA, src, sstride, src_frame_stride, nullptr, resize_out, resize_frame_stride, thumb_out,
                           thumb_frame_stride, wm_out, wm_frame_stride, "ipx_plan_run_host");
}
IPX_CATCH_STATUS

int ipx_plan_run_host_nrgba(ipx_ctx *ctx, const ipx_plan *pl, int n, const uint8_t *src, int sstride, size_t src_frame_stride,
                            uint8_t *resize_out, size_t resize_frame_stride, uint8_t *thumb_out, size_t thumb_frame_stride, uint8_t *wm_out,
                            size_t wm_frame_stride) try
{
    IPX_ENTER(ctx);
    if (!pl || n < 0 || !src || (long long)sstride < (long long)pl->p.sw * 4) { set_error("ipx_plan_run_host_nrgba: bad argument"); return IPX_ERR_INVALID; }
    IPX_PLAN_SRC("ipx_plan_run_host_nrgba", pl, sstride, 4);
    if (n == 0) return IPX_OK;
    return run_host_packed(ctx, pl, n, IPX_SRC_NRGBA, src, sstride, src_frame_stride, nullptr, resize_out, resize_frame_stride, thumb_out,
                           thumb_frame_stride, wm_out, wm_frame_stride, "ipx_plan_run_host_nrgba");
}
IPX_CATCH_STATUS

int ipx_plan_run_host_gray(ipx_ctx *ctx, const ipx_plan *pl, int n, const uint8_t *gray, int stride, size_t frame_stride,
                           uint8_t *resize_out, size_t resize_frame_stride, uint8_t *thumb_out, size_t thumb_frame_stride, uint8_t *wm_out,
                           size_t wm_frame_stride) try
{
    IPX_ENTER(ctx);
    if (!pl || n < 0 || !gray || stride < pl->p.sw) { set_error("ipx_plan_run_host_gray: bad argument"); return IPX_ERR_INVALID; }
    IPX_PLAN_SRC("ipx_plan_run_host_gray", pl, stride, 1);
    if (n == 0) return IPX_OK;
    return run_host_packed(ctx, pl, n, IPX_GRAY, gray, stride, frame_stride, nullptr, resize_out, resize_frame_stride, thumb_out,
                           thumb_frame_stride, wm_out, wm_frame_stride, "ipx_plan_run_host_gray");
}
IPX_CATCH_STATUS

int ipx_plan_run_host_paletted(ipx_ctx *ctx, const ipx_plan *pl, int n, const uint8_t *index, int stride, size_t frame_stride,
                               const uint8_t *palettes, uint8_t *resize_out, size_t resize_frame_stride, uint8_t *thumb_out,
                               size_t thumb_frame_stride, uint8_t *wm_out, size_t wm_frame_stride) try
{
    IPX_ENTER(ctx);
    if (!pl || n < 0 || !index || !palettes || stride < pl->p.sw) { set_error("ipx_plan_run_host_paletted: bad argument"); return IPX_ERR_INVALID; }
    IPX_PLAN_SRC("ipx_plan_run_host_paletted", pl, stride, 1);
    if (n == 0) return IPX_OK;
    return run_host_packed(ctx, pl, n, kPalettedKind, index, stride, frame_stride, palettes, resize_out, resize_frame_stride, thumb_out,
                           thumb_frame_stride, wm_out, wm_frame_stride, "ipx_plan_run_host_paletted");
}
IPX_CATCH_STATUS


// ---- decoded JPEG batches ---------------------------------------------------------------------------
// The reference treats a *image.YCbCr source differently per operator:
//   resize, non-crop thumbnail: scale_RGBA_YCbCr4xx_Src -- every TAP converted to 16 bit, then the lerp;
//   crop thumbnail: the crop copy converts to RGBA8 first (imageutil.DrawYCbCr), the scale then reads RGBA8;
//   watermark: draw.Draw converts to RGBA8 (DrawYCbCr), the glyphs go over that.
// So: one conversion pass into the watermark frame (or scratch), the RGBA band kernel for the crop
// thumbnail on the converted frames, the glyph composite in place, and a batched YCbCr scale.
// BandArgs of the fused kernels that convert their source on the fly (ipx_band_conv.hip, ipx_band_nrgba.hip): the plan's tiling, the
// outputs that are wanted, and per scaled output the conversion rule -- mode 0 = 16-bit taps (resizeImage on the source image itself),
// mode 1 = 8-bit RGBA first (the crop thumbnail scales the RGBA8 copy cropAndResize made, thumbnail.go:128-131)
// The text composite as a pass of its own over the watermark frames' text box (composite_kernel) after the band kernel has copied /
// converted every pixel, instead of inside the band kernel (IPX_FUSED_GLYPHS in ipx_internal.h has the why).
static bool glyphs_separate(bool fused_kernel) { return !fused_kernel || env_int("IPX_GLYPH_SEPARATE", 0) != 0; }
static hipError_t composite_after(const ipx_plan *pl, bool fused_kernel, uint8_t *wm, size_t wm_frame_stride, int n, hipStream_t s)
{
    if (!wm || pl->glyphs.n <= 0 || !glyphs_separate(fused_kernel)) return hipSuccess;
    const uint8_t *c = pl->p.glyphs ? pl->p.glyphs->col : nullptr;
    return launch_composite(wm, pl->p.sw * 4, wm_frame_stride, n, pl->glyphs.dev, pl->glyphs.n, pl->glyphs.bbox, c ? c[0] * 0x101u : 0,
                            c ? c[1] * 0x101u : 0, c ? c[2] * 0x101u : 0, c ? c[3] * 0x101u : 0, s);
}

static void fill_converting_band_args(ipx_ctx *ctx, const ipx_plan *pl, int n, uint8_t *res, size_t resize_frame_stride, uint8_t *th,
                                      size_t thumb_frame_stride, uint8_t *wm, size_t wm_frame_stride, const PlanGeom &g, bool fused_glyphs, BandArgs &a,
                                      int mode[2])
{
    const int sw = pl->p.sw, sh = pl->p.sh;
    a.sw = sw; a.sh = sh;
    a.band_rows = g.band_rows; a.nbands = g.nbands;
    a.blk_cols = g.blk_cols; a.ncolblk = g.ncolblk;
    a.nframes = n;
    a.pipe_wgs = std::max(1, env_int("IPX_PIPE_WGS", 8));
    a.pipe_nt = 512; a.pipe_order = 1;
    a.cus = ctx->cus;
#if IPX_DIAG
    a.dbg = env_int("IPX_DBG", 0);
#endif
    a.wm = wm; a.wm_frame_stride = wm_frame_stride; a.wm_stride = sw * 4;
    uint8_t *outs[2] = {res, th};
    const size_t ostr[2] = {resize_frame_stride, thumb_frame_stride};
    for (int k = 0; k < 2; k++) {
        const PlanScale &ps = pl->sc[k];
        if (!outs[k] || ps.dw <= 0 || ps.dh <= 0) continue;
        mode[a.nscale] = k == 1 && pl->p.crop_to_fit ? 1 : 0;
        ScaleOut &o = a.sc[a.nscale++];
        o.out = outs[k]; o.frame_stride = ostr[k]; o.ostride = ps.dw * 4;
        o.dw = ps.dw; o.dh = ps.dh; o.sr_x0 = ps.sr.x0; o.sr_y0 = ps.sr.y0;
        o.xt = ps.xt; o.yt = ps.yt; o.row_begin = g.row_begin[k]; o.col_begin = g.col_begin[k];
        o.dyadic_shift = ps.dyadic_shift;
        if (mode[a.nscale - 1] == 1) { o.imul = g.yrow[k] ? ps.imul : 0; o.yrow = g.yrow[k]; }   // RGBA8 taps only
        else o.yrow16 = g.yrow16[k];
        // 16-bit taps in u32: x0*tap + x1*tap < 2^24 and the weights below 2^9 keep every product in 24 x 24 bits
        if (mode[a.nscale - 1] == 0 && (ps.kx > 8 || ps.ky > 8)) o.dyadic_shift = -1;
        a.nx_out[a.nscale - 1] = g.nx_out[k];
    }
    if (a.nscale == 1) { a.sc[1] = a.sc[0]; mode[1] = mode[0]; }
    const uint8_t *c = pl->p.glyphs ? pl->p.glyphs->col : nullptr;
    a.glyphs = pl->glyphs.dev; a.nglyphs = wm && !glyphs_separate(fused_glyphs) ? pl->glyphs.n : 0; a.gbox = pl->glyphs.bbox;
    a.cr = c ? c[0] * 0x101u : 0; a.cg = c ? c[1] * 0x101u : 0; a.cb = c ? c[2] * 0x101u : 0; a.ca = c ? c[3] * 0x101u : 0;
}

// flat_chroma: cb == cr == one row of 128s read with stride 0 (a Gray frame seen as YCbCr); everything else as the public entry
static int run_dev_ycbcr(ipx_ctx *ctx, hipStream_t s, const ipx_plan *pl, int n, const ipx_ycbcr_batch *src, bool flat_chroma,
                         uint8_t *resize_out, size_t resize_frame_stride, uint8_t *thumb_out, size_t thumb_frame_stride, uint8_t *wm_out,
                         size_t wm_frame_stride);

int ipx_plan_run_dev_ycbcr(ipx_ctx *ctx, void *stream, const ipx_plan *pl, int n, const ipx_ycbcr_batch *src,
                           uint8_t *resize_out, size_t resize_frame_stride, uint8_t *thumb_out,
                           size_t thumb_frame_stride, uint8_t *wm_out, size_t wm_frame_stride) try
{
    IPX_ENTER(ctx);
    if (!pl || n < 0 || !src || !src->y || !src->cb || !src->cr || src->ratio < 0 || src->ratio > IPX_YCBCR_440 ||
        src->ystride < pl->p.sw) {
        set_error("ipx_plan_run_dev_ycbcr: bad argument");
        return IPX_ERR_INVALID;
    }
    IPX_PLAN_SRC("ipx_plan_run_dev_ycbcr", pl, src->ystride, 1);
    if (!frame_span_ok(pl->p.sw, pl->p.sh, src->cstride, 1)) { set_error("ipx_plan_run_dev_ycbcr: chroma planes beyond the addressable span"); return IPX_ERR_UNSUPPORTED; }
    if (n == 0) return IPX_OK;
    if (n > 65535) { set_error("ipx_plan_run_dev_ycbcr: at most 65535 frames per call"); return IPX_ERR_UNSUPPORTED; }
    return run_dev_ycbcr(ctx, stream ? (hipStream_t)stream : ctx->stream, pl, n, src, false, resize_out, resize_frame_stride, thumb_out,
                         thumb_frame_stride, wm_out, wm_frame_stride);
}
IPX_CATCH_STATUS

static int run_dev_ycbcr(ipx_ctx *ctx, hipStream_t s, const ipx_plan *pl, int n, const ipx_ycbcr_batch *src, bool flat_chroma,
                         uint8_t *resize_out, size_t resize_frame_stride, uint8_t *thumb_out, size_t thumb_frame_stride, uint8_t *wm_out,
                         size_t wm_frame_stride)
{
    const int sw = pl->p.sw, sh = pl->p.sh;
    uint8_t *res = pl->sc[0].on ? resize_out : nullptr;
    uint8_t *th = pl->sc[1].on ? thumb_out : nullptr;
    uint8_t *wm = pl->p.do_watermark ? wm_out : nullptr;
    const bool crop_thumb = th && pl->p.crop_to_fit;

    // one fused pass over the planes when the tile shape and alignments allow it (ipx_band_conv.hip)
    if (pl->fused && env_int("IPX_YCC_FUSED", 1) && pl->conv.ok && pl->conv.most_rows <= 64) {
        YccArgs A{};
        BandArgs &a = A.b;
        fill_converting_band_args(ctx, pl, n, res, resize_frame_stride, th, thumb_frame_stride, wm, wm_frame_stride, pl->conv, IPX_FUSED_GLYPHS_CONV, a, A.mode);
        if (!wm && a.nscale == 0) return IPX_OK;
        A.y = src->y; A.cb = src->cb; A.cr = src->cr; A.ystride = src->ystride; A.cstride = src->cstride;
        A.y_fs = src->y_frame_stride; A.c_fs = src->c_frame_stride; A.ratio = src->ratio;
        A.cw = (src->ratio == IPX_YCBCR_422 || src->ratio == IPX_YCBCR_420) ? (sw + 1) / 2 : sw;
        A.ch = (src->ratio == IPX_YCBCR_420 || src->ratio == IPX_YCBCR_440) ? (sh + 1) / 2 : sh;
#if IPX_DIAG
        static unsigned long long *stamp_buf = nullptr;   // IPX_STAMPS=1: phase stamps, never in a timed run
        if (env_int("IPX_STAMPS", 0)) {
            if (!stamp_buf) IPX_HIP(hipMalloc((void **)&stamp_buf, 8 * sizeof(unsigned long long)));
            IPX_HIP(hipMemsetAsync(stamp_buf, 0, 8 * sizeof(unsigned long long), s));
            a.stamps = stamp_buf;
        }
#endif
        bool matched = false;
        if (src->cstride >= A.cw || flat_chroma) IPX_HIP(launch_band_ycc(A, s, &matched));
        if (matched) IPX_HIP(composite_after(pl, IPX_FUSED_GLYPHS_CONV, wm, wm_frame_stride, n, s));
#if IPX_DIAG
        if (matched && a.stamps) {
            unsigned long long h[8];
            IPX_HIP(hipMemcpyAsync(h, a.stamps, sizeof h, hipMemcpyDeviceToHost, s));
            IPX_HIP(hipStreamSynchronize(s));
            const double wi = (double)pl->conv.nbands * pl->conv.ncolblk * n * 8.0;
            fprintf(stderr, "[ipx stamps ycc] cycles per item per wave: wait-loads %.0f  drain %.0f  barrier1 %.0f  issue %.0f  compute %.0f  barrier2 %.0f\n",
                    h[0] / wi, h[1] / wi, h[2] / wi, h[3] / wi, h[4] / wi, h[5] / wi);
        }
#endif
        if (matched) return IPX_OK;
    }

    if (flat_chroma) return 1;   // Gray frames the fused kernel does not take: the caller expands them and uses the RGBA pass

    DevSrc ysrc;
    ysrc.kind = IPX_SRC_YCBCR; ysrc.pix = src->y; ysrc.stride = src->ystride; ysrc.cb = src->cb; ysrc.cr = src->cr;
    ysrc.cstride = src->cstride; ysrc.ratio = src->ratio; ysrc.w = sw; ysrc.h = sh;
    ysrc.nframes = n; ysrc.frame_stride = src->y_frame_stride; ysrc.c_frame_stride = src->c_frame_stride;

    // RGBA8 conversion of the whole batch, straight into the watermark frames when they are wanted
    uint8_t *conv = wm;
    size_t conv_fs = wm_frame_stride;
    uint8_t *scratch = nullptr;
    if (!conv && crop_thumb) {
        conv_fs = (size_t)sw * sh * 4;
        IPX_HIP(hipMallocAsync((void **)&scratch, conv_fs * n, s));
        conv = scratch;
    }
    int rc = IPX_OK;
    if (conv)
        IPX_HIP(launch_draw_ycbcr(conv, sw * 4, src->y, src->ystride, src->cb, src->cr, src->cstride, src->ratio, 0, 0, sw,
                                  sh, s, n, conv_fs, src->y_frame_stride, src->c_frame_stride));
    if (crop_thumb) {
        ipx_plan *sub = nullptr;
        {
            std::lock_guard<std::mutex> lk(pl->mu);
            if (!pl->thumb_only) {
                ipx_plan_params tp;
                memset(&tp, 0, sizeof tp);
                tp.sw = sw; tp.sh = sh; tp.do_thumbnail = 1; tp.thumb_size = pl->p.thumb_size; tp.crop_to_fit = 1;
                rc = ipx_plan_create(ctx, &tp, &pl->thumb_only);
            }
            sub = pl->thumb_only;
        }
        if (!rc) rc = ipx_plan_run_dev(ctx, s, sub, n, conv, sw * 4, conv_fs, nullptr, 0, th, thumb_frame_stride, nullptr, 0);
    }
    if (!rc && wm && pl->glyphs.n && pl->p.glyphs) {
        const uint8_t *c = pl->p.glyphs->col;
        hipError_t e = launch_composite(wm, sw * 4, wm_frame_stride, n, pl->glyphs.dev, pl->glyphs.n, pl->glyphs.bbox,
                                        c[0] * 0x101u, c[1] * 0x101u, c[2] * 0x101u, c[3] * 0x101u, s);
        if (e != hipSuccess) { set_error("composite launch failed: %s", hipGetErrorString(e)); rc = IPX_ERR_HIP; }
    }
    for (int k = 0; k < 2 && !rc; k++) {   // 16-bit-tap scales straight from the planes
        const PlanScale &ps = pl->sc[k];
        uint8_t *o = k == 0 ? res : (crop_thumb ? nullptr : th);
        const size_t ofs = k == 0 ? resize_frame_stride : thumb_frame_stride;
        if (!o || ps.dw <= 0 || ps.dh <= 0) continue;
        rc = dev_scale_src(s, nullptr, o, ps.dw, ps.dh, ps.dw * 4, Rect{0, 0, ps.dw, ps.dh}, ysrc, ps.sr, IPX_OP_SRC, ofs);
    }
    if (scratch) (void)hipFreeAsync(scratch, s);
    return rc;
}

int ipx_plan_run_dev_nrgba(ipx_ctx *ctx, void *stream, const ipx_plan *pl, int n, const uint8_t *src, int sstride, size_t src_frame_stride,
                           uint8_t *resize_out, size_t resize_frame_stride, uint8_t *thumb_out, size_t thumb_frame_stride, uint8_t *wm_out,
                           size_t wm_frame_stride) try
{
    IPX_ENTER(ctx);
    if (!pl || n < 0 || !src || (long long)sstride < (long long)pl->p.sw * 4) { set_error("ipx_plan_run_dev_nrgba: bad argument"); return IPX_ERR_INVALID; }
    IPX_PLAN_SRC("ipx_plan_run_dev_nrgba", pl, sstride, 4);
    if (n == 0) return IPX_OK;
    if (n > 65535) { set_error("ipx_plan_run_dev_nrgba: at most 65535 frames per call"); return IPX_ERR_UNSUPPORTED; }
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    const int sw = pl->p.sw, sh = pl->p.sh;
    uint8_t *res = pl->sc[0].on ? resize_out : nullptr;
    uint8_t *th = pl->sc[1].on ? thumb_out : nullptr;
    uint8_t *wm = pl->p.do_watermark ? wm_out : nullptr;
    const bool crop_thumb = th && pl->p.crop_to_fit;
    // one fused pass over the frames when the tile shape and alignments allow it: the converted-tile kernel (every source pixel
    // premultiplied once, ipx_band_conv.hip) on the plan's `conv` tiling, else the per-tap kernel of ipx_band_nrgba.hip
    if (pl->fused && env_int("IPX_NRGBA_FUSED", 1) && env_int("IPX_NRGBA_CONV", 1) && pl->conv.ok && pl->conv.most_rows <= 64) {
        NrgbaArgs A{};
        BandArgs &a = A.b;
        fill_converting_band_args(ctx, pl, n, res, resize_frame_stride, th, thumb_frame_stride, wm, wm_frame_stride, pl->conv, IPX_FUSED_GLYPHS_CONV, a, A.mode);
        if (!wm && a.nscale == 0) return IPX_OK;
        a.src = src; a.src_frame_stride = src_frame_stride; a.sstride = sstride;
        bool matched = false;
        IPX_HIP(launch_band_nrgba_conv(A, s, &matched));
        if (matched) IPX_HIP(composite_after(pl, IPX_FUSED_GLYPHS_CONV, wm, wm_frame_stride, n, s));
        if (matched) return IPX_OK;
    }
    if (pl->fused && env_int("IPX_NRGBA_FUSED", 1) && pl->g.band_rows <= 8 && pl->g.most_rows <= 64) {
        NrgbaArgs A{};
        BandArgs &a = A.b;
        fill_converting_band_args(ctx, pl, n, res, resize_frame_stride, th, thumb_frame_stride, wm, wm_frame_stride, pl->g, IPX_FUSED_GLYPHS_RGBA, a, A.mode);
        if (!wm && a.nscale == 0) return IPX_OK;
        a.src = src; a.src_frame_stride = src_frame_stride; a.sstride = sstride;
        bool matched = false;
        IPX_HIP(launch_band_nrgba(A, s, &matched));
        if (matched) IPX_HIP(composite_after(pl, IPX_FUSED_GLYPHS_RGBA, wm, wm_frame_stride, n, s));
        if (matched) return IPX_OK;
    }
    // premultiplied RGBA8 of the whole batch (drawNRGBASrc == drawNRGBAOver onto a zeroed frame), into the watermark frames when wanted
    uint8_t *conv = wm;
    size_t conv_fs = wm_frame_stride;
    uint8_t *scratch = nullptr;
    if (!conv && crop_thumb) {
        conv_fs = (size_t)sw * sh * 4;
        IPX_HIP(hipMallocAsync((void **)&scratch, conv_fs * n, s));
        conv = scratch;
    }
    int rc = IPX_OK;
    if (conv) {
        hipError_t e = launch_draw_nrgba(conv, sw * 4, src, sstride, sw, sh, IPX_OP_SRC, s, n, conv_fs, src_frame_stride);
        if (e != hipSuccess) { set_error("premultiply launch failed: %s", hipGetErrorString(e)); rc = IPX_ERR_HIP; }
    }
    if (!rc && crop_thumb) {
        ipx_plan *sub = nullptr;
        {
            std::lock_guard<std::mutex> lk(pl->mu);
            if (!pl->thumb_only) {
                ipx_plan_params tp;
                memset(&tp, 0, sizeof tp);
                tp.sw = sw; tp.sh = sh; tp.do_thumbnail = 1; tp.thumb_size = pl->p.thumb_size; tp.crop_to_fit = 1;
                rc = ipx_plan_create(ctx, &tp, &pl->thumb_only);
            }
            sub = pl->thumb_only;
        }
        if (!rc) rc = ipx_plan_run_dev(ctx, s, sub, n, conv, sw * 4, conv_fs, nullptr, 0, th, thumb_frame_stride, nullptr, 0);
    }
    if (!rc && wm && pl->glyphs.n && pl->p.glyphs) {
        const uint8_t *c = pl->p.glyphs->col;
        hipError_t e = launch_composite(wm, sw * 4, wm_frame_stride, n, pl->glyphs.dev, pl->glyphs.n, pl->glyphs.bbox,
                                        c[0] * 0x101u, c[1] * 0x101u, c[2] * 0x101u, c[3] * 0x101u, s);
        if (e != hipSuccess) { set_error("composite launch failed: %s", hipGetErrorString(e)); rc = IPX_ERR_HIP; }
    }
    DevSrc nsrc;
    nsrc.kind = IPX_SRC_NRGBA; nsrc.pix = src; nsrc.stride = sstride; nsrc.w = sw; nsrc.h = sh;
    nsrc.nframes = n; nsrc.frame_stride = src_frame_stride;
    for (int k = 0; k < 2 && !rc; k++) {   // 16-bit premultiplied taps straight from the source; Over onto the zeroed frame == Src
        const PlanScale &ps = pl->sc[k];
        uint8_t *o = k == 0 ? res : (crop_thumb ? nullptr : th);
        const size_t ofs = k == 0 ? resize_frame_stride : thumb_frame_stride;
        if (!o || ps.dw <= 0 || ps.dh <= 0) continue;
        rc = dev_scale_src(s, nullptr, o, ps.dw, ps.dh, ps.dw * 4, Rect{0, 0, ps.dw, ps.dh}, nsrc, ps.sr, IPX_OP_SRC, ofs);
    }
    if (scratch) (void)hipFreeAsync(scratch, s);
    return rc;
}
IPX_CATCH_STATUS

// The deep source types: one expansion pass to frames of 16-bit taps (what every consumer of these types reads: At(x, y).RGBA()), then
// the converted-tile kernel on them, or the three-kernel path -- top bytes into the watermark frames (drawRGBA / drawCMYK with Src), the
// crop thumbnail from those RGBA8 frames, the text pass, and the per-tap scale from the taps.
int ipx_plan_run_dev_deep(ipx_ctx *ctx, void *stream, const ipx_plan *pl, int n, int kind, const uint8_t *src, int sstride, size_t src_frame_stride,
                          uint8_t *resize_out, size_t resize_frame_stride, uint8_t *thumb_out, size_t thumb_frame_stride, uint8_t *wm_out,
                          size_t wm_frame_stride) try
{
    IPX_ENTER(ctx);
    if (kind != IPX_DEEP_NRGBA64 && kind != IPX_DEEP_RGBA64 && kind != IPX_DEEP_GRAY16 && kind != IPX_DEEP_CMYK) {
        set_error("ipx_plan_run_dev_deep: unknown source type %d", kind);
        return IPX_ERR_INVALID;
    }
    const int bpp = deep_bpp(kind);
    const uintptr_t al = kind == IPX_DEEP_CMYK ? 3 : 1;
    if (!pl || n < 0 || !src || (long long)sstride < (long long)pl->p.sw * bpp || ((((uintptr_t)src) | (uintptr_t)sstride | src_frame_stride) & al)) {
        set_error("ipx_plan_run_dev_deep: bad argument");
        return IPX_ERR_INVALID;
    }
    IPX_PLAN_SRC("ipx_plan_run_dev_deep", pl, sstride, bpp);
    IPX_PLAN_SRC("ipx_plan_run_dev_deep", pl, (long long)pl->p.sw * 8, 8);        // the frames of taps
    if (n == 0) return IPX_OK;
    if (n > 65535) { set_error("ipx_plan_run_dev_deep: at most 65535 frames per call"); return IPX_ERR_UNSUPPORTED; }
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    const int sw = pl->p.sw, sh = pl->p.sh;
    uint8_t *res = pl->sc[0].on ? resize_out : nullptr;
    uint8_t *th = pl->sc[1].on ? thumb_out : nullptr;
    uint8_t *wm = pl->p.do_watermark ? wm_out : nullptr;
    if (!res && !th && !wm) return IPX_OK;
    const bool crop_thumb = th && pl->p.crop_to_fit;
    // one fused pass straight from Go's Pix when the tile shape and the alignments allow it
    if (pl->fused && env_int("IPX_DEEP_FUSED", 1) && env_int("IPX_DEEP_DIRECT", 1) && pl->conv.ok && pl->conv.most_rows <= 64) {
        NrgbaArgs A{};
        BandArgs &a = A.b;
        fill_converting_band_args(ctx, pl, n, res, resize_frame_stride, th, thumb_frame_stride, wm, wm_frame_stride, pl->conv, IPX_FUSED_GLYPHS_CONV, a, A.mode);
        a.src = src; a.src_frame_stride = src_frame_stride; a.sstride = sstride;
        bool matched = false;
        IPX_HIP(launch_band_deep(A, kind, s, &matched));
        if (matched) IPX_HIP(composite_after(pl, IPX_FUSED_GLYPHS_CONV, wm, wm_frame_stride, n, s));
        if (matched) return IPX_OK;
    }
    const size_t tfs = align256((size_t)sw * sh * 8);
    uint8_t *taps = nullptr;
    IPX_HIP(hipMallocAsync((void **)&taps, tfs * n, s));
    struct Free { uint8_t *p; hipStream_t s; ~Free() { if (p) (void)hipFreeAsync(p, s); } } free_taps{taps, s}, free_scratch{nullptr, s};
    {
        hipError_t e = launch_deep_expand(taps, tfs, src, sstride, src_frame_stride, kind, sw, sh, n, s);
        if (e != hipSuccess) { set_error("tap expansion failed: %s", hipGetErrorString(e)); return IPX_ERR_HIP; }
    }
    if (pl->fused && env_int("IPX_DEEP_FUSED", 1) && pl->conv.ok && pl->conv.most_rows <= 64) {
        NrgbaArgs A{};
        BandArgs &a = A.b;
        fill_converting_band_args(ctx, pl, n, res, resize_frame_stride, th, thumb_frame_stride, wm, wm_frame_stride, pl->conv, IPX_FUSED_GLYPHS_CONV, a, A.mode);
        a.src = taps; a.src_frame_stride = tfs; a.sstride = sw * 8;
        bool matched = false;
        IPX_HIP(launch_band_tap64_conv(A, s, &matched));
        if (matched) IPX_HIP(composite_after(pl, IPX_FUSED_GLYPHS_CONV, wm, wm_frame_stride, n, s));
        if (matched) return IPX_OK;
    }
    uint8_t *conv = wm;
    size_t conv_fs = wm_frame_stride;
    if (!conv && crop_thumb) {
        conv_fs = (size_t)sw * sh * 4;
        IPX_HIP(hipMallocAsync((void **)&free_scratch.p, conv_fs * n, s));
        conv = free_scratch.p;
    }
    int rc = IPX_OK;
    if (conv) {
        hipError_t e = launch_draw_tap64(conv, sw * 4, taps, sw * 8, sw, sh, IPX_OP_SRC, s, n, conv_fs, tfs);
        if (e != hipSuccess) { set_error("tap narrowing failed: %s", hipGetErrorString(e)); return IPX_ERR_HIP; }
    }
    if (crop_thumb) {
        ipx_plan *sub = nullptr;
        {
            std::lock_guard<std::mutex> lk(pl->mu);
            if (!pl->thumb_only) {
                ipx_plan_params tp;
                memset(&tp, 0, sizeof tp);
                tp.sw = sw; tp.sh = sh; tp.do_thumbnail = 1; tp.thumb_size = pl->p.thumb_size; tp.crop_to_fit = 1;
                rc = ipx_plan_create(ctx, &tp, &pl->thumb_only);
            }
            sub = pl->thumb_only;
        }
        if (!rc) rc = ipx_plan_run_dev(ctx, s, sub, n, conv, sw * 4, conv_fs, nullptr, 0, th, thumb_frame_stride, nullptr, 0);
    }
    if (!rc && wm && pl->glyphs.n && pl->p.glyphs) {
        const uint8_t *c = pl->p.glyphs->col;
        hipError_t e = launch_composite(wm, sw * 4, wm_frame_stride, n, pl->glyphs.dev, pl->glyphs.n, pl->glyphs.bbox,
                                        c[0] * 0x101u, c[1] * 0x101u, c[2] * 0x101u, c[3] * 0x101u, s);
        if (e != hipSuccess) { set_error("composite launch failed: %s", hipGetErrorString(e)); rc = IPX_ERR_HIP; }
    }
    DevSrc tsrc;
    tsrc.kind = IPX_SRC_TAP64; tsrc.pix = taps; tsrc.stride = sw * 8; tsrc.w = sw; tsrc.h = sh;
    tsrc.nframes = n; tsrc.frame_stride = tfs;
    for (int k = 0; k < 2 && !rc; k++) {   // 16-bit taps; Over onto the zeroed frame == Src
        const PlanScale &ps = pl->sc[k];
        uint8_t *o = k == 0 ? res : (crop_thumb ? nullptr : th);
        const size_t ofs = k == 0 ? resize_frame_stride : thumb_frame_stride;
        if (!o || ps.dw <= 0 || ps.dh <= 0) continue;
        rc = dev_scale_src(s, nullptr, o, ps.dw, ps.dh, ps.dw * 4, Rect{0, 0, ps.dw, ps.dh}, tsrc, ps.sr, IPX_OP_SRC, ofs);
    }
    return rc;
}
IPX_CATCH_STATUS

int ipx_plan_run_host_deep(ipx_ctx *ctx, const ipx_plan *pl, int n, int kind, const uint8_t *src, int sstride, size_t src_frame_stride,
                           uint8_t *resize_out, size_t resize_frame_stride, uint8_t *thumb_out, size_t thumb_frame_stride, uint8_t *wm_out,
                           size_t wm_frame_stride) try
{
    IPX_ENTER(ctx);
    if (kind != IPX_DEEP_NRGBA64 && kind != IPX_DEEP_RGBA64 && kind != IPX_DEEP_GRAY16 && kind != IPX_DEEP_CMYK) {
        set_error("ipx_plan_run_host_deep: unknown source type %d", kind);
        return IPX_ERR_INVALID;
    }
    const int bpp = deep_bpp(kind);
    if (!pl || n < 0 || !src || (long long)sstride < (long long)pl->p.sw * bpp) { set_error("ipx_plan_run_host_deep: bad argument"); return IPX_ERR_INVALID; }
    IPX_PLAN_SRC("ipx_plan_run_host_deep", pl, sstride, bpp);
    IPX_PLAN_SRC("ipx_plan_run_host_deep", pl, (long long)pl->p.sw * 8, 8);
    if (n == 0) return IPX_OK;
    return run_host_packed(ctx, pl, n, kDeepKind + kind, src, sstride, src_frame_stride, nullptr, resize_out, resize_frame_stride, thumb_out,
                           thumb_frame_stride, wm_out, wm_frame_stride, "ipx_plan_run_host_deep");
}
IPX_CATCH_STATUS

int ipx_plan_run_dev_paletted(ipx_ctx *ctx, void *stream, const ipx_plan *pl, int n, const uint8_t *index, int stride, size_t frame_stride,
                              const uint8_t *palettes, uint8_t *resize_out, size_t resize_frame_stride, uint8_t *thumb_out,
                              size_t thumb_frame_stride, uint8_t *wm_out, size_t wm_frame_stride) try
{
    IPX_ENTER(ctx);
    if (!pl || n < 0 || !index || !palettes || stride < pl->p.sw || ((uintptr_t)palettes & 3)) {
        set_error("ipx_plan_run_dev_paletted: bad argument");
        return IPX_ERR_INVALID;
    }
    IPX_PLAN_SRC("ipx_plan_run_dev_paletted", pl, stride, 1);
    if (n == 0) return IPX_OK;
    if (n > 65535) { set_error("ipx_plan_run_dev_paletted: at most 65535 frames per call"); return IPX_ERR_UNSUPPORTED; }
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    const int sw = pl->p.sw, sh = pl->p.sh;
    const size_t fs = align256((size_t)sw * sh * 4);
    uint8_t *nrgba = nullptr;
    IPX_HIP(hipMallocAsync((void **)&nrgba, fs * n, s));
    hipError_t e = launch_palette_expand(nrgba, fs, index, stride, frame_stride, palettes, sw, sh, n, s);
    int rc = IPX_OK;
    if (e != hipSuccess) { set_error("palette expansion failed: %s", hipGetErrorString(e)); rc = IPX_ERR_HIP; }
    if (!rc) rc = ipx_plan_run_dev_nrgba(ctx, s, pl, n, nrgba, sw * 4, fs, resize_out, resize_frame_stride, thumb_out, thumb_frame_stride, wm_out,
                                         wm_frame_stride);
    (void)hipFreeAsync(nrgba, s);
    return rc;
}
IPX_CATCH_STATUS

int ipx_plan_run_dev_gray(ipx_ctx *ctx, void *stream, const ipx_plan *pl, int n, const uint8_t *gray, int stride, size_t frame_stride,
                          uint8_t *resize_out, size_t resize_frame_stride, uint8_t *thumb_out, size_t thumb_frame_stride, uint8_t *wm_out,
                          size_t wm_frame_stride) try
{
    IPX_ENTER(ctx);
    if (!pl || n < 0 || !gray || stride < pl->p.sw) { set_error("ipx_plan_run_dev_gray: bad argument"); return IPX_ERR_INVALID; }
    IPX_PLAN_SRC("ipx_plan_run_dev_gray", pl, stride, 1);
    if (n == 0) return IPX_OK;
    if (n > 65535) { set_error("ipx_plan_run_dev_gray: at most 65535 frames per call"); return IPX_ERR_UNSUPPORTED; }
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    const int sw = pl->p.sw, sh = pl->p.sh;
    // A Gray pixel read as YCbCr with Cb = Cr = 128 converts to (y, y, y, 0xff) in both of the reference's conversions --
    // color.YCbCrToRGB: (y*0x10101) >> 16 = y; color.YCbCr.RGBA: (y*0x10101) >> 8 = y*0x101 = color.Gray.RGBA -- so the batch takes the
    // planar pass with a stride-0 row of 128s as both chroma planes: 1 byte per pixel is read and nothing is expanded in HBM.
    if (env_int("IPX_GRAY_FLAT", 1) && (size_t)(sw + 1) / 2 + 8 <= ipx_ctx::kFlatChromaBytes) {
        ipx_ycbcr_batch b;
        b.y = gray; b.cb = b.cr = ctx->flat_chroma; b.ystride = stride; b.cstride = 0;
        b.y_frame_stride = frame_stride; b.c_frame_stride = 0;
        // IPX_GRAY: the converted-tile kernel's own Gray source (the Y plane alone, y * 0x101 per channel); IPX_GRAY_SRC=0: the YCbCr
        // source with the stride-0 row of 128s as both chroma planes (the same bytes, with the chroma arithmetic)
        b.ratio = env_int("IPX_GRAY_SRC", 1) ? IPX_GRAY : IPX_YCBCR_420;
        const int rc = run_dev_ycbcr(ctx, s, pl, n, &b, true, resize_out, resize_frame_stride, thumb_out, thumb_frame_stride, wm_out,
                                     wm_frame_stride);
        if (rc <= 0) return rc;      // 1: not a shape the planar kernel takes
    }
    const size_t fs = align256((size_t)sw * sh * 4);
    uint8_t *rgba = nullptr;
    IPX_HIP(hipMallocAsync((void **)&rgba, fs * n, s));
    hipError_t e = launch_gray_expand(rgba, fs, gray, stride, frame_stride, sw, sh, n, s);
    int rc = IPX_OK;
    if (e != hipSuccess) { set_error("gray expansion failed: %s", hipGetErrorString(e)); rc = IPX_ERR_HIP; }
    if (!rc) rc = ipx_plan_run_dev(ctx, s, pl, n, rgba, sw * 4, fs, resize_out, resize_frame_stride, thumb_out, thumb_frame_stride, wm_out, wm_frame_stride);
    (void)hipFreeAsync(rgba, s);
    return rc;
}
IPX_CATCH_STATUS

int ipx_plan_run_host_ycbcr(ipx_ctx *ctx, const ipx_plan *pl, int n, const ipx_ycbcr_batch *src, uint8_t *resize_out,
                            size_t resize_frame_stride, uint8_t *thumb_out, size_t thumb_frame_stride,
                            uint8_t *wm_out, size_t wm_frame_stride) try
{
    IPX_ENTER(ctx);
    if (!pl || n < 0 || !src || !src->y || !src->cb || !src->cr || src->ratio < 0 || src->ratio > IPX_YCBCR_440) {
        set_error("ipx_plan_run_host_ycbcr: bad argument");
        return IPX_ERR_INVALID;
    }
    IPX_PLAN_SRC("ipx_plan_run_host_ycbcr", pl, src->ystride, 1);
    if (!frame_span_ok(pl->p.sw, pl->p.sh, src->cstride, 1)) { set_error("ipx_plan_run_host_ycbcr: chroma planes beyond the addressable span"); return IPX_ERR_UNSUPPORTED; }
    if (n == 0) return IPX_OK;
    const int sw = pl->p.sw, sh = pl->p.sh;
    const int cw = (src->ratio == IPX_YCBCR_422 || src->ratio == IPX_YCBCR_420) ? (sw + 1) / 2 : sw;
    const int ch = (src->ratio == IPX_YCBCR_420 || src->ratio == IPX_YCBCR_440) ? (sh + 1) / 2 : sh;
    const size_t yb = align256((size_t)sw * sh), cbb = align256((size_t)cw * ch);
    const size_t fres = resize_out ? align256(pl->info.resize_bytes) : 0, fth = thumb_out ? align256(pl->info.thumb_bytes) : 0;
    const size_t fwm = wm_out ? align256(pl->info.wm_bytes) : 0;
    LaneLease lane(ctx);
    int rc = lane_reserve(lane.get(), (yb + 2 * cbb + fres + fth + fwm) * n + 1024);
    if (rc) return rc;
    hipStream_t s = lane->stream;
    uint8_t *dy = lane->dev, *dcb = dy + yb * n, *dcr = dcb + cbb * n;
    uint8_t *dres = fres ? dcr + cbb * n : nullptr, *dth = fth ? dcr + cbb * n + fres * n : nullptr;
    uint8_t *dwm = fwm ? dcr + cbb * n + (fres + fth) * n : nullptr;
    for (int i = 0; i < n; i++) {
        IPX_HIP(hipMemcpy2DAsync(dy + yb * i, sw, src->y + src->y_frame_stride * i, src->ystride, sw, sh, hipMemcpyHostToDevice, s));
        IPX_HIP(hipMemcpy2DAsync(dcb + cbb * i, cw, src->cb + src->c_frame_stride * i, src->cstride, cw, ch, hipMemcpyHostToDevice, s));
        IPX_HIP(hipMemcpy2DAsync(dcr + cbb * i, cw, src->cr + src->c_frame_stride * i, src->cstride, cw, ch, hipMemcpyHostToDevice, s));
    }
    ipx_ycbcr_batch d;
    d.y = dy; d.cb = dcb; d.cr = dcr; d.ystride = sw; d.cstride = cw; d.y_frame_stride = yb; d.c_frame_stride = cbb;
    d.ratio = src->ratio;
    rc = ipx_plan_run_dev_ycbcr(ctx, s, pl, n, &d, dres, fres, dth, fth, dwm, fwm);
    if (rc) { (void)hipStreamSynchronize(s); return rc; }
    for (int i = 0; i < n; i++) {
        if (dres && pl->info.resize_bytes)
            IPX_HIP(hipMemcpyAsync(resize_out + resize_frame_stride * i, dres + fres * i, pl->info.resize_bytes, hipMemcpyDeviceToHost, s));
        if (dth && pl->info.thumb_bytes)
            IPX_HIP(hipMemcpyAsync(thumb_out + thumb_frame_stride * i, dth + fth * i, pl->info.thumb_bytes, hipMemcpyDeviceToHost, s));
        if (dwm && pl->info.wm_bytes)
            IPX_HIP(hipMemcpyAsync(wm_out + wm_frame_stride * i, dwm + fwm * i, pl->info.wm_bytes, hipMemcpyDeviceToHost, s));
    }
    IPX_HIP(hipStreamSynchronize(s));
    return IPX_OK;
}
IPX_CATCH_STATUS

}  // extern "C"

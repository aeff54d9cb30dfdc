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
#include <thread>
#include <utility>
#include <vector>

#include "ipx_runtime_internal.h"
#include "ipx_ks.h"

#ifndef IPX_DIAG
#define IPX_DIAG 0
#endif

namespace {

// Scale's argument checks, shared by host- and device-pointer entries (kernelScaler.Scale's preamble: adr, empty rectangles)
struct ScalePrep {
    bool empty;
    Rect adr;       // relative to dr.Min
};

int scale_prepare(int dw, int dh, const Rect &dr, int sw, int sh, const Rect &sr, int op, ScalePrep *o)
{
    if (op != IPX_OP_OVER && op != IPX_OP_SRC) { set_error("scale: unknown op %d", op); return IPX_ERR_INVALID; }
    o->empty = false;
    Rect adr = Rect{0, 0, dw, dh}.intersect(dr);
    if (adr.empty() || sr.empty() || dr.dx() <= 0 || dr.dy() <= 0) { o->empty = true; return IPX_OK; }
    if (sr.x0 < 0 || sr.y0 < 0 || sr.x1 > sw || sr.y1 > sh) {
        set_error("scale: source rectangle (%d,%d)-(%d,%d) leaves the %dx%d source; the reference's "
                  "generic Image path is not covered", sr.x0, sr.y0, sr.x1, sr.y1, sw, sh);
        return IPX_ERR_UNSUPPORTED;
    }
    o->adr = adr.shifted(-dr.x0, -dr.y0);
    return IPX_OK;
}

// One axis of the kernel scaler (newDistrib for dw destination and sw source indices) resident in HBM, by (dw, sw): the per-operation
// seam sees the same few geometries over and over.  Entries live until the context does; past kMaxKsAxes the cache is flushed after a
// device-wide wait (launches in flight hold raw pointers into it).
constexpr size_t kMaxKsAxes = 512;
int ks_axis_get(ipx_ctx *ctx, int dw, int sw, KsAxisDev *out)
{
    std::lock_guard<std::mutex> lk(ctx->ks_mu);
    auto it = ctx->ks_axes.find({dw, sw});
    if (it != ctx->ks_axes.end()) { *out = it->second.second; return IPX_OK; }
    KsAxis ax;
    if (!ks_build_axis(dw, sw, &ax)) { set_error("scale: cannot tabulate an axis of %d <- %d", dw, sw); return IPX_ERR_INVALID; }
    if (ctx->ks_axes.size() >= kMaxKsAxes) {
        IPX_HIP(hipDeviceSynchronize());
        for (auto &e : ctx->ks_axes) (void)hipFree(e.second.first);
        ctx->ks_axes.clear();
    }
    const size_t bytes = ks_axis_bytes(ax);
    std::vector<uint8_t> h(bytes);
    uint8_t *d = nullptr;
    IPX_HIP(hipMalloc((void **)&d, bytes));
    KsAxisDev dev;
    ks_axis_pack(ax, h.data(), d, &dev);
    hipError_t e = hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d); set_error("axis table upload failed: %s", hipGetErrorString(e)); return IPX_ERR_HIP; }
    ctx->ks_axes[{dw, sw}] = {d, dev};
    *out = dev;
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
    bool le_alpha = false;          // IPX_SRC_TAP64: no colour tap exceeds its alpha tap (NRGBA64, Gray16, CMYK after the expansion; not RGBA64)
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
        for (int i = 0; i < std::max(1, src.nframes); i++)
            IPX_HIP(launch_draw(d + (size_t)i * dst_fs, dstride, src.pix + (size_t)i * src.frame_stride + (size_t)spy * src.stride + (size_t)spx * 4,
                                src.stride, r.dx(), r.dy(), op, s));
    return IPX_OK;
}

// xdraw.BiLinear.Scale(dst, dr, src, sr, op, nil) on device frames.  kind_override >= 0: the tap kind to read the source with (the
// crop thumbnail's second scale, ipx_ks.h); axes: the plan's own tables, or NULL to take them from the context's cache
int dev_scale_src(ipx_ctx *ctx, hipStream_t s, int *flag, uint8_t *dst, int dw, int dh, int dstride, const Rect &dr,
                  const DevSrc &src, const Rect &sr, int op, size_t dst_fs = 0, int kind_override = -1, const KsAxisDev *axes = nullptr)
{
    ScalePrep pr;
    int rc = scale_prepare(dw, dh, dr, src.w, src.h, sr, op, &pr);
    if (rc) return rc;
    if (pr.empty) return IPX_OK;
    if (src.kind == IPX_SRC_YCBCR) op = IPX_OP_SRC;   // (*image.YCbCr).Opaque() is always true
    if (op == IPX_OP_OVER && src.kind == IPX_SRC_TAP64) IPX_HIP(launch_opaque_scan_tap64(src.pix, src.w, src.h, src.stride, flag, s));
    else if (op == IPX_OP_OVER) IPX_HIP(launch_opaque_scan(src.pix, src.w, src.h, src.stride, flag, s));  // RGBA and NRGBA: alpha scan
    KsGenArgs a{};
    if (axes) { a.ax = axes[0]; a.ay = axes[1]; }
    else {
        if ((rc = ks_axis_get(ctx, dr.dx(), sr.dx(), &a.ax))) return rc;
        if ((rc = ks_axis_get(ctx, dr.dy(), sr.dy(), &a.ay))) return rc;
    }
    a.dst = dst; a.dstride = dstride; a.src = src.pix; a.sstride = src.stride;
    a.dr_x0 = dr.x0; a.dr_y0 = dr.y0;
    a.adr_x0 = pr.adr.x0; a.adr_y0 = pr.adr.y0; a.adr_x1 = pr.adr.x1; a.adr_y1 = pr.adr.y1;
    a.sr_x0 = sr.x0; a.sr_y0 = sr.y0;
    a.op = op; a.opaque_flag = op == IPX_OP_OVER ? flag : nullptr;
    a.kind = kind_override >= 0 ? kind_override : src.kind;
    a.cb = src.cb; a.cr = src.cr; a.cstride = src.cstride; a.ratio = src.ratio;
    a.nframes = src.nframes; a.src_fs = src.frame_stride; a.c_fs = src.c_frame_stride; a.dst_fs = dst_fs;
    IPX_HIP(launch_ks_generic(a, s));
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

int dev_scale(ipx_ctx *ctx, hipStream_t s, int *flag, uint8_t *dst, int dw, int dh, int dstride, const Rect &dr,
              const uint8_t *src, int sw, int sh, int sstride, const Rect &sr, int op)
{
    return dev_scale_src(ctx, s, flag, dst, dw, dh, dstride, dr, rgba_src(src, sw, sh, sstride), sr, op);
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

static void prepare_hip_env();
int ipx_device_count(void) try
{
    prepare_hip_env();
    clear_error();
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); set_error("hipGetDeviceCount: %s", hipGetErrorString(e)); return IPX_ERR_NODEVICE; }
    return n;
}
IPX_CATCH_STATUS

// ROCclr maps HIP streams onto a fixed number of hardware queues per device (GPU_MAX_HW_QUEUES, 4 by default) and reads the variable
// when the runtime initialises.  A context opens its own stream and five lane streams: with four queues the high-priority lane that
// serves single-frame calls shared a queue with a batch lane in about one context of ten, and its calls then waited for the batch
// (tools/seam_hunt.py).  So before the first HIP call of the process the variable is set -- unless the operator has set it -- to hold
// every stream of a default context on a queue of its own.
static void prepare_hip_env()
{
    static std::once_flag once;
    std::call_once(once, [] {
        if (!getenv("GPU_MAX_HW_QUEUES")) {
            char v[16];
            snprintf(v, sizeof v, "%d", std::max(8, env_int("IPX_LANES", 5) + 3));
            setenv("GPU_MAX_HW_QUEUES", v, 0);
        }
    });
}

int ipx_create(const ipx_config *cfg, ipx_ctx **out) try
{
    clear_error();
    if (!out) { set_error("ipx_create: null out"); return IPX_ERR_INVALID; }
    *out = nullptr;
    prepare_hip_env();
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
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    // the last lane is the one a batch leaves free: single-frame calls land on it, ahead of the batch's queued work.  Its stream is
    // created first: streams take hardware queues in the order they are made
    hipError_t e = hipSuccess;
    if (c->lanes.size() >= 3) e = hipStreamCreateWithPriority(&c->lanes.back().stream, hipStreamNonBlocking, prio_hi);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    for (size_t i = 0; i < c->lanes.size(); i++) {
        Lane &l = c->lanes[i];
        if (e == hipSuccess && !l.stream) e = hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking);
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
        for (auto &ev : l.ev) if (ev) (void)hipEventDestroy(ev);
        if (l.dec) (void)hipFree(l.dec);
        if (l.flag) (void)hipFree(l.flag);
    }
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    if (c->flat_chroma) (void)hipFree(c->flat_chroma);
    for (auto &e : c->ks_axes) (void)hipFree(e.second.first);
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
    void *p = nullptr;
    if (getenv("IPX_DEBUG")) fprintf(stderr, "[ipx] pinning %zu MiB (cache holds %zu MiB in %zu blocks)\n", want >> 20, ctx->host_cached >> 20, ctx->host_free_blocks.size());
    // A fresh block is pinned on a thread bound to the CPUs next to the context's GPU (sysfs local_cpulist), so that its pages land on
    // that NUMA node: measured on the driver's two-socket box, copies from and to such blocks move 80 GB/s both directions summed against
    // 57 GB/s for blocks pinned wherever the calling thread happened to run (bench.py's pool leg against its context leg, round 3).
    // Rare: freed blocks are cached (below).  IPX_POOL_NUMA=0 switches the binding off.
    hipError_t e = hipSuccess;
    {
        std::thread t([&] {
            (void)hipSetDevice(ctx->device);
            bind_near_device(ctx->device);
            e = hipHostMalloc(&p, want, hipHostMallocDefault);
            if (e == hipSuccess && env_int("IPX_HOST_TOUCH", 1)) memset(p, 0, want);     // first touch, where the pages are to live
        });
        t.join();
    }
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
    // A device-to-device hipMemcpy returns before the copy has run, and the null stream it runs on does not order against the
    // context's non-blocking streams: a launch queued right after it could read the destination half copied (a 1024-slot test batch
    // tiled with this call did, once).  Queue it on the context's stream and wait: done on return, like the two host copies.
    if (bytes) {
        IPX_HIP(hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, ctx->stream));
        IPX_HIP(hipStreamSynchronize(ctx->stream));
    }
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

// What the host link gives pinned copies on this box, in this process: one direction alone, and both at once with the byte mix of a
// call (bench.py holds the PCIe-inclusive legs against it).  Not part of the path.
}  // extern "C"
namespace {
__global__ void link_store_kernel(uint4 *__restrict__ dst, const uint4 *__restrict__ src, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
}  // namespace
extern "C" {

int ipx_link_probe(ipx_ctx *ctx, size_t up_bytes, size_t down_bytes, int reps, double out_gbps[4]) try
{
    IPX_ENTER(ctx);
    if (!out_gbps || !up_bytes || !down_bytes || (down_bytes & 15) || reps < 1) { set_error("ipx_link_probe: bad argument"); return IPX_ERR_INVALID; }
    uint8_t *hu = (uint8_t *)ipx_host_alloc(ctx, up_bytes), *hd = (uint8_t *)ipx_host_alloc(ctx, down_bytes);
    uint8_t *du = nullptr, *dd = nullptr;
    hipStream_t s1 = nullptr, s2 = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hu && hd ? hipSuccess : hipErrorOutOfMemory;
    if (e == hipSuccess) e = hipMalloc((void **)&du, up_bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&dd, down_bytes);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) { memset(hu, 1, up_bytes); e = hipMemset(dd, 2, down_bytes); }
    uint8_t *hd_dev = e == hipSuccess ? pinned_device_view(hd, down_bytes) : nullptr;
    const size_t piece = (size_t)std::max(1, env_int("IPX_PROBE_CHUNK_MB", 32)) << 20;
    // down_by_kernel: the download as a kernel's stores into the pinned block (how run_host_packed delivers outputs) instead of a copy
    auto timed = [&](bool up, bool down, bool down_by_kernel, double *gb_up, double *gb_down) {
        double best = 1e30;
        for (int r = 0; r < reps + 1 && e == hipSuccess; r++) {           // (the first repetition warms the path)
            const auto t0 = std::chrono::steady_clock::now();
            if (down && down_by_kernel) {
                hipLaunchKernelGGL(link_store_kernel, dim3(512), dim3(256), 0, s2, (uint4 *)hd_dev, (const uint4 *)dd, down_bytes / 16);
                e = hipGetLastError();
            }
            // copies go in pieces, enqueued alternately: the runtime picks a copy engine per stream when the stream's first copy is
            // enqueued, the lowest one idle at that moment, and two large copies enqueued together land on the same engine
            for (size_t o = 0; e == hipSuccess && (o < up_bytes || o < down_bytes); o += piece) {
                if (up && o < up_bytes) e = hipMemcpyAsync(du + o, hu + o, std::min(piece, up_bytes - o), hipMemcpyHostToDevice, s1);
                if (down && !down_by_kernel && o < down_bytes && e == hipSuccess) e = hipMemcpyAsync(hd + o, dd + o, std::min(piece, down_bytes - o), hipMemcpyDeviceToHost, s2);
            }
            if (e == hipSuccess) e = hipStreamSynchronize(s1);
            if (e == hipSuccess) e = hipStreamSynchronize(s2);
            const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (r > 0) best = std::min(best, sec);
        }
        if (gb_up) *gb_up = up ? up_bytes / best / 1e9 : 0;
        if (gb_down) *gb_down = down ? down_bytes / best / 1e9 : 0;
    };
    if (e == hipSuccess) timed(true, false, false, &out_gbps[0], nullptr);
    if (e == hipSuccess) timed(false, true, false, nullptr, &out_gbps[1]);
    if (e == hipSuccess) timed(true, true, false, &out_gbps[2], &out_gbps[3]);
    if (e == hipSuccess && hd_dev) {        // the better of the two ways down, alone and beside the upload
        double u = 0, d = 0;
        timed(false, true, true, nullptr, &d);
        out_gbps[1] = std::max(out_gbps[1], d);
        if (e == hipSuccess) timed(true, true, true, &u, &d);
        if (e == hipSuccess && u + d > out_gbps[2] + out_gbps[3]) { out_gbps[2] = u; out_gbps[3] = d; }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (s1) (void)hipStreamDestroy(s1);
    if (s2) (void)hipStreamDestroy(s2);
    if (du) (void)hipFree(du);
    if (dd) (void)hipFree(dd);
    if (hu) (void)ipx_host_free(ctx, hu);
    if (hd) (void)ipx_host_free(ctx, hd);
    if (e != hipSuccess) { set_error("ipx_link_probe: %s", hipGetErrorString(e)); return e == hipErrorOutOfMemory ? IPX_ERR_NOMEM : IPX_ERR_HIP; }
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
    int rc = dev_scale(ctx, s, flag, dst, dw, dh, dstride, to_rect(dr), src, sw, sh, sstride, to_rect(sr), op);
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
    // Over reads the destination; Src leaves pixels outside adr untouched
    IPX_HIP(hipMemcpy2DAsync(ddst, (size_t)dw * 4, dst, dstride, (size_t)dw * 4, dh, hipMemcpyHostToDevice, s));
    rc = dev_scale(ctx, s, lane->flag, ddst, dw, dh, dw * 4, to_rect(dr), dsrc, sw, sh, sw * 4, to_rect(sr), op);
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
        return dev_scale_src(ctx, s, flag, ddst, dw, dh, dw * 4, to_rect(dr), rgba_src(dsrc, sw, sh, sw * 4, IPX_SRC_NRGBA), to_rect(sr), op);
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
        return r2 ? r2 : dev_scale_src(ctx, s, flag, ddst, dw, dh, dw * 4, to_rect(dr), t, to_rect(sr), op);
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
        return dev_scale_src(ctx, s, flag, ddst, dw, dh, dw * 4, to_rect(dr), d, to_rect(sr), IPX_OP_SRC);
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

    // newDistrib of both axes of every scaled output (an output with a zero dimension -- resize.go:70-72 has no guard -- is simply empty)
    std::vector<uint8_t> blob;
    auto put = [&](size_t bytes) {
        const size_t off = (blob.size() + 15) & ~(size_t)15;
        blob.resize(off + bytes);
        return off;
    };
    size_t axoff[2][2] = {{0, 0}, {0, 0}};
    bool have[2] = {false, false};
    for (int k = 0; k < 2; k++) {
        PlanScale &sk = pl->sc[k];
        if (!sk.on || sk.dw <= 0 || sk.dh <= 0) continue;
        if (!ks_build_axis(sk.dw, sk.sr.dx(), &sk.hx) || !ks_build_axis(sk.dh, sk.sr.dy(), &sk.hy)) {
            set_error("ipx_plan_create: cannot tabulate the axes of a %dx%d <- %dx%d scale", sk.dw, sk.dh, sk.sr.dx(), sk.sr.dy());
            delete pl;
            return IPX_ERR_INVALID;
        }
        have[k] = true;
        axoff[k][0] = put(ks_axis_bytes(sk.hx));
        axoff[k][1] = put(ks_axis_bytes(sk.hy));
    }
    // the one-pass kernel's tiling and tables (source pixels of 4 bytes in LDS)
    KsFusedIn fin[2];
    for (int k = 0; k < 2; k++) {
        const PlanScale &sk = pl->sc[k];
        fin[k].dw = sk.dw; fin[k].dh = sk.dh; fin[k].sr_x0 = sk.sr.x0; fin[k].sr_y0 = sk.sr.y0; fin[k].hx = &sk.hx; fin[k].hy = &sk.hy;
    }
    size_t fused_off[3] = {0, 0, 0};
    if (!env_int("IPX_NO_FUSE", 0)) {
        const int px_bytes[3] = {4, 8, 2};
        for (int f = 0; f < 3; f++) {
            std::vector<uint8_t> fblob;
            fin[1].top_taps = f == 1 && pl->p.crop_to_fit;       // (the YCbCr and NRGBA sources' crop thumbnails: IPX_SRC_YCBCR_CROP / IPX_SRC_NRGBA_CROP taps)
            if (ks_fused_plan(sw, sh, have[0] ? &fin[0] : nullptr, have[1] ? &fin[1] : nullptr, px_bytes[f], &fblob, &pl->fused[f])) {
                fused_off[f] = put(fblob.size());
                memcpy(blob.data() + fused_off[f], fblob.data(), fblob.size());
            }
        }
    }
    if (!blob.empty()) {
        hipError_t e = hipMalloc((void **)&pl->blob, blob.size());
        if (e != hipSuccess) {
            set_error("plan table allocation failed: %s", hipGetErrorString(e));
            ipx_plan_destroy(ctx, pl);
            return IPX_ERR_NOMEM;
        }
        for (int k = 0; k < 2; k++) {
            if (!have[k]) continue;
            ks_axis_pack(pl->sc[k].hx, blob.data() + axoff[k][0], pl->blob + axoff[k][0], &pl->sc[k].ax[0]);
            ks_axis_pack(pl->sc[k].hy, blob.data() + axoff[k][1], pl->blob + axoff[k][1], &pl->sc[k].ax[1]);
        }
        for (int f = 0; f < 3; f++)
            if (pl->fused[f].ok) ks_fused_rebase(&pl->fused[f], pl->blob + fused_off[f]);
        e = hipMemcpy(pl->blob, blob.data(), blob.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            set_error("plan table upload failed: %s", hipGetErrorString(e));
            ipx_plan_destroy(ctx, pl);
            return IPX_ERR_HIP;
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
    if (plan->owned_gs) ipx_glyphset_destroy(ctx, plan->owned_gs);
    delete plan;
}

// ---- plans by content --------------------------------------------------------------------------------------------------------
// The per-operator entries (ipx_*_process, ipx_processor_process) and the pool describe their operators per call; building a glyph set
// and a plan per call cost three hipMalloc / hipFree pairs, each of which waits for every stream of the device.  The context keeps
// plans (with their glyph sets) by content instead: operator parameters, colour, and every glyph's rectangle and mask bytes.
// The cache holds at most IPX_PLAN_CACHE_MAX (256) plans; a new content then takes the place of the plan that was handed out longest
// ago and that no call holds (a worker fed arbitrary upload sizes would otherwise fill the cache with sizes it never sees again, and
// sizes that become frequent later could not enter).
static size_t max_cached_plans() { return (size_t)std::max(1, env_int("IPX_PLAN_CACHE_MAX", 256)); }

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
    if (use_cache && it != ctx->plan_cache.end()) {
        *plan = it->second.second; *cached = 1;
        (*plan)->cache_refs++; (*plan)->cache_stamp = ++ctx->plan_clock;
        return IPX_OK;
    }
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
    if (use_cache && ctx->plan_cache.size() >= max_cached_plans()) {   // make room: the least recently used plan nobody holds
        auto victim = ctx->plan_cache.end();
        for (auto i2 = ctx->plan_cache.begin(); i2 != ctx->plan_cache.end(); ++i2)
            if (i2->second.second->cache_refs == 0 && (victim == ctx->plan_cache.end() || i2->second.second->cache_stamp < victim->second.second->cache_stamp))
                victim = i2;
        if (victim != ctx->plan_cache.end()) {
            ipx_plan_destroy(ctx, victim->second.second);               // (hipFree waits for the launches that still read its tables)
            ctx->plan_cache.erase(victim);
        }
    }
    if (use_cache && ctx->plan_cache.size() < max_cached_plans()) {
        pl->cache_refs = 1; pl->cache_stamp = ++ctx->plan_clock;
        ctx->plan_cache.emplace(std::move(key), std::make_pair(gs, pl));
        *cached = 1;
    }
    *plan = pl;
    return IPX_OK;
}
IPX_CATCH_STATUS

void ipx_plan_release(ipx_ctx *ctx, ipx_plan *plan, int cached)
{
    if (!plan) return;
    if (!cached) { ipx_plan_destroy(ctx, plan); return; }   // a plan the cache did not take (every cached plan was in use) lives for one call
    if (!ctx) return;
    std::lock_guard<std::mutex> lk(ctx->plan_mu);
    if (plan->cache_refs > 0) plan->cache_refs--;
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

// ---- one batch of decoded frames of any source type through a plan --------------------------------------------------------------------
// The one-pass kernel where it is built for the source type and the shapes; else per output: draw.Draw into the watermark frames
// (launch_draw*), the text, and one generic Scale per scaled output -- the crop thumbnail (thumbnail.go:128-131) reads the source
// through the `_CROP` tap kind of its type (ipx_ks.h), which is its 8-bit crop copy without the copy.
static hipError_t composite_text(const ipx_plan *pl, uint8_t *wm, size_t wm_frame_stride, int n, hipStream_t s)
{
    if (!wm || pl->glyphs.n <= 0 || !pl->p.glyphs) return hipSuccess;
    const uint8_t *c = pl->p.glyphs->col;
    return launch_composite(wm, pl->p.sw * 4, wm_frame_stride, n, pl->glyphs.dev, pl->glyphs.n, pl->glyphs.bbox, c[0] * 0x101u, c[1] * 0x101u,
                            c[2] * 0x101u, c[3] * 0x101u, s);
}

static int run_dev_any(ipx_ctx *ctx, hipStream_t s, const ipx_plan *pl, int n, const DevSrc &src, uint8_t *resize_out, size_t resize_frame_stride,
                       uint8_t *thumb_out, size_t thumb_frame_stride, uint8_t *wm_out, size_t wm_frame_stride)
{
    const int sw = pl->p.sw, sh = pl->p.sh;
    uint8_t *outs[2] = {pl->sc[0].on ? resize_out : nullptr, pl->sc[1].on ? thumb_out : nullptr};
    const size_t ostr[2] = {resize_frame_stride, thumb_frame_stride};
    uint8_t *wm = pl->p.do_watermark ? wm_out : nullptr;
    for (int k = 0; k < 2; k++)
        if (pl->sc[k].dw <= 0 || pl->sc[k].dh <= 0) outs[k] = nullptr;
    if (!wm && !outs[0] && !outs[1]) return IPX_OK;
    int kinds[2] = {src.kind, pl->p.crop_to_fit ? ks_crop_kind(src.kind) : src.kind};

    const bool gray = src.kind == IPX_SRC_YCBCR && src.cstride == 0;
    const KsFusedPlan &fp = pl->fused[src.kind == IPX_SRC_RGBA ? 0 : gray ? 2 : 1];
    if (fp.ok && env_int("IPX_FUSED", 1)) {
        KsFusedArgs a{};
        a.src = src.pix; a.src_fs = src.frame_stride; a.sstride = src.stride; a.sw = sw; a.sh = sh;
        a.src_kind = src.kind; a.cb = src.cb; a.cr = src.cr; a.cstride = src.cstride; a.ratio = src.ratio; a.c_fs = src.c_frame_stride;
        a.wm = wm; a.wm_fs = wm_frame_stride; a.wm_stride = sw * 4;
        a.nframes = n;
        a.taps_le_alpha = src.le_alpha;
        for (int k = 0; k < 2; k++) {
            if (!outs[k]) continue;
            const PlanScale &ps = pl->sc[k];
            KsFusedOut &o = a.o[a.nout++];
            o.out = outs[k]; o.frame_stride = ostr[k]; o.ostride = ps.dw * 4; o.obytes = ps.dw * ps.dh * 4;
            o.dw = ps.dw; o.dh = ps.dh; o.sr_x0 = ps.sr.x0; o.sr_y0 = ps.sr.y0;
            o.kind = kinds[k]; o.pk = k;
        }
        // RGBA frames: the speculative opaque pass first (IPX_KS_SPEC=0: the general kernel alone), one flag per item.  RGBA, YCbCr and
        // Gray frames: the float pass (IPX_KS_FAST=0: float64 throughout) with a list per frame for the pixels it cannot decide --
        // 1 / 96 of the output's pixels each, twenty times what photographs put there (IPX_KS_STATS=1 prints the fill); a frame that fills
        // its list is redone in float64.
        // One stream-ordered block: [flags][counts][lists].
        int *redo = nullptr;
        KsFix fixv, *fix = nullptr;
        const int max_items = n * fp.nstrips * std::max(fp.whole.nseg, fp.split.nseg);
        const bool spec = src.kind == IPX_SRC_RGBA && env_int("IPX_KS_SPEC", 1);
        // (three more launches and a memset; measured faster than float64 throughout at every size from one 640x360 frame to 8K frames)
        const bool fast = (spec || src.kind == IPX_SRC_YCBCR || src.kind == IPX_SRC_NRGBA || (src.kind == IPX_SRC_TAP64 && src.le_alpha)) &&
                          env_int("IPX_KS_FAST", 1) != 0;
        if (spec || fast) {
            const int cap_env = env_int("IPX_KS_FIX_CAP", 0);     // test knob: tiny lists, so that frames fill them
            int cap[2] = {0, 0};
            for (int k = 0; k < 2; k++)
                if (fast && outs[k]) cap[k] = cap_env > 0 ? cap_env : (int)std::min<size_t>(std::max<size_t>((size_t)pl->sc[k].dw * pl->sc[k].dh / 96, 256), (size_t)1 << 20);
            const size_t flags = align256((size_t)max_items * sizeof(int)), counts = fast ? align256((size_t)n * 2 * sizeof(int)) : 0;
            IPX_HIP(hipMallocAsync((void **)&redo, flags + counts + (size_t)n * (cap[0] + cap[1]) * sizeof(uint2), s));
            if (fast) {
                fixv.count = (int *)((uint8_t *)redo + flags);
                fixv.list = (uint2 *)((uint8_t *)redo + flags + counts);
                fixv.cap[0] = cap[0]; fixv.cap[1] = cap[1];
                for (int k = 0; k < 2; k++) { fixv.ax[k] = pl->sc[k].ax[0]; fixv.ay[k] = pl->sc[k].ax[1]; }
                IPX_HIP(hipMemsetAsync(fixv.count, 0, (size_t)n * 2 * sizeof(int), s));
                fix = &fixv;
            }
        }
        a.redo = redo;
#if IPX_DIAG
        static unsigned long long *stamp_buf = nullptr;   // IPX_STAMPS=1: phase stamps, never in a timed run
        if (env_int("IPX_STAMPS", 0)) {
            if (!stamp_buf) IPX_HIP(hipMalloc((void **)&stamp_buf, 24 * sizeof(unsigned long long)));
            IPX_HIP(hipMemsetAsync(stamp_buf, 0, 24 * sizeof(unsigned long long), s));
            a.stamps = stamp_buf;
        }
#endif
        bool matched = false;
        hipError_t e = launch_ks_fused(fp, a, fix, ctx->cus, s, &matched);
        if (fix && matched && e == hipSuccess && env_int("IPX_KS_STATS", 0)) {   // diagnostic: how full the float pass's lists got, how many items went to float64
            std::vector<int> cnt((size_t)n * 2), flags((size_t)max_items);
            (void)hipMemcpyAsync(cnt.data(), fixv.count, cnt.size() * sizeof(int), hipMemcpyDeviceToHost, s);
            (void)hipMemcpyAsync(flags.data(), redo, flags.size() * sizeof(int), hipMemcpyDeviceToHost, s);
            (void)hipStreamSynchronize(s);
            long long tot[2] = {0, 0}; int mx[2] = {0, 0}, nredo = 0;
            for (int i = 0; i < n; i++) for (int k = 0; k < 2; k++) { tot[k] += cnt[2 * i + k]; mx[k] = std::max(mx[k], cnt[2 * i + k]); }
            for (int f : flags) nredo += f != 0;
            fprintf(stderr, "[ipx ks stats] %d frames: undecided pixels per frame resize mean %.1f max %d (room %d), thumbnail mean %.1f max %d (room %d); %d of %d items redone in float64\n", n,
                    (double)tot[0] / n, mx[0], fixv.cap[0], (double)tot[1] / n, mx[1], fixv.cap[1], nredo, max_items);
        }
        if (redo) (void)hipFreeAsync(redo, s);
#if IPX_DIAG
        if (a.stamps && matched) {
            unsigned long long h[24];
            IPX_HIP(hipMemcpyAsync(h, a.stamps, sizeof h, hipMemcpyDeviceToHost, s));
            IPX_HIP(hipStreamSynchronize(s));
            const char *who[3] = {"resize waves", "thumbnail waves", "idle waves"};
            for (int r = 0; r < 3; r++) {
                double tot = 0;
                for (int i = 0; i < 6; i++) tot += (double)h[r * 8 + i];
                if (tot > 0)
                    fprintf(stderr, "[ipx stamps] %-16s share of wave time: wait-loads %.1f%%  drain %.1f%%  barrier %.1f%%  issue %.1f%%  scaleX %.1f%%  scaleY %.1f%%  (%.3g cycles)\n", who[r],
                            100 * h[r * 8] / tot, 100 * h[r * 8 + 1] / tot, 100 * h[r * 8 + 2] / tot, 100 * h[r * 8 + 3] / tot, 100 * h[r * 8 + 4] / tot, 100 * h[r * 8 + 5] / tot, tot);
            }
        }
#endif
        if (e != hipSuccess) { set_error("one-pass kernel launch failed: %s", hipGetErrorString(e)); return IPX_ERR_HIP; }
        if (matched) {
            IPX_HIP(composite_text(pl, wm, wm_frame_stride, n, s));
            return IPX_OK;
        }
    }

    if (wm) {   // draw.Draw(result, bounds, img, ZP, draw.Src) for the batch
        const int rc = dev_draw_src(s, wm, sw, sh, sw * 4, Rect{0, 0, sw, sh}, src, 0, 0, IPX_OP_SRC, wm_frame_stride);
        if (rc) return rc;
        IPX_HIP(composite_text(pl, wm, wm_frame_stride, n, s));
    }
    for (int k = 0; k < 2; k++) {   // Over onto the zeroed frame of image.NewRGBA == Src
        if (!outs[k]) continue;
        const PlanScale &ps = pl->sc[k];
        const int rc = dev_scale_src(ctx, s, nullptr, outs[k], ps.dw, ps.dh, ps.dw * 4, Rect{0, 0, ps.dw, ps.dh}, src, ps.sr, IPX_OP_SRC, ostr[k],
                                     kinds[k], ps.ax);
        if (rc) return rc;
    }
    return IPX_OK;
}

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
    if (n > 65535) { set_error("ipx_plan_run_dev: at most 65535 frames per call"); return IPX_ERR_UNSUPPORTED; }
    DevSrc d = rgba_src(src, pl->p.sw, pl->p.sh, sstride);
    d.nframes = n; d.frame_stride = src_frame_stride;
    return run_dev_any(ctx, stream ? (hipStream_t)stream : ctx->stream, pl, n, d, resize_out, resize_frame_stride, thumb_out, thumb_frame_stride,
                       wm_out, wm_frame_stride);
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
    // The host link moves both directions at once only when uploads and downloads sit on streams of their own AND go in pieces: the
    // runtime picks a copy engine per hipMemcpyAsync, and two large copies enqueued together share one (tools/link_streams.py,
    // ipx_link_probe: 2 x 512 MiB as single copies 28.7 + 28.7 GB/s, in 32 MiB pieces 47.9 + 47.9).  So the call is a three-stage
    // pipeline over the lanes it took: every upload on the first lane's stream, every kernel on the third's, every download on the
    // second's, the lanes' scratch buffers as the slots chunks rotate through, events between the stages.
    const size_t piece = (size_t)std::max(1, env_int("IPX_COPY_PIECE_MB", 32)) << 20;
    auto copy_pieces = [&](uint8_t *dst, const uint8_t *from, size_t bytes, hipMemcpyKind k, hipStream_t st) {
        hipError_t r = hipSuccess;
        for (size_t o = 0; o < bytes && r == hipSuccess; o += piece) r = hipMemcpyAsync(dst + o, from + o, std::min(piece, bytes - o), k, st);
        return r;
    };
    auto d2h = [&](uint8_t *host, size_t host_stride, const uint8_t *dev, size_t dev_stride, size_t bytes, int i0, int m,
                   hipStream_t st) {
        if (!dev || !bytes) return hipSuccess;
        if (host_stride == dev_stride)  // tight on both sides: the chunk as one run of bytes
            return copy_pieces(host + (size_t)i0 * host_stride, dev, dev_stride * (m - 1) + bytes, hipMemcpyDeviceToHost, st);
        hipError_t r = hipSuccess;
        for (int i = 0; i < m && r == hipSuccess; i++)
            r = hipMemcpyAsync(host + (size_t)(i0 + i) * host_stride, dev + dev_stride * i, bytes, hipMemcpyDeviceToHost, st);
        return r;
    };
    // Outputs in pinned memory (hipHostMalloc / hipHostRegister: mapped into the device's address space) are written by the kernels
    // themselves, over the link, instead of into the lane's scratch and from there by a copy engine: uploads are then the only copies
    // in flight, so the two directions cannot land on one engine, and a chunk has two stages instead of three.
    uint8_t *vres = nullptr, *vth = nullptr, *vwm = nullptr;
    bool direct = !bounce && env_int("IPX_HOST_DIRECT", 1) != 0;
    // (the kernels store 16 bytes per lane: frames whose stride would break that alignment go through the lane's scratch, which pads)
    if (((uintptr_t)resize_out | (uintptr_t)thumb_out | (uintptr_t)wm_out | (resize_out ? resize_frame_stride : 0) | (thumb_out ? thumb_frame_stride : 0) |
         (wm_out ? wm_frame_stride : 0)) & 15) direct = false;
    if (direct && resize_out) direct = (vres = pinned_device_view(resize_out, resize_frame_stride * (n - 1) + pl->info.resize_bytes)) != nullptr;
    if (direct && thumb_out) direct = (vth = pinned_device_view(thumb_out, thumb_frame_stride * (n - 1) + pl->info.thumb_bytes)) != nullptr;
    if (direct && wm_out) direct = (vwm = pinned_device_view(wm_out, wm_frame_stride * (n - 1) + pl->info.wm_bytes)) != nullptr;
    const size_t S = lanes.size();
    const hipStream_t s_up = lanes[0]->stream, s_down = lanes[S > 1 ? 1 : 0]->stream, s_run = lanes[S > 2 ? 2 : 0]->stream;
    const bool staged = S > 1 && env_int("IPX_HOST_STAGED", 1) != 0;
    if (staged)
        for (auto *l : lanes)
            for (auto &ev : l->ev)
                if (!ev && e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    for (int i0 = 0, c = 0; !rc && e == hipSuccess && i0 < n; i0 += chunk, c++) {
        Lane &l = *lanes[c % S];
        const hipStream_t up = staged ? s_up : l.stream, run = staged ? s_run : l.stream, down = staged ? s_down : l.stream;
        const int m = std::min(chunk, n - i0);
        uint8_t *dsrc = (uint8_t *)(((uintptr_t)l.dev + 255) & ~(uintptr_t)255);
        uint8_t *dres = fres ? dsrc + fsrc * chunk : nullptr;
        uint8_t *dth = fth ? dsrc + (fsrc + fres) * chunk : nullptr;
        uint8_t *dwm = fwm ? dsrc + (fsrc + fres + fth) * chunk : nullptr;
        uint8_t *dpal = fpal ? dsrc + (fsrc + fres + fth + fwm) * chunk : nullptr;
        size_t sres = fres, sth = fth, swm = fwm;
        if (direct) {
            dres = fres ? vres + (size_t)i0 * resize_frame_stride : nullptr; sres = resize_frame_stride;
            dth = fth ? vth + (size_t)i0 * thumb_frame_stride : nullptr; sth = thumb_frame_stride;
            dwm = fwm ? vwm + (size_t)i0 * wm_frame_stride : nullptr; swm = wm_frame_stride;
        }
        // reuse of a lane's scratch: chunk c waits until chunk c - S has been downloaded (stream order when one stream does it all)
        if (staged && c >= (int)S) e = hipStreamWaitEvent(up, l.ev[2], 0);
        if (e != hipSuccess) break;
        if (bounce) {
            for (int y = 0; y < sh; y++) memcpy(l.pin + (size_t)y * sw * bpp, src + (size_t)y * sstride, (size_t)sw * bpp);
            e = hipMemcpyAsync(dsrc, l.pin, (size_t)sw * sh * bpp, hipMemcpyHostToDevice, up);
        } else if (src_tight) {
            e = copy_pieces(dsrc, src + (size_t)i0 * src_frame_stride, fsrc * m, hipMemcpyHostToDevice, up);
        } else {
            for (int i = 0; i < m && e == hipSuccess; i++)
                e = hipMemcpy2DAsync(dsrc + fsrc * i, (size_t)sw * bpp, src + (size_t)(i0 + i) * src_frame_stride, sstride,
                                     (size_t)sw * bpp, sh, hipMemcpyHostToDevice, up);
        }
        if (e == hipSuccess && dpal) e = hipMemcpyAsync(dpal, palettes + (size_t)i0 * 1024, (size_t)m * 1024, hipMemcpyHostToDevice, up);
        if (e == hipSuccess && staged) e = hipEventRecord(l.ev[0], up);
        if (e == hipSuccess && staged) e = hipStreamWaitEvent(run, l.ev[0], 0);
        if (e != hipSuccess) break;
        switch (kind) {
        case IPX_SRC_RGBA: rc = ipx_plan_run_dev(ctx, run, pl, m, dsrc, sw * 4, fsrc, dres, sres, dth, sth, dwm, swm); break;
        case IPX_SRC_NRGBA: rc = ipx_plan_run_dev_nrgba(ctx, run, pl, m, dsrc, sw * 4, fsrc, dres, sres, dth, sth, dwm, swm); break;
        case IPX_GRAY: rc = ipx_plan_run_dev_gray(ctx, run, pl, m, dsrc, sw, fsrc, dres, sres, dth, sth, dwm, swm); break;
        case kPalettedKind: rc = ipx_plan_run_dev_paletted(ctx, run, pl, m, dsrc, sw, fsrc, dpal, dres, sres, dth, sth, dwm, swm); break;
        default: rc = ipx_plan_run_dev_deep(ctx, run, pl, m, kind - kDeepKind, dsrc, sw * bpp, fsrc, dres, sres, dth, sth, dwm, swm); break;
        }
        if (rc) break;
        if (direct) {          // the outputs are where they belong when the kernels have finished
            if (staged) e = hipEventRecord(l.ev[2], run);
            continue;
        }
        if (staged) {
            e = hipEventRecord(l.ev[1], run);
            if (e == hipSuccess) e = hipStreamWaitEvent(down, l.ev[1], 0);
            if (e != hipSuccess) break;
        }
        if (bounce) {          // one copy of the three outputs (they follow the source in the lane's scratch), handed out after the sync below
            if (fres + fth + fwm) e = hipMemcpyAsync(l.pin + fsrc, dsrc + fsrc, fres + fth + fwm, hipMemcpyDeviceToHost, down);
            continue;
        }
        e = d2h(resize_out, resize_frame_stride, dres, fres, pl->info.resize_bytes, i0, m, down);
        if (e == hipSuccess) e = d2h(thumb_out, thumb_frame_stride, dth, fth, pl->info.thumb_bytes, i0, m, down);
        if (e == hipSuccess) e = d2h(wm_out, wm_frame_stride, dwm, fwm, pl->info.wm_bytes, i0, m, down);
        if (e == hipSuccess && staged) e = hipEventRecord(l.ev[2], down);
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
    DevSrc d;
    d.kind = IPX_SRC_YCBCR; d.pix = src->y; d.stride = src->ystride; d.cb = src->cb; d.cr = src->cr;
    d.cstride = src->cstride; d.ratio = src->ratio; d.w = pl->p.sw; d.h = pl->p.sh;
    d.nframes = n; d.frame_stride = src->y_frame_stride; d.c_frame_stride = src->c_frame_stride;
    return run_dev_any(ctx, stream ? (hipStream_t)stream : ctx->stream, pl, n, d, resize_out, resize_frame_stride, thumb_out, thumb_frame_stride,
                       wm_out, wm_frame_stride);
}
IPX_CATCH_STATUS

int ipx_plan_run_dev_nrgba(ipx_ctx *ctx, void *stream, const ipx_plan *pl, int n, const uint8_t *src, int sstride, size_t src_frame_stride,
                           uint8_t *resize_out, size_t resize_frame_stride, uint8_t *thumb_out, size_t thumb_frame_stride, uint8_t *wm_out,
                           size_t wm_frame_stride) try
{
    IPX_ENTER(ctx);
    if (!pl || n < 0 || !src || (long long)sstride < (long long)pl->p.sw * 4) { set_error("ipx_plan_run_dev_nrgba: bad argument"); return IPX_ERR_INVALID; }
    IPX_PLAN_SRC("ipx_plan_run_dev_nrgba", pl, sstride, 4);
    if (n == 0) return IPX_OK;
    if (n > 65535) { set_error("ipx_plan_run_dev_nrgba: at most 65535 frames per call"); return IPX_ERR_UNSUPPORTED; }
    DevSrc d = rgba_src(src, pl->p.sw, pl->p.sh, sstride, IPX_SRC_NRGBA);
    d.nframes = n; d.frame_stride = src_frame_stride;
    return run_dev_any(ctx, stream ? (hipStream_t)stream : ctx->stream, pl, n, d, resize_out, resize_frame_stride, thumb_out, thumb_frame_stride,
                       wm_out, wm_frame_stride);
}
IPX_CATCH_STATUS

// The deep source types: one expansion pass to frames of 16-bit taps (what every consumer of these types reads: At(x, y).RGBA()), then
// the batch runner on them: top bytes into the watermark frames (drawRGBA / drawCMYK with Src), the scales from the taps.
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
    const size_t tfs = align256((size_t)sw * sh * 8);
    uint8_t *taps = nullptr;
    IPX_HIP(hipMallocAsync((void **)&taps, tfs * n, s));
    struct Free { uint8_t *p; hipStream_t s; ~Free() { if (p) (void)hipFreeAsync(p, s); } } free_taps{taps, s};
    {
        hipError_t e = launch_deep_expand(taps, tfs, src, sstride, src_frame_stride, kind, sw, sh, n, s);
        if (e != hipSuccess) { set_error("tap expansion failed: %s", hipGetErrorString(e)); return IPX_ERR_HIP; }
    }
    DevSrc d;
    d.kind = IPX_SRC_TAP64; d.pix = taps; d.stride = sw * 8; d.w = sw; d.h = sh;
    d.nframes = n; d.frame_stride = tfs;
    d.le_alpha = kind != IPX_DEEP_RGBA64;     // (a premultiplied RGBA64 file may hold a colour above its alpha; the others cannot)
    return run_dev_any(ctx, s, pl, n, d, resize_out, resize_frame_stride, thumb_out, thumb_frame_stride, wm_out, wm_frame_stride);
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
    // scaleX_Gray / drawGray read a pixel as (y, y, y, 0xff), tmp alpha 1.  A YCbCr pixel with Cb = Cr = 128 converts to exactly that in both of the
    // reference's conversions -- color.YCbCrToRGB: (y*0x10101) >> 16 = y; color.YCbCr.RGBA: (y*0x10101) >> 8 = y*0x101 = color.Gray.RGBA -- and
    // scaleX_YCbCr4xx writes the same tmp alpha, so the batch runs as 4:4:4 planes with a stride-0 row of 128s as both chroma planes.
    if ((size_t)pl->p.sw + 8 > ipx_ctx::kFlatChromaBytes) { set_error("ipx_plan_run_dev_gray: frames wider than %zu pixels", ipx_ctx::kFlatChromaBytes - 8); return IPX_ERR_UNSUPPORTED; }
    DevSrc d;
    d.kind = IPX_SRC_YCBCR; d.pix = gray; d.stride = stride; d.cb = d.cr = ctx->flat_chroma;
    d.cstride = 0; d.ratio = IPX_YCBCR_444; d.w = pl->p.sw; d.h = pl->p.sh;
    d.nframes = n; d.frame_stride = frame_stride; d.c_frame_stride = 0;
    return run_dev_any(ctx, stream ? (hipStream_t)stream : ctx->stream, pl, n, d, resize_out, resize_frame_stride, thumb_out, thumb_frame_stride,
                       wm_out, wm_frame_stride);
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

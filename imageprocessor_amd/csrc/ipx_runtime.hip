// ipx_runtime.hip -- context, staging lanes, glyph sets, plans and the C ABI of include/ipx.h.
//
// Threading model (worker.go:88-96: WORKER_CONCURRENCY goroutines share one processor): a
// context is safe to call from any number of OS threads.  Host-pointer calls borrow one of
// `lanes` staging lanes (stream + device scratch), blocking only when all lanes are busy;
// device-pointer calls are plain asynchronous launches on the caller's stream.
#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <utility>
#include <vector>

#include "ipx_internal.h"

#ifndef IPX_DIAG
#define IPX_DIAG 0
#endif

using namespace ipx;

// ---------------------------------------------------------------------------------------------
struct Lane {
    hipStream_t stream = nullptr;
    uint8_t *dev = nullptr;   // device scratch
    size_t dev_bytes = 0;
    int *flag = nullptr;      // device int for the opaque() scan
    bool busy = false;
};

struct ipx_ctx {
    int device = 0;
    int cus = 256;                 // compute units of the device
    hipStream_t stream = nullptr;  // default stream for device-pointer calls
    std::vector<Lane> lanes;
    size_t lane_bytes = 0;
    std::mutex mu;
    std::condition_variable cv;
    // pinned host blocks: hipHostMalloc / hipHostFree cost milliseconds each, and the encoder hands out one block per
    // batch and output, so freed blocks are kept (up to host_cache_limit bytes) and reused for requests they fit
    std::mutex host_mu;
    std::map<void *, size_t> host_size;            // every live block handed out by ipx_host_alloc
    std::multimap<size_t, void *> host_free_blocks;
    size_t host_cached = 0, host_cache_limit = (size_t)2 << 30;
};

struct GlyphHost {
    size_t mask_off;  // offset of this glyph's mask in the packed blob
    int mw, mh;
    Rect dr;
    int mpx, mpy;
};

struct ClippedGlyphs {
    DevGlyph *dev = nullptr;
    int n = 0;
    Rect bbox{0, 0, 0, 0};
};

struct ipx_glyphset {
    int device = 0;
    std::vector<GlyphHost> g;
    uint8_t *masks_dev = nullptr;
    size_t masks_bytes = 0;
    uint8_t col[4] = {0, 0, 0, 0};
    mutable std::mutex mu;
    mutable std::map<std::pair<int, int>, ClippedGlyphs> clipped;  // per frame size
};

struct PlanScale {
    bool on = false;
    int dw = 0, dh = 0;
    Rect sr{0, 0, 0, 0};
    AxisTap *xt = nullptr, *yt = nullptr;
    int *row_begin = nullptr, *col_begin = nullptr;
    int dyadic_shift = -1;
    int kx = -1, ky = -1;   // dyadic bits per axis (dyadic_shift = kx + ky), -1 = not dyadic
};

struct ipx_plan {
    ipx_plan_params p{};
    ipx_plan_info info{};
    bool fused = false;
    int band_rows = 0, blk_cols = 0, nbands = 0, ncolblk = 0;
    int nx_out[2] = {0, 0}; // per output: ceil(widest column block / 256)
    int most_rows = 0;    // most destination rows any band owns, over the scaled outputs
    PlanScale sc[2];      // 0 = resize, 1 = thumbnail
    uint8_t *blob = nullptr;
    ClippedGlyphs glyphs;
    mutable std::mutex mu;
    mutable ipx_plan *thumb_only = nullptr;   // RGBA sub-plan for YCbCr batches (thumbnail of the converted frame)
};

namespace {

int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

struct DeviceGuard {  // hipSetDevice is per-thread; callers may arrive on any OS thread
    explicit DeviceGuard(int dev) { ok = hipSetDevice(dev) == hipSuccess; }
    bool ok;
};

#define IPX_ENTER(ctx)                                                         \
    clear_error();                                                             \
    if (!(ctx)) { set_error("%s: null context", __func__); return IPX_ERR_INVALID; } \
    DeviceGuard guard_((ctx)->device);                                         \
    if (!guard_.ok) { set_error("hipSetDevice(%d) failed", (ctx)->device); return IPX_ERR_HIP; }

class LaneLease {
public:
    explicit LaneLease(ipx_ctx *c) : c_(c)
    {
        std::unique_lock<std::mutex> lk(c->mu);
        c->cv.wait(lk, [&] {
            for (auto &l : c->lanes) if (!l.busy) return true;
            return false;
        });
        for (auto &l : c->lanes) if (!l.busy) { l.busy = true; lane_ = &l; break; }
    }
    ~LaneLease()
    {
        {
            std::lock_guard<std::mutex> lk(c_->mu);
            lane_->busy = false;
        }
        c_->cv.notify_one();
    }
    Lane *operator->() { return lane_; }
    Lane &get() { return *lane_; }
private:
    ipx_ctx *c_;
    Lane *lane_ = nullptr;
};

int lane_reserve(Lane &l, size_t bytes)
{
    if (bytes <= l.dev_bytes) return IPX_OK;
    if (l.dev) { IPX_HIP(hipStreamSynchronize(l.stream)); IPX_HIP(hipFree(l.dev)); l.dev = nullptr; l.dev_bytes = 0; }
    const size_t want = std::max(bytes, l.dev_bytes * 2);
    hipError_t e = hipMalloc((void **)&l.dev, want);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("device allocation of %zu bytes failed: %s", want, hipGetErrorString(e));
        return IPX_ERR_NOMEM;
    }
    l.dev_bytes = want;
    return IPX_OK;
}

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

bool frame_args_ok(const void *p, int w, int h, int stride)
{
    return p && w >= 0 && h >= 0 && (long long)stride >= (long long)w * 4;
}

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
    if (op == IPX_OP_OVER) IPX_HIP(launch_opaque_scan(src.pix, src.w, src.h, src.stride, flag, s));  // RGBA and NRGBA: alpha scan
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

int ipx_device_count(void)
{
    clear_error();
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); set_error("hipGetDeviceCount: %s", hipGetErrorString(e)); return IPX_ERR_NODEVICE; }
    return n;
}

int ipx_create(const ipx_config *cfg, ipx_ctx **out)
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
    int lanes = cfg && cfg->lanes > 0 ? cfg->lanes : env_int("IPX_LANES", 3);
    c->lane_bytes = cfg && cfg->lane_bytes ? cfg->lane_bytes : (size_t)64 << 20;
    c->lanes.resize(lanes);
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    for (auto &l : c->lanes) {
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMalloc((void **)&l.flag, sizeof(int));
    }
    if (e != hipSuccess) {
        set_error("context setup failed: %s", hipGetErrorString(e));
        ipx_destroy(c);
        return IPX_ERR_HIP;
    }
    *out = c;
    return IPX_OK;
}

void ipx_destroy(ipx_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    for (auto &l : c->lanes) {
        if (l.stream) { (void)hipStreamSynchronize(l.stream); (void)hipStreamDestroy(l.stream); }
        if (l.dev) (void)hipFree(l.dev);
        if (l.flag) (void)hipFree(l.flag);
    }
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
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
            return p;
        }
    }
    (void)hipSetDevice(ctx->device);
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e != hipSuccess) { (void)hipGetLastError(); set_error("hipHostMalloc(%zu): %s", want, hipGetErrorString(e)); return nullptr; }
    std::lock_guard<std::mutex> lk(ctx->host_mu);
    ctx->host_size[p] = want;
    return p;
}

int ipx_host_free(ipx_ctx *ctx, void *p)
{
    IPX_ENTER(ctx);
    if (!p) return IPX_OK;
    {
        std::lock_guard<std::mutex> lk(ctx->host_mu);
        auto it = ctx->host_size.find(p);
        if (it != ctx->host_size.end()) {
            const size_t sz = it->second;
            ctx->host_size.erase(it);
            if (ctx->host_cached + sz <= ctx->host_cache_limit) {
                ctx->host_free_blocks.emplace(sz, p);
                ctx->host_cached += sz;
                return IPX_OK;
            }
        }
    }
    IPX_HIP(hipHostFree(p));
    return IPX_OK;
}

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

int ipx_dev_free(ipx_ctx *ctx, void *p)
{
    IPX_ENTER(ctx);
    if (p) IPX_HIP(hipFree(p));
    return IPX_OK;
}

int ipx_memcpy_h2d(ipx_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes)
{
    IPX_ENTER(ctx);
    if (bytes) IPX_HIP(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
    return IPX_OK;
}

int ipx_memcpy_d2h(ipx_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes)
{
    IPX_ENTER(ctx);
    if (bytes) IPX_HIP(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
    return IPX_OK;
}

int ipx_memcpy_d2d(ipx_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes)
{
    IPX_ENTER(ctx);
    if (bytes) IPX_HIP(hipMemcpy(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice));
    return IPX_OK;
}

int ipx_device_sync(ipx_ctx *ctx)
{
    IPX_ENTER(ctx);
    IPX_HIP(hipDeviceSynchronize());
    return IPX_OK;
}

int ipx_stream_sync(ipx_ctx *ctx, void *stream)
{
    IPX_ENTER(ctx);
    IPX_HIP(hipStreamSynchronize(stream ? (hipStream_t)stream : ctx->stream));
    return IPX_OK;
}

void *ipx_event_create(ipx_ctx *ctx)
{
    clear_error();
    if (!ctx) return nullptr;
    (void)hipSetDevice(ctx->device);
    hipEvent_t ev = nullptr;
    if (hipEventCreate(&ev) != hipSuccess) { (void)hipGetLastError(); set_error("hipEventCreate failed"); return nullptr; }
    return ev;
}

int ipx_event_record(ipx_ctx *ctx, void *event, void *stream)
{
    IPX_ENTER(ctx);
    if (!event) { set_error("ipx_event_record: null event"); return IPX_ERR_INVALID; }
    IPX_HIP(hipEventRecord((hipEvent_t)event, stream ? (hipStream_t)stream : ctx->stream));
    return IPX_OK;
}

int ipx_event_elapsed_ms(ipx_ctx *ctx, void *start, void *stop, float *ms)
{
    IPX_ENTER(ctx);
    if (!start || !stop || !ms) { set_error("ipx_event_elapsed_ms: bad argument"); return IPX_ERR_INVALID; }
    IPX_HIP(hipEventSynchronize((hipEvent_t)stop));
    IPX_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return IPX_OK;
}

void ipx_event_destroy(ipx_ctx *ctx, void *event)
{
    if (!ctx || !event) return;
    (void)hipSetDevice(ctx->device);
    (void)hipEventDestroy((hipEvent_t)event);
}

// ---- device-pointer operations ---------------------------------------------------------------------
int ipx_dev_scale_bilinear_rgba8(ipx_ctx *ctx, void *stream, uint8_t *dst, int dw, int dh,
                                 int dstride, ipx_rect dr, const uint8_t *src, int sw, int sh,
                                 int sstride, ipx_rect sr, int op)
{
    IPX_ENTER(ctx);
    if (!frame_args_ok(dst, dw, dh, dstride) || !frame_args_ok(src, sw, sh, sstride)) {
        set_error("ipx_dev_scale_bilinear_rgba8: bad frame arguments");
        return IPX_ERR_INVALID;
    }
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

int ipx_dev_draw_rgba8(ipx_ctx *ctx, void *stream, uint8_t *dst, int dw, int dh, int dstride,
                       ipx_rect r, const uint8_t *src, int sw, int sh, int sstride, int spx, int spy,
                       int op)
{
    IPX_ENTER(ctx);
    if (!frame_args_ok(dst, dw, dh, dstride) || !frame_args_ok(src, sw, sh, sstride)) {
        set_error("ipx_dev_draw_rgba8: bad frame arguments");
        return IPX_ERR_INVALID;
    }
    return dev_draw(stream ? (hipStream_t)stream : ctx->stream, dst, dw, dh, dstride, to_rect(r), src,
                    sw, sh, sstride, spx, spy, op);
}

int ipx_glyphset_create(ipx_ctx *ctx, const ipx_glyph *glyphs, int n, const uint8_t col[4],
                        ipx_glyphset **out)
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

void ipx_glyphset_destroy(ipx_ctx *ctx, ipx_glyphset *gs)
{
    if (!gs) return;
    (void)hipSetDevice(ctx ? ctx->device : gs->device);
    for (auto &kv : gs->clipped) if (kv.second.dev) (void)hipFree(kv.second.dev);
    if (gs->masks_dev) (void)hipFree(gs->masks_dev);
    delete gs;
}

int ipx_dev_composite_glyphs_rgba8(ipx_ctx *ctx, void *stream, uint8_t *dst, int dw, int dh,
                                   int dstride, const ipx_glyphset *gs)
{
    IPX_ENTER(ctx);
    if (!frame_args_ok(dst, dw, dh, dstride) || !gs) { set_error("ipx_dev_composite_glyphs_rgba8: bad argument"); return IPX_ERR_INVALID; }
    return dev_composite(stream ? (hipStream_t)stream : ctx->stream, dst, dw, dh, dstride, 0, 1, gs);
}

// ---- host-pointer operations: stage through a lane ------------------------------------------------
int ipx_scale_bilinear_rgba8(ipx_ctx *ctx, uint8_t *dst, int dw, int dh, int dstride, ipx_rect dr,
                             const uint8_t *src, int sw, int sh, int sstride, ipx_rect sr, int op)
{
    IPX_ENTER(ctx);
    if (!frame_args_ok(dst, dw, dh, dstride) || !frame_args_ok(src, sw, sh, sstride)) {
        set_error("ipx_scale_bilinear_rgba8: bad frame arguments");
        return IPX_ERR_INVALID;
    }
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

int ipx_draw_rgba8(ipx_ctx *ctx, uint8_t *dst, int dw, int dh, int dstride, ipx_rect r,
                   const uint8_t *src, int sw, int sh, int sstride, int spx, int spy, int op)
{
    IPX_ENTER(ctx);
    if (!frame_args_ok(dst, dw, dh, dstride) || !frame_args_ok(src, sw, sh, sstride)) {
        set_error("ipx_draw_rgba8: bad frame arguments");
        return IPX_ERR_INVALID;
    }
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

int ipx_composite_glyphs_rgba8(ipx_ctx *ctx, uint8_t *dst, int dw, int dh, int dstride,
                               const ipx_glyph *glyphs, int n, const uint8_t col[4])
{
    IPX_ENTER(ctx);
    if (!frame_args_ok(dst, dw, dh, dstride) || n < 0 || (n && !glyphs) || !col) {
        set_error("ipx_composite_glyphs_rgba8: bad argument");
        return IPX_ERR_INVALID;
    }
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
                              const uint8_t *src, int sw, int sh, int sstride, ipx_rect sr, int op)
{
    IPX_ENTER(ctx);
    if (!frame_args_ok(dst, dw, dh, dstride) || !frame_args_ok(src, sw, sh, sstride)) {
        set_error("ipx_scale_bilinear_nrgba8: bad frame arguments");
        return IPX_ERR_INVALID;
    }
    if (!dw || !dh || !sw || !sh) return IPX_OK;
    return stage_and_run(ctx, dst, dw, dh, dstride, align256((size_t)sw * sh * 4), [&](hipStream_t s, int *flag, uint8_t *ddst, uint8_t *dsrc) -> int {
        IPX_HIP(hipMemcpy2DAsync(dsrc, (size_t)sw * 4, src, sstride, (size_t)sw * 4, sh, hipMemcpyHostToDevice, s));
        return dev_scale_src(s, flag, ddst, dw, dh, dw * 4, to_rect(dr), rgba_src(dsrc, sw, sh, sw * 4, IPX_SRC_NRGBA), to_rect(sr), op);
    });
}

int ipx_draw_nrgba8(ipx_ctx *ctx, uint8_t *dst, int dw, int dh, int dstride, ipx_rect r, const uint8_t *src,
                    int sw, int sh, int sstride, int spx, int spy, int op)
{
    IPX_ENTER(ctx);
    if (!frame_args_ok(dst, dw, dh, dstride) || !frame_args_ok(src, sw, sh, sstride)) {
        set_error("ipx_draw_nrgba8: bad frame arguments");
        return IPX_ERR_INVALID;
    }
    if (!dw || !dh || !sw || !sh) return IPX_OK;
    return stage_and_run(ctx, dst, dw, dh, dstride, align256((size_t)sw * sh * 4), [&](hipStream_t s, int *, uint8_t *ddst, uint8_t *dsrc) -> int {
        IPX_HIP(hipMemcpy2DAsync(dsrc, (size_t)sw * 4, src, sstride, (size_t)sw * 4, sh, hipMemcpyHostToDevice, s));
        return dev_draw_src(s, ddst, dw, dh, dw * 4, to_rect(r), rgba_src(dsrc, sw, sh, sw * 4, IPX_SRC_NRGBA), spx, spy, op);
    });
}

int ipx_scale_bilinear_ycbcr(ipx_ctx *ctx, uint8_t *dst, int dw, int dh, int dstride, ipx_rect dr,
                             const ipx_ycbcr *src, ipx_rect sr)
{
    IPX_ENTER(ctx);
    int cw = 0, ch = 0;
    if (!frame_args_ok(dst, dw, dh, dstride) || !ycbcr_ok(src, &cw, &ch)) {
        set_error("ipx_scale_bilinear_ycbcr: bad arguments");
        return IPX_ERR_INVALID;
    }
    if (!dw || !dh) return IPX_OK;
    const size_t bytes = align256((size_t)src->w * src->h) + 2 * align256((size_t)cw * ch);
    return stage_and_run(ctx, dst, dw, dh, dstride, bytes, [&](hipStream_t s, int *flag, uint8_t *ddst, uint8_t *dsrc) -> int {
        DevSrc d;
        int rc = upload_ycbcr(s, src, cw, ch, dsrc, &d);
        if (rc) return rc;
        return dev_scale_src(s, flag, ddst, dw, dh, dw * 4, to_rect(dr), d, to_rect(sr), IPX_OP_SRC);
    });
}

int ipx_draw_ycbcr(ipx_ctx *ctx, uint8_t *dst, int dw, int dh, int dstride, ipx_rect r, const ipx_ycbcr *src,
                   int spx, int spy)
{
    IPX_ENTER(ctx);
    int cw = 0, ch = 0;
    if (!frame_args_ok(dst, dw, dh, dstride) || !ycbcr_ok(src, &cw, &ch)) {
        set_error("ipx_draw_ycbcr: bad arguments");
        return IPX_ERR_INVALID;
    }
    if (!dw || !dh) return IPX_OK;
    const size_t bytes = align256((size_t)src->w * src->h) + 2 * align256((size_t)cw * ch);
    return stage_and_run(ctx, dst, dw, dh, dstride, bytes, [&](hipStream_t s, int *, uint8_t *ddst, uint8_t *dsrc) -> int {
        DevSrc d;
        int rc = upload_ycbcr(s, src, cw, ch, dsrc, &d);
        if (rc) return rc;
        return dev_draw_src(s, ddst, dw, dh, dw * 4, to_rect(r), d, spx, spy, IPX_OP_SRC);
    });
}

}  // extern "C"

extern "C" {

// ---- plans -------------------------------------------------------------------------------------------
int ipx_plan_create(ipx_ctx *ctx, const ipx_plan_params *p, ipx_plan **out)
{
    IPX_ENTER(ctx);
    if (!p || !out) { set_error("ipx_plan_create: bad argument"); return IPX_ERR_INVALID; }
    *out = nullptr;
    if (p->sw <= 0 || p->sh <= 0) { set_error("ipx_plan_create: frame size %dx%d", p->sw, p->sh); return IPX_ERR_INVALID; }
    ipx_plan *pl = new (std::nothrow) ipx_plan;
    if (!pl) { set_error("out of memory"); return IPX_ERR_NOMEM; }
    pl->p = *p;
    int rc = IPX_OK;
    const int sw = p->sw, sh = p->sh;
    if (p->do_resize) {
        int nw, nh;
        rc = ipx_resize_dims(sw, sh, p->resize_w, p->resize_h, p->keep_aspect, &nw, &nh);
        if (rc) { delete pl; return rc; }
        pl->sc[0].on = true; pl->sc[0].dw = nw; pl->sc[0].dh = nh; pl->sc[0].sr = Rect{0, 0, sw, sh};
        pl->info.resize_w = nw; pl->info.resize_h = nh;
        pl->info.resize_bytes = (size_t)nw * nh * 4;
    }
    if (p->do_thumbnail) {
        int nw, nh;
        ipx_rect crop;
        rc = ipx_thumb_geometry(sw, sh, p->thumb_size, p->crop_to_fit, &crop, &nw, &nh);
        if (rc) { delete pl; return rc; }
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
            const int kx = axis_dyadic_bits(xt[k].data(), s.dw, 12);
            const int ky = axis_dyadic_bits(yt[k].data(), s.dh, 12);
            if (kx >= 0 && ky >= 0 && kx + ky <= 16) { s.dyadic_shift = kx + ky; s.kx = kx; s.ky = ky; }  // 8 + kx + ky <= 24 bits
        }
    }

    // block shape: owned columns per workgroup (multiple of 4 pixels = 16 B) and owned rows; a
    // column block may hold at most 256 * kBandNX destination columns of any scaled output
    int max_cols = std::max(4, env_int("IPX_BLK_COLS", 2044)) & ~3;
    const size_t lds_budget = ((size_t)std::max(8, env_int("IPX_LDS_KB", 72)) << 10) - band_lds_bytes(-1, -4);
    int bc = 0, br = 0;
    std::vector<int> rb[2], cbv[2];
    for (;;) {
        const int ncb = (sw + max_cols - 1) / max_cols;
        bc = std::max(4, ((sw + ncb - 1) / ncb + 3) & ~3);
        br = (int)(lds_budget / ((size_t)(bc + 4) * 4)) - 1;
        br = std::max(1, std::min(br, env_int("IPX_BAND_ROWS_MAX", bc / 4 + 1 > 256 ? 8 : 16)));  // shapes of band_pipe_shape
        if (env_int("IPX_BAND_ROWS", 0) > 0) br = env_int("IPX_BAND_ROWS", 0);
        br = std::min(br, sh);
        pl->blk_cols = bc; pl->band_rows = br;
        pl->ncolblk = (sw + bc - 1) / bc;
        pl->nbands = (sh + br - 1) / br;
        int widest = 0;
        pl->most_rows = 0;
        pl->nx_out[0] = pl->nx_out[1] = 0;
        for (int k = 0; k < 2; k++) {
            PlanScale &s = pl->sc[k];
            if (xt[k].empty()) continue;
            rb[k].assign(pl->nbands + 1, 0); cbv[k].assign(pl->ncolblk + 1, 0);
            int d = 0;
            for (int b = 0; b <= pl->nbands; b++) {  // first output row whose tap pair starts in band b or below
                while (d < s.dh && s.sr.y0 + yt[k][d].base < b * br) d++;
                rb[k][b] = b == pl->nbands ? s.dh : d;
            }
            d = 0;
            for (int c = 0; c <= pl->ncolblk; c++) {
                while (d < s.dw && s.sr.x0 + xt[k][d].base < c * bc) d++;
                cbv[k][c] = c == pl->ncolblk ? s.dw : d;
            }
            int wk = 0;
            for (int c = 0; c < pl->ncolblk; c++) wk = std::max(wk, cbv[k][c + 1] - cbv[k][c]);
            widest = std::max(widest, wk);
            pl->nx_out[k] = (wk + 255) / 256;
            for (int b = 0; b < pl->nbands; b++) pl->most_rows = std::max(pl->most_rows, rb[k][b + 1] - rb[k][b]);
        }
        if (widest <= 256 * kBandNX) break;
        if (bc <= 4) { pl->fused = false; *out = pl; return IPX_OK; }  // enormous upscale: per-operation kernels
        max_cols = std::max(4, (int)((long long)bc * 256 * kBandNX / widest) & ~3);
        if (max_cols >= bc) max_cols = bc - 4;
    }

    // host tables -> one device blob
    std::vector<uint8_t> blob;
    auto put = [&](const void *src, size_t bytes) {
        const size_t off = (blob.size() + 15) & ~(size_t)15;
        blob.resize(off + bytes);
        memcpy(blob.data() + off, src, bytes);
        return off;
    };
    size_t off_xt[2] = {0, 0}, off_yt[2] = {0, 0}, off_rb[2] = {0, 0}, off_cb[2] = {0, 0};
    for (int k = 0; k < 2; k++) {
        if (xt[k].empty()) continue;
        off_xt[k] = put(xt[k].data(), xt[k].size() * sizeof(AxisTap));
        off_yt[k] = put(yt[k].data(), yt[k].size() * sizeof(AxisTap));
        off_rb[k] = put(rb[k].data(), rb[k].size() * sizeof(int));
        off_cb[k] = put(cbv[k].data(), cbv[k].size() * sizeof(int));
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
            PlanScale &s = pl->sc[k];
            if (!s.on || s.dw <= 0 || s.dh <= 0) continue;
            s.xt = (AxisTap *)(pl->blob + off_xt[k]);
            s.yt = (AxisTap *)(pl->blob + off_yt[k]);
            s.row_begin = (int *)(pl->blob + off_rb[k]);
            s.col_begin = (int *)(pl->blob + off_cb[k]);
        }
    }
    *out = pl;
    return IPX_OK;
}

void ipx_plan_destroy(ipx_ctx *ctx, ipx_plan *plan)
{
    if (!plan) return;
    if (ctx) (void)hipSetDevice(ctx->device);
    if (plan->blob) (void)hipFree(plan->blob);
    if (plan->thumb_only) ipx_plan_destroy(ctx, plan->thumb_only);
    delete plan;
}

int ipx_plan_query(const ipx_plan *plan, ipx_plan_info *info)
{
    clear_error();
    if (!plan || !info) { set_error("ipx_plan_query: bad argument"); return IPX_ERR_INVALID; }
    *info = plan->info;
    return IPX_OK;
}

int ipx_plan_run_dev(ipx_ctx *ctx, void *stream, const ipx_plan *pl, int n, const uint8_t *src,
                     int sstride, size_t src_frame_stride, uint8_t *resize_out,
                     size_t resize_frame_stride, uint8_t *thumb_out, size_t thumb_frame_stride,
                     uint8_t *wm_out, size_t wm_frame_stride)
{
    IPX_ENTER(ctx);
    if (!pl || n < 0 || !src || (long long)sstride < (long long)pl->p.sw * 4) {
        set_error("ipx_plan_run_dev: bad argument");
        return IPX_ERR_INVALID;
    }
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
        a.band_rows = pl->band_rows; a.nbands = pl->nbands;
        a.blk_cols = pl->blk_cols; a.ncolblk = pl->ncolblk;
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
        if (aligned && env_int("IPX_PIPE", 1) && pl->most_rows <= 64 && band_pipe_shape(pl->band_rows, pl->blk_cols, &pr, &pc)) {
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
            o.xt = ps.xt; o.yt = ps.yt; o.row_begin = ps.row_begin; o.col_begin = ps.col_begin;
            o.dyadic_shift = ps.dyadic_shift;
            a.nx_out[a.nscale - 1] = pl->nx_out[k];
        }
        if (a.nscale == 1) a.sc[1] = a.sc[0];  // keeps the kernel's unconditional tap loads legal
        a.glyphs = pl->glyphs.dev; a.nglyphs = wm ? pl->glyphs.n : 0; a.gbox = pl->glyphs.bbox;
        a.cr = col[0]; a.cg = col[1]; a.cb = col[2]; a.ca = col[3];
        if (!wm && a.nscale == 0) return IPX_OK;
        // item order of the persistent kernel: the grid-interleaved sweep is steadier when outputs are scaled
        // (run-to-run 3.63-3.74 ms against 3.65-4.13 ms on one box, profiles/r01_item_order.txt); a plain
        // watermark copy is ~2 % faster with one contiguous run per workgroup
        if (a.pipe_order < 0) a.pipe_order = env_int("IPX_PIPE_ORDER", a.nscale > 0 ? 1 : 0) ? 1 : 0;
        IPX_HIP(launch_band(a, s));
        if (a.stamps) {
            unsigned long long h[8];
            IPX_HIP(hipMemcpyAsync(h, a.stamps, sizeof h, hipMemcpyDeviceToHost, s));
            IPX_HIP(hipStreamSynchronize(s));
            const double items = (double)pl->nbands * pl->ncolblk * n, waves = 4.0;
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

int ipx_plan_run_host(ipx_ctx *ctx, const ipx_plan *pl, int n, const uint8_t *src, int sstride,
                      size_t src_frame_stride, uint8_t *resize_out, size_t resize_frame_stride,
                      uint8_t *thumb_out, size_t thumb_frame_stride, uint8_t *wm_out,
                      size_t wm_frame_stride)
{
    IPX_ENTER(ctx);
    if (!pl || n < 0 || !src || (long long)sstride < (long long)pl->p.sw * 4) {
        set_error("ipx_plan_run_host: bad argument");
        return IPX_ERR_INVALID;
    }
    if (n == 0) return IPX_OK;
    const int sw = pl->p.sw, sh = pl->p.sh;
    // device-side frame strides: tight when that keeps rows 16-byte aligned (then a whole chunk moves
    // with one copy per direction and buffer), padded to 256 otherwise
    auto dstride = [](size_t bytes) { return (bytes & 15) == 0 ? bytes : align256(bytes); };
    const size_t fsrc = dstride((size_t)sw * sh * 4);
    const size_t fres = resize_out ? dstride(pl->info.resize_bytes) : 0;
    const size_t fth = thumb_out ? dstride(pl->info.thumb_bytes) : 0;
    const size_t fwm = wm_out ? dstride(pl->info.wm_bytes) : 0;
    const size_t per_frame = fsrc + fres + fth + fwm;
    // chunk the batch so that H2D of one chunk, the kernel of another and D2H of a third overlap on
    // different lanes (one stream each); several chunks per lane keep all three engines busy
    const int nl = (int)ctx->lanes.size();
    int chunk = std::max(1, (n + 4 * nl - 1) / (4 * nl));
    chunk = (int)std::min<size_t>((size_t)chunk, std::max<size_t>(1, ctx->lane_bytes / per_frame));
    chunk = std::max(1, std::min(chunk, env_int("IPX_HOST_CHUNK", 16)));

    // take every lane: this call is the pipeline
    std::vector<Lane *> lanes;
    {
        std::unique_lock<std::mutex> lk(ctx->mu);
        ctx->cv.wait(lk, [&] { for (auto &l : ctx->lanes) if (l.busy) return false; return true; });
        for (auto &l : ctx->lanes) { l.busy = true; lanes.push_back(&l); }
    }
    int rc = IPX_OK;
    hipError_t e = hipSuccess;
    for (auto *l : lanes) {
        rc = lane_reserve(*l, per_frame * chunk + 256);
        if (rc) break;
    }
    const bool src_tight = sstride == sw * 4 && src_frame_stride == fsrc;
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
        // the lane's stream serialises reuse of its scratch: chunk c waits for chunk c - lanes
        if (src_tight) {
            e = hipMemcpyAsync(dsrc, src + (size_t)i0 * src_frame_stride, fsrc * m, hipMemcpyHostToDevice, l.stream);
        } else {
            for (int i = 0; i < m && e == hipSuccess; i++)
                e = hipMemcpy2DAsync(dsrc + fsrc * i, (size_t)sw * 4, src + (size_t)(i0 + i) * src_frame_stride, sstride,
                                     (size_t)sw * 4, sh, hipMemcpyHostToDevice, l.stream);
        }
        if (e != hipSuccess) break;
        rc = ipx_plan_run_dev(ctx, l.stream, pl, m, dsrc, sw * 4, fsrc, dres, fres, dth, fth, dwm, fwm);
        if (rc) break;
        e = d2h(resize_out, resize_frame_stride, dres, fres, pl->info.resize_bytes, i0, m, l.stream);
        if (e == hipSuccess) e = d2h(thumb_out, thumb_frame_stride, dth, fth, pl->info.thumb_bytes, i0, m, l.stream);
        if (e == hipSuccess) e = d2h(wm_out, wm_frame_stride, dwm, fwm, pl->info.wm_bytes, i0, m, l.stream);
    }
    for (auto *l : lanes) {
        hipError_t e2 = hipStreamSynchronize(l->stream);
        if (e == hipSuccess) e = e2;
    }
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        for (auto *l : lanes) l->busy = false;
    }
    ctx->cv.notify_all();
    if (!rc && e != hipSuccess) { set_error("ipx_plan_run_host: %s", hipGetErrorString(e)); rc = IPX_ERR_HIP; }
    return rc;
}


// ---- decoded JPEG batches ---------------------------------------------------------------------------
// The reference treats a *image.YCbCr source differently per operator:
//   resize, non-crop thumbnail: scale_RGBA_YCbCr4xx_Src -- every TAP converted to 16 bit, then the lerp;
//   crop thumbnail: the crop copy converts to RGBA8 first (imageutil.DrawYCbCr), the scale then reads RGBA8;
//   watermark: draw.Draw converts to RGBA8 (DrawYCbCr), the glyphs go over that.
// So: one conversion pass into the watermark frame (or scratch), the RGBA band kernel for the crop
// thumbnail on the converted frames, the glyph composite in place, and a batched YCbCr scale.
int ipx_plan_run_dev_ycbcr(ipx_ctx *ctx, void *stream, const ipx_plan *pl, int n, const ipx_ycbcr_batch *src,
                           uint8_t *resize_out, size_t resize_frame_stride, uint8_t *thumb_out,
                           size_t thumb_frame_stride, uint8_t *wm_out, size_t wm_frame_stride)
{
    IPX_ENTER(ctx);
    if (!pl || n < 0 || !src || !src->y || !src->cb || !src->cr || src->ratio < 0 || src->ratio > IPX_YCBCR_440 ||
        src->ystride < pl->p.sw) {
        set_error("ipx_plan_run_dev_ycbcr: bad argument");
        return IPX_ERR_INVALID;
    }
    if (n == 0) return IPX_OK;
    if (n > 65535) { set_error("ipx_plan_run_dev_ycbcr: at most 65535 frames per call"); return IPX_ERR_UNSUPPORTED; }
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    const int sw = pl->p.sw, sh = pl->p.sh;
    uint8_t *res = pl->sc[0].on ? resize_out : nullptr;
    uint8_t *th = pl->sc[1].on ? thumb_out : nullptr;
    uint8_t *wm = pl->p.do_watermark ? wm_out : nullptr;
    const bool crop_thumb = th && pl->p.crop_to_fit;

    // one fused pass over the planes when the tile shape and alignments allow it (ipx_band_ycc.hip)
    if (pl->fused && env_int("IPX_YCC_FUSED", 1) && pl->band_rows <= 8 && pl->most_rows <= 64) {
        YccArgs A{};
        BandArgs &a = A.b;
        a.sw = sw; a.sh = sh;
        a.band_rows = pl->band_rows; a.nbands = pl->nbands;
        a.blk_cols = pl->blk_cols; a.ncolblk = pl->ncolblk;
        a.nframes = n;
        a.pipe_wgs = std::max(1, env_int("IPX_PIPE_WGS", 8));
        a.pipe_nt = 512; a.pipe_order = 1;
        a.cus = ctx->cus;
        a.wm = wm; a.wm_frame_stride = wm_frame_stride; a.wm_stride = sw * 4;
        uint8_t *outs[2] = {res, th};
        const size_t ostr[2] = {resize_frame_stride, thumb_frame_stride};
        for (int k = 0; k < 2; k++) {
            const PlanScale &ps = pl->sc[k];
            if (!outs[k] || ps.dw <= 0 || ps.dh <= 0) continue;
            A.mode[a.nscale] = k == 1 && pl->p.crop_to_fit ? 1 : 0;
            ScaleOut &o = a.sc[a.nscale++];
            o.out = outs[k]; o.frame_stride = ostr[k]; o.ostride = ps.dw * 4;
            o.dw = ps.dw; o.dh = ps.dh; o.sr_x0 = ps.sr.x0; o.sr_y0 = ps.sr.y0;
            o.xt = ps.xt; o.yt = ps.yt; o.row_begin = ps.row_begin; o.col_begin = ps.col_begin;
            o.dyadic_shift = ps.dyadic_shift;
            // 16-bit taps in u32: x0*tap + x1*tap < 2^24 and the weights below 2^9 keep every product in 24 x 24 bits
            if (A.mode[a.nscale - 1] == 0 && (ps.kx > 8 || ps.ky > 8)) o.dyadic_shift = -1;
            a.nx_out[a.nscale - 1] = pl->nx_out[k];
        }
        if (a.nscale == 1) { a.sc[1] = a.sc[0]; A.mode[1] = A.mode[0]; }
        const uint8_t *c = pl->p.glyphs ? pl->p.glyphs->col : nullptr;
        a.glyphs = pl->glyphs.dev; a.nglyphs = wm ? pl->glyphs.n : 0; a.gbox = pl->glyphs.bbox;
        a.cr = c ? c[0] * 0x101u : 0; a.cg = c ? c[1] * 0x101u : 0; a.cb = c ? c[2] * 0x101u : 0; a.ca = c ? c[3] * 0x101u : 0;
        if (!wm && a.nscale == 0) return IPX_OK;
        A.y = src->y; A.cb = src->cb; A.cr = src->cr; A.ystride = src->ystride; A.cstride = src->cstride;
        A.y_fs = src->y_frame_stride; A.c_fs = src->c_frame_stride; A.ratio = src->ratio;
        A.cw = (src->ratio == IPX_YCBCR_422 || src->ratio == IPX_YCBCR_420) ? (sw + 1) / 2 : sw;
        A.ch = (src->ratio == IPX_YCBCR_420 || src->ratio == IPX_YCBCR_440) ? (sh + 1) / 2 : sh;
        bool matched = false;
        if (src->cstride >= A.cw) IPX_HIP(launch_band_ycc(A, s, &matched));
        if (matched) return IPX_OK;
    }

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
                           size_t wm_frame_stride)
{
    IPX_ENTER(ctx);
    if (!pl || n < 0 || !src || (long long)sstride < (long long)pl->p.sw * 4) { set_error("ipx_plan_run_dev_nrgba: bad argument"); return IPX_ERR_INVALID; }
    if (n == 0) return IPX_OK;
    if (n > 65535) { set_error("ipx_plan_run_dev_nrgba: at most 65535 frames per call"); return IPX_ERR_UNSUPPORTED; }
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    const int sw = pl->p.sw, sh = pl->p.sh;
    uint8_t *res = pl->sc[0].on ? resize_out : nullptr;
    uint8_t *th = pl->sc[1].on ? thumb_out : nullptr;
    uint8_t *wm = pl->p.do_watermark ? wm_out : nullptr;
    const bool crop_thumb = th && pl->p.crop_to_fit;
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

int ipx_plan_run_dev_gray(ipx_ctx *ctx, void *stream, const ipx_plan *pl, int n, const uint8_t *gray, int stride, size_t frame_stride,
                          uint8_t *resize_out, size_t resize_frame_stride, uint8_t *thumb_out, size_t thumb_frame_stride, uint8_t *wm_out,
                          size_t wm_frame_stride)
{
    IPX_ENTER(ctx);
    if (!pl || n < 0 || !gray || stride < pl->p.sw) { set_error("ipx_plan_run_dev_gray: bad argument"); return IPX_ERR_INVALID; }
    if (n == 0) return IPX_OK;
    if (n > 65535) { set_error("ipx_plan_run_dev_gray: at most 65535 frames per call"); return IPX_ERR_UNSUPPORTED; }
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    const int sw = pl->p.sw, sh = pl->p.sh;
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

int ipx_plan_run_host_ycbcr(ipx_ctx *ctx, const ipx_plan *pl, int n, const ipx_ycbcr_batch *src, uint8_t *resize_out,
                            size_t resize_frame_stride, uint8_t *thumb_out, size_t thumb_frame_stride,
                            uint8_t *wm_out, size_t wm_frame_stride)
{
    IPX_ENTER(ctx);
    if (!pl || n < 0 || !src || !src->y || !src->cb || !src->cr || src->ratio < 0 || src->ratio > IPX_YCBCR_440) {
        set_error("ipx_plan_run_host_ycbcr: bad argument");
        return IPX_ERR_INVALID;
    }
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

}  // extern "C"


// ---- jpeg.Encode: the entries that touch the device (tables / entropy coder: ipx_jpeg_host.cpp) ----------
#include <atomic>
#include <functional>
#include <memory>
#include <string>
#include <thread>

extern "C" {

int ipx_dev_jpeg_fdct_rgba8(ipx_ctx *ctx, void *stream, const uint8_t *src, int w, int h, int stride, size_t frame_stride,
                            int n, int quality, int16_t *coefs)
{
    IPX_ENTER(ctx);
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
    memcpy(a.recip, t.recip, sizeof a.recip);
    memcpy(a.div8, t.div8, sizeof a.div8);
    IPX_HIP(launch_jpeg_fdct(a, n, stream ? (hipStream_t)stream : ctx->stream));
    return IPX_OK;
}

}  // extern "C"

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
    const int nt = std::max(1, std::min((int)std::thread::hardware_concurrency(), n));
    std::vector<std::vector<uint8_t>> streams(n);
    std::atomic<int> next{0};
    auto work = [&] {
        for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1))
            jpeg_write_stream(host + (per / sizeof(int16_t)) * (size_t)i, w, h, t, &streams[i]);
    };
    std::vector<std::thread> pool;
    for (int i = 1; i < nt; i++) pool.emplace_back(work);
    work();
    for (auto &th : pool) th.join();
    (void)hipHostFree(host);
    size_t total = 0;
    for (int i = 0; i < n; i++) { offs[i] = total; lens[i] = streams[i].size(); total += (lens[i] + 15) & ~(size_t)15; }
    uint8_t *b = (uint8_t *)ipx_host_alloc(ctx, total ? total : 1);
    if (!b) return IPX_ERR_NOMEM;
    for (int i = 0; i < n; i++) memcpy(b + offs[i], streams[i].data(), lens[i]);
    *blob = b;
    return IPX_OK;
}

namespace {
struct AsyncFree {   // stream-ordered scratch of one call
    hipStream_t s;
    std::vector<void *> p;
    ~AsyncFree() { for (void *q : p) (void)hipFreeAsync(q, s); }
    template <class T> hipError_t get(T **out, size_t bytes)
    {
        void *q = nullptr;
        hipError_t e = hipMallocAsync(&q, bytes ? bytes : 1, s);
        if (e == hipSuccess) p.push_back(q);
        *out = (T *)q;
        return e;
    }
};
}  // namespace

extern "C" {

}  // extern "C"

// n frames in HBM -> streams in one pinned block, everything on stream s; dcoefs = n * ipx_jpeg_coef_count int16 of scratch
static int jpeg_encode_core(ipx_ctx *ctx, hipStream_t s, int16_t *dcoefs, const uint8_t *src, int w, int h, int stride, size_t frame_stride,
                            int n, int quality, uint8_t **blob, size_t *offs, size_t *lens);

extern "C" {

int ipx_jpeg_encode_batch_dev(ipx_ctx *ctx, const uint8_t *src, int w, int h, int stride, size_t frame_stride, int n,
                              int quality, uint8_t **blob, size_t *offs, size_t *lens)
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

}  // extern "C"

static int jpeg_encode_core(ipx_ctx *ctx, hipStream_t s, int16_t *dcoefs, const uint8_t *src, int w, int h, int stride, size_t frame_stride,
                            int n, int quality, uint8_t **blob, size_t *offs, size_t *lens)
{
    const size_t per = ipx_jpeg_coef_count(w, h) * sizeof(int16_t);
    int rc = ipx_dev_jpeg_fdct_rgba8(ctx, s, src, w, h, stride, frame_stride, n, quality, dcoefs);
    if (rc) return rc;

    // ---- entropy coding on the GPU: size, scan, place, stuff (ipx_jpeg_entropy.hip); two small read-backs ----
    const int nblk = (int)(per / 128);
    JpegTables t;
    jpeg_tables(quality, &t);
    std::vector<uint8_t> hdr;
    jpeg_write_header(w, h, t, &hdr);
    uint32_t packed[1024];
    jpeg_huff_packed(packed);
    AsyncFree mem{s, {}};
    uint32_t *d_tab, *d_len, *d_tot, *d_ubytes, *d_ff, *d_fftot;
    unsigned long long *d_ubase, *d_obase;
    uint8_t *d_hdr, *d_ustream = nullptr, *d_ostream = nullptr;
    IPX_HIP(mem.get(&d_tab, sizeof packed));
    IPX_HIP(mem.get(&d_len, (size_t)n * nblk * 4));
    IPX_HIP(mem.get(&d_tot, (size_t)n * 4));
    IPX_HIP(mem.get(&d_ubytes, (size_t)n * 4));
    IPX_HIP(mem.get(&d_fftot, (size_t)n * 4));
    IPX_HIP(mem.get(&d_ubase, (size_t)n * 8));
    IPX_HIP(mem.get(&d_obase, (size_t)n * 8));
    IPX_HIP(mem.get(&d_hdr, hdr.size()));
    IPX_HIP(hipMemcpyAsync(d_tab, packed, sizeof packed, hipMemcpyHostToDevice, s));
    IPX_HIP(hipMemcpyAsync(d_hdr, hdr.data(), hdr.size(), hipMemcpyHostToDevice, s));
    IPX_HIP(launch_jpeg_len(dcoefs, nblk, n, d_tab, d_len, s));
    IPX_HIP(launch_scan(d_len, nblk, n, d_tot, s));
    std::vector<uint32_t> tot(n), ubytes(n), ff(n);
    IPX_HIP(hipMemcpyAsync(tot.data(), d_tot, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    IPX_HIP(hipStreamSynchronize(s));
    std::vector<unsigned long long> ubase(n), obase(n);
    unsigned long long utotal = 0;
    uint32_t umax = 0;
    for (int i = 0; i < n; i++) {
        ubytes[i] = (tot[i] + 7) / 8;
        ubase[i] = utotal;
        utotal += align256((size_t)ubytes[i] + 8);
        umax = std::max(umax, ubytes[i]);
    }
    const int chunk = jpeg_chunk_bytes();
    const int max_chunks = (int)((umax + chunk - 1) / chunk);
    IPX_HIP(mem.get(&d_ustream, (size_t)utotal));
    IPX_HIP(mem.get(&d_ff, (size_t)n * max_chunks * 4));
    IPX_HIP(hipMemsetAsync(d_ustream, 0, (size_t)utotal, s));
    IPX_HIP(hipMemcpyAsync(d_ubase, ubase.data(), (size_t)n * 8, hipMemcpyHostToDevice, s));
    IPX_HIP(hipMemcpyAsync(d_ubytes, ubytes.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
    IPX_HIP(launch_jpeg_bits(dcoefs, nblk, n, d_tab, d_len, d_tot, d_ubase, d_ustream, s));
    IPX_HIP(launch_jpeg_ffcount(d_ustream, d_ubase, d_ubytes, max_chunks, n, d_ff, s));
    IPX_HIP(launch_scan(d_ff, max_chunks, n, d_fftot, s));
    IPX_HIP(hipMemcpyAsync(ff.data(), d_fftot, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    IPX_HIP(hipStreamSynchronize(s));
    unsigned long long ototal = 0;
    for (int i = 0; i < n; i++) {
        lens[i] = hdr.size() + ubytes[i] + ff[i] + 2;
        obase[i] = ototal;
        offs[i] = (size_t)ototal;
        ototal += (lens[i] + 15) & ~(size_t)15;
    }
    IPX_HIP(mem.get(&d_ostream, (size_t)ototal));
    IPX_HIP(hipMemcpyAsync(d_obase, obase.data(), (size_t)n * 8, hipMemcpyHostToDevice, s));
    IPX_HIP(launch_jpeg_stuff(d_ustream, d_ubase, d_ubytes, max_chunks, n, d_ff, d_hdr, (int)hdr.size(), d_obase, d_ostream, s));
    uint8_t *host = (uint8_t *)ipx_host_alloc(ctx, (size_t)ototal ? (size_t)ototal : 1);   // pinned: the download runs at link speed
    if (!host) return IPX_ERR_NOMEM;
    hipError_t e = hipMemcpyAsync(host, d_ostream, (size_t)ototal, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { (void)ipx_host_free(ctx, host); set_error("stream download failed: %s", hipGetErrorString(e)); return IPX_ERR_HIP; }
    *blob = host;
    return IPX_OK;
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
    const size_t ccoef = align256(std::max(cres, std::max(cth, cwm)));
    const size_t per_frame = fsrc + fres + fth + fwm + ccoef;
    // every lane runs its own host thread: upload, operators, three encodes (each with two small read-backs) --
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
        std::vector<size_t> offs(chunk), lens(chunk);
        for (int c = next.fetch_add(1); !rc && c < nchunks && status == IPX_OK; c = next.fetch_add(1)) {
            const int i0 = c * chunk, m = std::min(chunk, n - i0);
            uint8_t *dsrc = (uint8_t *)(((uintptr_t)l->dev + 255) & ~(uintptr_t)255);
            uint8_t *dres = fres ? dsrc + fsrc * chunk : nullptr;
            uint8_t *dth = fth ? dsrc + (fsrc + fres) * chunk : nullptr;
            uint8_t *dwm = fwm ? dsrc + (fsrc + fres + fth) * chunk : nullptr;
            int16_t *dcoef = (int16_t *)(dsrc + (fsrc + fres + fth + fwm) * chunk);
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
            for (int k = 0; k < 3 && !rc; k++) {
                const Out &o = outs[k];
                if (!o.dev || o.w <= 0 || o.h <= 0) continue;
                uint8_t *blob = nullptr;
                rc = jpeg_encode_core(ctx, l->stream, dcoef, o.dev, o.w, o.h, o.w * 4, o.fs, m, quality, &blob, offs.data(), lens.data());
                if (rc) break;
                for (int i = 0; i < m; i++) { o.dst[i0 + i].data = blob + offs[i]; o.dst[i0 + i].len = lens[i]; }
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
    std::vector<std::thread> pool;
    for (int i = 1; i < nl; i++) pool.emplace_back(worker, lanes[i]);
    worker(lanes[0]);
    for (auto &t : pool) t.join();
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
                           int quality, ipx_bytes *resize_out, ipx_bytes *thumb_out, ipx_bytes *wm_out, ipx_jpeg_result **result)
{
    IPX_ENTER(ctx);
    if (!pl || n < 0 || !src || !result || (long long)sstride < (long long)pl->p.sw * 4) {
        set_error("ipx_plan_run_host_jpeg: bad argument");
        return IPX_ERR_INVALID;
    }
    return run_host_jpeg_impl(ctx, pl, n, src, sstride, src_frame_stride, nullptr, quality, resize_out, thumb_out, wm_out, result);
}

int ipx_plan_run_host_ycbcr_jpeg(ipx_ctx *ctx, const ipx_plan *pl, int n, const ipx_ycbcr_batch *src, int quality,
                                 ipx_bytes *resize_out, ipx_bytes *thumb_out, ipx_bytes *wm_out, ipx_jpeg_result **result)
{
    IPX_ENTER(ctx);
    if (!pl || n < 0 || !result || !src || !src->y || !src->cb || !src->cr || src->ratio < 0 || src->ratio > IPX_YCBCR_440 ||
        src->ystride < pl->p.sw) {
        set_error("ipx_plan_run_host_ycbcr_jpeg: bad argument");
        return IPX_ERR_INVALID;
    }
    return run_host_jpeg_impl(ctx, pl, n, nullptr, 0, 0, src, quality, resize_out, thumb_out, wm_out, result);
}

int ipx_jpeg_encode_rgba8(ipx_ctx *ctx, const uint8_t *pix, int w, int h, int stride, int quality, uint8_t **out, size_t *len)
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

}  // extern "C"


// ---- image.Decode for JPEG batches -----------------------------------------------------------------------
struct ipx_jpeg_planes { std::vector<void *> dev; };

extern "C" {

void ipx_jpeg_planes_free(ipx_ctx *ctx, ipx_jpeg_planes *o)
{
    if (!o) return;
    if (ctx) (void)hipSetDevice(ctx->device);
    for (void *p : o->dev) (void)hipFree(p);
    delete o;
}

int ipx_jpeg_decode_batch(ipx_ctx *ctx, void *stream, const ipx_bytes *jpegs, int n, int *w, int *h, ipx_ycbcr_batch *planes,
                          int *status, ipx_jpeg_planes **owner)
{
    IPX_ENTER(ctx);
    if (!jpegs || n < 0 || !w || !h || !planes || !status || !owner) { set_error("ipx_jpeg_decode_batch: bad argument"); return IPX_ERR_INVALID; }
    *owner = nullptr;
    memset(planes, 0, sizeof *planes);
    if (n == 0) return IPX_OK;
    if (n > 65535) { set_error("ipx_jpeg_decode_batch: at most 65535 files per call"); return IPX_ERR_UNSUPPORTED; }
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    std::vector<JpegDecInfo> info(n);
    std::vector<JpegDecTables> tabs(n);
    std::vector<JpegDecImage> items;
    std::vector<uint8_t> valid(n, 0);
    std::vector<size_t> blob_off(n, 0);
    // host preparation runs on a few threads: parsing is trivial, but finding the RSTn markers and packing the scans walk
    // every compressed byte (0.3 GB for a thousand 1080p files)
    auto parallel_for = [&](int count, const std::function<void(int)> &fn) {
        const int nt = std::max(1, std::min({count / 8, (int)std::thread::hardware_concurrency(), 16}));
        std::atomic<int> next{0};
        auto work = [&] { for (int i = next.fetch_add(1); i < count; i = next.fetch_add(1)) fn(i); };
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; t++) pool.emplace_back(work);
        work();
        for (auto &t : pool) t.join();
    };
    // pieces of a scan: the whole scan, or one per restart interval.  Inside entropy-coded data 0xff is followed by 0x00 or by a
    // marker, so every 0xff 0xd0..0xd7 pair is an RSTn.
    std::vector<std::vector<uint32_t>> marks(n);
    parallel_for(n, [&](int i) {
        status[i] = !jpegs[i].data ? IPX_ERR_INVALID : (jpegs[i].len >= ((size_t)1 << 30) ? IPX_ERR_UNSUPPORTED : jpeg_parse(jpegs[i].data, jpegs[i].len, &info[i], &tabs[i]));
        if (status[i] != IPX_OK) return;
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
    int ref = -1;
    size_t blob_bytes = 0;
    std::vector<JpegParImage> par;
    const bool use_par = env_int("IPX_JPEG_PAR", 1) != 0;
    for (int i = 0; i < n; i++) {
        if (status[i] == IPX_OK) {
            if (ref < 0 && (*w <= 0 || (info[i].w == *w && info[i].h == *h))) ref = i;
            if (ref >= 0 && (info[i].w != info[ref].w || info[i].h != info[ref].h || info[i].h0 != info[ref].h0 || info[i].v0 != info[ref].v0 ||
                             info[i].ncomp != info[ref].ncomp))
                status[i] = IPX_ERR_UNSUPPORTED;
            else if (ref < 0) status[i] = IPX_ERR_UNSUPPORTED;   // a size other than the one asked for
        }
        if (status[i] != IPX_OK) continue;
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
            items.push_back(it);
        };
        size_t start = 0;
        int mcu = 0;
        for (uint32_t k : marks[i]) { push(start, k, mcu, I.ri); items.back().strict_end = 1; mcu += I.ri; start = (size_t)k + 2; }
        if (marks[i].empty() && use_par && I.scan_len >= (size_t)4 * jpeg_par_sub_bytes() && I.scan_len < ((size_t)1 << 28)) {
            // a long scan without restart markers: decoded in parallel inside the scan (ipx_jpeg_dec_par.hip)
            JpegParImage pi;
            memset(&pi, 0, sizeof pi);
            pi.scan_off = blob_bytes; pi.scan_len = (uint32_t)I.scan_len; pi.img = (uint32_t)i;
            pi.nsub = (uint32_t)((I.scan_len + jpeg_par_sub_bytes() - 1) / jpeg_par_sub_bytes());
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
    const JpegDecInfo &R = info[ref];
    *w = R.w; *h = R.h;
    JpegDecArgs a{};
    a.n = n; a.h0 = R.h0; a.v0 = R.v0;
    a.mxx = (R.w + 8 * R.h0 - 1) / (8 * R.h0); a.myy = (R.h + 8 * R.v0 - 1) / (8 * R.v0);
    const bool gray = R.ncomp == 1;                        // *image.Gray: one block per MCU, no chroma planes
    a.ybl = R.h0 * R.v0; a.bpm = gray ? 1 : a.ybl + 2;
    a.nblk = a.mxx * a.myy * a.bpm;
    JpegPlanes pl{};
    pl.ystride = 8 * R.h0 * a.mxx; pl.cstride = 8 * a.mxx;
    pl.y_fs = align256((size_t)pl.ystride * 8 * R.v0 * a.myy); pl.c_fs = gray ? 0 : align256((size_t)pl.cstride * 8 * a.myy);

    std::unique_ptr<ipx_jpeg_planes> own(new ipx_jpeg_planes);
    auto dalloc = [&](void **p, size_t bytes) {
        hipError_t e = hipMalloc(p, bytes ? bytes : 1);
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
    // scratch of this call, stream-ordered
    AsyncFree mem{s, {}};
    uint8_t *d_blob; JpegDecImage *d_img; JpegDecTables *d_tab; int16_t *d_coefs; int *d_status;
    if ((e = mem.get(&d_blob, blob_bytes + 16)) != hipSuccess) return fail(e, "scratch allocation");
    uint8_t *d_valid;
    if ((e = mem.get(&d_img, sizeof(JpegDecImage) * items.size())) != hipSuccess) return fail(e, "scratch allocation");
    if ((e = mem.get(&d_valid, (size_t)n)) != hipSuccess) return fail(e, "scratch allocation");
    if ((e = mem.get(&d_tab, sizeof(JpegDecTables) * n)) != hipSuccess) return fail(e, "scratch allocation");
    if ((e = mem.get(&d_coefs, (size_t)n * a.nblk * 128)) != hipSuccess) return fail(e, "scratch allocation");
    if ((e = mem.get(&d_status, sizeof(int) * n)) != hipSuccess) return fail(e, "scratch allocation");
    uint8_t *hblob = (uint8_t *)ipx_host_alloc(ctx, blob_bytes + 16);
    if (!hblob) { ipx_jpeg_planes_free(ctx, own.release()); return IPX_ERR_NOMEM; }
    parallel_for(n, [&](int i) { if (valid[i]) memcpy(hblob + blob_off[i], jpegs[i].data + info[i].scan_off, info[i].scan_len); });
    e = hipMemcpyAsync(d_blob, hblob, blob_bytes, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_img, items.data(), sizeof(JpegDecImage) * items.size(), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_valid, valid.data(), (size_t)n, hipMemcpyHostToDevice, s);
    pl.valid = d_valid;
    a.nitems = (int)items.size();
    if (e == hipSuccess) e = hipMemcpyAsync(d_tab, tabs.data(), sizeof(JpegDecTables) * n, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(d_coefs, 0, (size_t)n * a.nblk * 128, s);
    if (e == hipSuccess) e = hipMemsetAsync(d_status, 0, sizeof(int) * n, s);
    a.blob = d_blob; a.img = d_img; a.tab = d_tab; a.coefs = d_coefs; a.status = d_status;
    a.first_valid = ref;
    a.shared_tables = env_int("IPX_JPEG_SHARED_TABLES", 1);
    for (int i = 0; i < n && a.shared_tables; i++)
        if (valid[i] && (memcmp(tabs[i].lut, tabs[ref].lut, sizeof tabs[i].lut) || memcmp(tabs[i].maxcode, tabs[ref].maxcode, sizeof tabs[i].maxcode) ||
                              memcmp(tabs[i].valoff, tabs[ref].valoff, sizeof tabs[i].valoff) || memcmp(tabs[i].vals, tabs[ref].vals, sizeof tabs[i].vals)))
            a.shared_tables = 0;
    if (e == hipSuccess && a.nitems > 0) e = launch_jpeg_huff(a, s);
    if (e == hipSuccess && !par.empty()) {
        JpegParArgs P{};
        P.blob = d_blob; P.tab = d_tab; P.nimg = (int)par.size(); P.bpm = a.bpm; P.ybl = a.ybl; P.nblk = a.nblk;
        P.coefs = d_coefs; P.status = d_status;
        P.stage_rows = env_int("IPX_JPEG_PAR_STAGE", 0);   // measured: 109 ms staged (2 waves per CU) against 49 ms through L1 / L2 (1024 x 1080p)
        for (auto &pi : par) P.max_nsub = std::max(P.max_nsub, (int)pi.nsub);
        for (size_t k = 0; k < par.size(); k++) par[k].sub_off = k * (size_t)P.max_nsub;
        const size_t nsubs = par.size() * (size_t)P.max_nsub;
        JpegParImage *d_par = nullptr; uint32_t *d_tot = nullptr;
        if (e == hipSuccess) e = mem.get(&d_par, sizeof(JpegParImage) * par.size());
        if (e == hipSuccess) e = mem.get(&P.stuffed, nsubs * 4);
        if (e == hipSuccess) e = mem.get(&P.entry, nsubs * 8);
        if (e == hipSuccess) e = mem.get(&P.exit_a, nsubs * 8);
        if (e == hipSuccess) e = mem.get(&P.exit_b, nsubs * 8);
        if (e == hipSuccess) e = mem.get(&P.ends, nsubs * 4);
        if (e == hipSuccess) e = mem.get(&P.total_ends, par.size() * 4);
        if (e == hipSuccess) e = mem.get(&d_tot, par.size() * 4);
        if (e == hipSuccess) e = mem.get(&P.changed, 4);
        P.img = d_par;
        if (e == hipSuccess) e = hipMemcpyAsync(d_par, par.data(), sizeof(JpegParImage) * par.size(), hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipMemsetAsync(P.stuffed, 0, nsubs * 4, s);
        if (e == hipSuccess) e = hipMemsetAsync(P.entry, 0xff, nsubs * 8, s);
        if (e == hipSuccess) e = hipMemsetAsync(P.ends, 0, nsubs * 4, s);
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
    std::vector<int> dev_status(n, 0);
    if (e == hipSuccess) e = hipMemcpyAsync(dev_status.data(), d_status, sizeof(int) * n, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);   // the host tables and the blob may go now
    (void)ipx_host_free(ctx, hblob);
    if (e != hipSuccess) return fail(e, "jpeg decode");
    for (int i = 0; i < n; i++)
        if (status[i] == IPX_OK && dev_status[i]) status[i] = dev_status[i];
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
    LaneLease lane(ctx);
    hipStream_t s = lane->stream;
    int w = sw, h = sh;
    ipx_ycbcr_batch planes;
    ipx_jpeg_planes *owner = nullptr;
    int rc = ipx_jpeg_decode_batch(ctx, s, files, n, &w, &h, &planes, status, &owner);
    if (rc) return rc;
    if (!planes.y) return IPX_OK;                         // nothing decodable: every status says why
    struct Guard { ipx_ctx *c; ipx_jpeg_planes *o; ~Guard() { ipx_jpeg_planes_free(c, o); } } guard{ctx, owner};
    const size_t fres = resize_out ? align256(pl->info.resize_bytes) : 0, fth = thumb_out ? align256(pl->info.thumb_bytes) : 0;
    const size_t fwm = wm_out ? align256(pl->info.wm_bytes) : 0;
    const size_t cres = fres ? ipx_jpeg_coef_count(pl->info.resize_w, pl->info.resize_h) * 2 : 0;
    const size_t cth = fth ? ipx_jpeg_coef_count(pl->info.thumb_w, pl->info.thumb_h) * 2 : 0;
    const size_t cwm = fwm ? ipx_jpeg_coef_count(sw, sh) * 2 : 0;
    const size_t ccoef = align256(std::max(cres, std::max(cth, cwm)));
    const size_t per_frame = fres + fth + fwm + ccoef;
    if (per_frame == 0) return IPX_OK;
    const int chunk = std::max(1, std::min(n, env_int("IPX_JPEG_JPEG_CHUNK", 256)));
    rc = lane_reserve(lane.get(), per_frame * chunk + 256);
    if (rc) return rc;
    std::unique_ptr<ipx_jpeg_result> res(new ipx_jpeg_result);
    std::vector<size_t> offs(chunk), lens(chunk);
    for (int i0 = 0; i0 < n && !rc; i0 += chunk) {
        const int m = std::min(chunk, n - i0);
        uint8_t *base = (uint8_t *)(((uintptr_t)lane->dev + 255) & ~(uintptr_t)255);
        uint8_t *dres = fres ? base : nullptr, *dth = fth ? base + fres * chunk : nullptr, *dwm = fwm ? base + (fres + fth) * chunk : nullptr;
        int16_t *dcoef = (int16_t *)(base + (fres + fth + fwm) * chunk);
        ipx_ycbcr_batch d = planes;
        d.y += planes.y_frame_stride * i0;
        if (d.cb) { d.cb += planes.c_frame_stride * i0; d.cr += planes.c_frame_stride * i0; }
        if (planes.ratio == IPX_GRAY) rc = ipx_plan_run_dev_gray(ctx, s, pl, m, d.y, d.ystride, d.y_frame_stride, dres, fres, dth, fth, dwm, fwm);
        else rc = ipx_plan_run_dev_ycbcr(ctx, s, pl, m, &d, dres, fres, dth, fth, dwm, fwm);
        struct Out { uint8_t *dev; size_t fs; int w, h; ipx_bytes *dst; };
        const Out outs[3] = {{dres, fres, pl->info.resize_w, pl->info.resize_h, resize_out},
                             {dth, fth, pl->info.thumb_w, pl->info.thumb_h, thumb_out},
                             {dwm, fwm, sw, sh, wm_out}};
        for (int k = 0; k < 3 && !rc; k++) {
            const Out &o = outs[k];
            if (!o.dev || o.w <= 0 || o.h <= 0) continue;
            uint8_t *blob = nullptr;
            rc = jpeg_encode_core(ctx, s, dcoef, o.dev, o.w, o.h, o.w * 4, o.fs, m, quality, &blob, offs.data(), lens.data());
            if (rc) break;
            res->blobs.push_back(blob);
            for (int i = 0; i < m; i++)
                if (status[i0 + i] == IPX_OK) { o.dst[i0 + i].data = blob + offs[i]; o.dst[i0 + i].len = lens[i]; }
        }
    }
    (void)hipStreamSynchronize(s);
    if (rc) { ipx_jpeg_result_free(ctx, res.release()); return rc; }
    *result = res.release();
    return IPX_OK;
}


int ipx_plan_run_jpeg_jpeg(ipx_ctx *ctx, const ipx_plan *pl, int n, const ipx_bytes *files, int quality, ipx_bytes *resize_out,
                           ipx_bytes *thumb_out, ipx_bytes *wm_out, int *status, ipx_jpeg_result **result)
{
    IPX_ENTER(ctx);
    if (!pl || n < 0 || !files || !status || !result) { set_error("ipx_plan_run_jpeg_jpeg: bad argument"); return IPX_ERR_INVALID; }
    *result = nullptr;
    // a large batch is cut into parts that run on lanes of their own, one host thread each: while one part is in its (host-paced)
    // encode read-backs another decodes.  Two callers with 1024 files each measured 15 k images/s against 11.7 k for one.
    const int parts = std::max(1, std::min({(int)ctx->lanes.size(), n / std::max(1, env_int("IPX_JPEG_JPEG_PART", 384)), 3}));
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
    std::vector<std::thread> pool;
    for (int k = 1; k < parts; k++) pool.emplace_back(work, k);
    work(0);
    for (auto &t : pool) t.join();
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

}  // extern "C"

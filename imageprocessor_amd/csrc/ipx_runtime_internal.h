// ipx_runtime_internal.h -- the runtime's own types (context, lanes, glyph sets, plans) and the small helpers every
// translation unit that implements ABI entries needs (ipx_runtime.hip, ipx_jpeg_runtime.hip).  Not part of the ABI.
#pragma once

#include <sched.h>

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "ipx_internal.h"
#include "ipx_ks.h"

using namespace ipx;

// ---------------------------------------------------------------------------------------------
struct Lane {
    hipStream_t stream = nullptr;
    uint8_t *dev = nullptr;   // device scratch
    size_t dev_bytes = 0;
    int *flag = nullptr;      // device int for the opaque() scan
    uint8_t *dec = nullptr;   // second grow-only buffer: planes and scratch of a decode running on this lane (ipx_plan_run_jpeg_jpeg)
    size_t dec_bytes = 0;
    uint8_t *pin = nullptr;   // pinned bounce buffer for small single-frame calls on pageable memory (run_host_packed)
    size_t pin_bytes = 0;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};   // of the chunk in this lane's scratch: uploaded, computed, downloaded (run_host_packed)
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
    std::deque<void *> host_lru;                   // cached blocks, least recently freed first
    size_t host_cached = 0, host_cache_limit = (size_t)8 << 30;   // IPX_HOST_CACHE_MB
    // one row of 128s: the Cb / Cr "planes" (stride 0) that make a Gray frame a YCbCr frame with neutral chroma (ipx_plan_run_dev_gray)
    // compressed-in / compressed-out parts in flight (ipx_plan_run_jpeg_jpeg): capped at three per context.  Measured: two or three parts
    // side by side overlap nicely (one decodes while another is in its host-paced encode read-backs), four take 0.9 s EACH in their
    // decode step instead of 0.03 s -- whatever serialises there (allocation, synchronisation), a fourth part gains nothing
    int jj_active = 0;
    uint8_t *flat_chroma = nullptr;
    // plans (tap tables in HBM) and uploaded glyph sets by content, for callers that describe their operators per call
    // (ipx_plan_acquire: the per-operator seam, the pool): no hipMalloc / hipFree in the steady state
    std::mutex plan_mu;
    std::map<std::string, std::pair<ipx_glyphset *, ipx_plan *>> plan_cache;
    uint64_t plan_clock = 0;
    // axes of the kernel scaler in HBM by (destination extent, source extent), for the per-operation seam (ks_axis_get)
    std::mutex ks_mu;
    std::map<std::pair<int, int>, std::pair<uint8_t *, KsAxisDev>> ks_axes;
    static constexpr size_t kFlatChromaBytes = (size_t)64 << 10;
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
    KsAxis hx, hy;          // newDistrib of the two axes, as built on the host (the fused kernel's tables are cut from these)
    KsAxisDev ax[2]{};      // the same in the plan's blob: [0] horizontal, [1] vertical
};

struct ipx_plan {
    ipx_plan_params p{};
    ipx_plan_info info{};
    PlanScale sc[2];      // 0 = resize, 1 = thumbnail
    // tilings and tables of the one-pass kernel (ipx_ks_fused.hip) per LDS tile format -- [0] packed RGBA8 pixels (4 bytes), [1] four 16-bit
    // taps per pixel (8: NRGBA, YCbCr, deep sources), [2] one 16-bit tap (2: Gray); .ok = false: per-output kernels for that format
    KsFusedPlan fused[3];
    uint8_t *blob = nullptr;
    ClippedGlyphs glyphs;
    mutable std::mutex mu;
    ipx_glyphset *owned_gs = nullptr;         // a glyph set that lives and dies with this plan (ipx_plan_acquire)
    // a plan of the context's cache (guarded by ipx_ctx::plan_mu): calls holding it, and when it was last handed out
    int cache_refs = 0;
    uint64_t cache_stamp = 0;
};

inline int env_int(const char *name, int dflt)
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
        c_->cv.notify_all();     // waiters differ (one lane / every lane): wake them all, the predicates sort it out
    }
    Lane *operator->() { return lane_; }
    Lane &get() { return *lane_; }
private:
    ipx_ctx *c_;
    Lane *lane_ = nullptr;
};

inline int lane_reserve_dec(Lane &l, size_t bytes)
{
    if (bytes <= l.dec_bytes) return IPX_OK;
    if (l.dec) { IPX_HIP(hipStreamSynchronize(l.stream)); IPX_HIP(hipFree(l.dec)); l.dec = nullptr; l.dec_bytes = 0; }
    const size_t want = bytes + bytes / 8;        // a little headroom: batches of one size differ by a few files' worth of scan bytes
    hipError_t e = hipMalloc((void **)&l.dec, want);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("device allocation of %zu bytes failed: %s", want, hipGetErrorString(e));
        return IPX_ERR_NOMEM;
    }
    l.dec_bytes = want;
    return IPX_OK;
}

inline int lane_reserve(Lane &l, size_t bytes)
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

// status of a frame argument: IPX_ERR_INVALID for a malformed one, IPX_ERR_UNSUPPORTED for one beyond the addressable span
inline int frame_status(const char *who, const char *what, const void *p, int w, int h, long long stride, int bpp = 4)
{
    if (!p || w < 0 || h < 0 || stride < (long long)w * bpp) { set_error("%s: bad %s frame arguments", who, what); return IPX_ERR_INVALID; }
    if (!frame_span_ok(w, h, stride, bpp)) {
        set_error("%s: %s frame %dx%d (stride %lld) is beyond the 2 GiB / 65535-pixel span the kernels address", who, what, w, h, stride);
        return IPX_ERR_UNSUPPORTED;
    }
    return IPX_OK;
}
#define IPX_FRAME(who, what, p, w, h, stride) do { const int rc_ = frame_status(who, what, p, w, h, stride); if (rc_) return rc_; } while (0)

// The body of a worker thread: an exception must not leave the thread (std::terminate would take the Go / Python worker down);
// it becomes the status and text the spawning call reports.
template <class F> inline int guarded_status(F &&fn, std::string *text) noexcept
{
    try { fn(); return IPX_OK; }
    catch (...) {
        const int rc = status_of_exception();
        if (text) { try { *text = ipx_last_error(); } catch (...) { } }
        return rc;
    }
}

// CPUs local to a device's PCIe root, intersected with what the process may use.  Best effort: nothing happens if sysfs says nothing.
inline void bind_near_device(int device)
{
    if (env_int("IPX_POOL_NUMA", 1) == 0) return;
    char bus[32] = {0};
    if (hipDeviceGetPCIBusId(bus, sizeof bus, device) != hipSuccess) { (void)hipGetLastError(); return; }
    for (char *c = bus; *c; c++) *c = (char)tolower((unsigned char)*c);
    char path[128];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/local_cpulist", bus);
    FILE *f = fopen(path, "r");
    if (!f) return;
    char line[4096] = {0};
    const bool ok = fgets(line, sizeof line, f) != nullptr;
    fclose(f);
    if (!ok) return;
    cpu_set_t allowed, want;
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return;
    CPU_ZERO(&want);
    for (char *p = line; *p;) {            // "0-31,64-95"
        char *e = nullptr;
        long a = strtol(p, &e, 10);
        if (e == p) break;
        long b = a;
        if (*e == '-') { p = e + 1; b = strtol(p, &e, 10); }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++)
            if (CPU_ISSET((int)c, &allowed)) CPU_SET((int)c, &want);
        p = *e == ',' ? e + 1 : e;
        if (*e != ',') break;
    }
    if (CPU_COUNT(&want) > 0) (void)sched_setaffinity(0, sizeof want, &want);
}


// The device's view of `span` bytes of host memory a kernel may write: non-null when the range is pinned and mapped (hipHostMalloc,
// hipHostRegister), null for pageable memory.  Kernels then write outputs over the link themselves (run_host_packed, the pool's chunks).
inline uint8_t *pinned_device_view(uint8_t *host, size_t span)
{
    if (!host || !span) return nullptr;
    hipPointerAttribute_t at, last;
    if (hipPointerGetAttributes(&at, host) != hipSuccess || at.type != hipMemoryTypeHost || !at.devicePointer) { (void)hipGetLastError(); return nullptr; }
    if (hipPointerGetAttributes(&last, host + span - 1) != hipSuccess || last.type != hipMemoryTypeHost) { (void)hipGetLastError(); return nullptr; }   // a registered range may end before the batch does
    return (uint8_t *)at.devicePointer;
}

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// stream-ordered scratch of one call: freed (in stream order) when the call returns
struct AsyncFree {
    hipStream_t s;
    std::vector<void *> p;
    // optional arena (a lane's grow-only buffer): requests are bumped out of it while they fit, so a call that holds a lane allocates
    // nothing in the steady state.  Stream-ordered allocations of gigabytes per call turned out to stall for 0.5 - 5 s every few calls
    // on some boxes (the time sat in hipMallocAsync or behind it).
    uint8_t *arena = nullptr;
    size_t cap = 0, off = 0;
    ~AsyncFree() { for (void *q : p) (void)hipFreeAsync(q, s); }
    template <class T> hipError_t get(T **out, size_t bytes)
    {
        const size_t need = ((bytes ? bytes : 1) + 255) & ~(size_t)255;
        if (arena && off + need <= cap) { *out = (T *)(arena + off); off += need; return hipSuccess; }
        void *q = nullptr;
        hipError_t e = hipMallocAsync(&q, bytes ? bytes : 1, s);
        if (e == hipSuccess) p.push_back(q);
        *out = (T *)q;
        return e;
    }
};

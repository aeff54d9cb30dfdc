// ipx_pool.hip -- one process, several GPUs: a pool of contexts behind ONE largest-first queue, and the asynchronous job entries.
//
// The reference worker is one process with WORKER_CONCURRENCY goroutines pulling messages (worker.go:88-96, 112-149); messages are
// independent, nothing is exchanged.  A Go worker binds this: it submits a job (a batch of equally sized frames, or of JPEG files) and
// gets a ticket back at once, so no goroutine blocks an OS thread for the length of a batch; ipx_job_wait collects the result.
//
// The queue, the tickets and the feeder threads are ipx_pool_core.h (no GPU in it: tools/sanitize/pool_host_test.cpp runs it under
// ThreadSanitizer); this file is what a chunk does on a GPU, and the ABI.
// Inside: one ipx_ctx per pool slot (a device may be listed more than once), `lanes_per_device` feeder threads per slot, each with a
// HIP stream and a grow-only device buffer of its own.  A job is cut into chunks; chunks wait in one priority queue ordered by cost
// (bytes moved), and a feeder takes the most expensive chunk whenever it is free -- pull scheduling: a mixed batch balances itself
// (work stealing falls out of it), a uniform batch spreads round-robin over equally fast devices.  No data-path collective exists
// (SURVEY.md 8(e)): a frame never leaves the GPU it was uploaded to.
// Feeder threads bind themselves to the CPUs local to their GPU's PCIe root (sysfs local_cpulist) before they allocate and first-touch
// anything, and ipx_pool_host_alloc runs on such a thread, so pinned staging lands on the GPU's NUMA node.
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <queue>
#include <sched.h>
#include <string>
#include <thread>
#include <vector>

#include "ipx_pool_core.h"
#include "ipx_runtime_internal.h"

namespace {

struct PoolOps {                 // a deep copy of ipx_pool_ops: the caller may free its own right after submit
    ipx_pool_ops p{};
    std::vector<ipx_glyph> glyphs;
    std::vector<std::vector<uint8_t>> masks;
    std::string key;             // what a plan depends on, as bytes: the plan cache's key
};

struct JobState {
    ipx_job job{};
    PoolOps ops;
    std::vector<ipx_jpeg_result *> results;   // IPX_JOB_JPEG: pinned blocks the output streams live in, per slot that produced them
    std::vector<ipx_ctx *> result_ctx;        // (chunks append under the job's State::mu)
};
typedef ipx::PoolCore<JobState> Core;

struct Slot {
    int device = 0;
    ipx_ctx *ctx = nullptr;
    // every feeder of the slot uploads on this one stream: one copy engine at a time per direction is what the host link moves most
    // with (tools/link_streams.py: 1 stream up + 1 down 96 GB/s summed, 3 + 3 74); made by the slot's first feeder
    std::mutex up_mu;
    hipStream_t up_stream = nullptr;
};

}  // namespace

struct ipx_pool {
    std::vector<std::unique_ptr<Slot>> slots;
    size_t lane_bytes = (size_t)256 << 20;
    Core core;                      // last: its feeders stop before the slots go
};

namespace {

// bytes per pixel of a pixel job's frames; 0 for a kind that is none
int pixel_job_bpp(int kind)
{
    switch (kind) {
    case IPX_JOB_RGBA8: case IPX_JOB_NRGBA8: case IPX_JOB_CMYK: return 4;
    case IPX_JOB_GRAY8: return 1;
    case IPX_JOB_GRAY16: return 2;
    case IPX_JOB_NRGBA64: case IPX_JOB_RGBA64: return 8;
    default: return 0;
    }
}

double chunk_cost(const JobState &j, int m)
{
    if (j.job.kind == IPX_JOB_JPEG) {
        double b = 0;
        for (int i = 0; i < m; i++) b += (double)j.job.files[i].len;   // callers pass the chunk's own slice
        return b * 24 + (double)m * j.ops.p.sw * j.ops.p.sh * 8;         // decoded size dominates
    }
    return (double)m * ((double)j.ops.p.sw * j.ops.p.sh * (4 + pixel_job_bpp(j.job.kind)) + 4.0 * 1024 * 768);
}

int copy_ops(const ipx_pool_ops &in, PoolOps *out)
{
    out->p = in;
    if (in.n_glyphs < 0 || (in.n_glyphs && !in.glyphs)) { set_error("ipx_job_submit: bad glyph list"); return IPX_ERR_INVALID; }
    out->glyphs.assign(in.glyphs, in.glyphs + in.n_glyphs);
    out->masks.resize(in.n_glyphs);
    std::string key((const char *)&in, offsetof(ipx_pool_ops, glyphs));
    key.append((const char *)in.col, 4);
    for (int i = 0; i < in.n_glyphs; i++) {
        ipx_glyph &g = out->glyphs[i];
        if (g.mw < 0 || g.mh < 0 || (g.mw && g.mh && (!g.mask || g.mstride < g.mw))) { set_error("ipx_job_submit: glyph %d has a bad mask", i); return IPX_ERR_INVALID; }
        out->masks[i].resize((size_t)g.mw * g.mh);
        for (int y = 0; y < g.mh; y++) memcpy(out->masks[i].data() + (size_t)y * g.mw, g.mask + (size_t)y * g.mstride, g.mw);
        g.mask = out->masks[i].data();
        g.mstride = g.mw;
        key.append((const char *)&g.mw, sizeof(int32_t) * 3);
        key.append((const char *)&g.dr, sizeof g.dr + 2 * sizeof(int32_t));
        key.append((const char *)out->masks[i].data(), out->masks[i].size());
    }
    out->p.glyphs = out->glyphs.data();
    out->key = std::move(key);
    return IPX_OK;
}

struct Feeder {                     // what one feeder thread owns; made and destroyed on that thread
    hipStream_t stream = nullptr;
    hipEvent_t uploaded = nullptr;
    uint8_t *dev = nullptr;
    size_t dev_bytes = 0;
    ~Feeder()
    {
        if (dev) (void)hipFree(dev);
        if (stream) { (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); }
        if (uploaded) (void)hipEventDestroy(uploaded);
    }
};

int feeder_reserve(Feeder &f, size_t bytes)
{
    if (bytes <= f.dev_bytes) return IPX_OK;
    if (f.dev) { (void)hipStreamSynchronize(f.stream); (void)hipFree(f.dev); f.dev = nullptr; f.dev_bytes = 0; }
    if (hipMalloc((void **)&f.dev, bytes) != hipSuccess) { (void)hipGetLastError(); set_error("pool: device allocation of %zu bytes failed", bytes); return IPX_ERR_NOMEM; }
    f.dev_bytes = bytes;
    return IPX_OK;
}

// pixels in, pixels out: upload the chunk, one fused pass, download -- all on the feeder's stream
int run_pixel_chunk(Slot &s, Feeder &f, const JobState &j, ipx_plan *plan, int i0, int m)
{
    const ipx_job &q = j.job;
    ipx_plan_info info;
    int rc = ipx_plan_query(plan, &info);
    if (rc) return rc;
    const int sw = j.ops.p.sw, sh = j.ops.p.sh;
    const int bpp = pixel_job_bpp(q.kind);
    const size_t fsrc = align256((size_t)sw * sh * bpp);
    const size_t fres = q.resize_out ? align256(info.resize_bytes) : 0, fth = q.thumb_out ? align256(info.thumb_bytes) : 0;
    const size_t fwm = q.wm_out ? align256(info.wm_bytes) : 0;
    // outputs in pinned memory are written by the kernels themselves, over the link (run_host_packed in ipx_runtime.hip has the why)
    uint8_t *vres = nullptr, *vth = nullptr, *vwm = nullptr;
    bool direct = env_int("IPX_HOST_DIRECT", 1) != 0;
    if (((uintptr_t)q.resize_out | (uintptr_t)q.thumb_out | (uintptr_t)q.wm_out | (fres ? q.resize_frame_stride : 0) | (fth ? q.thumb_frame_stride : 0) |
         (fwm ? q.wm_frame_stride : 0)) & 15) direct = false;      // (16-byte stores: other strides go through the feeder's scratch)
    if (direct && fres) direct = (vres = pinned_device_view(q.resize_out + (size_t)i0 * q.resize_frame_stride, q.resize_frame_stride * (m - 1) + info.resize_bytes)) != nullptr;
    if (direct && fth) direct = (vth = pinned_device_view(q.thumb_out + (size_t)i0 * q.thumb_frame_stride, q.thumb_frame_stride * (m - 1) + info.thumb_bytes)) != nullptr;
    if (direct && fwm) direct = (vwm = pinned_device_view(q.wm_out + (size_t)i0 * q.wm_frame_stride, q.wm_frame_stride * (m - 1) + info.wm_bytes)) != nullptr;
    rc = feeder_reserve(f, (fsrc + (direct ? 0 : fres + fth + fwm)) * (size_t)m + 256);
    if (rc) return rc;
    uint8_t *dsrc = f.dev, *dres = fres ? dsrc + fsrc * m : nullptr, *dth = fth ? dsrc + (fsrc + fres) * m : nullptr;
    uint8_t *dwm = fwm ? dsrc + (fsrc + fres + fth) * m : nullptr;
    size_t sres = fres, sth = fth, swm = fwm;
    if (direct) {
        dres = vres; sres = q.resize_frame_stride;
        dth = vth; sth = q.thumb_frame_stride;
        dwm = vwm; swm = q.wm_frame_stride;
    }
    hipError_t e = hipSuccess;
    // direct outputs: the upload goes on the slot's shared stream and the feeder's own stream (kernels) waits for it
    const hipStream_t up = direct && s.up_stream && f.uploaded ? s.up_stream : f.stream;
    if (q.sstride == sw * bpp)
        for (int i = 0; i < m && e == hipSuccess; i++)
            e = hipMemcpyAsync(dsrc + fsrc * i, q.src + (size_t)(i0 + i) * q.src_frame_stride, (size_t)sw * sh * bpp, hipMemcpyHostToDevice, up);
    else for (int i = 0; i < m && e == hipSuccess; i++)
        e = hipMemcpy2DAsync(dsrc + fsrc * i, (size_t)sw * bpp, q.src + (size_t)(i0 + i) * q.src_frame_stride, q.sstride, (size_t)sw * bpp, sh,
                             hipMemcpyHostToDevice, up);
    if (e == hipSuccess && up != f.stream) {
        e = hipEventRecord(f.uploaded, up);
        if (e == hipSuccess) e = hipStreamWaitEvent(f.stream, f.uploaded, 0);
    }
    if (e != hipSuccess && up != f.stream) (void)hipStreamSynchronize(up);
    if (e != hipSuccess) { (void)hipStreamSynchronize(f.stream); set_error("pool: upload failed: %s", hipGetErrorString(e)); return IPX_ERR_HIP; }
    switch (q.kind) {
    case IPX_JOB_RGBA8: rc = ipx_plan_run_dev(s.ctx, f.stream, plan, m, dsrc, sw * 4, fsrc, dres, sres, dth, sth, dwm, swm); break;
    case IPX_JOB_NRGBA8: rc = ipx_plan_run_dev_nrgba(s.ctx, f.stream, plan, m, dsrc, sw * 4, fsrc, dres, sres, dth, sth, dwm, swm); break;
    case IPX_JOB_GRAY8: rc = ipx_plan_run_dev_gray(s.ctx, f.stream, plan, m, dsrc, sw, fsrc, dres, sres, dth, sth, dwm, swm); break;
    default:
        rc = ipx_plan_run_dev_deep(s.ctx, f.stream, plan, m, q.kind == IPX_JOB_NRGBA64 ? IPX_DEEP_NRGBA64 : q.kind == IPX_JOB_RGBA64 ? IPX_DEEP_RGBA64 :
                                   q.kind == IPX_JOB_GRAY16 ? IPX_DEEP_GRAY16 : IPX_DEEP_CMYK, dsrc, sw * bpp, fsrc, dres, sres, dth, sth, dwm, swm);
        break;
    }
    if (rc) { (void)hipStreamSynchronize(f.stream); return rc; }
    for (int i = 0; i < m && e == hipSuccess && !direct; i++) {
        if (dres && info.resize_bytes) e = hipMemcpyAsync(q.resize_out + (size_t)(i0 + i) * q.resize_frame_stride, dres + fres * i, info.resize_bytes, hipMemcpyDeviceToHost, f.stream);
        if (e == hipSuccess && dth && info.thumb_bytes) e = hipMemcpyAsync(q.thumb_out + (size_t)(i0 + i) * q.thumb_frame_stride, dth + fth * i, info.thumb_bytes, hipMemcpyDeviceToHost, f.stream);
        if (e == hipSuccess && dwm && info.wm_bytes) e = hipMemcpyAsync(q.wm_out + (size_t)(i0 + i) * q.wm_frame_stride, dwm + fwm * i, info.wm_bytes, hipMemcpyDeviceToHost, f.stream);
    }
    const hipError_t e2 = hipStreamSynchronize(f.stream);
    if (e == hipSuccess) e = e2;
    if (e != hipSuccess) { set_error("pool: download failed: %s", hipGetErrorString(e)); return IPX_ERR_HIP; }
    return IPX_OK;
}

// one chunk on this feeder; never throws (an exception becomes a status, as at the ABI)
int run_chunk(Slot &s, Feeder &f, JobState &j, int i0, int m, ipx_jpeg_result **res) noexcept
{
    try {
        if (!f.stream) { set_error("pool: the feeder has no stream"); return IPX_ERR_HIP; }
        ipx_plan *plan = nullptr;
        int cached = 0;
        int rc = ipx_plan_acquire(s.ctx, &j.ops.p, &plan, &cached);     // plans and glyph sets by content, per context
        if (rc) return rc;
        if (j.job.kind == IPX_JOB_JPEG)
            rc = ipx_plan_run_jpeg_jpeg(s.ctx, plan, m, j.job.files + i0, j.job.quality, j.job.resize_jpeg ? j.job.resize_jpeg + i0 : nullptr,
                                        j.job.thumb_jpeg ? j.job.thumb_jpeg + i0 : nullptr, j.job.wm_jpeg ? j.job.wm_jpeg + i0 : nullptr,
                                        j.job.status + i0, res);
        else rc = run_pixel_chunk(s, f, j, plan, i0, m);     // (both return with the chunk's GPU work finished)
        if (!cached) {
            std::string keep = rc ? ipx_last_error() : "";
            ipx_plan_release(s.ctx, plan, cached);
            if (rc) set_error("%s", keep.c_str());
        }
        return rc;
    } catch (...) {
        return status_of_exception();
    }
}

// a feeder's chunk function, made on the feeder's own thread (bound next to the slot's GPU before it allocates anything)
Core::ChunkFn make_feeder(ipx_pool *pool, int slot_index)
{
    Slot &s = *pool->slots[slot_index];
    (void)hipSetDevice(s.device);
    bind_near_device(s.device);
    std::shared_ptr<Feeder> f(new Feeder);
    if (hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); f->stream = nullptr; }
    if (hipEventCreateWithFlags(&f->uploaded, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); f->uploaded = nullptr; }
    {
        std::lock_guard<std::mutex> lk(s.up_mu);
        if (!s.up_stream && hipStreamCreateWithFlags(&s.up_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); s.up_stream = nullptr; }
    }
    return [&s, f](Core::State &st, int i0, int m, std::string *error) -> int {
        ipx_jpeg_result *res = nullptr;
        const int rc = run_chunk(s, *f, st.user, i0, m, &res);
        if (rc) *error = ipx_last_error();
        if (res) {
            std::lock_guard<std::mutex> lk(st.mu);
            st.user.results.push_back(res);
            st.user.result_ctx.push_back(s.ctx);
        }
        return rc;
    };
}

int job_check(const ipx_job *job)
{
    if (!job || job->n < 0) { set_error("ipx_job_submit: bad job"); return IPX_ERR_INVALID; }
    const ipx_pool_ops &o = job->ops;
    if (o.sw <= 0 || o.sh <= 0) { set_error("ipx_job_submit: frame size %dx%d", o.sw, o.sh); return IPX_ERR_INVALID; }
    if (const int bpp = pixel_job_bpp(job->kind)) {
        if (job->n && (!job->src || (long long)job->sstride < (long long)o.sw * bpp)) { set_error("ipx_job_submit: bad source frames"); return IPX_ERR_INVALID; }
        if (!frame_span_ok(o.sw, o.sh, job->sstride, bpp) || (bpp != 1 && !frame_span_ok(o.sw, o.sh, (long long)o.sw * std::max(bpp, 4), std::max(bpp, 4)))) {
            set_error("ipx_job_submit: %dx%d frames are beyond the span the kernels address", o.sw, o.sh);
            return IPX_ERR_UNSUPPORTED;
        }
    } else if (job->kind == IPX_JOB_JPEG) {
        if (job->n && (!job->files || !job->status)) { set_error("ipx_job_submit: a JPEG job needs files and a status array"); return IPX_ERR_INVALID; }
    } else {
        set_error("ipx_job_submit: unknown job kind %d", job->kind);
        return IPX_ERR_INVALID;
    }
    return IPX_OK;
}

}  // namespace

extern "C" {

int ipx_pool_create(const int *devices, int n_devices, const ipx_pool_config *cfg, ipx_pool **out) try
{
    clear_error();
    if (!out || n_devices <= 0 || n_devices > 64 || !devices) { set_error("ipx_pool_create: bad argument"); return IPX_ERR_INVALID; }
    *out = nullptr;
    const int lanes = cfg && cfg->lanes_per_device > 0 ? cfg->lanes_per_device : 4;   // (four chunks in flight per device keep the link busy: 3 measured 63 - 72 ms per 256 x 1080p, 4 63 - 64)
    std::unique_ptr<ipx_pool> pool(new ipx_pool);
    if (cfg && cfg->lane_bytes) pool->lane_bytes = cfg->lane_bytes;
    for (int i = 0; i < n_devices; i++) {
        std::unique_ptr<Slot> s(new Slot);
        s->device = devices[i];
        ipx_config cc;
        cc.device = devices[i]; cc.lanes = lanes + 1; cc.lane_bytes = 0;
        const int rc = ipx_create(&cc, &s->ctx);
        if (rc) {
            for (auto &t : pool->slots) ipx_destroy(t->ctx);
            return rc;
        }
        pool->slots.push_back(std::move(s));
    }
    ipx_pool *raw = pool.get();
    pool->core.start(n_devices, lanes, [raw](int slot) { return make_feeder(raw, slot); });
    *out = pool.release();
    return IPX_OK;
}
IPX_CATCH_STATUS

void ipx_pool_destroy(ipx_pool *pool)
{
    if (!pool) return;
    pool->core.stop();                           // feeders drain the queue before they leave
    for (auto &j : pool->core.leftovers())
        for (size_t i = 0; i < j->user.results.size(); i++) ipx_jpeg_result_free(j->user.result_ctx[i], j->user.results[i]);
    for (auto &s : pool->slots) {
        if (s->up_stream) { (void)hipSetDevice(s->device); (void)hipStreamSynchronize(s->up_stream); (void)hipStreamDestroy(s->up_stream); }
        ipx_destroy(s->ctx);     // (with the plans its cache holds)
    }
    delete pool;
}

int ipx_pool_slots(const ipx_pool *pool) { return pool ? (int)pool->slots.size() : 0; }

long long ipx_pool_frames_done(const ipx_pool *pool, int slot)
{
    return pool && slot >= 0 && slot < (int)pool->slots.size() ? pool->core.units_done(slot) : -1;
}

void *ipx_pool_host_alloc(ipx_pool *pool, int slot, size_t bytes) try
{
    clear_error();
    if (!pool || slot < 0 || slot >= (int)pool->slots.size() || !bytes) { set_error("ipx_pool_host_alloc: bad argument"); return nullptr; }
    Slot &s = *pool->slots[slot];
    void *p = nullptr;
    std::string text;
    // (ipx_host_alloc itself pins fresh blocks on a thread bound next to the context's GPU)
    p = ipx_host_alloc(s.ctx, bytes);
    if (p) memset(p, 0, bytes);
    else text = ipx_last_error();
    if (!p) set_error("%s", text.c_str());
    return p;
}
catch (...) { (void)status_of_exception(); return nullptr; }

int ipx_pool_host_free(ipx_pool *pool, int slot, void *p)
{
    clear_error();
    if (!pool || slot < 0 || slot >= (int)pool->slots.size()) { set_error("ipx_pool_host_free: bad argument"); return IPX_ERR_INVALID; }
    return ipx_host_free(pool->slots[slot]->ctx, p);
}

int ipx_job_submit(ipx_pool *pool, const ipx_job *job, ipx_ticket *ticket) try
{
    clear_error();
    if (!pool || !ticket) { set_error("ipx_job_submit: bad argument"); return IPX_ERR_INVALID; }
    int rc = job_check(job);
    if (rc) return rc;
    std::shared_ptr<Core::State> st(new Core::State);
    JobState *j = &st->user;
    j->job = *job;
    rc = copy_ops(job->ops, &j->ops);
    if (rc) return rc;
    if (job->kind == IPX_JOB_JPEG)
        for (int i = 0; i < job->n; i++) job->status[i] = IPX_OK;
    // chunks: about lane_bytes of frames each, and at least two per feeder so that uploads, kernels and downloads of different
    // chunks overlap; a JPEG job goes in parts of a few hundred files (ipx_plan_run_jpeg_jpeg pipelines inside a part)
    int per;
    if (job->kind == IPX_JOB_JPEG) per = std::max(1, env_int("IPX_POOL_JPEG_CHUNK", 256));
    else {
        const size_t fb = (size_t)job->ops.sw * job->ops.sh * (4 + pixel_job_bpp(job->kind)) + ((size_t)4 << 20);
        per = (int)std::max<size_t>(1, pool->lane_bytes / fb);
        const int want = 2 * pool->core.feeders();
        per = std::max(1, std::min(per, (job->n + want - 1) / std::max(1, want)));
    }
    std::vector<Core::Piece> pieces;
    for (int i0 = 0; i0 < job->n; i0 += per) {
        Core::Piece c;
        c.i0 = i0; c.m = std::min(per, job->n - i0);
        if (job->kind == IPX_JOB_JPEG) {
            JobState tmp;            // cost of this slice: its own file sizes
            tmp.job = *job; tmp.job.files = job->files + i0; tmp.ops.p = job->ops;
            c.cost = chunk_cost(tmp, c.m);
        } else c.cost = chunk_cost(*j, c.m);
        pieces.push_back(c);
    }
    uint64_t t = 0;
    if (!pool->core.submit(st, pieces, &t)) { set_error("ipx_job_submit: the pool is shutting down"); return IPX_ERR_INVALID; }
    *ticket = t;
    return IPX_OK;
}
IPX_CATCH_STATUS

int ipx_job_wait(ipx_pool *pool, ipx_ticket ticket, int *frames_done) try
{
    clear_error();
    if (!pool) { set_error("ipx_job_wait: null pool"); return IPX_ERR_INVALID; }
    std::shared_ptr<Core::State> j = pool->core.wait(ticket);
    if (!j) { set_error("ipx_job_wait: unknown ticket"); return IPX_ERR_INVALID; }
    if (frames_done) *frames_done = j->units_done;
    if (j->status) set_error("%s", j->error.c_str());
    return j->status;
}
IPX_CATCH_STATUS

int ipx_job_poll(ipx_pool *pool, ipx_ticket ticket, int *done)
{
    clear_error();
    if (!pool || !done) { set_error("ipx_job_poll: bad argument"); return IPX_ERR_INVALID; }
    const int d = pool->core.poll(ticket);
    if (d < 0) { set_error("ipx_job_poll: unknown ticket"); return IPX_ERR_INVALID; }
    *done = d;
    return IPX_OK;
}

int ipx_job_release(ipx_pool *pool, ipx_ticket ticket) try
{
    clear_error();
    if (!pool) { set_error("ipx_job_release: null pool"); return IPX_ERR_INVALID; }
    std::shared_ptr<Core::State> j = pool->core.release(ticket);     // releasing a running job waits for it
    if (!j) { set_error("ipx_job_release: unknown ticket"); return IPX_ERR_INVALID; }
    for (size_t i = 0; i < j->user.results.size(); i++) ipx_jpeg_result_free(j->user.result_ctx[i], j->user.results[i]);
    return IPX_OK;
}
IPX_CATCH_STATUS

int ipx_pool_run_host(ipx_pool *pool, const ipx_job *jobs, int n_jobs) try
{
    clear_error();
    if (!pool || n_jobs < 0 || (n_jobs && !jobs)) { set_error("ipx_pool_run_host: bad argument"); return IPX_ERR_INVALID; }
    for (int i = 0; i < n_jobs; i++)
        if (!pixel_job_bpp(jobs[i].kind)) { set_error("ipx_pool_run_host: pixel jobs only (JPEG jobs keep their outputs until ipx_job_release)"); return IPX_ERR_INVALID; }
    std::vector<ipx_ticket> tickets;
    int rc = IPX_OK;
    std::string text;
    for (int i = 0; i < n_jobs && !rc; i++) {
        ipx_ticket t = 0;
        rc = ipx_job_submit(pool, &jobs[i], &t);
        if (rc) text = ipx_last_error();
        else tickets.push_back(t);
    }
    for (ipx_ticket t : tickets) {
        const int r = ipx_job_wait(pool, t, nullptr);
        if (r && !rc) { rc = r; text = ipx_last_error(); }
        (void)ipx_job_release(pool, t);
    }
    if (rc) set_error("%s", text.c_str());
    return rc;
}
IPX_CATCH_STATUS

}  // extern "C"

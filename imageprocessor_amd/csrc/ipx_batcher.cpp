// ipx_batcher.cpp -- see ipx_batcher.h.  Reference: internal/worker/worker.go:88-96 (channel of concurrency * 2), :112-149 (one
// message per goroutine), :165-234 (processMessage: the message is committed only after its objects are saved -- the batcher keeps that:
// a submitter gets its result back on its own ticket and commits itself).
#include "ipx_batcher.h"

#include <cstdlib>

#include <algorithm>

namespace ipx {

Batcher::Batcher(const BatchBackend &be, int max_batch, int max_wait_us, int quality, int idle_jobs)
    : be_(be), max_batch_(std::max(1, max_batch)), quality_(quality), max_wait_(std::max(0, max_wait_us)), idle_jobs_(std::max(0, idle_jobs))
{
    if (const char *e = getenv("IPX_BATCHER_IDLE_FLUSH")) idle_jobs_ = std::max(0, atoi(e));
    timer_ = std::thread([this] { timer_loop(); });
}

Batcher::~Batcher()
{
    std::vector<std::shared_ptr<Batch>> rest;
    {
        std::lock_guard<std::mutex> lk(mu_);
        stop_ = true;
        for (auto &kv : pending_) rest.push_back(kv.second);
        pending_.clear();
    }
    cv_timer_.notify_all();
    if (timer_.joinable()) timer_.join();
    for (auto &b : rest) { { std::lock_guard<std::mutex> lk(mu_); running_++; } flush(b, ByTimer); }
    // every job still held: wait for it and hand its blocks back (tickets nobody collected)
    std::vector<std::shared_ptr<Batch>> all;
    {
        std::lock_guard<std::mutex> lk(mu_);
        for (auto &kv : tickets_) all.push_back(kv.second.first);
        tickets_.clear();
    }
    std::sort(all.begin(), all.end());
    all.erase(std::unique(all.begin(), all.end()), all.end());
    for (auto &b : all) {
        bool need = false;
        {
            std::lock_guard<std::mutex> lk(mu_);
            need = b->flushed && !b->submit_failed && !b->job_released;
            b->job_released = true;
        }
        if (need) { (void)be_.wait(be_.self, b->job); (void)be_.release(be_.self, b->job); }
    }
}

// deep copy + the grouping key: frame size, operator parameters, colour, every glyph's geometry and mask bytes
int Batcher::copy_ops(const ipx_pool_ops &in, OpsCopy *out, std::string *key, std::string *err)
{
    if (in.sw <= 0 || in.sh <= 0) { *err = "ipx_batcher_submit: the frame size of the file (ops.sw, ops.sh) is required"; return IPX_ERR_INVALID; }
    if (in.n_glyphs < 0 || (in.n_glyphs && !in.glyphs)) { *err = "ipx_batcher_submit: bad glyph list"; return IPX_ERR_INVALID; }
    out->p = in;
    key->assign((const char *)&in, offsetof(ipx_pool_ops, glyphs));
    key->append((const char *)in.col, 4);
    out->glyphs.assign(in.glyphs, in.glyphs + in.n_glyphs);
    out->masks.resize((size_t)in.n_glyphs);
    for (int i = 0; i < in.n_glyphs; i++) {
        const ipx_glyph &g = in.glyphs[i];
        if (g.mw < 0 || g.mh < 0 || (g.mw && g.mh && (!g.mask || g.mstride < g.mw))) { *err = "ipx_batcher_submit: a glyph has a bad mask"; return IPX_ERR_INVALID; }
        key->append((const char *)&g.mw, sizeof(int32_t) * 2);
        key->append((const char *)&g.dr, sizeof g.dr);
        key->append((const char *)&g.mpx, sizeof(int32_t) * 2);
        out->masks[i].resize((size_t)g.mw * g.mh);
        for (int y = 0; y < g.mh; y++) {
            memcpy(out->masks[i].data() + (size_t)y * g.mw, g.mask + (size_t)y * g.mstride, (size_t)g.mw);
            key->append((const char *)g.mask + (size_t)y * g.mstride, (size_t)g.mw);
        }
        out->glyphs[i].mask = out->masks[i].data();
        out->glyphs[i].mstride = g.mw;
    }
    out->p.glyphs = out->glyphs.empty() ? nullptr : out->glyphs.data();
    return IPX_OK;
}

// Components and luma sampling factors of a JPEG file's frame header (B.2.2), zeros for anything else.  A batch is one shape --
// ipx_plan_run_jpeg_jpeg takes the shape of its first decodable file and hands the others back with a status -- so the shape is part of
// the grouping key: a Gray upload among colour ones gets a batch of its own kind (the reference decodes each message by itself,
// image_processor.go:47) instead of being handed back or, arriving first, having its neighbours handed back.
static void jpeg_shape(const ipx_bytes &f, uint8_t out[3])
{
    out[0] = out[1] = out[2] = 0;
    const uint8_t *p = (const uint8_t *)f.data;
    const size_t n = f.len;
    if (n < 4 || p[0] != 0xff || p[1] != 0xd8) return;
    size_t i = 2;
    while (i + 4 <= n && p[i] == 0xff) {
        const uint8_t m = p[i + 1];
        if (m == 0xff) { i++; continue; }                                          // fill byte
        if (m == 0x01 || (m >= 0xd0 && m <= 0xd8)) { i += 2; continue; }            // markers without a segment
        if (m == 0xd9 || m == 0xda) return;                                        // no frame header before the scan
        const size_t len = (size_t)p[i + 2] << 8 | p[i + 3];
        if (len < 2 || i + 2 + len > n) return;
        if (m >= 0xc0 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc) {        // SOFn
            if (len >= 11 && p[i + 9] >= 1) { out[0] = p[i + 9]; out[1] = p[i + 11] >> 4; out[2] = p[i + 11] & 15; }
            return;
        }
        i += 2 + len;
    }
}

int Batcher::submit(const ipx_bytes &file, const ipx_pool_ops &ops, uint64_t *ticket, std::string *err)
{
    if (!ticket || !file.data || !file.len) { *err = "ipx_batcher_submit: bad argument"; return IPX_ERR_INVALID; }
    OpsCopy oc;
    std::string key;
    const int rc = copy_ops(ops, &oc, &key, err);
    if (rc) return rc;
    uint8_t shape[3];
    jpeg_shape(file, shape);
    key.append((const char *)shape, 3);
    std::shared_ptr<Batch> full;
    Why why = BySize;
    {
        std::lock_guard<std::mutex> lk(mu_);
        if (stop_) { *err = "ipx_batcher_submit: the batcher is shutting down"; return IPX_ERR_INVALID; }
        auto it = pending_.find(key);
        if (it == pending_.end()) {
            auto b = std::make_shared<Batch>();
            b->key = key;
            b->ops = std::move(oc);
            if (!b->ops.glyphs.empty()) {                       // the moved vectors kept their buffers; the pointers into them stay good
                for (size_t i = 0; i < b->ops.glyphs.size(); i++) b->ops.glyphs[i].mask = b->ops.masks[i].data();
                b->ops.p.glyphs = b->ops.glyphs.data();
            }
            b->deadline = std::chrono::steady_clock::now() + max_wait_;
            it = pending_.emplace(key, b).first;
            cv_timer_.notify_all();                             // a new earliest deadline, perhaps
        }
        Batch &b = *it->second;
        b.files.push_back(file);
        *ticket = next_ticket_++;
        tickets_[*ticket] = {it->second, (int)b.files.size() - 1};
        b.unreleased++;
        stats_.files++;
        if ((int)b.files.size() >= max_batch_) { full = it->second; pending_.erase(it); }
        // Nothing of this batcher runs on the GPU: waiting for company would only add the wait to the file's latency.  Under load
        // jobs are running, files gather while they do, and the group leaves when the last running job has been seen to finish
        // (job_seen_done), by size or by the timer -- batches form by themselves exactly when there is something to wait for.
        else if (running_ < idle_jobs_) { full = it->second; pending_.erase(it); why = WhenIdle; }
        if (full) running_++;                                   // (counted before the lock goes: a second submitter must not see "idle")
    }
    if (full) flush(full, why);
    return IPX_OK;
}

void Batcher::flush(const std::shared_ptr<Batch> &b, Why why)
{
    const int n = (int)b->files.size();
    b->res.assign(n, ipx_bytes{nullptr, 0}); b->th.assign(n, ipx_bytes{nullptr, 0}); b->wm.assign(n, ipx_bytes{nullptr, 0});
    b->status.assign(n, IPX_OK);
    ipx_job j;
    memset(&j, 0, sizeof j);
    j.kind = IPX_JOB_JPEG;
    j.ops = b->ops.p;
    j.n = n;
    j.files = b->files.data();
    j.quality = quality_;
    j.resize_jpeg = b->ops.p.do_resize ? b->res.data() : nullptr;
    j.thumb_jpeg = b->ops.p.do_thumbnail ? b->th.data() : nullptr;
    j.wm_jpeg = b->ops.p.do_watermark ? b->wm.data() : nullptr;
    j.status = b->status.data();
    ipx_ticket t = 0;
    const int rc = be_.submit(be_.self, &j, &t);
    std::string text = rc && be_.last_error ? be_.last_error() : "";
    {
        std::lock_guard<std::mutex> lk(mu_);
        b->job = t;
        b->rc = rc;
        b->submit_failed = rc != IPX_OK;
        b->error = text;
        b->flushed = true;
        if (b->submit_failed) { b->done = true; running_--; }     // (the caller counted it as running)
        stats_.batches++;
        if (why == ByTimer) stats_.flushed_by_timer++;
        else if (why == BySize) stats_.flushed_by_size++;
        else stats_.flushed_when_idle++;
        stats_.largest_batch = std::max<long long>(stats_.largest_batch, n);
    }
    b->cv.notify_all();
}

void Batcher::timer_loop()
{
    std::unique_lock<std::mutex> lk(mu_);
    while (!stop_) {
        if (pending_.empty()) { cv_timer_.wait(lk); continue; }
        auto first = pending_.begin();
        for (auto it = pending_.begin(); it != pending_.end(); ++it)
            if (it->second->deadline < first->second->deadline) first = it;
        const auto when = first->second->deadline;
        const auto now = std::chrono::steady_clock::now();
        if (now < when) {                                     // (then re-evaluate: the set may have changed)
#if defined(__SANITIZE_THREAD__)
            // gcc 11's ThreadSanitizer does not intercept pthread_cond_clockwait (what a steady_clock wait becomes on glibc >= 2.30) and
            // then loses track of the mutex: wait on the system clock in the sanitizer build only
            cv_timer_.wait_until(lk, std::chrono::system_clock::now() + (when - now));
#else
            cv_timer_.wait_until(lk, when);
#endif
            continue;
        }
        std::shared_ptr<Batch> b = first->second;
        pending_.erase(first);
        running_++;
        lk.unlock();
        flush(b, ByTimer);
        lk.lock();
    }
}

int Batcher::wait(uint64_t ticket, ipx_batch_result *res, std::string *err)
{
    std::shared_ptr<Batch> b;
    int idx = 0;
    {
        std::unique_lock<std::mutex> lk(mu_);
        auto it = tickets_.find(ticket);
        if (it == tickets_.end()) { *err = "ipx_batcher_wait: unknown ticket"; return IPX_ERR_INVALID; }
        b = it->second.first; idx = it->second.second;
        b->cv.wait(lk, [&] { return b->flushed; });
        if (b->submit_failed) { *err = b->error; return b->rc; }
    }
    const int rc = be_.wait(be_.self, b->job);          // the pool lets any number of threads wait for one job
    job_seen_done(b);
    if (res) {
        res->status = rc ? rc : b->status[idx];
        res->resize = b->res[idx]; res->thumb = b->th[idx]; res->wm = b->wm[idx];
    }
    if (rc) { *err = be_.last_error ? be_.last_error() : ""; return rc; }
    return IPX_OK;                                      // (a file the GPU path cannot take has its own status in *res; the call succeeded)
}

int Batcher::release(uint64_t ticket, std::string *err)
{
    std::shared_ptr<Batch> b;
    bool last = false;
    {
        std::unique_lock<std::mutex> lk(mu_);
        auto it = tickets_.find(ticket);
        if (it == tickets_.end()) { *err = "ipx_batcher_release: unknown ticket"; return IPX_ERR_INVALID; }
        b = it->second.first;
        b->cv.wait(lk, [&] { return b->flushed; });     // releasing a file that is still waiting for company waits for its batch
        tickets_.erase(it);
        last = --b->unreleased == 0 && !b->submit_failed && !b->job_released;
        if (last) b->job_released = true;
    }
    // a file's bytes are read until its batch has run: whoever lets go of a ticket -- with or without having waited for it -- may free
    // the file once this returns
    if (!b->submit_failed) { (void)be_.wait(be_.self, b->job); job_seen_done(b); }
    if (last) return be_.release(be_.self, b->job);     // the batch's output blocks are shared: they go back with its last file
    return IPX_OK;
}

void Batcher::job_seen_done(const std::shared_ptr<Batch> &b)
{
    std::vector<std::shared_ptr<Batch>> go;
    {
        std::lock_guard<std::mutex> lk(mu_);
        if (b->done) return;
        b->done = true;
        running_--;
        if (running_ >= idle_jobs_ || stop_) return;
        for (auto &kv : pending_) go.push_back(kv.second);      // what gathered while the jobs ran leaves now
        pending_.clear();
        running_ += (int)go.size();
    }
    for (auto &g : go) flush(g, WhenIdle);
}

void Batcher::stats(ipx_batcher_stats *out)
{
    std::lock_guard<std::mutex> lk(mu_);
    *out = stats_;
    out->pending_files = 0;
    for (auto &kv : pending_) out->pending_files += (long long)kv.second->files.size();
}

}  // namespace ipx

// ---- the C ABI over a pool ------------------------------------------------------------------------------------------------------
#ifndef IPX_BATCHER_NO_ABI
#include "ipx_internal.h"

struct ipx_batcher {
    ipx_pool *pool;
    ipx::Batcher *b;
};

extern "C" {

int ipx_batcher_create(ipx_pool *pool, const ipx_batcher_config *cfg, ipx_batcher **out) try
{
    ipx::clear_error();
    if (!pool || !out) { ipx::set_error("ipx_batcher_create: bad argument"); return IPX_ERR_INVALID; }
    ipx::BatchBackend be;
    be.self = pool;
    be.submit = [](void *p, const ipx_job *j, ipx_ticket *t) { return ipx_job_submit((ipx_pool *)p, j, t); };
    be.wait = [](void *p, ipx_ticket t) { return ipx_job_wait((ipx_pool *)p, t, nullptr); };
    be.release = [](void *p, ipx_ticket t) { return ipx_job_release((ipx_pool *)p, t); };
    be.last_error = [] { return ipx_last_error(); };
    ipx_batcher *b = new ipx_batcher;
    b->pool = pool;
    b->b = new ipx::Batcher(be, cfg && cfg->max_batch > 0 ? cfg->max_batch : 256, cfg && cfg->max_wait_us > 0 ? cfg->max_wait_us : 2000,
                            cfg && cfg->quality > 0 ? cfg->quality : 85,
                            4 * std::max(1, ipx_pool_slots(pool)));   // (the pool's default: four feeders per device)
    *out = b;
    return IPX_OK;
}
IPX_CATCH_STATUS

void ipx_batcher_destroy(ipx_batcher *b)
{
    if (!b) return;
    try { delete b->b; } catch (...) { }
    delete b;
}

int ipx_batcher_submit(ipx_batcher *b, const ipx_bytes *file, const ipx_pool_ops *ops, ipx_batch_ticket *ticket) try
{
    ipx::clear_error();
    if (!b || !file || !ops || !ticket) { ipx::set_error("ipx_batcher_submit: bad argument"); return IPX_ERR_INVALID; }
    std::string err;
    uint64_t t = 0;
    const int rc = b->b->submit(*file, *ops, &t, &err);
    if (rc) ipx::set_error("%s", err.c_str());
    *ticket = t;
    return rc;
}
IPX_CATCH_STATUS

int ipx_batcher_wait(ipx_batcher *b, ipx_batch_ticket ticket, ipx_batch_result *res) try
{
    ipx::clear_error();
    if (!b) { ipx::set_error("ipx_batcher_wait: null batcher"); return IPX_ERR_INVALID; }
    std::string err;
    const int rc = b->b->wait(ticket, res, &err);
    if (rc) ipx::set_error("%s", err.c_str());
    return rc;
}
IPX_CATCH_STATUS

int ipx_batcher_release(ipx_batcher *b, ipx_batch_ticket ticket) try
{
    ipx::clear_error();
    if (!b) { ipx::set_error("ipx_batcher_release: null batcher"); return IPX_ERR_INVALID; }
    std::string err;
    const int rc = b->b->release(ticket, &err);
    if (rc && !err.empty()) ipx::set_error("%s", err.c_str());
    return rc;
}
IPX_CATCH_STATUS

int ipx_batcher_get_stats(ipx_batcher *b, ipx_batcher_stats *out) try
{
    ipx::clear_error();
    if (!b || !out) { ipx::set_error("ipx_batcher_get_stats: bad argument"); return IPX_ERR_INVALID; }
    b->b->stats(out);
    return IPX_OK;
}
IPX_CATCH_STATUS

}  // extern "C"
#endif

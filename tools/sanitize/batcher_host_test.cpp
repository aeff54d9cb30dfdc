// batcher_host_test.cpp -- the batcher's queue / timer / ticket logic (csrc/ipx_batcher.cpp) and the process-wide thread pool
// (csrc/ipx_threads.h) on the CPU alone, built with -fsanitize=thread (tools/sanitize/run_tsan.sh).  The "device" is a fake backend:
// a job's work is done by whoever waits for it (a memcpy of every file into "outputs"), with a small random delay, one file per
// batch flagged as not decodable, and one batch in nine failing at submit.
//
// What it checks: every submitted file gets exactly its own bytes back under its own ticket; files never share a batch with another
// key; a batch never exceeds max_batch; the timer flushes partial batches; a failing job reports to every file of that batch only; out
// of order waits and releases, release without wait, destroy with work pending and tickets uncollected; no data race (TSan).
#define IPX_BATCHER_NO_ABI 1
#include "../../imageprocessor_amd/csrc/ipx_batcher.cpp"
#include "../../imageprocessor_amd/csrc/ipx_threads.h"

#include <atomic>
#include <cassert>
#include <cstdio>
#include <deque>
#include <random>
#include <set>

namespace {

struct FakeJob {
    ipx_job job;                       // shallow: the batcher keeps the arrays alive
    std::vector<std::vector<uint8_t>> out;
    bool done = false, released = false;
    std::mutex mu;
};
struct FakePool {
    std::mutex mu;
    std::map<ipx_ticket, std::shared_ptr<FakeJob>> jobs;
    ipx_ticket next = 1;
    std::atomic<long long> submitted{0}, files{0}, largest{0}, released{0};
    int max_batch = 0;
};
thread_local std::string g_err;

int fake_submit(void *self, const ipx_job *j, ipx_ticket *t)
{
    FakePool *p = (FakePool *)self;
    assert(j->kind == IPX_JOB_JPEG && j->n >= 1 && j->n <= p->max_batch);
    for (int i = 1; i < j->n; i++) assert(j->files[i].data[0] == j->files[0].data[0]);   // byte 0 of a test file is its key: one key per batch
    assert(j->ops.sw == 100 + j->files[0].data[0]);
    if (j->files[0].data[0] == 0xff)                      // files with a JPEG frame header: one shape (components, luma sampling) per batch
        for (int i = 1; i < j->n; i++) assert(j->files[i].data[11] == j->files[0].data[11] && j->files[i].data[13] == j->files[0].data[13]);
    std::lock_guard<std::mutex> lk(p->mu);
    const ipx_ticket id = p->next++;
    if (id % 9 == 0) { g_err = "fake: submit refused"; return IPX_ERR_NOMEM; }
    auto fj = std::make_shared<FakeJob>();
    fj->job = *j;
    p->jobs[id] = fj;
    p->submitted++; p->files += j->n;
    long long l = p->largest.load();
    while (j->n > l && !p->largest.compare_exchange_weak(l, j->n)) { }
    *t = id;
    return IPX_OK;
}
int fake_wait(void *self, ipx_ticket t)
{
    FakePool *p = (FakePool *)self;
    std::shared_ptr<FakeJob> fj;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        auto it = p->jobs.find(t);
        assert(it != p->jobs.end());
        fj = it->second;
    }
    std::lock_guard<std::mutex> lk(fj->mu);            // the first waiter "runs" the job
    if (!fj->done) {
        std::this_thread::sleep_for(std::chrono::microseconds(200 + (t * 37) % 700));
        fj->out.resize((size_t)fj->job.n);
        for (int i = 0; i < fj->job.n; i++) {
            const ipx_bytes &f = fj->job.files[i];
            if (f.data[1] == 0xee) { fj->job.status[i] = IPX_ERR_UNSUPPORTED; continue; }   // "progressive CMYK": not ours
            fj->out[i].assign(f.data, f.data + f.len);
            fj->job.status[i] = IPX_OK;
            if (fj->job.resize_jpeg) fj->job.resize_jpeg[i] = ipx_bytes{fj->out[i].data(), fj->out[i].size()};
            if (fj->job.wm_jpeg) fj->job.wm_jpeg[i] = ipx_bytes{fj->out[i].data(), 1};
        }
        fj->done = true;
    }
    return IPX_OK;
}
int fake_release(void *self, ipx_ticket t)
{
    FakePool *p = (FakePool *)self;
    std::lock_guard<std::mutex> lk(p->mu);
    auto it = p->jobs.find(t);
    assert(it != p->jobs.end() && !it->second->released);
    it->second->released = true;
    p->jobs.erase(it);
    p->released++;
    return IPX_OK;
}

}  // namespace

int main()
{
    using namespace ipx;
    // ---- the thread pool: nested loops from several callers at once, every item exactly once ----
    {
        std::atomic<long long> sum{0};
        std::vector<std::thread> callers;
        for (int c = 0; c < 4; c++)
            callers.emplace_back([&, c] {
                for (int rep = 0; rep < 20; rep++)
                    HostPool::instance().parallel_for(37 + c, 8, [&](int i) {
                        HostPool::instance().parallel_for(5, 3, [&](int k) { sum += (long long)i * 5 + k; });      // a loop inside a loop (a part's preparation)
                    });
            });
        for (auto &t : callers) t.join();
        long long want = 0;
        for (int c = 0; c < 4; c++)
            for (int i = 0; i < 37 + c; i++)
                for (int k = 0; k < 5; k++) want += 20LL * (i * 5 + k);
        if (sum != want) { fprintf(stderr, "thread pool: %lld != %lld\n", (long long)sum, want); return 1; }
        printf("thread pool ok: %d threads, usable cpus %d\n", HostPool::instance().size(), usable_cpus());
    }
    std::unique_ptr<FakePool> pool_owner(new FakePool);
    FakePool &pool = *pool_owner;
    pool.max_batch = 16;
    BatchBackend be;
    be.self = &pool; be.submit = fake_submit; be.wait = fake_wait; be.release = fake_release;
    be.last_error = [] { return g_err.c_str(); };
    const int nthreads = 8, per_thread = 400;
    std::atomic<long long> ok_files{0}, refused{0}, unsupported{0};
    std::vector<std::deque<std::vector<uint8_t>>> all_files(nthreads);   // outlive the batcher: some tickets are never collected
    {
        std::unique_ptr<Batcher> b_owner(new Batcher(be, pool.max_batch, 1500, 85, 3));   // (groups leave at once while fewer than three jobs run)
        Batcher &b = *b_owner;
        std::vector<std::thread> ts;
        for (int th = 0; th < nthreads; th++)
            ts.emplace_back([&, th] {
                std::mt19937 rng(1234 + th);
                std::deque<std::vector<uint8_t>> &files = all_files[th];  // a file's bytes stay where they are until the batcher is gone
                std::vector<std::pair<uint64_t, size_t>> tickets;       // (ticket, index of its file)
                uint8_t mask[6] = {1, 2, 3, 4, 5, 6};
                for (int i = 0; i < per_thread; i++) {
                    const uint8_t key = (uint8_t)(rng() % 3);
                    std::vector<uint8_t> f(8 + rng() % 40);
                    for (auto &v : f) v = (uint8_t)rng();
                    f[0] = key; f[1] = rng() % 23 == 0 ? 0xee : 0x11; f[2] = (uint8_t)th; f[3] = (uint8_t)i; f[4] = (uint8_t)(i >> 8);
                    files.push_back(f);
                    ipx_pool_ops ops;
                    memset(&ops, 0, sizeof ops);
                    ops.sw = 100 + key; ops.sh = 50; ops.do_resize = 1; ops.resize_w = 10; ops.resize_h = 10; ops.do_watermark = 1;
                    ipx_glyph g;
                    memset(&g, 0, sizeof g);
                    g.mask = mask; g.mw = 3; g.mh = 2; g.mstride = 3; g.dr = ipx_rect{1, 1, 4, 3};
                    ops.glyphs = &g; ops.n_glyphs = 1;
                    ipx_bytes fb{files.back().data(), files.back().size()};
                    uint64_t t = 0;
                    std::string err;
                    const int rc = b.submit(fb, ops, &t, &err);
                    assert(rc == IPX_OK);
                    tickets.push_back({t, files.size() - 1});
                    if (rng() % 5 == 0) std::this_thread::sleep_for(std::chrono::microseconds(rng() % 900));
                    // collect in a scrambled order, sometimes late, sometimes never (the destructor has to cope)
                    while (tickets.size() > 24 || (i == per_thread - 1 && tickets.size() > 7)) {
                        const size_t pick = rng() % tickets.size();
                        const uint64_t tk = tickets[pick].first;
                        const std::vector<uint8_t> &src = files[tickets[pick].second];
                        ipx_batch_result res;
                        memset(&res, 0, sizeof res);
                        std::string e2;
                        const bool skip_wait = rng() % 11 == 0;
                        if (!skip_wait) {
                            const int r2 = b.wait(tk, &res, &e2);
                            if (r2 == IPX_ERR_NOMEM) { assert(e2 == "fake: submit refused"); refused++; }
                            else {
                                assert(r2 == IPX_OK);
                                if (src[1] == 0xee) { assert(res.status == IPX_ERR_UNSUPPORTED); unsupported++; }
                                else {
                                    assert(res.status == IPX_OK && res.resize.len == src.size() && !memcmp(res.resize.data, src.data(), src.size()));
                                    assert(res.thumb.data == nullptr && res.wm.len == 1 && res.wm.data[0] == src[0]);
                                    ok_files++;
                                }
                            }
                        }
                        const int r3 = b.release(tk, &e2);
                        assert(r3 == IPX_OK);
                        tickets.erase(tickets.begin() + (long)pick);
                    }
                }
            });
        for (auto &t : ts) t.join();
        ipx_batcher_stats st;
        b.stats(&st);
        printf("batcher: %lld files in %lld batches (%lld by size, %lld by timer, %lld when idle), largest %lld; verified %lld, refused %lld, unsupported %lld\n", st.files, st.batches,
               st.flushed_by_size, st.flushed_by_timer, st.flushed_when_idle, st.largest_batch, (long long)ok_files, (long long)refused, (long long)unsupported);
        if (st.files != (long long)nthreads * per_thread || st.largest_batch > pool.max_batch || st.flushed_by_timer + st.flushed_when_idle == 0 || st.flushed_by_size == 0 ||
            st.batches != st.flushed_by_size + st.flushed_by_timer + st.flushed_when_idle) return 2;
    }   // ~Batcher: pending files flushed, uncollected tickets' jobs waited for and released
    {
        std::lock_guard<std::mutex> lk(pool.mu);
        if (!pool.jobs.empty()) { fprintf(stderr, "%zu jobs were never released\n", pool.jobs.size()); return 3; }
    }
    if (ok_files < 1000 || refused == 0 || unsupported == 0) return 4;
    {
        // the same operators, three JPEG shapes: three batches (fake_submit asserts one shape per job)
        FakePool pool2;
        pool2.max_batch = 16;
        BatchBackend be2 = be;
        be2.self = &pool2;
        std::vector<std::vector<uint8_t>> fs;
        const uint8_t shapes[3][2] = {{1, 0x11}, {3, 0x22}, {3, 0x11}};
        for (int i = 0; i < 9; i++) fs.push_back({0xff, 0xd8, 0xff, 0xc0, 0x00, 0x0b, 0x08, 0x00, 0x32, 0x01, 0x63, shapes[i % 3][0], 0x01, shapes[i % 3][1], 0x00, (uint8_t)i});
        {
            Batcher b2(be2, pool2.max_batch, 50000, 85, 0);
            ipx_pool_ops ops;
            memset(&ops, 0, sizeof ops);
            ops.sw = 100 + 0xff; ops.sh = 50; ops.do_resize = 1; ops.resize_w = 10; ops.resize_h = 10;
            for (auto &f : fs) {
                uint64_t t = 0;
                std::string err;
                const int rc = b2.submit(ipx_bytes{f.data(), f.size()}, ops, &t, &err);
                assert(rc == IPX_OK);
            }
        }   // ~Batcher flushes the three groups
        if (pool2.submitted != 3 || pool2.files != 9 || pool2.largest != 3) { fprintf(stderr, "shapes: %lld jobs, %lld files\n", (long long)pool2.submitted, (long long)pool2.files); return 5; }
    }
    printf("batcher ok: %lld jobs submitted, %lld released\n", (long long)pool.submitted, (long long)pool.released);
    return 0;
}

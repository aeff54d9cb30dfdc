// ipx_batcher.h -- micro-batching of single uploads below the C ABI (ipx_batcher_* of include/ipx.h).
//
// The reference pulls ONE message per goroutine (internal/worker/worker.go:112-149) from a channel of concurrency * 2 (:88); a GPU
// wants hundreds of files per launch.  The batcher stands between: every goroutine hands over its one file and waits for its ticket;
// files are grouped by frame size, JPEG shape (components, luma sampling: a batch is one shape) and operator content (parameters,
// colour, every glyph's rectangle and mask bytes), a group goes to
// the pool as one JPEG job when it holds max_batch files or when its first file has waited max_wait_us, and every file gets its own
// status -- a file the GPU path cannot decode does not fail its neighbours (the worker runs its own image.Decode path for it).
//
// All of it is host logic over three calls of a backend (submit / wait / release of a job), so it builds and runs without a GPU: the
// product binds the backend to ipx_job_* of a pool, tests/batcher_host_test.cpp binds it to a fake and runs under ThreadSanitizer.
#pragma once

#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ipx.h"

namespace ipx {

struct BatchBackend {
    void *self = nullptr;
    int (*submit)(void *self, const ipx_job *job, ipx_ticket *ticket) = nullptr;
    int (*wait)(void *self, ipx_ticket ticket) = nullptr;       // may be called by several threads for one ticket
    int (*release)(void *self, ipx_ticket ticket) = nullptr;
    const char *(*last_error)() = nullptr;
};

class Batcher {
public:
    // idle_jobs: a group leaves at once while fewer jobs than this are running (what the backend can run side by side: the pool's
    // feeders); 0 = size and timer only
    Batcher(const BatchBackend &be, int max_batch, int max_wait_us, int quality, int idle_jobs = 1);
    ~Batcher();                                     // flushes what is pending, waits for every job, releases them
    int submit(const ipx_bytes &file, const ipx_pool_ops &ops, uint64_t *ticket, std::string *err);
    int wait(uint64_t ticket, ipx_batch_result *res, std::string *err);
    int release(uint64_t ticket, std::string *err);
    void stats(ipx_batcher_stats *out);

private:
    struct OpsCopy {                                // a deep copy of ipx_pool_ops (the submitter may free its own at once)
        ipx_pool_ops p{};
        std::vector<ipx_glyph> glyphs;
        std::vector<std::vector<uint8_t>> masks;
    };
    struct Batch {
        std::string key;
        OpsCopy ops;
        std::chrono::steady_clock::time_point deadline;
        std::vector<ipx_bytes> files;
        std::vector<ipx_bytes> res, th, wm;
        std::vector<int32_t> status;
        // guarded by Batcher::mu_
        bool flushed = false, submit_failed = false, job_released = false, done = false;   // done: some waiter has seen the job finish
        int rc = IPX_OK;
        std::string error;
        ipx_ticket job = 0;
        int unreleased = 0;
        std::condition_variable cv;                 // flushed
    };
    static int copy_ops(const ipx_pool_ops &in, OpsCopy *out, std::string *key, std::string *err);
    enum Why { BySize, ByTimer, WhenIdle };
    void flush(const std::shared_ptr<Batch> &b, Why why);
    void job_seen_done(const std::shared_ptr<Batch> &b);   // the first waiter back from the backend: perhaps nothing runs any more
    void timer_loop();

    BatchBackend be_;
    int max_batch_, quality_;
    std::chrono::microseconds max_wait_;
    std::mutex mu_;
    std::condition_variable cv_timer_;
    std::map<std::string, std::shared_ptr<Batch>> pending_;
    std::map<uint64_t, std::pair<std::shared_ptr<Batch>, int>> tickets_;
    uint64_t next_ticket_ = 1;
    bool stop_ = false;
    int idle_jobs_ = 1;                             // a group leaves at once while fewer jobs than this run (IPX_BATCHER_IDLE_FLUSH: 0 = size and timer only)
    int running_ = 0;                               // jobs handed to the backend that nobody has seen finish yet
    ipx_batcher_stats stats_{};
    std::thread timer_;
};

}  // namespace ipx

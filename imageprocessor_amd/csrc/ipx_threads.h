// ipx_threads.h -- the library's host worker threads: one persistent pool per process, sized from the CPUs this process may actually
// use (its affinity mask and its cgroup quota), not from the machine's core count.
//
// Why: the codec entries prepare their batches on the host (marker parsing, RSTn search, scan packing, the scans of progressive
// files).  They used to start up to 16 std::threads per call and per part -- four parts of a compressed-in / compressed-out batch meant
// up to 64 threads on a worker that owns 16 CPUs of a 256-CPU box (std::thread::hardware_concurrency() reports the box) -- and paid
// thread creation on every call, which showed in the 5 ms floor of small batches.
#pragma once

#include <sched.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace ipx {

// CPUs this process may run on: the affinity mask, cut by the cgroup's CPU quota (cpu.max of cgroup v2, cfs_quota of v1) when one is set
inline int usable_cpus()
{
    int n = 0;
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
    if (n <= 0) n = (int)std::thread::hardware_concurrency();
    if (n <= 0) n = 1;
    auto quota = [](const char *path, bool v2) -> double {
        FILE *f = fopen(path, "r");
        if (!f) return 0;
        char a[64] = "", b[64] = "";
        double q = 0;
        if (v2) {
            if (fscanf(f, "%63s %63s", a, b) == 2 && a[0] != 'm') { const double per = atof(b); if (per > 0) q = atof(a) / per; }
        } else if (fscanf(f, "%63s", a) == 1) {
            const double us = atof(a);
            if (us > 0) {
                FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
                double per = 100000;
                if (g) { if (fscanf(g, "%63s", b) == 1 && atof(b) > 0) per = atof(b); fclose(g); }
                q = us / per;
            }
        }
        fclose(f);
        return q;
    };
    double q = quota("/sys/fs/cgroup/cpu.max", true);
    if (q <= 0) q = quota("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", false);
    if (q > 0) n = std::max(1, std::min(n, (int)(q + 0.5)));
    return n;
}

class HostPool {
public:
    // IPX_HOST_THREADS overrides the size (the callers' own thread counts as one worker: size - 1 threads are started)
    static HostPool &instance()
    {
        static HostPool p;
        return p;
    }
    int size() const { return nworkers_ + 1; }

    // fn(i) for i in [0, count), on at most `max_threads` threads (the caller's included), items handed out one at a time.  Returns when
    // every item is done.  fn must not throw (the callers wrap their bodies in guarded_status).  Several callers may loop at once.
    void parallel_for(int count, int max_threads, const std::function<void(int)> &fn)
    {
        if (count <= 0) return;
        const int helpers = std::min({max_threads - 1, nworkers_, count - 1});
        if (helpers <= 0) { for (int i = 0; i < count; i++) fn(i); return; }
        auto job = std::make_shared<Job>();
        job->fn = &fn; job->count = count; job->slots = helpers;
        {
            std::lock_guard<std::mutex> lk(mu_);
            jobs_.push_back(job);
        }
        cv_.notify_all();
        run(*job);                                     // the caller works too
        std::unique_lock<std::mutex> lk(mu_);
        job->slots = 0;                                // no worker may join from here on
        done_.wait(lk, [&] { return job->active == 0; });
        for (auto it = jobs_.begin(); it != jobs_.end(); ++it)
            if (it->get() == job.get()) { jobs_.erase(it); break; }
    }

private:
    struct Job {
        const std::function<void(int)> *fn = nullptr;
        int count = 0;
        std::atomic<int> next{0};
        int slots = 0;      // workers that may still join (guarded by mu_)
        int active = 0;     // workers inside run() (guarded by mu_)
    };
    static void run(Job &j)
    {
        for (int i = j.next.fetch_add(1); i < j.count; i = j.next.fetch_add(1)) (*j.fn)(i);
    }
    HostPool()
    {
        const char *v = getenv("IPX_HOST_THREADS");
        int n = v && atoi(v) > 0 ? atoi(v) : std::min(usable_cpus(), 16);
        nworkers_ = std::max(0, n - 1);
        for (int t = 0; t < nworkers_; t++) threads_.emplace_back([this] { worker(); });
    }
    ~HostPool()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : threads_) t.join();
    }
    void worker()
    {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            std::shared_ptr<Job> job;
            cv_.wait(lk, [&] {
                if (stop_) return true;
                for (auto &j : jobs_)
                    if (j->slots > 0 && j->next.load() < j->count) { job = j; return true; }
                return false;
            });
            if (stop_) return;
            job->slots--;
            job->active++;
            lk.unlock();
            run(*job);
            lk.lock();
            job->active--;
            if (job->active == 0) done_.notify_all();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_, done_;
    std::deque<std::shared_ptr<Job>> jobs_;
    std::vector<std::thread> threads_;
    int nworkers_ = 0;
    bool stop_ = false;
};

}  // namespace ipx

// ipx_pool_core.h -- the queue, the tickets and the feeder threads of ipx_pool (ipx_pool.hip), with no GPU in them.
//
// A job is cut into chunks; chunks wait in ONE priority queue (the most expensive first, submission order among equals) and every
// feeder thread takes the top one whenever it is free -- pull scheduling: a mixed load balances itself over slots of different speed.
// What a chunk DOES is a callable the feeder made for itself on its own thread (ipx_pool.hip: upload, kernels, download on the slot's
// GPU; tools/sanitize/pool_host_test.cpp: a memcpy), so this file compiles without HIP and runs under ThreadSanitizer on the CPU.
// Mirrors worker.go:88-96, 112-149: WORKER_CONCURRENCY goroutines pulling independent messages, nothing exchanged between them.
#pragma once

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <queue>
#include <string>
#include <thread>
#include <vector>

namespace ipx {

template <class Job>
class PoolCore {
public:
    struct State {
        Job user;                       // what the chunks work on; a chunk function may touch it under State::mu only where chunks share it
        std::mutex mu;                  // for the chunk functions (results several chunks append to); the core never takes it
        uint64_t id = 0;
        int chunks_left = 0, status = 0, units_done = 0;
        std::string error;
    };
    struct Piece { int i0, m; double cost; };
    // one chunk: units [i0, i0 + m) of the job; returns 0 or a status and sets *error
    using ChunkFn = std::function<int(State &, int i0, int m, std::string *error)>;
    // called once per feeder ON its thread; what the returned function captures is destroyed on that thread when the pool stops
    using FeederFactory = std::function<ChunkFn(int slot)>;

    PoolCore() { for (auto &c : units_by_slot_) c = 0; }
    ~PoolCore() { stop(); }
    PoolCore(const PoolCore &) = delete;
    PoolCore &operator=(const PoolCore &) = delete;

    void start(int slots, int feeders_per_slot, FeederFactory factory)
    {
        nslots_ = slots;
        for (int s = 0; s < slots; s++)
            for (int l = 0; l < feeders_per_slot; l++) threads_.emplace_back([this, s, factory] { feeder(s, factory); });
    }
    int feeders() const { return (int)threads_.size(); }

    // false: the pool is stopping (nothing was queued)
    bool submit(const std::shared_ptr<State> &j, const std::vector<Piece> &pieces, uint64_t *ticket)
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (stopping_) return false;
            j->id = next_id_++;
            j->chunks_left = (int)pieces.size();
            jobs_[j->id] = j;
            for (const Piece &p : pieces) queue_.push(Chunk{j, p.i0, p.m, p.cost, next_seq_++});
            *ticket = j->id;
        }
        cv_work_.notify_all();
        if (pieces.empty()) cv_done_.notify_all();
        return true;
    }
    // nullptr: unknown ticket.  Blocks until every chunk of the job has run.
    std::shared_ptr<State> wait(uint64_t ticket)
    {
        std::unique_lock<std::mutex> lk(mu_);
        auto it = jobs_.find(ticket);
        if (it == jobs_.end()) return nullptr;
        std::shared_ptr<State> j = it->second;
        cv_done_.wait(lk, [&] { return j->chunks_left == 0; });
        return j;
    }
    // -1: unknown ticket
    int poll(uint64_t ticket)
    {
        std::lock_guard<std::mutex> lk(mu_);
        auto it = jobs_.find(ticket);
        return it == jobs_.end() ? -1 : it->second->chunks_left == 0;
    }
    // waits for the job if it still runs, then forgets the ticket; the caller frees what the job holds
    std::shared_ptr<State> release(uint64_t ticket)
    {
        std::unique_lock<std::mutex> lk(mu_);
        auto it = jobs_.find(ticket);
        if (it == jobs_.end()) return nullptr;
        std::shared_ptr<State> j = it->second;
        cv_done_.wait(lk, [&] { return j->chunks_left == 0; });
        jobs_.erase(ticket);
        return j;
    }
    // no new jobs; the feeders drain what is queued and leave.  Afterwards every job still known is finished.
    void stop()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stopping_ = true;
        }
        cv_work_.notify_all();
        for (auto &t : threads_) if (t.joinable()) t.join();
        threads_.clear();
    }
    // jobs nobody released (call after stop())
    std::vector<std::shared_ptr<State>> leftovers()
    {
        std::lock_guard<std::mutex> lk(mu_);
        std::vector<std::shared_ptr<State>> v;
        for (auto &kv : jobs_) v.push_back(kv.second);
        jobs_.clear();
        return v;
    }
    long long units_done(int slot) const { return slot >= 0 && slot < nslots_ && slot < 64 ? (long long)units_by_slot_[slot] : -1; }

private:
    struct Chunk { std::shared_ptr<State> job; int i0, m; double cost; uint64_t seq; };
    struct Less { bool operator()(const Chunk &a, const Chunk &b) const { return a.cost != b.cost ? a.cost < b.cost : a.seq > b.seq; } };

    void feeder(int slot, const FeederFactory &factory)
    {
        ChunkFn run = factory(slot);
        for (;;) {
            Chunk c;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_work_.wait(lk, [&] { return stopping_ || !queue_.empty(); });
                if (queue_.empty()) break;          // stopping, and nothing left to drain
                c = queue_.top();
                queue_.pop();
            }
            std::string text;
            int rc;
            try { rc = run ? run(*c.job, c.i0, c.m, &text) : -1; }
            catch (const std::exception &e) { rc = -1; text = e.what(); }
            catch (...) { rc = -1; text = "unknown exception in a pool chunk"; }
            bool finished;
            {
                std::lock_guard<std::mutex> lk(mu_);
                State &j = *c.job;
                if (rc && j.status == 0) { j.status = rc; j.error = text; }
                if (!rc) j.units_done += c.m;
                if (slot < 64) units_by_slot_[slot] += c.m;
                finished = --j.chunks_left == 0;
            }
            if (finished) cv_done_.notify_all();
        }
    }

    std::mutex mu_;
    std::condition_variable cv_work_, cv_done_;
    std::priority_queue<Chunk, std::vector<Chunk>, Less> queue_;
    std::map<uint64_t, std::shared_ptr<State>> jobs_;
    std::vector<std::thread> threads_;
    uint64_t next_id_ = 1, next_seq_ = 1;
    bool stopping_ = false;
    int nslots_ = 0;
    std::atomic<long long> units_by_slot_[64];
};

}  // namespace ipx

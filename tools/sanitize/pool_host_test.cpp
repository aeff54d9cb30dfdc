// pool_host_test.cpp -- the pool's queue / ticket / feeder logic (csrc/ipx_pool_core.h, what ipx_pool.hip runs its GPU work on) on the
// CPU alone, built with -fsanitize=thread (tools/sanitize/run_tsan.sh).  The "device" work of a chunk is a memcpy of the chunk's units
// from the job's source to its destination, with a small delay that depends on the slot (slots of different speed), and jobs marked
// to fail report a status from one of their chunks.
//
// What it checks: 8 submitters with jobs of mixed size; waits out of order, polls, releases without a wait and releases of running
// jobs; every unit of every job copied exactly once (no chunk lost, none run twice), units_done and the per-slot counters add up; a
// failing job reports its own status and text and leaves its neighbours alone; the most expensive chunk first when the queue holds
// several; stop() with work queued drains it; submit after stop() is refused; tickets nobody collected are handed back by
// leftovers(); no data race (TSan).
#include "../../imageprocessor_amd/csrc/ipx_pool_core.h"

#include <cassert>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <random>

namespace {

struct TestJob {
    std::vector<uint8_t> src, dst;
    std::vector<std::atomic<int>> *touched = nullptr;   // per unit: how many chunks copied it
    int unit = 64;                                       // bytes per unit
    int fail_at = -1;                                    // the chunk that holds this unit fails
    std::vector<int> order;                              // costs in the order the chunks ran (single-feeder test), under State::mu
};
typedef ipx::PoolCore<TestJob> Core;

std::atomic<long long> g_chunks{0};

Core::ChunkFn make_feeder(int slot)
{
    auto scratch = std::make_shared<std::vector<uint8_t>>(1 << 16);     // a feeder's own buffer, freed on its thread
    return [slot, scratch](Core::State &st, int i0, int m, std::string *error) -> int {
        TestJob &j = st.user;
        g_chunks++;
        std::this_thread::sleep_for(std::chrono::microseconds(20 + 30 * slot));
        if (j.fail_at >= i0 && j.fail_at < i0 + m) { *error = "fake: unit " + std::to_string(j.fail_at) + " is not decodable"; return 7; }
        for (int i = i0; i < i0 + m; i++) {
            memcpy(scratch->data(), j.src.data() + (size_t)i * j.unit, j.unit);
            memcpy(j.dst.data() + (size_t)i * j.unit, scratch->data(), j.unit);
            (*j.touched)[i]++;
        }
        { std::lock_guard<std::mutex> lk(st.mu); j.order.push_back(m); }
        return 0;
    };
}

std::shared_ptr<Core::State> make_job(int units, int seed, int fail_at, std::vector<std::atomic<int>> *touched)
{
    auto st = std::make_shared<Core::State>();
    TestJob &j = st->user;
    j.src.resize((size_t)units * j.unit);
    j.dst.assign(j.src.size(), 0);
    std::mt19937 rng(seed);
    for (auto &b : j.src) b = (uint8_t)rng();
    j.fail_at = fail_at;
    j.touched = touched;
    return st;
}

std::vector<Core::Piece> cut(int units, int per)
{
    std::vector<Core::Piece> v;
    for (int i0 = 0; i0 < units; i0 += per) v.push_back(Core::Piece{i0, std::min(per, units - i0), (double)std::min(per, units - i0)});
    return v;
}

void test_many_submitters()
{
    Core core;
    core.start(3, 2, make_feeder);
    assert(core.feeders() == 6);
    std::atomic<long long> units_ok{0}, failed_jobs{0};
    std::vector<std::thread> subs;
    for (int t = 0; t < 8; t++)
        subs.emplace_back([&, t] {
            std::mt19937 rng(100 + t);
            struct Held { uint64_t ticket; std::shared_ptr<Core::State> st; std::unique_ptr<std::vector<std::atomic<int>>> touched; int units; bool fails; };
            std::vector<Held> held;
            for (int k = 0; k < 40; k++) {
                const int units = 1 + (int)(rng() % 300), per = 1 + (int)(rng() % 40);
                const bool fails = rng() % 7 == 0;
                Held h;
                h.units = units; h.fails = fails;
                h.touched.reset(new std::vector<std::atomic<int>>(units));
                for (auto &c : *h.touched) c = 0;
                h.st = make_job(units, (int)rng(), fails ? (int)(rng() % units) : -1, h.touched.get());
                const bool ok = core.submit(h.st, cut(units, per), &h.ticket);
                assert(ok);
                held.push_back(std::move(h));
                // collect some earlier job, not the oldest: waits out of order, some polls, some releases without a wait
                if (held.size() > 3 || k == 39) {
                    while (!held.empty() && (held.size() > 3 || k == 39)) {
                        const size_t pick = rng() % held.size();
                        Held g = std::move(held[pick]);
                        held.erase(held.begin() + (long)pick);
                        const int mode = (int)(rng() % 3);
                        if (mode == 0) { while (core.poll(g.ticket) == 0) std::this_thread::yield(); }
                        std::shared_ptr<Core::State> st = mode == 2 ? core.release(g.ticket) : core.wait(g.ticket);
                        assert(st && st.get() == g.st.get() && st->chunks_left == 0);
                        if (g.fails) {
                            assert(st->status == 7 && st->error.find("not decodable") != std::string::npos);
                            failed_jobs++;
                        } else {
                            assert(st->status == 0 && st->units_done == g.units);
                            assert(st->user.dst == st->user.src);
                            for (auto &c : *g.touched) assert(c == 1);
                            units_ok += g.units;
                        }
                        if (mode != 2) { auto again = core.release(g.ticket); assert(again.get() == g.st.get()); }
                        assert(core.poll(g.ticket) == -1 && !core.wait(g.ticket) && !core.release(g.ticket));   // the ticket is gone
                    }
                }
            }
        });
    for (auto &t : subs) t.join();
    long long by_slot = 0;
    for (int s = 0; s < 3; s++) { assert(core.units_done(s) > 0); by_slot += core.units_done(s); }
    assert(core.units_done(3) == -1);
    assert(by_slot >= units_ok);                         // (chunks of failing jobs that ran before the failing one count too)
    core.stop();
    assert(core.leftovers().empty());
    printf("pool core ok: 320 jobs from 8 submitters, %lld units copied once each, %lld jobs failed as told, %lld chunks\n", (long long)units_ok, (long long)failed_jobs,
           (long long)g_chunks);
}

void test_order_and_shutdown()
{
    // one feeder: the queue's order is observable.  The first job keeps the feeder busy while three more queue up.
    Core core;
    core.start(1, 1, make_feeder);
    std::vector<std::unique_ptr<std::vector<std::atomic<int>>>> touched;
    auto fresh = [&](int units) { touched.emplace_back(new std::vector<std::atomic<int>>(units)); for (auto &c : *touched.back()) c = 0; return touched.back().get(); };
    auto blocker = make_job(2000, 1, -1, fresh(2000));
    uint64_t tb, t1, t2;
    bool ok = core.submit(blocker, cut(2000, 2000), &tb);
    assert(ok);
    auto mixed = make_job(60, 2, -1, fresh(60));
    std::vector<Core::Piece> pieces = {{0, 5, 5.0}, {5, 30, 30.0}, {35, 10, 10.0}, {45, 15, 15.0}};
    ok = core.submit(mixed, pieces, &t1);
    assert(ok);
    auto st = core.wait(t1);
    assert(st->status == 0 && st->user.dst == st->user.src);
    assert((st->user.order == std::vector<int>{30, 15, 10, 5}));      // the most expensive chunk first
    // an empty job is finished at once
    auto empty = make_job(1, 3, -1, fresh(1));
    ok = core.submit(empty, {}, &t2);
    assert(ok && core.poll(t2) == 1 && core.wait(t2)->units_done == 0);
    // stop with work queued: everything queued still runs; nobody released these tickets
    std::vector<std::shared_ptr<Core::State>> pending;
    for (int k = 0; k < 20; k++) {
        pending.push_back(make_job(50, 10 + k, -1, fresh(50)));
        uint64_t t;
        ok = core.submit(pending.back(), cut(50, 7), &t);
        assert(ok);
    }
    core.stop();
    for (auto &p : pending) { assert(p->chunks_left == 0 && p->units_done == 50 && p->user.dst == p->user.src); }
    uint64_t t;
    assert(!core.submit(make_job(4, 99, -1, fresh(4)), cut(4, 2), &t));   // refused after stop()
    assert(core.leftovers().size() == 20 + 3);                             // the blocker, the mixed and the empty job were never released either
    printf("pool core ok: priority order, empty job, shutdown with 20 jobs queued, submit after stop refused\n");
}

}  // namespace

int main()
{
    test_many_submitters();
    test_order_and_shutdown();
    return 0;
}

// host_concurrency.cpp — the library's host-side concurrency (otters_amd/csrc/ott_host.h, the SAME header libotters_hip.so is
// built from) against a MOCK device, for -fsanitize=thread and -fsanitize=address,undefined (tests/host/Makefile, run by
// tests/test_host_concurrency.py in the CPU suite).  The reference gets this from the borrow checker (`&self` queries,
// src/vec.rs:387; MetaStore !Sync, src/meta.rs:54; rayon's fan-out, src/meta.rs:678); here it is tested.
//
// The mock store is ott_store's locking skeleton with host memory for HBM: `rw` (queries shared, appends exclusive), a context
// pool, rows staged on the host, a background builder that keeps a derived "plane" (a running checksum) up to date, and — for
// the multi-store — shards behind a front lock with a dirty flag.  Every scenario is a randomised schedule (seeded); each
// checks invariants that a race would break (and the sanitizer watches the rest):
//   pool      concurrent run_all callers on one ShardPool: every fn(g) runs exactly once per call, on the right slot
//   lifetime  pools and background workers created, used 0..n times, destroyed while idle / right after the last call
//   contexts  more threads than contexts: a context is never held twice, workers are bounded, waiters get served
//   appends   single-row appends staged + large appends + queries + readers: a query sees no staged rows, a prefix-consistent
//             store (checksum of rows [0, n) matches), len() never shrinks; the background builder wakes, coalesces, stops
//   multi     a front store over shards: appends dirty the layout, queries clean it under the exclusive lock and fan out
// usage: host_concurrency <schedules> [seed [kind]]
#include <stdio.h>
#include <stdlib.h>

#include <random>

#include "../../otters_amd/csrc/ott_host.h"

using namespace ott::host;

static std::atomic<uint64_t> g_fail{0};
#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) {                                                               \
            g_fail.fetch_add(1);                                                     \
            fprintf(stderr, "CHECK failed at %s:%d: %s\n", __FILE__, __LINE__, #cond); \
        }                                                                            \
    } while (0)

// ---- the mock ------------------------------------------------------------------------------------------------------------------
constexpr uint32_t DIM = 4;
static float row_value(uint64_t row, uint32_t c) { return (float)((row * 7 + c) % 1021); }
static uint64_t row_sum(uint64_t row) {
    uint64_t s = 0;
    for (uint32_t c = 0; c < DIM; c++) s += (uint64_t)row_value(row, c);
    return s;
}

struct MockStore {
    RwGate rw;
    std::mutex mu;  // the store's own query context
    ContextPool<MockStore> pool;
    bool is_worker = false;
    MockStore* owner = nullptr;
    std::atomic<int> in_use{0};  // contexts: must be 0 when handed out
    // "HBM": rows [0, n); the buffer is REPLACED when it grows (like realloc_store), so a reader that races an append faults
    std::unique_ptr<float[]> rows;
    uint64_t n = 0, cap = 0;
    std::vector<float> pend_mem;
    StagedRows pend;
    // the derived plane: checksum of rows [0, plane_rows), kept up to date by the background worker (under plane_mu)
    std::mutex plane_mu;
    uint64_t plane_rows = 0, plane_sum = 0;
    std::unique_ptr<QuietWorker> builder;

    ~MockStore() {
        builder.reset();  // stops and joins before the rows go
        for (MockStore* w : pool.workers) delete w;
    }
    uint64_t len() const { return n + pend.count(); }  // (n is read under rw by callers that need it exact)

    int append_resident(const float* src, uint64_t cnt) {  // rw exclusive + mu held
        if (n + cnt > cap) {
            uint64_t ncap = cap ? cap * 2 : 64;
            while (ncap < n + cnt) ncap *= 2;
            std::unique_ptr<float[]> fresh(new float[ncap * DIM]);
            if (n) memcpy(fresh.get(), rows.get(), n * DIM * sizeof(float));
            rows = std::move(fresh);  // the old buffer is freed here: anyone still reading it is a use-after-free (ASan)
            cap = ncap;
            std::lock_guard<std::mutex> g(plane_mu);  // like realloc_store: the plane is dropped and rebuilt
            plane_rows = plane_sum = 0;
        }
        memcpy(rows.get() + n * DIM, src, cnt * DIM * sizeof(float));
        n += cnt;
        return 0;
    }
    int flush_locked() {
        return pend.flush([this](const float* b, uint64_t cnt) { return append_resident(b, cnt); });
    }
    int flush() {
        if (!pend.count()) return 0;
        ExclusiveLock wr(rw);
        std::lock_guard<std::mutex> g(mu);
        return flush_locked();
    }
    void kick() {
        if (!builder)
            builder.reset(new QuietWorker([this] { build_plane(); }, std::chrono::milliseconds(1)));
        builder->kick();
    }
    void append(uint64_t cnt, bool staged) {
        std::vector<float> src(cnt * DIM);
        ExclusiveLock wr(rw);
        std::lock_guard<std::mutex> g(mu);
        const uint64_t first = n + pend.count();
        for (uint64_t r = 0; r < cnt; r++)
            for (uint32_t c = 0; c < DIM; c++) src[r * DIM + c] = row_value(first + r, c);
        if (staged) {
            if (!pend.fits(cnt, DIM)) CHECK(flush_locked() == 0);
            if (!pend.buf) {
                pend_mem.resize(64 * DIM);
                pend.buf = pend_mem.data();
                pend.cap_bytes = pend_mem.size() * sizeof(float);
            }
            if (pend.fits(cnt, DIM)) {
                pend.stage(src.data(), cnt, DIM);
                return;
            }
        }
        CHECK(flush_locked() == 0);
        CHECK(append_resident(src.data(), cnt) == 0);
        kick();
    }
    MockStore* ctx_acquire() {
        MockStore* c = pool.acquire(
            this, 3,
            [this]() -> MockStore* {
                MockStore* w = new MockStore();
                w->is_worker = true;
                w->owner = this;
                return w;
            },
            [this](MockStore* w) { w->n = n; });  // alias_corpus: the caller holds rw shared
        CHECK(c->in_use.fetch_add(1) == 0);       // never handed out twice
        return c;
    }
    void ctx_release(MockStore* c) {
        CHECK(c->in_use.fetch_sub(1) == 1);
        pool.release(c);
    }
    void build_plane() {  // the background worker's run: takes the store like a query
        SharedLock rd(rw);
        MockStore* c = ctx_acquire();
        {
            std::lock_guard<std::mutex> g(plane_mu);
            for (uint64_t r = plane_rows; r < n; r++) {
                uint64_t s = 0;
                for (uint32_t col = 0; col < DIM; col++) s += (uint64_t)rows[r * DIM + col];
                plane_sum += s;
            }
            plane_rows = n;
        }
        ctx_release(c);
    }
    // a query: nothing staged, rows [0, n) intact, the plane (if current) agrees
    void query(std::mt19937_64& rng) {
        SharedLock rd;
        CHECK(lock_shared_clean(rw, rd, [this] { return pend.count() != 0; }, [this] { return flush(); }) == 0);
        CHECK(pend.count() == 0);
        MockStore* c = ctx_acquire();
        const uint64_t cnt = n;
        uint64_t want = 0, got = 0;
        const uint64_t lo = cnt > 256 ? rng() % (cnt - 256) : 0, hi = cnt > 256 ? lo + 256 : cnt;  // a window: keeps a schedule short
        for (uint64_t r = lo; r < hi; r++) {
            want += row_sum(r);
            for (uint32_t col = 0; col < DIM; col++) got += (uint64_t)rows[r * DIM + col];
        }
        CHECK(want == got);
        {
            std::lock_guard<std::mutex> g(plane_mu);
            if (plane_rows == cnt && cnt <= 4096) {
                uint64_t all = 0;
                for (uint64_t r = 0; r < cnt; r++) all += row_sum(r);
                CHECK(all == plane_sum);
            }
            CHECK(plane_rows <= cnt);
        }
        ctx_release(c);
    }
};

// ---- scenarios -----------------------------------------------------------------------------------------------------------------
static void scenario_pool(std::mt19937_64& rng) {
    const size_t G = 1 + rng() % 8, callers = 1 + rng() % 4;
    const int calls = 1 + (int)(rng() % 6);
    ShardPool pool(G);
    std::vector<std::thread> th;
    for (size_t t = 0; t < callers; t++)
        th.emplace_back([&, t] {
            std::mt19937_64 r2(t * 977 + G);
            for (int i = 0; i < calls; i++) {
                std::vector<int> hit(G, 0);
                const int spin = (int)(r2() % 200);
                pool.run_all([&](size_t g) {
                    for (volatile int s = 0; s < spin; s++) {}
                    hit[g]++;  // each slot written by exactly one thread of this call
                });
                for (size_t g = 0; g < G; g++) CHECK(hit[g] == 1);
            }
        });
    for (auto& t : th) t.join();
    // destroyed here: idle, or right after the last call returned
}

static void scenario_lifetime(std::mt19937_64& rng) {
    {
        ShardPool idle(1 + rng() % 8);  // never used
    }
    std::atomic<int> runs{0};
    {
        QuietWorker w([&] { runs.fetch_add(1); }, std::chrono::milliseconds(rng() % 2));
        const int kicks = (int)(rng() % 4);
        for (int i = 0; i < kicks; i++) w.kick();
        if (rng() & 1) std::this_thread::sleep_for(std::chrono::microseconds(rng() % 3000));
        // destroyed idle, mid-quiet-period or mid-run
    }
    CHECK(runs.load() <= 4);
    {
        QuietWorker w([&] { std::this_thread::sleep_for(std::chrono::microseconds(200)); }, std::chrono::milliseconds(0));
        w.kick();
        std::this_thread::sleep_for(std::chrono::microseconds(rng() % 400));
        w.kick();  // a kick while work() runs: one more run after it, unless stopped first
        w.stop();
        w.stop();  // idempotent
    }
}

static void scenario_contexts(std::mt19937_64& rng) {
    MockStore s;
    const size_t T = 2 + rng() % 7;  // up to 8 threads on 1 + 3 contexts
    std::vector<std::thread> th;
    for (size_t t = 0; t < T; t++)
        th.emplace_back([&, t] {
            std::mt19937_64 r2(t + 31);
            for (int i = 0; i < 6; i++) {
                SharedLock rd(s.rw);
                MockStore* c = s.ctx_acquire();
                for (volatile int k = 0; k < (int)(r2() % 300); k++) {}
                s.ctx_release(c);
            }
        });
    for (auto& t : th) t.join();
    CHECK(s.pool.workers.size() <= 3);
    CHECK(s.pool.waiters.load() == 0);
}

static void scenario_appends(std::mt19937_64& rng) {
    MockStore s;
    const int appenders = 1 + (int)(rng() % 2), readers = 1 + (int)(rng() % 3);
    std::atomic<bool> done{false};
    std::vector<std::thread> th;
    for (int a = 0; a < appenders; a++)
        th.emplace_back([&, a] {
            std::mt19937_64 r2(a * 13 + 5);
            for (int i = 0; i < 24; i++) {
                const bool small = (r2() % 4) != 0;
                s.append(small ? 1 + r2() % 3 : 20 + r2() % 200, small);
            }
        });
    for (int r = 0; r < readers; r++)
        th.emplace_back([&, r] {
            std::mt19937_64 r2(r * 7 + 1);
            uint64_t last = 0;
            while (!done.load(std::memory_order_acquire)) {
                s.query(r2);
                uint64_t now;
                {
                    SharedLock rd(s.rw);
                    now = s.len();
                }
                CHECK(now >= last);  // VecStore::len never shrinks
                last = now;
            }
        });
    for (int a = 0; a < appenders; a++) th[(size_t)a].join();
    done.store(true, std::memory_order_release);
    for (size_t i = (size_t)appenders; i < th.size(); i++) th[i].join();
    std::mt19937_64 r3(1);
    s.query(r3);
    CHECK(s.pend.count() == 0);
    // destroyed with the builder possibly mid-run
}

// a front store over shards (ott_multi.hip's shape): appends go to the last shard under the front's exclusive lock and dirty
// the layout; a query cleans it (flush + "rebalance") under the exclusive lock, then fans out over the shards' threads
struct MockMulti {
    RwGate rw;
    std::atomic<bool> dirty{false};
    std::vector<std::unique_ptr<MockStore>> shards;
    ShardPool pool;
    uint64_t n = 0;
    explicit MockMulti(size_t G) : pool(G) {
        for (size_t g = 0; g < G; g++) shards.emplace_back(new MockStore());
    }
    void append(uint64_t cnt, std::mt19937_64& rng) {
        ExclusiveLock wr(rw);
        shards[rng() % shards.size()]->append(cnt, cnt < 4);
        n += cnt;
        dirty.store(true, std::memory_order_release);
    }
    int clean() {
        ExclusiveLock wr(rw);
        if (!dirty.load(std::memory_order_acquire)) return 0;
        for (auto& sh : shards) CHECK(sh->flush() == 0);
        dirty.store(false, std::memory_order_release);
        return 0;
    }
    void query() {
        SharedLock rd;
        CHECK(lock_shared_clean(rw, rd, [this] { return dirty.load(std::memory_order_acquire); }, [this] { return clean(); }) == 0);
        std::vector<uint64_t> seen(shards.size(), 0);
        pool.run_all([&](size_t g) {
            MockStore& sh = *shards[g];
            CHECK(sh.pend.count() == 0);  // clean: no shard has staged rows, none reallocates under us
            MockStore* c = sh.ctx_acquire();
            seen[g] = sh.n;
            if (sh.n) CHECK(sh.rows[(sh.n - 1) * DIM] >= 0.f);
            sh.ctx_release(c);
        });
        uint64_t total = 0;
        for (uint64_t v : seen) total += v;
        CHECK(total == n);
    }
};

static void scenario_multi(std::mt19937_64& rng) {
    MockMulti m(1 + rng() % 5);
    std::vector<std::thread> th;
    th.emplace_back([&] {
        std::mt19937_64 r2(9);
        for (int i = 0; i < 16; i++) m.append(1 + r2() % 40, r2);
    });
    const int q = 1 + (int)(rng() % 3);
    for (int t = 0; t < q; t++)
        th.emplace_back([&] {
            for (int i = 0; i < 8; i++) m.query();
        });
    for (auto& t : th) t.join();
    m.query();
}

int main(int argc, char** argv) {
    const uint64_t schedules = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1000;
    const uint64_t seed = argc > 2 ? strtoull(argv[2], nullptr, 10) : 1;
    const int only = argc > 3 ? atoi(argv[3]) : -1;  // one scenario kind (0 .. 4) instead of all five in turn
    uint64_t per[5] = {0, 0, 0, 0, 0};
    for (uint64_t i = 0; i < schedules; i++) {
        std::mt19937_64 rng(seed * 1000003 + i);
        const int which = only >= 0 ? only : (int)(i % 5);
        per[which]++;
        switch (which) {
            case 0: scenario_pool(rng); break;
            case 1: scenario_lifetime(rng); break;
            case 2: scenario_contexts(rng); break;
            case 3: scenario_appends(rng); break;
            default: scenario_multi(rng); break;
        }
        if (g_fail.load()) break;
    }
    if (g_fail.load()) {
        fprintf(stderr, "FAILED: %llu checks\n", (unsigned long long)g_fail.load());
        return 1;
    }
    printf("OK schedules=%llu (pool %llu, lifetime %llu, contexts %llu, appends %llu, multi %llu) seed=%llu\n", (unsigned long long)schedules,
           (unsigned long long)per[0], (unsigned long long)per[1], (unsigned long long)per[2], (unsigned long long)per[3], (unsigned long long)per[4],
           (unsigned long long)seed);
    return 0;
}

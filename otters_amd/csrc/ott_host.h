// ott_host.h — the host-side concurrency of libotters_hip, free of HIP.
//
// The reference gets its thread safety from the borrow checker: VecStore::query is `&self` (src/vec.rs:387), MetaStore is !Sync
// (src/meta.rs:54), the fan-out is rayon's (src/meta.rs:678).  The C++ behind the C ABI has to earn it: a thread pool that fans
// a query out over the shards of a multi-GPU store, a pool of query contexts so that overlapping `&self` queries run side by
// side, appends staged on the host that must be invisible to every reader, a background thread that converts rows after
// appends have gone quiet, the "take the store shared only once nothing is left to do first" step in front of every query, and
// the reader / writer lock itself.  Those six pieces live HERE, as plain C++17 with every device call behind a callback, so that the very code the library
// ships is also compiled into tests/host/host_concurrency.cpp and run against a mock device under -fsanitize=thread and
// -fsanitize=address,undefined in the CPU test suite (tests/test_host_concurrency.py).  ott_store.hip / ott_multi.hip /
// ott_api.hip only bind them to ott_store.
#pragma once

#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <thread>
#include <vector>

namespace ott {
namespace host {

inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
}

// ---- RwGate --------------------------------------------------------------------------------------------------------------------
// The store's reader / writer lock: queries hold it shared, everything that changes the store holds it exclusively.  A plain
// std::shared_mutex (glibc's pthread_rwlock) PREFERS READERS: while queries from two or more threads keep overlapping, an append
// waits for as long as they do — found by this header's own test (tests/host: a schedule of three query threads and one
// appender took 1.2 s under ThreadSanitizer, nearly all of it the appender waiting).  RwGate is that mutex with a door in front
// of the readers: a writer announces itself before it queues, and readers that arrive while a writer is announced let it in
// first.  Readers already inside are not disturbed; a reader that finds the door open pays one relaxed load.  Usable with
// std::shared_lock / std::unique_lock.
//
// NOT re-entrant: a thread that holds the gate shared must not take it shared again (through a nested library call on the same
// store, say) — once a writer has announced itself in between, the inner acquisition waits for the writer, the writer waits for the
// outer one to leave, and nobody moves.  The library takes the gate exactly once per public entry point (ott_api.hip, ott_comm.hip,
// ott_multi.hip); helpers below that level receive the store already locked.
//
// Readers held at the door sleep on a condition variable the writer signals once it HAS the exclusive lock (round 6; until then
// they polled with yield and then 50-us sleeps, so a query issued right behind an append could pick up 50 us or more for nothing):
// from there they queue on the mutex itself and are woken by its unlock.
class RwGate {
  public:
    void lock() {
        writers_.fetch_add(1, std::memory_order_acq_rel);
        m_.lock();
        if (writers_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
            { std::lock_guard<std::mutex> g(door_mu_); }  // (a reader between its check and its wait holds door_mu_: no lost wake-up)
            door_cv_.notify_all();
        }
    }
    bool try_lock() { return m_.try_lock(); }
    void unlock() { m_.unlock(); }
    void lock_shared() {
        if (writers_.load(std::memory_order_acquire) > 0) {
            for (int spin = 0; spin < 64 && writers_.load(std::memory_order_acquire) > 0; spin++) std::this_thread::yield();
            if (writers_.load(std::memory_order_acquire) > 0) {
                std::unique_lock<std::mutex> lk(door_mu_);
                door_cv_.wait(lk, [this] { return writers_.load(std::memory_order_acquire) == 0; });
            }
        }
        m_.lock_shared();
    }
    bool try_lock_shared() { return writers_.load(std::memory_order_acquire) == 0 && m_.try_lock_shared(); }
    void unlock_shared() { m_.unlock_shared(); }

  private:
    std::shared_mutex m_;
    std::atomic<int> writers_{0};  // writers announced: waiting for, or about to take, the exclusive lock
    std::mutex door_mu_;
    std::condition_variable door_cv_;
};
using SharedLock = std::shared_lock<RwGate>;
using ExclusiveLock = std::unique_lock<RwGate>;

// ---- ShardPool -----------------------------------------------------------------------------------------------------------------
// One persistent host thread per shard (the rayon pool of src/meta.rs:678, sized to the GPUs).  run_all(fn) runs fn(0) on the
// calling thread and fn(g) on shard g's thread, and returns when all are done; concurrent callers interleave per shard (every
// worker serves its queue in order).  A call's completion latch is owned jointly by the caller and by every task of the call
// (shared_ptr): whoever touches it last frees it.
class ShardPool {
  public:
    explicit ShardPool(size_t n) : w_(n) {
        for (size_t g = 1; g < n; g++) w_[g].th = std::thread([this, g] { loop(g); });
    }
    ~ShardPool() {
        for (size_t g = 1; g < w_.size(); g++) {
            {
                std::lock_guard<std::mutex> lk(w_[g].mu);
                w_[g].stop = true;
            }
            w_[g].cv.notify_all();
            if (w_[g].th.joinable()) w_[g].th.join();
        }
    }
    ShardPool(const ShardPool&) = delete;
    ShardPool& operator=(const ShardPool&) = delete;
    size_t size() const { return w_.size(); }

    void run_all(const std::function<void(size_t)>& fn) {
        const size_t n = w_.size();
        std::shared_ptr<Latch> latch;
        if (n > 1) {
            latch = std::make_shared<Latch>();
            latch->left.store((int)n - 1, std::memory_order_relaxed);
            for (size_t g = 1; g < n; g++) {
                {
                    std::lock_guard<std::mutex> lk(w_[g].mu);
                    w_[g].q.push_back(Task{&fn, latch});
                    w_[g].has_work.store(true, std::memory_order_release);
                }
                w_[g].cv.notify_one();
            }
        }
        fn(0);
        if (n > 1) {
            // the shards' tasks are a few launches each: spin briefly before sleeping (a wake-up costs more than most of them)
            for (int spin = 0; spin < 4000 && latch->left.load(std::memory_order_acquire) > 0; spin++) cpu_relax();
            if (latch->left.load(std::memory_order_acquire) > 0) {
                std::unique_lock<std::mutex> lk(latch->mu);
                latch->cv.wait(lk, [&] { return latch->left.load(std::memory_order_acquire) <= 0; });
            }
        }
    }

  private:
    struct Latch {
        std::atomic<int> left{0};
        std::mutex mu;
        std::condition_variable cv;
    };
    struct Task {
        const std::function<void(size_t)>* fn = nullptr;  // the caller's: alive until its run_all returns, i.e. until `left` is 0
        std::shared_ptr<Latch> latch;
    };
    struct Worker {
        std::thread th;
        std::mutex mu;
        std::condition_variable cv;
        std::deque<Task> q;
        std::atomic<bool> has_work{false};
        bool stop = false;
    };
    bool pop(Worker& w, Task& t) {  // w.mu held
        if (w.q.empty()) return false;
        t = std::move(w.q.front());
        w.q.pop_front();
        if (w.q.empty()) w.has_work.store(false, std::memory_order_release);
        return true;
    }
    void loop(size_t g) {
        Worker& w = w_[g];
        for (;;) {
            Task t;
            bool got = false;
            // a stream of queries keeps the shard threads hot: look for the next task for ~50 us before going to sleep (a
            // condition-variable wake-up costs 20-50 us, which is most of a small query's fan-out)
            for (int spin = 0; spin < 2000 && !got; spin++) {
                if (w.has_work.load(std::memory_order_acquire)) {
                    std::lock_guard<std::mutex> lk(w.mu);
                    got = pop(w, t);
                } else {
                    cpu_relax();
                }
            }
            if (!got) {
                std::unique_lock<std::mutex> lk(w.mu);
                w.cv.wait(lk, [&] { return w.stop || !w.q.empty(); });
                if (!pop(w, t)) return;  // stop, and nothing left to run
            }
            (*t.fn)(g);
            // the decrement happens under the latch's mutex, so the caller — spinning or asleep — cannot miss the notify
            std::lock_guard<std::mutex> lk(t.latch->mu);
            if (t.latch->left.fetch_sub(1, std::memory_order_acq_rel) == 1) t.latch->cv.notify_all();
        }
    }
    std::vector<Worker> w_;
};

// ---- ContextPool ---------------------------------------------------------------------------------------------------------------
// a query is re-entrant on one store from several host threads.  A query runs on a CONTEXT (stream, events,
// scratch), guarded by the context's `mu`: the store's own when it is free, otherwise a worker context created on first
// overlap (at most `max_workers`), otherwise the caller waits for a release.  Ctx needs a public `std::mutex mu`.
// acquire() returns a context with its `mu` held; release() gives it back.  `make` creates a worker (may return nullptr: the
// caller then waits like at the limit), `prepare` is run on a worker each time it is handed out (it aliases the owner's
// current corpus).  Workers are owned by the pool's user (`workers` is walked at destruction, when no query runs).
template <class Ctx>
struct ContextPool {
    std::mutex pool_mu;
    std::condition_variable pool_cv;  // signalled when a context is released while callers wait for one
    std::atomic<int> waiters{0};
    std::vector<Ctx*> workers;

    template <class Make, class Prepare>
    Ctx* acquire(Ctx* own, size_t max_workers, Make&& make, Prepare&& prepare) {
        if (own->mu.try_lock()) return own;  // the common, uncontended case: the store's own context
        std::unique_lock<std::mutex> g(pool_mu);
        for (;;) {
            if (own->mu.try_lock()) return own;
            for (Ctx* w : workers)
                if (w->mu.try_lock()) {
                    prepare(w);
                    return w;
                }
            if (workers.size() < max_workers) {
                Ctx* w = make();
                if (w) {
                    w->mu.lock();
                    workers.push_back(w);
                    prepare(w);
                    return w;
                }
            }
            // every context is busy: wait for a release (the timeout covers a release that slipped in before the wait)
            waiters.fetch_add(1);
            pool_cv.wait_for(g, std::chrono::milliseconds(2));
            waiters.fetch_sub(1);
        }
    }
    void release(Ctx* got) {
        got->mu.unlock();
        if (waiters.load() > 0) {
            std::lock_guard<std::mutex> g(pool_mu);
            pool_cv.notify_one();
        }
    }
};

// ---- StagedRows ----------------------------------------------------------------------------------------------------------------
// Small appends (VecStore::add_vector is one row per call, src/vec.rs:357-371) collect in a host buffer and go to the device
// together.  The buffer and `rows` are written only by a thread that holds the store EXCLUSIVELY; `rows` is atomic because
// len() (rows appended = resident + staged) and the "is anything staged?" look of lock_shared_clean read it without a lock.
struct StagedRows {
    float* buf = nullptr;  // capacity `cap_bytes` (pinned host memory in the library; owned by the user of this struct)
    size_t cap_bytes = 0;
    std::atomic<uint64_t> rows{0};

    uint64_t count() const { return rows.load(std::memory_order_acquire); }
    bool fits(uint64_t n_rows, uint32_t dim) const { return (rows.load(std::memory_order_relaxed) + n_rows) * (uint64_t)dim * 4 <= cap_bytes; }
    void stage(const float* src, uint64_t n_rows, uint32_t dim) {  // the caller has checked fits()
        const uint64_t p = rows.load(std::memory_order_relaxed);
        memcpy(buf + p * dim, src, (size_t)n_rows * dim * 4);
        rows.store(p + n_rows, std::memory_order_release);
    }
    // hands the staged rows to `sink(buf, n)`; they stay staged when the sink fails (nothing is lost, the next flush tries again)
    template <class Sink>
    int flush(Sink&& sink) {
        const uint64_t p = rows.load(std::memory_order_acquire);
        if (!p) return 0;
        const int rc = sink((const float*)buf, p);
        if (rc) return rc;
        rows.store(0, std::memory_order_release);
        return 0;
    }
};

// ---- lock_shared_clean -----------------------------------------------------------------------------------------------------------
// Takes `rw` SHARED with nothing left to do first.  `dirty()` says whether something must happen under the EXCLUSIVE lock
// before readers may look (rows still staged on the host; a multi-GPU store whose balance has not been looked at since the
// last append); `clean()` does it, taking `rw` exclusively itself, and returns 0 or an error.  A writer may slip in between
// clean()'s unlock and the shared lock, so dirty() is asked again under the shared lock and the step repeated.  Writers make
// the state dirty only under the exclusive lock, so whoever holds the returned lock can rely on it staying clean.
template <class Dirty, class Clean>
int lock_shared_clean(RwGate& rw, SharedLock& rd, Dirty&& dirty, Clean&& clean) {
    for (;;) {
        if (dirty()) {
            const int rc = clean();
            if (rc) return rc;
        }
        rd = SharedLock(rw);
        if (!dirty()) return 0;
        rd.unlock();
    }
}

// ---- QuietWorker ---------------------------------------------------------------------------------------------------------------
// A background thread that runs `work()` once kicks have been quiet for `quiet`: every append kicks it, a store loaded in
// pieces is not converted piece by piece, and a kick that arrives while work() runs causes one more run after the next quiet
// period.  stop() (and the destructor) let a running work() finish, then join.  work() runs without any lock of this class.
class QuietWorker {
  public:
    QuietWorker(std::function<void()> work, std::chrono::milliseconds quiet) : work_(std::move(work)), quiet_(quiet), th_([this] { loop(); }) {}
    ~QuietWorker() { stop(); }
    QuietWorker(const QuietWorker&) = delete;
    QuietWorker& operator=(const QuietWorker&) = delete;
    void kick() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            want_ = true;
        }
        cv_.notify_one();
    }
    void stop() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        if (th_.joinable()) th_.join();
    }
    uint64_t runs() const { return runs_.load(std::memory_order_acquire); }

  private:
    void loop() {
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return want_ || stop_; });
                if (stop_) return;
                while (want_ && !stop_) {  // wait until the kicks have been quiet for `quiet_`
                    want_ = false;
                    cv_.wait_for(lk, quiet_, [&] { return want_ || stop_; });
                }
                if (stop_) return;
            }
            work_();
            runs_.fetch_add(1, std::memory_order_release);
        }
    }
    std::function<void()> work_;
    std::chrono::milliseconds quiet_;
    std::mutex mu_;
    std::condition_variable cv_;
    bool want_ = false, stop_ = false;
    std::atomic<uint64_t> runs_{0};
    std::thread th_;  // last: started once everything above exists
};

}  // namespace host
}  // namespace ott

// ott_multi.hip — ONE store over several GPUs inside the host's one process (ott_store_create_multi).
//
// The reference is a single-process library: MetaQueryPlan::collect fans the surviving chunks out over a rayon pool and
// concat-sort-truncates the per-chunk top-k lists (src/meta.rs:678-709), and VecStore::query is an ordinary `&self` call
// (src/vec.rs:387).  This file is that shape with GPUs in place of rayon tasks: the store owns one SHARD per device — an
// ordinary single-GPU ott_store holding a contiguous range of chunks, with its own streams and scratch — and the unchanged
// entry points (ott_store_append*, ott_store_add_column, ott_store_eval_row_mask, ott_store_zone_stats, ott_query, ...) route
// by row range.  A query is
//   1. every shard scores its rows (one host thread per shard queues the work: the batch path waits on the host between its
//      levels, so a single thread would serialise the GPUs); chunk mask, row mask and metadata columns are sliced per shard;
//   2. ONE exchange of the fixed-size, sentinel-padded candidate blocks to the merging GPU (the first shard's): peer copies
//      ordered by events — or nothing at all for shards that share its device — or a grouped ncclAllGather over one RCCL
//      communicator per device (ncclCommInitAll) when the device ordinals are distinct;
//   3. merge_hits_kernel there, written straight into pinned host memory, and ONE host wait.
// Shards are in row order (shard g holds lower rows than shard g + 1) and start on multiples of lcm(chunk size, 8) rows, so
// ties across shards resolve exactly as on one GPU, in the canonical order and in the reference's (tie_order 1 / 2: the
// decision logic of ott_ties.hip runs here over candidates gathered from all shards).
//
// Layout.  `start[g]` = first row of shard g.  With a planned size (ott_store_reserve) the ranges are the even split of the
// plan's granules and appends fill them in order; without one, rows go to the last shard that has any, and before the first
// query that follows (or on ott_store_reserve) rows are MOVED between neighbours — hipMemcpyPeerAsync into freshly
// allocated buffers — whenever one shard holds more than 1.25x its even share.  Results never depend on where rows live.
#include <string.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <numeric>
#include <thread>
#include <vector>

#include "ott_internal.h"

using namespace ott;

namespace {

constexpr uint64_t NOT_YET = ~0ull;  // start[] of a shard the appends have not reached

uint64_t now_ns() {
    return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

using ott::host::ShardPool;  // one persistent host thread per shard (ott_host.h: the rayon pool of src/meta.rs:678, sized to the GPUs)

uint64_t gcd64(uint64_t a, uint64_t b) {
    while (b) {
        const uint64_t t = a % b;
        a = b;
        b = t;
    }
    return a;
}

}  // namespace

struct ott_multi {
    std::vector<ott_store*> shards;
    std::vector<int> devs;
    std::vector<int> logical;     // [G] logical device id of shard g: devs[g], or one of its own per shard under option multi_fake_distinct
    std::vector<uint64_t> start;  // [G] first row (counted from the store's first) of shard g; NOT_YET = not reached
    uint64_t plan_rows = 0;       // the size the current ranges were laid out for (0 = no plan yet)
    bool distinct = false;        // every (logical) device is different
    std::atomic<bool> layout_dirty{false};  // rows were appended since the balance was last looked at (set under the store's lock, read without)
    ShardPool* pool = nullptr;
    // RCCL transport: one communicator per shard (ncclCommInitAll), created on first need
    std::vector<void*> nccl;
    int transport = 0;            // resolved: 1 = peer copies, 2 = RCCL (0 = not decided yet)
    std::mutex xchg_mu;           // RCCL: one grouped collective at a time (concurrent queries must not interleave theirs)
    std::string rccl_why;         // why the automatic choice did not take RCCL
};

namespace {

size_t n_shards(const ott_store* ms) { return ms->multi->shards.size(); }

uint64_t granule_of(uint64_t chunk_size) { return chunk_size / gcd64(chunk_size, 8) * 8; }  // lcm(chunk size, 8)

// the even split of `total` rows over the shards that are worth bringing in: with fewer than `min_rows` rows per shard only the
// first total / min_rows of them are used (at least one), shard g of those starts at (g * granules / used) granules and the
// others sit empty at the end
std::vector<uint64_t> ideal_starts(uint64_t total, uint64_t granule, size_t G, uint64_t min_rows) {
    const uint64_t n_gran = (total + granule - 1) / granule;
    size_t used = G;
    if (min_rows > 0) {
        const uint64_t fit = total / min_rows;
        used = fit < 1 ? 1 : (fit < G ? (size_t)fit : G);
    }
    std::vector<uint64_t> st(G);
    for (size_t g = 0; g < G; g++) st[g] = g < used ? (uint64_t)((unsigned __int128)g * n_gran / used) * granule : n_gran * granule;
    return st;
}
uint64_t min_rows_of(const ott_store* ms) { return (uint64_t)ms->opt.multi_min_shard_rows; }
// shards the layout for `plan_rows` rows does not use (ideal_starts puts them at the plan's end) have not begun: appends past
// the plan go to the last shard that HAS (and the next look at the balance brings the next shard in)
void mark_unused(std::vector<uint64_t>& start, uint64_t plan_rows, uint64_t granule) {
    const uint64_t end = (plan_rows + granule - 1) / granule * granule;
    for (size_t g = start.size(); g-- > 1;) {
        if (start[g] != NOT_YET && start[g] < end) break;
        start[g] = NOT_YET;
    }
}

// the shard that takes row p next (the last one whose range has begun)
size_t route(const ott_multi* m, uint64_t p) {
    size_t g = 0;
    for (size_t i = 1; i < m->shards.size(); i++)
        if (m->start[i] != NOT_YET && m->start[i] <= p) g = i;
    return g;
}

// first row of shard g for indexing purposes (a shard not reached yet sits at the end)
uint64_t start_of(const ott_store* ms, size_t g) {
    const uint64_t s = ms->multi->start[g];
    return s == NOT_YET ? ms->n : s;
}

void set_shard_bases(ott_store* ms) {
    ott_multi* m = ms->multi;
    for (size_t g = 0; g < m->shards.size(); g++) m->shards[g]->base_offset = ms->base_offset + start_of(ms, g);
}

struct Piece {
    size_t g;
    uint64_t first_global;  // first row of the piece, counted from the store's first row
    uint64_t first_local;   // the same row inside shard g
    uint64_t count;
};

// rows [first, first + count) as per-shard pieces (existing rows)
std::vector<Piece> pieces_of(const ott_store* ms, uint64_t first, uint64_t count) {
    std::vector<Piece> out;
    const ott_multi* m = ms->multi;
    for (size_t g = 0; g < m->shards.size() && count; g++) {
        const uint64_t s0 = start_of(ms, g), n_g = store_rows(m->shards[g]);
        if (!n_g || first >= s0 + n_g || first + count <= s0) continue;
        const uint64_t lo = first > s0 ? first : s0, hi = (first + count < s0 + n_g) ? first + count : s0 + n_g;
        out.push_back({g, lo, lo - s0, hi - lo});
    }
    return out;
}

int run_on_shards(ott_store* ms, const std::function<int(size_t)>& fn) {
    ott_multi* m = ms->multi;
    const size_t G = m->shards.size();
    std::vector<int> rc(G, OTT_OK);
    std::vector<std::string> msg(G);
    m->pool->run_all([&](size_t g) {
        rc[g] = fn(g);
        if (rc[g]) msg[g] = ott_last_error();  // thread-local on the shard's thread: carried back to the caller's
    });
    for (size_t g = 0; g < G; g++)
        if (rc[g]) return fail(rc[g], "shard " + std::to_string(g) + " (device " + std::to_string(m->devs[g]) + "): " + msg[g]);
    return OTT_OK;
}

// ---- layout -----------------------------------------------------------------------------------------------------------------

bool any_columns(const ott_store* ms) {
    for (ott_store* s : ms->multi->shards)
        if (!s->columns.empty()) return true;
    return false;
}

// every shard that holds rows (beyond the first) starts on a granule boundary
bool layout_aligned(const ott_store* ms, uint64_t granule) {
    const ott_multi* m = ms->multi;
    for (size_t g = 1; g < m->shards.size(); g++)
        if (store_rows(m->shards[g]) && m->start[g] % granule != 0) return false;
    return true;
}

// Moves rows so that shard g holds [target[g], min(target[g + 1], n)).  Every shard whose range changes gets fresh buffers on
// its device (transiently old + new: when that does not fit, nothing has moved and OTT_ERR_OOM is returned), filled by peer
// copies from the old buffers of whichever shards hold its rows.  The caller holds the store exclusively.
int relayout(ott_store* ms, const std::vector<uint64_t>& target, uint64_t plan_rows) {
    ott_multi* m = ms->multi;
    const size_t G = m->shards.size();
    const uint64_t n = ms->n;
    for (ott_store* sh : m->shards) {  // rows of small appends still staged on the host are moved like the others
        const int rcf = store_flush(sh);
        if (rcf) return rcf;
    }
    struct Range {
        uint64_t lo, hi;
    };
    std::vector<Range> old_r(G), new_r(G);
    bool moves = false;
    for (size_t g = 0; g < G; g++) {
        const uint64_t s0 = start_of(ms, g);
        old_r[g] = {s0, s0 + store_rows(m->shards[g])};
        const uint64_t lo = target[g] < n ? target[g] : n;
        const uint64_t hi = (g + 1 < G) ? (target[g + 1] < n ? target[g + 1] : n) : n;
        new_r[g] = {lo, hi > lo ? hi : lo};
        if (new_r[g].hi - new_r[g].lo != old_r[g].hi - old_r[g].lo || (new_r[g].hi > new_r[g].lo && new_r[g].lo != old_r[g].lo)) moves = true;
    }
    if (moves && any_columns(ms))
        return fail(OTT_ERR_UNSUPPORTED, "multi-GPU store: rows cannot move between GPUs once metadata columns are resident (reserve the final size, "
                                         "set the chunk size and append the rows before ott_store_add_column)");
    struct Fresh {
        float* rows = nullptr;
        float* inv = nullptr;
        uint8_t* flag = nullptr;
        uint64_t cap = 0;
        bool changed = false;
    };
    std::vector<Fresh> fr(G);
    auto drop_fresh = [&]() {
        for (size_t g = 0; g < G; g++) {
            (void)use_device(m->shards[g]);
            if (fr[g].rows) (void)hipFree(fr[g].rows);
            if (fr[g].inv) (void)hipFree(fr[g].inv);
            if (fr[g].flag) (void)hipFree(fr[g].flag);
        }
    };
    if (moves) {
        // phase A: fresh buffers for the shards whose range changes (capacity: the planned range, at least what they hold)
        for (size_t g = 0; g < G; g++) {
            const uint64_t cnt = new_r[g].hi - new_r[g].lo;
            if (cnt == old_r[g].hi - old_r[g].lo && (cnt == 0 || new_r[g].lo == old_r[g].lo)) continue;
            ott_store* s = m->shards[g];
            fr[g].changed = true;
            uint64_t cap = cnt;
            if (g + 1 < G && target[g + 1] > target[g] && target[g + 1] - target[g] > cap) cap = target[g + 1] - target[g];
            if (cap < 1024) cap = 1024;
            if (cap > 0xFFFFFFF0ull) {
                drop_fresh();
                return fail(OTT_ERR_UNSUPPORTED, "a store holds at most 2^32-16 rows per GPU");
            }
            fr[g].cap = cap;
            hipError_t e = use_device(m->shards[g]);
            if (e == hipSuccess) e = hipMalloc((void**)&fr[g].rows, cap * s->ld * sizeof(float));
            if (e == hipSuccess) e = hipMalloc((void**)&fr[g].inv, cap * sizeof(float));
            if (e == hipSuccess) e = hipMalloc((void**)&fr[g].flag, cap);
            if (e == hipSuccess && s->ld != s->dim && cap > cnt)  // padding columns of rows still to come must be zero
                e = hipMemsetAsync(fr[g].rows + cnt * s->ld, 0, (cap - cnt) * s->ld * sizeof(float), s->stream);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                drop_fresh();
                return fail(e == hipErrorOutOfMemory ? OTT_ERR_OOM : OTT_ERR_HIP, std::string("multi-GPU store: moving rows between GPUs: ") + hipGetErrorString(e));
            }
        }
        // phase B: the copies, each on the receiving shard's stream
        for (size_t g = 0; g < G; g++) {
            if (!fr[g].changed) continue;
            ott_store* s = m->shards[g];
            OTT_HIP(use_device(m->shards[g]));
            for (size_t h = 0; h < G; h++) {
                const uint64_t lo = new_r[g].lo > old_r[h].lo ? new_r[g].lo : old_r[h].lo;
                const uint64_t hi = new_r[g].hi < old_r[h].hi ? new_r[g].hi : old_r[h].hi;
                if (hi <= lo) continue;
                const ott_store* src = m->shards[h];
                const uint64_t cnt = hi - lo, d0 = lo - new_r[g].lo, s0 = lo - old_r[h].lo;
                hipError_t e = hipMemcpyPeerAsync(fr[g].rows + d0 * s->ld, m->devs[g], src->d_rows + s0 * src->ld, m->devs[h], cnt * s->ld * sizeof(float), s->stream);
                if (e == hipSuccess) e = hipMemcpyPeerAsync(fr[g].inv + d0, m->devs[g], src->d_inv + s0, m->devs[h], cnt * sizeof(float), s->stream);
                if (e == hipSuccess) e = hipMemcpyPeerAsync(fr[g].flag + d0, m->devs[g], src->d_flag + s0, m->devs[h], cnt, s->stream);
                if (e != hipSuccess) {
                    (void)hipGetLastError();
                    for (size_t x = 0; x < G; x++) {
                        (void)use_device(m->shards[x]);
                        (void)hipStreamSynchronize(m->shards[x]->stream);
                    }
                    drop_fresh();
                    return fail(OTT_ERR_HIP, std::string("multi-GPU store: hipMemcpyPeerAsync: ") + hipGetErrorString(e));
                }
            }
        }
        for (size_t g = 0; g < G; g++) {
            OTT_HIP(use_device(m->shards[g]));
            OTT_HIP(hipStreamSynchronize(m->shards[g]->stream));
        }
        // phase C: the shards take the fresh buffers over (the old ones are freed)
        for (size_t g = 0; g < G; g++) {
            if (!fr[g].changed) continue;
            const int rc = store_adopt(m->shards[g], fr[g].rows, fr[g].inv, fr[g].flag, new_r[g].hi - new_r[g].lo, fr[g].cap);
            fr[g].rows = nullptr;
            fr[g].inv = nullptr;
            fr[g].flag = nullptr;
            if (rc) return rc;
        }
    }
    m->start = target;
    mark_unused(m->start, plan_rows, granule_of(ms->chunk_size));
    m->plan_rows = plan_rows;
    set_shard_bases(ms);
    ms->evalmask_bits = moves ? 0 : ms->evalmask_bits;
    return OTT_OK;
}

// Before a query (and on reserve): rows appended past the plan sit in one shard; move them when it holds more than 1.25x
// its even share (amortised: a relayout follows at least 25 % growth), or when the layout is not on granule boundaries.
int ensure_layout(ott_store* ms, bool force) {
    ott_multi* m = ms->multi;
    const size_t G = m->shards.size();
    const uint64_t gran = granule_of(ms->chunk_size);
    for (ott_store* sh : m->shards) {  // rows of small appends still staged on the host go to their GPUs first
        const int rcf = store_flush(sh);
        if (rcf) return rcf;
    }
    const bool aligned = layout_aligned(ms, gran);
    // (layout_dirty is cleared by the caller, clean_locked, once this has returned OK — a failed move is tried again)
    if (G == 1 || ms->n == 0) return OTT_OK;
    const uint64_t total = ms->n > m->plan_rows ? ms->n : m->plan_rows;
    const std::vector<uint64_t> tgt = ideal_starts(total, gran, G, min_rows_of(ms));
    if (aligned && !force) {
        if (!ms->opt.multi_rebalance) return OTT_OK;
        // balanced enough?  the fullest shard against the fullest range of an even split of the rows that exist
        const std::vector<uint64_t> even = ideal_starts(ms->n, gran, G, min_rows_of(ms));
        uint64_t share = 0, fullest = 0;
        for (size_t g = 0; g < G; g++) {
            const uint64_t hi = g + 1 < G ? even[g + 1] : ms->n;
            const uint64_t lo = even[g] < ms->n ? even[g] : ms->n;
            share = std::max(share, (hi < ms->n ? hi : ms->n) - lo);
            fullest = std::max(fullest, store_rows(m->shards[g]));
        }
        if (fullest <= share + share / 4 + gran) return OTT_OK;
        if (any_columns(ms)) return OTT_OK;  // (rows cannot move any more: stay as they are — slower, not wrong)
        const int rc = relayout(ms, ideal_starts(ms->n, gran, G, min_rows_of(ms)), ms->n);
        if (rc == OTT_ERR_OOM) return OTT_OK;  // no room for the transient copy: stay unbalanced
        return rc;
    }
    return relayout(ms, tgt, total);
}

// Takes the store SHARED with nothing left to do first: rows of small appends still staged on the host are on their GPUs and the
// balance has been looked at.  Both need the store exclusively, and an append may slip in between giving that up and taking
// the shared lock — so the flag is looked at again under the shared lock and the step repeated (appends set layout_dirty under
// the exclusive lock; staged rows only ever appear together with it).  Readers and queries that hold the returned lock can
// rely on: no shard has staged rows, no shard's buffers are reallocated under them.
int lock_clean(ott_store* ms, ott::host::SharedLock& rd) {
    ott_multi* m = ms->multi;
    return ott::host::lock_shared_clean(
        ms->rw, rd, [m] { return m->layout_dirty.load(std::memory_order_acquire); },
        [ms, m]() -> int {
            ott::host::ExclusiveLock wr(ms->rw);
            if (!m->layout_dirty.load(std::memory_order_acquire)) return OTT_OK;
            const int rc = ensure_layout(ms, false);  // (flushes every shard's staged rows first)
            if (rc) return rc;
            m->layout_dirty.store(false, std::memory_order_release);
            return OTT_OK;
        });
}

// ---- slicing the per-query masks ------------------------------------------------------------------------------------------------

// bits [first, first + count) of a BitVec<usize, Lsb0> -> words starting at bit 0 (first is a multiple of 8)
void slice_bits_bytes(const uint64_t* words, uint64_t first, uint64_t count, std::vector<uint64_t>& out) {
    out.assign((size_t)((count + 63) / 64) + 1, 0);
    if (!count) return;
    memcpy(out.data(), (const uint8_t*)words + first / 8, (size_t)((count + 7) / 8));
    if (count & 63) out[(size_t)(count >> 6)] &= (1ull << (count & 63)) - 1;  // nothing of the next shard's rows
}
void slice_bits_any(const uint64_t* words, uint64_t first, uint64_t count, std::vector<uint64_t>& out) {
    out.assign((size_t)((count + 63) / 64) + 1, 0);
    for (uint64_t i = 0; i < count; i++) {
        const uint64_t b = first + i;
        if ((words[b >> 6] >> (b & 63)) & 1) out[(size_t)(i >> 6)] |= 1ull << (i & 63);
    }
}

// ---- RCCL transport -------------------------------------------------------------------------------------------------------------

int nccl_error(const char* what, int code) {
    Rccl* r = rccl();
    const char* msg = (r->GetErrorString && code) ? r->GetErrorString(code) : "?";
    return fail(OTT_ERR_HIP, std::string(what) + ": " + msg);
}

// decides the transport once (under xchg_mu): 1 = peer copies, 2 = RCCL
int resolve_transport(ott_store* ms) {
    ott_multi* m = ms->multi;
    if (m->transport) return OTT_OK;
    const int want = ms->opt.multi_transport;
    const size_t G = m->shards.size();
    if (want == 1 || (want == 0 && (!m->distinct || G == 1))) {  // (one shard: nothing to exchange, no reason to load RCCL)
        m->transport = 1;
        return OTT_OK;
    }
    if (!m->distinct) return fail(OTT_ERR_UNSUPPORTED, "multi_transport = 2 (RCCL) needs distinct device ordinals: RCCL refuses two ranks on one GPU");
    Rccl* r = rccl();
    const bool usable = r->handle && r->CommInitAll && r->GroupStart && r->GroupEnd;
    if (!usable) {
        m->rccl_why = r->handle ? "librccl.so.1 lacks ncclCommInitAll / ncclGroupStart / ncclGroupEnd" : r->why;
        if (want == 2) return fail(OTT_ERR_UNSUPPORTED, m->rccl_why);
        m->transport = 1;
        return OTT_OK;
    }
    m->nccl.assign(G, nullptr);
    const int rc = r->CommInitAll(m->nccl.data(), (int)G, m->devs.data());
    if (rc) {
        m->nccl.clear();
        if (want == 2) return nccl_error("ncclCommInitAll", rc);
        m->rccl_why = std::string("ncclCommInitAll: ") + (r->GetErrorString ? r->GetErrorString(rc) : "?");
        m->transport = 1;
        return OTT_OK;
    }
    m->transport = 2;
    return OTT_OK;
}

// ---- one query ----------------------------------------------------------------------------------------------------------------------

struct ShardSlice {
    std::vector<uint64_t> chunk_mask, row_mask;
    ott_query_desc d;
    bool idle = false;  // no rows: the shard contributes a block of sentinels
};

struct MultiCall {
    ott_store* ms;
    ott_multi* m;
    size_t G;
    std::vector<ott_store*> ctx;  // one query context per shard, held for the call

    explicit MultiCall(ott_store* front) : ms(front), m(front->multi), G(front->multi->shards.size()), ctx(G, nullptr) {
        for (size_t g = 0; g < G; g++) ctx[g] = ctx_acquire(m->shards[g]);  // in shard order: concurrent calls cannot wait on each other in a cycle
    }
    ~MultiCall() {
        for (size_t g = 0; g < G; g++)
            if (ctx[g]) ctx_release(ctx[g]);
    }

    // the caller's desc cut to every shard's rows
    void slice(const ott_query_desc& d, uint64_t k, std::vector<ShardSlice>& out) const {
        out.assign(G, {});
        const uint64_t cs = ms->chunk_size;
        for (size_t g = 0; g < G; g++) {
            ShardSlice& sl = out[g];
            const ott_store* s = m->shards[g];
            const uint64_t s0 = start_of(ms, g), n_g = store_rows(s);
            sl.d = d;
            sl.d.k = k;
            sl.d.chunk_mask = nullptr;
            sl.d.row_mask = nullptr;
            sl.d.row_mask_bits = 0;
            sl.idle = n_g == 0;
            if (sl.idle) continue;
            if (d.chunk_mask) {
                const uint64_t c0 = s0 / cs, nc = (n_g + cs - 1) / cs;
                if ((c0 & 7) == 0) slice_bits_bytes(d.chunk_mask, c0, nc, sl.chunk_mask);
                else slice_bits_any(d.chunk_mask, c0, nc, sl.chunk_mask);
                sl.d.chunk_mask = sl.chunk_mask.data();
            }
            if (d.row_mask && d.row_mask_bits > s0) {  // rows at and beyond row_mask_bits are kept (src/vec.rs:234)
                const uint64_t bits = (d.row_mask_bits - s0) < n_g ? d.row_mask_bits - s0 : n_g;
                slice_bits_bytes(d.row_mask, s0, bits, sl.row_mask);
                sl.d.row_mask = sl.row_mask.data();
                sl.d.row_mask_bits = bits;
            }
        }
    }

    static void add_stats(ott_stats& a, const ott_stats& b) {
        a.total_chunks += b.total_chunks;
        a.pruned_chunks += b.pruned_chunks;
        a.evaluated_chunks += b.evaluated_chunks;
        a.vectors_compared += b.vectors_compared;
        a.bytes_scanned += b.bytes_scanned;
        a.rescored += b.rescored;
        a.retries += b.retries;
        a.refined += b.refined;
        a.gate_failed += b.gate_failed;
        a.bound_violations += b.bound_violations;
        a.i8_refined += b.i8_refined;
        a.prune_ns = std::max(a.prune_ns, b.prune_ns);  // the shards run side by side: the slowest one counts
        a.score_ns = std::max(a.score_ns, b.score_ns);
        a.merge_ns = std::max(a.merge_ns, b.merge_ns);
        a.passes = std::max(a.passes, b.passes);
        a.path_used = std::max(a.path_used, b.path_used);
        a.err_ratio_max = std::max(a.err_ratio_max, b.err_ratio_max);
    }

    // One round over all shards with take count k: groups' hits best first into out[0 .. *n_out) (PER_QUERY: group after
    // group, counts in per[]).  cap >= min(k, pool) per group.
    int round(const ott_query_desc& d, uint64_t k, const CoreOpts& co, ott_hit* out, uint64_t cap, uint64_t* n_out, uint64_t* per, ott_stats* st_out) {
        const bool perq = d.mode == OTT_MODE_PER_QUERY;
        const uint32_t nq = d.nq, groups = perq ? nq : 1u;
        if (n_out) *n_out = 0;
        if (per)
            for (uint32_t i = 0; i < nq; i++) per[i] = 0;
        ott_stats st;
        memset(&st, 0, sizeof(st));
        if (k == 0) {  // src/vec_compute.rs:174
            if (st_out) *st_out = st;
            return OTT_OK;
        }
        std::vector<ShardSlice> sl;
        slice(d, k, sl);
        const int rc = k > 512 ? round_large(d, k, co, sl, groups, out, cap, n_out, per, st) : round_small(d, k, co, sl, groups, out, cap, n_out, per, st, st_out != nullptr);
        if (st_out) *st_out = st;
        return rc;
    }

    // k <= 512: fixed-size blocks [groups][KS], gathered on the first shard's GPU, one device merge, one host wait.
    // EVERY failure leaves through drain(): once the shards have queued work, their streams may still be writing into the
    // merging GPU's receive buffer (and, with RCCL, the peers' halves of the grouped all-gather sit on other contexts' streams)
    // — the contexts must not go back to the pool, nor the buffers be reused, before all of that has ended.
    int round_small(const ott_query_desc& d, uint64_t k, const CoreOpts& co, std::vector<ShardSlice>& sl, uint32_t groups, ott_hit* out, uint64_t cap,
                    uint64_t* n_out, uint64_t* per, ott_stats& st, bool timing) {
        const int rc = round_small_body(d, k, co, sl, groups, out, cap, n_out, per, st, timing);
        if (rc) {
            const std::string msg = ott_last_error();
            drain();
            return fail(rc, msg);
        }
        return OTT_OK;
    }
    int round_small_body(const ott_query_desc& d, uint64_t k, const CoreOpts& co, std::vector<ShardSlice>& sl, uint32_t groups, ott_hit* out, uint64_t cap,
                         uint64_t* n_out, uint64_t* per, ott_stats& st, bool timing) {
        int rc;
        const bool perq = d.mode == OTT_MODE_PER_QUERY;
        const int E = list_E(k);
        const uint64_t KS = 64ull * (uint64_t)E;
        const size_t block = (size_t)groups * KS * sizeof(ott_hit);
        ott_store* root = ctx[0];
        const int root_dev = m->devs[0];
        {
            std::lock_guard<std::mutex> g(m->xchg_mu);
            if ((rc = resolve_transport(ms))) return rc;
        }
        const bool use_rccl = m->transport == 2;
        // peer copies: shards without rows (a store smaller than its device list: option multi_min_shard_rows) take no part at all —
        // the blocks of the others sit back to back in the merging GPU's buffer.  (A collective needs every rank: RCCL keeps them.)
        std::vector<size_t> slot(G, 0);
        size_t n_lists = 0;
        for (size_t g = 0; g < G; g++) {
            const bool takes_part = use_rccl || !sl[g].idle;
            slot[g] = takes_part ? n_lists++ : (size_t)-1;
        }
        OTT_HIP(use_device(root));
        if ((rc = root->x_recv.ensure(block * (n_lists ? n_lists : 1)))) return rc;
        char* recv = (char*)root->x_recv.p;

        // 1. every shard: scoring + top-k into its block, queued by the shard's own host thread
        std::vector<ott_stats> sst(G);
        std::vector<char> pending(G, 0);
        rc = run_on_shards(ms, [&](size_t g) -> int {
            ott_store* c = ctx[g];
            const int dev = m->devs[g];
            memset(&sst[g], 0, sizeof(ott_stats));
            if (slot[g] == (size_t)-1) return OTT_OK;
            OTT_HIP(use_device(c));
            int r;
            // where the block is written: straight into the merging GPU's receive buffer when this shard shares its device
            void* dst = recv + slot[g] * block;
            const bool remote = use_rccl || m->logical[g] != m->logical[0];
            if (remote) {
                if ((r = c->x_send.ensure(block))) return r;
                dst = c->x_send.p;
                if (use_rccl && g != 0 && (r = c->x_recv.ensure(block * G))) return r;
            }
            OTT_AUDIT_PTR(dst, c);  // (audit build: the block this shard's kernels write lives on this shard's device)
            if (sl[g].idle) {
                OTT_HIP(hipMemsetAsync(dst, 0xFF, block, c->stream));
            } else {
                bool ev_pending = false;
                if ((r = query_core(c, &sl[g].d, nullptr, dst, (uint64_t)groups * KS, nullptr, nullptr, nullptr, timing ? &sst[g] : nullptr, true, &ev_pending, co)))
                    return r;
                pending[g] = ev_pending ? 1 : 0;
            }
            if (!use_rccl) {
                if (remote) OTT_HIP(hipMemcpyPeerAsync(recv + slot[g] * block, root_dev, dst, dev, block, c->stream));
                if (g != 0) OTT_HIP(hipEventRecord(c->ev[6], c->stream));
            }
            return OTT_OK;
        });
        if (rc) return rc;
        // 2. the exchange
        OTT_HIP(use_device(root));
        if (timing) OTT_HIP(hipEventRecord(root->ev[2], root->stream));  // behind the merging GPU's own scoring
        const void* lists = recv;
        if (use_rccl) {
            std::lock_guard<std::mutex> g(m->xchg_mu);
            Rccl* r = rccl();
            int nrc = r->GroupStart();
            for (size_t g = 0; g < G && !nrc; g++)
                nrc = r->AllGather(ctx[g]->x_send.p, ctx[g]->x_recv.p, block, kNcclUint8, m->nccl[g], ctx[g]->stream);
            const int erc = r->GroupEnd();
            if (nrc || erc) return nccl_error("ncclAllGather (multi-GPU store)", nrc ? nrc : erc);
            lists = root->x_recv.p;
        } else {
            for (size_t g = 1; g < G; g++)
                if (slot[g] != (size_t)-1) OTT_HIP(hipStreamWaitEvent(root->stream, ctx[g]->ev[6], 0));
        }
        // 3. the merge (src/meta.rs:699-709) of G x groups lists on the first shard's GPU, hits written straight into pinned host memory
        const size_t hits_bytes = (size_t)groups * KS * sizeof(ott_hit), cnt_bytes = (((size_t)groups * 8) + 63) & ~(size_t)63;
        if ((rc = root->h_hits.ensure(hits_bytes + cnt_bytes))) return rc;
        char* hh = (char*)root->h_hits.p;
        void* mapped = nullptr;
        OTT_HIP(hipHostGetDevicePointer(&mapped, hh, 0));
        if (timing) OTT_HIP(hipEventRecord(root->ev[0], root->stream));
        if ((rc = launch_merge_hits(root, (const ott_hit*)lists, (uint32_t)n_lists, groups, (uint32_t)KS, (uint32_t)k, E, d.take == OTT_TAKE_MAX,
                                    (ott_hit*)((char*)mapped + cnt_bytes), (uint64_t*)mapped)))
            return rc;
        if (timing) OTT_HIP(hipEventRecord(root->ev[1], root->stream));
        OTT_HIP(hipStreamSynchronize(root->stream));  // the one wait of the call (the batch path waits inside its levels)
        if (use_rccl) drain();  // the other GPUs' halves of the collective (long over: they finish together)
        const uint64_t* cnt = (const uint64_t*)hh;
        const ott_hit* hits = (const ott_hit*)(hh + cnt_bytes);
        uint64_t total = 0;
        for (uint32_t g = 0; g < groups; g++) {
            if (total + cnt[g] > cap) return fail(OTT_ERR_INVALID, "ott_query: output capacity is smaller than min(k, rows*nq)");
            if (cnt[g]) memcpy(out + total, hits + (size_t)g * KS, cnt[g] * sizeof(ott_hit));
            if (per && perq) per[g] = cnt[g];
            total += cnt[g];
        }
        if (n_out) *n_out = total;
        if (timing) {
            for (size_t g = 0; g < G; g++) {
                if (pending[g]) {
                    (void)use_device(m->shards[g]);
                    read_exact_events(ctx[g], &sst[g]);
                }
                add_stats(st, sst[g]);
            }
            (void)use_device(root);
            float ms_f = 0.f;
            if (hipEventElapsedTime(&ms_f, root->ev[0], root->ev[1]) == hipSuccess) st.merge_ns += (uint64_t)(ms_f * 1e6);
            if (hipEventElapsedTime(&ms_f, root->ev[2], root->ev[0]) == hipSuccess) st.exchange_ns = (uint64_t)(ms_f * 1e6);
        }
        return OTT_OK;
    }

    // after a failure, or RCCL: nothing of this call is left running on any shard's stream
    void drain() {
        for (size_t g = 0; g < G; g++) {
            (void)use_device(m->shards[g]);
            (void)hipStreamSynchronize(ctx[g]->stream);
        }
        (void)hipGetLastError();
    }

    // k > 512 (e.g. the reference's default take = every row, src/vec.rs:213): every shard's sorted list comes to the host
    // whole and the lists — each already in order — are merged there (src/meta.rs:699-709: concat, sort, truncate(k))
    int round_large(const ott_query_desc& d, uint64_t k, const CoreOpts& co, std::vector<ShardSlice>& sl, uint32_t groups, ott_hit* out, uint64_t cap,
                    uint64_t* n_out, uint64_t* per, ott_stats& st) {
        const bool perq = d.mode == OTT_MODE_PER_QUERY;
        const uint32_t nq = d.nq;
        std::vector<std::vector<ott_hit>> mine(G);
        std::vector<std::vector<uint64_t>> cnt(G, std::vector<uint64_t>(groups, 0));
        std::vector<ott_stats> sst(G);
        int rc = run_on_shards(ms, [&](size_t g) -> int {
            memset(&sst[g], 0, sizeof(ott_stats));
            if (sl[g].idle) return OTT_OK;
            OTT_HIP(use_device(m->shards[g]));
            const uint64_t n_g = store_rows(m->shards[g]);
            const uint64_t pool = perq ? n_g : n_g * (uint64_t)nq;
            const uint64_t k_loc = k < pool ? k : pool;
            mine[g].resize((size_t)(k_loc * (perq ? nq : 1)) + 1);
            std::vector<uint64_t> pq(nq, 0);
            uint64_t n_mine = 0;
            const int r = query_core(ctx[g], &sl[g].d, mine[g].data(), nullptr, mine[g].size(), &n_mine, pq.data(), nullptr, &sst[g], false, nullptr, co);
            if (r) return r;
            mine[g].resize((size_t)n_mine);
            if (perq) for (uint32_t q = 0; q < nq; q++) cnt[g][q] = pq[q];
            else cnt[g][0] = n_mine;
            return OTT_OK;
        });
        if (rc) return rc;
        for (size_t g = 0; g < G; g++) add_stats(st, sst[g]);
        const uint64_t tm0 = now_ns();
        const CanonLess less{d.take == OTT_TAKE_MAX, co.tie_sh, ms->base_offset - (co.tie_sh ? (co.tie_off & 7u) : 0u)};
        // the groups' extents in the shards' lists and in the output
        struct Group {
            std::vector<const ott_hit*> head, end;  // [G] this group's slice of every shard's list
            uint64_t keep = 0, at = 0;              // hits kept, first output slot
        };
        std::vector<Group> grp(groups);
        std::vector<size_t> off(G, 0);
        uint64_t total = 0;
        for (uint32_t gq = 0; gq < groups; gq++) {
            Group& gr = grp[gq];
            gr.head.resize(G);
            gr.end.resize(G);
            uint64_t have = 0;
            for (size_t g = 0; g < G; g++) {
                gr.head[g] = mine[g].data() + off[g];
                gr.end[g] = gr.head[g] + cnt[g][gq];
                off[g] += (size_t)cnt[g][gq];
                have += cnt[g][gq];
            }
            gr.keep = have < k ? have : k;
            gr.at = total;
            total += gr.keep;
            if (per && perq) per[gq] = gr.keep;
        }
        if (total > cap) return fail(OTT_ERR_INVALID, "ott_query: output capacity is smaller than min(k, rows*nq)");
        // G-way merge of sorted lists, outputs [lo, hi) of a group given where each list stands at output lo; equal keys cannot
        // occur across shards (different rows), so the order is total
        auto merge_range = [&](const Group& gr, std::vector<const ott_hit*> head, uint64_t lo, uint64_t hi) {
            merge_heads(head, gr.end, less, out + gr.at + lo, hi - lo);
        };
        // where every list stands when `t` hits of the group are out: the hit of global rank t is found by a binary search in
        // each list over the rank (= the sum over all lists of the hits in front of it) — exactly one hit has that rank
        auto cut_at = [&](const Group& gr, uint64_t t, std::vector<const ott_hit*>& pos) {
            pos = gr.head;
            if (t == 0) return;
            auto before = [&](size_t h, const ott_hit& x) { return (uint64_t)(std::lower_bound(gr.head[h], gr.end[h], x, less) - gr.head[h]); };
            auto rank_of = [&](const ott_hit& x) {
                uint64_t r = 0;
                for (size_t h = 0; h < G; h++) r += before(h, x);
                return r;
            };
            for (size_t g = 0; g < G; g++) {
                const ott_hit* lo = gr.head[g];
                const ott_hit* hi = gr.end[g];
                while (lo < hi) {  // first hit of list g whose rank is at least t
                    const ott_hit* mid = lo + (hi - lo) / 2;
                    if (rank_of(*mid) < t) lo = mid + 1;
                    else hi = mid;
                }
                if (lo != gr.end[g] && rank_of(*lo) == t) {
                    for (size_t h = 0; h < G; h++) pos[h] = gr.head[h] + before(h, *lo);
                    return;
                }
            }
            pos = gr.end;  // t = all the hits there are
        };
        // Large results (the reference's default take is every row): one host thread merged 1M hits of 8 shards in 15 ms, next
        // to 0.3 ms of GPU work.  The shard threads share it: few large groups are cut into G ranges of equal length by rank
        // (every thread finds its own two cuts), many small groups are dealt out whole.
        const bool parallel = G > 1 && total >= 32768;
        const bool split_groups = parallel && (groups < G || total / groups >= 65536);
        if (!parallel) {
            for (const Group& gr : grp) merge_range(gr, gr.head, 0, gr.keep);
        } else {
            m->pool->run_all([&](size_t part) {
                std::vector<const ott_hit*> pos;
                for (uint32_t gq = 0; gq < groups; gq++) {
                    const Group& gr = grp[gq];
                    if (!split_groups) {
                        if (gq % G == part) merge_range(gr, gr.head, 0, gr.keep);
                        continue;
                    }
                    const uint64_t lo = (uint64_t)((unsigned __int128)gr.keep * part / G), hi = (uint64_t)((unsigned __int128)gr.keep * (part + 1) / G);
                    if (hi <= lo) continue;
                    cut_at(gr, lo, pos);
                    merge_range(gr, pos, lo, hi);
                }
            });
        }
        if (n_out) *n_out = total;
        st.merge_ns += now_ns() - tm0;
        return OTT_OK;
    }
};

}  // namespace

namespace ott {

int multi_destroy(ott_store* ms) {
    ott_multi* m = ms->multi;
    {
        ott::host::ExclusiveLock wr(ms->rw);  // no query is running
    }
    delete m->pool;
    m->pool = nullptr;
    if (!m->nccl.empty()) {
        Rccl* r = rccl();
        for (size_t g = 0; g < m->nccl.size(); g++)
            if (m->nccl[g]) {
                (void)use_device(m->shards[g]);
                (void)hipStreamSynchronize(m->shards[g]->stream);
                (void)r->CommDestroy(m->nccl[g]);
            }
    }
    for (ott_store* s : m->shards) ott_store_destroy(s);
    delete m;
    ms->multi = nullptr;
    ms->stream = nullptr;  // (was the first shard's)
    delete ms;
    return OTT_OK;
}

int multi_reserve(ott_store* ms, uint64_t n_rows) {
    ott::host::ExclusiveLock wr(ms->rw);
    ott_multi* m = ms->multi;
    const size_t G = m->shards.size();
    if (n_rows <= m->plan_rows) return OTT_OK;
    const uint64_t total = std::max(n_rows, ms->n);
    const uint64_t gran = granule_of(ms->chunk_size);
    const std::vector<uint64_t> tgt = ideal_starts(total, gran, G, min_rows_of(ms));
    int rc = relayout(ms, tgt, total);  // (no row moves when the rows that exist already lie inside their new ranges)
    if (rc) return rc;
    // every shard pre-sizes its range
    for (size_t g = 0; g < G; g++) {
        const uint64_t hi = g + 1 < G ? tgt[g + 1] : total;
        const uint64_t want = hi > tgt[g] ? hi - tgt[g] : 0;
        if (want > m->shards[g]->cap && (rc = ott_store_reserve(m->shards[g], want))) return rc;
    }
    return OTT_OK;
}

int multi_append(ott_store* ms, const AppendArgs& a, uint64_t n_rows) {
    ott::host::ExclusiveLock wr(ms->rw);
    ott_multi* m = ms->multi;
    const size_t G = m->shards.size();
    // the pieces: rows fill the planned ranges in order; past the plan they go to the last shard that has begun
    std::vector<Piece> pieces;
    std::vector<uint64_t> start = m->start;
    uint64_t p = ms->n, left = n_rows;
    while (left) {
        size_t g = 0;
        for (size_t i = 1; i < G; i++)
            if (start[i] != NOT_YET && start[i] <= p) g = i;
        uint64_t room = left;
        for (size_t i = g + 1; i < G; i++)
            if (start[i] != NOT_YET) {
                room = std::min<uint64_t>(left, start[i] - p);
                break;
            }
        const uint64_t s0 = start[g] == NOT_YET ? p : start[g];
        pieces.push_back({g, p, p - s0, room});
        p += room;
        left -= room;
    }
    // a shard that receives its first rows is told its base first (the generators are keyed by the global row)
    for (const Piece& pc : pieces) m->shards[pc.g]->base_offset = ms->base_offset + (m->start[pc.g] == NOT_YET ? pc.first_global : m->start[pc.g]);
    const uint64_t first = ms->n;
    const int root_dev = m->devs[0];
    const std::function<int(size_t)> on_shard = [&](size_t g) -> int {
        for (const Piece& pc : pieces) {
            if (pc.g != g) continue;
            ott_store* s = m->shards[g];
            if (store_rows(s) != pc.first_local) return fail(OTT_ERR_INVALID, "multi-GPU store: internal layout error (a shard is not filled up to the piece it receives)");
            const uint64_t off = pc.first_global - first;
            int r = OTT_OK;
            switch (a.kind) {
                case APPEND_HOST:
                    r = ott_store_append(s, (const float*)a.rows + off * ms->dim, pc.count);
                    break;
                case APPEND_DEVICE: {
                    const float* src = (const float*)a.rows + off * ms->dim;  // device memory of the FIRST shard's GPU
                    if (m->logical[g] == m->logical[0]) {
                        r = ott_store_append_device(s, src, pc.count);
                    } else {
                        OTT_HIP(use_device(m->shards[g]));
                        void* tmp = nullptr;
                        OTT_HIP(hipMalloc(&tmp, pc.count * ms->dim * sizeof(float)));
                        hipError_t e = hipMemcpyPeer(tmp, m->devs[g], src, root_dev, pc.count * ms->dim * sizeof(float));
                        if (e == hipSuccess) r = ott_store_append_device(s, tmp, pc.count);
                        (void)hipFree(tmp);
                        if (e != hipSuccess) return fail(OTT_ERR_HIP, std::string("hipMemcpyPeer: ") + hipGetErrorString(e));
                    }
                    break;
                }
                case APPEND_RANDOM:
                    r = ott_store_append_random(s, pc.count, a.seed);
                    break;
                default:
                    r = ott_store_append_clustered(s, pc.count, a.seed, a.n_clusters, a.spread, a.aniso);
                    break;
            }
            if (r) return r;
        }
        return OTT_OK;
    };
    // rows for ONE shard (VecStore::add_vector is a row per call, src/vec.rs:357-371): on the calling thread — waking the
    // other shards' threads for nothing cost 5-9 us per call (197k / 111k rows/s on 4 / 8 shards against 500k on one store)
    bool one_shard = !pieces.empty();
    for (const Piece& pc : pieces) one_shard = one_shard && pc.g == pieces[0].g;
    int rc;
    if (one_shard) {
        const size_t g = pieces[0].g;
        rc = on_shard(g);
        if (rc) rc = fail(rc, "shard " + std::to_string(g) + " (device " + std::to_string(m->devs[g]) + "): " + ott_last_error());
    } else {
        rc = run_on_shards(ms, on_shard);
    }
    // what arrived stays (like a failing try_for_each of the reference's add_vectors: rows before the failure are kept), as
    // long as the shards still tile a contiguous range
    uint64_t got = 0;
    for (const Piece& pc : pieces) {
        ott_store* s = m->shards[pc.g];
        if (store_rows(s) != pc.first_local + pc.count) break;
        if (m->start[pc.g] == NOT_YET) m->start[pc.g] = pc.first_global;
        got += pc.count;
    }
    ms->n = first + got;
    // a later piece that arrived although an earlier one failed would leave a hole: it is dropped again
    if (got != n_rows) {
        uint64_t expect = first + got;
        for (const Piece& pc : pieces) {
            ott_store* s = m->shards[pc.g];
            if (pc.first_global >= expect && store_rows(s) > pc.first_local) {
                (void)store_flush(s);
                if (s->n > pc.first_local) s->n = pc.first_local;
                s->pend.rows.store(0);
            }
        }
    }
    m->layout_dirty = true;
    ms->evalmask_bits = 0;
    return rc;
}

int multi_write_rows(ott_store* ms, uint64_t first_row, const float* rows_host, uint64_t n_rows) {
    if (n_rows == 0) return OTT_OK;
    if (!rows_host) return fail(OTT_ERR_INVALID, "ott_store_write_rows: rows is NULL");
    ott::host::ExclusiveLock wr(ms->rw);
    if (first_row + n_rows > ms->n) return fail(OTT_ERR_INVALID, "ott_store_write_rows: range exceeds store length");
    for (const Piece& pc : pieces_of(ms, first_row, n_rows)) {
        const int rc = ott_store_write_rows(ms->multi->shards[pc.g], pc.first_local, rows_host + (pc.first_global - first_row) * ms->dim, pc.count);
        if (rc) return rc;
    }
    return OTT_OK;
}

int multi_read(const ott_store* cms, bool inv_norms, uint64_t first_row, uint64_t n_rows, float* out_host) {
    ott_store* ms = const_cast<ott_store*>(cms);
    ott::host::SharedLock rd;
    // (staged rows go to their GPUs under the store's EXCLUSIVE lock: a shard flushing under a reader's shared lock could
    // reallocate its rows under a query that runs beside it)
    const int rcl = lock_clean(ms, rd);
    if (rcl) return rcl;
    if (first_row + n_rows > ms->n) return fail(OTT_ERR_INVALID, "ott_store_read_rows: range exceeds store length");
    for (const Piece& pc : pieces_of(ms, first_row, n_rows)) {
        const ott_store* s = ms->multi->shards[pc.g];
        const uint64_t off = pc.first_global - first_row;
        const int rc = inv_norms ? ott_store_read_inv_norms(s, pc.first_local, pc.count, out_host + off)
                                 : ott_store_read_rows(s, pc.first_local, pc.count, out_host + off * ms->dim);
        if (rc) return rc;
    }
    return OTT_OK;
}

int multi_set_chunk_size(ott_store* ms, uint64_t chunk_size) {
    ott::host::ExclusiveLock wr(ms->rw);
    const uint64_t cs = chunk_size < 1 ? 1 : chunk_size;  // src/meta.rs:86-89
    const uint64_t old = ms->chunk_size;
    ms->chunk_size = cs;
    for (ott_store* s : ms->multi->shards) s->chunk_size = cs;
    // chunks must not straddle GPUs: when the rows that exist no longer start on granule boundaries they are moved now
    if (ms->n && !layout_aligned(ms, granule_of(cs))) {
        const int rc = ensure_layout(ms, true);
        if (rc) {
            ms->chunk_size = old;
            for (ott_store* s : ms->multi->shards) s->chunk_size = old;
            return rc;
        }
    } else if (ms->n == 0 && ms->multi->plan_rows) {
        ms->multi->start = ideal_starts(ms->multi->plan_rows, granule_of(cs), n_shards(ms), min_rows_of(ms));  // an empty plan follows the new granule
        mark_unused(ms->multi->start, ms->multi->plan_rows, granule_of(cs));
        set_shard_bases(ms);
    }
    return OTT_OK;
}

int multi_set_base_offset(ott_store* ms, uint64_t base) {
    ott::host::ExclusiveLock wr(ms->rw);
    ms->base_offset = base;
    set_shard_bases(ms);
    return OTT_OK;
}

int multi_set_reduce_order(ott_store* ms, uint32_t reduce) {
    ott::host::ExclusiveLock wr(ms->rw);
    ms->reduce = reduce;
    for (ott_store* s : ms->multi->shards) s->reduce = reduce;
    return OTT_OK;
}

int multi_set_batch_image(ott_store* ms, int enabled) {
    ott::host::ExclusiveLock wr(ms->rw);
    for (ott_store* s : ms->multi->shards) {
        const int rc = ott_store_set_batch_image(s, enabled);
        if (rc) return rc;
    }
    return OTT_OK;
}

int multi_set_option(ott_store* ms, const char* name, int64_t value) {
    ott::host::ExclusiveLock wr(ms->rw);
    Options o = ms->opt;
    if (option_set(o, name, (long long)value)) return fail(OTT_ERR_INVALID, std::string("ott_store_set_option: unknown option or bad value: ") + name);
    if (std::string(name) == "multi_transport" && ms->multi->transport != 0 && o.multi_transport != ms->opt.multi_transport)
        return fail(OTT_ERR_INVALID, "ott_store_set_option: multi_transport is chosen before the first query of a multi-GPU store");
    if (o.multi_fake_distinct != ms->opt.multi_fake_distinct)
        return fail(OTT_ERR_INVALID, "ott_store_set_option: multi_fake_distinct is read when the store is created (OTT_MULTI_FAKE_DISTINCT=1)");
    ms->opt = o;
    for (ott_store* s : ms->multi->shards) {
        const int rc = ott_store_set_option(s, name, value);
        if (rc) return rc;
    }
    return OTT_OK;
}

int multi_prepare_batch(ott_store* ms) {
    ott::host::SharedLock rd;
    const int rcl = lock_clean(ms, rd);
    if (rcl) return rcl;
    return run_on_shards(ms, [&](size_t g) -> int { return ott_store_prepare_batch(ms->multi->shards[g]); });
}

int multi_batch_ready(ott_store* ms) {
    ott::host::SharedLock rd(ms->rw);
    for (ott_store* s : ms->multi->shards)
        if (store_rows(s) && !ott_store_batch_ready(s)) return 0;
    return ms->n ? 1 : 0;
}

int multi_sync(ott_store* ms) {
    for (ott_store* s : ms->multi->shards) {
        const int rc = ott_store_sync(s);
        if (rc) return rc;
    }
    return OTT_OK;
}

int multi_add_column(ott_store* ms, uint32_t dtype, const void* values_host, const uint64_t* nulls, uint64_t n, uint32_t* out_column_id) {
    size_t esz;
    switch (dtype) {
        case OTT_DT_INT32: case OTT_DT_FLOAT32: esz = 4; break;
        case OTT_DT_INT64: case OTT_DT_FLOAT64: case OTT_DT_DATETIME: esz = 8; break;
        default: return fail(OTT_ERR_INVALID, "ott_store_add_column: only numeric / datetime columns live on the GPU");
    }
    ott::host::ExclusiveLock wr(ms->rw);
    if (n != ms->n) return fail(OTT_ERR_INVALID, "ott_store_add_column: column length does not match the store length");
    if (n && !values_host) return fail(OTT_ERR_INVALID, "ott_store_add_column: values is NULL");
    ott_multi* m = ms->multi;
    int rc;
    if (m->layout_dirty) {  // last chance to balance: columns pin the rows
        if ((rc = ensure_layout(ms, false))) return rc;
        m->layout_dirty = false;
    }
    const size_t G = m->shards.size();
    std::vector<uint32_t> ids(G, 0);
    rc = run_on_shards(ms, [&](size_t g) -> int {
        ott_store* s = m->shards[g];
        const uint64_t s0 = start_of(ms, g), n_g = store_rows(s);
        std::vector<uint64_t> nl;
        if (nulls && n_g) slice_bits_bytes(nulls, s0, n_g, nl);
        return ott_store_add_column(s, dtype, n_g ? (const char*)values_host + s0 * esz : nullptr, (nulls && n_g) ? nl.data() : nullptr, n_g, &ids[g]);
    });
    if (rc) return rc;
    for (size_t g = 1; g < G; g++)
        if (ids[g] != ids[0]) return fail(OTT_ERR_INVALID, "multi-GPU store: internal error (column ids differ between shards)");
    *out_column_id = ids[0];
    return OTT_OK;
}

int multi_eval_row_mask(ott_store* ms, const ott_leaf* leaves, uint32_t n_leaves, uint32_t n_clauses, uint64_t* out_host) {
    ott::host::ExclusiveLock wr(ms->rw);
    ott_multi* m = ms->multi;
    const size_t G = m->shards.size();
    ms->evalmask_bits = 0;
    std::vector<std::vector<uint64_t>> part(G);
    int rc = run_on_shards(ms, [&](size_t g) -> int {
        ott_store* s = m->shards[g];
        if (out_host) part[g].assign((size_t)((store_rows(s) + 63) / 64) + 1, 0);
        return ott_store_eval_row_mask(s, leaves, n_leaves, n_clauses, out_host ? part[g].data() : nullptr);
    });
    if (rc) return rc;
    if (out_host) {
        memset(out_host, 0, (size_t)((ms->n + 63) / 64) * 8);
        for (size_t g = 0; g < G; g++) {
            const uint64_t n_g = store_rows(m->shards[g]);
            if (n_g) memcpy((uint8_t*)out_host + start_of(ms, g) / 8, part[g].data(), (size_t)((n_g + 7) / 8));  // shards start on multiples of 8 rows
        }
    }
    ms->evalmask_bits = ms->n;
    return OTT_OK;
}

int multi_zone_stats(ott_store* ms, uint32_t column, uint64_t chunk_size, void* out_min, void* out_max, uint64_t* out_non_null) {
    ott::host::ExclusiveLock wr(ms->rw);
    ott_multi* m = ms->multi;
    const size_t G = m->shards.size();
    for (size_t g = 1; g < G; g++)
        if (store_rows(m->shards[g]) && m->start[g] % chunk_size != 0)
            return fail(OTT_ERR_UNSUPPORTED, "ott_store_zone_stats on a multi-GPU store: chunk_size must divide the shard boundaries (use the store's chunk size)");
    return run_on_shards(ms, [&](size_t g) -> int {
        ott_store* s = m->shards[g];
        if (!store_rows(s)) return OTT_OK;
        const uint64_t c0 = start_of(ms, g) / chunk_size;
        return ott_store_zone_stats(s, column, chunk_size, (char*)out_min + c0 * 8, (char*)out_max + c0 * 8, out_non_null + c0);
    });
}

int multi_query(ott_store* ms, const ott_query_desc* d, ott_hit* out, uint64_t cap, uint64_t* n_out, uint64_t* n_per_query, ott_stats* stats) {
    int rc = validate_query(ms, d);
    if (rc) return rc;
    ott_multi* m = ms->multi;
    const uint64_t t0 = now_ns();
    // rows appended since the last look: staged ones go to their GPUs, the shards are balanced — and the store is taken shared
    // only once nothing of that is left to do (an append can slip in between the two locks)
    ott::host::SharedLock rd;
    if ((rc = lock_clean(ms, rd))) return rc;
    const int tie_order = ms->opt.tie_order;
    if (n_out) *n_out = 0;
    if (n_per_query)
        for (uint32_t i = 0; i < d->nq; i++) n_per_query[i] = 0;
    const bool perq = d->mode == OTT_MODE_PER_QUERY;
    const uint64_t pool = perq ? ms->n : ms->n * (uint64_t)d->nq;
    const uint64_t k_eff = d->k < pool ? d->k : pool;
    if (cap < (perq ? k_eff * d->nq : k_eff)) return fail(OTT_ERR_INVALID, "ott_query: output capacity is smaller than min(k, rows*nq)");
    ott_stats st;
    memset(&st, 0, sizeof(st));
    if (ms->n == 0 || k_eff == 0) {
        if (stats) *stats = st;
        return OTT_OK;
    }
    if (ms->opt.multi_transport != 2) {
        // ONE shard holds every row (a one-device list, or a store too small to be worth spreading: option multi_min_shard_rows):
        // nothing to fan out, exchange or merge — the shard's own query (options, tie order included, are its own; its rows
        // start at the store's first row, so masks and indices need no translation)
        for (ott_store* sh : m->shards)
            if (store_rows(sh) == ms->n) return ott_query(sh, d, out, cap, n_out, n_per_query, stats);
    }
    MultiCall mc(ms);
    if (tie_order == 0) {
        rc = mc.round(*d, k_eff, CoreOpts{}, out, cap, n_out, n_per_query, stats ? &st : nullptr);
    } else {
        // the reference's outcome at exact score ties: the decision logic of ott_ties.hip over candidates from all shards
        TieEnv env;
        env.tmax = d->take == OTT_TAKE_MAX;
        env.base = ms->base_offset;
        env.chunk_size = ms->chunk_size;
        env.dim = ms->dim;
        const auto run_off = [&mc, ms](const ott_query_desc& dd, uint64_t k, bool flat, uint32_t tie_off, std::vector<ott_hit>& o, std::vector<uint64_t>& per,
                                       ott_stats* st2) -> int {
            ott_query_desc d2 = dd;
            if (flat) d2.path = OTT_PATH_EXACT;
            const bool pq = dd.mode == OTT_MODE_PER_QUERY;
            const uint64_t pl = pq ? ms->n : ms->n * (uint64_t)dd.nq;
            const uint64_t ke = k < pl ? k : pl;
            o.resize((size_t)(pq ? ke * dd.nq : ke) + 1);
            per.assign(dd.nq, 0);
            uint64_t n2 = 0;
            CoreOpts co;
            co.tie_sh = 3;
            co.flat = flat;
            co.tie_off = tie_off;
            const int r = mc.round(d2, ke, co, o.data(), o.size(), &n2, per.data(), st2);
            if (r) return r;
            o.resize((size_t)n2);
            return OTT_OK;
        };
        env.run = [&run_off](const ott_query_desc& dd, uint64_t k, bool flat, std::vector<ott_hit>& o, std::vector<uint64_t>& per, ott_stats* st2) -> int {
            return run_off(dd, k, flat, 0, o, per, st2);
        };
        env.run_chunk = [&run_off, ms](uint64_t chunk, const ott_query_desc& dd, uint64_t k, bool flat, std::vector<ott_hit>& o, std::vector<uint64_t>& per,
                                   ott_stats* st2) -> int {
            const uint64_t n_chunks = (ms->n + ms->chunk_size - 1) / ms->chunk_size;
            std::vector<uint64_t> mask((size_t)((n_chunks + 63) / 64) + 1, 0);
            mask[(size_t)(chunk >> 6)] = 1ull << (chunk & 63);
            ott_query_desc d3 = dd;
            d3.chunk_mask = mask.data();
            // the chunk's 8-row blocks are counted from its first row; shards start on multiples of lcm(chunk size, 8), so the
            // offset is the same in the owning shard's local rows
            const uint32_t off = (uint32_t)((8 - (chunk * ms->chunk_size) % 8) % 8);
            return run_off(d3, k, flat, off, o, per, st2);
        };
        rc = ref_ties_collect(env, tie_order, d, out, cap, n_out, n_per_query, stats ? &st : nullptr);
    }
    if (rc) return rc;
    st.total_ns = now_ns() - t0;
    if (stats) *stats = st;
    return OTT_OK;
}

}  // namespace ott

extern "C" {

int ott_store_create_multi(uint32_t dim, uint32_t n_dev, const int* dev_ids, ott_store** out) {
    if (!out) return fail(OTT_ERR_INVALID, "ott_store_create_multi: out is NULL");
    *out = nullptr;
    if (dim == 0) return fail(OTT_ERR_INVALID, "ott_store_create_multi: dim must be > 0");
    if (n_dev == 0 || n_dev > 64 || !dev_ids) return fail(OTT_ERR_INVALID, "ott_store_create_multi: 1 to 64 device ordinals are needed");
    ott_store* ms = new ott_store();
    ott_multi* m = new ott_multi();
    ms->multi = m;
    ms->dim = dim;
    ms->ld = (dim + 3u) & ~3u;
    ms->dimq = (dim + 7u) & ~7u;
    ms->device = dev_ids[0];
    options_from_env(ms->opt);
    m->devs.assign(dev_ids, dev_ids + n_dev);
    // logical device ids: the ordinals — or, under the test option multi_fake_distinct (OTT_MULTI_FAKE_DISTINCT=1), one id per
    // shard: every place that asks "is this shard on the merging GPU?" then answers no, and the exchange, the row moves and
    // device appends take the code paths of distinct GPUs although the copies stay on the box's one GPU
    m->logical.assign(dev_ids, dev_ids + n_dev);
    if (ms->opt.multi_fake_distinct)
        for (uint32_t g = 0; g < n_dev; g++) m->logical[g] = 1000 + (int)g;
    ms->logical = m->logical[0];
    m->distinct = true;
    for (uint32_t i = 0; i < n_dev; i++)
        for (uint32_t j = i + 1; j < n_dev; j++)
            if (m->logical[i] == m->logical[j]) m->distinct = false;
    m->start.assign(n_dev, NOT_YET);
    m->start[0] = 0;
    for (uint32_t g = 0; g < n_dev; g++) {
        ott_store* s = nullptr;
        const int rc = store_create(dim, dev_ids[g], m->logical[g], &s);
        if (rc) {
            for (ott_store* x : m->shards) ott_store_destroy(x);
            delete m;
            ms->multi = nullptr;
            delete ms;
            return rc;
        }
        m->shards.push_back(s);
    }
    // distinct devices: direct peer access where the hardware offers it (xGMI), so that the block copies and the row moves go
    // from GPU to GPU instead of through host memory.  Best effort: without it hipMemcpyPeerAsync still works, staged.
    for (uint32_t a = 0; a < n_dev; a++)
        for (uint32_t b = 0; b < n_dev; b++) {
            if (dev_ids[a] == dev_ids[b]) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, dev_ids[a], dev_ids[b]) != hipSuccess || !can) continue;
            if (use_device_raw(dev_ids[a], dev_ids[a]) == hipSuccess) (void)hipDeviceEnablePeerAccess(dev_ids[b], 0);  // (already enabled: an error to ignore)
            (void)hipGetLastError();
        }
    (void)hipGetLastError();
    ms->n_cu = m->shards[0]->n_cu;
    ms->stream = m->shards[0]->stream;  // ott_store_stream: the merging shard's
    m->pool = new ShardPool(n_dev);
    *out = ms;
    return OTT_OK;
}

int ott_multi_plan(uint64_t n_rows, uint64_t chunk_size, uint32_t n_dev, uint64_t* out_first_rows) {
    if (!out_first_rows || n_dev == 0) return fail(OTT_ERR_INVALID, "ott_multi_plan: NULL output or no devices");
    const uint64_t cs = chunk_size < 1 ? 1 : chunk_size;
    Options o;
    options_from_env(o);  // (OTT_MULTI_MIN_SHARD_ROWS, like a store created now)
    const std::vector<uint64_t> st = ideal_starts(n_rows, granule_of(cs), n_dev, (uint64_t)o.multi_min_shard_rows);
    for (uint32_t g = 0; g < n_dev; g++) out_first_rows[g] = st[g] < n_rows ? st[g] : n_rows;
    return OTT_OK;
}

int ott_store_shard_count(const ott_store* s) {
    if (!s) return 0;
    return s->multi ? (int)s->multi->shards.size() : 1;
}

int ott_store_shard_info(const ott_store* s, uint32_t shard, int* device, uint64_t* first_row, uint64_t* n_rows) {
    if (!s) return fail(OTT_ERR_INVALID, "ott_store_shard_info: store is NULL");
    if (!s->multi) {
        if (shard != 0) return fail(OTT_ERR_INVALID, "ott_store_shard_info: no such shard");
        if (device) *device = s->device;
        if (first_row) *first_row = 0;
        if (n_rows) *n_rows = store_rows(s);
        return OTT_OK;
    }
    ott::host::SharedLock rd(const_cast<ott_store*>(s)->rw);
    if (shard >= s->multi->shards.size()) return fail(OTT_ERR_INVALID, "ott_store_shard_info: no such shard");
    if (device) *device = s->multi->devs[shard];
    if (first_row) *first_row = start_of(s, shard);
    if (n_rows) *n_rows = store_rows(s->multi->shards[shard]);
    return OTT_OK;
}

const char* ott_store_transport(const ott_store* s) {
    if (!s || !s->multi) return "none";
    return s->multi->transport == 2 ? "rccl" : s->multi->transport == 1 ? "peer" : "undecided";
}

}  // extern "C"

// ott_comm.hip — the one exchange of a sharded query, behind the C ABI.
//
// The reference fans chunks out over a rayon pool and concat-sort-truncates the per-chunk top-k lists
// (src/meta.rs:678-709).  Across GPUs: every rank scores its shard (ott_api.hip, no collective), then ONE all-gather of
// fixed-size sentinel-padded candidate blocks and the same merge kernel on every rank (merge_hits_kernel, ott_exact.hip).
// Transport: RCCL (ncclAllGather over xGMI, queued on the query context's stream right behind the scoring and merge
// kernels: score -> gather -> merge without a host synchronisation in between) or a host callback (blocks staged through
// pinned memory; for hosts that bring their own transport and for two test ranks on one GPU).  librccl is dlopen'ed on
// first use, so the library loads — and every single-GPU entry point works — on a machine without it.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "ott_internal.h"

using namespace ott;

namespace ott {
Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // by soname first: a host process that already carries an RCCL (e.g. the one bundled with PyTorch-ROCm, paired with
        // its own HIP runtime) gets that one
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
        }
        if (!r.handle) {
            const char* e = dlerror();
            r.why = std::string("librccl.so.1 could not be loaded: ") + (e ? e : "?");
            return;
        }
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.handle, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.handle, "ncclCommInitRank");
        r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.handle, "ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
        r.CommCount = (decltype(r.CommCount))dlsym(r.handle, "ncclCommCount");
        r.GetVersion = (decltype(r.GetVersion))dlsym(r.handle, "ncclGetVersion");
        r.GroupStart = (decltype(r.GroupStart))dlsym(r.handle, "ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.handle, "ncclGroupEnd");
        r.AllGather = (decltype(r.AllGather))dlsym(r.handle, "ncclAllGather");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.handle, "ncclGetErrorString");
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather) {
            r.why = "librccl.so.1 lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather";
            r.handle = nullptr;
        }
    });
    return &r;
}
}  // namespace ott

namespace {

int nccl_fail(const char* what, int code) {
    Rccl* r = rccl();
    const char* msg = (r->GetErrorString && code) ? r->GetErrorString(code) : "?";
    return fail(OTT_ERR_HIP, std::string(what) + ": " + msg);
}

uint64_t now_ns() {
    return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

struct ott_comm {
    int rank = 0, world = 1, device = -1;
    bool is_rccl = false;
    void* nccl = nullptr;
    ott_allgather_fn fn = nullptr;
    void* user = nullptr;
    hipStream_t stream = nullptr;  // RCCL transport: control-data gathers (ott_comm_all_gather_host)
    DevBuf d_send, d_recv;
    PinBuf h_send, h_recv;         // HOST transport staging of candidate blocks
    std::mutex mu;                 // one collective at a time per comm (every rank must issue them in the same order anyway)
    // RCCL transport, world > 1: how long a collective may stay unfinished before the call gives up with an error instead of
    // waiting for a peer that died (0 = wait for ever).  OTT_COMM_TIMEOUT_MS presets it when the comm is created.
    int64_t timeout_ms = 120000;
};

namespace {

// Waits for `stream` like hipStreamSynchronize, but no longer than the comm's timeout when a peer could be missing (RCCL
// transport with world > 1: a collective whose peer never joins stays queued for ever).  The first 100 ms spin on
// hipStreamQuery like hipStreamSynchronize does (a sharded query is a few ms of scoring + a latency-bound exchange: a sleep's
// wake-up jitter would cost the N-GPU run a percent or two per step), after that the thread sleeps in 50 us steps.  On a timeout the stream is left as it is (the work cannot be recalled): the comm and the store are to be
// destroyed, which is what a job that lost a rank does anyway.
int wait_stream(ott_comm* c, hipStream_t stream, const char* what) {
    if (!c->is_rccl || c->world <= 1 || c->timeout_ms <= 0) {
        OTT_HIP(hipStreamSynchronize(stream));
        return OTT_OK;
    }
    const uint64_t t0 = now_ns(), limit = (uint64_t)c->timeout_ms * 1000000ull;
    for (;;) {
        const hipError_t e = hipStreamQuery(stream);
        if (e == hipSuccess) return OTT_OK;
        if (e != hipErrorNotReady) return fail(OTT_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
        const uint64_t waited = now_ns() - t0;
        if (waited > limit)
            return fail(OTT_ERR_HIP, std::string(what) + ": the exchange did not complete within " + std::to_string(c->timeout_ms) +
                                         " ms (rank " + std::to_string(c->rank) + " of " + std::to_string(c->world) +
                                         "): a peer did not arrive at the collective (dead rank, or ranks calling in a different order)");
        if (waited > 100000000ull) std::this_thread::sleep_for(std::chrono::microseconds(50));  // (a healthy exchange is over long before)
    }
}

// all-gather of `bytes` per rank between DEVICE buffers, queued on `stream` (RCCL) or staged through the host callback
// (then `stream` is drained first and the gathered block is copied back asynchronously)
int gather_device(ott_comm* c, const void* send_dev, void* recv_dev, size_t bytes, hipStream_t stream) {
    if (c->is_rccl) {
        const int rc = rccl()->AllGather(send_dev, recv_dev, bytes, kNcclUint8, c->nccl, stream);
        if (rc) return nccl_fail("ncclAllGather", rc);
        return OTT_OK;
    }
    int e;
    if ((e = c->h_send.ensure(bytes))) return e;
    if ((e = c->h_recv.ensure(bytes * (size_t)c->world))) return e;
    OTT_HIP(hipMemcpyAsync(c->h_send.p, send_dev, bytes, hipMemcpyDeviceToHost, stream));
    OTT_HIP(hipStreamSynchronize(stream));
    if (c->fn(c->user, c->h_send.p, c->h_recv.p, (uint64_t)bytes)) return fail(OTT_ERR_HIP, "ott_comm: the host all-gather callback failed");
    OTT_HIP(hipMemcpyAsync(recv_dev, c->h_recv.p, bytes * (size_t)c->world, hipMemcpyHostToDevice, stream));
    return OTT_OK;
}

int gather_host_locked(ott_comm* c, const void* send, void* recv, uint64_t bytes) {
    if (bytes == 0) return OTT_OK;
    if (!c->is_rccl) {
        if (c->fn(c->user, send, recv, bytes)) return fail(OTT_ERR_HIP, "ott_comm: the host all-gather callback failed");
        return OTT_OK;
    }
    int e;
    OTT_HIP(hipSetDevice(c->device));
    if ((e = c->d_send.ensure(bytes))) return e;
    if ((e = c->d_recv.ensure(bytes * (size_t)c->world))) return e;
    OTT_HIP(hipMemcpyAsync(c->d_send.p, send, bytes, hipMemcpyHostToDevice, c->stream));
    const int rc = rccl()->AllGather(c->d_send.p, c->d_recv.p, bytes, kNcclUint8, c->nccl, c->stream);
    if (rc) return nccl_fail("ncclAllGather", rc);
    OTT_HIP(hipMemcpyAsync(recv, c->d_recv.p, bytes * (size_t)c->world, hipMemcpyDeviceToHost, c->stream));
    return wait_stream(c, c->stream, "ott_comm_all_gather_host");
}

// k > 512: every rank's sorted list travels whole.  counts first (per group), then the lists padded to the longest, then
// a host merge in the canonical order (src/meta.rs:699-709: concat, sort, truncate(k)).
int sharded_large_k(ott_store* ctx, ott_comm* c, const ott_query_desc* d, ott_hit* out, uint64_t cap, uint64_t* n_out, uint64_t* n_per_query,
                    ott_stats* stats, const CoreOpts& co) {
    const bool perq = d->mode == OTT_MODE_PER_QUERY;
    const uint32_t nq = d->nq, groups = perq ? nq : 1u;
    const uint64_t pool = perq ? ctx->n : ctx->n * (uint64_t)nq;
    const uint64_t k_loc = d->k < pool ? d->k : pool;
    std::vector<ott_hit> mine((size_t)(k_loc * (perq ? nq : 1)) + 1);
    std::vector<uint64_t> cnt_mine(groups, 0), per(nq, 0);
    uint64_t n_mine = 0;
    int rc = OTT_OK;
    // (co.tie_sh = 3: every shard ranks its candidates in the reference's visit order and the cross-shard merge keeps it —
    // shards are in row order and start on 8-row block boundaries)
    if (ctx->n) rc = query_core(ctx, d, mine.data(), nullptr, mine.size(), &n_mine, per.data(), nullptr, stats, false, nullptr, co);
    else if (stats) memset(stats, 0, sizeof(*stats));
    // a rank that failed still joins the collectives (with nothing), so the others do not hang; its error is returned after
    const int rc_local = rc;
    if (rc_local) n_mine = 0;
    if (perq) for (uint32_t q = 0; q < nq; q++) cnt_mine[q] = rc_local ? 0 : per[q];
    else cnt_mine[0] = n_mine;
    std::vector<uint64_t> cnt_all((size_t)groups * c->world);
    if ((rc = gather_host_locked(c, cnt_mine.data(), cnt_all.data(), (uint64_t)groups * 8))) return rc;
    uint64_t longest = 0;
    for (int r = 0; r < c->world; r++) {
        uint64_t t = 0;
        for (uint32_t g = 0; g < groups; g++) t += cnt_all[(size_t)r * groups + g];
        longest = t > longest ? t : longest;
    }
    std::vector<ott_hit> all;
    if (longest) {
        mine.resize((size_t)longest);
        all.resize((size_t)longest * c->world);
        if ((rc = gather_host_locked(c, mine.data(), all.data(), longest * sizeof(ott_hit)))) return rc;
    }
    if (rc_local) return rc_local;
    const CanonLess less{d->take == OTT_TAKE_MAX, co.tie_sh, 0};
    uint64_t total = 0;
    std::vector<size_t> off((size_t)c->world, 0);  // per rank: where the next group starts in its list
    std::vector<ott_hit> grp;
    for (uint32_t g = 0; g < groups; g++) {
        grp.clear();
        for (int r = 0; r < c->world; r++) {
            const uint64_t n = cnt_all[(size_t)r * groups + g];
            const ott_hit* src = all.data() + (size_t)r * longest + off[r];
            grp.insert(grp.end(), src, src + n);
            off[r] += (size_t)n;
        }
        const size_t keep = grp.size() < d->k ? grp.size() : (size_t)d->k;
        std::partial_sort(grp.begin(), grp.begin() + keep, grp.end(), less);
        if (total + keep > cap) return fail(OTT_ERR_INVALID, "ott_query_sharded: output capacity is smaller than the result");
        if (keep) memcpy(out + total, grp.data(), keep * sizeof(ott_hit));
        if (n_per_query && perq) n_per_query[g] = keep;
        total += keep;
    }
    if (n_out) *n_out = total;
    return OTT_OK;
}

int sharded_on(ott_store* ctx, ott_comm* c, const ott_query_desc* d, ott_hit* out, uint64_t cap, uint64_t* n_out, uint64_t* n_per_query,
               ott_stats* stats_out, const CoreOpts& co) {
    int rc;
    OTT_HIP(hipSetDevice(ctx->device));
    const uint64_t t0 = now_ns();
    const bool perq = d->mode == OTT_MODE_PER_QUERY;
    const uint32_t nq = d->nq, groups = perq ? nq : 1u;
    if (n_out) *n_out = 0;
    if (n_per_query)
        for (uint32_t i = 0; i < nq; i++) n_per_query[i] = 0;
    ott_stats st;
    memset(&st, 0, sizeof(st));
    if (d->k == 0) {  // src/vec_compute.rs:174; every rank sees the same k, so all skip the exchange together
        if (stats_out) *stats_out = st;
        return OTT_OK;
    }
    if (d->k > 512) {
        rc = sharded_large_k(ctx, c, d, out, cap, n_out, n_per_query, &st, co);
        st.total_ns = now_ns() - t0;
        if (stats_out) *stats_out = st;
        return rc;
    }
    // block geometry from k alone (every rank must agree): [groups][KS] slots, KS = the register list width for k
    const int E = list_E(d->k);
    const uint64_t KS = 64ull * (uint64_t)E;
    const size_t block = (size_t)groups * KS * sizeof(ott_hit);
    // A failure that only THIS rank sees must not keep it away from the exchange (its peers would sit in ncclAllGather until
    // their comm timeout): it is remembered, the rank contributes a block of sentinels, and the error is returned after the
    // gather.  The one exception is the exchange buffers themselves — without them there is nothing to gather into; the
    // peers then give up after the comm's timeout with a message that names the missing rank (wait_stream).
    int rc_local = OTT_OK;
    if (cap < (perq ? (uint64_t)nq * d->k : d->k))
        rc_local = fail(OTT_ERR_INVALID, "ott_query_sharded: output capacity is smaller than k (MERGED) or nq * k (PER_QUERY)");
    if ((rc = ctx->x_send.ensure(block))) return rc;
    if ((rc = ctx->x_recv.ensure(block * (size_t)c->world))) return rc;

    // 1. this shard: scoring + top-k, the block stays in HBM (an empty shard contributes sentinels)
    bool events_pending = false;
    if (rc_local == OTT_OK && ctx->n)
        rc_local = query_core(ctx, d, nullptr, ctx->x_send.p, (uint64_t)groups * KS, nullptr, nullptr, nullptr, &st, true, &events_pending, co);
    else if (hipMemsetAsync(ctx->x_send.p, 0xFF, block, ctx->stream) != hipSuccess && rc_local == OTT_OK)
        rc_local = fail(OTT_ERR_HIP, "ott_query_sharded: hipMemsetAsync failed");
    if (rc_local) {  // still join the exchange (with nothing), so the other ranks do not hang; the error is returned after
        events_pending = false;
        (void)hipMemsetAsync(ctx->x_send.p, 0xFF, block, ctx->stream);
    }
    // 2. the exchange, on the same stream
    if ((rc = gather_device(c, ctx->x_send.p, ctx->x_recv.p, block, ctx->stream))) return rc;
    if (rc_local) {
        (void)wait_stream(c, ctx->stream, "ott_query_sharded");
        return rc_local;  // (its message was the last one set on this thread unless the wait itself failed too)
    }
    // 3. the merge (src/meta.rs:699-709) of world x groups lists, hits written straight into pinned host memory
    const size_t hits_bytes = (size_t)groups * KS * sizeof(ott_hit), cnt_bytes = (((size_t)groups * 8) + 63) & ~(size_t)63;
    if ((rc = ctx->h_hits.ensure(hits_bytes + cnt_bytes))) return rc;
    char* hh = (char*)ctx->h_hits.p;
    void* mapped = nullptr;
    OTT_HIP(hipHostGetDevicePointer(&mapped, hh, 0));
    hipEvent_t m0 = ctx->ev[0], m1 = ctx->ev[1];
    const bool timing = stats_out != nullptr;
    if (timing) OTT_HIP(hipEventRecord(m0, ctx->stream));
    if ((rc = launch_merge_hits(ctx, (const ott_hit*)ctx->x_recv.p, (uint32_t)c->world, groups, (uint32_t)KS, (uint32_t)d->k, E,
                                d->take == OTT_TAKE_MAX, (ott_hit*)((char*)mapped + cnt_bytes), (uint64_t*)mapped)))
        return rc;
    if (timing) OTT_HIP(hipEventRecord(m1, ctx->stream));
    if ((rc = wait_stream(c, ctx->stream, "ott_query_sharded"))) return rc;  // the only wait of the call (EXACT path); bounded when a peer could be missing
    const uint64_t* cnt = (const uint64_t*)hh;
    const ott_hit* hits = (const ott_hit*)(hh + cnt_bytes);
    uint64_t total = 0;
    for (uint32_t g = 0; g < groups; g++) {
        if (total + cnt[g] > cap) return fail(OTT_ERR_INVALID, "ott_query_sharded: output capacity is smaller than the result");
        if (cnt[g]) memcpy(out + total, hits + (size_t)g * KS, cnt[g] * sizeof(ott_hit));
        if (n_per_query && perq) n_per_query[g] = cnt[g];
        total += cnt[g];
    }
    if (n_out) *n_out = total;
    if (events_pending) read_exact_events(ctx, &st);
    float ms = 0.f;
    if (timing && hipEventElapsedTime(&ms, m0, m1) == hipSuccess) st.merge_ns += (uint64_t)(ms * 1e6);
    st.total_ns = now_ns() - t0;
    if (stats_out) *stats_out = st;
    return OTT_OK;
}

// tie_order = 1 across shards: the reference's single collector over the WHOLE corpus (src/vec.rs:217-310).  Every rank gets the
// same k + 1 candidates in (score, visit order) — per-shard lists in visit order, shards in row order, the cross-shard merge
// breaks ties by (shard, position) — so every rank takes the same decision; only when some group's cut is ambiguous does a
// second collective follow: the fill phase (first k passing pairs in visit order), by the same exchange with every passing
// score ranked the same.  The closed form of ott_ties.hip then runs on every rank.
int sharded_ref_ties(ott_store* ctx, ott_comm* c, const ott_query_desc* d, ott_hit* out, uint64_t cap, uint64_t* n_out, uint64_t* n_per_query,
                     ott_stats* stats_out) {
    const bool perq = d->mode == OTT_MODE_PER_QUERY, tmax = d->take == OTT_TAKE_MAX;
    const uint32_t nq = d->nq, groups = perq ? nq : 1u;
    if (n_out) *n_out = 0;
    if (n_per_query)
        for (uint32_t i = 0; i < nq; i++) n_per_query[i] = 0;
    if (d->k == 0) {
        if (stats_out) memset(stats_out, 0, sizeof(*stats_out));
        return OTT_OK;
    }
    CoreOpts co;
    co.tie_sh = 3;
    ott_query_desc d1 = *d;
    d1.k = d->k == ~0ull ? d->k : d->k + 1;
    std::vector<ott_hit> all((size_t)(perq ? (uint64_t)nq * d1.k : d1.k) + 1);
    std::vector<uint64_t> per(nq, 0);
    uint64_t n1 = 0;
    int rc = sharded_on(ctx, c, &d1, all.data(), all.size(), &n1, per.data(), stats_out, co);
    if (rc) return rc;
    std::vector<std::vector<ott_hit>> cand(groups);
    if (perq) {
        size_t o = 0;
        for (uint32_t g = 0; g < groups; g++) {
            cand[g].assign(all.begin() + o, all.begin() + o + (size_t)per[g]);
            o += (size_t)per[g];
        }
    } else {
        cand[0].assign(all.begin(), all.begin() + (size_t)n1);
    }
    bool any = false;  // (the same on every rank: all hold the same candidates)
    for (uint32_t g = 0; g < groups; g++) any = any || ties_ambiguous(tmax, cand[g], d->k);
    std::vector<std::vector<ott_hit>> fill(groups);
    if (any) {
        co.flat = true;
        ott_query_desc d2 = *d;
        d2.path = OTT_PATH_EXACT;
        std::vector<ott_hit> f((size_t)(perq ? (uint64_t)nq * d->k : d->k) + 1);
        std::vector<uint64_t> fper(nq, 0);
        uint64_t nf = 0;
        if ((rc = sharded_on(ctx, c, &d2, f.data(), f.size(), &nf, fper.data(), nullptr, co))) return rc;
        if (perq) {
            size_t o = 0;
            for (uint32_t g = 0; g < groups; g++) {
                fill[g].assign(f.begin() + o, f.begin() + o + (size_t)fper[g]);
                o += (size_t)fper[g];
            }
        } else {
            fill[0].assign(f.begin(), f.begin() + (size_t)nf);
        }
    }
    uint64_t total = 0;
    std::vector<ott_hit> res;
    for (uint32_t g = 0; g < groups; g++) {
        if ((rc = ties_resolve(ctx, tmax, 0 /* global rows: shards start on 8-row boundaries */, cand[g], d->k, any ? &fill[g] : nullptr, res))) return rc;
        if (total + res.size() > cap) return fail(OTT_ERR_INVALID, "ott_query_sharded: output capacity is smaller than the result");
        if (!res.empty()) memcpy(out + total, res.data(), res.size() * sizeof(ott_hit));
        if (n_per_query && perq) n_per_query[g] = res.size();
        total += res.size();
    }
    if (n_out) *n_out = total;
    return OTT_OK;
}

}  // namespace

extern "C" {

int ott_comm_unique_id(void* id_out) {
    if (!id_out) return fail(OTT_ERR_INVALID, "ott_comm_unique_id: id_out is NULL");
    Rccl* r = rccl();
    if (!r->handle) return fail(OTT_ERR_UNSUPPORTED, r->why);
    NcclId id;
    const int rc = r->GetUniqueId(&id);
    if (rc) return nccl_fail("ncclGetUniqueId", rc);
    memcpy(id_out, &id, sizeof(id));
    return OTT_OK;
}

int ott_comm_create(const void* unique_id, int rank, int world, int device, ott_comm** out) {
    if (!out) return fail(OTT_ERR_INVALID, "ott_comm_create: out is NULL");
    *out = nullptr;
    if (!unique_id || world < 1 || rank < 0 || rank >= world) return fail(OTT_ERR_INVALID, "ott_comm_create: bad id / rank / world");
    Rccl* r = rccl();
    if (!r->handle) return fail(OTT_ERR_UNSUPPORTED, r->why);
    OTT_HIP(hipSetDevice(device));
    NcclId id;
    memcpy(&id, unique_id, sizeof(id));
    ott_comm* c = new ott_comm();
    c->rank = rank;
    c->world = world;
    c->device = device;
    c->is_rccl = true;
    if (const char* e = getenv("OTT_COMM_TIMEOUT_MS")) c->timeout_ms = atoll(e);  // read once, here (ott_comm_set_timeout_ms afterwards)
    int rc;
    if (world == 1 || c->timeout_ms <= 0) {
        rc = r->CommInitRank(&c->nccl, world, id, rank);
    } else {
        // ncclCommInitRank blocks until all `world` ranks have called it: a rank that died before the rendezvous would keep
        // its peers in here for good.  The call runs on a helper thread; if it has not come back within the comm timeout the
        // caller gets an error that says which rendezvous is incomplete.  The helper stays blocked inside RCCL's bootstrap
        // (it cannot be cancelled) and owns its state through a shared_ptr, so nothing it touches is freed under it; the
        // process is expected to shut down after such an error.
        struct Boot {
            std::mutex mu;
            std::condition_variable cv;
            bool done = false;
            int rc = 0;
            void* comm = nullptr;
        };
        auto boot = std::make_shared<Boot>();
        std::thread([boot, r, world, id, rank, device]() {
            int rc2 = hipSetDevice(device) == hipSuccess ? 0 : -1;
            void* comm = nullptr;
            if (rc2 == 0) rc2 = r->CommInitRank(&comm, world, id, rank);
            std::lock_guard<std::mutex> g(boot->mu);
            boot->rc = rc2;
            boot->comm = comm;
            boot->done = true;
            boot->cv.notify_all();
        }).detach();
        std::unique_lock<std::mutex> lk(boot->mu);
        if (!boot->cv.wait_for(lk, std::chrono::milliseconds(c->timeout_ms), [&] { return boot->done; })) {
            const int64_t t = c->timeout_ms;
            delete c;
            return fail(OTT_ERR_HIP, "ott_comm_create: ncclCommInitRank did not complete within " + std::to_string(t) + " ms (rank " +
                                         std::to_string(rank) + " of " + std::to_string(world) +
                                         "): a peer did not arrive at the rendezvous (OTT_COMM_TIMEOUT_MS sets the limit, 0 = wait for ever)");
        }
        rc = boot->rc;
        c->nccl = boot->comm;
        if (rc == -1) {
            delete c;
            return fail(OTT_ERR_HIP, "ott_comm_create: hipSetDevice failed on the bootstrap thread");
        }
    }
    if (rc) {
        delete c;
        return nccl_fail("ncclCommInitRank", rc);
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        (void)r->CommDestroy(c->nccl);
        delete c;
        return fail(OTT_ERR_HIP, "ott_comm_create: hipStreamCreate failed");
    }
    *out = c;
    return OTT_OK;
}

int ott_comm_create_host(int rank, int world, ott_allgather_fn fn, void* user, ott_comm** out) {
    if (!out) return fail(OTT_ERR_INVALID, "ott_comm_create_host: out is NULL");
    *out = nullptr;
    if (!fn || world < 1 || rank < 0 || rank >= world) return fail(OTT_ERR_INVALID, "ott_comm_create_host: bad callback / rank / world");
    ott_comm* c = new ott_comm();
    c->rank = rank;
    c->world = world;
    c->fn = fn;
    c->user = user;
    *out = c;
    return OTT_OK;
}

int ott_comm_destroy(ott_comm* c) {
    if (!c) return OTT_OK;
    if (c->is_rccl) {
        (void)hipSetDevice(c->device);
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        if (c->nccl) (void)rccl()->CommDestroy(c->nccl);
        if (c->stream) (void)hipStreamDestroy(c->stream);
    }
    c->d_send.release();
    c->d_recv.release();
    c->h_send.release();
    c->h_recv.release();
    delete c;
    return OTT_OK;
}

int ott_comm_set_timeout_ms(ott_comm* c, int64_t timeout_ms) {
    if (!c || timeout_ms < 0) return fail(OTT_ERR_INVALID, "ott_comm_set_timeout_ms: NULL comm or negative timeout");
    std::lock_guard<std::mutex> g(c->mu);
    c->timeout_ms = timeout_ms;
    return OTT_OK;
}

int ott_comm_rank(const ott_comm* c) { return c ? c->rank : -1; }
int ott_comm_world(const ott_comm* c) { return c ? c->world : 0; }
const char* ott_comm_transport(const ott_comm* c) { return !c ? "" : c->is_rccl ? "rccl" : "host"; }

int ott_comm_all_gather_host(ott_comm* c, const void* send_host, void* recv_host, uint64_t bytes) {
    if (!c || (bytes && (!send_host || !recv_host))) return fail(OTT_ERR_INVALID, "ott_comm_all_gather_host: NULL argument");
    std::lock_guard<std::mutex> g(c->mu);
    return gather_host_locked(c, send_host, recv_host, bytes);
}

int ott_query_sharded(ott_store* s, ott_comm* c, const ott_query_desc* d, ott_hit* out, uint64_t cap, uint64_t* n_out, uint64_t* n_per_query,
                      ott_stats* stats) {
    if (!c) return fail(OTT_ERR_INVALID, "ott_query_sharded: comm is NULL");
    if (!out && cap) return fail(OTT_ERR_INVALID, "ott_query_sharded: out is NULL");
    if (s && s->multi) return fail(OTT_ERR_UNSUPPORTED, "ott_query_sharded: the shard of a multi-process job is a single-GPU store (ott_store_create)");
    int rc = validate_query(s, d);
    if (rc) return rc;
    if (c->is_rccl && c->device != s->device) return fail(OTT_ERR_INVALID, "ott_query_sharded: the comm and the store live on different GPUs");
    std::lock_guard<std::mutex> g(c->mu);
    std::shared_lock<std::shared_mutex> rd(s->rw);
    ott_store* ctx = ctx_acquire(s);
    if (s->opt.tie_order == 1 && (s->base_offset & 7) == 0) {
        rc = sharded_ref_ties(ctx, c, d, out, cap, n_out, n_per_query, stats);
    } else {
        // tie_order = 2 (per-chunk collectors) across shards: candidates ranked in visit order, without the collector's anchor rule
        CoreOpts co;
        co.tie_sh = s->opt.tie_order ? 3u : 0u;
        rc = sharded_on(ctx, c, d, out, cap, n_out, n_per_query, stats, co);
    }
    ctx_release(ctx);
    return rc;
}

}  // extern "C"

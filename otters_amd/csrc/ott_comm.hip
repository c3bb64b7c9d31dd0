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
#include <stdio.h>
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
        // OTT_RCCL_LIBRARY (read once per process, here): another library that exports the nccl* entry points this file binds —
        // the tests' stand-in (tests/fake_rccl: N ranks on ONE device, the collective as stream-ordered copies), so that the
        // grouped all-gather branch of the multi-GPU store runs with more than one rank on a one-GPU box
        // It is a TEST hook: honoured only when OTT_TEST_HOOKS=1 is set beside it (a deployment that merely inherits an environment
        // does not get its collective library replaced by one variable), and said so on stderr once.
        const char* override_name = getenv("OTT_RCCL_LIBRARY");
        const char* hooks = getenv("OTT_TEST_HOOKS");
        if (override_name && *override_name && !(hooks && hooks[0] == '1' && hooks[1] == '\0')) {
            fprintf(stderr, "libotters_hip: OTT_RCCL_LIBRARY is ignored (a test hook: it needs OTT_TEST_HOOKS=1)\n");
            override_name = nullptr;
        }
        if (override_name && *override_name) {
            fprintf(stderr, "libotters_hip: TEST HOOK: the collectives come from %s, not from RCCL\n", override_name);
            r.handle = dlopen(override_name, RTLD_NOW | RTLD_LOCAL);
            if (!r.handle) {
                const char* e = dlerror();
                r.why = std::string("OTT_RCCL_LIBRARY=") + override_name + " could not be loaded: " + (e ? e : "?");
                return;
            }
        }
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (r.handle) break;
            r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
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
    // What every rank said about its shard (ott_query_sharded): tie order, base offset, chunk size, rows.  Gathered by the
    // first sharded query on the comm; after that every rank's four words ride in the header of every candidate exchange, so
    // a change on ANY rank (an append on one shard only) reaches all ranks with the same exchange and all of them update the
    // table together.  Whether a separate gather is issued, and which protocol a query takes, therefore follows from what
    // ALL ranks know — never from rank-local state (a rank that alone decided to re-gather would face peers that had gone
    // straight to the candidate exchange: mismatched collectives).
    struct Layout {
        uint64_t mine[4] = {0, 0, 0, 0};  // this rank's values for the call that is running (they travel in its headers)
        std::vector<uint64_t> all;        // [world][4]; empty = never gathered
        int verdict = OTT_OK;
        std::string why;
    } layout;
};

// header behind the hits of every rank's candidate block (HDR_SLOTS ott_hit slots = 64 bytes)
struct ShardHdr {
    uint32_t magic;
    int32_t status;  // this rank's scoring status (ott_status): a failure reaches every rank WITH the exchange, not as a timeout
    uint32_t pad[2];
    uint64_t layout[4];  // this rank's (tie order, base offset, chunk size, rows) for this call: see ott_comm::Layout
    uint64_t reserved[2];
};
static_assert(sizeof(ShardHdr) == 64, "ShardHdr is four ott_hit slots");
constexpr uint32_t HDR_SLOTS = 4, HDR_MAGIC = 0x4F545448u;
// internal status of one sharded round: the exchange showed that a rank's layout words differ from the table every rank holds.
// The table is updated (identically everywhere: all ranks saw the same headers) and ott_query_sharded starts the query again.
constexpr int OTT_LAYOUT_CHANGED = 1;

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
    OTT_HIP(use_device_raw(c->device, c->device));
    if ((e = c->d_send.ensure(bytes))) return e;
    if ((e = c->d_recv.ensure(bytes * (size_t)c->world))) return e;
    OTT_HIP(hipMemcpyAsync(c->d_send.p, send, bytes, hipMemcpyHostToDevice, c->stream));
    const int rc = rccl()->AllGather(c->d_send.p, c->d_recv.p, bytes, kNcclUint8, c->nccl, c->stream);
    if (rc) return nccl_fail("ncclAllGather", rc);
    OTT_HIP(hipMemcpyAsync(recv, c->d_recv.p, bytes * (size_t)c->world, hipMemcpyDeviceToHost, c->stream));
    return wait_stream(c, c->stream, "ott_comm_all_gather_host");
}

bool layout_from_headers(ott_comm* c, const uint64_t* words, size_t stride_words);

// k > 512: every rank's sorted list travels whole.  counts first (per group, with this rank's status behind them), then the
// lists padded to the longest, then a host merge in the canonical order (src/meta.rs:699-709: concat, sort, truncate(k)).
int sharded_large_k(ott_store* ctx, ott_comm* c, const ott_query_desc* d, ott_hit* out, uint64_t cap, uint64_t* n_out, uint64_t* n_per_query,
                    ott_stats* stats, const CoreOpts& co) {
    const bool perq = d->mode == OTT_MODE_PER_QUERY;
    const uint32_t nq = d->nq, groups = perq ? nq : 1u;
    const uint64_t pool = perq ? ctx->n : ctx->n * (uint64_t)nq;
    const uint64_t k_loc = d->k < pool ? d->k : pool;
    std::vector<ott_hit> mine((size_t)(k_loc * (perq ? nq : 1)) + 1);
    const size_t CW = (size_t)groups + 1 + 4;  // per rank: the groups' counts, its status, its four layout words
    std::vector<uint64_t> cnt_mine(CW, 0), per(nq, 0);
    uint64_t n_mine = 0;
    int rc = OTT_OK;
    // (co.tie_sh = 3: every shard ranks its candidates in the reference's visit order and the cross-shard merge keeps it —
    // shards are in row order and start on 8-row block boundaries)
    if (ctx->n) rc = query_core(ctx, d, mine.data(), nullptr, mine.size(), &n_mine, per.data(), nullptr, stats, false, nullptr, co);
    else if (stats) memset(stats, 0, sizeof(*stats));
    // a rank that failed still joins the collectives (with nothing), so the others do not hang; its status travels with the
    // counts, and every rank leaves before the second exchange
    const int rc_local = rc;
    const std::string msg_local = rc_local ? ott_last_error() : "";
    if (rc_local) n_mine = 0;
    if (perq) for (uint32_t q = 0; q < nq; q++) cnt_mine[q] = rc_local ? 0 : per[q];
    else cnt_mine[0] = n_mine;
    cnt_mine[groups] = (uint64_t)(uint32_t)(-rc_local);
    memcpy(&cnt_mine[(size_t)groups + 1], c->layout.mine, 4 * sizeof(uint64_t));
    std::vector<uint64_t> cnt_all(CW * c->world);
    if ((rc = gather_host_locked(c, cnt_mine.data(), cnt_all.data(), (uint64_t)CW * 8))) return rc;
    // (a layout change first: every rank sees it in the same words and leaves before the second exchange, like a failure)
    if (layout_from_headers(c, cnt_all.data() + groups + 1, CW)) return OTT_LAYOUT_CHANGED;
    if (rc_local) return fail(rc_local, msg_local);
    for (int r = 0; r < c->world; r++)
        if (cnt_all[(size_t)r * CW + groups] != 0)
            return fail(OTT_ERR_HIP, "ott_query_sharded: rank " + std::to_string(r) + " failed to score its shard (status -" +
                                         std::to_string(cnt_all[(size_t)r * CW + groups]) + "); no rank returns a result");
    uint64_t longest = 0;
    for (int r = 0; r < c->world; r++) {
        uint64_t t = 0;
        for (uint32_t g = 0; g < groups; g++) t += cnt_all[(size_t)r * CW + g];
        longest = t > longest ? t : longest;
    }
    std::vector<ott_hit> all;
    if (longest) {
        mine.resize((size_t)longest);
        all.resize((size_t)longest * c->world);
        if ((rc = gather_host_locked(c, mine.data(), all.data(), longest * sizeof(ott_hit)))) return rc;
    }
    // (visit order across ranks: 8-row blocks are counted from the JOB's first row — rank 0's base, which every rank knows; a
    //  one-chunk query, co.tie_off != 0, has entries on its owning rank only, so nothing is left to order across ranks)
    const CanonLess less{d->take == OTT_TAKE_MAX, co.tie_sh, co.tie_sh ? c->layout.all[1] : 0};
    uint64_t total = 0;
    std::vector<size_t> off((size_t)c->world, 0);  // per rank: where the next group starts in its list
    std::vector<const ott_hit*> head((size_t)c->world), end((size_t)c->world);
    for (uint32_t g = 0; g < groups; g++) {
        uint64_t have = 0;
        for (int r = 0; r < c->world; r++) {
            const uint64_t n = cnt_all[(size_t)r * CW + g];
            head[r] = all.data() + (size_t)r * longest + off[r];
            end[r] = head[r] + n;
            off[r] += (size_t)n;
            have += n;
        }
        const uint64_t keep = have < d->k ? have : d->k;
        if (total + keep > cap) return fail(OTT_ERR_INVALID, "ott_query_sharded: output capacity is smaller than the result");
        // every rank's list is already in order: a world-way merge of their heads (different ranks hold different rows, so no two
        // keys are equal and the order is total)
        merge_heads(head, end, less, out + total, keep);
        if (n_per_query && perq) n_per_query[g] = keep;
        total += keep;
    }
    if (n_out) *n_out = total;
    return OTT_OK;
}

int sharded_on(ott_store* ctx, ott_comm* c, const ott_query_desc* d, ott_hit* out, uint64_t cap, uint64_t* n_out, uint64_t* n_per_query,
               ott_stats* stats_out, const CoreOpts& co) {
    int rc;
    OTT_HIP(use_device(ctx));
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
    // block geometry from k alone (every rank must agree): [groups][KS] hit slots, KS = the register list width for k, and
    // HDR_SLOTS slots of header behind them
    const int E = list_E(d->k);
    const uint64_t KS = 64ull * (uint64_t)E;
    const size_t hits_bytes = (size_t)groups * KS * sizeof(ott_hit);
    const size_t block = hits_bytes + HDR_SLOTS * sizeof(ott_hit);
    // A failure that only THIS rank sees must not keep it away from the exchange (its peers would sit in ncclAllGather until
    // their comm timeout): it is remembered, the rank contributes a block of sentinels whose HEADER carries the status, and
    // after the exchange every rank knows — the failing one returns its own error, the others an error that names it.  The one
    // exception is the exchange buffers themselves — without them there is nothing to gather into; the peers then give up
    // after the comm's timeout with a message that names the missing rank (wait_stream).
    int rc_local = OTT_OK;
    std::string msg_local;
    if (cap < (perq ? (uint64_t)nq * d->k : d->k))
        rc_local = fail(OTT_ERR_INVALID, "ott_query_sharded: output capacity is smaller than k (MERGED) or nq * k (PER_QUERY)");
    if ((rc = ctx->x_send.ensure(block))) return rc;
    if ((rc = ctx->x_recv.ensure(block * (size_t)c->world))) return rc;
    if ((rc = ctx->h_hdr.ensure(sizeof(ShardHdr)))) return rc;

    // 1. this shard: scoring + top-k, the block stays in HBM (an empty shard contributes sentinels)
    bool events_pending = false;
    if (rc_local == OTT_OK && ctx->n)
        rc_local = query_core(ctx, d, nullptr, ctx->x_send.p, (uint64_t)groups * KS, nullptr, nullptr, nullptr, &st, true, &events_pending, co);
    else if (hipMemsetAsync(ctx->x_send.p, 0xFF, hits_bytes, ctx->stream) != hipSuccess && rc_local == OTT_OK)
        rc_local = fail(OTT_ERR_HIP, "ott_query_sharded: hipMemsetAsync failed");
    if (rc_local) {  // still join the exchange (with nothing), so the other ranks do not hang; the error is returned after
        msg_local = ott_last_error();
        events_pending = false;
        (void)hipGetLastError();
        (void)hipMemsetAsync(ctx->x_send.p, 0xFF, hits_bytes, ctx->stream);
    }
    ShardHdr* hdr = (ShardHdr*)ctx->h_hdr.p;  // (free again: the previous query on this context waited for its stream)
    memset(hdr, 0, sizeof(*hdr));
    hdr->magic = HDR_MAGIC;
    hdr->status = rc_local;
    memcpy(hdr->layout, c->layout.mine, sizeof(hdr->layout));
    OTT_HIP(hipMemcpyAsync((char*)ctx->x_send.p + hits_bytes, hdr, sizeof(*hdr), hipMemcpyHostToDevice, ctx->stream));
    // 2. the exchange, on the same stream
    const bool timing = stats_out != nullptr;
    if (timing) OTT_HIP(hipEventRecord(ctx->ev[6], ctx->stream));
    if ((rc = gather_device(c, ctx->x_send.p, ctx->x_recv.p, block, ctx->stream))) return rc;
    // 3. the merge (src/meta.rs:699-709) of world x groups lists, hits written straight into pinned host memory; the same launch
    //    hands every rank's header to the host
    const size_t cnt_bytes = (((size_t)groups * 8) + 63) & ~(size_t)63, hdr_bytes = (size_t)c->world * sizeof(ShardHdr);
    if ((rc = ctx->h_hits.ensure(hits_bytes + cnt_bytes + hdr_bytes))) return rc;
    char* hh = (char*)ctx->h_hits.p;
    void* mapped = nullptr;
    OTT_HIP(hipHostGetDevicePointer(&mapped, hh, 0));
    hipEvent_t m0 = ctx->ev[0], m1 = ctx->ev[1];
    if (timing) OTT_HIP(hipEventRecord(m0, ctx->stream));
    if ((rc = launch_merge_hits(ctx, (const ott_hit*)ctx->x_recv.p, (uint32_t)c->world, groups, (uint32_t)KS, (uint32_t)d->k, E,
                                d->take == OTT_TAKE_MAX, (ott_hit*)((char*)mapped + cnt_bytes), (uint64_t*)mapped, HDR_SLOTS,
                                (ott_hit*)((char*)mapped + cnt_bytes + hits_bytes))))
        return rc;
    if (timing) OTT_HIP(hipEventRecord(m1, ctx->stream));
    if ((rc = wait_stream(c, ctx->stream, "ott_query_sharded"))) return rc;  // the only wait of the call (EXACT path); bounded when a peer could be missing
    const ShardHdr* hdrs = (const ShardHdr*)(hh + cnt_bytes + hits_bytes);
    for (int r = 0; r < c->world; r++)
        if (hdrs[r].magic != HDR_MAGIC)
            return fail(OTT_ERR_HIP, "ott_query_sharded: rank " + std::to_string(r) + " sent a malformed block (ranks calling with different k / mode / library versions?)");
    // a rank whose shard changed since the table was made (rows appended on one rank only): every rank sees the same headers,
    // updates the table and starts the query again with it
    if (layout_from_headers(c, hdrs[0].layout, sizeof(ShardHdr) / sizeof(uint64_t))) return OTT_LAYOUT_CHANGED;
    if (rc_local) return fail(rc_local, msg_local);
    for (int r = 0; r < c->world; r++) {
        if (hdrs[r].status != OTT_OK)
            return fail(OTT_ERR_HIP, "ott_query_sharded: rank " + std::to_string(r) + " failed to score its shard (status " + std::to_string(hdrs[r].status) +
                                         "); no rank returns a result");
    }
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
    if (timing && hipEventElapsedTime(&ms, ctx->ev[6], m0) == hipSuccess) st.exchange_ns = (uint64_t)(ms * 1e6);
    st.total_ns = now_ns() - t0;
    if (stats_out) *stats_out = st;
    return OTT_OK;
}

// The job's layout, checked from GLOBAL facts: every rank's (tie order, base offset, chunk size, rows).  What a query does
// next — which exchange sizes, how many exchanges — depends only on what all ranks know, never on a rank-local property; a
// layout the reference's tie orders cannot be reproduced on fails on EVERY rank with the same message.
void judge_layout(ott_comm* c) {
    ott_comm::Layout& L = c->layout;
    const std::vector<uint64_t>& all = L.all;
    L.verdict = OTT_OK;
    L.why.clear();
    auto bad = [&](int code, const std::string& why) {
        L.verdict = code;
        L.why = why;
    };
    const uint64_t tie = all[0];
    for (int r = 0; r < c->world && !L.verdict; r++) {
        const uint64_t* a = &all[(size_t)r * 4];
        if (a[0] != tie) bad(OTT_ERR_INVALID, "ott_query_sharded: the ranks' stores have different tie_order options (rank 0: " + std::to_string(tie) + ", rank " + std::to_string(r) + ": " + std::to_string(a[0]) + ")");
        else if (r && a[1] < all[(size_t)(r - 1) * 4 + 1] + all[(size_t)(r - 1) * 4 + 3]) bad(OTT_ERR_INVALID, "ott_query_sharded: shards must be in rank order and must not overlap (rank " + std::to_string(r) + " starts inside rank " + std::to_string(r - 1) + "'s rows)");
        else if (tie == 1 && ((a[1] - all[1]) & 7) != 0)
            bad(OTT_ERR_UNSUPPORTED, "ott_query_sharded: tie_order = 1 (the reference's single collector) needs every shard to start a multiple of 8 rows after the first (rank " + std::to_string(r) + " starts at row " + std::to_string(a[1]) + ")");
        else if (tie == 2 && (a[2] != all[2] || (a[1] - all[1]) % a[2] != 0))
            bad(OTT_ERR_UNSUPPORTED, "ott_query_sharded: tie_order = 2 (the reference's per-chunk collectors) needs one chunk size on every rank and shards that start on chunk boundaries (rank " + std::to_string(r) + ")");
    }
}

// Before a sharded query.  The separate 32-byte gather is issued only when NO rank holds a usable table — the first sharded
// query on the comm, or after a verdict that failed the previous call on every rank alike — which all ranks know together.
// Every other change travels in the exchange headers (layout_from_headers).
int check_layout(ott_store* s, ott_comm* c) {
    ott_comm::Layout& L = c->layout;
    const uint64_t mine[4] = {(uint64_t)s->opt.tie_order, s->base_offset, s->chunk_size, s->n};
    memcpy(L.mine, mine, sizeof(mine));
    if (!L.all.empty() && L.verdict == OTT_OK) return OTT_OK;
    std::vector<uint64_t> all((size_t)c->world * 4);
    int rc = gather_host_locked(c, mine, all.data(), sizeof(mine));
    if (rc) return rc;
    L.all = all;
    judge_layout(c);
    return L.verdict ? fail(L.verdict, L.why) : OTT_OK;
}

// After an exchange: what every rank's header (or count block) says about its shard against the table.  true = they differ;
// the table then holds the new values and a new verdict — on every rank alike, since all ranks compared the same words.
bool layout_from_headers(ott_comm* c, const uint64_t* words, size_t stride_words) {
    ott_comm::Layout& L = c->layout;
    bool changed = false;
    for (int r = 0; r < c->world; r++)
        if (memcmp(&L.all[(size_t)r * 4], words + (size_t)r * stride_words, 4 * sizeof(uint64_t)) != 0) changed = true;
    if (!changed) return false;
    for (int r = 0; r < c->world; r++) memcpy(&L.all[(size_t)r * 4], words + (size_t)r * stride_words, 4 * sizeof(uint64_t));
    judge_layout(c);
    return true;
}

// tie_order 1 / 2 across ranks: the decision logic of ott_ties.hip over candidates every rank holds identically (per-shard
// lists in visit order, shards in row order, the cross-shard merge breaks ties by (shard, position)), so every rank takes the
// same decisions and issues the same further exchanges: the fill phase when a cut is ambiguous, and for tie_order 2 the
// per-chunk collectors of the chunks that hold candidates (each answered by the rank that owns the chunk).
int sharded_ref_ties(ott_store* s, ott_store* ctx, ott_comm* c, const ott_query_desc* d, int tie_order, ott_hit* out, uint64_t cap, uint64_t* n_out,
                     uint64_t* n_per_query, ott_stats* stats_out) {
    const ott_comm::Layout& L = c->layout;
    const uint64_t base0 = L.all[1];
    uint64_t total_rows = 0;
    for (int r = 0; r < c->world; r++) total_rows += L.all[(size_t)r * 4 + 3];
    TieEnv env;
    env.tmax = d->take == OTT_TAKE_MAX;
    env.base = base0;
    env.chunk_size = L.all[2];  // the TABLE's chunk size (tie order 2: one on every rank, judge_layout), never a rank-local value
    env.dim = s->dim;
    const auto run_off = [ctx, c, total_rows](const ott_query_desc& dd, uint64_t k, bool flat, uint32_t tie_off, std::vector<ott_hit>& o, std::vector<uint64_t>& per,
                                              ott_stats* st) -> int {
        ott_query_desc d2 = dd;
        if (flat) d2.path = OTT_PATH_EXACT;
        const bool pq = dd.mode == OTT_MODE_PER_QUERY;
        const uint64_t pool = pq ? total_rows : total_rows * (uint64_t)dd.nq;
        const uint64_t ke = k < pool ? k : pool;
        d2.k = ke;
        o.resize((size_t)(pq ? ke * dd.nq : ke) + 1);
        per.assign(dd.nq, 0);
        uint64_t n2 = 0;
        CoreOpts co;
        co.tie_sh = 3;
        co.flat = flat;
        co.tie_off = tie_off;
        const int rc = sharded_on(ctx, c, &d2, o.data(), o.size(), &n2, per.data(), st, co);
        if (rc) return rc;
        o.resize((size_t)n2);
        return OTT_OK;
    };
    env.run = [run_off](const ott_query_desc& dd, uint64_t k, bool flat, std::vector<ott_hit>& o, std::vector<uint64_t>& per, ott_stats* st) -> int {
        return run_off(dd, k, flat, 0, o, per, st);
    };
    env.run_chunk = [run_off, ctx, base0](uint64_t chunk, const ott_query_desc& dd, uint64_t k, bool flat, std::vector<ott_hit>& o,
                                        std::vector<uint64_t>& per, ott_stats* st) -> int {
        // the chunk (counted from the job's first row) in THIS rank's terms: one bit of its own chunk mask, or none at all
        const uint64_t cs = ctx->chunk_size, first = base0 + chunk * cs;
        const uint64_t n_chunks = (ctx->n + cs - 1) / cs;
        std::vector<uint64_t> mask((size_t)((n_chunks + 63) / 64) + 1, 0);
        uint32_t off = 0;  // the chunk's 8-row blocks, counted from its first row, in the OWNING rank's local rows (the others list nothing)
        if (first >= ctx->base_offset && first < ctx->base_offset + ctx->n) {
            const uint64_t lc = (first - ctx->base_offset) / cs;
            mask[(size_t)(lc >> 6)] = 1ull << (lc & 63);
            off = (uint32_t)((8 - (first - ctx->base_offset) % 8) % 8);
        }
        ott_query_desc d3 = dd;
        d3.chunk_mask = mask.data();
        return run_off(d3, k, flat, off, o, per, st);
    };
    return ref_ties_collect(env, tie_order, d, out, cap, n_out, n_per_query, stats_out);
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
    OTT_HIP(use_device_raw(device, device));
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
            int rc2 = use_device_raw(device, device) == hipSuccess ? 0 : -1;
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
        (void)use_device_raw(c->device, c->device);
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

int ott_comm_info(const ott_comm* c, int* nranks, int* version) {
    if (!c) return fail(OTT_ERR_INVALID, "ott_comm_info: comm is NULL");
    int n = c->world, v = 0;
    if (c->is_rccl) {
        Rccl* r = rccl();
        if (r->CommCount) {
            const int rc = r->CommCount(c->nccl, &n);
            if (rc) return nccl_fail("ncclCommCount", rc);
        }
        if (r->GetVersion) (void)r->GetVersion(&v);
    }
    if (nranks) *nranks = n;
    if (version) *version = v;
    return OTT_OK;
}

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
    if ((rc = store_flush(s))) return rc;
    std::lock_guard<std::mutex> g(c->mu);
    ott::host::SharedLock rd(s->rw);
    ott_store* ctx = nullptr;
    for (int attempt = 0;; attempt++) {
        // (one small gather on the comm's first sharded query; afterwards the ranks' layout words ride in every exchange)
        if ((rc = check_layout(s, c))) break;
        if (!ctx) ctx = ctx_acquire(s);
        // Which protocol runs (and with which block sizes) follows from the TABLE every rank holds — rank 0's tie order, which
        // judge_layout has shown to be everyone's — never from this rank's own option: a rank whose tie_order changed since the
        // table was gathered still issues the exchange its peers issue; its header words carry the change, every rank sees it in
        // that same exchange (OTT_LAYOUT_CHANGED) and the call goes round again with the new table and its verdict.
        const int tie = (int)c->layout.all[0];
        if (tie != 0) rc = sharded_ref_ties(s, ctx, c, d, tie, out, cap, n_out, n_per_query, stats);
        else rc = sharded_on(ctx, c, d, out, cap, n_out, n_per_query, stats, CoreOpts{});
        if (rc != OTT_LAYOUT_CHANGED) break;
        // some rank's shard differs from the table (every rank saw that in the same exchange and is here too): again, with the
        // updated table — its verdict first.  Within ONE call no rank's values can change, so the second round is final
        if (c->layout.verdict) {
            rc = fail(c->layout.verdict, c->layout.why);
            break;
        }
        if (attempt >= 2) {
            rc = fail(OTT_ERR_HIP, "ott_query_sharded: the ranks' shard layouts keep changing inside one call (ranks calling with different stores?)");
            break;
        }
    }
    if (ctx) ctx_release(ctx);
    return rc;
}

}  // extern "C"

// ott_internal.h — shared declarations of libotters_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <vector>

#include "../../include/otters_hip.h"
#include "ott_host.h"  // the host-side concurrency (thread pool, context pool, staged appends, background worker): HIP-free, sanitizer-tested
#ifdef OTT_DEVICE_AUDIT
#include "ott_audit.h"  // test build: every HIP call below goes through a device-affinity check (see "which GPU a call is for")
#endif

namespace ott {

// ---- error plumbing ------------------------------------------------------------------------
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

#define OTT_HIP(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            (void)hipGetLastError(); /* reported here: a later launch check must not see it again */ \
            return ::ott::fail(_e == hipErrorOutOfMemory ? OTT_ERR_OOM : OTT_ERR_HIP,              \
                               std::string(#expr) + ": " + hipGetErrorString(_e));                 \
        }                                                                                          \
    } while (0)

// ---- which GPU a call is for ------------------------------------------------------------------------------------------------
// Every store (shard, worker context) lives on one HIP device, and everything done for it — allocations, launches, event and
// stream calls — must happen with that device current on the calling thread.  use_device() is the ONE way the library selects
// a device.  Besides the physical ordinal a store carries a LOGICAL device id: equal to the ordinal normally; with the test
// option "multi_fake_distinct" every shard of a multi-GPU store gets an id of its own although the shards share the box's one
// GPU.  The device-affinity audit build (make audit: -DOTT_DEVICE_AUDIT, ott_audit.hip) records the logical id of the last
// use_device() in a thread-local and checks it at every allocation, launch, copy, event and stream call against the
// id the stream / event / buffer was created under: a missed use_device() on a shard thread, in the background plane builder
// or in drain() then fails on ONE GPU instead of corrupting memory on eight.
#ifndef OTT_DEVICE_AUDIT
#define OTT_AUDIT_PTR(ptr, store) ((void)0)
inline hipError_t use_device_raw(int device, int /*logical*/) { return hipSetDevice(device); }
#endif

// hipFuncSetAttribute applies to the CURRENT device: one bit per device ordinal says where a kernel already has its opt-in
// (a process may hold stores on several GPUs: ott_store_create_multi)
inline bool attr_needed(const std::atomic<uint64_t>& seen, int device) {
    return device > 63 || !((seen.load(std::memory_order_acquire) >> (device & 63)) & 1ull);
}
inline void attr_done(std::atomic<uint64_t>& seen, int device) {
    if (device <= 63) seen.fetch_or(1ull << (device & 63), std::memory_order_release);
}

// ---- device buffers that grow on demand -----------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes);  // keeps contents only if no reallocation is needed
    void release();
};
struct PinBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes);
    void release();
};

// Behaviour switches of one store.  Read ONCE from the environment (OTT_* variables of the same names, upper case) when
// the store is created, changed afterwards only through ott_store_set_option: the query path never looks at the
// environment.  SIXTEEN of them have a name in the product library (ott_store.hip: kOptNames; round 5 retired the rest):
// tie_order, hi_fmt, hi_prebuild, stage_appends, multi_transport, multi_rebalance, multi_min_shard_rows — behaviour a host may
// want; exact_small, large_k_from, small_sort, mfma_f32, no_hi_pass, no_batch_image — which of several equivalent paths
// runs (tests hold each to the oracle); force_fallback, eps_scale_ppm, multi_fake_distinct — tests only.  The fields marked
// [debug build] can be set by name only in a library built with -DOTT_MFMA_DEBUG_BUILD (kernel tuning / timing ablations);
// the fields marked [fallback] are set through the bits of force_fallback.
struct Options {
    int exact_small = -1;         // single-query small-store kernel: -1 = automatic, 0 = streaming kernel, 2 = rows8 (eight lanes per row)
                                  // (1, round 2's one-wave LDS-DMA variant, is retired: 31 us against rows8's 10)
    int force_fallback = 0;       // TESTS: bit mask of code paths the library otherwise takes only in rare conditions, forced on so that the
                                  // suite and the option fuzz hold them to the oracle: 1 = block lists merged by insertion (merge_kernel: the
                                  // rank merge's own fallback when a plateau overflows its buffer or there are > 4096 lists), 2 = k <= 64
                                  // through sorted heads + tree fold (merge_small_kernel: > 4096 lists), 4 = the 256-query blocks of a row
                                  // tile one after the other on one workgroup (what three blocks or a short round get anyway), 8 = the
                                  // sort path lists every pair (no prefix gate: what its first phase does anyway), 16 = the open first
                                  // round through the cursor atomics instead of dense stores, 32 = conservative emission thresholds between
                                  // the row rounds (what a store backs off to after a speculative gate failed), 64 = the sort path cuts its
                                  // rows into slices of 2^14 (row, query) pairs instead of 2^29 (what only stores of billions of pairs do)
    bool mfma_f32 = false;        // batch path: ONE candidate pass on the f32 matrix pipe (v_mfma_f32_32x32x2_f32)
    bool no_hi_pass = false;      // batch path starts at the split-bf16 pass (no hi plane is built)
    bool no_batch_image = false;  // no bf16 copies of the corpus at all (the split pass splits the f32 rows in registers)
    int mfma_wg = 0;              // [debug build] workgroups per CU of the candidate pass (0 = the tile's default)
    int mfma_growth = 8;          // [debug build] growth factor of the candidate pass's row rounds
    bool mfma_no_dense = false;   // [fallback 16] open first round through the cursor atomics instead of dense stores
    bool mfma_debug = false;      // [debug build] in-kernel cycle stamps
    int mfma_abl = 0;             // [debug build] timing ablations of the candidate kernel and the radix sort (results are then wrong)
    int mfma_spec = -1;           // [fallback 32] speculative emission thresholds between the row rounds of the first cascade level (-1 = default on, 0 / 1)
    int mfma_coop = -1;           // [fallback 4] > 256 queries: the query blocks of a row tile on sibling workgroups of one XCD at the same time (-1 = default, 0 / 1)
    int tie_order = 0;            // 0 = canonical total order; 1 = the reference's outcome at exact score ties, ONE collector over the store
                                  // (VecStore, src/vec.rs:217-310); 2 = one collector per chunk, then concat-sort-truncate (MetaStore,
                                  // src/meta.rs:678-709).  See ott_ties.hip
    int hi_fmt = -1;              // what the batch path's 16-bit / 8-bit copies of the corpus are: -1 (default) / 2 = an INT8 plane as the cascade's
                                  // first level (k <= 128; a quarter of the f32 bytes) with an IEEE-half hi plane behind it, built only
                                  // once a query needs it (k > 128, or what the int8 level could not certify); 1 = the half plane alone
                                  // (round 3-4's default); 0 = a bf16 plane alone.  Takes effect when a plane is (re)built
    int hi_tmin = 0;              // [debug build] the hi pass re-scores at least this many candidates per query (0 = 2k + 56; at most 512)
    int merge_rank1 = -1;         // [fallback 2] k <= 64: merge_rank_kernel (-1 / 1, default) or round 2's merge_small_kernel (0)
    bool merge_walk = false;      // [fallback 1] k > 64: merge the block lists by insertion (merge_kernel) instead of bound + gather + rank (merge_rank_kernel)
    int large_k_from = 0;         // experiments: k above which host-output queries take the sort path (0 = automatic: 512 for one query or a small store, 128 for several queries; at most 512)
    int large_k_pre = -1;         // [fallback 8] large-k (sort) path: score a prefix of the rows first and list, of the rest, only pairs that reach its k-th best (-1 / 1 = on, 0 = off)
    int hi_prebuild = -1;         // the batch path's 16-bit hi plane is built (extended) in the background right after appends, off the first batch's
                                  // critical path: -1 = automatic (stores of 262144 rows and more, while the plane takes at most a quarter of the free
                                  // HBM), 0 = never (built by the first batch query or ott_store_prepare_batch), 1 = always
    int stage_appends = -1;       // appends below 256 KB are staged in pinned host memory and sent to the GPU 4 MB at a time (-1 / 1 = on, 0 = every append goes at once)
    int small_sort = -1;          // results of up to 16384 (row, query) pairs sorted by rank in two launches (-1 / 1 = on, 0 = the radix sort path)
    int eps_scale_ppm = 1000000;  // TEST ONLY: the batch path's error bound multiplied by this many millionths (a deliberately-too-small bound
                                  // must be noticed by the measured |approximate - exact| / eps and answered by the next cascade level)
    int multi_transport = 0;      // multi-GPU store (ott_store_create_multi): how the shards' candidate blocks reach the merging GPU.
                                  // 0 = automatic (RCCL all-gather when the device ordinals are distinct and librccl loads, peer copies
                                  // otherwise), 1 = peer copies (hipMemcpyPeerAsync + events), 2 = RCCL (ncclCommInitAll + grouped ncclAllGather)
    int multi_rebalance = 1;      // multi-GPU store: 1 = rows are moved between the shards (before a query, after appends) when one shard holds
                                  // more than 1.25x its even share; 0 = never (rows stay where the appends put them)
    bool multi_fake_distinct = false;  // TEST ONLY (OTT_MULTI_FAKE_DISTINCT=1, read when the store is created): every shard of a multi-GPU
                                  // store counts as a device of its own although the ordinals repeat — the exchange, the row moves and
                                  // device appends then take the code paths of distinct GPUs (send buffer + hipMemcpyPeerAsync + event
                                  // wait, or the grouped all-gather) on a one-GPU box.  Results never depend on it.
    int multi_min_shard_rows = 32768;  // multi-GPU store: a shard is only brought in for this many rows (a store of fewer than twice as many stays
                                  // on its first GPU and is answered by that shard alone: the fan-out over N GPUs costs 50-150 us per query,
                                  // more than a small store's whole query); 0 = always split evenly over all shards
};
void options_from_env(Options& o);                                   // ott_store.hip; called by ott_store_create only
int option_set(Options& o, const char* name, long long value);       // 0, or -1 for an unknown name / bad value

// librccl.so.1, dlopen'ed on first use (ott_comm.hip): the library loads — and every single-GPU entry point works — without it
struct NcclId {
    char internal[OTT_COMM_ID_BYTES];
};
struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(NcclId*) = nullptr;
    int (*CommInitRank)(void**, int, NcclId, int) = nullptr;
    int (*CommInitAll)(void**, int, const int*) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;
    int (*GetVersion)(int*) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string why;  // load failure
};
constexpr int kNcclUint8 = 1;  // ncclDataType_t::ncclUint8 (rccl.h)
Rccl* rccl();

struct Column {
    uint32_t dtype;
    void* d_vals;
    uint64_t* d_nulls;  // may be null
    uint64_t n;
};

}  // namespace ott

// A run of consecutive surviving chunks: local rows [start, start+count).
struct ott_run {
    uint64_t start;
    uint64_t count;
};

struct ott_multi;  // ott_multi.hip: the shards of a store that spans several GPUs (ott_store_create_multi)

struct ott_store {
    ott_multi* multi = nullptr;  // set: this object is the FRONT of a multi-GPU store — dim / n / chunk_size / base_offset / opt / rw
                                 // are the whole store's, the rows live in multi->shards (every extern "C" entry point dispatches on it)
    int device = 0;
    int logical = 0;             // logical device id (= device, unless the multi-GPU store was created with "multi_fake_distinct"): see use_device
    uint32_t dim = 0;
    uint32_t ld = 0;    // row pitch in floats (dim rounded up to 4: rows are 16-B aligned)
    uint32_t dimq = 0;  // query pitch in floats (dim rounded up to 8)
    uint64_t n = 0, cap = 0;
    uint64_t chunk_size = 1024;
    uint64_t base_offset = 0;
    uint32_t reduce = OTT_REDUCE_AVX;
    int n_cu = 256;
    float min_pos_inv = __builtin_inff();  // smallest non-zero inverse norm appended so far (1/max row norm)
    ott::Options opt;

    float* d_rows = nullptr;  // [cap * ld]
    float* d_inv = nullptr;   // [cap]
    uint8_t* d_flag = nullptr;  // [cap] 1 = row norm is inf / NaN / > 1e18 / tiny but non-zero / underflowed (always re-scored exactly by the MFMA path)
    // Batch-path image of the corpus: every row pre-split into bf16 hi + bf16 lo, per 32-k stage [32 hi | 32 lo] (the same
    // 128 B a stage of f32 takes; row pitch = dim rounded up to 32 floats).  Built lazily by the first batch query, extended
    // after appends, dropped on write_rows / reallocation; doubles the store's HBM footprint (skipped when it does not fit:
    // the kernel then splits the f32 rows in registers).  Owner store only; guarded by img_mu.
    uint16_t* d_img = nullptr;
    uint64_t img_rows = 0, img_cap = 0;
    bool img_off = false;
    std::mutex img_mu;
    // Hi plane: the bf16 (round-to-nearest) value of every element, row pitch = dim rounded up to 64 elements — HALF the
    // corpus bytes.  The batch path's first candidate pass streams only this (one bf16 MFMA per 16 k); `imgh_rel` is the
    // MEASURED max over regular rows of ||v - bf16(v)|| / ||v|| (hi_rows_kernel), which is what makes that pass's error
    // bound rigorous and ~2.4x tighter than the worst case 2^-8.  Same life cycle as d_img (the split image is then only
    // built when a query falls through to the split pass).
    uint16_t* d_imgh = nullptr;
    uint64_t imgh_rows = 0;
    bool imgh_f16 = false;     // the plane holds IEEE half (round 3: 11 significant bits, ~8x tighter bound) instead of bf16
    float imgh_scale = 1.0f;   // half only: the power-of-two factor every row was multiplied by before the conversion
    bool imgh_off = false;
    uint32_t* d_imgh_rel = nullptr;  // device word behind imgh_rel (float bits, atomicMax)
    float imgh_rel = 0.0f;
    // Int8 plane (round 5, option hi_fmt = 2): every row as int8 with ONE f32 scale per row (s_v = max|v_i| / 127, element =
    // rint(v_i / s_v)), row pitch = dim rounded up to 128 bytes — a QUARTER of the corpus bytes.  The batch path's cheapest
    // candidate pass streams it (v_mfma_i32_32x32x32_i8: the integer accumulation is exact, so the pass's error bound is pure
    // quantisation, MEASURED per row when the plane is built: `img8_rel` = max over the regular rows of ||v - s_v v~|| / ||v||).
    // Rows that measure more than 2^-5 (one huge element among small ones) are marked irregular (bit 2 of d_flag): always
    // listed, always re-scored exactly.  Same life cycle as the hi plane; guarded by img_mu.
    // d_flag bytes (bit 1: the half plane's factor does not suit the row; bit 2: int8 loses too much of it): ONE writer at a time —
    // the plane builders, under img_mu — while other contexts' kernels read them.  Readers tolerate either value of a bit that is
    // being set or taken back: a set bit only forces the row into the candidate list and its exact re-score (never changes a result),
    // and a bit is cleared only together with dropping the plane whose passes consult it (ensure_i8_plane's rollback).
    int8_t* d_img8 = nullptr;
    float* d_img8_scale = nullptr;   // [cap] s_v
    uint64_t img8_rows = 0;
    bool img8_off = false;
    uint32_t* d_img8_rel = nullptr;  // [0] running max of the measured loss (float bits), [1] rows marked irregular
    float img8_rel = 0.0f;
    // hi-pass back-off: a batch in which ANY query falls through pays for both passes (the split pass streams the whole corpus
    // again for the few), so the hi pass only pays while fewer than ~half the batches need the second one.  When more than 1/8
    // of a batch falls through, or more than half of the recent batches needed the split pass, the next `hi_skip` batches go
    // straight to it; the skip doubles (4 .. 64) while re-probes keep failing
    std::atomic<int> hi_skip{0}, hi_backoff{0};
    std::atomic<int> i8_skip{0}, i8_backoff{0};  // the same back-off for the int8 level in front of it
    std::atomic<int> i8_t512{0};                 // calls left for which the int8 level re-scores 512 candidates per query (it failed queries at fewer)
    std::atomic<int> i8_fail_ema{0};             // share (x1024, exponential average) of recent int8-level batches that needed a second pass at all
    std::atomic<int> spec_skip{0};    // batches left that run with conservative gates (a speculative gate failed a query recently)
    std::atomic<int> spec_backoff{0};
    std::atomic<int> wide_first{0};   // batches left that start at the 4096-candidate level (the 512-candidate one kept failing)
    std::atomic<int> hi_t512{0};      // the hi pass re-scores 512 candidates per query on this store (it failed queries at 2k + 56: dense neighbourhoods)
    std::atomic<int> hi_fail_ema{0};  // share (x1024, exponential average) of recent hi-pass batches that needed the split pass at all

    hipStream_t stream = nullptr;
    hipEvent_t ev[7] = {};  // 0-2 batch path timing, 3-5 exact path timing, 6 multi-GPU store: this shard's candidate block is ready

    // per-query scratch
    ott::DevBuf d_queries, d_qinv, d_rowmask, d_runs, d_prefix, d_lists, d_lists2, d_hits, d_count, d_cand, d_misc;  // d_lists2: first stage of the two-stage merge
    ott::DevBuf d_minpos;    // device word behind min_pos_inv
    // MFMA path scratch
    ott::DevBuf m_Q, m_qinv, m_qnorm, m_tau, m_cntA, m_cntB, m_candA, m_candB, m_over, m_out, m_outcnt, m_uncert, m_prefix;
    ott::DevBuf l_keysA, l_keysB, l_qA, l_qB, l_tmp, l_cursor, l_hist, l_gate;
    ott::DevBuf l_ctl;         // rank-sort path (small results): cursor | tickets | ranks | histogram, kept zeroed between queries
    bool l_ctl_clean = false;  // the last query on this context left l_ctl zeroed
    // sort path, results of a million hits and more: written straight into the caller's host buffer (query_core sets
    // direct_out / direct_cap for the call; run_large_k sets direct_done and the groups' counts when it used them)
    ott_hit* direct_out = nullptr;
    uint64_t direct_cap = 0;
    bool direct_done = false;
    std::vector<uint64_t> direct_counts;  // large-k (sort) path
    ott::DevBuf x_send, x_recv;  // sharded queries: this shard's candidate block, the gathered blocks of all shards
    ott::DevBuf d_evalmask;  // mask built by ott_store_eval_row_mask
    uint64_t evalmask_bits = 0;
    // Small appends are STAGED: rows of appends below 256 KB (VecStore::add_vector is one row per call, src/vec.rs:357-371)
    // collect in pinned host memory and go to the GPU together — when 4 MB are full, and before anything looks at the rows
    // (queries, reads, columns, other kinds of append).  A single-row append costs a memcpy instead of a copy + a kernel + a
    // wait (60 us -> well under 1 us); results never depend on it.  Guarded like the rows themselves (exclusive `rw`).
    ott::PinBuf h_pend;            // the pinned memory behind `pend`
    ott::host::StagedRows pend;    // rows staged, not yet in HBM (VecStore::len counts them)
    ott::PinBuf h_stage, h_hits, h_hdr;  // h_hdr: this shard's block header of a sharded query (ott_comm.hip)
    size_t in_off_qinv = 0, in_off_runs = 0, in_off_prefix = 0;  // layout of the per-query input block in d_queries
    size_t res_hits_off = 0;                                       // hits offset inside d_hits (counts come first)
    // candidate order of the query this context is running right now (query_core sets them, the launch wrappers read them)
    uint32_t cur_tie_sh = 0, cur_tie_off = 0;
    bool cur_flat = false;

    ott::host::QuietWorker* builder = nullptr;  // ott_store.hip: the background thread behind option hi_prebuild (owner stores only)
    std::vector<ott::Column> columns;
    // Concurrency (ott_query is re-entrant on a store from several host threads, append needs exclusive
    // access).  `rw`: queries hold it shared, everything that changes the store holds it exclusive.  `mu` guards ONE query
    // context = this struct's stream, events and scratch.  When a query arrives while `mu` is taken, it runs on a worker
    // context instead: a second ott_store that aliases the corpus pointers and owns its own stream + scratch (created on
    // first need, at most OTT_MAX_WORKERS), so concurrent callers overlap on the GPU instead of queueing on a lock.
    ott::host::RwGate rw;
    std::mutex mu;
    ott::host::ContextPool<ott_store> pool;  // the worker contexts (ott_host.h)
    bool is_worker = false;
    ott_store* owner = nullptr;            // workers: the store they belong to
};

namespace ott {
inline hipError_t use_device(const ott_store* s) { return use_device_raw(s->device, s->logical); }
int store_create(uint32_t dim, int device, int logical, ott_store** out);  // ott_store.hip: ott_store_create with a logical device id
// ott_multi.hip: the multi-GPU store behind the single-store entry points
enum AppendKind { APPEND_HOST = 0, APPEND_DEVICE = 1, APPEND_RANDOM = 2, APPEND_CLUSTERED = 3 };
struct AppendArgs {
    int kind = APPEND_HOST;
    const void* rows = nullptr;
    uint64_t seed = 0;
    uint32_t n_clusters = 0;
    float spread = 0.f, aniso = 0.f;
};
int multi_destroy(ott_store* ms);
int multi_reserve(ott_store* ms, uint64_t n_rows);
int multi_append(ott_store* ms, const AppendArgs& a, uint64_t n_rows);
int multi_write_rows(ott_store* ms, uint64_t first_row, const float* rows_host, uint64_t n_rows);
int multi_read(const ott_store* ms, bool inv_norms, uint64_t first_row, uint64_t n_rows, float* out_host);
int multi_set_chunk_size(ott_store* ms, uint64_t chunk_size);
int multi_set_base_offset(ott_store* ms, uint64_t base);
int multi_set_reduce_order(ott_store* ms, uint32_t reduce);
int multi_set_batch_image(ott_store* ms, int enabled);
int multi_set_option(ott_store* ms, const char* name, int64_t value);
int multi_prepare_batch(ott_store* ms);
int multi_sync(ott_store* ms);
int multi_batch_ready(ott_store* ms);
int multi_add_column(ott_store* ms, uint32_t dtype, const void* values_host, const uint64_t* nulls, uint64_t n, uint32_t* out_column_id);
int multi_eval_row_mask(ott_store* ms, const ott_leaf* leaves, uint32_t n_leaves, uint32_t n_clauses, uint64_t* out_host);
int multi_zone_stats(ott_store* ms, uint32_t column, uint64_t chunk_size, void* out_min, void* out_max, uint64_t* out_non_null);
int multi_query(ott_store* ms, const ott_query_desc* d, ott_hit* out, uint64_t cap, uint64_t* n_out, uint64_t* n_per_query, ott_stats* stats);
// ott_store.hip: staged appends.  store_rows = rows appended (resident + staged); store_flush brings the staged ones to the GPU
// (takes the store exclusively when there are any; call it WITHOUT holding the store's locks)
void mfma_warm(hipStream_t stream, int device);  // ott_mfma.hip: loads the batch path's code object and warms the runtime's H2D copy path (background plane builder)
void kick_plane_build(ott_store* s);  // ott_store.hip: rows were appended — (re)build the hi plane in the background if the policy says so
inline uint64_t store_rows(const ott_store* s) { return s->n + s->pend.count(); }
int store_flush(ott_store* s);
int store_flush_locked(ott_store* s);  // the caller holds `rw` exclusively and `mu`
// ott_store.hip: a shard takes over freshly filled buffers (rows moved between the GPUs of a multi-GPU store)
int store_adopt(ott_store* s, float* rows, float* inv, uint8_t* flag, uint64_t n, uint64_t cap);

constexpr size_t OTT_MAX_WORKERS = 15;
ott_store* ctx_acquire(ott_store* s);  // returns s or a worker, with its `mu` held
void ctx_release(ott_store* w);
int ensure_batch_image(ott_store* ctx, const uint16_t** img_out);
int ensure_hi_plane(ott_store* ctx, const uint16_t** img_out, float* rel_max_out, bool* f16_out = nullptr, float* scale_out = nullptr);  // *img_out = nullptr when unavailable
bool hi_plane_ready(ott_store* ctx);  // the plane exists and covers every row (nothing is built by asking)
inline bool i8_wanted(const Options& o) { return (o.hi_fmt == -1 || o.hi_fmt == 2) && !o.mfma_f32 && !o.no_hi_pass && !o.no_batch_image; }
// the plane the cascade STARTS with on this store — the int8 plane where the options ask for it and it fits, else the hi plane —
// built / extended now (background builder, ott_store_prepare_batch); an existing hi plane is kept up to date beside it
int ensure_first_plane(ott_store* ctx);
bool first_plane_ready(ott_store* ctx);
// What the planes look like right now, read under img_mu (ensure_i8_plane / ensure_hi_plane change these fields from other query
// contexts while the store is only held shared): the AUTO cost model, the background builder and kick_plane_build decide from this
// snapshot.  A snapshot may be stale by the time it is used — every consumer only chooses a path or skips a build from it, and the
// ensure_* calls that follow re-read the fields under the mutex.
struct PlaneSnapshot {
    bool have_i8, i8_off, have_hi, hi_f16, hi_off, img_off;
    uint64_t i8_rows, hi_rows;
};
PlaneSnapshot plane_snapshot(const ott_store* s);
// the store's int8 plane (option hi_fmt = -1 / 2), built / extended on demand; *img_out = nullptr when it is unavailable
int ensure_i8_plane(ott_store* ctx, const int8_t** img_out, const float** scale_out, float* rel_max_out);
bool i8_plane_ready(ott_store* ctx);
// f32 rows -> int8 rows of pitch ld8 bytes.  common_scale > 0: every row quantised with THAT scale (the query operand block: one
// scale per batch, so that it folds into the kernel's row factors); else per row max|x| / 127, written to scale_out[r].
// pre[r] (optional): the row is multiplied by it first (cosine: 1 / ||q||).  rel_out[r] (optional) = ||x - s x~|| / ||x||.
int launch_i8_rows(hipStream_t stream, const float* rows, uint32_t ld, uint32_t dim, uint32_t ld8, uint64_t first, uint64_t n, int8_t* out,
                   const float* pre, float common_scale, float* scale_out, float* rel_out, uint32_t* rel_max, const uint8_t* flag, float rel_flag,
                   uint8_t* flag_rw, int n_cu);
// f32 rows -> bf16 (RNE) rows of pitch ldh elements (optionally row-scaled first); rel_out[r] (optional) = ||x - bf16(x)|| / ||x||
int launch_hi_rows(hipStream_t stream, const float* rows, uint32_t ld, uint32_t dim, uint32_t ldh, uint64_t n, uint16_t* out,
                   const float* scale, float* rel_out, int n_cu, bool f16 = false, float gscale = 1.0f);  // f16: IEEE half of x * scale[r] * gscale (gscale a power of two)
int launch_split_rows(hipStream_t stream, const float* rows, uint32_t ld, uint32_t dim, uint32_t ldi, uint64_t n, uint16_t* out,
                      const float* scale, int n_cu);  // f32 rows -> [32 hi | 32 lo] bf16 per 32-k stage (optionally row-scaled first)  // ott_store.hip; *img_out = nullptr when unavailable
}  // namespace ott

namespace ott {

// ---- key transform shared by all top-k code ---------------------------------------------------
// total order on f32 bits (== Rust f32::total_cmp): ascending unsigned == ascending total_cmp
__host__ __device__ inline uint32_t total_key(float f) {
    uint32_t b;
#if defined(__HIP_DEVICE_COMPILE__)
    b = __float_as_uint(f);
#else
    __builtin_memcpy(&b, &f, 4);
#endif
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
// "larger is better" ordinal for the requested take type; 0 is never produced by a non-NaN
__host__ __device__ inline uint32_t ord_of(float f, bool take_max) {
    uint32_t k = total_key(f);
    return take_max ? k : ~k;
}
__host__ __device__ inline float score_of(uint32_t ord, bool take_max) {
    uint32_t k = take_max ? ord : ~ord;
    uint32_t b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    float f;
#if defined(__HIP_DEVICE_COMPILE__)
    f = __uint_as_float(b);
#else
    __builtin_memcpy(&f, &b, 4);
#endif
    return f;
}

// One candidate in a block/partial list: key = ord<<32 | ~local_row  (bigger = better; lower
// row wins ties), q = query id (lower wins remaining ties).  16 bytes.
struct Cand {
    uint64_t key;
    uint32_t q;
    uint32_t pad;
};

// ---- launch wrappers (defined in the .hip files) ----------------------------------------------
struct ExactParams {
    const float* rows;
    const float* inv;
    const float* queries;  // [nq_total * dimq], zero padded
    const float* qinv;     // [nq_total]
    const uint64_t* row_mask;
    const ott_run* runs;
    const uint32_t* tile_prefix;  // [n_runs + 1]
    Cand* lists;                   // [n_lists_per_block? see ott_exact.hip]
    uint64_t row_mask_bits;
    uint32_t ld, dim, dimq;
    uint32_t n_runs, n_tiles;
    uint32_t q0, nq_total;  // first query of this pass, total queries
    uint32_t metric, take_max, cmp, reduce;
    float thr;
    uint32_t k;      // <= 64*E
    uint32_t perq;   // 1 = one list per query
    uint32_t list_stride;  // entries between consecutive lists in `lists`
    // single-query launches carry their inputs IN the kernel arguments (no H2D copy in front of the launch): the query
    // (zero padded to dimq), its inverse norm, and up to two runs with their tile prefix
    uint32_t tie_sh;  // 0 = canonical tie order (score, row, query); 3 = the reference's visit order (score, row >> 3, query, row & 7)
    uint32_t tie_off; // 0..7, added to the row in every candidate key: the 8-row blocks of the visit order are then counted from row
                      // (8 - tie_off) % 8 of the store instead of row 0 — a chunk whose first row is not a multiple of 8 (tie_order 2,
                      // src/meta_compute.rs:153-192: every chunk is a VecStore of its own); whoever turns keys back into rows subtracts it (tie_base)
    uint32_t flat;    // 1 = every passing score ranks the same: the list keeps the first k passing pairs in visit order (tie_sh = 3)
    uint32_t embedded;
    uint32_t small;  // 1 = small-grid kernel variant (single query, one tile per one-wave workgroup), 2 = rows8 (eight lanes per row, one 8-wave workgroup per tile)
    float eqinv;
    uint32_t eprefix[3];
    ott_run eruns[2];
    // large-k path: every passing (key, query) is appended here instead of the fused top-k
    uint64_t* dump_keys;
    uint32_t* dump_q;
    unsigned long long* dump_cursor;
    uint64_t dump_cap;
    const uint32_t* dump_gate;  // [nq_total] score ordinals a pair must reach to be listed (nullptr = none)
    // int8 candidate sweep of a SINGLE query (round 5, exact_kernel<..., I8>): `rows` is the store's int8 plane (ld = dim = dimq =
    // its row pitch in 4-byte units), the query is int8 too (embedded in qemb), scores are APPROXIMATE — (float)(q~ . v~) x
    // (s_v [x 1/||v||]) x s_Q — and the list keeps the T = k best of them for the exact re-score (run_i8_single, ott_mfma.hip)
    uint32_t i8;
    float i8_qscale;        // s_Q: the query's quantisation scale
    const float* i8_scale;  // [n] s_v
    const uint8_t* flag;    // [n] rows outside the pass's error model (bits 0, 2): always listed, ranked first
    float qemb[896];  // last: the embedded query (kernel arguments are limited to 4 KB)
};
constexpr uint32_t OTT_QEMB_MAX = 896;
static_assert(sizeof(ExactParams) <= 4096, "kernel arguments are limited to 4 KB");

int launch_exact(ott_store* s, const ExactParams& p, int nq_tile, int E, int grid);
int launch_exact_i8(ott_store* s, const ExactParams& p, int E, int grid);  // p.i8 = 1: one query, approximate int8 scores, list of 64 E candidates
int launch_exact_dump(ott_store* s, const ExactParams& p, int nq_tile, int grid);  // nq_tile: 1 or 4
int exact_grid(const ott_store* s, uint32_t n_tiles);
// merges `n_lists` sorted partial lists of k entries (stride list_stride) per output group
// (groups = 1 for MERGED, nq for PER_QUERY; group g's lists start at g*group_stride) into
// out_hits[g*out_stride ...] and out_counts[g]
int launch_merge(ott_store* s, const Cand* lists, uint32_t n_lists, uint32_t list_stride, uint64_t group_stride,
                 uint32_t groups, uint32_t k, int E, bool take_max, uint64_t base_offset, ott_hit* out_hits,
                 uint64_t out_stride, uint64_t* out_counts, uint32_t tie_sh);

// hdr_slots: ott_hit-sized header slots behind every rank's [n_groups][list_len] hits (the blocks of ott_query_sharded); they are
// copied to hdr_out ([n_lists][hdr_slots], e.g. pinned host memory) by the same launch
int launch_merge_hits(ott_store* s, const ott_hit* lists, uint32_t n_lists, uint32_t n_groups, uint32_t list_len, uint32_t k, int E,
                      bool take_max, ott_hit* out, uint64_t* count, uint32_t hdr_slots = 0, ott_hit* hdr_out = nullptr);

int launch_inv_norms(ott_store* s, uint64_t first_row, uint64_t n_rows);
int update_min_pos_inv(ott_store* s, uint64_t first_row, uint64_t n_rows);  // call after launch_inv_norms; syncs

// surviving chunk runs of one query (candidate_chunks, src/meta.rs:648-659)
struct RunPlan {
    std::vector<ott_run> runs;
    uint64_t rows_scored = 0, total_chunks = 0, evaluated = 0;
};
std::vector<uint32_t> tile_prefix(const RunPlan& pl, uint32_t tile_rows);

// large-k path (k > 512): score dump + device radix sort.  Entries [0, *n_entries) of (l_keysA|B, l_qA|B) are left sorted in
// canonical order (merged) or grouped by query (per-query); results are copied to `lists`.
int run_large_k(ott_store* s, const float* queries, uint32_t nq, const ott_query_desc* d, bool perq, const RunPlan& pl, uint64_t k_eff,
                const uint64_t* d_mask, uint64_t mask_bits, std::vector<std::vector<ott_hit>>& lists, ott_stats& st);
void fill_exact_params(ott_store* s, const ott_query_desc* d, const RunPlan& pl, uint32_t nq, const uint64_t* d_mask, uint64_t mask_bits,
                       uint32_t n_tiles, ExactParams& p);
float host_inv_norm_exact(const float* v, uint32_t dim);  // ott_api.hip: the query-side inverse norm in the reference's order
int upload_exact_inputs(ott_store* s, const float* queries, uint32_t nq, const RunPlan& pl, const std::vector<uint32_t>& prefix);

// MFMA batch path: per-query exact top-k lists on the host; uncertified[q] != 0 means the
// list for q could not be certified and must be recomputed on the exact path.
// level 0 = hi pass (bf16 hi plane, one MFMA per 16 k; needs mfma_hi_ok), level 1 = split-bf16 / f32-pipe pass
int run_mfma(ott_store* s, const ott_query_desc* d, const RunPlan& pl, uint64_t k_q, const uint64_t* d_mask, uint64_t mask_bits,
             std::vector<std::vector<ott_hit>>& out, std::vector<uint32_t>& uncertified, ott_stats& st, int level, uint32_t t_min,  // t_min: re-score at least this many (0, 512, 4096)
             bool spec_gate);  // speculative emission thresholds between the row rounds (select_kernel); first level of a cascade only
// the hi pass re-scores T >= 2k + 56 candidates per query (T <= 512); on a half plane, whose bound is ~8x tighter, k + k / 3 + 28
// is enough (k <= 363 instead of 228: about 0.23 k rows lie within the bound of the k-th score on uniform rows)
inline bool mfma_hi_k_ok(uint64_t k, bool half) { return half ? k + k / 3 + 28 <= 512 : 2 * k + 56 <= 512; }
int launch_rand_fill(ott_store* s, uint64_t first_row, uint64_t n_rows, uint64_t seed);
// ott_mfma.hip: the int8 level for ONE query as a streaming sweep (no matrix cores: the query is a vector) — exact_kernel<..., I8>
// over the int8 plane with a wave-list top-T in its epilogue, merge, exact re-score + certification (finalize_kernel): three
// launches and one wait instead of the cascade's five rounds.  Same outputs as run_mfma.
int run_i8_single(ott_store* s, const ott_query_desc* d, const RunPlan& pl, uint64_t k_q, const uint64_t* d_mask, uint64_t mask_bits,
                  std::vector<std::vector<ott_hit>>& out, std::vector<uint32_t>& uncertified, ott_stats& st, uint32_t t_min);

// canonical result order shared with the oracle: better score (total order on the bits), lower row, lower query.
// sh = 3 (tie_order = reference): better score, then the reference's visit order — 8-row block, query, row within the block
// (`base`: the store's base offset; blocks are counted from the store's first row, src/vec.rs:222-303)
struct CanonLess {
    bool tmax;
    uint32_t sh = 0;
    uint64_t base = 0;
    bool operator()(const ott_hit& a, const ott_hit& b) const {
        const uint32_t ka = ord_of(a.score, tmax), kb = ord_of(b.score, tmax);
        if (ka != kb) return ka > kb;
        if (sh) {
            const uint64_t ba = (a.index - base) >> sh, bb = (b.index - base) >> sh;
            if (ba != bb) return ba < bb;
            if (a.query != b.query) return a.query < b.query;
            return a.index < b.index;
        }
        if (a.index != b.index) return a.index < b.index;
        return a.query < b.query;
    }
};
// G-way merge of lists that are each in `less` order: the next n hits go to dst, the heads move on.  The heads' score ordinals
// are kept beside them (smaller = better): the scan compares integers and falls back to the full order only between equal
// scores (recomputing both ordinals in every comparison cost 22 ns per hit on 8 lists).  The caller guarantees n <= the hits left.
inline void merge_heads(std::vector<const ott_hit*>& head, const std::vector<const ott_hit*>& end, const CanonLess& less, ott_hit* dst, uint64_t n) {
    const size_t G = head.size();
    std::vector<size_t> act(G);
    std::vector<uint32_t> key(G);
    size_t n_act = 0;
    for (size_t g = 0; g < G; g++)
        if (head[g] != end[g]) {
            key[n_act] = ~ord_of(head[g]->score, less.tmax);
            act[n_act++] = g;
        }
    for (uint64_t i = 0; i < n; i++) {
        size_t bi = 0;
        for (size_t a = 1; a < n_act; a++)
            if (key[a] < key[bi] || (key[a] == key[bi] && less(*head[act[a]], *head[act[bi]]))) bi = a;
        const size_t b = act[bi];
        dst[i] = *head[b]++;
        if (head[b] == end[b]) {
            --n_act;
            act[bi] = act[n_act];
            key[bi] = key[n_act];
        } else {
            key[bi] = ~ord_of(head[b]->score, less.tmax);
        }
    }
}
inline int list_E(uint64_t k) { return k <= 64 ? 1 : k <= 128 ? 2 : k <= 256 ? 4 : 8; }  // register list entries per lane for k <= 512

// ott_api.hip.  validate_query: argument checks of ott_query.  query_on: one query on a context whose `mu` the caller
// holds (and the owner's `rw`, shared).  Device output (out_dev != nullptr): [groups][cap / groups] slots, sentinel
// padded.  nosync: do not wait for the stream before returning — on the EXACT path nothing then waits on the host at all
// (*events_pending tells the caller to read the kernel timing events ev[3..5] after its own synchronisation).
int validate_query(const ott_store* s, const ott_query_desc* d);
int query_on(ott_store* s, const ott_query_desc* d, ott_hit* out_host, void* out_dev, uint64_t cap, uint64_t* n_out, uint64_t* n_per_query,
             void* n_out_dev, ott_stats* stats_out, bool nosync = false, bool* events_pending = nullptr);
// One plain query on a context (what query_on was before the tie orders): `tie_sh` selects the candidate order of every
// kernel and host merge of this call, `flat` makes every passing score rank the same (EXACT path only: the first k passing
// pairs in visit order).  Host output only when either is set.
struct CoreOpts {
    uint32_t tie_sh = 0;
    bool flat = false;
    uint32_t tie_off = 0;  // with tie_sh = 3: 8-row blocks are counted from local row (8 - tie_off) % 8 (see ExactParams::tie_off)
};
// the base that turns a candidate key's row field back into a global index, and from which CanonLess counts 8-row blocks
inline uint64_t tie_base(const ott_store* s) { return s->base_offset - s->cur_tie_off; }
int query_core(ott_store* s, const ott_query_desc* d, ott_hit* out_host, void* out_dev, uint64_t cap, uint64_t* n_out, uint64_t* n_per_query,
               void* n_out_dev, ott_stats* stats_out, bool nosync, bool* events_pending, const CoreOpts& co);
// ott_ties.hip is written against this: where the candidate lists of a store come from.  `run`: one plain query over what `d`
// selects with take count k, candidates ranked (score, visit order: tie_sh = 3), `flat` = every passing score ranks the same
// (EXACT path); host vectors, PER_QUERY lists concatenated in query order with their counts in `per`.  `run_chunk`: the same
// restricted to ONE chunk (counted from `base`; whatever chunk mask `d` carries is replaced).
struct TieEnv {
    // (run_chunk ranks equal scores by the CHUNK's own 8-row blocks — counted from the chunk's first row, whatever the chunk size:
    //  the chunk is a VecStore of its own in the reference, src/meta_compute.rs:153-192)
    typedef std::function<int(const ott_query_desc& d, uint64_t k, bool flat, std::vector<ott_hit>& out, std::vector<uint64_t>& per, ott_stats* st)> Runner;
    bool tmax = true;
    uint64_t base = 0;        // global index of the (whole) store's first row: 8-row blocks and chunks are counted from here
    uint64_t chunk_size = 1024;
    uint32_t dim = 0;
    Runner run;
    std::function<int(uint64_t chunk, const ott_query_desc& d, uint64_t k, bool flat, std::vector<ott_hit>& out, std::vector<uint64_t>& per, ott_stats* st)> run_chunk;
};
int ref_ties_collect(const TieEnv& env, int tie_order, const ott_query_desc* d, ott_hit* out_host, uint64_t cap, uint64_t* n_out, uint64_t* n_per_query,
                     ott_stats* stats_out);
bool ties_ambiguous(bool tmax, const std::vector<ott_hit>& L, uint64_t k);
int ties_resolve(ott_store* s, bool tmax, uint64_t base, const std::vector<ott_hit>& L, uint64_t k, const std::vector<ott_hit>* fill,
                 std::vector<ott_hit>& out);
// ott_ties.hip: the reference's outcome at exact score ties (store option tie_order = 1 / 2), host output
int query_ref_ties(ott_store* s, const ott_query_desc* d, ott_hit* out_host, uint64_t cap, uint64_t* n_out, uint64_t* n_per_query,
                   ott_stats* stats_out);
void read_exact_events(ott_store* s, ott_stats* st);  // after a synchronisation: score_ns / merge_ns from ev[3..5]
int launch_pack_rows(ott_store* s, const float* dense_dev, uint64_t first_row, uint64_t n_rows);

}  // namespace ott

// ott_api.hip — ott_query: the body of VecQueryPlan::collect (src/vec.rs:206-311) and the
// score + merge block of MetaQueryPlan::collect (src/meta.rs:671-709) on one MI355X.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <chrono>

#include "ott_internal.h"

using namespace ott;

namespace ott {

// query-side inverse norm, src/vec.rs:390-396 (sequential f32 sum, sqrt, reciprocal; the file
// is built with -ffp-contract=off, `volatile` keeps the compiler from widening or reordering)
float host_inv_norm_exact(const float* v, uint32_t dim) {
    volatile float s = 0.0f;
    for (uint32_t i = 0; i < dim; i++) {
        volatile float sq = v[i] * v[i];
        s = s + sq;
    }
    float norm = sqrtf(s);
    return norm != 0.0f ? 1.0f / norm : 0.0f;
}

std::vector<uint32_t> tile_prefix(const RunPlan& pl, uint32_t tile_rows) {
    std::vector<uint32_t> pre;
    pre.push_back(0);
    uint64_t tiles = 0;
    for (const auto& r : pl.runs) {
        tiles += (r.count + tile_rows - 1) / tile_rows;
        pre.push_back((uint32_t)tiles);
    }
    return pre;
}

}  // namespace ott


namespace ott {

int upload_exact_inputs(ott_store* s, const float* queries, uint32_t nq, const RunPlan& pl, const std::vector<uint32_t>& prefix) {
    const uint32_t nq_pad = (nq + 7u) & ~7u;  // the kernels read whole NQ-wide query blocks: pad with zero rows
    const size_t q_bytes = (size_t)nq_pad * s->dimq * 4, qi_bytes = (size_t)nq_pad * 4;
    const size_t run_bytes = pl.runs.size() * sizeof(ott_run), pre_bytes = prefix.size() * 4;
    size_t off_q = 0, off_qi = off_q + q_bytes, off_run = (off_qi + qi_bytes + 15) & ~(size_t)15;
    size_t off_pre = off_run + run_bytes, total = off_pre + pre_bytes;
    int rc = s->h_stage.ensure(total);
    if (rc) return rc;
    char* hs = (char*)s->h_stage.p;
    float* hq = (float*)(hs + off_q);
    memset(hq, 0, q_bytes + qi_bytes);
    for (uint32_t i = 0; i < nq; i++) {
        memcpy(hq + (size_t)i * s->dimq, queries + (size_t)i * s->dim, (size_t)s->dim * 4);
        ((float*)(hs + off_qi))[i] = host_inv_norm_exact(queries + (size_t)i * s->dim, s->dim);
    }
    memcpy(hs + off_run, pl.runs.data(), run_bytes);
    memcpy(hs + off_pre, prefix.data(), pre_bytes);
    // one device block, one copy: [queries | qinv | runs | tile prefix]
    if ((rc = s->d_queries.ensure(total))) return rc;
    OTT_HIP(hipMemcpyAsync(s->d_queries.p, hs, total, hipMemcpyHostToDevice, s->stream));
    s->in_off_qinv = off_qi;
    s->in_off_runs = off_run;
    s->in_off_prefix = off_pre;
    return OTT_OK;
}

void fill_exact_params(ott_store* s, const ott_query_desc* d, const RunPlan& pl, uint32_t nq, const uint64_t* d_mask, uint64_t mask_bits,
                       uint32_t n_tiles, ExactParams& p) {
    memset(&p, 0, sizeof(p));
    p.rows = s->d_rows;
    p.inv = s->d_inv;
    p.queries = (const float*)s->d_queries.p;
    p.qinv = (const float*)((const char*)s->d_queries.p + s->in_off_qinv);
    p.row_mask = d_mask;
    p.row_mask_bits = mask_bits;
    p.runs = (const ott_run*)((const char*)s->d_queries.p + s->in_off_runs);
    p.tile_prefix = (const uint32_t*)((const char*)s->d_queries.p + s->in_off_prefix);
    p.ld = s->ld;
    p.dim = s->dim;
    p.dimq = s->dimq;
    p.n_runs = (uint32_t)pl.runs.size();
    p.n_tiles = n_tiles;
    p.nq_total = nq;
    p.metric = d->metric;
    p.take_max = d->take == OTT_TAKE_MAX;
    p.cmp = d->filter_cmp;
    p.thr = d->filter_thr;
    p.reduce = s->reduce;
    p.tie_sh = s->cur_tie_sh;
    p.tie_off = s->cur_tie_off;
    p.flat = s->cur_flat ? 1u : 0u;
}

}  // namespace ott

namespace {

uint64_t now_ns() {
    return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

uint32_t pow2ceil(uint32_t v) {
    uint32_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

// chunk mask -> runs of consecutive surviving chunks (candidate_chunks, src/meta.rs:648-659)
void build_plan(const ott_store* s, const uint64_t* chunk_mask, RunPlan& pl) {
    const uint64_t n = s->n, cs = s->chunk_size;
    pl.total_chunks = n ? (n + cs - 1) / cs : 0;
    if (!n) return;
    if (!chunk_mask) {
        pl.runs.push_back({0, n});
        pl.evaluated = pl.total_chunks;
        pl.rows_scored = n;
        return;
    }
    bool open = false;
    for (uint64_t c = 0; c < pl.total_chunks; c++) {
        const bool keep = (chunk_mask[c >> 6] >> (c & 63)) & 1;
        if (keep) {
            const uint64_t start = c * cs;
            const uint64_t len = (n - start) < cs ? (n - start) : cs;
            if (open) pl.runs.back().count += len;
            else pl.runs.push_back({start, len});
            open = true;
            pl.evaluated++;
            pl.rows_scored += len;
        } else {
            open = false;
        }
    }
}

}  // namespace

namespace ott {
int validate_query(const ott_store* s, const ott_query_desc* d) {
    if (!s) return fail(OTT_ERR_INVALID, "ott_query: store is NULL");
    if (!d) return fail(OTT_ERR_INVALID, "ott_query: desc is NULL");
    if (d->nq == 0) return fail(OTT_ERR_INVALID, "No queries provided");  // src/vec.rs:188-190
    if (!d->queries) return fail(OTT_ERR_INVALID, "ott_query: queries is NULL");
    if (d->metric > OTT_METRIC_DOT) return fail(OTT_ERR_INVALID, "ott_query: unknown metric");
    if (d->take > OTT_TAKE_MAX) return fail(OTT_ERR_INVALID, "ott_query: unknown take type");
    if (d->filter_cmp > OTT_CMP_EQ) return fail(OTT_ERR_INVALID, "ott_query: unknown filter comparator");
    if (d->mode > OTT_MODE_PER_QUERY) return fail(OTT_ERR_INVALID, "ott_query: unknown mode");
    if (d->path > OTT_PATH_MFMA) return fail(OTT_ERR_INVALID, "ott_query: unknown path");
    if (d->use_device_row_mask && s->evalmask_bits == 0 && s->n)
        return fail(OTT_ERR_INVALID, "ott_query: use_device_row_mask set but ott_store_eval_row_mask was not called");
    return OTT_OK;
}

void read_exact_events(ott_store* s, ott_stats* st) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s->ev[3], s->ev[4]) == hipSuccess) st->score_ns = (uint64_t)(ms * 1e6);
    if (hipEventElapsedTime(&ms, s->ev[4], s->ev[5]) == hipSuccess) st->merge_ns = (uint64_t)(ms * 1e6);
}
}  // namespace ott

namespace {

// total of the per-group counts a merge launch left in device memory (PER_QUERY device output)
__global__ void sum_counts_kernel(const uint64_t* counts, uint32_t n, uint64_t* out) {
    uint64_t t = 0;
    for (uint32_t i = threadIdx.x; i < n; i += 64) t += counts[i];
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
    if (threadIdx.x == 0) *out = t;
}

// EXACT path.  queries: host [nq*dim].  Results: per group (1 for merged, nq for per-query) on the
// host in `lists`.  `perq` selects the grouping.  If dev_copy != nullptr (merged only) the merged
// list is also left in device memory at s->d_hits (KS slots, sentinel padded) for ott_query_device.
int run_exact(ott_store* s, const float* queries, uint32_t nq, const ott_query_desc* d, bool perq, const RunPlan& pl, uint64_t k_eff,
              const uint64_t* d_mask, uint64_t mask_bits, bool fetch, std::vector<std::vector<ott_hit>>& lists, ott_stats& st,
              bool timing = true, ott_hit* hits_dev_direct = nullptr, uint32_t direct_stride = 0) {
    // hits_dev_direct (device output only): the merge kernel writes its [groups][KS] sentinel-padded hits straight there
    // (direct_stride must be the KS this call derives from k_eff) instead of into d_hits
    if (k_eff > 512) {  // beyond the fused register top-k: score dump + device radix sort
        if (!fetch) return fail(OTT_ERR_UNSUPPORTED, "ott_query_device: k > 512 is host-output only");
        return run_large_k(s, queries, nq, d, perq, pl, k_eff, d_mask, mask_bits, lists, st);
    }
    // 128 < k <= 512 — four and eight list entries per lane — against the sort path (profiles/round3/large_k_from.md).  Round 2:
    // the lists lost from 256 up (9.7 against 6.0 ms at 10M x 768, k = 512), and once the sort path's second phase listed only
    // what can still make the result they lost from 128 up.  With candidates merged into the lists as sorted blocks
    // (wl_merge_sorted) and the block lists merged by rank (merge_rank_kernel) a SINGLE query is faster on the lists at every k
    // they hold and every store size measured (10k rows: 0.16 against 0.23 ms at k = 512; 10M rows: 4.81 against 4.93).  Several
    // queries still go to the sort path from 128 up — its sweep carries four of them, a sweep of the long lists one: 1M rows, 4
    // queries, k = 300: 1.0 against 2.5 ms — unless the store is small (100k rows x 4 queries: lists 0.45 against 0.58 ms).
    {
        const uint64_t pairs = pl.rows_scored * nq;
        const uint64_t from = s->opt.large_k_from > 0 ? (uint64_t)s->opt.large_k_from : ((nq == 1 || pairs <= (1ull << 19)) ? 512u : 128u);
        if (k_eff > from && fetch && pairs <= (1ull << 28))
            return run_large_k(s, queries, nq, d, perq, pl, k_eff, d_mask, mask_bits, lists, st);
    }
    int E = k_eff <= 64 ? 1 : k_eff <= 128 ? 2 : k_eff <= 256 ? 4 : 8;
    const uint32_t KS = 64 * E;
    const std::vector<uint32_t> prefix = tile_prefix(pl, 64);
    const uint32_t n_tiles = prefix.back();
    // 0 = the streaming kernel, 1 = the one-wave small-grid variant (LDS-DMA ring; single query), 2 = rows8: eight lanes per
    // row, one 8-wave workgroup per tile, up to 8 queries per pass (round 3: 10k x 768 in 25-30 us instead of 64; the
    // default wherever a small variant fits: stores of up to 1024 tiles ~ 65k rows, batches of up to 16 queries)
    uint32_t small = 0;
    {
        const int forced = s->opt.exact_small;  // store option: 0 / 1 / 2 forces the choice
        const bool fits1 = nq == 1 && !perq && E <= 2 && s->dimq <= 2048 && n_tiles <= 1024;
        const bool fits8 = E <= 2 && s->dimq <= 2048 && n_tiles <= 1024 && nq <= 16;
        if (forced == 1) small = fits1 ? 1u : 0u;
        else if (forced == 2 || forced < 0) small = fits8 ? 2u : 0u;
    }
    uint32_t tile;
    // queries per corpus pass: 4 is the measured sweet spot of the streaming kernel (1-2 queries 4.95 ms, 4 queries 5.3 ms per
    // pass at 10M x 768; an 8-wide pass needs 233 VGPRs and ran 14 ms, slower than two 4-wide passes); rows8 takes up to 8
    // (one accumulator per query and lane)
    if (E >= 4) tile = 1;
    else tile = pow2ceil(nq) < (small == 2 ? 8u : 4u) ? pow2ceil(nq) : (small == 2 ? 8u : 4u);
    const uint32_t passes = (nq + tile - 1) / tile;
    const int grid = small ? (int)n_tiles : exact_grid(s, n_tiles);

    // single query, at most two runs: everything the kernel needs rides in its arguments (no H2D copy, no staging)
    const bool lean = nq == 1 && s->dimq <= OTT_QEMB_MAX && pl.runs.size() <= 2;
    int rc;
    if (!lean && (rc = upload_exact_inputs(s, queries, nq, pl, prefix))) return rc;
    const size_t n_lists_total = perq ? (size_t)nq * grid : (size_t)passes * grid;
    if ((rc = s->d_lists.ensure(n_lists_total * KS * sizeof(Cand)))) return rc;
    const uint32_t groups = perq ? nq : 1;
    // results block: [counts (groups x u64, padded to 64 B) | hits (groups x KS)].  Host output: the merge kernel
    // writes it straight into pinned host memory (no D2H copy behind the launch); device output: into d_hits
    const size_t cnt_pad = (((size_t)groups * sizeof(uint64_t)) + 63) & ~(size_t)63;
    const size_t res_bytes = cnt_pad + (size_t)groups * KS * sizeof(ott_hit);
    char* res_dev = nullptr;
    if (fetch) {
        if ((rc = s->h_hits.ensure(res_bytes))) return rc;
        void* mapped = nullptr;
        OTT_HIP(hipHostGetDevicePointer(&mapped, s->h_hits.p, 0));
        res_dev = (char*)mapped;
    } else {
        if ((rc = s->d_hits.ensure(res_bytes))) return rc;
        res_dev = (char*)s->d_hits.p;
    }
    uint64_t* d_counts = (uint64_t*)res_dev;
    ott_hit* d_hits = (ott_hit*)(res_dev + cnt_pad);
    s->res_hits_off = cnt_pad;
    if (!fetch && hits_dev_direct != nullptr && direct_stride == KS) d_hits = hits_dev_direct;

    ExactParams p;
    fill_exact_params(s, d, pl, nq, d_mask, mask_bits, n_tiles, p);
    p.k = (uint32_t)k_eff;
    p.perq = perq;
    p.list_stride = KS;
    p.small = small;
    if (lean) {
        p.embedded = 1;
        p.queries = nullptr;
        p.qinv = nullptr;
        p.runs = nullptr;
        p.tile_prefix = nullptr;
        memcpy(p.qemb, queries, (size_t)s->dim * 4);  // the tail up to dimq stays zero (fill_exact_params cleared the struct)
        p.eqinv = host_inv_norm_exact(queries, s->dim);
        for (size_t i = 0; i < pl.runs.size(); i++) p.eruns[i] = pl.runs[i];
        for (size_t i = 0; i < prefix.size(); i++) p.eprefix[i] = prefix[i];
    }

    if (timing) OTT_HIP(hipEventRecord(s->ev[3], s->stream));
    for (uint32_t ps = 0; ps < passes; ps++) {
        p.q0 = ps * tile;
        // merged: one list group per pass.  per-query: list (query, block) lives at (query*grid + block)*KS; a
        // 1-query pass runs the single-list kernel, so it is pointed at its query's slot (q0 == ps there)
        p.lists = (Cand*)s->d_lists.p + ((perq && tile > 1) ? 0 : (size_t)ps * grid * KS);
        if ((rc = launch_exact(s, p, (int)tile, E, grid))) return rc;
    }
    if (timing) OTT_HIP(hipEventRecord(s->ev[4], s->stream));
    if (perq)
        rc = launch_merge(s, (const Cand*)s->d_lists.p, (uint32_t)grid, KS, (uint64_t)grid * KS, nq, (uint32_t)k_eff, E,
                          p.take_max != 0, tie_base(s), d_hits, KS, d_counts, s->cur_tie_sh);
    else
        rc = launch_merge(s, (const Cand*)s->d_lists.p, (uint32_t)(passes * grid), KS, 0, 1, (uint32_t)k_eff, E, p.take_max != 0,
                          tie_base(s), d_hits, KS, d_counts, s->cur_tie_sh);
    if (rc) return rc;
    if (timing) OTT_HIP(hipEventRecord(s->ev[5], s->stream));
    st.passes += passes;
    st.bytes_scanned += (uint64_t)passes * pl.rows_scored * ((uint64_t)s->dim * 4 + (d->metric == OTT_METRIC_COSINE ? 4 : 0));
    if (!fetch) return OTT_OK;

    char* hh = (char*)s->h_hits.p;
    OTT_HIP(hipStreamSynchronize(s->stream));
    const uint64_t* counts = (const uint64_t*)hh;
    const ott_hit* hits = (const ott_hit*)(hh + cnt_pad);
    lists.assign(groups, {});
    for (uint32_t g = 0; g < groups; g++) lists[g].assign(hits + (size_t)g * KS, hits + (size_t)g * KS + counts[g]);
    float ms = 0.f;
    if (timing && hipEventElapsedTime(&ms, s->ev[3], s->ev[4]) == hipSuccess) st.score_ns += (uint64_t)(ms * 1e6);
    if (timing && hipEventElapsedTime(&ms, s->ev[4], s->ev[5]) == hipSuccess) st.merge_ns += (uint64_t)(ms * 1e6);
    return OTT_OK;
}

}  // namespace

// runs on a query context `s` (the store itself or one of its workers) whose `mu` the caller holds
int ott::query_on(ott_store* s, const ott_query_desc* d, ott_hit* out_host, void* out_dev, uint64_t cap, uint64_t* n_out,
                  uint64_t* n_per_query, void* n_out_dev, ott_stats* stats_out, bool nosync, bool* events_pending) {
    CoreOpts co;
    if (s->opt.tie_order != 0) {
        // host output: the reference's literal outcome (ott_ties.hip).  Device output (ott_query_device, the shard blocks of
        // ott_query_sharded): candidates ranked in the reference's visit order, without the collector's anchor rule
        if (out_host && !out_dev) {
            if (events_pending) *events_pending = false;
            return query_ref_ties(s, d, out_host, cap, n_out, n_per_query, stats_out);
        }
        co.tie_sh = 3;
    }
    return query_core(s, d, out_host, out_dev, cap, n_out, n_per_query, n_out_dev, stats_out, nosync, events_pending, co);
}

int ott::query_core(ott_store* s, const ott_query_desc* d, ott_hit* out_host, void* out_dev, uint64_t cap, uint64_t* n_out,
                    uint64_t* n_per_query, void* n_out_dev, ott_stats* stats_out, bool nosync, bool* events_pending, const CoreOpts& co) {
    int rc;
    if (events_pending) *events_pending = false;
    struct CurOrder {  // the candidate order of THIS call, for the launch wrappers (the context is ours until the caller releases it)
        ott_store* s;
        CurOrder(ott_store* st, const CoreOpts& c) : s(st) { s->cur_tie_sh = c.tie_sh; s->cur_tie_off = c.tie_sh ? (c.tie_off & 7u) : 0u; s->cur_flat = c.flat; }
        ~CurOrder() { s->cur_tie_sh = 0; s->cur_tie_off = 0; s->cur_flat = false; }
    } cur_order(s, co);
    OTT_HIP(use_device(s));
    const uint64_t t0 = now_ns();
    ott_stats st;
    memset(&st, 0, sizeof(st));

    RunPlan pl;
    build_plan(s, d->chunk_mask, pl);
    st.prune_ns = now_ns() - t0;
    st.total_chunks = pl.total_chunks;
    st.evaluated_chunks = pl.evaluated;
    st.pruned_chunks = pl.total_chunks - pl.evaluated;
    st.vectors_compared = pl.rows_scored * d->nq;  // sum chunk.len * nq, src/meta_compute.rs:166

    const bool perq = d->mode == OTT_MODE_PER_QUERY;
    const uint32_t nq = d->nq;
    const uint64_t pool = perq ? pl.rows_scored : pl.rows_scored * nq;
    const uint64_t k_eff = d->k < pool ? d->k : pool;  // take_count, src/vec.rs:213; a list never outgrows the pool
    const uint64_t need = perq ? k_eff * nq : k_eff;
    if (cap < need) return fail(OTT_ERR_INVALID, "ott_query: output capacity is smaller than min(k, rows*nq)");
    if (out_dev && perq && cap % nq != 0) return fail(OTT_ERR_INVALID, "ott_query_device: PER_QUERY capacity must be a multiple of nq");

    // device output: every slot the scoring path does not write must hold a sentinel.  The sentinel fill is queued lazily
    // (`fill_sentinels`): the exact path writing a block of exactly its own geometry, and the staged host lists, cover every
    // slot themselves
    auto fill_sentinels = [&]() -> int {
        OTT_HIP(hipMemsetAsync(out_dev, 0xFF, cap * sizeof(ott_hit), s->stream));
        return OTT_OK;
    };
    if (out_dev && n_out_dev) OTT_HIP(hipMemsetAsync(n_out_dev, 0, sizeof(uint64_t), s->stream));
    if (n_out) *n_out = 0;
    if (n_per_query)
        for (uint32_t i = 0; i < nq; i++) n_per_query[i] = 0;
    if (k_eff == 0 || pl.rows_scored == 0) {  // k == 0 (src/vec_compute.rs:174) or nothing to score
        if (out_dev && (rc = fill_sentinels())) return rc;
        if (out_dev && !nosync) OTT_HIP(hipStreamSynchronize(s->stream));
        st.total_ns = now_ns() - t0;
        if (stats_out) *stats_out = st;
        return OTT_OK;
    }

    // row mask -> device
    const uint64_t* d_mask = nullptr;
    uint64_t mask_bits = 0;
    if (d->use_device_row_mask) {
        d_mask = (const uint64_t*)s->d_evalmask.p;
        mask_bits = s->evalmask_bits;
    } else if (d->row_mask && d->row_mask_bits) {
        const size_t words = (size_t)((d->row_mask_bits + 63) / 64);
        if ((rc = s->d_rowmask.ensure(words * 8))) return rc;
        OTT_HIP(hipMemcpyAsync(s->d_rowmask.p, d->row_mask, words * 8, hipMemcpyHostToDevice, s->stream));
        d_mask = (const uint64_t*)s->d_rowmask.p;
        mask_bits = d->row_mask_bits;
    }

    // ---- path choice --------------------------------------------------------------------------------
    // per-query k for the batch path: the merged top-k is contained in the union of per-query top-k
    const uint64_t k_q = d->k < pl.rows_scored ? d->k : pl.rows_scored;
    const bool mfma_ok = k_q + 28 <= 512 && s->dim >= 8;
    bool use_mfma;
    if (co.flat) use_mfma = false;  // (the flat pass exists on the exact-order kernel only)
    else if (d->path == OTT_PATH_MFMA) {
        if (!mfma_ok) return fail(OTT_ERR_UNSUPPORTED, "ott_query: the MFMA path needs dim >= 8 and k <= 484");
        use_mfma = true;
    } else if (d->path == OTT_PATH_EXACT) use_mfma = false;
    else {
        // AUTO: cost model fitted to MI355X measurements (benchmarks/small_corpus.py, nq_sweep.py), in milliseconds.
        // exact: up to 4 queries share one pass; a pass of m queries costs 0.05 / 0.06 / 0.085 / 0.115 ms of launches + latency and
        //        streams at ~6.5 TB/s, 2.7 % slower per extra query (round 4, benchmarks/auto_choice.py on 300k .. 10M x 768:
        //        one query 0.187 / 0.517 / 1.40 / 4.54 ms, four 0.257 / 0.590 / 1.59 / 4.90; the 0.11 ms per pass this model
        //        carried since round 1 sent single queries on 262k-480k-row stores through the cascade, 20 % slower).
        // mfma:  ~0.16 ms of rounds / select / finalize / transfer, ~4.5 us per query of re-scoring and host merge, then the
        //        slower of the corpus stream (~6 TB/s per 256-query block) and the matrix pipe.
        const double bytes = (double)pl.rows_scored * (4.0 * s->dim + 4.0);
        const uint32_t full_passes = nq / 4, last_m = nq % 4;
        static const double pass_fixed[5] = {0.0, 0.05, 0.06, 0.085, 0.115};
        auto t_pass = [&](uint32_t m) { return pass_fixed[m] + bytes / 6.5e9 * (1.0 + 0.027 * (m - 1)); };
        const double t_exact = full_passes * t_pass(4) + (last_m ? t_pass(last_m) : 0.0);
        const uint32_t bn = nq <= 16 ? 16u : nq <= 32 ? 32u : nq <= 64 ? 64u : nq <= 128 ? 128u : 256u;
        const double nq_pad = (double)((nq + bn - 1) / bn * bn);
        const bool f32pipe = s->opt.mfma_f32;
        // (the plane's actual format once it exists — it may have fallen back to bf16 — else what the option asks for)
        const ott_store* own_c = s->owner ? s->owner : s;
        const PlaneSnapshot ps = plane_snapshot(s);  // (under img_mu: another context may be building or dropping a plane right now)
        const bool plane_half = ps.have_hi ? ps.hi_f16 : s->opt.hi_fmt != 0;
        const bool hi_ok = !f32pipe && mfma_hi_k_ok(d->k < pl.rows_scored ? d->k : pl.rows_scored, plane_half) && !s->opt.no_hi_pass;
        // round 5: the int8 plane in front (cosine / dot, k <= 128): a quarter of the bytes, twice the matrix rate, 512 candidates
        const bool i8_ok = hi_ok && i8_wanted(s->opt) && !ps.i8_off && k_q <= 128;
        // the hi pass streams the 16-bit hi plane: half the bytes
        const double t_stream = (i8_ok ? 0.25 : hi_ok ? 0.5 : 1.0) * bytes * (double)((nq + 255) / 256) / (i8_ok ? 6.0e9 : hi_ok ? (nq <= 32 ? 6.5e9 : 6.2e9) : 5.9e9);  // (non-temporal row pieces, round 2: 6.6-6.8 TB/s up to 32 queries, ~6 at 64-128)
        // matrix pipe: ~125 TFLOP/s on the f32 pipe, ~330 TFLOP/s (f32-equivalent) with the split-bf16 operands, ~800 for the hi pass, ~1500 int8
        const double t_pipe = 2.0 * s->dim * (double)pl.rows_scored * nq_pad / (i8_ok ? 1500e9 : hi_ok ? 800e9 : (bn >= 32 && !f32pipe) ? 330e9 : 125e9);
        // (the candidates re-scored per query grow with k — 2k + 56 on the hi pass, in steps of 64; 512 on the int8 pass — and finalize /
        //  select with them: top-100 costs the cascade 0.03-0.05 ms more than top-10 at one query, benchmarks/auto_choice.py)
        const double t_cand = i8_ok ? 0.0003 * 384.0 : hi_ok && k_q > 36 ? 0.0003 * (double)((2 * k_q + 56 + 63) / 64 * 64 - 128) : 0.0;
        double t_mfma = 0.16 + 0.0045 * nq + t_cand + (t_stream > t_pipe ? t_stream : t_pipe);
        // ONE query at the int8 level, k <= 24: a streaming sweep with the top-128 in its epilogue (run_i8_single): ~0.11 ms of
        // launches, merge and re-score around a quarter of the bytes at 6.5 TB/s (profiles/round5/auto_choice.md: 150k x 768 rows
        // 0.13 ms, 1M 0.24, 10M 1.29; the exact kernel 0.12 / 0.54 / 4.7)
        if (i8_ok && nq == 1 && k_q <= 24 && d->filter_cmp != OTT_CMP_EQ && d->metric != OTT_METRIC_EUCLIDEAN && s->dim <= 3584 && own_c->i8_t512.load() <= 0)
            t_mfma = 0.11 + 0.25 * bytes / 6.5e9;
        // a SINGLE query takes the exact-order kernel (no second copy of the corpus is built for the most common call) — unless
        // the bf16 hi plane is ALREADY resident (a batch query or ott_store_prepare_batch built it) and covers every row: then
        // the cascade streams half the bytes (10M x 768: 2.5 ms against 4.5) and returns the same bits;
        // 2-4 queries share one exact pass unless the hi pass (half the bytes) is cheaper; without it the batch path needs > 4
        const bool batch_worthy = nq > (hi_ok ? 1u : 4u) || (nq == 1 && hi_ok && first_plane_ready(s));
        use_mfma = mfma_ok && batch_worthy && pl.rows_scored >= 2048 && t_mfma < t_exact;
        // small stores, small batches (round 3): rows8 scores up to 8 queries per pass in ~(40 us + 0.7 us per thousand rows) behind
        // ~35 us of launches and merge, against the batch path's ~(120 us + 3.5 us per query) of rounds, select and finalize
        // (benchmarks/small_corpus.py: 10k x 768, 8 queries: 84 us against 150)
        const uint64_t k_e = d->k < pl.rows_scored * nq ? d->k : pl.rows_scored * nq;
        if (use_mfma && nq <= 16 && k_e <= 128 && s->dimq <= 2048 && s->opt.exact_small != 0 && s->opt.exact_small != 1 &&
            pl.rows_scored / 64 + pl.runs.size() <= 1024) {
            const double t_rows8 = 0.035 + (double)((nq + 7) / 8) * (0.040 + 0.7e-6 * (double)pl.rows_scored);
            if (t_rows8 < 0.9 * (0.12 + 0.0035 * nq)) use_mfma = false;
        }
    }

    std::vector<std::vector<ott_hit>> lists;  // groups: 1 (merged) or nq
    if (!use_mfma) {
        st.path_used = OTT_PATH_EXACT;
        // kernel timing (three event records, each a barrier packet between the launches) only when the caller asked for stats
        const bool timing = stats_out != nullptr;
        // device output of k <= 512: the merge kernel leaves [groups][KS] sentinel-padded hits in d_hits, copied to the caller's
        // block on the stream — nothing comes back to the host
        const bool dev_direct = out_dev != nullptr && k_eff <= 512;
        const uint32_t groups_x = perq ? nq : 1u;
        const uint64_t KSx = 64ull * (uint64_t)list_E(k_eff), gstride_x = out_dev ? cap / groups_x : 0;
        // when the caller's block has exactly the merge kernel's geometry ([groups][KS]: what ott_query_sharded asks for), the
        // merge writes into it directly: no sentinel fill in front, no copy behind
        const bool in_place = dev_direct && gstride_x == KSx;
        // host output in the canonical order: the sort path may write a large result straight into the caller's buffer
        struct DirectGuard {
            ott_store* c;
            ~DirectGuard() {
                c->direct_out = nullptr;
                c->direct_cap = 0;
                c->direct_done = false;
            }
        } direct_guard{s};
        if (out_host && !out_dev && s->cur_tie_sh == 0 && !s->cur_flat) {
            s->direct_out = out_host;
            s->direct_cap = cap;
        }
        s->direct_done = false;
        rc = run_exact(s, d->queries, nq, d, perq, pl, k_eff, d_mask, mask_bits, !dev_direct, lists, st, timing, in_place ? (ott_hit*)out_dev : nullptr,
                       (uint32_t)KSx);
        if (rc) return rc;
        if (s->direct_done) {
            uint64_t total = 0;
            for (size_t gq = 0; gq < s->direct_counts.size(); gq++) {
                if (n_per_query && perq) n_per_query[gq] = s->direct_counts[gq];
                total += s->direct_counts[gq];
            }
            if (n_out) *n_out = total;
            st.total_ns = now_ns() - t0;
            if (stats_out) *stats_out = st;
            return OTT_OK;
        }
        if (dev_direct) {
            const uint32_t groups = groups_x;
            const uint64_t KS = KSx, gstride = gstride_x;
            if (!in_place) {
                if ((rc = fill_sentinels())) return rc;
                const uint64_t width = (KS < gstride ? KS : gstride) * sizeof(ott_hit);  // k_eff <= gstride: no hit is cut
                const char* src = (const char*)s->d_hits.p + s->res_hits_off;
                if (groups == 1) OTT_HIP(hipMemcpyAsync(out_dev, src, width, hipMemcpyDeviceToDevice, s->stream));
                else OTT_HIP(hipMemcpy2DAsync(out_dev, gstride * sizeof(ott_hit), src, KS * sizeof(ott_hit), width, groups, hipMemcpyDeviceToDevice, s->stream));
            }
            if (n_out_dev) {
                hipLaunchKernelGGL(sum_counts_kernel, dim3(1), dim3(64), 0, s->stream, (const uint64_t*)s->d_hits.p, groups, (uint64_t*)n_out_dev);
                OTT_HIP(hipGetLastError());
            }
            if (nosync) {
                if (events_pending) *events_pending = timing;
            } else {
                OTT_HIP(hipStreamSynchronize(s->stream));  // the caller's collective runs on another stream
                if (timing) read_exact_events(s, &st);
            }
            st.total_ns = now_ns() - t0;
            if (stats_out) *stats_out = st;
            return OTT_OK;
        }
    } else {
        std::vector<std::vector<ott_hit>> pq;
        std::vector<uint32_t> unc;
        // Cascade of candidate passes, each certified against the exact re-score: hi pass (bf16 hi plane: half the bytes, a
        // third of the MFMAs, bound ~2^-8) -> split pass (bound ~2^-16) for the queries it could not certify -> exact path.
        ott_store* own = s->owner ? s->owner : s;
        bool hi_pass = mfma_hi_k_ok(k_q, s->opt.hi_fmt != 0) && !s->opt.mfma_f32 && !s->opt.no_hi_pass;
        // the hi plane is looked at (built, extended) only when a level is about to stream it: with the int8 level in front most
        // stores never need it
        bool hi_checked = false;
        bool hi_backing_off = false;  // the hi pass would run but is sitting out: the corpus is dense, the split pass keeps its 512 candidates
        auto check_hi = [&]() -> int {
            if (hi_checked || !hi_pass) return OTT_OK;
            hi_checked = true;
            const uint16_t* himg = nullptr;
            float hrel = 0.f;
            bool is_half = false;
            const int rc2 = ensure_hi_plane(s, &himg, &hrel, &is_half);
            if (rc2) return rc2;
            // the format the plane ACTUALLY has decides (a store whose norms spread over many binades falls back to bf16 by itself:
            // k in 229..363 would then re-score fewer candidates than the bf16 bound needs and every batch would pay a hi pass
            // that certifies nothing)
            hi_pass = himg != nullptr && mfma_hi_k_ok(k_q, is_half);
            if (hi_pass && own->hi_skip.load() > 0) {  // backing off: recent batches mostly needed the split pass anyway
                own->hi_skip.fetch_sub(1);
                hi_pass = false;
                hi_backing_off = true;
            }
            return OTT_OK;
        };
        // Round 5 (default; option hi_fmt = -1 / 2): an INT8 level in front of the hi pass (cosine / dot, k <= 128): a quarter of the
        // f32 bytes, one v_mfma_i32_32x32x32_i8 per 32 k, exact integer accumulation — its bound is the measured quantisation loss
        // alone (~8e-3 relative on uniform 768-d rows), so it re-scores 512 candidates per query and certifies where fewer than
        // 512 - k rows lie that close to the k-th score; what it leaves open goes to the hi pass.  Same back-off as the hi pass.
        bool i8_pass = hi_pass && i8_wanted(s->opt) && k_q <= 128;
        if (i8_pass) {
            const int8_t* i8 = nullptr;
            const float* i8s = nullptr;
            float i8rel = 0.f;
            if ((rc = ensure_i8_plane(s, &i8, &i8s, &i8rel))) return rc;
            i8_pass = i8 != nullptr;
        }
        if (i8_pass && own->i8_skip.load() > 0) {
            own->i8_skip.fetch_sub(1);
            i8_pass = false;
        }
        if (!i8_pass && (rc = check_hi())) return rc;
        // the split pass is then a later level: it re-scores 512 candidates per query — also while the hi pass backs off (it backs off
        // on dense or clustered corpora, exactly where k + 28 candidates certify nothing)
        const bool cascade = hi_pass || i8_pass || hi_backing_off;
        // the 4096-candidate level is there for every bf16 batch (also k > 228 or no hi plane: split pass, wide split pass, exact)
        const bool escalate = !s->opt.mfma_f32;
        bool spec_now = s->opt.mfma_spec != 0;
        if (spec_now && own->spec_skip.load() > 0) {  // backing off: a speculative gate failed a query on this store recently
            own->spec_skip.fetch_sub(1);
            spec_now = false;
        }
        // one level of the cascade over the queries `which` (indices into the batch; empty = all of it)
        auto run_level = [&](const std::vector<uint32_t>& which, int level, uint32_t t_min, bool first) -> int {
            std::vector<float> sub;
            ott_query_desc d2 = *d;
            if (!which.empty()) {
                sub.resize((size_t)which.size() * s->dim);
                for (size_t i = 0; i < which.size(); i++) memcpy(&sub[i * s->dim], d->queries + (size_t)which[i] * s->dim, (size_t)s->dim * 4);
                d2.queries = sub.data();
                d2.nq = (uint32_t)which.size();
            }
            std::vector<std::vector<ott_hit>> pq2;
            std::vector<uint32_t> unc2;
            ott_stats st2 = st;
            // speculative emission thresholds: only where a query meets the cascade FIRST (a level that re-runs the queries
            // another level could not certify uses conservative gates, whatever the reason they failed), and not while
            // backing off after a gate failed on this store
            const bool spec = first && spec_now;
            // (round 5: ONE query at the int8 level is a streaming sweep with the top-T in its epilogue — three launches instead of
            //  the cascade's five rounds)
            //  — while the list it keeps is 128 entries (k <= 24: the wave lists of 256 / 512 entries cost the sweep more than the
            //  rounds cost the cascade: top-100 at 10M x 768 1.97 ms against 1.46)
            const bool single_sweep = level == 2 && d2.nq == 1 && d2.filter_cmp != OTT_CMP_EQ && d2.metric != OTT_METRIC_EUCLIDEAN && s->dim <= 3584 && k_q <= 24 && t_min <= 128;  // (the sweep kernel scores cosine / dot)
            int rc2 = single_sweep ? run_i8_single(s, &d2, pl, k_q, d_mask, mask_bits, pq2, unc2, st2, t_min)
                                   : run_mfma(s, &d2, pl, k_q, d_mask, mask_bits, pq2, unc2, st2, level, t_min, spec);
            if (rc2) return rc2;
            st.gate_failed = st2.gate_failed;  // (st2 started as a copy of st: accumulated)
            st.bound_violations = st2.bound_violations;
            if (first) {
                st.score_ns = st2.score_ns; st.merge_ns = st2.merge_ns; st.rescored = st2.rescored; st.passes = st2.passes;
                st.bytes_scanned = st2.bytes_scanned; st.path_used = st2.path_used;
            } else {
                st.score_ns += st2.score_ns; st.merge_ns += st2.merge_ns; st.rescored += st2.rescored; st.passes += st2.passes;
                st.bytes_scanned += st2.bytes_scanned;
            }
            if (st2.err_ratio_max > st.err_ratio_max) st.err_ratio_max = st2.err_ratio_max;
            if (which.empty()) {
                pq = std::move(pq2);
                unc = std::move(unc2);
            } else {
                for (size_t i = 0; i < which.size(); i++) {
                    for (auto& h : pq2[i]) h.query = which[i];
                    pq[which[i]] = std::move(pq2[i]);
                    unc[which[i]] = unc2[i];
                }
            }
            return OTT_OK;
        };
        auto open_queries = [&]() {
            std::vector<uint32_t> v;
            for (uint32_t q = 0; q < nq; q++)
                if (unc[q]) v.push_back(q);
            return v;
        };
        const std::vector<uint32_t> all;
        std::vector<uint32_t> after_i8;  // the queries the int8 level left open (it ran and certified the rest)
        bool i8_ran = false;
        if (i8_pass) {
            // (i8_t512 = calls left that re-score 512 candidates per query: 64 after a failure at less, counted down by the calls
            //  that follow — one query in a dense neighbourhood does not widen the store's every later call for good)
            const bool i8_wide_now = own->i8_t512.load() > 0;
            if (i8_wide_now) own->i8_t512.fetch_sub(1);
            if ((rc = run_level(all, 2, i8_wide_now ? 512u : 0u, true))) return rc;
            i8_ran = true;
            after_i8 = open_queries();
            st.i8_refined = (uint32_t)after_i8.size();
            const size_t genuine8 = after_i8.size() > st.gate_failed ? after_i8.size() - st.gate_failed : 0;
            // A batch in which ANY query stays open pays a second pass — the hi pass over the half plane, ~2.4 ms at 10M x 768
            // however few the queries — so the int8 level only pays while most batches certify whole: measured on near-duplicate
            // clusters (7-17 of 256 queries open in EVERY batch) int8 first took 5.65 ms per batch where the hi pass alone takes
            // 4.8.  Batches of more than 512 queries are the exception: their int8 pass saves more than the second pass costs
            // (1024 queries: 9.4 + 2.4 ms against 15.4).  Back-off as for the hi pass: more than 1/8 of a batch open, or more than
            // ~half (small batches: ~40 %) of the recent batches needing the second pass at all.
            const int ema8 = (3 * own->i8_fail_ema.load() + (genuine8 == 0 ? 0 : 1024)) / 4;
            own->i8_fail_ema.store(ema8);
            if (genuine8 * 8 > nq && !i8_wide_now && 4 * k_q + 88 < 512) {
                own->i8_t512.store(64);  // first answer to dense neighbourhoods: re-score 512 per query for the next 64 calls
                own->i8_fail_ema.store(0);
            } else if (genuine8 * 8 > nq || (nq <= 512 && ema8 > (nq <= 128 ? 400 : 512))) {
                int b = own->i8_backoff.load() * 2;
                b = b < 4 ? 4 : b > 64 ? 64 : b;
                own->i8_backoff.store(b);
                own->i8_skip.store(b);
            } else if (genuine8 == 0) own->i8_backoff.store(0);
        }
        if (i8_ran && !after_i8.empty() && (rc = check_hi())) return rc;  // what the int8 level left open needs the hi plane now
        if (i8_ran && after_i8.empty()) {
            // every query certified by the int8 level: nothing left for the others
        } else if (hi_pass) {
            // candidates re-scored per query by the hi pass: 2k + 56, or 512 once this store's queries have failed at that
            // (dense neighbourhoods: clustered corpora), or what the hi_tmin option says
            const bool hi_wide_now = own->hi_t512.load() != 0;
            const uint32_t hi_t = s->opt.hi_tmin ? (uint32_t)s->opt.hi_tmin : (hi_wide_now ? 512u : 0u);
            if ((rc = run_level(i8_ran ? after_i8 : all, 0, hi_t, !i8_ran))) return rc;
            std::vector<uint32_t> refine = open_queries();
            st.refined = (uint32_t)refine.size();
            // queries that failed only through their speculative gate say nothing about the hi pass's error bound
            const size_t genuine = refine.size() > st.gate_failed ? refine.size() - st.gate_failed : 0;
            if (genuine * 8 > nq && !hi_wide_now && hi_t < 512u) {
                // first answer to a store whose queries sit in dense neighbourhoods: keep the hi pass, re-score 512 per query
                // from the next batch on (this batch's open queries go to the split pass below); only if THAT keeps failing does
                // the store back off from the hi pass
                own->hi_t512.store(1);
                own->hi_fail_ema.store(0);
            } else {
                const int ema = (3 * own->hi_fail_ema.load() + (genuine == 0 ? 0 : 1024)) / 4;
                own->hi_fail_ema.store(ema);
                if (genuine * 8 > nq || ema > 512) {
                    int b = own->hi_backoff.load() * 2;
                    b = b < 4 ? 4 : b > 64 ? 64 : b;
                    own->hi_backoff.store(b);
                    own->hi_skip.store(b);
                } else if (genuine == 0) own->hi_backoff.store(0);
            }
            if (!refine.empty() && (rc = run_level(refine, 1, 512, false))) return rc;
        } else if (escalate && nq > 8 && own->wide_first.load() > 0) {
            // the 512-candidate level has been failing on this store: start at the 4096-candidate one for a while
            own->wide_first.fetch_sub(1);
            if ((rc = run_level(i8_ran ? after_i8 : all, 1, 4096, !i8_ran))) return rc;
        } else {
            if ((rc = run_level(i8_ran ? after_i8 : all, 1, cascade ? 512u : 0u, !i8_ran))) return rc;
        }
        if (spec_now) {  // gate back-off: conservative for 8, 16, .. 256 batches after a failure, forgotten after a clean batch
            if (st.gate_failed) {
                int b = own->spec_backoff.load() * 2;
                b = b < 8 ? 8 : b > 256 ? 256 : b;
                own->spec_backoff.store(b);
                own->spec_skip.store(b);
            } else own->spec_backoff.store(0);
        }
        if (escalate) {
            // third level: still more than a couple of exact passes' worth of open queries (near-duplicate clusters: hundreds of
            // rows within the split pass's bound of the k-th score) — the split pass once more, re-scoring 4096 per query
            std::vector<uint32_t> wide = open_queries();
            if (wide.size() > 8 && st.rescored < (uint64_t)nq * 4096) {
                if (wide.size() * 2 > nq) own->wide_first.store(16);  // most of the batch: skip the 512 level next time
                if ((rc = run_level(wide, 1, 4096, false))) return rc;
            }
        }
        // uncertified queries: recompute on the exact path (per-query lists)
        std::vector<uint32_t> redo;
        for (uint32_t q = 0; q < nq; q++)
            if (unc[q]) redo.push_back(q);
        st.retries = (uint32_t)redo.size();
        if (!redo.empty()) {
            std::vector<float> sub((size_t)redo.size() * s->dim);
            for (size_t i = 0; i < redo.size(); i++) memcpy(&sub[i * s->dim], d->queries + (size_t)redo[i] * s->dim, (size_t)s->dim * 4);
            std::vector<std::vector<ott_hit>> fix;
            rc = run_exact(s, sub.data(), (uint32_t)redo.size(), d, true, pl, k_q, d_mask, mask_bits, true, fix, st);
            if (rc) return rc;
            for (size_t i = 0; i < redo.size(); i++) {
                for (auto& h : fix[i]) h.query = redo[i];
                pq[redo[i]] = std::move(fix[i]);
            }
        }
        if (perq) lists = std::move(pq);
        else {
            // reference semantics: one list over all (query, row) pairs (src/vec.rs:217-219).  The per-query lists are sorted
            // best first, so the merged top-k is a k-way merge over their heads: k pops of a heap of nq cursors (a
            // partial_sort over all nq x k hits was ~0.1 ms of a 256-query batch)
            const CanonLess less{d->take == OTT_TAKE_MAX, co.tie_sh, tie_base(s)};
            std::vector<std::pair<uint32_t, uint32_t>> heap;  // (list, position); the heap's top is the best head
            auto worse = [&](const std::pair<uint32_t, uint32_t>& a, const std::pair<uint32_t, uint32_t>& b) {
                return less(pq[b.first][b.second], pq[a.first][a.second]);
            };
            for (uint32_t q = 0; q < nq; q++)
                if (!pq[q].empty()) heap.emplace_back(q, 0u);
            std::make_heap(heap.begin(), heap.end(), worse);
            std::vector<ott_hit> all;
            all.reserve((size_t)k_eff < (size_t)nq * k_q ? (size_t)k_eff : (size_t)nq * k_q);
            while (!heap.empty() && all.size() < k_eff) {
                std::pop_heap(heap.begin(), heap.end(), worse);
                const std::pair<uint32_t, uint32_t> cur = heap.back();
                heap.pop_back();
                all.push_back(pq[cur.first][cur.second]);
                if (cur.second + 1 < pq[cur.first].size()) {
                    heap.emplace_back(cur.first, cur.second + 1);
                    std::push_heap(heap.begin(), heap.end(), worse);
                }
            }
            lists.assign(1, std::move(all));
        }
    }

    if (out_dev) {
        // device output of host-side lists (batch path, k > 512): [groups][cap / groups] slots, each list best first, the rest
        // sentinels (already memset).  ONE copy of the whole block from pinned staging (a copy per query was ~5 us of enqueue
        // each: 5 ms for the 1024 queries of a C4 shard)
        const uint32_t groups = perq ? nq : 1u;
        const uint64_t gstride = cap / groups;
        const size_t blk = (size_t)groups * gstride * sizeof(ott_hit);
        if ((rc = s->h_stage.ensure(blk + sizeof(uint64_t)))) return rc;
        char* hs = (char*)s->h_stage.p;
        memset(hs, 0xFF, blk);  // sentinels, as the memset of out_dev left them
        uint64_t tot = 0;
        for (uint32_t q = 0; q < groups; q++) {
            const size_t c = lists[q].size() < gstride ? lists[q].size() : (size_t)gstride;
            if (c) memcpy(hs + (size_t)q * gstride * sizeof(ott_hit), lists[q].data(), c * sizeof(ott_hit));
            tot += c;
        }
        memcpy(hs + blk, &tot, sizeof(uint64_t));
        OTT_HIP(hipMemcpyAsync(out_dev, hs, blk, hipMemcpyHostToDevice, s->stream));
        if (n_out_dev) OTT_HIP(hipMemcpyAsync(n_out_dev, hs + blk, sizeof(uint64_t), hipMemcpyHostToDevice, s->stream));
        if (!nosync) OTT_HIP(hipStreamSynchronize(s->stream));  // (nosync: h_stage stays untouched until the caller's own wait)
        st.total_ns = now_ns() - t0;
        if (stats_out) *stats_out = st;
        return OTT_OK;
    }
    uint64_t total = 0;
    for (size_t gq = 0; gq < lists.size(); gq++) {
        const size_t c = lists[gq].size();
        if (c) memcpy(out_host + total, lists[gq].data(), c * sizeof(ott_hit));
        if (n_per_query && perq) n_per_query[gq] = c;
        total += c;
    }
    if (n_out) *n_out = total;
    st.total_ns = now_ns() - t0;
    if (stats_out) *stats_out = st;
    return OTT_OK;
}

namespace {

int query_common(ott_store* s, const ott_query_desc* d, ott_hit* out_host, void* out_dev, uint64_t cap, uint64_t* n_out,
                 uint64_t* n_per_query, void* n_out_dev, ott_stats* stats_out) {
    int rc = validate_query(s, d);
    if (rc) return rc;
    // rows of small appends still staged on the host go to the GPU first (that takes the store exclusively); an append may
    // slip in before the shared lock is held, so the staged count is looked at again under it
    ott::host::SharedLock rd;  // the corpus cannot change while this query runs
    if ((rc = ott::host::lock_shared_clean(s->rw, rd, [s] { return s->pend.count() != 0; }, [s] { return store_flush(s); }))) return rc;
    ott_store* ctx = ott::ctx_acquire(s);
    rc = query_on(ctx, d, out_host, out_dev, cap, n_out, n_per_query, n_out_dev, stats_out);
    ott::ctx_release(ctx);
    return rc;
}

}  // namespace

extern "C" {

int ott_query(ott_store* s, const ott_query_desc* d, ott_hit* out, uint64_t cap, uint64_t* n_out, uint64_t* n_per_query,
              ott_stats* stats) {
    if (!out && cap) return fail(OTT_ERR_INVALID, "ott_query: out is NULL");
    if (s && s->multi) return multi_query(s, d, out, cap, n_out, n_per_query, stats);
    return query_common(s, d, out, nullptr, cap, n_out, n_per_query, nullptr, stats);
}

int ott_query_device(ott_store* s, const ott_query_desc* d, void* out_dev, uint64_t cap, void* n_out_dev, ott_stats* stats) {
    if (!out_dev) return fail(OTT_ERR_INVALID, "ott_query_device: out_dev is NULL");
    if (s && s->multi) return fail(OTT_ERR_UNSUPPORTED, "ott_query_device: a multi-GPU store returns its result to the host (ott_query)");
    return query_common(s, d, nullptr, out_dev, cap, nullptr, nullptr, n_out_dev, stats);
}

static int merge_hits_common(ott_store* s, const void* lists_dev, uint64_t n_lists, uint64_t n_groups, uint64_t list_len, uint32_t take,
                             uint64_t k, ott_hit* out_host, uint64_t* n_out, uint64_t* n_per_group) {
    if (!s || !lists_dev || !out_host) return fail(OTT_ERR_INVALID, "ott_merge_hits_device: NULL argument");
    if (s->multi) return fail(OTT_ERR_UNSUPPORTED, "ott_merge_hits_device: not on a multi-GPU store (its own merge runs inside ott_query)");
    if (take > OTT_TAKE_MAX) return fail(OTT_ERR_INVALID, "ott_merge_hits_device: unknown take type");
    if (n_lists * list_len > 0xFFFFFFF0ull || n_groups > 0xFFFFull * 16) return fail(OTT_ERR_INVALID, "ott_merge_hits_device: too many candidates");
    ott::host::SharedLock rd(s->rw);
    struct Ctx {  // query context for the duration of the call
        ott_store* c;
        explicit Ctx(ott_store* owner) : c(ott::ctx_acquire(owner)) {}
        ~Ctx() { ott::ctx_release(c); }
    } ctx(s);
    s = ctx.c;
    OTT_HIP(use_device(s));
    const uint64_t pool = n_lists * list_len;
    const uint64_t k_eff = k < pool ? k : pool;
    if (n_out) *n_out = 0;
    if (n_per_group)
        for (uint64_t i = 0; i < n_groups; i++) n_per_group[i] = 0;
    if (k_eff == 0 || n_groups == 0) return OTT_OK;
    if (k_eff > 512) {
        // beyond the register lists: the candidates come to the host and are merged there in the same order the kernel uses
        // (better score, then lower list, then lower position: shard order = global row order)
        std::vector<ott_hit> all((size_t)n_lists * n_groups * list_len);
        OTT_HIP(hipMemcpyAsync(all.data(), lists_dev, all.size() * sizeof(ott_hit), hipMemcpyDeviceToHost, s->stream));
        OTT_HIP(hipStreamSynchronize(s->stream));
        const bool tmax = take == OTT_TAKE_MAX;
        uint64_t total = 0;
        std::vector<std::pair<uint64_t, uint64_t>> keys;  // (ord << 32 | ~id ... as two words: ord, id)
        for (uint64_t gq = 0; gq < n_groups; gq++) {
            keys.clear();
            for (uint64_t li = 0; li < n_lists; li++)
                for (uint64_t pos = 0; pos < list_len; pos++) {
                    const ott_hit& h = all[(li * n_groups + gq) * list_len + pos];
                    if (h.index == ~0ull || h.score != h.score) continue;
                    keys.emplace_back((uint64_t)ord_of(h.score, tmax), li * list_len + pos);
                }
            const size_t keep = keys.size() < k_eff ? keys.size() : (size_t)k_eff;
            std::partial_sort(keys.begin(), keys.begin() + keep, keys.end(),
                              [](const std::pair<uint64_t, uint64_t>& a, const std::pair<uint64_t, uint64_t>& b) {
                                  return a.first != b.first ? a.first > b.first : a.second < b.second;
                              });
            for (size_t i = 0; i < keep; i++) {
                const uint64_t li = keys[i].second / list_len, pos = keys[i].second % list_len;
                out_host[total + i] = all[(li * n_groups + gq) * list_len + pos];
            }
            if (n_per_group) n_per_group[gq] = keep;
            total += keep;
        }
        if (n_out) *n_out = total;
        return OTT_OK;
    }
    const int E = list_E(k_eff);
    const uint32_t KS = 64 * E;
    int rc;
    const size_t hits_bytes = (size_t)n_groups * KS * sizeof(ott_hit), cnt_bytes = (size_t)n_groups * 8;
    // the kernel writes counts + hits straight into pinned host memory (no D2H copies behind the launch)
    if ((rc = s->h_hits.ensure(hits_bytes + cnt_bytes))) return rc;
    char* hh = (char*)s->h_hits.p;
    void* mapped = nullptr;
    OTT_HIP(hipHostGetDevicePointer(&mapped, hh, 0));
    rc = launch_merge_hits(s, (const ott_hit*)lists_dev, (uint32_t)n_lists, (uint32_t)n_groups, (uint32_t)list_len, (uint32_t)k_eff, E,
                           take == OTT_TAKE_MAX, (ott_hit*)((char*)mapped + cnt_bytes), (uint64_t*)mapped);
    if (rc) return rc;
    OTT_HIP(hipStreamSynchronize(s->stream));
    const uint64_t* cnt = (const uint64_t*)hh;
    const ott_hit* hits = (const ott_hit*)(hh + cnt_bytes);
    uint64_t total = 0;
    for (uint64_t gq = 0; gq < n_groups; gq++) {
        if (cnt[gq]) memcpy(out_host + total, hits + gq * KS, cnt[gq] * sizeof(ott_hit));
        if (n_per_group) n_per_group[gq] = cnt[gq];
        total += cnt[gq];
    }
    if (n_out) *n_out = total;
    return OTT_OK;
}

int ott_merge_hits_device(ott_store* s, const void* lists_dev, uint64_t n_lists, uint64_t list_len, uint32_t take, uint64_t k,
                          ott_hit* out_host, uint64_t* n_out) {
    return merge_hits_common(s, lists_dev, n_lists, 1, list_len, take, k, out_host, n_out, nullptr);
}

int ott_merge_hits_device_grouped(ott_store* s, const void* lists_dev, uint64_t n_lists, uint64_t n_groups, uint64_t list_len,
                                  uint32_t take, uint64_t k, ott_hit* out_host, uint64_t* n_out, uint64_t* n_per_group) {
    return merge_hits_common(s, lists_dev, n_lists, n_groups, list_len, take, k, out_host, n_out, n_per_group);
}

}  // extern "C"

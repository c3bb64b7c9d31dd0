// ott_ties.hip — store option tie_order: the reference's own outcome at exact score ties.
//
// By default the library ranks candidates in a canonical total order (better score, lower row, lower query).  The reference
// has no such order: its TopKCollector (src/vec_compute.rs:236-268) inserts a pair only if its score is STRICTLY better
// than the current k-th, at the position `binary_search_by` returns, and pops the last entry — so which of several
// equal-scoring (row, query) pairs survives the cut at take(k) depends on the order the scoring loop visits them in
// (src/vec.rs:222-303: blocks of eight rows, every query per block, the eight lanes in order; then the remainder rows, query
// by query) and on the collector's mechanics.  This file reproduces that outcome from GPU results, without replaying the
// stream:
//
//  * every kernel and host merge of the call ranks candidates by (score, row >> 3, query, row & 7) — the visit order
//    among equal scores (`tie_sh = 3`, see `before()` in ott_exact.hip) — and one more candidate than asked for is taken;
//  * if the (k+1)-th candidate scores worse than the k-th, nothing is ambiguous: the first k ARE the reference's set;
//  * otherwise the cut runs through a group G of equal scores, and the collector's rules decide (derivation below):
//    which members of G were ever inserted, and which one is the group's ANCHOR — the entry that sits last in the group's
//    run of the buffer and is therefore popped first.  Both follow from (a) the visit order of the candidates, which the
//    key carries, and (b) which pairs made the collector's FILL phase (the first k passing pairs, sorted once when the
//    buffer fills, src/vec_compute.rs:257-266) — obtained, only when needed, by one more pass of the exact kernel with
//    every passing score ranked the same (`flat`), whose top-k by visit order is exactly that set.
//
// Derivation (S = the k-th score, B = pairs strictly better than S, all of them in the result, c = k - |B| slots left for G;
// g_1, g_2, .. = G in visit order):
//   - a pair is inserted iff fewer than k earlier-visited pairs score at least as well; so G's inserted members are a
//     prefix g_1..g_j of the visit order, and g_{c+1} is inserted iff some member of B is visited after it;
//   - find_insert_position (src/vec_compute.rs:270-277; std's binary_search_by since 1.82: `base = if cmp == Greater { base }
//     else { mid }`) lands on the LAST entry of a run of equal scores, and Vec::insert puts the new pair in front of it: a run
//     keeps its last entry last.  That entry — the anchor — is the last member of the run that was present when the buffer
//     was first sorted (the sort of the oracle's literal restatement is stable: visit order inside a run), or, if none was,
//     the first member inserted afterwards.  A run therefore reads [members in visit order without the anchor.., anchor];
//   - pops take the buffer's last entry, i.e. the cut group loses its anchor first, then its latest members: the survivors
//     are the first c entries of that run order.
// (The reference's initial sort is `sort_unstable_by`: for k <= 20 that is an insertion sort and the statement above is exact;
// beyond, the order std's ipnsort leaves inside a run of equal scores is an implementation detail this file — like the
// oracle's OTTO_TIES_LITERAL, which it is tested against — replaces by the stable one.)
//
// tie_order = 1: ONE collector over the store (VecStore).  tie_order = 2: one collector per chunk, the per-chunk lists
// concatenated in chunk order, stably sorted by score and truncated (MetaStore, src/meta.rs:678-709 / process_chunk), for any
// chunk size (src/meta.rs:86-89).  When the cut is ambiguous, the chunks that hold candidates are re-queried one by one as
// stores of their own (tie_order = 1 on a one-chunk mask): a chunk's 8-row blocks are counted from the CHUNK's first row —
// the environment's run_chunk passes the offset (CoreOpts::tie_off) down to the kernels' keys and the closed form below runs
// with the chunk's first row as its base — so chunks of 1000 or 1021 rows reproduce their collectors' visit order too.
#include <string.h>

#include <algorithm>
#include <set>
#include <utility>

#include "ott_internal.h"

namespace ott {
namespace {

typedef std::set<std::pair<uint64_t, uint32_t>> PairSet;  // (global row, query)

// Everything below is written against a TieEnv (ott_internal.h): HOW a candidate list is obtained — one store (query_core), the
// shards of an in-process multi-GPU store (ott_multi.hip), the ranks of a sharded job (ott_comm.hip) — is the environment's
// business; WHICH of the equal-scoring candidates the reference keeps is decided here, once.
typedef TieEnv Ctx;
typedef TieEnv::Runner Runner;

inline uint32_t ord(const Ctx& c, const ott_hit& h) { return ord_of(h.score, c.tmax); }

// visit order of two pairs of equal score
inline bool visited_before(const Ctx& c, const ott_hit& a, const ott_hit& b) {
    const uint64_t ra = a.index - c.base, rb = b.index - c.base;
    if ((ra >> 3) != (rb >> 3)) return (ra >> 3) < (rb >> 3);
    if (a.query != b.query) return a.query < b.query;
    return ra < rb;
}

uint64_t k_plus_one(uint64_t k) { return k == ~0ull ? k : k + 1; }

// one plain query into host vectors through the environment's runner.  per: per-query counts (PER_QUERY), lists concatenated
// in query order
int run_core(const Runner& run, const ott_query_desc& d, uint64_t k, bool flat, std::vector<ott_hit>& out, std::vector<uint64_t>& per, ott_stats* st) {
    out.clear();
    per.assign(d.nq, 0);
    return run(d, k, flat, out, per, st);
}

// The collector's result for ONE candidate list.  L: up to k + 1 candidates, best first, equal scores in visit order.
// get_F: fills the fill-phase set on demand (returns a status).  want_order: arrange every run of equal scores the way the
// collector's buffer holds it (needed when the list is merged further by position: tie_order = 2); otherwise runs stay in
// visit order and the fill-phase pass is only made when the cut is ambiguous.
template <typename GetF>
int collector_result(const Ctx& c, const std::vector<ott_hit>& L, uint64_t k, GetF&& get_F, bool want_order, std::vector<ott_hit>& out) {
    out.clear();
    const size_t m = L.size();
    if (m == 0 || k == 0) return OTT_OK;
    const bool over = m > k;  // a (k+1)-th candidate exists
    const size_t kk = over ? (size_t)k : m;
    const bool ambiguous = over && ord(c, L[kk]) == ord(c, L[kk - 1]);
    PairSet F;
    bool have_F = false;
    auto need_F = [&]() -> int {
        if (have_F) return OTT_OK;
        have_F = true;
        return get_F(F);
    };
    auto in_F = [&](const ott_hit& h) { return F.count(std::make_pair(h.index, h.query)) != 0; };

    // the cut group
    size_t g0 = kk;  // first index of the k-th score's run inside L[0, kk)
    while (g0 > 0 && ord(c, L[g0 - 1]) == ord(c, L[kk - 1])) g0--;
    std::vector<ott_hit> cut;  // the cut run's survivors, in buffer order
    if (ambiguous) {
        const size_t cslots = kk - g0;  // c: slots left for the group
        const ott_hit& next = L[kk];    // g_{c+1}
        bool inserted = false;          // some strictly better pair is visited after g_{c+1}
        for (size_t i = 0; i < g0 && !inserted; i++) inserted = visited_before(c, next, L[i]);
        if (!inserted) {
            // g_{c+1} never entered: the group's inserted members are exactly g_1..g_c and all of them stay
            cut.assign(L.begin() + g0, L.begin() + kk);
            if (want_order) {
                int rc = need_F();
                if (rc) return rc;
                size_t mF = 0;
                while (mF < cut.size() && in_F(cut[mF])) mF++;
                const size_t anchor = mF ? mF - 1 : 0;
                std::rotate(cut.begin() + anchor, cut.begin() + anchor + 1, cut.end());  // the anchor goes last
            }
        } else {
            int rc = need_F();
            if (rc) return rc;
            // members known: g_1..g_{c+1} = L[g0 .. kk]; those of the fill phase are a prefix g_1..g_mF
            size_t mF = 0;
            while (mF <= cslots && in_F(L[g0 + mF])) mF++;
            if (mF == cslots + 1) {
                cut.assign(L.begin() + g0, L.begin() + kk);  // the anchor lies beyond g_{c+1}: the first c in visit order stay
            } else {
                const size_t anchor = mF ? mF - 1 : 0;       // g_mF, or g_1 when the fill phase held none of them
                for (size_t i = 0; i <= cslots; i++)
                    if (i != anchor) cut.push_back(L[g0 + i]);  // first c of [members without the anchor.., anchor]
            }
        }
    } else {
        cut.assign(L.begin() + g0, L.begin() + kk);
        if (want_order && cut.size() > 1) {
            int rc = need_F();
            if (rc) return rc;
            size_t last_in = cut.size();
            for (size_t i = 0; i < cut.size(); i++)
                if (in_F(cut[i])) last_in = i;
            const size_t anchor = last_in < cut.size() ? last_in : 0;
            std::rotate(cut.begin() + anchor, cut.begin() + anchor + 1, cut.end());
        }
    }
    // the runs in front of the cut group: every member was inserted and none was popped
    out.assign(L.begin(), L.begin() + g0);
    if (want_order) {
        size_t i = 0;
        while (i < g0) {
            size_t j = i + 1;
            while (j < g0 && ord(c, out[j]) == ord(c, out[i])) j++;
            if (j - i > 1) {
                int rc = need_F();
                if (rc) return rc;
                size_t last_in = j;
                for (size_t t = i; t < j; t++)
                    if (in_F(out[t])) last_in = t;
                const size_t anchor = last_in < j ? last_in : i;
                std::rotate(out.begin() + anchor, out.begin() + anchor + 1, out.begin() + j);
            }
            i = j;
        }
    }
    out.insert(out.end(), cut.begin(), cut.end());
    return OTT_OK;
}

// tie_order = 1 on whatever `d` selects (the whole store, or one chunk of it): merged or per query
int collect_vecstore(const Ctx& c, const Runner& run, const ott_query_desc& d, bool want_order, std::vector<std::vector<ott_hit>>& groups, ott_stats* st) {
    const bool perq = d.mode == OTT_MODE_PER_QUERY;
    std::vector<ott_hit> all;
    std::vector<uint64_t> per;
    int rc = run_core(run, d, k_plus_one(d.k), false, all, per, st);
    if (rc) return rc;
    const uint32_t ng = perq ? d.nq : 1u;
    std::vector<std::vector<ott_hit>> cand(ng);
    if (perq) {
        size_t o = 0;
        for (uint32_t g = 0; g < ng; g++) {
            cand[g].assign(all.begin() + o, all.begin() + o + (size_t)per[g]);
            o += (size_t)per[g];
        }
    } else {
        cand[0] = std::move(all);
    }
    // the fill phase of every group's collector: ONE flat pass (first k passing pairs in visit order), made on first demand
    std::vector<PairSet> fill(ng);
    bool filled = false;
    auto ensure_fill = [&]() -> int {
        if (filled) return OTT_OK;
        filled = true;
        std::vector<ott_hit> f;
        std::vector<uint64_t> fper;
        int rc2 = run_core(run, d, d.k, true, f, fper, nullptr);
        if (rc2) return rc2;
        if (perq) {
            size_t o = 0;
            for (uint32_t g = 0; g < ng; g++) {
                for (size_t i = 0; i < (size_t)fper[g]; i++) fill[g].insert(std::make_pair(f[o + i].index, f[o + i].query));
                o += (size_t)fper[g];
            }
        } else {
            for (const ott_hit& h : f) fill[0].insert(std::make_pair(h.index, h.query));
        }
        return OTT_OK;
    };
    groups.assign(ng, {});
    for (uint32_t g = 0; g < ng; g++) {
        rc = collector_result(c, cand[g], d.k,
                              [&](PairSet& F) -> int {
                                  int rc2 = ensure_fill();
                                  if (rc2) return rc2;
                                  F = fill[g];
                                  return OTT_OK;
                              },
                              want_order, groups[g]);
        if (rc) return rc;
    }
    return OTT_OK;
}

// tie_order = 2, one group (merged over d's queries): per-chunk collectors, concat in chunk order, stable sort, truncate
int collect_metastore_merged(const Ctx& c, const ott_query_desc& d, std::vector<ott_hit>& out, ott_stats* st) {
    std::vector<ott_hit> L;
    std::vector<uint64_t> per;
    int rc = run_core(c.run, d, k_plus_one(d.k), false, L, per, st);
    if (rc) return rc;
    const size_t k = (size_t)(d.k < L.size() ? d.k : L.size());
    if (L.size() <= d.k || ord(c, L[k]) != ord(c, L[k - 1])) {
        out.assign(L.begin(), L.begin() + k);  // the set is unambiguous (equal scores stay in visit order)
        return OTT_OK;
    }
    // ambiguous cut: the chunks that hold candidates, each as a store of its own (src/meta_compute.rs:153-192)
    const uint64_t cs = c.chunk_size;
    std::set<uint64_t> chunks;
    for (const ott_hit& h : L) chunks.insert((h.index - c.base) / cs);
    std::vector<ott_hit> concat;
    for (uint64_t ch : chunks) {
        ott_query_desc d3 = d;
        d3.mode = OTT_MODE_MERGED;
        const Runner one_chunk = [&c, ch](const ott_query_desc& dd, uint64_t k, bool flat, std::vector<ott_hit>& out, std::vector<uint64_t>& per,
                                          ott_stats* st2) { return c.run_chunk(ch, dd, k, flat, out, per, st2); };
        std::vector<std::vector<ott_hit>> one;
        Ctx cc = c;
        cc.base = c.base + ch * cs;  // the chunk is a VecStore of its own: its collector's 8-row blocks start at its first row
        if ((rc = collect_vecstore(cc, one_chunk, d3, true, one, nullptr))) return rc;
        concat.insert(concat.end(), one[0].begin(), one[0].end());
    }
    // src/meta.rs:702-705: sort by partial_cmp (IEEE order: -0.0 == +0.0), stable here as in the oracle's restatement
    if (c.tmax) std::stable_sort(concat.begin(), concat.end(), [](const ott_hit& a, const ott_hit& b) { return a.score > b.score; });
    else std::stable_sort(concat.begin(), concat.end(), [](const ott_hit& a, const ott_hit& b) { return a.score < b.score; });
    if (concat.size() > d.k) concat.resize((size_t)d.k);
    out = std::move(concat);
    return OTT_OK;
}

}  // namespace

// ---- building blocks for callers that obtain the candidate lists themselves (the sharded query, ott_comm.hip) -----------------
bool ties_ambiguous(bool tmax, const std::vector<ott_hit>& L, uint64_t k) {
    return k > 0 && L.size() > k && ord_of(L[(size_t)k].score, tmax) == ord_of(L[(size_t)k - 1].score, tmax);
}

// L: up to k + 1 candidates in (score, visit order); fill: the first k passing pairs in visit order (needed only when
// ties_ambiguous(L, k)); out: the collector's result, at most k hits
int ties_resolve(ott_store* s, bool tmax, uint64_t base, const std::vector<ott_hit>& L, uint64_t k, const std::vector<ott_hit>* fill,
                 std::vector<ott_hit>& out) {
    (void)s;
    Ctx c;
    c.tmax = tmax;
    c.base = base;
    return collector_result(c, L, k,
                            [&](PairSet& F) -> int {
                                if (!fill) return fail(OTT_ERR_INVALID, "ties_resolve: the fill phase is needed but was not supplied");
                                for (const ott_hit& h : *fill) F.insert(std::make_pair(h.index, h.query));
                                return OTT_OK;
                            },
                            false, out);
}

// The reference's outcome on whatever store `env` describes.  tie_order 1: ONE collector over the store; 2: one per chunk.
int ref_ties_collect(const TieEnv& c, int tie_order, const ott_query_desc* d, ott_hit* out_host, uint64_t cap, uint64_t* n_out, uint64_t* n_per_query,
                     ott_stats* stats_out) {
    if (n_out) *n_out = 0;
    if (n_per_query)
        for (uint32_t i = 0; i < d->nq; i++) n_per_query[i] = 0;
    const bool perq = d->mode == OTT_MODE_PER_QUERY;
    std::vector<std::vector<ott_hit>> groups;
    ott_stats st;
    memset(&st, 0, sizeof(st));
    int rc;
    if (tie_order == 2) {
        if (!perq) {
            groups.assign(1, {});
            if ((rc = collect_metastore_merged(c, *d, groups[0], &st))) return rc;
        } else {
            // per query: every query is a MetaStore query of its own
            groups.assign(d->nq, {});
            for (uint32_t q = 0; q < d->nq; q++) {
                ott_query_desc dq = *d;
                dq.queries = d->queries + (size_t)q * c.dim;
                dq.nq = 1;
                dq.mode = OTT_MODE_MERGED;
                ott_stats sq;
                memset(&sq, 0, sizeof(sq));
                if ((rc = collect_metastore_merged(c, dq, groups[q], &sq))) return rc;
                for (ott_hit& h : groups[q]) h.query = q;
                if (q == 0) st = sq;
                else {
                    st.vectors_compared += sq.vectors_compared;
                    st.score_ns += sq.score_ns; st.merge_ns += sq.merge_ns; st.bytes_scanned += sq.bytes_scanned; st.passes += sq.passes;
                }
            }
        }
    } else {
        if ((rc = collect_vecstore(c, c.run, *d, false, groups, &st))) return rc;
    }
    uint64_t total = 0;
    for (size_t g = 0; g < groups.size(); g++) {
        if (total + groups[g].size() > cap) return fail(OTT_ERR_INVALID, "ott_query: output capacity is smaller than min(k, rows*nq)");
        if (!groups[g].empty()) memcpy(out_host + total, groups[g].data(), groups[g].size() * sizeof(ott_hit));
        if (n_per_query && perq) n_per_query[g] = groups[g].size();
        total += groups[g].size();
    }
    if (n_out) *n_out = total;
    if (stats_out) *stats_out = st;
    return OTT_OK;
}

// one store: candidates come from query_core on the caller's context
int query_ref_ties(ott_store* s, const ott_query_desc* d, ott_hit* out_host, uint64_t cap, uint64_t* n_out, uint64_t* n_per_query,
                   ott_stats* stats_out) {
    TieEnv env;
    env.tmax = d->take == OTT_TAKE_MAX;
    env.base = s->base_offset;
    env.chunk_size = s->chunk_size;
    env.dim = s->dim;
    const auto run_off = [s](const ott_query_desc& dd, uint64_t k, bool flat, uint32_t tie_off, std::vector<ott_hit>& out, std::vector<uint64_t>& per, ott_stats* st) -> int {
        ott_query_desc d2 = dd;
        d2.k = k;
        if (flat) d2.path = OTT_PATH_EXACT;
        const bool perq = dd.mode == OTT_MODE_PER_QUERY;
        const uint64_t rows = s->n;
        const uint64_t pool = perq ? rows : rows * (uint64_t)dd.nq;
        const uint64_t k_eff = k < pool ? k : pool;
        const uint64_t cap2 = (perq ? k_eff * dd.nq : k_eff) + 1;
        out.resize((size_t)cap2);
        per.assign(dd.nq, 0);
        uint64_t n2 = 0;
        CoreOpts co;
        co.tie_sh = 3;
        co.flat = flat;
        co.tie_off = tie_off;
        const int rc = query_core(s, &d2, out.data(), nullptr, cap2, &n2, per.data(), nullptr, st, false, nullptr, co);
        if (rc) return rc;
        out.resize((size_t)n2);
        return OTT_OK;
    };
    env.run = [run_off](const ott_query_desc& dd, uint64_t k, bool flat, std::vector<ott_hit>& out, std::vector<uint64_t>& per, ott_stats* st) -> int {
        return run_off(dd, k, flat, 0, out, per, st);
    };
    env.run_chunk = [s, run_off](uint64_t chunk, const ott_query_desc& dd, uint64_t k, bool flat, std::vector<ott_hit>& out, std::vector<uint64_t>& per,
                              ott_stats* st) -> int {
        const uint64_t n_chunks = (s->n + s->chunk_size - 1) / s->chunk_size;
        std::vector<uint64_t> mask((size_t)((n_chunks + 63) / 64) + 1, 0);
        mask[(size_t)(chunk >> 6)] = 1ull << (chunk & 63);
        ott_query_desc d3 = dd;
        d3.chunk_mask = mask.data();
        const uint32_t off = (uint32_t)((8 - (chunk * s->chunk_size) % 8) % 8);  // blocks counted from the chunk's first (local) row
        return run_off(d3, k, flat, off, out, per, st);
    };
    return ref_ties_collect(env, s->opt.tie_order, d, out_host, cap, n_out, n_per_query, stats_out);
}

}  // namespace ott

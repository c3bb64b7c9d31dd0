/*
 * otters_oracle.c — CPU restatement of the otters hot path.  TEST INFRASTRUCTURE ONLY:
 * see otters_oracle.h for who may call this and for the pinning status.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (no FMA contraction, no reassociation:
 * the reference's Rust source multiplies then adds, vec_compute.rs:12-13, and rustc never
 * contracts or reassociates f32 arithmetic).
 *
 * All citations are file:line in the otters crate (`src/...`).
 */
#include "otters_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * L0 kernels — src/vec_compute.rs
 * ---------------------------------------------------------------------------------------- */

/* wide::f32x8::reduce_add.  AVX path: lo/hi 128-bit halves added lane-wise, then movehl, then
 * the last pair.  Fallback path: f32x8 is two f32x4 {a,b}; a.reduce_add() + b.reduce_add(),
 * each a sequential array sum.  (Third-party; see header.) */
static inline float reduce_add8(const float l[8], int mode) {
    if (mode == OTTO_REDUCE_SEQ4) {
        float a = ((l[0] + l[1]) + l[2]) + l[3];
        float b = ((l[4] + l[5]) + l[6]) + l[7];
        return a + b;
    }
    return ((l[0] + l[4]) + (l[2] + l[6])) + ((l[1] + l[5]) + (l[3] + l[7]));
}

/* vec_compute.rs:9-22: fold of f32x8 products over chunks_exact(8) (8 independent lane
 * accumulators, multiply then add), reduce_add, plus the sequential sum of the remainder
 * products. */
float otto_dot(const float* a, const float* b, size_t dim, int reduce_mode) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t full = dim / 8;
    for (size_t j = 0; j < full; j++) {
        const float* pa = a + 8 * j;
        const float* pb = b + 8 * j;
        for (int l = 0; l < 8; l++) {
            float prod = pa[l] * pb[l];
            acc[l] = acc[l] + prod;
        }
    }
    float tail = 0.0f;
    for (size_t i = full * 8; i < dim; i++) {
        float prod = a[i] * b[i];
        tail = tail + prod;
    }
    return reduce_add8(acc, reduce_mode) + tail;
}

/* vec_compute.rs:35-54: same structure on (a-b)^2; no sqrt. */
float otto_l2sq(const float* a, const float* b, size_t dim, int reduce_mode) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t full = dim / 8;
    for (size_t j = 0; j < full; j++) {
        const float* pa = a + 8 * j;
        const float* pb = b + 8 * j;
        for (int l = 0; l < 8; l++) {
            float d = pa[l] - pb[l];
            float sq = d * d;
            acc[l] = acc[l] + sq;
        }
    }
    float tail = 0.0f;
    for (size_t i = full * 8; i < dim; i++) {
        float d = a[i] - b[i];
        float sq = d * d;
        tail = tail + sq;
    }
    return reduce_add8(acc, reduce_mode) + tail;
}

/* vec_compute.rs:25-32: dot * inv1 * inv2, left to right. */
float otto_cosine(const float* a, const float* b, size_t dim, float inv_a, float inv_b, int reduce_mode) {
    float d = otto_dot(a, b, dim, reduce_mode);
    float t = d * inv_a;
    return t * inv_b;
}

/* vec.rs:365-367 (store side) and vec.rs:127-133, 390-396 (query side):
 * norm = sqrt(sequential sum of x*x); inv = norm != 0 ? 1/norm : 0. */
float otto_inv_norm(const float* v, size_t dim) {
    float s = 0.0f;
    for (size_t i = 0; i < dim; i++) {
        float sq = v[i] * v[i];
        s = s + sq;
    }
    float norm = sqrtf(s);
    return norm != 0.0f ? 1.0f / norm : 0.0f;
}

void otto_inv_norms(const float* rows, size_t n, size_t dim, float* out) {
    for (size_t r = 0; r < n; r++) out[r] = otto_inv_norm(rows + r * dim, dim);
}

static inline float score_one(const float* q, const float* v, size_t dim, int metric, float q_inv, float v_inv, int rm) {
    switch (metric) {
        case OTTO_METRIC_COSINE: return otto_cosine(q, v, dim, q_inv, v_inv, rm); /* vec.rs:257-259 */
        case OTTO_METRIC_EUCLIDEAN: return otto_l2sq(q, v, dim, rm);             /* vec.rs:260 */
        default: return otto_dot(q, v, dim, rm);                                  /* vec.rs:261 */
    }
}

/* vec_compute.rs:56-64 (SIMD compare: false on NaN) and :219-225 (scalar form). */
static inline int cmp_holds(float s, int cmp, float thr) {
    switch (cmp) {
        case OTTO_CMP_LT: return s < thr;
        case OTTO_CMP_GT: return s > thr;
        case OTTO_CMP_LTE: return s <= thr;
        case OTTO_CMP_GTE: return s >= thr;
        case OTTO_CMP_EQ: return s == thr;
        default: return 1;
    }
}

/* ------------------------------------------------------------------------------------------
 * TopKCollector — src/vec_compute.rs:76-294
 * ---------------------------------------------------------------------------------------- */

typedef struct {
    uint64_t idx;
    float score;
    uint32_t q;
} ent;

/* f32::total_cmp as a key: ascending unsigned order == total_cmp ascending. */
static inline uint32_t total_key(float f) {
    uint32_t b;
    memcpy(&b, &f, 4);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

/* canonical "a strictly before b" for the given take type: better score first (total
 * order on the bits), then lower row index, then lower query index. */
static inline int canon_before(const ent* a, const ent* b, int take) {
    uint32_t ka = total_key(a->score), kb = total_key(b->score);
    if (ka != kb) return take == OTTO_TAKE_MAX ? ka > kb : ka < kb;
    if (a->idx != b->idx) return a->idx < b->idx;
    return a->q < b->q;
}

typedef struct {
    ent* buf;        /* buffer: Vec<(usize,f32)>, :78 */
    size_t len, k;   /* :79 */
    int take;        /* :80 */
    int has_filter;  /* :81 */
    float f_thr;
    int f_cmp;
    int is_sorted;   /* :82 */
    float threshold; /* :83 */
    int has_eff;     /* effective_threshold / effective_cmp: Option<..>, :84-85 */
    float eff_thr;
    int eff_cmp;
    int ties;
} collector;

/* TopKCollector::new, :89-125 */
static void coll_new(collector* c, size_t k, size_t cap, int take, int f_cmp, float f_thr, int ties) {
    c->buf = (ent*)malloc((cap ? cap : 1) * sizeof(ent));
    c->len = 0;
    c->k = k;
    c->take = take;
    c->has_filter = f_cmp != OTTO_CMP_NONE;
    c->f_thr = f_thr;
    c->f_cmp = f_cmp;
    c->is_sorted = 1;
    c->threshold = take == OTTO_TAKE_MIN ? INFINITY : -INFINITY; /* :90-93 */
    c->ties = ties;
    if (c->has_filter) { /* :96-111 */
        float comb = f_thr;
        if (take == OTTO_TAKE_MIN && (f_cmp == OTTO_CMP_LT || f_cmp == OTTO_CMP_LTE)) comb = fminf(f_thr, c->threshold);
        else if (take == OTTO_TAKE_MAX && (f_cmp == OTTO_CMP_GT || f_cmp == OTTO_CMP_GTE)) comb = fmaxf(f_thr, c->threshold);
        c->has_eff = 1;
        c->eff_thr = comb;
        c->eff_cmp = f_cmp;
    } else { /* :112 */
        c->has_eff = 0;
        c->eff_thr = 0;
        c->eff_cmp = OTTO_CMP_NONE;
    }
}

/* get_effective_threshold, :127-141.  Returns 0 when there is no block test. */
static int coll_get_eff(const collector* c, float* thr, int* cmp) {
    if (c->has_eff) {
        *thr = c->eff_thr;
        *cmp = c->eff_cmp;
        return 1;
    }
    if (c->len == c->k) {
        *thr = c->threshold;
        *cmp = c->take == OTTO_TAKE_MIN ? OTTO_CMP_LT : OTTO_CMP_GT;
        return 1;
    }
    return 0;
}

/* update_effective_threshold, :143-165 */
static void coll_update_eff(collector* c) {
    if (c->len != c->k) return;
    if (c->has_eff) {
        if (c->take == OTTO_TAKE_MIN && (c->eff_cmp == OTTO_CMP_LT || c->eff_cmp == OTTO_CMP_LTE))
            c->eff_thr = fminf(c->eff_thr, c->threshold);
        else if (c->take == OTTO_TAKE_MAX && (c->eff_cmp == OTTO_CMP_GT || c->eff_cmp == OTTO_CMP_GTE))
            c->eff_thr = fmaxf(c->eff_thr, c->threshold);
    } else {
        c->has_eff = 1;
        c->eff_thr = c->threshold;
        c->eff_cmp = c->take == OTTO_TAKE_MIN ? OTTO_CMP_LT : OTTO_CMP_GT;
    }
}

/* sort, :279-288: sort_unstable_by(total_cmp) (reversed for Max).  Unstable in the reference,
 * so tie order is unspecified; this restatement uses a stable insertion/merge sort. */
static int ent_before_literal(const ent* a, const ent* b, int take) {
    uint32_t ka = total_key(a->score), kb = total_key(b->score);
    return take == OTTO_TAKE_MAX ? ka > kb : ka < kb;
}

static void stable_sort(ent* a, size_t n, int take, int canonical) {
    if (n < 2) return;
    ent* tmp = (ent*)malloc(n * sizeof(ent));
    for (size_t w = 1; w < n; w *= 2) {
        for (size_t lo = 0; lo < n; lo += 2 * w) {
            size_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            size_t i = lo, j = mid, o = lo;
            while (i < mid && j < hi) {
                int take_right = canonical ? canon_before(&a[j], &a[i], take) : ent_before_literal(&a[j], &a[i], take);
                tmp[o++] = take_right ? a[j++] : a[i++];
            }
            while (i < mid) tmp[o++] = a[i++];
            while (j < hi) tmp[o++] = a[j++];
        }
        memcpy(a, tmp, n * sizeof(ent));
    }
    free(tmp);
}

static void coll_sort(collector* c) {
    if (!c->is_sorted) {
        stable_sort(c->buf, c->len, c->take, c->ties == OTTO_TIES_CANONICAL);
        c->is_sorted = 1;
    }
}

/* find_insert_position, :270-277: slice::binary_search_by (std >= 1.82 loop shape, MSRV is
 * 1.88, Cargo.toml:10) with Min: probe.total_cmp(score), Max: score.total_cmp(probe);
 * Ok(i) | Err(i) both give i. */
static size_t coll_find_pos(const collector* c, const ent* x) {
    size_t size = c->len;
    if (size == 0) return 0;
    size_t base = 0;
    if (c->ties == OTTO_TIES_CANONICAL) {
        /* first position whose element does not come before x */
        size_t lo = 0, hi = size;
        while (lo < hi) {
            size_t mid = lo + (hi - lo) / 2;
            if (canon_before(&c->buf[mid], x, c->take)) lo = mid + 1;
            else hi = mid;
        }
        return lo;
    }
    uint32_t ks = total_key(x->score);
    while (size > 1) {
        size_t half = size / 2, mid = base + half;
        uint32_t kp = total_key(c->buf[mid].score);
        /* cmp = f(probe): Min -> probe vs score, Max -> score vs probe */
        int greater = c->take == OTTO_TAKE_MIN ? kp > ks : ks > kp;
        base = greater ? base : mid;
        size -= half;
    }
    uint32_t kp = total_key(c->buf[base].score);
    int equal = kp == ks;
    int less = c->take == OTTO_TAKE_MIN ? kp < ks : ks < kp;
    if (equal) return base;
    return base + (less ? 1 : 0);
}

/* push_single, :236-268 */
static void coll_push_single(collector* c, uint64_t idx, float score, uint32_t q) {
    if (isnan(score)) return; /* :237-239 */
    ent x = {idx, score, q};
    if (c->len == c->k) {
        int should;
        if (c->ties == OTTO_TIES_CANONICAL) should = canon_before(&x, &c->buf[c->k - 1], c->take);
        else should = c->take == OTTO_TAKE_MIN ? score < c->threshold : score > c->threshold; /* :243-246 */
        if (should) {
            size_t pos = coll_find_pos(c, &x);
            memmove(&c->buf[pos + 1], &c->buf[pos], (c->len - 1 - pos) * sizeof(ent)); /* insert + pop, :250-251 */
            c->buf[pos] = x;
            c->threshold = c->buf[c->k - 1].score; /* :252 */
            coll_update_eff(c);
        }
    } else {
        c->buf[c->len++] = x; /* :258-259 */
        c->is_sorted = 0;
        if (c->len == c->k) { /* :261-265 */
            coll_sort(c);
            c->threshold = c->buf[c->k - 1].score;
            coll_update_eff(c);
        }
    }
}

/* push_chunk_masked, :168-208.  rowmask NULL = None. */
static void coll_push_chunk_masked(collector* c, size_t chunk_idx, const float scores[8], const int* rowmask, uint32_t q) {
    if (c->k == 0) return;
    unsigned tbits = 0xFF;
    if (c->ties == OTTO_TIES_CANONICAL) {
        /* canonical form: the plain filter; top-k gating happens in push_single */
        if (c->has_filter) {
            tbits = 0;
            for (int i = 0; i < 8; i++)
                if (cmp_holds(scores[i], c->f_cmp, c->f_thr)) tbits |= 1u << i;
        }
    } else {
        float thr;
        int cmp;
        if (coll_get_eff(c, &thr, &cmp)) { /* :179-182, filter_mask_bits :56-74 */
            tbits = 0;
            for (int i = 0; i < 8; i++)
                if (cmp_holds(scores[i], cmp, thr)) tbits |= 1u << i;
        }
    }
    unsigned sbits = 0xFF;
    if (rowmask) { /* :185-196 */
        sbits = 0;
        for (int i = 0; i < 8; i++)
            if (rowmask[i]) sbits |= 1u << i;
    }
    unsigned bits = tbits & sbits;
    if (!bits) return;
    for (int i = 0; i < 8; i++)
        if ((bits >> i) & 1) coll_push_single(c, chunk_idx * 8 + (size_t)i, scores[i], q); /* :203-207 */
}

/* push_scalars, :210-234: the user filter only, then push_single */
static void coll_push_scalar(collector* c, uint64_t idx, float score, uint32_t q) {
    if (c->k == 0) return;
    if (c->has_filter && !cmp_holds(score, c->f_cmp, c->f_thr)) return;
    coll_push_single(c, idx, score, q);
}

/* BitVec::get(i).unwrap_or(true), vec.rs:234, 295-298 */
static inline int mask_keep(const uint64_t* m, size_t bits, size_t bit_off, size_t i) {
    if (!m) return 1;
    if (i >= bits) return 1;
    size_t b = bit_off + i;
    return (int)((m[b >> 6] >> (b & 63)) & 1);
}

/* ------------------------------------------------------------------------------------------
 * VecQueryPlan::collect — src/vec.rs:206-311 (validation lives in the host layer)
 * ---------------------------------------------------------------------------------------- */
static size_t vec_query_impl(const float* rows, const float* inv_norms, size_t n, size_t dim,
                             const float* queries, const float* q_inv, size_t nq, int metric, int take, size_t k,
                             int f_cmp, float f_thr, const uint64_t* row_mask, size_t mask_bits, size_t mask_off,
                             int rm, int ties, ent** out_buf) {
    collector c;
    size_t cap = k < n * nq ? k : n * nq;
    coll_new(&c, k, cap + 1, take, f_cmp, f_thr, ties);

    size_t full_chunks = n / 8; /* :222 */
    for (size_t chunk_idx = 0; chunk_idx < full_chunks; chunk_idx++) {
        size_t base_row = chunk_idx * 8;
        int bm[8];
        if (row_mask) /* :231-237 */
            for (int i = 0; i < 8; i++) bm[i] = mask_keep(row_mask, mask_bits, mask_off, base_row + (size_t)i);
        float scratch[8] = {0, 0, 0, 0, 0, 0, 0, 0}; /* :240 */
        for (size_t qi = 0; qi < nq; qi++) { /* :243-266 */
            const float* q = queries + qi * dim;
            for (int i = 0; i < 8; i++) {
                if (row_mask && !bm[i]) continue; /* :248-252 */
                size_t row = base_row + (size_t)i;
                scratch[i] = score_one(q, rows + row * dim, dim, metric, q_inv[qi], inv_norms[row], rm);
            }
            coll_push_chunk_masked(&c, chunk_idx, scratch, row_mask ? bm : NULL, (uint32_t)qi);
        }
    }
    size_t rem_start = full_chunks * 8; /* :270-303 */
    if (rem_start < n) {
        for (size_t qi = 0; qi < nq; qi++) {
            const float* q = queries + qi * dim;
            for (size_t row = rem_start; row < n; row++) {
                float s = score_one(q, rows + row * dim, dim, metric, q_inv[qi], inv_norms[row], rm);
                if (!mask_keep(row_mask, mask_bits, mask_off, row)) continue; /* :294-299 */
                coll_push_scalar(&c, row, s, (uint32_t)qi);
            }
        }
    }
    coll_sort(&c); /* into_sorted_vec, :290-293 */
    *out_buf = c.buf;
    return c.len;
}

static void query_inv_norms(const float* queries, size_t nq, size_t dim, float* out) {
    for (size_t i = 0; i < nq; i++) out[i] = otto_inv_norm(queries + i * dim, dim); /* vec.rs:390-396 */
}

size_t otto_vec_query(const float* rows, const float* inv_norms, size_t n, size_t dim,
                      const float* queries, size_t nq, int metric, int take, size_t k,
                      int filter_cmp, float filter_thr, const uint64_t* row_mask, size_t row_mask_bits,
                      int reduce_mode, int ties, otto_hit* out) {
    float* q_inv = (float*)malloc((nq ? nq : 1) * sizeof(float));
    query_inv_norms(queries, nq, dim, q_inv);
    ent* buf = NULL;
    size_t m = vec_query_impl(rows, inv_norms, n, dim, queries, q_inv, nq, metric, take, k, filter_cmp, filter_thr,
                              row_mask, row_mask_bits, 0, reduce_mode, ties, &buf);
    for (size_t i = 0; i < m; i++) {
        out[i].index = buf[i].idx;
        out[i].score = buf[i].score;
        out[i].query = buf[i].q;
    }
    free(buf);
    free(q_inv);
    return m;
}

/* ------------------------------------------------------------------------------------------
 * MetaQueryPlan::collect score + merge — src/meta.rs:646-721, process_chunk
 * src/meta_compute.rs:153-192
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const float *rows, *inv_norms, *queries, *q_inv;
    size_t n, dim, chunk_size, nq, k;
    int metric, take, f_cmp, rm, ties;
    float f_thr;
    const uint64_t* row_mask;
    const size_t* cand; /* candidate chunk ids */
    size_t n_cand;
    ent** res;          /* per candidate chunk results */
    size_t* res_len;
    int tid, n_threads;
    size_t* next; /* shared: the next candidate chunk nobody has taken yet */
} meta_job;

/* rayon's par_iter (src/meta.rs:678) balances the chunks over the pool's threads by work stealing; here every thread takes the
 * next chunk nobody has taken (one atomic per task) — the same effect for equal-sized tasks, and no thread idles while another
 * still holds several untouched chunks (the round-robin deal this replaced did that whenever tasks per thread was small). */
static void* meta_worker(void* arg) {
    meta_job* j = (meta_job*)arg;
    for (;;) {
        const size_t ci = __atomic_fetch_add(j->next, 1, __ATOMIC_RELAXED);
        if (ci >= j->n_cand) break;
        size_t c = j->cand[ci];
        size_t base = c * j->chunk_size; /* MetaChunk.base_offset, meta.rs:273-280 */
        size_t len = j->n - base < j->chunk_size ? j->n - base : j->chunk_size;
        /* the reference recomputes the query inv-norms per chunk (meta_compute.rs:173 ->
         * vec.rs:390-396); same values, so they are computed once here. */
        ent* buf = NULL;
        size_t m = vec_query_impl(j->rows + base * j->dim, j->inv_norms + base, len, j->dim, j->queries, j->q_inv, j->nq,
                                  j->metric, j->take, j->k, j->f_cmp, j->f_thr, j->row_mask, j->row_mask ? len : 0, base,
                                  j->rm, j->ties, &buf);
        for (size_t i = 0; i < m; i++) buf[i].idx += base; /* meta_compute.rs:184-188 */
        j->res[ci] = buf;
        j->res_len[ci] = m;
    }
    return NULL;
}

size_t otto_meta_query(const float* rows, const float* inv_norms, size_t n, size_t dim, size_t chunk_size,
                       const float* queries, size_t nq, int metric, int take, size_t k,
                       int filter_cmp, float filter_thr, const uint64_t* chunk_mask, const uint64_t* row_mask,
                       int reduce_mode, int ties, int n_threads, otto_hit* out, otto_stats* stats) {
    size_t n_chunks = chunk_size ? (n + chunk_size - 1) / chunk_size : 0;
    size_t* cand = (size_t*)malloc((n_chunks ? n_chunks : 1) * sizeof(size_t));
    size_t n_cand = 0;
    for (size_t c = 0; c < n_chunks; c++) /* meta.rs:648-659 */
        if (!chunk_mask || ((chunk_mask[c >> 6] >> (c & 63)) & 1)) cand[n_cand++] = c;

    float* q_inv = (float*)malloc((nq ? nq : 1) * sizeof(float));
    query_inv_norms(queries, nq, dim, q_inv);
    ent** res = (ent**)calloc(n_cand ? n_cand : 1, sizeof(ent*));
    size_t* res_len = (size_t*)calloc(n_cand ? n_cand : 1, sizeof(size_t));

    if (n_threads < 1) n_threads = 1;
    if ((size_t)n_threads > n_cand) n_threads = n_cand ? (int)n_cand : 1;
    meta_job* jobs = (meta_job*)malloc((size_t)n_threads * sizeof(meta_job));
    pthread_t* th = (pthread_t*)malloc((size_t)n_threads * sizeof(pthread_t));
    size_t next_task = 0;
    for (int t = 0; t < n_threads; t++) {
        meta_job jb = {rows, inv_norms, queries, q_inv, n, dim, chunk_size, nq, k, metric, take, filter_cmp, reduce_mode,
                       ties, filter_thr, row_mask, cand, n_cand, res, res_len, t, n_threads, &next_task};
        jobs[t] = jb;
    }
    if (n_threads == 1) meta_worker(&jobs[0]);
    else {
        for (int t = 0; t < n_threads; t++) pthread_create(&th[t], NULL, meta_worker, &jobs[t]);
        for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    }

    /* meta.rs:693-696: extend in chunk order; vectors_compared = sum len*nq (meta_compute.rs:166) */
    size_t total = 0, compared = 0;
    for (size_t ci = 0; ci < n_cand; ci++) {
        total += res_len[ci];
        size_t base = cand[ci] * chunk_size;
        size_t len = n - base < chunk_size ? n - base : chunk_size;
        compared += len * nq;
    }
    ent* agg = (ent*)malloc((total ? total : 1) * sizeof(ent));
    size_t o = 0;
    for (size_t ci = 0; ci < n_cand; ci++) {
        if (res_len[ci]) memcpy(agg + o, res[ci], res_len[ci] * sizeof(ent));
        o += res_len[ci];
        free(res[ci]);
    }
    /* meta.rs:702-708: sort_unstable_by(partial_cmp) then truncate(k).  partial_cmp orders
     * -0.0 == +0.0; no NaN can be present (vec_compute.rs:237).  Stable here. */
    if (ties == OTTO_TIES_CANONICAL) stable_sort(agg, total, take, 1);
    else {
        /* stable merge sort on IEEE '<' */
        ent* tmp = (ent*)malloc((total ? total : 1) * sizeof(ent));
        for (size_t w = 1; w < total; w *= 2) {
            for (size_t lo = 0; lo < total; lo += 2 * w) {
                size_t mid = lo + w < total ? lo + w : total, hi = lo + 2 * w < total ? lo + 2 * w : total;
                size_t i = lo, jx = mid, oo = lo;
                while (i < mid && jx < hi) {
                    int right = take == OTTO_TAKE_MIN ? agg[jx].score < agg[i].score : agg[jx].score > agg[i].score;
                    tmp[oo++] = right ? agg[jx++] : agg[i++];
                }
                while (i < mid) tmp[oo++] = agg[i++];
                while (jx < hi) tmp[oo++] = agg[jx++];
            }
            memcpy(agg, tmp, total * sizeof(ent));
        }
        free(tmp);
    }
    size_t m = total > k ? k : total;
    for (size_t i = 0; i < m; i++) {
        out[i].index = agg[i].idx;
        out[i].score = agg[i].score;
        out[i].query = agg[i].q;
    }
    if (stats) { /* meta.rs:666-669, 711-720 */
        stats->total_chunks = n_chunks;
        stats->evaluated_chunks = n_cand;
        stats->pruned_chunks = n_chunks - n_cand;
        stats->vectors_compared = compared;
    }
    free(agg);
    free(jobs);
    free(th);
    free(res);
    free(res_len);
    free(q_inv);
    free(cand);
    return m;
}

/* ------------------------------------------------------------------------------------------
 * zonemap chunk test — src/type_utils.rs:447-584 (8-wide) and :740-889 (drivers + scalar tail).
 * The 8-wide and scalar forms compute the same predicate (lane j <-> bit j), restated once.
 * ---------------------------------------------------------------------------------------- */
#define SET_BIT(out, i) ((out)[(i) >> 6] |= (uint64_t)1 << ((i) & 63))

#define DEF_CHUNK_MASK(NAME, T)                                                                                      \
    void NAME(const T* mn, const T* mx, const uint64_t* non_null, size_t n_chunks, int op, T thr, uint64_t* out) {    \
        for (size_t i = 0; i < n_chunks; i++) {                                                                      \
            int sat;                                                                                                 \
            switch (op) { /* type_utils.rs:762-769 */                                                                \
                case OTTO_OP_EQ: sat = mn[i] <= thr && thr <= mx[i]; break;                                          \
                case OTTO_OP_LT: sat = mn[i] < thr; break;                                                           \
                case OTTO_OP_LTE: sat = mn[i] <= thr; break;                                                         \
                case OTTO_OP_GT: sat = mx[i] > thr; break;                                                           \
                case OTTO_OP_GTE: sat = mx[i] >= thr; break;                                                         \
                default: sat = 1; break; /* Neq: always, still gated by non_null */                                  \
            }                                                                                                        \
            if (sat && non_null[i] > 0) SET_BIT(out, i);                                                             \
        }                                                                                                            \
    }
DEF_CHUNK_MASK(otto_chunk_mask_i32, int32_t)
DEF_CHUNK_MASK(otto_chunk_mask_i64, int64_t)
DEF_CHUNK_MASK(otto_chunk_mask_f32, float)
DEF_CHUNK_MASK(otto_chunk_mask_f64, double)

/* row test — src/type_utils.rs:306-444 (8-wide), :587-736 (drivers + scalar tail):
 * plain comparison AND not-null; nulls.get(i).unwrap_or(false). */
static inline int is_null(const uint64_t* nulls, size_t null_bits, size_t i) {
    if (!nulls || i >= null_bits) return 0;
    return (int)((nulls[i >> 6] >> (i & 63)) & 1);
}

#define DEF_ROWS_MASK(NAME, T)                                                                                        \
    void NAME(const T* vals, const uint64_t* nulls, size_t null_bits, size_t base, size_t len, int op, T thr,          \
              uint64_t* out) {                                                                                        \
        for (size_t off = 0; off < len; off++) {                                                                      \
            T v = vals[base + off];                                                                                   \
            int sat;                                                                                                  \
            switch (op) { /* type_utils.rs:609-616 */                                                                 \
                case OTTO_OP_EQ: sat = v == thr; break;                                                               \
                case OTTO_OP_NEQ: sat = v != thr; break;                                                              \
                case OTTO_OP_LT: sat = v < thr; break;                                                                \
                case OTTO_OP_LTE: sat = v <= thr; break;                                                              \
                case OTTO_OP_GT: sat = v > thr; break;                                                                \
                default: sat = v >= thr; break;                                                                       \
            }                                                                                                         \
            if (sat && !is_null(nulls, null_bits, base + off)) SET_BIT(out, off);                                     \
        }                                                                                                             \
    }
DEF_ROWS_MASK(otto_rows_mask_i32, int32_t)
DEF_ROWS_MASK(otto_rows_mask_i64, int64_t)
DEF_ROWS_MASK(otto_rows_mask_f32, float)
DEF_ROWS_MASK(otto_rows_mask_f64, double)

/* zone statistics — src/meta_compute.rs:41-98, 117-130: fold over non-null rows. */
#define DEF_ZONE_INT(NAME, T)                                                                                         \
    void NAME(const T* vals, const uint64_t* nulls, size_t null_bits, size_t start, size_t end, int64_t* mn,           \
              int64_t* mx, uint64_t* non_null) {                                                                      \
        int64_t lo = INT64_MAX, hi = INT64_MIN;                                                                       \
        uint64_t cnt = 0;                                                                                             \
        for (size_t i = start; i < end; i++) {                                                                        \
            if (is_null(nulls, null_bits, i)) continue;                                                               \
            int64_t v = (int64_t)vals[i];                                                                             \
            if (v < lo) lo = v;                                                                                       \
            if (v > hi) hi = v;                                                                                       \
            cnt++;                                                                                                    \
        }                                                                                                             \
        *mn = lo;                                                                                                     \
        *mx = hi;                                                                                                     \
        *non_null = cnt;                                                                                              \
    }
DEF_ZONE_INT(otto_zone_stat_i32, int32_t)
DEF_ZONE_INT(otto_zone_stat_i64, int64_t)

/* f64::min / f64::max ignore a NaN operand (== fmin/fmax). */
#define DEF_ZONE_FLT(NAME, T)                                                                                         \
    void NAME(const T* vals, const uint64_t* nulls, size_t null_bits, size_t start, size_t end, double* mn,            \
              double* mx, uint64_t* non_null) {                                                                       \
        double lo = INFINITY, hi = -INFINITY;                                                                         \
        uint64_t cnt = 0;                                                                                             \
        for (size_t i = start; i < end; i++) {                                                                        \
            if (is_null(nulls, null_bits, i)) continue;                                                               \
            double v = (double)vals[i];                                                                               \
            lo = fmin(lo, v);                                                                                         \
            hi = fmax(hi, v);                                                                                         \
            cnt++;                                                                                                    \
        }                                                                                                             \
        *mn = lo;                                                                                                     \
        *mx = hi;                                                                                                     \
        *non_null = cnt;                                                                                              \
    }
DEF_ZONE_FLT(otto_zone_stat_f32, float)
DEF_ZONE_FLT(otto_zone_stat_f64, double)

/* ------------------------------------------------------------------------------------------
 * synthetic corpus generator (shared definition with libotters_hip.so)
 * ---------------------------------------------------------------------------------------- */
float otto_rand_elem(uint64_t seed, uint64_t linear_index) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (linear_index + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    uint32_t u = (uint32_t)(z >> 40); /* 24 bits */
    return (float)u * (1.0f / 8388608.0f) - 1.0f; /* exact: [-1, 1) in steps of 2^-23 */
}

/* Clustered synthetic rows, bit-identical to the library's clustered_fill_kernel (ott_store_append_clustered): the file is
 * built with -ffp-contract=off, so centre + (spread * u) * w is three separately rounded operations on both sides. */
static uint64_t otto_mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

float otto_clustered_elem(uint64_t seed, uint64_t row, uint64_t c, uint64_t dim, uint32_t n_clusters, float spread, float aniso) {
    uint64_t cl = otto_mix64(seed + 0xC1057E25ull + 0x9E3779B97F4A7C15ull * (row + 1)) % n_clusters;
    float centre = otto_rand_elem(seed + 0x5EEDull, cl * dim + c);
    float u = otto_rand_elem(seed, row * dim + c);
    volatile float t = (float)c / (float)dim;
    volatile float den = 1.0f + aniso * t;
    volatile float w = 1.0f / den;
    volatile float su = spread * u;
    volatile float suw = su * w;
    return centre + suw;
}

void otto_clustered_fill(float* out, uint64_t first_row, uint64_t n_rows, uint64_t dim, uint64_t seed, uint32_t n_clusters, float spread,
                         float aniso) {
    for (uint64_t r = 0; r < n_rows; r++)
        for (uint64_t c = 0; c < dim; c++) out[r * dim + c] = otto_clustered_elem(seed, first_row + r, c, dim, n_clusters, spread, aniso);
}

void otto_rand_fill(float* out, uint64_t first_row, uint64_t n_rows, uint64_t dim, uint64_t seed) {
    for (uint64_t r = 0; r < n_rows; r++)
        for (uint64_t c = 0; c < dim; c++) out[r * dim + c] = otto_rand_elem(seed, (first_row + r) * dim + c);
}

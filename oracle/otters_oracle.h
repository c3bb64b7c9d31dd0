/*
 * otters_oracle.h — CPU restatement of the otters hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle for the MI355X backend: a plain-C, scalar restatement of
 * the reference's scoring loops, TopKCollector, VecQueryPlan::collect, the per-chunk
 * MetaStore score+merge block and the zonemap / row-mask helpers.  Every function cites
 * the reference file:line (relative to the otters crate root) it follows.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product path (otters_amd + libotters_hip.so) never links or calls it.
 *
 * PINNING STATUS
 *   - Everything the reference's own tests pin (tests/vec_store_tests.rs, meta_tests.rs,
 *     meta_zonemap_tests.rs, README 8x4 example) is reproduced by this oracle: see
 *     tests/golden/ and tests/test_oracle_golden.py.
 *   - Score BITS for dim >= 8 are "parity unpinned": the only order-dependent operation
 *     is wide::f32x8::reduce_add (third-party crate `wide = "0.7.33"`, Cargo.toml:37, no
 *     Cargo.lock, sources absent).  No reference test has dim >= 8.  The oracle therefore
 *     implements BOTH plausible horizontal-sum orders (OTTO_REDUCE_AVX, the order of
 *     wide's AVX code path; OTTO_REDUCE_SEQ4, the order of its two-f32x4 fallback) and the
 *     backend takes the same switch.  The contract for scores is BASELINE's 1e-5.
 *   - Tie order among equal scores is unspecified by the reference (binary_search_by /
 *     sort_unstable_by, vec_compute.rs:270-288, meta.rs:702-705).  OTTO_TIES_LITERAL keeps
 *     the streaming collector's arrival behaviour with a stable sort; OTTO_TIES_CANONICAL
 *     is the total order (score, row index, query index) the GPU backend produces.
 */
#ifndef OTTERS_ORACLE_H
#define OTTERS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* vec.rs:11-16 */
enum { OTTO_METRIC_COSINE = 0, OTTO_METRIC_EUCLIDEAN = 1, OTTO_METRIC_DOT = 2 };
/* vec.rs:18-22 */
enum { OTTO_TAKE_MIN = 0, OTTO_TAKE_MAX = 1 };
/* vec.rs:24-31 (NONE = Option::None filter) */
enum { OTTO_CMP_NONE = 0, OTTO_CMP_LT = 1, OTTO_CMP_GT = 2, OTTO_CMP_LTE = 3, OTTO_CMP_GTE = 4, OTTO_CMP_EQ = 5 };
/* expr.rs:83-91 */
enum { OTTO_OP_EQ = 0, OTTO_OP_NEQ = 1, OTTO_OP_LT = 2, OTTO_OP_LTE = 3, OTTO_OP_GT = 4, OTTO_OP_GTE = 5 };
/* horizontal-sum order of wide::f32x8::reduce_add (see header comment) */
enum { OTTO_REDUCE_AVX = 0, OTTO_REDUCE_SEQ4 = 1 };
enum { OTTO_TIES_LITERAL = 0, OTTO_TIES_CANONICAL = 1 };

typedef struct {
    uint64_t index; /* SearchResult.index, vec.rs:35-38 (global row for MetaStore, meta_compute.rs:185) */
    float score;
    uint32_t query; /* informational; the reference drops it */
} otto_hit;

/* meta.rs:832-842 (durations omitted: the oracle is not the thing measured) */
typedef struct {
    uint64_t total_chunks, pruned_chunks, evaluated_chunks, vectors_compared;
} otto_stats;

/* ---- L0: vec_compute.rs:9-54 ---- */
float otto_dot(const float* a, const float* b, size_t dim, int reduce_mode);
float otto_l2sq(const float* a, const float* b, size_t dim, int reduce_mode);
float otto_cosine(const float* a, const float* b, size_t dim, float inv_a, float inv_b, int reduce_mode);
/* vec.rs:365-367 / 390-396: 1/sqrt(sequential sum x*x), 0.0 for a zero norm */
float otto_inv_norm(const float* v, size_t dim);
void otto_inv_norms(const float* rows, size_t n, size_t dim, float* out);

/* ---- L1: VecQueryPlan::collect, vec.rs:206-311 ----
 * rows [n*dim] row-major, inv_norms [n]; queries [nq*dim]; row_mask: BitVec<usize,Lsb0>
 * words, row_mask_bits valid bits (bit i = row i, 1 = keep, missing bit => keep), NULL = none.
 * out must hold min(k, n*nq) hits.  Returns number of hits. */
size_t otto_vec_query(const float* rows, const float* inv_norms, size_t n, size_t dim,
                      const float* queries, size_t nq, int metric, int take, size_t k,
                      int filter_cmp, float filter_thr,
                      const uint64_t* row_mask, size_t row_mask_bits,
                      int reduce_mode, int ties, otto_hit* out);

/* ---- L2/L3: MetaQueryPlan::collect score+merge block, meta.rs:646-721 with
 * process_chunk, meta_compute.rs:153-192.  chunk_mask: n_chunks bits (NULL = all chunks,
 * i.e. no meta_filter); row_mask: n bits over GLOBAL rows (the concatenation of
 * build_row_mask_for_chunk outputs), NULL = none.  n_threads > 1 fans chunks out over
 * pthreads the way rayon's par_iter does (meta.rs:678-691); results are identical. */
size_t otto_meta_query(const float* rows, const float* inv_norms, size_t n, size_t dim, size_t chunk_size,
                       const float* queries, size_t nq, int metric, int take, size_t k,
                       int filter_cmp, float filter_thr,
                       const uint64_t* chunk_mask, const uint64_t* row_mask,
                       int reduce_mode, int ties, int n_threads,
                       otto_hit* out, otto_stats* stats);

/* ---- zonemap chunk test, type_utils.rs:447-584, 740-889 (ORs into `out` bits) ---- */
void otto_chunk_mask_i32(const int32_t* mn, const int32_t* mx, const uint64_t* non_null, size_t n_chunks, int op, int32_t thr, uint64_t* out);
void otto_chunk_mask_i64(const int64_t* mn, const int64_t* mx, const uint64_t* non_null, size_t n_chunks, int op, int64_t thr, uint64_t* out);
void otto_chunk_mask_f32(const float* mn, const float* mx, const uint64_t* non_null, size_t n_chunks, int op, float thr, uint64_t* out);
void otto_chunk_mask_f64(const double* mn, const double* mx, const uint64_t* non_null, size_t n_chunks, int op, double thr, uint64_t* out);

/* ---- row test, type_utils.rs:306-444, 587-736 (ORs into `out` bits, chunk-local bit off) ----
 * nulls: BitVec words over the whole column (1 = NULL), null_bits valid bits (missing => not null). */
void otto_rows_mask_i32(const int32_t* vals, const uint64_t* nulls, size_t null_bits, size_t base, size_t len, int op, int32_t thr, uint64_t* out);
void otto_rows_mask_i64(const int64_t* vals, const uint64_t* nulls, size_t null_bits, size_t base, size_t len, int op, int64_t thr, uint64_t* out);
void otto_rows_mask_f32(const float* vals, const uint64_t* nulls, size_t null_bits, size_t base, size_t len, int op, float thr, uint64_t* out);
void otto_rows_mask_f64(const double* vals, const uint64_t* nulls, size_t null_bits, size_t base, size_t len, int op, double thr, uint64_t* out);

/* ---- zone statistics, meta_compute.rs:32-132 (numeric types; min/max as i64 or f64) ---- */
void otto_zone_stat_i32(const int32_t* vals, const uint64_t* nulls, size_t null_bits, size_t start, size_t end, int64_t* mn, int64_t* mx, uint64_t* non_null);
void otto_zone_stat_i64(const int64_t* vals, const uint64_t* nulls, size_t null_bits, size_t start, size_t end, int64_t* mn, int64_t* mx, uint64_t* non_null);
void otto_zone_stat_f32(const float* vals, const uint64_t* nulls, size_t null_bits, size_t start, size_t end, double* mn, double* mx, uint64_t* non_null);
void otto_zone_stat_f64(const double* vals, const uint64_t* nulls, size_t null_bits, size_t start, size_t end, double* mn, double* mx, uint64_t* non_null);

/* ---- synthetic corpus: counter-based uniform [-1,1) f32, the distribution of
 * examples/demo.rs:4-7.  element (row, col) of a `dim`-wide matrix = f(seed, row*dim+col);
 * bit-identical to the generator in libotters_hip.so (ott_store_append_random). ---- */
float otto_rand_elem(uint64_t seed, uint64_t linear_index);
void otto_rand_fill(float* out, uint64_t first_row, uint64_t n_rows, uint64_t dim, uint64_t seed);
/* clustered synthetic rows, bit-identical to ott_store_append_clustered (include/otters_hip.h) */
float otto_clustered_elem(uint64_t seed, uint64_t row, uint64_t c, uint64_t dim, uint32_t n_clusters, float spread, float aniso);
void otto_clustered_fill(float* out, uint64_t first_row, uint64_t n_rows, uint64_t dim, uint64_t seed, uint32_t n_clusters, float spread,
                         float aniso);

#ifdef __cplusplus
}
#endif
#endif

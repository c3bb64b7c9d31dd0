/*
 * otters_hip.h — C ABI of libotters_hip.so, the MI355X (gfx950) backend for the otters
 * exact-vector-search hot path.
 *
 * The reference (AtharvBhat/otters, Rust) has no FFI seam; this header defines the one a
 * patched otters would bind with an `extern "C"` block.  It replaces,
 * and only replaces:
 *
 *   VecStore::{new, add_vector, add_vectors, len}          src/vec.rs:346-384
 *   the body of VecQueryPlan::collect                      src/vec.rs:206-311
 *     (dot_product / cosine_similarity / euclidean_distance_squared  src/vec_compute.rs:9-54,
 *      filter_mask_bits :56-74, TopKCollector :76-294)
 *   the score + merge block of MetaQueryPlan::collect      src/meta.rs:671-709
 *     (process_chunk  src/meta_compute.rs:153-192, rayon fan-out src/meta.rs:678-691)
 *   GPU-side evaluation of build_row_mask_for_chunk        src/meta_compute.rs:194-289,
 *                                                          src/type_utils.rs:306-444, 587-736
 *
 * Everything above that (plan builders, validation + error strings, Expr::compile, zonemap
 * pruning build_chunk_mask_for_plan, result materialisation) stays in the host language.
 *
 * Conventions: plain pointers and sizes, no C++/torch types.  Every function returns 0 on
 * success or a negative ott_status; ott_last_error() gives a thread-local message.  Input
 * pointers are borrowed for the duration of the call only.  The library owns all device
 * memory behind the opaque ott_store.  An ott_store lives on one GPU (ott_store_create) or spans
 * several GPUs of the host's one process (ott_store_create_multi: one shard per device, every call
 * below works on it unchanged); a multi-PROCESS job instead holds one single-GPU store per rank,
 * with different base offsets, and joins them with an ott_comm (ott_query_sharded).
 * Threading: queries (ott_query, ott_query_device, ott_merge_hits_device*) may be called on one
 * store from several host threads at once, like the reference's `&self` query (src/vec.rs:387):
 * overlapping calls run on separate streams with their own scratch.  Calls that change the
 * store (append*, reserve, write_rows, set_*, add_column, eval_row_mask, zone_stats) need and
 * take exclusive access (they wait for running queries).  A query that uses the device row
 * mask reads whatever the last ott_store_eval_row_mask left: pair the two calls under one
 * caller-side lock if several threads filter (the reference's MetaStore is !Sync, src/meta.rs:54).
 * Different stores are independent.
 */
#ifndef OTTERS_HIP_H
#define OTTERS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OTT_ABI_VERSION 4

typedef enum {
    OTT_OK = 0,
    OTT_ERR_INVALID = -1, /* bad argument */
    OTT_ERR_HIP = -2,     /* HIP runtime failure (message has the hipError string) */
    OTT_ERR_OOM = -3,
    OTT_ERR_UNSUPPORTED = -4
} ott_status;

/* src/vec.rs:11-16 */
typedef enum { OTT_METRIC_COSINE = 0, OTT_METRIC_EUCLIDEAN = 1, OTT_METRIC_DOT = 2 } ott_metric;
/* src/vec.rs:18-22 */
typedef enum { OTT_TAKE_MIN = 0, OTT_TAKE_MAX = 1 } ott_take;
/* src/vec.rs:24-31; NONE = no filter_criteria */
typedef enum { OTT_CMP_NONE = 0, OTT_CMP_LT = 1, OTT_CMP_GT = 2, OTT_CMP_LTE = 3, OTT_CMP_GTE = 4, OTT_CMP_EQ = 5 } ott_cmp;
/* src/expr.rs:83-91 */
typedef enum { OTT_OP_EQ = 0, OTT_OP_NEQ = 1, OTT_OP_LT = 2, OTT_OP_LTE = 3, OTT_OP_GT = 4, OTT_OP_GTE = 5 } ott_op;
/* src/type_utils.rs:11-19 (String columns stay host-side) */
typedef enum { OTT_DT_INT32 = 0, OTT_DT_INT64 = 1, OTT_DT_FLOAT32 = 2, OTT_DT_FLOAT64 = 3, OTT_DT_DATETIME = 5 } ott_dtype;

/* MERGED is the reference's semantics: one top-k over the flattened nq x n score matrix
 * (src/vec.rs:217-219).  PER_QUERY is an extension: k hits for each query. */
typedef enum { OTT_MODE_MERGED = 0, OTT_MODE_PER_QUERY = 1 } ott_mode;

/* Which scoring kernel family runs.  EXACT scores every row in the reference's summation order (one pass over the f32 rows
 * per 4 queries).  MFMA is the certified cascade, all three metrics, k <= 484: candidate passes over compact copies of the corpus
 * — an int8 plane first (k <= 128: a quarter of the f32 bytes; batches on the matrix cores, a single cosine / dot query as a
 * streaming sweep), a 16-bit hi plane for what that cannot certify, split bf16 behind it — every candidate
 * re-scored in the reference's order, the top-k CERTIFIED against a measured error bound, uncertifiable queries recomputed on
 * EXACT — so both return the same bits.  AUTO: a cost model picks the cheaper one.  A single query takes EXACT (no copy of the
 * corpus is BUILT for the most common call) unless the cascade's first plane is already resident and covers every row (the
 * background build after appends, a batch query or ott_store_prepare_batch made it) and the store is large enough for a quarter
 * of the bytes to pay (~200k x 768 rows): then it takes the cascade, same bits.  Batches: 2+ queries on large stores, 5+
 * elsewhere — except small batches (up to 16 queries) on small stores (up to ~65k rows), which stay on EXACT while its
 * small-store kernel (8 queries per pass) is cheaper than the cascade's fixed cost. */
typedef enum { OTT_PATH_AUTO = 0, OTT_PATH_EXACT = 1, OTT_PATH_MFMA = 2 } ott_path;

/* Horizontal-sum order of wide::f32x8::reduce_add (third-party, unpinned by the reference's tests; see
 * oracle/otters_oracle.h).  It follows the Rust build: AVX = ((l0+l4)+(l2+l6))+((l1+l5)+(l3+l7)), what the crate compiles to
 * under target_feature = "avx" (e.g. -C target-cpu=native); SEQ4 = (((l0+l1)+l2)+l3)+(((l4+l5)+l6)+l7), the two-f32x4
 * fallback of a default x86_64 build (the reference ships no .cargo/config.toml).  bindings/rust/patch/vec_hip.rs selects
 * it by cfg!(target_feature = "avx").  Default: AVX. */
typedef enum { OTT_REDUCE_AVX = 0, OTT_REDUCE_SEQ4 = 1 } ott_reduce;

/* SearchResult, src/vec.rs:34-38; `index` is the global row (shard base + local,
 * src/meta_compute.rs:185); `query` is informational (the reference drops it). */
typedef struct {
    uint64_t index;
    float score;
    uint32_t query;
} ott_hit;

typedef struct {
    const float* queries;       /* [nq * dim] row-major, host memory */
    uint32_t nq;
    uint32_t metric;            /* ott_metric */
    uint32_t take;              /* ott_take */
    uint32_t filter_cmp;        /* ott_cmp */
    float filter_thr;
    uint32_t mode;              /* ott_mode */
    uint64_t k;                 /* take_count */
    const uint64_t* chunk_mask; /* n_chunks bits, BitVec<usize,Lsb0> words; NULL = all chunks.
                                   Output of build_chunk_mask_for_plan, src/meta.rs:407-428 */
    const uint64_t* row_mask;   /* row_mask_bits bits over this store's LOCAL rows, 1 = keep;
                                   rows >= row_mask_bits are kept (src/vec.rs:234). NULL = none */
    uint64_t row_mask_bits;
    uint32_t use_device_row_mask; /* 1 = use the mask last built by ott_store_eval_row_mask */
    uint32_t path;              /* ott_path */
} ott_query_desc;

/* MetaQueryStats, src/meta.rs:832-842, plus device-side facts. */
typedef struct {
    uint64_t total_chunks, pruned_chunks, evaluated_chunks, vectors_compared;
    uint64_t prune_ns, score_ns, merge_ns, total_ns; /* score_ns / merge_ns: hipEvent time of the kernels */
    uint64_t bytes_scanned;     /* algorithmic: 4*dim*rows_scored (+4*rows_scored for cosine), per pass */
    uint32_t path_used;         /* ott_path */
    uint32_t passes;            /* corpus passes (EXACT path: ceil(nq / queries per pass)) */
    uint64_t rescored;          /* MFMA path: candidates re-scored in reference order */
    uint32_t retries;           /* MFMA path: queries no candidate pass could certify, recomputed on the exact path */
    uint32_t refined;           /* MFMA path: queries the hi pass (bf16 hi plane) could not certify, re-run through the split pass */
    float err_ratio_max;        /* MFMA path: max over the re-scored candidates of |approximate - exact score| / eps, eps the error bound
                                   the certification assumes for that query and pass.  <= 1 or the bound did not hold: see
                                   bound_violations */
    uint32_t gate_failed;       /* MFMA path: queries whose speculative emission threshold turned out too tight (fewer rows above it than
                                   the sample of rows seen so far suggested); answered by the next cascade level like any uncertified query */
    uint32_t bound_violations;  /* MFMA path: queries for which a re-scored candidate MEASURED |approximate - exact| > eps, i.e. the error
                                   bound the certification rests on did not hold (never observed; the matrix unit's accumulation order is
                                   not documented, so the library checks).  Such a query is treated as uncertified: next cascade level,
                                   finally the exact-order kernel — the result is the reference's either way */
    uint32_t i8_refined;        /* MFMA path: queries the int8 pass (the cascade's first level at k <= 128) could not certify,
                                   re-run through the hi pass (the field was `reserved` until round 5: same offset, same size) */
    uint64_t exchange_ns;       /* sharded queries (ott_query_sharded, a multi-GPU store): hipEvent time from the end of this GPU's own
                                   scoring to the start of the cross-GPU merge — the candidate exchange plus waiting for slower shards;
                                   merge_ns then includes the cross-GPU merge kernel */
} ott_stats;

/* One leaf of a compiled CNF filter (ColumnFilter::Numeric, src/expr.rs:199-205) bound to a
 * device-resident column; literal already coerced as src/meta_compute.rs:249-283 does. */
typedef struct {
    uint32_t column;  /* id returned by ott_store_add_column */
    uint32_t op;      /* ott_op */
    uint32_t clause;  /* leaves with the same clause id are OR-ed; clauses are AND-ed */
    uint32_t reserved;
    int64_t lit_i64;  /* for INT32 / INT64 / DATETIME columns */
    double lit_f64;   /* for FLOAT32 (narrowed to f32) / FLOAT64 columns */
} ott_leaf;

typedef struct ott_store ott_store;

int ott_abi_version(void);
const char* ott_last_error(void);
int ott_device_count(int* out);

/* VecStore::new, src/vec.rs:348-355.  `device` = HIP device ordinal. */
int ott_store_create(uint32_t dim, int device, ott_store** out);
/* The same store over SEVERAL GPUs of this process: one shard per entry of dev_ids, each holding a contiguous
 * range of chunks in row order (shard g holds lower rows than shard g + 1; ranges start on multiples of lcm(chunk size, 8)
 * rows).  Every function of this header that takes an ott_store works on it unchanged: appends / write_rows / read_rows /
 * add_column / eval_row_mask / zone_stats route by row range, and ott_query is the reference's own fan-out and merge
 * (src/meta.rs:678-709) with GPUs in place of rayon tasks — every shard scores its rows (chunk mask, row mask and metadata
 * columns sliced per shard), ONE exchange of fixed-size candidate blocks to the first shard's GPU, merge_hits_kernel there,
 * one host wait.  Results are bit-identical to the same rows in one single-GPU store (all metrics, k, modes, tie orders).
 * Exchange: a grouped ncclAllGather over one RCCL communicator per device (ncclCommInitAll) when the ordinals are distinct and
 * librccl loads; peer copies ordered by events otherwise (option "multi_transport": 0 automatic, 1 peer copies, 2 RCCL).
 * dev_ids may repeat an ordinal (several shards on one GPU: tests on a one-GPU box, or oversubscription) — then peer copies.
 * Layout: ott_store_reserve(n) plans the even split of n rows and appends fill it in order; without a plan rows go to the last
 * shard that has any and are moved between the GPUs before the next query (hipMemcpyPeerAsync; when a shard holds more than
 * 1.25x its even share; option "multi_rebalance" = 0 turns that off) — results never depend on where rows live.  Rows can no
 * longer move once metadata columns are resident: reserve, set the chunk size and append before ott_store_add_column.
 * Small stores stay on ONE GPU: a shard is brought in per "multi_min_shard_rows" rows (option / OTT_MULTI_MIN_SHARD_ROWS, default
 * 32768; 0 = always split evenly), the others stay empty; while one shard holds every row ott_query is that shard's own query —
 * no fan-out, no exchange (the fan-out over N GPUs costs 50-150 us per query, more than a 10k-row store's whole query).
 * Not on a multi-GPU store: ott_query_device, ott_merge_hits_device*, ott_query_sharded (OTT_ERR_UNSUPPORTED).
 * ott_store_device / ott_store_stream: the first shard's. */
int ott_store_create_multi(uint32_t dim, uint32_t n_dev, const int* dev_ids, ott_store** out);
/* The layout ott_store_reserve(n_rows) plans on a store of n_dev shards with this chunk size, without a store or a GPU: shard g
 * starts at out_first_rows[g] (a multiple of lcm(chunk_size, 8), clamped to n_rows) and ends where shard g + 1 starts (the last
 * one at n_rows; shards the store is too small for — "multi_min_shard_rows", read from the environment like a store created
 * now would — start at n_rows).  For hosts that slice their own per-row data the same way (and for the tests of the layout arithmetic). */
int ott_multi_plan(uint64_t n_rows, uint64_t chunk_size, uint32_t n_dev, uint64_t* out_first_rows);
/* Shards of a store (1 for a single-GPU store) and where shard `shard` lives: its device, its first row (counted from the
 * store's first row) and its row count.  Any out pointer may be NULL. */
int ott_store_shard_count(const ott_store* s);
int ott_store_shard_info(const ott_store* s, uint32_t shard, int* device, uint64_t* first_row, uint64_t* n_rows);
/* "none" (single-GPU store), "peer" or "rccl"; "undecided" before the first query of a multi-GPU store. */
const char* ott_store_transport(const ott_store* s);
int ott_store_destroy(ott_store* s);
/* Pre-size device storage for n_rows rows (avoids re-allocation while appending). */
int ott_store_reserve(ott_store* s, uint64_t n_rows);
/* VecStore::add_vectors, src/vec.rs:357-376: copy rows to HBM (row-major, host pointer)
 * and compute their inverse norms on the GPU in the reference's order.  Appends below 256 KB (VecStore::add_vector is one row
 * per call) are staged in pinned host memory and travel 4 MB at a time, and before anything looks at the rows (queries, reads,
 * columns, other kinds of append, ott_store_len counts them): a single-row append costs a memcpy, not a copy, a kernel and a
 * wait (measured 60 us -> under 1 us per row).  Option "stage_appends" = 0 sends every append at once. */
int ott_store_append(ott_store* s, const float* rows_host, uint64_t n_rows);
/* Same, rows already in device memory of this store's GPU ([n_rows*dim], dense). */
int ott_store_append_device(ott_store* s, const void* rows_dev, uint64_t n_rows);
/* Append synthetic rows: uniform [-1,1) (examples/demo.rs:4-7) from a counter-based
 * generator keyed (seed, global element index); bit-identical to oracle otto_rand_fill. */
int ott_store_append_random(ott_store* s, uint64_t n_rows, uint64_t seed);
/* Append synthetic CLUSTERED rows (the shape of real embedding corpora; what the batch path's cheapest candidate pass may
 * fail to certify): row r belongs to cluster hash(seed, r) % n_clusters, element c = centre[cluster][c] + spread * u * w(c),
 * u uniform [-1,1), centres uniform [-1,1), w(c) = 1 / (1 + aniso * c / dim) (aniso = 0: isotropic).  Counter-based, keyed by
 * the GLOBAL row; bit-identical to oracle otto_clustered_fill, so any row can be regenerated on the host. */
int ott_store_append_clustered(ott_store* s, uint64_t n_rows, uint64_t seed, uint32_t n_clusters, float spread, float aniso);
/* Overwrite existing rows [first, first+n) from host memory (tests plant known vectors). */
int ott_store_write_rows(ott_store* s, uint64_t first_row, const float* rows_host, uint64_t n_rows);
uint64_t ott_store_len(const ott_store* s);  /* VecStore::len, src/vec.rs:378 */
uint32_t ott_store_dim(const ott_store* s);
int ott_store_device(const ott_store* s);
/* MetaStore chunking: chunk c = local rows [c*chunk_size, ...) (src/meta.rs:203-281).  Default 1024. */
int ott_store_set_chunk_size(ott_store* s, uint64_t chunk_size);
/* The certified cascade (query batches; single queries once its first plane is resident) keeps compact copies of the corpus in
 * HBM: the INT8 plane (every row as int8 with one f32 scale: a QUARTER of the f32 rows; every metric at k <= 128), the hi plane
 * (every element as an IEEE half — or bf16, option "hi_fmt" — HALF the size of the f32 rows; built only once a query needs it:
 * k > 128, or what the int8 level could not certify) and, only once a query falls through the hi pass's
 * certification too, the batch image (every row pre-split into bf16 hi + bf16 lo, the SAME size as the f32 rows).  All are
 * extended after appends, refreshed by write_rows, dropped by a reallocation (and FIRST, when the new rows would not fit next to
 * them), and skipped on their own when HBM has no room (the cascade then starts further down and splits the rows in registers).
 * enabled = 0 frees them and keeps them off; 1 (the default) allows them again.  Results never depend on them. */
int ott_store_set_batch_image(ott_store* s, int enabled);
/* Builds (or extends after appends) the cascade's first plane now instead of inside the first batch query (~5 ms per 30 GB of
 * rows for the int8 plane).  Optional: queries do it on demand.  Takes the store like a query does (shared). */
int ott_store_prepare_batch(ott_store* s);
/* 1 when the cascade's first plane exists and covers every row (the next batch query starts scoring at once), else 0.  With option
 * "hi_prebuild" (-1 automatic: stores of 262144 rows and more while the plane takes at most a quarter of the free HBM; 0 never;
 * 1 always) the plane is built or extended in the background right after every append, so a host that loads and then queries
 * normally finds it ready without calling ott_store_prepare_batch. */
int ott_store_batch_ready(const ott_store* s);

/* Behaviour switches of one store.  The library reads the environment exactly once per store, in ott_store_create
 * (OTT_<NAME>=<int> presets the option of the same name); after that only this call changes them — the query path never calls
 * getenv.  Sixteen options (round 5 retired the rest: experiment switches whose measurements are in DESIGN.md 3.4 and profiles/dead_ends_rounds_2_4.md).
 * Behaviour a host may want:
 *   "tie_order"  0 (default): canonical total order — better score, lower row, lower query.  1: the reference's own outcome at
 *                exact score ties, ONE TopKCollector over the store (VecStore, src/vec.rs:217-310, src/vec_compute.rs:236-277).
 *                2: one collector per chunk, then concat-sort-truncate (MetaStore, src/meta.rs:678-709), for any chunk size
 *                (src/meta.rs:86-89).  See INTEGRATION.md 6a.
 *   "hi_fmt"     which compact copies of the corpus the cascade keeps: -1 (default) / 2: an INT8 plane as its first level (one f32
 *                scale per row, a quarter of the f32 bytes; k <= 128) with an IEEE-half plane behind it that is built
 *                only once a query needs it (k > 128, or what the int8 level could not certify); 1: the half plane
 *                alone (11 significant bits; falls back to bf16 by itself on stores whose row norms spread over many binades);
 *                0: a bf16 plane alone.  Takes effect when a plane is (re)built.
 *   "hi_prebuild"  -1 (default) automatic / 0 never / 1 always: the hi plane is built in the background after appends
 *                (ott_store_batch_ready).   "stage_appends"  0: every append goes to the GPU at once (default: small ones are staged).
 *   "multi_transport", "multi_rebalance", "multi_min_shard_rows": the multi-GPU store, see ott_store_create_multi.
 * Which of several equivalent paths runs (results never depend on them; the tests hold each to the oracle):
 *   "exact_small" (-1 auto / 0 streaming kernel / 2 rows8, eight lanes per row: which kernel answers on a small store),
 *   "large_k_from" (k above which host-output queries take the sort path; 0 = default: 512 for one query or a small store,
 *   128 for several queries on a large one), "small_sort" (0: results of up to 16384 (row, query) pairs with k > 512 through
 *   the radix sort instead of the rank sort), "mfma_f32" (batch path: one candidate pass on the f32 matrix pipe),
 *   "no_hi_pass" (batch path starts at the split-bf16 pass), "no_batch_image" (no 16-bit copies of the corpus).
 * Tests only:
 *   "force_fallback"  bit mask of code paths the library otherwise takes only in rare conditions: 1 block lists merged by
 *                insertion, 2 k <= 64 through sorted heads + tree fold, 4 the 256-query blocks of a row tile one after the other,
 *                8 the sort path without its prefix gate, 16 the open first round through cursor atomics, 32 conservative
 *                emission thresholds between the row rounds, 64 the sort path in slices of 2^14 (row, query) pairs.
 *   "eps_scale_ppm"  the batch path's error bound multiplied by this many millionths, to show that a violated bound is noticed.
 *   "multi_fake_distinct"  (environment only, at creation) every shard of a multi-GPU store counts as a device of its own.
 * Kernel tuning and timing ablations ("mfma_wg", "mfma_growth", "mfma_abl", "mfma_debug", "hi_tmin", and each fallback bit under
 * its own name) exist by name only in a library built with -DOTT_MFMA_DEBUG_BUILD.
 * Takes the store exclusively, like append. */
int ott_store_set_option(ott_store* s, const char* name, int64_t value);

/* Global index of local row 0 (shard base for multi-GPU; src/meta_compute.rs:185). */
int ott_store_set_base_offset(ott_store* s, uint64_t base);
int ott_store_set_reduce_order(ott_store* s, uint32_t reduce /* ott_reduce */);
/* Read back for tests / debugging. */
int ott_store_read_rows(const ott_store* s, uint64_t first_row, uint64_t n_rows, float* out_host);
int ott_store_read_inv_norms(const ott_store* s, uint64_t first_row, uint64_t n_rows, float* out_host);

/* Metadata columns resident in HBM for GPU-side row-mask evaluation
 * (src/col.rs storage: values + BitVec null mask, 1 = NULL).  values: n elements of the
 * dtype; nulls may be NULL.  n must equal the store length. */
int ott_store_add_column(ott_store* s, uint32_t dtype, const void* values_host, const uint64_t* nulls,
                         uint64_t n, uint32_t* out_column_id);
/* Zone statistics of an HBM-resident column for every chunk of `chunk_size` rows at once
 * (build_zone_stat_for_range, src/meta_compute.rs:41-98, 117-130): min / max over the non-null
 * rows and the non-null count.  Integer / datetime columns: out_min / out_max are int64[n_chunks]
 * (i64::MAX / i64::MIN for an all-null chunk); float columns: double[n_chunks] (+inf / -inf;
 * f64::min/max semantics: a NaN value is ignored).  out_non_null: uint64[n_chunks].  Host pointers. */
int ott_store_zone_stats(ott_store* s, uint32_t column, uint64_t chunk_size, void* out_min, void* out_max,
                         uint64_t* out_non_null);

/* build_row_mask_for_chunk over all rows at once (src/meta_compute.rs:194-232): CNF of
 * numeric leaves -> device row mask used by queries with use_device_row_mask = 1.
 * If out_host != NULL the mask words ((len+63)/64) are also copied back. */
int ott_store_eval_row_mask(ott_store* s, const ott_leaf* leaves, uint32_t n_leaves, uint32_t n_clauses,
                            uint64_t* out_host);

/* The hot path: VecQueryPlan::collect body (src/vec.rs:206-311) / MetaQueryPlan::collect
 * score+merge (src/meta.rs:671-709).  Writes up to `cap` hits, best first; *n_out = count.
 * cap must be >= min(k, rows*nq) (MERGED) or nq*min(k, rows) (PER_QUERY; hits grouped by
 * query, each group best first, *n_out = total, per-query counts in n_per_query if non-NULL). */
int ott_query(ott_store* s, const ott_query_desc* d, ott_hit* out, uint64_t cap, uint64_t* n_out,
              uint64_t* n_per_query, ott_stats* stats);

/* Same, but the result stays on the GPU: out_dev holds `cap` ott_hit slots in device memory
 * of the store's GPU, padded with sentinel hits (index = UINT64_MAX); *n_out_dev (device
 * uint64) receives the count.  PER_QUERY mode: cap must be a multiple of nq; query q's hits
 * start at slot q * (cap / nq), each group sentinel padded.  Launched on the store's stream; returns once the kernels have
 * completed (the caller's collective runs on another stream).  Used for the multi-GPU
 * all-gather of candidates. */
int ott_query_device(ott_store* s, const ott_query_desc* d, void* out_dev, uint64_t cap, void* n_out_dev,
                     ott_stats* stats);
int ott_store_sync(ott_store* s);
/* HIP stream the store launches on (hipStream_t as void*), for event timing by the caller. */
void* ott_store_stream(ott_store* s);

/* Final merge of candidate lists (src/meta.rs:699-709: concat, sort, truncate(k)) on the
 * GPU: `lists_dev` = n_lists * list_len ott_hit in device memory (sentinels ignored), output
 * k best hits (canonical order) to out_host. */
int ott_merge_hits_device(ott_store* s, const void* lists_dev, uint64_t n_lists, uint64_t list_len,
                          uint32_t take, uint64_t k, ott_hit* out_host, uint64_t* n_out);
/* The same merge for PER_QUERY results: `lists_dev` = [n_lists][n_groups][list_len] ott_hit (what an
 * all-gather of per-GPU ott_query_device PER_QUERY blocks produces, group = query); every group is
 * merged on its own.  out_host receives the groups' hits back to back (at most k each, needs
 * n_groups * min(k, n_lists*list_len) slots), *n_out the total, n_per_group[g] each group's count. */
int ott_merge_hits_device_grouped(ott_store* s, const void* lists_dev, uint64_t n_lists, uint64_t n_groups,
                                  uint64_t list_len, uint32_t take, uint64_t k, ott_hit* out_host,
                                  uint64_t* n_out, uint64_t* n_per_group);

/* ---- multi-GPU: the corpus sharded by contiguous chunk ranges, one process (and one ott_store) per GPU -------------------
 * The reference fans chunks out over a rayon pool and concat-sort-truncates the per-chunk top-k lists
 * (src/meta.rs:678-709).  Across GPUs the same two steps are: every rank scores ITS shard (no data-path collective), then
 * ONE exchange — an all-gather of fixed-size, sentinel-padded per-GPU candidate blocks — and the same merge kernel on
 * every rank.  An ott_comm is that exchange.  Two transports:
 *   RCCL  (ott_comm_create): ncclAllGather over xGMI on the store's stream; score -> gather -> merge run back to back on
 *         that stream with no host synchronisation in between.  librccl.so.1 is dlopen'ed on first use.
 *   HOST  (ott_comm_create_host): the caller supplies an all-gather of host buffers (MPI, gloo, sockets ...); blocks are
 *         staged through pinned memory.  For hosts without RCCL bootstrap and for tests that put two ranks on one GPU
 *         (RCCL refuses duplicate devices). */
typedef struct ott_comm ott_comm;
#define OTT_COMM_ID_BYTES 128
/* all-gather callback of the HOST transport: every rank contributes `bytes` bytes at `send`; `recv` (world * bytes)
 * receives the blocks in rank order.  Returns 0 on success.  Called from the thread that called into the library. */
typedef int (*ott_allgather_fn)(void* user, const void* send, void* recv, uint64_t bytes);

/* ncclGetUniqueId: rank 0 creates the id (OTT_COMM_ID_BYTES bytes) and hands it to the other ranks out of band. */
int ott_comm_unique_id(void* id_out);
/* ncclCommInitRank on `device` (collective: every rank calls it with the same id and world). */
int ott_comm_create(const void* unique_id, int rank, int world, int device, ott_comm** out);
int ott_comm_create_host(int rank, int world, ott_allgather_fn fn, void* user, ott_comm** out);
int ott_comm_destroy(ott_comm* c);
int ott_comm_rank(const ott_comm* c);
int ott_comm_world(const ott_comm* c);
/* "rccl" or "host" */
const char* ott_comm_transport(const ott_comm* c);
/* What the transport itself reports: *nranks = ncclCommCount of the communicator (HOST transport: the world given at creation),
 * *version = ncclGetVersion (e.g. 22606; 0 for the HOST transport).  Either pointer may be NULL. */
int ott_comm_info(const ott_comm* c, int* nranks, int* version);
/* RCCL transport with world > 1: how long ott_comm_create may wait for the other ranks at the rendezvous, and how long a
 * collective (ott_query_sharded, ott_comm_all_gather_host) may stay unfinished, before the call returns OTT_ERR_HIP with a
 * message naming the incomplete exchange instead of waiting for a peer that died.  Default 120 000 ms; the environment
 * variable OTT_COMM_TIMEOUT_MS presets it (read once, in ott_comm_create); 0 = wait for ever.  After such an error the comm
 * is only good for ott_comm_destroy. */
int ott_comm_set_timeout_ms(ott_comm* c, int64_t timeout_ms);
/* All-gather of small HOST buffers over the comm's transport (control data: shard sizes, stats, timing, the
 * materialised cells of the k hits).  recv_host holds world * bytes.  RCCL transport: staged through device memory of
 * the comm's GPU, synchronous.  Also serves as a barrier. */
int ott_comm_all_gather_host(ott_comm* c, const void* send_host, void* recv_host, uint64_t bytes);

/* The hot path across shards: ott_query on this rank's shard, the candidate exchange, the final merge
 * (MetaQueryPlan::collect's score + merge block, src/meta.rs:671-709, with GPUs in place of rayon tasks).  Collective:
 * every rank calls it with the same queries / metric / take / filter / k / mode; chunk_mask, row_mask and
 * use_device_row_mask refer to the rank's own shard.  The store's base offset must be the shard's first global row, and
 * shards must be in rank order (rank r holds lower global rows than rank r + 1: ties then resolve like on one GPU).
 * Every rank receives the same result: up to k hits (MERGED) or k per query (PER_QUERY), best first, in `out`
 * (cap >= k, or nq * k; hits beyond what exists are not written).  `stats` describes this rank's shard.
 * k <= 512: fixed-size blocks, one device merge.  k > 512 (e.g. the reference's default take = every row,
 * src/meta.rs:638-644): every rank's sorted list is exchanged whole (counts first) and merged on the host. */
int ott_query_sharded(ott_store* s, ott_comm* c, const ott_query_desc* d, ott_hit* out, uint64_t cap, uint64_t* n_out,
                      uint64_t* n_per_query, ott_stats* stats);

#ifdef __cplusplus
}
#endif
#endif

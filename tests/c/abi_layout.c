/* abi_layout.c — compiled proof that include/otters_hip.h is plain C11 with the layout the bindings assume.
 *
 * Built with `gcc -std=c11 -pedantic -Wall -Wextra -Werror` (tests/c/Makefile, __graft_entry__.build()).  The
 * _Static_asserts pin every size and offset that otters_amd/_native.py (ctypes) and the Rust `#[repr(C)]` block of
 * INTEGRATION.md section 2 rely on; main() prints them as JSON (tests/test_abi_symbols.py compares the ctypes structs
 * with THESE numbers, not with hand-typed ones) and calls ott_abi_version through the linked library.
 * No GPU is touched. */
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#include "otters_hip.h"

/* SearchResult + query id (src/vec.rs:34-38) */
_Static_assert(sizeof(ott_hit) == 16, "ott_hit is 16 bytes");
_Static_assert(offsetof(ott_hit, index) == 0, "ott_hit.index");
_Static_assert(offsetof(ott_hit, score) == 8, "ott_hit.score");
_Static_assert(offsetof(ott_hit, query) == 12, "ott_hit.query");

_Static_assert(sizeof(ott_query_desc) == 72, "ott_query_desc is 72 bytes on LP64");
_Static_assert(offsetof(ott_query_desc, queries) == 0, "desc.queries");
_Static_assert(offsetof(ott_query_desc, nq) == 8, "desc.nq");
_Static_assert(offsetof(ott_query_desc, metric) == 12, "desc.metric");
_Static_assert(offsetof(ott_query_desc, take) == 16, "desc.take");
_Static_assert(offsetof(ott_query_desc, filter_cmp) == 20, "desc.filter_cmp");
_Static_assert(offsetof(ott_query_desc, filter_thr) == 24, "desc.filter_thr");
_Static_assert(offsetof(ott_query_desc, mode) == 28, "desc.mode");
_Static_assert(offsetof(ott_query_desc, k) == 32, "desc.k");
_Static_assert(offsetof(ott_query_desc, chunk_mask) == 40, "desc.chunk_mask");
_Static_assert(offsetof(ott_query_desc, row_mask) == 48, "desc.row_mask");
_Static_assert(offsetof(ott_query_desc, row_mask_bits) == 56, "desc.row_mask_bits");
_Static_assert(offsetof(ott_query_desc, use_device_row_mask) == 64, "desc.use_device_row_mask");
_Static_assert(offsetof(ott_query_desc, path) == 68, "desc.path");

_Static_assert(sizeof(ott_stats) == 120, "ott_stats is 120 bytes");
_Static_assert(offsetof(ott_stats, total_chunks) == 0, "stats.total_chunks");
_Static_assert(offsetof(ott_stats, pruned_chunks) == 8, "stats.pruned_chunks");
_Static_assert(offsetof(ott_stats, evaluated_chunks) == 16, "stats.evaluated_chunks");
_Static_assert(offsetof(ott_stats, vectors_compared) == 24, "stats.vectors_compared");
_Static_assert(offsetof(ott_stats, prune_ns) == 32, "stats.prune_ns");
_Static_assert(offsetof(ott_stats, score_ns) == 40, "stats.score_ns");
_Static_assert(offsetof(ott_stats, merge_ns) == 48, "stats.merge_ns");
_Static_assert(offsetof(ott_stats, total_ns) == 56, "stats.total_ns");
_Static_assert(offsetof(ott_stats, bytes_scanned) == 64, "stats.bytes_scanned");
_Static_assert(offsetof(ott_stats, path_used) == 72, "stats.path_used");
_Static_assert(offsetof(ott_stats, passes) == 76, "stats.passes");
_Static_assert(offsetof(ott_stats, rescored) == 80, "stats.rescored");
_Static_assert(offsetof(ott_stats, retries) == 88, "stats.retries");
_Static_assert(offsetof(ott_stats, refined) == 92, "stats.refined");
_Static_assert(offsetof(ott_stats, err_ratio_max) == 96, "stats.err_ratio_max");
_Static_assert(offsetof(ott_stats, gate_failed) == 100, "stats.gate_failed");
_Static_assert(offsetof(ott_stats, bound_violations) == 104, "stats.bound_violations");
_Static_assert(offsetof(ott_stats, i8_refined) == 108, "stats.i8_refined");
_Static_assert(offsetof(ott_stats, exchange_ns) == 112, "stats.exchange_ns");

_Static_assert(sizeof(ott_leaf) == 32, "ott_leaf is 32 bytes");
_Static_assert(offsetof(ott_leaf, column) == 0, "leaf.column");
_Static_assert(offsetof(ott_leaf, op) == 4, "leaf.op");
_Static_assert(offsetof(ott_leaf, clause) == 8, "leaf.clause");
_Static_assert(offsetof(ott_leaf, lit_i64) == 16, "leaf.lit_i64");
_Static_assert(offsetof(ott_leaf, lit_f64) == 24, "leaf.lit_f64");

/* enum values the bindings hard-code (src/vec.rs:11-31, src/expr.rs:83-91, src/type_utils.rs:11-19) */
_Static_assert(OTT_METRIC_COSINE == 0 && OTT_METRIC_EUCLIDEAN == 1 && OTT_METRIC_DOT == 2, "ott_metric");
_Static_assert(OTT_TAKE_MIN == 0 && OTT_TAKE_MAX == 1, "ott_take");
_Static_assert(OTT_CMP_NONE == 0 && OTT_CMP_LT == 1 && OTT_CMP_GT == 2 && OTT_CMP_LTE == 3 && OTT_CMP_GTE == 4 && OTT_CMP_EQ == 5, "ott_cmp");
_Static_assert(OTT_OP_EQ == 0 && OTT_OP_NEQ == 1 && OTT_OP_LT == 2 && OTT_OP_LTE == 3 && OTT_OP_GT == 4 && OTT_OP_GTE == 5, "ott_op");
_Static_assert(OTT_DT_INT32 == 0 && OTT_DT_INT64 == 1 && OTT_DT_FLOAT32 == 2 && OTT_DT_FLOAT64 == 3 && OTT_DT_DATETIME == 5, "ott_dtype");
_Static_assert(OTT_MODE_MERGED == 0 && OTT_MODE_PER_QUERY == 1, "ott_mode");
_Static_assert(OTT_PATH_AUTO == 0 && OTT_PATH_EXACT == 1 && OTT_PATH_MFMA == 2, "ott_path");
_Static_assert(OTT_OK == 0 && OTT_ERR_INVALID == -1 && OTT_ERR_HIP == -2 && OTT_ERR_OOM == -3 && OTT_ERR_UNSUPPORTED == -4, "ott_status");
_Static_assert(OTT_COMM_ID_BYTES == 128, "RCCL unique id size");
_Static_assert(sizeof(ott_allgather_fn) == sizeof(void*), "callback is a plain function pointer");

#define FIELD(T, f) printf("    \"%s.%s\": %zu,\n", #T, #f, offsetof(T, f))

int main(void) {
    printf("{\n");
    printf("  \"abi_version_header\": %d,\n", OTT_ABI_VERSION);
    printf("  \"abi_version_library\": %d,\n", ott_abi_version());
    printf("  \"sizeof\": {\"ott_hit\": %zu, \"ott_query_desc\": %zu, \"ott_stats\": %zu, \"ott_leaf\": %zu},\n", sizeof(ott_hit),
           sizeof(ott_query_desc), sizeof(ott_stats), sizeof(ott_leaf));
    printf("  \"offsetof\": {\n");
    FIELD(ott_hit, index); FIELD(ott_hit, score); FIELD(ott_hit, query);
    FIELD(ott_query_desc, queries); FIELD(ott_query_desc, nq); FIELD(ott_query_desc, metric); FIELD(ott_query_desc, take);
    FIELD(ott_query_desc, filter_cmp); FIELD(ott_query_desc, filter_thr); FIELD(ott_query_desc, mode); FIELD(ott_query_desc, k);
    FIELD(ott_query_desc, chunk_mask); FIELD(ott_query_desc, row_mask); FIELD(ott_query_desc, row_mask_bits);
    FIELD(ott_query_desc, use_device_row_mask); FIELD(ott_query_desc, path);
    FIELD(ott_stats, total_chunks); FIELD(ott_stats, pruned_chunks); FIELD(ott_stats, evaluated_chunks); FIELD(ott_stats, vectors_compared);
    FIELD(ott_stats, prune_ns); FIELD(ott_stats, score_ns); FIELD(ott_stats, merge_ns); FIELD(ott_stats, total_ns);
    FIELD(ott_stats, bytes_scanned); FIELD(ott_stats, path_used); FIELD(ott_stats, passes); FIELD(ott_stats, rescored);
    FIELD(ott_stats, retries); FIELD(ott_stats, refined); FIELD(ott_stats, err_ratio_max); FIELD(ott_stats, gate_failed);
    FIELD(ott_stats, bound_violations); FIELD(ott_stats, i8_refined); FIELD(ott_stats, exchange_ns);
    FIELD(ott_leaf, column); FIELD(ott_leaf, op); FIELD(ott_leaf, clause); FIELD(ott_leaf, reserved); FIELD(ott_leaf, lit_i64);
    printf("    \"ott_leaf.lit_f64\": %zu\n  }\n}\n", offsetof(ott_leaf, lit_f64));
    return OTT_ABI_VERSION == ott_abi_version() ? 0 : 1;
}

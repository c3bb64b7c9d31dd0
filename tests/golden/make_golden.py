#!/usr/bin/env python3
"""Writes the golden fixtures under tests/golden/.

The reference (AtharvBhat/otters, Rust) cannot be built or imported here (no cargo/rustc),
so the fixtures are DATA transcribed from the known-answer cases the reference's own tests
and README hold for the hot path: the input literals and the expected outputs / properties
each test asserts, each tagged with the reference file:line it comes from.  No reference
source text is stored.  Run:  python tests/golden/make_golden.py
"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

T = "tests/vec_store_tests.rs"
STD5 = [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0], [1.0, 1.0, 0.0], [0.5, 0.5, 0.5]]  # :5-13


def case(name, ref, vectors, queries, metric, expect, dim=None, flt=None, take=None):
    if dim is None:
        dim = len(vectors[0]) if vectors else (len(queries[0]) if queries and isinstance(queries[0], list) else len(queries))
    return dict(name=name, ref=ref, dim=dim, vectors=vectors, queries=queries, metric=metric, filter=flt,
                take=take or [], expect=expect)


def f32list(a):
    return [float(np.float32(x)) for x in a]


vec_cases = [
    # ---- kernels: known answers -------------------------------------------------------------
    dict(name="kernel_dot", ref=f"{T}:506-515", kernel="dot", a=[1.0, 2.0, 3.0, 4.0], b=[2.0, 3.0, 4.0, 5.0], expect=40.0, exact=True),
    dict(name="kernel_l2sq", ref=f"{T}:518-527", kernel="l2sq", a=[1.0, 2.0], b=[4.0, 6.0], expect=25.0, exact=True),
    dict(name="kernel_cosine", ref=f"{T}:530-538", kernel="cosine", a=[1.0, 0.0], b=[1.0, 0.0], inv_a=1.0, inv_b=1.0, expect=1.0, tol=1e-6),
    # ---- plan behaviour -----------------------------------------------------------------------
    case("empty_store_single_query_ok", f"{T}:37-45", [], [1.0, 0.0, 0.0], "cosine", dict(len=0), dim=3),
    case("empty_store_multi_query_ok", f"{T}:37-45", [], [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]], "cosine", dict(len=0), dim=3),
    case("dim_mismatch_error", f"{T}:52-63", [[1.0, 0.0, 0.0]], [1.0, 0.0], "cosine",
         dict(error_contains="Query vector length 2 does not match expected dimension 3"), take=[["take", 5]]),
    case("empty_batch_error", f"{T}:66-76", [], [], "cosine", dict(error_eq="No queries provided"), dim=3, take=[["take", 5]]),
    case("error_through_chain", f"{T}:79-94", [], [1.0, 0.0], "cosine",
         dict(error_contains="Query vector length 2 does not match expected dimension 3"), dim=3,
         flt=[0.5, "gt"], take=[["take", 5], ["take_min", 3]]),
    case("valid_chain_filter_gt", f"{T}:97-119", [[1.0, 0.0], [0.8, 0.6], [0.0, 1.0]], [1.0, 0.0], "cosine",
         dict(all_scores=["gt", 0.5]), flt=[0.5, "gt"], take=[["take", 5]]),
    case("mixed_dim_batch_error", f"{T}:122-140", [[1.0, 0.0, 0.0]], [[1.0, 0.0, 0.0], [1.0, 0.0], [1.0, 0.0, 0.0]], "cosine",
         dict(error_contains="Query vector length 2 does not match expected dimension 3"), dim=3, take=[["take", 5]]),
    case("cosine_basic_self", f"{T}:147-162", STD5, [1.0, 0.0, 0.0], "cosine",
         dict(len=5, scores_by_index={"0": 1.0}, tol=1e-6), take=[["take", 5]]),
    case("cosine_orthogonal", f"{T}:165-183", [[1.0, 0.0], [0.0, 1.0]], [1.0, 0.0], "cosine",
         dict(len=2, scores_by_index={"0": 1.0, "1": 0.0}, tol=1e-6), take=[["take", 2]]),
    case("euclidean_basic_self", f"{T}:190-204", STD5, [1.0, 0.0, 0.0], "euclidean",
         dict(scores_by_index={"0": 0.0}, tol=1e-6), take=[["take_min", 5]]),
    case("dot_basic_self", f"{T}:211-225", STD5, [1.0, 0.0, 0.0], "dot",
         dict(scores_by_index={"0": 1.0}, tol=1e-6), take=[["take", 5]]),
    case("dot_orthogonal_scaled_opposite", f"{T}:228-252", [[1.0, 0.0], [0.0, 1.0], [2.0, 0.0], [-1.0, 0.0]], [1.0, 0.0], "dot",
         dict(len=4, scores_by_index={"0": 1.0, "1": 0.0, "2": 2.0, "3": -1.0}, tol=1e-6), take=[["take", 4]]),
    case("dot_ranking", f"{T}:255-277", [[3.0, 4.0], [1.0, 1.0], [0.0, 1.0], [-1.0, 0.0]], [3.0, 4.0], "dot",
         dict(len=4, sorted="desc", scores_in_order=[25.0, 7.0, 4.0, -3.0], indices_in_order=[0, 1, 2, 3], tol=1e-6),
         take=[["take", 4]]),
    case("dot_filter_gt_1", f"{T}:280-300", [[2.0, 0.0], [1.0, 0.0], [0.5, 0.0], [-1.0, 0.0]], [1.0, 0.0], "dot",
         dict(len=1, scores_in_order=[2.0], tol=1e-6), flt=[1.0, "gt"], take=[["take", 10]]),
    case("dot_take_max", f"{T}:303-323", [[1.0, 0.0], [2.0, 0.0], [0.5, 0.0], [-1.0, 0.0]], [1.0, 0.0], "dot",
         dict(len=2, scores_in_order=[2.0, 1.0], tol=1e-6), take=[["take_max", 2]]),
    case("dot_take_min", f"{T}:326-346", [[1.0, 0.0], [2.0, 0.0], [0.5, 0.0], [-1.0, 0.0]], [1.0, 0.0], "dot",
         dict(len=2, scores_in_order=[-1.0, 0.5], tol=1e-6), take=[["take_min", 2]]),
    case("dot_batch_take3", f"{T}:350-362", [[1.0, 0.0], [0.0, 1.0], [1.0, 1.0]], [[1.0, 0.0], [0.0, 1.0]], "dot",
         dict(len=3), take=[["take", 3]]),
    case("topk_cosine", f"{T}:369-391", [[1.0, 0.0], [0.8, 0.6], [0.0, 1.0], [-1.0, 0.0]], [1.0, 0.0], "cosine",
         dict(len=2, sorted="desc"), take=[["take", 2]]),
    case("topk_euclidean", f"{T}:394-416", [[1.0, 0.0], [1.1, 0.0], [0.0, 1.0], [-1.0, 0.0]], [1.0, 0.0], "euclidean",
         dict(len=2, sorted="asc"), take=[["take_min", 2]]),
    case("take_more_than_available", f"{T}:419-435", [[1.0, 0.0], [0.0, 1.0]], [1.0, 0.0], "cosine", dict(len=2), take=[["take", 10]]),
    case("take_zero", f"{T}:438-452", [[1.0, 0.0], [0.0, 1.0]], [1.0, 0.0], "cosine", dict(len=0), take=[["take", 0]]),
    case("filtering_gt_05", f"{T}:459-482", [[1.0, 0.0], [0.8, 0.6], [0.0, 1.0], [-1.0, 0.0]], [1.0, 0.0], "cosine",
         dict(all_scores=["gt", 0.5]), flt=[0.5, "gt"], take=[["take", 10]]),
    case("empty_store_take5", f"{T}:496-506", [], [1.0, 0.0, 0.0], "cosine", dict(len=0), dim=3, take=[["take", 5]]),
    # ---- mathematical correctness -------------------------------------------------------------
    case("cosine_correctness", f"{T}:545-608", [[1.0, 0.0], [-1.0, 0.0], [0.0, 1.0], [1.0, 1.0]], [1.0, 0.0], "cosine",
         dict(len=4, scores_by_index={"0": 1.0, "1": -1.0, "2": 0.0}, tol=1e-6,
              scores_by_index_loose={"3": 0.7071067811865475}, tol_loose=1e-5), take=[["take", 4]]),
    case("euclidean_correctness", f"{T}:611-656", [[0.0, 0.0], [3.0, 4.0], [1.0, 1.0], [0.0, 5.0], [-3.0, -4.0]], [0.0, 0.0], "euclidean",
         dict(len=5, scores_by_index={"0": 0.0, "1": 25.0, "2": 2.0, "3": 25.0, "4": 25.0}, tol=1e-6), take=[["take_min", 5]]),
    case("dot_correctness", f"{T}:659-745",
         [[2.0, 3.0, 1.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [1.0, 1.0, 1.0]], [2.0, 3.0, 1.0], "dot",
         dict(len=6, sorted="desc", scores_by_index={"0": 14.0, "1": 2.0, "2": 3.0, "3": 1.0, "4": -2.0, "5": 6.0}, tol=1e-6),
         take=[["take", 6]]),
    case("topk_ranking_cosine", f"{T}:748-798", [[1.0, 0.0], [0.8, 0.6], [0.6, 0.8], [0.0, 1.0]], [1.0, 0.0], "cosine",
         dict(len=4, sorted="desc", scores_in_order=[1.0, 0.8, 0.6, 0.0], tol=1e-6), take=[["take", 4]]),
    case("euclidean_ranking", f"{T}:801-851", [[0.0, 0.0], [1.0, 0.0], [0.0, 1.0], [1.0, 1.0], [2.0, 0.0], [3.0, 4.0]], [0.0, 0.0], "euclidean",
         dict(len=6, sorted="asc", scores_in_order=[0.0, 1.0, 1.0, 2.0, 4.0, 25.0], tol=1e-6), take=[["take_min", 6]]),
    case("filter_threshold_gt_07", f"{T}:854-871", [[1.0, 0.0], [0.8, 0.6], [0.6, 0.8], [0.0, 1.0], [-0.6, 0.8]], [1.0, 0.0], "cosine",
         dict(all_scores=["gt", 0.7]), flt=[0.7, "gt"], take=[["take", 10]]),
    case("filter_threshold_gte_06", f"{T}:873-882", [[1.0, 0.0], [0.8, 0.6], [0.6, 0.8], [0.0, 1.0], [-0.6, 0.8]], [1.0, 0.0], "cosine",
         dict(all_scores=["gte", 0.6]), flt=[0.6, "gte"], take=[["take", 10]]),
    case("filter_threshold_lt_05", f"{T}:884-895", [[1.0, 0.0], [0.8, 0.6], [0.6, 0.8], [0.0, 1.0], [-0.6, 0.8]], [1.0, 0.0], "cosine",
         dict(all_scores=["lt", 0.5]), flt=[0.5, "lt"], take=[["take", 10]]),
    case("batch_query_two_perfect", f"{T}:899-924", [[1.0, 0.0], [0.0, 1.0], [-1.0, 0.0]], [[1.0, 0.0], [0.0, 1.0]], "cosine",
         dict(count_score=[1.0, 2], tol=1e-6), take=[["take", 2]]),
    case("api_showcase_100rows", f"{T}:931-958",
         [f32list([np.float32(i) / np.float32(100.0), np.float32(i * 2) / np.float32(100.0), np.float32(i * 3) / np.float32(100.0)]) for i in range(100)],
         [0.5, 0.5, 0.5], "cosine", dict(all_scores=["gt", 0.8]), flt=[0.8, "gt"], take=[["take_min", 10]]),
    case("error_in_chain_stops", f"{T}:961-981", [], [1.0, 0.0], "cosine",
         dict(error_contains="Query vector length 2 does not match expected dimension 3"), dim=3,
         flt=[0.5, "gt"], take=[["take", 10], ["take_min", 5]]),
    dict(name="plan_new_unset", ref=f"{T}:988-997", plan_new=True, expect=dict(error_contains="Query vectors or their norms are not set")),
    case("empty_query_vectors_in_batch", f"{T}:1022-1029", [], [], "cosine", dict(error_contains="No queries provided"), dim=3),
    case("filter_lt_09", f"{T}:1045-1052", [[1.0, 0.0], [0.0, 1.0], [0.5, 0.5], [0.8, 0.6]], [1.0, 0.0], "cosine",
         dict(nonempty=True, all_scores=["lt", 0.9]), flt=[0.9, "lt"], take=[["take", 10]]),
    case("filter_gt_01", f"{T}:1054-1061", [[1.0, 0.0], [0.0, 1.0], [0.5, 0.5], [0.8, 0.6]], [1.0, 0.0], "cosine",
         dict(nonempty=True, all_scores=["gt", 0.1]), flt=[0.1, "gt"], take=[["take", 10]]),
    case("filter_lte_10", f"{T}:1063-1070", [[1.0, 0.0], [0.0, 1.0], [0.5, 0.5], [0.8, 0.6]], [1.0, 0.0], "cosine",
         dict(nonempty=True, all_scores=["lte", 1.0]), flt=[1.0, "lte"], take=[["take", 10]]),
    case("filter_gte_00", f"{T}:1072-1079", [[1.0, 0.0], [0.0, 1.0], [0.5, 0.5], [0.8, 0.6]], [1.0, 0.0], "cosine",
         dict(nonempty=True, all_scores=["gte", 0.0]), flt=[0.0, "gte"], take=[["take", 10]]),
    case("filter_eq_10", f"{T}:1081-1089", [[1.0, 0.0], [0.0, 1.0], [0.5, 0.5], [0.8, 0.6]], [1.0, 0.0], "cosine",
         dict(nonempty=True, all_scores=["eq", 1.0]), flt=[1.0, "eq"], take=[["take", 10]]),
    case("zero_norm_store_vector", f"{T}:1093-1110", [[0.0, 0.0, 0.0]], [1.0, 0.0, 0.0], "cosine", dict(ok=True), take=[["take", 1]]),
    case("zero_norm_query_vector", f"{T}:1113-1126", [[1.0, 0.0, 0.0]], [0.0, 0.0, 0.0], "cosine", dict(ok=True), take=[["take", 1]]),
    case("no_filter_take2", f"{T}:1131-1145", [[1.0, 0.0], [0.0, 1.0], [0.5, 0.5]], [1.0, 0.0], "cosine", dict(len=2), take=[["take", 2]]),
    dict(name="add_vectors_dim_mismatch", ref=f"{T}:1148-1164", add_vectors=dict(dim=3, vectors=[[1.0, 0.0, 0.0], [1.0, 0.0]]),
         expect=dict(error_contains="Input vector length 2 does not match expected dimension 3")),
    case("take_min_euclid", f"{T}:1167-1186", [[1.0, 0.0], [0.0, 1.0], [0.9, 0.1]], [1.0, 0.0], "euclidean", dict(len=2), take=[["take_min", 2]]),
    case("take_max_euclid", f"{T}:1188-1194", [[1.0, 0.0], [0.0, 1.0], [0.9, 0.1]], [1.0, 0.0], "euclidean", dict(len=2), take=[["take_max", 2]]),
    case("batch_take_min_1", f"{T}:1198-1205", [[1.0, 0.0], [0.0, 1.0], [0.9, 0.1]], [[1.0, 0.0], [0.0, 1.0]], "euclidean", dict(len=1), take=[["take_min", 1]]),
    case("batch_take_max_1", f"{T}:1206-1212", [[1.0, 0.0], [0.0, 1.0], [0.9, 0.1]], [[1.0, 0.0], [0.0, 1.0]], "euclidean", dict(len=1), take=[["take_max", 1]]),
    case("query_batch_single", f"{T}:1216-1227", [[1.0, 0.0, 0.0]], [1.0, 0.0, 0.0], "cosine", dict(len=1), take=[["take", 1]]),
    case("query_batch_multi", f"{T}:1229-1236", [[1.0, 0.0, 0.0]], [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]], "cosine", dict(len_le=2), take=[["take", 2]]),
    case("error_states_chained", f"{T}:1239-1257", [[1.0, 0.0, 0.0]], [1.0, 0.0], "cosine",
         dict(error_contains="does not match expected dimension"), flt=[0.5, "gt"],
         take=[["take", 5], ["take_min", 2], ["take_max", 1]]),
    case("filter_gt_15_empty", f"{T}:1260-1277", [[1.0, 0.0], [0.0, 1.0], [-1.0, 0.0]], [1.0, 0.0], "cosine", dict(len=0),
         flt=[1.5, "gt"], take=[["take", 10]]),
    case("filter_eq_1_single", f"{T}:1279-1286", [[1.0, 0.0], [0.0, 1.0], [-1.0, 0.0]], [1.0, 0.0], "cosine", dict(len=1),
         flt=[1.0, "eq"], take=[["take", 10]]),
]

# --------------------------------------------------------------------------------------------
# MetaStore cases: tests/meta_tests.rs, tests/meta_zonemap_tests.rs, README.md example
# expr JSON: ["cmp", col, op, literal] | ["and", a, b] | ["or", a, b]
# --------------------------------------------------------------------------------------------
M = "tests/meta_tests.rs"
Z = "tests/meta_zonemap_tests.rs"

zone_cols = [  # meta_zonemap_tests.rs:17-67 (build_store)
    dict(name="val", dtype="Int32", values=[1, 2, None, 10, 11, 12, None, None, None]),
    dict(name="ts", dtype="DateTime", values=["2024-01-01T00:00:00Z", None, "2024-06-01T00:00:00Z", "2026-01-01T00:00:00Z",
                                              "2026-06-01T00:00:00Z", "2024-12-31T23:59:59Z", None, None, None]),
    dict(name="grade", dtype="String", values=["A", "B", None, "C", "A", "A", None, None, None]),
]
zone_vecs = [[1.0, 0.0] for _ in range(9)]


def mcase(name, ref, vectors, columns, chunk_size, queries, metric, expect, meta_filter=None, vec_filter=None, take=None):
    return dict(name=name, ref=ref, vectors=vectors, columns=columns, chunk_size=chunk_size, queries=queries, metric=metric,
                meta_filter=meta_filter, vec_filter=vec_filter, take=take, expect=expect)


meta_cases = [
    mcase("meta_basic_pruning_and_stats", f"{M}:5-43",
          [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.5, 0.5, 0.0], [0.0, 0.0, 1.0]],
          [dict(name="age", dtype="Int32", values=[10, 20, 30, None]), dict(name="grade", dtype="String", values=["A", "B", "A", "C"])],
          2, [1.0, 0.0, 0.0], "cosine", dict(index_set=[2], stats=dict(total_chunks=2, evaluated_chunks_ge=1)),
          meta_filter=["and", ["cmp", "age", "gt", 15], ["cmp", "grade", "eq", "A"]], take=4),
    mcase("meta_string_eq_prunes_chunks", f"{M}:46-93",
          [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [1.0, 1.0, 0.0], [0.0, 0.0, 1.0], [1.0, 0.0, 1.0], [0.5, 0.5, 0.0]],
          [dict(name="age", dtype="Int32", values=[10, 11, 12, 20, 21, 22]),
           dict(name="grade", dtype="String", values=["B", "C", "B+", "A", "A", "C"])],
          3, [1.0, 0.0, 0.0], "cosine", dict(stats=dict(total_chunks=2, pruned_chunks_ge=1), index_set=[3, 4]),
          meta_filter=["cmp", "grade", "eq", "A"], take=6),
    mcase("meta_datetime_range_filter", f"{M}:96-124", [[1.0, 0.0], [0.0, 1.0], [1.0, 1.0]],
          [dict(name="ts", dtype="DateTime", values=["2023-01-01T00:00:00Z", "2023-06-01T00:00:00Z", "2024-01-01T00:00:00Z"])],
          2, [1.0, 0.0], "dot", dict(index_set=[0, 1]),
          meta_filter=["and", ["cmp", "ts", "gte", "2023-01-01T00:00:00Z"], ["cmp", "ts", "lt", "2024-01-01T00:00:00Z"]], take=3),
    mcase("meta_global_scope_merge_and_vec_threshold", f"{M}:127-158", [[1.0, 0.0], [0.0, 1.0], [1.0, 1.0], [2.0, 0.0]],
          [dict(name="grade", dtype="String", values=["A", "B", "A", "A"])],
          2, [[1.0, 0.0], [0.0, 1.0]], "dot",
          dict(len_le=2, stats=dict(evaluated_le_total=True), scores_in_order=[2.0, 1.0], tol=1e-6),
          meta_filter=["cmp", "grade", "eq", "A"], vec_filter=[0.5, "gt"], take=2),
    dict(name="meta_build_mismatched_column_len_errors", ref=f"{M}:161-171", build_error=True, vectors=[[1.0], [2.0]],
         columns=[dict(name="age", dtype="Int32", values=[1])], chunk_size=2),
    mcase("meta_stats_without_meta_filter", f"{M}:174-190", [[1.0, 0.0], [0.0, 1.0], [1.0, 1.0]], [], 2, [1.0, 0.0], "cosine",
          dict(len=3, stats=dict(vectors_compared=3, total_chunks=2, evaluated_chunks=2, pruned_chunks=0)), take=3),
    mcase("zonemap_prunes_numeric_with_nulls", f"{Z}:70-89", zone_vecs, zone_cols, 3, [1.0, 0.0], "dot",
          dict(index_set=[3, 4, 5], stats=dict(total_chunks=3, evaluated_chunks=1, pruned_chunks=2)),
          meta_filter=["cmp", "val", "gt", 5], take=9),
    mcase("zonemap_boundary_gte2", f"{Z}:92-104", zone_vecs, zone_cols, 3, [1.0, 0.0], "cosine",
          dict(stats=dict(total_chunks=3, pruned_chunks=1), index_set=[1, 3, 4, 5]),
          meta_filter=["cmp", "val", "gte", 2], take=9),
    mcase("zonemap_boundary_gt2", f"{Z}:106-116", zone_vecs, zone_cols, 3, [1.0, 0.0], "cosine",
          dict(stats=dict(evaluated_chunks=1, pruned_chunks=2), index_set=[3, 4, 5]),
          meta_filter=["cmp", "val", "gt", 2], take=9),
    mcase("zonemap_all_null_chunk_pruned_for_equality", f"{Z}:119-131", zone_vecs, zone_cols, 3, [1.0, 0.0], "cosine",
          dict(stats=dict(total_chunks=3, pruned_chunks_ge=1), index_set=[0, 4, 5]),
          meta_filter=["cmp", "grade", "eq", "A"], take=9),
    mcase("zonemap_and_clause_numeric_datetime", f"{Z}:134-156", zone_vecs, zone_cols, 3, [1.0, 0.0], "dot",
          dict(len=1, indices_in_order=[5], stats=dict(total_chunks=3, evaluated_chunks=1, pruned_chunks=2)),
          meta_filter=["and", ["cmp", "val", "gt", 5], ["cmp", "ts", "lt", "2025-01-01T00:00:00Z"]], take=9),
    mcase("zonemap_ne_comparator_with_null_only_chunk", f"{Z}:159-174", zone_vecs, zone_cols, 3, [1.0, 0.0], "cosine",
          dict(stats=dict(total_chunks=3, pruned_chunks_ge=1), index_set=[1, 3, 4, 5]),
          meta_filter=["cmp", "val", "neq", 1], take=9),
    # README.md:63-113 -> sample output :129-150.  Scores recomputed in f32 following the
    # dim<8 scalar path: 0.9701425 (0x3F785B42) and 0.70710677 (0x3F3504F3) twice (an exact tie).
    mcase("readme_example_8x4", "README.md:63-150",
          [[1.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0], [1.0, 1.0, 0.0, 0.0], [0.0, 0.0, 1.0, 0.0],
           [0.8, 0.2, 0.0, 0.0], [0.0, 0.0, 0.0, 1.0], [0.6, 0.6, 0.0, 0.0], [0.0, 0.5, 0.5, 0.0]],
          [dict(name="name", dtype="String", values=["widget", "gizmo", "adapter", "battery", "charger", "cable", "dock", "earbuds"]),
           dict(name="price", dtype="Float64", values=[19.99, 49.00, 12.50, 8.99, 29.99, 5.99, 39.50, 59.99]),
           dict(name="mfg", dtype="DateTime", values=["2024-01-05", "2024-01-10", "2024-02-15", "2024-03-01", "2024-03-20", "2024-04-05", "2024-05-01", "2024-05-12"]),
           dict(name="exp", dtype="DateTime", values=["2025-01-05", "2024-12-31", "2024-10-01", "2024-06-01", "2025-06-01", "2024-08-01", "2025-01-01", "2024-12-01"]),
           dict(name="version", dtype="Int32", values=[1, 2, 2, 1, 3, 1, 2, 3])],
          4, [1.0, 0.0, 0.0, 0.0], "cosine",
          dict(len=3, index_set=[4, 2, 6], indices_in_order_ties=[[4], [2, 6]], scores_in_order=[0.970142, 0.707107, 0.707107], tol=5e-7,
               score_bits_in_order=["0x3F785B42", "0x3F3504F3", "0x3F3504F3"],
               stats=dict(total_chunks=2, pruned_chunks=0, evaluated_chunks=2, vectors_compared=8)),
          meta_filter=["and", ["and", ["and", ["cmp", "price", "lte", 40.0], ["cmp", "version", "gte", 2]],
                               ["cmp", "mfg", "gte", "2024-01-01"]], ["cmp", "exp", "gte", "2024-06-01"]], take=5),
    # examples/demo.rs run as `demo 8 4` (BASELINE config 0): one 8-row chunk with the
    # "even chunk" metadata (demo.rs:36-77) that the demo's own filter (demo.rs:107-110) prunes.
    # Vectors are unseeded random in the demo; any vectors give 0 hits / pruned_chunks = 1.
    mcase("demo_8_4_plumbing", "examples/demo.rs:15-113",
          [[0.25, -0.5, 0.75, 0.125], [0.5, 0.5, -0.5, 0.25], [-0.75, 0.125, 0.25, 0.5], [0.125, 0.25, 0.5, -0.75],
           [0.9, -0.1, 0.2, 0.3], [-0.3, 0.6, 0.1, -0.2], [0.4, 0.4, 0.4, 0.4], [-0.5, -0.25, 0.75, 0.0]],
          [dict(name="name", dtype="String", values=[f"item_{i}" for i in range(8)]),
           dict(name="price", dtype="Float64", values=[80.0 + (i % 20) for i in range(8)]),
           dict(name="mfg", dtype="DateTime", values=["2024-01-01"] * 8),
           dict(name="exp", dtype="DateTime", values=["2024-12-31"] * 8),
           dict(name="version", dtype="Int32", values=[1] * 8)],
          128, [0.3, -0.2, 0.5, 0.1], "cosine",
          dict(len=0, stats=dict(total_chunks=1, pruned_chunks=1, evaluated_chunks=0, vectors_compared=0)),
          meta_filter=["and", ["and", ["cmp", "price", "lt", 50.0], ["cmp", "version", "gte", 2]], ["cmp", "exp", "gte", "2025-01-01"]],
          vec_filter=[0.1, "gt"], take=5),
]

# bit-order pins from tests/simd_types_tests.rs (lane j <-> bit j) and tests/column_tests.rs
# (BitVec Lsb0): a handful of known answers for the mask helpers.
mask_cases = [
    dict(name="i64x8_cmp_lane_bit_order", ref="tests/simd_types_tests.rs (lane j <-> bit j)", kind="i64",
         vals=[1, 2, 3, 4, 5, 6, 7, 8], op="gt", thr=4, expect_bits=[0, 0, 0, 0, 1, 1, 1, 1]),
    dict(name="f64x8_cmp_le", ref="tests/simd_types_tests.rs", kind="f64",
         vals=[1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0], op="lte", thr=3.0, expect_bits=[1, 1, 1, 0, 0, 0, 0, 0]),
    dict(name="i32_neq_with_null", ref="src/type_utils.rs:368-385", kind="i32",
         vals=[1, 2, -2147483648, 1, 5, 6, 7, 8, 9], nulls=[0, 0, 1, 0, 0, 0, 0, 0, 0], op="neq", thr=1,
         expect_bits=[0, 1, 0, 0, 1, 1, 1, 1, 1]),
]

# tests/simd_types_tests.rs compares two 8-lane vectors lane by lane and asserts on bits of the mask (lane j <-> bit j).
# On the path the same compares run as value-vs-literal (type_utils.rs numeric_simd_mask) and as per-chunk min / max
# (zone stats), so each lane pair (a[j], b[j]) is a known answer for one row compare, and (min, max) of the pair for a
# 2-row zone.  `set` / `clear` are exactly the bits the reference test asserts; lanes it leaves open are not pinned.
_A18 = [1, 2, 3, 4, 5, 6, 7, 8]
_DESC = [5, 4, 3, 2, 1, 0, -1, -2]
lane_pair_cases = []
for kind, conv, mod in (("i64", int, "i64x8_tests"), ("f64", float, "f64x8_tests")):
    for name, a, b, op, bits_set, bits_clear in (
            ("cmp_eq", _A18, [1, 2, 3, 4, 9, 10, 11, 12], "eq", 0x0F, 0xF0),
            ("cmp_gt", _DESC, _A18, "gt", 0x03, 0x00),
            ("cmp_gte", _DESC, [5, 3, 3, 3, 1, 1, 0, 0], "gte", 0b00010111, 0x00),
            ("cmp_lt", _A18, _DESC, "lt", 0x03, 0x00),
            ("cmp_lte", [1, 3, 3, 4, 1, 0, -1, -2], [5, 3, 3, 2, 1, 0, 0, 0], "lte", 0b00110111, 0x00),
            ("from_slice_vs_splat", _A18, [1] * 8, "eq", 0x01, 0xFE if kind == "f64" else 0x00)):
        lane_pair_cases.append(dict(name=f"{kind}x8_{name}", ref=f"tests/simd_types_tests.rs {mod}::test_{kind}x8_{name}", kind=kind,
                                    a=[conv(v) for v in a], b=[conv(v) for v in b], op=op, set=bits_set, clear=bits_clear))
    lane_pair_cases.append(dict(name=f"{kind}x8_min_max", ref=f"tests/simd_types_tests.rs {mod}::test_{kind}x8_min / _max", kind=kind,
                                a=[conv(v) for v in [5, 2, 7, 1, 9, 3, 8, 4]], b=[conv(v) for v in [3, 6, 4, 8, 2, 7, 1, 9]],
                                min=[conv(v) for v in [3, 2, 4, 1, 2, 3, 1, 4]], max=[conv(v) for v in [5, 6, 7, 8, 9, 7, 8, 9]]))


def main():
    with open(os.path.join(HERE, "vec_store_cases.json"), "w") as f:
        json.dump(vec_cases, f, indent=1)
    with open(os.path.join(HERE, "meta_cases.json"), "w") as f:
        json.dump(meta_cases, f, indent=1)
    with open(os.path.join(HERE, "mask_cases.json"), "w") as f:
        json.dump(mask_cases, f, indent=1)
    with open(os.path.join(HERE, "lane_pair_cases.json"), "w") as f:
        json.dump(lane_pair_cases, f, indent=1)
    print(f"wrote {len(vec_cases)} vec cases, {len(meta_cases)} meta cases, {len(mask_cases)} mask cases, {len(lane_pair_cases)} lane-pair cases")


if __name__ == "__main__":
    main()

// C++ host-mirror test: the reference's MetaStore tests (tests/meta_tests.rs, tests/meta_zonemap_tests.rs)
// and the README example, re-stated against include/otters_meta.hpp.
#include <cstdio>
#include <set>

#include "otters_meta.hpp"

using namespace otters;

static int failures = 0;
#define CHECK(cond)                                                     \
    do {                                                                \
        if (!(cond)) {                                                  \
            std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            failures++;                                                 \
        }                                                               \
    } while (0)

static std::set<std::size_t> as_set(const std::vector<std::size_t>& v) { return std::set<std::size_t>(v.begin(), v.end()); }

static MetaStore build_zone_store() {  // meta_zonemap_tests.rs:17-67
    std::vector<std::vector<float>> vectors(9, {1.0f, 0.0f});
    auto val = Column("val", DataType::Int32).from({1, 2, null, 10, 11, 12, null, null, null});
    auto ts = Column("ts", DataType::DateTime).from({"2024-01-01T00:00:00Z", null, "2024-06-01T00:00:00Z", "2026-01-01T00:00:00Z",
                                                     "2026-06-01T00:00:00Z", "2024-12-31T23:59:59Z", null, null, null});
    auto grade = Column("grade", DataType::String).from({"A", "B", null, "C", "A", "A", null, null, null});
    return MetaStore::from_columns({val, ts, grade}).with_vectors(vectors).with_chunk_size(3).build();
}

int main() {
    {   // meta_basic_pruning_and_stats, meta_tests.rs:5-43
        auto age = Column("age", DataType::Int32).from({10, 20, 30, null});
        auto grade = Column("grade", DataType::String).from({"A", "B", "A", "C"});
        auto meta = MetaStore::from_columns({age, grade}).with_vectors({{1, 0, 0}, {0, 1, 0}, {0.5f, 0.5f, 0}, {0, 0, 1}}).with_chunk_size(2).build();
        auto res = meta.query({1, 0, 0}, Metric::Cosine).meta_filter(col("age").gt(15).and_(col("grade").eq("A"))).take(4).collect();
        CHECK(as_set(res.indices) == std::set<std::size_t>{2});
        CHECK(meta.last_query_stats()->total_chunks == 2 && meta.last_query_stats()->evaluated_chunks >= 1);
        CHECK(std::get<std::string>(res.column("grade")->get(0).v) == "A");
    }
    {   // meta_datetime_range_filter, meta_tests.rs:96-124
        auto ts = Column("ts", DataType::DateTime).from({"2023-01-01T00:00:00Z", "2023-06-01T00:00:00Z", "2024-01-01T00:00:00Z"});
        auto meta = MetaStore::from_columns({ts}).with_vectors({{1, 0}, {0, 1}, {1, 1}}).with_chunk_size(2).build();
        auto res = meta.query({1, 0}, Metric::DotProduct)
                       .meta_filter(col("ts").gte("2023-01-01T00:00:00Z") & col("ts").lt("2024-01-01T00:00:00Z")).take(3).collect();
        CHECK(as_set(res.indices) == (std::set<std::size_t>{0, 1}));
    }
    {   // meta_global_scope_merge_and_vec_threshold, meta_tests.rs:127-158
        auto grade = Column("grade", DataType::String).from({"A", "B", "A", "A"});
        auto meta = MetaStore::from_columns({grade}).with_vectors({{1, 0}, {0, 1}, {1, 1}, {2, 0}}).with_chunk_size(2).build();
        auto res = meta.query_batch({{1, 0}, {0, 1}}, Metric::DotProduct).meta_filter(col("grade").eq("A")).vec_filter(0.5f, Cmp::Gt).take(2).collect();
        CHECK(res.len() == 2 && res.scores[0] == 2.0f && res.scores[1] == 1.0f);
    }
    {   // meta_build_mismatched_column_len_errors, meta_tests.rs:161-171 ; deferred compile error (CHANGELOG 0.1.0-alpha2)
        bool threw = false;
        try {
            MetaStore::from_columns({Column("age", DataType::Int32).from({1})}).with_vectors({{1}, {2}}).with_chunk_size(2).build();
        } catch (const Error&) { threw = true; }
        CHECK(threw);
        auto meta = MetaStore::from_columns({Column("age", DataType::Int32).from({1, 2})}).with_vectors({{1}, {2}}).build();
        auto plan = meta.query({1}, Metric::Cosine).meta_filter(col("nope").gt(1)).take(1);
        std::string msg;
        try { plan.collect(); } catch (const Error& e) { msg = e.what(); }
        CHECK(msg == "meta_filter compile error: Unknown column 'nope'");
    }
    {   // zonemap tests
        auto store = build_zone_store();
        auto r = store.query({1, 0}, Metric::DotProduct).meta_filter(col("val").gt(5)).take(9).collect();
        CHECK(as_set(r.indices) == (std::set<std::size_t>{3, 4, 5}));
        CHECK(store.last_query_stats()->total_chunks == 3 && store.last_query_stats()->evaluated_chunks == 1 && store.last_query_stats()->pruned_chunks == 2);
        store.query({1, 0}, Metric::Cosine).meta_filter(col("val").gte(2)).take(9).collect();
        CHECK(store.last_query_stats()->pruned_chunks == 1);
        store.query({1, 0}, Metric::Cosine).meta_filter(col("val").gt(2)).take(9).collect();
        CHECK(store.last_query_stats()->evaluated_chunks == 1 && store.last_query_stats()->pruned_chunks == 2);
        store.query({1, 0}, Metric::Cosine).meta_filter(col("grade").eq("A")).take(9).collect();
        CHECK(store.last_query_stats()->pruned_chunks >= 1);
        r = store.query({1, 0}, Metric::DotProduct).meta_filter(col("val").gt(5).and_(col("ts").lt("2025-01-01T00:00:00Z"))).take(9).collect();
        CHECK(r.len() == 1 && r.indices[0] == 5 && store.last_query_stats()->evaluated_chunks == 1);
        store.query({1, 0}, Metric::Cosine).meta_filter(col("val").neq(1)).take(9).collect();
        CHECK(store.last_query_stats()->pruned_chunks >= 1);
    }
    {   // README.md:63-150
        auto names = Column("name", DataType::String).from({"widget", "gizmo", "adapter", "battery", "charger", "cable", "dock", "earbuds"});
        auto prices = Column("price", DataType::Float64).from({19.99, 49.00, 12.50, 8.99, 29.99, 5.99, 39.50, 59.99});
        auto mfg = Column("mfg", DataType::DateTime).from({"2024-01-05", "2024-01-10", "2024-02-15", "2024-03-01", "2024-03-20", "2024-04-05", "2024-05-01", "2024-05-12"});
        auto exp = Column("exp", DataType::DateTime).from({"2025-01-05", "2024-12-31", "2024-10-01", "2024-06-01", "2025-06-01", "2024-08-01", "2025-01-01", "2024-12-01"});
        auto version = Column("version", DataType::Int32).from({1, 2, 2, 1, 3, 1, 2, 3});
        auto meta = MetaStore::from_columns({names, prices, mfg, exp, version})
                        .with_vectors({{1, 0, 0, 0}, {0, 1, 0, 0}, {1, 1, 0, 0}, {0, 0, 1, 0}, {0.8f, 0.2f, 0, 0}, {0, 0, 0, 1}, {0.6f, 0.6f, 0, 0}, {0, 0.5f, 0.5f, 0}})
                        .with_chunk_size(4).build();
        auto res = meta.query({1, 0, 0, 0}, Metric::Cosine)
                       .meta_filter(col("price").lte(40.0) & col("version").gte(2) & col("mfg").gte("2024-01-01") & col("exp").gte("2024-06-01"))
                       .take(5).collect();
        CHECK(res.len() == 3 && res.indices[0] == 4 && as_set(res.indices) == (std::set<std::size_t>{2, 4, 6}));
        uint32_t b0, b1;
        std::memcpy(&b0, &res.scores[0], 4);
        std::memcpy(&b1, &res.scores[1], 4);
        CHECK(b0 == 0x3F785B42u && b1 == 0x3F3504F3u && res.scores[1] == res.scores[2]);
        auto st = *meta.last_query_stats();
        CHECK(st.total_chunks == 2 && st.pruned_chunks == 0 && st.evaluated_chunks == 2 && st.vectors_compared == 8);
        CHECK(res.columns.size() == 5 && std::get<std::string>(res.column("name")->get(0).v) == "charger");
        // display.rs at this revision: the README's tables below their title lines (tests/golden/readme_output.txt holds the same text)
        CHECK(res.to_string() ==
              "+-------+----------+-------------------------+-------------------------+---------+---------+---------+\n"
              "| index | score    | exp                     | mfg                     | name    | price   | version |\n"
              "+-------+----------+-------------------------+-------------------------+---------+---------+---------+\n"
              "| 4     | 0.970142 | 2025-06-01 00:00:00 UTC | 2024-03-20 00:00:00 UTC | charger | 29.9900 | 3       |\n"
              "| 2     | 0.707107 | 2024-10-01 00:00:00 UTC | 2024-02-15 00:00:00 UTC | adapter | 12.5000 | 2       |\n"
              "| 6     | 0.707107 | 2025-01-01 00:00:00 UTC | 2024-05-01 00:00:00 UTC | dock    | 39.5000 | 2       |\n"
              "+-------+----------+-------------------------+-------------------------+---------+---------+---------+");
        const std::string head = meta.head_n(5);
        CHECK(head.rfind("MetaStore \xE2\x80\xA2 rows=8 \xE2\x80\xA2 chunks=2 \xE2\x80\xA2 chunk_size=4\n+-------+", 0) == 0);
        CHECK(head.find("| 1     | 2024-12-31 00:00:00 UTC | 2024-01-10 00:00:00 UTC | gizmo   | 49.0000 | 2       |") != std::string::npos);
        CHECK(meta.build_stats() && meta.build_stats()->n_rows == 8 && meta.build_stats()->dim == 4 && meta.build_stats()->n_chunks == 2);
        CHECK(meta.build_stats()->format().rfind("MetaStore Build Stats\n+------------------+", 0) == 0);
        CHECK(st.format().rfind("Last Meta Query Stats\n+------------------+", 0) == 0 && st.format().find("| vectors_compared | 8 ") != std::string::npos);
        meta.print_last_stats();
    }
    std::printf(failures ? "FAILED (%d)\n" : "ALL PASSED\n", failures);
    return failures ? 1 : 0;
}

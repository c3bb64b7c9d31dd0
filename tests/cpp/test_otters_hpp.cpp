// C++ host-mirror test: a handful of the reference's VecStore tests (tests/vec_store_tests.rs)
// re-stated against include/otters.hpp, i.e. through the C ABI from compiled host code.
#include <cmath>
#include <cstdio>
#include <string>

#include "otters.hpp"

using namespace otters;

static int failures = 0;
#define CHECK(cond)                                                     \
    do {                                                                \
        if (!(cond)) {                                                  \
            std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            failures++;                                                 \
        }                                                               \
    } while (0)

template <typename F>
static std::string error_of(F&& f) {
    try {
        f();
    } catch (const Error& e) {
        return e.what();
    }
    return "";
}

int main() {
    {   // test_dimension_mismatch_error_handling, vec_store_tests.rs:52-63
        VecStore store(3);
        store.add_vector({1.f, 0.f, 0.f});
        auto msg = error_of([&] { store.query({1.f, 0.f}, Metric::Cosine).take(5).collect(); });
        CHECK(msg.find("Query vector length 2 does not match expected dimension 3") != std::string::npos);
    }
    {   // test_empty_query_batch_error_handling, :66-76
        VecStore store(3);
        auto msg = error_of([&] { store.query(std::vector<std::vector<float>>{}, Metric::Cosine).take(5).collect(); });
        CHECK(msg == "No queries provided");
    }
    {   // test_vec_query_plan_new, :988-997
        auto msg = error_of([&] { VecQueryPlan().collect(); });
        CHECK(msg.find("Query vectors or their norms are not set") != std::string::npos);
    }
    {   // test_dimension_mismatch_during_add_vectors, :1148-1164
        VecStore store(3);
        auto msg = error_of([&] { store.add_vectors({{1.f, 0.f, 0.f}, {1.f, 0.f}}); });
        CHECK(msg.find("Input vector length 2 does not match expected dimension 3") != std::string::npos);
        CHECK(store.len() == 1);
    }
    {   // test_dot_product_take_max / take_min, :303-346
        VecStore store(2);
        store.add_vectors({{1.f, 0.f}, {2.f, 0.f}, {0.5f, 0.f}, {-1.f, 0.f}});
        auto r = store.query({1.f, 0.f}, Metric::DotProduct).take_max(2).collect();
        CHECK(r.size() == 2 && r[0].score == 2.0f && r[1].score == 1.0f && r[0].index == 1 && r[1].index == 0);
        r = store.query({1.f, 0.f}, Metric::DotProduct).take_min(2).collect();
        CHECK(r.size() == 2 && r[0].score == -1.0f && r[1].score == 0.5f);
    }
    {   // test_cosine_similarity_correctness, :545-608
        VecStore store(2);
        store.add_vectors({{1.f, 0.f}, {-1.f, 0.f}, {0.f, 1.f}, {1.f, 1.f}});
        auto r = store.query({1.f, 0.f}, Metric::Cosine).take(4).collect();
        CHECK(r.size() == 4);
        for (auto& h : r) {
            if (h.index == 0) CHECK(std::fabs(h.score - 1.0f) < 1e-6f);
            if (h.index == 1) CHECK(std::fabs(h.score + 1.0f) < 1e-6f);
            if (h.index == 2) CHECK(std::fabs(h.score) < 1e-6f);
            if (h.index == 3) CHECK(std::fabs(h.score - 0.70710678f) < 1e-5f);
        }
    }
    {   // test_euclidean_ranking_correctness, :801-851
        VecStore store(2);
        store.add_vectors({{0.f, 0.f}, {1.f, 0.f}, {0.f, 1.f}, {1.f, 1.f}, {2.f, 0.f}, {3.f, 4.f}});
        auto r = store.query({0.f, 0.f}, Metric::Euclidean).take_min(6).collect();
        const float want[6] = {0.f, 1.f, 1.f, 2.f, 4.f, 25.f};
        CHECK(r.size() == 6);
        for (int i = 0; i < 6 && i < (int)r.size(); i++) CHECK(r[i].score == want[i]);
    }
    {   // test_dot_product_filtering, :280-300 ; test_filtering_edge_cases, :1260-1286
        VecStore store(2);
        store.add_vectors({{2.f, 0.f}, {1.f, 0.f}, {0.5f, 0.f}, {-1.f, 0.f}});
        auto r = store.query({1.f, 0.f}, Metric::DotProduct).filter(1.0f, Cmp::Gt).take(10).collect();
        CHECK(r.size() == 1 && r[0].score == 2.0f);
        VecStore s2(2);
        s2.add_vectors({{1.f, 0.f}, {0.f, 1.f}, {-1.f, 0.f}});
        CHECK(s2.query({1.f, 0.f}, Metric::Cosine).filter(1.5f, Cmp::Gt).take(10).collect().empty());
        CHECK(s2.query({1.f, 0.f}, Metric::Cosine).filter(1.0f, Cmp::Eq).take(10).collect().size() == 1);
    }
    {   // test_batch_query_correctness, :899-924 (merged semantics)
        VecStore store(2);
        store.add_vectors({{1.f, 0.f}, {0.f, 1.f}, {-1.f, 0.f}});
        auto r = store.query(std::vector<std::vector<float>>{{1.f, 0.f}, {0.f, 1.f}}, Metric::Cosine).take(2).collect();
        CHECK(r.size() == 2 && std::fabs(r[0].score - 1.f) < 1e-6f && std::fabs(r[1].score - 1.f) < 1e-6f);
    }
    {   // take_zero :438-452, take_more_than_available :419-435, empty store :496-506, row mask
        VecStore store(2);
        store.add_vectors({{1.f, 0.f}, {0.f, 1.f}});
        CHECK(store.query({1.f, 0.f}, Metric::Cosine).take(0).collect().empty());
        CHECK(store.query({1.f, 0.f}, Metric::Cosine).take(10).collect().size() == 2);
        VecStore empty(3);
        CHECK(empty.query({1.f, 0.f, 0.f}, Metric::Cosine).take(5).collect().empty());
        auto r = store.query({1.f, 0.f}, Metric::Cosine).with_row_mask({false, true}).take(10).collect();
        CHECK(r.size() == 1 && r[0].index == 1);
    }
    {   // tie order.  Scores 1, 5, 5, 5, 9 and take(3): the reference's collector (src/vec_compute.rs:236-277)
        // fills with rows 0, 1, 2, sorts to [5(1), 5(2), 1(0)], inserts row 3 (5 > 1) IN FRONT of the run's last entry
        // -> [5(1), 5(3), 5(2)], then row 4 (9) in front of all and pops row 2: {4, 1, 3}.  The canonical order keeps {4, 1, 2}.
        // The reference's outcome is what this mirror (like the Rust patch) returns BY DEFAULT.
        VecStore store(1);
        store.add_vectors({{1.f}, {5.f}, {5.f}, {5.f}, {9.f}});
        auto r = store.query(std::vector<float>{1.f}, Metric::DotProduct).take(3).collect();
        CHECK(r.size() == 3 && r[0].index == 4 && r[0].score == 9.0f && r[1].score == 5.0f && r[2].score == 5.0f);
        CHECK((r[1].index == 1 && r[2].index == 3) || (r[1].index == 3 && r[2].index == 1));
        store.use_reference_tie_order(false);
        r = store.query(std::vector<float>{1.f}, Metric::DotProduct).take(3).collect();
        CHECK(r.size() == 3 && r[0].index == 4 && r[1].index == 1 && r[2].index == 2);
        store.use_reference_tie_order();
        r = store.query(std::vector<float>{1.f}, Metric::DotProduct).take(3).collect();
        CHECK(r.size() == 3 && r[0].index == 4 && ((r[1].index == 1 && r[2].index == 3) || (r[1].index == 3 && r[2].index == 1)));
    }
    std::printf(failures ? "FAILED (%d)\n" : "ALL PASSED\n", failures);
    return failures ? 1 : 0;
}

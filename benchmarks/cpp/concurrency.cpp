// Queries/sec from several host threads on ONE store through the C ABI (no Python in the way): overlapping ott_query
// calls run on separate query contexts (streams), so small, latency-bound corpora scale with the number of callers.
//   g++ -std=c++17 -O2 -I../../include concurrency.cpp -L../../otters_amd/csrc -lotters_hip -Wl,-rpath,... -lpthread
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>

#include "otters_hip.h"

static void check(int rc) {
    if (rc != 0) {
        fprintf(stderr, "libotters_hip: %s\n", ott_last_error());
        exit(1);
    }
}

int main(int argc, char** argv) {
    const uint32_t dim = argc > 1 ? (uint32_t)atoi(argv[1]) : 768;
    printf("| rows | threads | queries/s | mean latency us |\n|---|---|---|---|\n");
    for (uint64_t n : {10000ull, 100000ull, 1000000ull}) {
        ott_store* s = nullptr;
        check(ott_store_create(dim, 0, &s));
        check(ott_store_append_random(s, n, 5));
        std::mt19937 rng(1);
        std::uniform_real_distribution<float> u(-1.f, 1.f);
        std::vector<float> qs(64 * (size_t)dim);
        for (auto& x : qs) x = u(rng);
        for (int nt : {1, 2, 4, 8, 16, 32}) {
            const int per = n <= 100000 ? 2000 : 300;
            auto work = [&](int id, int reps) {
                ott_hit out[10];
                for (int j = 0; j < reps; j++) {
                    ott_query_desc d{};
                    d.queries = qs.data() + (size_t)((id * 7 + j) % 64) * dim;
                    d.nq = 1;
                    d.metric = OTT_METRIC_COSINE;
                    d.take = OTT_TAKE_MAX;
                    d.k = 10;
                    uint64_t n_out = 0;
                    check(ott_query(s, &d, out, 10, &n_out, nullptr, nullptr));
                }
            };
            {   // warm-up: creates the worker contexts
                std::vector<std::thread> th;
                for (int i = 0; i < nt; i++) th.emplace_back(work, i, 20);
                for (auto& t : th) t.join();
            }
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            for (int i = 0; i < nt; i++) th.emplace_back(work, i, per);
            for (auto& t : th) t.join();
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("| %llu | %d | %.0f | %.0f |\n", (unsigned long long)n, nt, nt * per / dt, dt / per * 1e6);
            fflush(stdout);
        }
        check(ott_store_destroy(s));
    }
    return 0;
}

/* c_host.c — a plain-C host of libotters_hip.so: what a Rust `extern "C"` binding does, without a C++ or Python layer
 * in between.  The reference's README example (README.md:63-150; data in tests/golden/meta_cases.json,
 * "readme_example_8x4"): 8 x 4 vectors, chunk size 4, two metadata columns resident in HBM, the filter
 * price <= 40 AND version >= 2 evaluated on the GPU (ott_store_eval_row_mask), cosine query [1,0,0,0], take(5) — through
 * ott_query and through ott_query_sharded on a HOST-transport communicator of one rank (an all-gather callback written in
 * C), and once more on ONE store spread over several shards of this process (ott_store_create_multi, device list {0, 0, 0}:
 * the same calls, the columns and the mask routed by row range, the same hits).  tests/test_gpu_cpp_mirror.py compares the printed hits with the reference's documented result: rows [4, 2, 6],
 * scores 0.970142 / 0.707107 / 0.707107, stats 2 chunks / 2 evaluated / 8 compared.
 * Pure C11 (-pedantic -Werror); exit code 0 on success. */
#include <stdio.h>
#include <string.h>

#include "otters_hip.h"

static int gather_one_rank(void* user, const void* send, void* recv, uint64_t bytes) {
    (void)user;
    memcpy(recv, send, (size_t)bytes); /* world size 1: the gathered block is the block */
    return 0;
}

#define CHECK(call)                                                          \
    do {                                                                     \
        if ((call) != OTT_OK) {                                              \
            fprintf(stderr, "%s failed: %s\n", #call, ott_last_error());     \
            return 1;                                                        \
        }                                                                    \
    } while (0)

int main(void) {
    static const float rows[8 * 4] = {1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 1.0f, 1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f,
                                      0.8f, 0.2f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.6f, 0.6f, 0.0f, 0.0f, 0.0f, 0.5f, 0.5f, 0.0f};
    static const double price[8] = {19.99, 49.0, 12.5, 8.99, 29.99, 5.99, 39.5, 59.99};
    static const int32_t version[8] = {1, 2, 2, 1, 3, 1, 2, 3};
    static const float query[4] = {1.0f, 0.0f, 0.0f, 0.0f};
    ott_leaf leaves[2];
    uint32_t col_price = 0, col_version = 0;
    uint64_t mask_words[1] = {0};
    static const int devs[3] = {0, 0, 0};
    ott_store* store = NULL;
    ott_store* multi = NULL;
    ott_hit mhits[5];
    float far_rows[16 * 4];
    double price24[24];
    int32_t version24[24];
    uint64_t nm = 0, first = 0, cnt = 0, total = 0;
    int dev = -1, g;
    ott_comm* comm = NULL;
    ott_query_desc d;
    ott_hit hits[5], shits[5];
    ott_stats st;
    uint64_t n = 0, ns = 0, i;

    CHECK(ott_store_create(4, 0, &store));
    CHECK(ott_store_set_chunk_size(store, 4));
    CHECK(ott_store_append(store, rows, 8));
    CHECK(ott_store_add_column(store, OTT_DT_FLOAT64, price, NULL, 8, &col_price));
    CHECK(ott_store_add_column(store, OTT_DT_INT32, version, NULL, 8, &col_version));
    memset(leaves, 0, sizeof leaves);
    leaves[0].column = col_price;   leaves[0].op = OTT_OP_LTE; leaves[0].clause = 0; leaves[0].lit_f64 = 40.0;
    leaves[1].column = col_version; leaves[1].op = OTT_OP_GTE; leaves[1].clause = 1; leaves[1].lit_i64 = 2;
    CHECK(ott_store_eval_row_mask(store, leaves, 2, 2, mask_words));
    if (mask_words[0] != 0x54u) { /* rows 2, 4, 6 */
        fprintf(stderr, "row mask 0x%llx, expected 0x54\n", (unsigned long long)mask_words[0]);
        return 1;
    }
    memset(&d, 0, sizeof d);
    d.queries = query;
    d.nq = 1;
    d.metric = OTT_METRIC_COSINE;
    d.take = OTT_TAKE_MAX;
    d.filter_cmp = OTT_CMP_NONE;
    d.mode = OTT_MODE_MERGED;
    d.k = 5;
    d.use_device_row_mask = 1;
    CHECK(ott_query(store, &d, hits, 5, &n, NULL, &st));
    CHECK(ott_comm_create_host(0, 1, gather_one_rank, NULL, &comm));
    CHECK(ott_query_sharded(store, comm, &d, shits, 5, &ns, NULL, NULL));
    if (n != ns || memcmp(hits, shits, (size_t)n * sizeof(ott_hit)) != 0) {
        fprintf(stderr, "sharded result differs from the plain one\n");
        return 1;
    }
    /* the same example on ONE store over three shards of this process: nothing but the create call differs.  Shards hold whole
     * granules of lcm(chunk size, 8) rows, so the 8 README rows are followed by 16 rows that can never be hits (opposite the
     * query, price 99): one granule per shard, and the store is told to use every shard however small it is */
    CHECK(ott_store_create_multi(4, 3, devs, &multi));
    CHECK(ott_store_set_option(multi, "multi_min_shard_rows", 0));
    CHECK(ott_store_set_chunk_size(multi, 4));
    CHECK(ott_store_reserve(multi, 24));
    CHECK(ott_store_append(multi, rows, 3));      /* appended in pieces, as VecStore::add_vector would */
    CHECK(ott_store_append(multi, rows + 3 * 4, 5));
    for (i = 0; i < 16; i++) {
        far_rows[i * 4 + 0] = -1.0f;
        far_rows[i * 4 + 1] = far_rows[i * 4 + 2] = far_rows[i * 4 + 3] = 0.0f;
    }
    CHECK(ott_store_append(multi, far_rows, 16));
    for (i = 0; i < 24; i++) {
        price24[i] = i < 8 ? price[i] : 99.0;
        version24[i] = i < 8 ? version[i] : 1;
    }
    CHECK(ott_store_add_column(multi, OTT_DT_FLOAT64, price24, NULL, 24, &col_price));
    CHECK(ott_store_add_column(multi, OTT_DT_INT32, version24, NULL, 24, &col_version));
    leaves[0].column = col_price;
    leaves[1].column = col_version;
    mask_words[0] = 0;
    CHECK(ott_store_eval_row_mask(multi, leaves, 2, 2, mask_words));
    if (mask_words[0] != 0x54u) {
        fprintf(stderr, "multi-GPU store: row mask 0x%llx, expected 0x54\n", (unsigned long long)mask_words[0]);
        return 1;
    }
    CHECK(ott_query(multi, &d, mhits, 5, &nm, NULL, NULL));
    if (nm != n || memcmp(hits, mhits, (size_t)n * sizeof(ott_hit)) != 0) {
        fprintf(stderr, "multi-GPU store: result differs from the single store's\n");
        return 1;
    }
    if (ott_store_shard_count(multi) != 3 || ott_store_len(multi) != 24) return 1;
    for (g = 0; g < 3; g++) {
        CHECK(ott_store_shard_info(multi, (uint32_t)g, &dev, &first, &cnt));
        if (dev != 0 || first != (uint64_t)g * 8 || cnt != 8) return 1;
        total += cnt;
    }
    if (total != 24) return 1;
    printf("multi shards 3 transport %s\n", ott_store_transport(multi));
    CHECK(ott_store_destroy(multi));
    printf("chunks %llu evaluated %llu compared %llu\n", (unsigned long long)st.total_chunks, (unsigned long long)st.evaluated_chunks,
           (unsigned long long)st.vectors_compared);
    for (i = 0; i < n; i++) printf("hit %llu score %.6f\n", (unsigned long long)hits[i].index, (double)hits[i].score);
    CHECK(ott_comm_destroy(comm));
    CHECK(ott_store_destroy(store));
    return 0;
}

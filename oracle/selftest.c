/* oracle/selftest.c — sanitizer self-test of the oracle (TEST INFRASTRUCTURE).  Built with
 * -fsanitize=address,undefined by `make -C oracle asan` and run on the CPU; exercises every
 * entry point on ragged shapes so out-of-bounds reads / UB in the checker itself surface. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "otters_oracle.h"

static uint64_t rng_state = 88172645463325252ull;
static double frand(void) {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (double)(rng_state >> 11) / 9007199254740992.0;
}

int main(void) {
    int fails = 0;
    const size_t dims[] = {1, 3, 7, 8, 9, 31, 32, 33, 100};
    for (size_t di = 0; di < sizeof(dims) / sizeof(dims[0]); di++) {
        const size_t dim = dims[di];
        for (size_t n = 0; n <= 70; n += 7) {
            const size_t nq = 1 + (n % 3);
            float* rows = malloc((n * dim + 1) * sizeof(float));
            float* q = malloc(nq * dim * sizeof(float));
            float* inv = malloc((n + 1) * sizeof(float));
            for (size_t i = 0; i < n * dim; i++) rows[i] = (float)(frand() * 2 - 1);
            for (size_t i = 0; i < nq * dim; i++) q[i] = (float)(frand() * 2 - 1);
            if (n > 3) rows[2 * dim] = NAN;
            otto_inv_norms(rows, n, dim, inv);
            const size_t words = (n + 63) / 64 + 1;
            uint64_t* mask = calloc(words, 8);
            for (size_t i = 0; i < n; i++)
                if (frand() < 0.6) mask[i >> 6] |= 1ull << (i & 63);
            otto_hit* out = malloc((n * nq + 1) * sizeof(otto_hit));
            for (int metric = 0; metric < 3; metric++)
                for (int take = 0; take < 2; take++)
                    for (int ties = 0; ties < 2; ties++)
                        for (int cmp = 0; cmp <= 5; cmp++) {
                            size_t ks[] = {0, 1, 5, n * nq + 3};
                            for (int ki = 0; ki < 4; ki++) {
                                size_t m = otto_vec_query(rows, inv, n, dim, q, nq, metric, take, ks[ki], cmp, 0.1f, (ki & 1) ? mask : NULL,
                                                          n > 5 ? n - 5 : n, ties & 1, ties, out);
                                size_t cap = ks[ki] < n * nq ? ks[ki] : n * nq;
                                if (m > cap) fails++;
                                for (size_t i = 1; i < m; i++) {
                                    int bad = take == OTTO_TAKE_MAX ? out[i - 1].score < out[i].score : out[i - 1].score > out[i].score;
                                    if (bad) fails++;
                                }
                                if (n) {
                                    otto_stats st;
                                    size_t cs = 1 + n / 3;
                                    size_t nch = (n + cs - 1) / cs;
                                    uint64_t cm = 0;
                                    for (size_t c = 0; c < nch; c++)
                                        if (c % 2 == 0) cm |= 1ull << c;
                                    size_t m2 = otto_meta_query(rows, inv, n, dim, cs, q, nq, metric, take, ks[ki], cmp, 0.1f, &cm, (ki & 1) ? mask : NULL,
                                                                OTTO_REDUCE_AVX, ties, 1 + (ki % 3), out, &st);
                                    if (m2 > cap || st.total_chunks != nch) fails++;
                                }
                            }
                        }
            free(out);
            free(mask);
            free(inv);
            free(q);
            free(rows);
        }
    }
    /* mask helpers */
    {
        int32_t v32[19];
        int64_t v64[19];
        float f32[19];
        double f64[19];
        uint64_t nulls = 0x5a5a5, outw[2];
        for (int i = 0; i < 19; i++) {
            v32[i] = i - 9;
            v64[i] = i - 9;
            f32[i] = (float)(i - 9);
            f64[i] = i - 9;
        }
        for (int op = 0; op < 6; op++) {
            outw[0] = outw[1] = 0;
            otto_rows_mask_i32(v32, &nulls, 19, 2, 17, op, 0, outw);
            otto_rows_mask_i64(v64, &nulls, 19, 2, 17, op, 0, outw);
            otto_rows_mask_f32(f32, NULL, 0, 2, 17, op, 0, outw);
            otto_rows_mask_f64(f64, &nulls, 10, 2, 17, op, 0, outw);
            if (outw[0] >> 17) fails++;
            uint64_t nn[3] = {1, 0, 2};
            int32_t mn[3] = {-1, 5, 7}, mx[3] = {3, 9, 7};
            outw[0] = 0;
            otto_chunk_mask_i32(mn, mx, nn, 3, op, 7, outw);
            if (outw[0] & 2) fails++; /* all-null chunk never survives */
        }
        int64_t a, b;
        uint64_t c;
        otto_zone_stat_i32(v32, &nulls, 19, 0, 19, &a, &b, &c);
        double da, db;
        otto_zone_stat_f32(f32, &nulls, 19, 3, 3, &da, &db, &c);
        if (c != 0 || !isinf(da)) fails++;
    }
    float buf[40];
    otto_rand_fill(buf, 5, 4, 10, 7);
    for (int i = 0; i < 40; i++)
        if (!(buf[i] >= -1.0f && buf[i] < 1.0f)) fails++;
    printf(fails ? "SELFTEST FAILED (%d)\n" : "SELFTEST OK\n", fails);
    return fails ? 1 : 0;
}

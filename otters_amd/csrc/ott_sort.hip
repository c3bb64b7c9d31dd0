// ott_sort.hip — large-k path (k > 512, e.g. collect() with no take on a big store:
// take_count defaults to n_vecs, src/vec.rs:213).  The fused register top-k does not scale to
// thousands of entries, so the exact scorer dumps every passing (key, query) pair and a device
// radix sort orders them: the reference's own final step is a full sort of the
// collector (`into_sorted_vec`, src/vec_compute.rs:290-293; meta.rs:702-705).
//
// The sort is a plain stable LSD radix sort, 8 bits per pass, written here (no library): per pass a histogram kernel
// (per-workgroup digit counts of a contiguous chunk), one scan of the [digit][workgroup] table, and a scatter kernel that
// re-reads the chunk in order and ranks equal digits by (wave ballots, then waves in order, then tiles in order), so equal
// keys keep their input order.  HBM-bound: every pass reads and writes the pairs once.
#include <string.h>

#include <algorithm>

#include "ott_internal.h"

namespace ott {

// ---- stable LSD radix sort of (key, value) pairs -------------------------------------------------------------------------
constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = RS_THREADS / 64;

template <typename K>
__device__ __forceinline__ uint32_t rs_digit(K key, int shift, bool descending) {
    const uint32_t d = (uint32_t)(key >> shift) & 255u;
    return descending ? 255u - d : d;
}

// hist[digit * n_blocks + block] = how many keys of this block's chunk have that digit
template <typename K>
__global__ __launch_bounds__(RS_THREADS) void rs_hist_kernel(const K* __restrict__ keys, uint64_t n, uint64_t chunk, int shift, uint32_t descending,
                                                              uint32_t* __restrict__ hist) {
    __shared__ uint32_t cnt[256];
    cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t b0 = (uint64_t)blockIdx.x * chunk, b1 = b0 + chunk < n ? b0 + chunk : n;
    for (uint64_t i = b0 + threadIdx.x; i < b1; i += RS_THREADS) atomicAdd(&cnt[rs_digit(keys[i], shift, descending != 0)], 1u);
    __syncthreads();
    hist[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = cnt[threadIdx.x];
}

// one workgroup per digit: exclusive scan of that digit's row of the table (n_blocks <= 4096 counters, four per thread) in
// place, and the row's total into totals[digit]
__global__ __launch_bounds__(1024) void rs_rowscan_kernel(uint32_t* __restrict__ hist, uint32_t n_blocks, uint32_t* __restrict__ totals) {
    __shared__ uint32_t part[1024];
    uint32_t* row = hist + (size_t)blockIdx.x * n_blocks;
    uint32_t v[4], sum = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t i = threadIdx.x * 4 + j;
        v[j] = i < n_blocks ? row[i] : 0u;
        sum += v[j];
    }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {  // inclusive scan of the per-thread sums
        const uint32_t u = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += u;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t i = threadIdx.x * 4 + j;
        if (i < n_blocks) row[i] = run;
        run += v[j];
    }
    if (threadIdx.x == 1023) totals[blockIdx.x] = part[1023];
}

template <typename K, typename V>
__global__ __launch_bounds__(RS_THREADS) void rs_scatter_kernel(const K* __restrict__ keys_in, const V* __restrict__ vals_in, K* __restrict__ keys_out,
                                                                 V* __restrict__ vals_out, uint64_t n, uint64_t chunk, int shift, uint32_t descending,
                                                                 const uint32_t* __restrict__ offs, const uint32_t* __restrict__ totals) {
    __shared__ uint32_t base[256];            // next output slot of every digit for this block
    __shared__ uint32_t wcnt[RS_WAVES][256];  // per tile: how many keys of each digit every wave holds
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // where digit d starts in the output: the exclusive scan of the 256 digit totals (done by every block, in LDS)
    base[threadIdx.x] = totals[threadIdx.x];
    __syncthreads();
    for (uint32_t off = 1; off < 256; off <<= 1) {
        const uint32_t u = threadIdx.x >= off ? base[threadIdx.x - off] : 0u;
        __syncthreads();
        base[threadIdx.x] += u;
        __syncthreads();
    }
    {
        const uint32_t excl = base[threadIdx.x] - totals[threadIdx.x];
        __syncthreads();
        base[threadIdx.x] = excl + offs[(size_t)threadIdx.x * gridDim.x + blockIdx.x];
    }
#pragma unroll
    for (int w = 0; w < RS_WAVES; w++) wcnt[w][threadIdx.x] = 0;
    __syncthreads();
    const uint64_t b0 = (uint64_t)blockIdx.x * chunk, b1 = b0 + chunk < n ? b0 + chunk : n;
    for (uint64_t t0 = b0; t0 < b1; t0 += RS_THREADS) {
        const uint64_t i = t0 + threadIdx.x;
        const bool have = i < b1;
        K key = 0;
        V val = 0;
        if (have) {
            key = keys_in[i];
            val = vals_in[i];
        }
        const uint32_t d = have ? rs_digit(key, shift, descending != 0) : 0u;
        // lanes of this wave with the same digit: eight ballots, one per digit bit
        unsigned long long peers = __ballot(have);
#pragma unroll
        for (int bit = 0; bit < 8; bit++) {
            const unsigned long long m = __ballot((d >> bit) & 1u);
            peers &= ((d >> bit) & 1u) ? m : ~m;
        }
        const uint32_t rank_in_wave = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
        if (have && rank_in_wave == 0) wcnt[wave][d] = (uint32_t)__popcll(peers);  // the first lane of each digit group
        __syncthreads();
        if (have) {
            uint32_t before = 0;
            for (int w = 0; w < wave; w++) before += wcnt[w][d];
            const uint32_t pos = base[d] + before + rank_in_wave;
            keys_out[pos] = key;
            vals_out[pos] = val;
        }
        __syncthreads();
        {
            uint32_t tot = 0;
#pragma unroll
            for (int w = 0; w < RS_WAVES; w++) {
                tot += wcnt[w][threadIdx.x];
                wcnt[w][threadIdx.x] = 0;
            }
            base[threadIdx.x] += tot;
        }
        __syncthreads();
    }
}

// sorts n pairs by bits [0, bits) of the key, stable; the result ends in (*keys, *vals) (the buffers are swapped per pass).
// tmp: at least rs_tmp_bytes(n) bytes of device memory.
static size_t rs_blocks(uint64_t n) { return (size_t)std::min<uint64_t>(4096, (n + 4095) / 4096 ? (n + 4095) / 4096 : 1); }
static size_t rs_tmp_bytes(uint64_t n) { return (rs_blocks(n) + 1) * 256 * sizeof(uint32_t); }  // the table + the 256 digit totals

template <typename K, typename V>
static int radix_sort_pairs(hipStream_t stream, K** keys, K** keys_alt, V** vals, V** vals_alt, uint64_t n, int bits, bool descending, void* tmp) {
    if (n > 0xFFFFFFFFull) return fail(OTT_ERR_UNSUPPORTED, "radix_sort_pairs: more than 2^32 - 1 pairs");
    const uint32_t nb = (uint32_t)rs_blocks(n);
    const uint64_t chunk = ((n + nb - 1) / nb + RS_THREADS - 1) / RS_THREADS * RS_THREADS;  // whole tiles per block
    uint32_t* hist = (uint32_t*)tmp;
    uint32_t* totals = hist + (size_t)nb * 256;
    for (int shift = 0; shift < bits; shift += 8) {
        hipLaunchKernelGGL((rs_hist_kernel<K>), dim3(nb), dim3(RS_THREADS), 0, stream, *keys, n, chunk, shift, descending ? 1u : 0u, hist);
        hipLaunchKernelGGL(rs_rowscan_kernel, dim3(256), dim3(1024), 0, stream, hist, nb, totals);
        hipLaunchKernelGGL((rs_scatter_kernel<K, V>), dim3(nb), dim3(RS_THREADS), 0, stream, *keys, *vals, *keys_alt, *vals_alt, n, chunk, shift,
                           descending ? 1u : 0u, hist, totals);
        OTT_HIP(hipGetLastError());
        std::swap(*keys, *keys_alt);
        std::swap(*vals, *vals_alt);
    }
    return OTT_OK;
}

__global__ __launch_bounds__(256) void hits_from_sorted_kernel(const uint64_t* keys, const uint32_t* qs, uint64_t first, uint64_t count,
                                                                uint32_t take_max, uint64_t base, ott_hit* out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint64_t key = keys[first + i];
    ott_hit h;
    h.index = base + (uint32_t)~(uint32_t)(key & 0xFFFFFFFFull);
    h.score = score_of((uint32_t)(key >> 32), take_max != 0);
    h.query = qs[first + i];
    out[i] = h;
}

__global__ __launch_bounds__(256) void hist_q_kernel(const uint32_t* qs, uint64_t n, uint32_t* hist) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) atomicAdd(&hist[qs[i]], 1u);
}

int run_large_k(ott_store* s, const float* queries, uint32_t nq, const ott_query_desc* d, bool perq, const RunPlan& pl, uint64_t k_eff,
                const uint64_t* d_mask, uint64_t mask_bits, std::vector<std::vector<ott_hit>>& lists, ott_stats& st) {
    const uint64_t cap = pl.rows_scored * nq;
    if (cap > (1ull << 31)) return fail(OTT_ERR_UNSUPPORTED, "ott_query: k > 512 over more than 2^31 (row, query) pairs is not supported");
    const std::vector<uint32_t> prefix = tile_prefix(pl, 64);
    int rc = upload_exact_inputs(s, queries, nq, pl, prefix);
    if (rc) return rc;
    if ((rc = s->l_keysA.ensure(cap * 8))) return rc;
    if ((rc = s->l_keysB.ensure(cap * 8))) return rc;
    if ((rc = s->l_qA.ensure(cap * 4))) return rc;
    if ((rc = s->l_qB.ensure(cap * 4))) return rc;
    if ((rc = s->l_cursor.ensure(8))) return rc;
    OTT_HIP(hipMemsetAsync(s->l_cursor.p, 0, 8, s->stream));

    ExactParams p;
    fill_exact_params(s, d, pl, nq, d_mask, mask_bits, prefix.back(), p);
    p.k = 1;
    p.dump_keys = (uint64_t*)s->l_keysA.p;
    p.dump_q = (uint32_t*)s->l_qA.p;
    p.dump_cursor = (unsigned long long*)s->l_cursor.p;
    p.dump_cap = cap;
    const int tile = nq == 1 ? 1 : 4;
    const uint32_t passes = (nq + tile - 1) / tile;
    const int grid = exact_grid(s, prefix.back());
    OTT_HIP(hipEventRecord(s->ev[3], s->stream));
    for (uint32_t ps = 0; ps < passes; ps++) {
        p.q0 = ps * tile;
        if ((rc = launch_exact_dump(s, p, tile, grid))) return rc;
    }
    OTT_HIP(hipEventRecord(s->ev[4], s->stream));
    unsigned long long n_entries = 0;
    OTT_HIP(hipMemcpyAsync(&n_entries, s->l_cursor.p, 8, hipMemcpyDeviceToHost, s->stream));
    OTT_HIP(hipStreamSynchronize(s->stream));
    if (n_entries > cap) n_entries = cap;

    uint64_t* kA = (uint64_t*)s->l_keysA.p;
    uint64_t* kB = (uint64_t*)s->l_keysB.p;
    uint32_t* qA = (uint32_t*)s->l_qA.p;
    uint32_t* qB = (uint32_t*)s->l_qB.p;
    const uint32_t groups = perq ? nq : 1;
    lists.assign(groups, {});
    if (n_entries) {
        if ((rc = s->l_tmp.ensure(rs_tmp_bytes(n_entries)))) return rc;
        if (!perq) {
            // canonical merged order: key descending, ties by query ascending  (stable LSD: query first)
            if (nq > 1 && (rc = radix_sort_pairs<uint32_t, uint64_t>(s->stream, &qA, &qB, &kA, &kB, n_entries, 32, false, s->l_tmp.p))) return rc;
            if ((rc = radix_sort_pairs<uint64_t, uint32_t>(s->stream, &kA, &kB, &qA, &qB, n_entries, 64, true, s->l_tmp.p))) return rc;
        } else {
            // grouped by query, each group key descending: key first, then stable by query
            if ((rc = radix_sort_pairs<uint64_t, uint32_t>(s->stream, &kA, &kB, &qA, &qB, n_entries, 64, true, s->l_tmp.p))) return rc;
            if ((rc = radix_sort_pairs<uint32_t, uint64_t>(s->stream, &qA, &qB, &kA, &kB, n_entries, 32, false, s->l_tmp.p))) return rc;
        }
        // group extents
        std::vector<uint64_t> first(groups, 0), count(groups, 0);
        if (!perq) count[0] = n_entries < k_eff ? n_entries : k_eff;
        else {
            if ((rc = s->l_hist.ensure((size_t)nq * 4))) return rc;
            OTT_HIP(hipMemsetAsync(s->l_hist.p, 0, (size_t)nq * 4, s->stream));
            hipLaunchKernelGGL(hist_q_kernel, dim3((uint32_t)s->n_cu * 4), dim3(256), 0, s->stream, qA, (uint64_t)n_entries, (uint32_t*)s->l_hist.p);
            OTT_HIP(hipGetLastError());
            std::vector<uint32_t> h(nq);
            OTT_HIP(hipMemcpyAsync(h.data(), s->l_hist.p, (size_t)nq * 4, hipMemcpyDeviceToHost, s->stream));
            OTT_HIP(hipStreamSynchronize(s->stream));
            uint64_t off = 0;
            for (uint32_t q = 0; q < nq; q++) {
                first[q] = off;
                count[q] = h[q] < k_eff ? h[q] : k_eff;
                off += h[q];
            }
        }
        uint64_t total = 0;
        for (uint32_t g = 0; g < groups; g++) total += count[g];
        if ((rc = s->d_hits.ensure((size_t)(total ? total : 1) * sizeof(ott_hit)))) return rc;
        uint64_t o = 0;
        for (uint32_t g = 0; g < groups; g++) {
            if (!count[g]) continue;
            hipLaunchKernelGGL(hits_from_sorted_kernel, dim3((uint32_t)((count[g] + 255) / 256)), dim3(256), 0, s->stream, kA, qA, first[g],
                               count[g], d->take == OTT_TAKE_MAX ? 1u : 0u, s->base_offset, (ott_hit*)s->d_hits.p + o);
            OTT_HIP(hipGetLastError());
            o += count[g];
        }
        OTT_HIP(hipEventRecord(s->ev[5], s->stream));
        o = 0;
        for (uint32_t g = 0; g < groups; g++) {
            lists[g].resize(count[g]);
            if (count[g]) OTT_HIP(hipMemcpyAsync(lists[g].data(), (ott_hit*)s->d_hits.p + o, count[g] * sizeof(ott_hit), hipMemcpyDeviceToHost, s->stream));
            o += count[g];
        }
        OTT_HIP(hipStreamSynchronize(s->stream));
    } else {
        OTT_HIP(hipEventRecord(s->ev[5], s->stream));
        OTT_HIP(hipStreamSynchronize(s->stream));
    }
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s->ev[3], s->ev[4]) == hipSuccess) st.score_ns += (uint64_t)(ms * 1e6);
    if (hipEventElapsedTime(&ms, s->ev[4], s->ev[5]) == hipSuccess) st.merge_ns += (uint64_t)(ms * 1e6);
    st.passes += passes;
    st.bytes_scanned += (uint64_t)passes * pl.rows_scored * ((uint64_t)s->dim * 4 + (d->metric == OTT_METRIC_COSINE ? 4 : 0));
    return OTT_OK;
}

}  // namespace ott

// ott_sort.hip — large-k path (k > 512, e.g. collect() with no take on a big store:
// take_count defaults to n_vecs, src/vec.rs:213).  The fused register top-k does not scale to
// thousands of entries, so the exact scorer dumps every passing (key, query) pair and a device
// radix sort (hipCUB) orders them: the reference's own final step is a full sort of the
// collector (`into_sorted_vec`, src/vec_compute.rs:290-293; meta.rs:702-705).
#include <string.h>

#include <hipcub/hipcub.hpp>

#include "ott_internal.h"

namespace ott {

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
        // temp storage sized for the larger of the two sorts
        size_t t1 = 0, t2 = 0;
        (void)hipcub::DeviceRadixSort::SortPairsDescending(nullptr, t1, kA, kB, qA, qB, (size_t)n_entries, 0, 64, s->stream);
        (void)hipcub::DeviceRadixSort::SortPairs(nullptr, t2, qA, qB, kA, kB, (size_t)n_entries, 0, 32, s->stream);
        if ((rc = s->l_tmp.ensure(t1 > t2 ? t1 : t2))) return rc;
        size_t tb = s->l_tmp.cap;
        if (!perq) {
            // canonical merged order: key descending, ties by query ascending  (stable LSD: query first)
            if (nq > 1) {
                OTT_HIP(hipcub::DeviceRadixSort::SortPairs(s->l_tmp.p, tb, qA, qB, kA, kB, (size_t)n_entries, 0, 32, s->stream));
                std::swap(kA, kB);
                std::swap(qA, qB);
                tb = s->l_tmp.cap;
            }
            OTT_HIP(hipcub::DeviceRadixSort::SortPairsDescending(s->l_tmp.p, tb, kA, kB, qA, qB, (size_t)n_entries, 0, 64, s->stream));
            std::swap(kA, kB);
            std::swap(qA, qB);
        } else {
            // grouped by query, each group key descending: key first, then stable by query
            OTT_HIP(hipcub::DeviceRadixSort::SortPairsDescending(s->l_tmp.p, tb, kA, kB, qA, qB, (size_t)n_entries, 0, 64, s->stream));
            std::swap(kA, kB);
            std::swap(qA, qB);
            tb = s->l_tmp.cap;
            OTT_HIP(hipcub::DeviceRadixSort::SortPairs(s->l_tmp.p, tb, qA, qB, kA, kB, (size_t)n_entries, 0, 32, s->stream));
            std::swap(kA, kB);
            std::swap(qA, qB);
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

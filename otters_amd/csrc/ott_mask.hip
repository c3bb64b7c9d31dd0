// ott_mask.hip — GPU-side build_row_mask_for_chunk (src/meta_compute.rs:194-289) for numeric
// and datetime leaves: metadata columns and their null bitmaps live in HBM; a compiled CNF
// filter (src/expr.rs:213-226: AND of clauses, each an OR of ColumnFilter leaves) is evaluated
// for every row at once and written as BitVec<usize,Lsb0> words that the scoring kernels test.
// Row predicate = plain comparison AND not-null (src/type_utils.rs:306-444, 587-736); the
// reference evaluates it per surviving chunk on the host, the result is the same bits.
#include <math.h>
#include <string.h>

#include "ott_internal.h"

namespace ott {

struct DevLeaf {
    const void* vals;
    const uint64_t* nulls;  // 1 = NULL; may be null
    uint32_t dtype, op, clause, pad;
    int64_t lit_i64;
    double lit_f64;
};

template <typename T>
__device__ __forceinline__ bool op_holds(T v, uint32_t op, T t) {
    switch (op) {  // src/type_utils.rs:609-616
        case OTT_OP_EQ: return v == t;
        case OTT_OP_NEQ: return v != t;
        case OTT_OP_LT: return v < t;
        case OTT_OP_LTE: return v <= t;
        case OTT_OP_GT: return v > t;
        default: return v >= t;
    }
}

constexpr uint32_t MASK_EMBED = 32;  // leaves that travel in the kernel arguments (no H2D copy in front of the launch)
constexpr int MASK_U = 8;            // mask words per wave and step: eight independent loads in flight per leaf

struct MaskParams {
    const DevLeaf* leaves;  // used when n_leaves > MASK_EMBED
    uint32_t n_leaves, embedded;
    uint64_t n_rows;
    uint64_t* out;
    DevLeaf eleaves[MASK_EMBED];
};

template <typename T>
__device__ __forceinline__ void leaf_sat(const void* vals, uint32_t op, T lit, const uint64_t (&row)[MASK_U], const bool (&in)[MASK_U],
                                         bool (&sat)[MASK_U]) {
    T v[MASK_U];
#pragma unroll
    for (int u = 0; u < MASK_U; u++) v[u] = ((const T*)vals)[in[u] ? row[u] : 0];  // MASK_U loads issued back to back, no branch
#pragma unroll
    for (int u = 0; u < MASK_U; u++) sat[u] = in[u] && op_holds<T>(v[u], op, lit);
}

// one wave = MASK_U x 64 consecutive rows = MASK_U mask words per step; the leaves are wave-uniform and read through the
// constant address space (scalar loads), from the kernel arguments when they fit there
__global__ __launch_bounds__(256) void eval_mask_kernel(MaskParams p) {
    typedef __attribute__((address_space(4))) const DevLeaf* CLEAF;
    typedef __attribute__((address_space(4))) const char* CCH;
    const CLEAF leaves = p.embedded ? (CLEAF)((CCH)__builtin_amdgcn_kernarg_segment_ptr() + __builtin_offsetof(MaskParams, eleaves)) : (CLEAF)p.leaves;
    const int lane = threadIdx.x & 63;
    const uint64_t n_words = (p.n_rows + 63) / 64;
    for (uint64_t w0 = ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * MASK_U; w0 < n_words; w0 += (uint64_t)gridDim.x * 4 * MASK_U) {
        uint64_t row[MASK_U];
        bool in[MASK_U], all[MASK_U], any[MASK_U];
#pragma unroll
        for (int u = 0; u < MASK_U; u++) {
            row[u] = (w0 + u) * 64 + lane;
            in[u] = row[u] < p.n_rows;
            all[u] = true;   // fold(bitvec![1; len]) over clauses, src/meta_compute.rs:203
            any[u] = false;  // clause_mask = bitvec![0; len]
        }
        uint32_t cur = p.n_leaves ? leaves[0].clause : 0;
        for (uint32_t i = 0; i < p.n_leaves; i++) {
            const void* vals = leaves[i].vals;
            const uint64_t* nulls = leaves[i].nulls;
            const uint32_t dtype = leaves[i].dtype, op = leaves[i].op, clause = leaves[i].clause;
            if (clause != cur) {
#pragma unroll
                for (int u = 0; u < MASK_U; u++) {
                    all[u] = all[u] && any[u];
                    any[u] = false;
                }
                cur = clause;
            }
            uint64_t nw[MASK_U];
#pragma unroll
            for (int u = 0; u < MASK_U; u++) nw[u] = (nulls != nullptr && w0 + u < n_words) ? nulls[w0 + u] : 0ull;
            bool sat[MASK_U];
            switch (dtype) {
                case OTT_DT_INT32: leaf_sat<int32_t>(vals, op, (int32_t)leaves[i].lit_i64, row, in, sat); break;
                case OTT_DT_FLOAT32: leaf_sat<float>(vals, op, (float)leaves[i].lit_f64, row, in, sat); break;
                case OTT_DT_FLOAT64: leaf_sat<double>(vals, op, leaves[i].lit_f64, row, in, sat); break;
                default: leaf_sat<int64_t>(vals, op, leaves[i].lit_i64, row, in, sat); break;  // Int64 / DateTime
            }
#pragma unroll
            for (int u = 0; u < MASK_U; u++) any[u] = any[u] || (sat[u] && !((nw[u] >> lane) & 1));  // NULL never satisfies
        }
#pragma unroll
        for (int u = 0; u < MASK_U; u++) {
            if (p.n_leaves) all[u] = all[u] && any[u];
            const uint64_t word = __ballot(all[u] && in[u]);
            if (lane == 0 && w0 + u < n_words) p.out[w0 + u] = word;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// zone statistics per chunk (one wave per chunk): build_zone_stat_for_range, src/meta_compute.rs:32-132
// ---------------------------------------------------------------------------------------------
template <typename T, typename A, bool IS_FLOAT>
__global__ __launch_bounds__(256) void zone_stat_kernel(const T* __restrict__ vals, const uint64_t* __restrict__ nulls, uint64_t n,
                                                         uint64_t chunk_size, uint64_t n_chunks, A* __restrict__ out_min,
                                                         A* __restrict__ out_max, uint64_t* __restrict__ out_nn, A init_min, A init_max) {
    const int lane = threadIdx.x & 63;
    for (uint64_t c = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); c < n_chunks; c += (uint64_t)gridDim.x * 4) {
        const uint64_t lo = c * chunk_size, hi = (lo + chunk_size) < n ? (lo + chunk_size) : n;
        A mn = init_min, mx = init_max;
        uint64_t cnt = 0;
        // four independent loads per step (a chunk of 1024 rows is four steps per wave instead of sixteen dependent ones)
        for (uint64_t i0 = lo + lane; i0 < hi; i0 += 256) {
            T v[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint64_t i = i0 + 64 * u;
                ok[u] = i < hi;
                v[u] = vals[ok[u] ? i : lo];
                if (nulls != nullptr && ok[u] && ((nulls[i >> 6] >> (i & 63)) & 1)) ok[u] = false;  // NULL rows are skipped, :44
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (!ok[u]) continue;
                const A a = (A)v[u];
                if (IS_FLOAT) {
                    mn = (A)fmin((double)mn, (double)a);  // f64::min / max ignore a NaN operand
                    mx = (A)fmax((double)mx, (double)a);
                } else {
                    mn = a < mn ? a : mn;
                    mx = a > mx ? a : mx;
                }
                cnt++;
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const A omn = __shfl_xor(mn, o), omx = __shfl_xor(mx, o);
            const uint64_t oc = __shfl_xor(cnt, o);
            if (IS_FLOAT) {
                mn = (A)fmin((double)mn, (double)omn);
                mx = (A)fmax((double)mx, (double)omx);
            } else {
                mn = omn < mn ? omn : mn;
                mx = omx > mx ? omx : mx;
            }
            cnt += oc;
        }
        if (lane == 0) {
            out_min[c] = mn;
            out_max[c] = mx;
            out_nn[c] = cnt;
        }
    }
}

}  // namespace ott

using namespace ott;

extern "C" {

int ott_store_add_column(ott_store* s, uint32_t dtype, const void* values_host, const uint64_t* nulls, uint64_t n,
                         uint32_t* out_column_id) {
    if (!s || !out_column_id) return fail(OTT_ERR_INVALID, "ott_store_add_column: NULL argument");
    if (s->multi) return multi_add_column(s, dtype, values_host, nulls, n, out_column_id);
    size_t esz;
    switch (dtype) {
        case OTT_DT_INT32: case OTT_DT_FLOAT32: esz = 4; break;
        case OTT_DT_INT64: case OTT_DT_FLOAT64: case OTT_DT_DATETIME: esz = 8; break;
        default: return fail(OTT_ERR_INVALID, "ott_store_add_column: only numeric / datetime columns live on the GPU");
    }
    ott::host::ExclusiveLock wr(s->rw);  // no query is running on any context
    std::lock_guard<std::mutex> g(s->mu);
    {
        const int rcf = store_flush_locked(s);  // rows of small appends still staged on the host
        if (rcf) return rcf;
    }
    if (n != s->n) return fail(OTT_ERR_INVALID, "ott_store_add_column: column length does not match the store length");
    if (n && !values_host) return fail(OTT_ERR_INVALID, "ott_store_add_column: values is NULL");
    OTT_HIP(use_device(s));
    Column c;
    c.dtype = dtype;
    c.n = n;
    c.d_vals = nullptr;
    c.d_nulls = nullptr;
    const size_t words = (size_t)((n + 63) / 64);
    if (n) {
        OTT_HIP(hipMalloc(&c.d_vals, n * esz));
        OTT_HIP(hipMemcpy(c.d_vals, values_host, n * esz, hipMemcpyHostToDevice));
        if (nulls) {
            OTT_HIP(hipMalloc((void**)&c.d_nulls, words * 8));
            OTT_HIP(hipMemcpy(c.d_nulls, nulls, words * 8, hipMemcpyHostToDevice));
        }
    }
    s->columns.push_back(c);
    *out_column_id = (uint32_t)(s->columns.size() - 1);
    return OTT_OK;
}

int ott_store_eval_row_mask(ott_store* s, const ott_leaf* leaves, uint32_t n_leaves, uint32_t n_clauses, uint64_t* out_host) {
    if (!s) return fail(OTT_ERR_INVALID, "ott_store_eval_row_mask: store is NULL");
    if (n_leaves && !leaves) return fail(OTT_ERR_INVALID, "ott_store_eval_row_mask: leaves is NULL");
    if (s->multi) return multi_eval_row_mask(s, leaves, n_leaves, n_clauses, out_host);
    (void)n_clauses;
    ott::host::ExclusiveLock wr(s->rw);  // no query is running on any context
    std::lock_guard<std::mutex> g(s->mu);
    OTT_HIP(use_device(s));
    {
        const int rcf = store_flush_locked(s);
        if (rcf) return rcf;
    }
    const uint64_t n = s->n;
    const size_t words = (size_t)((n + 63) / 64);
    std::vector<DevLeaf> dl(n_leaves);
    for (uint32_t i = 0; i < n_leaves; i++) {
        if (leaves[i].column >= s->columns.size()) return fail(OTT_ERR_INVALID, "ott_store_eval_row_mask: unknown column id");
        if (i && leaves[i].clause < leaves[i - 1].clause) return fail(OTT_ERR_INVALID, "ott_store_eval_row_mask: leaves must be grouped by clause");
        const Column& c = s->columns[leaves[i].column];
        if (c.n != n) return fail(OTT_ERR_INVALID, "ott_store_eval_row_mask: column length no longer matches the store");
        dl[i].vals = c.d_vals;
        dl[i].nulls = c.d_nulls;
        dl[i].dtype = c.dtype;
        dl[i].op = leaves[i].op;
        dl[i].clause = leaves[i].clause;
        dl[i].pad = 0;
        dl[i].lit_i64 = leaves[i].lit_i64;
        dl[i].lit_f64 = leaves[i].lit_f64;
    }
    s->evalmask_bits = 0;
    if (!n) return OTT_OK;
    int rc;
    if ((rc = s->d_evalmask.ensure(words * 8))) return rc;
    MaskParams mp;
    memset(&mp, 0, sizeof(mp));
    mp.n_leaves = n_leaves;
    mp.n_rows = n;
    mp.out = (uint64_t*)s->d_evalmask.p;
    mp.embedded = n_leaves <= MASK_EMBED;
    if (mp.embedded) {
        if (n_leaves) memcpy(mp.eleaves, dl.data(), n_leaves * sizeof(DevLeaf));
    } else {
        if ((rc = s->d_misc.ensure(n_leaves * sizeof(DevLeaf)))) return rc;
        OTT_HIP(hipMemcpyAsync(s->d_misc.p, dl.data(), n_leaves * sizeof(DevLeaf), hipMemcpyHostToDevice, s->stream));
        mp.leaves = (const DevLeaf*)s->d_misc.p;
    }
    uint64_t blocks = (words + 4 * MASK_U - 1) / (4 * MASK_U);
    if (blocks > (uint64_t)s->n_cu * 8) blocks = (uint64_t)s->n_cu * 8;
    hipLaunchKernelGGL(eval_mask_kernel, dim3((uint32_t)blocks), dim3(256), 0, s->stream, mp);
    OTT_HIP(hipGetLastError());
    if (out_host) OTT_HIP(hipMemcpyAsync(out_host, s->d_evalmask.p, words * 8, hipMemcpyDeviceToHost, s->stream));
    OTT_HIP(hipStreamSynchronize(s->stream));
    s->evalmask_bits = n;
    return OTT_OK;
}

int ott_store_zone_stats(ott_store* s, uint32_t column, uint64_t chunk_size, void* out_min, void* out_max, uint64_t* out_non_null) {
    if (!s || !out_min || !out_max || !out_non_null) return fail(OTT_ERR_INVALID, "ott_store_zone_stats: NULL argument");
    if (chunk_size == 0) return fail(OTT_ERR_INVALID, "ott_store_zone_stats: chunk_size must be > 0");
    if (s->multi) return multi_zone_stats(s, column, chunk_size, out_min, out_max, out_non_null);
    ott::host::ExclusiveLock wr(s->rw);  // no query is running on any context
    std::lock_guard<std::mutex> g(s->mu);
    if (column >= s->columns.size()) return fail(OTT_ERR_INVALID, "ott_store_zone_stats: unknown column id");
    const Column& c = s->columns[column];
    const uint64_t n = c.n, n_chunks = (n + chunk_size - 1) / chunk_size;
    if (!n_chunks) return OTT_OK;
    OTT_HIP(use_device(s));
    int rc;
    if ((rc = s->d_misc.ensure(n_chunks * 24))) return rc;
    char* base = (char*)s->d_misc.p;
    void* dmn = base;
    void* dmx = base + n_chunks * 8;
    uint64_t* dnn = (uint64_t*)(base + n_chunks * 16);
    uint64_t blocks = (n_chunks + 3) / 4;
    if (blocks > (uint64_t)s->n_cu * 8) blocks = (uint64_t)s->n_cu * 8;
    const dim3 gr((uint32_t)blocks), bl(256);
    switch (c.dtype) {
        case OTT_DT_INT32:
            hipLaunchKernelGGL((zone_stat_kernel<int32_t, int64_t, false>), gr, bl, 0, s->stream, (const int32_t*)c.d_vals, c.d_nulls, n, chunk_size,
                               n_chunks, (int64_t*)dmn, (int64_t*)dmx, dnn, INT64_MAX, INT64_MIN);
            break;
        case OTT_DT_FLOAT32:
            hipLaunchKernelGGL((zone_stat_kernel<float, double, true>), gr, bl, 0, s->stream, (const float*)c.d_vals, c.d_nulls, n, chunk_size,
                               n_chunks, (double*)dmn, (double*)dmx, dnn, (double)INFINITY, -(double)INFINITY);
            break;
        case OTT_DT_FLOAT64:
            hipLaunchKernelGGL((zone_stat_kernel<double, double, true>), gr, bl, 0, s->stream, (const double*)c.d_vals, c.d_nulls, n, chunk_size,
                               n_chunks, (double*)dmn, (double*)dmx, dnn, (double)INFINITY, -(double)INFINITY);
            break;
        default:
            hipLaunchKernelGGL((zone_stat_kernel<int64_t, int64_t, false>), gr, bl, 0, s->stream, (const int64_t*)c.d_vals, c.d_nulls, n, chunk_size,
                               n_chunks, (int64_t*)dmn, (int64_t*)dmx, dnn, INT64_MAX, INT64_MIN);
            break;
    }
    OTT_HIP(hipGetLastError());
    OTT_HIP(hipMemcpyAsync(out_min, dmn, n_chunks * 8, hipMemcpyDeviceToHost, s->stream));
    OTT_HIP(hipMemcpyAsync(out_max, dmx, n_chunks * 8, hipMemcpyDeviceToHost, s->stream));
    OTT_HIP(hipMemcpyAsync(out_non_null, dnn, n_chunks * 8, hipMemcpyDeviceToHost, s->stream));
    OTT_HIP(hipStreamSynchronize(s->stream));
    return OTT_OK;
}

}  // extern "C"

// ott_mask.hip — GPU-side build_row_mask_for_chunk (src/meta_compute.rs:194-289) for numeric
// and datetime leaves: metadata columns and their null bitmaps live in HBM; a compiled CNF
// filter (src/expr.rs:213-226: AND of clauses, each an OR of ColumnFilter leaves) is evaluated
// for every row at once and written as BitVec<usize,Lsb0> words that the scoring kernels test.
// Row predicate = plain comparison AND not-null (src/type_utils.rs:306-444, 587-736); the
// reference evaluates it per surviving chunk on the host, the result is the same bits.
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

// one wave = 64 consecutive rows = one mask word
__global__ __launch_bounds__(256) void eval_mask_kernel(const DevLeaf* __restrict__ leaves, uint32_t n_leaves, uint64_t n_rows,
                                                         uint64_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const uint64_t n_words = (n_rows + 63) / 64;
    for (uint64_t w = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); w < n_words; w += (uint64_t)gridDim.x * 4) {
        const uint64_t row = w * 64 + lane;
        const bool in = row < n_rows;
        bool all = true;      // fold(bitvec![1; len]) over clauses, src/meta_compute.rs:203
        bool any = false;     // clause_mask = bitvec![0; len]
        uint32_t cur = n_leaves ? leaves[0].clause : 0;
        for (uint32_t i = 0; i < n_leaves; i++) {
            const DevLeaf lf = leaves[i];
            if (lf.clause != cur) {
                all = all && any;
                any = false;
                cur = lf.clause;
            }
            bool sat = false;
            if (in) {
                switch (lf.dtype) {
                    case OTT_DT_INT32: sat = op_holds<int32_t>(((const int32_t*)lf.vals)[row], lf.op, (int32_t)lf.lit_i64); break;
                    case OTT_DT_FLOAT32: sat = op_holds<float>(((const float*)lf.vals)[row], lf.op, (float)lf.lit_f64); break;
                    case OTT_DT_FLOAT64: sat = op_holds<double>(((const double*)lf.vals)[row], lf.op, lf.lit_f64); break;
                    default: sat = op_holds<int64_t>(((const int64_t*)lf.vals)[row], lf.op, lf.lit_i64); break;  // Int64 / DateTime
                }
                if (lf.nulls != nullptr && ((lf.nulls[w] >> lane) & 1)) sat = false;  // NULL never satisfies
            }
            any = any || sat;
        }
        if (n_leaves) all = all && any;
        const uint64_t word = __ballot(all && in);
        if (lane == 0) out[w] = word;
    }
}

}  // namespace ott

using namespace ott;

extern "C" {

int ott_store_add_column(ott_store* s, uint32_t dtype, const void* values_host, const uint64_t* nulls, uint64_t n,
                         uint32_t* out_column_id) {
    if (!s || !out_column_id) return fail(OTT_ERR_INVALID, "ott_store_add_column: NULL argument");
    size_t esz;
    switch (dtype) {
        case OTT_DT_INT32: case OTT_DT_FLOAT32: esz = 4; break;
        case OTT_DT_INT64: case OTT_DT_FLOAT64: case OTT_DT_DATETIME: esz = 8; break;
        default: return fail(OTT_ERR_INVALID, "ott_store_add_column: only numeric / datetime columns live on the GPU");
    }
    std::lock_guard<std::mutex> g(s->mu);
    if (n != s->n) return fail(OTT_ERR_INVALID, "ott_store_add_column: column length does not match the store length");
    if (n && !values_host) return fail(OTT_ERR_INVALID, "ott_store_add_column: values is NULL");
    OTT_HIP(hipSetDevice(s->device));
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
    (void)n_clauses;
    std::lock_guard<std::mutex> g(s->mu);
    OTT_HIP(hipSetDevice(s->device));
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
    if ((rc = s->d_misc.ensure((n_leaves ? n_leaves : 1) * sizeof(DevLeaf)))) return rc;
    if (n_leaves) OTT_HIP(hipMemcpyAsync(s->d_misc.p, dl.data(), n_leaves * sizeof(DevLeaf), hipMemcpyHostToDevice, s->stream));
    uint64_t blocks = (words + 3) / 4;
    if (blocks > (uint64_t)s->n_cu * 8) blocks = (uint64_t)s->n_cu * 8;
    hipLaunchKernelGGL(eval_mask_kernel, dim3((uint32_t)blocks), dim3(256), 0, s->stream, (const DevLeaf*)s->d_misc.p, n_leaves, n,
                       (uint64_t*)s->d_evalmask.p);
    OTT_HIP(hipGetLastError());
    if (out_host) OTT_HIP(hipMemcpyAsync(out_host, s->d_evalmask.p, words * 8, hipMemcpyDeviceToHost, s->stream));
    OTT_HIP(hipStreamSynchronize(s->stream));
    s->evalmask_bits = n;
    return OTT_OK;
}

}  // extern "C"

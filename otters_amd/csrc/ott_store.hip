// ott_store.hip — device-resident VecStore: one contiguous row-major f32 matrix in HBM
// plus per-row inverse norms (src/vec.rs:338-384).  The reference keeps one heap allocation
// per chunk (src/meta.rs:203-281); here a chunk is just a row range of the one matrix.
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <thread>

#include <algorithm>

#include "ott_internal.h"

namespace ott {

static thread_local std::string g_err;

void set_error(const std::string& msg) { g_err = msg; }
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
const char* last_error() { return g_err.c_str(); }

// ---- per-store options ----------------------------------------------------------------------
namespace {
struct OptName {
    const char* name;
    int kind;  // 0 = bool, 1 = tri-state (-1 automatic / 0 / 1), 2 = non-negative int
};
// the sixteen options of the product library (see struct Options) ...
const OptName kOptNames[] = {{"tie_order", 2},       {"hi_fmt", 1},          {"hi_prebuild", 1},          {"stage_appends", 1},       {"multi_transport", 2},
                             {"multi_rebalance", 0}, {"multi_min_shard_rows", 2}, {"exact_small", 1},     {"large_k_from", 2},        {"small_sort", 1},
                             {"mfma_f32", 0},        {"no_hi_pass", 0},      {"no_batch_image", 0},       {"force_fallback", 2},      {"eps_scale_ppm", 2},
                             {"multi_fake_distinct", 0},
#ifdef OTT_MFMA_DEBUG_BUILD
                             // ... and, in the diagnostic build only, kernel tuning, timing ablations and every fallback bit by its own name
                             {"mfma_wg", 2},         {"mfma_growth", 2},     {"mfma_debug", 0},           {"mfma_abl", 2},            {"hi_tmin", 2},
                             {"mfma_no_dense", 0},   {"mfma_coop", 1},       {"mfma_spec", 1},            {"large_k_pre", 1},         {"merge_walk", 0},
                             {"merge_rank1", 1},
#endif
};
static_assert(sizeof(kOptNames) / sizeof(kOptNames[0]) <= 16
#ifdef OTT_MFMA_DEBUG_BUILD
                                                            + 11
#endif
              , "the product library's option table stays at sixteen entries");
}  // namespace

int option_set(Options& o, const char* name, long long v) {
    if (!name) return -1;
    const std::string n(name);
    auto tri = [&](int& dst) { if (v < -1 || v > 1) return -1; dst = (int)v; return 0; };
    auto flag = [&](bool& dst) { if (v < 0 || v > 1) return -1; dst = v != 0; return 0; };
    if (n == "exact_small") {  // (1, round 2's one-wave variant: diagnostic build only)
#ifdef OTT_MFMA_DEBUG_BUILD
        if (v < -1 || v > 2) return -1;
#else
        if (v < -1 || v > 2 || v == 1) return -1;
#endif
        o.exact_small = (int)v;
        return 0;
    }
    if (n == "force_fallback") {
        if (v < 0 || v > 127) return -1;
        o.force_fallback = (int)v;
        o.merge_walk = (v & 1) != 0;
        o.merge_rank1 = (v & 2) ? 0 : -1;
        o.mfma_coop = (v & 4) ? 0 : -1;
        o.large_k_pre = (v & 8) ? 0 : -1;
        o.mfma_no_dense = (v & 16) != 0;
        o.mfma_spec = (v & 32) ? 0 : -1;
        return 0;
    }
    if (n == "mfma_f32") return flag(o.mfma_f32);
    if (n == "no_hi_pass") return flag(o.no_hi_pass);
    if (n == "no_batch_image") return flag(o.no_batch_image);
    if (n == "hi_fmt") { if (v < -1 || v > 2) return -1; o.hi_fmt = (int)v; return 0; }
    if (n == "small_sort") return tri(o.small_sort);
#ifdef OTT_MFMA_DEBUG_BUILD
    if (n == "mfma_coop") return tri(o.mfma_coop);
    if (n == "mfma_spec") return tri(o.mfma_spec);
    if (n == "mfma_no_dense") return flag(o.mfma_no_dense);
    if (n == "mfma_debug") return flag(o.mfma_debug);
    if (n == "merge_walk") return flag(o.merge_walk);
    if (n == "merge_rank1") return tri(o.merge_rank1);
    if (n == "large_k_pre") return tri(o.large_k_pre);
    if (n == "hi_tmin") { if (v < 0 || v > 512) return -1; o.hi_tmin = (int)v; return 0; }
    if (n == "mfma_abl") { if (v < 0 || v > 63) return -1; o.mfma_abl = (int)v; return 0; }
    if (n == "mfma_wg") { if (v < 0 || v > 8) return -1; o.mfma_wg = (int)v; return 0; }
    if (n == "mfma_growth") { if (v != 0 && (v < 2 || v > 64)) return -1; o.mfma_growth = v ? (int)v : 8; return 0; }
#endif
    if (n == "stage_appends") return tri(o.stage_appends);
    if (n == "hi_prebuild") return tri(o.hi_prebuild);
    if (n == "large_k_from") { if (v < 0 || v > 512) return -1; o.large_k_from = (int)v; return 0; }
    if (n == "eps_scale_ppm") { if (v < 1 || v > 1000000) return -1; o.eps_scale_ppm = (int)v; return 0; }
    if (n == "multi_transport") { if (v < 0 || v > 2) return -1; o.multi_transport = (int)v; return 0; }
    if (n == "multi_fake_distinct") return flag(o.multi_fake_distinct);
    if (n == "multi_rebalance") { if (v < 0 || v > 1) return -1; o.multi_rebalance = (int)v; return 0; }
    if (n == "multi_min_shard_rows") { if (v < 0 || v > 0x7FFFFFFF) return -1; o.multi_min_shard_rows = (int)v; return 0; }
    if (n == "tie_order") { if (v < 0 || v > 2) return -1; o.tie_order = (int)v; return 0; }
    return -1;
}

// OTT_<NAME>=<integer> for every option; a variable that is set but empty counts as 1 (the round-1 knobs were presence tests)
void options_from_env(Options& o) {
    for (const OptName& on : kOptNames) {
        std::string var = "OTT_";
        for (const char* c = on.name; *c; c++) var.push_back((char)(*c >= 'a' && *c <= 'z' ? *c - 32 : *c));
        const char* ev = getenv(var.c_str());
        if (!ev) continue;
        char* end = nullptr;
        long long v = strtoll(ev, &end, 10);
        if (end == ev) v = 1;
        (void)option_set(o, on.name, v);  // an out-of-range value leaves the default
    }
}

int DevBuf::ensure(size_t bytes) {
    if (bytes <= cap && p) return OTT_OK;
    size_t want = bytes < 256 ? 256 : bytes;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    OTT_HIP(hipMalloc(&p, want));
    cap = want;
    return OTT_OK;
}
void DevBuf::release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}
int PinBuf::ensure(size_t bytes) {
    if (bytes <= cap && p) return OTT_OK;
    size_t want = bytes < 4096 ? 4096 : bytes;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    OTT_HIP(hipHostMalloc(&p, want, hipHostMallocDefault));
    cap = want;
    return OTT_OK;
}
void PinBuf::release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
}

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------

// inverse norms in the reference's order (src/vec.rs:365-367): sequential sum of x*x, sqrt,
// 1/norm (0 for a zero norm).  Same lane = row / LDS-transpose scheme as the scorer so the
// loads stay coalesced: a wave stages 64 rows x 128 B per step.
constexpr int NKC = 32;
__global__ __launch_bounds__(256) void inv_norm_kernel(const float* __restrict__ rows, uint32_t ld, uint32_t dim,
                                                        uint64_t first, uint64_t n, float* __restrict__ inv, uint8_t* __restrict__ flag) {
    __shared__ __attribute__((aligned(16))) float smem[4 * 64 * NKC];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* st = smem + wave * 64 * NKC;
    const uint64_t n_tiles = (n + 63) / 64;
    const int lrow = lane >> 3, lslot = lane & 7, sw = (lane >> 1) & 7;
    const uint32_t nstages = (ld + NKC - 1) / NKC;
    for (uint64_t t = (uint64_t)blockIdx.x * 4 + wave; t < n_tiles; t += (uint64_t)gridDim.x * 4) {
        const uint64_t row0 = first + t * 64;
        const uint32_t cnt = (n - t * 64) < 64 ? (uint32_t)(n - t * 64) : 64u;
        float s = 0.0f;
        bool nonzero = false;  // any element != 0 (a row of tiny values can have a norm that underflows to 0)
        // branch-free staging like exact_kernel's: every load is issued (rows past a short tile's end clamped to its last
        // row, a column group past `ld` re-reads column 0), out-of-range values are zeroed on their way into LDS; the next
        // stage's loads are in flight while this one is summed; non-temporal (the rows were just written and are read once here)
        typedef float v4f __attribute__((ext_vector_type(4)));
        const float* rp[8];
        bool rok[8];
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const uint32_t row = 8 * m + lrow;
            rok[m] = row < cnt;
            rp[m] = rows + (row0 + (rok[m] ? row : cnt - 1)) * (uint64_t)ld;
        }
        v4f R[8];
        auto load_stage = [&](uint32_t sg) {
            const uint32_t col = sg * NKC + lslot * 4;
            const uint32_t c = col < ld ? col : 0u;
#pragma unroll
            for (int m = 0; m < 8; m++) R[m] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(rp[m] + c));
        };
        load_stage(0);
        for (uint32_t sg = 0; sg < nstages; sg++) {
            const bool cok = sg * NKC + lslot * 4 < ld;
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const uint32_t row = 8 * m + lrow;
                const bool ok = rok[m] & cok;
                const v4f v = R[m];
                *reinterpret_cast<float4*>(st + row * NKC + ((lslot ^ ((row >> 1) & 7)) << 2)) =
                    make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (sg + 1 < nstages) load_stage(sg + 1);
#pragma unroll
            for (int j = 0; j < NKC / 4; j++) {
                const float4 a = *reinterpret_cast<const float4*>(st + lane * NKC + ((j ^ sw) << 2));
                const float x[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                for (int l = 0; l < 4; l++)
                    if (sg * NKC + 4 * j + l < dim) {
                        s = __fadd_rn(s, __fmul_rn(x[l], x[l]));
                        nonzero |= x[l] != 0.0f;
                    }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if ((uint32_t)lane < cnt) {
            const float norm = sqrtf(s);  // correctly rounded (-fhip-fp32-correctly-rounded-divide-sqrt, the default); __fsqrt_rn lowers to the raw 1-ulp v_sqrt_f32
            inv[row0 + lane] = norm != 0.0f ? 1.0f / norm : 0.0f;
            // rows whose norm is inf / NaN / astronomically large break the error bound the MFMA path certifies with:
            // they are flagged and always re-scored exactly there (the exact path needs no flag)
            // ... and rows whose norm is tiny but not zero (below 1e-18: their squares underflow, and the bf16 split may flush
            // their elements): the bound is relative to the norms, so such rows are always re-scored exactly too
            flag[row0 + lane] = (norm <= 1e18f && ((norm == 0.0f && !nonzero) || norm >= 1e-18f)) ? 0 : 1;
        }
    }
}

// synthetic rows: uniform [-1,1), bit-identical to oracle otto_rand_elem
__device__ __forceinline__ float rand_elem(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    const uint32_t u = (uint32_t)(z >> 40);
    return __fsub_rn(__fmul_rn((float)u, 1.0f / 8388608.0f), 1.0f);
}

__global__ __launch_bounds__(256) void rand_fill_kernel(float* __restrict__ rows, uint32_t ld, uint32_t dim, uint64_t first,
                                                         uint64_t n, uint64_t global_first, uint64_t seed) {
    // one thread per 4 columns of the padded row
    const uint32_t ld4 = ld / 4;
    const uint64_t total = n * ld4;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / ld4;
        const uint32_t c = (uint32_t)(i - r * ld4) * 4;
        float v[4];
#pragma unroll
        for (int l = 0; l < 4; l++) v[l] = (c + l < dim) ? rand_elem(seed, (global_first + r) * dim + c + l) : 0.0f;
        *reinterpret_cast<float4*>(rows + (first + r) * (uint64_t)ld + c) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// Clustered / anisotropic synthetic rows (what embedding corpora look like, and what the batch path's first candidate pass
// may fail to certify): row r belongs to cluster hash(r) % n_clusters and is  centre[cluster][c] + (spread * u(r, c)) * w(c)
// with u uniform [-1,1), w(c) = 1 / (1 + aniso * c / dim) (aniso = 0: the same spread in every dimension).  Counter-based
// and bit-identical to the oracle's otto_clustered_elem, so any row can be rebuilt on the host.
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ float clustered_elem(uint64_t seed, uint64_t row, uint32_t c, uint32_t dim, uint32_t n_clusters, float spread, float aniso) {
    const uint64_t cl = mix64(seed + 0xC1057E25ull + 0x9E3779B97F4A7C15ull * (row + 1)) % n_clusters;
    const float centre = rand_elem(seed + 0x5EEDull, cl * dim + c);
    const float u = rand_elem(seed, row * dim + c);
    const float w = __fdiv_rn(1.0f, __fadd_rn(1.0f, __fmul_rn(aniso, __fdiv_rn((float)c, (float)dim))));
    return __fadd_rn(centre, __fmul_rn(__fmul_rn(spread, u), w));
}

__global__ __launch_bounds__(256) void clustered_fill_kernel(float* __restrict__ rows, uint32_t ld, uint32_t dim, uint64_t first, uint64_t n,
                                                              uint64_t global_first, uint64_t seed, uint32_t n_clusters, float spread, float aniso) {
    const uint32_t ld4 = ld / 4;
    const uint64_t total = n * ld4;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / ld4;
        const uint32_t c = (uint32_t)(i - r * ld4) * 4;
        float v[4];
#pragma unroll
        for (int l = 0; l < 4; l++) v[l] = (c + l < dim) ? clustered_elem(seed, global_first + r, c + l, dim, n_clusters, spread, aniso) : 0.0f;
        *reinterpret_cast<float4*>(rows + (first + r) * (uint64_t)ld + c) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// smallest non-zero inverse norm (positive floats order like their bit patterns)
__global__ __launch_bounds__(256) void min_pos_inv_kernel(const float* __restrict__ inv, uint64_t first, uint64_t n, uint32_t* out) {
    uint32_t best = 0x7F800000u;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t b = __float_as_uint(inv[first + i]);
        if (b != 0 && b < best) best = b;  // zero rows (inv = 0) score exactly 0 on every path
    }
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t other = __shfl_xor(best, o);
        best = other < best ? other : best;
    }
    if ((threadIdx.x & 63) == 0) atomicMin(out, best);
}

int update_min_pos_inv(ott_store* s, uint64_t first_row, uint64_t n_rows) {
    if (!n_rows) return OTT_OK;
    int rc = s->d_minpos.ensure(4);
    if (rc) return rc;
    const uint32_t init = 0x7F800000u;
    OTT_HIP(hipMemcpyAsync(s->d_minpos.p, &init, 4, hipMemcpyHostToDevice, s->stream));
    uint64_t blocks = (n_rows + 255) / 256;
    if (blocks > (uint64_t)s->n_cu * 8) blocks = (uint64_t)s->n_cu * 8;
    hipLaunchKernelGGL(min_pos_inv_kernel, dim3((uint32_t)blocks), dim3(256), 0, s->stream, s->d_inv, first_row, n_rows,
                       (uint32_t*)s->d_minpos.p);
    OTT_HIP(hipGetLastError());
    uint32_t got = init;
    OTT_HIP(hipMemcpyAsync(&got, s->d_minpos.p, 4, hipMemcpyDeviceToHost, s->stream));
    OTT_HIP(hipStreamSynchronize(s->stream));
    float f;
    memcpy(&f, &got, 4);
    if (f < s->min_pos_inv) s->min_pos_inv = f;
    return OTT_OK;
}

int launch_inv_norms(ott_store* s, uint64_t first_row, uint64_t n_rows) {
    if (!n_rows) return OTT_OK;
    uint64_t tiles = (n_rows + 63) / 64;
    uint64_t blocks = (tiles + 3) / 4;
    uint64_t cap = (uint64_t)s->n_cu * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(inv_norm_kernel, dim3((uint32_t)blocks), dim3(256), 0, s->stream, s->d_rows, s->ld, s->dim, first_row,
                       n_rows, s->d_inv, s->d_flag);
    OTT_HIP(hipGetLastError());
    return OTT_OK;
}

int launch_rand_fill(ott_store* s, uint64_t first_row, uint64_t n_rows, uint64_t seed) {
    if (!n_rows) return OTT_OK;
    hipLaunchKernelGGL(rand_fill_kernel, dim3((uint32_t)s->n_cu * 8), dim3(256), 0, s->stream, s->d_rows, s->ld, s->dim,
                       first_row, n_rows, s->base_offset + first_row, seed);
    OTT_HIP(hipGetLastError());
    return OTT_OK;
}

__global__ void clear_flag_bit_kernel(uint8_t* flag, uint64_t n, uint8_t mask);  // below

// (re)allocate rows / inv_norms / row flags for `ncap` rows, keeping the first s->n rows
static int realloc_store(ott_store* s, uint64_t ncap) {
    if (ncap > 0xFFFFFFF0ull) return fail(OTT_ERR_UNSUPPORTED, "a store holds at most 2^32-16 rows per GPU");
    float* nrows = nullptr;
    float* ninv = nullptr;
    uint8_t* nflag = nullptr;
    hipError_t e = hipSuccess;
    for (int attempt = 0; attempt < 2; attempt++) {
        e = hipMalloc((void**)&nrows, ncap * s->ld * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void**)&ninv, ncap * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void**)&nflag, ncap);
        if (e == hipSuccess) break;
        if (nrows) (void)hipFree(nrows);
        if (ninv) (void)hipFree(ninv);
        if (nflag) (void)hipFree(nflag);
        nrows = ninv = nullptr;
        nflag = nullptr;
        (void)hipGetLastError();  // reported here: the store stays as it was, and the next launch check must not see this again
        // The store's own copies of the corpus for the batch path (int8 plane, 16-bit plane, split image) are dropped by a
        // reallocation anyway: when the new buffers do not fit NEXT TO them, they go first and the allocation is tried once more
        // (a store that grows without a plan must not fail because the background builder took a quarter of the free memory)
        std::lock_guard<std::mutex> g(s->img_mu);
        if (attempt == 1 || e != hipErrorOutOfMemory || (!s->d_img && !s->d_imgh && !s->d_img8)) break;
        OTT_HIP(hipStreamSynchronize(s->stream));
        if (s->d_img) (void)hipFree(s->d_img);
        if (s->d_imgh) (void)hipFree(s->d_imgh);
        if (s->d_img8) (void)hipFree(s->d_img8);
        if (s->d_img8_scale) (void)hipFree(s->d_img8_scale);
        s->d_img = nullptr;
        s->d_imgh = nullptr;
        s->d_img8 = nullptr;
        s->d_img8_scale = nullptr;
        s->img_rows = s->img_cap = 0;
        s->imgh_rows = 0;
        s->img8_rows = 0;
    }
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? OTT_ERR_OOM : OTT_ERR_HIP, std::string("hipMalloc(store): ") + hipGetErrorString(e));
    if (s->n) {
        OTT_HIP(hipMemcpyAsync(nrows, s->d_rows, s->n * s->ld * sizeof(float), hipMemcpyDeviceToDevice, s->stream));
        OTT_HIP(hipMemcpyAsync(ninv, s->d_inv, s->n * sizeof(float), hipMemcpyDeviceToDevice, s->stream));
        OTT_HIP(hipMemcpyAsync(nflag, s->d_flag, s->n, hipMemcpyDeviceToDevice, s->stream));
    }
    if (s->ld != s->dim)  // padding columns must be zero
        OTT_HIP(hipMemsetAsync(nrows + s->n * s->ld, 0, (ncap - s->n) * s->ld * sizeof(float), s->stream));
    OTT_HIP(hipStreamSynchronize(s->stream));
    if (s->d_rows) (void)hipFree(s->d_rows);
    if (s->d_inv) (void)hipFree(s->d_inv);
    if (s->d_flag) (void)hipFree(s->d_flag);
    s->d_rows = nrows;
    s->d_inv = ninv;
    s->d_flag = nflag;
    s->cap = ncap;
    if (s->d_img) (void)hipFree(s->d_img);  // the batch image is rebuilt lazily at the new capacity
    s->d_img = nullptr;
    s->img_rows = s->img_cap = 0;
    if (s->d_imgh) (void)hipFree(s->d_imgh);
    s->d_imgh = nullptr;
    s->imgh_rows = 0;
    if (s->d_img8) (void)hipFree(s->d_img8);
    if (s->d_img8_scale) (void)hipFree(s->d_img8_scale);
    s->d_img8 = nullptr;
    s->d_img8_scale = nullptr;
    s->img8_rows = 0;
    return OTT_OK;
}

// A shard of a multi-GPU store takes over buffers the relayout filled (rows [0, n) valid, capacity `cap`): the old ones are
// freed, the 16-bit copies of the corpus are dropped (rebuilt lazily), the hi plane's per-row marks are cleared, the smallest
// inverse norm is measured again, the evaluated row mask is forgotten.  The caller holds the multi store exclusively.
int store_adopt(ott_store* s, float* rows, float* inv, uint8_t* flag, uint64_t n, uint64_t cap) {
    // the shard's own lock too: its background plane builder reads the rows under it (shared) and must be out before they go
    ott::host::ExclusiveLock wr(s->rw);
    std::lock_guard<std::mutex> g(s->mu);
    OTT_HIP(use_device(s));
    OTT_HIP(hipStreamSynchronize(s->stream));
    if (s->d_rows) (void)hipFree(s->d_rows);
    if (s->d_inv) (void)hipFree(s->d_inv);
    if (s->d_flag) (void)hipFree(s->d_flag);
    s->d_rows = rows;
    s->d_inv = inv;
    s->d_flag = flag;
    s->n = n;
    s->cap = cap;
    {
        std::lock_guard<std::mutex> g(s->img_mu);
        if (s->d_img) (void)hipFree(s->d_img);
        s->d_img = nullptr;
        s->img_rows = s->img_cap = 0;
        if (s->d_imgh) (void)hipFree(s->d_imgh);
        s->d_imgh = nullptr;
        s->imgh_rows = 0;
        if (s->d_img8) (void)hipFree(s->d_img8);
        if (s->d_img8_scale) (void)hipFree(s->d_img8_scale);
        s->d_img8 = nullptr;
        s->d_img8_scale = nullptr;
        s->img8_rows = 0;
    }
    s->evalmask_bits = 0;
    s->min_pos_inv = __builtin_inff();
    if (!n) return OTT_OK;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((n + 255) / 256, (uint64_t)s->n_cu * 8);
    hipLaunchKernelGGL(clear_flag_bit_kernel, dim3(grid), dim3(256), 0, s->stream, s->d_flag, n, (uint8_t)0xF9);  // the planes' marks (bits 1, 2)
    OTT_HIP(hipGetLastError());
    const int rc = update_min_pos_inv(s, 0, n);
    kick_plane_build(s);
    return rc;
}

// A store that grows without a plan doubles (round 4: hipMalloc + hipFree of multi-GB buffers cost ~80 ms a pair on this part —
// at 1.5x a 30-GB store built from 100k-row appends spent 1.9 of its 2.4 s in twelve of them), and falls back to smaller steps
// when the doubled size does not fit next to the old buffer: 1.25x, then exactly what is needed.
static int grow(ott_store* s, uint64_t need) {
    if (need <= s->cap) return OTT_OK;
    const uint64_t base = s->cap ? s->cap : 1024;
    uint64_t twice = base;
    while (twice < need) twice = twice * 2 + 1024;
    uint64_t quarter = base;
    while (quarter < need) quarter = quarter + quarter / 4 + 1024;
    int rc = OTT_OK;
    for (uint64_t ncap : {twice, quarter, need}) {
        if (ncap > 0xFFFFFFF0ull && need <= 0xFFFFFFF0ull) ncap = 0xFFFFFFF0ull;  // (the per-GPU row limit is not a reason to refuse a size that fits)
        rc = realloc_store(s, ncap);
        if (rc != OTT_ERR_OOM) return rc;
    }
    return rc;
}

// rows [s->n, s->n + n_rows) from a host buffer: copy, inverse norms, smallest inverse norm (one wait)
static int append_host_locked(ott_store* s, const float* rows_host, uint64_t n_rows) {
    int rc = grow(s, s->n + n_rows);
    if (rc) return rc;
    OTT_HIP(hipMemcpy2DAsync(s->d_rows + s->n * s->ld, (size_t)s->ld * 4, rows_host, (size_t)s->dim * 4, (size_t)s->dim * 4,
                             n_rows, hipMemcpyHostToDevice, s->stream));
    rc = launch_inv_norms(s, s->n, n_rows);
    if (rc) return rc;
    rc = update_min_pos_inv(s, s->n, n_rows);
    if (rc) return rc;
    s->n += n_rows;
    kick_plane_build(s);
    return OTT_OK;
}

constexpr size_t PEND_BYTES = (size_t)4 << 20, PEND_SMALL = (size_t)256 << 10;

int store_flush_locked(ott_store* s) {
    if (!s->pend.count()) return OTT_OK;
    OTT_HIP(use_device(s));
    return s->pend.flush([s](const float* rows, uint64_t n) { return append_host_locked(s, rows, n); });
}

int store_flush(ott_store* s) {
    if (!s || s->multi || !s->pend.count()) return OTT_OK;
    ott::host::ExclusiveLock wr(s->rw);  // no query is running on any context
    std::lock_guard<std::mutex> g(s->mu);
    return store_flush_locked(s);
}

}  // namespace ott

namespace ott {

static ott_store* make_worker(ott_store* s) {
    ott_store* w = new ott_store();
    w->is_worker = true;
    w->device = s->device;
    w->logical = s->logical;
    if (hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking) != hipSuccess) {
        delete w;
        return nullptr;
    }
    for (auto& ev : w->ev)
        if (hipEventCreate(&ev) != hipSuccess) {
            ott_store_destroy(w);
            return nullptr;
        }
    return w;
}

// the corpus as the owner sees it now (the caller holds the owner's `rw` shared, so it cannot change underneath)
static void alias_corpus(ott_store* w, const ott_store* s) {
    w->dim = s->dim;
    w->ld = s->ld;
    w->dimq = s->dimq;
    w->n = s->n;
    w->cap = s->cap;
    w->chunk_size = s->chunk_size;
    w->base_offset = s->base_offset;
    w->reduce = s->reduce;
    w->n_cu = s->n_cu;
    w->min_pos_inv = s->min_pos_inv;
    w->opt = s->opt;
    w->d_rows = s->d_rows;
    w->d_inv = s->d_inv;
    w->d_flag = s->d_flag;
    w->d_evalmask.p = s->d_evalmask.p;
    w->d_evalmask.cap = 0;
    w->evalmask_bits = s->evalmask_bits;
}

// rows [first, first + n) -> batch image: one thread per (row, 4 floats)
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ rows, uint32_t ld, uint32_t dim, uint32_t ldi,
                                                          uint64_t first, uint64_t n, uint16_t* __restrict__ img,
                                                          const float* __restrict__ scale) {  // scale: optional per-row factor
    const uint32_t quads = ldi / 4;
    const uint64_t total = n * quads;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = first + i / quads;
        const uint32_t c = (uint32_t)(i % quads) * 4;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < ld) x = *reinterpret_cast<const float4*>(rows + r * (uint64_t)ld + c);  // ld is a multiple of 4, padded with zeros
        const float v[4] = {x.x, x.y, x.z, x.w};
        uint16_t h[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const float xe = (c + e < dim) ? (scale ? v[e] * scale[r] : v[e]) : 0.0f;
            const __bf16 hb = (__bf16)xe;
            const __bf16 lb = (__bf16)(xe - (float)hb);
            h[e] = __builtin_bit_cast(uint16_t, hb);
            l[e] = __builtin_bit_cast(uint16_t, lb);
        }
        uint16_t* dst = img + r * (uint64_t)ldi * 2 + (c / 32) * 64 + (c % 32);
        *reinterpret_cast<uint2*>(dst) = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
        *reinterpret_cast<uint2*>(dst + 32) = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
    }
}

int launch_split_rows(hipStream_t stream, const float* rows, uint32_t ld, uint32_t dim, uint32_t ldi, uint64_t n, uint16_t* out,
                      const float* scale, int n_cu) {
    const uint64_t work = n * (ldi / 4);
    const uint32_t grid = (uint32_t)std::min<uint64_t>((work + 255) / 256, (uint64_t)n_cu * 16);
    hipLaunchKernelGGL(split_rows_kernel, dim3(grid ? grid : 1), dim3(256), 0, stream, rows, ld, dim, ldi, (uint64_t)0, n, out, scale);
    OTT_HIP(hipGetLastError());
    return OTT_OK;
}

// rows [first, first + n) -> hi plane (bf16 round-to-nearest of every element), one wave per row.  Also measures what the
// rounding lost: rel = ||x - bf16(x)|| / ||x|| per row (f64 sums: the squares of a 1e-18-norm row underflow in f32), written
// to rel_out[r] (optional) and folded into *rel_max (optional; float bits, rows with flag[r] != 0 excluded — those are always
// re-scored exactly).  This measured figure, not the worst case 2^-8, is what the hi pass's certification uses.
// F16 = false: bf16 (round to nearest even) of every element.  F16 = true (round 3): IEEE half — the same two bytes carry
// 11 significant bits instead of 8, so the measured rounding loss ||v - h(v)|| / ||v|| is ~8x smaller (2.1e-4 against 1.65e-3
// on uniform rows) and the hi pass's error bound with it; the price is half's narrow exponent range, met by ONE
// power-of-two factor for all rows (`gscale`, exact; chosen in ensure_hi_plane).  A row whose elements
// then fall into half's subnormals (a norm far below the store's largest) or overflow (appended after the plane was
// scaled) simply MEASURES a large loss: rows above `rel_flag` are marked irregular (`flag_rw`, bit 1) — excluded from the
// store's maximum, always listed, always re-scored exactly — exactly like rows outside the bf16 pass's error model.
template <bool F16>
__global__ __launch_bounds__(256) void hi_rows_kernel(const float* __restrict__ rows, uint32_t ld, uint32_t dim, uint32_t ldh,
                                                       uint64_t first, uint64_t n, uint16_t* __restrict__ img,
                                                       const float* __restrict__ scale, float* __restrict__ rel_out,
                                                       uint32_t* __restrict__ rel_max, const uint8_t* flag, float gscale,
                                                       float rel_flag, uint8_t* flag_rw) {  // (flag and flag_rw may be the same array)
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wid = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (uint64_t)gridDim.x * 4;
    for (uint64_t i = wid; i < n; i += nw) {
        const uint64_t r = first + i;
        const float sc = scale ? __fmul_rn(scale[r], gscale) : gscale;  // gscale is a power of two (1 for bf16): exact
        double se = 0.0, sx = 0.0;
        for (uint32_t c = lane * 4; c < ldh; c += 256) {
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < ld) x = *reinterpret_cast<const float4*>(rows + r * (uint64_t)ld + c);  // ld is a multiple of 4, padded with zeros
            const float v[4] = {x.x, x.y, x.z, x.w};
            uint16_t h[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float xe = (c + e < dim) ? ((scale || F16) ? __fmul_rn(v[e], sc) : v[e]) : 0.0f;
                float back;
                if constexpr (F16) {
                    const _Float16 hb = (_Float16)xe;  // v_cvt_f16_f32: round to nearest even, overflow -> inf, gradual underflow
                    back = (float)hb;
                    h[e] = __builtin_bit_cast(uint16_t, hb);
                } else {
                    const __bf16 hb = (__bf16)xe;
                    back = (float)hb;
                    h[e] = __builtin_bit_cast(uint16_t, hb);
                }
                const double df = (double)xe - (double)back;
                se += df * df;
                sx += (double)xe * (double)xe;
            }
            *reinterpret_cast<uint2*>(img + r * (uint64_t)ldh + c) =
                make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            se += __shfl_xor(se, off);
            sx += __shfl_xor(sx, off);
        }
        if (lane == 0) {
            // rounded up (1 + 1e-4 covers the f64 sums and the f32 conversion); a non-finite row measures as 1 = "cannot certify"
            float rel = sx > 0.0 ? (float)(sqrt(se / sx) * 1.0001) : 0.0f;
            if (!(rel <= 1.0f)) rel = 1.0f;
            if (rel_out) rel_out[i] = rel;
            bool irregular = flag && (flag[r] & 1u);
            if (flag_rw && rel > rel_flag && !irregular) {
                flag_rw[r] = (uint8_t)(flag_rw[r] | 2u);  // bit 1: outside the HALF hi pass's error model only (see mfma_score_kernel)
                irregular = true;
                if (rel_max) atomicAdd(rel_max + 1, 1u);  // how many rows the plane's one factor does not suit
            }
            // (look first: one atomic per row on ONE address serialises — 10M rows took 113 ms; the running max settles at once)
            if (rel_max && !irregular && __float_as_uint(rel) > __hip_atomic_load(rel_max, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                atomicMax(rel_max, __float_as_uint(rel));
        }
    }
}

int launch_hi_rows(hipStream_t stream, const float* rows, uint32_t ld, uint32_t dim, uint32_t ldh, uint64_t n, uint16_t* out,
                   const float* scale, float* rel_out, int n_cu, bool f16, float gscale) {
    const uint32_t grid = (uint32_t)std::min<uint64_t>((n + 3) / 4, (uint64_t)n_cu * 8);
    if (f16)
        hipLaunchKernelGGL(hi_rows_kernel<true>, dim3(grid ? grid : 1), dim3(256), 0, stream, rows, ld, dim, ldh, (uint64_t)0, n, out, scale, rel_out,
                           (uint32_t*)nullptr, (const uint8_t*)nullptr, gscale, 2.0f, (uint8_t*)nullptr);
    else
        hipLaunchKernelGGL(hi_rows_kernel<false>, dim3(grid ? grid : 1), dim3(256), 0, stream, rows, ld, dim, ldh, (uint64_t)0, n, out, scale, rel_out,
                           (uint32_t*)nullptr, (const uint8_t*)nullptr, 1.0f, 2.0f, (uint8_t*)nullptr);
    OTT_HIP(hipGetLastError());
    return OTT_OK;
}

// smallest non-zero inverse norm over the REGULAR rows (the largest norm the half plane's factor has to accommodate)
__global__ __launch_bounds__(256) void min_regular_inv_kernel(const float* __restrict__ inv, const uint8_t* __restrict__ flag, uint64_t n, uint32_t* out) {
    uint32_t best = 0x7F800000u;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t b = __float_as_uint(inv[i]);
        if (b != 0 && b < best && !(flag[i] & 1u)) best = b;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t other = __shfl_xor(best, o);
        best = other < best ? other : best;
    }
    if ((threadIdx.x & 63) == 0) atomicMin(out, best);
}

__global__ __launch_bounds__(256) void clear_flag_bit_kernel(uint8_t* flag, uint64_t n, uint8_t mask) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) flag[i] = (uint8_t)(flag[i] & mask);
}

// rows [first, first + cnt) of the store -> its hi plane, in the plane's format (half: scaled by the plane's factor, rows that
// measure a loss above 2^-10 — five times what a row of ordinary dynamic range measures — marked irregular)
static int launch_store_hi_rows(ott_store* own, hipStream_t stream, uint64_t first, uint64_t cnt) {
    const uint32_t ldh = (own->dim + 63u) & ~63u;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((cnt + 3) / 4, (uint64_t)own->n_cu * 8);
    if (own->imgh_f16)
        hipLaunchKernelGGL(hi_rows_kernel<true>, dim3(grid), dim3(256), 0, stream, own->d_rows, own->ld, own->dim, ldh, first, cnt, own->d_imgh,
                           (const float*)nullptr, (float*)nullptr, own->d_imgh_rel, own->d_flag, own->imgh_scale, 9.765625e-4f, own->d_flag);
    else
        hipLaunchKernelGGL(hi_rows_kernel<false>, dim3(grid), dim3(256), 0, stream, own->d_rows, own->ld, own->dim, ldh, first, cnt, own->d_imgh,
                           (const float*)nullptr, (float*)nullptr, own->d_imgh_rel, own->d_flag, 1.0f, 2.0f, (uint8_t*)nullptr);
    OTT_HIP(hipGetLastError());
    return OTT_OK;
}

// the store's hi plane, built / extended on demand (see ott_internal.h); *img_out = nullptr when it is unavailable
int ensure_hi_plane(ott_store* ctx, const uint16_t** img_out, float* rel_max_out, bool* f16_out, float* scale_out) {
    *img_out = nullptr;
    if (f16_out) *f16_out = false;
    if (scale_out) *scale_out = 1.0f;
    ott_store* own = ctx->owner ? ctx->owner : ctx;
    std::lock_guard<std::mutex> g(own->img_mu);
    if (own->imgh_off || own->img_off || own->n == 0) return OTT_OK;
    const uint32_t ldh = (own->dim + 63u) & ~63u;
    if (!own->d_imgh) {
        const size_t bytes = (size_t)own->cap * ldh * 2;
        size_t free_b = 0, total_b = 0;
        if (own->opt.no_batch_image || own->opt.no_hi_pass || hipMemGetInfo(&free_b, &total_b) != hipSuccess ||
            free_b < bytes + (size_t)(2ull << 30) || hipMalloc((void**)&own->d_imgh, bytes) != hipSuccess) {
            own->d_imgh = nullptr;
            own->imgh_off = true;  // does not fit (or switched off): the batch path starts at the split pass
            (void)hipGetLastError();
            return OTT_OK;
        }
        own->imgh_rows = 0;
    }
    if (!own->d_imgh_rel) {  // [0] running max of the measured rounding loss (float bits), [1] rows the half plane's factor does not suit, [2] scratch
        OTT_HIP(hipMalloc((void**)&own->d_imgh_rel, 16));
        OTT_HIP(hipMemsetAsync(own->d_imgh_rel, 0, 16, ctx->stream));
    }
    if (own->imgh_rows == 0) {
        // format of the plane: IEEE half unless the store asks for bf16 (option hi_fmt = 0).  Half needs ONE power-of-two factor
        // for all rows, derived from the largest REGULAR row norm
        own->imgh_f16 = own->opt.hi_fmt != 0;
        own->imgh_scale = 1.0f;
        OTT_HIP(hipMemsetAsync(own->d_imgh_rel, 0, 16, ctx->stream));
        if (own->imgh_f16) {
            const uint32_t init = 0x7F800000u;
            uint32_t got = init;
            OTT_HIP(hipMemcpyAsync(own->d_imgh_rel + 2, &init, 4, hipMemcpyHostToDevice, ctx->stream));
            const uint32_t grid = (uint32_t)std::min<uint64_t>((own->n + 255) / 256, (uint64_t)own->n_cu * 8);
            hipLaunchKernelGGL(min_regular_inv_kernel, dim3(grid), dim3(256), 0, ctx->stream, own->d_inv, own->d_flag, own->n, own->d_imgh_rel + 2);
            OTT_HIP(hipGetLastError());
            OTT_HIP(hipMemcpyAsync(&got, own->d_imgh_rel + 2, 4, hipMemcpyDeviceToHost, ctx->stream));
            OTT_HIP(hipStreamSynchronize(ctx->stream));
            float min_inv;
            memcpy(&min_inv, &got, 4);
            const float max_norm = (got != init && min_inv > 0.0f) ? 1.0f / min_inv : 1.0f;
            // factor = 2^-round(log2(max_norm) / 4).  The batch path multiplies its query operands by the RECIPROCAL (so the
            // accumulators need no correction): rows then sit around max_norm^0.75 / sqrt(dim), unit-length (cosine) queries
            // around max_norm^0.25 / sqrt(dim), raw (dot / L2) queries of similar length around max_norm^1.25 / sqrt(dim) — all
            // inside half's normal range for norms from ~1e-3 to a few thousand (cosine: to ~1e6).  Outside that, bf16.
            int e = 0;
            (void)frexpf(max_norm, &e);              // max_norm = m * 2^e, m in [0.5, 1)
            own->imgh_scale = ldexpf(1.0f, -(e / 4));
            if (!(own->imgh_scale > 0.0f) || !(own->imgh_scale < __builtin_inff()) || max_norm > 1e6f || max_norm < 1e-3f) {
                own->imgh_f16 = false;
                own->imgh_scale = 1.0f;
            }
        }
    }
    if (own->imgh_rows < own->n) {
        const bool from_scratch = own->imgh_rows == 0;
        const uint64_t first = own->imgh_rows, cnt = own->n - first;
        int rch = launch_store_hi_rows(own, ctx->stream, first, cnt);
        if (rch) return rch;
        uint32_t bits[2] = {0, 0};
        OTT_HIP(hipMemcpyAsync(bits, own->d_imgh_rel, 8, hipMemcpyDeviceToHost, ctx->stream));
        OTT_HIP(hipStreamSynchronize(ctx->stream));  // published below: other contexts' streams may read it at once
        if (own->imgh_f16 && from_scratch && (uint64_t)bits[1] * 64 > cnt) {
            // more than 1 row in 64 does not fit the one factor (norms spread over many binades): half is the wrong format for
            // this store.  The marks are taken back and the plane is built again as bf16, whose exponent range is f32's
            own->imgh_f16 = false;
            own->imgh_scale = 1.0f;
            const uint32_t grid = (uint32_t)std::min<uint64_t>((own->n + 255) / 256, (uint64_t)own->n_cu * 8);
            hipLaunchKernelGGL(clear_flag_bit_kernel, dim3(grid), dim3(256), 0, ctx->stream, own->d_flag, own->n, (uint8_t)0xFD);
            OTT_HIP(hipGetLastError());
            OTT_HIP(hipMemsetAsync(own->d_imgh_rel, 0, 16, ctx->stream));
            if ((rch = launch_store_hi_rows(own, ctx->stream, first, cnt))) return rch;
            OTT_HIP(hipMemcpyAsync(bits, own->d_imgh_rel, 8, hipMemcpyDeviceToHost, ctx->stream));
            OTT_HIP(hipStreamSynchronize(ctx->stream));
        }
        memcpy(&own->imgh_rel, &bits[0], 4);
        own->imgh_rows = own->n;
    }
    *img_out = own->d_imgh;
    *rel_max_out = own->imgh_rel;
    if (f16_out) *f16_out = own->imgh_f16;
    if (scale_out) *scale_out = own->imgh_scale;
    return OTT_OK;
}

bool hi_plane_ready(ott_store* ctx) {
    ott_store* own = ctx->owner ? ctx->owner : ctx;
    std::lock_guard<std::mutex> g(own->img_mu);
    return own->d_imgh != nullptr && !own->imgh_off && !own->img_off && own->n != 0 && own->imgh_rows == own->n;
}

int ensure_batch_image(ott_store* ctx, const uint16_t** img_out) {
    *img_out = nullptr;
    ott_store* own = ctx->owner ? ctx->owner : ctx;
    std::lock_guard<std::mutex> g(own->img_mu);
    if (own->img_off || own->n == 0) return OTT_OK;
    const uint32_t ldi = (own->dim + 31u) & ~31u;
    if (!own->d_img) {
        const size_t bytes = (size_t)own->cap * ldi * 4;
        size_t free_b = 0, total_b = 0;
        if (own->opt.no_batch_image || hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < bytes + (size_t)(2ull << 30) ||
            hipMalloc((void**)&own->d_img, bytes) != hipSuccess) {
            own->d_img = nullptr;
            own->img_off = true;  // does not fit (or switched off): split in registers instead
            (void)hipGetLastError();
            return OTT_OK;
        }
        own->img_cap = own->cap;
        own->img_rows = 0;
    }
    if (own->img_rows < own->n) {
        const uint64_t first = own->img_rows, cnt = own->n - first;
        const uint64_t work = cnt * (ldi / 4);
        const uint32_t grid = (uint32_t)std::min<uint64_t>((work + 255) / 256, (uint64_t)own->n_cu * 16);
        hipLaunchKernelGGL(split_rows_kernel, dim3(grid), dim3(256), 0, ctx->stream, own->d_rows, own->ld, own->dim, ldi, first, cnt, own->d_img, nullptr);
        OTT_HIP(hipGetLastError());
        OTT_HIP(hipStreamSynchronize(ctx->stream));  // published below: other contexts' streams may read it at once
        own->img_rows = own->n;
    }
    *img_out = own->d_img;
    return OTT_OK;
}

// ---- int8 plane (round 5) --------------------------------------------------------------------------------------------------
// rows [first, first + n) -> int8, one wave per row.  Per-row scale s = max|x| / 127 (or the caller's common scale), element =
// rint(x / s) clamped to +-127.  What the rounding lost is MEASURED in f64 against the values actually stored:
// rel = ||x - s x~|| / ||x|| (rounded up), into rel_out[r] and — regular rows only — the running maximum *rel_max; a row above
// rel_flag is marked irregular (bit 2 of flag_rw) and counted in rel_max[1] instead.
constexpr float I8_REL_FLAG = 0.03125f;  // 2^-5: eight times what a row of ordinary dynamic range measures at dim 768
__global__ __launch_bounds__(256) void i8_rows_kernel(const float* __restrict__ rows, uint32_t ld, uint32_t dim, uint32_t ld8, uint64_t first, uint64_t n,
                                                       int8_t* __restrict__ img, const float* __restrict__ pre, float common_scale,
                                                       float* __restrict__ scale_out, float* __restrict__ rel_out, uint32_t* __restrict__ rel_max,
                                                       const uint8_t* flag, float rel_flag, uint8_t* flag_rw) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wid = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (uint64_t)gridDim.x * 4;
    for (uint64_t i = wid; i < n; i += nw) {
        const uint64_t r = first + i;
        const float pf = pre ? pre[r] : 1.0f;
        const float* x = rows + r * (uint64_t)ld;
        float s = common_scale;
        if (!(common_scale > 0.0f)) {
            float mx = 0.0f;
            for (uint32_t c = lane * 4; c < ld; c += 256) {
                const float4 v = *reinterpret_cast<const float4*>(x + c);  // ld is a multiple of 4, padded with zeros
                mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v.x * pf), fabsf(v.y * pf)), fmaxf(fabsf(v.z * pf), fabsf(v.w * pf))));  // (fmaxf drops a NaN operand)
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
            s = mx / 127.0f;
            if (!(s < __builtin_inff())) s = 0.0f;  // a non-finite row: flagged at append, always re-scored exactly; its plane row is zeros
        }
        const float inv_s = s > 0.0f ? 1.0f / s : 0.0f;
        double se = 0.0, sx = 0.0;
        for (uint32_t c = lane * 4; c < ld8; c += 256) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < ld) v = *reinterpret_cast<const float4*>(x + c);
            const float xe[4] = {v.x * pf, v.y * pf, v.z * pf, v.w * pf};
            int q[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float t = (c + e < dim) ? rintf(xe[e] * inv_s) : 0.0f;
                t = t == t ? fminf(fmaxf(t, -127.0f), 127.0f) : 0.0f;
                q[e] = (int)t;
                if (c + e < dim) {
                    const double df = (double)xe[e] - (double)s * (double)q[e];
                    se += df * df;
                    sx += (double)xe[e] * (double)xe[e];
                }
            }
            *reinterpret_cast<uint32_t*>(img + r * (uint64_t)ld8 + c) =
                (uint32_t)(uint8_t)q[0] | ((uint32_t)(uint8_t)q[1] << 8) | ((uint32_t)(uint8_t)q[2] << 16) | ((uint32_t)(uint8_t)q[3] << 24);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            se += __shfl_xor(se, off);
            sx += __shfl_xor(sx, off);
        }
        if (lane == 0) {
            if (scale_out) scale_out[r] = s;
            float rel = sx > 0.0 ? (float)(sqrt(se / sx) * 1.0001) : 0.0f;
            if (!(rel <= 1.0f)) rel = 1.0f;
            if (rel_out) rel_out[i] = rel;
            bool irregular = flag && (flag[r] & 1u);
            if (flag_rw && !irregular) {
                if (rel > rel_flag) {
                    flag_rw[r] = (uint8_t)(flag_rw[r] | 4u);
                    irregular = true;
                    if (rel_max) atomicAdd(rel_max + 1, 1u);
                } else if (flag_rw[r] & 4u) {
                    flag_rw[r] = (uint8_t)(flag_rw[r] & ~4u);  // a rewritten row that suits the format again
                }
            }
            if (rel_max && !irregular && __float_as_uint(rel) > __hip_atomic_load(rel_max, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                atomicMax(rel_max, __float_as_uint(rel));
        }
    }
}

int launch_i8_rows(hipStream_t stream, const float* rows, uint32_t ld, uint32_t dim, uint32_t ld8, uint64_t first, uint64_t n, int8_t* out,
                   const float* pre, float common_scale, float* scale_out, float* rel_out, uint32_t* rel_max, const uint8_t* flag, float rel_flag,
                   uint8_t* flag_rw, int n_cu) {
    if (!n) return OTT_OK;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((n + 3) / 4, (uint64_t)n_cu * 8);
    hipLaunchKernelGGL(i8_rows_kernel, dim3(grid ? grid : 1), dim3(256), 0, stream, rows, ld, dim, ld8, first, n, out, pre, common_scale, scale_out, rel_out,
                       rel_max, flag, rel_flag, flag_rw);
    OTT_HIP(hipGetLastError());
    return OTT_OK;
}

int ensure_i8_plane(ott_store* ctx, const int8_t** img_out, const float** scale_out, float* rel_max_out) {
    *img_out = nullptr;
    *scale_out = nullptr;
    ott_store* own = ctx->owner ? ctx->owner : ctx;
    std::lock_guard<std::mutex> g(own->img_mu);
    if (own->img8_off || own->img_off || own->n == 0 || !i8_wanted(own->opt) || own->dim < 8) return OTT_OK;
    const uint32_t ld8 = (own->dim + 127u) & ~127u;
    if (!own->d_img8) {
        const size_t bytes = (size_t)own->cap * ld8;
        size_t free_b = 0, total_b = 0;
        if (own->opt.no_batch_image || own->opt.no_hi_pass || hipMemGetInfo(&free_b, &total_b) != hipSuccess ||
            free_b < bytes + (size_t)own->cap * 4 + (size_t)(2ull << 30) || hipMalloc((void**)&own->d_img8, bytes) != hipSuccess) {
            own->d_img8 = nullptr;
            own->img8_off = true;  // does not fit (or switched off): the cascade starts at the hi pass
            (void)hipGetLastError();
            return OTT_OK;
        }
        if (hipMalloc((void**)&own->d_img8_scale, (size_t)own->cap * 4) != hipSuccess) {
            (void)hipFree(own->d_img8);
            own->d_img8 = nullptr;
            own->d_img8_scale = nullptr;
            own->img8_off = true;
            (void)hipGetLastError();
            return OTT_OK;
        }
        own->img8_rows = 0;
    }
    if (!own->d_img8_rel) {
        OTT_HIP(hipMalloc((void**)&own->d_img8_rel, 16));
        OTT_HIP(hipMemsetAsync(own->d_img8_rel, 0, 16, ctx->stream));
    }
    if (own->img8_rows == 0) OTT_HIP(hipMemsetAsync(own->d_img8_rel, 0, 16, ctx->stream));
    if (own->img8_rows < own->n) {
        const bool from_scratch = own->img8_rows == 0;
        const uint64_t first = own->img8_rows, cnt = own->n - first;
        int rc = launch_i8_rows(ctx->stream, own->d_rows, own->ld, own->dim, ld8, first, cnt, own->d_img8, nullptr, 0.0f, own->d_img8_scale, nullptr,
                                own->d_img8_rel, own->d_flag, I8_REL_FLAG, own->d_flag, own->n_cu);
        if (rc) return rc;
        uint32_t bits[2] = {0, 0};
        OTT_HIP(hipMemcpyAsync(bits, own->d_img8_rel, 8, hipMemcpyDeviceToHost, ctx->stream));
        OTT_HIP(hipStreamSynchronize(ctx->stream));  // published below: other contexts' streams may read it at once
        if (from_scratch && (uint64_t)bits[1] * 64 > cnt) {
            // more than 1 row in 64 does not suit int8 (heavy-tailed elements): the wrong format for this store.  The marks are
            // taken back, the plane is freed and the cascade starts at the hi pass
            const uint32_t grid = (uint32_t)std::min<uint64_t>((own->n + 255) / 256, (uint64_t)own->n_cu * 8);
            hipLaunchKernelGGL(clear_flag_bit_kernel, dim3(grid), dim3(256), 0, ctx->stream, own->d_flag, own->n, (uint8_t)0xFB);
            OTT_HIP(hipGetLastError());
            OTT_HIP(hipStreamSynchronize(ctx->stream));
            (void)hipFree(own->d_img8);
            (void)hipFree(own->d_img8_scale);
            own->d_img8 = nullptr;
            own->d_img8_scale = nullptr;
            own->img8_off = true;
            return OTT_OK;
        }
        memcpy(&own->img8_rel, &bits[0], 4);
        own->img8_rows = own->n;
    }
    *img_out = own->d_img8;
    *scale_out = own->d_img8_scale;
    *rel_max_out = own->img8_rel;
    return OTT_OK;
}

int ensure_first_plane(ott_store* ctx) {
    ott_store* own = ctx->owner ? ctx->owner : ctx;
    float rel = 0.f;
    if (i8_wanted(own->opt)) {
        const int8_t* i8 = nullptr;
        const float* i8s = nullptr;
        const int rc = ensure_i8_plane(ctx, &i8, &i8s, &rel);
        if (rc) return rc;
        bool have_hi;
        {
            std::lock_guard<std::mutex> g(own->img_mu);
            have_hi = own->d_imgh != nullptr;
        }
        if (i8 && !have_hi) return OTT_OK;  // the hi plane is built when a query first needs it
    }
    const uint16_t* img = nullptr;
    return ensure_hi_plane(ctx, &img, &rel);
}

PlaneSnapshot plane_snapshot(const ott_store* s) {
    ott_store* own = const_cast<ott_store*>(s->owner ? s->owner : s);
    std::lock_guard<std::mutex> g(own->img_mu);
    return PlaneSnapshot{own->d_img8 != nullptr, own->img8_off, own->d_imgh != nullptr, own->imgh_f16, own->imgh_off, own->img_off,
                         own->img8_rows, own->imgh_rows};
}

bool first_plane_ready(ott_store* ctx) {
    ott_store* own = ctx->owner ? ctx->owner : ctx;
    bool i8_on;
    {
        std::lock_guard<std::mutex> g(own->img_mu);
        i8_on = i8_wanted(own->opt) && !own->img8_off && own->dim >= 8;
    }
    return i8_on ? i8_plane_ready(ctx) : hi_plane_ready(ctx);
}

bool i8_plane_ready(ott_store* ctx) {
    ott_store* own = ctx->owner ? ctx->owner : ctx;
    std::lock_guard<std::mutex> g(own->img_mu);
    return own->d_img8 != nullptr && !own->img8_off && !own->img_off && own->n != 0 && own->img8_rows == own->n;
}

// the store's own context when it is free, else a worker context that aliases the corpus (ott::host::ContextPool)
ott_store* ctx_acquire(ott_store* s) {
    return s->pool.acquire(
        s, OTT_MAX_WORKERS,
        [s]() -> ott_store* {
            (void)use_device(s);
            ott_store* w = make_worker(s);
            if (w) w->owner = s;
            return w;
        },
        [s](ott_store* w) { alias_corpus(w, s); });
}

void ctx_release(ott_store* w) { (w->owner ? w->owner : w)->pool.release(w); }

}  // namespace ott

// The hi plane off the first batch's critical path (round 4).  A first 256-query batch on a fresh 10M x 768 store took 19 ms:
// 15 of them the allocation and conversion of the 16-bit plane.  With option hi_prebuild (automatic for stores of 262144 rows
// and more) every append ends by waking this thread, which takes the store like a query does (shared), converts the rows that
// are new (~10 ms per 30 GB, on a context of its own) and goes back to sleep; a batch that arrives while it is at work waits for
// it on the plane's mutex exactly as it would have built the plane itself.  Results never depend on it.
namespace ott {

// one run of the background builder (ott::host::QuietWorker calls it once the appends have been quiet for 20 ms: a store loaded
// in pieces is not converted piece by piece — each conversion holds the store shared, i.e. the next append waits for it, and a
// growing store's reallocation drops the plane again: a 30-GB load in 100k-row pieces went from 2.4 to 4.0 s without the wait)
static void plane_builder_run(ott_store* s) {
    ott::host::SharedLock rd(s->rw);
    if (use_device(s) != hipSuccess) return;
    if (s->opt.hi_prebuild < 0) {  // automatic: only while the plane is a modest share of what is free
        size_t free_b = 0, total_b = 0;
        const PlaneSnapshot ps = plane_snapshot(s);
        const bool i8 = i8_wanted(s->opt) && !ps.i8_off;
        const size_t bytes = i8 ? (size_t)s->cap * ((s->dim + 127u) & ~127u) : (size_t)s->cap * ((s->dim + 63u) & ~63u) * 2;
        if (!(i8 ? ps.have_i8 : ps.have_hi) && (hipMemGetInfo(&free_b, &total_b) != hipSuccess || bytes > free_b / 4)) return;
    }
    ott_store* ctx = ctx_acquire(s);
    mfma_warm(ctx->stream, s->device);  // the batch path's kernels onto the device first: the first batch of a process paid 10-15 ms for that
    (void)ensure_first_plane(ctx);  // (a failure leaves the plane to the first batch, as before)
    ctx_release(ctx);
    (void)hipGetLastError();
}

void kick_plane_build(ott_store* s) {
    if (s->is_worker || s->multi) return;
    const int pol = s->opt.hi_prebuild;
    const PlaneSnapshot ps = plane_snapshot(s);
    if (pol == 0 || s->opt.no_hi_pass || s->opt.no_batch_image || s->opt.mfma_f32 || ps.hi_off || ps.img_off) return;
    if (pol < 0 && s->n < 262144) return;
    if (s->dim < 8) return;
    {
        const bool i8 = i8_wanted(s->opt) && !ps.i8_off;
        const bool i8_stale = i8 && ps.i8_rows < s->n, hi_stale = (!i8 || ps.have_hi) && ps.hi_rows < s->n;
        if (!i8_stale && !hi_stale) return;
    }
    if (!s->builder) s->builder = new ott::host::QuietWorker([s] { plane_builder_run(s); }, std::chrono::milliseconds(20));
    s->builder->kick();
}

}  // namespace ott

using namespace ott;

extern "C" {

int ott_abi_version(void) { return OTT_ABI_VERSION; }
const char* ott_last_error(void) { return ott::last_error(); }

int ott_device_count(int* out) {
    if (!out) return fail(OTT_ERR_INVALID, "ott_device_count: out is NULL");
    int n = 0;
    OTT_HIP(hipGetDeviceCount(&n));
    *out = n;
    return OTT_OK;
}

int ott_store_create(uint32_t dim, int device, ott_store** out) { return ott::store_create(dim, device, device, out); }

}  // extern "C"

int ott::store_create(uint32_t dim, int device, int logical, ott_store** out) {
    if (!out) return fail(OTT_ERR_INVALID, "ott_store_create: out is NULL");
    *out = nullptr;
    if (dim == 0) return fail(OTT_ERR_INVALID, "ott_store_create: dim must be > 0");
    OTT_HIP(use_device_raw(device, logical));
    hipDeviceProp_t prop;
    OTT_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(OTT_ERR_UNSUPPORTED, std::string("libotters_hip is built for gfx950 (MI355X) only; device is ") + prop.gcnArchName);
    ott_store* s = new ott_store();
    s->device = device;
    s->logical = logical;
    s->dim = dim;
    s->ld = (dim + 3u) & ~3u;
    s->dimq = (dim + 7u) & ~7u;
    s->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    options_from_env(s->opt);  // the ONLY place the library reads the environment
    hipError_t e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete s;
        return fail(OTT_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    }
    for (auto& ev : s->ev) {
        e = hipEventCreate(&ev);
        if (e != hipSuccess) {
            delete s;
            return fail(OTT_ERR_HIP, std::string("hipEventCreate: ") + hipGetErrorString(e));
        }
    }
    *out = s;
    return OTT_OK;
}

extern "C" {

int ott_store_destroy(ott_store* s) {
    if (!s) return OTT_OK;
    if (s->multi) return multi_destroy(s);
    if (s->builder) {  // the background plane builder finishes what it is at, then goes (~QuietWorker stops and joins)
        delete s->builder;
        s->builder = nullptr;
    }
    (void)use_device(s);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    for (ott_store* w : s->pool.workers) ott_store_destroy(w);
    s->pool.workers.clear();
    if (s->is_worker) {  // a worker only aliases the corpus and the evaluated row mask
        s->d_rows = nullptr;
        s->d_inv = nullptr;
        s->d_flag = nullptr;
        s->d_evalmask.p = nullptr;
        s->d_evalmask.cap = 0;
    }
    if (s->d_rows) (void)hipFree(s->d_rows);
    if (s->d_inv) (void)hipFree(s->d_inv);
    if (s->d_flag) (void)hipFree(s->d_flag);
    if (s->d_img && !s->is_worker) (void)hipFree(s->d_img);
    if (s->d_imgh && !s->is_worker) (void)hipFree(s->d_imgh);
    if (s->d_imgh_rel && !s->is_worker) (void)hipFree(s->d_imgh_rel);
    if (!s->is_worker) {
        if (s->d_img8) (void)hipFree(s->d_img8);
        if (s->d_img8_scale) (void)hipFree(s->d_img8_scale);
        if (s->d_img8_rel) (void)hipFree(s->d_img8_rel);
    }
    for (ott::DevBuf* b : {&s->d_queries, &s->d_qinv, &s->d_rowmask, &s->d_runs, &s->d_prefix, &s->d_lists, &s->d_lists2, &s->d_hits,
                           &s->d_count, &s->d_cand, &s->d_misc, &s->d_evalmask, &s->d_minpos, &s->m_Q, &s->m_qinv, &s->m_qnorm,
                           &s->m_tau, &s->m_cntA, &s->m_cntB, &s->m_candA, &s->m_candB, &s->m_over, &s->m_out, &s->m_outcnt,
                           &s->m_uncert, &s->m_prefix, &s->x_send, &s->x_recv, &s->l_keysA, &s->l_keysB, &s->l_qA, &s->l_qB, &s->l_tmp, &s->l_cursor, &s->l_hist, &s->l_gate, &s->l_ctl})
        b->release();
    s->h_stage.release();
    s->h_hits.release();
    s->h_hdr.release();
    s->h_pend.release();
    for (auto& c : s->columns) {
        if (c.d_vals) (void)hipFree(c.d_vals);
        if (c.d_nulls) (void)hipFree(c.d_nulls);
    }
    for (auto& ev : s->ev)
        if (ev) (void)hipEventDestroy(ev);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
    return OTT_OK;
}

int ott_store_reserve(ott_store* s, uint64_t n_rows) {
    if (!s) return fail(OTT_ERR_INVALID, "ott_store_reserve: store is NULL");
    if (s->multi) return multi_reserve(s, n_rows);
    ott::host::ExclusiveLock wr(s->rw);  // no query is running on any context
    std::lock_guard<std::mutex> g(s->mu);
    OTT_HIP(use_device(s));
    if (n_rows <= s->cap) return OTT_OK;
    return realloc_store(s, n_rows);  // exact-size allocation
}

int ott_store_append(ott_store* s, const float* rows_host, uint64_t n_rows) {
    if (!s) return fail(OTT_ERR_INVALID, "ott_store_append: store is NULL");
    if (n_rows == 0) return OTT_OK;
    if (!rows_host) return fail(OTT_ERR_INVALID, "ott_store_append: rows is NULL");
    if (s->multi) {
        AppendArgs a;
        a.kind = APPEND_HOST;
        a.rows = rows_host;
        return multi_append(s, a, n_rows);
    }
    ott::host::ExclusiveLock wr(s->rw);  // no query is running on any context
    std::lock_guard<std::mutex> g(s->mu);
    const size_t bytes = (size_t)n_rows * s->dim * 4;
    int rc;
    if (bytes <= PEND_SMALL && s->opt.stage_appends != 0) {
        // a small append (VecStore::add_vector: one row) is staged in pinned host memory; 4 MB of them travel together
        if (!s->pend.fits(n_rows, s->dim) && (rc = store_flush_locked(s))) return rc;
        if (!s->h_pend.p) {
            OTT_HIP(use_device(s));
            if ((rc = s->h_pend.ensure(PEND_BYTES))) return rc;
            s->pend.buf = (float*)s->h_pend.p;
            s->pend.cap_bytes = PEND_BYTES;
        }
        s->pend.stage(rows_host, n_rows, s->dim);
        return OTT_OK;
    }
    OTT_HIP(use_device(s));
    if ((rc = store_flush_locked(s))) return rc;  // staged rows come first
    return append_host_locked(s, rows_host, n_rows);
}

int ott_store_append_device(ott_store* s, const void* rows_dev, uint64_t n_rows) {
    if (!s) return fail(OTT_ERR_INVALID, "ott_store_append_device: store is NULL");
    if (n_rows == 0) return OTT_OK;
    if (!rows_dev) return fail(OTT_ERR_INVALID, "ott_store_append_device: rows is NULL");
    if (s->multi) {
        AppendArgs a;
        a.kind = APPEND_DEVICE;
        a.rows = rows_dev;
        return multi_append(s, a, n_rows);
    }
    ott::host::ExclusiveLock wr(s->rw);  // no query is running on any context
    std::lock_guard<std::mutex> g(s->mu);
    OTT_HIP(use_device(s));
    int rc = store_flush_locked(s);  // staged rows come first
    if (rc) return rc;
    if ((rc = grow(s, s->n + n_rows))) return rc;
    OTT_HIP(hipMemcpy2DAsync(s->d_rows + s->n * s->ld, (size_t)s->ld * 4, rows_dev, (size_t)s->dim * 4, (size_t)s->dim * 4,
                             n_rows, hipMemcpyDeviceToDevice, s->stream));
    rc = launch_inv_norms(s, s->n, n_rows);
    if (rc) return rc;
    rc = update_min_pos_inv(s, s->n, n_rows);
    if (rc) return rc;
    s->n += n_rows;
    kick_plane_build(s);
    return OTT_OK;
}

int ott_store_append_random(ott_store* s, uint64_t n_rows, uint64_t seed) {
    if (!s) return fail(OTT_ERR_INVALID, "ott_store_append_random: store is NULL");
    if (n_rows == 0) return OTT_OK;
    if (s->multi) {
        AppendArgs a;
        a.kind = APPEND_RANDOM;
        a.seed = seed;
        return multi_append(s, a, n_rows);
    }
    ott::host::ExclusiveLock wr(s->rw);  // no query is running on any context
    std::lock_guard<std::mutex> g(s->mu);
    OTT_HIP(use_device(s));
    int rc = store_flush_locked(s);  // staged rows come first
    if (rc) return rc;
    if ((rc = grow(s, s->n + n_rows))) return rc;
    rc = launch_rand_fill(s, s->n, n_rows, seed);
    if (rc) return rc;
    rc = launch_inv_norms(s, s->n, n_rows);
    if (rc) return rc;
    rc = update_min_pos_inv(s, s->n, n_rows);
    if (rc) return rc;
    s->n += n_rows;
    kick_plane_build(s);
    return OTT_OK;
}

int ott_store_append_clustered(ott_store* s, uint64_t n_rows, uint64_t seed, uint32_t n_clusters, float spread, float aniso) {
    if (!s) return fail(OTT_ERR_INVALID, "ott_store_append_clustered: store is NULL");
    if (n_clusters == 0 || !(spread >= 0.0f) || !(aniso >= 0.0f)) return fail(OTT_ERR_INVALID, "ott_store_append_clustered: n_clusters > 0, spread >= 0, aniso >= 0");
    if (n_rows == 0) return OTT_OK;
    if (s->multi) {
        AppendArgs a;
        a.kind = APPEND_CLUSTERED;
        a.seed = seed;
        a.n_clusters = n_clusters;
        a.spread = spread;
        a.aniso = aniso;
        return multi_append(s, a, n_rows);
    }
    ott::host::ExclusiveLock wr(s->rw);
    std::lock_guard<std::mutex> g(s->mu);
    OTT_HIP(use_device(s));
    int rc = store_flush_locked(s);  // staged rows come first
    if (rc) return rc;
    if ((rc = grow(s, s->n + n_rows))) return rc;
    hipLaunchKernelGGL(clustered_fill_kernel, dim3((uint32_t)s->n_cu * 8), dim3(256), 0, s->stream, s->d_rows, s->ld, s->dim, s->n, n_rows,
                       s->base_offset + s->n, seed, n_clusters, spread, aniso);
    OTT_HIP(hipGetLastError());
    if ((rc = launch_inv_norms(s, s->n, n_rows))) return rc;
    if ((rc = update_min_pos_inv(s, s->n, n_rows))) return rc;
    s->n += n_rows;
    kick_plane_build(s);
    return OTT_OK;
}

int ott_store_set_batch_image(ott_store* s, int enabled) {
    if (!s) return fail(OTT_ERR_INVALID, "ott_store_set_batch_image: store is NULL");
    if (s->multi) return multi_set_batch_image(s, enabled);
    ott::host::ExclusiveLock wr(s->rw);  // no query is running on any context
    std::lock_guard<std::mutex> g(s->img_mu);
    if (!enabled && s->d_img) {
        OTT_HIP(use_device(s));
        (void)hipFree(s->d_img);
        s->d_img = nullptr;
        s->img_rows = s->img_cap = 0;
    }
    if (!enabled && s->d_imgh) {
        OTT_HIP(use_device(s));
        (void)hipFree(s->d_imgh);
        s->d_imgh = nullptr;
        s->imgh_rows = 0;
    }
    if (!enabled && s->d_img8) {
        OTT_HIP(use_device(s));
        (void)hipFree(s->d_img8);
        if (s->d_img8_scale) (void)hipFree(s->d_img8_scale);
        s->d_img8 = nullptr;
        s->d_img8_scale = nullptr;
        s->img8_rows = 0;
    }
    if (enabled) s->img8_off = false;
    s->img_off = !enabled;
    if (enabled) s->imgh_off = false;
    return OTT_OK;
}

int ott_store_set_option(ott_store* s, const char* name, int64_t value) {
    if (!s || !name) return fail(OTT_ERR_INVALID, "ott_store_set_option: NULL argument");
    if (s->multi) return multi_set_option(s, name, value);
    ott::host::ExclusiveLock wr(s->rw);  // no query is running on any context
    std::lock_guard<std::mutex> g(s->img_mu);
    Options o = s->opt;
    if (option_set(o, name, (long long)value)) return fail(OTT_ERR_INVALID, std::string("ott_store_set_option: unknown option or bad value: ") + name);
    // a copy of the corpus that was declined because of an option can be built again once the option allows it
    if (s->opt.no_hi_pass && !o.no_hi_pass) s->imgh_off = false;
    if (s->opt.no_batch_image && !o.no_batch_image) s->imgh_off = s->img_off = s->img8_off = false;
    s->opt = o;
    return OTT_OK;
}

int ott_store_prepare_batch(ott_store* s) {
    if (!s) return fail(OTT_ERR_INVALID, "ott_store_prepare_batch: store is NULL");
    if (s->multi) return multi_prepare_batch(s);
    {
        const int rcf = store_flush(s);
        if (rcf) return rcf;
    }
    ott::host::SharedLock rd(s->rw);
    OTT_HIP(use_device(s));
    ott_store* ctx = ott::ctx_acquire(s);
    const int rc = ensure_first_plane(ctx);  // a no-op when it is up to date, switched off, or does not fit
    ott::ctx_release(ctx);
    return rc;
}

int ott_store_batch_ready(const ott_store* cs) {
    ott_store* s = const_cast<ott_store*>(cs);
    if (!s) return 0;
    if (s->multi) return multi_batch_ready(s);
    if (s->pend.count()) return 0;
    return first_plane_ready(s) ? 1 : 0;
}

int ott_store_write_rows(ott_store* s, uint64_t first_row, const float* rows_host, uint64_t n_rows) {
    if (!s) return fail(OTT_ERR_INVALID, "ott_store_write_rows: store is NULL");
    if (n_rows == 0) return OTT_OK;
    if (!rows_host) return fail(OTT_ERR_INVALID, "ott_store_write_rows: rows is NULL");
    if (s->multi) return multi_write_rows(s, first_row, rows_host, n_rows);
    ott::host::ExclusiveLock wr(s->rw);  // no query is running on any context
    std::lock_guard<std::mutex> g(s->mu);
    {
        const int rcf = store_flush_locked(s);
        if (rcf) return rcf;
    }
    if (first_row + n_rows > s->n) return fail(OTT_ERR_INVALID, "ott_store_write_rows: range exceeds store length");
    OTT_HIP(use_device(s));
    OTT_HIP(hipMemcpy2DAsync(s->d_rows + first_row * s->ld, (size_t)s->ld * 4, rows_host, (size_t)s->dim * 4,
                             (size_t)s->dim * 4, n_rows, hipMemcpyHostToDevice, s->stream));
    int rc = launch_inv_norms(s, first_row, n_rows);
    if (rc) return rc;
    if (s->d_img && first_row < s->img_rows) {  // keep the batch image in step with the rewritten rows
        const uint64_t cnt = (first_row + n_rows <= s->img_rows ? first_row + n_rows : s->img_rows) - first_row;
        const uint32_t ldi = (s->dim + 31u) & ~31u;
        const uint64_t work = cnt * (ldi / 4);
        const uint32_t grid = (uint32_t)std::min<uint64_t>((work + 255) / 256, (uint64_t)s->n_cu * 16);
        hipLaunchKernelGGL(split_rows_kernel, dim3(grid), dim3(256), 0, s->stream, s->d_rows, s->ld, s->dim, ldi, first_row, cnt, s->d_img, nullptr);
        OTT_HIP(hipGetLastError());
    }
    if (s->d_imgh && first_row < s->imgh_rows) {  // and the hi plane (its measured rounding loss can only grow)
        const uint64_t cnt = (first_row + n_rows <= s->imgh_rows ? first_row + n_rows : s->imgh_rows) - first_row;
        int rch = launch_store_hi_rows(s, s->stream, first_row, cnt);
        if (rch) return rch;
        uint32_t bits = 0;
        OTT_HIP(hipMemcpyAsync(&bits, s->d_imgh_rel, 4, hipMemcpyDeviceToHost, s->stream));
        OTT_HIP(hipStreamSynchronize(s->stream));
        memcpy(&s->imgh_rel, &bits, 4);
    }
    if (s->d_img8 && first_row < s->img8_rows) {  // and the int8 plane (its measured loss can only grow; marks of rewritten rows are re-taken)
        const uint64_t cnt = (first_row + n_rows <= s->img8_rows ? first_row + n_rows : s->img8_rows) - first_row;
        int rch = launch_i8_rows(s->stream, s->d_rows, s->ld, s->dim, (s->dim + 127u) & ~127u, first_row, cnt, s->d_img8, nullptr, 0.0f, s->d_img8_scale,
                                 nullptr, s->d_img8_rel, s->d_flag, I8_REL_FLAG, s->d_flag, s->n_cu);
        if (rch) return rch;
        uint32_t bits = 0;
        OTT_HIP(hipMemcpyAsync(&bits, s->d_img8_rel, 4, hipMemcpyDeviceToHost, s->stream));
        OTT_HIP(hipStreamSynchronize(s->stream));
        memcpy(&s->img8_rel, &bits, 4);
    }
    return update_min_pos_inv(s, first_row, n_rows);
}

uint64_t ott_store_len(const ott_store* s) { return s ? store_rows(s) : 0; }  // staged rows count: they were appended
uint32_t ott_store_dim(const ott_store* s) { return s ? s->dim : 0; }
int ott_store_device(const ott_store* s) { return s ? s->device : -1; }

int ott_store_set_chunk_size(ott_store* s, uint64_t chunk_size) {
    if (!s) return fail(OTT_ERR_INVALID, "ott_store_set_chunk_size: store is NULL");
    if (s->multi) return multi_set_chunk_size(s, chunk_size);
    s->chunk_size = chunk_size < 1 ? 1 : chunk_size;  // src/meta.rs:86-89
    return OTT_OK;
}
int ott_store_set_base_offset(ott_store* s, uint64_t base) {
    if (!s) return fail(OTT_ERR_INVALID, "ott_store_set_base_offset: store is NULL");
    if (s->multi) return multi_set_base_offset(s, base);
    s->base_offset = base;
    return OTT_OK;
}
int ott_store_set_reduce_order(ott_store* s, uint32_t reduce) {
    if (!s) return fail(OTT_ERR_INVALID, "ott_store_set_reduce_order: store is NULL");
    if (reduce > OTT_REDUCE_SEQ4) return fail(OTT_ERR_INVALID, "ott_store_set_reduce_order: unknown order");
    if (s->multi) return multi_set_reduce_order(s, reduce);
    s->reduce = reduce;
    return OTT_OK;
}

int ott_store_read_rows(const ott_store* s, uint64_t first_row, uint64_t n_rows, float* out_host) {
    if (!s || !out_host) return fail(OTT_ERR_INVALID, "ott_store_read_rows: NULL argument");
    if (!s->multi && store_flush(const_cast<ott_store*>(s))) return OTT_ERR_HIP;
    if (first_row + n_rows > s->n) return fail(OTT_ERR_INVALID, "ott_store_read_rows: range exceeds store length");
    if (!n_rows) return OTT_OK;
    if (s->multi) return multi_read(s, false, first_row, n_rows, out_host);
    ott::host::SharedLock rd(const_cast<ott_store*>(s)->rw);  // the rows cannot be reallocated under the copy
    if (first_row + n_rows > s->n) return fail(OTT_ERR_INVALID, "ott_store_read_rows: range exceeds store length");
    OTT_HIP(use_device(s));
    OTT_HIP(hipMemcpy2D(out_host, (size_t)s->dim * 4, s->d_rows + first_row * s->ld, (size_t)s->ld * 4, (size_t)s->dim * 4,
                        n_rows, hipMemcpyDeviceToHost));
    return OTT_OK;
}

int ott_store_read_inv_norms(const ott_store* s, uint64_t first_row, uint64_t n_rows, float* out_host) {
    if (!s || !out_host) return fail(OTT_ERR_INVALID, "ott_store_read_inv_norms: NULL argument");
    if (!s->multi && store_flush(const_cast<ott_store*>(s))) return OTT_ERR_HIP;
    if (first_row + n_rows > s->n) return fail(OTT_ERR_INVALID, "ott_store_read_inv_norms: range exceeds store length");
    if (!n_rows) return OTT_OK;
    if (s->multi) return multi_read(s, true, first_row, n_rows, out_host);
    ott::host::SharedLock rd(const_cast<ott_store*>(s)->rw);
    if (first_row + n_rows > s->n) return fail(OTT_ERR_INVALID, "ott_store_read_inv_norms: range exceeds store length");
    OTT_HIP(use_device(s));
    OTT_HIP(hipMemcpy(out_host, s->d_inv + first_row, n_rows * sizeof(float), hipMemcpyDeviceToHost));
    return OTT_OK;
}

int ott_store_sync(ott_store* s) {
    if (!s) return fail(OTT_ERR_INVALID, "ott_store_sync: store is NULL");
    if (s->multi) return multi_sync(s);
    OTT_HIP(use_device(s));
    OTT_HIP(hipStreamSynchronize(s->stream));
    return OTT_OK;
}

void* ott_store_stream(ott_store* s) { return s ? (void*)s->stream : nullptr; }  // (multi-GPU store: the stream of its merging shard)

}  // extern "C"

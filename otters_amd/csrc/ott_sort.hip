// ott_sort.hip — large-k path (k > 512, e.g. collect() with no take on a big store:
// take_count defaults to n_vecs, src/vec.rs:213).  The fused register top-k does not scale to
// thousands of entries, so the exact scorer dumps every passing (key, query) pair and a device
// radix sort orders them: the reference's own final step is a full sort of the
// collector (`into_sorted_vec`, src/vec_compute.rs:290-293; meta.rs:702-705).
//
// The sort is a stable LSD radix sort of (key, query) pairs written here (no library), ONE kernel per digit ("onesweep",
// round 3): a single histogram kernel reads the pairs once and counts every digit position of the pass plan; each digit pass
// then reads a tile of 4096 pairs, ranks them inside the workgroup (wave ballots -> wave histograms -> workgroup offsets,
// stable), learns where its digits start in the output by DECOUPLED LOOK-BACK over the tiles in front of it (per-digit
// status words: aggregate first, inclusive prefix once known; tiles take their index from an atomic ticket, so every tile a
// workgroup waits for is already running), reorders the tile through LDS and writes each digit's run with neighbouring
// lanes on neighbouring addresses.  A digit position on which all pairs agree (the top byte of ~row on a 10M-row store, the
// high bytes of the query id) is skipped outright — the histogram tells.  HBM-bound: a pass reads and writes the pairs
// once (24 B per pair); 10M pairs: 0.2 ms -> ~0.08 ms per pass, 1.6 -> ~0.6 ms for the merged single-query order.
// Pass plans (least significant first) express every order the large-k path needs — canonical (score, row, query),
// the reference's visit order (score, row >> 3, query, row & 7: store option tie_order), grouped by query — as a list of
// (source word, shift, width, direction) digits over the pair.
#include <math.h>
#include <stddef.h>
#include <string.h>

#include <algorithm>

#include "ott_internal.h"

namespace ott {

constexpr int RS_THREADS = 512;
constexpr int RS_WAVES = RS_THREADS / 64;
constexpr int RS_ITEMS = 8;
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;  // 4096 pairs
constexpr int RS_MAXP = 16;
constexpr uint32_t RS_SPIN_LIMIT = 1u << 24;    // look-back spins before a workgroup gives up (sets the error word: no hang)

struct RsPass {
    uint32_t src;    // 0 = key (u64), 1 = query id (u32)
    uint32_t shift;
    uint32_t mask;   // (1 << width) - 1, width <= 8
    uint32_t desc;   // 1 = larger digit first
};
struct RsPlan {
    RsPass pass[RS_MAXP];
    uint32_t n_pass;
    uint32_t abl;  // timing ablations (store option mfma_abl, results then WRONG): 1 no look-back, 2 no stores, 4 no loads, 8 no ranking
};
// The timing ablations exist only in the diagnostic build (make EXTRA=-DOTT_MFMA_DEBUG_BUILD): in the shipped library the tests
// below are the constant false and the radix passes carry no branch for them.
#ifdef OTT_MFMA_DEBUG_BUILD
#define SORT_ABL(plan, bit) (((plan).abl & (bit)) != 0u)
#else
#define SORT_ABL(plan, bit) false
#endif

__device__ __forceinline__ uint32_t rs_digit(const RsPass& ps, uint64_t key, uint32_t q) {
    const uint32_t d = (ps.src ? (q >> ps.shift) : (uint32_t)(key >> ps.shift)) & ps.mask;
    return ps.desc ? ps.mask - d : d;
}

// control block in device memory: [0, P*256) digit counts -> exclusive starts, then per pass: skip flag, buffer parity
struct RsCtl {
    uint32_t start[RS_MAXP * 256];  // histogram, then (rs_scan_kernel) the exclusive scan: where digit d of pass p starts
    uint32_t skip[RS_MAXP];         // 1 = every pair has the same digit at this position
    uint32_t parity[RS_MAXP + 1];   // number of passes that really ran before pass p (buffer A if even, B if odd); [n_pass] = total
    uint32_t ticket[RS_MAXP];       // next tile index of pass p
    uint32_t error;                 // a look-back gave up
};

__global__ __launch_bounds__(1024) void rs_hist_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ qs, uint64_t n, RsPlan plan,
                                                        RsCtl* __restrict__ ctl) {
    __shared__ uint32_t h[RS_MAXP * 256];
    for (uint32_t i = threadIdx.x; i < plan.n_pass * 256; i += 1024) h[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (uint64_t i0 = (uint64_t)blockIdx.x * 1024; i0 < n; i0 += (uint64_t)gridDim.x * 1024) {
        const uint64_t i = i0 + threadIdx.x;
        const bool have = i < n;
        const uint64_t key = have ? keys[i] : 0;
        const uint32_t q = have ? qs[i] : 0;
        const unsigned long long act = __ballot(have);
        for (uint32_t p = 0; p < plan.n_pass; p++) {
            const uint32_t d = rs_digit(plan.pass[p], key, q);
            const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
            // a whole wave on one bin (constant high digits) would serialise 64 LDS atomics: one lane adds the count instead
            if (__ballot(have && d != d0) == 0) {
                if (lane == 0 && act) atomicAdd(&h[p * 256 + d0], (uint32_t)__popcll(act));
            } else if (have) {
                atomicAdd(&h[p * 256 + d], 1u);
            }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < plan.n_pass * 256; i += 1024)
        if (h[i]) atomicAdd(&ctl->start[i], h[i]);
}

// one workgroup of 256 threads: per pass the exclusive scan of the 256 digit counts, the skip flag, the buffer parities
__global__ __launch_bounds__(256) void rs_scan_kernel(RsCtl* __restrict__ ctl, uint32_t n_pass, uint32_t n) {
    __shared__ uint32_t part[256];
    __shared__ uint32_t any_full;
    for (uint32_t p = 0; p < n_pass; p++) {
        const uint32_t v = ctl->start[p * 256 + threadIdx.x];
        if (threadIdx.x == 0) any_full = 0;
        part[threadIdx.x] = v;
        __syncthreads();
        if (v == n) any_full = 1;
        for (uint32_t off = 1; off < 256; off <<= 1) {
            const uint32_t u = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
            __syncthreads();
            part[threadIdx.x] += u;
            __syncthreads();
        }
        ctl->start[p * 256 + threadIdx.x] = part[threadIdx.x] - v;
        if (threadIdx.x == 0) ctl->skip[p] = any_full;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        uint32_t ran = 0;
        for (uint32_t p = 0; p < n_pass; p++) {
            ctl->parity[p] = ran;
            ran += ctl->skip[p] ? 0u : 1u;
        }
        ctl->parity[n_pass] = ran;
    }
}

// status word of (tile, digit): [63:62] 1 = aggregate (this tile's count), 2 = inclusive prefix (all tiles up to this one);
// [61:56] pass tag (pass index + 1: words left by an earlier pass read as "not there yet"); [55:0] the count
__device__ __forceinline__ uint64_t rs_word(uint32_t flag, uint32_t tag, uint64_t count) { return ((uint64_t)flag << 62) | ((uint64_t)tag << 56) | count; }

__global__ __launch_bounds__(RS_THREADS) void rs_pass_kernel(uint64_t* __restrict__ keysA, uint32_t* __restrict__ qsA, uint64_t* __restrict__ keysB,
                                                              uint32_t* __restrict__ qsB, uint32_t n, RsPlan plan, uint32_t p, RsCtl* __restrict__ ctl,
                                                              uint64_t* __restrict__ status) {
    __shared__ uint64_t s_key[RS_TILE];        // the reordered keys, then (second round) the reordered query ids in the same bytes
    __shared__ uint32_t whist[RS_WAVES][256];  // per wave: pairs of each digit seen so far (its running offset while ranking)
    __shared__ uint32_t dig_excl[256];         // where digit d starts inside the reordered tile
    __shared__ uint32_t gbase[256];            // where this tile's digit d starts in the output
    __shared__ uint32_t s_tile;
    __shared__ uint32_t s_wtot[4];
    if (ctl->skip[p]) return;  // every pair agrees on this digit: the order does not change
    const bool odd = ctl->parity[p] & 1u;
    const uint64_t* keys_in = odd ? keysB : keysA;
    const uint32_t* qs_in = odd ? qsB : qsA;
    uint64_t* keys_out = odd ? keysA : keysB;
    uint32_t* qs_out = odd ? qsA : qsB;
    const RsPass ps = plan.pass[p];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_tile = atomicAdd(&ctl->ticket[p], 1u);
    for (int w = 0; w < RS_WAVES; w++)
        if (threadIdx.x < 256) whist[w][threadIdx.x] = 0;
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint32_t t0 = tile * (uint32_t)RS_TILE;
    const uint32_t cnt = (n - t0) < (uint32_t)RS_TILE ? (n - t0) : (uint32_t)RS_TILE;

    // 1. load (wave-striped: item i of lane l is pair t0 + wave * 64 * ITEMS + i * 64 + l, so (i, lane) is the input order) and rank
    uint64_t key[RS_ITEMS];
    uint32_t q[RS_ITEMS], dg[RS_ITEMS], rk[RS_ITEMS];
    const uint32_t wbase = (uint32_t)wave * 64u * RS_ITEMS;
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
        const uint32_t pos = wbase + (uint32_t)i * 64u + (uint32_t)lane;
        const bool have = pos < cnt;
        key[i] = (have && !SORT_ABL(plan, 4u)) ? keys_in[t0 + pos] : (uint64_t)pos * 0x9E3779B97F4A7C15ull;
        q[i] = (have && !SORT_ABL(plan, 4u)) ? qs_in[t0 + pos] : 0;
    }
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
        const uint32_t pos = wbase + (uint32_t)i * 64u + (uint32_t)lane;
        const bool have = pos < cnt;
        const uint32_t d = rs_digit(ps, key[i], q[i]);
        dg[i] = d;
        if (SORT_ABL(plan, 8u)) {
            rk[i] = 0;
            continue;
        }
        // lanes of this wave that hold the same digit: eight ballots, one per digit bit
        unsigned long long peers = __ballot(have);
#pragma unroll
        for (int bit = 0; bit < 8; bit++) {
            const unsigned long long m = __ballot((d >> bit) & 1u);
            peers &= ((d >> bit) & 1u) ? m : ~m;
        }
        const uint32_t below = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
        const int leader = peers ? __builtin_ctzll(peers) : 0;
        uint32_t old = 0;
        if (have && lane == leader) {
            old = whist[wave][d];
            whist[wave][d] = old + (uint32_t)__popcll(peers);
        }
        old = (uint32_t)__shfl((int)old, leader);
        rk[i] = old + below;  // rank among this wave's pairs of digit d
    }
    __syncthreads();

    // 2. per digit (threads 0..255): wave offsets, the tile's count, its status word, the look-back
    uint32_t my_cnt = 0;
    if (threadIdx.x < 256) {
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < RS_WAVES; w++) {
            const uint32_t c = whist[w][threadIdx.x];
            whist[w][threadIdx.x] = run;  // exclusive over the waves
            run += c;
        }
        my_cnt = run;
        const uint32_t tag = p + 1u;
        uint64_t* mine = status + (size_t)tile * 256 + threadIdx.x;
        __hip_atomic_store(mine, rs_word(tile == 0 ? 2u : 1u, tag, run), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint64_t excl = 0;
        if (tile > 0 && !SORT_ABL(plan, 1u)) {
            // The walk back over the tiles in front: EIGHT status words per round trip (independent loads, issued back to back),
            // consumed in order.  One word per round trip made the walk the whole pass: with ~700 tiles in flight a tile finds
            // its nearest INCLUSIVE prefix hundreds of tiles back (107 us per pass for 24 B x 10M pairs; see profiles/dead_ends_rounds_2_4.md).
            constexpr int LB = 8;
            uint32_t spins = 0;
            int64_t t = (int64_t)tile - 1;
            bool done = false;
            while (!done && t >= 0) {
                uint64_t w[LB];
#pragma unroll
                for (int j = 0; j < LB; j++) {
                    const int64_t tt = t - j >= 0 ? t - j : 0;
                    w[j] = __hip_atomic_load(status + (size_t)tt * 256 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                int used = 0;
#pragma unroll
                for (int j = 0; j < LB; j++) {
                    if (done || used < j || t - j < 0) continue;
                    const uint32_t flag = (uint32_t)(w[j] >> 62), wtag = (uint32_t)(w[j] >> 56) & 63u;
                    if (wtag != tag || flag == 0) continue;  // that tile has not published yet: poll again from it
                    excl += w[j] & ((1ull << 56) - 1ull);
                    used = j + 1;
                    if (flag == 2) done = true;
                }
                if (done) break;
                if (used == 0) {  // the nearest tile is not there yet
                    if (++spins > RS_SPIN_LIMIT) {
                        ctl->error = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                t -= used;
            }
            __hip_atomic_store(mine, rs_word(2u, tag, excl + run), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        gbase[threadIdx.x] = ctl->start[p * 256 + threadIdx.x] + (uint32_t)excl;
    }
    __syncthreads();
    // exclusive scan of the tile's 256 digit counts: four waves scan 64 digits each with lane shuffles, then the three wave
    // totals in front are added (two barriers instead of the sixteen of a Hillis-Steele scan in LDS)
    {
        uint32_t incl = my_cnt;
        if (threadIdx.x < 256) {
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t u = (uint32_t)__shfl_up((int)incl, o);
                if (lane >= o) incl += u;
            }
            if (lane == 63) s_wtot[wave] = incl;
        }
        __syncthreads();
        if (threadIdx.x < 256) {
            uint32_t front = 0;
            for (int w = 0; w < wave; w++) front += s_wtot[w];
            dig_excl[threadIdx.x] = front + incl - my_cnt;
        }
        __syncthreads();
    }

    // 3. reorder through LDS — digit runs become contiguous, input order kept inside a run — and write with neighbouring
    //    lanes on neighbours of a run: keys first, then the query ids through the same LDS bytes
    uint32_t at_of[RS_ITEMS];
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
        const uint32_t pos = wbase + (uint32_t)i * 64u + (uint32_t)lane;
        at_of[i] = dig_excl[dg[i]] + whist[wave][dg[i]] + rk[i];
        if (pos < cnt) s_key[at_of[i]] = key[i];
    }
    __syncthreads();
    uint32_t out_of[RS_ITEMS];
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
        const uint32_t at = (uint32_t)i * RS_THREADS + threadIdx.x;
        out_of[i] = 0;
        if (at < cnt) {
            const uint64_t k2 = s_key[at];
            // the digit of a reordered pair: a key digit comes from the key just read; a query digit from the run the slot lies in
            uint32_t d;
            if (ps.src == 0) d = rs_digit(ps, k2, 0);
            else {
                uint32_t lo = 0, hi = 256;  // last digit whose run starts at or before `at` (runs may be empty)
                while (hi - lo > 1) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (dig_excl[mid] <= at) lo = mid;
                    else hi = mid;
                }
                d = lo;
            }
            out_of[i] = gbase[d] + (at - dig_excl[d]);
            if (!SORT_ABL(plan, 2u)) keys_out[out_of[i]] = k2;
        }
    }
    __syncthreads();
    uint32_t* s_q = reinterpret_cast<uint32_t*>(s_key);
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
        const uint32_t pos = wbase + (uint32_t)i * 64u + (uint32_t)lane;
        if (pos < cnt) s_q[at_of[i]] = q[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RS_ITEMS; i++) {
        const uint32_t at = (uint32_t)i * RS_THREADS + threadIdx.x;
        if (at < cnt && !SORT_ABL(plan, 2u)) qs_out[out_of[i]] = s_q[at];
    }
}

static size_t rs_tiles(uint64_t n) { return (size_t)((n + RS_TILE - 1) / RS_TILE); }
static size_t rs_tmp_bytes(uint64_t n) { return ((sizeof(RsCtl) + 255) & ~(size_t)255) + rs_tiles(n) * 256 * sizeof(uint64_t); }

static void rs_add_digits(RsPlan& pl, uint32_t src, uint32_t lo, uint32_t hi, bool desc) {  // bits [lo, hi) of the source word, LSD
    for (uint32_t b = lo; b < hi && pl.n_pass < RS_MAXP; b += 8) {
        const uint32_t w = hi - b < 8 ? hi - b : 8;
        pl.pass[pl.n_pass++] = RsPass{src, b, (1u << w) - 1u, desc ? 1u : 0u};
    }
}

// Sorts the n pairs (keysA, qsA) by the plan (stable, least significant digit first); the result is in the A buffers if
// *in_A, else in the B buffers.  tmp: rs_tmp_bytes(n) bytes.  Synchronises the stream once (the parity word).
static int radix_sort_plan(hipStream_t stream, uint64_t* keysA, uint32_t* qsA, uint64_t* keysB, uint32_t* qsB, uint64_t n, const RsPlan& plan, void* tmp,
                           int n_cu, bool* in_A) {
    *in_A = true;
    if (n < 2 || plan.n_pass == 0) return OTT_OK;
    if (n > 0xFFFFFFF0ull) return fail(OTT_ERR_UNSUPPORTED, "radix sort: more than 2^32 - 16 pairs");
    RsCtl* ctl = (RsCtl*)tmp;
    uint64_t* status = (uint64_t*)((char*)tmp + ((sizeof(RsCtl) + 255) & ~(size_t)255));
    OTT_HIP(hipMemsetAsync(tmp, 0, rs_tmp_bytes(n), stream));
    const uint32_t hb = (uint32_t)std::min<uint64_t>((uint64_t)n_cu, (n + 1023) / 1024);
    hipLaunchKernelGGL(rs_hist_kernel, dim3(hb), dim3(1024), 0, stream, (const uint64_t*)keysA, (const uint32_t*)qsA, n, plan, ctl);
    hipLaunchKernelGGL(rs_scan_kernel, dim3(1), dim3(256), 0, stream, ctl, plan.n_pass, (uint32_t)n);
    const uint32_t tiles = (uint32_t)rs_tiles(n);
    for (uint32_t p = 0; p < plan.n_pass; p++)
        hipLaunchKernelGGL(rs_pass_kernel, dim3(tiles), dim3(RS_THREADS), 0, stream, keysA, qsA, keysB, qsB, (uint32_t)n, plan, p, ctl, status);
    OTT_HIP(hipGetLastError());
    // parity[] .. error are adjacent in RsCtl: ONE copy brings back both words the host needs (each D2H copy in front of the
    // wait is ~20 us of turnaround on a path that has two sorts per query)
    struct { uint32_t parity[RS_MAXP + 1]; uint32_t ticket[RS_MAXP]; uint32_t error; } tail;
    static_assert(offsetof(RsCtl, error) - offsetof(RsCtl, parity) == sizeof(tail) - 4, "RsCtl: parity | ticket | error are contiguous");
    OTT_HIP(hipMemcpyAsync(&tail, &ctl->parity[0], sizeof(tail), hipMemcpyDeviceToHost, stream));
    OTT_HIP(hipStreamSynchronize(stream));
    if (tail.error) return fail(OTT_ERR_HIP, "radix sort: a look-back did not complete (device-side spin limit)");
    *in_A = (tail.parity[plan.n_pass] & 1u) == 0;
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

// ---------------------------------------------------------------------------------------------
// Small results, sorted by RANK (round 4).  The reference's default take is every row (src/vec.rs:213, src/meta.rs:638-644),
// so a query on a small store returns — sorted — everything it scored.  Up to SMALL_PAIRS (row, query) pairs take this path
// instead of the radix sort's passes and host waits: the scoring sweep dumps its (key, query) entries as before, then
//   small_rank_kernel   every entry counts the entries that come BEFORE it in result order (per query: inside its own query's
//                       group) — a 2-D grid, 256 entries x a 1024-entry slice of the list per workgroup, the slice staged in
//                       LDS, partial counts added atomically; all keys are distinct (row and query are part of them), so the
//                       counts are a permutation;
//   small_place_kernel  entry i goes to slot rank[i] of the result, in device memory;
//   small_copy_kernel   the result block and the groups' counts to pinned host memory, coalesced.
// No readback in front of a launch, one host wait behind the last one.  N^2 compares: 10k entries = 1e8, a few microseconds
// of the whole GPU.  10k rows x 768, default take: 0.22 -> 0.12 ms through the Python mirror (benchmarks/default_take_small.py).
// ---------------------------------------------------------------------------------------------
constexpr uint32_t SMALL_PAIRS = 16384;
constexpr uint32_t SMALL_PERQ_MAX = 1024;  // PER_QUERY: queries (their extents are prefix-summed in LDS)

// Result order as ONE 64-bit word compared descending: the score ordinal on top, below it the row and query bits that order
// equal scores — canonical (lower row, then lower query) or the reference's visit order (tie_sh = 3: 8-row block, query, row in
// the block).  `qbits` = bits of the largest query index; the host takes this path only while row bits + query bits fit 32.
// Per query the groups are kept apart by an equality test on the query, and the word is the dump's own key (score, lower row).
__device__ __forceinline__ uint64_t small_word(uint64_t key, uint32_t q, uint32_t perq, uint32_t sh, uint32_t qbits) {
    if (perq) return key;
    const uint32_t row = ~(uint32_t)(key & 0xFFFFFFFFull);
    uint32_t low;
    if (sh == 0) low = (row << qbits) | q;
    else low = (((row >> sh) << qbits | q) << sh) | (row & ((1u << sh) - 1u));
    return (key & 0xFFFFFFFF00000000ull) | (uint32_t)~low;
}

__global__ __launch_bounds__(256) void small_rank_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ qs,
                                                          const unsigned long long* __restrict__ cursor, uint32_t cap, uint32_t perq, uint32_t nq, uint32_t sh,
                                                          uint32_t qbits, uint32_t* __restrict__ rank, uint32_t* __restrict__ hist) {
    // the slice's order words in LDS, two per ds_read_b128 (and the queries, per-query mode); padding = 0: nothing ranks below it
    __shared__ __attribute__((aligned(16))) uint64_t sW[1024];
    __shared__ __attribute__((aligned(16))) uint32_t sQ[1024];
    __shared__ uint32_t sHist[SMALL_PERQ_MAX];
    const unsigned long long nn = *cursor;
    const uint32_t n = nn < cap ? (uint32_t)nn : cap;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x, c0 = blockIdx.y * 1024;
    if (blockIdx.x * 256 >= n || c0 >= n) return;  // (whole workgroups)
    const uint32_t cn = n - c0 < 1024u ? n - c0 : 1024u;
    const bool groups = perq && nq > 1;
    for (uint32_t j = threadIdx.x; j < 1024; j += 256) {
        uint64_t w = 0;
        uint32_t eq = 0xFFFFFFFFu;
        if (j < cn) {
            eq = qs[c0 + j];
            w = small_word(keys[c0 + j], eq, perq, sh, qbits);
        }
        sW[j] = w;
        if (groups) sQ[j] = eq;
    }
    const bool count_groups = perq && blockIdx.y == 0;  // the groups' sizes: counted once per entry, in LDS first — one global
    if (count_groups)                                   // atomic per (workgroup, query present), not one per entry on hist[q]
        for (uint32_t q = threadIdx.x; q < nq; q += 256) sHist[q] = 0;
    __syncthreads();
    const uint32_t q = i < n ? qs[i] : 0u;
    if (i < n) {
        const uint64_t mine = small_word(keys[i], q, perq, sh, qbits);
        uint32_t before = 0;
        typedef unsigned long long v2u64 __attribute__((ext_vector_type(2)));
        const v2u64* w2 = reinterpret_cast<const v2u64*>(sW);
        const uint32_t cn2 = (cn + 1u) >> 1;
        if (groups) {  // rank INSIDE the entry's own query group
            const uint2* q2 = reinterpret_cast<const uint2*>(sQ);
#pragma unroll 8
            for (uint32_t j = 0; j < cn2; j++) {
                const v2u64 e = w2[j];
                const uint2 f = q2[j];
                before += (f.x == q && e.x > mine) ? 1u : 0u;
                before += (f.y == q && e.y > mine) ? 1u : 0u;
            }
        } else {
#pragma unroll 8
            for (uint32_t j = 0; j < cn2; j++) {
                const v2u64 e = w2[j];
                before += e.x > mine ? 1u : 0u;
                before += e.y > mine ? 1u : 0u;
            }
        }
        if (before) atomicAdd(&rank[i], before);
        if (count_groups) atomicAdd(&sHist[q], 1u);
    }
    if (count_groups) {
        __syncthreads();
        for (uint32_t qq = threadIdx.x; qq < nq; qq += 256)
            if (sHist[qq]) atomicAdd(&hist[qq], sHist[qq]);
    }
}

// every entry to its slot of the result (device memory): rank[i] is complete here (the launch before counted every pair)
__global__ __launch_bounds__(256) void small_place_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ qs,
                                                           const unsigned long long* __restrict__ cursor, uint32_t cap, uint32_t perq, uint64_t k,
                                                           uint32_t stride, const uint32_t* __restrict__ rank, uint32_t take_max, uint64_t base,
                                                           ott_hit* __restrict__ out) {
    const unsigned long long nn = *cursor;
    const uint32_t n = nn < cap ? (uint32_t)nn : cap;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = rank[i];
    if (r >= k) return;
    const uint64_t key = keys[i];
    const uint32_t q = qs[i];
    ott_hit h;
    h.index = base + (uint32_t)~(uint32_t)(key & 0xFFFFFFFFull);
    h.score = score_of((uint32_t)(key >> 32), take_max != 0);
    h.query = q;
    out[perq ? (size_t)q * stride + r : (size_t)r] = h;
}

// The result block moved to (pinned host) memory in whole 16-byte hits, consecutive lanes to consecutive slots — the placing
// stores above go wherever an entry's rank says: scattered 16-byte writes, which over PCIe cost ~10 ns each (10k hits: 100 us
// measured) — and the groups' counts behind a complete histogram.  src: merged [k_out] hits; per query [nq][stride].
// It also leaves the control block (cursor, tickets, ranks, histogram) ZEROED for the next query on this context: no memset in
// front of the next dump (one API call less on a path that is bound by its launches).
__global__ __launch_bounds__(256) void small_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, uint32_t n16,
                                                          unsigned long long* __restrict__ cursor, uint32_t cap, uint32_t perq, uint32_t nq,
                                                          uint64_t k, uint32_t* __restrict__ hist, uint64_t* __restrict__ counts,
                                                          uint32_t* __restrict__ zero_words, uint32_t n_zero) {
    if (blockIdx.x == 0) {
        if (perq) {
            for (uint32_t q = threadIdx.x; q < nq; q += 256) {
                counts[q] = hist[q] < k ? hist[q] : k;
                hist[q] = 0;
            }
        } else if (threadIdx.x == 0) {
            const unsigned long long nn = *cursor;
            const uint64_t n = nn < cap ? nn : cap;
            counts[0] = n < k ? n : k;
        }
        __syncthreads();
        if (threadIdx.x == 0) *cursor = 0ull;
    }
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n_zero; i += gridDim.x * 256) zero_words[i] = 0u;  // tickets + ranks (nobody reads them here)
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) dst[i] = src[i];
}

// The same for ALL query groups of a per-query result in one launch (round 4: one launch and one copy per group were 64 + 64 API
// calls behind a 64-query batch on a small store).  The sorted entries are grouped by query; tab[q] = {first entry of q's group,
// first output slot of q's hits}; entry i of group q goes to slot (i - first) if that is below k.
__global__ __launch_bounds__(256) void hits_from_sorted_grouped_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ qs, uint64_t n,
                                                                        const uint64_t* __restrict__ tab, uint64_t k, uint32_t take_max, uint64_t base,
                                                                        ott_hit* __restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t q = qs[i];
        const uint64_t r = i - tab[2 * q];
        if (r >= k) continue;
        const uint64_t key = keys[i];
        ott_hit h;
        h.index = base + (uint32_t)~(uint32_t)(key & 0xFFFFFFFFull);
        h.score = score_of((uint32_t)(key >> 32), take_max != 0);
        h.query = q;
        out[tab[2 * q + 1] + r] = h;
    }
}

// Extents of the query groups of entries SORTED by query: start[q] = index of query q's first entry (start[] preset to
// 0xFFFFFFFF: a query without entries keeps it).  No atomics: round 2 counted the groups with one atomicAdd per entry on
// hist[q] — ten million atomics on ONE address for a single per-query list of every row: 114 ms behind a 5 ms scoring sweep.
__global__ __launch_bounds__(256) void group_start_kernel(const uint32_t* qs, uint64_t n, uint32_t* start) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t q = qs[i];
        if (i == 0 || qs[i - 1] != q) start[q] = (uint32_t)i;
    }
}

// The gates of the second phase from the sorted first-phase entries: the score ordinal of a result group's k-th best entry, or
// 0 (open) when fewer than k of its pairs passed.  Merged mode (start == nullptr): ONE group, every query gets its gate;
// per-query mode: the entries are grouped by query, start[q] = first entry of query q (group_start_kernel).  One thread: nq is small.
// `prev` (may be null, may be `gate` itself): gates carried in from earlier row slices of the same query (run_large_k) — a gate
// only ever rises: the new one is the larger of the two.
__global__ void gate_from_sorted_kernel(const uint64_t* keys, const uint32_t* start, uint64_t n, uint64_t k, uint32_t nq, uint32_t* gate, const uint32_t* prev) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    if (start == nullptr) {
        const uint32_t g = n >= k ? (uint32_t)(keys[k - 1] >> 32) : 0u;
        for (uint32_t q = 0; q < nq; q++) {
            const uint32_t pv = prev ? prev[q] : 0u;
            gate[q] = g > pv ? g : pv;
        }
        return;
    }
    uint64_t next = n;  // start of the next query that has entries
    for (uint32_t q = nq; q-- > 0;) {
        const uint32_t st = start[q];
        const uint32_t pv = prev ? prev[q] : 0u;
        if (st == 0xFFFFFFFFu) {
            gate[q] = pv;
            continue;
        }
        const uint32_t g = next - st >= k ? (uint32_t)(keys[(uint64_t)st + k - 1] >> 32) : 0u;
        gate[q] = g > pv ? g : pv;
        next = st;
    }
}

// the first `m_rows` rows of a plan and the rest
static void split_plan(const RunPlan& pl, uint64_t m_rows, RunPlan& a, RunPlan& b) {
    a = RunPlan();
    b = RunPlan();
    uint64_t left = m_rows;
    for (const ott_run& r : pl.runs) {
        const uint64_t take = r.count < left ? r.count : left;
        if (take) a.runs.push_back(ott_run{r.start, take});
        if (take < r.count) b.runs.push_back(ott_run{r.start + take, r.count - take});
        left -= take;
    }
    for (const ott_run& r : a.runs) a.rows_scored += r.count;
    for (const ott_run& r : b.runs) b.rows_scored += r.count;
}

// One SLICE of the sort path: the (row, query) pairs of `pl` — at most 2^30 of them (run_large_k cuts longer plans) — scored,
// listed, sorted; lists[g] = the slice's best k_eff hits of result group g, in result order.  gate_in (may be null): per query,
// a score ordinal that every pair worth listing must reach — the k-th best of the same group over EARLIER slices, a lower bound
// of the final k-th best (ties included: the merge decides among them).
static int large_k_slice(ott_store* s, const float* queries, uint32_t nq, const ott_query_desc* d, bool perq, const RunPlan& pl, uint64_t k_eff,
                         const uint64_t* d_mask, uint64_t mask_bits, std::vector<std::vector<ott_hit>>& lists, ott_stats& st,
                         const std::vector<uint32_t>* gate_in) {
    const uint64_t cap = pl.rows_scored * nq;
    if (cap > (1ull << 30)) return fail(OTT_ERR_INVALID, "large_k_slice: internal error (a slice of more than 2^30 pairs)");
    int rc;
    if ((rc = s->l_keysA.ensure(cap * 8))) return rc;
    if ((rc = s->l_keysB.ensure(cap * 8))) return rc;
    if ((rc = s->l_qA.ensure(cap * 4))) return rc;
    if ((rc = s->l_qB.ensure(cap * 4))) return rc;
    if ((rc = s->l_cursor.ensure(8))) return rc;

    uint64_t* kA = (uint64_t*)s->l_keysA.p;
    uint64_t* kB = (uint64_t*)s->l_keysB.p;
    uint32_t* qA = (uint32_t*)s->l_qA.p;
    uint32_t* qB = (uint32_t*)s->l_qB.p;
    const uint32_t groups = perq ? nq : 1;
    const int tile = nq == 1 ? 1 : 4;
    const uint32_t passes = (nq + tile - 1) / tile;

    // ---- small results: dump, rank, place — no radix passes, no readback in front of a launch, one host wait (see small_rank_kernel)
    uint32_t small_qbits = 0, small_rbits = 1;
    while (nq > 1 && small_qbits < 32 && ((uint64_t)(nq - 1) >> small_qbits) != 0) small_qbits++;
    while (small_rbits < 32 && ((s->n - 1 + s->cur_tie_off) >> small_rbits) != 0) small_rbits++;
    if (cap <= SMALL_PAIRS && (perq ? nq <= SMALL_PERQ_MAX : small_rbits + small_qbits <= 32) && s->opt.small_sort != 0 && gate_in == nullptr) {
        const std::vector<uint32_t> prefix = tile_prefix(pl, 64);
        const bool lean = nq == 1 && s->dimq <= OTT_QEMB_MAX && pl.runs.size() <= 2;
        if (!lean && (rc = upload_exact_inputs(s, queries, nq, pl, prefix))) return rc;
        // [cursor (8 B) | pad | tickets (64 x 4) | rank (cap x 4) | hist (nq x 4)]: one memset
        const size_t off_ticket = 64, off_rank = off_ticket + 64 * 4, off_hist = off_rank + (size_t)cap * 4, ctl_bytes = off_hist + (size_t)nq * 4;
        // the block is zero when a query finds it: zeroed when it is (re)allocated, and left zeroed by every query's last kernel
        // (l_ctl serves this path only; a failed query leaves the stream's work to finish and the block is zeroed again)
        if (s->l_ctl.cap < ctl_bytes || !s->l_ctl_clean) {
            if ((rc = s->l_ctl.ensure(ctl_bytes))) return rc;
            OTT_HIP(hipMemsetAsync(s->l_ctl.p, 0, s->l_ctl.cap, s->stream));
        }
        s->l_ctl_clean = false;
        char* ctl = (char*)s->l_ctl.p;
        ExactParams p;
        fill_exact_params(s, d, pl, nq, d_mask, mask_bits, prefix.back(), p);
        p.k = 1;
        p.dump_keys = kA;
        p.dump_q = qA;
        p.dump_cursor = (unsigned long long*)ctl;
        p.dump_cap = cap;
        p.dump_gate = nullptr;
        if (lean) {  // single query, at most two runs: everything rides in the kernel arguments (no H2D copy in front)
            p.embedded = 1;
            p.queries = nullptr;
            p.qinv = nullptr;
            p.runs = nullptr;
            p.tile_prefix = nullptr;
            memcpy(p.qemb, queries, (size_t)s->dim * 4);
            p.eqinv = host_inv_norm_exact(queries, s->dim);
            for (size_t i = 0; i < pl.runs.size(); i++) p.eruns[i] = pl.runs[i];
            for (size_t i = 0; i < prefix.size(); i++) p.eprefix[i] = prefix[i];
        }
        // the sweep: rows8 (eight lanes per row, a workgroup per 64-row tile, up to 8 queries per pass: 10 us for 10k x 768
        // where the streaming kernel needs 33) wherever it fits, which a store this small nearly always does
        const bool rows8 = s->dimq <= 2048 && prefix.back() <= 1024 && s->opt.exact_small != 0 && s->opt.exact_small != 1;
        uint32_t passes_run = passes;
        OTT_HIP(hipEventRecord(s->ev[3], s->stream));
        if (rows8) {
            uint32_t t8 = 1;
            while (t8 < nq && t8 < 8) t8 <<= 1;
            passes_run = (nq + t8 - 1) / t8;
            p.small = 2;
            p.perq = 0;
            p.list_stride = 64;
            for (uint32_t ps = 0; ps < passes_run; ps++) {
                p.q0 = ps * t8;
                if ((rc = launch_exact(s, p, (int)t8, 1, (int)prefix.back()))) return rc;
            }
        } else {
            const int grid = exact_grid(s, prefix.back());
            for (uint32_t ps = 0; ps < passes; ps++) {
                p.q0 = ps * tile;
                if ((rc = launch_exact_dump(s, p, tile, grid))) return rc;
            }
        }
        OTT_HIP(hipEventRecord(s->ev[4], s->stream));
        const uint64_t pool_g = perq ? pl.rows_scored : cap;
        const uint32_t stride = (uint32_t)(k_eff < pool_g ? k_eff : pool_g);  // slots per group
        const size_t cnt_bytes = (((size_t)groups * 8) + 63) & ~(size_t)63, hits_bytes = (size_t)groups * stride * sizeof(ott_hit);
        if ((rc = s->h_hits.ensure(cnt_bytes + hits_bytes))) return rc;
        char* hh = (char*)s->h_hits.p;
        void* mapped = nullptr;
        OTT_HIP(hipHostGetDevicePointer(&mapped, hh, 0));
        const uint32_t cap32 = (uint32_t)cap;
        // the entries land in device memory in result order, then travel to the host as one coalesced block
        if ((rc = s->d_hits.ensure(hits_bytes ? hits_bytes : 16))) return rc;
        hipLaunchKernelGGL(small_rank_kernel, dim3((cap32 + 255) / 256, (cap32 + 1023) / 1024), dim3(256), 0, s->stream, (const uint64_t*)kA, (const uint32_t*)qA,
                           (const unsigned long long*)ctl, cap32, perq ? 1u : 0u, nq, s->cur_tie_sh, small_qbits, (uint32_t*)(ctl + off_rank), (uint32_t*)(ctl + off_hist));
        OTT_HIP(hipGetLastError());
        hipLaunchKernelGGL(small_place_kernel, dim3((cap32 + 255) / 256), dim3(256), 0, s->stream, (const uint64_t*)kA, (const uint32_t*)qA,
                           (const unsigned long long*)ctl, cap32, perq ? 1u : 0u, k_eff, stride, (const uint32_t*)(ctl + off_rank),
                           d->take == OTT_TAKE_MAX ? 1u : 0u, tie_base(s), (ott_hit*)s->d_hits.p);
        OTT_HIP(hipGetLastError());
        {
            const uint32_t n16 = (uint32_t)(hits_bytes / 16);
            uint32_t blocks = (n16 + 255) / 256;
            blocks = blocks > 64u ? 64u : (blocks ? blocks : 1u);
            hipLaunchKernelGGL(small_copy_kernel, dim3(blocks), dim3(256), 0, s->stream, (const uint4*)s->d_hits.p, (uint4*)((char*)mapped + cnt_bytes), n16,
                               (unsigned long long*)ctl, cap32, perq ? 1u : 0u, nq, k_eff, (uint32_t*)(ctl + off_hist), (uint64_t*)mapped,
                               (uint32_t*)(ctl + off_ticket), (uint32_t)((off_hist - off_ticket) / 4));
            OTT_HIP(hipGetLastError());
        }
        OTT_HIP(hipEventRecord(s->ev[5], s->stream));
        OTT_HIP(hipStreamSynchronize(s->stream));  // the one wait
        s->l_ctl_clean = true;
        const uint32_t passes = passes_run;  // (what the stats below report)
        const uint64_t* cnt = (const uint64_t*)hh;
        const ott_hit* hits = (const ott_hit*)(hh + cnt_bytes);
        lists.assign(groups, {});
        for (uint32_t g = 0; g < groups; g++) lists[g].assign(hits + (size_t)g * stride, hits + (size_t)g * stride + cnt[g]);
        float ms2 = 0.f;
        if (hipEventElapsedTime(&ms2, s->ev[3], s->ev[4]) == hipSuccess) st.score_ns += (uint64_t)(ms2 * 1e6);
        if (hipEventElapsedTime(&ms2, s->ev[4], s->ev[5]) == hipSuccess) st.merge_ns += (uint64_t)(ms2 * 1e6);
        st.passes += passes;
        st.bytes_scanned += (uint64_t)passes * pl.rows_scored * ((uint64_t)s->dim * 4 + (d->metric == OTT_METRIC_COSINE ? 4 : 0));
        return OTT_OK;
    }

    // which kernel sweeps: rows8 on small stores (decided on the whole plan; the two phases' sub-plans are smaller still)
    const bool sweep8 = s->dimq <= 2048 && tile_prefix(pl, 64).back() <= 1024 && s->opt.exact_small != 0 && s->opt.exact_small != 1;
    uint32_t t8 = 1;
    while (t8 < nq && t8 < 8) t8 <<= 1;
    const uint32_t passes_eff = sweep8 ? (nq + t8 - 1) / t8 : passes;  // corpus passes one sweep makes (stats)
    // one scoring sweep over the rows of `plan`: every passing pair whose ordinal reaches its query's gate is appended to
    // (keys, qs) behind the `first` entries already there; the number of entries afterwards comes back in *n_entries
    auto dump = [&](const RunPlan& plan, uint64_t* keys, uint32_t* qs, uint64_t first, const uint32_t* gate, unsigned long long* n_entries) -> int {
        *n_entries = first;
        if (plan.rows_scored == 0) return OTT_OK;
        const std::vector<uint32_t> prefix = tile_prefix(plan, 64);
        int r = upload_exact_inputs(s, queries, nq, plan, prefix);
        if (r) return r;
        OTT_HIP(hipMemsetAsync(s->l_cursor.p, 0, 8, s->stream));
        if (first) OTT_HIP(hipMemsetD32Async((hipDeviceptr_t)s->l_cursor.p, (int)(uint32_t)first, 1, s->stream));  // first <= cap <= 2^30
        ExactParams p;
        fill_exact_params(s, d, plan, nq, d_mask, mask_bits, prefix.back(), p);
        p.k = 1;
        p.dump_keys = keys;
        p.dump_q = qs;
        p.dump_cursor = (unsigned long long*)s->l_cursor.p;
        p.dump_cap = cap;
        p.dump_gate = gate;
        // small stores: the rows8 sweep (eight lanes per row, a workgroup per 64-row tile, up to 8 queries per pass) — 16 queries
        // over 10k x 768: two passes of ~12 us where the streaming kernel needed four of ~110
        if (sweep8) {
            p.small = 2;
            p.perq = 0;
            p.list_stride = 64;
            for (uint32_t q0 = 0; q0 < nq; q0 += t8) {
                p.q0 = q0;
                if ((r = launch_exact(s, p, (int)t8, 1, (int)prefix.back()))) return r;
            }
        } else {
            const int grid = exact_grid(s, prefix.back());
            for (uint32_t ps = 0; ps < passes; ps++) {
                p.q0 = ps * tile;
                if ((r = launch_exact_dump(s, p, tile, grid))) return r;
            }
        }
        OTT_HIP(hipMemcpyAsync(n_entries, s->l_cursor.p, 8, hipMemcpyDeviceToHost, s->stream));
        OTT_HIP(hipStreamSynchronize(s->stream));
        if (*n_entries > cap) *n_entries = cap;
        return OTT_OK;
    };
    // sorts entries [0, n) of (kA, qA) into result order (merged: best first overall; per query: grouped by query, best first
    // in each group); on return kA / qA point at the sorted arrays and kB / qB at the other pair
    auto sort_entries = [&](uint64_t n, bool score_only) -> int {
        int r = s->l_tmp.ensure(rs_tmp_bytes(n));
        if (r) return r;
        uint32_t qbits = 0;
        while (nq > 1 && qbits < 32 && ((uint64_t)(nq - 1) >> qbits) != 0) qbits++;
        RsPlan plan;
        memset(&plan, 0, sizeof(plan));
        plan.abl = (uint32_t)s->opt.mfma_abl;  // (diagnostics only; 0 in normal use)
        // key = ord(score) << 32 | ~row: only the low bits of ~row that can differ between rows of this store are sorted on
        uint32_t rbits = 1;
        while (rbits < 32 && ((s->n - 1 + s->cur_tie_off) >> rbits) != 0) rbits++;  // (the key's row field is row + tie_off)
        const uint32_t sh = s->cur_tie_sh < rbits ? s->cur_tie_sh : 0u;
        auto key_digits = [&](uint32_t from) {  // bits [from, rbits) of ~row, then the 32 bits of the score ordinal
            rs_add_digits(plan, 0, from, rbits, true);
            rs_add_digits(plan, 0, 32, 64, true);
        };
        if (score_only) {
            // first phase: only the k-th best SCORE of each group is wanted (the order among equal scores is the final sort's business)
            rs_add_digits(plan, 0, 32, 64, true);
            if (perq) rs_add_digits(plan, 1, 0, qbits, false);
        } else if (!perq) {
            if (sh == 0) {
                // canonical merged order: key (score, then lower row) descending, ties by query ascending — LSD: query first
                rs_add_digits(plan, 1, 0, qbits, false);
                key_digits(0);
            } else {
                // the reference's visit order among equal scores: 8-row block, then query, then row within the block
                rs_add_digits(plan, 0, 0, sh, true);
                rs_add_digits(plan, 1, 0, qbits, false);
                key_digits(sh);
            }
        } else {
            // grouped by query, each group key descending (one query: row order IS the visit order): key first, then the query
            key_digits(0);
            rs_add_digits(plan, 1, 0, qbits, false);
        }
        bool in_A = true;
        if ((r = radix_sort_plan(s->stream, kA, qA, kB, qB, n, plan, s->l_tmp.p, s->n_cu, &in_A))) return r;
        if (!in_A) {
            std::swap(kA, kB);
            std::swap(qA, qB);
        }
        return OTT_OK;
    };
    // per-query entry counts of the sorted entries (per-query mode: the groups' extents)
    auto group_starts = [&](uint64_t n) -> int {  // l_hist[q] = first entry of query q among the sorted entries (0xFFFFFFFF: none)
        int r = s->l_hist.ensure((size_t)nq * 4);
        if (r) return r;
        OTT_HIP(hipMemsetAsync(s->l_hist.p, 0xFF, (size_t)nq * 4, s->stream));
        hipLaunchKernelGGL(group_start_kernel, dim3((uint32_t)s->n_cu * 4), dim3(256), 0, s->stream, qA, n, (uint32_t*)s->l_hist.p);
        OTT_HIP(hipGetLastError());
        return OTT_OK;
    };
    auto group_hist = [&](uint64_t n, std::vector<uint32_t>& h) -> int {
        int r = group_starts(n);
        if (r) return r;
        std::vector<uint32_t> st(nq);
        OTT_HIP(hipMemcpyAsync(st.data(), s->l_hist.p, (size_t)nq * 4, hipMemcpyDeviceToHost, s->stream));
        OTT_HIP(hipStreamSynchronize(s->stream));
        h.assign(nq, 0);
        uint64_t next = n;
        for (uint32_t q = nq; q-- > 0;) {
            if (st[q] == 0xFFFFFFFFu) continue;
            h[q] = (uint32_t)(next - st[q]);
            next = st[q];
        }
        return OTT_OK;
    };

    // Two phases when k is small beside the store (round 3; store option "large_k_pre").  Phase 1 scores the first m rows and
    // sorts them; the k-th best of a result group there is a LOWER bound of the group's final k-th best, so phase 2 lists, of
    // the remaining rows, only the pairs that reach it (ties included: the final sort decides among them) — ~k n / m pairs
    // instead of n, behind one wave-level atomic per tile that HOLDS a survivor instead of one per tile.  m = sqrt(k x pairs
    // per group) balances the two lists.  Rows in an unlucky order (best last) only cost the pruning, never the result.  The
    // flat fill pass of the reference tie order (every score ranks the same) has no bound to use and stays single-phase.
    uint64_t m_rows = 0;
    if (s->opt.large_k_pre != 0 && !s->cur_flat && pl.rows_scored > 0) {
        const double pairs = (double)pl.rows_scored * (perq ? 1.0 : (double)nq);
        const double f = sqrt((double)k_eff / pairs);
        if (f <= 0.25) {
            m_rows = ((uint64_t)ceil(f * (double)pl.rows_scored) + 63) & ~63ull;
            if (m_rows < 4096) m_rows = 4096;
            if (m_rows * 4 > pl.rows_scored) m_rows = 0;
        }
    }
    OTT_HIP(hipEventRecord(s->ev[3], s->stream));
    unsigned long long n_entries = 0;
    if ((rc = s->l_gate.ensure((size_t)nq * 4))) return rc;
    uint32_t* d_gate = (uint32_t*)s->l_gate.p;
    const uint32_t* gate0 = nullptr;  // the gates carried in from earlier slices, on the device
    if (gate_in) {
        OTT_HIP(hipMemcpyAsync(d_gate, gate_in->data(), (size_t)nq * 4, hipMemcpyHostToDevice, s->stream));  // (pageable: the copy is staged before the call returns)
        gate0 = d_gate;
    }
    if (m_rows) {
        RunPlan plA, plB;
        split_plan(pl, m_rows, plA, plB);
        if ((rc = dump(plA, kA, qA, 0, gate0, &n_entries))) return rc;
        if (n_entries) {
            if ((rc = sort_entries(n_entries, true))) return rc;
            const uint32_t* d_hist = nullptr;
            if (perq) {  // the groups' extents stay on the device
                if ((rc = group_starts(n_entries))) return rc;
                d_hist = (const uint32_t*)s->l_hist.p;
            }
            hipLaunchKernelGGL(gate_from_sorted_kernel, dim3(1), dim3(64), 0, s->stream, (const uint64_t*)kA, d_hist, (uint64_t)n_entries, k_eff, nq, d_gate, gate0);
            OTT_HIP(hipGetLastError());
        } else if (!gate0) {
            OTT_HIP(hipMemsetAsync(d_gate, 0, (size_t)nq * 4, s->stream));
        }
        if ((rc = dump(plB, kA, qA, n_entries, d_gate, &n_entries))) return rc;
    } else {
        if ((rc = dump(pl, kA, qA, 0, gate0, &n_entries))) return rc;
    }
    OTT_HIP(hipEventRecord(s->ev[4], s->stream));

    lists.assign(groups, {});
    if (n_entries) {
        if ((rc = sort_entries(n_entries, false))) return rc;
        // group extents
        std::vector<uint64_t> first(groups, 0), count(groups, 0);
        if (!perq) count[0] = n_entries < k_eff ? n_entries : k_eff;
        else {
            std::vector<uint32_t> h;
            if ((rc = group_hist(n_entries, h))) return rc;
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
        constexpr size_t PIECE = (size_t)128 * 1024;  // hits per piece (2 MB)
        // several groups, a result that fits one pinned block: ONE launch for all groups, ONE copy to the host
        const bool one_shot = perq && groups > 2 && total < 2 * PIECE;
        if (one_shot) {
            std::vector<uint64_t> tab((size_t)groups * 2);
            for (uint32_t g = 0; g < groups; g++) {
                tab[2 * g] = first[g];
                tab[2 * g + 1] = o;
                o += count[g];
            }
            if ((rc = s->d_misc.ensure(tab.size() * 8))) return rc;
            if ((rc = s->h_hits.ensure((size_t)(total ? total : 1) * sizeof(ott_hit) + tab.size() * 8))) return rc;
            uint64_t* htab = (uint64_t*)((char*)s->h_hits.p + (size_t)(total ? total : 1) * sizeof(ott_hit));
            memcpy(htab, tab.data(), tab.size() * 8);
            OTT_HIP(hipMemcpyAsync(s->d_misc.p, htab, tab.size() * 8, hipMemcpyHostToDevice, s->stream));
            const uint32_t blocks = (uint32_t)std::min<uint64_t>((n_entries + 255) / 256, (uint64_t)s->n_cu * 8);
            hipLaunchKernelGGL(hits_from_sorted_grouped_kernel, dim3(blocks), dim3(256), 0, s->stream, kA, qA, (uint64_t)n_entries, (const uint64_t*)s->d_misc.p,
                               k_eff, d->take == OTT_TAKE_MAX ? 1u : 0u, tie_base(s), (ott_hit*)s->d_hits.p);
            OTT_HIP(hipGetLastError());
            OTT_HIP(hipEventRecord(s->ev[5], s->stream));
            if (total) OTT_HIP(hipMemcpyAsync(s->h_hits.p, s->d_hits.p, (size_t)total * sizeof(ott_hit), hipMemcpyDeviceToHost, s->stream));
            OTT_HIP(hipStreamSynchronize(s->stream));
            const ott_hit* hh = (const ott_hit*)s->h_hits.p;
            o = 0;
            for (uint32_t g = 0; g < groups; g++) {
                lists[g].assign(hh + o, hh + o + count[g]);
                o += count[g];
            }
            float ms1 = 0.f;
            if (hipEventElapsedTime(&ms1, s->ev[3], s->ev[4]) == hipSuccess) st.score_ns += (uint64_t)(ms1 * 1e6);
            if (hipEventElapsedTime(&ms1, s->ev[4], s->ev[5]) == hipSuccess) st.merge_ns += (uint64_t)(ms1 * 1e6);
            st.passes += passes_eff;
            st.bytes_scanned += (uint64_t)passes_eff * pl.rows_scored * ((uint64_t)s->dim * 4 + (d->metric == OTT_METRIC_COSINE ? 4 : 0));
            return OTT_OK;
        }
        for (uint32_t g = 0; g < groups; g++) {
            if (!count[g]) continue;
            hipLaunchKernelGGL(hits_from_sorted_kernel, dim3((uint32_t)((count[g] + 255) / 256)), dim3(256), 0, s->stream, kA, qA, first[g],
                               count[g], d->take == OTT_TAKE_MAX ? 1u : 0u, tie_base(s), (ott_hit*)s->d_hits.p + o);
            OTT_HIP(hipGetLastError());
            o += count[g];
        }
        OTT_HIP(hipEventRecord(s->ev[5], s->stream));
        o = 0;
        // Results of a million hits (the reference's default take is every row): a device-to-host copy into pageable memory
        // runs at ~2 GB/s here, through pinned memory at PCIe speed.  Large results (256k hits and more) come over in 2-MB pieces through two
        // pinned buffers, each piece copied on to its list while the next one is on the wire (80 MB: 40 -> ~10 ms).
        if (total >= 2 * PIECE) {
            if ((rc = s->h_hits.ensure(2 * PIECE * sizeof(ott_hit)))) return rc;
            ott_hit* pin[2] = {(ott_hit*)s->h_hits.p, (ott_hit*)s->h_hits.p + PIECE};
            // straight into the caller's buffer when query_core offered it (the groups back to back, as it would copy them): the
            // lists stay empty.  Through the lists a result of 3M hits crossed host memory three more times, page faults included
            // (zero-filled vector, copy in, copy out: 24 ms behind 2 ms of GPU work)
            const bool direct = s->direct_out != nullptr && total <= s->direct_cap;
            if (direct) {
                s->direct_done = true;
                s->direct_counts.assign(count.begin(), count.end());
            } else {
                for (uint32_t g = 0; g < groups; g++) lists[g].resize(count[g]);
            }
            // pieces never straddle two lists: (list, offset, n) in order
            struct Piece { uint32_t g; uint64_t at, n, src; };
            std::vector<Piece> pieces;
            for (uint32_t g = 0; g < groups; g++) {
                for (uint64_t at = 0; at < count[g]; at += PIECE) pieces.push_back({g, at, std::min<uint64_t>(PIECE, count[g] - at), o + at});
                o += count[g];
            }
            for (size_t i = 0; i <= pieces.size(); i++) {
                if (i < pieces.size()) {
                    OTT_HIP(hipMemcpyAsync(pin[i & 1], (ott_hit*)s->d_hits.p + pieces[i].src, pieces[i].n * sizeof(ott_hit), hipMemcpyDeviceToHost, s->stream));
                    OTT_HIP(hipEventRecord(s->ev[i & 1], s->stream));
                }
                if (i > 0) {
                    const Piece& pc = pieces[i - 1];
                    OTT_HIP(hipEventSynchronize(s->ev[(i - 1) & 1]));
                    memcpy(direct ? s->direct_out + pc.src : lists[pc.g].data() + pc.at, pin[(i - 1) & 1], pc.n * sizeof(ott_hit));
                }
            }
        } else {
            for (uint32_t g = 0; g < groups; g++) {
                lists[g].resize(count[g]);
                if (count[g]) OTT_HIP(hipMemcpyAsync(lists[g].data(), (ott_hit*)s->d_hits.p + o, count[g] * sizeof(ott_hit), hipMemcpyDeviceToHost, s->stream));
                o += count[g];
            }
        }
        OTT_HIP(hipStreamSynchronize(s->stream));
    } else {
        OTT_HIP(hipEventRecord(s->ev[5], s->stream));
        OTT_HIP(hipStreamSynchronize(s->stream));
    }
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s->ev[3], s->ev[4]) == hipSuccess) st.score_ns += (uint64_t)(ms * 1e6);
    if (hipEventElapsedTime(&ms, s->ev[4], s->ev[5]) == hipSuccess) st.merge_ns += (uint64_t)(ms * 1e6);
    st.passes += passes_eff;
    st.bytes_scanned += (uint64_t)passes_eff * pl.rows_scored * ((uint64_t)s->dim * 4 + (d->metric == OTT_METRIC_COSINE ? 4 : 0));
    return OTT_OK;
}

// The sort path (k > 512, the reference's default take = every row: src/vec.rs:213-219) over ANY number of (row, query) pairs.
// Up to 2^29 pairs: one slice, as before.  Beyond (round 5; until then 2^31 pairs were the limit and 24 B of scratch per pair
// meant 51 GB at that limit): the rows are cut into slices of at most 2^29 pairs, every slice is scored, listed and sorted on
// its own — scratch stays at 24 B x 2^29 = 12.9 GB whatever the store — and its best k_eff hits per result group are merged
// into the running result on the host (two sorted lists, the result order is total).  The running result's k-th best score per
// group is a lower bound of the final one: it gates what the NEXT slices list (the prefix gate of the two-phase path, carried
// across slices), so for a take far below the pair count the later slices list next to nothing.  Counts are 64-bit throughout.
// Test hook: force_fallback bit 64 cuts at 2^14 pairs, so that small stores run many slices.
int run_large_k(ott_store* s, const float* queries, uint32_t nq, const ott_query_desc* d, bool perq, const RunPlan& pl, uint64_t k_eff,
                const uint64_t* d_mask, uint64_t mask_bits, std::vector<std::vector<ott_hit>>& lists, ott_stats& st) {
    const uint64_t slice_pairs = (s->opt.force_fallback & 64) ? (1ull << 14) : (1ull << 29);
    const uint64_t cap = pl.rows_scored * (uint64_t)nq;
    if (cap <= slice_pairs) return large_k_slice(s, queries, nq, d, perq, pl, k_eff, d_mask, mask_bits, lists, st, nullptr);
    uint64_t slice_rows = (slice_pairs / nq) & ~63ull;
    if (slice_rows < 64) slice_rows = 64;
    const uint32_t groups = perq ? nq : 1u;
    const bool tmax = d->take == OTT_TAKE_MAX;
    const CanonLess less{tmax, s->cur_tie_sh, tie_base(s)};
    struct NoDirect {  // slices return lists; the merged result goes to the caller's buffer the ordinary way
        ott_store* c;
        ott_hit* out;
        uint64_t cap;
        explicit NoDirect(ott_store* st) : c(st), out(st->direct_out), cap(st->direct_cap) { c->direct_out = nullptr; c->direct_cap = 0; }
        ~NoDirect() { c->direct_out = out; c->direct_cap = cap; }
    } no_direct(s);
    std::vector<std::vector<ott_hit>> run(groups), part, merged(groups);
    std::vector<uint32_t> gate(nq, 0u);
    bool gated = false;
    RunPlan rest = pl;
    while (rest.rows_scored) {
        RunPlan head, tail;
        split_plan(rest, slice_rows, head, tail);
        head.total_chunks = pl.total_chunks;
        head.evaluated = pl.evaluated;
        part.clear();
        const int rc = large_k_slice(s, queries, nq, d, perq, head, k_eff, d_mask, mask_bits, part, st, gated ? &gate : nullptr);
        if (rc) return rc;
        for (uint32_t g = 0; g < groups; g++) {
            std::vector<ott_hit>& a = run[g];
            const std::vector<ott_hit>& b = g < part.size() ? part[g] : merged[g];
            if (g >= part.size() || b.empty()) continue;
            if (a.empty()) {
                a = b;
            } else {
                std::vector<ott_hit>& o = merged[g];
                const uint64_t keep = std::min<uint64_t>(k_eff, (uint64_t)a.size() + b.size());
                o.resize((size_t)keep);
                std::vector<const ott_hit*> hd{a.data(), b.data()}, en{a.data() + a.size(), b.data() + b.size()};
                merge_heads(hd, en, less, o.data(), keep);
                a.swap(o);
            }
            if (a.size() > k_eff) a.resize((size_t)k_eff);
        }
        // the gates of the slices to come: a group's k-th best so far (0 = still open)
        for (uint32_t q = 0; q < nq; q++) {
            const std::vector<ott_hit>& a = run[perq ? q : 0];
            gate[q] = a.size() >= k_eff ? ord_of(a[(size_t)k_eff - 1].score, tmax) : 0u;
            gated = gated || gate[q] != 0;
        }
        rest = std::move(tail);
    }
    lists = std::move(run);
    return OTT_OK;
}

}  // namespace ott

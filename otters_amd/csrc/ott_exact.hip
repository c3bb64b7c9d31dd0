// ott_exact.hip — exact-order streaming scorer + fused wavefront top-k (gfx950).
//
// Replaces the scoring loop of VecQueryPlan::collect (src/vec.rs:222-303) with its
// dot_product / cosine_similarity / euclidean_distance_squared calls (src/vec_compute.rs:9-54)
// and the TopKCollector (src/vec_compute.rs:76-294).
//
// Arithmetic contract: scores are BIT-IDENTICAL to the reference's order of operations —
// 8 independent lane accumulators over chunks_exact(8) (separate multiply and add, no FMA),
// wide::f32x8::reduce_add, plus the sequential remainder sum, then (dot*q_inv)*v_inv for
// cosine.  This kernel is HBM-bound (0.5 flop/byte per query), so spending VALU on the exact
// order is free; what it needs is a layout in which ONE lane owns one row's 8 accumulator
// chains.  So: lane = row.  A wave streams a tile of 64 rows; every K-stage it loads
// 64 rows x 128 B with fully coalesced 16-B-per-lane loads (8 lanes cover one row's 128-B
// line), writes them to a wave-private 8 KB LDS tile with an XOR swizzle, and then each lane
// reads back ITS row's 32 B per step with conflict-free ds_read_b128.  Queries are read
// through the scalar cache (wave-uniform addresses -> s_load), so they cost no VGPRs.
// The next stage's global loads are in flight (in registers) while the current stage is
// consumed from LDS; waves are independent (no block barrier in the main loop).
//
// Top-k: each wave keeps a sorted list of k <= 64*E candidates in registers (E per lane),
// gated by the current k-th key; lanes that beat it are inserted with ballot + shuffle.
// Keys are (ord(score) << 32 | ~row, query): a total order (better score, lower row, lower
// query) so results are deterministic; the reference leaves tie order unspecified.
#include <atomic>

#include "ott_internal.h"

namespace ott {

constexpr int KC = 32;  // floats per row per stage: one 128-B line
constexpr int WAVES = 4;
constexpr int STAGE_FLOATS = 64 * KC;  // per wave: 8 KB
constexpr int EXACT_SMEM = WAVES * STAGE_FLOATS * 4;
constexpr int DUMP_QCAP = 128;  // large-k dump: survivors queued per wave between two appends (8 B key + 4 B query each)
constexpr int EXACT_SMEM_DUMP = EXACT_SMEM + WAVES * DUMP_QCAP * 12;
// Workgroups per CU of the persistent grid.  TWO since round 6 (8 waves per CU, 64 KB of row stages in flight per CU, 512 block lists for
// the merge): measured against 3, 4 (rounds 2-5), 5 and 8 on every instantiation — 1M x 128 dot top-10 89.4 + 11.8 -> 86.3 + 9.7 us
// (scoring + merge), 2M / 4M x 128 5 % / 3 % faster, 3M x 768 2 %, the 10M x 768 headline 4403 -> 4374 us, four queries per pass 4799
// -> 4552, top-500 on 1M x 128 294 -> 247, the single-query int8 sweep 1161 -> 1134 (profiles/round6/c1_latency.md).  Experiment
// builds override it: variants/build_exact.sh name "-DOTT_X_BLOCKS_PER_CU=4".
#ifndef OTT_X_BLOCKS_PER_CU
#define OTT_X_BLOCKS_PER_CU 2
#endif
constexpr int BLOCKS_PER_CU = OTT_X_BLOCKS_PER_CU;
// small-grid variant (a handful of tiles per CU: the launch is latency-bound, not bandwidth-bound): ONE wave per
// workgroup with a deep LDS-DMA ring
constexpr int SMALL_RING = 8;         // ring slots: 7 K stages in flight per wave (vmcnt counts to 63 = 7 x 8 + 7)
constexpr uint32_t SMALL_QMAX = 2048; // query floats kept in LDS
[[maybe_unused]] constexpr int EXACT_SMEM_SMALL = SMALL_RING * STAGE_FLOATS * 4 + SMALL_QMAX * 4;  // 72 KB: two workgroups per CU

// LDS-DMA piece: a wave instruction moves 8 rows x 128 B from global memory straight into a 1 KB block of LDS (lane i ->
// block base + 16*i); uniform 64-bit base in SGPRs + 32-bit lane byte offset; M0 carries the LDS byte address.  No
// "memory" clobber and no registers written: hipcc does not see a memory operation, so it inserts no wait for it — the
// kernel waits with counted `s_waitcnt vmcnt` itself (the loads retire in order).
__device__ __forceinline__ void exact_glds16(const char* sbase, uint32_t voff, uint32_t lds_addr) {
    uint32_t keep;
    lds_addr = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr);
    asm volatile(
        "s_nop 4\n\t"
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_addr)
        : "memory");
}

// Candidate order.  sh = 0: the canonical total order (better score, lower row, lower query).  sh = 3 (store option
// tie_order = reference): better score, then the reference's VISIT order — 8-row block, then query, then row within the block
// (src/vec.rs:222-303: blocks of eight rows, every query per block, lanes in order; the remainder rows all share the last
// block index, where the same three keys give query-then-row) — so that among equal scores the first one the reference's
// collector would have seen ranks first.  key = ord << 32 | ~row: key >> 3 is (ord, ~block), key & 7 is ~(row & 7).
__device__ __forceinline__ bool before(uint64_t ak, uint32_t aq, uint64_t bk, uint32_t bq, uint32_t sh) {
    const uint64_t ah = ak >> sh, bh = bk >> sh;
    return ah > bh || (ah == bh && (aq < bq || (aq == bq && ak > bk)));
}

__device__ __forceinline__ uint32_t rl32(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }
__device__ __forceinline__ uint64_t rl64(uint64_t v, int src) {
    return ((uint64_t)rl32((uint32_t)(v >> 32), src) << 32) | rl32((uint32_t)v, src);
}

__device__ __forceinline__ void wave_sync() {
    // orders this wave's LDS writes before its later LDS reads (DS ops of one wave execute in
    // order; this only stops the compiler from moving them across)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// sorted candidate list spread over a wave: position p = e*64 + lane
template <int E>
struct WaveList {
    uint64_t key[E];
    uint32_t q[E];
};

template <int E>
__device__ __forceinline__ void wl_init(WaveList<E>& L) {
#pragma unroll
    for (int e = 0; e < E; e++) {
        L.key[e] = 0;  // sentinel: worse than any real candidate (real ord >= 1)
        L.q[e] = 0xFFFFFFFFu;
    }
}

template <int E>
__device__ __forceinline__ void wl_insert(WaveList<E>& L, uint64_t xk, uint32_t xq, int lane, uint32_t sh) {
    int pos = 0;
#pragma unroll
    for (int e = 0; e < E; e++) pos += __popcll(__ballot(before(L.key[e], L.q[e], xk, xq, sh)));
#pragma unroll
    for (int e = E - 1; e >= 0; e--) {
        uint64_t upk = __shfl_up(L.key[e], 1);
        uint32_t upq = __shfl_up(L.q[e], 1);
        if (e > 0) {
            uint64_t pk = rl64(L.key[e - 1], 63);
            uint32_t pq = rl32(L.q[e - 1], 63);
            if (lane == 0) {
                upk = pk;
                upq = pq;
            }
        }
        int p = e * 64 + lane;
        if (p == pos) {
            L.key[e] = xk;
            L.q[e] = xq;
        } else if (p > pos) {
            L.key[e] = upk;
            L.q[e] = upq;
        }
    }
}

// key of the current k-th entry (position k-1)
template <int E>
__device__ __forceinline__ void wl_tau(const WaveList<E>& L, uint32_t k, uint64_t& tk, uint32_t& tq) {
    uint32_t p = k - 1;
#pragma unroll
    for (int e = 0; e < E; e++)
        if ((int)(p >> 6) == e) {
            tk = rl64(L.key[e], p & 63);
            tq = rl32(L.q[e], p & 63);
        }
}

template <int E>
__device__ __forceinline__ void wl_offer(WaveList<E>& L, uint64_t& tk, uint32_t& tq, uint32_t k, bool pass, uint64_t key,
                                         uint32_t q, int lane, uint32_t sh) {
    pass = pass && before(key, q, tk, tq, sh);
    uint64_t m = __ballot(pass);
    while (m) {
        int src = __builtin_ctzll(m);
        m &= m - 1;
        uint64_t xk = rl64(key, src);
        uint32_t xq = rl32(q, src);
        if (before(xk, xq, tk, tq, sh)) {
            wl_insert(L, xk, xq, lane, sh);
            wl_tau(L, k, tk, tq);
        }
    }
}

// Bitonic sort of one (key, q) entry per lane, best first (entries that are not `pass` become the sentinel and sort last).
__device__ __forceinline__ void wave_sort_desc(uint64_t& sk, uint32_t& sq, int lane, uint32_t sh) {
#pragma unroll
    for (int k2 = 2; k2 <= 64; k2 <<= 1) {
#pragma unroll
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            const uint64_t ok = __shfl_xor(sk, j);
            const uint32_t oq = __shfl_xor(sq, j);
            const bool mine_first = before(sk, sq, ok, oq, sh);               // my entry ranks before the partner's
            const bool want_first = ((lane & j) == 0) == ((lane & k2) == 0);  // this lane keeps the better one of the pair
            if (mine_first != want_first && !(sk == ok && sq == oq)) {
                sk = ok;
                sq = oq;
            }
        }
    }
}

// One register of 64 entries that is BITONIC (e.g. the lane-wise better halves of a descending and an ascending sequence)
// into descending order: the last six steps of the sort above.
__device__ __forceinline__ void wave_bitonic_merge_desc(uint64_t& sk, uint32_t& sq, int lane, uint32_t sh) {
#pragma unroll
    for (int j = 32; j > 0; j >>= 1) {
        const uint64_t ok = __shfl_xor(sk, j);
        const uint32_t oq = __shfl_xor(sq, j);
        const bool mine_first = before(sk, sq, ok, oq, sh);
        const bool want_first = (lane & j) == 0;
        if (mine_first != want_first && !(sk == ok && sq == oq)) {
            sk = ok;
            sq = oq;
        }
    }
}

// A SORTED register of 64 candidates (best first, sentinels behind the real ones) into the sorted list, as a block: register
// by register, the better halves of (list register, reversed block) stay, the worse halves move on to the next register, each
// half put back in order by a bitonic merge — 3 + 36 shuffles per register whatever the number of candidates, where
// inserting them one by one costs a ballot, a shift of the whole list and a new threshold EACH (round 3: a wave of a 1M-row
// store sees 500 rows, so at k = 100 a fifth of them entered its list that way: top-100 on 1M x 128 took 200 us, top-10 88).
template <int E>
__device__ __forceinline__ void wl_merge_sorted(WaveList<E>& L, uint64_t sk, uint32_t sq, int lane, uint32_t sh) {
#pragma unroll
    for (int e = 0; e < E; e++) {
        const uint64_t rk = __shfl(sk, 63 - lane);
        const uint32_t rq = __shfl(sq, 63 - lane);
        const bool mine = before(L.key[e], L.q[e], rk, rq, sh);
        uint64_t hk = mine ? L.key[e] : rk, lk = mine ? rk : L.key[e];
        uint32_t hq = mine ? L.q[e] : rq, lq = mine ? rq : L.q[e];
        wave_bitonic_merge_desc(hk, hq, lane, sh);
        wave_bitonic_merge_desc(lk, lq, lane, sh);
        L.key[e] = hk;
        L.q[e] = hq;
        sk = lk;
        sq = lq;
    }
}

// wl_offer for a tile with MANY candidates above the threshold (the first tiles of a wave, the lists of the other waves at the
// block fold): sort them once and merge the block; few candidates: one at a time as before.
constexpr int WL_BLOCK_MIN = 12;
template <int E>
__device__ __forceinline__ void wl_offer_block(WaveList<E>& L, uint64_t& tk, uint32_t& tq, uint32_t k, bool pass, uint64_t key, uint32_t q, int lane,
                                               uint32_t sh) {
    pass = pass && before(key, q, tk, tq, sh);
    if (__popcll(__ballot(pass)) < WL_BLOCK_MIN) {
        wl_offer(L, tk, tq, k, pass, key, q, lane, sh);
        return;
    }
    uint64_t sk = pass ? key : 0ull;
    uint32_t sq = pass ? q : 0xFFFFFFFFu;
    wave_sort_desc(sk, sq, lane, sh);
    wl_merge_sorted(L, sk, sq, lane, sh);
    wl_tau(L, k, tk, tq);
}

// First offer into an EMPTY one-entry-per-lane list (k <= 64): the sorted candidates ARE the list — 21 shuffle steps instead
// of up to 64 one-at-a-time insertions (a wave's first tile; for a store of one tile per wave that is the whole query).
// (Longer lists, k > 64: the 64 sorted candidates are positions 0 .. 63, the rest stays empty.  Inserting a tile's 64 rows one
// by one — at k > 64 every row of a wave's first tile is a candidate — was most of the 33 us rows8 took for a top-100 on a
// 10k-row store against 10 for a top-10.)
template <int E>
__device__ __forceinline__ void wl_fill_sorted(WaveList<E>& L, uint64_t& tk, uint32_t& tq, uint32_t k, bool pass, uint64_t key, uint32_t q, int lane,
                                               uint32_t sh) {
    uint64_t sk = pass ? key : 0ull;
    uint32_t sq = pass ? q : 0xFFFFFFFFu;
    wave_sort_desc(sk, sq, lane, sh);
    L.key[0] = (uint32_t)lane < k ? sk : 0ull;
    L.q[0] = (uint32_t)lane < k ? sq : 0xFFFFFFFFu;
#pragma unroll
    for (int e = 1; e < E; e++) {
        L.key[e] = 0ull;
        L.q[e] = 0xFFFFFFFFu;
    }
    wl_tau(L, k, tk, tq);
}

__device__ __forceinline__ bool cmp_holds(float s, uint32_t cmp, float thr) {
    // src/vec_compute.rs:56-64: ordered compares (false on NaN)
    switch (cmp) {
        case OTT_CMP_LT: return s < thr;
        case OTT_CMP_GT: return s > thr;
        case OTT_CMP_LTE: return s <= thr;
        case OTT_CMP_GTE: return s >= thr;
        case OTT_CMP_EQ: return s == thr;
        default: return true;
    }
}

typedef float v4f __attribute__((ext_vector_type(4)));

// wide::f32x8::reduce_add (see oracle/otters_oracle.h for the two orders)
__device__ __forceinline__ float reduce8(const float* l, uint32_t mode) {
    if (mode == OTT_REDUCE_SEQ4) {
        float a = __fadd_rn(__fadd_rn(__fadd_rn(l[0], l[1]), l[2]), l[3]);
        float b = __fadd_rn(__fadd_rn(__fadd_rn(l[4], l[5]), l[6]), l[7]);
        return __fadd_rn(a, b);
    }
    return __fadd_rn(__fadd_rn(__fadd_rn(l[0], l[4]), __fadd_rn(l[2], l[6])),
                     __fadd_rn(__fadd_rn(l[1], l[5]), __fadd_rn(l[3], l[7])));
}

// SMALL = the small-grid variant (single query, merged): with a tile or two per CU a launch is a chain of dependent
// latencies — per K stage the scalar-cache misses of the query loads (nothing is warm) and a ~2 us memory latency behind
// a single prefetched stage; the 24 stages of dim 768 came to 48 us whatever the corpus size.  Here a workgroup is ONE
// wave (launched with 64 threads, one tile each), the query is copied to LDS once and read back by broadcast, and the
// rows arrive by LDS-DMA through a ring of eight stages (seven in flight, no staging registers, counted waits).
// BLK: candidates enter the lists as sorted blocks where a tile brings many (wl_offer_block) — always at k > 64, at k <= 64 only
// from k = 17: the block code costs the k <= 16 instantiation registers for nothing (10M x 768 top-10: 4.45 -> 4.52-4.67 ms
// with it; 1M x 128 top-64: 119 -> 95 us), so the headline runs the kernel without it.
// I8 (round 5): the same sweep over the store's INT8 plane for ONE query — rows of 128-B stages hold 128 int8, the query's int8
// copy rides in the kernel arguments, a lane accumulates its row with v_dot4 (exact i32), and the "score" offered to the wave
// list is the APPROXIMATE (float)(q~ . v~) x row factor: the list's T best go to the exact re-score (run_i8_single).  Rows
// outside the pass's error model (flag bits 0 / 2) are always listed, ranked first.
template <bool L2, int NQ, int E, bool PERQ, bool DUMP = false, bool SMALL = false, bool BLK = (E > 1), bool I8 = false>
__global__ __launch_bounds__(SMALL ? 64 : 256) void exact_kernel(ExactParams p) {
    static_assert(!I8 || (NQ == 1 && !L2 && !PERQ && !DUMP && !SMALL), "the int8 sweep takes one query, cosine / dot, merged");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int BW = SMALL ? 1 : WAVES;  // waves per workgroup
    float* st = SMALL ? smem : smem + wave * STAGE_FLOATS;
    float* sQ = smem + SMALL_RING * STAGE_FLOATS;  // SMALL only
    constexpr int NL = PERQ ? NQ : 1;
    constexpr int KS = 64 * E;

    WaveList<E> L[NL];
    uint64_t tk[NL];
    uint32_t tq[NL];
    bool fresh[NL];  // the list has not been offered anything yet (wave-uniform)
#pragma unroll
    for (int i = 0; i < NL; i++) {
        wl_init(L[i]);
        tk[i] = 0;
        tq[i] = 0xFFFFFFFFu;
        fresh[i] = true;
    }

    const bool take_max = p.take_max != 0;
    const uint32_t nq_here = (p.nq_total - p.q0) < (uint32_t)NQ ? (p.nq_total - p.q0) : (uint32_t)NQ;
    // wave-uniform, read-only inputs are addressed through the CONSTANT address space: such loads are always scalar
    // (s_load), whereas for global pointers hipcc only selects s_load while it can prove nothing in the kernel may write
    // the memory (an inline-asm statement anywhere in the kernel ends that proof: every query value then becomes a
    // vector load + readfirstlane)
    typedef __attribute__((address_space(4))) const float* CF32;
    typedef __attribute__((address_space(4))) const uint32_t* CU32;
    typedef __attribute__((address_space(4))) const ott_run* CRUN;
    // embedded inputs live in the kernel-argument segment (constant address space), ExactParams being the only argument
    typedef __attribute__((address_space(4))) const char* CCH;
    const CCH karg = (CCH)__builtin_amdgcn_kernarg_segment_ptr();
    const CF32 Q = p.embedded ? (CF32)(karg + __builtin_offsetof(ExactParams, qemb)) : (CF32)(p.queries + (size_t)p.q0 * p.dimq);
    const CU32 tile_prefix = p.embedded ? (CU32)(karg + __builtin_offsetof(ExactParams, eprefix)) : (CU32)p.tile_prefix;
    const CRUN runs = p.embedded ? (CRUN)(karg + __builtin_offsetof(ExactParams, eruns)) : (CRUN)p.runs;
    float qinv[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) qinv[q] = (uint32_t)q < nq_here ? (p.embedded ? p.eqinv : p.qinv[p.q0 + q]) : 0.0f;
    // large-k path, second phase: a pair whose score ordinal is below its query's gate — the k-th best of the rows scored in
    // the first phase, a lower bound of the final k-th — cannot be in the result and is not listed (0 = open)
    uint32_t dgate[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) dgate[q] = (DUMP && p.dump_gate != nullptr && (uint32_t)q < nq_here) ? p.dump_gate[p.q0 + q] : 0u;
    // DUMP: a wave queues its survivors in LDS and appends them with ONE returning atomic on the list's cursor per flush (queue
    // full, or the wave is done).  One atomic per tile that holds a survivor — 70-150k of them on ONE address at 10M rows, each
    // with its wave waiting for the answer — cost the sweep 0.5-0.9 ms of 4.4 (kernel trace: 4.44 ms with nothing listed, 5.37
    // with 100k pairs listed from half of the tiles).
    uint64_t* dq_key = reinterpret_cast<uint64_t*>(smem + WAVES * STAGE_FLOATS) + wave * DUMP_QCAP;
    uint32_t* dq_q = reinterpret_cast<uint32_t*>(smem + WAVES * STAGE_FLOATS + WAVES * DUMP_QCAP * 2) + wave * DUMP_QCAP;
    uint32_t dqn = 0;  // wave-uniform
    auto dq_flush = [&]() {
        if (dqn == 0) return;
        wave_sync();
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(p.dump_cursor, (unsigned long long)dqn);
        base = rl64(base, 0);
        for (uint32_t i = lane; i < dqn; i += 64) {
            const uint64_t at = base + i;
            if (at < p.dump_cap) {
                p.dump_keys[at] = dq_key[i];
                p.dump_q[at] = dq_q[i];
            }
        }
        wave_sync();
        dqn = 0;
    };

    if constexpr (SMALL) {
        for (uint32_t i = threadIdx.x; i < p.dimq; i += 64 * BW) sQ[i] = Q[i];
        __syncthreads();
    }
    const uint32_t gw = blockIdx.x * BW + wave, nw = gridDim.x * BW;
    const int sw = (lane >> 1) & 7;
    const uint32_t nstages = (p.ld + KC - 1) / KC;
    const int lrow = lane >> 3;          // row within an 8-row load group
    const int lslot = lane & 7;          // 16-B slot within the 128-B line
    // its float offset, kept inside a short row (dim < 29): the staging loads are unconditional, so a slot past the row's
    // end must not make the LAST row of the store read past the allocation
    const uint32_t lsl4 = ((uint32_t)lslot * 4 < p.ld) ? (uint32_t)lslot * 4 : 0u;

    for (uint32_t t = gw; t < p.n_tiles; t += nw) {
        // tile -> run of surviving chunks (wave-uniform scalar search)
        uint32_t lo = 0, hi = p.n_runs;
        while (hi - lo > 1) {
            uint32_t mid = (lo + hi) >> 1;
            if (tile_prefix[mid] <= t) lo = mid;
            else hi = mid;
        }
        ott_run run;
        run.start = runs[lo].start;
        run.count = runs[lo].count;
        const uint64_t off = (uint64_t)(t - tile_prefix[lo]) * 64;
        const uint64_t row0 = run.start + off;
        const uint32_t cnt = (run.count - off) < 64 ? (uint32_t)(run.count - off) : 64u;
        const uint64_t my_row = row0 + lane;
        bool valid = (uint32_t)lane < cnt;
        if (p.row_mask != nullptr && valid && my_row < p.row_mask_bits)
            valid = (p.row_mask[my_row >> 6] >> (my_row & 63)) & 1;  // src/vec.rs:231-237
        if (__ballot(valid) == 0) continue;  // whole tile masked: its rows are never read

        // the row's inverse norm is fetched now and used after the K loop: its latency hides behind the stages (after the
        // loop it was a ~2 us bubble per tile, which shows at small dims where a tile is only a few stages long)
        float vinv = 0.0f;
        if (p.metric == OTT_METRIC_COSINE && valid) vinv = p.inv[my_row];
        float i8_rf = 0.0f;     // I8: the row factor s_v [x 1/||v||] x s_Q
        bool i8_forced = false; // I8: a row outside the error model
        if constexpr (I8) {
            if (valid) {
                i8_rf = ((p.metric == OTT_METRIC_COSINE ? vinv : 1.0f) * p.i8_scale[my_row]) * p.i8_qscale;
                i8_forced = (p.flag[my_row] & 5u) != 0;
            }
        }
        int iacc[2] = {0, 0};   // I8: two independent v_dot4 chains
        float acc[NQ][8];
        float tail[NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            tail[q] = 0.0f;
#pragma unroll
            for (int l = 0; l < 8; l++) acc[q][l] = 0.0f;
        }

        v4f R[8];
        const float* rp[8];
        bool rok[8];
        // SMALL: LDS-DMA.  The XOR swizzle is applied on the SOURCE side (the lane that owns physical slot `lslot` of
        // row `lrow` fetches logical slot lslot ^ f(row)), rows past a short tile's end are clamped to its last row and
        // a column group past `ld` to column 0: what lands there is never read (invalid lanes are never offered, and
        // the arithmetic stops at `dim`).
        const uint32_t lds_ring = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)st;
        const char* dma_base = reinterpret_cast<const char*>(p.rows + row0 * (uint64_t)p.ld);
        uint32_t dma_row[8];
        if constexpr (SMALL) {
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const uint32_t row = 8 * m + lrow;
                dma_row[m] = (row < cnt ? row : cnt - 1) * p.ld;
            }
        } else {
            // Branch-free staging: every load is always issued (rows past a short tile's end are clamped to its last row,
            // a column group past `ld` in the last stage re-reads stage 0) and the out-of-range values are zeroed when they
            // are written to LDS.  With the loads under `if`s the compiler split them into basic blocks, one per load.
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const uint32_t row = 8 * m + lrow;
                rok[m] = row < cnt;
                rp[m] = p.rows + (row0 + (rok[m] ? row : cnt - 1)) * (uint64_t)p.ld + lsl4;
            }
        }
        auto dma_stage = [&](uint32_t s) {
            const uint32_t slot_e = lslot ^ (lrow >> 1);
#pragma unroll
            for (int m = 0; m < 8; m++) {
                const uint32_t col = s * KC + ((m & 1) ? (slot_e ^ 4) : slot_e) * 4;
                exact_glds16(dma_base, (dma_row[m] + (col < p.ld ? col : 0u)) * 4u,
                             lds_ring + (uint32_t)((s % SMALL_RING) * STAGE_FLOATS * 4 + m * 1024));
            }
        };
        auto load_stage = [&](uint32_t s) {
            const uint32_t soff = (s * KC + lslot * 4 < p.ld) ? s * KC : 0u;
#pragma unroll
            for (int m = 0; m < 8; m++) {
                // non-temporal: the corpus is streamed once per pass; keeping it out of the way of L2 / Infinity Cache
                // replacement is worth +11 % on MI355X (6.17 -> 6.86 TB/s at 10M x 768)
                R[m] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(rp[m] + soff));
            }
        };

        if constexpr (SMALL) {
            for (uint32_t s = 0; s < (uint32_t)(SMALL_RING - 1) && s < nstages; s++) dma_stage(s);
        } else {
            load_stage(0);
        }
        for (uint32_t s = 0; s < nstages; s++) {
            const float* sst = st;
            if constexpr (SMALL) {
                // issued so far: stages 0 .. s+RING-2.  Stage s has landed when at most the later stages' pieces (8 each,
                // retiring in order) are outstanding; then stage s+RING-1 goes into the slot stage s-1 was read from
                const uint32_t later = (nstages - 1 - s) < (uint32_t)(SMALL_RING - 2) ? (nstages - 1 - s) : (uint32_t)(SMALL_RING - 2);
                switch (later) {
                    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                    case 1: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                    case 2: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
                    case 3: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
                    case 4: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
                    case 5: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
                    default: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
                }
                if (s + SMALL_RING - 1 < nstages) dma_stage(s + SMALL_RING - 1);
                sst = st + (s % SMALL_RING) * STAGE_FLOATS;
            } else {
                const bool cok = s * KC + lslot * 4 < p.ld;
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    const int row = 8 * m + lrow;
                    const bool ok = rok[m] & cok;
                    const v4f v = R[m];
                    *reinterpret_cast<float4*>(st + row * KC + ((lslot ^ ((row >> 1) & 7)) << 2)) =
                        make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
                }
                wave_sync();
                if (s + 1 < nstages) load_stage(s + 1);
            }

            if constexpr (I8) {
                // 128 int8 of the row per stage: eight dwords per step against eight query dwords (scalar loads), exact i32 sums
                typedef __attribute__((address_space(4))) const int* CI32;
                const CI32 qi = (CI32)Q;
#pragma unroll
                for (int j = 0; j < KC / 8; j++) {
                    const uint32_t col = s * KC + 8 * j;
                    const float4 a = *reinterpret_cast<const float4*>(sst + lane * KC + (((2 * j) ^ sw) << 2));
                    const float4 b = *reinterpret_cast<const float4*>(sst + lane * KC + (((2 * j + 1) ^ sw) << 2));
                    const int x[8] = {__float_as_int(a.x), __float_as_int(a.y), __float_as_int(a.z), __float_as_int(a.w),
                                      __float_as_int(b.x), __float_as_int(b.y), __float_as_int(b.z), __float_as_int(b.w)};
#pragma unroll
                    for (int l = 0; l < 8; l++) iacc[l & 1] = __builtin_amdgcn_sdot4(x[l], qi[col + l], iacc[l & 1], false);
                }
            } else {
#pragma unroll
            for (int j = 0; j < KC / 8; j++) {
                const uint32_t col = s * KC + 8 * j;
                if (col < p.dim) {
                    const float4 a = *reinterpret_cast<const float4*>(sst + lane * KC + (((2 * j) ^ sw) << 2));
                    const float4 b = *reinterpret_cast<const float4*>(sst + lane * KC + (((2 * j + 1) ^ sw) << 2));
                    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                    if (col + 8 <= p.dim) {
                        // one chunks_exact(8) step: acc = acc + (q * v)   (vec_compute.rs:12-13, 39-42)
                        // every slot of the NQ-wide pass is computed (the query block is zero padded to a multiple of
                        // 8 rows): no per-query branch, so the scalar query loads of a step are issued together
#pragma unroll
                        for (int q = 0; q < NQ; q++) {
                            const CF32 qp = Q + (size_t)q * p.dimq + col;
#pragma unroll
                            for (int l = 0; l < 8; l++) {
                                float qv;
                                if constexpr (SMALL) qv = sQ[col + l];  // NQ == 1: broadcast LDS read
                                else qv = qp[l];
                                float pr;
                                if (L2) {
                                    const float d = __fsub_rn(qv, x[l]);
                                    pr = __fmul_rn(d, d);
                                } else {
                                    pr = __fmul_rn(qv, x[l]);
                                }
                                acc[q][l] = __fadd_rn(acc[q][l], pr);
                            }
                        }
                    } else {
                        // remainder: sequential sum of the last dim%8 products (vec_compute.rs:15-21, 44-53)
                        const uint32_t nt = p.dim - col;
#pragma unroll
                        for (int q = 0; q < NQ; q++) {
                            const CF32 qp = Q + (size_t)q * p.dimq + col;
#pragma unroll
                            for (int l = 0; l < 7; l++) {
                                if ((uint32_t)l < nt) {
                                    float qv;
                                    if constexpr (SMALL) qv = sQ[col + l];  // NQ == 1: broadcast LDS read
                                    else qv = qp[l];
                                    float pr;
                                    if (L2) {
                                        const float d = __fsub_rn(qv, x[l]);
                                        pr = __fmul_rn(d, d);
                                    } else {
                                        pr = __fmul_rn(qv, x[l]);
                                    }
                                    tail[q] = __fadd_rn(tail[q], pr);
                                }
                            }
                        }
                    }
                }
            }
            }
            wave_sync();
        }

        // scores -> filter -> top-k gate
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            if ((uint32_t)q < nq_here) {
                float s = __fadd_rn(reduce8(acc[q], p.reduce), tail[q]);
                if (p.metric == OTT_METRIC_COSINE) s = __fmul_rn(__fmul_rn(s, qinv[q]), vinv);  // vec_compute.rs:31
                bool pass = valid && !(s != s) && cmp_holds(s, p.cmp, p.thr);  // NaN dropped: vec_compute.rs:237
                if constexpr (I8) {
                    // the APPROXIMATE score (the host relaxed the filter by the pass's error bound); a row outside the error
                    // model is listed whatever it scores, ahead of everything
                    s = (float)(iacc[0] + iacc[1]) * i8_rf;
                    pass = valid && !(s != s) && cmp_holds(s, p.cmp, p.thr);
                    if (i8_forced) {
                        s = take_max ? __builtin_inff() : -__builtin_inff();
                        pass = valid;
                    }
                }
                // (flat: every passing score ranks the same — as 0.0, a value the hit lists can carry — so the list keeps the FIRST k
                // passing pairs in visit order: the fill phase of the reference's collector, src/vec_compute.rs:257-266; used by
                // the reference tie order only)
                const uint64_t key = ((uint64_t)ord_of(p.flat ? 0.0f : s, take_max) << 32) | (uint32_t)(~((uint32_t)my_row + p.tie_off));
                if (DUMP) {
                    // large k: append every passing (key, query); the device radix sort orders them afterwards
                    const bool keep = pass && (uint32_t)(key >> 32) >= dgate[q];
                    const uint64_t m = __ballot(keep);
                    if (m) {
                        const uint32_t c = (uint32_t)__popcll(m);
                        if (dqn + c > DUMP_QCAP) dq_flush();
                        if (keep) {
                            const uint32_t at = dqn + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                            dq_key[at] = key;
                            dq_q[at] = p.q0 + q;
                        }
                        dqn += c;
                    }
                } else {
                    constexpr int li_max = NL - 1;
                    const int li = PERQ ? q : 0;
                    const int lx = li <= li_max ? li : 0;
                    if (fresh[lx]) {  // (wave-uniform) nothing in this list yet
                        wl_fill_sorted(L[lx], tk[lx], tq[lx], p.k, pass, key, p.q0 + q, lane, p.tie_sh);
                        fresh[lx] = false;
                        continue;
                    }
                    if constexpr (!BLK) wl_offer(L[lx], tk[lx], tq[lx], p.k, pass, key, p.q0 + q, lane, p.tie_sh);
                    else wl_offer_block(L[lx], tk[lx], tq[lx], p.k, pass, key, p.q0 + q, lane, p.tie_sh);
                }
            }
        }
    }

    if (DUMP) {
        dq_flush();
        return;
    }
    // block merge: waves 1..3 publish a list to LDS, wave 0 folds it in, then writes the block list
    Cand* sl = reinterpret_cast<Cand*>(smem);
#pragma unroll
    for (int i = 0; i < NL; i++) {
        __syncthreads();
        if (wave > 0) {
#pragma unroll
            for (int e = 0; e < E; e++) {
                Cand c;
                c.key = L[i].key[e];
                c.q = L[i].q[e];
                c.pad = 0;
                sl[(wave - 1) * KS + e * 64 + lane] = c;
            }
        }
        __syncthreads();
        if (wave == 0) {
            for (int w = 0; w < BW - 1; w++) {
#pragma unroll
                for (int e = 0; e < E; e++) {
                    const uint32_t ppos = e * 64 + lane;
                    const Cand c = sl[w * KS + ppos];
                    // (a register of another wave's list is already sorted: entries past k or empty become sentinels, which sort last)
                    const bool real = ppos < p.k && c.key != 0;
                    if constexpr (!BLK) {
                        wl_offer(L[i], tk[i], tq[i], p.k, real, c.key, c.q, lane, p.tie_sh);
                    } else {
                        const bool pass = real && before(c.key, c.q, tk[i], tq[i], p.tie_sh);
                        if (__popcll(__ballot(pass)) < WL_BLOCK_MIN) {
                            wl_offer(L[i], tk[i], tq[i], p.k, pass, c.key, c.q, lane, p.tie_sh);
                        } else {
                            wl_merge_sorted(L[i], real ? c.key : 0ull, real ? c.q : 0xFFFFFFFFu, lane, p.tie_sh);
                            wl_tau(L[i], p.k, tk[i], tq[i]);
                        }
                    }
                }
            }
            Cand* dst;
            if (PERQ) dst = p.lists + ((size_t)(p.q0 + i) * gridDim.x + blockIdx.x) * p.list_stride;
            else dst = p.lists + (size_t)blockIdx.x * p.list_stride;
            if (!PERQ || (uint32_t)i < nq_here) {
#pragma unroll
                for (int e = 0; e < E; e++) {
                    Cand c;
                    c.key = L[i].key[e];
                    c.q = L[i].q[e];
                    c.pad = 0;
                    dst[e * 64 + lane] = c;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// rows8: the small-store kernel, EIGHT LANES PER ROW (single query, merged, k <= 128).
//
// A store of a few thousand rows is one launch of latencies, not a stream: with lane = row a 64-row tile is ONE wave that
// issues all of the tile's 768 multiply-adds per lane on one SIMD (a 10k-row store is 157 such waves on 256 CUs: three
// SIMDs in four idle, 37-48 us whatever was tried on the memory side).  Here lane l of an 8-lane group owns accumulator
// chain l of the reference's f32x8 (src/vec_compute.rs:9-22: acc[l] += q[8j + l] * v[8j + l], strictly in j order), so
// a wave scores 8 rows with an eighth of the dependent work per lane, a 64-row tile is a WORKGROUP of 8 waves spread
// over the CU's four SIMDs, and the horizontal sum is three cross-lane adds in exactly wide's AVX order
// ((l0+l4)+(l2+l6))+((l1+l5)+(l3+l7)) — each lane computes the same tree up to operand order, and IEEE addition
// commutes bit for bit.  Same bits as exact_kernel by construction; the tests hold all three variants to the oracle.
// Rows are read straight from global memory, one float per lane and step (a wave instruction touches 8 rows x 32 B; the
// four steps that share a 128-B line hit in the vector L1), the query sits in LDS and is read by broadcast.
// Epilogue: the 64 scores of the tile go through LDS to wave 0, which builds the block list exactly like the other variants.
// ---------------------------------------------------------------------------------------------
constexpr int R8_WAVES = 8;
constexpr int R8_UNROLL = 12;
constexpr int R8_SMEM_MAX = (8 * (int)SMALL_QMAX + 8 * 64 + 64) * 4;  // 8 queries of 2048 floats + scores: 67.8 KB

// NQ queries share a pass (1, 2, 4 or 8: small batches on small stores; each lane then carries NQ accumulators for its chain),
// PERQ = one list per query instead of one merged list.  Dynamic LDS: [NQ x dimq query floats | NQ x 64 scores | 64 validity words].
template <bool L2, int E, int NQ, bool PERQ>
__global__ __launch_bounds__(64 * R8_WAVES) void exact_rows8_kernel(ExactParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sQ = smem;                                 // [NQ][dimq]
    float* sS = smem + (size_t)NQ * p.dimq;           // [NQ][64]
    uint32_t* sV = reinterpret_cast<uint32_t*>(sS + NQ * 64);  // [64]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = lane >> 3, c = lane & 7;
    typedef __attribute__((address_space(4))) const float* CF32;
    typedef __attribute__((address_space(4))) const uint32_t* CU32;
    typedef __attribute__((address_space(4))) const ott_run* CRUN;
    typedef __attribute__((address_space(4))) const char* CCH;
    const CCH karg = (CCH)__builtin_amdgcn_kernarg_segment_ptr();
    const CF32 Q = p.embedded ? (CF32)(karg + __builtin_offsetof(ExactParams, qemb)) : (CF32)(p.queries + (size_t)p.q0 * p.dimq);
    const CU32 tile_prefix = p.embedded ? (CU32)(karg + __builtin_offsetof(ExactParams, eprefix)) : (CU32)p.tile_prefix;
    const CRUN runs = p.embedded ? (CRUN)(karg + __builtin_offsetof(ExactParams, eruns)) : (CRUN)p.runs;
    const uint32_t nq_here = (p.nq_total - p.q0) < (uint32_t)NQ ? (p.nq_total - p.q0) : (uint32_t)NQ;
    float qinv[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) qinv[q] = (uint32_t)q < nq_here ? (p.embedded ? p.eqinv : p.qinv[p.q0 + q]) : 0.0f;
    // (the uploaded query block is zero padded to a multiple of 8 queries: rows past nq_here read as zeros)
    for (uint32_t i = threadIdx.x; i < (uint32_t)NQ * p.dimq; i += 64 * R8_WAVES) sQ[i] = Q[i];

    // tile -> run of surviving chunks (wave-uniform scalar search), as in exact_kernel
    const uint32_t t = blockIdx.x;
    uint32_t lo = 0, hi = p.n_runs;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (tile_prefix[mid] <= t) lo = mid;
        else hi = mid;
    }
    const uint64_t rstart = runs[lo].start, rcount = runs[lo].count;
    const uint64_t off = (uint64_t)(t - tile_prefix[lo]) * 64;
    const uint64_t row0 = rstart + off;
    const uint32_t cnt = (rcount - off) < 64 ? (uint32_t)(rcount - off) : 64u;
    const uint32_t lrow = 8u * (uint32_t)wave + (uint32_t)grp;  // this lane group's row within the tile
    bool valid = lrow < cnt;
    const uint64_t my_row = row0 + (valid ? lrow : cnt - 1);     // loads of a row past the tile's end are clamped, never skipped
    if (p.row_mask != nullptr && valid && my_row < p.row_mask_bits) valid = (p.row_mask[my_row >> 6] >> (my_row & 63)) & 1;  // src/vec.rs:231-237
    const float* rp = p.rows + my_row * (uint64_t)p.ld;
    float vinv = 0.0f;
    if (p.metric == OTT_METRIC_COSINE) vinv = p.inv[my_row];
    __syncthreads();  // the queries are in LDS

    // chain c of the row, for every query of the pass: acc = acc + q[8j + c] * v[8j + c], j ascending (vec_compute.rs:12-13, 39-42)
    const uint32_t full = p.dim >> 3;
    float acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) acc[q] = 0.0f;
    auto step = [&](uint32_t jj, float xv) {
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const float qv = sQ[(uint32_t)q * p.dimq + 8 * jj + c];
            float pr;
            if (L2) {
                const float d = __fsub_rn(qv, xv);
                pr = __fmul_rn(d, d);
            } else {
                pr = __fmul_rn(qv, xv);
            }
            acc[q] = __fadd_rn(acc[q], pr);
        }
    };
    uint32_t j = 0;
    for (; j + R8_UNROLL <= full; j += R8_UNROLL) {
        float x[R8_UNROLL];
#pragma unroll
        for (int u = 0; u < R8_UNROLL; u++) x[u] = rp[8 * (j + u) + c];
#pragma unroll
        for (int u = 0; u < R8_UNROLL; u++) step(j + u, x[u]);
    }
    for (; j < full; j++) step(j, rp[8 * j + c]);
    // remainder: sequential sum of the last dim % 8 products (vec_compute.rs:15-21, 44-53); every lane of the group computes it
    float tail[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) tail[q] = 0.0f;
    const uint32_t nt = p.dim & 7u;
    for (uint32_t l = 0; l < nt; l++) {
        const float xv = rp[8 * full + l];
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const float qv = sQ[(uint32_t)q * p.dimq + 8 * full + l];
            float pr;
            if (L2) {
                const float d = __fsub_rn(qv, xv);
                pr = __fmul_rn(d, d);
            } else {
                pr = __fmul_rn(qv, xv);
            }
            tail[q] = __fadd_rn(tail[q], pr);
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        // wide::f32x8::reduce_add across the group's eight lanes
        float sum;
        if (p.reduce == OTT_REDUCE_SEQ4) {
            const int b = lane & ~7;
            float l8[8];
#pragma unroll
            for (int i = 0; i < 8; i++) l8[i] = __shfl(acc[q], b + i);
            sum = reduce8(l8, OTT_REDUCE_SEQ4);
        } else {
            const float s1 = __fadd_rn(acc[q], __shfl_xor(acc[q], 4));  // l_c + l_{c^4}
            const float s2 = __fadd_rn(s1, __shfl_xor(s1, 2));          // (l0+l4)+(l2+l6) on even-pair lanes, (l1+l5)+(l3+l7) on the others
            sum = __fadd_rn(s2, __shfl_xor(s2, 1));
            // lanes with c odd hold ((l1+l5)+(l3+l7)) + ((l0+l4)+(l2+l6)): the same value (a + b == b + a bit for bit)
        }
        float sc = __fadd_rn(sum, tail[q]);
        if (p.metric == OTT_METRIC_COSINE) sc = __fmul_rn(__fmul_rn(sc, qinv[q]), vinv);  // vec_compute.rs:31
        if (c == 0) sS[q * 64 + lrow] = sc;
    }
    if (c == 0) sV[lrow] = valid ? 1u : 0u;
    __syncthreads();

    // lane = row of the tile: filter, key, block list(s) (as exact_kernel's epilogue for a wave's first tile).  Merged: wave 0
    // folds the NQ x 64 candidates into one list; per query: wave q builds query q's list
    const bool take_max = p.take_max != 0;
    const bool ok = sV[lane] != 0;
    const uint64_t row = row0 + lane;
    auto cand_of = [&](int q, bool& pass, uint64_t& key) {
        const float s = sS[q * 64 + lane];
        pass = ok && !(s != s) && cmp_holds(s, p.cmp, p.thr);  // NaN dropped: vec_compute.rs:237
        key = ((uint64_t)ord_of(p.flat ? 0.0f : s, take_max) << 32) | (uint32_t)(~((uint32_t)row + p.tie_off));
    };
    if (p.dump_keys != nullptr) {
        // large-k / default-take path on small stores (ott_sort.hip): every passing (key, query) pair of the tile is appended
        // behind the cursor instead of entering a top-k list — wave q lists query q's 64 rows (NQ <= 8 waves), and the WORKGROUP
        // takes its place in the list with ONE atomic for all its queries (one per (tile, query) queued 628 atomics on one
        // address behind a 10k-row sweep of four queries: 44 us of a 54-us kernel)
        static_assert(NQ <= R8_WAVES, "one wave per query of the pass");
        __shared__ uint32_t sDC[R8_WAVES];
        __shared__ unsigned long long sDB;
        bool pass = false;
        uint64_t key = 0;
        if (wave < NQ && (uint32_t)wave < nq_here) {
            cand_of(wave, pass, key);
            // second phase of the two-phase large-k path: pairs below their query's gate cannot be in the result (see exact_kernel)
            if (p.dump_gate != nullptr) pass = pass && (uint32_t)(key >> 32) >= p.dump_gate[p.q0 + (uint32_t)wave];
        }
        const unsigned long long m = __ballot(pass);
        const uint32_t cnt_q = (uint32_t)__popcll(m);
        if (lane == 0) sDC[wave] = cnt_q;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t total = 0;
#pragma unroll
            for (int w = 0; w < R8_WAVES; w++) total += sDC[w];
            sDB = total ? atomicAdd(p.dump_cursor, (unsigned long long)total) : 0ull;
        }
        __syncthreads();
        if (pass) {
            unsigned long long at = sDB + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
            for (int w = 0; w < wave; w++) at += sDC[w];
            if (at < p.dump_cap) {
                p.dump_keys[at] = key;
                p.dump_q[at] = p.q0 + (uint32_t)wave;
            }
        }
        return;
    }
    if constexpr (PERQ) {
        if (wave >= NQ || (uint32_t)wave >= nq_here) return;
        bool pass;
        uint64_t key;
        cand_of(wave, pass, key);
        WaveList<E> L;
        wl_init(L);
        uint64_t tk = 0;
        uint32_t tq = 0xFFFFFFFFu;
        wl_fill_sorted(L, tk, tq, p.k, pass, key, p.q0 + wave, lane, p.tie_sh);
        Cand* dst = p.lists + ((size_t)(p.q0 + wave) * gridDim.x + blockIdx.x) * p.list_stride;
#pragma unroll
        for (int e = 0; e < E; e++) {
            Cand cd;
            cd.key = L.key[e];
            cd.q = L.q[e];
            cd.pad = 0;
            dst[e * 64 + lane] = cd;
        }
    } else {
        if (wave != 0) return;
        WaveList<E> L;
        wl_init(L);
        uint64_t tk = 0;
        uint32_t tq = 0xFFFFFFFFu;
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            if ((uint32_t)q >= nq_here) break;
            bool pass;
            uint64_t key;
            cand_of(q, pass, key);
            if (q == 0) {
                wl_fill_sorted(L, tk, tq, p.k, pass, key, p.q0 + q, lane, p.tie_sh);
            } else {
                wl_offer_block(L, tk, tq, p.k, pass, key, p.q0 + q, lane, p.tie_sh);  // (the next query's 64 scores against a list of 64: as a block)
            }
        }
        Cand* dst = p.lists + (size_t)blockIdx.x * p.list_stride;
#pragma unroll
        for (int e = 0; e < E; e++) {
            Cand cd;
            cd.key = L.key[e];
            cd.q = L.q[e];
            cd.pad = 0;
            dst[e * 64 + lane] = cd;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// final merge of partial lists -> k hits   (src/meta.rs:699-709: concat, sort, truncate)
// ---------------------------------------------------------------------------------------------
constexpr int MERGE_WAVES = 16;

// PARTIAL: first of two stages (k > 64, a thousand block lists, a few queries: one workgroup walking them all took 0.18 ms at
// k = 100 and 0.69 ms at k = 256) — workgroup (group, part) merges the lists [part * per, (part + 1) * per) of its group into
// ONE list of KS entries (same Cand format, sentinel keys behind the real ones) in `out_lists`; the second stage is this
// kernel again, not PARTIAL, over the `parts` lists of each group.
template <int E, bool PARTIAL>
__device__ __forceinline__ void merge_walk(float* smem, const Cand* lists, uint32_t n_lists, uint32_t list_stride, uint64_t group_stride, uint32_t k,
                                           uint32_t take_max, uint64_t base, ott_hit* out, uint64_t out_stride, uint64_t* counts, Cand* out_lists,
                                           uint32_t parts, uint32_t tie_sh) {
    constexpr int KS = 64 * E;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t group = PARTIAL ? blockIdx.x / parts : blockIdx.x, part = PARTIAL ? blockIdx.x % parts : 0u;
    const uint32_t per = PARTIAL ? (n_lists + parts - 1) / parts : n_lists;
    const uint32_t l_begin = part * per, l_end = (l_begin + per) < n_lists ? (l_begin + per) : n_lists;
    const Cand* gl = lists + (size_t)group * group_stride;

    WaveList<E> L;
    wl_init(L);
    uint64_t tk = 0;
    uint32_t tq = 0xFFFFFFFFu;
    // Column-wise walk: each lane owns one sorted partial list and offers its elements in order.
    // A list dies as soon as one of its elements no longer beats the running k-th key (everything
    // deeper in it is worse), so the number of dependent load rounds is the depth of the deepest
    // contributing list (a few), not the number of lists.
    for (uint32_t l0 = l_begin + (uint32_t)wave * 64; l0 < l_end; l0 += MERGE_WAVES * 64) {
        const uint32_t li = l0 + lane;
        const Cand* src = gl + (size_t)(li < l_end ? li : l_begin) * list_stride;
        bool alive = li < l_end;
        bool any = true;
        for (uint32_t depth = 0; depth < k && any; depth += 4) {
            // four consecutive entries (one 64-B line) per lane per round trip
            Cand c4[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                c4[j].key = 0;
                c4[j].q = 0xFFFFFFFFu;
                if (alive && depth + j < k) c4[j] = src[depth + j];
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const Cand c = c4[j];
                const bool pass = alive && c.key != 0 && before(c.key, c.q, tk, tq, tie_sh);
                if (__ballot(pass) == 0) {
                    any = false;
                    break;
                }
                wl_offer(L, tk, tq, k, pass, c.key, c.q, lane, tie_sh);
                alive = pass && !before(tk, tq, c.key, c.q, tie_sh);  // still in the list (it is at least the k-th)
            }
        }
    }
    Cand* sl = reinterpret_cast<Cand*>(smem);
    if (wave > 0) {
#pragma unroll
        for (int e = 0; e < E; e++) {
            Cand c;
            c.key = L.key[e];
            c.q = L.q[e];
            c.pad = 0;
            sl[(wave - 1) * KS + e * 64 + lane] = c;
        }
    }
    __syncthreads();
    if (wave == 0) {
        for (int w = 0; w < MERGE_WAVES - 1; w++) {
#pragma unroll
            for (int e = 0; e < E; e++) {
                const uint32_t ppos = e * 64 + lane;
                const Cand c = sl[w * KS + ppos];
                wl_offer(L, tk, tq, k, ppos < k && c.key != 0, c.key, c.q, lane, tie_sh);
            }
        }
        if constexpr (PARTIAL) {
#pragma unroll
            for (int e = 0; e < E; e++) {
                const uint32_t ppos = e * 64 + lane;
                Cand c;
                c.key = ppos < k ? L.key[e] : 0ull;
                c.q = L.q[e];
                c.pad = 0;
                out_lists[(size_t)blockIdx.x * KS + ppos] = c;
            }
            return;
        }
        uint32_t total = 0;
        ott_hit* o = out + (size_t)blockIdx.x * out_stride;
#pragma unroll
        for (int e = 0; e < E; e++) {
            const uint32_t ppos = e * 64 + lane;
            const bool real = ppos < k && L.key[e] != 0;
            ott_hit h;  // every slot of the KS-wide output is written: real hits first, then sentinels
            h.index = ~0ull;
            h.score = __uint_as_float(0xFFFFFFFFu);
            h.query = 0xFFFFFFFFu;
            if (real) {
                h.index = base + (uint32_t)(~(uint32_t)(L.key[e] & 0xFFFFFFFFull));
                h.score = score_of((uint32_t)(L.key[e] >> 32), take_max != 0);
                h.query = L.q[e];
            }
            o[ppos] = h;
            total += __popcll(__ballot(real));
        }
        if (lane == 0) counts[blockIdx.x] = total;
    }
}

template <int E, bool PARTIAL = false>
__global__ __launch_bounds__(1024) void merge_kernel(const Cand* lists, uint32_t n_lists, uint32_t list_stride,
                                                      uint64_t group_stride, uint32_t k, uint32_t take_max, uint64_t base,
                                                      ott_hit* out, uint64_t out_stride, uint64_t* counts, Cand* out_lists,
                                                      uint32_t parts, uint32_t tie_sh) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    merge_walk<E, PARTIAL>(smem, lists, n_lists, list_stride, group_stride, k, take_max, base, out, out_stride, counts, out_lists, parts, tie_sh);
}

// The want-th largest of n u32 values in LDS (1 <= want <= n), by the whole 1024-thread workgroup: four 8-bit radix-select
// passes from the top.  s_hist: 256 words, s_ctl: 2 words.  Ends with a barrier.
__device__ __forceinline__ uint32_t block_select_kth_largest(const uint32_t* s_vals, uint32_t n, uint32_t want, uint32_t* s_hist, uint32_t* s_ctl, uint32_t tid) {
    uint32_t prefix = 0;
    for (int pass = 3; pass >= 0; pass--) {
        if (tid < 256) s_hist[tid] = 0;
        __syncthreads();
        const uint32_t shift = (uint32_t)pass * 8;
        const uint32_t himask = pass == 3 ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (uint32_t l = tid; l < n; l += 1024) {
            const uint32_t v = s_vals[l];
            if ((v & himask) == prefix) atomicAdd(&s_hist[(v >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid < 64) {  // wave 0: the digit holding the want-th largest value, from 255 down (4 digits per lane)
            const uint32_t d0 = 255u - 4u * tid;
            const uint32_t c0 = s_hist[d0], c1 = s_hist[d0 - 1], c2 = s_hist[d0 - 2], c3 = s_hist[d0 - 3];
            uint32_t incl = c0 + c1 + c2 + c3;  // inclusive scan over the lanes
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t o = (uint32_t)__shfl_up((int)incl, off);
                if ((int)tid >= off) incl += o;
            }
            const uint32_t excl = incl - (c0 + c1 + c2 + c3);
            if (excl < want && incl >= want) {  // exactly one lane
                uint32_t acc = excl, d = d0;
                if (acc + c0 < want) { acc += c0; d = d0 - 1;
                    if (acc + c1 < want) { acc += c1; d = d0 - 2;
                        if (acc + c2 < want) { acc += c2; d = d0 - 3; } } }
                s_ctl[0] = prefix | (d << shift);
                s_ctl[1] = want - acc;
            }
        }
        __syncthreads();
        prefix = s_ctl[0];
        want = s_ctl[1];
    }
    __syncthreads();
    return prefix;
}

// merge_rank_kernel (round 3): the merge of up to MS_VMAX sorted block lists per result group WITHOUT inserting candidates one
// at a time (merge_walk's wl_offer: 55-120 us for a top-100 over 16-500 lists — more than the scoring of a small store).
//  (1) a bound: the k-th largest score ordinal among the lists' first j = ceil(2 k / n_lists) entries — the exact k-th best of
//      a subset of about 2 k candidates, so the result's k-th best is no worse (one 4-pass radix select in LDS).  (First
//      version: the ceil(k / j)-th largest of the lists' j-th entries alone — with few lists that is the WORST list's j-th
//      entry, and a top-512 over 40 lists overflowed the survivor buffer: 0.87 ms on a 10k-row store);
//  (2) every list is walked from its head while its entries reach the bound (lists are sorted: typically one to three 64-B
//      lines) and the survivors — about k plus one per list — are appended to an LDS buffer;
//  (3) a survivor's rank is the number of survivors in front of it in the result order (`before`: a total order on (key,
//      query) pairs), counted against the LDS buffer; rank < k writes the hit to its slot.
// Ties at the bound are all kept, so any tie rule (`tie_sh`) is decided by (3) alone.  More than MS_CAP survivors (a plateau
// of equal scores across many lists) or more than MS_VMAX lists: merge_walk, in the same launch.
constexpr uint32_t MS_VMAX = 4096;   // lists per result group
constexpr uint32_t MS_NVAL = 8192;   // ordinals the bound is selected from (n_lists x j <= n_lists + k)
constexpr uint32_t MS_CAP = 2048;
constexpr size_t MS_SMEM = (size_t)(MS_NVAL + 256 + 8) * 4 + (size_t)MS_CAP * sizeof(Cand);

template <int E>
__global__ __launch_bounds__(1024) void merge_rank_kernel(const Cand* lists, uint32_t n_lists, uint32_t list_stride, uint64_t group_stride, uint32_t k,
                                                           uint32_t take_max, uint64_t base, ott_hit* out, uint64_t out_stride, uint64_t* counts,
                                                           uint32_t tie_sh) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr uint32_t KS = 64 * E;
    uint32_t* s_vals = reinterpret_cast<uint32_t*>(smem);  // [MS_NVAL] the score ordinals of the lists' first j entries
    uint32_t* s_hist = s_vals + MS_NVAL;                   // [256]
    uint32_t* s_ctl = s_hist + 256;                        // [8] 0 = selected prefix, 1 = rank still wanted, 2 = survivors
    Cand* s_buf = reinterpret_cast<Cand*>(s_ctl + 8);      // [MS_CAP]
    const uint32_t tid = threadIdx.x;
    const Cand* gl = lists + (size_t)blockIdx.x * group_stride;
    const uint32_t kk = k < KS ? k : KS;  // entries of a list that can matter
    bool slow = n_lists > MS_VMAX || n_lists == 0 || kk == 0;
    if (!slow) {
        // twice the k candidates the bound needs: the k-th largest of EXACTLY k values is their minimum — the worst list's
        // j-th entry again (k = 200 over 40 lists: j = 5, 200 values, every list walked to its end, 213 us for this kernel)
        const uint32_t j2 = (2 * kk + n_lists - 1) / n_lists;
        const uint32_t j = j2 < kk ? j2 : kk;  // 1 .. kk; n_lists j < n_lists + 2 kk <= MS_NVAL
        const uint32_t n_vals = n_lists * j;
        for (uint32_t v = tid; v < n_vals; v += 1024) {
            const uint64_t key = gl[(size_t)(v / j) * list_stride + (v % j)].key;
            s_vals[v] = key != 0 ? (uint32_t)(key >> 32) : 0u;  // (a list shorter than j: ordinal 0, below every real one)
        }
        __syncthreads();
        const uint32_t prefix = block_select_kth_largest(s_vals, n_vals, kk, s_hist, s_ctl, tid);  // 0: fewer than k real entries there — no bound
        const uint32_t bound = prefix;
        if (tid == 0) s_ctl[2] = 0;
        __syncthreads();
        for (uint32_t l = tid; l < n_lists; l += 1024) {
            const Cand* src = gl + (size_t)l * list_stride;
            bool more = true;
            for (uint32_t depth = 0; depth < kk && more; depth += 4) {
                Cand c4[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    c4[i].key = 0;
                    c4[i].q = 0xFFFFFFFFu;
                    if (depth + i < kk) c4[i] = src[depth + i];
                }
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    if (!more) break;
                    if (c4[i].key == 0 || (uint32_t)(c4[i].key >> 32) < bound) {
                        more = false;
                        break;
                    }
                    const uint32_t pos = atomicAdd(&s_ctl[2], 1u);
                    if (pos < MS_CAP) s_buf[pos] = c4[i];
                }
            }
        }
        __syncthreads();
        const uint32_t C = s_ctl[2];
        if (C <= MS_CAP) {
            ott_hit* o = out + (size_t)blockIdx.x * out_stride;
            for (uint32_t i = tid; i < C; i += 1024) {
                const Cand me = s_buf[i];
                uint32_t r = 0;
#pragma unroll 8
                for (uint32_t x = 0; x < C; x++) {
                    const Cand c = s_buf[x];
                    r += before(c.key, c.q, me.key, me.q, tie_sh) ? 1u : 0u;
                }
                if (r < kk) {
                    ott_hit h;
                    h.index = base + (uint32_t)(~(uint32_t)(me.key & 0xFFFFFFFFull));
                    h.score = score_of((uint32_t)(me.key >> 32), take_max != 0);
                    h.query = me.q;
                    o[r] = h;
                }
            }
            const uint32_t total = C < kk ? C : kk;
            for (uint32_t i = total + tid; i < KS; i += 1024) {  // every slot of the KS-wide output is written: sentinels behind the hits
                ott_hit h;
                h.index = ~0ull;
                h.score = __uint_as_float(0xFFFFFFFFu);
                h.query = 0xFFFFFFFFu;
                o[i] = h;
            }
            if (tid == 0) counts[blockIdx.x] = total;
            return;
        }
        slow = true;
        __syncthreads();  // the LDS is about to be reused
    }
    merge_walk<E, false>(smem, lists, n_lists, list_stride, group_stride, k, take_max, base, out, out_stride, counts, (Cand*)nullptr, 1u, tie_sh);
}

// merge_small_kernel: the same merge for k <= 64 (one list entry per lane), i.e. every top-10 call.  What cost merge_kernel<1> its
// 24-28 us was not memory but `wl_offer`'s one-candidate-at-a-time insertion: 64 list heads per wave offered serially, then
// wave 0 folding fifteen other waves' lists the same way.  Here (1) a wave SORTS the 64 heads it fetched with a bitonic
// network over the lanes (21 shuffle steps) — that IS its initial list, and its k-th key gates everything deeper; (2) only
// lists whose head made the wave's top k are walked further (a few serial offers); (3) the sixteen wave lists are folded as a
// binary tree (four levels of k offers in parallel instead of fifteen in a row).
__global__ __launch_bounds__(64 * MERGE_WAVES) void merge_small_kernel(const Cand* lists, uint32_t n_lists, uint32_t list_stride,
                                                                        uint64_t group_stride, uint32_t k, uint32_t take_max, uint64_t base,
                                                                        ott_hit* out, uint64_t out_stride, uint64_t* counts, uint32_t tie_sh) {
    __shared__ Cand sl[MERGE_WAVES * 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const Cand* gl = lists + (size_t)blockIdx.x * group_stride;
    WaveList<1> L;
    wl_init(L);
    uint64_t tk = 0;
    uint32_t tq = 0xFFFFFFFFu;
    for (uint32_t base_l = (uint32_t)wave * 64; base_l < n_lists; base_l += MERGE_WAVES * 64) {
        const uint32_t li = base_l + lane;
        const Cand* src = gl + (size_t)(li < n_lists ? li : 0) * list_stride;
        Cand head;
        head.key = 0;
        head.q = 0xFFFFFFFFu;
        if (li < n_lists) head = src[0];
        bool alive;
        if (base_l == (uint32_t)wave * 64) {
            // first round: the wave's list is empty, so the sorted heads ARE the list.  Bitonic sort, best first
            uint64_t sk = head.key;
            uint32_t sq = head.q;
            wave_sort_desc(sk, sq, lane, tie_sh);
            L.key[0] = (uint32_t)lane < k ? sk : 0ull;
            L.q[0] = (uint32_t)lane < k ? sq : 0xFFFFFFFFu;
            wl_tau(L, k, tk, tq);
            alive = head.key != 0 && !before(tk, tq, head.key, head.q, tie_sh);  // the head made the list (it is at least the k-th)
        } else {
            const bool pass = head.key != 0 && before(head.key, head.q, tk, tq, tie_sh);
            wl_offer(L, tk, tq, k, pass, head.key, head.q, lane, tie_sh);
            alive = pass && !before(tk, tq, head.key, head.q, tie_sh);
        }
        // deeper entries of the lists still alive, four (one 64-B line) per round trip
        bool any = __ballot(alive) != 0;
        for (uint32_t depth = 1; depth < k && any; depth += 4) {
            Cand c4[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                c4[j].key = 0;
                c4[j].q = 0xFFFFFFFFu;
                if (alive && depth + j < k) c4[j] = src[depth + j];
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const Cand c = c4[j];
                const bool pass = alive && c.key != 0 && before(c.key, c.q, tk, tq, tie_sh);
                if (__ballot(pass) == 0) {
                    any = false;
                    break;
                }
                wl_offer(L, tk, tq, k, pass, c.key, c.q, lane, tie_sh);
                alive = pass && !before(tk, tq, c.key, c.q, tie_sh);
            }
        }
    }
    // binary-tree fold of the wave lists
#pragma unroll
    for (int step = 1; step < MERGE_WAVES; step <<= 1) {
        if ((wave & (2 * step - 1)) == step) {
            Cand c;
            c.key = L.key[0];
            c.q = L.q[0];
            c.pad = 0;
            sl[wave * 64 + lane] = c;
        }
        __syncthreads();
        if ((wave & (2 * step - 1)) == 0) {
            const Cand c = sl[(wave + step) * 64 + lane];
            wl_offer(L, tk, tq, k, (uint32_t)lane < k && c.key != 0, c.key, c.q, lane, tie_sh);
        }
        __syncthreads();
    }
    if (wave == 0) {
        const bool real = (uint32_t)lane < k && L.key[0] != 0;
        ott_hit h;  // every slot of the 64-wide output is written: real hits first, then sentinels
        h.index = ~0ull;
        h.score = __uint_as_float(0xFFFFFFFFu);
        h.query = 0xFFFFFFFFu;
        if (real) {
            h.index = base + (uint32_t)(~(uint32_t)(L.key[0] & 0xFFFFFFFFull));
            h.score = score_of((uint32_t)(L.key[0] >> 32), take_max != 0);
            h.query = L.q[0];
        }
        out[(size_t)blockIdx.x * out_stride + lane] = h;
        const uint32_t total = (uint32_t)__popcll(__ballot(real));
        if (lane == 0) counts[blockIdx.x] = total;
    }
}

// ---------------------------------------------------------------------------------------------
// cross-GPU merge: n_lists candidate lists of ott_hit (each best-first, sentinel padded, in
// shard order) -> k best.  Candidate id = list*list_len + pos breaks ties, which equals the
// canonical (row, query) order because shard g holds lower global rows than shard g+1.
// ---------------------------------------------------------------------------------------------
template <int E>
__device__ __forceinline__ void merge_hits_walk(float* smem, const ott_hit* lists_all, uint32_t n_lists, uint32_t n_groups, uint32_t list_len, uint32_t k,
                                                uint32_t take_max, ott_hit* out_all, uint64_t* count, size_t rank_stride) {
    // one workgroup per group (= query in PER_QUERY mode): list `li` of group g starts at ((li * n_groups) + g) * list_len,
    // the layout an all-gather of per-GPU [n_groups][list_len] blocks produces
    const uint32_t grp = blockIdx.x;
    const size_t gstride = rank_stride;  // hits between consecutive ranks' blocks (n_groups * list_len, plus any header slots)
    const ott_hit* lists = lists_all + (size_t)grp * list_len;
    constexpr int KS = 64 * E;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    WaveList<E> L;
    wl_init(L);
    uint64_t tk = 0;
    uint32_t tq = 0xFFFFFFFFu;
    for (uint32_t li = wave; li < n_lists; li += MERGE_WAVES) {
        for (uint32_t c0 = 0; c0 < list_len; c0 += 64) {
            const uint32_t pos = c0 + lane;
            bool pass = false;
            uint64_t key = 0;
            if (pos < list_len) {
                const ott_hit h = lists[(size_t)li * gstride + pos];
                pass = h.index != ~0ull && !(h.score != h.score);
                key = ((uint64_t)ord_of(h.score, take_max != 0) << 32) | (uint32_t)(~(li * list_len + pos));
            }
            if (__ballot(pass && before(key, 0, tk, tq, 0u)) == 0) break;
            wl_offer(L, tk, tq, k, pass, key, 0u, lane, 0u);
        }
    }
    Cand* sl = reinterpret_cast<Cand*>(smem);
    if (wave > 0) {
#pragma unroll
        for (int e = 0; e < E; e++) {
            Cand c;
            c.key = L.key[e];
            c.q = L.q[e];
            c.pad = 0;
            sl[(wave - 1) * KS + e * 64 + lane] = c;
        }
    }
    __syncthreads();
    if (wave == 0) {
        for (int w = 0; w < MERGE_WAVES - 1; w++) {
#pragma unroll
            for (int e = 0; e < E; e++) {
                const uint32_t ppos = e * 64 + lane;
                const Cand c = sl[w * KS + ppos];
                wl_offer(L, tk, tq, k, ppos < k && c.key != 0, c.key, c.q, lane, 0u);
            }
        }
        uint32_t total = 0;
#pragma unroll
        for (int e = 0; e < E; e++) {
            const uint32_t ppos = e * 64 + lane;
            const bool real = ppos < k && L.key[e] != 0;
            ott_hit h;
            h.index = ~0ull;
            h.score = __uint_as_float(0xFFFFFFFFu);
            h.query = 0xFFFFFFFFu;
            if (real) {
                const uint32_t id = ~(uint32_t)(L.key[e] & 0xFFFFFFFFull);
                h = lists[(size_t)(id / list_len) * gstride + id % list_len];
            }
            out_all[(size_t)grp * KS + ppos] = h;
            total += __popcll(__ballot(real));
        }
        if (lane == 0) count[grp] = total;
    }
}

// merge_hits_kernel (round 3): the same merge by bound + gather + rank (see merge_rank_kernel): every real hit of the group's
// lists whose score ordinal reaches the bound goes to an LDS buffer as (ordinal << 32 | ~candidate id), its rank is the number
// of larger keys there, rank < k writes the hit.  A world of 8 x top-100 is 800 candidates: no insertion chain on the path
// between the all-gather and the host.  Overflow of the buffer (plateaus over many shards): merge_hits_walk, same launch.
template <int E>
__global__ __launch_bounds__(1024) void merge_hits_kernel(const ott_hit* lists_all, uint32_t n_lists, uint32_t n_groups, uint32_t list_len,
                                                           uint32_t k, uint32_t take_max, ott_hit* out_all, uint64_t* count, uint32_t walk,
                                                           uint32_t rank_stride, uint32_t hdr_slots, ott_hit* hdr_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr uint32_t KS = 64 * E;
    uint32_t* s_vals = reinterpret_cast<uint32_t*>(smem);      // [MS_NVAL] the score ordinals of the lists' first j hits
    uint32_t* s_hist = s_vals + MS_NVAL;                       // [256]
    uint32_t* s_ctl = s_hist + 256;                            // [8] 0, 1 = the select's, 2 = survivors, 3 = a NaN was met
    uint64_t* s_key = reinterpret_cast<uint64_t*>(s_ctl + 8);  // [MS_CAP]
    const uint32_t tid = threadIdx.x;
    const uint32_t grp = blockIdx.x;
    const size_t gstride = rank_stride;  // hits between consecutive ranks' blocks
    const ott_hit* lists = lists_all + (size_t)grp * list_len;
    // the ranks' block headers (status of every shard's scoring, ott_comm.hip) travel behind their hits: handed to the host here
    if (grp == 0 && hdr_out != nullptr)
        for (uint32_t i = tid; i < n_lists * hdr_slots; i += 1024)
            hdr_out[i] = lists_all[(size_t)(i / hdr_slots) * gstride + (size_t)n_groups * list_len + (i % hdr_slots)];
    const uint32_t kk = k < KS ? k : KS;
    const uint32_t depth_max = list_len < kk ? list_len : kk;  // entries of a list that can matter
    if (!walk && n_lists > 0 && n_lists <= MS_VMAX && depth_max > 0 && (uint64_t)n_lists * list_len < 0xFFFFFFFFull) {
        const uint32_t j0 = (2 * kk + n_lists - 1) / n_lists;  // twice the k candidates the bound needs (see merge_rank_kernel)
        const uint32_t j = j0 < depth_max ? j0 : depth_max;
        const uint32_t n_vals = n_lists * j;  // < n_lists + 2 kk <= MS_NVAL
        for (uint32_t v = tid; v < n_vals; v += 1024) {
            const ott_hit h = lists[(size_t)(v / j) * gstride + (v % j)];
            s_vals[v] = (h.index != ~0ull && !(h.score != h.score)) ? ord_of(h.score, take_max != 0) : 0u;
        }
        if (tid == 0) {
            s_ctl[2] = 0;
            s_ctl[3] = 0;  // a NaN score inside a list (never produced by ott_query_device; the walk below would count it among a list's first j): merge_hits_walk
        }
        __syncthreads();
        // the k-th largest of those ordinals (0 = fewer than k real hits there: no bound)
        const uint32_t sel = n_vals >= kk ? block_select_kth_largest(s_vals, n_vals, kk, s_hist, s_ctl, tid) : 0u;
        if (tid == 0) s_ctl[0] = sel;
        __syncthreads();
        const uint32_t bound = s_ctl[0];
        for (uint32_t l = tid; l < n_lists; l += 1024) {
            for (uint32_t pos = 0; pos < depth_max; pos++) {
                const ott_hit h = lists[(size_t)l * gstride + pos];
                if (h.index == ~0ull) break;  // sentinels behind the real hits
                if (h.score != h.score) {
                    s_ctl[3] = 1u;
                    break;
                }
                const uint32_t o = ord_of(h.score, take_max != 0);
                if (o < bound) break;
                const uint32_t at = atomicAdd(&s_ctl[2], 1u);
                if (at < MS_CAP) s_key[at] = ((uint64_t)o << 32) | (uint32_t)(~(l * list_len + pos));
            }
        }
        __syncthreads();
        const uint32_t C = s_ctl[2];
        if (C <= MS_CAP && s_ctl[3] == 0) {
            for (uint32_t i = tid; i < C; i += 1024) {
                const uint64_t me = s_key[i];
                uint32_t r = 0;
#pragma unroll 8
                for (uint32_t x = 0; x < C; x++) r += s_key[x] > me ? 1u : 0u;
                if (r < kk) {
                    const uint32_t id = ~(uint32_t)(me & 0xFFFFFFFFull);
                    out_all[(size_t)grp * KS + r] = lists[(size_t)(id / list_len) * gstride + id % list_len];
                }
            }
            const uint32_t total = C < kk ? C : kk;
            for (uint32_t i = total + tid; i < KS; i += 1024) {
                ott_hit h;
                h.index = ~0ull;
                h.score = __uint_as_float(0xFFFFFFFFu);
                h.query = 0xFFFFFFFFu;
                out_all[(size_t)grp * KS + i] = h;
            }
            if (tid == 0) count[grp] = total;
            return;
        }
        __syncthreads();  // the LDS is about to be reused
    }
    merge_hits_walk<E>(smem, lists_all, n_lists, n_groups, list_len, k, take_max, out_all, count, gstride);
}

int launch_merge_hits(ott_store* s, const ott_hit* lists, uint32_t n_lists, uint32_t n_groups, uint32_t list_len, uint32_t k, int E,
                      bool take_max, ott_hit* out, uint64_t* count, uint32_t hdr_slots, ott_hit* hdr_out) {
    const uint32_t rank_stride = n_groups * list_len + hdr_slots;
    const size_t smem_w = (size_t)(MERGE_WAVES - 1) * 64 * E * sizeof(Cand);
    const size_t smem_r = (size_t)(MS_NVAL + 256 + 8) * 4 + (size_t)MS_CAP * 8;
    const size_t smem = smem_w > smem_r ? smem_w : smem_r;
    const uint32_t walk = s->opt.merge_walk ? 1u : 0u;
#define OTT_MH(Ev)                                                                                                   \
    if (E == Ev) {                                                                                                   \
        if (smem > 64 * 1024) {                                                                                      \
            static std::atomic<uint64_t> attr_set{0};                                                                \
            if (ott::attr_needed(attr_set, s->device)) {                                                         \
                OTT_HIP(hipFuncSetAttribute((const void*)merge_hits_kernel<Ev>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
                ott::attr_done(attr_set, s->device);                                                     \
            }                                                                                                        \
        }                                                                                                            \
        hipLaunchKernelGGL((merge_hits_kernel<Ev>), dim3(n_groups), dim3(64 * MERGE_WAVES), smem, s->stream, lists, n_lists, \
                           n_groups, list_len, k, take_max ? 1u : 0u, out, count, walk, rank_stride, hdr_slots, hdr_out); \
        OTT_HIP(hipGetLastError());                                                                                  \
        return OTT_OK;                                                                                               \
    }
    OTT_MH(1) OTT_MH(2) OTT_MH(4) OTT_MH(8)
#undef OTT_MH
    return fail(OTT_ERR_INVALID, "launch_merge_hits: bad E");
}

// ---------------------------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------------------------
int exact_grid(const ott_store* s, uint32_t n_tiles) {
    uint32_t want = (n_tiles + WAVES - 1) / WAVES;
    uint32_t cap = (uint32_t)s->n_cu * BLOCKS_PER_CU;
    if (want < 1) want = 1;
    return (int)(want < cap ? want : cap);
}

template <bool L2, int NQ, int E, bool PERQ>
static int launch_one(ott_store* s, const ExactParams& p, int grid) {
#ifdef OTT_MFMA_DEBUG_BUILD  // round 2's one-wave LDS-DMA variant (SMALL): retired in round 5 (31 us against rows8's 10 on 10k x 768), instantiated in the diagnostic build only
    if constexpr (NQ == 1 && E <= 2 && !PERQ) {
        if (p.small == 1) {
            static std::atomic<uint64_t> attr_set{0};  // > 64 KB of dynamic LDS needs the opt-in, once per DEVICE (idempotent: a race only repeats it)
            auto kern = exact_kernel<L2, NQ, E, PERQ, false, true>;
            if (ott::attr_needed(attr_set, s->device)) {
                OTT_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, EXACT_SMEM_SMALL));
                ott::attr_done(attr_set, s->device);
            }
            hipLaunchKernelGGL(kern, dim3(grid), dim3(64), EXACT_SMEM_SMALL, s->stream, p);
            OTT_HIP(hipGetLastError());
            return OTT_OK;
        }
    }
#else
    if (p.small == 1) return fail(OTT_ERR_UNSUPPORTED, "exact_small = 1 (the one-wave LDS-DMA variant) exists in the diagnostic build only");
#endif
    if constexpr (E == 1) {
        if (p.k > 16) {  // (see BLK)
            hipLaunchKernelGGL((exact_kernel<L2, NQ, E, PERQ, false, false, true>), dim3(grid), dim3(256), EXACT_SMEM, s->stream, p);
            OTT_HIP(hipGetLastError());
            return OTT_OK;
        }
    }
    hipLaunchKernelGGL((exact_kernel<L2, NQ, E, PERQ>), dim3(grid), dim3(256), EXACT_SMEM, s->stream, p);
    OTT_HIP(hipGetLastError());
    return OTT_OK;
}

template <bool L2>
static int launch_l2(ott_store* s, const ExactParams& p, int nq_tile, int E, int grid) {
    const bool perq = p.perq != 0 && nq_tile > 1;
#define OTT_CASE(NQv, Ev, PQ) \
    if (nq_tile == NQv && E == Ev && perq == PQ) return launch_one<L2, NQv, Ev, PQ>(s, p, grid);
    OTT_CASE(1, 1, false) OTT_CASE(2, 1, false) OTT_CASE(4, 1, false)
    OTT_CASE(1, 2, false) OTT_CASE(2, 2, false) OTT_CASE(4, 2, false)
    OTT_CASE(1, 4, false) OTT_CASE(1, 8, false)
    OTT_CASE(2, 1, true) OTT_CASE(4, 1, true)
    OTT_CASE(2, 2, true) OTT_CASE(4, 2, true)
#undef OTT_CASE
    return fail(OTT_ERR_INVALID, "launch_exact: no kernel for this (nq_tile, E, mode)");
}

int launch_exact_dump(ott_store* s, const ExactParams& p, int nq_tile, int grid) {
    const bool l2 = p.metric == OTT_METRIC_EUCLIDEAN;
    if (nq_tile == 1) {
        if (l2) hipLaunchKernelGGL((exact_kernel<true, 1, 1, false, true>), dim3(grid), dim3(256), EXACT_SMEM_DUMP, s->stream, p);
        else hipLaunchKernelGGL((exact_kernel<false, 1, 1, false, true>), dim3(grid), dim3(256), EXACT_SMEM_DUMP, s->stream, p);
    } else {
        if (l2) hipLaunchKernelGGL((exact_kernel<true, 4, 1, false, true>), dim3(grid), dim3(256), EXACT_SMEM_DUMP, s->stream, p);
        else hipLaunchKernelGGL((exact_kernel<false, 4, 1, false, true>), dim3(grid), dim3(256), EXACT_SMEM_DUMP, s->stream, p);
    }
    OTT_HIP(hipGetLastError());
    return OTT_OK;
}

// rows8 (p.small == 2): eight lanes per row, one 8-wave workgroup per 64-row tile, 1 / 2 / 4 / 8 queries per pass
template <bool L2>
static int launch_rows8(ott_store* s, const ExactParams& p, int nq_tile, int E, int grid) {
    const bool perq = p.perq != 0 && nq_tile > 1;
    const size_t smem = ((size_t)nq_tile * p.dimq + (size_t)nq_tile * 64 + 64) * 4;
#define OTT_R8(NQv, Ev, PQ)                                                                                           \
    if (nq_tile == NQv && E == Ev && perq == PQ) {                                                                    \
        auto kern = exact_rows8_kernel<L2, Ev, NQv, PQ>;                                                              \
        if (smem > 48 * 1024) {                                                                                       \
            static std::atomic<uint64_t> attr_set{0};                                                                 \
            if (ott::attr_needed(attr_set, s->device)) {                                                          \
                OTT_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, R8_SMEM_MAX)); \
                ott::attr_done(attr_set, s->device);                                                      \
            }                                                                                                         \
        }                                                                                                             \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * R8_WAVES), smem, s->stream, p);                                \
        OTT_HIP(hipGetLastError());                                                                                   \
        return OTT_OK;                                                                                                \
    }
    OTT_R8(1, 1, false) OTT_R8(2, 1, false) OTT_R8(4, 1, false) OTT_R8(8, 1, false)
    OTT_R8(1, 2, false) OTT_R8(2, 2, false) OTT_R8(4, 2, false) OTT_R8(8, 2, false)
    OTT_R8(2, 1, true) OTT_R8(4, 1, true) OTT_R8(8, 1, true)
    OTT_R8(2, 2, true) OTT_R8(4, 2, true) OTT_R8(8, 2, true)
#undef OTT_R8
    return fail(OTT_ERR_INVALID, "launch_exact: no rows8 kernel for this (nq_tile, E, mode)");
}

int launch_exact_i8(ott_store* s, const ExactParams& p, int E, int grid) {
    switch (E) {
        case 2: hipLaunchKernelGGL((exact_kernel<false, 1, 2, false, false, false, true, true>), dim3(grid), dim3(256), EXACT_SMEM, s->stream, p); break;
        case 4: hipLaunchKernelGGL((exact_kernel<false, 1, 4, false, false, false, true, true>), dim3(grid), dim3(256), EXACT_SMEM, s->stream, p); break;
        case 8: hipLaunchKernelGGL((exact_kernel<false, 1, 8, false, false, false, true, true>), dim3(grid), dim3(256), EXACT_SMEM, s->stream, p); break;
        default: return fail(OTT_ERR_INVALID, "launch_exact_i8: the candidate list holds 128, 256 or 512 entries");
    }
    OTT_HIP(hipGetLastError());
    return OTT_OK;
}

int launch_exact(ott_store* s, const ExactParams& p, int nq_tile, int E, int grid) {
    if (p.small == 2) return p.metric == OTT_METRIC_EUCLIDEAN ? launch_rows8<true>(s, p, nq_tile, E, grid) : launch_rows8<false>(s, p, nq_tile, E, grid);
    if (p.metric == OTT_METRIC_EUCLIDEAN) return launch_l2<true>(s, p, nq_tile, E, grid);
    return launch_l2<false>(s, p, nq_tile, E, grid);
}

int launch_merge(ott_store* s, const Cand* lists, uint32_t n_lists, uint32_t list_stride, uint64_t group_stride,
                 uint32_t groups, uint32_t k, int E, bool take_max, uint64_t base_offset, ott_hit* out_hits,
                 uint64_t out_stride, uint64_t* out_counts, uint32_t tie_sh) {
    if (E == 1 && !(s->opt.merge_rank1 != 0 && !s->opt.merge_walk && n_lists <= MS_VMAX)) {  // option merge_rank1 = 0: k <= 64 through sorted heads + tree fold (round 2's merge)
        hipLaunchKernelGGL(merge_small_kernel, dim3(groups), dim3(64 * MERGE_WAVES), 0, s->stream, lists, n_lists, list_stride, group_stride, k,
                           take_max ? 1u : 0u, base_offset, out_hits, out_stride, out_counts, tie_sh);
        OTT_HIP(hipGetLastError());
        return OTT_OK;
    }
    const size_t smem = (size_t)(MERGE_WAVES - 1) * 64 * E * sizeof(Cand);
    if (n_lists <= MS_VMAX && !s->opt.merge_walk) {  // bound, gather, rank (merge_rank_kernel); merge_walk inside it when a plateau overflows the LDS buffer
        const size_t smem_r = smem > MS_SMEM ? smem : MS_SMEM;
#define OTT_MR(Ev)                                                                                                   \
    if (E == Ev) {                                                                                                   \
        auto kern = merge_rank_kernel<Ev>;                                                                           \
        if (smem_r > 64 * 1024) {                                                                                    \
            static std::atomic<uint64_t> attr_set{0};                                                                \
            if (ott::attr_needed(attr_set, s->device)) {                                                         \
                OTT_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_r)); \
                ott::attr_done(attr_set, s->device);                                                     \
            }                                                                                                        \
        }                                                                                                            \
        hipLaunchKernelGGL(kern, dim3(groups), dim3(1024), smem_r, s->stream, lists, n_lists, list_stride, group_stride, k, take_max ? 1u : 0u, \
                           base_offset, out_hits, out_stride, out_counts, tie_sh);                                   \
        OTT_HIP(hipGetLastError());                                                                                  \
        return OTT_OK;                                                                                               \
    }
        OTT_MR(1) OTT_MR(2) OTT_MR(4) OTT_MR(8)
#undef OTT_MR
    }
    // two stages when one workgroup per query would walk hundreds of long lists on its own: 32 parts first, then their merge
    const uint32_t parts = (n_lists >= 256 && groups <= 8) ? 32u : 1u;
    Cand* mid = nullptr;
    if (parts > 1) {
        int rc = s->d_lists2.ensure((size_t)groups * parts * 64 * E * sizeof(Cand));
        if (rc) return rc;
        mid = (Cand*)s->d_lists2.p;
    }
#define OTT_M(Ev)                                                                                                     \
    if (E == Ev) {                                                                                                    \
        if (parts > 1) {                                                                                              \
            hipLaunchKernelGGL((merge_kernel<Ev, true>), dim3(groups * parts), dim3(64 * MERGE_WAVES), smem, s->stream, lists, n_lists, \
                               list_stride, group_stride, k, take_max ? 1u : 0u, base_offset, (ott_hit*)nullptr, (uint64_t)0, \
                               (uint64_t*)nullptr, mid, parts, tie_sh);                                               \
            OTT_HIP(hipGetLastError());                                                                               \
            hipLaunchKernelGGL((merge_kernel<Ev, false>), dim3(groups), dim3(64 * MERGE_WAVES), smem, s->stream, (const Cand*)mid, parts, \
                               (uint32_t)(64 * Ev), (uint64_t)parts * 64 * Ev, k, take_max ? 1u : 0u, base_offset, out_hits, \
                               out_stride, out_counts, (Cand*)nullptr, 1u, tie_sh);                                   \
        } else {                                                                                                      \
            hipLaunchKernelGGL((merge_kernel<Ev, false>), dim3(groups), dim3(64 * MERGE_WAVES), smem, s->stream, lists, n_lists, \
                               list_stride, group_stride, k, take_max ? 1u : 0u, base_offset, out_hits, out_stride,   \
                               out_counts, (Cand*)nullptr, 1u, tie_sh);                                               \
        }                                                                                                             \
        OTT_HIP(hipGetLastError());                                                                                   \
        return OTT_OK;                                                                                                \
    }
    OTT_M(1) OTT_M(2) OTT_M(4) OTT_M(8)
#undef OTT_M
    return fail(OTT_ERR_INVALID, "launch_merge: bad E");
}

}  // namespace ott

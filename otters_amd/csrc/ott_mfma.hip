// ott_mfma.hip — the batch path: candidate passes on the matrix cores + exact re-score + certification.
//
// For query batches the scoring loop of VecQueryPlan::collect (src/vec.rs:243-266: every
// 8-row block is scored against ALL queries) is a dense contraction S = V · Qᵀ.  This file computes it as a tiled GEMM
// (mfma_score_kernel) whose epilogue never materialises S: each score is compared with a per-query running threshold
// and only survivors are appended to a small per-query candidate list.  Thresholds tighten
// between geometrically growing row rounds (select_kernel), so ~T·8 candidates per query per
// round survive.  Operands, cheapest first (run_mfma's `level` / env): the 16-bit rounding of rows and queries from the
// store's hi plane (IEEE half since round 3, bf16 as the fallback: one v_mfma_f32_32x32x16_f16 / _bf16 per 16 k, half the
// bytes), split bf16 hi + lo (three per 16 k), or f32 (v_mfma_f32_32x32x2_f32).  Because MFMA sums in a different order —
// and the 16-bit passes round the operands — the final
// per-query top-T candidates are RE-SCORED in the reference's exact order of
// operations (finalize_kernel; same arithmetic as ott_exact.hip) and the result is certified:
// if any row outside the re-scored set could still reach the k-th exact score (error bound
// eps on |approx - exact|, measured for the hi pass), the query is flagged and the caller (ott_api.hip) runs it through the
// next level of the cascade and finally the exact path.  So what ott_query returns is always the reference's result, bit for bit.
//
// GEMM tile: workgroup = 8 waves (2 per SIMD), tile 256 corpus rows x BN queries.  Wide variants (BN = 64 / 128 /
// 256): waves 4 x 2, wave tile 64 x BN/2, ONE workgroup per CU.  Narrow variant (BN = 32, batches of <= 32 queries,
// HBM-bound): waves 8 x 1, wave tile 32 x 32, TWO workgroups per CU so one streams while the other is in its
// prologue / epilogue.  K is staged 128 B of each row (32 f32 / 64 bf16) at a time by LDS-DMA into an XOR-swizzled
// ring (one barrier per stage) -> conflict-free ds_read_b128 fragments.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <utility>
#include <vector>

#include "ott_internal.h"

namespace ott {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <bool MICRO, bool I8 = false> struct AccT { typedef f32x16 type; };
template <> struct AccT<true, false> { typedef f32x4 type; };
template <> struct AccT<false, true> { typedef i32x16 type; };  // int8 pass: v_mfma_i32_32x32x32_i8 accumulates in i32, exactly
typedef __attribute__((address_space(1))) void* GPTR;
typedef __attribute__((address_space(3))) void* LPTR;

#ifndef OTT_STAGE_STAMPS
#define OTT_STAGE_STAMPS 1  // diagnostic build: 0 = no stamps inside the K stages (they cost several hundred cycles per stage themselves)
#endif
constexpr int BM = 256;   // corpus rows per tile
constexpr int MKC = 32;   // k per stage
constexpr int A_FLOATS = BM * MKC;
// per-wave LDS survivor queue (entries of 8 B), sized to what the ring leaves free: micro / narrow / NB = 1 / 2 / 4
__host__ __device__ constexpr uint32_t mfma_qw(int nb) { return nb == -1 ? 128u : nb == 0 ? 200u : nb == 1 ? 256u : nb == 2 ? 160u : 288u; }
// LDS ring depth of mfma_score_kernel: narrow / micro tiles (HBM-bound: <= 32 queries) keep THREE 36-KB stages in flight per CU
// (one workgroup per CU; with two workgroups of a 2-deep ring, drained at every stage barrier, the plane streamed at 6.0 TB/s:
// what ~70 KB in flight per CU buy on this part, MI355X_MICROARCH.md "Indexed rows: gather into LDS"); 256 queries 2 x 64 KB;
// 64 / 128 queries 3 x 40 / 48 KB
__host__ __device__ constexpr int mfma_nbuf(int nb) { return nb <= 0 ? 4 : nb == 4 ? 2 : 3; }
// queries per tile BN = 64 * NB (NB = 32-wide MFMA column blocks per wave: 4, 2 or 1), so small
// batches do not pay for 256 columns; LDS per stage = (256 + BN) rows x 128 B, double buffered

// The per-query list cursors are one to a 128-B line (cnt[q * CNT_STRIDE]): packed, the 256 cursors of a batch shared 8
// cache lines, and the ~0.5 M appends of a round queued up behind those 8 lines (a 256-tile round took 248 us).
constexpr uint32_t CNT_STRIDE = 32;

struct CandEntry {
    uint32_t row;
    float score;
};

struct MfmaParams {
    const float* rows;
    const uint16_t* img;  // BF3 == 2: the store's pre-split batch image; BF3 == 3 / 4: its hi plane; BF3 == 5: its int8 plane (row pitch ldq floats' worth of bytes)
    const float* inv;
    const float* i8_scale;  // BF3 == 5: [n] the rows' quantisation scales s_v
    float i8_qscale;        // BF3 == 5: the batch's ONE query quantisation scale s_Q; folded into the row factors when a tile loads them
    const uint8_t* flag;  // [n] 1 = irregular row (non-finite / huge norm): always a candidate
    const float* Q;      // [nq_pad][ldq] zero padded, this launch's BN block starts at q_base
    const float* qinv;   // [nq_pad]
    const float* tau;    // [nq_pad] emit when score is at least as good as tau
    uint32_t* cnt;       // [nq_pad * CNT_STRIDE]
    CandEntry* cand;     // [nq_pad][cap]
    const ott_run* runs;
    const uint32_t* tile_prefix;
    const uint64_t* row_mask;
    uint64_t row_mask_bits;
    uint32_t cap;
    uint32_t ld, dim, ldq;
    uint32_t n_runs, tile_begin, tile_end;  // tiles (of BM rows) [tile_begin, tile_end) of the run list
    uint32_t q_base;
    uint32_t n_qblk;  // consecutive blocks of BN queries this launch covers (0 = 1), processed tile by tile (mfma_score_kernel)
    uint32_t coop;    // > 1: `coop` == n_qblk sibling workgroups of one XCD share a row tile, one query block each (see the kernel)
    uint32_t dense;  // 1 = first round, thresholds open: every (row, query) pair is written at slot (tile - tile_begin) * 256 + row-in-tile
                     // of its query's list (absent pairs as row = UINT32_MAX): plain stores, no cursor atomics
    uint32_t metric, take_max;  // ott_metric; for EUCLIDEAN `qinv` holds ||q||^2 and the score is ||q||^2 + ||v||^2 - 2 q.v
    float flo, fhi;  // relaxed score filter: keep flo <= s <= fhi
    uint32_t dbg_wgs;         // diagnostic build: workgroup slots of the dbg layout
    uint32_t dbg_abl;         // diagnostic build: timing ablations of mfma_score_kernel (results are then garbage).  16: the query pieces of every
                              // THIRD row tile are not issued — the L2 -> LDS query traffic per row of a 384 x 256 tile (2/3 of today's) with
                              // today's matrix loop: an upper bound on what that tile can gain.  32: no query pieces at all (the fill path with
                              // the rows alone: what ANY scheme that keeps the queries out of the per-tile fill could gain)
    unsigned long long* dbg;  // diagnostic build only (DBG = true): per-block cycle sums [prologue, K loop, epilogue, tiles]
};

// LDS-DMA piece in inline asm: hipcc does not count an asm VMEM op, so it cannot insert its conservative
// `s_waitcnt vmcnt(0)` before every later ds_read (it cannot prove the DMA's LDS destination does not alias);
// completion is tracked by the kernel's own counted waits.  saddr form: uniform 64-bit base + 32-bit lane offset;
// M0 (LDS byte address of the 1 KB block) is written in the same statement that uses it and restored after.
template <bool NT = false>
__device__ __forceinline__ void glds16(const char* sbase, uint32_t voff, uint32_t lds_addr) {
    uint32_t keep;
    lds_addr = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr);  // wave-uniform by construction; make it provably so
    {   // same for the base pointer (an "s" operand the compiler believes divergent becomes an illegal VGPR copy)
        const unsigned long long b = (unsigned long long)sbase;
        const unsigned long long lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
        const unsigned long long hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32));
        sbase = (const char*)((hi << 32) | lo);
    }
    if constexpr (NT) {  // non-temporal: for bytes this CU reads once (corpus rows)
        asm volatile(
            "s_nop 4\n\t"
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %3\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, %2 nt\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(voff), "s"(sbase), "s"(lds_addr)
            : "memory");
    } else {
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
}

// The same piece for a base the caller has made wave-uniform once per GROUP of pieces (uniform_ptr in mfma_score_kernel): no
// v_readfirstlane per piece.  (glds16's two readfirstlanes are VALU instructions: their operand has to be a VGPR, which pulled
// the whole per-piece address arithmetic onto the vector unit — 64-bit adds per piece and stage, in the matrix loop's issue slots.)
#ifndef OTT_ROWS_NT_MAX_NB
#define OTT_ROWS_NT_MAX_NB 2
#endif
template <bool NT = false>
__device__ __forceinline__ void glds16u(const char* sbase, uint32_t voff, uint32_t lds_addr) {
    uint32_t keep;
    if constexpr (NT) {
        asm volatile(
            "s_nop 4\n\t"
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %3\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, %2 nt\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(voff), "s"(sbase), "s"(lds_addr)
            : "memory");
    } else {
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
}

// max of two accumulators (the epilogue's group quick test): integer for the int8 pass, v_max_f32 without the canonicalising copy otherwise
__device__ __forceinline__ int accmax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ float accmax(float a, float b) {
    float d;
    asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// max of three without the canonicalising v_max x, x the compiler puts in front of fmaxf operands (the operands here are
// MFMA results: never signalling NaNs).  volatile: stays behind the barrier statement that follows the last MFMA
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float d;
    asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// ordering key of an APPROXIMATE score: a non-finite value (forced candidate of an irregular row) ranks first, so it
// survives every compaction and is always among the re-scored
__device__ __forceinline__ uint32_t cand_ord(float sc, bool take_max) {
    return (sc - sc == 0.0f) ? ord_of(sc, take_max) : 0xFFFFFFFFu;
}

__device__ __forceinline__ int swz(int row, int slot) { return (row * MKC) + ((slot ^ ((row >> 1) & 7)) << 2); }

// 8 waves = 2 per SIMD.  NB >= 1 (wide): arranged 4 (rows) x 2 (queries), wave tile 64 rows x 32*NB queries = 2 x NB
// MFMA blocks (NB = 4: 128 accumulator registers), one workgroup per CU with 256 registers per wave.  Two waves per
// SIMD keep the matrix pipe fed while the partner issues its LDS-DMA pieces, waits for fragments or sits at the
// stage barrier (measured with ONE wave per SIMD: 62 % MFMA busy, 27 % of wave time parked at waitcnt/barrier).
// NB == 0 (narrow, BN = 32): arranged 8 x 1, wave tile 32 x 32 = one MFMA block, one workgroup per CU on a 4-deep ring.
// NB == -1 (micro, BN = 16, batches of <= 16 queries): arranged 8 x 1, wave tile 32 x 16 = two v_mfma_f32_16x16x4_f32
// blocks (same flops per cycle as 32x32x2, half the padded work).  The narrow kernels run next to a saturated HBM, where
// the chip holds the shader clock near 1.4 GHz (rocprofv3: GRBM_GUI_ACTIVE over the dispatch time) and the 32-wide tile
// is then matrix-pipe bound at 71 % MFMA-busy; halving the padded columns puts batches of <= 16 back on the HBM roof.
template <int NB_, bool DBG = false, int BF3 = 0>
__global__ __launch_bounds__(512, 2) /* (threads, waves per SIMD) */ void mfma_score_kernel(MfmaParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr bool MICRO = NB_ == -1;
    constexpr bool NARROW = NB_ <= 0;             // narrow or micro: 8 x 1 waves, a 4-deep stage ring
    constexpr int NB = NARROW ? 1 : NB_;          // MFMA column blocks per wave
    constexpr int MB = (NARROW && !MICRO) ? 1 : 2;  // MFMA row blocks per wave
    constexpr int RB = MICRO ? 16 : 32;           // rows / queries per MFMA block
    constexpr int RPER = MICRO ? 4 : 16;          // accumulator registers per block
    constexpr int WCOLS = NARROW ? 1 : 2;         // waves along the query axis
    constexpr int BN = RB * NB * WCOLS;
    constexpr int WN = RB * NB;  // queries per wave
    constexpr int WM = RB * MB;  // rows per wave
    constexpr int STAGE_F = A_FLOATS + BN * MKC;
    constexpr int NBUF = mfma_nbuf(NB_);          // LDS ring depth: 4 x 34 / 36 KB (micro / narrow), 2 x 64 KB or 3 x 48 / 40 KB
    constexpr bool I8 = BF3 == 5;
    typedef typename AccT<MICRO, I8>::type acc_t;
    // BF3: the candidate pass runs on the bf16 matrix pipe (8x the f32 rate per instruction) with each f32 operand split
    // into bf16 hi + bf16 lo: q.v ~ qh.vh + qh.vl + ql.vh (three v_mfma_f32_32x32x16_bf16 per 16 k instead of eight
    // 32x32x2 f32; products of bf16 are exact in f32, the dropped terms are <= 3*2^-16 |q_i v_i| each).  The corpus stays
    // f32 in HBM and LDS: rows are split in registers on their way into the fragments; the queries arrive pre-split from the
    // host (per 32-k stage: 32 hi then 32 lo bf16 = the same 128 B).  The error bound the certification uses grows
    // accordingly (run_mfma); what ott_query returns is still the exact-order f32 re-score.
    // BF3 == 2: the rows come pre-split from the store's batch image (same stage layout as the queries): no conversion here.
    // BF3 == 3 (hi pass): rows and queries are their bf16 roundings only (the store's hi plane: half the bytes), a 128-B
    // row-stage holds 64 k, ONE MFMA per 16 k.  Its error bound is ~2^-8 relative (measured at build, run_mfma), so it
    // re-scores more candidates per query and certifies less often; what it cannot certify falls through to the split pass.
    // BF3 == 4: the hi pass on an IEEE-half plane (same bytes, same layout, v_mfma_f32_32x32x16_f16: products of halves are
    // exact in f32 like products of bf16).  Rows and queries carry RECIPROCAL power-of-two factors (run_mfma), so the
    // accumulators are the plain dot products and the epilogue is the bf16 pass's, instruction for instruction.  (Round 3
    // first undid the factors in the epilogue through two more kernel arguments: the two live SGPRs cost the narrow tiles
    // 15 % — 2.30 -> 2.64 ms at 32 queries, same registers, same occupancy — and went away with them.)
    // BF3 == 5 (int8 pass, round 5): rows and queries as int8 with one f32 scale per row (s_v) and ONE per batch (s_Q) — a
    // QUARTER of the f32 bytes; a 128-B row-stage holds 128 k, ONE v_mfma_i32_32x32x32_i8 per 32 k (the byte layout of the
    // fragments is the 16-bit passes': lane (l31, lh) holds bytes 32 jg + 16 lh .. +15 of its row-stage).  The i32 accumulation
    // is EXACT, so the pass's error is the quantisation alone, measured like the hi pass's.  The scales never enter the matrix
    // loop: s_v (x 1/||v|| for cosine) x s_Q is the row factor the tile loads into LDS, and the epilogue's one multiply
    // av x rf is the score — the walk over the accumulators is the 16-bit passes', plus one v_cvt_f32_i32 per accumulator.
    static_assert(!BF3 || !MICRO, "the bf16 passes use the 32x32 tiles");
    constexpr bool HI = BF3 >= 3;
    const float* __restrict__ Arows = BF3 >= 2 ? reinterpret_cast<const float*>(p.img) : p.rows;  // both: 4 B units
    const uint32_t pitchA = BF3 >= 2 ? p.ldq : p.ld;
    // [BM] per-row epilogue pair (2 KB after the ring): .x = score factor, .y = 1 for an irregular row (listed for every query)
    float2* sRF = reinterpret_cast<float2*>(smem + NBUF * STAGE_F);
    // each wave queues its tile's survivors in a private LDS strip ({score bits, query-in-tile << 16 | row-in-tile};
    // the slot comes from a ballot, no atomics) and appends them to the per-query lists in one batch after the
    // tile: a returning global atomic inside the unrolled epilogue stalls the wave ~2000 cycles each time
    constexpr uint32_t QW = mfma_qw(NB_);
    uint2* sQ = reinterpret_cast<uint2*>(sRF + BM);          // [8][QW] per-wave survivor queues
    uint32_t* sFlagW = reinterpret_cast<uint32_t*>(sQ + 8 * QW);  // [4] does the tile hold a forced row (one word per wave of the prologue)
    float* sRS = reinterpret_cast<float*>(sFlagW + 4);       // int8 pass only: [BM] the rows' scales s_v x s_Q (squared L2: the factor slot holds ||v||^2 there)
    float2* sTQ = reinterpret_cast<float2*>(sRS + (I8 ? BM : 0));  // [n_qblk * BN] {tau, qinv} of this launch's queries (last: its size varies)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WCOLS, wn = wave % WCOLS;  // wave tile origin: rows wm*WM, queries wn*WN
    const int l31 = lane & 31, lh = lane >> 5;
    const int l15 = lane & 15, l4 = lane >> 4;   // micro: 16x16x4 operand / result lanes
    const int lq = MICRO ? l15 : l31;            // query within an MFMA block owned by this lane
    const int lrow = lane >> 3, lslot = lane & 7;
    const uint32_t nstages = (p.ldq + MKC - 1) / MKC;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(LPTR)smem;  // LDS byte address of the dynamic segment

    // Query blocks: a launch covers n_qblk consecutive blocks of BN queries (>= 1; more than one only for the 256-wide tile,
    // batches of more than 256 queries).  The workgroup takes them one after the other FOR THE SAME ROW TILE, so the tile's
    // rows come from HBM once and from the caches (Infinity Cache / L2: this CU fetched them a few microseconds earlier) for
    // the other blocks, instead of one full pass over the plane per block.
    //
    // coop (= n_qblk, a full persistent grid): the blocks of a row tile go to `coop` SIBLING workgroups instead — same XCD
    // (workgroups are dealt to the 8 XCDs round robin: blockIdx % 8), same tile at the same time, one query block each for the
    // whole launch.  The first of them to ask for a row stage brings it from HBM into the XCD's L2, the others hit there a
    // moment later (whoever runs ahead pays the HBM latency and is caught up: the group keeps itself together).  With the
    // loop above the re-reads come 28 us apart, by when 32 CUs x 393 KB per XCD have long pushed the tile out of a 4 MB L2.
    uint32_t n_qblk = p.n_qblk ? p.n_qblk : 1u, q_base = p.q_base;
    uint32_t t_first = blockIdx.x, t_step = gridDim.x;
    if (p.coop > 1) {
        const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
        q_base += (slot % p.coop) * BN;
        n_qblk = 1;
        t_first = (slot / p.coop) * 8u + xcd;
        t_step = gridDim.x / p.coop;
    }
    const float* __restrict__ Qb0 = p.Q + (size_t)q_base * p.ldq;

    // Staging by LDS-DMA (global_load_lds_dwordx4): a wave-instruction moves 8 rows x 128 B
    // straight into a 1 KB block of the LDS image (lane i -> block base + 16*i).  The XOR
    // swizzle is applied on the SOURCE side: the lane that owns physical slot `lslot` of row
    // `lrow` fetches logical slot lslot ^ f(row) of that row, so each row's 128-B line is
    // still read whole.  No staging VGPRs.  Wave w stages A rows and B queries [32w, 32w+32):
    // piece m < 4 = A rows 32w + 8m .., piece m >= 4 = B queries 32w + 8(m-4) ..
    // addresses = wave-uniform 64-bit base (SGPRs) + 32-bit per-lane byte offset, so the pieces need
    // no per-lane 64-bit pointers (those spilled, and a spill reload waits on vmcnt(0) = on the DMA)
    const uint32_t slotE = lslot ^ (lrow >> 1), slotO = slotE ^ 4;  // source-side swizzle, even / odd 8-row groups
    const uint32_t offA_e = (lrow * pitchA + slotE * 4) * 4u, offA_o = (lrow * pitchA + slotO * 4) * 4u;
    const uint32_t offB_e = (lrow * p.ldq + slotE * 4) * 4u, offB_o = (lrow * p.ldq + slotO * 4) * 4u;
    // narrow: this wave's 4 query rows are 4w + lrow (lrow < 4): swizzle term ((4w + lrow) >> 1) & 7
    // micro: 2 query rows 2w + lrow (lrow < 2): swizzle term w & 7
    const uint32_t offB_n = MICRO ? (lrow * p.ldq + (lslot ^ (wave & 7)) * 4) * 4u
                                  : (lrow * p.ldq + (lslot ^ ((2 * wave + (lrow >> 1)) & 7)) * 4) * 4u;

    // a tile of BM rows inside a run of surviving chunks
    struct Tile {
        uint64_t row0;
        uint32_t cnt;
        uint32_t offA[4];  // per-lane byte offsets of the 4 A pieces (rows past a short tile's end clamped to its last row)
        const char* baseA; // SADDR: the tile's first row (wave-uniform)
    };
    // The 256-query tile of the plane passes (config 2's first level) does two things differently from the other tiles:
    // RFPRE: the next tile's row factors are fetched during this tile's last K stage (fetch_rf) — the prologue was three dependent
    //        global loads per tile with the matrix pipe idle (2.7k cycles per tile -> 0.6k);
    // SADDR: a stage's pieces take their addresses from ONE wave-uniform base per group of four + per-lane offsets (dma_group):
    //        K loop 22.4k -> 19.8k cycles per tile.
    // (also the narrower tiles of the int8 pass: 8 queries 1.335 -> 1.27-1.29 ms, 32: 1.35-1.38 -> 1.30-1.31, 64: 1.48 -> 1.43, 128: 1.77 ->
    //  1.71; the 64- / 128-query tiles of the half pass measured 1 % SLOWER with it — 2.47 -> 2.49, 2.89 -> 2.91 — and keep the
    //  per-piece addresses)
    constexpr bool TUNED = BF3 >= 3 && (NB_ == 4 || (BF3 == 5 && NB_ >= 0));
    constexpr bool ROWS_NT_U = NB_ <= OTT_ROWS_NT_MAX_NB;  // (as ROWS_NT in dma_piece: non-temporal row pieces on the HBM-bound tiles)
    constexpr bool RFPRE = TUNED, SADDR = TUNED;
    // SADDR: per-lane byte offsets of the NB query pieces from the unit's first query (piece mm: query rows 8 NB wave + 8 mm ..)
    uint32_t offBq[NB];
#pragma unroll
    for (int mm = 0; mm < NB; mm++) {
        const int brow = wave * (8 * NB) + 8 * mm;
        offBq[mm] = (((brow >> 3) & 1) ? offB_o : offB_e) + (uint32_t)brow * p.ldq * 4u;
    }
    auto locate = [&](uint32_t t, Tile& T) {
        // the run table is read through the CONSTANT address space: scalar loads, so row0 / cnt (and the DMA base addresses
        // derived from them) live in SGPRs.  Through plain global pointers hipcc falls back to vector loads here, because
        // the kernel's inline-asm DMA stops it from proving the memory is never written.
        typedef __attribute__((address_space(4))) const uint32_t* CU32;
        typedef __attribute__((address_space(4))) const ott_run* CRUN;
        const CU32 tile_prefix = (CU32)p.tile_prefix;
        const CRUN runs = (CRUN)p.runs;
        uint32_t lo = 0, hi = p.n_runs;
        while (hi - lo > 1) {
            uint32_t mid = (lo + hi) >> 1;
            if (tile_prefix[mid] <= t) lo = mid;
            else hi = mid;
        }
        const uint64_t run_start = runs[lo].start, run_count = runs[lo].count;
        const uint64_t off = (uint64_t)(t - tile_prefix[lo]) * BM;
        T.row0 = run_start + off;
        T.cnt = (run_count - off) < BM ? (uint32_t)(run_count - off) : (uint32_t)BM;
        // rows past the end of a short tile are clamped to its last row (their scores are never read): every piece is
        // always issued, which keeps the per-stage DMA count exact for the counted wait
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const uint32_t r0 = wave * 32 + 8 * m, rbase = r0 < T.cnt ? r0 : 0;
            T.offA[m] = (m & 1) ? offA_o : offA_e;
            if (rbase + lrow >= T.cnt) T.offA[m] -= (rbase + lrow - (T.cnt - 1)) * pitchA * 4u;
            if constexpr (SADDR) T.offA[m] += rbase * pitchA * 4u;  // the piece's first row goes into the lane offsets
        }
        if constexpr (SADDR) {
            const unsigned long long b = (unsigned long long)(Arows + T.row0 * pitchA);
            const unsigned long long lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
            const unsigned long long hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32));
            T.baseA = (const char*)((hi << 32) | lo);
        }
    };
    // a tile's per-row inputs of the epilogue factor (fetch_rf: plain loads, waited for by the compiler at their first use)
    struct RowIn {
        float iv, sc;
        uint32_t fl;
        bool valid;
    };
    auto fetch_rf = [&](const Tile& T) -> RowIn {
        RowIn r;
        r.iv = 1.0f;
        r.sc = 1.0f;
        r.fl = 0u;
        r.valid = false;
        if (tid < BM) {
            const uint32_t rt = tid;
            const uint64_t grow = T.row0 + rt;
            r.valid = rt < T.cnt;
            if (r.valid) {
                if (p.metric != OTT_METRIC_DOT) r.iv = p.inv[grow];
                if constexpr (I8) r.sc = p.i8_scale[grow];
                r.fl = (uint32_t)p.flag[grow];
                if (p.row_mask != nullptr && grow < p.row_mask_bits) r.valid = (p.row_mask[grow >> 6] >> (grow & 63)) & 1;
            }
        }
        return r;
    };
    auto dma_piece = [&](const Tile& T, uint32_t s, int buf, int m, const float* __restrict__ Qb) {
        float* sA = smem + buf * STAGE_F;
        float* sB = sA + A_FLOATS;
        if (m < 4) {
            const uint32_t slot = (m & 1) ? slotO : slotE;
            const uint32_t col = s * MKC + slot * 4;
            float* blk = sA + (wave * 32 + 8 * m) * MKC;
            const uint32_t r0 = wave * 32 + 8 * m;
            const uint32_t rbase = r0 < T.cnt ? r0 : 0;
            const char* ubase = reinterpret_cast<const char*>(Arows + (T.row0 + (uint64_t)rbase) * pitchA + s * MKC);
            // Non-temporal row pieces on the HBM-bound tiles (<= 128 queries): the plane is streamed once per pass, and keeping it
            // out of the way of L2 / Infinity Cache replacement is worth as much here as in exact_kernel — 10M x 768 hi pass
            // 2.71 -> 2.46 ms at 1-8 queries (6.1 -> 6.8 TB/s in the large rounds), 3.51 -> 3.29 ms at 128; split pass 5.31 ->
            // 4.86 ms at 8 queries.  The 256-query tile is not HBM-bound: no difference there (and its sibling-workgroup
            // mapping WANTS the rows in L2).
            constexpr bool ROWS_NT = NB_ <= OTT_ROWS_NT_MAX_NB;
            if (BF3 >= 2 || (s + 1) * MKC <= p.ld) {  // whole stage inside the row (wave-uniform: every stage but possibly the last; the image is padded)
                glds16<ROWS_NT>(ubase, T.offA[m], lds_base + (uint32_t)((blk - smem) * 4));
            } else if (col < p.ld) {
                glds16<ROWS_NT>(ubase, T.offA[m], lds_base + (uint32_t)((blk - smem) * 4));
            } else {
                // K padding of the last stage must be exact zeros (0 * stale data is not 0 for inf/NaN)
                *reinterpret_cast<float4*>(blk + lane * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else if (NARROW) {
            // 32 (16) query rows over 8 waves: half (a quarter of) a piece each, lanes 0..31 (0..15) = 4 (2) rows x 128 B
            if (lane < (MICRO ? 16 : 32)) {
                const int brow = wave * (MICRO ? 2 : 4);
                float* blk = sB + brow * MKC;
                const char* ubase = reinterpret_cast<const char*>(Qb + (size_t)brow * p.ldq + s * MKC);
                glds16(ubase, offB_n, lds_base + (uint32_t)((blk - smem) * 4));
            }
        } else {
            const int mm = m - 4;                       // 0 .. NB-1
            const int brow = wave * (8 * NB) + 8 * mm;  // first of the 8 query rows of this piece
            float* blk = sB + brow * MKC;
            const char* ubase = reinterpret_cast<const char*>(Qb + (size_t)brow * p.ldq + s * MKC);
            if constexpr (DBG) {  // (diagnostic build only, see MfmaParams::dbg_abl)
                // skipped outright: the counted waits then return EARLY for these tiles (fewer pieces behind the ones waited
                // for), which only makes the ablated timing more optimistic — fine for an upper bound.  (A first version
                // pointed the pieces at one fixed 1-KB block instead: 256 CUs on one L2 line, 332 ms per batch.)
                if ((p.dbg_abl & 16u) && (T.row0 / BM) % 3u == 2u) return;
                if (p.dbg_abl & 32u) return;
            }
            const uint32_t off = ((brow >> 3) & 1) ? offB_o : offB_e;
            glds16(ubase, off, lds_base + (uint32_t)((blk - smem) * 4));
        }
    };

    // SADDR: a stage's four row pieces / four query pieces as a group: ONE wave-uniform base per group (two v_readfirstlane), the
    // piece's rows in the lane offsets (plane passes only: every stage lies whole inside the padded row)
    auto uniform_ptr = [&](const char* b_) -> const char* {
        const unsigned long long b = (unsigned long long)b_;
        const unsigned long long lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
        const unsigned long long hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32));
        return (const char*)((hi << 32) | lo);
    };
    auto dma_group = [&](const Tile& T, uint32_t s, int buf, int grp, const float* __restrict__ Qb) {  // grp 0: the rows, 1: the queries
        if constexpr (SADDR) {
            if (grp == 0) {
                const char* base = uniform_ptr(T.baseA + s * (MKC * 4));
                const uint32_t l0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds_base + (uint32_t)(buf * STAGE_F + wave * 32 * MKC) * 4u));
#pragma unroll
                for (int m = 0; m < 4; m++) glds16u<ROWS_NT_U>(base, T.offA[m], l0 + (uint32_t)(8 * m * MKC) * 4u);
            } else if constexpr (NARROW) {
                dma_piece(T, s, buf, 4, Qb);  // (the 32-query tile's one query piece: half a piece per wave)
            } else {
                if constexpr (DBG) {  // (diagnostic build only, see MfmaParams::dbg_abl)
                    if ((p.dbg_abl & 16u) && (T.row0 / BM) % 3u == 2u) return;
                    if (p.dbg_abl & 32u) return;
                }
                const char* base = uniform_ptr(reinterpret_cast<const char*>(Qb) + s * (MKC * 4));
                const uint32_t l0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds_base + (uint32_t)(buf * STAGE_F + A_FLOATS + wave * (8 * NB) * MKC) * 4u));
#pragma unroll
                for (int mm = 0; mm < NB; mm++) glds16u(base, offBq[mm], l0 + (uint32_t)(8 * mm * MKC) * 4u);
            }
        } else {
#pragma unroll
            for (int m = 0; m < 4; m++) dma_piece(T, s, buf, 4 * grp + m, Qb);
        }
    };

    // The stages of ALL this workgroup's tiles form one stream through an NBUF-deep LDS ring: while stage g is
    // consumed the pieces of stage g + L are issued (L = min(NBUF-1, stages per tile)), across tile boundaries, so the
    // next tile's first stages land during this tile's epilogue.  ONE barrier per stage.  The wait is COUNTED: every
    // wave issues exactly P = 4 + NB DMA instructions per stage (rows past the tile end are clamped, never skipped), so
    // `vmcnt(P)` retires this wave's pieces of stage g and leaves the next stage's in flight across the barrier (a plain
    // __syncthreads() would drain them).
    constexpr int P = 4 + NB;
    const uint32_t L = (uint32_t)(NBUF - 1) < nstages ? (uint32_t)(NBUF - 1) : nstages;  // 1 .. 3
    uint32_t t = p.tile_begin + t_first;
    if (t >= p.tile_end) return;
    Tile cur, nxt;
    locate(t, cur);
    nxt = cur;
    // per-query threshold / factor of this launch's BN queries: read once into LDS (the epilogue then issues no vector
    // memory loads, so it never waits on the DMA already in flight for the next tile)
    for (uint32_t i = tid; i < n_qblk * BN; i += 512) sTQ[i] = make_float2(p.tau[q_base + i], p.qinv[q_base + i]);
    int cbuf = 0, nbuf = (int)(L % NBUF);  // ring slots: being consumed / being filled (L stages ahead)
    // RFPRE: the row factors' inputs are fetched ahead of the tile that uses them — here for the first tile, in the last K stage of a
    // tile for the next one (stage()).  They are issued BEFORE that stage's DMA pieces, i.e. they are older than every piece a
    // counted wait leaves in flight.
    RowIn rin;
    rin.iv = rin.sc = 1.0f;
    rin.fl = 0u;
    rin.valid = false;
    if constexpr (RFPRE) rin = fetch_rf(cur);
    for (uint32_t i = 0; i < L; i++) {
        if constexpr (SADDR) {
            dma_group(cur, i, (int)(i % NBUF), 0, Qb0);
            dma_group(cur, i, (int)(i % NBUF), 1, Qb0);
        } else {
#pragma unroll
            for (int m = 0; m < P; m++) dma_piece(cur, i, (int)(i % NBUF), m, Qb0);
        }
    }

    uint32_t qblk = 0;  // the unit of work is (row tile, query block): all blocks of a tile, then the next tile
    for (;;) {
        const uint32_t tn = t + t_step;
        const bool last_blk = qblk + 1 == n_qblk;
        const bool has_next_tile = tn < p.tile_end;
        const bool has_next = !last_blk || has_next_tile;  // another unit follows
        if (qblk == 0 && has_next_tile) locate(tn, nxt);
        const Tile& nxtA = last_blk ? nxt : cur;            // whose rows the next unit reads
        const float* __restrict__ Qcur = Qb0 + (size_t)qblk * BN * p.ldq;
        const float* __restrict__ Qnx = Qb0 + (size_t)(last_blk ? 0u : qblk + 1u) * BN * p.ldq;
        const uint32_t epi_q_base = q_base + qblk * BN;
        const float2* epi_sTQ = sTQ + qblk * BN;
        const uint64_t row0 = cur.row0;
        const uint32_t cnt = cur.cnt;

        unsigned long long t0 = 0, t1 = 0, t2 = 0, r0 = 0, dbg_wait_dma = 0, dbg_wait_bar = 0;
        if (DBG) {
            t0 = __builtin_amdgcn_s_memtime();
            r0 = __builtin_amdgcn_s_memrealtime();  // constant 100 MHz: gives the shader clock the ticks ran at
        }
        acc_t acc[MB][NB];
#pragma unroll
        for (int mb = 0; mb < MB; mb++)
#pragma unroll
            for (int nb = 0; nb < NB; nb++)
#pragma unroll
                for (int r = 0; r < RPER; r++) acc[mb][nb][r] = 0;

        if (qblk == 0) {  // (a tile's row factors serve all of its query blocks)
        __syncthreads();  // every wave has left the previous tile's epilogue: its row factors can be replaced
        // per-row epilogue factor, fetched once per tile with one coalesced load (the first version loaded the inverse
        // norm per accumulator row inside the epilogue: 32 dependent global loads per lane, ~20 % of the tile time):
        // cosine 1/||v||, squared-L2 ||v||^2, dot 1; NaN for rows past the tile end or masked out, which makes every
        // comparison in the epilogue false for them
        if (tid < BM) {
            const uint32_t rt = tid;
            const uint64_t grow = row0 + rt;
            bool valid = rt < cnt;
            if constexpr (RFPRE) valid = rin.valid;
            else if (p.row_mask != nullptr && valid && grow < p.row_mask_bits) valid = (p.row_mask[grow >> 6] >> (grow & 63)) & 1;
            if constexpr (DBG) {  // ablated tiles (MfmaParams::dbg_abl): their scores are garbage, so no row of them may pass a threshold — the
                                  // epilogue then costs its compares but appends nothing (a real tile appends ~8 T survivors per query and round)
                if (((p.dbg_abl & 16u) && (row0 / BM) % 3u == 2u) || (p.dbg_abl & 32u)) valid = false;
            }
            float f = __uint_as_float(0x7FC00000u);
            float rs = 0.0f;
            if (valid) {
                f = 1.0f;
                if (p.metric != OTT_METRIC_DOT) {
                    const float iv = RFPRE ? rin.iv : p.inv[grow];
                    f = p.metric == OTT_METRIC_COSINE ? iv : (iv != 0.0f ? 1.0f / (iv * iv) : 0.0f);
                }
                if constexpr (I8) {
                    // cosine / dot: the scales fold into the factor (score = acc x f).  Squared L2: the factor slot keeps ||v||^2 and the
                    // scales go to sRS (score = ||q||^2 + ||v||^2 - 2 (acc x rs))
                    rs = (RFPRE ? rin.sc : p.i8_scale[grow]) * p.i8_qscale;
                    if (p.metric != OTT_METRIC_EUCLIDEAN) f = (f * (RFPRE ? rin.sc : p.i8_scale[grow])) * p.i8_qscale;
                }
            }
            // irregular rows (always listed, always re-scored): bit 0 = outside every pass's error model; bit 1 = outside the half
            // hi pass's only (its one scale factor does not suit the row: hi_rows_kernel).  In the half pass the factor of any
            // irregular row becomes infinite, so that its approximate score is non-finite and ranks first in every compaction
            // (cand_ord): half cannot be trusted to give such a row even a roughly right score (bf16 and f32 operands keep
            // f32's exponent range; there the approximate score of a bit-0 row is either close or non-finite by itself)
            uint32_t fl = valid ? (RFPRE ? rin.fl : (uint32_t)p.flag[grow]) : 0u;
            fl = BF3 == 4 ? (fl & 3u) : I8 ? (fl & 5u) : (fl & 1u);  // (bit 2: outside the int8 pass's error model, i8_rows_kernel)
            if ((BF3 == 4 || I8) && fl != 0u && valid) f = __builtin_inff();  // (bit-0 rows too: a norm below 1e-18 is zero in half whatever the factor)
            sRF[rt] = make_float2(f, fl ? 1.0f : 0.0f);
            if constexpr (I8) sRS[rt] = rs;
            // does this wave's quarter of the tile hold a forced row?  (waves 0 .. 3 are whole inside `tid < BM`; the epilogue's group
            // quick test is only valid for tiles without one)
            const unsigned long long fm = __ballot(fl != 0u);
            if (lane == 0) sFlagW[wave] = fm != 0ull ? 1u : 0u;
        }
        }
        if (DBG) t1 = __builtin_amdgcn_s_memtime();
        // one K stage: wait for stage s (this wave's pieces, then everyone's), issue stage `ns` of tile TT into ring slot
        // `nbuf`, consume ring slot `cbuf`.  `later` = stages behind this one that are already in flight and stay so across
        // the barrier (0 .. NBUF - 2)
        auto stage = [&](const Tile& TT, uint32_t ns, bool more, uint32_t later, const float* __restrict__ Qp, bool rf_now) {
            unsigned long long w0 = 0;
            if (DBG && OTT_STAGE_STAMPS) w0 = __builtin_amdgcn_s_memtime();
            if (NBUF >= 4 && later >= 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * P) : "memory");
            else if (NBUF >= 3 && later >= 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(P) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // lgkmcnt: the K-padding zero fill is a ds_write
            unsigned long long w1 = 0;
            if (DBG && OTT_STAGE_STAMPS) w1 = __builtin_amdgcn_s_memtime();
            asm volatile("s_barrier" ::: "memory");
            if (DBG && OTT_STAGE_STAMPS) {
                dbg_wait_dma += w1 - w0;
                dbg_wait_bar += __builtin_amdgcn_s_memtime() - w1;
            }
            if constexpr (RFPRE) {
                if (rf_now) rin = fetch_rf(nxt);
            }
            const float* sA = smem + cbuf * STAGE_F;
            const float* sB = sA + A_FLOATS;
            if constexpr (MICRO) {
                // 16x16x4: lane (l15, l4) holds A[row l15][k = 4*l4 + i] in element i of one b128 (the k order inside a
                // 16-k group is permuted the same way on both operands)
#pragma unroll
                for (int h = 0; h < MKC / 16; h++) {
                    float4 a[MB], b;
#pragma unroll
                    for (int mb = 0; mb < MB; mb++) a[mb] = *reinterpret_cast<const float4*>(sA + swz(wm * WM + mb * 16 + l15, 4 * h + l4));
                    b = *reinterpret_cast<const float4*>(sB + swz(l15, 4 * h + l4));
                    if (more && h == 0) {
#pragma unroll
                        for (int m = 0; m < P; m++) dma_piece(TT, ns, nbuf, m, Qp);
                    }
#pragma unroll
                    for (int mb = 0; mb < MB; mb++) {
                        acc[mb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb].x, b.x, acc[mb][0], 0, 0, 0);
                        acc[mb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb].y, b.y, acc[mb][0], 0, 0, 0);
                        acc[mb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb].z, b.z, acc[mb][0], 0, 0, 0);
                        acc[mb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb].w, b.w, acc[mb][0], 0, 0, 0);
                    }
                }
            } else if constexpr (HI) {
                const char *sbA = nullptr, *sbB = nullptr;  // SADDR, 64 / 128 queries: this stage's two scalar bases and LDS block addresses
                uint32_t slA = 0, slB = 0;
#pragma unroll
                for (int jg = 0; jg < 4; jg++) {
                    // 64 bf16 k per row-stage; 32x32x16: lane (l31, lh) holds k = 16*jg + 8*lh .. +7 of its row = 16-B slot 2jg + lh
                    bf16x8 ah[MB], bh[NB];
#pragma unroll
                    for (int mb = 0; mb < MB; mb++) ah[mb] = *reinterpret_cast<const bf16x8*>(sA + swz(wm * WM + mb * 32 + l31, 2 * jg + lh));
#pragma unroll
                    for (int nb = 0; nb < NB; nb++) bh[nb] = *reinterpret_cast<const bf16x8*>(sB + swz(wn * WN + nb * 32 + l31, 2 * jg + lh));
                    // 256-query tile: a DMA piece costs its wave 60-180 issue cycles next to MFMAs, so the next stage's pieces go
                    // BEHIND a k-group's MFMAs (which then run while the pieces issue), the four row pieces (HBM latency) behind the
                    // first k-group, the four query pieces (L2) behind the second.  (All eight in front of the first k-group's
                    // MFMAs: K loop 44k cycles per tile; behind it: 43k; split 4 + 4: 38.8k.  Two behind every k-group leaves the
                    // last ones too little time to land: 42.6k.  The 64- / 128-query tiles are HBM-bound: the same move changes nothing.)
                    constexpr bool DMA_BEHIND = NB == 4;
                    if constexpr (SADDR && !DMA_BEHIND) {
                        // (64 / 128 queries: one row piece and one query piece per k-group as below, from the stage's two scalar bases)
                        if (more) {
                            if (L == 1) {
                                if (jg == 0) {
                                    dma_group(TT, ns, nbuf, 0, Qp);
                                    dma_group(TT, ns, nbuf, 1, Qp);
                                }
                            } else {
                                if (jg == 0) {
                                    sbA = uniform_ptr(TT.baseA + ns * (MKC * 4));
                                    sbB = uniform_ptr(reinterpret_cast<const char*>(Qp) + ns * (MKC * 4));
                                    slA = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds_base + (uint32_t)(nbuf * STAGE_F + wave * 32 * MKC) * 4u));
                                    slB = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds_base + (uint32_t)(nbuf * STAGE_F + A_FLOATS + wave * (8 * NB) * MKC) * 4u));
                                }
                                glds16u<ROWS_NT_U>(sbA, TT.offA[jg], slA + (uint32_t)(8 * jg * MKC) * 4u);
                                if constexpr (NARROW) {  // (the 32-query tile's one query piece: half a piece per wave, its own offsets)
                                    if (jg == 0) dma_piece(TT, ns, nbuf, 4, Qp);
                                } else {
                                    if (jg < NB) glds16u(sbB, offBq[jg < NB ? jg : 0], slB + (uint32_t)(8 * jg * MKC) * 4u);
                                }
                            }
                        }
                    } else if (more && !DMA_BEHIND) {
                        if (NBUF == 2 || L == 1) {
                            if (jg == 0) {
#pragma unroll
                                for (int m = 0; m < P; m++) dma_piece(TT, ns, nbuf, m, Qp);
                            }
                        } else {
                            dma_piece(TT, ns, nbuf, jg, Qp);
                            if (jg < NB) dma_piece(TT, ns, nbuf, 4 + jg, Qp);
                        }
                    }
#pragma unroll
                    for (int mb = 0; mb < MB; mb++)
#pragma unroll
                        for (int nb = 0; nb < NB; nb++) {
                            if constexpr (I8)
                                acc[mb][nb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, ah[mb]), __builtin_bit_cast(i32x4, bh[nb]), acc[mb][nb], 0, 0, 0);
                            else if constexpr (BF3 == 4)
                                acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[mb]), __builtin_bit_cast(f16x8, bh[nb]), acc[mb][nb], 0, 0, 0);
                            else
                                acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mb], bh[nb], acc[mb][nb], 0, 0, 0);
                        }
                    if (DMA_BEHIND && more && jg < 2) {
                        __builtin_amdgcn_sched_barrier(0);
                        dma_group(TT, ns, nbuf, jg, Qp);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else if constexpr (BF3) {
#pragma unroll
                for (int jg = 0; jg < MKC / 16; jg++) {
                    // 32x32x16: lane (l31, lh) holds k = 16*jg + 8*lh .. +7 of its row: 8 f32 = logical 16-B slots 4jg+2lh, +1
                    bf16x8 ah[MB], al[MB], bh[NB], bl[NB];
#pragma unroll
                    for (int mb = 0; mb < MB; mb++) {
                        const int arow = wm * WM + mb * 32 + l31;
                        if constexpr (BF3 == 2) {
                            ah[mb] = *reinterpret_cast<const bf16x8*>(sA + swz(arow, 2 * jg + lh));
                            al[mb] = *reinterpret_cast<const bf16x8*>(sA + swz(arow, 4 + 2 * jg + lh));
                            continue;
                        }
                        const float4 x0 = *reinterpret_cast<const float4*>(sA + swz(arow, 4 * jg + 2 * lh));
                        const float4 x1 = *reinterpret_cast<const float4*>(sA + swz(arow, 4 * jg + 2 * lh + 1));
                        const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
                        for (int e = 0; e < 8; e++) {
                            const __bf16 h = (__bf16)x[e];                 // v_cvt_pk_bf16_f32: round to nearest even
                            ah[mb][e] = h;
                            al[mb][e] = (__bf16)(x[e] - (float)h);          // exact difference, rounded once
                        }
                    }
                    // queries: hi halves in logical slots 0..3 of the row-stage, lo halves in slots 4..7
#pragma unroll
                    for (int nb = 0; nb < NB; nb++) {
                        const int brow = wn * WN + nb * 32 + l31;
                        bh[nb] = *reinterpret_cast<const bf16x8*>(sB + swz(brow, 2 * jg + lh));
                        bl[nb] = *reinterpret_cast<const bf16x8*>(sB + swz(brow, 4 + 2 * jg + lh));
                    }
                    // 256-query tile: the next stage's pieces go behind MFMAs, as in the hi pass — rows behind the first row block
                    // of the first k-group, queries behind its second
                    constexpr bool SPLIT_BEHIND = NB == 4;
                    if (more && !SPLIT_BEHIND) {
                        if (NBUF == 2 || L == 1) {
                            if (jg == 0) {
#pragma unroll
                                for (int m = 0; m < P; m++) dma_piece(TT, ns, nbuf, m, Qp);
                            }
                        } else {
#pragma unroll
                            for (int m = 0; m < 2; m++) {
                                dma_piece(TT, ns, nbuf, 2 * jg + m, Qp);
                                if (2 * jg + m < NB) dma_piece(TT, ns, nbuf, 4 + 2 * jg + m, Qp);
                            }
                        }
                    }
#pragma unroll
                    for (int mb = 0; mb < MB; mb++) {
#pragma unroll
                        for (int nb = 0; nb < NB; nb++) {
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mb], bh[nb], acc[mb][nb], 0, 0, 0);
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mb], bl[nb], acc[mb][nb], 0, 0, 0);
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mb], bh[nb], acc[mb][nb], 0, 0, 0);
                        }
                        if (SPLIT_BEHIND && more && jg == 0) {
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int m = 0; m < 4; m++) dma_piece(TT, ns, nbuf, 4 * mb + m, Qp);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int o = 0; o < MKC / 8; o++) {
                    float4 a[MB], b[NB];
    #pragma unroll
                    for (int mb = 0; mb < MB; mb++) a[mb] = *reinterpret_cast<const float4*>(sA + swz(wm * WM + mb * 32 + l31, 2 * o + lh));
    #pragma unroll
                    for (int nb = 0; nb < NB; nb++) b[nb] = *reinterpret_cast<const float4*>(sB + swz(wn * WN + nb * 32 + l31, 2 * o + lh));
                    // the ring slot being refilled was last read one stage ago at the latest, which every wave left before this
                    // stage's barrier.  Lookahead 1: the pieces must land before the NEXT barrier, so all of them go out in the
                    // first octet (spread over the octets, the last ones had < 2048 cycles to land and the barrier wait showed it);
                    // lookahead 2: they have a whole extra stage, one A piece (+ one query piece) per octet keeps issue smooth
                    constexpr bool F32_BEHIND = NB == 4;  // 256-query tile: pieces behind the first octet's MFMAs (rows), the second's (queries)
                    if (more && !F32_BEHIND) {
                        if (NBUF == 2 || L == 1) {
                            if (o == 0) {
    #pragma unroll
                                for (int m = 0; m < P; m++) dma_piece(TT, ns, nbuf, m, Qp);
                            }
                        } else {
                            dma_piece(TT, ns, nbuf, o, Qp);
                            if (o < NB) dma_piece(TT, ns, nbuf, 4 + o, Qp);
                        }
                    }
    #pragma unroll
                    for (int mb = 0; mb < MB; mb++)
    #pragma unroll
                        for (int nb = 0; nb < NB; nb++) {
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mb].x, b[nb].x, acc[mb][nb], 0, 0, 0);
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mb].y, b[nb].y, acc[mb][nb], 0, 0, 0);
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mb].z, b[nb].z, acc[mb][nb], 0, 0, 0);
                            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mb].w, b[nb].w, acc[mb][nb], 0, 0, 0);
                        }
                    if (F32_BEHIND && more && o < 2) {
                        __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                        for (int m = 0; m < 4; m++) dma_piece(TT, ns, nbuf, 4 * o + m, Qp);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            cbuf = cbuf + 1 == NBUF ? 0 : cbuf + 1;
            nbuf = nbuf + 1 == NBUF ? 0 : nbuf + 1;
        };
        uint32_t s = 0;
        for (; s + L < nstages; s++) stage(cur, s + L, true, L - 1, Qcur, false);        // issues this unit's later stages
        for (; s < nstages; s++) {                                                      // last L stages: the next unit's first ones
            const uint32_t left = nstages - 1 - s;  // stages of this unit still behind this one
            stage(nxtA, s + L - nstages, has_next, has_next ? L - 1 : (left < L - 1 ? left : L - 1), Qnx,
                  RFPRE && s + 1 == nstages && last_blk && has_next_tile);
        }

        if (DBG) t2 = __builtin_amdgcn_s_memtime();
        // (one ds_read_b128 at a wave-uniform address: the four flag words the prologue's waves 0..3 wrote for this tile)
        const uint4 flagw = *reinterpret_cast<const uint4*>(sFlagW);
        const bool tile_noflag = __builtin_amdgcn_readfirstlane((int)(flagw.x | flagw.y | flagw.z | flagw.w)) == 0;
        const uint32_t epi_dense_base = (t - p.tile_begin) * BM;
#include "ott_mfma_epilogue.inc"
        unsigned long long t3 = 0;
        if (DBG) t3 = __builtin_amdgcn_s_memtime();
        // a wave that appended candidates drains its stores / atomics here: left outstanding they would be counted as
        // DMA pieces by the next tile's counted wait
        if (epi_vm) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (DBG) {
            const unsigned long long t4 = __builtin_amdgcn_s_memtime();
            if (tid == 0) {
                p.dbg[blockIdx.x * 4 + 0] += t1 - t0;
                p.dbg[blockIdx.x * 4 + 1] += t2 - t1;
                p.dbg[blockIdx.x * 4 + 2] += t4 - t2;  // (the epilogue's parts: slots 7 .. 10)
                p.dbg[blockIdx.x * 4 + 3] += 1;
                p.dbg[(size_t)p.dbg_wgs * 4 + blockIdx.x] += __builtin_amdgcn_s_memrealtime() - r0;
                p.dbg[(size_t)p.dbg_wgs * 5 + blockIdx.x] += dbg_wait_dma;
                p.dbg[(size_t)p.dbg_wgs * 6 + blockIdx.x] += dbg_wait_bar;
                p.dbg[(size_t)p.dbg_wgs * 7 + blockIdx.x] += epi_t_setup - t2;
                p.dbg[(size_t)p.dbg_wgs * 8 + blockIdx.x] += epi_t_walk - epi_t_setup;
                p.dbg[(size_t)p.dbg_wgs * 9 + blockIdx.x] += t3 - epi_t_walk;
                p.dbg[(size_t)p.dbg_wgs * 10 + blockIdx.x] += t4 - t3;
                p.dbg[(size_t)p.dbg_wgs * 11 + blockIdx.x] += epi_vm ? 1 : 0;
            }
        }
        if (!has_next) break;
        if (last_blk) {
            qblk = 0;
            t = tn;
            cur = nxt;
        } else {
            qblk++;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// select_kernel: one workgroup per query.  Finds the k-th best approximate score among the
// query's candidates (MSB-first radix select on the order-preserving key), raises tau[q] to it
// and copies the entries at least that good from the `in` list to the `out` list (ping-pong;
// the next scoring round appends to `out`).  Sets overflow[q] if the list overflowed its
// capacity (then nothing can be certified for q).
// ---------------------------------------------------------------------------------------------
//
// The GATE (gate[q]) is the threshold the next scoring round emits against.  Conservatively it is tau, the k-th best so far
// (k = T, the number of candidates re-scored per query): every row that can still belong to the T best overall is listed.
// With j_gate in 1 .. k-1 it is SPECULATIVE — the j_gate-th best so far, j_gate chosen by the host so that about 8 T rows
// of the whole store are expected above it (rows in no particular order: what has been seen is a sample) — and never
// lowered.  A tighter gate lists fewer pairs in the rounds that follow (the early rounds spent most of their time appending
// survivors).  It costs nothing in exactness: finalize_kernel takes the gate into the bound on what a row NOT listed can
// score, so a gate that turned out too tight (fewer than ~k rows above it in the end) leaves the query uncertified and the
// cascade answers it at the next level, conservatively.
constexpr int SEL_THREADS = 1024;  // one workgroup per query; the list scans (5 passes over <= 16K entries) are what it costs
constexpr uint32_t SEL_KEEP = 1024;  // kept entries whose keys are held in LDS for the gate's rank search
__global__ __launch_bounds__(SEL_THREADS) void select_kernel(const CandEntry* cand_in, const uint32_t* cnt_in, CandEntry* cand_out,
                                                      uint32_t* cnt_out, float* tau, uint32_t* overflow, uint32_t cap, uint32_t k,
                                                      uint32_t take_max, float* gate, uint32_t j_gate) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t s_prefix, s_remaining, s_out, s_ties, s_gate;
    __shared__ uint32_t s_keys[SEL_KEEP];
    const uint32_t q = blockIdx.x;
    const int tid = threadIdx.x;
    uint32_t n = cnt_in[(size_t)q * CNT_STRIDE];
    if (n > cap) {
        if (tid == 0) overflow[q] = 1;
        n = cap;
    }
    const CandEntry* c = cand_in + (size_t)q * cap;
    CandEntry* o = cand_out + (size_t)q * cap;
    // the list is read ONCE: a thread keeps its first SEL_PER entries (8 x 1024 = the dense first round's 8192 pairs) in
    // registers for the four radix passes and the compaction; only entries beyond that are read again from memory (the
    // passes were five dependent trips to L2: 38-42 us for the first select of a batch, 10-17 us for the later ones)
    constexpr int SEL_PER = 8;
    CandEntry mine[SEL_PER];
    uint32_t mkey[SEL_PER];
#pragma unroll
    for (int j = 0; j < SEL_PER; j++) {
        const uint32_t i = (uint32_t)tid + (uint32_t)j * SEL_THREADS;
        mine[j] = c[i < n ? i : 0];
        mkey[j] = mine[j].row == 0xFFFFFFFFu ? 0u : cand_ord(mine[j].score, take_max != 0);  // absent pair of the dense round: 0
    }
    uint32_t kth = 0;  // keep everything unless there are at least k candidates
    uint32_t need_ties = 0xFFFFFFFFu;  // how many entries AT the k-th value belong to the k best
    if (n >= k && k > 0) {
        uint32_t prefix = 0, mask = 0, remaining = k;
        for (int shift = 24; shift >= 0; shift -= 8) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < SEL_PER; j++) {
                const uint32_t i = (uint32_t)tid + (uint32_t)j * SEL_THREADS;
                if (i < n && (mkey[j] & mask) == prefix) atomicAdd(&hist[(mkey[j] >> shift) & 255], 1u);
            }
            for (uint32_t i = (uint32_t)tid + SEL_PER * SEL_THREADS; i < n; i += SEL_THREADS) {
                const uint32_t key = c[i].row == 0xFFFFFFFFu ? 0u : cand_ord(c[i].score, take_max != 0);
                if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255], 1u);
            }
            __syncthreads();
            if (tid < 64) {
                // which bin holds the `remaining`-th largest key: lane l owns bins 4l .. 4l+3; a wave suffix scan gives the
                // number of keys in higher lanes (a serial walk over the 256 LDS counters was 7 us per pass)
                const uint32_t h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
                const uint32_t loc = h0 + h1 + h2 + h3;
                uint32_t incl = loc;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t t = __shfl_down(incl, off);
                    if (tid + off < 64) incl += t;
                }
                const uint32_t above = incl - loc;
                if (above < remaining && remaining <= above + loc) {
                    uint32_t rem = remaining - above;
                    int b = 3;
                    if (h3 >= rem) b = 3;
                    else if (h3 + h2 >= rem) { rem -= h3; b = 2; }
                    else if (h3 + h2 + h1 >= rem) { rem -= h3 + h2; b = 1; }
                    else { rem -= h3 + h2 + h1; b = 0; }
                    s_prefix = prefix | ((uint32_t)(4 * tid + b) << shift);
                    s_remaining = rem;
                }
            }
            __syncthreads();
            prefix = s_prefix;
            remaining = s_remaining;
            mask |= 255u << shift;
            __syncthreads();
        }
        kth = prefix;
        // Entries tied with the k-th value beyond the k best are dropped: they become "outside" rows whose approximate score
        // is exactly the new tau, which the certification's bound (tau +/- eps) covers like any other outside row.  (Kept,
        // a plateau of equal scores made the list longer than what finalize_kernel sorts.)  A k-th value of "forced" is
        // left alone: finalize must see that more than T forced candidates exist.
        if (kth != 0xFFFFFFFFu) need_ties = remaining;
    }
    if (tid == 0) {
        s_out = 0;
        s_ties = 0;
    }
    __syncthreads();
    auto keep = [&](const CandEntry& e, uint32_t eo) {
        if (e.row == 0xFFFFFFFFu) return;
        if (eo > kth || (eo == kth && (need_ties == 0xFFFFFFFFu || atomicAdd(&s_ties, 1u) < need_ties))) {
            const uint32_t at = atomicAdd(&s_out, 1u);
            o[at] = e;
            if (at < SEL_KEEP) s_keys[at] = eo;
        }
    };
#pragma unroll
    for (int j = 0; j < SEL_PER; j++)
        if ((uint32_t)tid + (uint32_t)j * SEL_THREADS < n) keep(mine[j], mkey[j]);
    for (uint32_t i = (uint32_t)tid + SEL_PER * SEL_THREADS; i < n; i += SEL_THREADS) {
        const CandEntry e = c[i];
        keep(e, cand_ord(e.score, take_max != 0));
    }
    if (tid == 0) s_gate = 0;
    __syncthreads();
    const uint32_t kept = s_out;
    // speculative gate: the j_gate-th best of the kept entries (rank by counting: <= 1024 keys in LDS, one per thread)
    const bool spec = j_gate > 0 && j_gate < k && kept >= j_gate && kept <= SEL_KEEP;
    if (spec && (uint32_t)tid < kept) {
        const uint32_t mine = s_keys[tid];
        uint32_t rank = 0;
        for (uint32_t l = 0; l < kept; l++) {
            const uint32_t other = s_keys[l];
            rank += (other > mine || (other == mine && l < (uint32_t)tid)) ? 1u : 0u;
        }
        if (rank == j_gate - 1) s_gate = mine;
    }
    __syncthreads();
    if (tid == 0) {
        cnt_out[(size_t)q * CNT_STRIDE] = kept;
        const bool tmax = take_max != 0;
        float g = gate[q];
        if (kth != 0 && kth != 0xFFFFFFFFu) {  // (all-forced lists leave tau alone)
            const float t = score_of(kth, tmax);
            tau[q] = t;
            if (g == g) g = tmax ? fmaxf(g, t) : fminf(g, t);  // (a NaN gate = a query kept out of the approximate pass: stays)
        }
        const uint32_t sg = s_gate;
        if (spec && sg != 0 && sg != 0xFFFFFFFFu && g == g) {  // (a forced entry at rank j: no speculation this round)
            const float t = score_of(sg, tmax);
            g = tmax ? fmaxf(g, t) : fminf(g, t);
        }
        gate[q] = g;
    }
}

// ---------------------------------------------------------------------------------------------
// finalize_kernel: one wave per query.  (1) top-T candidates by approximate score, (2) exact
// re-score of those T rows in the reference's order of operations, (3) exact filter, exact
// canonical top-k, (4) certification.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool f_before(uint64_t ak, uint64_t bk) { return ak > bk; }

template <int E>
struct FList {  // sorted by key descending; keys unique (ord<<32 | ~row)
    uint64_t key[E];
};

template <int E>
__device__ __forceinline__ void fl_insert(FList<E>& L, uint64_t xk, int lane) {
    int pos = 0;
#pragma unroll
    for (int e = 0; e < E; e++) pos += __popcll(__ballot(L.key[e] > xk));
#pragma unroll
    for (int e = E - 1; e >= 0; e--) {
        uint64_t up = __shfl_up(L.key[e], 1);
        if (e > 0) {
            const uint64_t pk = __shfl(L.key[e - 1], 63);
            if (lane == 0) up = pk;
        }
        const int ppos = e * 64 + lane;
        if (ppos == pos) L.key[e] = xk;
        else if (ppos > pos) L.key[e] = up;
    }
}

template <int E>
__device__ __forceinline__ uint64_t fl_at(const FList<E>& L, uint32_t pos) {
    uint64_t v = 0;
#pragma unroll
    for (int e = 0; e < E; e++)
        if ((int)(pos >> 6) == e) v = __shfl(L.key[e], pos & 63);
    return v;
}

template <int E>
__device__ __forceinline__ void fl_offer(FList<E>& L, uint64_t& tk, uint32_t T, bool pass, uint64_t key, int lane) {
    pass = pass && key > tk;
    uint64_t m = __ballot(pass);
    while (m) {
        const int src = __builtin_ctzll(m);
        m &= m - 1;
        const uint64_t xk = __shfl(key, src);
        if (xk > tk) {
            fl_insert(L, xk, lane);
            tk = fl_at(L, T - 1);
        }
    }
}

// The int8 passes' accumulation constant for cosine / dot, in units of 2^-24 relative to ||q|| ||v||: a bound on
// |approximate - EXACT-ORDER f32 score| beyond the measured quantisation losses.  Two sides:
//  * the approximate score: the integer accumulation is exact; one i32 -> f32 conversion, the row factor's two multiplies and the
//    score's one, the per-element rounding of the pre-scaled cosine operand: 16 units cover them;
//  * the exact-order re-score it is compared with (src/vec_compute.rs:9-22) is NOT the real dot product: a lane chain is dim/8
//    rounded products and rounded adds, then three adds of reduce_add, the remainder's chain (at most 7 + 1 adds) and cosine's two
//    multiplies — |fl - real| <= gamma(dim/8 + 6) * sum|q_i v_i| <= (dim/8 + 16) units of ||q|| ||v|| (Higham, recursive summation;
//    the 1 / (1 - n u) factor is inside the slack for every dim the store accepts).
// Round 5 priced the first side only ("pure quantisation"): with int8-REPRESENTABLE rows and queries the measured losses are ~1e-7,
// the bound collapsed to ~17 units, and nearly constant rows — whose lane sums drift by 30-150 units, every add rounding the same
// way — were left outside a "certified" list (tests/adversarial_i8.py, tests/test_gpu_i8_bound.py, profiles/round6/i8_bound.md).
static inline float i8_c_eps_units(uint32_t dim) { return 0.125f * (float)dim + 32.0f; }

struct FinalParams {
    const float* rows;
    const float* inv;
    const float* Q;     // [nq_pad][ldq]
    const float* qinv;
    const float* tau;
    const float* gate;  // [nq_pad] the emission threshold of the last scoring round (== tau unless it was speculative, select_kernel)
    const uint32_t* cnt;
    const CandEntry* cand;
    const uint32_t* overflow;
    ott_hit* out;       // [nq][out_stride]
    uint64_t* out_cnt;  // [nq]
    uint32_t* uncertified;  // [nq]
    uint64_t base_offset;
    uint32_t cap, ld, dim, ldq, nq;
    uint32_t k, T, out_stride;
    uint32_t metric, take_max, cmp, reduce;
    float thr;
    float eps_c;         // (1.25*dim + 32) * 2^-24
    float max_norm;      // upper bound on ||v|| over the store
    const float* qnorm;  // [nq_pad] upper bound on ||q||
    const float* qrel;   // hi pass: [nq_pad] measured ||q - bf16(q)|| / ||q|| of the operand rows (added to eps_c); else NULL
    float qrel_cap;      // hi pass: the largest qrel the host's relaxed filter assumed; a query above it is not certified
    float eps_r;         // hi pass: (1 + 2^-8) x the measured rounding loss of the store's rows; 0 otherwise
    float eps_scale;     // 1, or what the test-only option eps_scale_ppm shrinks every term of the bound by
    uint32_t* err_ratio; // [nq] float bits: max over the re-scored candidates of |approximate - exact| / eps — how much of the
                         // certification's error bound the pass actually used (diagnostic; ott_stats.err_ratio_max)
};

__device__ __forceinline__ bool f_cmp(float s, uint32_t cmp, float thr) {
    switch (cmp) {
        case OTT_CMP_LT: return s < thr;
        case OTT_CMP_GT: return s > thr;
        case OTT_CMP_LTE: return s <= thr;
        case OTT_CMP_GTE: return s >= thr;
        case OTT_CMP_EQ: return s == thr;
        default: return true;
    }
}

#ifndef OTT_X_FIN_WAVES
#define OTT_X_FIN_WAVES 8
#endif
constexpr int FIN_WAVES = OTT_X_FIN_WAVES;  // all waves sort and re-score (the re-score is memory-latency bound); wave 0 certifies

// descending bitonic sort of s[0, N) (N a power of two) by the whole workgroup; ends with a barrier
__device__ __forceinline__ void block_sort_desc(uint64_t* s, uint32_t N, uint32_t tid, uint32_t nthreads) {
    for (uint32_t k2 = 2; k2 <= N; k2 <<= 1) {
        for (uint32_t j = k2 >> 1; j > 0; j >>= 1) {
            for (uint32_t idx = tid; idx < N / 2; idx += nthreads) {
                const uint32_t i = ((idx & ~(j - 1)) << 1) | (idx & (j - 1)), l = i | j;
                const uint64_t a = s[i], b = s[l];
                if ((a < b) == ((i & k2) == 0)) {
                    s[i] = b;
                    s[l] = a;
                }
            }
            __syncthreads();
        }
    }
}

// E sizes the register-resident exact top-k list (k <= 64 E); TCAP >= T the LDS arrays (64 E normally, 4096 for the cascade's
// wide level)
template <int E, int TCAP>
__global__ __launch_bounds__(64 * FIN_WAVES) void finalize_kernel(FinalParams p) {
    constexpr uint32_t SORTCAP = TCAP < 1024 ? 1024 : TCAP;
    __shared__ uint32_t sRows[TCAP];     // the T candidates' rows, best approximate score first
    constexpr bool MARGIN = TCAP <= 512;  // (the 4096-candidate level has no LDS left for a copy of the approximate scores: it reads them in place, see the re-score loop)
    __shared__ uint32_t sAppr[MARGIN ? TCAP : 1];  // their approximate ordinals (cand_ord), for the error-bound check
    __shared__ uint32_t sErr;
    if (threadIdx.x == 0) sErr = 0u;  // (ordered before its first use by the barriers of the candidate sort below)
    __shared__ uint64_t sKeys[SORTCAP];  // their exact keys (0 = failed the exact filter); before that, the approximate-key sort
    __shared__ uint32_t sNT;
    uint64_t* sSort = sKeys;  // the query's whole list, when it fits (it does after the last select: ~T entries); dead before sKeys is written
    __shared__ uint64_t sKeyT;
    const uint32_t q = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool tmax = p.take_max != 0;
    const uint32_t cnt_q = p.cnt[(size_t)q * CNT_STRIDE];
    const uint32_t n = cnt_q < p.cap ? cnt_q : p.cap;
    const CandEntry* c = p.cand + (size_t)q * p.cap;

    // (1) top-T by approximate score (ties by lower row).  Lists of <= 1024 entries (the rule: the last select left ~T) are
    // sorted in LDS by the whole workgroup; longer ones (heavy ties) go through wave 0's insertion list.  (Inserting ~T
    // entries one by one into a register-resident list was ~100 us of a 270 us launch at T = 256.)
    const bool small = n <= SORTCAP;
    if (!small && p.T > 64u * E) {  // wide level, list longer than the sort buffer (a plateau of equal scores): not certified
        if (threadIdx.x == 0) {
            p.out_cnt[q] = 0;
            p.uncertified[q] = 1u;
            if (p.err_ratio != nullptr) p.err_ratio[q] = 0u;
        }
        return;
    }
    FList<E> A;
#pragma unroll
    for (int e = 0; e < E; e++) A.key[e] = 0;
    uint32_t nT = 0;
    float outside = 0.0f;  // best possible approximate score of a row NOT re-scored
    if (small) {
        uint32_t N = 64;
        while (N < n) N <<= 1;
        for (uint32_t i = threadIdx.x; i < N; i += 64 * FIN_WAVES) {
            uint64_t key = 0;
            if (i < n) {
                const CandEntry e = c[i];
                if (e.row != 0xFFFFFFFFu) key = ((uint64_t)cand_ord(e.score, tmax) << 32) | (uint32_t)~e.row;  // absent pairs of a dense round: 0
            }
            sSort[i] = key;
        }
        __syncthreads();
        block_sort_desc(sSort, N, threadIdx.x, 64 * FIN_WAVES);
        if (threadIdx.x == 0) {
            sNT = 0;
            sKeyT = p.T - 1 < N ? sSort[p.T - 1] : 0ull;
        }
        __syncthreads();
        // (rows out first, through registers: sSort aliases sKeys, which the re-score below overwrites)
        for (uint32_t i0 = 0; i0 < (uint32_t)TCAP; i0 += 64 * FIN_WAVES) {
            const uint32_t i = i0 + threadIdx.x;
            const uint64_t key = (i < N && i < p.T) ? sSort[i] : 0ull;
            const uint64_t nextk = (i + 1 < N && i + 1 < p.T) ? sSort[i + 1] : 0ull;
            if (i < (uint32_t)TCAP) sRows[i] = ~(uint32_t)(key & 0xFFFFFFFFull);
            if (MARGIN && i < (uint32_t)TCAP) sAppr[i] = (uint32_t)(key >> 32);
            if (key != 0 && nextk == 0) sNT = i + 1;  // sorted: non-empty keys first, exactly one boundary
        }
        __syncthreads();
        nT = sNT;
        if (n > p.T && nT == p.T) {
            const uint32_t oT = (uint32_t)(sKeyT >> 32);
            outside = oT == 0xFFFFFFFFu ? __uint_as_float(0x7FC00000u) : score_of(oT, tmax);  // T forced entries: cannot certify
        } else outside = p.tau[q];  // every listed pair is re-scored: the rest failed the emission threshold
    } else {
        uint64_t tk = 0;
        for (uint32_t i0 = 0; wave == 0 && i0 < n; i0 += 64) {
            const uint32_t i = i0 + lane;
            bool pass = i < n;
            uint64_t key = 0;
            if (pass) {
                const CandEntry e = c[i];
                key = ((uint64_t)cand_ord(e.score, tmax) << 32) | (uint32_t)~e.row;
                pass = e.row != 0xFFFFFFFFu;
            }
            fl_offer(A, tk, p.T, pass, key, lane);
        }
        // candidates that will be re-scored: the list may hold absent pairs (dense first round), so count the real ones
        if (wave == 0) {
#pragma unroll
            for (int e = 0; e < E; e++) nT += __popcll(__ballot((uint32_t)(e * 64 + lane) < p.T && A.key[e] != 0));
            if (n > p.T && nT == p.T) {
                const uint32_t oT = (uint32_t)(fl_at(A, p.T - 1) >> 32);
                outside = oT == 0xFFFFFFFFu ? __uint_as_float(0x7FC00000u) : score_of(oT, tmax);  // T forced entries: cannot certify
            } else outside = p.tau[q];  // every listed pair is re-scored: the rest failed the emission threshold
#pragma unroll
            for (int e = 0; e < E; e++) {
                sRows[e * 64 + lane] = ~(uint32_t)(A.key[e] & 0xFFFFFFFFull);
                if (MARGIN) sAppr[e * 64 + lane] = (uint32_t)(A.key[e] >> 32);
            }
            if (lane == 0) sNT = nT;
        }
        __syncthreads();
        nT = sNT;
    }

    // (2)+(3) exact re-score, 8 lanes per pair (lane&7 = accumulator chain), 8 pairs per step
    // bound on |approx - exact| for this query.  eps_c: accumulation (and split) terms, relative to
    // ||q|| ||v|| — for squared L2 priced at (||q|| + ||v||)^2, which also covers the rounding of its norm terms.  r: the hi
    // pass's operand rounding loss, a bound on the DOT's error relative to ||q|| ||v||, so squared L2 (= norms - 2 dot)
    // takes it twice.
    float eps;
    const float qrel = p.qrel ? p.qrel[q] : 0.0f;
    {
        const float r = p.eps_r + 1.001f * qrel * p.eps_scale;
        if (p.metric == OTT_METRIC_COSINE) eps = p.eps_c + r;
        else if (p.metric == OTT_METRIC_DOT) eps = (p.eps_c + r) * p.qnorm[q] * p.max_norm;
        else eps = p.eps_c * (p.qnorm[q] + p.max_norm) * (p.qnorm[q] + p.max_norm) + 2.0f * r * p.qnorm[q] * p.max_norm;
    }
    float err_max = 0.0f;  // this lane's largest |approx - exact| / eps
    const float* __restrict__ qv = p.Q + (size_t)q * p.ldq;
    const float q_inv = p.qinv[q];
    const int chain = lane & 7, pr = lane >> 3;
    const uint32_t full = p.dim / 8;
    for (uint32_t j0 = 8 * wave; j0 < nT; j0 += 8 * FIN_WAVES) {
        const bool have = (j0 + pr) < nT;
        const uint32_t row = sRows[have ? j0 + pr : 0];
        const float* __restrict__ v = p.rows + (uint64_t)row * p.ld;
        float accum = 0.0f;
        // the chain's adds are sequential by contract, its loads are not: fetch 16 steps' operands together (one memory
        // latency per 16 steps instead of one per step: this loop was 0.34 ms per 128 candidates of dim 768)
        const bool l2 = p.metric == OTT_METRIC_EUCLIDEAN;
        for (uint32_t s0 = 0; s0 < full; s0 += 16) {
            float vv[16], qq[16];
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const uint32_t idx = 8 * ((s0 + u) < full ? (s0 + u) : s0) + chain;
                vv[u] = v[idx];
                qq[u] = qv[idx];
            }
#pragma unroll
            for (int u = 0; u < 16; u++) {
                if (s0 + u < full) {
                    if (l2) {
                        const float df = __fsub_rn(qq[u], vv[u]);
                        accum = __fadd_rn(accum, __fmul_rn(df, df));
                    } else {
                        accum = __fadd_rn(accum, __fmul_rn(qq[u], vv[u]));
                    }
                }
            }
        }
        // wide::f32x8::reduce_add over the 8 chains held by 8 consecutive lanes
        float red;
        {
            const float l0 = __shfl(accum, (lane & ~7) + 0), l1 = __shfl(accum, (lane & ~7) + 1);
            const float l2 = __shfl(accum, (lane & ~7) + 2), l3 = __shfl(accum, (lane & ~7) + 3);
            const float l4 = __shfl(accum, (lane & ~7) + 4), l5 = __shfl(accum, (lane & ~7) + 5);
            const float l6 = __shfl(accum, (lane & ~7) + 6), l7 = __shfl(accum, (lane & ~7) + 7);
            if (p.reduce == OTT_REDUCE_SEQ4)
                red = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(l0, l1), l2), l3), __fadd_rn(__fadd_rn(__fadd_rn(l4, l5), l6), l7));
            else
                red = __fadd_rn(__fadd_rn(__fadd_rn(l0, l4), __fadd_rn(l2, l6)), __fadd_rn(__fadd_rn(l1, l5), __fadd_rn(l3, l7)));
        }
        float tail = 0.0f;
        if (p.metric == OTT_METRIC_EUCLIDEAN) {
            for (uint32_t i = full * 8; i < p.dim; i++) {
                const float df = __fsub_rn(qv[i], v[i]);
                tail = __fadd_rn(tail, __fmul_rn(df, df));
            }
        } else {
            for (uint32_t i = full * 8; i < p.dim; i++) tail = __fadd_rn(tail, __fmul_rn(qv[i], v[i]));
        }
        float sc = __fadd_rn(red, tail);
        if (p.metric == OTT_METRIC_COSINE) sc = __fmul_rn(__fmul_rn(sc, q_inv), p.inv[row]);
        const bool pass = !(sc != sc) && f_cmp(sc, p.cmp, p.thr);
        // the candidate's approximate ordinal: sAppr, or — wide level, whose 4096 slots leave no LDS for a copy — the sorted
        // approximate key still sitting in the very slot its exact key is about to replace (sSort aliases sKeys, candidate j
        // lives at index j in both, and only this lane touches slot j)
        const uint32_t ao_slot = (have && chain == 0) ? (MARGIN ? sAppr[j0 + pr] : (uint32_t)(sKeys[j0 + pr] >> 32)) : 0u;
        if (have && chain == 0) sKeys[j0 + pr] = pass ? (((uint64_t)ord_of(sc, tmax) << 32) | (uint32_t)~row) : 0ull;
        if (have && chain == 0 && p.err_ratio != nullptr) {
            const uint32_t ao = ao_slot;
            if (ao != 0xFFFFFFFFu && ao != 0u && eps > 0.0f) {  // (a forced candidate has no approximate score)
                const float e = fabsf(score_of(ao, tmax) - sc) / eps;
                if (e == e && e > err_max) err_max = e;
            }
        }
    }
    if (p.err_ratio != nullptr && __ballot(err_max > 0.0f) != 0) {
        uint32_t eb = __float_as_uint(err_max);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t o2 = (uint32_t)__shfl_xor((int)eb, off);
            eb = o2 > eb ? o2 : eb;
        }
        if (lane == 0) atomicMax(&sErr, eb);  // (LDS; non-negative floats order like their bit patterns)
    }
    // exact canonical top-k of the re-scored rows: sort their exact keys (failed filter = 0 sorts last), keep the first k
    uint32_t N2 = 64;
    while (N2 < nT) N2 <<= 1;
    __syncthreads();
    for (uint32_t i = nT + threadIdx.x; i < N2; i += 64 * FIN_WAVES) sKeys[i] = 0ull;
    __syncthreads();
    block_sort_desc(sKeys, N2, threadIdx.x, 64 * FIN_WAVES);
    if (wave != 0) return;
    FList<E> X;
#pragma unroll
    for (int e = 0; e < E; e++) {
        const uint32_t pos = e * 64 + lane;
        X.key[e] = (pos < p.k && pos < N2) ? sKeys[pos] : 0ull;
    }

    // (4) certification.  U = the best approximate score any row NOT re-scored can have:
    // the T-th approximate score when the list was cut, else tau (rows below tau were never
    // listed; tau still at its initial -inf/+inf means every admissible row is listed).
    // |approx - exact| <= eps, so an outside row's exact score is no better than U (+/-) eps.
    // A speculative gate (select_kernel) may sit above the T-th listed score: rows below it were never listed, whatever they
    // score, so the bound on an outside row is the better of the two.  `outside_list` = what the bound would be without the
    // gate: a query that fails only because of the gate is reported as such (uncertified = 2; the host then backs off).
    const float outside_list = outside;
    {
        const float g = p.gate[q];
        if (g == g && outside == outside) outside = tmax ? fmaxf(outside, g) : fminf(outside, g);
    }
    const bool none_outside = !(n > p.T && nT == p.T) && (tmax ? (outside == -INFINITY) : (outside == INFINITY));
    const float bound = tmax ? outside + eps : outside - eps;
    const float bound_list = tmax ? outside_list + eps : outside_list - eps;
    uint32_t cnt_exact = 0;
#pragma unroll
    for (int e = 0; e < E; e++) cnt_exact += __popcll(__ballot((uint32_t)(e * 64 + lane) < p.k && X.key[e] != 0));
    bool certified, certified_list = false;  // (certified_list: with the list's own bound, i.e. had the gate not been speculative)
    if (p.overflow[q] != 0 || !(qrel <= p.qrel_cap)) certified = false;
    else if (none_outside) certified = true;
    else if (cnt_exact == p.k) {
        // full list: exact iff its k-th score STRICTLY beats everything an outside row can reach
        const float kth = score_of((uint32_t)(fl_at(X, p.k - 1) >> 32), tmax);
        certified = tmax ? (kth > bound) : (kth < bound);
        certified_list = tmax ? (kth > bound_list) : (kth < bound_list);
    } else {
        // short list: exact iff no outside row can pass the exact score filter
        if (tmax && (p.cmp == OTT_CMP_GT || p.cmp == OTT_CMP_GTE)) {
            certified = bound < p.thr;
            certified_list = bound_list < p.thr;
        } else if (!tmax && (p.cmp == OTT_CMP_LT || p.cmp == OTT_CMP_LTE)) {
            certified = bound > p.thr;
            certified_list = bound_list > p.thr;
        } else certified = false;
    }
    ott_hit* o = p.out + (size_t)q * p.out_stride;
#pragma unroll
    for (int e = 0; e < E; e++) {
        const uint32_t ppos = e * 64 + lane;
        if (ppos < p.k && ppos < p.out_stride) {
            ott_hit h;
            h.index = ~0ull;
            h.score = __uint_as_float(0xFFFFFFFFu);
            h.query = 0xFFFFFFFFu;
            if (X.key[e] != 0) {
                h.index = p.base_offset + (uint32_t)~(uint32_t)(X.key[e] & 0xFFFFFFFFull);
                h.score = score_of((uint32_t)(X.key[e] >> 32), tmax);
                h.query = q;
            }
            o[ppos] = h;
        }
    }
    if (lane == 0) {
        p.out_cnt[q] = cnt_exact;
        p.uncertified[q] = certified ? 0u : (certified_list ? 2u : 1u);
        if (p.err_ratio != nullptr) p.err_ratio[q] = sErr;
    }
}


// ---------------------------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------------------------
static float host_sqnorm(const float* v, uint32_t dim) {
    double s = 0;
    for (uint32_t i = 0; i < dim; i++) s += (double)v[i] * v[i];
    return (float)s;
}
static float host_norm(const float* v, uint32_t dim) {
    double s = 0;
    for (uint32_t i = 0; i < dim; i++) s += (double)v[i] * v[i];
    return (float)(sqrt(s) * (1.0 + 1e-6));
}
float host_inv_norm_exact(const float* v, uint32_t dim);  // ott_api.hip (reference order)

static double host_ms() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

int run_mfma(ott_store* s, const ott_query_desc* d, const RunPlan& pl, uint64_t k_q, const uint64_t* d_mask, uint64_t mask_bits,
             std::vector<std::vector<ott_hit>>& out, std::vector<uint32_t>& uncertified, ott_stats& st, int level, uint32_t t_min, bool spec_gate) {
    const double hm0 = host_ms();
    const uint32_t nq = d->nq;
    const bool i8 = level == 2;  // int8 pass (round 5): rows and queries as scaled int8, from the store's int8 plane; cosine / dot
    const bool hi = level == 0 || i8;  // hi pass: 16-bit roundings only, from the store's hi plane (the int8 pass shares its tile geometry and its measured bound)
    // tile width: 16 or 32 queries (micro / narrow variants, 4-deep ring), 64, 128 or 256 (the micro tile is f32 only)
    const int NB = (nq <= 16 && !hi) ? -1 : nq <= 32 ? 0 : nq <= 64 ? 1 : nq <= 128 ? 2 : 4;
    const uint32_t BN = NB == -1 ? 16u : NB == 0 ? 32u : 64u * NB;
    const uint32_t nq_pad = (nq + BN - 1) / BN * BN;
    // query blocks per launch (mfma_score_kernel): the 256-wide tile takes up to 4 blocks of one row tile back to back
    const uint32_t qblk_max = NB == 4 ? std::min<uint32_t>(4u, nq_pad / BN) : 1u;
    const size_t MFMA_SMEM = (size_t)mfma_nbuf(NB) * (A_FLOATS + BN * MKC) * 4 + BM * 8 + (size_t)BN * 8 * qblk_max + (size_t)8 * mfma_qw(NB) * 8 + 16 + (i8 ? BM * 4 : 0);
    // split-bf16 candidate pass (three bf16 MFMAs per 16 k) on every 32x32 tile; OTT_MFMA_F32=1 keeps the f32 matrix pipe
    const bool bf3 = hi || (NB >= 0 && !s->opt.mfma_f32);
    uint32_t wg_per_cu = 1;  // (narrow tiles ran two workgroups of a 2-deep ring per CU until the ring went 4 deep)
    if (s->opt.mfma_wg > 0) wg_per_cu = (uint32_t)s->opt.mfma_wg;  // store option (experiments)
    const uint32_t ldq = (s->dim + MKC - 1) / MKC * MKC;
    const uint32_t ldh = i8 ? ((s->dim + 127u) & ~127u) / 2 : (s->dim + 63u) & ~63u;  // hi pass: operand rows are ldh 16-bit units = ldh / 2 four-byte units (int8: ld8 bytes)
    const uint16_t* hi_img = nullptr;
    float hi_rel = 0.0f, hi_scale = 1.0f;
    bool hi_f16 = false;  // the plane (and therefore the query operands) are IEEE half, pre-scaled by powers of two
    const float* i8_scale = nullptr;
    if (i8) {
        const int8_t* img8 = nullptr;
        int rch = ensure_i8_plane(s, &img8, &i8_scale, &hi_rel);
        if (rch) return rch;
        if (!img8) return fail(OTT_ERR_UNSUPPORTED, "run_mfma: the int8 plane is unavailable");
        hi_img = (const uint16_t*)img8;
    } else if (hi) {
        int rch = ensure_hi_plane(s, &hi_img, &hi_rel, &hi_f16, &hi_scale);
        if (rch) return rch;
        if (!hi_img) return fail(OTT_ERR_UNSUPPORTED, "run_mfma: the hi plane is unavailable");
    }
    const bool cosine = d->metric == OTT_METRIC_COSINE;
    const bool tmax = d->take == OTT_TAKE_MAX;
    const uint32_t k = (uint32_t)k_q;
    // T = re-scored candidates per query: k plus slack, a multiple of 64.  The hi pass's bound is ~100x wider, so it needs
    // every row within it of the k-th score among the re-scored: at least 2k + 56
    int E = 1;
    // (t_min: the split pass as a later level of the cascade sees the queries whose k-th score sits in a dense neighbourhood —
    // that is why the pass before failed them — so it re-scores more: 512 as the second level, 4096 as the third)
    // (round 3: the hi pass re-scores `t_min` = 512 per query once it has failed a store's queries at 2k + 56 (ott_api.hip,
    //  hi_t512) — what certifies clustered corpora in ONE pass: 20 000 clusters of ~500 near neighbours each, 256 queries,
    //  top-100: 10.5 ms with every query through the split pass -> 4.9 ms; on uniform rows it would cost ~0.15 ms of wall per batch)
    const uint32_t hi_floor = t_min;
    // (int8: a bound ~15x the half plane's — every row within ~8e-3 of the k-th score must be among the re-scored: 512)
    // (the int8 pass: about 2.7 k rows of a uniform 768-d corpus lie within its bound of the k-th score — 4k + 88 re-scored, at most 512;
    //  a store whose queries fail at that is asked for 512 from then on, t_min)
    const uint32_t t_want = i8 ? (t_min > 4u * k + 88u ? t_min : (4u * k + 88u < 512u ? 4u * k + 88u : 512u)) : hi ? (hi_floor > 2u * k + 56u ? hi_floor : 2u * k + 56u) : (t_min > k + 28u ? t_min : k + 28u);
    while (64u * E < (t_want < 512u ? t_want : 512u) && E < 8) E *= 2;
    const bool wide = !hi && t_min > 512u;  // T = 4096: lists of 64K entries, finalize sorts 4096 candidates in LDS
    const uint32_t T = wide ? 4096u : 64u * E;
    if (wide) {  // E now only sizes the exact top-k list
        E = 1;
        while (64u * E < k && E < 8) E *= 2;
    }
    if (k > T || k > 512u) return fail(OTT_ERR_UNSUPPORTED, "run_mfma: k too large for the batch path");
    const uint32_t cap_max = wide ? 65536u : 16384u;  // a round leaves ~8 T survivors per query
    uint32_t cap = cap_max;
    {   // small stores: the list can hold every row, no need for all the slots
        uint64_t want = pl.rows_scored + 64;
        uint32_t c2 = 1024;
        while (c2 < want && c2 < cap_max) c2 <<= 1;
        cap = c2;
    }
    const std::vector<uint32_t> prefix = tile_prefix(pl, BM);
    const uint32_t n_tiles = prefix.back();

    // ---- error bound on |approx - exact| (see DESIGN.md "MFMA path: certification") -----------
    const float u = 5.9604645e-8f;  // 2^-24
    // f32 pipe: recursive-summation bounds of both orders.  Split bf16: three products per element are accumulated (3*dim
    // terms), and each element's product loses at most 3 * 2^-16 (1 + 2^-8) of |q_i v_i| to the dropped lo*lo / residual terms
    // Hi pass: |q~.v~ - q.v| = |q~.(v~ - v) + (q~ - q).v| <= ||q~|| ||v~ - v|| + ||q~ - q|| ||v|| with both rounding losses MEASURED
    // (rows: hi_rel, max over the store's regular rows; queries: per query, added in finalize_kernel), plus the accumulation terms.
    // what the relaxed filter below assumes of any query: the format's worst-case relative rounding loss (bf16 RNE 2^-8, half 2^-11)
    // int8: the queries share ONE scale per batch, so a query's loss depends on its largest element against the batch's; what the
    // relaxed filter assumes of any certified query is a measured loss of at most 2^-6 (uniform 768-d queries measure 4e-3)
    const float fmt_u = i8 ? 0.015625f : hi_f16 ? 4.8828125e-4f : 0.00390625f;
    const float qrel_cap = 1.01f * fmt_u;
    // test-only option eps_scale_ppm: every term of the bound shrunk on purpose, to show that a VIOLATED bound is noticed (the
    // measured |approximate - exact| / eps of the re-scored candidates exceeds 1) and the query falls through to the next level
    const float esc = s->opt.eps_scale_ppm == 1000000 ? 1.0f : (float)s->opt.eps_scale_ppm * 1e-6f;
    const float c_eps = esc * (i8    ? (d->metric == OTT_METRIC_EUCLIDEAN ? (2.0f * (float)s->dim + 32.0f) * u  // (||v||^2 from the stored inverse norm: see below)
                                                                          : i8_c_eps_units(s->dim) * u)
                               : hi  ? (2.5f * (float)s->dim + 32.0f) * u
                               : bf3 ? (3.75f * (float)s->dim + 32.0f) * u + 3.03f * 1.52587890625e-5f
                                     : ((d->metric == OTT_METRIC_EUCLIDEAN ? 2.0f : 1.25f) * (float)s->dim + 32.0f) * u);
    // (squared L2 on the f32 pipe: besides the two summation orders, ||v||^2 comes from the stored inverse norm, whose
    //  sequential f32 sum carries up to dim * 2^-24 of relative error itself)
    const float eps_r = hi ? esc * (1.001f * (1.0f + fmt_u) * hi_rel) : 0.0f;  // rows' share of the hi pass's rounding loss (||q~|| <= (1 + u) ||q||)
    const float r_max = hi ? eps_r + esc * (1.001f * qrel_cap) : 0.0f;         // + the most any certified query adds
    const uint32_t metric = d->metric;
    std::vector<float> qnorm(nq_pad, 0.f), qinv(nq_pad, 0.f), qamax(i8 ? nq : 0, 0.f);
    float qn_max = 0.f;
    // Norms of EIGHT queries at a time: the reference-order inverse norm is one dependent float add chain per query
    // (src/vec.rs:387-397: sequential sum of squares, separate multiply and add — this file is built with -ffp-contract=off,
    // and the baseline x86-64 target has no fused multiply-add anyway), so eight independent chains keep the host core busy.
    // The UPPER BOUND on ||q|| the error model needs comes from that same float sum with its worst-case error as margin
    // (relative error of a sequential f32 sum of non-negative terms <= dim * 2^-24; of its root, half that); the f64 sum is
    // taken only where it is needed: squared L2 (||q||^2 rides in the qinv slot) and queries whose float norm is outside
    // [1e-15, 1e18] (squares underflowing or overflowing in f32: zero, tiny, huge, non-finite — the irregular ones).
    // 1024 queries x 768 (a C4 shard's batch): the whole host prepare phase in front of the first launch 0.32-0.49 -> 0.18-0.24 ms
    // (diagnostic build's host timers, benchmarks/hostprof.py); 256 queries 0.10 -> 0.06 ms.
    {
        constexpr uint32_t G = 8;
        const bool need_f64 = metric == OTT_METRIC_EUCLIDEAN;
        const uint32_t dim = s->dim;
        for (uint32_t i0 = 0; i0 < nq; i0 += G) {
            const uint32_t g = nq - i0 < G ? nq - i0 : G;
            const float* v[G];
            float fs[G];
            double ds[G];
            for (uint32_t a = 0; a < G; a++) {
                v[a] = d->queries + (size_t)(i0 + (a < g ? a : 0)) * dim;
                fs[a] = 0.0f;
                ds[a] = 0.0;
            }
            if (need_f64) {
                for (uint32_t j = 0; j < dim; j++)
                    for (uint32_t a = 0; a < G; a++) {
                        const float x = v[a][j];
                        const float sq = x * x;
                        fs[a] = fs[a] + sq;
                        ds[a] += (double)x * x;
                    }
            } else {
                for (uint32_t j = 0; j < dim; j++)
                    for (uint32_t a = 0; a < G; a++) {
                        const float x = v[a][j];
                        const float sq = x * x;
                        fs[a] = fs[a] + sq;
                    }
            }
            if (i8)  // the largest element of every query: the batch's ONE quantisation scale comes from them
                for (uint32_t a = 0; a < g; a++) {
                    float m = 0.0f;
                    for (uint32_t j = 0; j < dim; j++) m = fmaxf(m, fabsf(v[a][j]));
                    qamax[i0 + a] = m;
                }
            for (uint32_t a = 0; a < g; a++) {
                const uint32_t i = i0 + a;
                const float nrm = sqrtf(fs[a]);
                double nd;
                if (need_f64) nd = sqrt(ds[a]);
                else if (nrm >= 1e-15f && nrm <= 1e18f) nd = (double)nrm * (1.0 + (double)dim * 5.9604644775390625e-8);
                else {
                    double t = 0.0;
                    for (uint32_t j = 0; j < dim; j++) t += (double)v[a][j] * (double)v[a][j];
                    nd = sqrt(t);
                    ds[a] = t;
                }
                qnorm[i] = (float)(nd * (1.0 + 1e-6));
                if (metric == OTT_METRIC_EUCLIDEAN) qinv[i] = (float)ds[a];  // ||q||^2 rides in the qinv slot
                else qinv[i] = nrm != 0.0f ? 1.0f / nrm : 0.0f;
                if (qnorm[i] > qn_max) qn_max = qnorm[i];
            }
        }
    }
    // Half plane, dot / squared L2: the operands are the RAW queries times the reciprocal of the plane's factor (below); when
    // that product is far outside half's range (query norms x sqrt(row norms) beyond ~10^5 per element: vectors with norms
    // in the tens of thousands) every operand would overflow and the pass could certify nothing — it is skipped outright
    // (every query reported open: the caller goes on with the split pass, whose operands keep f32's exponent range).
    // Cosine operands are unit vectors and never get here.
    if (hi && hi_f16 && !cosine && qn_max / hi_scale > 262016.0f) {
        out.assign(nq, {});
        uncertified.assign(nq, 1u);
        st.path_used = OTT_PATH_MFMA;
        st.passes = 0;
        st.score_ns = st.merge_ns = 0;
        st.rescored = 0;
        st.bytes_scanned = 0;
        return OTT_OK;
    }
    const float max_norm = s->min_pos_inv < __builtin_inff() ? (1.0f / s->min_pos_inv) * 1.000001f : 0.0f;
    float eps_max;
    if (cosine) eps_max = c_eps + r_max;
    else if (metric == OTT_METRIC_DOT) eps_max = (c_eps + r_max) * max_norm * qn_max;
    else eps_max = c_eps * (qn_max + max_norm) * (qn_max + max_norm) + 2.0f * r_max * qn_max * max_norm;
    if (!(eps_max < __builtin_inff())) return fail(OTT_ERR_UNSUPPORTED, "run_mfma: non-finite error bound");
    float flo = -__builtin_inff(), fhi = __builtin_inff();
    switch (d->filter_cmp) {
        case OTT_CMP_GT: case OTT_CMP_GTE: flo = d->filter_thr - eps_max; break;
        case OTT_CMP_LT: case OTT_CMP_LTE: fhi = d->filter_thr + eps_max; break;
        case OTT_CMP_EQ: flo = d->filter_thr - eps_max; fhi = d->filter_thr + eps_max; break;
        default: break;
    }

    // ---- buffers ------------------------------------------------------------------------------
    int rc;
    // ONE input block, staged in pinned memory and uploaded with one copy: Q (zero padded) | qinv | qnorm | tau |
    // cntA | cntB | overflow (zeros) | runs | tile prefix.  (Six copies and three memsets were ~50 us of blit kernels
    // in front of every batch.)
    const size_t q_bytes = (size_t)nq_pad * ldq * 4;
    const size_t off_qinv = q_bytes, off_qnorm = off_qinv + (size_t)nq_pad * 4, off_tau = off_qnorm + (size_t)nq_pad * 4;
    const size_t off_cntA = (off_tau + (size_t)nq_pad * 4 + 127) & ~(size_t)127, off_cntB = off_cntA + (size_t)nq_pad * CNT_STRIDE * 4;
    const size_t off_over = off_cntB + (size_t)nq_pad * CNT_STRIDE * 4;
    const size_t off_qrel = off_over + (size_t)nq_pad * 4;  // hi pass: measured rounding loss of each operand row
    const size_t off_gate = off_qrel + (size_t)nq_pad * 4;  // emission threshold of the scoring rounds (select_kernel)
    const size_t off_runs = (off_gate + (size_t)nq_pad * 4 + 15) & ~(size_t)15;
    const size_t off_prefix = off_runs + pl.runs.size() * sizeof(ott_run);
    // cosine: the MFMA operand is the query pre-scaled by 1/||q|| (one multiply less per accumulator in the epilogue; the
    // extra rounding, one ulp per element, is inside the error bound's slack); the exact re-score needs the raw query
    const bool own_operand = cosine || bf3;
    const size_t off_qraw = own_operand ? ((off_prefix + prefix.size() * 4 + 127) & ~(size_t)127) : 0;
    const size_t tot = own_operand ? off_qraw + q_bytes : off_prefix + prefix.size() * 4;
    if ((rc = s->m_Q.ensure(tot))) return rc;
    if ((rc = s->m_candA.ensure((size_t)nq_pad * cap * sizeof(CandEntry)))) return rc;
    if ((rc = s->m_candB.ensure((size_t)nq_pad * cap * sizeof(CandEntry)))) return rc;
    // results: finalize_kernel writes hits | counts | certification flags straight into pinned host memory (no D2H copies
    // behind the launch: three copy enqueues were ~40 us of a small batch)
    const size_t hb = (size_t)nq * k * sizeof(ott_hit), cb = (size_t)nq * 8, ub = (size_t)nq * 4;
    if ((rc = s->h_hits.ensure(hb + cb + 2 * ub))) return rc;  // hits | counts | certification flags | error ratios
    char* hh = (char*)s->h_hits.p;
    char* hh_dev = nullptr;
    OTT_HIP(hipHostGetDevicePointer((void**)&hh_dev, hh, 0));
    if ((rc = s->h_stage.ensure(tot))) return rc;
    char* hs = (char*)s->h_stage.p;
    // bf3: the operand region [0, q_bytes) is produced on the GPU (padded query rows included), so it is neither cleared nor
    // uploaded; everything behind it is
    const size_t up0 = bf3 ? q_bytes : 0;
    memset(hs + up0, 0, tot - up0);
    float* hQ = (float*)hs;
    for (uint32_t i = 0; i < nq; i++) {
        const float* src = d->queries + (size_t)i * s->dim;
        if (own_operand) memcpy(hs + off_qraw + (size_t)i * ldq * 4, src, (size_t)s->dim * 4);
        if (bf3) {
            // the operand block ([32 hi | 32 lo] bf16 per 32-k stage, pre-scaled by 1/||q|| for cosine) is produced on the GPU
            // from the raw queries below: at 1024 queries the host loop was 4 ms
        } else if (cosine) {
            for (uint32_t j = 0; j < s->dim; j++) hQ[(size_t)i * ldq + j] = src[j] * qinv[i];
        } else {
            memcpy(hQ + (size_t)i * ldq, src, (size_t)s->dim * 4);
        }
    }
    float* hqinv = (float*)(hs + off_qinv);
    float* hqnorm = (float*)(hs + off_qnorm);
    float* htau = (float*)(hs + off_tau);
    for (uint32_t i = 0; i < nq_pad; i++) {
        hqinv[i] = qinv[i];
        hqnorm[i] = qnorm[i];
        // padded queries never emit; real ones start fully open.  A query with a non-finite or astronomically
        // large norm is outside the error model: it is excluded here (NaN threshold = empty interval) and answered
        // by the exact path
        const bool irregular_q = i < nq && !(qnorm[i] <= 1e18f && (qnorm[i] == 0.0f || qnorm[i] >= 1e-18f));
        htau[i] = (i < nq && !irregular_q) ? (tmax ? -__builtin_inff() : __builtin_inff()) : __builtin_nanf("");
        ((float*)(hs + off_gate))[i] = htau[i];
    }
    memcpy(hs + off_runs, pl.runs.data(), pl.runs.size() * sizeof(ott_run));
    memcpy(hs + off_prefix, prefix.data(), prefix.size() * 4);
    OTT_HIP(hipMemcpyAsync((char*)s->m_Q.p + up0, hs + up0, tot - up0, hipMemcpyHostToDevice, s->stream));
    char* dblk = (char*)s->m_Q.p;
    // half operands: the queries are multiplied by the RECIPROCAL of the plane's power-of-two factor, so the accumulators hold
    // the plain dot products (nothing to undo in the epilogue).  The plane's factor is 2^-round(log2(largest regular norm) / 4)
    // (ensure_hi_plane: the compromise between unit-length cosine operands and raw dot / L2 operands); a query whose operand
    // leaves half's range measures a large rounding loss (or overflows: loss 1) and is not certified here
    const float q_scale = (hi && hi_f16) ? 1.0f / hi_scale : 1.0f;
    float i8_qscale = 1.0f;
    if (i8) {
        // ONE quantisation scale for the batch's operands (cosine: the unit-length queries): the largest element any of them has,
        // over 127.  It folds into the kernel's row factors, so the epilogue needs no per-query factor.  A query whose elements are
        // small beside the batch's largest simply measures a larger loss (rel_out) and may go uncertified to the next level.
        float m = 0.0f;
        for (uint32_t i = 0; i < nq; i++) {
            const float e = cosine ? qamax[i] * qinv[i] : qamax[i];
            if (e == e && e < __builtin_inff() && e > m) m = e;
        }
        i8_qscale = m > 0.0f ? m / 127.0f : 1.0f;
        if ((rc = launch_i8_rows(s->stream, (const float*)(dblk + off_qraw), ldq, s->dim, ldh * 2, 0, nq_pad, (int8_t*)dblk,
                                 cosine ? (const float*)(dblk + off_qinv) : nullptr, i8_qscale, nullptr, (float*)(dblk + off_qrel), nullptr, nullptr, 2.0f,
                                 nullptr, s->n_cu)))
            return rc;
    } else if (hi) {
        if ((rc = launch_hi_rows(s->stream, (const float*)(dblk + off_qraw), ldq, s->dim, ldh, nq_pad, (uint16_t*)dblk,
                                 cosine ? (const float*)(dblk + off_qinv) : nullptr, (float*)(dblk + off_qrel), s->n_cu, hi_f16, q_scale)))
            return rc;
    } else if (bf3 && (rc = launch_split_rows(s->stream, (const float*)(dblk + off_qraw), ldq, s->dim, ldq, nq_pad, (uint16_t*)dblk,
                                              cosine ? (const float*)(dblk + off_qinv) : nullptr, s->n_cu)))
        return rc;  // rows nq .. nq_pad of the raw block are zero, so are their operand rows
    float* d_qinv = (float*)(dblk + off_qinv);
    float* d_qnorm = (float*)(dblk + off_qnorm);
    float* d_tau = (float*)(dblk + off_tau);
    float* d_gate = (float*)(dblk + off_gate);
    uint32_t* d_cntA = (uint32_t*)(dblk + off_cntA);
    uint32_t* d_cntB = (uint32_t*)(dblk + off_cntB);
    uint32_t* d_over = (uint32_t*)(dblk + off_over);

    MfmaParams p;
    memset(&p, 0, sizeof(p));
    p.rows = s->d_rows;
    p.inv = s->d_inv;
    p.flag = s->d_flag;
    // operand mode of the candidate pass: 0 = f32 matrix pipe, 1 = split bf16 with the rows split in registers, 2 = split
    // bf16 from the store's pre-split batch image (built / extended here on first use; mode 1 when it does not fit)
    int bf3mode = 0;
    if (i8) {
        bf3mode = 5;
        p.img = hi_img;
        p.i8_scale = i8_scale;
        p.i8_qscale = i8_qscale;
    } else if (hi) {
        bf3mode = hi_f16 ? 4 : 3;
        p.img = hi_img;
    } else if (bf3) {
        const uint16_t* img = nullptr;
        if ((rc = ensure_batch_image(s, &img))) return rc;
        bf3mode = img ? 2 : 1;
        p.img = img;
    }
    p.Q = (const float*)s->m_Q.p;
    p.qinv = d_qinv;
    p.tau = d_gate;  // the scoring rounds emit against the gate (== tau unless speculative)
    p.runs = (const ott_run*)(dblk + off_runs);
    p.tile_prefix = (const uint32_t*)(dblk + off_prefix);
    p.row_mask = d_mask;
    p.row_mask_bits = mask_bits;
    p.cap = cap;
    p.ld = s->ld;
    p.dim = s->dim;
    p.ldq = hi ? ldh / 2 : ldq;
    p.n_runs = (uint32_t)pl.runs.size();
    p.metric = metric;
    p.take_max = tmax;
    p.flo = flo;
    p.fhi = fhi;

    uint32_t* cnt_cur = d_cntA;
    uint32_t* cnt_oth = d_cntB;
    CandEntry* cand_cur = (CandEntry*)s->m_candA.p;
    CandEntry* cand_oth = (CandEntry*)s->m_candB.p;

    // mfma_debug option: in-kernel s_memtime stamps (never quote such a run's time).  The stamped kernel variants exist only
    // in a library built with -DOTT_MFMA_DEBUG_BUILD (make EXTRA=-DOTT_MFMA_DEBUG_BUILD); the shipped one carries none.
#ifdef OTT_MFMA_DEBUG_BUILD
    const bool dbg_on = s->opt.mfma_debug;
#define OTT_KERN(NBv, BFv) (dbg_on ? mfma_score_kernel<NBv, true, BFv> : mfma_score_kernel<NBv, false, BFv>)
#else
    constexpr bool dbg_on = false;
    if (s->opt.mfma_debug) return fail(OTT_ERR_UNSUPPORTED, "mfma_debug needs a library built with -DOTT_MFMA_DEBUG_BUILD");
#define OTT_KERN(NBv, BFv) (mfma_score_kernel<NBv, false, BFv>)
#endif
    void (*kern)(MfmaParams) = nullptr;
    switch (NB) {
        case -1: kern = OTT_KERN(-1, 0); break;
#define OTT_PICK(NBv) kern = bf3mode == 5 ? OTT_KERN(NBv, 5) : bf3mode == 4 ? OTT_KERN(NBv, 4) : bf3mode == 3 ? OTT_KERN(NBv, 3) : bf3mode == 2 ? OTT_KERN(NBv, 2) : bf3mode == 1 ? OTT_KERN(NBv, 1) : OTT_KERN(NBv, 0)
        case 0: OTT_PICK(0); break;
        case 1: OTT_PICK(1); break;
        case 2: OTT_PICK(2); break;
        default: OTT_PICK(4); break;
#undef OTT_PICK
#undef OTT_KERN
    }
    const uint32_t wg_slots = (uint32_t)s->n_cu * wg_per_cu;  // the persistent grid
    const size_t smem_bytes = MFMA_SMEM;
    {   // once per kernel variant and device (the attribute call is not free: it sat in front of every batch)
        static std::mutex attr_mu;
        static std::vector<std::pair<const void*, int>> attr_done;
        std::lock_guard<std::mutex> g(attr_mu);
        const std::pair<const void*, int> key((const void*)kern, s->device);
        if (std::find(attr_done.begin(), attr_done.end(), key) == attr_done.end()) {
            OTT_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_bytes));
            attr_done.push_back(key);
        }
    }
    if (dbg_on) {
        if ((rc = s->d_misc.ensure((size_t)wg_slots * 14 * 8))) return rc;
        OTT_HIP(hipMemsetAsync(s->d_misc.p, 0, (size_t)wg_slots * 14 * 8, s->stream));
        p.dbg = (unsigned long long*)s->d_misc.p;
        p.dbg_wgs = wg_slots;
        p.dbg_abl = (uint32_t)s->opt.mfma_abl;
    }
    const double hm1 = host_ms();
    OTT_HIP(hipEventRecord(s->ev[0], s->stream));
    // geometric rounds: 32 tiles (8192 rows, thresholds open: every pair is listed), then x `growth` per round.  With
    // rows in no particular order a round of g x (rows so far) leaves ~T*g survivors per query
    const uint32_t growth = (uint32_t)s->opt.mfma_growth;  // store option (experiments); default 8
    uint32_t begin = 0, width = 32;
    while (begin < n_tiles) {
        uint32_t end = begin + width;
        if (end > n_tiles || n_tiles - end < width) end = n_tiles;  // fold a short tail into this round
        const uint32_t tiles = end - begin;
        const uint32_t slots = wg_slots;  // persistent workgroups: one per CU (more only through the mfma_wg option)
        const uint32_t grid = tiles < slots ? tiles : slots;
        // the first round lists every pair: with one slot per pair there is nothing to count
        const bool dense = begin == 0 && (uint64_t)tiles * BM <= cap && !s->opt.mfma_no_dense;
        if (dense) OTT_HIP(hipMemsetD32Async((hipDeviceptr_t)cnt_cur, (int)(tiles * BM), (size_t)nq_pad * CNT_STRIDE, s->stream));
        const uint32_t qstep = BN * qblk_max;
        for (uint32_t qb = 0; qb < nq_pad; qb += qstep) {
            p.tile_begin = begin;
            p.tile_end = end;
            p.q_base = qb;
            p.n_qblk = std::min<uint32_t>(qblk_max, (nq_pad - qb) / BN);
            // 2 or 4 blocks and enough tiles to fill the persistent grid: the blocks of a tile on sibling workgroups of one XCD
            p.coop = (s->opt.mfma_coop != 0 && (p.n_qblk == 2 || p.n_qblk == 4) && grid == slots && slots % (8u * p.n_qblk) == 0 &&
                      (uint64_t)tiles * p.n_qblk >= slots) ? p.n_qblk : 0u;
            p.cnt = cnt_cur;
            p.cand = cand_cur;
            p.dense = dense ? 1u : 0u;
            hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem_bytes, s->stream, p);
            OTT_HIP(hipGetLastError());
        }
        // speculative gate for the rounds that follow: the j-th best so far with j = 8 T x (share of the tiles seen), at least 8
        // — about 8 T rows of the whole store are expected above it; j >= T (an eighth of the store seen) = conservative
        uint32_t j_gate = 0;
        if (spec_gate && end < n_tiles) {
            const uint64_t jj = (8ull * T * end + n_tiles - 1) / n_tiles;
            j_gate = jj < 8 ? 8u : (uint32_t)jj;
            if (j_gate >= T) j_gate = 0;
        }
        hipLaunchKernelGGL(select_kernel, dim3(nq_pad), dim3(SEL_THREADS), 0, s->stream, cand_cur, cnt_cur, cand_oth, cnt_oth,
                           d_tau, d_over, cap, T, tmax ? 1u : 0u, d_gate, j_gate);  // keep the T best: k + slack
        OTT_HIP(hipGetLastError());
        std::swap(cnt_cur, cnt_oth);
        std::swap(cand_cur, cand_oth);
        // (two workgroups per CU, mfma_wg = 2: 512 slots, so the second round takes 512 tiles (x16) rather than leave half idle)
        width *= (begin == 0 && wg_per_cu == 2 && growth == 8) ? 16 : growth;
        begin = end;
    }
    OTT_HIP(hipEventRecord(s->ev[1], s->stream));
    if (dbg_on) {
        const size_t nwg = wg_slots;  // the last (largest) round ran with this grid
        std::vector<unsigned long long> h(nwg * 12);
        OTT_HIP(hipMemcpyAsync(h.data(), s->d_misc.p, h.size() * 8, hipMemcpyDeviceToHost, s->stream));
        OTT_HIP(hipStreamSynchronize(s->stream));
        double a = 0, b = 0, c = 0, t = 0;
        double rt = 0;
        for (size_t i = 0; i < nwg; i++) rt += h[nwg * 4 + i];
        for (size_t i = 0; i < nwg; i++) { a += h[i * 4]; b += h[i * 4 + 1]; c += h[i * 4 + 2]; t += h[i * 4 + 3]; }
        double wd = 0, wb = 0;
        for (size_t i = 0; i < nwg; i++) { wd += h[nwg * 5 + i]; wb += h[nwg * 6 + i]; }
        double e[5] = {0, 0, 0, 0, 0};
        for (int j = 0; j < 5; j++)
            for (size_t i = 0; i < nwg; i++) e[j] += h[nwg * (7 + j) + i];
        if (t > 0) fprintf(stderr, "[ott mfma dbg] per tile (s_memtime ticks, wave 0): prologue %.0f  K-loop %.0f (of which waiting for its DMA %.0f, at the stage barrier %.0f)  epilogue %.0f (setup %.0f, walk %.0f, queue flush %.0f, drain of its stores %.0f; wave 0 listed something in %.0f %% of the tiles)  (tiles %.0f; ~%.0f MHz)\n", a / t, b / t, wd / t, wb / t, c / t, e[0] / t, e[1] / t, e[2] / t, e[3] / t, 100.0 * e[4] / t, t, rt > 0 ? (a + b + c) / rt * 100.0 : 0.0);
    }

    FinalParams f;
    memset(&f, 0, sizeof(f));
    f.rows = s->d_rows;
    f.inv = s->d_inv;
    f.Q = (const float*)(dblk + off_qraw);  // == the operand block unless it was pre-scaled (cosine)
    f.qinv = d_qinv;
    f.tau = d_tau;
    f.gate = d_gate;
    f.cnt = cnt_cur;
    f.cand = cand_cur;
    f.overflow = d_over;
    f.out = (ott_hit*)hh_dev;
    f.out_cnt = (uint64_t*)(hh_dev + hb);
    f.uncertified = (uint32_t*)(hh_dev + hb + cb);
    f.base_offset = s->base_offset;
    f.cap = cap;
    f.ld = s->ld;
    f.dim = s->dim;
    f.ldq = ldq;
    f.nq = nq;
    f.k = k;
    f.T = T;
    f.out_stride = k;  // only the k exact hits travel back
    f.metric = metric;
    f.take_max = tmax;
    f.cmp = d->filter_cmp;
    f.reduce = s->reduce;
    f.thr = d->filter_thr;
    f.eps_c = c_eps;
    f.max_norm = max_norm;
    f.qnorm = d_qnorm;
    f.qrel = hi ? (const float*)(dblk + off_qrel) : nullptr;
    f.qrel_cap = hi ? qrel_cap : 0.0f;
    f.eps_r = eps_r;
    f.eps_scale = esc;
    f.err_ratio = (uint32_t*)(hh_dev + hb + cb + ub);
    if (wide) {
        switch (E) {
            case 1: hipLaunchKernelGGL((finalize_kernel<1, 4096>), dim3(nq), dim3(64 * FIN_WAVES), 0, s->stream, f); break;
            case 2: hipLaunchKernelGGL((finalize_kernel<2, 4096>), dim3(nq), dim3(64 * FIN_WAVES), 0, s->stream, f); break;
            case 4: hipLaunchKernelGGL((finalize_kernel<4, 4096>), dim3(nq), dim3(64 * FIN_WAVES), 0, s->stream, f); break;
            default: hipLaunchKernelGGL((finalize_kernel<8, 4096>), dim3(nq), dim3(64 * FIN_WAVES), 0, s->stream, f); break;
        }
    } else {
        switch (E) {
            case 1: hipLaunchKernelGGL((finalize_kernel<1, 64>), dim3(nq), dim3(64 * FIN_WAVES), 0, s->stream, f); break;
            case 2: hipLaunchKernelGGL((finalize_kernel<2, 128>), dim3(nq), dim3(64 * FIN_WAVES), 0, s->stream, f); break;
            case 4: hipLaunchKernelGGL((finalize_kernel<4, 256>), dim3(nq), dim3(64 * FIN_WAVES), 0, s->stream, f); break;
            default: hipLaunchKernelGGL((finalize_kernel<8, 512>), dim3(nq), dim3(64 * FIN_WAVES), 0, s->stream, f); break;
        }
    }
    OTT_HIP(hipGetLastError());
    OTT_HIP(hipEventRecord(s->ev[2], s->stream));

    // ---- results to host --------------------------------------------------------------------------
    const double hm2 = host_ms();
    OTT_HIP(hipStreamSynchronize(s->stream));
    const double hm3 = host_ms();
    const ott_hit* hits = (const ott_hit*)hh;
    const uint64_t* cnts = (const uint64_t*)(hh + hb);
    const uint32_t* unc = (const uint32_t*)(hh + hb + cb);
    out.assign(nq, {});
    uncertified.assign(nq, 0);
    uint64_t rescored = 0;
    for (uint32_t q = 0; q < nq; q++) {
        out[q].assign(hits + (size_t)q * k, hits + (size_t)q * k + cnts[q]);
        uncertified[q] = (unc[q] || !(qnorm[q] <= 1e18f && (qnorm[q] == 0.0f || qnorm[q] >= 1e-18f))) ? 1u : 0u;
        if (unc[q] == 2u) st.gate_failed++;  // would have been certified but for its speculative gate
        {
            float er;
            memcpy(&er, hh + hb + cb + ub + (size_t)q * 4, 4);
            if (er > st.err_ratio_max) st.err_ratio_max = er;
            // The certification checks itself: eps is a bound on |approximate - exact|, and every re-scored candidate MEASURES that
            // difference.  A ratio above 1 is a violated bound (the matrix unit's accumulation is not documented; the model behind
            // eps is this library's): whatever the kernel concluded from it is void — the query goes to the next level of the
            // cascade like any uncertified one, and finally to the exact-order kernel (src/vec_compute.rs:9-54).
            if (er > 1.0f) {
                st.bound_violations++;
                uncertified[q] = 1u;
            }
        }
        rescored += T;
    }
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s->ev[0], s->ev[1]) == hipSuccess) st.score_ns = (uint64_t)(ms * 1e6);
    if (hipEventElapsedTime(&ms, s->ev[1], s->ev[2]) == hipSuccess) st.merge_ns = (uint64_t)(ms * 1e6);
    st.path_used = OTT_PATH_MFMA;
    st.passes = (nq_pad / BN + qblk_max - 1) / qblk_max;  // passes over the plane from HBM
    st.rescored = rescored;
    st.bytes_scanned = (uint64_t)st.passes * pl.rows_scored * ((uint64_t)s->dim * 4 + (cosine ? 4 : 0));
    if (dbg_on) fprintf(stderr, "[ott mfma dbg] host ms: prepare %.3f  enqueue %.3f  wait %.3f  unpack %.3f\n", hm1 - hm0, hm2 - hm1, hm3 - hm2, host_ms() - hm3);
    return OTT_OK;
}


// ---------------------------------------------------------------------------------------------
// The int8 level for ONE query as a streaming sweep (round 5).  A single query is a vector: the matrix cores have nothing to
// tile, and the cascade's five geometric rounds (each a scoring launch + select_kernel) are ~0.25 ms of fixed cost around a
// 1.1 ms stream.  Here exact_kernel<..., I8> streams the int8 plane once (v_dot4, exact i32 sums, a lane per row), keeps the T
// best APPROXIMATE scores in its wave lists, merge_rank_kernel folds the block lists, and finalize_kernel re-scores the T rows
// in the reference's f32 order and certifies against the measured quantisation bound — three launches, one wait.
// ---------------------------------------------------------------------------------------------
// the merged approximate list ([T] ott_hit, index = local row) -> the (cnt, cand, tau, gate, overflow) form finalize_kernel reads
__global__ void i8_hits_to_cand_kernel(const ott_hit* hits, const uint64_t* count, uint32_t T, uint32_t take_max, CandEntry* cand, uint32_t* cnt,
                                       float* tau, float* gate, uint32_t* overflow) {
    const uint32_t n = (uint32_t)(count[0] < T ? count[0] : T);
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        CandEntry e;
        e.row = (uint32_t)hits[i].index;
        e.score = hits[i].score;
        cand[i] = e;
    }
    if (threadIdx.x == 0) {
        cnt[0] = n;
        // what a row NOT listed can score at best (approximately): the T-th listed score when the list is full — every wave
        // dropped only rows below its own T-th best — else nothing is outside (every passing row is listed)
        const float open = take_max ? -__builtin_inff() : __builtin_inff();
        float t = open;
        if (n == T && T > 0) {
            const float last = hits[T - 1].score;
            t = (last - last == 0.0f) ? last : __uint_as_float(0x7FC00000u);  // T forced rows: nothing can be certified
        }
        tau[0] = t;
        gate[0] = t;
        overflow[0] = 0u;
    }
}

int run_i8_single(ott_store* s, const ott_query_desc* d, const RunPlan& pl, uint64_t k_q, const uint64_t* d_mask, uint64_t mask_bits,
                  std::vector<std::vector<ott_hit>>& out, std::vector<uint32_t>& uncertified, ott_stats& st, uint32_t t_min) {
    if (d->nq != 1 || d->metric == OTT_METRIC_EUCLIDEAN || d->filter_cmp == OTT_CMP_EQ) return fail(OTT_ERR_UNSUPPORTED, "run_i8_single: one query, cosine / dot, no equality filter");
    const int8_t* img8 = nullptr;
    const float* i8_scale = nullptr;
    float i8_rel = 0.0f;
    int rc = ensure_i8_plane(s, &img8, &i8_scale, &i8_rel);
    if (rc) return rc;
    if (!img8) return fail(OTT_ERR_UNSUPPORTED, "run_i8_single: the int8 plane is unavailable");
    const uint32_t dim = s->dim, ld8 = (dim + 127u) & ~127u, ldq = (dim + MKC - 1) / MKC * MKC;
    if (ld8 > OTT_QEMB_MAX * 4) return fail(OTT_ERR_UNSUPPORTED, "run_i8_single: the query does not fit the kernel arguments");
    const bool cosine = d->metric == OTT_METRIC_COSINE, tmax = d->take == OTT_TAKE_MAX;
    const uint32_t k = (uint32_t)k_q;
    // the list the sweep keeps: 128 candidates while k <= 24 (about 2.7 k rows of a uniform 768-d corpus lie within the bound of the
    // k-th score), what the cascade's int8 level would re-score otherwise
    uint32_t t_want = t_min > 128u ? t_min : (k <= 24u ? 128u : (4u * k + 88u < 512u ? 4u * k + 88u : 512u));
    int E = 2;
    while (64u * E < t_want && E < 8) E *= 2;
    const uint32_t T = 64u * E;
    if (k > T) return fail(OTT_ERR_UNSUPPORTED, "run_i8_single: k too large");
    int Ek = 1;
    while (64u * Ek < k && Ek < 8) Ek *= 2;  // (finalize's exact list; its LDS arrays are sized by T)
    (void)Ek;

    // ---- the query: norm, int8 operand, measured loss ---------------------------------------------------------------
    const float* q = d->queries;
    const float q_inv = host_inv_norm_exact(q, dim);
    double n2 = 0.0;
    float amax = 0.0f;
    for (uint32_t i = 0; i < dim; i++) {
        n2 += (double)q[i] * q[i];
        amax = fmaxf(amax, fabsf(q[i]));
    }
    const float qnorm = (float)(sqrt(n2) * (1.0 + 1e-6));
    const bool irregular_q = !(qnorm <= 1e18f && (qnorm == 0.0f || qnorm >= 1e-18f));
    out.assign(1, {});
    uncertified.assign(1, 1u);
    st.path_used = OTT_PATH_MFMA;
    if (irregular_q) return OTT_OK;  // outside the error model: the next level / the exact path answers
    const float pf = cosine ? q_inv : 1.0f;
    const float e_max = amax * pf;
    const float s_q = (e_max > 0.0f && e_max < __builtin_inff()) ? e_max / 127.0f : 1.0f;
    const float inv_sq = 1.0f / s_q;
    int8_t q8[OTT_QEMB_MAX * 4];
    memset(q8, 0, ld8);
    double se = 0.0, sx = 0.0;
    for (uint32_t i = 0; i < dim; i++) {
        const float xe = q[i] * pf;
        float t = rintf(xe * inv_sq);
        t = t == t ? fminf(fmaxf(t, -127.0f), 127.0f) : 0.0f;
        q8[i] = (int8_t)(int)t;
        const double df = (double)xe - (double)s_q * (double)(int)t;
        se += df * df;
        sx += (double)xe * (double)xe;
    }
    float qrel = sx > 0.0 ? (float)(sqrt(se / sx) * 1.0001) : 0.0f;
    if (!(qrel <= 1.0f)) qrel = 1.0f;

    // ---- error bound (as run_mfma's int8 level) -----------------------------------------------------------------------
    const float u = 5.9604645e-8f;
    const float esc = s->opt.eps_scale_ppm == 1000000 ? 1.0f : (float)s->opt.eps_scale_ppm * 1e-6f;
    const float fmt_u = 0.015625f, qrel_cap = 1.01f * fmt_u;
    const float c_eps = esc * i8_c_eps_units(dim) * u;  // both sides of |approximate - exact-order|: see i8_c_eps_units
    const float eps_r = esc * (1.001f * (1.0f + fmt_u) * i8_rel);
    const float r_max = eps_r + esc * (1.001f * qrel_cap);
    const float max_norm = s->min_pos_inv < __builtin_inff() ? (1.0f / s->min_pos_inv) * 1.000001f : 0.0f;
    const float eps_max = cosine ? c_eps + r_max : (c_eps + r_max) * max_norm * qnorm;
    if (!(eps_max < __builtin_inff())) return fail(OTT_ERR_UNSUPPORTED, "run_i8_single: non-finite error bound");

    // ---- buffers: [raw query f32 (ldq) | qinv | qnorm | qrel | tau | gate | cnt (one line) | overflow] + candidates + lists ------
    const std::vector<uint32_t> prefix = tile_prefix(pl, 64);
    const uint32_t n_tiles = prefix.back();
    const int grid = exact_grid(s, n_tiles);
    const uint32_t KS = 64u * (uint32_t)E;
    const size_t off_qinv = (size_t)ldq * 4, off_qnorm = off_qinv + 4, off_qrel = off_qnorm + 4, off_tau = off_qrel + 4, off_gate = off_tau + 4;
    const size_t off_cnt = (off_gate + 4 + 127) & ~(size_t)127, off_over = off_cnt + CNT_STRIDE * 4, off_runs = (off_over + 4 + 15) & ~(size_t)15;
    const size_t off_prefix = off_runs + pl.runs.size() * sizeof(ott_run), off_q8 = (off_prefix + prefix.size() * 4 + 15) & ~(size_t)15;
    const size_t tot = off_q8 + ld8;
    if ((rc = s->m_Q.ensure(tot))) return rc;
    if ((rc = s->h_stage.ensure(tot))) return rc;
    if ((rc = s->m_candA.ensure((size_t)T * sizeof(CandEntry)))) return rc;
    if ((rc = s->d_lists.ensure((size_t)grid * KS * sizeof(Cand)))) return rc;
    const size_t cnt_pad = 64;
    if ((rc = s->d_hits.ensure(cnt_pad + (size_t)KS * sizeof(ott_hit)))) return rc;
    const size_t hb = (size_t)k * sizeof(ott_hit), cb = 8, ub = 4;
    if ((rc = s->h_hits.ensure(hb + cb + 2 * ub))) return rc;
    char* hh = (char*)s->h_hits.p;
    char* hh_dev = nullptr;
    OTT_HIP(hipHostGetDevicePointer((void**)&hh_dev, hh, 0));
    char* hs = (char*)s->h_stage.p;
    memset(hs, 0, tot);
    memcpy(hs, q, (size_t)dim * 4);
    *(float*)(hs + off_qinv) = q_inv;
    *(float*)(hs + off_qnorm) = qnorm;
    *(float*)(hs + off_qrel) = qrel;
    memcpy(hs + off_runs, pl.runs.data(), pl.runs.size() * sizeof(ott_run));
    memcpy(hs + off_prefix, prefix.data(), prefix.size() * 4);
    memcpy(hs + off_q8, q8, ld8);
    OTT_HIP(hipMemcpyAsync(s->m_Q.p, hs, tot, hipMemcpyHostToDevice, s->stream));
    char* dblk = (char*)s->m_Q.p;

    // ---- the sweep: approximate scores, relaxed filter, wave-list top-T ---------------------------------------------------
    ExactParams p;
    memset(&p, 0, sizeof(p));
    p.rows = reinterpret_cast<const float*>(img8);
    p.inv = s->d_inv;
    p.row_mask = d_mask;
    p.row_mask_bits = mask_bits;
    p.ld = p.dim = p.dimq = ld8 / 4;
    p.n_runs = (uint32_t)pl.runs.size();
    p.n_tiles = n_tiles;
    p.nq_total = 1;
    p.metric = d->metric;
    p.take_max = tmax;
    p.reduce = s->reduce;
    // relaxed score filter on the approximate score: nothing that can pass exactly is dropped
    p.cmp = OTT_CMP_NONE;
    switch (d->filter_cmp) {
        case OTT_CMP_GT: case OTT_CMP_GTE: p.cmp = OTT_CMP_GTE; p.thr = d->filter_thr - eps_max; break;
        case OTT_CMP_LT: case OTT_CMP_LTE: p.cmp = OTT_CMP_LTE; p.thr = d->filter_thr + eps_max; break;
        default: break;
    }
    p.k = T;
    p.list_stride = KS;
    p.lists = (Cand*)s->d_lists.p;
    p.i8 = 1;
    p.i8_qscale = s_q;
    p.i8_scale = i8_scale;
    p.flag = s->d_flag;
    if (pl.runs.size() <= 2) {  // everything the kernel needs rides in its arguments
        p.embedded = 1;
        memcpy(p.qemb, q8, ld8);
        for (size_t i = 0; i < pl.runs.size(); i++) p.eruns[i] = pl.runs[i];
        for (size_t i = 0; i < prefix.size(); i++) p.eprefix[i] = prefix[i];
    } else {  // a chunk mask with many runs (config 3: every second chunk): run table, tile prefix and the int8 query from the uploaded block
        p.queries = (const float*)(dblk + off_q8);
        p.qinv = (const float*)(dblk + off_qinv);
        p.runs = (const ott_run*)(dblk + off_runs);
        p.tile_prefix = (const uint32_t*)(dblk + off_prefix);
    }
    OTT_HIP(hipEventRecord(s->ev[0], s->stream));
    if ((rc = launch_exact_i8(s, p, E, grid))) return rc;
    ott_hit* d_hits = (ott_hit*)((char*)s->d_hits.p + cnt_pad);
    uint64_t* d_counts = (uint64_t*)s->d_hits.p;
    if ((rc = launch_merge(s, (const Cand*)s->d_lists.p, (uint32_t)grid, KS, 0, 1, T, E, tmax, 0, d_hits, KS, d_counts, 0))) return rc;
    OTT_HIP(hipEventRecord(s->ev[1], s->stream));
    hipLaunchKernelGGL(i8_hits_to_cand_kernel, dim3(1), dim3(256), 0, s->stream, (const ott_hit*)d_hits, (const uint64_t*)d_counts, T, tmax ? 1u : 0u,
                       (CandEntry*)s->m_candA.p, (uint32_t*)(dblk + off_cnt), (float*)(dblk + off_tau), (float*)(dblk + off_gate), (uint32_t*)(dblk + off_over));
    OTT_HIP(hipGetLastError());

    FinalParams f;
    memset(&f, 0, sizeof(f));
    f.rows = s->d_rows;
    f.inv = s->d_inv;
    f.Q = (const float*)dblk;
    f.qinv = (const float*)(dblk + off_qinv);
    f.tau = (const float*)(dblk + off_tau);
    f.gate = (const float*)(dblk + off_gate);
    f.cnt = (const uint32_t*)(dblk + off_cnt);
    f.cand = (const CandEntry*)s->m_candA.p;
    f.overflow = (const uint32_t*)(dblk + off_over);
    f.out = (ott_hit*)hh_dev;
    f.out_cnt = (uint64_t*)(hh_dev + hb);
    f.uncertified = (uint32_t*)(hh_dev + hb + cb);
    f.base_offset = s->base_offset;
    f.cap = T;
    f.ld = s->ld;
    f.dim = dim;
    f.ldq = ldq;
    f.nq = 1;
    f.k = k;
    f.T = T;
    f.out_stride = k;
    f.metric = d->metric;
    f.take_max = tmax;
    f.cmp = d->filter_cmp;
    f.reduce = s->reduce;
    f.thr = d->filter_thr;
    f.eps_c = c_eps;
    f.max_norm = max_norm;
    f.qnorm = (const float*)(dblk + off_qnorm);
    f.qrel = (const float*)(dblk + off_qrel);
    f.qrel_cap = qrel_cap;
    f.eps_r = eps_r;
    f.eps_scale = esc;
    f.err_ratio = (uint32_t*)(hh_dev + hb + cb + ub);
    switch (E) {
        case 2: hipLaunchKernelGGL((finalize_kernel<2, 128>), dim3(1), dim3(64 * FIN_WAVES), 0, s->stream, f); break;
        case 4: hipLaunchKernelGGL((finalize_kernel<4, 256>), dim3(1), dim3(64 * FIN_WAVES), 0, s->stream, f); break;
        default: hipLaunchKernelGGL((finalize_kernel<8, 512>), dim3(1), dim3(64 * FIN_WAVES), 0, s->stream, f); break;
    }
    OTT_HIP(hipGetLastError());
    OTT_HIP(hipEventRecord(s->ev[2], s->stream));
    OTT_HIP(hipStreamSynchronize(s->stream));
    const ott_hit* hits = (const ott_hit*)hh;
    const uint64_t cnt0 = *(const uint64_t*)(hh + hb);
    const uint32_t unc = *(const uint32_t*)(hh + hb + cb);
    out[0].assign(hits, hits + cnt0);
    uncertified[0] = unc ? 1u : 0u;
    float er;
    memcpy(&er, hh + hb + cb + ub, 4);
    if (er > st.err_ratio_max) st.err_ratio_max = er;
    if (er > 1.0f) {
        st.bound_violations++;
        uncertified[0] = 1u;
    }
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s->ev[0], s->ev[1]) == hipSuccess) st.score_ns = (uint64_t)(ms * 1e6);
    if (hipEventElapsedTime(&ms, s->ev[1], s->ev[2]) == hipSuccess) st.merge_ns = (uint64_t)(ms * 1e6);
    st.passes = 1;
    st.rescored = T;
    st.bytes_scanned = pl.rows_scored * ((uint64_t)dim * 4 + (cosine ? 4 : 0));
    return OTT_OK;
}

// The first launch of any kernel of this file loads the file's code object onto the device (10-15 ms measured in front of the
// first batch of a process).  The background plane builder asks for one kernel's attributes instead, off every query's path.
void mfma_warm(hipStream_t stream, int device) {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, (const void*)select_kernel);
    // ... and the runtime's host-to-device copy path: the FIRST hipMemcpyAsync from pinned memory of a process takes 7-10 ms
    // (measured inside the first batch's prepare step on a store filled by the GPU generators: the upload of the query block
    // was that process's first copy of the kind); a store loaded from host rows has paid it during the load
    static std::atomic<unsigned long long> copied{0};  // one bit per device ordinal (each GPU has copy queues of its own)
    const unsigned long long bit = 1ull << (device & 63);
    if (!(copied.fetch_or(bit) & bit)) {
        void* h = nullptr;
        void* d = nullptr;
        constexpr size_t WARM = (size_t)2 << 20;  // (large enough to take the path a batch's query block takes, not the small-copy one)
        if (hipHostMalloc(&h, WARM, hipHostMallocDefault) == hipSuccess && hipMalloc(&d, WARM) == hipSuccess) {
            memset(h, 0, WARM);
            if (hipMemcpyAsync(d, h, WARM, hipMemcpyHostToDevice, stream) == hipSuccess) (void)hipStreamSynchronize(stream);
        }
        if (d) (void)hipFree(d);
        if (h) (void)hipHostFree(h);
    }
    (void)hipGetLastError();
}

}  // namespace ott

// The L2 -> LDS fill path of the 256-query int8 tile (mfma_score_kernel<4, ., 5>, config 2 / config 4), measured in isolation.
//
// the tile stages 196 KB of rows (from HBM) AND 196 KB of queries (from the XCD's L2) per 256 rows, 15.4 GB per
// 10M x 768 batch, and round 5 ended on an empirical quotient (15.4 GB / 6.2 TB/s).  MI355X_MICROARCH.md ("Indexed rows: gather into
// LDS") gives 16.8-18.8 TB/s for rows shared out of an XCD's L2 against 6.0-6.1 from HBM: if the L2-sourced half of the fill really
// moves at 3x the HBM rate, 6.2 TB/s is not the roof.  This program issues the tile's own fill pattern — 512 threads, 8 waves, per
// stage and wave four pieces of 8 rows x 128 B by global_load_lds_dwordx4 (saddr + lane offset, as glds16u in ott_mfma.hip), a ring of
// NBUF stages with one counted wait and one barrier per stage — with the rows only, the queries only, or both, with one or two
// workgroups per CU, with and without the tile's matrix work beside it (32 v_mfma_i32_32x32x32_i8 per wave and stage on registers: 8 x 8 x 24 MFMAs per 256 x 256 x 768 tile over 8 waves and 6 stages).
//
//   hipcc -O3 --offload-arch=gfx950 fill_path.hip -o fill_path ;  ./fill_path [config-name]      (all configs without an argument)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x)                                                            \
    do {                                                                 \
        hipError_t e = (x);                                              \
        if (e != hipSuccess) {                                           \
            printf("%s: %s\n", #x, hipGetErrorString(e));                \
            exit(1);                                                     \
        }                                                                \
    } while (0)

typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ void glds16u(const char* sbase, uint32_t voff, uint32_t lds_addr) {
    uint32_t keep;
    if constexpr (NT) {
        asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
    } else {
        asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
    }
}
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
    const unsigned long long b = (unsigned long long)p;
    const unsigned long long lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
    const unsigned long long hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32));
    return (const char*)((hi << 32) | lo);
}

struct Args {
    const char* plane;   // [n_rows][pitch] int8 rows (streamed: HBM) — or a small region revisited (rows_span): L2
    const char* qblock;  // [256][pitch] the batch's int8 operands: every workgroup reads all of it for every tile (L2)
    uint32_t pitch;      // bytes per row (768)
    uint32_t n_tiles;    // tiles of RPS rows
    uint32_t span_tiles; // tile index is taken modulo this (n_tiles: a true stream; small: the rows come from L2 as well)
    int* out;
};

// RPS / QPS: rows / queries per stage (0 = that stream is off); NBUF: ring depth; NT: non-temporal row pieces; MF: v_mfma per wave and stage
// ORDER: 0 = a stage's row pieces first, then its query pieces (the tile's order until round 6); 1 = queries first; 2 = interleaved q, r, q, r ..
// GAP: shader cycles every wave idles at each tile boundary (the tile's epilogue: no fill is issued during it); PRE2: the next tile's
// SECOND stage is issued before that gap as well (one more barrier per tile), so two stages are in flight across it instead of one
template <int RPS, int QPS, int NBUF, bool NT, int MF, int ORDER = 0, int GAP = 0, bool PRE2 = false>
__global__ __launch_bounds__(512) void fill_kernel(Args a) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    constexpr int STAGE = (RPS + QPS) * 128;
    constexpr int PR = RPS / 64, PQ = QPS / 64;  // pieces per wave and stage
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane >> 3, lslot = lane & 7;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    const uint32_t kst = a.pitch / 128;  // K stages per tile
    const uint32_t voff = lrow * a.pitch + lslot * 16;
    uint32_t my_tiles = 0;
    for (uint32_t t = blockIdx.x; t < a.n_tiles; t += gridDim.x) my_tiles++;
    const uint32_t total = my_tiles * kst;
    i32x16 acc[4];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 16; j++) acc[i][j] = 0;
    const i32x4 fa = {tid, tid + 1, tid + 2, tid + 3}, fb = {lane, 1, 2, 3};
    int sink = 0;
    auto issue = [&](uint32_t it) {
        const uint32_t tile = (blockIdx.x + (it / kst) * gridDim.x) % a.span_tiles, s = it % kst;
        const int buf = (int)(it % NBUF);
        const char* rbase = PR > 0 ? uniform_ptr(a.plane + ((uint64_t)tile * RPS + (uint64_t)wave * (RPS / 8)) * a.pitch + s * 128) : nullptr;  // (row-major plane)
        const char* qbase = PQ > 0 ? uniform_ptr(a.qblock + (uint64_t)wave * (QPS / 8) * a.pitch + s * 128) : nullptr;
        if constexpr (ORDER == 4) {  // TILE-MAJOR plane: stage s of tile t is the contiguous block [(t * kst + s) * RPS * 128, + RPS * 128)
            rbase = PR > 0 ? uniform_ptr(a.plane + ((uint64_t)tile * kst + s) * (RPS * 128) + (uint64_t)wave * (RPS / 8) * 128) : nullptr;
        }
        auto row_piece = [&](int m) {
            if constexpr (ORDER == 4) glds16u<NT>(rbase, lane * 16 + m * 1024, lds_base + buf * STAGE + (wave * (RPS / 8) + 8 * m) * 128);
            else glds16u<NT>(rbase, voff + 8 * m * a.pitch, lds_base + buf * STAGE + (wave * (RPS / 8) + 8 * m) * 128);
        };
        auto qry_piece = [&](int m) { glds16u<false>(qbase, voff + 8 * m * a.pitch, lds_base + buf * STAGE + RPS * 128 + (wave * (QPS / 8) + 8 * m) * 128); };
        if constexpr (ORDER == 3 && PR == PQ && PR > 0) {
            // wave-specialised: waves 0-3 fetch ALL the stage's row pieces (2 PR each), waves 4-7 all its query pieces — no wave's L2
            // hits queue behind its own HBM misses
            const int w4 = wave & 3;
            if (wave < 4) {
                const char* b2 = uniform_ptr(a.plane + ((uint64_t)tile * RPS + (uint64_t)w4 * (RPS / 4)) * a.pitch + s * 128);
#pragma unroll
                for (int m = 0; m < 2 * PR; m++) glds16u<NT>(b2, voff + 8 * m * a.pitch, lds_base + buf * STAGE + (w4 * (RPS / 4) + 8 * m) * 128);
            } else {
                const char* b2 = uniform_ptr(a.qblock + (uint64_t)w4 * (QPS / 4) * a.pitch + s * 128);
#pragma unroll
                for (int m = 0; m < 2 * PQ; m++) glds16u<false>(b2, voff + 8 * m * a.pitch, lds_base + buf * STAGE + RPS * 128 + (w4 * (QPS / 4) + 8 * m) * 128);
            }
        } else if constexpr (ORDER == 2 && PR == PQ) {
#pragma unroll
            for (int m = 0; m < PR; m++) {
                qry_piece(m);
                row_piece(m);
            }
        } else {
            if constexpr (ORDER == 1) {
#pragma unroll
                for (int m = 0; m < PQ; m++) qry_piece(m);
            }
#pragma unroll
            for (int m = 0; m < PR; m++) row_piece(m);
            if constexpr (ORDER != 1) {
#pragma unroll
                for (int m = 0; m < PQ; m++) qry_piece(m);
            }
        }
    };
    // prologue: NBUF - 1 stages in flight
    for (uint32_t it = 0; it < (uint32_t)(NBUF - 1) && it < total; it++) issue(it);
    bool pre_issued = false;  // PRE2: stage it + 1 went out before the gap already
    for (uint32_t it = 0; it < total; it++) {
        // stage `it` has landed when at most the pieces of the NBUF - 2 stages issued after it are still out
        constexpr int LEFT = (NBUF - 2) * (PR + PQ);
        if (PRE2 && pre_issued) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PR + PQ) : "memory");  // (stage it + 1 may still be out)
        else if (it + NBUF - 1 <= total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LEFT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // every wave's pieces of stage `it` are in LDS; every wave has left stage it - 1 (its buffer is free)
        if (!(PRE2 && pre_issued) && it + NBUF - 1 < total) issue(it + NBUF - 1);
        pre_issued = false;
        // "consume" the stage: one LDS word per lane, and MF matrix instructions per wave
        sink ^= *reinterpret_cast<volatile int*>(smem + (it % NBUF) * STAGE + tid * 4);
#pragma unroll
        for (int i = 0; i < MF; i++) acc[i & 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, acc[i & 3], 0, 0, 0);
        if (GAP > 0 && it % kst == kst - 1) {  // the tile boundary
            if (PRE2 && NBUF == 2 && it + 2 < total) {
                __syncthreads();  // every wave has left stage `it`: its buffer takes the next tile's stage 1
                issue(it + 2);
                pre_issued = true;
            }
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // (the constant 100-MHz counter)
            while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)GAP) __builtin_amdgcn_s_sleep(4);
        }
    }
    int r = sink;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 16; j++) r ^= acc[i][j];
    if (r == 0x7fffffff) a.out[0] = r;
}

struct Config {
    const char* name;
    const char* what;
    void (*kern)(Args);
    int rps, qps, nbuf, wg_per_cu;
    bool rows_from_l2;
};

#define CFG(name, what, RPS, QPS, NBUF, NT, MF, WG, L2) Config{name, what, fill_kernel<RPS, QPS, NBUF, NT, MF>, RPS, QPS, NBUF, WG, L2}
#define CFGO(name, what, RPS, QPS, NBUF, NT, MF, WG, L2, ORD) Config{name, what, fill_kernel<RPS, QPS, NBUF, NT, MF, ORD>, RPS, QPS, NBUF, WG, L2}
#define CFGG(name, what, MF, GAPC, P2) Config{name, what, fill_kernel<256, 256, 2, false, MF, 0, GAPC, P2>, 256, 256, 2, 1, false}

int main(int argc, char** argv) {
    const uint32_t pitch = 768;
    const uint64_t n_rows = 10'000'000 / 256 * 256;
    int n_cu = 0;
    CK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0));
    char *plane = nullptr, *qblock = nullptr;
    int* out = nullptr;
    CK(hipMalloc((void**)&plane, n_rows * pitch));
    CK(hipMalloc((void**)&qblock, 256 * pitch));
    CK(hipMalloc((void**)&out, 64));
    CK(hipMemset(plane, 1, n_rows * pitch));
    CK(hipMemset(qblock, 2, 256 * pitch));
    const std::vector<Config> cfgs = {
        // (i) the rows alone, streamed from HBM
        CFG("rows_hbm", "rows only, HBM stream, the tile's ring (2 x 32 KB)", 256, 0, 2, false, 0, 1, false),
        CFG("rows_hbm_nt", "rows only, HBM stream, non-temporal", 256, 0, 2, true, 0, 1, false),
        CFG("rows_hbm_n3", "rows only, HBM stream, ring of 3", 256, 0, 3, false, 0, 1, false),
        CFG("rows_hbm_n4", "rows only, HBM stream, ring of 4", 256, 0, 4, false, 0, 1, false),
        CFG("rows_hbm_2wg", "rows only, HBM stream, TWO workgroups per CU (2 x 32 KB each)", 256, 0, 2, false, 0, 2, false),
        // (ii) the 196 KB query block alone, from L2
        CFG("queries_l2", "queries only (196 KB block, L2), the tile's ring", 0, 256, 2, false, 0, 1, false),
        CFG("queries_l2_n4", "queries only, ring of 4", 0, 256, 4, false, 0, 1, false),
        CFG("queries_l2_2wg", "queries only, TWO workgroups per CU", 0, 256, 2, false, 0, 2, false),
        // the ROW pattern served from L2 (each workgroup revisits a 2-tile region: 393 KB per workgroup slot, 12.6 MB per XCD would
        // not fit -> 8 tiles shared by everyone: 1.5 MB)
        CFG("rows_l2", "rows only, an 8-tile region every workgroup revisits (L2)", 256, 0, 2, false, 0, 1, true),
        // (iii) both, as the tile issues them
        CFG("both", "rows (HBM) + queries (L2): the tile's fill, ring of 2 x 64 KB", 256, 256, 2, false, 0, 1, false),
        CFG("both_nt", "rows (HBM, non-temporal) + queries (L2)", 256, 256, 2, true, 0, 1, false),
        CFG("both_2wg", "both, TWO workgroups per CU on 128 x 128 stages (2 x 32 KB each)", 128, 128, 2, false, 0, 2, false),
        CFG("both_l2", "rows from the 8-tile L2 region + queries (L2): the whole fill from L2", 256, 256, 2, false, 0, 1, true),
        // the ORDER in which a stage's pieces are issued (the vector memory pipe returns a CU's loads in order: an L2 hit queued
        // behind HBM misses waits for them)
        CFGO("both_qfirst", "both, a stage's QUERY pieces issued before its row pieces", 256, 256, 2, false, 0, 1, false, 1),
        CFGO("both_mixed", "both, query and row pieces interleaved", 256, 256, 2, false, 0, 1, false, 2),
        CFGO("both_qfirst_nt", "queries first, rows non-temporal", 256, 256, 2, true, 0, 1, false, 1),
        // a TILE-MAJOR plane: every stage of a tile is one contiguous 32-KB block (the plane is a derived structure: its layout is ours)
        CFGO("rows_tm", "rows only, tile-major plane (contiguous 32-KB stages), HBM", 256, 0, 2, false, 0, 1, false, 4),
        CFGO("rows_tm_nt", "rows only, tile-major, non-temporal", 256, 0, 2, true, 0, 1, false, 4),
        CFGO("both_tm", "tile-major rows (HBM) + queries (L2)", 256, 256, 2, false, 0, 1, false, 4),
        CFGO("both_tm_nt", "tile-major rows (HBM, non-temporal) + queries (L2)", 256, 256, 2, true, 0, 1, false, 4),
        CFGO("both_tm_mfma", "tile-major rows + queries + the matrix work", 256, 256, 2, false, 32, 1, false, 4),
        CFGO("both_wavesplit", "both, waves 0-3 fetch the rows, waves 4-7 the queries", 256, 256, 2, false, 0, 1, false, 3),
        CFGO("both_ws_mfma", "wave-split + the matrix work", 256, 256, 2, false, 32, 1, false, 3),
        CFGO("both_2wg_qf", "queries first, TWO workgroups per CU on 128 x 128 stages", 128, 128, 2, false, 0, 2, false, 1),
        // ... with the tile's matrix work beside the fill (32 MFMAs per wave and stage = 12.3k cycles per SIMD and tile)
        CFG("both_mfma", "both + 32 v_mfma_i32_32x32x32_i8 per wave and stage", 256, 256, 2, false, 32, 1, false),
        CFGO("both_qf_mfma", "queries first + the same matrix work", 256, 256, 2, false, 32, 1, false, 1),
        CFG("rows_hbm_mfma", "rows only (HBM) + the same matrix work", 256, 0, 2, false, 32, 1, false),
        CFG("queries_l2_mfma", "queries only (L2) + the same matrix work", 0, 256, 2, false, 32, 1, false),
        CFG("both_l2_mfma", "whole fill from L2 + the same matrix work", 256, 256, 2, false, 32, 1, true),
        // the tile boundary: an epilogue of ~2.8 us (s_memrealtime: 100 MHz, 280 ticks) during which no fill is issued
        CFGG("both_gap", "both + a 2.8-us gap per tile, ONE stage of the next tile in flight across it (the kernel today)", 0, 280, false),
        CFGG("both_gap_pre2", "both + the gap, TWO stages in flight across it (one more barrier per tile)", 0, 280, true),
        CFGG("both_gap_mfma", "both + gap + the matrix work, one stage across the gap", 32, 280, false),
        CFGG("both_gap_pre2_mfma", "both + gap + the matrix work, two stages across the gap", 32, 280, true),
        CFG("mfma_only", "no fill at all: the matrix work alone (one LDS word per lane and stage)", 0, 0, 2, false, 32, 1, false),
    };
    printf("%-16s %3s %4s %9s %9s %9s %9s %9s  %s\n", "config", "wg", "ring", "ms", "rows_TB/s", "qry_TB/s", "fill_TB/s", "GB/s/CU", "what");
    for (const Config& c : cfgs) {
        if (argc > 1 && strcmp(argv[1], c.name) != 0) continue;
        Args a;
        a.plane = plane;
        a.qblock = qblock;
        a.pitch = pitch;
        const int tile_rows = c.rps ? c.rps : 256;
        a.n_tiles = (uint32_t)(n_rows / tile_rows);
        a.span_tiles = c.rows_from_l2 ? 8u : a.n_tiles;
        a.out = out;
        const size_t smem = (size_t)c.nbuf * (c.rps + c.qps) * 128 + 2048;
        CK(hipFuncSetAttribute((const void*)c.kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        const dim3 grid((unsigned)(n_cu * c.wg_per_cu)), block(512);
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        for (int w = 0; w < 2; w++) hipLaunchKernelGGL(c.kern, grid, block, smem, 0, a);
        CK(hipDeviceSynchronize());
        const int reps = 6;
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; r++) hipLaunchKernelGGL(c.kern, grid, block, smem, 0, a);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        const double row_b = c.rps ? (double)a.n_tiles * c.rps * pitch : 0.0, qry_b = c.qps ? (double)a.n_tiles * c.qps * pitch : 0.0;
        printf("%-16s %3d %4d %9.3f %9.2f %9.2f %9.2f %9.1f  %s\n", c.name, c.wg_per_cu, c.nbuf, ms, row_b / ms * 1e-9, qry_b / ms * 1e-9,
               (row_b + qry_b) / ms * 1e-9, (row_b + qry_b) / ms * 1e-6 / n_cu, c.what);
        fflush(stdout);
    }
    return 0;
}

// Read-stream ceiling of MI355X against launch shape: the same 30.72 GB sweep as stream_patterns.hip (non-temporal 16-B loads,
// a wave owning a 64-row tile = 192 KB at a time, 8 rows x 128 B per instruction), for grids of 256 .. 4096 workgroups of 4 waves
// and 8 or 16 instructions in flight per wave; plus a variant where consecutive tiles go to consecutive WAVES of one workgroup
// (neighbouring tiles on one CU) instead of round the grid.   hipcc -O3 --offload-arch=gfx950 stream_occupancy.hip -o stream_occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int INF, bool BLOCKED>
__global__ __launch_bounds__(256) void sweep(const float* __restrict__ rows, uint64_t n_rows, uint32_t ld, float* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t n_tiles = n_rows / 64;
    const int lrow = lane >> 3, lslot = lane & 7;
    float acc = 0.f;
    const uint64_t nw = (uint64_t)gridDim.x * 4;
    // BLOCKED: the grid's waves split the tiles into contiguous ranges (wave w: tiles [w * per, (w+1) * per))
    const uint64_t per = (n_tiles + nw - 1) / nw;
    const uint64_t gw = (uint64_t)blockIdx.x * 4 + wave;
    const uint64_t t_begin = BLOCKED ? gw * per : gw, t_end = BLOCKED ? ((gw + 1) * per < n_tiles ? (gw + 1) * per : n_tiles) : n_tiles;
    const uint64_t t_step = BLOCKED ? 1 : nw;
    for (uint64_t t = t_begin; t < t_end; t += t_step) {
        const float* base = rows + t * 64 * (uint64_t)ld;
        for (uint32_t c = 0; c < ld; c += 32 * (INF / 8)) {
            v4f r[INF];
#pragma unroll
            for (int m = 0; m < INF; m++)
                r[m] = __builtin_nontemporal_load((const v4f*)(base + (uint64_t)((m & 7) * 8 + lrow) * ld + c + (m >> 3) * 32 + lslot * 4));
#pragma unroll
            for (int m = 0; m < INF; m++) acc += r[m].x + r[m].y + r[m].z + r[m].w;
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main(int argc, char** argv) {  // [rows] [dim]: default the headline's 10M x 768; `1000000 128` = config 1's 516 MB
    const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 10000000; const uint32_t ld = argc > 2 ? (uint32_t)atoi(argv[2]) : 768;
    float *d, *o;
    CK(hipMalloc(&d, n * ld * 4)); CK(hipMalloc(&o, 4));
    CK(hipMemset(d, 0x3c, n * ld * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto run = [&](const char* name, void (*k)(const float*, uint64_t, uint32_t, float*), int grid) {
        float best = 1e9f;
        for (int it = 0; it < 6; it++) {
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, n, ld, o);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (it && ms < best) best = ms;
        }
        printf("%-28s grid %5d  %.3f ms  %.2f TB/s\n", name, grid, best, n * ld * 4.0 / best / 1e9);
    };
    for (int grid : {256, 512, 1024, 2048, 4096}) {
        run("8 in flight, round robin", sweep<8, false>, grid);
        run("16 in flight, round robin", sweep<16, false>, grid);
        run("8 in flight, blocked", sweep<8, true>, grid);
        run("16 in flight, blocked", sweep<16, true>, grid);
    }
    return 0;
}

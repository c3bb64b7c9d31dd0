// How fast can MI355X stream a 30.72 GB row-major f32 matrix (10M x 768) with non-temporal loads, depending on the shape of
// a wave's requests?  Patterns (per wave-instruction of 64 lanes x 16 B):
//   seq    : 1 KB contiguous (the copy-benchmark shape)
//   row128 : 8 rows x 128 B, rows 3 KB apart (what exact_kernel issues: one K stage of a 64-row tile = 8 such instructions)
//   row256 : 4 rows x 256 B
//   row512 : 2 rows x 512 B
// Each wave owns a 64-row tile (192 KB) at a time and sweeps it stage by stage, like the scorer; the loaded values are
// folded into a checksum so nothing is optimised away.   hipcc -O3 --offload-arch=gfx950 stream_patterns.hip -o stream_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int ROWB>  // bytes per row per stage: 128, 256, 512; 0 = sequential
__global__ __launch_bounds__(256) void sweep(const float* __restrict__ rows, uint64_t n_rows, uint32_t ld, float* out) {
    const int lane = threadIdx.x & 63;
    const uint32_t gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    const uint64_t n_tiles = n_rows / 64;
    float acc = 0.f;
    for (uint64_t t = gw; t < n_tiles; t += nw) {
        const float* base = rows + t * 64 * (uint64_t)ld;
        if (ROWB == 0) {
            for (uint32_t i = 0; i < 64u * ld / 256; i += 8) {  // 8 instructions in flight, 1 KB each, contiguous
                v4f r[8];
#pragma unroll
                for (int m = 0; m < 8; m++) r[m] = __builtin_nontemporal_load((const v4f*)(base + (uint64_t)(i + m) * 256 + lane * 4));
#pragma unroll
                for (int m = 0; m < 8; m++) acc += r[m].x + r[m].y + r[m].z + r[m].w;
            }
        } else {
            constexpr int LPR = ROWB > 0 ? ROWB / 16 : 8;       // lanes per row
            constexpr int RPI = 64 / LPR;        // rows per instruction
            constexpr int IPS = 64 / RPI;        // instructions per stage (64 rows)
            const int lrow = lane / LPR, lslot = lane % LPR;
            for (uint32_t c = 0; c < ld; c += (ROWB > 0 ? ROWB / 4 : 32)) {
                v4f r[IPS < 8 ? IPS : 8];
#pragma unroll
                for (int g = 0; g < IPS; g += 8) {
#pragma unroll
                    for (int m = 0; m < 8 && g + m < IPS; m++)
                        r[m] = __builtin_nontemporal_load((const v4f*)(base + (uint64_t)((g + m) * RPI + lrow) * ld + c + lslot * 4));
#pragma unroll
                    for (int m = 0; m < 8 && g + m < IPS; m++) acc += r[m].x + r[m].y + r[m].z + r[m].w;
                }
            }
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main() {
    const uint64_t n = 10000000; const uint32_t ld = 768;
    float *d, *o;
    CK(hipMalloc(&d, n * ld * 4)); CK(hipMalloc(&o, 4));
    CK(hipMemset(d, 0, n * ld * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto run = [&](const char* name, void (*k)(const float*, uint64_t, uint32_t, float*)) {
        float best = 1e9f;
        for (int it = 0; it < 6; it++) {
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, d, n, ld, o);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (it && ms < best) best = ms;
        }
        printf("%-7s %.3f ms  %.2f TB/s\n", name, best, n * ld * 4.0 / best / 1e9);
    };
    run("seq", sweep<0>); run("row128", sweep<128>); run("row256", sweep<256>); run("row512", sweep<512>);
    return 0;
}

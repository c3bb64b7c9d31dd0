// Sustained f32 MFMA rate of the MI355X box: 8 waves per CU (2 per SIMD), each issuing v_mfma_f32_32x32x2_f32 back to back
// on 4 independent accumulators from registers — no memory, no LDS.  What C2's 134 TFLOP/s should be read against
// (the 157.3 TFLOP/s figure assumes 2.4 GHz; the chip settles lower under this load).
//   hipcc -O3 --offload-arch=gfx950 mfma_ceiling.hip -o mfma_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NACC>
__global__ __launch_bounds__(512) void burn(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; i++)
        for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16 / NACC; u++) {
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < NACC; i++)
        for (int r = 0; r < 16; r++) s += acc[i][r];
    if (s == 12345.678f) out[0] = s;
}

int main() {
    float* o; CK(hipMalloc(&o, 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int iters = 200000;  // x16 MFMAs per wave
    void (*ks[4])(float*, int, float, float) = {burn<1>, burn<2>, burn<4>, burn<8>};
    for (int rep = 0; rep < 8; rep++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(ks[rep / 2], dim3(256), dim3(512), 0, 0, o, iters, 1.0f, 0.5f);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        const double flops = 256.0 * 8 * iters * 16.0 * (32 * 32 * 2 * 2);
        // one MFMA occupies a SIMD's matrix pipe for 64 cycles; 2 waves per SIMD -> cycles per SIMD = 2 * iters * 16 * 64
        printf("%d independent accumulators per wave: %.2f ms  %.1f TFLOP/s  (implied matrix-pipe clock %.2f GHz)\n", 1 << (rep / 2), ms, flops / ms / 1e9, 2.0 * iters * 16 * 64 / ms / 1e6);
    }
    return 0;
}

// mfma_accum_probe.hip — what the certification bounds of the batch path assume about the matrix unit, MEASURED.
//
// The candidate passes of ott_mfma.hip sum products on the matrix cores, in an order and at an internal precision AMD does not
// document; the certification prices that as a recursive-summation error,
//     |mfma_sum - exact_sum| <= c * dim * 2^-24 * sum_i |a_i b_i|     (c = 1.25 f32 pipe incl. the reference's own order,
//                                                                       2.5 the 16-bit hi pass, 3.75 split bf16),
// on top of the operands' own rounding.  This probe feeds the three instructions the kernels use —
// v_mfma_f32_32x32x2_f32, v_mfma_f32_32x32x16_bf16, v_mfma_f32_32x32x16_f16 — 32 x 32 outputs of K-term sums whose operands
// are EXACT in the instruction's input format (so the only error is the accumulation), accumulates over K the way the kernels
// do (one accumulator chain per output, K / 2 or K / 16 instructions), and compares every output with the f64 sum:
//     ratio = |mfma - exact| / (2^-24 * sum |a_i b_i|)   in units of "one f32 rounding of the sum's magnitude".
// Same-sign operands (every product positive: errors cannot cancel), mixed signs, and operands spread over many binades.
// Prints one line per case; exit code 1 if any ratio exceeds K (the plain recursive-summation bound the model's constants sit
// above).  tests/test_gpu_mfma.py builds and runs it on the GPU box.
//   hipcc -O2 --offload-arch=gfx950 mfma_accum_probe.hip -o mfma_accum_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CK(x)                                                         \
    do {                                                              \
        hipError_t e = (x);                                           \
        if (e != hipSuccess) {                                        \
            printf("%s: %s\n", #x, hipGetErrorString(e));             \
            exit(2);                                                  \
        }                                                             \
    } while (0)

// A: [32][K] row-major, B: [32][K] row-major (column n of the product is row n of B), C: [32][32], as floats.
// MODE 0: f32 32x32x2 — lane (l31 = lane & 31, lh = lane >> 5) supplies a[l31][2j + lh], b[l31][2j + lh] at step j.
// MODE 1 / 2: bf16 / f16 32x32x16 — lane supplies k = 16 j + 8 lh .. + 7 of its row.
// Output layout (both): acc[r] of lane is C[row = (r & 3) + 8 (r >> 2) + 4 lh][col = l31].
template <int MODE>
__global__ __launch_bounds__(64) void probe(const float* A, const float* B, float* C, int K) {
    const int lane = threadIdx.x, l31 = lane & 31, lh = lane >> 5;
    f32x16 acc;
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
    if (MODE == 0) {
        for (int j = 0; j < K / 2; j++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[l31 * K + 2 * j + lh], B[l31 * K + 2 * j + lh], acc, 0, 0, 0);
    } else {
        for (int j = 0; j < K / 16; j++) {
            if (MODE == 1) {
                bf16x8 a, b;
                for (int e = 0; e < 8; e++) {
                    a[e] = (__bf16)A[l31 * K + 16 * j + 8 * lh + e];
                    b[e] = (__bf16)B[l31 * K + 16 * j + 8 * lh + e];
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
            } else {
                f16x8 a, b;
                for (int e = 0; e < 8; e++) {
                    a[e] = (_Float16)A[l31 * K + 16 * j + 8 * lh + e];
                    b[e] = (_Float16)B[l31 * K + 16 * j + 8 * lh + e];
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
            }
        }
    }
    for (int r = 0; r < 16; r++) C[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + l31] = acc[r];
}

static float to_fmt(float x, int mode) {  // round to the instruction's input format, on the host, so that the operands are exact in it
    if (mode == 0) return x;
    if (mode == 1) {
        unsigned u;
        std::memcpy(&u, &x, 4);
        u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;  // bf16 RNE (finite inputs only)
        float y;
        std::memcpy(&y, &u, 4);
        return y;
    }
    return (float)(_Float16)x;
}

int main() {
    const char* names[3] = {"v_mfma_f32_32x32x2_f32", "v_mfma_f32_32x32x16_bf16", "v_mfma_f32_32x32x16_f16"};
    const char* kinds[3] = {"same sign [0.5, 1)", "mixed signs [-1, 1)", "same sign, magnitudes over 12 binades"};
    int bad = 0;
    for (int K : {96, 768, 3072}) {
        for (int kind = 0; kind < 3; kind++) {
            for (int mode = 0; mode < 3; mode++) {
                std::mt19937 rng(1234 + K + 7 * kind);
                std::uniform_real_distribution<float> u01(0.0f, 1.0f);
                std::vector<float> A(32 * K), B(32 * K), C(32 * 32);
                for (auto* v : {&A, &B})
                    for (float& x : *v) {
                        float t = kind == 1 ? 2.0f * u01(rng) - 1.0f : 0.5f + 0.5f * u01(rng);
                        if (kind == 2) t = std::ldexp(t, (int)(12.0f * u01(rng)) - 6);
                        x = to_fmt(t, mode);
                    }
                float *dA, *dB, *dC;
                CK(hipMalloc(&dA, A.size() * 4));
                CK(hipMalloc(&dB, B.size() * 4));
                CK(hipMalloc(&dC, C.size() * 4));
                CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
                CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
                if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
                else if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
                else hipLaunchKernelGGL(probe<2>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
                double worst = 0.0;
                for (int m = 0; m < 32; m++)
                    for (int n = 0; n < 32; n++) {
                        double s = 0.0, mag = 0.0;
                        for (int k = 0; k < K; k++) {
                            const double p = (double)A[m * K + k] * (double)B[n * K + k];
                            s += p;
                            mag += std::fabs(p);
                        }
                        const double ratio = std::fabs((double)C[m * 32 + n] - s) / (5.9604644775390625e-8 * mag);
                        if (ratio > worst) worst = ratio;
                    }
                printf("K=%4d  %-26s %-40s max |mfma - exact| / (2^-24 sum|ab|) = %8.3f   (model allows >= %.0f)\n", K, names[mode], kinds[kind], worst,
                       (mode == 0 ? 1.25 : 2.5) * K);
                if (!(worst <= (double)K)) bad++;
                CK(hipFree(dA));
                CK(hipFree(dB));
                CK(hipFree(dC));
            }
        }
    }
    printf(bad ? "FAILED: %d case(s) above the recursive-summation bound\n" : "ALL WITHIN BOUND\n", bad);
    return bad ? 1 : 0;
}

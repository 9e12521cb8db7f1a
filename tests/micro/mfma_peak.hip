// Microbenchmark: fp32 MFMA issue rate with the head GEMM's grid (3072 blocks x 4 waves x 3072 MFMAs).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int OCC>
__global__ __launch_bounds__(256, OCC) void k(float *out, int iters, float a0, float b0) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[3], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 123.456f) out[0] = s;
}
int main() {
    float *out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 3072, iters = 3072 / 16;
    for (int rep = 0; rep < 3; ++rep) {
        for (int occ = 1; occ <= 2; ++occ) {
            float ms = 0.f;
            hipEventRecord(e0);
            for (int i = 0; i < 10; ++i) {
                if (occ == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 65536 * 0, 0, out, iters, 1.0f, 2.0f);
                else hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
            }
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            const double fl = 10.0 * blocks * 4 * (iters * 16.0) * 4096.0;
            printf("rep %d launch_bounds occ %d: %.1f us per launch, %.1f TFLOP/s\n", rep, occ, ms * 100.0, fl / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}

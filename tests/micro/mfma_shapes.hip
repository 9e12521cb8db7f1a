// Microbenchmark: sustained fp16 MFMA rate of the two tile shapes on pseudo-random operands held in registers
// (v_mfma_f32_32x32x16_f16 against v_mfma_f32_16x16x32_f16; 8 waves per CU, one workgroup per CU, several rounds).
// MI355X_MICROARCH.md reports the 16x16x32 loop at ~1.15 x the FLOP/s at equal cycles per FLOP (power-limited clock).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__device__ h8 rnd8(unsigned s) {
    h8 v;
    for (int e = 0; e < 8; ++e) { s = s * 1664525u + 1013904223u; v[e] = (_Float16)(((s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 2.0f); }
    return v;
}

template <int SHAPE>
__global__ __launch_bounds__(512, 1) void k(float *out, int iters) {
    h8 a[4], b[2];
    for (int i = 0; i < 4; ++i) a[i] = rnd8(threadIdx.x * 7 + i * 131 + blockIdx.x);
    for (int i = 0; i < 2; ++i) b[i] = rnd8(threadIdx.x * 13 + i * 977 + blockIdx.x * 3);
    float s = 0.f;
    if (SHAPE == 32) {
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i * 2 + j], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    } else {
        f32x4 acc[32];
        for (int i = 0; i < 32; ++i) for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u & 3], b[(u >> 2) & 1], acc[u], 0, 0, 0);
        }
        for (int i = 0; i < 32; ++i) for (int e = 0; e < 4; ++e) s += acc[i][e];
    }
    if (s == 123.456f) out[0] = s;
}

int main() {
    float *out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 3;
    for (int rep = 0; rep < 3; ++rep) {
        for (int shape = 0; shape < 2; ++shape) {
            const int iters = shape == 0 ? 4096 : 2048;      // same FLOPs: 8 x 32768 vs 32 x 16384 per iteration
            float ms = 0.f;
            hipEventRecord(e0);
            if (shape == 0) hipLaunchKernelGGL(k<32>, dim3(blocks), dim3(512), 0, 0, out, iters);
            else hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(512), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            const double fl = (double)blocks * 8 * iters * (shape == 0 ? 8 * 32768.0 : 32 * 16384.0);
            printf("rep %d %s: %.1f us, %.1f TFLOP/s\n", rep, shape == 0 ? "32x32x16" : "16x16x32", ms * 1e3, fl / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}

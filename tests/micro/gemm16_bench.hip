// Microbenchmark of the split-arithmetic head GEMM (diffusionnag_amd/csrc/dnag_gemm16.hip.h) at the shape of
// final.linears.1 for B = 4096: M = 32768, N = K = 1536.  Build with -DSPLIT_DBG=<mask> for timing ablations.  Also times a bare v_mfma_f32_32x32x16_f16 loop (the rate the chip sustains).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../diffusionnag_amd/csrc/dnag_gemm16.hip.h"

__global__ __launch_bounds__(512, 1) void mfma_only(float *out, int iters) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    h8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (threadIdx.x + e)); b[e] = (_Float16)(0.002f * (threadIdx.x * 3 + e)); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = mfma16(a, b, acc[u]);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 123.456f) out[0] = s;
}

int main() {
#ifndef BENCH_K
#define BENCH_K 1536
#endif
    const int M = 32768, N = 1536, K = BENCH_K;
    float *A, *W, *bias, *C;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&W, (size_t)N * K * 4); hipMalloc(&bias, N * 4); hipMalloc(&C, (size_t)M * N * 4);
    {   // random split-pair operands
        std::vector<float> w((size_t)N * K), ws((size_t)N * K);
        unsigned s = 1;
        for (auto &v : w) { s = s * 1664525u + 1013904223u; v = ((s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.05f; }
        split_pack(w.data(), N, K, ws.data());
        hipMemcpy(W, ws.data(), ws.size() * 4, hipMemcpyHostToDevice);
        std::vector<float> a((size_t)4096 * K), as((size_t)4096 * K);
        for (auto &v : a) { s = s * 1664525u + 1013904223u; v = ((s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 4.0f; }
        split_pack(a.data(), 4096, K, as.data());
        for (int i = 0; i < M / 4096; ++i) hipMemcpy(A + (size_t)i * 4096 * K, as.data(), as.size() * 4, hipMemcpyHostToDevice);
        hipMemset(bias, 0, N * 4);
    }
    typedef SplitGeom Ge;
#ifdef BENCH_OUTSPLIT
    auto kern = gemm_split_kernel<ACT_ELU, true, 0>;      // final.linears.0: split-pair output rows
#else
    auto kern = gemm_split_kernel<ACT_ELU, false, 1>;
#endif
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Ge::lds_bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = (M / Ge::BM) * (N / Ge::BN);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(Ge::T), Ge::lds_bytes, 0, A, W, bias, C, M, N, K, (const float *)nullptr, (float *)nullptr, SplitF32Args{});
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("gemm_split 256x256 DBG=%d: %.1f us per launch, %.1f TFLOP/s (algorithmic), %.1f TFLOP/s of fp16 MFMA issued\n", SPLIT_DBG,
               ms / 20 * 1e3, 2.0 * M * N * K / (ms / 20 * 1e-3) / 1e12, 6.0 * M * N * K / (ms / 20 * 1e-3) / 1e12);
    }
    for (int rep = 0; rep < 2; ++rep) {
        const int iters = 4096;
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_only, dim3(256 * 4), dim3(512), 0, 0, C, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fl = 256.0 * 4 * 8 * iters * 8 * (2.0 * 32 * 32 * 16);
        printf("bare f16 MFMA loop: %.1f us, %.1f TFLOP/s\n", ms * 1e3, fl / (ms * 1e-3) / 1e12);
    }
    return 0;
}

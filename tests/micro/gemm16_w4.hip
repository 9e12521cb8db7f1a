// Experiment: the split-arithmetic head GEMM with FOUR waves of 128 x 128 per 256 x 256 block tile (one wave per SIMD,
// 16 accumulator tiles per wave) instead of eight waves of 128 x 64: 16 fragment reads per 48 MFMAs instead of 12 per
// 24, i.e. a third less LDS read traffic per MFMA.  Plain fp32 epilogue; timing only (tests/micro/gemm16_bench.hip is
// the production kernel's benchmark).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../diffusionnag_amd/csrc/dnag_gemm16.hip.h"

__global__ __launch_bounds__(256, 1) void gemm4w_kernel(const float *__restrict__ A, const float *__restrict__ W,
                                                        const float *__restrict__ bias, float *__restrict__ C, int M, int Nout, int K) {
    constexpr int BM = 256, BN = 256, RPP = 32, NPA = BM / RPP, NPB = BN / RPP;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As0 = smem;
    float *Bs0 = smem + 2 * BM * GEMM_LS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int nbn = Nout / BN;
    const int nwg = gridDim.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7;
    const int wg = ((xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    const int tm = wg / nbn, tn = wg - tm * nbn;
    const int lrow = tid >> 3, lk = (tid & 7) * 4;
    const int KT = K / GEMM_BK;
    const float *Ag = A + (size_t)(tm * BM + lrow) * K + lk;
    const float *Wg = W + (size_t)(tn * BN + lrow) * K + lk;

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    v4f ra[NPA], rb[NPB];
    auto gload = [&](int kt) {
#pragma unroll
        for (int p = 0; p < NPA; ++p) ra[p] = *reinterpret_cast<const v4f *>(Ag + (size_t)(RPP * p) * K + kt * GEMM_BK);
#pragma unroll
        for (int p = 0; p < NPB; ++p) rb[p] = *reinterpret_cast<const v4f *>(Wg + (size_t)(RPP * p) * K + kt * GEMM_BK);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int p = 0; p < NPA; ++p) *reinterpret_cast<v4f *>(As0 + (buf * BM + lrow + RPP * p) * GEMM_LS + lk) = ra[p];
#pragma unroll
        for (int p = 0; p < NPB; ++p) *reinterpret_cast<v4f *>(Bs0 + (buf * BN + lrow + RPP * p) * GEMM_LS + lk) = rb[p];
    };
    gload(0);
    lstore(0);
    if (KT > 1) gload(1);
    __syncthreads();

    const float *as_base = As0 + (wm * 128 + r) * GEMM_LS + 8 * h;
    const float *bs_base = Bs0 + (wn * 128 + r) * GEMM_LS + 8 * h;
    h8 ah[4], al[4], bh[2][4], bl[2][4];
    auto read_a = [&](int i, int buf, int j) {
        const float *p = as_base + (buf * BM + i * 32) * GEMM_LS + 16 * j;
        ah[i] = *reinterpret_cast<const h8 *>(p);
        al[i] = *reinterpret_cast<const h8 *>(p + 4);
    };
    auto read_b = [&](auto set, int t, int buf, int j) {
        constexpr int S = decltype(set)::value;
        const float *p = bs_base + (buf * BN + t * 32) * GEMM_LS + 16 * j;
        bh[S][t] = *reinterpret_cast<const h8 *>(p);
        bl[S][t] = *reinterpret_cast<const h8 *>(p + 4);
    };
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
    auto mfma_step = [&](auto cs, auto ns, bool more, int nbuf, int nj) {
        constexpr int CS = decltype(cs)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[i][t] = mfma16(ah[i], bh[CS][t], acc[i][t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[i][t] = mfma16(ah[i], bl[CS][t], acc[i][t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[i][t] = mfma16(al[i], bh[CS][t], acc[i][t]);
            if (more) {
                read_b(ns, i, nbuf, nj);        // one B tile of the next step per A row: spread over the step
                read_a(i, nbuf, nj);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
#pragma unroll
    for (int t = 0; t < 4; ++t) read_b(S0{}, t, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) read_a(i, 0, 0);

    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < KT) lstore(cur ^ 1);
        if (kt + 2 < KT) gload(kt + 2);
        mfma_step(S0{}, S1{}, true, cur, 1);
        __syncthreads();
        mfma_step(S1{}, S0{}, kt + 1 < KT, cur ^ 1, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = tn * BN + wn * 128 + j * 32 + r;
        const float bv = bias[col];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row0 = tm * BM + wm * 128 + i * 32;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = fmaf(acc[i][j][e], SPLIT_OUT_INV, bv);
                v = v > 0.f ? v : __expf(v) - 1.0f;
                C[(size_t)(row0 + acc_row(e, h)) * Nout + col] = v;
            }
        }
    }
}

int main() {
    const int M = 32768, N = 1536, K = 1536;
    float *A, *W, *bias, *C, *C2;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&W, (size_t)N * K * 4); hipMalloc(&bias, N * 4);
    hipMalloc(&C, (size_t)M * N * 4); hipMalloc(&C2, (size_t)4096 * N * 4);
    {
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
    auto kern = gemm4w_kernel;
    auto kref = gemm_split_kernel<ACT_ELU, false, 1>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Ge::lds_bytes);
    hipFuncSetAttribute(reinterpret_cast<const void *>(kref), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Ge::lds_bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = (M / 256) * (N / 256);
    // correctness against the production kernel on the first 4096 rows
    hipLaunchKernelGGL(kref, dim3((4096 / 256) * (N / 256)), dim3(Ge::T), Ge::lds_bytes, 0, A, W, bias, C2, 4096, N, K, (const float *)nullptr, (float *)nullptr, SplitF32Args{});
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), Ge::lds_bytes, 0, A, W, bias, C, M, N, K);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    {
        std::vector<float> c1((size_t)4096 * N), c2((size_t)4096 * N);
        hipMemcpy(c1.data(), C, c1.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(c2.data(), C2, c2.size() * 4, hipMemcpyDeviceToHost);
        double md = 0, mx = 0;
        for (size_t i = 0; i < c1.size(); ++i) { md = fmax(md, fabs((double)c1[i] - c2[i])); mx = fmax(mx, fabs((double)c2[i])); }
        printf("4-wave vs production kernel: max |diff| %.3g of max %.3g\n", md, mx);
    }
    for (int which = 0; which < 2; ++which)
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            for (int i = 0; i < 20; ++i) {
                if (which == 0) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), Ge::lds_bytes, 0, A, W, bias, C, M, N, K);
                else hipLaunchKernelGGL(kref, dim3(grid), dim3(Ge::T), Ge::lds_bytes, 0, A, W, bias, C, M, N, K, (const float *)nullptr, (float *)nullptr, SplitF32Args{});
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%s: %.1f us per launch, %.1f TFLOP/s (algorithmic)\n", which == 0 ? "4 waves of 128x128" : "8 waves of 128x64 (production)",
                   ms / 20 * 1e3, 2.0 * M * N * K / (ms / 20 * 1e-3) / 1e12);
        }
    return 0;
}

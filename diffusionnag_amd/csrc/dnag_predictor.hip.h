// MetaNeuralPredictor surrogate: forward and analytic input gradient, one workgroup per G = 8 graphs
// (64 rows), 10 waves, all activations in LDS.  Included by dnag_lib.hip.
//
// Restates (reference NAS-Bench-201/): models/digcn_meta.py:109-129 (arch_encode), :39-48
// (DirectedGraphConvolution), :12-22 (normalize_adj, graph_pooling), :141-159 (predict/forward) and
// the autograd call of models/utils.py:130-145 (regress=True) as a hand-derived backward pass.
//
//   Ahat = rn(M + I);  A1 = rn(Ahat);  A2 = rn(Ahat^T)                (rn: divide rows by their sums)
//   H_{l+1} = 1/2 [relu(A1 (H_l W1_l)) + relu(A2 (H_l W2_l))]         evaluated as (A1 H_l) W1_l
//   p = sum_n H_L[n] / N + tau ;  a = Wfc1 p ;  y = w2 . tanh(W0 [D_mu | a] + b0) + b2
// Backward (dy/dx only): dpre = w2 (1 - tanh^2); da = W0[:, nz:]^T dpre; dp = Wfc1^T da;
//   dH_L[n] = dp / N;  per layer  dH_l = (A1^T (1/2 dH_{l+1} . [Z1>0])) W1^T + (A2^T (...[Z2>0])) W2^T.
//
// Every product with K >= 128 runs on v_mfma_f32_32x32x2_f32 (column tiles of 32: 144 = 4.5 tiles,
// one (column tile, row tile) job per wave); the 8x8 adjacency mixing, the K = 7 first layer and the
// hs-wide tanh head are plain FMAs.
#pragma once
#include "dnag_common.h"
#include "dnag_encoder.hip.h"

#ifndef DNAG_PRED_SKIP
#define DNAG_PRED_SKIP 0    // debug builds only (-DDNAG_PRED_SKIP=<mask>): phases switched off for timing ablations (wrong results)
#endif

#define PRED_GH 144        // gcn_hidden (configs/tr_meta_surrogate.py:97)
#define PRED_HS 148        // LDS row stride of [R][144] buffers (148 % 64 == 20: conflict-free b128 reads)
#define PRED_NT 5          // ceil(144 / 32) column tiles
#define PRED_G 8           // graphs per workgroup
#define PRED_NW 10         // waves per workgroup: one (column tile, row tile) job each
#define PRED_MAXHS 512

template <int N, int V>
struct PredGeom {
    static constexpr int G = PRED_G;
    static constexpr int R = G * N;
    static constexpr int MT = R / 32;
    static_assert(R % 32 == 0 && PRED_NT * MT == PRED_NW, "one MFMA job per wave");
    static constexpr int XS = 8;                                      // row stride of the staged x / A x buffers
    // dynamic LDS bytes for `hs` head units and `L` DiGCN layers
    static size_t lds_bytes(int hs, int L) {
        const size_t hs4 = align4((size_t)hs);
        const size_t fl = 3 * (size_t)R * PRED_HS      // H, U1, U2
                          + 2 * (size_t)G * N * N      // A1, A2
                          + (size_t)(G + 1) * hs4      // tanh / dpre, c0
                          + 3 * (size_t)R * XS         // x, A1 x, A2 x
                          + (size_t)L * 2 * R * PRED_NT;   // relu bit masks (uint32)
        return fl * sizeof(float);
    }
};

// x [batch][N][V]; temb [tstride ? batch : 1][144]; mask [mstride ? batch : 1][N][N]; d_mu [nz];
// y [batch] or null; grad [batch][N][V] or null.
// Two 32x32 output tiles of one wave, back to back: acc1 = A1 B1^T with B1 already in registers, acc2 = A2 B2^T with
// B2's fragments fetched chunk by chunk while the first tile's MFMAs run (into the registers B1's chunks leave).  The A rows sit in LDS as split-pair rows (dnag_gemm16.hip.h: per 8 values 16 B of fp16 hi, 16 B of lo),
// written once by the mixing phase with the phase's power-of-two scale `sa`, so a step is two ds_read_b128 and three
// MFMAs, no conversion work; a1p / a2p point at the lane's row and K/2 half.  F2 is a frag16_pack matrix
// (dnag_common.h), nt its column tile; inv1 / inv2 = 1 / (sa sw) of the two products (sw: PredLayout::wsc).
template <int K>
__device__ __forceinline__ void two_tiles16(const float *a1p, const float *a2p, const float (&b1)[K / 2], const float *F2,
                                            int nt, float inv1, float inv2, f32x16 &acc1, f32x16 &acc2) {
    typedef const __attribute__((address_space(1))) v4f *gv4p;
    static_assert(K % 16 == 0, "8 k values per lane half and step");
    const float *p2 = F2 + ((size_t)nt * (K / 8) * 64 + (threadIdx.x & 63)) * 4;
    float b2[K / 2];
    auto step = [&](const float *ap, const float (&b)[K / 2], int s, f32x16 &acc) {
        const h8 ahi = *reinterpret_cast<const h8 *>(ap + 8 * s);
        const h8 alo = *reinterpret_cast<const h8 *>(ap + 8 * s + 4);
        const h8 bhi = __builtin_bit_cast(h8, v4f{b[8 * s], b[8 * s + 1], b[8 * s + 2], b[8 * s + 3]});
        const h8 blo = __builtin_bit_cast(h8, v4f{b[8 * s + 4], b[8 * s + 5], b[8 * s + 6], b[8 * s + 7]});
        acc = mfma16(ahi, bhi, acc);
        acc = mfma16(ahi, blo, acc);
        acc = mfma16(alo, bhi, acc);
    };
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc1[i] = 0.f; acc2[i] = 0.f; }
#pragma unroll
    for (int s = 0; s < K / 16; ++s) {
        step(a1p, b1, s, acc1);
        asm volatile("" ::: "memory");     // chunk s of B2 is fetched AFTER chunk s of B1 is consumed: it reuses those registers
        const v4f u0 = *(gv4p)(p2 + (size_t)(2 * s) * 256), u1 = *(gv4p)(p2 + (size_t)(2 * s + 1) * 256);
        b2[8 * s + 0] = u0.x; b2[8 * s + 1] = u0.y; b2[8 * s + 2] = u0.z; b2[8 * s + 3] = u0.w;
        b2[8 * s + 4] = u1.x; b2[8 * s + 5] = u1.y; b2[8 * s + 6] = u1.z; b2[8 * s + 7] = u1.w;
    }
#pragma unroll
    for (int s = 0; s < K / 16; ++s) step(a2p, b2, s, acc2);
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc1[i] *= inv1; acc2[i] *= inv2; }
}

template <int N, int V>
__global__ __launch_bounds__(PRED_NW * 64) void predictor_kernel(const float *__restrict__ P, const PredLayout lay,
                                                                 const float *__restrict__ x, const float *__restrict__ temb,
                                                                 int tstride, const float *__restrict__ mask, int mstride,
                                                                 const float *__restrict__ d_mu, float *__restrict__ y,
                                                                 float *__restrict__ grad, int batch) {
    typedef PredGeom<N, V> Ge;
    static_assert(N == 8, "adjacency mixing is written for 8-node graphs");
    constexpr int G = Ge::G, R = Ge::R, MT = Ge::MT, HS = PRED_HS, GH = PRED_GH, NT = PRED_NW * 64, LH = 128, XS = Ge::XS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Hb = smem;                       // [R][HS]
    float *U1 = Hb + R * HS;                // [R][HS]
    float *U2 = U1 + R * HS;                // [R][HS]
    float *A1 = U2 + R * HS;                // [G][N][N]
    float *A2 = A1 + G * N * N;             // [G][N][N]
    const int L = lay.L, hs = lay.hs, nz = lay.nz;
    const int TS = (hs + 3) & ~3;           // row stride of tv
    float *tv = A2 + G * N * N;             // [G][TS]
    float *c0 = tv + G * TS;                // [TS]
    float *xin = c0 + TS;                   // [R][XS]
    float *ux1 = xin + R * XS;              // [R][XS] A1 x
    float *ux2 = ux1 + R * XS;              // [R][XS] A2 x
    // relu masks [L][MT][2][PRED_NT * 32]: one word per (row tile, lane half, column) in the MFMA accumulator layout --
    // bit i = [Z1 > 0], bit 16 + i = [Z2 > 0] at row acc_row(i, half) of the tile: a lane packs its 16 registers itself
    uint32_t *bits = reinterpret_cast<uint32_t *>(ux2 + R * XS);
    constexpr int BW = PRED_NT * 32, BL = MT * 2 * BW;   // words per (tile, half), per layer
    static_assert(BL == 2 * R * PRED_NT, "PredGeom::lds_bytes sizes the masks as L * 2 * R * PRED_NT words");
    float *Ah = U1;                         // scratch for Ahat during set-up

    const int tid = threadIdx.x;
    const int g0 = blockIdx.x * G;
#ifdef DNAG_PRED_STAMPS
    long long ts_[64];
    int nts_ = 0;
#define PRED_STAMP(k) do { if (nts_ < 64) ts_[nts_++] = clock64(); } while (0)
    PRED_STAMP(0);
#else
#define PRED_STAMP(k)
#endif

    // Split-arithmetic MFMAs (dnag_gemm16.hip.h): every GEMM phase reads fp16 hi/lo weight fragments (P + delta16) and
    // splits its A operand with a power-of-two scale derived from the operand's largest magnitude in this workgroup
    // (backward operands are gradients of unknown size): amaxv[k] holds max |value| of phase k's operand (head phases) or a
    // bound on it, the maximum of the mixing's input (layer phases; see mix_fwd).
    __shared__ unsigned int amaxv[24];
    __shared__ unsigned int colsum_max;       // largest column sum of A1 / A2 over this workgroup's graphs (float bits)
    if (tid < 24) amaxv[tid] = 0u;
    static_assert(G * N == 64, "the column-sum bound is reduced by wave 0");
    int phase = 0;
    auto note_max = [&](float m, int ph) {       // wave maximum -> one LDS atomic per wave
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if ((tid & 63) == 0) atomicMax(&amaxv[ph], __float_as_uint(m));
    };
    auto phase_scale = [&](int ph) { return pow2_scale(__uint_as_float(amaxv[ph])); };
    const float *P16 = P + lay.delta16;
    const float *WSC = P + lay.wsc;           // 1 / sw of every split-arithmetic matrix (dnag_pred_pack)

    // ---- inputs ---------------------------------------------------------------------------------
    for (int i = tid; i < R * XS; i += NT) {
        const int r = i / XS, v = i - r * XS;
        const int g = r / N;
        xin[i] = (v < V && g0 + g < batch) ? x[((size_t)g0 * N + r) * V + v] : 0.f;
    }
    // Ahat = rn(M + I)   (digcn_meta.py:120)
    for (int i = tid; i < G * N; i += NT) {
        const int g = i / N, r = i - g * N;
        const int gg = (g0 + g < batch) ? g0 + g : 0;
        const float *m = mask + (size_t)(mstride ? gg : 0) * N * N + r * N;
        float row[N], s = 0.f;
#pragma unroll
        for (int j = 0; j < N; ++j) { row[j] = m[j] + (j == r ? 1.f : 0.f); s += row[j]; }
#pragma unroll
        for (int j = 0; j < N; ++j) Ah[(g * N + r) * N + j] = row[j] / s;
    }
    __syncthreads();
    PRED_STAMP(1);
    // A1 = rn(Ahat), A2 = rn(Ahat^T)   (digcn_meta.py:42, :44)
    for (int i = tid; i < G * N; i += NT) {
        const int g = i / N, r = i - g * N;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < N; ++j) { s1 += Ah[(g * N + r) * N + j]; s2 += Ah[(g * N + j) * N + r]; }
#pragma unroll
        for (int j = 0; j < N; ++j) {
            A1[(g * N + r) * N + j] = Ah[(g * N + r) * N + j] / s1;
            A2[(g * N + r) * N + j] = Ah[(g * N + j) * N + r] / s2;
        }
    }
    // c0 = b0 + W0[:, :nz] D_mu   (the dataset half of pred_fc.0, constant per call)
    // 16 lanes per output: each sums every 16th term (independent loads, one latency), DPP row sum in a fixed order
    for (int i = tid; i < ((hs * 16 + NT - 1) / NT) * NT; i += NT) {
        const int j = i >> 4, sub = i & 15;
        float acc = 0.f;
        if (j < hs) {
            const float *w = P + lay.p0_w + (size_t)j * (nz + LH);
            for (int z = sub; z < nz; z += 16) acc = fmaf(w[z], d_mu[z], acc);
        }
        acc = row16_sum(acc);
        if (j < hs && sub == 0) c0[j] = acc + P[lay.p0_b + j];
    }
    __syncthreads();   // Ah (aliasing U1) is dead from here
    PRED_STAMP(2);

    // ---- layer 0 (K = V): A1 x and A2 x once, then Z = (A x) W on FMAs -----------------------------
    if (tid < G * N) {   // largest column sum of A1 / A2: bound of the backward mixing (mix_bwd)
        const int g = tid / N, j = tid - g * N;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int r = 0; r < N; ++r) { s1 += fabsf(A1[(g * N + r) * N + j]); s2 += fabsf(A2[(g * N + r) * N + j]); }
        float m = fmaxf(s1, s2);       // G * N == 64: exactly wave 0
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (tid == 0) colsum_max = __float_as_uint(m);
    }
    if (!(DNAG_PRED_SKIP & 1)) {
        for (int i = tid; i < R * XS; i += NT) {
            const int r = i / XS, v = i - r * XS;
            const int g = r / N, n = r - g * N;
            float u1 = 0.f, u2 = 0.f;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const float xv = xin[(g * N + j) * XS + v];
                u1 = fmaf(A1[(g * N + n) * N + j], xv, u1);
                u2 = fmaf(A2[(g * N + n) * N + j], xv, u2);
            }
            ux1[i] = u1;
            ux2[i] = u2;
        }
        for (int i = tid; i < BL; i += NT) bits[i] = 0u;     // layer 0's relu masks are OR-ed together below
        __syncthreads();
        PRED_STAMP(3);
        const float *W1 = P + lay.w1_0, *W2 = P + lay.w2_0;   // [V][GH]
        // thread = (column c, row class q): the column's 2 V weights stay in registers over its R / QN rows
        constexpr int QN = NT / GH;
        float hmax = 0.f;
        if (tid < QN * GH) {
            const int c = tid % GH, q = tid / GH;
            float w1[V], w2[V];
#pragma unroll
            for (int v = 0; v < V; ++v) { w1[v] = W1[v * GH + c]; w2[v] = W2[v * GH + c]; }
            static_assert(QN == 4 && R == 64, "rows r = q mod 4: accumulator register i = q + 4 * (r32 >> 3), half (r32 >> 2) & 1");
            uint32_t w[4] = {0u, 0u, 0u, 0u};       // [row tile][half]
#pragma unroll
            for (int k = 0; k < R / QN; ++k) {
                const int r = q + QN * k;
                float z1 = 0.f, z2 = 0.f;
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    z1 = fmaf(ux1[r * XS + v], w1[v], z1);
                    z2 = fmaf(ux2[r * XS + v], w2[v], z2);
                }
                const float hv = (fmaxf(z1, 0.f) + fmaxf(z2, 0.f)) * 0.5f;
                Hb[r * HS + c] = hv;
                hmax = fmaxf(hmax, hv);
                // r = 4k + q: tile k >> 3, row-in-tile r32 = 4 (k & 7) + q -> half k & 1, register q + 4 * ((k & 7) >> 1)
                const int i = q + 4 * ((k & 7) >> 1);
                w[(k >> 3) * 2 + (k & 1)] |= ((z1 > 0.f ? 1u : 0u) << i) | ((z2 > 0.f ? 1u : 0u) << (16 + i));
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) atomicOr(&bits[t * BW + c], w[t]);
        }
        note_max(hmax, 0);      // max |H_1|: scale bound of the first forward mixing (phase 0)
        __syncthreads();
        PRED_STAMP(4);
    }

    auto id_xf = [](v4f a, int) { return a; };
    // Adjacency mixing, one (graph, column) item per thread: the 8 rows of the column stay in registers and are written
    // to U1 / U2 ONCE as split-pair fp16 rows (the five column-tile waves that read a row would otherwise each repeat
    // the split).  The power-of-two scale must be known before the first value is written, so it comes from a bound:
    // the producer of Hb noted max |Hb| in amaxv[ph]; A1 and A2 are row-normalised with non-negative entries, so
    // |A H| <= max |H| (forward), and |A^T D| <= max |D| * (largest column sum of A1 / A2, colsum_max) (backward).
    // two adjacent columns (c even) of one row: the hi halves are one dword, the lo halves the dword 8 halves on
    auto put_split2 = [&](float *U, int row, int at, float v0, float v1, float sa) {
        _Float16 h0, l0, h1, l1;
        split16(v0, sa, h0, l0);
        split16(v1, sa, h1, l1);
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        _Float16 *d = reinterpret_cast<_Float16 *>(U + row * HS) + at;
        *reinterpret_cast<h2 *>(d) = h2{h0, h1};
        *reinterpret_cast<h2 *>(d + 8) = h2{l0, l1};
    };
    static_assert(GH % 2 == 0 && HS % 2 == 0, "mixing items are column pairs");
    typedef float f2v __attribute__((ext_vector_type(2)));
    auto mix_fwd = [&](int ph) {   // U1 = A1 H, U2 = A2 H; one (graph, column pair) per thread
        const float sa = pow2_scale(__uint_as_float(amaxv[ph]));
        int t0 = tid;
        asm volatile("" : "+v"(t0));       // the item's LDS addresses are computed here, not hoisted out of the layer loop (spills)
#pragma unroll 1
        for (int i = t0; i < G * (GH / 2); i += NT) {
            const int g = i / (GH / 2), c = 2 * (i - g * (GH / 2));
            const int at = (c >> 3) * 16 + (c & 7);          // split_at(c), in halves
            f2v hv[N];
#pragma unroll
            for (int j = 0; j < N; ++j) hv[j] = *reinterpret_cast<const f2v *>(Hb + (g * N + j) * HS + c);
#pragma unroll
            for (int n = 0; n < N; ++n) {
                float a1[N], a2[N];
                load_row<N>(A1 + (g * N + n) * N, a1);
                load_row<N>(A2 + (g * N + n) * N, a2);
                float u1x = 0.f, u1y = 0.f, u2x = 0.f, u2y = 0.f;
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    u1x = fmaf(a1[j], hv[j].x, u1x); u1y = fmaf(a1[j], hv[j].y, u1y);
                    u2x = fmaf(a2[j], hv[j].x, u2x); u2y = fmaf(a2[j], hv[j].y, u2y);
                }
                put_split2(U1, g * N + n, at, u1x, u1y, sa);
                put_split2(U2, g * N + n, at, u2x, u2y, sa);
            }
        }
        return sa;
    };
    // P1 = A1^T (1/2 dH . m1), P2 = A2^T (1/2 dH . m2) into U1, U2
    auto mix_bwd = [&](int l, int ph) {
        const uint32_t *bl = bits + (size_t)l * BL;
        const float sa = pow2_scale(__uint_as_float(amaxv[ph]) * 0.5f * __uint_as_float(colsum_max));
        int t0 = tid;
        asm volatile("" : "+v"(t0));       // the item's LDS addresses are computed here, not hoisted out of the layer loop (spills)
#pragma unroll 1
        for (int i = t0; i < G * (GH / 2); i += NT) {
            const int g = i / (GH / 2), c = 2 * (i - g * (GH / 2));
            const int at = (c >> 3) * 16 + (c & 7);
            // rows 8g .. 8g+7 = row tile g >> 2, rows-in-tile 8 (g & 3) + j: half j >> 2, register (j & 3) + 4 (g & 3)
            const uint32_t bw0[2] = {bl[((g >> 2) * 2 + 0) * BW + c], bl[((g >> 2) * 2 + 1) * BW + c]};
            const uint32_t bw1[2] = {bl[((g >> 2) * 2 + 0) * BW + c + 1], bl[((g >> 2) * 2 + 1) * BW + c + 1]};
            float p1x[N], p1y[N], p2x[N], p2y[N];
#pragma unroll
            for (int n = 0; n < N; ++n) { p1x[n] = 0.f; p1y[n] = 0.f; p2x[n] = 0.f; p2y[n] = 0.f; }
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const int rj = g * N + j;
                const f2v dh2 = *reinterpret_cast<const f2v *>(Hb + rj * HS + c);
                const float dhx = dh2.x * 0.5f, dhy = dh2.y * 0.5f;
                const int bi = (j & 3) + 4 * (g & 3);
                const float d1x = ((bw0[j >> 2] >> bi) & 1u) ? dhx : 0.f, d2x = ((bw0[j >> 2] >> (16 + bi)) & 1u) ? dhx : 0.f;
                const float d1y = ((bw1[j >> 2] >> bi) & 1u) ? dhy : 0.f, d2y = ((bw1[j >> 2] >> (16 + bi)) & 1u) ? dhy : 0.f;
                float a1[N], a2[N];
                load_row<N>(A1 + rj * N, a1);
                load_row<N>(A2 + rj * N, a2);
#pragma unroll
                for (int n = 0; n < N; ++n) {
                    p1x[n] = fmaf(a1[n], d1x, p1x[n]); p1y[n] = fmaf(a1[n], d1y, p1y[n]);
                    p2x[n] = fmaf(a2[n], d2x, p2x[n]); p2y[n] = fmaf(a2[n], d2y, p2y[n]);
                }
            }
#pragma unroll
            for (int n = 0; n < N; ++n) {
                put_split2(U1, g * N + n, at, p1x[n], p1y[n], sa);
                put_split2(U2, g * N + n, at, p2x[n], p2y[n], sa);
            }
        }
        return sa;
    };

    // ---- layers 1 .. L-1 forward -----------------------------------------------------------------
    static_assert(PRED_NW == PRED_NT * MT, "one (column tile, row tile) pair per wave in the layer GEMMs");
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int my_nt = wv / MT, my_mt = wv - my_nt * MT;
    const int ln = tid & 63, lr = ln & 31, lh = ln >> 5;
    const float *u1p = U1 + (my_mt * 32 + lr) * HS + lh * (GH / 2), *u2p = U2 + (my_mt * 32 + lr) * HS + lh * (GH / 2);
    // B fragments of the wave's first tile (pinned issue point)
    auto prefetch_b = [&](const float *W, float (&b)[GH / 2]) {
        asm volatile("" : "+s"(W));
        load_frag<GH>(W, my_nt, b);
    };
    for (int l = 1; l < L; ++l) {
        const int ph = phase++;
        const float *W1t = P16 + lay.w1f[l], *W2t = P16 + lay.w2f[l];   // fp16 fragments of W^T [out][in]
        const float sa = mix_fwd(ph);
        float bw1[GH / 2];
        prefetch_b(W1t, bw1);       // issued before the barrier: lands while the other waves finish their rows
        __syncthreads();
        PRED_STAMP(6);
        uint32_t *bl = bits + (size_t)l * BL;
        if (!(DNAG_PRED_SKIP & 4)) {
            // each wave owns one (column tile, row tile); Z1 and Z2 back to back
            f32x16 acc1, acc2;
            two_tiles16<GH>(u1p, u2p, bw1, W2t, my_nt, WSC[2 * l] / sa, WSC[2 * l + 1] / sa, acc1, acc2);
            const int col = my_nt * 32 + lr;
            float hmax = 0.f;
            uint32_t w = 0u;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = my_mt * 32 + acc_row(i, lh);
                w |= ((acc1[i] > 0.f ? 1u : 0u) << i) | ((acc2[i] > 0.f ? 1u : 0u) << (16 + i));
                const float hv = (fmaxf(acc1[i], 0.f) + fmaxf(acc2[i], 0.f)) * 0.5f;
                if (col < GH) { Hb[row * HS + col] = hv; hmax = fmaxf(hmax, hv); }
            }
            bl[(my_mt * 2 + lh) * BW + col] = w;
            if (l + 1 < L) note_max(hmax, ph + 1);     // scale bound of the next layer's mixing
        }
        __syncthreads();
        PRED_STAMP(7);
    }

    // ---- head forward: graphs become the rows 0..G-1 of a 32-row MFMA tile ---------------------------
    if (!(DNAG_PRED_SKIP & 8)) {
        const int ph = phase++;
        float vmax = 0.f;
        for (int i = tid; i < G * GH; i += NT) {
            const int g = i / GH, c = i - g * GH;
            float s = 0.f;
#pragma unroll
            for (int n = 0; n < N; ++n) s += Hb[(g * N + n) * HS + c];
            const int gg = (g0 + g < batch) ? g0 + g : 0;
            const float pv = s / (float)N + temb[(size_t)(tstride ? gg : 0) * GH + c];   // digcn_meta.py:19-22, :125
            U1[g * HS + c] = pv;
            vmax = fmaxf(vmax, fabsf(pv));
        }
        // rows G .. 31 of the MFMA tile hold stale values of the last mixing: an MFMA output row depends on its own A row
        // only, and the epilogue drops those rows, so they take no part in the scale (an overflow there is harmless)
        note_max(vmax, ph);
        __syncthreads();
        PRED_STAMP(8);
        const float sa = phase_scale(ph);
        // a = Wfc1 p  (fc1 [LH][GH]: rows K-contiguous) -> U2[g][0..127]
        run_jobs<GH, 1, PRED_NW, true>(
            LH / 32, U1, HS,
            [&](int nt, int, int, float (&b)[GH / 2]) { load_frag<GH>(P16 + lay.fc1f, nt, b); },
            id_xf,
            [&](int, int nt, int r, int h, const f32x16 &acc) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = acc_row(i, h);
                    if (row < G) U2[row * HS + nt * 32 + r] = acc[i];
                }
            }, sa, WSC[2 * L]);
        __syncthreads();
        PRED_STAMP(9);
        for (int i = tid; i < G * hs; i += NT) {   // tanh(W0a a + c0)  (p0_at [LH][hs])
            const int g = i / hs, j = i - g * hs;
            const float *w = P + lay.p0_at + j;
            float acc = 0.f;
#pragma unroll 8
            for (int o = 0; o < LH; ++o) acc = fmaf(w[(size_t)o * hs], U2[g * HS + o], acc);
            tv[g * TS + j] = tanhf(acc + c0[j]);
        }
    }
    __syncthreads();
    PRED_STAMP(10);
    if (y != nullptr && tid < G && g0 + tid < batch) {
        float acc = 0.f;
        for (int j = 0; j < hs; ++j) acc = fmaf(P[lay.p2_w + j], tv[tid * TS + j], acc);
        y[g0 + tid] = acc + P[lay.p2_b];
    }
    if (grad == nullptr) return;
    __syncthreads();
    PRED_STAMP(11);

    // ---- head backward ----------------------------------------------------------------------------
    if (!(DNAG_PRED_SKIP & 16)) {
        for (int i = tid; i < G * hs; i += NT) {   // dpre = w2 (1 - tanh^2)
            const int g = i / hs, j = i - g * hs;
            const float t = tv[g * TS + j];
            tv[g * TS + j] = P[lay.p2_w + j] * (1.f - t * t);
        }
        __syncthreads();
        PRED_STAMP(12);
        const int ph = phase++;
        float vmax = 0.f;
        for (int i = tid; i < G * LH; i += NT) {   // da[o] = sum_j W0[j][nz + o] dpre[j] -> U1[g][0..127]
            const int g = i / LH, o = i - g * LH;
            const float *w = P + lay.p0_w + nz + o;
            float acc = 0.f;
#pragma unroll 8
            for (int j = 0; j < hs; ++j) acc = fmaf(w[(size_t)j * (nz + LH)], tv[g * TS + j], acc);
            U1[g * HS + o] = acc;
            vmax = fmaxf(vmax, fabsf(acc));
        }
        note_max(vmax, ph);
        __syncthreads();
        PRED_STAMP(13);
        const float sa = phase_scale(ph);
        // dp = Wfc1^T da  (fc1t [GH][LH]: rows K-contiguous);  dH_L[n][c] = dp[c] / N for every node n
        run_jobs<LH, 1, PRED_NW, true>(
            PRED_NT, U1, HS,
            [&](int nt, int, int, float (&b)[LH / 2]) { load_frag<LH>(P16 + lay.fc1b, nt, b); },
            id_xf,
            [&](int, int nt, int r, int h, const f32x16 &acc) {
                const int col = nt * 32 + r;
                float dmax = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int g = acc_row(i, h);
                    if (g < G && col < GH) {
                        const float d = acc[i] / (float)N;
                        dmax = fmaxf(dmax, fabsf(d));
#pragma unroll
                        for (int n = 0; n < N; ++n) Hb[(g * N + n) * HS + col] = d;
                    }
                }
                note_max(dmax, ph + 1);    // max |dH_L|: scale bound of the first backward mixing
            }, sa, WSC[2 * L]);
    }
    __syncthreads();
    PRED_STAMP(14);

    // ---- layers L-1 .. 1 backward: dH_l = P1 W1^T + P2 W2^T  (B rows = W[in][:] K-contiguous) ------
    for (int l = L - 1; l >= 1; --l) {
        const int ph = phase++;
        const float *W1 = P16 + lay.w1b[l], *W2 = P16 + lay.w2b[l];   // fp16 fragments of W [in][out]
        const float sa = mix_bwd(l, ph);
        float bw1[GH / 2];
        prefetch_b(W1, bw1);
        __syncthreads();
        PRED_STAMP(16);
        if (!(DNAG_PRED_SKIP & 64)) {
            f32x16 acc1, acc2;
            two_tiles16<GH>(u1p, u2p, bw1, W2, my_nt, WSC[2 * l] / sa, WSC[2 * l + 1] / sa, acc1, acc2);
            const int col = my_nt * 32 + lr;
            float dmax = 0.f;
            if (col < GH) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float d = acc1[i] + acc2[i];
                    Hb[(my_mt * 32 + acc_row(i, lh)) * HS + col] = d;
                    dmax = fmaxf(dmax, fabsf(d));
                }
            }
            note_max(dmax, ph + 1);        // scale bound of the next mixing (layer l - 1)
        }
        __syncthreads();
        PRED_STAMP(17);
    }
    // ---- layer 0 backward: dx[r][v] = sum_c P1[r][c] W1[v][c] + P2[r][c] W2[v][c]  (one column tile) ----
    const int ph0 = phase++;
    const float sa0 = mix_bwd(0, ph0);
    __syncthreads();
    PRED_STAMP(18);
    if (!(DNAG_PRED_SKIP & 128) && wv < MT) {   // one row tile per wave
        const float *W1 = P16 + lay.w1b[0], *W2 = P16 + lay.w2b[0];   // fp16 fragments [V -> 32][GH], one tile
        float b1[GH / 2];
        load_frag<GH>(W1, 0, b1);
        f32x16 acc1, acc2;
        two_tiles16<GH>(U1 + (wv * 32 + lr) * HS + lh * (GH / 2), U2 + (wv * 32 + lr) * HS + lh * (GH / 2), b1, W2, 0, WSC[0] / sa0, WSC[1] / sa0, acc1, acc2);
        if (lr < V) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = wv * 32 + acc_row(i, lh);
                if (g0 + row / N < batch) grad[((size_t)g0 * N + row) * V + lr] = acc1[i] + acc2[i];
            }
        }
    }
#ifdef DNAG_PRED_STAMPS
    __syncthreads();
    PRED_STAMP(99);
    if (blockIdx.x == 5 && tid == 0) for (int i = 0; i < nts_; ++i) grad[((size_t)g0 * N) * V + i] = (float)(ts_[i] - ts_[0]);
#endif
}

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
#define PRED_HT 68         // row stride of the transposed activation buffer [144 columns][64 rows] (68 % 64 == 4: conflict-free b128)
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
        const size_t fl = (size_t)PRED_GH * PRED_HT    // H^T
                          + 2 * (size_t)R * PRED_HS    // U1, U2
                          + 4 * (size_t)G * N * N      // A1, A2 and their transposes
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
    constexpr int HT = PRED_HT;
    static_assert(R == 64 && HT >= R + 4, "H^T rows hold the workgroup's 64 rows");
    float *HbT = smem;                      // [GH][HT]: activations / their gradients, TRANSPOSED (column-major over the rows)
    float *U1 = HbT + GH * HT;              // [R][HS]
    float *U2 = U1 + R * HS;                // [R][HS]
    float *A1 = U2 + R * HS;                // [G][N][N]
    float *A2 = A1 + G * N * N;             // [G][N][N]
    float *A1T = A2 + G * N * N;            // [G][N][N] transposes (backward mixing)
    float *A2T = A1T + G * N * N;
    const int L = lay.L, hs = lay.hs, nz = lay.nz;
    const int TS = (hs + 3) & ~3;           // row stride of tv
    float *tv = A2T + G * N * N;            // [G][TS]
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
            const float a1v = Ah[(g * N + r) * N + j] / s1, a2v = Ah[(g * N + j) * N + r] / s2;
            A1[(g * N + r) * N + j] = a1v;
            A2[(g * N + r) * N + j] = a2v;
            A1T[(g * N + j) * N + r] = a1v;
            A2T[(g * N + j) * N + r] = a2v;
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
                HbT[c * HT + r] = hv;
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
    // Adjacency mixing on the matrix cores, one (column tile, row tile) job per wave like the layer GEMMs:
    //   D[c][row] = sum_j X[j][c] B[row][j],   X = H (forward) or 1/2 dH . relu mask (backward), B = A1 / A2 or their transposes.
    // B is block diagonal (8 x 8 per graph): of the 64 k only the 32 of the row tile's four graphs contribute, two k steps.
    // A operand: lane (column c, k half) reads 8 consecutive rows of H^T (two ds_read_b128) and splits them with `sa_in`;
    // B operand: lane (row, k half) holds its graph's adjacency row where the k half is its own graph, zeros elsewhere
    // (entries in [0, 1], scale 1024).  A lane of the result holds one row and quads of consecutive columns, which are
    // contiguous in the split-pair rows of U1 / U2: two 8-byte stores per quad.  `out_mul` turns the accumulator into the
    // stored operand of the following GEMM (its scale over sa_in 1024, times the 1/2 of the backward pass).
    static_assert(PRED_NW == PRED_NT * MT, "one (column tile, row tile) pair per wave");
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int my_nt = wv / MT, my_mt = wv - my_nt * MT;
    const int ln = tid & 63, lr = ln & 31, lh = ln >> 5;
    constexpr float ADJ_SCALE = 1024.0f;
    auto mix_mfma = [&](const float *B1, const float *B2, float sa_in, float out_mul, const uint32_t *bl) {
        const int cc = (my_nt * 32 + lr < GH) ? my_nt * 32 + lr : GH - 1;       // lanes past column 143: any valid address
        const float *xp = HbT + cc * HT + my_mt * 32 + 8 * lh;
        const int own = lr >> 3;                                                 // graph of this lane's row inside the tile
        const float *b1p = B1 + ((my_mt * 4 + own) * N + (lr & 7)) * N, *b2p = B2 + ((my_mt * 4 + own) * N + (lr & 7)) * N;
        uint32_t w0 = 0u, w1 = 0u;
        if (bl) { w0 = bl[(my_mt * 2 + 0) * BW + cc]; w1 = bl[(my_mt * 2 + 1) * BW + cc]; }
        f32x16 acc1, acc2;
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc1[i] = 0.f; acc2[i] = 0.f; }
        auto split8 = [&](const v4f a, const v4f b, float sc, h8 &hi, h8 &lo) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                _Float16 h_, l_;
                split16(e < 4 ? a[e] : b[e - 4], sc, h_, l_);
                hi[e] = h_; lo[e] = l_;
            }
        };
        auto masked = [&](const v4f a, uint32_t nib) {
            return v4f{(nib & 1u) ? a.x : 0.f, (nib & 2u) ? a.y : 0.f, (nib & 4u) ? a.z : 0.f, (nib & 8u) ? a.w : 0.f};
        };
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
            const v4f x0 = *reinterpret_cast<const v4f *>(xp + 16 * s_), x1 = *reinterpret_cast<const v4f *>(xp + 16 * s_ + 4);
            const bool mine = own == 2 * s_ + lh;
            v4f p0 = {0.f, 0.f, 0.f, 0.f}, p1 = p0, q0 = p0, q1 = p0;
            if (mine) {
                p0 = *reinterpret_cast<const v4f *>(b1p); p1 = *reinterpret_cast<const v4f *>(b1p + 4);
                q0 = *reinterpret_cast<const v4f *>(b2p); q1 = *reinterpret_cast<const v4f *>(b2p + 4);
            }
            h8 b1h, b1l, b2h, b2l;
            split8(p0, p1, ADJ_SCALE, b1h, b1l);
            split8(q0, q1, ADJ_SCALE, b2h, b2l);
            if (bl) {
                // rows 8 (2 s + lh) + e of the tile: e < 4 in the word of lane half 0 (bits 4 q + e), e >= 4 in that of half 1
                const int q4 = 4 * (2 * s_ + lh);
                h8 ah, al;
                split8(masked(x0, (w0 >> q4) & 15u), masked(x1, (w1 >> q4) & 15u), sa_in, ah, al);
                acc1 = mfma16(ah, b1h, acc1); acc1 = mfma16(ah, b1l, acc1); acc1 = mfma16(al, b1h, acc1);
                split8(masked(x0, (w0 >> (16 + q4)) & 15u), masked(x1, (w1 >> (16 + q4)) & 15u), sa_in, ah, al);
                acc2 = mfma16(ah, b2h, acc2); acc2 = mfma16(ah, b2l, acc2); acc2 = mfma16(al, b2h, acc2);
            } else {
                h8 ah, al;
                split8(x0, x1, sa_in, ah, al);
                acc1 = mfma16(ah, b1h, acc1); acc1 = mfma16(ah, b1l, acc1); acc1 = mfma16(al, b1h, acc1);
                acc2 = mfma16(ah, b2h, acc2); acc2 = mfma16(ah, b2l, acc2); acc2 = mfma16(al, b2h, acc2);
            }
        }
        // lane = row my_mt * 32 + lr; registers 4 q .. 4 q + 3 = columns my_nt * 32 + 8 q + 4 lh + (0 .. 3)
        const int row = my_mt * 32 + lr;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c0 = my_nt * 32 + 8 * q + 4 * lh;
            if (c0 < GH) {
                h4 h1_, l1_, h2_, l2_;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    _Float16 a_, b_;
                    split16(acc1[4 * q + e], out_mul, a_, b_); h1_[e] = a_; l1_[e] = b_;
                    split16(acc2[4 * q + e], out_mul, a_, b_); h2_[e] = a_; l2_[e] = b_;
                }
                const int at = (c0 >> 3) * 16 + (c0 & 7);            // split_at(c0), in halves
                _Float16 *d1 = reinterpret_cast<_Float16 *>(U1 + row * HS) + at, *d2 = reinterpret_cast<_Float16 *>(U2 + row * HS) + at;
                *reinterpret_cast<h4 *>(d1) = h1_; *reinterpret_cast<h4 *>(d1 + 8) = l1_;
                *reinterpret_cast<h4 *>(d2) = h2_; *reinterpret_cast<h4 *>(d2 + 8) = l2_;
            }
        }
    };
    // The power-of-two scale of the stored operand comes from a bound, the producer of H^T noted max |H| in amaxv[ph]:
    // A1 and A2 are row-normalised with non-negative entries, so |A H| <= max |H| (forward), and
    // |A^T D| <= max |D| * (largest column sum of A1 / A2, colsum_max) (backward).
    auto mix_fwd = [&](int ph) {   // U1 = A1 H, U2 = A2 H
        const float sa = pow2_scale(__uint_as_float(amaxv[ph]));
        mix_mfma(A1, A2, sa, 1.0f / ADJ_SCALE, nullptr);
        return sa;
    };
    auto mix_bwd = [&](int l, int ph) {   // P1 = A1^T (1/2 dH . m1), P2 = A2^T (1/2 dH . m2) into U1, U2
        const float sd = pow2_scale(__uint_as_float(amaxv[ph]));
        const float sa = pow2_scale(__uint_as_float(amaxv[ph]) * 0.5f * __uint_as_float(colsum_max));
        mix_mfma(A1T, A2T, sd, 0.5f * sa / (sd * ADJ_SCALE), bits + (size_t)l * BL);
        return sa;
    };

    // ---- layers 1 .. L-1 forward -----------------------------------------------------------------
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
            for (int q = 0; q < 4; ++q) {      // registers 4 q .. 4 q + 3 = rows my_mt * 32 + 8 q + 4 lh + (0 .. 3): one quad of H^T
                v4f h4v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * q + e;
                    w |= ((acc1[i] > 0.f ? 1u : 0u) << i) | ((acc2[i] > 0.f ? 1u : 0u) << (16 + i));
                    h4v[e] = (fmaxf(acc1[i], 0.f) + fmaxf(acc2[i], 0.f)) * 0.5f;
                    if (col < GH) hmax = fmaxf(hmax, h4v[e]);
                }
                if (col < GH) *reinterpret_cast<v4f *>(HbT + col * HT + my_mt * 32 + 8 * q + 4 * lh) = h4v;
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
            for (int n = 0; n < N; ++n) s += HbT[c * HT + g * N + n];
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
                        for (int n = 0; n < N; ++n) HbT[col * HT + g * N + n] = d;
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
                for (int q = 0; q < 4; ++q) {
                    v4f d4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        d4[e] = acc1[4 * q + e] + acc2[4 * q + e];
                        dmax = fmaxf(dmax, fabsf(d4[e]));
                    }
                    *reinterpret_cast<v4f *>(HbT + col * HT + my_mt * 32 + 8 * q + 4 * lh) = d4;
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

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
template <int N, int V>
__global__ __launch_bounds__(PRED_NW * 64) void predictor_kernel(const float *__restrict__ P, const PredLayout lay,
                                                                 const float *__restrict__ x, const float *__restrict__ temb,
                                                                 int tstride, const float *__restrict__ mask, int mstride,
                                                                 const float *__restrict__ d_mu, float *__restrict__ y,
                                                                 float *__restrict__ grad, int batch, int dbg_skip) {
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
    uint32_t *bits = reinterpret_cast<uint32_t *>(ux2 + R * XS);   // [L][2][R][NT] (cols of tile nt, row r)
    float *Ah = U1;                         // scratch for Ahat during set-up

    const int tid = threadIdx.x;
    const int g0 = blockIdx.x * G;

    // Split-arithmetic MFMAs (dnag_gemm16.hip.h): every GEMM phase reads fp16 hi/lo weight fragments (P + delta16) and
    // splits its A operand with a power-of-two scale derived from the operand's largest magnitude in this workgroup
    // (backward operands are gradients of unknown size): the producers of U1 / U2 leave max |value| of phase k in amaxv[k].
    __shared__ unsigned int amaxv[24];
    if (tid < 24) amaxv[tid] = 0u;
    int phase = 0;
    auto note_max = [&](float m, int ph) {       // wave maximum -> one LDS atomic per wave
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if ((tid & 63) == 0) atomicMax(&amaxv[ph], __float_as_uint(m));
    };
    auto phase_scale = [&](int ph) { return pow2_scale(__uint_as_float(amaxv[ph])); };
    const float *P16 = P + lay.delta16;

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
    for (int j = tid; j < hs; j += NT) {
        const float *w = P + lay.p0_w + (size_t)j * (nz + LH);
        float acc = 0.f;
        for (int z = 0; z < nz; ++z) acc = fmaf(w[z], d_mu[z], acc);
        c0[j] = acc + P[lay.p0_b + j];
    }
    __syncthreads();   // Ah (aliasing U1) is dead from here

    // ---- layer 0 (K = V): A1 x and A2 x once, then Z = (A x) W on FMAs -----------------------------
    if (!(dbg_skip & 1)) {
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
        __syncthreads();
        const float *W1 = P + lay.w1_0, *W2 = P + lay.w2_0;   // [V][GH]
        for (int i = tid; i < R * GH; i += NT) {
            const int r = i / GH, c = i - r * GH;
            float z1 = 0.f, z2 = 0.f;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                z1 = fmaf(ux1[r * XS + v], W1[v * GH + c], z1);
                z2 = fmaf(ux2[r * XS + v], W2[v * GH + c], z2);
            }
            Hb[r * HS + c] = (fmaxf(z1, 0.f) + fmaxf(z2, 0.f)) / 2.f;
            U1[r * HS + c] = z1 > 0.f ? 1.f : 0.f;   // relu masks, packed into bit words below
            U2[r * HS + c] = z2 > 0.f ? 1.f : 0.f;
        }
        __syncthreads();
        for (int i = tid; i < 2 * R * PRED_NT; i += NT) {
            const int nt = i % PRED_NT, r = (i / PRED_NT) % R, which = i / (PRED_NT * R);
            const float *src = (which ? U2 : U1) + r * HS + nt * 32;
            uint32_t w = 0;
            const int nb = (GH - nt * 32 < 32) ? GH - nt * 32 : 32;
            for (int b = 0; b < nb; ++b)
                if (src[b] != 0.f) w |= (1u << b);
            bits[(size_t)which * R * PRED_NT + r * PRED_NT + nt] = w;
        }
        __syncthreads();
    }

    auto id_xf = [](v4f a, int) { return a; };
    // adjacency mixing, one (graph, column) item per thread: the 8 rows of the column stay in registers
    auto mix_fwd = [&](int ph) {   // U1 = A1 H, U2 = A2 H
        float vmax = 0.f;
        for (int i = tid; i < G * GH; i += NT) {
            const int g = i / GH, c = i - g * GH;
            float hv[N];
#pragma unroll
            for (int j = 0; j < N; ++j) hv[j] = Hb[(g * N + j) * HS + c];
#pragma unroll
            for (int n = 0; n < N; ++n) {
                float a1[N], a2[N];
                load_row<N>(A1 + (g * N + n) * N, a1);
                load_row<N>(A2 + (g * N + n) * N, a2);
                float u1 = 0.f, u2 = 0.f;
#pragma unroll
                for (int j = 0; j < N; ++j) { u1 = fmaf(a1[j], hv[j], u1); u2 = fmaf(a2[j], hv[j], u2); }
                U1[(g * N + n) * HS + c] = u1;
                U2[(g * N + n) * HS + c] = u2;
                vmax = fmaxf(vmax, fmaxf(fabsf(u1), fabsf(u2)));
            }
        }
        note_max(vmax, ph);
    };
    // P1 = A1^T (1/2 dH . m1), P2 = A2^T (1/2 dH . m2) into U1, U2
    auto mix_bwd = [&](int l, int ph) {
        const uint32_t *bl = bits + (size_t)l * 2 * R * PRED_NT;
        float vmax = 0.f;
        for (int i = tid; i < G * GH; i += NT) {
            const int g = i / GH, c = i - g * GH;
            const int nt = c >> 5, bit = c & 31;
            float p1[N], p2[N];
#pragma unroll
            for (int n = 0; n < N; ++n) { p1[n] = 0.f; p2[n] = 0.f; }
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const int rj = g * N + j;
                const float dh = Hb[rj * HS + c] * 0.5f;
                const float d1 = ((bl[(0 * R + rj) * PRED_NT + nt] >> bit) & 1u) ? dh : 0.f;
                const float d2 = ((bl[(1 * R + rj) * PRED_NT + nt] >> bit) & 1u) ? dh : 0.f;
                float a1[N], a2[N];
                load_row<N>(A1 + rj * N, a1);
                load_row<N>(A2 + rj * N, a2);
#pragma unroll
                for (int n = 0; n < N; ++n) { p1[n] = fmaf(a1[n], d1, p1[n]); p2[n] = fmaf(a2[n], d2, p2[n]); }
            }
#pragma unroll
            for (int n = 0; n < N; ++n) {
                U1[(g * N + n) * HS + c] = p1[n];
                U2[(g * N + n) * HS + c] = p2[n];
                vmax = fmaxf(vmax, fmaxf(fabsf(p1[n]), fabsf(p2[n])));
            }
        }
        note_max(vmax, ph);
    };
    auto load_b144 = [&](const float *F, int nt, int, int, float (&b)[GH / 2]) {   // fragment-major, zero padded
        load_frag<GH>(F, nt, b);
    };

    // ---- layers 1 .. L-1 forward -----------------------------------------------------------------
    for (int l = 1; l < L; ++l) {
        const int ph = phase++;
        if (!(dbg_skip & 2)) mix_fwd(ph);
        __syncthreads();
        const float sa = phase_scale(ph);
        const float *W1t = P16 + lay.w1f[l], *W2t = P16 + lay.w2f[l];   // fp16 fragments of W^T [out][in]
        uint32_t *bl = bits + (size_t)l * 2 * R * PRED_NT;
        if (!(dbg_skip & 4)) {
            f32x16 acc1;
            // each wave owns one (column tile, row tile); Z1 and Z2 back to back with the same registers
            run_jobs<GH, MT, PRED_NW, true>(
                PRED_NT, U1, HS, [&](int nt, int r, int h, float (&b)[GH / 2]) { load_b144(W1t, nt, r, h, b); }, id_xf,
                [&](int, int, int, int, const f32x16 &acc) { acc1 = acc; }, sa);
            run_jobs<GH, MT, PRED_NW, true>(
                PRED_NT, U2, HS, [&](int nt, int r, int h, float (&b)[GH / 2]) { load_b144(W2t, nt, r, h, b); }, id_xf,
                [&](int mt, int nt, int r, int h, const f32x16 &acc2) {
                    const int col = nt * 32 + r;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int row = mt * 32 + acc_row(i, h);
                        const unsigned long long m1 = __ballot(acc1[i] > 0.f), m2 = __ballot(acc2[i] > 0.f);
                        if (r == 0) {   // lanes 0 and 32: one word per (row, tile) and mask
                            bl[(0 * R + row) * PRED_NT + nt] = (uint32_t)(h ? (m1 >> 32) : m1);
                            bl[(1 * R + row) * PRED_NT + nt] = (uint32_t)(h ? (m2 >> 32) : m2);
                        }
                        if (col < GH) Hb[row * HS + col] = (fmaxf(acc1[i], 0.f) + fmaxf(acc2[i], 0.f)) / 2.f;
                    }
                }, sa);
        }
        __syncthreads();
    }

    // ---- head forward: graphs become the rows 0..G-1 of a 32-row MFMA tile ---------------------------
    if (!(dbg_skip & 8)) {
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
            }, sa);
        __syncthreads();
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
    if (y != nullptr && tid < G && g0 + tid < batch) {
        float acc = 0.f;
        for (int j = 0; j < hs; ++j) acc = fmaf(P[lay.p2_w + j], tv[tid * TS + j], acc);
        y[g0 + tid] = acc + P[lay.p2_b];
    }
    if (grad == nullptr) return;
    __syncthreads();

    // ---- head backward ----------------------------------------------------------------------------
    if (!(dbg_skip & 16)) {
        for (int i = tid; i < G * hs; i += NT) {   // dpre = w2 (1 - tanh^2)
            const int g = i / hs, j = i - g * hs;
            const float t = tv[g * TS + j];
            tv[g * TS + j] = P[lay.p2_w + j] * (1.f - t * t);
        }
        __syncthreads();
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
        const float sa = phase_scale(ph);
        // dp = Wfc1^T da  (fc1t [GH][LH]: rows K-contiguous);  dH_L[n][c] = dp[c] / N for every node n
        run_jobs<LH, 1, PRED_NW, true>(
            PRED_NT, U1, HS,
            [&](int nt, int, int, float (&b)[LH / 2]) { load_frag<LH>(P16 + lay.fc1b, nt, b); },
            id_xf,
            [&](int, int nt, int r, int h, const f32x16 &acc) {
                const int col = nt * 32 + r;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int g = acc_row(i, h);
                    if (g < G && col < GH) {
                        const float d = acc[i] / (float)N;
#pragma unroll
                        for (int n = 0; n < N; ++n) Hb[(g * N + n) * HS + col] = d;
                    }
                }
            }, sa);
    }
    __syncthreads();

    // ---- layers L-1 .. 1 backward: dH_l = P1 W1^T + P2 W2^T  (B rows = W[in][:] K-contiguous) ------
    for (int l = L - 1; l >= 1; --l) {
        const int ph = phase++;
        if (!(dbg_skip & 32)) mix_bwd(l, ph);
        __syncthreads();
        const float sa = phase_scale(ph);
        const float *W1 = P16 + lay.w1b[l], *W2 = P16 + lay.w2b[l];   // fp16 fragments of W [in][out]
        if (!(dbg_skip & 64)) {
            f32x16 acc1;
            run_jobs<GH, MT, PRED_NW, true>(
                PRED_NT, U1, HS, [&](int nt, int r, int h, float (&b)[GH / 2]) { load_b144(W1, nt, r, h, b); }, id_xf,
                [&](int, int, int, int, const f32x16 &acc) { acc1 = acc; }, sa);
            run_jobs<GH, MT, PRED_NW, true>(
                PRED_NT, U2, HS, [&](int nt, int r, int h, float (&b)[GH / 2]) { load_b144(W2, nt, r, h, b); }, id_xf,
                [&](int mt, int nt, int r, int h, const f32x16 &acc2) {
                    const int col = nt * 32 + r;
                    if (col < GH) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) Hb[(mt * 32 + acc_row(i, h)) * HS + col] = acc1[i] + acc2[i];
                    }
                }, sa);
        }
        __syncthreads();
    }
    // ---- layer 0 backward: dx[r][v] = sum_c P1[r][c] W1[v][c] + P2[r][c] W2[v][c]  (one column tile) ----
    const int ph0 = phase++;
    if (!(dbg_skip & 32)) mix_bwd(0, ph0);
    __syncthreads();
    if (!(dbg_skip & 128)) {
        const float sa = phase_scale(ph0);
        const float *W1 = P16 + lay.w1b[0], *W2 = P16 + lay.w2b[0];   // fp16 fragments [V -> 32][GH], one tile
        f32x16 acc1;
        run_jobs<GH, MT, PRED_NW, true>(
            1, U1, HS, [&](int, int, int, float (&b)[GH / 2]) { load_frag<GH>(W1, 0, b); }, id_xf,
            [&](int, int, int, int, const f32x16 &acc) { acc1 = acc; }, sa);
        run_jobs<GH, MT, PRED_NW, true>(
            1, U2, HS, [&](int, int, int, float (&b)[GH / 2]) { load_frag<GH>(W2, 0, b); }, id_xf,
            [&](int mt, int, int r, int h, const f32x16 &acc2) {
                if (r < V) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int row = mt * 32 + acc_row(i, h);
                        if (g0 + row / N < batch) grad[((size_t)g0 * N + row) * V + r] = acc1[i] + acc2[i];
                    }
                }
            }, sa);
    }
}

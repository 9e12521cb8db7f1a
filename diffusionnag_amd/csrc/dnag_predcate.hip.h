// MetaPredictorCATE, the OFA surrogate: backward pass (input gradient) of its graph encoder and the
// small kernels between the tile GEMMs.  Included by dnag_lib.hip.
//
// Restates (reference NAS-Bench-201/): models/cate.py:311-327 (arch_encode: embedding without
// positional encoding, 2-layer encoder, `final` MLP, flatten, `regeress` MLP), :341-348 (predict) and
// the autograd call of models/utils.py:130-145 (regress=True) as a hand-derived backward pass.
//
// Forward (host sequence in dnag_lib.hip):  encoder_kernel<.., SAVE> -> concat [B N][128]
//   -> elu(final.0) -> elu(final.1) -> final.2 -> flatten [B][180] -> elu(regeress.0) -> regeress.1
//   -> tanh(pred_fc.0 [D_mu | r]) -> pred_fc.2 -> y.
// Backward (dy/dx of sum_b y_b): dz = w2 (1 - tanh^2) -> pred_fc.0^T -> regeress.1^T . elu' ->
//   regeress.0^T -> final.2^T . elu' -> final.1^T . elu' -> final.0^T -> dconcat -> encoder backward
//   (this file) -> dx.  Dense layers run on the tile GEMM of dnag_head.hip.h with transposed weights.
#pragma once
#include "dnag_common.h"
#include "dnag_encoder.hip.h"

template <int N, int G, int H, int DFF>
struct PcbGeom {
    static constexpr int R = G * N;
    static constexpr int MT = (R + 31) / 32;
    static constexpr int RP = MT * 32;
    static constexpr int HS = DNAG_D + 4;          // residual-width rows
    static constexpr int QS = 3 * DNAG_D + 4;      // q|k|v rows
    static constexpr int FS = DFF + 4;             // FFN hidden rows (aliases the q|k|v buffer)
    static constexpr size_t lds_floats = 2 * (size_t)RP * HS + (size_t)RP * QS + 2 * (size_t)G * H * N * N + (size_t)G * N * N;
    static_assert(G * H * N <= 256, "one thread per (graph, head, node) in the attention backward");
    static_assert(RP * FS <= RP * QS, "FFN hidden gradient aliases the q|k|v buffer");
};

// Backward of encoder_kernel for the G graphs of a workgroup, from the activations it saved (EncSave).
//   dconcat [rows_pad][L*64] gradient wrt the concatenated per-layer outputs;
//   grad [batch][N][V] <- d/dx.
// Per layer, last to first (u = LayerNorm without affine, affine folded into W' as in the forward):
//   dH += LN_l''^T(dconcat_l)                            layer output norm   (transformer.py:177)
//   dF = (dH W2) . [f > 0];  dH += LNhat^T(dF W1')        feed-forward block  (transformer.py:97-108)
//   dO = dH Wo;  softmax/attention backward -> dq|dk|dv;  dH += LNhat^T(dqkv Wqkv')   (:12-42, :69-80)
// and finally dx = 8 dH We (transformer.py:230).
template <int N, int V, int G, int H, int DFF>
__global__ __launch_bounds__(256, 1) void predcate_enc_bwd_kernel(const float *__restrict__ P, const PredCateLayout lay,
                                                                  const EncSave sv, const float *__restrict__ dconcat,
                                                                  const float *__restrict__ mask, int mstride,
                                                                  float *__restrict__ grad, int batch) {
    typedef PcbGeom<N, G, H, DFF> Ge;
    constexpr int R = Ge::R, MT = Ge::MT, RP = Ge::RP, HS = Ge::HS, QS = Ge::QS, FS = Ge::FS;
    constexpr int D = DNAG_D, DK = D / H;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *dH = smem;                    // [RP][HS] gradient wrt the residual stream
    float *tmp = dH + RP * HS;           // [RP][HS] gradient wrt a normalised row / dO
    float *qb = tmp + RP * HS;           // [RP][QS] q|k|v, then dq|dk|dv in place
    float *fb = qb;                      // [RP][FS] dF (dead before q|k|v is loaded)
    float *Pm = qb + RP * QS;            // [G][H][N][N] attention probabilities
    float *Sm = Pm + G * H * N * N;      // [G][H][N][N] dS / sqrt(dk)
    float *maskb = Sm + G * H * N * N;   // [G][N][N]

    const int tid = threadIdx.x;
    const int g0 = blockIdx.x * G;
    const size_t row0 = (size_t)g0 * N;
    const int L = lay.enc.L, F = lay.enc.F;
    auto live = [&](int lr) { return lr < R && g0 + lr / N < batch; };

    for (int i = tid; i < RP * HS; i += 256) dH[i] = 0.f;
    for (int i = tid; i < G * N * N; i += 256) {
        const int g = i / (N * N), ij = i - g * N * N;
        const int gg = (g0 + g < batch) ? g0 + g : 0;
        maskb[i] = mask[(size_t)(mstride ? gg : 0) * N * N + ij];
    }
    __syncthreads();

    // dH[r] += LayerNorm^T(g) for the row x = xsrc[row0 + r]: 16 lanes per row, DPP statistics.
    //   xh = (x - m) rs;  dx = rs (g - mean(g) - xh mean(g xh))
    auto ln_bwd = [&](const float *xsrc, auto getg) {
        const int sub = tid & 15, slot = tid >> 4;
        // every pass's row is requested before the first is used (a dead row reads the workgroup's row 0 and is zeroed)
        v4f xs[RP / 16], gs[RP / 16];
#pragma unroll
        for (int p = 0; p < RP / 16; ++p) {
            const int r = slot + 16 * p, rr = live(r) ? r : 0;
            xs[p] = *reinterpret_cast<const v4f *>(xsrc + (row0 + rr) * D + 4 * sub);
            gs[p] = getg(rr, sub);
        }
#pragma unroll
        for (int p = 0; p < RP / 16; ++p) {
            const int r = slot + 16 * p;
            const bool lv = live(r);
            v4f xv = xs[p], gv = gs[p];
            if (!lv) { xv = v4f{0.f, 0.f, 0.f, 0.f}; gv = v4f{0.f, 0.f, 0.f, 0.f}; }
            const float m = row16_sum((xv.x + xv.y) + (xv.z + xv.w)) * (1.0f / 64.0f);
            xv.x -= m; xv.y -= m; xv.z -= m; xv.w -= m;
            const float var = row16_sum((xv.x * xv.x + xv.y * xv.y) + (xv.z * xv.z + xv.w * xv.w)) * (1.0f / 64.0f);
            const float rs = 1.0f / sqrtf(var + 1e-5f);
            xv.x *= rs; xv.y *= rs; xv.z *= rs; xv.w *= rs;
            const float mg = row16_sum((gv.x + gv.y) + (gv.z + gv.w)) * (1.0f / 64.0f);
            const float mgx = row16_sum((gv.x * xv.x + gv.y * xv.y) + (gv.z * xv.z + gv.w * xv.w)) * (1.0f / 64.0f);
            if (lv) {
                v4f *dp = reinterpret_cast<v4f *>(dH + r * HS + 4 * sub);
                v4f d4 = *dp;
                d4.x += rs * (gv.x - mg - xv.x * mgx); d4.y += rs * (gv.y - mg - xv.y * mgx);
                d4.z += rs * (gv.z - mg - xv.z * mgx); d4.w += rs * (gv.w - mg - xv.w * mgx);
                *dp = d4;
            }
        }
    };
    auto from_tmp = [&](int r, int sub) { return *reinterpret_cast<const v4f *>(tmp + r * HS + 4 * sub); };
    auto ident = [](v4f a, int) { return a; };

    for (int l = L - 1; l >= 0; --l) {
        const float *LPf = P + lay.enc.layer0 + (size_t)l * lay.enc.layer_stride;
        const float *LPb = P + lay.bl[l];
        // ---- layer output norm ---------------------------------------------------------------------
        {
            const float *og = LPf + lay.enc.l_ong;
            ln_bwd(sv.h + (size_t)(l + 1) * sv.rows * D, [&](int r, int sub) {
                const v4f g4 = *reinterpret_cast<const v4f *>(og + 4 * sub);
                v4f d4 = *reinterpret_cast<const v4f *>(dconcat + (row0 + r) * F + l * D + 4 * sub);
                d4.x *= g4.x; d4.y *= g4.y; d4.z *= g4.z; d4.w *= g4.w;
                return d4;
            });
        }
        __syncthreads();
        // ---- feed-forward block: dF = (dH W2) . [f > 0] ------------------------------------------------
        {
            const float *W = LPb + lay.b_w2T;
            const float *fsv = sv.f + ((size_t)l * sv.rows + row0) * DFF;
            run_jobs<D, MT, 4>(DFF / 32, dH, HS, [&](int nt, int, int, float (&b)[D / 2]) { load_frag<D>(W, nt, b); }, ident,
                               [&](int mt, int nt, int r, int h, const f32x16 &acc) {
                                   const int col = nt * 32 + r;
                                   float fv[16];                 // all sixteen requested before the first is used
#pragma unroll
                                   for (int i = 0; i < 16; ++i) {
                                       const int lr = mt * 32 + acc_row(i, h);
                                       fv[i] = fsv[(size_t)(live(lr) ? lr : 0) * DFF + col];
                                   }
#pragma unroll
                                   for (int i = 0; i < 16; ++i) {
                                       const int lr = mt * 32 + acc_row(i, h);
                                       fb[lr * FS + col] = (live(lr) && fv[i] > 0.f) ? acc[i] : 0.f;
                                   }
                               });
        }
        __syncthreads();
        // ---- du2 = dF W1' -> tmp -----------------------------------------------------------------------
        {
            const float *W = LPb + lay.b_w1T;
            run_jobs<DFF, MT, 4>(D / 32, fb, FS, [&](int nt, int, int, float (&b)[DFF / 2]) { load_frag<DFF>(W, nt, b); }, ident,
                                 [&](int mt, int nt, int r, int h, const f32x16 &acc) {
                                     const int col = nt * 32 + r;
#pragma unroll
                                     for (int i = 0; i < 16; ++i) tmp[(mt * 32 + acc_row(i, h)) * HS + col] = acc[i];
                                 });
        }
        __syncthreads();
        ln_bwd(sv.h1 + (size_t)l * sv.rows * D, from_tmp);      // dH is now the gradient wrt h1
        __syncthreads();
        // ---- dO = dH Wo -> tmp; q|k|v of this layer -> qb -------------------------------------------------
        {
            const float *W = LPb + lay.b_woT;
            run_jobs<D, MT, 4>(D / 32, dH, HS, [&](int nt, int, int, float (&b)[D / 2]) { load_frag<D>(W, nt, b); }, ident,
                               [&](int mt, int nt, int r, int h, const f32x16 &acc) {
                                   const int col = nt * 32 + r;
#pragma unroll
                                   for (int i = 0; i < 16; ++i) tmp[(mt * 32 + acc_row(i, h)) * HS + col] = acc[i];
                               });
            const float *qsv = sv.qkv + ((size_t)l * sv.rows + row0) * (3 * D);
            // four quads per thread requested before the first is stored (a dead row reads row 0 and is zeroed)
            constexpr int NQ = RP * (3 * D / 4) / 256;
            static_assert(RP * (3 * D / 4) % 256 == 0 && NQ % 4 == 0, "whole passes of four quads per thread");
            for (int q0 = 0; q0 < NQ; q0 += 4) {
                v4f v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = tid + 256 * (q0 + u), r = i / (3 * D / 4), c4 = i - r * (3 * D / 4);
                    v[u] = *reinterpret_cast<const v4f *>(qsv + (size_t)(live(r) ? r : 0) * (3 * D) + 4 * c4);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = tid + 256 * (q0 + u), r = i / (3 * D / 4), c4 = i - r * (3 * D / 4);
                    *reinterpret_cast<v4f *>(qb + r * QS + 4 * c4) = live(r) ? v[u] : v4f{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        __syncthreads();
        // ---- attention backward: SP lanes per (graph, head, node), each DK / SP of the head's dimensions --------
        // (two heads of 32: one lane per item left 120 of the 256 threads with 96 accumulators each; two lanes per item
        // halve the registers, the LDS reads and the FMAs of a lane, and all four waves work.  The two partial dot
        // products of a score meet through one DPP exchange.)
        {
            constexpr int SP = (2 * G * H * N <= 256 && DK % 8 == 0) ? 2 : 1;
            constexpr int DS = DK / SP;
            const bool act = tid < G * H * N * SP;
            const int it = act ? tid / SP : 0;
            const int eo = (SP == 2 ? (tid & 1) : 0) * DS;          // this lane's first dimension of the head
            const int n = it % N, hd = (it / N) % H, g = it / (N * H);
            const int row = g * N + n;
            const float *Pg = Pm + (g * H + hd) * N * N;
            float dq[DS], dk[DS], dv[DS];
            auto both = [&](float v) { if constexpr (SP == 2) v += dpp_mov<0xB1>(v); return v; };   // quad_perm [1,0,3,2]
            if (act) {
                // as query n: P[n][:] as the forward computes it (up to the order of the score's sum), dS[n][:], dq_n
                // (an item's two lanes are both inside or both outside this branch: the DPP exchange has its partner)
                float q[DS], o[DS];
                load_row<DS>(qb + row * QS + hd * DK + eo, q);
                load_row<DS>(tmp + row * HS + hd * DK + eo, o);
                float s[N], dp[N];
                float mx = -3.0e38f;
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const float *kp = qb + (g * N + j) * QS + D + hd * DK + eo;
                    const float *vp = kp + D;
                    float acc = 0.f, accv = 0.f;
#pragma unroll
                    for (int e = 0; e < DS; e += 4) {
                        const v4f k4 = *reinterpret_cast<const v4f *>(kp + e);
                        const v4f v4 = *reinterpret_cast<const v4f *>(vp + e);
                        acc = fmaf(q[e], k4.x, acc); acc = fmaf(q[e + 1], k4.y, acc);
                        acc = fmaf(q[e + 2], k4.z, acc); acc = fmaf(q[e + 3], k4.w, acc);
                        accv = fmaf(o[e], v4.x, accv); accv = fmaf(o[e + 1], v4.y, accv);
                        accv = fmaf(o[e + 2], v4.z, accv); accv = fmaf(o[e + 3], v4.w, accv);
                    }
                    acc = both(acc);
                    accv = both(accv);
                    acc = acc * (1.0f / sqrtf((float)DK));    // constant-folded reciprocal
                    if (maskb[(g * N + n) * N + j] == 0.f) acc = -1e9f;
                    s[j] = acc;
                    dp[j] = accv;
                    mx = fmaxf(mx, acc);
                    if ((j & 1) == 1) __builtin_amdgcn_sched_barrier(0);     // two keys' rows in flight, not all twenty
                }
                float den = 0.f;
#pragma unroll
                for (int j = 0; j < N; ++j) { s[j] = __expf(s[j] - mx); den += s[j]; }
                float c = 0.f;
#pragma unroll
                for (int j = 0; j < N; ++j) { s[j] = s[j] * (1.0f / den); c = fmaf(s[j], dp[j], c); }
#pragma unroll
                for (int e = 0; e < DS; ++e) dq[e] = 0.f;
                float *Pr = Pm + ((g * H + hd) * N + n) * N;
                float *Sr = Sm + ((g * H + hd) * N + n) * N;
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    // masked_fill (transformer.py:16) passes no gradient to a filled score
                    const float ds = (maskb[(g * N + n) * N + j] == 0.f) ? 0.f : s[j] * (dp[j] - c) * (1.0f / sqrtf((float)DK));
                    if (eo == 0) { Pr[j] = s[j]; Sr[j] = ds; }
                    const float *kp = qb + (g * N + j) * QS + D + hd * DK + eo;
#pragma unroll
                    for (int e = 0; e < DS; e += 4) {
                        const v4f k4 = *reinterpret_cast<const v4f *>(kp + e);
                        dq[e] = fmaf(ds, k4.x, dq[e]); dq[e + 1] = fmaf(ds, k4.y, dq[e + 1]);
                        dq[e + 2] = fmaf(ds, k4.z, dq[e + 2]); dq[e + 3] = fmaf(ds, k4.w, dq[e + 3]);
                    }
                    if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();
            if (act) {
                // as key/value n: dk_n = sum_i dS[i][n] q_i, dv_n = sum_i P[i][n] dO_i
                const float *Sg = Sm + (g * H + hd) * N * N;
#pragma unroll
                for (int e = 0; e < DS; ++e) { dk[e] = 0.f; dv[e] = 0.f; }
                for (int i = 0; i < N; ++i) {
                    const float ds = Sg[i * N + n], p = Pg[i * N + n];
                    const float *qp = qb + (g * N + i) * QS + hd * DK + eo;
                    const float *op = tmp + (g * N + i) * HS + hd * DK + eo;
#pragma unroll
                    for (int e = 0; e < DS; e += 4) {
                        const v4f q4 = *reinterpret_cast<const v4f *>(qp + e);
                        const v4f o4 = *reinterpret_cast<const v4f *>(op + e);
                        dk[e] = fmaf(ds, q4.x, dk[e]); dk[e + 1] = fmaf(ds, q4.y, dk[e + 1]);
                        dk[e + 2] = fmaf(ds, q4.z, dk[e + 2]); dk[e + 3] = fmaf(ds, q4.w, dk[e + 3]);
                        dv[e] = fmaf(p, o4.x, dv[e]); dv[e + 1] = fmaf(p, o4.y, dv[e + 1]);
                        dv[e + 2] = fmaf(p, o4.z, dv[e + 2]); dv[e + 3] = fmaf(p, o4.w, dv[e + 3]);
                    }
                }
            }
            __syncthreads();
            if (act) {
                float *qp = qb + row * QS + hd * DK + eo;
#pragma unroll
                for (int e = 0; e < DS; e += 4) {
                    *reinterpret_cast<v4f *>(qp + e) = v4f{dq[e], dq[e + 1], dq[e + 2], dq[e + 3]};
                    *reinterpret_cast<v4f *>(qp + D + e) = v4f{dk[e], dk[e + 1], dk[e + 2], dk[e + 3]};
                    *reinterpret_cast<v4f *>(qp + 2 * D + e) = v4f{dv[e], dv[e + 1], dv[e + 2], dv[e + 3]};
                }
            }
        }
        __syncthreads();
        // ---- du1 = dqkv Wqkv' -> tmp ---------------------------------------------------------------------
        {
            const float *W = LPb + lay.b_wqkvT;
            run_jobs<3 * D, MT, 4>(D / 32, qb, QS, [&](int nt, int, int, float (&b)[3 * D / 2]) { load_frag<3 * D>(W, nt, b); }, ident,
                                   [&](int mt, int nt, int r, int h, const f32x16 &acc) {
                                       const int col = nt * 32 + r;
#pragma unroll
                                       for (int i = 0; i < 16; ++i) tmp[(mt * 32 + acc_row(i, h)) * HS + col] = acc[i];
                                   });
        }
        __syncthreads();
        ln_bwd(sv.h + (size_t)l * sv.rows * D, from_tmp);       // dH is now the gradient wrt the layer input
        __syncthreads();
    }
    // ---- embedding: emb = (x We^T + be) * 8 + temb  ->  dx = 8 dH We ------------------------------------
    const float *We = P + lay.enc.emb_w;
    for (int i = tid; i < R * V; i += 256) {
        const int r = i / V, v = i - r * V;
        if (!live(r)) continue;
        float acc = 0.f;
#pragma unroll 8
        for (int c = 0; c < D; ++c) acc = fmaf(dH[r * HS + c], We[c * V + v], acc);
        grad[(row0 + r) * V + v] = acc * 8.0f;
    }
}

// dh2[r][c] = (sum_v dflat[g][n V + v] W3[v][c]) * elu'(z) with h2 = elu(z) stored in `h2` and replaced
// in place by the gradient.  W3 = final.linears.2 [V][K]; dflat rows have stride ldf; rows = batch*N.
template <int N, int V>
__global__ __launch_bounds__(256) void final2_bwd_kernel(float *h2, const float *__restrict__ dflat, int ldf,
                                                         const float *__restrict__ W3, int K, size_t rows) {
    const int tid = threadIdx.x;
    for (int c = tid; c < K; c += 256) {
        float w[V];
#pragma unroll
        for (int v = 0; v < V; ++v) w[v] = W3[(size_t)v * K + c];
        for (size_t r = blockIdx.x; r < rows; r += gridDim.x) {
            const size_t g = r / N;
            const int n = (int)(r - g * N);
            const float *d = dflat + g * ldf + n * V;
            float acc = 0.f;
#pragma unroll
            for (int v = 0; v < V; ++v) acc = fmaf(d[v], w[v], acc);
            const float a = h2[r * K + c];
            h2[r * K + c] = acc * (a > 0.f ? 1.0f : a + 1.0f);
        }
    }
}

// c0 = b0 + W0[:, :nz] D_mu : the dataset half of pred_fc.0 (cate.py:342-348), constant per call.
__global__ void predfc0_bias_kernel(const float *__restrict__ wd, const float *__restrict__ b0, const float *__restrict__ d_mu,
                                    int hs, int nz, float *__restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= hs) return;
    float acc = 0.f;
    for (int z = 0; z < nz; ++z) acc = fmaf(wd[(size_t)j * nz + z], d_mu[z], acc);
    out[j] = acc + b0[j];
}

// y[b] = w2 . h[b] + b2 (pred_fc.2) and, in place, h[b] <- dz = w2 (1 - h^2): the gradient of sum_b y_b
// wrt the pre-activation of the tanh layer.  One block per graph.
__global__ __launch_bounds__(256) void predfc2_kernel(float *h, int hs, const float *__restrict__ w2, const float *__restrict__ b2,
                                                      float *__restrict__ y, int with_grad) {
    __shared__ float sh[256];
    const int tid = threadIdx.x;
    float *row = h + (size_t)blockIdx.x * hs;
    float acc = 0.f;
    for (int j = tid; j < hs; j += 256) {
        const float t = row[j], w = w2[j];
        acc = fmaf(t, w, acc);
        if (with_grad) row[j] = w * (1.0f - t * t);
    }
    sh[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) sh[tid] = sh[tid] + sh[tid + s];
        __syncthreads();
    }
    if (tid == 0 && y) y[blockIdx.x] = sh[0] + b2[0];
}

// Graph-transformer encoder of the CATE score network, one workgroup per G graphs, activations
// resident in LDS across all layers.  Included by dnag_lib.hip.
//
// Restates (reference NAS-Bench-201/): models/cate.py:145-160 (embedding), models/transformer.py:166-179
// (Encoder), :127-136 (EncoderBlock), :69-80 (SelfAttentionBlock), :23-42 (MultiHeadAttention),
// :12-20 (attention, mask fill -1e9), :97-108 (FeedForwardBlock), models/cate.py:83-86 (concat).
#pragma once
#include "dnag_common.h"

// ---------------------------------------------------------------------------------------------
// Tile jobs: out[32 x 32] = A[32 x K] * W[32 x K]^T on v_mfma_f32_32x32x2_f32.
//   A lives in LDS (row stride `as` floats, as % 64 == 4 or 20 keeps ds_read_b128 conflict free),
//   W rows are K-contiguous in global memory.  The K index is permuted between A and B alike:
//   lane half h = lane>>5 owns k in [h*K/2, (h+1)*K/2), so every lane reads contiguous floats.
// Jobs (nt, mt) are dealt in contiguous chunks to the NW waves; B fragments stay in registers
// while nt does not change.
// ---------------------------------------------------------------------------------------------
template <int K, int MT, int NW, typename BLoad, typename AXf, typename Epi>
__device__ __forceinline__ void run_jobs(int ntiles, const float *A, int as, BLoad bload, AXf axf, Epi epi) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int njobs = ntiles * MT;
    const int per = (njobs + NW - 1) / NW;
    const int j0 = wave * per;
    const int j1 = (j0 + per < njobs) ? j0 + per : njobs;
    float b[K / 2];
    int cur_nt = -1;
    for (int j = j0; j < j1; ++j) {
        const int nt = j / MT, mt = j - nt * MT;
        if (nt != cur_nt) {
            bload(nt, r, h, b);
            cur_nt = nt;
        }
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        const float *ap = A + (size_t)(mt * 32 + r) * as + h * (K / 2);
#pragma unroll
        for (int q = 0; q < K / 8; ++q) {
            v4f a4 = *reinterpret_cast<const v4f *>(ap + 4 * q);
            a4 = axf(a4, mt * 32 + r);
            acc = mfma32(a4.x, b[4 * q + 0], acc);
            acc = mfma32(a4.y, b[4 * q + 1], acc);
            acc = mfma32(a4.z, b[4 * q + 2], acc);
            acc = mfma32(a4.w, b[4 * q + 3], acc);
        }
        epi(mt, nt, r, h, acc);
    }
}

// load K/2 contiguous floats (16-byte aligned) into registers
template <int KH>
__device__ __forceinline__ void load_row(const float *p, float (&b)[KH]) {
#pragma unroll
    for (int i = 0; i < KH / 4; ++i) {
        const v4f v = *reinterpret_cast<const v4f *>(p + 4 * i);
        b[4 * i + 0] = v.x; b[4 * i + 1] = v.y; b[4 * i + 2] = v.z; b[4 * i + 3] = v.w;
    }
}
template <int KH>
__device__ __forceinline__ void zero_row(float (&b)[KH]) {
#pragma unroll
    for (int i = 0; i < KH; ++i) b[i] = 0.f;
}

// LayerNorm statistics of RP rows of width 64 (row stride HS): 16 lanes per row.
template <int RP, int HS, int NT>
__device__ __forceinline__ void ln_stats(const float *hbuf, float *mean, float *rstd) {
    const int sub = threadIdx.x & 15, slot = threadIdx.x >> 4;
    for (int r = slot; r < RP; r += NT / 16) {
        const v4f v = *reinterpret_cast<const v4f *>(hbuf + r * HS + 4 * sub);
        float s = (v.x + v.y) + (v.z + v.w);
        s += __shfl_xor(s, 8, 16); s += __shfl_xor(s, 4, 16); s += __shfl_xor(s, 2, 16); s += __shfl_xor(s, 1, 16);
        const float m = s * (1.0f / 64.0f);
        const float dx = v.x - m, dy = v.y - m, dz = v.z - m, dw = v.w - m;
        float q = (dx * dx + dy * dy) + (dz * dz + dw * dw);
        q += __shfl_xor(q, 8, 16); q += __shfl_xor(q, 4, 16); q += __shfl_xor(q, 2, 16); q += __shfl_xor(q, 1, 16);
        if (sub == 0) {
            mean[r] = m;
            rstd[r] = 1.0f / sqrtf(q * (1.0f / 64.0f) + 1e-5f);
        }
    }
}

// LDS floats needed by encoder_kernel<N,V,G,H,DFF>
template <int N, int G, int DFF>
struct EncGeom {
    static constexpr int R = G * N;
    static constexpr int MT = (R + 31) / 32;
    static constexpr int RP = MT * 32;
    static constexpr int HS = DNAG_D + 4;
    static constexpr int QW = (3 * DNAG_D > DFF) ? 3 * DNAG_D : DFF;
    static constexpr int QS = QW + 4;
    static constexpr size_t lds_floats = (size_t)RP * HS + (size_t)RP * QS + 2 * RP + (size_t)G * N * N;
};

// x [batch][N][V]; temb [tstride ? batch : 1][64]; mask [mstride ? batch : 1][N][N];
// concat [rows_pad][L*64] (row = graph*N + node); attn0 [batch][H][N][N] or null.
template <int N, int V, int G, int H, int DFF, bool WITH_PE>
__global__ __launch_bounds__(256, 2) void encoder_kernel(const float *__restrict__ P, const CateLayout lay,
                                                         const float *__restrict__ x, const float *__restrict__ temb,
                                                         int tstride, const float *__restrict__ mask, int mstride,
                                                         float *__restrict__ concat, float *__restrict__ attn0, int batch) {
    typedef EncGeom<N, G, DFF> Ge;
    constexpr int R = Ge::R, MT = Ge::MT, RP = Ge::RP, HS = Ge::HS, QS = Ge::QS;
    constexpr int D = DNAG_D, DK = D / H;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *hbuf = smem;                 // [RP][HS] residual stream
    float *qkv = hbuf + RP * HS;        // [RP][QS] q|k|v, then attention output in the q slot, then FFN hidden
    float *mean = qkv + RP * QS;        // [RP]
    float *rstd = mean + RP;            // [RP]
    float *maskb = rstd + RP;           // [G][N][N]

    const int tid = threadIdx.x;
    const int g0 = blockIdx.x * G;
    const int L = lay.L;

    // ---- stage inputs -------------------------------------------------------------------------
    float *xin = qkv;                   // [R][V] alias
    for (int i = tid; i < R * V; i += 256) {
        const int g = i / (N * V);
        xin[i] = (g0 + g < batch) ? x[(size_t)g0 * N * V + i] : 0.f;
    }
    for (int i = tid; i < G * N * N; i += 256) {
        const int g = i / (N * N), ij = i - g * N * N;
        const int gg = (g0 + g < batch) ? g0 + g : 0;
        maskb[i] = mask[(size_t)(mstride ? gg : 0) * N * N + ij];
    }
    __syncthreads();
    // ---- embedding: (x We^T + be) * sqrt(d) + PE + temb   (transformer.py:230, cate.py:149-159) ----
    {
        const float *We = P + lay.emb_w, *be = P + lay.emb_b, *pe = P + lay.pe;
        for (int i = tid; i < RP * D; i += 256) {
            const int r = i / D, c = i - r * D;
            float val = 0.f;
            if (r < R) {
                const int g = r / N, n = r - g * N;
                if (g0 + g < batch) {
                    float acc = 0.f;
#pragma unroll
                    for (int v = 0; v < V; ++v) acc = fmaf(xin[r * V + v], We[c * V + v], acc);
                    val = (acc + be[c]) * 8.0f;
                    if (WITH_PE) val = val + pe[n * D + c];
                    val = val + temb[(size_t)(tstride ? g0 + g : 0) * D + c];
                }
            }
            hbuf[r * HS + c] = val;
        }
    }
    __syncthreads();
    ln_stats<RP, HS, 256>(hbuf, mean, rstd);
    __syncthreads();

    auto ln_xf = [&](v4f a, int row) {
        const float m = mean[row], s = rstd[row];
        a.x = (a.x - m) * s; a.y = (a.y - m) * s; a.z = (a.z - m) * s; a.w = (a.w - m) * s;
        return a;
    };
    auto id_xf = [](v4f a, int) { return a; };

    for (int l = 0; l < L; ++l) {
        const float *LP = P + lay.layer0 + (size_t)l * lay.layer_stride;
        // ---- QKV = LN1(h) Wqkv'^T + bqkv' --------------------------------------------------------
        {
            const float *W = LP + lay.l_wqkv, *bias = LP + lay.l_bqkv;
            run_jobs<D, MT, 4>(
                3 * D / 32, hbuf, HS,
                [&](int nt, int r, int h, float (&b)[D / 2]) { load_row<D / 2>(W + (size_t)(nt * 32 + r) * D + h * (D / 2), b); },
                ln_xf,
                [&](int mt, int nt, int r, int h, const f32x16 &acc) {
                    const int col = nt * 32 + r;
                    const float bv = bias[col];
#pragma unroll
                    for (int i = 0; i < 16; ++i) qkv[(mt * 32 + acc_row(i, h)) * QS + col] = acc[i] + bv;
                });
        }
        __syncthreads();
        // ---- attention per (graph, head, query) ---------------------------------------------------
        for (int it = tid; it < G * H * N; it += 256) {
            const int i = it % N, hd = (it / N) % H, g = it / (N * H);
            const int row = g * N + i;
            float q[DK];
            {
                const float *qp = qkv + row * QS + hd * DK;
#pragma unroll
                for (int e = 0; e < DK; e += 4) {
                    const v4f v4 = *reinterpret_cast<const v4f *>(qp + e);
                    q[e] = v4.x; q[e + 1] = v4.y; q[e + 2] = v4.z; q[e + 3] = v4.w;
                }
            }
            float s[N];
            float mx = -3.0e38f;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const float *kp = qkv + (g * N + j) * QS + D + hd * DK;
                float acc = 0.f;
#pragma unroll
                for (int e = 0; e < DK; e += 4) {
                    const v4f v4 = *reinterpret_cast<const v4f *>(kp + e);
                    acc = fmaf(q[e], v4.x, acc); acc = fmaf(q[e + 1], v4.y, acc);
                    acc = fmaf(q[e + 2], v4.z, acc); acc = fmaf(q[e + 3], v4.w, acc);
                }
                acc = acc / sqrtf((float)DK);
                if (maskb[(g * N + i) * N + j] == 0.f) acc = -1e9f;
                s[j] = acc;
                mx = fmaxf(mx, acc);
            }
            float den = 0.f;
#pragma unroll
            for (int j = 0; j < N; ++j) { s[j] = expf(s[j] - mx); den += s[j]; }
            float o[DK];
#pragma unroll
            for (int e = 0; e < DK; ++e) o[e] = 0.f;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const float p = s[j] / den;
                if (attn0 != nullptr && l == 0 && g0 + g < batch)
                    attn0[(((size_t)(g0 + g) * H + hd) * N + i) * N + j] = p;
                const float *vp = qkv + (g * N + j) * QS + 2 * D + hd * DK;
#pragma unroll
                for (int e = 0; e < DK; e += 4) {
                    const v4f v4 = *reinterpret_cast<const v4f *>(vp + e);
                    o[e] = fmaf(p, v4.x, o[e]); o[e + 1] = fmaf(p, v4.y, o[e + 1]);
                    o[e + 2] = fmaf(p, v4.z, o[e + 2]); o[e + 3] = fmaf(p, v4.w, o[e + 3]);
                }
            }
            float *op = qkv + row * QS + hd * DK;   // overwrite this item's own q slot
#pragma unroll
            for (int e = 0; e < DK; e += 4) *reinterpret_cast<v4f *>(op + e) = v4f{o[e], o[e + 1], o[e + 2], o[e + 3]};
        }
        __syncthreads();
        // ---- h += attn Wo^T + bo ------------------------------------------------------------------
        {
            const float *W = LP + lay.l_wo, *bias = LP + lay.l_bo;
            run_jobs<D, MT, 4>(
                D / 32, qkv, QS,
                [&](int nt, int r, int h, float (&b)[D / 2]) { load_row<D / 2>(W + (size_t)(nt * 32 + r) * D + h * (D / 2), b); },
                id_xf,
                [&](int mt, int nt, int r, int h, const f32x16 &acc) {
                    const int col = nt * 32 + r;
                    const float bv = bias[col];
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        float *hp = hbuf + (mt * 32 + acc_row(i, h)) * HS + col;
                        *hp = (acc[i] + bv) + *hp;
                    }
                });
        }
        __syncthreads();
        ln_stats<RP, HS, 256>(hbuf, mean, rstd);
        __syncthreads();
        // ---- hidden = relu(LN2(h) W1'^T + b1') -----------------------------------------------------
        {
            const float *W = LP + lay.l_w1, *bias = LP + lay.l_b1;
            run_jobs<D, MT, 4>(
                DFF / 32, hbuf, HS,
                [&](int nt, int r, int h, float (&b)[D / 2]) { load_row<D / 2>(W + (size_t)(nt * 32 + r) * D + h * (D / 2), b); },
                ln_xf,
                [&](int mt, int nt, int r, int h, const f32x16 &acc) {
                    const int col = nt * 32 + r;
                    const float bv = bias[col];
#pragma unroll
                    for (int i = 0; i < 16; ++i) qkv[(mt * 32 + acc_row(i, h)) * QS + col] = fmaxf(acc[i] + bv, 0.f);
                });
        }
        __syncthreads();
        // ---- h += hidden W2^T + b2 ----------------------------------------------------------------
        {
            const float *W = LP + lay.l_w2, *bias = LP + lay.l_b2;
            run_jobs<DFF, MT, 4>(
                D / 32, qkv, QS,
                [&](int nt, int r, int h, float (&b)[DFF / 2]) { load_row<DFF / 2>(W + (size_t)(nt * 32 + r) * DFF + h * (DFF / 2), b); },
                id_xf,
                [&](int mt, int nt, int r, int h, const f32x16 &acc) {
                    const int col = nt * 32 + r;
                    const float bv = bias[col];
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        float *hp = hbuf + (mt * 32 + acc_row(i, h)) * HS + col;
                        *hp = (acc[i] + bv) + *hp;
                    }
                });
        }
        __syncthreads();
        // ---- layer output norm (transformer.py:177) -> concat slice; same stats feed the next LN1 --
        ln_stats<RP, HS, 256>(hbuf, mean, rstd);
        __syncthreads();
        {
            const float *og = LP + lay.l_ong, *ob = LP + lay.l_onb;
            const int sub = tid & 15, slot = tid >> 4;
            const v4f g4 = *reinterpret_cast<const v4f *>(og + 4 * sub);
            const v4f b4 = *reinterpret_cast<const v4f *>(ob + 4 * sub);
            for (int r = slot; r < R; r += 16) {
                const int g = r / N;
                if (g0 + g >= batch) continue;
                const v4f v = *reinterpret_cast<const v4f *>(hbuf + r * HS + 4 * sub);
                const float m = mean[r], s = rstd[r];
                v4f o4;
                o4.x = (v.x - m) * s * g4.x + b4.x; o4.y = (v.y - m) * s * g4.y + b4.y;
                o4.z = (v.z - m) * s * g4.z + b4.z; o4.w = (v.w - m) * s * g4.w + b4.w;
                *reinterpret_cast<v4f *>(concat + ((size_t)g0 * N + r) * lay.F + l * D + 4 * sub) = o4;
            }
        }
        // no barrier needed: the next phase only reads hbuf/mean/rstd and writes qkv
    }
}

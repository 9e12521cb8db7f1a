// Graph-transformer encoder of the CATE score network, one workgroup per G graphs, activations
// resident in LDS across all layers.  Included by dnag_lib.hip.
//
// Restates (reference NAS-Bench-201/): models/cate.py:145-160 (embedding), models/transformer.py:166-179
// (Encoder), :127-136 (EncoderBlock), :69-80 (SelfAttentionBlock), :23-42 (MultiHeadAttention),
// :12-20 (attention, mask fill -1e9), :97-108 (FeedForwardBlock), models/cate.py:83-86 (concat).
#pragma once
#include "dnag_common.h"

#ifndef DNAG_ENC_SKIP
#define DNAG_ENC_SKIP 0     // debug builds only (-DDNAG_ENC_SKIP=<mask>): phases switched off for timing ablations (wrong results)
#endif

// ---------------------------------------------------------------------------------------------
// Tile jobs: out[32 x 32] = A[32 x K] * W[32 x K]^T on v_mfma_f32_32x32x2_f32.
//   A lives in LDS (row stride `as` floats, as % 64 == 4 or 20 keeps ds_read_b128 conflict free),
//   W rows are K-contiguous in global memory.  The K index is permuted between A and B alike:
//   lane half h = lane>>5 owns k in [h*K/2, (h+1)*K/2), so every lane reads contiguous floats.
// Jobs (nt, mt) are dealt in contiguous chunks to the NW waves; B fragments stay in registers
// while nt does not change.
// ---------------------------------------------------------------------------------------------
// F16: split-arithmetic MFMAs (dnag_gemm16.hip.h): `bload` fills the B registers with frag16_pack fragments (fp16
// hi/lo of 256 W); every A value is scaled by the power of two `sa`, split, and three v_mfma_f32_32x32x16_f16 per
// 8 k values replace eight v_mfma_f32_32x32x2_f32; the accumulator is scaled back before the epilogue.
template <int K, int MT, int NW, bool F16 = false, typename BLoad, typename AXf, typename Epi>
__device__ __forceinline__ void run_jobs(int ntiles, const float *A, int as, BLoad bload, AXf axf, Epi epi, float sa = 16.0f,
                                         float winv = 1.0f / 256.0f) {
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
        if constexpr (F16) {
            static_assert(K % 16 == 0, "8 k values per lane half and step");
#pragma unroll
            for (int s = 0; s < K / 16; ++s) {
                v4f a0 = *reinterpret_cast<const v4f *>(ap + 8 * s);
                v4f a1 = *reinterpret_cast<const v4f *>(ap + 8 * s + 4);
                a0 = axf(a0, mt * 32 + r);
                a1 = axf(a1, mt * 32 + r);
                h8 ahi, alo;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = (e < 4 ? a0[e] : a1[e - 4]) * sa;
                    const _Float16 hi = (_Float16)v;
                    ahi[e] = hi;
                    alo[e] = (_Float16)(v - (float)hi);
                }
                const h8 bhi = __builtin_bit_cast(h8, v4f{b[8 * s], b[8 * s + 1], b[8 * s + 2], b[8 * s + 3]});
                const h8 blo = __builtin_bit_cast(h8, v4f{b[8 * s + 4], b[8 * s + 5], b[8 * s + 6], b[8 * s + 7]});
                acc = mfma16(ahi, bhi, acc);
                acc = mfma16(ahi, blo, acc);
                acc = mfma16(alo, bhi, acc);
            }
            const float inv = winv / sa;          // sa, 1 / winv: the power-of-two scales of A and of the weight fragments
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] *= inv;
        } else {
#pragma unroll
            for (int q = 0; q < K / 8; ++q) {
                v4f a4 = *reinterpret_cast<const v4f *>(ap + 4 * q);
                a4 = axf(a4, mt * 32 + r);
                acc = mfma32(a4.x, b[4 * q + 0], acc);
                acc = mfma32(a4.y, b[4 * q + 1], acc);
                acc = mfma32(a4.z, b[4 * q + 2], acc);
                acc = mfma32(a4.w, b[4 * q + 3], acc);
            }
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
// B fragments of column tile `nt` from a fragment-major matrix (dnag_common.h: frag_index): every
// load instruction of the wave reads one contiguous KiB.
template <int K>
__device__ __forceinline__ void load_frag(const float *F, int nt, float (&b)[K / 2]) {
    // explicit global address space: a pointer that went through the opaque asm of the prefetch lambdas would otherwise
    // be loaded with flat instructions, which count on lgkmcnt too, so every LDS wait would also wait for the prefetch
    typedef const __attribute__((address_space(1))) v4f *gv4p;
    const float *p = F + ((size_t)nt * (K / 8) * 64 + (threadIdx.x & 63)) * 4;
#pragma unroll
    for (int q = 0; q < K / 8; ++q) {
        const v4f v = *(gv4p)(p + (size_t)q * 256);
        b[4 * q + 0] = v.x; b[4 * q + 1] = v.y; b[4 * q + 2] = v.z; b[4 * q + 3] = v.w;
    }
}
template <int KH>
__device__ __forceinline__ void zero_row(float (&b)[KH]) {
#pragma unroll
    for (int i = 0; i < KH; ++i) b[i] = 0.f;
}

// ---------------------------------------------------------------------------------------------
// DPP row reduction: total over the 16 lanes of a row, delivered to every lane of that row.
// ---------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);   // row_half_mirror
    v += dpp_mov<0x140>(v);   // row_mirror
    return v;
}

// ---------------------------------------------------------------------------------------------
// Tile jobs, second form: the B fragments of the wave's first tile are already in registers
// (prefetched by the caller while the previous phase ran); a second tile, if the wave's job chunk
// spans one, is loaded into b1 at entry.  LN = true normalises every A row on load:
// (a - mean) * rstd with the statistics of the 64-wide row (this lane's 32 values + lane^32's),
// i.e. LayerNorm without affine (gamma/beta are folded into W', b' at pack time).
// ---------------------------------------------------------------------------------------------
template <int NTILES, int MT, int NW>
struct JobSplit {
    static constexpr int njobs = NTILES * MT;
    static constexpr int per = (njobs + NW - 1) / NW;
    static constexpr int max_tiles() {
        int m = 0;
        for (int w = 0; w < NW; ++w) {
            const int j0 = w * per, j1 = (j0 + per < njobs) ? j0 + per : njobs;
            if (j0 < j1) { const int t = (j1 - 1) / MT - j0 / MT + 1; m = t > m ? t : m; }
        }
        return m;
    }
    __device__ static __forceinline__ int first_tile(int wave) { return wave * per < njobs ? (wave * per) / MT : -1; }
};

// F16: split-arithmetic MFMAs (dnag_gemm16.hip.h): the B registers hold frag16_pack fragments (fp16 hi/lo of sw W),
// every A value is scaled by `sa` and split after the LayerNorm, three v_mfma_f32_32x32x16_f16 per 8 k values replace
// eight v_mfma_f32_32x32x2_f32, and the accumulator is scaled back by inv = 1 / (sa sw) before the epilogue (the
// packer's scale table of the layer, dnag_common.h ENC_SC_*).
template <int K, int NTILES, int MT, int NW, bool LN, bool F16 = false, typename BLoad, typename Epi>
__device__ __forceinline__ void run_jobs2(const float *A, int as, float (&b0)[K / 2], float (&b1)[K / 2], BLoad bload, Epi epi,
                                          float sa = 16.0f, float inv = 1.0f / 4096.0f) {
    typedef JobSplit<NTILES, MT, NW> JS;
    static_assert(JS::max_tiles() <= 2, "a wave's job chunk may span at most two column tiles");
    static_assert(!LN || K == DNAG_D, "fused LayerNorm needs the whole row in one K pass");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int j0 = wave * JS::per;
    const int j1 = (j0 + JS::per < JS::njobs) ? j0 + JS::per : JS::njobs;
    if (j0 >= j1) return;
    const int nt0 = j0 / MT;
    if constexpr (JS::max_tiles() == 2) {
        const int ntl = (j1 - 1) / MT;
        if (ntl != nt0) bload(ntl, r, h, b1);
    }
    for (int j = j0; j < j1; ++j) {
        const int nt = j / MT, mt = j - nt * MT;
        constexpr int CH = (K / 2 > 32) ? 32 : K / 2;       // A is consumed in chunks of <= 32 floats per lane
        const float *ap = A + (size_t)(mt * 32 + r) * as + h * (K / 2);
        const bool first = (JS::max_tiles() == 1) || nt == nt0;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int c = 0; c < K / 2; c += CH) {
            float a[CH];
            load_row<CH>(ap + c, a);
            if constexpr (LN) {
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < CH; ++i) s += a[i];
                s += __shfl_xor(s, 32);
                const float m = s * (1.0f / K);
                float q = 0.f;
#pragma unroll
                for (int i = 0; i < CH; ++i) { a[i] -= m; q = fmaf(a[i], a[i], q); }
                q += __shfl_xor(q, 32);
                const float rs = 1.0f / sqrtf(q * (1.0f / K) + 1e-5f);
#pragma unroll
                for (int i = 0; i < CH; ++i) a[i] *= rs;
            }
            if constexpr (F16) {
#pragma unroll
                for (int s = 0; s < CH / 8; ++s) {
                    h8 ahi, alo;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float v = a[8 * s + j] * sa;
                        const _Float16 hi = (_Float16)v;
                        ahi[j] = hi;
                        alo[j] = (_Float16)(v - (float)hi);
                    }
                    const int o = c + 8 * s;
                    const h8 bhi = first ? __builtin_bit_cast(h8, v4f{b0[o], b0[o + 1], b0[o + 2], b0[o + 3]})
                                         : __builtin_bit_cast(h8, v4f{b1[o], b1[o + 1], b1[o + 2], b1[o + 3]});
                    const h8 blo = first ? __builtin_bit_cast(h8, v4f{b0[o + 4], b0[o + 5], b0[o + 6], b0[o + 7]})
                                         : __builtin_bit_cast(h8, v4f{b1[o + 4], b1[o + 5], b1[o + 6], b1[o + 7]});
                    acc = mfma16(ahi, bhi, acc);
                    acc = mfma16(ahi, blo, acc);
                    acc = mfma16(alo, bhi, acc);
                }
            } else if (first) {
#pragma unroll
                for (int i = 0; i < CH; ++i) acc = mfma32(a[i], b0[c + i], acc);
            } else {
#pragma unroll
                for (int i = 0; i < CH; ++i) acc = mfma32(a[i], b1[c + i], acc);
            }
        }
        if constexpr (F16) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] *= inv;
        }
        epi(mt, nt, r, h, acc);
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
    static constexpr size_t lds_floats = (size_t)RP * HS + (size_t)RP * QS + (size_t)G * N * N;
};

// Activations a backward pass needs (SAVE = true: the surrogate's encoder, dnag_predcate.hip.h), row =
// graph*N + node, `rows` rows per layer slot:
//   h [L+1][rows][64] residual stream entering layer l (slot L: leaving the last layer),
//   h1 [L][rows][64] after the attention block, qkv [L][rows][192] projections incl. bias,
//   f [L][rows][DFF] FFN hidden after relu.
struct EncSave {
    float *h, *h1, *qkv, *f;
    size_t rows;
    float *amax_concat;     // optional (encoder8_kernel's training modes, encoder_kernel with SAVE and fp32 concat rows): max |concat| as float bits, atomicMax; the caller clears it
};

// Dropout keep masks of a training-mode forward (models/cate.py:147, transformer.py:19, :54, :80, :108):
// one byte per site element, 1 = keep.  Layout, rows = batch*N, row = graph*N + node:
//   emb [rows][64], then per layer l a slab [rows][per_layer] with per_layer = H*N + 64 + DFF + 64 and
//   a row holding  attention probabilities [H][N] (keys of this query) | self-attention block output
//   [64] | FFN hidden [DFF] | FFN output [64].
// A dropped value is 0, a kept one v / keep_prob (keep_prob = 1 - p), as nn.Dropout computes it.
struct DropMasks {
    const uint8_t *keep;   // null: dropout off
    float keep_prob;
    size_t rows;
    int per_layer;
};
__device__ __forceinline__ float drop_apply(float v, uint8_t keep, float keep_prob) { return __fdiv_rn(v * (float)keep, keep_prob); }

// x [batch][N][V]; temb [tstride ? batch : 1][64]; mask [mstride ? batch : 1][N][N];
// concat [rows_pad][L*64] (row = graph*N + node); attn0 [batch][H][N][N] or null.
// SPLIT: concat rows are written as k-tile-blocked split-pair fp16 rows (dnag_gemm16.hip.h) for the fp16-MFMA head GEMM.
// F16: the four linears of every layer run on the split-arithmetic fp16 MFMAs (weights from the frag16 copy at delta16).
template <int N, int V, int G, int H, int DFF, bool WITH_PE, bool SAVE = false, bool DROP = false, bool SPLIT = false, bool F16 = false>
__global__ __launch_bounds__(256, 2) void encoder_kernel(const float *__restrict__ P, const CateLayout lay,
                                                         const float *__restrict__ x, const float *__restrict__ temb,
                                                         int tstride, const float *__restrict__ mask, int mstride,
                                                         float *__restrict__ concat, float *__restrict__ attn0, int batch,
                                                         const EncSave sv, const DropMasks dm) {
    typedef EncGeom<N, G, DFF> Ge;
    constexpr int R = Ge::R, MT = Ge::MT, RP = Ge::RP, HS = Ge::HS, QS = Ge::QS;
    constexpr int D = DNAG_D, DK = D / H;
    constexpr int NT_QKV = 3 * D / 32, NT_O = D / 32, NT_F1 = DFF / 32, NT_F2 = D / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *hbuf = smem;                 // [RP][HS] residual stream
    float *qkv = hbuf + RP * HS;        // [RP][QS] q|k|v, then attention output in the q slot, then FFN hidden
    float *maskb = qkv + RP * QS;       // [G][N][N]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g0 = blockIdx.x * G;
    const int L = lay.L;
    const float sa_concat = SPLIT ? P[lay.hsc + HEAD_SC_SA_CONCAT] : 0.f;
    float cmax = 0.f;            // SAVE: largest |concat| value this lane wrote (EncSave::amax_concat)
    const size_t rows_pad = ((size_t)batch * N + 255) / 256 * 256;      // SPLIT: the concat is k-tile blocked (dnag_gemm16.hip.h kblock_at)
#ifdef DNAG_ENC_STAMPS
    long long ts_[96];
    int nts_ = 0;
#define ENC_STAMP() do { if (nts_ < 96) ts_[nts_++] = clock64(); } while (0)
    ENC_STAMP();
#else
#define ENC_STAMP()
#endif

    // weight tiles kept in registers across phases: bP/bQ (K = 64) and bR (K = DFF)
    float bP[D / 2], bQ[D / 2], bR[DFF / 2];
    // The weights are const __restrict__, so the compiler may hoist these loads above barriers and then
    // spill them; passing the address through an opaque asm pins the issue point of every prefetch.
    auto load_w64 = [&](const float *W, int nt, float (&b)[D / 2]) {
        asm volatile("" : "+s"(W));
        load_frag<D>(W, nt, b);
    };
    const int ft_qkv = JobSplit<NT_QKV, MT, 4>::first_tile(wave);
    const int ft_o = JobSplit<NT_O, MT, 4>::first_tile(wave);
    const int ft_f1 = JobSplit<NT_F1, MT, 4>::first_tile(wave);
    const int ft_f2 = JobSplit<NT_F2, MT, 4>::first_tile(wave);
    const size_t wofs = F16 ? lay.delta16 : 0;     // weight matrices come from the fp16 fragment copy
    if (ft_qkv >= 0) load_w64(P + lay.layer0 + wofs + lay.l_wqkv, ft_qkv, bP);

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
    ENC_STAMP();
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
                    if constexpr (DROP) val = drop_apply(val, dm.keep[((size_t)g0 * N + r) * D + c], dm.keep_prob);   // cate.py:147
                    if (WITH_PE) val = val + pe[n * D + c];
                    val = val + temb[(size_t)(tstride ? g0 + g : 0) * D + c];
                }
            }
            hbuf[r * HS + c] = val;
            if constexpr (SAVE) if (r < R && g0 + r / N < batch) sv.h[((size_t)g0 * N + r) * D + c] = val;
        }
    }
    __syncthreads();
    ENC_STAMP();
    if (DNAG_ENC_SKIP & 128) return;
    // row `lr` of this workgroup is a real graph row (SAVE stores skip the padding)
    auto live = [&](int lr) { return lr < R && g0 + lr / N < batch; };

    // co-resident workgroups swap a raised wave priority every layer (dnag_encoder8.hip.h: the arbiter serves the older
    // wave first, and the launch lasts as long as the starved generation)
    const int gen = (blockIdx.x >> 8) & 1;
    for (int l = 0; l < L; ++l) {
#ifndef DNAG_ENC_NO_PRIO
        if ((l ^ gen) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#endif
        const float *LP = P + lay.layer0 + (size_t)l * lay.layer_stride;
        const float *SC = LP + lay.l_sc;          // power-of-two operand scales of this layer (dnag_common.h ENC_SC_*)
        // keep masks of this layer's sites for local row lr (DROP only; dead rows never read them)
        const uint8_t *KL = nullptr;
        if constexpr (DROP) KL = dm.keep + dm.rows * D + ((size_t)l * dm.rows + (size_t)g0 * N) * dm.per_layer;
        // ---- QKV = LN1(h) Wqkv'^T + bqkv' --------------------------------------------------------
        if (!(DNAG_ENC_SKIP & 1)) {
            const float *W = LP + wofs + lay.l_wqkv, *bias = LP + lay.l_bqkv;
            run_jobs2<D, NT_QKV, MT, 4, true, F16>(
                hbuf, HS, bP, bQ, [&](int nt, int, int, float (&b)[D / 2]) { load_w64(W, nt, b); },
                [&](int mt, int nt, int r, int h, const f32x16 &acc) {
                    const int col = nt * 32 + r;
                    const float bv = bias[col];
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int lr = mt * 32 + acc_row(i, h);
                        qkv[lr * QS + col] = acc[i] + bv;
                        if constexpr (SAVE) if (live(lr)) sv.qkv[((size_t)l * sv.rows + (size_t)g0 * N + lr) * (3 * D) + col] = acc[i] + bv;
                    }
                }, 16.0f, SC[ENC_SC_INV_QKV]);
        }
        constexpr bool PF_ATT = N <= 8;   // larger graphs need the registers for the attention rows
        if (PF_ATT && ft_o >= 0) load_w64(LP + wofs + lay.l_wo, ft_o, bP);   // lands while attention runs
        __syncthreads();
        ENC_STAMP();
        // ---- attention per (graph, head, query) ---------------------------------------------------
        // SP lanes per (graph, head, query) item, DK / SP of the head's dimensions each: with few wide heads (the OFA
        // surrogate: 2 heads of 32, 120 items) one lane per item leaves half the threads idle with 64 registers of q and o
        // each; the lanes' partial dot products meet through one DPP exchange (both get the same sum)
        constexpr int SP = (N > 8 && !DROP && 2 * G * H * N <= 256 && DK % 8 == 0) ? 2 : 1;
        constexpr int DS = DK / SP;
        if (!(DNAG_ENC_SKIP & 2))
        for (int t2 = tid; t2 < G * H * N * SP; t2 += 256) {
            const int it = t2 / SP, eo = (t2 % SP) * DS;
            const int i = it % N, hd = (it / N) % H, g = it / (N * H);
            const int row = g * N + i;
            float q[DS];
            load_row<DS>(qkv + row * QS + hd * DK + eo, q);
            if constexpr (N > 8 && !DROP) {
                // Larger graphs: the N scores are not kept (N registers per item, and fully unrolled key loops, pushed the
                // weight prefetches into scratch); they are computed twice instead, once for the row maximum and once
                // for exp / sum / P V, with the normalisation applied to the 8 outputs at the end.
                auto score = [&](int j) {
                    const float *kp = qkv + (g * N + j) * QS + D + hd * DK + eo;
                    float acc = 0.f;
#pragma unroll
                    for (int e = 0; e < DS; e += 4) {
                        const v4f v4 = *reinterpret_cast<const v4f *>(kp + e);
                        acc = fmaf(q[e], v4.x, acc); acc = fmaf(q[e + 1], v4.y, acc);
                        acc = fmaf(q[e + 2], v4.z, acc); acc = fmaf(q[e + 3], v4.w, acc);
                    }
                    if constexpr (SP == 2) acc += dpp_mov<0xB1>(acc);      // quad_perm [1,0,3,2]: the item's other lane
                    acc = acc * (1.0f / sqrtf((float)DK));
                    return maskb[(g * N + i) * N + j] == 0.f ? -1e9f : acc;
                };
                float mx = -3.0e38f;
#pragma unroll 4
                for (int j = 0; j < N; ++j) mx = fmaxf(mx, score(j));
                float den = 0.f;
                float o[DS];
#pragma unroll
                for (int e = 0; e < DS; ++e) o[e] = 0.f;
#pragma unroll 4
                for (int j = 0; j < N; ++j) {
                    const float p = __expf(score(j) - mx);
                    den += p;
                    const float *vp = qkv + (g * N + j) * QS + 2 * D + hd * DK + eo;
#pragma unroll
                    for (int e = 0; e < DS; e += 4) {
                        const v4f v4 = *reinterpret_cast<const v4f *>(vp + e);
                        o[e] = fmaf(p, v4.x, o[e]); o[e + 1] = fmaf(p, v4.y, o[e + 1]);
                        o[e + 2] = fmaf(p, v4.z, o[e + 2]); o[e + 3] = fmaf(p, v4.w, o[e + 3]);
                    }
                }
                const float rden = 1.0f / den;
                if (attn0 != nullptr && l == 0 && g0 + g < batch)
                    for (int j = 0; j < N; ++j) {
                        const float pj = __expf(score(j) - mx) * rden;        // both lanes take part in the exchange
                        if (eo == 0) attn0[(((size_t)(g0 + g) * H + hd) * N + i) * N + j] = pj;
                    }
                float *op = qkv + row * QS + hd * DK + eo;   // overwrite this item's own q slot
#pragma unroll
                for (int e = 0; e < DS; e += 4) *reinterpret_cast<v4f *>(op + e) = v4f{o[e] * rden, o[e + 1] * rden, o[e + 2] * rden, o[e + 3] * rden};
                continue;
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
                acc = acc * (1.0f / sqrtf((float)DK));    // constant-folded reciprocal
                if (maskb[(g * N + i) * N + j] == 0.f) acc = -1e9f;
                s[j] = acc;
                mx = fmaxf(mx, acc);
            }
            float den = 0.f;
#pragma unroll
            for (int j = 0; j < N; ++j) { s[j] = __expf(s[j] - mx); den += s[j]; }
            float o[DK];
#pragma unroll
            for (int e = 0; e < DK; ++e) o[e] = 0.f;
            const float rden = 1.0f / den;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                float p = s[j] * rden;
                if (attn0 != nullptr && l == 0 && g0 + g < batch)
                    attn0[(((size_t)(g0 + g) * H + hd) * N + i) * N + j] = p;
                if constexpr (DROP) if (g0 + g < batch) p = drop_apply(p, KL[(size_t)row * dm.per_layer + hd * N + j], dm.keep_prob);   // transformer.py:19
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
        if (!PF_ATT && ft_o >= 0) load_w64(LP + wofs + lay.l_wo, ft_o, bP);
        __syncthreads();
        ENC_STAMP();
        // ---- h += attn Wo^T + bo ------------------------------------------------------------------
        if (!(DNAG_ENC_SKIP & 4)) {
            const float *W = LP + wofs + lay.l_wo, *bias = LP + lay.l_bo;
            run_jobs2<D, NT_O, MT, 4, false, F16>(
                qkv, QS, bP, bP, [&](int nt, int, int, float (&b)[D / 2]) { load_w64(W, nt, b); },
                [&](int mt, int nt, int r, int h, const f32x16 &acc) {
                    const int col = nt * 32 + r;
                    const float bv = bias[col];
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int lr = mt * 32 + acc_row(i, h);
                        float *hp = hbuf + lr * HS + col;
                        float yv = acc[i] + bv;
                        if constexpr (DROP) if (live(lr)) yv = drop_apply(yv, KL[(size_t)lr * dm.per_layer + H * N + col], dm.keep_prob);   // transformer.py:80
                        const float hv = yv + *hp;
                        *hp = hv;
                        if constexpr (SAVE) if (live(lr)) sv.h1[((size_t)l * sv.rows + (size_t)g0 * N + lr) * D + col] = hv;
                    }
                }, SC[ENC_SC_SA_O], SC[ENC_SC_INV_O]);
        }
        if (ft_f1 >= 0) load_w64(LP + wofs + lay.l_w1, ft_f1, bQ);
        if (ft_f2 >= 0) {
            const float *w = LP + wofs + lay.l_w2;
            asm volatile("" : "+s"(w));
            load_frag<DFF>(w, ft_f2, bR);
        }
        __syncthreads();
        ENC_STAMP();
        // ---- hidden = relu(LN2(h) W1'^T + b1') -----------------------------------------------------
        if (!(DNAG_ENC_SKIP & 8)) {
            const float *W = LP + wofs + lay.l_w1, *bias = LP + lay.l_b1;
            run_jobs2<D, NT_F1, MT, 4, true, F16>(
                hbuf, HS, bQ, bP, [&](int nt, int, int, float (&b)[D / 2]) { load_w64(W, nt, b); },
                [&](int mt, int nt, int r, int h, const f32x16 &acc) {
                    const int col = nt * 32 + r;
                    const float bv = bias[col];
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int lr = mt * 32 + acc_row(i, h);
                        float fv = fmaxf(acc[i] + bv, 0.f);
                        if constexpr (DROP) if (live(lr)) fv = drop_apply(fv, KL[(size_t)lr * dm.per_layer + H * N + D + col], dm.keep_prob);   // transformer.py:54
                        qkv[lr * QS + col] = fv;
                        if constexpr (SAVE) if (live(lr)) sv.f[((size_t)l * sv.rows + (size_t)g0 * N + lr) * DFF + col] = fv;
                    }
                }, 16.0f, SC[ENC_SC_INV_F1]);
        }
        if (l + 1 < L && ft_qkv >= 0) load_w64(LP + wofs + lay.layer_stride + lay.l_wqkv, ft_qkv, bP);   // next layer's first tile
        __syncthreads();
        ENC_STAMP();
        // ---- h += hidden W2^T + b2 ----------------------------------------------------------------
        if (!(DNAG_ENC_SKIP & 16)) {
            const float *W = LP + wofs + lay.l_w2, *bias = LP + lay.l_b2;
            run_jobs2<DFF, NT_F2, MT, 4, false, F16>(
                qkv, QS, bR, bR,
                [&](int nt, int, int, float (&b)[DFF / 2]) { load_frag<DFF>(W, nt, b); },
                [&](int mt, int nt, int r, int h, const f32x16 &acc) {
                    const int col = nt * 32 + r;
                    const float bv = bias[col];
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int lr = mt * 32 + acc_row(i, h);
                        float *hp = hbuf + lr * HS + col;
                        float yv = acc[i] + bv;
                        if constexpr (DROP) if (live(lr)) yv = drop_apply(yv, KL[(size_t)lr * dm.per_layer + H * N + D + DFF + col], dm.keep_prob);   // transformer.py:108
                        const float hv = yv + *hp;
                        *hp = hv;
                        if constexpr (SAVE) if (live(lr)) sv.h[((size_t)(l + 1) * sv.rows + (size_t)g0 * N + lr) * D + col] = hv;
                    }
                }, SC[ENC_SC_SA_F2], SC[ENC_SC_INV_F2]);
        }
        __syncthreads();
        ENC_STAMP();
        // ---- layer output norm (transformer.py:177) -> concat slice: 16 lanes per row, DPP statistics ----
        if (!(DNAG_ENC_SKIP & 64)) {
            const float *og = LP + lay.l_ong, *ob = LP + lay.l_onb;
            const int sub = tid & 15, slot = tid >> 4;
            const v4f g4 = *reinterpret_cast<const v4f *>(og + 4 * sub);
            const v4f b4 = *reinterpret_cast<const v4f *>(ob + 4 * sub);
#pragma unroll
            for (int p = 0; p < RP / 16; ++p) {
                const int r = slot + 16 * p;
                v4f v = *reinterpret_cast<const v4f *>(hbuf + r * HS + 4 * sub);
                const float m = row16_sum((v.x + v.y) + (v.z + v.w)) * (1.0f / 64.0f);
                v.x -= m; v.y -= m; v.z -= m; v.w -= m;
                const float var = row16_sum((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w)) * (1.0f / 64.0f);
                const float s = 1.0f / sqrtf(var + 1e-5f);
                const int g = r / N;
                if (r < R && g0 + g < batch) {
                    v4f o4;
                    o4.x = v.x * s * g4.x + b4.x; o4.y = v.y * s * g4.y + b4.y;
                    o4.z = v.z * s * g4.z + b4.z; o4.w = v.w * s * g4.w + b4.w;
                    if constexpr (SPLIT) {
                        typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
                        h4_t hi4, lo4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float sv16 = v[e] * s * sa_concat;      // LN(h) without the affine (folded into final.linears.0, CateLayout::f0_s)
                            const _Float16 hi = (_Float16)sv16;
                            hi4[e] = hi;
                            lo4[e] = (_Float16)(sv16 - (float)hi);
                        }
                        const int k = l * D + 4 * sub;
                        _Float16 *dst = reinterpret_cast<_Float16 *>(concat) + kblock_half_at((size_t)g0 * N + r, (k >> 3) * 16 + (k & 7), rows_pad);
                        *reinterpret_cast<h4_t *>(dst) = hi4;
                        *reinterpret_cast<h4_t *>(dst + 8) = lo4;
                    } else {
                        *reinterpret_cast<v4f *>(concat + ((size_t)g0 * N + r) * lay.F + l * D + 4 * sub) = o4;
                        if constexpr (SAVE) cmax = fmaxf(fmaxf(cmax, fmaxf(fabsf(o4.x), fabsf(o4.y))), fmaxf(fabsf(o4.z), fabsf(o4.w)));
                    }
                }
            }
        }
        ENC_STAMP();
        // no barrier needed: the next phase only reads hbuf and writes qkv
    }
    if constexpr (SAVE && !SPLIT) {
        if (sv.amax_concat != nullptr) {             // max |concat| of the rows this workgroup wrote (a later GEMM's operand scale)
            cmax = fmaxf(cmax, dpp_mov<0xB1>(cmax));
            cmax = fmaxf(cmax, dpp_mov<0x4E>(cmax));
            cmax = fmaxf(cmax, dpp_mov<0x141>(cmax));
            cmax = fmaxf(cmax, dpp_mov<0x140>(cmax));
            if ((tid & 15) == 0) atomicMax(reinterpret_cast<unsigned int *>(sv.amax_concat), __float_as_uint(cmax));
        }
    }
#ifdef DNAG_ENC_STAMPS
    if (blockIdx.x == 5 && tid == 0) for (int i = 1; i < nts_; ++i) printf("enc %d %lld\n", i, ts_[i] - ts_[i - 1]);
#endif
}

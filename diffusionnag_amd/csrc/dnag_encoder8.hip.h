// Sampling-path encoder of the CATE score network for 8-node graphs (NAS-Bench-201): the same mathematics as
// encoder_kernel<8, 7, 8, 8, 128, true, false, false, true, true> (dnag_encoder.hip.h, which stays the kernel of the
// training step, of the OFA shape and of the attention-probability output), organised around what that kernel's
// per-phase cycle stamps showed: it is bound by vector-ALU issue, most of it the LayerNorm and
// the fp16 hi/lo split of the A operand, which every 32x32 tile job repeated (12 QKV jobs for 2 row tiles).
//
// Here every activation that feeds a linear is normalised / scaled / split ONCE, by the phase that produces it, into an
// LDS image of split-pair rows (dnag_gemm16.hip.h: per 8 values 16 B of fp16 hi, 16 B of lo), so a tile job is two
// ds_read_b128 and three v_mfma_f32_32x32x16_f16 per 16 k:
//   P0  row statistics of h: concat slice of the previous layer (output norm, transformer.py:177) and LN1(h) -> A16
//   P1  q|k|v = A16 Wqkv'^T + b            (3 tile jobs per wave)                    -> qkv fp32 [64][192]
//   P2  attention: one thread per (graph, head, query pair): K and V rows are read once for two queries and the
//       two dot products / two output rows ride on packed fp32 FMAs                   -> A16 (scale sa_o)
//   P3  h += A16 Wo^T + bo                 (the wave's tile of h lives in registers) -> hbuf
//   P4  LN2(h) -> A16
//   P5  hidden = relu(A16 W1'^T + b1'), MFMA operands swapped so that a lane owns one batch row and writes its 16
//       columns as 4-value runs of the split image                                     -> hidden (scale sa_f2)
//   P6  h += hidden W2^T + b2                                                         -> hbuf
// The residual stream h is kept twice: in the accumulator layout in registers (one 32x32 tile per wave: P3 and P6 have
// exactly four jobs), where the residual additions happen, and as fp32 rows in LDS for the row statistics.
// Restates (reference NAS-Bench-201/): models/cate.py:145-160, models/transformer.py:12-20, :23-42, :69-80, :97-108,
// :127-136, :166-179, models/cate.py:83-86.
#pragma once
#include "dnag_common.h"
#include "dnag_encoder.hip.h"

// Two queries per thread as scalar pairs; the compiler's SLP pass turns most of these pairs into v_pk_fma_f32 /
// v_pk_mul_f32 / v_pk_add_f32 with op_sel broadcasts on its own (244 packed instructions in the production ISA).
// Round 2 also carried a hand-packed variant (float2 ext vectors, -DENC8_PK) that was bit-exact at one workgroup per CU
// and wrong in lanes 32-63 at two; its ISA was diffed against this build in round 3 (profiles/r03_enc8_pk_isa.md: same
// barriers, waits and register budget, no spills; it differs in which softmax steps ride on packed adds behind
// v_exp_f32 / v_rcp_f32 results), no cause could be named from the ISA, and the variant was deleted.
struct v2f {
    float x, y;
    __device__ __forceinline__ v2f operator*(v2f o) const { return v2f{x * o.x, y * o.y}; }
    __device__ __forceinline__ v2f operator*(float o) const { return v2f{x * o, y * o}; }
    __device__ __forceinline__ v2f &operator+=(v2f o) { x += o.x; y += o.y; return *this; }
};
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return v2f{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)}; }
__device__ __forceinline__ v2f pk_max(v2f a, v2f b) { return v2f{fmaxf(a.x, b.x), fmaxf(a.y, b.y)}; }

struct Enc8Geom {
    static constexpr int N = 8, G = 8, R = 64, D = 64, H = 8, DK = 8, DFF = 128;
    static constexpr int AS = D + 4;            // dword row stride of A16 and hbuf (== 4 mod 64: conflict-free ds_read_b128)
    static constexpr int QS = 3 * D + 4;        // q|k|v rows
    static constexpr int FS = DFF + 4;          // split hidden rows (DFF values = DFF dwords)
    static constexpr int BIG = R * FS + R * AS; // hidden, then hbuf behind it (qkv [R][QS] fits in front: R*QS <= BIG)
    static_assert(R * QS <= BIG, "q|k|v aliases the hidden + hbuf region");
    static constexpr int VEC = 3 * D + D + DFF + D;   // staged per layer: bqkv' | bo | b1' | b2, then two parity slots of on_g | on_b
    static constexpr size_t lds_floats = (size_t)R * AS + BIG + (size_t)G * N * N + VEC + 4 * D;
};

// hi/lo halves of four values that already carry their power-of-two scale
__device__ __forceinline__ void split4(const v4f v, h4 &hi, h4 &lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const _Float16 x = (_Float16)v[e];
        hi[e] = x;
        lo[e] = (_Float16)(v[e] - (float)x);
    }
}

template <int V>
__global__ __launch_bounds__(256, 2) void encoder8_kernel(const float *__restrict__ P, const CateLayout lay,
                                                          const float *__restrict__ x, const float *__restrict__ temb,
                                                          int tstride, const float *__restrict__ mask, int mstride,
                                                          float *__restrict__ concat, int batch) {
    typedef Enc8Geom Ge;
    constexpr int N = Ge::N, G = Ge::G, R = Ge::R, D = Ge::D, H = Ge::H, DK = Ge::DK, DFF = Ge::DFF;
    constexpr int AS = Ge::AS, QS = Ge::QS, FS = Ge::FS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *a16 = smem;                       // [R][AS] split-pair rows, 64 values each
    float *big = a16 + R * AS;
    float *qkv = big;                        // [R][QS] fp32 (P1 -> P2)
    float *hid = big;                        // [R][FS] split-pair rows, 128 values each (P5 -> P6)
    float *hbuf = big + R * FS;              // [R][AS] fp32 residual stream for the row statistics (P3 -> P4, P6 -> P0)
    float *maskb = big + Ge::BIG;            // [G][N][N]
    float *bst = maskb + G * N * N;          // this layer's bias vectors: bqkv' [192] | bo [64] | b1' [128] | b2 [64]
    float *onst = bst + Ge::VEC;             // [2][on_g 64 | on_b 64]: output-norm affine of layer l in slot l & 1

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int g0 = blockIdx.x * G;
    const int L = lay.L;
    const float sa_concat = P[lay.hsc + HEAD_SC_SA_CONCAT];
    const size_t rows_pad = ((size_t)batch * N + 255) / 256 * 256;      // the concat is k-tile blocked (dnag_gemm16.hip.h kblock_at)
    const float *P16 = P + lay.delta16;      // fp16 hi/lo fragment copy of the layer matrices

    // this wave's tile of the residual stream (P3 / P6 jobs): rows tmt*32 .., columns tnt*32 ..
#ifdef ENC8_STAMPS
    long long ts_[128];
    int nts_ = 0;
#define E8_STAMP() do { if (nts_ < 128) ts_[nts_++] = __builtin_readcyclecounter(); } while (0)
#else
#define E8_STAMP()
#endif
    E8_STAMP();
    const int tnt = wave >> 1, tmt = wave & 1;
    // P1: jobs j = 3 wave .. 3 wave + 2 over (nt = j / 2, mt = j % 2): column tiles q0 (first) and q0 + 1
    const int q0 = (3 * wave) >> 1;

    // weight fragments in registers, fetched a phase or more ahead of their use (issue points pinned by the asm)
    float bP[D / 2], bQ[D / 2], bR[DFF / 2];
    auto fetch64 = [&](const float *W, int nt, float (&b)[D / 2]) {
        asm volatile("" : "+s"(W));
        load_frag<D>(W, nt, b);
    };
    auto fetch128 = [&](const float *W, int nt, float (&b)[DFF / 2]) {
        asm volatile("" : "+s"(W));
        load_frag<DFF>(W, nt, b);
    };
    {
        const float *W = P16 + lay.layer0 + lay.l_wqkv;
        fetch64(W, q0, bP);
        fetch64(W, q0 + 1, bQ);
    }

    // ---- stage inputs ---------------------------------------------------------------------------
    float *xin = qkv;                        // [R][V] alias
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
        for (int i = tid; i < R * D; i += 256) {
            const int rr = i / D, c = i - rr * D;
            const int g = rr / N, n = rr - g * N;
            float val = 0.f;
            if (g0 + g < batch) {
                float acc = 0.f;
#pragma unroll
                for (int v = 0; v < V; ++v) acc = fmaf(xin[rr * V + v], We[c * V + v], acc);
                val = (acc + be[c]) * 8.0f;
                val = val + pe[n * D + c];
                val = val + temb[(size_t)(tstride ? g0 + g : 0) * D + c];
            }
            hbuf[rr * AS + c] = val;
        }
    }
    __syncthreads();
    f32x16 hacc;
#pragma unroll
    for (int i = 0; i < 16; ++i) hacc[i] = hbuf[(tmt * 32 + acc_row(i, h)) * AS + tnt * 32 + r];

    // Row statistics of h (16 lanes per row, DPP sums): writes LN(h) * 16 as split-pair rows into A16 (to_a16) and
    // the output norm of layer `lprev` as a concat slice (lprev >= 0).
    // The vectors of a layer (CateLayout::l_vec) are fetched one 16-byte piece per thread well ahead (stage_fetch: before
    // the loop, then behind P5 of the layer before) and parked in LDS at the start of P0 (stage_put), so no epilogue
    // waits for a global load that queues behind the weight prefetches.
    constexpr int NST = (Ge::VEC + 2 * D) / 4;
    v4f st4 = {0.f, 0.f, 0.f, 0.f};
    auto stage_fetch = [&](int lstage) {
        if (tid < NST) st4 = *reinterpret_cast<const v4f *>(P + lay.layer0 + (size_t)lstage * lay.layer_stride + lay.l_vec + 4 * tid);
    };
    auto stage_put = [&](int lstage) {
        if (tid < NST) {
            float *dst = tid < Ge::VEC / 4 ? bst + 4 * tid : onst + (lstage & 1) * 2 * D + (4 * tid - Ge::VEC);
            *reinterpret_cast<v4f *>(dst) = st4;
        }
    };
    // Row statistics of h (16 lanes per row, DPP sums): writes LN(h) * 16 as split-pair rows into A16 (to_a16) and
    // the output norm of layer `lprev` (lprev >= 0) as its concat slice.  (Holding the slice in registers to store it
    // behind P1 -- whose wait for weight fragments also waits for these stores -- was measured: the 16 registers spill.)
    auto norm_pass = [&](bool to_a16, int lprev) {
        const int sub = tid & 15, slot = tid >> 4;
        v4f g4 = {0.f, 0.f, 0.f, 0.f}, b4 = {0.f, 0.f, 0.f, 0.f};
        if (lprev >= 0) {
            g4 = *reinterpret_cast<const v4f *>(onst + (lprev & 1) * 2 * D + 4 * sub);
            b4 = *reinterpret_cast<const v4f *>(onst + (lprev & 1) * 2 * D + D + 4 * sub);
        }
#pragma unroll
        for (int p = 0; p < R / 16; ++p) {
            const int rr = slot + 16 * p;
            v4f v = *reinterpret_cast<const v4f *>(hbuf + rr * AS + 4 * sub);
            const float m = row16_sum((v.x + v.y) + (v.z + v.w)) * (1.0f / 64.0f);
            v.x -= m; v.y -= m; v.z -= m; v.w -= m;
            const float var = row16_sum((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w)) * (1.0f / 64.0f);
            const float s = __builtin_amdgcn_rsqf(var + 1e-5f);     // v_rsq_f32: 1 ulp, the argument is a normal number
            if (to_a16) {
                // value k = 4 sub + e of the row sits at half (k >> 3) * 16 + (k & 7) of its split-pair row (lo: + 8)
                const int at = (sub >> 1) * 16 + (sub & 1) * 4;
                h4 hi, lo;
                split4(v4f{v.x * s * 16.0f, v.y * s * 16.0f, v.z * s * 16.0f, v.w * s * 16.0f}, hi, lo);
                _Float16 *dst = reinterpret_cast<_Float16 *>(a16 + rr * AS) + at;
                *reinterpret_cast<h4 *>(dst) = hi;
                *reinterpret_cast<h4 *>(dst + 8) = lo;
            }
            if (lprev >= 0) {
                v4f o4;
                o4.x = v.x * s * g4.x + b4.x; o4.y = v.y * s * g4.y + b4.y;
                o4.z = v.z * s * g4.z + b4.z; o4.w = v.w * s * g4.w + b4.w;
                h4 hi, lo;
                split4(o4 * sa_concat, hi, lo);
                if (g0 + rr / N < batch) {
                    const int at = (sub >> 1) * 16 + (sub & 1) * 4;
                    _Float16 *dst = reinterpret_cast<_Float16 *>(concat) + kblock_half_at((size_t)g0 * N + rr, lprev * (2 * D) + at, rows_pad);
                    *reinterpret_cast<h4 *>(dst) = hi;
                    *reinterpret_cast<h4 *>(dst + 8) = lo;
                }
            }
        }
    };
    stage_fetch(0);
    E8_STAMP();
    for (int l = 0; l <= L; ++l) {
        // ---- P0: statistics of h -> concat slice l - 1, A16 = LN1(h) -----------------------------------
        if (l < L) stage_put(l);
        norm_pass(l < L, l - 1);
        if (l == L) break;
        const float *LP = P + lay.layer0 + (size_t)l * lay.layer_stride;
        const float *LP16 = P16 + lay.layer0 + (size_t)l * lay.layer_stride;
        const float *SC = LP + lay.l_sc;
        __syncthreads();
        E8_STAMP();
        // ---- P1: q|k|v -------------------------------------------------------------------------------
        {
            const float inv = SC[ENC_SC_INV_QKV];
            const float *bias = bst;
            auto job = [&](const float (&b)[D / 2], int nt, int mt) {
                const float *ap = a16 + (mt * 32 + r) * AS + h * (D / 2);
                f32x16 acc;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
                for (int s = 0; s < D / 16; ++s) {
                    const h8 ahi = *reinterpret_cast<const h8 *>(ap + 8 * s);
                    const h8 alo = *reinterpret_cast<const h8 *>(ap + 8 * s + 4);
                    const int o = 8 * s;
                    const h8 bhi = __builtin_bit_cast(h8, v4f{b[o], b[o + 1], b[o + 2], b[o + 3]});
                    const h8 blo = __builtin_bit_cast(h8, v4f{b[o + 4], b[o + 5], b[o + 6], b[o + 7]});
                    acc = mfma16(ahi, bhi, acc);
                    acc = mfma16(ahi, blo, acc);
                    acc = mfma16(alo, bhi, acc);
                }
                const int col = nt * 32 + r;
                const float bv = bias[col];
#pragma unroll
                for (int i = 0; i < 16; ++i) qkv[(mt * 32 + acc_row(i, h)) * QS + col] = acc[i] * inv + bv;
            };
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) {
                const int j = 3 * wave + jj;
                const int nt = j >> 1, mt = j & 1;
                if (nt == q0) job(bP, nt, mt);       // wave-uniform: a scalar branch, not per-register selects
                else job(bQ, nt, mt);
            }
        }
        fetch64(LP16 + lay.l_wo, tnt, bP);         // lands while attention runs (P3)
        __syncthreads();
        E8_STAMP();
        // ---- P2: attention, one thread per (graph, head, query pair) ----------------------------------------
        {
            const int qp = tid & 3, hd = (tid >> 2) & 7, g = tid >> 5;
            const int row0 = g * N + 2 * qp;
            const float sa_o = SC[ENC_SC_SA_O];
            v2f q[DK];
            {
                const float *qa = qkv + row0 * QS + hd * DK, *qb = qa + QS;
#pragma unroll
                for (int e = 0; e < DK; e += 4) {
                    const v4f a = *reinterpret_cast<const v4f *>(qa + e), b = *reinterpret_cast<const v4f *>(qb + e);
                    q[e] = v2f{a.x, b.x}; q[e + 1] = v2f{a.y, b.y}; q[e + 2] = v2f{a.z, b.z}; q[e + 3] = v2f{a.w, b.w};
                }
            }
            v2f s[N];
            v2f mx = {-3.0e38f, -3.0e38f};
            const float *mrow = maskb + (g * N + 2 * qp) * N;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const float *kp = qkv + (g * N + j) * QS + D + hd * DK;
                v2f acc = {0.f, 0.f};
#pragma unroll
                for (int e = 0; e < DK; e += 4) {
                    const v4f k4 = *reinterpret_cast<const v4f *>(kp + e);
                    acc = pk_fma(q[e], v2f{k4.x, k4.x}, acc);
                    acc = pk_fma(q[e + 1], v2f{k4.y, k4.y}, acc);
                    acc = pk_fma(q[e + 2], v2f{k4.z, k4.z}, acc);
                    acc = pk_fma(q[e + 3], v2f{k4.w, k4.w}, acc);
                }
                acc = acc * (1.0f / sqrtf((float)DK));
                if (mrow[j] == 0.f) acc.x = -1e9f;          // transformer.py:16
                if (mrow[N + j] == 0.f) acc.y = -1e9f;
                s[j] = acc;
                mx = pk_max(mx, acc);
            }
            v2f den = {0.f, 0.f};
#pragma unroll
            for (int j = 0; j < N; ++j) {
                s[j] = v2f{__expf(s[j].x - mx.x), __expf(s[j].y - mx.y)};
                den += s[j];
            }
            const v2f rden = v2f{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};     // v_rcp_f32, 1 ulp; den in [1, 8]
            v2f o[DK];
#pragma unroll
            for (int e = 0; e < DK; ++e) o[e] = v2f{0.f, 0.f};
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const v2f p = s[j] * rden;
                const float *vp = qkv + (g * N + j) * QS + 2 * D + hd * DK;
#pragma unroll
                for (int e = 0; e < DK; e += 4) {
                    const v4f v4 = *reinterpret_cast<const v4f *>(vp + e);
                    o[e] = pk_fma(p, v2f{v4.x, v4.x}, o[e]);
                    o[e + 1] = pk_fma(p, v2f{v4.y, v4.y}, o[e + 1]);
                    o[e + 2] = pk_fma(p, v2f{v4.z, v4.z}, o[e + 2]);
                    o[e + 3] = pk_fma(p, v2f{v4.w, v4.w}, o[e + 3]);
                }
            }
            // head hd = values 8 hd .. 8 hd + 7 of the row = split group hd: 16 B of hi, 16 B of lo
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                h8 hi, lo;
#pragma unroll
                for (int e = 0; e < DK; ++e) {
                    const float v = (w ? o[e].y : o[e].x) * sa_o;
                    const _Float16 xh = (_Float16)v;
                    hi[e] = xh;
                    lo[e] = (_Float16)(v - (float)xh);
                }
                float *dst = a16 + (row0 + w) * AS + hd * 8;
                *reinterpret_cast<h8 *>(dst) = hi;
                *reinterpret_cast<h8 *>(dst + 4) = lo;
            }
        }
        // fetched behind the attention, whose registers they would otherwise compete for
        fetch64(LP16 + lay.l_w1, wave, bQ);        // P5
        fetch128(LP16 + lay.l_w2, tnt, bR);        // P6
        __syncthreads();
        E8_STAMP();
        // ---- P3: h += attn Wo^T + bo --------------------------------------------------------------------
        {
            const float inv = SC[ENC_SC_INV_O];
            const float *ap = a16 + (tmt * 32 + r) * AS + h * (D / 2);
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int s = 0; s < D / 16; ++s) {
                const h8 ahi = *reinterpret_cast<const h8 *>(ap + 8 * s);
                const h8 alo = *reinterpret_cast<const h8 *>(ap + 8 * s + 4);
                const int o = 8 * s;
                const h8 bhi = __builtin_bit_cast(h8, v4f{bP[o], bP[o + 1], bP[o + 2], bP[o + 3]});
                const h8 blo = __builtin_bit_cast(h8, v4f{bP[o + 4], bP[o + 5], bP[o + 6], bP[o + 7]});
                acc = mfma16(ahi, bhi, acc);
                acc = mfma16(ahi, blo, acc);
                acc = mfma16(alo, bhi, acc);
            }
            const int col = tnt * 32 + r;
            const float bv = bst[3 * D + col];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                hacc[i] = (acc[i] * inv + bv) + hacc[i];
                hbuf[(tmt * 32 + acc_row(i, h)) * AS + col] = hacc[i];
            }
        }
        if (l + 1 < L) fetch64(LP16 + lay.layer_stride + lay.l_wqkv, q0, bP);      // next layer's P1
        __syncthreads();
        E8_STAMP();
        // ---- P4: A16 = LN2(h) -----------------------------------------------------------------------------
        norm_pass(true, -1);
        __syncthreads();
        E8_STAMP();
        // ---- P5: hidden = relu(LN2(h) W1'^T + b1'), transposed tiles: lane <-> batch row ---------------------
        {
            const float inv = SC[ENC_SC_INV_F1], sa_f2 = SC[ENC_SC_SA_F2];
            const int nt = wave;
            // the lane's 16 output columns nt*32 + 8 gq + 4 h + (0..3), gq = 0..3
            v4f bias4[4];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) bias4[gq] = *reinterpret_cast<const v4f *>(bst + 4 * D + nt * 32 + 8 * gq + 4 * h);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const float *ap = a16 + (mt * 32 + r) * AS + h * (D / 2);
                f32x16 acc;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
                for (int s = 0; s < D / 16; ++s) {
                    const h8 ahi = *reinterpret_cast<const h8 *>(ap + 8 * s);
                    const h8 alo = *reinterpret_cast<const h8 *>(ap + 8 * s + 4);
                    const int o = 8 * s;
                    const h8 bhi = __builtin_bit_cast(h8, v4f{bQ[o], bQ[o + 1], bQ[o + 2], bQ[o + 3]});
                    const h8 blo = __builtin_bit_cast(h8, v4f{bQ[o + 4], bQ[o + 5], bQ[o + 6], bQ[o + 7]});
                    acc = mfma16(bhi, ahi, acc);      // D[n][m]: lane = batch row m, registers = columns n
                    acc = mfma16(blo, ahi, acc);
                    acc = mfma16(bhi, alo, acc);
                }
                _Float16 *hrow = reinterpret_cast<_Float16 *>(hid + (mt * 32 + r) * FS);
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    v4f f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) f[e] = fmaxf(acc[4 * gq + e] * inv + bias4[gq][e], 0.f) * sa_f2;
                    h4 hi, lo;
                    split4(f, hi, lo);
                    _Float16 *dst = hrow + (nt * 4 + gq) * 16 + 4 * h;
                    *reinterpret_cast<h4 *>(dst) = hi;
                    *reinterpret_cast<h4 *>(dst + 8) = lo;
                }
            }
        }
        if (l + 1 < L) {
            stage_fetch(l + 1);
            fetch64(LP16 + lay.layer_stride + lay.l_wqkv, q0 + 1, bQ);
        }
        __syncthreads();
        E8_STAMP();
        // ---- P6: h += hidden W2^T + b2 ------------------------------------------------------------------------
        {
            const float inv = SC[ENC_SC_INV_F2];
            const float *ap = hid + (tmt * 32 + r) * FS + h * (DFF / 2);
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int s = 0; s < DFF / 16; ++s) {
                const h8 ahi = *reinterpret_cast<const h8 *>(ap + 8 * s);
                const h8 alo = *reinterpret_cast<const h8 *>(ap + 8 * s + 4);
                const int o = 8 * s;
                const h8 bhi = __builtin_bit_cast(h8, v4f{bR[o], bR[o + 1], bR[o + 2], bR[o + 3]});
                const h8 blo = __builtin_bit_cast(h8, v4f{bR[o + 4], bR[o + 5], bR[o + 6], bR[o + 7]});
                acc = mfma16(ahi, bhi, acc);
                acc = mfma16(ahi, blo, acc);
                acc = mfma16(alo, bhi, acc);
            }
            const int col = tnt * 32 + r;
            const float bv = bst[4 * D + DFF + col];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                hacc[i] = (acc[i] * inv + bv) + hacc[i];
                hbuf[(tmt * 32 + acc_row(i, h)) * AS + col] = hacc[i];
            }
        }
        __syncthreads();
        E8_STAMP();
    }
#ifdef ENC8_STAMPS
    if (blockIdx.x == 5 && tid == 0) for (int i = 1; i < nts_; ++i) printf("e8 %d %lld\n", i, ts_[i] - ts_[i - 1]);
#endif
}

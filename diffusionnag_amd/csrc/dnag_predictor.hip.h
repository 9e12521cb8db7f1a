// MetaNeuralPredictor surrogate: forward and analytic input gradient, one workgroup per G = 16 graphs
// (128 rows), 10 waves, one workgroup per CU: 4096 graphs are ONE round of 256 workgroups.  Included by dnag_lib.hip.
//
// Restates (reference NAS-Bench-201/): models/digcn_meta.py:109-129 (arch_encode), :39-48
// (DirectedGraphConvolution), :12-22 (normalize_adj, graph_pooling), :141-159 (predict/forward) and
// the autograd call of models/utils.py:130-145 (regress=True) as a hand-derived backward pass.
//
//   Ahat = rn(M + I);  A1 = rn(Ahat);  A2 = rn(Ahat^T)                (rn: divide rows by their sums)
//   H_{l+1} = 1/2 [relu(A1 (H_l W1_l)) + relu(A2 (H_l W2_l))]         evaluated as A1 (H_l W1_l): mixing AFTER the GEMM
//   p = sum_n H_L[n] / N + tau ;  a = Wfc1 p ;  y = w2 . tanh(W0 [D_mu | a] + b0) + b2
// Backward (dy/dx only): dpre = w2 (1 - tanh^2); da = W0[:, nz:]^T dpre; dp = Wfc1^T da;
//   dH_L[n] = dp / N;  per layer  dH_l = A1^T ((1/2 dH_{l+1} . [Z1>0]) W1^T) + A2^T ((1/2 dH_{l+1} . [Z2>0]) W2^T).
//
// Round-3 organisation (round 2: 8 graphs per workgroup, 152 KB of LDS, two rounds of 512 workgroups, mixing through a
// transposed copy of the activations in LDS; 146 us at B = 4096):
//  * Every matrix product is split-arithmetic fp16 MFMA (dnag_gemm16.hip.h): v_mfma_f32_32x32x16_f16, three per 16 k.
//  * A wave owns one 32-column tile and TWO 32-row tiles (8 graphs): the weight fragments stream from L2 once per wave
//    through a three-step register ring and serve both row tiles.
//  * A GEMM leaves a wave with one column and 16 rows per lane.  The rows of a tile are PERMUTED (tile row 8q + 4h + e
//    holds node 4 (q & 1) + e of graph 2h + (q >> 1) of the tile) so that those 16 rows are two complete graphs: the
//    adjacency mixing then runs on the matrix cores STRAIGHT FROM THE ACCUMULATOR REGISTERS (A operand = the lane's 8
//    rows of one graph, split in registers with the wave's own power-of-two scale; B operand = the graph's adjacency row,
//    pre-split in LDS), and its result lands with one row and quads of consecutive columns per lane, i.e. as 8-byte
//    stores into the split-pair rows the next GEMM reads.  No transposed activation copy, no second operand buffer:
//    LDS holds ONE operand buffer [128][144] (76 KB), the pre-split adjacency rows (16 KB) and the relu masks.
//  * The relu masks are one word per lane, row tile and layer (bit i = [Z1 > 0], bit 16 + i = [Z2 > 0] for the lane's
//    16 columns); the backward pass meets them in the same lanes.
//  * The backward GEMMs need two differently masked operands (dH . m1, dH . m2): they go through the one buffer one after
//    the other (four barriers per layer instead of two).
//  * Layer 0 (K = 7 padded to 16), the pooled head (fc1, the tanh layer, its transpose) all run as MFMA tile jobs; the
//    dataset half of pred_fc.0 (c0 = b0 + W0[:, :nz] D_mu, the same for every graph) is computed once per call by
//    pred_c0_kernel instead of once per workgroup.
#pragma once
#include "dnag_common.h"
#include "dnag_encoder.hip.h"

#ifndef DNAG_PRED_SKIP
#define DNAG_PRED_SKIP 0    // debug builds only (-DDNAG_PRED_SKIP=<mask>): 1 = no mixing, 2 = no layer GEMMs; timing ablations, wrong results
#endif
#define PRED_GH 144        // gcn_hidden (configs/tr_meta_surrogate.py:97)
#define PRED_HS 148        // dword row stride of the operand buffer (148 % 64 == 20: conflict-free b128 reads)
#define PRED_NT 5          // ceil(144 / 32) column tiles
#define PRED_G 16          // graphs per workgroup
#define PRED_NW 10         // waves per workgroup: (column tile, pair of row tiles)
#define PRED_MAXHS 512

template <int N, int V>
struct PredGeom {
    static constexpr int G = PRED_G;
    static constexpr int R = G * N;          // 128 rows
    static constexpr int MT = R / 32;        // 4 row tiles
    // 20 (column tile, row tile) jobs per layer GEMM, two per wave.  (Twelve waves -- two jobs for waves 0..7, one for waves
    // 8..11, five jobs per SIMD if waves w, w + 4, w + 8 share one -- measured 113 us against 101 us: round 3.)
    static_assert(N == 8 && MT == 4 && PRED_NT * (MT / 2) == PRED_NW, "one (column tile, row-tile pair) per wave");
    // dynamic LDS bytes for `L` DiGCN layers: operand buffer | pre-split adjacency rows (A1, A2, A1^T, A2^T) | zero row | relu masks
    static size_t lds_bytes(int hs, int L) {
        (void)hs;                            // the head's buffers alias the operand buffer (hs <= 512)
        return ((size_t)R * PRED_HS + 4 * (size_t)R * 8 + 8 + (size_t)L * 2 * PRED_NW * 64) * sizeof(float);
    }
};

// c0 = b0 + W0[:, :nz] D_mu: the dataset half of pred_fc.0 (digcn_meta.py:141-148), constant per call.  One wave per
// output, lanes stride over nz, fixed-order butterfly.
__global__ __launch_bounds__(64) void pred_c0_kernel(const float *__restrict__ P, const PredLayout lay, const float *__restrict__ d_mu,
                                                     float *__restrict__ c0) {
    const int j = blockIdx.x, lane = threadIdx.x;
    const float *w = P + lay.p0_w + (size_t)j * (lay.nz + lay.lh);
    float acc = 0.f;
    for (int z = lane; z < lay.nz; z += 64) acc = fmaf(w[z], d_mu[z], acc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) c0[j] = acc + P[lay.p0_b + j];
}

// x [batch][N][V]; temb [tstride ? batch : 1][144]; mask [mstride ? batch : 1][N][N]; c0 [hs] (pred_c0_kernel);
// y [batch] or null; grad [batch][N][V] or null.
template <int N, int V>
__global__ __launch_bounds__(PRED_NW * 64) void predictor_kernel(const float *__restrict__ P, const PredLayout lay,
                                                                 const float *__restrict__ x, const float *__restrict__ temb,
                                                                 int tstride, const float *__restrict__ mask, int mstride,
                                                                 const float *__restrict__ c0g, float *__restrict__ y,
                                                                 float *__restrict__ grad, int batch) {
    typedef PredGeom<N, V> Ge;
    typedef const __attribute__((address_space(1))) v4f *gv4p;
    constexpr int G = Ge::G, R = Ge::R, HS = PRED_HS, GH = PRED_GH, NT = PRED_NW * 64, LH = 128;
    constexpr float ADJ_SCALE = 1024.0f;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Hb = smem;                        // [R][HS] split-pair rows: the A operand of every layer GEMM
    float *adjop = Hb + R * HS;              // [4][R][8]: per (matrix, graph, node) 8 values * 1024 as 16 B of hi + 16 B of lo
    float *zrow = adjop + 4 * R * 8;         // 8 zero dwords: the B operand of lanes outside the step's graph
    uint32_t *mbits = reinterpret_cast<uint32_t *>(zrow + 8);   // [L][2][NT]: relu masks, one word per lane and row tile
    // head buffers alias the operand buffer (dead between the last forward GEMM and the first backward write)
    float *hA = Hb;                          // [32][HS] fp32: pooled p, later da
    float *hB = Hb + 32 * HS;                // [32][HS] fp32: a, later dH_L per graph
    float *tv = Hb + 64 * HS;                // [G][TS]  tanh / dpre
    __shared__ __attribute__((aligned(16))) float Araw[2][G * N * N];     // A1, A2 (fp32) while the operand rows are built
    __shared__ unsigned int amaxv[32];
    const int L = lay.L, hs = lay.hs;
    const int TS = lay.hs16 + 4;             // row stride of tv (== 4 mod 16: the rows of a ds_read_b128 spread over the banks)
    static_assert((PRED_MAXHS + 4) * G <= 64 * PRED_HS, "tv fits behind the two 32-row head buffers");

    int tid = threadIdx.x;
    const int g0 = blockIdx.x * G;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr bool two = true;                           // every wave has two row tiles (the single-tile path is kept for odd splits)
    const int ct = wv >> 1, mt0 = 2 * (wv & 1);          // column tile, first of the two row tiles
    int ln = tid & 63, lr = ln & 31, lh = ln >> 5;
#ifdef DNAG_PRED_STAMPS
    long long ts_[64];
    int nts_ = 0;
#define PRED_STAMP(k) do { if (nts_ < 64) ts_[nts_++] = clock64(); } while (0)
    PRED_STAMP(0);
#else
#define PRED_STAMP(k)
#endif
    // amaxv: largest |value| of an operand in this workgroup -> its power-of-two scale (the operands' sizes are data
    // dependent: the activations are piecewise linear in x, the gradients carry no bound at all)
    enum { AM_X = 0, AM_H = 1, AM_P = 10, AM_A = 11, AM_D = 12, AM_DA = 13, AM_DH = 14 };   // AM_H + l, AM_DH + l (l < DNAG_MAX_GCN)
    static_assert(AM_DH + DNAG_MAX_GCN <= 32 && AM_H + DNAG_MAX_GCN <= AM_P, "amaxv slots");
    auto wave_max = [&](float m) {           // DPP inside the rows of 16, then the four rows through scalar registers
        m = fmaxf(m, dpp_mov<0xB1>(m));
        m = fmaxf(m, dpp_mov<0x4E>(m));
        m = fmaxf(m, dpp_mov<0x141>(m));
        m = fmaxf(m, dpp_mov<0x140>(m));
        const int mi = __float_as_int(m);
        const float a = __int_as_float(__builtin_amdgcn_readlane(mi, 0)), b = __int_as_float(__builtin_amdgcn_readlane(mi, 16));
        const float c = __int_as_float(__builtin_amdgcn_readlane(mi, 32)), d = __int_as_float(__builtin_amdgcn_readlane(mi, 48));
        return fmaxf(fmaxf(a, b), fmaxf(c, d));
    };
    auto note_max = [&](float m, int slot) {     // wave maximum -> one LDS atomic per wave (every lane must call)
        m = wave_max(m);
        if (ln == 0) atomicMax(&amaxv[slot], __float_as_uint(m));
    };
    auto slot_scale = [&](int slot) { return pow2_scale(__uint_as_float(amaxv[slot])); };
    const float *P16 = P + lay.delta16;
    const float *WSC = P + lay.wsc;           // 1 / sw of every split-arithmetic matrix (dnag_pred_pack)
    const float *MIXS = P + lay.mixs;

    // tile row rho = 8q + 4h + e  <->  row 16h + 4q + e of the tile's 32 buffer rows (graph 2h + (q >> 1), node 4 (q & 1) + e)
    int prow = 16 * ((lr >> 2) & 1) + 4 * (lr >> 3) + (lr & 3);
    int row0 = mt0 * 32 + prow, row1 = two ? row0 + 32 : row0;       // this lane's buffer rows in the wave's tile(s)
    bool mix_active = lh == ((lr >> 2) & 1);                         // B operand of the mixing: see mixf
    int mix_step = lr >> 4;
    // The lane ids are derived afresh at the top of every layer iteration (through an opaque asm): with one set for the
    // whole kernel the compiler hoists each phase's address arithmetic out of the loops, runs out of its 168 registers
    // and reloads the spilled addresses inside the GEMM loops -- each reload behind an s_waitcnt vmcnt(0) that also drains
    // the weight ring.
    auto refresh_ids = [&]() {
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));
        tid = t; ln = t & 63; lr = ln & 31; lh = ln >> 5;
        prow = 16 * ((lr >> 2) & 1) + 4 * (lr >> 3) + (lr & 3);
        row0 = mt0 * 32 + prow; row1 = two ? row0 + 32 : row0;
        mix_active = lh == ((lr >> 2) & 1);
        mix_step = lr >> 4;
    };

    // ---- weight-fragment ring --------------------------------------------------------------------
    // stream k: fwd layers 1..L-1 (W1^T, W2^T), then bwd layers L-1..1 (W1, W2), then layer 0 bwd (column tile 0 only)
    const int nfwd = 2 * (L - 1);
    auto stream = [&](int k) -> const float * {
        if (k < nfwd) { const int l = 1 + (k >> 1); return P16 + ((k & 1) ? lay.w2f[l] : lay.w1f[l]); }
        const int kk = k - nfwd;
        if (kk < nfwd) { const int l = L - 1 - (kk >> 1); return P16 + ((kk & 1) ? lay.w2b[l] : lay.w1b[l]); }
        if (kk < nfwd + 2 && ct == 0) return P16 + ((kk & 1) ? lay.w2b[0] : lay.w1b[0]);
        // past the end (and layer 0's backward matrices for the waves that do not use them): a valid dummy.  Every ring
        // load is issued unconditionally -- with a load under a condition the compiler loses count of the loads in
        // flight and every later s_waitcnt vmcnt drains the whole ring.
        return P16 + lay.w1f[1];
    };
    v4f rb[3][2];                            // [slot][hi | lo]: fragments of three consecutive 16-k steps
#pragma unroll
    for (int i = 0; i < 3; ++i) { rb[i][0] = v4f{0.f, 0.f, 0.f, 0.f}; rb[i][1] = rb[i][0]; }
    auto ring_load = [&](const float *F, int s, int slot) {
        const float *p = F + ((size_t)(ct * (GH / 8) + 2 * s) * 64 + ln) * 4;
        rb[slot][0] = *(gv4p)p;
        rb[slot][1] = *(gv4p)(p + 256);
    };
    auto ring_prime = [&](const float *F) {
#pragma unroll
        for (int s = 0; s < 3; ++s) ring_load(F, s, s);
    };
    // Two 32x32 output tiles (the wave's row tiles) against stream `Fc`, K = 144: acc_t = A_t B^T.  The ring holds steps
    // 0..2 of Fc at entry and steps 0..2 of `Fn` (if any) at exit.  The A rows are split-pair rows of Hb.
    auto gemm2 = [&](const float *Fc, const float *Fn, f32x16 &acc0, f32x16 &acc1) {
#if DNAG_PRED_SKIP & 2
        {                                    // timing ablation (wrong results)
            for (int i = 0; i < 16; ++i) { acc0[i] = Hb[row0 * HS + i]; acc1[i] = Hb[row1 * HS + i]; }
            return;
        }
#endif
        const float *a0p = Hb + row0 * HS + lh * (GH / 2), *a1p = Hb + row1 * HS + lh * (GH / 2);
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
        // software pipeline, pinned by compiler barriers (left alone, the scheduler sinks every ring load to just before
        // its use and waits a full L2 round trip per step): the A fragments of step s + 1 are read from LDS before the
        // MFMAs of step s, the weight fragments of step s + 3 are requested right behind them
        h8 af[2][4];
        auto aread = [&](int s, int buf) {
            af[buf][0] = *reinterpret_cast<const h8 *>(a0p + 8 * s); af[buf][1] = *reinterpret_cast<const h8 *>(a0p + 8 * s + 4);
            af[buf][2] = *reinterpret_cast<const h8 *>(a1p + 8 * s); af[buf][3] = *reinterpret_cast<const h8 *>(a1p + 8 * s + 4);
        };
        aread(0, 0);
#pragma unroll
        for (int s = 0; s < GH / 16; ++s) {
            const int slot = s % 3, cur = s & 1;
            if (s + 1 < GH / 16) aread(s + 1, cur ^ 1);
            const h8 bh = __builtin_bit_cast(h8, rb[slot][0]), bl = __builtin_bit_cast(h8, rb[slot][1]);
            acc0 = mfma16(af[cur][0], bh, acc0);
            acc0 = mfma16(af[cur][0], bl, acc0);
            acc0 = mfma16(af[cur][1], bh, acc0);
            if (two) {
                acc1 = mfma16(af[cur][2], bh, acc1);
                acc1 = mfma16(af[cur][2], bl, acc1);
                acc1 = mfma16(af[cur][3], bh, acc1);
            }
            if (s + 3 < GH / 16) ring_load(Fc, s + 3, slot);
            else ring_load(Fn, s + 3 - GH / 16, slot);
            asm volatile("" ::: "memory");
        }
    };
    static_assert((GH / 16) % 3 == 0, "the ring slots of consecutive streams line up");

    // Adjacency mixing from the accumulator registers.  `yv`: one column (lane) x 16 rows = graphs 2 lh, 2 lh + 1 of the
    // tile, 8 nodes each (registers 0..7, 8..15).  Two k steps: step s contracts over the nodes of graph 2 kh + s of
    // every k half kh.  B operand of lane (rho' = lr, kh = lh): the adjacency row of rho' -- `arow`, this lane's own
    // pre-split row of matrix `mat` -- in the one step and k half that hold rho's graph, zeros elsewhere.
    // `ms` = {sy, 1 / (1024 sy)}: the power-of-two scale of yv (a bound from the weights, PredLayout::mixs) and what
    // divides it and the adjacency's 1024 out again (applied by the caller).  Result: lane = row rho', registers =
    // columns 8q + 4 lh + e of the wave's tile.
    auto mixf = [&](const f32x16 &yv, const float *arow, const float *ms) -> f32x16 {
#if DNAG_PRED_SKIP & 1
        return yv;                           // timing ablation (wrong results)
#endif
        const float sy = ms[0];
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            h8 ah, al;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = yv[8 * s + e] * sy;
                const _Float16 hi = (_Float16)v;
                ah[e] = hi;
                al[e] = (_Float16)(v - (float)hi);
            }
            const float *bp = (mix_active && mix_step == s) ? arow : zrow;
            const h8 bh = *reinterpret_cast<const h8 *>(bp), bl = *reinterpret_cast<const h8 *>(bp + 4);
            acc = mfma16(ah, bh, acc);
            acc = mfma16(ah, bl, acc);
            acc = mfma16(al, bh, acc);
        }
        return acc;                          // the caller multiplies by ms[1] (and by its own scale) in one go
    };
    auto arow_of = [&](int mat, int row) { return adjop + ((size_t)mat * R + row) * 8; };
    // the lane's 16 values of a row (columns ct*32 + 8q + 4 lh + e), times `sc`, as split-pair quads into Hb
    auto put_rows = [&](const f32x16 &v, int row, float sc) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c0 = ct * 32 + 8 * q + 4 * lh;
            if (c0 < GH) {
                h4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float s_ = v[4 * q + e] * sc;
                    const _Float16 a_ = (_Float16)s_;
                    hi[e] = a_;
                    lo[e] = (_Float16)(s_ - (float)a_);
                }
                _Float16 *d = reinterpret_cast<_Float16 *>(Hb + row * HS) + (c0 >> 3) * 16 + (c0 & 7);
                *reinterpret_cast<h4 *>(d) = hi;
                *reinterpret_cast<h4 *>(d + 8) = lo;
            }
        }
    };
    auto col_valid = [&](int i) { return ct * 32 + 8 * (i >> 2) + 4 * lh + (i & 3) < GH; };

    // ---- set-up ------------------------------------------------------------------------------------
    if (tid < 32) amaxv[tid] = 0u;
    ring_prime(stream(0));
    v4f f0[2][2];                            // layer 0's weight fragments (K = 16: one step): [W1 | W2][hi | lo]
#pragma unroll
    for (int w = 0; w < 2; ++w) {
        const float *F = P + (w ? lay.w2f0 : lay.w1f0) + ((size_t)(ct * 2) * 64 + ln) * 4;
        f0[w][0] = *(gv4p)F;
        f0[w][1] = *(gv4p)(F + 256);
    }
    float xr[V];
#pragma unroll
    for (int v = 0; v < V; ++v) xr[v] = 0.f;
    if (tid < R && g0 + tid / N < batch) {
#pragma unroll
        for (int v = 0; v < V; ++v) xr[v] = x[((size_t)g0 * N + tid) * V + v];
    }
    float *Ah = Hb;                          // scratch for Ahat = rn(M + I)   (digcn_meta.py:120)
    if (tid < G * N) {
        const int g = tid / N, r = tid - g * N;
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
    {
        float xm = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) xm = fmaxf(xm, fabsf(xr[v]));
        note_max(xm, AM_X);
    }
    // A1 = rn(Ahat), A2 = rn(Ahat^T)   (digcn_meta.py:42, :44)
    if (tid < G * N) {
        const int g = tid / N, r = tid - g * N;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < N; ++j) { s1 += Ah[(g * N + r) * N + j]; s2 += Ah[(g * N + j) * N + r]; }
#pragma unroll
        for (int j = 0; j < N; ++j) {
            Araw[0][(g * N + r) * N + j] = Ah[(g * N + r) * N + j] / s1;
            Araw[1][(g * N + r) * N + j] = Ah[(g * N + j) * N + r] / s2;
        }
    }
    __syncthreads();     // Ah (aliasing Hb) is dead from here
    PRED_STAMP(2);
    // operand rows of the mixing: mat 0 / 1 = rows of A1 / A2 (forward), mat 2 / 3 = rows of A1^T / A2^T (backward)
    if (tid < 4 * R) {
        const int mat = tid / R, rr = tid - mat * R, g = rr / N, n = rr - g * N;
        const float *A = Araw[mat & 1] + g * N * N;
        h8 hi, lo;
#pragma unroll
        for (int t = 0; t < N; ++t) {
            const float v = (mat < 2 ? A[n * N + t] : A[t * N + n]) * ADJ_SCALE;
            const _Float16 a_ = (_Float16)v;
            hi[t] = a_;
            lo[t] = (_Float16)(v - (float)a_);
        }
        float *d = adjop + ((size_t)mat * R + rr) * 8;
        *reinterpret_cast<h8 *>(d) = hi;
        *reinterpret_cast<h8 *>(d + 4) = lo;
    }
    if (tid < 8) zrow[tid] = 0.f;
    // layer 0's A operand: x as split-pair rows with K = 16 (k 7..15 zero): dwords 0..7 = k half 0 (hi | lo), 8..15 = k half 1
    const float sx = slot_scale(AM_X);
    if (tid < R) {
        h8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = e < V ? xr[e] * sx : 0.f;
            const _Float16 a_ = (_Float16)v;
            hi[e] = a_;
            lo[e] = (_Float16)(v - (float)a_);
        }
        float *d = Hb + tid * HS;
        *reinterpret_cast<h8 *>(d) = hi;
        *reinterpret_cast<h8 *>(d + 4) = lo;
        *reinterpret_cast<v4f *>(d + 8) = v4f{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<v4f *>(d + 12) = v4f{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    PRED_STAMP(3);

    // ---- forward -----------------------------------------------------------------------------------
    f32x16 h0, h1;                           // the wave's rows of H_{l+1} (lane = row, 16 columns), then of dH
    int sidx = 0;                            // next weight stream
    float sa = sx;                           // scale of the operand in Hb
#pragma unroll
    for (int i = 0; i < 16; ++i) h1[i] = 0.f;           // waves with one row tile never touch tile 1
    for (int l = 0; l < L; ++l) {
        refresh_ids();
        f32x16 r0, r1;                       // relu(Z1)
        uint32_t w0 = 0u, w1 = 0u;
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            f32x16 y0, y1;
            if (l == 0) {                    // K = 16: one step, fragments prefetched at entry
                const h8 bh = __builtin_bit_cast(h8, f0[w][0]), bl = __builtin_bit_cast(h8, f0[w][1]);
                const float *a0p = Hb + row0 * HS + lh * 8, *a1p = Hb + row1 * HS + lh * 8;
                const h8 a0h = *reinterpret_cast<const h8 *>(a0p), a0l = *reinterpret_cast<const h8 *>(a0p + 4);
#pragma unroll
                for (int i = 0; i < 16; ++i) { y0[i] = 0.f; y1[i] = 0.f; }
                y0 = mfma16(a0h, bh, y0);
                y0 = mfma16(a0h, bl, y0);
                y0 = mfma16(a0l, bh, y0);
                if (two) {
                    const h8 a1h = *reinterpret_cast<const h8 *>(a1p), a1l = *reinterpret_cast<const h8 *>(a1p + 4);
                    y1 = mfma16(a1h, bh, y1);
                    y1 = mfma16(a1h, bl, y1);
                    y1 = mfma16(a1l, bh, y1);
                }
            } else {
                gemm2(stream(sidx), stream(sidx + 1), y0, y1);
                ++sidx;
            }
            const float *ms = MIXS + (2 * l + w) * 4;
            const float inv = WSC[2 * l + w] * __builtin_amdgcn_rcpf(sa) * ms[1];    // sa and the mixing's scales are powers of two: exact
            auto epi = [&](const f32x16 &z, f32x16 &r, f32x16 &h, uint32_t &wb) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float a_ = z[i] * inv;
                    wb |= (a_ > 0.f ? 1u : 0u) << (16 * w + i);
                    if (w == 0) r[i] = fmaxf(a_, 0.f);
                    else h[i] = (r[i] + fmaxf(a_, 0.f)) * 0.5f;
                }
            };
            epi(mixf(y0, arow_of(w, row0), ms), r0, h0, w0);
            if (two) epi(mixf(y1, arow_of(w, row1), ms), r1, h1, w1);
        }
        mbits[(l * 2 + 0) * NT + tid] = w0;
        mbits[(l * 2 + 1) * NT + tid] = w1;
        if (l + 1 < L) {
            float hmax = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) if (col_valid(i)) hmax = fmaxf(hmax, fmaxf(h0[i], h1[i]));
            note_max(hmax, AM_H + l + 1);
        }
        __syncthreads();                     // every wave is done reading Hb; the maximum is complete
        PRED_STAMP(4);
        if (l + 1 < L) {
            sa = slot_scale(AM_H + l + 1);
            put_rows(h0, row0, sa);
            if (two) put_rows(h1, row1, sa);
            __syncthreads();
            PRED_STAMP(5);
        }
    }

    // ---- head forward: graphs become the rows 0..G-1 of a 32-row MFMA tile ---------------------------
    {
        refresh_ids();
        // node sum of a graph = sum over lane bits 0, 1, 3 of rho' (digcn_meta.py:19-22)
        float vmax = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (t == 1 && !two) break;
            const f32x16 &hv = t ? h1 : h0;
            const int gi = (mt0 + t) * 4 + 2 * ((lr >> 2) & 1) + (lr >> 4);
            const int gg = (g0 + gi < batch) ? g0 + gi : 0;
            // the time embedding of the lane's 16 columns, fetched by every lane up front (column clamped): one load per
            // value inside the condition below cost an L2 round trip each, sixteen in a row
            v4f te[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c0_ = ct * 32 + 8 * q + 4 * lh;
                te[q] = *(gv4p)(temb + (size_t)(tstride ? gg : 0) * GH + (c0_ < GH ? c0_ : 0));
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float s = hv[i];
                s += dpp_mov<0xB1>(s);      // quad_perm [1,0,3,2]: lane ^ 1
                s += dpp_mov<0x4E>(s);      // quad_perm [2,3,0,1]: lane ^ 2
                s += dpp_mov<0x128>(s);     // row_ror 8: lane ^ 8 within the row of 16
                const int c = ct * 32 + 8 * (i >> 2) + 4 * lh + (i & 3);
                if ((lr & 11) == 0 && c < GH) {
                    const float pv = s * (1.0f / (float)N) + te[i >> 2][i & 3];   // :125
                    hA[gi * HS + c] = pv;
                    vmax = fmaxf(vmax, fabsf(pv));
                }
            }
        }
        note_max(vmax, AM_P);
        __syncthreads();
        PRED_STAMP(8);
        auto id_xf = [](v4f a, int) { return a; };
        // a = Wfc1 p  (fc1 [LH][GH]: rows K-contiguous) -> hB[g][0..127].  Rows G..31 of the tile hold stale values: an
        // MFMA output row depends on its own A row only and the epilogues drop those rows.
        {
            float amx = 0.f;
            run_jobs<GH, 1, PRED_NW, true>(
                LH / 32, hA, HS,
                [&](int nt, int, int, float (&b)[GH / 2]) { load_frag<GH>(P16 + lay.fc1f, nt, b); },
                id_xf,
                [&](int, int nt, int r, int h, const f32x16 &acc) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int row = acc_row(i, h);
                        if (row < G) { hB[row * HS + nt * 32 + r] = acc[i]; amx = fmaxf(amx, fabsf(acc[i])); }
                    }
                }, slot_scale(AM_P), WSC[2 * L]);
            note_max(amx, AM_A);
        }
        __syncthreads();
        PRED_STAMP(9);
        // tanh(W0a a + c0)  (p0f: fragments of W0a [hs][LH]) -> tv[g][j]
        run_jobs<LH, 1, PRED_NW, true>(
            (hs + 31) / 32, hB, HS,
            [&](int nt, int, int, float (&b)[LH / 2]) { load_frag<LH>(P + lay.p0f, nt, b); },
            id_xf,
            [&](int, int nt, int r, int h, const f32x16 &acc) {
                const int j = nt * 32 + r;
                if (j < hs) {
                    const float cj = c0g[j];
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int row = acc_row(i, h);
                        if (row < G) tv[row * TS + j] = tanhf(acc[i] + cj);
                    }
                }
            }, slot_scale(AM_A), WSC[2 * L + 1]);
        __syncthreads();
        PRED_STAMP(10);
        // y = w2 . tanh + b2 and dpre = w2 (1 - tanh^2) in one pass: 16 lanes per graph
        {
            float dmax = 0.f;
            if (tid < G * 16) {
                const int g = tid >> 4, sub = tid & 15;
                float ys = 0.f;
                for (int j = sub; j < hs; j += 16) {
                    const float t = tv[g * TS + j], w2 = P[lay.p2_w + j];
                    ys = fmaf(w2, t, ys);
                    const float d = w2 * (1.f - t * t);
                    tv[g * TS + j] = d;
                    dmax = fmaxf(dmax, fabsf(d));
                }
                for (int j = hs + sub; j < lay.hs16; j += 16) tv[g * TS + j] = 0.f;     // K padding of the da product
                ys = row16_sum(ys);
                if (sub == 0 && y != nullptr && g0 + g < batch) y[g0 + g] = ys + P[lay.p2_b];
            }
            note_max(dmax, AM_D);
        }
        if (grad == nullptr) return;
        __syncthreads();
        PRED_STAMP(11);
        // ---- head backward: da[o] = sum_j W0a[j][o] dpre[j]  (p0b: fragments of W0a^T [LH][hs16], K = hs16 at run time)
        {
            const float sd = slot_scale(AM_D);
            const int K = lay.hs16;
            float amx = 0.f;
            if (wv < LH / 32) {
                const int nt = wv;
                const float *ap = tv + (lr & (G - 1)) * TS + lh * (K / 2);        // rows >= G repeat the first ones (dropped below)
                const float *bp = P + lay.p0b + ((size_t)nt * (K / 8) * 64 + ln) * 4;
                f32x16 acc;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
                for (int s = 0; s < K / 16; ++s) {
                    const v4f a0 = *reinterpret_cast<const v4f *>(ap + 8 * s), a1 = *reinterpret_cast<const v4f *>(ap + 8 * s + 4);
                    h8 ah, al;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float v = (e < 4 ? a0[e] : a1[e - 4]) * sd;
                        const _Float16 hi = (_Float16)v;
                        ah[e] = hi;
                        al[e] = (_Float16)(v - (float)hi);
                    }
                    const h8 bh = __builtin_bit_cast(h8, *(gv4p)(bp + (size_t)(2 * s) * 256));
                    const h8 bl = __builtin_bit_cast(h8, *(gv4p)(bp + (size_t)(2 * s + 1) * 256));
                    acc = mfma16(ah, bh, acc);
                    acc = mfma16(ah, bl, acc);
                    acc = mfma16(al, bh, acc);
                }
                const float inv = WSC[2 * L + 1] * __builtin_amdgcn_rcpf(sd);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = acc_row(i, lh);
                    if (row < G) { const float d = acc[i] * inv; hA[row * HS + nt * 32 + lr] = d; amx = fmaxf(amx, fabsf(d)); }
                }
            }
            note_max(amx, AM_DA);
        }
        __syncthreads();
        PRED_STAMP(12);
        // dp = Wfc1^T da  (fc1b: fragments of fc1^T [GH][LH]);  dH_L[n][c] = dp[c] / N for every node n -> hB[g][c]
        {
            float dmax = 0.f;
            run_jobs<LH, 1, PRED_NW, true>(
                PRED_NT, hA, HS,
                [&](int nt, int, int, float (&b)[LH / 2]) { load_frag<LH>(P16 + lay.fc1b, nt, b); },
                id_xf,
                [&](int, int nt, int r, int h, const f32x16 &acc) {
                    const int col = nt * 32 + r;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int g = acc_row(i, h);
                        if (g < G && col < GH) {
                            const float d = acc[i] * (1.0f / (float)N);
                            hB[g * HS + col] = d;
                            dmax = fmaxf(dmax, fabsf(d));
                        }
                    }
                }, slot_scale(AM_DA), WSC[2 * L]);
            note_max(dmax, AM_DH + L - 1);
        }
        __syncthreads();
        PRED_STAMP(13);
        // every lane fetches dH_L of its two rows' graphs for its 16 columns
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (t == 1 && !two) break;
            f32x16 &dv = t ? h1 : h0;
            const int gi = (mt0 + t) * 4 + 2 * ((lr >> 2) & 1) + (lr >> 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c0 = ct * 32 + 8 * q + 4 * lh;
                v4f d4 = v4f{0.f, 0.f, 0.f, 0.f};
                if (c0 < GH) d4 = *reinterpret_cast<const v4f *>(hB + gi * HS + c0);
                dv[4 * q + 0] = d4.x; dv[4 * q + 1] = d4.y; dv[4 * q + 2] = d4.z; dv[4 * q + 3] = d4.w;
            }
        }
        __syncthreads();                     // hB (inside Hb) is overwritten by the first backward operand
        PRED_STAMP(14);
    }

    // ---- backward: layers L-1 .. 0 -------------------------------------------------------------------
    sidx = nfwd;
    for (int l = L - 1; l >= 0; --l) {
        refresh_ids();
        const float sd = slot_scale(AM_DH + l);
        const uint32_t w0 = mbits[(l * 2 + 0) * NT + tid], w1 = mbits[(l * 2 + 1) * NT + tid];
        f32x16 e0, e1;
#pragma unroll
        for (int i = 0; i < 16; ++i) e1[i] = 0.f;
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            {   // operand: dH . [Z_w > 0] (the factor 1/2 joins the result)
                f32x16 m0;
#pragma unroll
                for (int i = 0; i < 16; ++i) m0[i] = ((w0 >> (16 * w + i)) & 1u) ? h0[i] : 0.f;
                put_rows(m0, row0, sd);
                if (two) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) m0[i] = ((w1 >> (16 * w + i)) & 1u) ? h1[i] : 0.f;
                    put_rows(m0, row1, sd);
                }
            }
            __syncthreads();
            PRED_STAMP(16);
            if (l > 0 || ct == 0) {          // layer 0: dx has V <= 32 columns, one column tile
                f32x16 y0, y1;
                gemm2(stream(sidx + w), stream(sidx + w + 1), y0, y1);
                const float *ms = MIXS + (2 * l + w) * 4 + 2;
                const float inv = 0.5f * WSC[2 * l + w] * __builtin_amdgcn_rcpf(sd) * ms[1];   // powers of two: exact
                const f32x16 z0 = mixf(y0, arow_of(2 + w, row0), ms);
#pragma unroll
                for (int i = 0; i < 16; ++i) e0[i] = w == 0 ? z0[i] * inv : fmaf(z0[i], inv, e0[i]);
                if (two) {
                    const f32x16 z1 = mixf(y1, arow_of(2 + w, row1), ms);
#pragma unroll
                    for (int i = 0; i < 16; ++i) e1[i] = w == 0 ? z1[i] * inv : fmaf(z1[i], inv, e1[i]);
                }
            }
            if (w == 0 || l > 0) {
                if (w == 1) {
                    float dmax = 0.f;
#pragma unroll
                    for (int i = 0; i < 16; ++i) if (col_valid(i)) dmax = fmaxf(dmax, fmaxf(fabsf(e0[i]), fabsf(e1[i])));
                    note_max(dmax, AM_DH + l - 1);
                }
                __syncthreads();             // every wave is done reading Hb (and the next maximum is complete)
                PRED_STAMP(17);
            }
        }
        sidx += 2;
        if (l > 0) { h0 = e0; h1 = e1; }
        else if (ct == 0) {
            // dx: lane = row, columns 8q + 4 lh + e < V
#pragma unroll
            for (int t = 0; t < 2; ++t) {             // column tile 0 belongs to waves 0 and 1: two row tiles each
                const f32x16 &ev = t ? e1 : e0;
                const int row = t ? row1 : row0;
                if (g0 + row / N < batch) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int c = 8 * (i >> 2) + 4 * lh + (i & 3);
                        if (c < V) grad[((size_t)g0 * N + row) * V + c] = ev[i];
                    }
                }
            }
        }
    }
#ifdef DNAG_PRED_STAMPS
    __syncthreads();
    PRED_STAMP(99);
    if (blockIdx.x == 5 && tid == 0) for (int i = 0; i < nts_; ++i) grad[((size_t)g0 * N) * V + i] = (float)(ts_[i] - ts_[0]);
#endif
}

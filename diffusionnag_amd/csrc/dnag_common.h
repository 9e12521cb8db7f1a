// Shared declarations of the dnag HIP library: packed-weight layouts and device helpers.
// gfx950 (CDNA4) only: wave = 64 lanes, fp32-input MFMA (v_mfma_f32_32x32x2_f32).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/dnag.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));   // native 16-byte vector (HIP float4 is a struct that SROA may leave in scratch)

#define DNAG_D 64          // d_model of every transformer on the path (configs/tr_scorenet.py:96)

static inline __host__ __device__ size_t align4(size_t x) { return (x + 3) & ~(size_t)3; }

// ---------------------------------------------------------------------------------------------
// Kernel layout of the CATE score network (offsets in floats into the packed blob).
// Per layer (LayerNorm affine of self_attn.norm / feed_forward.norm folded into Wqkv / W1):
//   Wqkv'[3d][d] bqkv'[3d] Wo[d][d] bo[d] W1'[dff][d] b1'[dff] W2[d][dff] b2[d] on_g[d] on_b[d]
// The four matrices are stored FRAGMENT-MAJOR (frag_index below): the 64 lanes of a wave read one
// contiguous KiB per load instruction instead of 64 separate cache lines.
// ---------------------------------------------------------------------------------------------
struct CateLayout {
    int N, V, d, H, dff, L;
    size_t emb_w, emb_b;          // opEmb.fc [d][V], [d]
    size_t t1_w, t1_b, t2_w, t2_b;  // timeEmb1 [4d][d],[4d]; timeEmb2 [d][4d],[d]
    size_t layer0, layer_stride;
    size_t l_wqkv, l_bqkv, l_wo, l_bo, l_w1, l_b1, l_w2, l_b2, l_ong, l_onb;  // relative to a layer base
    size_t l_sc;                  // per layer: 8 power-of-two scale factors of the split arithmetic (ENC_SC_*)
    size_t l_vec;                 // per layer: contiguous copy bqkv' [3d] | bo [d] | b1' [dff] | b2 [d] | on_g [d] | on_b [d] (ENC_VEC_*),
                                  // what encoder8_kernel stages into LDS with one 16-byte load per thread
    size_t f0_w, f0_b, f1_w, f1_b, f2_w, f2_b;  // final.linears.{0,1,2}
    size_t pe, freq;              // [N][d], [d/2]
    size_t f0_s, f1_s;            // final.linears.{0,1} weights as split-pair fp16 rows (dnag_gemm16.hip.h), same sizes.  f0_s holds
                                  // W0 diag(gamma) with the affine of the per-layer output norms (encoder_f.norms.l) folded in: the
                                  // encoders write the concat WITHOUT that affine (LN(h) * 16, the very rows the next layer's QKV reads)
    size_t f0_bs;                 // [Hd] b0 + W0 beta: the bias that goes with f0_s
    size_t hsc;                   // 16 power-of-two scale factors of the head GEMMs (HEAD_SC_*)
    size_t delta16;               // layer0 + delta16: second copy of the layer blocks whose four matrices are fp16 hi/lo
                                  // fragments scaled by 256 (frag16_pack) for the encoder's split-arithmetic MFMAs
    size_t total;
    int F, Hd;                    // F = L*d (concat width), Hd = 2F (head hidden)
};

static inline __host__ __device__ CateLayout cate_layout(const dnag_cate_dims &c) {
    CateLayout y;
    y.N = c.max_node; y.V = c.n_vocab; y.d = c.d_model; y.H = c.n_head; y.dff = c.d_ff; y.L = c.n_layers;
    const size_t d = y.d, dff = y.dff, V = y.V;
    y.F = y.L * y.d; y.Hd = 2 * y.F;
    size_t o = 0;
    y.emb_w = o; o += align4(d * V);
    y.emb_b = o; o += d;
    y.t1_w = o; o += 4 * d * d;
    y.t1_b = o; o += 4 * d;
    y.t2_w = o; o += 4 * d * d;
    y.t2_b = o; o += d;
    y.layer0 = o;
    size_t l = 0;
    y.l_wqkv = l; l += 3 * d * d;
    y.l_bqkv = l; l += 3 * d;
    y.l_wo = l; l += d * d;
    y.l_bo = l; l += d;
    y.l_w1 = l; l += dff * d;
    y.l_b1 = l; l += align4(dff);
    y.l_w2 = l; l += d * dff;
    y.l_b2 = l; l += d;
    y.l_ong = l; l += d;
    y.l_onb = l; l += d;
    y.l_sc = l; l += 8;
    y.l_vec = l; l += 3 * d + d + align4(dff) + d + d + d;
    y.layer_stride = l;
    o += l * y.L;
    const size_t F = y.F, Hd = y.Hd;
    y.f0_w = o; o += Hd * F;
    y.f0_b = o; o += Hd;
    y.f1_w = o; o += Hd * Hd;
    y.f1_b = o; o += Hd;
    y.f2_w = o; o += align4(V * Hd);
    y.f2_b = o; o += align4(V);
    y.pe = o; o += align4((size_t)y.N * d);
    y.freq = o; o += align4(d / 2);
    y.f0_s = o; o += Hd * F;
    y.f1_s = o; o += Hd * Hd;
    y.f0_bs = o; o += Hd;
    y.hsc = o; o += 16;
    y.delta16 = o - y.layer0; o += y.layer_stride * y.L;
    y.total = o;
    return y;
}

// ---------------------------------------------------------------------------------------------
// Operand scales of the split arithmetic (dnag_gemm16.hip.h).  Every operand of a split MFMA product is multiplied
// by a power of two before it is cut into fp16 hi + lo, and the product of the two scales is divided out of the fp32
// accumulator (exactly).  The scales are chosen when the weights are packed, from the weights alone, so that no fp16
// value can overflow whatever the input is:
//   weights      sw = 2^k <= 256 with sw * max|w| <= SPLIT_LIMIT (256 for any matrix below |w| = 234);
//   activations  sa = 2^k <= 16  with sa * bound   <= SPLIT_LIMIT, where `bound` is a worst-case magnitude derived
//                from the producing layer's weights: 8 for a LayerNorm output without affine (|u_i| < sqrt(64)),
//                8 max|gamma| + max|beta| with affine, max_c (8 ||W_c||_2 + |b_c|) for a linear on a LayerNorm output
//                (Cauchy-Schwarz, ||u||_2 <= 8; attention outputs are convex combinations of such values),
//                max_c (||W_c||_1 amax_in + |b_c|) for a linear on a bounded input; ReLU / ELU do not raise a bound >= 1.
//                Dropout: the sampling path runs in eval mode (models/utils.py:63-65); the training forward multiplies
//                the attention probabilities and the FFN hidden by 1 / (1 - p) before they are split, and
//                train_scales_kernel multiplies those two bounds by the same factor.
// Networks with ordinary weights get 256 / 16 everywhere (the only values round 1 used); a checkpoint with large
// weights gets smaller powers of two and keeps its accuracy relative to the magnitudes involved.
// ---------------------------------------------------------------------------------------------
#define SPLIT_LIMIT 60000.0f
static inline __host__ __device__ float split_pow2_scale(float bound, int kmax) {
    int k = kmax;
    if (bound > 0.f) {
        int e;
        frexpf(SPLIT_LIMIT / bound, &e);          // SPLIT_LIMIT / bound = m 2^e, m in [0.5, 1): 2^(e-1) <= ratio
        k = e - 1 < kmax ? e - 1 : kmax;
        if (k < -60) k = -60;
    }
    return ldexpf(1.0f, k);
}
static inline __host__ __device__ float weight_scale(float amax) { return split_pow2_scale(amax, 8); }
// power of two s with s * bound in [512, 1024) (1 for bound = 0): the run-time operand scale (pow2_scale below), also
// evaluated on the host for bounds that follow from the weights
static inline __host__ __device__ float pow2_scale_of(float bound) {
    if (!(bound > 0.f)) return 1.0f;
    int e;
    frexpf(bound, &e);
    int k = 10 - e;
    k = k > 40 ? 40 : (k < -40 ? -40 : k);
    return ldexpf(1.0f, k);
}
static inline __host__ __device__ float act_scale(float bound) { return split_pow2_scale(bound, 4); }

// per-layer table at l_sc
#define ENC_SC_INV_QKV 0   // 1 / (16 sw_qkv)        A = LayerNorm output, scale 16
#define ENC_SC_SA_O    1   // scale of the attention output (A operand of the output projection)
#define ENC_SC_INV_O   2   // 1 / (sa_o sw_o)
#define ENC_SC_INV_F1  3   // 1 / (16 sw_1)
#define ENC_SC_SA_F2   4   // scale of the FFN hidden
#define ENC_SC_INV_F2  5   // 1 / (sa_f2 sw_2)
// head table at hsc
#define HEAD_SC_SA_CONCAT 0   // scale of the concat rows the encoders write: 16, the scale of every LayerNorm image (|LN(h)| < 8)
#define HEAD_SC_F0        4   // final.linears.0: [0] 1 / (sa_in sw), [1] sa_out (the hidden it writes), [2], [3] unused
#define HEAD_SC_F1        8   // final.linears.1 + .2 fused: [0] 1 / (sa_in sw), [1] sa of the ELU output, [2] sw of W3, [3] 1 / ([1] [2])

// ---------------------------------------------------------------------------------------------
// Fragment-major storage of a weight matrix W[rows][K] (rows K-contiguous) that is the B operand of
// 32x32 MFMA tile jobs: tile nt covers rows nt*32 .. nt*32+31; lane l = 32*h + r of a wave owns row
// nt*32 + r and the K-half h.  Element e of chunk q of that lane sits at
//     frag_index = ((nt * (K/8) + q) * 64 + l) * 4 + e      <->   W[nt*32 + r][h*K/2 + 4*q + e]
// (rows beyond `rows` are zero).  A tile takes 32*K floats.
// ---------------------------------------------------------------------------------------------
static inline __host__ __device__ size_t frag_floats(size_t rows, size_t K) { return (rows + 31) / 32 * 32 * K; }

static inline void frag_pack(const float *W, size_t rows, size_t K, size_t ld, float *out) {
    const size_t ntiles = (rows + 31) / 32, KH = K / 2, NQ = K / 8;
    for (size_t nt = 0; nt < ntiles; ++nt)
        for (size_t q = 0; q < NQ; ++q)
            for (size_t l = 0; l < 64; ++l)
                for (size_t e = 0; e < 4; ++e) {
                    const size_t row = nt * 32 + (l & 31), k = (l >> 5) * KH + 4 * q + e;
                    out[((nt * NQ + q) * 64 + l) * 4 + e] = row < rows ? W[row * ld + k] : 0.f;
                }
}

// Split-arithmetic variant of frag_pack (dnag_gemm16.hip.h explains the arithmetic): the same fragment-major
// addressing and byte size, but the 16-byte chunk q of a lane holds 8 fp16 values: chunks 2s and 2s+1 are the hi and
// the lo halves of sw * W[nt*32 + r][h*K/2 + 8s .. 8s+7], i.e. the B fragments of 32x32x16 step s (sw: weight_scale
// of the matrix).
static inline void frag16_pack(const float *W, size_t rows, size_t K, size_t ld, float *out, float sw = 256.0f) {
    const size_t ntiles = (rows + 31) / 32, KH = K / 2, NQ = K / 8;
    _Float16 *o = reinterpret_cast<_Float16 *>(out);
    for (size_t nt = 0; nt < ntiles; ++nt)
        for (size_t q = 0; q < NQ; ++q)
            for (size_t l = 0; l < 64; ++l)
                for (size_t j = 0; j < 8; ++j) {
                    const size_t row = nt * 32 + (l & 31), k = (l >> 5) * KH + 8 * (q >> 1) + j;
                    const float w = row < rows ? W[row * ld + k] * sw : 0.f;
                    const _Float16 hi = (_Float16)w;
                    o[((nt * NQ + q) * 64 + l) * 8 + j] = (q & 1) ? (_Float16)(w - (float)hi) : hi;
                }
}

// ---------------------------------------------------------------------------------------------
// Kernel layout of the MetaNeuralPredictor surrogate.
//   gcn 0: W1, W2 [V][g] plain (forward on FMAs);
//   gcn l: w1b/w2b = fragment-major W[in][out] (backward: dH_l = P W^T sums over `out`),
//          w1f/w2f = fragment-major W^T[out][in] (forward, l >= 1);
//   fc1f = fragment-major fc1 [lh][g] (a = fc1 p), fc1b = fragment-major fc1^T [g][lh] (dp = fc1^T da).
// fp16 hi/lo fragments only (frag16_pack, no fp32 twin): w1f0 / w2f0 = gcn.0 W^T [g][V -> 16] (layer 0 forward as one
// 32x32x16 step), p0f = arch half of pred_fc.0 [hs][lh] (tanh layer), p0b = its transpose [lh][hs -> hs16] (da = W0a^T dpre).
// ---------------------------------------------------------------------------------------------
#define DNAG_MAX_GCN 8
struct PredLayout {
    int N, V, g, L, lh, hs, nz;
    size_t w1_0, w2_0;            // gcn.0 weights [V][g], plain
    size_t w1b[DNAG_MAX_GCN], w2b[DNAG_MAX_GCN], w1f[DNAG_MAX_GCN], w2f[DNAG_MAX_GCN];
    size_t fc1f, fc1b;
    size_t t1_w, t1_b, t2_w, t2_b;  // [4g][g],[4g],[g][4g],[g]
    size_t p0_w, p0_b;            // pred_fc.0 [hs][nz+lh], [hs]
    size_t p0_at;                 // transposed arch part [lh][hs]
    size_t p2_w, p2_b;            // pred_fc.2 [hs], [1]
    size_t freq;                  // [g/2]
    size_t wsc;                   // 1 / sw of the split-arithmetic matrices: [2 l + {0: W1_l, 1: W2_l}], [2 L]: fc1, [2 L + 1]: pred_fc.0
    size_t w1f0, w2f0, p0f, p0b;  // fp16 hi/lo fragments (see above)
    size_t mixs;                  // [l][w][4]: power-of-two scale of the mixing's A operand (the raw accumulator of the GEMM before it)
                                  // and its reciprocal over 1024, forward then backward: a BOUND from the weights alone, since the
                                  // GEMM's own A operand is always scaled to below 1024: |acc| <= 1024 sw max_c ||W_c||_1
    int hs16;                     // hs rounded up to a multiple of 16 (K of the da product)
    size_t delta16;               // P + delta16 + (offset of w1b .. fc1b): the same fragment-major matrices as fp16 hi/lo
                                  // fragments scaled by 256 (frag16_pack) for the split-arithmetic MFMAs
    size_t total;
};

static inline __host__ __device__ PredLayout pred_layout(const dnag_pred_dims &c) {
    PredLayout y;
    y.N = c.max_node; y.V = c.n_vocab; y.g = c.gcn_hidden; y.L = c.gcn_layers; y.lh = c.linear_hidden;
    y.hs = c.hs; y.nz = c.nz;
    const size_t g = y.g, lh = y.lh, hs = y.hs, nz = y.nz;
    size_t o = 0;
    y.w1_0 = o; o += align4((size_t)y.V * g);
    y.w2_0 = o; o += align4((size_t)y.V * g);
    for (int l = 0; l < y.L && l < DNAG_MAX_GCN; ++l) {
        const size_t in = l == 0 ? (size_t)y.V : g;
        y.w1b[l] = o; o += frag_floats(in, g);
        y.w2b[l] = o; o += frag_floats(in, g);
        y.w1f[l] = o; o += l == 0 ? 0 : frag_floats(g, in);
        y.w2f[l] = o; o += l == 0 ? 0 : frag_floats(g, in);
    }
    y.fc1f = o; o += frag_floats(lh, g);
    y.fc1b = o; o += frag_floats(g, lh);
    y.t1_w = o; o += 4 * g * g;
    y.t1_b = o; o += 4 * g;
    y.t2_w = o; o += 4 * g * g;
    y.t2_b = o; o += g;
    y.p0_w = o; o += align4(hs * (nz + lh));
    y.p0_b = o; o += align4(hs);
    y.p0_at = o; o += align4(lh * hs);
    y.p2_w = o; o += align4(hs);
    y.p2_b = o; o += 4;
    y.freq = o; o += align4(g / 2);
    y.wsc = o; o += align4(2 * DNAG_MAX_GCN + 2);
    y.delta16 = o - y.w1b[0];
    o += (y.fc1b + frag_floats(g, lh)) - y.w1b[0];
    y.hs16 = (int)((hs + 15) / 16 * 16);
    y.mixs = o; o += 8 * DNAG_MAX_GCN;
    y.w1f0 = o; o += frag_floats(g, 16);
    y.w2f0 = o; o += frag_floats(g, 16);
    y.p0f = o; o += frag_floats(hs, lh);
    y.p0b = o; o += frag_floats(lh, (size_t)y.hs16);
    y.total = o;
    return y;
}

// ---------------------------------------------------------------------------------------------
// Kernel layout of the MetaPredictorCATE surrogate (OFA).  Its embedding, encoder and `final` MLP
// have the CATE structure (L = 2, H = 2, d_ff = 32, no positional table), so that part is a
// CateLayout at offset 0.  Behind it:
//   per encoder layer, fragment-major TRANSPOSED operands of the backward pass
//     wqkvT [d][3d]  woT [d][d]  w1T [d][dff]  w2T [dff][d]     (dA = dC W  ==  dC (W^T)^T)
//   f0T [F][Hd], f1T [Hd][Hd]                  final.linears.{0,1} transposed, row-major
//   r0_w [R2P][RK] r0_b [R2P] r0T [RNP][R2P]   regeress.linears.0, zero padded (R = N V = 180,
//   r1_w [R2P][R2P] r1_b [R2P] r1T [R2P][R2P]    RK = 192, 2R = 360 -> R2P = 384, RNP = 256)
//   p0_wd [hs][nz] p0_wa [hs][R2P] p0_b [hs] p0T [R2P][hs]   pred_fc.0 split into dataset | arch
//   p2_w [hs] p2_b
// ---------------------------------------------------------------------------------------------
#define DNAG_MAX_PCL 4
struct PredCateLayout {
    CateLayout enc;
    int hs, nz, R, RK, R2, R2P, RNP;
    size_t bl[DNAG_MAX_PCL];           // per-layer base of the backward operands
    size_t b_wqkvT, b_woT, b_w1T, b_w2T;   // relative to bl[l]
    size_t f0T, f1T;
    size_t r0_w, r0_b, r0T, r1_w, r1_b, r1T;
    size_t p0_wd, p0_wa, p0_b, p0T, p2_w, p2_b;
    size_t total;
};

static inline __host__ __device__ dnag_cate_dims predcate_as_cate(const dnag_predcate_dims &c) {
    dnag_cate_dims e;
    e.max_node = c.max_node; e.n_vocab = c.n_vocab; e.d_model = c.d_model; e.n_head = c.n_head;
    e.d_ff = c.d_ff; e.n_layers = c.n_layers; e.stagewise_pos = 0;
    return e;
}

static inline __host__ __device__ size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static inline __host__ __device__ PredCateLayout predcate_layout(const dnag_predcate_dims &c) {
    PredCateLayout y;
    y.enc = cate_layout(predcate_as_cate(c));
    y.hs = c.hs; y.nz = c.nz;
    y.R = c.max_node * c.n_vocab; y.RK = (int)round_up(y.R, 32);
    y.R2 = 2 * y.R; y.R2P = (int)round_up(y.R2, 128); y.RNP = (int)round_up(y.R, 128);
    const size_t d = y.enc.d, dff = y.enc.dff, F = y.enc.F, Hd = y.enc.Hd, hs = y.hs, nz = y.nz;
    size_t o = y.enc.total;
    size_t l = 0;
    y.b_wqkvT = l; l += frag_floats(d, 3 * d);
    y.b_woT = l; l += frag_floats(d, d);
    y.b_w1T = l; l += frag_floats(d, dff);
    y.b_w2T = l; l += frag_floats(dff, d);
    for (int i = 0; i < DNAG_MAX_PCL; ++i) { y.bl[i] = o; if (i < y.enc.L) o += l; }
    y.f0T = o; o += F * Hd;
    y.f1T = o; o += Hd * Hd;
    y.r0_w = o; o += (size_t)y.R2P * y.RK;
    y.r0_b = o; o += y.R2P;
    y.r0T = o; o += (size_t)y.RNP * y.R2P;
    y.r1_w = o; o += (size_t)y.R2P * y.R2P;
    y.r1_b = o; o += y.R2P;
    y.r1T = o; o += (size_t)y.R2P * y.R2P;
    y.p0_wd = o; o += align4(hs * nz);
    y.p0_wa = o; o += hs * y.R2P;
    y.p0_b = o; o += align4(hs);
    y.p0T = o; o += (size_t)y.R2P * hs;
    y.p2_w = o; o += align4(hs);
    y.p2_b = o; o += 4;
    y.total = o;
    return y;
}

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
#ifdef __HIPCC__
// D[32x32] += A[32x2] * B[2x32].  Lane l supplies A[l&31][l>>5] and B[l>>5][l&31]; lane l holds
// D[(reg&3) + 8*(reg>>2) + 4*(l>>5)][l&31] in acc[reg]  (cdna_hip_programming.md section 3).
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }
// D[32x32] += A[32x16] * B[16x32] on fp16: lane (r, h) supplies A[r][8h .. 8h+7] and B[8h .. 8h+7][r]; same C/D layout
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x16 mfma16(h8 a, h8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// power of two s with s * amax in [512, 1024) (1 for amax = 0; capped so that products of two scales stay finite):
// operand scale of the split-arithmetic MFMAs when the magnitude of an operand is only known at run time
__device__ __forceinline__ float pow2_scale(float amax) { return pow2_scale_of(amax); }

// a * b rounded to fp32 and NOT fused into a following addition: hipcc contracts a product into its consumer by default
// (-ffp-contract=fast), through __fmul_rn / __fadd_rn and through `#pragma clang fp contract(off)` alike (both tried:
// the EMA update came out as one v_fma_f32, tests/micro/ema_dbg.py).  The empty asm hides the product from that pass.
// Used where the reference's expression rounds after every operation (predictor / corrector updates, Adam, EMA).
__device__ __forceinline__ float mul_rn(float a, float b) {
    float r = a * b;
    asm volatile("" : "+v"(r));
    return r;
}

__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : expm1f(x); }
__device__ __forceinline__ float silu(float x) { return x / (1.0f + expf(-x)); }

// Philox4x32-10 (Salmon et al. 2011); returns 4 uint32.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// One standard normal for global element index `e` of draw `draw` (draw = 0: prior, 1 + 2*step:
// corrector noise of that step, 2 + 2*step: predictor noise).  Four consecutive elements share one
// Philox block; Box-Muller on (out0,out1) and (out2,out3).
__device__ __forceinline__ float philox_normal(uint64_t seed, uint64_t e, uint32_t draw) {
    uint32_t r[4];
    const uint64_t blk = e >> 2;
    philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), draw, 0x5eedu, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const int c = (int)(e & 3);
    const uint32_t ua = (c < 2) ? r[0] : r[2], ub = (c < 2) ? r[1] : r[3];
    const float u1 = ((float)ua + 1.0f) * 2.3283064365386963e-10f;   // (0, 1]
    const float u2 = (float)ub * 2.3283064365386963e-10f;            // [0, 1]
    const float rad = sqrtf(-2.0f * logf(u1));
    const float ang = 6.283185307179586f * u2;
    return (c & 1) ? rad * sinf(ang) : rad * cosf(ang);
}
// k-tile-blocked order of split-pair rows: the 128-byte chunk (32 k) number kt of row r moves from [r][kt] to [kt][r], so
// that the rows of one k tile are contiguous (one 16 KB half tile of dnag_gemm_dma.hip.h = one contiguous run; with
// row-major rows of 6 KB every line of a k tile fell on the same two memory channels).  K % 32 == 0.
static inline __host__ __device__ size_t kblock_at(size_t r, size_t dword, size_t rows) { return ((dword >> 5) * rows + r) * 32 + (dword & 31); }
// the same for an index in units of _Float16 (two per dword)
static inline __host__ __device__ size_t kblock_half_at(size_t r, size_t half, size_t rows) { return kblock_at(r, half >> 1, rows) * 2 + (half & 1); }

#endif

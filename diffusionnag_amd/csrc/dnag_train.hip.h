// DSM training step of the CATE score network: layouts, on-device weight packing, the encoder's
// backward pass with weight gradients, and the small kernels around the tile GEMMs.
// Included by dnag_lib.hip.
//
// Restates (reference NAS-Bench-201/): losses.py:127-166 (get_sde_loss_fn_nas.loss_fn, likelihood_weighting
// = False) and the `loss.backward()` of losses.py:222-223 through models/cate.py:145-164 and
// models/transformer.py:12-179 in training mode (dropout active, models/utils.py:66-68) as a hand-derived
// backward pass.  The optimiser, gradient clipping and EMA (losses.py:22-36, models/ema.py) stay on the host.
#pragma once
#include "dnag_common.h"
#include "dnag_encoder.hip.h"

// ---------------------------------------------------------------------------------------------
// Layouts
// ---------------------------------------------------------------------------------------------
// state_dict order of CATE, offsets in floats
struct CateStateLayout {
    size_t emb_w, emb_b, t1_w, t1_b, t2_w, t2_b;
    size_t layer0, layer_stride;
    size_t s_ln1g, s_ln1b, s_wq, s_bq, s_wk, s_bk, s_wv, s_bv, s_wo, s_bo, s_ln2g, s_ln2b, s_w1, s_b1, s_w2, s_b2;  // within a layer
    size_t norms0;                 // encoder_f.norms.l.{weight,bias}: norms0 + l*2d
    size_t f0_w, f0_b, f1_w, f1_b, f2_w, f2_b;
    size_t total;
};

static inline __host__ __device__ CateStateLayout cate_state_layout(const dnag_cate_dims &c) {
    CateStateLayout y;
    const size_t d = c.d_model, V = c.n_vocab, dff = c.d_ff, L = c.n_layers, F = L * d, Hd = 2 * F;
    size_t o = 0;
    y.emb_w = o; o += d * V;
    y.emb_b = o; o += d;
    y.t1_w = o; o += 4 * d * d;
    y.t1_b = o; o += 4 * d;
    y.t2_w = o; o += 4 * d * d;
    y.t2_b = o; o += d;
    y.layer0 = o;
    size_t l = 0;
    y.s_ln1g = l; l += d;
    y.s_ln1b = l; l += d;
    y.s_wq = l; l += d * d;
    y.s_bq = l; l += d;
    y.s_wk = l; l += d * d;
    y.s_bk = l; l += d;
    y.s_wv = l; l += d * d;
    y.s_bv = l; l += d;
    y.s_wo = l; l += d * d;
    y.s_bo = l; l += d;
    y.s_ln2g = l; l += d;
    y.s_ln2b = l; l += d;
    y.s_w1 = l; l += dff * d;
    y.s_b1 = l; l += dff;
    y.s_w2 = l; l += d * dff;
    y.s_b2 = l; l += d;
    y.layer_stride = l;
    o += l * L;
    y.norms0 = o; o += L * 2 * d;
    y.f0_w = o; o += Hd * F;
    y.f0_b = o; o += Hd;
    y.f1_w = o; o += Hd * Hd;
    y.f1_b = o; o += Hd;
    y.f2_w = o; o += V * Hd;
    y.f2_b = o; o += V;
    y.total = o;
    return y;
}

// Packed training operands: the forward CateLayout, then per encoder layer the fragment-major
// TRANSPOSED matrices of the data-gradient GEMMs (LayerNorm affine folded as in the forward)
//   wqkvT [d][3d]  woT [d][d]  w1T [d][dff]  w2T [dff][d],
// then final.linears.{0,1} transposed row-major: f0T [F][Hd], f1T [Hd][Hd].
struct TrainLayout {
    CateLayout enc;
    size_t bl0, bl_stride, b_wqkvT, b_woT, b_w1T, b_w2T;
    size_t f0T, f1T;
    size_t total;
};

static inline __host__ __device__ TrainLayout train_layout(const dnag_cate_dims &c) {
    TrainLayout y;
    y.enc = cate_layout(c);
    const size_t d = y.enc.d, dff = y.enc.dff, F = y.enc.F, Hd = y.enc.Hd;
    size_t o = y.enc.total;
    size_t l = 0;
    y.b_wqkvT = l; l += frag_floats(d, 3 * d);
    y.b_woT = l; l += frag_floats(d, d);
    y.b_w1T = l; l += frag_floats(d, dff);
    y.b_w2T = l; l += frag_floats(dff, d);
    y.bl0 = o; y.bl_stride = l;
    o += l * y.enc.L;
    y.f0T = o; o += F * Hd;
    y.f1T = o; o += Hd * Hd;
    y.total = o;
    return y;
}

// Gradients of one workgroup of the encoder backward, in the FOLDED parameterisation (W' = W diag(gamma),
// b' = b + W beta), plain row-major [out][in]:
//   emb_w [d][V] emb_b [d], then per layer wqkv [3d][d] bqkv [3d] wo [d][d] bo [d] w1 [dff][d] b1 [dff]
//   w2 [d][dff] b2 [d] ong [d] onb [d].
struct GradLayout {
    size_t emb_w, emb_b, layer0, layer_stride;
    size_t l_wqkv, l_bqkv, l_wo, l_bo, l_w1, l_b1, l_w2, l_b2, l_ong, l_onb;
    size_t total;
};

static inline __host__ __device__ GradLayout grad_layout(const dnag_cate_dims &c) {
    GradLayout y;
    const size_t d = c.d_model, V = c.n_vocab, dff = c.d_ff;
    size_t o = 0;
    y.emb_w = o; o += align4(d * V);
    y.emb_b = o; o += d;
    y.layer0 = o;
    size_t l = 0;
    y.l_wqkv = l; l += 3 * d * d;
    y.l_bqkv = l; l += 3 * d;
    y.l_wo = l; l += d * d;
    y.l_bo = l; l += d;
    y.l_w1 = l; l += dff * d;
    y.l_b1 = l; l += dff;
    y.l_w2 = l; l += d * dff;
    y.l_b2 = l; l += d;
    y.l_ong = l; l += d;
    y.l_onb = l; l += d;
    y.layer_stride = l;
    o += l * c.n_layers;
    y.total = o;
    return y;
}

// ---------------------------------------------------------------------------------------------
// On-device packing: state_dict order -> TrainLayout (what cate_pack_impl does on the host, plus the
// backward operands).  grid (blocks, L): one launch covers every encoder layer.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ size_t frag_at(size_t K, size_t row, size_t k) {
    const size_t KH = K / 2, h = k / KH, kk = k - h * KH;
    return (((row >> 5) * (K / 8) + (kk >> 2)) * 64 + 32 * h + (row & 31)) * 4 + (kk & 3);
}

// half index of value (row, k) inside a frag16_pack matrix (dnag_common.h): hi at the returned index, lo 64 * 8 halves
// further on (the next 16-byte chunk row of the same tile and step)
__device__ __forceinline__ size_t frag16_at(size_t K, size_t row, size_t k) {
    const size_t KH = K / 2, h = k / KH, kk = k - h * KH;
    return (((row >> 5) * (K / 8) + 2 * (kk >> 3)) * 64 + 32 * h + (row & 31)) * 8 + (kk & 7);
}
__device__ __forceinline__ void frag16_put(float *base, size_t K, size_t row, size_t k, float w, float sw) {
    _Float16 *o = reinterpret_cast<_Float16 *>(base);
    const float v = w * sw;
    const _Float16 hi = (_Float16)v;
    const size_t at = frag16_at(K, row, k);
    o[at] = hi;
    o[at + 64 * 8] = (_Float16)(v - (float)hi);
}

// Power-of-two operand scales of one encoder layer (dnag_common.h "Operand scales", what cate_pack_impl derives on the
// host), from the current parameters: grid (L), 256 threads; runs before train_pack_layers_kernel, which reads them.
// `inv_keep` = 1 / (1 - p) of the training forward's dropout (1 without): the attention probabilities and the FFN hidden
// are multiplied by it before they are split (transformer.py:19, :54), so the two activation bounds grow by that factor.
__global__ __launch_bounds__(256) void train_scales_kernel(const float *__restrict__ S, const CateStateLayout sl,
                                                           float *__restrict__ P, const TrainLayout tl, const float inv_keep) {
    const int l = blockIdx.x, t = threadIdx.x;
    const int d = tl.enc.d, dff = tl.enc.dff;
    const float *SL = S + sl.layer0 + (size_t)l * sl.layer_stride;
    float *SC = P + tl.enc.layer0 + (size_t)l * tl.enc.layer_stride + tl.enc.l_sc;
    __shared__ unsigned int m[6];     // amax qkv', o, w1', w2; bound of V rows, bound of the FFN hidden (float bits)
    if (t < 6) m[t] = 0u;
    __syncthreads();
    auto note = [&](int i, float v) { atomicMax(&m[i], __float_as_uint(v)); };
    // folded row r of W (W diag(gamma)): largest magnitude, and the worst case of |W' u + b'| over LayerNorm outputs u
    auto row_stats = [&](const float *W, const float *b, const float *gam, const float *bet, float &amax, float &bound) {
        float a = 0.f, q = 0.f, bb = *b;
        // 16-byte loads, four in flight (d is a multiple of 16; the rows, gamma and beta are 16-byte aligned); the sums keep
        // their element order
#pragma unroll 4
        for (int k = 0; k < d; k += 4) {
            const v4f w4 = *reinterpret_cast<const v4f *>(W + k), g4 = *reinterpret_cast<const v4f *>(gam + k);
            const v4f b4 = *reinterpret_cast<const v4f *>(bet + k);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float w = w4[e] * g4[e];
                a = fmaxf(a, fabsf(w));
                q = fmaf(w, w, q);
                bb = fmaf(w4[e], b4[e], bb);
            }
        }
        amax = a;
        bound = (sqrtf((float)d) * sqrtf(q) + fabsf(bb)) * 1.0001f;
    };
    if (t < 3 * d) {
        const int which = t / d, r = t - which * d;
        const size_t srcw = which == 0 ? sl.s_wq : which == 1 ? sl.s_wk : sl.s_wv;
        const size_t srcb = which == 0 ? sl.s_bq : which == 1 ? sl.s_bk : sl.s_bv;
        float a, bd;
        row_stats(SL + srcw + (size_t)r * d, SL + srcb + r, SL + sl.s_ln1g, SL + sl.s_ln1b, a, bd);
        note(0, a);
        if (which == 2) note(4, bd);
    }
    if (t < dff) {
        float a, bd;
        row_stats(SL + sl.s_w1 + (size_t)t * d, SL + sl.s_b1 + t, SL + sl.s_ln2g, SL + sl.s_ln2b, a, bd);
        note(2, a);
        note(5, bd);
    }
    if (t < d) {
        float a = 0.f, a2 = 0.f;
#pragma unroll 4
        for (int k = 0; k < d; k += 4) {
            const v4f w4 = *reinterpret_cast<const v4f *>(SL + sl.s_wo + (size_t)t * d + k);
            a = fmaxf(fmaxf(a, fmaxf(fabsf(w4.x), fabsf(w4.y))), fmaxf(fabsf(w4.z), fabsf(w4.w)));
        }
#pragma unroll 4
        for (int k = 0; k < dff; k += 4) {
            const v4f w4 = *reinterpret_cast<const v4f *>(SL + sl.s_w2 + (size_t)t * dff + k);
            a2 = fmaxf(fmaxf(a2, fmaxf(fabsf(w4.x), fabsf(w4.y))), fmaxf(fabsf(w4.z), fabsf(w4.w)));
        }
        note(1, a);
        note(3, a2);
    }
    __syncthreads();
    if (t == 0) {
        const float sw_qkv = weight_scale(__uint_as_float(m[0])), sw_o = weight_scale(__uint_as_float(m[1]));
        const float sw_1 = weight_scale(__uint_as_float(m[2])), sw_2 = weight_scale(__uint_as_float(m[3]));
        const float sa_o = act_scale(__uint_as_float(m[4]) * inv_keep), sa_f2 = act_scale(__uint_as_float(m[5]) * inv_keep);
        SC[ENC_SC_INV_QKV] = 1.0f / (16.0f * sw_qkv);
        SC[ENC_SC_SA_O] = sa_o;
        SC[ENC_SC_INV_O] = 1.0f / (sa_o * sw_o);
        SC[ENC_SC_INV_F1] = 1.0f / (16.0f * sw_1);
        SC[ENC_SC_SA_F2] = sa_f2;
        SC[ENC_SC_INV_F2] = 1.0f / (sa_f2 * sw_2);
    }
}

__global__ __launch_bounds__(256) void train_pack_layers_kernel(const float *__restrict__ S, const CateStateLayout sl,
                                                                float *__restrict__ P, const TrainLayout tl) {
    const int l = blockIdx.y;
    const size_t d = tl.enc.d, dff = tl.enc.dff;
    const float *SL = S + sl.layer0 + (size_t)l * sl.layer_stride;
    float *LF = P + tl.enc.layer0 + (size_t)l * tl.enc.layer_stride;
    // weight scales of this layer, back from train_scales_kernel's table (all exact powers of two)
    const float *SC = LF + tl.enc.l_sc;
    const float sw_qkv = 1.0f / (16.0f * SC[ENC_SC_INV_QKV]), sw_o = 1.0f / (SC[ENC_SC_SA_O] * SC[ENC_SC_INV_O]);
    const float sw_1 = 1.0f / (16.0f * SC[ENC_SC_INV_F1]), sw_2 = 1.0f / (SC[ENC_SC_SA_F2] * SC[ENC_SC_INV_F2]);
    float *LB = P + tl.bl0 + (size_t)l * tl.bl_stride;
    float *L16 = LF + tl.enc.delta16;      // fp16 fragment copy of the forward matrices (split-arithmetic forward)
    const size_t n_qkv = 3 * d * d, n_o = d * d, n_1 = dff * d, n_2 = d * dff;
    const size_t n_bias = 3 * d + d + dff + d + 2 * d;
    const size_t total = n_qkv + n_o + n_1 + n_2 + n_bias;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        size_t j = i;
        if (j < n_qkv) {            // Wq|Wk|Wv [d][d] each, folded with self_attn.norm gamma
            const size_t which = j / (d * d), rk = j - which * d * d, r = rk / d, k = rk - r * d;
            const size_t srcw = which == 0 ? sl.s_wq : which == 1 ? sl.s_wk : sl.s_wv;
            const float v = SL[srcw + rk] * SL[sl.s_ln1g + k];
            LF[tl.enc.l_wqkv + frag_at(d, which * d + r, k)] = v;
            frag16_put(L16 + tl.enc.l_wqkv, d, which * d + r, k, v, sw_qkv);
            LB[tl.b_wqkvT + frag_at(3 * d, k, which * d + r)] = v;
            continue;
        }
        j -= n_qkv;
        if (j < n_o) {
            const size_t r = j / d, k = j - r * d;
            const float v = SL[sl.s_wo + j];
            LF[tl.enc.l_wo + frag_at(d, r, k)] = v;
            frag16_put(L16 + tl.enc.l_wo, d, r, k, v, sw_o);
            LB[tl.b_woT + frag_at(d, k, r)] = v;
            continue;
        }
        j -= n_o;
        if (j < n_1) {              // W1 [dff][d], folded with feed_forward.norm gamma
            const size_t r = j / d, k = j - r * d;
            const float v = SL[sl.s_w1 + j] * SL[sl.s_ln2g + k];
            LF[tl.enc.l_w1 + frag_at(d, r, k)] = v;
            frag16_put(L16 + tl.enc.l_w1, d, r, k, v, sw_1);
            LB[tl.b_w1T + frag_at(dff, k, r)] = v;
            continue;
        }
        j -= n_1;
        if (j < n_2) {              // W2 [d][dff]
            const size_t r = j / dff, k = j - r * dff;
            const float v = SL[sl.s_w2 + j];
            LF[tl.enc.l_w2 + frag_at(dff, r, k)] = v;
            frag16_put(L16 + tl.enc.l_w2, dff, r, k, v, sw_2);
            LB[tl.b_w2T + frag_at(d, k, r)] = v;
            continue;
        }
        j -= n_2;
        if (j < 3 * d) {            // bqkv' = b + W beta  (double accumulation, as the host packer)
            const size_t which = j / d, r = j - which * d;
            const size_t srcw = which == 0 ? sl.s_wq : which == 1 ? sl.s_wk : sl.s_wv;
            const size_t srcb = which == 0 ? sl.s_bq : which == 1 ? sl.s_bk : sl.s_bv;
            double acc = SL[srcb + r];
            for (size_t k = 0; k < d; ++k) acc += (double)SL[srcw + r * d + k] * (double)SL[sl.s_ln1b + k];
            LF[tl.enc.l_bqkv + j] = (float)acc;
            LF[tl.enc.l_vec + j] = (float)acc;          // l_vec: the contiguous copy encoder8_kernel stages (bqkv' | bo | b1' | b2 | on_g | on_b)
            continue;
        }
        j -= 3 * d;
        if (j < d) { LF[tl.enc.l_bo + j] = SL[sl.s_bo + j]; LF[tl.enc.l_vec + 3 * d + j] = SL[sl.s_bo + j]; continue; }
        j -= d;
        if (j < dff) {
            double acc = SL[sl.s_b1 + j];
            for (size_t k = 0; k < d; ++k) acc += (double)SL[sl.s_w1 + j * d + k] * (double)SL[sl.s_ln2b + k];
            LF[tl.enc.l_b1 + j] = (float)acc;
            LF[tl.enc.l_vec + 4 * d + j] = (float)acc;
            continue;
        }
        j -= dff;
        if (j < d) { LF[tl.enc.l_b2 + j] = SL[sl.s_b2 + j]; LF[tl.enc.l_vec + 4 * d + dff + j] = SL[sl.s_b2 + j]; continue; }
        j -= d;
        if (j < d) { const float v = S[sl.norms0 + (size_t)l * 2 * d + j]; LF[tl.enc.l_ong + j] = v; LF[tl.enc.l_vec + 5 * d + dff + j] = v; continue; }
        j -= d;
        { const float v = S[sl.norms0 + (size_t)l * 2 * d + d + j]; LF[tl.enc.l_onb + j] = v; LF[tl.enc.l_vec + 6 * d + dff + j] = v; }
    }
}

// out[c][r] = in[r][c] for r < rows, 0 for rows <= r < rows_out.  in rows have stride ld_in, out rows ld_out.
__global__ __launch_bounds__(256) void transpose_kernel(const float *__restrict__ in, int rows, int cols, int ld_in,
                                                        float *__restrict__ out, int rows_out, int ld_out) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        tile[ty + 8 * i][tx] = (r < rows && c < cols) ? in[(size_t)r * ld_in + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;
        if (c < cols && r < rows_out) out[(size_t)c * ld_out + r] = tile[tx][ty + 8 * i];
    }
}

// ---------------------------------------------------------------------------------------------
// Small reductions over rows
// ---------------------------------------------------------------------------------------------
// out[z][i][j] = sum over the z-th chunk of rows r of fa(A[r][i]) * fb(B[r][j]);  ACTB = 1: fb = silu.
// One thread per (i, j); gridDim.z chunks of `chunk` rows.  out rows have stride J.
template <int ACTB>
__global__ __launch_bounds__(256) void tn_small_kernel(const float *__restrict__ A, int lda, const float *__restrict__ Bm, int ldb,
                                                       int rows, int chunk, int I, int J, float *__restrict__ out) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= I * J) return;
    const int i = idx / J, j = idx - i * J;
    const int r0 = blockIdx.z * chunk, r1 = (r0 + chunk < rows) ? r0 + chunk : rows;
    float acc = 0.f;
    for (int r = r0; r < r1; ++r) {
        float b = Bm[(size_t)r * ldb + j];
        if (ACTB == 1) b = silu(b);
        acc = fmaf(A[(size_t)r * lda + i], b, acc);
    }
    out[(size_t)blockIdx.z * I * J + idx] = acc;
}

// The same for I <= IMAX rows of output: one thread per column j carries all I sums, so B is read once instead of I
// times (dW3 = dout^T H2: I = V = 7 against 50 MB of H2).  Per (i, j) the rows are added in the same order as above.
template <int IMAX>
__global__ __launch_bounds__(256) void tn_few_kernel(const float *__restrict__ A, int lda, const float *__restrict__ Bm, int ldb,
                                                     int rows, int chunk, int I, int J, float *__restrict__ out) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= J) return;
    const int r0 = blockIdx.z * chunk, r1 = (r0 + chunk < rows) ? r0 + chunk : rows;
    float acc[IMAX];
#pragma unroll
    for (int i = 0; i < IMAX; ++i) acc[i] = 0.f;
#pragma unroll 8
    for (int r = r0; r < r1; ++r) {
        const float b = Bm[(size_t)r * ldb + j];
#pragma unroll
        for (int i = 0; i < IMAX; ++i) acc[i] = fmaf(A[(size_t)r * lda + (i < I ? i : 0)], b, acc[i]);
    }
#pragma unroll
    for (int i = 0; i < IMAX; ++i)
        if (i < I) out[(size_t)blockIdx.z * I * J + (size_t)i * J + j] = acc[i];
}

// out[r][v] = sum_k H[r][k] W[v][k] + b[v] for V <= 8 output columns: one wave per row, lanes stride over k with 16-byte
// loads, butterfly reduction in a fixed order (the training step's final.linears.2 forward: head_out_kernel's k-loop path
// has 64 rows per block = 128 blocks at B = 1024 and took 56 us for 50 MB).
template <int V>
__global__ __launch_bounds__(256) void rowdot_kernel(const float *__restrict__ H, const float *__restrict__ W, const float *__restrict__ b,
                                                     int K, size_t rows, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const size_t r = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = 0.f;
    const v4f *h4 = reinterpret_cast<const v4f *>(H + r * K);
    for (int k4 = lane; k4 < K / 4; k4 += 64) {
        const v4f x = h4[k4];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const v4f w = reinterpret_cast<const v4f *>(W + (size_t)v * K)[k4];
            acc[v] = fmaf(x.x, w.x, acc[v]); acc[v] = fmaf(x.y, w.y, acc[v]);
            acc[v] = fmaf(x.z, w.z, acc[v]); acc[v] = fmaf(x.w, w.w, acc[v]);
        }
    }
#pragma unroll
    for (int v = 0; v < V; ++v) {
        float a = acc[v];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
        if (lane == v) out[r * V + v] = a + b[v];
    }
}

// Up to eight contiguous runs copied by one launch (the training step's state -> kernel-layout copies were seven
// hipMemcpyAsync calls, 4-10 us apart on the stream; a copy kernel per run costs its 4.5 us dispatch floor each).
struct CopySegs { const float *src[8]; float *dst[8]; unsigned long long n[8]; int count; };
__global__ __launch_bounds__(256) void copy_segs_kernel(const CopySegs c) {
    for (int k = 0; k < c.count; ++k) {
        const float *__restrict__ s = c.src[k];
        float *__restrict__ d = c.dst[k];
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < c.n[k]; i += (size_t)gridDim.x * 256) d[i] = s[i];
    }
}

// out[z][j] = sum over the z-th chunk of rows of A[r][j]
__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ A, int lda, int rows, int chunk, int J,
                                                     float *__restrict__ out) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= J) return;
    const int r0 = blockIdx.z * chunk, r1 = (r0 + chunk < rows) ? r0 + chunk : rows;
    float acc = 0.f;
    int r = r0;
    for (; r + 8 <= r1; r += 8) {            // eight rows requested together, added in row order
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = A[(size_t)(r + k) * lda + j];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k];
    }
    for (; r < r1; ++r) acc += A[(size_t)r * lda + j];
    out[(size_t)blockIdx.z * J + j] = acc;
}

// out[i] = scale * sum_p in[p * stride + i], p ascending (fixed order: reproducible)
__global__ __launch_bounds__(256) void reduce_parts_kernel(const float *__restrict__ in, int nparts, size_t stride, size_t n,
                                                           float *__restrict__ out, float scale) {
    if (((n | stride) & 3) == 0 && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
        for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * 1024) {
            // eight slabs requested before the first is added (a thread streams one 16-byte column of up to 256 slabs: with
            // one load in flight the 614 MB of a training step's slabs moved at 2 TB/s); the additions keep their order
            v4f acc = *reinterpret_cast<const v4f *>(in + i);
            for (int p0 = 1; p0 < nparts; p0 += 8) {
                v4f v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int p = p0 + k < nparts ? p0 + k : nparts - 1;       // clamped: an unconditional load
                    v[k] = *reinterpret_cast<const v4f *>(in + (size_t)p * stride + i);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (p0 + k < nparts) { acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w; }
            }
            acc.x *= scale; acc.y *= scale; acc.z *= scale; acc.w *= scale;
            *reinterpret_cast<v4f *>(out + i) = acc;
        }
        return;
    }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float acc = 0.f;
        for (int p = 0; p < nparts; ++p) acc += in[(size_t)p * stride + i];
        out[i] = acc * scale;
    }
}

// ---------------------------------------------------------------------------------------------
// Perturbation and loss (losses.py:143-155)
// ---------------------------------------------------------------------------------------------
// xt = mean_coef[b] * x + std[b] * z          (losses.py:143-145; VPSDE mean = exp(lmc) x, VESDE mean = x)
__global__ void perturb_kernel(const float *__restrict__ x, const float *__restrict__ z, const float *__restrict__ mean_coef,
                               const float *__restrict__ std, float *__restrict__ xt, size_t nelem, int per_graph) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < nelem; e += (size_t)gridDim.x * blockDim.x) {
        const size_t b = e / per_graph;
        xt[e] = __fadd_rn(__fmul_rn(mean_coef[b], x[e]), __fmul_rn(std[b], z[e]));
    }
}

// r = score * std + z with score = -out / std (VP) or out (VE); block partial sums of r^2; and
// dout [rows_pad][V] = d loss / d out:  reduce_mean: loss = mean_b mean_nv r^2, else mean_b 0.5 sum_nv r^2.
__global__ __launch_bounds__(256) void dsm_loss_kernel(const float *__restrict__ out, const float *__restrict__ z,
                                                       const float *__restrict__ std, int score_mode, int reduce_mean, int batch,
                                                       int per_graph, size_t nelem_pad, float *__restrict__ dout,
                                                       float *__restrict__ partials) {
    __shared__ float sh[256];
    const size_t nelem = (size_t)batch * per_graph;
    const float w = reduce_mean ? 2.0f / ((float)per_graph * (float)batch) : 1.0f / (float)batch;
    float acc = 0.f;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < nelem_pad; e += (size_t)gridDim.x * 256) {
        float g = 0.f;
        if (e < nelem) {
            const float s = std[e / per_graph];
            const float score = score_mode == DNAG_SCORE_VP ? __fdiv_rn(-out[e], s) : out[e];
            const float r = __fadd_rn(__fmul_rn(score, s), z[e]);
            acc = fmaf(r, r, acc);
            const float dscore = (w * r) * s;
            g = score_mode == DNAG_SCORE_VP ? -dscore / s : dscore;
        }
        dout[e] = g;
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[blockIdx.x] = sh[0];
}

// dh2[r][c] = (sum_v dout[r][v] W3[v][c]) * elu'(z) with h2 = elu(z); rows >= rows_live get 0.
template <int V>
__global__ __launch_bounds__(256) void head_out_bwd_kernel(const float *__restrict__ h2, const float *__restrict__ dout,
                                                           const float *__restrict__ W3, int K, size_t rows_live, size_t rows_pad,
                                                           float *__restrict__ dh2, float *__restrict__ amax_out) {
    __shared__ float shmax[256];
    float vmax = 0.f;
    const int tid = threadIdx.x;
    for (int c = tid; c < K; c += 256) {
        float w[V];
#pragma unroll
        for (int v = 0; v < V; ++v) w[v] = W3[(size_t)v * K + c];
        for (size_t r = blockIdx.x; r < rows_pad; r += gridDim.x) {
            float g = 0.f;
            if (r < rows_live) {
                const float *d = dout + r * V;
                float acc = 0.f;
#pragma unroll
                for (int v = 0; v < V; ++v) acc = fmaf(d[v], w[v], acc);
                const float a = h2[r * K + c];
                g = acc * (a > 0.f ? 1.0f : a + 1.0f);
            }
            dh2[r * K + c] = g;
            vmax = fmaxf(vmax, fabsf(g));
        }
    }
    // max |dh2| (float bits) for the operand scale of the split GEMMs that consume dh2
    shmax[tid] = vmax;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if (tid < s2) shmax[tid] = fmaxf(shmax[tid], shmax[tid + s2]);
        __syncthreads();
    }
    if (tid == 0 && amax_out) atomicMax(reinterpret_cast<unsigned int *>(amax_out), __float_as_uint(shmax[0]));
}

// ---------------------------------------------------------------------------------------------
// Time-embedding MLP in training: forward with saved activations, backward (cate.py:155-157, :363-374)
// ---------------------------------------------------------------------------------------------
// One block per graph: e [d] sinusoid, pre [4d] = W1 e + b1, tau [d] = W2 silu(pre) + b2.
__global__ __launch_bounds__(256) void time_embed_train_kernel(const float *__restrict__ labels, int d, const float *__restrict__ freq,
                                                               const float *__restrict__ W1, const float *__restrict__ b1,
                                                               const float *__restrict__ W2, const float *__restrict__ b2,
                                                               float *__restrict__ out, float *__restrict__ save_e,
                                                               float *__restrict__ save_pre) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *e = sm;
    float *hid = sm + d;
    const int tid = threadIdx.x, half = d / 2;
    const float lab = labels[blockIdx.x];
    for (int j = tid; j < d; j += 256) {
        const float arg = __fmul_rn(lab, freq[j < half ? j : j - half]);
        e[j] = j < half ? sinf(arg) : cosf(arg);
        save_e[(size_t)blockIdx.x * d + j] = e[j];
    }
    __syncthreads();
    for (int i = tid; i < 4 * d; i += 256) {
        float acc = 0.f;
        const float *w = W1 + (size_t)i * d;
        for (int j = 0; j < d; ++j) acc = fmaf(w[j], e[j], acc);
        const float pre = acc + b1[i];
        save_pre[(size_t)blockIdx.x * 4 * d + i] = pre;
        hid[i] = silu(pre);
    }
    __syncthreads();
    for (int c = tid; c < d; c += 256) {
        float acc = 0.f;
        const float *w = W2 + (size_t)c * 4 * d;
        for (int i = 0; i < 4 * d; ++i) acc = fmaf(w[i], hid[i], acc);
        out[(size_t)blockIdx.x * d + c] = acc + b2[c];
    }
}

// dpre[b][i] = (sum_c W2[c][i] dtau[b][c]) * silu'(pre[b][i]); one block per graph, in place over `pre` is
// not done: dpre is a separate buffer because pre feeds the W2 gradient (tn_small_kernel<1>).
__global__ __launch_bounds__(256) void time_embed_bwd_kernel(const float *__restrict__ dtau, const float *__restrict__ pre, int d,
                                                             const float *__restrict__ W2, float *__restrict__ dpre) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x;
    for (int c = tid; c < d; c += 256) sm[c] = dtau[(size_t)blockIdx.x * d + c];
    __syncthreads();
    for (int i = tid; i < 4 * d; i += 256) {
        float acc = 0.f;
        for (int c = 0; c < d; ++c) acc = fmaf(W2[(size_t)c * 4 * d + i], sm[c], acc);
        const float x = pre[(size_t)blockIdx.x * 4 * d + i];
        const float sg = 1.0f / (1.0f + expf(-x));
        dpre[(size_t)blockIdx.x * 4 * d + i] = acc * (sg * (1.0f + x * (1.0f - sg)));
    }
}

// ---------------------------------------------------------------------------------------------
// Encoder backward with weight gradients
// ---------------------------------------------------------------------------------------------
template <int N, int G, int H, int DFF>
struct TrbGeom {
    static constexpr int R = G * N;
    static constexpr int RP = (R + 31) / 32 * 32;
    static constexpr int MT = RP / 32;
    static constexpr int HS = DNAG_D + 4;
    static constexpr int QS = 3 * DNAG_D + 4;
    static constexpr int FS = DFF + 4;
    static constexpr int NIT = (G * H * N + 255) / 256;        // attention items per thread
    static constexpr size_t lds_floats = 4 * (size_t)RP * HS + (size_t)RP * QS + 2 * (size_t)G * H * N * N + (size_t)G * N * N;
    static_assert(RP == 64 || RP == 32, "the weight-gradient tiles reduce over 32 or 64 rows");
    static_assert(RP * FS <= RP * QS, "FFN hidden aliases the q|k|v buffer");
    static_assert(2 * G * H * N * N >= 2 * 16 * DNAG_D, "norm-gradient scratch aliases the probability buffers");
    static_assert(R * 16 <= RP * QS, "staged x aliases the q|k|v buffer");
};

// acc[reg] = sum_k Y[k][m0 + acc_row(reg, h)] * X[k][n0 + r] over the KR rows k of two LDS buffers
// (strides sy, sx == 4 mod 64).  Lane half h takes rows {16 c + 8 h + i}: the two halves of a ds_read_b32
// hit disjoint bank halves.
template <int KR>
__device__ __forceinline__ f32x16 wgrad_tile(const float *Y, int sy, int m0, const float *X, int sx, int n0) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const float *yp = Y + (8 * h) * sy + m0 + r;
    const float *xp = X + (8 * h) * sx + n0 + r;
#pragma unroll
    for (int c = 0; c < KR / 16; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc = mfma32(yp[(16 * c + i) * sy], xp[(16 * c + i) * sx], acc);
    return acc;
}

// dst[(m0 + row)][n0 + col] = acc, row-major with stride ld
__device__ __forceinline__ void wgrad_store(float *dst, int ld, int m0, int n0, const f32x16 &acc) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < 16; ++i) dst[(size_t)(m0 + acc_row(i, h)) * ld + n0 + r] = acc[i];
}

// Backward of encoder_kernel<.., SAVE, DROP> for the G graphs of a workgroup.
//   xt [batch][N][V] the network input; dconcat [rows_pad][L*64]; part [gridDim.x][gl.total] receives this
//   workgroup's parameter gradients (folded parameterisation, GradLayout); dtau [batch][64] the gradient
//   wrt the time embedding of each graph.
// Per layer, last to first (u = LayerNorm without affine; Dk = dropout keep / keep_prob):
//   dH += LN''^T(dconcat_l . gamma''),  d gamma'' = sum dconcat_l . xhat,  d beta'' = sum dconcat_l
//   dY2 = Dk(dH);  dW2 = dY2^T f;  dF = Dk(dY2 W2) . [f > 0];  dW1' = dF^T u2;  dH += LNhat^T(dF W1')
//   dYo = Dk(dH);  dWo = dYo^T o;  dO = dYo Wo;  attention backward -> dq|dk|dv;
//   dWqkv' = dqkv^T u1;  dH += LNhat^T(dqkv Wqkv')
// and finally d emb: dtau = sum_n dH, dWe = 8 Dk(dH)^T x, dbe = 8 sum Dk(dH).
template <int N, int V, int G, int H, int DFF>
__global__ __launch_bounds__(256, 1) void cate_enc_bwd_kernel(const float *__restrict__ P, const TrainLayout lay, const EncSave sv,
                                                              const float *__restrict__ xt, const float *__restrict__ dconcat,
                                                              const float *__restrict__ mask, int mstride, const DropMasks dm,
                                                              float *__restrict__ part, const GradLayout gl,
                                                              float *__restrict__ dtau, int batch) {
    typedef TrbGeom<N, G, H, DFF> Ge;
    constexpr int R = Ge::R, MT = Ge::MT, RP = Ge::RP, HS = Ge::HS, QS = Ge::QS, FS = Ge::FS, NIT = Ge::NIT;
    constexpr int D = DNAG_D, DK = D / H;
    constexpr int OFF_SA = H * N, OFF_FH = H * N + D, OFF_FO = H * N + D + DFF;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *dH = smem;                    // [RP][HS] gradient wrt the residual stream
    float *tmp = dH + RP * HS;           // [RP][HS] dY2 / du2 / dYo / du1
    float *xh = tmp + RP * HS;           // [RP][HS] u2, then the attention output o
    float *ob = xh + RP * HS;            // [RP][HS] dO, then u1
    float *qb = ob + RP * HS;            // [RP][QS] q|k|v, then dq|dk|dv in place
    float *fb = qb;                      // [RP][FS] f, then dF in place (dead before q|k|v is loaded)
    float *Pm = qb + RP * QS;            // [G][H][N][N] attention probabilities after dropout
    float *Sm = Pm + G * H * N * N;      // [G][H][N][N] dS / sqrt(dk)
    float *maskb = Sm + G * H * N * N;   // [G][N][N]
    float *red = Pm;                     // [2][16][64] scratch of the output-norm parameter gradients

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g0 = blockIdx.x * G;
    const size_t row0 = (size_t)g0 * N;
    const int L = lay.enc.L, F = lay.enc.F;
    const bool drop = dm.keep != nullptr;
    const float kp = dm.keep_prob;
    const int PL = dm.per_layer;
    float *mypart = part + (size_t)blockIdx.x * gl.total;
    auto live = [&](int lr) { return lr < R && g0 + lr / N < batch; };

    for (int i = tid; i < RP * HS; i += 256) dH[i] = 0.f;
    for (int i = tid; i < G * N * N; i += 256) {
        const int g = i / (N * N), ij = i - g * N * N;
        const int gg = (g0 + g < batch) ? g0 + g : 0;
        maskb[i] = mask[(size_t)(mstride ? gg : 0) * N * N + ij];
    }
    __syncthreads();

    // row statistics of x = xsrc[row0 + r]: returns xhat (4 values of this lane), rs
    // dH[r] += LayerNorm^T(g): xh = (x - m) rs;  dx = rs (g - mean(g) - xh mean(g xh))
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
    // dst[r] = LayerNorm without affine of xsrc[row0 + r] (0 for dead rows)
    auto ln_hat = [&](const float *xsrc, float *dst) {
        const int sub = tid & 15, slot = tid >> 4;
        v4f xs[RP / 16];
#pragma unroll
        for (int p = 0; p < RP / 16; ++p) {
            const int r = slot + 16 * p;
            xs[p] = *reinterpret_cast<const v4f *>(xsrc + (row0 + (live(r) ? r : 0)) * D + 4 * sub);
        }
#pragma unroll
        for (int p = 0; p < RP / 16; ++p) {
            const int r = slot + 16 * p;
            v4f xv = live(r) ? xs[p] : v4f{0.f, 0.f, 0.f, 0.f};
            const float m = row16_sum((xv.x + xv.y) + (xv.z + xv.w)) * (1.0f / 64.0f);
            xv.x -= m; xv.y -= m; xv.z -= m; xv.w -= m;
            const float var = row16_sum((xv.x * xv.x + xv.y * xv.y) + (xv.z * xv.z + xv.w * xv.w)) * (1.0f / 64.0f);
            const float rs = 1.0f / sqrtf(var + 1e-5f);
            xv.x *= rs; xv.y *= rs; xv.z *= rs; xv.w *= rs;
            *reinterpret_cast<v4f *>(dst + r * HS + 4 * sub) = xv;
        }
    };
    auto from_tmp = [&](int r, int sub) { return *reinterpret_cast<const v4f *>(tmp + r * HS + 4 * sub); };
    auto ident = [](v4f a, int) { return a; };
    // column sums over the 64 rows of an LDS buffer -> global
    auto colsum = [&](const float *buf, int stride, int ncols, float *dst) {
        for (int c = tid; c < ncols; c += 256) {
            float s = 0.f;
#pragma unroll 8
            for (int k = 0; k < RP; ++k) s += buf[k * stride + c];
            dst[c] = s;
        }
    };
    // dst[r][c] = Dk(src[r][c]) for a 64-wide site at byte offset `off` of the layer slab (0 for dead rows)
    // (four columns per thread: the keep bytes of every quad the thread handles are read first, as words, without a
    // condition -- a dead row reads the workgroup's row 0 -- so that their round trips overlap)
    auto drop_rows = [&](const float *src, float *dst, const uint8_t *KL, int kstride, int off, float scale) {
        constexpr int NQ = RP * (D / 4) / 256;
        static_assert(RP * (D / 4) % 256 == 0 && D == 64, "whole passes of 256 quads");
        uint32_t kb[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int i = tid + 256 * q, r = i >> 4, c = (i & 15) * 4;
            kb[q] = 0x01010101u;
            if (drop) kb[q] = *reinterpret_cast<const uint32_t *>(KL + (size_t)(row0 + (live(r) ? r : 0)) * kstride + off + c);
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int i = tid + 256 * q, r = i >> 4, c = (i & 15) * 4;
            v4f v = v4f{0.f, 0.f, 0.f, 0.f};
            if (live(r)) {
                v = *reinterpret_cast<const v4f *>(src + r * HS + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (drop) v[e] = drop_apply(v[e], (uint8_t)(kb[q] >> (8 * e)), kp);
                    v[e] *= scale;
                }
            }
            *reinterpret_cast<v4f *>(dst + r * HS + c) = v;
        }
    };

    for (int l = L - 1; l >= 0; --l) {
        const float *LPf = P + lay.enc.layer0 + (size_t)l * lay.enc.layer_stride;
        const float *LPb = P + lay.bl0 + (size_t)l * lay.bl_stride;
        float *GP = mypart + gl.layer0 + (size_t)l * gl.layer_stride;
        const uint8_t *KL = drop ? dm.keep + dm.rows * D + (size_t)l * dm.rows * PL : nullptr;
        // ---- (a) layer output norm: data gradient and d gamma'', d beta'' ---------------------------------
        {
            const float *og = LPf + lay.enc.l_ong;
            const float *xsrc = sv.h + (size_t)(l + 1) * sv.rows * D;
            const int sub = tid & 15, slot = tid >> 4;
            const v4f g4 = *reinterpret_cast<const v4f *>(og + 4 * sub);
            v4f ag = v4f{0.f, 0.f, 0.f, 0.f}, ab = v4f{0.f, 0.f, 0.f, 0.f};
            v4f xs[RP / 16], ds[RP / 16];
#pragma unroll
            for (int p = 0; p < RP / 16; ++p) {
                const int r = slot + 16 * p, rr = live(r) ? r : 0;
                xs[p] = *reinterpret_cast<const v4f *>(xsrc + (row0 + rr) * D + 4 * sub);
                ds[p] = *reinterpret_cast<const v4f *>(dconcat + (row0 + rr) * F + l * D + 4 * sub);
            }
#pragma unroll
            for (int p = 0; p < RP / 16; ++p) {
                const int r = slot + 16 * p;
                const bool lv = live(r);
                v4f xv = xs[p], dc = ds[p];
                if (!lv) { xv = v4f{0.f, 0.f, 0.f, 0.f}; dc = v4f{0.f, 0.f, 0.f, 0.f}; }
                const float m = row16_sum((xv.x + xv.y) + (xv.z + xv.w)) * (1.0f / 64.0f);
                xv.x -= m; xv.y -= m; xv.z -= m; xv.w -= m;
                const float var = row16_sum((xv.x * xv.x + xv.y * xv.y) + (xv.z * xv.z + xv.w * xv.w)) * (1.0f / 64.0f);
                const float rs = 1.0f / sqrtf(var + 1e-5f);
                xv.x *= rs; xv.y *= rs; xv.z *= rs; xv.w *= rs;
                ag.x += dc.x * xv.x; ag.y += dc.y * xv.y; ag.z += dc.z * xv.z; ag.w += dc.w * xv.w;
                ab.x += dc.x; ab.y += dc.y; ab.z += dc.z; ab.w += dc.w;
                v4f gv;
                gv.x = dc.x * g4.x; gv.y = dc.y * g4.y; gv.z = dc.z * g4.z; gv.w = dc.w * g4.w;
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
            *reinterpret_cast<v4f *>(red + slot * D + 4 * sub) = ag;
            *reinterpret_cast<v4f *>(red + 16 * D + slot * D + 4 * sub) = ab;
        }
        __syncthreads();
        if (tid < 2 * D) {
            const int which = tid >> 6, c = tid & 63;
            float s = 0.f;
#pragma unroll
            for (int sl = 0; sl < 16; ++sl) s += red[which * 16 * D + sl * D + c];
            GP[(which ? gl.l_onb : gl.l_ong) + c] = s;
        }
        // ---- (b) dY2 = Dk(dH) -> tmp;  f -> fb;  u2 = LNhat(h1) -> xh --------------------------------------
        drop_rows(dH, tmp, KL, PL, OFF_FO, 1.0f);
        {
            const float *fsv = sv.f + ((size_t)l * sv.rows + row0) * DFF;
            for (int i = tid; i < RP * (DFF / 4); i += 256) {
                const int r = i / (DFF / 4), c4 = i - r * (DFF / 4);
                v4f v = v4f{0.f, 0.f, 0.f, 0.f};
                if (live(r)) v = *reinterpret_cast<const v4f *>(fsv + (size_t)r * DFF + 4 * c4);
                *reinterpret_cast<v4f *>(fb + r * FS + 4 * c4) = v;
            }
        }
        ln_hat(sv.h1 + (size_t)l * sv.rows * D, xh);
        __syncthreads();
        // ---- (c1) dW2 = dY2^T f;  db2 ------------------------------------------------------------------------
        for (int j = wave; j < (D / 32) * (DFF / 32); j += 4) {
            const int mt = j / (DFF / 32), nt = j - mt * (DFF / 32);
            wgrad_store(GP + gl.l_w2, DFF, mt * 32, nt * 32, wgrad_tile<RP>(tmp, HS, mt * 32, fb, FS, nt * 32));
        }
        colsum(tmp, HS, D, GP + gl.l_b2);
        __syncthreads();
        // ---- (c2) dF = Dk(dY2 W2) . [f > 0], in place over f ---------------------------------------------------
        {
            const float *W = LPb + lay.b_w2T;
            run_jobs<D, MT, 4>(DFF / 32, tmp, HS, [&](int nt, int, int, float (&b)[D / 2]) { load_frag<D>(W, nt, b); }, ident,
                               [&](int mt, int nt, int r, int h, const f32x16 &acc) {
                                   const int col = nt * 32 + r;
                                   uint8_t kb[16];              // read first, without a condition (a dead row: row 0's)
#pragma unroll
                                   for (int i = 0; i < 16; ++i) {
                                       const int lr = mt * 32 + acc_row(i, h);
                                       kb[i] = 1;
                                       if (drop) kb[i] = KL[(size_t)(row0 + (live(lr) ? lr : 0)) * PL + OFF_FH + col];
                                   }
#pragma unroll
                                   for (int i = 0; i < 16; ++i) {
                                       const int lr = mt * 32 + acc_row(i, h);
                                       float g = 0.f;
                                       if (live(lr) && fb[lr * FS + col] > 0.f) {
                                           g = acc[i];
                                           if (drop) g = drop_apply(g, kb[i], kp);
                                       }
                                       fb[lr * FS + col] = g;
                                   }
                               });
        }
        __syncthreads();
        // ---- (d) dW1' = dF^T u2;  db1';  du2 = dF W1' -> tmp ---------------------------------------------------
        for (int j = wave; j < (DFF / 32) * (D / 32); j += 4) {
            const int mt = j / (D / 32), nt = j - mt * (D / 32);
            wgrad_store(GP + gl.l_w1, D, mt * 32, nt * 32, wgrad_tile<RP>(fb, FS, mt * 32, xh, HS, nt * 32));
        }
        colsum(fb, FS, DFF, GP + gl.l_b1);
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
        // ---- (f) dYo = Dk(dH) -> tmp;  q|k|v -> qb ---------------------------------------------------------------
        drop_rows(dH, tmp, KL, PL, OFF_SA, 1.0f);
        {
            const float *qsv = sv.qkv + ((size_t)l * sv.rows + row0) * (3 * D);
            // (requesting these in batches, as the LayerNorm rows above, was slower here: 608 -> 636 us)
            for (int i = tid; i < RP * (3 * D / 4); i += 256) {
                const int r = i / (3 * D / 4), c4 = i - r * (3 * D / 4);
                v4f v = v4f{0.f, 0.f, 0.f, 0.f};
                if (live(r)) v = *reinterpret_cast<const v4f *>(qsv + (size_t)r * (3 * D) + 4 * c4);
                *reinterpret_cast<v4f *>(qb + r * QS + 4 * c4) = v;
            }
        }
        __syncthreads();
        // ---- (g) dO = dYo Wo -> ob --------------------------------------------------------------------------------
        {
            const float *W = LPb + lay.b_woT;
            run_jobs<D, MT, 4>(D / 32, tmp, HS, [&](int nt, int, int, float (&b)[D / 2]) { load_frag<D>(W, nt, b); }, ident,
                               [&](int mt, int nt, int r, int h, const f32x16 &acc) {
                                   const int col = nt * 32 + r;
#pragma unroll
                                   for (int i = 0; i < 16; ++i) ob[(mt * 32 + acc_row(i, h)) * HS + col] = acc[i];
                               });
        }
        __syncthreads();
        // ---- (h) attention backward, one item per (graph, head, node) ------------------------------------------------
        {
            float dq[NIT][DK], dk[NIT][DK], dv[NIT][DK];
#pragma unroll
            for (int u = 0; u < NIT; ++u) {
                const int it = tid + 256 * u;
                if (it >= G * H * N) continue;
                const int n = it % N, hd = (it / N) % H, g = it / (N * H);
                const int row = g * N + n;
                // as query n: P[n][:] as the forward computes it, a = Dk(P), o_n = a v, dS[n][:], dq_n
                float q[DK], o[DK], oo[DK];
                // keep bytes of P[n][:], read before the scores (a graph past the batch: row 0's, unused)
                const bool lv = g0 + g < batch;
                const bool ka = drop && lv;
                uint8_t kbn[N];
                if (drop) {
                    const uint8_t *src = KL + (size_t)(row0 + (lv ? row : 0)) * PL + hd * N;
                    if constexpr (N % 4 == 0) {
#pragma unroll
                        for (int j = 0; j < N; j += 4) {
                            const uint32_t w = *reinterpret_cast<const uint32_t *>(src + j);
                            kbn[j] = (uint8_t)w; kbn[j + 1] = (uint8_t)(w >> 8); kbn[j + 2] = (uint8_t)(w >> 16); kbn[j + 3] = (uint8_t)(w >> 24);
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < N; ++j) kbn[j] = src[j];
                    }
                }
                load_row<DK>(qb + row * QS + hd * DK, q);
                load_row<DK>(ob + row * HS + hd * DK, o);
                float s[N], dp[N];
                float mx = -3.0e38f;
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const float *kp_ = qb + (g * N + j) * QS + D + hd * DK;
                    const float *vp = kp_ + D;
                    float acc = 0.f, accv = 0.f;
#pragma unroll
                    for (int e = 0; e < DK; e += 4) {
                        const v4f k4 = *reinterpret_cast<const v4f *>(kp_ + e);
                        const v4f v4 = *reinterpret_cast<const v4f *>(vp + e);
                        acc = fmaf(q[e], k4.x, acc); acc = fmaf(q[e + 1], k4.y, acc);
                        acc = fmaf(q[e + 2], k4.z, acc); acc = fmaf(q[e + 3], k4.w, acc);
                        accv = fmaf(o[e], v4.x, accv); accv = fmaf(o[e + 1], v4.y, accv);
                        accv = fmaf(o[e + 2], v4.z, accv); accv = fmaf(o[e + 3], v4.w, accv);
                    }
                    acc = acc * (1.0f / sqrtf((float)DK));    // constant-folded reciprocal
                    if (maskb[(g * N + n) * N + j] == 0.f) acc = -1e9f;
                    s[j] = acc;
                    dp[j] = accv;                   // d loss / d a[n][j]
                    mx = fmaxf(mx, acc);
                }
                float den = 0.f;
#pragma unroll
                for (int j = 0; j < N; ++j) { s[j] = __expf(s[j] - mx); den += s[j]; }
                float c = 0.f;
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    s[j] = s[j] * (1.0f / den);
                    if (ka) dp[j] = drop_apply(dp[j], kbn[j], kp);     // d loss / d P[n][j]
                    c = fmaf(s[j], dp[j], c);
                }
#pragma unroll
                for (int e = 0; e < DK; ++e) { dq[u][e] = 0.f; oo[e] = 0.f; }
                float *Pr = Pm + ((g * H + hd) * N + n) * N;
                float *Sr = Sm + ((g * H + hd) * N + n) * N;
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    // masked_fill (transformer.py:16) passes no gradient to a filled score
                    const float ds = (maskb[(g * N + n) * N + j] == 0.f) ? 0.f : s[j] * (dp[j] - c) * (1.0f / sqrtf((float)DK));
                    const float a = ka ? drop_apply(s[j], kbn[j], kp) : s[j];
                    Pr[j] = a;
                    Sr[j] = ds;
                    const float *kp_ = qb + (g * N + j) * QS + D + hd * DK;
                    const float *vp = kp_ + D;
#pragma unroll
                    for (int e = 0; e < DK; e += 4) {
                        const v4f k4 = *reinterpret_cast<const v4f *>(kp_ + e);
                        const v4f v4 = *reinterpret_cast<const v4f *>(vp + e);
                        dq[u][e] = fmaf(ds, k4.x, dq[u][e]); dq[u][e + 1] = fmaf(ds, k4.y, dq[u][e + 1]);
                        dq[u][e + 2] = fmaf(ds, k4.z, dq[u][e + 2]); dq[u][e + 3] = fmaf(ds, k4.w, dq[u][e + 3]);
                        oo[e] = fmaf(a, v4.x, oo[e]); oo[e + 1] = fmaf(a, v4.y, oo[e + 1]);
                        oo[e + 2] = fmaf(a, v4.z, oo[e + 2]); oo[e + 3] = fmaf(a, v4.w, oo[e + 3]);
                    }
                }
                float *op = xh + row * HS + hd * DK;      // attention output (u2 is dead)
#pragma unroll
                for (int e = 0; e < DK; e += 4) *reinterpret_cast<v4f *>(op + e) = v4f{oo[e], oo[e + 1], oo[e + 2], oo[e + 3]};
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < NIT; ++u) {
                const int it = tid + 256 * u;
                if (it >= G * H * N) continue;
                const int n = it % N, hd = (it / N) % H, g = it / (N * H);
                // as key/value n: dk_n = sum_i dS[i][n] q_i, dv_n = sum_i a[i][n] dO_i
                const float *Sg = Sm + (g * H + hd) * N * N, *Pg = Pm + (g * H + hd) * N * N;
#pragma unroll
                for (int e = 0; e < DK; ++e) { dk[u][e] = 0.f; dv[u][e] = 0.f; }
                for (int i = 0; i < N; ++i) {
                    const float ds = Sg[i * N + n], p = Pg[i * N + n];
                    const float *qp = qb + (g * N + i) * QS + hd * DK;
                    const float *op = ob + (g * N + i) * HS + hd * DK;
#pragma unroll
                    for (int e = 0; e < DK; e += 4) {
                        const v4f q4 = *reinterpret_cast<const v4f *>(qp + e);
                        const v4f o4 = *reinterpret_cast<const v4f *>(op + e);
                        dk[u][e] = fmaf(ds, q4.x, dk[u][e]); dk[u][e + 1] = fmaf(ds, q4.y, dk[u][e + 1]);
                        dk[u][e + 2] = fmaf(ds, q4.z, dk[u][e + 2]); dk[u][e + 3] = fmaf(ds, q4.w, dk[u][e + 3]);
                        dv[u][e] = fmaf(p, o4.x, dv[u][e]); dv[u][e + 1] = fmaf(p, o4.y, dv[u][e + 1]);
                        dv[u][e + 2] = fmaf(p, o4.z, dv[u][e + 2]); dv[u][e + 3] = fmaf(p, o4.w, dv[u][e + 3]);
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < NIT; ++u) {
                const int it = tid + 256 * u;
                if (it >= G * H * N) continue;
                const int n = it % N, hd = (it / N) % H, g = it / (N * H);
                float *qp = qb + (g * N + n) * QS + hd * DK;
#pragma unroll
                for (int e = 0; e < DK; e += 4) {
                    *reinterpret_cast<v4f *>(qp + e) = v4f{dq[u][e], dq[u][e + 1], dq[u][e + 2], dq[u][e + 3]};
                    *reinterpret_cast<v4f *>(qp + D + e) = v4f{dk[u][e], dk[u][e + 1], dk[u][e + 2], dk[u][e + 3]};
                    *reinterpret_cast<v4f *>(qp + 2 * D + e) = v4f{dv[u][e], dv[u][e + 1], dv[u][e + 2], dv[u][e + 3]};
                }
            }
            ln_hat(sv.h + (size_t)l * sv.rows * D, ob);     // u1 (dO is dead after the second pass)
        }
        __syncthreads();
        // ---- (i) dWo = dYo^T o;  dbo;  dWqkv' = dqkv^T u1;  dbqkv' ------------------------------------------------------
        for (int j = wave; j < (D / 32) * (D / 32) + (3 * D / 32) * (D / 32); j += 4) {
            if (j < (D / 32) * (D / 32)) {
                const int mt = j / (D / 32), nt = j - mt * (D / 32);
                wgrad_store(GP + gl.l_wo, D, mt * 32, nt * 32, wgrad_tile<RP>(tmp, HS, mt * 32, xh, HS, nt * 32));
            } else {
                const int jj = j - (D / 32) * (D / 32);
                const int mt = jj / (D / 32), nt = jj - mt * (D / 32);
                wgrad_store(GP + gl.l_wqkv, D, mt * 32, nt * 32, wgrad_tile<RP>(qb, QS, mt * 32, ob, HS, nt * 32));
            }
        }
        colsum(tmp, HS, D, GP + gl.l_bo);
        colsum(qb, QS, 3 * D, GP + gl.l_bqkv);
        __syncthreads();
        // ---- (j) du1 = dqkv Wqkv' -> tmp ----------------------------------------------------------------------------------
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
    // ---- embedding: emb = Dk((x We^T + be) * 8) + PE + tau -----------------------------------------------------------
    for (int i = tid; i < G * D; i += 256) {
        const int g = i >> 6, c = i & 63;
        if (g0 + g < batch) {
            float s = 0.f;
#pragma unroll
            for (int n = 0; n < N; ++n) s += dH[(g * N + n) * HS + c];
            dtau[(size_t)(g0 + g) * D + c] = s;
        }
    }
    drop_rows(dH, tmp, dm.keep, D, 0, 8.0f);       // the emb slab: one row of 64 per node
    float *xs = qb;                                // [R][V]
    for (int i = tid; i < R * V; i += 256) xs[i] = live(i / V) ? xt[row0 * V + i] : 0.f;
    __syncthreads();
    for (int i = tid; i < D * V; i += 256) {
        const int c = i / V, v = i - c * V;
        float s = 0.f;
        for (int r = 0; r < R; ++r) s = fmaf(tmp[r * HS + c], xs[r * V + v], s);
        mypart[gl.emb_w + i] = s;
    }
    colsum(tmp, HS, D, mypart + gl.emb_b);
}

__device__ __forceinline__ float tree16(const float (&a)[16]) {
    return (((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]))) +
           (((a[8] + a[9]) + (a[10] + a[11])) + ((a[12] + a[13]) + (a[14] + a[15])));
}

// ---------------------------------------------------------------------------------------------
// Folded-space gradients (summed over workgroups) -> state_dict order.
//   W' = W diag(gamma), b' = b + W beta   =>   dW = dW' diag(gamma) + db' beta^T,  d gamma = sum_r dW'[r] . W[r],
//   d beta = W^T db',  db = db'.
// grid (blocks, L).  gsum: GradLayout; S: parameters (state order); out: gradients (state order).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void train_unfold_layers_kernel(const float *__restrict__ gsum, const GradLayout gl,
                                                                  const float *__restrict__ S, const CateStateLayout sl,
                                                                  float *__restrict__ out, int d, int dff) {
    const int l = blockIdx.y;
    const float *GL = gsum + gl.layer0 + (size_t)l * gl.layer_stride;
    const float *SL = S + sl.layer0 + (size_t)l * sl.layer_stride;
    float *OL = out + sl.layer0 + (size_t)l * sl.layer_stride;
    const size_t D = d, DFF = dff;
    const size_t n_qkv = 3 * D * D, n_o = D * D, n_1 = DFF * D, n_2 = D * DFF;
    const size_t n_rest = 3 * D + D + DFF + D + 4 * D + 2 * D;   // biases, ln1 g/b, ln2 g/b, out-norm g/b
    const size_t total = n_qkv + n_o + n_1 + n_2 + n_rest;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        size_t j = i;
        if (j < n_qkv) {
            const size_t which = j / (D * D), rk = j - which * D * D, r = rk / D, k = rk - r * D;
            const size_t dst = which == 0 ? sl.s_wq : which == 1 ? sl.s_wk : sl.s_wv;
            OL[dst + rk] = GL[gl.l_wqkv + j] * SL[sl.s_ln1g + k] + GL[gl.l_bqkv + which * D + r] * SL[sl.s_ln1b + k];
            continue;
        }
        j -= n_qkv;
        if (j < n_o) { OL[sl.s_wo + j] = GL[gl.l_wo + j]; continue; }
        j -= n_o;
        if (j < n_1) {
            const size_t r = j / D, k = j - r * D;
            OL[sl.s_w1 + j] = GL[gl.l_w1 + j] * SL[sl.s_ln2g + k] + GL[gl.l_b1 + r] * SL[sl.s_ln2b + k];
            continue;
        }
        j -= n_1;
        if (j < n_2) { OL[sl.s_w2 + j] = GL[gl.l_w2 + j]; continue; }
        j -= n_2;
        if (j < 3 * D) {
            const size_t which = j / D, r = j - which * D;
            OL[(which == 0 ? sl.s_bq : which == 1 ? sl.s_bk : sl.s_bv) + r] = GL[gl.l_bqkv + j];
            continue;
        }
        j -= 3 * D;
        if (j < D) { OL[sl.s_bo + j] = GL[gl.l_bo + j]; continue; }
        j -= D;
        if (j < DFF) { OL[sl.s_b1 + j] = GL[gl.l_b1 + j]; continue; }
        j -= DFF;
        if (j < D) { OL[sl.s_b2 + j] = GL[gl.l_b2 + j]; continue; }
        j -= D;
        if (j < 2 * D) {            // self_attn.norm: gamma (j < D), beta
            const size_t k = j < D ? j : j - D;
            // sixteen independent chains, 32 loads in flight: the loop is a string of L2 round trips (57 us with four)
            float acc[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) acc[u] = 0.f;
            for (size_t r0 = 0; r0 < 3 * D; r0 += 16) {
                const size_t which = r0 / D;             // D is a multiple of 16: one of q, k, v per group
                const float *wsrc = SL + (which == 0 ? sl.s_wq : which == 1 ? sl.s_wk : sl.s_wv) + (r0 - which * D) * D + k;
#pragma unroll
                for (size_t u = 0; u < 16; ++u)
                    acc[u] = fmaf(j < D ? GL[gl.l_wqkv + (r0 + u) * D + k] : GL[gl.l_bqkv + r0 + u], wsrc[u * D], acc[u]);
            }
            OL[(j < D ? sl.s_ln1g : sl.s_ln1b) + k] = tree16(acc);
            continue;
        }
        j -= 2 * D;
        if (j < 2 * D) {            // feed_forward.norm
            const size_t k = j < D ? j : j - D;
            float acc[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) acc[u] = 0.f;
            for (size_t r0 = 0; r0 < DFF; r0 += 16) {
#pragma unroll
                for (size_t u = 0; u < 16; ++u) {
                    const size_t r = r0 + u;
                    acc[u] = fmaf(j < D ? GL[gl.l_w1 + r * D + k] : GL[gl.l_b1 + r], SL[sl.s_w1 + r * D + k], acc[u]);
                }
            }
            OL[(j < D ? sl.s_ln2g : sl.s_ln2b) + k] = tree16(acc);
            continue;
        }
        j -= 2 * D;
        if (j < D) { out[sl.norms0 + (size_t)l * 2 * D + j] = GL[gl.l_ong + j]; continue; }
        j -= D;
        out[sl.norms0 + (size_t)l * 2 * D + D + j] = GL[gl.l_onb + j];
    }
}

// Philox keep masks: one Philox4x32-10 block yields eight 16-bit uniforms -> eight keep bytes
// (keep = u16 >= p * 65536; the rate is exact to 2^-16).
__global__ void dropout_fill_kernel(uint8_t *keep, size_t n, float p, uint64_t seed) {
    const uint32_t thr = (uint32_t)(p * 65536.0f + 0.5f);
    const size_t ngroups = (n + 7) / 8;
    for (size_t gi = (size_t)blockIdx.x * blockDim.x + threadIdx.x; gi < ngroups; gi += (size_t)gridDim.x * blockDim.x) {
        uint32_t r[4];
        philox4x32_10((uint32_t)gi, (uint32_t)(gi >> 32), 0xd509u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
        uint8_t k[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            k[2 * i] = (r[i] & 0xffffu) >= thr ? 1 : 0;
            k[2 * i + 1] = (r[i] >> 16) >= thr ? 1 : 0;
        }
        const size_t e0 = gi * 8;
        if (e0 + 8 <= n && (reinterpret_cast<uintptr_t>(keep + e0) & 7) == 0) {
            uint64_t v = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) v |= (uint64_t)k[i] << (8 * i);
            *reinterpret_cast<uint64_t *>(keep + e0) = v;
        } else {
            for (int i = 0; i < 8 && e0 + i < n; ++i) keep[e0 + i] = k[i];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Optimiser side of the step (include/dnag.h, dnag_optim_step): global gradient norm, clipping, Adam, EMA
// ---------------------------------------------------------------------------------------------
#define OPTIM_BLOCKS 256
// partial[b] = sum of squares of the b-th contiguous chunk, accumulated in double in a fixed order
__global__ __launch_bounds__(256) void optim_sumsq_kernel(const float *__restrict__ g, size_t n, double *__restrict__ partial) {
    __shared__ double sh[256];
    const size_t chunk = ((n + OPTIM_BLOCKS - 1) / OPTIM_BLOCKS + 1023) / 1024 * 1024;
    const size_t lo = (size_t)blockIdx.x * chunk < n ? (size_t)blockIdx.x * chunk : n, hi = lo + chunk < n ? lo + chunk : n;
    double acc = 0.0;
    if (chunk % 1024 == 0 && (reinterpret_cast<uintptr_t>(g) & 15) == 0) {
        // 16-byte loads, four passes (four independent sums) in flight; a block's range is whole passes of 1024 values
        double a4[4] = {0.0, 0.0, 0.0, 0.0};
        size_t i = lo + 4 * (size_t)threadIdx.x;
        for (; i + 3 * 1024 + 4 <= hi; i += 4 * 1024) {
            v4f v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const v4f *>(g + i + 1024 * u);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                a4[u] += ((double)v[u].x * (double)v[u].x + (double)v[u].y * (double)v[u].y) +
                         ((double)v[u].z * (double)v[u].z + (double)v[u].w * (double)v[u].w);
        }
        for (; i < hi; i += 1024)
            for (int e = 0; e < 4 && i + e < hi; ++e) a4[0] += (double)g[i + e] * (double)g[i + e];
        acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
    } else {
        for (size_t i = lo + threadIdx.x; i < hi; i += 256) acc += (double)g[i] * (double)g[i];
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if ((int)threadIdx.x < s2) sh[threadIdx.x] += sh[threadIdx.x + s2];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

__global__ __launch_bounds__(256) void optim_update_kernel(const dnag_optim_args a, const double *__restrict__ partial) {
    float coef = 1.0f;
    if ((a.flags & DNAG_OPTIM_ADAM) && (a.grad_clip >= 0.f || a.grad_norm_out)) {
        // every block sums the partials in the same order: one value everywhere, no grid barrier
        __shared__ double tot;
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int b = 0; b < OPTIM_BLOCKS; ++b) t += partial[b];
            tot = t;
        }
        __syncthreads();
        const float norm = (float)sqrt(tot);
        if (a.grad_norm_out && blockIdx.x == 0 && threadIdx.x == 0) *a.grad_norm_out = norm;
        if (a.grad_clip >= 0.f) {
            coef = __fdiv_rn(a.grad_clip, __fadd_rn(norm, 1e-6f));      // clip_grad_norm_: max_norm / (total_norm + 1e-6)
            coef = coef > 1.0f ? 1.0f : coef;                            // clamp(max=1.0); the gradient is always multiplied
        }
    }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (size_t)gridDim.x * 256) {
        float p = a.params[i];
        if (a.flags & DNAG_OPTIM_ADAM) {
            float g = mul_rn(a.grads[i], coef);
            if (a.weight_decay != 0.f) g = __fadd_rn(g, mul_rn(a.weight_decay, p));              // grad.add(param, alpha=weight_decay)
            float m = a.exp_avg[i], v = a.exp_avg_sq[i];
            // operation orders of torch's CPU kernels (the goldens' source): lerp = fma(w, end - start, start) for w < 0.5,
            // addcmul = self + (value * t1) * t2, addcdiv = self + (value * t1) / t2
            m = __fmaf_rn(a.one_minus_beta1, __fsub_rn(g, m), m);                                   // exp_avg.lerp_(grad, 1 - beta1)
            v = __fadd_rn(mul_rn(v, a.beta2), mul_rn(mul_rn(a.one_minus_beta2, g), g));    // mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
            const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), a.bc2_sqrt), a.eps);             // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
            p = __fadd_rn(p, __fdiv_rn(mul_rn(-a.step_size, m), denom));                         // param.addcdiv_(exp_avg, denom, value=-step_size)
            a.exp_avg[i] = m;
            a.exp_avg_sq[i] = v;
            a.params[i] = p;
        }
        if (a.flags & DNAG_OPTIM_EMA) {
            const float s = a.shadow[i];
            a.shadow[i] = __fsub_rn(s, mul_rn(a.ema_one_minus_decay, __fsub_rn(s, p)));          // s_param.sub_(one_minus_decay * (s_param - param))
        }
    }
}

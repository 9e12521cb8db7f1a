// Head MLP GEMMs of the score network on the fp16 matrix cores at fp32 accuracy ("split" arithmetic).
// Included by dnag_lib.hip.
//
// An fp32 value a is carried as two fp16 numbers, a = hi + lo / 2048 with hi = fp16(a) and
// lo = fp16((a - hi) * 2048): 22 significant bits, and the factor 2048 keeps lo out of the fp16 subnormals.
// A product of two such values is  hi hi' + (hi lo' + lo hi') / 2048  (+ lo lo' / 2^22, dropped: below fp32
// rounding), i.e. THREE v_mfma_f32_32x32x16_f16 instead of one chain of v_mfma_f32_32x32x2_f32 -- 16 x the
// rate per instruction, so 5.3 x the fp32 MFMA peak.  Each fp16 product is exact in the fp32 accumulator; the
// hi*hi sum and the correction sum live in separate accumulators and meet once, in the epilogue.
// Range: |a| < 65504 (the operands here are LayerNorm outputs and ELU activations; weights).
//
// Operand storage ("split-pair" rows): a row of K values takes 4K bytes like fp32; every group of 8 values is
// 16 bytes of hi followed by 16 bytes of lo, which is exactly one MFMA A/B fragment each (lane (r, h) of a
// 32x32x16 step holds k = 8h .. 8h+7).  Producers write this format directly (the encoder's output norm, the
// first GEMM's epilogue, the host packer for the weights), so the GEMM main loop moves bytes and issues MFMAs.
#pragma once
#include "dnag_common.h"
#include "dnag_head.hip.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
#define SPLIT_SCALE 2048.0f
#define SPLIT_INV (1.0f / 2048.0f)

static inline __host__ __device__ void split16(float a, _Float16 &hi, _Float16 &lo) {
    hi = (_Float16)a;
    lo = (_Float16)((a - (float)hi) * SPLIT_SCALE);
}
// float index of value k of a split-pair row, in units of _Float16: hi at split_at(k), lo at split_at(k) + 8
static inline __host__ __device__ size_t split_at(size_t k) { return (k >> 3) * 16 + (k & 7); }

// host: W [rows][K] fp32 -> split-pair rows (K % 8 == 0), same byte size
static inline void split_pack(const float *W, size_t rows, size_t K, float *out) {
    _Float16 *o = reinterpret_cast<_Float16 *>(out);
    for (size_t r = 0; r < rows; ++r)
        for (size_t k = 0; k < K; ++k) {
            _Float16 hi, lo;
            split16(W[r * K + k], hi, lo);
            o[r * 2 * K + split_at(k)] = hi;
            o[r * 2 * K + split_at(k) + 8] = lo;
        }
}

#ifdef __HIPCC__
__device__ __forceinline__ f32x16 mfma16(h8 a, h8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// split-pair rows -> fp32 (tests and the concat_out of dnag_cate_forward)
__global__ void unsplit_kernel(const float *__restrict__ in, float *__restrict__ out, size_t rows, int K) {
    const _Float16 *p = reinterpret_cast<const _Float16 *>(in);
    const size_t n = rows * (size_t)K;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const size_t r = e / K, k = e - r * K;
        const size_t at = r * 2 * (size_t)K + split_at(k);
        out[e] = (float)p[at] + (float)p[at + 8] * SPLIT_INV;
    }
}

// C = act(A W^T + bias) with A [M][K] and W [Nout][K] in split-pair rows (row pitch K floats).
// OUT_SPLIT: C is written in split-pair rows (the next GEMM's A operand), else plain fp32 [M][Nout].
// Tiles, staging pipeline, LDS image and XCD-aware block order are those of gemm_bias_act_kernel: a tile row
// is 32 values = 128 bytes either way; only the consumer of the LDS image differs.
template <int ACT, bool OUT_SPLIT, int TAG>
__global__ __launch_bounds__(256, 2) void gemm_split_kernel(const float *__restrict__ A, const float *__restrict__ W,
                                                            const float *__restrict__ bias, float *__restrict__ C,
                                                            int M, int Nout, int K) {
    __shared__ __attribute__((aligned(16))) float As[2][GEMM_BM * GEMM_LS];
    __shared__ __attribute__((aligned(16))) float Bs[2][GEMM_BN * GEMM_LS];
    static_assert(GEMM_BK == 32, "two 32x32x16 steps per staged tile");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    const int nbn = Nout / GEMM_BN;
    const int nwg = gridDim.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7;
    const int wg = ((xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    const int tm = wg / nbn, tn = wg - tm * nbn;

    const int lrow = tid / GEMM_TPR, lk = (tid % GEMM_TPR) * 4;
    const float *Ag = A + (size_t)(tm * GEMM_BM + lrow) * K + lk;
    const float *Wg = W + (size_t)(tn * GEMM_BN + lrow) * K + lk;

    f32x16 accM[2][2], accC[2][2];     // hi*hi sums; (hi*lo + lo*hi) sums, still scaled by 2048
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { accM[i][j][e] = 0.f; accC[i][j][e] = 0.f; }

    v4f ra[GEMM_NP], rb[GEMM_NP];
    const int KT = K / GEMM_BK;
    auto gload = [&](int kt) {
#pragma unroll
        for (int p = 0; p < GEMM_NP; ++p) {
            ra[p] = *reinterpret_cast<const v4f *>(Ag + (size_t)(GEMM_RPP * p) * K + kt * GEMM_BK);
            rb[p] = *reinterpret_cast<const v4f *>(Wg + (size_t)(GEMM_RPP * p) * K + kt * GEMM_BK);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int p = 0; p < GEMM_NP; ++p) {
            *reinterpret_cast<v4f *>(&As[buf][(lrow + GEMM_RPP * p) * GEMM_LS + lk]) = ra[p];
            *reinterpret_cast<v4f *>(&Bs[buf][(lrow + GEMM_RPP * p) * GEMM_LS + lk]) = rb[p];
        }
    };
    gload(0);
    lstore(0);
    if (KT > 1) gload(1);
    __syncthreads();

    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < KT) lstore(cur ^ 1);
        if (kt + 2 < KT) gload(kt + 2);
        // lane (r, h) of 32x32x16 step j reads group 2j + h of its row: hi at float 8 (2j + h), lo 4 floats on
        const float *as = &As[cur][(wm * 64 + r) * GEMM_LS + 8 * h];
        const float *bs = &Bs[cur][(wn * 64 + r) * GEMM_LS + 8 * h];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            h8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *reinterpret_cast<const h8 *>(as + i * 32 * GEMM_LS + 16 * j);
                al[i] = *reinterpret_cast<const h8 *>(as + i * 32 * GEMM_LS + 16 * j + 4);
                bh[i] = *reinterpret_cast<const h8 *>(bs + i * 32 * GEMM_LS + 16 * j);
                bl[i] = *reinterpret_cast<const h8 *>(bs + i * 32 * GEMM_LS + 16 * j + 4);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    accM[i][t] = mfma16(ah[i], bh[t], accM[i][t]);
                    accC[i][t] = mfma16(ah[i], bl[t], accC[i][t]);
                    accC[i][t] = mfma16(al[i], bh[t], accC[i][t]);
                }
        }
        __syncthreads();
    }

    // epilogue: join the two sums, bias, activation; fp32 rows or split-pair rows
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = tn * GEMM_BN + wn * 64 + j * 32 + r;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row0 = tm * GEMM_BM + wm * 64 + i * 32;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const size_t row = (size_t)(row0 + acc_row(e, h));
                float v = fmaf(accC[i][j][e], SPLIT_INV, accM[i][j][e]) + bv;
                if (ACT == ACT_ELU) v = v > 0.f ? v : __expf(v) - 1.0f;
                if constexpr (!OUT_SPLIT) {
                    C[row * Nout + col] = v;
                } else {
                    // lanes 2q, 2q+1 hold adjacent columns: the even lane stores both hi halves, the odd lane both lo
                    _Float16 hi, lo;
                    split16(v, hi, lo);
                    const uint32_t hb = (uint32_t)__builtin_bit_cast(uint16_t, hi), lb = (uint32_t)__builtin_bit_cast(uint16_t, lo);
                    const uint32_t send = (r & 1) ? hb : lb;
                    const uint32_t recv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)send, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
                    const uint32_t word = (r & 1) ? (recv | (lb << 16)) : (hb | (recv << 16));
                    const int c0 = col & ~1;                                   // even column of the pair
                    uint32_t *dst = reinterpret_cast<uint32_t *>(C + row * Nout) + ((split_at(c0) + ((r & 1) ? 8 : 0)) >> 1);
                    *dst = word;
                }
            }
        }
    }
}
#endif

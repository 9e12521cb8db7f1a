// Head MLP GEMMs of the sampling path, second generation (final.linears.0 and final.linears.1 + .2, cate.py:55-73):
// the split arithmetic and the split-pair operand rows of dnag_gemm16.hip.h, with
//   * v_mfma_f32_16x16x32_f16 instead of 32x32x16: on this chip the matrix cores are power limited, and the 16x16x32 form
//     sustains 1.12-1.13 x the rate on random operands (tests/micro/mfma_shapes.hip: 1844 against 1632 TFLOP/s);
//   * LDS-DMA staging (global_load_lds_dwordx4): no staging registers, no ds_write pass.  One wave instruction fills
//     8 rows x 128 B of an UNPADDED image; bank conflicts are avoided by an XOR swizzle instead of padding: 16-byte
//     slot p of row r sits at physical slot p ^ (r & 7), applied to the per-lane SOURCE address of the DMA (its LDS
//     destination is lane-linear by construction) and to the fragment reads.  Within a staged row of 32 k the slots are
//     planar: p = kg for the fp16 hi half of k group kg (8 values), p = 4 + kg for its lo half, so lane (row l & 15,
//     k group l >> 4) of a 16x16x32 fragment reads one slot for hi and one for lo; both are conflict free;
//   * two LDS buffers of 64 KB, the DMA of tile t+1 in flight under the 96 MFMAs per wave of tile t, one barrier per tile.
// Block tile 256 x 256 x 32, 8 waves (2 x 4) of 128 x 64 = 8 x 4 MFMA tiles, one workgroup per CU.
// Included by dnag_lib.hip.
#pragma once
#include "dnag_common.h"
#include "dnag_gemm16.hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma16x(h8 a, h8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

struct DmaGeom {
    static constexpr int BM = 256, BN = 256, BK = 32, T = 512;
    static constexpr int ROWW = 32;                                   // dwords per staged row (32 k: 16 of hi, 16 of lo)
    static constexpr size_t lds_bytes = 2 * (size_t)(BM + BN) * ROWW * sizeof(float);
};

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

// C = act(A W^T + bias); A [M][K], W [Nout][K] split-pair rows (pitch K floats); sc: the GEMM's power-of-two scales
// (dnag_common.h HEAD_SC_F0 / _F1).  OUT_SPLIT: C is written as split-pair rows scaled by sc[1] (final.linears.0).
// FUSE_V > 0: operands swapped (lane <-> batch row) and final.linears.2 (W3 [FUSE_V][Nout]) applied to the ELU output in
// registers on the same MFMAs; parts [Nout / 256][M][FUSE_V] receives this block's partial sums (dnag_gemm16.hip.h).
template <int ACT, bool OUT_SPLIT, int FUSE_V>
__global__ __launch_bounds__(512, 1) void gemm_dma_kernel(const float *__restrict__ A, const float *__restrict__ W,
                                                          const float *__restrict__ bias, float *__restrict__ C, int M, int Nout,
                                                          int K, const float *__restrict__ W3, float *__restrict__ parts,
                                                          const float *__restrict__ sc) {
    typedef DmaGeom Ge;
    constexpr int BM = Ge::BM, BN = Ge::BN, ROWW = Ge::ROWW;
    static_assert(OUT_SPLIT != (FUSE_V > 0), "either the split-pair output or the fused last layer");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As0 = smem;                          // [2][BM][ROWW]
    float *Bs0 = smem + 2 * BM * ROWW;          // [2][BN][ROWW]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int wm = wave >> 2, wn = wave & 3;

    const int nbn = Nout / BN;
    const int nwg = gridDim.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7;
    const int wg = ((xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    const int tm = wg / nbn, tn = wg - tm * nbn;
    const float out_inv = sc[0], sa_out = sc[1];

    // ---- LDS-DMA staging: this wave fills rows 32 wave .. 32 wave + 31 of the A and of the B tile, 8 rows per instruction.
    // Lane = (row in the piece, physical slot); the slot's content is planar slot p = ps ^ (row & 7), i.e. 16-byte chunk
    // 2 (p & 3) + (p >> 2) of the row's 128 bytes in global memory (split-pair order: hi, lo per k group).
    const int prow = lane >> 3, ps = lane & 7, pp = ps ^ prow;       // (32 wave + 8 q + prow) & 7 == prow
    const int chunk = 2 * (pp & 3) + (pp >> 2);
    const float *ag = A + (size_t)(tm * BM + wave * 32 + prow) * K + chunk * 4;
    const float *wgp = W + (size_t)(tn * BN + wave * 32 + prow) * K + chunk * 4;
    const size_t piece = (size_t)8 * K;          // floats between two 8-row pieces
    auto stage = [&](int kt, int buf) {
        float *ad = As0 + (buf * BM + wave * 32) * ROWW, *bd = Bs0 + (buf * BN + wave * 32) * ROWW;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            __builtin_amdgcn_global_load_lds((gptr_t)(ag + q * piece + (size_t)kt * Ge::BK), (lptr_t)(ad + q * 8 * ROWW), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(wgp + q * piece + (size_t)kt * Ge::BK), (lptr_t)(bd + q * 8 * ROWW), 16, 0, 0);
        }
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment addresses: row l15 of a 16-row tile, k group g; hi slot g ^ (row & 7), lo slot the same ^ 4
    const int slot_hi = (g ^ (l15 & 7)) * 4;
    const float *a_hi = As0 + (wm * 128 + l15) * ROWW + slot_hi, *a_lo = As0 + (wm * 128 + l15) * ROWW + (slot_hi ^ 16);
    const float *b_hi = Bs0 + (wn * 64 + l15) * ROWW + slot_hi, *b_lo = Bs0 + (wn * 64 + l15) * ROWW + (slot_hi ^ 16);

    const int KT = K / Ge::BK;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < KT) stage(kt + 1, cur ^ 1);      // the other buffer was last read before the barrier that ended tile kt - 1
        const int ao = cur * BM * ROWW, bo = cur * BN * ROWW;
        h8 bh[4], bl[4], ah[2], al[2];
        bh[0] = *reinterpret_cast<const h8 *>(b_hi + bo);
        bl[0] = *reinterpret_cast<const h8 *>(b_lo + bo);
        ah[0] = *reinterpret_cast<const h8 *>(a_hi + ao);
        al[0] = *reinterpret_cast<const h8 *>(a_lo + ao);
#pragma unroll
        for (int j = 1; j < 4; ++j) {
            bh[j] = *reinterpret_cast<const h8 *>(b_hi + bo + j * 16 * ROWW);
            bl[j] = *reinterpret_cast<const h8 *>(b_lo + bo + j * 16 * ROWW);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int s = i & 1;
            if (i + 1 < 8) {        // next row tile's fragments while this one's 12 MFMAs run
                ah[s ^ 1] = *reinterpret_cast<const h8 *>(a_hi + ao + (i + 1) * 16 * ROWW);
                al[s ^ 1] = *reinterpret_cast<const h8 *>(a_lo + ao + (i + 1) * 16 * ROWW);
            }
            // the three MFMAs of one accumulator are three others apart: no back-to-back dependency
            if constexpr (FUSE_V > 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16x(bh[j], ah[s], acc[i][j]);     // D[n][m]: lane <-> batch row
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16x(bl[j], ah[s], acc[i][j]);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16x(bh[j], al[s], acc[i][j]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16x(ah[s], bh[j], acc[i][j]);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16x(ah[s], bl[j], acc[i][j]);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16x(al[s], bh[j], acc[i][j]);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of tile kt + 1 have landed in LDS
        __syncthreads();                                      // ... and everybody's; everybody is done reading tile kt
    }

    if constexpr (FUSE_V > 0) {
        constexpr int V = FUSE_V;
        static_assert(V <= 16, "one 16-row MFMA operand of W3");
        const float sw3 = sc[2], out_inv3 = sc[3];
        float *w3s = smem;                          // [V][BN] slice of W3
        float *b1s = w3s + V * BN;                  // [BN] slice of the bias
        float *red = b1s + BN;                      // [4][BM][V] per-wave-column partial sums
        for (int i = tid; i < V * BN; i += 512) { const int v = i / BN, c = i - v * BN; w3s[i] = W3[(size_t)v * Nout + tn * BN + c]; }
        for (int i = tid; i < BN; i += 512) b1s[i] = bias ? bias[tn * BN + i] : 0.f;
        __syncthreads();
        // final.linears.2 on the matrix cores: P[v][row] = sum_col W3[v][col] x[col][row].  A lane holds, for its batch row,
        // columns 16 j + 4 g + e of column tile j; one 16x16x32 MFMA takes the column tiles (2 pr, 2 pr + 1): k slot 8 g +
        // 4 t + e of lane group g <-> column 16 (2 pr + t) + 4 g + e, for the W3 fragment (row v = l15) and for x alike.
        h8 w3h[2], w3l[2];
#pragma unroll
        for (int pr = 0; pr < 2; ++pr)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                v4f w4 = {0.f, 0.f, 0.f, 0.f};
                if (l15 < V) w4 = *reinterpret_cast<const v4f *>(w3s + l15 * BN + wn * 64 + (2 * pr + t) * 16 + 4 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    _Float16 hi, lo;
                    split16(w4[e], sw3, hi, lo);
                    w3h[pr][4 * t + e] = hi; w3l[pr][4 * t + e] = lo;
                }
            }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            f32x4 pa = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                h8 xh, xl;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int j = 2 * pr + t;
                    const v4f b4 = *reinterpret_cast<const v4f *>(b1s + wn * 64 + j * 16 + 4 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = fmaf(acc[i][j][e], out_inv, b4[e]);
                        if (ACT == ACT_ELU) x = x > 0.f ? x : __expf(x) - 1.0f;
                        _Float16 hi, lo;
                        split16(x, sa_out, hi, lo);
                        xh[4 * t + e] = hi; xl[4 * t + e] = lo;
                    }
                }
                pa = mfma16x(w3h[pr], xh, pa);
                pa = mfma16x(w3l[pr], xh, pa);
                pa = mfma16x(w3h[pr], xl, pa);
            }
            // lane (batch row l15 of row tile i, group g) holds P[v = 4 g + e][row]
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * g + e < V) red[(wn * BM + wm * 128 + i * 16 + l15) * V + 4 * g + e] = pa[e] * out_inv3;
        }
        __syncthreads();
        float *dst = parts + ((size_t)tn * M + (size_t)tm * BM) * V;
        for (int i = tid; i < BM * V; i += 512)
            dst[i] = (red[i] + red[BM * V + i]) + (red[2 * BM * V + i] + red[3 * BM * V + i]);
        return;
    } else {
        // epilogue: scale back, bias, activation, split-pair rows.  Lane (column l15 of column tile j, group g) holds rows
        // 4 g + e of row tile i; lanes 2q, 2q + 1 hold adjacent columns: the even lane stores both hi halves, the odd both lo.
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = tn * BN + wn * 64 + j * 16 + l15;
            const float bv = bias ? bias[col] : 0.f;
            const int c0 = col & ~1;
            const size_t dw = (split_at(c0) + ((l15 & 1) ? 8 : 0)) >> 1;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const size_t row = (size_t)(tm * BM + wm * 128 + i * 16 + 4 * g + e);
                    float v = fmaf(acc[i][j][e], out_inv, bv);
                    if (ACT == ACT_ELU) v = v > 0.f ? v : __expf(v) - 1.0f;
                    _Float16 hi, lo;
                    split16(v, sa_out, hi, lo);
                    const uint32_t hb = (uint32_t)__builtin_bit_cast(uint16_t, hi), lb = (uint32_t)__builtin_bit_cast(uint16_t, lo);
                    const uint32_t send = (l15 & 1) ? hb : lb;
                    const uint32_t recv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)send, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
                    const uint32_t word = (l15 & 1) ? (recv | (lb << 16)) : (hb | (recv << 16));
                    reinterpret_cast<uint32_t *>(C + row * Nout)[dw] = word;
                }
            }
        }
    }
}

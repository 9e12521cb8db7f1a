// Head MLP GEMMs of the score network on the fp16 matrix cores at fp32 accuracy ("split" arithmetic).
// Included by dnag_lib.hip.
//
// An fp32 value a is carried as two fp16 numbers, s a = hi + lo with hi = fp16(s a) and lo = fp16(s a - hi):
// 22 significant bits.  The power-of-two scale s (16 for activations, 256 for weights, smaller when the packer's
// worst-case bound of the operand asks for it: dnag_common.h "Operand scales") lifts lo out of the fp16
// subnormals for the magnitudes that matter (below |a| = 2^-6 resp. 2^-10 the absolute error is 2^-29 resp. 2^-33:
// nothing against the O(1) terms of the same dot product) and is divided out, exactly, in the epilogue.
// A product of two such values is  hi hi' + hi lo' + lo hi'  (+ lo lo', dropped: 2^-22 relative, below fp32
// rounding), i.e. THREE v_mfma_f32_32x32x16_f16 instead of one chain of v_mfma_f32_32x32x2_f32 -- 16 x the rate
// per instruction, so 5.3 x the fp32 MFMA peak.  Every fp16 product is exact in the fp32 accumulator.
// Range: none to respect.  The scales of the sampling path come from the packed blob (CateLayout::hsc, l_sc), where
// the host packer derived them from the weights so that s |a| < 60000 for every reachable activation and weight.
//
// Operand storage ("split-pair" rows): a row of K values takes 4K bytes like fp32; every group of 8 values is
// 16 bytes of hi followed by 16 bytes of lo, which is exactly one MFMA A/B fragment each (lane (r, h) of a
// 32x32x16 step holds k = 8h .. 8h+7).  Producers write this format directly (the encoder's output norm, the
// first GEMM's epilogue, the host packer for the weights), so the GEMM main loop moves bytes and issues MFMAs.
#pragma once
#include <vector>
#include <cstring>
#include <type_traits>

#include "dnag_common.h"
#include "dnag_head.hip.h"

#define SPLIT_SA 16.0f        // default activation scale (IN_F32 GEMMs without a device-side maximum)
#define SPLIT_SW 256.0f       // default weight scale

static inline __host__ __device__ void split16(float a, float scale, _Float16 &hi, _Float16 &lo) {
    const float v = a * scale;
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}
// index of value k of a split-pair row, in units of _Float16: hi at split_at(k), lo at split_at(k) + 8
static inline __host__ __device__ size_t split_at(size_t k) { return (k >> 3) * 16 + (k & 7); }

// host: W [rows][K] fp32 -> split-pair rows scaled by sw (K % 8 == 0), same byte size
static inline void split_pack(const float *W, size_t rows, size_t K, float *out, float sw) {
    _Float16 *o = reinterpret_cast<_Float16 *>(out);
    for (size_t r = 0; r < rows; ++r)
        for (size_t k = 0; k < K; ++k) {
            _Float16 hi, lo;
            split16(W[r * K + k], sw, hi, lo);
            o[r * 2 * K + split_at(k)] = hi;
            o[r * 2 * K + split_at(k) + 8] = lo;
        }
}

// rows -> k-tile-blocked order (dnag_common.h kblock_at), in place.  K % 32 == 0.
static inline void kblock_rows(float *m, size_t rows, size_t K) {
    std::vector<float> t(m, m + rows * K);
    for (size_t r = 0; r < rows; ++r)
        for (size_t d = 0; d < K; d += 32) memcpy(m + kblock_at(r, d, rows), t.data() + r * K + d, 32 * sizeof(float));
}

#ifdef __HIPCC__
// k-tile-blocked split-pair activation rows ([rows_pad] of them) -> fp32 (tests and the concat_out of dnag_cate_forward)
__global__ void unsplit_kernel(const float *__restrict__ in, float *__restrict__ out, size_t rows, size_t rows_pad, int K,
                               const float *__restrict__ sa) {
    const _Float16 *p = reinterpret_cast<const _Float16 *>(in);
    const size_t n = rows * (size_t)K;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const size_t r = e / K, k = e - r * K;
        const size_t at = kblock_half_at(r, split_at(k), rows_pad);
        out[e] = ((float)p[at] + (float)p[at + 8]) / *sa;
    }
}

// The concat the encoders write (LN(h) * 16 per layer slice, k-tile-blocked split-pair rows) -> the reference's concat
// (cate.py:83-86): o_l = LN(h) gamma_l + beta_l, fp32 [rows][L * d].  For tests and the concat_out of dnag_cate_forward.
__global__ void unsplit_concat_kernel(const float *__restrict__ in, float *__restrict__ out, size_t rows, size_t rows_pad,
                                      const float *__restrict__ P, const CateLayout lay) {
    const _Float16 *p = reinterpret_cast<const _Float16 *>(in);
    const size_t K = (size_t)lay.F, n = rows * K;
    const float sa = P[lay.hsc + HEAD_SC_SA_CONCAT];
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const size_t r = e / K, k = e - r * K, l = k / lay.d, c = k - l * lay.d;
        const size_t at = kblock_half_at(r, split_at(k), rows_pad);
        const float *LP = P + lay.layer0 + l * lay.layer_stride;
        out[e] = ((float)p[at] + (float)p[at + 8]) / sa * LP[lay.l_ong + c] + LP[lay.l_onb + c];
    }
}

// C = act(A W^T + bias) with A [M][K] (scale SPLIT_SA) and W [Nout][K] (scale SPLIT_SW) in split-pair rows,
// row pitch K floats.  OUT_SPLIT: C is written as split-pair activation rows (the next GEMM's A operand), else
// plain fp32 [M][Nout].
// Block tile 256 x 256 x 32, 8 waves (2 x 4) of 128 x 64 each = 4 x 2 MFMA tiles with one accumulator per tile,
// one workgroup per CU (144 KB of dynamic LDS).  Measured on the 256 x 128 predecessor (tests/micro/
// gemm16_bench.hip): staging the operands L2 -> LDS, not the MFMAs, set the pace, so the tile is as large as the
// accumulators allow (768 B of L2 traffic per 1k algorithmic FLOPs, a third less), and every 32x32x16 step runs
// 24 MFMAs behind its 12 fragment reads.  A staged tile row is 32 values = 128 bytes, LDS row stride 36 dwords
// (conflict-free ds_read_b128); register-staged double buffering and XCD-aware block order as in
// gemm_bias_act_kernel.
struct SplitGeom {
    static constexpr int BM = 256, BN = 256, T = 512;
    static constexpr int RPP = T / 8;                 // tile rows staged per pass (8 threads x 16 bytes per row)
    static constexpr int NPA = BM / RPP, NPB = BN / RPP;
    static constexpr size_t lds_bytes = 2 * (size_t)(BM + BN) * GEMM_LS * sizeof(float);
};

#ifndef SPLIT_DBG
#define SPLIT_DBG 0      // timing ablations of tests/micro/gemm16_bench.hip (results become wrong): 1 no global loads,
#endif                   // 2 no LDS stores, 4 no MFMAs, 8 hi*hi MFMAs only, 16 no barriers in the loop, 64 no epilogue

// IN_F32 (training step, dnag_train.hip.h): operands are plain fp32 rows (pitches lda / ldw) and are split while
// they are staged into LDS (four VALU operations per element, hidden behind the MFMAs), with power-of-two scales
// that are either the fixed ones or derived on the device from the operand's largest magnitude (gradients are
// tiny: `amax_a` / `amax_w` point at max|x| as float bits, written by amax_kernel); C is plain fp32 with pitch ldc,
// epilogues ACT_NONE / ACT_ELU / ACT_MUL_ELUGRAD (aux), and gridDim.y > 1 splits K (slice y stores its partial
// product at C + y M ldc, to be summed in slice order).
struct SplitF32Args {
    int lda, ldw, ldc;
    const float *amax_a, *amax_w;     // device, or null: SPLIT_SA / SPLIT_SW
    const float *aux;                 // ACT_MUL_ELUGRAD: aux = elu(z), multiply by elu'(z)
    float *amax_out;                  // optional: max |C| (float bits, atomicMax) for a later GEMM's operand scale
    const float *sc;                  // !IN_F32: device table of this GEMM's scales (dnag_common.h HEAD_SC_F0 / _F1):
                                      // [0] 1 / (sa_in sw), [1] sa of the output it writes, [2] sw of W3, [3] 1 / ([1] [2])
};

// *out = max(*out, max |x[i]|) as float bits (non-negative floats order like unsigned integers); *out starts at 0
__global__ __launch_bounds__(256) void amax_kernel(const float *__restrict__ x, size_t n, float *__restrict__ out) {
    __shared__ float sh[256];
    float m = 0.f;
    if ((n & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {       // 16-byte loads, four in flight
        const v4f *x4 = reinterpret_cast<const v4f *>(x);
        const size_t n4 = n >> 2, stride = (size_t)gridDim.x * 256;
        size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
        for (; i + 3 * stride < n4; i += 4 * stride) {
            const v4f a = x4[i], b = x4[i + stride], c = x4[i + 2 * stride], d = x4[i + 3 * stride];
            m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))), fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w)))));
            m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(c.x), fabsf(c.y)), fmaxf(fabsf(c.z), fabsf(c.w))), fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fmaxf(fabsf(d.z), fabsf(d.w)))));
        }
        for (; i < n4; i += stride) {
            const v4f a = x4[i];
            m = fmaxf(m, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))));
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
    }
    sh[threadIdx.x] = m;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if ((int)threadIdx.x < s2) sh[threadIdx.x] = fmaxf(sh[threadIdx.x], sh[threadIdx.x + s2]);
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicMax(reinterpret_cast<unsigned int *>(out), __float_as_uint(sh[0]));
}

// c[i] *= elu'(z) given aux[i] = elu(z): 1 for aux > 0, else aux + 1; optionally max |c| -> amax_out.  The ELU-gradient
// GEMMs of the split kernel run as ACT_NONE + this pass: with the aux read fused into its epilogue (ACT_MUL_ELUGRAD) the
// IN_F32 kernel spills ~150-290 registers beside its 128 accumulators and the epilogue outlasts the k loop (235 us for
// 8192 x 1536 x 1536, against 120 us + 25 us this way).
__global__ __launch_bounds__(256) void mul_elugrad_kernel(float *__restrict__ c, const float *__restrict__ aux, size_t n4,
                                                          float *__restrict__ amax_out) {
    __shared__ float sh[256];
    float m = 0.f;
    v4f *c4 = reinterpret_cast<v4f *>(c);
    const v4f *a4 = reinterpret_cast<const v4f *>(aux);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        v4f v = c4[i];
        const v4f a = a4[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] *= a[e] > 0.f ? 1.0f : a[e] + 1.0f; m = fmaxf(m, fabsf(v[e])); }
        c4[i] = v;
    }
    if (amax_out == nullptr) return;
    sh[threadIdx.x] = m;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if ((int)threadIdx.x < s2) sh[threadIdx.x] = fmaxf(sh[threadIdx.x], sh[threadIdx.x + s2]);
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicMax(reinterpret_cast<unsigned int *>(amax_out), __float_as_uint(sh[0]));
}

// FUSE_V > 0 (final.linears.1 of the score network): the MFMA operands swap roles, so that a lane holds ONE row of
// the batch and its registers 16 output columns, and the epilogue applies final.linears.2 (W3 [FUSE_V][Nout],
// cate.py:55-73) to the ELU output in registers: parts[tn][row][v] = sum over this block's 256 columns of
// elu(...)[row][col] W3[v][col], again on split-arithmetic MFMAs, the 4 waves along N summed in a fixed order.  The 1536-wide
// hidden activation never reaches HBM; the exit kernel adds the Nout / 256 partial sums and the bias.  C is unused.
template <int ACT, bool OUT_SPLIT, int TAG, int FUSE_V = 0, bool IN_F32 = false>
__global__ __launch_bounds__(512, 1) void gemm_split_kernel(const float *__restrict__ A, const float *__restrict__ W,
                                                            const float *__restrict__ bias, float *__restrict__ C,
                                                            int M, int Nout, int K, const float *__restrict__ W3,
                                                            float *__restrict__ parts, const SplitF32Args fa) {
    typedef SplitGeom Ge;
    constexpr int BM = Ge::BM, BN = Ge::BN, RPP = Ge::RPP, NPA = Ge::NPA, NPB = Ge::NPB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As0 = smem;                          // [2][BM][LS]
    float *Bs0 = smem + 2 * BM * GEMM_LS;       // [2][BN][LS]
    static_assert(GEMM_BK == 32, "two 32x32x16 steps per staged tile");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;

    const int nbn = Nout / BN;
    const int nwg = gridDim.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7;
    const int wg = ((xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    const int tm = wg / nbn, tn = wg - tm * nbn;

    const int lrow = tid >> 3, lk = (tid & 7) * 4;
    const int KT = K / GEMM_BK / (IN_F32 ? (int)gridDim.y : 1);         // k tiles of this slice
    const int kbase = IN_F32 ? (int)blockIdx.y * KT * GEMM_BK : 0;
    const int lda = IN_F32 ? fa.lda : K, ldw = IN_F32 ? fa.ldw : K;
    const float *Ag = A + (size_t)(tm * BM + lrow) * lda + lk + kbase;
    const float *Wg = W + (size_t)(tn * BN + lrow) * ldw + lk + kbase;
    float sa = SPLIT_SA, sw = SPLIT_SW;
    float out_inv = 0.f, sa_out = 0.f, sw3 = 0.f, out_inv3 = 0.f;
    if constexpr (IN_F32) {
        if (fa.amax_a) sa = pow2_scale(*fa.amax_a);
        if (fa.amax_w) sw = pow2_scale(*fa.amax_w);
    } else {
        out_inv = fa.sc[0]; sa_out = fa.sc[1]; sw3 = fa.sc[2]; out_inv3 = fa.sc[3];
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    v4f ra[NPA], rb[NPB];
    auto gload = [&](int kt) {
#pragma unroll
        for (int p = 0; p < NPA; ++p) ra[p] = *reinterpret_cast<const v4f *>(Ag + (size_t)(RPP * p) * lda + kt * GEMM_BK);
#pragma unroll
        for (int p = 0; p < NPB; ++p) rb[p] = *reinterpret_cast<const v4f *>(Wg + (size_t)(RPP * p) * ldw + kt * GEMM_BK);
    };
    // IN_F32: four fp32 values -> 4 hi halves at their place in the group of 8, 4 lo halves 8 halves further on
    auto put_split = [&](float *rowp, v4f x, float scale) {
        h4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = x[e] * scale;
            hi[e] = (_Float16)v;
            lo[e] = (_Float16)(v - (float)hi[e]);
        }
        _Float16 *d = reinterpret_cast<_Float16 *>(rowp) + (lk >> 3) * 16 + (lk & 7);
        *reinterpret_cast<h4 *>(d) = hi;
        *reinterpret_cast<h4 *>(d + 8) = lo;
    };
    auto lstore = [&](int buf) {
        if constexpr (IN_F32) {
#pragma unroll
            for (int p = 0; p < NPA; ++p) put_split(As0 + (buf * BM + lrow + RPP * p) * GEMM_LS, ra[p], sa);
#pragma unroll
            for (int p = 0; p < NPB; ++p) put_split(Bs0 + (buf * BN + lrow + RPP * p) * GEMM_LS, rb[p], sw);
        } else {
#pragma unroll
            for (int p = 0; p < NPA; ++p) *reinterpret_cast<v4f *>(As0 + (buf * BM + lrow + RPP * p) * GEMM_LS + lk) = ra[p];
#pragma unroll
            for (int p = 0; p < NPB; ++p) *reinterpret_cast<v4f *>(Bs0 + (buf * BN + lrow + RPP * p) * GEMM_LS + lk) = rb[p];
        }
    };
    gload(0);
    lstore(0);
    if (KT > 1) gload(1);
    __syncthreads();

    // Fragment pipeline: the A fragments of tile row i are reloaded for the NEXT 32x32x16 step as soon as row i's six
    // MFMAs of this step are issued (18 MFMAs of slack), the B fragments alternate between two register sets, and
    // the barrier sits between the two steps of a k tile: after it the tile stored at the top of the iteration is
    // visible, so the second step prefetches across the tile boundary and no step starts behind an LDS round trip.
    const float *as_base = As0 + (wm * 128 + r) * GEMM_LS + 8 * h;
    const float *bs_base = Bs0 + (wn * 64 + r) * GEMM_LS + 8 * h;
    h8 ah[4], al[4], bh[2][2], bl[2][2];
    auto read_a = [&](int i, int buf, int j) {
        const float *p = as_base + (buf * BM + i * 32) * GEMM_LS + 16 * j;
        ah[i] = *reinterpret_cast<const h8 *>(p);
        al[i] = *reinterpret_cast<const h8 *>(p + 4);
    };
    auto read_b = [&](auto set, int buf, int j) {
        constexpr int S = decltype(set)::value;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float *p = bs_base + (buf * BN + t * 32) * GEMM_LS + 16 * j;
            bh[S][t] = *reinterpret_cast<const h8 *>(p);
            bl[S][t] = *reinterpret_cast<const h8 *>(p + 4);
        }
    };
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
    // one 32x32x16 step on B set `cs`, prefetching step (nbuf, nj) into the A registers and B set `ns`
    auto mfma_step = [&](auto cs, auto ns, bool more, int nbuf, int nj) {
        constexpr int CS = decltype(cs)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (SPLIT_DBG & 4) {
                acc[i][0][0] += (float)ah[i][0] + (float)al[i][0] + (float)bh[CS][0][0] + (float)bl[CS][0][0] + (float)bh[CS][1][0] + (float)bl[CS][1][0];
            } else {
                // the three MFMAs of one accumulator are interleaved with the other tile's: no back-to-back dependency
                if constexpr (FUSE_V > 0) {     // D[n][m]: lane <-> batch row, registers <-> output columns
                    acc[i][0] = mfma16(bh[CS][0], ah[i], acc[i][0]);
                    acc[i][1] = mfma16(bh[CS][1], ah[i], acc[i][1]);
                    acc[i][0] = mfma16(bl[CS][0], ah[i], acc[i][0]);
                    acc[i][1] = mfma16(bl[CS][1], ah[i], acc[i][1]);
                    acc[i][0] = mfma16(bh[CS][0], al[i], acc[i][0]);
                    acc[i][1] = mfma16(bh[CS][1], al[i], acc[i][1]);
                } else {
                acc[i][0] = mfma16(ah[i], bh[CS][0], acc[i][0]);
                acc[i][1] = mfma16(ah[i], bh[CS][1], acc[i][1]);
                if (!(SPLIT_DBG & 8)) {
                    acc[i][0] = mfma16(ah[i], bl[CS][0], acc[i][0]);
                    acc[i][1] = mfma16(ah[i], bl[CS][1], acc[i][1]);
                    acc[i][0] = mfma16(al[i], bh[CS][0], acc[i][0]);
                    acc[i][1] = mfma16(al[i], bh[CS][1], acc[i][1]);
                }
                }
            }
            if (more && !(SPLIT_DBG & 32)) {
                if (i == 0) read_b(ns, nbuf, nj);
                read_a(i, nbuf, nj);
            }
            __builtin_amdgcn_sched_barrier(0);     // keep the prefetch where it is: the scheduler sinks it to the next wait
        }
    };
    read_b(S0{}, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) read_a(i, 0, 0);

    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (!(SPLIT_DBG & 2)) if (kt + 1 < KT) lstore(cur ^ 1);
        if (!(SPLIT_DBG & 1)) if (kt + 2 < KT) gload(kt + 2);
        mfma_step(S0{}, S1{}, true, cur, 1);
        if (!(SPLIT_DBG & 16)) __syncthreads();
        mfma_step(S1{}, S0{}, kt + 1 < KT, cur ^ 1, 0);
    }

    if constexpr (FUSE_V > 0) {
        constexpr int V = FUSE_V;
        __syncthreads();                            // every wave is done with the staged tiles: LDS is free
        float *w3s = smem;                          // [V][BN] slice of W3
        float *b1s = w3s + V * BN;                  // [BN] slice of the bias
        float *red = b1s + BN;                      // [4][BM][V] per-wave-column partial sums
        for (int i = tid; i < V * BN; i += 512) { const int v = i / BN, c = i - v * BN; w3s[i] = W3[(size_t)v * Nout + tn * BN + c]; }
        for (int i = tid; i < BN; i += 512) b1s[i] = bias ? bias[tn * BN + i] : 0.f;
        __syncthreads();
        // final.linears.2 on the matrix cores too: P[v][row] = sum_col W3[v][col] x[col][row] as 32x32x16 fp16 MFMAs in
        // the same split arithmetic (W3 scaled by sc[2], x by sc[1]).  A lane's 16 accumulator registers of one
        // tile all belong to its batch row, so registers 8s .. 8s+7 ARE its B fragment of step s; the A fragment holds
        // W3 at the columns those registers stand for (rows >= V of the 32-row operand are zero).
        h8 w3h[2][2], w3l[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                v4f wa = {0.f, 0.f, 0.f, 0.f}, wb = {0.f, 0.f, 0.f, 0.f};
                if (r < V) {
                    const float *p = w3s + r * BN + wn * 64 + t * 32 + 16 * s2 + 4 * h;
                    wa = *reinterpret_cast<const v4f *>(p);
                    wb = *reinterpret_cast<const v4f *>(p + 8);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    _Float16 hi, lo;
                    split16(wa[u], sw3, hi, lo);
                    w3h[t][s2][u] = hi; w3l[t][s2][u] = lo;
                    split16(wb[u], sw3, hi, lo);
                    w3h[t][s2][4 + u] = hi; w3l[t][s2][4 + u] = lo;
                }
            }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x16 pa;
#pragma unroll
            for (int e = 0; e < 16; ++e) pa[e] = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    h8 xh, xl;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int c0 = wn * 64 + t * 32 + 16 * s2 + 8 * q + 4 * h;     // columns of registers 8 s2 + 4q ..
                        const v4f b4 = *reinterpret_cast<const v4f *>(b1s + c0);
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            float x = fmaf(acc[i][t][8 * s2 + 4 * q + u], out_inv, b4[u]);
                            if (ACT == ACT_ELU) x = x > 0.f ? x : __expf(x) - 1.0f;
                            _Float16 hi, lo;
                            split16(x, sa_out, hi, lo);
                            xh[4 * q + u] = hi; xl[4 * q + u] = lo;
                        }
                    }
                    pa = mfma16(w3h[t][s2], xh, pa);
                    pa = mfma16(w3l[t][s2], xh, pa);
                    pa = mfma16(w3h[t][s2], xl, pa);
                }
            // lane (row r, half h) holds P[v][row] for v = acc_row(e, h) in register e
            static_assert(V <= 32, "one 32-row MFMA operand of W3");
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if ((e & 3) + 8 * (e >> 2) < V && acc_row(e, h) < V)
                    red[(wn * BM + wm * 128 + i * 32 + r) * V + acc_row(e, h)] = pa[e] * out_inv3;
        }
        __syncthreads();
        float *dst = parts + ((size_t)tn * M + (size_t)tm * BM) * V;
        for (int i = tid; i < BM * V; i += 512)
            dst[i] = (red[i] + red[BM * V + i]) + (red[2 * BM * V + i] + red[3 * BM * V + i]);
        return;
    }
    if constexpr (IN_F32) {
        const float inv = 1.0f / (sa * sw);
        float *Cs = C + (size_t)blockIdx.y * (size_t)M * fa.ldc;
        float vmax = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = tn * BN + wn * 64 + j * 32 + r;
            const float bv = bias ? bias[col] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row0 = tm * BM + wm * 128 + i * 32;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const size_t at = (size_t)(row0 + acc_row(e, h)) * fa.ldc + col;
                    float v = fmaf(acc[i][j][e], inv, bv);
                    if (ACT == ACT_ELU) v = v > 0.f ? v : __expf(v) - 1.0f;
                    if (ACT == ACT_MUL_ELUGRAD) { const float a = fa.aux[at]; v *= a > 0.f ? 1.0f : a + 1.0f; }
                    Cs[at] = v;
                    vmax = fmaxf(vmax, fabsf(v));
                }
            }
        }
        if (fa.amax_out) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
            if (lane == 0) atomicMax(reinterpret_cast<unsigned int *>(fa.amax_out), __float_as_uint(vmax));
        }
        return;
    }
    // epilogue: scale back, bias, activation; fp32 rows or split-pair rows
    if ((SPLIT_DBG & 64) && acc[0][0][0] != 123.456f) return;     // timing ablation: no epilogue
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = tn * BN + wn * 64 + j * 32 + r;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row0 = tm * BM + wm * 128 + i * 32;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const size_t row = (size_t)(row0 + acc_row(e, h));
                float v = fmaf(acc[i][j][e], out_inv, bv);
                if (ACT == ACT_ELU) v = v > 0.f ? v : __expf(v) - 1.0f;
                if constexpr (!OUT_SPLIT) {
                    C[row * Nout + col] = v;
                } else {
                    // lanes 2q, 2q+1 hold adjacent columns: the even lane stores both hi halves, the odd lane both lo
                    _Float16 hi, lo;
                    split16(v, sa_out, hi, lo);
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

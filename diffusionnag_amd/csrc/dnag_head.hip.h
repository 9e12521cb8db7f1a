// Head MLP of the CATE score network (models/cate.py:55-73 built at :125-126, ~90 % of the path's
// FLOPs) and the fused exit of a score evaluation.  Included by dnag_lib.hip.
//   gemm_bias_elu_kernel : C = elu(A W^T + b), fp32 MFMA, 128x128x32 tiles          (final.linears.0/1)
//   head_out_kernel      : out = H2 W3^T + b3 (V = 7/9 columns, HBM-bound) fused with
//                          score scaling (models/utils.py:103), guidance sum (sampling.py:502-505),
//                          Langevin norm partials (sampling.py:249-250) or the reverse-diffusion
//                          predictor update (sampling.py:208-213, sde_lib.py:119-125, :198-211).
#pragma once
#include "dnag_common.h"

#define GEMM_BM 128
#define GEMM_BN 128
#ifndef GEMM_BK
#define GEMM_BK 32
#endif
#define GEMM_LS (GEMM_BK + 4)   // LDS row stride (floats): 20 or 36 -> conflict-free ds_read_b128
#define GEMM_TPR (GEMM_BK / 4)  // threads per tile row (one 16-byte load each)
#define GEMM_RPP (256 / GEMM_TPR)   // rows per staging pass
#define GEMM_NP (GEMM_BM / GEMM_RPP)  // staging passes per operand
#define GEMM_OCC (GEMM_BK == 16 ? 3 : 2)

// Epilogues of the tile GEMM
enum { ACT_NONE = 0, ACT_ELU = 1, ACT_TANH = 2, ACT_MUL_ELUGRAD = 3 };

// C[M][nvalid] = act(A W^T + bias).  A rows have stride lda (>= K, a multiple of 4), C and aux rows stride
// ldc; M % 128 == 0, Nout % 128 == 0, K % 32 == 0 (callers zero-pad the weights; only the first nvalid
// <= Nout columns are stored).  bias may be null.
// ACT_MUL_ELUGRAD multiplies by elu'(z) given aux = elu(z): 1 for aux > 0, else aux + 1 (backward
// pass of an ELU layer).  TAG only separates instantiations in profiles.
// Split-K: with gridDim.y = S > 1, slice y reduces K range [y K/S, (y+1) K/S) and stores its partial product
// at C + y * M * ldc (callers pass ACT_NONE without bias and sum the S slices in a fixed order afterwards).
template <int ACT, int TAG>
__global__ __launch_bounds__(256, GEMM_OCC) void gemm_bias_act_kernel(const float *__restrict__ A, const float *__restrict__ W,
                                                               const float *__restrict__ bias, float *__restrict__ C,
                                                               const float *__restrict__ aux,
                                                               int M, int Nout, int K, int lda, int ldc, int nvalid) {
    __shared__ __attribute__((aligned(16))) float As[2][GEMM_BM * GEMM_LS];
    __shared__ __attribute__((aligned(16))) float Bs[2][GEMM_BN * GEMM_LS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware bijective remap: blocks that share an XCD (same blockIdx % 8) walk a contiguous
    // run of tiles, N fastest, so an A row panel is reused out of that XCD's L2.
    const int nbn = Nout / GEMM_BN;
    const int nwg = gridDim.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7;
    const int wg = ((xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    const int tm = wg / nbn, tn = wg - tm * nbn;

    const int lrow = tid / GEMM_TPR, lk = (tid % GEMM_TPR) * 4;
    const int KT = K / GEMM_BK / (int)gridDim.y;          // k tiles of this slice
    const int kbase = (int)blockIdx.y * KT * GEMM_BK;
    const float *Ag = A + (size_t)(tm * GEMM_BM + lrow) * lda + lk + kbase;
    const float *Wg = W + (size_t)(tn * GEMM_BN + lrow) * K + lk + kbase;
    C += (size_t)blockIdx.y * (size_t)M * ldc;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // Pipeline: tile kt+1 sits in registers while tile kt is consumed from LDS; it is written to the
    // other LDS buffer at the START of iteration kt (its loads had a whole iteration to land) and the
    // loads of tile kt+2 are issued right after, so neither the global latency nor the ds_write burst
    // sits between the last MFMA of a tile and the barrier.
    v4f ra[GEMM_NP], rb[GEMM_NP];
    auto gload = [&](int kt) {
#pragma unroll
        for (int p = 0; p < GEMM_NP; ++p) {
            ra[p] = *reinterpret_cast<const v4f *>(Ag + (size_t)(GEMM_RPP * p) * lda + kt * GEMM_BK);
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
        const float *as = &As[cur][(wm * 64 + r) * GEMM_LS + 4 * h];
        const float *bs = &Bs[cur][(wn * 64 + r) * GEMM_LS + 4 * h];
        v4f fa[2][2], fb[2][2];   // [ping-pong][tile]
        fa[0][0] = *reinterpret_cast<const v4f *>(as);
        fa[0][1] = *reinterpret_cast<const v4f *>(as + 32 * GEMM_LS);
        fb[0][0] = *reinterpret_cast<const v4f *>(bs);
        fb[0][1] = *reinterpret_cast<const v4f *>(bs + 32 * GEMM_LS);
#pragma unroll
        for (int kc = 0; kc < GEMM_BK / 8; ++kc) {
            const int c = kc & 1, n = c ^ 1;
            if (kc + 1 < GEMM_BK / 8) {
                fa[n][0] = *reinterpret_cast<const v4f *>(as + (kc + 1) * 8);
                fa[n][1] = *reinterpret_cast<const v4f *>(as + 32 * GEMM_LS + (kc + 1) * 8);
                fb[n][0] = *reinterpret_cast<const v4f *>(bs + (kc + 1) * 8);
                fb[n][1] = *reinterpret_cast<const v4f *>(bs + 32 * GEMM_LS + (kc + 1) * 8);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0][0] = mfma32(fa[c][0][e], fb[c][0][e], acc[0][0]);
                acc[0][1] = mfma32(fa[c][0][e], fb[c][1][e], acc[0][1]);
                acc[1][0] = mfma32(fa[c][1][e], fb[c][0][e], acc[1][0]);
                acc[1][1] = mfma32(fa[c][1][e], fb[c][1][e], acc[1][1]);
            }
        }
        __syncthreads();
    }

    // epilogue: bias + ELU, 128-byte row segments per half-wave
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = tn * GEMM_BN + wn * 64 + j * 32 + r;
        if (col >= nvalid) continue;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row0 = tm * GEMM_BM + wm * 64 + i * 32;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const size_t at = (size_t)(row0 + acc_row(e, h)) * ldc + col;
                float v = acc[i][j][e] + bv;
                if (ACT == ACT_ELU) v = v > 0.f ? v : __expf(v) - 1.0f;   // elu: exp(x) - 1 on the hardware exponential
                if (ACT == ACT_TANH) v = tanhf(v);   // pred_fc.0 of the OFA surrogate (cate.py:348)
                if (ACT == ACT_MUL_ELUGRAD) { const float a = aux[at]; v *= a > 0.f ? 1.0f : a + 1.0f; }
                C[at] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Exit of one score evaluation.
// ---------------------------------------------------------------------------------------------
enum { EXIT_FORWARD = 0, EXIT_CORRECTOR = 1, EXIT_PREDICTOR = 2 };

struct ExitArgs {
    int mode;
    int batch;
    // score scaling / guidance
    int score_mode;               // DNAG_SCORE_VP: score = -out / std; DNAG_SCORE_VE: score = out
    int pred_kind;                // DNAG_PREDICTOR_REVERSE_DIFFUSION or _EULER_MARUYAMA (EXIT_PREDICTOR)
    float std;                    // marginal std at t          (EXIT_CORRECTOR / EXIT_PREDICTOR)
    float cscale;                 // classifier_scale * label sign; cgrad may be null
    const float *cgrad;           // [batch][N][V]
    // EXIT_FORWARD
    float *out;                   // [batch][N][V] raw model output
    int out_gstride;              // floats between graphs in `out` (0: N*V, contiguous)
    // EXIT_CORRECTOR
    float *gbuf;                  // [batch][N][V] total gradient
    float *partials;              // [2][gridDim.x]: per-block sums of ||grad_b||, ||noise_b||
    // EXIT_PREDICTOR
    float *x;                     // state, updated in place
    float *x_mean;                // [batch][N][V]
    float sqrt_alpha, G, G2;      // sched columns SQRTA, G, G2 (their Euler-Maruyama meaning: include/dnag.h)
    float dt;                     // sched column DT
    // noise of this exit (corrector: only its norm is needed here)
    const float *noise;           // tape slice [batch][N][V] or null -> Philox
    uint64_t seed, stream_offset;
    uint32_t draw;
    // optional records
    float *rec_score, *rec_total, *rec_a, *rec_b;   // predictor: rec_a = x_mean, rec_b = x_pred
    // final.linears.2 already applied by the fused GEMM epilogue (dnag_gemm16.hip.h): nparts partial sums
    // [nparts][rows_pad][V], part_stride floats apart; null: compute it here from H2
    const float *parts;
    int nparts;
    size_t part_stride;
    // guidance weighting of the MobileNetV3 sampler (MobileNetV3/sampling.py:880-891): 0 none, 1 per-sample norm ratio,
    // 2 per-element |score| / |grad|
    int wr_mode;
    float wr_p;
    int prod_w;
};

// H2 [rows_pad][K]; W3 [V][K]; b3 [V].  One block = GF graphs = GF*N <= 64 rows.
template <int N, int V, int GF>
__global__ __launch_bounds__(256) void head_out_kernel(const float *__restrict__ H2, const float *__restrict__ W3,
                                                       const float *__restrict__ b3, int K, ExitArgs a) {
    constexpr int R = GF * N;
    static_assert(R <= 64, "rows per block");
    constexpr int AS = 68;
    __shared__ __attribute__((aligned(16))) float As[64 * AS];
    __shared__ __attribute__((aligned(16))) float Ws[V * 64];
    __shared__ float red[4][64][V];
    __shared__ float tot[64 * V];
    __shared__ float nz[64 * V];
    __shared__ float gn[GF], nn[GF];

    const int tid = threadIdx.x;
    const int row0 = blockIdx.x * R;
    const int r = tid & 63, kq = tid >> 6;
    float acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = 0.f;

    const int lrow = tid >> 4, lk = (tid & 15) * 4;
    for (int k0 = 0; k0 < (a.parts ? 0 : K); k0 += 64) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int rr = lrow + 16 * p;
            v4f v4 = v4f{0.f, 0.f, 0.f, 0.f};
            if (rr < R && (blockIdx.x * GF + rr / N) < a.batch) v4 = *reinterpret_cast<const v4f *>(H2 + (size_t)(row0 + rr) * K + k0 + lk);
            *reinterpret_cast<v4f *>(&As[rr * AS + lk]) = v4;
        }
        for (int i = tid; i < V * 16; i += 256) {
            const int v = i >> 4, kk = (i & 15) * 4;
            *reinterpret_cast<v4f *>(&Ws[v * 64 + kk]) = *reinterpret_cast<const v4f *>(W3 + (size_t)v * K + k0 + kk);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const v4f a4 = *reinterpret_cast<const v4f *>(&As[r * AS + kq * 16 + 4 * i]);
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const v4f w4 = *reinterpret_cast<const v4f *>(&Ws[v * 64 + kq * 16 + 4 * i]);
                acc[v] = fmaf(a4.x, w4.x, acc[v]); acc[v] = fmaf(a4.y, w4.y, acc[v]);
                acc[v] = fmaf(a4.z, w4.z, acc[v]); acc[v] = fmaf(a4.w, w4.w, acc[v]);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int v = 0; v < V; ++v) red[kq][r][v] = acc[v];
    __syncthreads();

    // fused final.linears.2: the partial sums of the Hd / 256 column blocks, added in block order.  All of them are
    // requested before the first is used (a loop of load, wait, add cost one memory round trip per block).
    constexpr int MAXP = 8;
    auto model_out = [&](int rr, int v, size_t e) -> float {
        if (a.parts) {
            float val;
            if (a.nparts <= MAXP) {
                float pv[MAXP];
#pragma unroll
                for (int p = 0; p < MAXP; ++p) pv[p] = a.parts[(size_t)(p < a.nparts ? p : 0) * a.part_stride + e];
                val = pv[0];
#pragma unroll
                for (int p = 1; p < MAXP; ++p) if (p < a.nparts) val += pv[p];
            } else {
                val = a.parts[e];
                for (int p = 1; p < a.nparts; ++p) val += a.parts[(size_t)p * a.part_stride + e];
            }
            return val + b3[v];
        }
        return ((red[0][rr][v] + red[1][rr][v]) + (red[2][rr][v] + red[3][rr][v])) + b3[v];
    };
    // per-sample weight ratio ||score_b||_p / ||grad_b||_p: scores and gradients of the block's graphs go through LDS,
    // one thread per graph sums them in element order
    __shared__ float gratio[GF];
    if (a.wr_mode == 1 && a.cgrad != nullptr && a.mode != EXIT_FORWARD) {
        for (int i = tid; i < 64 * V; i += 256) {
            const int rr = i / V, v = i - rr * V;
            const bool valid = rr < R && (blockIdx.x * GF + rr / N) < a.batch;
            float sc = 0.f, cg = 0.f;
            if (valid) {
                const size_t e = (size_t)(row0 + rr) * V + v;
                const float val = model_out(rr, v, e);
                sc = a.score_mode == DNAG_SCORE_VP ? __fdiv_rn(-val, a.std) : val;
                cg = a.cgrad[e];
            }
            tot[i] = sc;
            nz[i] = cg;
        }
        __syncthreads();
        if (tid < GF) {
            float sp = 0.f, cp = 0.f;
            for (int i = 0; i < N * V; ++i) {
                const float s1 = fabsf(tot[tid * N * V + i]), c1 = fabsf(nz[tid * N * V + i]) * (a.prod_w ? fabsf(a.cscale) : 1.0f);
                if (a.wr_p == 1.0f) { sp += s1; cp += c1; }
                else if (a.wr_p == 2.0f) { sp = fmaf(s1, s1, sp); cp = fmaf(c1, c1, cp); }
                else { sp += powf(s1, a.wr_p); cp += powf(c1, a.wr_p); }
            }
            if (a.wr_p == 2.0f) { sp = sqrtf(sp); cp = sqrtf(cp); }
            else if (a.wr_p != 1.0f) { sp = powf(sp, 1.0f / a.wr_p); cp = powf(cp, 1.0f / a.wr_p); }
            gratio[tid] = __fdiv_rn(sp, cp);
        }
        __syncthreads();
    }
    // one thread per (row, v) element of this block.  Every global read of an element is issued up front and without a
    // condition (an absent operand reads b3[0] instead; threads beyond the batch read element 0): the reads then overlap
    // instead of costing one memory round trip each behind the branches that select them.
    const bool upd = a.mode != EXIT_FORWARD && a.mode != EXIT_CORRECTOR;
    for (int i = tid; i < 64 * V; i += 256) {
        const int rr = i / V, v = i - rr * V;
        const int g = rr / N;
        const bool valid = rr < R && (blockIdx.x * GF + g) < a.batch;
        const size_t e = valid ? (size_t)(row0 + rr) * V + v : 0;
        const float val = model_out(valid ? rr : 0, v, e);
        const float cg = (a.cgrad ? a.cgrad : b3)[a.cgrad ? e : 0];
        const float nzin = (a.noise ? a.noise : b3)[a.noise ? e : 0];
        const float xv = (upd ? a.x : b3)[upd ? e : 0];
        float total = 0.f, noise = 0.f;
        if (valid) {
            if (a.mode == EXIT_FORWARD) {
                if (a.out_gstride) a.out[(size_t)(blockIdx.x * GF + g) * a.out_gstride + (rr - g * N) * V + v] = val;
                else a.out[e] = val;
            } else {
                const float score = a.score_mode == DNAG_SCORE_VP ? __fdiv_rn(-val, a.std) : val;   // models/utils.py:103 | :116
                total = score;
                if (a.cgrad != nullptr) {
                    float w = a.cscale;
                    if (a.wr_mode == 1) w = mul_rn(w, gratio[g]);                                        // MBV3 sampling.py:886
                    else if (a.wr_mode == 2) w = mul_rn(w, __fdiv_rn(fabsf(score), fabsf(cg)));        // :890-891
                    total = __fadd_rn(score, mul_rn(w, cg));                                            // sampling.py:505
                }
                if (a.rec_score) a.rec_score[e] = score;
                if (a.rec_total) a.rec_total[e] = total;
                noise = a.noise ? nzin : philox_normal(a.seed, a.stream_offset + e, a.draw);
                if (a.mode == EXIT_CORRECTOR) {
                    a.gbuf[e] = total;
                } else {
                    // f = sqrt(alpha) x - x ; rev_f = f - G^2 total ; x_mean = x - rev_f ; x = x_mean + G z
                    float xm;
                    if (a.pred_kind == DNAG_PREDICTOR_EULER_MARUYAMA) {
                        // drift = c x - diffusion^2 total ; x_mean = x + drift dt ; x = x_mean + diffusion sqrt(-dt) z
                        // (sampling.py:194-200, sde_lib.py:95-103; a VESDE has c = 0)
                        const float drift = __fsub_rn(mul_rn(a.sqrt_alpha, xv), mul_rn(a.G2, total));
                        xm = __fadd_rn(xv, mul_rn(drift, a.dt));
                    } else {
                        const float f = __fsub_rn(mul_rn(a.sqrt_alpha, xv), xv);
                        const float rev_f = __fsub_rn(f, mul_rn(a.G2, total));
                        xm = __fsub_rn(xv, rev_f);
                    }
                    const float xn = __fadd_rn(xm, mul_rn(a.G, noise));
                    a.x_mean[e] = xm;
                    a.x[e] = xn;
                    if (a.rec_a) a.rec_a[e] = xm;
                    if (a.rec_b) a.rec_b[e] = xn;
                }
            }
        }
        tot[i] = total;
        nz[i] = noise;
    }
    if (a.mode != EXIT_CORRECTOR) return;
    __syncthreads();
    // per-graph L2 norms (sampling.py:249-250), fixed summation order -> reproducible
    if (tid < 2 * GF) {
        const int g = tid % GF;
        const float *src = (tid < GF) ? tot : nz;
        float s = 0.f;
        for (int i = 0; i < N * V; ++i) { const float t = src[g * N * V + i]; s = fmaf(t, t, s); }
        const float nrm = (blockIdx.x * GF + g < a.batch) ? sqrtf(s) : 0.f;
        if (tid < GF) gn[g] = nrm; else nn[g] = nrm;
    }
    __syncthreads();
    if (tid == 0) {
        float sg = 0.f, sn = 0.f;
        for (int g = 0; g < GF; ++g) { sg += gn[g]; sn += nn[g]; }
        a.partials[blockIdx.x] = sg;
        a.partials[gridDim.x + blockIdx.x] = sn;
    }
}

// ---------------------------------------------------------------------------------------------
// Langevin corrector update (sampling.py:249-254).  Every block reduces the per-block partial sums
// in the same fixed order, so all blocks derive the same step size without a grid barrier.
// ---------------------------------------------------------------------------------------------
struct CorrArgs {
    const float *partials; int nparts;   // [nseg][2][nparts / nseg]: one segment per rank (nseg = 1: this shard alone)
    int nseg;
    int batch; size_t nelem;
    float snr, alpha;
    const float *gbuf; float *x;
    const float *noise; uint64_t seed, stream_offset; uint32_t draw;
    float *rec_x;                 // [nelem] or null
    float *rec_step;              // [3] or null
};

__device__ __forceinline__ float block_sum_256(float v, float *sh) {
    const int tid = threadIdx.x;
    sh[tid] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) sh[tid] = sh[tid] + sh[tid + s];
        __syncthreads();
    }
    const float out = sh[0];
    __syncthreads();
    return out;
}

__global__ __launch_bounds__(256) void corrector_update_kernel(CorrArgs a) {
    __shared__ float sh[256], sh2[256];
    const int tid = threadIdx.x;
    // the element operands do not depend on the step size: requested (and the noise drawn) before the reduction, whose
    // barriers then hide their latency.  Unconditional reads (an absent tape reads gbuf; threads past the end element 0).
    auto fetch = [&](size_t e, float &xv, float &gv, float &nz) {
        const size_t ec = e < a.nelem ? e : 0;
        xv = a.x[ec]; gv = a.gbuf[ec];
        const float nzin = (a.noise ? a.noise : a.gbuf)[ec];
        nz = a.noise ? nzin : philox_normal(a.seed, a.stream_offset + ec, a.draw);
    };
    size_t e = (size_t)blockIdx.x * 256 + tid;
    float xv, gv, nz;
    fetch(e, xv, gv, nz);
    float sg = 0.f, sn = 0.f;
    // block k of the (global) batch: segment k / seg, entry k % seg; the order of the sums depends on k alone, so a
    // batch cut into equal shards reduces exactly like the whole batch
    const int seg = a.nparts / a.nseg;
    for (int i = tid; i < a.nparts; i += 256) {
        const int s_ = i / seg, j = i - s_ * seg;
        sg += a.partials[(size_t)s_ * 2 * seg + j];
        sn += a.partials[(size_t)s_ * 2 * seg + seg + j];
    }
    // both sums ride on one tree (each with the pairing of block_sum_256: the same bits, half the barriers)
    sh[tid] = sg; sh2[tid] = sn;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) { sh[tid] = sh[tid] + sh[tid + s]; sh2[tid] = sh2[tid] + sh2[tid + s]; }
        __syncthreads();
    }
    const float grad_norm = sh[0] / (float)a.batch;
    const float noise_norm = sh2[0] / (float)a.batch;
    // step_size = (snr * noise_norm / grad_norm) ** 2 * 2 * alpha
    const float ratio = __fdiv_rn(mul_rn(a.snr, noise_norm), grad_norm);
    const float step = mul_rn(mul_rn(mul_rn(ratio, ratio), 2.0f), a.alpha);
    const float nscale = sqrtf(mul_rn(step, 2.0f));
    if (blockIdx.x == 0 && tid == 0 && a.rec_step) {
        a.rec_step[0] = grad_norm; a.rec_step[1] = noise_norm; a.rec_step[2] = step;
    }
    while (e < a.nelem) {
        const float xm = __fadd_rn(xv, mul_rn(step, gv));
        const float xn = __fadd_rn(xm, mul_rn(nscale, nz));
        a.x[e] = xn;
        if (a.rec_x) a.rec_x[e] = xn;
        e += (size_t)gridDim.x * 256;
        if (e < a.nelem) fetch(e, xv, gv, nz);
    }
}

// x0 = prior sample (sde_lib.py:190-191): copy of the caller's tensor, or Philox draw 0.
__global__ void prior_kernel(float *x, const float *x0, size_t nelem, uint64_t seed, uint64_t stream_offset, float scale) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < nelem; e += (size_t)gridDim.x * blockDim.x)
        x[e] = x0 ? x0[e] : mul_rn(philox_normal(seed, stream_offset + e, 0u), scale);
}

// out = (x_mean + 1) / 2  (datasets_nas.py:21-28)
__global__ void inverse_scale_kernel(const float *xm, float *out, size_t nelem) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < nelem; e += (size_t)gridDim.x * blockDim.x)
        out[e] = __fdiv_rn(__fadd_rn(xm[e], 1.0f), 2.0f);
}

// utils.py:254-255
__global__ void quantize_kernel(const float *x, float *q, size_t n) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x)
        q[e] = x[e] >= 0.5f ? 1.0f : 0.0f;
}

// ---------------------------------------------------------------------------------------------
// Time embedding MLP: tau = W2 silu(W1 [sin(l w) | cos(l w)] + b1) + b2  (cate.py:155-157, :363-374;
// digcn_meta.py:116-118, :162-173).  One block per distinct label; d <= 256.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void time_embed_kernel(const float *__restrict__ labels, int d, const float *__restrict__ freq,
                                                         const float *__restrict__ W1, const float *__restrict__ b1,
                                                         const float *__restrict__ W2, const float *__restrict__ b2,
                                                         float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *e = sm;            // [d]
    float *hid = sm + d;      // [4d]
    const int tid = threadIdx.x, half = d / 2;
    const float lab = labels[blockIdx.x];
    for (int j = tid; j < d; j += 256) {
        const float arg = mul_rn(lab, freq[j < half ? j : j - half]);
        e[j] = j < half ? sinf(arg) : cosf(arg);
    }
    __syncthreads();
    for (int i = tid; i < 4 * d; i += 256) {
        float acc = 0.f;
        const float *w = W1 + (size_t)i * d;
        for (int j = 0; j < d; ++j) acc = fmaf(w[j], e[j], acc);
        hid[i] = silu(acc + b1[i]);
    }
    __syncthreads();
    for (int c = tid; c < d; c += 256) {
        float acc = 0.f;
        const float *w = W2 + (size_t)c * 4 * d;
        for (int i = 0; i < 4 * d; ++i) acc = fmaf(w[i], hid[i], acc);
        out[(size_t)blockIdx.x * d + c] = acc + b2[c];
    }
}

// Encoder of the CATE score network, 64 rows (8 graphs of 8 nodes, NAS-Bench-201, or 3 graphs of 20, MobileNetV3 / OFA) per
// workgroup: the sampling path (MODE 0) and the training step's forward (MODE 1 / 2: saved activations, dropout).  The same
// mathematics as encoder_kernel<...> (dnag_encoder.hip.h, which stays the kernel of the attention-probability output and
// of the OFA surrogate's own encoder), organised around what that kernel's per-phase cycle stamps showed: it was bound by
// vector-ALU issue, most of it the LayerNorm and the fp16 hi/lo split of the A operand, which every 32x32 tile job
// repeated (12 QKV jobs for 2 row tiles).
//
// Here every activation that feeds a linear is normalised / scaled / split ONCE, by the phase that produces it, into an
// LDS image of split-pair rows (dnag_gemm16.hip.h: per 8 values 16 B of fp16 hi, 16 B of lo), so a tile job is two
// ds_read_b128 and three v_mfma_f32_32x32x16_f16 per 16 k:
//   P0  finish the row statistics of h (see below): concat slice of the previous layer (output norm, transformer.py:177)
//       and LN1(h) -> A16, from the wave's registers
//   P1  q|k|v = A16 Wqkv'^T + b            (3 tile jobs per wave)                    -> qkv fp32 [64][192]
//   P2  attention: one thread per (graph, head, query pair): K and V rows are read once for two queries (scalar pairs that
//       the compiler packs into v_pk_fma_f32 where it can); 20 keys: two blocks with a running maximum   -> A16 (scale sa_o)
//   P3  h += A16 Wo^T + bo                 (the wave's tile of h lives in registers)
//   P4  LN2(h) -> A16, from the wave's registers
//   P5  hidden = relu(A16 W1'^T + b1'), MFMA operands swapped so that a lane owns one batch row and writes its 16
//       columns as 4-value runs of the split image                                     -> hidden (scale sa_f2)
//   P6  h += hidden W2^T + b2
// The residual stream h lives in registers only (round 2 also kept fp32 rows in LDS and re-read them for every
// LayerNorm): one 32x32 tile per wave (P3 and P6 have exactly four jobs), computed with the MFMA operands swapped so that a
// lane owns one batch row and 16 of the tile's 32 columns.  Row statistics: mean and centred sum of squares of the
// lane's 16 values, merged with the other lane half (v_permlane32_swap) and -- through 8 bytes per row and tile in LDS,
// across the barrier that separates the phases anyway -- with the wave that holds the row's other 32 columns (Chan's
// pairwise update: no cancellation), then the wave normalises its own registers and writes its part of the split image.
// Weight fragments travel global -> registers one to two phases ahead, one 1 KB piece behind each MFMA of the accumulation
// chains; every global operation of the layer loop is unconditional (exact s_waitcnt vmcnt counts); lane ids are re-derived
// per phase (E8_IDS); the two workgroups of a CU swap a raised wave priority every layer (profiles/r03d_pmc_sq_and_stamps.md).
// Restates (reference NAS-Bench-201/): models/cate.py:145-160, models/transformer.py:12-20, :23-42, :69-80, :97-108,
// :127-136, :166-179, models/cate.py:83-86.
#pragma once
#include "dnag_common.h"
#include "dnag_encoder.hip.h"
#include <type_traits>

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

// N nodes per graph: 8 (NAS-Bench-201, 8 graphs per workgroup) or 20 (MobileNetV3 / OFA: 3 graphs = 60 of the 64 rows; the four
// spare rows stay zero, every row-wise phase carries them along and nothing reads them back)
template <int N_>
struct Enc8Geom {
    static constexpr int N = N_, G = 64 / N_, R = 64, RV = G * N, D = 64, H = 8, DK = 8, DFF = 128;
    static_assert(N % 2 == 0 && G * H * (N / 2) <= 256, "one thread per (graph, head, query pair)");
    static constexpr int AS = D + 4;            // dword row stride of A16 (== 4 mod 64: conflict-free ds_read_b128)
    static constexpr int QS = 3 * D + 4;        // q|k|v rows
    static constexpr int FS = DFF + 4;          // split hidden rows (DFF values = DFF dwords)
    static constexpr int BIG = R * FS + R * AS; // hidden, then the embedding's staging rows / the row statistics (qkv [R][QS] aliases the front)
    static_assert(R * QS + 4 * R <= BIG, "q|k|v aliases the hidden region; the row statistics [R][2][2] sit behind it");
    static constexpr int VEC = 8 + 3 * D + D + DFF + D;   // staged per layer: scales (CateLayout::l_sc, 8) | bqkv' | bo | b1' | b2
    static constexpr size_t lds_floats = (size_t)R * AS + BIG + (size_t)G * N * N + VEC;
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

// MODE 0: sampling path (concat = k-tile-blocked split-pair LayerNorm image).  MODE 1 / 2: the training step's forward
// (dnag_cate_dsm_step): the activations the backward pass needs are written to `sv` (EncSave, dnag_encoder.hip.h), the
// concat is plain fp32 rows [row][F] WITH the output norm's affine (the training head keeps the unfolded weights), and
// MODE 2 applies the dropout keep masks `dm` at the reference's five sites (cate.py:147, transformer.py:19, :54, :80, :108).
// The training modes are bounded at one workgroup per CU: the keep words read ahead of each phase's MFMA chain and the
// save addresses take 268-272 VGPRs (24 spilled registers at the 256 of two workgroups; 173 -> 152 us at B = 1024, and no
// difference at B = 8192, where the saves' HBM traffic sets the pace).
template <int N_, int V, int MODE = 0>
__global__ __launch_bounds__(256, MODE == 0 ? 2 : 1) void encoder8_kernel(const float *__restrict__ P, const CateLayout lay,
                                                          const float *__restrict__ x, const float *__restrict__ temb,
                                                          int tstride, const float *__restrict__ mask, int mstride,
                                                          float *__restrict__ concat, int batch, const EncSave sv,
                                                          const DropMasks dm) {
    typedef Enc8Geom<N_> Ge;
    constexpr bool SAVE = MODE >= 1, DROP = MODE == 2;
    constexpr int N = Ge::N, G = Ge::G, R = Ge::R, RV = Ge::RV, D = Ge::D, H = Ge::H, DK = Ge::DK, DFF = Ge::DFF;
    constexpr int AS = Ge::AS, QS = Ge::QS, FS = Ge::FS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *a16 = smem;                       // [R][AS] split-pair rows, 64 values each
    float *big = a16 + R * AS;
    float *qkv = big;                        // [R][QS] fp32 (P1 -> P2)
    float *hid = big;                        // [R][FS] split-pair rows, 128 values each (P5 -> P6)
    float *emb = big + 1024;                 // [R][AS] fp32 staging of the embedding (before layer 0 only; behind the staged x)
    float *rst = big + R * QS;               // [R][2 column tiles][mean, M2]: 32-column row statistics, exchanged between the two waves of a row tile
    float *maskb = big + Ge::BIG;            // [G][N][N]
    float *scl = maskb + G * N * N;          // this layer's scales (ENC_SC_*), staged with the vectors: a global load in a phase would
                                             // wait, in order, behind every weight prefetch in flight
    float *bst = scl + 8;                    // this layer's bias vectors: bqkv' [192] | bo [64] | b1' [128] | b2 [64]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int g0 = blockIdx.x * G;
    const int L = lay.L;
    auto live = [&](int row) { return row < RV && g0 + row / N < batch; };     // a real graph row of this workgroup
    // saved activations have rows_pad rows (a multiple of 256): with no spare rows in the tile (RV == R) the rows of the
    // graphs past the batch fall into that padding, and the stores need no condition (conditional stores make every later
    // s_waitcnt conservative)
    auto sv_live = [&](int row) { if constexpr (RV == R) return true; else return live(row); };
    // four keep words (16 keep bytes: columns c .. c + 3 of column groups 0 .. 3, 8 apart) of tile row `row` at one dropout
    // site, read ahead of the phase's MFMA chain; a row past the batch reads row 0's (its values go nowhere)
    auto keep4 = [&](const uint8_t *KLp, int row, int c, uint32_t (&kb)[4]) {
        const uint8_t *src = KLp + (size_t)(live(row) ? row : 0) * dm.per_layer + c;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) kb[gq] = *reinterpret_cast<const uint32_t *>(src + 8 * gq);
    };
    const size_t rows_pad = ((size_t)batch * N + 255) / 256 * 256;      // the concat is k-tile blocked (dnag_gemm16.hip.h kblock_at)
    const float *P16 = P + lay.delta16;      // fp16 hi/lo fragment copy of the layer matrices

    // this wave's tile of the residual stream (P3 / P6 jobs): rows tmt*32 .., columns tnt*32 ..
#ifdef ENC8_STAMPS
    long long ts_[128];
    int nts_ = 0;
#define E8_STAMP() do { if (nts_ < 128) ts_[nts_++] = __builtin_readcyclecounter(); } while (0)
    // sub-phase stamps of layer 5, every wave: 'S' = a point inside a phase (pinned), printed relative to the phase's first stamp
    long long sb_[24];
    int nsb_ = 0;
    const long long wc0_ = wall_clock64();       // constant 100 MHz
#define E8_SUB() do { if (l == 5 && nsb_ < 24) { __builtin_amdgcn_sched_barrier(0); sb_[nsb_++] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define E8_STAMP()
#define E8_SUB()
#endif
    E8_STAMP();
    const int tnt = wave >> 1, tmt = wave & 1;
    // P1: three of the 6 x 2 (column tile, row tile) jobs per wave: column tile `wave` (weights in bP) for both row tiles,
    // column tile qx = 4 + wave / 2 (bQ) for row tile wave & 1 -- the same straight-line code for every wave
    const int qx = 4 + (wave >> 1);

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
    // The vectors of a layer (CateLayout::l_vec) are fetched one 16-byte piece per thread well ahead (stage_fetch: before
    // the loop, then behind P5 of the layer before) and parked in LDS at the start of P0 (stage_put), so no epilogue
    // waits for a global load that queues behind the weight prefetches.
    constexpr int NST = Ge::VEC / 4;
    v4f st4 = {0.f, 0.f, 0.f, 0.f};
    auto stage_fetch = [&](int tid, int lstage) {
        // every thread loads (the surplus ones piece 0): a load under a condition leaves the compiler without an exact count
        // of the loads in flight, and every later s_waitcnt vmcnt falls back to waiting for (nearly) all of them
        st4 = *reinterpret_cast<const v4f *>(P + lay.layer0 + (size_t)lstage * lay.layer_stride + lay.l_sc + 4 * (tid < NST ? tid : 0));      // l_vec follows l_sc
    };
    auto stage_put = [&](int tid) {
        if (tid < NST) *reinterpret_cast<v4f *>(scl + 4 * tid) = st4;
    };
    // ---- stage inputs ---------------------------------------------------------------------------
    float *xin = qkv;                        // [R][V] alias
    for (int i = tid; i < R * V; i += 256) {
        const int g = i / (N * V);
        xin[i] = (i < RV * V && g0 + g < batch) ? x[(size_t)g0 * N * V + i] : 0.f;
    }
    for (int i = tid; i < G * N * N; i += 256) {
        const int g = i / (N * N), ij = i - g * N * N;
        const int gg = (g0 + g < batch) ? g0 + g : 0;
        maskb[i] = mask[(size_t)(mstride ? gg : 0) * N * N + ij];
    }
    __syncthreads();
    // ---- embedding: (x We^T + be) * sqrt(d) + PE + temb   (transformer.py:230, cate.py:149-159) ----
    {
        // 256 threads over D = 64 columns: a thread keeps its column for every row it
        // embeds, so the column's weights are read once
        const float *pe = P + lay.pe;
        const int c = tid & (D - 1);
        float wec[V];
#pragma unroll
        for (int v = 0; v < V; ++v) wec[v] = P[lay.emb_w + c * V + v];
        const float bec = P[lay.emb_b + c];
        for (int rr = tid / D; rr < R; rr += 256 / D) {
            const int g = rr / N, n = rr - g * N;
            float val = 0.f;
            if (rr < RV && g0 + g < batch) {
                float acc = 0.f;
#pragma unroll
                for (int v = 0; v < V; ++v) acc = fmaf(xin[rr * V + v], wec[v], acc);
                val = (acc + bec) * 8.0f;
                if constexpr (DROP) val = drop_apply(val, dm.keep[((size_t)g0 * N + rr) * D + c], dm.keep_prob);   // cate.py:147
                val = val + pe[n * D + c];
                val = val + temb[(size_t)(tstride ? g0 + g : 0) * D + c];
            }
            emb[rr * AS + c] = val;
            if constexpr (SAVE) if (sv_live(rr)) sv.h[((size_t)g0 * N + rr) * D + c] = val;
        }
    }
    __syncthreads();
    // the wave's tile of h: lane = batch row tmt*32 + r, registers 4 gq + e = columns tnt*32 + 8 gq + 4 h + e
    f32x16 hacc;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        const v4f v = *reinterpret_cast<const v4f *>(emb + (tmt * 32 + r) * AS + tnt * 32 + 8 * gq + 4 * h);
        hacc[4 * gq + 0] = v.x; hacc[4 * gq + 1] = v.y; hacc[4 * gq + 2] = v.z; hacc[4 * gq + 3] = v.w;
    }

    // Row statistics of h, first half (behind the phase that produced hacc): mean and centred sum of squares of the
    // lane's 16 values, merged with the other lane half (the tile's other 16 columns of the same row), parked in LDS for
    // the wave that holds the row's other column tile.  own_m / own_q keep this tile's pair.
    float own_m = 0.f, own_q = 0.f;
    auto stats_put = [&](int r, int h) {
        float s_ = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s_ += hacc[i];
        const float m16 = s_ * (1.0f / 16.0f);
        float q16 = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { const float d_ = hacc[i] - m16; q16 = fmaf(d_, d_, q16); }
        // v_permlane32_swap with both operands = v: the first result carries lanes 0..31 of v in both halves, the second
        // lanes 32..63
        const auto pm = __builtin_amdgcn_permlane32_swap(__float_as_uint(m16), __float_as_uint(m16), false, false);
        const auto pq = __builtin_amdgcn_permlane32_swap(__float_as_uint(q16), __float_as_uint(q16), false, false);
        const float ma = __uint_as_float(pm[0]), mb = __uint_as_float(pm[1]);
        const float dm = ma - mb;
        own_m = (ma + mb) * 0.5f;
        own_q = (__uint_as_float(pq[0]) + __uint_as_float(pq[1])) + dm * dm * 8.0f;        // n_a n_b / (n_a + n_b) = 8
        if (h == 0) *reinterpret_cast<float2 *>(rst + ((tmt * 32 + r) * 2 + tnt) * 2) = float2{own_m, own_q};
    };
    // second half (behind the barrier): the other tile's pair -> mean and 1 / std of the 64-wide row
    auto stats_get = [&](int r, float &mean, float &rstd) {
        const float2 o = *reinterpret_cast<const float2 *>(rst + ((tmt * 32 + r) * 2 + (tnt ^ 1)) * 2);
        // merged in a fixed order (tile 0 first) so that both waves of a row derive the same bits
        const float m0 = tnt ? o.x : own_m, m1 = tnt ? own_m : o.x, q0 = tnt ? o.y : own_q, q1 = tnt ? own_q : o.y;
        const float dm = m0 - m1;
        mean = (m0 + m1) * 0.5f;
        const float var = ((q0 + q1) + dm * dm * 16.0f) * (1.0f / 64.0f);
        rstd = __builtin_amdgcn_rsqf(var + 1e-5f);     // v_rsq_f32: 1 ulp, the argument is a normal number
    };
    // LN(h) * 16 of the wave's registers as split-pair quads into A16.  These rows are both the next layer's QKV operand
    // and -- the affine of the output norm being folded into final.linears.0 (CateLayout::f0_s) -- the concat slice of
    // the layer that just ended, which concat_out copies to HBM.
    float cmax = 0.f;            // SAVE modes: largest |concat| value this lane wrote (sv.amax_concat)
    auto norm_write = [&](int r, int h, int lc) {
        float mean, rstd;
        stats_get(r, mean, rstd);
        const int row = tmt * 32 + r;
        if constexpr (SAVE) {            // concat slice lc (>= 0): LN(h) gamma + beta of layer lc's output norm, fp32 rows
            if (lc >= 0) {
                const float *og = P + lay.layer0 + (size_t)lc * lay.layer_stride + lay.l_ong, *ob = P + lay.layer0 + (size_t)lc * lay.layer_stride + lay.l_onb;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int c0 = tnt * 32 + 8 * gq + 4 * h;
                    const v4f g4 = *reinterpret_cast<const v4f *>(og + c0), b4 = *reinterpret_cast<const v4f *>(ob + c0);
                    v4f o4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o4[e] = (hacc[4 * gq + e] - mean) * rstd * g4[e] + b4[e];
                    if (live(row)) {
                        *reinterpret_cast<v4f *>(concat + ((size_t)g0 * N + row) * lay.F + (size_t)lc * D + c0) = o4;
                        cmax = fmaxf(fmaxf(cmax, fmaxf(fabsf(o4[0]), fabsf(o4[1]))), fmaxf(fabsf(o4[2]), fabsf(o4[3])));
                    }
                }
            }
        }
        rstd *= 16.0f;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int c0 = tnt * 32 + 8 * gq + 4 * h;                 // first of the quad's columns
            const int at = (c0 >> 3) * 16 + (c0 & 7);                  // its place in a split-pair row, in halves (lo: + 8)
            v4f u;
#pragma unroll
            for (int e = 0; e < 4; ++e) u[e] = (hacc[4 * gq + e] - mean) * rstd;
            h4 hi, lo;
            split4(u, hi, lo);
            _Float16 *dst = reinterpret_cast<_Float16 *>(a16 + row * AS) + at;
            *reinterpret_cast<h4 *>(dst) = hi;
            *reinterpret_cast<h4 *>(dst + 8) = lo;
        }
    };
    // concat slice `lc` = the A16 image, copied out with full 128-byte lines: 64 rows x 2 chunks x 8 pieces of 16 bytes,
    // four pieces per thread (k-tile-blocked rows: dnag_gemm16.hip.h kblock_at).  Runs between the barrier behind P0 and
    // the one behind P1 (P2 is the next writer of A16).
    // piece k of a thread: 16 bytes of row pc / 16.  A piece that falls on a spare row or on a graph beyond the batch is
    // replaced by the same piece of the workgroup's first graph (always valid): the thread then reads and stores what another
    // thread stores to the same address -- a benign duplicate, and every store stays unconditional (see stage_fetch)
    auto piece_of = [&](int tid, int k, int &row, int &dw) {
        const int pc = tid + 256 * k;
        row = pc >> 4; dw = (pc & 15) * 4;
        // (8 nodes: all 64 rows are the workgroup's own, and rows of graphs beyond the batch are pad rows of the concat,
        // rows_pad >= 64 gridDim.x: stored as they are, finite and read by no one)
        if constexpr (RV < R) { if (row >= RV || g0 + row / N >= batch) row = row % N; }
    };
    auto concat_read = [&](int tid, v4f (&cc)[4]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int row, dw;
            piece_of(tid, k, row, dw);
            cc[k] = *reinterpret_cast<const v4f *>(a16 + row * AS + dw);
        }
    };
    auto concat_write1 = [&](int tid, const v4f &c, int k, int lc) {
        int row, dw;
        piece_of(tid, k, row, dw);
        *reinterpret_cast<v4f *>(concat + kblock_at((size_t)g0 * N + row, (size_t)lc * D + dw, rows_pad)) = c;
    };
    auto concat_write = [&](int tid, const v4f (&cc)[4], int lc) {
#pragma unroll
        for (int k = 0; k < 4; ++k) concat_write1(tid, cc[k], k, lc);
    };
    // ---- tile jobs.  All A fragments of a 32-row tile (K = 64: four k steps x hi, lo) are read from LDS BEFORE the
    // accumulation chain starts and the chain is one run of dependent MFMAs (a single chain issues back to back at the
    // matrix cores' rate); the compiler's own order -- two reads, wait, three MFMAs, per k step -- exposed the LDS latency
    // in every step (round 3 stamps: P1 3.4 k cycles for 36 MFMAs = 1.15 k).  The empty asm statements pin that order.
    typedef h8 afrag_t[4][2];
    auto load_a64 = [&](const float *ap, afrag_t &a) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            a[s][0] = *reinterpret_cast<const h8 *>(ap + 8 * s);
            a[s][1] = *reinterpret_cast<const h8 *>(ap + 8 * s + 4);
        }
    };
    // One 1 KB piece (k step q / 2, hi or lo) of column tile `nt` of a fragment-major matrix into b[4 q ..].  The weight
    // prefetch is issued piece by piece BEHIND the MFMAs of the accumulation chains (hook(i) runs after MFMA i of a chain):
    // the eight waves of a CU share one 64 B/clk path from the L2, a wave that issues 8-17 loads in a row blocks at issue
    // until the queue drains, and done in front of a barrier (round 3 stamps: 500-1300 cycles per burst) that wait was
    // the workgroup's; behind an MFMA it overlaps the 36 cycles the dependent chain waits anyway.
    typedef const __attribute__((address_space(1))) v4f *gv4p_;
    auto piece = [&](const float *W, int nt, int K, float *b, int q) {
        const v4f v = *(gv4p_)(W + ((size_t)nt * (K / 8) * 64 + (threadIdx.x & 63)) * 4 + (size_t)q * 256);
        b[4 * q + 0] = v.x; b[4 * q + 1] = v.y; b[4 * q + 2] = v.z; b[4 * q + 3] = v.w;
    };
    // MFMAs and vector-memory operations keep their program order; VALU, SALU and LDS instructions may move across
#define E8_HOLD() __builtin_amdgcn_sched_barrier(0x2 | 0x4 | 0x80 | 0x100 | 0x200 | 0x400)
    // acc += A W^T over K = 64 (SWAP: the transposed tile, lane = batch row, registers = columns)
    auto chain64 = [&](const afrag_t &a, const float *b, f32x16 &acc, auto swap, auto hook) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int o = 8 * s;
            const h8 bhi = __builtin_bit_cast(h8, v4f{b[o], b[o + 1], b[o + 2], b[o + 3]});
            const h8 blo = __builtin_bit_cast(h8, v4f{b[o + 4], b[o + 5], b[o + 6], b[o + 7]});
            if constexpr (decltype(swap)::value) {
                acc = mfma16(bhi, a[s][0], acc); hook(3 * s); E8_HOLD();
                acc = mfma16(blo, a[s][0], acc); hook(3 * s + 1); E8_HOLD();
                acc = mfma16(bhi, a[s][1], acc); hook(3 * s + 2); E8_HOLD();
            } else {
                acc = mfma16(a[s][0], bhi, acc); hook(3 * s); E8_HOLD();
                acc = mfma16(a[s][0], blo, acc); hook(3 * s + 1); E8_HOLD();
                acc = mfma16(a[s][1], bhi, acc); hook(3 * s + 2); E8_HOLD();
            }
        }
    };
    auto no_hook = [](int) {};
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
// lane ids derived afresh in every phase: with one set for the whole kernel the compiler hoists every phase's address
// arithmetic out of the layer loop and keeps (or spills) some sixty registers of loop invariants
#define E8_IDS() int tid_ = threadIdx.x; asm volatile("" : "+v"(tid_)); const int tid = tid_, r = tid_ & 31, h = (tid_ >> 5) & 1; (void)tid; (void)r; (void)h
#define E8_PIN() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
    stats_put(r, h);
    E8_STAMP();
    // First layer's vectors and q|k|v weights, requested HERE and in the loop's own order (vectors, tile `wave`, tile
    // qx) with no other vector-memory traffic pending: the compiler's s_waitcnt model merges the state at loop entry
    // with the back edge's, and only identical states let P1 wait for exactly the fragment it needs (fetched at kernel
    // entry, among the embedding's loads, its first MFMA waited for all but 7 of the 20 operations in flight).
    {
        const float *W = P16 + lay.layer0 + lay.l_wqkv;
        fetch64(W, wave, bP);                // the loop's own order: tile `wave` (behind P5), vectors and tile qx (behind P6)
        stage_fetch(tid, 0);
        fetch64(W, qx, bQ);
    }
    // Two workgroups share a CU (256 CUs x 2 at B = 4096) and the wave arbiter serves the older wave first: the first 256
    // workgroups ran 115 us, their younger partners 145 (wall-clock stamps, round 3).  The two generations (bit 8 of the
    // workgroup index: the dispatcher hands the workgroups of an XCD, index / 8, to its 32 CUs in order) swap the raised
    // priority every layer, so that both advance at the same rate and neither is left to finish alone.
    const int gen = (blockIdx.x >> 8) & 1;
    for (int l = 0; l <= L; ++l) {
#ifndef ENC8_NO_PRIO
        if ((l ^ gen) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#endif
        // ---- P0: statistics of h -> concat slice l - 1, A16 = LN1(h) -----------------------------------
        __syncthreads();                     // the statistics of both column tiles are in LDS; every reader of A16 / bst is done
        E8_STAMP();
        {
            E8_IDS();
            if (l < L) stage_put(tid);
            norm_write(r, h, l - 1);
        }
        if (l == L) {
            __syncthreads();
            if constexpr (!SAVE) {
                v4f cc[4];
                concat_read(tid, cc);
                concat_write(tid, cc, L - 1);
            }
            break;
        }
        const float *LP16 = P16 + lay.layer0 + (size_t)l * lay.layer_stride;
        const float *SC = scl;
        // keep masks of this layer's sites, row 0 of the workgroup (DROP only)
        const uint8_t *KL = nullptr;
        if constexpr (DROP) KL = dm.keep + dm.rows * D + ((size_t)l * dm.rows + (size_t)g0 * N) * dm.per_layer;
        __syncthreads();
        E8_STAMP();
        // ---- P1: q|k|v (and the concat slice of the layer before: the rows P1 reads) -----------------------
        {
            E8_IDS();
            const float inv = SC[ENC_SC_INV_QKV];
            const float *bias = bst;
            const int mx = wave & 1;
            afrag_t xa, ya;
            load_a64(a16 + (mx * 32 + r) * AS + h * (D / 2), xa);
            load_a64(a16 + ((mx ^ 1) * 32 + r) * AS + h * (D / 2), ya);
            E8_PIN();
            auto put = [&](const f32x16 &acc, int nt, int mt) {
                const int col = nt * 32 + r;
                const float bv = bias[col];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int lr = mt * 32 + acc_row(i, h);
                    const float v = acc[i] * inv + bv;
                    qkv[lr * QS + col] = v;
                    if constexpr (SAVE) if (sv_live(lr)) sv.qkv[((size_t)l * sv.rows + (size_t)g0 * N + lr) * (3 * D) + col] = v;
                }
            };
            f32x16 c1 = zero16, c2 = zero16, c3 = zero16;
            E8_SUB();
            // behind the 36 MFMAs: the 16 pieces of W2 (P6) and, once both jobs on column tile `wave` are issued, the 8 of Wo
            // (P3) into the registers they free
            const float *W2p = LP16 + lay.l_w2, *Wop = LP16 + lay.l_wo;
            // (20 nodes: the attention's registers cannot spare the 64 of W2 in flight; it is fetched behind P2 instead)
            chain64(xa, bP, c1, std::false_type{}, [&](int i) { if constexpr (N <= 8) piece(W2p, tnt, DFF, bR, i); });
            E8_SUB();
            chain64(ya, bP, c2, std::false_type{}, [&](int i) { if constexpr (N <= 8) { if (i < 4) piece(W2p, tnt, DFF, bR, 12 + i); } });
            put(c1, wave, mx);
            // The concat slice of the layer before (= the A16 rows this phase reads) is read once more from LDS here, when the
            // first two jobs have released their registers, and stored behind the last four MFMAs.  Every store is
            // unconditional: layer 0 writes its (meaningless) image into slice 0, which the same threads overwrite with
            // the real slice one layer later (same wave, same addresses: in order).
            v4f cc[4];
            if constexpr (!SAVE) concat_read(tid, cc);
            const int lc = l > 0 ? l - 1 : 0;
            chain64(xa, bQ, c3, std::false_type{}, [&](int i) {
                if (i < 8) { if constexpr (N <= 8) piece(Wop, tnt, D, bP, i); }
                else if constexpr (!SAVE) concat_write1(tid, cc[i - 8], i - 8, lc);
            });
            put(c2, wave, mx ^ 1);
            put(c3, qx, mx);
            E8_SUB();
        }
        __syncthreads();
        E8_STAMP();
        // ---- P2: attention, one thread per (graph, head, query pair) ----------------------------------------
        {
            E8_IDS();
            if constexpr (N <= 8) {
                const int qp = tid & 3, hd = (tid >> 2) & 7, g = tid >> 5;
                const int row0 = g * N + 2 * qp;
                const float sa_o = SC[ENC_SC_SA_O];
                v2f q[DK];
                // keep bytes of the two queries' N probabilities of this head, read before anything else of the phase
                uint32_t kp[2][N / 4];
                if constexpr (DROP) {                 // transformer.py:19
                    const uint8_t *src = KL + (size_t)(g0 + g < batch ? row0 : 0) * dm.per_layer + hd * N;
#pragma unroll
                    for (int e = 0; e < N / 4; ++e) {
                        kp[0][e] = *reinterpret_cast<const uint32_t *>(src + 4 * e);
                        kp[1][e] = *reinterpret_cast<const uint32_t *>(src + dm.per_layer + 4 * e);
                    }
                }
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
                static_assert(N % 4 == 0, "mask rows as 16-byte pieces");
                float mk[2 * N];                     // the two queries' mask rows, adjacent in LDS
                {
                    const float *mrow = maskb + (g * N + 2 * qp) * N;
#pragma unroll
                    for (int e = 0; e < 2 * N; e += 4) {
                        const v4f m4 = *reinterpret_cast<const v4f *>(mrow + e);
                        mk[e] = m4.x; mk[e + 1] = m4.y; mk[e + 2] = m4.z; mk[e + 3] = m4.w;
                    }
                }
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
                    if (mk[j] == 0.f) acc.x = -1e9f;            // transformer.py:16
                    if (mk[N + j] == 0.f) acc.y = -1e9f;
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
                    v2f p = s[j] * rden;
                    if constexpr (DROP) {
                        p.x = drop_apply(p.x, (uint8_t)(kp[0][j >> 2] >> (8 * (j & 3))), dm.keep_prob);
                        p.y = drop_apply(p.y, (uint8_t)(kp[1][j >> 2] >> (8 * (j & 3))), dm.keep_prob);
                    }
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
            } else {
                constexpr int QP = N / 2;
                const bool act = tid < G * H * QP;             // N = 20: 240 of the 256 threads
                const int t2 = act ? tid : 0;
                const int qp = t2 % QP, hd = (t2 / QP) % H, g = t2 / (QP * H);
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
                const float *mrow = maskb + (g * N + 2 * qp) * N;
                static_assert(N % 4 == 0, "mask rows as 16-byte pieces");
                v4f mk[2][N / 4];                    // the two queries' mask rows
#pragma unroll
                for (int e = 0; e < N / 4; ++e) {
                    mk[0][e] = *reinterpret_cast<const v4f *>(mrow + 4 * e);
                    mk[1][e] = *reinterpret_cast<const v4f *>(mrow + N + 4 * e);
                }
                v2f o[DK];
#pragma unroll
                for (int e = 0; e < DK; ++e) o[e] = v2f{0.f, 0.f};
                auto score = [&](int j) {
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
                    if (mk[0][j >> 2][j & 3] == 0.f) acc.x = -1e9f;          // transformer.py:16
                    if (mk[1][j >> 2][j & 3] == 0.f) acc.y = -1e9f;
                    return acc;
                };
                auto add_v = [&](int j, v2f p) {
                    const float *vp = qkv + (g * N + j) * QS + 2 * D + hd * DK;
#pragma unroll
                    for (int e = 0; e < DK; e += 4) {
                        const v4f v4 = *reinterpret_cast<const v4f *>(vp + e);
                        o[e] = pk_fma(p, v2f{v4.x, v4.x}, o[e]);
                        o[e + 1] = pk_fma(p, v2f{v4.y, v4.y}, o[e + 1]);
                        o[e + 2] = pk_fma(p, v2f{v4.z, v4.z}, o[e + 2]);
                        o[e + 3] = pk_fma(p, v2f{v4.w, v4.w}, o[e + 3]);
                    }
                };
                {
                    // 20 keys: two blocks of ten with a running maximum (40 registers of scores at once, beside the 112 of weight
                    // prefetches in flight, spilled): o and the denominator are rescaled by exp(m_old - m_new) between the blocks
                    // and o is divided once at the end -- a few ulp from softmax-then-product, well inside the path's tolerance
                    constexpr int KB = N / 2;
                    v2f mx = {-3.0e38f, -3.0e38f}, den = {0.f, 0.f};
#pragma unroll
                    for (int b0 = 0; b0 < N; b0 += KB) {
                        v2f s[KB];
                        v2f mb = mx;
#pragma unroll
                        for (int j = 0; j < KB; ++j) {
                            s[j] = score(b0 + j); mb = pk_max(mb, s[j]);
                        }
                        const v2f sc = v2f{__expf(mx.x - mb.x), __expf(mx.y - mb.y)};      // first block: exp(-3e38 - m) = 0
                        den = den * sc;
#pragma unroll
                        for (int e = 0; e < DK; ++e) o[e] = o[e] * sc;
#pragma unroll
                        for (int j = 0; j < KB; ++j) {
                            const v2f p = v2f{__expf(s[j].x - mb.x), __expf(s[j].y - mb.y)};
                            den += p;
                            add_v(b0 + j, p);
                        }
                        mx = mb;
                    }
                    const v2f rden = v2f{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
#pragma unroll
                    for (int e = 0; e < DK; ++e) o[e] = o[e] * rden;
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
                    if (act) {
                        *reinterpret_cast<h8 *>(dst) = hi;
                        *reinterpret_cast<h8 *>(dst + 4) = lo;
                    }
                }
            }
        }
        if constexpr (N > 8) { fetch64(LP16 + lay.l_wo, tnt, bP); fetch128(LP16 + lay.l_w2, tnt, bR); }
        __syncthreads();
        E8_STAMP();
        // ---- P3: h += attn Wo^T + bo --------------------------------------------------------------------
        {
            E8_IDS();
            const float inv = SC[ENC_SC_INV_O];
            afrag_t xa;
            load_a64(a16 + (tmt * 32 + r) * AS + h * (D / 2), xa);
            uint32_t kb3[4] = {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u};
            if constexpr (DROP) keep4(KL, tmt * 32 + r, H * N + tnt * 32 + 4 * h, kb3);      // transformer.py:80
            E8_PIN();
            f32x16 acc = zero16;
            E8_SUB();
            const float *W1p = LP16 + lay.l_w1;
            // D[n][m]: lane = batch row m, registers = columns n (as in P5); behind the MFMAs: W1's tile for P5
            chain64(xa, bP, acc, std::true_type{}, [&](int i) { if (i < 8) piece(W1p, wave, D, bQ, i); });
            E8_SUB();
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int c0 = tnt * 32 + 8 * gq + 4 * h;
                const v4f b4 = *reinterpret_cast<const v4f *>(bst + 3 * D + c0);
                const int lr = tmt * 32 + r;
                const uint32_t kb = kb3[gq];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float yv = acc[4 * gq + e] * inv + b4[e];
                    if constexpr (DROP) yv = drop_apply(yv, (uint8_t)(kb >> (8 * e)), dm.keep_prob);
                    hacc[4 * gq + e] = yv + hacc[4 * gq + e];
                }
                if constexpr (SAVE) if (sv_live(lr))
                    *reinterpret_cast<v4f *>(sv.h1 + ((size_t)l * sv.rows + (size_t)g0 * N + lr) * D + c0) =
                        v4f{hacc[4 * gq], hacc[4 * gq + 1], hacc[4 * gq + 2], hacc[4 * gq + 3]};
            }
            stats_put(r, h);
            E8_SUB();
        }
        __syncthreads();
        E8_STAMP();
        // ---- P4: A16 = LN2(h), from the registers ------------------------------------------------------------
        {
            E8_IDS();
            norm_write(r, h, -1);
        }
        __syncthreads();
        E8_STAMP();
        // ---- P5: hidden = relu(LN2(h) W1'^T + b1'), transposed tiles: lane <-> batch row ---------------------
        {
            E8_IDS();
            const float inv = SC[ENC_SC_INV_F1], sa_f2 = SC[ENC_SC_SA_F2];
            const int nt = wave;
            // the lane's 16 output columns nt*32 + 8 gq + 4 h + (0..3), gq = 0..3
            const float *b1p = bst + 4 * D + nt * 32 + 4 * h;
            afrag_t xa, ya;
            load_a64(a16 + r * AS + h * (D / 2), xa);
            load_a64(a16 + (32 + r) * AS + h * (D / 2), ya);
            v4f bias4[4];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) bias4[gq] = *reinterpret_cast<const v4f *>(b1p + 8 * gq);
            uint32_t kb5[2][4] = {{0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u}, {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u}};
            if constexpr (DROP) {                                                               // transformer.py:54
                keep4(KL, r, H * N + D + nt * 32 + 4 * h, kb5[0]);
                keep4(KL, 32 + r, H * N + D + nt * 32 + 4 * h, kb5[1]);
            }
            E8_PIN();
            auto put = [&](const f32x16 &acc, int mt) {
                _Float16 *hrow = reinterpret_cast<_Float16 *>(hid + (mt * 32 + r) * FS);
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    v4f f;
                    const int lr = mt * 32 + r, c0 = nt * 32 + 8 * gq + 4 * h;
                    const uint32_t kb = kb5[mt][gq];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float fv = fmaxf(acc[4 * gq + e] * inv + bias4[gq][e], 0.f);
                        if constexpr (DROP) fv = drop_apply(fv, (uint8_t)(kb >> (8 * e)), dm.keep_prob);
                        f[e] = fv;
                    }
                    if constexpr (SAVE) if (sv_live(lr)) *reinterpret_cast<v4f *>(sv.f + ((size_t)l * sv.rows + (size_t)g0 * N + lr) * DFF + c0) = f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) f[e] *= sa_f2;
                    h4 hi, lo;
                    split4(f, hi, lo);
                    _Float16 *dst = hrow + (nt * 4 + gq) * 16 + 4 * h;
                    *reinterpret_cast<h4 *>(dst) = hi;
                    *reinterpret_cast<h4 *>(dst + 8) = lo;
                }
            };
            f32x16 c0 = zero16, c1 = zero16;
            E8_SUB();
            // D[n][m]: lane = batch row m, registers = columns n; behind the first job's MFMAs: the next layer's q|k|v tile
            // `wave` (the last layer fetches its own once more: an unconditional, exactly counted prefetch)
            const float *NP16 = P16 + lay.layer0 + (size_t)(l + 1 < L ? l + 1 : l) * lay.layer_stride;
            const float *Wqp = NP16 + lay.l_wqkv;
            chain64(xa, bQ, c0, std::true_type{}, [&](int i) { if (i < 8) piece(Wqp, wave, D, bP, i); });
            E8_SUB();
            chain64(ya, bQ, c1, std::true_type{}, no_hook);
            put(c0, 0);
            put(c1, 1);
            E8_SUB();
        }
        __syncthreads();
        E8_STAMP();
        // ---- P6: h += hidden W2^T + b2 ------------------------------------------------------------------------
        {
            E8_IDS();
            const float inv = SC[ENC_SC_INV_F2];
            const float *ap = hid + (tmt * 32 + r) * FS + h * (DFF / 2);
            // K = 128 = eight k steps in four pairs: two pairs of fragments are in registers before the chain starts, pair
            // p + 2 is read behind the MFMAs of pair p (into the registers that pair frees)
            h8 a[8][2];
            auto ld = [&](int s) {
                a[s][0] = *reinterpret_cast<const h8 *>(ap + 8 * s);
                a[s][1] = *reinterpret_cast<const h8 *>(ap + 8 * s + 4);
            };
            // behind the MFMAs: the next layer's vectors and its q|k|v tile qx (bQ is free since P5)
            const int ln_ = l + 1 < L ? l + 1 : l;
            const float *Wqp = P16 + lay.layer0 + (size_t)ln_ * lay.layer_stride + lay.l_wqkv;
            auto hook = [&](int i) {
                if (i == 0) stage_fetch(tid, ln_);
                else if (i <= 8) piece(Wqp, qx, D, bQ, i - 1);
            };
            auto mm = [&](int s, f32x16 &acc) {
                const int o = 8 * s;
                const h8 bhi = __builtin_bit_cast(h8, v4f{bR[o], bR[o + 1], bR[o + 2], bR[o + 3]});
                const h8 blo = __builtin_bit_cast(h8, v4f{bR[o + 4], bR[o + 5], bR[o + 6], bR[o + 7]});
                acc = mfma16(bhi, a[s][0], acc); hook(3 * s); E8_HOLD();
                acc = mfma16(blo, a[s][0], acc); hook(3 * s + 1); E8_HOLD();
                acc = mfma16(bhi, a[s][1], acc); hook(3 * s + 2); E8_HOLD();
            };
            ld(0); ld(1); ld(2); ld(3);
            uint32_t kb6[4] = {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u};
            if constexpr (DROP) keep4(KL, tmt * 32 + r, H * N + D + DFF + tnt * 32 + 4 * h, kb6);      // transformer.py:108
            E8_PIN();
            f32x16 acc = zero16;
            E8_SUB();
            mm(0, acc); mm(1, acc);
            E8_SUB();
            ld(4); ld(5);
            E8_HOLD();
            mm(2, acc); mm(3, acc);
            ld(6); ld(7);
            E8_HOLD();
            mm(4, acc); mm(5, acc);
            mm(6, acc); mm(7, acc);
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int c0 = tnt * 32 + 8 * gq + 4 * h;
                const v4f b4 = *reinterpret_cast<const v4f *>(bst + 4 * D + DFF + c0);
                const int lr = tmt * 32 + r;
                const uint32_t kb = kb6[gq];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float yv = acc[4 * gq + e] * inv + b4[e];
                    if constexpr (DROP) yv = drop_apply(yv, (uint8_t)(kb >> (8 * e)), dm.keep_prob);
                    hacc[4 * gq + e] = yv + hacc[4 * gq + e];
                }
                if constexpr (SAVE) if (sv_live(lr))
                    *reinterpret_cast<v4f *>(sv.h + ((size_t)(l + 1) * sv.rows + (size_t)g0 * N + lr) * D + c0) =
                        v4f{hacc[4 * gq], hacc[4 * gq + 1], hacc[4 * gq + 2], hacc[4 * gq + 3]};
            }
            stats_put(r, h);
            E8_SUB();
        }
    }
    if constexpr (SAVE) {
        if (sv.amax_concat != nullptr) {
            cmax = fmaxf(cmax, dpp_mov<0xB1>(cmax));
            cmax = fmaxf(cmax, dpp_mov<0x4E>(cmax));
            cmax = fmaxf(cmax, dpp_mov<0x141>(cmax));
            cmax = fmaxf(cmax, dpp_mov<0x140>(cmax));      // every lane: the maximum of its row of 16
            if ((tid & 15) == 0) atomicMax(reinterpret_cast<unsigned int *>(sv.amax_concat), __float_as_uint(cmax));
        }
    }
#ifdef ENC8_STAMPS
    if (blockIdx.x == 5 && tid == 0) for (int i = 1; i < nts_; ++i) printf("e8 %d %lld\n", i, ts_[i] - ts_[i - 1]);
    if ((blockIdx.x % 37) == 5 && tid == 0) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        printf("wg %d xcc %u se %u cu %u : start %lld life %lld x10ns, %lld cycles\n", (int)blockIdx.x, xcc & 15, (hwid >> 13) & 7, (hwid >> 8) & 15, wc0_, wall_clock64() - wc0_, ts_[nts_ - 1] - ts_[0]);
    }
    // layer 5's phase stamps are ts_[2 + 7 * 5 + k]: k = 0 P0 start ... 1 P1 start, 3 P3 start, 5 P5 start, 6 P6 start
    if (blockIdx.x == 5 && lane == 0) {
        const long long b1 = ts_[2 + 35 + 1], b3 = ts_[2 + 35 + 3], b5 = ts_[2 + 35 + 5], b6 = ts_[2 + 35 + 6];
        printf("sub wave %d: P1 pre %lld c1 %lld end %lld (next bar %lld) | P3 pre %lld chain %lld end %lld (bar %lld) | P5 pre %lld c0 %lld end %lld (bar %lld) | P6 pre %lld 6mfma %lld end %lld (bar %lld)\n", wave,
               sb_[0] - b1, sb_[1] - b1, sb_[2] - b1, ts_[2 + 35 + 2] - b1, sb_[3] - b3, sb_[4] - b3, sb_[5] - b3, ts_[2 + 35 + 4] - b3,
               sb_[6] - b5, sb_[7] - b5, sb_[8] - b5, b6 - b5, sb_[9] - b6, sb_[10] - b6, sb_[11] - b6, ts_[2 + 35 + 7] - b6);
    }
#endif
}

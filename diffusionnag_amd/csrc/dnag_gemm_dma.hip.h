// Head MLP GEMMs of the sampling path, third generation (final.linears.0 and final.linears.1 + .2, cate.py:55-73):
// the split arithmetic and the split-pair operand rows of dnag_gemm16.hip.h, with
//   * v_mfma_f32_16x16x32_f16 instead of 32x32x16: on this chip the matrix cores are power limited, and the 16x16x32 form
//     sustains 1.12-1.13 x the rate on random operands (tests/micro/mfma_shapes.hip: 1844 against 1632 TFLOP/s);
//   * LDS-DMA staging (global_load_lds_dwordx4): no staging registers, no ds_write pass.  One wave instruction fills
//     8 rows x 128 B of an UNPADDED image; bank conflicts are avoided by an XOR swizzle instead of padding: 16-byte
//     slot p of row r sits at physical slot p ^ (r & 7), applied to the per-lane SOURCE address of the DMA (its LDS
//     destination is lane-linear by construction) and to the fragment reads.  Within a staged row of 32 k the slots are
//     planar: p = kg for the fp16 hi half of k group kg (8 values), p = 4 + kg for its lo half, so lane (row l & 15,
//     k group l >> 4) of a 16x16x32 fragment reads one slot for hi and one for lo; both are conflict free;
//   * a phased main loop.  A k tile (32 k) is staged as four half tiles of 128 rows x 128 B = 16 KB (A rows 0-127, B rows
//     0-127, B rows 128-255, A rows 128-255), two k tiles of them in LDS (128 KB).  A wave owns a 64 x 32 piece of each
//     quadrant (A half, B half) of the 256 x 256 block tile and works through the quadrants in two phases per k tile;
//     a phase is a load segment (fragment reads of the half tiles new to the phase, four DMA pieces, a counted vmcnt), a
//     raw s_barrier, a compute segment (48 MFMAs between an s_setprio pair: it keeps hipcc from moving them across the
//     barriers), a raw s_barrier.  The two waves of a SIMD (w and w + 4: groups 0 and 1) run one segment apart, so in
//     every segment one of them feeds the matrix cores while the other reads LDS and issues DMA.  The DMA stays in flight
//     across the barriers (no fence) for at least one k tile and is retired by a counted vmcnt in the load segment of the
//     phase BEFORE the one that reads it (the barriers between order the other waves' pieces); a half tile's slot is
//     restaged two phases after its last read, one phase after for group 1, whose segments lie later.
//     Measured on the way here (same box, final.linears.1 at B = 4096): with the DMA removed a one-barrier loop ran at
//     93 % MFMA issue, with it 30-50 % slower whether the 64 pieces of a k tile were issued as a burst, spread between
//     the MFMAs or staggered between the waves of a SIMD; a four-phase version of this loop (24 MFMAs per segment, eight
//     barriers per k tile) was 1-2.5 % slower than this one; without the s_setprio pair or without the stagger +15 %.
// Block tile 256 x 256 x 32, 8 waves (2 x 4), one workgroup per CU.
// Included by dnag_lib.hip.
#pragma once
#include "dnag_common.h"
#include "dnag_gemm16.hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma16x(h8 a, h8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

struct DmaGeom {
    static constexpr int BM = 256, BN = 256, BK = 32, T = 512;
    static constexpr int ROWW = 32;                                   // dwords per staged row (32 k: 16 of hi, 16 of lo)
    static constexpr int HALF = 128 * ROWW;                           // floats per half tile
    static constexpr size_t lds_bytes = 2 * 4 * (size_t)HALF * sizeof(float);
};

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

// C = act(A W^T + bias); A [M][K] split-pair rows (pitch K floats, or k-tile blocked if A_KBLK), W [Nout][K] split-pair
// rows, k-tile blocked (dnag_gemm16.hip.h kblock_rows); sc: the GEMM's power-of-two scales (dnag_common.h HEAD_SC_F0 /
// _F1).  OUT_SPLIT: C is written as k-tile-blocked split-pair rows scaled by sc[1] (final.linears.0).
// FUSE_V > 0: operands swapped (lane <-> batch row) and final.linears.2 (W3 [FUSE_V][Nout]) applied to the ELU output in
// registers on the same MFMAs; parts [Nout / 256][M][FUSE_V] receives this block's partial sums (dnag_gemm16.hip.h).
// M and Nout multiples of 256, K a multiple of 32.
template <int ACT, bool OUT_SPLIT, int FUSE_V, bool A_KBLK>
__global__ __launch_bounds__(512, 1) void gemm_dma_kernel(const float *__restrict__ A, const float *__restrict__ W,
                                                          const float *__restrict__ bias, float *__restrict__ C, int M, int Nout,
                                                          int K, const float *__restrict__ W3, float *__restrict__ parts,
                                                          const float *__restrict__ sc) {
    typedef DmaGeom Ge;
    constexpr int BM = Ge::BM, BN = Ge::BN, ROWW = Ge::ROWW, HALF = Ge::HALF;
    static_assert(OUT_SPLIT != (FUSE_V > 0), "either the split-pair output or the fused last layer");
    extern __shared__ __attribute__((aligned(16))) float smem[];     // [k tile parity][A0, B0, B1, A1][128][ROWW]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int wm = (wave >> 2) & 1, wn = wave & 3;
    const int grp = wave >> 2;                  // waves w and w + 4 share a SIMD: group 1 runs one segment behind group 0

    const int nbn = Nout / BN;
    const int nwg = gridDim.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7;
    const int wg = ((xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    const int tm = wg / nbn, tn = wg - tm * nbn;
    const float out_inv = sc[0], sa_out = sc[1];

    // ---- LDS-DMA staging: a half tile is staged by the four waves of ONE group, wave w4 of the group fills its rows
    // 32 w4 .. 32 w4 + 31, 8 rows per instruction.  Lane = (row in the piece, physical slot); the slot's content is planar
    // slot p = ps ^ (row & 7), i.e. 16-byte chunk 2 (p & 3) + (p >> 2) of the row's 128 bytes in global memory (split-pair
    // order: hi, lo per k group).
    const int prow = lane >> 3, ps = lane & 7, pp = ps ^ prow;       // (32 w4 + 8 q + prow) & 7 == prow
    const int chunk = 2 * (pp & 3) + (pp >> 2);
    // W is k-tile blocked ([k tile][row][32 dwords], kblock_rows), A too if A_KBLK, else row-major with pitch K
    const size_t a_row = A_KBLK ? ROWW : K, a_kt = A_KBLK ? (size_t)M * ROWW : Ge::BK, w_kt = (size_t)Nout * ROWW;
    const int w4 = wave & 3;
    const float *ag = A + (size_t)(tm * BM + w4 * 32 + prow) * a_row + chunk * 4;
    const float *wgp = W + (size_t)(tn * BN + w4 * 32 + prow) * ROWW + chunk * 4;
    // half tile kinds in slot order: 0 = A rows 0-127, 1 = B rows 0-127, 2 = B rows 128-255, 3 = A rows 128-255
    auto stage_half_grp = [&](int kt, int kind) {
        const bool isa = (kind == 0 || kind == 3);
        const size_t row = isa ? a_row : ROWW;
        const float *src = (isa ? ag : wgp) + ((kind >= 2) ? 128 * row : 0) + (size_t)kt * (isa ? a_kt : w_kt);
        float *dst = smem + ((kt & 1) * 4 + kind) * HALF + w4 * 32 * ROWW;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + q * 8 * row), (lptr_t)(dst + q * 8 * ROWW), 16, 0, 0);
    };

    f32x4 acc[2][2][4][2];          // [A half][B half][row tile][column tile]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[h][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment addresses inside a half tile: row l15 of a 16-row tile, k group g; hi slot g ^ (row & 7), lo slot that ^ 4
    const int slot_hi = (g ^ (l15 & 7)) * 4;
    const float *fa_hi = smem + (wm * 64 + l15) * ROWW + slot_hi, *fa_lo = smem + (wm * 64 + l15) * ROWW + (slot_hi ^ 16);
    const float *fb_hi = smem + (wn * 32 + l15) * ROWW + slot_hi, *fb_lo = smem + (wn * 32 + l15) * ROWW + (slot_hi ^ 16);
    h8 ah[4], al[4], bh[2][2], bl[2][2];
    auto read_a = [&](int par, int kind) {
        const int o = (par * 4 + kind) * HALF;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ah[i] = *reinterpret_cast<const h8 *>(fa_hi + o + i * 16 * ROWW);
            al[i] = *reinterpret_cast<const h8 *>(fa_lo + o + i * 16 * ROWW);
        }
    };
    auto read_b = [&](int par, int b) {
        const int o = (par * 4 + 1 + b) * HALF;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bh[b][j] = *reinterpret_cast<const h8 *>(fb_hi + o + j * 16 * ROWW);
            bl[b][j] = *reinterpret_cast<const h8 *>(fb_lo + o + j * 16 * ROWW);
        }
    };
    // 24 MFMAs of one quadrant; the three of one accumulator are seven others apart
    auto quadrant = [&](int h, int b) {
        __builtin_amdgcn_s_setprio(1);
        if constexpr (FUSE_V > 0) {          // D[n][m]: lane <-> batch row
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[h][b][i][j] = mfma16x(bh[b][j], ah[i], acc[h][b][i][j]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[h][b][i][j] = mfma16x(bl[b][j], ah[i], acc[h][b][i][j]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[h][b][i][j] = mfma16x(bh[b][j], al[i], acc[h][b][i][j]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[h][b][i][j] = mfma16x(ah[i], bh[b][j], acc[h][b][i][j]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[h][b][i][j] = mfma16x(ah[i], bl[b][j], acc[h][b][i][j]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[h][b][i][j] = mfma16x(al[i], bh[b][j], acc[h][b][i][j]);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    // segment boundary: raw barrier, no fence (a fence would drain the DMA in flight)
#define DNAG_BAR() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
    const int KT = K / Ge::BK;
#ifdef DNAG_GEMM_TILE_STAMPS
    const long long tw0 = wall_clock64();
#endif
    // Two phases per k tile (48 MFMAs per compute segment, four barriers per k tile): X = quadrants (A0, B0), (A0, B1) after
    // reading B0, B1, A0; Y = (A1, B1), (A1, B0) after reading A1.  A half tile is staged by the four waves of one group
    // (four pieces each), in the group's own load segments: group 0 stages B1 of k tile t + 1 in X(t) and A1 of t + 1 in
    // Y(t); group 1, whose segments lie one later, may reuse a slot one phase earlier and stages B0 of t + 1 in X(t) and
    // A0 of t + 2 in Y(t).  Every half tile is in flight for at least two phases (one k tile) and is retired, by the
    // group that issued it, in the load segment of the phase before its first read: vmcnt(4) after the phase's own four
    // pieces (group 1 has nothing to retire in X).
    if (grp == 0) { stage_half_grp(0, 0); stage_half_grp(0, 1); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    else {
        stage_half_grp(0, 2); stage_half_grp(0, 3);
        if (KT > 1) { stage_half_grp(1, 0); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    DNAG_BAR();
#ifdef DNAG_GEMM_TILE_STAMPS
    const long long tw1 = wall_clock64();
#endif
    if (grp == 1) DNAG_BAR();
    for (int kt = 0; kt < KT; ++kt) {
        const int par = kt & 1;
        // phase X
        read_b(par, 0);
        read_b(par, 1);
        read_a(par, 0);
        if (grp == 0) {
            if (kt + 1 < KT) { stage_half_grp(kt + 1, 2); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (kt + 1 < KT) stage_half_grp(kt + 1, 1);
        DNAG_BAR();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        quadrant(0, 0);
        quadrant(0, 1);
        DNAG_BAR();
        // phase Y
        read_a(par, 3);
        if (grp == 0) {
            if (kt + 1 < KT) { stage_half_grp(kt + 1, 3); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            if (kt + 2 < KT) { stage_half_grp(kt + 2, 0); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        DNAG_BAR();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        quadrant(1, 1);
        quadrant(1, 0);
        DNAG_BAR();
    }
    if (grp == 0) DNAG_BAR();
#undef DNAG_BAR
    __syncthreads();
#ifdef DNAG_GEMM_TILE_STAMPS
    const long long tw2 = wall_clock64();
#define DNAG_TILE_REPORT() do { if (tid == 0) { unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid)); unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); \
        printf("tile V %d blk %d xcc %u cu %u se %u : start %lld prologue %lld loop %lld epilogue %lld\n", FUSE_V, (int)blockIdx.x, xcc & 15, (hwid >> 8) & 15, (hwid >> 13) & 7, tw0, tw1 - tw0, tw2 - tw1, wall_clock64() - tw2); } } while (0)
#else
#define DNAG_TILE_REPORT()
#endif

    // block-relative row of row tile (h, i) and column of column tile (b, j) of this wave
    auto row_of = [&](int h, int i) { return h * 128 + wm * 64 + i * 16; };
    auto col_of = [&](int b, int j) { return b * 128 + wn * 32 + j * 16; };

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
        // columns col_of(b, j) + 4 g + e; one 16x16x32 MFMA takes the two column tiles of B half b: k slot 8 g + 4 j + e of
        // lane group g <-> column col_of(b, j) + 4 g + e, for the W3 fragment (row v = l15) and for x alike.
        h8 w3h[2], w3l[2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                v4f w4 = {0.f, 0.f, 0.f, 0.f};
                if (l15 < V) w4 = *reinterpret_cast<const v4f *>(w3s + l15 * BN + col_of(b, j) + 4 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    _Float16 hi, lo;
                    split16(w4[e], sw3, hi, lo);
                    w3h[b][4 * j + e] = hi; w3l[b][4 * j + e] = lo;
                }
            }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 pa = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    h8 xh, xl;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const v4f b4 = *reinterpret_cast<const v4f *>(b1s + col_of(b, j) + 4 * g);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float x = fmaf(acc[h][b][i][j][e], out_inv, b4[e]);
                            if (ACT == ACT_ELU) x = x > 0.f ? x : __expf(x) - 1.0f;
                            _Float16 hi, lo;
                            split16(x, sa_out, hi, lo);
                            xh[4 * j + e] = hi; xl[4 * j + e] = lo;
                        }
                    }
                    pa = mfma16x(w3h[b], xh, pa);
                    pa = mfma16x(w3l[b], xh, pa);
                    pa = mfma16x(w3h[b], xl, pa);
                }
                // lane (batch row l15 of row tile (h, i), group g) holds P[v = 4 g + e][row]
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (4 * g + e < V) red[(wn * BM + row_of(h, i) + l15) * V + 4 * g + e] = pa[e] * out_inv3;
            }
        __syncthreads();
        float *dst = parts + ((size_t)tn * M + (size_t)tm * BM) * V;
        for (int i = tid; i < BM * V; i += 512)
            dst[i] = (red[i] + red[BM * V + i]) + (red[2 * BM * V + i] + red[3 * BM * V + i]);
        DNAG_TILE_REPORT();
        return;
    } else {
        // epilogue: scale back, bias, activation, k-tile-blocked split-pair rows, staged through LDS.  In the blocked layout
        // the 32 columns x 256 rows of one output k tile are 32 KB of contiguous memory, and a wave's 32 columns of a B
        // half are exactly one k tile: per B half the block writes its 128 KB image [k tile wn][row][128 B] into LDS (two
        // 2-byte writes per value; 16-byte slot s of row r at slot s ^ ((r >> 2) & 7), which spreads the four row groups of
        // a write over the banks) and copies it out with 16-byte loads and fully coalesced 1 KB stores (dword stores from
        // the accumulator layout kept the address path busy for 15 us per tile).
        // Lane (column l15 of a column tile, group g) holds rows 4 g + e of a row tile.
        _Float16 *img = reinterpret_cast<_Float16 *>(smem);
        const float *imgf = smem;
        const int ktg0 = (tn * BN) / 32;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int k = j * 16 + l15;                                  // column inside the wave's k tile
                const float bv = bias ? bias[tn * BN + col_of(b, j) + l15] : 0.f;
                const int slot = (k >> 3) * 2, within = k & 7;               // hi slot; the lo slot is the next one
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int row = row_of(h, i) + 4 * g + e;
                            float v = fmaf(acc[h][b][i][j][e], out_inv, bv);
                            if (ACT == ACT_ELU) v = v > 0.f ? v : __expf(v) - 1.0f;
                            _Float16 hi, lo;
                            split16(v, sa_out, hi, lo);
                            const int sw = (row >> 2) & 7;
                            _Float16 *rp = img + (size_t)(wn * BM + row) * 64;
                            rp[((slot ^ sw) << 3) + within] = hi;
                            rp[(((slot + 1) ^ sw) << 3) + within] = lo;
                        }
            }
            __syncthreads();
#ifdef DNAG_GEMM_TILE_STAMPS
            if (b == 0 && tid == 0 && blockIdx.x < 4) printf("  epi blk %d: values+lds of half 0 done at %lld\n", (int)blockIdx.x, wall_clock64() - tw2);
#endif
            // copy out: chunk c = 8 rows x 128 B of k tile c / 32; lane = (row of the chunk, 16-byte slot)
#pragma unroll 4
            for (int q = 0; q < 16; ++q) {
                const int c = q * 8 + wave, kt2 = c >> 5, row = (c & 31) * 8 + (lane >> 3), sl = lane & 7;
                const v4f val = *reinterpret_cast<const v4f *>(imgf + (size_t)(kt2 * BM + row) * 32 + ((sl ^ ((row >> 2) & 7)) << 2));
                *reinterpret_cast<v4f *>(C + ((size_t)(ktg0 + b * 4 + kt2) * M + (size_t)tm * BM + row) * 32 + sl * 4) = val;
            }
            __syncthreads();
#ifdef DNAG_GEMM_TILE_STAMPS
            if (b == 0 && tid == 0 && blockIdx.x < 4) printf("  epi blk %d: copy of half 0 done at %lld\n", (int)blockIdx.x, wall_clock64() - tw2);
#endif
        }
        DNAG_TILE_REPORT();
    }
}

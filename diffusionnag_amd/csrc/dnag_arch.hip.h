// Device-side exit of a sampling round for NAS-Bench-201: threshold, validity rules, architecture code and
// first-occurrence table, so that the round loop of generate_archs_meta (main_exp/diffusion/run_lib.py:131-152)
// needs one small readback per round instead of a per-sample .cpu() loop.  Included by dnag_lib.hip.
//
// Restates (reference NAS-Bench-201/): utils.py:244-262 (quantize: x >= 0.5), analysis/arch_functions.py:284-329
// (check_single_node_type, check_start_end_nodes, check_interm_node_types, is_valid_NAS201_x and its error
// codes), :209-216 (argmax decode of rows 1..6).
#pragma once
#include "dnag_common.h"

#define NB201_N 8
#define NB201_V 7
#define NB201_NCODES 15625            // 5^6: six edges, five operations (types 2..6)
#define NB201_UNSEEN 0x7fffffff

// One thread per graph.  status: -1 valid, 1 MULTIPLE_NODE_TYPES, 2 No_START_END, 3 INTERM_START_END
// (arch_functions.py:308-313).  code = sum_i (argmax(row i+1) - 2) 5^i for a valid graph, else -1.
// first_seen [NB201_NCODES] (may be null): min over graphs of (index_base + graph index) per code.
__global__ __launch_bounds__(256) void nb201_validity_kernel(const float *__restrict__ x, int batch, int32_t *__restrict__ status,
                                                             int32_t *__restrict__ code, int32_t *__restrict__ first_seen,
                                                             int index_base) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= batch) return;
    const float *xb = x + (size_t)b * NB201_N * NB201_V;
    int type[NB201_N];
    bool single = true;
    uint32_t has_start = 0, has_end = 0;      // bit n: row n has a 1 in the START (0) / END (1) column
#pragma unroll
    for (int n = 0; n < NB201_N; ++n) {
        int cnt = 0, arg = 0;
#pragma unroll
        for (int v = 0; v < NB201_V; ++v) {
            const bool one = xb[n * NB201_V + v] >= 0.5f;              // utils.py:254-255
            if (one && cnt == 0) arg = v;                              // torch.argmax: first maximal entry
            cnt += one ? 1 : 0;
            if (one && v == 0) has_start |= 1u << n;
            if (one && v == 1) has_end |= 1u << n;
        }
        if (cnt != 1) single = false;                                  // :284-288
        type[n] = arg;
    }
    int st = -1, cd = -1;
    if (!single) st = 1;
    else if (!(has_start & 1u) || !(has_end & (1u << (NB201_N - 1)))) st = 2;               // :291-296
    else if (((has_start | has_end) >> 1) & ((1u << (NB201_N - 2)) - 1)) st = 3;             // :299-305
    else {
        cd = 0;
        int mul = 1;
#pragma unroll
        for (int i = 1; i <= 6; ++i) { cd += (type[i] - 2) * mul; mul *= 5; }
        if (first_seen) atomicMin(&first_seen[cd], index_base + b);
    }
    status[b] = st;
    code[b] = cd;
}

// counts[0] = graphs with status -1 among `batch`; counts[1] = codes seen so far.  One block.
__global__ __launch_bounds__(1024) void nb201_count_kernel(const int32_t *__restrict__ status, int batch,
                                                           const int32_t *__restrict__ first_seen, int32_t *__restrict__ counts) {
    __shared__ int sv[1024], su[1024];
    int nv = 0, nu = 0;
    for (int i = threadIdx.x; i < batch; i += 1024) nv += status[i] == -1;
    if (first_seen)
        for (int i = threadIdx.x; i < NB201_NCODES; i += 1024) nu += first_seen[i] != NB201_UNSEEN;
    sv[threadIdx.x] = nv; su[threadIdx.x] = nu;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { sv[threadIdx.x] += sv[threadIdx.x + s]; su[threadIdx.x] += su[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { counts[0] = sv[0]; counts[1] = su[0]; }
}

__global__ void fill_i32_kernel(int32_t *p, size_t n, int32_t v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

// ---------------------------------------------------------------------------------------------
// MobileNetV3 / OFA search space: 5 stages x 4 layer slots = 20 nodes, 9 (kernel, expand) types; a row without a 1 is a
// skipped layer.  Restates (reference MobileNetV3/): utils.py:244-262 (quantize: x >= 0.5),
// analysis/arch_functions.py:374-380 (check_multiple_nodes), :386-392 (layers 1 and 2 of every stage must exist),
// :394-402 (layer 4 needs layer 3), :413-429 (is_valid_OFA_x and its error codes).
// One thread per graph.  status: -1 valid, 1 MULIPLE_NODES, 2 INVALID_1OR2_LAYERS, 3 INVALID_3AND4_LAYERS.
// types [batch][20]: the row's type 0..8 (first 1 of the row), -1 for an empty row -- what
// get_string_from_onehot_x (:135-144) needs; written for every graph.
// ---------------------------------------------------------------------------------------------
#define OFA_N 20
#define OFA_V 9
__global__ __launch_bounds__(256) void ofa_validity_kernel(const float *__restrict__ x, int batch, int32_t *__restrict__ status,
                                                           int8_t *__restrict__ types) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= batch) return;
    const float *xb = x + (size_t)b * OFA_N * OFA_V;
    uint32_t empty = 0, multi = 0;            // bit n: row n has no 1 / more than one 1
    for (int n = 0; n < OFA_N; ++n) {
        int cnt = 0, arg = -1;
#pragma unroll
        for (int v = 0; v < OFA_V; ++v) {
            const bool one = xb[n * OFA_V + v] >= 0.5f;
            if (one && cnt == 0) arg = v;
            cnt += one ? 1 : 0;
        }
        if (cnt == 0) empty |= 1u << n;
        if (cnt > 1) multi |= 1u << n;
        types[(size_t)b * OFA_N + n] = (int8_t)arg;
    }
    int st = -1;
    if (multi) st = 1;
    else {
        const uint32_t first_second = 0x33333u;       // rows 0,1, 4,5, 8,9, 12,13, 16,17
        const uint32_t third = 0x44444u;              // rows 2, 6, 10, 14, 18
        if (empty & first_second) st = 2;
        else if (((empty & third) << 1) & ~empty) st = 3;     // third empty, fourth not
    }
    status[b] = st;
}

// counts[0..3] = graphs with status -1, 1, 2, 3.  One block.
__global__ __launch_bounds__(1024) void ofa_count_kernel(const int32_t *__restrict__ status, int batch, int32_t *__restrict__ counts) {
    __shared__ int sv[4][1024];
    int c[4] = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < batch; i += 1024) {
        const int s = status[i];
        c[s < 0 ? 0 : s] += 1;
    }
    for (int k = 0; k < 4; ++k) sv[k][threadIdx.x] = c[k];
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
            for (int k = 0; k < 4; ++k) sv[k][threadIdx.x] += sv[k][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x < 4) counts[threadIdx.x] = sv[threadIdx.x][0];
}

// Internal launcher of the fp32 MFMA GEMM (gemm.hip), shared with topk.hip.
#pragma once
#include "common.hpp"

struct MiGemmArgs {
    int64_t M, N, K;
    const float* A;       // element (m, k) at A[row(m) * sa_m + k * sa_k]
    int64_t sa_m, sa_k;
    const int64_t* a_rows;  // optional row gather for A: row(m) = a_rows[m]; null = identity
    const float* B;       // element (k, n) at B[n * sb_n + k * sb_k]
    int64_t sb_n, sb_k;
    const float* bias;    // optional [N]
    float* C;             // [M, N] row-major, leading dimension ldc
    int64_t ldc;
    int accumulate;       // C += ...
    int act;              // 0 none, 1 relu
    // split-K (set by the launcher): this launch covers K in `splits` slices of `k_per_split`;
    // slice z writes its raw accumulator to partial[z, M, N] and a second kernel reduces in slice order
    int splits;
    int64_t k_per_split;
    float* partial;
};

// Number of K slices the launcher will use for this shape (1 = no split) and the workspace it needs.
int mi_gemm_splits(int64_t M, int64_t N, int64_t K);
int mi_gemm_launch(MiGemmArgs g, void* ws, size_t ws_bytes, hipStream_t stream);

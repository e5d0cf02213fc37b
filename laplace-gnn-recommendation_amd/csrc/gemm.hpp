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
};

int mi_gemm_launch(const MiGemmArgs& g, hipStream_t stream);

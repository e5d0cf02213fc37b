/*
 * ORACLE — test infrastructure only.  Nothing in the product imports, links or executes this;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may.
 *
 * CPU restatement of the sparse aggregate the reference reaches through torch_sparse
 * (not vendored under /root/reference; environment.yml:28-30 leaves it unpinned, ~0.6.13):
 *   torch_sparse.matmul(adj_t, x)  called at  model/lightgcn.py:87
 * whose CPU kernel (csrc/cpu/spmm_cpu.cpp, `spmm_cpu`, reduce = sum) walks rows in parallel
 * (at::parallel_for) and, per row, accumulates val[e] * mat[col[e], :] over e in CSR order into
 * a per-row vector of K floats, one fused-or-not multiply-add per element, fp32.
 * The loop below is that algorithm; OpenMP over rows stands in for at::parallel_for.
 *
 * Also: the exact score definition the top-K path is checked against
 *   (utils/metrics_lightgcn.py:137  scores = user_embeddings[user_id] @ article_embeddings.T)
 * as a k-ordered fmaf chain, which is what the gfx950 f32 MFMA computes bit for bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int ref_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* Y[r,:] = sum_e val[e] * X[col[e],:], e in [rowptr[r], rowptr[r+1]) in order. */
void ref_spmm_csr_f32(int64_t n_rows, int64_t d, const int32_t* rowptr, const int32_t* col,
                      const float* val, const float* X, int64_t ldx, float* Y, int64_t ldy) {
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t r = 0; r < n_rows; ++r) {
        float* y = Y + r * ldy;
        for (int64_t k = 0; k < d; ++k) y[k] = 0.0f;
        for (int32_t e = rowptr[r]; e < rowptr[r + 1]; ++e) {
            const float v = val[e];
            const float* x = X + (int64_t)col[e] * ldx;
            for (int64_t k = 0; k < d; ++k) y[k] += v * x[k];
        }
    }
}

/* Same in double accumulation, for error budgeting of the fp32 paths. */
void ref_spmm_csr_f64acc(int64_t n_rows, int64_t d, const int32_t* rowptr, const int32_t* col,
                         const float* val, const float* X, int64_t ldx, double* Y, int64_t ldy) {
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t r = 0; r < n_rows; ++r) {
        double* y = Y + r * ldy;
        for (int64_t k = 0; k < d; ++k) y[k] = 0.0;
        for (int32_t e = rowptr[r]; e < rowptr[r + 1]; ++e) {
            const double v = val[e];
            const float* x = X + (int64_t)col[e] * ldx;
            for (int64_t k = 0; k < d; ++k) y[k] += v * (double)x[k];
        }
    }
}

/* scores[q, i] = fma chain over k ascending of U[q,k] * I[i,k], starting from 0. */
void ref_scores_fma_f32(int64_t n_q, int64_t n_items, int64_t d, const float* U, int64_t ldu,
                        const float* I, int64_t ldi, float* out, int64_t ldo) {
#pragma omp parallel for schedule(static)
    for (int64_t q = 0; q < n_q; ++q) {
        const float* u = U + q * ldu;
        for (int64_t i = 0; i < n_items; ++i) {
            const float* it = I + i * ldi;
            float acc = 0.0f;
            for (int64_t k = 0; k < d; ++k) acc = fmaf(u[k], it[k], acc);
            out[q * ldo + i] = acc;
        }
    }
}

/* Dense Adam exactly as torch.optim.Adam (single-tensor path, no amsgrad / weight decay):
 *   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g ; p -= step_size * m / (sqrt(v)/sqrt(bc2) + eps) */
void ref_adam_f32(int64_t n, float* p, const float* g, float* m, float* v, float b1, float b2,
                  float step_size, float bc2_sqrt, float eps) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        float mm = b1 * m[i] + (1.0f - b1) * g[i];
        float vv = b2 * v[i] + (1.0f - b2) * g[i] * g[i];
        p[i] = p[i] - step_size * (mm / (sqrtf(vv) / bc2_sqrt + eps));
        m[i] = mm;
        v[i] = vv;
    }
}

/* General GEMM with the MFMA's arithmetic: per output element a k-ascending fmaf chain from 0,
 * then + bias, + C (accumulate), relu.  Element (m,k) of A at A[m*sa_m + k*sa_k]; (k,n) of B at
 * B[n*sb_n + k*sb_k].  Restates torch.nn.Linear / its backward products (model/layers.py:35-56)
 * in the summation order csrc/gemm.hip uses, so the comparison is bitwise. */
void ref_gemm_fma_f32(int64_t M, int64_t N, int64_t K, const float* A, int64_t sa_m, int64_t sa_k,
                      const float* B, int64_t sb_n, int64_t sb_k, const float* bias, float* C,
                      int64_t ldc, int accumulate, int act) {
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; ++m) {
        for (int64_t n = 0; n < N; ++n) {
            float acc = 0.0f;
            for (int64_t k = 0; k < K; ++k) acc = fmaf(A[m * sa_m + k * sa_k], B[n * sb_n + k * sb_k], acc);
            if (bias) acc += bias[n];
            if (accumulate) acc += C[m * ldc + n];
            if (act == 1) acc = acc > 0.0f ? acc : 0.0f;
            C[m * ldc + n] = acc;
        }
    }
}

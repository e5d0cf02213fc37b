// Internal twin launchers (round 4): TWO independent small launches of the same kernel — the customer-side and the
// article-side twin of a ranker iteration's step — as ONE launch whose workgroups are dealt to the two problems by block
// index.  Same device code per element as the single launches (the bodies are shared), so results are bitwise those of two
// separate launches; what is saved is a launch (~5 us of a ~0.5 ms iteration each) and the tail of the smaller twin.
// Not part of the C ABI: called by the native executors only (csrc/ranker_exec.hip).  Every function returns
// MI_ERR_UNSUPPORTED — nothing enqueued — when the pair does not share a kernel instantiation; the caller then issues the two
// single launches.
#pragma once
#include "common.hpp"

namespace mi_pairs {

struct EmbedSide {           // one mi_embed_concat_f32 call
    int64_t n; int32_t n_cols;
    const int64_t* x; const float* const* tables; const int64_t* table_rows; const int32_t* dims;
    float* out; int64_t ldo;
};
int embed_concat_pair(const EmbedSide& a, const EmbedSide& b, float max_norm, hipStream_t s);

struct SpmmSide {            // one plan-less dense mi_spmm_csr_ex_f32 call: acc = A X; Y = acc (nullable); S = addend + acc (nullable)
    int64_t n_rows, d;
    const int32_t* rowptr; const int32_t* col; const float* val;
    const float* X; float* Y; const float* addend; float* S;     // all leading dimensions = d
};
int spmm_planless_pair(const SpmmSide& a, const SpmmSide& b, hipStream_t s);

struct BnSide {              // one training-mode mi_batchnorm_fwd_f32 / mi_batchnorm_bwd_f32 call (leading dimensions = c)
    int64_t n;
    const float* X;                      // the layer's input
    const float *gamma, *beta;           // nullable
    float *running_mean, *running_var;   // nullable (forward)
    float momentum, eps;
    float *save_mean, *save_invstd;      // written by the forward, read by the backward
    float* Y;                            // forward output
    const float* dY; float* dX; float *dgamma, *dbeta;   // backward
    void* ws;                            // mi_batchnorm_workspace_bytes(c)
};
int batchnorm_fwd_pair(const BnSide& a, const BnSide& b, int64_t c, hipStream_t s);
int batchnorm_bwd_pair(const BnSide& a, const BnSide& b, int64_t c, hipStream_t s);

// both halves of the decoder's gather-cat backward: dZ0[v] = sum of dOut[e, 0:c] over idx0[e] == v, dZ1[v] = sum of
// dOut[e, c:2c] over idx1[e] == v, in edge order (mi_gather_cat_bwd_f32 twice).  dZ0 / dZ1 must have been zero-filled.
int gather_cat_bwd_pair(int64_t n_edges, int64_t c, const int64_t* idx0, const int64_t* idx1, const float* dOut, int64_t ldo,
                        float* dZ0, float* dZ1, hipStream_t s);

// mi_gather_cat_f32 with the decoder's Philox feature dropout applied on the way out (out = dropout(cat(Zu[row], Zi[col]))): one
// launch instead of gather + dropout.  A lane owns one float4 of the concatenated row and draws ONE Philox block for it —
// the counter (flat float4 index of the [n_edges, cu + ci] matrix, site, step) is exactly dropout_kernel's, so the backward's
// regenerated mask matches.  cu, ci multiples of 4, 16-byte aligned rows; else MI_ERR_UNSUPPORTED.
int gather_cat_dropout(int64_t n_edges, int64_t cu, int64_t ci, const int64_t* row, const int64_t* col, const float* Zu, const float* Zi,
                       float* out, float p, uint64_t seed, uint32_t site, uint32_t step_lo, hipStream_t s);

// mi_linear1_bwd_f32 as ONE launch: the band partials are reduced by the workgroup that finishes last (agent-scope ticket in
// `counter`, which must be zero at entry and is zero again at exit), in band order — the same sums as the two-launch form.
int linear1_bwd_lastblock(int64_t n, int64_t in, const float* dy, const float* w, const float* x, float* dx, float* gw, float* gb,
                          void* ws, size_t ws_bytes, int32_t* counter, hipStream_t s);

}  // namespace mi_pairs

// Ranker-side kernels that are not the SpMM or the GEMM:
//   K5 (max)  mi_segment_max_f32 / mi_segment_max_bwd_f32 — SAGEConv(aggr="max") message passing
//             (model/layers.py:11-24 -> torch_scatter.scatter(..., reduce="max")); aggr="add"/"mean"
//             run on mi_spmm_csr_f32 with unit / 1/in-degree weights (mi_scale_csr_f32).
//   K7        mi_embed_concat_f32 — Encoder_Decoder_Model.__embedding (model/encoder_decoder.py:116-125):
//             per integer column an Embedding(max_norm=1) lookup, concatenated along dim 1.
#include "common.hpp"
#include "pairs.hpp"
#include <algorithm>

namespace {

constexpr int kBlock = 256;

// One wavefront per destination row; lanes stride over the feature columns.
__global__ __launch_bounds__(kBlock) void segment_max_kernel(int64_t n_dst, int d,
                                                             const int32_t* __restrict__ rowptr,
                                                             const int32_t* __restrict__ col,
                                                             const float* __restrict__ X, int64_t ldx,
                                                             float* __restrict__ Y, int64_t ldy,
                                                             int32_t* __restrict__ arg) {
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / MI_WAVE;
    if (r >= n_dst) return;
    const int lane = mi_lane();
    const int32_t b = rowptr[r], e = rowptr[r + 1];
    for (int c = lane; c < d; c += MI_WAVE) {
        float best = 0.f;  // empty segment -> 0 (torch_scatter fills untouched outputs with 0)
        int32_t who = -1;
        for (int32_t p = b; p < e; ++p) {
            const int32_t s = col[p];
            const float v = X[(int64_t)s * ldx + c];
            if (who < 0 || v > best) {  // first maximum wins, in CSR (sorted source id) order
                best = v;
                who = s;
            }
        }
        Y[r * ldy + c] = best;
        if (arg) arg[r * (int64_t)d + c] = who;
    }
}

// Backward of the max: dX[s, c] = sum of dY[r, c] over the destinations r of source s whose arg-max at column c is s.
// One wavefront per SOURCE row walks that source's destinations in the by-source CSR (sorted, so the sum has a fixed
// order and a duplicated edge is seen once): one writer per element, no float atomics — bitwise reproducible.
__global__ __launch_bounds__(kBlock) void segment_max_bwd_kernel(int64_t n_src, int d,
                                                                 const int32_t* __restrict__ src_rowptr,
                                                                 const int32_t* __restrict__ src_col,
                                                                 const int32_t* __restrict__ arg,
                                                                 const float* __restrict__ dY, int64_t ldy,
                                                                 float* __restrict__ dX, int64_t ldx) {
    const int64_t s = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / MI_WAVE;
    if (s >= n_src) return;
    const int lane = mi_lane();
    const int32_t b = src_rowptr[s], e = src_rowptr[s + 1];
    for (int c = lane; c < d; c += MI_WAVE) {
        float g = 0.f;
        int32_t prev = -1;
        for (int32_t p = b; p < e; ++p) {
            const int32_t r = src_col[p];
            if (r == prev) continue;  // the same edge listed twice won the max once
            prev = r;
            if (arg[(int64_t)r * d + c] == (int32_t)s) g += dY[(int64_t)r * ldy + c];
        }
        dX[s * ldx + c] = g;
    }
}

struct EmbedCols {
    const float* table[16];
    int32_t dim[16];
    int32_t off[16];
    int64_t rows[16];
};

// One SG-lane sub-group per (node, column): lanes cover the column's embedding width; the row's L2 norm is reduced across
// the sub-group and the max_norm scale applied on the fly.  SG = 16 when no column is wider than 64 floats (the H&M
// tables: 4 (node, column) pairs per wavefront instead of one — the article side of a ranker batch went 35.6 -> see
// profiles/r03_ranker_native.md), else a whole wavefront.
struct EmbedCol { const float* table; int32_t dim, off; int64_t rows; };
template <int SG, class ColOf>
__device__ __forceinline__ void embed_concat_body(unsigned bid, int64_t n, int n_cols,
                                                  const int64_t* __restrict__ x, ColOf col_of,
                                                  float max_norm, float* __restrict__ out,
                                                  int64_t ldo) {
    const int64_t w = ((int64_t)bid * blockDim.x + threadIdx.x) / SG;
    const bool live = w < n * n_cols;          // no early return: the shuffles below want every lane of the wavefront
    const int64_t node = live ? w / n_cols : 0;
    const int c = live ? (int)(w - node * n_cols) : 0;
    const int lane = (int)(threadIdx.x % SG);
    const EmbedCol ec = col_of(c);
    int64_t id = x[node * n_cols + c];
    if (id < 0) id = 0;
    if (id >= ec.rows) id = ec.rows - 1;
    const int dim = ec.dim;
    const float* src = ec.table + id * dim;
    float v[4];   // SG = 16: dim <= 64 -> at most four elements per lane, kept for the write
    float ss = 0.f;
    if (SG == 16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = lane + 16 * j;
            v[j] = k < dim ? src[k] : 0.f;
            ss = fmaf(v[j], v[j], ss);
        }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 16);
    } else {
        for (int k = lane; k < dim; k += MI_WAVE) ss = fmaf(src[k], src[k], ss);
        ss = mi_wave_sum(ss);
    }
    if (!live) return;
    const float norm = sqrtf(ss);
    const float scale = (max_norm > 0.f && norm > max_norm) ? max_norm / (norm + 1e-7f) : 1.0f;
    float* dst = out + node * ldo + ec.off;
    if (SG == 16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = lane + 16 * j;
            if (k < dim) dst[k] = v[j] * scale;
        }
    } else {
        for (int k = lane; k < dim; k += MI_WAVE) dst[k] = src[k] * scale;
    }
}
template <int SG>
__global__ __launch_bounds__(kBlock) void embed_concat_kernel(int64_t n, int n_cols,
                                                              const int64_t* __restrict__ x, EmbedCols ec,
                                                              float max_norm, float* __restrict__ out,
                                                              int64_t ldo) {
    embed_concat_body<SG>(blockIdx.x, n, n_cols, x, [&](int c) { return EmbedCol{ec.table[c], ec.dim[c], ec.off[c], ec.rows[c]}; },
                          max_norm, out, ldo);
}
// twin launch (mi_pairs): the first `split` workgroups look up node type 0, the rest node type 1.  ONE argument struct with
// both sides' columns (entry 16 w + c): a reference or a per-entry choice between two by-value structs sends both to scratch
// (776 bytes per lane: 293 us instead of 17 for the pair — measured, round 4)
struct EmbedOne { int64_t n; int n_cols; const int64_t* x; float* out; int64_t ldo; };
struct EmbedCols2 {
    const float* table[32];
    int32_t dim[32];
    int32_t off[32];
    int64_t rows[32];
};
template <int SG>
__global__ __launch_bounds__(kBlock) void embed_concat_pair_kernel(EmbedOne a, EmbedOne b, EmbedCols2 ec, float max_norm, unsigned split) {
    const bool first = blockIdx.x < split;
    const int base = first ? 0 : 16;
    embed_concat_body<SG>(first ? blockIdx.x : blockIdx.x - split, first ? a.n : b.n, first ? a.n_cols : b.n_cols, first ? a.x : b.x,
                          [&](int c) { return EmbedCol{ec.table[base + c], ec.dim[base + c], ec.off[base + c], ec.rows[base + c]}; },
                          max_norm, first ? a.out : b.out, first ? a.ldo : b.ldo);
}

bool embed_cols(const mi_pairs::EmbedSide& d, EmbedCols& ec, int& widest) {
    if (d.n <= 0 || d.n_cols <= 0 || d.n_cols > 16 || !d.x || !d.tables || !d.table_rows || !d.dims || !d.out) return false;
    int32_t off = 0;
    for (int c = 0; c < d.n_cols; ++c) {
        if (!d.tables[c] || d.dims[c] <= 0 || d.table_rows[c] <= 0) return false;
        ec.table[c] = d.tables[c];
        ec.dim[c] = d.dims[c];
        ec.rows[c] = d.table_rows[c];
        ec.off[c] = off;
        off += d.dims[c];
        widest = std::max(widest, (int)d.dims[c]);
    }
    return d.ldo >= off;
}

}  // namespace

namespace mi_pairs {
int embed_concat_pair(const EmbedSide& a, const EmbedSide& b, float max_norm, hipStream_t s) {
    EmbedCols eca, ecb;
    int widest = 0;
    if (!embed_cols(a, eca, widest) || !embed_cols(b, ecb, widest)) return MI_ERR_UNSUPPORTED;
    const EmbedOne qa{a.n, (int)a.n_cols, a.x, a.out, a.ldo}, qb{b.n, (int)b.n_cols, b.x, b.out, b.ldo};
    EmbedCols2 ec;
    memset(&ec, 0, sizeof(ec));
    for (int w = 0; w < 2; ++w) {
        const EmbedCols& src = w ? ecb : eca;
        for (int c = 0; c < (w ? b.n_cols : a.n_cols); ++c) {
            ec.table[16 * w + c] = src.table[c]; ec.dim[16 * w + c] = src.dim[c]; ec.off[16 * w + c] = src.off[c]; ec.rows[16 * w + c] = src.rows[c];
        }
    }
    if (widest <= 64) {   // both sides take the 16-lane instantiation, as their single launches would
        const unsigned ga = (unsigned)mi_ceil_div(a.n * a.n_cols * 16, kBlock), gb = (unsigned)mi_ceil_div(b.n * b.n_cols * 16, kBlock);
        hipLaunchKernelGGL(embed_concat_pair_kernel<16>, dim3(ga + gb), dim3(kBlock), 0, s, qa, qb, ec, max_norm, ga);
    } else {
        int wa = 0, wb = 0;
        for (int c = 0; c < a.n_cols; ++c) wa = std::max(wa, (int)a.dims[c]);
        for (int c = 0; c < b.n_cols; ++c) wb = std::max(wb, (int)b.dims[c]);
        if (wa <= 64 || wb <= 64) return MI_ERR_UNSUPPORTED;   // the single launches would pick different instantiations
        const unsigned ga = (unsigned)mi_ceil_div(a.n * a.n_cols * MI_WAVE, kBlock), gb = (unsigned)mi_ceil_div(b.n * b.n_cols * MI_WAVE, kBlock);
        hipLaunchKernelGGL(embed_concat_pair_kernel<MI_WAVE>, dim3(ga + gb), dim3(kBlock), 0, s, qa, qb, ec, max_norm, ga);
    }
    return mi_launch_status();
}
}  // namespace mi_pairs

extern "C" {

int mi_segment_max_f32(int64_t n_dst, int64_t d, const int32_t* rowptr, const int32_t* col,
                       const float* X, int64_t ldx, float* Y, int64_t ldy, int32_t* arg,
                       mi_stream_t stream) {
    MI_CHECK_ARG(n_dst >= 0 && d > 0);
    if (n_dst == 0) return 0;
    MI_CHECK_ARG(rowptr && Y && ldy >= d && (X == nullptr || ldx >= d));
    dim3 g((unsigned)mi_ceil_div(n_dst * MI_WAVE, kBlock));
    hipLaunchKernelGGL(segment_max_kernel, g, dim3(kBlock), 0, (hipStream_t)stream, n_dst, (int)d, rowptr, col, X, ldx,
                       Y, ldy, arg);
    return mi_launch_status();
}

int mi_segment_max_bwd_f32(int64_t n_src, int64_t d, const int32_t* src_rowptr, const int32_t* src_col,
                           const int32_t* arg, const float* dY, int64_t ldy, float* dX, int64_t ldx,
                           mi_stream_t stream) {
    MI_CHECK_ARG(n_src >= 0 && d > 0);
    if (n_src == 0) return 0;
    MI_CHECK_ARG(src_rowptr && dX && ldx >= d && (dY == nullptr || ldy >= d));
    dim3 g((unsigned)mi_ceil_div(n_src * MI_WAVE, kBlock));
    hipLaunchKernelGGL(segment_max_bwd_kernel, g, dim3(kBlock), 0, (hipStream_t)stream, n_src, (int)d, src_rowptr,
                       src_col, arg, dY, ldy, dX, ldx);
    return mi_launch_status();
}

int mi_embed_concat_f32(int64_t n, int32_t n_cols, const int64_t* x, const float* const* tables,
                        const int64_t* table_rows, const int32_t* dims, float max_norm, float* out,
                        int64_t ldo, mi_stream_t stream) {
    MI_CHECK_ARG(n >= 0 && n_cols >= 0);
    if (n == 0 || n_cols == 0) return 0;
    if (n_cols > 16) return MI_ERR_UNSUPPORTED;
    MI_CHECK_ARG(x && tables && table_rows && dims && out);
    EmbedCols ec;
    int32_t off = 0;
    for (int c = 0; c < n_cols; ++c) {  // tables/table_rows/dims are HOST arrays (a handful of entries)
        MI_CHECK_ARG(tables[c] && dims[c] > 0 && table_rows[c] > 0);
        ec.table[c] = tables[c];
        ec.dim[c] = dims[c];
        ec.rows[c] = table_rows[c];
        ec.off[c] = off;
        off += dims[c];
    }
    MI_CHECK_ARG(ldo >= off);
    int widest = 0;
    for (int c = 0; c < n_cols; ++c) widest = std::max(widest, (int)dims[c]);
    if (widest <= 64) {
        dim3 g((unsigned)mi_ceil_div(n * n_cols * 16, kBlock));
        hipLaunchKernelGGL(embed_concat_kernel<16>, g, dim3(kBlock), 0, (hipStream_t)stream, n, (int)n_cols, x, ec, max_norm, out, ldo);
    } else {
        dim3 g((unsigned)mi_ceil_div(n * n_cols * MI_WAVE, kBlock));
        hipLaunchKernelGGL(embed_concat_kernel<MI_WAVE>, g, dim3(kBlock), 0, (hipStream_t)stream, n, (int)n_cols, x, ec, max_norm, out,
                           ldo);
    }
    return mi_launch_status();
}

}  // extern "C"

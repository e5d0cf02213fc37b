// K3 / K4: adjacency construction on device — COO -> sorted CSR, CSR transpose,
// gcn_norm, row expansion.  One-off per adjacency (cached by the host layer), so the
// device-wide sort and scan come from rocPRIM; everything else is plain HIP.
//
// Reference semantics restated here:
//   torch_sparse.SparseTensor(row, col, sparse_sizes)  data/lightgcn_loader.py:65-79
//   torch_geometric gcn_norm(adj_t, add_self_loops=False)  model/lightgcn.py:56
#include "common.hpp"
#include <rocprim/rocprim.hpp>

namespace {

constexpr int kBlock = 256;

// Sort keys are PACKED: (row << cb) | col with cb = bits of the column range, so the radix sort walks
// bits(rows) + bits(cols) key bits instead of 32 + bits(rows) — a per-batch subgraph of the ranker (2^15 x 2^15)
// sorts in 4 digit passes instead of 6 (round 2: the two sorts per batch were ~100 us of a 1.3 ms iteration).
__global__ void make_keys_coo(int64_t nnz, const int64_t* __restrict__ row,
                              const int64_t* __restrict__ col, unsigned cb, uint64_t* __restrict__ keys,
                              int32_t* __restrict__ vals) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nnz) return;
    keys[i] = ((uint64_t)(uint32_t)row[i] << cb) | (uint64_t)(uint32_t)col[i];
    vals[i] = (int32_t)i;
}

// first position p in [0, n) with (keys[p] >> cb) >= r
__device__ __forceinline__ int32_t lower_bound_hi(const uint64_t* __restrict__ keys, int64_t n, unsigned cb,
                                                  uint32_t r) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if ((uint32_t)(keys[mid] >> cb) < r) lo = mid + 1; else hi = mid;
    }
    return (int32_t)lo;
}

__global__ void rowptr_from_keys(int64_t n_rows, int64_t nnz, unsigned cb, const uint64_t* __restrict__ keys,
                                 int32_t* __restrict__ rowptr) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_rows) return;
    rowptr[r] = (r == n_rows) ? (int32_t)nnz : lower_bound_hi(keys, nnz, cb, (uint32_t)r);
}

__global__ void decode_keys(int64_t nnz, unsigned cb, const uint64_t* __restrict__ keys,
                            const int32_t* __restrict__ vals, int32_t* __restrict__ col_out,
                            int32_t* __restrict__ perm) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nnz) return;
    col_out[i] = (int32_t)(uint32_t)(keys[i] & (((uint64_t)1 << cb) - 1));
    if (perm) perm[i] = vals[i];
}

// row of entry p: last r with rowptr[r] <= p
__device__ __forceinline__ int32_t row_of(const int32_t* __restrict__ rowptr, int64_t n_rows,
                                          int32_t p) {
    int64_t lo = 0, hi = n_rows;  // invariant: rowptr[lo] <= p < rowptr[hi]
    while (hi - lo > 1) {
        int64_t mid = (lo + hi) >> 1;
        if (rowptr[mid] <= p) lo = mid; else hi = mid;
    }
    return (int32_t)lo;
}

__global__ void make_keys_transpose(int64_t n_rows, int64_t nnz, const int32_t* __restrict__ rowptr,
                                    const int32_t* __restrict__ col, unsigned cb, uint64_t* __restrict__ keys,
                                    int32_t* __restrict__ vals) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nnz) return;
    uint32_t r = (uint32_t)row_of(rowptr, n_rows, (int32_t)i);
    keys[i] = ((uint64_t)(uint32_t)col[i] << cb) | (uint64_t)r;
    vals[i] = (int32_t)i;
}

__global__ void expand_rows_kernel(int64_t n_rows, int64_t nnz, const int32_t* __restrict__ rowptr,
                                   int32_t* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nnz) return;
    out[i] = row_of(rowptr, n_rows, (int32_t)i);
}

__global__ void gather_f32_kernel(int64_t n, const float* __restrict__ src,
                                  const int32_t* __restrict__ idx, float* __restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = src[idx[i]];
}

// deg^-1/2 with inf -> 0.  One wave per row; with unit values the degree is the count.
__global__ void gcn_dis_kernel(int64_t n, const int32_t* __restrict__ rowptr,
                               const float* __restrict__ val, float* __restrict__ dis) {
    int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / MI_WAVE;
    if (r >= n) return;
    int lane = mi_lane();
    int32_t b = rowptr[r], e = rowptr[r + 1];
    float deg;
    if (val == nullptr) {
        deg = (float)(e - b);
    } else {
        float s = 0.f;
        for (int32_t p = b + lane; p < e; p += MI_WAVE) s += val[p];
        deg = mi_wave_sum(s);
    }
    if (lane == 0) dis[r] = (deg > 0.f) ? (1.0f / sqrtf(deg)) : 0.f;
}

// val_out[p] = (val_in[p] * row_scale[row(p)]) * col_scale[col[p]]   (null pointer = all ones)
__global__ void gcn_scale_kernel(int64_t n, int64_t nnz, const int32_t* __restrict__ rowptr,
                                 const int32_t* __restrict__ col, const float* __restrict__ val_in,
                                 const float* __restrict__ row_scale,
                                 const float* __restrict__ col_scale, float* __restrict__ val_out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nnz) return;
    float v = val_in ? val_in[i] : 1.0f;
    if (row_scale) v = v * row_scale[row_of(rowptr, n, (int32_t)i)];
    if (col_scale) v = v * col_scale[col[i]];
    val_out[i] = v;
}

inline unsigned bits_for(int64_t n) {  // bits needed to represent values in [0, n)
    unsigned b = 0;
    while (b < 32 && ((int64_t)1 << b) < n) ++b;
    return b == 0 ? 1 : b;
}

inline dim3 grid_for(int64_t n) { return dim3((unsigned)mi_ceil_div(n > 0 ? n : 1, kBlock)); }

// sort (keys, vals) ascending on the low `end_bit` bits; result pointers returned.
int sort_pairs(uint64_t* k0, uint64_t* k1, int32_t* v0, int32_t* v1, int64_t n, unsigned end_bit,
               MiArena& arena, hipStream_t s, uint64_t** k_sorted, int32_t** v_sorted) {
    rocprim::double_buffer<uint64_t> kb(k0, k1);
    rocprim::double_buffer<int32_t> vb(v0, v1);
    size_t tmp_bytes = 0;
    MI_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, kb, vb, (size_t)n, 0u, end_bit, s));
    char* tmp = arena.take<char>(tmp_bytes ? tmp_bytes : 1);
    if (!tmp) return MI_ERR_WORKSPACE;
    MI_HIP(rocprim::radix_sort_pairs(tmp, tmp_bytes, kb, vb, (size_t)n, 0u, end_bit, s));
    *k_sorted = kb.current();
    *v_sorted = vb.current();
    return 0;
}

size_t sort_ws_bound(int64_t nnz) {
    // two key buffers, two value buffers, rocPRIM scratch (histograms + look-back state)
    size_t n = (size_t)(nnz > 0 ? nnz : 1);
    return 2 * mi_align_up(n * 8, 256) + 2 * mi_align_up(n * 4, 256) + (n * 2) + ((size_t)64 << 20);
}

}  // namespace

extern "C" {

size_t mi_coo_to_csr_workspace_bytes(int64_t n_rows, int64_t nnz) {
    (void)n_rows;
    return sort_ws_bound(nnz);
}

int mi_coo_to_csr_i32(int64_t n_rows, int64_t n_cols, int64_t nnz, const int64_t* row,
                      const int64_t* col, int32_t* rowptr, int32_t* col_out, int32_t* perm,
                      void* ws, size_t ws_bytes, mi_stream_t stream) {
    MI_CHECK_ARG(n_rows >= 0 && n_cols >= 0 && nnz >= 0 && rowptr);
    MI_CHECK_ARG(nnz == 0 || (row && col && col_out && ws));
    if (n_rows >= INT32_MAX || n_cols >= INT32_MAX || nnz >= INT32_MAX) return MI_ERR_TOO_LARGE;
    hipStream_t s = (hipStream_t)stream;
    if (nnz == 0) {
        MI_HIP(hipMemsetAsync(rowptr, 0, (size_t)(n_rows + 1) * sizeof(int32_t), s));
        return 0;
    }
    MiArena arena(ws, ws_bytes);
    uint64_t* k0 = arena.take<uint64_t>(nnz);
    uint64_t* k1 = arena.take<uint64_t>(nnz);
    int32_t* v0 = arena.take<int32_t>(nnz);
    int32_t* v1 = arena.take<int32_t>(nnz);
    if (!k0 || !k1 || !v0 || !v1) return MI_ERR_WORKSPACE;
    const unsigned cb = bits_for(n_cols);
    hipLaunchKernelGGL(make_keys_coo, grid_for(nnz), dim3(kBlock), 0, s, nnz, row, col, cb, k0, v0);
    uint64_t* ks; int32_t* vs;
    int rc = sort_pairs(k0, k1, v0, v1, nnz, cb + bits_for(n_rows), arena, s, &ks, &vs);
    if (rc) return rc;
    hipLaunchKernelGGL(rowptr_from_keys, grid_for(n_rows + 1), dim3(kBlock), 0, s, n_rows, nnz, cb, ks, rowptr);
    hipLaunchKernelGGL(decode_keys, grid_for(nnz), dim3(kBlock), 0, s, nnz, cb, ks, vs, col_out, perm);
    return mi_launch_status();
}

size_t mi_csr_transpose_workspace_bytes(int64_t n_rows, int64_t nnz) {
    (void)n_rows;
    return sort_ws_bound(nnz);
}

int mi_csr_transpose_i32(int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t* rowptr,
                         const int32_t* col, int32_t* rowptr_t, int32_t* col_t, int32_t* perm_t,
                         void* ws, size_t ws_bytes, mi_stream_t stream) {
    MI_CHECK_ARG(n_rows >= 0 && n_cols >= 0 && nnz >= 0 && rowptr && rowptr_t);
    MI_CHECK_ARG(nnz == 0 || (col && col_t && ws));
    if (n_rows >= INT32_MAX || n_cols >= INT32_MAX || nnz >= INT32_MAX) return MI_ERR_TOO_LARGE;
    hipStream_t s = (hipStream_t)stream;
    if (nnz == 0) {
        MI_HIP(hipMemsetAsync(rowptr_t, 0, (size_t)(n_cols + 1) * sizeof(int32_t), s));
        return 0;
    }
    MiArena arena(ws, ws_bytes);
    uint64_t* k0 = arena.take<uint64_t>(nnz);
    uint64_t* k1 = arena.take<uint64_t>(nnz);
    int32_t* v0 = arena.take<int32_t>(nnz);
    int32_t* v1 = arena.take<int32_t>(nnz);
    if (!k0 || !k1 || !v0 || !v1) return MI_ERR_WORKSPACE;
    const unsigned cb = bits_for(n_rows);  // transposed: the old row is the minor key
    hipLaunchKernelGGL(make_keys_transpose, grid_for(nnz), dim3(kBlock), 0, s, n_rows, nnz, rowptr, col, cb, k0, v0);
    uint64_t* ks; int32_t* vs;
    int rc = sort_pairs(k0, k1, v0, v1, nnz, cb + bits_for(n_cols), arena, s, &ks, &vs);
    if (rc) return rc;
    hipLaunchKernelGGL(rowptr_from_keys, grid_for(n_cols + 1), dim3(kBlock), 0, s, n_cols, nnz, cb, ks, rowptr_t);
    hipLaunchKernelGGL(decode_keys, grid_for(nnz), dim3(kBlock), 0, s, nnz, cb, ks, vs, col_t, perm_t);
    return mi_launch_status();
}

int mi_gather_f32(int64_t n, const float* src, const int32_t* idx, float* out, mi_stream_t stream) {
    MI_CHECK_ARG(n >= 0);
    if (n == 0) return 0;
    MI_CHECK_ARG(src && idx && out);
    hipLaunchKernelGGL(gather_f32_kernel, grid_for(n), dim3(kBlock), 0, (hipStream_t)stream, n, src, idx, out);
    return mi_launch_status();
}

int mi_csr_expand_rows(int64_t n_rows, const int32_t* rowptr, int32_t* row_of_edge, int64_t nnz,
                       mi_stream_t stream) {
    MI_CHECK_ARG(n_rows >= 0 && nnz >= 0 && rowptr);
    if (nnz == 0) return 0;
    MI_CHECK_ARG(row_of_edge);
    hipLaunchKernelGGL(expand_rows_kernel, grid_for(nnz), dim3(kBlock), 0, (hipStream_t)stream, n_rows, nnz, rowptr, row_of_edge);
    return mi_launch_status();
}

int mi_gcn_norm_csr_f32(int64_t n, int64_t nnz, const int32_t* rowptr, const int32_t* col,
                        const float* val_in, float* val_out, float* dis_out, mi_stream_t stream) {
    MI_CHECK_ARG(n >= 0 && nnz >= 0 && rowptr && dis_out);
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gcn_dis_kernel, grid_for(n * MI_WAVE), dim3(kBlock), 0, s, n, rowptr, val_in, dis_out);
    if (nnz > 0) {
        MI_CHECK_ARG(col && val_out);
        hipLaunchKernelGGL(gcn_scale_kernel, grid_for(nnz), dim3(kBlock), 0, s, n, nnz, rowptr, col, val_in, dis_out, dis_out, val_out);
    }
    return mi_launch_status();
}

int mi_scale_csr_f32(int64_t n_rows, int64_t nnz, const int32_t* rowptr, const int32_t* col,
                     const float* val_in, const float* row_scale, const float* col_scale,
                     float* val_out, mi_stream_t stream) {
    MI_CHECK_ARG(n_rows >= 0 && nnz >= 0);
    if (nnz == 0) return 0;
    MI_CHECK_ARG(rowptr && col && val_out);
    hipLaunchKernelGGL(gcn_scale_kernel, grid_for(nnz), dim3(kBlock), 0, (hipStream_t)stream, n_rows, nnz,
                       rowptr, col, val_in, row_scale, col_scale, val_out);
    return mi_launch_status();
}

}  // extern "C"

// K1 / K2: CSR SpMM with fused epilogue — the LightGCN propagate (model/lightgcn.py:63-68,87).
//
//   acc[r,:] = sum_p val[p] * X[col[p],:]
//   Y[r,:]   = acc                               (optional)
//   S[r,:]   = scale * (addend[r,:] + acc)        (optional; the running layer sum / mean)
//
// HBM-bound gather (0.49 flop/byte at D=128): the design is about keeping many whole-row
// 16-byte-per-lane loads in flight per wavefront, not about arithmetic.
//
// Mapping (wave64): a row of D floats is covered by LPR = D/4 lanes of float4, so one load
// instruction fetches NB = 64/LPR neighbour rows at once (D=128: 2 rows, 1 KiB per
// instruction).  (col,val) pairs are read 64 at a time, coalesced, one per lane, and handed to
// the sub-groups with ds_bpermute (__shfl); UNROLL independent row loads per lane are issued
// before the first fma.  The NB partial sums are combined with log2(NB) DPP/xor steps at the
// end.  One wavefront owns one destination row; rows longer than plan->chunk are cut into
// chunk-sized work items whose partial sums are reduced in item order by a fix-up kernel, so
// there are no float atomics and results are bitwise reproducible.
#include "common.hpp"
#include <rocprim/rocprim.hpp>

namespace {

#ifndef MI_SPMM_NT
#define MI_SPMM_NT 0   // 1: non-temporal loads of col/val and stores of Y/S.  Measured on C2 (tools/ab_spmm.py,
                       // round 1): 1.226 ms vs 1.227 ms — no gain; UNROLL 8: 1.241 ms; both: 1.274 ms.  Kept off.
#endif
#ifndef MI_SPMM_UNROLL
#define MI_SPMM_UNROLL 4
#endif
typedef float mi_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mi_nt_store4(float4* p, const float4& v) {
    mi_f4v x = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(x, reinterpret_cast<mi_f4v*>(p));
}
constexpr int kWavesPerBlock = 4;
constexpr int kBlock = kWavesPerBlock * MI_WAVE;

struct Epilogue {
    float4* Y;            int64_t ldy4;
    const float4* addend; int64_t lda4;
    float4* S;            int64_t lds4;
    float scale;
};

// Sparse-operand extensions (mi_spmm_csr_ex_f32); all pointers nullable, branches are wave-uniform.
struct Ex {
    const int32_t* x_map;       // [n_cols]: X is compact, neighbour c reads X[x_map[c]]; < 0 = an all-zero row, skipped
    const int32_t* addend_map;  // [n_rows]: addend is compact, row r adds addend[addend_map[r]]; < 0 = nothing
    const int32_t* row_list;    // [n_list]: compute only these rows; Y / S / addend are compact, indexed by list position
    const int32_t* n_list_dev;  // device count of valid row_list entries (<= the launch bound), or null
    const int32_t* long_index;  // [n_rows]: index into the plan's long rows, -1 for short rows (row_list mode)
};

// Accumulates entries [beg, end) of one row into acc (valid in lanes of sub-group 0 after
// the final cross-sub-group reduce).
template <int LPR, int VPL, int UNROLL, bool SPARSE>
__device__ __forceinline__ void wave_accumulate(const int32_t* __restrict__ col,
                                                const float* __restrict__ val,
                                                const float4* __restrict__ X4, int64_t ldx4, int d4,
                                                int32_t beg, int32_t end, float4 (&acc)[VPL],
                                                const int32_t* __restrict__ x_map = nullptr) {
    constexpr int NB = MI_WAVE / LPR;
    const int lane = mi_lane();
    const int g = lane / LPR;
    const int li = lane % LPR;
#pragma unroll
    for (int v = 0; v < VPL; ++v) acc[v] = mi_f4_zero();

    for (int32_t base = beg; base < end; base += MI_WAVE) {
        const int n = min((int32_t)MI_WAVE, end - base);
        int32_t my_c = 0;
        float my_v = 0.f;
        if (lane < n) {
#if MI_SPMM_NT
            my_c = __builtin_nontemporal_load(col + base + lane);
            my_v = __builtin_nontemporal_load(val + base + lane);
#else
            my_c = col[base + lane];
            my_v = val[base + lane];
#endif
            if (SPARSE && x_map) my_c = x_map[my_c];  // compact row of X, or < 0 for a row that is all zeros
        }
        if (SPARSE && x_map && __ballot(my_c >= 0 && lane < n) == 0ull) continue;  // nothing to gather in this 64-entry group
        for (int j = 0; j < n; j += NB * UNROLL) {
            float w[UNROLL];
            float4 x[UNROLL][VPL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int idx = j + u * NB + g;
                const int32_t c = __shfl(my_c, idx & (MI_WAVE - 1), MI_WAVE);
                w[u] = __shfl(my_v, idx & (MI_WAVE - 1), MI_WAVE);
                const bool ok = idx < n && c >= 0;
                const float4* src = X4 + (int64_t)c * ldx4;
#pragma unroll
                for (int v = 0; v < VPL; ++v) {
                    const int e = li + v * LPR;
                    x[u][v] = (ok && e < d4) ? src[e] : mi_f4_zero();
                }
                if (!ok) w[u] = 0.f;
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
#pragma unroll
                for (int v = 0; v < VPL; ++v) mi_f4_fma(acc[v], w[u], x[u][v]);
        }
    }
#pragma unroll
    for (int m = MI_WAVE / 2; m >= LPR; m >>= 1)
#pragma unroll
        for (int v = 0; v < VPL; ++v) acc[v] = mi_f4_add(acc[v], mi_f4_shfl_xor(acc[v], m));
}

// r = output row index (compact position in row_list mode)
template <int LPR, int VPL>
__device__ __forceinline__ void apply_epilogue(const Epilogue& ep, int64_t r, int d4,
                                               const float4 (&acc)[VPL], const float4 (&a)[VPL]) {
    const int lane = mi_lane();
    if (lane >= LPR) return;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        const int e = lane + v * LPR;
        if (e >= d4) continue;
#if MI_SPMM_NT
        if (ep.Y) mi_nt_store4(ep.Y + r * ep.ldy4 + e, acc[v]);
#else
        if (ep.Y) ep.Y[r * ep.ldy4 + e] = acc[v];
#endif
        if (ep.S) {
            float4 o;
            o.x = ep.scale * (a[v].x + acc[v].x);
            o.y = ep.scale * (a[v].y + acc[v].y);
            o.z = ep.scale * (a[v].z + acc[v].z);
            o.w = ep.scale * (a[v].w + acc[v].w);
#if MI_SPMM_NT
            mi_nt_store4(ep.S + r * ep.lds4 + e, o);
#else
            ep.S[r * ep.lds4 + e] = o;
#endif
        }
    }
}

// ar = addend row index, < 0 for "no addend row"
template <int LPR, int VPL>
__device__ __forceinline__ void load_addend(const Epilogue& ep, int64_t ar, int d4, float4 (&a)[VPL]) {
    const int lane = mi_lane();
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        const int e = lane + v * LPR;
        a[v] = (ep.S && ep.addend && ar >= 0 && lane < LPR && e < d4) ? ep.addend[ar * ep.lda4 + e] : mi_f4_zero();
    }
}

// One wavefront per row; rows with more than `chunk` entries are left to the split path.
// SPARSE = false: the dense product (every entry gathered; addend_map allowed) — the kernel the
// roofline is quoted on.  SPARSE = true: x_map / row_list launches of the fused train step, kept as a
// separate instantiation so that profiles list them apart.
template <int LPR, int VPL, int UNROLL, bool SPARSE>
__global__ __launch_bounds__(kBlock) void spmm_rows_kernel(int64_t n_rows, int d4,
                                                           const int32_t* __restrict__ rowptr,
                                                           const int32_t* __restrict__ col,
                                                           const float* __restrict__ val,
                                                           const float4* __restrict__ X4, int64_t ldx4,
                                                           Epilogue ep, int32_t chunk, Ex ex) {
    const int64_t i = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x / MI_WAVE);
    if (i >= n_rows) return;
    const bool listed = SPARSE && ex.row_list != nullptr;
    if (listed && ex.n_list_dev && i >= *ex.n_list_dev) return;
    const int64_t r = listed ? ex.row_list[i] : i;
    const int32_t beg = rowptr[r], end = rowptr[r + 1];
    if (end - beg > chunk) return;
    float4 a[VPL], acc[VPL];
    const int64_t ar = listed ? i : (ex.addend_map ? (int64_t)ex.addend_map[r] : r);
    load_addend<LPR, VPL>(ep, ar, d4, a);
    wave_accumulate<LPR, VPL, UNROLL, SPARSE>(col, val, X4, ldx4, d4, beg, end, acc, ex.x_map);
    apply_epilogue<LPR, VPL>(ep, listed ? i : r, d4, acc, a);
}

// Split rows: one wavefront per work item (row, begin, end, slot) -> partial[slot, :].
template <int LPR, int VPL, int UNROLL, bool SPARSE>
__global__ __launch_bounds__(kBlock) void spmm_items_kernel(int32_t n_items, int d4,
                                                            const int32_t* __restrict__ items,
                                                            const int32_t* __restrict__ col,
                                                            const float* __restrict__ val,
                                                            const float4* __restrict__ X4, int64_t ldx4,
                                                            float4* __restrict__ partial,
                                                            const int32_t* __restrict__ x_map) {
    const int32_t it = blockIdx.x * kWavesPerBlock + (threadIdx.x / MI_WAVE);
    if (it >= n_items) return;
    const int32_t beg = items[4 * it + 1], end = items[4 * it + 2], slot = items[4 * it + 3];
    float4 acc[VPL];
    wave_accumulate<LPR, VPL, UNROLL, SPARSE>(col, val, X4, ldx4, d4, beg, end, acc, x_map);
    const int lane = mi_lane();
    if (lane < LPR) {
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            const int e = lane + v * LPR;
            if (e < d4) partial[(int64_t)slot * d4 + e] = acc[v];
        }
    }
}

// One 256-thread block per split row.  Its 4 wavefronts x NB sub-groups stride over the row's
// partial sums (several loads in flight each), then combine through LDS in a fixed order, so the
// result does not depend on scheduling.  Wave 0 applies the epilogue.
template <int LPR, int VPL, bool SPARSE>
__global__ __launch_bounds__(kBlock) void spmm_fixup_kernel(int32_t n_long, int d4,
                                                            const int32_t* __restrict__ long_rows,
                                                            const int32_t* __restrict__ item_ptr,
                                                            const float4* __restrict__ partial,
                                                            Epilogue ep, Ex ex) {
    constexpr int NB = MI_WAVE / LPR;
    constexpr int NSG = NB * kWavesPerBlock;  // sub-groups per block
    __shared__ float4 red[kWavesPerBlock][VPL][LPR];
    // all exits below depend on blockIdx only: the whole block leaves together, before any barrier
    int32_t i = blockIdx.x;
    int64_t r, out_row, ar;
    if (SPARSE && ex.row_list) {  // n_long = launch bound on the list length
        if (i >= n_long || (ex.n_list_dev && i >= *ex.n_list_dev)) return;
        r = ex.row_list[i];
        out_row = ar = i;
        i = ex.long_index[r];
        if (i < 0) return;  // a short row: done by spmm_rows_kernel
    } else {
        if (i >= n_long) return;
        r = long_rows[i];
        out_row = r;
        ar = ex.addend_map ? (int64_t)ex.addend_map[r] : r;
    }
    const int32_t sb = item_ptr[i], se = item_ptr[i + 1];
    const int lane = mi_lane();
    const int wave = threadIdx.x / MI_WAVE;
    const int g = lane / LPR, li = lane % LPR;
    const int sg = wave * NB + g;
    float4 acc[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) acc[v] = mi_f4_zero();
    for (int32_t s = sb + sg; s < se; s += NSG * 4) {
        float4 x[4][VPL];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < VPL; ++v) {
                const int e = li + v * LPR;
                const int32_t ss = s + u * NSG;
                x[u][v] = (ss < se && e < d4) ? partial[(int64_t)ss * d4 + e] : mi_f4_zero();
            }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < VPL; ++v) acc[v] = mi_f4_add(acc[v], x[u][v]);
    }
#pragma unroll
    for (int m = MI_WAVE / 2; m >= LPR; m >>= 1)
#pragma unroll
        for (int v = 0; v < VPL; ++v) acc[v] = mi_f4_add(acc[v], mi_f4_shfl_xor(acc[v], m));
    if (lane < LPR) {
#pragma unroll
        for (int v = 0; v < VPL; ++v) red[wave][v][lane] = acc[v];
    }
    __syncthreads();
    if (wave != 0) return;
    float4 a[VPL];
    load_addend<LPR, VPL>(ep, ar, d4, a);
    if (lane < LPR) {
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            float4 t = red[0][v][lane];
#pragma unroll
            for (int w = 1; w < kWavesPerBlock; ++w) t = mi_f4_add(t, red[w][v][lane]);
            acc[v] = t;
        }
    }
    apply_epilogue<LPR, VPL>(ep, out_row, d4, acc, a);
}

// ---- plan construction --------------------------------------------------------------------
__global__ void plan_flags_kernel(int64_t n_rows, const int32_t* __restrict__ rowptr, int32_t chunk,
                                  int32_t* __restrict__ is_long, int32_t* __restrict__ n_it) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_rows) return;
    int32_t deg = (r < n_rows) ? rowptr[r + 1] - rowptr[r] : 0;
    bool lg = deg > chunk;
    is_long[r] = lg ? 1 : 0;
    n_it[r] = lg ? (deg + chunk - 1) / chunk : 0;
}

__global__ void plan_fill_kernel(int64_t n_rows, const int32_t* __restrict__ rowptr, int32_t chunk,
                                 const int32_t* __restrict__ long_off, const int32_t* __restrict__ item_off,
                                 int32_t* __restrict__ long_rows, int32_t* __restrict__ item_ptr,
                                 int32_t* __restrict__ items, int32_t* __restrict__ long_index) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_rows) return;
    if (r == n_rows) {  // sentinel: item_ptr[n_long] = n_items
        item_ptr[long_off[r]] = item_off[r];
        return;
    }
    const int32_t b = rowptr[r], e = rowptr[r + 1];
    if (e - b <= chunk) {
        if (long_index) long_index[r] = -1;
        return;
    }
    const int32_t li = long_off[r], io = item_off[r];
    if (long_index) long_index[r] = li;
    long_rows[li] = (int32_t)r;
    item_ptr[li] = io;
    int32_t k = 0;
    for (int32_t p = b; p < e; p += chunk, ++k) {
        items[4 * (io + k) + 0] = (int32_t)r;
        items[4 * (io + k) + 1] = p;
        items[4 * (io + k) + 2] = min(p + chunk, e);
        items[4 * (io + k) + 3] = io + k;
    }
}

template <int LPR, int VPL, bool SPARSE>
int launch_spmm_mode(int64_t n_rows, int d4, const int32_t* rowptr, const int32_t* col, const float* val,
                     const float4* X4, int64_t ldx4, const Epilogue& ep, const mi_spmm_plan* plan,
                     float4* partial, const Ex& ex, int64_t n_list, hipStream_t s) {
    constexpr int UNROLL = (VPL == 1) ? MI_SPMM_UNROLL : 2;
    const int32_t chunk = plan ? plan->chunk : INT32_MAX;
    const bool listed = SPARSE && ex.row_list != nullptr;
    if (plan && plan->n_items > 0) {  // row_list mode still reduces every split row: hubs are few and almost always wanted
        dim3 gi((unsigned)mi_ceil_div(plan->n_items, kWavesPerBlock));
        hipLaunchKernelGGL((spmm_items_kernel<LPR, VPL, UNROLL, SPARSE>), gi, dim3(kBlock), 0, s, plan->n_items, d4,
                           plan->items, col, val, X4, ldx4, partial, ex.x_map);
    }
    const int64_t n_out = listed ? n_list : n_rows;
    if (n_out > 0) {
        dim3 gr((unsigned)mi_ceil_div(n_out, kWavesPerBlock));
        hipLaunchKernelGGL((spmm_rows_kernel<LPR, VPL, UNROLL, SPARSE>), gr, dim3(kBlock), 0, s, n_out, d4, rowptr,
                           col, val, X4, ldx4, ep, chunk, ex);
    }
    if (plan && plan->n_long_rows > 0) {
        const int64_t nf = listed ? n_list : (int64_t)plan->n_long_rows;
        if (nf > 0)
            hipLaunchKernelGGL((spmm_fixup_kernel<LPR, VPL, SPARSE>), dim3((unsigned)nf), dim3(kBlock), 0, s,
                               (int32_t)nf, d4, plan->long_rows, plan->item_ptr, partial, ep, ex);
    }
    return mi_launch_status();
}

template <int LPR, int VPL>
int launch_spmm(int64_t n_rows, int d4, const int32_t* rowptr, const int32_t* col, const float* val,
                const float4* X4, int64_t ldx4, const Epilogue& ep, const mi_spmm_plan* plan,
                float4* partial, const Ex& ex, int64_t n_list, hipStream_t s) {
    if (ex.x_map || ex.row_list)
        return launch_spmm_mode<LPR, VPL, true>(n_rows, d4, rowptr, col, val, X4, ldx4, ep, plan, partial, ex, n_list, s);
    return launch_spmm_mode<LPR, VPL, false>(n_rows, d4, rowptr, col, val, X4, ldx4, ep, plan, partial, ex, n_list, s);
}

}  // namespace

extern "C" {

int mi_spmm_plan_bounds(int64_t n_rows, int64_t nnz, int32_t chunk, int64_t* max_long_rows,
                        int64_t* max_items) {
    MI_CHECK_ARG(n_rows >= 0 && nnz >= 0 && chunk > 0 && max_long_rows && max_items);
    int64_t ml = nnz / ((int64_t)chunk + 1);  // each long row holds > chunk entries
    if (ml > n_rows) ml = n_rows;
    *max_long_rows = ml;
    *max_items = nnz / chunk + ml;  // sum ceil(deg/chunk) <= nnz/chunk + n_long
    return 0;
}

size_t mi_spmm_plan_workspace_bytes(int64_t n_rows) {
    size_t n = (size_t)n_rows + 1;
    return 4 * mi_align_up(n * 4, 256) + ((size_t)16 << 20);
}

int mi_spmm_plan_build(int64_t n_rows, const int32_t* rowptr, int32_t chunk, mi_spmm_plan* plan,
                       void* ws, size_t ws_bytes, mi_stream_t stream) {
    MI_CHECK_ARG(n_rows >= 0 && rowptr && chunk > 0 && plan && ws);
    if (n_rows >= INT32_MAX) return MI_ERR_TOO_LARGE;
    hipStream_t s = (hipStream_t)stream;
    plan->chunk = chunk;
    plan->n_long_rows = 0;
    plan->n_items = 0;
    plan->reserved = 0;
    const int64_t n1 = n_rows + 1;
    MiArena arena(ws, ws_bytes);
    int32_t* is_long = arena.take<int32_t>(n1);
    int32_t* n_it = arena.take<int32_t>(n1);
    int32_t* long_off = arena.take<int32_t>(n1);
    int32_t* item_off = arena.take<int32_t>(n1);
    if (!is_long || !n_it || !long_off || !item_off) return MI_ERR_WORKSPACE;
    dim3 g((unsigned)mi_ceil_div(n1, 256));
    hipLaunchKernelGGL(plan_flags_kernel, g, dim3(256), 0, s, n_rows, rowptr, chunk, is_long, n_it);
    size_t tmp_bytes = 0;
    MI_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, is_long, long_off, 0, (size_t)n1,
                                   rocprim::plus<int32_t>(), s));
    char* tmp = arena.take<char>(tmp_bytes ? tmp_bytes : 1);
    if (!tmp) return MI_ERR_WORKSPACE;
    MI_HIP(rocprim::exclusive_scan(tmp, tmp_bytes, is_long, long_off, 0, (size_t)n1,
                                   rocprim::plus<int32_t>(), s));
    MI_HIP(rocprim::exclusive_scan(tmp, tmp_bytes, n_it, item_off, 0, (size_t)n1,
                                   rocprim::plus<int32_t>(), s));
    int32_t totals[2] = {0, 0};
    MI_HIP(hipMemcpyAsync(&totals[0], long_off + n_rows, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    MI_HIP(hipMemcpyAsync(&totals[1], item_off + n_rows, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    MI_HIP(hipStreamSynchronize(s));
    plan->n_long_rows = totals[0];
    plan->n_items = totals[1];
    if (totals[0] > 0) {
        MI_CHECK_ARG(plan->long_rows && plan->item_ptr && plan->items);
        hipLaunchKernelGGL(plan_fill_kernel, g, dim3(256), 0, s, n_rows, rowptr, chunk, long_off, item_off,
                           plan->long_rows, plan->item_ptr, plan->items, plan->long_index);
        MI_HIP(hipStreamSynchronize(s));  // ws may be released by the caller on return
    } else if (plan->long_index) {
        MI_HIP(hipMemsetAsync(plan->long_index, 0xFF, (size_t)n_rows * sizeof(int32_t), s));
        MI_HIP(hipStreamSynchronize(s));
    }
    return mi_launch_status();
}

size_t mi_spmm_workspace_bytes(const mi_spmm_plan* plan, int64_t d) {
    if (!plan || plan->n_items <= 0) return 0;
    return mi_align_up((size_t)plan->n_items * (size_t)d * sizeof(float), 256);
}

int mi_spmm_csr_ex_f32(int64_t n_rows, int64_t d, const int32_t* rowptr, const int32_t* col,
                       const float* val, const float* X, int64_t ldx, float* Y, int64_t ldy,
                       const float* addend, int64_t lda, float* S, int64_t lds, float scale,
                       const mi_spmm_plan* plan, const mi_spmm_ex* exh, void* ws, size_t ws_bytes,
                       mi_stream_t stream) {
    MI_CHECK_ARG(n_rows >= 0 && d > 0 && rowptr);
    if (n_rows == 0) return 0;
    if (d % 4 != 0 || d > 512) return MI_ERR_UNSUPPORTED;
    if (n_rows >= INT32_MAX) return MI_ERR_TOO_LARGE;
    MI_CHECK_ARG(X && (Y || S));
    MI_CHECK_ARG(ldx % 4 == 0 && ldx >= d && mi_aligned16(X));
    MI_CHECK_ARG(!Y || (ldy % 4 == 0 && ldy >= d && mi_aligned16(Y) && Y != X));
    MI_CHECK_ARG(!S || (lds % 4 == 0 && lds >= d && mi_aligned16(S) && S != X));
    MI_CHECK_ARG(!addend || (lda % 4 == 0 && lda >= d && mi_aligned16(addend)));
    Ex ex = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int64_t n_list = 0;
    if (exh) {
        ex.x_map = exh->x_map;
        ex.addend_map = exh->addend_map;
        ex.row_list = exh->row_list;
        ex.n_list_dev = exh->n_list_dev;
        n_list = exh->n_list;
        if (ex.row_list) {
            MI_CHECK_ARG(n_list >= 0 && !ex.addend_map);
            if (n_list == 0) return 0;
            if (plan && plan->n_long_rows > 0) {
                MI_CHECK_ARG(plan->long_index);
                ex.long_index = plan->long_index;
            }
        }
    }
    float4* partial = nullptr;
    if (plan && plan->n_items > 0) {
        if (!ws || ws_bytes < mi_spmm_workspace_bytes(plan, d)) return MI_ERR_WORKSPACE;
        MI_CHECK_ARG(mi_aligned16(ws));
        partial = reinterpret_cast<float4*>(ws);
    }
    Epilogue ep;
    ep.Y = reinterpret_cast<float4*>(Y);                 ep.ldy4 = ldy / 4;
    ep.addend = reinterpret_cast<const float4*>(addend); ep.lda4 = lda / 4;
    ep.S = reinterpret_cast<float4*>(S);                 ep.lds4 = lds / 4;
    ep.scale = scale;
    const float4* X4 = reinterpret_cast<const float4*>(X);
    const int d4 = (int)(d / 4);
    hipStream_t s = (hipStream_t)stream;
    if (d4 <= 8)   return launch_spmm<8, 1>(n_rows, d4, rowptr, col, val, X4, ldx / 4, ep, plan, partial, ex, n_list, s);
    if (d4 <= 16)  return launch_spmm<16, 1>(n_rows, d4, rowptr, col, val, X4, ldx / 4, ep, plan, partial, ex, n_list, s);
    if (d4 <= 32)  return launch_spmm<32, 1>(n_rows, d4, rowptr, col, val, X4, ldx / 4, ep, plan, partial, ex, n_list, s);
    if (d4 <= 64)  return launch_spmm<64, 1>(n_rows, d4, rowptr, col, val, X4, ldx / 4, ep, plan, partial, ex, n_list, s);
    return launch_spmm<64, 2>(n_rows, d4, rowptr, col, val, X4, ldx / 4, ep, plan, partial, ex, n_list, s);
}

int mi_spmm_csr_f32(int64_t n_rows, int64_t d, const int32_t* rowptr, const int32_t* col,
                    const float* val, const float* X, int64_t ldx, float* Y, int64_t ldy,
                    const float* addend, int64_t lda, float* S, int64_t lds, float scale,
                    const mi_spmm_plan* plan, void* ws, size_t ws_bytes, mi_stream_t stream) {
    return mi_spmm_csr_ex_f32(n_rows, d, rowptr, col, val, X, ldx, Y, ldy, addend, lda, S, lds, scale, plan,
                              nullptr, ws, ws_bytes, stream);
}

}  // extern "C"

// LightGCN training-step kernels other than the propagate:
//   K9   mi_sample_bpr_batch   data/lightgcn_loader.py:95-112 (+ PyG structured_negative_sampling)
//   a7/8 mi_bpr_fwd_bwd_f32    run_pipeline_lightgcn.py:133-155, utils/metrics_lightgcn.py:9-45
//   a9   mi_adam_dense_f32     run_pipeline_lightgcn.py:157-159 (torch.optim.Adam)
#include "common.hpp"
#include <cmath>
#include <rocprim/rocprim.hpp>

namespace {

constexpr int kBlock = 256;

// ------------------------------------------------------------------ sampler ------------
// Domain tags keep the edge-pick stream and the negative stream apart.
constexpr uint32_t kTagEdge = 0x45444745u;  // "EDGE"
constexpr uint32_t kTagNeg = 0x4E454721u;   // "NEG!"
constexpr int kMaxNegAttempts = 4096;

__device__ __forceinline__ bool csr_contains(const int32_t* __restrict__ rowptr,
                                             const int32_t* __restrict__ col, int64_t r,
                                             int32_t key) {
    int32_t lo = rowptr[r], hi = rowptr[r + 1];
    const int32_t end = hi;
    while (lo < hi) {
        int32_t mid = (lo + hi) >> 1;
        if (col[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo < end && col[lo] == key;
}

__global__ void sample_bpr_kernel(int64_t batch, int64_t nnz, const int32_t* __restrict__ rowptr,
                                  const int32_t* __restrict__ col,
                                  const int32_t* __restrict__ row_of_edge, int64_t neg_range,
                                  int32_t quirk, int32_t edges_in_order, uint64_t seed, uint64_t step,
                                  int64_t* __restrict__ users, int64_t* __restrict__ pos,
                                  int64_t* __restrict__ neg) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const uint32_t s0 = (uint32_t)step, s1 = (uint32_t)(step >> 32);
    MiPhilox r = mi_philox4x32((uint32_t)b, (uint32_t)((uint64_t)b >> 32), s0, s1 ^ kTagEdge, k0, k1);
    const uint64_t e = edges_in_order ? (uint64_t)b : (((uint64_t)r.c[0] << 32) | r.c[1]) % (uint64_t)nnz;
    const int64_t u = row_of_edge[e];
    const int32_t p = col[e];
    int32_t cand = 0;
    for (int t = 0; t < kMaxNegAttempts; ++t) {
        MiPhilox q = mi_philox4x32((uint32_t)e, (uint32_t)t, s0, s1 ^ kTagNeg, k0, k1);
        cand = (int32_t)((((uint64_t)q.c[0] << 32) | q.c[1]) % (uint64_t)neg_range);
        bool hit = csr_contains(rowptr, col, u, cand);
        // reference key collision: (u-1, item neg_range) aliases (u, 0) in row*num_nodes+col
        if (!hit && (quirk & 1) && cand == 0 && u > 0) hit = csr_contains(rowptr, col, u - 1, (int32_t)neg_range);
        // contains_neg_self_loops=False: the key u*num_nodes+u of every node u < num_nodes is in the rejection set too
        if (!hit && (quirk & 2) && (int64_t)cand == u) hit = true;
        if (!hit) break;
    }
    users[b] = u;
    pos[b] = p;
    neg[b] = cand;
}

// ------------------------------------------------------------------ BPR ----------------
__device__ __forceinline__ float softplus_ref(float x) {  // torch softplus, beta=1, threshold=20
    return x > 20.f ? x : log1pf(expf(x));
}
__device__ __forceinline__ float softplus_grad_ref(float x) {  // torch softplus_backward
    if (x > 20.f) return 1.f;
    float z = expf(x);
    return z / (z + 1.f);
}

// One wavefront per batch slot.  Each lane covers d/64 (rounded up) strided elements.
__global__ __launch_bounds__(kBlock) void bpr_slot_kernel(
    int64_t batch, int d, int64_t n_users, const int64_t* __restrict__ users,
    const int64_t* __restrict__ pos, const int64_t* __restrict__ neg,
    const float* __restrict__ fin, int64_t ldf, const float* __restrict__ e0, int64_t lde,
    float inv_batch, float g_scale, float reg_coef /* = lambda*reg_scale */,
    float* __restrict__ softplus_out, float* __restrict__ reg_out,
    float* __restrict__ coef_out, float* __restrict__ reg_w /* only when no gradient pass follows */,
    const int32_t* __restrict__ node_map) {
    const int64_t b = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / MI_WAVE;
    if (b >= batch) return;
    const int lane = mi_lane();
    const int64_t u = users[b], p = n_users + pos[b], n = n_users + neg[b];
    // rows of final / g_final: node ids, or compact slots when the batch-node map is given
    const int64_t fu = node_map ? node_map[u] : u, fp = node_map ? node_map[p] : p, fn = node_map ? node_map[n] : n;
    const float* uf = fin + fu * ldf;
    const float* pf = fin + fp * ldf;
    const float* nf = fin + fn * ldf;
    const float* u0 = e0 + u * lde;
    const float* p0 = e0 + p * lde;
    const float* n0 = e0 + n * lde;
    float sp = 0.f, sn = 0.f, rg = 0.f;
    for (int k = lane; k < d; k += MI_WAVE) {
        float a = uf[k];
        sp = fmaf(a, pf[k], sp);
        sn = fmaf(a, nf[k], sn);
        float x0 = u0[k], x1 = p0[k], x2 = n0[k];
        rg = fmaf(x0, x0, rg);
        rg = fmaf(x1, x1, rg);
        rg = fmaf(x2, x2, rg);
    }
    sp = mi_wave_sum(sp);
    sn = mi_wave_sum(sn);
    rg = mi_wave_sum(rg);
    const float x = sp - sn;
    if (lane == 0) {
        softplus_out[b] = softplus_ref(x);
        reg_out[b] = rg;
    }
    if (coef_out && lane == 0) {
        // loss = -mean softplus(x)  =>  dL/dx = -sigmoid'(x)/B; consumed by the segmented gradient kernels below
        coef_out[b] = -softplus_grad_ref(x) * inv_batch * g_scale;
    }
    // (with a gradient pass the weights come from the sorted references instead, bpr_chunk_kernel: the atomics of the few
    // hundred samples that share the most popular item serialised on one address — 49 of this kernel's 65 us at B = 16 384)
    if (reg_w && lane == 0) {
        const float w = 2.0f * reg_coef;
        atomicAdd(reg_w + u, w);
        atomicAdd(reg_w + p, w);
        atomicAdd(reg_w + n, w);
    }
}

// Single block, fixed order: loss = -sum(softplus)/B + lambda*sum(reg).
// ---- gradient of the final embeddings, without float atomics -------------------------------------------------
// A batch touches 3B rows with repeats (the most popular item is the positive of hundreds of samples).  The 3B
// references (sample b, role) are keyed by the row they touch — its compact slot under node_map, its node id otherwise —
// and sorted (stable radix sort: the references of a row stay in (role, b) order).  The sorted list is cut into chunks
// of 64 references, one wavefront each: a run of equal keys that lies inside a chunk is summed in order and added to its
// row by that wavefront alone; a run that crosses chunk borders leaves one partial row per chunk, and the wavefront of
// the chunk where the run starts adds those up in chunk order.  One writer per row, fixed orders throughout: the whole
// train step is bitwise reproducible.
constexpr int kRefChunk = 64;
constexpr int kMaxDPerLane = 8;  // d <= 512

__global__ void bpr_refs_kernel(int64_t batch, int64_t n_users, const int64_t* __restrict__ users,
                                const int64_t* __restrict__ pos, const int64_t* __restrict__ neg,
                                const int32_t* __restrict__ node_map, uint32_t* __restrict__ keys,
                                uint32_t* __restrict__ refs) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // reference id = role * batch + b
    if (r >= 3 * batch) return;
    const int role = (int)(r / batch);
    const int64_t b = r - role * batch;
    const int64_t node = role == 0 ? users[b] : (role == 1 ? n_users + pos[b] : n_users + neg[b]);
    keys[r] = (uint32_t)(node_map ? node_map[node] : node);
    refs[r] = (uint32_t)r;
}

template <int VPT>  // floats of a row per lane: ceil(d / 64)
__global__ __launch_bounds__(kBlock) void bpr_chunk_kernel(
    int64_t batch, int d, int64_t n_users, const int64_t* __restrict__ users, const int64_t* __restrict__ pos,
    const int64_t* __restrict__ neg, const float* __restrict__ fin, int64_t ldf, const float* __restrict__ coef,
    const uint32_t* __restrict__ keys, const uint32_t* __restrict__ refs, const int32_t* __restrict__ node_map,
    float* __restrict__ g_final, int64_t ldg, float* __restrict__ part_head, float* __restrict__ part_tail,
    float* __restrict__ reg_w, float reg_unit /* 2 * lambda * reg_scale: one reference's share */) {
    const int64_t n_ref = 3 * batch;
    const int64_t chunk = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / MI_WAVE;
    const int64_t j0 = chunk * kRefChunk;
    if (j0 >= n_ref) return;
    const int lane = mi_lane();
    const int n_here = (int)min((int64_t)kRefChunk, n_ref - j0);
    // this lane's reference: destination key, up to two (source row, weight) terms
    uint32_t my_key = 0xFFFFFFFFu;
    int64_t srcA = 0, srcB = 0;
    float wA = 0.f, wB = 0.f;
    if (lane < n_here) {
        my_key = keys[j0 + lane];
        const uint32_t r = refs[j0 + lane];
        const int role = (int)(r / batch);
        const int64_t b = r - (int64_t)role * batch;
        const float c = coef[b];
        const int64_t u = users[b], p = n_users + pos[b], n = n_users + neg[b];
        const int64_t fu = node_map ? node_map[u] : u, fp = node_map ? node_map[p] : p, fn = node_map ? node_map[n] : n;
        if (role == 0) { srcA = fp; wA = c; srcB = fn; wB = -c; }   // d x / d u = p - n
        else if (role == 1) { srcA = fu; wA = c; }
        else { srcA = fu; wA = -c; }
        // L2 weight of the node: (its references in the batch) x reg_unit.  The reference that opens the node's run in the
        // sorted list finds the run's end by bisection and is the entry's only writer (reg_w is zero on entry).
        if (reg_w) {
            const int64_t j = j0 + lane;
            if (j == 0 || keys[j - 1] != my_key) {
                int64_t lo = j + 1, hi = n_ref;        // first position in (j, n_ref] whose key differs
                while (lo < hi) {
                    const int64_t mid = (lo + hi) >> 1;
                    if (keys[mid] == my_key) lo = mid + 1; else hi = mid;
                }
                const int64_t node = role == 0 ? u : (role == 1 ? p : n);
                reg_w[node] = (float)(lo - j) * reg_unit;
            }
        }
    }
    const bool head_open = j0 > 0 && keys[j0 - 1] == __shfl(my_key, 0, MI_WAVE);
    const bool next_same = (j0 + n_here < n_ref) && keys[j0 + n_here] == __shfl(my_key, n_here - 1, MI_WAVE);
    float acc[VPT];
#pragma unroll
    for (int v = 0; v < VPT; ++v) acc[v] = 0.f;
    int run_start = 0;
    constexpr int U = 8;  // references whose source rows are in flight together
    for (int q0 = 0; q0 < n_here; q0 += U) {
        float xa[U][VPT], xb[U][VPT], wa[U], wb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = min(q0 + u, n_here - 1);
            const bool live = q0 + u < n_here;
            wa[u] = live ? __shfl(wA, q, MI_WAVE) : 0.f;
            wb[u] = live ? __shfl(wB, q, MI_WAVE) : 0.f;
            const float* ra = fin + __shfl(srcA, q, MI_WAVE) * ldf;
            const float* rb = fin + __shfl(srcB, q, MI_WAVE) * ldf;
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                const int k = lane + v * MI_WAVE;
                xa[u][v] = (live && k < d) ? ra[k] : 0.f;
                xb[u][v] = (wb[u] != 0.f && k < d) ? rb[k] : 0.f;   // wave-uniform condition
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = q0 + u;
            if (q >= n_here) break;
            const uint32_t kq = __shfl(my_key, q, MI_WAVE);
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                acc[v] = fmaf(wa[u], xa[u][v], acc[v]);
                if (wb[u] != 0.f) acc[v] = fmaf(wb[u], xb[u][v], acc[v]);
            }
            const bool last_of_run = (q + 1 == n_here) || __shfl(my_key, q + 1, MI_WAVE) != kq;
            if (!last_of_run) continue;
            const bool from_prev = run_start == 0 && head_open;
            const bool into_next = (q + 1 == n_here) && next_same;
            float* dst;
            if (from_prev) dst = part_head + chunk * d;          // finished by the chunk where the run starts
            else if (into_next) dst = part_tail + chunk * d;     // this chunk starts the run and finishes it below
            else dst = g_final + (int64_t)kq * ldg;              // the run lies inside this chunk: the row's only writer
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                const int k = lane + v * MI_WAVE;
                if (k < d) dst[k] = acc[v];   // g_final is zero on entry (header contract): a store, not a read-modify-write
                acc[v] = 0.f;
            }
            run_start = q + 1;
        }
    }
}

// One wavefront per chunk whose trailing run starts in it and runs on: tail partial + the head partials of the following
// chunks, in chunk order, added to the row.
template <int VPT>
__global__ __launch_bounds__(kBlock) void bpr_combine_kernel(int64_t batch, int d, const uint32_t* __restrict__ keys,
                                                             const float* __restrict__ part_head,
                                                             const float* __restrict__ part_tail,
                                                             float* __restrict__ g_final, int64_t ldg) {
    const int64_t n_ref = 3 * batch;
    const int64_t chunk = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / MI_WAVE;
    const int64_t j0 = chunk * kRefChunk;
    if (j0 >= n_ref) return;
    const int64_t j_last = min(j0 + kRefChunk, n_ref) - 1;
    if (j_last + 1 >= n_ref) return;                       // nothing after this chunk
    const uint32_t row = keys[j_last];
    if (keys[j_last + 1] != row) return;                   // the trailing run ends here
    if (keys[j0] == row && j0 > 0 && keys[j0 - 1] == row) return;  // the run started in an earlier chunk: not the owner
    const int lane = mi_lane();
    float acc[VPT];
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        const int k = lane + v * MI_WAVE;
        acc[v] = k < d ? part_tail[chunk * d + k] : 0.f;
    }
    for (int64_t nb = chunk + 1; nb * kRefChunk < n_ref && keys[nb * kRefChunk] == row; ++nb) {
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            const int k = lane + v * MI_WAVE;
            if (k < d) acc[v] += part_head[nb * d + k];
        }
        if (keys[min((nb + 1) * kRefChunk, n_ref) - 1] != row) break;   // the run ends inside chunk nb
    }
    float* dst = g_final + (int64_t)row * ldg;
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        const int k = lane + v * MI_WAVE;
        if (k < d) dst[k] = acc[v];
    }
}

__global__ __launch_bounds__(1024) void bpr_finish_kernel(int64_t batch,
                                                          const float* __restrict__ softplus_v,
                                                          const float* __restrict__ reg_v,
                                                          float inv_batch, float lambda,
                                                          float* __restrict__ loss_out) {
    __shared__ float sh_a[1024];
    __shared__ float sh_b[1024];
    const int t = threadIdx.x;
    float a = 0.f, r = 0.f;
#pragma unroll 4
    for (int64_t i = t; i < batch; i += 1024) {   // coalesced, independent loads (a contiguous slice per thread was a chain
        a += softplus_v[i];                        // of 16 strided loads: 17 us at B = 16 384); fixed order all the same
        r += reg_v[i];
    }
    sh_a[t] = a;
    sh_b[t] = r;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (t < s) {
            sh_a[t] += sh_a[t + s];
            sh_b[t] += sh_b[t + s];
        }
        __syncthreads();
    }
    if (t == 0) loss_out[0] = -sh_a[0] * inv_batch + lambda * sh_b[0];
}

// ------------------------------------------------------------------ batch node set ------
__global__ void mark_batch_nodes_kernel(int64_t batch, int64_t n_users, const int64_t* __restrict__ users,
                                        const int64_t* __restrict__ pos, const int64_t* __restrict__ neg,
                                        int32_t* __restrict__ flag) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    flag[users[b]] = 1;  // same value from every writer: order does not matter
    flag[n_users + pos[b]] = 1;
    flag[n_users + neg[b]] = 1;
}

__global__ void finish_batch_nodes_kernel(int64_t n_nodes, int64_t n_users, const int32_t* __restrict__ flag,
                                          const int32_t* __restrict__ slot, int32_t* __restrict__ gmap,
                                          int32_t* __restrict__ nodes, int32_t* __restrict__ count) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_nodes) return;
    if (r == n_users) count[1] = slot[r];  // unique USER nodes: slots [0, count[1]) are users, the rest items
    if (r == n_nodes) {
        count[0] = slot[r];
        return;
    }
    if (flag[r]) {
        gmap[r] = slot[r];
        nodes[slot[r]] = (int32_t)r;
    } else {
        gmap[r] = -1;
    }
}

// dst[i,:] = scale * ((accumulate ? dst[i,:] : 0) + src[rows[i] - row_offset,:]) for i in [*begin_dev, *n_dev);
// one wavefront per row, float4 lanes
__global__ __launch_bounds__(kBlock) void gather_rows_kernel(int64_t n_max, const int32_t* __restrict__ n_dev,
                                                             const int32_t* __restrict__ begin_dev,
                                                             int d4, const int32_t* __restrict__ rows,
                                                             int64_t row_offset,
                                                             const float4* __restrict__ src, int64_t lds4,
                                                             float4* __restrict__ dst, int64_t ldd4, int accumulate,
                                                             float scale) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / MI_WAVE;
    if (i >= n_max || (n_dev && i >= *n_dev) || (begin_dev && i < *begin_dev)) return;
    const int64_t r = (int64_t)rows[i] - row_offset;
    for (int e = mi_lane(); e < d4; e += MI_WAVE) {
        float4 v = src[r * lds4 + e];
        if (accumulate) v = mi_f4_add(dst[i * ldd4 + e], v);
        if (scale != 1.0f) { v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale; }
        dst[i * ldd4 + e] = v;
    }
}

// dst[rows[i] - row_offset,:] = src[i,:] for i in [*begin_dev, *n_dev): the inverse of gather_rows (rows are distinct)
__global__ __launch_bounds__(kBlock) void scatter_rows_kernel(int64_t n_max, const int32_t* __restrict__ n_dev,
                                                              const int32_t* __restrict__ begin_dev, int d4,
                                                              const int32_t* __restrict__ rows, int64_t row_offset,
                                                              const float4* __restrict__ src, int64_t lds4,
                                                              float4* __restrict__ dst, int64_t ldd4) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / MI_WAVE;
    if (i >= n_max || (n_dev && i >= *n_dev) || (begin_dev && i < *begin_dev)) return;
    const int64_t r = (int64_t)rows[i] - row_offset;
    for (int e = mi_lane(); e < d4; e += MI_WAVE) dst[r * ldd4 + e] = src[i * lds4 + e];
}

// ------------------------------------------------------------------ Adam ---------------
// One wavefront per group of rows: lane -> (row within group, float4 column), so the row index (needed
// for the L2 weight) costs no integer division, and every lane keeps 4 independent 16-byte loads in flight.
__global__ __launch_bounds__(kBlock) void adam_kernel(int64_t n_rows, int d4, float4* __restrict__ p,
                                                      int64_t ldp4, const float4* __restrict__ g,
                                                      int64_t ldg4, float4* __restrict__ m,
                                                      float4* __restrict__ v,
                                                      const float* __restrict__ reg_w, MiAdamConsts c) {
    // rows per pass of one block: blockDim.x / lanes-per-row, lanes-per-row = smallest power of two >= d4 (<= 256)
    int lpr = 1;
    while (lpr < d4 && lpr < (int)blockDim.x) lpr <<= 1;
    const int rows_per_block = blockDim.x / lpr;
    const int sub = threadIdx.x / lpr, e0 = threadIdx.x % lpr;
    for (int64_t r = (int64_t)blockIdx.x * rows_per_block + sub; r < n_rows; r += (int64_t)gridDim.x * rows_per_block) {
        const float w = reg_w ? reg_w[r] : 0.f;
        for (int e = e0; e < d4; e += lpr) {
            const int64_t i = r * d4 + e;
            float4 pp = p[r * ldp4 + e];
            float4 mm = m[i];
            float4 vv = v[i];
            mi_adam_update4(pp, g[r * ldg4 + e], mm, vv, reg_w != nullptr, w, c);
            p[r * ldp4 + e] = pp;
            m[i] = mm;
            v[i] = vv;
        }
    }
}

}  // namespace

extern "C" {

int mi_sample_bpr_batch(int64_t batch, int64_t nnz, const int32_t* rowptr, const int32_t* col,
                        const int32_t* row_of_edge, int64_t neg_range, int32_t quirk_user_rows,
                        int32_t edges_in_order, uint64_t seed, uint64_t step, int64_t* users, int64_t* pos,
                        int64_t* neg, mi_stream_t stream) {
    MI_CHECK_ARG(batch >= 0 && (!edges_in_order || batch <= nnz));
    if (batch == 0) return 0;
    MI_CHECK_ARG(nnz > 0 && neg_range > 0 && rowptr && col && row_of_edge && users && pos && neg);
    if (nnz >= INT32_MAX || neg_range >= INT32_MAX) return MI_ERR_TOO_LARGE;
    dim3 g((unsigned)mi_ceil_div(batch, kBlock));
    hipLaunchKernelGGL(sample_bpr_kernel, g, dim3(kBlock), 0, (hipStream_t)stream, batch, nnz, rowptr, col,
                       row_of_edge, neg_range, quirk_user_rows, edges_in_order, seed, step, users, pos, neg);
    return mi_launch_status();
}

size_t mi_batch_nodes_workspace_bytes(int64_t n_nodes) {
    const size_t n1 = (size_t)(n_nodes > 0 ? n_nodes : 0) + 1;
    return 2 * mi_align_up(n1 * sizeof(int32_t), 256) + ((size_t)16 << 20);
}

int mi_batch_nodes_i32(int64_t batch, int64_t n_users, int64_t n_nodes, const int64_t* users,
                       const int64_t* pos, const int64_t* neg, int32_t* gmap, int32_t* nodes,
                       int32_t* count, void* ws, size_t ws_bytes, mi_stream_t stream) {
    MI_CHECK_ARG(batch > 0 && n_users >= 0 && n_nodes >= n_users && n_nodes > 0);
    MI_CHECK_ARG(users && pos && neg && gmap && nodes && count && ws);
    if (n_nodes >= INT32_MAX) return MI_ERR_TOO_LARGE;
    hipStream_t s = (hipStream_t)stream;
    const int64_t n1 = n_nodes + 1;
    MiArena arena(ws, ws_bytes);
    int32_t* flag = arena.take<int32_t>(n1);
    int32_t* slot = arena.take<int32_t>(n1);
    if (!flag || !slot) return MI_ERR_WORKSPACE;
    MI_HIP(hipMemsetAsync(flag, 0, (size_t)n1 * sizeof(int32_t), s));
    hipLaunchKernelGGL(mark_batch_nodes_kernel, dim3((unsigned)mi_ceil_div(batch, kBlock)), dim3(kBlock), 0, s, batch,
                       n_users, users, pos, neg, flag);
    size_t tmp_bytes = 0;
    MI_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, flag, slot, 0, (size_t)n1, rocprim::plus<int32_t>(), s));
    char* tmp = arena.take<char>(tmp_bytes ? tmp_bytes : 1);
    if (!tmp) return MI_ERR_WORKSPACE;
    MI_HIP(rocprim::exclusive_scan(tmp, tmp_bytes, flag, slot, 0, (size_t)n1, rocprim::plus<int32_t>(), s));
    hipLaunchKernelGGL(finish_batch_nodes_kernel, dim3((unsigned)mi_ceil_div(n1, kBlock)), dim3(kBlock), 0, s, n_nodes,
                       n_users, flag, slot, gmap, nodes, count);
    return mi_launch_status();
}

int mi_gather_rows_f32(int64_t n_max, const int32_t* n_dev, const int32_t* begin_dev, int64_t d,
                       const int32_t* rows, int64_t row_offset, const float* src, int64_t ld_src, float* dst,
                       int64_t ld_dst, int32_t accumulate, float scale, mi_stream_t stream) {
    MI_CHECK_ARG(n_max >= 0 && d > 0);
    if (n_max == 0) return 0;
    if (d % 4 != 0) return MI_ERR_UNSUPPORTED;
    MI_CHECK_ARG(rows && src && dst && ld_src % 4 == 0 && ld_dst % 4 == 0 && ld_src >= d && ld_dst >= d);
    MI_CHECK_ARG(mi_aligned16(src) && mi_aligned16(dst));
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)mi_ceil_div(n_max * MI_WAVE, kBlock)), dim3(kBlock), 0,
                       (hipStream_t)stream, n_max, n_dev, begin_dev, (int)(d / 4), rows, row_offset,
                       reinterpret_cast<const float4*>(src), ld_src / 4, reinterpret_cast<float4*>(dst), ld_dst / 4,
                       accumulate, scale);
    return mi_launch_status();
}

int mi_scatter_rows_f32(int64_t n_max, const int32_t* n_dev, const int32_t* begin_dev, int64_t d,
                        const int32_t* rows, int64_t row_offset, const float* src, int64_t ld_src, float* dst,
                        int64_t ld_dst, mi_stream_t stream) {
    MI_CHECK_ARG(n_max >= 0 && d > 0);
    if (n_max == 0) return 0;
    if (d % 4 != 0) return MI_ERR_UNSUPPORTED;
    MI_CHECK_ARG(rows && src && dst && ld_src % 4 == 0 && ld_dst % 4 == 0 && ld_src >= d && ld_dst >= d);
    MI_CHECK_ARG(mi_aligned16(src) && mi_aligned16(dst));
    hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)mi_ceil_div(n_max * MI_WAVE, kBlock)), dim3(kBlock), 0,
                       (hipStream_t)stream, n_max, n_dev, begin_dev, (int)(d / 4), rows, row_offset,
                       reinterpret_cast<const float4*>(src), ld_src / 4, reinterpret_cast<float4*>(dst), ld_dst / 4);
    return mi_launch_status();
}

static size_t bpr_sort_tmp_bytes(int64_t batch) { return ((size_t)1 << 20) + mi_align_up((size_t)3 * batch * 2, 256); }

size_t mi_bpr_workspace_bytes(int64_t batch) {
    const size_t b = (size_t)(batch > 0 ? batch : 1);
    const size_t chunks = (3 * b + 63) / 64;
    // softplus, reg, coef per sample; (key, reference) double buffers of the 3B row references; rocPRIM scratch;
    // head / tail partial rows per 64-reference chunk (d <= 512)
    return 3 * mi_align_up(b * sizeof(float), 256) + 4 * mi_align_up(3 * b * sizeof(uint32_t), 256) + bpr_sort_tmp_bytes((int64_t)b) +
           2 * mi_align_up(chunks * 512 * sizeof(float), 256);
}

int mi_bpr_fwd_bwd_f32(int64_t batch, int64_t d, int64_t n_users, const int64_t* users,
                       const int64_t* pos, const int64_t* neg, const float* final_emb, int64_t ldf,
                       const float* e0, int64_t lde, float lambda, float g_scale, float reg_scale,
                       float* loss_out, float* g_final, int64_t ldg, float* reg_w,
                       const int32_t* node_map, void* ws, size_t ws_bytes, mi_stream_t stream) {
    MI_CHECK_ARG(batch > 0 && d > 0 && n_users >= 0);
    MI_CHECK_ARG(users && pos && neg && final_emb && e0 && loss_out && ws);
    MI_CHECK_ARG(ldf >= d && lde >= d && (!g_final || ldg >= d));
    if (3 * batch >= INT32_MAX) return MI_ERR_TOO_LARGE;
    if (g_final && d > MI_WAVE * kMaxDPerLane) return MI_ERR_UNSUPPORTED;
    if (ws_bytes < mi_bpr_workspace_bytes(batch)) return MI_ERR_WORKSPACE;
    const int64_t n_ref = 3 * batch, n_chunks = mi_ceil_div(n_ref, kRefChunk);
    MiArena arena(ws, ws_bytes);
    float* spv = arena.take<float>(batch);
    float* rgv = arena.take<float>(batch);
    float* coef = arena.take<float>(batch);
    uint32_t* k0 = arena.take<uint32_t>(n_ref);
    uint32_t* k1 = arena.take<uint32_t>(n_ref);
    uint32_t* r0 = arena.take<uint32_t>(n_ref);
    uint32_t* r1 = arena.take<uint32_t>(n_ref);
    const size_t tmp_cap = bpr_sort_tmp_bytes(batch);
    char* tmp = arena.take<char>(tmp_cap);
    float* part_head = arena.take<float>((size_t)n_chunks * 512);
    float* part_tail = arena.take<float>((size_t)n_chunks * 512);
    if (!spv || !rgv || !coef || !k0 || !k1 || !r0 || !r1 || !tmp || !part_head || !part_tail) return MI_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const float inv_b = 1.0f / (float)batch;
    dim3 g((unsigned)mi_ceil_div(batch * MI_WAVE, kBlock));
    hipLaunchKernelGGL(bpr_slot_kernel, g, dim3(kBlock), 0, s, batch, (int)d, n_users, users, pos, neg,
                       final_emb, ldf, e0, lde, inv_b, g_scale, reg_scale * lambda, spv, rgv,
                       g_final ? coef : nullptr, g_final ? nullptr : reg_w, node_map);
    if (g_final) {
        hipLaunchKernelGGL(bpr_refs_kernel, dim3((unsigned)mi_ceil_div(n_ref, 256)), dim3(256), 0, s, batch, n_users, users,
                           pos, neg, node_map, k0, r0);
        // compact slots are < 3B; node ids need all 32 bits
        unsigned bits = 32;
        if (node_map) {
            bits = 1;
            while (((int64_t)1 << bits) < n_ref) ++bits;
        }
        rocprim::double_buffer<uint32_t> keys(k0, k1), refs(r0, r1);
        size_t need = 0;
        MI_HIP(rocprim::radix_sort_pairs(nullptr, need, keys, refs, (size_t)n_ref, 0, bits, s));
        if (need > tmp_cap) return MI_ERR_WORKSPACE;
        MI_HIP(rocprim::radix_sort_pairs(tmp, need, keys, refs, (size_t)n_ref, 0, bits, s));
        dim3 gc((unsigned)mi_ceil_div(n_chunks * MI_WAVE, kBlock));
#define MI_BPR_GO(V)                                                                                                       \
    do {                                                                                                                    \
        hipLaunchKernelGGL(bpr_chunk_kernel<V>, gc, dim3(kBlock), 0, s, batch, (int)d, n_users, users, pos, neg, final_emb, \
                           ldf, coef, keys.current(), refs.current(), node_map, g_final, ldg, part_head, part_tail,        \
                           reg_w, 2.0f * (reg_scale * lambda));                                                             \
        hipLaunchKernelGGL(bpr_combine_kernel<V>, gc, dim3(kBlock), 0, s, batch, (int)d, keys.current(), part_head,         \
                           part_tail, g_final, ldg);                                                                        \
    } while (0)
        if (d <= 64) MI_BPR_GO(1);
        else if (d <= 128) MI_BPR_GO(2);
        else if (d <= 256) MI_BPR_GO(4);
        else MI_BPR_GO(8);
#undef MI_BPR_GO
    }
    hipLaunchKernelGGL(bpr_finish_kernel, dim3(1), dim3(1024), 0, s, batch, spv, rgv, inv_b, lambda, loss_out);
    return mi_launch_status();
}

int mi_adam_dense_f32(int64_t n_rows, int64_t d, float* p, int64_t ldp, const float* grad,
                      int64_t ldgr, float* m, float* v, const float* reg_w, double lr, double beta1,
                      double beta2, double eps, int64_t step, mi_stream_t stream) {
    MI_CHECK_ARG(n_rows >= 0 && d > 0 && step >= 1);
    if (n_rows == 0) return 0;
    MI_CHECK_ARG(p && grad && m && v);
    if (d % 4 != 0) return MI_ERR_UNSUPPORTED;
    MI_CHECK_ARG(ldp % 4 == 0 && ldgr % 4 == 0 && ldp >= d && ldgr >= d);
    MI_CHECK_ARG(mi_aligned16(p) && mi_aligned16(grad) && mi_aligned16(m) && mi_aligned16(v));
    // scalar constants in double, as torch.optim.Adam derives them, then rounded once to fp32
    const MiAdamConsts c = mi_adam_consts(lr, beta1, beta2, eps, step);
    const int64_t total = n_rows * (d / 4);
    int64_t blocks = mi_ceil_div(total, kBlock);
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, n_rows,
                       (int)(d / 4), reinterpret_cast<float4*>(p), ldp / 4,
                       reinterpret_cast<const float4*>(grad), ldgr / 4, reinterpret_cast<float4*>(m),
                       reinterpret_cast<float4*>(v), reg_w, c);
    return mi_launch_status();
}

}  // extern "C"

// N5: PinSAGE samplers on device (reference: pinsage/sampler.py:16-106 over DGL's random_walk /
// PinSAGESampler; semantics restated in oracle/pinsage_ref.py, which mirrors these kernels draw for draw).
//
//   mi_pinsage_item_pairs   heads uniform over items, tail = end of one item -> user -> item walk
//                           (uniform neighbour per step, -1 when the item has no users), negative tail uniform
//   mi_pinsage_neighbors    per seed: W walks of L traversals, termination probability p before every
//                           traversal but the first; items reached at traversal ends are counted and the T most
//                           visited (count desc, id asc) are returned with their visit counts
//
// One thread per head / per seed: a seed's whole walk set is W*L <= a few dozen draws.
#include "common.hpp"

namespace {

enum { P_HEAD = 11, P_NEG = 12, P_WALK = 13 };

__device__ __forceinline__ MiPhilox pw(uint32_t purpose, uint32_t a, uint32_t b, uint32_t c, uint64_t seed, uint64_t step) {
    const uint32_t c3 = (purpose & 0xFFu) | ((uint32_t)(step & 0xFFFFFFu) << 8);
    return mi_philox4x32(a, b, c, c3, (uint32_t)seed, (uint32_t)((seed >> 32) ^ (step >> 24)));
}

struct Bip {
    const int32_t* iu_ptr; const int32_t* iu_idx;  // item -> users
    const int32_t* ui_ptr; const int32_t* ui_idx;  // user -> items
};

__device__ __forceinline__ int32_t hop(const Bip& g, int32_t item, uint32_t w0, uint32_t w1) {
    const int32_t b = g.iu_ptr[item], du = g.iu_ptr[item + 1] - b;
    if (du == 0) return -1;
    const int32_t u = g.iu_idx[b + (int32_t)(w0 % (uint32_t)du)];
    const int32_t b2 = g.ui_ptr[u], di = g.ui_ptr[u + 1] - b2;
    if (di == 0) return -1;
    return g.ui_idx[b2 + (int32_t)(w1 % (uint32_t)di)];
}

__global__ void item_pairs_kernel(int64_t batch, int64_t n_items, Bip g, uint64_t seed, uint64_t step,
                                  int64_t* __restrict__ heads, int64_t* __restrict__ tails,
                                  int64_t* __restrict__ negs) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    MiPhilox w = pw(P_HEAD, (uint32_t)b, 0, 0, seed, step);
    const int32_t h = (int32_t)(w.c[0] % (uint32_t)n_items);
    heads[b] = h;
    tails[b] = hop(g, h, w.c[1], w.c[2]);
    negs[b] = pw(P_NEG, (uint32_t)b, 0, 0, seed, step).c[0] % (uint32_t)n_items;
}

__global__ void neighbors_kernel(int64_t n_seeds, const int64_t* __restrict__ seeds, Bip g, int walk_length,
                                 uint32_t restart_thr, int num_walks, int T, int layer, uint64_t seed, uint64_t step,
                                 int32_t* __restrict__ visit_ws, int64_t* __restrict__ nb, int64_t* __restrict__ wt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_seeds) return;
    const int32_t s = (int32_t)seeds[i];
    int32_t* vis = visit_ws + i * (int64_t)(num_walks * walk_length);
    int m = 0;
    for (int wk = 0; wk < num_walks; ++wk) {
        int32_t cur = s;
        for (int tr = 0; tr < walk_length; ++tr) {
            MiPhilox w = pw(P_WALK, (uint32_t)(wk * walk_length + tr), (uint32_t)layer, (uint32_t)s, seed, step);
            if (tr > 0 && w.c[2] < restart_thr) break;
            cur = hop(g, cur, w.c[0], w.c[1]);
            if (cur == -1) break;
            vis[m++] = cur;
        }
    }
    // sort visited ids ascending, then take the T best runs by (count desc, id asc)
    for (int a = 1; a < m; ++a) {
        int32_t v = vis[a];
        int b = a - 1;
        while (b >= 0 && vis[b] > v) { vis[b + 1] = vis[b]; --b; }
        vis[b + 1] = v;
    }
    for (int j = 0; j < T; ++j) {
        int32_t best = -1, best_cnt = 0;
        for (int a = 0; a < m;) {
            int e = a;
            while (e < m && vis[e] == vis[a]) ++e;
            const int cnt = e - a;
            if (vis[a] >= 0 && cnt > best_cnt) {  // strictly greater: among equal counts the smaller id (first run) wins
                best = vis[a];
                best_cnt = cnt;
            }
            a = e;
        }
        nb[i * T + j] = best;
        wt[i * T + j] = best_cnt;
        if (best < 0) continue;
        int k = 0;  // drop the consumed run; the rest stays sorted
        for (int a = 0; a < m; ++a)
            if (vis[a] != best) vis[k++] = vis[a];
        m = k;
    }
}

}  // namespace

extern "C" {

int mi_pinsage_item_pairs(int64_t batch, int64_t n_items, const int32_t* iu_ptr, const int32_t* iu_idx,
                          const int32_t* ui_ptr, const int32_t* ui_idx, uint64_t seed, uint64_t step, int64_t* heads,
                          int64_t* tails, int64_t* neg_tails, mi_stream_t stream) {
    MI_CHECK_ARG(batch >= 0 && n_items > 0);
    if (batch == 0) return 0;
    MI_CHECK_ARG(iu_ptr && iu_idx && ui_ptr && ui_idx && heads && tails && neg_tails);
    if (n_items >= INT32_MAX) return MI_ERR_TOO_LARGE;
    Bip g = {iu_ptr, iu_idx, ui_ptr, ui_idx};
    hipLaunchKernelGGL(item_pairs_kernel, dim3((unsigned)mi_ceil_div(batch, 256)), dim3(256), 0, (hipStream_t)stream, batch,
                       n_items, g, seed, step, heads, tails, neg_tails);
    return mi_launch_status();
}

size_t mi_pinsage_neighbors_workspace_bytes(int64_t n_seeds, int32_t walk_length, int32_t num_walks) {
    if (n_seeds <= 0 || walk_length <= 0 || num_walks <= 0) return 256;
    return mi_align_up((size_t)n_seeds * (size_t)walk_length * (size_t)num_walks * sizeof(int32_t), 256);
}

int mi_pinsage_neighbors(int64_t n_seeds, const int64_t* seeds, const int32_t* iu_ptr, const int32_t* iu_idx,
                         const int32_t* ui_ptr, const int32_t* ui_idx, int32_t walk_length, double restart_prob,
                         int32_t num_walks, int32_t num_neighbors, int32_t layer, uint64_t seed, uint64_t step,
                         int64_t* neighbors, int64_t* weights, void* ws, size_t ws_bytes, mi_stream_t stream) {
    MI_CHECK_ARG(n_seeds >= 0 && walk_length > 0 && num_walks > 0 && num_neighbors > 0);
    MI_CHECK_ARG(restart_prob >= 0.0 && restart_prob < 1.0);
    if (n_seeds == 0) return 0;
    MI_CHECK_ARG(seeds && iu_ptr && iu_idx && ui_ptr && ui_idx && neighbors && weights && ws);
    if (ws_bytes < mi_pinsage_neighbors_workspace_bytes(n_seeds, walk_length, num_walks)) return MI_ERR_WORKSPACE;
    Bip g = {iu_ptr, iu_idx, ui_ptr, ui_idx};
    const uint32_t thr = (uint32_t)(restart_prob * 4294967296.0);
    hipLaunchKernelGGL(neighbors_kernel, dim3((unsigned)mi_ceil_div(n_seeds, 64)), dim3(64), 0, (hipStream_t)stream, n_seeds,
                       seeds, g, walk_length, thr, num_walks, num_neighbors, layer, seed, step, static_cast<int32_t*>(ws),
                       neighbors, weights);
    return mi_launch_status();
}

}  // extern "C"

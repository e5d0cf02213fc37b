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

constexpr int kNbThreads = 256;
// w_pad lanes per seed (a power of two >= num_walks), seeds per block, dynamic LDS; false when the walks of one seed do not
// fit a block / the LDS budget (callers report MI_ERR_UNSUPPORTED: hundreds of walks per seed are not a PinSAGE setting)
static bool neighbors_launch_shape(int num_walks, int walk_length, int* w_pad, int* spb, size_t* lds) {
    int w = 1;
    while (w < num_walks) w <<= 1;
    if (w > kNbThreads) return false;
    *w_pad = w;
    *spb = kNbThreads / w;
    *lds = (size_t)(*spb) * (size_t)num_walks * (size_t)walk_length * sizeof(int32_t);
    return *lds <= 64 * 1024;
}

// One lane per (seed, walk): the walks of a seed are independent chains of dependent global loads (item -> user -> item per
// traversal), so they run side by side — one thread per seed walked them one after another: ~120 dependent loads, 98 us
// per launch at the reference's batch (round 3: two thirds of a training iteration's sampling).  A block serves
// blockDim / w_pad seeds (w_pad = num_walks rounded up to a power of two); visits go to the seed's slots in LDS (-1 = the
// walk ended before that traversal), then the seed's first lane sorts them and picks the T most visited.  Same Philox
// counters as before: the result is bit for bit the serial kernel's (and the mirror's, tests/test_pinsage.py).
__global__ void neighbors_kernel(int64_t n_seeds, const int32_t* __restrict__ n_seeds_dev, const int64_t* __restrict__ seeds,
                                 Bip g, int walk_length, uint32_t restart_thr, int num_walks, int T, int layer, uint64_t seed,
                                 uint64_t step, int w_pad, int64_t* __restrict__ nb, int64_t* __restrict__ wt) {
    extern __shared__ int32_t nvis[];   // [seeds per block][num_walks * walk_length]
    const int spb = blockDim.x / w_pad;
    const int ls = threadIdx.x / w_pad, wk = threadIdx.x % w_pad;
    const int64_t i = (int64_t)blockIdx.x * spb + ls;
    if (n_seeds_dev) n_seeds = min(n_seeds, (int64_t)*n_seeds_dev);  // launched over an upper bound, the count is on the device
    const bool live = ls < spb && i < n_seeds;
    const int slots = num_walks * walk_length;
    int32_t* vis = nvis + (ls < spb ? ls : 0) * slots;
    const int32_t s = live ? (int32_t)seeds[i] : 0;
    if (live && wk < num_walks) {
        int32_t cur = s;
        int tr = 0;
        for (; tr < walk_length; ++tr) {
            MiPhilox w = pw(P_WALK, (uint32_t)(wk * walk_length + tr), (uint32_t)layer, (uint32_t)s, seed, step);
            if (tr > 0 && w.c[2] < restart_thr) break;
            cur = hop(g, cur, w.c[0], w.c[1]);
            if (cur == -1) break;
            vis[wk * walk_length + tr] = cur;
        }
        for (; tr < walk_length; ++tr) vis[wk * walk_length + tr] = -1;
    }
    __syncthreads();
    if (!live || wk != 0) return;
    int m = slots;   // the -1 slots sort first and are never picked
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


// ---- whole-batch construction on the device (round 3) -----------------------------------------------------------------
// pinsage/sampler.py:93-106 (sample_from_item_pairs) and :73-91 (sample_blocks) are a handful of dgl calls over a few
// hundred nodes: as torch index ops on the GPU they were ~25 launches and five host read-backs per layer (boolean
// indexing, unique).  Here the seeds of a batch and every block are built by ONE workgroup each, in LDS: bitonic sorts,
// block scans, binary searches.  The host reads the counts back once per batch.
constexpr int kBT = 1024;          // threads of the single-workgroup kernels
constexpr int kSeedPairsMax = 1024;   // batch (pairs) limit of pinsage_seeds_kernel
constexpr int kBlockEdgesMax = 16384; // n * T limit of pinsage_block_kernel (two int32 arrays of that length in LDS)

template <typename T>
__device__ __forceinline__ void lds_bitonic_sort(T* a, int n_pow2) {
    for (int k = 2; k <= n_pow2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n_pow2; i += kBT) {
                const int x = i ^ j;
                if (x > i) {
                    const T u = a[i], v = a[x];
                    if ((u > v) == ((i & k) == 0)) { a[i] = v; a[x] = u; }
                }
            }
            __syncthreads();
        }
}

// exclusive scan of a[0..n) in place; returns the total (to every thread).  part: LDS int32[kBT + 1].
__device__ __forceinline__ int lds_exclusive_scan(int32_t* a, int n, int32_t* part) {
    const int per = (n + kBT - 1) / kBT;
    const int lo = min(n, (int)threadIdx.x * per), hi = min(n, lo + per);
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += a[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < kBT; off <<= 1) {   // Hillis-Steele inclusive scan of the 1024 partials
        const int v = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    const int total = part[kBT - 1];
    int run = threadIdx.x ? part[threadIdx.x - 1] : 0;
    for (int i = lo; i < hi; ++i) { const int v = a[i]; a[i] = run; run += v; }
    __syncthreads();
    return total;
}

template <typename T>
__device__ __forceinline__ int lds_lower_bound(const T* a, int n, T key) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ int next_pow2(int n) { int p = 2; while (p < n) p <<= 1; return p; }

// seeds = sorted unique of the heads / tails / negative tails of the pairs whose walk survived; pos_u / pos_v / neg_v:
// positions of every surviving pair's three items in `seeds` (pairs in their original order); banned: the sorted keys
// head * n_items + tail and head * n_items + neg (the label pairs no frontier edge may repeat).
// counts[0] = surviving pairs, counts[1] = seeds.
__global__ __launch_bounds__(kBT) void pinsage_seeds_kernel(int batch, int64_t n_items, const int64_t* __restrict__ heads,
                                                            const int64_t* __restrict__ tails, const int64_t* __restrict__ negs,
                                                            int64_t* __restrict__ seeds, int64_t* __restrict__ pos_u,
                                                            int64_t* __restrict__ pos_v, int64_t* __restrict__ neg_v,
                                                            int64_t* __restrict__ banned, int32_t* __restrict__ counts) {
    __shared__ int32_t S[4 * kSeedPairsMax];      // the 3B item ids, sorted
    __shared__ int32_t U[4 * kSeedPairsMax];      // flags / scan, then the unique ids
    __shared__ int64_t K[2 * kSeedPairsMax];      // banned keys
    __shared__ int32_t cidx[kSeedPairsMax];
    __shared__ int32_t part[kBT + 1];
    const int tid = threadIdx.x;
    const int n3 = next_pow2(3 * batch), n2 = next_pow2(2 * batch);
    for (int i = tid; i < n3; i += kBT) S[i] = INT32_MAX;
    for (int i = tid; i < n2; i += kBT) K[i] = INT64_MAX;
    for (int i = tid; i < batch; i += kBT) cidx[i] = tails[i] != -1 ? 1 : 0;
    __syncthreads();
    const int n_pairs = lds_exclusive_scan(cidx, batch, part);
    for (int i = tid; i < batch; i += kBT) {
        if (tails[i] == -1) continue;
        const int64_t h = heads[i], tl = tails[i], g = negs[i];
        S[3 * i] = (int32_t)h; S[3 * i + 1] = (int32_t)tl; S[3 * i + 2] = (int32_t)g;
        K[2 * i] = h * n_items + tl;
        K[2 * i + 1] = h * n_items + g;
    }
    __syncthreads();
    lds_bitonic_sort(S, n3);
    lds_bitonic_sort(K, n2);
    for (int i = tid; i < n3; i += kBT) U[i] = (S[i] != INT32_MAX && (i == 0 || S[i] != S[i - 1])) ? 1 : 0;
    __syncthreads();
    const int n_seeds = lds_exclusive_scan(U, n3, part);
    // unique ids to global and, compacted, back into S's place via a temporary pass over U's positions
    for (int i = tid; i < n3; i += kBT) {
        const bool first = S[i] != INT32_MAX && (i == 0 || S[i] != S[i - 1]);
        if (first) seeds[U[i]] = S[i];
    }
    __syncthreads();
    __threadfence_block();
    for (int i = tid; i < n_seeds; i += kBT) U[i] = (int32_t)seeds[i];   // (this workgroup's own global writes, after the barrier)
    __syncthreads();
    for (int i = tid; i < batch; i += kBT) {
        if (tails[i] == -1) continue;
        const int c = cidx[i];
        pos_u[c] = lds_lower_bound(U, n_seeds, (int32_t)heads[i]);
        pos_v[c] = lds_lower_bound(U, n_seeds, (int32_t)tails[i]);
        neg_v[c] = lds_lower_bound(U, n_seeds, (int32_t)negs[i]);
    }
    for (int i = tid; i < 2 * n_pairs; i += kBT) banned[i] = K[i];
    if (tid == 0) { counts[0] = n_pairs; counts[1] = n_seeds; }
}

struct PinBlockOut {
    int64_t* src_ids;      // [n_max * (1 + T)]: the block's nodes, destination nodes (= seeds) first, new sources ascending
    int64_t* edge_src;     // [n_max * T] block-local source of every kept frontier edge, destination-major order
    int64_t* edge_dst;
    float* weights;        // visit counts
    int32_t* dst_rowptr;   // [n_max + 1]   CSR by destination, columns (local sources) ascending,
    int32_t* dst_col;      //               values w / max(sum of the destination's w, 1): WeightedSAGEConv's mean
    float* dst_val;
    int32_t* src_rowptr;   // [n_max * (1 + T) + 1]  its transpose (CSR by source, destinations ascending)
    int32_t* src_col;
    float* src_val;
    float* val_tmp;        // [n_max * T] scratch
    int32_t* counts;       // [2]: nodes of the block (= seeds of the next layer), kept edges
};

// One block (pinsage/sampler.py:73-91): frontier edges nbr[i, j] -> seeds[i] with weight cnt[i, j]; edges that repeat a
// label pair removed; dgl.to_block numbering.  pos: int32[n_items] scratch, all -1 on entry and on exit.
__global__ __launch_bounds__(kBT) void pinsage_block_kernel(int n_max, const int32_t* __restrict__ n_dev, int T, int64_t n_items,
                                                            const int64_t* __restrict__ seeds, const int64_t* __restrict__ nb,
                                                            const int64_t* __restrict__ wt, const int64_t* __restrict__ banned,
                                                            const int32_t* __restrict__ n_pairs_dev, int32_t* __restrict__ pos,
                                                            PinBlockOut o) {
    extern __shared__ int32_t lds[];
    int32_t* keep = lds;                        // [E_pow2] flags -> exclusive positions
    int32_t* keys = lds + kBlockEdgesMax;       // [E_pow2] sort keys
    __shared__ int32_t part[kBT + 1];
    const int tid = threadIdx.x;
    const int n = min(n_max, *n_dev);
    const int E = n * T, EP = next_pow2(max(E, 2));
    const int n_ban = n_pairs_dev ? 2 * *n_pairs_dev : 0;
    for (int i = tid; i < n; i += kBT) {
        pos[seeds[i]] = i;
        o.src_ids[i] = seeds[i];
    }
    __syncthreads();
    __threadfence_block();
    for (int e = tid; e < EP; e += kBT) {
        int k = 0;
        int32_t cand = INT32_MAX;
        if (e < E) {
            const int64_t v = nb[e];
            if (v >= 0) {
                k = 1;
                if (n_ban) {   // frontier edge v -> s is dropped when (v, s) is a label pair
                    const int64_t key = v * n_items + seeds[e / T];
                    int lo = 0, hi = n_ban;
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if (banned[mid] < key) lo = mid + 1; else hi = mid; }
                    if (lo < n_ban && banned[lo] == key) k = 0;
                }
                if (k && pos[v] < 0) cand = (int32_t)v;
            }
        }
        keep[e] = k;
        keys[e] = cand;
    }
    __syncthreads();
    lds_bitonic_sort(keys, EP);
    // new sources: unique ascending; their block ids follow the seeds'
    int my_new = 0;
    for (int i = tid; i < EP; i += kBT) my_new += (keys[i] != INT32_MAX && (i == 0 || keys[i] != keys[i - 1])) ? 1 : 0;
    // positions of the unique entries: a second scan array would not fit; count with a block scan over per-thread totals of
    // STRIDED ownership is not order-preserving, so the flags are scanned in place in `part`-sized chunks instead:
    // each thread owns a CONTIGUOUS chunk of the sorted array
    {
        const int per = (EP + kBT - 1) / kBT;
        const int lo = min(EP, tid * per), hi = min(EP, lo + per);
        int cnt = 0;
        for (int i = lo; i < hi; ++i) cnt += (keys[i] != INT32_MAX && (i == 0 || keys[i] != keys[i - 1])) ? 1 : 0;
        part[tid] = cnt;
        __syncthreads();
        for (int off = 1; off < kBT; off <<= 1) {
            const int v = tid >= off ? part[tid - off] : 0;
            __syncthreads();
            part[tid] += v;
            __syncthreads();
        }
        int run = tid ? part[tid - 1] : 0;
        for (int i = lo; i < hi; ++i)
            if (keys[i] != INT32_MAX && (i == 0 || keys[i] != keys[i - 1])) {
                o.src_ids[n + run] = keys[i];
                pos[keys[i]] = n + run;
                ++run;
            }
    }
    (void)my_new;
    const int n_new = part[kBT - 1];
    const int n_src = n + n_new;
    __syncthreads();
    __threadfence_block();
    // kept edges, compacted in destination-major order
    const int n_edges = lds_exclusive_scan(keep, EP, part);   // keep[e] = position of edge e if kept
    for (int e = tid; e < E; e += kBT) {
        const int k = keep[e];
        const int k_next = e + 1 < EP ? keep[e + 1] : n_edges;
        if (k_next == k) continue;   // not kept
        o.edge_src[k] = pos[nb[e]];
        o.edge_dst[k] = e / T;
        o.weights[k] = (float)wt[e];
    }
    // CSR by destination: row d = its kept edges sorted by local source; values = w / max(sum w, 1)
    for (int d = tid; d <= n; d += kBT) o.dst_rowptr[d] = d < n ? keep[d * T] : n_edges;
    for (int d = tid; d < n; d += kBT) {
        const int k0 = keep[d * T];
        const int k1 = (d + 1) * T < EP ? keep[(d + 1) * T] : n_edges;
        const int m = k1 - k0;
        if (m == 0) continue;
        int32_t cs[16];
        float ws[16];
        int j = 0;
        float sum = 0.f;
        for (int q = 0; q < T; ++q) {
            const int e = d * T + q;
            const int k = keep[e], kn = e + 1 < EP ? keep[e + 1] : n_edges;
            if (kn == k) continue;
            const int32_t c = pos[nb[e]];
            const float w = (float)wt[e];
            sum += w;
            int b = j - 1;                      // insertion by column
            while (b >= 0 && cs[b] > c) { cs[b + 1] = cs[b]; ws[b + 1] = ws[b]; --b; }
            cs[b + 1] = c; ws[b + 1] = w;
            ++j;
        }
        const float den = fmaxf(sum, 1.0f);
        for (int q = 0; q < m; ++q) {
            o.dst_col[k0 + q] = cs[q];
            o.dst_val[k0 + q] = ws[q] / den;
        }
        // the same normalised weights in edge order (for the transpose)
        int q2 = 0;
        for (int q = 0; q < T; ++q) {
            const int e = d * T + q;
            const int k = keep[e], kn = e + 1 < EP ? keep[e + 1] : n_edges;
            if (kn == k) continue;
            o.val_tmp[k0 + q2] = (float)wt[e] / den;
            ++q2;
        }
    }
    __syncthreads();
    __threadfence_block();
    // CSR by source: sort (local source, edge position); edge positions ascend with the destination
    for (int k = tid; k < EP; k += kBT) keys[k] = k < n_edges ? (int32_t)((o.edge_src[k] << 14) | k) : INT32_MAX;
    __syncthreads();
    lds_bitonic_sort(keys, EP);
    for (int j = tid; j < n_edges; j += kBT) {
        const int k = keys[j] & 0x3FFF;
        o.src_col[j] = (int32_t)o.edge_dst[k];
        o.src_val[j] = o.val_tmp[k];
    }
    for (int r = tid; r <= n_src; r += kBT) o.src_rowptr[r] = r < n_src ? lds_lower_bound(keys, n_edges, (int32_t)(r << 14)) : n_edges;
    // scratch back to all -1
    for (int i = tid; i < n_src; i += kBT) pos[o.src_ids[i]] = -1;
    if (tid == 0) { o.counts[0] = n_src; o.counts[1] = n_edges; }
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
    int w_pad, spb;
    size_t lds;
    if (!neighbors_launch_shape(num_walks, walk_length, &w_pad, &spb, &lds)) return MI_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(neighbors_kernel, dim3((unsigned)mi_ceil_div(n_seeds, spb)), dim3(kNbThreads), lds, (hipStream_t)stream,
                       n_seeds, (const int32_t*)nullptr, seeds, g, walk_length, thr, num_walks, num_neighbors, layer, seed, step,
                       w_pad, neighbors, weights);
    return mi_launch_status();
}


size_t mi_pinsage_batch_workspace_bytes(int64_t batch, int32_t walk_length, int32_t num_walks, int32_t num_neighbors,
                                        int32_t num_layers) {
    if (batch <= 0 || walk_length <= 0 || num_walks <= 0 || num_neighbors <= 0 || num_layers <= 0) return 256;
    int64_t n = 3 * batch;
    size_t total = 0;
    for (int l = 0; l < num_layers; ++l) {
        total += mi_align_up((size_t)n * walk_length * num_walks * sizeof(int32_t), 256);   // visit lists
        total += 2 * mi_align_up((size_t)n * num_neighbors * sizeof(int64_t), 256);           // neighbours, visit counts
        total += mi_align_up((size_t)n * num_neighbors * sizeof(float), 256);                 // val_tmp
        n *= (1 + num_neighbors);
    }
    total += mi_align_up((size_t)2 * batch * sizeof(int64_t), 256) + 3 * mi_align_up((size_t)batch * sizeof(int64_t), 256);
    return total;
}

int mi_pinsage_sample_batch(const mi_pinsage_batch_desc* d, uint64_t seed, uint64_t step, const mi_pinsage_batch_out* out,
                            void* ws, size_t ws_bytes, mi_stream_t stream) {
    MI_CHECK_ARG(d && out && ws);
    const int64_t B = d->batch;
    const int T = d->num_neighbors, NL = d->num_layers;
    MI_CHECK_ARG(B > 0 && d->n_items > 0 && d->walk_length > 0 && d->num_walks > 0 && T > 0 && NL > 0 && NL <= MI_PINSAGE_MAX_LAYERS);
    MI_CHECK_ARG(d->restart_prob >= 0.0 && d->restart_prob < 1.0);
    MI_CHECK_ARG(d->iu_ptr && d->iu_idx && d->ui_ptr && d->ui_idx && d->pos_scratch);
    MI_CHECK_ARG(out->seeds && out->pos_u && out->pos_v && out->neg_v && out->counts);
    if (d->n_items >= INT32_MAX) return MI_ERR_TOO_LARGE;
    if (B > kSeedPairsMax || T > 16) return MI_ERR_UNSUPPORTED;
    {   // every layer's n_max * T must fit the single-workgroup block kernel
        int64_t n = 3 * B;
        for (int l = 0; l < NL; ++l) {
            if (n * T > kBlockEdgesMax || n * (1 + T) >= (1 << 17)) return MI_ERR_UNSUPPORTED;
            n *= (1 + T);
        }
    }
    if (ws_bytes < mi_pinsage_batch_workspace_bytes(B, d->walk_length, d->num_walks, T, NL)) return MI_ERR_WORKSPACE;
    int nb_w_pad, nb_spb;
    size_t nb_lds;
    if (!neighbors_launch_shape(d->num_walks, d->walk_length, &nb_w_pad, &nb_spb, &nb_lds)) return MI_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    MiArena ar(ws, ws_bytes);
    int64_t* heads = ar.take<int64_t>((size_t)B);
    int64_t* tails = ar.take<int64_t>((size_t)B);
    int64_t* negs = ar.take<int64_t>((size_t)B);
    int64_t* banned = ar.take<int64_t>((size_t)2 * B);
    Bip g = {d->iu_ptr, d->iu_idx, d->ui_ptr, d->ui_idx};
    hipLaunchKernelGGL(item_pairs_kernel, dim3((unsigned)mi_ceil_div(B, 256)), dim3(256), 0, s, B, d->n_items, g, seed, step, heads,
                       tails, negs);
    hipLaunchKernelGGL(pinsage_seeds_kernel, dim3(1), dim3(kBT), 0, s, (int)B, d->n_items, heads, tails, negs, out->seeds, out->pos_u,
                       out->pos_v, out->neg_v, banned, out->counts);
    static bool attr_set = false;
    const size_t lds = (size_t)2 * kBlockEdgesMax * sizeof(int32_t);
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(pinsage_block_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return MI_ERR_UNSUPPORTED;
        attr_set = true;
    }
    const uint32_t thr = (uint32_t)(d->restart_prob * 4294967296.0);
    int64_t n_max = 3 * B;
    const int64_t* seeds = out->seeds;
    const int32_t* n_dev = out->counts + 1;
    for (int l = 0; l < NL; ++l) {
        const mi_pinsage_block_out& bo = out->blocks[l];
        MI_CHECK_ARG(bo.src_ids && bo.edge_src && bo.edge_dst && bo.weights && bo.dst_rowptr && bo.dst_col && bo.dst_val &&
                     bo.src_rowptr && bo.src_col && bo.src_val);
        int32_t* vis = ar.take<int32_t>((size_t)n_max * d->walk_length * d->num_walks);
        int64_t* nb = ar.take<int64_t>((size_t)n_max * T);
        int64_t* wt = ar.take<int64_t>((size_t)n_max * T);
        float* val_tmp = ar.take<float>((size_t)n_max * T);
        if (!vis || !nb || !wt || !val_tmp) return MI_ERR_WORKSPACE;
        (void)vis;
        hipLaunchKernelGGL(neighbors_kernel, dim3((unsigned)mi_ceil_div(n_max, nb_spb)), dim3(kNbThreads), nb_lds, s, n_max, n_dev,
                           seeds, g, d->walk_length, thr, d->num_walks, T, l, seed, step, nb_w_pad, nb, wt);
        PinBlockOut o = {bo.src_ids, bo.edge_src, bo.edge_dst, bo.weights, bo.dst_rowptr, bo.dst_col, bo.dst_val,
                         bo.src_rowptr, bo.src_col, bo.src_val, val_tmp, out->counts + 2 + 2 * l};
        hipLaunchKernelGGL(pinsage_block_kernel, dim3(1), dim3(kBT), lds, s, (int)n_max, n_dev, T, d->n_items, seeds, nb, wt, banned,
                           out->counts, d->pos_scratch, o);
        seeds = bo.src_ids;
        n_dev = out->counts + 2 + 2 * l;
        n_max *= (1 + T);
    }
    return mi_launch_status();
}

}  // extern "C"

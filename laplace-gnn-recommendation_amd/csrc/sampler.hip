// N1: on-device N-hop subgraph sampler for the ranker — the algorithm of the reference's live sampler
// `GraphDataset.__getitem__` (data/dataset.py:39-309, train mode) for a whole batch of seed users at
// once, with no host round trip except one read-back of four totals to size the outputs.
//
// Per seed user (one workgroup per sample in most phases):
//   seed     positives with replacement, negatives (fast path: uniform in [0, id_max); exact path for
//            tiny graphs), hop-0 article cut                                   data/dataset.py:42-106,189-230
//   expand   per hop: mark the users of the <= n queued articles in a per-sample bitmap (atomicOr),
//            drop the explored ones, popcount-scan, draw a uniform n-subset of the set bits (Floyd on
//            ranks, rank -> id by select), cut the next article queue          data/dataset.py:258-293
//   relabel  article bitmap + popcount prefix = sorted-unique buckets; users ranked directly
//            (bucketize, data/dataset.py:134-150,233-241)
//   emit     collated batch (disjoint union of the samples, PyG collate semantics): global node ids,
//            edge_index, edge_label_index, edge_label, per-sample node offsets
//
// Every random choice is a counter-based Philox4x32-10 draw keyed on (seed, step) and counted by
// (purpose, seed user, i, j): oracle/sampler_ref.py mirrors it bit for bit (integer work => exact parity).
#include "common.hpp"

namespace {

enum { P_POS = 1, P_NEG = 2, P_ART_CUT = 3, P_USER_CUT = 4, P_NEG_EXACT = 5, P_USER_REJ = 6 };
constexpr int kRejectCap = 1 << 16;  // draws; unreachable when reject_min respects its bound (see header)
constexpr int kMaxFan = 1024;       // num_neighbors cap of this implementation
constexpr int kSelThreads = 1024;

struct Smp {
    int32_t B, H, n;
    int64_t U, A, E, id_max;
    const int32_t* uptr; const int32_t* uidx;
    const int32_t* aptr; const int32_t* aidx;
    const int32_t* cptr; const int32_t* cidx;  // evaluation mode: matcher candidates per user (null = train mode)
    const int64_t* seeds;
    double pos_ratio, neg_ratio;
    int32_t k, randomization;
    uint64_t seed, step;
    int32_t max_pos, max_neg;
    uint32_t* bm_users; int32_t WU;
    uint32_t* bm_art;   int32_t WA;
    int32_t* pre_a;
    int32_t* uq; int32_t* uq_n; int32_t* uq_local; int32_t* uq_estart;
    int32_t* uq_rstart;  // edge start of a user's row when the sample's users are laid out in RANK order (CSR emit)
    int32_t* aq; int32_t* aq_n;
    int64_t* aq_L;      // [B] total length of the queued articles' user lists
    int64_t reject_min; // above this the next frontier is drawn by rejection instead of being materialised
    int32_t* pos_items; int32_t* neg_items; int32_t* n_pos; int32_t* n_neg;
    int32_t* cnt;   // [B][4]
    int32_t* off;   // [4][B+1], then the four totals again, contiguous (one copy to the host)
    int32_t* banned;    // [B][max_pos] exact negative path: the sorted distinct sampled positives
    int32_t* csr_long;  // [B] CSR emit: the sample has an article row too long for the one-wavefront row sort
};

__device__ __forceinline__ uint64_t rand_below(uint64_t m, uint32_t purpose, uint32_t seed_user, uint32_t i,
                                               uint32_t j, uint64_t seed, uint64_t step) {
    const uint32_t c3 = (purpose & 0xFFu) | ((uint32_t)(step & 0xFFFFFFu) << 8);
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)((seed >> 32) ^ (step >> 24));
    MiPhilox r = mi_philox4x32(i, j, seed_user, c3, k0, k1);
    return (((uint64_t)r.c[0] << 32) | r.c[1]) % m;
}

// Floyd: n distinct positions of [0, L), ascending, into out[] (all of [0, L) when L <= n). One thread.
__device__ int floyd_subset(int64_t L, int n, uint32_t purpose, uint32_t seed_user, uint32_t i, uint64_t seed,
                            uint64_t step, int32_t* out) {
    if (L <= n) {
        for (int t = 0; t < (int)L; ++t) out[t] = t;
        return (int)L;
    }
    int c = 0;
    for (int64_t j = L - n; j < L; ++j) {
        int32_t t = (int32_t)rand_below((uint64_t)(j + 1), purpose, seed_user, i, (uint32_t)j, seed, step);
        bool seen = false;
        for (int q = 0; q < c; ++q) seen |= (out[q] == t);
        out[c++] = seen ? (int32_t)j : t;
    }
    for (int a = 1; a < c; ++a) {  // insertion sort, ascending
        int32_t v = out[a];
        int b = a - 1;
        while (b >= 0 && out[b] > v) { out[b + 1] = out[b]; --b; }
        out[b + 1] = v;
    }
    return c;
}

__device__ __forceinline__ int s_of(const Smp& p, const int32_t* aq_n_ptr) { return (int)(aq_n_ptr - p.aq_n); }

// Ascending in-place sort of n DISTINCT values in LDS by rank (every thread places the values it owns): no serial
// insertion sort.  tmp: LDS scratch of n entries.  Whole block; ends with a barrier.
__device__ __forceinline__ void block_rank_sort_distinct(int32_t* a, int n, int32_t* tmp) {
    for (int q = threadIdx.x; q < n; q += blockDim.x) tmp[q] = a[q];
    __syncthreads();
    for (int q = threadIdx.x; q < n; q += blockDim.x) {
        const int32_t v = tmp[q];
        int r = 0;
        for (int x = 0; x < n; ++x) r += tmp[x] < v ? 1 : 0;
        a[r] = v;
    }
    __syncthreads();
}

// Floyd's subset by the whole block: the same draws and the same result as floyd_subset (draw j is Philox(j) and the
// commit order is j ascending), but the n Philox draws are made in parallel, the "already drawn?" test of draw j is a
// parallel compare over the j earlier picks, and the sort is by rank.  out / tmp: LDS, n entries each.  Returns the count
// to every thread; ends with a barrier.
__device__ int floyd_subset_block(int64_t L, int n, uint32_t purpose, uint32_t seed_user, uint32_t i, uint64_t seed,
                                  uint64_t step, int32_t* out, int32_t* tmp) {
    const int tid = threadIdx.x;
    if (L <= n) {
        for (int t = tid; t < (int)L; t += blockDim.x) out[t] = t;
        __syncthreads();
        return (int)L;
    }
    for (int c = tid; c < n; c += blockDim.x) {
        const int64_t j = L - n + c;
        tmp[c] = (int32_t)rand_below((uint64_t)(j + 1), purpose, seed_user, i, (uint32_t)j, seed, step);
    }
    __syncthreads();
    // commit in order, by wavefront 0 alone (no barriers inside the loop: LDS operations of one wavefront execute in program
    // order): pick c is tmp[c] unless one of the picks 0..c-1 already is that value — then it is j itself, which no earlier
    // pick can be (earlier draws are <= their own j < this j)
    if (tid < MI_WAVE) {
        volatile int32_t* vo = out;
        for (int c = 0; c < n; ++c) {
            const int32_t t = tmp[c];
            bool seen = false;
            for (int q = tid; q < c; q += MI_WAVE) seen |= (vo[q] == t);
            const bool any = __ballot(seen) != 0ull;
            if (tid == 0) vo[c] = any ? (int32_t)(L - n + c) : t;
        }
    }
    __syncthreads();
    block_rank_sort_distinct(out, n, tmp);
    return n;
}

// positions (ascending) of the concatenated article lists of `users[0..nu)` -> article ids.  Whole block (round 3: one
// thread did all of it — n Philox draws, an O(n^2) membership test, an insertion sort and 3 n dependent global loads:
// 108 us of smp_seed_kernel at fan-out 64).  sel / tmp: LDS int32[kMaxFan]; pre: LDS int64[kMaxFan + 1].
__device__ void cut_articles_block(const Smp& p, const int32_t* users, int nu, uint32_t seed_user, int hop, int32_t* sel,
                                   int32_t* tmp, int64_t* pre, int32_t* aq_out, int32_t* aq_n_out) {
    const int tid = threadIdx.x;
    for (int t = tid; t < nu; t += blockDim.x) tmp[t] = p.uptr[users[t] + 1] - p.uptr[users[t]];
    __syncthreads();
    if (tid == 0) {
        int64_t run = 0;
        for (int t = 0; t < nu; ++t) { pre[t] = run; run += tmp[t]; }
        pre[nu] = run;
    }
    __syncthreads();
    const int64_t L = pre[nu];
    const int c = floyd_subset_block(L, p.n, P_ART_CUT, seed_user, (uint32_t)hop, p.seed, p.step, sel, tmp);
    int64_t mine = 0;
    for (int q = tid; q < c; q += blockDim.x) {
        const int64_t pos = sel[q];
        int lo = 0, hi = nu;   // last t with pre[t] <= pos
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (pre[mid] <= pos) lo = mid; else hi = mid;
        }
        const int32_t a = p.uidx[p.uptr[users[lo]] + (pos - pre[lo])];
        aq_out[q] = a;
        mine += p.aptr[a + 1] - p.aptr[a];
    }
    // total length of the queued articles' user lists: integer sum, any order
    __syncthreads();
    int64_t* acc = pre;   // pre is dead (every thread is past its binary searches)
    if (tid == 0) acc[0] = 0;
    __syncthreads();
    if (mine) atomicAdd(reinterpret_cast<unsigned long long*>(acc), (unsigned long long)mine);
    __syncthreads();
    if (tid == 0) {
        *aq_n_out = c;
        p.aq_L[s_of(p, aq_n_out)] = acc[0];
    }
    __syncthreads();
}

// ---- phase 1: label edges + hop-0 article cut -------------------------------------------------
__global__ __launch_bounds__(256) void smp_seed_kernel(Smp p) {
    __shared__ int32_t sel[kMaxFan];
    __shared__ int32_t cut_tmp[kMaxFan];
    __shared__ int64_t cut_pre[kMaxFan + 1];
    __shared__ int32_t sh_npos, sh_nneg, sh_fast;
    const int s = blockIdx.x, tid = threadIdx.x;
    const int32_t u = (int32_t)p.seeds[s];
    const int32_t beg = p.uptr[u], deg = p.uptr[u + 1] - beg;
    int32_t* pos_items = p.pos_items + (int64_t)s * p.max_pos;
    int32_t* neg_items = p.neg_items + (int64_t)s * p.max_neg;
    if (tid == 0) {
        int npos = 0, nneg = 0, fast = 1;
        if (deg > 0) {
            if (p.randomization) {
                const int64_t cut = (int64_t)floor((double)deg * p.pos_ratio);
                npos = (int)(cut < 1 ? 1 : cut);
                const double ratio = npos <= 1 ? (double)(p.k - 1) : p.neg_ratio;
                nneg = (int)(ratio * (double)npos);
                fast = (nneg == 0) || ((double)p.E / (double)nneg > 100.0);
            } else {
                npos = 2;
                nneg = 1;
            }
        }
        if (npos > p.max_pos) npos = p.max_pos;  // capacities are sized from the max degree: cannot trigger
        if (nneg > p.max_neg) nneg = p.max_neg;
        sh_npos = npos; sh_nneg = nneg; sh_fast = fast;
        p.n_pos[s] = npos;
        p.uq[((int64_t)s * p.H) * p.n] = u;
        p.uq_n[(int64_t)s * p.H] = 1;
        for (int h = 1; h < p.H; ++h) p.uq_n[(int64_t)s * p.H + h] = 0;
        p.aq_n[s] = 0;
        p.aq_L[s] = 0;
    }
    __syncthreads();
    const int npos = sh_npos, nneg = sh_nneg;
    if (p.randomization) {
        for (int i = tid; i < npos; i += blockDim.x)
            pos_items[i] = p.uidx[beg + (int32_t)rand_below((uint64_t)deg, P_POS, (uint32_t)u, (uint32_t)i, 0, p.seed, p.step)];
        if (sh_fast && !p.cptr) {
            for (int i = tid; i < nneg; i += blockDim.x)
                neg_items[i] = (int32_t)rand_below((uint64_t)p.id_max, P_NEG, (uint32_t)u, (uint32_t)i, 0, p.seed, p.step);
        }
    } else if (tid == 0 && deg > 0) {  // argmin / argmax (first occurrence), negative = id_max
        int amin = 0, amax = 0;
        for (int i = 1; i < deg; ++i) {
            if (p.uidx[beg + i] < p.uidx[beg + amin]) amin = i;
            if (p.uidx[beg + i] > p.uidx[beg + amax]) amax = i;
        }
        pos_items[0] = p.uidx[beg + amin];
        pos_items[1] = p.uidx[beg + amax];
        neg_items[0] = (int32_t)p.id_max;
    }
    __syncthreads();
    if (p.cptr) {
        // evaluation mode: label-0 ids = symmetric difference of the distinct candidates and the user's items, ascending.
        // The sample's article bitmap (all zero here) is the scratch: OR the candidates in, XOR the (distinct) items,
        // enumerate the set bits in order, clear them again.
        __shared__ int32_t wscan[256 + 1];
        uint32_t* bm = p.bm_art + (int64_t)s * p.WA;
        for (int32_t q = p.cptr[u] + tid; q < p.cptr[u + 1]; q += blockDim.x) {
            const int32_t c = p.cidx[q];
            if (c >= 0 && c < p.A) atomicOr(bm + (c >> 5), 1u << (c & 31));
        }
        __syncthreads();
        for (int i = tid; i < deg; i += blockDim.x) {
            const int32_t v = p.uidx[beg + i];
            atomicXor(bm + (v >> 5), 1u << (v & 31));
        }
        __syncthreads();
        int base = 0;
        for (int w0 = 0; w0 < p.WA; w0 += blockDim.x) {
            const int w = w0 + tid;
            uint32_t word = w < p.WA ? bm[w] : 0u;
            // block-wide exclusive scan of the popcounts of this slab of words
            wscan[tid] = __popc(word);
            __syncthreads();
            if (tid == 0) {
                int run = 0;
                for (int x = 0; x < (int)blockDim.x; ++x) { const int c = wscan[x]; wscan[x] = run; run += c; }
                wscan[blockDim.x] = run;
            }
            __syncthreads();
            int r = base + wscan[tid];
            while (word) {
                const int b = __ffs(word) - 1;
                if (r < p.max_neg) neg_items[r] = w * 32 + b;
                ++r;
                word &= word - 1;
            }
            if (w < p.WA) bm[w] = 0u;
            base += wscan[blockDim.x];
            __syncthreads();
        }
        if (tid == 0) p.n_neg[s] = deg > 0 ? (base < p.max_neg ? base : p.max_neg) : 0;
        if (p.H >= 2 && deg > 0)   // block-uniform
            cut_articles_block(p, &u, 1, (uint32_t)u, 0, sel, cut_tmp, cut_pre, p.aq + (int64_t)s * p.n, p.aq_n + s);
        return;
    }
    if (tid == 0) {
        int nn = nneg;
        if (p.randomization && !sh_fast) {
            // exact path (tiny graphs): uniform subset of {0..id_max} minus the distinct sampled positives.
            int32_t* banned = p.banned + (int64_t)s * p.max_pos;  // sorted distinct sampled positives
            int nb = 0;
            for (int i = 0; i < npos; ++i) {
                const int32_t v = pos_items[i];
                if (v > p.id_max) continue;
                int q = 0;
                while (q < nb && banned[q] < v) ++q;
                if (q < nb && banned[q] == v) continue;
                for (int r = nb; r > q; --r) banned[r] = banned[r - 1];
                banned[q] = v;
                ++nb;
            }
            const int64_t M = p.id_max + 1 - nb;
            if (nneg <= kMaxFan) {
                nn = floyd_subset(M, nneg, P_NEG_EXACT, (uint32_t)u, 0, p.seed, p.step, sel);
                for (int q = 0; q < nn; ++q) {
                    int32_t id = sel[q];
                    for (int b = 0; b < nb; ++b)
                        if (banned[b] <= id) ++id;
                    neg_items[q] = id;
                }
            } else {
                // more negatives than the LDS list holds (a heavy user of a tiny graph): the same Floyd draws with the
                // sample's article bitmap (all zero at this point, M <= id_max + 1 bits) as the "seen" set, read back in
                // ascending order and mapped past the banned ids by a two-pointer walk
                uint32_t* bm = p.bm_art + (int64_t)s * p.WA;
                if (M <= nneg) {
                    nn = (int)M;
                    int b = 0;
                    for (int32_t r = 0; r < (int32_t)M; ++r) {
                        while (b < nb && banned[b] <= r + b) ++b;
                        neg_items[r] = r + b;
                    }
                } else {
                    for (int64_t j = M - nneg; j < M; ++j) {
                        const int32_t tdraw = (int32_t)rand_below((uint64_t)(j + 1), P_NEG_EXACT, (uint32_t)u, 0, (uint32_t)j, p.seed, p.step);
                        const bool seen = (bm[tdraw >> 5] >> (tdraw & 31)) & 1u;
                        const int32_t pick = seen ? (int32_t)j : tdraw;
                        bm[pick >> 5] |= 1u << (pick & 31);
                    }
                    nn = 0;
                    int b = 0;
                    const int words = (int)((M + 31) / 32);
                    for (int w = 0; w < words; ++w) {
                        uint32_t word = bm[w];
                        bm[w] = 0u;
                        while (word) {
                            const int32_t r = w * 32 + (__ffs(word) - 1);
                            word &= word - 1;
                            while (b < nb && banned[b] <= r + b) ++b;
                            neg_items[nn++] = r + b;
                        }
                    }
                }
            }
        }
        p.n_neg[s] = nn;
    }
    __syncthreads();   // thread 0 used sel[] in the exact negative path
    if (p.H >= 2 && deg > 0)   // block-uniform
        cut_articles_block(p, &u, 1, (uint32_t)u, 0, sel, cut_tmp, cut_pre, p.aq + (int64_t)s * p.n, p.aq_n + s);
}

// ---- phase 2 (per hop): mark the users of the queued articles -------------------------------------
// grid (n, B, kMarkSplit): a hub article's user list (10^5..10^6 entries) is spread over kMarkSplit workgroups
constexpr int kMarkSplit = 32;
__global__ __launch_bounds__(256) void smp_mark_users_kernel(Smp p) {
    const int s = blockIdx.y, j = blockIdx.x;
    if (j >= p.aq_n[s] || p.aq_L[s] > p.reject_min) return;  // long lists: drawn by rejection, nothing to mark
    const int32_t a = p.aq[(int64_t)s * p.n + j];
    uint32_t* bm = p.bm_users + (int64_t)s * p.WU;
    for (int32_t q = p.aptr[a] + blockIdx.z * blockDim.x + threadIdx.x; q < p.aptr[a + 1]; q += kMarkSplit * blockDim.x) {
        const int32_t v = p.aidx[q];
        atomicOr(bm + (v >> 5), 1u << (v & 31));
    }
}

// ---- phase 3 (per hop): distinct unexplored candidates -> uniform n-subset -> next queue -----------
__global__ __launch_bounds__(kSelThreads) void smp_select_users_kernel(Smp p, int hop) {
    __shared__ int32_t scan[kSelThreads + 1];
    __shared__ int32_t sel[kMaxFan];
    __shared__ int64_t pre[kMaxFan + 1];
    __shared__ int32_t cand_v[kSelThreads];
    const int s = blockIdx.x, tid = threadIdx.x;
    uint32_t* bm = p.bm_users + (int64_t)s * p.WU;
    const int32_t u = (int32_t)p.seeds[s];
    int32_t* uq = p.uq + (int64_t)s * p.H * p.n;
    int32_t* uq_n = p.uq_n + (int64_t)s * p.H;
    if (p.aq_L[s] > p.reject_min) {  // block-uniform
        // Rejection pick (oracle/sampler_ref.py:reject_pick_users): position of the concatenated user lists ->
        // user v with probability ~ m(v); accept with probability 1/m(v); skip explored / picked; draws are
        // evaluated 64 at a time by wavefront 0 and committed in counter order.
        __shared__ int32_t aq[kMaxFan];      // queue order: positions of the concatenated lists refer to it
        __shared__ int32_t aq_sorted[kMaxFan];  // ascending: multiplicity lookups by binary search
        __shared__ int32_t cand_ok[kSelThreads];
        __shared__ int32_t cand_b[kSelThreads], cand_e[kSelThreads];
        __shared__ int32_t expl[4 * kMaxFan];  // users explored so far (hops 0..hop), staged once
        __shared__ int sh_np, sh_t0, sh_nexpl;
        volatile int32_t* picked = sel;
        const int naq = p.aq_n[s];
        // set-up by the whole block (round 3; one thread used to walk the explored users, the queued articles' list lengths
        // — two dependent global loads each — and an insertion sort: about half of this kernel's 214 us at fan-out 64)
        for (int q = tid; q < naq; q += blockDim.x) {
            const int32_t a = p.aq[(int64_t)s * p.n + q];
            aq[q] = a;
            cand_e[q] = p.aptr[a + 1] - p.aptr[a];   // list length, staged for the prefix below (cand_e is free until the rounds)
        }
        {
            int ne = 0;   // every thread: the explored users' offsets per hop (a handful of hops)
            for (int h = 0; h <= hop; ++h) {
                const int cnt = uq_n[h];
                if (ne >= 0 && ne + cnt > 4 * kMaxFan) ne = -1;   // does not fit: read the lists from memory below
                if (ne >= 0) {
                    for (int q = tid; q < cnt; q += blockDim.x) expl[ne + q] = uq[h * p.n + q];
                    ne += cnt;
                }
            }
            if (tid == 0) sh_nexpl = ne;
        }
        __syncthreads();
        if (tid == 0) {
            int64_t run = 0;
            for (int q = 0; q < naq; ++q) { pre[q] = run; run += cand_e[q]; }
            pre[naq] = run;
            sh_np = 0;
            sh_t0 = 0;
        }
        for (int q = tid; q < naq; q += blockDim.x) {   // aq_sorted by rank (duplicates keep their queue order)
            const int32_t v = aq[q];
            int r = 0;
            for (int x = 0; x < naq; ++x) r += (aq[x] < v || (aq[x] == v && x < q)) ? 1 : 0;
            aq_sorted[r] = v;
        }
        __syncthreads();
        const int64_t L = pre[naq];
        const int n_expl = sh_nexpl;
        const uint32_t c3 = (P_USER_REJ & 0xFFu) | ((uint32_t)(p.step & 0xFFFFFFu) << 8);
        const uint32_t k0 = (uint32_t)p.seed, k1 = (uint32_t)((p.seed >> 32) ^ (p.step >> 24));
        // One round = R draws evaluated in parallel, committed in counter order.  Round 3: R follows the fan-out instead of
        // being the block size — a pick is accepted with probability (distinct users) / (list entries), typically well over
        // one half, so n = 64 picks need ~100 draws, and evaluating 1 024 per round meant every wavefront walked 64
        // candidates' article lists where 8 do (280 -> ~60 us per launch at the H&M shape).  The draws themselves and the
        // order they are committed in are unchanged: draw t is Philox(t), whatever the round it falls in.
        __shared__ int32_t cand_m[kSelThreads];
        int R = 2 * p.n;
        R = (R + MI_WAVE - 1) / MI_WAVE * MI_WAVE;
        R = max(2 * MI_WAVE, min(R, kSelThreads));
        const int lane = tid & (MI_WAVE - 1), wv = tid / MI_WAVE;
        while (true) {
            const int t0 = sh_t0;
            if (sh_np >= p.n || t0 >= kRejectCap) break;  // block-uniform (shared)
            MiPhilox w = mi_philox4x32((uint32_t)(t0 + tid), (uint32_t)hop, (uint32_t)u, c3, k0, k1);
            int32_t v = -1;
            if (tid < R) {
                const int64_t pos = (int64_t)((((uint64_t)w.c[0] << 32) | w.c[1]) % (uint64_t)L);
                int lo = 0, hi = naq;  // last q with pre[q] <= pos
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (pre[mid] <= pos) lo = mid; else hi = mid;
                }
                v = p.aidx[p.aptr[aq[lo]] + (pos - pre[lo])];
                cand_v[tid] = v;
                cand_b[tid] = p.uptr[v];
                cand_e[tid] = p.uptr[v + 1];
            }
            __syncthreads();
            // m(v) = number of queued articles (duplicates counted) that v bought: every wavefront walks the article lists of
            // its share of the round's candidates (interleaved: wavefront wv takes wv, wv + 16, ...) with all its lanes
            for (int ci = wv; ci < R; ci += kSelThreads / MI_WAVE) {
                const int32_t xb = cand_b[ci], xe = cand_e[ci];
                uint32_t cnt = 0;
                for (int32_t x = xb + lane; x < xe; x += MI_WAVE) {
                    const int32_t a = p.uidx[x];
                    int l2 = 0, h2 = naq;
                    while (l2 < h2) {
                        const int mid = (l2 + h2) >> 1;
                        if (aq_sorted[mid] < a) l2 = mid + 1; else h2 = mid;
                    }
                    while (l2 < naq && aq_sorted[l2] == a) { ++cnt; ++l2; }
                }
#pragma unroll
                for (int off = MI_WAVE / 2; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, MI_WAVE);
                if (lane == 0) cand_m[ci] = (int32_t)cnt;
            }
            __syncthreads();
            if (tid < R) {
                const uint32_t m = (uint32_t)cand_m[tid];
                bool ok = (m > 0) && (w.c[2] % m == 0);
                if (n_expl >= 0) {
                    for (int q = 0; ok && q < n_expl; ++q) ok &= (expl[q] != v);
                } else {
                    for (int h = 0; ok && h <= hop; ++h)
                        for (int q = 0; q < uq_n[h]; ++q) ok &= (uq[h * p.n + q] != v);
                }
                cand_ok[tid] = ok ? 1 : 0;
            }
            __syncthreads();
            if (tid < MI_WAVE) {
                int np = sh_np;
                for (int c = 0; c < R / MI_WAVE && np < p.n; ++c) {
                    const int32_t cv = cand_v[c * MI_WAVE + tid];
                    unsigned long long mask = __ballot(cand_ok[c * MI_WAVE + tid] != 0);
                    while (mask && np < p.n) {
                        const int l = __ffsll((long long)mask) - 1;
                        mask &= mask - 1;
                        const int32_t vv = __shfl(cv, l, MI_WAVE);
                        bool dup = false;
                        for (int q = tid; q < np; q += MI_WAVE) dup |= (picked[q] == vv);
                        if (__ballot(dup) == 0ull) {
                            if (tid == 0) picked[np] = vv;
                            ++np;
                        }
                    }
                }
                if (tid == 0) {
                    sh_np = np;
                    sh_t0 = t0 + R;
                }
            }
            __syncthreads();
        }
        {
            const int np = sh_np;
            block_rank_sort_distinct(sel, np, scan);   // ascending; the picks are distinct
            for (int q = tid; q < np; q += blockDim.x) uq[(hop + 1) * p.n + q] = sel[q];
            if (tid == 0) {
                uq_n[hop + 1] = np;
                p.aq_n[s] = 0;
                p.aq_L[s] = 0;
            }
            __syncthreads();   // uq's new row is read back by the cut below
            if (hop + 1 <= p.H - 2 && np > 0)   // block-uniform
                cut_articles_block(p, uq + (hop + 1) * p.n, np, (uint32_t)u, hop + 1, sel, scan, pre, p.aq + (int64_t)s * p.n, p.aq_n + s);
        }
        return;
    }
    // explored users are not candidates
    for (int h = 0; h <= hop; ++h)
        for (int q = tid; q < uq_n[h]; q += blockDim.x) {
            const int32_t v = uq[h * p.n + q];
            atomicAnd(bm + (v >> 5), ~(1u << (v & 31)));
        }
    __syncthreads();
    const int per = (p.WU + kSelThreads - 1) / kSelThreads;
    const int w0 = min(tid * per, p.WU), w1 = min(w0 + per, p.WU);
    int c = 0;
    for (int w = w0; w < w1; ++w) c += __popc(bm[w]);
    scan[tid + 1] = c;
    if (tid == 0) scan[0] = 0;
    __syncthreads();
    for (int off = 1; off < kSelThreads; off <<= 1) {  // inclusive scan of scan[1..]
        int v = (tid + 1 > off) ? scan[tid + 1 - off] : 0;
        __syncthreads();
        scan[tid + 1] += (tid + 1 > off) ? v : 0;
        __syncthreads();
    }
    const int nsel = floyd_subset_block(scan[kSelThreads], p.n, P_USER_CUT, (uint32_t)u, (uint32_t)hop, p.seed, p.step, sel, cand_v);
    for (int q = tid; q < nsel; q += blockDim.x) {  // rank -> id
        const int32_t r = sel[q];
        int lo = 0, hi = kSelThreads;  // last t with scan[t] <= r
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (scan[mid] <= r) lo = mid; else hi = mid;
        }
        int32_t rem = r - scan[lo];
        int32_t id = -1;
        for (int w = min(lo * per, p.WU); w < min(lo * per + per, p.WU); ++w) {
            uint32_t word = bm[w];
            const int pc = __popc(word);
            if (rem < pc) {
                for (int b = 0; b < rem; ++b) word &= word - 1;  // drop the `rem` lowest set bits
                id = w * 32 + (__ffs(word) - 1);
                break;
            }
            rem -= pc;
        }
        uq[(hop + 1) * p.n + q] = id;
    }
    __syncthreads();
    for (int w = w0; w < w1; ++w) bm[w] = 0u;  // ready for the next hop / batch
    if (tid == 0) {
        uq_n[hop + 1] = nsel;
        p.aq_n[s] = 0;
        p.aq_L[s] = 0;
    }
    __syncthreads();   // uq's new row (written above by the whole block) is read back by the cut
    if (hop + 1 <= p.H - 2 && nsel > 0)  // block-uniform: these users will be expanded too: cut their article lists
        cut_articles_block(p, uq + (hop + 1) * p.n, nsel, (uint32_t)u, hop + 1, sel, cand_v, pre, p.aq + (int64_t)s * p.n, p.aq_n + s);
}

// ---- phase 4: touched articles -> bitmap ---------------------------------------------------------
__global__ __launch_bounds__(256) void smp_mark_articles_kernel(Smp p) {
    const int s = blockIdx.y, j = blockIdx.x;
    uint32_t* bm = p.bm_art + (int64_t)s * p.WA;
    if (j == 0) {  // label items
        const int32_t* pi = p.pos_items + (int64_t)s * p.max_pos;
        const int32_t* ni = p.neg_items + (int64_t)s * p.max_neg;
        for (int q = threadIdx.x; q < p.n_pos[s]; q += blockDim.x) atomicOr(bm + (pi[q] >> 5), 1u << (pi[q] & 31));
        for (int q = threadIdx.x; q < p.n_neg[s]; q += blockDim.x) atomicOr(bm + (ni[q] >> 5), 1u << (ni[q] & 31));
        return;
    }
    const int e = j - 1, h = e / p.n, q = e % p.n;  // user slot (h, q); h = 0 is the seed
    if (h >= p.H || q >= p.uq_n[(int64_t)s * p.H + h]) return;
    const int32_t usr = p.uq[((int64_t)s * p.H + h) * p.n + q];
    for (int32_t x = p.uptr[usr] + threadIdx.x; x < p.uptr[usr + 1]; x += blockDim.x) {
        const int32_t a = p.uidx[x];
        atomicOr(bm + (a >> 5), 1u << (a & 31));
    }
}

// ---- phase 5: per-sample counts, article rank prefix, user ranks, edge starts ----------------------
__global__ __launch_bounds__(kSelThreads) void smp_count_kernel(Smp p) {
    __shared__ int32_t scan[kSelThreads + 1];
    const int s = blockIdx.x, tid = threadIdx.x;
    const uint32_t* bm = p.bm_art + (int64_t)s * p.WA;
    int32_t* pre = p.pre_a + (int64_t)s * p.WA;
    const int per = (p.WA + kSelThreads - 1) / kSelThreads;
    const int w0 = min(tid * per, p.WA), w1 = min(w0 + per, p.WA);
    int c = 0;
    for (int w = w0; w < w1; ++w) c += __popc(bm[w]);
    scan[tid + 1] = c;
    if (tid == 0) scan[0] = 0;
    __syncthreads();
    for (int off = 1; off < kSelThreads; off <<= 1) {
        int v = (tid + 1 > off) ? scan[tid + 1 - off] : 0;
        __syncthreads();
        scan[tid + 1] += (tid + 1 > off) ? v : 0;
        __syncthreads();
    }
    int run = scan[tid];
    for (int w = w0; w < w1; ++w) {
        pre[w] = run;
        run += __popc(bm[w]);
    }
    const int32_t* uq = p.uq + (int64_t)s * p.H * p.n;
    const int32_t* uq_n = p.uq_n + (int64_t)s * p.H;
    int32_t* ul = p.uq_local + (int64_t)s * p.H * p.n;
    // local id of a user = number of smaller user ids in the sample (all entries are distinct)
    for (int e = tid; e < p.H * p.n; e += blockDim.x) {
        const int h = e / p.n, q = e % p.n;
        if (q >= uq_n[h]) continue;
        const int32_t v = uq[e];
        int r = 0, before = 0;
        for (int h2 = 0; h2 < p.H; ++h2)
            for (int q2 = 0; q2 < uq_n[h2]; ++q2) {
                const int32_t w = uq[h2 * p.n + q2];
                r += (w < v);
                before += (w < v) ? p.uptr[w + 1] - p.uptr[w] : 0;
            }
        ul[e] = r;
        p.uq_rstart[(int64_t)s * p.H * p.n + e] = before;
    }
    if (tid == 0) {
        int32_t* es = p.uq_estart + (int64_t)s * p.H * p.n;
        int nu = 0, ne = 0;
        for (int h = 0; h < p.H; ++h)
            for (int q = 0; q < uq_n[h]; ++q) {
                const int32_t v = uq[h * p.n + q];
                es[h * p.n + q] = ne;
                ne += p.uptr[v + 1] - p.uptr[v];
                ++nu;
            }
        int32_t* cnt = p.cnt + (int64_t)s * 4;
        cnt[0] = nu;
        cnt[1] = scan[kSelThreads];
        cnt[2] = ne;
        cnt[3] = p.n_pos[s] + p.n_neg[s];
    }
}

__global__ void smp_offsets_kernel(Smp p) {  // B is small: one thread per counter
    const int c = threadIdx.x;
    if (c >= 4) return;
    int32_t run = 0;
    for (int s = 0; s < p.B; ++s) {
        p.off[c * (p.B + 1) + s] = run;
        run += p.cnt[(int64_t)s * 4 + c];
    }
    p.off[c * (p.B + 1) + p.B] = run;
    p.off[4 * (p.B + 1) + c] = run;
}

struct SmpOut {
    int64_t* user_ids; int64_t* article_ids;
    int64_t* edge_index; int64_t n_edges;        // [2, n_edges]
    int64_t* label_index; int64_t n_labels;      // [2, n_labels]
    int64_t* labels;                             // [n_labels]
    int64_t* user_ptr; int64_t* article_ptr;     // [B+1]
    int32_t three_rows;                          // edge_index / label_index are [3, n]: row 2 repeats row 0 (rows 1..2 = the reversed relation)
};

__device__ __forceinline__ int32_t article_rank(const Smp& p, int s, int32_t a) {
    const uint32_t word = p.bm_art[(int64_t)s * p.WA + (a >> 5)];
    return p.pre_a[(int64_t)s * p.WA + (a >> 5)] + __popc(word & ((1u << (a & 31)) - 1u));
}

// ---- phase 6: emit ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void smp_emit_edges_kernel(Smp p, SmpOut o) {
    const int s = blockIdx.y, j = blockIdx.x;
    const int32_t off_u = p.off[0 * (p.B + 1) + s], off_a = p.off[1 * (p.B + 1) + s];
    const int32_t off_e = p.off[2 * (p.B + 1) + s], off_l = p.off[3 * (p.B + 1) + s];
    if (j == 0) {  // label edges: positives in draw order, then negatives
        const int32_t seed_local = p.uq_local[(int64_t)s * p.H * p.n];
        const int np_ = p.n_pos[s], nn = p.n_neg[s];
        for (int q = threadIdx.x; q < np_ + nn; q += blockDim.x) {
            const int32_t a = q < np_ ? p.pos_items[(int64_t)s * p.max_pos + q] : p.neg_items[(int64_t)s * p.max_neg + q - np_];
            o.label_index[off_l + q] = off_u + seed_local;
            o.label_index[o.n_labels + off_l + q] = off_a + article_rank(p, s, a);
            if (o.three_rows) o.label_index[2 * o.n_labels + off_l + q] = off_u + seed_local;
            o.labels[off_l + q] = q < np_ ? 1 : 0;
        }
        if (threadIdx.x == 0) {
            o.user_ptr[s] = off_u;
            o.article_ptr[s] = off_a;
            if (s == p.B - 1) {
                o.user_ptr[p.B] = p.off[0 * (p.B + 1) + p.B];
                o.article_ptr[p.B] = p.off[1 * (p.B + 1) + p.B];
            }
        }
        return;
    }
    const int e = j - 1, h = e / p.n, q = e % p.n;
    if (h >= p.H || q >= p.uq_n[(int64_t)s * p.H + h]) return;
    const int64_t slot = ((int64_t)s * p.H + h) * p.n + q;
    const int32_t usr = p.uq[slot], ul = p.uq_local[slot], es = p.uq_estart[slot];
    if (threadIdx.x == 0) o.user_ids[off_u + ul] = usr;
    const int32_t beg = p.uptr[usr], deg = p.uptr[usr + 1] - beg;
    for (int x = threadIdx.x; x < deg; x += blockDim.x) {
        o.edge_index[off_e + es + x] = off_u + ul;
        o.edge_index[o.n_edges + off_e + es + x] = off_a + article_rank(p, s, p.uidx[beg + x]);
        if (o.three_rows) o.edge_index[2 * o.n_edges + off_e + es + x] = off_u + ul;
    }
}

__global__ __launch_bounds__(256) void smp_emit_articles_kernel(Smp p, SmpOut o) {
    const int s = blockIdx.y;
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= p.WA) return;
    uint32_t word = p.bm_art[(int64_t)s * p.WA + w];
    int32_t r = p.off[1 * (p.B + 1) + s] + p.pre_a[(int64_t)s * p.WA + w];
    while (word) {
        const int b = __ffs(word) - 1;
        o.article_ids[r++] = (int64_t)w * 32 + b;
        word &= word - 1;
    }
}

// ---- phase 6b: the batch's message-passing edges as two sorted CSRs ---------------------------------
// The encoder reads every relation as a CSR sorted by (row, column) in both directions (model/layers.py
// BipartiteGraph).  The samples of a batch are disjoint graphs with their nodes numbered in id order, so both CSRs
// can be written straight from the walk's tables: customer rows are the explored users in rank order with their
// article lists relabelled and sorted; article rows are the transpose, at most one sample's users long.
struct SmpCsr {
    int32_t* u_rowptr; int32_t* u_col;   // customers x articles
    int32_t* a_rowptr; int32_t* a_col;   // articles x customers
    int32_t* a_cur;                      // [total articles] fill cursors
    int32_t n_users, n_articles, n_edges;
};
constexpr int kCsrSortCap = 8192;  // keys one workgroup sorts in LDS; longer lists take the quadratic path
constexpr int kCsrRowRegs = 8;     // article rows up to 64 * kCsrRowRegs customers (guarded on the host)

__global__ __launch_bounds__(256) void smp_csr_users_kernel(Smp p, SmpCsr o) {
    __shared__ int32_t key[kCsrSortCap];
    const int s = blockIdx.y, e = blockIdx.x, h = e / p.n, q = e % p.n;
    const int tid = threadIdx.x;
    const int32_t off_u = p.off[0 * (p.B + 1) + s], off_a = p.off[1 * (p.B + 1) + s], off_e = p.off[2 * (p.B + 1) + s];
    if (s == p.B - 1 && e == 0 && tid == 0) o.u_rowptr[o.n_users] = o.n_edges;
    if (h >= p.H || q >= p.uq_n[(int64_t)s * p.H + h]) return;
    const int64_t slot = ((int64_t)s * p.H + h) * p.n + q;
    const int32_t usr = p.uq[slot], ul = p.uq_local[slot];
    const int32_t base = off_e + p.uq_rstart[slot];
    if (tid == 0) o.u_rowptr[off_u + ul] = base;
    const int32_t beg = p.uptr[usr], deg = p.uptr[usr + 1] - beg;
    if (deg <= 0) return;
    if (deg > kCsrSortCap) {  // rank by counting over the list itself: correct for any length, slow, rare
        for (int x = tid; x < deg; x += blockDim.x) {
            const int32_t r = article_rank(p, s, p.uidx[beg + x]);
            int pos = 0;
            for (int y = 0; y < deg; ++y) {
                const int32_t r2 = article_rank(p, s, p.uidx[beg + y]);
                pos += (r2 < r) || (r2 == r && y < x);
            }
            o.u_col[base + pos] = off_a + r;
            atomicAdd(o.a_rowptr + off_a + r + 1, 1);
        }
        return;
    }
    for (int x = tid; x < deg; x += blockDim.x) key[x] = article_rank(p, s, p.uidx[beg + x]);
    __syncthreads();
    if (deg <= 256) {  // one key per thread: position = keys below it (ties by list position)
        if (tid < deg) {
            const int32_t r = key[tid];
            int pos = 0;
            for (int y = 0; y < deg; ++y) pos += (key[y] < r) || (key[y] == r && y < tid);
            o.u_col[base + pos] = off_a + r;
            atomicAdd(o.a_rowptr + off_a + r + 1, 1);
        }
        return;
    }
    int n2 = 512;
    while (n2 < deg) n2 <<= 1;
    for (int x = deg + tid; x < n2; x += blockDim.x) key[x] = INT32_MAX;
    __syncthreads();
    for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int x = tid; x < n2; x += blockDim.x) {
                const int y = x ^ j;
                if (y > x) {
                    const int32_t a = key[x], b = key[y];
                    const bool up = (x & k) == 0;
                    if ((a > b) == up) { key[x] = b; key[y] = a; }
                }
            }
            __syncthreads();
        }
    for (int x = tid; x < deg; x += blockDim.x) {
        o.u_col[base + x] = off_a + key[x];
        atomicAdd(o.a_rowptr + off_a + key[x] + 1, 1);
    }
}

// per sample: counts (left in a_rowptr[off_a + 1 + i] by the kernel above) -> row ends, row starts -> cursors
__global__ __launch_bounds__(kSelThreads) void smp_csr_scan_kernel(Smp p, SmpCsr o) {
    __shared__ int32_t scan[kSelThreads + 1];
    const int s = blockIdx.x, tid = threadIdx.x;
    const int32_t off_a = p.off[1 * (p.B + 1) + s], off_e = p.off[2 * (p.B + 1) + s];
    const int na = p.cnt[(int64_t)s * 4 + 1];
    const int per = (na + kSelThreads - 1) / kSelThreads;
    const int i0 = min(tid * per, na), i1 = min(i0 + per, na);
    int32_t* cntp = o.a_rowptr + off_a + 1;
    __shared__ int longest;
    if (tid == 0) longest = 0;
    __syncthreads();
    int c = 0, mx = 0;
    for (int i = i0; i < i1; ++i) {
        c += cntp[i];
        mx = max(mx, cntp[i]);
    }
    if (mx > MI_WAVE * kCsrRowRegs) atomicMax(&longest, mx);
    scan[tid + 1] = c;
    if (tid == 0) scan[0] = 0;
    __syncthreads();
    if (tid == 0) p.csr_long[s] = longest > MI_WAVE * kCsrRowRegs;  // (a customer may hold an article many times: rows are not bounded by the sample's users)
    for (int off = 1; off < kSelThreads; off <<= 1) {
        int v = (tid + 1 > off) ? scan[tid + 1 - off] : 0;
        __syncthreads();
        scan[tid + 1] += (tid + 1 > off) ? v : 0;
        __syncthreads();
    }
    int run = off_e + scan[tid];
    for (int i = i0; i < i1; ++i) {
        o.a_cur[off_a + i] = run;
        run += cntp[i];
        cntp[i] = run;
    }
}

__global__ __launch_bounds__(256) void smp_csr_fill_kernel(Smp p, SmpCsr o) {
    const int s = blockIdx.y, e = blockIdx.x, h = e / p.n, q = e % p.n;
    if (h >= p.H || q >= p.uq_n[(int64_t)s * p.H + h]) return;
    const int32_t off_u = p.off[0 * (p.B + 1) + s], off_a = p.off[1 * (p.B + 1) + s];
    const int64_t slot = ((int64_t)s * p.H + h) * p.n + q;
    const int32_t usr = p.uq[slot], ul = p.uq_local[slot];
    for (int32_t x = p.uptr[usr] + threadIdx.x; x < p.uptr[usr + 1]; x += blockDim.x) {
        const int32_t pos = atomicAdd(o.a_cur + off_a + article_rank(p, s, p.uidx[x]), 1);
        o.a_col[pos] = off_u + ul;
    }
}

// one wave per article row: the cursors filled it in arrival order; put it in customer order.  Every lane ranks
// its own entries against the whole row (reads), then writes them — the wave runs in lockstep, so all reads of
// the row have returned before the first store is issued.
__global__ __launch_bounds__(256) void smp_csr_sort_rows_kernel(SmpCsr o) {
    const int row = blockIdx.x * (blockDim.x / MI_WAVE) + threadIdx.x / MI_WAVE, lane = threadIdx.x % MI_WAVE;
    if (row >= o.n_articles) return;
    const int32_t b = o.a_rowptr[row], len = o.a_rowptr[row + 1] - b;
    if (len <= 1 || len > MI_WAVE * kCsrRowRegs) return;  // longer rows: their whole sample is refilled in order (smp_csr_refill_kernel)
    int32_t* r = o.a_col + b;
    int32_t mine[kCsrRowRegs], pos[kCsrRowRegs];
#pragma unroll
    for (int k = 0; k < kCsrRowRegs; ++k) {
        const int i = lane + MI_WAVE * k;
        mine[k] = i < len ? r[i] : 0;
        pos[k] = 0;
    }
    for (int y = 0; y < len; ++y) {
        const int32_t v = r[y];
#pragma unroll
        for (int k = 0; k < kCsrRowRegs; ++k) {
            const int i = lane + MI_WAVE * k;
            pos[k] += (v < mine[k]) || (v == mine[k] && y < i);
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < kCsrRowRegs; ++k)
        if (lane + MI_WAVE * k < len) r[pos[k]] = mine[k];
}


// Samples with an article row longer than the row sort handles (a customer holding one article hundreds of times): the
// sample's article rows are filled again, this time customer by customer in rank order with a barrier in between, so
// every row comes out sorted without a sort.  Slow (one barrier per customer) and rare; all other samples return at once.
__global__ __launch_bounds__(kSelThreads) void smp_csr_refill_kernel(Smp p, SmpCsr o) {
    __shared__ int32_t by_rank[MI_WAVE * kCsrRowRegs];
    const int s = blockIdx.x, tid = threadIdx.x;
    if (!p.csr_long[s]) return;
    const int32_t off_u = p.off[0 * (p.B + 1) + s], off_a = p.off[1 * (p.B + 1) + s];
    const int na = p.cnt[(int64_t)s * 4 + 1], nu = p.cnt[(int64_t)s * 4 + 0];
    for (int i = tid; i < na; i += blockDim.x) o.a_cur[off_a + i] = o.a_rowptr[off_a + i];  // row starts
    const int32_t* uq_n = p.uq_n + (int64_t)s * p.H;
    for (int e = tid; e < p.H * p.n; e += blockDim.x) {
        const int h = e / p.n, q = e % p.n;
        if (q < uq_n[h]) by_rank[p.uq_local[(int64_t)s * p.H * p.n + e]] = e;
    }
    __syncthreads();
    for (int r = 0; r < nu; ++r) {
        const int32_t usr = p.uq[(int64_t)s * p.H * p.n + by_rank[r]];
        for (int32_t x = p.uptr[usr] + tid; x < p.uptr[usr + 1]; x += blockDim.x) {
            const int32_t pos = atomicAdd(o.a_cur + off_a + article_rank(p, s, p.uidx[x]), 1);
            o.a_col[pos] = off_u + r;
        }
        __syncthreads();  // the next customer's entries go behind this one's (the atomics have returned)
    }
}

size_t smp_scratch_layout(const Smp& p, Smp* out, char* base) {
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char* r = base ? base + off : nullptr;
        off += mi_align_up(bytes, 256);
        return r;
    };
    const size_t B = p.B, H = p.H, n = p.n;
    char* a0 = take(B * p.WU * 4);
    char* a1 = take(B * p.WA * 4);
    char* a2 = take(B * p.WA * 4);
    char* a3 = take(B * H * n * 4);
    char* a4 = take(B * H * 4);
    char* a5 = take(B * H * n * 4);
    char* a6 = take(B * H * n * 4);
    char* a6b = take(B * H * n * 4);
    char* a7 = take(B * n * 4);
    char* a8 = take(B * 4);
    char* a8b = take(B * 8);
    char* a9 = take(B * (size_t)p.max_pos * 4);
    char* a10 = take(B * (size_t)p.max_neg * 4);
    char* a11 = take(B * 4);
    char* a12 = take(B * 4);
    char* a13 = take(B * 4 * 4);
    char* a14 = take((4 * (B + 1) + 4) * 4);
    char* a15 = take(B * 4);
    char* a16 = take(B * (size_t)p.max_pos * 4);
    if (out) {
        out->bm_users = (uint32_t*)a0; out->bm_art = (uint32_t*)a1; out->pre_a = (int32_t*)a2;
        out->uq = (int32_t*)a3; out->uq_n = (int32_t*)a4; out->uq_local = (int32_t*)a5; out->uq_estart = (int32_t*)a6; out->uq_rstart = (int32_t*)a6b;
        out->aq = (int32_t*)a7; out->aq_n = (int32_t*)a8; out->aq_L = (int64_t*)a8b; out->pos_items = (int32_t*)a9; out->neg_items = (int32_t*)a10;
        out->n_pos = (int32_t*)a11; out->n_neg = (int32_t*)a12; out->cnt = (int32_t*)a13; out->off = (int32_t*)a14; out->csr_long = (int32_t*)a15; out->banned = (int32_t*)a16;
    }
    return off;
}

int fill_params(Smp& p, const mi_sampler_desc* d) {
    MI_CHECK_ARG(d && d->batch > 0 && d->n_hops >= 1 && d->num_neighbors >= 1 && d->num_neighbors <= kMaxFan);
    MI_CHECK_ARG(d->num_users > 0 && d->num_articles > 0 && d->users_ptr && d->users_idx && d->articles_ptr && d->articles_idx);
    MI_CHECK_ARG(d->max_pos >= 2 && d->max_neg >= 1);
    if (d->num_users >= INT32_MAX || d->num_articles >= INT32_MAX || d->num_edges >= INT32_MAX) return MI_ERR_TOO_LARGE;
    p.B = d->batch; p.H = d->n_hops; p.n = d->num_neighbors;
    p.U = d->num_users; p.A = d->num_articles; p.E = d->num_edges; p.id_max = d->id_max;
    p.uptr = d->users_ptr; p.uidx = d->users_idx; p.aptr = d->articles_ptr; p.aidx = d->articles_idx;
    p.cptr = d->cand_ptr; p.cidx = d->cand_idx;
    MI_CHECK_ARG(!p.cptr || p.cidx);
    p.pos_ratio = d->positive_edges_ratio; p.neg_ratio = d->negative_edges_ratio; p.k = d->k;
    p.randomization = d->randomization;
    p.max_pos = d->max_pos; p.max_neg = d->max_neg;
    // termination of the rejection pick needs >= n distinct unexplored users among the lists (header)
    const int64_t rmin_floor = (int64_t)d->num_neighbors * ((int64_t)d->num_neighbors * (d->n_hops + 1) + 1);
    p.reject_min = d->reject_min_entries > 0 ? d->reject_min_entries : 4 * rmin_floor;
    MI_CHECK_ARG(p.reject_min >= rmin_floor);
    p.WU = (int32_t)((d->num_users + 31) / 32);
    p.WA = (int32_t)((d->num_articles + 31) / 32);
    return 0;
}

}  // namespace

extern "C" {

size_t mi_sampler_workspace_bytes(const mi_sampler_desc* d) {
    Smp p;
    if (fill_params(p, d)) return 0;
    return smp_scratch_layout(p, nullptr, nullptr);
}

// Phase A: everything up to the per-sample counts; writes totals[4] (users, articles, edges, labels)
// to the HOST array after synchronising the stream.
int mi_sampler_count_async(const mi_sampler_desc* d, const int64_t* seed_users, uint64_t seed, uint64_t step, void* ws,
                           size_t ws_bytes, int32_t* totals_pinned, mi_stream_t stream) {
    Smp p;
    int rc = fill_params(p, d);
    if (rc) return rc;
    MI_CHECK_ARG(seed_users && ws && totals_pinned);
    if (ws_bytes < smp_scratch_layout(p, nullptr, nullptr)) return MI_ERR_WORKSPACE;
    smp_scratch_layout(p, &p, static_cast<char*>(ws));
    p.seeds = seed_users; p.seed = seed; p.step = step;
    hipStream_t s = (hipStream_t)stream;
    MI_HIP(hipMemsetAsync(p.bm_users, 0, (size_t)p.B * p.WU * 4, s));
    MI_HIP(hipMemsetAsync(p.bm_art, 0, (size_t)p.B * p.WA * 4, s));
    hipLaunchKernelGGL(smp_seed_kernel, dim3(p.B), dim3(256), 0, s, p);
    for (int hop = 0; hop <= p.H - 2; ++hop) {
        hipLaunchKernelGGL(smp_mark_users_kernel, dim3(p.n, p.B, kMarkSplit), dim3(256), 0, s, p);
        hipLaunchKernelGGL(smp_select_users_kernel, dim3(p.B), dim3(kSelThreads), 0, s, p, hop);
    }
    hipLaunchKernelGGL(smp_mark_articles_kernel, dim3(1 + p.H * p.n, p.B), dim3(256), 0, s, p);
    hipLaunchKernelGGL(smp_count_kernel, dim3(p.B), dim3(kSelThreads), 0, s, p);
    hipLaunchKernelGGL(smp_offsets_kernel, dim3(1), dim3(64), 0, s, p);
    MI_HIP(hipMemcpyAsync(totals_pinned, p.off + 4 * (p.B + 1), 4 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    return mi_launch_status();
}

int mi_sampler_count(const mi_sampler_desc* d, const int64_t* seed_users, uint64_t seed, uint64_t step, void* ws,
                     size_t ws_bytes, int64_t* totals_host, mi_stream_t stream) {
    MI_CHECK_ARG(totals_host);
    int32_t tot[4] = {0, 0, 0, 0};
    int rc = mi_sampler_count_async(d, seed_users, seed, step, ws, ws_bytes, tot, stream);
    if (rc) return rc;
    MI_HIP(hipStreamSynchronize((hipStream_t)stream));
    for (int c = 0; c < 4; ++c) totals_host[c] = tot[c];
    return mi_launch_status();
}

// Phase B: emit into caller-allocated outputs sized from the totals of phase A (same ws, untouched between).
int mi_sampler_emit(const mi_sampler_desc* d, const int64_t* seed_users, void* ws, size_t ws_bytes,
                    const int64_t* totals_host, int64_t* user_ids, int64_t* article_ids, int64_t* edge_index,
                    int64_t* edge_label_index, int64_t* edge_label, int64_t* user_ptr, int64_t* article_ptr,
                    mi_stream_t stream) {
    return mi_sampler_emit3(d, seed_users, ws, ws_bytes, totals_host, user_ids, article_ids, edge_index, edge_label_index, edge_label,
                            user_ptr, article_ptr, 0, stream);
}

int mi_sampler_emit3(const mi_sampler_desc* d, const int64_t* seed_users, void* ws, size_t ws_bytes,
                     const int64_t* totals_host, int64_t* user_ids, int64_t* article_ids, int64_t* edge_index,
                     int64_t* edge_label_index, int64_t* edge_label, int64_t* user_ptr, int64_t* article_ptr,
                     int32_t three_rows, mi_stream_t stream) {
    Smp p;
    int rc = fill_params(p, d);
    if (rc) return rc;
    MI_CHECK_ARG(ws && totals_host && user_ids && article_ids && edge_label_index && edge_label && user_ptr && article_ptr);
    MI_CHECK_ARG(totals_host[2] == 0 || edge_index);
    if (ws_bytes < smp_scratch_layout(p, nullptr, nullptr)) return MI_ERR_WORKSPACE;
    smp_scratch_layout(p, &p, static_cast<char*>(ws));
    p.seeds = seed_users;
    SmpOut o;
    o.user_ids = user_ids; o.article_ids = article_ids;
    o.edge_index = edge_index; o.n_edges = totals_host[2];
    o.label_index = edge_label_index; o.n_labels = totals_host[3];
    o.labels = edge_label; o.user_ptr = user_ptr; o.article_ptr = article_ptr;
    o.three_rows = three_rows ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(smp_emit_edges_kernel, dim3(1 + p.H * p.n, p.B), dim3(256), 0, s, p, o);
    hipLaunchKernelGGL(smp_emit_articles_kernel, dim3((p.WA + 255) / 256, p.B), dim3(256), 0, s, p, o);
    return mi_launch_status();
}

// Phase B': the same edges as the two sorted CSRs the encoder consumes (after phase A, before or after mi_sampler_emit).
int mi_sampler_emit_csr(const mi_sampler_desc* d, void* ws, size_t ws_bytes, const int64_t* totals_host,
                        int32_t* customer_rowptr, int32_t* customer_col, int32_t* article_rowptr, int32_t* article_col,
                        int32_t* article_cursor, mi_stream_t stream) {
    Smp p;
    int rc = fill_params(p, d);
    if (rc) return rc;
    MI_CHECK_ARG(ws && totals_host && customer_rowptr && article_rowptr);
    MI_CHECK_ARG(totals_host[2] == 0 || (customer_col && article_col));
    MI_CHECK_ARG(totals_host[1] == 0 || article_cursor);
    if ((int64_t)p.H * p.n > (int64_t)MI_WAVE * kCsrRowRegs) return MI_ERR_UNSUPPORTED;  // bound on an article row
    if (ws_bytes < smp_scratch_layout(p, nullptr, nullptr)) return MI_ERR_WORKSPACE;
    smp_scratch_layout(p, &p, static_cast<char*>(ws));
    SmpCsr o;
    o.u_rowptr = customer_rowptr; o.u_col = customer_col; o.a_rowptr = article_rowptr; o.a_col = article_col;
    o.a_cur = article_cursor;
    o.n_users = (int32_t)totals_host[0]; o.n_articles = (int32_t)totals_host[1]; o.n_edges = (int32_t)totals_host[2];
    hipStream_t s = (hipStream_t)stream;
    MI_HIP(hipMemsetAsync(article_rowptr, 0, ((size_t)o.n_articles + 1) * 4, s));
    hipLaunchKernelGGL(smp_csr_users_kernel, dim3(p.H * p.n, p.B), dim3(256), 0, s, p, o);
    hipLaunchKernelGGL(smp_csr_scan_kernel, dim3(p.B), dim3(kSelThreads), 0, s, p, o);
    if (o.n_edges > 0) {
        hipLaunchKernelGGL(smp_csr_fill_kernel, dim3(p.H * p.n, p.B), dim3(256), 0, s, p, o);
        hipLaunchKernelGGL(smp_csr_sort_rows_kernel, dim3((o.n_articles + 3) / 4), dim3(256), 0, s, o);
        hipLaunchKernelGGL(smp_csr_refill_kernel, dim3(p.B), dim3(kSelThreads), 0, s, p, o);
    }
    return mi_launch_status();
}

}  // extern "C"

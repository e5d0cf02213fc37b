// K1 / K2: CSR SpMM with fused epilogue — the LightGCN propagate (model/lightgcn.py:63-68,87).
//
//   acc[r,:] = sum_p val[p] * X[col[p],:]
//   Y[r,:]   = acc                               (optional)
//   S[r,:]   = scale * (addend[r,:] + acc)        (optional; the running layer sum / mean)
//
// HBM-bound gather (0.49 flop/byte at D=128): the design is about bytes — keeping many whole-row
// 16-byte-per-lane loads in flight, and making the gathers of the hub rows hit L2.
//
// Mapping (wave64): a row of D floats is covered by a SUB-GROUP of LPR = D/4 lanes of float4, so a
// wavefront holds NB = 64/LPR sub-groups and one load instruction fetches NB neighbour rows (D=128:
// 2 rows, 1 KiB).  Each sub-group owns one destination row (or one work item of a split row) and
// accumulates its entries in list order: (col,val) pairs are read LPR at a time, coalesced, one per
// lane, and handed round with ds_bpermute (__shfl); UNROLL independent row loads per lane are issued
// before the first fma.  No cross-lane reduce is needed.  (Round-1 history: one wavefront per row
// with the sub-groups splitting the row's entries was bound by the chain rowptr -> col -> gather ->
// store over 10^6 ten-entry user rows; tools/ab_spmm.py.)
//
// Rows with more than plan->chunk entries are SPLIT into work items whose partial sums are reduced in
// slot order by a fix-up kernel, so there are no float atomics and results are bitwise reproducible.
// When the plan is banded, a split row is cut at multiples of `band` columns as well, and the work
// items are launched band by band with workgroups of one band pinned to one XCD (the dispatcher deals
// workgroups round-robin over the 8 XCDs, so launch blocks are interleaved 8 ways): the ~7 gathers per
// user row that the hub items of a recommendation graph make then meet in that XCD's 4 MB L2 instead
// of each going to HBM.  Measured on C2 (tools/ab_band.py, round 1): 1.20 -> 1.04 ms per launch with
// one wavefront per item; see DESIGN.md for the final figures.
#include "common.hpp"
#include "pairs.hpp"
#include <algorithm>
#include <rocprim/rocprim.hpp>

namespace {

#ifndef MI_SPMM_UNROLL
#define MI_SPMM_UNROLL 4   // row loads in flight per lane (VPL = 1).  Round 1 (row per wavefront half, C2): 4 -> 8 gained 2 %.  Round 4, on
                           // today's kernels (tools/exp_c4.py, bench.py --config c2 / c4, same loss to the last digit): 8 -> 4 takes the C2
                           // step 5.209 -> 5.005 ms (dense launch 0.893 -> 0.851, sparse 0.515 -> 0.488) and C4's dense launch 9.15-9.22 ->
                           // 9.06-9.12 ms; 2 loses (C4 54.1 ms / step, C2 5.30).  The short-row kernel WITH the Adam epilogue keeps 8
                           // (C4's Adam launch 12.0-12.15 vs 12.3 ms at 4; same-box triples of the C4 step: mixed 50.84 / all-8 51.21 /
                           // all-4 51.12 and 51.79 / 51.84 / 52.05 ms)
#endif
#ifndef MI_SPMM_UNROLL_ADAM
#define MI_SPMM_UNROLL_ADAM 8
#endif
#ifndef MI_SPMM_UNROLL_PAIR
#define MI_SPMM_UNROLL_PAIR 8   // the ranker's plan-less twin launches (spmm_rows_pair_kernel): measured at 8 (profiles/r04_ranker_native_timeline.txt)
#endif
#ifndef MI_SPMM_ROWS_RPS
#define MI_SPMM_ROWS_RPS 1  // rows a sub-group handles in sequence; A/B on C2: 1: 1.373 ms, 2: 1.408, 4: 1.397.  Two rows walked
                            // as ONE list per sub-group (shared (col,val) loads and gather batches): rows kernel 851 -> 916 us.
                            // The same pairing for the sparse-operand launches only (x_map / row_list): 0.61 -> 0.73 ms per launch; packing the
                            // entries that survive x_map to the front of the sub-group (ds_permute): 0.613 -> 0.597 ms, not kept.
                            // More rows per wavefront do not help: the kernel is bound by its L2-miss bytes
#endif
#ifndef MI_SPMM_NT
#define MI_SPMM_NT 0   // bit 0: non-temporal (col,val) loads; bit 1: partial-sum stores; bit 2: Y / S stores; bit 3: addend
                       // loads.  A/B on C2, round 1 (tools/prof_spmm.py under rocprofv3): every combination within +-3 %
                       // of off (nt keeps the line in L2 on gfx950), write-through sc1 partial stores 0.3 %.  Kept off.
#endif
typedef float mi_f4v __attribute__((ext_vector_type(4)));
#ifndef MI_SPMM_SC1
#define MI_SPMM_SC1 6  // bit 1: partial-sum stores, bit 2: Y / S stores leave with sc1: written-once streams do not stay in the XCD's
                       // L2 (MI355X_MICROARCH.md, store flavours), which the gathers need.  A/B on C2, round 2 (tools/exp_locality.py
                       // under rocprofv3): Y/S only: rows kernel 839 -> 827 us, 4.80 -> 4.65 GB of L2-miss traffic; + partials: items
                       // kernel 367 -> 360 us, 2.28 -> 2.20 GB
#endif
template <int BIT>
__device__ __forceinline__ void mi_store4(float4* p, const float4& v, bool streaming) {
    if (((MI_SPMM_SC1 >> BIT) & 1) && streaming) {
        mi_f4v x = {v.x, v.y, v.z, v.w};
        // s_nop: a store of more than 64 bits reads its data registers over several cycles and the hazard recogniser does not
        // look into inline asm — without the wait states the next instruction may overwrite them (seen as run-to-run different
        // partial rows once a kernel reused the accumulator registers right behind the store, round 4)
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(x) : "memory");
    } else if ((MI_SPMM_NT >> BIT) & 1) {
        mi_f4v x = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(x, reinterpret_cast<mi_f4v*>(p));
    } else {
        *p = v;
    }
}
template <int BIT>
__device__ __forceinline__ float4 mi_load4(const float4* p) {
    if ((MI_SPMM_NT >> BIT) & 1) {
        mi_f4v x = __builtin_nontemporal_load(reinterpret_cast<const mi_f4v*>(p));
        return make_float4(x.x, x.y, x.z, x.w);
    }
    return *p;
}
#ifndef MI_SPMM_FIXUP_UNROLL
#define MI_SPMM_FIXUP_UNROLL 32  // the most split row (C2: 2 100 partial rows) is one block's serial chain: 4: 81 us, 8: 65, 16: 58, 32: 51
#endif
constexpr int kWavesPerBlock = 4;
constexpr int kBlock = kWavesPerBlock * MI_WAVE;
constexpr int kPlanGroup = MI_SPMM_GROUP;  // launch slots per XCD-interleave block of a banded plan

struct Epilogue {
    float4* Y;            int64_t ldy4;
    const float4* addend; int64_t lda4;
    float4* S;            int64_t lds4;
    float scale;
    bool streaming;       // outputs far larger than the L2s: stores leave with sc1 (MI_SPMM_SC1)
    // optimizer epilogue (mi_adam_args): S's value is the gradient of parameter row r; null p = off
    float4* p;            int64_t ldp4;
    float4* m;
    float4* v;
    const float* reg_w;
    MiAdamConsts adam;
};

// Sparse-operand extensions (mi_spmm_csr_ex_f32); all pointers nullable.
struct Ex {
    const int32_t* x_map;       // [n_cols]: X is compact, neighbour c reads X[x_map[c]]; < 0 = an all-zero row, skipped
    const int32_t* addend_map;  // [n_rows]: addend is compact, row r adds addend[addend_map[r]]; < 0 = nothing
    const int32_t* row_list;    // [n_list]: compute only these rows; Y / S / addend are compact, indexed by list position
    const int32_t* n_list_dev;  // device count of valid row_list entries (<= the launch bound), or null
    const int32_t* long_index;  // [n_rows]: index into the plan's long rows, -1 for short rows (row_list mode)
    int32_t parts;              // mask of MI_SPMM_SHORT_ROWS / MI_SPMM_SPLIT_ROWS (host side only)
    int32_t hot_base, hot_rows, hot_threads;  // persistent short-row launch (host side only): hot_rows 0 = plain launch, > 0 = LDS cache rows, < 0 = no cache
    const uint32_t* x_bits;     // bit c = (x_map[c] >= 0): "live columns are rare" — on a packed plan the work items run as spmm_items_xscan_kernel
    uint8_t* slot_live;         // [plan->n_items], behind the partial rows: 0 = the work item gathered nothing and wrote no partial row
};

// Largest n over the sub-groups of the wavefront (loop bounds must be wave-uniform around __shfl).
template <int LPR>
__device__ __forceinline__ int wave_max_over_subgroups(int n) {
#pragma unroll
    for (int m = MI_WAVE / 2; m >= LPR; m >>= 1) n = max(n, __shfl_xor(n, m, MI_WAVE));
    return n;
}

// The calling sub-group accumulates entries [beg, beg + n) into acc, in list order.
template <int LPR, int VPL, int UNROLL, bool SPARSE>
__device__ __forceinline__ void subgroup_accumulate(const int32_t* __restrict__ col,
                                                    const float* __restrict__ val,
                                                    const float4* __restrict__ X4, int64_t ldx4, int d4,
                                                    int32_t beg, int n, int nmax, int li,
                                                    const int32_t* __restrict__ x_map, float4 (&acc)[VPL]) {
    static_assert(LPR % UNROLL == 0, "a batch of LPR entries is consumed UNROLL at a time");
#pragma unroll
    for (int v = 0; v < VPL; ++v) acc[v] = mi_f4_zero();
    for (int base = 0; base < nmax; base += LPR) {
        int32_t my_c = -1;  // < 0: nothing to gather for this lane's entry
        float my_v = 0.f;
        if (base + li < n) {
            if (MI_SPMM_NT & 1) {
                my_c = __builtin_nontemporal_load(col + beg + base + li);
                my_v = __builtin_nontemporal_load(val + beg + base + li);
            } else {
                my_c = col[beg + base + li];
                my_v = val[beg + base + li];
            }
            if (SPARSE && x_map) my_c = x_map[my_c];  // compact row of X, or < 0 for a row that is all zeros
        }
        if (SPARSE && __ballot(my_c >= 0) == 0ull) continue;
        const int m = min(LPR, nmax - base);
        for (int j = 0; j < m; j += UNROLL) {
            float w[UNROLL];
            float4 x[UNROLL][VPL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int32_t c = __shfl(my_c, j + u, LPR);
                w[u] = __shfl(my_v, j + u, LPR);
                const bool ok = c >= 0;
                const float4* src = X4 + (int64_t)(ok ? c : 0) * ldx4;
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
}

// Y / S rows of one sub-group (li = lane within the sub-group); r = output row (compact position in row_list mode).
template <int LPR, int VPL, bool ADAM>
__device__ __forceinline__ void store_epilogue(const Epilogue& ep, int64_t r, int d4, int li,
                                               const float4 (&acc)[VPL], const float4 (&a)[VPL]) {
    constexpr bool adam = ADAM;
    const float w = (adam && ep.reg_w) ? ep.reg_w[r] : 0.f;  // ADAM launches are their own instantiations: profiles list them apart
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        const int e = li + v * LPR;
        if (e >= d4) continue;
        if (ep.Y) mi_store4<2>(ep.Y + r * ep.ldy4 + e, acc[v], ep.streaming);
        if (ep.S || adam) {
            float4 o;
            o.x = ep.scale * (a[v].x + acc[v].x);
            o.y = ep.scale * (a[v].y + acc[v].y);
            o.z = ep.scale * (a[v].z + acc[v].z);
            o.w = ep.scale * (a[v].w + acc[v].w);
            if (ep.S) mi_store4<2>(ep.S + r * ep.lds4 + e, o, ep.streaming);
            if (adam) {
                const int64_t i = r * d4 + e;
                float4 pp = ep.p[r * ep.ldp4 + e], mm = ep.m[i], vv = ep.v[i];
                mi_adam_update4(pp, o, mm, vv, ep.reg_w != nullptr, w, ep.adam);
                ep.p[r * ep.ldp4 + e] = pp;
                ep.m[i] = mm;
                ep.v[i] = vv;
            }
        }
    }
}

// ar = addend row index, < 0 for "no addend row"
template <int LPR, int VPL>
__device__ __forceinline__ void load_addend(const Epilogue& ep, int64_t ar, int d4, int li, float4 (&a)[VPL]) {
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        const int e = li + v * LPR;
        a[v] = ((ep.S || ep.p) && ep.addend && ar >= 0 && e < d4) ? mi_load4<3>(ep.addend + ar * ep.lda4 + e) : mi_f4_zero();
    }
}

// One sub-group per row; rows with more than `chunk` entries are left to the split path.
// SPARSE = false: the dense product (every entry gathered; addend_map allowed) — the kernel the
// roofline is quoted on.  SPARSE = true: x_map / row_list launches of the fused train step, kept as a
// separate instantiation so that profiles list them apart.
#ifndef MI_SPMM_COOP_MIN
#define MI_SPMM_COOP_MIN 128  // plan-less launches: rows longer than this are summed by the whole workgroup (512: a 500-entry
                              // row was still a 60 us serial walk inside a 31 us average launch, profiles/r2_ranker_v4.md)
#endif
template <int LPR, int VPL, int UNROLL, int RPS, bool SPARSE, bool ADAM>
__device__ __forceinline__ void spmm_rows_body(const int64_t block_id, const int64_t n_blocks, int64_t n_out, int d4,
                                               const int32_t* __restrict__ rowptr,
                                               const int32_t* __restrict__ col,
                                               const float* __restrict__ val,
                                               const float4* __restrict__ X4, int64_t ldx4,
                                               const Epilogue& ep, int32_t chunk, const Ex& ex, int32_t coop) {
    constexpr int NB = MI_WAVE / LPR, SG = NB * kWavesPerBlock;
    // coop (plan-less dense launches only: the per-batch subgraphs of the ranker, which cannot afford a plan's host
    // read-backs): a hub row of such a graph — an article bought by thousands of the batch's users — used to be one
    // sub-group's serial walk (135 us of a 35 us average launch, profiles/r2_ranker_v2.md).  Rows longer than
    // MI_SPMM_COOP_MIN are set aside in the first pass and then summed by all SG sub-groups of the block over equal
    // consecutive pieces, reduced through LDS in sub-group order: fixed association, no atomics on data.
    constexpr bool kCoop = !SPARSE && !ADAM && RPS == 1;
    __shared__ float4 coop_red[kCoop ? SG : 1][VPL][LPR];
    __shared__ int32_t coop_beg[kCoop ? SG : 1], coop_len[kCoop ? SG : 1];
    __shared__ int64_t coop_row[kCoop ? SG : 1];
    __shared__ int coop_n;
    const int lane = mi_lane();
    const int li = lane % LPR;
    const int sgi = (threadIdx.x / MI_WAVE) * NB + lane / LPR;
    const bool listed = SPARSE && ex.row_list != nullptr;
    int64_t n_valid = n_out;
    if (listed && ex.n_list_dev) n_valid = min(n_out, (int64_t)*ex.n_list_dev);
#ifndef MI_SPMM_XCD_RANGES
#define MI_SPMM_XCD_RANGES 0
#endif
    // MI_SPMM_XCD_RANGES: workgroup w (XCD w & 7) takes the (w >> 3)-th block of the XCD's CONTIGUOUS eighth of the rows
    // instead of block w: under the locality order neighbouring rows share their cold gathers, which then meet in one L2.
    // A/B on C2, round 2 (tools/exp_locality.py --reorder cold): traffic 4.27 -> 4.16 GB but 808 -> 1 491 us — the eighths
    // are unequal work (the item rows and the high-degree users sit at the end of the order).  Off.
    int64_t blk = block_id;
    if (MI_SPMM_XCD_RANGES && !listed) {
        const int64_t per = n_blocks / 8;  // the launcher rounds the grid up to a multiple of 8
        blk = (int64_t)(block_id & 7) * per + (block_id >> 3);
    }
    const int64_t base = blk * (SG * RPS);
    if (kCoop && coop) {
        if (threadIdx.x == 0) coop_n = 0;
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < RPS; ++k) {
        const int64_t i = base + k * SG + sgi;  // output position
        int64_t r = 0;
        int32_t beg = 0;
        int n = 0;
        bool mine = i < n_valid;
        if (mine) {
            r = listed ? (int64_t)ex.row_list[i] : i;
            beg = rowptr[r];
            n = rowptr[r + 1] - beg;
            mine = n <= chunk;  // split rows belong to the items / fix-up kernels
        }
        if (kCoop && coop && mine && n > MI_SPMM_COOP_MIN) {
            if (li == 0) {
                const int q = atomicAdd(&coop_n, 1);  // list order is irrelevant: each row is finished on its own
                coop_row[q] = r;
                coop_beg[q] = beg;
                coop_len[q] = n;
            }
            mine = false;
        }
        if (!mine) n = 0;
        const int nmax = wave_max_over_subgroups<LPR>(n);
        float4 a[VPL], acc[VPL];
        const int64_t ar = !mine ? -1 : (listed ? i : (ex.addend_map ? (int64_t)ex.addend_map[r] : r));
        load_addend<LPR, VPL>(ep, ar, d4, li, a);
        subgroup_accumulate<LPR, VPL, UNROLL, SPARSE>(col, val, X4, ldx4, d4, beg, n, nmax, li, ex.x_map, acc);
        if (mine) store_epilogue<LPR, VPL, ADAM>(ep, listed ? i : r, d4, li, acc, a);
    }
    if (kCoop && coop) {
        __syncthreads();
        const int n_long = coop_n;  // block-uniform
        for (int q = 0; q < n_long; ++q) {
            const int32_t beg = coop_beg[q], len = coop_len[q];
            const int64_t r = coop_row[q];
            const int32_t per = (len + SG - 1) / SG;
            const int32_t b = min(sgi * per, len);
            const int n = min(per, len - b);
            const int nmax = wave_max_over_subgroups<LPR>(n);
            float4 acc[VPL];
            subgroup_accumulate<LPR, VPL, UNROLL, false>(col, val, X4, ldx4, d4, beg + b, n, nmax, li, nullptr, acc);
#pragma unroll
            for (int v = 0; v < VPL; ++v) coop_red[sgi][v][li] = acc[v];
            __syncthreads();
            if (sgi == 0) {
                float4 a[VPL];
                load_addend<LPR, VPL>(ep, ex.addend_map ? (int64_t)ex.addend_map[r] : r, d4, li, a);
#pragma unroll
                for (int v = 0; v < VPL; ++v) {
                    float4 t = coop_red[0][v][li];
                    for (int g = 1; g < SG; ++g) t = mi_f4_add(t, coop_red[g][v][li]);
                    acc[v] = t;
                }
                store_epilogue<LPR, VPL, ADAM>(ep, r, d4, li, acc, a);
            }
            __syncthreads();
        }
    }
}
template <int LPR, int VPL, int UNROLL, int RPS, bool SPARSE, bool ADAM>
__global__ __launch_bounds__(kBlock) void spmm_rows_kernel(int64_t n_out, int d4,
                                                           const int32_t* __restrict__ rowptr,
                                                           const int32_t* __restrict__ col,
                                                           const float* __restrict__ val,
                                                           const float4* __restrict__ X4, int64_t ldx4,
                                                           Epilogue ep, int32_t chunk, Ex ex, int32_t coop) {
    spmm_rows_body<LPR, VPL, UNROLL, RPS, SPARSE, ADAM>(blockIdx.x, gridDim.x, n_out, d4, rowptr, col, val, X4, ldx4, ep, chunk, ex, coop);
}
// Twin launch (pairs.hpp): two plan-less dense products in one grid — the first `split` workgroups take product a, the rest
// product b.  The same body as the single launch (same instantiation), so each product's rows are bitwise the single launch's.
struct RowsSide {
    int64_t n_out; int d4;
    const int32_t* rowptr; const int32_t* col; const float* val;
    const float4* X4; int64_t ldx4;
    Epilogue ep;
};
template <int LPR, int UNROLL>
__global__ __launch_bounds__(kBlock) void spmm_rows_pair_kernel(RowsSide a, RowsSide b, unsigned split) {
    // every field picked on its own (scalar selects of kernel arguments): a reference chosen between the two structs makes
    // the compiler copy both into scratch (376 bytes per lane, 94 us instead of 22 for the pair — measured, round 4)
    const bool first = blockIdx.x < split;
#define MI_PICK(f) (first ? a.f : b.f)
    Epilogue ep;
    ep.Y = MI_PICK(ep.Y);           ep.ldy4 = MI_PICK(ep.ldy4);
    ep.addend = MI_PICK(ep.addend); ep.lda4 = MI_PICK(ep.lda4);
    ep.S = MI_PICK(ep.S);           ep.lds4 = MI_PICK(ep.lds4);
    ep.scale = MI_PICK(ep.scale);
    ep.streaming = MI_PICK(ep.streaming);
    ep.p = nullptr; ep.ldp4 = 0; ep.m = nullptr; ep.v = nullptr; ep.reg_w = nullptr;
    ep.adam = MiAdamConsts{};
    const Ex ex = {nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, nullptr, nullptr};
    spmm_rows_body<LPR, 1, UNROLL, MI_SPMM_ROWS_RPS, false, false>(first ? blockIdx.x : blockIdx.x - split, first ? split : gridDim.x - split,
                                                                  MI_PICK(n_out), MI_PICK(d4), MI_PICK(rowptr), MI_PICK(col), MI_PICK(val),
                                                                  MI_PICK(X4), MI_PICK(ldx4), ep, INT32_MAX, ex, 1);
#undef MI_PICK
}


// ---- LDS hot-row cache (mi_spmm_ex.hot_rows) ---------------------------------------------------------------------------
// Rows [base, base + n) of X staged in LDS (byte offset 0 of the workgroup's dynamic LDS), row_bytes each.
struct HotRows {
    int32_t base, n;
    int32_t row_bytes;  // d4 * 16
};

// Eight neighbour rows for the calling lane in ONE issue group: lane holds columns c[0..7] (< 0 = nothing to fetch);
// a column inside the cached range is read from LDS (ds_read_b128), any other from X (global_load_dwordx4), both into the
// same destination registers, and the group ends with a single wait.  Written as one asm block because the compiler turns
// "LDS or global into one value" into a join that waits for every load before the next one is issued (measured on the
// C++ form: s_waitcnt vmcnt(0) lgkmcnt(0) after each entry); here eight loads are in flight per lane, as in the plain
// kernel.  exec is narrowed per entry and restored; vcc is scratch.  xb = X + (lane's float4 column) * 16 as a 64-bit
// address, ld_bytes = row stride of X in bytes (< 2^32), li16 = byte offset of the lane's float4 inside a row.
#define MI_HOT_ENTRY(U)                                                                                              \
    "v_subrev_u32_e32 %[t2], %[hbase], %[c" #U "]\n"      /* t2 = c - base (unsigned test below covers c < base, c < 0) */ \
    "v_cmp_gt_u32_e32 vcc, %[hn], %[t2]\n"                /* vcc: cached row                                          */ \
    "v_cmp_lt_i32_e64 %[m], -1, %[c" #U "]\n"             /* m: a row to fetch at all                                 */ \
    "s_andn2_b64 %[m], %[m], vcc\n"                       /* m: ... from X                                            */ \
    "s_and_b64 exec, %[save], vcc\n"                                                                                 \
    "v_mad_u32_u24 %[t2], %[t2], %[rowb], %[li16]\n"                                                                  \
    "ds_read_b128 %[x" #U "], %[t2]\n"                                                                                \
    "s_and_b64 exec, %[save], %[m]\n"                                                                                 \
    "v_mad_u64_u32 %[t01], vcc, %[c" #U "], %[ldb], %[xb]\n"                                                          \
    "global_load_dwordx4 %[x" #U "], %[t01], off\n"                                                                   \
    "s_mov_b64 exec, %[save]\n"

// Prefetch registers of the persistent launch: filled by loads that are issued with the first gather group of a row and
// complete at that group's wait (see subgroup_accumulate_hot); the compiler never sees them in flight.
struct Prefetch {
    uint64_t rp;   // rowptr[r], rowptr[r + 1] of the wavefront's next-but-one row
    int32_t c;     // first (col, val) batch of its next row, one entry per lane
    float v;
};

template <bool PRE>
__device__ __forceinline__ void hot_gather8(mi_f4v (&x)[8], const int32_t (&c)[8], const HotRows& hot, uint32_t ld_bytes,
                                            uint64_t xb, uint32_t li16, Prefetch& pf, const int32_t* rp_addr,
                                            const int32_t* c_addr, const float* v_addr) {
    uint64_t t01, save, m;
    uint32_t t2;
    if (PRE) {
        // issue only; the block below names pf's registers as read-write operands and ends with the wait, so every
        // later use of them is ordered after it
        asm volatile("global_load_dwordx2 %[rp], %[ra], off\n"
                     "global_load_dword %[pc], %[ca], off\n"
                     "global_load_dword %[pv], %[va], off\n"
                     : [rp] "=&v"(pf.rp), [pc] "=&v"(pf.c), [pv] "=&v"(pf.v)
                     : [ra] "v"(rp_addr), [ca] "v"(c_addr), [va] "v"(v_addr)
                     : "memory");
    }
    asm volatile("s_mov_b64 %[save], exec\n"
                 MI_HOT_ENTRY(0) MI_HOT_ENTRY(1) MI_HOT_ENTRY(2) MI_HOT_ENTRY(3)
                 MI_HOT_ENTRY(4) MI_HOT_ENTRY(5) MI_HOT_ENTRY(6) MI_HOT_ENTRY(7)
                 "s_waitcnt vmcnt(0) lgkmcnt(0)\n"
                 : [x0] "+v"(x[0]), [x1] "+v"(x[1]), [x2] "+v"(x[2]), [x3] "+v"(x[3]), [x4] "+v"(x[4]), [x5] "+v"(x[5]),
                   [x6] "+v"(x[6]), [x7] "+v"(x[7]), [t01] "=&v"(t01), [t2] "=&v"(t2), [save] "=&s"(save), [m] "=&s"(m),
                   [rp] "+v"(pf.rp), [pc] "+v"(pf.c), [pv] "+v"(pf.v)
                 : [c0] "v"(c[0]), [c1] "v"(c[1]), [c2] "v"(c[2]), [c3] "v"(c[3]), [c4] "v"(c[4]), [c5] "v"(c[5]),
                   [c6] "v"(c[6]), [c7] "v"(c[7]), [hbase] "s"(hot.base), [hn] "s"(hot.n), [rowb] "s"(hot.row_bytes),
                   [ldb] "s"(ld_bytes), [xb] "v"(xb), [li16] "v"(li16)
                 : "vcc", "memory");
}
#undef MI_HOT_ENTRY

// The prefetch alone, for a wavefront whose current rows have nothing to gather.
__device__ __forceinline__ void hot_prefetch_only(Prefetch& pf, const int32_t* rp_addr, const int32_t* c_addr,
                                                  const float* v_addr) {
    asm volatile("global_load_dwordx2 %[rp], %[ra], off\n"
                 "global_load_dword %[pc], %[ca], off\n"
                 "global_load_dword %[pv], %[va], off\n"
                 "s_waitcnt vmcnt(0)\n"
                 : [rp] "=&v"(pf.rp), [pc] "=&v"(pf.c), [pv] "=&v"(pf.v)
                 : [ra] "v"(rp_addr), [ca] "v"(c_addr), [va] "v"(v_addr)
                 : "memory");
}

// subgroup_accumulate of the persistent launch: same entries, same order, same arithmetic (one float4 per lane).  The
// first batch of LPR (col, val) pairs arrives in registers (c0, v0: prefetched one row ahead); later ones are read here.
// The first gather group of the call also carries the wavefront's prefetch loads (pf <- *rp_addr, *c_addr, *v_addr).
template <int LPR>
__device__ __forceinline__ void subgroup_accumulate_hot(const int32_t* __restrict__ col, const float* __restrict__ val,
                                                        const float4* __restrict__ X4, int64_t ldx4, int d4, int32_t beg,
                                                        int n, int nmax, int li, int32_t c0, float v0, const HotRows& hot,
                                                        uint32_t lds_base, float4& acc, Prefetch& pf,
                                                        const int32_t* rp_addr, const int32_t* c_addr, const float* v_addr) {
    static_assert(LPR % 8 == 0, "entries are consumed eight at a time");
    acc = mi_f4_zero();
    if (nmax == 0) {  // wave-uniform
        hot_prefetch_only(pf, rp_addr, c_addr, v_addr);
        return;
    }
    const uint64_t xb = (uint64_t)(X4 + li);
    const uint32_t ld_bytes = (uint32_t)(ldx4 * 16), li16 = lds_base + (uint32_t)li * 16u;
    const bool lane_on = li < d4;
    for (int base = 0; base < nmax; base += LPR) {
        int32_t my_c = c0;
        float my_v = v0;
        if (base > 0) {
            my_c = -1;
            my_v = 0.f;
            if (base + li < n) {
                my_c = col[beg + base + li];
                my_v = val[beg + base + li];
            }
        }
        const int m = min(LPR, nmax - base);
        for (int j = 0; j < m; j += 8) {
            float w[8];
            int32_t c[8];
            mi_f4v x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int32_t cu = __shfl(my_c, j + u, LPR);
                w[u] = __shfl(my_v, j + u, LPR);   // 0 where there is no entry
                c[u] = lane_on ? cu : -1;
                x[u] = mi_f4v{0.f, 0.f, 0.f, 0.f};
            }
            if (base == 0 && j == 0) hot_gather8<true>(x, c, hot, ld_bytes, xb, li16, pf, rp_addr, c_addr, v_addr);
            else hot_gather8<false>(x, c, hot, ld_bytes, xb, li16, pf, rp_addr, c_addr, v_addr);
#pragma unroll
            for (int u = 0; u < 8; ++u) mi_f4_fma(acc, w[u], make_float4(x[u].x, x[u].y, x[u].z, x[u].w));
        }
    }
}

// Y / S stores of the persistent launch, issued from inline asm on both paths.  gfx9 has ONE counter for loads and stores
// (vmcnt) and they complete out of order with each other, so a compiler that has a store pending waits with vmcnt(0)
// before it touches the next loaded value: in a persistent loop that puts the store's acknowledgement latency at the
// top of every iteration (measured: no gain over the plain launch).  Stores the compiler does not see leave the loop's
// waits exact; the gather group's own wait (vmcnt(0)) still covers them, overlapped with the gathers.
// s_nop: a VALU write to the data registers of a > 8-byte store needs 2 wait states after it (VMEM store-data hazard).
__device__ __forceinline__ void hot_store4(float4* p, const float4& v, bool streaming) {
    mi_f4v x = {v.x, v.y, v.z, v.w};
#ifdef MI_HOT_NOSTORE  // timing probe only (tools/ab_hot.py): what the loop costs when no store is ever outstanding
    asm volatile("" : : "v"(p), "v"(x));
    return;
#endif
    if (streaming) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(x) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(p), "v"(x) : "memory");
}

__device__ __forceinline__ void hot_store_epilogue(const Epilogue& ep, int64_t r, int d4, int li, const float4& acc,
                                                   const float4& a) {
    if (li >= d4) return;
    if (ep.Y) hot_store4(ep.Y + r * ep.ldy4 + li, acc, ep.streaming && ((MI_SPMM_SC1 >> 2) & 1));
    if (ep.S) {
        float4 o;
        o.x = ep.scale * (a.x + acc.x);
        o.y = ep.scale * (a.y + acc.y);
        o.z = ep.scale * (a.z + acc.z);
        o.w = ep.scale * (a.w + acc.w);
        hot_store4(ep.S + r * ep.lds4 + li, o, ep.streaming && ((MI_SPMM_SC1 >> 2) & 1));
    }
}

// Short rows as a PERSISTENT, software-pipelined launch (dense products of a planned adjacency, one float4 per lane).
// The plain launch is bound by a chain of dependent memory latencies per row — rowptr -> (col, val) -> gathers -> the
// next eight gathers — times rows / resident wavefronts (C2 user rows: ~8.5 us per row pair, 633 us measured for 10^6
// rows; tools/ab_hot.py), not by bytes.  Here the grid is what the chip holds at once and every wavefront strides
// over the row groups (wavefront g of G takes rows g*NB.., (g+G)*NB.., ...), which lets it fetch the row pointers of its
// next-but-one group and the first (col, val) batch of its next group WITH the first gathers of the current one (same
// issue group, one wait): the chain per row shrinks to the gathers themselves.  At any moment the chip still works on
// one contiguous window of rows, and consecutive row groups sit in one workgroup, i.e. one XCD's L2.
// Optional LDS hot-row cache (mi_spmm_ex.hot_rows): every workgroup stages rows [hot_base, hot_base + hot_n) of X in
// LDS once; a gather of such a row is served from there (hot_gather8).  No barrier after the fill, no atomics; the sum
// of a row is the plain kernel's, bit for bit.
struct RowSpan { int32_t beg; int n; bool mine; };

// (beg, n, mine) of the lane's row of group g from its two row pointers; n = 0 for rows that are not this kernel's
template <int LPR>
__device__ __forceinline__ RowSpan hot_span(uint64_t rp, int64_t g, int64_t n_groups, int64_t n_rows, int32_t chunk, int lane) {
    constexpr int NB = MI_WAVE / LPR;
    RowSpan s;
    s.beg = (int32_t)(uint32_t)rp;
    s.n = (int32_t)(uint32_t)(rp >> 32) - s.beg;
    s.mine = g < n_groups && g * NB + lane / LPR < n_rows && s.n <= chunk;  // split rows belong to the sweep / items / fix-up kernels
    if (!s.mine) s.n = 0;
    return s;
}

template <int LPR, int UNROLL, bool ADAM>
__global__ __launch_bounds__(1024) void spmm_rows_hot_kernel(int64_t n_rows, int d4, const int32_t* __restrict__ rowptr,
                                                             const int32_t* __restrict__ col, const float* __restrict__ val,
                                                             const float4* __restrict__ X4, int64_t ldx4, Epilogue ep,
                                                             int32_t chunk, Ex ex, int32_t hot_base, int32_t hot_n) {
    extern __shared__ float4 hot_tab[];  // [hot_n][d4]
    constexpr int NB = MI_WAVE / LPR;
    if (hot_n > 0) {
        const int total = hot_n * d4;
        for (int i = threadIdx.x; i < total; i += blockDim.x) {
            const int h = i / d4, e = i - h * d4;
            hot_tab[i] = X4[(int64_t)(hot_base + h) * ldx4 + e];
        }
        __syncthreads();
    }
    const HotRows hot{hot_base, hot_n, d4 * 16};
    // byte address of the table inside the workgroup's LDS (0 today: the kernel has no static LDS; not assumed)
    // (the low half of an LDS object's generic address is its offset in the aperture)
    const uint32_t lds_base = (uint32_t)(uintptr_t)hot_tab;
    const int lane = mi_lane();
    const int li = lane % LPR;
    const int waves_per_wg = blockDim.x / MI_WAVE;
    const int64_t n_groups = (n_rows + NB - 1) / NB;
    const int64_t stride = (int64_t)gridDim.x * waves_per_wg;
    int64_t g = (int64_t)blockIdx.x * waves_per_wg + threadIdx.x / MI_WAVE;
    // addresses that are always readable: a row past the end reads the last row's pointers, a lane past its row's entries
    // the row's last entry, a row without entries the entry before it (position 0 when there is none) — validity is
    // decided arithmetically afterwards, so the prefetch loads carry no branch
    auto rp_addr_of = [&](int64_t gg) {
        const int64_t r = min(gg * NB + lane / LPR, n_rows - 1);
        return rowptr + r;
    };
    auto cv_index_of = [&](const RowSpan& sp) {
        return sp.n > 0 ? (int64_t)sp.beg + min(li, sp.n - 1) : (int64_t)max(sp.beg - 1, 0);
    };
    // pipeline: `cur` = the group being summed (its first (col, val) batch in c0 / v0), `nxt` = the one after it
    RowSpan cur, nxt;
    {
        const int32_t* a0 = rp_addr_of(g);
        const int32_t* a1 = rp_addr_of(g + stride);
        cur = hot_span<LPR>((uint64_t)(uint32_t)a0[0] | ((uint64_t)(uint32_t)a0[1] << 32), g, n_groups, n_rows, chunk, lane);
        nxt = hot_span<LPR>((uint64_t)(uint32_t)a1[0] | ((uint64_t)(uint32_t)a1[1] << 32), g + stride, n_groups, n_rows, chunk, lane);
    }
    int32_t c0 = -1;
    float v0 = 0.f;
    if (li < cur.n) {
        c0 = col[cur.beg + li];
        v0 = val[cur.beg + li];
    }
    for (; g < n_groups; g += stride) {
        const int64_t r = g * NB + lane / LPR;
        const int nmax = wave_max_over_subgroups<LPR>(cur.n);
        float4 a[1], acc[1];
        const int64_t ar = !cur.mine ? -1 : (ex.addend_map ? (int64_t)ex.addend_map[r] : r);
        load_addend<LPR, 1>(ep, ar, d4, li, a);
        Prefetch pf;
        const int64_t ci = cv_index_of(nxt);
        subgroup_accumulate_hot<LPR>(col, val, X4, ldx4, d4, cur.beg, cur.n, nmax, li, c0, v0, hot, lds_base, acc[0], pf,
                                     rp_addr_of(g + 2 * stride), col + ci, val + ci);
        if (cur.mine) {
            if (ADAM) store_epilogue<LPR, 1, ADAM>(ep, r, d4, li, acc, a);
            else hot_store_epilogue(ep, r, d4, li, acc[0], a[0]);
        }
        c0 = li < nxt.n ? pf.c : -1;
        v0 = li < nxt.n ? pf.v : 0.f;
        cur = nxt;
        nxt = hot_span<LPR>(pf.rp, g + 2 * stride, n_groups, n_rows, chunk, lane);
    }
}

// Split rows: one sub-group per work item (row, begin, end, slot) -> partial[slot, :]; slot < 0 = padding.
// Launch slot -> workgroup: linear for a row-major plan; for a banded plan the launch array is made of
// blocks of kPlanGroup slots dealt to 8 queues (queue = band mod 8), and workgroup w serves queue w mod 8.
template <int LPR, int VPL, int UNROLL, int RPS, bool SPARSE>
__global__ __launch_bounds__(kBlock) void spmm_items_kernel(int64_t n_launch, int32_t banded, int d4,
                                                            const int4* __restrict__ items,
                                                            const int32_t* __restrict__ epos,
                                                            const int32_t* __restrict__ col,
                                                            const float* __restrict__ val,
                                                            const float4* __restrict__ X4, int64_t ldx4,
                                                            float4* __restrict__ partial,
                                                            const int32_t* __restrict__ x_map, bool streaming) {
    constexpr int NB = MI_WAVE / LPR, SG = NB * kWavesPerBlock, G = SG * RPS;
    static_assert(kPlanGroup % G == 0, "a workgroup serves a whole fraction of a launch block");
    const int lane = mi_lane();
    const int li = lane % LPR;
    const int sgi = (threadIdx.x / MI_WAVE) * NB + lane / LPR;
    int64_t base;
    if (banded) {
        constexpr int per = kPlanGroup / G;
        const int64_t x = blockIdx.x & 7, tq = blockIdx.x >> 3;
        base = ((tq / per) * 8 + x) * kPlanGroup + (tq % per) * G;
    } else {
        base = (int64_t)blockIdx.x * G;
    }
#pragma unroll
    for (int k = 0; k < RPS; ++k) {
        const int64_t q = base + k * SG + sgi;
        int4 it = make_int4(0, 0, 0, -1);
        int32_t first = 0;
        if (q < n_launch) {
            it = items[q];
            first = epos ? epos[q] : it.y;   // packed plan: col / val are the plan's launch-ordered copies
        }
        const int32_t slot = it.w;
        const int n = slot >= 0 ? it.z - it.y : 0;
        const int nmax = wave_max_over_subgroups<LPR>(n);
        float4 acc[VPL];
        subgroup_accumulate<LPR, VPL, UNROLL, SPARSE>(col, val, X4, ldx4, d4, first, n, nmax, li, x_map, acc);
        if (slot >= 0) {
#pragma unroll
            for (int v = 0; v < VPL; ++v) {
                const int e = li + v * LPR;
                if (e < d4) mi_store4<1>(partial + (int64_t)slot * d4 + e, acc[v], streaming);
            }
        }
    }
}

// Work items with a MAPPED operand whose live columns are rare (the first backward product of the fused train step: x_map
// names the batch's 131 K users among 8 M columns; ~4 % of the split rows' entries are live, 60 % of the work items have
// none).  The plain kernel above pays one dependent chain per work item (descriptor -> (col, val) -> map -> rows -> store:
// ~5 us of wavefront lifetime for two work items; 2.2 ms for C4's 7.7 M).  Taking four work items per sub-group into that
// chain (first version of this round) only moved it to 1.8 ms: the launch stays bound by wavefront lifetime x generations.
// So this kernel does not walk work items at all.  It needs a PACKED plan (mi_spmm_plan.epos / ecol / eval: the entries in
// launch order): a wavefront owns 64 consecutive launch slots, whose entries are ONE contiguous run of the packed arrays,
// and scans that run in tiles of 512 entries — 8 coalesced column loads per lane in flight, then the 8 bit tests
// (x_bits: the 1 MB bitmap of the map sits in every XCD's L2), then map and value of the few live entries — compacting the
// live ones, in list order, into LDS with the slot each belongs to (a search over the 64 slot starts, live entries only).
// The sub-groups then walk that short list (slot s belongs to sub-group s mod NB), four row gathers in flight, summing
// per slot in list order — the sum the plain kernel forms, bit for bit, since a dead entry added 0 * 0 — and write a
// partial row when the slot changes.  A slot without a live entry gets no partial row, only its flag (slot_live).
// Measured on C4 (profiles/r04_c4_item_rows_experiments.md): alone the kernel takes about as long as the chained forms
// (2.0 ms: 0.37 ms of scan, the rest is the walk's gather latency in the hub rows' 32-tile runs), but it keeps few loads
// in flight and moves a seventh of the bytes, so beside the short-row kernel on the other stream it all but disappears:
// sparse launch 6.30 -> 5.36 ms, step 52.8 -> 50.95 ms, where the chained forms got 52.8 -> 52.2.
#ifndef MI_SPMM_XSCAN_U
#define MI_SPMM_XSCAN_U 4   // row gathers in flight per sub-group
#endif
template <int LPR, int VPL>
__global__ __launch_bounds__(kBlock) void spmm_items_xscan_kernel(int64_t n_launch, int d4, const int4* __restrict__ items,
                                                                  const int32_t* __restrict__ epos,
                                                                  const int32_t* __restrict__ ecol,
                                                                  const float* __restrict__ eval,
                                                                  const float4* __restrict__ X4, int64_t ldx4,
                                                                  float4* __restrict__ partial,
                                                                  const int32_t* __restrict__ x_map,
                                                                  const uint32_t* __restrict__ x_bits,
                                                                  uint8_t* __restrict__ slot_live, bool streaming) {
    constexpr int NB = MI_WAVE / LPR;   // sub-groups per wavefront
    constexpr int UNR = 8, T = UNR * MI_WAVE;
    constexpr int U = MI_SPMM_XSCAN_U;
    __shared__ int32_t s_first[kWavesPerBlock][MI_WAVE + 1];
    __shared__ int32_t s_slot[kWavesPerBlock][MI_WAVE];
    __shared__ uint8_t s_has[kWavesPerBlock][MI_WAVE];
    __shared__ uint8_t l_slot[kWavesPerBlock][T];
    __shared__ int32_t l_m[kWavesPerBlock][T];
    __shared__ float l_v[kWavesPerBlock][T];
    const int lane = mi_lane(), wave = threadIdx.x / MI_WAVE;
    const int li = lane % LPR, g = lane / LPR;
    const int64_t q0 = ((int64_t)blockIdx.x * kWavesPerBlock + wave) * MI_WAVE;
    if (q0 >= n_launch) return;   // the whole wavefront; nothing below synchronises across wavefronts
    auto wave_sync = [] {          // LDS traffic of one wavefront is in order; the compiler must not move it across
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    const int64_t q = q0 + lane;
    int4 it = make_int4(0, 0, 0, -1);
    int32_t first = 0;
    if (q < n_launch) {
        it = items[q];
        first = epos[q];
    }
    const int32_t len = it.w >= 0 ? it.z - it.y : 0;
    const int32_t e_begin = __shfl(first, 0, MI_WAVE);
    int32_t e_end = q < n_launch ? first + len : 0;   // epos ascends with the slot: the run ends where the last slot ends
#pragma unroll
    for (int m = MI_WAVE / 2; m > 0; m >>= 1) e_end = max(e_end, __shfl_xor(e_end, m, MI_WAVE));
    s_first[wave][lane] = q < n_launch ? first : e_end;
    if (lane == 0) s_first[wave][MI_WAVE] = e_end;
    s_slot[wave][lane] = it.w;
    s_has[wave][lane] = 0;
    wave_sync();
    int cur = -1;   // the slot (0..63 of this wavefront) whose sum this sub-group holds
    float4 acc[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) acc[v] = mi_f4_zero();
    auto flush = [&] {
        const int32_t slot = s_slot[wave][cur];
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            const int e = li + v * LPR;
            if (e < d4) mi_store4<1>(partial + (int64_t)slot * d4 + e, acc[v], streaming);
            acc[v] = mi_f4_zero();
        }
    };
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    for (int32_t t0 = e_begin; t0 < e_end; t0 += T) {
        // ---- scan one tile: columns, bit tests, then map + value of the live entries, compacted in list order
        int32_t c[UNR];
        bool live[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int32_t i = t0 + u * MI_WAVE + lane;
            c[u] = i < e_end ? __builtin_nontemporal_load(ecol + i) : -1;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) live[u] = c[u] >= 0 && ((x_bits[c[u] >> 5] >> (c[u] & 31)) & 1u);
        int32_t mm[UNR];
        float vv[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            mm[u] = -1;
            vv[u] = 0.f;
            if (live[u]) {
                mm[u] = x_map[c[u]];
                vv[u] = __builtin_nontemporal_load(eval + t0 + u * MI_WAVE + lane);
            }
        }
        int cnt = 0;   // wavefront-uniform
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const unsigned long long b = __ballot(live[u]);
            if (live[u]) {
                const int32_t i = t0 + u * MI_WAVE + lane;
                int lo = 0, hi = MI_WAVE;   // the LAST slot whose start is <= i: of a run of slots with equal starts (padding,
                while (hi - lo > 1) {       // empty work items) all but the last are empty, so this one owns entry i
                    const int mid = (lo + hi) >> 1;
                    if (s_first[wave][mid] <= i) lo = mid; else hi = mid;
                }
                const int pos = cnt + __popcll(b & lt_mask);
                l_slot[wave][pos] = (uint8_t)lo;
                l_m[wave][pos] = mm[u];
                l_v[wave][pos] = vv[u];
                s_has[wave][lo] = 1;
            }
            cnt += __popcll(b);
        }
        wave_sync();
        // ---- walk the tile's live list: sub-group g takes the entries of slots s with s % NB == g, in order
        int idx = 0;   // sub-group-uniform cursor
        while (__ballot(idx < cnt) != 0ull) {
            // the next LPR list entries: which are this sub-group's?
            const int p = idx + li;
            const bool own = p < cnt && (int)(l_slot[wave][p] & (NB - 1)) == g;
            unsigned long long ob = __ballot(own);
            if constexpr (LPR < 64) ob = (ob >> (g * LPR)) & ((1ull << LPR) - 1ull);
            int e[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                ok[u] = ob != 0ull;
                e[u] = ok[u] ? idx + __ffsll((long long)ob) - 1 : 0;
                ob &= ob - 1ull;
            }
            // the cursor moves behind the U-th own entry, or over the whole window when it held fewer (idx >= cnt ends the walk)
            const int next = ob != 0ull ? e[U - 1] + 1 : idx + LPR;
            float w[U];
            int sl[U];
            float4 x[U][VPL];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int32_t m = ok[u] ? l_m[wave][e[u]] : 0;
                w[u] = ok[u] ? l_v[wave][e[u]] : 0.f;
                sl[u] = ok[u] ? (int)l_slot[wave][e[u]] : 0;
                const float4* src = X4 + (int64_t)m * ldx4;
#pragma unroll
                for (int v = 0; v < VPL; ++v) {
                    const int el = li + v * LPR;
                    x[u][v] = (ok[u] && el < d4) ? src[el] : mi_f4_zero();
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (ok[u]) {
                    if (sl[u] != cur) {
                        if (cur >= 0) flush();
                        cur = sl[u];
                    }
#pragma unroll
                    for (int v = 0; v < VPL; ++v) mi_f4_fma(acc[v], w[u], x[u][v]);
                }
            }
            idx = next;
        }
        wave_sync();   // the list is rewritten by the next tile
    }
    if (cur >= 0) flush();
    if (it.w >= 0) slot_live[it.w] = s_has[wave][lane];
}

// bits[w] bit b = (map[32 w + b] >= 0)
__global__ __launch_bounds__(kBlock) void map_live_bits_kernel(int64_t n, const int32_t* __restrict__ map, uint32_t* __restrict__ bits) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool live = i < n && map[i] >= 0;
    const unsigned long long b = __ballot(live);
    const int lane = mi_lane();
    if ((lane & 31) == 0 && i < n) bits[i >> 5] = (uint32_t)(b >> lane);
}

// Split rows, SWEEP form (mi_spmm_sweep): workgroup w serves XCD w & 7 and holds 32 consecutive streams of it; every
// sub-group walks its own stream front to back — (col, val) read LPR at a time, coalesced; rows gathered UNROLL at a
// time — adding each product into the accumulator its column's tag names.  A run of entries with the same tag is summed
// in registers and meets its LDS accumulator once.  Streams are sorted by band, so the XCD's sub-groups gather from
// the same band of X at about the same time without any synchronisation (nothing depends on it but the L2 hit rate).
#ifndef MI_SWEEP_MAX_POLLS
#define MI_SWEEP_MAX_POLLS 64
#endif
template <int LPR, int UNROLL, bool SPARSE>
__global__ __launch_bounds__(1024) void spmm_sweep_kernel(mi_spmm_sweep sw, int d4, const float4* __restrict__ X4, int64_t ldx4,
                                                          float4* __restrict__ partial, const int32_t* __restrict__ x_map,
                                                          bool streaming) {
    extern __shared__ float4 sweep_acc[];  // [32 sub-groups][8 accumulators][LPR]
    constexpr int NB = MI_WAVE / LPR;
    constexpr int kColMask = 0x07FFFFFF;
    const int lane = mi_lane();
    const int li = lane % LPR;
    const int sg = (threadIdx.x / MI_WAVE) * NB + lane / LPR;  // 0..31
    const int x = blockIdx.x & 7, wg = blockIdx.x >> 3;
    const int k = wg * 32 + sg;
    float4* mine = sweep_acc + sg * 8 * LPR;
#pragma unroll
    for (int q = 0; q < 8; ++q) mine[q * LPR + li] = mi_f4_zero();
    const int32_t beg = sw.stream_ptr[x * sw.n_streams + k];
    const int n = sw.stream_ptr[x * sw.n_streams + k + 1] - beg;
    const int nmax = wave_max_over_subgroups<LPR>(n);
    // Pacing (performance only, never correctness).  Streams are sorted by band and balanced per band, but wavefronts
    // run at their own speed and nothing pulls a slow one back: measured without pacing every row of X is fetched 3.1
    // times although an XCD's L2 keeps a whole 4 MB region under 32 concurrent readers (tools/probes/l2_merge.hip).
    // Communication between 512 wavefronts per XCD costs more than it saves (progress counters + polling: 2.7 ms).
    // So every wavefront follows the same TIME TABLE instead: band t is not started before t0 + t * pace ticks of the
    // device-wide constant-rate clock (s_memrealtime, 100 MHz); a wavefront that is late simply does not wait.
    const uint64_t t0 = wall_clock64();
    const uint64_t pace = (uint64_t)sw.slack;  // ticks per band; 0 = off
    int my_band = 0;
    __syncthreads();
    int cur = -1;                 // tag of the run being summed in registers
    float4 run = mi_f4_zero();
    for (int base = 0; base < nmax; base += LPR) {
        int32_t my_c = -1;
        float my_v = 0.f;
        if (base + li < n) {
            my_c = sw.col[beg + base + li];
            my_v = sw.val[beg + base + li];
        }
#ifdef MI_SWEEP_WG_SYNC
        // EXPERIMENT (profiles/r03_sweep.md): the 16 wavefronts of a workgroup advance band by band
        // in lockstep — a wavefront whose first stream enters band t first passes barriers up to t; every wavefront
        // executes exactly sw.epoch (= bands per XCD) barriers in all.  Band = 2 048 columns, XCD-interleaved.
        {
            const int first_c = __builtin_amdgcn_readfirstlane(my_c);
            const int tb0 = first_c >= 0 ? (((first_c & kColMask) >> 11) >> 3) : sw.epoch;
            while (my_band < tb0 && my_band < sw.epoch) { __syncthreads(); ++my_band; }
        }
#endif
        unsigned long long band_starts = 0ull;  // bit j: entry j of this batch opens a band (first sub-group's stream)
        if (pace) {
            const unsigned long long fl = __ballot(my_c >= 0 && (my_c & 0x08000000));
            band_starts = (LPR == MI_WAVE) ? fl : (fl & ((1ull << LPR) - 1));
        }
        if (SPARSE && x_map && my_c >= 0) {
            const int32_t mapped = x_map[my_c & kColMask];
            my_c = mapped < 0 ? -1 : ((my_c & 0x70000000) | mapped);
        }
        if (!(SPARSE && __ballot(my_c >= 0) == 0ull)) {
            const int m = min(LPR, nmax - base);
            for (int j = 0; j < m; j += UNROLL) {
                if (pace) {  // wave-uniform: a group of UNROLL entries that opens a band waits for that band's slot
                    const int opens = __popcll((band_starts >> j) & ((1ull << UNROLL) - 1));
                    if (opens) {
                        my_band += opens;  // EXACT band count of the wavefront's first sub-group; the others ride along
                        const uint64_t due = t0 + (uint64_t)(my_band - 1) * pace;
                        while (wall_clock64() < due) __builtin_amdgcn_s_sleep(2);
                    }
                }
                float w[UNROLL];
                float4 xr[UNROLL];
                int tg[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    const int32_t c = __shfl(my_c, j + u, LPR);
                    w[u] = __shfl(my_v, j + u, LPR);
                    const bool ok = c >= 0;
                    tg[u] = ok ? (c >> 28) : -1;
                    const float4* src = X4 + (int64_t)(ok ? (c & kColMask) : 0) * ldx4;
                    xr[u] = (ok && li < d4) ? src[li] : mi_f4_zero();
                }
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    if (tg[u] < 0) continue;
                    if (tg[u] != cur) {   // sub-group-uniform: the tag was broadcast
                        if (cur >= 0) mine[cur * LPR + li] = mi_f4_add(mine[cur * LPR + li], run);
                        cur = tg[u];
                        run = mi_f4_zero();
                    }
                    mi_f4_fma(run, w[u], xr[u]);
                }
            }
        }
    }
#ifdef MI_SWEEP_WG_SYNC
    while (my_band < sw.epoch) { __syncthreads(); ++my_band; }
#endif
    if (cur >= 0) mine[cur * LPR + li] = mi_f4_add(mine[cur * LPR + li], run);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int32_t slot = sw.slot_of[k * 8 + q];
        if (slot >= 0 && li < d4) mi_store4<1>(partial + ((int64_t)slot * 8 + x) * d4 + li, mine[q * LPR + li], streaming);
    }
}

// One 256-thread block per split row.  Its 4 wavefronts x NB sub-groups stride over the row's
// partial sums (several loads in flight each), then combine through LDS in a fixed order, so the
// result does not depend on scheduling.  Wave 0 applies the epilogue.
template <int LPR, int VPL, bool SPARSE, bool ADAM>
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
    constexpr int FU = MI_SPMM_FIXUP_UNROLL / VPL;  // partial rows in flight per lane
    constexpr int kTile = 2048;                      // a multiple of NSG * FU: the tiles cut the walk below between two of its trips
    static_assert(kTile % (NSG * FU) == 0, "tile = whole trips");
    __shared__ uint8_t live_sh[SPARSE ? kTile : 1];
    const bool flagged = SPARSE && ex.slot_live != nullptr;   // mapped operand with x_bits: a work item that gathered nothing wrote no partial row
    for (int32_t t0 = sb; t0 < se; t0 += kTile) {
        const int32_t t1 = min(se, t0 + kTile);
        if (flagged) {   // block-uniform: the row's flags of this tile, one coalesced read, instead of a flag load in front of every row load
            __syncthreads();
            for (int32_t i = t0 + (int32_t)threadIdx.x; i < t1; i += kBlock) live_sh[i - t0] = ex.slot_live[i];
            __syncthreads();
        }
        for (int32_t s = t0 + sg; s < t1; s += NSG * FU) {
            float4 x[FU][VPL];
#pragma unroll
            for (int u = 0; u < FU; ++u)
#pragma unroll
                for (int v = 0; v < VPL; ++v) {
                    const int e = li + v * LPR;
                    const int32_t ss = s + u * NSG;
                    bool take = ss < t1 && e < d4;
                    if (SPARSE && flagged && take) take = live_sh[ss - t0] != 0;
                    x[u][v] = take ? partial[(int64_t)ss * d4 + e] : mi_f4_zero();
                }
#pragma unroll
            for (int u = 0; u < FU; ++u)
#pragma unroll
                for (int v = 0; v < VPL; ++v) acc[v] = mi_f4_add(acc[v], x[u][v]);
        }
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
    if (wave != 0 || lane >= LPR) return;
    float4 a[VPL];
    load_addend<LPR, VPL>(ep, ar, d4, lane, a);
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        float4 t = red[0][v][lane];
#pragma unroll
        for (int w = 1; w < kWavesPerBlock; ++w) t = mi_f4_add(t, red[w][v][lane]);
        acc[v] = t;
    }
    store_epilogue<LPR, VPL, ADAM>(ep, out_row, d4, lane, acc, a);
}

// ---- plan construction --------------------------------------------------------------------
// Phase 1 (mi_spmm_plan_count) keeps its arrays in the caller's workspace; phase 2 (mi_spmm_plan_fill)
// carves the same layout again from the counts in mi_spmm_plan_info.
struct PlanWs {
    int32_t *is_long, *ldeg, *long_off, *lnz_off;  // [n_rows + 1]
    int32_t* qstart;                               // [16]
    char* tmp; size_t tmp_bytes;                   // rocPRIM scratch
    int32_t *lrows, *lprefix;                      // [n_long + 1]
    int32_t *flags, *seg_of;                       // [nnz_long]
    int32_t* seg_start;                            // [n_seg]
    uint64_t *keys0, *keys1;                       // [n_seg]
};

size_t plan_tmp_bytes(int64_t n_max) { return mi_align_up((size_t)n_max * 2 + ((size_t)32 << 20), 256); }

bool plan_carve(MiArena& ar, int64_t n_rows, int64_t nnz_total, int64_t n_long, int64_t nnz_long, int64_t n_seg,
                PlanWs& w) {
    const size_t n1 = (size_t)n_rows + 1;
    w.is_long = ar.take<int32_t>(n1);
    w.ldeg = ar.take<int32_t>(n1);
    w.long_off = ar.take<int32_t>(n1);
    w.lnz_off = ar.take<int32_t>(n1);
    w.qstart = ar.take<int32_t>(16);
    w.tmp_bytes = plan_tmp_bytes(nnz_total > n_rows + 1 ? nnz_total : n_rows + 1);
    w.tmp = ar.take<char>(w.tmp_bytes);
    if (!w.is_long || !w.ldeg || !w.long_off || !w.lnz_off || !w.qstart || !w.tmp) return false;
    if (n_long < 0) return true;
    w.lrows = ar.take<int32_t>((size_t)n_long + 1);
    w.lprefix = ar.take<int32_t>((size_t)n_long + 1);
    w.flags = ar.take<int32_t>((size_t)(nnz_long > 0 ? nnz_long : 1));
    w.seg_of = ar.take<int32_t>((size_t)(nnz_long > 0 ? nnz_long : 1));
    if (!w.lrows || !w.lprefix || !w.flags || !w.seg_of) return false;
    if (n_seg < 0) return true;
    w.seg_start = ar.take<int32_t>((size_t)(n_seg > 0 ? n_seg : 1));
    w.keys0 = ar.take<uint64_t>((size_t)(n_seg > 0 ? n_seg : 1));
    w.keys1 = ar.take<uint64_t>((size_t)(n_seg > 0 ? n_seg : 1));
    return w.seg_start && w.keys0 && w.keys1;
}

__global__ void plan_flags_kernel(int64_t n_rows, const int32_t* __restrict__ rowptr, int32_t chunk, int32_t max_deg,
                                  int32_t* __restrict__ is_long, int32_t* __restrict__ ldeg) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_rows) return;
    const int32_t deg = (r < n_rows) ? rowptr[r + 1] - rowptr[r] : 0;
    const bool lg = deg > chunk && deg <= max_deg;  // rows above max_deg belong to another plan (the SWEEP half of a hybrid)
    is_long[r] = lg ? 1 : 0;
    ldeg[r] = lg ? deg : 0;
}

__global__ void plan_long_rows_kernel(int64_t n_rows, const int32_t* __restrict__ is_long,
                                      const int32_t* __restrict__ long_off, const int32_t* __restrict__ lnz_off,
                                      int32_t* __restrict__ lrows, int32_t* __restrict__ lprefix) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_rows) return;
    if (r == n_rows) {
        lprefix[long_off[r]] = lnz_off[r];  // sentinel: entries of all long rows
        return;
    }
    if (is_long[r]) {
        lrows[long_off[r]] = (int32_t)r;
        lprefix[long_off[r]] = lnz_off[r];
    }
}

// index i with prefix[i] <= t < prefix[i + 1]  (prefix has n + 1 ascending entries)
__device__ __forceinline__ int32_t plan_find(const int32_t* __restrict__ prefix, int32_t n, int32_t t) {
    int32_t lo = 0, hi = n;  // answer in [lo, hi)
    while (hi - lo > 1) {
        const int32_t mid = (lo + hi) >> 1;
        if (prefix[mid] <= t) lo = mid; else hi = mid;
    }
    return lo;
}

// flags[t] = 1 when entry t of the long rows' entry list starts a work item: the first entry of a row, the
// first entry of a band within the row, and every chunk-th entry after either.
// min_per > 1 (A/B only, LAPLACE_SPMM_BAND_MIN_PER): a row cuts at multiples of band * 2^j instead, j the smallest for which a
// cut of the row holds min_per entries on average.  The idea: every work item costs a 512-byte partial row written and read
// back by the fix-up kernel, and a row with about one entry per band (C4's typical item: 10^3 entries over 990 bands) pays
// that for EVERY entry — 11.3 M partial rows, 5.8 GB written + 6 GB read per launch, a quarter of the launch's traffic.
// Measured on C4 at N = 1 (tools/ab_c4_band.sh, round 3): min_per 1 / 8 / 32 / 128 -> dense launch 9.78 / 11.89 / 14.85 /
// 17.45 ms, step 57.2 / 66.1 / 80.9 / 94.1 ms (same loss to the last digit).  The band is not about a row reusing its own
// gathers: the work items of one band run together on one XCD, and it is the OTHER rows' gathers of the same 8 192 user
// rows that meet in that L2.  A thin row cut wider leaves that company and every gather goes to memory; the partial rows
// are the cheaper evil.  Stays 1.
__global__ void plan_seg_flags_kernel(int32_t nnz_long, int32_t n_long, const int32_t* __restrict__ lrows,
                                      const int32_t* __restrict__ lprefix, const int32_t* __restrict__ rowptr,
                                      const int32_t* __restrict__ col, int32_t chunk, int32_t band, int64_t n_cols,
                                      int32_t min_per, int32_t* __restrict__ flags) {
    const int32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nnz_long) return;
    const int32_t i = plan_find(lprefix, n_long, t);
    const int32_t rb = rowptr[lrows[i]];
    const int32_t p = rb + (t - lprefix[i]);
    int32_t run = rb;  // first entry of p's band within the row
    if (band > 0) {
        if (min_per > 1) {
            const int64_t deg = rowptr[lrows[i] + 1] - rb;
            while ((int64_t)band * deg < (int64_t)min_per * n_cols && band < (1 << 29)) band <<= 1;
        }
        const int32_t first_col = col[p] / band * band;
        int32_t lo = rb, hi = p;  // smallest index in [rb, p] whose column is >= first_col (columns ascend within a row)
        while (lo < hi) {
            const int32_t mid = (lo + hi) >> 1;
            if (col[mid] >= first_col) hi = mid; else lo = mid + 1;
        }
        run = lo;
    }
    flags[t] = (p == rb || (p - run) % chunk == 0) ? 1 : 0;
}

__global__ void plan_seg_starts_kernel(int32_t nnz_long, int32_t n_long, const int32_t* __restrict__ lrows,
                                       const int32_t* __restrict__ lprefix, const int32_t* __restrict__ rowptr,
                                       const int32_t* __restrict__ col, int32_t band,
                                       const int32_t* __restrict__ flags, const int32_t* __restrict__ seg_of,
                                       int32_t* __restrict__ seg_start, uint64_t* __restrict__ keys) {
    const int32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nnz_long || !flags[t]) return;
    const int32_t s = seg_of[t];
    seg_start[s] = t;
    if (band > 0) {
        const int32_t i = plan_find(lprefix, n_long, t);
        const int32_t p = rowptr[lrows[i]] + (t - lprefix[i]);
        const uint64_t b = (uint64_t)(col[p] / band);
        keys[s] = ((b & 7ull) << 60) | (b << 32) | (uint64_t)(uint32_t)s;  // (queue, band, row-major slot)
    }
}

__global__ void plan_queue_bounds_kernel(int32_t n_seg, const uint64_t* __restrict__ keys, int32_t* __restrict__ qstart) {
    const int x = threadIdx.x;
    if (x > 8) return;
    const uint64_t want = (uint64_t)x << 60;
    int32_t lo = 0, hi = n_seg;
    while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if (keys[mid] >= want) hi = mid; else lo = mid + 1;
    }
    qstart[x] = (x == 8) ? n_seg : lo;
}

__global__ void plan_fill_rows_kernel(int64_t n_rows, const int32_t* __restrict__ is_long,
                                      const int32_t* __restrict__ long_off, int32_t* __restrict__ long_rows,
                                      int32_t* __restrict__ long_index) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    if (is_long[r]) long_rows[long_off[r]] = (int32_t)r;
    if (long_index) long_index[r] = is_long[r] ? long_off[r] : -1;
}

__global__ void plan_item_ptr_kernel(int32_t n_long, int32_t n_seg, const int32_t* __restrict__ lprefix,
                                     const int32_t* __restrict__ seg_of, int32_t* __restrict__ item_ptr) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_long) return;
    item_ptr[i] = (i < n_long) ? seg_of[lprefix[i]] : n_seg;
}

__global__ void plan_pad_kernel(int64_t n_launch, int4* __restrict__ items) {
    int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n_launch) items[q] = make_int4(0, 0, 0, -1);
}

struct QueueStarts { int32_t q[9]; };

__global__ void plan_items_kernel(int32_t n_seg, int32_t banded, const uint64_t* __restrict__ keys, QueueStarts qs,
                                  const int32_t* __restrict__ seg_start, int32_t n_long,
                                  const int32_t* __restrict__ lrows, const int32_t* __restrict__ lprefix,
                                  const int32_t* __restrict__ rowptr, int4* __restrict__ items) {
    const int32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_seg) return;
    int32_t s = j;
    int64_t pos = j;
    if (banded) {
        const uint64_t k = keys[j];
        s = (int32_t)(uint32_t)(k & 0xFFFFFFFFull);
        const int x = (int)(k >> 60);
        const int64_t rank = j - qs.q[x];
        pos = ((rank / kPlanGroup) * 8 + x) * kPlanGroup + rank % kPlanGroup;
    }
    const int32_t t = seg_start[s];
    const int32_t i = plan_find(lprefix, n_long, t);
    const int32_t r = lrows[i];
    const int32_t p = rowptr[r] + (t - lprefix[i]);
    const int32_t t_next = (s + 1 < n_seg) ? seg_start[s + 1] : INT32_MAX;
    const int32_t end = (t_next < lprefix[i + 1]) ? p + (t_next - t) : rowptr[r + 1];
    items[pos] = make_int4(r, p, end, s);
}

// mi_spmm_plan_pack_entries: entries of launch slot q -> [epos[q], epos[q] + len) of the packed arrays
__global__ void plan_item_len_kernel(int64_t n, const int4* __restrict__ items, int32_t* __restrict__ len) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const int4 it = items[q];
    len[q] = it.w >= 0 ? it.z - it.y : 0;
}
__global__ __launch_bounds__(256) void plan_pack_kernel(int64_t n, const int4* __restrict__ items, const int32_t* __restrict__ epos,
                                                        int32_t nnz_long, const int32_t* __restrict__ col,
                                                        const float* __restrict__ val, int32_t* __restrict__ ecol,
                                                        float* __restrict__ eval) {
    const int64_t q = (int64_t)blockIdx.x * 8 + threadIdx.x / 32;
    if (q >= n) return;
    const int4 it = items[q];
    if (it.w < 0) return;
    const int32_t len = it.z - it.y, first = epos[q];
    if (first < 0 || first + len > nnz_long) return;   // a plan / nnz_long mismatch writes nothing out of bounds
    for (int i = threadIdx.x % 32; i < len; i += 32) {
        if (ecol) ecol[first + i] = col[it.y + i];
        eval[first + i] = val[it.y + i];
    }
}

#ifndef MI_SPMM_BAND_MIN_PER
#define MI_SPMM_BAND_MIN_PER 1   // entries a band cut of a split row should hold on average (plan_seg_flags_kernel); 1 = every row cuts at `band`
#endif
int32_t plan_band_min_per() {
    const char* e = getenv("LAPLACE_SPMM_BAND_MIN_PER");   // A/B
    const int v = e ? atoi(e) : MI_SPMM_BAND_MIN_PER;
    return v < 1 ? 1 : v;
}

dim3 plan_grid(int64_t n) { return dim3((unsigned)mi_ceil_div(n > 0 ? n : 1, 256)); }

#ifndef MI_SPMM_HOT_WAVES
#define MI_SPMM_HOT_WAVES 20  // wavefronts per CU of the persistent launch (workgroups per CU = this / wavefronts per workgroup)
#endif
struct HotLaunch { int state = 0; int cus = 0; size_t lds_cap = 0; int occ_key = -1; int occ = 0; };

template <int LPR, int VPL, bool SPARSE, bool ADAM>
int launch_spmm_mode(int64_t n_rows, int d4, const int32_t* rowptr, const int32_t* col, const float* val,
                     const float4* X4, int64_t ldx4, const Epilogue& ep, const mi_spmm_plan* plan,
                     float4* partial, const Ex& ex, int64_t n_list, hipStream_t s, const mi_spmm_sweep* sweep) {
    constexpr int UNROLL = (VPL == 1) ? MI_SPMM_UNROLL : 4;                                        // work items, sweep
    constexpr int UNROLL_ROWS = (VPL == 1) ? (ADAM ? MI_SPMM_UNROLL_ADAM : MI_SPMM_UNROLL) : 4;   // the short-row kernel (the one that carries the Adam epilogue)
    constexpr int SG = (MI_WAVE / LPR) * kWavesPerBlock;
    constexpr int ROWS_RPS = MI_SPMM_ROWS_RPS;
#ifndef MI_SPMM_ITEMS_SLOTS
#define MI_SPMM_ITEMS_SLOTS 8  // A/B on C2 (items kernel, band 8192): 8: 397 us, 16: 415 us, 32: 461 us — the fewer work items an
#endif                         // XCD has in flight, the fewer bands its L2 has to hold at once
    constexpr int ITEMS_RPS = (SG >= MI_SPMM_ITEMS_SLOTS) ? 1 : MI_SPMM_ITEMS_SLOTS / SG;  // launch slots per workgroup
    const int32_t chunk = plan ? plan->chunk : INT32_MAX;
    const bool listed = SPARSE && ex.row_list != nullptr;
    const bool do_short = (ex.parts & MI_SPMM_SHORT_ROWS) != 0, do_split = (ex.parts & MI_SPMM_SPLIT_ROWS) != 0;
    if (do_split && plan && plan->n_items > 0 && sweep) {
        if constexpr (VPL == 1 && LPR <= 32) {
            constexpr int NB = MI_WAVE / LPR;
            const int threads = 32 / NB * MI_WAVE;  // 32 sub-groups per workgroup
            const size_t lds = (size_t)32 * 8 * LPR * sizeof(float4);
            auto kern = spmm_sweep_kernel<LPR, UNROLL, SPARSE>;
            static bool attr_set = false;  // per instantiation; idempotent
            if (!attr_set) {
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds) != hipSuccess)
                    return MI_ERR_UNSUPPORTED;
                attr_set = true;
            }
            hipLaunchKernelGGL(kern, dim3((unsigned)(8 * (sweep->n_streams / 32))), dim3(threads), lds, s, *sweep, d4, X4, ldx4,
                               partial, ex.x_map, ep.streaming);
        } else {
            return MI_ERR_UNSUPPORTED;
        }
    }
    bool xmap_form = false;   // the work items ran as spmm_items_xscan_kernel: their live flags are valid
    // HYBRID plan (round 4): the hub rows in SWEEP form AND the remaining split rows as banded work items — `items` is
    // then non-null beside the sweep, its slots are absolute (they start behind the sweep's 8 * n_slots partial rows)
    if (do_split && plan && plan->n_items > 0 && (!sweep || plan->items)) {  // row_list mode still reduces every split row: hubs are few and almost always wanted
        const int64_t n_launch = plan->n_launch;
        dim3 gi((unsigned)mi_ceil_div(n_launch, SG * ITEMS_RPS));
        const int32_t* ecol = plan->epos ? plan->ecol : col;   // packed plan: the work items read the plan's launch-ordered copies
        const float* eval = plan->epos ? plan->eval : val;
        bool done = false;
        xmap_form = false;
        if constexpr (SPARSE) {
            // a mapped operand declared rare-live (x_bits) on a packed plan: the scan form
            if (ex.x_map && ex.x_bits && ex.slot_live && plan->epos) {
                hipLaunchKernelGGL((spmm_items_xscan_kernel<LPR, VPL>),
                                   dim3((unsigned)mi_ceil_div(n_launch, (int64_t)(kWavesPerBlock * MI_WAVE))), dim3(kBlock), 0, s,
                                   n_launch, d4, reinterpret_cast<const int4*>(plan->items), plan->epos, plan->ecol, plan->eval,
                                   X4, ldx4, partial, ex.x_map, ex.x_bits, ex.slot_live, ep.streaming);
                done = xmap_form = true;
            }
        }
        if (!done)
            hipLaunchKernelGGL((spmm_items_kernel<LPR, VPL, UNROLL, ITEMS_RPS, SPARSE>), gi, dim3(kBlock), 0, s, n_launch,
                               plan->band > 0 ? 1 : 0, d4, reinterpret_cast<const int4*>(plan->items), plan->epos, ecol, eval,
                               X4, ldx4, partial, ex.x_map, ep.streaming);
    }
    const int64_t n_out = listed ? n_list : n_rows;
    bool short_done = false;
    if constexpr (!SPARSE && VPL == 1) {
        if (do_short && n_out > 0 && plan && ex.hot_rows != 0) {
            // persistent pipelined launch, optionally with the hot rows of X in LDS; falls through to the plain launch
            // when the device query fails
            auto kern = spmm_rows_hot_kernel<LPR, UNROLL, ADAM>;
            static HotLaunch hl;  // per instantiation: attribute + device properties queried once
            const int thr_arg = ex.hot_threads & 0xFFF, waves_arg = ex.hot_threads >> 12;
            const int threads = thr_arg == 512 ? 512 : (thr_arg == 1024 ? 1024 : 256);
            const size_t row_bytes = (size_t)d4 * sizeof(float4);
            if (hl.state == 0) {
                hl.state = -1;
                hipDeviceProp_t prop;
                int dev = 0;
                if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) {
                    hl.cus = prop.multiProcessorCount;
                    hl.lds_cap = (size_t)prop.maxSharedMemoryPerMultiProcessor;
                    const size_t optin = hl.lds_cap > 1024 ? hl.lds_cap - 1024 : 0;
                    if (hl.cus > 0 && optin >= row_bytes &&
                        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)optin) == hipSuccess)
                        hl.state = 1;
                }
            }
            if (hl.state == 1) {
                // workgroups per CU: the wavefront budget (MI_SPMM_HOT_WAVES per CU unless the caller says otherwise) cut
                // to what the kernel's registers allow; each workgroup gets an equal share of the LDS for its copy of the
                // hot rows — a small cache loses little (Zipf: 64 rows serve 46 % of the gathers, 304 rows 59 %)
                int wg_per_cu = std::max(1, (waves_arg > 0 ? waves_arg : MI_SPMM_HOT_WAVES) / (threads / MI_WAVE));
                const size_t share = (hl.lds_cap / (size_t)wg_per_cu) - 512;
                const int hot_n = (int)std::min<size_t>((size_t)std::max(ex.hot_rows, 0), share / row_bytes);
                const size_t lds = (size_t)hot_n * row_bytes;
                const int key = threads * 4 + (hot_n > 0 ? 1 : 0);  // occupancy depends on the registers, and on the LDS only via wg_per_cu
                if (hl.occ_key != key) {
                    int occ = 0;
                    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(kern), threads, lds) != hipSuccess)
                        occ = 0;
                    hl.occ_key = key;
                    hl.occ = occ;
                }
                wg_per_cu = std::min(wg_per_cu, hl.occ);
                if (wg_per_cu > 0) {
                    hipLaunchKernelGGL(kern, dim3((unsigned)(hl.cus * wg_per_cu)), dim3(threads), lds, s, n_rows, d4, rowptr, col,
                                       val, X4, ldx4, ep, chunk, ex, ex.hot_base, hot_n);
                    short_done = true;
                }
            }
        }
    }
    if (do_short && n_out > 0 && !short_done) {
        dim3 gr((unsigned)mi_ceil_div(n_out, SG * ROWS_RPS));
        if (MI_SPMM_XCD_RANGES && !listed) gr.x = (gr.x + 7u) / 8u * 8u;
        hipLaunchKernelGGL((spmm_rows_kernel<LPR, VPL, UNROLL_ROWS, ROWS_RPS, SPARSE, ADAM>), gr, dim3(kBlock), 0, s, n_out, d4,
                           rowptr, col, val, X4, ldx4, ep, chunk, ex, plan ? 0 : 1);
    }
    if (do_split && plan && plan->n_long_rows > 0) {
        const int64_t nf = listed ? n_list : (int64_t)plan->n_long_rows;
        Ex exf = ex;
        if (!xmap_form) exf.slot_live = nullptr;
        if (nf > 0)
            hipLaunchKernelGGL((spmm_fixup_kernel<LPR, VPL, SPARSE, ADAM>), dim3((unsigned)nf), dim3(kBlock), 0, s,
                               (int32_t)nf, d4, plan->long_rows, plan->item_ptr, partial, ep, exf);
    }
    return mi_launch_status();
}

template <int LPR, int VPL>
int launch_spmm(int64_t n_rows, int d4, const int32_t* rowptr, const int32_t* col, const float* val,
                const float4* X4, int64_t ldx4, const Epilogue& ep, const mi_spmm_plan* plan,
                float4* partial, const Ex& ex, int64_t n_list, hipStream_t s, const mi_spmm_sweep* sweep) {
    const bool sparse = ex.x_map || ex.row_list;
#define MI_SPMM_GO(SP, AD) \
    return launch_spmm_mode<LPR, VPL, SP, AD>(n_rows, d4, rowptr, col, val, X4, ldx4, ep, plan, partial, ex, n_list, s, sweep)
    if (ep.p) {
        if (sparse) MI_SPMM_GO(true, true);
        MI_SPMM_GO(false, true);
    }
    if (sparse) MI_SPMM_GO(true, false);
    MI_SPMM_GO(false, false);
#undef MI_SPMM_GO
}

}  // namespace

static inline bool n_cols_ok(const mi_spmm_sweep*) { return true; }  // columns are < 2^28 by construction of the plan (checked on the host side)

extern "C" {

size_t mi_spmm_plan_workspace_bytes(int64_t n_rows, int64_t nnz) {
    if (n_rows < 0 || nnz < 0) return 0;
    const size_t n1 = (size_t)n_rows + 1, z = (size_t)(nnz > 0 ? nnz : 1);
    // worst case: every row is split and every entry is its own work item
    return 4 * mi_align_up(n1 * 4, 256) + 256 + plan_tmp_bytes(nnz > n_rows + 1 ? nnz : n_rows + 1) +
           2 * mi_align_up(n1 * 4, 256) + 3 * mi_align_up(z * 4, 256) + 2 * mi_align_up(z * 8, 256);
}

int mi_spmm_plan_count(int64_t n_rows, int64_t n_cols, const int32_t* rowptr, const int32_t* col,
                       int32_t chunk, int32_t band, void* ws, size_t ws_bytes, mi_spmm_plan_info* info,
                       mi_stream_t stream) {
    return mi_spmm_plan_count_range(n_rows, n_cols, rowptr, col, chunk, INT32_MAX, band, ws, ws_bytes, info, stream);
}

int mi_spmm_plan_count_range(int64_t n_rows, int64_t n_cols, const int32_t* rowptr, const int32_t* col,
                             int32_t chunk, int32_t max_deg, int32_t band, void* ws, size_t ws_bytes,
                             mi_spmm_plan_info* info, mi_stream_t stream) {
    MI_CHECK_ARG(n_rows >= 0 && n_cols >= 0 && rowptr && chunk > 0 && max_deg >= chunk && band >= 0 && ws && info);
    if (n_rows >= INT32_MAX || n_cols >= INT32_MAX) return MI_ERR_TOO_LARGE;
    hipStream_t s = (hipStream_t)stream;
    memset(info, 0, sizeof(*info));
    info->chunk = chunk;
    info->band = band;
    info->n_bands = band > 0 ? (int32_t)mi_ceil_div(n_cols > 0 ? n_cols : 1, band) : 0;
    int32_t nnz32 = 0;
    MI_HIP(hipMemcpyAsync(&nnz32, rowptr + n_rows, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    MI_HIP(hipStreamSynchronize(s));
    const int64_t nnz = nnz32;
    MI_CHECK_ARG(band == 0 || col || nnz == 0);  // an empty adjacency has no columns to band
    PlanWs w{};
    {
        MiArena ar(ws, ws_bytes);
        if (!plan_carve(ar, n_rows, nnz, -1, -1, -1, w)) return MI_ERR_WORKSPACE;
    }
    const int64_t n1 = n_rows + 1;
    hipLaunchKernelGGL(plan_flags_kernel, plan_grid(n1), dim3(256), 0, s, n_rows, rowptr, chunk, max_deg, w.is_long, w.ldeg);
    size_t tb = w.tmp_bytes;
    MI_HIP(rocprim::exclusive_scan(w.tmp, tb, w.is_long, w.long_off, 0, (size_t)n1, rocprim::plus<int32_t>(), s));
    tb = w.tmp_bytes;
    MI_HIP(rocprim::exclusive_scan(w.tmp, tb, w.ldeg, w.lnz_off, 0, (size_t)n1, rocprim::plus<int32_t>(), s));
    int32_t totals[2] = {0, 0};
    MI_HIP(hipMemcpyAsync(&totals[0], w.long_off + n_rows, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    MI_HIP(hipMemcpyAsync(&totals[1], w.lnz_off + n_rows, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    MI_HIP(hipStreamSynchronize(s));
    const int32_t n_long = totals[0], nnz_long = totals[1];
    info->n_long_rows = n_long;
    info->nnz_long = nnz_long;
    if (n_long == 0) return mi_launch_status();
    {
        MiArena ar(ws, ws_bytes);
        if (!plan_carve(ar, n_rows, nnz, n_long, nnz_long, -1, w)) return MI_ERR_WORKSPACE;
    }
    hipLaunchKernelGGL(plan_long_rows_kernel, plan_grid(n1), dim3(256), 0, s, n_rows, w.is_long, w.long_off, w.lnz_off,
                       w.lrows, w.lprefix);
    hipLaunchKernelGGL(plan_seg_flags_kernel, plan_grid(nnz_long), dim3(256), 0, s, nnz_long, n_long, w.lrows, w.lprefix,
                       rowptr, col, chunk, band, n_cols, plan_band_min_per(), w.flags);
    tb = w.tmp_bytes;
    MI_HIP(rocprim::exclusive_scan(w.tmp, tb, w.flags, w.seg_of, 0, (size_t)nnz_long, rocprim::plus<int32_t>(), s));
    int32_t last[2] = {0, 0};
    MI_HIP(hipMemcpyAsync(&last[0], w.seg_of + nnz_long - 1, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    MI_HIP(hipMemcpyAsync(&last[1], w.flags + nnz_long - 1, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    MI_HIP(hipStreamSynchronize(s));
    const int32_t n_seg = last[0] + last[1];
    info->n_items = n_seg;
    {
        MiArena ar(ws, ws_bytes);
        if (!plan_carve(ar, n_rows, nnz, n_long, nnz_long, n_seg, w)) return MI_ERR_WORKSPACE;
    }
    hipLaunchKernelGGL(plan_seg_starts_kernel, plan_grid(nnz_long), dim3(256), 0, s, nnz_long, n_long, w.lrows, w.lprefix,
                       rowptr, col, band, w.flags, w.seg_of, w.seg_start, w.keys0);
    if (band == 0) {
        info->n_launch = n_seg;
        return mi_launch_status();
    }
    rocprim::double_buffer<uint64_t> keys(w.keys0, w.keys1);
    tb = w.tmp_bytes;
    MI_HIP(rocprim::radix_sort_keys(w.tmp, tb, keys, (size_t)n_seg, 0, 64, s));
    info->keys_in_second = keys.current() == w.keys1 ? 1 : 0;
    hipLaunchKernelGGL(plan_queue_bounds_kernel, dim3(1), dim3(64), 0, s, n_seg, keys.current(), w.qstart);
    MI_HIP(hipMemcpyAsync(info->queue_start, w.qstart, 9 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    MI_HIP(hipStreamSynchronize(s));
    int32_t longest = 0;
    for (int x = 0; x < 8; ++x) longest = std::max(longest, info->queue_start[x + 1] - info->queue_start[x]);
    info->queue_len = (int32_t)(mi_ceil_div(longest, kPlanGroup) * kPlanGroup);
    info->n_launch = 8 * (int64_t)info->queue_len;
    return mi_launch_status();
}

int mi_spmm_plan_fill(int64_t n_rows, const int32_t* rowptr, const mi_spmm_plan_info* info, mi_spmm_plan* plan,
                      void* ws, size_t ws_bytes, mi_stream_t stream) {
    MI_CHECK_ARG(n_rows >= 0 && rowptr && info && plan && ws);
    hipStream_t s = (hipStream_t)stream;
    plan->chunk = info->chunk;
    plan->n_long_rows = (int32_t)info->n_long_rows;
    plan->n_items = (int32_t)info->n_items;
    plan->n_launch = (int32_t)info->n_launch;
    plan->band = info->band;
    plan->n_bands = info->n_bands;
    if (info->n_launch >= INT32_MAX) return MI_ERR_TOO_LARGE;
    int32_t nnz32 = 0;
    MI_HIP(hipMemcpyAsync(&nnz32, rowptr + n_rows, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    MI_HIP(hipStreamSynchronize(s));
    PlanWs w{};
    MiArena ar(ws, ws_bytes);
    if (info->n_long_rows == 0) {
        if (!plan_carve(ar, n_rows, nnz32, -1, -1, -1, w)) return MI_ERR_WORKSPACE;
        if (plan->long_index) MI_HIP(hipMemsetAsync(plan->long_index, 0xFF, (size_t)n_rows * sizeof(int32_t), s));
        MI_HIP(hipStreamSynchronize(s));
        return mi_launch_status();
    }
    MI_CHECK_ARG(plan->long_rows && plan->item_ptr && plan->items);
    if (!plan_carve(ar, n_rows, nnz32, info->n_long_rows, info->nnz_long, info->n_items, w)) return MI_ERR_WORKSPACE;
    const int32_t n_long = (int32_t)info->n_long_rows, n_seg = (int32_t)info->n_items;
    hipLaunchKernelGGL(plan_fill_rows_kernel, plan_grid(n_rows), dim3(256), 0, s, n_rows, w.is_long, w.long_off,
                       plan->long_rows, plan->long_index);
    hipLaunchKernelGGL(plan_item_ptr_kernel, plan_grid(n_long + 1), dim3(256), 0, s, n_long, n_seg, w.lprefix, w.seg_of,
                       plan->item_ptr);
    int4* items = reinterpret_cast<int4*>(plan->items);
    if (info->band > 0)
        hipLaunchKernelGGL(plan_pad_kernel, plan_grid(info->n_launch), dim3(256), 0, s, info->n_launch, items);
    QueueStarts qs;
    for (int x = 0; x < 9; ++x) qs.q[x] = info->queue_start[x];
    hipLaunchKernelGGL(plan_items_kernel, plan_grid(n_seg), dim3(256), 0, s, n_seg, info->band > 0 ? 1 : 0,
                       info->keys_in_second ? w.keys1 : w.keys0, qs, w.seg_start, n_long, w.lrows, w.lprefix, rowptr,
                       items);
    MI_HIP(hipStreamSynchronize(s));  // ws may be released by the caller on return
    return mi_launch_status();
}

size_t mi_spmm_plan_pack_workspace_bytes(int64_t n_launch) {
    const size_t n = (size_t)(n_launch > 0 ? n_launch : 0) + 1;
    return 2 * mi_align_up(n * sizeof(int32_t), 256) + plan_tmp_bytes((int64_t)n);
}

int mi_spmm_plan_pack_entries(const mi_spmm_plan* plan, int64_t nnz_long, const int32_t* col, const float* val,
                              int32_t values_only, void* ws, size_t ws_bytes, mi_stream_t stream) {
    MI_CHECK_ARG(plan && plan->items && plan->epos && plan->ecol && plan->eval && val && (values_only || col));
    MI_CHECK_ARG(plan->n_launch > 0 && nnz_long >= 0 && nnz_long < INT32_MAX && mi_aligned16(plan->items));
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = plan->n_launch;
    if (!values_only) {
        MiArena ar(ws, ws_bytes);
        int32_t* len = ar.take<int32_t>((size_t)n + 1);
        const size_t tmp_bytes = plan_tmp_bytes(n + 1);
        char* tmp = ar.take<char>(tmp_bytes);
        if (!len || !tmp) return MI_ERR_WORKSPACE;
        hipLaunchKernelGGL(plan_item_len_kernel, plan_grid(n), dim3(256), 0, s, n, reinterpret_cast<const int4*>(plan->items), len);
        size_t tb = tmp_bytes;
        MI_HIP(rocprim::exclusive_scan(tmp, tb, len, plan->epos, 0, (size_t)n, rocprim::plus<int32_t>(), s));
    }
    constexpr int kPerBlock = 256 / 32;   // one 32-lane group per launch slot
    hipLaunchKernelGGL(plan_pack_kernel, dim3((unsigned)mi_ceil_div(n, (int64_t)kPerBlock)), dim3(256), 0, s, n,
                       reinterpret_cast<const int4*>(plan->items), plan->epos, (int32_t)nnz_long, col, val,
                       values_only ? nullptr : plan->ecol, plan->eval);
    return mi_launch_status();
}

size_t mi_spmm_workspace_bytes(const mi_spmm_plan* plan, int64_t d) {
    if (!plan || plan->n_items <= 0) return 0;
    // the partial rows, then one live flag per work item (mapped launches with mi_spmm_ex.x_bits)
    return mi_align_up((size_t)plan->n_items * (size_t)d * sizeof(float), 256) + mi_align_up((size_t)plan->n_items, 256);
}

int mi_map_live_bits_i32(int64_t n, const int32_t* map, uint32_t* bits, mi_stream_t stream) {
    MI_CHECK_ARG(n >= 0 && (n == 0 || (map && bits)));
    if (n == 0) return 0;
    hipLaunchKernelGGL(map_live_bits_kernel, dim3((unsigned)mi_ceil_div(n, (int64_t)kBlock)), dim3(kBlock), 0, (hipStream_t)stream, n, map,
                       bits);
    return mi_launch_status();
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
    const mi_adam_args* adam = exh ? exh->adam : nullptr;
    MI_CHECK_ARG(X && (Y || S || adam));
    MI_CHECK_ARG(ldx % 4 == 0 && ldx >= d && mi_aligned16(X));
    MI_CHECK_ARG(!Y || (ldy % 4 == 0 && ldy >= d && mi_aligned16(Y) && Y != X));
    MI_CHECK_ARG(!S || (lds % 4 == 0 && lds >= d && mi_aligned16(S) && S != X));
    MI_CHECK_ARG(!addend || (lda % 4 == 0 && lda >= d && mi_aligned16(addend)));
    Ex ex = {nullptr, nullptr, nullptr, nullptr, nullptr, MI_SPMM_SHORT_ROWS | MI_SPMM_SPLIT_ROWS, 0, 0, 0, nullptr, nullptr};
    int64_t n_list = 0;
    if (exh) {
        ex.x_map = exh->x_map;
        ex.addend_map = exh->addend_map;
        ex.row_list = exh->row_list;
        ex.n_list_dev = exh->n_list_dev;
        n_list = exh->n_list;
        if (exh->parts) {
            MI_CHECK_ARG((exh->parts & ~(MI_SPMM_SHORT_ROWS | MI_SPMM_SPLIT_ROWS)) == 0);
            ex.parts = exh->parts;
        }
        if (!ex.x_map && !ex.row_list && d <= 256) {  // hints: ignored where the persistent launch does not apply
            MI_CHECK_ARG(exh->hot_base >= 0 && exh->hot_threads >= 0);
            ex.hot_base = exh->hot_base;
            ex.hot_rows = exh->hot_rows;   // the caller's contract: hot_base + max(hot_rows, 0) <= rows of X
            ex.hot_threads = exh->hot_threads;
        }
        if (ex.row_list) {
            MI_CHECK_ARG(n_list >= 0 && !ex.addend_map);
            if (n_list == 0) return 0;
            if (plan && plan->n_long_rows > 0) {
                MI_CHECK_ARG(plan->long_index);
                ex.long_index = plan->long_index;
            }
        }
    }
    const mi_spmm_sweep* sweep = (exh && plan && plan->n_items > 0) ? exh->sweep : nullptr;
    if (sweep) {
        MI_CHECK_ARG(sweep->col && sweep->val && sweep->stream_ptr && sweep->slot_of);
        MI_CHECK_ARG(sweep->n_streams > 0 && sweep->n_streams % 32 == 0 && sweep->n_streams <= 32 * 32);
        // plain sweep plan: every partial row is the sweep's; hybrid: work items (plan->items) own the slots behind them
        MI_CHECK_ARG(sweep->n_slots > 0 && n_cols_ok(sweep) &&
                     (plan->items ? plan->n_items > 8 * sweep->n_slots : plan->n_items == 8 * sweep->n_slots));
        MI_CHECK_ARG(sweep->slack >= 0);
        if (d > 128) return MI_ERR_UNSUPPORTED;
    }
    float4* partial = nullptr;
    if (plan && plan->n_items > 0) {
        const int64_t own_items = (int64_t)plan->n_items - (sweep ? 8 * (int64_t)sweep->n_slots : 0);  // the work items' partial rows
        MI_CHECK_ARG(plan->item_ptr && plan->long_rows && (sweep || plan->items));
        MI_CHECK_ARG(!plan->items || plan->n_launch >= own_items);
        MI_CHECK_ARG(!plan->items || plan->band == 0 || plan->n_launch % (8 * kPlanGroup) == 0);
        MI_CHECK_ARG(!plan->epos || (plan->items && plan->ecol && plan->eval));
        if (!ws || ws_bytes < mi_spmm_workspace_bytes(plan, d)) return MI_ERR_WORKSPACE;
        MI_CHECK_ARG(mi_aligned16(ws) && (!plan->items || mi_aligned16(plan->items)));
        partial = reinterpret_cast<float4*>(ws);
        // the work items of a mapped operand whose live columns are rare (x_bits): spmm_items_xscan_kernel + live flags (packed plans).
        // Not with a sweep (its slots carry no flags) and not with row_list (the fix-up then walks listed rows only — fine —
        // but the combination has no caller and no test)
        if (exh && exh->x_bits && ex.x_map && plan->items && plan->epos && !sweep && !ex.row_list) {
            ex.x_bits = exh->x_bits;
            ex.slot_live = reinterpret_cast<uint8_t*>(ws) + mi_align_up((size_t)plan->n_items * (size_t)d * sizeof(float), 256);
        }
    }
    Epilogue ep;
    ep.Y = reinterpret_cast<float4*>(Y);                 ep.ldy4 = ldy / 4;
    ep.addend = reinterpret_cast<const float4*>(addend); ep.lda4 = lda / 4;
    ep.S = reinterpret_cast<float4*>(S);                 ep.lds4 = lds / 4;
    ep.scale = scale;
    ep.streaming = (size_t)n_rows * (size_t)d * sizeof(float) >= ((size_t)64 << 20);  // 2x the aggregate L2
    ep.p = nullptr; ep.ldp4 = 0; ep.m = nullptr; ep.v = nullptr; ep.reg_w = nullptr;
    ep.adam = MiAdamConsts{};
    if (adam) {
        MI_CHECK_ARG(!ex.row_list && adam->p && adam->m && adam->v && adam->step >= 1);
        MI_CHECK_ARG(adam->ldp % 4 == 0 && adam->ldp >= d && mi_aligned16(adam->p) && mi_aligned16(adam->m) &&
                     mi_aligned16(adam->v));
        MI_CHECK_ARG(adam->p != X && adam->p != Y && adam->p != S && adam->p != addend);
        ep.p = reinterpret_cast<float4*>(adam->p);
        ep.ldp4 = adam->ldp / 4;
        ep.m = reinterpret_cast<float4*>(adam->m);
        ep.v = reinterpret_cast<float4*>(adam->v);
        ep.reg_w = adam->reg_w;
        ep.adam = mi_adam_consts(adam->lr, adam->beta1, adam->beta2, adam->eps, adam->step);
    }
    const float4* X4 = reinterpret_cast<const float4*>(X);
    const int d4 = (int)(d / 4);
    hipStream_t s = (hipStream_t)stream;
    if (d4 <= 8)   return launch_spmm<8, 1>(n_rows, d4, rowptr, col, val, X4, ldx / 4, ep, plan, partial, ex, n_list, s, sweep);
    if (d4 <= 16)  return launch_spmm<16, 1>(n_rows, d4, rowptr, col, val, X4, ldx / 4, ep, plan, partial, ex, n_list, s, sweep);
    if (d4 <= 32)  return launch_spmm<32, 1>(n_rows, d4, rowptr, col, val, X4, ldx / 4, ep, plan, partial, ex, n_list, s, sweep);
    if (d4 <= 64)  return launch_spmm<64, 1>(n_rows, d4, rowptr, col, val, X4, ldx / 4, ep, plan, partial, ex, n_list, s, sweep);
    return launch_spmm<64, 2>(n_rows, d4, rowptr, col, val, X4, ldx / 4, ep, plan, partial, ex, n_list, s, sweep);
}

int mi_spmm_csr_f32(int64_t n_rows, int64_t d, const int32_t* rowptr, const int32_t* col,
                    const float* val, const float* X, int64_t ldx, float* Y, int64_t ldy,
                    const float* addend, int64_t lda, float* S, int64_t lds, float scale,
                    const mi_spmm_plan* plan, void* ws, size_t ws_bytes, mi_stream_t stream) {
    return mi_spmm_csr_ex_f32(n_rows, d, rowptr, col, val, X, ldx, Y, ldy, addend, lda, S, lds, scale, plan,
                              nullptr, ws, ws_bytes, stream);
}

}  // extern "C"

// ---- twin launcher (pairs.hpp) ----------------------------------------------------------------------------------------------
namespace mi_pairs {

static int lpr_class(int d4) { return d4 <= 8 ? 8 : d4 <= 16 ? 16 : d4 <= 32 ? 32 : d4 <= 64 ? 64 : 0; }

static bool rows_side(const SpmmSide& q, RowsSide& r) {
    if (q.n_rows <= 0 || q.d <= 0 || q.d % 4 != 0 || q.n_rows >= INT32_MAX || !q.rowptr || !q.X || (!q.Y && !q.S)) return false;
    if (!mi_aligned16(q.X) || (q.Y && (!mi_aligned16(q.Y) || q.Y == q.X)) || (q.S && (!mi_aligned16(q.S) || q.S == q.X)) ||
        (q.addend && !mi_aligned16(q.addend)))
        return false;
    r.n_out = q.n_rows; r.d4 = (int)(q.d / 4);
    r.rowptr = q.rowptr; r.col = q.col ? q.col : q.rowptr; r.val = q.val ? q.val : reinterpret_cast<const float*>(q.rowptr);
    r.X4 = reinterpret_cast<const float4*>(q.X); r.ldx4 = q.d / 4;
    Epilogue& ep = r.ep;
    ep.Y = reinterpret_cast<float4*>(q.Y);                 ep.ldy4 = q.d / 4;
    ep.addend = reinterpret_cast<const float4*>(q.addend); ep.lda4 = q.d / 4;
    ep.S = reinterpret_cast<float4*>(q.S);                 ep.lds4 = q.d / 4;
    ep.scale = 1.0f;
    ep.streaming = (size_t)q.n_rows * (size_t)q.d * sizeof(float) >= ((size_t)64 << 20);   // mi_spmm_csr_ex_f32's own rule
    ep.p = nullptr; ep.ldp4 = 0; ep.m = nullptr; ep.v = nullptr; ep.reg_w = nullptr;
    ep.adam = MiAdamConsts{};
    return true;
}

template <int LPR>
static int launch_rows_pair(const RowsSide& a, const RowsSide& b, hipStream_t s) {
    constexpr int SG = (MI_WAVE / LPR) * kWavesPerBlock;
    const unsigned ga = (unsigned)mi_ceil_div(a.n_out, SG * MI_SPMM_ROWS_RPS), gb = (unsigned)mi_ceil_div(b.n_out, SG * MI_SPMM_ROWS_RPS);
    hipLaunchKernelGGL((spmm_rows_pair_kernel<LPR, MI_SPMM_UNROLL_PAIR>), dim3(ga + gb), dim3(kBlock), 0, s, a, b, ga);
    return mi_launch_status();
}

int spmm_planless_pair(const SpmmSide& a, const SpmmSide& b, hipStream_t s) {
    RowsSide ra, rb;
    if (MI_SPMM_XCD_RANGES || !rows_side(a, ra) || !rows_side(b, rb)) return MI_ERR_UNSUPPORTED;
    const int ca = lpr_class(ra.d4), cb = lpr_class(rb.d4);
    if (ca == 0 || ca != cb) return MI_ERR_UNSUPPORTED;   // different instantiations (or the two-register rows of d > 256)
    switch (ca) {
        case 8: return launch_rows_pair<8>(ra, rb, s);
        case 16: return launch_rows_pair<16>(ra, rb, s);
        case 32: return launch_rows_pair<32>(ra, rb, s);
        default: return launch_rows_pair<64>(ra, rb, s);
    }
}

}  // namespace mi_pairs

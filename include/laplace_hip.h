/*
 * laplace_hip.h — C ABI of liblaplace_hip.so, the MI355X (gfx950) hot path of the
 * laplace GNN link-prediction engine.
 *
 * The reference (dream-faster/laplace-gnn-recommendation) has no FFI of its own: its
 * hot path reaches native code only through third-party wheels (torch_sparse,
 * torch_scatter, PyG).  Each entry point below replaces one of those call sites; the
 * "replaces:" line cites the reference file:line (paths relative to the reference root)
 * whose work it does.  INTEGRATION.md shows the ctypes binding a reference maintainer
 * would add at each site.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller unless marked "host";
 *  - every function enqueues on `stream` (a hipStream_t passed as void*) and returns
 *    without synchronising, except the *_plan_build / *_count functions, which say so;
 *  - return value: 0 = success; >0 = hipError_t from the runtime; <0 = MI_ERR_* below;
 *  - no global mutable state: the library is re-entrant and thread-safe;
 *  - node / edge indices inside the library are int32 (n, nnz < 2^31 is checked);
 *    edge lists coming from the reference's tensors are int64 and converted once
 *    in mi_coo_to_csr_i32;
 *  - floating point is fp32 end to end (the reference's dtype); no reduced precision.
 */
#ifndef LAPLACE_HIP_H
#define LAPLACE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_ABI_VERSION 9

#define MI_ERR_BAD_ARG      (-1)  /* null pointer, negative size, misaligned buffer   */
#define MI_ERR_TOO_LARGE    (-2)  /* a size does not fit int32 indexing                */
#define MI_ERR_WORKSPACE    (-3)  /* workspace smaller than *_workspace_bytes says     */
#define MI_ERR_UNSUPPORTED  (-4)  /* feature width / k outside the compiled kernels    */

typedef void* mi_stream_t; /* hipStream_t */

int         mi_abi_version(void);
const char* mi_error_string(int code); /* host string, static storage */

/* ------------------------------------------------------------------------------------
 * K4  COO -> sorted CSR.
 * replaces: torch_sparse.SparseTensor(row=, col=, sparse_sizes=) at
 *           data/lightgcn_loader.py:65-79 (sort by row*N+col, no duplicate merging,
 *           rowptr = ind2ptr(row)).
 * row/col: int64[nnz] as the reference's edge_index rows.  Outputs: rowptr int32[n_rows+1],
 * col_out int32[nnz] (sorted by (row, col)), perm int32[nnz] (perm[p] = index of the
 * input edge now at position p; nullable).
 * ---------------------------------------------------------------------------------- */
size_t mi_coo_to_csr_workspace_bytes(int64_t n_rows, int64_t nnz);
int    mi_coo_to_csr_i32(int64_t n_rows, int64_t n_cols, int64_t nnz,
                         const int64_t* row, const int64_t* col,
                         int32_t* rowptr, int32_t* col_out, int32_t* perm,
                         void* ws, size_t ws_bytes, mi_stream_t stream);

/* CSR(A) -> CSR(A^T).  replaces: SparseTensor.csr2csc()/colptr used by the backward of
 * torch_sparse.matmul (autograd through model/lightgcn.py:64).  perm_t[q] = position in
 * A of the entry now at position q of A^T (so val_t = val[perm_t]). */
size_t mi_csr_transpose_workspace_bytes(int64_t n_rows, int64_t nnz);
int    mi_csr_transpose_i32(int64_t n_rows, int64_t n_cols, int64_t nnz,
                            const int32_t* rowptr, const int32_t* col,
                            int32_t* rowptr_t, int32_t* col_t, int32_t* perm_t,
                            void* ws, size_t ws_bytes, mi_stream_t stream);

/* out[i] = src[idx[i]] — used to permute edge values. */
int mi_gather_f32(int64_t n, const float* src, const int32_t* idx, float* out,
                  mi_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K3  gcn_norm on a square CSR.
 * replaces: torch_geometric gcn_norm(SparseTensor, add_self_loops=False) at
 *           model/lightgcn.py:56:  deg = rowsum(A); dis = deg^-1/2 (inf -> 0);
 *           val[p] = (val[p] * dis[row]) * dis[col[p]].
 * val_in nullable (= all ones, SparseTensor without value).  dis_out float[n] is written
 * (deg^-1/2) and may be reused by the caller.
 * ---------------------------------------------------------------------------------- */
int mi_gcn_norm_csr_f32(int64_t n, int64_t nnz, const int32_t* rowptr, const int32_t* col,
                        const float* val_in, float* val_out, float* dis_out,
                        mi_stream_t stream);

/* Edge re-weighting: val_out[p] = (val_in[p] * row_scale[row(p)]) * col_scale[col[p]], any null
 * input meaning all ones.  gcn_norm is this with row_scale = col_scale = deg^-1/2; the
 * user-sharded multi-GPU path calls it directly because an item's degree is the sum over all
 * shards (one RCCL all-reduce of the degree vector at set-up), and SAGEConv's mean aggregation
 * (model/layers.py:11-24, aggr="mean") is row_scale = 1/in-degree. */
int mi_scale_csr_f32(int64_t n_rows, int64_t nnz, const int32_t* rowptr, const int32_t* col,
                     const float* val_in, const float* row_scale, const float* col_scale,
                     float* val_out, mi_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K1/K2  CSR SpMM with fused epilogue — the LightGCN propagate.
 * replaces: torch_sparse.matmul(adj_t, x) at model/lightgcn.py:87 (called K times per
 *           forward at :63-65, and K times on A^T in backward), plus the
 *           stack/mean at model/lightgcn.py:67-68 through the epilogue.
 *
 *   acc[r,:] = sum_{p in [rowptr[r], rowptr[r+1])} val[p] * X[col[p], :]
 *   if (Y) Y[r,:] = acc
 *   if (S) S[r,:] = scale * ((addend ? addend[r,:] : 0) + acc)
 *
 * d (feature width) must be a multiple of 4 and <= 512; X, Y, addend, S rows are
 * 16-byte aligned with leading dimensions ld* (in floats, multiples of 4).  S may alias
 * addend (in-place running sum).  Y/S must not alias X.
 *
 * Rows longer than the plan's chunk are split into work items whose partial sums are
 * reduced in a fixed order (no float atomics): results are bitwise reproducible run to run.
 * A BANDED plan (band > 0; needs `col`, columns ascending within each row) also cuts the
 * split rows at multiples of `band` columns and launches the work items band by band, the
 * workgroups of one band on one XCD, so that the gathers the hub rows make into the same
 * band of X meet in that XCD's L2 (choose band * d * 4 bytes ~ the 4 MB L2).  The plan
 * depends only on the adjacency structure; build it once.
 * ---------------------------------------------------------------------------------- */
#define MI_SPMM_GROUP 32   /* launch slots per XCD-interleave block of a banded plan */

typedef struct mi_spmm_plan {
    int32_t  chunk;        /* max nnz per work item of a split row                   */
    int32_t  n_long_rows;  /* rows with more than `chunk` entries                    */
    int32_t  n_items;      /* work items over all long rows = partial-sum rows       */
    int32_t  n_launch;     /* launch slots >= n_items (banded: a multiple of 8 * MI_SPMM_GROUP) */
    int32_t* long_rows;    /* device int32[n_long_rows]                              */
    int32_t* item_ptr;     /* device int32[n_long_rows+1]: slots of long row i       */
    int32_t* items;        /* device int32[4*n_launch], 16-byte aligned, in LAUNCH order:
                              row, begin, end, slot; slot < 0 = padding              */
    int32_t* long_index;   /* device int32[n_rows]: index into long_rows, -1 for
                              short rows; nullable (needed only by row_list launches)  */
    int32_t  band;         /* columns per band; 0 = row-major launch order           */
    int32_t  n_bands;
    /* Optional (round 4; all three or none): the split rows' entries copied into LAUNCH order by
     * mi_spmm_plan_pack_entries.  A work item's ~12 entries are then read from the same cache lines as its launch
     * neighbours' instead of two lines of their own in the CSR arrays (7.7 M work items on BASELINE configs[3]: ~2 GB of
     * line fetches per launch for 0.7 GB of entries).  epos[q] = first packed entry of launch slot q.  ecol / eval hold
     * VALUES: when the CSR's val changes, call mi_spmm_plan_pack_entries again with values_only = 1.                    */
    int32_t* epos;         /* device int32[n_launch]; nullable                       */
    int32_t* ecol;         /* device int32[nnz_long]                                 */
    float*   eval;         /* device float[nnz_long]                                 */
} mi_spmm_plan;

/* What mi_spmm_plan_count found; the caller allocates the plan arrays from it. */
typedef struct mi_spmm_plan_info {
    int64_t n_long_rows, n_items, n_launch, nnz_long;
    int32_t chunk, band, n_bands, queue_len;
    int32_t queue_start[9];
    int32_t keys_in_second;
} mi_spmm_plan_info;

/* Two phases, both SYNCHRONISE `stream` (counter read-backs); set-up time only.
 *   count: analyses the adjacency into `ws` and reports the array sizes in `info`;
 *   fill:  writes plan->long_rows[n_long_rows], item_ptr[n_long_rows+1], items[4*n_launch],
 *          long_index[n_rows] (nullable) — caller-allocated — and the scalar fields.
 * `ws` (mi_spmm_plan_workspace_bytes; worst case ~28 B per entry) must be the same buffer,
 * untouched in between.  col may be null when band == 0. */
size_t mi_spmm_plan_workspace_bytes(int64_t n_rows, int64_t nnz);
int    mi_spmm_plan_count(int64_t n_rows, int64_t n_cols, const int32_t* rowptr, const int32_t* col,
                          int32_t chunk, int32_t band, void* ws, size_t ws_bytes,
                          mi_spmm_plan_info* info, mi_stream_t stream);
/* The same analysis restricted to the rows with chunk < degree <= max_deg (round 4): the banded half of a HYBRID plan whose
 * rows above max_deg are laid out in SWEEP form (mi_spmm_sweep) by the host.  The host merges the two: long_rows / item_ptr
 * concatenated, the work items' slots shifted behind the sweep's 8 * n_slots partial rows, n_items = the sum; passing both
 * `items` and a sweep to mi_spmm_csr_ex_f32 runs both split-row kernels and ONE fix-up. */
int    mi_spmm_plan_count_range(int64_t n_rows, int64_t n_cols, const int32_t* rowptr, const int32_t* col,
                                int32_t chunk, int32_t max_deg, int32_t band, void* ws, size_t ws_bytes,
                                mi_spmm_plan_info* info, mi_stream_t stream);
int    mi_spmm_plan_fill(int64_t n_rows, const int32_t* rowptr, const mi_spmm_plan_info* info,
                         mi_spmm_plan* plan, void* ws, size_t ws_bytes, mi_stream_t stream);
/* Fills plan->epos / ecol / eval (caller-allocated: n_launch, nnz_long, nnz_long elements; nnz_long from
 * mi_spmm_plan_info) from the filled plan and the CSR's col / val.  values_only != 0: epos and ecol are kept, eval is
 * copied again (the adjacency was re-weighted).  ws: mi_spmm_plan_pack_workspace_bytes(plan->n_launch).  Enqueues only. */
size_t mi_spmm_plan_pack_workspace_bytes(int64_t n_launch);
int    mi_spmm_plan_pack_entries(const mi_spmm_plan* plan, int64_t nnz_long, const int32_t* col, const float* val,
                                 int32_t values_only, void* ws, size_t ws_bytes, mi_stream_t stream);
/* Bytes of partial-sum workspace mi_spmm_csr_f32 needs for this plan and width. */
size_t mi_spmm_workspace_bytes(const mi_spmm_plan* plan, int64_t d);

int mi_spmm_csr_f32(int64_t n_rows, int64_t d,
                    const int32_t* rowptr, const int32_t* col, const float* val,
                    const float* X, int64_t ldx,
                    float* Y, int64_t ldy,
                    const float* addend, int64_t lda,
                    float* S, int64_t lds, float scale,
                    const mi_spmm_plan* plan, void* ws, size_t ws_bytes,
                    mi_stream_t stream);

/* Sparse-operand form of the same product — same arithmetic, fewer bytes.  Used by the fused
 * train step, where the operands around the K propagates are known to be mostly zero rows
 * (the BPR gradient touches <= 3*batch rows) or only a few output rows are consumed (the last
 * forward layer feeds only the batch rows of the loss).  Every field is nullable:
 *   x_map      int32[n_cols]: X is COMPACT; column c reads X[x_map[c]], and x_map[c] < 0
 *              declares row c of the logical X to be all zeros (its entries are skipped —
 *              adding exact zeros changes no sum).
 *   addend_map int32[n_rows]: addend is COMPACT; row r adds addend[addend_map[r]], < 0 = none.
 *   row_list   int32[n_list]: only these rows are computed; Y, S and addend are COMPACT,
 *              indexed by position in the list (addend_map must be null).  n_list is the launch
 *              bound; n_list_dev (device int32[1], nullable) the actual count, so that a
 *              device-built list needs no host read-back.  Needs plan->long_index when the plan
 *              has split rows.
 * Results equal the dense call on the corresponding dense operands bit for bit (up to the
 * sign of zero). */
/* Optimizer epilogue: the product's S value g[r,:] = scale * (addend[r,:] + acc) is the gradient of
 * parameter row r and is consumed in registers by the Adam update of mi_adam_dense_f32 (same arithmetic,
 * bit for bit) instead of being written and read back: p / m / v rows are updated in place, reg_w as
 * there.  S may be null (the gradient is then never stored).  p must not alias X, Y, S or addend. */
typedef struct mi_adam_args {
    float*       p;  int64_t ldp;   /* [n_rows, d] parameters, ld in floats */
    float*       m;                 /* [n_rows, d] dense */
    float*       v;                 /* [n_rows, d] dense */
    const float* reg_w;             /* [n_rows] or null */
    double       lr, beta1, beta2, eps;
    int64_t      step;              /* counts from 1 */
} mi_adam_args;

/* SWEEP form of the split-row half (round 2).  The banded work items above pay one partial row per (row, band)
 * pair, which forbids bands small enough to sit comfortably in an XCD's 4 MB L2, and their short (col, val) segments
 * are re-fetched by every XCD that holds a neighbouring band.  Here the entries of the long rows are re-laid out ONCE
 * into 8 * n_streams STREAMS — stream (x, k) = everything sub-group k of XCD x will ever process: the entries whose
 * column band is congruent to x mod 8, of the <= 8 row parts ("slots") that sub-group owns, sorted by band, then slot —
 * so each sub-group reads its (col, val) pairs as one contiguous run, walks the bands in step with its XCD's other
 * sub-groups (streams are balanced per band), keeps its <= 8 accumulators in LDS across ALL bands, and writes one
 * partial row per (slot, XCD) at the end: 8 * n_slots partial rows whatever the band size.  The plan's long_rows /
 * item_ptr / long_index / chunk keep their meaning (item_ptr[i] = 8 * first slot of long row i; n_items = 8 * n_slots)
 * and the same fix-up kernel reduces them in (slot, XCD) order: no float atomics, bitwise reproducible.
 * Feature widths up to 128 (one float4 per lane); wider products use the work-item form. */
typedef struct mi_spmm_sweep {
    const int32_t* col;         /* int32[nnz_long], stream order: bits 28..30 = accumulator of the owning sub-group (0..7),
                                   bit 27 = first entry of a new band in this stream, bits 0..26 = column                */
    const float*   val;         /* float[nnz_long], stream order                                                        */
    const int32_t* stream_ptr;  /* int32[8 * n_streams + 1]: stream (x, k) = entries [ptr[x*n_streams+k], ptr[..+1])    */
    const int32_t* slot_of;     /* int32[n_streams * 8]: slot held in accumulator q of sub-group k; < 0 = unused        */
    int32_t        n_streams;   /* sub-groups per XCD: a multiple of 32, at most 32 * 32                                */
    int32_t        n_slots;
    int32_t*       progress;    /* reserved (null)                                                                          */
    int32_t        epoch;       /* bands per XCD of the plan (read by the MI_SWEEP_WG_SYNC experiment build only; else ignored)      */
    int32_t        slack;       /* pacing: ticks of the 100 MHz device clock per band; wavefronts do not start band t before
                                   t * slack ticks after their start (a late wavefront never waits).  0 = no pacing.  For the
                                   L2 hit rate only: results do not depend on it                                          */
} mi_spmm_sweep;

typedef struct mi_spmm_ex {
    const int32_t* x_map;
    const int32_t* addend_map;
    const int32_t* row_list;
    const int32_t* n_list_dev;
    int64_t        n_list;
    const mi_adam_args* adam;       /* nullable; not with row_list */
    int32_t        parts;           /* 0 = the whole product; else a mask: MI_SPMM_SHORT_ROWS computes the rows of at most
                                       plan->chunk entries, MI_SPMM_SPLIT_ROWS the split rows (work items + fix-up).  The two
                                       halves touch disjoint output rows, so a caller may enqueue them on two streams */
    int32_t        hot_rows;        /* 0 (the default, also of a null mi_spmm_ex and of mi_spmm_csr_f32): the plain
                                       one-workgroup-per-8-rows short-row launch — the measured best (profiles/
                                       r03_spmm_rows_persistent.md, r04_c4_item_rows_experiments.md).  != 0 (opt-in, dense launches
                                       with a plan, d <= 256): the short rows run as a PERSISTENT, software-pipelined launch — the
                                       grid is what the chip holds at once, wavefronts stride over the row pairs and fetch the next
                                       rows' pointers and (col, val) entries under the current gathers; hot_rows > 0 adds an LDS
                                       cache: rows [hot_base, hot_base + hot_rows) of X are staged once per workgroup and gathers
                                       of them are served from LDS instead of L2 (a recommendation graph under the locality order
                                       keeps its most popular items in the first item rows: 64 rows take ~46 %, 300 rows ~59 % of
                                       all user-row gathers), clipped to the workgroup's LDS share; hot_rows < 0 = persistent
                                       without a cache.  Hints for speed only: same values summed in the same order, bitwise the
                                       same result in all three forms.  (Round 3 had 0 = persistent and < 0 = plain: a zero-
                                       initialised struct silently chose the slower form.) */
    const mi_spmm_sweep* sweep;     /* nullable: the split rows run in SWEEP form (plan->items is then unused) */
    int32_t        hot_base;        /* first cached row of X (see hot_rows) */
    int32_t        hot_threads;     /* tuning of the hot-row launch, 0 = defaults: bits 0..11 workgroup size (256 / 512 / 1024),
                                       bits 12.. wavefronts per CU (workgroups per CU = that / wavefronts per workgroup) */
    const uint32_t* x_bits;         /* nullable, with x_map: bit c of the array (word c / 32, bit c % 32; mi_map_live_bits_i32 writes
                                       it) = (x_map[c] >= 0).  Saying so declares the live columns RARE (the batch's users among all
                                       users: first backward product of the fused step).  On a PACKED plan (epos / ecol / eval) the
                                       split rows' work items are then not walked one by one: wavefronts scan the packed entries,
                                       test the bit before the map, gather only the live entries in list order, write no partial
                                       row for a work item without any, and the fix-up skips those — same sums in the same order,
                                       bitwise the result without the hint.  Ignored with a sweep, a row_list or an unpacked plan. */
} mi_spmm_ex;
#define MI_SPMM_SHORT_ROWS 1
#define MI_SPMM_SPLIT_ROWS 2

int mi_spmm_csr_ex_f32(int64_t n_rows, int64_t d,
                       const int32_t* rowptr, const int32_t* col, const float* val,
                       const float* X, int64_t ldx,
                       float* Y, int64_t ldy,
                       const float* addend, int64_t lda,
                       float* S, int64_t lds, float scale,
                       const mi_spmm_plan* plan, const mi_spmm_ex* ex,
                       void* ws, size_t ws_bytes, mi_stream_t stream);

/* bits[w] bit b = (map[32 w + b] >= 0), w < ceil(n / 32): mi_spmm_ex.x_bits of a map. */
int mi_map_live_bits_i32(int64_t n, const int32_t* map, uint32_t* bits, mi_stream_t stream);

/* dst[i,:] = scale * ((accumulate ? dst[i,:] : 0) + src[rows[i] - row_offset,:])
 * for i in [begin_dev ? *begin_dev : 0, min(n_max, n_dev ? *n_dev : n_max)).
 * The running layer sum of the batch rows in the fused step (model/lightgcn.py:67-68 restricted
 * to the rows the loss reads); begin / row_offset address the item part of a batch node list
 * against an item-only table in the sharded step. */
int mi_gather_rows_f32(int64_t n_max, const int32_t* n_dev, const int32_t* begin_dev, int64_t d,
                       const int32_t* rows, int64_t row_offset, const float* src, int64_t ld_src,
                       float* dst, int64_t ld_dst, int32_t accumulate, float scale,
                       mi_stream_t stream);

/* dst[rows[i] - row_offset,:] = src[i,:] for i in [begin, min(n_max, *n_dev)); rows distinct.
 * Expands the item part of the compact batch gradient into the dense item table that the sharded
 * step all-reduces. */
int mi_scatter_rows_f32(int64_t n_max, const int32_t* n_dev, const int32_t* begin_dev, int64_t d,
                        const int32_t* rows, int64_t row_offset, const float* src, int64_t ld_src,
                        float* dst, int64_t ld_dst, mi_stream_t stream);

/* Unique node set of a BPR batch: gmap int32[n_nodes] = compact slot of node r (slots ordered by
 * node id) or -1; nodes int32[>= 3*batch] = slot -> node; count = device int32[2]:
 * count[0] = unique nodes, count[1] = unique USER nodes (their slots come first).
 * Users are nodes [0, n_users), item i is node n_users + i. */
size_t mi_batch_nodes_workspace_bytes(int64_t n_nodes);
int    mi_batch_nodes_i32(int64_t batch, int64_t n_users, int64_t n_nodes,
                          const int64_t* users, const int64_t* pos, const int64_t* neg,
                          int32_t* gmap, int32_t* nodes, int32_t* count,
                          void* ws, size_t ws_bytes, mi_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K9  mini-batch sampler: B positive edges with replacement + one structured negative
 *     each, entirely on device.
 * replaces: sample_mini_batch at data/lightgcn_loader.py:95-112
 *           (structured_negative_sampling over all E edges on the CPU, then
 *           random.choices(range(E), k=B)).
 * Edge e = Philox(seed, step, slot b) mod nnz of the *train* CSR (rows = users, cols =
 * item ids in [0, n_items)); the negative for edge e is drawn from [0, neg_range) by
 * Philox keyed on (seed, step, e) — the same edge drawn twice in one step gets the same
 * negative, as in the reference — and redrawn while it is a neighbour of the user
 * (binary search in the user's sorted CSR row).  `quirk_user_rows` is a bit mask:
 * bit 0 (MI_SAMPLE_KEY_COLLISION): also rejects negative 0 for user u if user u-1 has an edge to
 *   item `neg_range` (the reference's row*num_nodes+col key collision);
 * bit 1 (MI_SAMPLE_NO_SELF_LOOPS): also rejects the negative whose id equals the user's id —
 *   structured_negative_sampling(contains_neg_self_loops=False) as evaluation() calls it
 *   (run_pipeline_lightgcn.py:40-44) puts the keys i*num_nodes+i of all i < num_nodes into the
 *   rejection set; sample_mini_batch (data/lightgcn_loader.py:105) leaves the default, True.
 * row_of_edge int32[nnz] is the expanded row index (mi_csr_expand_rows).
 * edges_in_order != 0: slot b takes edge b of the CSR instead of a random one (batch <= nnz) —
 * one negative per edge of a split, as evaluation() does (run_pipeline_lightgcn.py:40-44).
 * Outputs int64[B] each (the reference's index dtype).
 * ---------------------------------------------------------------------------------- */
#define MI_SAMPLE_KEY_COLLISION 1
#define MI_SAMPLE_NO_SELF_LOOPS 2
int mi_csr_expand_rows(int64_t n_rows, const int32_t* rowptr, int32_t* row_of_edge,
                       int64_t nnz, mi_stream_t stream);
int mi_sample_bpr_batch(int64_t batch, int64_t nnz,
                        const int32_t* rowptr, const int32_t* col,
                        const int32_t* row_of_edge,
                        int64_t neg_range, int32_t quirk_user_rows, int32_t edges_in_order,
                        uint64_t seed, uint64_t step,
                        int64_t* users, int64_t* pos, int64_t* neg,
                        mi_stream_t stream);

/* ------------------------------------------------------------------------------------
 * a7+a8  fused batch gather + BPR loss forward/backward.
 * replaces: the six index_selects at run_pipeline_lightgcn.py:133-144 and bpr_loss at
 *           utils/metrics_lightgcn.py:9-45, and their autograd backward.
 *   x_b  = <u_b,p_b> - <u_b,n_b>          (rows of final_emb)
 *   loss = -mean_b softplus(x_b) + lambda * sum_b (|u0_b|^2+|p0_b|^2+|n0_b|^2)   (rows of e0)
 * final_emb / e0: [n_users + n_items, d] tables (item i is row n_users + i).
 * loss_out: device float[1], written (fixed-order reduction).
 * When g_final is non-null (dense [n, d], ALL ZERO on entry) the rows the batch touches are written with
 *   g_final[row,:] = g_scale * dL/dfinal[row,:]          (untouched rows stay zero)
 * and reg_w (float[n], zeroed by the caller; nullable) receives reg_w[row] = (occurrences of the
 * node in the batch) * 2*lambda*reg_scale, so that the L2 term's gradient is
 * reg_w[row] * e0[row,:] (applied by mi_adam_dense_f32 or by the caller).  The count is the length
 * of the node's run in the sorted reference list: one writer per entry, no atomics.
 * node_map (nullable, from mi_batch_nodes_i32): final_emb and g_final are then COMPACT
 * [count, d] tables addressed by node_map[node]; e0 and reg_w stay addressed by node id.
 * No float atomics on the gradient: the 3*batch row references are radix-sorted by row, summed in
 * 64-reference chunks in reference order and combined in chunk order, one writer per row — the
 * result is bitwise reproducible.  d <= 512.
 * ---------------------------------------------------------------------------------- */
size_t mi_bpr_workspace_bytes(int64_t batch);
int    mi_bpr_fwd_bwd_f32(int64_t batch, int64_t d, int64_t n_users,
                          const int64_t* users, const int64_t* pos, const int64_t* neg,
                          const float* final_emb, int64_t ldf,
                          const float* e0, int64_t lde,
                          float lambda, float g_scale, float reg_scale,
                          float* loss_out,
                          float* g_final, int64_t ldg,
                          float* reg_w,
                          const int32_t* node_map,
                          void* ws, size_t ws_bytes, mi_stream_t stream);

/* ------------------------------------------------------------------------------------
 * a9  dense Adam step (torch.optim.Adam semantics, no weight decay, no amsgrad).
 * replaces: optimizer.step() at run_pipeline_lightgcn.py:159.
 *   g   = grad[i] + (reg_w ? reg_w[row(i)] * p[i] : 0)
 *   m   = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g
 *   p  -= (lr / (1 - b1^step)) * m / (sqrt(v)/sqrt(1 - b2^step) + eps)
 * Hyper-parameters arrive as doubles and the derived scalars are formed in double on the
 * host before one rounding to fp32, as torch does; `step` counts from 1.
 * n_rows x d row-major, ld in floats; m, v dense [n_rows, d].
 * ---------------------------------------------------------------------------------- */
int mi_adam_dense_f32(int64_t n_rows, int64_t d,
                      float* p, int64_t ldp, const float* grad, int64_t ldgr,
                      float* m, float* v, const float* reg_w,
                      double lr, double beta1, double beta2, double eps, int64_t step,
                      mi_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K6  dense fp32 GEMM on the f32 MFMA (exact fp32, k-ordered fma chain):
 *   C[m,n] = act( sum_k A[m,k] * B(k,n) + bias[n] + (accumulate ? C[m,n] : 0) )
 * replaces: torch.nn.Linear inside SAGEConv (lin_l, lin_r) and the decoder MLP
 *           (model/layers.py:35-56, model/encoder_decoder.py:55-72) and their backward.
 * trans_a: A is stored [k,m] (lda = m-stride);  trans_b: B is stored [n,k] (the
 * nn.Linear weight layout).  act: 0 = none, 1 = relu.
 * Per output element the sum is one k-ascending fma chain (bitwise = oracle/spmm_ref.c), except
 * when the output grid is too small to fill the chip and k >= 2048 (weight gradients
 * dW = dY^T X): then, if the caller passes the workspace mi_gemm_workspace_bytes asks for, K is
 * cut into slices whose chains are added in a fixed association (still deterministic).  ws may be null.
 * ---------------------------------------------------------------------------------- */
size_t mi_gemm_workspace_bytes(int64_t m, int64_t n, int64_t k);
int    mi_gemm_f32(int32_t trans_a, int32_t trans_b, int64_t m, int64_t n, int64_t k,
                   const float* A, int64_t lda, const float* B, int64_t ldb,
                   const float* bias, float* C, int64_t ldc,
                   int32_t accumulate, int32_t act, void* ws, size_t ws_bytes, mi_stream_t stream);

/* Grouped form: any number of independent products, one launch per 8 (the ranker's per-batch products are
 * launch-bound, see csrc/gemm.hip).  Problem i:
 *     C = act( A @ B  [+ A2 @ B2]  + bias + (accumulate ? C : 0) )
 * with op(A) [m, k], op(B) [k, n] as in mi_gemm_f32 and an optional second pair sharing m, n and the trans flags
 * (k2 = 0: none) — lin_l(agg) + lin_r(x_dst) of SAGEConv (model/layers.py:11-24) as ONE product.  a_mask (nullable):
 * same storage layout as A; A's element is read as 0 where the mask element is <= 0 — the backward of a relu fused
 * into the producing product.  Operands must be float4-addressable (leading dimensions and k multiples of 4, 16-byte
 * aligned bases); otherwise MI_ERR_UNSUPPORTED and the caller uses mi_gemm_f32.  Long-k / tiny-output problems are
 * split over k like mi_gemm_f32 (same slice rule, one grouped reduce).  ws: mi_gemm_group_workspace_bytes(). */
typedef struct mi_gemm_problem {
    int32_t trans_a, trans_b;
    int64_t m, n, k;
    const float* A; int64_t lda;
    const float* B; int64_t ldb;
    int64_t k2;
    const float* A2; int64_t lda2;
    const float* B2; int64_t ldb2;
    const float* a_mask;
    const float* bias;
    float* C; int64_t ldc;
    int32_t accumulate, act;
} mi_gemm_problem;
size_t mi_gemm_group_workspace_bytes(const mi_gemm_problem* problems, int32_t n);
int mi_gemm_group_f32(const mi_gemm_problem* problems, int32_t n, void* ws, size_t ws_bytes, mi_stream_t stream);
/* 1 when mi_gemm_group_f32 would take every problem (float4-addressable operands), 0 when it would return
 * MI_ERR_UNSUPPORTED; enqueues nothing. */
int mi_gemm_group_supported(const mi_gemm_problem* problems, int32_t n);

/* ------------------------------------------------------------------------------------
 * K10  batched exact top-K with per-user exclusion.
 * replaces: make_predictions_for_user at utils/metrics_lightgcn.py:125-142
 *           (scores = e_u @ E_i^T; topk(k+|ignore|); order-preserving setdiff; [:k]).
 * scores[u, i] = fmaf-chain over d ascending of user_emb[uid[u], :] . item_emb[i, :]
 * (bitwise equal to oracle/spmm_ref.c:ref_scores_fma_f32); excluded items (excl_ptr/excl_idx: CSR by
 * position in uid, item ids sorted or not) never appear; output int64[n_q, k] item ids
 * by descending score, ties by ascending item id; -1 pads when fewer than k remain.
 * k <= 1024.  For n_items >= 32 768 and d <= 128 (multiple of 4, 16-byte aligned tables) the score block is never
 * written: scores are compared in the MFMA epilogue with a per-row threshold taken from a 4 096-item sample and only the
 * survivors are kept; a row whose survivors are fewer than k or overflow its list recomputes its scores and takes
 * the exact multi-pass selection, so the result is the same as on the materialised path (smaller item sets).
 * For d = 64 / 128 and k <= 256 the candidates are picked by a bf16x3 PREFILTER (csrc/topk_prefilter.hpp): both tables
 * split into bf16 hi + lo parts, u_hi.i_hi + u_lo.i_hi + u_hi.i_lo on the bf16 MFMA, every score within a proven
 * eps(u) = 2^-12 |u| max|i| of the f32 chain; the items that can still reach the answer under that bound are scored
 * again with the exact fma chain and the answer is picked from those — ids and scores are the f32 path's bit for bit
 * (environment LAPLACE_TOPK_PREFILTER=0, read per call, selects the f32 fused kernel: A/B and tests of both; about 2.7 x
 * the f32 kernel's throughput: 10.6 M against 3.9 M users/s at k = 12, 100 K items, D = 128).
 * Workspace (mi_topk_workspace_bytes): address space for the [n_q, n_items] fp32 score block (touched only by
 * fallback rows on the fused path) + sample scores, exclusion bitmap, candidate lists, the split tables: callers chunk
 * n_q (multiples of 2 048 queries fill the prefilter's grid in whole rounds).
 * ---------------------------------------------------------------------------------- */
size_t mi_topk_workspace_bytes(int64_t n_q, int64_t n_items, int64_t k);
int    mi_topk_excl_f32(int64_t n_q, int64_t n_items, int64_t d, int64_t k,
                        const int64_t* uid,
                        const float* user_emb, int64_t ldu,
                        const float* item_emb, int64_t ldi,
                        const int32_t* excl_ptr, const int32_t* excl_idx,
                        int64_t* out_idx, float* out_score /* nullable */,
                        void* ws, size_t ws_bytes, mi_stream_t stream);
/* The same with flags.  MI_TOPK_ITEMS_PREPARED: `ws` still holds the ITEM side of the bf16 prefilter (the sampled item rows, the
 * bf16 hi / lo split of the item table and of the sample, the largest |item|^2) from the previous call on this stream with
 * the same item table, n_q, n_items, d and k — callers that cut one request into equal chunks of queries set it from the second
 * chunk on (one split of a 100 K-item table is ~40 us of a ~550 us chunk).  Ignored on the paths without a prefilter. */
#define MI_TOPK_ITEMS_PREPARED 1u
int    mi_topk_excl_ex_f32(int64_t n_q, int64_t n_items, int64_t d, int64_t k, const int64_t* uid,
                           const float* user_emb, int64_t ldu, const float* item_emb, int64_t ldi,
                           const int32_t* excl_ptr, const int32_t* excl_idx, int64_t* out_idx,
                           float* out_score, void* ws, size_t ws_bytes, uint32_t flags, mi_stream_t stream);

/* Diagnostic of K10's bf16x3 prefilter (csrc/topk_prefilter.hpp), d = 64 / 128: scores[q, i] = the APPROXIMATE score the
 * prefilter compares with its thresholds, eps[q] = the bound it assumes, |scores[q, i] - exact fma chain| <= eps[q] for
 * every item (tests/test_gpu_topk_gemm.py asserts it on adversarial tables).  Not on any product path: the candidates'
 * exact scores are what mi_topk_excl_f32 returns. */
size_t mi_topk_prefilter_scores_workspace_bytes(int64_t n_q, int64_t n_items);
int    mi_topk_prefilter_scores_f32(int64_t n_q, int64_t n_items, int64_t d,
                                    const int64_t* uid /* nullable */,
                                    const float* user_emb, int64_t ldu,
                                    const float* item_emb, int64_t ldi,
                                    float* scores /* [n_q, n_items] */, float* eps /* [n_q] */,
                                    void* ws, size_t ws_bytes, mi_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K5  SAGEConv message passing.
 * replaces: SAGEConv.propagate -> torch_scatter.scatter(x_j, index, reduce=aggr) reached from
 *           model/layers.py:11-24 via model/encoder_decoder.py:32,44.
 * The CSR is by DESTINATION node (rowptr over dst, col = source ids, sorted).
 *   aggr="add":  mi_spmm_csr_f32 with val = 1          (no [E, C] message tensor, no atomics)
 *   aggr="mean": mi_spmm_csr_f32 with val = 1/in-degree (mi_scale_csr_f32); empty segment -> 0
 *   aggr="max":  mi_segment_max_f32 below; empty segment -> 0; `arg` int32[n_dst, d] (nullable)
 *                receives the winning source id (-1 for empty) for the backward
 *                dX[s, c] = sum over destinations r of s with arg[r, c] == s of dY[r, c],
 *                computed per SOURCE row over the by-source CSR of the same relation (src_rowptr over
 *                sources, src_col = destination ids, sorted): one writer per element, fixed order,
 *                no float atomics; every row of dX is written.
 * ---------------------------------------------------------------------------------- */
int mi_segment_max_f32(int64_t n_dst, int64_t d, const int32_t* rowptr, const int32_t* col,
                       const float* X, int64_t ldx, float* Y, int64_t ldy, int32_t* arg,
                       mi_stream_t stream);
int mi_segment_max_bwd_f32(int64_t n_src, int64_t d, const int32_t* src_rowptr, const int32_t* src_col,
                           const int32_t* arg, const float* dY, int64_t ldy, float* dX, int64_t ldx,
                           mi_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K7  multi-column categorical embedding lookup + concat, with max_norm renorm.
 * replaces: Encoder_Decoder_Model.__embedding at model/encoder_decoder.py:116-125
 *           (Embedding(num_cat+1, size, max_norm=1)(x[:, i]) per column, cat on dim 1).
 * x int64[n, n_cols] (device).  tables / table_rows / dims are HOST arrays of n_cols (<= 16)
 * entries: device pointer to float[rows_c, dim_c], rows_c, dim_c.  out float[n, sum(dim_c)].
 * A looked-up row whose L2 norm exceeds max_norm is scaled by max_norm / (norm + 1e-7) on the
 * fly — the value torch's in-place embedding_renorm_ would leave in the table — so the stored
 * (frozen) table is never written.  max_norm <= 0 disables it.
 * ---------------------------------------------------------------------------------- */
int mi_embed_concat_f32(int64_t n, int32_t n_cols, const int64_t* x,
                        const float* const* tables, const int64_t* table_rows, const int32_t* dims,
                        float max_norm, float* out, int64_t ldo, mi_stream_t stream);

/* ------------------------------------------------------------------------------------
 * K8  per-node-type BatchNorm1d over the nodes of a batch.
 * replaces: self.encoder_layer_norm_customer / _article = BatchNorm1d(out_channels) applied at
 *           model/encoder_decoder.py:144-150 (declared :98-99) and its autograd backward.
 * training != 0: batch statistics (biased variance for the normalisation; running_mean /
 * running_var updated in place with `momentum` and the UNBIASED variance, torch semantics;
 * pass both null to skip); save_mean / save_invstd float[c] are written for the backward.
 * training == 0: y = (x - running_mean) / sqrt(running_var + eps) * gamma + beta.
 * gamma / beta nullable (affine=False).  Backward: dX (nullable) and dgamma / dbeta (nullable)
 * from the saved statistics.  Sums are reduced in double in a fixed order: bitwise reproducible.
 * ws: mi_batchnorm_workspace_bytes(c) bytes, device.  c <= 512.
 * ---------------------------------------------------------------------------------- */
size_t mi_batchnorm_workspace_bytes(int64_t c);
int mi_batchnorm_fwd_f32(int64_t n, int64_t c, const float* X, int64_t ldx,
                         const float* gamma, const float* beta,
                         float* running_mean, float* running_var, float momentum, float eps,
                         int32_t training, float* save_mean, float* save_invstd,
                         float* Y, int64_t ldy, void* ws, size_t ws_bytes, mi_stream_t stream);
int mi_batchnorm_bwd_f32(int64_t n, int64_t c, const float* X, int64_t ldx,
                         const float* dY, int64_t ldy, const float* gamma,
                         const float* save_mean, const float* save_invstd,
                         float* dX, int64_t lddx, float* dgamma, float* dbeta,
                         void* ws, size_t ws_bytes, mi_stream_t stream);

/* ------------------------------------------------------------------------------------
 * b9  the ranker's objective and its gradient in one launch.
 * replaces: torch.nn.BCEWithLogitsLoss()(logits, edge_label.float()) at training.py:26-31 and the backward of it.
 * loss[0] = mean(max(x,0) - x*y + log1p(exp(-|x|))); dlogits (nullable) = (sigmoid(x) - y) / n.  One workgroup,
 * double sums in a fixed tree.
 * ---------------------------------------------------------------------------------- */
int mi_bce_logits_f32(int64_t n, const float* logits, const float* labels, float* loss, float* dlogits,
                      mi_stream_t stream);

/* ------------------------------------------------------------------------------------
 * b1  weight gradients of SAGEConv relations (model/layers.py:11-24: out = lin_l(AGG x_src) + lin_r(x_dst)):
 *   gw1 [m, n1] = (dy * relu')^T b1,   gb [m] = (dy * relu')^T 1 (nullable),   gw2 [m, n2] = (dy * relu')^T b2 (n2 = 0: none)
 * with dy [k, m] the gradient at the layer's output (k = destination nodes), mask [k, m] nullable (dy is read as 0 where
 * mask <= 0: the relu behind the layer), b1 = the aggregated source features [k, n1], b2 = the destination features [k, n2].
 * replaces: what autograd derives for the two Linear layers of torch_geometric.nn.SAGEConv (three transposed products).
 * All matrices dense row-major (leading dimension = width).  m a multiple of 32, <= 128; n1, n2 <= 512; up to 4 problems
 * per call (the two relations of a layer).  No LDS staging: rows of dy / b1 / b2 go straight into the operand registers
 * of the f32 32x32x2 MFMA; K is cut into slices whose partial outputs are summed in slice order (deterministic).
 * MI_ERR_UNSUPPORTED (nothing enqueued) for other shapes: callers use the grouped GEMM.
 * ---------------------------------------------------------------------------------- */
typedef struct mi_wgrad_problem {
    int64_t k;
    int32_t m, n1, n2, reserved;
    const float *dy, *mask, *b1, *b2;
    float *gw1, *gb, *gw2;
} mi_wgrad_problem;
int    mi_sage_wgrad_supported(const mi_wgrad_problem* problems, int32_t n);
size_t mi_sage_wgrad_workspace_bytes(const mi_wgrad_problem* problems, int32_t n);
int    mi_sage_wgrad_f32(const mi_wgrad_problem* problems, int32_t n, void* ws, size_t ws_bytes, mi_stream_t stream);

/* ------------------------------------------------------------------------------------
 * b7  backward of a Linear layer with ONE output feature — the last decoder layer (model/encoder_decoder.py:66-72,
 * Linear(128, 1)): dx[r, :] = dy[r] * w (nullable), gw = sum_r dy[r] x[r, :], gb = sum_r dy[r] (nullable).
 * replaces: the three [n, 1]-shaped products autograd derives for that layer (torch.nn.Linear backward).
 * Deterministic (bands of 64 rows reduced in band order).  ws: mi_linear1_bwd_workspace_bytes(n, in).
 * ---------------------------------------------------------------------------------- */
size_t mi_linear1_bwd_workspace_bytes(int64_t n, int64_t in);
int    mi_linear1_bwd_f32(int64_t n, int64_t in, const float* dy, const float* w, const float* x, int64_t ldx,
                          float* dx, int64_t lddx, float* gw, float* gb, void* ws, size_t ws_bytes, mi_stream_t stream);

/* ------------------------------------------------------------------------------------
 * b7  decoder input: z = cat(z_user[row], z_item[col], dim=-1) over the label edges.
 * replaces: EdgeDecoder.forward's two index_selects + cat at model/encoder_decoder.py:57-63
 *           and their backward (index_add into the two node tables).
 * out float[n_edges, cu + ci].  Backward, one call per node table: dZ[v, :] = sum over the
 * label edges e (ascending) with idx[e] == v of dOut[e, off : off + c]; dZ must be zero-filled
 * by the caller (rows no edge names stay zero).  One writer per row, fixed order, no atomics;
 * supports up to mi_gather_cat_bwd_max_edges() label edges (MI_ERR_UNSUPPORTED beyond).
 * ---------------------------------------------------------------------------------- */
int mi_gather_cat_f32(int64_t n_edges, int64_t cu, int64_t ci, const int64_t* row, const int64_t* col,
                      const float* Zu, int64_t ldu, const float* Zi, int64_t ldi,
                      float* out, int64_t ldo, mi_stream_t stream);
int64_t mi_gather_cat_bwd_max_edges(void);
int mi_gather_cat_bwd_f32(int64_t n_edges, int64_t c, int64_t off, const int64_t* idx,
                          const float* dOut, int64_t ldo, float* dZ, int64_t ldz, mi_stream_t stream);

/* ------------------------------------------------------------------------------------
 * b9  native executor of ONE ranker training iteration (round 3).
 * replaces: the loop body of training.py:19-34 (`zero_grad -> model(...) -> BCEWithLogitsLoss -> backward ->
 *           optimizer.step`) for the model shape the reference builds by default (model/encoder_decoder.py:75-150,
 *           run_pipeline.py:47-73): two node types (customer, article) with categorical features and frozen embedding
 *           tables, the two relations customer -buys-> article and its reverse, L SAGEConv layers (add / mean) duplicated
 *           per relation, per-type BatchNorm1d, an MLP decoder over cat(z_customer[row], z_article[col]), Adam.
 * One call enqueues the whole iteration — the same mi_* launches, in the same order and with the same operands, as
 * the package's FusedRankerStep issues one by one from Python — so the host cost is one FFI call instead of ~75 op
 * calls.  Every buffer the iteration needs is carved from `ws` (mi_ranker_step_workspace_bytes); gradients are
 * written to the caller's persistent gradient buffers and consumed by a multi-tensor Adam launch (skippable:
 * data-parallel callers all-reduce the gradients first).  Feature dropout draws Philox4x32-10 masks keyed on
 * (seed, step, site) and regenerates them in the backward.  Returns MI_ERR_UNSUPPORTED (nothing enqueued) for shapes
 * outside the above; the caller then runs the op-by-op path.
 * ---------------------------------------------------------------------------------- */
#define MI_RANKER_MAX_LAYERS 4
#define MI_RANKER_MAX_COLS 16
#define MI_RANKER_MAX_PARAMS 48
typedef struct mi_ranker_conv {        /* SAGEConv of one relation in one layer: out = lin_l(AGG x_src) + lin_r(x_dst) */
    const float *w_l, *b_l, *w_r;      /* [c_out, c_src], [c_out] (nullable), [c_out, c_dst] */
    float *gw_l, *gb_l, *gw_r;         /* gradients, same shapes (gb_l nullable with b_l) */
    int32_t c_src, c_dst, c_out, reserved;
} mi_ranker_conv;
typedef struct mi_ranker_norm {        /* BatchNorm1d of one node type */
    const float *gamma, *beta;         /* nullable (affine=False) */
    float *running_mean, *running_var; /* nullable (track_running_stats=False) */
    int64_t* num_batches_tracked;      /* nullable; incremented by one */
    float *g_gamma, *g_beta;
    float momentum, eps;
} mi_ranker_norm;
typedef struct mi_ranker_linear {
    const float *w, *b;                /* [out, in], [out] (nullable) */
    float *gw, *gb;
    int32_t in, out;
} mi_ranker_linear;
typedef struct mi_ranker_param {       /* one tensor of the optimizer's parameter list */
    float *p, *g, *m, *v;
    int64_t n;
} mi_ranker_param;
typedef struct mi_ranker_model {
    int32_t n_enc_layers, n_dec_layers;
    int32_t aggr;                      /* 0 = add, 1 = mean */
    int32_t batch_normalize;
    float   p_dropout;                 /* feature dropout in front of every non-last encoder / decoder layer; 0 = none */
    float   max_norm;                  /* of the embedding lookups (reference: 1) */
    /* node type 0 = customer, 1 = article */
    int32_t n_cols[2];
    const float* tables[2][MI_RANKER_MAX_COLS];
    int64_t table_rows[2][MI_RANKER_MAX_COLS];
    int32_t dims[2][MI_RANKER_MAX_COLS];
    /* conv[l][0]: customer -> article (destination: article); conv[l][1]: article -> customer */
    mi_ranker_conv conv[MI_RANKER_MAX_LAYERS][2];
    mi_ranker_norm norm[2];
    mi_ranker_linear dec[MI_RANKER_MAX_LAYERS];
    int32_t n_params, apply_adam;      /* apply_adam = 0: stop after the gradients */
    mi_ranker_param params[MI_RANKER_MAX_PARAMS];
    double lr, beta1, beta2, eps;
    int64_t step;                      /* Adam step of THIS iteration, counts from 1 */
    const float* ones4;                /* device float[n_ones, 4] of ones (bias gradients ride in the grouped dW launch) */
    int64_t n_ones;
} mi_ranker_model;
typedef struct mi_ranker_batch {
    int64_t n_nodes[2];                /* customers, articles of the batch */
    const int64_t* x[2];               /* int64[n_nodes[t], n_cols[t]] categorical features */
    /* the batch's `buys` edges as two sorted CSRs (what mi_sampler_emit_csr writes) */
    const int32_t *by_customer_ptr, *by_customer_col;   /* rows = customers, columns = articles */
    const int32_t *by_article_ptr, *by_article_col;     /* rows = articles, columns = customers */
    int64_t nnz;
    int64_t n_label;
    const int64_t *label_row, *label_col;  /* customer / article index of every label edge */
    const int64_t* label;                  /* int64[n_label] 0 / 1 (the sampler's edge_label), or null when ... */
    uint64_t seed, step;                   /* dropout stream */
    float* loss;                           /* device float[1] */
    const float* label_f32;                /* ... the caller already holds the labels as float[n_label] */
    /* optional second stream: pairs of independent small launches (customer / article twins) then run side by side.
     * aux_stream: a hipStream_t; ev_fork / ev_join: two hipEvent_t (timing disabled) owned by the caller and used by no one
     * else while the call is in flight.  All three null = everything on `stream`.  Results do not depend on it. */
    void *aux_stream, *ev_fork, *ev_join;
    /* INFERENCE (round 4; model(x, edge_index, edge_label_index) under model.eval(): run_submission.py:55-58): non-null =
     * forward only, in evaluation mode — no dropout, BatchNorm with its running statistics (required) — and the decoder's
     * output per label edge goes to logits[n_label].  label / label_f32 / loss, every gradient pointer of the model and its
     * parameter list are then neither required nor read; nothing of the model is written. */
    float* logits;
} mi_ranker_batch;
int64_t mi_ranker_sizeof(int32_t which);  /* sizeof of: 0 model, 1 batch, 2 conv, 3 norm, 4 linear, 5 param (binding self-check) */
size_t mi_ranker_step_workspace_bytes(const mi_ranker_model* model, const mi_ranker_batch* batch);
int    mi_ranker_step_f32(const mi_ranker_model* model, const mi_ranker_batch* batch, void* ws, size_t ws_bytes,
                          mi_stream_t stream);
/* The validation pass of mi_ranker_step_f32 alone: 0 when the call would be taken, MI_ERR_UNSUPPORTED / MI_ERR_ARG when it
 * would be declined.  Enqueues nothing and reads no device memory (host-side walk of the two descriptors), so data-parallel
 * callers can agree on taking or declining a batch BEFORE any rank enters the gradient all-reduce (round 4). */
int    mi_ranker_step_check(const mi_ranker_model* model, const mi_ranker_batch* batch, void* ws, size_t ws_bytes);
/* The update alone, for data-parallel callers: run mi_ranker_step_f32 with apply_adam = 0, all-reduce(sum) the gradient
 * buffers (params[i].g; the host side keeps them in ONE flat allocation so that is one collective), then this:
 * Adam (a9's arithmetic, model->lr/beta/eps/step) over model->params with every gradient multiplied by grad_scale
 * (1 / world size) first.  The gradient buffers are left as reduced (unscaled).  One launch. */
int    mi_ranker_adam_f32(const mi_ranker_model* model, float grad_scale, mi_stream_t stream);

/* ------------------------------------------------------------------------------------
 * N3  candidate matcher: items bought by users who share an item with the query user.
 * replaces: UsersWithCommonItemsMatcher.get_matches at
 *           data/matching/users_with_common_purchases.py:14-26 (flatten of the co-purchasers'
 *           lists, t.cat of all their article lists, [:k]).
 * users_ptr/users_idx and articles_ptr/articles_idx: the adjacency lists IN LIST ORDER (the
 * reference's edges_*.pt / rev_edges_*.pt) as int32 CSR on device.  query_users int64[n] (null =
 * users 0..n-1).  out int32[n, k]: the first k entries of the concatenation, -1 padded;
 * out_count int32[n] (nullable): entries written.  One wavefront per query, stops at k.
 * ---------------------------------------------------------------------------------- */
int mi_match_common_items_i32(int64_t n_queries, const int64_t* query_users,
                              const int32_t* users_ptr, const int32_t* users_idx,
                              const int32_t* articles_ptr, const int32_t* articles_idx,
                              int32_t k, int32_t* out, int32_t* out_count, mi_stream_t stream);

/* N3  candidate matcher: items bought by the customers at the query user's location.
 * replaces: UsersSameLocationMatcher.get_matches at data/matching/fashion/users_same_location.py:15-25
 *           (t.cat of the article lists of customers_per_location[location_for_user[u]], [:k]).
 * location_of_user int32[num_users] (negative = unknown: no proposals); loc_ptr/loc_idx: customers per location in
 * list order; users_ptr/users_idx as above.  Output as mi_match_common_items_i32. */
int mi_match_same_location_i32(int64_t n_queries, const int64_t* query_users, const int32_t* location_of_user,
                               const int32_t* loc_ptr, const int32_t* loc_idx,
                               const int32_t* users_ptr, const int32_t* users_idx,
                               int32_t k, int32_t* out, int32_t* out_count, mi_stream_t stream);

/* ------------------------------------------------------------------------------------
 * N1  on-device N-hop subgraph sampler for the ranker.
 * replaces: GraphDataset.__getitem__ + helpers (data/dataset.py:39-309, train mode) for a whole
 *           batch of seed users, and the PyG collate of the resulting HeteroData items
 *           (data/data_loader.py:48): sampled positive label edges (with
 *           replacement), negatives (fast path / exact path), the seed user's own edges, the
 *           N-hop neighbourhood with the fan-out cap on both frontiers, sorted-unique relabelling.
 * users_ptr/users_idx: CSR user -> article ids IN LIST ORDER (edges_*.pt); articles_ptr/idx:
 * CSR article -> user ids (rev_edges_*.pt); all int32, device.  id_max = max article id of the
 * graph, num_edges its edge count (the fast-path test E / n_neg > 100).  max_pos / max_neg:
 * capacity per sample of the label lists, >= max(2, floor(max_degree*ratio)) and
 * >= max(k-1, int(neg_ratio*max_pos)).
 * Two calls per batch because the output sizes are data dependent:
 *   mi_sampler_count  runs the walk, SYNCHRONISES, returns totals_host[4] = {user nodes,
 *                     article nodes, message-passing edges, label edges} of the collated batch;
 *   mi_sampler_count_async  the same without the wait: the four totals land in caller-owned host
 *                     memory (pin it) when `stream` reaches that point — the caller waits on an event
 *                     of its own, so a batch can be sampled on a side stream while the previous one trains;
 *   mi_sampler_emit   writes user_ids int64[tot0], article_ids int64[tot1] (global ids, sorted
 *                     within each sample), edge_index int64[2, tot2], edge_label_index
 *                     int64[2, tot3] (batch-local ids), edge_label int64[tot3],
 *                     user_ptr / article_ptr int64[batch+1] (node offsets per sample).
 *   mi_sampler_emit_csr  (optional, after mi_sampler_count*, same ws) writes the SAME message-passing edges as the
 *                     two CSRs mi_coo_to_csr_i32 would build from edge_index — customers x articles
 *                     (customer_rowptr int32[tot0+1], customer_col int32[tot2]) and articles x customers
 *                     (article_rowptr int32[tot1+1], article_col int32[tot2]), both sorted by (row, column),
 *                     batch-local ids — so the encoder needs no sort per batch (model/layers.py BipartiteGraph;
 *                     replaces the SparseTensor construction inside PyG's SAGEConv propagate).  article_cursor:
 *                     int32[tot1] scratch.  MI_ERR_UNSUPPORTED when n_hops*num_neighbors > 512.
 * Frontier: when the queued articles' user lists total more than reject_min_entries, the
 * uniform num_neighbors-subset of their distinct unexplored users is drawn by rejection (position of
 * the concatenated lists, acceptance 1/multiplicity) instead of materialising the set — hub
 * articles carry 10^5..10^6 users; the bound guarantees >= num_neighbors candidates exist.
 * Randomness: Philox4x32-10 keyed on (seed, step); bit-exact mirror in oracle/sampler_ref.py.
 * ---------------------------------------------------------------------------------- */
typedef struct mi_sampler_desc {
    int32_t batch, n_hops, num_neighbors, k;
    int32_t randomization, max_pos, max_neg, reserved;
    int64_t num_users, num_articles, num_edges, id_max;
    const int32_t* users_ptr;    const int32_t* users_idx;
    const int32_t* articles_ptr; const int32_t* articles_idx;
    double positive_edges_ratio, negative_edges_ratio;
    int64_t reject_min_entries; /* 0 = default 4*n*(n*(hops+1)+1); must be >= n*(n*(hops+1)+1) */
    /* evaluation mode (data/dataset.py:94-105, train=False): when cand_ptr is non-null the label-0 edges of seed
     * user u are not sampled but the ids that occur exactly once in cat(unique(candidates of u), items of u) —
     * candidates that are not purchases plus, as the reference writes it, purchases no matcher proposed — in
     * ascending order.  cand_ptr int32[num_users + 1] / cand_idx int32[]: the matchers' proposals per user
     * (duplicates allowed, any order).  max_neg >= max_u(candidates of u + items of u). */
    const int32_t* cand_ptr;     const int32_t* cand_idx;
} mi_sampler_desc;

size_t mi_sampler_workspace_bytes(const mi_sampler_desc* d);
int    mi_sampler_count(const mi_sampler_desc* d, const int64_t* seed_users, uint64_t seed, uint64_t step,
                        void* ws, size_t ws_bytes, int64_t* totals_host, mi_stream_t stream);
int    mi_sampler_count_async(const mi_sampler_desc* d, const int64_t* seed_users, uint64_t seed, uint64_t step,
                              void* ws, size_t ws_bytes, int32_t* totals_pinned, mi_stream_t stream);
int    mi_sampler_emit(const mi_sampler_desc* d, const int64_t* seed_users, void* ws, size_t ws_bytes,
                       const int64_t* totals_host, int64_t* user_ids, int64_t* article_ids,
                       int64_t* edge_index, int64_t* edge_label_index, int64_t* edge_label,
                       int64_t* user_ptr, int64_t* article_ptr, mi_stream_t stream);
/* mi_sampler_emit with edge_index / edge_label_index written as [3, n] when three_rows != 0: row 2 repeats row 0, so rows 0..1
 * are the `buys` relation's index and rows 1..2 the reversed relation's (`rev_buys`: data/dataset.py builds it with flip(0)) —
 * two views of one buffer instead of two flip launches per batch (round 4). */
int    mi_sampler_emit3(const mi_sampler_desc* d, const int64_t* seed_users, void* ws, size_t ws_bytes,
                        const int64_t* totals_host, int64_t* user_ids, int64_t* article_ids,
                        int64_t* edge_index, int64_t* edge_label_index, int64_t* edge_label,
                        int64_t* user_ptr, int64_t* article_ptr, int32_t three_rows, mi_stream_t stream);
int    mi_sampler_emit_csr(const mi_sampler_desc* d, void* ws, size_t ws_bytes, const int64_t* totals_host,
                           int32_t* customer_rowptr, int32_t* customer_col, int32_t* article_rowptr,
                           int32_t* article_col, int32_t* article_cursor, mi_stream_t stream);

/* ------------------------------------------------------------------------------------
 * N5  PinSAGE samplers (reference: pinsage/sampler.py:16-106 over DGL's random_walk and
 *     PinSAGESampler; DGL is absent and the reference's pinsage/ cannot import — SURVEY F11 — so the
 *     semantics are restated in oracle/pinsage_ref.py, which mirrors these kernels bit for bit).
 * iu_*: CSR item -> users; ui_*: CSR user -> items (int32, device).
 * mi_pinsage_item_pairs: heads uniform over items; tails[b] = end of one item->user->item walk from
 *   heads[b] (uniform neighbour per step), -1 when the walk dies; neg_tails uniform.  int64[batch] each.
 * mi_pinsage_neighbors: per seed, num_walks walks of walk_length traversals with termination
 *   probability restart_prob before every traversal but the first; the items reached at the end of
 *   each traversal are counted and the num_neighbors most visited (count desc, id asc) returned:
 *   neighbors int64[n_seeds, T] (-1 padded), weights int64[n_seeds, T] (visit counts).
 * ---------------------------------------------------------------------------------- */
int    mi_pinsage_item_pairs(int64_t batch, int64_t n_items,
                             const int32_t* iu_ptr, const int32_t* iu_idx,
                             const int32_t* ui_ptr, const int32_t* ui_idx,
                             uint64_t seed, uint64_t step,
                             int64_t* heads, int64_t* tails, int64_t* neg_tails, mi_stream_t stream);
size_t mi_pinsage_neighbors_workspace_bytes(int64_t n_seeds, int32_t walk_length, int32_t num_walks);
int    mi_pinsage_neighbors(int64_t n_seeds, const int64_t* seeds,
                            const int32_t* iu_ptr, const int32_t* iu_idx,
                            const int32_t* ui_ptr, const int32_t* ui_idx,
                            int32_t walk_length, double restart_prob, int32_t num_walks,
                            int32_t num_neighbors, int32_t layer, uint64_t seed, uint64_t step,
                            int64_t* neighbors, int64_t* weights,
                            void* ws, size_t ws_bytes, mi_stream_t stream);

/* N5, whole batch on the device (round 3): item pairs -> seeds -> per layer (neighbours -> block), six launches for the
 * reference's two layers, no host read-back inside.  replaces: sample_from_item_pairs + sample_blocks at
 * pinsage/sampler.py:73-106 (dgl.to_block numbering: destination nodes first, new sources ascending; frontier edges
 * that repeat a label pair removed) AND the two sorts per block the model needed for its weighted-mean aggregation
 * (pinsage/layers.py:150-160): every block comes with its CSR by destination and by source, values
 * w / max(sum of the destination's w, 1).
 * Buffers are sized for the upper bounds n_max(0) = 3 * batch, n_max(l + 1) = n_max(l) * (1 + num_neighbors); the actual
 * counts land in counts: [0] surviving pairs, [1] seeds, [2 + 2l] nodes of block l, [3 + 2l] edges of block l (block 0 =
 * the one built FIRST, around the seeds; the model consumes them in reverse).  MI_ERR_UNSUPPORTED when batch > 1024 or a
 * layer's n_max * num_neighbors exceeds 16384 (the caller then builds the blocks with its own index ops). */
#define MI_PINSAGE_MAX_LAYERS 4
typedef struct mi_pinsage_batch_desc {
    int64_t batch, n_items;
    const int32_t *iu_ptr, *iu_idx, *ui_ptr, *ui_idx;
    int32_t walk_length, num_walks, num_neighbors, num_layers;
    double  restart_prob;
    int32_t* pos_scratch;          /* device int32[n_items], all -1 on entry and on return */
} mi_pinsage_batch_desc;
typedef struct mi_pinsage_block_out {
    int64_t *src_ids, *edge_src, *edge_dst;   /* [n_max*(1+T)], [n_max*T], [n_max*T] */
    float*   weights;                          /* [n_max*T] visit counts */
    int32_t *dst_rowptr, *dst_col; float* dst_val;   /* [n_max+1], [n_max*T], [n_max*T] */
    int32_t *src_rowptr, *src_col; float* src_val;   /* [n_max*(1+T)+1], [n_max*T], [n_max*T] */
} mi_pinsage_block_out;
typedef struct mi_pinsage_batch_out {
    int64_t *seeds, *pos_u, *pos_v, *neg_v;   /* [3*batch], [batch] x 3: positions of head / tail / negative in seeds */
    int32_t* counts;                          /* [2 + 2*num_layers] */
    mi_pinsage_block_out blocks[MI_PINSAGE_MAX_LAYERS];
} mi_pinsage_batch_out;
size_t mi_pinsage_batch_workspace_bytes(int64_t batch, int32_t walk_length, int32_t num_walks, int32_t num_neighbors,
                                        int32_t num_layers);
int    mi_pinsage_sample_batch(const mi_pinsage_batch_desc* desc, uint64_t seed, uint64_t step,
                               const mi_pinsage_batch_out* out, void* ws, size_t ws_bytes, mi_stream_t stream);

/* ------------------------------------------------------------------------------------
 * N5  one PinSAGE training iteration as ONE call (round 3).
 * replaces: the loop body of pinsage/model.py:118-131 (train) on PinSAGEModel (pinsage/model.py:16-34): LinearProjector
 *           over the item-id feature (an embedding row per node), SAGENet of WeightedSAGEConv layers
 *           (pinsage/layers.py:121-156: n = relu(Q dropout(h_src)), weighted mean over the sampled neighbours,
 *           z = relu(W dropout([agg, h_dst])), row L2-normalisation), h_dst + SAGENet output, ItemToItemScorer
 *           (pinsage/layers.py:181-203: dot + both endpoints' bias), hinge (neg - pos + 1).clamp(min = 0).mean(),
 *           backward, torch.optim.Adam step.  Issued op by op through autograd the iteration is ~200 launches of a few
 *           microseconds (1.8 ms at the reference's 32 pairs per batch); here ~40.
 * The batch is one built by mi_pinsage_sample_batch (blocks in MODEL order: input layer first; destination nodes are the
 * first rows of every block's src_ids, so the seeds are the first n_seeds rows of block 0's).  Dropout: Philox4x32-10
 * masks keyed on (seed, step, site), regenerated in the backward.  Gradients: the embedding table's and the scorer
 * bias's gradient buffers are DENSE and must be all zero on entry; the rows of this batch are written, Adam runs over
 * the whole tables (torch.optim.Adam's dense semantics: every moment decays every step) and the rows are zeroed again
 * — unless apply_adam = 0, which stops after the gradients and leaves them in place (the caller zeroes them).
 * hidden % 4 == 0, hidden <= 128.  Everything is deterministic (no float atomics).
 * ---------------------------------------------------------------------------------- */
#define MI_PINSAGE_MAX_PARAMS 24
typedef struct mi_pinsage_conv {
    const float *q_w, *q_b, *w_w, *w_b;        /* Q [hidden, hidden] + [hidden]; W [hidden, 2 hidden] + [hidden] */
    float *g_q_w, *g_q_b, *g_w_w, *g_w_b;      /* gradients, same shapes */
} mi_pinsage_conv;
typedef struct mi_pinsage_model {
    int32_t n_layers, hidden;
    int64_t n_items;
    float *proj, *g_proj, *m_proj, *v_proj;    /* [n_items + 1, hidden]: table, dense gradient (zero on entry), Adam moments */
    const float* bias; float* g_bias;          /* [n_items] scorer bias and its dense gradient (zero on entry) */
    mi_pinsage_conv conv[MI_PINSAGE_MAX_LAYERS];
    mi_ranker_param params[MI_PINSAGE_MAX_PARAMS];   /* every parameter EXCEPT proj (the scorer bias included): multi-tensor Adam */
    int32_t n_params, apply_adam;
    float   p_dropout; int32_t reserved;
    double  lr, beta1, beta2, eps;
    int64_t step;                               /* Adam step of THIS iteration, counts from 1 */
    const float* ones4; int64_t n_ones;         /* [n_ones, 4] ones, n_ones >= the largest block's n_src */
} mi_pinsage_model;
typedef struct mi_pinsage_step_block {
    int64_t n_src, n_dst, nnz;
    const int64_t* src_ids;                                      /* [n_src] global item ids, destination nodes first */
    const int32_t *dst_rowptr, *dst_col; const float* dst_val;   /* CSR by destination [n_dst x n_src], w / max(sum w, 1) */
    const int32_t *src_rowptr, *src_col; const float* src_val;   /* the same entries by source [n_src x n_dst] */
} mi_pinsage_step_block;
typedef struct mi_pinsage_step_batch {
    int32_t n_blocks, reserved;
    mi_pinsage_step_block blocks[MI_PINSAGE_MAX_LAYERS];         /* model order */
    int64_t n_seeds, n_pairs;
    const int64_t *seeds, *pos_u, *pos_v, *neg_v;                /* [n_seeds] ids; [n_pairs] positions in seeds */
    uint64_t seed, step;                                         /* dropout stream */
    float*   loss;                                               /* device float[1] */
    /* data-parallel callers (both or neither; requires model->apply_adam = 0): the projector / bias gradients of THIS batch
     * are written compactly — rows_out [blocks[0].n_src, hidden] (row r belongs to item blocks[0].src_ids[r]), bias_out
     * [n_seeds] (entry s belongs to item seeds[s]) — instead of into the dense buffers, which are not touched. */
    float*   rows_out;
    float*   bias_out;
} mi_pinsage_step_batch;
/* One list of compact gradient rows (a rank's contribution), as written by mi_pinsage_step_f32 through rows_out / bias_out. */
typedef struct mi_pinsage_grad_list {
    int64_t n_rows, n_seeds;
    const int64_t* ids;        /* [n_rows] distinct item ids; the first n_seeds are the seeds */
    const float*   rows;       /* [n_rows, hidden] */
    const float*   bias;       /* [n_seeds] */
} mi_pinsage_grad_list;
int64_t mi_pinsage_step_sizeof(int32_t which);  /* sizeof of: 0 model, 1 batch, 2 conv, 3 block, 4 grad list (binding self-check) */
size_t mi_pinsage_step_workspace_bytes(const mi_pinsage_model* model, const mi_pinsage_step_batch* batch);
int    mi_pinsage_step_f32(const mi_pinsage_model* model, const mi_pinsage_step_batch* batch, void* ws, size_t ws_bytes,
                           mi_stream_t stream);
/* The validation pass of mi_pinsage_step_f32 alone (see mi_ranker_step_check): nothing enqueued, no device memory read. */
int    mi_pinsage_step_check(const mi_pinsage_model* model, const mi_pinsage_step_batch* batch, void* ws, size_t ws_bytes);
/* The update of a data-parallel iteration: the lists of every rank (exchanged by the caller: an all-gather of a few hundred
 * KB instead of an all-reduce of the dense 27 MB table gradient), in rank order, are added into the zero-kept dense
 * buffers — projector rows times grad_scale (1 / world), bias entries as they are; lists are applied one after the other,
 * ids within a list are distinct: one writer per entry at any time, the sum does not depend on scheduling — then Adam over
 * model->params with every gradient times grad_scale (the caller has all-reduced(sum) those dense gradients), the dense
 * Adam over the projector table, and the touched rows / entries are cleared again.  n_lists <= 64. */
int    mi_pinsage_apply_f32(const mi_pinsage_model* model, const mi_pinsage_grad_list* lists, int32_t n_lists, float grad_scale,
                            mi_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LAPLACE_HIP_H */

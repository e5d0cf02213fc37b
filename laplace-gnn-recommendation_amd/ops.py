"""Tensor-level wrappers over the C ABI (include/laplace_hip.h).

PyTorch is plumbing here: device memory, the current HIP stream, dtype/shape checks.  Every
function requires GPU tensors and raises otherwise — there is no CPU path in the product.
"""
from __future__ import annotations

import ctypes
import os
from os import environ as _os_environ
from dataclasses import dataclass, field
from typing import Optional, Tuple

import torch as t
from torch import Tensor

from . import _lib
from ._lib import SpmmExStruct, SpmmPlanStruct, SpmmSweepStruct, check

DEFAULT_CHUNK = 256  # max nnz per work item of a split (hub) row
# columns per band of a banded work-item plan (the adjacencies the SWEEP form does not take: C4's 8 M x 100 K graph on one GPU).
# The work items of a band run together on one XCD and meet in its L2.  A/B on C4 at N = 1, round 3 (tools/ab_c4_bandwidth.sh,
# profiles/r03_c4_band_width.txt; ms per step / dense launch / sparse-operand launch): 2 048: 73.5 / 12.38 / 9.90, 4 096: 65.3 /
# 11.00 / 8.50, 8 192 (rounds 1-2: "the 4 MB L2"): 58.3 / 9.98 / 7.27, 16 384: 54.3 / 9.39 / 6.39, 32 768: 56.2 / 9.98 / 6.24,
# 65 536: 58.6 / 10.62 / 6.16, 131 072: 60.4 / 11.06 / 6.29 — fewer, longer work items (half the partial rows) outweigh the
# band outgrowing one L2 up to 8 MB of rows.  Widening only the THIN rows' bands is slower (csrc/spmm.hip, plan_seg_flags_kernel).
DEFAULT_BAND = int(_os_environ.get("LAPLACE_SPMM_BAND", 16384))
MIN_BANDED_COLS = 131072  # narrower adjacencies keep the row-major plan (nothing to block for)

# bench.py sets this to a list to collect (start, end, kind) HIP events around every propagate launch,
# recorded on the stream the kernels are launched on.  None = no timing overhead.
SPMM_EVENTS = None

# The short-row kernel of a planned product runs on a side stream beside the split-row kernels (mi_spmm_ex.parts): disjoint
# output rows, the same kernels, bitwise the single-stream result (tests/test_gpu_lightgcn.py).  Round 1, against the
# work-item form of the split rows: 1 % (step 5.855 -> 5.782 ms), left off.  Round 3, against the SWEEP form — one 1 024-thread
# workgroup per CU that is bound by its L2 -> CU gathers (profiles/r03_sweep.md) and leaves half of every CU's wavefront
# slots and the fabric to the short rows: dense launch 0.961 -> 0.902 ms, step 5.52 -> 5.34 ms on C2 (same box, bench.py
# LAPLACE_SPMM_TWO_STREAMS=0 / 1; the enqueue order of the two halves does not matter).  On by default;
# LAPLACE_SPMM_TWO_STREAMS=0 switches it off.  (2 = split rows enqueued first: an A/B setting.)
SPMM_TWO_STREAMS = int(os.environ.get("LAPLACE_SPMM_TWO_STREAMS", "1") or 0)
PACK_ENTRIES = os.environ.get("LAPLACE_SPMM_PACK", "1") != "0"   # A/B: banded plans carry launch-ordered copies of their entries
X_RARE_BITS = os.environ.get("LAPLACE_X_RARE", "1") != "0"   # A/B: spmm(..., x_rare=True) honoured (mi_spmm_ex.x_bits)
_SIDE = {}


def _side_stream(device) -> "t.cuda.Stream":
    key = t.device(device).index
    if key not in _SIDE:
        _SIDE[key] = t.cuda.Stream(device=device)
    return _SIDE[key]

# Adjacencies smaller than this run without a split-row plan (one wavefront per row whatever its
# length): building a plan reads two counters back to the host, which per-batch subgraphs of the
# ranker cannot afford and do not need.
PLAN_MIN_NNZ = 1 << 20


_stream = _lib.current_stream


def _ptr(x: Optional[Tensor]) -> Optional[int]:
    if x is None:
        return None
    return x.data_ptr() if x.numel() > 0 else None


def _need(x: Tensor, dtype: t.dtype, name: str, contiguous: bool = True) -> None:
    try:  # the valid case in three attribute reads; the messages below are for everything else
        if x.dtype is dtype and x.is_cuda and (not contiguous or x.is_contiguous()):
            return
    except AttributeError:
        pass
    if not isinstance(x, Tensor):
        raise TypeError(f"{name}: expected a tensor, got {type(x)}")
    if not x.is_cuda:
        raise _lib.MiError(f"{name}: tensor is on {x.device}; the HIP path needs a GPU tensor (no CPU fallback)")
    if x.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {x.dtype}")
    if contiguous and not x.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")


def _rows_ok(x: Tensor, name: str) -> int:
    """Checks a 2-D fp32 row-major matrix with unit inner stride; returns its leading dimension."""
    try:  # the valid case first and in one frame: this runs ~130 times per ranker iteration
        if x.dtype is t.float32 and x.is_cuda and x.dim() == 2:
            (r, c), (s0, s1) = x.shape, x.stride()
            if c <= 1 or s1 == 1:
                return s0 if r > 1 else max(c, s0)
    except AttributeError:
        pass
    _need(x, t.float32, name, contiguous=False)
    if x.dim() != 2 or (x.shape[1] > 1 and x.stride(1) != 1):
        raise ValueError(f"{name}: expected a row-major 2-D matrix, got shape {tuple(x.shape)} strides {x.stride()}")
    return x.stride(0) if x.shape[0] > 1 else max(x.shape[1], x.stride(0))


def _ws(nbytes: int, device) -> Tensor:
    return t.empty(max(int(nbytes), 256), dtype=t.uint8, device=device)


_WS_CACHE = {}


def _ws_reused(nbytes: int, device) -> Tensor:
    """Scratch for launches that consume it before they return control to the stream's next launch (the GEMM split-K
    partials): one buffer per (device, stream), grown on demand, instead of an allocation per call — the ranker issues
    ~10 such launches per iteration.  Launches on one stream run in order, so the next user cannot overtake the last."""
    dev = t.device(device) if not isinstance(device, int) else t.device("cuda", device)
    key = (dev.index if dev.index is not None else t.cuda.current_device(), _stream())
    cur = _WS_CACHE.get(key)
    if cur is None or cur.numel() < nbytes:
        cur = t.empty(max(int(nbytes) * 2, 1 << 20), dtype=t.uint8, device=dev)
        _WS_CACHE[key] = cur
    return cur


# --------------------------------------------------------------------------------------
# CSR container
# --------------------------------------------------------------------------------------
@dataclass
class SpmmPlan:
    struct: SpmmPlanStruct
    long_rows: Optional[Tensor]
    item_ptr: Optional[Tensor]
    items: Optional[Tensor]
    long_index: Optional[Tensor] = None
    partial: dict = field(default_factory=dict)  # d -> workspace tensor
    sweep: Optional[SpmmSweepStruct] = None       # split rows in SWEEP form (mi_spmm_sweep); tensors kept in sweep_t
    sweep_t: tuple = ()
    wide: Optional["SpmmPlan"] = None             # work-item plan of the same adjacency for widths the sweep form lacks
    packed: tuple = ()                            # (epos, ecol, eval): the split rows' entries in launch order (mi_spmm_plan_pack_entries)
    packed_of: tuple = ()                         # (val.data_ptr(), val._version) the packed values were copied from
    nnz_long: int = 0

    @property
    def n_long_rows(self) -> int:
        return int(self.struct.n_long_rows)

    @property
    def n_items(self) -> int:
        return int(self.struct.n_items)


@dataclass
class DeviceCSR:
    """Sorted CSR on the GPU: int32 rowptr/col, optional fp32 val, optional perm to the input order."""
    n_rows: int
    n_cols: int
    rowptr: Tensor
    col: Tensor
    val: Optional[Tensor] = None
    perm: Optional[Tensor] = None
    plan: Optional[SpmmPlan] = None
    hot: Optional[Tuple[int, int]] = None   # (first row, rows) of X worth an LDS cache in dense launches: see spmm()

    @property
    def nnz(self) -> int:
        return int(self.col.numel())

    @property
    def device(self):
        return self.rowptr.device


def coo_to_csr(row: Tensor, col: Tensor, n_rows: int, n_cols: int, want_perm: bool = True) -> DeviceCSR:
    """K4 — replaces SparseTensor(row=, col=, sparse_sizes=) (data/lightgcn_loader.py:65-79)."""
    _need(row, t.int64, "row")
    _need(col, t.int64, "col")
    if row.shape != col.shape or row.dim() != 1:
        raise ValueError("row/col must be 1-D tensors of equal length")
    nnz = row.numel()
    dev = row.device
    rowptr = t.empty(n_rows + 1, dtype=t.int32, device=dev)
    col_out = t.empty(nnz, dtype=t.int32, device=dev)
    perm = t.empty(nnz, dtype=t.int32, device=dev) if want_perm else None
    L = _lib.lib()
    ws = _ws(L.mi_coo_to_csr_workspace_bytes(n_rows, nnz), dev)
    check(L.mi_coo_to_csr_i32(n_rows, n_cols, nnz, _ptr(row), _ptr(col), _ptr(rowptr), _ptr(col_out),
                              _ptr(perm), ws.data_ptr(), ws.numel(), _stream()), "mi_coo_to_csr_i32")
    return DeviceCSR(n_rows, n_cols, rowptr, col_out, None, perm)


def csr_transpose(a: DeviceCSR) -> DeviceCSR:
    """CSR(A) -> CSR(A^T); values follow through perm_t (replaces csr2csc of torch_sparse)."""
    dev = a.device
    rowptr_t = t.empty(a.n_cols + 1, dtype=t.int32, device=dev)
    col_t = t.empty(a.nnz, dtype=t.int32, device=dev)
    perm_t = t.empty(a.nnz, dtype=t.int32, device=dev)
    L = _lib.lib()
    ws = _ws(L.mi_csr_transpose_workspace_bytes(a.n_rows, a.nnz), dev)
    check(L.mi_csr_transpose_i32(a.n_rows, a.n_cols, a.nnz, _ptr(a.rowptr), _ptr(a.col), _ptr(rowptr_t),
                                 _ptr(col_t), _ptr(perm_t), ws.data_ptr(), ws.numel(), _stream()),
          "mi_csr_transpose_i32")
    val_t = gather_f32(a.val, perm_t) if a.val is not None else None
    return DeviceCSR(a.n_cols, a.n_rows, rowptr_t, col_t, val_t, perm_t)


def gather_f32(src: Tensor, idx: Tensor) -> Tensor:
    _need(src, t.float32, "src")
    _need(idx, t.int32, "idx")
    out = t.empty(idx.numel(), dtype=t.float32, device=src.device)
    check(_lib.lib().mi_gather_f32(idx.numel(), _ptr(src), _ptr(idx), _ptr(out), _stream()), "mi_gather_f32")
    return out


def gcn_norm(a: DeviceCSR, val_in: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """K3 — gcn_norm(adj, add_self_loops=False) (model/lightgcn.py:56). Returns (val, deg^-1/2)."""
    if a.n_rows != a.n_cols:
        raise ValueError("gcn_norm needs a square adjacency")
    if val_in is not None:
        _need(val_in, t.float32, "val_in")
    val = t.empty(a.nnz, dtype=t.float32, device=a.device)
    dis = t.empty(a.n_rows, dtype=t.float32, device=a.device)
    check(_lib.lib().mi_gcn_norm_csr_f32(a.n_rows, a.nnz, _ptr(a.rowptr), _ptr(a.col), _ptr(val_in),
                                         _ptr(val), _ptr(dis), _stream()), "mi_gcn_norm_csr_f32")
    return val, dis


def scale_csr(a: DeviceCSR, val_in: Optional[Tensor] = None, row_scale: Optional[Tensor] = None,
              col_scale: Optional[Tensor] = None) -> Tensor:
    """val[p] = (val_in[p] * row_scale[row]) * col_scale[col]; None = ones."""
    for n, x in (("val_in", val_in), ("row_scale", row_scale), ("col_scale", col_scale)):
        if x is not None:
            _need(x, t.float32, n)
    if row_scale is not None and row_scale.numel() != a.n_rows:
        raise ValueError("row_scale must have one entry per row")
    if col_scale is not None and col_scale.numel() != a.n_cols:
        raise ValueError("col_scale must have one entry per column")
    out = t.empty(a.nnz, dtype=t.float32, device=a.device)
    check(_lib.lib().mi_scale_csr_f32(a.n_rows, a.nnz, _ptr(a.rowptr), _ptr(a.col), _ptr(val_in), _ptr(row_scale),
                                      _ptr(col_scale), _ptr(out), _stream()), "mi_scale_csr_f32")
    return out


def row_slice(a: DeviceCSR, r0: int, r1: int) -> DeviceCSR:
    """Rows [r0, r1) of `a` as a CSR that shares col/val storage (rowptr keeps absolute offsets)."""
    return DeviceCSR(r1 - r0, a.n_cols, a.rowptr[r0:r1 + 1], a.col, a.val, None, None, a.hot)


# LDS hot-row cache of the dense short-row launches (include/laplace_hip.h, mi_spmm_ex.hot_rows): module switches for A/B
# Persistent, software-pipelined short-row launch (+ optional LDS hot-row cache).  OFF by default: measured on C2 (round 3,
# profiles/r03_spmm_rows_persistent.md) it is 5-15 % SLOWER than the plain launch — the propagate runs at the chip's
# random-row gather ceiling for the bytes that miss L2 (5.0-5.5 TB/s of fabric traffic against the 5.5-5.8 TB/s the
# guide measures for random whole rows), rows the LDS cache serves were L2 hits already, and fewer resident wavefronts
# (86-105 VGPRs against 78) cost more than the shorter dependency chain returns.  Kept as a bitwise-equal alternative
# (tests/test_gpu_lightgcn.py) for graphs / parts whose balance differs.
PERSISTENT_ROWS = _os_environ.get("LAPLACE_PERSISTENT_ROWS", "0") == "1"
HOT_ROWS = int(_os_environ.get("LAPLACE_HOT_ROWS", 304))        # rows offered to the cache (the kernel clips to its LDS share); 0 = off
HOT_THREADS = int(_os_environ.get("LAPLACE_HOT_THREADS", 0))    # workgroup size of the persistent launch; 0 = the library's default


def hot_item_rows(num_users: int, num_items: int) -> Optional[Tuple[int, int]]:
    """The `hot` hint of a [users; items] adjacency whose items are numbered by popularity (interactions.LocalityOrder):
    the first item rows are the ones most gathers go to."""
    if HOT_ROWS <= 0 or num_items <= 0:
        return None
    return (int(num_users), int(min(HOT_ROWS, num_items)))


SWEEP_BAND = 2048     # columns per band of a sweep plan: 2048 rows x 128 floats = 1 MB, a quarter of an XCD's L2
SWEEP_STREAMS = 1024  # sub-groups per XCD: 32 CUs x one workgroup of 32 sub-groups
SWEEP_MIN_BANDS = 64  # narrower adjacencies have nothing to sweep
SWEEP_PACE = 0        # pacing of the sweep: ticks (10 ns) per band of the time table; 0 = none


def build_sweep_plan(a: DeviceCSR, chunk: int = DEFAULT_CHUNK, band: int = SWEEP_BAND,
                     n_streams: int = SWEEP_STREAMS, min_deg: Optional[int] = None) -> Optional[SpmmPlan]:
    """SWEEP-form split-row plan (include/laplace_hip.h, mi_spmm_sweep), or None when the adjacency does not qualify
    (no long rows, fewer than SWEEP_MIN_BANDS bands, more row parts than the 8 * n_streams accumulators of an XCD,
    columns >= 2^28, no values yet).  Set-up: a few device sorts and one host pass over the long rows' degrees.
    min_deg (default: chunk): only the rows with MORE entries than this are laid out — the hub half of a hybrid plan
    (build_hybrid_plan), whose remaining split rows are banded work items."""
    import heapq
    if a.val is None or a.nnz == 0 or a.n_cols >= (1 << 27) or a.n_cols < SWEEP_MIN_BANDS * band:
        return None
    dev = a.device
    deg = (a.rowptr[1:] - a.rowptr[:-1]).to(t.int64)

    def slots_for(ch: int):
        rows = t.nonzero(deg > ch).view(-1)
        if rows.numel() == 0:
            return rows, [], 0, [], 0
        host = deg[rows].cpu().tolist()
        tot = sum(host)
        # parts per row: a slot should carry about a third of a stream's share, so that streams can be balanced
        target = max(ch, tot // (3 * n_streams))
        pp = [max(1, -(-d // target)) for d in host]
        return rows, host, tot, pp, sum(pp)

    long_rows, dl_host, total, parts, n_slots = slots_for(chunk if min_deg is None else max(int(min_deg), chunk))
    if long_rows.numel() == 0:
        return None
    if n_slots > 8 * n_streams:
        # More long rows than an XCD has accumulators (a graph whose typical row is long, e.g. C4's item rows).  Raising the
        # threshold until the longest rows fit was measured on C4 (round 2): dense launch 10.1 -> 10.7 ms against the banded
        # work items, so such adjacencies keep those.
        return None
    n_long = int(long_rows.numel())
    dl = deg[long_rows]
    # longest-processing-time assignment of slots to streams (<= 8 per stream)
    loads = []
    slot_base = [0] * (n_long + 1)
    for i, (d, p) in enumerate(zip(dl_host, parts)):
        slot_base[i + 1] = slot_base[i] + p
        loads += [(d / p, slot_base[i] + j) for j in range(p)]
    loads.sort(key=lambda z: (-z[0], z[1]))
    heap = [(0.0, k, 0) for k in range(n_streams)]
    heapq.heapify(heap)
    owner = [0] * n_slots
    qidx = [0] * n_slots
    slot_of = [-1] * (8 * n_streams)
    full = []
    for w, sid in loads:
        while True:
            load, k, cnt = heapq.heappop(heap)
            if cnt < 8:
                break
            full.append((load, k, cnt))
        owner[sid], qidx[sid] = k, cnt
        slot_of[k * 8 + cnt] = sid
        heapq.heappush(heap, (load + w, k, cnt + 1))
    owner_t = t.tensor(owner, dtype=t.int64, device=dev)
    q_t = t.tensor(qidx, dtype=t.int64, device=dev)
    parts_t = t.tensor(parts, dtype=t.int64, device=dev)
    base_t = t.tensor(slot_base, dtype=t.int64, device=dev)
    # entries of the long rows, in CSR order
    starts = a.rowptr[long_rows].to(t.int64)
    row_i = t.repeat_interleave(t.arange(n_long, device=dev), dl)                       # long-row index per entry
    first = t.cumsum(dl, 0) - dl
    e = starts[row_i] + (t.arange(total, device=dev) - first[row_i])                    # position in a.col / a.val
    c = a.col[e].to(t.int64)
    b = c // band
    n_b = (a.n_cols + band - 1) // band
    grp = row_i * n_b + b                                                               # (row, band): contiguous runs
    _, counts = t.unique_consecutive(grp, return_counts=True)
    g_first = t.cumsum(counts, 0) - counts
    g_of = t.repeat_interleave(t.arange(counts.numel(), device=dev), counts)
    rank = t.arange(total, device=dev) - g_first[g_of]
    part = (rank * parts_t[row_i]) // counts[g_of]                                      # equal consecutive pieces
    slot = base_t[row_i] + part
    x, tb = b & 7, b >> 3
    n_t = (n_b + 7) // 8
    key = (((x * n_streams + owner_t[slot]) * n_t + tb) << 3) | q_t[slot]
    order = t.argsort(key, stable=True)                                                 # columns stay ascending in a run
    skey = key[order]
    stream_key = (skey >> 3) // n_t
    band_key = skey >> 3                                                                # (stream, band)
    new_band = t.ones_like(band_key, dtype=t.bool)
    new_band[1:] = band_key[1:] != band_key[:-1]                                        # first entry of a band in its stream
    col_s = (((q_t[slot] << 28) | c)[order] | (new_band.to(t.int64) << 27)).to(t.int32).contiguous()
    val_s = a.val[e][order].contiguous()
    stream_ptr = t.searchsorted(stream_key, t.arange(8 * n_streams + 1, device=dev)).to(t.int32).contiguous()
    slot_of_t = t.tensor(slot_of, dtype=t.int32, device=dev)
    long_rows32 = long_rows.to(t.int32).contiguous()
    item_ptr = (8 * base_t).to(t.int32).contiguous()
    long_index = t.full((max(a.n_rows, 1),), -1, dtype=t.int32, device=dev)
    long_index[long_rows] = t.arange(n_long, dtype=t.int32, device=dev)
    st = SpmmPlanStruct()
    st.chunk, st.n_long_rows, st.n_items, st.n_launch = chunk, n_long, 8 * n_slots, 8 * n_slots
    st.long_rows, st.item_ptr, st.items = long_rows32.data_ptr(), item_ptr.data_ptr(), None
    st.long_index = long_index.data_ptr()
    st.band, st.n_bands = band, n_b
    # `epoch` carries the bands per XCD: read only by the MI_SWEEP_WG_SYNC experiment build of the kernel (profiles/r03_sweep.md)
    sw = SpmmSweepStruct(col_s.data_ptr(), val_s.data_ptr(), stream_ptr.data_ptr(), slot_of_t.data_ptr(), n_streams, n_slots,
                         None, int(n_t), max(SWEEP_PACE, 0))
    return SpmmPlan(st, long_rows32, item_ptr, None, long_index, sweep=sw, sweep_t=(col_s, val_s, stream_ptr, slot_of_t))


def sweep_degree_threshold(a: DeviceCSR, chunk: int, n_streams: int = SWEEP_STREAMS, fill: float = 1.0) -> Optional[int]:
    """The smallest degree T such that the rows with MORE than T entries fit the sweep form's 8 * n_streams row-part
    accumulators (build_sweep_plan's own part count: ceil(deg / max(chunk, total / (3 n_streams))) per row); None when even
    the single longest row does not, or no row is longer than chunk.  One host pass over the long rows' degrees."""
    import numpy as np
    deg = (a.rowptr[1:] - a.rowptr[:-1])
    dl = deg[deg > chunk]
    if dl.numel() == 0:
        return None
    ds = np.sort(dl.cpu().numpy().astype(np.int64))[::-1]           # descending
    cum = np.cumsum(ds)
    cap = int(8 * n_streams * fill)

    def slots(h):   # row parts of the h longest rows
        target = max(chunk, int(cum[h - 1]) // (3 * n_streams))
        return int(np.sum(-(-ds[:h] // target)))
    if slots(1) > cap:
        return None
    lo, hi = 1, len(ds)                                                # largest h with slots(h) <= cap (monotone in practice: checked)
    while lo < hi:
        mid = (lo + hi + 1) // 2
        if slots(mid) <= cap:
            lo = mid
        else:
            hi = mid - 1
    h = lo
    if h == len(ds):
        return int(chunk)
    T = int(ds[h])                                                     # rows with deg > T: a prefix of the h longest (ties cut)
    while True:
        hh = int(np.searchsorted(-ds, -T, side="left"))                # number of rows with deg > T
        if hh == 0:
            return None
        if slots(hh) <= cap:
            return T
        T = int(ds[hh - 1])                                            # not monotone here: drop the shortest kept degree


def build_hybrid_plan(a: DeviceCSR, chunk: int = DEFAULT_CHUNK, band: Optional[int] = None, tail_whole: bool = False,
                      sweep_band: Optional[int] = None, n_streams: int = SWEEP_STREAMS) -> Optional[SpmmPlan]:
    """HYBRID split-row plan (round 4) for adjacencies whose typical row is long (C4's item rows: 100 K rows of ~10^3 entries):
    the hub rows — as many of the longest as the sweep form's 8 192 row-part accumulators hold, ~80 % of the entries on a
    Zipf graph — in SWEEP form (every row of X fetched into ONE XCD's L2 about once, 8 partial rows per row part), the
    remaining split rows either as banded work items (tail_whole=False) or, with the split threshold raised to the hub
    rows' degree, whole in the short-row kernel (tail_whole=True).  None when the adjacency does not qualify."""
    T = sweep_degree_threshold(a, chunk, n_streams)
    if T is None:
        return None
    sb = SWEEP_BAND if sweep_band is None else sweep_band
    if tail_whole or T <= chunk:
        return build_sweep_plan(a, chunk=max(T, chunk), band=sb, n_streams=n_streams)
    sw = build_sweep_plan(a, chunk=chunk, band=sb, n_streams=n_streams, min_deg=T)
    if sw is None:
        return None
    # the banded half: rows with chunk < degree <= T
    L = _lib.lib()
    if band is None:
        band = HYBRID_TAIL_BAND if a.n_cols >= MIN_BANDED_COLS else 0
    dev = a.device
    ws = _ws(L.mi_spmm_plan_workspace_bytes(a.n_rows, a.nnz), dev)
    info = _lib.SpmmPlanInfo()
    check(L.mi_spmm_plan_count_range(a.n_rows, a.n_cols, _ptr(a.rowptr), _ptr(a.col), chunk, T, band, ws.data_ptr(), ws.numel(),
                                     ctypes.byref(info), _stream()), "mi_spmm_plan_count_range")
    nl, nlaunch = int(info.n_long_rows), int(info.n_launch)
    if nl == 0:
        return sw
    long_rows_b = t.empty(nl, dtype=t.int32, device=dev)
    item_ptr_b = t.empty(nl + 1, dtype=t.int32, device=dev)
    items = t.empty(4 * max(nlaunch, 1), dtype=t.int32, device=dev)
    st_b = SpmmPlanStruct()
    st_b.long_rows, st_b.item_ptr, st_b.items, st_b.long_index = long_rows_b.data_ptr(), item_ptr_b.data_ptr(), items.data_ptr(), None
    check(L.mi_spmm_plan_fill(a.n_rows, _ptr(a.rowptr), ctypes.byref(info), ctypes.byref(st_b), ws.data_ptr(), ws.numel(),
                              _stream()), "mi_spmm_plan_fill")
    off = int(sw.struct.n_items)                                     # 8 * n_slots partial rows of the sweep come first
    it = items.view(-1, 4)
    it[:, 3] += (it[:, 3] >= 0).to(t.int32) * off                    # padding slots stay negative
    long_rows = t.cat([sw.long_rows, long_rows_b]).contiguous()
    item_ptr = t.cat([sw.item_ptr[:-1], item_ptr_b + off]).contiguous()
    long_index = t.full((max(a.n_rows, 1),), -1, dtype=t.int32, device=dev)
    long_index[long_rows.long()] = t.arange(long_rows.numel(), dtype=t.int32, device=dev)
    st = SpmmPlanStruct()
    st.chunk, st.n_long_rows = chunk, int(long_rows.numel())
    st.n_items, st.n_launch = off + int(info.n_items), nlaunch
    st.long_rows, st.item_ptr, st.items, st.long_index = long_rows.data_ptr(), item_ptr.data_ptr(), items.data_ptr(), long_index.data_ptr()
    st.band, st.n_bands = int(info.band), int(info.n_bands)
    return SpmmPlan(st, long_rows, item_ptr, items, long_index, sweep=sw.sweep, sweep_t=sw.sweep_t)


HYBRID_TAIL_BAND = int(_os_environ.get("LAPLACE_HYBRID_TAIL_BAND", 16384))  # columns per band of a hybrid plan's banded tail


import os as _os
SWEEP = _os.environ.get("LAPLACE_SWEEP", "1") != "0"  # module switch (A/B): off keeps the banded work-item plans
SWEEP_BAND = int(_os.environ.get("LAPLACE_SWEEP_BAND", SWEEP_BAND))
SWEEP_PACE = int(_os.environ.get("LAPLACE_SWEEP_PACE", SWEEP_PACE))


def build_spmm_plan(a: DeviceCSR, chunk: int = DEFAULT_CHUNK, band: Optional[int] = None, sweep: Optional[bool] = None) -> SpmmPlan:
    """Split-row plan of an adjacency.  band = columns per band of a banded plan (see include/laplace_hip.h),
    0 = row-major, None = DEFAULT_BAND when the adjacency has at least MIN_BANDED_COLS columns.  sweep (default: the
    module switch SWEEP, and only when band is not given): the SWEEP form when the adjacency qualifies."""
    if (SWEEP if sweep is None else sweep) and band is None:
        plan = build_sweep_plan(a, chunk, band=SWEEP_BAND)
        if plan is not None:
            return plan
    L = _lib.lib()
    if band is None:
        band = DEFAULT_BAND if a.n_cols >= MIN_BANDED_COLS else 0
    dev = a.device
    ws = _ws(L.mi_spmm_plan_workspace_bytes(a.n_rows, a.nnz), dev)
    info = _lib.SpmmPlanInfo()
    check(L.mi_spmm_plan_count(a.n_rows, a.n_cols, _ptr(a.rowptr), _ptr(a.col), chunk, band, ws.data_ptr(), ws.numel(),
                               ctypes.byref(info), _stream()), "mi_spmm_plan_count")
    nl, nlaunch = int(info.n_long_rows), int(info.n_launch)
    long_rows = t.empty(max(nl, 1), dtype=t.int32, device=dev)
    item_ptr = t.empty(max(nl, 1) + 1, dtype=t.int32, device=dev)
    items = t.empty(4 * max(nlaunch, 1), dtype=t.int32, device=dev)
    long_index = t.empty(max(a.n_rows, 1), dtype=t.int32, device=dev)
    st = SpmmPlanStruct()
    st.long_rows, st.item_ptr, st.items = long_rows.data_ptr(), item_ptr.data_ptr(), items.data_ptr()
    st.long_index = long_index.data_ptr()
    check(L.mi_spmm_plan_fill(a.n_rows, _ptr(a.rowptr), ctypes.byref(info), ctypes.byref(st), ws.data_ptr(), ws.numel(),
                              _stream()), "mi_spmm_plan_fill")
    plan = SpmmPlan(st, long_rows, item_ptr, items, long_index)
    plan.nnz_long = int(info.nnz_long)
    del ws
    if PACK_ENTRIES and nlaunch > 0 and plan.nnz_long > 0 and a.val is not None:
        _pack_plan_entries(a, plan, values_only=False)
    return plan


def _pack_plan_entries(a: DeviceCSR, plan: SpmmPlan, values_only: bool) -> None:
    """The split rows' (col, val) copied into launch order (mi_spmm_plan.epos / ecol / eval): a banded plan's work items are
    launched band by band, so in the CSR arrays every work item's ~12 entries sit in two cache lines of their own."""
    L = _lib.lib()
    dev, st = a.device, plan.struct
    if not values_only:
        plan.packed = (t.empty(int(st.n_launch), dtype=t.int32, device=dev), t.empty(plan.nnz_long, dtype=t.int32, device=dev),
                       t.empty(plan.nnz_long, dtype=t.float32, device=dev))
        st.epos, st.ecol, st.eval = (x.data_ptr() for x in plan.packed)
    ws = _ws(L.mi_spmm_plan_pack_workspace_bytes(int(st.n_launch)), dev)
    check(L.mi_spmm_plan_pack_entries(ctypes.byref(st), plan.nnz_long, _ptr(a.col), _ptr(a.val), 1 if values_only else 0,
                                      ws.data_ptr(), ws.numel(), _stream()), "mi_spmm_plan_pack_entries")
    plan.packed_of = (a.val.data_ptr(), a.val._version)


def spmm(a: DeviceCSR, X: Tensor, *, Y: Optional[Tensor] = None, addend: Optional[Tensor] = None,
         S: Optional[Tensor] = None, scale: float = 1.0, x_map: Optional[Tensor] = None,
         addend_map: Optional[Tensor] = None, row_list: Optional[Tensor] = None,
         n_list_dev: Optional[Tensor] = None, adam: Optional[dict] = None, x_rare: bool = False) -> None:
    """K1/K2 — acc = A @ X; Y = acc (optional); S = scale * (addend + acc) (optional).

    adam = dict(p=, m=, v=, step=, lr=, beta1=, beta2=, eps=, reg_w=): S's value is the gradient of parameter
    table p and is consumed in registers by the Adam update (adam_step's arithmetic); S itself may be None.

    a.hot = (first row, rows): a speed hint for dense launches (no x_map / row_list) of a planned adjacency — those rows
    of X are cached in LDS by a persistent short-row launch.  Same values, same summation order: the result is bitwise
    the one without the hint.

    Sparse-operand forms (mi_spmm_csr_ex_f32): x_map int32[n_cols] — X is compact, column c reads
    X[x_map[c]], negative = an all-zero row; addend_map int32[n_rows] — addend is compact; row_list
    int32[n] (+ n_list_dev, device int32[1]) — compute only these rows, Y/S/addend compact by list position.
    x_rare (with x_map): few columns are live (mi_spmm_ex.x_bits) — the split rows' work items test a bitmap of the map
    first, gather only live entries and skip work items without any; bitwise the result without the hint."""
    if a.val is None:
        raise ValueError("spmm needs edge values (run gcn_norm or set val)")
    d = X.shape[1]
    ldx = _rows_ok(X, "X")
    for nm, mp, ln in (("x_map", x_map, a.n_cols), ("addend_map", addend_map, a.n_rows)):
        if mp is not None:
            _need(mp, t.int32, nm)
            if mp.numel() != ln:
                raise ValueError(f"{nm} must have {ln} entries")
    if row_list is not None:
        _need(row_list, t.int32, "row_list")
        if addend_map is not None:
            raise ValueError("row_list and addend_map are exclusive (addend is indexed by list position)")
    if n_list_dev is not None:
        _need(n_list_dev, t.int32, "n_list_dev")
    if x_map is None and X.shape[0] != a.n_cols:
        raise ValueError(f"X has {X.shape[0]} rows, adjacency has {a.n_cols} columns")
    n_out = row_list.numel() if row_list is not None else a.n_rows
    for name, m in (("Y", Y), ("S", S)) + ((("addend", addend),) if addend_map is None else ()):
        if m is not None and (m.shape[0] != n_out or m.shape[1] != d):
            raise ValueError(f"{name} must be [{n_out}, {d}], got {tuple(m.shape)}")
    if addend is not None and addend.shape[1] != d:
        raise ValueError("addend width differs from X")
    if Y is None and S is None and adam is None:
        raise ValueError("spmm needs an output (Y and/or S)")
    adam_args = None
    if adam is not None:
        if row_list is not None:
            raise ValueError("the optimizer epilogue updates every parameter row: not with row_list")
        p_, m_, v_ = adam["p"], adam["m"], adam["v"]
        ldp = _rows_ok(p_, "adam.p")
        _need(m_, t.float32, "adam.m")
        _need(v_, t.float32, "adam.v")
        if tuple(p_.shape) != (a.n_rows, d) or m_.shape != p_.shape or v_.shape != p_.shape:
            raise ValueError(f"adam.p/m/v must be [{a.n_rows}, {d}]")
        rw = adam.get("reg_w")
        if rw is not None:
            _need(rw, t.float32, "adam.reg_w")
        adam_args = _lib.AdamArgs(p_.data_ptr(), ldp, m_.data_ptr(), v_.data_ptr(), _ptr(rw), float(adam["lr"]),
                                  float(adam.get("beta1", 0.9)), float(adam.get("beta2", 0.999)),
                                  float(adam.get("eps", 1e-8)), int(adam["step"]))
    ldy = _rows_ok(Y, "Y") if Y is not None else 0
    lda = _rows_ok(addend, "addend") if addend is not None else 0
    lds = _rows_ok(S, "S") if S is not None else 0
    if a.plan is None and (a.nnz >= PLAN_MIN_NNZ or row_list is not None):
        a.plan = build_spmm_plan(a)
    plan = a.plan
    if plan is not None and plan.sweep is not None and d > 128:  # the sweep form stops at 128 floats per row
        if plan.wide is None:
            plan.wide = build_spmm_plan(a, chunk=int(plan.struct.chunk), sweep=False)
        plan = plan.wide
    L = _lib.lib()
    if plan is not None and plan.packed and plan.packed_of != (a.val.data_ptr(), a.val._version):
        _pack_plan_entries(a, plan, values_only=True)     # the adjacency was re-weighted since the plan copied its values
    ws_ptr, ws_bytes = None, 0
    if plan is not None and plan.n_items > 0:
        if d not in plan.partial:
            plan.partial[d] = _ws(L.mi_spmm_workspace_bytes(ctypes.byref(plan.struct), d), a.device)
        ws_ptr, ws_bytes = plan.partial[d].data_ptr(), plan.partial[d].numel()
    col_ptr = _ptr(a.col) if a.nnz else a.rowptr.data_ptr()
    val_ptr = _ptr(a.val) if a.nnz else a.rowptr.data_ptr()
    ev = None
    if SPMM_EVENTS is not None:
        ev = (t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True))
        ev[0].record()
    sweep = plan.sweep if (plan is not None and plan.sweep is not None) else None
    hot_base, hot_rows = 0, (-1 if PERSISTENT_ROWS else 0)   # mi_spmm_ex.hot_rows: 0 = the plain launch, < 0 = persistent without a cache
    if (a.hot is not None and plan is not None and x_map is None and row_list is None and d <= 256 and HOT_ROWS > 0
            and PERSISTENT_ROWS):
        hot_base, hot_rows = int(a.hot[0]), int(a.hot[1])
        if hot_rows == 0:
            hot_rows = -1       # an empty range: still the persistent form the switch asks for
        if hot_base < 0 or hot_rows < -1 or hot_base + max(hot_rows, 0) > X.shape[0]:
            raise ValueError(f"hot rows [{hot_base}, {hot_base + hot_rows}) lie outside X ({X.shape[0]} rows)")

    x_bits = None
    if (x_rare and X_RARE_BITS and x_map is not None and row_list is None and sweep is None and plan is not None
            and plan.n_items > 0):
        x_bits = getattr(a, "_x_bits", None)
        if x_bits is None:
            x_bits = a._x_bits = t.empty((a.n_cols + 31) // 32, dtype=t.int32, device=a.device)
        check(L.mi_map_live_bits_i32(a.n_cols, _ptr(x_map), _ptr(x_bits), _stream()), "mi_map_live_bits_i32")

    def launch(parts: int, stream: int) -> None:
        exs = None
        if (x_map is not None or addend_map is not None or row_list is not None or adam_args is not None or parts
                or sweep is not None or hot_rows != 0 or HOT_THREADS):
            exs = SpmmExStruct(_ptr(x_map), _ptr(addend_map), _ptr(row_list), _ptr(n_list_dev),
                               row_list.numel() if row_list is not None else 0,
                               ctypes.pointer(adam_args) if adam_args is not None else None, parts, hot_rows,
                               ctypes.pointer(sweep) if sweep is not None else None, hot_base, HOT_THREADS, _ptr(x_bits))
        check(L.mi_spmm_csr_ex_f32(a.n_rows, d, _ptr(a.rowptr), col_ptr, val_ptr, X.data_ptr(), ldx,
                                   _ptr(Y), ldy, _ptr(addend), lda, _ptr(S), lds, float(scale),
                                   ctypes.byref(plan.struct) if plan is not None else None,
                                   ctypes.byref(exs) if exs is not None else None, ws_ptr, ws_bytes,
                                   stream), "mi_spmm_csr_ex_f32")

    if SPMM_TWO_STREAMS and plan is not None and plan.n_items > 0:
        # short rows beside the split rows' work items + fix-up: disjoint output rows, joined before returning
        cur = t.cuda.current_stream()
        side = _side_stream(a.device)
        side.wait_stream(cur)
        if SPMM_TWO_STREAMS == 2:    # the split rows' one-workgroup-per-CU sweep takes its half of every CU first
            launch(_lib.MI_SPMM_SPLIT_ROWS, cur.cuda_stream)
            launch(_lib.MI_SPMM_SHORT_ROWS, side.cuda_stream)
        else:
            launch(_lib.MI_SPMM_SHORT_ROWS, side.cuda_stream)
            launch(_lib.MI_SPMM_SPLIT_ROWS, cur.cuda_stream)
        cur.wait_stream(side)
    else:
        launch(0, _stream())
    if ev is not None:
        ev[1].record()
        kind = "sparse" if (x_map is not None or row_list is not None) else ("dense_adam" if adam is not None else "dense")
        SPMM_EVENTS.append((ev[0], ev[1], kind, None, a.n_rows, a))


def expand_rows(a: DeviceCSR) -> Tensor:
    out = t.empty(a.nnz, dtype=t.int32, device=a.device)
    check(_lib.lib().mi_csr_expand_rows(a.n_rows, _ptr(a.rowptr), _ptr(out), a.nnz, _stream()),
          "mi_csr_expand_rows")
    return out


def sample_bpr_batch(r: DeviceCSR, row_of_edge: Tensor, batch: int, neg_range: int, seed: int, step: int,
                     quirk: bool = False, out: Optional[Tuple[Tensor, Tensor, Tensor]] = None,
                     edges_in_order: bool = False, no_self_loops: bool = False) -> Tuple[Tensor, Tensor, Tensor]:
    """K9 — replaces sample_mini_batch (data/lightgcn_loader.py:95-112) on device.  quirk: the reference's key
    collision; no_self_loops: structured_negative_sampling(contains_neg_self_loops=False) — never negative id == user id."""
    if r.nnz == 0:
        raise ValueError("cannot sample from an empty edge set")
    if neg_range <= 0:
        raise ValueError("neg_range must be positive")
    dev = r.device
    if out is None:
        out = tuple(t.empty(batch, dtype=t.int64, device=dev) for _ in range(3))
    users, pos, neg = out
    check(_lib.lib().mi_sample_bpr_batch(batch, r.nnz, _ptr(r.rowptr), _ptr(r.col), _ptr(row_of_edge),
                                         int(neg_range), (1 if quirk else 0) | (2 if no_self_loops else 0),
                                         1 if edges_in_order else 0,
                                         int(seed) & (2**64 - 1),
                                         int(step) & (2**64 - 1), _ptr(users), _ptr(pos), _ptr(neg),
                                         _stream()), "mi_sample_bpr_batch")
    return users, pos, neg


def bpr_fwd_bwd(users: Tensor, pos: Tensor, neg: Tensor, final_emb: Tensor, e0: Tensor, n_users: int,
                lambda_val: float, *, g_final: Optional[Tensor] = None, reg_w: Optional[Tensor] = None,
                g_scale: float = 1.0, reg_scale: float = 1.0, loss_out: Optional[Tensor] = None,
                node_map: Optional[Tensor] = None) -> Tensor:
    """a7+a8 — batch gather + bpr_loss (utils/metrics_lightgcn.py:9-45) forward and backward."""
    for n, x in (("users", users), ("pos", pos), ("neg", neg)):
        _need(x, t.int64, n)
    batch = users.numel()
    d = final_emb.shape[1]
    ldf = _rows_ok(final_emb, "final_emb")
    lde = _rows_ok(e0, "e0")
    ldg = _rows_ok(g_final, "g_final") if g_final is not None else 0
    if reg_w is not None:
        _need(reg_w, t.float32, "reg_w")
    dev = final_emb.device
    if loss_out is None:
        loss_out = t.empty(1, dtype=t.float32, device=dev)
    L = _lib.lib()
    ws = _ws(L.mi_bpr_workspace_bytes(batch), dev)
    check(L.mi_bpr_fwd_bwd_f32(batch, d, n_users, _ptr(users), _ptr(pos), _ptr(neg), final_emb.data_ptr(), ldf,
                               e0.data_ptr(), lde, float(lambda_val), float(g_scale), float(reg_scale),
                               loss_out.data_ptr(), _ptr(g_final), ldg, _ptr(reg_w), _ptr(node_map), ws.data_ptr(),
                               ws.numel(), _stream()), "mi_bpr_fwd_bwd_f32")
    return loss_out


def adam_step(p: Tensor, grad: Tensor, m: Tensor, v: Tensor, *, step: int, lr: float, beta1: float = 0.9,
              beta2: float = 0.999, eps: float = 1e-8, reg_w: Optional[Tensor] = None) -> None:
    """a9 — dense Adam (torch.optim.Adam semantics) in one pass."""
    ldp = _rows_ok(p, "p")
    ldg = _rows_ok(grad, "grad")
    _need(m, t.float32, "m")
    _need(v, t.float32, "v")
    if m.shape != p.shape or v.shape != p.shape or grad.shape != p.shape:
        raise ValueError("p, grad, m, v must have equal shapes")
    check(_lib.lib().mi_adam_dense_f32(p.shape[0], p.shape[1], p.data_ptr(), ldp, grad.data_ptr(), ldg,
                                       m.data_ptr(), v.data_ptr(), _ptr(reg_w), float(lr), float(beta1),
                                       float(beta2), float(eps), int(step), _stream()), "mi_adam_dense_f32")


def gemm(A: Tensor, B: Tensor, *, trans_a: bool = False, trans_b: bool = True, bias: Optional[Tensor] = None,
         out: Optional[Tensor] = None, accumulate: bool = False, relu: bool = False) -> Tensor:
    """K6 — C = act(op(A) @ op(B) + bias (+ C)) on the f32 MFMA.  Defaults give nn.Linear: A [m, k],
    B = weight [n, k] (trans_b).  trans_a: A is stored [k, m];  trans_b=False: B is stored [k, n]."""
    lda = _rows_ok(A, "A")
    ldb = _rows_ok(B, "B")
    m, k = (A.shape[1], A.shape[0]) if trans_a else (A.shape[0], A.shape[1])
    n, kb = (B.shape[0], B.shape[1]) if trans_b else (B.shape[1], B.shape[0])
    if k != kb:
        raise ValueError(f"inner dimensions differ: {k} vs {kb}")
    if bias is not None:
        _need(bias, t.float32, "bias")
        if bias.numel() != n:
            raise ValueError("bias must have n entries")
    if out is None:
        if accumulate:
            raise ValueError("accumulate needs an output tensor")
        out = t.empty(m, n, dtype=t.float32, device=A.device)
    ldc = _rows_ok(out, "out")
    if out.shape != (m, n):
        raise ValueError(f"out must be [{m}, {n}]")
    L = _lib.lib()
    ws_bytes = L.mi_gemm_workspace_bytes(m, n, k)
    ws = _ws_reused(ws_bytes, A.device) if ws_bytes else None
    check(L.mi_gemm_f32(1 if trans_a else 0, 1 if trans_b else 0, m, n, k, _ptr(A), lda, _ptr(B), ldb,
                        _ptr(bias), out.data_ptr(), ldc, 1 if accumulate else 0, 1 if relu else 0,
                        ws.data_ptr() if ws is not None else None, ws_bytes, _stream()), "mi_gemm_f32")
    return out


def gemm_problem(A: Tensor, B: Tensor, out: Tensor, *, trans_a: bool = False, trans_b: bool = True,
                 A2: Optional[Tensor] = None, B2: Optional[Tensor] = None, a_mask: Optional[Tensor] = None,
                 bias: Optional[Tensor] = None, accumulate: bool = False, relu: bool = False):
    """One problem of gemm_group: out = act(op(A) @ op(B) [+ op(A2) @ op(B2)] + bias (+ out)); a_mask: A is read as 0
    where a_mask <= 0 (same shape and strides as A).  Shapes as in gemm().  Returns (descriptor, tensors): the
    descriptor holds raw pointers, the tuple keeps their storage alive until the launch has been enqueued."""
    lda, ldb, ldc = _rows_ok(A, "A"), _rows_ok(B, "B"), _rows_ok(out, "out")
    m, k = (A.shape[1], A.shape[0]) if trans_a else (A.shape[0], A.shape[1])
    n, kb = (B.shape[0], B.shape[1]) if trans_b else (B.shape[1], B.shape[0])
    if k != kb or tuple(out.shape) != (m, n):
        raise ValueError(f"gemm problem shapes differ: A {tuple(A.shape)}, B {tuple(B.shape)}, out {tuple(out.shape)}")
    k2 = lda2 = ldb2 = 0
    if A2 is not None:
        lda2, ldb2 = _rows_ok(A2, "A2"), _rows_ok(B2, "B2")
        m2, k2 = (A2.shape[1], A2.shape[0]) if trans_a else (A2.shape[0], A2.shape[1])
        n2, kb2 = (B2.shape[0], B2.shape[1]) if trans_b else (B2.shape[1], B2.shape[0])
        if (m2, n2) != (m, n) or k2 != kb2:
            raise ValueError("second pair must share m, n")
    if a_mask is not None and (_rows_ok(a_mask, "a_mask") != lda or a_mask.shape != A.shape):
        raise ValueError("a_mask must have A's shape and leading dimension")
    return (_lib.GemmProblem(1 if trans_a else 0, 1 if trans_b else 0, m, n, k, A.data_ptr(), lda, B.data_ptr(), ldb,
                             k2, _ptr(A2), lda2, _ptr(B2), ldb2, _ptr(a_mask), _ptr(bias), out.data_ptr(), ldc,
                             1 if accumulate else 0, 1 if relu else 0), (A, B, out, A2, B2, a_mask, bias))


def gemm_group(problems) -> bool:
    """Runs a list of gemm_problem()s in one launch per 8 (+ one grouped split-K reduce).  Returns False — nothing
    launched for that call — when an operand is not float4-addressable: the caller then issues gemm()s."""
    n = len(problems)
    if n == 0:
        return True
    arr = (_lib.GemmProblem * n)(*[p[0] for p in problems])  # p[1] keeps the operands alive across the allocation below
    L = _lib.lib()
    ws_bytes = L.mi_gemm_group_workspace_bytes(arr, n)
    ws = _ws_reused(ws_bytes, t.cuda.current_device()) if ws_bytes else None
    rc = L.mi_gemm_group_f32(arr, n, ws.data_ptr() if ws is not None else None, ws_bytes, _stream())
    if rc == _lib.MI_ERR_UNSUPPORTED:
        return False
    check(rc, "mi_gemm_group_f32")
    return True


TOPK_WS_BYTES = 2 << 30  # score block per launch; queries are processed in chunks of this size.  A/B at 100 K items, 65 536 queries
                         # (tools/bench_topk.py --ws-gib): 1 GiB (2 048 queries per chunk): 10.02 M / 8.03 M users/s (k = 12 / 256),
                         # 2 GiB (4 096): 10.63 M / 8.60 M, 4 GiB (8 192): 10.29 M / 8.09 M, 8 GiB (20 480): 10.69 M / 8.14 M
TOPK_CHUNK_QUANTUM = 2048  # chunks are multiples of this when they can be: the bf16 prefilter kernel (csrc/topk_prefilter.hpp)
                           # keeps 256 queries per workgroup and deals the workgroups of an item slice to one XCD's 32 CUs —
                           # 2 048 / 4 096 queries fill the chip in one round (100 K items: 4 096 per chunk)
# (Two-stream chunking: slower for the f32 fused kernel — 3.50-3.55 M -> 2.95-3.02 M users/s at k = 12, its grid fills the chip
# in whole rounds and a second one beside it breaks the rounds of both — but a gain for the bf16 prefilter path, which is the
# default: see TOPK_STREAMS below.)


# streams the chunks of one topk_excl call alternate over (each with its own workspace).  A/B, 65 536 queries against 100 K items
# (tools/bench_topk.py --streams N): 1: 8.45 M / 6.17 M users/s (k = 12 / 256), 2: 8.87 M / 6.70 M, 3: 8.55 M / 6.65 M — one
# chunk's side kernels (split, sample scores, threshold, refine) run beside the other's prefilter kernel, which holds one
# wavefront per SIMD and 96 KB of LDS per CU.  (The same switch was slower for the f32 fused kernel, whose grid fills the chip
# in whole rounds of two workgroups per CU.)
TOPK_STREAMS = int(_os_environ.get("LAPLACE_TOPK_STREAMS", 2))
_TOPK_SIDE = {}


def _topk_side_streams(dev, n: int):
    key = (t.device(dev).index if t.device(dev).index is not None else t.cuda.current_device(), n)
    if key not in _TOPK_SIDE:
        _TOPK_SIDE[key] = [t.cuda.Stream(device=dev) for _ in range(n)]
    return _TOPK_SIDE[key]


def topk_excl(uid: Tensor, user_emb: Tensor, item_emb: Tensor, k: int, excl: Optional[DeviceCSR] = None,
              want_scores: bool = False):
    """K10 — for each query user uid[q]: the k best item ids by (score desc, id asc) among items not in
    excl row q (a CSR with one row per query position).  Returns ids [n_q, k] (-1 pads) and optionally scores."""
    _need(uid, t.int64, "uid")
    ldu = _rows_ok(user_emb, "user_emb")
    ldi = _rows_ok(item_emb, "item_emb")
    n_q, n_items, d = uid.numel(), item_emb.shape[0], item_emb.shape[1]
    if user_emb.shape[1] != d:
        raise ValueError("user and item embeddings differ in width")
    if excl is not None and excl.n_rows != n_q:
        raise ValueError("exclusion CSR needs one row per query")
    dev = item_emb.device
    out_idx = t.empty(n_q, k, dtype=t.int64, device=dev)
    out_sc = t.empty(n_q, k, dtype=t.float32, device=dev) if want_scores else None
    L = _lib.lib()
    chunk = max(1, min(n_q, TOPK_WS_BYTES // max(4 * n_items, 1)))
    if n_q > chunk >= TOPK_CHUNK_QUANTUM:
        chunk -= chunk % TOPK_CHUNK_QUANTUM
    n_chunks = (n_q + chunk - 1) // chunk
    lanes = TOPK_STREAMS if (TOPK_STREAMS > 1 and n_chunks > 1) else 1
    # transient footprint = lanes x workspace (~2.2 GiB each at the default chunk): fall back to one lane, then to smaller
    # chunks, when the device does not have that much free
    ws_bytes = int(L.mi_topk_workspace_bytes(chunk, n_items, k))
    free = t.cuda.mem_get_info(dev)[0] + t.cuda.memory_reserved(dev) - t.cuda.memory_allocated(dev)
    if lanes > 1 and lanes * ws_bytes > 0.8 * free:
        lanes = 1
    while ws_bytes > 0.8 * free and chunk > 256:
        chunk = max(256, chunk // 2)
        ws_bytes = int(L.mi_topk_workspace_bytes(chunk, n_items, k))
    wss = [_ws(ws_bytes, dev) for _ in range(lanes)]

    prepared = {}   # workspace -> the chunk size whose item-side tables it holds (MI_TOPK_ITEMS_PREPARED)

    def launch(q0: int, ws: Tensor):
        q1 = min(n_q, q0 + chunk)
        # rowptr keeps absolute offsets into excl.col, so a chunk is just a slice of rowptr
        ep = excl.rowptr[q0:q1 + 1] if excl is not None else None
        ei = excl.col if (excl is not None and excl.nnz) else None
        # from a workspace's second chunk of the same size on, the item side of the prefilter (sample rows, the bf16 split of the
        # item table, the largest item norm: ~40 us per chunk at 100 K items) is still in it: same table, same layout
        flags = _lib.MI_TOPK_ITEMS_PREPARED if prepared.get(id(ws)) == q1 - q0 else 0
        check(L.mi_topk_excl_ex_f32(q1 - q0, n_items, d, k, uid[q0:q1].data_ptr(), user_emb.data_ptr(), ldu,
                                    item_emb.data_ptr(), ldi, _ptr(ep), _ptr(ei),
                                    out_idx[q0:q1].data_ptr(), out_sc[q0:q1].data_ptr() if want_scores else None,
                                    ws.data_ptr(), ws.numel(), flags, _stream()), "mi_topk_excl_ex_f32")
        prepared[id(ws)] = q1 - q0

    if lanes == 1:
        for q0 in range(0, n_q, chunk):
            launch(q0, wss[0])
    else:
        # chunks alternate over `lanes` streams, each with its own workspace: one chunk's side kernels (split, sample scores,
        # threshold, refine: ~30 % of its time) run beside the other's prefilter kernel, which holds one wavefront per SIMD
        cur = t.cuda.current_stream()
        side = _topk_side_streams(dev, lanes)
        for st in side:
            st.wait_stream(cur)
        for i, q0 in enumerate(range(0, n_q, chunk)):
            with t.cuda.stream(side[i % lanes]):
                launch(q0, wss[i % lanes])
        for st, ws in zip(side, wss):
            cur.wait_stream(st)
            ws.record_stream(st)
        for x in (uid, user_emb, item_emb, out_idx, out_sc, excl.rowptr if excl is not None else None,
                  excl.col if excl is not None else None):
            if x is not None:
                for st in side:
                    x.record_stream(st)
    return (out_idx, out_sc) if want_scores else out_idx


def topk_prefilter_scores(uid: Optional[Tensor], user_emb: Tensor, item_emb: Tensor) -> Tuple[Tensor, Tensor]:
    """Diagnostic of the bf16x3 prefilter behind topk_excl (csrc/topk_prefilter.hpp): the approximate scores [n_q, n_items]
    it compares with its thresholds and the per-query bound eps it assumes for |approximate - exact fma chain|."""
    ldu = _rows_ok(user_emb, "user_emb")
    ldi = _rows_ok(item_emb, "item_emb")
    n_q = uid.numel() if uid is not None else user_emb.shape[0]
    n_items, d = item_emb.shape
    if uid is not None:
        _need(uid, t.int64, "uid")
    dev = item_emb.device
    sc = t.empty(n_q, n_items, dtype=t.float32, device=dev)
    eps = t.empty(n_q, dtype=t.float32, device=dev)
    L = _lib.lib()
    ws = _ws(L.mi_topk_prefilter_scores_workspace_bytes(n_q, n_items), dev)
    check(L.mi_topk_prefilter_scores_f32(n_q, n_items, d, _ptr(uid), user_emb.data_ptr(), ldu, item_emb.data_ptr(), ldi,
                                         sc.data_ptr(), eps.data_ptr(), ws.data_ptr(), ws.numel(), _stream()),
          "mi_topk_prefilter_scores_f32")
    return sc, eps


def segment_max(a: DeviceCSR, X: Tensor, want_arg: bool = True):
    """K5 (aggr="max") — Y[r] = max over the sources of destination r (0 for none); arg = winning source ids."""
    ldx = _rows_ok(X, "X")
    d = X.shape[1]
    Y = t.empty(a.n_rows, d, dtype=t.float32, device=X.device)
    arg = t.empty(a.n_rows, d, dtype=t.int32, device=X.device) if want_arg else None
    colp = _ptr(a.col) if a.nnz else a.rowptr.data_ptr()
    check(_lib.lib().mi_segment_max_f32(a.n_rows, d, _ptr(a.rowptr), colp, X.data_ptr(), ldx, Y.data_ptr(), d,
                                        _ptr(arg), _stream()), "mi_segment_max_f32")
    return Y, arg


def segment_max_bwd(by_src: DeviceCSR, arg: Tensor, dY: Tensor) -> Tensor:
    """Gradient of segment_max wrt X: by_src is the same relation as a CSR over SOURCE nodes (col = destination ids).
    Deterministic: each dX element has one writer that sums its destinations in CSR order."""
    _need(arg, t.int32, "arg")
    ldy = _rows_ok(dY, "dY")
    d = dY.shape[1]
    dX = t.empty(by_src.n_rows, d, dtype=t.float32, device=dY.device)
    if by_src.n_rows:
        colp = _ptr(by_src.col) if by_src.nnz else by_src.rowptr.data_ptr()
        check(_lib.lib().mi_segment_max_bwd_f32(by_src.n_rows, d, _ptr(by_src.rowptr), colp, arg.data_ptr(),
                                                dY.data_ptr() if dY.numel() else None, ldy, dX.data_ptr(), d, _stream()),
              "mi_segment_max_bwd_f32")
    return dX


def embed_concat(x: Tensor, tables, max_norm: float = 1.0) -> Tensor:
    """K7 — cat_i Embedding_i(max_norm)(x[:, i]).  tables: list of [rows_i, dim_i] fp32 GPU tensors."""
    _need(x, t.int64, "x")
    if x.dim() != 2 or x.shape[1] != len(tables):
        raise ValueError("x must be [n, n_cols] with one table per column")
    for i, tb in enumerate(tables):
        _need(tb, t.float32, f"tables[{i}]")
    n, nc = x.shape
    width = sum(int(tb.shape[1]) for tb in tables)
    out = t.empty(n, width, dtype=t.float32, device=x.device)
    if n == 0 or nc == 0:
        return out
    off = 0
    for c0 in range(0, nc, 16):  # the kernel takes up to 16 table descriptors by value: wider inputs go in groups
        tabs = tables[c0:c0 + 16]
        g = len(tabs)
        xs = x if (c0 == 0 and g == nc) else x[:, c0:c0 + g].contiguous()
        ptrs = (ctypes.c_void_p * g)(*[tb.data_ptr() for tb in tabs])
        rows = (ctypes.c_int64 * g)(*[int(tb.shape[0]) for tb in tabs])
        dims = (ctypes.c_int32 * g)(*[int(tb.shape[1]) for tb in tabs])
        check(_lib.lib().mi_embed_concat_f32(n, g, xs.data_ptr(), ptrs, rows, dims, float(max_norm),
                                             out.data_ptr() + 4 * off, width, _stream()), "mi_embed_concat_f32")
        off += sum(int(tb.shape[1]) for tb in tabs)
    return out


def match_common_items(users_ptr: Tensor, users_idx: Tensor, articles_ptr: Tensor, articles_idx: Tensor, k: int,
                       query_users: Optional[Tensor] = None, n_queries: Optional[int] = None):
    """N3 — UsersWithCommonItemsMatcher for many users at once: (out int32[n, k] with -1 pads, counts int32[n])."""
    for n, x in (("users_ptr", users_ptr), ("users_idx", users_idx), ("articles_ptr", articles_ptr), ("articles_idx", articles_idx)):
        _need(x, t.int32, n)
    if query_users is not None:
        _need(query_users, t.int64, "query_users")
        n_queries = query_users.numel()
    elif n_queries is None:
        n_queries = users_ptr.numel() - 1
    out = t.empty(n_queries, int(k), dtype=t.int32, device=users_ptr.device)
    cnt = t.empty(n_queries, dtype=t.int32, device=users_ptr.device)
    check(_lib.lib().mi_match_common_items_i32(n_queries, _ptr(query_users), users_ptr.data_ptr(), users_idx.data_ptr() if users_idx.numel() else users_ptr.data_ptr(),
                                               articles_ptr.data_ptr(), articles_idx.data_ptr() if articles_idx.numel() else articles_ptr.data_ptr(),
                                               int(k), _ptr(out), _ptr(cnt), _stream()), "mi_match_common_items_i32")
    return out, cnt


def match_same_location(location_of_user: Tensor, loc_ptr: Tensor, loc_idx: Tensor, users_ptr: Tensor, users_idx: Tensor,
                        k: int, query_users: Optional[Tensor] = None, n_queries: Optional[int] = None):
    """N3 — UsersSameLocationMatcher for many users at once: (out int32[n, k] with -1 pads, counts int32[n])."""
    for n, x in (("location_of_user", location_of_user), ("loc_ptr", loc_ptr), ("loc_idx", loc_idx),
                 ("users_ptr", users_ptr), ("users_idx", users_idx)):
        _need(x, t.int32, n)
    if query_users is not None:
        _need(query_users, t.int64, "query_users")
        n_queries = query_users.numel()
    elif n_queries is None:
        n_queries = location_of_user.numel()
    if location_of_user.numel() != users_ptr.numel() - 1:
        raise ValueError("location_of_user must have one entry per user")
    dev = users_ptr.device
    out = t.empty(n_queries, int(k), dtype=t.int32, device=dev)
    cnt = t.empty(n_queries, dtype=t.int32, device=dev)
    check(_lib.lib().mi_match_same_location_i32(n_queries, _ptr(query_users), location_of_user.data_ptr(), loc_ptr.data_ptr(),
                                                loc_idx.data_ptr() if loc_idx.numel() else loc_ptr.data_ptr(),
                                                users_ptr.data_ptr(), users_idx.data_ptr() if users_idx.numel() else users_ptr.data_ptr(),
                                                int(k), _ptr(out), _ptr(cnt), _stream()), "mi_match_same_location_i32")
    return out, cnt


_BN_WS = {}


def _bn_ws(c: int, device) -> Tensor:
    key = (t.device(device).index, int(c))
    if key not in _BN_WS:
        _BN_WS[key] = _ws(_lib.lib().mi_batchnorm_workspace_bytes(int(c)), device)
    return _BN_WS[key]


def batchnorm_fwd(x: Tensor, gamma: Optional[Tensor], beta: Optional[Tensor], running_mean: Optional[Tensor],
                  running_var: Optional[Tensor], momentum: float, eps: float, training: bool):
    """K8 — BatchNorm1d over the rows of x [n, c].  Returns (y, save_mean, save_invstd); the running statistics are
    updated in place in training mode (torch semantics: unbiased variance, momentum)."""
    ldx = _rows_ok(x, "x")
    n, c = x.shape
    y = t.empty(n, c, dtype=t.float32, device=x.device)
    sm = t.empty(c, dtype=t.float32, device=x.device) if training else None
    si = t.empty(c, dtype=t.float32, device=x.device) if training else None
    ws = _bn_ws(c, x.device)
    check(_lib.lib().mi_batchnorm_fwd_f32(n, c, _ptr(x), ldx, _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var),
                                          float(momentum), float(eps), 1 if training else 0, _ptr(sm), _ptr(si), _ptr(y), c,
                                          ws.data_ptr(), ws.numel(), _stream()), "mi_batchnorm_fwd_f32")
    return y, sm, si


def batchnorm_bwd(x: Tensor, dy: Tensor, gamma: Optional[Tensor], save_mean: Tensor, save_invstd: Tensor,
                  need_dx: bool = True, need_dw: bool = True):
    """(dx, dgamma, dbeta) of batchnorm_fwd in training mode."""
    ldx, ldy = _rows_ok(x, "x"), _rows_ok(dy, "dy")
    n, c = x.shape
    dx = t.empty(n, c, dtype=t.float32, device=x.device) if need_dx else None
    dg = t.empty(c, dtype=t.float32, device=x.device) if need_dw else None
    db = t.empty(c, dtype=t.float32, device=x.device) if need_dw else None
    ws = _bn_ws(c, x.device)
    check(_lib.lib().mi_batchnorm_bwd_f32(n, c, _ptr(x), ldx, _ptr(dy), ldy, _ptr(gamma), _ptr(save_mean), _ptr(save_invstd),
                                          _ptr(dx), c, _ptr(dg), _ptr(db), ws.data_ptr(), ws.numel(), _stream()),
          "mi_batchnorm_bwd_f32")
    return dx, dg, db


def bce_logits(logits: Tensor, labels: Tensor, want_grad: bool = True):
    """b9 — BCEWithLogitsLoss(mean)(logits, labels) and d loss / d logits in one launch: (loss[1], dlogits or None)."""
    _need(logits, t.float32, "logits")
    _need(labels, t.float32, "labels")
    n = logits.numel()
    if labels.numel() != n or n == 0:
        raise ValueError("logits and labels must be non-empty and of equal length")
    loss = t.empty(1, dtype=t.float32, device=logits.device)
    dl = t.empty(n, dtype=t.float32, device=logits.device) if want_grad else None
    check(_lib.lib().mi_bce_logits_f32(n, _ptr(logits), _ptr(labels), _ptr(loss), _ptr(dl), _stream()), "mi_bce_logits_f32")
    return loss, dl


def gather_cat(zu: Tensor, zi: Tensor, row: Tensor, col: Tensor) -> Tensor:
    """b7 — cat(zu[row], zi[col], dim=-1) in one launch (model/encoder_decoder.py:57-63)."""
    _need(row, t.int64, "row")
    _need(col, t.int64, "col")
    ldu, ldi = _rows_ok(zu, "zu"), _rows_ok(zi, "zi")
    cu, ci, ne = zu.shape[1], zi.shape[1], row.numel()
    out = t.empty(ne, cu + ci, dtype=t.float32, device=zu.device)
    check(_lib.lib().mi_gather_cat_f32(ne, cu, ci, _ptr(row), _ptr(col), _ptr(zu), ldu, _ptr(zi), ldi, _ptr(out), cu + ci,
                                       _stream()), "mi_gather_cat_f32")
    return out


def gather_cat_bwd(d_out: Tensor, idx: Tensor, n_rows: int, c: int, off: int) -> Tensor:
    """dZ [n_rows, c]: rows of d_out[:, off:off+c] summed per idx value, in edge order (deterministic: no float
    atomics on either path)."""
    _need(idx, t.int64, "idx")
    ldo = _rows_ok(d_out, "d_out")
    ne = idx.numel()
    if ne > int(_lib.lib().mi_gather_cat_bwd_max_edges()):
        # beyond the all-pairs kernel's range: the same sum as a product with the [n_rows, ne] incidence matrix — a sorted
        # CSR (columns = edge positions, ascending) through the SpMM kernel, i.e. still summed in edge order
        if c % 4 or off % 4 or ldo % 4:
            raise ValueError("gather_cat_bwd beyond mi_gather_cat_bwd_max_edges needs widths / offsets that are multiples of 4")
        inc = coo_to_csr(idx.contiguous(), t.arange(ne, dtype=t.int64, device=idx.device), n_rows, ne, want_perm=False)
        inc.val = t.ones(ne, dtype=t.float32, device=idx.device)
        dz = t.empty(n_rows, c, dtype=t.float32, device=d_out.device)
        spmm(inc, d_out[:, off:off + c], Y=dz)
        return dz
    dz = t.zeros(n_rows, c, dtype=t.float32, device=d_out.device)
    check(_lib.lib().mi_gather_cat_bwd_f32(ne, c, off, _ptr(idx), _ptr(d_out), ldo, _ptr(dz), c, _stream()),
          "mi_gather_cat_bwd_f32")
    return dz


def sage_wgrad(problems) -> bool:
    """Weight gradients of up to four SAGEConv relations in one launch pair (mi_sage_wgrad_f32).  problems: dicts(dy [k, m],
    mask [k, m] or None, b1 [k, n1], b2 [k, n2] or None, gw1 [m, n1], gb [m] or None, gw2 [m, n2] or None), all contiguous
    float32.  False (nothing enqueued) when the shapes are outside the kernel's: the caller uses the grouped GEMM."""
    n = len(problems)
    if n == 0 or n > 4:
        return False
    arr = (_lib.WgradProblem * n)()
    for q, s in zip(arr, problems):
        dy, b1, b2 = s["dy"], s["b1"], s.get("b2")
        tensors = [dy, b1, s["gw1"]] + [x for x in (s.get("mask"), b2, s.get("gb"), s.get("gw2")) if x is not None]
        if any(x.dtype != t.float32 or not x.is_contiguous() for x in tensors) or (b2 is None) != (s.get("gw2") is None):
            return False
        if dy.dim() != 2 or b1.dim() != 2 or b1.shape[0] != dy.shape[0] or (b2 is not None and b2.shape[0] != dy.shape[0]):
            return False
        q.k, q.m, q.n1, q.n2 = int(dy.shape[0]), int(dy.shape[1]), int(b1.shape[1]), int(b2.shape[1]) if b2 is not None else 0
        q.dy, q.mask, q.b1, q.b2 = _ptr(dy), _ptr(s.get("mask")), _ptr(b1), _ptr(b2)
        q.gw1, q.gb, q.gw2 = _ptr(s["gw1"]), _ptr(s.get("gb")), _ptr(s.get("gw2"))
    L = _lib.lib()
    if not L.mi_sage_wgrad_supported(arr, n):
        return False
    ws = t.empty(max(int(L.mi_sage_wgrad_workspace_bytes(arr, n)), 256), dtype=t.uint8, device=problems[0]["dy"].device)
    rc = L.mi_sage_wgrad_f32(arr, n, _ptr(ws), ws.numel(), _stream())
    if rc == _lib.MI_ERR_UNSUPPORTED:
        return False
    check(rc, "mi_sage_wgrad_f32")
    return True


def linear1_bwd(dy: Tensor, weight: Tensor, x: Tensor, need_dx: bool = True):
    """Backward of a Linear layer with one output feature (weight [1, in]): (dx [n, in] or None, dW [1, in], db [1]).
    mi_linear1_bwd_f32: deterministic band sums instead of three [n, 1]-shaped products."""
    _need(dy, t.float32, "dy"); _need(weight, t.float32, "weight")
    ldx = _rows_ok(x, "x")
    n, k = int(x.shape[0]), int(x.shape[1])
    if dy.numel() != n or weight.numel() != k or not dy.is_contiguous() or not weight.is_contiguous():
        raise ValueError("linear1_bwd: dy [n] / [n, 1], weight [1, in], x [n, in] expected")
    dev = x.device
    dx = t.empty(n, k, dtype=t.float32, device=dev) if need_dx else None
    gw = t.empty(1, k, dtype=t.float32, device=dev)
    gb = t.empty(1, dtype=t.float32, device=dev)
    L = _lib.lib()
    ws = t.empty(int(L.mi_linear1_bwd_workspace_bytes(n, k)), dtype=t.uint8, device=dev)
    check(L.mi_linear1_bwd_f32(n, k, _ptr(dy), _ptr(weight), _ptr(x), ldx, _ptr(dx) if need_dx else None, k, _ptr(gw), _ptr(gb),
                               _ptr(ws), ws.numel(), _stream()), "mi_linear1_bwd_f32")
    return dx, gw, gb


def batch_nodes(users: Tensor, pos: Tensor, neg: Tensor, n_users: int, n_nodes: int, *, gmap: Optional[Tensor] = None,
                nodes: Optional[Tensor] = None, count: Optional[Tensor] = None, ws: Optional[Tensor] = None):
    """Unique node set of a BPR batch: (gmap int32[n_nodes], nodes int32[3B], count int32[2] on device:
    count[0] = unique nodes, count[1] = unique user nodes, whose slots come first)."""
    for n, x in (("users", users), ("pos", pos), ("neg", neg)):
        _need(x, t.int64, n)
    B, dev = users.numel(), users.device
    gmap = gmap if gmap is not None else t.empty(n_nodes, dtype=t.int32, device=dev)
    nodes = nodes if nodes is not None else t.empty(3 * B, dtype=t.int32, device=dev)
    count = count if count is not None else t.empty(2, dtype=t.int32, device=dev)
    L = _lib.lib()
    ws = ws if ws is not None else _ws(L.mi_batch_nodes_workspace_bytes(n_nodes), dev)
    check(L.mi_batch_nodes_i32(B, n_users, n_nodes, _ptr(users), _ptr(pos), _ptr(neg), _ptr(gmap), _ptr(nodes),
                               _ptr(count), ws.data_ptr(), ws.numel(), _stream()), "mi_batch_nodes_i32")
    return gmap, nodes, count


def gather_rows(dst: Tensor, src: Tensor, rows: Tensor, n_dev: Optional[Tensor] = None, accumulate: bool = False,
                scale: float = 1.0, begin_dev: Optional[Tensor] = None, row_offset: int = 0) -> None:
    """dst[i] = scale * ((accumulate ? dst[i] : 0) + src[rows[i] - row_offset]) for i in [*begin_dev, *n_dev)."""
    _need(rows, t.int32, "rows")
    lds_, ldd = _rows_ok(src, "src"), _rows_ok(dst, "dst")
    if dst.shape[0] < rows.numel() or dst.shape[1] != src.shape[1]:
        raise ValueError("dst must be [len(rows), d]")
    check(_lib.lib().mi_gather_rows_f32(rows.numel(), _ptr(n_dev), _ptr(begin_dev), src.shape[1], _ptr(rows),
                                        int(row_offset), src.data_ptr(), lds_, dst.data_ptr(), ldd,
                                        1 if accumulate else 0, float(scale), _stream()), "mi_gather_rows_f32")


def scatter_rows(dst: Tensor, src: Tensor, rows: Tensor, n_dev: Optional[Tensor] = None,
                 begin_dev: Optional[Tensor] = None, row_offset: int = 0) -> None:
    """dst[rows[i] - row_offset] = src[i] for i in [*begin_dev, *n_dev) (rows distinct)."""
    _need(rows, t.int32, "rows")
    lds_, ldd = _rows_ok(src, "src"), _rows_ok(dst, "dst")
    if src.shape[0] < rows.numel() or dst.shape[1] != src.shape[1]:
        raise ValueError("src must be [len(rows), d]")
    check(_lib.lib().mi_scatter_rows_f32(rows.numel(), _ptr(n_dev), _ptr(begin_dev), src.shape[1], _ptr(rows),
                                         int(row_offset), src.data_ptr(), lds_, dst.data_ptr(), ldd, _stream()),
          "mi_scatter_rows_f32")

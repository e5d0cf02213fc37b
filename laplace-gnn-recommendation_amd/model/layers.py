"""Ranker layers — factories with the reference's names and return types (model/layers.py:6-56):
`get_SAGEConv_layers` -> ModuleList of SAGEConv, `get_linear_layers` -> ModuleList of Linear.

SAGEConv follows PyG 2.0.4 (`SAGEConv((-1,-1,-1), C, aggr, normalize=False, bias=True)`):
    out = lin_l( AGG_{j -> i} x_src[j] ) + lin_r( x_dst[i] )          lin_l has the bias, lin_r none
with lazily sized weights.  Aggregation runs on the HIP SpMM (add / mean) or segment-max kernel
over a CSR sorted by destination — no [E, C] message tensor, no atomics in the forward; the two
Linear products run on the f32-MFMA GEMM.  Both are torch.autograd.Functions whose backward calls
the same kernels (transposed CSR; dX = dY W, dW = dY^T X).
"""
from __future__ import annotations

import math
from copy import deepcopy
from typing import Optional, Tuple, Union

import torch as t
import torch.nn.functional as F
from torch import Tensor, nn

from .. import ops


# ------------------------------------------------------------------------------------------------
# Linear on mi_gemm_f32
# ------------------------------------------------------------------------------------------------
class _LinearFn(t.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias: Optional[Tensor], relu: bool):
        x2 = x if x.stride(-1) == 1 else x.contiguous()
        y = ops.gemm(x2, weight, trans_b=True, bias=bias, relu=relu)
        ctx.save_for_backward(x2, weight, y if relu else None)
        ctx.has_bias, ctx.relu = bias is not None, relu
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        x, weight, y = ctx.saved_tensors
        dy = dy.contiguous()
        if ctx.relu:
            dy = dy * (y > 0)
        dx = ops.gemm(dy, weight, trans_b=False) if ctx.needs_input_grad[0] else None   # [m,n] @ [n,k]
        dw = ops.gemm(dy, x, trans_a=True, trans_b=False) if ctx.needs_input_grad[1] else None  # dy^T @ x
        db = dy.sum(dim=0) if ctx.has_bias else None
        return dx, dw, db, None


class Linear(nn.Module):
    """torch.nn.Linear semantics and state_dict keys (weight [out, in], bias [out]); in_features = -1
    sizes the weight on the first forward (PyG lazy Linear)."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True):
        super().__init__()
        self.in_features, self.out_features, self.use_bias = int(in_features), int(out_features), bias
        if self.in_features > 0:
            self._materialize(self.in_features, None)
        else:
            self.register_parameter("weight", None)
            self.register_parameter("bias", None)

    def _materialize(self, in_features: int, device) -> None:
        self.in_features = in_features
        self.weight = nn.Parameter(t.empty(self.out_features, in_features, device=device))
        self.bias = nn.Parameter(t.empty(self.out_features, device=device)) if self.use_bias else None
        self.reset_parameters()

    def reset_parameters(self) -> None:
        if self.weight is None:
            return
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))  # = U(-1/sqrt(in), 1/sqrt(in)), as nn.Linear / PyG
        if self.bias is not None:
            bound = 1 / math.sqrt(self.in_features) if self.in_features > 0 else 0
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x: Tensor, relu: bool = False) -> Tensor:
        if self.weight is None:
            self._materialize(int(x.shape[-1]), x.device)
        return _LinearFn.apply(x, self.weight, self.bias, relu)

    def extra_repr(self) -> str:
        return f"in_features={self.in_features}, out_features={self.out_features}, bias={self.use_bias}"


# ------------------------------------------------------------------------------------------------
# message passing on a destination-sorted CSR
# ------------------------------------------------------------------------------------------------
class BipartiteGraph:
    """One relation of a batch: edges src -> dst as two sorted CSRs (by dst for the forward aggregate,
    by src for its backward), built once per batch on device and shared by all layers."""

    def __init__(self, edge_index: Optional[Tensor], n_src: int, n_dst: int):
        self.n_src, self.n_dst = int(n_src), int(n_dst)
        self._vals = {}
        if edge_index is None:  # filled by reversed()
            return
        src, dst = edge_index[0].contiguous(), edge_index[1].contiguous()
        self.by_dst = ops.coo_to_csr(dst, src, self.n_dst, self.n_src, want_perm=False)
        self.by_src = ops.coo_to_csr(src, dst, self.n_src, self.n_dst, want_perm=False)

    @classmethod
    def of(cls, edge_index: Tensor, n_src: int, n_dst: int) -> "BipartiteGraph":
        """The relation of `edge_index`; when its producer already holds the two sorted CSRs (the device sampler
        attaches them as `_sorted_csr` = (by source, by destination)) they are taken as they are."""
        pre = getattr(edge_index, "_sorted_csr", None)
        if pre is None or pre[0].n_rows != int(n_src) or pre[1].n_rows != int(n_dst):
            return cls(edge_index, n_src, n_dst)
        g = cls(None, n_src, n_dst)
        g.by_src, g.by_dst = pre
        return g

    def reversed(self) -> "BipartiteGraph":
        """The relation with every edge turned round (the reference's `rev_*` relation is `edge_index.flip(0)`,
        data/dataset.py:176-182): the same two CSRs with their roles swapped — no second pair of sorts."""
        g = BipartiteGraph(None, self.n_dst, self.n_src)
        g.by_dst, g.by_src = self.by_src, self.by_dst
        return g

    def weights(self, aggr: str) -> Tuple[Tensor, Tensor]:
        """(values for by_dst, values for by_src) realising `aggr` as a weighted sum."""
        if aggr not in self._vals:
            if aggr == "add":
                ones = t.ones(self.by_dst.nnz, device=self.by_dst.device)
                self._vals[aggr] = (ones, ones)
            elif aggr == "mean":
                deg = (self.by_dst.rowptr[1:] - self.by_dst.rowptr[:-1]).to(t.float32)
                inv = t.where(deg > 0, 1.0 / deg, t.zeros_like(deg))
                self._vals[aggr] = (ops.scale_csr(self.by_dst, None, inv, None), ops.scale_csr(self.by_src, None, None, inv))
            else:
                raise ValueError(aggr)
        return self._vals[aggr]


def _pad4(x: Tensor) -> Tensor:
    pad = (-x.shape[1]) % 4
    x = x if x.stride(-1) == 1 else x.contiguous()
    return F.pad(x, (0, pad)) if pad else x


class _AggregateFn(t.autograd.Function):
    @staticmethod
    def forward(ctx, x_src: Tensor, graph: BipartiteGraph, aggr: str):
        d = x_src.shape[1]
        ctx.graph, ctx.aggr, ctx.d = graph, aggr, d
        if aggr == "max":
            y, arg = ops.segment_max(graph.by_dst, x_src if x_src.stride(-1) == 1 else x_src.contiguous())
            ctx.save_for_backward(arg)
            return y
        xp = _pad4(x_src)
        v_dst, _ = graph.weights(aggr)
        a = ops.DeviceCSR(graph.by_dst.n_rows, graph.by_dst.n_cols, graph.by_dst.rowptr, graph.by_dst.col, v_dst, None,
                          graph.by_dst.plan)
        y = t.empty(graph.n_dst, xp.shape[1], device=x_src.device)
        ops.spmm(a, xp, Y=y)
        graph.by_dst.plan = a.plan
        return y[:, :d] if xp.shape[1] != d else y

    @staticmethod
    def backward(ctx, dy: Tensor):
        graph, aggr, d = ctx.graph, ctx.aggr, ctx.d
        if aggr == "max":
            (arg,) = ctx.saved_tensors
            return ops.segment_max_bwd(graph.by_src, arg, dy.contiguous()), None, None
        dyp = _pad4(dy)
        _, v_src = graph.weights(aggr)
        a = ops.DeviceCSR(graph.by_src.n_rows, graph.by_src.n_cols, graph.by_src.rowptr, graph.by_src.col, v_src, None,
                          graph.by_src.plan)
        dx = t.empty(graph.n_src, dyp.shape[1], device=dy.device)
        ops.spmm(a, dyp, Y=dx)
        graph.by_src.plan = a.plan
        return (dx[:, :d] if dyp.shape[1] != d else dx), None, None


class _SAGEConvFn(t.autograd.Function):
    """The whole layer as ONE autograd node: act(lin_l(AGG_{j->i} x_src[j]) + lin_r(x_dst[i])).

    A per-batch subgraph is ~10^4 nodes: the iteration is bound by launches and by the host's per-op cost, not
    by bytes (profiles/r2_ranker_*).  Compared with the aggregate / two Linear / add / relu chain of separate nodes
    this is 3 launches forward (SpMM or segment-max, GEMM with bias, GEMM accumulating into the same output with the
    activation in its epilogue) and one backward call that issues dAgg, A^T dAgg, dX_dst, dW_l, dW_r, db only for the
    inputs that need them — layer 0 reads frozen embeddings, so its dX products are never launched."""

    @staticmethod
    def forward(ctx, x_src: Tensor, x_dst: Optional[Tensor], w_l: Tensor, b_l: Optional[Tensor], w_r: Optional[Tensor],
                graph: BipartiteGraph, aggr: str, relu: bool):
        d = x_src.shape[1]
        arg = None
        if aggr == "max":
            agg, arg = ops.segment_max(graph.by_dst, x_src if x_src.stride(-1) == 1 else x_src.contiguous())
        else:
            xp = _pad4(x_src)
            v_dst, _ = graph.weights(aggr)
            a = ops.DeviceCSR(graph.by_dst.n_rows, graph.by_dst.n_cols, graph.by_dst.rowptr, graph.by_dst.col, v_dst, None,
                              graph.by_dst.plan)
            agg = t.empty(graph.n_dst, xp.shape[1], device=x_src.device)
            ops.spmm(a, xp, Y=agg)
            graph.by_dst.plan = a.plan
            if xp.shape[1] != d:
                agg = agg[:, :d]
        root = w_r is not None and x_dst is not None
        out = ops.gemm(agg, w_l, trans_b=True, bias=b_l, relu=relu and not root)
        if root:
            x_dst = x_dst if x_dst.stride(-1) == 1 else x_dst.contiguous()
            ops.gemm(x_dst, w_r, trans_b=True, out=out, accumulate=True, relu=relu)
        ctx.save_for_backward(agg, x_dst if root else None, w_l, w_r if root else None, arg, out if relu else None)
        ctx.graph, ctx.aggr, ctx.d, ctx.root, ctx.has_bias = graph, aggr, d, root, b_l is not None
        return out

    @staticmethod
    def backward(ctx, dy: Tensor):
        agg, x_dst, w_l, w_r, arg, out = ctx.saved_tensors
        graph, aggr, d = ctx.graph, ctx.aggr, ctx.d
        need = ctx.needs_input_grad
        dy = dy.contiguous()
        if out is not None:
            dy = dy * (out > 0)
        dx_src = dx_dst = dw_l = db = dw_r = None
        if need[0]:
            d_agg = ops.gemm(dy, w_l, trans_b=False)                       # [n_dst, C_src]
            if aggr == "max":
                dx_src = ops.segment_max_bwd(graph.by_src, arg, d_agg)
            else:
                dp = _pad4(d_agg)
                _, v_src = graph.weights(aggr)
                a = ops.DeviceCSR(graph.by_src.n_rows, graph.by_src.n_cols, graph.by_src.rowptr, graph.by_src.col, v_src,
                                  None, graph.by_src.plan)
                dx_src = t.empty(graph.n_src, dp.shape[1], device=dy.device)
                ops.spmm(a, dp, Y=dx_src)
                graph.by_src.plan = a.plan
                if dp.shape[1] != d:
                    dx_src = dx_src[:, :d]
        if ctx.root and need[1]:
            dx_dst = ops.gemm(dy, w_r, trans_b=False)
        if need[2]:
            dw_l = ops.gemm(dy, agg, trans_a=True, trans_b=False)           # dY^T @ agg
        if ctx.has_bias and need[3]:
            db = dy.sum(dim=0)
        if ctx.root and need[4]:
            dw_r = ops.gemm(dy, x_dst, trans_a=True, trans_b=False)
        return dx_src, dx_dst, dw_l, db, dw_r, None, None, None


# ------------------------------------------------------------------------------------------------
# one hetero layer (all relations) as one autograd node on the grouped GEMM
# ------------------------------------------------------------------------------------------------
_ONES4 = {}


def _ones4(n: int, device) -> Tensor:
    """[>= n, 4] of ones: dY^T @ ones gives the bias gradient inside the grouped weight-gradient launch."""
    key = t.device(device).index
    cur = _ONES4.get(key)
    if cur is None or cur.shape[0] < n:
        cur = t.ones(max(n, 1 << 16), 4, device=device)
        _ONES4[key] = cur
    return cur[:n]


def _run_products(specs) -> None:
    """specs: dicts(A, B, out, trans_a, trans_b, A2, B2, mask, bias, relu).  One grouped launch when every operand is
    float4-addressable, else one gemm() per product with the mask applied as a tensor op."""
    if not specs:
        return
    probs = [ops.gemm_problem(s["A"], s["B"], s["out"], trans_a=s.get("trans_a", False), trans_b=s.get("trans_b", True),
                              A2=s.get("A2"), B2=s.get("B2"), a_mask=s.get("mask"), bias=s.get("bias"),
                              relu=s.get("relu", False)) for s in specs]
    if ops.gemm_group(probs):
        return
    masked = {}
    for s in specs:
        A = s["A"]
        m = s.get("mask")
        if m is not None:
            key = (A.data_ptr(), m.data_ptr())
            if key not in masked:
                masked[key] = A * (m > 0)
            A = masked[key]
        two = s.get("A2") is not None
        ops.gemm(A, s["B"], trans_a=s.get("trans_a", False), trans_b=s.get("trans_b", True), bias=s.get("bias"),
                 out=s["out"], relu=s.get("relu", False) and not two)
        if two:
            ops.gemm(s["A2"], s["B2"], trans_a=s.get("trans_a", False), trans_b=s.get("trans_b", True), out=s["out"],
                     accumulate=True, relu=s.get("relu", False))


class _HeteroSAGELayerFn(t.autograd.Function):
    """Every relation of one hetero SAGEConv layer — `to_hetero`'s per-edge-type copies of the layer
    (model/encoder_decoder.py:29-46,93-95) — as ONE autograd node, for the case the default metadata gives: each
    destination type receives exactly one relation (so the destination-wise reduction is the identity and the relu can
    ride in the producing product).  Forward: one aggregate launch per relation + ONE grouped GEMM launch computing
    act(agg_r W_l^T + b_l + x_dst W_r^T) for all relations.  Backward: one grouped launch for all dAgg / dX_dst
    (relu mask fused into the A operand), one transposed aggregate per relation, one grouped split-K launch for all
    dW_l / dW_r / db (+ one grouped reduce).  Layer 0 reads frozen embeddings: its dX half is never launched.

    apply(rels, relu, n_x, x_0..x_{n_x-1}, (w_l, b_l, w_r) per relation); rels[i] = (src index, dst index, graph, aggr)."""

    @staticmethod
    def forward(ctx, rels, relu: bool, n_x: int, *tensors):
        xs = [x if x.stride(-1) == 1 else x.contiguous() for x in tensors[:n_x]]
        wts = tensors[n_x:]
        outs, aggs, args = hetero_layer_forward(rels, relu, xs, wts)
        ctx.rels, ctx.relu, ctx.n_x, ctx.n_rel = rels, relu, n_x, len(rels)
        ctx.save_for_backward(*xs, *wts, *aggs, *[a for a in args if a is not None], *(outs if relu else []))
        ctx.has_arg = [a is not None for a in args]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        rels, relu, n_x, n_rel = ctx.rels, ctx.relu, ctx.n_x, ctx.n_rel
        sv = list(ctx.saved_tensors)
        xs, sv = sv[:n_x], sv[n_x:]
        wts, sv = sv[:3 * n_rel], sv[3 * n_rel:]
        aggs, sv = sv[:n_rel], sv[n_rel:]
        args = []
        for h in ctx.has_arg:
            args.append(sv.pop(0) if h else None)
        outs = sv[:n_rel] if relu else [None] * n_rel
        dx, grads_w = hetero_layer_backward(rels, xs, wts, aggs, args, outs, dys, ctx.needs_input_grad[3:])
        return (None, None, None, *dx, *grads_w)


def hetero_layer_forward(rels, relu: bool, xs, wts):
    """Forward of one hetero SAGEConv layer without autograd: (outs, aggs, args) per relation.  xs: node-type tensors
    (unit inner stride), wts: (w_l, b_l, w_r) per relation, rels[i] = (src index, dst index, graph, aggr)."""
    aggs, args, outs, specs = [], [], [], []
    for i, (si, di, graph, aggr) in enumerate(rels):
        w_l, b_l, w_r = wts[3 * i: 3 * i + 3]
        x_src, x_dst = xs[si], xs[di]
        d = x_src.shape[1]
        arg = None
        if aggr == "max":
            agg, arg = ops.segment_max(graph.by_dst, x_src)
        else:
            xp = _pad4(x_src)
            v_dst, _ = graph.weights(aggr)
            a = ops.DeviceCSR(graph.by_dst.n_rows, graph.by_dst.n_cols, graph.by_dst.rowptr, graph.by_dst.col, v_dst, None,
                              graph.by_dst.plan)
            agg = t.empty(graph.n_dst, xp.shape[1], device=x_src.device)
            ops.spmm(a, xp, Y=agg)
            graph.by_dst.plan = a.plan
            if xp.shape[1] != d:
                agg = agg[:, :d]
        out = t.empty(graph.n_dst, w_l.shape[0], device=x_src.device)
        specs.append(dict(A=agg, B=w_l, out=out, A2=x_dst, B2=w_r, bias=b_l, relu=relu))
        aggs.append(agg); args.append(arg); outs.append(out)
    _run_products(specs)
    return outs, aggs, args


def hetero_layer_backward(rels, xs, wts, aggs, args, outs, dys, need):
    """Backward of hetero_layer_forward.  outs[i]: the layer's output when the relu was fused (its mask), else None.
    need: flags for (x_0..x_{n-1}, then w_l, b_l, w_r per relation).  Returns ([dx per node type], [dw_l, db_l, dw_r per relation])."""
    n_x, n_rel = len(xs), len(rels)
    dev = xs[0].device
    dys = [None if dy is None else dy.contiguous() for dy in dys]
    # ---- dX half: dAgg_r = dY W_l, dXdst_r = dY W_r for every relation in one launch
    dx = [None] * n_x
    d_aggs = [None] * n_rel
    specs = []
    dst_bufs = {}
    for i, (si, di, graph, aggr) in enumerate(rels):
        dy = dys[i]
        if dy is None:
            continue
        w_l, _, w_r = wts[3 * i: 3 * i + 3]
        if need[si]:
            d_aggs[i] = t.empty(graph.n_dst, w_l.shape[1], device=dev)
            specs.append(dict(A=dy, B=w_l, out=d_aggs[i], trans_b=False, mask=outs[i]))
        if need[di]:
            buf = t.empty(graph.n_dst, w_r.shape[1], device=dev)
            specs.append(dict(A=dy, B=w_r, out=buf, trans_b=False, mask=outs[i]))
            dst_bufs.setdefault(di, []).append(buf)
    _run_products(specs)
    for di, bufs in dst_bufs.items():   # summed only now: the buffers are filled by the launch above
        acc = bufs[0]
        for b in bufs[1:]:
            acc = acc + b
        dx[di] = acc
    for i, (si, di, graph, aggr) in enumerate(rels):
        if d_aggs[i] is None:
            continue
        d = xs[si].shape[1]
        if aggr == "max":
            g_src = ops.segment_max_bwd(graph.by_src, args[i], d_aggs[i])
            dx[si] = g_src if dx[si] is None else dx[si] + g_src
        else:
            dp = _pad4(d_aggs[i])
            _, v_src = graph.weights(aggr)
            a = ops.DeviceCSR(graph.by_src.n_rows, graph.by_src.n_cols, graph.by_src.rowptr, graph.by_src.col, v_src,
                              None, graph.by_src.plan)
            if dx[si] is not None and dp.shape[1] == d:   # A^T dAgg + the dX_dst already there, in the epilogue
                ops.spmm(a, dp, addend=dx[si], S=dx[si])
            else:
                g_src = t.empty(graph.n_src, dp.shape[1], device=dev)
                ops.spmm(a, dp, Y=g_src)
                if dp.shape[1] != d:
                    g_src = g_src[:, :d]
                dx[si] = g_src if dx[si] is None else dx[si] + g_src
            graph.by_src.plan = a.plan
    # ---- dW half: dW_l = dY^T agg, dW_r = dY^T x_dst, db = dY^T 1 for every relation in one split-K launch
    grads_w = [None] * (3 * n_rel)
    # (round 3) every relation's three transposed products as one launch of the LDS-free kernel (csrc/wgrad.hip) when all
    # of them are wanted and the shapes are the kernel's; the native executor makes the same choice, so the two stay bitwise equal
    wq = []
    for i, (si, di, graph, aggr) in enumerate(rels):
        dy = dys[i]
        w_l, b_l, w_r = wts[3 * i: 3 * i + 3]
        if (dy is None or w_r is None or not need[n_x + 3 * i] or not need[n_x + 3 * i + 2]
                or (b_l is not None and not need[n_x + 3 * i + 1])):
            wq = None
            break
        wq.append(dict(dy=dy, mask=outs[i], b1=aggs[i], b2=xs[di], gw1=t.empty_like(w_l),
                       gb=t.empty(w_l.shape[0], device=dev) if b_l is not None else None, gw2=t.empty_like(w_r)))
    if wq and len(wq) <= 4 and all(q["b1"].is_contiguous() and q["b2"].is_contiguous() and (q["mask"] is None or q["mask"].is_contiguous())
                                   for q in wq) and ops.sage_wgrad(wq):
        for i, q in enumerate(wq):
            grads_w[3 * i], grads_w[3 * i + 1], grads_w[3 * i + 2] = q["gw1"], q["gb"], q["gw2"]
        return [dx[j] if need[j] else None for j in range(n_x)], grads_w
    specs, db4 = [], {}
    for i, (si, di, graph, aggr) in enumerate(rels):
        dy = dys[i]
        if dy is None:
            continue
        w_l, b_l, w_r = wts[3 * i: 3 * i + 3]
        if need[n_x + 3 * i]:
            grads_w[3 * i] = t.empty_like(w_l)
            specs.append(dict(A=dy, B=aggs[i], out=grads_w[3 * i], trans_a=True, trans_b=False, mask=outs[i]))
        if b_l is not None and need[n_x + 3 * i + 1]:
            db4[i] = t.empty(w_l.shape[0], 4, device=dev)
            specs.append(dict(A=dy, B=_ones4(dy.shape[0], dev), out=db4[i], trans_a=True, trans_b=False, mask=outs[i]))
        if w_r is not None and need[n_x + 3 * i + 2]:
            grads_w[3 * i + 2] = t.empty_like(w_r)
            specs.append(dict(A=dy, B=xs[di], out=grads_w[3 * i + 2], trans_a=True, trans_b=False, mask=outs[i]))
    _run_products(specs)
    for i, b in db4.items():
        grads_w[3 * i + 1] = b[:, 0].contiguous()
    return [dx[j] if need[j] else None for j in range(n_x)], grads_w


def hetero_sage_layer(convs: dict, graphs: dict, x_dict: dict, relu: bool) -> Optional[dict]:
    """x_dict after one hetero layer, through _HeteroSAGELayerFn — or None when the layer is not of the fused form
    (several relations into one destination type, normalize=True, no root weight): the caller then runs the relations
    one by one.  convs: {edge type: SAGEConv}; graphs: {edge type: BipartiteGraph}."""
    types = list(x_dict)
    arriving = {}
    for et in graphs:
        arriving[et[2]] = arriving.get(et[2], 0) + 1
    if not graphs or any(v != 1 for v in arriving.values()):
        return None
    col = hetero_layer_relations(convs, graphs, x_dict)
    if col is None:
        return None
    rels, wts = col
    outs = _HeteroSAGELayerFn.apply(tuple(rels), relu, len(types), *[x_dict[k] for k in types], *wts)
    return {et[2]: o for et, o in zip(graphs, outs)}


def hetero_layer_relations(convs: dict, graphs: dict, x_dict: dict):
    """(rels, wts) of a layer in the fused form, or None (see hetero_sage_layer); sizes lazy weights from x_dict."""
    types = list(x_dict)
    arriving = {}
    for et in graphs:
        arriving[et[2]] = arriving.get(et[2], 0) + 1
    if not graphs or any(v != 1 for v in arriving.values()):
        return None
    rels, wts = [], []
    for et, graph in graphs.items():
        conv = convs[et]
        if not isinstance(conv, SAGEConv) or conv.normalize or not conv.root_weight:
            return None
        x_src, x_dst = x_dict[et[0]], x_dict[et[2]]
        if conv.lin_l.weight is None:
            conv.lin_l._materialize(int(x_src.shape[-1]), x_src.device)
        if conv.lin_r.weight is None:
            conv.lin_r._materialize(int(x_dst.shape[-1]), x_dst.device)
        rels.append((types.index(et[0]), types.index(et[2]), graph, conv.aggr))
        wts += [conv.lin_l.weight, conv.lin_l.bias, conv.lin_r.weight]
    return rels, wts


class SAGEConv(nn.Module):
    def __init__(self, in_channels: Union[int, Tuple[int, ...]], out_channels: int, aggr: str = "mean",
                 normalize: bool = False, root_weight: bool = True, bias: bool = True):
        super().__init__()
        if aggr not in ("add", "sum", "mean", "max"):
            raise ValueError(f"aggr={aggr!r} is not supported by the HIP path (add, mean, max)")
        if isinstance(in_channels, int):
            in_channels = (in_channels, in_channels)
        self.in_channels, self.out_channels = tuple(in_channels), int(out_channels)
        self.aggr = "add" if aggr == "sum" else aggr
        self.normalize, self.root_weight = normalize, root_weight
        self.lin_l = Linear(self.in_channels[0], out_channels, bias=bias)
        if root_weight:
            self.lin_r = Linear(self.in_channels[1], out_channels, bias=False)

    def reset_parameters(self) -> None:
        self.lin_l.reset_parameters()
        if self.root_weight:
            self.lin_r.reset_parameters()

    def forward(self, x, edge_index, relu: bool = False) -> Tensor:
        """x: Tensor or (x_src, x_dst).  edge_index: int64 [2, E] (row 0 = source, row 1 = destination)
        or a prebuilt BipartiteGraph."""
        x_src, x_dst = (x, x) if isinstance(x, Tensor) else x
        graph = edge_index if isinstance(edge_index, BipartiteGraph) else BipartiteGraph(
            edge_index, x_src.shape[0], x_dst.shape[0])
        if self.lin_l.weight is None:       # lazy sizes (PyG's (-1, -1, -1)): the first batch decides
            self.lin_l._materialize(int(x_src.shape[-1]), x_src.device)
        root = self.root_weight and x_dst is not None
        if root and self.lin_r.weight is None:
            self.lin_r._materialize(int(x_dst.shape[-1]), x_dst.device)
        fuse_act = relu and not self.normalize
        out = _SAGEConvFn.apply(x_src, x_dst if root else None, self.lin_l.weight, self.lin_l.bias,
                                self.lin_r.weight if root else None, graph, self.aggr, fuse_act)
        if self.normalize:
            out = F.normalize(out, p=2.0, dim=-1)
            return out.relu() if relu else out
        return out


def get_SAGEConv_layers(num_layers: int, hidden_channels: int, out_channels: int, agg_type: str) -> nn.ModuleList:
    """[(num_layers-1) x SAGEConv(-1 -> hidden)] + [SAGEConv(-1 -> out)]  (model/layers.py:6-32)."""
    single = SAGEConv((-1, -1, -1), hidden_channels, aggr=agg_type, normalize=False, bias=True)
    last = SAGEConv((-1, -1, -1), out_channels, aggr=agg_type, normalize=False, bias=True)
    if num_layers == 1:
        return nn.ModuleList([last])
    return nn.ModuleList([deepcopy(single) for _ in range(num_layers - 1)] + [last])


def get_linear_layers(num_layers: int, in_channels: int, hidden_channels: int, out_channels: int) -> nn.ModuleList:
    """Decoder stack (model/layers.py:35-56).  Modules are created in the reference's order so that a
    seeded build draws the same initial weights (tests/golden/linear_layers.pt)."""
    first = Linear(in_channels, hidden_channels)
    middle = Linear(hidden_channels, hidden_channels)
    last = Linear(hidden_channels, out_channels)
    if num_layers == 1:
        return nn.ModuleList([Linear(in_channels, out_channels)])
    if num_layers == 2:
        return nn.ModuleList([Linear(in_channels, hidden_channels), Linear(hidden_channels, out_channels)])
    return nn.ModuleList([first] + [deepcopy(middle) for _ in range(num_layers - 2)] + [last])

"""One ranker training iteration as ONE C call (include/laplace_hip.h: mi_ranker_step_f32, csrc/ranker_exec.hip).

`ranker_step.FusedRankerStep` already runs the iteration of the reference's training.py:19-34 as straight-line code —
but still as ~75 op calls from Python, and at the reference's batch size (24 users) each of them costs more on the host
than its kernel does on the GPU.  `NativeRankerStep` hands the C executor a description of the model (parameter,
gradient and Adam-state pointers, built once) and of the batch (feature tensors, the two sorted CSRs the device sampler
emits, the label edges) and the executor enqueues the same launches in the same order itself, dropout (Philox) and the
multi-tensor Adam update included: no at::native kernel is left in the loop.

The model, its parameters / buffers / state_dict and the optimizer object stay torch's own: gradients land in
`param.grad`, Adam's moments in `optimizer.state[p]["exp_avg" / "exp_avg_sq"]`, its step count in `state[p]["step"]`.
Shapes the executor does not take (MI_ERR_UNSUPPORTED, nothing enqueued) fall to FusedRankerStep / autograd.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, Optional

import torch as t
from torch import Tensor

from . import _lib, ops
from ._lib import RankerBatch, RankerModel
from .model.encoder_decoder import Encoder_Decoder_Model, _key
from .model.layers import Linear, SAGEConv, _ones4
from .utils.constants import Constants


class NativeRankerStep:
    def __init__(self, model: Encoder_Decoder_Model, optimizer: t.optim.Optimizer, before_step=None, seed: Optional[int] = None,
                 data_parallel: bool = False, group=None):
        """before_step: called once the gradients are in place and BEFORE the update; the executor then stops after the
        gradients and `optimizer.step()` applies them (the generic hook).
        data_parallel: the native data-parallel iteration (ranker half) — the executor stops after the
        gradients, which live in ONE flat buffer (`flat_grads`; every `param.grad` is a view into it), that buffer is
        all-reduced (sum) over `group` in one collective, and mi_ranker_adam_f32 applies mean gradient + Adam in one
        launch.  Three host calls per iteration instead of cat / all-reduce / split / foreach-Adam."""
        why = self.unsupported_reason(model, optimizer)
        if why:
            raise ValueError(f"NativeRankerStep: {why}")
        if data_parallel and before_step is not None:
            raise ValueError("NativeRankerStep: data_parallel and before_step are two ways of doing the same thing; pass one")
        self.model, self.optimizer, self.before_step = model, optimizer, before_step
        self.data_parallel, self.group = bool(data_parallel), group
        self.flat_grads: Optional[Tensor] = None
        self.seed = int(t.initial_seed() if seed is None else seed) & ((1 << 64) - 1)
        self.iteration = 0
        self._desc: Optional[RankerModel] = None
        self._snapshot = None      # the pointers the descriptor was built from, as Python ints (see _params_current)
        self._keep = []            # tensors the descriptor points into
        self._ws: Optional[Tensor] = None
        self._ws_dims = None       # the largest (customers, articles, edges, label edges) the workspace was sized for
        self._adam_step = 0
        self.declined: Optional[str] = None
        # a second stream for the customer-side twins of independent launch pairs (mi_ranker_batch.aux_stream).  OFF by
        # default: measured at the H&M shape, 24 users per batch (round 3): 0.775 ms per iteration with it against 0.750
        # without — eight fork / join event pairs per iteration cost more than running 5-20 us kernels side by side returns.
        # LAPLACE_RANKER_AUX=1 switches it on (results are identical either way: tests/test_gpu_ranker.py).
        self._aux = None
        if os.environ.get("LAPLACE_RANKER_AUX", "0") == "1":
            dev = next(model.parameters()).device
            with t.cuda.device(dev):
                st = t.cuda.Stream(device=dev)
                ev = (t.cuda.Event(enable_timing=False), t.cuda.Event(enable_timing=False))
                for e in ev:
                    e.record(st)        # events are created lazily: materialise the handles
            self._aux = (st, ev[0], ev[1])

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def unsupported_reason(model, optimizer) -> Optional[str]:
        why = NativeRankerStep.model_unsupported_reason(model)
        if why:
            return why
        if type(optimizer) is not t.optim.Adam or len(optimizer.param_groups) != 1:
            return "optimizer other than a single-group torch.optim.Adam"
        g = optimizer.param_groups[0]
        if g.get("amsgrad") or g.get("weight_decay", 0) or g.get("maximize") or g.get("capturable") or g.get("differentiable"):
            return "Adam options (amsgrad / weight_decay / maximize / capturable)"
        if isinstance(g["lr"], Tensor):
            return "tensor learning rate"
        if len(g["params"]) > _lib.MI_RANKER_MAX_PARAMS or any(p.dtype != t.float32 or not p.is_cuda for p in g["params"]):
            return "parameter list"
        return None

    @staticmethod
    def model_unsupported_reason(model) -> Optional[str]:
        """The model half of unsupported_reason (NativeRankerForward has no optimizer)."""
        if not isinstance(model, Encoder_Decoder_Model):
            return "not an Encoder_Decoder_Model"
        enc, dec = model.encoder, model.decoder
        if not model.embedding:
            return "dense (non-categorical) node features"
        if enc.node_types != [Constants.node_user, Constants.node_item]:
            return "node types other than [customer, article]"
        if [tuple(e) for e in enc.edge_types] != [tuple(Constants.edge_key), tuple(Constants.rev_edge_key)]:
            return "edge types other than buys / rev_buys"
        if not (1 <= len(enc.layers) <= _lib.MI_RANKER_MAX_LAYERS and 1 <= len(dec.layers) <= _lib.MI_RANKER_MAX_LAYERS):
            return "layer count"
        aggrs = set()
        for convs in enc.layers:
            for conv in convs.values():
                if not isinstance(conv, SAGEConv) or conv.normalize or not conv.root_weight or conv.lin_l.weight is None:
                    return "a conv that is not a materialised SAGEConv with root weight"
                aggrs.add("add" if conv.aggr in ("add", "sum") else conv.aggr)
        if len(aggrs) != 1 or next(iter(aggrs)) not in ("add", "mean"):
            return "aggregation other than add / mean"
        if not all(isinstance(l, Linear) and l.weight is not None for l in dec.layers):
            return "a decoder layer that is not a materialised Linear"
        for bn in (model.encoder_layer_norm_customer, model.encoder_layer_norm_article):
            if model.batch_normalize and (bn.momentum is None or (bn.weight is None) != (bn.bias is None)):
                return "BatchNorm with cumulative momentum"
        for tables in model.embedding_layers.values():
            if len(tables) > _lib.MI_RANKER_MAX_COLS:
                return "more categorical columns than the executor takes"
        return None

    @classmethod
    def supports(cls, model, optimizer) -> bool:
        return cls.unsupported_reason(model, optimizer) is None

    # ------------------------------------------------------------------------------------------
    def _build(self) -> RankerModel:
        model, opt = self.model, self.optimizer
        enc, dec = model.encoder, model.decoder
        d = RankerModel()
        keep = self._keep = []

        # every gradient is a view into one flat buffer (16-byte aligned pieces): a data-parallel caller exchanges it in
        # one collective; parameters that are not the optimizer's keep an allocation of their own
        plist = opt.param_groups[0]["params"]
        offs, total = {}, 0
        for p in plist:
            offs[id(p)] = total
            total += (p.numel() + 3) // 4 * 4
        flat = self.flat_grads
        ok = flat is not None and flat.numel() == total and all(
            p.grad is not None and p.grad.data_ptr() == flat.data_ptr() + 4 * offs[id(p)] for p in plist)
        if not ok and plist:
            flat = t.zeros(total, dtype=t.float32, device=plist[0].device)
            for p in plist:
                view = flat[offs[id(p)]: offs[id(p)] + p.numel()].view(p.shape)
                if p.grad is not None and p.grad.shape == p.shape:
                    view.copy_(p.grad)
                p.grad = view
            self.flat_grads = flat
        keep.append(flat)

        def grad_of(p: Tensor) -> int:
            if p.grad is None or p.grad.shape != p.shape or not p.grad.is_contiguous():
                p.grad = t.zeros_like(p)
            keep.append(p.grad)
            return p.grad.data_ptr()

        d.n_enc_layers, d.n_dec_layers = len(enc.layers), len(dec.layers)
        first = next(iter(enc.layers[0].values()))
        d.aggr = 1 if first.aggr == "mean" else 0
        d.batch_normalize = 1 if model.batch_normalize else 0
        d.max_norm = 1.0
        for ti, key in enumerate((Constants.node_user, Constants.node_item)):
            tables = model.embedding_layers[key]
            d.n_cols[ti] = len(tables)
            for c, tb in enumerate(tables):
                if not tb.is_contiguous():
                    raise ValueError("embedding tables must be contiguous")
                d.tables[ti][c] = tb.data_ptr()
                d.table_rows[ti][c] = int(tb.shape[0])
                d.dims[ti][c] = int(tb.shape[1])
        for l, convs in enumerate(enc.layers):
            for r, et in enumerate((Constants.edge_key, Constants.rev_edge_key)):
                conv = convs[_key(tuple(et))]
                cv = d.conv[l][r]
                cv.w_l, cv.w_r = conv.lin_l.weight.data_ptr(), conv.lin_r.weight.data_ptr()
                cv.gw_l, cv.gw_r = grad_of(conv.lin_l.weight), grad_of(conv.lin_r.weight)
                if conv.lin_l.bias is not None:
                    cv.b_l, cv.gb_l = conv.lin_l.bias.data_ptr(), grad_of(conv.lin_l.bias)
                cv.c_out, cv.c_src = (int(x) for x in conv.lin_l.weight.shape)
                cv.c_dst = int(conv.lin_r.weight.shape[1])
        for ti, bn in enumerate((model.encoder_layer_norm_customer, model.encoder_layer_norm_article)):
            nm = d.norm[ti]
            if bn.weight is not None:
                nm.gamma, nm.beta = bn.weight.data_ptr(), bn.bias.data_ptr()
                nm.g_gamma, nm.g_beta = grad_of(bn.weight), grad_of(bn.bias)
            if bn.track_running_stats:
                nm.running_mean, nm.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
                nm.num_batches_tracked = bn.num_batches_tracked.data_ptr()
            nm.momentum, nm.eps = float(bn.momentum if bn.momentum is not None else 0.1), float(bn.eps)
        for j, layer in enumerate(dec.layers):
            ln = d.dec[j]
            ln.w, ln.gw = layer.weight.data_ptr(), grad_of(layer.weight)
            if layer.bias is not None:
                ln.b, ln.gb = layer.bias.data_ptr(), grad_of(layer.bias)
            ln.out, ln.in_ = (int(x) for x in layer.weight.shape)
        # the optimizer's parameter list with its state (created the way torch.optim.Adam creates it on its first step)
        group = opt.param_groups[0]
        d.n_params = len(group["params"])
        for i, p in enumerate(group["params"]):
            st = opt.state[p]
            if len(st) == 0:
                on_device = bool(group.get("fused") or group.get("capturable"))
                st["step"] = t.zeros((), dtype=t.float32, device=p.device) if on_device else t.tensor(0.0, dtype=t.float32)
                st["exp_avg"] = t.zeros_like(p, memory_format=t.preserve_format)
                st["exp_avg_sq"] = t.zeros_like(p, memory_format=t.preserve_format)
            q = d.params[i]
            q.p, q.g, q.m, q.v, q.n = p.data_ptr(), grad_of(p), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()
        steps = [opt.state[p]["step"] for p in group["params"]]
        self._adam_step = int(steps[0]) if steps else 0     # one read-back when the steps live on the device (fused=True)
        return d

    def _params_current(self, d: RankerModel) -> bool:
        """The descriptor holds raw pointers: rebuilt when a parameter, gradient, optimizer-state, BatchNorm buffer or
        embedding table was replaced (zero_grad(set_to_none=True), model.to(...), load of a new optimizer state, ...).
        Compared against plain Python ints captured at build time: reading the pointers back out of the ctypes descriptor
        (four fields x ~25 tensors) was 30 us of a host-bound 0.47 ms iteration (round 4, tools/prof_host_native.py)."""
        snap = self._snapshot
        if snap is None:
            return False
        params = self.optimizer.param_groups[0]["params"]
        if len(params) != len(snap[0]):
            return False
        state = self.optimizer.state
        for p, (q, pp, gp, mp, vp) in zip(params, snap[0]):
            g = p.grad
            st = state.get(p)
            if (p is not q or g is None or not st or p.data_ptr() != pp or g.data_ptr() != gp or st["exp_avg"].data_ptr() != mp
                    or st["exp_avg_sq"].data_ptr() != vp):
                return False
        for tensor_of, ptr in snap[1]:
            if tensor_of().data_ptr() != ptr:
                return False
        return True

    def _take_snapshot(self) -> None:
        params = self.optimizer.param_groups[0]["params"]
        state = self.optimizer.state
        rows = [(p, p.data_ptr(), p.grad.data_ptr(), state[p]["exp_avg"].data_ptr(), state[p]["exp_avg_sq"].data_ptr()) for p in params]
        model = self.model
        others = []
        for key, name in ((Constants.node_user, "encoder_layer_norm_customer"), (Constants.node_item, "encoder_layer_norm_article")):
            bn = getattr(model, name)
            if bn.track_running_stats:
                others.append((lambda bn=bn: bn.running_mean, bn.running_mean.data_ptr()))
            tables = model.embedding_layers[key]
            for c in range(len(tables)):
                others.append((lambda tables=tables, c=c: tables[c], tables[c].data_ptr()))
        self._snapshot = (rows, others)

    # ------------------------------------------------------------------------------------------
    def _prepare(self, x_dict: Dict[str, Tensor], edge_index_dict: dict, edge_label_index: Tensor, labels: Tensor):
        """Everything up to (not including) the launch: descriptors built, workspace sized, the executor's own validation
        pass run (mi_ranker_step_check).  Returns (d, b, loss, steps, keep-alive) or None with self.declined set; nothing has
        been enqueued either way."""
        model = self.model
        if not model.training:
            self.declined = "model in eval mode"
            return None
        if model.batch_normalize and not (model.encoder_layer_norm_customer.training and model.encoder_layer_norm_article.training):
            self.declined = "BatchNorm layers in eval mode"
            return None
        ei = None
        for k, v in edge_index_dict.items():
            if tuple(k) == tuple(Constants.edge_key):
                ei = v
        if ei is None or edge_label_index.dtype != t.int64:
            self.declined = "no buys relation / label index not int64"
            return None
        if labels.dtype not in (t.int64, t.float32):
            self.declined = "labels are neither int64 nor float32"
            return None
        xc, xa = x_dict[Constants.node_user], x_dict[Constants.node_item]
        if xc.dtype != t.int64 or xa.dtype != t.int64:
            self.declined = "features are not int64 categorical columns"
            return None
        n_c, n_a = int(xc.shape[0]), int(xa.shape[0])
        pre = getattr(ei, "_sorted_csr", None)
        if pre is not None and pre[0].n_rows == n_c and pre[1].n_rows == n_a:
            by_c, by_a = pre
        else:  # a batch from a host loader: the two sorts the device sampler would have done
            src, dst = ei[0].contiguous(), ei[1].contiguous()
            by_c = ops.coo_to_csr(src, dst, n_c, n_a, want_perm=False)
            by_a = ops.coo_to_csr(dst, src, n_a, n_c, want_perm=False)
        if self._desc is None or not self._params_current(self._desc):
            self._desc = self._build()
            self._take_snapshot()
        d = self._desc
        p = model.encoder.p_dropout_features
        d.p_dropout = float(p) if p else 0.0
        group = self.optimizer.param_groups[0]
        d.lr, (d.beta1, d.beta2), d.eps = float(group["lr"]), (float(b) for b in group["betas"]), float(group["eps"])
        d.apply_adam = 0 if (self.before_step is not None or self.data_parallel) else 1
        steps = [self.optimizer.state[q]["step"] for q in group["params"]]
        d.step = self._adam_step + 1
        xc, xa = xc.contiguous(), xa.contiguous()
        row, col = edge_label_index[0].contiguous(), edge_label_index[1].contiguous()
        labels = labels.contiguous()
        n_lab = int(labels.numel())
        ones = _ones4(max(n_c, n_a, n_lab), xc.device)
        d.ones4, d.n_ones = ones.data_ptr(), int(ones.shape[0])
        b = RankerBatch()
        b.n_nodes[0], b.n_nodes[1] = n_c, n_a
        b.x[0], b.x[1] = xc.data_ptr(), xa.data_ptr()
        b.by_customer_ptr, b.by_customer_col = by_c.rowptr.data_ptr(), (by_c.col.data_ptr() if by_c.nnz else by_c.rowptr.data_ptr())
        b.by_article_ptr, b.by_article_col = by_a.rowptr.data_ptr(), (by_a.col.data_ptr() if by_a.nnz else by_a.rowptr.data_ptr())
        b.nnz, b.n_label = by_c.nnz, n_lab
        b.label_row, b.label_col = row.data_ptr(), col.data_ptr()
        if labels.dtype == t.int64:     # the sampler's edge_label as it is: cast inside the executor
            b.label = labels.data_ptr()
        else:
            b.label_f32 = labels.data_ptr()
        b.seed, b.step = self.seed, self.iteration
        loss = t.empty(1, dtype=t.float32, device=xc.device)
        b.loss = loss.data_ptr()
        if self._aux is not None:
            b.aux_stream, b.ev_fork, b.ev_join = self._aux[0].cuda_stream, self._aux[1].cuda_event, self._aux[2].cuda_event
        L = _lib.lib()
        dims = (n_c, n_a, int(by_c.nnz), n_lab, d.p_dropout > 0.0)
        sized = self._ws_dims      # (dims of the largest batch the counting pass has seen, its byte count)
        if self._ws is None or sized is None or dims[4] != sized[0][4] or any(x > y for x, y in zip(dims[:4], sized[0][:4])):
            # the workspace is a sum of arrays proportional to these four counts, so a batch no larger in ANY of them than one the
            # counting pass has sized needs no more than that one: the pass (a host walk of the whole iteration) is skipped for it
            need = int(L.mi_ranker_step_workspace_bytes(ctypes.byref(d), ctypes.byref(b)))
            if self._ws is None or self._ws.numel() < need:
                self._ws = t.empty(int(need * 1.25) + (1 << 20), dtype=t.uint8, device=xc.device)
            if sized is None or dims[4] != sized[0][4] or need >= sized[1]:
                self._ws_dims = (dims, need)
        if self._world() > 1:    # the validation pass on its own only where the ranks must agree BEFORE anything is enqueued;
            #                      a single process lets mi_ranker_step_f32 run it (the loop is host-bound: every call counts)
            rc = L.mi_ranker_step_check(ctypes.byref(d), ctypes.byref(b), self._ws.data_ptr(), self._ws.numel())
            if rc == _lib.MI_ERR_UNSUPPORTED:
                self.declined = "mi_ranker_step_f32: MI_ERR_UNSUPPORTED (shape outside the executor's)"
                return None
            _lib.check(rc, "mi_ranker_step_check")
        return d, b, loss, steps, (xc, xa, row, col, labels, ones, by_c, by_a)

    def _world(self) -> int:
        import torch.distributed as dist
        return dist.get_world_size(self.group) if (self.data_parallel and dist.is_initialized()) else 1

    def _all_ranks_take_it(self, mine: bool, device) -> bool:
        """Data-parallel only: the decline is COLLECTIVE.  A rank whose batch lies outside the executor's shapes must not
        leave its peers alone in the gradient all-reduce (they would wait for the collective's timeout, or reduce against
        the fallback path's differently sized buffer): one all-reduce(MIN) of a 1-int flag BEFORE anything is enqueued, and
        every rank takes the same branch."""
        import torch.distributed as dist
        flag = t.tensor([1 if mine else 0], dtype=t.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(flag.item()))

    def step(self, x_dict: Dict[str, Tensor], edge_index_dict: dict, edge_label_index: Tensor, labels: Tensor) -> Optional[Tensor]:
        """One iteration; the loss as a 1-element device tensor, or None when the executor declines this batch (nothing
        has been enqueued then; the caller runs FusedRankerStep / autograd).  With data_parallel=True and more than one
        rank the decision is collective: every rank returns None when ANY rank's batch is declined, so all of them take the
        caller's fallback together (`declined` then names the local reason, or says that a peer declined)."""
        self.declined = None       # why the last call returned None (diagnostics)
        prep = self._prepare(x_dict, edge_index_dict, edge_label_index, labels)
        if self._world() > 1:
            if not self._all_ranks_take_it(prep is not None, next(self.model.parameters()).device):
                if prep is not None:
                    self.declined = "a peer rank declined its batch (collective decision: every rank takes the fallback)"
                return None
        elif prep is None:
            return None
        d, b, loss, steps, _keep = prep
        group = self.optimizer.param_groups[0]
        L = _lib.lib()
        rc = L.mi_ranker_step_f32(ctypes.byref(d), ctypes.byref(b), self._ws.data_ptr(), self._ws.numel(), _lib.current_stream())
        if rc == _lib.MI_ERR_WORKSPACE:        # (found by the validation pass: nothing enqueued) the skipped counting pass after all
            need = int(L.mi_ranker_step_workspace_bytes(ctypes.byref(d), ctypes.byref(b)))
            self._ws = t.empty(int(need * 1.25) + (1 << 20), dtype=t.uint8, device=self._ws.device)
            self._ws_dims = None
            rc = L.mi_ranker_step_f32(ctypes.byref(d), ctypes.byref(b), self._ws.data_ptr(), self._ws.numel(), _lib.current_stream())
        if rc == _lib.MI_ERR_UNSUPPORTED:
            if self._world() > 1:              # cannot happen: mi_ranker_step_check took the same descriptors
                raise _lib.MiError("mi_ranker_step_f32 declined a batch its own validation pass had accepted")
            self.declined = "mi_ranker_step_f32: MI_ERR_UNSUPPORTED (shape outside the executor's)"
            return None
        _lib.check(rc, "mi_ranker_step_f32")
        self.iteration += 1
        if self.before_step is not None:       # gradients are in param.grad: exchange them, then torch's own update
            self.before_step()
            self.optimizer.step()
            self._adam_step = int(steps[0]) if steps and not steps[0].is_cuda else self._adam_step + 1
            return loss
        if self.data_parallel:
            import torch.distributed as dist
            world = dist.get_world_size(self.group) if dist.is_initialized() else 1
            if world > 1:
                dist.all_reduce(self.flat_grads, op=dist.ReduceOp.SUM, group=self.group)
            _lib.check(L.mi_ranker_adam_f32(ctypes.byref(d), 1.0 / world, _lib.current_stream()), "mi_ranker_adam_f32")
        self._adam_step += 1
        if steps and steps[0].is_cuda:      # fused=True keeps its step counts on the device: one foreach launch
            t._foreach_add_(steps, 1)
        else:
            for s in steps:                 # the default Adam's host scalars
                s += 1
        return loss


class NativeRankerForward:
    """model(x_dict, edge_index_dict, edge_label_index) under model.eval() as ONE C call (mi_ranker_step_f32 with
    mi_ranker_batch.logits set: the executor's forward launches only, no dropout, BatchNorm with its running statistics).
    run_submission.make_predictions' loop is bound by the Python side of the op-by-op forward (0.69 of a 1.2 ms batch of 128
    customers, tools/prof_inference.py); this is that forward without it.  `logits(...)` returns the decoder's output per label
    edge as float32[n_label], or None when the executor does not take the model / batch (the caller runs the model itself)."""

    def __init__(self, model: Encoder_Decoder_Model):
        why = NativeRankerStep.model_unsupported_reason(model)
        if not why and model.batch_normalize:
            for bn in (model.encoder_layer_norm_customer, model.encoder_layer_norm_article):
                if not bn.track_running_stats or bn.running_mean is None:
                    why = "BatchNorm without running statistics"
        if not why and not all(p.is_cuda and p.dtype == t.float32 for p in model.parameters()):
            why = "parameters not float32 on the GPU"
        if why:
            raise ValueError(f"NativeRankerForward: {why}")
        self.model = model
        self._desc: Optional[RankerModel] = None
        self._ptrs = None
        self._ws: Optional[Tensor] = None
        self.declined: Optional[str] = None

    @classmethod
    def supports(cls, model) -> bool:
        try:
            cls(model)
            return True
        except ValueError:
            return False

    def _tensors(self):
        """Every tensor the descriptor points into, in a fixed order (its pointers are the descriptor's identity)."""
        model = self.model
        out = []
        for key in (Constants.node_user, Constants.node_item):
            out += list(model.embedding_layers[key])
        for convs in model.encoder.layers:
            for et in (Constants.edge_key, Constants.rev_edge_key):
                conv = convs[_key(tuple(et))]
                out += [conv.lin_l.weight, conv.lin_r.weight] + ([conv.lin_l.bias] if conv.lin_l.bias is not None else [])
        for bn in (model.encoder_layer_norm_customer, model.encoder_layer_norm_article):
            out += [x for x in (bn.weight, bn.bias, bn.running_mean, bn.running_var) if x is not None]
        for layer in model.decoder.layers:
            out += [layer.weight] + ([layer.bias] if layer.bias is not None else [])
        return out

    def _build(self) -> RankerModel:
        model = self.model
        enc, dec = model.encoder, model.decoder
        d = RankerModel()
        d.n_enc_layers, d.n_dec_layers = len(enc.layers), len(dec.layers)
        first = next(iter(enc.layers[0].values()))
        d.aggr = 1 if first.aggr == "mean" else 0
        d.batch_normalize = 1 if model.batch_normalize else 0
        d.max_norm = 1.0
        for ti, key in enumerate((Constants.node_user, Constants.node_item)):
            tables = model.embedding_layers[key]
            d.n_cols[ti] = len(tables)
            for c, tb in enumerate(tables):
                if not tb.is_contiguous():
                    raise ValueError("embedding tables must be contiguous")
                d.tables[ti][c], d.table_rows[ti][c], d.dims[ti][c] = tb.data_ptr(), int(tb.shape[0]), int(tb.shape[1])
        for l, convs in enumerate(enc.layers):
            for r, et in enumerate((Constants.edge_key, Constants.rev_edge_key)):
                conv = convs[_key(tuple(et))]
                cv = d.conv[l][r]
                cv.w_l, cv.w_r = conv.lin_l.weight.data_ptr(), conv.lin_r.weight.data_ptr()
                if conv.lin_l.bias is not None:
                    cv.b_l = conv.lin_l.bias.data_ptr()
                cv.c_out, cv.c_src = (int(x) for x in conv.lin_l.weight.shape)
                cv.c_dst = int(conv.lin_r.weight.shape[1])
        for ti, bn in enumerate((model.encoder_layer_norm_customer, model.encoder_layer_norm_article)):
            nm = d.norm[ti]
            if bn.weight is not None:
                nm.gamma, nm.beta = bn.weight.data_ptr(), bn.bias.data_ptr()
            if bn.track_running_stats and bn.running_mean is not None:
                nm.running_mean, nm.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
            nm.momentum, nm.eps = float(bn.momentum if bn.momentum is not None else 0.1), float(bn.eps)
        for j, layer in enumerate(dec.layers):
            ln = d.dec[j]
            ln.w = layer.weight.data_ptr()
            if layer.bias is not None:
                ln.b = layer.bias.data_ptr()
            ln.out, ln.in_ = (int(x) for x in layer.weight.shape)
        d.n_params = 0
        d.p_dropout = 0.0
        return d

    def logits(self, x_dict: Dict[str, Tensor], edge_index_dict: dict, edge_label_index: Tensor) -> Optional[Tensor]:
        self.declined = None
        ei = None
        for k, v in edge_index_dict.items():
            if tuple(k) == tuple(Constants.edge_key):
                ei = v
        xc, xa = x_dict.get(Constants.node_user), x_dict.get(Constants.node_item)
        if ei is None or xc is None or xa is None or edge_label_index.dtype != t.int64 or xc.dtype != t.int64 or xa.dtype != t.int64:
            self.declined = "no buys relation, or features / label index not int64"
            return None
        if not xc.is_cuda:
            self.declined = "batch not on the GPU"
            return None
        ptrs = tuple(x.data_ptr() for x in self._tensors())
        if self._desc is None or ptrs != self._ptrs:     # a table / weight / buffer was replaced (load_state_dict keeps storages; .to() does not)
            self._desc, self._ptrs = self._build(), ptrs
        d = self._desc
        n_c, n_a = int(xc.shape[0]), int(xa.shape[0])
        pre = getattr(ei, "_sorted_csr", None)
        if pre is not None and pre[0].n_rows == n_c and pre[1].n_rows == n_a:
            by_c, by_a = pre
        else:
            src, dst = ei[0].contiguous(), ei[1].contiguous()
            by_c = ops.coo_to_csr(src, dst, n_c, n_a, want_perm=False)
            by_a = ops.coo_to_csr(dst, src, n_a, n_c, want_perm=False)
        xc, xa = xc.contiguous(), xa.contiguous()
        row, col = edge_label_index[0].contiguous(), edge_label_index[1].contiguous()
        n_lab = int(row.numel())
        if n_lab == 0:
            return t.empty(0, dtype=t.float32, device=xc.device)
        ones = _ones4(max(n_c, n_a, n_lab), xc.device)
        d.ones4, d.n_ones = ones.data_ptr(), int(ones.shape[0])
        out = t.empty(n_lab, dtype=t.float32, device=xc.device)
        b = RankerBatch()
        b.n_nodes[0], b.n_nodes[1] = n_c, n_a
        b.x[0], b.x[1] = xc.data_ptr(), xa.data_ptr()
        b.by_customer_ptr, b.by_customer_col = by_c.rowptr.data_ptr(), (by_c.col.data_ptr() if by_c.nnz else by_c.rowptr.data_ptr())
        b.by_article_ptr, b.by_article_col = by_a.rowptr.data_ptr(), (by_a.col.data_ptr() if by_a.nnz else by_a.rowptr.data_ptr())
        b.nnz, b.n_label = by_c.nnz, n_lab
        b.label_row, b.label_col = row.data_ptr(), col.data_ptr()
        b.logits = out.data_ptr()
        L = _lib.lib()
        for attempt in (0, 1):
            if self._ws is None or attempt == 1:
                need = int(L.mi_ranker_step_workspace_bytes(ctypes.byref(d), ctypes.byref(b)))
                if need == 0:
                    self.declined = "mi_ranker_step_f32: shape outside the executor's"
                    return None
                self._ws = t.empty(int(need * 1.25) + (1 << 20), dtype=t.uint8, device=xc.device)
            rc = L.mi_ranker_step_f32(ctypes.byref(d), ctypes.byref(b), self._ws.data_ptr(), self._ws.numel(), _lib.current_stream())
            if rc != _lib.MI_ERR_WORKSPACE:
                break
        if rc == _lib.MI_ERR_UNSUPPORTED:
            self.declined = "mi_ranker_step_f32: MI_ERR_UNSUPPORTED (shape outside the executor's)"
            return None
        _lib.check(rc, "mi_ranker_step_f32 (inference)")
        self._keep = (xc, xa, row, col, ones, by_c, by_a)   # until the next call: the launches may still be reading them
        return out

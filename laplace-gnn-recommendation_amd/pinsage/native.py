"""One PinSAGE training iteration as one C call (mi_pinsage_step_f32, csrc/pinsage_exec.hip) — the loop body of the
reference's pinsage/model.py:118-131 on PinSAGEModel (pinsage/model.py:16-34, pinsage/layers.py:121-203).

The model, its parameters and the optimizer stay torch's own: gradients land in `param.grad`, Adam's moments in
`optimizer.state[p]`.  Two dense gradient buffers (the projector table's and the scorer bias's) are kept all-zero between
iterations by the executor itself: it writes the rows of the batch, runs torch.optim.Adam's dense update over the whole
tables and clears the rows again, so after a step those two `.grad`s read zero (keep_grads=True stops after the gradients
and leaves them in place instead).  Batches must come from PinSAGESampler (its block layout: destination nodes first; the
index-op path's blocks get their CSRs built here); a model / optimizer / batch outside the executor's shapes is declined
and the caller takes the autograd path (pinsage.model.train_epoch does that by itself).
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch as t
from torch import Tensor

from .. import _lib
from .._lib import PinsageModel, PinsageStepBatch
from ..model.layers import _ones4
from .model import PinSAGEModel


class NativePinSAGEStep:
    def __init__(self, model: PinSAGEModel, optimizer: t.optim.Optimizer, seed: Optional[int] = None, keep_grads: bool = False,
                 data_parallel: bool = False, group=None):
        """data_parallel (BASELINE configs[4]: 4 GPUs): every rank runs the executor on its own batch with the projector / bias
        gradients written COMPACTLY (the rows of the batch), the ranks all-gather those lists (a few hundred KB instead of an
        all-reduce of the dense 27 MB table gradient), all-reduce the dense layers' gradients (one flat buffer), and
        mi_pinsage_apply_f32 adds every rank's rows in rank order, applies Adam on the mean gradient and clears the rows:
        replicas stay bitwise identical.  Batches must keep the sampler's size bounds (they size the exchange buffer)."""
        why = self.unsupported_reason(model, optimizer)
        if why:
            raise ValueError(f"NativePinSAGEStep: {why}")
        if data_parallel and keep_grads:
            raise ValueError("NativePinSAGEStep: keep_grads is a single-process probe")
        self.model, self.optimizer, self.keep_grads = model, optimizer, bool(keep_grads)
        self.data_parallel, self.group = bool(data_parallel), group
        self._flat_small: Optional[Tensor] = None     # data-parallel: the dense layers' gradients, one allocation
        self._xbuf = None                              # data-parallel: (send int32 buffer, gathered buffer, capacity in rows)
        self.build_csrs = True     # blocks that come without their CSRs (the sampler's index-op path) get them here
        self.seed = int(t.initial_seed() if seed is None else seed) & ((1 << 64) - 1)
        self.iteration = 0
        self._desc: Optional[PinsageModel] = None
        self._keep: list = []
        self._ws: Optional[Tensor] = None
        self._adam_step = 0
        self.declined: Optional[str] = None

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def unsupported_reason(model, optimizer) -> Optional[str]:
        if not isinstance(model, PinSAGEModel):
            return "not a PinSAGEModel"
        if type(optimizer) is not t.optim.Adam or len(optimizer.param_groups) != 1:
            return "optimizer is not a single-group torch.optim.Adam"
        g = optimizer.param_groups[0]
        if g.get("amsgrad") or g.get("weight_decay", 0) or g.get("maximize") or g.get("capturable") or g.get("differentiable"):
            return "Adam options (amsgrad / weight_decay / maximize / capturable)"
        params = list(model.parameters())
        if len(g["params"]) != len(params) or any(a is not b for a, b in zip(g["params"], params)):
            return "the optimizer's parameter list is not model.parameters()"
        hidden = model.proj.weight.shape[1]
        if hidden % 4 or hidden > 128 or not (1 <= len(model.convs) <= _lib.MI_PINSAGE_MAX_LAYERS):
            return "hidden size / layer count outside the executor's"
        if len(params) - 1 > _lib.MI_PINSAGE_MAX_PARAMS:
            return "too many parameter tensors"
        for cv in model.convs:
            if tuple(cv.Q.weight.shape) != (hidden, hidden) or tuple(cv.W.weight.shape) != (hidden, 2 * hidden):
                return "layer widths differ from the hidden size"
            if cv.Q.bias is None or cv.W.bias is None:
                return "a layer without bias"
            if cv.dropout.p != model.convs[0].dropout.p:
                return "layers with different dropout rates"
        if any(p.dtype != t.float32 or not p.is_cuda or not p.is_contiguous() or not p.requires_grad for p in params):
            return "parameters are not contiguous float32 CUDA tensors with requires_grad"
        return None

    @classmethod
    def supports(cls, model, optimizer) -> bool:
        return cls.unsupported_reason(model, optimizer) is None

    # ------------------------------------------------------------------------------------------
    def _build(self) -> PinsageModel:
        model, opt = self.model, self.optimizer
        d = PinsageModel()
        keep = self._keep = []
        group = opt.param_groups[0]
        if self.data_parallel:   # every gradient except the two dense tables': views of one flat buffer (one all-reduce)
            small = [p for p in group["params"] if p is not model.proj.weight and p is not model.bias]
            offs, total = [], 0
            for p in small:
                offs.append(total)
                total += (p.numel() + 3) // 4 * 4
            flat = self._flat_small
            if flat is None or flat.numel() != total or any(
                    p.grad is None or p.grad.data_ptr() != flat.data_ptr() + 4 * o for p, o in zip(small, offs)):
                flat = t.zeros(total, dtype=t.float32, device=model.proj.weight.device)
                for p, o in zip(small, offs):
                    p.grad = flat[o: o + p.numel()].view(p.shape)
                self._flat_small = flat
            keep.append(flat)
        for p in group["params"]:
            if p.grad is None or p.grad.shape != p.shape or not p.grad.is_contiguous():
                p.grad = t.zeros_like(p)
            st = opt.state[p]
            if len(st) == 0:   # created the way torch.optim.Adam creates it on its first step
                on_device = bool(group.get("fused") or group.get("capturable"))
                st["step"] = t.zeros((), dtype=t.float32, device=p.device) if on_device else t.tensor(0.0, dtype=t.float32)
                st["exp_avg"] = t.zeros_like(p, memory_format=t.preserve_format)
                st["exp_avg_sq"] = t.zeros_like(p, memory_format=t.preserve_format)
            keep += [p.grad, st["exp_avg"], st["exp_avg_sq"]]
        proj, bias = model.proj.weight, model.bias
        proj.grad.zero_()      # the two dense buffers the executor keeps all-zero between iterations
        bias.grad.zero_()
        d.n_layers, d.hidden, d.n_items = len(model.convs), int(proj.shape[1]), int(bias.shape[0])
        if proj.shape[0] != d.n_items + 1:
            raise ValueError("NativePinSAGEStep: projector table and scorer bias disagree on the item count")
        sp = opt.state[proj]
        d.proj, d.g_proj, d.m_proj, d.v_proj = proj.data_ptr(), proj.grad.data_ptr(), sp["exp_avg"].data_ptr(), sp["exp_avg_sq"].data_ptr()
        d.bias, d.g_bias = bias.data_ptr(), bias.grad.data_ptr()
        for l, cv in enumerate(model.convs):
            c = d.conv[l]
            c.q_w, c.q_b, c.w_w, c.w_b = (x.data_ptr() for x in (cv.Q.weight, cv.Q.bias, cv.W.weight, cv.W.bias))
            c.g_q_w, c.g_q_b, c.g_w_w, c.g_w_b = (x.grad.data_ptr() for x in (cv.Q.weight, cv.Q.bias, cv.W.weight, cv.W.bias))
        i = 0
        for p in group["params"]:
            if p is proj:
                continue
            st = opt.state[p]
            q = d.params[i]
            q.p, q.g, q.m, q.v, q.n = p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()
            i += 1
        d.n_params = i
        steps = [opt.state[p]["step"] for p in group["params"]]
        self._adam_step = int(steps[0]) if steps else 0
        return d

    def _current(self, d: PinsageModel) -> bool:
        """The descriptor holds raw pointers: rebuilt when a parameter, gradient or optimizer-state tensor was replaced."""
        group = self.optimizer.param_groups[0]
        proj = self.model.proj.weight
        i = 0
        for p in group["params"]:
            st = self.optimizer.state.get(p)
            if p.grad is None or not st:
                return False
            if p is proj:
                if (d.proj, d.g_proj, d.m_proj, d.v_proj) != (p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(),
                                                              st["exp_avg_sq"].data_ptr()):
                    return False
                continue
            q = d.params[i]
            if (q.p, q.g, q.m, q.v) != (p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()):
                return False
            i += 1
        return i == d.n_params

    # ------------------------------------------------------------------------------------------
    def _prepare(self, batch: dict):
        """Everything up to (not including) the launch, the executor's own validation pass included (mi_pinsage_step_check).
        Returns (d, b, loss, keep-alive) or None with self.declined set; nothing has been enqueued either way."""
        model = self.model
        if not model.training:
            self.declined = "model in eval mode"
            return None
        blocks = batch["blocks"]
        if len(blocks) != len(model.convs):
            self.declined = "as many blocks as layers expected"
            return None
        if any("csr" not in b for b in blocks):
            if not self.build_csrs:
                self.declined = "blocks without their CSRs (not a device-built batch)"
                return None
            from .model import block_csr           # index-op batches (sizes beyond the device builder's): two sorts per block
            for b in blocks:
                if "csr" not in b:
                    b["csr"] = block_csr(b)
        seeds, (pu, pv), (nu, nv) = batch["seeds"], batch["pos"], batch["neg"]
        if pu.numel() == 0 or nu.data_ptr() != pu.data_ptr():
            self.declined = "no pairs / negative pairs with their own heads"
            return None
        if self._desc is None or not self._current(self._desc):
            self._desc = self._build()
        elif self.keep_grads:      # the previous call left its rows in the two dense buffers
            model.proj.weight.grad.zero_()
            model.bias.grad.zero_()
        d = self._desc
        group = self.optimizer.param_groups[0]
        d.p_dropout = float(model.convs[0].dropout.p)
        d.lr, (d.beta1, d.beta2), d.eps = float(group["lr"]), (float(b) for b in group["betas"]), float(group["eps"])
        d.apply_adam = 0 if (self.keep_grads or self.data_parallel) else 1
        d.step = self._adam_step + 1
        b = PinsageStepBatch()
        b.n_blocks = len(blocks)
        keep = []
        for l, blk in enumerate(blocks):
            by_dst, by_src = blk["csr"]
            sb = b.blocks[l]
            sb.n_src, sb.n_dst, sb.nnz = int(blk["src_ids"].numel()), int(blk["n_dst"]), int(by_dst.nnz)
            sb.src_ids = blk["src_ids"].data_ptr()
            sb.dst_rowptr, sb.src_rowptr = by_dst.rowptr.data_ptr(), by_src.rowptr.data_ptr()
            if by_dst.nnz:
                sb.dst_col, sb.dst_val = by_dst.col.data_ptr(), by_dst.val.data_ptr()
                sb.src_col, sb.src_val = by_src.col.data_ptr(), by_src.val.data_ptr()
            keep.append((by_dst, by_src))
        b.n_seeds, b.n_pairs = int(seeds.numel()), int(pu.numel())
        b.seeds, b.pos_u, b.pos_v, b.neg_v = seeds.data_ptr(), pu.data_ptr(), pv.data_ptr(), nv.data_ptr()
        b.seed, b.step = self.seed, self.iteration
        loss = t.empty(1, dtype=t.float32, device=seeds.device)
        b.loss = loss.data_ptr()
        ones = _ones4(max(int(blk["src_ids"].numel()) for blk in blocks), seeds.device)
        d.ones4, d.n_ones = ones.data_ptr(), int(ones.shape[0])
        if self.data_parallel:
            send = self._exchange_buffer(b, blocks[0]["src_ids"], seeds)
            if send is None:
                return None
        L = _lib.lib()
        need = int(L.mi_pinsage_step_workspace_bytes(ctypes.byref(d), ctypes.byref(b)))
        if self._ws is None or self._ws.numel() < need:
            self._ws = t.empty(int(need * 1.25) + (1 << 20), dtype=t.uint8, device=seeds.device)
        import torch.distributed as dist
        if self.data_parallel and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            # the validation pass on its own only where the ranks must agree before anything is enqueued
            rc = L.mi_pinsage_step_check(ctypes.byref(d), ctypes.byref(b), self._ws.data_ptr(), self._ws.numel())
            if rc == _lib.MI_ERR_UNSUPPORTED:
                self.declined = "mi_pinsage_step_f32: MI_ERR_UNSUPPORTED (shape outside the executor's)"
                self._desc = None      # the caller's own step may leave anything in the gradient buffers: start clean next time
                return None
            _lib.check(rc, "mi_pinsage_step_check")
        return d, b, loss, (keep, ones, seeds, pu, pv, nv)

    def step(self, batch: dict) -> Optional[Tensor]:
        """One iteration on a PinSAGESampler batch; the loss as a 1-element device tensor, or None when declined (nothing
        has been enqueued then).  Data-parallel with more than one rank: the decline is COLLECTIVE — one all-reduce(MIN) of
        a 1-int flag before anything is enqueued, so that a rank whose batch lies outside the executor's shapes does not
        leave its peers waiting in the row exchange; every rank returns None together."""
        import torch.distributed as dist
        self.declined = None
        prep = self._prepare(batch)
        world = dist.get_world_size(self.group) if (self.data_parallel and dist.is_initialized()) else 1
        if world > 1:
            flag = t.tensor([1 if prep is not None else 0], dtype=t.int32, device=self.model.proj.weight.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            if not int(flag.item()):
                if prep is not None:
                    self.declined = "a peer rank declined its batch (collective decision: every rank takes the fallback)"
                    self._desc = None
                return None
        elif prep is None:
            return None
        d, b, loss, _keep = prep
        group = self.optimizer.param_groups[0]
        L = _lib.lib()
        rc = L.mi_pinsage_step_f32(ctypes.byref(d), ctypes.byref(b), self._ws.data_ptr(), self._ws.numel(), _lib.current_stream())
        if rc == _lib.MI_ERR_UNSUPPORTED:
            if world > 1:                      # cannot happen: mi_pinsage_step_check took the same descriptors
                raise _lib.MiError("mi_pinsage_step_f32 declined a batch its own validation pass had accepted")
            self.declined = "mi_pinsage_step_f32: MI_ERR_UNSUPPORTED (shape outside the executor's)"
            self._desc = None      # the caller's own step may leave anything in the gradient buffers: start clean next time
            return None
        _lib.check(rc, "mi_pinsage_step_f32")
        self.iteration += 1
        if self.data_parallel:
            self._exchange_and_apply(d)
        if not self.keep_grads:
            self._adam_step += 1
            steps = [self.optimizer.state[q]["step"] for q in group["params"]]
            if steps and steps[0].is_cuda:
                t._foreach_add_(steps, 1)
            else:
                for s in steps:
                    s += 1
        return loss

    # ------------------------------------------------------------------------------------------
    # data-parallel exchange.  One int32 buffer per rank: [n_rows, n_seeds, 0, 0 | ids as int64 (2 words each, cap entries) |
    # rows as float32 (cap * hidden) | bias as float32 (cap_seeds)]; the executor writes rows / bias straight into it.
    def _layout(self, hidden: int):
        cap, cap_s = self._xbuf[2], self._xbuf[3]
        o_ids = 4
        o_rows = o_ids + 2 * cap
        o_bias = o_rows + cap * hidden
        return cap, cap_s, o_ids, o_rows, o_bias, o_bias + cap_s

    def _exchange_buffer(self, b: PinsageStepBatch, ids0: Tensor, seeds: Tensor):
        import torch.distributed as dist
        hidden = int(self.model.proj.weight.shape[1])
        n0, ns = int(ids0.numel()), int(seeds.numel())
        if self._xbuf is None:
            # capacity from the sampler's bounds: 3 B seeds, (1 + T) growth per layer — the same on every rank.  Callers with
            # other batch sources set .exchange_capacity = (rows, seeds) before the first step.
            cap = getattr(self, "exchange_capacity", None)
            if cap is None:
                self.declined = "data_parallel needs .exchange_capacity = (max rows of block 0, max seeds), equal on every rank"
                return None
            world = dist.get_world_size(self.group) if dist.is_initialized() else 1
            rows_cap, seeds_cap = (int(cap[0]) + 3) // 4 * 4, (int(cap[1]) + 3) // 4 * 4   # keeps every section 16-byte aligned
            words = 4 + 2 * rows_cap + rows_cap * hidden + seeds_cap
            dev = seeds.device
            self._xbuf = (t.zeros(words, dtype=t.int32, device=dev), t.zeros(world, words, dtype=t.int32, device=dev), rows_cap,
                          seeds_cap, world)
        cap, cap_s, o_ids, o_rows, o_bias, _ = self._layout(hidden)
        if n0 > cap or ns > cap_s:
            self.declined = f"batch larger than the exchange capacity ({n0} rows / {ns} seeds vs {cap} / {cap_s})"
            return None
        send = self._xbuf[0]
        send[0:2] = t.tensor([n0, ns], dtype=t.int32, device=send.device)
        send[o_ids: o_ids + 2 * n0].view(t.int64).copy_(ids0)
        b.rows_out = send.data_ptr() + 4 * o_rows
        b.bias_out = send.data_ptr() + 4 * o_bias
        return send

    def _exchange_and_apply(self, d: PinsageModel) -> None:
        import torch.distributed as dist
        send, gathered, cap, cap_s, world = self._xbuf
        hidden = int(self.model.proj.weight.shape[1])
        _, _, o_ids, o_rows, o_bias, _ = self._layout(hidden)
        if world > 1:
            # the all-gather as an all-reduce(sum) of a [world, words] int32 buffer that is zero outside the rank's own
            # slot (bit patterns + 0 = bit patterns): the payload is tiny (world x 0.4 MB), and gloo's all_gather takes
            # 225 ms for it on this image (tools/probes/gloo_ops.py) where its all_reduce takes 0.3 ms
            gathered.zero_()
            gathered[dist.get_rank(self.group)].copy_(send)
            dist.all_reduce(gathered, op=dist.ReduceOp.SUM, group=self.group)
            dist.all_reduce(self._flat_small, op=dist.ReduceOp.SUM, group=self.group)
        else:
            gathered[0].copy_(send)
        counts = gathered[:, 0:2].cpu().tolist()     # the one read-back of the exchange (2 ints per rank)
        lists = (_lib.PinsageGradList * world)()
        base, stride = gathered.data_ptr(), 4 * gathered.shape[1]
        for r in range(world):
            lists[r].n_rows, lists[r].n_seeds = int(counts[r][0]), int(counts[r][1])
            lists[r].ids = base + r * stride + 4 * o_ids
            lists[r].rows = base + r * stride + 4 * o_rows
            lists[r].bias = base + r * stride + 4 * o_bias
        _lib.check(_lib.lib().mi_pinsage_apply_f32(ctypes.byref(d), lists, world, 1.0 / world, _lib.current_stream()),
                   "mi_pinsage_apply_f32")

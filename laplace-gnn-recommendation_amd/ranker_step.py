"""One ranker training iteration without the autograd engine.

`training.__train` of the reference (training.py:19-34) is `zero_grad -> model(...) -> BCEWithLogitsLoss -> backward ->
optimizer.step`.  At the reference's batch size (24 users, ~3*10^4 nodes, ~5*10^4 edges) every kernel of that iteration
runs for 5-40 us, so the iteration is bound by launches and — on the host — by what PyTorch does around each of them:
module dispatch, autograd-node construction, the engine's walk of the graph (measured, tools/prof_host_ranker.py:
1.35 ms of Python per iteration against 1.07 ms of kernels).  `FusedRankerStep` runs the SAME arithmetic — the same
C-ABI launches in the same order as `Encoder_Decoder_Model.forward` + autograd would issue — as straight-line code:
forward, hand-derived backward (the functions the autograd nodes of model/layers.py and model/encoder_decoder.py call),
gradients written to `param.grad`, then the caller's `optimizer.step()`.  The model object, its parameters, buffers,
state_dict and the optimizer stay torch's own; `training.train_with_dataloader` switches to this step when the model
has the shape it supports (`supports(model)`) and falls back to autograd otherwise.

Dropout uses torch's own generator (`native_dropout`), so the masks are the ones `F.dropout` would have drawn.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch as t
from torch import Tensor

from . import ops
from .model.encoder_decoder import Encoder_Decoder_Model, _key
from .model.layers import BipartiteGraph, Linear, _run_products, _ones4, hetero_layer_backward, hetero_layer_forward, \
    hetero_layer_relations
from .utils.constants import Constants


def _dropout(x: Tensor, p: Optional[float]):
    if p is None or p <= 0.0:
        return x, None
    return t.native_dropout(x, p, True)


def _dropout_bwd(dy: Tensor, mask: Optional[Tensor], p: Optional[float]) -> Tensor:
    if mask is None:
        return dy
    return t.ops.aten.native_dropout_backward(dy, mask, 1.0 / (1.0 - p))


class FusedRankerStep:
    def __init__(self, model: Encoder_Decoder_Model, optimizer: t.optim.Optimizer, before_step=None):
        """before_step: called with no arguments once every gradient is in place, before optimizer.step() — the data-parallel
        ranker passes `lambda: dist_ranker.allreduce_gradients(model.parameters())`."""
        if not self.supports(model):
            raise ValueError("model is not of the shape FusedRankerStep supports")
        self.model, self.optimizer, self.before_step = model, optimizer, before_step

    @staticmethod
    def supports(model) -> bool:
        """Encoder_Decoder_Model whose decoder is a stack of this package's Linear layers; the per-layer check of the
        encoder (one relation per destination type, root weight, no normalize) happens on the first batch."""
        return (isinstance(model, Encoder_Decoder_Model) and all(isinstance(l, Linear) for l in model.decoder.layers)
                and len(model.decoder.layers) >= 1)

    # ------------------------------------------------------------------------------------------
    def step(self, x_dict: Dict[str, Tensor], edge_index_dict: dict, edge_label_index: Tensor, labels: Tensor) -> Optional[Tensor]:
        """One iteration; returns the loss (1-element device tensor), or None when this batch's graph is not of the
        fused form (the caller then runs the autograd path)."""
        model = self.model
        enc, dec = model.encoder, model.decoder
        p = enc.p_dropout_features
        training = model.training
        if model.batch_normalize and not (model.encoder_layer_norm_customer.training and model.encoder_layer_norm_article.training):
            return None                              # eval-mode statistics during training: left to the autograd path
        if model.embedding:
            x_dict = model._embed(dict(x_dict))
        types = list(x_dict)
        # ---- graphs (as HeteroGNNEncoder.forward builds them)
        graphs, built = {}, {}
        for et, ei in edge_index_dict.items():
            et = tuple(et)
            if _key(et) in enc.layers[0]:
                fwd = getattr(ei, "_reverse_of", None)
                if isinstance(ei, BipartiteGraph):
                    graphs[et] = ei
                elif fwd is not None and id(fwd) in built:
                    graphs[et] = built[id(fwd)].reversed()
                else:
                    graphs[et] = BipartiteGraph.of(ei, x_dict[et[0]].shape[0], x_dict[et[2]].shape[0])
                    built[id(ei)] = graphs[et]
        # ---- encoder forward
        n_layers = len(enc.layers)
        saved = []
        xs = [x_dict[k] for k in types]
        for index, convs in enumerate(enc.layers):
            last = index == n_layers - 1
            masks = [None] * len(xs)
            if not last and p is not None and training:
                dropped = [_dropout(x, p) for x in xs]
                xs, masks = [d[0] for d in dropped], [d[1] for d in dropped]
            xs = [x if x.stride(-1) == 1 else x.contiguous() for x in xs]
            col = hetero_layer_relations({et: convs[_key(et)] for et in graphs}, graphs, dict(zip(types, xs)))
            if col is None:
                return None
            rels, wts = col
            outs, aggs, args = hetero_layer_forward(rels, not last, xs, wts)
            saved.append((rels, wts, xs, aggs, args, outs if not last else [None] * len(rels), masks, convs))
            nxt: List[Optional[Tensor]] = [None] * len(types)
            for (si, di, _, _), o in zip(rels, outs):
                nxt[di] = o
            if any(v is None for v in nxt):   # a node type no relation arrives at drops out, as in to_hetero
                return None
            xs = nxt
        iu, ii = types.index(Constants.node_user), types.index(Constants.node_item)
        zu, zi = xs[iu], xs[ii]
        # ---- BatchNorm
        bn_saved = None
        if model.batch_normalize:
            bn_saved = []
            z = []
            for bn, x in ((model.encoder_layer_norm_customer, zu), (model.encoder_layer_norm_article, zi)):
                train_bn = bn.training or not bn.track_running_stats
                momentum = 0.0
                if train_bn and bn.track_running_stats:
                    bn.num_batches_tracked.add_(1)
                    momentum = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
                rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
                y, mean, invstd = ops.batchnorm_fwd(x, bn.weight, bn.bias, rm, rv, float(momentum), float(bn.eps), train_bn)
                bn_saved.append((bn, x, mean, invstd))
                z.append(y)
            zu, zi = z
        # ---- decoder forward
        row, col_ = edge_label_index[0].contiguous(), edge_label_index[1].contiguous()
        h = ops.gather_cat(zu, zi, row, col_)
        dec_saved = []
        n_dec = len(dec.layers)
        pd = dec.p_dropout_features
        for index, layer in enumerate(dec.layers):
            if layer.weight is None:
                layer._materialize(int(h.shape[-1]), h.device)
            last = index == n_dec - 1
            mask = None
            if not last and pd is not None and training:
                h, mask = _dropout(h, pd)
            out = ops.gemm(h, layer.weight, trans_b=True, bias=layer.bias, relu=not last)
            dec_saved.append((layer, h, mask, out if not last else None))
            h = out
        logits = h.view(-1)
        loss, dlogits = ops.bce_logits(logits, labels.to(t.float32).contiguous())
        # ---- decoder backward
        dev = logits.device
        dh = dlogits.view(h.shape)
        for layer, x_in, mask, out in reversed(dec_saved):
            if layer.out_features == 1 and out is None and x_in.shape[0] > 0:
                # the last layer, Linear(128, 1): one band-sum kernel pair instead of three [n, 1]-shaped products
                dx, dw, db = ops.linear1_bwd(dh.reshape(-1), layer.weight, x_in)
                layer.weight.grad = dw
                if layer.bias is not None:
                    layer.bias.grad = db
                dh = _dropout_bwd(dx, mask, pd)
                continue
            dw = t.empty_like(layer.weight)
            dx = t.empty(x_in.shape, device=dev)
            specs = [dict(A=dh, B=layer.weight, out=dx, trans_b=False, mask=out),
                     dict(A=dh, B=x_in, out=dw, trans_a=True, trans_b=False, mask=out)]
            db4 = None
            if layer.bias is not None:
                db4 = t.empty(layer.out_features, 4, device=dev)
                specs.append(dict(A=dh, B=_ones4(dh.shape[0], dev), out=db4, trans_a=True, trans_b=False, mask=out))
            _run_products(specs)
            layer.weight.grad = dw
            if db4 is not None:
                layer.bias.grad = db4[:, 0].contiguous()
            dh = _dropout_bwd(dx, mask, pd)
        cu, ci = zu.shape[1], zi.shape[1]
        dzu = ops.gather_cat_bwd(dh, row, zu.shape[0], cu, 0)
        dzi = ops.gather_cat_bwd(dh, col_, zi.shape[0], ci, cu)
        # ---- BatchNorm backward
        if bn_saved is not None:
            grads = []
            for (bn, x, mean, invstd), dz in zip(bn_saved, (dzu, dzi)):
                dx, dg, db = ops.batchnorm_bwd(x, dz, bn.weight, mean, invstd, need_dx=True, need_dw=bn.weight is not None)
                if bn.weight is not None:
                    bn.weight.grad, bn.bias.grad = dg, db
                grads.append(dx)
            dzu, dzi = grads
        # ---- encoder backward
        dxs: List[Optional[Tensor]] = [None] * len(types)
        dxs[iu], dxs[ii] = dzu, dzi
        for index in range(n_layers - 1, -1, -1):
            rels, wts, xs_in, aggs, args, outs, masks, convs = saved[index]
            dys = [dxs[di] for (_, di, _, _) in rels]
            need_x = index > 0            # layer 0 reads frozen embeddings / raw features
            need = [need_x] * len(types) + [True, True, True] * len(rels)
            for i, w in enumerate(wts):
                if w is None:
                    need[len(types) + i] = False
            dxl, gw = hetero_layer_backward(rels, xs_in, wts, aggs, args, outs, dys, need)
            for i, et in enumerate(graphs):
                conv = convs[_key(et)]
                conv.lin_l.weight.grad = gw[3 * i]
                if conv.lin_l.bias is not None:
                    conv.lin_l.bias.grad = gw[3 * i + 1]
                conv.lin_r.weight.grad = gw[3 * i + 2]
            if need_x:
                dxs = [None if d is None else _dropout_bwd(d, m, p) for d, m in zip(dxl, masks)]
        if self.before_step is not None:
            self.before_step()
        self.optimizer.step()
        return loss

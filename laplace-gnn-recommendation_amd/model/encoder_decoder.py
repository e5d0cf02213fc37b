"""Encoder-decoder ranker — same class, constructor and methods as the reference's
model/encoder_decoder.py:17-164, on the HIP kernels.

    forward = embeddings (K7) -> hetero SAGEConv encoder (K5 + K6) -> per-type BatchNorm1d
              -> MLP edge decoder over cat(z_user[row], z_item[col]) (K6)

`to_hetero(GNNEncoder, metadata, aggr)` (model/encoder_decoder.py:93-95) is restated as
`HeteroGNNEncoder`: every layer is duplicated per edge type with fresh parameters, the conv of
relation (s, r, d) sees (x_s, x_d), and the outputs arriving at one destination type are combined
with `aggr` (text spec: temporary_hetero.py:171-228,337-360).

Faithful quirks: the categorical embedding tables live in a plain dict (not parameters, not in the
state_dict, frozen) — here they are at least moved by .to(device).  BatchNorm (K8) and the decoder's
gather + concat run on csrc/norm.hip behind torch's own BatchNorm1d modules (parameters, running statistics and
state_dict keys unchanged); dropout stays torch's (its mask is torch's RNG stream, as in the reference).
"""
from __future__ import annotations

from collections import deque
from copy import deepcopy
from typing import Dict, List, Optional, Tuple

import torch as t
import torch.nn.functional as F
from torch import Tensor, nn
from torch.nn import BatchNorm1d, ModuleDict, ModuleList

from .. import ops
from ..utils.constants import Constants
from .layers import BipartiteGraph, hetero_sage_layer


def _key(edge_type: Tuple[str, str, str]) -> str:
    return "__".join(edge_type)


_PAIR = {"sum": t.add, "mean": t.add, "max": t.maximum, "min": t.minimum, "mul": t.mul}


def _combine(outs: List[Tensor], aggr: str) -> Tensor:
    """Outputs of the relations arriving at one destination type, reduced the way PyG's to_hetero wires it
    (temporary_hetero.py:203-228): PAIRWISE through a queue — pop the two oldest, combine, append the result — and
    for "mean" one division by the relation count at the end (:120-128).  Three relations [a, b, c] therefore give
    c (+) (a (+) b); the order is kept because float addition is not associative."""
    if aggr not in _PAIR:
        raise ValueError(f"unknown heterogeneous aggregation {aggr!r}")
    if len(outs) == 1:
        return outs[0]
    queue = deque(outs)
    while len(queue) >= 2:
        a, b = queue.popleft(), queue.popleft()
        queue.append(_PAIR[aggr](a, b))
    out = queue.popleft()
    return t.div(out, len(outs)) if aggr == "mean" else out


class _BatchNormFn(t.autograd.Function):
    """BatchNorm1d on csrc/norm.hip: 2 launches forward, 2 backward, sums in double in a fixed order."""

    @staticmethod
    def forward(ctx, x: Tensor, weight: Optional[Tensor], bias: Optional[Tensor], running_mean: Optional[Tensor],
                running_var: Optional[Tensor], momentum: float, eps: float, training: bool):
        x = x if x.stride(-1) == 1 else x.contiguous()
        y, mean, invstd = ops.batchnorm_fwd(x, weight, bias, running_mean, running_var, momentum, eps, training)
        if not training:
            invstd = t.rsqrt(running_var + eps)
            mean = running_mean
        ctx.save_for_backward(x, weight, mean, invstd)
        ctx.training = training
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        x, weight, mean, invstd = ctx.saved_tensors
        dy = dy.contiguous()
        need_dw = weight is not None and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        if ctx.training:
            dx, dg, db = ops.batchnorm_bwd(x, dy, weight, mean, invstd, need_dx=ctx.needs_input_grad[0], need_dw=need_dw)
        else:  # eval-mode statistics are constants (not a training path): plain tensor arithmetic
            scale = invstd if weight is None else invstd * weight
            dx = dy * scale if ctx.needs_input_grad[0] else None
            dg = (dy * (x - mean) * invstd).sum(0) if need_dw else None
            db = dy.sum(0) if need_dw else None
        return dx, dg, db, None, None, None, None, None


def batch_norm(bn: BatchNorm1d, x: Tensor) -> Tensor:
    """bn(x) for a torch BatchNorm1d module (its parameters, buffers and state_dict stay torch's), computed by the
    hand-written kernels."""
    training = bn.training or not bn.track_running_stats
    momentum = 0.0
    if training and bn.track_running_stats:
        bn.num_batches_tracked.add_(1)
        momentum = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
    rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
    return _BatchNormFn.apply(x, bn.weight, bn.bias, rm, rv, float(momentum), float(bn.eps), training)


class _GatherCatFn(t.autograd.Function):
    """cat(z_user[row], z_item[col], dim=-1) in one launch; backward = one deterministic segmented sum per table."""

    @staticmethod
    def forward(ctx, zu: Tensor, zi: Tensor, row: Tensor, col: Tensor):
        zu = zu if zu.stride(-1) == 1 else zu.contiguous()
        zi = zi if zi.stride(-1) == 1 else zi.contiguous()
        row, col = row.contiguous(), col.contiguous()
        ctx.save_for_backward(row, col)
        ctx.shapes = (zu.shape, zi.shape)
        return ops.gather_cat(zu, zi, row, col)

    @staticmethod
    def backward(ctx, dz: Tensor):
        row, col = ctx.saved_tensors
        (nu, cu), (ni, ci) = ctx.shapes
        dz = dz.contiguous()
        du = ops.gather_cat_bwd(dz, row, nu, cu, 0) if ctx.needs_input_grad[0] else None
        di = ops.gather_cat_bwd(dz, col, ni, ci, cu) if ctx.needs_input_grad[1] else None
        return du, di, None, None


class HeteroGNNEncoder(nn.Module):
    """GNNEncoder.forward (model/encoder_decoder.py:29-46) applied per edge type."""

    def __init__(self, layers: ModuleList, metadata, aggr: str, p_dropout_edges: Optional[float],
                 p_dropout_features: Optional[float]):
        super().__init__()
        self.node_types, self.edge_types = list(metadata[0]), [tuple(e) for e in metadata[1]]
        self.aggr = aggr
        self.p_dropout_edges, self.p_dropout_features = p_dropout_edges, p_dropout_features
        per_layer = []
        for layer in layers:
            convs = {}
            for et in self.edge_types:
                conv = deepcopy(layer)
                conv.reset_parameters()
                convs[_key(et)] = conv
            per_layer.append(ModuleDict(convs))
        self.layers = ModuleList(per_layer)

    def forward(self, x_dict: Dict[str, Tensor], edge_index_dict: Dict[Tuple[str, str, str], Tensor]):
        graphs = {}
        built = {}  # id(edge_index tensor) -> its graph, for relations declared to be another one reversed
        for et, ei in edge_index_dict.items():
            et = tuple(et)
            if _key(et) in self.layers[0]:
                fwd = getattr(ei, "_reverse_of", None)
                if isinstance(ei, BipartiteGraph):
                    graphs[et] = ei
                elif fwd is not None and id(fwd) in built:
                    graphs[et] = built[id(fwd)].reversed()
                else:
                    graphs[et] = BipartiteGraph.of(ei, x_dict[et[0]].shape[0], x_dict[et[2]].shape[0])
                    built[id(ei)] = graphs[et]
        n_layers = len(self.layers)
        for index, convs in enumerate(self.layers):
            last = index == n_layers - 1
            if not last and self.p_dropout_features is not None:
                x_dict = {k: F.dropout(v, p=self.p_dropout_features, training=self.training) for k, v in x_dict.items()}
            fused = hetero_sage_layer({et: convs[_key(et)] for et in graphs}, graphs, x_dict, relu=not last)
            if fused is not None:  # one autograd node, grouped GEMM launches (model/layers.py)
                x_dict = fused
                continue
            arriving: Dict[str, int] = {}
            for et in graphs:
                arriving[et[2]] = arriving.get(et[2], 0) + 1
            by_dst: Dict[str, List[Tensor]] = {}
            for et, graph in graphs.items():
                # one relation into this destination (the default metadata): the relu rides in the conv's GEMM epilogue
                act = (not last) and arriving[et[2]] == 1
                out = convs[_key(et)]((x_dict[et[0]], x_dict[et[2]]), graph, relu=act)
                by_dst.setdefault(et[2], []).append(out)
            x_dict = {dst: (outs[0] if len(outs) == 1 else _combine(outs, self.aggr)) for dst, outs in by_dst.items()}
            if not last:
                x_dict = {k: (v if arriving[k] == 1 else v.relu()) for k, v in x_dict.items()}
        return x_dict


class EdgeDecoder(nn.Module):
    def __init__(self, layers: ModuleList, p_dropout_features: Optional[float]):
        super().__init__()
        self.layers = layers
        self.p_dropout_features = p_dropout_features

    def forward(self, z_dict: dict, edge_label_index) -> Tensor:
        customer_index, article_index = edge_label_index
        z = _GatherCatFn.apply(z_dict[Constants.node_user], z_dict[Constants.node_item], customer_index, article_index)
        n_layers = len(self.layers)
        for index, layer in enumerate(self.layers):
            if index == n_layers - 1:
                z = layer(z)
            else:
                if self.p_dropout_features is not None:
                    z = F.dropout(z, p=self.p_dropout_features, training=self.training)
                z = layer(z, relu=True)  # relu fused into the GEMM epilogue
        return z.view(-1)


class Encoder_Decoder_Model(nn.Module):
    def __init__(self, encoder_layers: ModuleList, decoder_layers: ModuleList, feature_info: dict, metadata,
                 embedding: bool, heterogeneous_prop_agg_type: str, batch_normalize: bool,
                 p_dropout_edges: Optional[float], p_dropout_features: Optional[float]):
        super().__init__()
        self.embedding = embedding
        self.batch_normalize = batch_normalize
        self.encoder = HeteroGNNEncoder(encoder_layers, metadata, heterogeneous_prop_agg_type, p_dropout_edges,
                                        p_dropout_features)
        self.decoder = EdgeDecoder(decoder_layers, p_dropout_features)
        self.encoder_layer_norm_customer = BatchNorm1d(encoder_layers[-1].out_channels)
        self.encoder_layer_norm_article = BatchNorm1d(encoder_layers[-1].out_channels)

        # plain dict on purpose (model/encoder_decoder.py:101): frozen tables, outside parameters()/state_dict
        self.embedding_layers: Dict[str, List[Tensor]] = dict()
        if self.embedding:
            for key, item in feature_info.items():
                tables = []
                for i in range(item.num_feat):
                    # nn.Embedding draws N(0, 1): create it exactly like the reference so a seeded build matches
                    emb = nn.Embedding(num_embeddings=int(item.num_cat[i] + 1), embedding_dim=int(item.embedding_size[i]))
                    tables.append(emb.weight.detach().clone())
                self.embedding_layers[key] = tables

    def _apply(self, fn, *args, **kwargs):
        super()._apply(fn, *args, **kwargs)
        self.embedding_layers = {k: [fn(tb) for tb in v] for k, v in self.embedding_layers.items()}
        return self

    def _embed(self, x_dict: dict) -> dict:
        """Encoder_Decoder_Model.__embedding: per column Embedding(max_norm=1) lookup, concatenated."""
        for key, tables in self.embedding_layers.items():
            x_dict[key] = ops.embed_concat(x_dict[key].contiguous(), tables, max_norm=1.0)
        return x_dict

    def validate_features(self, x_dict: dict) -> None:
        """nn.Embedding raises on an id outside its table (model/encoder_decoder.py:116-125); the lookup kernel only
        clamps such ids to stay memory-safe, so a mis-encoded column would train on the wrong rows in silence.  This
        is the check, once per dataset or batch (one host read-back): call it on the full graph's x_dict; 
        initialize_encoder_input_size() runs it on the batch it is given."""
        for key, tables in self.embedding_layers.items():
            x = x_dict[key]
            if x.dim() != 2 or x.shape[1] != len(tables):
                raise ValueError(f"{key}: expected {len(tables)} categorical columns, got shape {tuple(x.shape)}")
            if x.numel() == 0:
                continue
            lo, hi = x.amin(dim=0).tolist(), x.amax(dim=0).tolist()
            for i, tb in enumerate(tables):
                if lo[i] < 0 or hi[i] >= tb.shape[0]:
                    raise IndexError(f"{key} column {i}: ids span [{lo[i]}, {hi[i]}] but its embedding table has "
                                     f"{tb.shape[0]} rows (num_cat + 1)")

    def initialize_encoder_input_size(self, data) -> None:
        x_dict, edge_index_dict = data.x_dict, data.edge_index_dict
        if self.embedding:
            self.validate_features(x_dict)
            x_dict = self._embed(x_dict)
        with t.no_grad():
            self.encoder(x_dict, edge_index_dict)

    def forward(self, x_dict, edge_index_dict: dict, edge_label_index: Tensor) -> Tensor:
        if self.embedding:
            x_dict = self._embed(x_dict)
        z_dict = self.encoder(x_dict, edge_index_dict)
        if self.batch_normalize:
            z_dict[Constants.node_user] = batch_norm(self.encoder_layer_norm_customer, z_dict[Constants.node_user])
            z_dict[Constants.node_item] = batch_norm(self.encoder_layer_norm_article, z_dict[Constants.node_item])
        return self.decoder(z_dict, edge_label_index)

    def infer(self, x_dict, edge_index_dict: dict, edge_label_index: Tensor) -> Tensor:
        self.eval()
        out = self.forward(x_dict, edge_index_dict, edge_label_index).detach()
        # re-batch by user: one row of candidate scores per sorted-unique user, padded on the right
        users, inverse = edge_label_index[0].unique(sorted=True, return_inverse=True)
        counts = t.bincount(inverse, minlength=users.numel())
        order = t.argsort(inverse, stable=True)
        start = t.cumsum(counts, 0) - counts
        pos = t.arange(out.numel(), device=out.device) - start[inverse[order]]
        res = t.full((users.numel(), int(counts.max())), float(-(1 << 50)), dtype=out.dtype, device=out.device)
        res[inverse[order], pos] = out[order]
        return res

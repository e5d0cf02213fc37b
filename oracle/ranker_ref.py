"""ORACLE — test infrastructure only.  Plain-torch CPU restatement of the ranker
(model/layers.py, model/encoder_decoder.py, training.py:19-34 of the reference).

SAGEConv, to_hetero and scatter live in torch_geometric / torch_scatter (absent, unpinned ~PyG 2.0.4):
restated from their published semantics — PARITY UNPINNED for those (see oracle/__init__.py):
    SAGEConv(aggr, normalize=False, root_weight=True, bias=True):
        out = lin_l(scatter_aggr(x_src[edge_index[0]], edge_index[1], dim_size=n_dst)) + lin_r(x_dst)
        (lin_l with bias, lin_r without; mean over an empty set = 0; max over an empty set = 0)
    to_hetero: one copy of every layer per edge type; the conv of (s, r, d) is fed (x_s, x_d); outputs
        per destination type are combined with `aggr` (temporary_hetero.py:171-228).
Everything else is the reference's own arithmetic with real torch modules: nn.Linear,
nn.Embedding(max_norm=1), BatchNorm1d, F.dropout, BCEWithLogitsLoss.
Module names match the product's so that a state_dict moves across unchanged.
"""
from typing import Dict, List, Optional, Tuple

import torch as t
import torch.nn.functional as F
from torch import Tensor, nn


def scatter_aggr(src: Tensor, index: Tensor, dim_size: int, aggr: str) -> Tensor:
    """torch_scatter.scatter(src, index, dim=0, dim_size=dim_size, reduce=aggr)."""
    d = src.shape[1]
    if aggr in ("add", "sum"):
        return t.zeros(dim_size, d, dtype=src.dtype).index_add_(0, index, src)
    if aggr == "mean":
        s = t.zeros(dim_size, d, dtype=src.dtype).index_add_(0, index, src)
        cnt = t.bincount(index, minlength=dim_size).clamp(min=1).to(src.dtype)
        return s / cnt[:, None]
    if aggr == "max":
        out = t.zeros(dim_size, d, dtype=src.dtype)
        out = out.scatter_reduce(0, index[:, None].expand(-1, d), src, reduce="amax", include_self=False)
        empty = t.bincount(index, minlength=dim_size) == 0
        return out.masked_fill(empty[:, None], 0.0)
    raise ValueError(aggr)


class SAGEConvRef(nn.Module):
    def __init__(self, in_src: int, in_dst: int, out_channels: int, aggr: str):
        super().__init__()
        self.aggr, self.out_channels = aggr, out_channels
        self.lin_l = nn.Linear(in_src, out_channels, bias=True)
        self.lin_r = nn.Linear(in_dst, out_channels, bias=False)

    def forward(self, x: Tuple[Tensor, Tensor], edge_index: Tensor) -> Tensor:
        x_src, x_dst = x
        agg = scatter_aggr(x_src[edge_index[0]], edge_index[1], x_dst.shape[0], self.aggr)
        return self.lin_l(agg) + self.lin_r(x_dst)


def _key(et) -> str:
    return "__".join(et)


def hetero_reduce(outs: List[Tensor], aggr: str) -> Tensor:
    """Destination-wise aggregation exactly as to_hetero's transformer emits it (temporary_hetero.py:203-228 with the
    op table at :120-128): the per-relation outputs go into a queue in metadata order; while two or more are left the
    first two are popped, combined with torch.add / torch.add / torch.max / torch.min / torch.mul for
    sum / mean / max / min / mul, and the result is appended at the END; "mean" divides the survivor by the number of
    relations.  A single relation is passed through untouched (:185-190)."""
    if len(outs) == 1:
        return outs[0]
    op = {"sum": t.add, "mean": t.add, "max": t.max, "min": t.min, "mul": t.mul}[aggr]
    names = list(outs)
    while len(names) >= 2:
        key1, key2 = names.pop(0), names.pop(0)
        names.append(op(key1, key2))
    out = names.pop(0)
    if aggr == "mean":
        out = t.div(out, len(outs))
    return out


class HeteroEncoderRef(nn.Module):
    def __init__(self, dims: List[Dict[str, Tuple[int, int, int]]], aggr_conv: str, aggr_hetero: str,
                 p_dropout_features: Optional[float]):
        """dims[l][key] = (in_src, in_dst, out) for layer l and relation key."""
        super().__init__()
        self.aggr_hetero, self.p = aggr_hetero, p_dropout_features
        self.layers = nn.ModuleList([nn.ModuleDict({k: SAGEConvRef(a, b, c, aggr_conv) for k, (a, b, c) in layer.items()})
                                     for layer in dims])

    def forward(self, x_dict, edge_index_dict):
        n = len(self.layers)
        for index, convs in enumerate(self.layers):
            last = index == n - 1
            if not last and self.p is not None:
                x_dict = {k: F.dropout(v, p=self.p, training=self.training) for k, v in x_dict.items()}
            by_dst: Dict[str, List[Tensor]] = {}
            for et, ei in edge_index_dict.items():
                if _key(et) not in convs:
                    continue
                by_dst.setdefault(et[2], []).append(convs[_key(et)]((x_dict[et[0]], x_dict[et[2]]), ei))
            x_dict = {d: hetero_reduce(o, self.aggr_hetero) for d, o in by_dst.items()}
            if not last:
                x_dict = {k: v.relu() for k, v in x_dict.items()}
        return x_dict


class DecoderRef(nn.Module):
    def __init__(self, sizes: List[Tuple[int, int]], p: Optional[float], user_key: str, item_key: str):
        super().__init__()
        self.layers = nn.ModuleList([nn.Linear(a, b) for a, b in sizes])
        self.p, self.user_key, self.item_key = p, user_key, item_key

    def forward(self, z_dict, edge_label_index):
        ci, ai = edge_label_index
        z = t.cat([z_dict[self.user_key][ci], z_dict[self.item_key][ai]], dim=-1)
        for i, layer in enumerate(self.layers):
            if i == len(self.layers) - 1:
                z = layer(z)
            else:
                if self.p is not None:
                    z = F.dropout(z, p=self.p, training=self.training)
                z = layer(z).relu()
        return z.view(-1)


class RankerRef(nn.Module):
    """Encoder_Decoder_Model (model/encoder_decoder.py:75-164) with torch modules only."""

    def __init__(self, enc_dims, dec_sizes, embedding_tables: Optional[Dict[str, List[Tensor]]], aggr_conv: str,
                 aggr_hetero: str, batch_normalize: bool, p_dropout_features: Optional[float], out_channels: int,
                 user_key: str = "customer", item_key: str = "article"):
        super().__init__()
        self.encoder = HeteroEncoderRef(enc_dims, aggr_conv, aggr_hetero, p_dropout_features)
        self.decoder = DecoderRef(dec_sizes, p_dropout_features, user_key, item_key)
        self.encoder_layer_norm_customer = nn.BatchNorm1d(out_channels)
        self.encoder_layer_norm_article = nn.BatchNorm1d(out_channels)
        self.batch_normalize, self.user_key, self.item_key = batch_normalize, user_key, item_key
        self.embedding_layers = {}
        if embedding_tables is not None:
            for k, tabs in embedding_tables.items():  # plain dict, frozen, max_norm=1
                mods = []
                for tb in tabs:
                    e = nn.Embedding(tb.shape[0], tb.shape[1], max_norm=1)
                    with t.no_grad():
                        e.weight.copy_(tb)
                    e.weight.requires_grad_(False)
                    mods.append(e)
                self.embedding_layers[k] = mods

    def embed(self, x_dict):
        out = dict(x_dict)
        for k, mods in self.embedding_layers.items():
            out[k] = t.cat([m(x_dict[k][:, i]) for i, m in enumerate(mods)], dim=1)
        return out

    def forward(self, x_dict, edge_index_dict, edge_label_index):
        if self.embedding_layers:
            x_dict = self.embed(x_dict)
        z = self.encoder(x_dict, edge_index_dict)
        if self.batch_normalize:
            z[self.user_key] = self.encoder_layer_norm_customer(z[self.user_key])
            z[self.item_key] = self.encoder_layer_norm_article(z[self.item_key])
        return self.decoder(z, edge_label_index)

    def infer(self, x_dict, edge_index_dict, edge_label_index, pad_value=-(1 << 50)):
        self.eval()
        out = self.forward(x_dict, edge_index_dict, edge_label_index).detach()
        users = edge_label_index[0].unique(sorted=True)
        rows = [out[edge_label_index[0] == u] for u in users]
        width = max(r.numel() for r in rows)
        return t.stack([F.pad(r, (0, width - r.numel()), value=pad_value) for r in rows])


def ref_from_product(model, x_dict_sample: dict) -> RankerRef:
    """Builds the torch-only twin of a (lazily initialised) product Encoder_Decoder_Model and copies its
    weights.  x_dict_sample gives the raw feature dict (only used to read shapes)."""
    enc = model.encoder
    dims = []
    for convs in enc.layers:
        dims.append({k: (c.lin_l.in_features, c.lin_r.in_features, c.out_channels) for k, c in convs.items()})
    aggr_conv = next(iter(enc.layers[0].values())).aggr
    dec_sizes = [(l.in_features, l.out_features) for l in model.decoder.layers]
    tables = {k: [tb.detach().cpu().clone() for tb in v] for k, v in model.embedding_layers.items()} if model.embedding else None
    ref = RankerRef(dims, dec_sizes, tables, aggr_conv, enc.aggr, model.batch_normalize, enc.p_dropout_features,
                    model.encoder_layer_norm_customer.num_features)
    ref.load_state_dict({k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
    return ref

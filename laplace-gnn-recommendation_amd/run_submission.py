"""Batch inference / submission file — the flow of the reference's run_submission.py:14-96:
newest checkpoint -> test dataloader (candidates from the matchers) -> model.infer -> top-k of the
non-purchased candidates per customer -> ids mapped back -> `customer_id,prediction` CSV.

The reference's script cannot run as written; this module keeps its function names and flow and fixes
exactly what stops it (each noted where it happens):
  * load_model returns the checkpoint's state_dict and `.infer` is then called on that dict
    (run_submission.py:14-22,57) -> build_model reconstructs the model the way run_pipeline.py:47-73 does;
  * the per-edge label mask is applied to infer's per-user matrix (run_submission.py:60-61) -> the mask is
    applied to the per-edge scores before they are regrouped by user;
  * the picks are batch-local article positions but are mapped as if global (run_submission.py:64-66,36-37)
    -> batches carry `n_id` (global node ids) and the picks are translated through it;
  * t.stack over users with different candidate counts -> rows are padded with -1 (no prediction).
"""
from __future__ import annotations

import os
from os import listdir
from os.path import isfile, join
from typing import Dict, Optional, Tuple

import numpy as np
import torch as t
from torch import Tensor

from .config import Config, link_pred_config
from .data.dataset import GraphDataset
from .hetero import DataLoader
from .model.encoder_decoder import Encoder_Decoder_Model
from .model.layers import get_linear_layers, get_SAGEConv_layers
from .utils.constants import Constants
from .utils.get_info import get_feature_info, select_properties


def load_model(url: str) -> dict:
    """The checkpoint with the largest version number (run_submission.py:14-22): files `<name>_<version>.pt`."""
    files = [f for f in listdir(url) if isfile(join(url, f))]
    if not files:
        raise FileNotFoundError(f"no checkpoint under {url}")
    version_nums = [int(filename.split("_")[1].split(".")[0]) for filename in files]
    return t.load(join(url, files[int(np.argmax(version_nums))]), map_location="cpu")


def build_model(config: Config, full_data, first_batch, state_dict: Optional[dict] = None, device: str = "cuda"):
    """run_pipeline.py:47-73: same constructor arguments, lazy sizes from one batch, then the saved weights."""
    model = Encoder_Decoder_Model(
        encoder_layers=get_SAGEConv_layers(num_layers=config.num_gnn_layers, hidden_channels=config.hidden_layer_size,
                                           out_channels=config.encoder_layer_output_size, agg_type=config.conv_agg_type),
        decoder_layers=get_linear_layers(num_layers=config.num_linear_layers,
                                         in_channels=config.encoder_layer_output_size * 2,
                                         hidden_channels=config.hidden_layer_size, out_channels=1),
        feature_info=get_feature_info(full_data), metadata=first_batch.metadata(), embedding=True,
        heterogeneous_prop_agg_type=config.heterogeneous_prop_agg_type, batch_normalize=config.batch_norm,
        p_dropout_edges=config.p_dropout_edges, p_dropout_features=config.p_dropout_features).to(device)
    with t.no_grad():
        model.initialize_encoder_input_size(first_batch.to(device))
    if state_dict is not None:
        model.load_state_dict(state_dict)
    return model


FAST_SELECT = True      # tests switch these off to compare the paths
NATIVE_FORWARD = True   # the model's evaluation forward as one C call where the executor takes the model


@t.no_grad()
def make_predictions(model, dataloader, k: int, device: str = "cuda") -> Tuple[Tensor, Tensor]:
    """(customers [n] int64 global ids, predictions [n, k] int64 global article ids, -1 padded), customers in
    loader order.  Only label-0 edges compete (run_submission.py:59-66: the positives of the evaluation sample
    are articles the customer already bought)."""
    model.eval()
    customers, predictions = [], []
    native = None
    if NATIVE_FORWARD and str(device).startswith("cuda"):
        from .ranker_native import NativeRankerForward
        if NativeRankerForward.supports(model):
            native = NativeRankerForward(model)      # the forward as one C call (mi_ranker_batch.logits); declines fall to model(...)
    for batch in dataloader:
        seeds, user_ptr = getattr(batch, "_seed_users", None), getattr(batch, "_user_ptr", None)
        batch = batch.to(device)
        x, edge_index_dict, edge_label_index, edge_label = select_properties(batch)
        scores = native.logits(x, edge_index_dict, edge_label_index) if native is not None else None
        if scores is None:
            scores = model(x, edge_index_dict, edge_label_index).view(-1)
        if seeds is not None and user_ptr is not None and FAST_SELECT:
            # a device-built batch (data/device_sampler.py): sample s owns the customer nodes [user_ptr[s], user_ptr[s + 1]) and its
            # label edges are contiguous, so the row of a label edge is a searchsorted away and nothing below needs a host read —
            # the generic path costs five host waits per batch (unique, two .max(), two .cpu()): 0.68 of a 1.9 ms batch
            # (tools/prof_inference.py).  Same rows, same candidates per row in the same order, same top-k.
            B = int(seeds.numel())
            rows = t.searchsorted(user_ptr[1:].contiguous(), edge_label_index[0].contiguous(), right=True)
            keep = edge_label == 0
            ones = keep.to(t.int64)
            counts = t.zeros(B, dtype=t.int64, device=scores.device).scatter_add_(0, rows, ones)
            pos_all = t.cumsum(ones, 0) - ones                       # label-0 edges in front of this one, over the whole batch
            start = t.cumsum(counts, 0) - counts
            pos = pos_all - start[rows]                              # ... in front of it in its own row (rows are contiguous runs)
            width = max(int(getattr(batch, "_max_candidates", 0)) or int(edge_label.numel()), k)
            mat = t.full((B, width + 1), float("-inf"), device=scores.device)
            ids = t.full((B, width + 1), -1, dtype=t.int64, device=scores.device)
            col = t.where(keep, pos, t.full_like(pos, width))        # label-1 edges land in a spare last column
            mat[rows, col] = t.where(keep, scores, t.full_like(scores, float("-inf")))
            ids[rows, col] = t.where(keep, batch[Constants.node_item].n_id[edge_label_index[1]], t.full_like(pos, -1))
            top = t.topk(mat[:, :width], k=k, dim=1).indices
            customers.append(seeds.to(scores.device))
            predictions.append(t.gather(ids[:, :width], 1, top))
            continue
        keep = edge_label == 0
        users_l, arts_l, scores = edge_label_index[0][keep], edge_label_index[1][keep], scores[keep]
        # every customer of the batch gets a row, also one without candidates
        label_users = t.unique(edge_label_index[0])
        row_of = t.full((int(label_users.max()) + 1,), -1, dtype=t.int64, device=scores.device)
        row_of[label_users] = t.arange(label_users.numel(), device=scores.device)
        rows = row_of[users_l]
        counts = t.bincount(rows, minlength=label_users.numel())
        width = max(int(counts.max()) if counts.numel() else 0, k)
        order = t.argsort(rows, stable=True)
        start = t.cumsum(counts, 0) - counts
        pos = t.arange(order.numel(), device=scores.device) - start[rows[order]]
        mat = t.full((label_users.numel(), width), float("-inf"), device=scores.device)
        ids = t.full((label_users.numel(), width), -1, dtype=t.int64, device=scores.device)
        mat[rows[order], pos] = scores[order]
        ids[rows[order], pos] = batch[Constants.node_item].n_id[arts_l[order]]
        top = t.topk(mat, k=k, dim=1).indices
        customers.append(batch[Constants.node_user].n_id[label_users])
        predictions.append(t.gather(ids, 1, top))
    # one transfer at the end instead of two host waits per batch
    return t.cat(customers).cpu(), t.cat(predictions).cpu()


def map_to_id(customers: Tensor, predictions: Tensor, customer_id_map: Dict[str, str], article_id_map: Dict[str, str]):
    """Index -> original ids through the `*_id_map_forward.json` dictionaries (keys are stringified indices,
    run_submission.py:33-46); without a map the indices themselves are written."""
    import pandas as pd
    art = (lambda a: article_id_map[str(a)]) if article_id_map else str
    cus = (lambda c: customer_id_map[str(c)]) if customer_id_map else str
    rows = [" ".join(art(a) for a in p if a >= 0) for p in predictions.tolist()]
    return pd.DataFrame({"customer_id": [cus(c) for c in customers.tolist()], "prediction": rows})


def save_csv(df, path: str = "data/derived/submission.csv") -> None:
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    df.loc[:, ["customer_id", "prediction"]].to_csv(path, index=False)


def submission_pipeline(config: Config = link_pred_config, *, splits: dict, matchers, model_dir: str = "model/saved",
                        out_csv: str = "data/derived/submission.csv", customer_id_map: Optional[dict] = None,
                        article_id_map: Optional[dict] = None, device: str = "cuda", seed: int = 0,
                        device_sampler: bool = False):
    """splits / matchers as run_pipeline takes them; every customer of the test split is scored once, in id order.
    device_sampler: build the evaluation samples on the GPU (csrc/sampler.hip, evaluation mode) instead of with the
    host GraphDataset — the only practical way at 10^6 customers."""
    from .data.data_loader import to_undirected
    graph, users_adj, articles_adj = splits["test"]
    if device_sampler:
        from .data.device_sampler import DeviceGraphSampler
        loader = DeviceGraphSampler(config, graph, users_adj, articles_adj, device=device, seed=seed, train=False,
                                    matchers=matchers, shuffle=False)
    else:
        ds = GraphDataset(config, graph, users_adj, articles_adj, train=False, matchers=matchers, split_type="test",
                          seed=seed)
        loader = DataLoader(ds, batch_size=config.batch_size, shuffle=False)
    g_tr, u_tr, a_tr = splits["train"]
    full = to_undirected(GraphDataset(config, g_tr, u_tr, a_tr, train=True, split_type="train", seed=seed).graph)
    model = build_model(config, full, next(iter(loader)), load_model(model_dir), device)
    customers, predictions = make_predictions(model, loader, k=config.k, device=device)
    df = map_to_id(customers, predictions, customer_id_map or {}, article_id_map or {})
    save_csv(df, out_csv)
    return customers, predictions, df

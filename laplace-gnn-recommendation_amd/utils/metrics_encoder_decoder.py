"""Ranker evaluation arithmetic (reference: utils/metrics_encoder_decoder.py:9-86).  Note that the
reference compares top-k POSITIONS in the padded candidate matrix with item ids (SURVEY Appendix
A.8); that arithmetic is reproduced as is and pinned by tests/golden/metrics_universal.pt."""
from typing import List, Tuple

import torch as t
from torch import Tensor

from .metrics import NDCGatK_r, RecallPrecision_ATk
from .metrics_lightgcn import create_adj_list


def get_metrics_universal(model_output: Tensor, edge_index: Tensor, edge_label_index: Tensor,
                          exclude_edge_indices: List[Tensor], k: int) -> Tuple[float, float, float]:
    edge_index = edge_index.detach().to("cpu")
    edge_label_index = edge_label_index.detach().to("cpu")
    ratings = model_output.detach().to("cpu")
    if ratings.dim() < 2:
        ratings = ratings.unsqueeze(0)
    for excl in exclude_edge_indices:
        ratings[excl[0].to("cpu"), excl[1].to("cpu")] = -(1 << 10)
    _, top_k = t.topk(ratings, k=k)
    users = edge_label_index[0].unique(sorted=True)
    positives = create_adj_list(edge_index, users)
    r = t.stack([t.isin(top_k[i], positives[i]) for i in range(len(users))])
    recall, precision = RecallPrecision_ATk(positives, r, k)
    return recall, precision, NDCGatK_r(positives, r, k)

"""Ranker train / eval loops with the reference's signatures (training.py:19-106)."""
from typing import List, Optional, Tuple

import numpy as np
import torch as t
from torch import Tensor
from torch.nn import Module
from torch.optim import Optimizer

from .utils.constants import Constants
from .utils.get_info import select_properties
from .utils.metrics_encoder_decoder import get_metrics_universal


def _train(train_data, model: Module, optimizer: Optimizer) -> Tensor:
    x, edge_index, edge_label_index, edge_label = select_properties(train_data)
    criterion = t.nn.BCEWithLogitsLoss()
    optimizer.zero_grad()
    out = model(x, edge_index, edge_label_index).view(-1)
    loss = criterion(out, edge_label)
    loss.backward()
    optimizer.step()
    return loss


@t.no_grad()
def _test(data, model, exclude_edge_indices: list, k: int) -> Tuple[float, float]:
    x, edge_index_dict, edge_label_index, _ = select_properties(data)
    output = model.infer(x, edge_index_dict, edge_label_index)
    recall, precision, _ = get_metrics_universal(output, edge_index_dict[Constants.edge_key], edge_label_index,
                                                 exclude_edge_indices, k=k)
    return recall, precision


def train_with_dataloader(model: Module, optimizer: Optimizer, data_loader, epoch: int, device: str) -> List[float]:
    losses = []
    for data in data_loader:
        loss = _train(data.to(device), model, optimizer)
        losses.append(loss.detach().cpu().item())
    return losses


def test_with_dataloader(mode: str, model, data_loader, device: str, k: int, break_at: Optional[int]
                         ) -> Tuple[float, float]:
    recalls, precisions = [], []
    for i, data in enumerate(data_loader):
        if break_at and i == break_at:
            break
        recall, precision = _test(data.to(device), model, [], k=k)
        recalls.append(recall)
        precisions.append(precision)
    return float(np.mean(recalls)), float(np.mean(precisions))

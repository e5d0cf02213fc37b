"""Epoch drivers of the ranker.  Call surface of the reference's training.py (`train_with_dataloader`,
`test_with_dataloader`: same arguments, same return types), different plumbing:

  * the per-batch losses stay on the device and are read back ONCE per epoch — the reference synchronises the host
    after every step (training.py:75), which here would also stall the sampler that prepares the next batch on a
    side stream (data/device_sampler.py);
  * the criterion object is built once per epoch, not once per batch;
  * the iteration is ONE C call where the model has the reference's default shape (`ranker_native.py` ->
    mi_ranker_step_f32: forward, backward, dropout and Adam enqueued by a native executor); otherwise it runs without
    the autograd engine (`ranker_step.py`: the same launches, hand-derived backward, gradients handed to the caller's
    optimizer), or, failing that, with backward in the calling thread (the iteration is launch-bound: model/layers.py).

Objective and evaluation arithmetic are the reference's: BCE-with-logits (mean) on the label edges, Adam step per
batch (training.py:19-34); recall / precision @ k of `infer`'s per-user candidate matrix (training.py:37-57).
"""
from typing import Iterable, List, Optional, Tuple

import torch as t
from torch.nn import Module
from torch.optim import Optimizer

from .utils.constants import Constants
from .utils.get_info import select_properties
from .utils.metrics_encoder_decoder import get_metrics_universal


class _EpochMeter:
    """Collects 0-d device tensors; one host transfer when the values are asked for."""

    def __init__(self) -> None:
        self._vals: List[t.Tensor] = []

    def add(self, value: t.Tensor) -> None:
        self._vals.append(value.detach().reshape(()))

    def tolist(self) -> List[float]:
        return t.stack(self._vals).cpu().tolist() if self._vals else []


def _optimise_on(batch, model: Module, optimizer: Optimizer, objective) -> t.Tensor:
    features, graph, label_pairs, labels = select_properties(batch)
    optimizer.zero_grad()
    logits = model(features, graph, label_pairs).view(-1)
    value = objective(logits, labels)
    value.backward()
    optimizer.step()
    return value


def train_with_dataloader(model: Module, optimizer: Optimizer, data_loader: Iterable, epoch: int, device: str
                          ) -> List[float]:
    """One pass over `data_loader`; the mean-BCE of every batch, in order."""
    del epoch  # progress display only upstream
    objective = t.nn.BCEWithLogitsLoss()
    meter = _EpochMeter()
    # the backward graph is a chain of small custom nodes: running it in the calling thread saves the hand-over to
    # autograd's device thread at every one of them (measured: 1.54 -> 1.35 ms per iteration, tools/prof_host_ranker.py)
    from .ranker_native import NativeRankerStep
    from .ranker_step import FusedRankerStep
    fused = FusedRankerStep(model, optimizer) if (FusedRankerStep.supports(model) and model.training) else None
    # the whole iteration as one C call where the model has the reference's default shape (ranker_native.py)
    native = NativeRankerStep(model, optimizer) if (model.training and NativeRankerStep.supports(model, optimizer)) else None
    with t.autograd.set_multithreading_enabled(False):
        for batch in data_loader:
            batch = batch.to(device)
            value = None
            if native is not None:
                labelled = batch[Constants.edge_key]   # labels as the sampler wrote them (int64): the executor casts
                value = native.step(batch.x_dict, batch.edge_index_dict, labelled.edge_label_index, labelled.edge_label)
                if value is None:
                    native = None  # a batch / shape the executor declines: the op-by-op paths from here on
            if value is None and fused is not None:  # the same launches as forward + autograd, as straight-line code (ranker_step.py)
                value = fused.step(*select_properties(batch))
                if value is None:
                    fused = None   # this model / metadata is not of the fused form: autograd from here on
            if value is None:
                value = _optimise_on(batch, model, optimizer, objective)
            meter.add(value)
    return meter.tolist()


@t.no_grad()
def _rank_quality(batch, model, k: int) -> Tuple[float, float]:
    features, graph, label_pairs, _ = select_properties(batch)
    per_user_scores = model.infer(features, graph, label_pairs)
    recall, precision, _ndcg = get_metrics_universal(per_user_scores, graph[Constants.edge_key], label_pairs, [], k=k)
    return recall, precision


def test_with_dataloader(mode: str, model, data_loader: Iterable, device: str, k: int, break_at: Optional[int]
                         ) -> Tuple[float, float]:
    """Mean recall@k and precision@k over the batches of `data_loader` (at most `break_at` of them when given)."""
    del mode  # "VAL" / "TEST": progress display only upstream
    seen, recall_sum, precision_sum = 0, 0.0, 0.0
    for index, batch in enumerate(data_loader):
        if break_at and index == break_at:
            break
        r, p = _rank_quality(batch.to(device), model, k)
        recall_sum, precision_sum, seen = recall_sum + r, precision_sum + p, seen + 1
    return (recall_sum / seen, precision_sum / seen) if seen else (float("nan"), float("nan"))

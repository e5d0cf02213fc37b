"""Ranker pipeline — the flow of the reference's run_pipeline.py:24-153 (dataloaders -> model ->
lazy init on the first batch -> Adam -> train / validate per epoch -> test), without wandb and
without writing checkpoints unless asked.  The graphs are passed in (see data/data_loader.py)."""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch as t

from .config import Config, link_pred_config
from .data.data_loader import create_dataloaders
from .model.encoder_decoder import Encoder_Decoder_Model
from .model.layers import get_linear_layers, get_SAGEConv_layers
from .reporting.types import Stats
from .training import test_with_dataloader, train_with_dataloader
from .utils.get_info import get_feature_info


def _adam_kwargs(device) -> dict:
    """torch.optim.Adam(lr=...) as at run_pipeline.py:75 of the reference; on the GPU its fused multi-tensor form (one
    launch for all ~20 parameter tensors instead of a dozen foreach launches: the iteration is launch-bound)."""
    return {"fused": True} if t.device(device).type == "cuda" else {}


def run_pipeline(config: Config = link_pred_config, *, splits: dict, matchers: Optional[dict] = None,
                 device: str = "cuda", seed: int = 5, save_dir: Optional[str] = None, verbose: bool = True) -> Stats:
    if verbose:
        config.print()
    config.check_validity()
    assert config.k <= config.candidate_pool_size * 2, "k must be smaller than candidate_pool_size * 2"
    t.manual_seed(seed)  # seed_everything(5) in the reference (run_pipeline.py:30)
    np.random.seed(seed)
    train_loader, val_loader, test_loader, _, _, full_data = create_dataloaders(config, splits, matchers, seed=seed)
    first = next(iter(train_loader))
    model = Encoder_Decoder_Model(
        encoder_layers=get_SAGEConv_layers(num_layers=config.num_gnn_layers, hidden_channels=config.hidden_layer_size,
                                           out_channels=config.encoder_layer_output_size, agg_type=config.conv_agg_type),
        decoder_layers=get_linear_layers(num_layers=config.num_linear_layers,
                                         in_channels=config.encoder_layer_output_size * 2,
                                         hidden_channels=config.hidden_layer_size, out_channels=1),
        feature_info=get_feature_info(full_data), metadata=first.metadata(), embedding=True,
        heterogeneous_prop_agg_type=config.heterogeneous_prop_agg_type, batch_normalize=config.batch_norm,
        p_dropout_edges=config.p_dropout_edges, p_dropout_features=config.p_dropout_features).to(device)
    with t.no_grad():  # lazy sizes from the first batch (run_pipeline.py:72-73)
        model.initialize_encoder_input_size(first.to(device))
    optimizer = t.optim.Adam(model.parameters(), lr=config.learning_rate, **_adam_kwargs(device))

    loss_mean = float("nan")
    val_recall = val_precision = 0.0
    save_at = max(1, int(config.epochs * config.save_every))
    for epoch in range(config.epochs):
        losses = train_with_dataloader(model, optimizer, train_loader, epoch, device)
        loss_mean = float(np.mean(losses))
        if epoch % config.eval_every == 0 and matchers is not None:
            val_recall, val_precision = test_with_dataloader("VAL", model, val_loader, device, config.k,
                                                             config.evaluate_break_at)
        if verbose:
            print(f"[epoch {epoch}] loss {loss_mean:.4f} val_recall@{config.k} {val_recall:.4f} "
                  f"val_precision@{config.k} {val_precision:.4f}")
        if save_dir is not None and epoch % save_at == 0:
            os.makedirs(save_dir, exist_ok=True)
            t.save(model.state_dict(), os.path.join(save_dir, f"model_{epoch:03d}.pt"))
    test_recall = test_precision = 0.0
    if matchers is not None:
        test_recall, test_precision = test_with_dataloader("TEST", model, test_loader, device, config.k,
                                                           config.evaluate_break_at)
    return Stats(loss=loss_mean, recall_val=val_recall, recall_test=test_recall, precision_val=val_precision,
                 precision_test=test_precision)

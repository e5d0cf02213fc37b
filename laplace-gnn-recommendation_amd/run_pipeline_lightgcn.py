"""LightGCN candidate-generation pipeline — the loop of the reference's run_pipeline_lightgcn.py
(evaluation(): 20-73, train(): 76-232) with every stage on the GPU.

Differences a caller sees: the graph is passed in (edge_index + node counts) instead of being
unpickled from PyG files, and the top-`num_recommendations` dump is returned as one [U, k] tensor
(optionally saved) instead of a dict of per-user tensors built in a Python loop.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch as t
from torch import Tensor

from . import ops
from .config import LightGCNConfig, lightgcn_config
from .data.lightgcn_loader import create_dataloaders_lightgcn
from .interactions import Interactions
from .model.lightgcn import LightGCN
from .reporting.types import Stats
from .sparse import SparseTensor
from .trainer import LightGCNTrainer
from .utils.metrics_lightgcn import get_metrics_lightgcn, topk_for_users


def evaluation(model: LightGCN, edge_index: Tensor, sparse_edge_index: SparseTensor,
               exclude_edge_indices: List[Tensor], k: int, lambda_val: float, seed: int = 0
               ) -> Tuple[float, float, float, float]:
    """bpr loss over every edge of the split (one structured negative each) on the split's own
    adjacency, plus recall / precision / ndcg @ k (run_pipeline_lightgcn.py:20-73)."""
    with t.no_grad():
        users_final, users_0, items_final, items_0 = model.forward(sparse_edge_index)
        n_users, n_items = model.num_users, model.num_items
        inter = Interactions(edge_index, n_users, n_items)
        neg_range = int(edge_index[1].max())  # reference: num_nodes = max(edge_index[1])
        u, p, n = ops.sample_bpr_batch(inter.csr(), inter.row_of_edge(), inter.num_edges, neg_range, seed, 0,
                                       quirk=True, edges_in_order=True, no_self_loops=True)  # :40-44 contains_neg_self_loops=False
        final = t.cat([users_final, items_final])
        loss = ops.bpr_fwd_bwd(u, p, n, final, model.table(), n_users, lambda_val)
    recall, precision, ndcg = get_metrics_lightgcn(model, edge_index, exclude_edge_indices, k)
    return float(loss), recall, precision, ndcg


def train(config: LightGCNConfig = lightgcn_config, *, edge_index: Tensor, num_users: int, num_articles: int,
          compat: str = "reference", device: str = "cuda", seed: int = 0, save_dir: Optional[str] = None,
          verbose: bool = True) -> Stats:
    if verbose:
        config.print()
    (train_sparse, val_sparse, test_sparse, train_edges, val_edges, test_edges, all_edges, num_users,
     num_articles) = create_dataloaders_lightgcn(edge_index, num_users, num_articles, compat=compat, device=device)

    t.manual_seed(seed)
    model = LightGCN(num_users, num_articles, embedding_dim=config.hidden_layer_size,
                     num_iterations=config.num_iterations).to(device)
    model.train()
    train_inter = Interactions(train_edges, num_users, num_articles)
    # reference sampler: negatives from [0, max train item id) with its key-collision quirk
    trainer = LightGCNTrainer(model, train_sparse, train_inter, lr=config.learning_rate, Lambda=config.Lambda,
                              batch_size=config.batch_size, seed=seed, neg_range=int(train_edges[1].max()),
                              reference_sampler_quirks=(compat == "reference"))
    train_loss = t.zeros(1, device=device)
    recall = precision = 0.0
    try:   # the trainer permutes the model's rows in place: whatever happens, hand them back under their original ids
        for it in range(config.epochs):
            train_loss = trainer.step()
            if it % config.eval_every == 0:
                model.eval()
                trainer.to_original_order()  # evaluation reads the model's tables by original id; step() re-enters the training order
                val_loss, recall, precision, ndcg = evaluation(model, val_edges, val_sparse, [train_edges], config.k,
                                                               config.Lambda, seed)
                if verbose:
                    print(f"[Iter {it}/{config.epochs}] train_loss: {round(float(train_loss), 5)}, val_loss: "
                          f"{round(val_loss, 5)}, val_recall@{config.k}: {round(recall, 6)}, val_precision@{config.k}: "
                          f"{round(precision, 6)}, val_ndcg@{config.k}: {round(ndcg, 6)}")
                model.train()
            if it % config.lr_decay_every == 0 and it != 0:
                trainer.decay_lr(0.95)
    finally:
        trainer.finish()

    model.eval()
    test_loss, test_recall, test_precision, test_ndcg = evaluation(
        model, test_edges, test_sparse, [train_edges, val_edges], config.k, config.Lambda, seed)
    if verbose:
        print(f"[test_loss: {round(test_loss, 5)}, test_recall@{config.k}: {round(test_recall, 5)}, "
              f"test_precision@{config.k}: {round(test_precision, 5)}, test_ndcg@{config.k}: {round(test_ndcg, 5)}")

    # predictions for the matcher: top `num_recommendations` unseen items per user, layer-0 scores (F8)
    top_items = save_predictions(model, all_edges, config.num_recommendations, save_dir)
    return Stats(loss=float(train_loss), recall_val=recall, recall_test=test_recall, precision_val=precision,
                 precision_test=test_precision)


def save_predictions(model: LightGCN, all_edges: Tensor, num_recommendations: int,
                     save_dir: Optional[str] = None) -> Tensor:
    """[U, num_recommendations] item ids (run_pipeline_lightgcn.py:210-238)."""
    k = min(num_recommendations, model.num_items)
    users = t.arange(model.num_users, dtype=t.int64, device=all_edges.device)
    top = topk_for_users(model.users_emb.weight.detach(), model.items_emb.weight.detach(), users, all_edges, k)
    if save_dir is not None:
        import os
        os.makedirs(save_dir, exist_ok=True)
        t.save(top.cpu(), os.path.join(save_dir, "lightgcn_output.pt"))
        t.save(model.users_emb.weight.detach().cpu(), os.path.join(save_dir, "users_emb_final_lightgcn.pt"))
        t.save(model.items_emb.weight.detach().cpu(), os.path.join(save_dir, "items_emb_final_lightgcn.pt"))
    return top

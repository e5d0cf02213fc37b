"""Configuration surface — field for field the reference's config.py:22-151 (Config,
LightGCNConfig and the module-level instances the scripts import), checked against
tests/golden/config_defaults.pt.  Preprocessing / Neo4j / wandb options are accepted and ignored
by the hot path exactly as the reference ignores them."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Optional

from .utils.constants import Constants

# categorical cardinality (upper bound, as a string key) -> embedding width  (config.py:12-19)
embedding_range_dict = {"2": 2, "10": 4, "1000": 12, "10000": 20, "100000": 40, "1000000": 60}


def _dump(obj) -> None:
    print("\nConfiguration is:")
    for name, value in vars(obj).items():
        print(f"{name:>20}: {value}")
    print("\x1b[0m")


@dataclass
class Config:
    wandb_enabled: bool
    epochs: int
    hidden_layer_size: int
    encoder_layer_output_size: int
    k: int
    num_gnn_layers: int
    num_linear_layers: int
    learning_rate: float
    conv_agg_type: str                  # "add" | "mean" | "max"
    heterogeneous_prop_agg_type: str    # "sum" | "mean" | "min" | "max" | "mul"
    save_model: bool
    eval_every: int
    save_every: float

    batch_size: int                     # users per batch
    num_neighbors: int                  # fan-out cap per hop
    n_hop_neighbors: int
    num_workers: int
    candidate_pool_size: int
    positive_edges_ratio: float
    negative_edges_ratio: float
    batch_norm: bool
    matchers: str                       # "fashion" | "movielens"

    p_dropout_edges: Optional[float]
    p_dropout_features: Optional[float]

    default_edge_types: list
    other_edge_types: list
    node_types: list

    profiler: Optional[Any] = None
    evaluate_break_at: Optional[int] = None
    neo4j: bool = False

    def print(self):
        _dump(self)

    def check_validity(self):
        assert self.positive_edges_ratio <= 1.0, "Positive Edges ratio has to be smaller than 1.0"
        assert self.p_dropout_edges <= 1.0, "p_dropout_edges cannot be bigger than 1.0"
        assert self.p_dropout_features <= 1.0, "p_dropout_features cannot be bigger than 1.0"


@dataclass
class LightGCNConfig:
    epochs: int                 # counts ITERATIONS (run_pipeline_lightgcn.py:117)
    hidden_layer_size: int
    k: int
    learning_rate: float
    save_model: bool
    eval_every: int
    lr_decay_every: int
    Lambda: float
    batch_size: int
    num_iterations: int
    show_graph: bool
    num_recommendations: int

    def print(self):
        _dump(self)


link_pred_config = Config(
    matchers="movielens", wandb_enabled=False, epochs=4, k=12,
    num_gnn_layers=2, num_linear_layers=2, hidden_layer_size=128, encoder_layer_output_size=64,
    conv_agg_type="add", heterogeneous_prop_agg_type="sum", learning_rate=0.01, save_model=False,
    batch_size=24, num_neighbors=64, n_hop_neighbors=3, num_workers=1, candidate_pool_size=20,
    positive_edges_ratio=0.5, negative_edges_ratio=3.0, eval_every=1, save_every=0.2,
    profiler=None, evaluate_break_at=None, p_dropout_edges=0.2, p_dropout_features=0.3,
    batch_norm=True, neo4j=True,
    default_edge_types=[Constants.edge_key], other_edge_types=[],
    node_types=[Constants.node_user, Constants.node_item],
)

lightgcn_config = LightGCNConfig(
    epochs=10000, k=12, hidden_layer_size=32, learning_rate=1e-3, save_model=False,
    batch_size=128, num_iterations=4, eval_every=100, lr_decay_every=100, Lambda=1e-6,
    show_graph=False, num_recommendations=256,
)

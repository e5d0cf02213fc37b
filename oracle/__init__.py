"""ORACLE — test infrastructure, not product.

CPU restatement of the reference's hot path (dream-faster/laplace-gnn-recommendation), used only
as the checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under
laplace-gnn-recommendation_amd/ imports it.

Pinning status:
  * bpr_loss, make_predictions_for_user, recall/precision/NDCG, padded_stack, get_linear_layers,
    Config defaults: PINNED — golden vectors in tests/golden/ were produced by importing the
    reference's own modules (tests/make_golden.py).
  * LightGCN.forward / gcn_norm / SpMM / SAGEConv / structured_negative_sampling: the arithmetic
    lives in torch_sparse / torch_scatter / torch_geometric, which are neither vendored under
    /root/reference nor installable here (environment.yml:28-30, unpinned, ~PyG 2.0.4 /
    torch-sparse 0.6.13).  Restated from their published algorithms and anchored on dense
    float64 known-answer math and the reference's call sites: PARITY UNPINNED for those rows.
"""

"""Generates tests/golden/*.pt by IMPORTING the reference's own (importable) modules.

Run in the build container only (needs /root/reference):  python tests/make_golden.py
The fixtures hold inputs and the reference's outputs — data, no reference source text.  They pin
the oracle (oracle/*.py) for the rows of SURVEY §8 whose arithmetic lives in the reference's own
files: bpr_loss (+ its autograd gradients), make_predictions_for_user, get_metrics_lightgcn,
RecallPrecision_ATk / NDCGatK_r, get_metrics_universal, padded_stack, difference_1d,
get_linear_layers, Config / LightGCNConfig defaults, embedding_range_dict, and (N2) the chronological
split + adjacency dicts of run_data_splitting.py / utils/preprocessing.py.
"""
import dataclasses
import os
import sys
from types import SimpleNamespace

import torch as t

REF = os.environ.get("LAPLACE_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def main() -> None:
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    os.makedirs(OUT, exist_ok=True)
    from utils.metrics_lightgcn import bpr_loss, make_predictions_for_user, get_metrics_lightgcn, create_adj_dict
    from utils.metrics import RecallPrecision_ATk, NDCGatK_r
    from utils.metrics_encoder_decoder import get_metrics_universal
    from utils.tensor import padded_stack, difference_1d
    from model.layers import get_linear_layers
    import config as ref_config

    # ---- bpr_loss + gradients ------------------------------------------------------------
    cases = []
    for seed, (B, D, lam) in enumerate([(16, 8, 1e-6), (128, 64, 1e-6), (33, 32, 1e-2), (5, 128, 0.0)]):
        g = t.Generator().manual_seed(100 + seed)
        ins = [(t.randn(B, D, generator=g) * (0.1 if seed % 2 == 0 else 1.0)).requires_grad_(True) for _ in range(6)]
        loss = bpr_loss(*ins, lam)
        grads = t.autograd.grad(loss, ins)
        cases.append({"inputs": [x.detach().clone() for x in ins], "lambda": lam,
                      "loss": loss.detach().clone(), "grads": [x.clone() for x in grads]})
    t.save(cases, os.path.join(OUT, "bpr_loss.pt"))

    # ---- make_predictions_for_user / get_metrics_lightgcn ---------------------------------
    g = t.Generator().manual_seed(7)
    U, I, D = 40, 300, 16
    ue, ie = t.randn(U, D, generator=g) * 0.1, t.randn(I, D, generator=g) * 0.1
    n_tr, n_ev = 400, 120
    tr = t.stack([t.randint(0, U, (n_tr,), generator=g), t.randint(0, I, (n_tr,), generator=g)])
    tr = t.unique(tr, dim=1)
    ev = t.stack([t.randint(0, U, (n_ev,), generator=g), t.randint(0, I, (n_ev,), generator=g)])
    ev = t.unique(ev, dim=1)
    excl = create_adj_dict(tr)
    preds = {}
    for k in (1, 12, 50):
        preds[k] = {u: make_predictions_for_user(ue, ie, u, excl, k).clone() for u in range(U)}
    model = SimpleNamespace(users_emb=SimpleNamespace(weight=ue), items_emb=SimpleNamespace(weight=ie))
    metrics = {k: get_metrics_lightgcn(model, ev, [tr], k) for k in (5, 12)}
    t.save({"users_emb": ue, "items_emb": ie, "train_edges": tr, "eval_edges": ev,
            "excl": {int(k): v.clone() for k, v in excl.items()}, "preds": preds, "metrics": metrics},
           os.path.join(OUT, "topk_metrics.pt"))

    # ---- recall / precision / ndcg ----------------------------------------------------------
    g = t.Generator().manual_seed(11)
    k = 12
    gt = [t.randint(0, 100, (int(n),), generator=g) for n in t.randint(1, 20, (25,), generator=g)]
    r = (t.rand(25, k, generator=g) < 0.2)
    rp = RecallPrecision_ATk(gt, r, k)
    nd = NDCGatK_r(gt, r, k)
    t.save({"groundTruth": gt, "r": r, "k": k, "recall_precision": rp, "ndcg": nd}, os.path.join(OUT, "rank_metrics.pt"))

    # ---- get_metrics_universal (ranker eval arithmetic) -------------------
    g = t.Generator().manual_seed(13)
    n_users, n_cand = 6, 20
    out = t.randn(n_users, n_cand, generator=g)
    eli = t.stack([t.arange(n_users).repeat_interleave(n_cand), t.randint(0, 50, (n_users * n_cand,), generator=g)])
    ei = t.stack([t.randint(0, n_users, (40,), generator=g), t.randint(0, 50, (40,), generator=g)])
    mu = get_metrics_universal(out.clone(), ei, eli, [], k=12)
    t.save({"model_output": out, "edge_index": ei, "edge_label_index": eli, "k": 12, "metrics": mu},
           os.path.join(OUT, "metrics_universal.pt"))

    # ---- padded_stack / difference_1d ----------------------------------------------------------
    g = t.Generator().manual_seed(17)
    tens = [t.randn(int(n), generator=g) for n in (3, 7, 1, 5)]
    t.save({"tensors": tens, "right": padded_stack(tens, value=-(1 << 50)), "left": padded_stack(tens, side="left", value=0.5),
            "a": t.tensor([9, 3, 7, 1, 5, 8]), "b": t.tensor([7, 8, 2]),
            "diff": difference_1d(t.tensor([9, 3, 7, 1, 5, 8]), t.tensor([7, 8, 2]), assume_unique=True)},
           os.path.join(OUT, "tensor_utils.pt"))

    # ---- get_linear_layers ---------------------------------------------------------------------
    layers = {}
    for (n, i, h, o) in [(1, 128, 128, 1), (2, 128, 128, 1), (4, 64, 32, 8)]:
        t.manual_seed(23)
        ml = get_linear_layers(n, i, h, o)
        layers[(n, i, h, o)] = {"shapes": [(m.in_features, m.out_features, m.bias is not None) for m in ml],
                                "state": [{k: v.clone() for k, v in m.state_dict().items()} for m in ml]}
    t.save(layers, os.path.join(OUT, "linear_layers.pt"))

    # ---- config defaults -------------------------------------------------------------------------
    def plain(v):
        if isinstance(v, (int, float, str, bool, type(None))):
            return v
        if isinstance(v, (list, tuple)):
            return [plain(x) for x in v]
        return repr(v)
    cfg = {
        "Config_fields": [f.name for f in dataclasses.fields(ref_config.Config)],
        "LightGCNConfig_fields": [f.name for f in dataclasses.fields(ref_config.LightGCNConfig)],
        "link_pred_config": {k: plain(v) for k, v in vars(ref_config.link_pred_config).items()},
        "lightgcn_config": {k: plain(v) for k, v in vars(ref_config.lightgcn_config).items()},
        "embedding_range_dict": dict(ref_config.embedding_range_dict),
    }
    t.save(cfg, os.path.join(OUT, "config_defaults.pt"))
    print("golden fixtures written to", OUT)
    for f in sorted(os.listdir(OUT)):
        print(f"  {f}: {os.path.getsize(os.path.join(OUT, f))} bytes")




def time_split_golden() -> None:
    """tests/golden/time_split.pt: the reference's train_test_split_by_time (run_data_splitting.py:36-52) and
    extract_edges / extract_reverse_edges (utils/preprocessing.py:84-89) on a seeded transaction table."""
    sys.path.insert(0, REF)
    import numpy as np
    import pandas as pd
    from run_data_splitting import train_test_split_by_time
    from utils.preprocessing import extract_edges, extract_reverse_edges
    rng = np.random.default_rng(0)
    n = 400
    df = pd.DataFrame({"customer_id": rng.integers(0, 60, n), "article_id": rng.integers(0, 45, n),
                       "timestamp": np.sort(rng.integers(0, 10 ** 6, n))})
    df.loc[df.customer_id == 7, "customer_id"] = 8
    df.loc[df.index[:3], "customer_id"] = [58, 58, 59]
    out = train_test_split_by_time(df.copy(), "customer_id")
    if out.index.nlevels > 1:  # pandas >= 2 prepends the group key to the index; the reference's era did not
        out = out.reset_index(level=0, drop=True).sort_index()
    tr = out[out["train_mask"] == True]  # noqa: E712
    t.save({"customer_id": df.customer_id.to_numpy(), "article_id": df.article_id.to_numpy(),
            "train_mask": out.train_mask.to_numpy(), "val_mask": out.val_mask.to_numpy(),
            "test_mask": out.test_mask.to_numpy(), "edges_train": extract_edges(tr),
            "rev_edges_train": extract_reverse_edges(tr)}, os.path.join(OUT, "time_split.pt"))


def matchers_golden() -> None:
    """tests/golden/matchers.pt: the reference's OWN matcher classes (data/matching/*.py) run on small dict-of-list
    files written into a temporary cwd in the layout their constructors read (data/derived/edges_<split>.pt, ...).
    Stored: the inputs (plain dicts / lists) and get_matches(user) for every user that has purchases, several k."""
    import tempfile
    import numpy as np
    sys.path.insert(0, REF)
    from data.matching import LightGCNMatcher, PopularItemsMatcher, UsersSameLocationMatcher, UsersWithCommonItemsMatcher
    rng = np.random.default_rng(5)
    U, A, E, L = 70, 45, 420, 9
    cu, ca = rng.integers(0, U, E), rng.integers(0, A, E)
    cu[cu == 11] = 12                                  # user 11 buys nothing; duplicates (repeat purchases) are kept
    ca[:40] = 3                                        # a hub article
    edges, rev = {}, {}
    for u, a in zip(cu.tolist(), ca.tolist()):         # list order = transaction order (utils/preprocessing.py:84-89)
        edges.setdefault(u, []).append(a)
        rev.setdefault(a, []).append(u)
    popular = [int(x) for x in np.argsort(-np.bincount(ca, minlength=A), kind="stable")]
    g = t.Generator().manual_seed(3)
    top = t.stack([t.randperm(A, generator=g)[:20] for _ in range(U)])
    location_for_user = {u: int(rng.integers(0, L)) for u in range(U)}
    customers_per_location = {}
    for u in range(U):
        customers_per_location.setdefault(location_for_user[u], []).append(u)
    cwd = os.getcwd()
    out = {"edges": edges, "rev_edges": rev, "popular": popular, "lightgcn_top": top, "location_for_user": location_for_user,
           "customers_per_location": customers_per_location, "num_users": U, "num_articles": A, "matches": {}}
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "data", "derived"))
        d = os.path.join(tmp, "data", "derived")
        t.save(edges, os.path.join(d, "edges_train.pt"))
        t.save(rev, os.path.join(d, "rev_edges_train.pt"))
        t.save(popular, os.path.join(d, "most_popular_products.pt"))
        t.save(top, os.path.join(d, "lightgcn_output.pt"))
        t.save(location_for_user, os.path.join(d, "location_for_user.pt"))
        t.save(customers_per_location, os.path.join(d, "customers_per_location.pt"))
        os.chdir(tmp)
        try:
            for k in (1, 7, 50, 300):
                common = UsersWithCommonItemsMatcher(k, "train")
                loc = UsersSameLocationMatcher(k, "train")
                pop = PopularItemsMatcher(k)
                lg = LightGCNMatcher(k)
                res = {"common": {}, "location": {}, "popular": pop.get_matches(0).clone(), "lightgcn": {}}
                for u in range(U):
                    res["lightgcn"][u] = lg.get_matches(u).clone()
                    if u in edges:                      # the reference raises KeyError for a user without purchases
                        res["common"][u] = common.get_matches(u).clone()
                    # same-location: every customer at the location must have purchases or the reference raises
                    if all(v in edges for v in customers_per_location[location_for_user[u]]):
                        res["location"][u] = loc.get_matches(u).clone()
                out["matches"][k] = res
        finally:
            os.chdir(cwd)
    t.save(out, os.path.join(OUT, "matchers.pt"))
    n_loc = len(out["matches"][7]["location"])
    print(f"  matchers.pt: {os.path.getsize(os.path.join(OUT, 'matchers.pt'))} bytes ({len(out['matches'][7]['common'])} common-item users, {n_loc} same-location users)")


if __name__ == "__main__":
    if "--only-matchers" not in sys.argv:
        main()
        time_split_golden()
    matchers_golden()

"""recall / precision / NDCG @ k — same definitions as the reference's utils/metrics.py:6-57,
checked against tests/golden/rank_metrics.pt."""
from typing import List, Tuple

import torch as t
from torch import Tensor


def RecallPrecision_ATk(groundTruth: List[Tensor], r: Tensor, k: int) -> Tuple[float, float]:
    """r[u, j] says whether the j-th recommendation of user u is in groundTruth[u].
    recall = mean_u(hits_u / |GT_u|); precision = mean_u(hits_u) / k."""
    hits = r.sum(dim=-1).float()
    liked = t.tensor([float(len(g)) for g in groundTruth])
    return (hits / liked).mean().item(), (hits.mean() / k).item()


def NDCGatK_r(groundTruth: List[Tensor], r: Tensor, k: int) -> float:
    """DCG of r against the ideal DCG of min(|GT_u|, k) leading hits, log2 discount; 0/0 -> 0."""
    assert len(r) == len(groundTruth)
    discount = 1.0 / t.log2(t.arange(2, k + 2))
    n_ideal = t.tensor([min(len(g), k) for g in groundTruth])
    ideal = (t.arange(k)[None, :] < n_ideal[:, None]).float()
    idcg = (ideal * discount).sum(dim=1)
    dcg = (r * discount).sum(dim=1)
    idcg[idcg == 0.0] = 1.0
    ndcg = dcg / idcg
    ndcg[t.isnan(ndcg)] = 0.0
    return ndcg.mean().item()


def MAPatK(groundTruth: List[Tensor], predictions: Tensor, k: int = 12) -> float:
    """Mean average precision @ k as the H&M Kaggle competition scores a submission (BASELINE.json's metric
    line names it; the reference itself reports only recall / precision / NDCG):
        AP_u = (1 / min(|GT_u|, k)) * sum_{j<=k} P_u(j) * rel_u(j),   MAP = mean over users with |GT_u| > 0,
    P_u(j) = hits among the first j predictions / j; a repeated prediction counts once.  predictions
    [n_users, >= k] int64, -1 = no prediction."""
    total, n = 0.0, 0
    for gt, pred in zip(groundTruth, predictions):
        truth = set(int(x) for x in gt.tolist())
        if not truth:
            continue
        seen, hits, ap = set(), 0, 0.0
        for j, a in enumerate(pred[:k].tolist()):
            if a >= 0 and a in truth and a not in seen:
                hits += 1
                ap += hits / (j + 1.0)
            seen.add(a)
        total += ap / min(len(truth), k)
        n += 1
    return total / max(n, 1)

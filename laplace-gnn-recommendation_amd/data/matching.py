"""Candidate matchers for evaluation (reference: data/matching/*.py, `Matcher.get_matches(user_id) ->
LongTensor`).  Same outputs, computed over the CSR adjacency instead of dicts of lists, and lazily:
the reference materialises every article of every co-purchasing user and then keeps the first k.

  LightGCNMatcher              first k of the user's LightGCN top-N row (data/matching/lightgcn.py:5-11);
                               fed directly by run_pipeline_lightgcn.save_predictions' [U, N] tensor
  PopularItemsMatcher          first k most popular items (data/matching/fashion/popular_items.py)
  UsersWithCommonItemsMatcher  first k entries of: for each article of the user (list order), for each
                               user of that article (list order, the user itself included), all their
                               articles (data/matching/users_with_common_purchases.py:14-26)

  UsersSameLocationMatcher     first k entries of: for each customer at the user's location (list order, the user
                               itself included), all their articles (data/matching/fashion/users_same_location.py:8-25;
                               commented out of the reference's get_matchers, kept here the same way)

All four are pinned against the reference's own classes: tests/golden/matchers.pt (tests/make_golden.py runs them on
small dict-of-list files) — host forms in tests/test_oracle_golden.py, device forms in tests/test_gpu_matching.py.

Device forms: every matcher also answers for ALL users at once as a device tensor [U, k] (int64, -1 = no
proposal) through `matches_for_all_device(num_users, device)` — the common-purchase expansion is the HIP kernel
mi_match_common_items_i32, popularity is a device argsort of the degrees, the LightGCN rows are a slice of the top-K
dump that is already on the device — and `device_sampler.candidate_csr_device` turns them into the candidate CSR the
evaluation sampler reads, with no per-user host loop.  The host `get_matches` stays as the checker.
"""
from __future__ import annotations

from typing import List

import numpy as np
import torch as t
from torch import Tensor

from .dataset import AdjList


class Matcher:
    def get_matches(self, user_id: int) -> Tensor:
        raise NotImplementedError


class LightGCNMatcher(Matcher):
    def __init__(self, top_articles_per_user: Tensor, k: int):
        self._top_dev = top_articles_per_user if top_articles_per_user.is_cuda else None  # save_predictions' dump, kept where it is
        self.top, self.k = top_articles_per_user.cpu(), int(k)

    def get_matches(self, user_id: int) -> Tensor:
        row = self.top[user_id][: self.k]
        return row[row >= 0]

    def matches_for_all(self, num_users: int):
        """[num_users, k] proposals at once (-1 = none): what the device sampler's evaluation mode consumes."""
        return self.top[:num_users, : self.k].numpy()

    def matches_for_all_device(self, num_users: int, device) -> Tensor:
        src = self._top_dev if getattr(self, "_top_dev", None) is not None else self.top
        return src[:num_users, : self.k].to(device=device, dtype=t.int64)


class PopularItemsMatcher(Matcher):
    def __init__(self, popular_items, k: int):
        self.popular_items, self.k = t.as_tensor(popular_items).to(t.long), int(k)

    @classmethod
    def from_adjacency(cls, articles_adj, k: int) -> "PopularItemsMatcher":
        adj = AdjList(articles_adj) if not isinstance(articles_adj, AdjList) else articles_adj
        deg = np.diff(adj.ptr)
        order = np.argsort(-deg, kind="stable")
        return cls(order.copy(), k)

    def get_matches(self, user_id: int) -> Tensor:
        return self.popular_items[: self.k].cpu()

    def matches_for_all(self, num_users: int):
        return np.broadcast_to(self.popular_items[: self.k].cpu().numpy(), (num_users, min(self.k, self.popular_items.numel())))

    @classmethod
    def from_degrees_device(cls, article_degrees: Tensor, k: int) -> "PopularItemsMatcher":
        """Most popular first, ties by id (stable), computed where the degrees live."""
        return cls(t.argsort(article_degrees.to(t.int64), descending=True, stable=True), k)

    def matches_for_all_device(self, num_users: int, device) -> Tensor:
        row = self.popular_items[: self.k].to(device)
        return row[None, :].expand(num_users, row.numel())


class UsersWithCommonItemsMatcher(Matcher):
    def __init__(self, users_adj, articles_adj, k: int):
        self.users = users_adj if isinstance(users_adj, AdjList) else AdjList(users_adj)
        self.articles = articles_adj if isinstance(articles_adj, AdjList) else AdjList(articles_adj)
        self.k = int(k)

    def matches_for_all_device(self, num_users: int, device) -> Tensor:
        from .. import ops
        dev = t.device(device)
        if dev.type == "cuda" and dev.index is None:       # "cuda" and "cuda:0" are the same place: compare normalised devices
            dev = t.device("cuda", t.cuda.current_device())
        if num_users > len(self.users):
            raise IndexError(f"{num_users} query users but the purchase lists cover {len(self.users)}")
        if getattr(self, "_dev", None) is None or self._dev[0].device != dev:
            to32 = lambda a: t.from_numpy(np.ascontiguousarray(a.astype(np.int32))).to(dev)
            self._dev = (to32(self.users.ptr), to32(self.users.idx), to32(self.articles.ptr), to32(self.articles.idx))
        out, _ = ops.match_common_items(*self._dev, self.k, n_queries=num_users)
        return out.to(t.int64)

    def get_matches(self, user_id: int) -> Tensor:
        out: List[np.ndarray] = []
        have = 0
        for a in self.users[user_id]:
            for v in self.articles[int(a)]:
                lst = self.users[int(v)]
                out.append(lst)
                have += len(lst)
                if have >= self.k:
                    return t.from_numpy(np.concatenate(out)[: self.k].astype(np.int64))
        return t.from_numpy(np.concatenate(out).astype(np.int64)) if out else t.empty(0, dtype=t.long)


class UsersSameLocationMatcher(Matcher):
    def __init__(self, customers_per_location, location_for_user, users_adj, k: int):
        """customers_per_location: dict location -> [customer ids] (customers_per_location.pt); location_for_user: dict
        or array customer -> location (location_for_user.pt); users_adj: edges_<split>.pt."""
        self.users = users_adj if isinstance(users_adj, AdjList) else AdjList(users_adj)
        n_users = len(self.users)
        if isinstance(location_for_user, dict):
            loc = np.full(n_users, -1, dtype=np.int64)
            for u, l in location_for_user.items():
                if 0 <= int(u) < n_users:
                    loc[int(u)] = int(l)
        else:
            loc = np.asarray(location_for_user, dtype=np.int64)
        self.location_for_user = loc
        n_loc = max([int(loc.max()) + 1 if loc.size else 0] + [int(l) + 1 for l in customers_per_location])
        self.customers = AdjList({int(l): list(v) for l, v in customers_per_location.items()}, n_loc)
        self.k = int(k)
        # the device kernel indexes the customers' purchase lists and the location table with these ids unchecked: validate once
        idx = np.asarray(self.customers.idx)
        if idx.size and (int(idx.min()) < 0 or int(idx.max()) >= n_users):
            raise IndexError(f"customers_per_location names customer {int(idx.max())} but there are {n_users} customers")
        if loc.size and int(loc.max()) >= n_loc:
            raise IndexError("location_for_user names a location beyond customers_per_location")

    def get_matches(self, user_id: int) -> Tensor:
        loc = int(self.location_for_user[user_id])
        out: List[np.ndarray] = []
        have = 0
        if loc >= 0:
            for v in self.customers[loc]:
                lst = self.users[int(v)]
                out.append(lst)
                have += len(lst)
                if have >= self.k:
                    break
        return t.from_numpy(np.concatenate(out)[: self.k].astype(np.int64)) if out else t.empty(0, dtype=t.long)

    def matches_for_all_device(self, num_users: int, device) -> Tensor:
        from .. import ops
        dev = t.device(device)
        if dev.type == "cuda" and dev.index is None:       # "cuda" and "cuda:0" are the same place: compare normalised devices
            dev = t.device("cuda", t.cuda.current_device())
        if num_users > self.location_for_user.size:
            raise IndexError(f"{num_users} query users but location_for_user has {self.location_for_user.size} entries")
        if getattr(self, "_dev", None) is None or self._dev[0].device != dev:
            to32 = lambda a: t.from_numpy(np.ascontiguousarray(np.asarray(a).astype(np.int32))).to(dev)
            self._dev = (to32(self.location_for_user), to32(self.customers.ptr), to32(self.customers.idx),
                         to32(self.users.ptr), to32(self.users.idx))
        out, _ = ops.match_same_location(*self._dev, self.k, n_queries=num_users)
        return out.to(t.int64)


def get_matchers(dataset_type: str, users_adj, articles_adj, candidate_pool_size: int) -> List[Matcher]:
    """data/matching/__init__.py:9-24."""
    if dataset_type == "movielens":
        return [UsersWithCommonItemsMatcher(users_adj, articles_adj, candidate_pool_size)]
    if dataset_type == "fashion":
        return [PopularItemsMatcher.from_adjacency(articles_adj, candidate_pool_size),
                UsersWithCommonItemsMatcher(users_adj, articles_adj, candidate_pool_size)]
    raise ValueError("Unknown matchers type: {}".format(dataset_type))

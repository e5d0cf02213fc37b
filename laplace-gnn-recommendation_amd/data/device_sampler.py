"""On-device N-hop sampler: whole batches of the ranker's training subgraphs are
sampled, relabelled and collated on the GPU (csrc/sampler.hip) — the mini-batch never exists on
the host.  Same algorithm as `GraphDataset.__getitem__` (data/dataset.py of the reference) + the
PyG collate; draws come from a counter-based Philox stream (oracle/sampler_ref.py mirrors it)."""
from __future__ import annotations

import ctypes
import os
import math
from typing import Iterator, Optional

import numpy as np
import torch as t
from torch import Tensor

from .. import _lib
from .._lib import SamplerDesc, check
from ..hetero import HeteroData
from ..utils.constants import Constants
from .dataset import AdjList


def candidate_csr(matchers, num_users: int):
    """(ptr int64[U + 1], idx int64[]) of cat(m.get_matches(u) for m in matchers) per user.  Matchers that can
    answer for all users at once (`matches_for_all`) are not looped over."""
    dense = [m.matches_for_all(num_users) if hasattr(m, "matches_for_all") else None for m in matchers]
    if matchers and all(d is not None for d in dense):  # all vectorised: concatenate per user, drop the -1 pads
        cat = np.concatenate([np.asarray(d, dtype=np.int64) for d in dense], axis=1)
        keep = cat >= 0
        ptr = np.concatenate([[0], np.cumsum(keep.sum(axis=1))]).astype(np.int64)
        return ptr, cat[keep]
    per_user_counts = np.zeros(num_users, dtype=np.int64)
    blocks = []
    for m, d in zip(matchers, dense):
        if d is not None:  # [U, k] int64, -1 = no proposal
            d = np.asarray(d, dtype=np.int64)
            keep = d >= 0
            blocks.append((np.repeat(np.arange(num_users), d.shape[1])[keep.reshape(-1)], d[keep]))
            per_user_counts += keep.sum(axis=1)
        else:
            us, its = [], []
            for u in range(num_users):
                got = np.asarray(m.get_matches(u)).astype(np.int64).reshape(-1)
                us.append(np.full(got.shape[0], u, dtype=np.int64))
                its.append(got)
                per_user_counts[u] += got.shape[0]
            blocks.append((np.concatenate(us), np.concatenate(its)))
    users = np.concatenate([b[0] for b in blocks]) if blocks else np.empty(0, dtype=np.int64)
    items = np.concatenate([b[1] for b in blocks]) if blocks else np.empty(0, dtype=np.int64)
    order = np.argsort(users, kind="stable")
    ptr = np.concatenate([[0], np.cumsum(per_user_counts)]).astype(np.int64)
    return ptr, items[order]


def candidate_csr_device(matchers, num_users: int, device):
    """candidate_csr with every stage on the device: (ptr int32[U + 1], idx int32[]) tensors, or None when a matcher
    has no device form.  Row u = cat(m.get_matches(u) for m in matchers), -1 pads dropped, order kept."""
    if not matchers or not all(hasattr(m, "matches_for_all_device") for m in matchers):
        return None
    cat = t.cat([m.matches_for_all_device(num_users, device) for m in matchers], dim=1)
    keep = cat >= 0
    ptr = t.zeros(num_users + 1, dtype=t.int64, device=device)
    t.cumsum(keep.sum(dim=1), dim=0, out=ptr[1:])
    return ptr.to(t.int32), cat[keep].to(t.int32)


class DeviceGraphSampler:
    def __init__(self, config, graph: HeteroData, users_adj_list, articles_adj_list, batch_size: Optional[int] = None,
                 randomization: bool = True, device: str = "cuda", seed: int = 0, prefetch: bool = True,
                 train: bool = True, matchers=None, shuffle: bool = True, emit_csr: bool = True):
        """emit_csr: also write each batch's edges as the two sorted CSRs the encoder consumes (mi_sampler_emit_csr),
        attached to edge_index as `_sorted_csr`; the encoder then builds no CSR of its own for the batch.
        train=False: evaluation samples as GraphDataset(train=False) builds them — the label-0 edges are the
        matchers' candidates (minus purchases, plus purchases no matcher proposed; data/dataset.py:94-105)."""
        self.config, self.device, self.seed = config, t.device(device), int(seed)
        self.prefetch, self._side = ("thread" if prefetch == "thread" else bool(prefetch)), None
        self.train, self.shuffle = bool(train), bool(shuffle)
        self.emit_csr = bool(emit_csr) and int(config.n_hop_neighbors) * int(config.num_neighbors) <= 512
        self.batch_size = int(batch_size if batch_size is not None else config.batch_size)
        self.randomization = randomization
        ux, ax = graph[Constants.node_user].x, graph[Constants.node_item].x
        n_users, n_articles = ux.shape[0], ax.shape[0]
        users, articles = AdjList(users_adj_list, n_users), AdjList(articles_adj_list, n_articles)
        ei = graph[Constants.edge_key].edge_index
        self.num_edges, self.id_max = int(ei.shape[1]), int(ei[1].max())
        dev = self.device
        to32 = lambda a: t.from_numpy(np.ascontiguousarray(a.astype(np.int32))).to(dev)
        self.uptr, self.uidx = to32(users.ptr), to32(users.idx)
        self.aptr, self.aidx = to32(articles.ptr), to32(articles.idx)
        self.user_x, self.article_x = ux.to(dev), ax.to(dev)
        self.num_users, self.num_articles = n_users, n_articles
        max_deg = int(np.diff(users.ptr).max()) if n_users else 1
        self.max_pos = max(2, math.floor(max_deg * config.positive_edges_ratio) if randomization else 2)
        self.max_neg = max(1, config.k - 1, int(config.negative_edges_ratio * self.max_pos))
        self.cptr = self.cidx = None
        if not self.train:
            assert matchers is not None, "Must provide matchers for test"
            on_dev = candidate_csr_device(matchers, n_users, dev) if dev.type == "cuda" else None
            if on_dev is not None:  # N3: proposals computed and kept on the device
                self.cptr, self.cidx = on_dev
                widest = int((self.cptr[1:] - self.cptr[:-1]).max()) if n_users else 0
                if self.cidx.numel() == 0:
                    self.cidx = t.zeros(1, dtype=t.int32, device=dev)
            else:
                cptr, cidx = candidate_csr(matchers, n_users)
                self.cptr, self.cidx = to32(cptr), to32(cidx if cidx.size else np.zeros(1, dtype=np.int64))
                widest = int(np.diff(cptr).max())
            self.max_neg = max(self.max_neg, widest + max_deg)
        self._desc = self._make_desc(self.batch_size)
        self._ws = t.empty(int(_lib.lib().mi_sampler_workspace_bytes(ctypes.byref(self._desc))), dtype=t.uint8, device=dev)
        self.step = 0

    def _make_desc(self, batch: int) -> SamplerDesc:
        c = self.config
        return SamplerDesc(batch, int(c.n_hop_neighbors), int(c.num_neighbors), int(c.k), 1 if self.randomization else 0,
                           self.max_pos, self.max_neg, 0, self.num_users, self.num_articles, self.num_edges, self.id_max,
                           self.uptr.data_ptr(), self.uidx.data_ptr(), self.aptr.data_ptr(), self.aidx.data_ptr(),
                           float(c.positive_edges_ratio), float(c.negative_edges_ratio),
                           int(getattr(c, "reject_min_entries", 0) or 0),
                           self.cptr.data_ptr() if self.cptr is not None else None,
                           self.cidx.data_ptr() if self.cidx is not None else None)

    def __len__(self) -> int:
        return (self.num_users + self.batch_size - 1) // self.batch_size

    def sample(self, seed_users: Tensor, step: Optional[int] = None, raw: bool = False):
        """Collated batch for `seed_users` (int64, any device). raw=True returns the flat arrays."""
        if step is None:
            step = self.step
            self.step += 1
        seeds = seed_users.to(self.device, t.int64).contiguous()
        B = seeds.numel()
        if B > self.batch_size:
            raise ValueError("more seed users than the sampler's batch size")
        desc = self._desc if B == self.batch_size else self._make_desc(B)
        L = _lib.lib()
        stream = _lib.current_stream()
        totals = (ctypes.c_int64 * 4)()
        check(L.mi_sampler_count(ctypes.byref(desc), seeds.data_ptr(), self.seed & (2**64 - 1), int(step) & (2**64 - 1),
                                 self._ws.data_ptr(), self._ws.numel(), totals, stream), "mi_sampler_count")
        return self._emit(seeds, desc, [int(x) for x in totals], stream, raw)

    def _emit(self, seeds: Tensor, desc: SamplerDesc, totals, stream: int, raw: bool = False, ws: Optional[Tensor] = None):
        """Phase B on `stream` (the current torch stream must be that stream: the feature gathers follow it).  ws: the
        workspace phase A of this batch ran in (default: the sampler's own)."""
        ws = self._ws if ws is None else ws
        nu, na, ne, nl = totals
        B = seeds.numel()
        L = _lib.lib()
        tot = (ctypes.c_int64 * 4)(nu, na, ne, nl)
        dev = self.device
        # ONE int64 block and ONE int32 block per batch, the outputs are views of them (round 4: 18 allocations per batch and
        # 14 record_stream calls were a visible part of a host-bound loop).  edge_index / edge_label_index are written as
        # [3, n] — row 2 repeats row 0 — so the reversed relation's index (data/dataset.py: flip(0)) is rows 1..2 of the same
        # buffer: a view, not two flip launches.
        r4 = lambda n: (n + 3) & ~3                      # every piece starts on a 32-byte (int64) / 16-byte (int32) boundary
        sizes = (r4(nu), r4(na), r4(3 * ne), r4(3 * nl), r4(nl), r4(B + 1), r4(B + 1))
        blk = t.empty(sum(sizes), dtype=t.int64, device=dev)
        off, parts = 0, []
        for n_alloc in sizes:
            parts.append(off)
            off += n_alloc
        user_ids, article_ids = blk[parts[0]:parts[0] + nu], blk[parts[1]:parts[1] + na]
        edge3 = blk[parts[2]:parts[2] + 3 * ne].view(3, ne)
        label3 = blk[parts[3]:parts[3] + 3 * nl].view(3, nl)
        labels = blk[parts[4]:parts[4] + nl]
        user_ptr, article_ptr = blk[parts[5]:parts[5] + B + 1], blk[parts[6]:parts[6] + B + 1]
        edge_index, label_index = edge3[0:2], label3[0:2]
        check(L.mi_sampler_emit3(ctypes.byref(desc), seeds.data_ptr(), ws.data_ptr(), ws.numel(), tot,
                                 user_ids.data_ptr(), article_ids.data_ptr(), edge3.data_ptr() if ne else None,
                                 label3.data_ptr(), labels.data_ptr(), user_ptr.data_ptr(), article_ptr.data_ptr(), 1,
                                 stream), "mi_sampler_emit3")
        csr = None
        blocks = [blk]
        if getattr(self, "emit_csr", False):
            from .. import ops
            s32 = (r4(nu + 1), r4(na + 1), r4(ne), r4(ne), r4(max(na, 1)))
            b32 = t.empty(sum(s32), dtype=t.int32, device=dev)
            o0, o1, o2, o3 = s32[0], s32[0] + s32[1], s32[0] + s32[1] + s32[2], s32[0] + s32[1] + s32[2] + s32[3]
            u_rowptr, a_rowptr = b32[0:nu + 1], b32[o0:o0 + na + 1]
            u_col, a_col, cursor = b32[o1:o1 + ne], b32[o2:o2 + ne], b32[o3:o3 + max(na, 1)]
            check(L.mi_sampler_emit_csr(ctypes.byref(desc), ws.data_ptr(), ws.numel(), tot, u_rowptr.data_ptr(),
                                        u_col.data_ptr() if ne else None, a_rowptr.data_ptr(), a_col.data_ptr() if ne else None,
                                        cursor.data_ptr(), stream), "mi_sampler_emit_csr")
            # (rows = customers, rows = articles): by-source and by-destination forms of the customer -> article relation
            csr = (ops.DeviceCSR(nu, na, u_rowptr, u_col), ops.DeviceCSR(na, nu, a_rowptr, a_col))
            blocks.append(b32)
        if raw:
            out = {"user_ids": user_ids, "article_ids": article_ids, "edge_index": edge_index,
                    "edge_label_index": label_index, "edge_label": labels, "user_ptr": user_ptr, "article_ptr": article_ptr}
            if csr is not None:
                out["csr_by_customer"], out["csr_by_article"] = csr
            return out
        data = HeteroData()
        xc, xa = self.user_x[user_ids], self.article_x[article_ids]
        data[Constants.node_user].x = xc
        data[Constants.node_item].x = xa
        data[Constants.node_user].n_id = user_ids
        data[Constants.node_item].n_id = article_ids
        if csr is not None:
            edge_index._sorted_csr = csr  # (by source, by destination), read by model/layers.py BipartiteGraph.of
        data[Constants.edge_key].edge_index = edge_index
        data[Constants.edge_key].edge_label_index = label_index
        data[Constants.edge_key].edge_label = labels
        rev = edge3[1:3]                  # = edge_index.flip(0): rows (article, customer)
        rev._reverse_of = edge_index      # lets the encoder reuse the forward relation's sorted CSRs (model/layers.py)
        data[Constants.rev_edge_key].edge_index = rev
        data[Constants.rev_edge_key].edge_label_index = label3[1:3]
        data[Constants.rev_edge_key].edge_label = labels
        data._blocks = blocks + [xc, xa]  # the storages behind every tensor of the batch (the iterator's record_stream calls)
        data._seed_users, data._user_ptr = seeds, user_ptr   # per-sample structure (run_submission.make_predictions' sync-free path)
        data._max_candidates = int(self.max_neg)             # capacity of a sample's label-0 list
        return data

    def iter_users(self, users: Tensor) -> Iterator[HeteroData]:
        """The batches of `users` (int64, in this order, batch_size at a time) through the same pipelined iterator as an epoch:
        batch i is sampled with Philox step self.step + i, exactly what sample(users[i * B:(i + 1) * B], step=self.step + i)
        returns — evaluation of a subset of the customers (a submission shard, held-out users) without a host wait per batch."""
        self._order_override = users.detach().to("cpu", t.int64).contiguous()
        return iter(self)

    def __iter__(self) -> Iterator[HeteroData]:
        """One epoch: every user once, shuffled (DataLoader(shuffle=True) semantics).  With `prefetch` (default)
        sampling runs three batches ahead on a side stream while the consumer trains on batch i: nothing of the
        sampler sits between two steps.  Same batches, same order, same Philox steps as the serial loop."""
        g = t.Generator(device="cpu").manual_seed(self.seed + self.step)
        order = getattr(self, "_order_override", None)
        self._order_override = None
        if order is None:
            order = t.randperm(self.num_users, generator=g) if self.shuffle else t.arange(self.num_users)
        n_order = int(order.numel())
        batches = [order[b:b + self.batch_size] for b in range(0, n_order, self.batch_size)]
        if not getattr(self, "prefetch", True) or not batches:
            for seeds in batches:
                yield self.sample(seeds)
            return
        L = _lib.lib()
        main = t.cuda.current_stream(self.device)
        DEPTH = 3   # batches in flight: the walk of batch i + 3 is enqueued while the consumer trains on batch i (see below)
        if getattr(self, "_side", None) is None:
            self._side = t.cuda.Stream(device=self.device)
            self._pinned = [t.empty(4, dtype=t.int32).pin_memory() for _ in range(DEPTH)]
            # a workspace per batch in flight: phase A of batch j + DEPTH may be enqueued before phase B of batch j + 1 ...
            self._ws_ring = [self._ws] + [t.empty_like(self._ws) for _ in range(DEPTH - 1)]
        side = self._side
        step0 = self.step

        # the epoch's seed order goes to the device ONCE (round 4: a 24-element host-to-device copy per batch was 40 us of the
        # loop's host time — the loop is host-bound, tools/prof_host_native.py); a batch's seeds are a view of it
        order_dev = order.to(self.device, t.int64)
        batches_dev = [order_dev[b:b + self.batch_size] for b in range(0, n_order, self.batch_size)]

        side_raw = side.cuda_stream

        def start(i: int):  # phase A of batch i, enqueued on the side stream (raw handle: no torch op, no stream context), no host wait
            seeds = batches_dev[i]
            desc = self._desc if seeds.numel() == self.batch_size else self._make_desc(seeds.numel())
            check(L.mi_sampler_count_async(ctypes.byref(desc), seeds.data_ptr(), self.seed & (2**64 - 1),
                                           int(step0 + i) & (2**64 - 1), self._ws_ring[i % DEPTH].data_ptr(), self._ws_ring[i % DEPTH].numel(),
                                           self._pinned[i % DEPTH].data_ptr(), side_raw), "mi_sampler_count_async")
            ev = t.cuda.Event()
            ev.record(side)
            return seeds, desc, ev

        def finish(i: int, pend):  # phase B of batch i on the side stream, once its four totals have landed
            seeds, desc, ev = pend
            ev.synchronize()
            totals = self._pinned[i % DEPTH].tolist()
            with t.cuda.stream(side):
                data = self._emit(seeds, desc, totals, side_raw, ws=self._ws_ring[i % DEPTH])
                ready = t.cuda.Event()
                ready.record(side)
            return data, ready

        def hand_out(data, ready):
            main.wait_event(ready)
            for blk in data._blocks:      # every tensor of the batch is a view of one of these storages
                blk.record_stream(main)
            return data

        # THREE batches ahead (round 4; two before): while the consumer trains on batch i, batch i+1 is being emitted and the
        # walks of batches i+2 and i+3 are queued behind it.  With two, finish(i+2) waited for a walk that had been enqueued
        # only one host-side step issue (~0.2 ms) earlier — the walk + emit chain of a batch is ~0.3 ms of side-stream kernels
        # plus ~0.15 ms of host work, and the loop ran at THAT period (0.47 ms) although the step's kernels take 0.41 and the
        # host issues an iteration in ~0.3 (tools/prof_host_native.py).  Now the wait is on work enqueued a whole iteration
        # earlier.  Side-stream order: ... emit(i+1), count(i+3), emit(i+2), count(i+4) ...; batch j works in workspace j mod 3,
        # so count(j+3) overwrites workspace j mod 3 only after emit(j) — same stream, enqueued earlier.
        side.wait_stream(main)
        nb = len(batches)
        if getattr(self, "prefetch", True) == "thread":
            # the sampler's host side (a dozen launches and allocations per batch) on a thread of its own: the training
            # step is bound by its own launches, so this takes the sampler's share off the loop's critical path
            import queue
            import threading
            out: "queue.Queue" = queue.Queue(maxsize=2)
            dev_index = self.device.index if self.device.index is not None else t.cuda.current_device()
            stop = threading.Event()

            def produce():
                try:
                    t.cuda.set_device(dev_index)
                    pend = start(0)
                    for i in range(nb):
                        item = finish(i, pend)
                        pend = start(i + 1) if i + 1 < nb else None
                        while not stop.is_set():
                            try:
                                out.put(item, timeout=0.05)
                                break
                            except queue.Full:
                                continue
                        if stop.is_set():
                            return
                except BaseException as exc:  # surfaces in the consumer
                    out.put(exc)

            import sys
            interval = sys.getswitchinterval()
            sys.setswitchinterval(min(interval, float(os.environ.get("LAPLACE_SAMPLER_SWITCH", "1e-4"))))  # neither thread may sit on the interpreter lock for 5 ms
            worker = threading.Thread(target=produce, name="laplace-sampler", daemon=True)
            worker.start()
            try:
                for i in range(nb):
                    item = out.get()
                    if isinstance(item, BaseException):
                        raise item
                    self.step = step0 + i + 1
                    yield hand_out(*item)
            finally:
                stop.set()
                worker.join()
                sys.setswitchinterval(interval)
                t.cuda.current_stream(self.device).wait_stream(side)
            return
        pend = {j: start(j) for j in range(min(DEPTH, nb))}      # phase A of the first DEPTH batches
        cur = finish(0, pend.pop(0))
        for i in range(nb):
            data, ready = cur
            if i + 1 < nb:
                cur = finish(i + 1, pend.pop(i + 1))
                if i + DEPTH < nb:
                    pend[i + DEPTH] = start(i + DEPTH)
            self.step = step0 + i + 1
            yield hand_out(data, ready)

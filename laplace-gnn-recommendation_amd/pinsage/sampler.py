"""PinSAGE sampling on device — the roles of ItemToItemBatchSampler, NeighborSampler and the block
construction of the reference's pinsage/sampler.py:16-106, over csrc/pinsage.hip.

A block (DGL "message flow graph") is a dict: src_ids (global item ids, destination nodes first), n_dst,
edge_src / edge_dst (block-local ids), weights (random-walk visit counts).  Random walks and the
top-T selection run in the HIP kernels; relabelling the few hundred nodes of a batch is torch index
plumbing on the GPU."""
from __future__ import annotations

import ctypes
from typing import List, Optional, Tuple

import numpy as np
import torch as t
from torch import Tensor

from .. import _lib
from .._lib import check
from ..data.dataset import AdjList


class PinSAGESampler:
    def __init__(self, users_adj_list, articles_adj_list, num_users: int, num_items: int, *, batch_size: int = 32,
                 random_walk_length: int = 2, random_walk_restart_prob: float = 0.5, num_random_walks: int = 10,
                 num_neighbors: int = 3, num_layers: int = 2, device: str = "cuda", seed: int = 0):
        self.device = t.device(device)
        users, items = AdjList(users_adj_list, num_users), AdjList(articles_adj_list, num_items)
        to32 = lambda a: t.from_numpy(np.ascontiguousarray(a.astype(np.int32))).to(self.device)
        self.ui_ptr, self.ui_idx = to32(users.ptr), to32(users.idx)      # user -> items
        self.iu_ptr, self.iu_idx = to32(items.ptr), to32(items.idx)      # item -> users
        self.num_users, self.num_items = num_users, num_items
        self.batch_size, self.L, self.p = int(batch_size), int(random_walk_length), float(random_walk_restart_prob)
        self.W, self.T, self.n_layers, self.seed = int(num_random_walks), int(num_neighbors), int(num_layers), int(seed)
        self._pos = t.full((num_items,), -1, dtype=t.int64, device=self.device)  # scratch for relabelling
        self._pos32 = t.full((num_items,), -1, dtype=t.int32, device=self.device)  # the device path's scratch (kept all -1)
        self.step = 0
        self.device_batches = True   # whole-batch construction on the device (mi_pinsage_sample_batch) where it applies

    def _stream(self):
        return _lib.current_stream()

    def item_pairs(self, step: int) -> Tuple[Tensor, Tensor, Tensor]:
        """(heads, tails, neg_tails), pairs whose walk died removed (pinsage/sampler.py:26-41)."""
        B, dev = self.batch_size, self.device
        heads, tails, negs = (t.empty(B, dtype=t.int64, device=dev) for _ in range(3))
        check(_lib.lib().mi_pinsage_item_pairs(B, self.num_items, self.iu_ptr.data_ptr(), self.iu_idx.data_ptr(),
                                               self.ui_ptr.data_ptr(), self.ui_idx.data_ptr(), self.seed & (2**64 - 1),
                                               step, heads.data_ptr(), tails.data_ptr(), negs.data_ptr(), self._stream()),
              "mi_pinsage_item_pairs")
        keep = tails != -1
        return heads[keep], tails[keep], negs[keep]

    def neighbors(self, seeds: Tensor, layer: int, step: int) -> Tuple[Tensor, Tensor]:
        n, dev = seeds.numel(), self.device
        nb = t.empty(n, self.T, dtype=t.int64, device=dev)
        wt = t.empty(n, self.T, dtype=t.int64, device=dev)
        L = _lib.lib()
        ws = t.empty(int(L.mi_pinsage_neighbors_workspace_bytes(n, self.L, self.W)), dtype=t.uint8, device=dev)
        check(L.mi_pinsage_neighbors(n, seeds.data_ptr(), self.iu_ptr.data_ptr(), self.iu_idx.data_ptr(),
                                     self.ui_ptr.data_ptr(), self.ui_idx.data_ptr(), self.L, self.p, self.W, self.T, layer,
                                     self.seed & (2**64 - 1), step, nb.data_ptr(), wt.data_ptr(), ws.data_ptr(), ws.numel(),
                                     self._stream()), "mi_pinsage_neighbors")
        return nb, wt

    def sample_blocks(self, seeds: Tensor, step: int, heads: Optional[Tensor] = None, tails: Optional[Tensor] = None,
                      neg_tails: Optional[Tensor] = None) -> List[dict]:
        """NeighborSampler.sample_blocks (pinsage/sampler.py:73-91); input layer first."""
        blocks: List[dict] = []
        seeds = seeds.to(self.device, t.int64).contiguous()
        banned = None
        if heads is not None:
            banned = t.unique(t.cat([heads * self.num_items + tails, heads * self.num_items + neg_tails]))
        for layer in range(self.n_layers):
            nb, wt = self.neighbors(seeds, layer, step)
            n = seeds.numel()
            dst = t.arange(n, device=self.device)[:, None].expand(n, self.T)
            keep = nb >= 0
            if banned is not None:  # frontier edge v -> s is removed when (v, s) is a label pair (head -> tail)
                keep &= ~t.isin(nb * self.num_items + seeds[:, None], banned)
            es_g, ed, ew = nb[keep], dst[keep], wt[keep].to(t.float32)
            self._pos[seeds] = t.arange(n, device=self.device)
            new = t.unique(es_g)
            new = new[self._pos[new] < 0]                       # sources that are not destination nodes, ascending
            src_ids = t.cat([seeds, new])
            self._pos[new] = t.arange(n, n + new.numel(), device=self.device)
            es = self._pos[es_g]
            self._pos[src_ids] = -1                              # scratch back to all -1
            blocks.insert(0, {"src_ids": src_ids, "n_dst": n, "edge_src": es, "edge_dst": ed, "weights": ew})
            seeds = src_ids
        return blocks

    def _batch_buffers(self) -> dict:
        """Output buffers of one device-built batch at their upper bounds + the descriptor structs that point into them."""
        dev = self.device
        B, T, NL = self.batch_size, self.T, self.n_layers
        out = _lib.PinsageBatchOut()
        i64 = lambda n: t.empty(n, dtype=t.int64, device=dev)
        i32 = lambda n: t.empty(n, dtype=t.int32, device=dev)
        f32 = lambda n: t.empty(n, dtype=t.float32, device=dev)
        seeds, pos_u, pos_v, neg_v, counts = i64(3 * B), i64(B), i64(B), i64(B), i32(2 + 2 * NL)
        out.seeds, out.pos_u, out.pos_v, out.neg_v, out.counts = (x.data_ptr() for x in (seeds, pos_u, pos_v, neg_v, counts))
        bufs, n_max = [], 3 * B
        for l in range(NL):
            e_max, s_max = n_max * T, n_max * (1 + T)
            b = dict(src_ids=i64(s_max), edge_src=i64(e_max), edge_dst=i64(e_max), weights=f32(e_max), dst_rowptr=i32(n_max + 1),
                     dst_col=i32(e_max), dst_val=f32(e_max), src_rowptr=i32(s_max + 1), src_col=i32(e_max), src_val=f32(e_max))
            for k, v in b.items():
                setattr(out.blocks[l], k, v.data_ptr())
            bufs.append(b)
            n_max = s_max
        ws = t.empty(int(_lib.lib().mi_pinsage_batch_workspace_bytes(B, self.L, self.W, T, NL)), dtype=t.uint8, device=dev)
        host = t.empty(2 + 2 * NL, dtype=t.int32).pin_memory() if dev.type == "cuda" else t.empty(2 + 2 * NL, dtype=t.int32)
        return dict(out=out, seeds=seeds, pos_u=pos_u, pos_v=pos_v, neg_v=neg_v, counts=counts, bufs=bufs, ws=ws, host=host)

    def _launch_batch(self, step: int, buf: Optional[dict] = None, pos32: Optional[Tensor] = None) -> Optional[dict]:
        """Enqueue the whole batch (one C call, six launches for two layers) and the copy of its counts to pinned host
        memory on the current stream; None when the sizes are outside the single-workgroup kernels'."""
        if self.n_layers > _lib.MI_PINSAGE_MAX_LAYERS:
            return None
        buf = buf if buf is not None else self._batch_buffers()
        desc = _lib.PinsageBatchDesc(self.batch_size, self.num_items, self.iu_ptr.data_ptr(), self.iu_idx.data_ptr(),
                                     self.ui_ptr.data_ptr(), self.ui_idx.data_ptr(), self.L, self.W, self.T, self.n_layers, self.p,
                                     (self._pos32 if pos32 is None else pos32).data_ptr())
        rc = _lib.lib().mi_pinsage_sample_batch(ctypes.byref(desc), self.seed & (2**64 - 1), step, ctypes.byref(buf["out"]),
                                                buf["ws"].data_ptr(), buf["ws"].numel(), self._stream())
        if rc == _lib.MI_ERR_UNSUPPORTED:
            return None
        check(rc, "mi_pinsage_sample_batch")
        buf["host"].copy_(buf["counts"], non_blocking=True)
        return buf

    def _finish_batch(self, buf: dict) -> dict:
        """Views of the batch at its actual sizes; the caller has made sure the count copy has landed."""
        from ..ops import DeviceCSR
        c = buf["host"].tolist()
        n_pairs, n_seeds = c[0], c[1]
        blocks, n_dst = [], n_seeds
        for l, b in enumerate(buf["bufs"]):
            n_src, n_e = c[2 + 2 * l], c[3 + 2 * l]
            by_dst = DeviceCSR(n_dst, n_src, b["dst_rowptr"][: n_dst + 1], b["dst_col"][:n_e], b["dst_val"][:n_e])
            by_src = DeviceCSR(n_src, n_dst, b["src_rowptr"][: n_src + 1], b["src_col"][:n_e], b["src_val"][:n_e])
            blocks.insert(0, {"src_ids": b["src_ids"][:n_src], "n_dst": n_dst, "edge_src": b["edge_src"][:n_e],
                              "edge_dst": b["edge_dst"][:n_e], "weights": b["weights"][:n_e], "csr": (by_dst, by_src)})
            n_dst = n_src
        pu = buf["pos_u"][:n_pairs]
        return {"seeds": buf["seeds"][:n_seeds], "pos": (pu, buf["pos_v"][:n_pairs]), "neg": (pu, buf["neg_v"][:n_pairs]),
                "blocks": blocks}

    def _sample_batch_device(self, step: int) -> Optional[dict]:
        """The whole batch in one C call and ONE host read-back (the counts); None when the sizes are outside the
        single-workgroup kernels' (the caller then takes the index-op path)."""
        buf = self._launch_batch(step)
        if buf is None:
            return None
        if self.device.type == "cuda":
            t.cuda.current_stream(self.device).synchronize()
        return self._finish_batch(buf)

    def batches(self, n: int):
        """n training batches (steps self.step ...), sampled AHEAD of the caller on two side streams: while the caller trains
        on batch i, the chains of batches i + 1 and i + 2 are in flight, one per stream (round 4; one stream and one batch
        ahead in round 3).  A batch is ~six dependent single-workgroup launches — a LATENCY chain of 0.25-0.33 ms that one
        side stream runs back to back, so the loop ran at the sampler's period (0.318 ms at walk length 3 against ~0.2 ms of
        step); two chains side by side do not compete for the chip (one workgroup each) and halve that period.  Each stream has
        its own position scratch; four rotating buffer sets: a batch's tensors stay valid until the caller has asked for the
        batch after the next.  Same batches as n calls of sample_batch()."""
        if not (self.device_batches and self.device.type == "cuda") or n <= 0:
            for _ in range(max(n, 0)):
                yield self.sample_batch()
            return
        main = t.cuda.current_stream(self.device)
        LANES, SETS = 2, 4
        if getattr(self, "_lanes", None) is None:
            self._lanes = [t.cuda.Stream(device=self.device) for _ in range(LANES)]
            self._lane_pos = [t.full_like(self._pos32, -1) for _ in range(LANES)]   # not the serial path's: an abandoned epoch's chains may still be running
            self._sets = [None] * SETS
        first = self.step

        def launch(i: int):
            k, lane = i % SETS, self._lanes[i % LANES]
            lane.wait_stream(main)       # the set's previous batch (four back) has been consumed by then
            with t.cuda.stream(lane):
                if self._sets[k] is None:
                    self._sets[k] = self._batch_buffers()
                buf = self._launch_batch(first + i, self._sets[k], pos32=self._lane_pos[i % LANES])
                ev = t.cuda.Event()
                ev.record(lane)
            return buf, ev

        pending = [launch(j) for j in range(min(LANES, n))]
        for i in range(n):
            buf, ev = pending.pop(0)
            if buf is None:              # sizes outside the device path: everything through sample_batch
                self.step = first + i
                for lane in self._lanes:
                    lane.synchronize()
                for _ in range(n - i):
                    yield self.sample_batch()
                return
            if i + LANES < n:
                pending.append(launch(i + LANES))
            ev.synchronize()             # host: the counts are in pinned memory
            main.wait_event(ev)          # device: the batch's tensors are complete
            self.step = first + i + 1
            yield self._finish_batch(buf)

    def sample_batch(self, step: Optional[int] = None) -> dict:
        """One training batch: pair graphs compacted to `seeds`, blocks rooted at them (sampler.py:93-106)."""
        if step is None:
            step = self.step
            self.step += 1
        if self.device_batches:
            batch = self._sample_batch_device(step)
            if batch is not None:
                return batch
        heads, tails, negs = self.item_pairs(step)
        seeds = t.unique(t.cat([heads, tails, negs]))
        loc = lambda x: t.searchsorted(seeds, x)
        blocks = self.sample_blocks(seeds, step, heads, tails, negs)
        pu = loc(heads)
        return {"seeds": seeds, "pos": (pu, loc(tails)), "neg": (pu, loc(negs)), "blocks": blocks}

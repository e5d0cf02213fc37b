"""Minimal heterogeneous-graph container with the attribute surface the reference uses from
torch_geometric.data.HeteroData (data/dataset.py:164-182, utils/get_info.py:39-48, training.py:72-75)
and the collate the PyG DataLoader applies to it (data/data_loader.py:48-50):
node features concatenated per type, edge_index / edge_label_index offset by the cumulative node
counts of their source / destination types, edge_label concatenated.
"""
from __future__ import annotations

from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import torch as t
from torch import Tensor

EdgeType = Tuple[str, str, str]


class Store(dict):
    """Attribute-style storage of one node or edge type."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e

    def __setattr__(self, name, value):
        self[name] = value


class HeteroData:
    def __init__(self):
        self._nodes: Dict[str, Store] = {}
        self._edges: Dict[EdgeType, Store] = {}

    def __getitem__(self, key):
        if isinstance(key, str):
            return self._nodes.setdefault(key, Store())
        key = tuple(key)
        return self._edges.setdefault(key, Store())

    @property
    def node_types(self) -> List[str]:
        return list(self._nodes)

    @property
    def edge_types(self) -> List[EdgeType]:
        return list(self._edges)

    def metadata(self) -> Tuple[List[str], List[EdgeType]]:
        return self.node_types, self.edge_types

    @property
    def x_dict(self) -> Dict[str, Tensor]:
        return {k: s["x"] for k, s in self._nodes.items() if "x" in s}

    @property
    def edge_index_dict(self) -> Dict[EdgeType, Tensor]:
        return {k: s["edge_index"] for k, s in self._edges.items() if "edge_index" in s}

    def to(self, device) -> "HeteroData":
        out = HeteroData()
        for k, s in self._nodes.items():
            out._nodes[k] = Store({a: (v.to(device) if isinstance(v, Tensor) else v) for a, v in s.items()})
        for k, s in self._edges.items():
            out._edges[k] = Store({a: (v.to(device) if isinstance(v, Tensor) else v) for a, v in s.items()})
        return out

    def num_nodes(self, node_type: str) -> int:
        return int(self._nodes[node_type]["x"].shape[0])


def collate(samples: Sequence[HeteroData]) -> HeteroData:
    """Batch.from_data_list for HeteroData: one disjoint-union graph."""
    out = HeteroData()
    first = samples[0]
    offsets: Dict[str, List[int]] = {}
    for nt in first.node_types:
        counts = [s.num_nodes(nt) for s in samples]
        offsets[nt] = [0]
        for c in counts[:-1]:
            offsets[nt].append(offsets[nt][-1] + c)
        out[nt].x = t.cat([s[nt].x for s in samples], dim=0)
        for attr in first[nt]:  # per-node extras, e.g. n_id (global node ids, as PyG's sampled batches carry)
            if attr != "x" and isinstance(first[nt][attr], Tensor):
                out[nt][attr] = t.cat([s[nt][attr] for s in samples], dim=0)
        out[nt].batch = t.cat([t.full((c,), i, dtype=t.int64) for i, c in enumerate(counts)])
    for et in first.edge_types:
        src_t, _, dst_t = et
        for attr in first[et]:
            vals = [s[et][attr] for s in samples]
            if attr.endswith("index"):
                shifted = []
                for i, v in enumerate(vals):
                    off = t.tensor([[offsets[src_t][i]], [offsets[dst_t][i]]], dtype=v.dtype, device=v.device)
                    shifted.append(v + off)
                out[et][attr] = t.cat(shifted, dim=1)
            else:
                out[et][attr] = t.cat(vals, dim=0)
    return out


class DataLoader:
    """Mini-batches of `batch_size` dataset items, collated (torch_geometric.loader.DataLoader stand-in)."""

    def __init__(self, dataset, batch_size: int = 1, shuffle: bool = False, generator: Optional[t.Generator] = None):
        self.dataset, self.batch_size, self.shuffle, self.generator = dataset, int(batch_size), shuffle, generator

    def __len__(self) -> int:
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def __iter__(self) -> Iterator[HeteroData]:
        n = len(self.dataset)
        order = t.randperm(n, generator=self.generator).tolist() if self.shuffle else list(range(n))
        for b in range(0, n, self.batch_size):
            yield collate([self.dataset[i] for i in order[b:b + self.batch_size]])

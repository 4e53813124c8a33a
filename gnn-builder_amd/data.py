"""Minimal graph / dataset containers with the attribute surface ``Project`` reads from a PyG
dataset (reference code_gen.py:252-285): ``dataset.indices()``, ``dataset[idx]``,
``dataset.num_classes`` and per graph ``x``, ``edge_index`` ([2, E]), ``y``, ``num_nodes``,
``num_edges``.  A real ``torch_geometric`` dataset works unchanged where PyG is installed."""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch

from .batching import GraphBatch


class GraphData:
    def __init__(self, x, edge_index, y=None):
        self.x = torch.as_tensor(x, dtype=torch.float32)
        self.edge_index = torch.as_tensor(edge_index, dtype=torch.long).reshape(2, -1)
        self.y = torch.zeros(1) if y is None else torch.as_tensor(y)

    @property
    def num_nodes(self) -> int:
        return int(self.x.shape[0])

    @property
    def num_edges(self) -> int:
        return int(self.edge_index.shape[1])


class ListDataset:
    def __init__(self, graphs: Sequence[GraphData], num_classes: int = 1):
        self._graphs: List[GraphData] = list(graphs)
        self.num_classes = num_classes

    def indices(self):
        return range(len(self._graphs))

    def __len__(self):
        return len(self._graphs)

    def __getitem__(self, idx):
        return self._graphs[idx]

    def __iter__(self):
        return iter(self._graphs)

    @classmethod
    def from_batch(cls, batch: GraphBatch, y_dim: int = 1, seed: int = 0) -> "ListDataset":
        rng = np.random.default_rng(seed)
        graphs = []
        for g in range(batch.num_graphs):
            x, coo = batch.graph(g)
            graphs.append(GraphData(x, coo.T.astype(np.int64), rng.uniform(-1, 1, y_dim).astype(np.float32)))
        return cls(graphs, num_classes=y_dim)

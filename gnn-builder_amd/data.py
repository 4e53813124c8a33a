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


def load_tb_data(tb_dir, num_features: int, out_dim: int | None = None):
    """Read a ``tb_data/`` directory in the reference's on-disk format (reference code_gen.py:227-305;
    what ``Project.gen_testbench_data`` writes and the generated ``model_tb.cpp`` reads,
    model_tb.cpp.jinja:100-131): ``dataset_info.txt`` + per graph ``graph_<idx>_{info,coo,node_features,
    model_golden_output}.bin`` (raw little-endian int32 / fp32, no headers).  Returns
    ``(GraphBatch, golden [B, out_dim] or None, indices)`` -- the graphs packed in file order, ready for
    ``gnnb_forward_batched``."""
    from pathlib import Path

    from .batching import pack_graphs

    tb = Path(tb_dir)
    lines = (tb / "dataset_info.txt").read_text().split()
    if len(lines) < 2 or lines[0] != "num_graphs":
        raise ValueError("dataset_info.txt: expected 'num_graphs <n>' followed by one index per line")
    n = int(lines[1])
    indices = [int(v) for v in lines[2:2 + n]]
    if len(indices) != n:
        raise ValueError("dataset_info.txt lists fewer graph indices than num_graphs")
    graphs, golden = [], []
    for idx in indices:
        base = tb / "graphs" / f"graph_{idx}"
        info = np.fromfile(f"{base}_info.bin", dtype="<i4")
        if info.size != 2:
            raise ValueError(f"{base}_info.bin must hold (num_nodes, num_edges)")
        nn, ne = int(info[0]), int(info[1])
        coo = np.fromfile(f"{base}_coo.bin", dtype="<i4")
        x = np.fromfile(f"{base}_node_features.bin", dtype="<f4")
        if coo.size != 2 * ne or x.size != nn * num_features:
            raise ValueError(f"graph_{idx}: file sizes disagree with its info record")
        graphs.append((x.reshape(nn, num_features), coo.reshape(ne, 2)))
        gp = Path(f"{base}_model_golden_output.bin")
        if gp.exists():
            golden.append(np.fromfile(gp, dtype="<f4"))
    out = None
    if golden and len(golden) == n:
        out = np.stack(golden)
        if out_dim is not None and out.shape[1] != out_dim:
            raise ValueError("golden outputs do not have the expected width")
    return pack_graphs(graphs), out, indices

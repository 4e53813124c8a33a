#!/usr/bin/env python3
"""Development: the row-class GEMM of a PNA layer on a batch whose nodes all have in-degree 2 (rings): one class, the stable
sort leaves the rows in place -- what does the row-class mode cost without the scattered rows?  (run under rocprofv3)"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, torch
from helpers import make_model, to_dev
from gnnbuilder_amd import runtime
from gnnbuilder_amd.batching import pack_graphs
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
n, B = 18, 8192
ring = np.array([[i, (i + 1) % n] for i in range(n)] + [[(i + 1) % n, i] for i in range(n)], np.int32)
graphs = [(rng.uniform(-1, 1, (n, 11)).astype(np.float32), ring) for _ in range(B)]
batch = pack_graphs(graphs)
model = make_model("pna", in_dim=11, hidden=128, layers=3, act="relu", pools=("add", "mean", "max"), task_out=19)
cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges)
cm.set_max_degree(2)
args = to_dev(batch, dev)
for _ in range(12):
    cm.forward(*args)
torch.cuda.synchronize()
pass
print("ok", batch.num_nodes)

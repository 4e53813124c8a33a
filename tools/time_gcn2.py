#!/usr/bin/env python3
"""Launch-loop time of the fused GCN stack kernel on the BASELINE config 2 batch (HIP events), both math modes."""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch, numpy as np
import bench
from gnnbuilder_amd import runtime, synthetic
w = bench.WORKLOADS["c2"]; dev = torch.device("cuda:0")
model = bench.build_model(w)
b = synthetic.make_batch(w["shape"], w["batch"], seed=0)
cm = runtime.CompiledModel.from_model(model, b.num_graphs, b.num_nodes, b.num_edges, max_graph_nodes=int(np.diff(b.node_ptr).max()))
bd = tuple(torch.from_numpy(a).to(dev) for a in (b.x, b.coo, b.node_ptr, b.edge_ptr))
cm.graph_prep(bd[1], bd[2], bd[3], int(bd[0].shape[0]))
for math in (0, 1):
    runtime.set_option("math", math)
    t = [cm.gcn_stack_timed(bd[0], 200) for _ in range(3)]
    print(f"math {math}: " + " ".join(f"{v:.2f}" for v in t) + " us")
runtime.set_option("math", 0)

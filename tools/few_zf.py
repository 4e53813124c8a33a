#!/usr/bin/env python3
"""A FEW launches of the C2 conv-stack kernel (for rocprofv3 --pmc passes, where every dispatch is slow)."""
import sys
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
print(cm.gcn_stack_timed(bd[0], int(sys.argv[1]) if len(sys.argv) > 1 else 8))

#!/usr/bin/env python3
"""Fused GCN stack kernel time vs batch size (QM9-shaped graphs): how much of the time at the BASELINE
batch (4096 graphs = ~4 stages per resident workgroup) is prologue / tail quantisation."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np, torch
import bench
from gnnbuilder_amd import runtime, synthetic
w = dict(bench.WORKLOADS["c2"]); dev = torch.device("cuda:0")
model = bench.build_model(w)
for B in (1024, 2048, 4096, 8192, 16384, 32768):
    b = synthetic.make_batch(w["shape"], B, seed=0)
    cm = runtime.CompiledModel.from_model(model, b.num_graphs, b.num_nodes, b.num_edges, max_graph_nodes=int(np.diff(b.node_ptr).max()))
    bd = tuple(torch.from_numpy(a).to(dev) for a in (b.x, b.coo, b.node_ptr, b.edge_ptr))
    cm.graph_prep(bd[1], bd[2], bd[3], int(bd[0].shape[0]))
    cm.gcn_stack_timed(bd[0], 100)
    us = min(cm.gcn_stack_timed(bd[0], 100) for _ in range(2))
    flops = 2.0 * b.num_nodes * (11 * 128 + 128 * 128)
    print(f"batch {B:6d}: {us:8.2f} us  {us / B * 1e3:6.2f} ns/graph  {flops / us / 1e6:6.1f} TFLOP/s ({flops / us / 1e6 / 157.3 * 100:4.1f}% of fp32 MFMA peak)")

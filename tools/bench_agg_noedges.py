#!/usr/bin/env python3
"""Diagnostic: the aggregate kernels on the C2 batch with ALL EDGES REMOVED (pure row copy with
the kernel's own indexing/launch structure) -- separates gather cost from structural overhead."""
import json, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench
from gnnbuilder_amd import runtime, synthetic
from gnnbuilder_amd.batching import GraphBatch
w = bench.WORKLOADS["c2"]; dev = torch.device("cuda:0")
model = bench.build_model(w)
b = synthetic.make_batch(w["shape"], w["batch"], seed=0)
for edges in ("all", "none"):
    bb = b if edges == "all" else GraphBatch(b.x, np.zeros((0, 2), np.int32), b.node_ptr, np.zeros_like(b.edge_ptr))
    cm = runtime.CompiledModel.from_model(model, bb.num_graphs, bb.num_nodes, max(bb.num_edges, 1))
    bd = tuple(torch.from_numpy(a).to(dev) for a in (bb.x, bb.coo, bb.node_ptr, bb.edge_ptr))
    for opts in ({"agg_variant": 0}, {"agg_variant": 2, "agg_lds_kb": 52, "tile_rows": 8}, {"agg_variant": 3}):
        for k, v in opts.items():
            runtime.set_option(k, v)
        alg, res = bench.measure_aggregate_roofline(cm, bd, 128, dev, iters=200)
        print(edges, opts, "hbm", round(res["hbm"]["us"], 2), "l3", round(res["l3_resident"]["us"], 2))

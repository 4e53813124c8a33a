#!/usr/bin/env python3
"""How fast can the host ISSUE steps?  The C2 forward on a tiny batch (GPU time negligible) and on the BASELINE batch, with
the issue loop timed separately from the drain: if issue time ~ total time the pipeline is host-bound, not kernel-bound.
usage: host_rate.py [steps]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from gnnbuilder_amd import runtime, synthetic  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
runtime.load_library(require_gpu=True)
dev = torch.device("cuda", 0)
w = bench.WORKLOADS["c2"]
model = bench.build_model(w)
for nb in (32, w["batch"]):
    batches = [synthetic.make_batch(w["shape"], nb, seed=i) for i in range(8)]
    maxn = max(b.num_nodes for b in batches)
    maxe = max(b.num_edges for b in batches)
    mg = int(max(np.diff(b.node_ptr).max() for b in batches))
    for ns in (1, 3):
        cms = [runtime.CompiledModel.from_model(model, nb, maxn, maxe, max_graph_nodes=mg) for _ in range(ns)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
        db = [tuple(torch.from_numpy(a).to(dev) for a in (b.x, b.coo, b.node_ptr, b.edge_ptr)) for b in batches]
        outs = [torch.empty(b.num_graphs, cms[0].out_dim, device=dev) for b in batches]

        def step(i):
            k = i % len(db)
            cms[i % ns].forward(*db[k], out=outs[k], stream=streams[i % ns])
        for i in range(50):
            step(i)
        torch.cuda.synchronize()
        res = []
        for _ in range(3):
            t0 = time.perf_counter()
            for i in range(steps):
                step(i)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            res.append(((t1 - t0) / steps * 1e6, (t2 - t0) / steps * 1e6))
        print(f"batch {nb:5d} streams {ns}: issue / total us per step:", "  ".join(f"{a:.1f}/{b:.1f}" for a, b in res), flush=True)

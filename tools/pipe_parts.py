#!/usr/bin/env python3
"""Which part of the C2 step costs what in the three-stream pipeline?  Period of (a) the whole forward, (b) the forward on
prepared topology (conv stack + readout, no graph prep), (c) graph prep alone, all on 3 streams / 3 workspaces."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from gnnbuilder_amd import runtime, synthetic  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 3
runtime.load_library(require_gpu=True)
dev = torch.device("cuda", 0)
w = bench.WORKLOADS["c2"]
model = bench.build_model(w)
nb = w["batch"]
batches = [synthetic.make_batch(w["shape"], nb, seed=i) for i in range(ns)]
maxn = max(b.num_nodes for b in batches)
maxe = max(b.num_edges for b in batches)
mg = int(max(np.diff(b.node_ptr).max() for b in batches))
cms = [runtime.CompiledModel.from_model(model, nb, maxn, maxe, max_graph_nodes=mg) for _ in range(ns)]
streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
db = [tuple(torch.from_numpy(a).to(dev) for a in (b.x, b.coo, b.node_ptr, b.edge_ptr)) for b in batches]
outs = [torch.empty(b.num_graphs, cms[0].out_dim, device=dev) for b in batches]


def full(i):
    k = i % ns
    cms[k].forward(*db[k], out=outs[k], stream=streams[k])


def prepared(i):
    k = i % ns
    cms[k].forward_prepared(db[k][0], out=outs[k], stream=streams[k])


def prep(i):
    k = i % ns
    cms[k].graph_prep(db[k][1], db[k][2], db[k][3], int(db[k][0].shape[0]), stream=streams[k])


for name, fn in (("forward (prep + stack + readout)", full), ("prepared topology (stack + readout)", prepared),
                 ("graph prep alone", prep), ("forward again", full)):
    for i in range(60):
        fn(i)
    torch.cuda.synchronize()
    res = []
    for _ in range(4):
        t0 = time.perf_counter()
        for i in range(steps):
            fn(i)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / steps * 1e6)
    print(f"{name:40s} us per step:", " ".join(f"{r:.1f}" for r in res), flush=True)

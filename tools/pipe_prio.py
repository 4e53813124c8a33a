#!/usr/bin/env python3
"""Experiment: graph prep on NORMAL-priority streams, conv stack + readout on HIGH-priority streams (event between them), so that
at the boundary between two stack kernels the dispatcher places the next stack kernel's workgroups before the waiting
graph-prep workgroups.  Period per step against the plain one-stream-per-batch order."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from gnnbuilder_amd import runtime, synthetic  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
ns = 3
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
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "?")
hi = [torch.cuda.Stream(device=dev, priority=-1) for _ in range(ns)]
lo = [torch.cuda.Stream(device=dev, priority=0) for _ in range(ns)]
db = [tuple(torch.from_numpy(a).to(dev) for a in (b.x, b.coo, b.node_ptr, b.edge_ptr)) for b in batches]
outs = [torch.empty(b.num_graphs, cms[0].out_dim, device=dev) for b in batches]
evs = [[torch.cuda.Event() for _ in range(2)] for _ in range(ns)]


def plain(i):
    k = i % ns
    cms[k].forward(*db[k], out=outs[k], stream=lo[k])


def plain_hi(i):
    k = i % ns
    cms[k].forward(*db[k], out=outs[k], stream=hi[k])


def split(i):
    k = i % ns
    lo[k].wait_event(evs[k][1])              # the previous forward on this workspace has read its tables
    cms[k].graph_prep(db[k][1], db[k][2], db[k][3], int(db[k][0].shape[0]), stream=lo[k])
    evs[k][0].record(lo[k])
    hi[k].wait_event(evs[k][0])
    cms[k].forward_prepared(db[k][0], out=outs[k], stream=hi[k])
    evs[k][1].record(hi[k])


for name, fn in (("one stream per batch, normal priority", plain), ("one stream per batch, high priority", plain_hi),
                 ("prep normal / stack + readout high", split), ("one stream per batch, normal priority", plain)):
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
    print(f"{name:42s} us per step:", " ".join(f"{r:.1f}" for r in res), flush=True)

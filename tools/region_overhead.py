#!/usr/bin/env python3
"""What a timed region pays besides its steps (C2): host-side time from `t0` to the return of the closing synchronize for
an empty region, one graph prep, one whole forward, and K forwards on three streams (fit: T = a + K p)."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from gnnbuilder_amd import runtime, synthetic  # noqa: E402

runtime.load_library(require_gpu=True)
dev = torch.device("cuda", 0)
w = bench.WORKLOADS["c2"]
model = bench.build_model(w)
nb, ns = w["batch"], 3
batches = [synthetic.make_batch(w["shape"], nb, seed=i) for i in range(8)]
maxn = max(b.num_nodes for b in batches)
maxe = max(b.num_edges for b in batches)
mg = int(max(np.diff(b.node_ptr).max() for b in batches))
cms = [runtime.CompiledModel.from_model(model, nb, maxn, maxe, max_graph_nodes=mg) for _ in range(ns)]
streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
db = [tuple(torch.from_numpy(a).to(dev) for a in (b.x, b.coo, b.node_ptr, b.edge_ptr)) for b in batches]
outs = [torch.empty(b.num_graphs, cms[0].out_dim, device=dev) for b in batches]


def step(i):
    k = i % len(db)
    cms[i % ns].forward(*db[k], out=outs[k], stream=streams[i % ns])


def region(fn, reps=30):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6)
    return np.median(ts[5:]), min(ts)


for i in range(200):
    step(i)
torch.cuda.synchronize()
print("empty region                 median %.1f us (min %.1f)" % region(lambda: None))
print("one graph prep               median %.1f us (min %.1f)" % region(lambda: cms[0].graph_prep(db[0][1], db[0][2], db[0][3], int(db[0][0].shape[0]), stream=streams[0])))
print("one forward                  median %.1f us (min %.1f)" % region(lambda: step(0)))
print("one forward, prepared        median %.1f us (min %.1f)" % region(lambda: cms[0].forward_prepared(db[0][0], out=outs[0], stream=streams[0])))
for K in (2, 3, 5, 10, 20, 40):
    print("K = %2d forwards on 3 streams  median %.1f us (min %.1f)" % ((K,) + region(lambda: [step(i) for i in range(K)])))

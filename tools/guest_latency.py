#!/usr/bin/env python3
"""Does a SMALL kernel on another stream make progress while the conv-stack kernel owns the chip?  Times (HIP events on its
own stream) a tiny elementwise kernel and the library's graph prep, alone and beside back-to-back C2 forwards."""
import sys
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
nb = w["batch"]
b = synthetic.make_batch(w["shape"], nb, seed=0)
mg = int(np.diff(b.node_ptr).max())
cmA = runtime.CompiledModel.from_model(model, nb, b.num_nodes, b.num_edges, max_graph_nodes=mg)
cmB = runtime.CompiledModel.from_model(model, nb, b.num_nodes, b.num_edges, max_graph_nodes=mg)
sA, sB = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
d = tuple(torch.from_numpy(a).to(dev) for a in (b.x, b.coo, b.node_ptr, b.edge_ptr))
out = torch.empty(nb, cmA.out_dim, device=dev)
cmA.graph_prep(d[1], d[2], d[3], int(d[0].shape[0]))
torch.cuda.synchronize()
tiny = torch.zeros(1024, device=dev)


def guest_tiny():
    with torch.cuda.stream(sB):
        tiny.add_(1.0)


def guest_prep():
    cmB.graph_prep(d[1], d[2], d[3], int(d[0].shape[0]), stream=sB)


for gname, guest in (("tiny elementwise", guest_tiny), ("graph prep", guest_prep)):
    for busy in (False, True):
        lat = []
        for i in range(150):
            if busy:
                for _ in range(3):
                    cmA.forward_prepared(d[0], out=out, stream=sA)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(sB)
            guest()
            e1.record(sB)
            lat.append((e0, e1))
            if i % 10 == 9:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        v = np.array([a.elapsed_time(c) * 1e3 for a, c in lat[20:]])
        print(f"{gname:18s} {'beside the conv stack' if busy else 'alone':22s}: median {np.median(v):6.1f} us  p10 {np.percentile(v, 10):6.1f}  p90 {np.percentile(v, 90):6.1f}", flush=True)

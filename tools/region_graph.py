#!/usr/bin/env python3
"""The K-step timed region of bench.py, eager against ONE hipGraph replay of the same K steps on the same streams.
usage: region_graph.py [K] [streams] [workload]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from gnnbuilder_amd import runtime, synthetic  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
S = int(sys.argv[2]) if len(sys.argv) > 2 else 3
runtime.load_library(require_gpu=True)
dev = torch.device("cuda", 0)
w = bench.WORKLOADS[sys.argv[3] if len(sys.argv) > 3 else "c2"]
model = bench.build_model(w)
batches = [synthetic.make_batch(w["shape"], w["batch"], seed=i) for i in range(8)]
segs = [None] * 8
mg, md = bench.workload_promises(w, batches, segs)
pipe = bench.Pipeline(model, batches, segs, S, dev, mg, md)
for i in range(40):
    pipe.step(i)
torch.cuda.synchronize()
ref = [o.clone() for o in pipe.outs]


def region_eager():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        pipe.step(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6


g = torch.cuda.CUDAGraph()
cap = torch.cuda.Stream(device=dev)
torch.cuda.synchronize()
with torch.cuda.graph(g, stream=cap):
    ev = torch.cuda.Event()
    ev.record(cap)
    for st in pipe.streams:
        st.wait_event(ev)
    for i in range(K):
        pipe.step(i)
    for st in pipe.streams:
        e = torch.cuda.Event()
        e.record(st)
        cap.wait_event(e)
torch.cuda.synchronize()


def region_graph():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6


for o in pipe.outs:
    o.zero_()
g.replay()
torch.cuda.synchronize()
same = all(torch.equal(a, b) for a, b in zip(ref[:min(K, 8)], pipe.outs[:min(K, 8)]))
for name, f in (("eager", region_eager), ("graph", region_graph), ("eager", region_eager), ("graph", region_graph)):
    ts = sorted(f() for _ in range(15))
    print("%s K=%d S=%d: median %.1f us (%.2f us/step), min %.1f" % (name, K, S, ts[7], ts[7] / K, ts[0]), flush=True)
print("graph replay reproduces the eager outputs bit for bit:", same)

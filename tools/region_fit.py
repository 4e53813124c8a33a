#!/usr/bin/env python3
"""Fixed cost of a timed region: bench.py's K-step region (sync, K steps on 3 streams, sync) for K = 1 .. 200 -> a + b K.
Also the cost of an empty region (sync, sync) and of the two halves (issue, drain)."""
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
batches = [synthetic.make_batch(w["shape"], w["batch"], seed=i) for i in range(8)]
segs = [None] * 8
mg, md = bench.workload_promises(w, batches, segs)
pipe = bench.Pipeline(model, batches, segs, 3, dev, mg, md)
for i in range(40):
    pipe.step(i)
torch.cuda.synchronize()
ts = []
for _ in range(200):
    t0 = time.perf_counter()
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
print("empty torch.cuda.synchronize(): median %.1f us" % (np.median(ts) * 1e6))
for K in (1, 2, 3, 5, 10, 20, 40, 100, 200):
    tot, iss = [], []
    for _ in range(15):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            pipe.step(i)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        tot.append(t2 - t0)
        iss.append(t1 - t0)
    print("K=%3d: region %.1f us = %.2f us/step; issue %.1f us (%.1f/step); drain after issue %.1f us" % (
        K, np.median(tot) * 1e6, np.median(tot) * 1e6 / K, np.median(iss) * 1e6, np.median(iss) * 1e6 / K, (np.median(tot) - np.median(iss)) * 1e6), flush=True)

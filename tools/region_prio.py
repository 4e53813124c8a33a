#!/usr/bin/env python3
"""The K-step region of bench.py with HIP stream priorities on the pipeline's streams (does a fixed dispatch order between the
conv-stack kernels of consecutive batches shorten the region?).  usage: region_prio.py [K] [workload]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from gnnbuilder_amd import runtime, synthetic  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
runtime.load_library(require_gpu=True)
dev = torch.device("cuda", 0)
w = bench.WORKLOADS[sys.argv[2] if len(sys.argv) > 2 else "c2"]
model = bench.build_model(w)
batches = [synthetic.make_batch(w["shape"], w["batch"], seed=i) for i in range(8)]
segs = [None] * 8
mg, md = bench.workload_promises(w, batches, segs)
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "?")
for name, prios in (("equal", (0, 0, 0)), ("high-mid-low", (-1, 0, 1)), ("high-low-low", (-1, 0, 0)), ("equal", (0, 0, 0)), ("all high", (-1, -1, -1))):
    pipe = bench.Pipeline(model, batches, segs, 3, dev, mg, md)
    try:
        pipe.streams = [torch.cuda.Stream(device=dev, priority=p) for p in prios]
    except Exception as e:  # noqa: BLE001
        print(name, "not available:", e)
        continue
    for i in range(40):
        pipe.step(i)
    torch.cuda.synchronize()
    ts = []
    for r in range(21):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            pipe.step(i)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6)
    ts.sort()
    print("%-14s K=%d: median %.1f us (%.2f us/step), min %.1f" % (name, K, ts[10], ts[10] / K, ts[0]), flush=True)

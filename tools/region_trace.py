#!/usr/bin/env python3
"""One bench-like timed region (sync, K steps on S streams, sync), a few times, for a rocprofv3 --kernel-trace timeline.
usage: region_trace.py [K] [streams] [workload]"""
import sys
import time
from pathlib import Path

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
for r in range(6):
    torch.cuda.synchronize()
    time.sleep(0.002)  # a visible gap between regions in the trace
    t0 = time.perf_counter()
    for i in range(K):
        pipe.step(i)
    torch.cuda.synchronize()
    print("region %d: %.1f us" % (r, (time.perf_counter() - t0) * 1e6), flush=True)

#!/usr/bin/env python3
"""Does the host's wake-up from torch.cuda.synchronize() cost the K = 20 region anything?  The region closed by a spin on the
streams (stream.query()) in front of the synchronisation, against the plain synchronisation."""
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
for K in (20, 100):
    for spin in (0, 1, 0, 1):
        ts = []
        for _ in range(15):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(K):
                pipe.step(i)
            if spin:
                while not all(s.query() for s in pipe.streams):
                    pass
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        print("K=%d spin=%d: %.1f us/step (region %.1f us)" % (K, spin, np.median(ts) * 1e6 / K, np.median(ts) * 1e6), flush=True)

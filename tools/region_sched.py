#!/usr/bin/env python3
"""The K-step region of bench.py under another issue schedule: TWO streams, two alternating workspaces per stream, the graph
prep of step i + 2 enqueued IN FRONT of the forward of step i on the same stream -- so that at most ONE successor conv-stack
kernel is ever ready (kernels of consecutive batches then take the chip one after the other instead of splitting its CUs) and it
is ready early (its prep ran beside the kernel before).  Every step is still one graph prep + one forward of its batch.
usage: region_sched.py [K] [workload]"""
import sys
import time
from pathlib import Path

import numpy as np
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
pipe = bench.Pipeline(model, batches, segs, 3, dev, mg, md)


class Hoisted:
    def __init__(self, nstreams, depth):
        maxn, maxe, maxb = (max(getattr(b, a) for b in batches) for a in ("num_nodes", "num_edges", "num_graphs"))
        self.ns, self.depth = nstreams, depth
        self.cms = [[runtime.CompiledModel.from_model(model, maxb, maxn, maxe, max_graph_nodes=mg) for _ in range(depth)] for _ in range(nstreams)]
        if md:
            for row in self.cms:
                for c in row:
                    c.set_max_degree(md)
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
        self.dev_batches, self.outs = pipe.dev_batches, [torch.empty_like(o) for o in pipe.outs]

    def _prep(self, i):
        k, j = i % len(self.dev_batches), i % self.ns
        x, coo, nptr, eptr = self.dev_batches[k]
        self.cms[j][(i // self.ns) % self.depth].graph_prep(coo, nptr, eptr, int(x.shape[0]), stream=self.streams[j])

    def _fwd(self, i):
        k, j = i % len(self.dev_batches), i % self.ns
        self.cms[j][(i // self.ns) % self.depth].forward_prepared(self.dev_batches[k][0], out=self.outs[k], stream=self.streams[j])

    def run(self, k):
        ahead = self.ns * (self.depth - 1)
        for i in range(min(ahead, k)):
            self._prep(i)
        for i in range(k):
            if i + ahead < k:
                self._prep(i + ahead)
            self._fwd(i)


def timed(f, reps=21):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6)
    ts.sort()
    return ts


def eager():
    for i in range(K):
        pipe.step(i)


for i in range(40):
    pipe.step(i)
torch.cuda.synchronize()
ref = [o.clone() for o in pipe.outs]
for name, f, chk in [("3 streams, prep + forward per step (bench.py today)", eager, None)] + \
        [("%d streams x %d workspaces, prep hoisted" % (ns, dp), Hoisted(ns, dp), True) for ns, dp in ((2, 2), (3, 2), (2, 3), (1, 2), (1, 3))] + \
        [("3 streams, prep + forward per step (bench.py today)", eager, None)]:
    run = f if chk is None else (lambda f=f: f.run(K))
    run()
    torch.cuda.synchronize()
    ok = "" if chk is None else (" outputs identical: %s" % all(torch.equal(a, b) for a, b in zip(ref[:min(K, 8)], f.outs[:min(K, 8)])))
    ts = timed(run)
    print("%-52s K=%d: median %.1f us (%.2f us/step), min %.1f, max %.1f%s" % (name, K, ts[10], ts[10] / K, ts[0], ts[-1], ok), flush=True)

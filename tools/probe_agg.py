#!/usr/bin/env python3
"""Diagnostic: per-workgroup phase times of the aggregate kernel from the -DGNNB_PROBE build
(make -C gnn-builder_amd/csrc probe).  Run with GNNB_HIP_LIB=gnn-builder_amd/libgnnb_hip_probe.so."""
import ctypes as C
import json
import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("GNNB_HIP_LIB", str(ROOT / "gnn-builder_amd" / "libgnnb_hip_probe.so"))
import bench  # noqa: E402
from gnnbuilder_amd import runtime, synthetic  # noqa: E402

opts = json.loads(sys.argv[1]) if len(sys.argv) > 1 else {}
w = bench.WORKLOADS["c2"]
dev = torch.device("cuda:0")
model = bench.build_model(w)
batch = synthetic.make_batch(w["shape"], w["batch"], seed=0)
cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges)
x, coo, nptr, eptr = (torch.from_numpy(a).to(dev) for a in (batch.x, batch.coo, batch.node_ptr, batch.edge_ptr))
for k, v in opts.items():
    runtime.set_option(k, v)
cm.graph_prep(coo, nptr, eptr, batch.num_nodes)
width = 128
nbuf = 9
ins = [torch.rand(batch.num_nodes, width, device=dev) for _ in range(nbuf)]
outs = [torch.empty(batch.num_nodes, width, device=dev) for _ in range(nbuf)]
for i in range(30):
    cm.aggregate("gcn", ins[i % nbuf], out=outs[i % nbuf])
torch.cuda.synchronize()
lib = runtime.load_library()
n = 8 * 8192
buf = (C.c_ulonglong * n)()
lib.gnnb_probe_read(buf, n)
a = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 4, 2).astype(np.int64)
used = a[:, 0, 0] > 0
a = a[used]
wall = a[:, :, 0].astype(np.float64) / 100.0  # us (100 MHz)
cyc = a[:, :, 1].astype(np.float64)
t0 = wall[:, 0].min()
print(f"opts={opts} workgroups={a.shape[0]}")
print(f"kernel span (first start -> last end): {wall[:, 3].max() - t0:.2f} us; last start at +{wall[:, 0].max() - t0:.2f} us")
for name, i, j in (("issue loads", 0, 1), ("wait+barrier", 1, 2), ("reduce+store", 2, 3), ("total", 0, 3)):
    d = wall[:, j] - wall[:, i]
    print(f"  {name:13s} mean {d.mean():6.2f} us  p50 {np.median(d):6.2f}  max {d.max():6.2f}")
clk = (cyc[:, 3] - cyc[:, 0]) / np.maximum(wall[:, 3] - wall[:, 0], 1e-3)
print(f"  shader clock during kernel: {np.median(clk):.0f} MHz")

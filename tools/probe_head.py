#!/usr/bin/env python3
"""Diagnostic: phase times of the fused readout kernel (probe build)."""
import ctypes as C
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

w = bench.WORKLOADS["c2"]
dev = torch.device("cuda:0")
model = bench.build_model(w)
batch = synthetic.make_batch(w["shape"], w["batch"], seed=0)
cm = runtime.CompiledModel.from_model(model, batch.num_graphs, batch.num_nodes, batch.num_edges)
bd = tuple(torch.from_numpy(a).to(dev) for a in (batch.x, batch.coo, batch.node_ptr, batch.edge_ptr))
for _ in range(10):
    cm.forward(*bd)
torch.cuda.synchronize()
lib = runtime.load_library()
n = 8 * 8192
buf = (C.c_ulonglong * n)()
lib.gnnb_probe_read(buf, n)
p = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8).astype(np.float64)
p = p[:256]
t0 = p[:, 0]
names = ["dma issue", "pool", "wait+barrier", "mlp"]
marks = [p[:, 0], p[:, 2], p[:, 4], p[:, 6], p[:, 1]]
print(f"span {(p[:, 1].max() - t0.min()) / 100:.2f} us, WG lifetime mean {((p[:, 1] - t0) / 100).mean():.2f} us, last start +{(t0.max() - t0.min()) / 100:.2f}")
for i, nm in enumerate(names):
    dlt = (marks[i + 1] - marks[i]) / 100
    print(f"  {nm:13s} mean {dlt.mean():6.2f} us  max {dlt.max():6.2f}")

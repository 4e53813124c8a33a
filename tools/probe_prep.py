#!/usr/bin/env python3
"""Diagnostic: phase times of k_graph_prep (probe build); stamps are per workgroup (wave 0)."""
import ctypes as C, os, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("GNNB_HIP_LIB", str(ROOT / "gnn-builder_amd" / "libgnnb_hip_probe.so"))
import bench
from gnnbuilder_amd import runtime, synthetic
w = bench.WORKLOADS["c2"]; dev = torch.device("cuda:0")
model = bench.build_model(w)
b = synthetic.make_batch(w["shape"], w["batch"], seed=0)
cm = runtime.CompiledModel.from_model(model, b.num_graphs, b.num_nodes, b.num_edges)
bd = tuple(torch.from_numpy(a).to(dev) for a in (b.x, b.coo, b.node_ptr, b.edge_ptr))
for _ in range(5):
    cm.graph_prep(bd[1], bd[2], bd[3], b.num_nodes)
torch.cuda.synchronize()
lib = runtime.load_library(); n = 8 * 8192
buf = (C.c_ulonglong * n)(); lib.gnnb_probe_read(buf, n)
p = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 4, 2).astype(np.float64)[:1024]
wall = p[:, :, 0] / 100.0
t0 = wall[:, 0].min()
print(f"span {wall[:, 3].max() - t0:.2f} us; last start +{wall[:, 0].max() - t0:.2f}")
for nm, i, j in (("ptr+edge fetch", 0, 1), ("count+scan", 1, 2), ("fill+stores", 2, 3), ("total", 0, 3)):
    d = wall[:, j] - wall[:, i]
    print(f"  {nm:15s} mean {d.mean():6.2f} max {d.max():6.2f}")

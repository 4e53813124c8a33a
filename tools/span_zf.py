#!/usr/bin/env python3
"""Development only (-DGNNB_ZF_ABLATE): in-kernel span (first workgroup start -> last workgroup end, wall clock) of
consecutive k_gcn2_zf launches against the launch period, i.e. what the boundary between two launches costs."""
import ctypes as C, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np, torch
import bench
from gnnbuilder_amd import runtime, synthetic
w = bench.WORKLOADS["c2"]; dev = torch.device("cuda:0")
model = bench.build_model(w)
b = synthetic.make_batch(w["shape"], w["batch"], seed=0)
cm = runtime.CompiledModel.from_model(model, b.num_graphs, b.num_nodes, b.num_edges, max_graph_nodes=int(np.diff(b.node_ptr).max()))
bd = tuple(torch.from_numpy(a).to(dev) for a in (b.x, b.coo, b.node_ptr, b.edge_ptr))
lib = runtime.load_library()
for shape in (1, 0):
    runtime.set_option("zf_shape", shape)
    cm.graph_prep(bd[1], bd[2], bd[3], int(bd[0].shape[0]))
    cm.gcn_stack_timed(bd[0], 50)
    torch.cuda.synchronize()
    lib.gnnb_zf_dbg_reset()
    us = cm.gcn_stack_timed(bd[0], 200)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 512)(); n = C.c_int()
    lib.gnnb_zf_dbg_spans(buf, C.byref(n))
    a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 2).astype(np.float64) / 100.0
    idx = list(range(50, min(n.value, 256) - 1))
    span = a[idx, 1] - a[idx, 0]
    gap = np.array([a[idx[i + 1], 0] - a[idx[i], 1] for i in range(len(idx) - 1)])
    period = np.array([a[idx[i + 1], 0] - a[idx[i], 0] for i in range(len(idx) - 1)])
    print(f"shape {shape}: events {us:.2f} us/launch; span median {np.median(span):.2f}, gap end->next start median {np.median(gap):.2f} (min {gap.min():.2f}), period {np.median(period):.2f}")
